"""-m gpu: the captured bf16 training step against the eager engine and the oracle at BASELINE configs[2] (4 x 512 x 512); body in
tests/test_gpu_captured_step.py.  A file of its own for pytest-xdist (whole files are distributed; this case is ~1.5 minutes of oracle time)."""
import pytest

from tests.test_gpu_captured_step import captured_step_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,S,ncls", [(4, 512, 7)], ids=["cfg3-4x512"])
def test_captured_bf16_step_equals_eager_and_tracks_the_oracle(B, S, ncls):
    captured_step_case(B, S, ncls)
