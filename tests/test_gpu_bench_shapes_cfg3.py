"""-m gpu: BASELINE configs[2] (batch 4, 512 x 512, 7 classes) in bf16 against the fp32 CPU oracle -- the body is tests/test_gpu_bench_shapes.py's
full_size_bf16_model_case; a file of its own because pytest-xdist distributes whole files and this one case is two to three minutes of oracle time."""
import pytest

from tests.test_gpu_bench_shapes import full_size_bf16_model_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,S,ncls", [(4, 512, 7)], ids=["cfg3-4x512"])
def test_full_size_bf16_model_vs_fp32_oracle(B, S, ncls):
    full_size_bf16_model_case(B, S, ncls)
