"""-m gpu: the fp32 eval forward against the float64 oracle at the reference's other shipped tile sizes, 224 / 384 / 448 (tests/test_gpu_model.py holds the
body and the small cases).  A file of its own: pytest-xdist distributes whole files and these three cases are up to four minutes of float64 host time."""
import pytest

from tests.test_gpu_model import FORWARD_CASES_TILES, forward_logits_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("backbone,B,S,ncls", FORWARD_CASES_TILES)
def test_forward_logits_match_oracle_eval(backbone, B, S, ncls):
    forward_logits_case(backbone, B, S, ncls)
