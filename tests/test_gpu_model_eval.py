"""-m gpu: the fp32 eval forward against the float64 oracle at the small sizes and the other backbone depths (body: tests/test_gpu_model.py's
forward_logits_case).  A file of its own for pytest-xdist, which distributes whole files."""
import pytest

from tests.test_gpu_model import FORWARD_CASES_SMALL, forward_logits_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("backbone,B,S,ncls", FORWARD_CASES_SMALL)
def test_forward_logits_match_oracle_eval(backbone, B, S, ncls):
    forward_logits_case(backbone, B, S, ncls)
