"""-m gpu: the fp32 eval forward against the float64 oracle at the BASELINE configurations' real sizes (tests/test_gpu_model.py holds the body
and the smaller cases).  A file of its own: pytest-xdist distributes whole files, and these three cases are four minutes of float64 host time
that used to sit at the end of test_gpu_model.py's serial chain (the suite's critical path)."""
import pytest

from tests.test_gpu_model import FORWARD_CASES_FULL, forward_logits_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("backbone,B,S,ncls", FORWARD_CASES_FULL)
def test_forward_logits_match_oracle_eval(backbone, B, S, ncls):
    forward_logits_case(backbone, B, S, ncls)
