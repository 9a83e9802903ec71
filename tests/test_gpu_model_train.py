"""-m gpu: the two heaviest train-mode parity cases of tests/test_gpu_model.py (which holds their bodies) in a file of their own -- pytest-xdist distributes
whole files, and these are two to three minutes of CPU-oracle time that sat in that file's serial chain (the suite's critical path)."""
import pytest

from tests.test_gpu_model import full_size_train_step_case, train_forward_and_gradients_case

pytestmark = pytest.mark.gpu


def test_train_forward_and_gradients_match_oracle():
    train_forward_and_gradients_case()


def test_full_size_train_step_matches_oracle():
    full_size_train_step_case()
