"""-m gpu: the 50-step training trajectory and the 512 x 512 train step against the oracle (bodies in tests/test_gpu_model.py) in a file of their own:
pytest-xdist distributes whole files and these are two to three minutes of CPU-oracle time that sat in that file's serial chain."""
import pytest

from tests.test_gpu_model import fifty_step_training_trajectory_case, large_tile_train_step_512_case

pytestmark = pytest.mark.gpu


def test_fifty_step_training_trajectory_tracks_the_oracle():
    fifty_step_training_trajectory_case()


def test_large_tile_train_step_matches_oracle_512():
    large_tile_train_step_512_case()
