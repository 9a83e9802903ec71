"""-m gpu: `python bench.py --gpus 2` with NO launcher -- the command shape the driver uses -- must start two rank processes
by itself and print a line with n_gpus == 2 (reference: ranks from the launcher, semantic_segmentation/train.py:116-123).
The box has one GPU and RCCL refuses two ranks on one device, so the ranks share GPU 0 and talk over gloo (the test aids
EMRT_ALL_RANKS_ON_GPU0 / EMRT_DIST_BACKEND of tests/test_gpu_dp2.py); everything above the transport is the N > 1 code."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(EMRT_ALL_RANKS_ON_GPU0="1", EMRT_DIST_BACKEND="gloo")
    return e


def test_bench_gpus_2_starts_two_ranks_itself():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 16
    assert out["value"] > 0 and out["scaling"] == "weak" and out["cpu_baseline"] is None
    assert "process group: backend gloo, world size 2" in r.stderr


def test_train_gpus_2_starts_two_ranks_itself(tmp_path):
    r = subprocess.run([sys.executable, "-m", "emrt_amd.train", "--gpus", "2", "--iters", "4", "--no-eval", "--dtype", "fp32",
                        "--save_dir", str(tmp_path)], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "Total params" in r.stdout
    assert os.path.exists(os.path.join(str(tmp_path), "iter_4_model_state.pdparams"))
