"""Launcher-less multi-GPU start (`bench.py --gpus N`, `python -m emrt_amd.train --gpus N`): the parent process starts N rank
processes itself (reference: ranks come from paddle.distributed.launch, semantic_segmentation/train.py:116-123).  CPU-side
checks of the launch logic; the 2-rank run on a GPU is tests/test_gpu_launch.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "EMRT_ALL_RANKS_ON_GPU0")}
    e.update(env)
    return subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)


def test_spawn_ranks_sets_the_rank_environment_and_relays_rank0():
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "from emrt_amd.distributed import spawn_ranks\n"
            "child = 'import os; print(os.environ[\"RANK\"], os.environ[\"LOCAL_RANK\"], os.environ[\"WORLD_SIZE\"], "
            "os.environ[\"MASTER_ADDR\"], int(os.environ[\"MASTER_PORT\"]) > 0)'\n"
            "codes, out = spawn_ranks(3, [sys.executable, '-c', child], capture_rank0=True)\n"
            "print(codes, out.strip())\n" % ROOT)
    r = _run([sys.executable, "-c", code], EMRT_ALL_RANKS_ON_GPU0="1")
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == "[0, 0, 0] 0 0 3 127.0.0.1 True"


def test_spawn_ranks_a_failing_rank_takes_the_job_down():
    code = ("import os, sys, time; sys.path.insert(0, %r)\n"
            "from emrt_amd.distributed import spawn_ranks\n"
            "child = 'import os, sys, time\\nif os.environ[\"RANK\"] == \"1\": sys.exit(7)\\ntime.sleep(120)'\n"
            "t0 = time.time()\n"
            "codes, _ = spawn_ranks(2, [sys.executable, '-c', child])\n"
            "print(codes[1], codes[0] != 0, time.time() - t0 < 60)\n" % ROOT)
    r = _run([sys.executable, "-c", code], EMRT_ALL_RANKS_ON_GPU0="1")
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == "7 True True"


def test_bench_gpus_n_refuses_more_ranks_than_gpus():
    import torch
    if torch.cuda.device_count() >= 2:
        return
    r = _run([sys.executable, "bench.py", "--gpus", "2"])
    assert r.returncode == 2 and "2 ranks requested" in r.stderr and r.stdout.strip() == ""


def test_bench_world_size_must_match_gpus():
    r = _run([sys.executable, "bench.py", "--gpus", "4"], WORLD_SIZE="2", RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE=2 but --gpus 4" in r.stderr and r.stdout.strip() == ""
    r = _run([sys.executable, "-m", "emrt_amd.train", "--gpus", "4"], WORLD_SIZE="2", RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE=2 but --gpus 4" in r.stderr
