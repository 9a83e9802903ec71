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


def test_spawn_ranks_kills_its_ranks_when_the_parent_is_terminated(tmp_path):
    """a parent that ends early (SIGTERM, KeyboardInterrupt, an exception) must not leave ranks behind holding GPUs."""
    import signal
    import time
    pidfile = tmp_path / "pids"
    child = tmp_path / "child.py"
    child.write_text("import os, time\nopen(%r + os.environ['RANK'], 'w').write(str(os.getpid()))\ntime.sleep(300)\n" % str(pidfile))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from emrt_amd.distributed import spawn_ranks\n"
            "spawn_ranks(2, [sys.executable, %r])\n" % (ROOT, str(child)))
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    e["EMRT_ALL_RANKS_ON_GPU0"] = "1"
    parent = subprocess.Popen([sys.executable, "-c", code], cwd=ROOT, env=e)
    try:
        t0 = time.time()
        while not all(os.path.exists(str(pidfile) + r) and open(str(pidfile) + r).read() for r in "01"):
            assert time.time() - t0 < 120 and parent.poll() is None
            time.sleep(0.2)
        pids = [int(open(str(pidfile) + r).read()) for r in "01"]
        parent.send_signal(signal.SIGTERM)
        parent.wait(timeout=60)
        t0 = time.time()
        alive = pids
        while alive and time.time() - t0 < 30:
            alive = [p for p in alive if os.path.exists("/proc/%d" % p) and open("/proc/%d/stat" % p).read().split()[2] != "Z"]
            time.sleep(0.2)
        assert not alive, "ranks %s survived their parent" % alive
    finally:
        if parent.poll() is None:
            parent.kill()


def test_bench_without_gpus_flag_takes_the_launchers_world_size():
    """`torchrun ... bench.py` (no --gpus) runs with WORLD_SIZE, as emrt_amd.train does; only an explicit mismatch is an error."""
    import argparse
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    old = os.environ.get("WORLD_SIZE")
    try:
        os.environ["WORLD_SIZE"] = "4"
        a = argparse.Namespace(gpus=None)
        mod.check_world(a)
        assert a.gpus == 4
        import pytest
        with pytest.raises(SystemExit):
            mod.check_world(argparse.Namespace(gpus=2))
        del os.environ["WORLD_SIZE"]
        a = argparse.Namespace(gpus=None)
        mod.check_world(a)
        assert a.gpus == 1
    finally:
        if old is None:
            os.environ.pop("WORLD_SIZE", None)
        else:
            os.environ["WORLD_SIZE"] = old


def test_spawn_ranks_at_eight_ranks():
    """BASELINE configs[3]: 8 ranks, one per GPU of the node -- the rank environment every child gets."""
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "from emrt_amd.distributed import spawn_ranks\n"
            "child = 'import os; open(os.environ[\"OUT\"] + os.environ[\"RANK\"], \"w\").write(\" \".join(os.environ[k] for k in "
            "(\"RANK\", \"LOCAL_RANK\", \"WORLD_SIZE\", \"LOCAL_WORLD_SIZE\", \"MASTER_ADDR\", \"HSA_ENABLE_IPC_MODE_LEGACY\")))'\n"
            "codes, _ = spawn_ranks(8, [sys.executable, '-c', child])\n"
            "print(codes)\n" % ROOT)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        r = _run([sys.executable, "-c", code], EMRT_ALL_RANKS_ON_GPU0="1", OUT=os.path.join(d, "rank"))
        assert r.returncode == 0 and r.stdout.strip() == str([0] * 8), (r.stdout, r.stderr)
        got = [open(os.path.join(d, "rank%d" % i)).read() for i in range(8)]
    assert got == ["%d %d 8 8 127.0.0.1 0" % (i, i) for i in range(8)], got


def test_bench_dry_run_validates_the_eight_rank_plan_without_a_gpu():
    """`bench.py --gpus 8 --dry-run`: eight rank processes, a gloo rendezvous, and the N > 1 step's plan compared across ranks -- the three
    gradient-exchange ranges tile the trainable part of the flat buffer, the five SyncBatchNorm layers (paddle_EMRT.py:64, fcn_head.py:53)
    form the statistics group, every rank gets the same number of distinct tiles.  No GPU and no libemrt_hip.so involved."""
    import json
    r = _run([sys.executable, "bench.py", "--gpus", "8", "--dry-run"], EMRT_HIP_LIB="/nonexistent/libemrt_hip.so")
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["dry_run"] and d["ok"] and d["n_gpus"] == 8 and d["global_batch"] == 64 and not d["problems"]
    assert sum(x["elements"] for x in d["exchange_ranges"]) == d["gradient_elements"] and len(d["exchange_ranges"]) == 3
    assert 0.80 < d["exchange_ranges"][0]["elements"] / d["gradient_elements"] < 0.88          # heads + transformer + layer4 go first
    assert d["exchange_ranges"][2]["mbytes_fp32"] < 8                                          # the one exposed collective is small
    assert len(d["sync_batchnorm_layers"]) == 5 and d["sync_batchnorm_layers"][-1].startswith("auxlayer")
    # under a launcher whose size disagrees with --gpus the dry run fails before any rendezvous
    r = _run([sys.executable, "bench.py", "--gpus", "8", "--dry-run"], WORLD_SIZE="4", RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE=4 but --gpus 8" in r.stderr


def test_ride_along_child_failure_falls_back(monkeypatch, capsys):
    """bench.py measures the ride-along configs of the default command in child processes; a child that cannot run or prints no JSON line must make
    the caller fall back to the in-process measurement (return None), never break the default command."""
    import argparse
    import bench
    args = argparse.Namespace(no_cpu_baseline=True, cpu_threads=0, keep_gc=False)
    monkeypatch.setattr(sys, "executable", "/bin/false")
    assert bench._ride_along(args, "cfg3", 2, 1) is None
    monkeypatch.setattr(sys, "executable", "/nonexistent/python")
    assert bench._ride_along(args, "cfg3", 2, 1) is None
    assert "measuring in-process" in capsys.readouterr().err

