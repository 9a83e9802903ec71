import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# an xdist worker shares the host's cores with its siblings (the CPU oracle is most of the suite's time): without a cap every worker's torch
# starts one thread per core and four oversubscribed pools run several times SLOWER than one (measured: 725 s for 4 files against ~400 s
# in-process).  Set before any test module imports torch.
if os.environ.get("PYTEST_XDIST_WORKER"):
    _n = max(1, int(os.environ.get("PYTEST_XDIST_WORKER_COUNT", "1")))
    _threads = str(max(2, (os.cpu_count() or 8) // _n))
    os.environ.setdefault("OMP_NUM_THREADS", _threads)
    os.environ.setdefault("MKL_NUM_THREADS", _threads)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """`pytest -m gpu` is spread over 6 worker processes, FILE BY FILE (pytest-xdist, --dist loadfile): most of the suite's wall time is the
    CPU oracle (fp32 / float64 torch on the host cores) and process start-up of the multi-rank tests, not GPU time, and the files are
    independent (every test initialises the device context it needs; knobs of the library are per process).  Tests of one file still run in
    order in one process.  EMRT_TEST_WORKERS=n overrides (1 = in-process); an explicit -n on the command line wins; without pytest-xdist
    the suite runs in-process as before."""
    if hasattr(config, "workerinput") or not hasattr(config.option, "numprocesses"):
        return None
    want = os.environ.get("EMRT_TEST_WORKERS")
    if want is None and (config.option.markexpr or "").strip() == "gpu":
        want = "6"      # (4 until round 6: the suite's wall time is then ~total / 4 with a 100 s tail; the heavy files are split so that 6 balance, threads per worker = cores / 6)
    if want and int(want) > 1 and config.option.numprocesses is None and not config.option.collectonly and not config.getoption("usepdb", False):
        config.option.numprocesses = int(want)
        config.option.dist = "loadfile"
    return None


@pytest.fixture(scope="session")
def repo_root():
    return ROOT
