"""Experiment: per-kernel cost of tiny C-ABI launches replayed from a torch-captured hipGraph (vs tools/exp/launch_floor.hip)."""
import sys, time
sys.path.insert(0, ".")
import torch
from emrt_amd import functional as Fn, _lib
from emrt_amd.runtime import BF16, ctx
from tests.hip_utils import init, dev

c = init(BF16)
L = _lib.lib()
a = dev(torch.randn(1024)); b = dev(torch.randn(1024)); o = dev(torch.zeros(1024))
big_a = dev(torch.randn(2 << 20)); big_o = dev(torch.zeros(2 << 20))
P = Fn.P
N = 400

def tiny():
    for _ in range(N):
        L.call("emrt_add", P(a), P(b), P(o), 1024, 1024, BF16, c.stream)

def mid():       # 4 MB in + 4 MB out
    for _ in range(N):
        L.call("emrt_add", P(big_a), P(big_a), P(big_o), 2 << 20, 2 << 20, BF16, c.stream)

def memsets():
    for _ in range(N):
        L.call("emrt_memset", P(o), 0, 2048, c.stream)

x = dev(torch.randn(8, 16, 16, 256)); y = dev(torch.zeros(8, 16, 16, 256))
sums = torch.zeros(8 * 2 * 256, dtype=torch.float64, device="cuda")
mean = torch.zeros(256, device="cuda"); inv = torch.ones(256, device="cuda"); rm = torch.zeros(256, device="cuda"); rv = torch.ones(256, device="cuda")
gam = torch.ones(256, device="cuda"); bet = torch.zeros(256, device="cuda")

def bn():
    for _ in range(N):
        L.call("emrt_bn_apply", P(x), 256, None, 0, P(y), 256, None, 1.0, 1e-5, 0.9, None, None, P(rm), P(rv), P(gam), P(bet), 8 * 16 * 16, 256, 1, BF16, c.stream)

for name, fn in (("emrt_add 1K elements", tiny), ("emrt_add 2M elements (8 MB traffic)", mid), ("emrt_memset 2 KB", memsets), ("emrt_bn_apply eval 8x16x16x256", bn)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / N * 1e6
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    gr = (time.perf_counter() - t0) / 5 / N * 1e6
    print("%-40s eager %.2f us/launch   torch hipGraph replay %.2f us/launch" % (name, eager, gr), flush=True)
