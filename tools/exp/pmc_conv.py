"""Experiment: run one convolution shape a few times (forward, dgrad, wgrad) for rocprofv3 --pmc counter collection."""
import sys
sys.path.insert(0, ".")
import torch
from emrt_amd import _lib
from tests.hip_utils import init
from emrt_amd.runtime import BF16

c = init(BF16)
L = _lib.lib()
P = lambda t: t.data_ptr()
N, H, W, C, OC, k, s, pad = [int(v) for v in (sys.argv[1:9] if len(sys.argv) > 8 else (8, 128, 128, 256, 256, 3, 1, 1))]
OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
dev = "cuda"
x = torch.randn(N, H, W, C, device=dev).bfloat16()
wf = (torch.randn(OC, k, k, C, device=dev) * 0.05).bfloat16()
wb = (torch.randn(C, k, k, OC, device=dev) * 0.05).bfloat16()
y = torch.randn(N, OH, OW, OC, device=dev).bfloat16()
dx = torch.empty_like(x)
dw = torch.zeros(OC, k, k, C, device=dev, dtype=torch.float32)
st = c.stream
import os
if os.environ.get("CONV_TILE"):          # force an igemm tile (7 = the 256x256 LDS-DMA 8-phase kernel)
    L.set_tuning("conv_tile", int(os.environ["CONV_TILE"]))
for _ in range(3):
    L._raw_emrt_conv2d(P(x), P(wf), P(y), None, None, N, H, W, C, C, H * W * C, OH, OW, OC, OC, OH * OW * OC, 0, 0, k, k, s, pad, 0, 0, 0, None, None, 0, 0, 1, None, 1, st)
    L._raw_emrt_conv2d(P(y), P(wb), P(dx), None, None, N, OH, OW, OC, OC, OH * OW * OC, H, W, C, C, H * W * C, 0, 0, k, k, s, pad, 1, 0, 0, None, None, 0, 0, 1, None, 1, st)
    L._raw_emrt_conv2d_wgrad(P(x), P(y), P(dw), N, H, W, C, C, H * W * C, OH, OW, OC, OC, OH * OW * OC, k, k, s, pad, None, 1, 1, st)
torch.cuda.synchronize()
print("ok")
