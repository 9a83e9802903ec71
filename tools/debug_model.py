import sys; sys.path.insert(0, '.')
import torch
from emrt_amd.runtime import F32
from emrt_amd.src.models.emrt import EMRT
torch.manual_seed(0)
m = EMRT(num_classes=6, backbone=sys.argv[1] if len(sys.argv) > 1 else "resnet18")
m.to_hip("cuda:0", F32)
m.eval()
x = torch.randn(2, 3, 64, 64)
out = m(x.cuda())
torch.cuda.synchronize()
print("forward ok", out[0].abs().max().item())
