"""emrt_adaptive_avgpool_fwd for one set of pyramid scales (argv), to be run under rocprofv3 --kernel-trace --stats."""
import sys
sys.path.insert(0, ".")
import torch
from emrt_amd import functional as Fn
from emrt_amd.runtime import BF16
from tests.hip_utils import init
c = init(BF16)
scales = [int(v) for v in sys.argv[1:]] or [1, 3, 6, 8]
x = torch.randn(8, 32, 32, 1536, device="cuda").bfloat16()[..., :256]      # a channel slice of the concat buffer, as in the model
for _ in range(30):
    Fn.adaptive_avgpool_tokens(x, scales)
torch.cuda.synchronize()
