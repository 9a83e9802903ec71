"""single-graph vs multi-graph engine loss traces (tests/test_gpu_model.py::test_split_graphs_keep_their_order...) under the environment's knobs"""
import os, sys
sys.path.insert(0, ".")
import torch
import torch.distributed as dist
from tests.test_gpu_model import make_config
from emrt_amd.runtime import BF16
from emrt_amd.src.models import get_model
from emrt_amd.src.models.losses import get_loss_function
from emrt_amd.src.models.solver import get_optimizer, get_scheduler
from emrt_amd.engine import TrainEngine
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
g = torch.Generator().manual_seed(3)
B, S = 8, 256
x = torch.randn(B, 3, S, S, generator=g); labels = torch.randint(0, 6, (B, S, S), generator=g)
modes = sys.argv[1:] or ["eager", "single", "split"]
for mode in modes:
    torch.manual_seed(7)
    cfg = make_config("resnet50", iters=1000)
    model = get_model(cfg); model.to_hip("cuda:0", BF16, seed=11); model.set_dropout(0.0)
    opt = get_optimizer(model, get_scheduler(cfg), cfg)
    eng = TrainEngine(model, opt, get_loss_function(cfg), 1, use_graph=(mode != "eager"), warmup_eager=1, two_phase=(mode == "split"))
    xs, ls = x.cuda(), labels.cuda()
    tr = []
    for _ in range(24):
        tr.append(eng.step(xs, ls).item())
    print(mode, " ".join("%.4f" % v for v in tr), "gradnorm %.4f" % opt.grad_norm(), flush=True)
dist.destroy_process_group()
