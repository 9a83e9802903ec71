import sys; sys.path.insert(0, '.')
import torch
from emrt_amd.runtime import ctx, F32, BF16
from emrt_amd import functional as Fn
from tests.test_gpu_model import build_pair, calibrated_oracle, perturb_sampling_offsets, make_config
from oracle import train_ref
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
g = torch.Generator().manual_seed(13)
x = torch.randn(2, 3, S, S, generator=g)
labels = torch.randint(0, 6, (2, S, S), generator=g)
# ---- (a) oracle fp32 vs fp64 gradients
ref = calibrated_oracle("resnet50", x); perturb_sampling_offsets(ref)
sd = {k: v.clone() for k, v in ref.state_dict().items()}
ref.train(); out = ref(x); train_ref.mix_softmax_ce_loss(out, labels).backward()
g32 = {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}
ref.zero_grad(); ref.load_state_dict(sd); ref.double(); out = ref(x.double()); train_ref.mix_softmax_ce_loss(out, labels).backward()
rows = sorted(((((g32[n] - p.grad.float()).norm() / (p.grad.float().norm() + 1e-12)).item(), n) for n, p in ref.named_parameters() if p.grad is not None and p.grad.norm() > 1e-9), reverse=True)
print("oracle fp32 vs fp64 gradient rel err, worst:", rows[:6], "median", rows[len(rows)//2])
# ---- (b) stage-wise bf16 vs fp32 on the HIP path (train mode, no dropout)
feats = {}
for dt in (F32, BF16):
    ref2, model = build_pair("resnet50", x, dt)
    model.train(); c = ctx(); c.training = True; c.tape = None
    if model.store.dirty: model.store.pack()
    xi = Fn.nchw_to_nhwc(x.cuda())
    c1, c2, c3, c4 = model.backbone(xi)
    B, H, W, _ = xi.shape; Sx = H // 8
    psp_cat = c.empty((B, Sx, Sx, 1536))
    xc = model.spatial_branch(xi, out=Fn.narrow(psp_cat, 3, 0, 256))
    xp = model.psp_module(xc)
    hs, mem, shapes, spans = model.model([c2, c3, c4], xp)
    out = model(x.cuda())
    feats[dt] = dict(c1=c1.float().cpu(), c2=c2.float().cpu(), c3=c3.float().cpu(), c4=c4.float().cpu(), xc=xc.float().cpu().contiguous(), xp=xp.float().cpu(), mem=mem.float().cpu(), hs=hs.float().cpu(), logits=out[0].cpu(), aux=out[1].cpu())
for k in feats[F32]:
    a, b = feats[F32][k], feats[BF16][k]
    print("%-7s rel err %.4f  cos %.5f  (|ref| rms %.3g)" % (k, ((a - b).norm() / a.norm()).item(), torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item(), a.pow(2).mean().sqrt().item()))
