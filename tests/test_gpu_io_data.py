"""-m gpu: the callers and data formats either side of the hot path ON the GPU path:
  * a `.pdparams` file written by the independent writer (tests/golden/make_pdparams_fixture.py, Paddle's paddle.save layout)
    loaded into the HIP model by the importer and into a fresh oracle by the fixture's own reader -> same logits
    (reference: src/utils/checkpoint.py:21-93);
  * one TileLoader batch decoded from a Potsdam-layout directory tree -> TrainEngine.step against the oracle's train step on
    the same decoded batch (reference: src/datasets/potsdam.py:50-66, transforms.py:209-270,391-478, train.py:141-159);
  * weights edited AFTER a hipGraph capture are seen by the next replay (SlidingWindowEngine / TrainEngine);
  * the optimizer checkpoint is keyed by parameter name and resumes bit-for-bit."""
import argparse
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_pdparams_fixture as fx                                          # noqa: E402

from emrt_amd.config import get_config, update_config                      # noqa: E402
from emrt_amd.engine import TrainEngine                                      # noqa: E402
from emrt_amd.runtime import BF16, F32                                       # noqa: E402
from emrt_amd.src.models import get_model                                    # noqa: E402
from emrt_amd.src.models.losses import get_loss_function                     # noqa: E402
from emrt_amd.src.models.solver import get_optimizer, get_scheduler          # noqa: E402
from oracle import train_ref                                                 # noqa: E402
from test_gpu_model import assert_argmax_match, make_config, oracle_no_dropout   # noqa: E402

CFG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "emrt_amd/configs/EMRT")


def test_independent_pdparams_file_gives_oracle_logits_on_the_hip_path(tmp_path):
    from emrt_amd.src.utils.checkpoint import load_entire_model
    from oracle.emrt_torch import EMRT as OracleEMRT
    src = fx.seeded_oracle("resnet18")
    path = str(tmp_path / "emrt_r18.pdparams")
    fx.write_fixture(path, src)
    model = get_model(make_config("resnet18"))
    assert load_entire_model(model, path) == len(model.state_dict())
    model.to_hip("cuda:0", F32)
    model.eval()
    torch.manual_seed(123)
    ref = fx.read_into_oracle(path, OracleEMRT(6, "resnet18")).eval()       # a DIFFERENT initialisation, overwritten from the file
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 3, 128, 128, generator=g)
    with torch.no_grad():
        ref.double()
        want = [t.float() for t in ref(x.double())]
    got = model(x.cuda())
    for a, b in zip(got, want):
        err = (a.cpu() - b).abs().max().item()
        assert err < 1e-3, "logits from the imported file differ from the oracle's: %g" % err
    assert_argmax_match(got[0].cpu(), want[0])


def _potsdam_tree(root, n=8, size=96):
    from PIL import Image
    rng = np.random.RandomState(3)
    for sub in ("train", "test"):
        os.makedirs(os.path.join(root, sub))
        os.makedirs(os.path.join(root, sub + "_convert_labels"))
        for i in range(n):
            Image.fromarray(rng.randint(0, 256, (size, size, 3), dtype=np.uint8)).save(os.path.join(root, sub, "%d.tif" % (7 * i + 2)))
            Image.fromarray(rng.randint(0, 6, (size, size), dtype=np.uint8)).save(os.path.join(root, sub + "_convert_labels", "%d.png" % (7 * i + 2)))


def test_tileloader_batches_train_step_matches_oracle(tmp_path):
    from emrt_amd.distributed import DistributedTileSampler
    from emrt_amd.src import transforms as T
    from emrt_amd.src.datasets import get_dataset, TileLoader
    from oracle.emrt_torch import EMRT as OracleEMRT
    root = str(tmp_path / "potsdam")
    _potsdam_tree(root)
    cfg = update_config(get_config(), argparse.Namespace(cfg=os.path.join(CFG_DIR, "EMRT_256x256_160k_potsdam.yaml")))
    cfg.DATA.DATA_PATH = root
    cfg.DATA.CROP_SIZE = [64, 64]
    cfg.DATA.BATCH_SIZE = 2
    cfg.MODEL.ENCODER.TYPE = "resnet18"
    cfg.TRAIN.ITERS = 100
    np.random.seed(4)
    ds = get_dataset(cfg, T.get_transforms(cfg), "train")
    sampler = DistributedTileSampler(len(ds), 2, 0, 1, shuffle=True, drop_last=True, seed=1)
    batches = TileLoader(ds, sampler, torch.device("cuda", 0), workers=2, prefetch=2).epochs()
    torch.manual_seed(31)
    ref = OracleEMRT(6, "resnet18")
    oracle_no_dropout(ref)
    ref.train()
    model = get_model(cfg)
    model.load_state_dict(ref.state_dict())
    model.to_hip("cuda:0", F32)
    model.set_dropout(0.0)
    opt = get_optimizer(model, get_scheduler(cfg), cfg)
    eng = TrainEngine(model, opt, get_loss_function(cfg), 1, use_graph=False)
    ropt = train_ref.MomentumRef(list(ref.named_parameters()), 0.9, 1e-4, 1.0)
    for step in range(3):
        bx, by = next(batches)
        assert bx.is_cuda and tuple(bx.shape) == (2, 3, 64, 64) and bx.dtype == torch.float32 and by.dtype == torch.int64
        assert set(by.unique().tolist()) <= set(range(6)) | {255}          # padded crop borders carry the ignore label
        want_loss, lr = train_ref.train_step(ref, ropt, bx.cpu(), by.cpu(), step, 0.01, 0.0, 100, 0.9)
        loss = eng.step(bx, by).item()
        assert abs(loss - want_loss) < 2e-3 * max(1.0, want_loss), (step, loss, want_loss)
        assert abs(opt.grad_norm() - ropt.last_grad_norm) < 2e-2 * ropt.last_grad_norm, (step, opt.grad_norm(), ropt.last_grad_norm)
    batches.close()
    # after three optimizer steps on loader batches the weights still agree (the clipped step is lr * |v| ~ 1e-2 at most)
    refp = dict(ref.named_parameters())
    num = den = 0.0
    for n, p in model.named_parameters():
        d = p.detach().cpu().double() - refp[n].detach().double()
        num += float((d * d).sum())
        den += float((refp[n].detach().double() ** 2).sum())
    assert (num / den) ** 0.5 < 1e-4, (num / den) ** 0.5


@pytest.mark.parametrize("dtype", [BF16, F32])
def test_weights_loaded_after_capture_are_seen_by_the_replay(dtype):
    """the compute-dtype weight mirror is refreshed by EMRT.__call__, which does not run on graph.replay();
    a load_state_dict after the capture must still reach the replayed kernels."""
    from emrt_amd.src.api.infer import SlidingWindowEngine
    torch.manual_seed(2)
    model = get_model(make_config("resnet18"))
    model.to_hip("cuda:0", dtype)
    model.eval()
    g = torch.Generator().manual_seed(3)
    img = torch.randn(3, 128, 128, generator=g).cuda()
    eng = SlidingWindowEngine(model, (3, 128, 128), (64, 64), (64, 64), 6, warmup=1)
    eng(img)
    eng(img)                                    # captured here
    assert eng.graph is not None
    before = eng.logits.clone()
    torch.manual_seed(77)
    other = get_model(make_config("resnet18"))  # different weights
    model.load_state_dict(other.state_dict())
    eng(img)                                    # replay
    after = eng.logits.clone()
    fresh = SlidingWindowEngine(model, (3, 128, 128), (64, 64), (64, 64), 6, warmup=1)
    fresh(img)                                  # eager forward with the new weights
    assert (after - before).abs().max().item() > 1e-3, "the replay still used the old weights"
    assert torch.equal(after, fresh.logits), (after - fresh.logits).abs().max().item()


def test_train_engine_sees_weights_loaded_after_capture_and_optimizer_state_resumes():
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 3, 64, 64, generator=g).cuda()
    labels = torch.randint(0, 6, (2, 64, 64), generator=g).cuda()
    cfg = make_config("resnet18", iters=100)

    def build(seed):
        torch.manual_seed(seed)
        m = get_model(cfg)
        m.to_hip("cuda:0", BF16, seed=5)
        m.set_dropout(0.0)
        o = get_optimizer(m, get_scheduler(cfg), cfg)
        return m, o, TrainEngine(m, o, get_loss_function(cfg), 1, use_graph=True, warmup_eager=1)

    m1, o1, e1 = build(1)
    for _ in range(3):
        e1.step(x, labels)                      # eager, capture, replay
    state = {k: v.clone() for k, v in m1.state_dict().items()}
    ostate = o1.state_dict()
    assert ostate["format"] == "per-parameter" and set(ostate["velocity"]) == set(m1.store.train_order)
    assert tuple(ostate["velocity"]["backbone.conv1.weight"].shape) == (64, 3, 7, 7)     # logical shape, not the padded flat slice
    l_next = [e1.step(x, labels).item() for _ in range(2)]
    # a second engine, captured on OTHER weights, then handed the checkpoint: must continue exactly like the first
    m2, o2, e2 = build(2)
    for _ in range(3):
        e2.step(x, labels)
    m2.load_state_dict(state)
    o2.set_state_dict(ostate)
    l_resumed = [e2.step(x, labels).item() for _ in range(2)]
    # (weight gradients are summed with fp32 atomics: run-to-run differences of ~1e-7 relative are expected, nothing larger)
    assert all(abs(a - b) < 1e-4 * max(1.0, abs(b)) for a, b in zip(l_resumed, l_next)), (l_resumed, l_next)
    # an old flat-buffer checkpoint of another layout is refused with a clear message
    with pytest.raises(ValueError, match="layout"):
        o2.set_state_dict({"velocity": torch.zeros(17), "step": 0})
