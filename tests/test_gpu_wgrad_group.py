"""-m gpu: batched weight gradients (emrt_conv2d_wgrad_group, csrc/conv.hip: wgrad_group_kernel).  Many conv / linear layers of different
geometry run their backward on ONE tape: every data gradient goes out alone (dw == NULL), the weight gradients are queued and launched in
groups of up to 24.  Each dW / dbias is compared with torch (fp32 CPU, same rounded inputs) and with the layer-by-layer path
(Context.wgrad_batch = 0: the pair kernel / the single-problem kernels).  Reference operators: nn.Conv2D / nn.Linear weight gradients of
loss.backward(), train.py:142-149."""
import math
import random

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from emrt_amd import _lib                        # noqa: E402
from emrt_amd import nn as hnn                    # noqa: E402
from emrt_amd.runtime import BF16, F32, Tape     # noqa: E402
from tests.hip_utils import Holder, dev_map, host, host_map, init, rnd     # noqa: E402


def _geoms(seed, n, odd=False):
    rng = random.Random(seed)
    out = []
    while len(out) < n:
        k = rng.choice([1, 1, 3, 3])
        stride = rng.choice([1, 1, 2])
        dil = rng.choice([1, 1, 2]) if k > 1 else 1
        pad = dil * (k // 2) if rng.random() < 0.8 else 0
        H, W = rng.randint(6, 36), rng.randint(6, 36)
        if H + 2 * pad - dil * (k - 1) - 1 < 0 or W + 2 * pad - dil * (k - 1) - 1 < 0:
            continue
        if odd and rng.random() < 0.3:
            Cin, Cout = rng.choice([3, 20, 24]), rng.choice([30, 48, 100])          # off the 8-element grid: launched alone (scalar loaders)
        else:
            Cin, Cout = rng.choice([64, 128, 256, 320]), rng.choice([64, 96, 128, 256, 512])
        out.append((rng.randint(1, 4), H, W, Cin, Cout, k, stride, pad, dil, rng.random() < 0.5))
    return out


_REFS = {}


def _run(geoms, dtype, batch, seed, cleared=False, passes=1):
    """-> per layer (dW, dbias, dx) on the host; all layers' backward on one tape.  cleared: the gradients are cleared through
    ParamStore.zero_grad() first, which also marks every weight gradient 'known zero' (one-slice problems then STORE their tiles);
    passes > 1 replays the same backward into the same gradients without clearing."""
    c = init(dtype)
    c.wgrad_batch = batch
    g = torch.Generator().manual_seed(seed)
    layers, xs, dys, refs = {}, [], [], []
    memo = _REFS.get((tuple(geoms), dtype, seed))      # the CPU reference of a (geometry list, seed) is computed once per process (it is most of a case's time)
    for i, (N, H, W, Cin, Cout, k, stride, pad, dil, bias) in enumerate(geoms):
        conv = hnn.Conv2D(Cin, Cout, k, stride, pad, bias=bias, dilation=dil)
        with torch.no_grad():
            conv.weight.copy_(rnd(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)))
        layers["l%d" % i] = conv
        x = rnd(torch.randn(N, Cin, H, W, generator=g))
        OHs = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
        OWs = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
        dy = rnd(torch.randn((N, Cout, OHs, OWs), generator=g))
        if memo is None:
            xr, wr = x.clone().requires_grad_(True), conv.weight.detach().clone().requires_grad_(True)
            br = torch.zeros(Cout, requires_grad=True) if bias else None
            yr = F.conv2d(xr, wr, br, stride=stride, padding=pad, dilation=dil)
            assert tuple(yr.shape) == tuple(dy.shape)
            yr.backward(dy)
            refs.append((wr.grad, br.grad if bias else None, xr.grad))
        xs.append(x)
        dys.append(dy)
    if memo is None:
        _REFS[(tuple(geoms), dtype, seed)] = refs
    else:
        refs = memo
    holder = Holder(**layers).place()
    if cleared:
        holder.store.grad.fill_(7.0)         # (whatever was there is cleared by zero_grad, not by the kernels)
        holder.store.zero_grad()
    L_ = _lib.lib()
    for _ in range(passes):
        tape = Tape()
        c.tape = tape
        xd = [dev_map(x) for x in xs]
        ys = [layers["l%d" % i](xd[i]) for i in range(len(geoms))]
        c.tape = None
        for x_ in xd:
            tape.watch(x_)
        for y, dy in zip(ys, dys):
            tape.add_grad(y, dev_map(dy))
        L_.start_record()
        tape.backward()
        rec = L_.stop_record()
    torch.cuda.synchronize()
    got = []
    for i in range(len(geoms)):
        conv = layers["l%d" % i]
        got.append((host(conv.weight.grad), host(conv.bias.grad) if conv.bias is not None else None, host_map(tape.result(xd[i]))))
    c.wgrad_batch = 24
    return got, refs, [n for n, _ in rec]


@pytest.mark.parametrize("dtype", [BF16, F32], ids=["bf16", "fp32"])
def test_batched_weight_gradients_match_torch_and_the_layerwise_path(dtype):
    geoms = _geoms(11, 9)
    got, refs, names = _run(geoms, dtype, 24, 5)
    assert names.count("emrt_conv2d_wgrad_group") == 1 and "emrt_conv2d_wgrad" not in names
    base, _, names0 = _run(geoms, dtype, 0, 5)
    assert "emrt_conv2d_wgrad_group" not in names0
    tol_w, tol_x = (1e-3, 4e-3) if dtype == BF16 else (2e-5, 2e-5)
    for i, ((dw, db, dx), (rw, rb, rx), (bw, bb, bx)) in enumerate(zip(got, refs, base)):
        rel = ((dw - rw).norm() / rw.norm()).item()
        relb = ((dw - bw).norm() / bw.norm()).item()
        relx = ((dx - rx).norm() / rx.norm()).item()
        print("layer %d %s: dW vs torch %.2e, vs layer-by-layer %.2e, dx vs torch %.2e" % (i, geoms[i], rel, relb, relx))
        assert rel < tol_w and relb < 5e-6 and relx < tol_x, (i, geoms[i], rel, relb, relx)
        # the data gradient comes from a different tile / k-split variant than inside the pair kernel: another fp32 summation order, and
        # in bf16 the occasional flipped last bit of a stored output
        assert ((dx - bx).norm() / bx.norm()).item() < (1e-3 if dtype == BF16 else 1e-5)
        if db is not None:
            assert ((db - rb).norm() / rb.norm()).item() < tol_w and ((db - bb).norm() / bb.norm()).item() < 5e-6


def test_batches_larger_than_one_launch_and_odd_shapes():
    """31 layers: two grouped launches (24 + the rest), the shapes off the vector path and the 256x256-kernel layers (forced) launched alone."""
    geoms = _geoms(23, 29, odd=True) + [(2, 16, 16, 256, 256, 3, 1, 1, 1, True), (1, 8, 8, 512, 256, 1, 1, 0, 1, False)]
    L_ = _lib.lib()
    old = L_.set_tuning("wgrad8p_force", 1)
    try:
        got, refs, names = _run(geoms, BF16, 64, 9)
    finally:
        L_.set_tuning("wgrad8p_force", old)
    assert names.count("emrt_conv2d_wgrad_group") == 1      # one C-ABI call; the library cuts it into launches
    for i, ((dw, db, dx), (rw, rb, rx)) in enumerate(zip(got, refs)):
        rel = ((dw - rw).norm() / rw.norm()).item()
        assert rel < 1e-3, (i, geoms[i], rel)
        if db is not None:
            assert ((db - rb).norm() / rb.norm()).item() < 1e-3


@pytest.mark.parametrize("knobs", [dict(wgroup_blocks=64), dict(wgroup_blocks=100000, wgroup_min_steps=1), dict(wgroup_max=3)],
                         ids=["few-long-blocks", "one-tile-blocks", "three-per-launch"])
def test_batched_weight_gradients_under_extreme_plans(knobs):
    geoms = _geoms(37, 7)
    L_ = _lib.lib()
    old = [(k, L_.set_tuning(k, v)) for k, v in knobs.items()]
    try:
        got, refs, _ = _run(geoms, BF16, 24, 3)
    finally:
        for k, v in old:
            L_.set_tuning(k, v)
    for i, ((dw, db, dx), (rw, rb, rx)) in enumerate(zip(got, refs)):
        assert ((dw - rw).norm() / rw.norm()).item() < 1e-3, (i, geoms[i])


@pytest.mark.parametrize("dtype", [BF16, F32], ids=["bf16", "fp32"])
def test_first_contribution_is_stored_and_later_ones_are_added(dtype):
    """EmrtWgradDesc.dw_is_zero: after ParamStore.zero_grad() the first weight gradient of a layer that runs as ONE pixel slice is stored
    instead of added with atomics.  Same bits as the accumulating path (0 + v == v), and a second backward without clearing must ADD."""
    geoms = _geoms(41, 12) + [(1, 6, 6, 512, 512, 3, 1, 1, 1, True), (1, 8, 8, 256, 1024, 1, 1, 0, 1, False)]     # + few pixels, large dW
    L_ = _lib.lib()
    old = L_.set_tuning("wgrad_no_overwrite", 1)
    try:
        base, refs, _ = _run(geoms, dtype, 24, 13, cleared=True)
    finally:
        L_.set_tuning("wgrad_no_overwrite", old)
    got, _, _ = _run(geoms, dtype, 24, 13, cleared=True)
    twice, _, _ = _run(geoms, dtype, 24, 13, cleared=True, passes=2)
    tol = 1e-3 if dtype == BF16 else 2e-5
    for i, ((dw, db, _), (bw, bb, _), (tw, tb, _), (rw, rb, _)) in enumerate(zip(got, base, twice, refs)):
        assert ((dw - rw).norm() / rw.norm()).item() < tol, (i, geoms[i])
        # one-slice problems are bit-identical to the accumulating path; multi-slice ones are atomics in both (order-dependent last bits)
        assert ((dw - bw).norm() / bw.norm()).item() < 5e-6, (i, geoms[i])
        assert ((tw - 2 * rw).norm() / rw.norm()).item() < 2 * tol, (i, geoms[i])
        if db is not None:
            assert ((tb - 2 * rb).norm() / rb.norm()).item() < 2 * tol


@pytest.mark.parametrize("dtype", [BF16, F32], ids=["bf16", "fp32"])
def test_a_weight_used_twice_in_one_step_sums_both_contributions(dtype):
    """A shared layer: the same conv applied to two inputs on one tape.  Its first weight gradient is queued as 'dW is zero:
    store the tile', the second as an add; stored tiles must never race with (or land after) the other use's atomic adds.  Few pixels and a
    large dW, i.e. the one-slice case where the store path is taken; the sum must equal torch's, and the same through the C-ABI with both
    problems in ONE call (the library drops the store for problems whose dw aliases another problem's)."""
    import ctypes
    from emrt_amd import functional as Fn
    c = init(dtype)
    c.wgrad_batch = 24
    g = torch.Generator().manual_seed(77)
    N, H, W, Cin, Cout = 1, 6, 6, 512, 512
    conv = hnn.Conv2D(Cin, Cout, 3, 1, 1, bias=True)
    other = hnn.Conv2D(Cin, 256, 1, 1, 0, bias=False)          # an unrelated layer between the two uses
    with torch.no_grad():
        conv.weight.copy_(rnd(torch.randn(Cout, Cin, 3, 3, generator=g) / 68))
        other.weight.copy_(rnd(torch.randn(256, Cin, 1, 1, generator=g) / 23))
    wr = conv.weight.detach().clone().requires_grad_(True)          # (host copy: place() moves the parameter into the device store)
    holder = Holder(conv=conv, other=other).place()
    xs = [rnd(torch.randn(N, Cin, H, W, generator=g)) for _ in range(3)]
    dys = [rnd(torch.randn(N, Cout, H, W, generator=g)) for _ in range(2)] + [rnd(torch.randn(N, 256, H, W, generator=g))]
    br = torch.zeros(Cout, requires_grad=True)
    for x, dy in zip(xs[:2], dys[:2]):
        F.conv2d(x, wr, br, padding=1).backward(dy)
    tol = 1e-3 if dtype == BF16 else 2e-5
    for trial in range(3):          # (a race would not show every time)
        holder.store.grad.fill_(3.0)
        holder.store.zero_grad()
        tape = Tape()
        c.tape = tape
        xd = [dev_map(x) for x in xs]
        ys = [conv(xd[0]), other(xd[2]), conv(xd[1])]
        c.tape = None
        for y, dy in zip(ys, [dys[0], dys[2], dys[1]]):
            tape.add_grad(y, dev_map(dy))
        tape.backward()
        torch.cuda.synchronize()
        rel = ((host(conv.weight.grad) - wr.grad).norm() / wr.grad.norm()).item()
        relb = ((host(conv.bias.grad) - br.grad).norm() / br.grad.norm()).item()
        assert rel < tol and relb < tol, (trial, rel, relb)
    # the C-ABI alone: two descriptors with the same dw, both claiming "dw is zero", in one call
    xd = [dev_map(x) for x in xs[:2]]
    dyd = [dev_map(dy) for dy in dys[:2]]
    for trial in range(3):
        conv.weight.grad.zero_()
        conv.bias.grad.zero_()
        arr = (Fn._WgradDesc * 2)()
        for d, x_, dy_ in zip(arr, xd, dyd):
            _, _, _, _, ldx, x_bs = Fn._check_map(x_)
            _, _, _, _, lddy, dy_bs = Fn._check_map(dy_)
            (d.x, d.dy, d.dw, d.dbias, d.N, d.H, d.W, d.C, d.ldx, d.x_bs, d.OH, d.OW, d.OC, d.lddy, d.dy_bs, d.KH, d.KW, d.stride, d.pad,
             d.dilation, d.dw_is_zero) = (x_.data_ptr(), dy_.data_ptr(), conv.weight.grad.data_ptr(), conv.bias.grad.data_ptr(), N, H, W, Cin,
                                          ldx, x_bs, H, W, Cout, lddy, dy_bs, 3, 3, 1, 1, 1, 1)
        _lib.lib().call("emrt_conv2d_wgrad_group", arr, 2, c.dtype, c.stream)
        torch.cuda.synchronize()
        rel = ((host(conv.weight.grad) - wr.grad).norm() / wr.grad.norm()).item()
        assert rel < tol, (trial, rel)


# ---- the problems of a batch that fit the 256x256 LDS-DMA kernel, grouped on it (csrc/wgrad8p.hpp: wgrad8p_group_kernel) ----------------------------
_GEOMS_256 = [
    # (N, H, W, Cin, Cout, k, stride, pad, dil, bias): C and OC multiples of 256, OH * OW multiples of 64, >= 8 steps of 64 pixels
    (8, 8, 8, 512, 512, 3, 1, 1, 1, True),         # 36 tiles x 8 steps (ResNet layer4 conv2)
    (2, 16, 16, 256, 256, 3, 1, 1, 1, False),      # 9 tiles x 8 steps
    (4, 16, 16, 1024, 256, 1, 1, 0, 1, False),     # 4 tiles x 16 steps (layer3 conv1)
    (2, 24, 32, 256, 512, 1, 1, 0, 1, True),       # 2 tiles x 24 steps, non-square map
    (4, 32, 32, 256, 256, 3, 2, 1, 1, False),      # stride 2
    (2, 16, 16, 256, 256, 3, 1, 2, 2, True),       # dilation 2
    (2, 1, 1344, 256, 1024, 1, 1, 0, 1, True),     # a linear layer over 1344 tokens: 4 tiles x 42 steps
    (2, 1, 1344, 1024, 256, 1, 1, 0, 1, False),
    # not eligible: stay on the 128x128 group kernel of the same call
    (2, 10, 10, 256, 256, 3, 1, 1, 1, True),       # 100 pixels per image
    (3, 16, 16, 128, 256, 1, 1, 0, 1, False),      # C = 128
    (1, 8, 8, 256, 256, 1, 1, 0, 1, True),         # one step
]


@pytest.mark.parametrize("knobs,cleared", [(dict(), False), (dict(), True), (dict(wgroup8_blocks=16), True), (dict(wgroup8_blocks=4096), False),
                                           (dict(wgroup8_blocks=4096, wgrad8p_slab=0), True), (dict(wgroup8_blocks=100000, wgrad8p_min_steps=1), False)],
                         ids=["default-plan-accumulate", "default-plan-dw-known-zero", "one-slice-each-stored", "many-slices-slab", "many-slices-atomics",
                              "more-slices-than-slab-tiles"])
def test_grouped_256_tile_weight_gradients(knobs, cleared):
    """Eight layers of one batch on wgrad8p_group_kernel (stored tiles / atomics / partial tiles + the grouped reduce launch, whatever the plan gives
    each), three on the 128x128 group kernel: every dW and dbias against torch and against the same call with the knob off."""
    L_ = _lib.lib()
    old0 = L_.set_tuning("wgroup8", 0)
    try:
        base, refs, _ = _run(_GEOMS_256, BF16, 24, 17, cleared=cleared)
    finally:
        L_.set_tuning("wgroup8", old0)
    old = [(k, L_.set_tuning(k, v)) for k, v in dict(wgroup8=1, wgroup8_min_work=1, **knobs).items()]
    try:
        got, _, names = _run(_GEOMS_256, BF16, 24, 17, cleared=cleared)
        twice, _, _ = _run(_GEOMS_256, BF16, 24, 17, cleared=cleared, passes=2)
    finally:
        for k, v in old:
            L_.set_tuning(k, v)
    assert names.count("emrt_conv2d_wgrad_group") == 1
    for i, ((dw, db, _), (bw, bb, _), (tw, tb, _), (rw, rb, _)) in enumerate(zip(got, base, twice, refs)):
        rel, relb = ((dw - rw).norm() / rw.norm()).item(), ((dw - bw).norm() / bw.norm()).item()
        print("layer %d %s: dW vs torch %.2e, vs the 128x128 group kernel %.2e" % (i, _GEOMS_256[i], rel, relb))
        assert rel < 1e-3 and relb < 2e-5, (i, _GEOMS_256[i], rel, relb)
        assert ((tw - 2 * rw).norm() / rw.norm()).item() < 2e-3, (i, _GEOMS_256[i])          # a second backward without clearing ADDS
        if db is not None:
            assert ((db - rb).norm() / rb.norm()).item() < 1e-3 and ((db - bb).norm() / bb.norm()).item() < 2e-5
            assert ((tb - 2 * rb).norm() / rb.norm()).item() < 2e-3


def test_grouped_256_tile_weight_gradients_leave_shared_weights_to_the_atomic_kernels():
    """A weight used twice in one call (two problems, one dw) never joins the 256x256 group (its reduce launch adds without atomics): through the C-ABI,
    both contributions plus an unrelated eligible layer."""
    import ctypes
    from emrt_amd.functional import _WgradDesc
    c = init(BF16)
    c.ensure_scratch()
    g = torch.Generator().manual_seed(5)
    N, H, W, C, OC = 4, 16, 16, 256, 256
    xs = [rnd(torch.randn(N, H, W, C, generator=g)) for _ in range(3)]
    dys = [rnd(torch.randn(N, H, W, OC, generator=g)) for _ in range(3)]
    dws = [torch.zeros(OC, 3, 3, C, device="cuda"), torch.zeros(OC, 3, 3, C, device="cuda")]
    xd, dyd = [x.cuda().bfloat16() for x in xs], [d.cuda().bfloat16() for d in dys]
    arr = (_WgradDesc * 3)()
    for i, d in enumerate(arr):
        dw = dws[0] if i < 2 else dws[1]
        d.x, d.dy, d.dw, d.dbias = xd[i].data_ptr(), dyd[i].data_ptr(), dw.data_ptr(), None
        d.N, d.H, d.W, d.C, d.ldx, d.x_bs = N, H, W, C, C, H * W * C
        d.OH, d.OW, d.OC, d.lddy, d.dy_bs = H, W, OC, OC, H * W * OC
        d.KH, d.KW, d.stride, d.pad, d.dilation, d.dw_is_zero = 3, 3, 1, 1, 1, 1
    L_ = _lib.lib()
    old = [(k, L_.set_tuning(k, v)) for k, v in dict(wgroup8=1, wgroup8_min_work=1, wgroup8_blocks=4096).items()]
    try:
        L_.call("emrt_conv2d_wgrad_group", arr, 3, BF16, c.stream)
    finally:
        for k, v in old:
            L_.set_tuning(k, v)
    torch.cuda.synchronize()

    def ref(x, dy):      # dW[oc][kh][kw][c] = sum_m dy[m][oc] * x[pix(m) + tap][c]
        xw = x.permute(0, 3, 1, 2).clone().requires_grad_(False)
        w = torch.zeros(OC, C, 3, 3, requires_grad=True)
        y = F.conv2d(xw, w, padding=1)
        y.backward(dy.permute(0, 3, 1, 2))
        return w.grad.permute(0, 2, 3, 1)
    r0 = ref(xs[0], dys[0]) + ref(xs[1], dys[1])
    r1 = ref(xs[2], dys[2])
    assert ((dws[0].cpu() - r0).norm() / r0.norm()).item() < 1e-3
    assert ((dws[1].cpu() - r1).norm() / r1.norm()).item() < 1e-3
