"""CPU: host logic of the HIP path with a recording stand-in for the C-ABI (tests/fake_abi.py): the full EMRT train step
is driven through emrt_amd (model -> loss -> backward tape -> optimizer) and the launch sequence is checked for
completeness and argument sanity.  Nothing is computed; this guards shapes / strides / tape wiring without a GPU."""
import collections

import pytest
import torch

from tests import fake_abi


@pytest.fixture()
def fake():
    f = fake_abi.install()
    yield f
    fake_abi.uninstall()


def _place(model):
    from emrt_amd import nn as hnn
    from emrt_amd.runtime import ctx, F32
    from emrt_amd.src.models.emrt import NOGRAD_PARAMS
    model.store = hnn.ParamStore(model, ctx().device, F32, nograd_names=NOGRAD_PARAMS, fused_groups=model.fused_groups(),
                                 lr_mult_names=model.lr_mult_names())
    hnn.bind_all(model, model.store)
    model.store.pack()
    return model


@pytest.mark.parametrize("backbone", ["resnet18", "resnet50"])
def test_train_step_launch_sequence(fake, backbone):
    from emrt_amd.src.models.emrt import EMRT
    from emrt_amd.src.models.losses import MixSoftmaxCrossEntropyLoss
    from emrt_amd.src.models.solver import Momentum, PolynomialDecay
    torch.manual_seed(0)
    m = _place(EMRT(num_classes=6, backbone=backbone))
    st = m.store
    # flat layout: fused offsets|logits weights adjacent, no-grad parameters outside the trainable range
    for name, mod in m.named_modules():
        if type(mod).__name__ == "MSDeformableAttention":
            assert st.offsets[name + ".attention_weights.weight"] == st.offsets[name + ".sampling_offsets.weight"] + 288 * 256
            assert st.offsets[name + ".attention_weights.bias"] == st.offsets[name + ".sampling_offsets.bias"] + 288
    for n in ("backbone.fc.weight", "backbone.fc.bias", "model.tgt_embed.weight"):
        assert st.offsets[n] >= st.n_train
    assert len(st.lr_ranges) == 14 and all(b <= st.n_train for _, b in st.lr_ranges)
    if backbone == "resnet50":
        assert st.n_total - st.n_train >= 2048 * 1000 + 1000 + 110 * 256 and abs(st.n_train - 54.03e6) < 0.05e6
    x, lab = torch.randn(2, 3, 64, 64), torch.randint(0, 6, (2, 64, 64))
    m.eval()
    out = m(x)
    assert tuple(out[0].shape) == tuple(out[1].shape) == (2, 6, 64, 64) and out.tape is None
    n_eval = len(fake.calls)
    fake.calls.clear()
    m.train()
    opt = Momentum(m, PolynomialDecay(0.01, 100), 0.9, 1e-4, 1.0)
    m.clear_gradients()
    out = m(x)
    n_fwd = len(fake.calls)
    loss = MixSoftmaxCrossEntropyLoss()(out, lab)
    loss.backward()
    opt.step()
    cnt = collections.Counter(n for n, _ in fake.calls)
    assert n_fwd > n_eval                                              # training adds dropout / statistics launches
    n_grouped = sum(a[1] for n, a in fake.calls if n == "emrt_conv2d_bwd_group")     # problems inside grouped backward launches
    # (4 encoder layers + input_proj) x 3 levels, + value_proj | offsets-logits projection of the 6 deformable attentions as pairs,
    # + the q|k and v projections of the decoder's 2 softmax attentions as pairs
    # + the four pyramid-pooling branches (conv1x1 -> BatchNorm -> ReLU: functional.conv_bn_small_group) as one grouped launch each way
    # + EFP's three Conv2dBlocks level by level: conv1 of the three levels, then conv2 (+ x), grouped each way
    # forward only: conv1 and the shortcut conv of every bottleneck stage's first block share one grouped launch (functional.conv_bn_pair)
    n_pairs = sum(1 for mod in m.modules() if type(mod).__name__ == "BottleneckBlock" and mod.downsample is not None)
    # + cls_psp's second conv -> BatchNorm -> ReLU beside the auxiliary head's (two independent 3x3 stages: EMRT.forward, one rank), grouped each way
    # forward only: the spatial branch's six conv -> BatchNorm stages ride as GUESTS in the conv1 launches of six bottleneck blocks (functional.SideJobs /
    # conv_bn_many: the host's 1x1 conv1 [+ the shortcut conv of a stage's first block] first, the guest last, another input); BasicBlock backbones host none
    hosted = [a for n, a in fake.calls if n == "emrt_conv2d_group" and all(a[0][i].bn_stats for i in range(a[1])) and a[0][0].KH == 1
              and a[0][a[1] - 1].KH == 3 and a[0][a[1] - 1].inp != a[0][0].inp]
    assert len(hosted) == (6 if n_pairs else 0) and all(a[1] in (2, 3) for a in hosted)
    n_hosted_extra = sum(1 + (1 if a[1] == 2 else 0) for a in hosted)      # the guest, and a host conv1 that used to be a launch of its own
    assert n_grouped == 15 + 12 + 4 + 4 + 6 + 2 and sum(a[1] for n, a in fake.calls if n == "emrt_conv2d_group") == n_grouped + 2 * n_pairs + n_hosted_extra
    assert cnt["emrt_bn_group_apply"] == cnt["emrt_bn_group_bwd"] == 4
    assert sorted(a[1] for n, a in fake.calls if n == "emrt_bn_group_apply") == [2, 3, 3, 4]
    assert sum(1 for n, a in fake.calls if n == "emrt_bn_group_apply" for i in range(a[1]) if a[0][i].res) == 3      # "conv2(conv1(x)) + x" added by the BatchNorm launch
    pairs = [a for n, a in fake.calls if n == "emrt_conv2d_group" and a[1] == 2]
    stage_pairs = [a for a in pairs if a[0][0].bn_stats and a[0][1].bn_stats and a[0][0].inp == a[0][1].inp]          # conv1 | shortcut conv of a stage's first block
    assert len(stage_pairs) == n_pairs - sum(1 for a in hosted if a[1] == 3) and all(a[0][1].OC == 4 * a[0][0].OC for a in stage_pairs)
    head_pairs = [a for a in pairs if a[0][0].bn_stats and a[0][1].bn_stats and a[0][0].inp != a[0][1].inp and a[0][0].KH == 3]      # (KH == 1 first: a hosted guest)
    assert len(head_pairs) == 1 and head_pairs[0][0][0].KH == head_pairs[0][0][1].KH == 3 and head_pairs[0][0][0].OC == 256 and head_pairs[0][0][1].W * 2 == head_pairs[0][0][0].W
    pairs = [a for a in pairs if not (a[0][0].bn_stats and a[0][1].bn_stats)]
    msda_pairs = [a for a in pairs if a[0][1].out_f32 == 1]
    mha_pairs = [a for a in pairs if a[0][1].out_f32 == 0]
    # the decoder's two deformable attentions as pairs; the four encoder layers' projections ride in the forward launch of their layer's per-level 3x3
    # convolutions (functional.level_conv_gn(linears=): 3 levels + value_proj + offsets | logits = 5 problems), their data gradients stay a pair of their own
    assert len(msda_pairs) == 2 and all(a[0][0].out_f32 == 0 and a[0][1].OC == 432 for a in msda_pairs)
    fronts = [a for n, a in fake.calls if n == "emrt_conv2d_group" and a[1] == 5]
    assert len(fronts) == 4 and all(a[0][0].KH == a[0][1].KH == a[0][2].KH == 3 and a[0][3].KH == 1 and a[0][3].OC == 256 and a[0][3].out_f32 == 0
                                    and a[0][4].OC == 432 and a[0][4].out_f32 == 1 and a[0][3].inp == a[0][0].inp for a in fronts)
    assert sum(1 for n, a in fake.calls if n == "emrt_conv2d_bwd_group" and a[1] == 2 and a[0][1].OC == 432) == 6      # (backward: six pairs as before)
    assert len(mha_pairs) == 2 and all(a[0][0].OC == 512 and a[0][1].OC == 256 and a[0][0].W == a[0][1].W == 110 for a in mha_pairs)
    # every GEMM weight gets exactly one weight gradient: immediately (the 1x1 classifiers' one-pass backward) or in a batched
    # emrt_conv2d_wgrad_group call made while backward runs; a layer whose weight gradient is batched passes dw = NULL to its data gradient
    batched = [a[0][i] for n, a in fake.calls if n == "emrt_conv2d_wgrad_group" for i in range(a[1])]
    immediate = [a for n, a in fake.calls if n == "emrt_conv2d_bwd" and a[7] is not None]
    # (UpHead's classifier: forward and backward with its BatchNorm operand, emrt_bn_pointwise_fwd / _bwd, dW inside the one-pass backward)
    assert cnt["emrt_bn_pointwise_fwd"] == cnt["emrt_bn_pointwise_bwd"] == 1
    assert cnt["emrt_conv2d_wgrad"] == 0 and len(batched) + len(immediate) + 1 == len(st.gemms)
    assert len(immediate) == 2 and all(a[17] <= 8 for a in immediate)      # the auxiliary head's classifier (OC = num_classes) and reference_points (OC = 2)
    assert all(a[8] is None for n, a in fake.calls if n == "emrt_conv2d_bwd" and a[7] is None)      # dbias travels with the batched dW
    assert all(a[0][i].dw is None and a[0][i].dbias is None for n, a in fake.calls if n == "emrt_conv2d_bwd_group" for i in range(a[1]))
    assert 4 <= cnt["emrt_conv2d_wgrad_group"] <= 12 and max(a[1] for n, a in fake.calls if n == "emrt_conv2d_wgrad_group") <= 24
    last_bwd = max(i for i, (n, a) in enumerate(fake.calls) if n in ("emrt_conv2d_bwd", "emrt_conv2d_bwd_group"))
    first_opt = min(i for i, (n, a) in enumerate(fake.calls) if n == "emrt_grad_clip_scale")
    assert last_bwd < max(i for i, (n, a) in enumerate(fake.calls) if n == "emrt_conv2d_wgrad_group") < first_opt      # flushed before the optimizer
    n_bn = sum(1 for mod in m.modules() if type(mod).__name__ == "BatchNorm2D")
    # five BatchNorm + ReLU layers are applied by their streaming consumer's loads (functional.PendingBN): the stem's and the spatial
    # branch's two into a max-pool, UpHead's first two into a x2 resize; their backward re-derives the ReLU mask from the raw map
    n_stream = cnt["emrt_bn_maxpool_fwd"] + cnt["emrt_bn_resize_bilinear_fwd"]
    assert cnt["emrt_bn_maxpool_fwd"] == 3 and cnt["emrt_bn_resize_bilinear_fwd"] == 2 and cnt["emrt_maxpool_fwd"] == 1
    masked_x = [a for n, a in fake.calls if n == "emrt_bn_bwd_dx" and a[22] is not None]
    assert len(masked_x) == n_stream and all(a[4] is None and a[20] is None and a[21] == 0 for a in masked_x)
    assert sum(1 for n, a in fake.calls if n == "emrt_bn_bwd_reduce" and a[12] is not None and a[4] is None) == n_stream
    # ... and the shortcut BatchNorm of every stage's first block by the join that adds it (emrt_bn_apply_join)
    n_join_defer = cnt["emrt_bn_apply_join"]
    assert n_join_defer == sum(1 for mod in m.modules() if getattr(mod, "downsample", None) is not None) > 0
    # ... and the BatchNorm + ReLU between two convolutions by the consuming convolution's operand loads (emrt_conv2d_bna; the fake library says
    # "supported" for every layer): bn1 -> conv2 of every block, bn2 -> conv3 of every bottleneck (EFP's Conv2dBlocks go level by level in
    # grouped launches instead: conv_bn_small_group)
    n_blocks = sum(1 for mod in m.modules() if type(mod).__name__ in ("BasicBlock", "BottleneckBlock"))
    n_bottle = sum(1 for mod in m.modules() if type(mod).__name__ == "BottleneckBlock")
    n_conv_fused = cnt["emrt_conv2d_bna"]
    assert n_conv_fused == n_blocks + n_bottle      # (cls_psp's first BatchNorm is materialised: its consumer runs beside the auxiliary head's conv in one grouped launch)
    bna = [a for n, a in fake.calls if n == "emrt_conv2d_bna"]
    assert all(a[26] is not None and a[37] is not None and a[36] == 1 for a in bna)      # sums, a_out, ReLU
    n_psp = 4 + 6 + 2  # the pyramid-pooling branches, EFP's six BatchNorms, cls_psp's second beside the auxiliary head's: grouped apply / backward launches of their own (emrt_bn_group_apply / _bwd)
    n_defer = n_stream + 1 + n_join_defer + n_conv_fused + n_psp
    assert cnt["emrt_bn_apply"] + cnt["emrt_bn_apply_join"] == n_bn - n_defer and cnt["emrt_bn_bwd_dx"] == n_bn - n_psp and cnt["emrt_bn_stats"] == 0
    fwd_convs = [a for n, a in fake.calls if n == "emrt_conv2d" and a[22] == 0]
    bna_stats = sum(1 for a in bna if a[24] is not None)      # (emrt_conv2d_bna: bn_stats is argument 24)
    assert not any(n == "emrt_conv2d" and a[22] == 1 for n, a in fake.calls)      # data gradients go through emrt_conv2d_bwd
    dgrads = [a for n, a in fake.calls if n == "emrt_conv2d_bwd"]
    grouped_stats = sum(1 for n, a in fake.calls if n == "emrt_conv2d_group" for i in range(a[1]) if a[0][i].bn_stats)
    assert sum(1 for a in fwd_convs if a[25] is not None) + bna_stats + grouped_stats == n_bn     # forward statistics fused into the conv epilogue
    # BatchNorm -> ReLU -> conv chains: the conv's dgrad carries the ReLU mask and the BatchNorm's backward sums, and the
    # separate reduction pass only remains for the other BatchNorms (residual joins, multi-consumer outputs, no ReLU)
    n_fused = sum(1 for a in dgrads if a[24] is not None)
    # the data gradients of the layers that shared a forward launch with a guest (conv_bn_many) go through emrt_conv2d_dgrad_multi -- the same arguments as
    # emrt_conv2d_bwd's data-gradient half, fused epilogues included: the guest's waits in the tape's stash and rides in its host's launch
    multi = [a for n, a in fake.calls if n == "emrt_conv2d_dgrad_multi"]
    assert sum(1 for a in multi if a[1] == 2) == max(len(hosted) - 1, 0) and all(a[1] in (1, 2) for a in multi)      # (the branch's first conv reads the image: no data gradient)
    assert all(a[0][0].KH == 1 and a[0][1].KH == 3 and a[0][0].dx != a[0][1].dx for a in multi if a[1] == 2)      # [host conv1 / shortcut, guest 3x3]
    n_fused += sum(1 for a in multi for i in range(a[1]) if a[0][i].bn_stats)
    assert all(a[25] is not None and a[28] == 1.0 for a in dgrads if a[24] is not None)
    # FFNs: linear2's dgrad applies the dropout mask and the ReLU mask of dropout(relu(linear1)) (mask source = its input,
    # scale 1/(1-p)); no separate mask pass is left for them
    n_ffn = sum(1 for mod in m.modules() if type(mod).__name__ in ("TransformerEncoderLayer", "TransformerDecoderLayer"))
    # ... and the same for the Dropout2D in front of the aux head's classifier and of UpHead's first conv (two more dgrads with a scaled mask)
    ffn = [a for a in dgrads if a[25] is not None and a[24] is None]
    assert len(ffn) == n_ffn + 2 > 2 and all(abs(a[28] - 1.0 / 0.9) < 1e-6 and a[25].value == a[0].value for a in ffn)
    # ... and their forward dropout is drawn in linear1's epilogue (emrt_conv2d_drop): the separate dropout launches left are the two Dropout2D
    # (forward only: their backward mask rides in the consumer's data gradient, above)
    assert cnt["emrt_conv2d_drop"] == n_ffn and cnt["emrt_mask_bwd"] == 0 and cnt["emrt_dropout_fwd"] == 2
    # residual joins relu(BatchNorm(x) + residual): the dgrad of a conv that consumes the join folds the earlier
    # contributions in (addend a[32], or in place a[6]), masks with the join's output and sums against the BatchNorm INPUT
    # (stat_x a[29]); the join's backward then runs without its reduction pass (sums_vs_x a[21] of emrt_bn_bwd_dx)
    joins = [a for a in dgrads if a[29] is not None]
    assert joins and all(a[24] is not None and a[25] is not None and a[25].value == a[0].value for a in joins)
    assert any(a[32] is not None for a in joins) and all(not (a[6] and a[32] is not None) for a in dgrads)
    assert all(a[29] is None for a in dgrads if a[25] is None)
    fused_y = sum(1 for n, a in fake.calls if n == "emrt_bn_bwd_dx" and a[20] is not None)
    fused_x = sum(1 for n, a in fake.calls if n == "emrt_bn_bwd_dx" and a[21] == 1)
    assert not any(a[20] is not None and a[21] == 1 for n, a in fake.calls if n == "emrt_bn_bwd_dx")
    assert all(a[4] is None for n, a in fake.calls if n == "emrt_bn_bwd_dx" and (a[20] is not None or a[21] == 1))     # fused: dy arrives masked
    # (the classifier's one-pass backward hands its BatchNorm the (sum, sum * y) form too, without being an emrt_conv2d_bwd call)
    assert fused_y + fused_x + cnt["emrt_bn_bwd_reduce"] == n_bn - n_psp and 0 < fused_y + fused_x <= n_fused + 1
    n_join_bn = sum(1 for n, a in fake.calls if n == "emrt_bn_apply" and a[2] is not None and a[18] == 1) + cnt["emrt_bn_apply_join"]
    assert 0 < fused_x <= n_join_bn and fused_x >= n_join_bn - 4, (fused_x, n_join_bn)     # every join inside the backbone stages
    assert fused_y >= (n_bn - n_psp) // 3, (fused_y, n_bn)
    assert cnt["emrt_msda_fwd"] == cnt["emrt_msda_bwd"] == 6 and cnt["emrt_mha_fwd"] == cnt["emrt_mha_bwd"] == 2
    assert cnt["emrt_layernorm_fwd"] == cnt["emrt_layernorm_bwd"] == 14
    assert cnt["emrt_groupnorm_fwd"] == cnt["emrt_groupnorm_bwd"] == 0 and cnt["emrt_groupnorm_levels_fwd"] == cnt["emrt_groupnorm_levels_bwd"] == 5
    assert cnt["emrt_softmax_ce_pair_fwd"] == cnt["emrt_softmax_ce_pair_bwd"] == 1 and cnt["emrt_softmax_ce_fwd"] == cnt["emrt_scalar_axpby"] == 0      # main + aux head in one pass
    assert cnt["emrt_grad_clip_scale"] == cnt["emrt_sgd_momentum_step"] == cnt["emrt_pack_weights"] == 1
    # wgrad destinations are distinct slices inside the trainable gradient range
    g0 = st.grad.data_ptr()
    dws = ([a[7].value - g0 for a in immediate] + [d.dw - g0 for d in batched] +
           [a[10].value - g0 for n, a in fake.calls if n == "emrt_bn_pointwise_bwd"])
    assert len(set(dws)) == len(dws) == len(st.gemms) and all(0 <= d < 4 * st.n_train for d in dws)
    assert (sum(1 for d in batched if d.dbias) + sum(1 for a in immediate if a[8] is not None) +
            sum(1 for n, a in fake.calls if n == "emrt_bn_pointwise_bwd" and a[11] is not None)) == sum(1 for g_ in st.gemms if g_.bias is not None)
    # every device pointer handed to a conv is 2-byte aligned at least and non-null
    for n, a in fake.calls:
        if n == "emrt_conv2d":
            assert a[0].value and a[1].value and a[2].value


def test_tape_accumulates_into_views(fake):
    """alias bookkeeping: gradients of token-slab / channel-slice views land in the base buffer's gradient."""
    from emrt_amd import functional as Fn
    from emrt_amd.runtime import ctx, Tape
    c = ctx()
    tape = Tape()
    c.tape = tape
    base = c.zeros((2, 21, 8))
    v = Fn.tokens_as_map(Fn.narrow(base, 1, 5, 16), 4, 4)
    c.tape = None
    g = c.zeros((2, 4, 4, 8))
    tape.add_grad(v, g)
    bg = tape.grad(base)
    assert bg is not None and tuple(bg.shape) == (2, 21, 8)
    assert tape.grad(v).data_ptr() == bg.data_ptr() + 5 * 8 * bg.element_size()
    name, args = fake.calls[-1]
    assert name == "emrt_acc3d" and args[6:9] == (2, 16, 8) and (args[1], args[2]) == (21 * 8, 8) and (args[4], args[5]) == (16 * 8, 8)
    # a gradient handed to two targets is never modified in place through either of them
    t1, t2, shared = c.zeros((4, 8)), c.zeros((4, 8)), c.zeros((4, 8))
    tape.add_grad(t1, shared)
    tape.add_grad(t2, shared)
    tape.add_grad(t1, c.zeros((4, 8)))
    assert tape.grad(t1).data_ptr() != shared.data_ptr() and tape.grad(t2).data_ptr() == shared.data_ptr()


def test_strided_geometry_helper():
    from emrt_amd import functional as Fn
    g = torch.zeros(2, 21, 8)
    v = g.narrow(1, 4, 16)
    m = v.as_strided((2, 4, 4, 8), (168, 32, 8, 1), v.storage_offset())
    assert Fn._geom3(m, torch.zeros(2, 4, 4, 8)) == (2, 16, 8, [(168, 8), (128, 8)])
    cat = torch.zeros(2, 4, 4, 24)
    assert Fn._geom3(m, cat[..., 8:16]) == (2, 16, 8, [(168, 8), (384, 24)])
    assert Fn._geom3(torch.zeros(3, 8), torch.zeros(3, 8)) == (1, 1, 24, [(24, 24), (24, 24)])


def test_window_grid_properties():
    hyp = pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st
    from emrt_amd.src.api import infer
    from oracle import infer_ref

    @settings(max_examples=200, deadline=None)
    @given(st.integers(1, 700), st.integers(1, 700), st.integers(16, 300), st.integers(8, 400))
    def run(h, w, crop, stride):
        wins = infer.window_grid(h, w, (crop, crop), (stride, stride))
        assert wins == infer_ref.window_grid(h, w, (crop, crop), (stride, stride))
        for (a, b, c_, d) in wins:
            assert 0 <= a < c_ <= h and 0 <= b < d <= w and c_ - a <= crop and d - b <= crop
        if stride <= crop:
            cov = torch.zeros(h, w, dtype=torch.int32)
            for (a, b, c_, d) in wins:
                cov[a:c_, b:d] += 1
            assert int(cov.min()) >= 1          # full coverage whenever stride <= crop
    run()


def test_state_dict_matches_oracle_key_by_key():
    """The product's state dict has exactly the oracle's (= the reference's) keys and shapes."""
    from emrt_amd.src.models.emrt import EMRT
    from oracle.emrt_torch import EMRT as OracleEMRT
    for bb in ("resnet18", "resnet50"):
        a, b = OracleEMRT(6, bb).state_dict(), EMRT(num_classes=6, backbone=bb).state_dict()
        assert list(a.keys()) == list(b.keys()) or set(a) == set(b)
        assert all(a[k].shape == b[k].shape for k in a)
    assert len(a) == 545 and sum(v.numel() for k, v in a.items() if not k.endswith(("_mean", "_variance"))) == 56107286


@pytest.mark.parametrize("backbone", ["resnet18", "resnet50"])
def test_backward_split_keeps_early_gradient_ranges_final(fake, backbone):
    """engine.py's early gradient exchange: after backward_until_split() the flat-gradient ranges split_ranges() calls
    `early` are all-reduced while the rest of backward runs, so no launch of the rest may carry a pointer into them
    (and no launch before the split one into a `late` range)."""
    import ctypes
    from emrt_amd.src.models.emrt import EMRT, LATE_GRAD_PREFIXES
    from emrt_amd.src.models.losses import MixSoftmaxCrossEntropyLoss
    torch.manual_seed(0)
    m = _place(EMRT(num_classes=6, backbone=backbone))
    st = m.store
    early, late = st.split_ranges(LATE_GRAD_PREFIXES)
    cover = sorted(early + late)
    assert cover[0][0] == 0 and cover[-1][1] == st.n_train and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    n_late = sum(e - a for a, e in late)
    assert 0.05 < n_late / st.n_train < 0.35            # layer4 + heads + transformer hold most of the elements
    base = st.grad.data_ptr()

    def grad_offsets(calls):
        out = []
        for name, args in calls:
            for a in args:
                v = a.value if isinstance(a, ctypes.c_void_p) else None
                if v is not None and base <= v < base + 4 * st.n_total:
                    out.append((name, (v - base) // 4))
        return out

    def inside(off, ranges):
        return any(a <= off < e for a, e in ranges)

    x, lab = torch.randn(2, 3, 64, 64), torch.randint(0, 6, (2, 64, 64))
    m.train()
    m.clear_gradients()
    # the N > 1 engine's order: the spatial branch after the backbone (with one rank its conv stages ride in layer3 / layer4's launches -- functional.SideJobs --
    # and its gradients would become final in the middle of a later segment; EMRT.forward keeps the reference's order whenever world_size > 1 or sync_always)
    from emrt_amd.runtime import ctx as _ctx
    old_side, _ctx().side_branch = _ctx().side_branch, False      # (what world_size > 1 / sync_always switch off in EMRT.forward)
    try:
        out = m(x)
    finally:
        _ctx().side_branch = old_side
    loss = MixSoftmaxCrossEntropyLoss()(out, lab)
    fake.calls.clear()
    rest = loss.backward_until_split()
    first = grad_offsets(fake.calls)
    fake.calls.clear()
    rest()
    second = grad_offsets(fake.calls)
    assert len(first) > 50 and len(second) > 20
    assert all(inside(off, early) for _, off in first), [x for x in first if not inside(x[1], early)][:5]
    assert all(inside(off, late) for _, off in second), [x for x in second if not inside(x[1], late)][:5]

    # the engine's finer segmentation (marks before layer3 and before layer4): segment i only touches ranges_i
    from emrt_amd.src.models.emrt import GRAD_SEGMENT_PREFIXES
    seg_ranges = st.segment_ranges(GRAD_SEGMENT_PREFIXES)
    allr = sorted(r for seg in seg_ranges for r in seg)
    assert allr[0][0] == 0 and allr[-1][1] == st.n_train and all(a[1] == b[0] for a, b in zip(allr, allr[1:]))
    assert seg_ranges[0] == early and sum(e - a for seg in seg_ranges[1:] for a, e in seg) == sum(e - a for a, e in late)
    m.clear_gradients()
    old_side, _ctx().side_branch = _ctx().side_branch, False
    try:
        out = m(x)
    finally:
        _ctx().side_branch = old_side
    loss = MixSoftmaxCrossEntropyLoss()(out, lab)
    fake.calls.clear()
    segs = loss.backward_until_split(segments=True)
    touched = [grad_offsets(fake.calls)]
    for seg in segs:
        fake.calls.clear()
        seg()
        touched.append(grad_offsets(fake.calls))
    assert len(touched) == len(seg_ranges) == 3 and all(len(t) > 5 for t in touched)
    for i, (t, ranges) in enumerate(zip(touched, seg_ranges)):
        assert all(inside(off, ranges) for _, off in t), (i, [x for x in t if not inside(x[1], ranges)][:5])
