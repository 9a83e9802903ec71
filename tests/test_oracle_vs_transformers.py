"""CPU: pins the oracle's restatement of the deformable transformer against the one INDEPENDENT implementation available
offline, `transformers`' Deformable-DETR (same published algorithm the reference re-implements in Paddle).

The reference ships no tests or vectors and PaddlePaddle cannot run here, so parity stays "unpinned" by rule; these tests
remove the likeliest silent restatement errors instead -- the places where transformer_encoder_decoder.py differs from a
plain reading: the offset normaliser's (W, H) flip (:98-99), reference points x valid ratios (:223-228, :467), the
softmax over L*P followed by the [L, P] reshape (:92-96), where value_proj / output_proj sit (:83, :106), the packed
in-projection of MultiHeadAttention sliced per q/k/v (layers.py:221-234), the post-norm wiring of the decoder layer
(:282-295), the sine embedding (position_encoding.py:59-75) and the pixel-centre reference grid (:213-228).
"""
import pytest
import torch

tr = pytest.importorskip("transformers.models.deformable_detr.modeling_deformable_detr")
import types      # noqa: E402

from oracle.emrt_torch import (MSDeformableAttention, TransformerDecoderLayer, TransformerEncoder,      # noqa: E402
                               sine_position_embedding)

SHAPES = [(8, 6), (4, 3), (2, 2)]      # H != W so that an (H, W) / (W, H) mix-up cannot cancel
LV = sum(h * w for h, w in SHAPES)


def _config():
    # the attributes the attention / decoder-layer modules read (DeformableDetrConfig itself wants a backbone from the hub)
    return types.SimpleNamespace(d_model=256, num_feature_levels=3, encoder_attention_heads=8, decoder_attention_heads=8,
                                 encoder_n_points=6, decoder_n_points=6, decoder_ffn_dim=1024, encoder_ffn_dim=1024,
                                 activation_function="relu", dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
                                 disable_custom_kernels=True, _attn_implementation="eager")


def _copy_msda(dst, src):
    """oracle MSDeformableAttention -> transformers DeformableDetrMultiscaleDeformableAttention (both torch [out, in])."""
    with torch.no_grad():
        for n in ("sampling_offsets", "attention_weights", "value_proj", "output_proj"):
            getattr(dst, n).weight.copy_(getattr(src, n).weight)
            getattr(dst, n).bias.copy_(getattr(src, n).bias)


def _random_msda(seed):
    torch.manual_seed(seed)
    m = MSDeformableAttention(256, 8, 3, 6)
    with torch.no_grad():      # _reset_parameters zeroes the offset / weight projections: make every path carry signal
        m.sampling_offsets.weight.normal_(0, 0.02)
        m.sampling_offsets.bias.add_(torch.randn_like(m.sampling_offsets.bias) * 0.3)
        m.attention_weights.weight.normal_(0, 0.05)
        m.attention_weights.bias.normal_(0, 0.5)
        m.value_proj.bias.normal_(0, 0.1)
        m.output_proj.bias.normal_(0, 0.1)
    return m


def _spatial():
    shapes_t = torch.tensor(SHAPES, dtype=torch.long)
    starts = torch.cat([shapes_t.new_zeros(1), (shapes_t[:, 0] * shapes_t[:, 1]).cumsum(0)[:-1]])
    return shapes_t, starts


def test_msda_module_encoder_form_matches_transformers():
    """query = src + pos, value = src, one reference point per level scaled by valid ratios (t_e_d.py:65-107, 213-228)."""
    m = _random_msda(0)
    hf = tr.DeformableDetrMultiscaleDeformableAttention(_config(), num_heads=8, n_points=6).eval()
    _copy_msda(hf, m)
    g = torch.Generator().manual_seed(1)
    B = 2
    src, pos = torch.randn(B, LV, 256, generator=g), torch.randn(B, LV, 256, generator=g)
    valid = 0.6 + 0.4 * torch.rand(B, 3, 2, generator=g)                  # non-trivial valid ratios
    ref = TransformerEncoder.get_reference_points(SHAPES, valid)           # [B, Lv, 3, 2]
    ref_hf = tr.DeformableDetrEncoder.get_reference_points(SHAPES, valid, "cpu")
    assert torch.allclose(ref, ref_hf, atol=1e-6), "reference-point grid differs"
    shapes_t, starts = _spatial()
    with torch.no_grad():
        a = m(src + pos, ref, src, SHAPES)
        b, _ = hf(hidden_states=src, encoder_hidden_states=src, position_embeddings=pos, reference_points=ref_hf,
                  spatial_shapes=shapes_t, spatial_shapes_list=SHAPES, level_start_index=starts)
    assert (a - b).abs().max().item() < 2e-5, (a - b).abs().max().item()


def test_msda_module_decoder_form_matches_transformers():
    """110 queries with ONE sigmoid reference point each, broadcast over levels through the valid ratios (t_e_d.py:466-467)."""
    m = _random_msda(2)
    hf = tr.DeformableDetrMultiscaleDeformableAttention(_config(), num_heads=8, n_points=6).eval()
    _copy_msda(hf, m)
    g = torch.Generator().manual_seed(3)
    B, Lq = 2, 110
    tgt, qpos, memory = (torch.randn(B, n, 256, generator=g) for n in (Lq, Lq, LV))
    valid = 0.6 + 0.4 * torch.rand(B, 3, 2, generator=g)
    pts = torch.sigmoid(torch.randn(B, Lq, 2, generator=g))
    ref = pts[:, :, None] * valid[:, None]                                  # [B, Lq, 3, 2] as EncoderDecoder.forward builds it
    shapes_t, starts = _spatial()
    with torch.no_grad():
        a = m(tgt + qpos, ref, memory, SHAPES)
        b, _ = hf(hidden_states=tgt, encoder_hidden_states=memory, position_embeddings=qpos, reference_points=ref,
                  spatial_shapes=shapes_t, spatial_shapes_list=SHAPES, level_start_index=starts)
    assert (a - b).abs().max().item() < 2e-5, (a - b).abs().max().item()


def test_msda_value_mask_matches_transformers():
    """padding mask zeroes projected values (t_e_d.py:84-86); EMRT passes all-ones, the arithmetic is still on the path."""
    m = _random_msda(4)
    hf = tr.DeformableDetrMultiscaleDeformableAttention(_config(), num_heads=8, n_points=6).eval()
    _copy_msda(hf, m)
    g = torch.Generator().manual_seed(5)
    src = torch.randn(1, LV, 256, generator=g)
    mask = torch.rand(1, LV, generator=g) > 0.3
    ref = TransformerEncoder.get_reference_points(SHAPES, torch.ones(1, 3, 2))
    shapes_t, starts = _spatial()
    with torch.no_grad():
        a = m(src, ref, src, SHAPES, mask)
        b, _ = hf(hidden_states=src, attention_mask=mask, encoder_hidden_states=src, reference_points=ref, spatial_shapes=shapes_t,
                  spatial_shapes_list=SHAPES, level_start_index=starts)
    assert (a - b).abs().max().item() < 2e-5


def test_decoder_layer_matches_transformers():
    """Whole TransformerDecoderLayer (t_e_d.py:282-295): packed-in-projection self-attention with q = k = tgt + query_pos and
    v = tgt, deformable cross-attention, FFN, three post-norms."""
    torch.manual_seed(6)
    layer = TransformerDecoderLayer(256, 8, 1024, 0.0, 3, 6).eval()
    with torch.no_grad():
        for p in layer.parameters():          # break zero biases / unit LayerNorm scales
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
        layer.cross_attn.sampling_offsets.weight.normal_(0, 0.02)
        layer.cross_attn.attention_weights.weight.normal_(0, 0.05)
    hf = tr.DeformableDetrDecoderLayer(_config()).eval()
    E = 256
    with torch.no_grad():
        w, b = layer.self_attn.in_proj_weight, layer.self_attn.in_proj_bias
        for i, proj in enumerate((hf.self_attn.q_proj, hf.self_attn.k_proj, hf.self_attn.v_proj)):
            proj.weight.copy_(w[i * E:(i + 1) * E])
            proj.bias.copy_(b[i * E:(i + 1) * E])
        hf.self_attn.o_proj.weight.copy_(layer.self_attn.out_proj.weight)
        hf.self_attn.o_proj.bias.copy_(layer.self_attn.out_proj.bias)
        _copy_msda(hf.encoder_attn, layer.cross_attn)
        for dst, src in ((hf.self_attn_layer_norm, layer.norm1), (hf.encoder_attn_layer_norm, layer.norm2), (hf.final_layer_norm, layer.norm3),
                         (hf.mlp.fc1, layer.linear1), (hf.mlp.fc2, layer.linear2)):
            dst.weight.copy_(src.weight)
            dst.bias.copy_(src.bias)
    g = torch.Generator().manual_seed(7)
    B, Lq = 2, 110
    tgt, qpos, memory = (torch.randn(B, n, 256, generator=g) for n in (Lq, Lq, LV))
    ref = torch.sigmoid(torch.randn(B, Lq, 1, 2, generator=g)).expand(B, Lq, 3, 2).contiguous()
    shapes_t, starts = _spatial()
    with torch.no_grad():
        a = layer(tgt, ref, memory, SHAPES, None, qpos)
        b = hf(tgt, object_queries_position_embeddings=qpos, reference_points=ref, spatial_shapes=shapes_t, spatial_shapes_list=SHAPES,
               level_start_index=starts, encoder_hidden_states=memory)
    b = b[0] if isinstance(b, tuple) else b
    assert (a - b).abs().max().item() < 5e-5, (a - b).abs().max().item()


def test_sine_position_embedding_matches_transformers():
    """position_encoding.py:59-75 (normalize=True, offset -0.5, temperature 10000, 128 features per axis), ragged mask."""
    mask = torch.ones(2, 7, 5, dtype=torch.bool)
    mask[1, 5:, :] = False
    mask[1, :, 4:] = False
    a = sine_position_embedding(mask, 128)
    hf = tr.DeformableDetrSinePositionEmbedding(128, normalize=True)
    b = hf.build_sine_position_embedding.__wrapped__(torch.Size((2, 256, 7, 5)), "cpu", torch.float32, 128, True, hf.scale, 10000, mask) \
        if hasattr(hf.build_sine_position_embedding, "__wrapped__") else hf(torch.Size((2, 256, 7, 5)), "cpu", torch.float32, mask)
    assert a.shape == b.shape == (2, 256, 7, 5)
    assert (a - b).abs().max().item() < 1e-5


@pytest.mark.parametrize("depth", [50, 18])
def test_resnet_backbone_matches_transformers(depth):
    """The oracle's ResNet (restating paddle_vision_resnet.py:43-257: 7x7/2 stem, 3x3/2 max-pool, bottlenecks with the stride on the
    3x3 conv, 1x1 strided shortcut on the first block of a stage; BasicBlock for depth 18) against transformers' ResNetModel with the
    same weights mapped name by name: the four stage outputs c1..c4 in eval mode (running statistics: randomised, so a wrong
    mean/variance mapping cannot hide), and in train mode (batch statistics)."""
    from transformers import ResNetConfig, ResNetModel
    from oracle.emrt_torch import ResNet
    torch.manual_seed(depth)
    bottleneck = depth >= 50
    cfg = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048] if bottleneck else [64, 128, 256, 512],
                       depths=ResNet.layer_cfg[depth], layer_type="bottleneck" if bottleneck else "basic", hidden_act="relu",
                       downsample_in_first_stage=False, downsample_in_bottleneck=False)
    hf = ResNetModel(cfg)
    ref = ResNet(depth)
    with torch.no_grad():
        for n, b in ref.named_buffers():          # non-trivial running statistics
            b.copy_(torch.rand_like(b) + 0.5 if n.endswith("_variance") else torch.randn_like(b) * 0.3)
        for n, p in ref.named_parameters():
            if p.dim() == 1 and "bn" in n or ".downsample.1." in n:
                p.copy_(torch.rand_like(p) + 0.5 if n.endswith("weight") else torch.randn_like(p) * 0.2)
    rs = ref.state_dict()
    mapped = {}

    def bn(dst, src):
        mapped[dst + ".weight"], mapped[dst + ".bias"] = rs[src + ".weight"], rs[src + ".bias"]
        mapped[dst + ".running_mean"], mapped[dst + ".running_var"] = rs[src + "._mean"], rs[src + "._variance"]
    mapped["embedder.embedder.convolution.weight"] = rs["conv1.weight"]
    bn("embedder.embedder.normalization", "bn1")
    for s, nblk in enumerate(ResNet.layer_cfg[depth]):
        for i in range(nblk):
            src, dst = "layer%d.%d" % (s + 1, i), "encoder.stages.%d.layers.%d" % (s, i)
            for j in range(3 if bottleneck else 2):
                mapped["%s.layer.%d.convolution.weight" % (dst, j)] = rs["%s.conv%d.weight" % (src, j + 1)]
                bn("%s.layer.%d.normalization" % (dst, j), "%s.bn%d" % (src, j + 1))
            if src + ".downsample.0.weight" in rs:
                mapped[dst + ".shortcut.convolution.weight"] = rs[src + ".downsample.0.weight"]
                bn(dst + ".shortcut.normalization", src + ".downsample.1")
    own = hf.state_dict()
    missing = [k for k in own if k not in mapped and not k.endswith("num_batches_tracked")]
    assert not missing, missing[:5]
    hf.load_state_dict({**{k: v for k, v in own.items() if k.endswith("num_batches_tracked")}, **mapped})
    assert len(mapped) == sum(1 for k in rs if not k.startswith("fc."))        # every oracle tensor but the unused fc went somewhere
    x = torch.randn(2, 3, 96, 64)
    for train in (False, True):
        ref.train(train)
        hf.train(train)
        for m in hf.modules():              # same momentum convention for the side effect; irrelevant to the outputs compared here
            if isinstance(m, torch.nn.BatchNorm2d):
                m.momentum = 0.1
        with torch.no_grad():
            got = ref(x)
            want = hf(x, output_hidden_states=True).hidden_states[1:]
        assert len(got) == len(want) == 4
        for a, b in zip(got, want):
            assert a.shape == b.shape
            assert (a - b).abs().max().item() < 1e-4 * max(1.0, b.abs().max().item()), (train, (a - b).abs().max().item())
