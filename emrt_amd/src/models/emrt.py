"""EMRT on the MI355X HIP path: ResNet backbone -> multiscale deformable-attention encoder -> cross-attention decoder
-> segmentation head, NHWC / [B, L, C] end to end, every arithmetic op a libemrt_hip.so kernel.

Mirrors the reference model's module tree and state-dict keys:
  /root/reference/semantic_segmentation/src/models/paddle_EMRT.py:13-304  (Conv2dBlock, EFP, PyramidPoolingModule,
      branch_block, spatial_branch, UpHead, EMRT)
  .../EMRT_utils/transformer_encoder_decoder.py:21-473 (MSDeformableAttention, Transformer*Layer, EncoderDecoder)
  .../EMRT_utils/layers.py:144-311 (MultiHeadAttention), .../EMRT_utils/position_encoding.py:59-75 (sine embedding)
  .../backbones/paddle_vision_resnet.py:43-257 (ResNet), .../decoders/fcn_head.py:19-81 (FCNHead)

What is different from the reference by design (not by result):
  * no NCHW<->[B,L,C] transposes/concats: a level's NHWC map IS its token slab, `memory` is three slabs back to back,
    the 1536-channel concat buffer is written slice-wise by its producers (paddle_EMRT.py:266-293 disappears);
  * sampling_offsets | attention_weights are one 256->432 GEMM whose fp32 output feeds the fused MSDA kernel;
  * sine position embedding and encoder reference points are constants of (H, W) and are cached per shape
    (the reference recomputes them and syncs to the host ~40 times per forward);
  * Linear weights are stored [out, in]; MHA in_proj_weight [3E, E] (Paddle: transposed) -- see INTEGRATION.md.
"""
import ctypes
import math

import torch
import torch.nn as tnn

from ... import functional as Fn
from ... import nn as hnn
from ...runtime import ctx, Tape, F32

NOGRAD_PARAMS = ("backbone.fc.weight", "backbone.fc.bias", "model.tgt_embed.weight")
# parameters whose gradient is only final after the ops recorded before the tape's split mark (ResNet.forward) have run
# backward; everything else can be exchanged between ranks while those run (engine.TrainEngine, two-phase step)
# The image enters the network as an 8-channel NHWC map (channels 3..7 zero) and the three convolutions that read it (ResNet stem,
# resnet50c deep stem, spatial_branch.Enc0) store their weights with 8 input channels: their forward and weight-gradient GEMMs then
# take the 16-byte operand path (element-wise operand assembly at C = 3 cost 0.2 ms of an 10.6 ms step for 1 % of the FLOPs).
IMAGE_CHANNELS = 8

LATE_GRAD_PREFIXES = ("backbone.conv1.", "backbone.bn1.", "backbone.layer1.", "backbone.layer2.", "backbone.layer3.")   # (conv1.* also covers resnet50c's conv1.0 / .3 / .6)
# the same per backward segment (two marks: before layer3 and before layer4): what the 2nd / 3rd segment complete
GRAD_SEGMENT_PREFIXES = (("backbone.layer3.",), ("backbone.conv1.", "backbone.bn1.", "backbone.layer1.", "backbone.layer2."))


def _salt():
    return ctx().next_salt()


# ---------------------------------------------------------------------------------------------------
# ResNet (paddle_vision_resnet.py)
# ---------------------------------------------------------------------------------------------------
class BasicBlock(hnn.HipLayer):  # :43-88
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = hnn.Conv2D(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = hnn.BatchNorm2D(planes)
        self.conv2 = hnn.Conv2D(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = hnn.BatchNorm2D(planes)
        self.downsample = downsample

    def forward(self, x):
        out = Fn.conv_bn(self.conv1, self.bn1, x, relu=True, defer="conv")      # BatchNorm + ReLU applied by conv2's operand loads
        # (the shortcut's BatchNorm is applied by the join's loads: Fn.batch_norm with a PendingBN residual)
        identity = x if self.downsample is None else Fn.conv_bn(self.downsample[0], self.downsample[1], x, defer="join")
        return Fn.conv_bn(self.conv2, self.bn2, out, relu=True, residual=identity)


class BottleneckBlock(hnn.HipLayer):  # :91-149
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = hnn.Conv2D(inplanes, planes, 1, bias=False)
        self.bn1 = hnn.BatchNorm2D(planes)
        self.conv2 = hnn.Conv2D(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = hnn.BatchNorm2D(planes)
        self.conv3 = hnn.Conv2D(planes, planes * 4, 1, bias=False)
        self.bn3 = hnn.BatchNorm2D(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        # bn1 -> relu -> conv2 and bn2 -> relu -> conv3 (paddle_vision_resnet.py:129-149): each BatchNorm + ReLU is applied by the NEXT convolution's
        # operand loads (Fn.conv2d on a PendingBN: emrt_conv2d_bna), which writes the normalised map backward needs on the way
        pair = None
        c = ctx()
        side = c.side.pending() if c.side is not None else None
        if side is not None:
            # a layer3 / layer4 block (<= 256 tiles of 64 x 64): the spatial branch's pending conv -> BatchNorm stage rides in conv1's launch (Fn.conv_bn_many)
            items = [(self.conv1, self.bn1, x, True, "conv", None)]
            if self.downsample is not None:
                items.append((self.downsample[0], self.downsample[1], x, False, "join", None))
            res = Fn.conv_bn_many(items + [side], host_tiles=c.side_host_tiles)
            if res is not None:
                pair = (res[0], res[1] if self.downsample is not None else None)
                c.side.deliver(res[-1])
        if pair is None and self.downsample is not None:
            # conv1 and the stage's shortcut conv read the same x: one grouped forward launch (Fn.conv_bn_pair)
            pair = Fn.conv_bn_pair([(self.conv1, self.bn1, True, "conv"), (self.downsample[0], self.downsample[1], False, "join")], x)
        if pair is not None:
            out, identity = pair
        else:
            out = Fn.conv_bn(self.conv1, self.bn1, x, relu=True, defer="conv")
            identity = None
        out = Fn.conv_bn(self.conv2, self.bn2, out, relu=True, defer="conv")
        if identity is None:
            identity = x if self.downsample is None else Fn.conv_bn(self.downsample[0], self.downsample[1], x, defer="join")
        return Fn.conv_bn(self.conv3, self.bn3, out, relu=True, residual=identity)


class ResNet(hnn.HipLayer):  # :152-257
    layer_cfg = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}

    def __init__(self, depth, num_classes=1000):
        super().__init__()
        block = BasicBlock if depth in (18, 34) else BottleneckBlock
        layers = self.layer_cfg[depth]
        self.inplanes = 64
        self.conv1 = hnn.Conv2D(3, 64, 7, 2, 3, bias=False, need_dx=False, pad_cin=IMAGE_CHANNELS)
        self.bn1 = hnn.BatchNorm2D(64)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], 2)
        self.layer3 = self._make_layer(block, 256, layers[2], 2)
        self.layer4 = self._make_layer(block, 512, layers[3], 2)
        self.fc = hnn.Linear(512 * block.expansion, num_classes)   # in the reference state dict, never used (:213-214)
        self.fc.standalone = False

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = hnn.Sequential(hnn.Conv2D(self.inplanes, planes * block.expansion, 1, stride, 0, bias=False),
                                        hnn.BatchNorm2D(planes * block.expansion))
        mods = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            mods.append(block(self.inplanes, planes))
        return hnn.Sequential(*mods)

    def forward(self, x):
        x = Fn.conv_bn(self.conv1, self.bn1, x, relu=True, defer=True)      # BatchNorm + ReLU applied by the max-pool's loads
        x = Fn.maxpool(x, 3, 2, 1)
        feats = []
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            if (layer is self.layer3 or layer is self.layer4) and ctx().tape is not None:
                ctx().tape.splits.append(len(ctx().tape.ops))     # engine.py: the gradients of everything recorded from here
                                                                   # on are final once backward is back at this point
            for blk in layer._modules.values():
                x = blk(x)
            feats.append(x)
        return feats


class BottleneckV1b(hnn.HipLayer):  # backbones/resnet.py:61-99 (stride and dilation on conv2, padding = dilation)
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None):
        super().__init__()
        self.conv1 = hnn.Conv2D(inplanes, planes, 1, bias=False)
        self.bn1 = hnn.BatchNorm2D(planes)
        self.conv2 = hnn.Conv2D(planes, planes, 3, stride, dilation, bias=False, dilation=dilation)
        self.bn2 = hnn.BatchNorm2D(planes)
        self.conv3 = hnn.Conv2D(planes, planes * 4, 1, bias=False)
        self.bn3 = hnn.BatchNorm2D(planes * 4)
        self.downsample = downsample

    forward = BottleneckBlock.forward


class ResNetV1c(hnn.HipLayer):  # backbones/resnet.py:102-221 with deep_stem=True (resnet50c, :224-234), multi_grid off
    def __init__(self, layers=(3, 4, 6, 3), output_stride=32, num_classes=1000):
        super().__init__()
        if output_stride not in (8, 16, 32):
            raise NotImplementedError
        dilations, strides = {32: ([1, 1], [2, 2]), 16: ([1, 2], [2, 1]), 8: ([2, 4], [1, 1])}[output_stride]
        self.inplanes = 128
        # nn.Sequential(conv, bn, relu, conv, bn, relu, conv): state-dict indices 0, 1, 3, 4, 6 (:124-134)
        self.conv1 = hnn.Sequential(hnn.Conv2D(3, 64, 3, 2, 1, bias=False, need_dx=False, pad_cin=IMAGE_CHANNELS), hnn.BatchNorm2D(64), None,
                                    hnn.Conv2D(64, 64, 3, 1, 1, bias=False), hnn.BatchNorm2D(64), None,
                                    hnn.Conv2D(64, 128, 3, 1, 1, bias=False))
        self.bn1 = hnn.BatchNorm2D(128)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], stride=2)
        self.layer3 = self._make_layer(256, layers[2], stride=strides[0], dilation=dilations[0])
        self.layer4 = self._make_layer(512, layers[3], stride=strides[1], dilation=dilations[1])
        self.fc = hnn.Linear(2048, num_classes)
        self.fc.standalone = False

    def _make_layer(self, planes, blocks, stride=1, dilation=1):  # :175-207
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = hnn.Sequential(hnn.Conv2D(self.inplanes, planes * 4, 1, stride, 0, bias=False), hnn.BatchNorm2D(planes * 4))
        if dilation not in (1, 2, 4):
            raise RuntimeError("=> unknown dilation size: {}".format(dilation))
        mods = [BottleneckV1b(self.inplanes, planes, stride, dilation=2 if dilation == 4 else 1, downsample=downsample)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            mods.append(BottleneckV1b(self.inplanes, planes, dilation=dilation))
        return hnn.Sequential(*mods)

    def forward(self, x):  # :209-221
        x = Fn.conv_bn(self.conv1[0], self.conv1[1], x, relu=True)
        x = Fn.conv_bn(self.conv1[3], self.conv1[4], x, relu=True)
        x = Fn.conv_bn(self.conv1[6], self.bn1, x, relu=True, defer=True)
        x = Fn.maxpool(x, 3, 2, 1)
        feats = []
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            if (layer is self.layer3 or layer is self.layer4) and ctx().tape is not None:
                ctx().tape.splits.append(len(ctx().tape.ops))
            for blk in layer._modules.values():
                x = blk(x)
            feats.append(x)
        return feats


# ---------------------------------------------------------------------------------------------------
# FCNHead (fcn_head.py:19-81)
# ---------------------------------------------------------------------------------------------------
class FCNHead(hnn.HipLayer):
    def __init__(self, in_channels, channels, num_classes, dropout_ratio=0.1, up_ratio=16):
        super().__init__()
        self.up_ratio, self.p = up_ratio, dropout_ratio
        self.convs = hnn.Sequential(hnn.Sequential(hnn.Conv2D(in_channels, channels, 3, 1, 1, bias=False),
                                                   hnn.BatchNorm2D(channels, sync=True), None))
        self.conv_seg = hnn.Conv2D(channels, num_classes, 1)
        self.salt = _salt()

    def forward(self, x, features=None):
        """features: the output of convs[0] (conv -> SyncBN -> ReLU) when the caller already ran it inside a statistics group
        (EMRT.forward at N > 1: one all-reduce for the pyramid-pooling branches and this head)."""
        N, H, W, _ = x.shape
        o = features if features is not None else Fn.conv_bn(self.convs[0][0], self.convs[0][1], x, relu=True)
        o = Fn.dropout(o, self.p, self.salt, mode=1, hw=H * W, sole_consumer_is_linear=True)       # conv_seg's data gradient applies the mask (y > 0, 1 / (1 - p))
        o = self.conv_seg(o)
        return Fn.resize_bilinear(o, H * self.up_ratio, W * self.up_ratio, False, out_nchw_f32=True)


# ---------------------------------------------------------------------------------------------------
# Deformable transformer (transformer_encoder_decoder.py)
# ---------------------------------------------------------------------------------------------------
class MSDeformableAttention(hnn.HipLayer):  # :21-107
    def __init__(self, embed_dim=256, num_heads=8, num_levels=3, num_points=6):
        super().__init__()
        self.embed_dim, self.num_heads, self.num_levels, self.num_points = embed_dim, num_heads, num_levels, num_points
        tp = num_heads * num_levels * num_points
        self.total_points = tp
        self.sampling_offsets = hnn.Linear(embed_dim, tp * 2)      # lr_mult 0.1 (:36-38)
        self.attention_weights = hnn.Linear(embed_dim, tp)
        self.value_proj = hnn.Linear(embed_dim, embed_dim)
        self.output_proj = hnn.Linear(embed_dim, embed_dim)
        self.sampling_offsets.standalone = False                    # fused with attention_weights: one 256 -> 3*tp GEMM
        self.attention_weights.standalone = False
        self.offw_gemm = None
        self._reset_parameters()

    @torch.no_grad()
    def _reset_parameters(self):  # :46-63
        tnn.init.zeros_(self.sampling_offsets.weight)
        thetas = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = grid / grid.abs().max(-1, keepdim=True)[0]
        grid = grid.reshape(self.num_heads, 1, 1, 2).repeat(1, self.num_levels, self.num_points, 1)
        grid = grid * torch.arange(1, self.num_points + 1, dtype=torch.float32).reshape(1, 1, -1, 1)
        self.sampling_offsets.bias.copy_(grid.flatten())
        tnn.init.zeros_(self.attention_weights.weight)
        tnn.init.zeros_(self.attention_weights.bias)
        tnn.init.xavier_uniform_(self.value_proj.weight)
        tnn.init.zeros_(self.value_proj.bias)
        tnn.init.xavier_uniform_(self.output_proj.weight)
        tnn.init.zeros_(self.output_proj.bias)

    def bind(self, store, prefix):
        ow, ob = store.offsets[prefix + "sampling_offsets.weight"], store.offsets[prefix + "sampling_offsets.bias"]
        assert store.offsets[prefix + "attention_weights.weight"] == ow + self.total_points * 2 * self.embed_dim
        assert store.offsets[prefix + "attention_weights.bias"] == ob + self.total_points * 2
        self.offw_gemm = store.make_gemm(ow, self.total_points * 3, self.embed_dim, 1, 1, ob)

    def forward(self, query, reference_points, value, spatial_shapes, need_dref=False, projected=None):  # :65-107 (value_mask is all ones)
        """projected = (value_proj(value), offsets | logits projection of the query) when the caller already ran both (the encoder layer puts them into
        its conv branch's grouped launch: Fn.level_conv_gn(linears=))"""
        c = ctx()
        if projected is not None:
            value, offw = projected
        elif c.group_attn_proj and value.dim() == 3 and query.dim() == 3 and value.is_contiguous() and query.is_contiguous() and value is not query:
            # value_proj(value) and the offsets | logits projection of the query: independent, one grouped launch forward, one for both data gradients
            value, offw = Fn.linear_group([(value, self.value_proj.gw, False), (query, self.offw_gemm, True)])
        else:
            value = self.value_proj(value)
            offw = Fn.linear(query, self.offw_gemm, out_f32=True)      # [B, Lq, 2*tp offsets | tp logits], fp32
        out = Fn.msda(value, offw, reference_points, spatial_shapes, self.num_heads, self.num_points, need_dref=need_dref)
        return self.output_proj(out)


class MultiHeadAttention(hnn.HipLayer):  # layers.py:144-311
    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.in_proj_weight = tnn.Parameter(torch.empty(3 * embed_dim, embed_dim))   # rows: q | k | v
        self.in_proj_bias = tnn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = hnn.Linear(embed_dim, embed_dim)
        self.salt = _salt()
        with torch.no_grad():  # layers.py:214-219 (xavier over the packed [E, 3E] matrix)
            bound = math.sqrt(6.0 / (embed_dim + 3 * embed_dim))
            self.in_proj_weight.uniform_(-bound, bound)
            tnn.init.xavier_uniform_(self.out_proj.weight)
            tnn.init.zeros_(self.out_proj.bias)

    def bind(self, store, prefix):
        E = self.embed_dim
        w, b = store.offsets[prefix + "in_proj_weight"], store.offsets[prefix + "in_proj_bias"]
        self.qk_gemm = store.make_gemm(w, 2 * E, E, 1, 1, b)
        self.v_gemm = store.make_gemm(w + 2 * E * E, E, E, 1, 1, b + 2 * E)

    def forward(self, qk_in, v_in):  # :236-311 with query == key
        if ctx().group_attn_proj and qk_in.is_contiguous() and v_in.is_contiguous() and qk_in.dim() == 3:
            # q | k projection of (tgt + query_pos) and v projection of tgt: two independent 110-row GEMMs, one launch each way
            qk, v = Fn.linear_group([(qk_in, self.qk_gemm, False), (v_in, self.v_gemm, False)])
        else:
            qk = Fn.linear(qk_in, self.qk_gemm)
            v = Fn.linear(v_in, self.v_gemm)
        out = Fn.mha(qk, v, self.num_heads, self.dropout, self.salt)
        return self.out_proj(out)


def _linear_init_(m):  # initializer.py:267-270 (Paddle weight [in, out] => bound 1/sqrt(in))
    bound = 1 / math.sqrt(m.weight.shape[1])
    tnn.init.uniform_(m.weight, -bound, bound)
    tnn.init.uniform_(m.bias, -bound, bound)


@torch.no_grad()
def _paddle_conv_default_(conv):
    fan = conv.weight.shape[1] * conv.weight.shape[2] * conv.weight.shape[3]
    conv.weight.normal_(0.0, math.sqrt(2.0 / fan))
    if conv.bias is not None:
        conv.bias.zero_()


class TransformerEncoderLayer(hnn.HipLayer):  # :109-204
    def __init__(self, d_model=256, n_head=8, dim_feedforward=1024, dropout=0.1, n_levels=3, n_points=6):
        super().__init__()
        self.p = dropout
        self.self_attn = MSDeformableAttention(d_model, n_head, n_levels, n_points)
        self.norm1 = hnn.LayerNorm(d_model)
        self.linear1 = hnn.Linear(d_model, dim_feedforward)
        self.linear2 = hnn.Linear(dim_feedforward, d_model)
        self.norm2 = hnn.LayerNorm(d_model)
        self.conv0 = hnn.Sequential(hnn.Conv2D(d_model, d_model, 3, 1, 1, bias=False), hnn.GroupNorm(32, d_model), None)
        self.conv1 = hnn.Sequential(hnn.Conv2D(d_model, d_model, 3, 1, 1, bias=False), hnn.GroupNorm(32, d_model), None)
        self.conv2 = hnn.Sequential(hnn.Conv2D(d_model, d_model, 3, 1, 1, bias=False), hnn.GroupNorm(32, d_model), None)
        self.salts = [_salt(), _salt(), _salt()]
        with torch.no_grad():  # :148-152
            for seq in (self.conv0, self.conv1, self.conv2):
                _paddle_conv_default_(seq[0])
            _linear_init_(self.linear1)
            _linear_init_(self.linear2)
            tnn.init.xavier_uniform_(self.linear1.weight)
            tnn.init.xavier_uniform_(self.linear2.weight)

    def forward(self, src, reference_points, spatial_shapes, level_spans, pos, pos_bgrad, q=None, next_q=False):  # :184-204
        """q: with_pos_embed(src, pos) when the previous layer's norm2 already wrote it (Fn.layer_norm(q_pos=)); next_q: this layer's norm2 writes
        the next layer's query -> returns (out, q_next)."""
        B, Lv, C = src.shape
        seqs = (self.conv0, self.conv1, self.conv2)
        projected = None
        if src.is_contiguous() and all(n == h * w and n <= 4096 for (h, w), (_, n) in zip(spatial_shapes, level_spans)):
            # all levels in one grouped conv launch + one multi-level GroupNorm launch (Fn.level_conv_gn)
            gws = [seq[0].gw for seq in seqs]
            gns = [(seq[1].weight.data, seq[1].bias.data, seq[1].weight.grad, seq[1].bias.grad) for seq in seqs]
            c = ctx()
            if q is None:
                q = Fn.add(src, pos, period=Lv * C, bgrad=pos_bgrad)
            lin = [(src, self.self_attn.value_proj.gw, False), (q, self.self_attn.offw_gemm, True)]
            if c.enc_front and c.group_attn_proj and q.is_contiguous() and q is not src and Fn.level_conv_gn_takes_linears(src, gws, lin):
                # ... and the attention's two input projections, which read the same tokens, in that launch too (forward)
                src_flatten, projected = Fn.level_conv_gn(src, gws, gns, spatial_shapes, level_spans, linears=lin)
            else:
                src_flatten = Fn.level_conv_gn(src, gws, gns, spatial_shapes, level_spans)
            identity = src_flatten          # = branch(src) + src, consumed only by norm2 below: norm1's backward sums its gradient into d src
        else:
            identity = None
            src_flatten = ctx().empty((B, Lv, C))
            for (h, w), (s0, n), seq in zip(spatial_shapes, level_spans, seqs):
                x_l = Fn.tokens_as_map(Fn.narrow(src, 1, s0, n), h, w)
                y = seq[0](x_l)
                seq[1](y, gelu=True, residual=x_l, out=Fn.tokens_as_map(Fn.narrow(src_flatten, 1, s0, n), h, w))
        if q is None:
            q = Fn.add(src, pos, period=Lv * C, bgrad=pos_bgrad)
        src2 = self.self_attn(q, reference_points, src, spatial_shapes, projected=projected)
        src = self.norm1(src, src2, drop_p=self.p, drop_salt=self.salts[0], identity_from=identity)       # LN(src + dropout1(src2)), dropout inside the LN kernels
        h1 = self.linear1(src, relu=True, drop=(self.p, self.salts[1]))     # dropout(relu(linear1)) in one launch; both masks are applied by linear2's dgrad
        ff = self.linear2(h1)
        # LN(src + dropout(ffn)) + conv-branch tokens (:202-203) [+ the next layer's query in the same launch]
        return self.norm2(src, ff, post=src_flatten, drop_p=self.p, drop_salt=self.salts[2], q_pos=pos if next_q else None, q_bgrad=pos_bgrad)


class TransformerEncoder(hnn.HipLayer):  # :207-239
    def __init__(self, make_layer, num_layers):
        super().__init__()
        first = make_layer()
        layers = [first]
        for _ in range(1, num_layers):
            lyr = make_layer()
            lyr.load_state_dict(first.state_dict())   # _get_clones deep-copies one layer (utils.py:31-32)
            layers.append(lyr)
        self.layers = tnn.ModuleList(layers)


class TransformerDecoderLayer(hnn.HipLayer):  # :242-295
    def __init__(self, d_model=256, n_head=8, dim_feedforward=1024, dropout=0.1, n_levels=3, n_points=6):
        super().__init__()
        self.p = dropout
        self.self_attn = MultiHeadAttention(d_model, n_head, dropout=dropout)
        self.norm1 = hnn.LayerNorm(d_model)
        self.cross_attn = MSDeformableAttention(d_model, n_head, n_levels, n_points)
        self.norm2 = hnn.LayerNorm(d_model)
        self.linear1 = hnn.Linear(d_model, dim_feedforward)
        self.linear2 = hnn.Linear(dim_feedforward, d_model)
        self.norm3 = hnn.LayerNorm(d_model)
        self.salts = [_salt(), _salt(), _salt(), _salt()]
        with torch.no_grad():  # :267-271
            _linear_init_(self.linear1)
            _linear_init_(self.linear2)
            tnn.init.xavier_uniform_(self.linear1.weight)
            tnn.init.xavier_uniform_(self.linear2.weight)

    def forward(self, tgt, reference_points, memory, spatial_shapes, query_pos, qpos_bgrad, q=None, next_q=False):  # :282-295
        """q / next_q: as TransformerEncoderLayer.forward (the queries tgt + query_pos written by the LayerNorm launches that produce tgt)."""
        Lq, C = tgt.shape[1], tgt.shape[2]
        fuse = ctx().ln_query
        if q is None:
            q = Fn.add(tgt, query_pos, period=Lq * C, bgrad=qpos_bgrad)
        if fuse:
            tgt, q2 = self.norm1(tgt, self.self_attn(q, tgt), drop_p=self.p, drop_salt=self.salts[0], q_pos=query_pos, q_bgrad=qpos_bgrad)
        else:
            tgt = self.norm1(tgt, self.self_attn(q, tgt), drop_p=self.p, drop_salt=self.salts[0])        # dropout inside the LN kernels
            q2 = Fn.add(tgt, query_pos, period=Lq * C, bgrad=qpos_bgrad)
        tgt2 = self.cross_attn(q2, reference_points, memory, spatial_shapes, need_dref=True)
        tgt = self.norm2(tgt, tgt2, drop_p=self.p, drop_salt=self.salts[1])
        h1 = self.linear1(tgt, relu=True, drop=(self.p, self.salts[2]))
        return self.norm3(tgt, self.linear2(h1), drop_p=self.p, drop_salt=self.salts[3], q_pos=query_pos if next_q else None, q_bgrad=qpos_bgrad)


class TransformerDecoder(hnn.HipLayer):  # :298-334
    def __init__(self, make_layer, num_layers):
        super().__init__()
        first = make_layer()
        layers = [first]
        for _ in range(1, num_layers):
            lyr = make_layer()
            lyr.load_state_dict(first.state_dict())
            layers.append(lyr)
        self.layers = tnn.ModuleList(layers)


def sine_position_embedding(h, w, num_pos_feats=128, temperature=10000, offset=-0.5, eps=1e-6, scale=2 * math.pi):
    """position_encoding.py:59-75 for an all-ones mask -> [h*w, 2*num_pos_feats] (host, fp32)."""
    y_embed = torch.arange(1, h + 1, dtype=torch.float32).reshape(h, 1).expand(h, w)
    x_embed = torch.arange(1, w + 1, dtype=torch.float32).reshape(1, w).expand(h, w)
    y_embed = (y_embed + offset) / (y_embed[-1:, :] + eps) * scale
    x_embed = (x_embed + offset) / (x_embed[:, -1:] + eps) * scale
    dim_t = 2 * (torch.arange(num_pos_feats) // 2).to(torch.float32)
    dim_t = temperature ** (dim_t / num_pos_feats)
    pos_x = x_embed.unsqueeze(-1) / dim_t
    pos_y = y_embed.unsqueeze(-1) / dim_t
    pos_x = torch.stack((pos_x[:, :, 0::2].sin(), pos_x[:, :, 1::2].cos()), dim=3).flatten(2)
    pos_y = torch.stack((pos_y[:, :, 0::2].sin(), pos_y[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((pos_y, pos_x), dim=2).reshape(h * w, 2 * num_pos_feats)


def encoder_reference_points(spatial_shapes):
    """transformer_encoder_decoder.py:213-228 with valid_ratios == 1 -> [1, Lv, 1, 2] (x, y), identical for every level."""
    pts = []
    for (H, W) in spatial_shapes:
        ref_y, ref_x = torch.meshgrid(torch.linspace(0.5, H - 0.5, H), torch.linspace(0.5, W - 0.5, W), indexing="ij")
        pts.append(torch.stack((ref_x.flatten() / W, ref_y.flatten() / H), dim=-1))
    return torch.cat(pts, 0).reshape(1, -1, 1, 2).contiguous()


class EncoderDecoder(hnn.HipLayer):  # :337-473
    def __init__(self, num_queries=110, backbone_num_channels=(512, 1024, 2048), num_feature_levels=3,
                 num_encoder_points=6, num_decoder_points=6, hidden_dim=256, nhead=8, num_encoder_layers=4,
                 num_decoder_layers=2, dim_feedforward=1024, dropout=0.1):
        super().__init__()
        self.hidden_dim, self.nhead, self.num_queries = hidden_dim, nhead, num_queries
        self.encoder = TransformerEncoder(lambda: TransformerEncoderLayer(hidden_dim, nhead, dim_feedforward, dropout,
                                                                          num_feature_levels, num_encoder_points), num_encoder_layers)
        self.decoder = TransformerDecoder(lambda: TransformerDecoderLayer(hidden_dim, nhead, dim_feedforward, dropout,
                                                                          num_feature_levels, num_decoder_points), num_decoder_layers)
        self.level_embed = hnn.Embedding(num_feature_levels, hidden_dim)
        self.tgt_embed = hnn.Embedding(num_queries, hidden_dim)         # created, never used (:368, :469)
        self.query_pos_embed = hnn.Embedding(num_queries, hidden_dim)
        self.reference_points = hnn.Linear(hidden_dim, 2)               # lr_mult 0.1 (:371-372)
        self.input_proj = tnn.ModuleList([hnn.Sequential(hnn.Conv2D(c, hidden_dim, 1), hnn.GroupNorm(32, hidden_dim))
                                          for c in backbone_num_channels])
        self._const_cache = {}
        with torch.no_grad():  # :394-402
            tnn.init.normal_(self.level_embed.weight)
            tnn.init.normal_(self.tgt_embed.weight)
            tnn.init.normal_(self.query_pos_embed.weight)
            tnn.init.xavier_uniform_(self.reference_points.weight)
            tnn.init.zeros_(self.reference_points.bias)
            for l in self.input_proj:
                tnn.init.xavier_uniform_(l[0].weight)
                tnn.init.zeros_(l[0].bias)

    def _constants(self, spatial_shapes):
        key = (tuple(spatial_shapes), ctx().dtype)
        if key not in self._const_cache:
            dev = ctx().device
            sine = torch.cat([sine_position_embedding(h, w, self.hidden_dim // 2) for h, w in spatial_shapes], 0)
            self._const_cache[key] = (sine.to(device=dev, dtype=ctx().tdtype).contiguous(),
                                      encoder_reference_points(spatial_shapes).to(dev))
        return self._const_cache[key]

    def forward(self, src_feats, src_psp):  # :416-473
        c = ctx()
        B = src_feats[0].shape[0]
        C = self.hidden_dim
        spatial_shapes = [(f.shape[1], f.shape[2]) for f in src_feats]
        spans, s0 = [], 0
        for h, w in spatial_shapes:
            spans.append((s0, h * w))
            s0 += h * w
        Lv = s0
        if all(f.is_contiguous() and n <= 4096 for f, (_, n) in zip(src_feats, spans)) and len(src_feats) <= 4:
            # 1x1 conv + bias and GroupNorm of every level: one grouped conv launch + one multi-level GroupNorm launch
            src, _ = Fn.level_proj_gn(list(src_feats), [p_[0].gw for p_ in self.input_proj],
                                      [(p_[1].weight.data, p_[1].bias.data, p_[1].weight.grad, p_[1].bias.grad) for p_ in self.input_proj])
        else:
            src = c.empty((B, Lv, C))
            for f, (h, w), (a, n), proj in zip(src_feats, spatial_shapes, spans, self.input_proj):
                y = proj[0](f)                                                       # 1x1 conv + bias
                proj[1](y, out=Fn.tokens_as_map(Fn.narrow(src, 1, a, n), h, w))      # GroupNorm straight into the token slab
        sine, ref_enc = self._constants(spatial_shapes)
        pos = c.empty((Lv, C))
        lvl = self.level_embed.weight
        if len(spans) <= 4 and lvl.data.is_contiguous():                         # pos = sine + level_embed[l]  (:447-448), all levels in one launch
            starts = (ctypes.c_int * len(spans))(*[a for a, _ in spans])
            Fn._L().call("emrt_add_f32row_levels", Fn.P(sine), Fn.P(lvl.data), Fn.P(pos), starts, len(spans), Lv, C, c.dtype, c.stream)
        else:
            for l, (a, n) in enumerate(spans):
                Fn._L().call("emrt_add_f32row", Fn.P(sine[a:a + n]), Fn.P(lvl.data[l]), Fn.P(pos[a:a + n]), n * C, C, c.dtype, c.stream)

        # d level_embed[l] = sum over layers, batch and the level's tokens of the query gradients: the layers' gradients are
        # first summed (one add each) and reduced once, by the tape entry below, which runs after every layer's backward
        pos_acc = []

        def pos_bgrad(g):
            pos_acc.append(g)          # kept until pos_reduce: ONE launch sums every layer's query gradient per level (emrt_colsum_levels_multi)

        if c.tape is not None:
            def pos_reduce():
                if not pos_acc:
                    return
                if (len(pos_acc) <= 8 and len(spans) <= 4 and all(g.is_contiguous() and tuple(g.shape) == (B, Lv, C) and g.dtype == pos_acc[0].dtype for g in pos_acc)
                        and lvl.grad.is_contiguous() and C % 4 == 0 and 256 % (C // 4) == 0):
                    ptrs = (ctypes.c_void_p * len(pos_acc))(*[g.data_ptr() for g in pos_acc])
                    st_ = (ctypes.c_int * len(spans))(*[a for a, _ in spans])
                    cn_ = (ctypes.c_int * len(spans))(*[n for _, n in spans])
                    Fn._L().call("emrt_colsum_levels_multi", ptrs, len(pos_acc), st_, cn_, len(spans), B, Lv, C, Fn.P(lvl.grad), Fn.dtype_of(pos_acc[0]), c.stream)
                else:
                    for g in pos_acc:
                        for l, (a, n) in enumerate(spans):
                            Fn.colsum_acc(g.narrow(1, a, n), lvl.grad[l])
                pos_acc.clear()
            c.tape.record(pos_reduce)

        memory, q = src, None
        n_enc = len(self.encoder.layers)
        for i, layer in enumerate(self.encoder.layers):
            nq = c.ln_query and i + 1 < n_enc
            res = layer(memory, ref_enc, spatial_shapes, spans, pos, pos_bgrad, q=q, next_q=nq)
            memory, q = res if nq else (res, None)

        qpe = self.query_pos_embed.weight
        query_pos = Fn.param_input(qpe.data, qpe.grad)                           # [110, C] in the compute dtype

        def qpos_bgrad(g):  # d query_pos_embed += sum over batch of the broadcast-add gradient
            Fn.colsum_acc(g.reshape(g.shape[0], -1), qpe.grad.view(-1))

        ref_logit = self.reference_points(query_pos, out_f32=True)               # [110, 2] fp32   (:466)
        ref_dec = Fn.view_as(Fn.sigmoid_f32(ref_logit), (1, self.num_queries, 1, 2))   # one point per query, all levels
        tgt, q = src_psp, None
        n_dec = len(self.decoder.layers)
        for i, layer in enumerate(self.decoder.layers):
            nq = c.ln_query and i + 1 < n_dec
            res = layer(tgt, ref_dec, memory, spatial_shapes, query_pos, qpos_bgrad, q=q, next_q=nq)
            tgt, q = res if nq else (res, None)
        return tgt, memory, spatial_shapes, spans


# ---------------------------------------------------------------------------------------------------
# EMRT parts (paddle_EMRT.py)
# ---------------------------------------------------------------------------------------------------
class Conv2dBlock(hnn.HipLayer):  # :13-29
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = hnn.Sequential(hnn.Conv2D(cin, cout, 3, 1, 1, bias=False), hnn.BatchNorm2D(cout), None)
        self.conv2 = hnn.Sequential(hnn.Conv2D(cout, cout, 3, 1, 1, bias=False), hnn.BatchNorm2D(cout), None)

    def forward(self, x):
        o = Fn.conv_bn(self.conv1[0], self.conv1[1], x, relu=True, defer="conv")
        o = Fn.conv_bn(self.conv2[0], self.conv2[1], o, relu=True)
        return Fn.add_maps(o, x)


class EFP(hnn.HipLayer):  # :31-48
    def __init__(self, cin=256, cout=256):
        super().__init__()
        self.conv0, self.conv1, self.conv2 = Conv2dBlock(cin, cout), Conv2dBlock(cin, cout), Conv2dBlock(cin, cout)

    def forward(self, x0, x1, x2, out):
        blocks, xs = (self.conv0, self.conv1, self.conv2), (x0, x1, x2)
        c1, b1 = [b.conv1[0] for b in blocks], [b.conv1[1] for b in blocks]
        c2, b2 = [b.conv2[0] for b in blocks], [b.conv2[1] for b in blocks]
        if Fn._small_group_ok(c1, b1, xs):
            # the three Conv2dBlocks level by level in grouped launches: conv1 of all levels | BatchNorm + ReLU | conv2 | BatchNorm + ReLU + x
            mid = Fn.conv_bn_small_group(c1, b1, xs, relu=True)
            if Fn._small_group_ok(c2, b2, mid, post_adds=xs):
                o0, o1, o2 = Fn.conv_bn_small_group(c2, b2, mid, relu=True, post_adds=xs)
            else:
                o0, o1, o2 = [Fn.add_maps(Fn.conv_bn(cv, b, m_, relu=True), x) for cv, b, m_, x in zip(c2, b2, mid, xs)]
            x21 = Fn.resize_bilinear(o2, x1.shape[1], x1.shape[2], True, add_t=o1)
            return Fn.resize_bilinear(x21, x0.shape[1], x0.shape[2], True, add_t=o0, out=out)
        o2 = self.conv2(x2)
        o1 = self.conv1(x1)
        x21 = Fn.resize_bilinear(o2, x1.shape[1], x1.shape[2], True, add_t=o1)            # conv1(x1) + up(conv2(x2))
        o0 = self.conv0(x0)
        return Fn.resize_bilinear(x21, x0.shape[1], x0.shape[2], True, add_t=o0, out=out)  # conv0(x0) + up(.)


class PyramidPoolingModule(hnn.HipLayer):  # :50-78
    def __init__(self, pool_scales, in_channels, channels):
        super().__init__()
        self.pool_scales = list(pool_scales)
        self.pool_branches = tnn.ModuleList([
            hnn.Sequential(None, hnn.Conv2D(in_channels, channels, 1, bias=False), hnn.BatchNorm2D(channels, sync=True), None)
            for _ in pool_scales])

    def forward(self, x, extra=None):
        """extra = (conv, SyncBatchNorm, input) of ANOTHER conv -> SyncBN -> ReLU stage (the auxiliary head's, fcn_head.py:47-56) that joins
        the group: its statistics travel in the same all-reduce as the four pooling branches'.  Returns tokens, or (tokens, extra output)."""
        tokens = Fn.adaptive_avgpool_tokens(x, self.pool_scales)       # all four pools in one launch -> [B, 110, C]
        slices, s0 = [], 0
        for k in self.pool_scales:
            slices.append(Fn.narrow(tokens, 1, s0, k * k))
            s0 += k * k
        convs, bns = [br[1] for br in self.pool_branches], [br[2] for br in self.pool_branches]
        if extra is not None:
            convs, bns, slices = convs + [extra[0]], bns + [extra[1]], slices + [extra[2]]
        # the SyncBatchNorm branches share one statistics all-reduce per direction (Fn.conv_bn_group)
        parts = Fn.conv_bn_group(convs, bns, slices, relu=True)
        if extra is not None:
            return Fn.concat_tokens(parts[:-1]), parts[-1]
        return Fn.concat_tokens(parts)


class branch_block(hnn.HipLayer):  # :80-97
    def __init__(self, cin, cout, first=False):
        super().__init__()
        self.first = first
        self.encode = hnn.Sequential(hnn.Conv2D(cin, cout, 3, 1, 1, bias=False, need_dx=not first, pad_cin=IMAGE_CHANNELS if first else None),
                                     hnn.BatchNorm2D(cout), None,
                                     hnn.Conv2D(cout, cout, 3, 1, 1, bias=False), hnn.BatchNorm2D(cout), None)

    def forward(self, x, out=None):
        x = Fn.maxpool(x, 3, 2, 1, need_dx=not self.first)
        x = Fn.conv_bn(self.encode[0], self.encode[1], x, relu=True)
        # (without `out` the only consumer is the next block's max-pool: its loads apply this BatchNorm + ReLU)
        return Fn.conv_bn(self.encode[3], self.encode[4], x, relu=True, out=out, defer=out is None)


class spatial_branch(hnn.HipLayer):  # :99-113
    def __init__(self, in_channels=3):
        super().__init__()
        self.Enc0, self.Enc1, self.Enc2 = branch_block(in_channels, 64, first=True), branch_block(64, 128), branch_block(128, 256)

    def forward(self, x, out):
        return self.Enc2(self.Enc1(self.Enc0(x)), out=out)

    def jobs(self, x, out):
        """forward() as a generator for Fn.SideJobs: the same launches in the same order, but every conv -> BatchNorm stage is REQUESTED (yielded) so that the
        ResNet's layer3 / layer4 blocks can take it into their own under-filled launches -- the spatial branch depends on nothing the backbone computes"""
        for blk, o in ((self.Enc0, None), (self.Enc1, None), (self.Enc2, out)):
            x = Fn.maxpool(x, 3, 2, 1, need_dx=not blk.first)
            x = yield (blk.encode[0], blk.encode[1], x, True, False, None)
            x = yield (blk.encode[3], blk.encode[4], x, True, o is None, o)
        return x


class UpHead(hnn.HipLayer):  # :115-181 (num_conv == 3)
    def __init__(self, embed_dim=256, num_classes=6):
        super().__init__()
        self.conv_0 = hnn.Conv2D(embed_dim, 256, 3, 1, 1)
        self.conv_1 = hnn.Conv2D(256, 256, 3, 1, 1)
        self.conv_2 = hnn.Conv2D(256, 256, 3, 1, 1)
        self.conv_3 = hnn.Conv2D(256, num_classes, 1)
        self.syncbn_fc_0, self.syncbn_fc_1, self.syncbn_fc_2 = (hnn.BatchNorm2D(256, after=self.conv_0), hnn.BatchNorm2D(256, after=self.conv_1),
                                                                hnn.BatchNorm2D(256, after=self.conv_2))

    def forward(self, x):  # :164-180
        x = Fn.conv_bn(self.conv_0, self.syncbn_fc_0, x, relu=True, defer=True)      # BatchNorm + ReLU applied by the resize's loads
        x = Fn.resize_bilinear(x, 2 * x.shape[1], 2 * x.shape[2], False)
        x = Fn.conv_bn(self.conv_1, self.syncbn_fc_1, x, relu=True, defer=True)
        x = Fn.resize_bilinear(x, 2 * x.shape[1], 2 * x.shape[2], False)
        # (the classifier's loads apply the last BatchNorm + ReLU when it is one of the thin shapes: <= 8 classes)
        x = Fn.conv_bn(self.conv_2, self.syncbn_fc_2, x, relu=True, defer=Fn.pointwise_takes_pending(self.conv_3.gw, 256))
        x = self.conv_3(x)
        return Fn.resize_bilinear(x, 2 * x.shape[1], 2 * x.shape[2], False, out_nchw_f32=True)


class LogitsTuple(tuple):
    """(logits, aux_logits) as the reference returns them (paddle_EMRT.py:304) + the tape that produced them."""
    tape = None


class EMRT(hnn.HipLayer):  # :184-304
    def __init__(self, config=None, num_classes=None, backbone=None):
        super().__init__()
        if config is not None:
            num_classes = config.DATA.NUM_CLASSES
            backbone = config.MODEL.ENCODER.TYPE.lower()
        depth = int(backbone.replace("resnet", "")) if backbone.startswith("resnet") and backbone[6:].isdigit() else None
        if backbone == "resnet50c":
            depth = 50
        if depth not in ResNet.layer_cfg:
            raise NotImplementedError("EMRT HIP path supports resnet18/34/50/50c/101/152 backbones, got %r" % backbone)
        output_stride = int(config.MODEL.OUTPUT_STRIDE) if config is not None else 32
        if backbone == "resnet50c" and output_stride == 8:
            # the auxiliary head's x16 upsample (fcn_head.py:80) then lands on 2H x 2W and the reference's final resize to the
            # input size (paddle_EMRT.py:301) is no longer the identity this path drops; no EMRT yaml uses OUTPUT_STRIDE 8
            raise NotImplementedError("MODEL.OUTPUT_STRIDE 8 is not supported by the EMRT HIP path (16 and 32 are)")
        if backbone == "resnet50c" and config is not None:
            # options the reference's ResNetV1 constructor reads (backbones/resnet.py:106-124,175-207) that this path does not build: refused, not
            # ignored -- a silently different backbone would still "run".  No EMRT yaml sets them ("resnet50c options").
            enc = config.MODEL.ENCODER
            if bool(getattr(enc, "MULTI_GRID", False)) or getattr(enc, "MULTI_DILATION", None) not in (None, [], ()):
                raise NotImplementedError("MODEL.ENCODER.MULTI_GRID / MULTI_DILATION (backbones/resnet.py:186-199) are not supported by the EMRT HIP path")
            if float(getattr(config.MODEL, "BACKBONE_SCALE", 1.0)) != 1.0:
                raise NotImplementedError("MODEL.BACKBONE_SCALE != 1.0 (backbones/resnet.py:108,122-124) is not supported by the EMRT HIP path: "
                                          "EMRT hard-codes the stage widths [512, 1024, 2048] (paddle_EMRT.py:188-192)")
        self.nclass = num_classes
        # resnet18/34: build-side extension (the reference hard-codes [512,1024,2048], paddle_EMRT.py:188-192)
        self.backbone_num_channels = [128, 256, 512] if depth in (18, 34) else [512, 1024, 2048]
        self.hidden_dim = 256
        self.psp_scale = [1, 3, 6, 8]
        self.spatial_branch = spatial_branch(3)
        self.psp_module = PyramidPoolingModule(self.psp_scale, 256, 256)
        self.uphead = UpHead(256, num_classes)
        self.cls_psp = hnn.Sequential(hnn.Conv2D(256 * (2 + len(self.psp_scale)), 512, 3, 1, 1, bias=False), hnn.BatchNorm2D(512), None,
                                      hnn.Conv2D(512, 256, 3, 1, 1, bias=False), hnn.BatchNorm2D(256), None, None)
        self.cls_salt = _salt()
        self.cls_p = 0.1
        self.EFP = EFP(256, 256)
        self.auxlayer = FCNHead(self.backbone_num_channels[1], self.backbone_num_channels[1] // 4, num_classes)
        with torch.no_grad():  # :217-225 (before the backbone / transformer exist)
            for m in self.modules():
                if isinstance(m, hnn.Conv2D):
                    tnn.init.kaiming_normal_(m.weight, a=0, mode="fan_in", nonlinearity="relu")
        if backbone == "resnet50c":      # :227-228: deep stem, dilation from MODEL.OUTPUT_STRIDE (backbones/resnet.py)
            self.backbone = ResNetV1c(output_stride=output_stride)
            with torch.no_grad():        # resnet.py:151-160: KaimingNormal on every conv
                for m in self.backbone.modules():
                    if isinstance(m, hnn.Conv2D):
                        tnn.init.kaiming_normal_(m.weight, a=0, mode="fan_in", nonlinearity="relu")
        else:
            self.backbone = ResNet(depth)
            with torch.no_grad():   # reference downloads ImageNet weights (:231-232); offline => Paddle default init
                for m in self.backbone.modules():
                    if isinstance(m, hnn.Conv2D):
                        _paddle_conv_default_(m)
        with torch.no_grad():
            tnn.init.xavier_uniform_(self.backbone.fc.weight)
            tnn.init.zeros_(self.backbone.fc.bias)
        self.model = EncoderDecoder(backbone_num_channels=self.backbone_num_channels, hidden_dim=256, dim_feedforward=1024,
                                    dropout=0.1, num_feature_levels=3, nhead=8, num_encoder_layers=4, num_decoder_layers=2,
                                    num_encoder_points=6, num_decoder_points=6)
        self.store = None
        self.compute_aux_in_eval = True   # the reference always evaluates the aux head (paddle_EMRT.py:300-302)

    # ---- device placement -----------------------------------------------------------------------
    def fused_groups(self):
        groups = []
        for name, m in self.named_modules():
            if isinstance(m, MSDeformableAttention):
                groups.append([name + ".sampling_offsets.weight", name + ".attention_weights.weight"])
                groups.append([name + ".sampling_offsets.bias", name + ".attention_weights.bias"])
        return groups

    def lr_mult_names(self):
        names = []
        for name, m in self.named_modules():
            if isinstance(m, MSDeformableAttention):
                names += [name + ".sampling_offsets.weight", name + ".sampling_offsets.bias"]
        return names + ["model.reference_points.weight", "model.reference_points.bias"]

    def to_hip(self, device="cuda:0", dtype=F32, seed=1234):
        """Move the model onto the GPU: flat parameter store + packed GEMM weights.  dtype: runtime.F32 / runtime.BF16, or
        runtime.F16 for inference (the kernels' backward entry points reject it; train() then fails at the first launch)."""
        c = ctx()
        c.init_device(device, dtype, seed)
        self.store = hnn.ParamStore(self, c.device, dtype, nograd_names=NOGRAD_PARAMS, fused_groups=self.fused_groups(),
                                    lr_mult_names=self.lr_mult_names(), lr_mult=0.1)
        hnn.bind_all(self, self.store)
        self.store.pack()
        self.late_grad_prefixes = LATE_GRAD_PREFIXES
        self.grad_segment_prefixes = GRAD_SEGMENT_PREFIXES
        return self

    def set_dropout(self, p):
        """Override every dropout probability (parity tests run with p = 0; the recipe's value is 0.1 everywhere)."""
        for m in self.modules():
            if isinstance(m, (TransformerEncoderLayer, TransformerDecoderLayer, FCNHead)):
                m.p = p
            if isinstance(m, MultiHeadAttention):
                m.dropout = p
        self.cls_p = p

    def sync_weights(self):
        """Re-pack the compute-dtype weight copies after parameters were modified from outside (state-dict load)."""
        self.store.dirty = True

    def load_state_dict(self, state_dict, strict=True):
        r = super().load_state_dict(state_dict, strict)
        if self.store is not None:
            self.store.dirty = True
        return r

    set_state_dict = load_state_dict       # paddle spelling used by the reference (checkpoint.py)

    def clear_gradients(self):
        self.store.zero_grad()

    # ---- forward --------------------------------------------------------------------------------
    def __call__(self, images):
        if self.store is None:
            raise RuntimeError("call model.to_hip(device, dtype) before the first forward")
        c = ctx()
        if self.store.dirty:
            self.store.pack()
        c.training = self.training
        if self.training:
            c.begin_step()
        else:
            c.end_step()
            if c.fold_eval_bn:
                self.store.fold_bn()
                c.fold_live = True
        tape = Tape() if self.training else None
        c.tape = tape
        try:
            out = self.forward(images)
            if tape is not None:
                # newest model op = first to run in backward (after the loss's own): transposed dgrad copies of the weights from
                # the forward operands the optimizer keeps current (ParamStore.pack, solver.Momentum.step)
                # (the engine's step makes them beside the forward on its prologue stream instead: Context.pack_bwd_done)
                if not c.pack_bwd_done:
                    tape.record(lambda: self.store.pack(bwd_only=True))
        finally:
            c.tape = None
            c.fold_live = False
        res = LogitsTuple(out)
        res.tape = tape
        return res

    def forward(self, inputs):  # :252-304
        c = ctx()
        assert inputs.dim() == 4 and inputs.shape[1] == 3 and inputs.shape[2] % 32 == 0 and inputs.shape[3] % 32 == 0, \
            "EMRT expects fp32 NCHW images whose H and W are multiples of 32 (paddle_EMRT.py:293)"
        x = Fn.nchw_to_nhwc(inputs.contiguous(), c_out=IMAGE_CHANNELS)
        B, H, W, _ = x.shape
        SH, SW = H // 8, W // 8                 # x_context.shape[2:] (:283-288); tiles need not be square
        psp_cat = c.empty((B, SH, SW, 256 * (2 + len(self.psp_scale))))
        if (self.training and c.side_branch and c.tape is not None and c.world_size == 1 and not c.sync_always and not c.segment_order and not c.overlap
                and isinstance(self.backbone, ResNet)):
            # one rank: the spatial branch's conv stages ride in the launches of the ResNet's layer3 / layer4 blocks (Fn.SideJobs); with more ranks the
            # reference's order is kept (the early gradient exchange counts on the order in which the segments' gradients become final)
            c.side = Fn.SideJobs(self.spatial_branch.jobs(x, out=Fn.narrow(psp_cat, 3, 0, 256)))
            try:
                c1, c2, c3, c4 = self.backbone(x)
                x_context = c.side.finish()
            finally:
                c.side = None
        else:
            c1, c2, c3, c4 = self.backbone(x)
            x_context = self.spatial_branch(x, out=Fn.narrow(psp_cat, 3, 0, 256))
        # data-parallel training: the five SyncBatchNorm layers (paddle_EMRT.py:64, fcn_head.py:53) all-reduce their statistics, and inside a
        # captured step every such collective is a cut between two hipGraphs.  The auxiliary head's conv -> SyncBN only needs c3, so it joins
        # the pyramid-pooling group here: ONE collective per direction for all five layers (2 cuts per step instead of 4).  With one rank
        # (no collective, nothing to merge) the reference's order is kept.
        aux_feat = None
        head0 = self.auxlayer.convs[0]
        if self.training and Fn._sync_active(head0[1].state):
            x_psp, aux_feat = self.psp_module(x_context, extra=(head0[0], head0[1], c3))
        else:
            x_psp = self.psp_module(x_context)
        hs, memory, shapes, spans = self.model([c2, c3, c4], x_psp)
        maps = [Fn.tokens_as_map(Fn.narrow(memory, 1, a, n), h, w) for (h, w), (a, n) in zip(shapes, spans)]
        nps = len(self.psp_scale)
        self.EFP(maps[0], maps[1], maps[2], out=Fn.narrow(psp_cat, 3, 256 * (1 + nps), 256))
        Fn.pyramid_tokens_to_maps(hs, self.psp_scale, SH, SW, [Fn.narrow(psp_cat, 3, 256 * (1 + i), 256) for i in range(nps)])  # :281-291
        o = Fn.conv_bn(self.cls_psp[0], self.cls_psp[1], psp_cat, relu=True, defer="conv")
        # One rank: cls_psp's second conv -> BatchNorm -> ReLU (32^2 x 512 -> 256, paddle_EMRT.py:201-209) and the auxiliary head's (16^2 x 1024 -> 256 on c3,
        # fcn_head.py:47-56) are independent 3x3 stages of 512 + 128 tiles: side by side in ONE grouped launch per pass (Fn.conv_bn_small_group) instead
        # of two latency-bound chains of conv | BatchNorm apply and col_reduce | BatchNorm backward | data gradient
        if aux_feat is None and self.training and c.head_pair and B * SH * SW <= 16384:
            om = o.materialize() if isinstance(o, Fn.PendingBN) else o
            pair_convs, pair_bns = [self.cls_psp[3], head0[0]], [self.cls_psp[4], head0[1]]
            if Fn._small_group_ok(pair_convs, pair_bns, [om, c3]):
                o, aux_feat = Fn.conv_bn_small_group(pair_convs, pair_bns, [om, c3], relu=True)
            else:
                o = Fn.conv_bn(self.cls_psp[3], self.cls_psp[4], om, relu=True)
        else:
            o = Fn.conv_bn(self.cls_psp[3], self.cls_psp[4], o, relu=True)
        o = Fn.dropout(o, self.cls_p, self.cls_salt, mode=1, hw=SH * SW, sole_consumer_is_linear=True)   # UpHead's conv_0 is its only consumer
        logits = self.uphead(o)
        if self.training or self.compute_aux_in_eval:
            aux = self.auxlayer(c3, features=aux_feat)      # x16 bilinear; the reference's final align_corners=True resize is the identity here
            assert aux.shape[2] == H and aux.shape[3] == W
        else:
            aux = None
        return (logits, aux)
