"""Torch-CPU fp32 restatement of the reference EMRT model (ORACLE -- test infrastructure only).

PARITY UNPINNED (see oracle/__init__.py): the reference is PaddlePaddle code that cannot
run here; this file re-expresses it op for op with torch.nn.functional in the reference's own
NCHW / [B, L, C] layouts.  It is deliberately naive (same flatten/transpose/concat traffic as
the reference) so that it can be audited against the reference line by line.

Reference files (all under /root/reference/semantic_segmentation/):
  src/models/paddle_EMRT.py                                  -> Conv2dBlock, EFP, PyramidPoolingModule,
                                                                 branch_block, spatial_branch, UpHead, EMRT
  src/models/EMRT_utils/transformer_encoder_decoder.py        -> MSDeformableAttention, Transformer*Layer,
                                                                 TransformerEncoder/Decoder, EncoderDecoder
  src/models/EMRT_utils/utils.py:64-97                        -> deformable_attention_core_func
  src/models/EMRT_utils/layers.py:144-311                     -> MultiHeadAttention
  src/models/EMRT_utils/position_encoding.py:59-75            -> PositionEmbedding (sine)
  src/models/backbones/paddle_vision_resnet.py:43-257         -> BasicBlock, BottleneckBlock, ResNet
  src/models/decoders/fcn_head.py:19-81                       -> FCNHead

Module / parameter names reproduce the reference's state-dict keys.
Conventions that differ from Paddle and must be handled by a .pdparams importer:
  * nn.Linear.weight is [out, in] here (Paddle: [in, out]); MHA in_proj_weight is [3E, E] (Paddle [E, 3E]).
  * BatchNorm buffers keep Paddle's names `_mean`, `_variance`.
"""
import copy
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# Paddle-semantics BatchNorm (Appendix B#3): eps 1e-5, momentum 0.9 meaning
# running = 0.9*running + 0.1*batch, running variance updated with the BIASED batch variance.
# nn.SyncBatchNorm on one rank is the same computation (paddle_EMRT.py:64, fcn_head.py:53).
# --------------------------------------------------------------------------------------
class BatchNorm2D(nn.Module):
    def __init__(self, num_features, momentum=0.9, epsilon=1e-5):
        super().__init__()
        self.num_features = num_features
        self.momentum = momentum
        self.epsilon = epsilon
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("_mean", torch.zeros(num_features))
        self.register_buffer("_variance", torch.ones(num_features))

    def forward(self, x):
        if self.training:
            mean = x.mean(dim=(0, 2, 3))
            var = x.var(dim=(0, 2, 3), unbiased=False)
            with torch.no_grad():
                self._mean.mul_(self.momentum).add_(mean.detach(), alpha=1.0 - self.momentum)
                self._variance.mul_(self.momentum).add_(var.detach(), alpha=1.0 - self.momentum)
        else:
            mean, var = self._mean, self._variance
        xh = (x - mean[None, :, None, None]) * torch.rsqrt(var[None, :, None, None] + self.epsilon)
        return xh * self.weight[None, :, None, None] + self.bias[None, :, None, None]


SyncBatchNorm = BatchNorm2D  # single-rank semantics; DP statistics sync is tested separately


def _conv(cin, cout, k, stride=1, padding=0, bias=True):
    return nn.Conv2d(cin, cout, k, stride=stride, padding=padding, bias=bias)


@torch.no_grad()
def _paddle_conv_default_(conv):
    """Paddle nn.Conv2D default initialiser: Normal(0, sqrt(2 / (kh*kw*Cin))), bias 0 (Appendix D; believed,
    unverifiable here).  Used where the reference never re-initialises a conv."""
    fan = conv.weight.shape[1] * conv.weight.shape[2] * conv.weight.shape[3]
    conv.weight.normal_(0.0, math.sqrt(2.0 / fan))
    if conv.bias is not None:
        conv.bias.zero_()


# --------------------------------------------------------------------------------------
# ResNet  (paddle_vision_resnet.py:43-257)
# --------------------------------------------------------------------------------------
class BasicBlock(nn.Module):  # :43-88
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = BatchNorm2D(planes)
        self.relu = nn.ReLU()
        self.conv2 = _conv(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = BatchNorm2D(planes)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class BottleneckBlock(nn.Module):  # :91-149 (stride sits on conv2)
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        width = planes
        self.conv1 = _conv(inplanes, width, 1, bias=False)
        self.bn1 = BatchNorm2D(width)
        self.conv2 = _conv(width, width, 3, stride, 1, bias=False)
        self.bn2 = BatchNorm2D(width)
        self.conv3 = _conv(width, planes * 4, 1, bias=False)
        self.bn3 = BatchNorm2D(planes * 4)
        self.relu = nn.ReLU()
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class ResNet(nn.Module):  # :152-257
    layer_cfg = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}

    def __init__(self, depth, num_classes=1000):
        super().__init__()
        block = BasicBlock if depth in (18, 34) else BottleneckBlock
        layers = self.layer_cfg[depth]
        self.inplanes = 64
        self.conv1 = _conv(3, 64, 7, 2, 3, bias=False)
        self.bn1 = BatchNorm2D(64)
        self.relu = nn.ReLU()
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        # `fc` exists in the reference state dict (:213-214) but is never used by forward (:246-257)
        self.fc = nn.Linear(512 * block.expansion, num_classes)

    def _make_layer(self, block, planes, blocks, stride=1):  # :216-244
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                _conv(self.inplanes, planes * block.expansion, 1, stride, 0, bias=False),
                BatchNorm2D(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):  # :246-257
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        c1 = self.layer1(x)
        c2 = self.layer2(c1)
        c3 = self.layer3(c2)
        c4 = self.layer4(c3)
        return c1, c2, c3, c4


# --------------------------------------------------------------------------------------
# FCNHead (fcn_head.py:19-81)
# --------------------------------------------------------------------------------------
class FCNHead(nn.Module):
    def __init__(self, in_channels, channels, num_classes, dropout_ratio=0.1, up_ratio=16):
        super().__init__()
        self.up_ratio = up_ratio
        self.convs = nn.Sequential(nn.Sequential(
            _conv(in_channels, channels, 3, 1, 1, bias=False), SyncBatchNorm(channels), nn.ReLU()))
        self.dropout = nn.Dropout2d(p=dropout_ratio)
        self.conv_seg = _conv(channels, num_classes, 1)

    def forward(self, x):  # :72-81
        up = [self.up_ratio * s for s in x.shape[2:]]
        out = self.convs(x)
        out = self.dropout(out)
        out = self.conv_seg(out)
        return F.interpolate(out, up, mode="bilinear", align_corners=False)


# --------------------------------------------------------------------------------------
# Deformable attention (utils.py:64-97, transformer_encoder_decoder.py:21-107)
# --------------------------------------------------------------------------------------
def deformable_attention_core_func(value, value_spatial_shapes, sampling_locations, attention_weights):
    """utils.py:64-97 re-expressed with torch.  value [B,Lv,M,D]; shapes list of (h,w);
    sampling_locations [B,Lq,M,L,P,2] (x,y in [0,1]); attention_weights [B,Lq,M,L,P] -> [B,Lq,M*D]."""
    bs, _, n_head, c = value.shape
    _, Len_q, _, n_levels, n_points, _ = sampling_locations.shape
    sizes = [int(h) * int(w) for h, w in value_spatial_shapes]
    value_list = value.split(sizes, dim=1)                                            # :77
    sampling_grids = 2 * sampling_locations - 1                                       # :79
    sampling_value_list = []
    for level, (h, w) in enumerate(value_spatial_shapes):                             # :82
        value_l_ = value_list[level].flatten(2).transpose(1, 2).reshape(bs * n_head, c, int(h), int(w))
        sampling_grid_l_ = sampling_grids[:, :, :, level].permute(0, 2, 1, 3, 4).flatten(0, 1)
        sampling_value_l_ = F.grid_sample(value_l_, sampling_grid_l_, mode="bilinear",
                                          padding_mode="zeros", align_corners=False)   # :87-88
        sampling_value_list.append(sampling_value_l_)
    attention_weights = attention_weights.permute(0, 2, 1, 3, 4).reshape(
        bs * n_head, 1, Len_q, n_levels * n_points)                                   # :91-92
    output = (torch.stack(sampling_value_list, dim=-2).flatten(-2) * attention_weights).sum(-1)
    output = output.reshape(bs, n_head * c, Len_q)                                    # :94-95
    return output.transpose(1, 2)                                                     # :97


class MSDeformableAttention(nn.Module):  # t_e_d.py:21-107
    def __init__(self, embed_dim=256, num_heads=8, num_levels=4, num_points=4):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.num_levels, self.num_points = num_levels, num_points
        self.total_points = num_heads * num_levels * num_points
        self.head_dim = embed_dim // num_heads
        self.sampling_offsets = nn.Linear(embed_dim, self.total_points * 2)   # lr_mult 0.1 (:36-38)
        self.attention_weights = nn.Linear(embed_dim, self.total_points)
        self.value_proj = nn.Linear(embed_dim, embed_dim)
        self.output_proj = nn.Linear(embed_dim, embed_dim)
        self._reset_parameters()

    @torch.no_grad()
    def _reset_parameters(self):  # :46-63
        nn.init.zeros_(self.sampling_offsets.weight)
        thetas = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        grid_init = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid_init = grid_init / grid_init.abs().max(-1, keepdim=True)[0]
        grid_init = grid_init.reshape(self.num_heads, 1, 1, 2).repeat(1, self.num_levels, self.num_points, 1)
        scaling = torch.arange(1, self.num_points + 1, dtype=torch.float32).reshape(1, 1, -1, 1)
        grid_init = grid_init * scaling
        self.sampling_offsets.bias.copy_(grid_init.flatten())
        nn.init.zeros_(self.attention_weights.weight)
        nn.init.zeros_(self.attention_weights.bias)
        nn.init.xavier_uniform_(self.value_proj.weight)
        nn.init.zeros_(self.value_proj.bias)
        nn.init.xavier_uniform_(self.output_proj.weight)
        nn.init.zeros_(self.output_proj.bias)

    def forward(self, query, reference_points, value, value_spatial_shapes, value_mask=None):  # :65-107
        bs, Len_q = query.shape[:2]
        Len_v = value.shape[1]
        assert sum(int(h) * int(w) for h, w in value_spatial_shapes) == Len_v                 # :81
        value = self.value_proj(value)
        if value_mask is not None:
            value = value * value_mask.to(value.dtype).unsqueeze(-1)
        value = value.reshape(bs, Len_v, self.num_heads, self.head_dim)
        sampling_offsets = self.sampling_offsets(query).reshape(
            bs, Len_q, self.num_heads, self.num_levels, self.num_points, 2)
        attention_weights = self.attention_weights(query).reshape(
            bs, Len_q, self.num_heads, self.num_levels * self.num_points)
        attention_weights = F.softmax(attention_weights, -1).reshape(
            bs, Len_q, self.num_heads, self.num_levels, self.num_points)
        shapes_t = torch.tensor([[int(h), int(w)] for h, w in value_spatial_shapes], dtype=query.dtype)
        offset_normalizer = shapes_t.flip([1]).reshape(1, 1, 1, self.num_levels, 1, 2)        # (W_l, H_l)
        sampling_locations = reference_points.reshape(bs, Len_q, 1, self.num_levels, 1, 2) \
            + sampling_offsets / offset_normalizer
        output = deformable_attention_core_func(value, value_spatial_shapes, sampling_locations, attention_weights)
        return self.output_proj(output)


# --------------------------------------------------------------------------------------
# resnet50c: deep-stem, optionally dilated ResNet-50 (backbones/resnet.py:61-99 BottleneckV1b, :102-221 ResNetV1,
# :224-234 resnet50c; selected by MODEL.ENCODER.TYPE "resnet50c", paddle_EMRT.py:227-228)
# --------------------------------------------------------------------------------------
class BottleneckV1b(nn.Module):  # resnet.py:61-99 (stride AND dilation on conv2, padding = dilation)
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 1, bias=False)
        self.bn1 = BatchNorm2D(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, dilation, dilation, bias=False)
        self.bn2 = BatchNorm2D(planes)
        self.conv3 = _conv(planes, planes * 4, 1, bias=False)
        self.bn3 = BatchNorm2D(planes * 4)
        self.relu = nn.ReLU()
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class ResNetV1c(nn.Module):  # resnet.py:102-221 with deep_stem=True, multi_grid=False (every EMRT yaml), scale 1.0
    def __init__(self, layers=(3, 4, 6, 3), output_stride=32, num_classes=1000):
        super().__init__()
        dilations, strides = {32: ([1, 1], [2, 2]), 16: ([1, 2], [2, 1]), 8: ([2, 4], [1, 1])}[output_stride]     # :109-120
        self.inplanes = 128
        self.conv1 = nn.Sequential(_conv(3, 64, 3, 2, 1, bias=False), BatchNorm2D(64), nn.ReLU(),                  # :124-134
                                   _conv(64, 64, 3, 1, 1, bias=False), BatchNorm2D(64), nn.ReLU(),
                                   _conv(64, 128, 3, 1, 1, bias=False))
        self.bn1 = BatchNorm2D(128)
        self.relu = nn.ReLU()
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], stride=2)
        self.layer3 = self._make_layer(256, layers[2], stride=strides[0], dilation=dilations[0])
        self.layer4 = self._make_layer(512, layers[3], stride=strides[1], dilation=dilations[1])
        self.fc = nn.Linear(2048, num_classes)
        with torch.no_grad():                                                                                       # :151-160
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    nn.init.kaiming_normal_(m.weight, a=0, mode="fan_in", nonlinearity="relu")

    def _make_layer(self, planes, blocks, stride=1, dilation=1):  # :175-207
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(_conv(self.inplanes, planes * 4, 1, stride, 0, bias=False), BatchNorm2D(planes * 4))
        if dilation in (1, 2):
            first = 1
        elif dilation == 4:
            first = 2
        else:
            raise RuntimeError("=> unknown dilation size: {}".format(dilation))
        layers = [BottleneckV1b(self.inplanes, planes, stride, dilation=first, downsample=downsample)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(BottleneckV1b(self.inplanes, planes, dilation=dilation))
        return nn.Sequential(*layers)

    def forward(self, x):  # :209-221
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        c1 = self.layer1(x)
        c2 = self.layer2(c1)
        c3 = self.layer3(c2)
        c4 = self.layer4(c3)
        return c1, c2, c3, c4


# --------------------------------------------------------------------------------------
# MultiHeadAttention (layers.py:144-311) -- packed in-proj, softmax(QK^T/sqrt(d)), dropout on weights
# --------------------------------------------------------------------------------------
class MultiHeadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))   # torch [out,in]
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        self._reset_parameters()

    @torch.no_grad()
    def _reset_parameters(self):  # layers.py:214-219 ; xavier over the packed [E,3E] matrix
        E = self.embed_dim
        bound = math.sqrt(6.0 / (E + 3 * E))
        self.in_proj_weight.uniform_(-bound, bound)
        nn.init.zeros_(self.in_proj_bias)
        nn.init.xavier_uniform_(self.out_proj.weight)
        nn.init.zeros_(self.out_proj.bias)

    def compute_qkv(self, tensor, index):  # :221-234
        E = self.embed_dim
        t = F.linear(tensor, self.in_proj_weight[index * E:(index + 1) * E], self.in_proj_bias[index * E:(index + 1) * E])
        B, L, _ = t.shape
        return t.reshape(B, L, self.num_heads, self.head_dim).transpose(1, 2)

    def forward(self, query, key=None, value=None):  # :236-311
        key = query if key is None else key
        value = query if value is None else value
        q, k, v = (self.compute_qkv(t, i) for i, t in enumerate([query, key, value]))
        product = torch.matmul(q, k.transpose(-1, -2)) * (float(self.head_dim) ** -0.5)
        weights = F.softmax(product, dim=-1)
        if self.dropout:
            weights = F.dropout(weights, self.dropout, training=self.training)
        out = torch.matmul(weights, v)
        out = out.transpose(1, 2)
        out = out.reshape(out.shape[0], out.shape[1], -1)
        return self.out_proj(out)


# --------------------------------------------------------------------------------------
# Sine position embedding (position_encoding.py:59-75) with offset=-0.5, normalize=True
# --------------------------------------------------------------------------------------
def sine_position_embedding(mask, num_pos_feats=128, temperature=10000, offset=-0.5, eps=1e-6, scale=2 * math.pi):
    """mask [B,H,W] bool -> pos [B, 2*num_pos_feats, H, W]."""
    mask = mask.to(torch.float32)
    y_embed = mask.cumsum(1, dtype=torch.float32)
    x_embed = mask.cumsum(2, dtype=torch.float32)
    y_embed = (y_embed + offset) / (y_embed[:, -1:, :] + eps) * scale
    x_embed = (x_embed + offset) / (x_embed[:, :, -1:] + eps) * scale
    dim_t = 2 * (torch.arange(num_pos_feats) // 2).to(torch.float32)
    dim_t = temperature ** (dim_t / num_pos_feats)
    pos_x = x_embed.unsqueeze(-1) / dim_t
    pos_y = y_embed.unsqueeze(-1) / dim_t
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


# --------------------------------------------------------------------------------------
# Transformer encoder / decoder (t_e_d.py:109-473)
# --------------------------------------------------------------------------------------
def _get_clones(module, N):  # utils.py:31-32
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


def _linear_init_(m):  # initializer.py:267-270 with weight [in,out] => bound 1/sqrt(in_features)
    bound = 1 / math.sqrt(m.weight.shape[1])
    nn.init.uniform_(m.weight, -bound, bound)
    nn.init.uniform_(m.bias, -bound, bound)


class TransformerEncoderLayer(nn.Module):  # :109-204
    def __init__(self, d_model=256, n_head=8, dim_feedforward=1024, dropout=0.1, n_levels=3, n_points=6):
        super().__init__()
        self.self_attn = MSDeformableAttention(d_model, n_head, n_levels, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.conv0 = nn.Sequential(_conv(d_model, d_model, 3, 1, 1, bias=False), nn.GroupNorm(32, d_model), nn.GELU())
        self.conv1 = nn.Sequential(_conv(d_model, d_model, 3, 1, 1, bias=False), nn.GroupNorm(32, d_model), nn.GELU())
        self.conv2 = nn.Sequential(_conv(d_model, d_model, 3, 1, 1, bias=False), nn.GroupNorm(32, d_model), nn.GELU())
        with torch.no_grad():  # :148-152
            for seq in (self.conv0, self.conv1, self.conv2):   # no reset in the reference: Paddle Conv2D default
                _paddle_conv_default_(seq[0])
            _linear_init_(self.linear1)
            _linear_init_(self.linear2)
            nn.init.xavier_uniform_(self.linear1.weight)
            nn.init.xavier_uniform_(self.linear2.weight)

    def forward_ffn(self, src):  # :157-161
        src2 = self.linear2(self.dropout2(F.relu(self.linear1(src))))
        src = src + self.dropout3(src2)
        return self.norm2(src)

    @staticmethod
    def seq2_2D(src, spatial_shapes):  # :163-182
        bs, _, c = src.shape
        outs, start = [], 0
        for (h, w) in spatial_shapes:
            n = int(h) * int(w)
            outs.append(src[:, start:start + n].transpose(1, 2).reshape(bs, c, int(h), int(w)))
            start += n
        return outs

    def forward(self, src, reference_points, spatial_shapes, src_mask=None, pos_embed=None):  # :184-204
        x0, x1, x2 = self.seq2_2D(src, spatial_shapes)
        src0 = self.conv0(x0) + x0
        src1 = self.conv1(x1) + x1
        src2 = self.conv2(x2) + x2
        src_flatten = torch.cat([s.flatten(2).transpose(1, 2) for s in (src0, src1, src2)], 1)
        q = src if pos_embed is None else src + pos_embed
        src2 = self.self_attn(q, reference_points, src, spatial_shapes, src_mask)
        src = self.norm1(src + self.dropout1(src2))
        src = self.forward_ffn(src)
        return src + src_flatten


class TransformerEncoder(nn.Module):  # :207-239
    def __init__(self, encoder_layer, num_layers):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios):  # :213-228
        valid_ratios = valid_ratios.unsqueeze(1)
        pts = []
        for i, (H, W) in enumerate(spatial_shapes):
            H, W = int(H), int(W)
            ref_y, ref_x = torch.meshgrid(torch.linspace(0.5, H - 0.5, H), torch.linspace(0.5, W - 0.5, W), indexing="ij")
            ref_y = ref_y.flatten().unsqueeze(0) / (valid_ratios[:, :, i, 1] * H)
            ref_x = ref_x.flatten().unsqueeze(0) / (valid_ratios[:, :, i, 0] * W)
            pts.append(torch.stack((ref_x, ref_y), dim=-1))
        reference_points = torch.cat(pts, 1).unsqueeze(2)
        return reference_points * valid_ratios

    def forward(self, src, spatial_shapes, src_mask=None, pos_embed=None, valid_ratios=None):  # :230-239
        if valid_ratios is None:
            valid_ratios = torch.ones(src.shape[0], len(spatial_shapes), 2)
        reference_points = self.get_reference_points(spatial_shapes, valid_ratios)
        output = src
        for layer in self.layers:
            output = layer(output, reference_points, spatial_shapes, src_mask, pos_embed)
        return output


class TransformerDecoderLayer(nn.Module):  # :242-295
    def __init__(self, d_model=256, n_head=8, dim_feedforward=1024, dropout=0.1, n_levels=3, n_points=6):
        super().__init__()
        self.self_attn = MultiHeadAttention(d_model, n_head, dropout=dropout)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.cross_attn = MSDeformableAttention(d_model, n_head, n_levels, n_points)
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)
        with torch.no_grad():  # :267-271
            _linear_init_(self.linear1)
            _linear_init_(self.linear2)
            nn.init.xavier_uniform_(self.linear1.weight)
            nn.init.xavier_uniform_(self.linear2.weight)

    def forward(self, tgt, reference_points, memory, memory_spatial_shapes, memory_mask=None, query_pos_embed=None):
        q = k = tgt if query_pos_embed is None else tgt + query_pos_embed               # :283
        tgt2 = self.self_attn(q, k, value=tgt)
        tgt = self.norm1(tgt + self.dropout1(tgt2))
        q2 = tgt if query_pos_embed is None else tgt + query_pos_embed
        tgt2 = self.cross_attn(q2, reference_points, memory, memory_spatial_shapes, memory_mask)
        tgt = self.norm2(tgt + self.dropout2(tgt2))
        tgt2 = self.linear2(self.dropout3(F.relu(self.linear1(tgt))))                   # :276-280
        return self.norm3(tgt + self.dropout4(tgt2))


class TransformerDecoder(nn.Module):  # :298-334
    def __init__(self, decoder_layer, num_layers):
        super().__init__()
        self.layers = _get_clones(decoder_layer, num_layers)

    def forward(self, tgt, memory, reference_points, memory_spatial_shapes, memory_mask=None, query_pos_embed=None):
        output = tgt
        for layer in self.layers:
            output = layer(output, reference_points, memory, memory_spatial_shapes, memory_mask, query_pos_embed)
        return output.unsqueeze(0)


class EncoderDecoder(nn.Module):  # :337-473
    def __init__(self, num_queries=110, backbone_num_channels=(512, 1024, 2048), num_feature_levels=3,
                 num_encoder_points=6, num_decoder_points=6, hidden_dim=256, nhead=8, num_encoder_layers=4,
                 num_decoder_layers=2, dim_feedforward=1024, dropout=0.1):
        super().__init__()
        self.hidden_dim, self.nhead, self.num_feature_levels = hidden_dim, nhead, num_feature_levels
        enc_layer = TransformerEncoderLayer(hidden_dim, nhead, dim_feedforward, dropout, num_feature_levels, num_encoder_points)
        self.encoder = TransformerEncoder(enc_layer, num_encoder_layers)
        dec_layer = TransformerDecoderLayer(hidden_dim, nhead, dim_feedforward, dropout, num_feature_levels, num_decoder_points)
        self.decoder = TransformerDecoder(dec_layer, num_decoder_layers)
        self.level_embed = nn.Embedding(num_feature_levels, hidden_dim)
        self.tgt_embed = nn.Embedding(num_queries, hidden_dim)           # created, never used (:368, :469)
        self.query_pos_embed = nn.Embedding(num_queries, hidden_dim)
        self.reference_points = nn.Linear(hidden_dim, 2)                 # lr_mult 0.1 (:371-372)
        self.input_proj = nn.ModuleList([
            nn.Sequential(_conv(c, hidden_dim, 1), nn.GroupNorm(32, hidden_dim)) for c in backbone_num_channels])
        with torch.no_grad():  # :394-402
            nn.init.normal_(self.level_embed.weight)
            nn.init.normal_(self.tgt_embed.weight)
            nn.init.normal_(self.query_pos_embed.weight)
            nn.init.xavier_uniform_(self.reference_points.weight)
            nn.init.zeros_(self.reference_points.bias)
            for l in self.input_proj:
                nn.init.xavier_uniform_(l[0].weight)
                nn.init.zeros_(l[0].bias)

    @staticmethod
    def _get_valid_ratio(mask):  # :408-414
        mask = mask.to(torch.float32)
        _, H, W = mask.shape
        valid_ratio_h = mask[:, :, 0].sum(1) / H
        valid_ratio_w = mask[:, 0, :].sum(1) / W
        return torch.stack([valid_ratio_w, valid_ratio_h], -1)

    def forward(self, src_feats, src_psp):  # :416-473 (src_mask is always None on the EMRT path)
        srcs = [self.input_proj[i](src_feats[i]) for i in range(len(src_feats))]
        src_flatten, mask_flatten, lvl_pos_embed_flatten, spatial_shapes, valid_ratios = [], [], [], [], []
        for level, src in enumerate(srcs):
            bs, c, h, w = src.shape
            spatial_shapes.append((h, w))
            src_flatten.append(src.flatten(2).transpose(1, 2))
            mask = torch.ones(bs, h, w, dtype=torch.bool)
            valid_ratios.append(self._get_valid_ratio(mask))
            pos_embed = sine_position_embedding(mask, self.hidden_dim // 2).flatten(2).transpose(1, 2)
            lvl_pos_embed_flatten.append(pos_embed + self.level_embed.weight[level].reshape(1, 1, -1))
            mask_flatten.append(mask.to(src.dtype).flatten(1))
        src_flatten = torch.cat(src_flatten, 1)
        mask_flatten = torch.cat(mask_flatten, 1)
        lvl_pos_embed_flatten = torch.cat(lvl_pos_embed_flatten, 1)
        valid_ratios = torch.stack(valid_ratios, 1)
        memory = self.encoder(src_flatten, spatial_shapes, mask_flatten, lvl_pos_embed_flatten, valid_ratios)
        bs = memory.shape[0]
        query_embed = self.query_pos_embed.weight.unsqueeze(0).repeat(bs, 1, 1)
        reference_points = torch.sigmoid(self.reference_points(query_embed))
        reference_points_input = reference_points.unsqueeze(2) * valid_ratios.unsqueeze(1)
        src_psp = src_psp.transpose(1, 2)
        hs = self.decoder(src_psp, memory, reference_points_input, spatial_shapes, mask_flatten, query_embed)
        return hs, memory


# --------------------------------------------------------------------------------------
# EMRT model parts (paddle_EMRT.py:13-304)
# --------------------------------------------------------------------------------------
class Conv2dBlock(nn.Module):  # :13-29
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = nn.Sequential(_conv(cin, cout, 3, 1, 1, bias=False), BatchNorm2D(cout), nn.ReLU())
        self.conv2 = nn.Sequential(_conv(cout, cout, 3, 1, 1, bias=False), BatchNorm2D(cout), nn.ReLU())

    def forward(self, x):
        return self.conv2(self.conv1(x)) + x


class EFP(nn.Module):  # :31-48
    def __init__(self, cin=256, cout=256):
        super().__init__()
        self.conv0, self.conv1, self.conv2 = Conv2dBlock(cin, cout), Conv2dBlock(cin, cout), Conv2dBlock(cin, cout)

    def forward(self, x0, x1, x2):
        x_out2 = F.interpolate(self.conv2(x2), size=x1.shape[2:], mode="bilinear", align_corners=True)
        x_out21 = self.conv1(x1) + x_out2
        x_out21 = F.interpolate(x_out21, size=x0.shape[2:], mode="bilinear", align_corners=True)
        return self.conv0(x0) + x_out21


class PyramidPoolingModule(nn.Module):  # :50-78
    def __init__(self, pool_scales, in_channels, channels):
        super().__init__()
        self.pool_branches = nn.ModuleList([
            nn.Sequential(nn.AdaptiveAvgPool2d(s), _conv(in_channels, channels, 1, bias=False),
                          SyncBatchNorm(channels), nn.ReLU()) for s in pool_scales])

    def forward(self, x):
        n, c = x.shape[:2]
        return torch.cat([b(x).reshape(n, c, -1) for b in self.pool_branches], dim=-1)


class branch_block(nn.Module):  # :80-97
    def __init__(self, cin, cout, downsample=True):
        super().__init__()
        self.downsample = downsample
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.encode = nn.Sequential(
            _conv(cin, cout, 3, 1, 1, bias=False), BatchNorm2D(cout), nn.ReLU(),
            _conv(cout, cout, 3, 1, 1, bias=False), BatchNorm2D(cout), nn.ReLU())

    def forward(self, x):
        if self.downsample:
            x = self.maxpool(x)
        return self.encode(x)


class spatial_branch(nn.Module):  # :99-113
    def __init__(self, in_channels=3):
        super().__init__()
        self.Enc0, self.Enc1, self.Enc2 = branch_block(in_channels, 64), branch_block(64, 128), branch_block(128, 256)

    def forward(self, x):
        return self.Enc2(self.Enc1(self.Enc0(x)))


class UpHead(nn.Module):  # :115-181, num_conv == 3 branch (the one EMRT instantiates, :198-199)
    def __init__(self, embed_dim=256, num_classes=6, align_corners=False):
        super().__init__()
        self.align_corners = align_corners
        self.conv_0 = _conv(embed_dim, 256, 3, 1, 1)
        self.conv_1 = _conv(256, 256, 3, 1, 1)
        self.conv_2 = _conv(256, 256, 3, 1, 1)
        self.conv_3 = _conv(256, num_classes, 1)
        self.syncbn_fc_0, self.syncbn_fc_1, self.syncbn_fc_2 = BatchNorm2D(256), BatchNorm2D(256), BatchNorm2D(256)

    def _up2(self, x):
        return F.interpolate(x, [2 * s for s in x.shape[2:]], mode="bilinear", align_corners=self.align_corners)

    def forward(self, x):  # :164-180
        x = self._up2(F.relu(self.syncbn_fc_0(self.conv_0(x))))
        x = self._up2(F.relu(self.syncbn_fc_1(self.conv_1(x))))
        x = F.relu(self.syncbn_fc_2(self.conv_2(x)))
        return self._up2(self.conv_3(x))


class EMRT(nn.Module):  # :184-304
    """`backbone` in {"resnet18","resnet34","resnet50","resnet101"}.  resnet18/34 are build-side
    extensions with channels [128,256,512] (the reference hard-codes [512,1024,2048], :188-192)."""

    def __init__(self, num_classes=6, backbone="resnet50", output_stride=32):
        super().__init__()
        depth = 50 if backbone == "resnet50c" else int(backbone.replace("resnet", ""))
        self.nclass = num_classes
        self.backbone_num_channels = [128, 256, 512] if depth in (18, 34) else [512, 1024, 2048]
        self.hidden_dim = 256
        self.psp_scale = [1, 3, 6, 8]
        self.spatial_branch = spatial_branch(3)
        self.psp_module = PyramidPoolingModule(self.psp_scale, 256, 256)
        self.uphead = UpHead(256, num_classes)
        self.cls_psp = nn.Sequential(
            _conv(256 * (2 + len(self.psp_scale)), 512, 3, 1, 1, bias=False), BatchNorm2D(512), nn.ReLU(),
            _conv(512, 256, 3, 1, 1, bias=False), BatchNorm2D(256), nn.ReLU(), nn.Dropout2d(p=0.1))
        self.EFP = EFP(256, 256)
        self.auxlayer = FCNHead(self.backbone_num_channels[1], self.backbone_num_channels[1] // 4, num_classes)
        with torch.no_grad():  # :217-225 -- runs BEFORE the backbone / transformer exist
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    nn.init.kaiming_normal_(m.weight, a=0, mode="fan_in", nonlinearity="relu")
        if backbone == "resnet50c":     # :227-228 get_segmentation_backbone: deep stem, dilation by MODEL.OUTPUT_STRIDE
            self.backbone = ResNetV1c(output_stride=output_stride)
        else:
            self.backbone = ResNet(depth)   # reference downloads ImageNet weights (:231-232); offline => Paddle defaults
            with torch.no_grad():
                for m in self.backbone.modules():
                    if isinstance(m, nn.Conv2d):
                        _paddle_conv_default_(m)
        with torch.no_grad():
            nn.init.xavier_uniform_(self.backbone.fc.weight)
            nn.init.zeros_(self.backbone.fc.bias)
        self.model = EncoderDecoder(backbone_num_channels=self.backbone_num_channels, hidden_dim=256,
                                    dim_feedforward=1024, dropout=0.1, num_feature_levels=3, nhead=8,
                                    num_encoder_layers=4, num_decoder_layers=2, num_encoder_points=6,
                                    num_decoder_points=6)

    def forward(self, inputs):  # :252-304
        c1, c2, c3, c4 = self.backbone(inputs)
        x_fea = [c2, c3, c4]
        x_context = self.spatial_branch(inputs)
        x_psp = self.psp_module(x_context)
        x_trans, memory = self.model(x_fea, x_psp)
        x_trans = x_trans.squeeze(0).transpose(1, 2)
        maps, start = [], 0
        for f in x_fea:  # :268-277
            n = f.shape[-1] * f.shape[-2]
            maps.append(memory[:, start:start + n].transpose(1, 2).reshape(f.shape[0], 256, f.shape[-2], f.shape[-1]))
            start += n
        x_fpn = self.EFP(*maps)
        psp_idx, psp_cat = 0, x_context
        bs, ctx_c = x_context.shape[:2]
        for i in self.psp_scale:  # :281-291
            pooled = x_trans[:, :, psp_idx:psp_idx + i * i].reshape(bs, ctx_c, i, i)
            psp_cat = torch.cat([psp_cat, F.interpolate(pooled, size=x_context.shape[2:], mode="bilinear",
                                                        align_corners=True)], 1)
            psp_idx += i * i
        psp_cat = torch.cat([psp_cat, x_fpn], 1)
        x = self.uphead(self.cls_psp(psp_cat))
        auxout = self.auxlayer(c3)
        auxout = F.interpolate(auxout, inputs.shape[2:], mode="bilinear", align_corners=True)
        return (x, auxout)


LR_MULT_SUFFIXES = ("sampling_offsets.weight", "sampling_offsets.bias",
                    "model.reference_points.weight", "model.reference_points.bias")


def lr_mult_of(name):
    """ParamAttr(learning_rate=0.1) sites: t_e_d.py:36-38 (sampling_offsets), :371-372 (reference_points)."""
    return 0.1 if name.endswith(LR_MULT_SUFFIXES) else 1.0
