"""numpy float64 direct restatement of `deformable_attention_core_func` (ORACLE -- test infra only).

Reference: /root/reference/semantic_segmentation/src/models/EMRT_utils/utils.py:64-97.
The reference calls F.grid_sample(bilinear, zeros, align_corners=False) per level on
grids = 2*loc-1 (utils.py:79,87-88); with align_corners=False the un-normalised pixel
coordinate is x_pix = ((g+1)*W - 1)/2 = loc_x*W - 0.5 (same for y with H), the four
neighbouring pixels contribute with bilinear weights and out-of-range neighbours contribute 0.
Output channel index = head*D + d (utils.py:94-97).

PARITY UNPINNED: Paddle cannot run here; this file pins the torch restatement
(oracle/emrt_torch.py) and is itself cross-checked against transformers'
MultiScaleDeformableAttention in tests/test_oracle_msda.py.
"""
import numpy as np


def msda_core_f64(value, spatial_shapes, sampling_locations, attention_weights):
    """value [B,Lv,M,D]; spatial_shapes [(h,w)]*L; sampling_locations [B,Lq,M,L,P,2] (x,y);
    attention_weights [B,Lq,M,L,P]  ->  [B,Lq,M*D] float64 (vectorised over B,Lq,M,P)."""
    value = np.asarray(value, dtype=np.float64)
    loc = np.asarray(sampling_locations, dtype=np.float64)
    aw = np.asarray(attention_weights, dtype=np.float64)
    B, Lv, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = np.zeros((B, Lq, M, D), dtype=np.float64)
    start = 0
    bidx = np.arange(B)[:, None, None, None]
    midx = np.arange(M)[None, None, :, None]
    for l, (h, w) in enumerate(spatial_shapes):
        h, w = int(h), int(w)
        v = value[:, start:start + h * w].reshape(B, h, w, M, D)
        start += h * w
        x = loc[:, :, :, l, :, 0] * w - 0.5          # [B,Lq,M,P]
        y = loc[:, :, :, l, :, 1] * h - 0.5
        x0 = np.floor(x).astype(np.int64)
        y0 = np.floor(y).astype(np.int64)
        lx, ly = x - x0, y - y0
        acc = np.zeros((B, Lq, M, P, D), dtype=np.float64)
        for dy, wy in ((0, 1.0 - ly), (1, ly)):
            for dx, wx in ((0, 1.0 - lx), (1, lx)):
                yy, xx = y0 + dy, x0 + dx
                ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
                g = v[bidx, np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1), midx]    # [B,Lq,M,P,D]
                acc += g * (wy * wx * ok)[..., None]
        out += (acc * aw[:, :, :, l, :, None]).sum(3)
    return out.reshape(B, Lq, M * D)


def msda_core_loops(value, spatial_shapes, sampling_locations, attention_weights):
    """Pure-python scalar loops (small cases only) -- independent of the vectorised version above."""
    value = np.asarray(value, dtype=np.float64)
    B, Lv, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    starts = np.cumsum([0] + [int(h) * int(w) for h, w in spatial_shapes])
    out = np.zeros((B, Lq, M * D))
    for b in range(B):
        for q in range(Lq):
            for m in range(M):
                for l, (h, w) in enumerate(spatial_shapes):
                    h, w = int(h), int(w)
                    for p in range(P):
                        x = float(sampling_locations[b, q, m, l, p, 0]) * w - 0.5
                        y = float(sampling_locations[b, q, m, l, p, 1]) * h - 0.5
                        x0, y0 = int(np.floor(x)), int(np.floor(y))
                        a = float(attention_weights[b, q, m, l, p])
                        for yy, wy in ((y0, 1 - (y - y0)), (y0 + 1, y - y0)):
                            for xx, wx in ((x0, 1 - (x - x0)), (x0 + 1, x - x0)):
                                if 0 <= yy < h and 0 <= xx < w:
                                    out[b, q, m * D:(m + 1) * D] += a * wy * wx * value[b, starts[l] + yy * w + xx, m]
    return out
