// Spatial resampling kernels for gfx950 (NHWC, HBM-bound, 4-channel vectors where C % 4 == 0):
// bilinear resize (both align_corners conventions; fused "+ addend", strided concat-slice output, fp32 NCHW logits
// output), its gather-form backward, multi-scale adaptive average pooling, 3x3/s2 max pooling, NCHW->NHWC ingest.
//
// Reference call sites replaced: F.interpolate (paddle_EMRT.py:40,44,169,174,180,
// 288-289,301; fcn_head.py:80), nn.AdaptiveAvgPool2D (paddle_EMRT.py:62), nn.MaxPool2D(3,2,1)
// (paddle_vision_resnet.py:201, paddle_EMRT.py:84), paddle.concat (paddle_EMRT.py:290-293 -- replaced by writing
// each producer straight into its channel slice of the [B,S,S,1536] buffer).
#include "common.hpp"
#include "bn_operand.hpp"

using namespace emrt;

struct Axis {  // source coordinate of destination index d:  src = max(0?, s*d + t)
  float s, t;
  int clamp0;  // align_corners=False clamps negative src to 0
};

__host__ __device__ inline Axis make_axis(int in, int out, int align_corners) {
  Axis a;
  if (align_corners) { a.s = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; a.t = 0.f; a.clamp0 = 0; }
  else { a.s = (float)in / (float)out; a.t = 0.5f * a.s - 0.5f; a.clamp0 = 1; }
  return a;
}

__device__ __forceinline__ void axis_src(const Axis& a, int d, int in, int& i0, int& i1, float& l0, float& l1) {
  float src = a.clamp0 ? a.s * ((float)d + 0.5f) - 0.5f : a.s * (float)d;
  if (a.clamp0 && src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}

struct ResizeArgs {
  const void* in; long long in_bs; int in_ld; int IH, IW;
  void* out; long long out_bs; int out_ld; int OH, OW;
  const void* add; long long add_bs; int add_ld;
  int N, C;
  Axis ay, ax;
  int out_nchw_f32;   // out is float [N][C][OH][OW]
};

// VEC elements per access: 1 (scalar), 4 (8 B bf16 / 16 B f32) or 8 (16 B bf16 / 32 B f32)
template <class T, int VEC>
struct VecIO;
template <class T>
struct VecIO<T, 1> {
  __device__ static __forceinline__ void load(const T* p, float (&o)[1]) { o[0] = to_f32(p[0]); }
  __device__ static __forceinline__ void store(T* p, const float (&o)[1]) { p[0] = from_f32<T>(o[0]); }
};
template <class T>
struct VecIO<T, 4> : Vec4<T> {};
template <class T>
struct VecIO<T, 8> : Vec8<T> {};

template <class T, int VEC>
__global__ __launch_bounds__(256) void resize_fwd_kernel(ResizeArgs a) {
  const int cv = a.C / VEC;
  const long long total = (long long)a.N * a.OH * a.OW * cv;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    int c, ow, oh, n;
    unravel4(idx, cv, a.OW, a.OH, total <= 0xffffffffll, c, ow, oh, n);
    c *= VEC;
    int y0, y1, x0, x1;
    float wy0, wy1, wx0, wx1;
    axis_src(a.ay, oh, a.IH, y0, y1, wy0, wy1);
    axis_src(a.ax, ow, a.IW, x0, x1, wx0, wx1);
    const T* ip = (const T*)a.in + (long long)n * a.in_bs + c;
    float o[VEC];
    {
      float v00[VEC], v01[VEC], v10[VEC], v11[VEC];
      VecIO<T, VEC>::load(ip + ((long long)y0 * a.IW + x0) * a.in_ld, v00);
      VecIO<T, VEC>::load(ip + ((long long)y0 * a.IW + x1) * a.in_ld, v01);
      VecIO<T, VEC>::load(ip + ((long long)y1 * a.IW + x0) * a.in_ld, v10);
      VecIO<T, VEC>::load(ip + ((long long)y1 * a.IW + x1) * a.in_ld, v11);
#pragma unroll
      for (int e = 0; e < VEC; ++e) o[e] = wy0 * (wx0 * v00[e] + wx1 * v01[e]) + wy1 * (wx0 * v10[e] + wx1 * v11[e]);
    }
    if (a.add) {
      const T* ap = (const T*)a.add + (long long)n * a.add_bs + ((long long)oh * a.OW + ow) * a.add_ld + c;
      float q[VEC];
      VecIO<T, VEC>::load(ap, q);
#pragma unroll
      for (int e = 0; e < VEC; ++e) o[e] += q[e];
    }
    if (a.out_nchw_f32) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) ((float*)a.out)[(((long long)n * a.C + c + e) * a.OH + oh) * a.OW + ow] = o[e];
    } else {
      T* op = (T*)a.out + (long long)n * a.out_bs + ((long long)oh * a.OW + ow) * a.out_ld + c;
      VecIO<T, VEC>::store(op, o);
    }
  }
}

// out = resize(relu(BN(in))): `in` is the raw pre-BatchNorm map of the producing conv, normalised tap by tap as it is loaded
// (bn_operand.hpp).  The ReLU comes BEFORE the interpolation, so each of the four taps is transformed on its own.
template <class T, int VEC>
struct ResizeTaps {
  float v00[VEC], v01[VEC], v10[VEC], v11[VEC];
  float wy0, wy1, wx0, wx1;
  long long out_off;
};
template <class T, int VEC>
__device__ __forceinline__ void resize_taps_load(const ResizeArgs& a, long long idx, int cv, bool small, ResizeTaps<T, VEC>& t) {
  int c, ow, oh, n;
  unravel4(idx, cv, a.OW, a.OH, small, c, ow, oh, n);
  c *= VEC;
  int y0, y1, x0, x1;
  axis_src(a.ay, oh, a.IH, y0, y1, t.wy0, t.wy1);
  axis_src(a.ax, ow, a.IW, x0, x1, t.wx0, t.wx1);
  const T* ip = (const T*)a.in + (long long)n * a.in_bs + c;
  VecIO<T, VEC>::load(ip + ((long long)y0 * a.IW + x0) * a.in_ld, t.v00);
  VecIO<T, VEC>::load(ip + ((long long)y0 * a.IW + x1) * a.in_ld, t.v01);
  VecIO<T, VEC>::load(ip + ((long long)y1 * a.IW + x0) * a.in_ld, t.v10);
  VecIO<T, VEC>::load(ip + ((long long)y1 * a.IW + x1) * a.in_ld, t.v11);
  t.out_off = (long long)n * a.out_bs + ((long long)oh * a.OW + ow) * a.out_ld + c;
}
// host: 256 % (C / VEC) == 0, so a thread keeps ONE channel group over its grid-stride loop: its scale / shift live in registers.  The
// first element's four taps are requested before the per-channel preamble and every later element one iteration ahead (a block is a
// latency chain: gather -> blend -> store).
template <class T, int VEC>
__global__ __launch_bounds__(256) void resize_fwd_bn_kernel(ResizeArgs a, BnOperand bn) {
  extern __shared__ float bn_lds[];          // [2][C]: scale, shift
  const int cv = a.C / VEC;
  const long long total = (long long)a.N * a.OH * a.OW * cv;
  const bool small = total <= 0xffffffffll;
  const long long step = (long long)gridDim.x * blockDim.x;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  ResizeTaps<T, VEC> cur, nxt;
  if (idx < total) resize_taps_load<T, VEC>(a, idx, cv, small, cur);
  bn_operand_preamble(bn, a.C, bn_lds, blockIdx.x == 0);
  const float lo = bn.relu ? 0.f : -INFINITY;
  const int c = ((int)threadIdx.x % cv) * VEC;
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { sc[e] = bn_lds[c + e]; sh[e] = bn_lds[a.C + c + e]; }
  while (idx < total) {
    const long long nidx = idx + step;
    if (nidx < total) resize_taps_load<T, VEC>(a, nidx, cv, small, nxt);
    float o[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float t00 = fmaxf(fmaf(cur.v00[e], sc[e], sh[e]), lo), t01 = fmaxf(fmaf(cur.v01[e], sc[e], sh[e]), lo);
      const float t10 = fmaxf(fmaf(cur.v10[e], sc[e], sh[e]), lo), t11 = fmaxf(fmaf(cur.v11[e], sc[e], sh[e]), lo);
      o[e] = cur.wy0 * (cur.wx0 * t00 + cur.wx1 * t01) + cur.wy1 * (cur.wx0 * t10 + cur.wx1 * t11);
    }
    VecIO<T, VEC>::store((T*)a.out + cur.out_off, o);
    cur = nxt;
    idx = nidx;
  }
}

// NCHW-ordered variant for the final logits (C = num classes): thread index runs over ow fastest so the fp32 NCHW
// stores coalesce.
template <class T>
__global__ __launch_bounds__(256) void resize_fwd_nchw_kernel(ResizeArgs a) {
  const long long total = (long long)a.N * a.C * a.OH * a.OW;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    int ow, oh, c, n;
    unravel4(idx, a.OW, a.OH, a.C, total <= 0xffffffffll, ow, oh, c, n);
    int y0, y1, x0, x1;
    float wy0, wy1, wx0, wx1;
    axis_src(a.ay, oh, a.IH, y0, y1, wy0, wy1);
    axis_src(a.ax, ow, a.IW, x0, x1, wx0, wx1);
    const T* ip = (const T*)a.in + (long long)n * a.in_bs + c;
    const float v = wy0 * (wx0 * to_f32(ip[((long long)y0 * a.IW + x0) * a.in_ld]) + wx1 * to_f32(ip[((long long)y0 * a.IW + x1) * a.in_ld])) +
                    wy1 * (wx0 * to_f32(ip[((long long)y1 * a.IW + x0) * a.in_ld]) + wx1 * to_f32(ip[((long long)y1 * a.IW + x1) * a.in_ld]));
    ((float*)a.out)[idx] = v;
  }
}

// Backward in gather form (deterministic, no atomics): one thread per INPUT pixel (x VEC channels) visits the
// destination pixels whose 2x2 footprint can include it and re-derives their weights exactly as the forward does.
struct ResizeBwdArgs {
  const void* dout; long long do_bs; int do_ld; int OH, OW;   // NHWC T, or fp32 NCHW when dout_nchw_f32
  void* din; long long di_bs; int di_ld; int IH, IW;
  int N, C;
  Axis ay, ax;
  int dout_nchw_f32;
};

__device__ __forceinline__ void axis_range(const Axis& a, int i, int out, int& lo, int& hi) {
  if (a.s <= 0.f) { lo = 0; hi = out - 1; return; }
  const float t = a.clamp0 ? a.t : 0.f;
  // destination d touches source i only when s*d + t lies in (i-1, i+1) (or is clamped onto i at a border, which the
  // clamps below keep inside the window); inclusive floor/ceil bounds leave one index of slack for float rounding
  lo = (int)floorf(((float)i - 1.f - t) / a.s);
  hi = (int)ceilf(((float)i + 1.f - t) / a.s);
  if (lo < 0) lo = 0;
  if (hi > out - 1) hi = out - 1;
}

// weight with which destination index d reads source index i along one axis (0 when it does not)
__device__ __forceinline__ float axis_weight(const Axis& a, int d, int in, int i) {
  int i0, i1;
  float l0, l1;
  axis_src(a, d, in, i0, i1, l0, l1);
  return (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
}

constexpr int RB_WIN = 12;  // separable fast path: the destination window of a source pixel is at most 12 wide (up to x4 upsampling)

template <class T, int VEC>
__global__ __launch_bounds__(256) void resize_bwd_kernel(ResizeBwdArgs a) {
  const int cv = a.C / VEC;
  const long long total = (long long)a.N * a.IH * a.IW * cv;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    int c, iw, ih, n;
    unravel4(idx, cv, a.IW, a.IH, total <= 0xffffffffll, c, iw, ih, n);
    c *= VEC;
    int ylo, yhi, xlo, xhi;
    axis_range(a.ay, ih, a.OH, ylo, yhi);
    axis_range(a.ax, iw, a.OW, xlo, xhi);
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    const T* gimg = (const T*)a.dout + (long long)n * a.do_bs + c;
    if (xhi - xlo < RB_WIN) {
      // the x weights do not depend on the row: evaluate them once (registers), then stream the rows
      float wxa[RB_WIN];
#pragma unroll
      for (int j = 0; j < RB_WIN; ++j) wxa[j] = (xlo + j <= xhi) ? axis_weight(a.ax, xlo + j, a.IW, iw) : 0.f;
      for (int oh = ylo; oh <= yhi; ++oh) {
        const float wy = axis_weight(a.ay, oh, a.IH, ih);
        if (wy == 0.f) continue;
        const T* grow = gimg + ((long long)oh * a.OW + xlo) * a.do_ld;
#pragma unroll
        for (int j = 0; j < RB_WIN; ++j) {
          if (wxa[j] == 0.f) continue;
          const float wgt = wy * wxa[j];
          float g[VEC];
          VecIO<T, VEC>::load(grow + (long long)j * a.do_ld, g);
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[e] = fmaf(wgt, g[e], acc[e]);
        }
      }
    } else {
      for (int oh = ylo; oh <= yhi; ++oh) {
        const float wy = axis_weight(a.ay, oh, a.IH, ih);
        if (wy == 0.f) continue;
        for (int ow = xlo; ow <= xhi; ++ow) {
          const float wx = axis_weight(a.ax, ow, a.IW, iw);
          if (wx == 0.f) continue;
          const float wgt = wy * wx;
          const T* gp = gimg + ((long long)oh * a.OW + ow) * a.do_ld;
          float g[VEC];
          VecIO<T, VEC>::load(gp, g);
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[e] = fmaf(wgt, g[e], acc[e]);
        }
      }
    }
    T* dp = (T*)a.din + (long long)n * a.di_bs + ((long long)ih * a.IW + iw) * a.di_ld + c;
    VecIO<T, VEC>::store(dp, acc);
  }
}

// Large upsampling factors (the 1x1 / 3x3 / 6x6 / 8x8 pyramid maps blown up to 32x32): a source pixel collects from
// hundreds of destination pixels, so one BLOCK takes one source pixel: threads = channel quads x phases over its
// destination window, reduced through LDS.
constexpr int RBW_THREADS = 1024;
template <class T>
__device__ __forceinline__ void resize_bwd_wide_body(const ResizeBwdArgs& a, const int pixel_block, const int chunk, const int nchunks, float* red) {
  const int iw = pixel_block % a.IW, ih = (pixel_block / a.IW) % a.IH, n = pixel_block / (a.IW * a.IH);
  int ylo, yhi, xlo, xhi;
  axis_range(a.ay, ih, a.OH, ylo, yhi);
  axis_range(a.ax, iw, a.OW, xlo, xhi);
  const int bw = xhi - xlo + 1, npix = (yhi - ylo + 1) * bw;
  // chunk: channel chunk (a 1x1 or 3x3 source map has only 8 / 72 pixels in the batch: splitting the channels over
  // nchunks blocks gives each block fewer quads, hence more phases over the 1024-pixel window and more blocks in flight)
  const int cq = a.C / 4 / nchunks;         // host guarantees C % 4 == 0, nchunks | C/4 and cq <= RBW_THREADS
  const int phases = RBW_THREADS / cq;
  const int q = (int)threadIdx.x % cq, ph = (int)threadIdx.x / cq;
  const int qg = chunk * cq + q;         // global channel quad
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (ph < phases) {
    const T* gimg = (const T*)a.dout + (long long)n * a.do_bs + qg * 4;
    for (int pxi = ph; pxi < npix; pxi += phases) {
      const int oh = ylo + pxi / bw, ow = xlo + pxi % bw;
      const float wgt = axis_weight(a.ay, oh, a.IH, ih) * axis_weight(a.ax, ow, a.IW, iw);
      if (wgt == 0.f) continue;
      float g[4];
      Vec4<T>::load(gimg + ((long long)oh * a.OW + ow) * a.do_ld, g);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = fmaf(wgt, g[e], acc[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[threadIdx.x * 4 + e] = acc[e];
  __syncthreads();
  if (ph == 0) {
    for (int o = 1; o < phases; ++o)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += red[(threadIdx.x + o * cq) * 4 + e];
    Vec4<T>::store((T*)a.din + (long long)n * a.di_bs + ((long long)ih * a.IW + iw) * a.di_ld + qg * 4, acc);
  }
}

template <class T>
__global__ __launch_bounds__(RBW_THREADS) void resize_bwd_wide_kernel(ResizeBwdArgs a) {
  __shared__ float red[RBW_THREADS * 4];
  resize_bwd_wide_body<T>(a, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y, red);
}

// The pyramid maps (paddle_EMRT.py:281-291: 1x1 / 3x3 / 6x6 / 8x8 token maps blown up to the feature map): all scales in one launch, forward
// and backward -- each alone is 8 ... 512 blocks, at the floor of a launch.
constexpr int PYR_MAX = 4;
struct PyramidResize {
  ResizeArgs f[PYR_MAX];
  ResizeBwdArgs b[PYR_MAX];
  long long first[PYR_MAX + 1];      // forward: first thread item of a scale; backward: first block
  int chunks[PYR_MAX];               // backward: channel chunks per source pixel of a scale (the rule of the one-scale launch)
  int n;
};
template <class T, int VEC>
__global__ __launch_bounds__(256) void resize_fwd_pyramid_kernel(PyramidResize m) {
  const long long total = m.first[m.n];
  for (long long gidx = (long long)blockIdx.x * blockDim.x + threadIdx.x; gidx < total; gidx += (long long)gridDim.x * blockDim.x) {
    int s = 0;
    while (s + 1 < m.n && gidx >= m.first[s + 1]) ++s;
    const ResizeArgs& a = m.f[s];
    const long long idx = gidx - m.first[s];
    const int cv = a.C / VEC;
    int c, ow, oh, n;
    unravel4(idx, cv, a.OW, a.OH, true, c, ow, oh, n);
    c *= VEC;
    int y0, y1, x0, x1;
    float wy0, wy1, wx0, wx1;
    axis_src(a.ay, oh, a.IH, y0, y1, wy0, wy1);
    axis_src(a.ax, ow, a.IW, x0, x1, wx0, wx1);
    const T* ip = (const T*)a.in + (long long)n * a.in_bs + c;
    float v00[VEC], v01[VEC], v10[VEC], v11[VEC], o[VEC];
    VecIO<T, VEC>::load(ip + ((long long)y0 * a.IW + x0) * a.in_ld, v00);
    VecIO<T, VEC>::load(ip + ((long long)y0 * a.IW + x1) * a.in_ld, v01);
    VecIO<T, VEC>::load(ip + ((long long)y1 * a.IW + x0) * a.in_ld, v10);
    VecIO<T, VEC>::load(ip + ((long long)y1 * a.IW + x1) * a.in_ld, v11);
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = wy0 * (wx0 * v00[e] + wx1 * v01[e]) + wy1 * (wx0 * v10[e] + wx1 * v11[e]);
    VecIO<T, VEC>::store((T*)a.out + (long long)n * a.out_bs + ((long long)oh * a.OW + ow) * a.out_ld + c, o);
  }
}
template <class T>
__global__ __launch_bounds__(RBW_THREADS) void resize_bwd_pyramid_kernel(PyramidResize m) {
  __shared__ float red[RBW_THREADS * 4];
  int s = 0;
  while (s + 1 < m.n && (long long)blockIdx.x >= m.first[s + 1]) ++s;
  const int local = (int)((long long)blockIdx.x - m.first[s]), nch = m.chunks[s];
  resize_bwd_wide_body<T>(m.b[s], local / nch, local % nch, nch, red);
}

// fp32 NCHW gradient (the logits): separable two-pass gather.  Pass W folds the columns, tmp[n][c][oh][iw] =
// sum_ow wx(ow -> iw) dout[n][c][oh][ow] (threads run along iw, the reads sweep each dout row once); pass H folds the rows
// and writes the NHWC result.  dout is read once instead of once per source pixel of its window (16x16 for the x16 aux head).
__global__ __launch_bounds__(256) void resize_bwd_nchw_w_kernel(ResizeBwdArgs a, float* __restrict__ tmp) {
  const long long total = (long long)a.N * a.C * a.OH * a.IW;
  const float* __restrict__ g = (const float*)a.dout;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    int iw;
    long long row;                               // (n, c, oh)
    unravel2(idx, a.IW, total <= 0xffffffffll, iw, row);
    int xlo, xhi;
    axis_range(a.ax, iw, a.OW, xlo, xhi);
    const float* gr = g + row * a.OW;
    float acc = 0.f;
    for (int ow = xlo; ow <= xhi; ++ow) acc = fmaf(axis_weight(a.ax, ow, a.IW, iw), gr[ow], acc);
    tmp[idx] = acc;
  }
}

template <class T>
__global__ __launch_bounds__(256) void resize_bwd_nchw_h_kernel(ResizeBwdArgs a, const float* __restrict__ tmp) {
  const long long total = (long long)a.N * a.C * a.IH * a.IW;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    int iw, ih, c, n;
    unravel4(idx, a.IW, a.IH, a.C, total <= 0xffffffffll, iw, ih, c, n);
    int ylo, yhi;
    axis_range(a.ay, ih, a.OH, ylo, yhi);
    const float* tp = tmp + ((long long)n * a.C + c) * a.OH * a.IW + iw;
    float acc = 0.f;
    for (int oh = ylo; oh <= yhi; ++oh) acc = fmaf(axis_weight(a.ay, oh, a.IH, ih), tp[(long long)oh * a.IW], acc);
    ((T*)a.din)[(long long)n * a.di_bs + ((long long)ih * a.IW + iw) * a.di_ld + c] = from_f32<T>(acc);
  }
}

// ---- multi-scale adaptive average pooling: in [N][H][W][C] -> out [N][sum k_i^2][C] (token order: scale, row, col) ----
struct PoolArgs {
  const void* in; long long in_bs; int in_ld; int H, W;
  void* out; long long out_bs; int out_ld;
  int N, C, nscales;
  int k[4], tok0[4];
  int ntok;
  int parts[4], item0[4], nitem;      // split pooling: blocks per bin of scale s, first work item of scale s, work items per image
};

__device__ __forceinline__ void bin_of(int i, int k, int S, int& b0, int& b1) {
  b0 = (i * S) / k;
  b1 = ((i + 1) * S + k - 1) / k;
}

// One block per (image, token): the threads cover the channels (4 at a time) x row phases of the bin, so that the
// 32x32-pixel bin of the 1x1 scale is not one thread's 1024-step serial loop; the phases are reduced through LDS.
constexpr int POOL_THREADS = 1024;

template <class T>
__global__ __launch_bounds__(POOL_THREADS) void adaptive_pool_fwd_kernel(PoolArgs a) {
  __shared__ float red[POOL_THREADS * 4];
  const int tok = blockIdx.x % a.ntok, n = blockIdx.x / a.ntok;
  int s = 0;
  while (s + 1 < a.nscales && tok >= a.tok0[s + 1]) ++s;
  const int k = a.k[s], t = tok - a.tok0[s];
  const int oi = t / k, oj = t - oi * k;
  int h0, h1, w0, w1;
  bin_of(oi, k, a.H, h0, h1);
  bin_of(oj, k, a.W, w0, w1);
  const int bw = w1 - w0, npix = (h1 - h0) * bw;
  const int cq = (a.C + 3) / 4;                       // channel quads
  int phases = POOL_THREADS / cq;
  if (phases < 1) phases = 1;
  const bool vec = (a.C % 4 == 0) && (a.in_ld % 4 == 0) && (a.in_bs % 4 == 0) && (((uintptr_t)a.in) % 16 == 0);
  const float inv = 1.f / (float)npix;
  for (int q0 = 0; q0 < cq; q0 += POOL_THREADS) {     // (only when C > 2048)
    const int q = q0 + (int)threadIdx.x % (cq < POOL_THREADS ? cq : POOL_THREADS);
    const int ph = (int)threadIdx.x / (cq < POOL_THREADS ? cq : POOL_THREADS);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (q < cq && ph < phases) {
      const T* ip = (const T*)a.in + (long long)n * a.in_bs + q * 4;
      if (vec) {
        // eight independent loads in flight per thread: the 32x32 bin of the 1x1 scale is 64 pixels per phase, and one load per
        // iteration made the block wait 64 L2 round trips (38 us for a 4 MB map)
        constexpr int U = 8;
        for (int px0 = ph; px0 < npix; px0 += phases * U) {
          float v[U][4];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int pxi = px0 + u * phases;
            const int pc = pxi < npix ? pxi : ph;
            const int hh = h0 + pc / bw, ww = w0 + pc % bw;
            Vec4<T>::load(ip + ((long long)hh * a.W + ww) * a.in_ld, v[u]);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const bool ok = px0 + u * phases < npix;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += ok ? v[u][e] : 0.f;
          }
        }
      } else {
        for (int pxi = ph; pxi < npix; pxi += phases) {
          const int hh = h0 + pxi / bw, ww = w0 + pxi % bw;
          const T* pp = ip + ((long long)hh * a.W + ww) * a.in_ld;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (q * 4 + e < a.C) acc[e] += to_f32(pp[e]);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[threadIdx.x * 4 + e] = acc[e];
    __syncthreads();
    if (q < cq && ph == 0) {
      const int stride_t = cq < POOL_THREADS ? cq : POOL_THREADS;
      for (int o = 1; o < phases; ++o)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += red[(threadIdx.x + o * stride_t) * 4 + e];
      T* op = (T*)a.out + (long long)n * a.out_bs + (long long)tok * a.out_ld + q * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (q * 4 + e < a.C) op[e] = from_f32<T>(acc[e] * inv);
    }
    __syncthreads();
  }
}

// backward: `in` = d(out) tokens, `out` = d(in) map (fully overwritten).  A pixel lies in at most two (overlapping)
// bins per axis and scale: the bin floor(h*k/H) and its predecessor.
template <class T>
__global__ __launch_bounds__(256) void adaptive_pool_bwd_kernel(PoolArgs a) {
  const long long total = (long long)a.N * a.H * a.W * a.C;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    int c, w, h, n;
    unravel4(idx, a.C, a.W, a.H, total <= 0xffffffffll, c, w, h, n);
    const T* gp = (const T*)a.out + (long long)n * a.out_bs + c;   // tokens gradient
    float acc = 0.f;
    for (int s = 0; s < a.nscales; ++s) {
      const int k = a.k[s];
      const int oi_hi = (h * k) / a.H, oj_hi = (w * k) / a.W;
      for (int oi = oi_hi > 0 ? oi_hi - 1 : 0; oi <= oi_hi + 1 && oi < k; ++oi) {
        int h0, h1;
        bin_of(oi, k, a.H, h0, h1);
        if (h < h0 || h >= h1) continue;
        for (int oj = oj_hi > 0 ? oj_hi - 1 : 0; oj <= oj_hi + 1 && oj < k; ++oj) {
          int w0, w1;
          bin_of(oj, k, a.W, w0, w1);
          if (w < w0 || w >= w1) continue;
          acc += to_f32(gp[(long long)(a.tok0[s] + oi * k + oj) * a.out_ld]) / (float)((h1 - h0) * (w1 - w0));
        }
      }
    }
    ((T*)const_cast<void*>(a.in))[(long long)n * a.in_bs + ((long long)h * a.W + w) * a.in_ld + c] = from_f32<T>(acc);
  }
}

// the same with 4 channels per thread: the bin geometry (integer divisions) is worked out once per pixel quad instead of
// once per element, token gradients come in 8/16-byte loads (the scalar version spent 65 us on a 4 MB map in divisions)
// The same gather with the bin membership of every row and column worked out ONCE per block: a pixel lies in at most two bins per axis and
// scale (neighbouring adaptive bins overlap by at most one pixel), and finding them costs ~14 integer divisions per scale -- the kernel above
// does that per thread (and tries 3 x 3 candidate bins), which is what its 25 us for a 4 MB map were.  Table: [scale][H + W] entries of
// (first bin, second bin or -1, 1 / extent of each) in LDS, built by the first (H + W) * nscales threads' worth of work; the pixel loop is then
// <= 2 x 2 loads per scale with one multiply each.  Host: H + W <= 512.
struct PoolAxisEntry { short i0, i1; float inv0, inv1; };
template <class T>
__global__ __launch_bounds__(256) void adaptive_pool_bwd_tab_kernel(PoolArgs a) {
  __shared__ PoolAxisEntry tab[4 * 512];
  const int HW_ = a.H + a.W;
  for (int e = threadIdx.x; e < HW_ * a.nscales; e += blockDim.x) {
    const int s = e / HW_, r = e - s * HW_;
    const bool is_row = r < a.H;
    const int x = is_row ? r : r - a.H, S = is_row ? a.H : a.W, k = a.k[s];
    const int hi = (x * k) / S;
    PoolAxisEntry en;
    en.i0 = en.i1 = -1;
    en.inv0 = en.inv1 = 0.f;
    for (int i = hi > 0 ? hi - 1 : 0; i <= hi + 1 && i < k; ++i) {
      int b0, b1;
      bin_of(i, k, S, b0, b1);
      if (x < b0 || x >= b1) continue;
      if (en.i0 < 0) { en.i0 = (short)i; en.inv0 = 1.f / (float)(b1 - b0); }
      else { en.i1 = (short)i; en.inv1 = 1.f / (float)(b1 - b0); }
    }
    tab[s * 512 + r] = en;
  }
  __syncthreads();
  const int cq = a.C / 4;
  const long long total = (long long)a.N * a.H * a.W * cq;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    int c, w, h, n;
    unravel4(idx, cq, a.W, a.H, total <= 0xffffffffll, c, w, h, n);
    c *= 4;
    const T* gp = (const T*)a.out + (long long)n * a.out_bs + c;   // tokens gradient
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < a.nscales; ++s) {
      const int k = a.k[s];
      const PoolAxisEntry er = tab[s * 512 + h], ec = tab[s * 512 + a.H + w];
#pragma unroll
      for (int ri = 0; ri < 2; ++ri) {
        const int oi = ri ? er.i1 : er.i0;
        if (oi < 0) continue;
        const float wr = ri ? er.inv1 : er.inv0;
#pragma unroll
        for (int ci = 0; ci < 2; ++ci) {
          const int oj = ci ? ec.i1 : ec.i0;
          if (oj < 0) continue;
          const float wt = wr * (ci ? ec.inv1 : ec.inv0);
          float g[4];
          Vec4<T>::load(gp + (long long)(a.tok0[s] + oi * k + oj) * a.out_ld, g);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = fmaf(g[e], wt, acc[e]);
        }
      }
    }
    Vec4<T>::store((T*)const_cast<void*>(a.in) + (long long)n * a.in_bs + ((long long)h * a.W + w) * a.in_ld + c, acc);
  }
}

template <class T>
__global__ __launch_bounds__(256) void adaptive_pool_bwd_vec_kernel(PoolArgs a) {
  const int cq = a.C / 4;
  const long long total = (long long)a.N * a.H * a.W * cq;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    int c, w, h, n;
    unravel4(idx, cq, a.W, a.H, total <= 0xffffffffll, c, w, h, n);
    c *= 4;
    const T* gp = (const T*)a.out + (long long)n * a.out_bs + c;   // tokens gradient
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < a.nscales; ++s) {
      const int k = a.k[s];
      const int oi_hi = (h * k) / a.H, oj_hi = (w * k) / a.W;
      for (int oi = oi_hi > 0 ? oi_hi - 1 : 0; oi <= oi_hi + 1 && oi < k; ++oi) {
        int h0, h1;
        bin_of(oi, k, a.H, h0, h1);
        if (h < h0 || h >= h1) continue;
        for (int oj = oj_hi > 0 ? oj_hi - 1 : 0; oj <= oj_hi + 1 && oj < k; ++oj) {
          int w0, w1;
          bin_of(oj, k, a.W, w0, w1);
          if (w < w0 || w >= w1) continue;
          float g[4];
          Vec4<T>::load(gp + (long long)(a.tok0[s] + oi * k + oj) * a.out_ld, g);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] += g[e] / (float)((h1 - h0) * (w1 - w0));
        }
      }
    }
    Vec4<T>::store((T*)const_cast<void*>(a.in) + (long long)n * a.in_bs + ((long long)h * a.W + w) * a.in_ld + c, acc);
  }
}

// ---- max pooling (k x k, stride, pad; -inf padding; first maximum wins, as torch/paddle) ----
// The forward stores, next to the maxima, the window slot (kh * k + kw, one byte) that won; the backward is then a
// gather over the <= ceil(k/stride)^2 windows that contain an input pixel instead of a re-run of every window's scan.
struct MaxPoolArgs {
  const void* in; void* out; const void* dout; void* din;
  unsigned char* arg;
  int N, H, W, C, OH, OW, k, stride, pad;
};

template <class T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(MaxPoolArgs a) {
  const long long total = (long long)a.N * a.OH * a.OW * a.C;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    int c, ow, oh, n;
    unravel4(idx, a.C, a.OW, a.OH, total <= 0xffffffffll, c, ow, oh, n);
    float best = -INFINITY;
    int slot = 255;
    for (int kh = 0; kh < a.k; ++kh) {
      const int h = oh * a.stride - a.pad + kh;
      if ((unsigned)h >= (unsigned)a.H) continue;
      for (int kw = 0; kw < a.k; ++kw) {
        const int w = ow * a.stride - a.pad + kw;
        if ((unsigned)w >= (unsigned)a.W) continue;
        const float v = to_f32(((const T*)a.in)[(((long long)n * a.H + h) * a.W + w) * a.C + c]);
        if (v > best || v != v || slot == 255) { best = v; slot = kh * a.k + kw; }
      }
    }
    ((T*)a.out)[idx] = from_f32<T>(best);
    if (a.arg) a.arg[idx] = (unsigned char)slot;
  }
}

// eight channels per thread: one 16 / 32-byte load per window tap, one 8-byte store of the argmax slots (the scalar kernel above
// issues nine 2-byte loads and three divisions per output element: 18 us for a 2 M-element map)
template <class T>
__global__ __launch_bounds__(256) void maxpool_fwd_vec8_kernel(MaxPoolArgs a) {
  const int cv = a.C / 8;
  const long long total = (long long)a.N * a.OH * a.OW * cv;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    int c, ow, oh, n;
    unravel4(idx, cv, a.OW, a.OH, total <= 0xffffffffll, c, ow, oh, n);
    c *= 8;
    float best[8];
    int slot[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; slot[e] = 255; }
    for (int kh = 0; kh < a.k; ++kh) {
      const int h = oh * a.stride - a.pad + kh;
      if ((unsigned)h >= (unsigned)a.H) continue;
      for (int kw = 0; kw < a.k; ++kw) {
        const int w = ow * a.stride - a.pad + kw;
        if ((unsigned)w >= (unsigned)a.W) continue;
        float v[8];
        Vec8<T>::load((const T*)a.in + (((long long)n * a.H + h) * a.W + w) * a.C + c, v);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (v[e] > best[e] || v[e] != v[e] || slot[e] == 255) { best[e] = v[e]; slot[e] = kh * a.k + kw; }
      }
    }
    const long long o = (((long long)n * a.OH + oh) * a.OW + ow) * a.C + c;
    Vec8<T>::store((T*)a.out + o, best);
    if (a.arg) {
      uint2 pk;
      pk.x = (unsigned)slot[0] | ((unsigned)slot[1] << 8) | ((unsigned)slot[2] << 16) | ((unsigned)slot[3] << 24);
      pk.y = (unsigned)slot[4] | ((unsigned)slot[5] << 8) | ((unsigned)slot[6] << 16) | ((unsigned)slot[7] << 24);
      *reinterpret_cast<uint2*>(a.arg + o) = pk;
    }
  }
}

// out = maxpool(relu(BN(in))) on the raw pre-BatchNorm map (bn_operand.hpp): every window tap is normalised as it is loaded (gamma may be
// negative, so the transform is not monotone and cannot be applied to the window's maximum instead).
template <class T>
__global__ __launch_bounds__(256) void maxpool_fwd_bn_kernel(MaxPoolArgs a, BnOperand bn) {
  extern __shared__ float bn_lds[];          // [2][C]: scale, shift
  bn_operand_preamble(bn, a.C, bn_lds, blockIdx.x == 0);
  const float lo = bn.relu ? 0.f : -INFINITY;
  const int cv = a.C / 8;
  const long long total = (long long)a.N * a.OH * a.OW * cv;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    int c, ow, oh, n;
    unravel4(idx, cv, a.OW, a.OH, total <= 0xffffffffll, c, ow, oh, n);
    c *= 8;
    float best[8], sc[8], sh[8];
    int slot[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; slot[e] = 255; sc[e] = bn_lds[c + e]; sh[e] = bn_lds[a.C + c + e]; }
    for (int kh = 0; kh < a.k; ++kh) {
      const int h = oh * a.stride - a.pad + kh;
      if ((unsigned)h >= (unsigned)a.H) continue;
      for (int kw = 0; kw < a.k; ++kw) {
        const int w = ow * a.stride - a.pad + kw;
        if ((unsigned)w >= (unsigned)a.W) continue;
        float v[8];
        Vec8<T>::load((const T*)a.in + (((long long)n * a.H + h) * a.W + w) * a.C + c, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float t = fmaxf(fmaf(v[e], sc[e], sh[e]), lo);
          if (t > best[e] || t != t || slot[e] == 255) { best[e] = t; slot[e] = kh * a.k + kw; }
        }
      }
    }
    const long long o = (((long long)n * a.OH + oh) * a.OW + ow) * a.C + c;
    Vec8<T>::store((T*)a.out + o, best);
    if (a.arg) {
      uint2 pk;
      pk.x = (unsigned)slot[0] | ((unsigned)slot[1] << 8) | ((unsigned)slot[2] << 16) | ((unsigned)slot[3] << 24);
      pk.y = (unsigned)slot[4] | ((unsigned)slot[5] << 8) | ((unsigned)slot[6] << 16) | ((unsigned)slot[7] << 24);
      *reinterpret_cast<uint2*>(a.arg + o) = pk;
    }
  }
}

// VEC = 4: four channels per thread (one 4-byte load of argmax slots, one 8/16-byte load of dout per window)
template <class T, int VEC>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(MaxPoolArgs a) {
  const int cv = a.C / VEC;
  const long long total = (long long)a.N * a.H * a.W * cv;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    int c, w, h, n;
    unravel4(idx, cv, a.W, a.H, total <= 0xffffffffll, c, w, h, n);
    c *= VEC;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    // output windows that contain (h, w): oh in [ceil((h+pad-k+1)/s), floor((h+pad)/s)]
    int oh_lo = (h + a.pad - a.k + 1 + a.stride - 1);
    oh_lo = oh_lo > 0 ? oh_lo / a.stride : 0;
    int oh_hi = (h + a.pad) / a.stride;
    if (oh_hi > a.OH - 1) oh_hi = a.OH - 1;
    int ow_lo = (w + a.pad - a.k + 1 + a.stride - 1);
    ow_lo = ow_lo > 0 ? ow_lo / a.stride : 0;
    int ow_hi = (w + a.pad) / a.stride;
    if (ow_hi > a.OW - 1) ow_hi = a.OW - 1;
    for (int oh = oh_lo; oh <= oh_hi; ++oh)
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        const long long o = (((long long)n * a.OH + oh) * a.OW + ow) * a.C + c;
        const int slot = (h - (oh * a.stride - a.pad)) * a.k + (w - (ow * a.stride - a.pad));
        if (VEC == 4) {
          const unsigned packed = *reinterpret_cast<const unsigned*>(a.arg + o);
          if (((packed & 0xffu) == (unsigned)slot) | (((packed >> 8) & 0xffu) == (unsigned)slot) | (((packed >> 16) & 0xffu) == (unsigned)slot) |
              ((packed >> 24) == (unsigned)slot)) {
            float g[4];
            Vec4<T>::load((const T*)a.dout + o, g);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (((packed >> (8 * e)) & 0xffu) == (unsigned)slot) acc[e] += g[e];
          }
        } else if ((int)a.arg[o] == slot) acc[0] += to_f32(((const T*)a.dout)[o]);
      }
    T* dp = (T*)a.din + (((long long)n * a.H + h) * a.W + w) * a.C + c;
    if (VEC == 4) {
      float w4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) w4[e] = acc[e < VEC ? e : 0];
      Vec4<T>::store(dp, w4);
    } else dp[0] = from_f32<T>(acc[0]);
  }
}

// fp32 NCHW images -> T NHWC
template <class T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, T* __restrict__ out, int N, int C, int H, int W, int CO) {
  // CO >= C output channels, the extra ones zero: the 3-channel image as an 8-channel map puts the first convolutions (and their
  // weight gradients) on the 16-byte vector path of the GEMM kernels
  const long long total = (long long)N * H * W * CO;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    int c, w, h, n;
    unravel4(idx, CO, W, H, total <= 0xffffffffll, c, w, h, n);
    out[idx] = from_f32<T>(c < C ? in[(((long long)n * C + c) * H + h) * W + w] : 0.f);
  }
}

// one thread per pixel: C coalesced plane reads, one 8 / 16 / 32-byte store of the CO = 4 or 8 output channels (the element-wise kernel
// above stores 2 bytes at a time: 24 us for a 16 x 3 x 256 x 256 batch)
template <class T, int CO>
__global__ __launch_bounds__(256) void nchw_to_nhwc_pix_kernel(const float* __restrict__ in, T* __restrict__ out, int N, int C, long long HW) {
  const long long total = (long long)N * HW;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    long long n, p;
    if (total <= 0xffffffffll) { const unsigned q = (unsigned)idx / (unsigned)HW; n = q; p = (unsigned)idx - q * (unsigned)HW; }      // (32-bit division: common.hpp, unravel)
    else { n = idx / HW; p = idx - n * HW; }
    float v[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) v[c] = c < C ? in[(n * C + c) * HW + p] : 0.f;
    if constexpr (CO == 8) Vec8<T>::store(out + idx * 8, v);
    else Vec4<T>::store(out + idx * 4, v);
  }
}

static inline int ew_grid(long long total) {
  long long g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

#define DT3(dtype, KERNEL, GRID, ...)                                                                 \
  do {                                                                                                \
    if ((dtype) == EMRT_F32) hipLaunchKernelGGL((KERNEL<float>), dim3(GRID), dim3(256), 0, st, __VA_ARGS__); \
    else if ((dtype) == EMRT_BF16) hipLaunchKernelGGL((KERNEL<bf16_t>), dim3(GRID), dim3(256), 0, st, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<f16_t>), dim3(GRID), dim3(256), 0, st, __VA_ARGS__);               \
  } while (0)
#define DT2(dtype, KERNEL, GRID, ...)                                                                 \
  do {                                                                                                \
    if ((dtype) == EMRT_F32) hipLaunchKernelGGL((KERNEL<float>), dim3(GRID), dim3(256), 0, st, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<bf16_t>), dim3(GRID), dim3(256), 0, st, __VA_ARGS__);              \
  } while (0)

extern "C" int emrt_resize_bilinear_fwd(const void* in, long long in_bs, int in_ld, int IH, int IW, void* out, long long out_bs,
                                        int out_ld, int OH, int OW, const void* add, long long add_bs, int add_ld, int N, int C,
                                        int align_corners, int out_nchw_f32, int dtype, void* stream) {
  EMRT_REQUIRE_FWD_DTYPE(dtype);
  EMRT_REQUIRE(in && out, "null pointer");
  EMRT_REQUIRE(N > 0 && C > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0, "bad dims");
  EMRT_REQUIRE(!(out_nchw_f32 && add), "addend not supported with NCHW output");
  ResizeArgs a;
  a.in = in; a.in_bs = in_bs; a.in_ld = in_ld; a.IH = IH; a.IW = IW;
  a.out = out; a.out_bs = out_bs; a.out_ld = out_ld; a.OH = OH; a.OW = OW;
  a.add = add; a.add_bs = add_bs; a.add_ld = add_ld; a.N = N; a.C = C;
  a.ay = make_axis(IH, OH, align_corners); a.ax = make_axis(IW, OW, align_corners); a.out_nchw_f32 = out_nchw_f32;
  hipStream_t st = (hipStream_t)stream;
  if (out_nchw_f32) {
    const int grid = ew_grid((long long)N * C * OH * OW);
    DT3(dtype, resize_fwd_nchw_kernel, grid, a);
    return check_launch("emrt_resize_bilinear_fwd");
  }
  const bool v4 = C % 4 == 0 && in_ld % 4 == 0 && out_ld % 4 == 0 && in_bs % 4 == 0 && out_bs % 4 == 0 &&
                  (!add || (add_ld % 4 == 0 && add_bs % 4 == 0)) && ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0) &&
                  (!add || (uintptr_t)add % 16 == 0);
  const bool v8 = v4 && dtype != EMRT_F32 && C % 8 == 0 && in_ld % 8 == 0 && out_ld % 8 == 0 && in_bs % 8 == 0 && out_bs % 8 == 0 &&
                  (!add || (add_ld % 8 == 0 && add_bs % 8 == 0));      // bf16: 16-byte accesses
  if (v8) {
    const int grid = ew_grid((long long)N * OH * OW * (C / 8));
    if (dtype == EMRT_BF16) hipLaunchKernelGGL((resize_fwd_kernel<bf16_t, 8>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((resize_fwd_kernel<f16_t, 8>), dim3(grid), dim3(256), 0, st, a);
  } else if (v4) {
    const int grid = ew_grid((long long)N * OH * OW * (C / 4));
    if (dtype == EMRT_F32) hipLaunchKernelGGL((resize_fwd_kernel<float, 4>), dim3(grid), dim3(256), 0, st, a);
    else if (dtype == EMRT_BF16) hipLaunchKernelGGL((resize_fwd_kernel<bf16_t, 4>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((resize_fwd_kernel<f16_t, 4>), dim3(grid), dim3(256), 0, st, a);
  } else {
    const int grid = ew_grid((long long)N * OH * OW * C);
    if (dtype == EMRT_F32) hipLaunchKernelGGL((resize_fwd_kernel<float, 1>), dim3(grid), dim3(256), 0, st, a);
    else if (dtype == EMRT_BF16) hipLaunchKernelGGL((resize_fwd_kernel<bf16_t, 1>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((resize_fwd_kernel<f16_t, 1>), dim3(grid), dim3(256), 0, st, a);
  }
  return check_launch("emrt_resize_bilinear_fwd");
}

// every block of a BatchNorm-operand kernel first derives the per-channel constants (C x 16 fp64 loads from L2 + fp64 math, ~2 us): few,
// long-lived blocks (4 per CU), the element loop is grid-strided
static inline int bn_operand_grid(long long total) {
  const int g = ew_grid(total);
  const int cap = g_tune.bn_operand_blocks > 0 ? g_tune.bn_operand_blocks : 1024;      // (measured: 1024 <= 2048 < 4096 on the step's five maps)
  return g > cap ? cap : g;
}

static int fill_bn_operand(BnOperand& b, const double* sums, double count, float eps, float momentum, float* mean, float* invstd,
                           float* run_mean, float* run_var, const float* gamma, const float* beta, int relu) {
  if (!sums || !mean || !invstd || !gamma || !beta || !(count > 0.0) || (run_mean != nullptr) != (run_var != nullptr)) return -1;
  b.sums = sums; b.inv_count = 1.0 / count; b.eps = eps; b.momentum = momentum; b.mean = mean; b.invstd = invstd;
  b.run_mean = run_mean; b.run_var = run_var; b.gamma = gamma; b.beta = beta; b.relu = relu;
  return 0;
}

// out = bilinear_resize([relu](BatchNorm_train(in))) in ONE pass over the raw map: emrt_bn_apply + emrt_resize_bilinear_fwd without the
// normalised intermediate (bn_operand.hpp).  `sums` are the complete fp64 batch sums [8][2C] (emrt_conv2d's bn_stats / emrt_bn_stats,
// all-reduced over ranks by the caller for SyncBatchNorm, `count` then the global row count); mean / invstd are saved for backward and the
// running statistics updated exactly as emrt_bn_apply does.  Vector path only (C % 4 == 0, 16-byte aligned, C <= 4096): anything else is
// an error, the caller then keeps the two separate launches.
extern "C" int emrt_bn_resize_bilinear_fwd(const void* in, long long in_bs, int in_ld, int IH, int IW, void* out, long long out_bs,
                                           int out_ld, int OH, int OW, int N, int C, int align_corners, const double* sums, double count,
                                           float eps, float momentum, float* mean, float* invstd, float* run_mean, float* run_var,
                                           const float* gamma, const float* beta, int relu, int dtype, void* stream) {
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  EMRT_REQUIRE(in && out, "null pointer");
  EMRT_REQUIRE(N > 0 && C > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0, "bad dims");
  ResizeArgs a;
  a.in = in; a.in_bs = in_bs; a.in_ld = in_ld; a.IH = IH; a.IW = IW;
  a.out = out; a.out_bs = out_bs; a.out_ld = out_ld; a.OH = OH; a.OW = OW;
  a.add = nullptr; a.add_bs = 0; a.add_ld = 0; a.N = N; a.C = C;
  a.ay = make_axis(IH, OH, align_corners); a.ax = make_axis(IW, OW, align_corners); a.out_nchw_f32 = 0;
  BnOperand b;
  EMRT_REQUIRE(fill_bn_operand(b, sums, count, eps, momentum, mean, invstd, run_mean, run_var, gamma, beta, relu) == 0,
               "BatchNorm operand: sums, mean, invstd, gamma, beta and a positive count are required");
  const bool v4 = C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0 && in_ld % 4 == 0 && out_ld % 4 == 0 && in_bs % 4 == 0 && out_bs % 4 == 0 &&
                  ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0);
  EMRT_REQUIRE(v4, "vector path only: C / 4 a divisor of 256, 16-byte aligned rows");
  const bool v8 = dtype != EMRT_F32 && C % 8 == 0 && 256 % (C / 8) == 0 && in_ld % 8 == 0 && out_ld % 8 == 0 && in_bs % 8 == 0 && out_bs % 8 == 0;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)2 * C * sizeof(float);
  if (v8) hipLaunchKernelGGL((resize_fwd_bn_kernel<bf16_t, 8>), dim3(bn_operand_grid((long long)N * OH * OW * (C / 8))), dim3(256), lds, st, a, b);
  else if (dtype == EMRT_F32) hipLaunchKernelGGL((resize_fwd_bn_kernel<float, 4>), dim3(bn_operand_grid((long long)N * OH * OW * (C / 4))), dim3(256), lds, st, a, b);
  else hipLaunchKernelGGL((resize_fwd_bn_kernel<bf16_t, 4>), dim3(bn_operand_grid((long long)N * OH * OW * (C / 4))), dim3(256), lds, st, a, b);
  return check_launch("emrt_bn_resize_bilinear_fwd");
}

extern "C" size_t emrt_resize_bwd_workspace_bytes(int N, int C, int OH, int IW, int dout_nchw_f32) {
  return dout_nchw_f32 ? (size_t)N * C * OH * IW * sizeof(float) : 0;
}

extern "C" int emrt_resize_bilinear_bwd(const void* dout, long long do_bs, int do_ld, int OH, int OW, void* din, long long di_bs,
                                        int di_ld, int IH, int IW, int N, int C, int align_corners, int dout_nchw_f32,
                                        void* workspace, int dtype, void* stream) {
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  EMRT_REQUIRE(dout && din, "null pointer");
  ResizeBwdArgs a;
  a.dout = dout; a.do_bs = do_bs; a.do_ld = do_ld; a.OH = OH; a.OW = OW;
  a.din = din; a.di_bs = di_bs; a.di_ld = di_ld; a.IH = IH; a.IW = IW; a.N = N; a.C = C;
  a.ay = make_axis(IH, OH, align_corners); a.ax = make_axis(IW, OW, align_corners); a.dout_nchw_f32 = dout_nchw_f32;
  hipStream_t st = (hipStream_t)stream;
  if (dout_nchw_f32) {
    EMRT_REQUIRE(workspace, "fp32 NCHW gradient needs the emrt_resize_bwd_workspace_bytes() workspace");
    float* tmp = (float*)workspace;
    hipLaunchKernelGGL(resize_bwd_nchw_w_kernel, dim3(ew_grid((long long)N * C * OH * IW)), dim3(256), 0, st, a, tmp);
    const int grid = ew_grid((long long)N * C * IH * IW);
    if (dtype == EMRT_F32) hipLaunchKernelGGL((resize_bwd_nchw_h_kernel<float>), dim3(grid), dim3(256), 0, st, a, (const float*)tmp);
    else hipLaunchKernelGGL((resize_bwd_nchw_h_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, a, (const float*)tmp);
    return check_launch("emrt_resize_bilinear_bwd");
  }
  const bool v4 = C % 4 == 0 && do_ld % 4 == 0 && di_ld % 4 == 0 && do_bs % 4 == 0 && di_bs % 4 == 0 &&
                  ((uintptr_t)dout % 16 == 0) && ((uintptr_t)din % 16 == 0);
  if (v4 && C / 4 <= RBW_THREADS && ((long long)OH * OW >= 16ll * IH * IW)) {     // >= x4 per axis: block per source pixel
    int chunks = 1;
    while (chunks < 8 && (long long)N * IH * IW * chunks < 256 && (C / 4) % (chunks * 2) == 0 && (C / 4) / (chunks * 2) >= 8) chunks *= 2;
    const dim3 grid((unsigned)(N * IH * IW), (unsigned)chunks);
    if (dtype == EMRT_F32) hipLaunchKernelGGL((resize_bwd_wide_kernel<float>), grid, dim3(RBW_THREADS), 0, st, a);
    else hipLaunchKernelGGL((resize_bwd_wide_kernel<bf16_t>), grid, dim3(RBW_THREADS), 0, st, a);
    return check_launch("emrt_resize_bilinear_bwd");
  }
  const bool v8 = v4 && dtype != EMRT_F32 && C % 8 == 0 && do_ld % 8 == 0 && di_ld % 8 == 0 && do_bs % 8 == 0 && di_bs % 8 == 0;
  if (v8) {
    const int grid = ew_grid((long long)N * IH * IW * (C / 8));
    hipLaunchKernelGGL((resize_bwd_kernel<bf16_t, 8>), dim3(grid), dim3(256), 0, st, a);
  } else if (v4) {
    const int grid = ew_grid((long long)N * IH * IW * (C / 4));
    if (dtype == EMRT_F32) hipLaunchKernelGGL((resize_bwd_kernel<float, 4>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((resize_bwd_kernel<bf16_t, 4>), dim3(grid), dim3(256), 0, st, a);
  } else {
    const int grid = ew_grid((long long)N * IH * IW * C);
    if (dtype == EMRT_F32) hipLaunchKernelGGL((resize_bwd_kernel<float, 1>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((resize_bwd_kernel<bf16_t, 1>), dim3(grid), dim3(256), 0, st, a);
  }
  return check_launch("emrt_resize_bilinear_bwd");
}

// ABI 6: the decoder's pyramid maps in ONE launch per direction.  tokens [N][sum k^2][C] dense (the k x k map of scale i starts at token
// sum_{j<i} k_j^2); outs[i] / douts[i]: NHWC maps [N][OH][OW][C] with row stride ld[i] and image stride bs[i] (channel slices of the concat
// buffer).  scales, outs, ld, bs: HOST arrays of nscales <= 4 entries.  C % 4 == 0, 16-byte aligned rows (the model's shapes: anything else
// is an error -- call emrt_resize_bilinear_fwd / _bwd per scale).  Backward: every map is >= x4 smaller than OH x OW per axis.
static int pyramid_fill(PyramidResize& m, const void* tokens, const int* scales, int nscales, void* const* maps, const int* ld, const long long* bs,
                        int OH, int OW, int N, int C, int align_corners, int esz) {
  if (!tokens || !scales || !maps || !ld || !bs || nscales < 1 || nscales > PYR_MAX || N <= 0 || C <= 0 || OH <= 0 || OW <= 0) return -1;
  if (C % 4 || ((uintptr_t)tokens % 16)) return -2;
  int ntok = 0;
  for (int i = 0; i < nscales; ++i) ntok += scales[i] * scales[i];
  int start = 0;
  m.n = nscales;
  for (int i = 0; i < nscales; ++i) {
    const int k = scales[i];
    if (k <= 0 || !maps[i] || ld[i] % 4 || bs[i] % 4 || ((uintptr_t)maps[i] % 16)) return -2;
    const char* tk = (const char*)tokens + (size_t)start * C * esz;
    ResizeArgs& f = m.f[i];
    f.in = tk; f.in_bs = (long long)ntok * C; f.in_ld = C; f.IH = k; f.IW = k;
    f.out = maps[i]; f.out_bs = bs[i]; f.out_ld = ld[i]; f.OH = OH; f.OW = OW;
    f.add = nullptr; f.add_bs = 0; f.add_ld = 0; f.N = N; f.C = C;
    f.ay = make_axis(k, OH, align_corners); f.ax = make_axis(k, OW, align_corners); f.out_nchw_f32 = 0;
    ResizeBwdArgs& b = m.b[i];
    b.dout = maps[i]; b.do_bs = bs[i]; b.do_ld = ld[i]; b.OH = OH; b.OW = OW;
    b.din = const_cast<char*>(tk); b.di_bs = (long long)ntok * C; b.di_ld = C; b.IH = k; b.IW = k; b.N = N; b.C = C;
    b.ay = f.ay; b.ax = f.ax; b.dout_nchw_f32 = 0;
    start += k * k;
  }
  return 0;
}

extern "C" int emrt_pyramid_resize_fwd(const void* tokens, const int* scales, int nscales, void* const* outs, const int* out_ld,
                                       const long long* out_bs, int OH, int OW, int N, int C, int align_corners, int dtype, void* stream) {
  EMRT_REQUIRE_FWD_DTYPE(dtype);
  PyramidResize m;
  const int rc = pyramid_fill(m, tokens, scales, nscales, outs, out_ld, out_bs, OH, OW, N, C, align_corners, dtype == EMRT_F32 ? 4 : 2);
  EMRT_REQUIRE(rc != -1, "null pointer or bad dims (1..4 scales)");
  EMRT_REQUIRE(rc == 0, "vector path only: C % 4 == 0, 16-byte aligned rows");
  const bool v8 = dtype != EMRT_F32 && C % 8 == 0;
  bool all8 = v8;
  for (int i = 0; i < nscales; ++i) all8 = all8 && out_ld[i] % 8 == 0 && out_bs[i] % 8 == 0;
  const int vec = all8 ? 8 : 4;
  m.first[0] = 0;
  for (int i = 0; i < nscales; ++i) m.first[i + 1] = m.first[i] + (long long)N * OH * OW * (C / vec);
  EMRT_REQUIRE(m.first[nscales] <= 0xffffffffll, "too many elements");
  hipStream_t st = (hipStream_t)stream;
  const int grid = ew_grid(m.first[nscales]);
  if (vec == 8) {
    if (dtype == EMRT_BF16) hipLaunchKernelGGL((resize_fwd_pyramid_kernel<bf16_t, 8>), dim3(grid), dim3(256), 0, st, m);
    else hipLaunchKernelGGL((resize_fwd_pyramid_kernel<f16_t, 8>), dim3(grid), dim3(256), 0, st, m);
  } else if (dtype == EMRT_F32) hipLaunchKernelGGL((resize_fwd_pyramid_kernel<float, 4>), dim3(grid), dim3(256), 0, st, m);
  else if (dtype == EMRT_BF16) hipLaunchKernelGGL((resize_fwd_pyramid_kernel<bf16_t, 4>), dim3(grid), dim3(256), 0, st, m);
  else hipLaunchKernelGGL((resize_fwd_pyramid_kernel<f16_t, 4>), dim3(grid), dim3(256), 0, st, m);
  return check_launch("emrt_pyramid_resize_fwd");
}

extern "C" int emrt_pyramid_resize_bwd(const void* const* douts, const int* do_ld, const long long* do_bs, int OH, int OW, void* dtokens,
                                       const int* scales, int nscales, int N, int C, int align_corners, int dtype, void* stream) {
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  PyramidResize m;
  const int rc = pyramid_fill(m, dtokens, scales, nscales, const_cast<void* const*>(douts), do_ld, do_bs, OH, OW, N, C, align_corners, dtype == EMRT_F32 ? 4 : 2);
  EMRT_REQUIRE(rc != -1, "null pointer or bad dims (1..4 scales)");
  EMRT_REQUIRE(rc == 0, "vector path only: C % 4 == 0, 16-byte aligned rows");
  EMRT_REQUIRE(C / 4 <= RBW_THREADS, "C too large");
  m.first[0] = 0;
  for (int i = 0; i < nscales; ++i) {
    EMRT_REQUIRE((long long)OH * OW >= 16ll * scales[i] * scales[i], "every pyramid map must be at least x4 smaller than the output per axis");
    // a 1x1 map has N source pixels, each collecting from the WHOLE output: its channels are split over up to 8 blocks (as the one-scale
    // launch does: same chunks, same summation order, same bits); the launch lasts as long as its slowest block
    int chunks = 1;
    while (chunks < 8 && (long long)N * scales[i] * scales[i] * chunks < 256 && (C / 4) % (chunks * 2) == 0 && (C / 4) / (chunks * 2) >= 8) chunks *= 2;
    m.chunks[i] = chunks;
    m.first[i + 1] = m.first[i] + (long long)N * scales[i] * scales[i] * chunks;
  }
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)m.first[nscales]);
  if (dtype == EMRT_F32) hipLaunchKernelGGL((resize_bwd_pyramid_kernel<float>), grid, dim3(RBW_THREADS), 0, st, m);
  else hipLaunchKernelGGL((resize_bwd_pyramid_kernel<bf16_t>), grid, dim3(RBW_THREADS), 0, st, m);
  return check_launch("emrt_pyramid_resize_bwd");
}

static int fill_pool(PoolArgs& a, const int* scales, int nscales) {
  if (nscales < 1 || nscales > 4) return -1;
  int tok = 0;
  for (int s = 0; s < 4; ++s) { a.k[s] = 1; a.tok0[s] = 0; }
  for (int s = 0; s < nscales; ++s) { a.k[s] = scales[s]; a.tok0[s] = tok; tok += scales[s] * scales[s]; }
  a.ntok = tok;
  a.nscales = nscales;
  return 0;
}

// Large bins cut over several blocks.  One block per (image, bin) makes the 1x1 scale's bin -- the whole map: 32x32 pixels x 512 bytes at
// 256x256 tiles, 64x64 at 512x512 -- ONE block's serial read through one CU (36 us / 102 us of the step for 4 / 16 MB).  Here block
// (bin, part) sums every pixel of its part of the bin and WRITES the partial, already divided by the bin's size, to its own slot of an fp32
// workspace [N][items][C]; adaptive_pool_final_kernel adds a bin's parts in part order and rounds to the output type.  No atomics: the
// result does not depend on the order the blocks ran in (two evaluations of the same tile give the same bits).
template <class T>
__global__ __launch_bounds__(POOL_THREADS) void adaptive_pool_part_kernel(PoolArgs a, float* __restrict__ ws) {
  __shared__ float red[POOL_THREADS * 4];
  // work item -> (scale, bin, part): the grid holds only as many blocks per bin as its scale's largest bin needs at 128 pixels each
  const int item = blockIdx.x % a.nitem, n = blockIdx.x / a.nitem;
  int s = 0;
  while (s + 1 < a.nscales && item >= a.item0[s + 1]) ++s;
  const int k = a.k[s], P_ = a.parts[s];
  const int t = (item - a.item0[s]) / P_, part = (item - a.item0[s]) - t * P_;
  const int oi = t / k, oj = t - oi * k;
  int h0, h1, w0, w1;
  bin_of(oi, k, a.H, h0, h1);
  bin_of(oj, k, a.W, w0, w1);
  const int bw = w1 - w0, npix = (h1 - h0) * bw;
  const int per = (npix + P_ - 1) / P_;
  const int p0 = part * per, p1 = p0 + per < npix ? p0 + per : npix;      // (p0 >= npix: a small bin of this scale has fewer parts: a zero slot)
  const int cq = a.C / 4, phases = POOL_THREADS / cq;          // host: C % 4 == 0, C / 4 <= 256 and a power of two
  const int q = (int)threadIdx.x % cq, ph = (int)threadIdx.x / cq;
  const T* ip = (const T*)a.in + (long long)n * a.in_bs + q * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  constexpr int U = 8;          // a part is <= 128 pixels (host): with C = 256 all of its loads are in flight at once
  for (int px0 = p0 + ph; px0 < p1; px0 += phases * U) {
    float v[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int pxi = px0 + u * phases;
      const int pc = pxi < p1 ? pxi : p0;
      const int hh = h0 + pc / bw, ww = w0 + pc % bw;
      Vec4<T>::load(ip + ((long long)hh * a.W + ww) * a.in_ld, v[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = px0 + u * phases < p1;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += ok ? v[u][e] : 0.f;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[threadIdx.x * 4 + e] = acc[e];
  __syncthreads();
  if (ph == 0) {
    for (int o = 1; o < phases; ++o)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += red[(threadIdx.x + o * cq) * 4 + e];
    const float inv = 1.f / (float)npix;
    float4 o4;
    o4.x = acc[0] * inv; o4.y = acc[1] * inv; o4.z = acc[2] * inv; o4.w = acc[3] * inv;
    *reinterpret_cast<float4*>(ws + ((long long)n * a.nitem + item) * a.C + q * 4) = o4;
  }
}

template <class T>
__global__ __launch_bounds__(256) void adaptive_pool_final_kernel(PoolArgs a, const float* __restrict__ ws) {
  const long long total = (long long)a.N * a.ntok * (a.C / 4);
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int cq = a.C / 4;
    const int q = (int)(idx % cq);
    const long long nt = idx / cq;
    const int tok = (int)(nt % a.ntok), n = (int)(nt / a.ntok);
    int s = 0;
    while (s + 1 < a.nscales && tok >= a.tok0[s + 1]) ++s;
    const int P_ = a.parts[s];
    const float* wp = ws + ((long long)n * a.nitem + a.item0[s] + (long long)(tok - a.tok0[s]) * P_) * a.C + q * 4;
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < P_; ++p) {          // fixed order
      const float4 v = *reinterpret_cast<const float4*>(wp + (long long)p * a.C);
      o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w;
    }
    Vec4<T>::store((T*)a.out + (long long)n * a.out_bs + (long long)tok * a.out_ld + q * 4, o);
  }
}

// The split plan of emrt_adaptive_avgpool_fwd: parts per bin of each scale (128 pixels per part), items = sum k^2 * parts; false when the
// map is small (largest bin < 512 pixels) or the channel count is off the vector path: one block per bin then.
static bool pool_split_plan(PoolArgs& a, int H, int W, int C, const int* scales, int nscales) {
  int kmin = scales[0];
  for (int i = 1; i < nscales; ++i) kmin = scales[i] < kmin ? scales[i] : kmin;
  const long long big = (long long)((H + kmin - 1) / kmin + 1) * ((W + kmin - 1) / kmin + 1);      // pixels of the largest bin (upper bound)
  const int cq = C / 4;
  if (!((C % 4 == 0) && cq <= 256 && (cq & (cq - 1)) == 0 && big >= 512)) return false;
  a.nitem = 0;
  for (int i = 0; i < 4; ++i) { a.parts[i] = 1; a.item0[i] = 0; }
  for (int i = 0; i < nscales; ++i) {
    const long long mx = (long long)((H + scales[i] - 1) / scales[i] + 1) * ((W + scales[i] - 1) / scales[i] + 1);      // largest bin of this scale (upper bound)
    long long pp = (mx + 127) / 128;
    if (pp > 256) pp = 256;
    a.parts[i] = (int)pp;
    a.item0[i] = a.nitem;
    a.nitem += scales[i] * scales[i] * (int)pp;
  }
  return true;
}

// bytes of the fp32 partial-sum workspace emrt_adaptive_avgpool_fwd can use for this call (0: the map is pooled by one block per bin and
// no workspace is needed).  The workspace need not be cleared.
extern "C" size_t emrt_adaptive_avgpool_workspace_bytes(int H, int W, int N, int C, const int* scales /*host*/, int nscales) {
  PoolArgs a;
  if (!scales || N <= 0 || H <= 0 || W <= 0 || C <= 0 || fill_pool(a, scales, nscales) != 0) return 0;
  if (!pool_split_plan(a, H, W, C, scales, nscales)) return 0;
  return (size_t)N * a.nitem * C * sizeof(float);
}

// out tokens [N][sum k^2][C] (row stride out_ld, batch stride out_bs)
extern "C" int emrt_adaptive_avgpool_fwd(const void* in, long long in_bs, int in_ld, int H, int W, void* out, long long out_bs,
                                         int out_ld, int N, int C, const int* scales /*host*/, int nscales, float* workspace, size_t workspace_bytes,
                                         int dtype, void* stream) {
  EMRT_REQUIRE_FWD_DTYPE(dtype);
  EMRT_REQUIRE(in && out && scales, "null pointer");
  PoolArgs a;
  a.in = in; a.in_bs = in_bs; a.in_ld = in_ld; a.H = H; a.W = W; a.out = out; a.out_bs = out_bs; a.out_ld = out_ld; a.N = N; a.C = C;
  EMRT_REQUIRE(fill_pool(a, scales, nscales) == 0, "1..4 scales supported");
  hipStream_t st = (hipStream_t)stream;
  if (workspace) {
    const int esz = dtype == EMRT_F32 ? 4 : 2;
    const bool vec = (in_ld % 4 == 0) && (in_bs % 4 == 0) && (out_ld % 4 == 0) && (out_bs % 4 == 0) && (((uintptr_t)in) % (4 * esz) == 0) &&
                     (((uintptr_t)out) % (4 * esz) == 0) && (((uintptr_t)workspace) % 16 == 0);
    if (vec && pool_split_plan(a, H, W, C, scales, nscales)) {
      EMRT_REQUIRE(workspace_bytes >= (size_t)N * a.nitem * C * sizeof(float), "workspace smaller than emrt_adaptive_avgpool_workspace_bytes()");
      const dim3 grid((unsigned)(N * a.nitem));
      const int fg = ew_grid((long long)N * a.ntok * (C / 4));
      if (dtype == EMRT_F32) {
        hipLaunchKernelGGL((adaptive_pool_part_kernel<float>), grid, dim3(POOL_THREADS), 0, st, a, workspace);
        hipLaunchKernelGGL((adaptive_pool_final_kernel<float>), dim3(fg), dim3(256), 0, st, a, (const float*)workspace);
      } else if (dtype == EMRT_BF16) {
        hipLaunchKernelGGL((adaptive_pool_part_kernel<bf16_t>), grid, dim3(POOL_THREADS), 0, st, a, workspace);
        hipLaunchKernelGGL((adaptive_pool_final_kernel<bf16_t>), dim3(fg), dim3(256), 0, st, a, (const float*)workspace);
      } else {
        hipLaunchKernelGGL((adaptive_pool_part_kernel<f16_t>), grid, dim3(POOL_THREADS), 0, st, a, workspace);
        hipLaunchKernelGGL((adaptive_pool_final_kernel<f16_t>), dim3(fg), dim3(256), 0, st, a, (const float*)workspace);
      }
      return check_launch("emrt_adaptive_avgpool_fwd");
    }
  }
  if (dtype == EMRT_F32) hipLaunchKernelGGL((adaptive_pool_fwd_kernel<float>), dim3(N * a.ntok), dim3(POOL_THREADS), 0, st, a);
  else if (dtype == EMRT_BF16) hipLaunchKernelGGL((adaptive_pool_fwd_kernel<bf16_t>), dim3(N * a.ntok), dim3(POOL_THREADS), 0, st, a);
  else hipLaunchKernelGGL((adaptive_pool_fwd_kernel<f16_t>), dim3(N * a.ntok), dim3(POOL_THREADS), 0, st, a);
  return check_launch("emrt_adaptive_avgpool_fwd");
}

extern "C" int emrt_adaptive_avgpool_bwd(const void* dout, long long do_bs, int do_ld, void* din, long long di_bs, int di_ld, int H,
                                         int W, int N, int C, const int* scales, int nscales, int dtype, void* stream) {
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  EMRT_REQUIRE(dout && din && scales, "null pointer");
  PoolArgs a;
  a.in = din; a.in_bs = di_bs; a.in_ld = di_ld; a.H = H; a.W = W; a.out = const_cast<void*>(dout); a.out_bs = do_bs; a.out_ld = do_ld;
  a.N = N; a.C = C;
  EMRT_REQUIRE(fill_pool(a, scales, nscales) == 0, "1..4 scales supported");
  hipStream_t st = (hipStream_t)stream;
  const int esz = dtype == EMRT_F32 ? 4 : 2;
  const bool vec = C % 4 == 0 && do_ld % 4 == 0 && do_bs % 4 == 0 && di_ld % 4 == 0 && di_bs % 4 == 0 &&
                   ((uintptr_t)dout % (4 * esz) == 0) && ((uintptr_t)din % (4 * esz) == 0);
  // the table kernel keeps at most TWO bins per row / column and scale, which holds while every scale k <= min(H, W) (bins then overlap by at
  // most one pixel); a pooled map smaller than a scale (k > S: a pixel can belong to 3 or more bins) takes the general gather kernels
  bool two_bins = true;
  for (int i = 0; i < nscales; ++i) two_bins = two_bins && scales[i] <= (H < W ? H : W);
  if (vec && H + W <= 512 && two_bins) {
    const int grid = ew_grid((long long)N * H * W * (C / 4));
    DT2(dtype, adaptive_pool_bwd_tab_kernel, grid, a);
  } else if (vec) {
    const int grid = ew_grid((long long)N * H * W * (C / 4));
    DT2(dtype, adaptive_pool_bwd_vec_kernel, grid, a);
  } else {
    const int grid = ew_grid((long long)N * H * W * C);
    DT2(dtype, adaptive_pool_bwd_kernel, grid, a);
  }
  return check_launch("emrt_adaptive_avgpool_bwd");
}

extern "C" int emrt_maxpool_fwd(const void* in, void* out, unsigned char* argmax, int N, int H, int W, int C, int k, int stride, int pad,
                                int dtype, void* stream) {
  EMRT_REQUIRE_FWD_DTYPE(dtype);
  EMRT_REQUIRE(in && out, "null pointer");
  EMRT_REQUIRE(k > 0 && k <= 15 && stride > 0 && pad >= 0 && pad < k, "bad window");
  MaxPoolArgs a;
  memset(&a, 0, sizeof(a));
  a.in = in; a.out = out; a.arg = argmax; a.N = N; a.H = H; a.W = W; a.C = C; a.k = k; a.stride = stride; a.pad = pad;
  a.OH = (H + 2 * pad - k) / stride + 1; a.OW = (W + 2 * pad - k) / stride + 1;
  hipStream_t st = (hipStream_t)stream;
  const int esz = dtype == EMRT_F32 ? 4 : 2;
  if (C % 8 == 0 && ((uintptr_t)in % (8 * esz) == 0) && ((uintptr_t)out % (8 * esz) == 0) && (!argmax || (uintptr_t)argmax % 8 == 0)) {
    const int grid = ew_grid((long long)N * a.OH * a.OW * (C / 8));
    DT3(dtype, maxpool_fwd_vec8_kernel, grid, a);
    return check_launch("emrt_maxpool_fwd");
  }
  const int grid = ew_grid((long long)N * a.OH * a.OW * C);
  DT3(dtype, maxpool_fwd_kernel, grid, a);
  return check_launch("emrt_maxpool_fwd");
}

// out = maxpool([relu](BatchNorm_train(in))) in one pass over the raw map (see emrt_bn_resize_bilinear_fwd).  C % 8 == 0, C <= 4096.
extern "C" int emrt_bn_maxpool_fwd(const void* in, void* out, unsigned char* argmax, int N, int H, int W, int C, int k, int stride, int pad,
                                   const double* sums, double count, float eps, float momentum, float* mean, float* invstd,
                                   float* run_mean, float* run_var, const float* gamma, const float* beta, int relu, int dtype, void* stream) {
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  EMRT_REQUIRE(in && out, "null pointer");
  EMRT_REQUIRE(k > 0 && k <= 15 && stride > 0 && pad >= 0 && pad < k, "bad window");
  MaxPoolArgs a;
  memset(&a, 0, sizeof(a));
  a.in = in; a.out = out; a.arg = argmax; a.N = N; a.H = H; a.W = W; a.C = C; a.k = k; a.stride = stride; a.pad = pad;
  a.OH = (H + 2 * pad - k) / stride + 1; a.OW = (W + 2 * pad - k) / stride + 1;
  BnOperand b;
  EMRT_REQUIRE(fill_bn_operand(b, sums, count, eps, momentum, mean, invstd, run_mean, run_var, gamma, beta, relu) == 0,
               "BatchNorm operand: sums, mean, invstd, gamma, beta and a positive count are required");
  const int esz = dtype == EMRT_F32 ? 4 : 2;
  EMRT_REQUIRE(C % 8 == 0 && C <= 4096 && ((uintptr_t)in % (8 * esz) == 0) && ((uintptr_t)out % (8 * esz) == 0) && (!argmax || (uintptr_t)argmax % 8 == 0),
               "vector path only: C % 8 == 0, C <= 4096, aligned maps");
  hipStream_t st = (hipStream_t)stream;
  const int grid = bn_operand_grid((long long)N * a.OH * a.OW * (C / 8));
  const size_t lds = (size_t)2 * C * sizeof(float);
  if (dtype == EMRT_F32) hipLaunchKernelGGL((maxpool_fwd_bn_kernel<float>), dim3(grid), dim3(256), lds, st, a, b);
  else hipLaunchKernelGGL((maxpool_fwd_bn_kernel<bf16_t>), dim3(grid), dim3(256), lds, st, a, b);
  return check_launch("emrt_bn_maxpool_fwd");
}

extern "C" int emrt_maxpool_bwd(const unsigned char* argmax, const void* dout, void* din, int N, int H, int W, int C, int k, int stride,
                                int pad, int dtype, void* stream) {
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  EMRT_REQUIRE(argmax && dout && din, "null pointer");
  EMRT_REQUIRE(k > 0 && k <= 15 && stride > 0 && pad >= 0 && pad < k, "bad window");
  MaxPoolArgs a;
  memset(&a, 0, sizeof(a));
  a.arg = const_cast<unsigned char*>(argmax); a.dout = dout; a.din = din; a.N = N; a.H = H; a.W = W; a.C = C; a.k = k; a.stride = stride;
  a.pad = pad;
  a.OH = (H + 2 * pad - k) / stride + 1; a.OW = (W + 2 * pad - k) / stride + 1;
  hipStream_t st = (hipStream_t)stream;
  const bool v4 = C % 4 == 0 && ((uintptr_t)dout % 16 == 0) && ((uintptr_t)din % 16 == 0) && ((uintptr_t)argmax % 4 == 0);
  if (v4) {
    const int grid = ew_grid((long long)N * H * W * (C / 4));
    if (dtype == EMRT_F32) hipLaunchKernelGGL((maxpool_bwd_kernel<float, 4>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((maxpool_bwd_kernel<bf16_t, 4>), dim3(grid), dim3(256), 0, st, a);
  } else {
    const int grid = ew_grid((long long)N * H * W * C);
    if (dtype == EMRT_F32) hipLaunchKernelGGL((maxpool_bwd_kernel<float, 1>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((maxpool_bwd_kernel<bf16_t, 1>), dim3(grid), dim3(256), 0, st, a);
  }
  return check_launch("emrt_maxpool_bwd");
}

extern "C" int emrt_nchw_to_nhwc(const float* in, void* out, int N, int C, int H, int W, int c_out, int dtype, void* stream) {
  EMRT_REQUIRE_FWD_DTYPE(dtype);
  EMRT_REQUIRE(in && out, "null pointer");
  EMRT_REQUIRE(c_out >= C, "the output has at least the input's channels");
  hipStream_t st = (hipStream_t)stream;
  if ((c_out == 8 || c_out == 4) && C <= c_out && ((uintptr_t)out % 32 == 0)) {
    const long long HW = (long long)H * W;
    const int g2 = ew_grid((long long)N * HW);
#define INGEST(T, CO) hipLaunchKernelGGL((nchw_to_nhwc_pix_kernel<T, CO>), dim3(g2), dim3(256), 0, st, in, (T*)out, N, C, HW)
    if (c_out == 8) { if (dtype == EMRT_F32) INGEST(float, 8); else if (dtype == EMRT_BF16) INGEST(bf16_t, 8); else INGEST(f16_t, 8); }
    else { if (dtype == EMRT_F32) INGEST(float, 4); else if (dtype == EMRT_BF16) INGEST(bf16_t, 4); else INGEST(f16_t, 4); }
#undef INGEST
    return check_launch("emrt_nchw_to_nhwc");
  }
  const int grid = ew_grid((long long)N * c_out * H * W);
  if (dtype == EMRT_F32) hipLaunchKernelGGL((nchw_to_nhwc_kernel<float>), dim3(grid), dim3(256), 0, st, in, (float*)out, N, C, H, W, c_out);
  else if (dtype == EMRT_BF16) hipLaunchKernelGGL((nchw_to_nhwc_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, in, (bf16_t*)out, N, C, H, W, c_out);
  else hipLaunchKernelGGL((nchw_to_nhwc_kernel<f16_t>), dim3(grid), dim3(256), 0, st, in, (f16_t*)out, N, C, H, W, c_out);
  return check_launch("emrt_nchw_to_nhwc");
}

// ------------------------------------------------------------------------------------------------
// Sliding-window inference glue (reference: src/api/infer.py:22-80, 145-155): crop the windows of one image into a
// batch, accumulate the windows' logits with a hit count, normalise, argmax.  fp32 NCHW throughout, as the reference.
// ------------------------------------------------------------------------------------------------
#define EMRT_MAX_WINDOWS 64
struct WindowArgs {
  int n, C, H, W, ch, cw;
  int y0[EMRT_MAX_WINDOWS], x0[EMRT_MAX_WINDOWS];
};

__global__ __launch_bounds__(256) void crop_windows_kernel(const float* __restrict__ img, float* __restrict__ batch, WindowArgs a) {
  const long long total = (long long)a.n * a.C * a.ch * a.cw;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    int x, y, c, j;
    unravel4(idx, a.cw, a.ch, a.C, total <= 0xffffffffll, x, y, c, j);
    batch[idx] = img[((long long)c * a.H + a.y0[j] + y) * a.W + a.x0[j] + x];
  }
}

// one thread per image pixel and class: sums the windows that cover it (any overlap pattern, deterministic order)
__global__ __launch_bounds__(256) void window_accumulate_kernel(const float* __restrict__ logits, float* __restrict__ final,
                                                                float* __restrict__ count, WindowArgs a) {
  const long long total = (long long)a.C * a.H * a.W;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    int x, y;
    long long cc;
    unravel3(idx, a.W, a.H, total <= 0xffffffffll, x, y, cc);
    const int c = (int)cc;
    float s = 0.f, k = 0.f;
    for (int j = 0; j < a.n; ++j) {
      const int yy = y - a.y0[j], xx = x - a.x0[j];
      if ((unsigned)yy < (unsigned)a.ch && (unsigned)xx < (unsigned)a.cw) {
        s += logits[(((long long)j * a.C + c) * a.ch + yy) * a.cw + xx];
        k += 1.f;
      }
    }
    final[idx] += s;
    if (c == 0) count[(long long)y * a.W + x] += k;
  }
}

// four consecutive x per thread when every window origin / width and the image width are multiples of 4 (the regular grids of the
// configs): one cover test and one 16-byte load per window instead of four
__global__ __launch_bounds__(256) void window_accumulate_vec4_kernel(const float* __restrict__ logits, float* __restrict__ final,
                                                                     float* __restrict__ count, WindowArgs a) {
  const int W4 = a.W / 4;
  const long long total = (long long)a.C * a.H * W4;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    int x, y;
    long long cc;
    unravel3(idx, W4, a.H, total <= 0xffffffffll, x, y, cc);
    x *= 4;
    const int c = (int)cc;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    float k = 0.f;
    for (int j = 0; j < a.n; ++j) {
      const int yy = y - a.y0[j], xx = x - a.x0[j];
      if ((unsigned)yy < (unsigned)a.ch && (unsigned)xx < (unsigned)a.cw) {
        const float4 v = *reinterpret_cast<const float4*>(logits + (((long long)j * a.C + c) * a.ch + yy) * a.cw + xx);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        k += 1.f;
      }
    }
    float4* fp = reinterpret_cast<float4*>(final + ((long long)c * a.H + y) * a.W + x);
    float4 f = *fp;
    f.x += s.x; f.y += s.y; f.z += s.z; f.w += s.w;
    *fp = f;
    if (c == 0) {
      float4* cp = reinterpret_cast<float4*>(count + (long long)y * a.W + x);
      float4 cc = *cp;
      cc.x += k; cc.y += k; cc.z += k; cc.w += k;
      *cp = cc;
    }
  }
}

__global__ __launch_bounds__(256) void window_normalise_kernel(const float* __restrict__ final, const float* __restrict__ count,
                                                               float* __restrict__ out, int C, long long HW) {
  const long long total = (long long)C * HW;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x)
    out[idx] = final[idx] / count[idx % HW];        // uncovered pixels: 0/0 = NaN, exactly as the reference (infer.py:79)
}

// argmax over the class axis of fp32 [N][C][H][W]; first maximum wins, a NaN counts as the maximum (torch / paddle)
__global__ __launch_bounds__(256) void argmax_nchw_kernel(const float* __restrict__ logits, int* __restrict__ pred, int N, int C, long long HW) {
  const long long total = (long long)N * HW;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long n = idx / HW, p = idx - n * HW;
    const float* lp = logits + n * C * HW + p;
    float best = lp[0];
    int bi = 0;
    for (int c = 1; c < C; ++c) {
      const float v = lp[(long long)c * HW];
      if (best == best && (v > best || v != v)) { best = v; bi = c; }
    }
    pred[idx] = bi;
  }
}

static int fill_windows(WindowArgs& a, const int* origins_yx, int n, int C, int H, int W, int ch, int cw) {
  if (n < 1 || n > EMRT_MAX_WINDOWS) return -1;
  a.n = n; a.C = C; a.H = H; a.W = W; a.ch = ch; a.cw = cw;
  for (int j = 0; j < n; ++j) {
    a.y0[j] = origins_yx[2 * j];
    a.x0[j] = origins_yx[2 * j + 1];
    if (a.y0[j] < 0 || a.x0[j] < 0 || a.y0[j] + ch > H || a.x0[j] + cw > W) return -1;
  }
  return 0;
}

extern "C" int emrt_crop_windows(const float* img, float* batch, const int* origins_yx /*host, [n][2]*/, int n, int C, int H, int W,
                                 int ch, int cw, void* stream) {
  EMRT_REQUIRE(img && batch && origins_yx, "null pointer");
  WindowArgs a;
  EMRT_REQUIRE(fill_windows(a, origins_yx, n, C, H, W, ch, cw) == 0, "1..64 windows inside the image");
  hipLaunchKernelGGL(crop_windows_kernel, dim3(ew_grid((long long)n * C * ch * cw)), dim3(256), 0, (hipStream_t)stream, img, batch, a);
  return check_launch("emrt_crop_windows");
}

extern "C" int emrt_window_accumulate(const float* logits, float* final, float* count, const int* origins_yx /*host*/, int n, int C,
                                      int H, int W, int ch, int cw, void* stream) {
  EMRT_REQUIRE(logits && final && count && origins_yx, "null pointer");
  WindowArgs a;
  EMRT_REQUIRE(fill_windows(a, origins_yx, n, C, H, W, ch, cw) == 0, "1..64 windows inside the image");
  bool v4 = W % 4 == 0 && cw % 4 == 0 && ((uintptr_t)logits % 16 == 0) && ((uintptr_t)final % 16 == 0) && ((uintptr_t)count % 16 == 0);
  for (int j = 0; j < n; ++j) v4 = v4 && a.x0[j] % 4 == 0;
  if (v4) hipLaunchKernelGGL(window_accumulate_vec4_kernel, dim3(ew_grid((long long)C * H * (W / 4))), dim3(256), 0, (hipStream_t)stream, logits, final, count, a);
  else hipLaunchKernelGGL(window_accumulate_kernel, dim3(ew_grid((long long)C * H * W)), dim3(256), 0, (hipStream_t)stream, logits, final, count, a);
  return check_launch("emrt_window_accumulate");
}

extern "C" int emrt_window_normalise(const float* final, const float* count, float* out, int C, int H, int W, void* stream) {
  EMRT_REQUIRE(final && count && out, "null pointer");
  hipLaunchKernelGGL(window_normalise_kernel, dim3(ew_grid((long long)C * H * W)), dim3(256), 0, (hipStream_t)stream, final, count, out, C,
                     (long long)H * W);
  return check_launch("emrt_window_normalise");
}

extern "C" int emrt_argmax_nchw(const float* logits, int* pred, int N, int C, int H, int W, void* stream) {
  EMRT_REQUIRE(logits && pred && C >= 1, "bad arguments");
  hipLaunchKernelGGL(argmax_nchw_kernel, dim3(ew_grid((long long)N * H * W)), dim3(256), 0, (hipStream_t)stream, logits, pred, N, C, (long long)H * W);
  return check_launch("emrt_argmax_nchw");
}

// ---- multi-scale / flip inference glue (reference: src/api/infer.py:160-260): horizontal flip of fp32 NCHW maps and
// "final += softmax(logits, axis=1)" over the class axis
__global__ __launch_bounds__(256) void flip_w_kernel(const float* __restrict__ in, float* __restrict__ out, long long rows, int W) {
  const long long total = rows * W;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long r = idx / W;
    const int x = (int)(idx - r * W);
    out[idx] = in[r * W + (W - 1 - x)];
  }
}

__global__ __launch_bounds__(256) void softmax_nchw_acc_kernel(const float* __restrict__ logits, float* __restrict__ acc, int N, int C, long long HW) {
  const long long total = (long long)N * HW;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long n = idx / HW, p = idx - n * HW;
    const float* lp = logits + n * C * HW + p;
    float* ap = acc + n * C * HW + p;
    float mx = lp[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, lp[(long long)c * HW]);
    float den = 0.f;
    for (int c = 0; c < C; ++c) den += expf(lp[(long long)c * HW] - mx);
    const float inv = 1.f / den;
    for (int c = 0; c < C; ++c) ap[(long long)c * HW] += expf(lp[(long long)c * HW] - mx) * inv;
  }
}

extern "C" int emrt_flip_w(const float* in, float* out, long long rows, int W, void* stream) {
  EMRT_REQUIRE(in && out && in != out, "null or aliased pointers");
  hipLaunchKernelGGL(flip_w_kernel, dim3(ew_grid(rows * W)), dim3(256), 0, (hipStream_t)stream, in, out, rows, W);
  return check_launch("emrt_flip_w");
}

extern "C" int emrt_softmax_nchw_acc(const float* logits, float* acc, int N, int C, int H, int W, void* stream) {
  EMRT_REQUIRE(logits && acc && C >= 1, "bad arguments");
  hipLaunchKernelGGL(softmax_nchw_acc_kernel, dim3(ew_grid((long long)N * H * W)), dim3(256), 0, (hipStream_t)stream, logits, acc, N, C, (long long)H * W);
  return check_launch("emrt_softmax_nchw_acc");
}
