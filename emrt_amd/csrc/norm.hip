// Normalisation kernels for gfx950: BatchNorm (train/eval, +residual, +ReLU), GroupNorm(+GELU)(+residual),
// residual-add + LayerNorm, and per-channel column sums (bias gradients).  All HBM-bound: rows of C
// contiguous channels (NHWC), 4 channels per thread (16 B f32 / 8 B bf16 accesses), fp32 statistics.
//
// Reference call sites replaced: nn.BatchNorm2D / nn.SyncBatchNorm
// (paddle_vision_resnet.py:132-147, paddle_EMRT.py:18,64,131,139-141,203,206, fcn_head.py:53),
// nn.GroupNorm(32,256)+nn.GELU (transformer_encoder_decoder.py:125-144,378), nn.LayerNorm(256)
// (transformer_encoder_decoder.py:116,123,251,256,264).
#include "common.hpp"
#include "bn_operand.hpp"
#include <stdlib.h>

using namespace emrt;

// ------------------------------------------------------------------------------------------------
// column reductions over rows of a [M][C] (row stride ld) matrix.
// MODE 0: sum x, sum x^2                       (BN statistics)
// MODE 1: sum dy', sum dy' * xhat              (BN backward; dy' = dy masked by y > 0 when relu)
// MODE 2: sum x                                (bias gradient)
// partial[blk][2][C]; a finalize kernel combines the blocks in double precision (deterministic).
// Block = 256 threads = TX channel-quads x TY row lanes; grid.y walks channel chunks of 4*TX.
// ------------------------------------------------------------------------------------------------
template <class T, int MODE>
__global__ __launch_bounds__(256) void col_reduce_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dy, int lddy,
                                                         const T* __restrict__ y, int ldy, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, long long M, int C, int tx_n,
                                                         float* __restrict__ partial, long long rpb, long long bs,
                                                         double* __restrict__ dsums, float* __restrict__ facc,
                                                         const float* __restrict__ mgamma = nullptr, const float* __restrict__ mbeta = nullptr) {
  __shared__ float red[256 * 8];
  const int ty_n = 256 / tx_n;
  const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
  const int c = (blockIdx.y * tx_n + tx) * 4;
  float s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  if (c < C) {
    float mu[4] = {0, 0, 0, 0}, is[4] = {1, 1, 1, 1};
    float msc[MODE == 3 ? 4 : 1], msh[MODE == 3 ? 4 : 1];      // MODE 3 = MODE 1 with the ReLU mask re-derived from x (bn_operand.hpp)
    if (MODE == 1 || MODE == 3) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { mu[e] = mean[c + e]; is[e] = invstd[c + e]; }
      if constexpr (MODE == 3) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bn_scale_shift(mu[e], is[e], mgamma[c + e], mbeta[c + e], msc[e], msh[e]);
      }
    }
    for (long long r = (long long)blockIdx.x * ty_n + ty; r < M; r += (long long)gridDim.x * ty_n) {
      float v[4];
      if (MODE == 0) {
        Vec4<T>::load(x + r * ldx + c, v);   // BN statistics: always dense rows
#pragma unroll
        for (int e = 0; e < 4; ++e) { s0[e] += v[e]; s1[e] = fmaf(v[e], v[e], s1[e]); }
      } else if (MODE == 1 || MODE == 3) {
        // four rows per iteration, all their loads issued before the first use: one row per trip left 4-6 KB in flight per
        // block and the kernel at the latency of its load chain (13 us for 33 MB)
        const long long rs = (long long)gridDim.x * ty_n;
        float vv[4][4], gg[4][4], oo[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long long ru = r + u * rs < M ? r + u * rs : r;
          Vec4<T>::load(x + ru * ldx + c, vv[u]);
          Vec4<T>::load(dy + ru * lddy + c, gg[u]);
          if (y) Vec4<T>::load(y + ru * ldy + c, oo[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool ok = r + u * rs < M;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float g = ok ? gg[u][e] : 0.f;
            if (y) g = oo[u][e] > 0.f ? g : 0.f;
            if constexpr (MODE == 3) g = fmaf(vv[u][e], msc[e], msh[e]) > 0.f ? g : 0.f;
            s0[e] += g;
            s1[e] = fmaf(g, (vv[u][e] - mu[e]) * is[e], s1[e]);
          }
        }
        r += 3 * rs;
      } else {
        const long long bb = (M | rpb) <= 0xffffffffll ? (long long)((unsigned)r / (unsigned)rpb) : r / rpb;   // batch-strided rows: row r = (batch bb, row r - bb*rpb); 32-bit division when it can be
        Vec4<T>::load(x + bb * bs + (r - bb * rpb) * ldx + c, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) s0[e] += v[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[threadIdx.x * 8 + e] = s0[e]; red[threadIdx.x * 8 + 4 + e] = s1[e]; }
  __syncthreads();
  if (ty == 0 && c < C) {
    for (int t = 1; t < ty_n; ++t) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { s0[e] += red[(t * tx_n + tx) * 8 + e]; s1[e] += red[(t * tx_n + tx) * 8 + 4 + e]; }
    }
    if (dsums) {      // BatchNorm path: fp64 atomics into one of 8 replicas of the [2C] sums (no partial buffer, no combine launch)
      double* rep = dsums + (long long)(blockIdx.x & 7) * 2 * C;
#pragma unroll
      for (int e = 0; e < 4; ++e) { atomicAdd(rep + c + e, (double)s0[e]); atomicAdd(rep + C + c + e, (double)s1[e]); }
    } else if (facc) {    // bias / embedding gradients: fp32 atomics straight into the gradient (no combine launch)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c + e < C) atomicAdd(facc + c + e, s0[e]);
    } else {
      float* pp = partial + (long long)blockIdx.x * 2 * C;
#pragma unroll
      for (int e = 0; e < 4; ++e) { pp[c + e] = s0[e]; pp[C + c + e] = s1[e]; }
    }
  }
}

static inline void col_reduce_geometry(long long M, int C, int& tx_n, int& gx, int& gy) {
  int quads = (C + 3) / 4;
  tx_n = 64;
  while (tx_n > 1 && tx_n / 2 >= quads) tx_n /= 2;
  gy = (quads + tx_n - 1) / tx_n;
  int ty_n = 256 / tx_n;
  long long want = (M + (long long)ty_n * 8 - 1) / ((long long)ty_n * 8);
  long long cap = 512 / gy;
  if (cap < 1) cap = 1;
  gx = (int)(want < 1 ? 1 : (want > cap ? cap : want));
}

// BN train finalize: partial[nblk][2][C] (local sums) -> mean, invstd (saved for bwd), running stats update
// (Paddle convention: running = mom*running + (1-mom)*batch, biased variance).
// `count` is the number of rows the sums cover (after a cross-rank all-reduce of `sums` for SyncBN it is
// the global count).  Split in two so SyncBN can all-reduce between them.
__global__ __launch_bounds__(256) void bn_sum_partials_kernel(const float* __restrict__ partial, int nblk, int C, float* __restrict__ sums /*[2][C]*/) {
  // 32 columns x 8 partial-lanes per block; each lane walks every 8th partial row (coalesced over columns), fp64 combine
  __shared__ double red[8][33];
  const int cx = threadIdx.x & 31, py = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  double s = 0.0;
  if (c < 2 * C)
    for (int b = py; b < nblk; b += 8) s += (double)partial[(long long)b * 2 * C + c];
  red[py][cx] = s;
  __syncthreads();
  if (py == 0 && c < 2 * C) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][cx];
    sums[c] = (float)t;
  }
}

// partial[nblk][2][C] -> dbeta[c] += column sums of slot 0, dgamma[c] += column sums of slot 1 (either may be null):
// bn_sum_partials + bn_bwd_finalize in one launch for the callers that do not need the sums afterwards.
__global__ __launch_bounds__(256) void partials_acc_kernel(const float* __restrict__ partial, int nblk, int C, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta) {
  __shared__ double red[8][33];
  const int cx = threadIdx.x & 31, py = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  double s = 0.0;
  if (c < 2 * C)
    for (int b = py; b < nblk; b += 8) s += (double)partial[(long long)b * 2 * C + c];
  red[py][cx] = s;
  __syncthreads();
  if (py == 0 && c < 2 * C) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][cx];
    if (c < C) { if (dbeta) dbeta[c] += (float)t; }
    else if (dgamma) dgamma[c - C] += (float)t;
  }
}

// BN backward finalize: sums[2][C] = (sum dy', sum dy'*xhat) -> dgamma += , dbeta +=   (sums kept for dx)
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ sums, int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (dbeta) dbeta[c] += sums[c];
  if (dgamma) dgamma[c] += sums[C + c];
}

// y = [relu]((x - mean) * invstd * gamma + beta [+ res]).  Training (sums != null): mean/invstd come from the fp64
// sums; block 0 also saves them for backward and updates the running statistics (Paddle convention: momentum 0.9 =>
// running = 0.9*running + 0.1*batch, biased variance).  Threads own a fixed channel quad, so the per-channel constants
// are computed once per thread and the row loop is a pure streaming fma.
template <class T, bool JOIN = false>
__global__ __launch_bounds__(512) void bn_apply_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ res, int ldres,
                                                       T* __restrict__ y, int ldy, const double* __restrict__ sums, double inv_count,
                                                       float eps, float momentum, float* __restrict__ mean_out,
                                                       float* __restrict__ invstd_out, float* __restrict__ run_mean,
                                                       float* __restrict__ run_var, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, long long M, int C, int relu, int rows_per_pass,
                                                       BnOperand rbn) {
  // rbn.sums != null: `res` is the RAW output of the shortcut's conv and its own training-mode BatchNorm (no ReLU) is applied as it is
  // loaded -- out = relu(BN(x) + BN_shortcut(res)), the first block of a ResNet stage (paddle_vision_resnet.py:132-147, 226-233) -- instead
  // of a separate launch that writes the normalised shortcut for this one to read back
  const int quads = C / 4;
  const int c = (threadIdx.x % quads) * 4;
  const int lane_row = threadIdx.x / quads;
  // A block only streams ~8 KB, so its time is latency: the first row's loads go out BEFORE the per-channel preamble
  // (fp64 replica sums -> scale/shift -> LDS -> barrier) and every later row is fetched one iteration ahead.
  const long long rstep = (long long)gridDim.x * rows_per_pass;
  long long r = (long long)blockIdx.x * rows_per_pass + lane_row;
  float v[4], w[4];
  if (r < M) {
    Vec4<T>::load(x + r * ldx + c, v);
    if (res) Vec4<T>::load(res + r * ldres + c, w);
  }
  // per-channel scale/shift computed cooperatively (one channel per thread) and shared through LDS
  extern __shared__ float bn_lds[];          // [2][C] (+ [2][C] for the shortcut's BatchNorm)
  if constexpr (JOIN) bn_operand_preamble(rbn, C, bn_lds + 2 * C, blockIdx.x == 0);      // (its barrier is harmless here)
  for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {
    const BnChan k = bn_chan(sums, run_mean, run_var, C, ch, inv_count, eps);
    float scale, shift;
    bn_scale_shift(k.mean, k.invstd, gamma[ch], beta[ch], scale, shift);
    bn_lds[ch] = scale;
    bn_lds[C + ch] = shift;
    if (sums && blockIdx.x == 0) {
      mean_out[ch] = k.mean;
      invstd_out[ch] = k.invstd;
      if (run_mean) {
        const double mu = rep_sum(sums, C, ch) * inv_count;
        double var = rep_sum(sums, C, C + ch) * inv_count - mu * mu;
        if (var < 0.0) var = 0.0;
        run_mean[ch] = momentum * run_mean[ch] + (1.f - momentum) * (float)mu;
        run_var[ch] = momentum * run_var[ch] + (1.f - momentum) * (float)var;
      }
    }
  }
  __syncthreads();
  float sc[4], sh[4], rsc[JOIN ? 4 : 1], rsh[JOIN ? 4 : 1];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sc[e] = bn_lds[c + e]; sh[e] = bn_lds[C + c + e];
    if constexpr (JOIN) { rsc[e] = bn_lds[2 * C + c + e]; rsh[e] = bn_lds[3 * C + c + e]; }
  }
  while (r < M) {
    const long long rn = r + rstep;
    float vn[4], wn[4], o[4];
    if (rn < M) {
      Vec4<T>::load(x + rn * ldx + c, vn);
      if (res) Vec4<T>::load(res + rn * ldres + c, wn);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fmaf(v[e], sc[e], sh[e]);
    if (res) {
      if constexpr (JOIN) {
        // the separate path stores the normalised shortcut in T before adding it: round the same way so both paths give the same bits
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += to_f32(from_f32<T>(fmaf(w[e], rsc[e], rsh[e])));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += w[e];
      }
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
    }
    Vec4<T>::store(y + r * ldy + c, o);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = vn[e]; w[e] = wn[e]; }
    r = rn;
  }
}

// dx = gamma*invstd*(dy' - sum_dy/count - xhat*sum_dyxhat/count); optional dres = dy' (gradient of the fused residual);
// block 0 accumulates dgamma += sum dy'*xhat, dbeta += sum dy' (from this rank's `lsums` when given: SyncBN).
// MX: the ReLU mask is re-derived from x (mbeta; bn_operand.hpp) -- a separate instantiation: with the mask parameters in the common
// kernel it went from 64 to 68 registers (8 -> 7 waves per SIMD) and from 8.3 to 9.8 us per launch x 75 launches
template <class T, bool MX = false>
__global__ __launch_bounds__(512) void bn_bwd_dx_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dy, int lddy,
                                                        const T* __restrict__ y, int ldy, T* __restrict__ dx, int lddx,
                                                        T* __restrict__ dres, int lddres, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                        const double* __restrict__ sums, const double* __restrict__ lsums,
                                                        double inv_count, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                        long long M, int C, int rows_per_pass, const float* __restrict__ beta_y, int sums_vs_x,
                                                        const float* __restrict__ mbeta) {
  const int quads = C / 4;
  const int c = (threadIdx.x % quads) * 4;
  const int lane_row = threadIdx.x / quads;
  // as in bn_apply_kernel: first row fetched before the preamble, later rows one iteration ahead
  const long long rstep = (long long)gridDim.x * rows_per_pass;
  long long r = (long long)blockIdx.x * rows_per_pass + lane_row;
  float v[4], g[4], yy[4];
  if (r < M) {
    Vec4<T>::load(x + r * ldx + c, v);
    Vec4<T>::load(dy + r * lddy + c, g);
    if (y) Vec4<T>::load(y + r * ldy + c, yy);
  }
  extern __shared__ float bn_lds[];          // [2][C]: sum_dy/count, sum_dyxhat/count
  for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {
    const double t0 = rep_sum(sums, C, ch);
    double t1 = rep_sum(sums, C, C + ch);
    if (beta_y) {      // the sums are (sum dy', sum dy' * y) from the consumer's dgrad epilogue: xhat = (y - beta) / gamma where y > 0
      const double gm = (double)gamma[ch];
      t1 = gm != 0.0 ? (t1 - (double)beta_y[ch] * t0) / gm : 0.0;
    }
    if (sums_vs_x) t1 = (t1 - (double)mean[ch] * t0) * (double)invstd[ch];      // (sum dy', sum dy' * x): xhat = (x - mean) * invstd
    bn_lds[ch] = (float)(t0 * inv_count);
    bn_lds[C + ch] = (float)(t1 * inv_count);
    if (blockIdx.x == 0) {
      if (dbeta) dbeta[ch] += (float)(lsums ? rep_sum(lsums, C, ch) : t0);
      if (dgamma) dgamma[ch] += (float)(lsums ? rep_sum(lsums, C, C + ch) : t1);
    }
  }
  __syncthreads();
  float mu[4], is[4], k0[4], k1[4], gi[4], msc[MX ? 4 : 1], msh[MX ? 4 : 1];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    mu[e] = mean[c + e];
    is[e] = invstd[c + e];
    gi[e] = gamma[c + e] * is[e];
    k0[e] = bn_lds[c + e];
    k1[e] = bn_lds[C + c + e];
    if constexpr (MX) bn_scale_shift(mu[e], is[e], gamma[c + e], mbeta[c + e], msc[e], msh[e]);      // ReLU mask re-derived from x (bn_operand.hpp)
  }
  while (r < M) {
    const long long rn = r + rstep;
    float vn[4], gn[4], yn[4], o[4];
    if (rn < M) {
      Vec4<T>::load(x + rn * ldx + c, vn);
      Vec4<T>::load(dy + rn * lddy + c, gn);
      if (y) Vec4<T>::load(y + rn * ldy + c, yn);
    }
    if (y) {
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = yy[e] > 0.f ? g[e] : 0.f;
    }
    if constexpr (MX) {
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = fmaf(v[e], msc[e], msh[e]) > 0.f ? g[e] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = gi[e] * (g[e] - k0[e] - (v[e] - mu[e]) * is[e] * k1[e]);
    Vec4<T>::store(dx + r * lddx + c, o);
    if (dres) Vec4<T>::store(dres + r * lddres + c, g);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = vn[e]; g[e] = gn[e]; yy[e] = yn[e]; }
    r = rn;
  }
}

// eval-mode BN backward is never needed (no training in eval); not provided.

// ------------------------------------------------------------------------------------------------
// GroupNorm (+ exact-erf GELU) (+ residual):  out = act(GN(x)) + res.   One 1024-thread block per image:
// TX = C/4 channel quads x TY pixel lanes, two passes over that image's [HW][C] slab (second pass is L2-hot).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// GroupNorm is split over many blocks per image (a level map is only 8 images): pass 1 accumulates fp64 group sums with
// atomics, pass 2 is a fully parallel elementwise apply.  Threads own a fixed channel quad; 256 threads = (C/4) x lanes.
template <class T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, int ldx, long long x_bs, double* __restrict__ sums,
                                                       int HW, int C, int G, int pix_per_block) {
  __shared__ float red[256 * 2];
  const int quads = C / 4, lanes = 256 / quads;
  const int q = threadIdx.x % quads, ty = threadIdx.x / quads;
  const int n = blockIdx.y, c = q * 4, cpg = C / G;
  const int p0 = blockIdx.x * pix_per_block;
  int p1 = p0 + pix_per_block;
  if (p1 > HW) p1 = HW;
  const T* xp = x + (long long)n * x_bs;
  float s0 = 0.f, s1 = 0.f;
  for (int p = p0 + ty; p < p1; p += lanes) {
    float v[4];
    Vec4<T>::load(xp + (long long)p * ldx + c, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) { s0 += v[e]; s1 = fmaf(v[e], v[e], s1); }
  }
  red[threadIdx.x * 2] = s0;
  red[threadIdx.x * 2 + 1] = s1;
  __syncthreads();
  if ((int)threadIdx.x < G) {
    const int g = threadIdx.x;
    double a = 0.0, b = 0.0;
    const int q0 = g * cpg / 4, q1 = (g + 1) * cpg / 4;
    for (int t = 0; t < lanes; ++t)
      for (int qq = q0; qq < q1; ++qq) { a += red[(t * quads + qq) * 2]; b += red[(t * quads + qq) * 2 + 1]; }
    atomicAdd(sums + ((long long)n * G + g) * 2, a);
    atomicAdd(sums + ((long long)n * G + g) * 2 + 1, b);
  }
}

template <class T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, int ldx, long long x_bs, const T* __restrict__ res,
                                                       int ldres, long long res_bs, T* __restrict__ out, int ldout, long long out_bs,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const double* __restrict__ sums, float* __restrict__ mean_out,
                                                       float* __restrict__ rstd_out, int HW, int C, int G, float eps, int gelu,
                                                       int pix_per_block) {
  const int quads = C / 4, lanes = 256 / quads;
  const int q = threadIdx.x % quads, ty = threadIdx.x / quads;
  const int n = blockIdx.y, c = q * 4, cpg = C / G, g = c / cpg;
  const double cnt = (double)HW * cpg;
  const double mu_d = sums[((long long)n * G + g) * 2] / cnt;
  double var = sums[((long long)n * G + g) * 2 + 1] / cnt - mu_d * mu_d;
  if (var < 0.0) var = 0.0;
  const float mu = (float)mu_d, rs = (float)(1.0 / sqrt(var + (double)eps));
  if (mean_out && blockIdx.x == 0 && ty == 0 && (c % cpg) == 0) { mean_out[n * G + g] = mu; rstd_out[n * G + g] = rs; }
  float sc[4], sh[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { sc[e] = rs * gamma[c + e]; sh[e] = beta[c + e] - mu * sc[e]; }
  const int p0 = blockIdx.x * pix_per_block;
  int p1 = p0 + pix_per_block;
  if (p1 > HW) p1 = HW;
  const T* xp = x + (long long)n * x_bs;
  for (int p = p0 + ty; p < p1; p += lanes) {
    float v[4], o[4];
    Vec4<T>::load(xp + (long long)p * ldx + c, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float u = fmaf(v[e], sc[e], sh[e]);
      o[e] = gelu ? gelu_f(u) : u;
    }
    if (res) {
      float w[4];
      Vec4<T>::load(res + (long long)n * res_bs + (long long)p * ldres + c, w);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] += w[e];
    }
    Vec4<T>::store(out + (long long)n * out_bs + (long long)p * ldout + c, o);
  }
}

// backward pass 1: chs[n][c][2] (fp64, pre-zeroed) += per-channel (sum dy', sum dy'*xhat) over the block's pixels
template <class T>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const T* __restrict__ x, int ldx, long long x_bs, const T* __restrict__ dy,
                                                            int lddy, long long dy_bs, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, double* __restrict__ chs, int HW, int C,
                                                            int G, int gelu, int pix_per_block) {
  __shared__ float red[256 * 8];
  const int quads = C / 4, lanes = 256 / quads;
  const int q = threadIdx.x % quads, ty = threadIdx.x / quads;
  const int n = blockIdx.y, c = q * 4, cpg = C / G, g = c / cpg;
  const float mu = mean[n * G + g], rs = rstd[n * G + g];
  float ga[4], be[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { ga[e] = gamma[c + e]; be[e] = beta[c + e]; }
  const int p0 = blockIdx.x * pix_per_block;
  int p1 = p0 + pix_per_block;
  if (p1 > HW) p1 = HW;
  const T* xp = x + (long long)n * x_bs;
  const T* gp = dy + (long long)n * dy_bs;
  float s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  for (int p = p0 + ty; p < p1; p += lanes) {
    float v[4], d[4];
    Vec4<T>::load(xp + (long long)p * ldx + c, v);
    Vec4<T>::load(gp + (long long)p * lddy + c, d);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (v[e] - mu) * rs;
      const float dd = gelu ? d[e] * gelu_grad_f(xh * ga[e] + be[e]) : d[e];
      s0[e] += dd;
      s1[e] = fmaf(dd, xh, s1[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[threadIdx.x * 8 + e] = s0[e]; red[threadIdx.x * 8 + 4 + e] = s1[e]; }
  __syncthreads();
  if (ty == 0) {
    for (int t = 1; t < lanes; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) { s0[e] += red[(t * quads + q) * 8 + e]; s1[e] += red[(t * quads + q) * 8 + 4 + e]; }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      atomicAdd(chs + ((long long)n * C + c + e) * 2, (double)s0[e]);
      atomicAdd(chs + ((long long)n * C + c + e) * 2 + 1, (double)s1[e]);
    }
  }
}

// backward pass 2: dx = rstd*(gamma*dy' - A_g - xhat*B_g), A_g/B_g = group means of gamma*(per-channel sums); the
// pixel-chunk-0 block of every image adds that image's per-channel sums into dgamma / dbeta.
template <class T>
__global__ __launch_bounds__(256) void gn_bwd_dx_kernel(const T* __restrict__ x, int ldx, long long x_bs, const T* __restrict__ dy,
                                                        int lddy, long long dy_bs, T* __restrict__ dx, int lddx, long long dx_bs,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const double* __restrict__ chs, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta, int HW, int C, int G, int gelu, int pix_per_block) {
  extern __shared__ float gsm[];      // [G][2]
  const int quads = C / 4, lanes = 256 / quads;
  const int q = threadIdx.x % quads, ty = threadIdx.x / quads;
  const int n = blockIdx.y, c = q * 4, cpg = C / G, g = c / cpg;
  if ((int)threadIdx.x < G) {
    const int gg = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int cc = gg * cpg; cc < (gg + 1) * cpg; ++cc) {
      a += (double)gamma[cc] * chs[((long long)n * C + cc) * 2];
      b += (double)gamma[cc] * chs[((long long)n * C + cc) * 2 + 1];
    }
    const double inv = 1.0 / ((double)HW * cpg);
    gsm[gg * 2] = (float)(a * inv);
    gsm[gg * 2 + 1] = (float)(b * inv);
  }
  if (blockIdx.x == 0) {
    for (int cc = threadIdx.x; cc < C; cc += blockDim.x) {
      if (dbeta) atomicAdd(dbeta + cc, (float)chs[((long long)n * C + cc) * 2]);
      if (dgamma) atomicAdd(dgamma + cc, (float)chs[((long long)n * C + cc) * 2 + 1]);
    }
  }
  __syncthreads();
  const float mu = mean[n * G + g], rs = rstd[n * G + g];
  const float A = gsm[g * 2], Bq = gsm[g * 2 + 1];
  float ga[4], be[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { ga[e] = gamma[c + e]; be[e] = beta[c + e]; }
  const int p0 = blockIdx.x * pix_per_block;
  int p1 = p0 + pix_per_block;
  if (p1 > HW) p1 = HW;
  const T* xp = x + (long long)n * x_bs;
  const T* gp = dy + (long long)n * dy_bs;
  for (int p = p0 + ty; p < p1; p += lanes) {
    float v[4], d[4], o[4];
    Vec4<T>::load(xp + (long long)p * ldx + c, v);
    Vec4<T>::load(gp + (long long)p * lddy + c, d);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (v[e] - mu) * rs;
      const float dd = gelu ? d[e] * gelu_grad_f(xh * ga[e] + be[e]) : d[e];
      o[e] = rs * (ga[e] * dd - A - xh * Bq);
    }
    Vec4<T>::store(dx + (long long)n * dx_bs + (long long)p * lddx + c, o);
  }
}

// ---- single-launch GroupNorm for small maps: one block per (image, group) --------------------------------------------
// The split kernels above need two launches each way; at the EMRT sizes (<= 32x32 pixels per level, 8 channels per
// group) a whole (image, group) slice is 16 KB, so one block can take it in two passes over L1/L2-hot data and the layer
// costs one launch.  Threads = QG channel quads x (256 / QG) pixel lanes.
template <class T>
__global__ __launch_bounds__(256) void gn_fused_fwd_kernel(const T* __restrict__ x, int ldx, long long x_bs, const T* __restrict__ res,
                                                           int ldres, long long res_bs, T* __restrict__ out, int ldout, long long out_bs,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ mean_out, float* __restrict__ rstd_out, int HW, int C, int G,
                                                           float eps, int gelu) {
  __shared__ double red[2][4];
  const int cpg = C / G, qg = cpg / 4, lanes = 256 / qg;
  const int n = blockIdx.x / G, g = blockIdx.x % G;
  const int q = threadIdx.x % qg, ty = threadIdx.x / qg;
  const int c = g * cpg + q * 4;
  const T* xp = x + (long long)n * x_bs + c;
  float s0 = 0.f, s1 = 0.f;
  if (ty < lanes)
    for (int p = ty; p < HW; p += lanes) {
      float v[4];
      Vec4<T>::load(xp + (long long)p * ldx, v);
#pragma unroll
      for (int e = 0; e < 4; ++e) { s0 += v[e]; s1 = fmaf(v[e], v[e], s1); }
    }
  double a = (double)wave_sum(s0), b = (double)wave_sum(s1);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  const double cnt = (double)HW * cpg;
  const double mu_d = a / cnt;
  double var = b / cnt - mu_d * mu_d;
  if (var < 0.0) var = 0.0;
  const float mu = (float)mu_d, rs = (float)(1.0 / sqrt(var + (double)eps));
  if (mean_out && threadIdx.x == 0) { mean_out[n * G + g] = mu; rstd_out[n * G + g] = rs; }
  if (ty >= lanes) return;
  float sc[4], sh[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { sc[e] = rs * gamma[c + e]; sh[e] = beta[c + e] - mu * sc[e]; }
  for (int p = ty; p < HW; p += lanes) {
    float v[4], o[4];
    Vec4<T>::load(xp + (long long)p * ldx, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float u = fmaf(v[e], sc[e], sh[e]);
      o[e] = gelu ? gelu_f(u) : u;
    }
    if (res) {
      float w[4];
      Vec4<T>::load(res + (long long)n * res_bs + (long long)p * ldres + c, w);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] += w[e];
    }
    Vec4<T>::store(out + (long long)n * out_bs + (long long)p * ldout + c, o);
  }
}

template <class T>
__global__ __launch_bounds__(256) void gn_fused_bwd_kernel(const T* __restrict__ x, int ldx, long long x_bs, const T* __restrict__ dy,
                                                           int lddy, long long dy_bs, T* __restrict__ dx, int lddx, long long dx_bs,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int HW, int C, int G, int gelu) {
  __shared__ float chs[256 * 8];       // per-thread (sum dy', sum dy'*xhat) for its 4 channels
  __shared__ float grp[2];
  const int cpg = C / G, qg = cpg / 4, lanes = 256 / qg;
  const int n = blockIdx.x / G, g = blockIdx.x % G;
  const int q = threadIdx.x % qg, ty = threadIdx.x / qg;
  const int c = g * cpg + q * 4;
  const float mu = mean[n * G + g], rs = rstd[n * G + g];
  float ga[4], be[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { ga[e] = gamma[c + e]; be[e] = beta[c + e]; }
  const T* xp = x + (long long)n * x_bs + c;
  const T* gp = dy + (long long)n * dy_bs + c;
  float s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  if (ty < lanes)
    for (int p = ty; p < HW; p += lanes) {
      float v[4], d[4];
      Vec4<T>::load(xp + (long long)p * ldx, v);
      Vec4<T>::load(gp + (long long)p * lddy, d);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (v[e] - mu) * rs;
        const float dd = gelu ? d[e] * gelu_grad_f(xh * ga[e] + be[e]) : d[e];
        s0[e] += dd;
        s1[e] = fmaf(dd, xh, s1[e]);
      }
    }
#pragma unroll
  for (int e = 0; e < 4; ++e) { chs[threadIdx.x * 8 + e] = s0[e]; chs[threadIdx.x * 8 + 4 + e] = s1[e]; }
  __syncthreads();
  __shared__ float gsum[256 * 2];      // gamma-weighted channel sums (separate array: the columns of chs are still being read)
  if ((int)threadIdx.x < cpg) {        // channel cc of the group: column sums over the pixel lanes, in fp64
    const int cc = threadIdx.x, qq = cc / 4, e = cc % 4;
    double a = 0.0, b = 0.0;
    for (int t = 0; t < lanes; ++t) { a += chs[(t * qg + qq) * 8 + e]; b += chs[(t * qg + qq) * 8 + 4 + e]; }
    if (dbeta) atomicAdd(dbeta + g * cpg + cc, (float)a);
    if (dgamma) atomicAdd(dgamma + g * cpg + cc, (float)b);
    gsum[cc * 2] = (float)((double)gamma[g * cpg + cc] * a);
    gsum[cc * 2 + 1] = (float)((double)gamma[g * cpg + cc] * b);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int cc = 0; cc < cpg; ++cc) { a += gsum[cc * 2]; b += gsum[cc * 2 + 1]; }
    const double inv = 1.0 / ((double)HW * cpg);
    grp[0] = (float)(a * inv);
    grp[1] = (float)(b * inv);
  }
  __syncthreads();
  if (ty >= lanes) return;
  const float A = grp[0], Bq = grp[1];
  for (int p = ty; p < HW; p += lanes) {
    float v[4], d[4], o[4];
    Vec4<T>::load(xp + (long long)p * ldx, v);
    Vec4<T>::load(gp + (long long)p * lddy, d);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (v[e] - mu) * rs;
      const float dd = gelu ? d[e] * gelu_grad_f(xh * ga[e] + be[e]) : d[e];
      o[e] = rs * (ga[e] * dd - A - xh * Bq);
    }
    Vec4<T>::store(dx + (long long)n * dx_bs + (long long)p * lddx + c, o);
  }
}

// ---- GroupNorm over several levels of one token tensor in ONE launch -------------------------------------------------
// x / res / out / dy / dx are [N][Lv][C]-shaped token tensors (row stride ld, batch stride bs); level l owns the rows
// [start[l], start[l] + hw[l]) and has its own gamma / beta (and gradient) vectors.  One block per (level, image, group),
// same two-pass body as the single-map kernels above.
#define GN_MAX_LEVELS 4
struct GnLevels {
  int L;
  int start[GN_MAX_LEVELS], hw[GN_MAX_LEVELS];
  const float* gamma[GN_MAX_LEVELS];
  const float* beta[GN_MAX_LEVELS];
  float* dgamma[GN_MAX_LEVELS];
  float* dbeta[GN_MAX_LEVELS];
};

template <class T>
__global__ __launch_bounds__(256) void gn_levels_fwd_kernel(const T* __restrict__ x, int ldx, long long x_bs, const T* __restrict__ res,
                                                            int ldres, long long res_bs, T* __restrict__ out, int ldout, long long out_bs,
                                                            GnLevels lv, float* __restrict__ mean_out, float* __restrict__ rstd_out, int N,
                                                            int C, int G, float eps, int gelu) {
  __shared__ double red[2][4];
  const int per_level = N * G;
  const int l = blockIdx.x / per_level, rem = blockIdx.x % per_level;
  const int n = rem / G, g = rem % G;
  int HW = lv.hw[0], row0 = lv.start[0];
  const float* gam = lv.gamma[0];
  const float* bet = lv.beta[0];
#pragma unroll
  for (int k = 1; k < GN_MAX_LEVELS; ++k)
    if (l == k) { HW = lv.hw[k]; row0 = lv.start[k]; gam = lv.gamma[k]; bet = lv.beta[k]; }
  const int cpg = C / G, qg = cpg / 4, lanes = 256 / qg;
  const int q = threadIdx.x % qg, ty = threadIdx.x / qg;
  const int c = g * cpg + q * 4;
  const T* xp = x + (long long)n * x_bs + (long long)row0 * ldx + c;
  float s0 = 0.f, s1 = 0.f;
  for (int p = ty; p < HW; p += lanes) {
    float v[4];
    Vec4<T>::load(xp + (long long)p * ldx, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) { s0 += v[e]; s1 = fmaf(v[e], v[e], s1); }
  }
  double a = (double)wave_sum(s0), b = (double)wave_sum(s1);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  const double cnt = (double)HW * cpg;
  const double mu_d = a / cnt;
  double var = b / cnt - mu_d * mu_d;
  if (var < 0.0) var = 0.0;
  const float mu = (float)mu_d, rs = (float)(1.0 / sqrt(var + (double)eps));
  if (mean_out && threadIdx.x == 0) { mean_out[blockIdx.x] = mu; rstd_out[blockIdx.x] = rs; }      // [level][n][g]
  float sc[4], sh[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { sc[e] = rs * gam[c + e]; sh[e] = bet[c + e] - mu * sc[e]; }
  for (int p = ty; p < HW; p += lanes) {
    float v[4], o[4];
    Vec4<T>::load(xp + (long long)p * ldx, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float u = fmaf(v[e], sc[e], sh[e]);
      o[e] = gelu ? gelu_f(u) : u;
    }
    if (res) {
      float w[4];
      Vec4<T>::load(res + (long long)n * res_bs + (long long)(row0 + p) * ldres + c, w);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] += w[e];
    }
    Vec4<T>::store(out + (long long)n * out_bs + (long long)(row0 + p) * ldout + c, o);
  }
}

template <class T>
__global__ __launch_bounds__(256) void gn_levels_bwd_kernel(const T* __restrict__ x, int ldx, long long x_bs, const T* __restrict__ dy,
                                                            int lddy, long long dy_bs, T* __restrict__ dx, int lddx, long long dx_bs,
                                                            GnLevels lv, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            int N, int C, int G, int gelu) {
  __shared__ float chs[256 * 8];
  __shared__ float gsum[256 * 2];
  __shared__ float grp[2];
  const int per_level = N * G;
  const int l = blockIdx.x / per_level, rem = blockIdx.x % per_level;
  const int n = rem / G, g = rem % G;
  int HW = lv.hw[0], row0 = lv.start[0];
  const float* gam = lv.gamma[0];
  const float* bet = lv.beta[0];
  float* dgam = lv.dgamma[0];
  float* dbet = lv.dbeta[0];
#pragma unroll
  for (int k = 1; k < GN_MAX_LEVELS; ++k)
    if (l == k) { HW = lv.hw[k]; row0 = lv.start[k]; gam = lv.gamma[k]; bet = lv.beta[k]; dgam = lv.dgamma[k]; dbet = lv.dbeta[k]; }
  const int cpg = C / G, qg = cpg / 4, lanes = 256 / qg;
  const int q = threadIdx.x % qg, ty = threadIdx.x / qg;
  const int c = g * cpg + q * 4;
  const float mu = mean[blockIdx.x], rs = rstd[blockIdx.x];
  float ga[4], be[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { ga[e] = gam[c + e]; be[e] = bet[c + e]; }
  const T* xp = x + (long long)n * x_bs + (long long)row0 * ldx + c;
  const T* gp = dy + (long long)n * dy_bs + (long long)row0 * lddy + c;
  float s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  for (int p = ty; p < HW; p += lanes) {
    float v[4], d[4];
    Vec4<T>::load(xp + (long long)p * ldx, v);
    Vec4<T>::load(gp + (long long)p * lddy, d);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (v[e] - mu) * rs;
      const float dd = gelu ? d[e] * gelu_grad_f(xh * ga[e] + be[e]) : d[e];
      s0[e] += dd;
      s1[e] = fmaf(dd, xh, s1[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { chs[threadIdx.x * 8 + e] = s0[e]; chs[threadIdx.x * 8 + 4 + e] = s1[e]; }
  __syncthreads();
  if ((int)threadIdx.x < cpg) {
    const int cc = threadIdx.x, qq = cc / 4, e = cc % 4;
    double a = 0.0, b = 0.0;
    for (int t = 0; t < lanes; ++t) { a += chs[(t * qg + qq) * 8 + e]; b += chs[(t * qg + qq) * 8 + 4 + e]; }
    if (dbet) atomicAdd(dbet + g * cpg + cc, (float)a);
    if (dgam) atomicAdd(dgam + g * cpg + cc, (float)b);
    gsum[cc * 2] = (float)((double)gam[g * cpg + cc] * a);
    gsum[cc * 2 + 1] = (float)((double)gam[g * cpg + cc] * b);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int cc = 0; cc < cpg; ++cc) { a += gsum[cc * 2]; b += gsum[cc * 2 + 1]; }
    const double inv = 1.0 / ((double)HW * cpg);
    grp[0] = (float)(a * inv);
    grp[1] = (float)(b * inv);
  }
  __syncthreads();
  const float A = grp[0], Bq = grp[1];
  for (int p = ty; p < HW; p += lanes) {
    float v[4], d[4], o[4];
    Vec4<T>::load(xp + (long long)p * ldx, v);
    Vec4<T>::load(gp + (long long)p * lddy, d);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (v[e] - mu) * rs;
      const float dd = gelu ? d[e] * gelu_grad_f(xh * ga[e] + be[e]) : d[e];
      o[e] = rs * (ga[e] * dd - A - xh * Bq);
    }
    Vec4<T>::store(dx + (long long)n * dx_bs + (long long)(row0 + p) * lddx + c, o);
  }
}

// ---- row-major multi-level GroupNorm (8 channels per group: the EMRT case, GroupNorm(32, 256)) ------------------------------------
// The one-block-per-(image, group) kernels above read 16 bytes out of every 512-byte token row: 2 KB in flight per block and one
// dependent load per iteration (44 us at B = 16 for 33 MB of traffic).  Here a thread owns one GROUP of one row (8 channels = one
// 16-byte access for bf16 / fp16), 256 / G rows per pass, so a block reads whole rows; the statistics become a separate launch:
//   stats:  block = (image, level, chunk of rows) -> per-group partial sums, fp64 atomics into ws[(level, image, group)][2]
//   apply:  every thread re-derives mean / rstd of its group from ws (the block of chunk 0 records them for the backward)
// and the same split backward (sums of dy*gamma, dy*gamma*xhat per group + dgamma / dbeta per channel, then dx).
struct GnRows {
  GnLevels lv;
  int rows_per_block;
  int nblk[GN_MAX_LEVELS];       // blocks per image of each level
  int per_image;
};
__device__ __forceinline__ void gn_rows_locate(const GnRows& g, int& n, int& l, int& r0, int& r1, int& HW, int& row0) {
  n = blockIdx.x / g.per_image;
  int b = blockIdx.x % g.per_image;
  l = 0;
#pragma unroll
  for (int k = 0; k < GN_MAX_LEVELS - 1; ++k)
    if (l == k && b >= g.nblk[k]) { b -= g.nblk[k]; l = k + 1; }
  HW = g.lv.hw[0]; row0 = g.lv.start[0];
#pragma unroll
  for (int k = 1; k < GN_MAX_LEVELS; ++k)
    if (l == k) { HW = g.lv.hw[k]; row0 = g.lv.start[k]; }
  r0 = b * g.rows_per_block;
  r1 = r0 + g.rows_per_block;
  if (r1 > HW) r1 = HW;
}
template <int K>
__device__ __forceinline__ const float* gn_pick(const float* const (&a)[GN_MAX_LEVELS], int l) {
  const float* p = a[0];
#pragma unroll
  for (int k = 1; k < GN_MAX_LEVELS; ++k)
    if (l == k) p = a[k];
  return p;
}

template <class T>
__global__ __launch_bounds__(256) void gn_rows_stats_kernel(const T* __restrict__ x, int ldx, long long x_bs, GnRows gr, double* __restrict__ ws,
                                                            int N, int G) {
  __shared__ float red[256][2];
  int n, l, r0, r1, HW, row0;
  gn_rows_locate(gr, n, l, r0, r1, HW, row0);
  const int g = threadIdx.x % G, rl = threadIdx.x / G, nrl = 256 / G;
  const T* xp = x + (long long)n * x_bs + (long long)row0 * ldx + g * 8;
  float s0 = 0.f, s1 = 0.f;
  for (int p = r0 + rl; p < r1; p += nrl) {
    float v[8];
    Vec8<T>::load(xp + (long long)p * ldx, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) { s0 += v[e]; s1 = fmaf(v[e], v[e], s1); }
  }
  red[threadIdx.x][0] = s0; red[threadIdx.x][1] = s1;
  __syncthreads();
  if ((int)threadIdx.x < 2 * G) {
    const int gg = threadIdx.x >> 1, w = threadIdx.x & 1;
    double a = 0.0;
    for (int t = 0; t < nrl; ++t) a += (double)red[t * G + gg][w];
    atomicAdd(&ws[(((long long)l * N + n) * G + gg) * 2 + w], a);
  }
}

template <class T>
__global__ __launch_bounds__(256) void gn_rows_apply_kernel(const T* __restrict__ x, int ldx, long long x_bs, const T* __restrict__ res, int ldres,
                                                            long long res_bs, T* __restrict__ out, int ldout, long long out_bs, GnRows gr,
                                                            const double* __restrict__ ws, float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out, int N, int G, float eps, int gelu) {
  int n, l, r0, r1, HW, row0;
  gn_rows_locate(gr, n, l, r0, r1, HW, row0);
  const int g = threadIdx.x % G, rl = threadIdx.x / G, nrl = 256 / G;
  const long long sidx = ((long long)l * N + n) * G + g;
  const double cnt = (double)HW * 8.0;
  const double mu_d = ws[sidx * 2] / cnt;
  double var = ws[sidx * 2 + 1] / cnt - mu_d * mu_d;
  if (var < 0.0) var = 0.0;
  const float mu = (float)mu_d, rs = (float)(1.0 / sqrt(var + (double)eps));
  if (mean_out && r0 == 0 && rl == 0) { mean_out[sidx] = mu; rstd_out[sidx] = rs; }      // [level][n][g]
  const float* gam = gn_pick<0>(gr.lv.gamma, l) + g * 8;
  const float* bet = gn_pick<0>(gr.lv.beta, l) + g * 8;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc[e] = rs * gam[e]; sh[e] = bet[e] - mu * sc[e]; }
  const T* xp = x + (long long)n * x_bs + (long long)row0 * ldx + g * 8;
  for (int p = r0 + rl; p < r1; p += nrl) {
    float v[8], o[8];
    Vec8<T>::load(xp + (long long)p * ldx, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float u = fmaf(v[e], sc[e], sh[e]);
      o[e] = gelu ? gelu_f(u) : u;
    }
    if (res) {
      float w[8];
      Vec8<T>::load(res + (long long)n * res_bs + (long long)(row0 + p) * ldres + g * 8, w);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += w[e];
    }
    Vec8<T>::store(out + (long long)n * out_bs + (long long)(row0 + p) * ldout + g * 8, o);
  }
}

template <class T>
__global__ __launch_bounds__(256) void gn_rows_bwd_stats_kernel(const T* __restrict__ x, int ldx, long long x_bs, const T* __restrict__ dy, int lddy,
                                                                long long dy_bs, GnRows gr, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, double* __restrict__ ws, int N, int G, int gelu) {
  __shared__ float chs[256][17];        // [thread][8 x (sum dd, sum dd*xhat)], padded against bank conflicts in the column read below
  int n, l, r0, r1, HW, row0;
  gn_rows_locate(gr, n, l, r0, r1, HW, row0);
  const int g = threadIdx.x % G, rl = threadIdx.x / G, nrl = 256 / G;
  const long long sidx = ((long long)l * N + n) * G + g;
  const float mu = mean[sidx], rs = rstd[sidx];
  const float* gam = gn_pick<0>(gr.lv.gamma, l);
  const float* bet = gn_pick<0>(gr.lv.beta, l);
  float ga[8], be[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { ga[e] = gam[g * 8 + e]; be[e] = bet[g * 8 + e]; }
  const T* xp = x + (long long)n * x_bs + (long long)row0 * ldx + g * 8;
  const T* gp = dy + (long long)n * dy_bs + (long long)row0 * lddy + g * 8;
  float s0[8], s1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s0[e] = 0.f; s1[e] = 0.f; }
  for (int p = r0 + rl; p < r1; p += nrl) {
    float v[8], d[8];
    Vec8<T>::load(xp + (long long)p * ldx, v);
    Vec8<T>::load(gp + (long long)p * lddy, d);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (v[e] - mu) * rs;
      const float dd = gelu ? d[e] * gelu_grad_f(xh * ga[e] + be[e]) : d[e];
      s0[e] += dd;
      s1[e] = fmaf(dd, xh, s1[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { chs[threadIdx.x][e] = s0[e]; chs[threadIdx.x][8 + e] = s1[e]; }
  __syncthreads();
  // thread c = channel c of the row (C = 8 G <= 256): sum over the row lanes, parameter gradients, then the group's two sums
  const int C = G * 8;
  if ((int)threadIdx.x < C) {
    const int c = threadIdx.x, gg = c >> 3, e = c & 7;
    float a = 0.f, b = 0.f;
    for (int t = 0; t < nrl; ++t) { a += chs[t * G + gg][e]; b += chs[t * G + gg][8 + e]; }
    float* dgam = nullptr; float* dbet = nullptr;
#pragma unroll
    for (int k = 0; k < GN_MAX_LEVELS; ++k)
      if (l == k) { dgam = gr.lv.dgamma[k]; dbet = gr.lv.dbeta[k]; }
    if (dbet) atomicAdd(dbet + c, a);
    if (dgam) atomicAdd(dgam + c, b);
    float wa = gam[c] * a, wb = gam[c] * b;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { wa += __shfl_xor(wa, o, 64); wb += __shfl_xor(wb, o, 64); }
    if (e == 0) {
      const long long si = ((long long)l * N + n) * G + gg;
      atomicAdd(&ws[si * 2], (double)wa);
      atomicAdd(&ws[si * 2 + 1], (double)wb);
    }
  }
}

template <class T>
__global__ __launch_bounds__(256) void gn_rows_bwd_dx_kernel(const T* __restrict__ x, int ldx, long long x_bs, const T* __restrict__ dy, int lddy,
                                                             long long dy_bs, T* __restrict__ dx, int lddx, long long dx_bs, GnRows gr,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const double* __restrict__ ws, int N, int G, int gelu) {
  int n, l, r0, r1, HW, row0;
  gn_rows_locate(gr, n, l, r0, r1, HW, row0);
  const int g = threadIdx.x % G, rl = threadIdx.x / G, nrl = 256 / G;
  const long long sidx = ((long long)l * N + n) * G + g;
  const float mu = mean[sidx], rs = rstd[sidx];
  const double inv = 1.0 / ((double)HW * 8.0);
  const float A = (float)(ws[sidx * 2] * inv), Bq = (float)(ws[sidx * 2 + 1] * inv);
  const float* gam = gn_pick<0>(gr.lv.gamma, l) + g * 8;
  const float* bet = gn_pick<0>(gr.lv.beta, l) + g * 8;
  float ga[8], be[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { ga[e] = gam[e]; be[e] = bet[e]; }
  const T* xp = x + (long long)n * x_bs + (long long)row0 * ldx + g * 8;
  const T* gp = dy + (long long)n * dy_bs + (long long)row0 * lddy + g * 8;
  for (int p = r0 + rl; p < r1; p += nrl) {
    float v[8], d[8], o[8];
    Vec8<T>::load(xp + (long long)p * ldx, v);
    Vec8<T>::load(gp + (long long)p * lddy, d);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (v[e] - mu) * rs;
      const float dd = gelu ? d[e] * gelu_grad_f(xh * ga[e] + be[e]) : d[e];
      o[e] = rs * (ga[e] * dd - A - xh * Bq);
    }
    Vec8<T>::store(dx + (long long)n * dx_bs + (long long)(row0 + p) * lddx + g * 8, o);
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm over the last dim C (C % 4 == 0, C <= 1024) with fused residual add:
//   z = a (+ b);  out = LN(z)*gamma + beta (+ post)        one wave per row, 4 rows per 256-thread block.
// `post` is the encoder's "src = src + src_flatten" (transformer_encoder_decoder.py:203), fused here.
// z (the LN input) and mean/rstd are saved for backward.
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ post,
                                                     T* __restrict__ z, T* __restrict__ out, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, long long rows, int C, float eps, float pdrop,
                                                     const unsigned long long* __restrict__ seed, unsigned salt,
                                                     const T* __restrict__ qpos, int qpos_rows, T* __restrict__ q_out) {
  // q_out (optional): out + qpos[row % qpos_rows] -- the NEXT attention's query, with_pos_embed(out, pos) (transformer_encoder_decoder.py:186,283-289),
  // written beside `out` instead of by an add launch that re-reads it (same value: the stored `out`, rounded through T, plus the embedding)
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  // optional inverted dropout on the branch input b (element key = its linear index, as emrt_dropout_fwd mode 0)
  const unsigned long long sd = pdrop > 0.f ? seed[0] : 0ull;
  const float ks = pdrop > 0.f ? 1.f / (1.f - pdrop) : 1.f;
  const int nq = C / 256 + ((C % 256) ? 1 : 0);
  float v[4][4];
  float s = 0.f;
  for (int j = 0; j < nq; ++j) {
    const int c = j * 256 + lane * 4;
    if (c < C) {
      Vec4<T>::load(a + row * C + c, v[j]);
      if (b) {
        float w[4];
        Vec4<T>::load(b + row * C + c, w);
        if (pdrop > 0.f) {
          uint32_t hq[2];
          drop_quad(sd, salt, (unsigned long long)(row * C + c) >> 2, hq);      // (C % 4 == 0: an aligned quad of the flat index)
          const uint32_t thr = drop_thr16(pdrop);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // the dropped value is rounded through T exactly as a separate dropout kernel would store it
            const float d = drop_quad_keep(hq, e, thr) ? w[e] * ks : 0.f;
            w[e] = to_f32(from_f32<T>(d));
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[j][e] += w[e];
      }
      if (z) {
        // round z through T so forward and backward see the same LN input
        Vec4<T>::store(z + row * C + c, v[j]);
        if (sizeof(T) == 2) {      // (in registers: reading the stored value back put a memory round trip on every row's critical path)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[j][e] = to_f32(from_f32<T>(v[j][e]));
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) s += v[j][e];
    }
  }
  const float mu = wave_sum(s) / (float)C;
  float q = 0.f;
  for (int j = 0; j < nq; ++j) {
    const int c = j * 256 + lane * 4;
    if (c < C) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[j][e] - mu; q = fmaf(d, d, q); }
    }
  }
  const float rs = rsqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0 && mean_out) { mean_out[row] = mu; rstd_out[row] = rs; }
  for (int j = 0; j < nq; ++j) {
    const int c = j * 256 + lane * 4;
    if (c < C) {
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mu) * rs * gamma[c + e] + beta[c + e];
      if (post) {
        float w[4];
        Vec4<T>::load(post + row * C + c, w);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += w[e];
      }
      Vec4<T>::store(out + row * C + c, o);
      if (q_out) {
        float w[4];
        Vec4<T>::load(qpos + (row % qpos_rows) * C + c, w);
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] += to_f32(from_f32<T>(o[e]));
        Vec4<T>::store(q_out + row * C + c, w);
      }
    }
  }
}

// LN backward: dz = rstd*(g*dy - mean(g*dy) - xhat*mean(g*dy*xhat));  per-block partial dgamma/dbeta
// -> partial[blk][2][C]; combined by bn_sum_partials_kernel + bn_bwd_finalize_kernel.
template <class T, int R, int NQ, int TPB = 256, bool DY2 = false>
__global__ __launch_bounds__(TPB) void ln_bwd_kernel(const T* __restrict__ z, const T* __restrict__ dy, T* __restrict__ dz,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, float* __restrict__ partial, long long rows,
                                                     int C, int rows_per_block, T* __restrict__ dzb, float pdrop,
                                                     const unsigned long long* __restrict__ seed, unsigned salt,
                                                     float* __restrict__ dgamma_direct, float* __restrict__ dbeta_direct, const T* __restrict__ addend,
                                                     const T* __restrict__ dy2, T* __restrict__ dysum) {
  // dysum (optional, with dy2): dy + dy2 written out -- the gradient of the forward's `post` addend, which saw both
  // dy2 (optional, [rows][C]): a second gradient of the LayerNorm output -- the gradient of the query q_out = out + qpos that the forward wrote beside
  // `out` -- summed with dy as it is loaded (the accumulate launch that used to merge the two is gone)
  // dzb (optional): gradient of the dropped branch input, dz * mask / (1 - p)
  // addend (optional, [rows][C]): another gradient contribution to the residual input, summed into dz here (dzb does not get it)
  const unsigned long long sd = (dzb && pdrop > 0.f) ? seed[0] : 0ull;
  const float ks = pdrop > 0.f ? 1.f / (1.f - pdrop) : 1.f;
  extern __shared__ float sm[];  // [NWV waves][2][C]
  constexpr int NWV = TPB / 64;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nq = C / 256 + ((C % 256) ? 1 : 0);       // <= NQ
  float dg[NQ][4], db[NQ][4];
#pragma unroll
  for (int j = 0; j < NQ; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) { dg[j][e] = 0.f; db[j][e] = 0.f; }
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  long long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  // R rows per wave and iteration (rows wv, wv + 4, ...): the kernel is bound by the latency of the per-row load -> wave reduction
  // -> store chain; R independent chains in one basic block let the scheduler overlap them (R = 4 for C <= 256, the EMRT case)
  for (long long row0 = r0 + wv; row0 < r1; row0 += NWV * R) {
    long long row[R];
    bool ok[R];
    float mu[R], rs[R], s0[R], s1[R];
    float xh[R][NQ][4], gd[R][NQ][4];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      ok[k] = row0 + NWV * k < r1;
      row[k] = ok[k] ? row0 + NWV * k : row0;
      mu[k] = mean[row[k]]; rs[k] = rstd[row[k]];
      s0[k] = 0.f; s1[k] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int c = j * 256 + lane * 4;
      if (j < nq && c < C) {
        float zz[R][4], dd[R][4];
#pragma unroll
        for (int k = 0; k < R; ++k) {
          Vec4<T>::load(z + row[k] * C + c, zz[k]);
          Vec4<T>::load(dy + row[k] * C + c, dd[k]);
        }
        if constexpr (DY2) {      // (its own instantiation: as a run-time branch in this load phase it cost EVERY LayerNorm backward 2-5 us -- 14 more registers
                                  // and the R independent load chains no longer in one basic block; profiles/r6a_timeline_cfg2.txt)
          float d2[R][4];
#pragma unroll
          for (int k = 0; k < R; ++k) Vec4<T>::load(dy2 + row[k] * C + c, d2[k]);
#pragma unroll
          for (int k = 0; k < R; ++k) {
#pragma unroll
            for (int e = 0; e < 4; ++e) dd[k][e] += d2[k][e];
            if (dysum && ok[k]) Vec4<T>::store(dysum + row[k] * C + c, dd[k]);
          }
        }
        float gm[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) gm[e] = gamma[c + e];
#pragma unroll
        for (int k = 0; k < R; ++k) {
          const float wk = ok[k] ? 1.f : 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            xh[k][j][e] = (zz[k][e] - mu[k]) * rs[k];
            gd[k][j][e] = gm[e] * dd[k][e];
            s0[k] += gd[k][j][e];
            s1[k] = fmaf(gd[k][j][e], xh[k][j][e], s1[k]);
            dg[j][e] = fmaf(wk * dd[k][e], xh[k][j][e], dg[j][e]);
            db[j][e] = fmaf(wk, dd[k][e], db[j][e]);
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < R; ++k) { s0[k] = wave_sum(s0[k]) / (float)C; s1[k] = wave_sum(s1[k]) / (float)C; }
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int c = j * 256 + lane * 4;
      if (j < nq && c < C) {
#pragma unroll
        for (int k = 0; k < R; ++k) {
          if (!ok[k]) continue;
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = rs[k] * (gd[k][j][e] - s0[k] - xh[k][j][e] * s1[k]);
          if (addend) {
            float ad[4], oa[4];
            Vec4<T>::load(addend + row[k] * C + c, ad);
#pragma unroll
            for (int e = 0; e < 4; ++e) oa[e] = o[e] + ad[e];
            Vec4<T>::store(dz + row[k] * C + c, oa);
          } else {
            Vec4<T>::store(dz + row[k] * C + c, o);
          }
          if (dzb) {
            if (pdrop > 0.f) {
              uint32_t hq[2];
              drop_quad(sd, salt, (unsigned long long)(row[k] * C + c) >> 2, hq);
              const uint32_t thr = drop_thr16(pdrop);
#pragma unroll
              for (int e = 0; e < 4; ++e)   // the mask is applied to the STORED dz (rounded through T), as a separate mask kernel would read it
                o[e] = drop_quad_keep(hq, e, thr) ? to_f32(from_f32<T>(o[e])) * ks : 0.f;
            }
            Vec4<T>::store(dzb + row[k] * C + c, o);
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int c = j * 256 + lane * 4;
    if (j < nq && c < C) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { sm[(wv * 2 + 0) * C + c + e] = db[j][e]; sm[(wv * 2 + 1) * C + c + e] = dg[j][e]; }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += TPB) {
    float b0 = 0.f, g0 = 0.f;
    for (int w = 0; w < NWV; ++w) { b0 += sm[(w * 2 + 0) * C + c]; g0 += sm[(w * 2 + 1) * C + c]; }
    if (dgamma_direct || dbeta_direct) {
      // <= 512 blocks x 2C fp32 atomics straight into the parameter gradients: they drain while other blocks still run,
      // and the separate finalize launch (a kernel boundary, ~9 us for a few hundred KB) disappears
      if (dbeta_direct) atomicAdd(dbeta_direct + c, b0);
      if (dgamma_direct) atomicAdd(dgamma_direct + c, g0);
    } else {
      partial[(long long)blockIdx.x * 2 * C + c] = b0;       // slot 0: dbeta  (matches bn_bwd_finalize: sums[c] -> dbeta)
      partial[(long long)blockIdx.x * 2 * C + C + c] = g0;   // slot 1: dgamma
    }
  }
}

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
#define DT_SWITCH(dtype, EXPR_F32, EXPR_BF16) \
  do {                                        \
    if ((dtype) == EMRT_F32) { EXPR_F32; } else { EXPR_BF16; } \
  } while (0)
// forward (inference-capable) entry points also take fp16
#define DT_SWITCH3(dtype, EXPR_F32, EXPR_BF16, EXPR_F16) \
  do {                                                   \
    if ((dtype) == EMRT_F32) { EXPR_F32; } else if ((dtype) == EMRT_BF16) { EXPR_BF16; } else { EXPR_F16; } \
  } while (0)

extern "C" size_t emrt_colreduce_workspace_bytes(long long M, int C) {
  int tx, gx, gy;
  col_reduce_geometry(M, C, tx, gx, gy);
  return ((size_t)gx * 2 * C + 2 * (size_t)C) * sizeof(float);
}

// threads / rows-per-pass / grid for the hoisted (fixed channel quad per thread) BN kernels
static inline bool bn_rowgeom(long long M, int C, int& threads, int& rows_per_pass, int& grid) {
  const int quads = C / 4;
  if (quads <= 256) { if (256 % quads) return false; threads = 256; rows_per_pass = 256 / quads; }
  else if (quads <= 512) { threads = quads; rows_per_pass = 1; }          // C = 2048 -> 512 threads
  else return false;
  // every block first derives the per-channel constants (C x 16 fp64 loads + fp64 math, ~2 us of latency), so it should
  // then stream a decent amount of data: ~32 KB of input per block, between 64 and 1024 blocks
  const long long per_block = (long long)(g_tune.bn_block_kb > 0 ? g_tune.bn_block_kb : 8) * 1024;
  long long g = (M * C * 2 + per_block - 1) / per_block;
  const long long gmax = (M + rows_per_pass - 1) / rows_per_pass;
  if (g < 64) g = 64;
  if (g > 4096) g = 4096;
  if (g > gmax) g = gmax;
  if (g < 1) g = 1;
  grid = (int)g;
  return true;
}

// BatchNorm statistics when the producer could not fuse them: sums[2C] (fp64, PRE-ZEROED by the caller) += (sum x, sum x^2)
extern "C" int emrt_bn_stats(const void* x, int ldx, long long M, int C, double* sums, int dtype, void* stream) {
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  EMRT_REQUIRE(x && sums, "null pointer");
  EMRT_REQUIRE(C % 4 == 0 && ldx % 4 == 0, "C and ld must be multiples of 4");
  int tx, gx, gy;
  col_reduce_geometry(M, C, tx, gx, gy);
  hipStream_t st = (hipStream_t)stream;
  DT_SWITCH(dtype,
            hipLaunchKernelGGL((col_reduce_kernel<float, 0>), dim3(gx, gy), dim3(256), 0, st, (const float*)x, ldx, nullptr, 0, nullptr, 0, nullptr, nullptr, M, C, tx, nullptr, M, 0LL, sums, nullptr),
            hipLaunchKernelGGL((col_reduce_kernel<bf16_t, 0>), dim3(gx, gy), dim3(256), 0, st, (const bf16_t*)x, ldx, nullptr, 0, nullptr, 0, nullptr, nullptr, M, C, tx, nullptr, M, 0LL, sums, nullptr));
  return check_launch("emrt_bn_stats");
}

// y = [relu](BN(x) [+ res]).  sums != null: training -- statistics = sums / count (count may be the global row count after a
// cross-rank all-reduce of sums for SyncBatchNorm); mean/invstd are saved and run_* updated.  sums == null: eval -- run_* are used.
extern "C" int emrt_bn_apply(const void* x, int ldx, const void* res, int ldres, void* y, int ldy, const double* sums, double count,
                             float eps, float momentum, float* mean, float* invstd, float* run_mean, float* run_var,
                             const float* gamma, const float* beta, long long M, int C, int relu, int dtype, void* stream) {
  EMRT_REQUIRE_FWD_DTYPE(dtype);
  EMRT_REQUIRE(x && y && gamma && beta, "null pointer");
  EMRT_REQUIRE(sums ? (mean && invstd) : (run_mean && run_var), "training needs mean/invstd outputs, eval needs running statistics");
  EMRT_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && (!res || ldres % 4 == 0), "C and ld must be multiples of 4");
  int threads, rpp, grid;
  EMRT_REQUIRE(bn_rowgeom(M, C, threads, rpp, grid), "unsupported channel count (C/4 must divide 256, or C <= 2048)");
  hipStream_t st = (hipStream_t)stream;
  const double inv_count = sums ? 1.0 / count : 0.0;
  BnOperand none;
  memset(&none, 0, sizeof(none));
  DT_SWITCH3(dtype,
            hipLaunchKernelGGL((bn_apply_kernel<float>), dim3(grid), dim3(threads), (size_t)2 * C * sizeof(float), st, (const float*)x, ldx, (const float*)res, ldres, (float*)y, ldy, sums, inv_count, eps, momentum, mean, invstd, run_mean, run_var, gamma, beta, M, C, relu, rpp, none),
            hipLaunchKernelGGL((bn_apply_kernel<bf16_t>), dim3(grid), dim3(threads), (size_t)2 * C * sizeof(float), st, (const bf16_t*)x, ldx, (const bf16_t*)res, ldres, (bf16_t*)y, ldy, sums, inv_count, eps, momentum, mean, invstd, run_mean, run_var, gamma, beta, M, C, relu, rpp, none),
            hipLaunchKernelGGL((bn_apply_kernel<f16_t>), dim3(grid), dim3(threads), (size_t)2 * C * sizeof(float), st, (const f16_t*)x, ldx, (const f16_t*)res, ldres, (f16_t*)y, ldy, sums, inv_count, eps, momentum, mean, invstd, run_mean, run_var, gamma, beta, M, C, relu, rpp, none));
  return check_launch("emrt_bn_apply");
}

// out = [relu](BN_train(x) + BN_train(res_raw)): the join at the end of the FIRST block of a ResNet stage, where the shortcut is
// conv1x1 -> BatchNorm (no ReLU; paddle_vision_resnet.py:132-147, 226-233), with the shortcut's BatchNorm applied as its raw conv output is
// loaded.  Both BatchNorms take their arguments as emrt_bn_apply does (r_*: the shortcut's); both save mean / invstd and update their
// running statistics.  The normalised shortcut is rounded to the storage type before the add, so the result equals the two-launch path's.
extern "C" int emrt_bn_apply_join(const void* x, int ldx, const void* res_raw, int ldres, void* y, int ldy, const double* sums, double count,
                                  float eps, float momentum, float* mean, float* invstd, float* run_mean, float* run_var, const float* gamma,
                                  const float* beta, const double* r_sums, double r_count, float r_eps, float r_momentum, float* r_mean,
                                  float* r_invstd, float* r_run_mean, float* r_run_var, const float* r_gamma, const float* r_beta, long long M,
                                  int C, int relu, int dtype, void* stream) {
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  EMRT_REQUIRE(x && res_raw && y && gamma && beta && sums && mean && invstd, "null pointer");
  EMRT_REQUIRE(r_sums && r_mean && r_invstd && r_gamma && r_beta && r_count > 0.0 && count > 0.0 && (r_run_mean != nullptr) == (r_run_var != nullptr),
               "the shortcut's BatchNorm needs sums, mean, invstd, gamma, beta and a positive count");
  EMRT_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldres % 4 == 0, "C and ld must be multiples of 4");
  int threads, rpp, grid;
  EMRT_REQUIRE(bn_rowgeom(M, C, threads, rpp, grid), "unsupported channel count (C/4 must divide 256, or C <= 2048)");
  BnOperand r;
  r.sums = r_sums; r.inv_count = 1.0 / r_count; r.eps = r_eps; r.momentum = r_momentum; r.mean = r_mean; r.invstd = r_invstd;
  r.run_mean = r_run_mean; r.run_var = r_run_var; r.gamma = r_gamma; r.beta = r_beta; r.relu = 0;
  hipStream_t st = (hipStream_t)stream;
  const double inv_count = 1.0 / count;
  const size_t lds = (size_t)4 * C * sizeof(float);
  DT_SWITCH(dtype,
            hipLaunchKernelGGL((bn_apply_kernel<float, true>), dim3(grid), dim3(threads), lds, st, (const float*)x, ldx, (const float*)res_raw, ldres, (float*)y, ldy, sums, inv_count, eps, momentum, mean, invstd, run_mean, run_var, gamma, beta, M, C, relu, rpp, r),
            hipLaunchKernelGGL((bn_apply_kernel<bf16_t, true>), dim3(grid), dim3(threads), lds, st, (const bf16_t*)x, ldx, (const bf16_t*)res_raw, ldres, (bf16_t*)y, ldy, sums, inv_count, eps, momentum, mean, invstd, run_mean, run_var, gamma, beta, M, C, relu, rpp, r));
  return check_launch("emrt_bn_apply_join");
}

// Eval-mode BatchNorm as a per-channel affine map, for every BatchNorm of a model in one launch: block i handles row i of
// desc = [n][7] int64 (gamma, beta offsets into `params`; running mean, variance offsets into `buffers`; C; offset into `out`; offset
// of the producing convolution's own bias in `params` or -1) and writes out[o .. o+C) = s = gamma / sqrt(var + eps),
// out[o+C .. o+2C) = beta + (conv_bias - mean) * s.  The pair is folded into the producing
// convolution's epilogue (emrt_conv2d: out_scale / bias), so inference launches no BatchNorm kernel and writes no pre-BN tensor.
__global__ __launch_bounds__(256) void bn_fold_kernel(const float* __restrict__ params, const float* __restrict__ buffers,
                                                      const long long* __restrict__ desc, float eps, float* __restrict__ out) {
  const long long* d = desc + 7 * (long long)blockIdx.x;
  const float* gamma = params + d[0];
  const float* beta = params + d[1];
  const float* mean = buffers + d[2];
  const float* var = buffers + d[3];
  const int C = (int)d[4];
  float* o = out + d[5];
  const float* cbias = d[6] >= 0 ? params + d[6] : nullptr;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float sc = gamma[c] * rsqrtf(var[c] + eps);
    o[c] = sc;
    o[C + c] = beta[c] + ((cbias ? cbias[c] : 0.f) - mean[c]) * sc;
  }
}
extern "C" int emrt_bn_fold(const float* params, const float* buffers, const long long* desc, int n, float eps, float* out, void* stream) {
  EMRT_REQUIRE(params && buffers && desc && out && n > 0, "null pointer");
  hipLaunchKernelGGL(bn_fold_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, params, buffers, desc, eps, out);
  return check_launch("emrt_bn_fold");
}

// BN backward step 1: sums[2C] (fp64, PRE-ZEROED) += (sum dy', sum dy'*xhat); y (post-ReLU output) may be null when no ReLU was fused.
// mask_gamma / mask_beta (both or neither; y must then be null): the layer's ReLU output was never written (its consumer applied
// BatchNorm + ReLU on load, emrt_bn_resize_bilinear_fwd / emrt_bn_maxpool_fwd): the mask is re-derived from x.
extern "C" int emrt_bn_bwd_reduce(const void* x, int ldx, const void* dy, int lddy, const void* y, int ldy, const float* mean,
                                  const float* invstd, long long M, int C, double* sums, const float* mask_gamma, const float* mask_beta,
                                  int dtype, void* stream) {
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  EMRT_REQUIRE(x && dy && mean && invstd && sums, "null pointer");
  EMRT_REQUIRE((mask_gamma != nullptr) == (mask_beta != nullptr) && !(mask_beta && y), "mask_gamma and mask_beta come together and replace y");
  EMRT_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && lddy % 4 == 0 && (!y || ldy % 4 == 0), "C and ld must be multiples of 4");
  int tx, gx, gy;
  col_reduce_geometry(M, C, tx, gx, gy);
  hipStream_t st = (hipStream_t)stream;
  if (mask_beta) {
    DT_SWITCH(dtype,
              hipLaunchKernelGGL((col_reduce_kernel<float, 3>), dim3(gx, gy), dim3(256), 0, st, (const float*)x, ldx, (const float*)dy, lddy, (const float*)nullptr, 0, mean, invstd, M, C, tx, nullptr, M, 0LL, sums, nullptr, mask_gamma, mask_beta),
              hipLaunchKernelGGL((col_reduce_kernel<bf16_t, 3>), dim3(gx, gy), dim3(256), 0, st, (const bf16_t*)x, ldx, (const bf16_t*)dy, lddy, (const bf16_t*)nullptr, 0, mean, invstd, M, C, tx, nullptr, M, 0LL, sums, nullptr, mask_gamma, mask_beta));
    return check_launch("emrt_bn_bwd_reduce");
  }
  DT_SWITCH(dtype,
            hipLaunchKernelGGL((col_reduce_kernel<float, 1>), dim3(gx, gy), dim3(256), 0, st, (const float*)x, ldx, (const float*)dy, lddy, (const float*)y, ldy, mean, invstd, M, C, tx, nullptr, M, 0LL, sums, nullptr),
            hipLaunchKernelGGL((col_reduce_kernel<bf16_t, 1>), dim3(gx, gy), dim3(256), 0, st, (const bf16_t*)x, ldx, (const bf16_t*)dy, lddy, (const bf16_t*)y, ldy, mean, invstd, M, C, tx, nullptr, M, 0LL, sums, nullptr));
  return check_launch("emrt_bn_bwd_reduce");
}

// BN backward step 2: dx (and optional dres = masked dy); dgamma += , dbeta += from `local_sums` when given (SyncBN: the
// dx formula uses the rank-summed `sums` with the global count, the parameter gradients use this rank's sums) else from `sums`.
// beta_y_moments != null: `sums` hold (sum dy', sum dy' * y) as accumulated by emrt_conv2d(mask_y = y) -- the dgrad of the
// conv that consumes y = relu(BN(x)) -- and are converted with xhat = (y - beta) / gamma; emrt_bn_bwd_reduce is then not needed.
// sums_vs_x != 0: they hold (sum dy', sum dy' * x) (emrt_conv2d_bwd with stat_x = x: the relu(BN(x) + residual) joins).
extern "C" int emrt_bn_bwd_dx(const void* x, int ldx, const void* dy, int lddy, const void* y, int ldy, void* dx, int lddx,
                              void* dres, int lddres, const float* mean, const float* invstd, const float* gamma,
                              const double* sums, const double* local_sums, double count, float* dgamma, float* dbeta, long long M,
                              int C, const float* beta_y_moments, int sums_vs_x, const float* mask_beta, int dtype, void* stream) {
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  EMRT_REQUIRE(x && dy && dx && mean && invstd && gamma && sums, "null pointer");
  EMRT_REQUIRE(!(mask_beta && (y || beta_y_moments || sums_vs_x)), "mask_beta (ReLU mask re-derived from x) excludes y and the fused-sum forms");
  EMRT_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0, "C and ld must be multiples of 4");
  int threads, rpp, grid;
  EMRT_REQUIRE(bn_rowgeom(M, C, threads, rpp, grid), "unsupported channel count (C/4 must divide 256, or C <= 2048)");
  hipStream_t st = (hipStream_t)stream;
  if (mask_beta) {
    DT_SWITCH(dtype,
              hipLaunchKernelGGL((bn_bwd_dx_kernel<float, true>), dim3(grid), dim3(threads), (size_t)2 * C * sizeof(float), st, (const float*)x, ldx, (const float*)dy, lddy, (const float*)nullptr, 0, (float*)dx, lddx, (float*)dres, lddres, mean, invstd, gamma, sums, local_sums, 1.0 / count, dgamma, dbeta, M, C, rpp, nullptr, 0, mask_beta),
              hipLaunchKernelGGL((bn_bwd_dx_kernel<bf16_t, true>), dim3(grid), dim3(threads), (size_t)2 * C * sizeof(float), st, (const bf16_t*)x, ldx, (const bf16_t*)dy, lddy, (const bf16_t*)nullptr, 0, (bf16_t*)dx, lddx, (bf16_t*)dres, lddres, mean, invstd, gamma, sums, local_sums, 1.0 / count, dgamma, dbeta, M, C, rpp, nullptr, 0, mask_beta));
    return check_launch("emrt_bn_bwd_dx");
  }
  DT_SWITCH(dtype,
            hipLaunchKernelGGL((bn_bwd_dx_kernel<float>), dim3(grid), dim3(threads), (size_t)2 * C * sizeof(float), st, (const float*)x, ldx, (const float*)dy, lddy, (const float*)y, ldy, (float*)dx, lddx, (float*)dres, lddres, mean, invstd, gamma, sums, local_sums, 1.0 / count, dgamma, dbeta, M, C, rpp, beta_y_moments, sums_vs_x, mask_beta),
            hipLaunchKernelGGL((bn_bwd_dx_kernel<bf16_t>), dim3(grid), dim3(threads), (size_t)2 * C * sizeof(float), st, (const bf16_t*)x, ldx, (const bf16_t*)dy, lddy, (const bf16_t*)y, ldy, (bf16_t*)dx, lddx, (bf16_t*)dres, lddres, mean, invstd, gamma, sums, local_sums, 1.0 / count, dgamma, dbeta, M, C, rpp, beta_y_moments, sums_vs_x, mask_beta));
  return check_launch("emrt_bn_bwd_dx");
}

// narrow matrices (C not a multiple of 4: class logits, reference-point coordinates): one block per channel
template <class T>
__global__ __launch_bounds__(256) void colsum_scalar_kernel(const T* __restrict__ x, int ldx, long long rpb, long long bs, long long M,
                                                            float* __restrict__ dbias) {
  __shared__ float red[4];
  const int c = blockIdx.x;
  float s = 0.f;
  for (long long r = threadIdx.x; r < M; r += 256) {
    const long long bb = r / rpb;
    s += to_f32(x[bb * bs + (r - bb * rpb) * ldx + c]);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) dbias[c] += red[0] + red[1] + red[2] + red[3];
}

// per-channel column sum (bias / embedding gradient): dbias[c] += sum_r x[r][c]; row r lives at
// x + (r / rows_per_batch) * x_bs + (r % rows_per_batch) * ldx   (rows_per_batch == M, x_bs == 0 for a dense matrix)
extern "C" int emrt_colsum_acc(const void* x, int ldx, long long rows_per_batch, long long x_bs, long long M, int C, float* dbias,
                               void* workspace, int dtype, void* stream) {
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  EMRT_REQUIRE(rows_per_batch > 0, "bad batch geometry");
  EMRT_REQUIRE(x && dbias && workspace, "null pointer");
  if (C % 4 != 0 || ldx % 4 != 0 || x_bs % 4 != 0) {
    hipStream_t st0 = (hipStream_t)stream;
    if (dtype == EMRT_F32) hipLaunchKernelGGL((colsum_scalar_kernel<float>), dim3(C), dim3(256), 0, st0, (const float*)x, ldx, rows_per_batch, x_bs, M, dbias);
    else hipLaunchKernelGGL((colsum_scalar_kernel<bf16_t>), dim3(C), dim3(256), 0, st0, (const bf16_t*)x, ldx, rows_per_batch, x_bs, M, dbias);
    return check_launch("emrt_colsum_acc(scalar)");
  }
  int tx, gx, gy;
  col_reduce_geometry(M, C, tx, gx, gy);
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace;
  const bool direct = g_tune.ln_atomic != 0;       // developer knob shared with emrt_layernorm_bwd: 0 = partials + finalize launch
  DT_SWITCH(dtype,
            hipLaunchKernelGGL((col_reduce_kernel<float, 2>), dim3(gx, gy), dim3(256), 0, st, (const float*)x, ldx, nullptr, 0, nullptr, 0, nullptr, nullptr, M, C, tx, partial, rows_per_batch, x_bs, nullptr, direct ? dbias : nullptr),
            hipLaunchKernelGGL((col_reduce_kernel<bf16_t, 2>), dim3(gx, gy), dim3(256), 0, st, (const bf16_t*)x, ldx, nullptr, 0, nullptr, 0, nullptr, nullptr, M, C, tx, partial, rows_per_batch, x_bs, nullptr, direct ? dbias : nullptr));
  if (!direct) hipLaunchKernelGGL(partials_acc_kernel, dim3((C + 31) / 32), dim3(256), 0, st, partial, gx, C, (float*)nullptr, dbias);   // slot 0 only
  return check_launch("emrt_colsum_acc");
}

// dst[l][c] += sum over the T tensors, the B batch elements and the tokens of level l of x_t[b][tok][c]: the gradient of the level embedding
// (transformer_encoder_decoder.py:447-448: pos = sine + level_embed[l], added to the query of EVERY encoder layer) from the layers' query gradients
// in ONE launch -- round 5 summed the layers' gradients pairwise (an add per layer) and then reduced each level on its own (three launches).
struct ColsumLevelsArgs {
  const void* x[8];
  int T, L, B, Lv, C;
  int start[4], count[4];
  float* dst;
};
template <class T>
__global__ __launch_bounds__(256) void colsum_levels_kernel(ColsumLevelsArgs a) {
  // block (j, l): token rows j, j + gridDim.x, ... of level l, every tensor and batch element; thread = (row lane, channel quad)
  const int quads = a.C / 4;
  const int cq = threadIdx.x % quads, lane_row = threadIdx.x / quads, lanes = 256 / quads;
  const int l = blockIdx.y;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  // (tensor, batch) pairs p = t * B + b; blockIdx.z takes every gridDim.z-th pair, four loads in flight per thread (the first version walked all
  // T * B = 32 rows of a token one after the other in 384 blocks: 27.8 us of load latency, profiles/r6a_timeline_cfg2.txt)
  const int npair = a.T * a.B;
  for (int r = blockIdx.x * lanes + lane_row; r < a.count[l]; r += gridDim.x * lanes) {
    const long long row = a.start[l] + r;
    for (int p0 = blockIdx.z; p0 < npair; p0 += 4 * gridDim.z) {
      float v[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pp = p0 + u * gridDim.z;
        const int t = pp < npair ? pp / a.B : 0, b = pp < npair ? pp - t * a.B : 0;
        Vec4<T>::load((const T*)a.x[t] + ((long long)b * a.Lv + row) * a.C + cq * 4, v[u]);
        if (pp >= npair) { v[u][0] = 0.f; v[u][1] = 0.f; v[u][2] = 0.f; v[u][3] = 0.f; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += v[u][e];
    }
  }
  __shared__ float red[256 * 4];
#pragma unroll
  for (int e = 0; e < 4; ++e) red[(lane_row * quads + cq) * 4 + e] = acc[e];
  __syncthreads();
  for (int c = threadIdx.x; c < a.C; c += 256) {
    float s = 0.f;
    for (int q = 0; q < lanes; ++q) s += red[(q * quads + c / 4) * 4 + (c & 3)];
    atomicAdd(a.dst + (long long)l * a.C + c, s);
  }
}

extern "C" int emrt_colsum_levels_multi(const void* const* xs, int T, const int* level_start, const int* level_count, int L, int B, int Lv, int C,
                                        float* dst, int dtype, void* stream) {
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  EMRT_REQUIRE(xs && level_start && level_count && dst, "null pointer");
  EMRT_REQUIRE(T >= 1 && T <= 8 && L >= 1 && L <= 4 && B > 0 && Lv > 0, "1..8 tensors, 1..4 levels");
  EMRT_REQUIRE(C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0, "C / 4 must divide 256");
  ColsumLevelsArgs a;
  memset(&a, 0, sizeof(a));
  int most = 0;
  for (int t = 0; t < T; ++t) {
    EMRT_REQUIRE(xs[t] && ((uintptr_t)xs[t]) % 16 == 0, "null or unaligned tensor");
    a.x[t] = xs[t];
  }
  for (int l = 0; l < L; ++l) {
    EMRT_REQUIRE(level_start[l] >= 0 && level_count[l] > 0 && level_start[l] + level_count[l] <= Lv, "level outside the token axis");
    a.start[l] = level_start[l]; a.count[l] = level_count[l];
    most = level_count[l] > most ? level_count[l] : most;
  }
  a.T = T; a.L = L; a.B = B; a.Lv = Lv; a.C = C; a.dst = dst;
  const int lanes = 256 / (C / 4);
  int gx = (most + lanes - 1) / lanes;          // one token row per row lane and block at most
  if (gx > 64) gx = 64;
  const int gz = T * B >= 8 ? 4 : 1;            // the (tensor, batch) pairs over 4 blocks: <= 64 x L x 4 blocks, each ending in C fp32 atomics
  hipStream_t st = (hipStream_t)stream;
  DT_SWITCH(dtype, hipLaunchKernelGGL((colsum_levels_kernel<float>), dim3(gx, L, gz), dim3(256), 0, st, a),
            hipLaunchKernelGGL((colsum_levels_kernel<bf16_t>), dim3(gx, L, gz), dim3(256), 0, st, a));
  return check_launch("emrt_colsum_levels_multi");
}

// ------------------------------------------------------------------------------------------------
// Grouped BatchNorm over several SMALL, independent problems in one launch per pass (ABI 8): the four pyramid-pooling branches
// (paddle_EMRT.py:61-66,70-78: AdaptiveAvgPool(k) -> conv1x1 -> SyncBatchNorm -> ReLU, k = 1 / 3 / 6 / 8: 8 ... 512 rows of 256 channels at batch 8)
// were 4 emrt_bn_apply launches forward and 4 x (emrt_bn_bwd_reduce + emrt_bn_bwd_dx) backward of ~5 us each for a few hundred KB.  Same arithmetic
// as the single-problem kernels (bn_chan / bn_scale_shift / one fmaf: the forward is bit-identical to emrt_bn_apply); plain kernels, not tuned for
// streaming -- every problem here is latency, which is the point of sharing the launch.  Not for SyncBatchNorm over ranks (the sums would have to be
// all-reduced between the passes: functional.conv_bn_group keeps its own path for that).
// ------------------------------------------------------------------------------------------------
#define EMRT_MAX_BNGROUP 8
struct EmrtBnGroupDesc {
  const void* x;          // raw map [M][C], row stride ldx
  void* y;                // forward: out = [relu](BN(x)); backward: the forward's output (ReLU mask source; nullable when relu == 0)
  const void* dy;         // backward: gradient of y
  void* dx;               // backward: gradient of x
  double* sums;           // forward: complete (sum x, sum x^2) [8][2C]; backward: ZEROED [8][2C], receives (sum dy', sum dy' * xhat)
  float* mean; float* invstd; float* run_mean; float* run_var;
  const float* gamma; const float* beta; float* dgamma; float* dbeta;
  double count;
  float eps, momentum;
  int M, C, ldx, ldy, lddy, lddx, relu;
  const void* res;        // forward, nullable: y = [relu](BN(x)) + res  (Conv2dBlock's "conv2(conv1(x)) + x", paddle_EMRT.py:24-29; its gradient is dy itself)
  int ldres, res_hw;      // row r of the problem is pixel r % res_hw of image r / res_hw: res + image * res_bs + pixel * ldres (a level slab of the token tensor)
  long long res_bs;
};
struct BnGroupArgs { EmrtBnGroupDesc d[EMRT_MAX_BNGROUP]; int first[EMRT_MAX_BNGROUP + 1]; int n; };

__device__ __forceinline__ int bn_group_pick(const BnGroupArgs& g, int& local, int& nblk) {
  int i = 0;
  for (int k = 1; k < g.n; ++k) i += (int)blockIdx.x >= g.first[k] ? 1 : 0;
  local = (int)blockIdx.x - g.first[i];
  nblk = g.first[i + 1] - g.first[i];
  return i;
}

template <class T>
__global__ __launch_bounds__(256) void bn_group_apply_kernel(BnGroupArgs g) {
  int local, nblk;
  const EmrtBnGroupDesc& d = g.d[bn_group_pick(g, local, nblk)];
  extern __shared__ float bn_lds[];      // [2][C]
  const int C = d.C, quads = C / 4, lanes = 256 / quads;
  const int c = ((int)threadIdx.x % quads) * 4, lane_row = (int)threadIdx.x / quads;
  const double inv_count = 1.0 / d.count;
  for (int ch = threadIdx.x; ch < C; ch += 256) {
    const BnChan k = bn_chan(d.sums, nullptr, nullptr, C, ch, inv_count, d.eps);
    float sc, sh;
    bn_scale_shift(k.mean, k.invstd, d.gamma[ch], d.beta[ch], sc, sh);
    bn_lds[ch] = sc;
    bn_lds[C + ch] = sh;
    if (local == 0) {
      d.mean[ch] = k.mean;
      d.invstd[ch] = k.invstd;
      if (d.run_mean) {
        const double mu = rep_sum(d.sums, C, ch) * inv_count;
        double var = rep_sum(d.sums, C, C + ch) * inv_count - mu * mu;
        if (var < 0.0) var = 0.0;
        d.run_mean[ch] = d.momentum * d.run_mean[ch] + (1.f - d.momentum) * (float)mu;
        d.run_var[ch] = d.momentum * d.run_var[ch] + (1.f - d.momentum) * (float)var;
      }
    }
  }
  __syncthreads();
  if (lane_row >= lanes) return;
  float sc[4], sh[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { sc[e] = bn_lds[c + e]; sh[e] = bn_lds[C + c + e]; }
  const T* x = (const T*)d.x;
  T* y = (T*)d.y;
  for (int r = local * lanes + lane_row; r < d.M; r += nblk * lanes) {
    float v[4], o[4];
    Vec4<T>::load(x + (long long)r * d.ldx + c, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = fmaf(v[e], sc[e], sh[e]); if (d.relu) o[e] = fmaxf(o[e], 0.f); }
    if (d.res) {
      float w[4];
      const int img = r / d.res_hw;
      Vec4<T>::load((const T*)d.res + (long long)img * d.res_bs + (long long)(r - img * d.res_hw) * d.ldres + c, w);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] += w[e];
    }
    Vec4<T>::store(y + (long long)r * d.ldy + c, o);
  }
}

// backward: the ReLU mask is re-derived from the raw x with the forward's own expression (bn_scale_shift + one fmaf on the saved mean / invstd: the same
// bits, so forward and backward agree on every element) -- the output y may hold "+ res" and is not read
// backward, pass 1: sums[0][c] += sum_r dy', sums[0][C + c] += sum_r dy' * xhat   (dy' = dy where y > 0; xhat from the raw x and the saved statistics)
template <class T>
__global__ __launch_bounds__(256) void bn_group_bwd_reduce_kernel(BnGroupArgs g) {
  int local, nblk;
  const EmrtBnGroupDesc& d = g.d[bn_group_pick(g, local, nblk)];
  __shared__ float red[2][256 * 4];
  const int C = d.C, quads = C / 4, lanes = 256 / quads;
  const int c = ((int)threadIdx.x % quads) * 4, lane_row = (int)threadIdx.x / quads;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  if (lane_row < lanes) {
    float mu[4], is[4], sc[4], sh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { mu[e] = d.mean[c + e]; is[e] = d.invstd[c + e]; bn_scale_shift(mu[e], is[e], d.gamma[c + e], d.beta[c + e], sc[e], sh[e]); }
    const T* x = (const T*)d.x;
    const T* dy = (const T*)d.dy;
    for (int r = local * lanes + lane_row; r < d.M; r += nblk * lanes) {
      float v[4], gq[4];
      Vec4<T>::load(x + (long long)r * d.ldx + c, v);
      Vec4<T>::load(dy + (long long)r * d.lddy + c, gq);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gd = (!d.relu || fmaf(v[e], sc[e], sh[e]) > 0.f) ? gq[e] : 0.f;
        s1[e] += gd;
        s2[e] = fmaf(gd, (v[e] - mu[e]) * is[e], s2[e]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[0][threadIdx.x * 4 + e] = s1[e]; red[1][threadIdx.x * 4 + e] = s2[e]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int which = i / C, ch = i - which * C;
    float a = 0.f;
    for (int q = 0; q < lanes; ++q) a += red[which][(q * quads + ch / 4) * 4 + (ch & 3)];
    atomicAdd(d.sums + (long long)(local & 7) * 2 * C + i, (double)a);
  }
}

// backward, pass 2: dx = gamma * invstd * (dy' - S1 / count - xhat * S2 / count); the problem's first block adds dgamma += S2, dbeta += S1
template <class T>
__global__ __launch_bounds__(256) void bn_group_bwd_dx_kernel(BnGroupArgs g) {
  int local, nblk;
  const EmrtBnGroupDesc& d = g.d[bn_group_pick(g, local, nblk)];
  extern __shared__ float bn_lds[];      // [4][C]: k = gamma * invstd, m1 = S1 / count, m2 = S2 / count, (unused)
  const int C = d.C, quads = C / 4, lanes = 256 / quads;
  const int c = ((int)threadIdx.x % quads) * 4, lane_row = (int)threadIdx.x / quads;
  const double inv_count = 1.0 / d.count;
  for (int ch = threadIdx.x; ch < C; ch += 256) {
    const double S1 = rep_sum(d.sums, C, ch), S2 = rep_sum(d.sums, C, C + ch);
    bn_lds[ch] = d.gamma[ch] * d.invstd[ch];
    bn_lds[C + ch] = (float)(S1 * inv_count);
    bn_lds[2 * C + ch] = (float)(S2 * inv_count);
    if (local == 0) {
      if (d.dbeta) d.dbeta[ch] += (float)S1;
      if (d.dgamma) d.dgamma[ch] += (float)S2;
    }
  }
  __syncthreads();
  if (lane_row >= lanes) return;
  float k[4], m1[4], m2[4], mu[4], is[4], sc[4], sh[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    k[e] = bn_lds[c + e]; m1[e] = bn_lds[C + c + e]; m2[e] = bn_lds[2 * C + c + e]; mu[e] = d.mean[c + e]; is[e] = d.invstd[c + e];
    bn_scale_shift(mu[e], is[e], d.gamma[c + e], d.beta[c + e], sc[e], sh[e]);
  }
  const T* x = (const T*)d.x;
  const T* dy = (const T*)d.dy;
  T* dx = (T*)d.dx;
  for (int r = local * lanes + lane_row; r < d.M; r += nblk * lanes) {
    float v[4], gq[4], o[4];
    Vec4<T>::load(x + (long long)r * d.ldx + c, v);
    Vec4<T>::load(dy + (long long)r * d.lddy + c, gq);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gd = (!d.relu || fmaf(v[e], sc[e], sh[e]) > 0.f) ? gq[e] : 0.f;
      o[e] = k[e] * (gd - m1[e] - (v[e] - mu[e]) * is[e] * m2[e]);
    }
    Vec4<T>::store(dx + (long long)r * d.lddx + c, o);
  }
}

static int bn_group_fill(BnGroupArgs& g, const EmrtBnGroupDesc* descs, int n, int pass, const char* fn) {
  int total = 0;
  g.n = n;
  for (int i = 0; i < n; ++i) {
    const EmrtBnGroupDesc& d = descs[i];
    if (!d.x || !d.sums || !d.mean || !d.invstd || !d.gamma || !d.beta || d.M <= 0 || !(d.count > 0.0)) return fail(fn, "null pointer / bad sizes");
    if (d.C % 4 != 0 || d.C / 4 > 256 || 256 % (d.C / 4) != 0) return fail(fn, "C / 4 must divide 256");
    if (d.ldx % 4 || ((uintptr_t)d.x) % 8) return fail(fn, "x rows must be 8-byte aligned");
    if (pass == 0 && (!d.y || d.ldy % 4)) return fail(fn, "forward needs y");
    if (pass != 0 && (!d.dy || (pass == 2 && !d.dx) || d.lddy % 4 || (pass == 2 && d.lddx % 4))) return fail(fn, "backward needs dy and dx");
    if (pass == 0 && d.res && (d.ldres % 4 || d.res_bs % 4 || d.res_hw <= 0)) return fail(fn, "res rows must be 8-byte aligned, res_hw > 0");
    if ((d.run_mean != nullptr) != (d.run_var != nullptr)) return fail(fn, "running statistics come in pairs");
    g.d[i] = d;
    g.first[i] = total;
    const int lanes = 256 / (d.C / 4);
    int nb = (d.M + lanes * 4 - 1) / (lanes * 4);          // ~4 rows per thread
    if (nb < 1) nb = 1;
    if (nb > 256) nb = 256;
    total += nb;
  }
  for (int i = n; i <= EMRT_MAX_BNGROUP; ++i) g.first[i] = total;
  for (int i = n; i < EMRT_MAX_BNGROUP; ++i) g.d[i] = g.d[0];
  return total > 0 ? 0 : fail(fn, "no work");
}

extern "C" int emrt_bn_group_apply(const EmrtBnGroupDesc* descs, int n, int dtype, void* stream) {
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  EMRT_REQUIRE(descs && n >= 1 && n <= EMRT_MAX_BNGROUP, "1..8 problems");
  BnGroupArgs g;
  if (bn_group_fill(g, descs, n, 0, "emrt_bn_group_apply")) return -1;
  int cmax = 0;
  for (int i = 0; i < n; ++i) cmax = descs[i].C > cmax ? descs[i].C : cmax;
  hipStream_t st = (hipStream_t)stream;
  DT_SWITCH(dtype, hipLaunchKernelGGL((bn_group_apply_kernel<float>), dim3(g.first[n]), dim3(256), 2 * cmax * sizeof(float), st, g),
            hipLaunchKernelGGL((bn_group_apply_kernel<bf16_t>), dim3(g.first[n]), dim3(256), 2 * cmax * sizeof(float), st, g));
  return check_launch("emrt_bn_group_apply");
}

// both backward passes (reduce, then dx: two launches); `sums` of every problem must be ZERO on entry
extern "C" int emrt_bn_group_bwd(const EmrtBnGroupDesc* descs, int n, int dtype, void* stream) {
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  EMRT_REQUIRE(descs && n >= 1 && n <= EMRT_MAX_BNGROUP, "1..8 problems");
  BnGroupArgs g;
  if (bn_group_fill(g, descs, n, 2, "emrt_bn_group_bwd")) return -1;
  int cmax = 0;
  for (int i = 0; i < n; ++i) cmax = descs[i].C > cmax ? descs[i].C : cmax;
  hipStream_t st = (hipStream_t)stream;
  DT_SWITCH(dtype, hipLaunchKernelGGL((bn_group_bwd_reduce_kernel<float>), dim3(g.first[n]), dim3(256), 0, st, g),
            hipLaunchKernelGGL((bn_group_bwd_reduce_kernel<bf16_t>), dim3(g.first[n]), dim3(256), 0, st, g));
  DT_SWITCH(dtype, hipLaunchKernelGGL((bn_group_bwd_dx_kernel<float>), dim3(g.first[n]), dim3(256), 4 * cmax * sizeof(float), st, g),
            hipLaunchKernelGGL((bn_group_bwd_dx_kernel<bf16_t>), dim3(g.first[n]), dim3(256), 4 * cmax * sizeof(float), st, g));
  return check_launch("emrt_bn_group_bwd");
}

// one block per (image, group) when the slice is small enough for two cheap passes and the group is 4..256 channels wide
static inline bool gn_use_fused(int HW, int C, int G) {
  const int cpg = C / G;
  return HW <= 4096 && cpg % 4 == 0 && cpg / 4 <= 64 && 256 % (cpg / 4) == 0 && (cpg / 4) * 64 >= 64;
}

static inline int gn_geom(int HW, int C, int* pix_per_block) {
  const int lanes = 256 / (C / 4);
  int ppb = lanes * 8;                       // each thread walks ~8 pixels
  int blocks = (HW + ppb - 1) / ppb;
  if (blocks > 64) { blocks = 64; ppb = (HW + 63) / 64; ppb = (ppb + lanes - 1) / lanes * lanes; blocks = (HW + ppb - 1) / ppb; }
  *pix_per_block = ppb;
  return blocks;
}

// workspace: fp64 [N*G*2], PRE-ZEROED by the caller (group sums)
extern "C" int emrt_groupnorm_fwd(const void* x, int ldx, long long x_bs, const void* res, int ldres, long long res_bs, void* out,
                                  int ldout, long long out_bs, const float* gamma, const float* beta, float* mean, float* rstd,
                                  double* workspace, int N, int HW, int C, int G, float eps, int gelu, int dtype, void* stream) {
  EMRT_REQUIRE_FWD_DTYPE(dtype);
  EMRT_REQUIRE(x && out && gamma && beta && workspace, "null pointer");
  EMRT_REQUIRE(C % 4 == 0 && (C / 4) <= 256 && 256 % (C / 4) == 0 && G > 0 && G <= 256 && C % G == 0 && (C / G) % 4 == 0, "unsupported C/G");
  EMRT_REQUIRE(ldx % 4 == 0 && ldout % 4 == 0 && x_bs % 4 == 0 && out_bs % 4 == 0, "strides must be multiples of 4");
  hipStream_t st = (hipStream_t)stream;
  if (gn_use_fused(HW, C, G)) {
    DT_SWITCH3(dtype,
              hipLaunchKernelGGL((gn_fused_fwd_kernel<float>), dim3(N * G), dim3(256), 0, st, (const float*)x, ldx, x_bs, (const float*)res, ldres, res_bs, (float*)out, ldout, out_bs, gamma, beta, mean, rstd, HW, C, G, eps, gelu),
              hipLaunchKernelGGL((gn_fused_fwd_kernel<bf16_t>), dim3(N * G), dim3(256), 0, st, (const bf16_t*)x, ldx, x_bs, (const bf16_t*)res, ldres, res_bs, (bf16_t*)out, ldout, out_bs, gamma, beta, mean, rstd, HW, C, G, eps, gelu),
              hipLaunchKernelGGL((gn_fused_fwd_kernel<f16_t>), dim3(N * G), dim3(256), 0, st, (const f16_t*)x, ldx, x_bs, (const f16_t*)res, ldres, res_bs, (f16_t*)out, ldout, out_bs, gamma, beta, mean, rstd, HW, C, G, eps, gelu));
    return check_launch("emrt_groupnorm_fwd");
  }
  int ppb;
  const int blocks = gn_geom(HW, C, &ppb);
  DT_SWITCH3(dtype,
            hipLaunchKernelGGL((gn_stats_kernel<float>), dim3(blocks, N), dim3(256), 0, st, (const float*)x, ldx, x_bs, workspace, HW, C, G, ppb),
            hipLaunchKernelGGL((gn_stats_kernel<bf16_t>), dim3(blocks, N), dim3(256), 0, st, (const bf16_t*)x, ldx, x_bs, workspace, HW, C, G, ppb),
            hipLaunchKernelGGL((gn_stats_kernel<f16_t>), dim3(blocks, N), dim3(256), 0, st, (const f16_t*)x, ldx, x_bs, workspace, HW, C, G, ppb));
  DT_SWITCH3(dtype,
            hipLaunchKernelGGL((gn_apply_kernel<float>), dim3(blocks, N), dim3(256), 0, st, (const float*)x, ldx, x_bs, (const float*)res, ldres, res_bs, (float*)out, ldout, out_bs, gamma, beta, workspace, mean, rstd, HW, C, G, eps, gelu, ppb),
            hipLaunchKernelGGL((gn_apply_kernel<bf16_t>), dim3(blocks, N), dim3(256), 0, st, (const bf16_t*)x, ldx, x_bs, (const bf16_t*)res, ldres, res_bs, (bf16_t*)out, ldout, out_bs, gamma, beta, workspace, mean, rstd, HW, C, G, eps, gelu, ppb),
            hipLaunchKernelGGL((gn_apply_kernel<f16_t>), dim3(blocks, N), dim3(256), 0, st, (const f16_t*)x, ldx, x_bs, (const f16_t*)res, ldres, res_bs, (f16_t*)out, ldout, out_bs, gamma, beta, workspace, mean, rstd, HW, C, G, eps, gelu, ppb));
  return check_launch("emrt_groupnorm_fwd");
}

// workspace: fp64 [N*C*2], PRE-ZEROED by the caller (per-image per-channel sums)
extern "C" int emrt_groupnorm_bwd(const void* x, int ldx, long long x_bs, const void* dy, int lddy, long long dy_bs, void* dx,
                                  int lddx, long long dx_bs, const float* gamma, const float* beta, const float* mean,
                                  const float* rstd, float* dgamma, float* dbeta, double* workspace, int N, int HW, int C, int G,
                                  int gelu, int dtype, void* stream) {
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  EMRT_REQUIRE(x && dy && dx && gamma && beta && mean && rstd && workspace, "null pointer");
  EMRT_REQUIRE(C % 4 == 0 && (C / 4) <= 256 && 256 % (C / 4) == 0 && G > 0 && G <= 256 && C % G == 0 && (C / G) % 4 == 0, "unsupported C/G");
  hipStream_t st = (hipStream_t)stream;
  if (gn_use_fused(HW, C, G)) {
    DT_SWITCH(dtype,
              hipLaunchKernelGGL((gn_fused_bwd_kernel<float>), dim3(N * G), dim3(256), 0, st, (const float*)x, ldx, x_bs, (const float*)dy, lddy, dy_bs, (float*)dx, lddx, dx_bs, gamma, beta, mean, rstd, dgamma, dbeta, HW, C, G, gelu),
              hipLaunchKernelGGL((gn_fused_bwd_kernel<bf16_t>), dim3(N * G), dim3(256), 0, st, (const bf16_t*)x, ldx, x_bs, (const bf16_t*)dy, lddy, dy_bs, (bf16_t*)dx, lddx, dx_bs, gamma, beta, mean, rstd, dgamma, dbeta, HW, C, G, gelu));
    return check_launch("emrt_groupnorm_bwd");
  }
  int ppb;
  const int blocks = gn_geom(HW, C, &ppb);
  DT_SWITCH(dtype,
            hipLaunchKernelGGL((gn_bwd_reduce_kernel<float>), dim3(blocks, N), dim3(256), 0, st, (const float*)x, ldx, x_bs, (const float*)dy, lddy, dy_bs, gamma, beta, mean, rstd, workspace, HW, C, G, gelu, ppb),
            hipLaunchKernelGGL((gn_bwd_reduce_kernel<bf16_t>), dim3(blocks, N), dim3(256), 0, st, (const bf16_t*)x, ldx, x_bs, (const bf16_t*)dy, lddy, dy_bs, gamma, beta, mean, rstd, workspace, HW, C, G, gelu, ppb));
  const size_t lds = (size_t)2 * G * sizeof(float);
  DT_SWITCH(dtype,
            hipLaunchKernelGGL((gn_bwd_dx_kernel<float>), dim3(blocks, N), dim3(256), lds, st, (const float*)x, ldx, x_bs, (const float*)dy, lddy, dy_bs, (float*)dx, lddx, dx_bs, gamma, beta, mean, rstd, workspace, dgamma, dbeta, HW, C, G, gelu, ppb),
            hipLaunchKernelGGL((gn_bwd_dx_kernel<bf16_t>), dim3(blocks, N), dim3(256), lds, st, (const bf16_t*)x, ldx, x_bs, (const bf16_t*)dy, lddy, dy_bs, (bf16_t*)dx, lddx, dx_bs, gamma, beta, mean, rstd, workspace, dgamma, dbeta, HW, C, G, gelu, ppb));
  return check_launch("emrt_groupnorm_bwd");
}

extern "C" int emrt_layernorm_fwd(const void* a, const void* b, const void* post, void* z, void* out, const float* gamma,
                                  const float* beta, float* mean, float* rstd, long long rows, int C, float eps, float pdrop,
                                  const unsigned long long* seed, unsigned salt, const void* qpos, int qpos_rows, void* q_out, int dtype, void* stream) {
  EMRT_REQUIRE_FWD_DTYPE(dtype);
  EMRT_REQUIRE(a && out && gamma && beta, "null pointer");
  EMRT_REQUIRE((q_out == nullptr) == (qpos == nullptr) && (!q_out || (qpos_rows > 0 && q_out != out)), "q_out needs qpos [qpos_rows][C] and a buffer of its own");
  EMRT_REQUIRE(pdrop >= 0.f && pdrop < 1.f && (pdrop == 0.f || (seed && b)), "dropout needs 0 <= p < 1, a device seed and a branch input b");
  EMRT_REQUIRE(C % 4 == 0 && C <= 1024, "C must be a multiple of 4 and <= 1024");
  hipStream_t st = (hipStream_t)stream;
  const int grid = (int)((rows + 3) / 4);
  DT_SWITCH3(dtype,
            hipLaunchKernelGGL((ln_fwd_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)a, (const float*)b, (const float*)post, (float*)z, (float*)out, gamma, beta, mean, rstd, rows, C, eps, pdrop, seed, salt, (const float*)qpos, qpos_rows, (float*)q_out),
            hipLaunchKernelGGL((ln_fwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, (const bf16_t*)post, (bf16_t*)z, (bf16_t*)out, gamma, beta, mean, rstd, rows, C, eps, pdrop, seed, salt, (const bf16_t*)qpos, qpos_rows, (bf16_t*)q_out),
            hipLaunchKernelGGL((ln_fwd_kernel<f16_t>), dim3(grid), dim3(256), 0, st, (const f16_t*)a, (const f16_t*)b, (const f16_t*)post, (f16_t*)z, (f16_t*)out, gamma, beta, mean, rstd, rows, C, eps, pdrop, seed, salt, (const f16_t*)qpos, qpos_rows, (f16_t*)q_out));
  return check_launch("emrt_layernorm_fwd");
}

// blocks of the LayerNorm backward: ~32 rows (8 per wave) each, at most 512 partial-sum rows for the finalize
static inline bool ln_bwd_wide(int C) { return C <= 256 && g_tune.ln_bwd_threads >= 512; }      // 8 waves per block, 32 rows per iteration
static inline long long ln_bwd_blocks(long long rows, int C) {
  // every block ends in 2C fp32 atomics on the SAME 2C addresses: few, long blocks (sweep at 10 752 rows x 256: 336 blocks of 256 threads 225 us over the
  // step's 14 launches, 168 blocks of 512 threads 215 us, 672 blocks of 256 threads 270 us)
  const bool wide = ln_bwd_wide(C);
  const int per = g_tune.ln_bwd_rows > 0 ? g_tune.ln_bwd_rows : (wide ? 64 : 32);
  const int cap = g_tune.ln_bwd_max_blocks > 0 ? g_tune.ln_bwd_max_blocks : (wide ? 256 : 512);
  long long blocks = (rows + per - 1) / per;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return blocks;
}

extern "C" size_t emrt_layernorm_bwd_workspace_bytes(long long rows, int C) {
  return ((size_t)(ln_bwd_blocks(rows, C) + 1) * 2 * C + 2 * (size_t)C) * sizeof(float);
}

extern "C" int emrt_layernorm_bwd(const void* z, const void* dy, void* dz, const float* gamma, const float* mean, const float* rstd,
                                  float* dgamma, float* dbeta, long long rows, int C, void* workspace, void* dz_branch, float pdrop,
                                  const unsigned long long* seed, unsigned salt, const void* dz_addend, const void* dy2, void* dysum, int dtype, void* stream) {
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  EMRT_REQUIRE(z && dy && dz && gamma && mean && rstd && workspace, "null pointer");
  EMRT_REQUIRE(!dysum || (dy2 && dysum != dz && dysum != dy && dysum != dy2), "dysum (= dy + dy2) needs dy2 and a buffer of its own");
  EMRT_REQUIRE(pdrop >= 0.f && pdrop < 1.f && (pdrop == 0.f || (seed && dz_branch)), "dropout needs 0 <= p < 1, a device seed and dz_branch");
  EMRT_REQUIRE(!dz_addend || dz_addend != dz, "dz_addend must not alias dz");
  EMRT_REQUIRE(C % 4 == 0 && C <= 1024, "C must be a multiple of 4 and <= 1024");
  long long blocks = ln_bwd_blocks(rows, C);
  int rpb = (int)((rows + blocks - 1) / blocks);
  const bool small = C <= 256;                        // four rows in flight per wave (16 per block and iteration), else two
  const bool wide = ln_bwd_wide(C);
  const int step = wide ? 32 : small ? 16 : 8;
  rpb = (rpb + step - 1) / step * step;
  blocks = (rows + rpb - 1) / rpb;
  float* partial = (float*)workspace;
  const bool direct = (dgamma || dbeta) && g_tune.ln_atomic != 0;       // developer knob: 0 = partials + finalize launch
  const size_t lds = (size_t)(wide ? 16 : 8) * C * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
#define LN_BWD_ARGS(T) (const T*)z, (const T*)dy, (T*)dz, gamma, mean, rstd, partial, rows, C, rpb, (T*)dz_branch, pdrop, seed, salt, direct ? dgamma : nullptr, direct ? dbeta : nullptr, (const T*)dz_addend, (const T*)dy2, (T*)dysum
#define LN_BWD_LAUNCH_W(T, R, NQ) do { if (dy2) hipLaunchKernelGGL((ln_bwd_kernel<T, R, NQ, 512, true>), dim3((unsigned)blocks), dim3(512), lds, st, LN_BWD_ARGS(T)); \
                                       else hipLaunchKernelGGL((ln_bwd_kernel<T, R, NQ, 512, false>), dim3((unsigned)blocks), dim3(512), lds, st, LN_BWD_ARGS(T)); } while (0)
#define LN_BWD_LAUNCH(T, R, NQ) do { if (dy2) hipLaunchKernelGGL((ln_bwd_kernel<T, R, NQ, 256, true>), dim3((unsigned)blocks), dim3(256), lds, st, LN_BWD_ARGS(T)); \
                                     else hipLaunchKernelGGL((ln_bwd_kernel<T, R, NQ, 256, false>), dim3((unsigned)blocks), dim3(256), lds, st, LN_BWD_ARGS(T)); } while (0)
  if (wide) DT_SWITCH(dtype, LN_BWD_LAUNCH_W(float, 4, 1), LN_BWD_LAUNCH_W(bf16_t, 4, 1));
  else if (small) DT_SWITCH(dtype, LN_BWD_LAUNCH(float, 4, 1), LN_BWD_LAUNCH(bf16_t, 4, 1));
  else DT_SWITCH(dtype, LN_BWD_LAUNCH(float, 2, 4), LN_BWD_LAUNCH(bf16_t, 2, 4));
#undef LN_BWD_LAUNCH
#undef LN_BWD_LAUNCH_W
#undef LN_BWD_ARGS
  if (!direct) hipLaunchKernelGGL(partials_acc_kernel, dim3((2 * C + 31) / 32), dim3(256), 0, st, partial, (int)blocks, C, dgamma, dbeta);
  return check_launch("emrt_layernorm_bwd");
}

static int gn_fill_levels(GnLevels& lv, const int* level_start, const int* level_hw, int L, const float* const* gamma, const float* const* beta,
                          float* const* dgamma, float* const* dbeta) {
  if (L < 1 || L > GN_MAX_LEVELS) return -1;
  lv.L = L;
  for (int l = 0; l < GN_MAX_LEVELS; ++l) {
    const int s = l < L ? l : 0;
    lv.start[l] = level_start[s]; lv.hw[l] = level_hw[s];
    lv.gamma[l] = gamma[s]; lv.beta[l] = beta[s];
    lv.dgamma[l] = dgamma ? dgamma[s] : nullptr; lv.dbeta[l] = dbeta ? dbeta[s] : nullptr;
    if (!lv.gamma[l] || !lv.beta[l] || lv.hw[l] < 1 || lv.hw[l] > 4096 || lv.start[l] < 0) return -1;
  }
  return 0;
}

// GroupNorm (+GELU) (+residual) of L level slabs of token tensors [N][Lv][C] in one launch; mean/rstd are [L][N*G].
static int gn_rows_plan(GnRows& gr, int rows_per_block) {
  gr.rows_per_block = rows_per_block;
  gr.per_image = 0;
  for (int l = 0; l < GN_MAX_LEVELS; ++l) {
    gr.nblk[l] = l < gr.lv.L ? (gr.lv.hw[l] + rows_per_block - 1) / rows_per_block : 0;
    gr.per_image += gr.nblk[l];
  }
  return gr.per_image;
}
// the row-major path: 8 channels per group, a whole number of rows per 256-thread pass, 16-byte aligned groups
static inline bool gn_rows_ok(int C, int G, int ld0, int ld1, int ld2, long long bs0, long long bs1, long long bs2) {
  return C == 8 * G && 256 % G == 0 && C <= 256 && ld0 % 8 == 0 && ld1 % 8 == 0 && ld2 % 8 == 0 && bs0 % 8 == 0 && bs1 % 8 == 0 && bs2 % 8 == 0;
}

extern "C" int emrt_groupnorm_levels_fwd(const void* x, int ldx, long long x_bs, const void* res, int ldres, long long res_bs, void* out,
                                         int ldout, long long out_bs, const float* const* gamma, const float* const* beta, float* mean,
                                         float* rstd, const int* level_start, const int* level_hw, int L, int N, int C, int G, float eps,
                                         int gelu, double* stat_ws, int dtype, void* stream) {
  EMRT_REQUIRE_FWD_DTYPE(dtype);
  EMRT_REQUIRE(x && out && gamma && beta && level_start && level_hw, "null pointer");
  EMRT_REQUIRE(G > 0 && C % G == 0 && gn_use_fused(1, C, G), "unsupported C / G for the one-block-per-group kernel");
  EMRT_REQUIRE(ldx % 4 == 0 && ldout % 4 == 0 && x_bs % 4 == 0 && out_bs % 4 == 0 && (!res || (ldres % 4 == 0 && res_bs % 4 == 0)), "strides must be multiples of 4");
  GnLevels lv;
  EMRT_REQUIRE(gn_fill_levels(lv, level_start, level_hw, L, gamma, beta, nullptr, nullptr) == 0, "1..4 levels of at most 4096 rows");
  hipStream_t st = (hipStream_t)stream;
  if (stat_ws && mean && rstd && !g_tune.gn_group_blocks && gn_rows_ok(C, G, ldx, res ? ldres : 8, ldout, x_bs, res ? res_bs : 8, out_bs)) {
    GnRows gs, ga;
    gs.lv = lv; ga.lv = lv;
    const unsigned bs = (unsigned)(gn_rows_plan(gs, g_tune.gn_stat_rows) * N), ba = (unsigned)(gn_rows_plan(ga, g_tune.gn_apply_rows) * N);
    DT_SWITCH3(dtype,
              hipLaunchKernelGGL((gn_rows_stats_kernel<float>), dim3(bs), dim3(256), 0, st, (const float*)x, ldx, x_bs, gs, stat_ws, N, G),
              hipLaunchKernelGGL((gn_rows_stats_kernel<bf16_t>), dim3(bs), dim3(256), 0, st, (const bf16_t*)x, ldx, x_bs, gs, stat_ws, N, G),
              hipLaunchKernelGGL((gn_rows_stats_kernel<f16_t>), dim3(bs), dim3(256), 0, st, (const f16_t*)x, ldx, x_bs, gs, stat_ws, N, G));
    DT_SWITCH3(dtype,
              hipLaunchKernelGGL((gn_rows_apply_kernel<float>), dim3(ba), dim3(256), 0, st, (const float*)x, ldx, x_bs, (const float*)res, ldres, res_bs, (float*)out, ldout, out_bs, ga, stat_ws, mean, rstd, N, G, eps, gelu),
              hipLaunchKernelGGL((gn_rows_apply_kernel<bf16_t>), dim3(ba), dim3(256), 0, st, (const bf16_t*)x, ldx, x_bs, (const bf16_t*)res, ldres, res_bs, (bf16_t*)out, ldout, out_bs, ga, stat_ws, mean, rstd, N, G, eps, gelu),
              hipLaunchKernelGGL((gn_rows_apply_kernel<f16_t>), dim3(ba), dim3(256), 0, st, (const f16_t*)x, ldx, x_bs, (const f16_t*)res, ldres, res_bs, (f16_t*)out, ldout, out_bs, ga, stat_ws, mean, rstd, N, G, eps, gelu));
    return check_launch("emrt_groupnorm_levels_fwd");
  }
  DT_SWITCH3(dtype,
            hipLaunchKernelGGL((gn_levels_fwd_kernel<float>), dim3(L * N * G), dim3(256), 0, st, (const float*)x, ldx, x_bs, (const float*)res, ldres, res_bs, (float*)out, ldout, out_bs, lv, mean, rstd, N, C, G, eps, gelu),
            hipLaunchKernelGGL((gn_levels_fwd_kernel<bf16_t>), dim3(L * N * G), dim3(256), 0, st, (const bf16_t*)x, ldx, x_bs, (const bf16_t*)res, ldres, res_bs, (bf16_t*)out, ldout, out_bs, lv, mean, rstd, N, C, G, eps, gelu),
            hipLaunchKernelGGL((gn_levels_fwd_kernel<f16_t>), dim3(L * N * G), dim3(256), 0, st, (const f16_t*)x, ldx, x_bs, (const f16_t*)res, ldres, res_bs, (f16_t*)out, ldout, out_bs, lv, mean, rstd, N, C, G, eps, gelu));
  return check_launch("emrt_groupnorm_levels_fwd");
}

extern "C" int emrt_groupnorm_levels_bwd(const void* x, int ldx, long long x_bs, const void* dy, int lddy, long long dy_bs, void* dx,
                                         int lddx, long long dx_bs, const float* const* gamma, const float* const* beta,
                                         const float* mean, const float* rstd, float* const* dgamma, float* const* dbeta,
                                         const int* level_start, const int* level_hw, int L, int N, int C, int G, int gelu, double* stat_ws,
                                         int dtype, void* stream) {
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  EMRT_REQUIRE(x && dy && dx && gamma && beta && mean && rstd && level_start && level_hw, "null pointer");
  EMRT_REQUIRE(G > 0 && C % G == 0 && gn_use_fused(1, C, G), "unsupported C / G for the one-block-per-group kernel");
  GnLevels lv;
  EMRT_REQUIRE(gn_fill_levels(lv, level_start, level_hw, L, gamma, beta, dgamma, dbeta) == 0, "1..4 levels of at most 4096 rows");
  hipStream_t st = (hipStream_t)stream;
  if (stat_ws && !g_tune.gn_group_blocks && gn_rows_ok(C, G, ldx, lddy, lddx, x_bs, dy_bs, dx_bs)) {
    GnRows gs, ga;
    gs.lv = lv; ga.lv = lv;
    const unsigned bs = (unsigned)(gn_rows_plan(gs, g_tune.gn_bwd_stat_rows) * N), ba = (unsigned)(gn_rows_plan(ga, g_tune.gn_apply_rows) * N);
    DT_SWITCH(dtype,
              hipLaunchKernelGGL((gn_rows_bwd_stats_kernel<float>), dim3(bs), dim3(256), 0, st, (const float*)x, ldx, x_bs, (const float*)dy, lddy, dy_bs, gs, mean, rstd, stat_ws, N, G, gelu),
              hipLaunchKernelGGL((gn_rows_bwd_stats_kernel<bf16_t>), dim3(bs), dim3(256), 0, st, (const bf16_t*)x, ldx, x_bs, (const bf16_t*)dy, lddy, dy_bs, gs, mean, rstd, stat_ws, N, G, gelu));
    DT_SWITCH(dtype,
              hipLaunchKernelGGL((gn_rows_bwd_dx_kernel<float>), dim3(ba), dim3(256), 0, st, (const float*)x, ldx, x_bs, (const float*)dy, lddy, dy_bs, (float*)dx, lddx, dx_bs, ga, mean, rstd, stat_ws, N, G, gelu),
              hipLaunchKernelGGL((gn_rows_bwd_dx_kernel<bf16_t>), dim3(ba), dim3(256), 0, st, (const bf16_t*)x, ldx, x_bs, (const bf16_t*)dy, lddy, dy_bs, (bf16_t*)dx, lddx, dx_bs, ga, mean, rstd, stat_ws, N, G, gelu));
    return check_launch("emrt_groupnorm_levels_bwd");
  }
  DT_SWITCH(dtype,
            hipLaunchKernelGGL((gn_levels_bwd_kernel<float>), dim3(L * N * G), dim3(256), 0, st, (const float*)x, ldx, x_bs, (const float*)dy, lddy, dy_bs, (float*)dx, lddx, dx_bs, lv, mean, rstd, N, C, G, gelu),
            hipLaunchKernelGGL((gn_levels_bwd_kernel<bf16_t>), dim3(L * N * G), dim3(256), 0, st, (const bf16_t*)x, ldx, x_bs, (const bf16_t*)dy, lddy, dy_bs, (bf16_t*)dx, lddx, dx_bs, lv, mean, rstd, N, C, G, gelu));
  return check_launch("emrt_groupnorm_levels_bwd");
}
