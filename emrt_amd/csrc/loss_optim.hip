// Loss, optimizer and weight-packing kernels for gfx950 (all HBM-bound streaming kernels).
//
//  * softmax cross-entropy with ignore_index over fp32 NCHW logits (reference:
//    losses/mix_softmax_cross_entropy_loss.py:27-35 -> paddle nn.CrossEntropyLoss(ignore_index=255, axis=1))
//  * ClipGradByGlobalNorm + L2 decay + Momentum over ONE flat fp32 parameter buffer
//    (reference: solver/optimizer.py:29-40; PolynomialDecay lr_scheduler.py:244-248 evaluated on the device from a
//    step counter so the whole step can live in one hipGraph)
//  * per-step weight packing: fp32 master [OC][taps][C] -> compute-dtype forward copy [OC][taps][C] and transposed
//    dgrad copy [C][taps][OC] (LDS 32x32 tile transpose, one launch for every GEMM weight of the model).
#include "common.hpp"

using namespace emrt;

// ------------------------------------------------------------------------------------------------
// cross entropy
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, const long long* __restrict__ labels, int N,
                                                     int C, long long HW, int ignore_index, float* __restrict__ partial) {
  __shared__ float red[2 * 4];
  const long long total = (long long)N * HW;
  float ls = 0.f, cnt = 0.f;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long lab = labels[idx];
    if (lab == ignore_index) continue;
    const long long n = total <= 0xffffffffll ? (long long)((unsigned)idx / (unsigned)HW) : idx / HW, p = idx - n * HW;      // (32-bit division when it can be)
    const float* lp = logits + n * C * HW + p;
    float mx = -3.0e38f;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, lp[c * HW]);
    float den = 0.f;
    for (int c = 0; c < C; ++c) den += __expf(lp[c * HW] - mx);
    const float picked = (lab >= 0 && lab < C) ? lp[lab * HW] : 0.f;
    ls += logf(den) + mx - picked;
    cnt += 1.f;
  }
  ls = wave_sum(ls);
  cnt = wave_sum(cnt);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { red[wv * 2] = ls; red[wv * 2 + 1] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f;
    for (int w = 0; w < 4; ++w) { a += red[w * 2]; b += red[w * 2 + 1]; }
    partial[blockIdx.x * 2] = a;
    partial[blockIdx.x * 2 + 1] = b;
  }
}

// result[0] = mean loss over non-ignored pixels, result[1] = count
// one block: fp64 tree reduction of the per-block partial (loss, count) pairs (a single-thread loop over up to 4096
// partials took 70 us of dependent loads)
__global__ __launch_bounds__(256) void ce_finalize_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ result) {
  __shared__ double ra[256], rb[256];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) { a += partial[i * 2]; b += partial[i * 2 + 1]; }
  ra[threadIdx.x] = a;
  rb[threadIdx.x] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { ra[threadIdx.x] += ra[threadIdx.x + o]; rb[threadIdx.x] += rb[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    // Paddle's cross_entropy divides by count + (count == 0): a batch whose pixels are all ignore_index gives loss 0, not 0/0
    result[0] = (float)(ra[0] / (rb[0] > 0.0 ? rb[0] : 1.0));
    result[1] = (float)rb[0];
  }
}

// dlogits = weight * upstream * (softmax - onehot) / count     (upstream: device scalar or null == 1)
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                                                     const float* __restrict__ result, const float* __restrict__ upstream,
                                                     float weight, int N, int C, long long HW, int ignore_index,
                                                     float* __restrict__ dlogits) {
  const long long total = (long long)N * HW;
  const float g = weight * (upstream ? upstream[0] : 1.f) / fmaxf(result[1], 1.f);      // (all-ignored batch: zero gradient)
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long lab = labels[idx];
    const long long n = total <= 0xffffffffll ? (long long)((unsigned)idx / (unsigned)HW) : idx / HW, p = idx - n * HW;      // (32-bit division when it can be)
    const float* lp = logits + n * C * HW + p;
    float* dp = dlogits + n * C * HW + p;
    if (lab == ignore_index) {
      for (int c = 0; c < C; ++c) dp[c * HW] = 0.f;
      continue;
    }
    float mx = -3.0e38f;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, lp[c * HW]);
    float den = 0.f;
    for (int c = 0; c < C; ++c) den += __expf(lp[c * HW] - mx);
    const float inv = 1.f / den;
    for (int c = 0; c < C; ++c) dp[c * HW] = g * (__expf(lp[c * HW] - mx) * inv - (c == lab ? 1.f : 0.f));
  }
}

// ---- the two heads of MixSoftmaxCrossEntropyLoss in one pass (mix_softmax_cross_entropy_loss.py:29-35,44-51: CE(main) + 0.4 CE(aux) on the SAME
// labels): forward = one streaming launch over both logit tensors (labels read once) + one finalize launch that also forms the weighted total
// (was: 2 x (stream + finalize) + a scalar axpby = 5 launches, each at the ~4.6 us floor of a dependent kernel in the step's graph); backward = one
// launch writing both gradients.  Per-pixel arithmetic and the fp64 finalize are those of the single-head kernels: bit-identical results.
__device__ __forceinline__ float ce_pixel(const float* lp, int C, long long HW, long long lab) {
  float mx = -3.0e38f;
  for (int c = 0; c < C; ++c) mx = fmaxf(mx, lp[c * HW]);
  float den = 0.f;
  for (int c = 0; c < C; ++c) den += __expf(lp[c * HW] - mx);
  const float picked = (lab >= 0 && lab < C) ? lp[lab * HW] : 0.f;
  return logf(den) + mx - picked;
}
__global__ __launch_bounds__(256) void ce_pair_fwd_kernel(const float* __restrict__ la, const float* __restrict__ lb, const long long* __restrict__ labels,
                                                          int N, int C, long long HW, int ignore_index, float* __restrict__ partial) {
  __shared__ float red[3 * 4];
  const long long total = (long long)N * HW;
  float sa = 0.f, sb = 0.f, cnt = 0.f;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long lab = labels[idx];
    if (lab == ignore_index) continue;
    const long long n = total <= 0xffffffffll ? (long long)((unsigned)idx / (unsigned)HW) : idx / HW, p = idx - n * HW;
    sa += ce_pixel(la + n * C * HW + p, C, HW, lab);
    sb += ce_pixel(lb + n * C * HW + p, C, HW, lab);
    cnt += 1.f;
  }
  sa = wave_sum(sa); sb = wave_sum(sb); cnt = wave_sum(cnt);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { red[wv * 3] = sa; red[wv * 3 + 1] = sb; red[wv * 3 + 2] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f, c = 0.f;
    for (int w = 0; w < 4; ++w) { a += red[w * 3]; b += red[w * 3 + 1]; c += red[w * 3 + 2]; }
    partial[blockIdx.x * 3] = a; partial[blockIdx.x * 3 + 1] = b; partial[blockIdx.x * 3 + 2] = c;
  }
}
__global__ __launch_bounds__(256) void ce_pair_finalize_kernel(const float* __restrict__ partial, int nblk, float wa, float wb, float* __restrict__ res_a,
                                                               float* __restrict__ res_b, float* __restrict__ total) {
  __shared__ double ra[256], rb[256], rc[256];
  double a = 0.0, b = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) { a += partial[i * 3]; b += partial[i * 3 + 1]; c += partial[i * 3 + 2]; }
  ra[threadIdx.x] = a; rb[threadIdx.x] = b; rc[threadIdx.x] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { ra[threadIdx.x] += ra[threadIdx.x + o]; rb[threadIdx.x] += rb[threadIdx.x + o]; rc[threadIdx.x] += rc[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double den = rc[0] > 0.0 ? rc[0] : 1.0;
    res_a[0] = (float)(ra[0] / den); res_a[1] = (float)rc[0];
    res_b[0] = (float)(rb[0] / den); res_b[1] = (float)rc[0];
    total[0] = wa * res_a[0] + wb * res_b[0];
  }
}
__global__ __launch_bounds__(256) void ce_pair_bwd_kernel(const float* __restrict__ la, const float* __restrict__ lb, const long long* __restrict__ labels,
                                                          const float* __restrict__ res_a, const float* __restrict__ up_a, const float* __restrict__ up_b,
                                                          float wa, float wb, int N, int C, long long HW, int ignore_index, float* __restrict__ da,
                                                          float* __restrict__ db) {
  const long long total = (long long)N * HW;
  const float inv_cnt = 1.f / fmaxf(res_a[1], 1.f);
  const float ga = wa * (up_a ? up_a[0] : 1.f) * inv_cnt, gb = wb * (up_b ? up_b[0] : 1.f) * inv_cnt;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long lab = labels[idx];
    const long long n = total <= 0xffffffffll ? (long long)((unsigned)idx / (unsigned)HW) : idx / HW, p = idx - n * HW;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float* lp = (h ? lb : la) + n * C * HW + p;
      float* dp = (h ? db : da) + n * C * HW + p;
      const float g = h ? gb : ga;
      if (lab == ignore_index) {
        for (int c = 0; c < C; ++c) dp[c * HW] = 0.f;
        continue;
      }
      float mx = -3.0e38f;
      for (int c = 0; c < C; ++c) mx = fmaxf(mx, lp[c * HW]);
      float den = 0.f;
      for (int c = 0; c < C; ++c) den += __expf(lp[c * HW] - mx);
      const float inv = 1.f / den;
      for (int c = 0; c < C; ++c) dp[c * HW] = g * (__expf(lp[c * HW] - mx) * inv - (c == lab ? 1.f : 0.f));
    }
  }
}

__global__ void axpby_scalar_kernel(float* out, const float* a, float wa, const float* b, float wb) {
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = wa * a[0] + (b ? wb * b[0] : 0.f);
}

extern "C" size_t emrt_ce_workspace_bytes(void) { return 1024 * 3 * sizeof(float); }

// result[2] (device): {mean loss, non-ignored count}
extern "C" int emrt_softmax_ce_fwd(const float* logits, const long long* labels, int N, int C, int H, int W, int ignore_index,
                                   float* result, void* workspace, void* stream) {
  EMRT_REQUIRE(logits && labels && result && workspace, "null pointer");
  const long long total = (long long)N * H * W;
  int grid = (int)((total + 255) / 256);
  if (grid > 1024) grid = 1024;
  if (grid < 1) grid = 1;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(grid), dim3(256), 0, st, logits, labels, N, C, (long long)H * W, ignore_index, (float*)workspace);
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, grid, result);
  return check_launch("emrt_softmax_ce_fwd");
}

// both heads at once: res_a / res_b (device float[2] each: {mean loss, non-ignored count}), total[0] = wa * loss_a + wb * loss_b.
// workspace: emrt_ce_workspace_bytes().
extern "C" int emrt_softmax_ce_pair_fwd(const float* logits_a, const float* logits_b, const long long* labels, int N, int C, int H, int W,
                                        int ignore_index, float wa, float wb, float* res_a, float* res_b, float* total, void* workspace, void* stream) {
  EMRT_REQUIRE(logits_a && logits_b && labels && res_a && res_b && total && workspace, "null pointer");
  const long long npix = (long long)N * H * W;
  int grid = (int)((npix + 255) / 256);
  if (grid > 1024) grid = 1024;
  if (grid < 1) grid = 1;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_pair_fwd_kernel, dim3(grid), dim3(256), 0, st, logits_a, logits_b, labels, N, C, (long long)H * W, ignore_index, (float*)workspace);
  hipLaunchKernelGGL(ce_pair_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, grid, wa, wb, res_a, res_b, total);
  return check_launch("emrt_softmax_ce_pair_fwd");
}

// d logits_a = wa * up_a * (softmax - onehot) / count, d logits_b likewise (up_*: device scalars or NULL == 1); res_a from the forward.
extern "C" int emrt_softmax_ce_pair_bwd(const float* logits_a, const float* logits_b, const long long* labels, const float* res_a, const float* up_a,
                                        const float* up_b, float wa, float wb, int N, int C, int H, int W, int ignore_index, float* dlogits_a,
                                        float* dlogits_b, void* stream) {
  EMRT_REQUIRE(logits_a && logits_b && labels && res_a && dlogits_a && dlogits_b, "null pointer");
  const long long npix = (long long)N * H * W;
  int grid = (int)((npix + 255) / 256);
  if (grid > 4096) grid = 4096;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(ce_pair_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits_a, logits_b, labels, res_a, up_a, up_b, wa, wb, N, C,
                     (long long)H * W, ignore_index, dlogits_a, dlogits_b);
  return check_launch("emrt_softmax_ce_pair_bwd");
}

extern "C" int emrt_softmax_ce_bwd(const float* logits, const long long* labels, const float* result, const float* upstream,
                                   float weight, int N, int C, int H, int W, int ignore_index, float* dlogits, void* stream) {
  EMRT_REQUIRE(logits && labels && result && dlogits, "null pointer");
  const long long total = (long long)N * H * W;
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, labels, result, upstream, weight, N, C,
                     (long long)H * W, ignore_index, dlogits);
  return check_launch("emrt_softmax_ce_bwd");
}

extern "C" int emrt_scalar_axpby(float* out, const float* a, float wa, const float* b, float wb, void* stream) {
  EMRT_REQUIRE(out && a, "null pointer");
  hipLaunchKernelGGL(axpby_scalar_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out, a, wa, b, wb);
  return check_launch("emrt_scalar_axpby");
}

// ------------------------------------------------------------------------------------------------
// optimizer
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, long long n, float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  const long long n4 = n / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n - n4 * 4)) { const float v = g[n4 * 4 + threadIdx.x]; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// state[0] = clip scale (clip / max(norm, clip), or 1 when clip <= 0), state[1] = global grad norm
__global__ __launch_bounds__(256) void clip_scale_kernel(const float* __restrict__ partial, int nblk, float clip, float* __restrict__ state) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0]);
    state[1] = norm;
    state[0] = clip > 0.f ? clip / fmaxf(norm, clip) : 1.f;
  }
}

struct SgdArgs {
  float* p; const float* g; float* v;
  long long n;
  const float* state;            // clip scale at [0]
  const long long* step;         // device step counter (0-based index of this step)
  float base_lr, end_lr, power; long long decay_steps;
  float momentum, weight_decay;
  int nranges;
  long long r0[32], r1[32];      // element ranges whose learning rate is multiplied by `range_mult`
  float range_mult;
  float* lr_out;                 // optional: lr used this step
  void* mirror;                  // optional: compute-dtype copy of the parameters, same indexing as p
};

template <class MT, bool NT>
__global__ __launch_bounds__(256) void sgd_momentum_kernel(SgdArgs a) {
  long long t = a.step ? a.step[0] : 0;
  if (t > a.decay_steps) t = a.decay_steps;
  const float frac = 1.f - (float)((double)t / (double)a.decay_steps);
  const float lr = (a.base_lr - a.end_lr) * powf(frac, a.power) + a.end_lr;
  if (a.lr_out && blockIdx.x == 0 && threadIdx.x == 0) a.lr_out[0] = lr;
  const float scale = a.state ? a.state[0] : 1.f;
  MT* mirror = (MT*)a.mirror;
  // four elements per thread (16-byte accesses on the three fp32 streams); the compute-dtype mirror of the parameters -- the
  // forward GEMM operand, same index as the master copy -- is written here instead of by a second pass over the master buffer
  const long long n4 = a.n / 4;
  for (long long i4 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i4 < n4; i4 += (long long)gridDim.x * blockDim.x) {
    const long long i = i4 * 4;
    // master copy, velocity and gradient are touched once per step (1.1 GB at 56 M parameters): non-temporal, so that they do not push the
    // compute-dtype mirror written below -- the next forward's GEMM operand -- out of the last-level cache (knob sgd_nt, default on)
    typedef __attribute__((ext_vector_type(4))) float sgd_f32x4;
    sgd_f32x4 pq, gq, vq;
    if (NT) {
      pq = __builtin_nontemporal_load(reinterpret_cast<const sgd_f32x4*>(a.p) + i4);
      gq = __builtin_nontemporal_load(reinterpret_cast<const sgd_f32x4*>(a.g) + i4);
      vq = __builtin_nontemporal_load(reinterpret_cast<const sgd_f32x4*>(a.v) + i4);
    } else {
      pq = reinterpret_cast<const sgd_f32x4*>(a.p)[i4];
      gq = reinterpret_cast<const sgd_f32x4*>(a.g)[i4];
      vq = reinterpret_cast<const sgd_f32x4*>(a.v)[i4];
    }
    const float4 p4 = make_float4(pq[0], pq[1], pq[2], pq[3]), g4 = make_float4(gq[0], gq[1], gq[2], gq[3]), v4 = make_float4(vq[0], vq[1], vq[2], vq[3]);
    float p[4] = {p4.x, p4.y, p4.z, p4.w}, v[4] = {v4.x, v4.y, v4.z, v4.w};
    const float g[4] = {g4.x, g4.y, g4.z, g4.w};
    bool any = false;
    for (int r = 0; r < a.nranges; ++r) any |= (i + 3 >= a.r0[r] && i < a.r1[r]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float mult = 1.f;
      if (any)
        for (int r = 0; r < a.nranges; ++r)
          if (i + e >= a.r0[r] && i + e < a.r1[r]) mult = a.range_mult;
      // explicit fused multiply-adds: the contraction is then the same in every instantiation of this kernel (left to the compiler, the
      // non-temporal and the plain variant differed in the last bit on ~1 % of the elements)
      const float gg = fmaf(a.weight_decay, p[e], g[e] * scale);
      v[e] = fmaf(a.momentum, v[e], gg);
      p[e] = fmaf(-(lr * mult), v[e], p[e]);
    }
    if (NT) {
      __builtin_nontemporal_store((sgd_f32x4){v[0], v[1], v[2], v[3]}, reinterpret_cast<sgd_f32x4*>(a.v) + i4);
      __builtin_nontemporal_store((sgd_f32x4){p[0], p[1], p[2], p[3]}, reinterpret_cast<sgd_f32x4*>(a.p) + i4);
    } else {
      reinterpret_cast<float4*>(a.v)[i4] = make_float4(v[0], v[1], v[2], v[3]);
      reinterpret_cast<float4*>(a.p)[i4] = make_float4(p[0], p[1], p[2], p[3]);
    }
    if (mirror) Vec4<MT>::store(mirror + i, p);
  }
  if (blockIdx.x == 0) {
    for (long long i = n4 * 4 + threadIdx.x; i < a.n; i += blockDim.x) {
      float mult = 1.f;
      for (int r = 0; r < a.nranges; ++r)
        if (i >= a.r0[r] && i < a.r1[r]) mult = a.range_mult;
      const float p = a.p[i];
      const float g = fmaf(a.weight_decay, p, a.g[i] * scale);
      const float v = fmaf(a.momentum, a.v[i], g);
      a.v[i] = v;
      const float pn = fmaf(-(lr * mult), v, p);
      a.p[i] = pn;
      if (mirror) mirror[i] = from_f32<MT>(pn);
    }
  }
}

__global__ void counter_add_kernel(long long* c, long long d) {
  if (threadIdx.x == 0 && blockIdx.x == 0) c[0] += d;
}

// Per-class areas of a prediction against its labels (reference: src/utils/metrics.py:20-59 calculate_area): intersect / prediction / label pixel counts with
// ignore_index, accumulated into out[3][ncls] (int64).  Per-block LDS histograms, one global atomic per (block, class, kind).
template <class LT>
__global__ __launch_bounds__(256) void seg_areas_kernel(const int* __restrict__ pred, const LT* __restrict__ label, long long n, int ncls, int ignore,
                                                        unsigned long long* __restrict__ out) {
  __shared__ unsigned cnt[3 * 256];
  for (int i = threadIdx.x; i < 3 * ncls; i += blockDim.x) cnt[i] = 0;
  __syncthreads();
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long l = (long long)label[i];
    if (l == ignore) continue;
    const int p = pred[i];
    const bool pv = p >= 0 && p < ncls, lv = l >= 0 && l < ncls;
    if (pv) atomicAdd(&cnt[ncls + p], 1u);
    if (lv) atomicAdd(&cnt[2 * ncls + (int)l], 1u);
    if (pv && (long long)p == l) atomicAdd(&cnt[p], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * ncls; i += blockDim.x)
    if (cnt[i]) atomicAdd(&out[i], (unsigned long long)cnt[i]);
}

extern "C" int emrt_segmentation_areas(const int* pred, const void* label, int label_is_int64, long long n, int num_classes, int ignore_index,
                                       long long* out, void* stream) {
  EMRT_REQUIRE(pred && label && out, "null pointer");
  EMRT_REQUIRE(num_classes >= 1 && num_classes <= 256 && n >= 0, "1..256 classes");
  if (n == 0) return 0;
  long long grid = (n + 256 * 16 - 1) / (256 * 16);
  if (grid > 1024) grid = 1024;
  hipStream_t st = (hipStream_t)stream;
  if (label_is_int64) hipLaunchKernelGGL((seg_areas_kernel<long long>), dim3((unsigned)grid), dim3(256), 0, st, pred, (const long long*)label, n, num_classes, ignore_index, (unsigned long long*)out);
  else hipLaunchKernelGGL((seg_areas_kernel<int>), dim3((unsigned)grid), dim3(256), 0, st, pred, (const int*)label, n, num_classes, ignore_index, (unsigned long long*)out);
  return check_launch("emrt_segmentation_areas");
}

extern "C" size_t emrt_gradnorm_workspace_bytes(void) { return 2048 * sizeof(float); }

extern "C" int emrt_grad_clip_scale(const float* grads, long long n, float clip, float* state /*[2]*/, void* workspace, void* stream) {
  EMRT_REQUIRE(grads && state && workspace, "null pointer");
  EMRT_REQUIRE(((uintptr_t)grads) % 16 == 0, "grads must be 16-byte aligned");
  int grid = (int)((n / 4 + 255) / 256);
  if (grid > 2048) grid = 2048;
  if (grid < 1) grid = 1;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(grid), dim3(256), 0, st, grads, n, (float*)workspace);
  hipLaunchKernelGGL(clip_scale_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, grid, clip, state);
  return check_launch("emrt_grad_clip_scale");
}

extern "C" int emrt_sgd_momentum_step(float* params, const float* grads, float* velocity, long long n, const float* clip_state,
                                      const long long* step, float base_lr, float end_lr, float power, long long decay_steps,
                                      float momentum, float weight_decay, const long long* ranges /*host [nranges][2]*/,
                                      int nranges, float range_mult, float* lr_out, void* mirror, int mirror_dtype, void* stream) {
  EMRT_REQUIRE(params && grads && velocity, "null pointer");
  EMRT_REQUIRE(!mirror || mirror_dtype == EMRT_BF16 || mirror_dtype == EMRT_F16, "the parameter mirror is bf16 or fp16");
  EMRT_REQUIRE(((uintptr_t)params | (uintptr_t)grads | (uintptr_t)velocity) % 16 == 0 && (!mirror || (uintptr_t)mirror % 8 == 0), "buffers must be 16-byte aligned");
  EMRT_REQUIRE(nranges >= 0 && nranges <= 32 && (nranges == 0 || ranges), "0..32 lr-mult ranges");
  EMRT_REQUIRE(decay_steps > 0, "decay_steps must be positive");
  SgdArgs a;
  memset(&a, 0, sizeof(a));
  a.p = params; a.g = grads; a.v = velocity; a.n = n; a.state = clip_state; a.step = step;
  a.base_lr = base_lr; a.end_lr = end_lr; a.power = power; a.decay_steps = decay_steps;
  a.momentum = momentum; a.weight_decay = weight_decay; a.nranges = nranges; a.range_mult = range_mult; a.lr_out = lr_out;
  a.mirror = mirror;
  for (int r = 0; r < nranges; ++r) { a.r0[r] = ranges[2 * r]; a.r1[r] = ranges[2 * r + 1]; }
  int grid = (int)((n / 4 + 255) / 256);
  if (grid > 8192) grid = 8192;
  if (grid < 1) grid = 1;
  const bool nt = g_tune.sgd_nt != 0;
  if (mirror && mirror_dtype == EMRT_F16) {
    if (nt) hipLaunchKernelGGL((sgd_momentum_kernel<f16_t, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((sgd_momentum_kernel<f16_t, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  } else {
    if (nt) hipLaunchKernelGGL((sgd_momentum_kernel<bf16_t, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((sgd_momentum_kernel<bf16_t, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  }
  return check_launch("emrt_sgd_momentum_step");
}

extern "C" int emrt_counter_add(long long* counter, long long delta, void* stream) {
  EMRT_REQUIRE(counter, "null pointer");
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter, delta);
  return check_launch("emrt_counter_add");
}

// ------------------------------------------------------------------------------------------------
// weight packing.  Descriptor table (device, int64 x 8 per entry):
//   {src_off, fwd_off (-1: none), bwd_off (-1: none), OC, taps, C, tile_prefix (first flat tile id), unused}
// One 32x32 (oc x c) tile per block per tap; flat tile id -> descriptor by binary search on tile_prefix.
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ master, T* __restrict__ packed,
                                                           const long long* __restrict__ desc, int ndesc, int bwd_only) {
  __shared__ float tile[32][33];
  const long long tid = blockIdx.x;
  int lo = 0, hi = ndesc - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (desc[mid * 8 + 6] <= tid) lo = mid; else hi = mid - 1;
  }
  const long long* d = desc + lo * 8;
  const long long src = d[0], fo = bwd_only ? -1 : d[1], bo = d[2];
  if (fo < 0 && bo < 0) return;
  // bwd_only: the forward copy (the optimizer's mirror) is current -- transpose from it (half the bytes of the fp32 master)
  const T* mir = (bwd_only && d[1] >= 0) ? packed + d[1] : nullptr;
  const int OC = (int)d[3], taps = (int)d[4], C = (int)d[5];
  const int tc = (C + 31) / 32, toc = (OC + 31) / 32;
  long long local = tid - d[6];
  const int ct = (int)(local % tc); local /= tc;
  const int ot = (int)(local % toc);
  const int tap = (int)(local / toc);
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int oc = ot * 32 + ty + 8 * r, c = ct * 32 + tx;
    float v = 0.f;
    if (oc < OC && c < C) {
      const long long idx = ((long long)oc * taps + tap) * C + c;
      v = mir ? to_f32(mir[idx]) : master[src + idx];
      if (fo >= 0) packed[fo + idx] = from_f32<T>(v);
    }
    tile[ty + 8 * r][tx] = v;
  }
  if (bo < 0) return;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = ct * 32 + ty + 8 * r, oc = ot * 32 + tx;
    if (oc < OC && c < C) packed[bo + ((long long)c * taps + tap) * OC + oc] = from_f32<T>(tile[tx][ty + 8 * r]);
  }
}

// The per-step form: only the transposed dgrad copies, 64x64 (oc x c) tiles, 16-byte accesses on both sides (the 32x32 kernel above
// writes 2-byte elements in 64-byte runs: 1.8 TB/s).  Source = the compute-dtype forward copy when the descriptor has one (the
// optimizer's mirror, current by construction), else the fp32 master.  desc[7] = first 64x64 tile id of the descriptor.
template <class T>
__global__ __launch_bounds__(256) void pack_bwd64_kernel(const float* __restrict__ master, T* __restrict__ packed,
                                                         const long long* __restrict__ desc, int ndesc) {
  __shared__ float tile[64][65];
  const long long tid = blockIdx.x;
  int lo = 0, hi = ndesc - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (desc[mid * 8 + 7] <= tid) lo = mid; else hi = mid - 1;
  }
  const long long* d = desc + lo * 8;
  const long long src = d[0], fo = d[1], bo = d[2];
  if (bo < 0) return;
  const int OC = (int)d[3], taps = (int)d[4], C = (int)d[5];
  const int tc = (C + 63) / 64, toc = (OC + 63) / 64;
  long long local = tid - d[7];
  const int ct = (int)(local % tc); local /= tc;
  const int ot = (int)(local % toc);
  const int tap = (int)(local / toc);
  if (tap >= taps) return;
  const T* mir = fo >= 0 ? packed + fo : nullptr;
  const int sub = threadIdx.x & 7, rw = threadIdx.x >> 3;      // 8 chunks of 8 elements x 32 rows per pass
  const bool vc = (C % 8) == 0, voc = (OC % 8) == 0;
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int ol = rw + 32 * ps, oc = ot * 64 + ol, c0 = ct * 64 + sub * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (oc < OC && c0 < C) {
      const long long idx = ((long long)oc * taps + tap) * C + c0;
      if (vc) {
        if (mir) Vec8<T>::load(mir + idx, v);
        else Vec8<float>::load(master + src + idx, v);
      } else {
        for (int e = 0; e < 8; ++e)
          if (c0 + e < C) v[e] = mir ? to_f32(mir[idx + e]) : master[src + idx + e];
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[ol][sub * 8 + e] = v[e];
  }
  __syncthreads();
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int cl = rw + 32 * ps, c = ct * 64 + cl, o0 = ot * 64 + sub * 8;
    if (c >= C || o0 >= OC) continue;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[sub * 8 + e][cl];
    T* dst = packed + bo + ((long long)c * taps + tap) * OC + o0;
    if (voc) Vec8<T>::store(dst, v);
    else
      for (int e = 0; e < 8; ++e)
        if (o0 + e < OC) dst[e] = from_f32<T>(v[e]);
  }
}

extern "C" int emrt_pack_weights(const float* master, void* packed, const long long* desc_dev, int ndesc, long long total_tiles,
                                 long long total_tiles64, int bwd_only, int dtype, void* stream) {
  EMRT_REQUIRE_FWD_DTYPE(dtype);
  EMRT_REQUIRE(master && packed && desc_dev, "null pointer");
  EMRT_REQUIRE(ndesc > 0 && total_tiles > 0 && total_tiles < 2147483647LL && total_tiles64 >= 0 && total_tiles64 < 2147483647LL, "bad descriptor table");
  hipStream_t st = (hipStream_t)stream;
  if (bwd_only && total_tiles64 > 0) {
    if (dtype == EMRT_F32) hipLaunchKernelGGL((pack_bwd64_kernel<float>), dim3((unsigned)total_tiles64), dim3(256), 0, st, master, (float*)packed, desc_dev, ndesc);
    else if (dtype == EMRT_BF16) hipLaunchKernelGGL((pack_bwd64_kernel<bf16_t>), dim3((unsigned)total_tiles64), dim3(256), 0, st, master, (bf16_t*)packed, desc_dev, ndesc);
    else hipLaunchKernelGGL((pack_bwd64_kernel<f16_t>), dim3((unsigned)total_tiles64), dim3(256), 0, st, master, (f16_t*)packed, desc_dev, ndesc);
    return check_launch("emrt_pack_weights");
  }
  if (dtype == EMRT_F32) hipLaunchKernelGGL((pack_weights_kernel<float>), dim3((unsigned)total_tiles), dim3(256), 0, st, master, (float*)packed, desc_dev, ndesc, bwd_only);
  else if (dtype == EMRT_BF16) hipLaunchKernelGGL((pack_weights_kernel<bf16_t>), dim3((unsigned)total_tiles), dim3(256), 0, st, master, (bf16_t*)packed, desc_dev, ndesc, bwd_only);
  else hipLaunchKernelGGL((pack_weights_kernel<f16_t>), dim3((unsigned)total_tiles), dim3(256), 0, st, master, (f16_t*)packed, desc_dev, ndesc, bwd_only);
  return check_launch("emrt_pack_weights");
}
