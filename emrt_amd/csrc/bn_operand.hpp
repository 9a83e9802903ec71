// BatchNorm as an OPERAND TRANSFORM of a streaming consumer (internal).  In training the producing conv leaves the raw pre-BatchNorm map
// and the fp64 batch sums; a consumer that only streams the map once (x2 bilinear resize, 3x3 max-pool) applies
//      a = relu(x * scale + shift),   scale = invstd * gamma,  shift = beta - mean * scale
// to every element it loads, so the normalised map is never written or re-read and the emrt_bn_apply launch disappears
// (reference: nn.SyncBatchNorm -> ReLU -> F.interpolate in paddle_EMRT.py:164-175, BatchNorm2D -> ReLU -> MaxPool2D in
// paddle_vision_resnet.py:199-201 and paddle_EMRT.py:84-91).  The backward kernels re-derive the ReLU mask from the same expression
// (bn_scale_shift + one fmaf), so forward and backward agree on every element by construction.
#pragma once
#include "common.hpp"

namespace emrt {

// BatchNorm sums are accumulated into 8 replicas [8][2C] (fewer adders per address); consumers add them up.
#define BN_REPLICAS 8
__device__ __forceinline__ double rep_sum(const double* __restrict__ sums, int C, int idx) {
  double t = 0.0;
#pragma unroll
  for (int r = 0; r < BN_REPLICAS; ++r) t += sums[(long long)r * 2 * C + idx];
  return t;
}

// Per-channel constants of one BatchNorm layer, from fp64 sums (training) or the running statistics (eval).
struct BnChan { float mean, invstd; };
__device__ __forceinline__ BnChan bn_chan(const double* __restrict__ sums, const float* __restrict__ run_mean,
                                          const float* __restrict__ run_var, int C, int c, double inv_count, float eps) {
  BnChan o;
  if (sums) {
    const double mu = rep_sum(sums, C, c) * inv_count;
    double var = rep_sum(sums, C, C + c) * inv_count - mu * mu;
    if (var < 0.0) var = 0.0;
    o.mean = (float)mu;
    o.invstd = (float)(1.0 / sqrt(var + (double)eps));
  } else {
    o.mean = run_mean[c];
    o.invstd = rsqrtf(run_var[c] + eps);
  }
  return o;
}

// THE expression of the affine form: every kernel that applies BatchNorm or re-derives its ReLU mask goes through here
__device__ __forceinline__ void bn_scale_shift(float mean, float invstd, float gamma, float beta, float& scale, float& shift) {
  scale = invstd * gamma;
  shift = fmaf(-mean, scale, beta);
}

// Training-mode BatchNorm of a consumer's input operand (kernel argument, plain pointers)
struct BnOperand {
  const double* sums;      // [8][2C] fp64 (sum x, sum x^2) of the raw map, complete when the consumer starts
  double inv_count;        // 1 / rows the sums cover
  float eps, momentum;
  float* mean;             // [C] out: saved for backward   (written by block 0)
  float* invstd;           // [C] out
  float* run_mean;         // [C] in/out or null: running statistics update (block 0)
  float* run_var;
  const float* gamma;
  const float* beta;
  int relu;
};

// All threads of the block: lds[0..C) = scale, lds[C..2C) = shift; block `first` also saves mean / invstd and updates the running
// statistics (Paddle convention: running = momentum * running + (1 - momentum) * batch, biased variance).  Ends with a barrier.
__device__ __forceinline__ void bn_operand_preamble(const BnOperand& b, int C, float* lds, bool first) {
  for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {
    const BnChan k = bn_chan(b.sums, nullptr, nullptr, C, ch, b.inv_count, b.eps);
    float sc, sh;
    bn_scale_shift(k.mean, k.invstd, b.gamma[ch], b.beta[ch], sc, sh);
    lds[ch] = sc;
    lds[C + ch] = sh;
    if (first) {
      b.mean[ch] = k.mean;
      b.invstd[ch] = k.invstd;
      if (b.run_mean) {
        const double mu = rep_sum(b.sums, C, ch) * b.inv_count;
        double var = rep_sum(b.sums, C, C + ch) * b.inv_count - mu * mu;
        if (var < 0.0) var = 0.0;
        b.run_mean[ch] = b.momentum * b.run_mean[ch] + (1.f - b.momentum) * (float)mu;
        b.run_var[ch] = b.momentum * b.run_var[ch] + (1.f - b.momentum) * (float)var;
      }
    }
  }
  __syncthreads();
}

}  // namespace emrt
