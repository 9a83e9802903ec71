// Fused softmax(Q K^T / sqrt(d)) V for the decoder's 110-query self-attention (gfx950).
//
// Replaces the reference's matmul -> scale -> softmax -> dropout -> matmul chain
// (EMRT_utils/layers.py:283-303): [B, 8, 110, 32] x [B, 8, 32, 110] -> [B, 8, 110, 110] -> [B, 8, 110, 32].
// 0.2 % of the model's FLOPs and launch-bound; the whole (batch, head) problem lives in LDS: one 128-thread block
// per (b, head), thread i owns query row i, K/V rows are LDS-broadcast reads.  MFMA is deliberately not used here:
// 110x110x32 per block is < 1 us of VALU work and the kernel is bound by its launch and its few global accesses.
// Probabilities (pre-dropout) are saved in fp32 for the backward pass.
#include "common.hpp"

using namespace emrt;

#define MHA_MAXL 128
#define MHA_D 32
#define MHA_P 36      /* LDS row pitch in floats: 16-byte aligned rows so that the broadcast row reads are ds_read_b128 */

struct MhaArgs {
  const void* q; const void* k; const void* v;   // row (b*L + i), column head*32 + d
  int ldq, ldk, ldv;
  void* o; int ldo;
  float* probs;                                   // [B][M][L][L]
  int B, M, L;
  float scale, pdrop;
  const unsigned long long* seed; unsigned salt;
  // backward
  const void* dout; int lddo;
  void* dq; void* dk; void* dv; int lddq, lddk, lddv;
};

template <class T>
__device__ __forceinline__ void load_row32(const T* p, float* dst) {
#pragma unroll
  for (int c = 0; c < MHA_D; c += 8) {
    float t[8];
    Vec8<T>::load(p + c, t);
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[c + e] = t[e];
  }
}
template <class T>
__device__ __forceinline__ void store_row32(T* p, const float* src) {
#pragma unroll
  for (int c = 0; c < MHA_D; c += 8) {
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = src[c + e];
    Vec8<T>::store(p + c, t);
  }
}


// dot of a register row with an LDS row / axpy of an LDS row into a register row, 4 floats per LDS read
__device__ __forceinline__ float dot_row(const float* __restrict__ r, const float* __restrict__ srow) {
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < MHA_D; d += 4) {
    const float4 k4 = *reinterpret_cast<const float4*>(srow + d);
    s = fmaf(r[d], k4.x, s); s = fmaf(r[d + 1], k4.y, s); s = fmaf(r[d + 2], k4.z, s); s = fmaf(r[d + 3], k4.w, s);
  }
  return s;
}
__device__ __forceinline__ void axpy_row(float w, const float* __restrict__ srow, float* __restrict__ acc) {
#pragma unroll
  for (int d = 0; d < MHA_D; d += 4) {
    const float4 v4 = *reinterpret_cast<const float4*>(srow + d);
    acc[d] = fmaf(w, v4.x, acc[d]); acc[d + 1] = fmaf(w, v4.y, acc[d + 1]); acc[d + 2] = fmaf(w, v4.z, acc[d + 2]); acc[d + 3] = fmaf(w, v4.w, acc[d + 3]);
  }
}

template <class T>
__global__ __launch_bounds__(MHA_MAXL) void mha_fwd_kernel(MhaArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int L = a.L;
  float* sK = sm;                       // [L][MHA_P]
  float* sV = sK + L * MHA_P;              // [L][MHA_P]
  float* sS = sV + L * MHA_P;              // [MHA_MAXL][L + 1]
  const int b = blockIdx.x / a.M, m = blockIdx.x % a.M;
  const int i = threadIdx.x;
  const long long r0 = (long long)b * L;
  float qi[MHA_D];
  if (i < L) {
    float t[MHA_D];
    load_row32<T>((const T*)a.k + (r0 + i) * a.ldk + m * MHA_D, t);
#pragma unroll
    for (int d = 0; d < MHA_D; ++d) sK[i * MHA_P + d] = t[d];
    load_row32<T>((const T*)a.v + (r0 + i) * a.ldv + m * MHA_D, t);
#pragma unroll
    for (int d = 0; d < MHA_D; ++d) sV[i * MHA_P + d] = t[d];
    load_row32<T>((const T*)a.q + (r0 + i) * a.ldq + m * MHA_D, qi);
  }
  __syncthreads();
  if (i >= L) return;
  float* Si = sS + i * (L + 1);
  float mx = -3.0e38f;
  for (int j = 0; j < L; ++j) {
    float s = dot_row(qi, sK + j * MHA_P);
    s *= a.scale;
    Si[j] = s;
    mx = fmaxf(mx, s);
  }
  float den = 0.f;
  for (int j = 0; j < L; ++j) { const float e = __expf(Si[j] - mx); Si[j] = e; den += e; }
  const float inv = 1.f / den;
  float out[MHA_D];
#pragma unroll
  for (int d = 0; d < MHA_D; ++d) out[d] = 0.f;
  float* pg = a.probs + (((long long)b * a.M + m) * L + i) * L;
  const unsigned long long seed = a.pdrop > 0.f ? *a.seed : 0ull;
  const float keep_scale = a.pdrop > 0.f ? 1.f / (1.f - a.pdrop) : 1.f;
  for (int j = 0; j < L; ++j) {
    float pj = Si[j] * inv;
    pg[j] = pj;
    if (a.pdrop > 0.f) {
      const unsigned long long idx = (((unsigned long long)b * a.M + m) * L + i) * L + j;
      pj = uniform01(seed, a.salt, idx) >= a.pdrop ? pj * keep_scale : 0.f;
    }
    axpy_row(pj, sV + j * MHA_P, out);
  }
  store_row32<T>((T*)a.o + (r0 + i) * a.ldo + m * MHA_D, out);
}

template <class T>
__global__ __launch_bounds__(MHA_MAXL) void mha_bwd_kernel(MhaArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int L = a.L;
  float* sK = sm;                 // [L][MHA_P]
  float* sV = sK + L * MHA_P;
  float* sQ = sV + L * MHA_P;
  float* sG = sQ + L * MHA_P;        // dout rows
  float* sD = sG + L * MHA_P;        // dS  [MHA_MAXL][L + 1]
  const int b = blockIdx.x / a.M, m = blockIdx.x % a.M;
  const int i = threadIdx.x;
  const long long r0 = (long long)b * L;
  float qi[MHA_D], gi[MHA_D];
  if (i < L) {
    float t[MHA_D];
    load_row32<T>((const T*)a.k + (r0 + i) * a.ldk + m * MHA_D, t);
#pragma unroll
    for (int d = 0; d < MHA_D; ++d) sK[i * MHA_P + d] = t[d];
    load_row32<T>((const T*)a.v + (r0 + i) * a.ldv + m * MHA_D, t);
#pragma unroll
    for (int d = 0; d < MHA_D; ++d) sV[i * MHA_P + d] = t[d];
    load_row32<T>((const T*)a.q + (r0 + i) * a.ldq + m * MHA_D, qi);
    load_row32<T>((const T*)a.dout + (r0 + i) * a.lddo + m * MHA_D, gi);
#pragma unroll
    for (int d = 0; d < MHA_D; ++d) { sQ[i * MHA_P + d] = qi[d]; sG[i * MHA_P + d] = gi[d]; }
  }
  __syncthreads();
  const float* pgb = a.probs + ((long long)b * a.M + m) * L * L;
  const unsigned long long seed = a.pdrop > 0.f ? *a.seed : 0ull;
  const float keep_scale = a.pdrop > 0.f ? 1.f / (1.f - a.pdrop) : 1.f;
  if (i < L) {
    float* Di = sD + i * (L + 1);
    const float* pi = pgb + (long long)i * L;
    float dot = 0.f;
    for (int j = 0; j < L; ++j) {
      float dp = dot_row(gi, sV + j * MHA_P);
      if (a.pdrop > 0.f) {
        const unsigned long long idx = (((unsigned long long)b * a.M + m) * L + i) * L + j;
        dp = uniform01(seed, a.salt, idx) >= a.pdrop ? dp * keep_scale : 0.f;
      }
      Di[j] = dp;
      dot = fmaf(dp, pi[j], dot);
    }
    float dq[MHA_D];
#pragma unroll
    for (int d = 0; d < MHA_D; ++d) dq[d] = 0.f;
    for (int j = 0; j < L; ++j) {
      const float ds = pi[j] * (Di[j] - dot);
      Di[j] = ds;
      axpy_row(ds, sK + j * MHA_P, dq);
    }
#pragma unroll
    for (int d = 0; d < MHA_D; ++d) dq[d] *= a.scale;
    store_row32<T>((T*)a.dq + (r0 + i) * a.lddq + m * MHA_D, dq);
  }
  __syncthreads();
  if (i < L) {
    const int j = i;  // this thread now owns key/value row j
    float dk[MHA_D], dv[MHA_D];
#pragma unroll
    for (int d = 0; d < MHA_D; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
    for (int r = 0; r < L; ++r) {
      const float ds = sD[r * (L + 1) + j];
      float pd = pgb[(long long)r * L + j];
      if (a.pdrop > 0.f) {
        const unsigned long long idx = (((unsigned long long)b * a.M + m) * L + r) * L + j;
        pd = uniform01(seed, a.salt, idx) >= a.pdrop ? pd * keep_scale : 0.f;
      }
      axpy_row(ds, sQ + r * MHA_P, dk);
      axpy_row(pd, sG + r * MHA_P, dv);
    }
#pragma unroll
    for (int d = 0; d < MHA_D; ++d) dk[d] *= a.scale;
    store_row32<T>((T*)a.dk + (r0 + j) * a.lddk + m * MHA_D, dk);
    store_row32<T>((T*)a.dv + (r0 + j) * a.lddv + m * MHA_D, dv);
  }
}

extern "C" int emrt_mha_fwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo, float* probs,
                            int B, int M, int L, int D, float scale, float pdrop, const unsigned long long* seed, unsigned salt,
                            int dtype, void* stream) {
  EMRT_REQUIRE(q && k && v && o && probs, "null pointer");
  EMRT_REQUIRE(D == MHA_D, "head dim must be 32");
  EMRT_REQUIRE(L >= 1 && L <= MHA_MAXL, "sequence length must be <= 128");
  EMRT_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0, "row strides must be multiples of 8");
  EMRT_REQUIRE(pdrop == 0.f || seed, "dropout needs a device seed");
  MhaArgs a;
  memset(&a, 0, sizeof(a));
  a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.o = o; a.ldo = ldo; a.probs = probs;
  a.B = B; a.M = M; a.L = L; a.scale = scale; a.pdrop = pdrop; a.seed = seed; a.salt = salt;
  const size_t lds = (size_t)(2 * L * MHA_P + MHA_MAXL * (L + 1)) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  static bool attr_done = false;   // > 64 KiB of dynamic LDS needs the opt-in attribute (gfx950 has 160 KiB per CU)
  if (!attr_done) {
    hipFuncSetAttribute((const void*)mha_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)mha_fwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  if (dtype == EMRT_F32) hipLaunchKernelGGL((mha_fwd_kernel<float>), dim3(B * M), dim3(MHA_MAXL), lds, st, a);
  else hipLaunchKernelGGL((mha_fwd_kernel<bf16_t>), dim3(B * M), dim3(MHA_MAXL), lds, st, a);
  return check_launch("emrt_mha_fwd");
}

extern "C" int emrt_mha_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const float* probs,
                            const void* dout, int lddo, void* dq, int lddq, void* dk, int lddk, void* dv, int lddv, int B, int M,
                            int L, int D, float scale, float pdrop, const unsigned long long* seed, unsigned salt, int dtype,
                            void* stream) {
  EMRT_REQUIRE(q && k && v && probs && dout && dq && dk && dv, "null pointer");
  EMRT_REQUIRE(D == MHA_D, "head dim must be 32");
  EMRT_REQUIRE(L >= 1 && L <= MHA_MAXL, "sequence length must be <= 128");
  EMRT_REQUIRE(pdrop == 0.f || seed, "dropout needs a device seed");
  MhaArgs a;
  memset(&a, 0, sizeof(a));
  a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.probs = const_cast<float*>(probs);
  a.B = B; a.M = M; a.L = L; a.scale = scale; a.pdrop = pdrop; a.seed = seed; a.salt = salt;
  a.dout = dout; a.lddo = lddo; a.dq = dq; a.dk = dk; a.dv = dv; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
  const size_t lds = (size_t)(4 * L * MHA_P + MHA_MAXL * (L + 1)) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)mha_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)mha_bwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  if (dtype == EMRT_F32) hipLaunchKernelGGL((mha_bwd_kernel<float>), dim3(B * M), dim3(MHA_MAXL), lds, st, a);
  else hipLaunchKernelGGL((mha_bwd_kernel<bf16_t>), dim3(B * M), dim3(MHA_MAXL), lds, st, a);
  return check_launch("emrt_mha_bwd");
}
