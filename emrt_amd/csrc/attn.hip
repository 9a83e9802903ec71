// Fused softmax(Q K^T / sqrt(d)) V for the decoder's 110-query self-attention (gfx950).
//
// Replaces the reference's matmul -> scale -> softmax -> dropout -> matmul chain
// (EMRT_utils/layers.py:283-303): [B, 8, 110, 32] x [B, 8, 32, 110] -> [B, 8, 110, 110] -> [B, 8, 110, 32].
// 0.2 % of the model's FLOPs and launch-bound; the whole (batch, head) problem lives in LDS: one 512-thread block
// per (b, head), 4 lanes per query row, K/V rows are LDS-broadcast reads.  MFMA is deliberately not used here:
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
  int use_sp;                                     // backward: dropped probabilities staged in LDS (fits up to L = 110) or re-derived from `probs`
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

// Thread layout of both kernels: 4 lanes per row (tid = 4 * row + part), 512 threads per (batch, head) block.
// Score-shaped phases give lane `part` the keys j = part, part + 4, ...; output-shaped phases give it the 8-wide slice
// [8 * part, 8 * part + 8) of the 32 head dims over ALL keys.  The serial loops are 4x shorter than with one thread per
// row and the CU runs 8 waves instead of 2 (the kernel is latency-bound: 64 blocks on 256 CUs).
#define MHA_THREADS (4 * MHA_MAXL)

__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1, 64));
  return fmaxf(v, __shfl_xor(v, 2, 64));
}
__device__ __forceinline__ float quad_add(float v) {
  v += __shfl_xor(v, 1, 64);
  return v + __shfl_xor(v, 2, 64);
}
// acc[0..8) += w * srow[8 * part .. 8 * part + 8)
__device__ __forceinline__ void axpy8(float w, const float* __restrict__ s8, float* __restrict__ acc) {
  const float4 u = *reinterpret_cast<const float4*>(s8), v = *reinterpret_cast<const float4*>(s8 + 4);
  acc[0] = fmaf(w, u.x, acc[0]); acc[1] = fmaf(w, u.y, acc[1]); acc[2] = fmaf(w, u.z, acc[2]); acc[3] = fmaf(w, u.w, acc[3]);
  acc[4] = fmaf(w, v.x, acc[4]); acc[5] = fmaf(w, v.y, acc[5]); acc[6] = fmaf(w, v.z, acc[6]); acc[7] = fmaf(w, v.w, acc[7]);
}
// each of the 4 lanes of a row stages its 8-wide slice of the row into LDS
template <class T>
__device__ __forceinline__ void stage8(const T* g8, float* s8) {
  float t[8];
  Vec8<T>::load(g8, t);
  *reinterpret_cast<float4*>(s8) = make_float4(t[0], t[1], t[2], t[3]);
  *reinterpret_cast<float4*>(s8 + 4) = make_float4(t[4], t[5], t[6], t[7]);
}

// Forward: one block per (batch, head, chunk of MHA_FWD_ROWS query rows), 4 lanes per row.  The rows of a head are independent given
// K and V, so the head is cut into L / 32 blocks: the one-block-per-head version kept 64 of the 256 CUs busy and was bound by its own
// LDS traffic (every thread re-reads all K rows twice and all V rows once: 5.7 MB of ds_read_b128 per block, ~19 us); a chunk block
// stages the same K / V (14 KB each) and reads a quarter of that.
#define MHA_FWD_ROWS 32
template <class T>
__global__ __launch_bounds__(4 * MHA_FWD_ROWS) void mha_fwd_kernel(MhaArgs a, int nchunk) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int L = a.L;
  float* sK = sm;                       // [L][MHA_P]
  float* sV = sK + L * MHA_P;           // [L][MHA_P]
  float* sQ = sV + L * MHA_P;           // [MHA_FWD_ROWS][MHA_P]
  float* sS = sQ + MHA_FWD_ROWS * MHA_P;   // [MHA_FWD_ROWS][L + 1] scores -> exp -> (dropped) probabilities
  const int bm = blockIdx.x / nchunk, chunk = blockIdx.x % nchunk;
  const int b = bm / a.M, m = bm % a.M;
  const int il = threadIdx.x >> 2, part = threadIdx.x & 3;
  const int i = chunk * MHA_FWD_ROWS + il;
  const long long r0 = (long long)b * L;
  const bool live = i < L;
  const long long co = m * MHA_D + 8 * part;
  for (int r = il; r < L; r += MHA_FWD_ROWS) {
    stage8<T>((const T*)a.k + (r0 + r) * a.ldk + co, sK + r * MHA_P + 8 * part);
    stage8<T>((const T*)a.v + (r0 + r) * a.ldv + co, sV + r * MHA_P + 8 * part);
  }
  if (live) stage8<T>((const T*)a.q + (r0 + i) * a.ldq + co, sQ + il * MHA_P + 8 * part);
  else *reinterpret_cast<float4*>(sQ + il * MHA_P + 8 * part) = *reinterpret_cast<float4*>(sQ + il * MHA_P + 8 * part + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  float* Si = sS + il * (L + 1);
  float qi[MHA_D];
#pragma unroll
  for (int d = 0; d < MHA_D; d += 4) {
    const float4 t = *reinterpret_cast<const float4*>(sQ + il * MHA_P + d);
    qi[d] = t.x; qi[d + 1] = t.y; qi[d + 2] = t.z; qi[d + 3] = t.w;
  }
  float mx = -3.0e38f;
  for (int j = part; j < L; j += 4) {
    const float s = dot_row(qi, sK + j * MHA_P) * a.scale;
    Si[j] = s;
    mx = fmaxf(mx, s);
  }
  mx = quad_max(mx);
  float den = 0.f;
  for (int j = part; j < L; j += 4) den += __expf(Si[j] - mx);          // (this lane's own scores: no barrier needed)
  den = quad_add(den);
  const float inv = 1.f / den;
  const unsigned long long seed = a.pdrop > 0.f ? *a.seed : 0ull;
  const float keep_scale = a.pdrop > 0.f ? 1.f / (1.f - a.pdrop) : 1.f;
  if (live) {
    float* pg = a.probs + (((long long)b * a.M + m) * L + i) * L;
    for (int j = part; j < L; j += 4) {
      float pj = __expf(Si[j] - mx) * inv;
      pg[j] = pj;
      if (a.pdrop > 0.f) {
        const unsigned long long idx = (((unsigned long long)b * a.M + m) * L + i) * L + j;
        pj = uniform01(seed, a.salt, idx) >= a.pdrop ? pj * keep_scale : 0.f;
      }
      Si[j] = pj;
    }
  }
  __syncthreads();
  if (!live) return;
  float out[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < L; ++j) axpy8(Si[j], sV + j * MHA_P + 8 * part, out);
  Vec8<T>::store((T*)a.o + (r0 + i) * a.ldo + m * MHA_D + 8 * part, out);
}

template <class T>
__global__ __launch_bounds__(MHA_THREADS) void mha_bwd_kernel(MhaArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int L = a.L;
  float* sK = sm;                 // [L][MHA_P]
  float* sV = sK + L * MHA_P;
  float* sQ = sV + L * MHA_P;
  float* sG = sQ + L * MHA_P;     // dout rows
  float* sD = sG + L * MHA_P;     // [L][L + 1]: dP (dropped) -> dS
  float* sP = sD + L * (L + 1);   // [L][L + 1]: dropped probabilities (for dV)
  const int b = blockIdx.x / a.M, m = blockIdx.x % a.M;
  const int i = threadIdx.x >> 2, part = threadIdx.x & 3;
  const long long r0 = (long long)b * L;
  const bool live = i < L;
  if (live) {
    const long long co = m * MHA_D + 8 * part;
    stage8<T>((const T*)a.k + (r0 + i) * a.ldk + co, sK + i * MHA_P + 8 * part);
    stage8<T>((const T*)a.v + (r0 + i) * a.ldv + co, sV + i * MHA_P + 8 * part);
    stage8<T>((const T*)a.q + (r0 + i) * a.ldq + co, sQ + i * MHA_P + 8 * part);
    stage8<T>((const T*)a.dout + (r0 + i) * a.lddo + co, sG + i * MHA_P + 8 * part);
  }
  __syncthreads();
  const int ii = live ? i : 0;
  const float* pgb = a.probs + ((long long)b * a.M + m) * L * L;
  const unsigned long long seed = a.pdrop > 0.f ? *a.seed : 0ull;
  const float keep_scale = a.pdrop > 0.f ? 1.f / (1.f - a.pdrop) : 1.f;
  float* Di = sD + ii * (L + 1);
  float* Pi = sP + ii * (L + 1);
  const float* pi = pgb + (long long)ii * L;
  {
    // score-shaped: dP_ij = <dout_i, v_j> (dropped), dot_i = sum_j dP_ij p_ij, dS_ij = p_ij (dP_ij - dot_i)
    float gi[MHA_D];
#pragma unroll
    for (int d = 0; d < MHA_D; d += 4) {
      const float4 t = *reinterpret_cast<const float4*>(sG + ii * MHA_P + d);
      gi[d] = t.x; gi[d + 1] = t.y; gi[d + 2] = t.z; gi[d + 3] = t.w;
    }
    float dot = 0.f;
    for (int j = part; j < L; j += 4) {
      float dp = dot_row(gi, sV + j * MHA_P);
      float pd = pi[j];
      if (a.pdrop > 0.f) {
        const unsigned long long idx = (((unsigned long long)b * a.M + m) * L + ii) * L + j;
        const bool keep = uniform01(seed, a.salt, idx) >= a.pdrop;
        dp = keep ? dp * keep_scale : 0.f;
        pd = keep ? pd * keep_scale : 0.f;
      }
      dot = fmaf(dp, pi[j], dot);
      if (live) { Di[j] = dp; if (a.use_sp) Pi[j] = pd; }
    }
    dot = quad_add(dot);
    if (live)
      for (int j = part; j < L; j += 4) Di[j] = pi[j] * (Di[j] - dot);
  }
  __syncthreads();
  if (!live) return;
  // output-shaped: this lane owns dims [8 part, 8 part + 8) of dq_i (over keys) and of dk_i, dv_i (over queries)
  float dq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dk[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f},
        dv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < L; ++j) {
    axpy8(Di[j], sK + j * MHA_P + 8 * part, dq);
    axpy8(sD[j * (L + 1) + i], sQ + j * MHA_P + 8 * part, dk);      // dS_ji, query row j
    float pd;
    if (a.use_sp) pd = sP[j * (L + 1) + i];                          // dropped p_ji
    else {
      pd = pgb[(long long)j * L + i];
      if (a.pdrop > 0.f) {
        const unsigned long long idx = (((unsigned long long)b * a.M + m) * L + j) * L + i;
        pd = uniform01(seed, a.salt, idx) >= a.pdrop ? pd * keep_scale : 0.f;
      }
    }
    axpy8(pd, sG + j * MHA_P + 8 * part, dv);
  }
#pragma unroll
  for (int d = 0; d < 8; ++d) { dq[d] *= a.scale; dk[d] *= a.scale; }
  const long long co = m * MHA_D + 8 * part;
  Vec8<T>::store((T*)a.dq + (r0 + i) * a.lddq + co, dq);
  Vec8<T>::store((T*)a.dk + (r0 + i) * a.lddk + co, dk);
  Vec8<T>::store((T*)a.dv + (r0 + i) * a.lddv + co, dv);
}

extern "C" int emrt_mha_fwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo, float* probs,
                            int B, int M, int L, int D, float scale, float pdrop, const unsigned long long* seed, unsigned salt,
                            int dtype, void* stream) {
  EMRT_REQUIRE_FWD_DTYPE(dtype);
  EMRT_REQUIRE(q && k && v && o && probs, "null pointer");
  EMRT_REQUIRE(D == MHA_D, "head dim must be 32");
  EMRT_REQUIRE(L >= 1 && L <= MHA_MAXL, "sequence length must be <= 128");
  EMRT_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0, "row strides must be multiples of 8");
  EMRT_REQUIRE(pdrop == 0.f || seed, "dropout needs a device seed");
  MhaArgs a;
  memset(&a, 0, sizeof(a));
  a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.o = o; a.ldo = ldo; a.probs = probs;
  a.B = B; a.M = M; a.L = L; a.scale = scale; a.pdrop = pdrop; a.seed = seed; a.salt = salt;
  const size_t lds = (size_t)(2 * L * MHA_P + MHA_FWD_ROWS * MHA_P + MHA_FWD_ROWS * (L + 1)) * sizeof(float);      // <= 58 KB at L = 128
  const int nchunk = (L + MHA_FWD_ROWS - 1) / MHA_FWD_ROWS;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EMRT_F32) hipLaunchKernelGGL((mha_fwd_kernel<float>), dim3(B * M * nchunk), dim3(4 * MHA_FWD_ROWS), lds, st, a, nchunk);
  else if (dtype == EMRT_BF16) hipLaunchKernelGGL((mha_fwd_kernel<bf16_t>), dim3(B * M * nchunk), dim3(4 * MHA_FWD_ROWS), lds, st, a, nchunk);
  else hipLaunchKernelGGL((mha_fwd_kernel<f16_t>), dim3(B * M * nchunk), dim3(4 * MHA_FWD_ROWS), lds, st, a, nchunk);
  return check_launch("emrt_mha_fwd");
}

extern "C" int emrt_mha_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const float* probs,
                            const void* dout, int lddo, void* dq, int lddq, void* dk, int lddk, void* dv, int lddv, int B, int M,
                            int L, int D, float scale, float pdrop, const unsigned long long* seed, unsigned salt, int dtype,
                            void* stream) {
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  EMRT_REQUIRE(q && k && v && probs && dout && dq && dk && dv, "null pointer");
  EMRT_REQUIRE(D == MHA_D, "head dim must be 32");
  EMRT_REQUIRE(L >= 1 && L <= MHA_MAXL, "sequence length must be <= 128");
  EMRT_REQUIRE(pdrop == 0.f || seed, "dropout needs a device seed");
  MhaArgs a;
  memset(&a, 0, sizeof(a));
  a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.probs = const_cast<float*>(probs);
  a.B = B; a.M = M; a.L = L; a.scale = scale; a.pdrop = pdrop; a.seed = seed; a.salt = salt;
  a.dout = dout; a.lddo = lddo; a.dq = dq; a.dk = dk; a.dv = dv; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
  size_t lds = (size_t)(4 * L * MHA_P + 2 * L * (L + 1)) * sizeof(float);
  a.use_sp = lds <= 159 * 1024;
  if (!a.use_sp) lds = (size_t)(4 * L * MHA_P + L * (L + 1)) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)mha_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)mha_bwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) != hipSuccess)
      return fail("emrt_mha_bwd", "cannot raise the dynamic LDS limit");
    attr_done = true;
  }
  if (dtype == EMRT_F32) hipLaunchKernelGGL((mha_bwd_kernel<float>), dim3(B * M), dim3(MHA_THREADS), lds, st, a);
  else hipLaunchKernelGGL((mha_bwd_kernel<bf16_t>), dim3(B * M), dim3(MHA_THREADS), lds, st, a);
  return check_launch("emrt_mha_bwd");
}
