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

// ------------------------------------------------------------------------------------------------------------------------------------
// MFMA path (bf16 / fp16 storage): v_mfma_f32_16x16x32 -- the head dim IS the instruction's K = 32, so a 16 x 16 score tile is ONE MFMA.
// One block per (batch, head); wave w owns tile w of the (padded) sequence: 16 query rows in the forward, and in the backward the same 16
// rows as QUERIES (dq) and as KEYS (dk, dv).  Everything is computed TRANSPOSED so that no operand ever needs a lane transpose:
//     S^T[j][i] = sum_d K[j][d] Q[i][d]          A = rows of K (16 B per lane, straight from global), B = rows of Q
//   the C/D layout then gives lane (i = lane & 15, g = lane >> 4) the scores of ITS query i against keys j = 16 t + 4 g + r of tile t: the
//   softmax over j is a reduction over the lane's own registers plus two cross-lane steps (xor 16, 32), and for the second product
//     O^T[d][i] = sum_j V[j][d] Pd[i][j]         B = the probabilities AS THEY STAND in the accumulator registers (packed pairwise),
//   with k order permuted: k-step s takes element jj of lane group g from key j = 32 s + 16 (jj >> 2) + 4 g + (jj & 3); the A operand (V^T,
//   read from a transposed LDS image, two 8-byte reads) uses the same map, and a contraction does not care in which order k is walked
//   (cdna_hip_programming.md 3, "An accumulator tile as the next MFMA's operand").
// Saved for backward: (row max, 1 / row sum) per query -- 2 L floats per (batch, head) in the `probs` buffer -- instead of L x L
// probabilities; the backward recomputes P from them and uses rowdot_i = sum_j dP_ij P_ij from its own row pass.
// Dropout on the weights (layers.py:297): one 64-bit hash per (query, 4 consecutive keys) = four 16-bit uniforms; forward and backward
// call the same function, so they agree by construction (the fp32 VALU kernels above keep the per-element hash).
// 0.07 GMAC per tile: the point is not MFMA throughput but that a (batch, head) is ~100 matrix instructions per wave instead of ~3000
// dependent VALU instructions per lane: 17 -> ~6 us forward, 38 -> ~8 us backward at B = 8 (64 blocks).
// ------------------------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 mha_bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 mha_f16x8_t;
typedef __attribute__((ext_vector_type(4))) float mha_f32x4_t;

template <class T>
__device__ __forceinline__ mha_f32x4_t mha_mma(const uint4& a, const uint4& b, mha_f32x4_t c);
template <>
__device__ __forceinline__ mha_f32x4_t mha_mma<bf16_t>(const uint4& a, const uint4& b, mha_f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mha_bf16x8_t, a), __builtin_bit_cast(mha_bf16x8_t, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ mha_f32x4_t mha_mma<f16_t>(const uint4& a, const uint4& b, mha_f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mha_f16x8_t, a), __builtin_bit_cast(mha_f16x8_t, b), c, 0, 0, 0);
}
template <class T>
__device__ __forceinline__ uint32_t mha_pack2(float lo, float hi);
template <>
__device__ __forceinline__ uint32_t mha_pack2<bf16_t>(float lo, float hi) { return pack_bf16x2(lo, hi); }
template <>
__device__ __forceinline__ uint32_t mha_pack2<f16_t>(float lo, float hi) { return pack_f16x2(lo, hi); }

#define MHA_TP 136      /* pitch (elements) of the transposed [32][128] LDS images: 272-byte rows, 8-byte aligned quads */
#define MHA_TILES 8     /* 16-row tiles of the padded sequence (L <= 128) */

// 16 bytes of row `r` of a [B * L][ld] operand at head column `col`, zero for rows beyond this (batch, head)'s L
template <class T>
__device__ __forceinline__ uint4 mha_row16(const void* base, long long r0, int r, int L, int ld, int col) {
  if (r >= L) return make_uint4(0u, 0u, 0u, 0u);
  return *reinterpret_cast<const uint4*>((const T*)base + (r0 + r) * ld + col);
}
// the whole block writes the transposed image sXt[d][j] = X[j][d] (j < L, zero up to 128) of one [L][32] operand
template <class T>
__device__ __forceinline__ void mha_stage_t(const void* base, long long r0, int L, int ld, int col0, T* sXt) {
  for (int t = threadIdx.x; t < 128 * 4; t += blockDim.x) {
    const int j = t >> 2, c = t & 3;
    const uint4 v = mha_row16<T>(base, r0, j, L, ld, col0 + 8 * c);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) sXt[(8 * c + e) * MHA_TP + j].v = (unsigned short)((w[e >> 1] >> (16 * (e & 1))) & 0xffffu);
  }
}
// A fragment of k-step s for output rows d = 16 dt + (lane & 15): element jj <- column 32 s + 16 (jj >> 2) + 4 g + (jj & 3) of the image
template <class T>
__device__ __forceinline__ uint4 mha_frag_t(const T* sXt, int dt, int s, int lane) {
  const T* row = sXt + (16 * dt + (lane & 15)) * MHA_TP + 32 * s + 4 * (lane >> 4);
  const uint2 lo = *reinterpret_cast<const uint2*>(row), hi = *reinterpret_cast<const uint2*>(row + 16);
  return make_uint4(lo.x, lo.y, hi.x, hi.y);
}
// four 16-bit uniforms for (row, quad of 4 consecutive columns); keep iff u16 >= thr
__device__ __forceinline__ void mha_drop4(unsigned long long seed, unsigned salt, unsigned group, uint32_t (&h)[2]) {
  h[0] = mix32((group ^ (uint32_t)seed) + salt * 0x9E3779B9U);      // (the salt in the first round: see common.hpp drop_quad)
  h[1] = mix32(h[0] ^ (uint32_t)(seed >> 32) ^ (salt * 0x85EBCA6BU));
}
__device__ __forceinline__ bool mha_keep4(const uint32_t (&h)[2], int e, uint32_t thr) { return ((h[e >> 1] >> (16 * (e & 1))) & 0xffffu) >= thr; }

template <class T>
__global__ __launch_bounds__(64 * MHA_TILES) void mha_fwd_mfma_kernel(MhaArgs a) {
  __shared__ __attribute__((aligned(16))) T sVt[32 * MHA_TP];
  const int L = a.L, bm = blockIdx.x, b = bm / a.M, m = bm % a.M;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nt = (L + 15) >> 4;
  const int il = lane & 15, g = lane >> 4;
  const long long r0 = (long long)b * L;
  const int col = m * MHA_D;
  mha_stage_t<T>(a.v, r0, L, a.ldv, col, sVt);
  const int i = 16 * w + il;                                      // this lane's query
  const uint4 qf = mha_row16<T>(a.q, r0, i, L, a.ldq, col + 8 * g);
  mha_f32x4_t st[MHA_TILES];
#pragma unroll
  for (int t = 0; t < MHA_TILES; ++t) {
    st[t] = (mha_f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (t < nt) st[t] = mha_mma<T>(mha_row16<T>(a.k, r0, 16 * t + il, L, a.ldk, col + 8 * g), qf, st[t]);
  }
  float mx = -3.0e38f;
#pragma unroll
  for (int t = 0; t < MHA_TILES; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 16 * t + 4 * g + r;
      st[t][r] = j < L ? st[t][r] * a.scale : -3.0e38f;
      mx = fmaxf(mx, st[t][r]);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float den = 0.f;
#pragma unroll
  for (int t = 0; t < MHA_TILES; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      st[t][r] = (16 * t + 4 * g + r) < L ? __expf(st[t][r] - mx) : 0.f;
      den += st[t][r];
    }
  den += __shfl_xor(den, 16, 64);
  den += __shfl_xor(den, 32, 64);
  const float inv = 1.f / den;
  if (g == 0 && i < L) {                                          // what the backward recomputes the probabilities from
    float* stats = a.probs + (long long)bm * L * L;
    stats[i] = mx;
    stats[L + i] = inv;
  }
  const unsigned long long seed = a.pdrop > 0.f ? *a.seed : 0ull;
  const uint32_t thr = (uint32_t)(a.pdrop * 65536.f);
  const float ks = a.pdrop > 0.f ? inv / (1.f - a.pdrop) : inv;
  uint32_t pk[MHA_TILES][2];
#pragma unroll
  for (int t = 0; t < MHA_TILES; ++t) {
    float pv[4];
    uint32_t h[2];
    if (a.pdrop > 0.f) mha_drop4(seed, a.salt, ((unsigned)bm * (unsigned)L + (unsigned)i) * 32u + (unsigned)(4 * t + g), h);
#pragma unroll
    for (int r = 0; r < 4; ++r) pv[r] = (a.pdrop > 0.f && !mha_keep4(h, r, thr)) ? 0.f : st[t][r] * ks;
    pk[t][0] = mha_pack2<T>(pv[0], pv[1]);
    pk[t][1] = mha_pack2<T>(pv[2], pv[3]);
  }
  __syncthreads();                                                // the V^T image is complete
  if (w >= nt) return;
  mha_f32x4_t o[2] = {(mha_f32x4_t){0.f, 0.f, 0.f, 0.f}, (mha_f32x4_t){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int s = 0; s < MHA_TILES / 2; ++s) {
    if (2 * s >= nt) break;
    const uint4 pf = make_uint4(pk[2 * s][0], pk[2 * s][1], pk[2 * s + 1][0], pk[2 * s + 1][1]);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) o[dt] = mha_mma<T>(mha_frag_t<T>(sVt, dt, s, lane), pf, o[dt]);
  }
  if (i < L) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      uint2 v;
      v.x = mha_pack2<T>(o[dt][0], o[dt][1]);
      v.y = mha_pack2<T>(o[dt][2], o[dt][3]);
      *reinterpret_cast<uint2*>((T*)a.o + (r0 + i) * a.ldo + col + 16 * dt + 4 * g) = v;
    }
  }
}

template <class T>
__global__ __launch_bounds__(64 * MHA_TILES) void mha_bwd_mfma_kernel(MhaArgs a) {
  __shared__ __attribute__((aligned(16))) T sKt[32 * MHA_TP];
  __shared__ __attribute__((aligned(16))) T sQt[32 * MHA_TP];
  __shared__ __attribute__((aligned(16))) T sGt[32 * MHA_TP];
  __shared__ float sMx[128], sInv[128], sDot[128];
  const int L = a.L, bm = blockIdx.x, b = bm / a.M, m = bm % a.M;
  // TWO blocks per (batch, head) (gridDim.y == 2; 64 (batch, head) pairs are a quarter of the CUs and the launch is one dependent chain of
  // load -> MFMA -> exp -> MFMA -> store): both run pass 1's front (probabilities, dP, every row's rowdot -- a few MFMAs, redundantly),
  // block 0 then finishes dq, block 1 runs pass 2 (dk, dv): the launch lasts front + max(dq, pass 2) instead of their sum.  gridDim.y == 1: one block does all.
  const int role = gridDim.y == 2 ? (int)blockIdx.y : -1;      // 0: dq only, 1: dk / dv only, -1: everything
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nt = (L + 15) >> 4;
  const int il = lane & 15, g = lane >> 4;
  const long long r0 = (long long)b * L;
  const int col = m * MHA_D;
  if (role != 1) mha_stage_t<T>(a.k, r0, L, a.ldk, col, sKt);
  if (role != 0) {
    mha_stage_t<T>(a.q, r0, L, a.ldq, col, sQt);
    mha_stage_t<T>(a.dout, r0, L, a.lddo, col, sGt);
  }
  const float* stats = a.probs + (long long)bm * L * L;
  for (int t = threadIdx.x; t < 128; t += blockDim.x) {
    sMx[t] = t < L ? stats[t] : 0.f;
    sInv[t] = t < L ? stats[L + t] : 0.f;
    if (t >= 16 * nt) sDot[t] = 0.f;      // rows of the tiles no wave owns: pass 2 multiplies them by a zero probability, and 0 x (whatever a previous
                                          // kernel left in LDS -- a NaN pattern under load, round 5) is a NaN in dk / dv
  }
  const unsigned long long seed = a.pdrop > 0.f ? *a.seed : 0ull;
  const uint32_t thr = (uint32_t)(a.pdrop * 65536.f);
  const float kd = a.pdrop > 0.f ? 1.f / (1.f - a.pdrop) : 1.f;
  const int x = 16 * w + il;            // this lane's query (pass 1) / key (pass 2)
  // ---- pass 1: the row strip of query tile w: P, dP, rowdot, dS -> dq ------------------------------------------------------------
  {
    const uint4 qf = mha_row16<T>(a.q, r0, x, L, a.ldq, col + 8 * g), gf = mha_row16<T>(a.dout, r0, x, L, a.lddo, col + 8 * g);
    const float mx = x < L ? stats[x] : 0.f, inv = x < L ? stats[L + x] : 0.f;
    mha_f32x4_t st[MHA_TILES], dp[MHA_TILES];
#pragma unroll
    for (int t = 0; t < MHA_TILES; ++t) {
      st[t] = (mha_f32x4_t){0.f, 0.f, 0.f, 0.f};
      dp[t] = (mha_f32x4_t){0.f, 0.f, 0.f, 0.f};
      if (t < nt) {
        st[t] = mha_mma<T>(mha_row16<T>(a.k, r0, 16 * t + il, L, a.ldk, col + 8 * g), qf, st[t]);
        dp[t] = mha_mma<T>(mha_row16<T>(a.v, r0, 16 * t + il, L, a.ldv, col + 8 * g), gf, dp[t]);
      }
    }
    float dot = 0.f;
#pragma unroll
    for (int t = 0; t < MHA_TILES; ++t) {
      uint32_t h[2];
      if (a.pdrop > 0.f) mha_drop4(seed, a.salt, ((unsigned)bm * (unsigned)L + (unsigned)x) * 32u + (unsigned)(4 * t + g), h);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool in = (16 * t + 4 * g + r) < L && x < L;
        const float p = in ? __expf(st[t][r] * a.scale - mx) * inv : 0.f;
        const bool keep = !(a.pdrop > 0.f) || mha_keep4(h, r, thr);
        const float d = keep ? dp[t][r] * kd : 0.f;        // dP_ij (gradient of the UNdropped probability)
        st[t][r] = p;
        dp[t][r] = d;
        dot = fmaf(d, p, dot);
      }
    }
    dot += __shfl_xor(dot, 16, 64);
    dot += __shfl_xor(dot, 32, 64);
    if (g == 0) sDot[x] = dot;
    uint32_t pk[MHA_TILES][2];
#pragma unroll
    for (int t = 0; t < MHA_TILES; ++t) {
      float ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) ds[r] = st[t][r] != 0.f ? st[t][r] * (dp[t][r] - dot) * a.scale : 0.f;      // (padding rows / keys: exactly 0, whatever dot holds)
      pk[t][0] = mha_pack2<T>(ds[0], ds[1]);
      pk[t][1] = mha_pack2<T>(ds[2], ds[3]);
    }
    __syncthreads();                                              // transposed images, statistics and every row's rowdot are in LDS
    if (w < nt && role != 1) {
      mha_f32x4_t dq[2] = {(mha_f32x4_t){0.f, 0.f, 0.f, 0.f}, (mha_f32x4_t){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int s = 0; s < MHA_TILES / 2; ++s) {
        if (2 * s >= nt) break;
        const uint4 pf = make_uint4(pk[2 * s][0], pk[2 * s][1], pk[2 * s + 1][0], pk[2 * s + 1][1]);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) dq[dt] = mha_mma<T>(mha_frag_t<T>(sKt, dt, s, lane), pf, dq[dt]);
      }
      if (x < L) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          uint2 v;
          v.x = mha_pack2<T>(dq[dt][0], dq[dt][1]);
          v.y = mha_pack2<T>(dq[dt][2], dq[dt][3]);
          *reinterpret_cast<uint2*>((T*)a.dq + (r0 + x) * a.lddq + col + 16 * dt + 4 * g) = v;
        }
      }
    }
  }
  if (w >= nt || role == 0) return;
  // ---- pass 2: the column strip of key tile w: S[i][j], dPd[i][j] with the lane's key j = x fixed and the queries i in the registers ----
  {
    const uint4 kf = mha_row16<T>(a.k, r0, x, L, a.ldk, col + 8 * g), vf = mha_row16<T>(a.v, r0, x, L, a.ldv, col + 8 * g);
    mha_f32x4_t st[MHA_TILES], dp[MHA_TILES];
#pragma unroll
    for (int t = 0; t < MHA_TILES; ++t) {
      st[t] = (mha_f32x4_t){0.f, 0.f, 0.f, 0.f};
      dp[t] = (mha_f32x4_t){0.f, 0.f, 0.f, 0.f};
      if (t < nt) {
        st[t] = mha_mma<T>(mha_row16<T>(a.q, r0, 16 * t + il, L, a.ldq, col + 8 * g), kf, st[t]);
        dp[t] = mha_mma<T>(mha_row16<T>(a.dout, r0, 16 * t + il, L, a.lddo, col + 8 * g), vf, dp[t]);
      }
    }
    uint32_t pks[MHA_TILES][2], pkp[MHA_TILES][2];
#pragma unroll
    for (int t = 0; t < MHA_TILES; ++t) {
      float ds[4], pd[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * t + 4 * g + r;
        const bool in = i < L && x < L;
        const float p = in ? __expf(st[t][r] * a.scale - sMx[i]) * sInv[i] : 0.f;
        bool keep = true;
        if (a.pdrop > 0.f) {      // the forward's group is (query i, the quad of keys x belongs to): one hash per register here
          uint32_t h[2];
          mha_drop4(seed, a.salt, ((unsigned)bm * (unsigned)L + (unsigned)i) * 32u + (unsigned)(x >> 2), h);
          keep = mha_keep4(h, x & 3, thr);
        }
        const float d = keep ? dp[t][r] * kd : 0.f;
        ds[r] = in ? p * (d - sDot[i]) * a.scale : 0.f;
        pd[r] = (in && keep) ? p * kd : 0.f;
      }
      pks[t][0] = mha_pack2<T>(ds[0], ds[1]); pks[t][1] = mha_pack2<T>(ds[2], ds[3]);
      pkp[t][0] = mha_pack2<T>(pd[0], pd[1]); pkp[t][1] = mha_pack2<T>(pd[2], pd[3]);
    }
    mha_f32x4_t dk[2] = {(mha_f32x4_t){0.f, 0.f, 0.f, 0.f}, (mha_f32x4_t){0.f, 0.f, 0.f, 0.f}};
    mha_f32x4_t dv[2] = {(mha_f32x4_t){0.f, 0.f, 0.f, 0.f}, (mha_f32x4_t){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < MHA_TILES / 2; ++s) {
      if (2 * s >= nt) break;
      const uint4 sf = make_uint4(pks[2 * s][0], pks[2 * s][1], pks[2 * s + 1][0], pks[2 * s + 1][1]);
      const uint4 pf = make_uint4(pkp[2 * s][0], pkp[2 * s][1], pkp[2 * s + 1][0], pkp[2 * s + 1][1]);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        dk[dt] = mha_mma<T>(mha_frag_t<T>(sQt, dt, s, lane), sf, dk[dt]);
        dv[dt] = mha_mma<T>(mha_frag_t<T>(sGt, dt, s, lane), pf, dv[dt]);
      }
    }
    if (x < L) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        uint2 v;
        v.x = mha_pack2<T>(dk[dt][0], dk[dt][1]); v.y = mha_pack2<T>(dk[dt][2], dk[dt][3]);
        *reinterpret_cast<uint2*>((T*)a.dk + (r0 + x) * a.lddk + col + 16 * dt + 4 * g) = v;
        v.x = mha_pack2<T>(dv[dt][0], dv[dt][1]); v.y = mha_pack2<T>(dv[dt][2], dv[dt][3]);
        *reinterpret_cast<uint2*>((T*)a.dv + (r0 + x) * a.lddv + col + 16 * dt + 4 * g) = v;
      }
    }
  }
}

extern "C" int emrt_mha_fwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo, float* probs,
                            int B, int M, int L, int D, float scale, float pdrop, const unsigned long long* seed, unsigned salt,
                            int* path_out, int dtype, void* stream) {
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
  // bf16 / fp16: the MFMA kernels (one block per (batch, head), a wave per 16 rows); `probs` then holds (row max, 1 / row sum) in the first 2 L
  // floats of each (batch, head) slab -- the backward of the SAME dtype reads them back.  mha_valu = 1: the VALU kernels for every dtype (A/B).
  const bool mfma = dtype != EMRT_F32 && !g_tune.mha_valu && L >= 2 && ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0 &&
                    (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 16 == 0) && ((uintptr_t)o % 8 == 0);
  if (path_out) *path_out = mfma ? 1 : 0;      // what `probs` holds now: the backward must be told (emrt_mha_bwd: path)
  if (mfma) {
    const int threads = 64 * ((L + 15) / 16);
    if (dtype == EMRT_BF16) hipLaunchKernelGGL((mha_fwd_mfma_kernel<bf16_t>), dim3(B * M), dim3(threads), 0, st, a);
    else hipLaunchKernelGGL((mha_fwd_mfma_kernel<f16_t>), dim3(B * M), dim3(threads), 0, st, a);
    return check_launch("emrt_mha_fwd");
  }
  if (dtype == EMRT_F32) hipLaunchKernelGGL((mha_fwd_kernel<float>), dim3(B * M * nchunk), dim3(4 * MHA_FWD_ROWS), lds, st, a, nchunk);
  else if (dtype == EMRT_BF16) hipLaunchKernelGGL((mha_fwd_kernel<bf16_t>), dim3(B * M * nchunk), dim3(4 * MHA_FWD_ROWS), lds, st, a, nchunk);
  else hipLaunchKernelGGL((mha_fwd_kernel<f16_t>), dim3(B * M * nchunk), dim3(4 * MHA_FWD_ROWS), lds, st, a, nchunk);
  return check_launch("emrt_mha_fwd");
}

extern "C" int emrt_mha_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const float* probs,
                            const void* dout, int lddo, void* dq, int lddq, void* dk, int lddk, void* dv, int lddv, int B, int M,
                            int L, int D, float scale, float pdrop, const unsigned long long* seed, unsigned salt, int path, int dtype,
                            void* stream) {
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  EMRT_REQUIRE(q && k && v && probs && dout && dq && dk && dv, "null pointer");
  EMRT_REQUIRE(path == 0 || path == 1, "path: what emrt_mha_fwd reported for the call that filled `probs` (0: L x L probabilities, 1: row statistics)");
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
  // `probs` holds (row max, 1 / row sum) after the MFMA forward and the L x L probabilities after the VALU forward: the backward runs the kernel
  // that matches what the forward REPORTED (round 5 re-derived the choice from its own operands and the mutable knob: a dout view with another
  // alignment, or the knob flipped in between, made it read the slab as the other layout)
  const bool mfma = path == 1;
  if (mfma) {
    EMRT_REQUIRE(dtype == EMRT_BF16 && L >= 2 && ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && lddo % 8 == 0 &&
                     (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dout) % 16 == 0) && lddq % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0 &&
                     (((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) % 8 == 0),
                 "the forward took the MFMA kernel (probs = row statistics) but these gradient operands cannot run its backward: 16-byte aligned q / k / v / dout "
                 "with row strides that are multiples of 8, 8-byte aligned dq / dk / dv with row strides that are multiples of 4");
    hipLaunchKernelGGL((mha_bwd_mfma_kernel<bf16_t>), dim3(B * M, g_tune.mha_bwd_split ? 2 : 1), dim3(64 * ((L + 15) / 16)), 0, st, a);
    return check_launch("emrt_mha_bwd");
  }
  if (dtype == EMRT_F32) hipLaunchKernelGGL((mha_bwd_kernel<float>), dim3(B * M), dim3(MHA_THREADS), lds, st, a);
  else hipLaunchKernelGGL((mha_bwd_kernel<bf16_t>), dim3(B * M), dim3(MHA_THREADS), lds, st, a);
  return check_launch("emrt_mha_bwd");
}
