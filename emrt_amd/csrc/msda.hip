// Multi-scale deformable attention core for gfx950 -- the EMRT path's irregular gather.
//
// Replaces, in ONE kernel, the reference's softmax + location arithmetic + 3x grid_sample + stack + mul + sum
// (transformer_encoder_decoder.py:89-104 and utils.py:64-97), which materialise a [B*M, D, Lq, L*P] tensor.
//   a[b,q,m,:]   = softmax_{l,p}( logits[b,q,m,l,p] )
//   loc[...,l,p] = ref[b,q,l,:] + off[b,q,m,l,p,:] / (W_l, H_l)                (x, y in [0,1])
//   out[b,q,m,:] = sum_{l,p} a * bilinear_zero_pad(value_l[b,:,m,:], x = loc_x*W_l - 0.5, y = loc_y*H_l - 0.5)
//
// Layout: value [B][Lv][M*D] (row stride ldv), D = 32.  `offw` is the fp32 output of the fused
// sampling_offsets|attention_weights projection: row (b*Lq+q) holds M*L*P*2 offsets then M*L*P logits.
// Work split: 4 lanes x 8 channels per (q, head); a wave covers 2 queries x 8 heads (M = 8) so its output is two
// full 256-channel rows.  Every sample corner is one 16-byte (bf16) / 32-byte (f32) load per lane; 4 lanes
// fetch one contiguous 64/128-byte head slice.  HBM-compulsory bytes are tiny; the kernel is
// bound by L2/TA gather throughput, which is why value stays L2-resident across the B*M slabs.
#include "common.hpp"
#include <type_traits>
#include <stdlib.h>

using namespace emrt;

struct MsdaArgs {
  const void* value;
  int ldv;
  long long v_bs;
  const float* offw;
  int ldo;
  const float* ref;
  long long ref_bs;   // 0 => same reference points for every batch element
  int ref_L;          // reference points per query: L (one per level) or 1 (shared by all levels)
  void* out;
  int B, Lq, M;
  int h[4], w[4], start[4];
  float inv_h[4], inv_w[4];   // 1/H_l, 1/W_l: the offset normalisation off / (W, H) as a multiply (exact for power-of-two maps)
  // backward only
  const void* dout;
  float* dvalue;      // fp32 [B][Lv][M*32] dense, accumulated with atomics (caller zeroes)
  long long dv_bs;
  void* doffw;        // [B*Lq][ldo] (offset + logit gradients), fully overwritten: fp32, or the compute dtype T when doffw_t
  int doffw_t;
  float* dref;        // fp32 [B][Lq][ref_L][2] or null, fully overwritten
  float* probs;       // fp32 [B*Lq][M*L*P] softmax probabilities (written by the gradient kernel, read by the LDS scatter)
  void* dvalue_t;     // LDS path: [B][Lv][M*32] in the compute dtype, fully overwritten
  int Lv;
  int g_level[16], g_pix0[16], g_npix[16];   // scatter blocks: (level, first flat pixel, pixel count) of each LDS slab range
  int g_npix_max;                         // largest g_npix (+ the two guard bands): the per-half-wave sample records sit behind a slab of that size
  int g_qs[16], g_nqs[16];                // query split of a scatter block: it walks the query groups qs, qs + nqs, ... (nqs = 1: all of them)
  int g_part[16];                         // nqs > 1: offset (ints) of the block's partial slab inside its (batch, head) stretch of `part`
  int* part;                              // int32 partial slabs of the query-split blocks, [B * M][part_stride]
  long long part_stride;
  int f_n, f_pix0[8], f_npix[8], f_nqs[8], f_part[8];      // the ranges that were split: summed and written by msda_bwd_value_finalize_kernel
  int g_guard;                            // zero guard pixels on both sides of a scatter slab (widest level + 2): corners whose weight is 0 are
                                          // still ADDED (branch-free inner loop) and may fall up to W + 1 pixels outside the block's range
  float* gmax;        // [B*M][gmax_n] per-block max |dout| of a (batch, head) slice, left by the LDS gradient kernel for the scatter
  int gmax_n;         // 0: the scatter scans dout itself
};

template <class T>
__device__ __forceinline__ void load8(const T* p, float (&o)[8]) { Vec8<T>::load(p, o); }

// one sample's offset / logit gradients into the [ldo] row: fp32, or rounded to the compute dtype when the consumer (the
// offsets|logits projection's backward GEMM) reads T anyway (saves the cast launch and half the bytes)
template <class T>
__device__ __forceinline__ void store_doffw(const MsdaArgs& a, long long bq, int off_idx, int log_idx, float gx, float gy, float gl) {
  if (a.doffw_t) {
    T* row = (T*)a.doffw + bq * a.ldo;
    row[off_idx] = from_f32<T>(gx);
    row[off_idx + 1] = from_f32<T>(gy);
    row[log_idx] = from_f32<T>(gl);
  } else {
    float* row = (float*)a.doffw + bq * a.ldo;
    *reinterpret_cast<float2*>(row + off_idx) = make_float2(gx, gy);
    row[log_idx] = gl;
  }
}

// ---- forward -------------------------------------------------------------------------------------------------------
// A quad (4 lanes x 8 channels) owns one (query, head) pair.  Everything that does not depend on the channel -- the
// softmax over the L*P logits, the sample coordinates, the four bilinear corner weights with the attention probability
// and the zero padding folded in -- is computed ONCE per pair: lane `sub` of the quad takes samples sub, sub + 4, ...
// (the first version had all four lanes repeat all of it: 3/4 of ~900 instructions per pair wasted in a kernel that is
// bound by VALU issue), and every sample's five values (4 corner weights + pixel index) are then broadcast inside the
// quad with DPP quad_perm moves -- register to register, no LDS, no barrier.  A corner whose weight is 0 (outside the
// map: grid_sample's zero padding, utils.py:87-88) contributes fma(0, v, acc) = acc exactly, so "skip it" (global
// kernel) and "read a harmless address" (LDS kernel) give bit-identical sums.
template <int K>
__device__ __forceinline__ float quad_bcast(float v) {        // value of lane K of every quad
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), K | (K << 2) | (K << 4) | (K << 6), 0xf, 0xf, false));
}
template <int K>
__device__ __forceinline__ int quad_bcast(int v) {
  return __builtin_amdgcn_mov_dpp(v, K | (K << 2) | (K << 4) | (K << 6), 0xf, 0xf, false);
}
__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false)));   // quad_perm [1,0,3,2]
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false)));   // quad_perm [2,3,0,1]
  return v;
}
__device__ __forceinline__ float quad_add(float v) {          // same value in all four lanes (order fixed: (a+b)+(c+d))
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
  return v;
}

// The samples lane `sub` of a quad prepares for its pair: slot j holds sample sub + 4 j.
template <int L, int P>
struct MsdaPrep {
  static constexpr int LP = L * P;
  static constexpr int NS = (LP + 3) / 4;          // slots per lane
  float w00[NS], w01[NS], w10[NS], w11[NS];        // corner weights (probability * bilinear weight, 0 outside the map)
  int idx[NS];                                     // y0 * W + x0 of the sample's level (0 when no corner is inside)

  // band kernels only: bit j set = slot j's sample has a valid corner outside the rows the block has staged; its weights are zeroed
  // here (the branch-free LDS loop then adds nothing for it) and the block gathers it from global memory afterwards
  unsigned miss;
  float smx, sinv;                                 // softmax max / 1 / denominator of the pair (to re-derive a missed sample)
  // forward kernels, 16-bit value types: the weights of a sample's two corner ROWS, rounded to the value type and packed --
  // wp0 = (w00 | w01 << 16), wp1 = (w10 | w11 << 16): operand B of a 2-way dot product whose operand A is (corner x0, corner x0 + 1) of one
  // channel (msda_dot_row).  After pack16() the fp32 weights are dead in the forward kernels (3 values per sample to broadcast, not 5).
  unsigned wp0[NS], wp1[NS];
  template <class T>
  __device__ __forceinline__ void pack16() {
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      if constexpr (sizeof(T) == 2 && !std::is_same<T, bf16_t>::value) {
        // v_cvt_pk_f16_f32 (round to nearest even) spelled out: through pack_f16x2's vector convert the slot arrays stayed in scratch memory
        // (+3 us on the fp16 encoder call), and scalar converts let hipcc fuse the weights' last multiply into v_fma_mix in one kernel and
        // not in the other (single against double rounding: the LDS and the global kernel then differed in 1 output of 15 000 by one ulp)
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(wp0[j]) : "v"(w00[j]), "v"(w01[j]));
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(wp1[j]) : "v"(w10[j]), "v"(w11[j]));
      }
      else { wp0[j] = pack_bf16x2(w00[j], w01[j]); wp1[j] = pack_bf16x2(w10[j], w11[j]); }
    }
  }

  // row: the pair's fp32 [M*LP*2 offsets | M*LP logits] row; refp: its reference point(s); live: false for tail lanes
  // BAND: lo / hi = first / one-past-last staged row of every level (msda_fwd_band_kernel); samples outside get idx = lo * W
  template <bool BAND = false>
  __device__ __forceinline__ void run(const MsdaArgs& a, const float* row, const float* refp, int m, int sub, bool live,
                                      const int* lo = nullptr, const int* hi = nullptr) {
    miss = 0u;
    const float* offp = row + m * LP * 2;
    const float* logp = row + a.M * LP * 2 + m * LP;
    float lg[NS];
    float2 of[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {                 // all loads first: one round trip
      const int smp = sub + 4 * j;
      const bool has = live && smp < LP;
      lg[j] = has ? logp[smp] : -3.0e38f;
      of[j] = has ? *reinterpret_cast<const float2*>(offp + smp * 2) : make_float2(0.f, 0.f);
    }
    float rx[L], ry[L];
    const int rls = a.ref_L == 1 ? 0 : 2;
#pragma unroll
    for (int l = 0; l < L; ++l) { rx[l] = live ? refp[l * rls] : 0.f; ry[l] = live ? refp[l * rls + 1] : 0.f; }
    float mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < NS; ++j) mx = fmaxf(mx, lg[j]);
    mx = quad_max(mx);
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < NS; ++j) { lg[j] = (sub + 4 * j < LP) ? __expf(lg[j] - mx) : 0.f; den += lg[j]; }
    den = quad_add(den);
    const float inv = 1.f / den;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      // level of sample sub + 4 j: the division is by a compile-time P, the level-dependent constants are selected
      const int smp = sub + 4 * j;
      int l = 0;
#pragma unroll
      for (int t = 1; t < L; ++t) l += smp >= t * P ? 1 : 0;
      int H = a.h[0], W = a.w[0];
      float ih = a.inv_h[0], iw = a.inv_w[0], rxl = rx[0], ryl = ry[0];
#pragma unroll
      for (int t = 1; t < L; ++t)
        if (l == t) { H = a.h[t]; W = a.w[t]; ih = a.inv_h[t]; iw = a.inv_w[t]; rxl = rx[t]; ryl = ry[t]; }
      const float x = (rxl + of[j].x * iw) * (float)W - 0.5f;
      const float y = (ryl + of[j].y * ih) * (float)H - 0.5f;
      const float xf = floorf(x), yf = floorf(y);
      const float lx = x - xf, ly = y - yf;
      // samples far outside would overflow the int conversion: clamp (any value below -1 / above the map is "outside")
      const int x0 = (int)fminf(fmaxf(xf, -2.f), 16777216.f), y0 = (int)fminf(fmaxf(yf, -2.f), 16777216.f);
      const float aw = lg[j] * inv;
      const bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)(x0 + 1) < (unsigned)W;
      const bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)(y0 + 1) < (unsigned)H;
      w00[j] = (vy0 && vx0) ? aw * (1.f - ly) * (1.f - lx) : 0.f;
      w01[j] = (vy0 && vx1) ? aw * (1.f - ly) * lx : 0.f;
      w10[j] = (vy1 && vx0) ? aw * ly * (1.f - lx) : 0.f;
      w11[j] = (vy1 && vx1) ? aw * ly * lx : 0.f;
      // with at least one corner inside, y0 is in [-1, H-1] and x0 in [-1, W-1]: the four corner indices stay within
      // [-(W+1), H*W + W] of the level -- the range the LDS slab's guard bands cover
      const bool any_in = (vy0 || vy1) && (vx0 || vx1) && smp < LP;
      idx[j] = any_in ? y0 * W + x0 : 0;
      if constexpr (BAND) {
        int blo = lo[0], bhi = hi[0];
#pragma unroll
        for (int t = 1; t < L; ++t)
          if (l == t) { blo = lo[t]; bhi = hi[t]; }
        const bool in_band = (!vy0 || (y0 >= blo && y0 < bhi)) && (!vy1 || (y0 + 1 >= blo && y0 + 1 < bhi));
        if (!any_in || !in_band) {
          if (any_in && live) miss |= 1u << j;
          w00[j] = 0.f; w01[j] = 0.f; w10[j] = 0.f; w11[j] = 0.f;
          idx[j] = blo * W;                        // first staged pixel of the level: a harmless address
        }
      }
    }
    smx = mx; sinv = inv;
  }

  // the five values of sample smp of this pair again, from global memory (band kernels: a sample that missed the staged rows)
  __device__ __forceinline__ void one(const MsdaArgs& a, const float* row, const float* refp, int m, int smp, float& c00, float& c01,
                                      float& c10, float& c11, int& id, int& lev) const {
    const float* offp = row + m * LP * 2;
    const float* logp = row + a.M * LP * 2 + m * LP;
    int l = 0;
#pragma unroll
    for (int t = 1; t < L; ++t) l += smp >= t * P ? 1 : 0;
    lev = l;
    const int rls = a.ref_L == 1 ? 0 : 2;
    const int H = a.h[l], W = a.w[l];
    const float2 of = *reinterpret_cast<const float2*>(offp + smp * 2);
    const float x = (refp[l * rls] + of.x * a.inv_w[l]) * (float)W - 0.5f;
    const float y = (refp[l * rls + 1] + of.y * a.inv_h[l]) * (float)H - 0.5f;
    const float xf = floorf(x), yf = floorf(y);
    const float lx = x - xf, ly = y - yf;
    const int x0 = (int)fminf(fmaxf(xf, -2.f), 16777216.f), y0 = (int)fminf(fmaxf(yf, -2.f), 16777216.f);
    const float aw = __expf(logp[smp] - smx) * sinv;
    const bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)(x0 + 1) < (unsigned)W;
    const bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)(y0 + 1) < (unsigned)H;
    c00 = (vy0 && vx0) ? aw * (1.f - ly) * (1.f - lx) : 0.f;
    c01 = (vy0 && vx1) ? aw * (1.f - ly) * lx : 0.f;
    c10 = (vy1 && vx0) ? aw * ly * (1.f - lx) : 0.f;
    c11 = (vy1 && vx1) ? aw * ly * lx : 0.f;
    id = y0 * W + x0;
  }
};

// the five values of sample SMP (compile-time), broadcast from the lane of the quad that prepared it
#define MSDA_BCAST5(PP, SMP, C00, C01, C10, C11, ID)                                                     \
  do {                                                                                                   \
    constexpr int K_ = (SMP) & 3, J_ = (SMP) >> 2;                                                       \
    C00 = quad_bcast<K_>((PP).w00[J_]); C01 = quad_bcast<K_>((PP).w01[J_]); C10 = quad_bcast<K_>((PP).w10[J_]); \
    C11 = quad_bcast<K_>((PP).w11[J_]); ID = quad_bcast<K_>((PP).idx[J_]);                               \
  } while (0)

// acc += w * v over the lane's 8 channels (v: 16 raw bytes of T / 32 of float)
template <class T>
__device__ __forceinline__ void axpy8(float (&acc)[8], float w, const float (&v)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = fmaf(w, v[e], acc[e]);
}

// 16-bit value types: acc[e] += r0[e] * w.lo + r1[e] * w.hi over the lane's 8 channels, r0 / r1 = the raw 16 bytes of the corners x0 and
// x0 + 1 of one row, w = their two weights in the value type.  Per pair of channels: two byte permutes bring (r0[c], r1[c]) and
// (r0[c+1], r1[c+1]) together, two v_dot2_f32_{bf16,f16} accumulate in fp32: ONE instruction per multiply-add where unpack + fma needs two
// (the gather is bound by VALU issue: 72 corners x 8 channels per lane and pair).  The price is the weights' rounding to the value type --
// 2^-9 relative for bf16, the rounding every stored activation already carries; the products and the sums stay fp32.
typedef __attribute__((ext_vector_type(2))) __bf16 msda_bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 msda_f16x2_t;
template <class T>
__device__ __forceinline__ void msda_dot_pair(float& acc_even, float& acc_odd, unsigned x0, unsigned x1, unsigned w) {
  const unsigned lo = __builtin_amdgcn_perm(x1, x0, 0x05040100u);      // (r0 even channel, r1 even channel)
  const unsigned hi = __builtin_amdgcn_perm(x1, x0, 0x07060302u);      // (r0 odd channel, r1 odd channel)
  if constexpr (std::is_same<T, bf16_t>::value) {
    acc_even = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(msda_bf16x2_t, lo), __builtin_bit_cast(msda_bf16x2_t, w), acc_even, false);
    acc_odd = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(msda_bf16x2_t, hi), __builtin_bit_cast(msda_bf16x2_t, w), acc_odd, false);
  } else {
    acc_even = __builtin_amdgcn_fdot2(__builtin_bit_cast(msda_f16x2_t, lo), __builtin_bit_cast(msda_f16x2_t, w), acc_even, false);
    acc_odd = __builtin_amdgcn_fdot2(__builtin_bit_cast(msda_f16x2_t, hi), __builtin_bit_cast(msda_f16x2_t, w), acc_odd, false);
  }
}
template <class T>
__device__ __forceinline__ void msda_dot_row(float (&acc)[8], const uint4& r0, const uint4& r1, unsigned w) {
  msda_dot_pair<T>(acc[0], acc[1], r0.x, r1.x, w);      // (components by name: a local array of them ended up in scratch memory)
  msda_dot_pair<T>(acc[2], acc[3], r0.y, r1.y, w);
  msda_dot_pair<T>(acc[4], acc[5], r0.z, r1.z, w);
  msda_dot_pair<T>(acc[6], acc[7], r0.w, r1.w, w);
}
#define MSDA_BCAST3(PP, SMP, W0, W1, ID)                                                                 \
  do {                                                                                                   \
    constexpr int K_ = (SMP) & 3, J_ = (SMP) >> 2;                                                       \
    W0 = (unsigned)quad_bcast<K_>((int)(PP).wp0[J_]); W1 = (unsigned)quad_bcast<K_>((int)(PP).wp1[J_]); ID = quad_bcast<K_>((PP).idx[J_]); \
  } while (0)

// 16 bytes through a buffer descriptor: an offset with bit 31 set is out of the descriptor's range and returns zeros -- a corner of weight
// 0 (it may lie outside the tensor) costs neither a branch nor a select of addresses (which hipcc lowered to flat loads from a stack slot)
typedef __attribute__((ext_vector_type(4))) unsigned int msda_u32x4_t;
__device__ __forceinline__ uint4 msda_buf_load16(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  const msda_u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}
constexpr unsigned MSDA_OOB = 0x80000000u;

#define MSDA_FWD_PITCH 64        /* bytes per LDS row of the staged forward slab (unpadded: see msda_fwd_lds_kernel) */
// vb: the lane's pointer to (batch, pixel 0, head slice); rs / vb_bytes: the whole value tensor as a buffer + the lane's byte offset of the same
template <class T, int L, int P, int SMP>
__device__ __forceinline__ void msda_gather_global(const MsdaArgs& a, const MsdaPrep<L, P>& pp, const T* vb, __amdgpu_buffer_rsrc_t rs, unsigned vb_bytes, float (&acc)[8]) {
  if constexpr (SMP < L * P) {
    constexpr int l = SMP / P;
    const int W = a.w[l];
    if constexpr (sizeof(T) == 2) {                  // same arithmetic, same order as msda_gather_lds: bit-identical results
      unsigned w0, w1;
      int id;
      MSDA_BCAST3(pp, SMP, w0, w1, id);
      // a corner of weight 0 is not read (it may lie outside the tensor): its offset is out of range, the load returns 0, it contributes 0 * 0
      const unsigned ld_b = (unsigned)a.ldv * 2u;
      const unsigned o00 = vb_bytes + (unsigned)(a.start[l] + id) * ld_b;      // (garbage when no corner is valid: never used in range then)
      const uint4 r00 = msda_buf_load16(rs, (w0 & 0xffffu) ? o00 : MSDA_OOB);
      const uint4 r01 = msda_buf_load16(rs, (w0 >> 16) ? o00 + ld_b : MSDA_OOB);
      const uint4 r10 = msda_buf_load16(rs, (w1 & 0xffffu) ? o00 + (unsigned)W * ld_b : MSDA_OOB);
      const uint4 r11 = msda_buf_load16(rs, (w1 >> 16) ? o00 + (unsigned)(W + 1) * ld_b : MSDA_OOB);
      msda_dot_row<T>(acc, r00, r01, w0);
      msda_dot_row<T>(acc, r10, r11, w1);
    } else {
      float c00, c01, c10, c11;
      int id;
      MSDA_BCAST5(pp, SMP, c00, c01, c10, c11, id);
      const T* p00 = vb + ((long long)a.start[l] + id) * a.ldv;
      float v[8];
      if (c00 != 0.f) { load8<T>(p00, v); axpy8<T>(acc, c00, v); }
      if (c01 != 0.f) { load8<T>(p00 + a.ldv, v); axpy8<T>(acc, c01, v); }
      if (c10 != 0.f) { load8<T>(p00 + (long long)W * a.ldv, v); axpy8<T>(acc, c10, v); }
      if (c11 != 0.f) { load8<T>(p00 + (long long)(W + 1) * a.ldv, v); axpy8<T>(acc, c11, v); }
    }
    msda_gather_global<T, L, P, SMP + 1>(a, pp, vb, rs, vb_bytes, acc);
  }
}

// The LDS gather: a RUNTIME loop over the prepared slots, four samples (one per preparing lane of the quad) per iteration,
// with the slot registers shifted down after each iteration so that every DPP broadcast reads slot 0.  (Fully unrolled --
// 4 * L * P branch-free ds_read_b128 -- the compiler hoists the reads to the top, wants ~350 registers under the
// 128-register cap of a 16-wave block and spills hundreds of them; a scheduling barrier per sample made it worse.)
template <class T, int L, int P>
__device__ __forceinline__ void msda_gather_lds(const MsdaArgs& a, MsdaPrep<L, P>& pp, const unsigned char* vslab_sub, float (&acc)[8]) {
  constexpr int LP = L * P, NS = MsdaPrep<L, P>::NS;
#pragma unroll 1
  for (int j = 0; j < NS; ++j) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int smp = 4 * j + k;                       // wave-uniform
      if (smp < LP) {
        int l = 0;
#pragma unroll
        for (int t = 1; t < L; ++t) l += smp >= t * P ? 1 : 0;
        int start = a.start[0], W = a.w[0];
#pragma unroll
        for (int t = 1; t < L; ++t)
          if (l == t) { start = a.start[t]; W = a.w[t]; }
        unsigned w0, w1;
        int id;
        switch (k) {
          case 0: MSDA_BCAST3(pp, 0, w0, w1, id); break;
          case 1: MSDA_BCAST3(pp, 1, w0, w1, id); break;
          case 2: MSDA_BCAST3(pp, 2, w0, w1, id); break;
          default: MSDA_BCAST3(pp, 3, w0, w1, id); break;
        }
        const unsigned char* p00 = vslab_sub + (start + id) * MSDA_FWD_PITCH;
        const unsigned char* p10 = p00 + W * MSDA_FWD_PITCH;
        const uint4 r00 = *reinterpret_cast<const uint4*>(p00);
        const uint4 r01 = *reinterpret_cast<const uint4*>(p00 + MSDA_FWD_PITCH);
        const uint4 r10 = *reinterpret_cast<const uint4*>(p10);
        const uint4 r11 = *reinterpret_cast<const uint4*>(p10 + MSDA_FWD_PITCH);
        msda_dot_row<T>(acc, r00, r01, w0);
        msda_dot_row<T>(acc, r10, r11, w1);
      }
    }
#pragma unroll
    for (int i = 0; i + 1 < NS; ++i) {                // next slot becomes slot 0
      pp.wp0[i] = pp.wp0[i + 1]; pp.wp1[i] = pp.wp1[i + 1]; pp.idx[i] = pp.idx[i + 1];
    }
  }
}

// Gather straight from global memory (L2): fp32 maps, slabs that do not fit in LDS (512x512 tiles), small launches.
template <class T, int L, int P>
__global__ __launch_bounds__(256) void msda_fwd_kernel(MsdaArgs a) {
  const int lane = threadIdx.x & 63;
  const long long pair_raw = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (lane >> 2);
  const long long total = (long long)a.B * a.Lq * a.M;
  const bool live = pair_raw < total;
  const long long pair = live ? pair_raw : total - 1;   // tail lanes shadow a live pair: the quad shuffles stay defined
  const int sub = lane & 3;
  // (32-bit divisions when the pair count allows: a 64-bit one is ~100 instructions, common.hpp unravel)
  const bool small = total <= 0xffffffffll;
  const long long bq = small ? (long long)((unsigned)pair / (unsigned)a.M) : pair / a.M;
  const int m = (int)(pair - bq * a.M);
  const int b = small ? (int)((unsigned)bq / (unsigned)a.Lq) : (int)(bq / a.Lq);
  const int q = (int)(bq - (long long)b * a.Lq);
  MsdaPrep<L, P> pp;
  pp.run(a, a.offw + bq * a.ldo, a.ref + (long long)b * a.ref_bs + (long long)q * a.ref_L * 2, m, sub, live);
  if constexpr (sizeof(T) == 2) pp.template pack16<T>();
  const T* vb = (const T*)a.value + (long long)b * a.v_bs + m * 32 + sub * 8;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  // (16-bit value types: the host guarantees the tensor spans less than 2 GiB, so byte offsets fit the descriptor's 32 bits with bit 31 free)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.value, 0, (int)0x80000000u, 0x00020000);
  const unsigned vb_bytes = (unsigned)(((long long)b * a.v_bs + m * 32 + sub * 8) * (long long)sizeof(T));
  msda_gather_global<T, L, P, 0>(a, pp, vb, rs, vb_bytes, acc);
  if (live) Vec8<T>::store((T*)a.out + bq * (a.M * 32) + m * 32 + sub * 8, acc);
}

// Forward with the (batch, head) value slab staged in LDS: one block per (batch, head, chunk of queries).  The global
// kernel gathers 64-byte head slices from L2 through the texture path (396 MB of L2 -> CU traffic per encoder call for
// 5.5 MB of value, ~64 B/clk/CU); from LDS the same corner reads are ds_read_b128 at up to 256 B/clk/CU.
//  * slab layout: [guard | Lv rows | guard], 64 bytes per pixel row, UNPADDED: a quad's 64-byte corner read covers one
//    aligned quarter of the 64 banks, so two of the four random pixels a ds_read_b128 lane group touches collide with
//    probability 1/4 -- with the 80-byte pitch of the first version the bank windows were unaligned and overlapped with
//    probability 7/16.  Zero guard bands of `guard` rows on both sides: a corner with weight 0 is still READ (branch-free
//    inner loop) and may fall up to W + 1 pixels outside its level -- into a neighbouring level's rows or a guard band,
//    always finite data, times 0;
//  * staging by LDS-DMA (global_load_lds_dwordx4: 1 KiB = 16 pixel rows per wave instruction, no staging registers, the
//    whole slab in flight at once -- the first version's load/store loop paid one memory round trip per iteration), and
//    the first pass's per-sample preparation (global loads of the offsets / logits, softmax, coordinates) overlaps it;
//  * block -> (batch, head, chunk): consecutive block ids go to the 8 XCDs round-robin, so the map sends the chunks of
//    one slab and the heads of one batch element to ONE XCD: the slab and the offset rows are fetched into one L2.
// Same arithmetic in the same order as msda_fwd_kernel: the two are bit-identical (tests/test_gpu_bench_shapes.py).
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* glb_ptr_t;
template <class T, int L, int P>
__global__ __launch_bounds__(1024) void msda_fwd_lds_kernel(MsdaArgs a, int q_per_block, int chunks, int guard, int probe) {
  static_assert(sizeof(T) == 2, "the staged slab is sized for 2-byte elements");
  extern __shared__ __attribute__((aligned(16))) unsigned char vslab_raw[];
  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
  const int bm = bid / chunks, chunk = bid - bm * chunks;
  const int b = bm / a.M, m = bm - b * a.M;
  unsigned char* vslab = vslab_raw + guard * MSDA_FWD_PITCH;       // row 0 of the value tensor
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane & 3, nwave = blockDim.x >> 6;
  const int q_begin = chunk * q_per_block;
  int q_end = q_begin + q_per_block;
  if (q_end > a.Lq) q_end = a.Lq;

  // ---- stage: piece k = pixels [16 k, 16 k + 16), lane i carries 16-byte part (i & 3) of pixel 16 k + (i >> 2) ----
  {
    const T* src = (const T*)a.value + (long long)b * a.v_bs + m * 32 + (lane & 3) * 8;
    const int npiece = (a.Lv + 15) >> 4;
    for (int k = wave; k < npiece; k += nwave) {
      const int pix = k * 16 + (lane >> 2);
      if (pix < a.Lv && !(probe & 2))
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + (long long)pix * a.ldv), (lds_ptr_t)(vslab + k * 1024), 16, 0, 0);
    }
    for (int i = threadIdx.x; i < guard * (MSDA_FWD_PITCH / 16); i += blockDim.x) {       // zero guard bands
      *reinterpret_cast<uint4*>(vslab_raw + i * 16) = make_uint4(0, 0, 0, 0);
      *reinterpret_cast<uint4*>(vslab + a.Lv * MSDA_FWD_PITCH + i * 16) = make_uint4(0, 0, 0, 0);
    }
  }
  MsdaPrep<L, P> pp;
  int qw = q_begin + wave * 16;                   // this wave's first query of the pass (wave-uniform)
  auto prepare = [&]() {
    const int q = qw + (lane >> 2);
    const bool live = q < q_end;
    const int qq = live ? q : q_end - 1;          // tail lanes shadow the last query: the quad shuffles stay defined
    if (!(probe & 4)) { pp.run(a, a.offw + ((long long)b * a.Lq + qq) * a.ldo, a.ref + (long long)b * a.ref_bs + (long long)qq * a.ref_L * 2, m, sub, live); pp.template pack16<T>(); }
  };
  if (probe & 4) {
#pragma unroll
    for (int j = 0; j < MsdaPrep<L, P>::NS; ++j) { pp.w00[j] = 0.25f; pp.w01[j] = 0.25f; pp.w10[j] = 0.25f; pp.w11[j] = 0.25f; pp.idx[j] = (lane * 37 + j * 101) & 63; }
    pp.template pack16<T>();
  }
  if (qw < q_end) prepare();
  __syncthreads();                                // (drains the LDS-DMA: vmcnt(0) in front of the barrier)
  const unsigned char* vslab_sub = vslab + sub * 16;
  while (qw < q_end) {
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (!(probe & 1)) msda_gather_lds<T, L, P>(a, pp, vslab_sub, acc);
    const int q = qw + (lane >> 2);
    if (q < q_end) Vec8<T>::store((T*)a.out + ((long long)b * a.Lq + q) * (a.M * 32) + m * 32 + sub * 8, acc);
    qw += nwave * 16;
    if (qw < q_end) prepare();
  }
}

// Forward for pyramids whose (batch, head) slab does NOT fit in LDS (Lv = 5376 at 512x512: 344 KB): one block per (batch, head, BAND).
// Self-attention over the pyramid itself (Lq == Lv, queries stored level by level, row-major -- the encoder,
// transformer_encoder_decoder.py:230-239): band k of NB owns the query rows [k H_l / NB, (k+1) H_l / NB) of EVERY level, and those queries
// sample around their own position (sampling offsets are a few pixels: _reset_parameters puts them on a compass grid of radius 1..P),
// so the block stages, per level, only its rows +- `halo` (whole levels when they are that small): at 512x512 with 8 bands 22 of 64 +
// 18 of 32 + 16 of 16 rows = 152 KB instead of 344 KB.  Nothing is assumed about the offsets: a sample with a corner outside the staged
// rows gets weight 0 in the branch-free LDS loop (MsdaPrep::run<true>) and is gathered from global memory afterwards, quad by quad.
// The value tensor is then read from HBM / L2 once per band (the global kernel re-fetched it through L2 for every query: 1.67x the
// algorithmic bytes, profiles/r2d_pmc_traffic_cfg3.json) and the 72 corner reads per pair are ds_read_b128.
// Sums are accumulated in sample order except for missed samples, which come last: bit-identical to msda_fwd_kernel when nothing misses.
template <class T, int L, int P>
__global__ __launch_bounds__(1024) void msda_fwd_band_kernel(MsdaArgs a, int NB, int halo, int guard) {
  static_assert(sizeof(T) == 2, "the staged slab is sized for 2-byte elements");
  extern __shared__ __attribute__((aligned(16))) unsigned char vslab_raw[];
  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);       // the bands of a (batch, head) and the heads of a batch element on one XCD
  const int bm = bid / NB, band = bid - bm * NB;
  const int b = bm / a.M, m = bm - b * a.M;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane & 3, nwave = blockDim.x >> 6;
  // per level: own query rows [c0, c1), staged rows [lo, hi), slab offset (pixels, multiple of 16 = one DMA piece)
  int c0[L], c1[L], lo[L], hi[L], soff[L];
  int npx = guard;
  MsdaArgs ab = a;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    c0[l] = band * a.h[l] / NB;
    c1[l] = (band + 1) * a.h[l] / NB;
    lo[l] = c0[l] - halo < 0 ? 0 : c0[l] - halo;
    hi[l] = c1[l] + halo > a.h[l] ? a.h[l] : c1[l] + halo;
    soff[l] = (npx + 15) & ~15;
    npx = soff[l] + (hi[l] - lo[l]) * a.w[l];
    ab.start[l] = soff[l] - lo[l] * a.w[l];        // slab pixel of the level's (0, 0): pixel (y, x) sits at start + y W + x for lo <= y < hi
  }
  // ---- stage: zero guards + the gaps between levels, DMA the bands (piece = 16 pixels = 1 KiB per wave instruction) ----
  {
    const int total16 = ((npx + guard + 15) >> 4);
    for (int i = threadIdx.x; i < soff[0] * 4; i += blockDim.x) *reinterpret_cast<uint4*>(vslab_raw + i * 16) = make_uint4(0, 0, 0, 0);   // front guard (up to the first piece boundary)
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const int n = (hi[l] - lo[l]) * a.w[l];
      const int endpx = soff[l] + n;
      const int nextpx = l + 1 < L ? ((endpx + 15) & ~15) : (total16 << 4);      // zero [endpx, next level's offset / end of slab)
      for (int i = endpx * 4 + threadIdx.x; i < nextpx * 4; i += blockDim.x) *reinterpret_cast<uint4*>(vslab_raw + i * 16) = make_uint4(0, 0, 0, 0);
      const T* src = (const T*)a.value + (long long)b * a.v_bs + (long long)(a.start[l] + lo[l] * a.w[l]) * a.ldv + m * 32 + (lane & 3) * 8;
      const int npiece = (n + 15) >> 4;
      for (int k = wave; k < npiece; k += nwave) {
        const int pix = k * 16 + (lane >> 2);
        if (pix < n)
          __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + (long long)pix * a.ldv), (lds_ptr_t)(vslab_raw + (soff[l] + k * 16) * MSDA_FWD_PITCH), 16, 0, 0);
      }
    }
  }
  MsdaPrep<L, P> pp;
  const unsigned char* vslab_sub = vslab_raw + sub * 16;
  const __amdgpu_buffer_rsrc_t rs_val = __builtin_amdgcn_make_buffer_rsrc((void*)a.value, 0, (int)0x80000000u, 0x00020000);      // (misses: see msda_buf_load16)
  bool staged = false;
  // the block's queries: its rows of every level, walked as ONE index space (a wave's 16 queries may straddle two levels)
  int seg_n[L], seg_q0[L];
  int nq = 0;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    seg_n[l] = (c1[l] - c0[l]) * a.w[l];
    seg_q0[l] = a.start[l] + c0[l] * a.w[l];
    nq += seg_n[l];
  }
  {
    for (int qw = wave * 16; qw < nq; qw += nwave * 16) {
      const int qi_raw = qw + (lane >> 2);
      const bool live = qi_raw < nq;
      const int qi = live ? qi_raw : nq - 1;         // tail lanes shadow the last query: the quad shuffles stay defined
      int q = seg_q0[0] + qi, rem = qi - seg_n[0];
#pragma unroll
      for (int t = 1; t < L; ++t) {
        if (rem >= 0) q = seg_q0[t] + rem;
        rem -= seg_n[t];
      }
      const int qq = q;
      const float* row = a.offw + ((long long)b * a.Lq + qq) * a.ldo;
      const float* refp = a.ref + (long long)b * a.ref_bs + (long long)qq * a.ref_L * 2;
      pp.template run<true>(a, row, refp, m, sub, live, lo, hi);
      pp.template pack16<T>();
      unsigned mm = pp.miss;                        // (the gather below shifts the slots down: read the mask first)
      if (!staged) { __syncthreads(); staged = true; }        // drains the LDS-DMA (vmcnt(0) in front of the barrier)
      float acc[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = 0.f;
      msda_gather_lds<T, L, P>(ab, pp, vslab_sub, acc);
      // ---- samples that missed the staged rows: from global memory, one per quad and round ----
      while (__any(mm != 0u)) {
        // the lowest lane of the quad that still has a miss owns this round
        const int has = mm != 0u ? 1 : 0;
        const int h0 = quad_bcast<0>(has), h1 = quad_bcast<1>(has), h2 = quad_bcast<2>(has), h3 = quad_bcast<3>(has);
        const int owner = h0 ? 0 : h1 ? 1 : h2 ? 2 : 3;
        const bool quad_has = (h0 | h1 | h2 | h3) != 0;
        float c00 = 0.f, c01 = 0.f, c10 = 0.f, c11 = 0.f;
        int id = 0, lev = 0;
        if (quad_has && sub == owner) {
          const int j = __ffs((int)mm) - 1;
          pp.one(a, row, refp, m, sub + 4 * j, c00, c01, c10, c11, id, lev);
          mm &= mm - 1u;
        }
        const int src_lane = (lane & ~3) + owner;
        c00 = __shfl(c00, src_lane, 64); c01 = __shfl(c01, src_lane, 64); c10 = __shfl(c10, src_lane, 64); c11 = __shfl(c11, src_lane, 64);
        id = __shfl(id, src_lane, 64); lev = __shfl(lev, src_lane, 64);
        if (quad_has) {
          int W = a.w[0], st0 = a.start[0];
#pragma unroll
          for (int t = 1; t < L; ++t)
            if (lev == t) { W = a.w[t]; st0 = a.start[t]; }
          // same rounded weights and dot products as the staged samples (a sample's contribution does not depend on where it was read)
          unsigned w0, w1;
          if constexpr (std::is_same<T, bf16_t>::value) { w0 = pack_bf16x2(c00, c01); w1 = pack_bf16x2(c10, c11); }
          else {      // (the same instruction as MsdaPrep::pack16)
            asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w0) : "v"(c00), "v"(c01));
            asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w1) : "v"(c10), "v"(c11));
          }
          const unsigned ld_b = (unsigned)a.ldv * 2u;
          const unsigned o00 = (unsigned)(((long long)b * a.v_bs + m * 32 + sub * 8) * 2) + (unsigned)(st0 + id) * ld_b;
          const uint4 r00 = msda_buf_load16(rs_val, (w0 & 0xffffu) ? o00 : MSDA_OOB);
          const uint4 r01 = msda_buf_load16(rs_val, (w0 >> 16) ? o00 + ld_b : MSDA_OOB);
          const uint4 r10 = msda_buf_load16(rs_val, (w1 & 0xffffu) ? o00 + (unsigned)W * ld_b : MSDA_OOB);
          const uint4 r11 = msda_buf_load16(rs_val, (w1 >> 16) ? o00 + (unsigned)(W + 1) * ld_b : MSDA_OOB);
          msda_dot_row<T>(acc, r00, r01, w0);
          msda_dot_row<T>(acc, r10, r11, w1);
        }
      }
      if (live) Vec8<T>::store((T*)a.out + ((long long)b * a.Lq + q) * (a.M * 32) + m * 32 + sub * 8, acc);
    }
  }
  if (!staged) __syncthreads();
}

// <dout, value corner> over a lane's 8 channels.  bf16: straight from the packed 16-byte loads with v_dot2c_f32_bf16
// (the gradient kernel is VALU-bound: unpack + fma per element cost 4x the instructions).
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
template <class T>
struct Dot8;
template <>
struct Dot8<bf16_t> {
  typedef uint4 Raw;
  static __device__ __forceinline__ Raw load(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
  static __device__ __forceinline__ Raw zero() { return make_uint4(0, 0, 0, 0); }
  static __device__ __forceinline__ float dot(const Raw& a, const Raw& b) {
    float acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.x), __builtin_bit_cast(bf16x2_t, b.x), 0.f, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.y), __builtin_bit_cast(bf16x2_t, b.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.z), __builtin_bit_cast(bf16x2_t, b.z), acc, false);
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.w), __builtin_bit_cast(bf16x2_t, b.w), acc, false);
  }
};
template <>
struct Dot8<float> {
  struct Raw { float v[8]; };
  static __device__ __forceinline__ Raw load(const float* p) { Raw r; Vec8<float>::load(p, r.v); return r; }
  static __device__ __forceinline__ Raw zero() { Raw r; for (int e = 0; e < 8; ++e) r.v[e] = 0.f; return r; }
  static __device__ __forceinline__ float dot(const Raw& a, const Raw& b) {
    float acc = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc = fmaf(a.v[e], b.v[e], acc);
    return acc;
  }
};

__device__ __forceinline__ float quad_sum(float v) {  // sum over the 4 lanes that share one (q, head)
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  return v;
}

// Backward with recompute.  grad_value is scattered with fp32 atomics (v1; LDS-privatised slabs are the planned
// follow-up); offset/logit gradients are written by lane sub==0 of each quad; the reference-point
// gradient (decoder only) is reduced over the 8 heads of a query inside the wave.
template <class T, int L, int P, bool ATOMIC_DV>
__global__ __launch_bounds__(256) void msda_bwd_kernel(MsdaArgs a) {
  constexpr int LP = L * P;
  const int lane = threadIdx.x & 63;
  const long long pair_raw = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (lane >> 2);
  const long long total = (long long)a.B * a.Lq * a.M;
  const bool live = pair_raw < total;
  const long long pair = live ? pair_raw : total - 1;   // dead lanes shadow a live pair so shuffles stay full-wave
  const int sub = lane & 3;
  // (32-bit divisions when the pair count allows: a 64-bit one is ~100 instructions, common.hpp unravel)
  const bool small = total <= 0xffffffffll;
  const long long bq = small ? (long long)((unsigned)pair / (unsigned)a.M) : pair / a.M;
  const int m = (int)(pair - bq * a.M);
  const int b = small ? (int)((unsigned)bq / (unsigned)a.Lq) : (int)(bq / a.Lq);
  const int q = (int)(bq - (long long)b * a.Lq);

  const float* row = a.offw + bq * a.ldo;
  const float* offp = row + m * LP * 2;
  const float* logp = row + a.M * LP * 2 + m * LP;
  float pr[LP];
  float mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < LP; ++i) { pr[i] = logp[i]; mx = fmaxf(mx, pr[i]); }
  float den = 0.f;
#pragma unroll
  for (int i = 0; i < LP; ++i) { pr[i] = __expf(pr[i] - mx); den += pr[i]; }
  const float inv = 1.f / den;
#pragma unroll
  for (int i = 0; i < LP; ++i) pr[i] *= inv;

  float go[8];
  typename Dot8<T>::Raw go_raw = Dot8<T>::zero();
  if constexpr (ATOMIC_DV) {
    load8<T>((const T*)a.dout + bq * (a.M * 32) + m * 32 + sub * 8, go);
    if (!live) {
#pragma unroll
      for (int e = 0; e < 8; ++e) go[e] = 0.f;
    }
  } else {
    if (live) go_raw = Dot8<T>::load((const T*)a.dout + bq * (a.M * 32) + m * 32 + sub * 8);
  }

  const T* vb = (const T*)a.value + (long long)b * a.v_bs + m * 32 + sub * 8;
  float* gvb = a.dvalue + (long long)b * a.dv_bs + m * 32 + sub * 8;
  const float* refp = a.ref + (long long)b * a.ref_bs + (long long)q * a.ref_L * 2;
  const int rls = a.ref_L == 1 ? 0 : 2;
  const int ldg = a.M * 32;

  float dA[LP];          // d loss / d attention prob
  float gx[LP], gy[LP];  // d loss / d pixel coordinate (== d offset)
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int H = a.h[l], W = a.w[l];
    const float rx = refp[l * rls], ry = refp[l * rls + 1];
    const T* vl = vb + (long long)a.start[l] * a.ldv;
    float* gl = gvb + (long long)a.start[l] * ldg;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int i = l * P + p;
      const float2 o = *reinterpret_cast<const float2*>(offp + i * 2);
      const float x = (rx + o.x * a.inv_w[l]) * (float)W - 0.5f;
      const float y = (ry + o.y * a.inv_h[l]) * (float)H - 0.5f;
      const float xf = floorf(x), yf = floorf(y);
      const float lx = x - xf, ly = y - yf;
      const int x0 = (int)xf, y0 = (int)yf;
      const float aw = pr[i];
      const bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)(x0 + 1) < (unsigned)W;
      const bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)(y0 + 1) < (unsigned)H;
      float d00 = 0.f, d01 = 0.f, d10 = 0.f, d11 = 0.f;   // <dout, v_corner> over this lane's 8 channels
      if constexpr (!ATOMIC_DV) {
        const long long o00 = (long long)y0 * W + x0;
        if (vy0 && vx0) d00 = Dot8<T>::dot(go_raw, Dot8<T>::load(vl + o00 * a.ldv));
        if (vy0 && vx1) d01 = Dot8<T>::dot(go_raw, Dot8<T>::load(vl + (o00 + 1) * a.ldv));
        if (vy1 && vx0) d10 = Dot8<T>::dot(go_raw, Dot8<T>::load(vl + (o00 + W) * a.ldv));
        if (vy1 && vx1) d11 = Dot8<T>::dot(go_raw, Dot8<T>::load(vl + (o00 + W + 1) * a.ldv));
      } else {
      float v[8];
      if (vy0 && vx0) {
        const long long off = (long long)y0 * W + x0;
        load8<T>(vl + off * a.ldv, v);
        const float c = aw * (1.f - ly) * (1.f - lx);
#pragma unroll
        for (int e = 0; e < 8; ++e) { d00 = fmaf(go[e], v[e], d00); if (live) atomicAdd(gl + off * ldg + e, c * go[e]); }
      }
      if (vy0 && vx1) {
        const long long off = (long long)y0 * W + x0 + 1;
        load8<T>(vl + off * a.ldv, v);
        const float c = aw * (1.f - ly) * lx;
#pragma unroll
        for (int e = 0; e < 8; ++e) { d01 = fmaf(go[e], v[e], d01); if (live) atomicAdd(gl + off * ldg + e, c * go[e]); }
      }
      if (vy1 && vx0) {
        const long long off = (long long)(y0 + 1) * W + x0;
        load8<T>(vl + off * a.ldv, v);
        const float c = aw * ly * (1.f - lx);
#pragma unroll
        for (int e = 0; e < 8; ++e) { d10 = fmaf(go[e], v[e], d10); if (live) atomicAdd(gl + off * ldg + e, c * go[e]); }
      }
      if (vy1 && vx1) {
        const long long off = (long long)(y0 + 1) * W + x0 + 1;
        load8<T>(vl + off * a.ldv, v);
        const float c = aw * ly * lx;
#pragma unroll
        for (int e = 0; e < 8; ++e) { d11 = fmaf(go[e], v[e], d11); if (live) atomicAdd(gl + off * ldg + e, c * go[e]); }
      }
      }
      // dA, gx, gy are linear in the corner dots: combine per lane first, then 3 quad reductions instead of 4
      dA[i] = quad_sum((1.f - ly) * ((1.f - lx) * d00 + lx * d01) + ly * ((1.f - lx) * d10 + lx * d11));
      gx[i] = aw * quad_sum((1.f - ly) * (d01 - d00) + ly * (d11 - d10));
      gy[i] = aw * quad_sum((1.f - lx) * (d10 - d00) + lx * (d11 - d01));
    }
  }
  // softmax backward: dlogit_i = p_i * (dA_i - sum_j p_j dA_j)
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < LP; ++i) dot = fmaf(pr[i], dA[i], dot);
  if (!ATOMIC_DV && live && sub == 0) {
    float* pp = a.probs + bq * (a.M * LP) + m * LP;
#pragma unroll
    for (int i = 0; i < LP; ++i) pp[i] = pr[i];
  }
  if (live && sub == 0) {
#pragma unroll
    for (int i = 0; i < LP; ++i) store_doffw<T>(a, bq, (m * LP + i) * 2, a.M * LP * 2 + m * LP + i, gx[i], gy[i], pr[i] * (dA[i] - dot));
  }
  if (a.dref) {
    // loc = ref + off/W  =>  d ref_x = W * d x ; sum over points, then over the 8 heads (lane bits 2..4; needs M == 8)
    float tx = 0.f, ty = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      float sx = 0.f, sy = 0.f;
#pragma unroll
      for (int p = 0; p < P; ++p) { sx += gx[l * P + p]; sy += gy[l * P + p]; }
      sx *= (float)a.w[l]; sy *= (float)a.h[l];
      if (!live) { sx = 0.f; sy = 0.f; }
      sx += __shfl_xor(sx, 4, 64); sx += __shfl_xor(sx, 8, 64); sx += __shfl_xor(sx, 16, 64);
      sy += __shfl_xor(sy, 4, 64); sy += __shfl_xor(sy, 8, 64); sy += __shfl_xor(sy, 16, 64);
      tx += sx; ty += sy;
      if (a.ref_L != 1 && live && sub == 0 && m == 0)
        *reinterpret_cast<float2*>(a.dref + (bq * L + l) * 2) = make_float2(sx, sy);
    }
    if (a.ref_L == 1 && live && sub == 0 && m == 0) *reinterpret_cast<float2*>(a.dref + bq * 2) = make_float2(tx, ty);
  }
}

// Gradient kernel with the (batch, head) value slab staged in LDS -- the backward counterpart of msda_fwd_lds_kernel, same
// slab (LDS-DMA, unpadded rows, zero guard bands), same block -> (batch, head, chunk) map, same quad-cooperative layout:
//   * lane `sub` of a quad prepares samples sub, sub + 4, ... of its (query, head) pair: probability, bilinear fractions,
//     validity of the four corners, pixel index;
//   * for every sample all four lanes read their 8 channels of the four corners from LDS (branch-free; invalid corners are
//     read from harmless addresses and dropped by the owner) and dot them with their 8 channels of dout (v_dot2 on the packed
//     bf16), four DPP quad reductions give every lane the four corner dots, and the OWNER lane keeps them;
//   * after the loop each lane turns its own samples' dots into d prob, d x, d y, the quad reduces sum_i p_i dA_i for the
//     softmax backward, and each lane writes its samples' offset / logit gradients and probabilities.
// The first version (msda_bwd_kernel, still used for fp32 maps, slabs that do not fit and the decoder's dref) gathers the
// corners from L2 through the texture path and repeats the per-sample arithmetic in all four lanes.
// BAND = true: the row-band form of msda_fwd_band_kernel (pyramids whose slab does not fit in LDS; q_per_block = number of bands NB,
// chunks = halo rows): the block owns the query rows of band k of every level and stages those rows +- halo; a sample with a valid
// corner outside the staged rows takes its four corner dots from global memory after the LDS pass.
template <class T, int L, int P, bool BAND = false>
__global__ __launch_bounds__(1024) void msda_bwd_lds_kernel(MsdaArgs a, int q_per_block, int chunks, int guard) {
  static_assert(sizeof(T) == 2, "the staged slab is sized for 2-byte elements");
  constexpr int LP = L * P, NS = (LP + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char vslab_raw[];
  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
  const int nper = BAND ? q_per_block : chunks;       // blocks per (batch, head)
  const int bm = bid / nper, chunk = bid - bm * nper;
  const int b = bm / a.M, m = bm - b * a.M;
  unsigned char* vslab = vslab_raw + (BAND ? 0 : guard * MSDA_FWD_PITCH);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane & 3, nwave = blockDim.x >> 6;
  int q_begin = chunk * q_per_block;
  int q_end = q_begin + q_per_block;
  if (q_end > a.Lq) q_end = a.Lq;
  // band form: per level own query rows [c0, c1), staged rows [lo, hi), slab offset; ab.start = slab pixel of the level's (0, 0)
  int lo[L], hi[L], seg_n[L], seg_q0[L];
  MsdaArgs ab = a;
  if constexpr (BAND) {
    const int NB = q_per_block, halo = chunks;
    int soff[L];
    int npx = guard;
    int nq = 0;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const int c0 = chunk * a.h[l] / NB, c1 = (chunk + 1) * a.h[l] / NB;
      lo[l] = c0 - halo < 0 ? 0 : c0 - halo;
      hi[l] = c1 + halo > a.h[l] ? a.h[l] : c1 + halo;
      soff[l] = (npx + 15) & ~15;
      npx = soff[l] + (hi[l] - lo[l]) * a.w[l];
      ab.start[l] = soff[l] - lo[l] * a.w[l];
      seg_n[l] = (c1 - c0) * a.w[l];
      seg_q0[l] = a.start[l] + c0 * a.w[l];
      nq += seg_n[l];
    }
    q_begin = 0;
    q_end = nq;                                          // the block's queries as one index space (mapped to token rows below)
    const int total16 = ((npx + guard + 15) >> 4);
    for (int i = threadIdx.x; i < soff[0] * 4; i += blockDim.x) *reinterpret_cast<uint4*>(vslab_raw + i * 16) = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const int n = (hi[l] - lo[l]) * a.w[l];
      const int endpx = soff[l] + n;
      const int nextpx = l + 1 < L ? ((endpx + 15) & ~15) : (total16 << 4);
      for (int i = endpx * 4 + threadIdx.x; i < nextpx * 4; i += blockDim.x) *reinterpret_cast<uint4*>(vslab_raw + i * 16) = make_uint4(0, 0, 0, 0);
      const T* src = (const T*)a.value + (long long)b * a.v_bs + (long long)(a.start[l] + lo[l] * a.w[l]) * a.ldv + m * 32 + (lane & 3) * 8;
      const int npiece = (n + 15) >> 4;
      for (int k = wave; k < npiece; k += nwave) {
        const int pix = k * 16 + (lane >> 2);
        if (pix < n)
          __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + (long long)pix * a.ldv), (lds_ptr_t)(vslab_raw + (soff[l] + k * 16) * MSDA_FWD_PITCH), 16, 0, 0);
      }
    }
  } else {
    const T* src = (const T*)a.value + (long long)b * a.v_bs + m * 32 + (lane & 3) * 8;
    const int npiece = (a.Lv + 15) >> 4;
    for (int k = wave; k < npiece; k += nwave) {
      const int pix = k * 16 + (lane >> 2);
      if (pix < a.Lv)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + (long long)pix * a.ldv), (lds_ptr_t)(vslab + k * 1024), 16, 0, 0);
    }
    for (int i = threadIdx.x; i < guard * (MSDA_FWD_PITCH / 16); i += blockDim.x) {
      *reinterpret_cast<uint4*>(vslab_raw + i * 16) = make_uint4(0, 0, 0, 0);
      *reinterpret_cast<uint4*>(vslab + a.Lv * MSDA_FWD_PITCH + i * 16) = make_uint4(0, 0, 0, 0);
    }
  }
  const unsigned char* vslab_sub = vslab + sub * 16;
  const int rls = a.ref_L == 1 ? 0 : 2;
  bool staged = false;
  unsigned gmax_bits = 0;
  for (int qw = q_begin + wave * 16; qw < q_end; qw += nwave * 16) {
    const int qi = qw + (lane >> 2);
    const bool live = qi < q_end;
    int q = live ? qi : q_end - 1;                                          // tail lanes shadow the last query
    if constexpr (BAND) {                                                   // band form: index into the block's rows of every level -> token row
      int rem = q - seg_n[0];
      q = seg_q0[0] + q;
#pragma unroll
      for (int t = 1; t < L; ++t) {
        if (rem >= 0) q = seg_q0[t] + rem;
        rem -= seg_n[t];
      }
    }
    const long long bq = (long long)b * a.Lq + q;
    const float* row = a.offw + bq * a.ldo;
    const float* offp = row + m * LP * 2;
    const float* logp = row + a.M * LP * 2 + m * LP;
    const float* refp = a.ref + (long long)b * a.ref_bs + (long long)q * a.ref_L * 2;
    // ---- this lane's samples ----
    float pr[NS], lx[NS], ly[NS];
    int idx[NS], vmask[NS];          // vmask: bit0 (y0,x0) bit1 (y0,x0+1) bit2 (y0+1,x0) bit3 (y0+1,x0+1) inside the map
    int idx_g[BAND ? NS : 1];        // band form: y0 * W + x0 of every sample (idx holds a harmless staged pixel for the ones that missed)
    unsigned miss = 0u;              // band form: bit j = sample sub + 4 j has a valid corner outside the staged rows
    {
      float lg[NS];
      float2 of[NS];
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        const int smp = sub + 4 * j;
        const bool has = smp < LP;
        lg[j] = has ? logp[smp] : -3.0e38f;
        of[j] = has ? *reinterpret_cast<const float2*>(offp + smp * 2) : make_float2(0.f, 0.f);
      }
      float rx[L], ry[L];
#pragma unroll
      for (int l = 0; l < L; ++l) { rx[l] = refp[l * rls]; ry[l] = refp[l * rls + 1]; }
      float mx = -3.0e38f;
#pragma unroll
      for (int j = 0; j < NS; ++j) mx = fmaxf(mx, lg[j]);
      mx = quad_max(mx);
      float den = 0.f;
#pragma unroll
      for (int j = 0; j < NS; ++j) { lg[j] = (sub + 4 * j < LP) ? __expf(lg[j] - mx) : 0.f; den += lg[j]; }
      den = quad_add(den);
      const float inv = 1.f / den;
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        const int smp = sub + 4 * j;
        int l = 0;
#pragma unroll
        for (int t = 1; t < L; ++t) l += smp >= t * P ? 1 : 0;
        int H = a.h[0], W = a.w[0];
        float ih = a.inv_h[0], iw = a.inv_w[0], rxl = rx[0], ryl = ry[0];
#pragma unroll
        for (int t = 1; t < L; ++t)
          if (l == t) { H = a.h[t]; W = a.w[t]; ih = a.inv_h[t]; iw = a.inv_w[t]; rxl = rx[t]; ryl = ry[t]; }
        const float x = (rxl + of[j].x * iw) * (float)W - 0.5f;
        const float y = (ryl + of[j].y * ih) * (float)H - 0.5f;
        const float xf = floorf(x), yf = floorf(y);
        lx[j] = x - xf; ly[j] = y - yf;
        const int x0 = (int)fminf(fmaxf(xf, -2.f), 16777216.f), y0 = (int)fminf(fmaxf(yf, -2.f), 16777216.f);
        pr[j] = lg[j] * inv;
        const bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)(x0 + 1) < (unsigned)W;
        const bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)(y0 + 1) < (unsigned)H;
        vmask[j] = smp < LP ? ((vy0 && vx0) ? 1 : 0) | ((vy0 && vx1) ? 2 : 0) | ((vy1 && vx0) ? 4 : 0) | ((vy1 && vx1) ? 8 : 0) : 0;
        idx[j] = vmask[j] ? y0 * W + x0 : 0;
        if constexpr (BAND) {
          int blo = lo[0], bhi = hi[0];
#pragma unroll
          for (int t = 1; t < L; ++t)
            if (l == t) { blo = lo[t]; bhi = hi[t]; }
          const bool in_band = (!vy0 || (y0 >= blo && y0 < bhi)) && (!vy1 || (y0 + 1 >= blo && y0 + 1 < bhi));
          idx_g[j] = idx[j];
          if (!vmask[j] || !in_band) {
            if (vmask[j] && live) miss |= 1u << j;
            idx[j] = blo * W;
          }
        }
      }
    }
    const uint4 go = live ? *reinterpret_cast<const uint4*>((const T*)a.dout + bq * (a.M * 32) + m * 32 + sub * 8) : make_uint4(0, 0, 0, 0);
    {   // running max |dout| of this (batch, head) slice for the scatter's fixed-point scale: bf16 magnitudes order like their bits
      const unsigned w4[4] = {go.x, go.y, go.z, go.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned lo = w4[i] & 0x7fffu, hi = (w4[i] >> 16) & 0x7fffu;
        gmax_bits = lo > gmax_bits ? lo : gmax_bits;
        gmax_bits = hi > gmax_bits ? hi : gmax_bits;
      }
    }
    if (!staged) { __syncthreads(); staged = true; }       // (drains the LDS-DMA; every wave has at least one pass: chunks hold >= 16 * nwave queries or the block's waves beyond the chunk skip the loop -- see below)
    // ---- corner dots of every sample, kept by the owner lane ----
    float d00[NS], d01[NS], d10[NS], d11[NS];
    int idx_run[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) idx_run[j] = idx[j];
#pragma unroll 1
    for (int j = 0; j < NS; ++j) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int smp = 4 * j + k;                       // wave-uniform
        if (smp < LP) {
          int l = 0;
#pragma unroll
          for (int t = 1; t < L; ++t) l += smp >= t * P ? 1 : 0;
          int start = ab.start[0], W = a.w[0];           // (ab == a unless BAND: slab pixel of the level's (0, 0))
#pragma unroll
          for (int t = 1; t < L; ++t)
            if (l == t) { start = ab.start[t]; W = a.w[t]; }
          int id;
          switch (k) {
            case 0: id = quad_bcast<0>(idx_run[0]); break;
            case 1: id = quad_bcast<1>(idx_run[0]); break;
            case 2: id = quad_bcast<2>(idx_run[0]); break;
            default: id = quad_bcast<3>(idx_run[0]); break;
          }
          const unsigned char* p00 = vslab_sub + (start + id) * MSDA_FWD_PITCH;
          const unsigned char* p10 = p00 + W * MSDA_FWD_PITCH;
          const float e00 = quad_add(Dot8<T>::dot(go, *reinterpret_cast<const uint4*>(p00)));
          const float e01 = quad_add(Dot8<T>::dot(go, *reinterpret_cast<const uint4*>(p00 + MSDA_FWD_PITCH)));
          const float e10 = quad_add(Dot8<T>::dot(go, *reinterpret_cast<const uint4*>(p10)));
          const float e11 = quad_add(Dot8<T>::dot(go, *reinterpret_cast<const uint4*>(p10 + MSDA_FWD_PITCH)));
          if (sub == k) { d00[NS - 1] = e00; d01[NS - 1] = e01; d10[NS - 1] = e10; d11[NS - 1] = e11; }
        }
      }
      // rotate: slot 0 of idx_run is always the current one; the dots land in slot NS-1 and move down with the rotation, so
      // after NS iterations slot j holds the dots of sample sub + 4 j
#pragma unroll
      for (int i = 0; i + 1 < NS; ++i) { idx_run[i] = idx_run[i + 1]; }
      if (j + 1 < NS) {
#pragma unroll
        for (int i = 0; i + 1 < NS; ++i) { d00[i] = d00[i + 1]; d01[i] = d01[i + 1]; d10[i] = d10[i + 1]; d11[i] = d11[i + 1]; }
      }
    }
    if constexpr (BAND) {
      // ---- samples that missed the staged rows: their four corner dots from global memory, one per quad and round ----
      unsigned mm = miss;
      while (__any(mm != 0u)) {
        const int has = mm != 0u ? 1 : 0;
        const int h0 = quad_bcast<0>(has), h1 = quad_bcast<1>(has), h2 = quad_bcast<2>(has), h3 = quad_bcast<3>(has);
        const int owner = h0 ? 0 : h1 ? 1 : h2 ? 2 : 3;
        const bool quad_has = (h0 | h1 | h2 | h3) != 0;
        const bool mine = quad_has && sub == owner;
        const int jm = mine ? __ffs((int)mm) - 1 : 0;
        int id = 0, vm = 0;
#pragma unroll
        for (int jj = 0; jj < NS; ++jj)
          if (mine && jj == jm) { id = idx_g[jj]; vm = vmask[jj]; }
        int lev = 0;
        {
          const int smp = sub + 4 * jm;
#pragma unroll
          for (int t = 1; t < L; ++t) lev += smp >= t * P ? 1 : 0;
        }
        const int src_lane = (lane & ~3) + owner;
        id = __shfl(id, src_lane, 64); vm = __shfl(vm, src_lane, 64); lev = __shfl(lev, src_lane, 64);
        int W = a.w[0], st0 = a.start[0];
#pragma unroll
        for (int t = 1; t < L; ++t)
          if (lev == t) { W = a.w[t]; st0 = a.start[t]; }
        const T* p00 = (const T*)a.value + (long long)b * a.v_bs + m * 32 + sub * 8 + ((long long)st0 + id) * a.ldv;
        const uint4 z4 = make_uint4(0, 0, 0, 0);
        const uint4 r00 = (vm & 1) ? *reinterpret_cast<const uint4*>(p00) : z4;
        const uint4 r01 = (vm & 2) ? *reinterpret_cast<const uint4*>(p00 + a.ldv) : z4;
        const uint4 r10 = (vm & 4) ? *reinterpret_cast<const uint4*>(p00 + (long long)W * a.ldv) : z4;
        const uint4 r11 = (vm & 8) ? *reinterpret_cast<const uint4*>(p00 + (long long)(W + 1) * a.ldv) : z4;
        const float e00 = quad_add(Dot8<T>::dot(go, r00)), e01 = quad_add(Dot8<T>::dot(go, r01));
        const float e10 = quad_add(Dot8<T>::dot(go, r10)), e11 = quad_add(Dot8<T>::dot(go, r11));
#pragma unroll
        for (int jj = 0; jj < NS; ++jj)
          if (mine && jj == jm) { d00[jj] = e00; d01[jj] = e01; d10[jj] = e10; d11[jj] = e11; }
        if (mine) mm &= mm - 1u;
      }
    }
    // ---- per-sample gradients of this lane's samples ----
    float dA[NS], gx[NS], gy[NS];
    float dotp = 0.f;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const float c00 = (vmask[j] & 1) ? d00[j] : 0.f, c01 = (vmask[j] & 2) ? d01[j] : 0.f;
      const float c10 = (vmask[j] & 4) ? d10[j] : 0.f, c11 = (vmask[j] & 8) ? d11[j] : 0.f;
      dA[j] = (1.f - ly[j]) * ((1.f - lx[j]) * c00 + lx[j] * c01) + ly[j] * ((1.f - lx[j]) * c10 + lx[j] * c11);
      gx[j] = pr[j] * ((1.f - ly[j]) * (c01 - c00) + ly[j] * (c11 - c10));
      gy[j] = pr[j] * ((1.f - lx[j]) * (c10 - c00) + lx[j] * (c11 - c01));
      dotp = fmaf(pr[j], dA[j], dotp);
    }
    dotp = quad_add(dotp);
    if (live) {
      float* pp = a.probs + bq * (a.M * LP) + m * LP;
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        const int smp = sub + 4 * j;
        if (smp < LP) {
          store_doffw<T>(a, bq, (m * LP + smp) * 2, a.M * LP * 2 + m * LP + smp, gx[j], gy[j], pr[j] * (dA[j] - dotp));
          pp[smp] = pr[j];
        }
      }
    }
    if (a.dref) {
      // reference-point gradient (decoder cross-attention): loc = ref + off / (W, H)  =>  d ref_x(l) = W_l * sum over the level's points
      // of d x; this quad holds one head's samples, the 8 heads of a query sit in 8 different blocks: fp32 atomics into the
      // caller-zeroed dref (8 addends per element)
      float sx[L], sy[L];
#pragma unroll
      for (int l = 0; l < L; ++l) { sx[l] = 0.f; sy[l] = 0.f; }
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        const int smp = sub + 4 * j;
        const int lev = smp / P;
#pragma unroll
        for (int l = 0; l < L; ++l) {
          const bool mine = smp < LP && lev == l;
          sx[l] += mine ? gx[j] : 0.f;
          sy[l] += mine ? gy[j] : 0.f;
        }
      }
      float tx = 0.f, ty = 0.f;
#pragma unroll
      for (int l = 0; l < L; ++l) {
        sx[l] = quad_add(sx[l]) * (float)a.w[l];
        sy[l] = quad_add(sy[l]) * (float)a.h[l];
        tx += sx[l]; ty += sy[l];
        if (a.ref_L != 1 && live && sub == 0) {
          atomicAdd(a.dref + (bq * L + l) * 2, sx[l]);
          atomicAdd(a.dref + (bq * L + l) * 2 + 1, sy[l]);
        }
      }
      if (a.ref_L == 1 && live && sub == 0) {
        atomicAdd(a.dref + bq * 2, tx);
        atomicAdd(a.dref + bq * 2 + 1, ty);
      }
    }
  }
  if (!staged) __syncthreads();        // waves without a query still take part in the block's one barrier
  if (a.gmax) {
    __shared__ unsigned gred[16];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)gmax_bits, o, 64); gmax_bits = t > gmax_bits ? t : gmax_bits; }
    if (lane == 0) gred[wave] = gmax_bits;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned mxb = 0;
      for (int i = 0; i < nwave; ++i) mxb = gred[i] > mxb ? gred[i] : mxb;
      a.gmax[bm * a.gmax_n + chunk] = __uint_as_float(mxb << 16);      // (a NaN / Inf magnitude stays NaN / Inf: the scatter writes zeros)
    }
  }
}

// max |dout| per (batch, head) for the scatter's fixed-point scale when the gradient kernel that ran before is the global-gather one (it
// does not leave it): block = (batch, 64 query rows), whole 512-byte rows, one partial per head and block -- the scatter blocks then read
// Lq / 64 partials instead of scanning Lq * 32 strided values each (50 of 179 us at Lq = 5376).
template <class T>
__global__ __launch_bounds__(256) void msda_absmax_kernel(const T* __restrict__ dout, int Lq, int M, int nparts, float* __restrict__ gmax) {
  __shared__ float red[64][8];                             // [row lane][head]: rows_per_pass = 256 / (4 M) <= 64 (M = 1), heads <= 8
  const int b = blockIdx.x / nparts, part = blockIdx.x % nparts;
  const int cpr = M * 4;                                   // 8-channel chunks per row
  const int rows_per_pass = 256 / cpr;                     // the host only sends M in {1, 2, 4, 8}: 256 % (4 M) == 0
  const int ch = threadIdx.x % cpr, rl = threadIdx.x / cpr;
  float mx = 0.f;
  if (rl < rows_per_pass)
    for (int r = part * 64 + rl; r < min(Lq, part * 64 + 64); r += rows_per_pass) {
      float v[8];
      Vec8<T>::load(dout + ((long long)b * Lq + r) * (M * 32) + ch * 8, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(v[e]));
    }
  mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 2, 64));                   // the 4 chunks of a head
  if ((ch & 3) == 0 && rl < rows_per_pass) red[rl][ch >> 2] = mx;      // EVERY row lane (M = 4: 16 of them; M = 8: 8)
  __syncthreads();
  if ((int)threadIdx.x < M) {
    float m2 = 0.f;
    for (int k = 0; k < rows_per_pass; ++k) m2 = fmaxf(m2, red[k][threadIdx.x]);
    gmax[((long long)b * M + threadIdx.x) * nparts + part] = m2;
  }
}

// d value via LDS-privatised scatter.  Float LDS atomics run at ~1 lane / 4 clk on gfx950 (measured: 245 clk per
// wave-level ds_add_f32) while integer LDS atomics are native rate, so contributions are accumulated in 32-bit fixed
// point.  The scale is safe by construction: a query adds at most |g| to any (pixel, channel) (its sample weights are
// probabilities times bilinear weights), so |sum| <= Lq * max|g|, and scale = 2^30 / (Lq * max|g|) taken over the
// block's (batch, head) slice cannot overflow; integer adds commute, so the result is bit-reproducible.
// One block per (batch, head, slab range); a range is a run of rows of ONE level (<= MSDA_MAX_NPIX pixels), so a block
// only walks that level's P samples of every query.  Its slab [npix][33] x int32 lives in LDS (odd pitch spreads
// pixels over banks).  Each half-wave takes QB = 32 / P queries per iteration: lane (qq, p) computes sample p of query qq
// once, then the QB * P samples are broadcast one by one (width-32 shuffles) and the 32 lanes, one per channel, add
// their corner contributions.
#define MSDA_SLAB_PITCH 33
template <class T, int L, int P>
__global__ __launch_bounds__(1024) void msda_bwd_value_lds_kernel(MsdaArgs a, int probe) {
  constexpr int LP = L * P;
  constexpr int QB = 32 / P;          // queries per half-wave iteration
  static_assert(P <= 32, "one lane per sample");
  extern __shared__ __attribute__((aligned(16))) int slab[];
  __shared__ float red[16];
  const int b = blockIdx.x / a.M, m = blockIdx.x % a.M;
  const int lev = a.g_level[blockIdx.y], pix0 = a.g_pix0[blockIdx.y], npix = a.g_npix[blockIdx.y];
  int H = a.h[0], W = a.w[0], lstart = a.start[0];
#pragma unroll
  for (int l = 1; l < L; ++l)
    if (lev == l) { H = a.h[l]; W = a.w[l]; lstart = a.start[l]; }
  int* slab_c = slab + a.g_guard * MSDA_SLAB_PITCH;      // pixel 0 of the block's range; [-guard, npix + guard) is allocated and zeroed
  for (int i = threadIdx.x; i < (npix + 2 * a.g_guard) * MSDA_SLAB_PITCH; i += blockDim.x) slab[i] = 0;
  const T* gbase = (const T*)a.dout + (long long)b * a.Lq * (a.M * 32) + m * 32;
  float mx = (probe & 1) ? 4.f : 0.f;
  if (a.gmax_n > 0) {        // left by the gradient kernel that has just read all of dout (the scan below costs 12 us of a 60 us kernel)
    for (int i = threadIdx.x; i < a.gmax_n; i += blockDim.x) {
      mx = fmaxf(mx, a.gmax[(long long)blockIdx.x * a.gmax_n + i]);
    }
  } else if (!(probe & 1)) {
    for (int i = threadIdx.x; i < a.Lq * 32; i += blockDim.x) mx = fmaxf(mx, fabsf(to_f32(gbase[(long long)(i >> 5) * (a.M * 32) + (i & 31)])));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) mx = fmaxf(mx, red[i]);
  const bool usable = mx > 0.f && mx < 3.0e38f;       // all-zero (or non-finite) gradients produce zeros
  const float scale = usable ? 1073741824.f / ((float)a.Lq * mx) : 0.f;
  const float inv_scale = usable ? ((float)a.Lq * mx) * (1.f / 1073741824.f) : 0.f;

  const int ch = threadIdx.x & 31;
  const int half = threadIdx.x >> 5, nhalf = blockDim.x >> 5;
  const int my_qq = ch / P, my_p = ch - my_qq * P;    // this lane's (query, point) in the per-sample prologue
  const float fW = (float)W, fH = (float)H, ifW = 1.f / fW, ifH = 1.f / fH;
  const int rel0 = lstart - pix0;                      // flat index of the level's pixel (0,0) relative to the slab
  // per-half-wave sample records (corner index + the 4 corner weights, probability and validity folded in), written by
  // the prologue lanes and read back by all 32 lanes with broadcast LDS reads: the step loop is bound by VALU issue, and
  // this replaces three cross-lane shuffles plus the per-lane floor / weight / validity arithmetic of every step
  float4* rec_w = reinterpret_cast<float4*>(slab + a.g_npix_max * MSDA_SLAB_PITCH) + half * 32;
  int* rec_f = reinterpret_cast<int*>(reinterpret_cast<float4*>(slab + a.g_npix_max * MSDA_SLAB_PITCH) + nhalf * 32) + half * 32;
  const int qs = a.g_qs[blockIdx.y], nqs = a.g_nqs[blockIdx.y];
  for (int q0 = (qs * nhalf + half) * QB; q0 < ((probe & 2) ? 0 : a.Lq); q0 += nqs * nhalf * QB) {
    {
      float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
      int f00 = 0;
      if (my_qq < QB && q0 + my_qq < a.Lq) {
        const long long bq = (long long)b * a.Lq + q0 + my_qq;
        const int smp = lev * P + my_p;
        const float2 o = *reinterpret_cast<const float2*>(a.offw + bq * a.ldo + (m * LP + smp) * 2);
        const float* refq = a.ref + (long long)b * a.ref_bs + (long long)(q0 + my_qq) * a.ref_L * 2 + (a.ref_L == 1 ? 0 : lev * 2);
        const float x = (refq[0] + o.x * ifW) * fW - 0.5f;
        const float y = (refq[1] + o.y * ifH) * fH - 0.5f;
        const float aw = a.probs[bq * (a.M * LP) + m * LP + smp];
        const float xf = floorf(x), yf = floorf(y);
        const float lx = x - xf, ly = y - yf;
        const int x0 = (int)xf, y0 = (int)yf;
        f00 = rel0 + y0 * W + x0;          // flat pixel index relative to this block's range
        const bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)(x0 + 1) < (unsigned)W;
        const bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)(y0 + 1) < (unsigned)H;
        w4.x = (vy0 && vx0 && (unsigned)f00 < (unsigned)npix) ? aw * (1.f - ly) * (1.f - lx) : 0.f;
        w4.y = (vy0 && vx1 && (unsigned)(f00 + 1) < (unsigned)npix) ? aw * (1.f - ly) * lx : 0.f;
        w4.z = (vy1 && vx0 && (unsigned)(f00 + W) < (unsigned)npix) ? aw * ly * (1.f - lx) : 0.f;
        w4.w = (vy1 && vx1 && (unsigned)(f00 + W + 1) < (unsigned)npix) ? aw * ly * lx : 0.f;
      }
      // a sample with a corner in the range has f00 in [-(W + 1), npix): its four cells stay inside the guarded slab, so they are all
      // added without a test (weight 0 adds 0).  Samples with nothing in the range are marked and skipped with ONE test.
      const bool hit_any = w4.x != 0.f || w4.y != 0.f || w4.z != 0.f || w4.w != 0.f;
      if (!hit_any) f00 = (int)0x80000000;
      if (probe & 128) {
        // knob msda_scatter_merge: consecutive points of one query whose bilinear footprint is the SAME 2 x 2 cells are added in registers
        // before they touch the LDS -- point p hands its four weights to point p + 1 and drops out (one test in the step loop skips it), so a run
        // of k such points costs 4 LDS atomics instead of 4 k.  Lanes are ordered (query, point): the neighbour is one lane up (wave_shl /
        // wave_shr DPP moves: VALU, not the LDS crossbar).  On the untrained compass offsets of _reset_parameters the coarser levels see runs
        // (steps of 1 px at level 0 are steps of 1/2 and 1/4 px there); once offsets have spread by >= 2 px hardly any
        // (profiles/r4_msda_scatter_merge_analysis.txt), which is why this is a knob and its measurement decides.
        const int f_next = __builtin_amdgcn_update_dpp(0, f00, 0x130, 0xf, 0xf, false);      // wave_shl:1: lane + 1's f00
        const bool gives = my_p + 1 < P && my_qq < QB && f00 != (int)0x80000000 && f_next == f00;
#pragma unroll
        for (int t = 1; t < P; ++t) {      // weights flow up a run, one point per step (point t takes what point t - 1 has gathered so far)
          const float ax = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, w4.x), 0x138, 0xf, 0xf, false));      // wave_shr:1: lane - 1
          const float ay = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, w4.y), 0x138, 0xf, 0xf, false));
          const float az = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, w4.z), 0x138, 0xf, 0xf, false));
          const float aw4 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, w4.w), 0x138, 0xf, 0xf, false));
          const bool prev_gives = __builtin_amdgcn_update_dpp(0, (int)gives, 0x138, 0xf, 0xf, false) != 0;
          if (my_p == t && prev_gives) { w4.x += ax; w4.y += ay; w4.z += az; w4.w += aw4; }
        }
        if (gives) f00 = (int)0x80000000;
      }
      // a wave whose 2 x QB queries put no sample of this level inside the block's pixel range has nothing to add: skip the
      // dout loads and the sample loop (wave-uniform branch).  Queries are stored row-major per level and sample near their
      // own reference point, so for a level cut into several ranges most waves of a block skip most of their iterations
      // (Lv = 5376: level 0 in 4 ranges, every block walked all 5376 queries' samples to keep a quarter of them).
      if (__ballot(hit_any) == 0ull && !(probe & 64)) continue;
      rec_w[ch] = w4;
      rec_f[ch] = f00;
    }
    float gq[QB];
#pragma unroll
    for (int qq = 0; qq < QB; ++qq)
      gq[qq] = q0 + qq < a.Lq ? to_f32(gbase[(long long)(q0 + qq) * (a.M * 32) + ch]) * scale : 0.f;
#pragma unroll
    for (int qq = 0; qq < QB; ++qq) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int src = qq * P + p;
        const int f00 = rec_f[src];            // same address in all 32 lanes: broadcast
        if (f00 != (int)0x80000000) {
          const float4 w4 = rec_w[src];
          int* cell = slab_c + f00 * MSDA_SLAB_PITCH + ch;
          const float g = gq[qq];
          atomicAdd(cell, __float2int_rn(g * w4.x));
          atomicAdd(cell + MSDA_SLAB_PITCH, __float2int_rn(g * w4.y));
          atomicAdd(cell + W * MSDA_SLAB_PITCH, __float2int_rn(g * w4.z));
          atomicAdd(cell + (W + 1) * MSDA_SLAB_PITCH, __float2int_rn(g * w4.w));
        }
      }
    }
  }
  __syncthreads();
  if (nqs > 1) {      // one of several blocks that share this range: leave the integer sums, the finalize kernel adds them up
    int* outi = a.part + (long long)blockIdx.x * a.part_stride + a.g_part[blockIdx.y];
    for (int i = threadIdx.x; i < npix * 32; i += blockDim.x) outi[i] = slab_c[(i >> 5) * MSDA_SLAB_PITCH + (i & 31)];
    return;
  }
  T* outp = (T*)a.dvalue_t + ((long long)b * a.Lv + pix0) * (a.M * 32) + m * 32;
  for (int i = threadIdx.x; i < npix * 32; i += blockDim.x) {
    const int pix = i >> 5, c = i & 31;
    outp[(long long)pix * (a.M * 32) + c] = from_f32<T>((float)slab_c[pix * MSDA_SLAB_PITCH + c] * inv_scale);
  }
}

// Ranges whose queries were split over several scatter blocks (the small levels of a large pyramid: every query hits them, so one
// block per range is the kernel's critical path): sum the blocks' integer slabs, undo the fixed-point scale, write the compute dtype.
// Same scale as the scatter blocks used: it only depends on max |dout| of the (batch, head) slice.
template <class T>
__global__ __launch_bounds__(256) void msda_bwd_value_finalize_kernel(MsdaArgs a) {
  __shared__ float red[4];
  const int b = blockIdx.x / a.M, m = blockIdx.x % a.M;
  const int r = blockIdx.y;
  float mx = 0.f;
  for (int i = threadIdx.x; i < a.gmax_n; i += blockDim.x) mx = fmaxf(mx, a.gmax[(long long)blockIdx.x * a.gmax_n + i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const bool usable = mx > 0.f && mx < 3.0e38f;
  const float inv_scale = usable ? ((float)a.Lq * mx) * (1.f / 1073741824.f) : 0.f;
  const int npix = a.f_npix[r], nq = a.f_nqs[r];
  const int* src = a.part + (long long)blockIdx.x * a.part_stride + a.f_part[r];
  T* outp = (T*)a.dvalue_t + ((long long)b * a.Lv + a.f_pix0[r]) * (a.M * 32) + m * 32;
  // blockIdx.z cuts the range's cells into chunks (a lone block per range ran 43 us on dependent loads); a thread sums 4 consecutive
  // channels of a pixel over the nq <= 8 slabs with 16-byte loads, all slabs' loads issued before the first add
  const int ncell4 = npix * 8;
  for (int i = blockIdx.z * blockDim.x + threadIdx.x; i < ncell4; i += gridDim.z * blockDim.x) {
    int4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = k < nq ? *reinterpret_cast<const int4*>(src + (long long)k * npix * 32 + i * 4) : make_int4(0, 0, 0, 0);
    int4 acc = v[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }      // (integer: exact, order-independent)
    const float o[4] = {(float)acc.x * inv_scale, (float)acc.y * inv_scale, (float)acc.z * inv_scale, (float)acc.w * inv_scale};
    Vec4<T>::store(outp + (long long)(i >> 3) * (a.M * 32) + (i & 7) * 4, o);
  }
}

// ---- value gradient as a matrix product (bf16) ---------------------------------------------------------------------------------------------------
// The scatter above spends one LDS atomic per (sample corner, CHANNEL): 32 lanes add g[q][ch] * w into the corner's 32 cells, 4 wave
// instructions of ~7 LDS cycles per sample, and that IS its time (96 atomics x 7 clk per query and level).  The channel
// dimension carries no sparsity -- every channel of a query goes to the same cells with the same weight -- so it belongs on the matrix core:
//     dvalue[p][ch] = sum_q  A[p][q] * dout[q][ch],       A[p][q] = sum over the (point, corner) pairs of query q that land on pixel p of  prob * bilinear weight
// per (batch, head, level).  Only the scalar weights are scattered (one LDS integer add per corner: 32 x fewer), into a [pixels][32 queries]
// tile of 2^-24 fixed-point integers (adds commute: bit-reproducible; a tile entry is <= 1, the level's probabilities sum to <= 1); the tile
// is then read as the A operand of v_mfma_f32_16x16x32_bf16 (16 pixels x 32 queries) against dout^T staged as [32 channels][32 queries].
// One block = one (batch, head) x one band of whole rows of one level (<= 256 pixels = 16 MFMA row tiles over 4 waves), walking ALL queries
// in chunks of 32; a chunk none of whose samples falls into the band is skipped after its sample arithmetic (queries are stored row-major
// per level and sample near their own reference point).  Every dvalue element is written exactly once, by the block that owns its pixel.
// Rounding: the summed weight of a (pixel, query) pair is rounded ONCE to bf16 (2^-9 relative), dout is bf16 already, products and sums are fp32.
struct MsdaMfPlan { int rows[4], nb[4], npix_cap; };
#define MSDA_MF_PITCH 36      // ints per pixel row of the weight tile: 32 queries + 4, so that the 16 rows of a fragment read (16 B per lane) cover all 64 banks
#define MSDA_MF_NPIX 512      // most pixels of a band: 32 MFMA row tiles over 4 consumer waves; TWO weight tiles of 72 KiB (one being filled, one being multiplied)
#define MSDA_MF_GP 40         // bf16 per channel row of dout^T
typedef __attribute__((ext_vector_type(8))) __bf16 msda_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float msda_f32x4_t;

// Work split inside a block (512 threads), one barrier per chunk of 32 queries:
//   waves 0..2  one thread per (query of the chunk, point): sample geometry, <= 4 integer LDS adds into weight tile (c & 1), a flag per touched row tile
//   wave  3     dout^T of the chunk into Gt (c & 1)
//   waves 4..7  chunk c - 1: every flagged row tile of weight tile ((c - 1) & 1) -> bf16 A fragment (and zeroes behind it) -> two MFMAs (channels 0..15, 16..31)
// The per-sample geometry is ~120 wave instructions (4 cycles each on a SIMD16) per chunk and producer wave -- as much as the MFMA side -- which
// is why the two sides run on different waves at the same time instead of one after the other (the first version: 65 us per encoder call).
template <int L, int P>
__global__ __launch_bounds__(512) void msda_bwd_value_mfma_kernel(MsdaArgs a, MsdaMfPlan pl) {
  constexpr int LP = L * P;
  static_assert(32 * P <= 192, "one thread of waves 0..2 per (query of the chunk, point)");
  extern __shared__ __attribute__((aligned(16))) int Ai[];                                          // [2][npix_cap][36] fixed-point weights
  const int asz = pl.npix_cap * MSDA_MF_PITCH;
  unsigned short* Gt = reinterpret_cast<unsigned short*>(Ai + 2 * asz);                              // [2][32 channels][40] dout^T of a chunk
  int* tflag = reinterpret_cast<int*>(Gt + 2 * 32 * MSDA_MF_GP);                                     // [2][4 consumer waves][8]: row tile w + 4 i has entries
  const int b = blockIdx.x / a.M, m = blockIdx.x % a.M;
  int lev = 0, bi = blockIdx.y;
#pragma unroll
  for (int l = 0; l + 1 < L; ++l)
    if (lev == l && bi >= pl.nb[l]) { bi -= pl.nb[l]; lev = l + 1; }
  int H = a.h[0], W = a.w[0], lstart = a.start[0], rows = pl.rows[0];
#pragma unroll
  for (int l = 1; l < L; ++l)
    if (lev == l) { H = a.h[l]; W = a.w[l]; lstart = a.start[l]; rows = pl.rows[l]; }
  const int r0 = bi * rows;
  const int nrows = H - r0 < rows ? H - r0 : rows;
  const int pix0 = r0 * W, npix = nrows * W;                  // level-relative first pixel / pixel count of the band
  const int ntile = (npix + 15) >> 4;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, lr = lane & 15, lg = lane >> 4;
  for (int i = t; i < 2 * asz / 4; i += 512) reinterpret_cast<int4*>(Ai)[i] = make_int4(0, 0, 0, 0);
  if (t < 64) tflag[t] = 0;

  const bool smp_role = t < 32 * P;
  const int sq = smp_role ? t / P : 0, sp = t - sq * P;       // sample role: query of the chunk, point
  const bool stg_role = wave == 3;                            // staging role: lane -> (query, 8-channel group) pairs lane and lane + 64
  const int gq = lane >> 2, gc = lane & 3;
  const float fW = (float)W, fH = (float)H, ifW = 1.f / fW, ifH = 1.f / fH;
  const int smp = lev * P + sp;
  const unsigned short* gbase = (const unsigned short*)a.dout + (long long)b * a.Lq * (a.M * 32) + m * 32 + gc * 8;
  const int nchunk = (a.Lq + 31) >> 5;

  // operands of SC = 4 chunks are loaded at once, one super-chunk ahead of their use
  constexpr int SC = 4;
  float2 o_n[SC];
  float rx_n[SC], ry_n[SC], aw_n[SC];
  uint4 g_n[SC][2];
  auto prefetch = [&](int sc) {
#pragma unroll
    for (int k = 0; k < SC; ++k) {
      const int q = (sc * SC + k) * 32 + sq;
      o_n[k] = make_float2(0.f, 0.f);
      rx_n[k] = 0.f; ry_n[k] = 0.f; aw_n[k] = 0.f;
      if (smp_role && q < a.Lq) {
        const long long bq = (long long)b * a.Lq + q;
        o_n[k] = *reinterpret_cast<const float2*>(a.offw + bq * a.ldo + (m * LP + smp) * 2);
        const float* refq = a.ref + (long long)b * a.ref_bs + (long long)q * a.ref_L * 2 + (a.ref_L == 1 ? 0 : lev * 2);
        rx_n[k] = refq[0];
        ry_n[k] = refq[1];
        aw_n[k] = a.probs[bq * (a.M * LP) + m * LP + smp];
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int qg = (sc * SC + k) * 32 + gq + 16 * h;
        g_n[k][h] = make_uint4(0, 0, 0, 0);
        if (stg_role && qg < a.Lq) g_n[k][h] = *reinterpret_cast<const uint4*>(gbase + (long long)qg * (a.M * 32));
      }
    }
  };
  prefetch(0);
  msda_f32x4_t acc[8][2];
#pragma unroll
  for (int i = 0; i < 8; ++i) { acc[i][0] = (msda_f32x4_t){0.f, 0.f, 0.f, 0.f}; acc[i][1] = acc[i][0]; }
  __syncthreads();

  const int nsc = (nchunk + 1 + SC - 1) / SC;          // one more step than chunks: the consumers run one chunk behind
  for (int sc = 0; sc < nsc; ++sc) {
    float2 o_c[SC];
    float rx_c[SC], ry_c[SC], aw_c[SC];
    uint4 g_c[SC][2];
#pragma unroll
    for (int k = 0; k < SC; ++k) { o_c[k] = o_n[k]; rx_c[k] = rx_n[k]; ry_c[k] = ry_n[k]; aw_c[k] = aw_n[k]; g_c[k][0] = g_n[k][0]; g_c[k][1] = g_n[k][1]; }
    if ((sc + 1) * SC < nchunk) prefetch(sc + 1);
#pragma unroll
    for (int k = 0; k < SC; ++k) {
      const int c = sc * SC + k;
      if (c > nchunk) break;
      if (wave < 4) {
        if (c < nchunk) {
          int* A = Ai + (c & 1) * asz;
          int* tf = tflag + (c & 1) * 32;
          if (smp_role) {
            const float2 o = o_c[k];
            const float aw = aw_c[k];
            // this thread's sample: cell index relative to the band and the four corner weights (probability folded in)
            const float x = (rx_c[k] + o.x * ifW) * fW - 0.5f;
            const float y = (ry_c[k] + o.y * ifH) * fH - 0.5f;
            const float xf = floorf(x), yf = floorf(y);
            const float lx = x - xf, ly = y - yf;
            const int x0 = (int)xf, y0 = (int)yf;
            const int f00 = y0 * W + x0 - pix0;
            const bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)(x0 + 1) < (unsigned)W;
            const bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)(y0 + 1) < (unsigned)H;
            const bool live = aw != 0.f && x > -2.f && y > -2.f && x < fW + 1.f && y < fH + 1.f;      // (also keeps NaN / huge coordinates away from the int conversions)
            const bool h00 = live && vy0 && vx0 && (unsigned)f00 < (unsigned)npix;
            const bool h01 = live && vy0 && vx1 && (unsigned)(f00 + 1) < (unsigned)npix;
            const bool h10 = live && vy1 && vx0 && (unsigned)(f00 + W) < (unsigned)npix;
            const bool h11 = live && vy1 && vx1 && (unsigned)(f00 + W + 1) < (unsigned)npix;
            const float s24 = 16777216.f * aw;
            int* cell = A + f00 * MSDA_MF_PITCH + sq;
            // slot of row tile s in tf: consumer wave (s & 3) owns it as its entry s >> 2
            if (h00) { atomicAdd(cell, __float2int_rn(s24 * (1.f - ly) * (1.f - lx))); const int s = f00 >> 4; tf[(s & 3) * 8 + (s >> 2)] = 1; }
            if (h01) { atomicAdd(cell + MSDA_MF_PITCH, __float2int_rn(s24 * (1.f - ly) * lx)); const int s = (f00 + 1) >> 4; tf[(s & 3) * 8 + (s >> 2)] = 1; }
            if (h10) { atomicAdd(cell + W * MSDA_MF_PITCH, __float2int_rn(s24 * ly * (1.f - lx))); const int s = (f00 + W) >> 4; tf[(s & 3) * 8 + (s >> 2)] = 1; }
            if (h11) { atomicAdd(cell + (W + 1) * MSDA_MF_PITCH, __float2int_rn(s24 * ly * lx)); const int s = (f00 + W + 1) >> 4; tf[(s & 3) * 8 + (s >> 2)] = 1; }
          } else if (stg_role) {      // dout^T: 8 channels of one query, one 2-byte store per channel row
            unsigned short* G = Gt + (c & 1) * 32 * MSDA_MF_GP;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const uint4 g = g_c[k][h];
              const unsigned gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                G[(gc * 8 + 2 * e) * MSDA_MF_GP + gq + 16 * h] = (unsigned short)(gv[e] & 0xffffu);
                G[(gc * 8 + 2 * e + 1) * MSDA_MF_GP + gq + 16 * h] = (unsigned short)(gv[e] >> 16);
              }
            }
          }
        }
      } else if (c >= 1) {
        const int cw = wave - 4;
        int* A = Ai + ((c - 1) & 1) * asz;
        int* tf = tflag + ((c - 1) & 1) * 32;
        const unsigned short* G = Gt + ((c - 1) & 1) * 32 * MSDA_MF_GP;
        const int4 fa = reinterpret_cast<const int4*>(tf)[cw * 2], fb = reinterpret_cast<const int4*>(tf)[cw * 2 + 1];      // (wave-uniform: broadcast reads)
        if ((fa.x | fa.y | fa.z | fa.w | fb.x | fb.y | fb.z | fb.w) != 0) {
          if (lane == 0) { reinterpret_cast<int4*>(tf)[cw * 2] = make_int4(0, 0, 0, 0); reinterpret_cast<int4*>(tf)[cw * 2 + 1] = make_int4(0, 0, 0, 0); }
          const int fl8[8] = {fa.x, fa.y, fa.z, fa.w, fb.x, fb.y, fb.z, fb.w};
          const msda_bf16x8_t b0 = *reinterpret_cast<const msda_bf16x8_t*>(&G[lr * MSDA_MF_GP + lg * 8]);
          const msda_bf16x8_t b1 = *reinterpret_cast<const msda_bf16x8_t*>(&G[(16 + lr) * MSDA_MF_GP + lg * 8]);
#pragma unroll
          for (int half = 0; half < 2; ++half) {      // four row tiles' fragment reads in flight before the first conversion
            int4 lo[4], hi[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int i = half * 4 + j;
              if (fl8[i]) {
                int4* cell = reinterpret_cast<int4*>(&A[((cw + 4 * i) * 16 + lr) * MSDA_MF_PITCH + lg * 8]);
                lo[j] = cell[0];
                hi[j] = cell[1];
                cell[0] = make_int4(0, 0, 0, 0);          // the tile is empty again for chunk c + 1 (this lane is the only reader of these 8 words)
                cell[1] = make_int4(0, 0, 0, 0);
              }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int i = half * 4 + j;
              if (fl8[i]) {
                const float sc24 = 1.f / 16777216.f;
                msda_bf16x8_t av;
                av[0] = (__bf16)((float)lo[j].x * sc24); av[1] = (__bf16)((float)lo[j].y * sc24); av[2] = (__bf16)((float)lo[j].z * sc24); av[3] = (__bf16)((float)lo[j].w * sc24);
                av[4] = (__bf16)((float)hi[j].x * sc24); av[5] = (__bf16)((float)hi[j].y * sc24); av[6] = (__bf16)((float)hi[j].z * sc24); av[7] = (__bf16)((float)hi[j].w * sc24);
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b0, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b1, acc[i][1], 0, 0, 0);
              }
            }
          }
        }
      }
      __syncthreads();
    }
  }
  // accumulators -> [pixel][33] fp32 in the first weight tile's LDS -> 64-byte rows of dvalue
  float* fl = reinterpret_cast<float*>(Ai);
  if (wave >= 4) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int s = wave - 4 + 4 * i;
      if (s < ntile) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) fl[(s * 16 + lg * 4 + r) * 33 + n * 16 + lr] = acc[i][n][r];
      }
    }
  }
  __syncthreads();
  bf16_t* outp = (bf16_t*)a.dvalue_t + ((long long)b * a.Lv + lstart + pix0) * (a.M * 32) + m * 32;
  for (int i = t; i < npix * 4; i += 512) {
    const int pix = i >> 2, c8 = (i & 3) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fl[pix * 33 + c8 + e];
    Vec8<bf16_t>::store(outp + (long long)pix * (a.M * 32) + c8, v);
  }
}

// bands of the matrix-product scatter: whole rows, <= 512 pixels (a whole level when it fits: every band re-reads the offsets, probabilities and
// dout of ALL queries -- 152 B per query and head)
static bool msda_mf_plan(const MsdaArgs& a, int L, MsdaMfPlan& pl, int& nbands) {
  nbands = 0;
  pl.npix_cap = 16;
  for (int l = 0; l < 4; ++l) { pl.rows[l] = 1; pl.nb[l] = 0; }
  const int cap = g_tune.msda_mf_bands > 0 ? (g_tune.msda_mf_bands < MSDA_MF_NPIX ? g_tune.msda_mf_bands : MSDA_MF_NPIX) : MSDA_MF_NPIX;      // knob: most pixels per band
  for (int l = 0; l < L; ++l) {
    if (a.w[l] > cap) return false;
    int rows = cap / a.w[l];
    if (rows > a.h[l]) rows = a.h[l];
    pl.rows[l] = rows;
    pl.nb[l] = (a.h[l] + rows - 1) / rows;
    nbands += pl.nb[l];
    const int np = (rows * a.w[l] + 15) & ~15;
    if (np > pl.npix_cap) pl.npix_cap = np;
  }
  return nbands <= 65535;
}

static int msda_launch_mf(const MsdaArgs& a, int L, int P, const MsdaMfPlan& pl, int nbands, hipStream_t st) {
  size_t lds = (size_t)2 * pl.npix_cap * MSDA_MF_PITCH * 4 + 2 * 32 * MSDA_MF_GP * 2 + 64 * 4;
  const size_t epi = (size_t)pl.npix_cap * 33 * 4;          // the epilogue's fp32 image starts at the first weight tile
  if (lds < epi) lds = epi;
#define MSDA_MF_CASE(LL, PP)                                                                                  \
  if (L == LL && P == PP) {                                                                                   \
    static bool attr = false;                                                                                 \
    if (!attr) { (void)hipFuncSetAttribute((const void*)msda_bwd_value_mfma_kernel<LL, PP>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024); attr = true; } \
    hipLaunchKernelGGL((msda_bwd_value_mfma_kernel<LL, PP>), dim3(a.B * a.M, nbands), dim3(512), lds, st, a, pl); \
    return check_launch("emrt_msda_bwd(matrix-product scatter)");                                             \
  }
  MSDA_MF_CASE(3, 6)
  MSDA_MF_CASE(4, 4)
  MSDA_MF_CASE(3, 4)
  MSDA_MF_CASE(1, 4)
#undef MSDA_MF_CASE
  return fail("emrt_msda_bwd", "unsupported (levels, points) for the matrix-product scatter");
}

// pixels of a scatter range: (npix + 2 guard) * 33 * 4 B of slab + 20 480 B of sample records must fit the 160 KiB LDS
static int msda_max_npix(int guard) { return ((160 * 1024 - 20480) / (MSDA_SLAB_PITCH * 4) - 2 * guard) & ~3; }

// Slab ranges: every level is cut into equal runs of whole rows, at least enough that a run fits in LDS, and more when
// the grid would otherwise leave CUs idle (the largest level is the one worth cutting further).  Returns the number of ranges
// (<= 8 unless the developer knob forces more) or -1.
// Measured (round 3, tools/bench_msda.py): a block's time is ~25 instructions per sample of every wave that has ANY sample in its range
// (not memory latency, not LDS bandwidth), and the samples of a small level spread over all of its rows: cutting the 16 x 16 / 8 x 8 levels
// into row ranges multiplies the work instead of dividing it (batch 8, 256x256: 4 ranges 87 us, 9 ranges 124 us for the whole backward call).
#define MSDA_MAX_RANGES 16
static int msda_ranges(MsdaArgs& a, int L, int bm, bool allow_split) {
  const int MSDA_MAX_NPIX = msda_max_npix(a.g_guard);
  if (MSDA_MAX_NPIX < 64) return -1;
  int cuts[4];
  int total = 0;
  for (int l = 0; l < L; ++l) {
    const int px = a.h[l] * a.w[l];
    cuts[l] = (px + MSDA_MAX_NPIX - 1) / MSDA_MAX_NPIX;
    if (cuts[l] > a.h[l]) return -1;                    // a single row does not fit
    total += cuts[l];
  }
  while (total < 8 && (long long)total * bm < 256) {     // below one block per CU: halve the tallest remaining piece
    int best = -1, best_rows = 1;
    for (int l = 0; l < L; ++l) {
      const int rows = (a.h[l] + cuts[l] - 1) / cuts[l];
      if (rows > best_rows) { best_rows = rows; best = l; }
    }
    if (best < 0) break;
    ++cuts[best];
    ++total;
  }
  if (g_tune.msda_scatter_cuts > 0) {                    // developer knob: the same number of cuts for every level
    total = 0;
    for (int l = 0; l < L; ++l) {
      const int need = (a.h[l] * a.w[l] + MSDA_MAX_NPIX - 1) / MSDA_MAX_NPIX;
      cuts[l] = g_tune.msda_scatter_cuts < need ? need : (g_tune.msda_scatter_cuts > a.h[l] ? a.h[l] : g_tune.msda_scatter_cuts);
      total += cuts[l];
    }
  }
  if (total > MSDA_MAX_RANGES) return -1;
  // Query split.  A block's time is the number of LDS atomics it issues (7 cycles per wave instruction, measured), i.e. the number of
  // samples that land in its range; every level receives the same number of samples, so a range of a level with c cuts carries 1 / c of
  // a level's samples: at Lv = 5376 the block that owns the whole 16 x 16 level carries 5x what a level-0 block does and IS the kernel
  // (216 of 216 us).  Ranges are therefore also split by QUERIES (nqs blocks, each walking every nqs-th group of queries into its own
  // slab; msda_bwd_value_finalize_kernel adds the slabs) until every block carries about what the finest-cut level's blocks do -- if
  // the estimate (rounds of 256 blocks x heaviest block) improves by 20 % or more; needs the per-(batch, head) max |dout| partials.
  int nqs[4];
  for (int l = 0; l < L; ++l) nqs[l] = 1;
  a.f_n = 0;
  a.part_stride = 0;
  if (allow_split && g_tune.msda_scatter_qsplit >= 0) {
    int cmax = 1;
    for (int l = 0; l < L; ++l) cmax = cuts[l] > cmax ? cuts[l] : cmax;
    int entries = 0, cand[4];
    double heavy = 0.0;
    for (int l = 0; l < L; ++l) {
      cand[l] = (cmax + cuts[l] - 1) / cuts[l];
      if (cand[l] > 8) cand[l] = 8;
      if (g_tune.msda_scatter_qsplit > 0) cand[l] = cuts[l] < cmax ? g_tune.msda_scatter_qsplit : 1;
      entries += cuts[l] * cand[l];
      const double w = 1.0 / (cuts[l] * cand[l]);
      heavy = w > heavy ? w : heavy;
    }
    const double rounds_new = (double)(((long long)entries * bm + 255) / 256), rounds_old = (double)(((long long)total * bm + 255) / 256);
    int cmin = cuts[0];
    for (int l = 1; l < L; ++l) cmin = cuts[l] < cmin ? cuts[l] : cmin;
    if (entries <= MSDA_MAX_RANGES && (rounds_new * heavy < 0.8 * rounds_old / cmin || g_tune.msda_scatter_qsplit > 0))
      for (int l = 0; l < L; ++l) nqs[l] = cand[l];
  }
  int g = 0;
  for (int l = 0; l < L; ++l) {
    const int rows_per = (a.h[l] + cuts[l] - 1) / cuts[l];
    for (int r0 = 0; r0 < a.h[l]; r0 += rows_per) {
      const int rows = a.h[l] - r0 < rows_per ? a.h[l] - r0 : rows_per;
      if (nqs[l] > 1) {
        if (a.f_n >= 8) return -1;
        a.f_pix0[a.f_n] = a.start[l] + r0 * a.w[l];
        a.f_npix[a.f_n] = rows * a.w[l];
        a.f_nqs[a.f_n] = nqs[l];
        a.f_part[a.f_n] = (int)a.part_stride;
        ++a.f_n;
      }
      for (int k = 0; k < nqs[l]; ++k) {
        a.g_level[g] = l;
        a.g_pix0[g] = a.start[l] + r0 * a.w[l];
        a.g_npix[g] = rows * a.w[l];
        a.g_qs[g] = k;
        a.g_nqs[g] = nqs[l];
        a.g_part[g] = (int)a.part_stride;
        if (nqs[l] > 1) a.part_stride += (long long)rows * a.w[l] * 32;
        ++g;
      }
    }
  }
  return g;
}

extern "C" int emrt_msda_bwd_uses_lds(const int* shapes_hw, int L) {
  (void)shapes_hw;
  return (L >= 1 && L <= 4) ? 1 : 0;     // pixel-range slabs fit for every map size
}

static int msda_fill(MsdaArgs& a, const int* shapes_hw, int L, int Lv);
// Whether the query-split scatter may be planned for a call: it needs the max |dout| partials some gradient kernel leaves (every
// path with Lq >= 1024 or the LDS / band gradient kernels does) -- decided from the same quantities in both functions below.
static bool msda_split_allowed(int B, int Lq, int M, int dtype_is_2byte) {
  (void)B;
  return dtype_is_2byte && Lq >= 1024 && M <= 8 && 256 % (M * 4) == 0;
}

static size_t msda_part_offset_floats(int B, int Lq, int M, int L, int P) {      // where the partial slabs start inside the workspace
  // max |dout| partials per (batch, head): Lq / 64 from the separate scan, <= Lq / 128 chunks from the LDS gradient kernel, up to 64 row
  // bands from the band kernel whatever Lq is (narrow maps: ADVICE r3)
  const size_t gparts = (size_t)((Lq + 63) / 64 > 64 ? (Lq + 63) / 64 : 64);
  return ((size_t)B * Lq * M * L * P + 512 + 2 * (size_t)B * M + (size_t)B * M * gparts + 3) & ~(size_t)3;      // 16-byte aligned
}

extern "C" size_t emrt_msda_bwd_workspace_bytes(int B, int Lq, int M, int L, int P, const int* shapes_hw, int dtype) {
  // softmax probabilities + per-block max |dout| partials (the LDS gradient kernel leaves at most 256 / (B M) + 1 per (batch, head), the
  // separate scan Lq / 64) + the integer partial slabs of a query-split scatter (the plan emrt_msda_bwd will make for these shapes)
  if (!shapes_hw || L < 1 || L > 4 || B < 1 || Lq < 1 || M < 1 || P < 1) {      // (a size that left the partial slabs out would be written past)
    fail("emrt_msda_bwd_workspace_bytes", "needs the level shapes (L = 1..4) and positive sizes; returns 0");
    return 0;
  }
  size_t n = msda_part_offset_floats(B, Lq, M, L, P);
  {
    MsdaArgs a;
    memset(&a, 0, sizeof(a));
    int Lv = 0;
    for (int l = 0; l < L; ++l) Lv += shapes_hw[2 * l] * shapes_hw[2 * l + 1];
    if (msda_fill(a, shapes_hw, L, Lv) == 0) {
      int wmax = 1;
      for (int l = 0; l < L; ++l) wmax = a.w[l] > wmax ? a.w[l] : wmax;
      a.g_guard = wmax + 2;
      if (msda_ranges(a, L, B * M, msda_split_allowed(B, Lq, M, dtype != EMRT_F32)) > 0) n += (size_t)B * M * (size_t)a.part_stride;
    }
  }
  return n * sizeof(float);
}

template <class T>
static int msda_launch(const MsdaArgs& a, int L, int P, int mode /*0 fwd, 1 bwd atomic, 2 bwd grads only*/, hipStream_t st) {
  const long long pairs = (long long)a.B * a.Lq * a.M;
  const unsigned grid = (unsigned)((pairs + 63) / 64);
#define MSDA_CASE(LL, PP)                                                                                     \
  if (L == LL && P == PP) {                                                                                   \
    if (mode == 0) hipLaunchKernelGGL((msda_fwd_kernel<T, LL, PP>), dim3(grid), dim3(256), 0, st, a);          \
    else if (mode == 1) hipLaunchKernelGGL((msda_bwd_kernel<T, LL, PP, true>), dim3(grid), dim3(256), 0, st, a); \
    else hipLaunchKernelGGL((msda_bwd_kernel<T, LL, PP, false>), dim3(grid), dim3(256), 0, st, a);             \
    return check_launch(mode ? "emrt_msda_bwd" : "emrt_msda_fwd");                                            \
  }
  MSDA_CASE(3, 6)
  MSDA_CASE(4, 4)
  MSDA_CASE(3, 4)
  MSDA_CASE(1, 4)
#undef MSDA_CASE
  return fail("emrt_msda", "unsupported (levels, points): built for (3,6), (4,4), (3,4), (1,4)");
}

// forward only (the one MSDA kernel fp16 inference needs)
template <class T>
static int msda_launch_fwd(const MsdaArgs& a, int L, int P, hipStream_t st) {
  const long long pairs = (long long)a.B * a.Lq * a.M;
  const unsigned grid = (unsigned)((pairs + 63) / 64);
#define MSDA_FWD_CASE(LL, PP)                                                                                 \
  if (L == LL && P == PP) {                                                                                   \
    hipLaunchKernelGGL((msda_fwd_kernel<T, LL, PP>), dim3(grid), dim3(256), 0, st, a);                        \
    return check_launch("emrt_msda_fwd");                                                                     \
  }
  MSDA_FWD_CASE(3, 6)
  MSDA_FWD_CASE(4, 4)
  MSDA_FWD_CASE(3, 4)
  MSDA_FWD_CASE(1, 4)
#undef MSDA_FWD_CASE
  return fail("emrt_msda_fwd", "unsupported (levels, points): built for (3,6), (4,4), (3,4), (1,4)");
}

template <class T>
static int msda_launch_fwd_lds(const MsdaArgs& a, int L, int P, int chunks, int qpb, int guard, size_t slab, hipStream_t st) {
  int threads = g_tune.msda_fwd_threads;
  if (threads < 64 || threads > 1024 || (threads & 63)) threads = 1024;
#define MSDA_FWD_LDS_CASE(LL, PP)                                                                                         \
  if (L == LL && P == PP) {                                                                                             \
    static bool attr = false;                                                                                           \
    if (!attr) { (void)hipFuncSetAttribute((const void*)msda_fwd_lds_kernel<T, LL, PP>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024); attr = true; } \
    hipLaunchKernelGGL((msda_fwd_lds_kernel<T, LL, PP>), dim3(a.B * a.M * chunks), dim3(threads), slab, st, a, qpb, chunks, guard, g_tune.msda_fwd_probe); \
    return check_launch("emrt_msda_fwd(lds)");                                                                          \
  }
  MSDA_FWD_LDS_CASE(3, 6)
  MSDA_FWD_LDS_CASE(4, 4)
  MSDA_FWD_LDS_CASE(3, 4)
  MSDA_FWD_LDS_CASE(1, 4)
#undef MSDA_FWD_LDS_CASE
  return fail("emrt_msda_fwd", "unsupported (levels, points): built for (3,6), (4,4), (3,4), (1,4)");
}

template <class T>
static int msda_launch_fwd_band(const MsdaArgs& a, int L, int P, int NB, int halo, int guard, size_t slab, hipStream_t st) {
#define MSDA_FWD_BAND_CASE(LL, PP)                                                                                        \
  if (L == LL && P == PP) {                                                                                             \
    static bool attr = false;                                                                                           \
    if (!attr) { (void)hipFuncSetAttribute((const void*)msda_fwd_band_kernel<T, LL, PP>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024); attr = true; } \
    hipLaunchKernelGGL((msda_fwd_band_kernel<T, LL, PP>), dim3(a.B * a.M * NB), dim3(1024), slab, st, a, NB, halo, guard); \
    return check_launch("emrt_msda_fwd(band)");                                                                         \
  }
  MSDA_FWD_BAND_CASE(3, 6)
  MSDA_FWD_BAND_CASE(4, 4)
  MSDA_FWD_BAND_CASE(3, 4)
#undef MSDA_FWD_BAND_CASE
  return fail("emrt_msda_fwd", "unsupported (levels, points) for the band kernel");
}

// Band plan for a pyramid whose whole slab does not fit: the number of bands NB (every level's height divisible by it) and the halo
// (rows staged beyond a band on both sides, the same for every level) with the largest halo <= 7 whose slab fits 159 KB; NB as small as
// gives about one block per CU.  Returns false when no plan fits (the global-gather kernels take the call).
static bool msda_band_plan(const MsdaArgs& a, int L, int bm, int guard, int& NB_out, int& halo_out, size_t& slab_out) {
  for (int NB = 2; NB <= 64; NB *= 2) {
    bool div = true;
    for (int l = 0; l < L; ++l) div = div && a.h[l] % NB == 0;
    if (!div) break;
    if ((long long)NB * bm < 192 && NB * 2 <= 64) {        // too few blocks: try more bands first (if the heights allow)
      bool div2 = true;
      for (int l = 0; l < L; ++l) div2 = div2 && a.h[l] % (NB * 2) == 0;
      if (div2) continue;
    }
    for (int halo = g_tune.msda_band_halo > 0 ? g_tune.msda_band_halo : 7; halo >= 3; --halo) {
      int npx = guard;
      for (int l = 0; l < L; ++l) {
        const int rows = a.h[l] / NB + 2 * halo < a.h[l] ? a.h[l] / NB + 2 * halo : a.h[l];
        npx = ((npx + 15) & ~15) + rows * a.w[l];
      }
      const size_t slab = (size_t)(((npx + guard + 15) >> 4) << 4) * MSDA_FWD_PITCH;
      if (slab <= 159 * 1024) { NB_out = NB; halo_out = halo; slab_out = slab; return true; }
    }
  }
  return false;
}

template <class T>
static int msda_launch_bwd_grad_lds(const MsdaArgs& a, int L, int P, int chunks, int qpb, int guard, size_t slab, hipStream_t st) {
#define MSDA_BWD_GLDS_CASE(LL, PP)                                                                                        \
  if (L == LL && P == PP) {                                                                                             \
    static bool attr = false;                                                                                           \
    if (!attr) { (void)hipFuncSetAttribute((const void*)msda_bwd_lds_kernel<T, LL, PP>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024); attr = true; } \
    hipLaunchKernelGGL((msda_bwd_lds_kernel<T, LL, PP>), dim3(a.B * a.M * chunks), dim3(1024), slab, st, a, qpb, chunks, guard); \
    return check_launch("emrt_msda_bwd(lds gradients)");                                                                \
  }
  MSDA_BWD_GLDS_CASE(3, 6)
  MSDA_BWD_GLDS_CASE(4, 4)
  MSDA_BWD_GLDS_CASE(3, 4)
  MSDA_BWD_GLDS_CASE(1, 4)
#undef MSDA_BWD_GLDS_CASE
  return fail("emrt_msda_bwd", "unsupported (levels, points)");
}

template <class T>
static int msda_launch_bwd_grad_band(const MsdaArgs& a, int L, int P, int NB, int halo, int guard, size_t slab, hipStream_t st) {
#define MSDA_BWD_BAND_CASE(LL, PP)                                                                                        \
  if (L == LL && P == PP) {                                                                                             \
    static bool attr = false;                                                                                           \
    if (!attr) { (void)hipFuncSetAttribute((const void*)msda_bwd_lds_kernel<T, LL, PP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024); attr = true; } \
    hipLaunchKernelGGL((msda_bwd_lds_kernel<T, LL, PP, true>), dim3(a.B * a.M * NB), dim3(1024), slab, st, a, NB, halo, guard); \
    return check_launch("emrt_msda_bwd(band gradients)");                                                               \
  }
  MSDA_BWD_BAND_CASE(3, 6)
  MSDA_BWD_BAND_CASE(4, 4)
  MSDA_BWD_BAND_CASE(3, 4)
#undef MSDA_BWD_BAND_CASE
  return fail("emrt_msda_bwd", "unsupported (levels, points) for the band kernel");
}

template <class T>
static int msda_launch_lds(const MsdaArgs& a, int L, int P, int ngroups, size_t lds, hipStream_t st) {
#define MSDA_LDS_CASE(LL, PP)                                                                                 \
  if (L == LL && P == PP) {                                                                                   \
    static bool attr = false;                                                                                 \
    if (!attr) { (void)hipFuncSetAttribute((const void*)msda_bwd_value_lds_kernel<T, LL, PP>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024); attr = true; } \
    hipLaunchKernelGGL((msda_bwd_value_lds_kernel<T, LL, PP>), dim3(a.B * a.M, ngroups), dim3(1024), lds, st, a, ((g_tune.msda_fwd_probe >> 4) & 127) | (g_tune.msda_scatter_merge ? 128 : 0)); \
    return check_launch("emrt_msda_bwd(lds scatter)");                                                        \
  }
  MSDA_LDS_CASE(3, 6)
  MSDA_LDS_CASE(4, 4)
  MSDA_LDS_CASE(3, 4)
  MSDA_LDS_CASE(1, 4)
#undef MSDA_LDS_CASE
  return fail("emrt_msda_bwd", "unsupported (levels, points) for the LDS scatter");
}

static int msda_fill(MsdaArgs& a, const int* shapes_hw, int L, int Lv) {
  int start = 0;
  for (int l = 0; l < 4; ++l) { a.h[l] = 1; a.w[l] = 1; a.start[l] = 0; a.inv_h[l] = 1.f; a.inv_w[l] = 1.f; }
  for (int l = 0; l < L; ++l) {
    a.h[l] = shapes_hw[2 * l];
    a.w[l] = shapes_hw[2 * l + 1];
    a.inv_h[l] = 1.f / (float)a.h[l];
    a.inv_w[l] = 1.f / (float)a.w[l];
    a.start[l] = start;
    start += a.h[l] * a.w[l];
  }
  return start == Lv ? 0 : -1;
}

extern "C" int emrt_msda_fwd(const void* value, int ldv, long long v_bs, const float* offw, int ldo, const float* ref,
                             long long ref_bs, int ref_L, void* out, int B, int Lq, int Lv, int M, int D, int L, int P,
                             const int* shapes_hw /*host, [L][2]*/, int dtype, void* stream) {
  EMRT_REQUIRE_FWD_DTYPE(dtype);
  EMRT_REQUIRE(ref_L == 1 || ref_L == L, "ref_L must be 1 or L");
  EMRT_REQUIRE(value && offw && ref && out && shapes_hw, "null pointer");
  EMRT_REQUIRE(D == 32, "head dim must be 32");
  EMRT_REQUIRE(M >= 1 && L >= 1 && L <= 4 && P >= 1, "bad M/L/P");
  EMRT_REQUIRE(ldv % 8 == 0 && v_bs % 8 == 0 && ldo % 2 == 0, "value strides must be multiples of 8 elements");
  EMRT_REQUIRE(ldo >= M * L * P * 3, "offw row too short");
  EMRT_REQUIRE(dtype == EMRT_F32 || ((long long)(B - 1) * v_bs + (long long)Lv * ldv) * 2 < (1ll << 31), "value tensor spans 2 GiB or more (32-bit buffer offsets)");
  MsdaArgs a;
  memset(&a, 0, sizeof(a));
  a.value = value; a.ldv = ldv; a.v_bs = v_bs; a.offw = offw; a.ldo = ldo; a.ref = ref; a.ref_bs = ref_bs; a.ref_L = ref_L; a.out = out;
  a.B = B; a.Lq = Lq; a.M = M;
  EMRT_REQUIRE(msda_fill(a, shapes_hw, L, Lv) == 0, "sum(h*w) != Lv");
  hipStream_t st = (hipStream_t)stream;
  a.Lv = Lv;
  int wmax = 1;
  for (int l = 0; l < L; ++l) wmax = a.w[l] > wmax ? a.w[l] : wmax;
  const int guard = wmax + 2;                                  // zero rows on both sides of the staged slab (msda_fwd_lds_kernel)
  const size_t slab = (size_t)(Lv + 2 * guard) * MSDA_FWD_PITCH;
  if (dtype != EMRT_F32 && slab <= 159 * 1024 && (long long)B * M * Lq >= g_tune.msda_lds_min_pairs && !g_tune.msda_fwd_global) {
    // one block per CU (the slab takes most of its LDS): as close to 256 blocks as whole chunks of >= 128 queries allow
    int chunks = (256 + B * M / 2) / (B * M);
    if (g_tune.msda_fwd_chunks > 0) chunks = g_tune.msda_fwd_chunks;
    if (chunks > (Lq + 127) / 128) chunks = (Lq + 127) / 128;
    if (chunks < 1) chunks = 1;
    const int qpb = (Lq + chunks - 1) / chunks;
    chunks = (Lq + qpb - 1) / qpb;
    return dtype == EMRT_BF16 ? msda_launch_fwd_lds<bf16_t>(a, L, P, chunks, qpb, guard, slab, st) : msda_launch_fwd_lds<f16_t>(a, L, P, chunks, qpb, guard, slab, st);
  }
  if (dtype != EMRT_F32 && slab > 159 * 1024 && Lq == Lv && L >= 2 && (long long)B * M * Lq >= g_tune.msda_lds_min_pairs &&
      !g_tune.msda_fwd_global) {
    // self-attention over a pyramid too large for one slab: row bands (msda_fwd_band_kernel)
    int NB = 0, halo = 0;
    size_t bslab = 0;
    if (msda_band_plan(a, L, B * M, guard, NB, halo, bslab))
      return dtype == EMRT_BF16 ? msda_launch_fwd_band<bf16_t>(a, L, P, NB, halo, guard, bslab, st) : msda_launch_fwd_band<f16_t>(a, L, P, NB, halo, guard, bslab, st);
  }
  if (dtype == EMRT_F16) return msda_launch_fwd<f16_t>(a, L, P, st);
  return dtype == EMRT_F32 ? msda_launch_fwd<float>(a, L, P, st) : msda_launch_fwd<bf16_t>(a, L, P, st);
}

// dvalue: when emrt_msda_bwd_uses_lds(shapes) it is [B][Lv][M*D] in the COMPUTE dtype and fully overwritten (workspace
// of emrt_msda_bwd_workspace_bytes required); otherwise fp32, pre-zeroed by the caller, accumulated with global atomics.
extern "C" int emrt_msda_bwd(const void* value, int ldv, long long v_bs, const float* offw, int ldo, const float* ref,
                             long long ref_bs, int ref_L, const void* dout, void* dvalue, void* doffw, int doffw_compute_dtype, float* dref,
                             int B, int Lq, int Lv, int M, int D, int L, int P, const int* shapes_hw, void* workspace, size_t workspace_bytes, int dtype, void* stream) {
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  EMRT_REQUIRE(ref_L == 1 || ref_L == L, "ref_L must be 1 or L");
  EMRT_REQUIRE(value && offw && ref && dout && dvalue && doffw && shapes_hw, "null pointer");
  EMRT_REQUIRE(D == 32, "head dim must be 32");
  EMRT_REQUIRE(M >= 1 && L >= 1 && L <= 4 && P >= 1, "bad M/L/P");
  EMRT_REQUIRE(!dref || M == 8, "reference-point gradient needs M == 8");
  EMRT_REQUIRE(ldv % 8 == 0 && v_bs % 8 == 0 && ldo % 2 == 0, "value strides must be multiples of 8 elements");
  MsdaArgs a;
  memset(&a, 0, sizeof(a));
  a.value = value; a.ldv = ldv; a.v_bs = v_bs; a.offw = offw; a.ldo = ldo; a.ref = ref; a.ref_bs = ref_bs; a.ref_L = ref_L;
  a.dout = dout; a.dv_bs = (long long)Lv * M * 32; a.doffw = doffw; a.doffw_t = (doffw_compute_dtype && dtype != EMRT_F32) ? 1 : 0; a.dref = dref;
  a.B = B; a.Lq = Lq; a.M = M; a.Lv = Lv;
  EMRT_REQUIRE(msda_fill(a, shapes_hw, L, Lv) == 0, "sum(h*w) != Lv");
  hipStream_t st = (hipStream_t)stream;
  const bool lds_ok = (L == 3 && P == 6) || (L == 4 && P == 4) || (L == 3 && P == 4) || (L == 1 && P == 4);
  if (lds_ok) {
    EMRT_REQUIRE(workspace, "LDS scatter path needs the probability workspace");
    EMRT_REQUIRE(workspace_bytes >= emrt_msda_bwd_workspace_bytes(B, Lq, M, L, P, shapes_hw, dtype), "workspace smaller than emrt_msda_bwd_workspace_bytes() for these arguments");
    a.probs = (float*)workspace;
    a.dvalue_t = dvalue;
    int wmax = 1;
    for (int l = 0; l < L; ++l) wmax = a.w[l] > wmax ? a.w[l] : wmax;
    const int guard = wmax + 2;
    a.g_guard = guard;
    const int ng = msda_ranges(a, L, B * M, msda_split_allowed(B, Lq, M, dtype != EMRT_F32));
    EMRT_REQUIRE(ng > 0, "value map rows too long for the LDS slab");
    a.part = (int*)((float*)workspace + msda_part_offset_floats(B, Lq, M, L, P));
    int npix_max = 0;
    for (int g = 0; g < ng; ++g) npix_max = a.g_npix[g] > npix_max ? a.g_npix[g] : npix_max;
    // offset / logit gradients: from the LDS-staged slab when it fits (same conditions and launch shape as the forward)
    const size_t slab = (size_t)(Lv + 2 * guard) * MSDA_FWD_PITCH;
    size_t bslab = 0;
    int rc;
    if (dtype == EMRT_BF16 && (!dref || g_tune.msda_bwd_dref_lds) && slab <= 159 * 1024 && (long long)B * M * Lq >= g_tune.msda_lds_min_pairs && !g_tune.msda_bwd_global) {
      int chunks = (256 + B * M / 2) / (B * M);
      if (chunks > (Lq + 127) / 128) chunks = (Lq + 127) / 128;
      if (chunks < 1) chunks = 1;
      const int qpb = (Lq + chunks - 1) / chunks;
      chunks = (Lq + qpb - 1) / qpb;
      a.gmax = (float*)workspace + (size_t)B * Lq * M * L * P;
      a.gmax_n = chunks;
      rc = msda_launch_bwd_grad_lds<bf16_t>(a, L, P, chunks, qpb, guard, slab, st);
    } else if (int NB = 0, halo = 0; dtype == EMRT_BF16 && !dref && slab > 159 * 1024 && Lq == Lv && L >= 2 && !g_tune.msda_bwd_global &&
               (long long)B * M * Lq >= g_tune.msda_lds_min_pairs && msda_band_plan(a, L, B * M, guard, NB, halo, bslab)) {
      // self-attention over a pyramid too large for one slab: row bands (msda_bwd_lds_kernel<BAND>), leaves max |dout| per block too
      a.gmax = (float*)workspace + (size_t)B * Lq * M * L * P;
      a.gmax_n = NB;
      rc = msda_launch_bwd_grad_band<bf16_t>(a, L, P, NB, halo, guard, bslab, st);
    } else {
      rc = dtype == EMRT_F32 ? msda_launch<float>(a, L, P, 2, st) : msda_launch<bf16_t>(a, L, P, 2, st);
      if (!rc && Lq >= 1024 && M <= 8 && 256 % (M * 4) == 0) {      // long scans only: the extra launch costs ~3 us
        a.gmax = (float*)workspace + (size_t)B * Lq * M * L * P;
        a.gmax_n = (Lq + 63) / 64;
        if (dtype == EMRT_F32) hipLaunchKernelGGL((msda_absmax_kernel<float>), dim3(B * a.gmax_n), dim3(256), 0, st, (const float*)dout, Lq, M, a.gmax_n, a.gmax);
        else hipLaunchKernelGGL((msda_absmax_kernel<bf16_t>), dim3(B * a.gmax_n), dim3(256), 0, st, (const bf16_t*)dout, Lq, M, a.gmax_n, a.gmax);
      }
    }
    if (rc) return rc;
    if (MsdaMfPlan pl; dtype == EMRT_BF16 && g_tune.msda_scatter_mfma) {
      // every band of a level re-reads and re-derives ALL queries' samples of that level: worth it while at most one level is cut in two
      // (cfg2: 32 x 32 | 16 x 16 | 8 x 8 = 4 bands, 256 blocks at batch 8: 45 vs 59 us; cfg3's 64 x 64 level would be 8 bands: 323 vs 199 us per call).
      // knob msda_scatter_mfma: 0 = never, 1 = this rule, 2 = whenever a plan exists (tests)
      int nbands = 0;
      if (msda_mf_plan(a, L, pl, nbands) && (nbands <= L + 1 || g_tune.msda_scatter_mfma >= 2)) return msda_launch_mf(a, L, P, pl, nbands, st);
    }
    a.g_npix_max = (npix_max + 2 * guard + 3) & ~3;          // slab + both guard bands; keeps the records 16-byte aligned
    const size_t lds = (size_t)a.g_npix_max * MSDA_SLAB_PITCH * sizeof(int) + 32 * 32 * (sizeof(float4) + sizeof(int));
    EMRT_REQUIRE(a.f_n == 0 || a.gmax_n > 0, "internal: query-split scatter planned without the max |dout| partials");
    rc = dtype == EMRT_F32 ? msda_launch_lds<float>(a, L, P, ng, lds, st) : msda_launch_lds<bf16_t>(a, L, P, ng, lds, st);
    if (rc || a.f_n == 0) return rc;
    hipLaunchKernelGGL((msda_bwd_value_finalize_kernel<bf16_t>), dim3(B * M, a.f_n, 8), dim3(256), 0, st, a);
    return check_launch("emrt_msda_bwd(scatter finalize)");
  }
  a.dvalue = (float*)dvalue;
  return dtype == EMRT_F32 ? msda_launch<float>(a, L, P, 1, st) : msda_launch<bf16_t>(a, L, P, 1, st);
}
