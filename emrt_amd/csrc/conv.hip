// Implicit-GEMM convolution / linear kernels for gfx950 (MFMA 32x32, NHWC activations).
//
// Replaces the cuDNN/cuBLAS work behind paddle.nn.Conv2D / nn.Linear on the EMRT path
// (SURVEY.md 2.2 K1, K3, K4, K9): reference call sites e.g. paddle_vision_resnet.py:108-123,
// paddle_EMRT.py:16-23,134-138,201-209, transformer_encoder_decoder.py:36-42,118-121,125-144.
//
// Data layout: activations NHWC with an explicit pixel stride (`ld`, elements) and batch stride
// (`bs`, elements) so that a level slab of the [B, Lv, C] token tensor or a channel slice of a
// concat buffer is a valid operand without a copy.  Weights are "packed" K-major rows:
//   fwd  : Wf[OC][KH][KW][C]          (== torch [OC,C,KH,KW] in channels_last memory)
//   dgrad: Wb[C ][KH][KW][OC]         (transposed copy made once per optimizer step, optim.hip)
// so both MFMA operands are read from LDS as 16-byte k-contiguous chunks.
//
// out[m][n] = sum_{tap,c} in[pix(m,tap)][c] * Wp[n][tap*C + c]      (m = output pixel, n = out channel)
//   MODE 0 (fwd)  : pix = (oh*stride - pad + kh, ow*stride - pad + kw)
//   MODE 1 (dgrad): pix = ((oh + pad - kh)/stride, (ow + pad - kw)/stride) when divisible
// wgrad: dW[oc][tap*C + c] += sum_m dy[m][oc] * x[pix(m,tap)][c]   (fp32 atomics, split over m)
#include "common.hpp"

using namespace emrt;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

struct ConvArgs {
  const void* in;
  const void* w;
  void* out;
  const float* bias;
  const void* res;
  int N, H, W, C, ldin;
  long long in_bs;
  int OH, OW, OC, ldout;
  long long out_bs;
  int ldres;
  long long res_bs;
  int KH, KW, stride, pad;
  int relu, out_f32;
  double* stats;   // optional [8][2*OC]: per-channel sum / sum of squares of the STORED outputs (BatchNorm statistics);
                   // fp64 atomics spread over 8 replicas (by M-tile index) so that blocks do not pile onto one address
};

template <class T>
__device__ __forceinline__ void mma_chunk(f32x16_t& acc, const uint4& a, const uint4& b);
template <>
__device__ __forceinline__ void mma_chunk<bf16_t>(f32x16_t& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_chunk<float>(f32x16_t& acc, const uint4& a, const uint4& b) {
  // lane (r, h) holds 4 consecutive k of its row; MFMA e pairs k-slot h with element e of both operands.
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// fwd / dgrad implicit GEMM.  256 threads = 4 waves laid out WR x WC, each wave TM x TN tiles of 32x32.
// LDS: one [BM + BN] x 128-byte k-tile (rows padded to 144 B: conflict-free ds_read_b128), register
// prefetch of the next k-tile while the MFMAs of the current one run.
// ------------------------------------------------------------------------------------------------
// VEC = true : C % (16 B of elements) == 0 and 16-byte aligned rows -> one 16-byte load per chunk; the last k-tile may
//              be partial (K % BK != 0, e.g. the fused 432-wide offsets|logits projection) and is zero-filled.
// VEC = false: any C / alignment (7x7x3 stem, 3x3x3 branch conv, 6-class logits): chunks are assembled from element loads.
template <class T, int TM, int TN, int WR, int WC, int MODE, bool VEC>
__global__ __launch_bounds__(256) void igemm_kernel(ConvArgs p) {
  constexpr int BM = WR * TM * 32, BN = WC * TN * 32;
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int BK = 8 * EPC;
  constexpr int PITCH = 144;
  constexpr int AR = BM / 32, BR = BN / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;
  unsigned char* sB = smem + BM * PITCH;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WC, wc = wave % WC;
  const int tiles_n = (p.OC + BN - 1) / BN;
  const int bm = blockIdx.x / tiles_n, bn = blockIdx.x % tiles_n;
  const int OHW = p.OH * p.OW;
  const long long M = (long long)p.N * OHW;
  const int chunk = tid & 7, row0 = tid >> 3;

  const T* __restrict__ inp = (const T*)p.in;
  const T* __restrict__ wp = (const T*)p.w;
  const int K = p.KH * p.KW * p.C;
  const int nkt = (K + BK - 1) / BK;

  int a_h[AR], a_w[AR];
  long long a_base[AR];
  bool a_ok[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    long long m = (long long)bm * BM + row0 + 32 * i;
    a_ok[i] = m < M;
    long long mm = a_ok[i] ? m : 0;
    int nb = (int)(mm / OHW);
    int r = (int)(mm - (long long)nb * OHW);
    int oh = r / p.OW, ow = r - oh * p.OW;
    a_base[i] = (long long)nb * p.in_bs;
    if (MODE == 0) { a_h[i] = oh * p.stride - p.pad; a_w[i] = ow * p.stride - p.pad; }
    else { a_h[i] = oh + p.pad; a_w[i] = ow + p.pad; }
  }
  long long b_off[BR];
  bool b_ok[BR];
#pragma unroll
  for (int j = 0; j < BR; ++j) {
    int n = bn * BN + row0 + 32 * j;
    b_ok[j] = n < p.OC;
    b_off[j] = (long long)(b_ok[j] ? n : 0) * K;
  }

  constexpr int NST = 3;      // k-tiles in flight in registers: hides the ~1 us L2/HBM round trip of short-grid launches
  uint4 ra[NST][AR], rb[NST][BR];
  auto a_pixel = [&](int i, int kh, int kw, long long& off) -> bool {
    int hi, wi;
    bool ok = a_ok[i];
    if (MODE == 0) { hi = a_h[i] + kh; wi = a_w[i] + kw; }
    else {
      int th = a_h[i] - kh, tw = a_w[i] - kw;
      hi = th / p.stride; wi = tw / p.stride;
      ok = ok && th >= 0 && tw >= 0 && (hi * p.stride == th) && (wi * p.stride == tw);
    }
    ok = ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
    off = a_base[i] + ((long long)hi * p.W + wi) * p.ldin;
    return ok;
  };
  auto load_tile = [&](int kt, uint4 (&ra_)[AR], uint4 (&rb_)[BR]) {
    const int kc = kt * BK + chunk * EPC;       // first k index of this thread's 16-byte chunk
    if constexpr (VEC) {
      const bool kok = kc < K;
      const int tap = kok ? kc / p.C : 0;
      const int c0 = kc - tap * p.C;
      const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        long long off;
        const bool ok = a_pixel(i, kh, kw, off) && kok;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ok) v = *reinterpret_cast<const uint4*>(inp + off + c0);
        ra_[i] = v;
      }
#pragma unroll
      for (int j = 0; j < BR; ++j) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (b_ok[j] && kok) v = *reinterpret_cast<const uint4*>(wp + b_off[j] + kc);
        rb_[j] = v;
      }
    } else {
      T ea[AR][EPC], eb[BR][EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const int kk = kc + e;
        const bool kok = kk < K;
        const int tap = kok ? kk / p.C : 0;
        const int cc = kk - tap * p.C;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
          long long off;
          const bool ok = a_pixel(i, kh, kw, off) && kok;
          ea[i][e] = ok ? inp[off + cc] : from_f32<T>(0.f);
        }
#pragma unroll
        for (int j = 0; j < BR; ++j) eb[j][e] = (b_ok[j] && kok) ? wp[b_off[j] + kk] : from_f32<T>(0.f);
      }
#pragma unroll
      for (int i = 0; i < AR; ++i) ra_[i] = *reinterpret_cast<const uint4*>(&ea[i][0]);
#pragma unroll
      for (int j = 0; j < BR; ++j) rb_[j] = *reinterpret_cast<const uint4*>(&eb[j][0]);
    }
  };

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int d = 0; d < NST; ++d)
    if (d < nkt) load_tile(d, ra[d], rb[d]);
  const int frow = lane & 31, fh = lane >> 5;
  for (int kt0 = 0; kt0 < nkt; kt0 += NST) {
#pragma unroll
    for (int d = 0; d < NST; ++d) {
      const int kt = kt0 + d;
      if (kt < nkt) {
#pragma unroll
        for (int i = 0; i < AR; ++i) *reinterpret_cast<uint4*>(sA + (row0 + 32 * i) * PITCH + chunk * 16) = ra[d][i];
#pragma unroll
        for (int j = 0; j < BR; ++j) *reinterpret_cast<uint4*>(sB + (row0 + 32 * j) * PITCH + chunk * 16) = rb[d][j];
        __syncthreads();
        if (kt + NST < nkt) load_tile(kt + NST, ra[d], rb[d]);     // refill the stage just written to LDS
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          uint4 fa[TM], fb[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i)
            fa[i] = *reinterpret_cast<const uint4*>(sA + ((wr * TM + i) * 32 + frow) * PITCH + (2 * s + fh) * 16);
#pragma unroll
          for (int j = 0; j < TN; ++j)
            fb[j] = *reinterpret_cast<const uint4*>(sB + ((wc * TN + j) * 32 + frow) * PITCH + (2 * s + fh) * 16);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) mma_chunk<T>(acc[i][j], fa[i], fb[j]);
        }
        __syncthreads();
      }
    }
  }

  // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const T* resp = (const T*)p.res;
  float st_s[TN], st_q[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) { st_s[j] = 0.f; st_q[j] = 0.f; }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (wr * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
      const long long m = (long long)bm * BM + row;
      if (m >= M) continue;
      const int nb = (int)(m / OHW);
      const int pix = (int)(m - (long long)nb * OHW);
      const long long obase = (long long)nb * p.out_bs + (long long)pix * p.ldout;
      const long long rbase = (long long)nb * p.res_bs + (long long)pix * p.ldres;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = bn * BN + (wc * TN + j) * 32 + frow;
        if (n >= p.OC) continue;
        float v = acc[i][j][r];
        if (p.bias) v += p.bias[n];
        if (resp) v += to_f32(resp[rbase + n]);
        if (p.relu) v = fmaxf(v, 0.f);
        if (p.out_f32) ((float*)p.out)[obase + n] = v;
        else {
          const T q = from_f32<T>(v);
          ((T*)p.out)[obase + n] = q;
          v = to_f32(q);                 // statistics of what the next kernel will actually read
        }
        st_s[j] += v;
        st_q[j] = fmaf(v, v, st_q[j]);
      }
    }
  }
  if (p.stats) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const float s2 = st_s[j] + __shfl_xor(st_s[j], 32, 64);     // the two half-waves hold the same column
      const float q2 = st_q[j] + __shfl_xor(st_q[j], 32, 64);
      const int n = bn * BN + (wc * TN + j) * 32 + frow;
      if (fh == 0 && n < p.OC) {
        double* rep = p.stats + (long long)(bm & 7) * 2 * p.OC;
        atomicAdd(rep + n, (double)s2);
        atomicAdd(rep + p.OC + n, (double)q2);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// wgrad: dW[oc][k] += sum_m dy[m][oc] * xcol[m][k], 128(oc) x 128(k) tile per block, reduction over
// pixels m split across blockIdx.z, fp32 atomics into dW.  Both operands are pixel-major in memory, so
// the MFMA fragments (8 consecutive m per lane) come from LDS through the transposing read
// ds_read_b64_tr_b16 (bf16) or plain ds_read_b32 (f32, one k-slot per lane).
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
  const void* x;
  const void* dy;
  float* dw;
  int N, H, W, C, ldx;
  long long x_bs;
  int OH, OW, OC, lddy;
  long long dy_bs;
  int KH, KW, stride, pad;
  int tiles_per_split;  // number of BKm pixel tiles each z-slice processes
  float* dbias;         // optional [OC]: += sum_m dy[m][oc] (bias gradient), accumulated by the k-tile-0 blocks from the dy tiles they stream
};

template <class T>
struct WgradCfg;
template <>
struct WgradCfg<bf16_t> {
  static constexpr int CPR = 16, RPP = 16, BKM = 64, PITCH = 320;
};
template <>
struct WgradCfg<float> {
  static constexpr int CPR = 32, RPP = 8, BKM = 32, PITCH = 528;
};

template <class T, bool VEC>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs p) {
  using Cfg = WgradCfg<T>;
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int CPR = Cfg::CPR, RPP = Cfg::RPP, BKM = Cfg::BKM, PITCH = Cfg::PITCH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sP = smem;                 // dy tile  [BKM][128 oc]
  unsigned char* sQ = smem + BKM * PITCH;   // x  tile  [BKM][128 k ]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int K = p.KH * p.KW * p.C;
  const int oc0 = blockIdx.y * 128, k0 = blockIdx.x * 128;
  const int OHW = p.OH * p.OW;
  const long long M = (long long)p.N * OHW;
  const T* __restrict__ xp = (const T*)p.x;
  const T* __restrict__ dyp = (const T*)p.dy;

  const int col = tid % CPR, prow = tid / CPR;
  // this thread's fixed k-chunk (Q) and oc-chunk (P)
  const int kq = k0 + col * EPC;
  const bool kq_ok = kq < K;
  const int tap = kq_ok ? kq / p.C : 0;
  const int cq = kq - tap * p.C;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int ocp = oc0 + col * EPC;
  const bool ocp_ok = ocp < p.OC;

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  constexpr int NST = 3;
  uint4 rp[NST][4], rq[NST][4];
  const bool do_bias = p.dbias != nullptr && blockIdx.x == 0;
  float bsum[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) bsum[e] = 0.f;
  // scalar path: per-element decode of this thread's fixed chunk columns (done once)
  int e_c[EPC], e_kh[EPC], e_kw[EPC];
  bool e_kok[EPC], e_ocok[EPC];
  if constexpr (!VEC) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const int kk = kq + e;
      e_kok[e] = kk < K;
      const int tp = e_kok[e] ? kk / p.C : 0;
      e_c[e] = kk - tp * p.C;
      e_kh[e] = tp / p.KW;
      e_kw[e] = tp - e_kh[e] * p.KW;
      e_ocok[e] = ocp + e < p.OC;
    }
  }
  auto load_tile = [&](long long mt, uint4 (&rp_)[4], uint4 (&rq_)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long m = mt * BKM + prow + RPP * i;
      uint4 vp = make_uint4(0, 0, 0, 0), vq = make_uint4(0, 0, 0, 0);
      if (m < M) {
        const int nb = (int)(m / OHW);
        const int pix = (int)(m - (long long)nb * OHW);
        const int oh = pix / p.OW, ow = pix - oh * p.OW;
        const T* dyrow = dyp + (long long)nb * p.dy_bs + (long long)pix * p.lddy;
        const T* ximg = xp + (long long)nb * p.x_bs;
        if constexpr (VEC) {
          if (ocp_ok) vp = *reinterpret_cast<const uint4*>(dyrow + ocp);
          const int hi = oh * p.stride - p.pad + kh, wi = ow * p.stride - p.pad + kw;
          if (kq_ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
            vq = *reinterpret_cast<const uint4*>(ximg + ((long long)hi * p.W + wi) * p.ldx + cq);
        } else {
          T ep[EPC], eq[EPC];
#pragma unroll
          for (int e = 0; e < EPC; ++e) {
            ep[e] = e_ocok[e] ? dyrow[ocp + e] : from_f32<T>(0.f);
            const int hi = oh * p.stride - p.pad + e_kh[e], wi = ow * p.stride - p.pad + e_kw[e];
            const bool ok = e_kok[e] && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            eq[e] = ok ? ximg[((long long)hi * p.W + wi) * p.ldx + e_c[e]] : from_f32<T>(0.f);
          }
          vp = *reinterpret_cast<const uint4*>(&ep[0]);
          vq = *reinterpret_cast<const uint4*>(&eq[0]);
        }
      }
      rp_[i] = vp;
      rq_[i] = vq;
    }
  };

  const long long mt_begin = (long long)blockIdx.z * p.tiles_per_split;
  const long long mt_total = (M + BKM - 1) / BKM;
  long long mt_end = mt_begin + p.tiles_per_split;
  if (mt_end > mt_total) mt_end = mt_total;
  if (mt_begin >= mt_end) return;

#pragma unroll
  for (int d = 0; d < NST; ++d)
    if (mt_begin + d < mt_end) load_tile(mt_begin + d, rp[d], rq[d]);
  for (long long mt0 = mt_begin; mt0 < mt_end; mt0 += NST) {
#pragma unroll
    for (int d = 0; d < NST; ++d) {
      const long long mt = mt0 + d;
      if (mt < mt_end) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<uint4*>(sP + (prow + RPP * i) * PITCH + col * 16) = rp[d][i];
      *reinterpret_cast<uint4*>(sQ + (prow + RPP * i) * PITCH + col * 16) = rq[d][i];
      if (do_bias) {
        const T* ev = reinterpret_cast<const T*>(&rp[d][i]);
#pragma unroll
        for (int e = 0; e < EPC; ++e) bsum[e] += to_f32(ev[e]);
      }
    }
    __syncthreads();
    if (mt + NST < mt_end) load_tile(mt + NST, rp[d], rq[d]);
    if constexpr (sizeof(T) == 2) {
      // lane: group g = lane>>4 (h = g>>1 picks k rows 8h.., half = g&1 picks 16 columns), t = lane&15.
      const int g = lane >> 4, t = lane & 15;
      const int h = g >> 1, half = g & 1, q = t >> 2, pp = t & 3;
#pragma unroll
      for (int s = 0; s < BKM / 16; ++s) {
        uint4 fa[2], fb[2];
        const int r0 = 16 * s + 8 * h + q;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int cb = (wr * 64 + i * 32 + 16 * half + 4 * pp) * 2;
          short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(sP + r0 * PITCH + cb));
          short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(sP + (r0 + 4) * PITCH + cb));
          uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
          fa[i] = make_uint4(l2.x, l2.y, h2.x, h2.y);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int cb = (wc * 64 + j * 32 + 16 * half + 4 * pp) * 2;
          short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(sQ + r0 * PITCH + cb));
          short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(sQ + (r0 + 4) * PITCH + cb));
          uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
          fb[j] = make_uint4(l2.x, l2.y, h2.x, h2.y);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[i]), __builtin_bit_cast(bf16x8_t, fb[j]), acc[i][j], 0, 0, 0);
      }
    } else {
      const int r = lane & 31, h = lane >> 5;
#pragma unroll 4
      for (int s = 0; s < BKM / 2; ++s) {
        float fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const float*>(sP + (2 * s + h) * PITCH + (wr * 64 + i * 32 + r) * 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const float*>(sQ + (2 * s + h) * PITCH + (wc * 64 + j * 32 + r) * 4);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
      }
    }
  }

  if (do_bias) {      // block-level reduction over the RPP row lanes in LDS, then ONE atomic per output channel
    float* sb = reinterpret_cast<float*>(smem);     // [RPP][128]; the k-loop's last barrier has retired every LDS read
#pragma unroll
    for (int e = 0; e < EPC; ++e) sb[prow * 128 + col * EPC + e] = bsum[e];
    __syncthreads();
    if (tid < 128 && oc0 + tid < p.OC) {
      float t = 0.f;
      for (int r = 0; r < RPP; ++r) t += sb[r * 128 + tid];
      atomicAdd(p.dbias + oc0 + tid, t);
    }
  }
  const int frow = lane & 31, fh = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int oc = oc0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
      if (oc >= p.OC) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int k = k0 + wc * 64 + j * 32 + frow;
        if (k < K) atomicAdd(p.dw + (long long)oc * K + k, acc[i][j][r]);
      }
    }
}

// ------------------------------------------------------------------------------------------------
// host dispatch
// ------------------------------------------------------------------------------------------------
template <class T, int TM, int TN, int WR, int WC, int MODE, bool VEC>
static int launch_igemm(const ConvArgs& a, hipStream_t st) {
  constexpr int BM = WR * TM * 32, BN = WC * TN * 32;
  const long long M = (long long)a.N * a.OH * a.OW;
  const long long grid = ((M + BM - 1) / BM) * ((a.OC + BN - 1) / BN);
  const size_t lds = (size_t)(BM + BN) * 144;
  hipLaunchKernelGGL((igemm_kernel<T, TM, TN, WR, WC, MODE, VEC>), dim3((unsigned)grid), dim3(256), lds, st, a);
  return check_launch("emrt_conv2d");
}

template <class T, int MODE, bool VEC>
static int conv_pick_tile(const ConvArgs& a, hipStream_t st) {
  // largest tile that still leaves >= 2-3 blocks per CU; everything else takes 64x64
  const long long M = (long long)a.N * a.OH * a.OW;
  auto blocks = [&](int bmv, int bnv) { return ((M + bmv - 1) / bmv) * ((a.OC + bnv - 1) / bnv); };
  if (a.OC <= 32) return launch_igemm<T, 2, 1, 4, 1, MODE, VEC>(a, st);
  // measured: 64x64 tiles at >= 2 blocks per CU beat 128x64 tiles at ~1 per CU on the 32x32 / token GEMMs
  if (a.OC > 64 && blocks(128, 128) >= 384) return launch_igemm<T, 2, 2, 2, 2, MODE, VEC>(a, st);
  if (blocks(128, 64) >= 768) return launch_igemm<T, 2, 1, 2, 2, MODE, VEC>(a, st);
  return launch_igemm<T, 1, 1, 2, 2, MODE, VEC>(a, st);
}

template <class T, int MODE>
static int conv_dispatch(const ConvArgs& a, hipStream_t st) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const bool vec = (a.C % EPC == 0) && (a.ldin % EPC == 0) && (a.in_bs % EPC == 0) &&
                   (((uintptr_t)a.in) % 16 == 0) && (((uintptr_t)a.w) % 16 == 0);
  return vec ? conv_pick_tile<T, MODE, true>(a, st) : conv_pick_tile<T, MODE, false>(a, st);
}

extern "C" int emrt_conv2d(const void* in, const void* w_packed, void* out, const float* bias, const void* residual,
                           int N, int H, int W, int C, int ldin, long long in_bs,
                           int OH, int OW, int OC, int ldout, long long out_bs,
                           int ldres, long long res_bs,
                           int KH, int KW, int stride, int pad,
                           int mode, int relu, int out_f32, double* bn_stats, int dtype, void* stream) {
  EMRT_REQUIRE(in && w_packed && out, "null pointer");
  EMRT_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && OH > 0 && OW > 0 && OC > 0, "bad dims");
  EMRT_REQUIRE(KH > 0 && KW > 0 && stride > 0 && pad >= 0, "bad kernel geometry");
  EMRT_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (fwd) or 1 (dgrad)");
  EMRT_REQUIRE(dtype == EMRT_F32 || dtype == EMRT_BF16, "dtype must be 0 (f32) or 1 (bf16)");
  if (mode == 0) {
    EMRT_REQUIRE(OH == (H + 2 * pad - KH) / stride + 1 && OW == (W + 2 * pad - KW) / stride + 1, "fwd: output size mismatch");
  } else {
    EMRT_REQUIRE(H == (OH + 2 * pad - KH) / stride + 1 && W == (OW + 2 * pad - KW) / stride + 1, "dgrad: size mismatch");
  }
  ConvArgs a;
  a.in = in; a.w = w_packed; a.out = out; a.bias = bias; a.res = residual;
  a.N = N; a.H = H; a.W = W; a.C = C; a.ldin = ldin; a.in_bs = in_bs;
  a.OH = OH; a.OW = OW; a.OC = OC; a.ldout = ldout; a.out_bs = out_bs;
  a.ldres = ldres; a.res_bs = res_bs;
  a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.relu = relu; a.out_f32 = out_f32; a.stats = bn_stats;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EMRT_F32) return mode == 0 ? conv_dispatch<float, 0>(a, st) : conv_dispatch<float, 1>(a, st);
  return mode == 0 ? conv_dispatch<bf16_t, 0>(a, st) : conv_dispatch<bf16_t, 1>(a, st);
}

template <class T>
static int wgrad_dispatch(const WgradArgs& a0, hipStream_t st) {
  using Cfg = WgradCfg<T>;
  constexpr int EPC = 16 / (int)sizeof(T);
  WgradArgs a = a0;
  const int K = a.KH * a.KW * a.C;
  const long long M = (long long)a.N * a.OH * a.OW;
  const bool vec = (a.C % EPC == 0) && (a.OC % EPC == 0) && (a.ldx % EPC == 0) && (a.lddy % EPC == 0) &&
                   (a.x_bs % EPC == 0) && (a.dy_bs % EPC == 0) && (((uintptr_t)a.x) % 16 == 0) && (((uintptr_t)a.dy) % 16 == 0);
  const int tx = (K + 127) / 128, ty = (a.OC + 127) / 128;
  const long long mt_total = (M + Cfg::BKM - 1) / Cfg::BKM;
  long long want = (512 + (long long)tx * ty - 1) / ((long long)tx * ty);   // ~2 blocks per CU; every extra slice re-adds dW atomically
  if (want < 1) want = 1;
  long long max_split = mt_total / 8;
  if (max_split < 1) max_split = 1;
  long long S = want < max_split ? want : max_split;
  a.tiles_per_split = (int)((mt_total + S - 1) / S);
  S = (mt_total + a.tiles_per_split - 1) / a.tiles_per_split;
  const size_t lds = 2 * (size_t)Cfg::BKM * Cfg::PITCH;
  if (vec) hipLaunchKernelGGL((wgrad_kernel<T, true>), dim3(tx, ty, (unsigned)S), dim3(256), lds, st, a);
  else hipLaunchKernelGGL((wgrad_kernel<T, false>), dim3(tx, ty, (unsigned)S), dim3(256), lds, st, a);
  return check_launch("emrt_conv2d_wgrad");
}

extern "C" int emrt_conv2d_wgrad(const void* x, const void* dy, float* dw,
                                 int N, int H, int W, int C, int ldx, long long x_bs,
                                 int OH, int OW, int OC, int lddy, long long dy_bs,
                                 int KH, int KW, int stride, int pad, float* dbias, int dtype, void* stream) {
  EMRT_REQUIRE(x && dy && dw, "null pointer");
  EMRT_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && OH > 0 && OW > 0 && OC > 0, "bad dims");
  EMRT_REQUIRE(OH == (H + 2 * pad - KH) / stride + 1 && OW == (W + 2 * pad - KW) / stride + 1, "output size mismatch");
  EMRT_REQUIRE(dtype == EMRT_F32 || dtype == EMRT_BF16, "dtype must be 0 (f32) or 1 (bf16)");
  WgradArgs a;
  a.x = x; a.dy = dy; a.dw = dw;
  a.N = N; a.H = H; a.W = W; a.C = C; a.ldx = ldx; a.x_bs = x_bs;
  a.OH = OH; a.OW = OW; a.OC = OC; a.lddy = lddy; a.dy_bs = dy_bs;
  a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.tiles_per_split = 0; a.dbias = dbias;
  hipStream_t st = (hipStream_t)stream;
  return dtype == EMRT_F32 ? wgrad_dispatch<float>(a, st) : wgrad_dispatch<bf16_t>(a, st);
}
