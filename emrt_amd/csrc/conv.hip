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
#include "bn_operand.hpp"
#include <stdlib.h>
#include <type_traits>

using namespace emrt;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

struct ConvArgs {
  const void* in;
  const void* w;
  void* out;
  const float* bias;
  const float* scale;   // optional per-output-channel factor applied to the accumulator before the bias: an eval-mode BatchNorm folded
                        // into the convolution (y = conv * gamma / sqrt(var + eps) + (beta - mean * gamma / sqrt(var + eps)))
  const void* res;
  int N, H, W, C, ldin;
  long long in_bs;
  int OH, OW, OC, ldout;
  long long out_bs;
  int ldres;
  long long res_bs;
  int KH, KW, stride, pad;
  int dil;         // dilation of the kernel taps (resnet50c's dilated stages, backbones/resnet.py:65-66); 1 everywhere else
  int relu, out_f32;
  int cmajor;      // igemm8p only: channel-block-major k order (developer A/B knob igemm8p_cmajor), 0 everywhere else
  double* stats;   // optional [8][2*OC]: per-channel sum / sum of squares of the STORED outputs (BatchNorm statistics);
                   // fp64 atomics spread over 8 replicas (by M-tile index) so that blocks do not pile onto one address
  // optional ReLU mask: outputs are zeroed where mask_y <= 0 (same geometry as the output, own strides), and with it the
  // statistics become (sum v, sum v * mask_y).  dgrad of a conv whose input is y = relu(BatchNorm(x)) uses it to produce
  // the masked dy AND that BatchNorm's backward sums in one pass: where y > 0, xhat = (y - beta) / gamma.
  const void* mask_y;
  float mask_scale;   // kept outputs are multiplied by this (1 for a ReLU; 1/(1-p) for dropout(relu(.)))
  // optional: the second statistic multiplies with this tensor instead of mask_y (same geometry, own strides): the
  // BatchNorm INPUT when mask_y is the output of relu(BatchNorm(x) + residual), whose xhat cannot be recovered from y
  const void* stat_x;
  int ldsx;
  long long sx_bs;
  int ldy;
  long long y_bs;
  // cross-block K split (igemm_body<..., XK>): the grid is xk_S copies of the tile grid, copy s walks the k-tiles [s * per, (s + 1) * per);
  // partial fp32 tiles go to xk_part [tile][s][BM][BN], arrivals are counted in xk_tick[tile], and the LAST block to arrive sums the
  // partials (in s order: bit-reproducible) and runs the epilogue.  0 / nullptr everywhere else.
  int xk_S;
  float* xk_part;
  unsigned* xk_tick;
  // optional inverted dropout of the stored outputs, after the ReLU (emrt_conv2d_drop: linear1 -> ReLU -> Dropout of the FFN, t_e_d.py:157-161):
  // kept values are multiplied by 1 / (1 - p); the mask comes from drop_words8(seed, salt, (row * OC + col) / 8)
  const unsigned long long* drop_seed;
  unsigned drop_salt;
  float drop_p;
  // A-operand BatchNorm (igemm_body<..., BNA>; emrt_conv2d_bna): `in` is the RAW output of the producing conv whose training-mode BatchNorm (+ ReLU) has
  // not been applied.  Every block derives the per-channel scale / shift from the complete fp64 batch sums in its preamble (bn_operand.hpp, as the
  // other consumers that apply a BatchNorm on load) and turns every 16-byte chunk it loads into [relu](x * scale + shift) between the global load and
  // the LDS write -- the same fmaf + max + rounding as bn_apply_kernel, so the GEMM sees the bits the separate launch would have stored.  The tiles of
  // the first column of the tile grid (bn == 0) also WRITE the transformed chunks of the centre tap to a_out (dense [N][H][W][C]): backward (the weight
  // gradient's operand, the ReLU mask and BatchNorm sums of this layer's data gradient) keeps reading a materialised map, only the emrt_bn_apply
  // launch and its read of the raw map are gone.  bna.sums == nullptr everywhere else.
  BnOperand bna = {};
  void* a_out = nullptr;
};

template <class T>
__device__ __forceinline__ void mma_chunk(f32x16_t& acc, const uint4& a, const uint4& b);
template <>
__device__ __forceinline__ void mma_chunk<bf16_t>(f32x16_t& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
template <>
__device__ __forceinline__ void mma_chunk<f16_t>(f32x16_t& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_chunk<float>(f32x16_t& acc, const uint4& a, const uint4& b) {
  // lane (r, h) holds 4 consecutive k of its row; MFMA e pairs k-slot h with element e of both operands.
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

// ---- raw buffer loads -------------------------------------------------------------------------
// Every operand is addressed as base + 32-bit byte offset through a buffer descriptor whose range is BUF_RANGE bytes.
// Offsets with bit 31 set are out of range and the hardware returns zeros for them (the host checks that no operand
// spans 2 GiB or more, so a valid offset never has that bit).
constexpr unsigned BUF_RANGE = 0x80000000u;
constexpr unsigned BUF_OOB = 0x80000000u;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}
template <class T>
__device__ __forceinline__ uint32_t buf_load_elem(__amdgpu_buffer_rsrc_t rs, unsigned off);   // element bits, zero-extended
template <>
__device__ __forceinline__ uint32_t buf_load_elem<bf16_t>(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  return (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)off, 0, 0);
}
template <>
__device__ __forceinline__ uint32_t buf_load_elem<f16_t>(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  return (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs, (int)off, 0, 0);
}
template <>
__device__ __forceinline__ uint32_t buf_load_elem<float>(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// fwd / dgrad implicit GEMM.  256 threads = 4 waves laid out WR x WC, each wave TM x TN tiles of 32x32.
// LDS: two [BM + BN] x 128-byte k-tile buffers (rows padded to 144 B: conflict-free ds_read_b128), ONE barrier per
// k-tile.  Global loads run NST k-tiles ahead in a register ring.  Every load is unconditional (an out-of-range or
// padding chunk reads the operand's base address and is zeroed by a select afterwards) and the ring is refilled in
// straight-line code, so that the compiler can wait with a counted vmcnt(N) for the oldest stage only; predicated
// loads behind branches made it drain the whole ring (vmcnt(0)) at every k-tile.
// ------------------------------------------------------------------------------------------------
// VEC = true : C % (16 B of elements) == 0 and 16-byte aligned rows -> one 16-byte load per chunk; the last k-tile may
//              be partial (K % BK != 0, e.g. the fused 432-wide offsets|logits projection) and is zero-filled.
// VEC = false: any C / alignment (7x7x3 stem, 3x3x3 branch conv, 6-class logits): chunks are assembled from element loads.
// G > 1 (in-block K split): the block has G groups of 4 waves; group g walks the k-tiles g, g+G, ... with its own LDS
// buffers and the groups' accumulators are summed through LDS before the epilogue.  Small-M layers (16x16 / 8x8 maps)
// launch fewer blocks than there are CUs and are bound by the latency of the serial k loop: the split shortens that
// chain G-fold with waves the CU would otherwise leave idle, without global atomics or a second pass.
// Occupancy: the 64x64 tile without K split is what every many-block, short-K layer runs (token linears, 1x1 convs);
// those are prologue / epilogue dominated, so it is held to 128 registers = 4 blocks per CU (measured: 180 registers, i.e.
// 2 blocks per CU, cost 25-35 % on those shapes).
// S2 (data gradient of a stride-2 convolution, dilation 1, even output size, C % BK == 0; 64x64 tile): an output pixel of parity class
// (oh & 1, ow & 1) only receives the taps with kh = (oh + pad) mod 2, kw = (ow + pad) mod 2 -- a quarter of a 3x3 kernel's taps on average
// and, for a 1x1 kernel, ONE class in four receives anything at all -- while the generic loop walks every tap and multiplies zeros
// (out-of-range loads) for the others.  Here the rows of the implicit GEMM are enumerated class by class (m' = class * M/4 + the pixel's
// index on the half-resolution grid), so that a 64-row tile is parity-pure (host: M/4 % 64 == 0) and its k loop visits its own taps only;
// a class without taps skips the loop and runs the epilogue (addend / mask / statistics) on zeros.  Same result, bit for bit, as the
// generic kernel: the skipped k-tiles contributed exact zeros.
// XK (cross-block K split, 64x64 vector-path tile): few-tile, long-K layers (layer4's 8x8 maps, the auxiliary head's 16x16x1024 3x3, layer3's
// 3x3 convs: 64-128 tiles of 64x64) leave most CUs idle, and a CU streams its operands at ~50-70 GB/s whatever runs on it (G = 1 / 2 /
// 4 wave groups per block give 27 / 24 / 21 us on 8x8x512 -> 512 3x3 -- the L1 fill rate of the FEW CUs that have a block is the bound).  Here the
// k range is cut over S blocks per tile, so S times as many CUs pull the same bytes; each block leaves its fp32 partial tile in the registered
// scratch with write-through (sc1) stores, every storing wave drains its stores, one lane takes a ticket (agent-scope atomic add), and the
// block whose ticket is the last one acquires (agent scope), sums the S partials in s order and runs the usual epilogue (bias / residual / mask
// / BatchNorm sums / store) -- no finishing launch.  MI355X_MICROARCH.md "splitk-seam"; cdna_hip_programming.md Guideline 16, R1.
template <class T, int TM, int TN, int WR, int WC, int MODE, bool VEC, int NST, int G, bool S2 = false, bool XK = false, bool DROP = false, bool BNA = false>
__device__ __forceinline__ void igemm_body(const ConvArgs& p, const int block_id, const int block_count, unsigned char* smem_all) {
  static_assert(!S2 || (MODE == 1 && VEC && G == 1 && TM == 1 && TN == 1), "S2 is the 64x64 vector-path data gradient");
  static_assert(!XK || (VEC && G == 1 && !S2), "XK is a vector-path tile without the in-block K split");
  static_assert(!BNA || (MODE == 0 && VEC && !S2 && !DROP), "BNA is the forward vector path");
  constexpr int BM = WR * TM * 32, BN = WC * TN * 32;
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int BK = 8 * EPC;
  constexpr int PITCH = 144;
  constexpr int AR = BM / 32, BR = BN / 32;
  constexpr int STAGE_BYTES = (BM + BN) * PITCH;

  const int grp = G == 1 ? 0 : (int)threadIdx.x >> 8;
  unsigned char* smem = smem_all + grp * 2 * STAGE_BYTES;
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WC, wc = wave % WC;
  const int tiles_n = (p.OC + BN - 1) / BN;
  // consecutive workgroup ids are dealt round-robin to the 8 XCDs: give each XCD a contiguous range of tiles so that the
  // tiles that share an activation row block (same bm) hit the same L2
  int bid = block_id;
  const int nblk = block_count;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
  int xk_s = 0, xk_tile = 0;
  if constexpr (XK) {      // copy-major: the blocks of one k range are neighbours (they share the weight rows of that range)
    const int ntile = nblk / p.xk_S;
    xk_s = bid / ntile;
    xk_tile = bid - xk_s * ntile;
    bid = xk_tile;
  }
  const int bm = bid / tiles_n, bn = bid % tiles_n;
  const int OHW = p.OH * p.OW;
  const long long M = (long long)p.N * OHW;
  const int chunk = tid & 7, row0 = tid >> 3;

  const int K = p.KH * p.KW * p.C;
  // S2: this tile's parity class and its tap lattice kh = s2_kh0 + 2 i, kw = s2_kw0 + 2 j
  const unsigned s2_mq = S2 ? (unsigned)(M >> 2) : 1u;                       // rows per class
  const int s2_cls = S2 ? (int)(((unsigned)bm * (unsigned)BM) / s2_mq) : 0;  // (oh & 1) * 2 + (ow & 1) of every row of the tile
  const int s2_kh0 = S2 ? (((s2_cls >> 1) + p.pad) & 1) : 0, s2_kw0 = S2 ? (((s2_cls & 1) + p.pad) & 1) : 0;
  const int s2_nh = S2 ? (p.KH - s2_kh0 + 1) / 2 : 0, s2_nw = S2 ? (p.KW - s2_kw0 + 1) / 2 : 0;
  int nkt = S2 ? s2_nh * s2_nw * (p.C / BK) : ((K + BK - 1) / BK + G - 1) / G;       // k-tiles walked by each group (tiles past K are all-zero)
  int xk_kt0 = 0;
  if constexpr (XK) {
    const int per = (nkt + p.xk_S - 1) / p.xk_S;
    xk_kt0 = xk_s * per;
    nkt = nkt - xk_kt0 < per ? nkt - xk_kt0 : per;      // (<= 0 for a trailing copy without k-tiles: it still arrives, with zeros)
  }

  // Operands are read through raw buffer descriptors: an out-of-range byte offset returns zeros, which is how padding
  // taps, stride holes, rows beyond M / OC and the k tail are zero-filled without a branch or a select on the data.
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)BUF_RANGE, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)BUF_RANGE, 0x00020000);

  int a_h[AR], a_w[AR];
  unsigned a_base[AR];      // byte offset of the row's image
  bool a_ok[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    // 32-bit arithmetic: the host guarantees N * OH * OW < 2^31 (a 64-bit division is ~150 instructions, and there was one per row
    // here and one per row in the epilogue: ~1 us of every launch of the small layers)
    const unsigned m = (unsigned)bm * (unsigned)BM + (unsigned)(row0 + 32 * i);
    a_ok[i] = (long long)m < M;
    const unsigned mm = a_ok[i] ? m : 0u;
    int nb, oh, ow;
    if constexpr (S2) {      // class-major row order: index on the half-resolution grid, then the class's parity offsets
      const unsigned q = mm - (unsigned)s2_cls * s2_mq, hw2 = (unsigned)(OHW >> 2), ow2 = (unsigned)(p.OW >> 1);
      nb = (int)(q / hw2);
      const unsigned r2 = q - (unsigned)nb * hw2;
      const int oh2 = (int)(r2 / ow2);
      oh = 2 * oh2 + (s2_cls >> 1);
      ow = 2 * (int)(r2 - (unsigned)oh2 * ow2) + (s2_cls & 1);
    } else {
      nb = (int)(mm / (unsigned)OHW);
      const int r = (int)(mm - (unsigned)nb * (unsigned)OHW);
      oh = (int)((unsigned)r / (unsigned)p.OW);
      ow = r - oh * p.OW;
    }
    a_base[i] = (unsigned)((long long)nb * p.in_bs * (long long)sizeof(T));
    if (MODE == 0) { a_h[i] = oh * p.stride - p.pad; a_w[i] = ow * p.stride - p.pad; }
    else { a_h[i] = oh + p.pad; a_w[i] = ow + p.pad; }
  }
  unsigned b_off[BR];       // byte offset of the weight row, BUF_OOB when the row is beyond OC
#pragma unroll
  for (int j = 0; j < BR; ++j) {
    int n = bn * BN + row0 + 32 * j;
    b_off[j] = n < p.OC ? (unsigned)((long long)n * K * (long long)sizeof(T)) : BUF_OOB;
  }

  // BNA: scale / shift table [2][C] behind every group's k-tile buffers (dead once the k loop is over: the parked accumulators may overwrite it)
  // and, per ring stage, what the transform needs to know about the chunk that stage holds: bits 0-15 its first channel, bit 16 "centre tap", bit 17 + i "row i was in range" (padding taps stay exact zeros)
  float* bn_tab = reinterpret_cast<float*>(smem_all + (size_t)G * 2 * STAGE_BYTES);
  unsigned rmeta[BNA ? NST : 1];
  unsigned wb_off[BNA ? AR : 1];      // byte offset of the row's own pixel in a_out (BUF_OOB: not this block's to write)
  __amdgpu_buffer_rsrc_t rs_aout;
  if constexpr (BNA) {
    rs_aout = __builtin_amdgcn_make_buffer_rsrc(p.a_out, 0, (int)BUF_RANGE, 0x00020000);
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const unsigned m = (unsigned)bm * (unsigned)BM + (unsigned)(row0 + 32 * i);
      wb_off[i] = (bn == 0 && a_ok[i]) ? m * (unsigned)p.C * (unsigned)sizeof(T) : BUF_OOB;
    }
  }
  uint4 ra[NST][AR], rb[NST][BR];
  // byte offset of input pixel (row i, tap kh,kw), or BUF_OOB when the tap falls on padding / a stride hole / m >= M
  auto a_pixel = [&](int i, int kh, int kw) -> unsigned {
    int hi, wi;
    bool ok = a_ok[i];
    if (MODE == 0) { hi = a_h[i] + kh * p.dil; wi = a_w[i] + kw * p.dil; }
    else {
      const int th = a_h[i] - kh * p.dil, tw = a_w[i] - kw * p.dil;
      if (p.stride == 1) { hi = th; wi = tw; }                 // (negative values fail the range test below)
      else if (p.stride == 2) { hi = th >> 1; wi = tw >> 1; ok = ok && ((th | tw) & 1) == 0; }
      else {
        hi = th / p.stride; wi = tw / p.stride;
        ok = ok && th >= 0 && tw >= 0 && (hi * p.stride == th) && (wi * p.stride == tw);
      }
    }
    ok = ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
    const unsigned off = a_base[i] + (unsigned)((hi * p.W + wi) * p.ldin) * (unsigned)sizeof(T);
    return ok ? off : BUF_OOB;
  };
  // The pixel of a row only changes when the tap does (never for 1x1 kernels: every linear layer, the bottleneck 1x1
  // convs; every C / BK tiles for 3x3): its byte offset is cached and re-derived on a tap change only.  The k loop of the
  // 64x64 tile is bound by this address arithmetic (VALU), not by the MFMAs.
  unsigned a_cur[AR];
  const bool one_tap = p.KH * p.KW == 1;
  // loader cursor: k index / tap / channel of this thread's 16-byte chunk in the NEXT k-tile to fetch.  It keeps
  // advancing past K (the ring prefetches beyond the last tile): those chunks are all-zero.
  constexpr int KSTEP = BK * G;        // a group's consecutive tiles are G tiles apart
  int ld_kc = (grp + xk_kt0) * BK + chunk * EPC, ld_kh, ld_kw, ld_c0;
  if constexpr (S2) {
    ld_c0 = chunk * EPC;
    ld_kh = (s2_nh > 0 && s2_nw > 0) ? s2_kh0 : p.KH;      // a class without taps: every prefetch of the ring is out of range (zeros, no access)
    ld_kw = s2_kw0 < p.KW ? s2_kw0 : 0;
    ld_kc = ld_kh < p.KH ? (ld_kh * p.KW + ld_kw) * p.C + ld_c0 : 0;
  } else {
    const int tap = ld_kc / p.C;
    ld_c0 = ld_kc - tap * p.C;
    ld_kh = tap / p.KW;
    ld_kw = tap - ld_kh * p.KW;
  }
  const bool c_ge_bk = p.C >= KSTEP;
  if constexpr (VEC) {
#pragma unroll
    for (int i = 0; i < AR; ++i) a_cur[i] = a_pixel(i, ld_kh, ld_kw);
  }
  auto load_tile = [&](uint4 (&ra_)[AR], uint4 (&rb_)[BR], unsigned& meta_) {
    if constexpr (VEC) {
      const unsigned kbad = (S2 ? ld_kh < p.KH : ld_kc < K) ? 0u : BUF_OOB;      // OR-ing BUF_OOB into an offset < BUF_OOB puts it out of range
      const unsigned cbytes = (unsigned)ld_c0 * (unsigned)sizeof(T), kbytes = (unsigned)ld_kc * (unsigned)sizeof(T);
#pragma unroll
      for (int i = 0; i < AR; ++i) ra_[i] = buf_load16(rs_in, (a_cur[i] | kbad) + cbytes);
      if constexpr (BNA) {
        unsigned mt = (unsigned)ld_c0 | ((ld_kh == (p.KH >> 1) && ld_kw == (p.KW >> 1)) ? 0x10000u : 0u);
#pragma unroll
        for (int i = 0; i < AR; ++i) mt |= ((a_cur[i] | kbad) < BUF_OOB) ? (0x20000u << i) : 0u;
        meta_ = mt;
      }
#pragma unroll
      for (int j = 0; j < BR; ++j) rb_[j] = buf_load16(rs_w, (b_off[j] | kbad) + kbytes);
      ld_kc += KSTEP;
      bool new_tap;
      if constexpr (S2) {     // next 64 channels of this tap, else the next tap of the class's lattice (two columns / rows further)
        ld_c0 += BK;
        new_tap = ld_c0 >= p.C;
        if (new_tap) {
          ld_c0 -= p.C;
          ld_kw += 2;
          const bool wrap2 = ld_kw >= p.KW;
          ld_kw = wrap2 ? s2_kw0 : ld_kw;
          ld_kh += wrap2 ? 2 : 0;
          ld_kc = ld_kh < p.KH ? (ld_kh * p.KW + ld_kw) * p.C + ld_c0 : 0;      // (past the last tap: ld_kh >= KH marks the tile all-zero)
        }
      } else if (c_ge_bk) {          // at most one tap boundary per step: plain selects
        ld_c0 += KSTEP;
        new_tap = ld_c0 >= p.C;
        ld_c0 -= new_tap ? p.C : 0;
        ld_kw += new_tap ? 1 : 0;
        const bool wrap2 = ld_kw == p.KW;
        ld_kw = wrap2 ? 0 : ld_kw;
        ld_kh += wrap2 ? 1 : 0;
      } else {
        const int tap = ld_kc / p.C;
        ld_c0 = ld_kc - tap * p.C;
        ld_kh = tap / p.KW;
        ld_kw = tap - ld_kh * p.KW;
        new_tap = true;
      }
      if (new_tap && !one_tap) {
#pragma unroll
        for (int i = 0; i < AR; ++i) a_cur[i] = a_pixel(i, ld_kh, ld_kw);
      }
    } else {
      uint32_t wa[AR][4], wb[BR][4];
#pragma unroll
      for (int i = 0; i < AR; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) wa[i][q] = 0u;
#pragma unroll
      for (int j = 0; j < BR; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) wb[j][q] = 0u;
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const int kk = ld_kc + e;
        const unsigned kbad = kk < K ? 0u : BUF_OOB;
        const int tap = kk / p.C;
        const int cc = kk - tap * p.C;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
          const uint32_t v = buf_load_elem<T>(rs_in, (a_pixel(i, kh, kw) | kbad) + (unsigned)cc * (unsigned)sizeof(T));
          if constexpr (sizeof(T) == 2) wa[i][e >> 1] |= v << (16 * (e & 1));
          else wa[i][e] = v;
        }
#pragma unroll
        for (int j = 0; j < BR; ++j) {
          const uint32_t v = buf_load_elem<T>(rs_w, (b_off[j] | kbad) + (unsigned)kk * (unsigned)sizeof(T));
          if constexpr (sizeof(T) == 2) wb[j][e >> 1] |= v << (16 * (e & 1));
          else wb[j][e] = v;
        }
      }
#pragma unroll
      for (int i = 0; i < AR; ++i) ra_[i] = make_uint4(wa[i][0], wa[i][1], wa[i][2], wa[i][3]);
#pragma unroll
      for (int j = 0; j < BR; ++j) rb_[j] = make_uint4(wb[j][0], wb[j][1], wb[j][2], wb[j][3]);
      ld_kc += KSTEP;
    }
  };

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, fh = lane >> 5;
  // one k-tile: ring stage d -> LDS buffer `par`, barrier, (refill stage d), MFMAs.  A buffer is rewritten two tiles
  // later, i.e. after the barrier of the tile in between, which every wave reaches only after its reads of this one.
  auto k_tile = [&](uint4 (&ra_)[AR], uint4 (&rb_)[BR], unsigned& meta_, int par, bool refill) {
    unsigned char* sA = smem + par * STAGE_BYTES;
    unsigned char* sB = sA + BM * PITCH;
    if constexpr (BNA) {
      // one 32-bit word of the chunk (2 elements; 1 in fp32) at a time for all rows: the per-channel constants are read from LDS right where they are
      // used and at most two of them are live (the first version read the chunk's 16 constants up front: 15-20 registers spilled at the 128-register cap)
      const unsigned mt = meta_;
      const int c0 = (int)(mt & 0xffffu);
      const float lo = p.bna.relu ? 0.f : -INFINITY;
      const unsigned ctr_bad = (mt & 0x10000u) ? 0u : BUF_OOB;
      const float* tsc = bn_tab + c0;
      const float* tsh = bn_tab + p.C + c0;
      // (member by member: indexing the 32-bit words of a uint4 through a pointer sent the whole register ring to scratch memory)
      // two 16-bit elements per call, six instructions: unpack (2), v_pk_fma_f32, v_cvt_pk_bf16_f32, the ReLU as a packed signed-integer max on the
      // rounded pair (a negative value has its sign bit set; max(x, 0) then rounds to the same bits as rounding max(x, 0.f)), the validity select.
      // The first version spent 56 instructions per 8-element chunk -- half of the 64x64 loop's own arithmetic per k-tile.
      typedef float f32x2_t __attribute__((ext_vector_type(2)));
      typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
      typedef short i16x2_t __attribute__((ext_vector_type(2)));
      const bool do_relu = p.bna.relu != 0;
      auto word2 = [&](uint32_t w, const float2 s2, const float2 h2, bool ok) -> uint32_t {
        uint32_t o = 0u;
        if constexpr (std::is_same<T, bf16_t>::value) {
          const f32x2_t e = {__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
          const f32x2_t r = __builtin_elementwise_fma(e, f32x2_t{s2.x, s2.y}, f32x2_t{h2.x, h2.y});
          const bf16x2_t b = __builtin_convertvector(r, bf16x2_t);
          i16x2_t bi = __builtin_bit_cast(i16x2_t, b);
          if (do_relu) bi = __builtin_elementwise_max(bi, i16x2_t{0, 0});
          o = __builtin_bit_cast(uint32_t, bi);
        } else if constexpr (sizeof(T) == 2) {
          float e0, e1;
          unpack_f16x2(w, e0, e1);
          e0 = fmaxf(fmaf(e0, s2.x, h2.x), lo);
          e1 = fmaxf(fmaf(e1, s2.y, h2.y), lo);
          o = pack_f16x2(e0, e1);
        }
        return ok ? o : 0u;      // a padding tap (or a row past M) is a zero of the NORMALISED map, not relu(shift)
      };
      auto word1 = [&](uint32_t w, float s1, float h1, bool ok) -> uint32_t {      // one fp32 element
        return ok ? __float_as_uint(fmaxf(fmaf(__uint_as_float(w), s1, h1), lo)) : 0u;
      };
      // (through a local copy and ONE whole-vector assignment per row: member-wise updates of the ring's registers sent the ring to scratch memory)
      uint32_t wq[AR][4];
#pragma unroll
      for (int i = 0; i < AR; ++i) { wq[i][0] = ra_[i].x; wq[i][1] = ra_[i].y; wq[i][2] = ra_[i].z; wq[i][3] = ra_[i].w; }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if constexpr (sizeof(T) == 2) {
          const float2 s2 = *reinterpret_cast<const float2*>(tsc + 2 * q), h2 = *reinterpret_cast<const float2*>(tsh + 2 * q);
#pragma unroll
          for (int i = 0; i < AR; ++i) wq[i][q] = word2(wq[i][q], s2, h2, (mt & (0x20000u << i)) != 0u);
        } else {
          const float s1 = tsc[q], h1 = tsh[q];
#pragma unroll
          for (int i = 0; i < AR; ++i) wq[i][q] = word1(wq[i][q], s1, h1, (mt & (0x20000u << i)) != 0u);
        }
      }
#pragma unroll
      for (int i = 0; i < AR; ++i) ra_[i] = make_uint4(wq[i][0], wq[i][1], wq[i][2], wq[i][3]);
      // the first tile column writes the normalised map (each pixel once: the centre tap, whose input pixel IS the output pixel at stride 1); every
      // other store has an out-of-range offset and is dropped by the hardware
#pragma unroll
      for (int i = 0; i < AR; ++i)
        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{ra_[i].x, ra_[i].y, ra_[i].z, ra_[i].w}, rs_aout, (int)((wb_off[i] | ctr_bad) + (unsigned)c0 * (unsigned)sizeof(T)), 0, 0);
    }
#pragma unroll
    for (int i = 0; i < AR; ++i) *reinterpret_cast<uint4*>(sA + (row0 + 32 * i) * PITCH + chunk * 16) = ra_[i];
#pragma unroll
    for (int j = 0; j < BR; ++j) *reinterpret_cast<uint4*>(sB + (row0 + 32 * j) * PITCH + chunk * 16) = rb_[j];
    __syncthreads();
    if (refill) load_tile(ra_, rb_, meta_);
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
  };

#pragma unroll
  for (int d = 0; d < NST; ++d) load_tile(ra[d], rb[d], rmeta[BNA ? d : 0]);
  // (the table is built AFTER the ring's first loads went out: they do not depend on it, and in front of them the preamble's memory round trip
  // -- fp64 sums -> scale / shift -> LDS -> barrier -- was a serial ~1 us at the head of every block)
  if constexpr (BNA) bn_operand_preamble(p.bna, p.C, bn_tab, block_id == 0);
  int kt = 0;
  for (; kt + NST <= nkt; kt += NST) {
#pragma unroll
    for (int d = 0; d < NST; ++d) k_tile(ra[d], rb[d], rmeta[BNA ? d : 0], (kt + d) & 1, true);
  }
  const int rem = nkt - kt;
#pragma unroll
  for (int d = 0; d < NST - 1; ++d)
    if (d < rem) k_tile(ra[d], rb[d], rmeta[BNA ? d : 0], (kt + d) & 1, false);

  // (group / thread indices re-derived from an opaque copy of the thread id: the compiler otherwise keeps the prologue's copies alive across the
  // k loop for the code below, and at the 128-register cap of the 1024-thread block that cost the K-split variants two spilled registers)
  int grp_e = grp;
  if constexpr (G > 1) {
    int tl = (int)threadIdx.x;
    asm volatile("" : "+v"(tl));
    grp_e = tl >> 8;
    const int tid_e = tl & 255;
    // sum the groups' accumulators: groups 1..G-1 park theirs in LDS ([group-1][register][thread], conflict-free),
    // group 0 adds them and alone runs the epilogue
    __syncthreads();
    float* park = reinterpret_cast<float*>(smem_all);
    constexpr int NACC = TM * TN * 16;
    if (grp_e > 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) park[((grp_e - 1) * NACC + (i * TN + j) * 16 + r) * 256 + tid_e] = acc[i][j][r];
    }
    __syncthreads();
    if (grp_e == 0) {
#pragma unroll
      for (int g = 1; g < G; ++g)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += park[((g - 1) * NACC + (i * TN + j) * 16 + r) * 256 + tid_e];
    }
  }

  // ---- epilogue, row-vectorised ----------------------------------------------------------------------------------
  // The C/D layout of the 32x32 MFMA gives a lane one COLUMN (col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)):
  // storing from it means 2-byte accesses in 64-byte runs, 16 store instructions per wave tile, and the same again for
  // a residual or a mask (measured: the stores were 12-30 % of a 64x64-tile kernel, a 22 MB residual read +15 us).
  // So the block's accumulators go through LDS (free after the k loop) into a row-major fp32 tile, and every thread
  // handles 8 consecutive channels of a row: 16-byte loads / stores in runs of BN channels.
  {
    constexpr int CP = BN + 8;                       // fp32 tile pitch: rows 16-B aligned, half-waves on disjoint banks
    constexpr int CGR = BN / 8;                      // 8-channel groups per row
    constexpr int NT = 256 * G;
    constexpr int RP = NT / CGR;                     // rows per pass
    static_assert(2 * BN <= NT, "the statistics reduction assigns one thread per (statistic, column)");
    const bool al16 = (((uintptr_t)p.out) & 15) == 0 && (!p.res || (((uintptr_t)p.res) & 15) == 0) && (!p.mask_y || (((uintptr_t)p.mask_y) & 15) == 0) &&
                      (!p.stat_x || (((uintptr_t)p.stat_x) & 15) == 0);
    const int eo = p.out_f32 ? 4 : EPC;              // elements per 16 bytes of the output
    const bool vec_ok = al16 && p.ldout % eo == 0 && p.out_bs % eo == 0 && (p.OC % 8 == 0) &&
                        (!p.res || (p.ldres % EPC == 0 && p.res_bs % EPC == 0)) && (!p.mask_y || (p.ldy % EPC == 0 && p.y_bs % EPC == 0)) &&
                        (!p.stat_x || (p.ldsx % EPC == 0 && p.sx_bs % EPC == 0));
    if (vec_ok) {
      float* tile = reinterpret_cast<float*>(smem_all);
      __syncthreads();                               // every wave is done with the k-tile buffers / the parked accumulators
      if (grp_e == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              tile[((wr * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * CP + (wc * TN + j) * 32 + frow] = acc[i][j][r];
      }
      __syncthreads();
      int t = (int)threadIdx.x;
      // the thread id "again", opaque to the compiler: it otherwise keeps the prologue's 64-bit row index alive across the whole k loop for this
      // epilogue -- at the 128-register cap of the 64x64 tile that was the one value spilled to scratch memory (a kernel that touches scratch
      // pays for its set-up in every wave launch), and re-deriving it costs three instructions
      asm volatile("" : "+v"(t));
      const int cg = t % CGR, rr = t / CGR;
      const int n0 = bn * BN + cg * 8;
      const bool col_ok = n0 < p.OC;                 // OC % 8 == 0: a group is all in or all out
      __amdgpu_buffer_rsrc_t rs_part;
      if constexpr (XK) {
        // publish this block's partial tile ([BM][BN] fp32, row-major) write-through, take a ticket, and leave unless it is the last one
        const int S = p.xk_S;
        float* part = p.xk_part + (size_t)xk_tile * (size_t)S * (size_t)(BM * BN);
        rs_part = __builtin_amdgcn_make_buffer_rsrc((void*)part, 0, (int)((unsigned)S * (unsigned)(BM * BN) * 4u), 0x00020000);
#pragma unroll 2
        for (int row = rr; row < BM; row += RP) {
          const u32x4_t a = *reinterpret_cast<const u32x4_t*>(tile + row * CP + cg * 8);
          const u32x4_t b = *reinterpret_cast<const u32x4_t*>(tile + row * CP + cg * 8 + 4);
          const int off = ((xk_s * BM + row) * BN + cg * 8) * 4;
          __builtin_amdgcn_raw_buffer_store_b128(a, rs_part, off, 0, 16);            // aux 16 = sc1: write-through, nothing left dirty in this XCD's L2
          __builtin_amdgcn_raw_buffer_store_b128(b, rs_part, off + 16, 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // EVERY storing wave drains its stores before the block's ticket is taken
        __syncthreads();
        volatile unsigned* arrived = reinterpret_cast<volatile unsigned*>(smem_all + BM * CP * 4);      // (past the fp32 tile)
        // the ticket is the release / acquire point of the seam (agent scope): the partial-tile stores above are ordered before it by the memory model,
        // not only by the write-through stores' behaviour (-DEMRT_XK_RELAXED_TICKET: round 5's relaxed ticket, for the A/B)
#ifdef EMRT_XK_RELAXED_TICKET
        if (t == 0) *arrived = __hip_atomic_fetch_add(p.xk_tick + xk_tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        if (t == 0) *arrived = __hip_atomic_fetch_add(p.xk_tick + xk_tile, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
#endif
        __syncthreads();
        if (*arrived != (unsigned)(S - 1)) return;
        if (t == 0) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // this CU's L1 may hold lines of the scratch from an earlier tile / launch
          __hip_atomic_store(p.xk_tick + xk_tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // every copy has arrived: ready for the next launch
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      float bv[8], sv[8], ss[8], sq[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { bv[e] = (p.bias && col_ok) ? p.bias[n0 + e] : 0.f; sv[e] = (p.scale && col_ok) ? p.scale[n0 + e] : 1.f; ss[e] = 0.f; sq[e] = 0.f; }
      const T* resp = (const T*)p.res;
      const T* ymask = (const T*)p.mask_y;
      // (read ONCE, before the row loop: left inside it the seed was a dependent global load per row -- the pointer may alias the stores)
      const unsigned long long drop_sd = DROP ? p.drop_seed[0] : 0ull;
      const uint32_t drop_thr = DROP ? (uint32_t)(p.drop_p * 65536.f) : 0u;
      const float drop_ks = DROP ? 1.f / (1.f - p.drop_p) : 1.f;
#pragma unroll 2
      for (int row = rr; row < BM; row += RP) {
        const unsigned m = (unsigned)bm * (unsigned)BM + (unsigned)row;
        if ((long long)m >= M || !col_ok) continue;
        int e_nb, e_pix;
        if constexpr (S2) {
          const unsigned q = m - (unsigned)s2_cls * s2_mq, hw2 = (unsigned)(OHW >> 2), ow2 = (unsigned)(p.OW >> 1);
          e_nb = (int)(q / hw2);
          const unsigned r2 = q - (unsigned)e_nb * hw2;
          const unsigned oh2 = r2 / ow2;
          e_pix = (int)((2u * oh2 + (unsigned)(s2_cls >> 1)) * (unsigned)p.OW + 2u * (r2 - oh2 * ow2) + (unsigned)(s2_cls & 1));
        } else {
          e_nb = (int)(m / (unsigned)OHW);
          e_pix = (int)(m - (unsigned)e_nb * (unsigned)OHW);
        }
        float v[8];
        if constexpr (XK) {      // the S partials in s order, this block's own included (read back: the sum does not depend on who came last)
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = 0.f;
          for (int q = 0; q < p.xk_S; ++q) {
            const int off = ((q * BM + row) * BN + cg * 8) * 4;
            const u32x4_t a = __builtin_amdgcn_raw_buffer_load_b128(rs_part, off, 0, 16);
            const u32x4_t b = __builtin_amdgcn_raw_buffer_load_b128(rs_part, off + 16, 0, 16);
            v[0] += __uint_as_float(a.x); v[1] += __uint_as_float(a.y); v[2] += __uint_as_float(a.z); v[3] += __uint_as_float(a.w);
            v[4] += __uint_as_float(b.x); v[5] += __uint_as_float(b.y); v[6] += __uint_as_float(b.z); v[7] += __uint_as_float(b.w);
          }
        } else {
          const float4 a = *reinterpret_cast<const float4*>(tile + row * CP + cg * 8);
          const float4 b = *reinterpret_cast<const float4*>(tile + row * CP + cg * 8 + 4);
          v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], sv[e], bv[e]);
        if (resp) {
          float w8[8];
          Vec8<T>::load(resp + (long long)e_nb * p.res_bs + (long long)e_pix * p.ldres + n0, w8);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += w8[e];
        }
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if constexpr (DROP) {      // (its own instantiation -- igemm_drop_kernel: at the 128-register cap of this tile the extra epilogue code cost the
                                   // K-split and XK variants their last free registers: 8-12 bytes of scratch per lane)
          uint32_t hw[4];
          drop_words8(drop_sd, p.drop_salt, (m * (unsigned)p.OC + (unsigned)n0) >> 3, hw);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = drop_keep8(hw, e, drop_thr) ? v[e] * drop_ks : 0.f;
        }
        float second[8];
        if (ymask) {
          Vec8<T>::load(ymask + (long long)e_nb * p.y_bs + (long long)e_pix * p.ldy + n0, second);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = second[e] > 0.f ? v[e] * p.mask_scale : 0.f;
          if (p.stat_x) Vec8<T>::load((const T*)p.stat_x + (long long)e_nb * p.sx_bs + (long long)e_pix * p.ldsx + n0, second);
        }
        const long long obase = (long long)e_nb * p.out_bs + (long long)e_pix * p.ldout + n0;
        if (p.out_f32) {
          Vec8<float>::store((float*)p.out + obase, v);
        } else {
          Vec8<T>::store((T*)p.out + obase, v);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = to_f32(from_f32<T>(v[e]));      // statistics of what the next kernel will read
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          ss[e] += v[e];
          sq[e] = fmaf(v[e], ymask ? second[e] : v[e], sq[e]);
        }
      }
      if (p.stats) {
        // column sums: [RP][BN] partials per statistic through LDS (the tile is consumed), one atomic per column
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem_all);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          red[rr * BN + cg * 8 + e] = ss[e];
          red[(RP + rr) * BN + cg * 8 + e] = sq[e];
        }
        __syncthreads();
        if (t < 2 * BN) {
          const int which = t / BN, col = t % BN;
          const int n = bn * BN + col;
          float a = 0.f;
          for (int q = 0; q < RP; ++q) a += red[(which * RP + q) * BN + col];
          if (n < p.OC) atomicAdd(p.stats + (long long)(bm & 7) * 2 * p.OC + (long long)which * p.OC + n, (double)a);
        }
      }
      return;
    }
  }
  if (grp_e > 0) return;
  if constexpr (XK) return;      // (the host only takes XK when the row-vectorised epilogue applies: igemm_xk_ok)

  // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  Rows are visited in
  // increasing order, so (image, pixel) is carried along instead of divided out per row.
  const T* resp = (const T*)p.res;
  const T* ymask = (const T*)p.mask_y;
  float st_s[TN], st_q[TN];
  float bias_v[TN], scale_v[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    st_s[j] = 0.f; st_q[j] = 0.f;
    const int n = bn * BN + (wc * TN + j) * 32 + frow;
    bias_v[j] = (p.bias && n < p.OC) ? p.bias[n] : 0.f;
    scale_v[j] = (p.scale && n < p.OC) ? p.scale[n] : 1.f;
  }
  long long m_cur = (long long)bm * BM + wr * TM * 32 + 4 * fh;
  int e_nb = (int)((unsigned)m_cur / (unsigned)OHW);
  int e_pix = (int)((unsigned)m_cur - (unsigned)e_nb * (unsigned)OHW);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // step from the previous visited row to this one: +1 inside a group of 4, +5 between groups, +5 to the next tile
      const int step = (i == 0 && r == 0) ? 0 : ((r & 3) != 0 ? 1 : 5);
      m_cur += step;
      e_pix += step;
      while (e_pix >= OHW) { e_pix -= OHW; ++e_nb; }
      if (m_cur >= M) continue;
      const long long obase = (long long)e_nb * p.out_bs + (long long)e_pix * p.ldout;
      const long long rbase = (long long)e_nb * p.res_bs + (long long)e_pix * p.ldres;
      const long long ybase = (long long)e_nb * p.y_bs + (long long)e_pix * p.ldy;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = bn * BN + (wc * TN + j) * 32 + frow;
        if (n >= p.OC) continue;
        float v = fmaf(acc[i][j][r], scale_v[j], bias_v[j]);
        if (resp) v += to_f32(resp[rbase + n]);
        if (p.relu) v = fmaxf(v, 0.f);

        float second = 0.f;              // what the second statistic multiplies v with
        if (ymask) {
          second = to_f32(ymask[ybase + n]);
          v = second > 0.f ? v * p.mask_scale : 0.f;
          if (p.stat_x) second = to_f32(((const T*)p.stat_x)[(long long)e_nb * p.sx_bs + (long long)e_pix * p.ldsx + n]);
        }
        if (p.out_f32) ((float*)p.out)[obase + n] = v;
        else {
          const T q = from_f32<T>(v);
          ((T*)p.out)[obase + n] = q;
          v = to_f32(q);                 // statistics of what the next kernel will actually read
        }
        st_s[j] += v;
        st_q[j] = fmaf(v, ymask ? second : v, st_q[j]);
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

template <class T, int TM, int TN, int WR, int WC, int MODE, bool VEC, int NST, int G>
__global__ __launch_bounds__(256 * G, (TM * TN == 1 && G == 1) ? 4 : 1) void igemm_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  igemm_body<T, TM, TN, WR, WC, MODE, VEC, NST, G>(p, (int)blockIdx.x, (int)gridDim.x, smem_all);
}

template <class T, int MODE>
__global__ __launch_bounds__(256, 4) void igemm_xk_kernel(ConvArgs p) {      // 64x64 tile, K cut over xk_S blocks per tile: see igemm_body, XK
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  igemm_body<T, 1, 1, 2, 2, MODE, true, 3, 1, false, true>(p, (int)blockIdx.x, (int)gridDim.x, smem_all);
}

template <class T>
__global__ __launch_bounds__(256, 4) void igemm_drop_kernel(ConvArgs p) {      // forward 64x64 tile whose epilogue draws the dropout mask: see ConvArgs::drop_seed
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  igemm_body<T, 1, 1, 2, 2, 0, true, 3, 1, false, false, true>(p, (int)blockIdx.x, (int)gridDim.x, smem_all);
}

template <class T>
__global__ __launch_bounds__(256, 4) void igemm_s2_kernel(ConvArgs p) {      // data gradient of a stride-2 convolution: see igemm_body, S2
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  igemm_body<T, 1, 1, 2, 2, 1, true, 3, 1, true>(p, (int)blockIdx.x, (int)gridDim.x, smem_all);
}

// forward 64x64 tiles whose A operand is the RAW map of a conv -> BatchNorm (-> ReLU) chain: see ConvArgs::bna.  G wave groups as igemm_kernel.
template <class T, int G>
__global__ __launch_bounds__(256 * G, G == 1 ? 4 : 1) void igemm_bna_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  igemm_body<T, 1, 1, 2, 2, 0, true, (G == 2 ? 6 : 2), G, false, false, false, true>(p, (int)blockIdx.x, (int)gridDim.x, smem_all);
}
template <class T>
__global__ __launch_bounds__(256, 4) void igemm_xk_bna_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  igemm_body<T, 1, 1, 2, 2, 0, true, 2, 1, false, true, false, true>(p, (int)blockIdx.x, (int)gridDim.x, smem_all);
}

#include "igemm8p.hpp"

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
  int dil;              // dilation of the kernel taps
  int tiles_per_split;  // number of BKm pixel tiles each z-slice processes
  float* dbias;         // optional [OC]: += sum_m dy[m][oc] (bias gradient), accumulated by the k-tile-0 blocks from the dy tiles they stream
  int overwrite;        // 1: the caller vouches that dw is all zero AND this launch has ONE pixel slice: the tile is stored, not added with atomics
                        // (fp32 atomics run at 1.3 TB/s against ~6 TB/s for stores: layer3 / layer4's large dW with few pixels is all epilogue)
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

// G = 2: two groups of 4 waves per block take alternate pixel tiles of the block's range (own LDS buffers) and their
// accumulators are summed through LDS before the atomic epilogue: half the reduction slices (fp32 atomic traffic into dW,
// ~14 MB per launch by WRITE_SIZE) for the same number of waves.  Kept for experiments; the dispatcher uses G = 1.
template <class T, bool VEC, int G, int NSTW = 2>
__device__ __forceinline__ void wgrad_body(const WgradArgs& p, const int block_x, const int block_y, const int block_z, unsigned char* smem_all) {
  using Cfg = WgradCfg<T>;
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int CPR = Cfg::CPR, RPP = Cfg::RPP, BKM = Cfg::BKM, PITCH = Cfg::PITCH;
  // pixel tiles in flight in registers (8 x 16 B each per thread).  Two, not three: with three the kernel needs ~320 registers
  // against the 256 of a 2-waves-per-SIMD block and spilled 58-63 of them into the loop (wgrad, pair and group kernels alike)
  constexpr int NST = VEC ? NSTW : 1;
  constexpr int STAGE_BYTES = 2 * BKM * PITCH;   // dy tile [BKM][128 oc] + x tile [BKM][128 k]

  const int grp = G == 1 ? 0 : (int)threadIdx.x >> 8;
  unsigned char* smem = smem_all + grp * 2 * STAGE_BYTES;
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int K = p.KH * p.KW * p.C;
  const int oc0 = block_y * 128, k0 = block_x * 128;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)BUF_RANGE, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)BUF_RANGE, 0x00020000);

  const int col = tid % CPR, prow = tid / CPR;
  // this thread's fixed k-chunk (x operand) and oc-chunk (dy operand); BUF_OOB when the chunk is outside the matrix
  const int kq = k0 + col * EPC;
  const int tap = kq < K ? kq / p.C : 0;
  const int cq = kq - tap * p.C;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int ocp = oc0 + col * EPC;
  const unsigned q_bad = kq < K ? 0u : BUF_OOB;
  const unsigned p_bad = ocp < p.OC ? 0u : BUF_OOB;
  // scalar path: per-element decode of this thread's fixed chunk columns (done once)
  int e_c[EPC], e_kh[EPC], e_kw[EPC];
  unsigned e_qbad[EPC], e_pbad[EPC];
  if constexpr (!VEC) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const int kk = kq + e;
      e_qbad[e] = kk < K ? 0u : BUF_OOB;
      const int tp = kk < K ? kk / p.C : 0;
      e_c[e] = kk - tp * p.C;
      e_kh[e] = tp / p.KW;
      e_kw[e] = tp - e_kh[e] * p.KW;
      e_pbad[e] = ocp + e < p.OC ? 0u : BUF_OOB;
    }
  }

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const long long mt_begin = (long long)block_z * p.tiles_per_split;
  const long long mt_total = ((long long)p.N * p.OH * p.OW + BKM - 1) / BKM;
  long long mt_end = mt_begin + p.tiles_per_split;
  if (mt_end > mt_total) mt_end = mt_total;
  if (mt_begin >= mt_end) return;
  const int ntile = (int)(mt_end - mt_begin);

  // loader cursor: (image, oh, ow) of this thread's 4 pixel rows in the NEXT tile to fetch, advanced by BKM * G pixels per
  // tile with mixed-radix carries, plus the running byte offsets of the dy row and of the x image (no division and no
  // 64-bit arithmetic in the loop: this address arithmetic, not the MFMAs, bounds the loop).  Rows past the last image
  // read out of range (= zeros).
  constexpr unsigned ESZ = (unsigned)sizeof(T);
  int r_nb[4], r_oh[4], r_ow[4];
  unsigned r_dy[4], r_ximg[4];
  {
    const int OHW = p.OH * p.OW;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned m = (unsigned)((mt_begin + grp) * BKM) + (unsigned)(prow + RPP * i);      // < 2^31 (host-checked): 32-bit divisions
      r_nb[i] = (int)(m / (unsigned)OHW);
      const int pix = (int)(m - (unsigned)r_nb[i] * (unsigned)OHW);
      r_oh[i] = (int)((unsigned)pix / (unsigned)p.OW);
      r_ow[i] = pix - r_oh[i] * p.OW;
      r_dy[i] = (unsigned)(((long long)r_nb[i] * p.dy_bs + (long long)pix * p.lddy) * (long long)ESZ);      // (garbage past the last image: masked)
      r_ximg[i] = (unsigned)((long long)r_nb[i] * p.x_bs * (long long)ESZ);
    }
  }
  const int adv_w = (BKM * G) % p.OW, adv_q = (BKM * G) / p.OW;      // a group's consecutive tiles are G tiles apart
  const int adv_h = adv_q % p.OH, adv_n = adv_q / p.OH;
  const unsigned dy_step = (unsigned)(BKM * G) * (unsigned)p.lddy * ESZ;                                    // BKM*G pixels further in a dense image
  const unsigned dy_wrap = (unsigned)((p.dy_bs - (long long)p.OH * p.OW * p.lddy) * (long long)ESZ);        // extra per image boundary crossed
  const unsigned x_bs_b = (unsigned)(p.x_bs * (long long)ESZ);
  const unsigned dy_adv = dy_step + (unsigned)adv_n * dy_wrap, x_adv = (unsigned)adv_n * x_bs_b;
  const unsigned ldx_b = (unsigned)p.ldx * ESZ, cq_b = (unsigned)cq * ESZ, ocp_b = (unsigned)ocp * ESZ;
  const int hi0 = kh * p.dil - p.pad, wi0 = kw * p.dil - p.pad;
  int ld_t = grp;                                                   // tile (relative to mt_begin) fetched next

  uint4 rp[NST][4], rq[NST][4];
  auto load_tile = [&](uint4 (&rp_)[4], uint4 (&rq_)[4]) {
    const unsigned t_bad = (G == 1 || ld_t < ntile) ? 0u : BUF_OOB;  // G > 1: the odd group may run one tile past the range
    ld_t += G;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned m_bad = (r_nb[i] < p.N ? 0u : BUF_OOB) | t_bad;
      if constexpr (VEC) {
        rp_[i] = buf_load16(rs_dy, (r_dy[i] | m_bad | p_bad) + ocp_b);
        // 24-bit multiplies (full rate): coordinates, H*W and the pixel stride in bytes are all < 2^24 (host-checked)
        const int hi = __mul24(r_oh[i], p.stride) + hi0, wi = __mul24(r_ow[i], p.stride) + wi0;
        const bool inb = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        const unsigned off = r_ximg[i] + __umul24((unsigned)(__mul24(hi, p.W) + wi), ldx_b) + cq_b;
        rq_[i] = buf_load16(rs_x, inb ? (off | m_bad | q_bad) : BUF_OOB);
      } else {
        uint32_t wp_[4] = {0u, 0u, 0u, 0u}, wq_[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          const uint32_t vp = buf_load_elem<T>(rs_dy, (r_dy[i] | m_bad | e_pbad[e]) + (unsigned)(ocp + e) * ESZ);
          const int hi = r_oh[i] * p.stride - p.pad + e_kh[e] * p.dil, wi = r_ow[i] * p.stride - p.pad + e_kw[e] * p.dil;
          const bool inb = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
          const unsigned off = r_ximg[i] + (unsigned)((hi * p.W + wi) * p.ldx + e_c[e]) * ESZ;
          const uint32_t vq = buf_load_elem<T>(rs_x, inb ? (off | m_bad | e_qbad[e]) : BUF_OOB);
          if constexpr (sizeof(T) == 2) { wp_[e >> 1] |= vp << (16 * (e & 1)); wq_[e >> 1] |= vq << (16 * (e & 1)); }
          else { wp_[e] = vp; wq_[e] = vq; }
        }
        rp_[i] = make_uint4(wp_[0], wp_[1], wp_[2], wp_[3]);
        rq_[i] = make_uint4(wq_[0], wq_[1], wq_[2], wq_[3]);
      }
      // advance this row by BKM * G pixels
      r_ow[i] += adv_w;
      const int c1 = r_ow[i] >= p.OW ? 1 : 0;
      r_ow[i] -= c1 ? p.OW : 0;
      r_oh[i] += adv_h + c1;
      const int c2 = r_oh[i] >= p.OH ? 1 : 0;
      r_oh[i] -= c2 ? p.OH : 0;
      r_nb[i] += adv_n + c2;                     // image boundaries crossed: adv_n (+1 on a carry)
      r_dy[i] += dy_adv + (c2 ? dy_wrap : 0u);
      r_ximg[i] += x_adv + (c2 ? x_bs_b : 0u);
    }
  };

  const bool do_bias = p.dbias != nullptr && block_x == 0;
  float bsum[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) bsum[e] = 0.f;

  // one pixel tile: ring stage -> LDS buffer `par`, barrier, (refill the stage), MFMAs; see igemm_kernel for the hazards
  auto m_tile = [&](uint4 (&rp_)[4], uint4 (&rq_)[4], int par, bool refill) {
    unsigned char* sP = smem + par * STAGE_BYTES;
    unsigned char* sQ = sP + BKM * PITCH;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<uint4*>(sP + (prow + RPP * i) * PITCH + col * 16) = rp_[i];
      *reinterpret_cast<uint4*>(sQ + (prow + RPP * i) * PITCH + col * 16) = rq_[i];
    }
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t w4[4] = {rp_[i].x, rp_[i].y, rp_[i].z, rp_[i].w};
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          if constexpr (sizeof(T) == 2) bsum[e] += bf16_bits_to_f32((w4[e >> 1] >> (16 * (e & 1))) & 0xffffu);
          else bsum[e] += __uint_as_float(w4[e]);
        }
      }
    }
    __syncthreads();
    if (refill) load_tile(rp_, rq_);
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
  };

#pragma unroll
  for (int d = 0; d < NST; ++d) load_tile(rp[d], rq[d]);
  const int niter = (ntile + G - 1) / G;      // pixel tiles walked by each group
  int t = 0;
  for (; t + NST <= niter; t += NST) {
#pragma unroll
    for (int d = 0; d < NST; ++d) m_tile(rp[d], rq[d], (t + d) & 1, true);
  }
  const int rem = niter - t;
#pragma unroll
  for (int d = 0; d < NST - 1; ++d)
    if (d < rem) m_tile(rp[d], rq[d], (t + d) & 1, false);
  __syncthreads();      // the reductions below reuse the LDS tiles

  if constexpr (G > 1) {      // fold the groups' accumulators ([register][thread] in LDS) into group 0
    float* park = reinterpret_cast<float*>(smem_all);
    if (grp > 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) park[(((grp - 1) * 4 + i * 2 + j) * 16 + r) * 256 + tid] = acc[i][j][r];
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
      for (int g = 1; g < G; ++g)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += park[(((g - 1) * 4 + i * 2 + j) * 16 + r) * 256 + tid];
    }
    __syncthreads();
  }

  if (do_bias) {      // block-level reduction over the (group, row) lanes in LDS, then ONE atomic per output channel
    float* sb = reinterpret_cast<float*>(smem_all);     // [G * RPP][128]
#pragma unroll
    for (int e = 0; e < EPC; ++e) sb[(grp * RPP + prow) * 128 + col * EPC + e] = bsum[e];
    __syncthreads();
    if (grp == 0 && tid < 128 && oc0 + tid < p.OC) {
      float tsum = 0.f;
      for (int r = 0; r < G * RPP; ++r) tsum += sb[r * 128 + tid];
      atomicAdd(p.dbias + oc0 + tid, tsum);
    }
  }
  if (grp > 0) return;
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
        if (k < K) {
          if (p.overwrite) p.dw[(long long)oc * K + k] = acc[i][j][r];
          else atomicAdd(p.dw + (long long)oc * K + k, acc[i][j][r]);
        }
      }
    }
}

// One-dimensional grid of 8 * ceil(tx * ty * S / 8) blocks.  Consecutive workgroup ids go round-robin to the 8 XCDs (one L2 each) and the
// tx * ty output tiles of ONE pixel slice stream the same dy / x rows: the work items (slice-major) are cut into 8 contiguous ranges, XCD
// id % 8 takes range id % 8 (wgrad8p.hpp has the measurement: 7x less fabric traffic on UpHead conv_2).  xcd = 0: launch order (A/B knob).
template <class T, bool VEC, int G, int NSTW = 2>
__global__ __launch_bounds__(256 * G, 2) void wgrad_kernel(WgradArgs p, int tx, int ty, int S, int xcd) {   // 2 waves per SIMD (<= 256 registers)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  const int nb = tx * ty * S;
  int w = (int)blockIdx.x;
  if (xcd) {
    const int k = w & 7, idx = w >> 3;
    const int w0 = (int)(((long long)k * nb) >> 3), w1 = (int)(((long long)(k + 1) * nb) >> 3);
    w = w0 + idx;
    if (w >= w1) return;
  } else if (w >= nb) return;
  // (integer divisions run on the vector ALU: the wave-uniform results go back to scalar registers)
  const int bz = __builtin_amdgcn_readfirstlane(w / (tx * ty)), t = __builtin_amdgcn_readfirstlane(w - bz * (tx * ty));
  const int by = __builtin_amdgcn_readfirstlane(t / tx);
  wgrad_body<T, VEC, G, NSTW>(p, __builtin_amdgcn_readfirstlane(t - by * tx), by, bz, smem_all);
}

// Backward pair: ONE launch runs the data-gradient tiles (blocks [0, n_dgrad)) and the weight-gradient tiles (the rest)
// of a layer.  For the ~100 small layers of the step either kernel alone leaves most CUs idle and spends about half of
// its duration in launch / ramp / tail; side by side the two grids fill the machine and share one launch.  Both bodies
// run 256-thread blocks; the block takes the larger LDS / register budget of the two (wgrad's), which is why large
// grids, where dgrad wants its 4 blocks per CU, are still launched separately.
template <class T, int TM, int TN, int WR, int WC, int NST, int NSTW = 2>
__global__ __launch_bounds__(256, 2) void bwd_pair_kernel(ConvArgs pd, WgradArgs pw, int n_dgrad, int wtx, int wty) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  if ((int)blockIdx.x < n_dgrad) {
    igemm_body<T, TM, TN, WR, WC, 1, true, NST, 1>(pd, (int)blockIdx.x, n_dgrad, smem_all);
  } else {
    int r = (int)blockIdx.x - n_dgrad;
    const int bx = r % wtx;
    r /= wtx;
    wgrad_body<T, true, 1, NSTW>(pw, bx, r % wty, r / wty, smem_all);
  }
}

// ------------------------------------------------------------------------------------------------
// host dispatch
// ------------------------------------------------------------------------------------------------
template <class T, int TM, int TN, int WR, int WC, int MODE, bool VEC, int G, int NST>
static int launch_igemm_nst(const ConvArgs& a, hipStream_t st) {
  constexpr int BM = WR * TM * 32, BN = WC * TN * 32;
  const long long M = (long long)a.N * a.OH * a.OW;
  const long long grid = ((M + BM - 1) / BM) * ((a.OC + BN - 1) / BN);
  size_t lds = (size_t)G * 2 * (BM + BN) * 144;
  if (G > 1 && lds < (size_t)(G - 1) * TM * TN * 16 * 256 * 4) lds = (size_t)(G - 1) * TM * TN * 16 * 256 * 4;
  auto kern = igemm_kernel<T, TM, TN, WR, WC, MODE, VEC, NST, G>;
  static bool attr_done = false;      // one flag per instantiation
  if (!attr_done && lds > 65536) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail("emrt_conv2d", "cannot raise the dynamic LDS limit");
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256 * G), lds, st, a);
  return check_launch("emrt_conv2d");
}

template <class T, int TM, int TN, int WR, int WC, int MODE, bool VEC, int G = 1>
static int launch_igemm(const ConvArgs& a, hipStream_t st) {
  constexpr int BM = WR * TM * 32, BN = WC * TN * 32;
  // ring depth: ~96 VGPRs of loads in flight per thread whatever the tile (16 B x (BM + BN) / 32 per stage)
  // (a 1024-thread block, G = 4, has 128 registers per thread: 3 stages fit without spills)
  // (the 64x64 tile at 128 registers: 3 stages; with 4 it spilled 2-4 registers into the loop: 705 -> 710 tiles/s)
  constexpr int NST = VEC ? (G >= 4 ? 3 : (BM + BN) / 32 <= 4 ? (G == 1 ? 3 : 6) : (BM + BN) / 32 <= 6 ? 4 : 3) : 2;
  if constexpr (VEC && G == 1 && BM + BN == 128) {
    if (g_tune.igemm64_nst == 4) return launch_igemm_nst<T, TM, TN, WR, WC, MODE, VEC, G, 4>(a, st);     // A/B knob: the four-stage ring
  }
  return launch_igemm_nst<T, TM, TN, WR, WC, MODE, VEC, G, NST>(a, st);
}

// stride-2 data gradient through the parity-class kernel: whole classes per 64-row tile, 64-channel k-tiles, the row-vectorised epilogue
template <class T>
static bool igemm_s2_ok(const ConvArgs& a) {
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int BK = 8 * EPC;
  const long long M = (long long)a.N * a.OH * a.OW;
  if (a.KH * a.KW == 1 && g_tune.no_s2_dgrad != -1) return false;      // 1x1: three classes in four skip their loop, but the launch lasts as long as the
                                                                       // fourth class's blocks (measured: 20.0 vs 19.8 us, 23.5 vs 22.3); -1 = tests take it anyway
  if (g_tune.no_s2_dgrad == 1 || a.stride != 2 || a.dil != 1 || a.C % BK != 0 || (a.OH & 1) || (a.OW & 1) || (M % 4) || ((M / 4) % 64) || a.OC <= 32 || a.OC % 8) return false;
  if (a.out_f32 || a.bias || a.scale) return false;
  auto al = [](const void* q) { return ((uintptr_t)q) % 16 == 0; };
  if (!al(a.out) || (a.res && !al(a.res)) || (a.mask_y && !al(a.mask_y)) || (a.stat_x && !al(a.stat_x))) return false;
  if (a.ldout % EPC || a.out_bs % EPC || (a.res && (a.ldres % EPC || a.res_bs % EPC)) || (a.mask_y && (a.ldy % EPC || a.y_bs % EPC)) ||
      (a.stat_x && (a.ldsx % EPC || a.sx_bs % EPC))) return false;
  return true;
}

// Cross-block K split (igemm_body XK): S copies of the 64x64 tile grid.  Needs the registered scratch on THIS stream (partials + counters), the
// row-vectorised epilogue (16-byte aligned rows of every epilogue operand) and a 2-byte element type.  Returns S (0 = not applicable).
template <class T>
static int igemm_xk_copies(const ConvArgs& a, hipStream_t st, int want) {
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int BK = 8 * EPC;
  if (sizeof(T) != 2 || g_tune.xk < 0 || !g_scratch.xk_part || !g_scratch.tick || g_scratch.stream != (void*)st) return 0;
  if (a.OC <= 32 || a.OC % 8 || a.out_f32) return 0;
  auto al = [](const void* q) { return ((uintptr_t)q) % 16 == 0; };
  if (!al(a.out) || (a.res && !al(a.res)) || (a.mask_y && !al(a.mask_y)) || (a.stat_x && !al(a.stat_x))) return 0;
  if (a.ldout % EPC || a.out_bs % EPC || (a.res && (a.ldres % EPC || a.res_bs % EPC)) || (a.mask_y && (a.ldy % EPC || a.y_bs % EPC)) ||
      (a.stat_x && (a.ldsx % EPC || a.sx_bs % EPC))) return 0;
  const long long M = (long long)a.N * a.OH * a.OW;
  const long long nb = ((M + 63) / 64) * ((a.OC + 63) / 64);
  const int nkt = (a.KH * a.KW * a.C + BK - 1) / BK;
  int S = want;
  if (S <= 0) {
    // Measured on MI355X (tools/bench_conv.py xk, profiles/r5_xk_sweep.txt; us, old choice -> S = 2 / 4 / 8): the seam -- write-through drain, ticket,
    // acquire, S partial reads -- costs ~7 us, more than priced, so the split only pays where a block's operand stream is LONG (a CU pulls
    // ~55 GB/s whatever runs on it) and the grid leaves most CUs idle:
    //   8x8x512 -> 512 3x3 (64 tiles, 72 k-tiles)      22.4 -> 20.3 / 17.3 / 20.3     16x16x512 -> 512 3x3 s2 (64, 72)     22.7 -> 20.4 / 17.3 / 20.5
    //   16x16x1024 -> 256 3x3 (128 tiles, 144 k-tiles)  38.0 -> 33.7 / 28.5 / 35.9     8x8x256 -> 256 3x3 (32, 36)          15.4 -> 13.9 / 12.8 / 15.0
    //   16x16x256 -> 256 3x3 (128 tiles, 36 k-tiles)    16.3 -> 16.5 / 19.1 / 28.6     8x8x2048 -> 512 1x1 (64, 32)         12.9 -> 13.6 / 13.6 / 18.7
    //   batch 16: 8x8x512 3x3 (128 tiles, 72)           23.4 -> 22.3 / 22.0 / 28.9     16x16x256 3x3 (256 tiles, 36)        16.3 -> 20.5 / 26.2 / 53.7
    // i.e. four copies, for <= 64 tiles with >= 36 k-tiles or <= 128 tiles with >= 128 k-tiles; everything else keeps the in-block split.
    if (!((nb <= 64 && nkt >= 36) || (nb <= 128 && nkt >= 128))) return 0;
    S = 4;
  }
  if (S > nkt) S = nkt;
  if (S >= 2) {      // every copy owns at least one k-tile: copy s walks [s * per, (s + 1) * per), so S = ceil(nkt / per) (9 k-tiles on 8 copies: per = 2 -> 5 copies)
    const int per = (nkt + S - 1) / S;
    S = (nkt + per - 1) / per;
  }
  if (S < 2 || nb > SCRATCH_TICKS || (size_t)nb * (size_t)S * 64 * 64 * 4 > SCRATCH_XK_BYTES) return 0;
  return S;
}

template <class T, int MODE>
static int launch_igemm_xk(const ConvArgs& a0, hipStream_t st, int S) {
  ConvArgs a = a0;
  const long long M = (long long)a.N * a.OH * a.OW;
  const long long nb = ((M + 63) / 64) * ((a.OC + 63) / 64);
  a.xk_S = S; a.xk_part = g_scratch.xk_part; a.xk_tick = g_scratch.tick;
  hipLaunchKernelGGL((igemm_xk_kernel<T, MODE>), dim3((unsigned)(nb * S)), dim3(256), (size_t)2 * 128 * 144, st, a);
  return check_launch("emrt_conv2d");
}

template <class T, int MODE, bool VEC>
static int conv_pick_tile(const ConvArgs& a, hipStream_t st) {
  const long long M = (long long)a.N * a.OH * a.OW;
  auto blocks = [&](int bmv, int bnv) { return ((M + bmv - 1) / bmv) * ((a.OC + bnv - 1) / bnv); };
  if (a.drop_seed) {      // emrt_conv2d_drop: the 64x64 tile with the dropout epilogue (the FFN's first linear: short K, thousands of tiles)
    if constexpr (MODE == 0 && VEC && sizeof(T) != 0) {
      constexpr int EPC = 16 / (int)sizeof(T);
      const bool vec_out = ((uintptr_t)a.out) % 16 == 0 && a.ldout % EPC == 0 && a.out_bs % EPC == 0 && a.OC % 8 == 0 && !a.out_f32 && !a.res && !a.mask_y;
      if (vec_out) {
        hipLaunchKernelGGL((igemm_drop_kernel<T>), dim3((unsigned)blocks(64, 64)), dim3(256), (size_t)2 * 128 * 144, st, a);
        return check_launch("emrt_conv2d_drop");
      }
    }
    return fail("emrt_conv2d_drop", "needs 16-byte aligned rows of C, OC multiples of 8 elements (the vector path)");
  }
  if constexpr (MODE == 1 && VEC) {
    if (!g_tune.conv_tile && igemm_s2_ok<T>(a)) {
      // measured (tools/bench_conv.py s2): the three 3x3 stride-2 data gradients of the ResNet-50 step 25.3 / 24.3 / 27.3 -> 13.8 / 13.0 / 14.5 us
      hipLaunchKernelGGL((igemm_s2_kernel<T>), dim3((unsigned)blocks(64, 64)), dim3(256), (size_t)2 * 128 * 144, st, a);
      return check_launch("emrt_conv2d");
    }
  }
  if (g_tune.conv_tile) {                               // developer knob for tools/bench_conv.py; 0 in production
    switch (g_tune.conv_tile) {
      case 1: return launch_igemm<T, 1, 1, 2, 2, MODE, VEC>(a, st);
      case 2: return launch_igemm<T, 2, 1, 2, 2, MODE, VEC>(a, st);
      case 3: return launch_igemm<T, 2, 2, 2, 2, MODE, VEC>(a, st);
      case 4: return launch_igemm<T, 2, 1, 4, 1, MODE, VEC>(a, st);
      case 5: if constexpr (VEC) return launch_igemm<T, 1, 1, 2, 2, MODE, VEC, 2>(a, st); break;
      case 6: if constexpr (VEC) return launch_igemm<T, 1, 1, 2, 2, MODE, VEC, 4>(a, st); break;
      case 7: if constexpr (VEC && sizeof(T) == 2) { if (igemm8p_ok<T>(a)) return launch_igemm8p<T, MODE>(a, st); } break;
      case 8: if constexpr (VEC && sizeof(T) == 2) return launch_igemm<T, 2, 2, 2, 2, MODE, VEC, 2>(a, st); break;      // 128x128, K split over two wave groups
      default: break;
    }
  }
  if (a.OC <= 32) return launch_igemm<T, 2, 1, 4, 1, MODE, VEC>(a, st);
  if constexpr (VEC && sizeof(T) == 2) {
    if (!g_tune.conv_tile) {
      const int S = igemm_xk_copies<T>(a, st, g_tune.xk > 0 ? g_tune.xk : 0);
      if (S >= 2) return launch_igemm_xk<T, MODE>(a, st, S);
    }
  }
  if constexpr (VEC && sizeof(T) == 2) {
    // the 256 x 256 LDS-DMA kernel (igemm8p.hpp) wins once its grid covers most of the 256 CUs (one 128 KiB block per CU), or about half
    // of them with a long k loop; measured with tools/bench_conv.py big: UpHead conv_2 214 -> 149 us, cls_psp.0 dgrad 154 -> 110 us at
    // batch 8; 256-block grids 98 -> 72 us; 128-block grids tie at 36 k-tiles and win 276 -> 250 us at 216; 64-block grids lose
    const long long nb256 = blocks(256, 256);
    const int nkt64 = a.KH * a.KW * a.C / 64;
    const int minb = g_tune.igemm8p_min_blocks;
    if (minb > 0 && nkt64 >= 16 && (nb256 >= minb || (nb256 >= (minb * 3) / 5 && nkt64 >= 144)) && igemm8p_ok<T>(a)) return launch_igemm8p<T, MODE>(a, st);
    // (Round 6 tried the FFN's 256 <-> 1024 linears over 10 752 rows here as well -- a SHORT k loop with a wide output: 168 tiles of 256x256 pull a quarter
    // of the 64x64 grid's 172 MB of operands.  Alone the kernels win (tools/bench_conv.py mid: forward 21.4 -> 17.5 us, data gradient 20.4 -> 16.9 us);
    // inside the captured step, with linear2's mask epilogue reading the 22 MB activation, the data gradient took 29.8 us against 27.5 us on the 64x64 tile
    // and the forward 22.4 against 24.2 (profiles/r6a_timeline_cfg2.txt): +1 us per layer in all, not kept.)
  }
  // measured on MI355X (tools/bench_conv.py): 128x128 tiles win only when the k loop is long enough to amortise their
  // prologue / epilogue (>= 16 k-tiles) and there is at least one block per CU; everything else is fastest on 64x64
  constexpr int BK = 8 * (16 / (int)sizeof(T));
  const int nkt = (a.KH * a.KW * a.C + BK - 1) / BK;
  if constexpr (VEC && sizeof(T) == 2) {
    // exactly one 128x128 block per CU and a long k loop: a second wave group walking the other half of the k-tiles (147 KB of LDS) keeps
    // the MFMA pipe fed where 4 waves per CU cannot (measured: the decoder's 32x32x1536 -> 512 3x3 forward 158 -> 138 us, 32x32x512 -> 256 data
    // gradient 35.8 -> 32.2; with two rounds of blocks -- 512 at 64x64x256 -> 256 -- the plain tile wins 51 vs 63)
    if (a.OC > 64 && nkt >= 32 && blocks(128, 128) == 256 && !g_tune.no_ksplit128) return launch_igemm<T, 2, 2, 2, 2, MODE, VEC, 2>(a, st);
  }
  if (a.OC > 64 && nkt >= 16 && blocks(128, 128) >= 256) return launch_igemm<T, 2, 2, 2, 2, MODE, VEC>(a, st);
  if constexpr (VEC) {
    // at most one 64x64 block per CU and a long k loop: split K inside the block (see igemm_kernel; thresholds measured)
    const long long nb = blocks(64, 64);
    if (nb <= 128 && nkt >= 32) return launch_igemm<T, 1, 1, 2, 2, MODE, VEC, 4>(a, st);
    if (nb <= 256 && nkt >= 16) return launch_igemm<T, 1, 1, 2, 2, MODE, VEC, 2>(a, st);
  }
  return launch_igemm<T, 1, 1, 2, 2, MODE, VEC>(a, st);
}

template <class T, int MODE>
static int conv_dispatch(const ConvArgs& a, hipStream_t st) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const bool vec = (a.C % EPC == 0) && (a.ldin % EPC == 0) && (a.in_bs % EPC == 0) &&
                   (((uintptr_t)a.in) % 16 == 0) && (((uintptr_t)a.w) % 16 == 0);
  return vec ? conv_pick_tile<T, MODE, true>(a, st) : conv_pick_tile<T, MODE, false>(a, st);
}

static int conv2d_impl(const void* in, const void* w_packed, void* out, const float* bias, const void* residual,
                       int N, int H, int W, int C, int ldin, long long in_bs,
                       int OH, int OW, int OC, int ldout, long long out_bs,
                       int ldres, long long res_bs,
                       int KH, int KW, int stride, int pad,
                       int mode, int relu, int out_f32, double* bn_stats, const void* mask_y, int ldy, long long y_bs,
                       int dilation, const float* out_scale, float drop_p, const unsigned long long* drop_seed, unsigned drop_salt, int dtype, void* stream) {
  EMRT_REQUIRE(in && w_packed && out, "null pointer");
  EMRT_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && OH > 0 && OW > 0 && OC > 0, "bad dims");
  EMRT_REQUIRE(KH > 0 && KW > 0 && stride > 0 && pad >= 0 && dilation >= 1, "bad kernel geometry");
  EMRT_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (fwd) or 1 (dgrad)");
  EMRT_REQUIRE((long long)N * OH * OW + 512 < (1ll << 31) && (long long)N * H * W + 512 < (1ll << 31), "more than 2^31 pixels (32-bit pixel arithmetic)");
  EMRT_REQUIRE_FWD_DTYPE(dtype);
  EMRT_REQUIRE(dtype != EMRT_F16 || mode == 0, "fp16 (dtype 2) is inference-only: forward convolution (mode 0)");
  if (mode == 0) {
    EMRT_REQUIRE(OH == (H + 2 * pad - dilation * (KH - 1) - 1) / stride + 1 && OW == (W + 2 * pad - dilation * (KW - 1) - 1) / stride + 1, "fwd: output size mismatch");
  } else {
    EMRT_REQUIRE(H == (OH + 2 * pad - dilation * (KH - 1) - 1) / stride + 1 && W == (OW + 2 * pad - dilation * (KW - 1) - 1) / stride + 1, "dgrad: size mismatch");
  }
  {
    const long long esz = dtype == EMRT_F32 ? 4 : 2;
    const long long in_ext = ((long long)(N - 1) * in_bs + ((long long)H * W - 1) * ldin + C) * esz;
    const long long w_ext = (long long)OC * KH * KW * C * esz;
    EMRT_REQUIRE(in_bs >= 0 && in_ext < (1ll << 31) && w_ext < (1ll << 31), "operand spans 2 GiB or more (32-bit buffer offsets)");
  }
  ConvArgs a;
  a.in = in; a.w = w_packed; a.out = out; a.bias = bias; a.scale = out_scale; a.res = residual;
  a.N = N; a.H = H; a.W = W; a.C = C; a.ldin = ldin; a.in_bs = in_bs;
  a.OH = OH; a.OW = OW; a.OC = OC; a.ldout = ldout; a.out_bs = out_bs;
  a.ldres = ldres; a.res_bs = res_bs;
  a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.dil = dilation; a.relu = relu; a.out_f32 = out_f32; a.cmajor = g_tune.igemm8p_cmajor; a.stats = bn_stats;
  a.mask_y = mask_y; a.ldy = ldy; a.y_bs = y_bs; a.mask_scale = 1.f; a.stat_x = nullptr; a.ldsx = 0; a.sx_bs = 0;
  a.xk_S = 0; a.xk_part = nullptr; a.xk_tick = nullptr; a.drop_seed = nullptr; a.drop_salt = 0; a.drop_p = 0.f;
  if (drop_p > 0.f) { a.drop_seed = drop_seed; a.drop_salt = drop_salt; a.drop_p = drop_p; }
  EMRT_REQUIRE(!mask_y || !out_f32, "the ReLU mask needs an output in the compute dtype");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EMRT_F32) return mode == 0 ? conv_dispatch<float, 0>(a, st) : conv_dispatch<float, 1>(a, st);
  if (dtype == EMRT_F16) return conv_dispatch<f16_t, 0>(a, st);
  return mode == 0 ? conv_dispatch<bf16_t, 0>(a, st) : conv_dispatch<bf16_t, 1>(a, st);
}

extern "C" int emrt_conv2d(const void* in, const void* w_packed, void* out, const float* bias, const void* residual,
                           int N, int H, int W, int C, int ldin, long long in_bs,
                           int OH, int OW, int OC, int ldout, long long out_bs,
                           int ldres, long long res_bs,
                           int KH, int KW, int stride, int pad,
                           int mode, int relu, int out_f32, double* bn_stats, const void* mask_y, int ldy, long long y_bs,
                           int dilation, const float* out_scale, int dtype, void* stream) {
  return conv2d_impl(in, w_packed, out, bias, residual, N, H, W, C, ldin, in_bs, OH, OW, OC, ldout, out_bs, ldres, res_bs, KH, KW, stride, pad,
                     mode, relu, out_f32, bn_stats, mask_y, ldy, y_bs, dilation, out_scale, 0.f, nullptr, 0u, dtype, stream);
}

// out = dropout_p(relu(linear(in) + bias)): the first half of the transformer FFN (linear1 -> ReLU -> Dropout, transformer_encoder_decoder.py:
// 118-121,157-161,259-262) in ONE launch -- the mask is drawn in the GEMM epilogue, the dropped activation is the only tensor written (the
// separate path wrote relu(linear1), read it back and wrote the dropped copy: a 10 us launch and 44 MB per encoder layer at batch 8).
// 1x1 / stride 1 / no residual; 0 < p < 1; OC % 8 == 0; training dtypes.  The backward needs no mask tensor and no seed: the consumer's data
// gradient masks with (out > 0) and scales by 1 / (1 - p) (emrt_conv2d_bwd: mask_y = out, mask_scale).
extern "C" int emrt_conv2d_drop(const void* in, const void* w_packed, void* out, const float* bias, int M, int C, int ldin, int OC, int ldout,
                                float p, const unsigned long long* seed, unsigned salt, int dtype, void* stream) {
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  EMRT_REQUIRE(seed && p > 0.f && p < 1.f, "needs a device seed and 0 < p < 1");
  EMRT_REQUIRE(M > 0 && C > 0 && OC > 0 && OC % 8 == 0 && (long long)M * OC < (1ll << 32), "bad dims (OC a multiple of 8, fewer than 2^32 outputs)");
  return conv2d_impl(in, w_packed, out, bias, nullptr, 1, 1, M, C, ldin, (long long)M * ldin, 1, M, OC, ldout, (long long)M * ldout, 0, 0, 1, 1, 1, 0,
                     0, 1, 0, nullptr, nullptr, 0, 0, 1, nullptr, p, seed, salt, dtype, stream);
}

// ---- forward convolution whose input BatchNorm (+ ReLU) is applied by the operand loads (ABI 8) -------------------------------------------------
// Which kernel would run this layer with the BatchNorm on its A operand: 0 = none (the dispatcher would take a tile the transform is not built
// into -- 256x32, 128x128, the 256x256 LDS-DMA kernel -- or the geometry is outside it), 1 / 2 / 4 = the 64x64 tile with that many wave groups,
// 100 + S = the cross-block K split with S copies.  Mirrors conv_pick_tile's order of decisions for a forward vector-path layer.
template <class T>
static int bna_choice(const ConvArgs& a, hipStream_t st) {
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int BK = 8 * EPC;
  const bool vec = (a.C % EPC == 0) && (a.ldin % EPC == 0) && (a.in_bs % EPC == 0) && (((uintptr_t)a.in) % 16 == 0) && (((uintptr_t)a.w) % 16 == 0);
  if (!vec || a.C % BK != 0 || a.C > 1024 || a.OC <= 32) return 0;
  if (a.stride != 1 || a.KH != a.KW || (a.KH != 1 && a.KH != 3) || a.pad != a.dil * (a.KH >> 1) || a.OH != a.H || a.OW != a.W) return 0;
  if (((uintptr_t)a.a_out) % 16 != 0) return 0;
  // Measured on MI355X (tools/r6/bench_bna.py, profiles/r6_bna_microbench.txt; batch 8, bf16, us: emrt_bn_apply + emrt_conv2d -> this kernel):
  //   1x1  64x64x64->256   20.0 -> 18.0    32x32x128->512   15.0 -> 12.0    16x16x256->1024  12.3 -> 11.0    8x8x512->2048    14.5 -> 13.2
  //   3x3  64x64x64->64    17.0 -> 16.5    32x32x128->128   16.2 -> 15.0    16x16x256->256   19.9 -> 21.2    8x8x512->512     25.8 -> 28.2
  //        32x32x256->256  30.2 -> 37.1    32x32x512->256   56.3 -> 72.4
  // The transform runs once per loaded chunk, i.e. once per N-tile and nine times per pixel of a 3x3 layer: it pays where the k loop is short (every
  // 1x1; 3x3 up to 128 channels = 18 k-tiles) and loses where the loop is long -- there the separate launch stays (knob no_bna = -1: tests take them all)
  if (a.KH == 3 && a.C > 128 && g_tune.no_bna != -1) return 0;
  const long long M = (long long)a.N * a.OH * a.OW;
  auto blocks = [&](int bmv, int bnv) { return ((M + bmv - 1) / bmv) * ((a.OC + bnv - 1) / bnv); };
  if (g_tune.conv_tile) return g_tune.conv_tile == 1 ? 1 : g_tune.conv_tile == 5 ? 2 : g_tune.conv_tile == 6 ? 4 : 0;
  if constexpr (sizeof(T) == 2) {
    const int S = igemm_xk_copies<T>(a, st, g_tune.xk > 0 ? g_tune.xk : 0);
    if (S >= 2) return 100 + S;
    const long long nb256 = blocks(256, 256);
    const int nkt64 = a.KH * a.KW * a.C / 64;
    const int minb = g_tune.igemm8p_min_blocks;
    if (minb > 0 && nkt64 >= 16 && (nb256 >= minb || (nb256 >= (minb * 3) / 5 && nkt64 >= 144)) && igemm8p_ok<T>(a)) return 0;
  }
  const int nkt = (a.KH * a.KW * a.C + BK - 1) / BK;
  if (a.OC > 64 && nkt >= 16 && blocks(128, 128) >= 256) return 0;
  const long long nb = blocks(64, 64);
  if (nb <= 128 && nkt >= 32) return 4;
  if (nb <= 256 && nkt >= 16) return 2;
  return 1;
}

template <class T>
static int launch_bna(const ConvArgs& a0, hipStream_t st, int choice) {
  ConvArgs a = a0;
  const long long M = (long long)a.N * a.OH * a.OW;
  const long long nb = ((M + 63) / 64) * ((a.OC + 63) / 64);
  const size_t tab = (size_t)2 * a.C * sizeof(float);
  if (choice >= 100) {
    a.xk_S = choice - 100; a.xk_part = g_scratch.xk_part; a.xk_tick = g_scratch.tick;
    hipLaunchKernelGGL((igemm_xk_bna_kernel<T>), dim3((unsigned)(nb * a.xk_S)), dim3(256), (size_t)2 * 128 * 144 + tab, st, a);
    return check_launch("emrt_conv2d_bna");
  }
  const int G = choice;
  size_t lds = (size_t)G * 2 * 128 * 144 + tab;
  if (G > 1 && lds < (size_t)(G - 1) * 16 * 256 * 4) lds = (size_t)(G - 1) * 16 * 256 * 4;
  static bool attr_done = false;      // one flag per element type
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_bna_kernel<T, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_bna_kernel<T, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return fail("emrt_conv2d_bna", "cannot raise the dynamic LDS limit");
    attr_done = true;
  }
  if (G == 1) hipLaunchKernelGGL((igemm_bna_kernel<T, 1>), dim3((unsigned)nb), dim3(256), lds, st, a);
  else if (G == 2) hipLaunchKernelGGL((igemm_bna_kernel<T, 2>), dim3((unsigned)nb), dim3(512), lds, st, a);
  else hipLaunchKernelGGL((igemm_bna_kernel<T, 4>), dim3((unsigned)nb), dim3(1024), lds, st, a);
  return check_launch("emrt_conv2d_bna");
}

static int bna_fill(ConvArgs& a, const void* in, const void* w_packed, void* out, const float* bias, const void* residual, int N, int H, int W, int C, int ldin,
                    long long in_bs, int OH, int OW, int OC, int ldout, long long out_bs, int ldres, long long res_bs, int KH, int KW, int stride, int pad,
                    int relu, int out_f32, double* bn_stats, int dilation, const double* sums, double count, float eps, float momentum, float* mean,
                    float* invstd, float* run_mean, float* run_var, const float* gamma, const float* beta, int in_relu, void* a_out) {
  a.in = in; a.w = w_packed; a.out = out; a.bias = bias; a.scale = nullptr; a.res = residual;
  a.N = N; a.H = H; a.W = W; a.C = C; a.ldin = ldin; a.in_bs = in_bs; a.OH = OH; a.OW = OW; a.OC = OC; a.ldout = ldout; a.out_bs = out_bs;
  a.ldres = ldres; a.res_bs = res_bs; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.dil = dilation; a.relu = relu; a.out_f32 = out_f32;
  a.cmajor = 0; a.stats = bn_stats; a.mask_y = nullptr; a.ldy = 0; a.y_bs = 0; a.mask_scale = 1.f; a.stat_x = nullptr; a.ldsx = 0; a.sx_bs = 0;
  a.xk_S = 0; a.xk_part = nullptr; a.xk_tick = nullptr; a.drop_seed = nullptr; a.drop_salt = 0; a.drop_p = 0.f;
  a.bna.sums = sums; a.bna.inv_count = count > 0.0 ? 1.0 / count : 0.0; a.bna.eps = eps; a.bna.momentum = momentum; a.bna.mean = mean; a.bna.invstd = invstd;
  a.bna.run_mean = run_mean; a.bna.run_var = run_var; a.bna.gamma = gamma; a.bna.beta = beta; a.bna.relu = in_relu;
  a.a_out = a_out;
  return 0;
}

// 1 when emrt_conv2d_bna would run these arguments (same argument list), 0 when the caller has to apply the BatchNorm with its own launch
// (emrt_bn_apply) and call emrt_conv2d -- the layer's tile has no operand transform, or the geometry is outside it (not 1x1 / 3x3 "same" stride 1,
// C not a multiple of 64 (32 in fp32) or > 1024, OC <= 32, unaligned rows).  No launch, no error state.
extern "C" int emrt_conv2d_bna_supported(const void* in, const void* w_packed, void* out, const float* bias, const void* residual, int N, int H, int W, int C,
                                         int ldin, long long in_bs, int OH, int OW, int OC, int ldout, long long out_bs, int ldres, long long res_bs, int KH,
                                         int KW, int stride, int pad, int relu, int out_f32, double* bn_stats, int dilation, const double* sums, double count,
                                         float eps, float momentum, float* mean, float* invstd, float* run_mean, float* run_var, const float* gamma,
                                         const float* beta, int in_relu, void* a_out, int dtype, void* stream) {
  if (!(dtype == EMRT_F32 || dtype == EMRT_BF16) || !in || !w_packed || !out || !sums || !mean || !invstd || !gamma || !beta || !a_out || count <= 0.0) return 0;
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || OC <= 0 || dilation < 1 || g_tune.no_bna == 1) return 0;
  const long long esz = dtype == EMRT_F32 ? 4 : 2;
  if ((long long)N * H * W * C * esz >= (1ll << 31) || (long long)N * H * W + 512 >= (1ll << 31)) return 0;
  if (((long long)(N - 1) * in_bs + ((long long)H * W - 1) * ldin + C) * esz >= (1ll << 31) || (long long)OC * KH * KW * C * esz >= (1ll << 31)) return 0;
  ConvArgs a;
  bna_fill(a, in, w_packed, out, bias, residual, N, H, W, C, ldin, in_bs, OH, OW, OC, ldout, out_bs, ldres, res_bs, KH, KW, stride, pad, relu, out_f32, bn_stats,
           dilation, sums, count, eps, momentum, mean, invstd, run_mean, run_var, gamma, beta, in_relu, a_out);
  hipStream_t st = (hipStream_t)stream;
  return (dtype == EMRT_F32 ? bna_choice<float>(a, st) : bna_choice<bf16_t>(a, st)) != 0 ? 1 : 0;
}

// out = conv([relu](BatchNorm_train(in))) with the BatchNorm applied by the convolution's own operand loads and the normalised map written to a_out
// (dense [N][H][W][C]) on the way: replaces emrt_bn_apply + emrt_conv2d for the BatchNorm -> ReLU -> conv chains of the backbone
// (paddle_vision_resnet.py:129-149: bn1 -> relu -> conv2, bn2 -> relu -> conv3), of Conv2dBlock and cls_psp (paddle_EMRT.py:16-23,201-209).  The
// convolution's arguments are emrt_conv2d's (forward: mode 0, no mask, no folded scale), the BatchNorm's are emrt_bn_apply's (sums complete; mean /
// invstd saved and the running statistics updated by one block).  Fails when emrt_conv2d_bna_supported says 0.
extern "C" int emrt_conv2d_bna(const void* in, const void* w_packed, void* out, const float* bias, const void* residual, int N, int H, int W, int C,
                               int ldin, long long in_bs, int OH, int OW, int OC, int ldout, long long out_bs, int ldres, long long res_bs, int KH,
                               int KW, int stride, int pad, int relu, int out_f32, double* bn_stats, int dilation, const double* sums, double count,
                               float eps, float momentum, float* mean, float* invstd, float* run_mean, float* run_var, const float* gamma,
                               const float* beta, int in_relu, void* a_out, int dtype, void* stream) {
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  EMRT_REQUIRE(in && w_packed && out && sums && mean && invstd && gamma && beta && a_out, "null pointer");
  EMRT_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && OH > 0 && OW > 0 && OC > 0 && dilation >= 1 && count > 0.0, "bad dims");
  EMRT_REQUIRE((run_mean != nullptr) == (run_var != nullptr), "running statistics come in pairs");
  EMRT_REQUIRE(a_out != out && a_out != in, "a_out is a buffer of its own");
  EMRT_REQUIRE((long long)N * OH * OW + 512 < (1ll << 31), "more than 2^31 pixels (32-bit pixel arithmetic)");
  {
    const long long esz = dtype == EMRT_F32 ? 4 : 2;
    const long long in_ext = ((long long)(N - 1) * in_bs + ((long long)H * W - 1) * ldin + C) * esz;
    EMRT_REQUIRE(in_bs >= 0 && in_ext < (1ll << 31) && (long long)OC * KH * KW * C * esz < (1ll << 31) && (long long)N * H * W * C * esz < (1ll << 31),
                 "operand spans 2 GiB or more (32-bit buffer offsets)");
  }
  ConvArgs a;
  bna_fill(a, in, w_packed, out, bias, residual, N, H, W, C, ldin, in_bs, OH, OW, OC, ldout, out_bs, ldres, res_bs, KH, KW, stride, pad, relu, out_f32, bn_stats,
           dilation, sums, count, eps, momentum, mean, invstd, run_mean, run_var, gamma, beta, in_relu, a_out);
  hipStream_t st = (hipStream_t)stream;
  const int choice = dtype == EMRT_F32 ? bna_choice<float>(a, st) : bna_choice<bf16_t>(a, st);
  EMRT_REQUIRE(choice != 0, "this layer has no operand-transform kernel (ask emrt_conv2d_bna_supported first)");
  return dtype == EMRT_F32 ? launch_bna<float>(a, st, choice) : launch_bna<bf16_t>(a, st, choice);
}

// vector path eligibility of a weight-gradient problem
template <class T>
static bool wgrad_is_vec(const WgradArgs& a) {
  constexpr int EPC = 16 / (int)sizeof(T);
  return (a.C % EPC == 0) && (a.OC % EPC == 0) && (a.ldx % EPC == 0) && (a.lddy % EPC == 0) && (a.x_bs % EPC == 0) && (a.dy_bs % EPC == 0) &&
         (((uintptr_t)a.x) % 16 == 0) && (((uintptr_t)a.dy) % 16 == 0);
}

// Split of the pixel reduction over blockIdx.z; fills a.tiles_per_split and returns the grid (tx, ty, S).
// Cost model fitted to tools/bench_conv.py on MI355X:
//   t(S) = tiles_per_block * 1 us * max(1, blocks / 1024)  +  S * |dW| / 1.3 TB/s (every slice re-adds dW atomically)
// i.e. a lone 4-wave block needs ~1 us per pixel tile, up to ~4 blocks per CU overlap for free, fp32 atomics run at ~1.3 TB/s.
template <class T>
static void wgrad_plan(WgradArgs& a, int& tx, int& ty, int& S_out) {
  using Cfg = WgradCfg<T>;
  const int K = a.KH * a.KW * a.C;
  const long long M = (long long)a.N * a.OH * a.OW;
  tx = (K + 127) / 128;
  ty = (a.OC + 127) / 128;
  const long long mt_total = (M + Cfg::BKM - 1) / Cfg::BKM;
  const double atom_us = (double)tx * 128.0 * (double)ty * 128.0 * 4.0 / 1.3e6;
  long long S = 1;
  double best = 1e30;
  static const int cand[] = {1, 2, 4, 8, 16, 32, 64, 128};
  for (int ci = 0; ci < (int)(sizeof(cand) / sizeof(cand[0])); ++ci) {
    const long long sc = cand[ci];
    if (sc > mt_total) break;
    if (sc > 1 && (long long)tx * ty >= 256) break;      // already one block per CU and a large dW: slices only add atomics
    const double per = (double)((mt_total + sc - 1) / sc);
    const double nblk = (double)tx * ty * sc;
    const double t = per * (nblk > 1024.0 ? nblk / 1024.0 : 1.0) + sc * atom_us;
    if (t < best) { best = t; S = sc; }
  }
  if (g_tune.wgrad_split > 0) S = g_tune.wgrad_split < mt_total ? g_tune.wgrad_split : mt_total;   // developer knob (tools/bench_conv.py)
  a.tiles_per_split = (int)((mt_total + S - 1) / S);
  S_out = (int)((mt_total + a.tiles_per_split - 1) / a.tiles_per_split);
}

#include "wgrad8p.hpp"

template <class T>
static int wgrad_dispatch(const WgradArgs& a0, hipStream_t st) {
  using Cfg = WgradCfg<T>;
  WgradArgs a = a0;
  if constexpr (std::is_same<T, bf16_t>::value) {
    int tk, toc, S8, per;
    if (wgrad8p_plan<T>(a, tk, toc, S8, per)) return launch_wgrad8p(a, tk, toc, S8, per, st);
  }
  const bool vec = wgrad_is_vec<T>(a);
  int tx, ty, S;
  wgrad_plan<T>(a, tx, ty, S);
  if (S > 1 || !vec || g_tune.wgrad_no_overwrite) a.overwrite = 0;      // (the scalar-path kernel and multi-slice launches always accumulate)
  // wave groups per block on the vector path (wgrad_kernel).  G = 2 measured SLOWER than two independent G = 1 blocks per
  // CU on every EMRT shape but one (coupled barriers, 160 KB of LDS per block), so 1 it is.
  constexpr int G = 1;
  const int gv = vec ? G : 1;
  const size_t lds = (size_t)gv * 2 * 2 * Cfg::BKM * Cfg::PITCH;      // per group: two stages of (dy tile + x tile)
  static bool attr_done = false;      // one flag per element type
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<T, true, G>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)((size_t)G * 4 * Cfg::BKM * Cfg::PITCH)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<T, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)((size_t)4 * Cfg::BKM * Cfg::PITCH)) != hipSuccess)
      return fail("emrt_conv2d_wgrad", "cannot raise the dynamic LDS limit");
    attr_done = true;
  }
  const int xcd = g_tune.wgrad8p_xcd;
  const unsigned grid1 = 8u * (unsigned)(((long long)tx * ty * S + 7) / 8);
  if (vec && g_tune.wgrad_nst == 3) {       // A/B knob: the spilling three-stage ring
    static bool attr3 = false;
    if (!attr3) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<T, true, G, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)G * 4 * Cfg::BKM * Cfg::PITCH)); attr3 = true; }
    hipLaunchKernelGGL((wgrad_kernel<T, true, G, 3>), dim3(grid1), dim3(256 * G), lds, st, a, tx, ty, S, xcd);
  } else if (vec) hipLaunchKernelGGL((wgrad_kernel<T, true, G>), dim3(grid1), dim3(256 * G), lds, st, a, tx, ty, S, xcd);
  else hipLaunchKernelGGL((wgrad_kernel<T, false, 1>), dim3(grid1), dim3(256), lds, st, a, tx, ty, S, xcd);
  return check_launch("emrt_conv2d_wgrad");
}

extern "C" int emrt_conv2d_wgrad(const void* x, const void* dy, float* dw,
                                 int N, int H, int W, int C, int ldx, long long x_bs,
                                 int OH, int OW, int OC, int lddy, long long dy_bs,
                                 int KH, int KW, int stride, int pad, float* dbias, int dilation, int dtype, void* stream) {
  EMRT_REQUIRE(x && dy && dw, "null pointer");
  EMRT_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && OH > 0 && OW > 0 && OC > 0 && dilation >= 1, "bad dims");
  EMRT_REQUIRE(OH == (H + 2 * pad - dilation * (KH - 1) - 1) / stride + 1 && OW == (W + 2 * pad - dilation * (KW - 1) - 1) / stride + 1, "output size mismatch");
  EMRT_REQUIRE((long long)N * OH * OW + 512 < (1ll << 31) && (long long)N * H * W + 512 < (1ll << 31), "more than 2^31 pixels (32-bit pixel arithmetic)");
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  {
    const long long esz = dtype == EMRT_F32 ? 4 : 2;
    const long long x_ext = ((long long)(N - 1) * x_bs + ((long long)H * W - 1) * ldx + C) * esz;
    const long long dy_ext = ((long long)(N - 1) * dy_bs + ((long long)OH * OW - 1) * lddy + OC) * esz;
    EMRT_REQUIRE(x_bs >= 0 && dy_bs >= 0 && x_ext < (1ll << 31) && dy_ext < (1ll << 31), "operand spans 2 GiB or more (32-bit buffer offsets)");
    EMRT_REQUIRE((long long)H * W < (1 << 24) && (long long)ldx * esz < (1 << 24) && stride < (1 << 12), "map too large for the 24-bit address arithmetic");
  }
  WgradArgs a;
  a.x = x; a.dy = dy; a.dw = dw;
  a.N = N; a.H = H; a.W = W; a.C = C; a.ldx = ldx; a.x_bs = x_bs;
  a.OH = OH; a.OW = OW; a.OC = OC; a.lddy = lddy; a.dy_bs = dy_bs;
  a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.dil = dilation; a.tiles_per_split = 0; a.dbias = dbias; a.overwrite = 0;
  hipStream_t st = (hipStream_t)stream;
  return dtype == EMRT_F32 ? wgrad_dispatch<float>(a, st) : wgrad_dispatch<bf16_t>(a, st);
}

// ------------------------------------------------------------------------------------------------
// Thin backward: 1x1, stride 1, OC <= 8 (the per-pixel classifier, 256 -> num_classes at 128x128).  As GEMMs its dgrad has
// K = OC and its wgrad a OC x C output: MFMA tiles would be > 80 % padding and the two kernels would stream x, y and dx
// separately.  Here ONE pass reads x (which is also the ReLU-mask source y when they coincide) and dy once and writes dx
// once: a thread owns 8 channels of one pixel, its W^T slice and its dW partial stay in registers; the block reduces
// dW / dbias / the BatchNorm sums through LDS and issues one atomic per value.  Same arithmetic order as the igemm
// epilogue: accumulate -> mask -> round -> statistics of the rounded value.
// ------------------------------------------------------------------------------------------------
struct ThinBwdArgs {
  const void* x; const void* dy; const void* wd; void* dx; float* dw; float* dbias; double* stats; const void* mask_y;
  int HW, C, OC, ldx, lddy, lddx, ldy, accumulate, pix_per_block, cg_shift;
  float mask_scale;
  long long x_bs, dy_bs, dx_bs, y_bs, M;
  const float* bn_mean; const float* bn_invstd; const float* bn_gamma; const float* bn_beta;      // BNX: x is the RAW map, the layer's input is relu(BN(x))
};

// MASK: 0 no ReLU mask, 1 the mask source is x itself (y = relu(bn(.)) feeds this conv), 2 a separate tensor.  ACC: dx +=.
// CH channels per thread (4: 128 registers, 4 blocks per CU; 8: 16-byte bf16 accesses, 2 blocks per CU).
template <class T, int CH>
struct VecN;
template <class T>
struct VecN<T, 4> : Vec4<T> {};
template <class T>
struct VecN<T, 8> : Vec8<T> {};

// BNX (with MASK == 1): x is the raw pre-BatchNorm map and the conv's input a = relu(x * scale + shift) is re-derived per element as it is
// loaded (bn_operand.hpp; the forward was emrt_bn_pointwise_fwd, the normalised map was never written): a feeds dW, the mask and the sums.
template <class T, int CH, int MASK, bool ACC, bool BNX = false>
__global__ __launch_bounds__(256, CH == 4 ? 4 : 2) void thin_bwd_kernel(ThinBwdArgs p) {
  constexpr int U = 2;              // pixels in flight per thread
  using V = VecN<T, CH>;
  __shared__ float red[32 * 256];
  const T* x = (const T*)p.x;
  const T* dy = (const T*)p.dy;
  const T* wd = (const T*)p.wd;
  const T* ym = (const T*)p.mask_y;
  T* dx = (T*)p.dx;
  const int tid = threadIdx.x;
  const int CG = 1 << p.cg_shift, ppb = 256 >> p.cg_shift;
  const int cg = tid & (CG - 1), pl = tid >> p.cg_shift;
  const int c0 = (int)blockIdx.y * (CH << p.cg_shift) + cg * CH;     // blockIdx.y: channel chunk
  // the dy row of a pixel (OC <= 8 values) is fetched by the first OC lanes of each lane group working on that pixel
  // (ONE load instruction per wave) and handed round with ds_bpermute: eight same-address scalar loads per pixel
  // were what bound the first version of this kernel (the texture-address path, not HBM)
  const int lane = tid & 63;
  const int grp = CG < 64 ? CG : 64;
  const int gpos = lane & (grp - 1), gbase = lane - gpos;
  float w[8][CH], dwa[8][CH], ss[CH], sq[CH], db[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    db[o] = 0.f;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      w[o][j] = o < p.OC ? to_f32(wd[(c0 + j) * p.OC + o]) : 0.f;
      dwa[o][j] = 0.f;
    }
  }
#pragma unroll
  for (int j = 0; j < CH; ++j) { ss[j] = 0.f; sq[j] = 0.f; }
  float bsc[BNX ? CH : 1], bsh[BNX ? CH : 1];
  if (BNX) {
#pragma unroll
    for (int j = 0; j < CH; ++j) bn_scale_shift(p.bn_mean[c0 + j], p.bn_invstd[c0 + j], p.bn_gamma[c0 + j], p.bn_beta[c0 + j], bsc[j], bsh[j]);
  }
  // 32-bit element offsets throughout (thin_bwd_ok checks that every operand stays below 2^31 elements)
  const int xbs = (int)p.x_bs, dybs = (int)p.dy_bs, dxbs = (int)p.dx_bs, ybs = (int)p.y_bs, M = (int)p.M;
  const int m0 = (int)blockIdx.x * p.pix_per_block;
  const int m1 = m0 + p.pix_per_block < M ? m0 + p.pix_per_block : M;
  // mb is uniform per lane group, so ok[u] is too and the shuffles below stay inside converged groups
  for (int mb = m0 + pl; mb < m1; mb += U * ppb) {
    float xv[U][CH], yv[U][CH], old[U][CH], mydy[U];
    bool ok[U];
    int oo[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {            // all loads of the U pixels first
      const int m = mb + u * ppb;
      ok[u] = m < m1;
      mydy[u] = 0.f;
      if (!ok[u]) continue;
      const int nb = m / p.HW;
      const int pix = m - nb * p.HW;
      oo[u] = nb * dxbs + pix * p.lddx + c0;
      V::load(x + (nb * xbs + pix * p.ldx + c0), xv[u]);
      if (MASK == 2) V::load(ym + (nb * ybs + pix * p.ldy + c0), yv[u]);
      if (ACC) V::load(dx + oo[u], old[u]);
      if (gpos < p.OC) mydy[u] = to_f32(dy[nb * dybs + pix * p.lddy + gpos]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float dyv[8];
#pragma unroll
      for (int o = 0; o < 8; ++o) dyv[o] = __shfl(mydy[u], gbase + o, 64);     // executed by every lane (0 for the dead tail)
      if (!ok[u]) continue;
      if (BNX) {
#pragma unroll
        for (int j = 0; j < CH; ++j) xv[u][j] = fmaxf(fmaf(xv[u][j], bsc[j], bsh[j]), 0.f);
      }
      float v[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        float a = 0.f;
#pragma unroll
        for (int o = 0; o < 8; ++o) a = fmaf(dyv[o], w[o][j], a);
        if (ACC) a += old[u][j];
        float second = 0.f;
        if (MASK) {
          second = MASK == 1 ? xv[u][j] : yv[u][j];
          a = second > 0.f ? a * p.mask_scale : 0.f;
        }
        v[j] = a;
        const float q = to_f32(from_f32<T>(a));
        ss[j] += q;
        sq[j] = fmaf(q, MASK ? second : q, sq[j]);
      }
      V::store(dx + oo[u], v);
#pragma unroll
      for (int o = 0; o < 8; ++o) {
#pragma unroll
        for (int j = 0; j < CH; ++j) dwa[o][j] = fmaf(dyv[o], xv[u][j], dwa[o][j]);
        if (cg == 0 && blockIdx.y == 0) db[o] += dyv[o];
      }
    }
  }
  // dW through red[k][tid] in rounds of 32 values (k = o * CH + j); thread t then owns channel group t % CG and every
  // ppb-th k of the round
#pragma unroll
  for (int rnd = 0; rnd < CH / 4; ++rnd) {
    if (rnd * 32 >= p.OC * CH) break;
    if (rnd) __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; ++k) red[k * 256 + tid] = dwa[(rnd * 32 + k) / CH][(rnd * 32 + k) % CH];
    __syncthreads();
    for (int k = pl; k < 32; k += ppb) {
      const int kk = rnd * 32 + k;
      if (kk >= p.OC * CH) break;
      float a = 0.f;
      for (int q = 0; q < ppb; ++q) a += red[k * 256 + q * CG + cg];
      atomicAdd(p.dw + (kk / CH) * p.C + c0 + (kk % CH), a);
    }
  }
  if (p.stats || p.dbias) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      red[k * 256 + tid] = ss[k];
      red[(CH + k) * 256 + tid] = sq[k];
    }
    if (cg == 0) {      // (zeros from the channel chunks other than the first)
#pragma unroll
      for (int k = 0; k < 8; ++k) red[(2 * CH + k) * 256 + pl] = db[k];
    }
    __syncthreads();
    if (p.stats) {
      double* rep = p.stats + (long long)(blockIdx.x & 7) * 2 * p.C;
      for (int k = pl; k < 2 * CH; k += ppb) {
        float a = 0.f;
        for (int q = 0; q < ppb; ++q) a += red[k * 256 + q * CG + cg];
        atomicAdd(rep + (k / CH) * p.C + c0 + (k % CH), (double)a);
      }
    }
    if (p.dbias && tid < p.OC) {
      float a = 0.f;
      for (int q = 0; q < ppb; ++q) a += red[(2 * CH + tid) * 256 + q];
      atomicAdd(p.dbias + tid, a);
    }
  }
}

template <class T>
static bool thin_bwd_ok(const ConvArgs& d, const WgradArgs& w) {
  constexpr int EPC = 4;                  // a thread moves 4 channels: 16 B (f32) / 8 B (bf16)
  const int C = w.C;
  if (w.KH != 1 || w.KW != 1 || w.stride != 1 || w.pad != 0 || w.OC > 8) return false;
  if (d.stat_x || (d.res && d.mask_y) || (d.res && d.res != d.out)) return false;      // the generic path handles those
  if (C < 32 || C > 1024 || (C & (C - 1))) return false;
  const long long px = (long long)w.H * w.W - 1, LIM = 1ll << 31;
  if ((long long)w.N * w.H * w.W >= LIM || (w.N - 1) * w.x_bs + px * w.ldx + C >= LIM || (w.N - 1) * d.out_bs + px * d.ldout + C >= LIM ||
      (w.N - 1) * w.dy_bs + px * w.lddy + w.OC >= LIM || (d.mask_y && (w.N - 1) * d.y_bs + px * d.ldy + C >= LIM)) return false;
  auto al = [](const void* q) { return ((uintptr_t)q) % 16 == 0; };
  if (!al(w.x) || !al(d.out) || (d.mask_y && !al(d.mask_y))) return false;
  if (w.ldx % EPC || w.x_bs % EPC || d.ldout % EPC || d.out_bs % EPC) return false;
  if (d.mask_y && (d.ldy % EPC || d.y_bs % EPC)) return false;
  return true;
}

template <class T, int CH>
static int thin_bwd_launch_ch(const ConvArgs& d, const WgradArgs& w, hipStream_t st, const float* const* xbn = nullptr) {
  ThinBwdArgs a;
  a.x = w.x; a.dy = w.dy; a.wd = d.w; a.dx = d.out; a.dw = w.dw; a.dbias = w.dbias; a.stats = d.stats; a.mask_y = d.mask_y;
  a.HW = w.H * w.W; a.C = w.C; a.OC = w.OC; a.ldx = w.ldx; a.lddy = w.lddy; a.lddx = d.ldout; a.ldy = d.ldy;
  a.accumulate = d.res != nullptr;
  a.mask_scale = d.mask_scale;
  a.x_bs = w.x_bs; a.dy_bs = w.dy_bs; a.dx_bs = d.out_bs; a.y_bs = d.y_bs;
  a.M = (long long)w.N * w.H * w.W;
  // Grid = pixel chunks x channel chunks.  Every block ends with OC * (its channels) fp32 atomics into dW, and those are
  // what bounds the kernel (measured on 8x128x128x256 -> 6: one channel chunk with 1024 pixel chunks 155 us, 256 chunks 90 us, without the atomics
  // 47-60 us; 4 channel chunks x 128 pixel chunks 46 us; the MFMA pair before it 214 us): few, long pixel
  // chunks keep the atomic count down, the channel chunks (whole 128-byte lines per pixel) restore the block count.
  int cblk = g_tune.thin_cblk;                     // developer knobs (defaults 64 / 128)
  if (cblk < 8 * CH) cblk = 8 * CH;               // a lane group must hold the OC <= 8 lanes that fetch the dy row
  if (cblk > w.C) cblk = w.C;
  if (cblk > 256 * CH) cblk = 256 * CH;
  int sh = 0;
  while ((CH << sh) < cblk) ++sh;
  a.cg_shift = sh;
  const int cchunks = w.C / (CH << sh);
  const int ppb = 256 >> sh;
  const long long want_chunks = g_tune.thin_blocks > 0 ? g_tune.thin_blocks : 128;
  long long per = (a.M + want_chunks - 1) / want_chunks;
  per = (per + 2 * ppb - 1) / (2 * ppb) * (2 * ppb);
  if (per < 8 * ppb) per = 8 * ppb;
  a.pix_per_block = (int)per;
  const long long blocks = (a.M + per - 1) / per;
  const bool same = d.mask_y && d.mask_y == w.x && d.ldy == w.ldx && d.y_bs == w.x_bs;
  const int mask = !d.mask_y ? 0 : (same ? 1 : 2);
  const dim3 grid((unsigned)blocks, (unsigned)cchunks), block(256);
  a.bn_mean = a.bn_invstd = a.bn_gamma = a.bn_beta = nullptr;
  if (xbn) {
    if (mask != 1 || a.accumulate) return fail("emrt_bn_pointwise_bwd", "internal: the BatchNorm-operand form is the mask-from-input, overwrite form");
    a.bn_mean = xbn[0]; a.bn_invstd = xbn[1]; a.bn_gamma = xbn[2]; a.bn_beta = xbn[3];
    hipLaunchKernelGGL((thin_bwd_kernel<T, CH, 1, false, true>), grid, block, 0, st, a);
    return check_launch("emrt_bn_pointwise_bwd");
  }
  if (a.accumulate) {      // (the C-ABI excludes accumulate together with a mask)
    hipLaunchKernelGGL((thin_bwd_kernel<T, CH, 0, true>), grid, block, 0, st, a);
  } else if (mask == 0) {
    hipLaunchKernelGGL((thin_bwd_kernel<T, CH, 0, false>), grid, block, 0, st, a);
  } else if (mask == 1) {
    hipLaunchKernelGGL((thin_bwd_kernel<T, CH, 1, false>), grid, block, 0, st, a);
  } else {
    hipLaunchKernelGGL((thin_bwd_kernel<T, CH, 2, false>), grid, block, 0, st, a);
  }
  return check_launch("emrt_conv2d_bwd");
}

template <class T>
static int thin_bwd_launch(const ConvArgs& d, const WgradArgs& w, hipStream_t st, const float* const* xbn = nullptr) {
  const int want = g_tune.thin_ch;            // developer knob (default 8)
  constexpr int EPC = 16 / (int)sizeof(T);
  const bool can8 = w.C >= 64 && w.ldx % 8 == 0 && w.x_bs % 8 == 0 && d.ldout % 8 == 0 && d.out_bs % 8 == 0 &&
                    (!d.mask_y || (d.ldy % 8 == 0 && d.y_bs % 8 == 0)) && EPC <= 8;
  if (want == 8 && can8) return thin_bwd_launch_ch<T, 8>(d, w, st, xbn);
  return thin_bwd_launch_ch<T, 4>(d, w, st, xbn);
}

// ------------------------------------------------------------------------------------------------
// Thin forward with a BatchNorm operand: out[p][o] = bias[o] + sum_c relu(BN(x[p][c])) * w[o][c], OC <= 8 (the classifier behind
// conv -> SyncBatchNorm -> ReLU, paddle_EMRT.py:176-179).  x is the RAW conv output (67 MB at batch 8 / 256x256 tiles); the separate
// path reads it, writes the normalised map, and reads that again through a 256x32-tile GEMM (35 + 20 us): here it is read once.
// VALU kernel (0.4 GFLOP): a pixel is spread over CG = C / 8 lanes (8 channels = one 16-byte load each), a thread works on U = 4 pixels
// at a time, and the 4 x 8 partial dot products are summed across the CG lanes by a reduce-scatter butterfly (each step hands half of the
// values to the partner lane: 16 + 8 + 4 + 2 + 1 exchanges instead of 32 x 5), after which lane i of the group owns value i.
// ------------------------------------------------------------------------------------------------
struct ThinFwdArgs {
  const void* x; const void* w; const float* bias; void* out;
  int HW, C, OC, ldx, ldo;
  long long x_bs, o_bs, M;
};

// eight elements as they come from memory (16 B of bf16 / 32 B of fp32), unpacked where they are used: the prefetched pixels stay packed
template <class T>
struct Raw8;
template <>
struct Raw8<bf16_t> {
  uint4 q;
  __device__ __forceinline__ void load(const bf16_t* p) { q = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void unpack(float (&o)[8]) const { Vec8<bf16_t>::unpack(q, o); }
};
template <>
struct Raw8<float> {
  float4 a, b;
  __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4); }
  __device__ __forceinline__ void unpack(float (&o)[8]) const { o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w; }
};

template <class T, int CGS>
__global__ __launch_bounds__(256, 2) void thin_fwd_bn_kernel(ThinFwdArgs p, BnOperand bn) {
  constexpr int CH = 8, U = 4;
  extern __shared__ float bn_lds[];          // [2][C]: scale, shift
  const T* x = (const T*)p.x;
  const T* wp = (const T*)p.w;
  T* out = (T*)p.out;
  const int tid = threadIdx.x;
  constexpr int CG = 1 << CGS, ppb = 256 >> CGS;          // 8 <= CG <= 32: a lane group lies inside one wave
  const int cg = tid & (CG - 1), pl = tid >> CGS;
  const int c0 = cg * CH;
  // 32-bit element offsets throughout (the host checks that every operand stays below 2^31 elements)
  const int M = (int)p.M, HW = p.HW, xbs = (int)p.x_bs, obs = (int)p.o_bs;
  const int step = (int)gridDim.x * ppb * U;
  int mb = (int)blockIdx.x * ppb * U + pl;
  auto xoff = [&](int m) {
    const int mc = m < M ? m : 0;          // (a dead tail pixel reads pixel 0 and is zeroed below)
    const int nb = mc / HW;
    return nb * xbs + (mc - nb * HW) * p.ldx + c0;
  };
  // first group's loads before the per-channel preamble
  Raw8<T> xv[U], xn[U];
#pragma unroll
  for (int u = 0; u < U; ++u) xv[u].load(x + xoff(mb + u * ppb));
  bn_operand_preamble(bn, p.C, bn_lds, blockIdx.x == 0);
  float sc[CH], sh[CH], w[8][CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) { sc[j] = bn_lds[c0 + j]; sh[j] = bn_lds[p.C + c0 + j]; }
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    float wv[CH];
    Vec8<T>::load(wp + ((o < p.OC ? o : 0) * p.C + c0), wv);
#pragma unroll
    for (int j = 0; j < CH; ++j) w[o][j] = o < p.OC ? wv[j] : 0.f;
  }
  const float lo = bn.relu ? 0.f : -INFINITY;
  while (mb - pl < M) {          // (uniform per block: every lane takes part in the exchanges)
    const int mnext = mb + step;
#pragma unroll
    for (int u = 0; u < U; ++u) xn[u].load(x + xoff(mnext + u * ppb));
    float part[U * 8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = mb + u * ppb < M;
      float a[CH];
      xv[u].unpack(a);
#pragma unroll
      for (int j = 0; j < CH; ++j) a[j] = ok ? fmaxf(fmaf(a[j], sc[j], sh[j]), lo) : 0.f;
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < CH; ++j) t = fmaf(a[j], w[o][j], t);
        part[u * 8 + o] = t;
      }
    }
    // reduce-scatter over the CG lanes of the group: at the step with distance d a lane keeps the half of its values selected by its
    // bit d and adds the partner's copy of that half; n values -> n / 2.  CG = 32: 32 -> 16 -> 8 -> 4 -> 2 -> 1.
#pragma unroll
    for (int k = 0; k < CGS; ++k) {
      const int n = (U * 8 / 2) >> k, d = (CG / 2) >> k;      // compile-time after unrolling
      const bool hi = (cg & d) != 0;
#pragma unroll
      for (int i = 0; i < n; ++i) {
        // (the empty asm keeps the two elements values: without it the optimizer folds the selects into ONE element read with a
        // lane-dependent index, i.e. a 32-way compare / select chain per value -- 1920 of them, 147 us instead of 20)
        float lo_v = part[i], hi_v = part[n + i];
        asm volatile("" : "+v"(lo_v), "+v"(hi_v));
        const float keep = hi ? hi_v : lo_v;
        const float send = hi ? lo_v : hi_v;
        part[i] = keep + __shfl_xor(send, d, 64);
      }
    }
    // bit d of cg selected the upper half (offset + n) at the step with distance d, so the lane now owns indices [cg * nleft, + nleft)
    constexpr int nleft = (U * 8) / CG;          // 1, 2 or 4 values per lane
#pragma unroll
    for (int i = 0; i < nleft; ++i) {
      const int idx = cg * nleft + i, u = idx >> 3, o = idx & 7;
      const int m = mb + u * ppb;
      if (o < p.OC && m < M) {
        const int nb = m / HW;
        out[nb * obs + (m - nb * HW) * p.ldo + o] = from_f32<T>(part[i] + (p.bias ? p.bias[o] : 0.f));
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) xv[u] = xn[u];
    mb = mnext;
  }
}

// ------------------------------------------------------------------------------------------------
// Backward of one conv / linear layer in one call: dx (= or +=, any NHWC strides: a token slab or channel slice of a
// gradient buffer is written in place) = dgrad(dy, W^T) [masked by y > 0 with BatchNorm sums, see ConvArgs::mask_y]
// and dW += wgrad(x, dy) [+ dbias].  Small layers go out as ONE launch (bwd_pair_kernel); large
// grids and non-vector shapes as the two separate kernels.
// ------------------------------------------------------------------------------------------------
template <class T>
static int conv_bwd_dispatch(const ConvArgs& d, const WgradArgs& w0, hipStream_t st) {
  using Cfg = WgradCfg<T>;
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int BK = 8 * EPC;
  WgradArgs w = w0;
  const bool vec_d = (d.C % EPC == 0) && (d.ldin % EPC == 0) && (d.in_bs % EPC == 0) && (((uintptr_t)d.in) % 16 == 0) && (((uintptr_t)d.w) % 16 == 0);
  const bool vec_w = wgrad_is_vec<T>(w);
  const long long Md = (long long)d.N * d.OH * d.OW;
  const long long nd = ((Md + 63) / 64) * ((d.OC + 63) / 64);                 // 64x64 dgrad tiles
  const int nkt = (d.KH * d.KW * d.C + BK - 1) / BK;
  const bool big_tile = d.OC > 64 && nkt >= 16 && ((Md + 127) / 128) * ((d.OC + 127) / 128) >= 256;   // conv_pick_tile would take 128x128
  if (!w0.dw) return conv_dispatch<T, 1>(d, st);      // data gradient only: the caller batches the weight gradient (emrt_conv2d_wgrad_group)
  if (thin_bwd_ok<T>(d, w0) && !g_tune.no_thin_bwd) return thin_bwd_launch<T>(d, w0, st);
  int tx = 0, ty = 0, S = 0;
  if (vec_w) wgrad_plan<T>(w, tx, ty, S);
  const long long nw = (long long)tx * ty * S;
  // pairing pays while the dgrad grid is small (measured per shape, tools/bench_conv.py bwd: 16x16 / 8x8 layers -30..-40 %,
  // token linears -5..-10 %; from ~1000 dgrad tiles on, dgrad misses its 4 blocks per CU and the pair is 5-20 % slower)
  const long long pair_max = g_tune.pair_max;                // developer knob, default 768; measured: 0 -> 15.03 ms, 384 -> 14.46, 768 -> 14.42, 1536 -> 14.35, 3072+ -> 14.43
  if (vec_d && vec_w && d.OC > 32 && !big_tile && nd <= pair_max && nd + nw <= 4096) {
    auto kern = bwd_pair_kernel<T, 1, 1, 2, 2, 6>;
    auto kern3 = bwd_pair_kernel<T, 1, 1, 2, 2, 6, 3>;
    const size_t lds_d = (size_t)2 * 128 * 144, lds_w = (size_t)4 * Cfg::BKM * Cfg::PITCH;
    const size_t lds = lds_d > lds_w ? lds_d : lds_w;
    static bool attr_done = false;
    if (!attr_done) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(kern3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return fail("emrt_conv2d_bwd", "cannot raise the dynamic LDS limit");
      attr_done = true;
    }
    if (g_tune.wgrad_nst == 3) hipLaunchKernelGGL(kern3, dim3((unsigned)(nd + nw)), dim3(256), lds, st, d, w, (int)nd, tx, ty);
    else hipLaunchKernelGGL(kern, dim3((unsigned)(nd + nw)), dim3(256), lds, st, d, w, (int)nd, tx, ty);
    return check_launch("emrt_conv2d_bwd");
  }
  int rc = wgrad_dispatch<T>(w0, st);
  if (rc) return rc;
  return conv_dispatch<T, 1>(d, st);
}

// ---- the classifier behind conv -> SyncBatchNorm -> ReLU with the BatchNorm applied by its loads (ABI 6) --------------------------
// forward: out[N][HW][OC] = bias + relu(BN_train(x)) . w^T, x the RAW map [N][HW][C] of the producing conv, w the forward-packed [OC][C]
// weight (a 1x1 conv / linear), OC <= 8, C in {64, 128, 256} (C / 8 lanes per pixel inside one wave).  BatchNorm arguments as
// emrt_bn_apply (sums complete; mean / invstd saved; running statistics updated).
extern "C" int emrt_bn_pointwise_fwd(const void* x, int ldx, long long x_bs, const void* w_packed, const float* bias, void* out, int ldo,
                                     long long o_bs, int N, int HW, int C, int OC, const double* sums, double count, float eps,
                                     float momentum, float* mean, float* invstd, float* run_mean, float* run_var, const float* gamma,
                                     const float* beta, int relu, int dtype, void* stream) {
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  EMRT_REQUIRE(x && w_packed && out && sums && mean && invstd && gamma && beta, "null pointer");
  EMRT_REQUIRE(N > 0 && HW > 0 && OC >= 1 && OC <= 8 && (C == 64 || C == 128 || C == 256), "OC <= 8 and C in {64, 128, 256}");
  EMRT_REQUIRE(count > 0.0 && (run_mean != nullptr) == (run_var != nullptr), "bad BatchNorm operand");
  {
    const long long LIM = 1ll << 31, px = (long long)HW - 1;
    EMRT_REQUIRE((long long)N * HW + 8192ll * 128 < LIM && (N - 1) * x_bs + px * ldx + C < LIM && (N - 1) * o_bs + px * ldo + OC < LIM,
                 "operand spans 2^31 elements or more (32-bit offsets)");
  }
  const int esz = dtype == EMRT_F32 ? 4 : 2;
  EMRT_REQUIRE(ldx % 8 == 0 && x_bs % 8 == 0 && ((uintptr_t)x % (8 * esz) == 0) && ((uintptr_t)w_packed % (8 * esz) == 0), "x rows and the weight must be 8-element aligned");
  ThinFwdArgs a;
  a.x = x; a.w = w_packed; a.bias = bias; a.out = out; a.HW = HW; a.C = C; a.OC = OC; a.ldx = ldx; a.ldo = ldo; a.x_bs = x_bs; a.o_bs = o_bs;
  a.M = (long long)N * HW;
  BnOperand b;
  b.sums = sums; b.inv_count = 1.0 / count; b.eps = eps; b.momentum = momentum; b.mean = mean; b.invstd = invstd; b.run_mean = run_mean;
  b.run_var = run_var; b.gamma = gamma; b.beta = beta; b.relu = relu;
  const int cgs = C == 256 ? 5 : (C == 128 ? 4 : 3);
  const long long per_iter = (long long)(256 >> cgs) * 4;          // pixels one block covers per iteration
  long long blocks = (a.M + per_iter - 1) / per_iter;
  // 256 registers: two blocks per CU are resident, so 512 blocks are one round and every block pays the per-channel preamble once
  // (measured on 8 x 128 x 128 x 256 -> 6: 512 blocks 28 us, 1024 31 us, 4096 41 us)
  const long long cap = g_tune.bn_operand_blocks > 0 ? g_tune.bn_operand_blocks : 512;
  if (blocks > cap) blocks = cap;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)2 * C * sizeof(float);
#define EMRT_THIN_FWD(TT)                                                                                                     \
  do {                                                                                                                        \
    if (cgs == 5) hipLaunchKernelGGL((thin_fwd_bn_kernel<TT, 5>), dim3((unsigned)blocks), dim3(256), lds, st, a, b);           \
    else if (cgs == 4) hipLaunchKernelGGL((thin_fwd_bn_kernel<TT, 4>), dim3((unsigned)blocks), dim3(256), lds, st, a, b);      \
    else hipLaunchKernelGGL((thin_fwd_bn_kernel<TT, 3>), dim3((unsigned)blocks), dim3(256), lds, st, a, b);                    \
  } while (0)
  if (dtype == EMRT_F32) EMRT_THIN_FWD(float);
  else EMRT_THIN_FWD(bf16_t);
#undef EMRT_THIN_FWD
  return check_launch("emrt_bn_pointwise_fwd");
}

// backward of the same layer in one pass: da = dy . w masked by a = relu(BN(x)) > 0 (a re-derived from the raw x per element),
// dW += dy^T . a, dbias += sum dy, and the BatchNorm's backward sums (sum da, sum da * a) into the ZEROED fp64 `stats` [8][2C] -- the form
// emrt_bn_bwd_dx takes with beta_y_moments.  w_bwd_packed: the [C][OC] weight.  da: dense [N][HW][C].
extern "C" int emrt_bn_pointwise_bwd(const void* x, int ldx, long long x_bs, const void* dy, int lddy, long long dy_bs, const void* w_bwd_packed,
                                     void* da, int ldda, long long da_bs, float* dw, float* dbias, double* stats, int N, int HW, int C, int OC,
                                     const float* mean, const float* invstd, const float* gamma, const float* beta, int dtype, void* stream) {
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  EMRT_REQUIRE(x && dy && w_bwd_packed && da && dw && mean && invstd && gamma && beta, "null pointer");
  EMRT_REQUIRE(N > 0 && HW > 0 && OC >= 1 && OC <= 8, "OC <= 8");
  ConvArgs d;
  memset(&d, 0, sizeof(d));
  WgradArgs w;
  memset(&w, 0, sizeof(w));
  // the data-gradient view: "input" dy [N][HW][OC], "output" da [N][HW][C]
  d.in = dy; d.w = w_bwd_packed; d.out = da; d.N = N; d.H = 1; d.W = HW; d.C = OC; d.ldin = lddy; d.in_bs = dy_bs;
  d.OH = 1; d.OW = HW; d.OC = C; d.ldout = ldda; d.out_bs = da_bs; d.KH = d.KW = 1; d.stride = 1; d.pad = 0; d.dil = 1;
  d.mask_y = x; d.ldy = ldx; d.y_bs = x_bs; d.mask_scale = 1.f; d.stats = stats;
  w.x = x; w.dy = dy; w.dw = dw; w.dbias = dbias; w.N = N; w.H = 1; w.W = HW; w.C = C; w.ldx = ldx; w.x_bs = x_bs;
  w.OH = 1; w.OW = HW; w.OC = OC; w.lddy = lddy; w.dy_bs = dy_bs; w.KH = w.KW = 1; w.stride = 1; w.pad = 0; w.dil = 1; w.overwrite = 0;
  const float* xbn[4] = {mean, invstd, gamma, beta};
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EMRT_F32) {
    EMRT_REQUIRE(thin_bwd_ok<float>(d, w), "shape outside the thin kernel (C a power of two in 32..1024, 16-byte aligned rows)");
    return thin_bwd_launch<float>(d, w, st, xbn);
  }
  EMRT_REQUIRE(thin_bwd_ok<bf16_t>(d, w), "shape outside the thin kernel (C a power of two in 32..1024, 16-byte aligned rows)");
  return thin_bwd_launch<bf16_t>(d, w, st, xbn);
}


extern "C" int emrt_conv2d_bwd(const void* x, const void* dy, const void* w_bwd_packed, void* dx, int lddx, long long dx_bs,
                               int accumulate, float* dw, float* dbias,
                               int N, int H, int W, int C, int ldx, long long x_bs, int OH, int OW, int OC, int lddy, long long dy_bs,
                               int KH, int KW, int stride, int pad, double* bn_stats, const void* mask_y, int ldy, long long y_bs,
                               float mask_scale, const void* stat_x, int ldsx, long long sx_bs, const void* addend, int ldadd, long long add_bs,
                               int dilation, int dtype, void* stream) {
  EMRT_REQUIRE(x && dy && w_bwd_packed && dx, "null pointer");
  EMRT_REQUIRE(dw || !dbias, "dw == NULL asks for the data gradient only: the bias gradient travels with the deferred weight gradient");
  EMRT_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && OH > 0 && OW > 0 && OC > 0, "bad dims");
  EMRT_REQUIRE(KH > 0 && KW > 0 && stride > 0 && pad >= 0 && dilation >= 1, "bad kernel geometry");
  EMRT_REQUIRE(OH == (H + 2 * pad - dilation * (KH - 1) - 1) / stride + 1 && OW == (W + 2 * pad - dilation * (KW - 1) - 1) / stride + 1, "output size mismatch");
  EMRT_REQUIRE((long long)N * OH * OW + 512 < (1ll << 31) && (long long)N * H * W + 512 < (1ll << 31), "more than 2^31 pixels (32-bit pixel arithmetic)");
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  {
    const long long esz = dtype == EMRT_F32 ? 4 : 2;
    const long long x_ext = ((long long)(N - 1) * x_bs + ((long long)H * W - 1) * ldx + C) * esz;
    const long long dy_ext = ((long long)(N - 1) * dy_bs + ((long long)OH * OW - 1) * lddy + OC) * esz;
    const long long w_ext = (long long)OC * KH * KW * C * esz;
    EMRT_REQUIRE(x_bs >= 0 && dy_bs >= 0 && x_ext < (1ll << 31) && dy_ext < (1ll << 31) && w_ext < (1ll << 31), "operand spans 2 GiB or more (32-bit buffer offsets)");
    EMRT_REQUIRE((long long)H * W < (1 << 24) && (long long)ldx * esz < (1 << 24) && stride < (1 << 12), "map too large for the 24-bit address arithmetic");
  }
  ConvArgs d;       // dgrad: a convolution of dy with the transposed weights, output = dx (dense NHWC)
  d.in = dy; d.w = w_bwd_packed; d.out = dx; d.bias = nullptr; d.scale = nullptr; d.res = nullptr;
  d.N = N; d.H = OH; d.W = OW; d.C = OC; d.ldin = lddy; d.in_bs = dy_bs;
  d.OH = H; d.OW = W; d.OC = C; d.ldout = lddx; d.out_bs = dx_bs;
  d.ldres = 0; d.res_bs = 0;
  if (accumulate) { d.res = dx; d.ldres = lddx; d.res_bs = dx_bs; }      // dx += : every element is read and written by the same thread
  else if (addend) { d.res = addend; d.ldres = ldadd; d.res_bs = add_bs; }      // dx = dgrad + addend (gradient contributions made so far)
  EMRT_REQUIRE(lddx >= C && dx_bs >= 0, "bad dx strides");
  EMRT_REQUIRE(!(accumulate && addend), "accumulate adds into dx itself; addend is a different tensor");
  EMRT_REQUIRE(!stat_x || mask_y, "stat_x replaces the mask tensor in the second statistic: it needs mask_y");
  d.KH = KH; d.KW = KW; d.stride = stride; d.pad = pad; d.dil = dilation; d.relu = 0; d.out_f32 = 0; d.cmajor = g_tune.igemm8p_cmajor; d.stats = bn_stats;
  d.mask_y = mask_y; d.ldy = ldy; d.y_bs = y_bs; d.mask_scale = mask_scale; d.stat_x = stat_x; d.ldsx = ldsx; d.sx_bs = sx_bs;
  d.xk_S = 0; d.xk_part = nullptr; d.xk_tick = nullptr; d.drop_seed = nullptr; d.drop_salt = 0; d.drop_p = 0.f;
  WgradArgs w;
  w.x = x; w.dy = dy; w.dw = dw;
  w.N = N; w.H = H; w.W = W; w.C = C; w.ldx = ldx; w.x_bs = x_bs;
  w.OH = OH; w.OW = OW; w.OC = OC; w.lddy = lddy; w.dy_bs = dy_bs;
  w.KH = KH; w.KW = KW; w.stride = stride; w.pad = pad; w.dil = dilation; w.tiles_per_split = 0; w.dbias = dbias; w.overwrite = 0;
  hipStream_t st = (hipStream_t)stream;
  return dtype == EMRT_F32 ? conv_bwd_dispatch<float>(d, w, st) : conv_bwd_dispatch<bf16_t>(d, w, st);
}

// ------------------------------------------------------------------------------------------------
// Grouped launches: up to 6 independent small problems (the per-level 3x3 convs of an encoder layer, ...) in ONE launch.
// Each problem alone is a 10-30 us, latency-bound launch that fills a fraction of the GPU; their tiles side by side
// share the launch and hide each other's latency.  Plain-C descriptors (include/emrt_hip.h).
// ------------------------------------------------------------------------------------------------
#define EMRT_MAX_GROUP 6
struct EmrtConvDesc {
  const void* in; const void* w_packed; void* out; const float* bias; const void* residual;
  int N, H, W, C, ldin; long long in_bs;
  int OH, OW, OC, ldout; long long out_bs;
  int ldres; long long res_bs;
  int KH, KW, stride, pad, relu;
  double* bn_stats;
  int out_f32;
};
struct EmrtConvBwdDesc {
  const void* x; const void* dy; const void* w_bwd_packed; void* dx; int lddx; long long dx_bs; int accumulate; float* dw; float* dbias;
  int N, H, W, C, ldx; long long x_bs; int OH, OW, OC, lddy; long long dy_bs; int KH, KW, stride, pad;
};
struct ConvGroupArgs { ConvArgs p[EMRT_MAX_GROUP]; int first[EMRT_MAX_GROUP + 1]; };
struct BwdGroupArgs {
  ConvArgs d[EMRT_MAX_GROUP];
  WgradArgs w[EMRT_MAX_GROUP];
  int first[EMRT_MAX_GROUP + 1];      // block range of problem i: [first[i], first[i+1]); inside it nd[i] dgrad tiles, then wgrad
  int nd[EMRT_MAX_GROUP], wtx[EMRT_MAX_GROUP], wty[EMRT_MAX_GROUP];
};

template <class T, int MODE = 0>
__global__ __launch_bounds__(256, 4) void igemm_group_kernel(ConvGroupArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  int i = 0;
#pragma unroll
  for (int k = 1; k < EMRT_MAX_GROUP; ++k) i += (int)blockIdx.x >= g.first[k] ? 1 : 0;
  igemm_body<T, 1, 1, 2, 2, MODE, true, 3, 1>(g.p[i], (int)blockIdx.x - g.first[i], g.first[i + 1] - g.first[i], smem_all);
}

template <class T>
__global__ __launch_bounds__(256, 2) void bwd_group_kernel(BwdGroupArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  int i = 0;
#pragma unroll
  for (int k = 1; k < EMRT_MAX_GROUP; ++k) i += (int)blockIdx.x >= g.first[k] ? 1 : 0;
  const int local = (int)blockIdx.x - g.first[i];
  if (local < g.nd[i]) {
    igemm_body<T, 1, 1, 2, 2, 1, true, 6, 1>(g.d[i], local, g.nd[i], smem_all);
  } else {
    int r = local - g.nd[i];
    const int bx = r % g.wtx[i];
    r /= g.wtx[i];
    wgrad_body<T, true, 1>(g.w[i], bx, r % g.wty[i], r / g.wty[i], smem_all);
  }
}

static void conv_args_from_desc(ConvArgs& a, const EmrtConvDesc& d) {
  a.in = d.in; a.w = d.w_packed; a.out = d.out; a.bias = d.bias; a.scale = nullptr; a.res = d.residual;
  a.N = d.N; a.H = d.H; a.W = d.W; a.C = d.C; a.ldin = d.ldin; a.in_bs = d.in_bs;
  a.OH = d.OH; a.OW = d.OW; a.OC = d.OC; a.ldout = d.ldout; a.out_bs = d.out_bs;
  a.ldres = d.ldres; a.res_bs = d.res_bs;
  a.KH = d.KH; a.KW = d.KW; a.stride = d.stride; a.pad = d.pad; a.dil = 1; a.relu = d.relu; a.out_f32 = d.out_f32 ? 1 : 0; a.cmajor = 0; a.stats = d.bn_stats;
  a.mask_y = nullptr; a.ldy = 0; a.y_bs = 0; a.mask_scale = 1.f; a.stat_x = nullptr; a.ldsx = 0; a.sx_bs = 0;
  a.xk_S = 0; a.xk_part = nullptr; a.xk_tick = nullptr; a.drop_seed = nullptr; a.drop_salt = 0; a.drop_p = 0.f;
}

template <class T>
static bool conv_desc_is_vec(const ConvArgs& a) {
  constexpr int EPC = 16 / (int)sizeof(T);
  return (a.C % EPC == 0) && (a.ldin % EPC == 0) && (a.in_bs % EPC == 0) && (((uintptr_t)a.in) % 16 == 0) && (((uintptr_t)a.w) % 16 == 0);
}

template <class T>
static int conv_group_dispatch(const EmrtConvDesc* descs, int n, hipStream_t st) {
  ConvGroupArgs g;
  bool groupable = true;
  long long total = 0;
  for (int i = 0; i < n; ++i) {
    conv_args_from_desc(g.p[i], descs[i]);
    const long long M = (long long)g.p[i].N * g.p[i].OH * g.p[i].OW;
    g.first[i] = (int)total;
    total += ((M + 63) / 64) * ((g.p[i].OC + 63) / 64);
    groupable = groupable && conv_desc_is_vec<T>(g.p[i]) && g.p[i].OC > 32;
  }
  for (int i = n; i <= EMRT_MAX_GROUP; ++i) g.first[i] = (int)total;
  for (int i = n; i < EMRT_MAX_GROUP; ++i) g.p[i] = g.p[0];
  if (!groupable || total > 4096) {        // not the small vector problems this launch is for: one launch each
    for (int i = 0; i < n; ++i) {
      const int rc = conv_dispatch<T, 0>(g.p[i], st);
      if (rc) return rc;
    }
    return 0;
  }
  const size_t lds = (size_t)2 * 128 * 144;
  hipLaunchKernelGGL((igemm_group_kernel<T>), dim3((unsigned)total), dim3(256), lds, st, g);
  return check_launch("emrt_conv2d_group");
}

extern "C" int emrt_conv2d_group(const EmrtConvDesc* descs, int n, int dtype, void* stream) {
  EMRT_REQUIRE(descs && n >= 1 && n <= EMRT_MAX_GROUP, "1..6 problems");
  EMRT_REQUIRE_FWD_DTYPE(dtype);
  for (int i = 0; i < n; ++i) {
    const EmrtConvDesc& d = descs[i];
    EMRT_REQUIRE(d.in && d.w_packed && d.out, "null pointer");
    EMRT_REQUIRE(d.N > 0 && d.H > 0 && d.W > 0 && d.C > 0 && d.OC > 0 && d.KH > 0 && d.KW > 0 && d.stride > 0 && d.pad >= 0, "bad dims");
    EMRT_REQUIRE(d.OH == (d.H + 2 * d.pad - d.KH) / d.stride + 1 && d.OW == (d.W + 2 * d.pad - d.KW) / d.stride + 1, "output size mismatch");
    EMRT_REQUIRE((long long)d.N * d.OH * d.OW + 512 < (1ll << 31) && (long long)d.N * d.H * d.W + 512 < (1ll << 31), "more than 2^31 pixels");
    const long long esz = dtype == EMRT_F32 ? 4 : 2;
    const long long in_ext = ((long long)(d.N - 1) * d.in_bs + ((long long)d.H * d.W - 1) * d.ldin + d.C) * esz;
    EMRT_REQUIRE(d.in_bs >= 0 && in_ext < (1ll << 31) && (long long)d.OC * d.KH * d.KW * d.C * esz < (1ll << 31), "operand spans 2 GiB or more");
  }
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EMRT_F16) return conv_group_dispatch<f16_t>(descs, n, st);
  return dtype == EMRT_F32 ? conv_group_dispatch<float>(descs, n, st) : conv_group_dispatch<bf16_t>(descs, n, st);
}

// dw == NULL in every descriptor: the data gradients only, as one grouped launch of 64x64 dgrad tiles (the weight gradients are
// batched by the caller: emrt_conv2d_wgrad_group)
template <class T>
static int conv_dgrad_group_dispatch(const EmrtConvBwdDesc* descs, int n, hipStream_t st) {
  ConvGroupArgs g;
  bool groupable = true;
  long long total = 0;
  for (int i = 0; i < n; ++i) {
    const EmrtConvBwdDesc& b = descs[i];
    ConvArgs& d = g.p[i];
    d.in = b.dy; d.w = b.w_bwd_packed; d.out = b.dx; d.bias = nullptr; d.scale = nullptr; d.res = b.accumulate ? b.dx : nullptr;
    d.N = b.N; d.H = b.OH; d.W = b.OW; d.C = b.OC; d.ldin = b.lddy; d.in_bs = b.dy_bs;
    d.OH = b.H; d.OW = b.W; d.OC = b.C; d.ldout = b.lddx; d.out_bs = b.dx_bs;
    d.ldres = b.accumulate ? b.lddx : 0; d.res_bs = b.accumulate ? b.dx_bs : 0;
    d.KH = b.KH; d.KW = b.KW; d.stride = b.stride; d.pad = b.pad; d.dil = 1; d.relu = 0; d.out_f32 = 0; d.cmajor = 0; d.stats = nullptr;
    d.mask_y = nullptr; d.ldy = 0; d.y_bs = 0; d.mask_scale = 1.f; d.stat_x = nullptr; d.ldsx = 0; d.sx_bs = 0;
    d.xk_S = 0; d.xk_part = nullptr; d.xk_tick = nullptr; d.drop_seed = nullptr; d.drop_salt = 0; d.drop_p = 0.f;
    const long long Md = (long long)d.N * d.OH * d.OW;
    g.first[i] = (int)total;
    total += ((Md + 63) / 64) * ((d.OC + 63) / 64);
    groupable = groupable && conv_desc_is_vec<T>(d) && d.OC > 32;
  }
  for (int i = n; i <= EMRT_MAX_GROUP; ++i) g.first[i] = (int)total;
  for (int i = n; i < EMRT_MAX_GROUP; ++i) g.p[i] = g.p[0];
  if (!groupable || total > 4096) {
    for (int i = 0; i < n; ++i) {
      const int rc = conv_dispatch<T, 1>(g.p[i], st);
      if (rc) return rc;
    }
    return 0;
  }
  hipLaunchKernelGGL((igemm_group_kernel<T, 1>), dim3((unsigned)total), dim3(256), (size_t)2 * 128 * 144, st, g);
  return check_launch("emrt_conv2d_bwd_group");
}

template <class T>
static int conv_bwd_group_dispatch(const EmrtConvBwdDesc* descs, int n, hipStream_t st) {
  using Cfg = WgradCfg<T>;
  if (!descs[0].dw) return conv_dgrad_group_dispatch<T>(descs, n, st);
  BwdGroupArgs g;
  bool groupable = true;
  long long total = 0;
  for (int i = 0; i < n; ++i) {
    const EmrtConvBwdDesc& b = descs[i];
    ConvArgs& d = g.d[i];
    d.in = b.dy; d.w = b.w_bwd_packed; d.out = b.dx; d.bias = nullptr; d.scale = nullptr; d.res = b.accumulate ? b.dx : nullptr;
    d.N = b.N; d.H = b.OH; d.W = b.OW; d.C = b.OC; d.ldin = b.lddy; d.in_bs = b.dy_bs;
    d.OH = b.H; d.OW = b.W; d.OC = b.C; d.ldout = b.lddx; d.out_bs = b.dx_bs;
    d.ldres = b.accumulate ? b.lddx : 0; d.res_bs = b.accumulate ? b.dx_bs : 0;
    d.KH = b.KH; d.KW = b.KW; d.stride = b.stride; d.pad = b.pad; d.dil = 1; d.relu = 0; d.out_f32 = 0; d.cmajor = 0; d.stats = nullptr;
    d.mask_y = nullptr; d.ldy = 0; d.y_bs = 0; d.mask_scale = 1.f; d.stat_x = nullptr; d.ldsx = 0; d.sx_bs = 0;
    d.xk_S = 0; d.xk_part = nullptr; d.xk_tick = nullptr; d.drop_seed = nullptr; d.drop_salt = 0; d.drop_p = 0.f;
    WgradArgs& w = g.w[i];
    w.x = b.x; w.dy = b.dy; w.dw = b.dw;
    w.N = b.N; w.H = b.H; w.W = b.W; w.C = b.C; w.ldx = b.ldx; w.x_bs = b.x_bs;
    w.OH = b.OH; w.OW = b.OW; w.OC = b.OC; w.lddy = b.lddy; w.dy_bs = b.dy_bs;
    w.KH = b.KH; w.KW = b.KW; w.stride = b.stride; w.pad = b.pad; w.dil = 1; w.tiles_per_split = 0; w.dbias = b.dbias; w.overwrite = 0;
    const bool vec = conv_desc_is_vec<T>(d) && wgrad_is_vec<T>(w) && d.OC > 32;
    groupable = groupable && vec;
    int tx = 1, ty = 1, S = 1;
    if (vec) wgrad_plan<T>(w, tx, ty, S);
    const long long Md = (long long)d.N * d.OH * d.OW;
    g.nd[i] = (int)(((Md + 63) / 64) * ((d.OC + 63) / 64));
    g.wtx[i] = tx; g.wty[i] = ty;
    g.first[i] = (int)total;
    total += g.nd[i] + (long long)tx * ty * S;
  }
  for (int i = n; i <= EMRT_MAX_GROUP; ++i) g.first[i] = (int)total;
  for (int i = n; i < EMRT_MAX_GROUP; ++i) { g.d[i] = g.d[0]; g.w[i] = g.w[0]; g.nd[i] = 0; g.wtx[i] = 1; g.wty[i] = 1; }
  if (!groupable || total > 4096) {
    for (int i = 0; i < n; ++i) {
      const int rc = conv_bwd_dispatch<T>(g.d[i], g.w[i], st);
      if (rc) return rc;
    }
    return 0;
  }
  auto kern = bwd_group_kernel<T>;
  const size_t lds_d = (size_t)2 * 128 * 144, lds_w = (size_t)4 * Cfg::BKM * Cfg::PITCH;
  const size_t lds = lds_d > lds_w ? lds_d : lds_w;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail("emrt_conv2d_bwd_group", "cannot raise the dynamic LDS limit");
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), lds, st, g);
  return check_launch("emrt_conv2d_bwd_group");
}

extern "C" int emrt_conv2d_bwd_group(const EmrtConvBwdDesc* descs, int n, int dtype, void* stream) {
  EMRT_REQUIRE(descs && n >= 1 && n <= EMRT_MAX_GROUP, "1..6 problems");
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  for (int i = 0; i < n; ++i) {
    const EmrtConvBwdDesc& b = descs[i];
    EMRT_REQUIRE(b.x && b.dy && b.w_bwd_packed && b.dx, "null pointer");
    EMRT_REQUIRE((b.dw == nullptr) == (descs[0].dw == nullptr) && (b.dw || !b.dbias), "dw == NULL (data gradients only) must hold for every problem of the group, without dbias");
    EMRT_REQUIRE(b.N > 0 && b.H > 0 && b.W > 0 && b.C > 0 && b.OC > 0 && b.KH > 0 && b.KW > 0 && b.stride > 0 && b.pad >= 0, "bad dims");
    EMRT_REQUIRE(b.OH == (b.H + 2 * b.pad - b.KH) / b.stride + 1 && b.OW == (b.W + 2 * b.pad - b.KW) / b.stride + 1, "output size mismatch");
    EMRT_REQUIRE((long long)b.N * b.OH * b.OW + 512 < (1ll << 31) && (long long)b.N * b.H * b.W + 512 < (1ll << 31), "more than 2^31 pixels");
    EMRT_REQUIRE(b.lddx >= b.C && b.dx_bs >= 0, "bad dx strides");
    const long long esz = dtype == EMRT_F32 ? 4 : 2;
    const long long x_ext = ((long long)(b.N - 1) * b.x_bs + ((long long)b.H * b.W - 1) * b.ldx + b.C) * esz;
    const long long dy_ext = ((long long)(b.N - 1) * b.dy_bs + ((long long)b.OH * b.OW - 1) * b.lddy + b.OC) * esz;
    EMRT_REQUIRE(b.x_bs >= 0 && b.dy_bs >= 0 && x_ext < (1ll << 31) && dy_ext < (1ll << 31), "operand spans 2 GiB or more");
    EMRT_REQUIRE((long long)b.H * b.W < (1 << 24) && (long long)b.ldx * esz < (1 << 24) && b.stride < (1 << 12), "map too large for the 24-bit address arithmetic");
  }
  hipStream_t st = (hipStream_t)stream;
  return dtype == EMRT_F32 ? conv_bwd_group_dispatch<float>(descs, n, st) : conv_bwd_group_dispatch<bf16_t>(descs, n, st);
}

// ---- data gradients of INDEPENDENT layers with their fused epilogues, side by side (ABI 9) ----------------------------------------------------
// emrt_conv2d_bwd_group's descriptor has no room for what the ResNet's data gradients carry in their epilogues (the producer's ReLU mask, the
// BatchNorm backward sums, an addend).  This one mirrors the data-gradient half of emrt_conv2d_bwd argument by argument, so that a layer3 block's
// conv1 data gradient (512 tiles of 4 k-tiles: a launch that is all ramp and tail) can take the spatial branch's pending 3x3 data gradient
// into its launch (functional.py: the tape's stash).  n == 1 is emrt_conv2d_bwd with dw == NULL.
struct EmrtConvDgradDesc {
  const void* dy; const void* w_bwd_packed; void* dx; int lddx; long long dx_bs; int accumulate;
  int N, H, W, C, OH, OW, OC, lddy; long long dy_bs; int KH, KW, stride, pad, dilation;
  double* bn_stats; const void* mask_y; int ldy; long long y_bs; float mask_scale; const void* stat_x; int ldsx; long long sx_bs;
  const void* addend; int ldadd; long long add_bs;
};

static void dgrad_args_from_desc(ConvArgs& d, const EmrtConvDgradDesc& b) {
  d.in = b.dy; d.w = b.w_bwd_packed; d.out = b.dx; d.bias = nullptr; d.scale = nullptr; d.res = nullptr;
  d.N = b.N; d.H = b.OH; d.W = b.OW; d.C = b.OC; d.ldin = b.lddy; d.in_bs = b.dy_bs;
  d.OH = b.H; d.OW = b.W; d.OC = b.C; d.ldout = b.lddx; d.out_bs = b.dx_bs;
  d.ldres = 0; d.res_bs = 0;
  if (b.accumulate) { d.res = b.dx; d.ldres = b.lddx; d.res_bs = b.dx_bs; }
  else if (b.addend) { d.res = b.addend; d.ldres = b.ldadd; d.res_bs = b.add_bs; }
  d.KH = b.KH; d.KW = b.KW; d.stride = b.stride; d.pad = b.pad; d.dil = b.dilation; d.relu = 0; d.out_f32 = 0; d.cmajor = g_tune.igemm8p_cmajor; d.stats = b.bn_stats;
  d.mask_y = b.mask_y; d.ldy = b.ldy; d.y_bs = b.y_bs; d.mask_scale = b.mask_scale; d.stat_x = b.stat_x; d.ldsx = b.ldsx; d.sx_bs = b.sx_bs;
  d.xk_S = 0; d.xk_part = nullptr; d.xk_tick = nullptr; d.drop_seed = nullptr; d.drop_salt = 0; d.drop_p = 0.f;
}

template <class T>
static int conv_dgrad_multi_dispatch(const EmrtConvDgradDesc* descs, int n, hipStream_t st) {
  ConvGroupArgs g;
  bool groupable = n >= 2;
  long long total = 0;
  for (int i = 0; i < n; ++i) {
    ConvArgs& d = g.p[i];
    dgrad_args_from_desc(d, descs[i]);
    const long long Md = (long long)d.N * d.OH * d.OW;
    const long long nb = ((Md + 63) / 64) * ((d.OC + 63) / 64);
    g.first[i] = (int)total;
    total += nb;
    groupable = groupable && conv_desc_is_vec<T>(d) && d.OC > 32 && nb <= 2048;
  }
  for (int i = n; i <= EMRT_MAX_GROUP; ++i) g.first[i] = (int)total;
  for (int i = n; i < EMRT_MAX_GROUP; ++i) g.p[i] = g.p[0];
  if (!groupable || total > 4096) {
    for (int i = 0; i < n; ++i) {
      const int rc = conv_dispatch<T, 1>(g.p[i], st);
      if (rc) return rc;
    }
    return 0;
  }
  hipLaunchKernelGGL((igemm_group_kernel<T, 1>), dim3((unsigned)total), dim3(256), (size_t)2 * 128 * 144, st, g);
  return check_launch("emrt_conv2d_dgrad_multi");
}

extern "C" int emrt_conv2d_dgrad_multi(const EmrtConvDgradDesc* descs, int n, int dtype, void* stream) {
  EMRT_REQUIRE(descs && n >= 1 && n <= EMRT_MAX_GROUP, "1..6 problems");
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  for (int i = 0; i < n; ++i) {
    const EmrtConvDgradDesc& b = descs[i];
    EMRT_REQUIRE(b.dy && b.w_bwd_packed && b.dx, "null pointer");
    EMRT_REQUIRE(b.N > 0 && b.H > 0 && b.W > 0 && b.C > 0 && b.OH > 0 && b.OW > 0 && b.OC > 0, "bad dims");
    EMRT_REQUIRE(b.KH > 0 && b.KW > 0 && b.stride > 0 && b.pad >= 0 && b.dilation >= 1, "bad kernel geometry");
    EMRT_REQUIRE(b.OH == (b.H + 2 * b.pad - b.dilation * (b.KH - 1) - 1) / b.stride + 1 && b.OW == (b.W + 2 * b.pad - b.dilation * (b.KW - 1) - 1) / b.stride + 1, "output size mismatch");
    EMRT_REQUIRE((long long)b.N * b.OH * b.OW + 512 < (1ll << 31) && (long long)b.N * b.H * b.W + 512 < (1ll << 31), "more than 2^31 pixels (32-bit pixel arithmetic)");
    const long long esz = dtype == EMRT_F32 ? 4 : 2;
    const long long dy_ext = ((long long)(b.N - 1) * b.dy_bs + ((long long)b.OH * b.OW - 1) * b.lddy + b.OC) * esz;
    const long long w_ext = (long long)b.OC * b.KH * b.KW * b.C * esz;
    EMRT_REQUIRE(b.dy_bs >= 0 && dy_ext < (1ll << 31) && w_ext < (1ll << 31), "operand spans 2 GiB or more (32-bit buffer offsets)");
    EMRT_REQUIRE((long long)b.OH * b.OW < (1 << 24) && (long long)b.lddy * esz < (1 << 24) && b.stride < (1 << 12), "map too large for the 24-bit address arithmetic");
    EMRT_REQUIRE(b.lddx >= b.C && b.dx_bs >= 0, "bad dx strides");
    EMRT_REQUIRE(!(b.accumulate && b.addend), "accumulate adds into dx itself; addend is a different tensor");
    EMRT_REQUIRE(!b.stat_x || b.mask_y, "stat_x replaces the mask tensor in the second statistic: it needs mask_y");
    for (int j = 0; j < i; ++j) EMRT_REQUIRE(descs[j].dx != b.dx, "two problems of one launch must not write the same dx");
  }
  hipStream_t st = (hipStream_t)stream;
  return dtype == EMRT_F32 ? conv_dgrad_multi_dispatch<float>(descs, n, st) : conv_dgrad_multi_dispatch<bf16_t>(descs, n, st);
}

// ------------------------------------------------------------------------------------------------
// Batched weight gradients.  dW(L) needs only x(L) and dy(L): nothing in backward waits for it, so the caller (functional.py) runs each
// small layer's DATA gradient as its own launch (emrt_conv2d_bwd with dw == NULL: 64x64 tiles at 4 blocks per CU instead of the pair
// kernel's 2) and hands the weight gradients of many layers to ONE launch here.  A lone small weight gradient has to cut its pixel
// reduction into 8-64 slices to fill 256 CUs and then pays S * |dW| of fp32 atomics (measured: 14 of the 35 us of a 32x32x256->256 3x3
// layer); a batch of 8-24 layers fills the machine with 1-4 slices each.
// ------------------------------------------------------------------------------------------------
#define EMRT_MAX_WGROUP 24
struct EmrtWgradDesc {
  const void* x; const void* dy; float* dw; float* dbias;
  int N, H, W, C, ldx; long long x_bs;
  int OH, OW, OC, lddy; long long dy_bs;
  int KH, KW, stride, pad, dilation;
  int dw_is_zero;
};
struct WgradGroupArgs {
  WgradArgs w[EMRT_MAX_WGROUP];
  int first[EMRT_MAX_WGROUP + 1];      // work items of problem i: [first[i], first[i+1]), slice-major (slice, oc tile, k tile)
  short wtx[EMRT_MAX_WGROUP], wty[EMRT_MAX_WGROUP];
  int n, xcd;
};

template <class T>
__global__ __launch_bounds__(256, 2) void wgrad_group_kernel(WgradGroupArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  const int nb = g.first[g.n];
  int w = (int)blockIdx.x;
  if (g.xcd) {      // as wgrad_kernel: the work items are cut into 8 contiguous ranges, one per XCD (the tiles of a slice stream the same rows)
    const int k = w & 7, idx = w >> 3;
    const int w0 = (int)(((long long)k * nb) >> 3), w1 = (int)(((long long)(k + 1) * nb) >> 3);
    w = w0 + idx;
    if (w >= w1) return;
  } else if (w >= nb) return;
  int i = 0;
  for (int k = 1; k < g.n; ++k) i += w >= g.first[k] ? 1 : 0;
  i = __builtin_amdgcn_readfirstlane(i);
  const int local = w - g.first[i];
  const int tx = g.wtx[i], ty = g.wty[i];
  const int bz = __builtin_amdgcn_readfirstlane(local / (tx * ty)), t = __builtin_amdgcn_readfirstlane(local - bz * (tx * ty));
  const int by = __builtin_amdgcn_readfirstlane(t / tx);
  wgrad_body<T, true, 1>(g.w[i], __builtin_amdgcn_readfirstlane(t - by * tx), by, bz, smem_all);
}

static void wgrad_args_from_desc(WgradArgs& a, const EmrtWgradDesc& d) {
  a.x = d.x; a.dy = d.dy; a.dw = d.dw;
  a.N = d.N; a.H = d.H; a.W = d.W; a.C = d.C; a.ldx = d.ldx; a.x_bs = d.x_bs;
  a.OH = d.OH; a.OW = d.OW; a.OC = d.OC; a.lddy = d.lddy; a.dy_bs = d.dy_bs;
  a.KH = d.KH; a.KW = d.KW; a.stride = d.stride; a.pad = d.pad; a.dil = d.dilation; a.tiles_per_split = 0; a.dbias = d.dbias;
  a.overwrite = d.dw_is_zero ? 1 : 0;      // (cleared below for every problem that ends up with more than one slice)
}

template <class T>
static int wgrad_group_dispatch(const EmrtWgradDesc* descs, int n, hipStream_t st) {
  using Cfg = WgradCfg<T>;
  WgradArgs pend[EMRT_MAX_WGROUP];
  int npend = 0;
  auto flush = [&]() -> int {
    if (npend == 0) return 0;
    if (npend == 1) { const int rc = wgrad_dispatch<T>(pend[0], st); npend = 0; return rc; }
    int tx[EMRT_MAX_WGROUP], ty[EMRT_MAX_WGROUP], S1[EMRT_MAX_WGROUP];
    long long mt[EMRT_MAX_WGROUP], work = 0;
    for (int i = 0; i < npend; ++i) {
      WgradArgs tmp = pend[i];
      wgrad_plan<T>(tmp, tx[i], ty[i], S1[i]);
      mt[i] = ((long long)pend[i].N * pend[i].OH * pend[i].OW + Cfg::BKM - 1) / Cfg::BKM;
      work += (long long)tx[i] * ty[i] * mt[i];
    }
    int order[EMRT_MAX_WGROUP];
    long long tps[EMRT_MAX_WGROUP], Sl[EMRT_MAX_WGROUP];
    {
      // ONE block length T (pixel tiles per block) for the whole batch; problem i is cut into ceil(mt_i / T) slices, never more than it would
      // take alone.  T = max(work / wgroup_blocks, wgroup_min_steps) = max(work / 1024, 32), from sweeps of the step's own mixes on MI355X
      // (tools/bench_conv.py wgroup mixes, profiles/r4_wgroup_plan.txt): large batches (an encoder layer's 24 weight gradients, 87 GFLOP)
      // want ~1000 blocks (158 us; 217 us with 128 long blocks), but cutting blocks shorter than ~32 tiles only buys fp32 atomic traffic --
      // every slice re-adds its problem's whole dW at 1.3 TB/s: ResNet layer3 x18 (28 MB of dW) 59 us with one slice each, 86 us with two to
      // three; the decoder's twelve small linears 26 vs 48 us.  (A fitted cost model -- block time c0 + c1 T, 512 resident blocks, atomics at
      // 1.3 TB/s -- was tried in its place: it cannot see how much of the atomic tail the next wave of blocks hides, and chose worse.)
      long long Tt = (work + g_tune.wgroup_blocks - 1) / (g_tune.wgroup_blocks > 0 ? g_tune.wgroup_blocks : 1024);
      if (Tt < g_tune.wgroup_min_steps) Tt = g_tune.wgroup_min_steps;
      if (Tt < 1) Tt = 1;
      for (int i = 0; i < npend; ++i) {
        long long S = (mt[i] + Tt - 1) / Tt;
        Sl[i] = S > S1[i] ? S1[i] : (S < 1 ? 1 : S);
      }
    }
    // longest blocks first (the tail of the launch is then made of short ones)
    for (int i = 0; i < npend; ++i) {
      tps[i] = (mt[i] + Sl[i] - 1) / Sl[i];
      order[i] = i;
    }
    for (int i = 1; i < npend; ++i)
      for (int j = i; j > 0 && tps[order[j]] > tps[order[j - 1]]; --j) { const int t_ = order[j]; order[j] = order[j - 1]; order[j - 1] = t_; }
    WgradGroupArgs g;
    long long total = 0;
    for (int q = 0; q < npend; ++q) {
      const int i = order[q];
      g.w[q] = pend[i];
      g.w[q].tiles_per_split = (int)tps[i];
      const long long S = (mt[i] + tps[i] - 1) / tps[i];
      if (S > 1 || g_tune.wgrad_no_overwrite) g.w[q].overwrite = 0;
      g.first[q] = (int)total;
      g.wtx[q] = (short)tx[i]; g.wty[q] = (short)ty[i];
      total += (long long)tx[i] * ty[i] * S;
    }
    for (int q = npend; q <= EMRT_MAX_WGROUP; ++q) g.first[q] = (int)total;
    for (int q = npend; q < EMRT_MAX_WGROUP; ++q) { g.w[q] = g.w[0]; g.wtx[q] = 1; g.wty[q] = 1; }
    g.n = npend; g.xcd = g_tune.wgrad8p_xcd;
    npend = 0;
    if (total >= (1ll << 30)) return fail("emrt_conv2d_wgrad_group", "too many blocks");
    auto kern = wgrad_group_kernel<T>;
    const size_t lds = (size_t)4 * Cfg::BKM * Cfg::PITCH;
    static bool attr_done = false;      // one flag per element type
    if (!attr_done) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return fail("emrt_conv2d_wgrad_group", "cannot raise the dynamic LDS limit");
      attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(8u * (unsigned)((total + 7) / 8)), dim3(256), lds, st, g);
    return check_launch("emrt_conv2d_wgrad_group");
  };
  // the problems that fit the 256x256 LDS-DMA kernel's shape rules but not its "fills the machine alone" clause go out TOGETHER on that kernel
  // (wgrad8p.hpp: launch_wgrad8p_group) when the call holds enough of them: an encoder layer's FFN linears and 3x3 convolutions (3 192 tile-steps),
  // ResNet layer3 / layer4 (few pixels, large dW: one slice each, tiles stored)
  WgradArgs pend8[EMRT_MAX_WGROUP8];
  int npend8 = 0;
  auto flush8 = [&]() -> int {
    if (npend8 == 0) return 0;
    const int rc = launch_wgrad8p_group(pend8, npend8, st);
    npend8 = 0;
    return rc;
  };
  auto aliased = [&](int i) { for (int j = 0; j < n; ++j) if (j != i && descs[j].dw == descs[i].dw) return true; return false; };
  long long work8 = 0;
  if constexpr (std::is_same<T, bf16_t>::value) {
    for (int i = 0; i < n && g_tune.wgroup8 > 0; ++i) {
      WgradArgs a;
      wgrad_args_from_desc(a, descs[i]);
      int tk, toc, S8, per, tiles8, steps8;
      if (wgrad_is_vec<T>(a) && !wgrad8p_plan<T>(a, tk, toc, S8, per) && wgrad8p_group_ok<T>(a, tiles8, steps8) && !aliased(i)) work8 += (long long)tiles8 * steps8;
    }
  }
  const bool use8 = g_tune.wgroup8 > 0 && work8 >= (g_tune.wgroup8_min_work > 0 ? g_tune.wgroup8_min_work : 1);
  for (int i = 0; i < n; ++i) {
    WgradArgs a;
    wgrad_args_from_desc(a, descs[i]);
    // a weight used more than once in the step (a shared layer) appears as several problems with the same dw: a stored tile of one would
    // race with the atomic adds of the other (same launch), or land after them ('alone' problems are launched before the pended batch):
    // every problem whose dw another problem of this call also writes accumulates
    const bool alias = aliased(i);
    if (alias) a.overwrite = 0;
    bool alone = !wgrad_is_vec<T>(a) || a.KH * a.KW * a.C >= 32768 * 128 || a.OC >= 32768 * 128;     // (tile counts are shorts)
    if constexpr (std::is_same<T, bf16_t>::value) {
      int tk, toc, S8, per;
      if (wgrad8p_plan<T>(a, tk, toc, S8, per)) alone = true;      // a large layer: the 256x256 LDS-DMA kernel, its own launch
    }
    if (alone) {
      const int rc = wgrad_dispatch<T>(a, st);
      if (rc) return rc;
      continue;
    }
    if constexpr (std::is_same<T, bf16_t>::value) {
      int tiles8, steps8;
      if (use8 && !alias && wgrad8p_group_ok<T>(a, tiles8, steps8)) {
        pend8[npend8++] = a;
        if (npend8 == EMRT_MAX_WGROUP8) {
          const int rc = flush8();
          if (rc) return rc;
        }
        continue;
      }
    }
    pend[npend++] = a;
    if (npend == EMRT_MAX_WGROUP || (g_tune.wgroup_max > 0 && npend >= g_tune.wgroup_max)) {
      const int rc = flush();
      if (rc) return rc;
    }
  }
  {
    const int rc = flush8();
    if (rc) return rc;
  }
  return flush();
}

extern "C" int emrt_conv2d_wgrad_group(const EmrtWgradDesc* descs, int n, int dtype, void* stream) {
  EMRT_REQUIRE(descs && n >= 1, "no problems");
  EMRT_REQUIRE_TRAIN_DTYPE(dtype);
  for (int i = 0; i < n; ++i) {
    const EmrtWgradDesc& d = descs[i];
    EMRT_REQUIRE(d.x && d.dy && d.dw, "null pointer");
    EMRT_REQUIRE(d.N > 0 && d.H > 0 && d.W > 0 && d.C > 0 && d.OH > 0 && d.OW > 0 && d.OC > 0 && d.dilation >= 1, "bad dims");
    EMRT_REQUIRE(d.KH > 0 && d.KW > 0 && d.stride > 0 && d.pad >= 0, "bad kernel geometry");
    EMRT_REQUIRE(d.OH == (d.H + 2 * d.pad - d.dilation * (d.KH - 1) - 1) / d.stride + 1 && d.OW == (d.W + 2 * d.pad - d.dilation * (d.KW - 1) - 1) / d.stride + 1, "output size mismatch");
    EMRT_REQUIRE((long long)d.N * d.OH * d.OW + 512 < (1ll << 31) && (long long)d.N * d.H * d.W + 512 < (1ll << 31), "more than 2^31 pixels (32-bit pixel arithmetic)");
    const long long esz = dtype == EMRT_F32 ? 4 : 2;
    const long long x_ext = ((long long)(d.N - 1) * d.x_bs + ((long long)d.H * d.W - 1) * d.ldx + d.C) * esz;
    const long long dy_ext = ((long long)(d.N - 1) * d.dy_bs + ((long long)d.OH * d.OW - 1) * d.lddy + d.OC) * esz;
    EMRT_REQUIRE(d.x_bs >= 0 && d.dy_bs >= 0 && x_ext < (1ll << 31) && dy_ext < (1ll << 31), "operand spans 2 GiB or more (32-bit buffer offsets)");
    EMRT_REQUIRE((long long)d.H * d.W < (1 << 24) && (long long)d.ldx * esz < (1 << 24) && d.stride < (1 << 12), "map too large for the 24-bit address arithmetic");
  }
  hipStream_t st = (hipStream_t)stream;
  return dtype == EMRT_F32 ? wgrad_group_dispatch<float>(descs, n, st) : wgrad_group_dispatch<bf16_t>(descs, n, st);
}
