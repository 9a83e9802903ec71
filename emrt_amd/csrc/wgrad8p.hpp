// 256 (oc) x 256 (k) weight-gradient GEMM for the LARGE layers, bf16 -- included by conv.hip after igemm8p.hpp.
//
//   dW[oc][tap * C + c] += sum_m dy[m][oc] * x[pix(m, tap)][c]          (m = output pixel: the REDUCTION dimension)
//
// Same eight-phase LDS-DMA skeleton as igemm8p.hpp (read its header for the schedule and the WAR / RAW argument; they hold verbatim:
// a "k-tile" here is one step of 64 pixels).  What differs:
//   * a block owns ONE tap and 256 of its channels (C % 256 == 0) x 256 output channels, and a slice z of the pixels;
//   * the grid is ONE-dimensional and XCD-aware: consecutive workgroup ids go round-robin to the 8 XCDs, each with its own L2, and the
//     blocks that stream the SAME pixels (same slice, the KH*KW taps x channel blocks x oc tiles) must share one: the work items
//     w = (slice, oc tile, channel block, tap) in that order are cut into 8 contiguous ranges, XCD k = id % 8 takes range k.  Without it the
//     taps of a slice sat on 8 different XCDs and every block pulled its 64 KiB per step from the fabric: 1.2 GB per launch on UpHead's
//     conv_2 instead of the 134 MB of x + dy (measured: tools/bench_conv.py wbig, profiles/r3_wgrad_8phase_vs_128.txt, r3_pmc_wgrad_8phase.txt);
//   * both operands are PIXEL-major in memory (64 pixels x 512 bytes per step and operand), the MFMA wants 8 consecutive pixels per lane:
//     fragments come out of LDS through the transposing read ds_read_b64_tr_b16 (two per 32x16 fragment), as in wgrad_body;
//   * the DMA source address is per lane, so the LDS image need not look like memory: every step's tile is stored as COLUMN SLICES, one per
//     (wave row / column, half) -- dy as 4 sub-images [64 pixels][64 oc] (128-byte rows), x as 8 sub-images [64 pixels][32 c] (64-byte rows)
//     -- which is what makes the four 16 KiB units of a step (a0 / a1 halves of dy, b0 / b1 halves of x) separately re-stageable;
//     the 128-byte-row images XOR the 64-byte half of a row with bit 1 of the row (conflict-free transposed reads), permutation on the source;
//   * a thread stages ONE dy pixel row and ONE x pixel row per step (four 16-byte pieces of each); OH * OW % 64 == 0 (host), so a step lies
//     inside one image: the image and the step's first pixel are wave-uniform and live in scalar registers (the buffer instruction's
//     soffset), a lane keeps one constant byte offset (dy) or its (oh, ow) pair advanced with one carry (x); no division in the loop;
//   * the 256 x 256 fp32 tile is a PARTIAL sum over the block's pixel slice: written to a scratch slab (emrt_set_scratch) and added
//     into dW by wgrad8p_reduce_kernel, or -- without scratch -- added with fp32 atomics (two 128-byte segments per wave instruction);
//   * the bias gradient (sum_m dy[m][oc]) is taken from the dy FRAGMENTS by the wave column 0 of the blocks of the k-tile whose turn it
//     is (step % k-tiles), so that it costs every block the same ~1 % instead of one block in nine 10 %.
#pragma once

struct Wgrad8pArgs {
  const void* x;
  const void* dy;
  float* dw;
  float* slab;          // nullptr: atomics into dw; else [S][tiles_oc][tiles_k][256][256] partial tiles
  float* dbias;
  int N, H, W, C, ldx;
  long long x_bs;
  int OH, OW, OC, lddy;
  long long dy_bs;
  int KH, KW, stride, pad, dil;
  int steps_per_split, steps_total;
  int tiles_k, tiles_oc, S, xcd_aware;
  int direct;           // 1: this launch is the ONLY contribution to a dW the caller vouches is all zero (S == 1): the tile is stored, neither slab nor atomics
};

// LDS-DMA wave instruction with a wave-uniform byte offset in the instruction's soffset (not part of the range check)
__device__ __forceinline__ void lds_dma16s(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_addr), "s"(rs), "s"(__builtin_amdgcn_readfirstlane(soff)) : "memory");
}

__device__ __forceinline__ uint2 lds_tr16(const unsigned char* p) {
  short4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)p);
  return __builtin_bit_cast(uint2, v);
}

// One block's work: work item w = (slice, oc tile, channel block, tap) of problem p (the caller has mapped its block id to w).  smem = the block's whole
// LDS (128 KiB, at LDS address 0: the DMA writes take absolute LDS addresses).
__device__ __forceinline__ void wgrad8p_body(const Wgrad8pArgs& p, const int w, unsigned char* smem) {
  constexpr unsigned BUFB = 65536u, AB = 32768u;      // bytes per step buffer; offset of the x tile inside it
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int cblks = p.C >> 8, ntap = p.KH * p.KW;
  // the work item's coordinates; integer divisions run on the vector ALU: their wave-uniform results go back to scalar registers
  int bz, by, kt, tap, c0;
  {
    const int tiles = p.tiles_k * p.tiles_oc;
    bz = __builtin_amdgcn_readfirstlane(w / tiles);
    const int t = __builtin_amdgcn_readfirstlane(w - bz * tiles);
    by = __builtin_amdgcn_readfirstlane(t / p.tiles_k);
    const int r = __builtin_amdgcn_readfirstlane(t - by * p.tiles_k);
    const int cb = __builtin_amdgcn_readfirstlane(r / ntap);
    tap = __builtin_amdgcn_readfirstlane(r - cb * ntap);
    c0 = cb << 8;
    kt = tap * cblks + cb;                             // k-tile index in dW's k order (tap-major)
  }
  const int kh = __builtin_amdgcn_readfirstlane(tap / p.KW), kw = __builtin_amdgcn_readfirstlane(tap - kh * p.KW);
  const int oc0 = by << 8;
  const int K = p.KH * p.KW * p.C;
  const int st_begin = bz * p.steps_per_split;
  int st_end = st_begin + p.steps_per_split;
  if (st_end > p.steps_total) st_end = p.steps_total;
  const int nst = st_end - st_begin;                   // >= 1 (host)
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)BUF_RANGE, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)BUF_RANGE, 0x00020000);
  const int OHW = p.OH * p.OW;

  // ---- loader state: ONE dy pixel row (A) and ONE x pixel row (B) of the step per thread ------------------------------------------
  // A piece (u, j): sub-image (wave row j, half u) = output channels oc0 + 128 j + 64 u + [0, 64), 8 pixel rows 8 wave + (lane >> 3);
  // the lane's 16-byte chunk sits at position lane & 7 and holds source chunk (lane & 7) ^ 4 * bit 1 of the row.
  // B piece (u, j): sub-image (wave column (wave >> 2) + 2 j, half u) = channels c0 + 64 wcol + 32 u + [0, 32), 16 pixel rows 16 (wave & 3) + (lane >> 2)
  // OH * OW is a multiple of 64 (host): a step lies inside ONE image, so the image and the step's first pixel are wave-uniform (scalar
  // registers, the buffer instruction's soffset) and a lane keeps only its row's constant byte offset (dy) or its (oh, ow) pair (x).
  const int rowA = 8 * wave + (lane >> 3), rowB = 16 * (wave & 3) + (lane >> 2);
  const unsigned lddy_b = (unsigned)p.lddy * 2u, ldx_b = (unsigned)p.ldx * 2u;
  const unsigned a_lane = (unsigned)rowA * lddy_b + (unsigned)(((lane & 7) ^ (((rowA >> 1) & 1) << 2)) * 16);
  const unsigned chunkB = (unsigned)((lane & 3) * 16);
  const unsigned dy_img_b = (unsigned)(p.dy_bs * 2), x_img_b = (unsigned)(p.x_bs * 2);
  int a_pix0, b_pix0;                                  // first pixel of the next step to stage, inside its image (scalar)
  unsigned a_base, b_base;                             // byte offset of that pixel's dy row / of the x image (scalar)
  int b_oh, b_ow;                                      // this lane's output pixel in the next x step
  {
    const unsigned m0 = (unsigned)st_begin * 64u;      // < 2^31 (host-checked)
    const int img = __builtin_amdgcn_readfirstlane((int)(m0 / (unsigned)OHW));      // (the division runs on the vector ALU: back to a scalar register)
    a_pix0 = b_pix0 = __builtin_amdgcn_readfirstlane((int)(m0 - (unsigned)img * (unsigned)OHW));
    a_base = (unsigned)img * dy_img_b + (unsigned)a_pix0 * lddy_b;
    b_base = (unsigned)img * x_img_b;
    const int pixB = b_pix0 + rowB;
    b_oh = (int)((unsigned)pixB / (unsigned)p.OW);
    b_ow = pixB - b_oh * p.OW;
  }
  const int adv_w = 64 % p.OW, adv_h = 64 / p.OW;      // (adv_h < OH: 64 pixels are less than an image)
  const int hi0 = kh * p.dil - p.pad, wi0 = kw * p.dil - p.pad;
  int ast = 0, bst = 0;                                // steps staged so far (A units / B units advance at different phases)
  auto advance_a = [&]() {
    ++ast;
    a_pix0 += 64;
    a_base += 64u * lddy_b;
    if (a_pix0 == OHW) { a_pix0 = 0; a_base += dy_img_b - (unsigned)OHW * lddy_b; }
  };
  auto advance_b = [&]() {
    ++bst;
    b_pix0 += 64;
    if (b_pix0 == OHW) { b_pix0 = 0; b_base += x_img_b; }
    b_ow += adv_w;
    const int c1 = b_ow >= p.OW ? 1 : 0;
    b_ow -= c1 ? p.OW : 0;
    b_oh += adv_h + c1;
    b_oh -= b_oh >= p.OH ? p.OH : 0;                   // (exactly when the scalar cursor wrapped)
  };
  auto stage_a = [&](int u, int par) {                 // unit UAu (dy columns of half u, both wave rows) of step ast -> buffer par
    const unsigned voff = a_lane | (ast < nst ? 0u : BUF_OOB);      // (the range check sees voffset only: bit 31 = out of range = zeros)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      lds_dma16s(rs_dy, voff, a_base + (unsigned)(oc0 + 128 * j + 64 * u) * 2u,
                 (unsigned)par * BUFB + (unsigned)(2 * j + u) * 8192u + (unsigned)(8 * wave) * 128u);
  };
  auto stage_b = [&](int u, int par) {                 // unit UBu (x columns of half u, all four wave columns) of step bst -> buffer par
    // 24-bit multiplies (full rate): coordinates, H * W and the pixel stride in bytes are all < 2^24 (host-checked)
    const int hi = __mul24(b_oh, p.stride) + hi0, wi = __mul24(b_ow, p.stride) + wi0;
    const bool ok = bst < nst && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
    const unsigned voff = ok ? __umul24((unsigned)(__mul24(hi, p.W) + wi), ldx_b) + chunkB : BUF_OOB;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int wcol = (wave >> 2) + 2 * j;
      lds_dma16s(rs_x, voff, b_base + (unsigned)(c0 + 64 * wcol + 32 * u) * 2u,
                 (unsigned)par * BUFB + AB + (unsigned)(2 * wcol + u) * 4096u + (unsigned)(16 * (wave & 3)) * 64u);
    }
  };

  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- fragment addresses (transposing reads): lane -> group g = lane >> 4 (h = g >> 1: pixel rows 8 h .., half = g & 1: 16 columns),
  // t = lane & 15 (q = t >> 2: row, pp = t & 3: 4 columns) -----------------------------------------------------------------------------
  const int frow = lane & 31, fh = lane >> 5;
  // row of sub-step s = 16 s + 8 h + q: bit 1 of the row (the swizzle) is bit 1 of q for every s, so a lane's address is ONE base per m-tile
  // plus compile-time offsets (2048 s for dy, 1024 s for x; the +4-row read of a fragment: +512 / +256): the instruction's immediate field
  unsigned baseA[2], baseB;
  {
    const int g = lane >> 4, t = lane & 15;
    const int h = g >> 1, half = g & 1, q = t >> 2, pp = t & 3;
    const int row = 8 * h + q;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int chunk = (i * 4 + 2 * half + (pp >> 1)) ^ (((row >> 1) & 1) << 2);
      baseA[i] = (unsigned)(row * 128 + chunk * 16 + 8 * (pp & 1)) + (unsigned)(wr * 2) * 8192u;      // + sub-images (wr, half 0)
    }
    baseB = (unsigned)(row * 64 + 32 * half + 8 * pp) + AB + (unsigned)(wc * 2) * 4096u;              // + sub-images (wc, half 0)
  }
  uint4 fa[2][4], fb0[4], fb1[4];
  auto read_a = [&](unsigned off, int i, int s) -> uint4 {       // 32 oc x 16 pixels of dy
    const unsigned char* q = smem + baseA[i] + off + (unsigned)s * 2048u;
    const uint2 lo = lds_tr16(q), hi = lds_tr16(q + 512);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
  };
  auto read_b = [&](unsigned off, int s) -> uint4 {              // 32 k x 16 pixels of x
    const unsigned char* q = smem + baseB + off + (unsigned)s * 1024u;
    const uint2 lo = lds_tr16(q), hi = lds_tr16(q + 256);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
  };
  // bias gradient: sum over the pixels of this lane's dy fragment elements (rows of the A operand), in the blocks whose turn it is
  const int ktiles = p.tiles_k;
  int turn = 0;                                        // step % ktiles
  const bool bias_wave = p.dbias != nullptr && wc == 0;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  auto bias_add = [&](int it, const uint4& f) {
    const unsigned w4[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) bsum[it] += bf16_bits_to_f32(w4[e] & 0xffffu) + __uint_as_float(w4[e] & 0xffff0000u);
  };

  // ---- prologue ---------------------------------------------------------------------------------------------------------------------
  stage_a(0, 0); stage_b(0, 0); stage_b(1, 0); stage_a(1, 0);
  advance_a(); advance_b();
  stage_a(0, 1); stage_b(1, 1); stage_a(1, 1);        // UB0 of step 1 follows in phase 1 of step 0
  advance_a();
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();

  auto phase = [&](auto Qc, auto PARc) {
    constexpr int Q = decltype(Qc)::value, PAR = decltype(PARc)::value;
    const unsigned buf = (unsigned)PAR * BUFB;
    const unsigned a_sub = buf, b_sub = buf;            // (the wave's sub-image pair is part of baseA / baseB; half 1: + 8192 / + 4096)
    if constexpr (Q == 1) {
#pragma unroll
      for (int s = 0; s < 4; ++s) fb0[s] = read_b(b_sub, s);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 4; ++s) fa[i][s] = read_a(a_sub, i, s);
      stage_b(0, PAR ^ 1);
      advance_b();
    } else if constexpr (Q == 2) {
#pragma unroll
      for (int s = 0; s < 4; ++s) fb1[s] = read_b(b_sub + 4096u, s);
      stage_a(0, PAR);
    } else if constexpr (Q == 3) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 4; ++s) fa[i][s] = read_a(a_sub + 8192u, i, s);
      stage_b(1, PAR);
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) fb0[s] = read_b(b_sub, s);
      stage_a(1, PAR);
      advance_a();
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
    constexpr int AH = (Q >= 3) ? 1 : 0, BH = (Q == 2 || Q == 3) ? 1 : 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        acc[AH * 2 + i][BH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[i][s]),
                                                                        __builtin_bit_cast(bf16x8_t, BH ? fb1[s] : fb0[s]), acc[AH * 2 + i][BH], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    if constexpr (Q == 1 || Q == 3) {                  // the a0 / a1 fragments have just been used: their pixel sums, when it is this block's turn
      if (bias_wave && turn == kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int s = 0; s < 4; ++s) bias_add(AH * 2 + i, fa[i][s]);
      }
    }
    if constexpr (Q == 4) turn = turn + 1 == ktiles ? 0 : turn + 1;
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>;
  for (int step = 0; step < nst; step += 2) {          // (an odd tail multiplies one all-zero step)
    phase(I1{}, I0{}); phase(I2{}, I0{}); phase(I3{}, I0{}); phase(I4{}, I0{});
    phase(I1{}, I1{}); phase(I2{}, I1{}); phase(I3{}, I1{}); phase(I4{}, I1{});
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: the partial tile straight from the accumulators (a lane holds one k column: 32 lanes = 128 contiguous bytes per row) ------
  const int kcol0 = wc * 64 + frow;                    // + 32 jb
#pragma unroll
  for (int it = 0; it < 4; ++it)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ocl = wr * 128 + it * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        const int kl = kcol0 + 32 * jb;
        if (p.slab) {
          float* dst = p.slab + (((long long)bz * p.tiles_oc + by) * p.tiles_k + kt) * 65536ll + ocl * 256 + kl;
          __builtin_nontemporal_store(acc[it][jb][r], dst);
        } else if (p.direct) {
          p.dw[(long long)(oc0 + ocl) * K + (long long)tap * p.C + c0 + kl] = acc[it][jb][r];
        } else {
          atomicAdd(p.dw + (long long)(oc0 + ocl) * K + (long long)tap * p.C + c0 + kl, acc[it][jb][r]);
        }
      }
    }
  if (bias_wave) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const float v = bsum[it] + __shfl_xor(bsum[it], 32, 64);       // the two half-waves hold the two pixel halves of the same oc row
      if (fh == 0) atomicAdd(p.dbias + oc0 + wr * 128 + it * 32 + frow, v);
    }
  }
}

__global__ __launch_bounds__(512, 2) void wgrad8p_kernel(Wgrad8pArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // work item of this block (see the header)
  const int nb = p.tiles_k * p.tiles_oc * p.S;
  const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
  const int w0 = (int)(((long long)xcd * nb) >> 3), w1 = (int)(((long long)(xcd + 1) * nb) >> 3);
  const int w = p.xcd_aware ? w0 + idx : (int)blockIdx.x;      // (0: A/B knob wgrad8p_xcd, work items in launch order)
  if (w >= (p.xcd_aware ? w1 : nb)) return;            // (the grid is 8 x the longest range)
  wgrad8p_body(p, w, smem);
}

// Several weight gradients on 256 x 256 tiles in ONE launch (emrt_conv2d_wgrad_group): the layers of a batch that fit the kernel's shape rules but are too
// small to fill 256 CUs alone -- an encoder layer's FFN linears and 3x3 convolutions, ResNet layer3 / layer4 (8-32 steps of 64 pixels, 4-36 tiles each).
// The work items of all problems form one list (problem-major, each problem slice-major as above); the host cuts it into 8 contiguous ranges of about
// equal COST (a block's steps + its prologue / epilogue), XCD k = block id % 8 walks range k: the blocks that stream the same pixels still share an L2.
#define EMRT_MAX_WGROUP8 20
struct Wgrad8pGroupArgs {
  Wgrad8pArgs w[EMRT_MAX_WGROUP8];
  int first[EMRT_MAX_WGROUP8 + 1];      // work items of problem i: [first[i], first[i + 1])
  int xfirst[9];                        // work items of XCD k: [xfirst[k], xfirst[k + 1])
  int n;
};
__global__ __launch_bounds__(512, 2) void wgrad8p_group_kernel(Wgrad8pGroupArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
  const int w = g.xfirst[xcd] + idx;
  if (w >= g.xfirst[xcd + 1]) return;
  int i = 0;
  for (int k = 1; k < g.n; ++k) i += w >= g.first[k] ? 1 : 0;
  i = __builtin_amdgcn_readfirstlane(i);
  const Wgrad8pArgs p = g.w[i];      // a private copy (scalar registers): through the reference every field would be re-read from the kernel-argument segment in the loop
  wgrad8p_body(p, __builtin_amdgcn_readfirstlane(w - g.first[i]), smem);
}

// dW[oc][tap * C + c] += sum over the pixel slices z of slab[z][oc tile][k tile][oc % 256][k % 256]; grid (64, tiles), one float4 per thread
__device__ __forceinline__ void wgrad8p_reduce_tile(const float* __restrict__ slab, float* __restrict__ dw, int S, int tiles_k, int tiles_oc, int K, const long long tile) {
  const int kt = (int)(tile % tiles_k), ot = (int)(tile / tiles_k);
  const long long zstride = (long long)tiles_k * tiles_oc * 65536ll;
  const int i4 = (int)(blockIdx.x * blockDim.x + threadIdx.x);      // < 16384
  typedef __attribute__((ext_vector_type(4))) float nt_f32x4;
  const nt_f32x4* src = reinterpret_cast<const nt_f32x4*>(slab + tile * 65536ll + i4 * 4);
  const long long zs = zstride / 4;
  nt_f32x4 a = __builtin_nontemporal_load(src);
  int z = 1;
  for (; z + 4 <= S; z += 4) {                         // four independent loads in flight per thread (fixed summation order)
    const nt_f32x4 b0 = __builtin_nontemporal_load(src + z * zs), b1 = __builtin_nontemporal_load(src + (z + 1) * zs);
    const nt_f32x4 b2 = __builtin_nontemporal_load(src + (z + 2) * zs), b3 = __builtin_nontemporal_load(src + (z + 3) * zs);
    a += b0; a += b1; a += b2; a += b3;
  }
  for (; z < S; ++z) a += __builtin_nontemporal_load(src + z * zs);
  const int ocl = (i4 * 4) >> 8, kl = (i4 * 4) & 255;
  float4* d = reinterpret_cast<float4*>(dw + (long long)(ot * 256 + ocl) * K + (long long)kt * 256 + kl);
  float4 o = *d;
  o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
  *d = o;
}

__global__ __launch_bounds__(256) void wgrad8p_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int S, int tiles_k, int tiles_oc, int K) {
  wgrad8p_reduce_tile(slab, dw, S, tiles_k, tiles_oc, K, (long long)blockIdx.y);      // oc_tile * tiles_k + k_tile
}

// the same for the multi-slice problems of a grouped launch: blockIdx.y walks the output tiles of all of them
struct Wgrad8pReduceGroupArgs {
  const float* slab[EMRT_MAX_WGROUP8];
  float* dw[EMRT_MAX_WGROUP8];
  int S[EMRT_MAX_WGROUP8], tiles_k[EMRT_MAX_WGROUP8], tiles_oc[EMRT_MAX_WGROUP8], K[EMRT_MAX_WGROUP8];
  int tfirst[EMRT_MAX_WGROUP8 + 1];
  int n;
};
__global__ __launch_bounds__(256) void wgrad8p_reduce_group_kernel(Wgrad8pReduceGroupArgs g) {
  const int t = (int)blockIdx.y;
  int i = 0;
  for (int k = 1; k < g.n; ++k) i += t >= g.tfirst[k] ? 1 : 0;
  wgrad8p_reduce_tile(g.slab[i], g.dw[i], g.S[i], g.tiles_k[i], g.tiles_oc[i], g.K[i], (long long)(t - g.tfirst[i]));
}

template <class T>
static bool wgrad8p_plan(const WgradArgs& a, int& tiles_k, int& tiles_oc, int& S, int& steps_per) {
  if (!std::is_same<T, bf16_t>::value || g_tune.wgrad8p_min_steps <= 0) return false;
  if (a.C % 256 != 0 || a.OC % 256 != 0) return false;
  if (a.ldx % 8 || a.lddy % 8 || a.x_bs % 8 || a.dy_bs % 8 || ((uintptr_t)a.x) % 16 || ((uintptr_t)a.dy) % 16 || ((uintptr_t)a.dw) % 16) return false;
  const long long M = (long long)a.N * a.OH * a.OW;
  if ((a.OH * a.OW) % 64 != 0) return false;           // whole 64-pixel steps, none across two images
  const int steps = (int)(M / 64);
  tiles_k = a.KH * a.KW * (a.C / 256);
  tiles_oc = a.OC / 256;
  const int tiles = tiles_k * tiles_oc;
  if (tiles > 256) return false;
  const int min_steps = g_tune.wgrad8p_force ? 1 : g_tune.wgrad8p_min_steps;
  S = 256 / tiles;                                     // one block per CU
  if (g_tune.wgrad_split > 0) S = g_tune.wgrad_split;  // developer knob (tools/bench_conv.py)
  if (S > steps / min_steps) S = steps / min_steps;
  if (S < 1) return false;                             // too few pixels per block: prologue / epilogue dominated
  steps_per = (steps + S - 1) / S;
  S = (steps + steps_per - 1) / steps_per;
  // >= 8 output tiles: with fewer (the FFN linears at 512x512: 4 tiles x 42 slices) the slab traffic per FLOP is too high and the 128 x 128
  // kernel wins (tools/bench_conv.py wbig: 35.0 vs 38.8 us)
  return g_tune.wgrad8p_force || ((long long)tiles * S >= 160 && tiles >= 8);
}

static int launch_wgrad8p(const WgradArgs& a, int tiles_k, int tiles_oc, int S, int steps_per, hipStream_t st) {
  Wgrad8pArgs w;
  w.x = a.x; w.dy = a.dy; w.dw = a.dw; w.dbias = a.dbias;
  w.N = a.N; w.H = a.H; w.W = a.W; w.C = a.C; w.ldx = a.ldx; w.x_bs = a.x_bs;
  w.OH = a.OH; w.OW = a.OW; w.OC = a.OC; w.lddy = a.lddy; w.dy_bs = a.dy_bs;
  w.KH = a.KH; w.KW = a.KW; w.stride = a.stride; w.pad = a.pad; w.dil = a.dil;
  w.steps_per_split = steps_per; w.steps_total = (int)((long long)a.N * a.OH * a.OW / 64);
  const size_t need = (size_t)S * tiles_k * tiles_oc * 65536 * sizeof(float);
  // the slab / reduce pair is only safe on the stream the scratch was registered for: a launch on any other stream (a second-stream
  // weight-gradient experiment, Context.overlap) takes the atomic epilogue instead of sharing the slab with a kernel it is not ordered against
  w.slab = (g_tune.wgrad8p_slab && S > 1 && g_scratch.ptr && g_scratch.bytes >= need && g_scratch.stream == (void*)st) ? (float*)g_scratch.ptr : nullptr;
  auto kern = wgrad8p_kernel;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072) != hipSuccess)
      return fail("emrt_conv2d_wgrad", "cannot raise the dynamic LDS limit to 128 KiB");
    attr_done = true;
  }
  w.tiles_k = tiles_k; w.tiles_oc = tiles_oc; w.S = S; w.xcd_aware = g_tune.wgrad8p_xcd; w.direct = 0;
  const int nb = tiles_k * tiles_oc * S;
  hipLaunchKernelGGL(kern, dim3(8 * ((nb + 7) / 8)), dim3(512), 131072, st, w);
  int rc = check_launch("emrt_conv2d_wgrad(8-phase)");
  if (rc || !w.slab) return rc;
  hipLaunchKernelGGL(wgrad8p_reduce_kernel, dim3(64, tiles_k * tiles_oc), dim3(256), 0, st, (const float*)w.slab, a.dw, S, tiles_k, tiles_oc, a.KH * a.KW * a.C);
  return check_launch("emrt_conv2d_wgrad(8-phase reduce)");
}

// ---- grouped launch (emrt_conv2d_wgrad_group) ------------------------------------------------------------------------------------------------
// Shape rules of the kernel without wgrad8p_plan's "fills the machine alone" clause: what may JOIN a grouped launch.
template <class T>
static bool wgrad8p_group_ok(const WgradArgs& a, int& tiles, int& steps) {
  if (!std::is_same<T, bf16_t>::value || g_tune.wgroup8 <= 0 || g_tune.wgrad8p_min_steps <= 0) return false;
  if (a.C % 256 != 0 || a.OC % 256 != 0) return false;
  if (a.ldx % 8 || a.lddy % 8 || a.x_bs % 8 || a.dy_bs % 8 || ((uintptr_t)a.x) % 16 || ((uintptr_t)a.dy) % 16 || ((uintptr_t)a.dw) % 16) return false;
  if ((a.OH * a.OW) % 64 != 0) return false;           // whole 64-pixel steps, none across two images
  steps = (int)((long long)a.N * a.OH * a.OW / 64);
  tiles = a.KH * a.KW * (a.C / 256) * (a.OC / 256);
  return tiles <= 256 && steps >= g_tune.wgrad8p_min_steps;
}

// n (2 .. EMRT_MAX_WGROUP8) problems that passed wgrad8p_group_ok, none of them sharing its dw with another problem of the call.  Plan: ONE block length
// T (steps) for the whole batch (chosen below); problem i is cut into ceil(steps_i / T) pixel slices.  One slice and a
// dW known to be zero: the block stores its tile; one slice otherwise: atomics; several slices: partial tiles in the registered scratch (while it lasts:
// 256 tiles) + ONE grouped reduce launch, else atomics.
static int launch_wgrad8p_group(const WgradArgs* a, int n, hipStream_t st) {
  int tiles[EMRT_MAX_WGROUP8], steps[EMRT_MAX_WGROUP8], S[EMRT_MAX_WGROUP8], per[EMRT_MAX_WGROUP8], order[EMRT_MAX_WGROUP8];
  long long work = 0;
  for (int i = 0; i < n; ++i) {
    tiles[i] = a[i].KH * a[i].KW * (a[i].C / 256) * (a[i].OC / 256);
    steps[i] = (int)((long long)a[i].N * a[i].OH * a[i].OW / 64);
    work += (long long)tiles[i] * steps[i];
  }
  const int min_steps = g_tune.wgrad8p_min_steps > 0 ? g_tune.wgrad8p_min_steps : 8;
  const double c0 = 8.0;             // a block's prologue + 256 KiB epilogue in steps (tools/bench_conv.py wgroup8)
  // One block per CU (128 KiB of LDS), XCD k's 32 CUs take the blocks of range k in order: a plan's time is the longest XCD's greedy schedule, and a
  // block count just above a multiple of 256 costs a whole extra round.  The block length T is therefore CHOSEN BY SIMULATING the candidates (a dozen
  // schedules of a few hundred blocks: microseconds of host time) instead of aimed at a block count; multi-slice plans also pay for their partial tiles'
  // trip through the slab (the grouped reduce launch).
  auto plan = [&](long long Tt, bool commit) -> double {
    int S_[EMRT_MAX_WGROUP8], per_[EMRT_MAX_WGROUP8], ord[EMRT_MAX_WGROUP8];
    long long slab_tiles = 0;
    double cost = 0.0;
    for (int i = 0; i < n; ++i) {
      int sl = (int)((steps[i] + Tt - 1) / Tt);
      if (sl > steps[i] / min_steps) sl = steps[i] / min_steps;
      if (sl < 1) sl = 1;
      per_[i] = (steps[i] + sl - 1) / sl;
      S_[i] = (steps[i] + per_[i] - 1) / per_[i];
      ord[i] = i;
      if (S_[i] > 1) slab_tiles += (long long)S_[i] * tiles[i];
      cost += (double)tiles[i] * S_[i] * (per_[i] + c0);
    }
    for (int i = 1; i < n; ++i)      // longest blocks first
      for (int j = i; j > 0 && per_[ord[j]] > per_[ord[j - 1]]; --j) { const int t_ = ord[j]; ord[j] = ord[j - 1]; ord[j - 1] = t_; }
    if (commit) {
      for (int i = 0; i < n; ++i) { S[i] = S_[i]; per[i] = per_[i]; order[i] = ord[i]; }
      return 0.0;
    }
    // greedy schedule of the blocks in order, cut into 8 ranges of equal cost, 32 CUs each
    double makespan = 0.0, fin[32], acc = 0.0;
    int k = 0, q = 0, left = n > 0 ? tiles[ord[0]] * S_[ord[0]] : 0;
    for (int c = 0; c < 32; ++c) fin[c] = 0.0;
    while (q < n) {
      if (left == 0) { if (++q < n) left = tiles[ord[q]] * S_[ord[q]]; continue; }
      const double len = per_[ord[q]] + c0;
      if (k < 7 && acc + 0.5 * len > cost * (k + 1) / 8.0) {      // this block opens the next XCD's range
        for (int c = 0; c < 32; ++c) { makespan = fin[c] > makespan ? fin[c] : makespan; fin[c] = 0.0; }
        ++k;
        continue;
      }
      int cmin = 0;
      for (int c = 1; c < 32; ++c) cmin = fin[c] < fin[cmin] ? c : cmin;
      fin[cmin] += len;
      acc += len;
      --left;
    }
    for (int c = 0; c < 32; ++c) makespan = fin[c] > makespan ? fin[c] : makespan;
    return makespan + (slab_tiles > 0 ? 3.0 + 0.03 * (double)slab_tiles : 0.0);      // reduce launch: ~6 us + 16 GB/s-equivalents per tile, in 2-us steps
  };
  long long Tbest = 0;
  // (the same batches come back every step: the chosen T of the last few (tiles, steps) signatures is remembered -- eager steps pay the simulation once)
  static unsigned long long memo_key[16];
  static int memo_T[16], memo_n = 0;
  unsigned long long key = 1469598103934665603ull ^ (unsigned long long)min_steps;
  for (int i = 0; i < n; ++i) key = (key ^ (unsigned long long)(unsigned)(tiles[i] * 65536 + steps[i])) * 1099511628211ull;
  int memo_hit = -1;
  for (int m = 0; m < (memo_n < 16 ? memo_n : 16); ++m) if (memo_key[m] == key) memo_hit = m;
  if (g_tune.wgroup8_blocks > 0) {     // developer knob: aim at a block count (tools/bench_conv.py wgroup8)
    Tbest = (work + g_tune.wgroup8_blocks - 1) / g_tune.wgroup8_blocks;
  } else if (memo_hit >= 0) {
    Tbest = memo_T[memo_hit];
  } else {
    double best = 1e30;
    int smax = 1;
    for (int i = 0; i < n; ++i) smax = steps[i] > smax ? steps[i] : smax;
    static const int cand[] = {64, 96, 128, 160, 192, 208, 224, 240, 256, 288, 320, 384, 448, 512, 640, 768};
    for (int ci = -1; ci < (int)(sizeof(cand) / sizeof(cand[0])); ++ci) {
      long long Tt = ci < 0 ? smax : (work + cand[ci] - 1) / cand[ci];
      if (Tt < min_steps) Tt = min_steps;
      const double t = plan(Tt, false);
      if (t < best) { best = t; Tbest = Tt; }
    }
    memo_key[memo_n & 15] = key; memo_T[memo_n & 15] = (int)Tbest; ++memo_n;
  }
  if (Tbest < min_steps) Tbest = min_steps;
  (void)plan(Tbest, true);
  const bool slab_ok = g_tune.wgrad8p_slab && g_scratch.ptr && g_scratch.stream == (void*)st;
  long long slab_tiles_left = slab_ok ? (long long)(g_scratch.bytes / 262144) : 0, slab_off = 0;
  Wgrad8pGroupArgs g;
  Wgrad8pReduceGroupArgs r;
  r.n = 0; r.tfirst[0] = 0;
  long long total = 0;
  double cost = 0.0, cum[EMRT_MAX_WGROUP8 + 1];
  for (int q = 0; q < n; ++q) {
    const int i = order[q];
    const WgradArgs& s = a[i];
    Wgrad8pArgs& w = g.w[q];
    w.x = s.x; w.dy = s.dy; w.dw = s.dw; w.dbias = s.dbias;
    w.N = s.N; w.H = s.H; w.W = s.W; w.C = s.C; w.ldx = s.ldx; w.x_bs = s.x_bs;
    w.OH = s.OH; w.OW = s.OW; w.OC = s.OC; w.lddy = s.lddy; w.dy_bs = s.dy_bs;
    w.KH = s.KH; w.KW = s.KW; w.stride = s.stride; w.pad = s.pad; w.dil = s.dil;
    w.steps_per_split = per[i]; w.steps_total = steps[i];
    w.tiles_k = s.KH * s.KW * (s.C / 256); w.tiles_oc = s.OC / 256; w.S = S[i]; w.xcd_aware = 1;
    w.slab = nullptr; w.direct = 0;
    if (S[i] == 1) {
      w.direct = (s.overwrite && !g_tune.wgrad_no_overwrite) ? 1 : 0;
    } else if ((long long)S[i] * tiles[i] <= slab_tiles_left) {
      w.slab = (float*)g_scratch.ptr + slab_off * 65536ll;
      slab_off += (long long)S[i] * tiles[i];
      slab_tiles_left -= (long long)S[i] * tiles[i];
      r.slab[r.n] = w.slab; r.dw[r.n] = s.dw; r.S[r.n] = S[i]; r.tiles_k[r.n] = w.tiles_k; r.tiles_oc[r.n] = w.tiles_oc; r.K[r.n] = s.KH * s.KW * s.C;
      r.tfirst[r.n + 1] = r.tfirst[r.n] + tiles[i];
      ++r.n;
    }
    g.first[q] = (int)total;
    cum[q] = cost;
    total += (long long)tiles[i] * S[i];
    cost += (double)tiles[i] * S[i] * (per[i] + c0);
  }
  cum[n] = cost;
  for (int q = n; q <= EMRT_MAX_WGROUP8; ++q) g.first[q] = (int)total;
  for (int q = n; q < EMRT_MAX_WGROUP8; ++q) g.w[q] = g.w[0];
  for (int q = r.n; q < EMRT_MAX_WGROUP8; ++q) { r.slab[q] = nullptr; r.dw[q] = nullptr; r.S[q] = 1; r.tiles_k[q] = 1; r.tiles_oc[q] = 1; r.K[q] = 0; r.tfirst[q + 1] = r.tfirst[r.n]; }
  g.n = n;
  // 8 contiguous ranges of about equal cost: range k starts at the first work item whose cumulative cost reaches k / 8 of the total
  int longest = 0;
  g.xfirst[0] = 0; g.xfirst[8] = (int)total;
  for (int k = 1; k < 8; ++k) {
    const double want = cost * k / 8.0;
    int q = 0;
    while (q + 1 < n && cum[q + 1] <= want) ++q;
    const int i = order[q];
    const double unit = per[i] + c0;
    long long wq = g.first[q] + (long long)((want - cum[q]) / unit + 0.5);
    if (wq > g.first[q + 1]) wq = g.first[q + 1];
    if (wq < g.xfirst[k - 1]) wq = g.xfirst[k - 1];
    g.xfirst[k] = (int)wq;
  }
  for (int k = 0; k < 8; ++k) longest = g.xfirst[k + 1] - g.xfirst[k] > longest ? g.xfirst[k + 1] - g.xfirst[k] : longest;
  if (longest < 1) return 0;
  auto kern = wgrad8p_group_kernel;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072) != hipSuccess)
      return fail("emrt_conv2d_wgrad_group", "cannot raise the dynamic LDS limit to 128 KiB");
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(8 * longest), dim3(512), 131072, st, g);
  int rc = check_launch("emrt_conv2d_wgrad_group(8-phase)");
  if (rc || r.n == 0) return rc;
  hipLaunchKernelGGL(wgrad8p_reduce_group_kernel, dim3(64, r.tfirst[r.n]), dim3(256), 0, st, r);
  return check_launch("emrt_conv2d_wgrad_group(8-phase reduce)");
}
