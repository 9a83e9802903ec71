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

__global__ __launch_bounds__(512, 2) void wgrad8p_kernel(Wgrad8pArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr unsigned BUFB = 65536u, AB = 32768u;      // bytes per step buffer; offset of the x tile inside it
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int cblks = p.C >> 8, ntap = p.KH * p.KW;
  // work item of this block (see the header); integer divisions run on the vector ALU: their wave-uniform results go back to scalar registers
  int bz, by, kt, tap, c0;
  {
    const int tiles = p.tiles_k * p.tiles_oc, nb = tiles * p.S;
    const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
    const int w0 = (int)(((long long)xcd * nb) >> 3), w1 = (int)(((long long)(xcd + 1) * nb) >> 3);
    const int w = p.xcd_aware ? w0 + idx : (int)blockIdx.x;      // (0: A/B knob wgrad8p_xcd, work items in launch order)
    if (w >= (p.xcd_aware ? w1 : nb)) return;          // (the grid is 8 x the longest range)
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

// dW[oc][tap * C + c] += sum over the pixel slices z of slab[z][oc tile][k tile][oc % 256][k % 256]; grid (64, tiles), one float4 per thread
__global__ __launch_bounds__(256) void wgrad8p_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int S, int tiles_k, int tiles_oc, int K) {
  const long long tile = blockIdx.y;                   // oc_tile * tiles_k + k_tile
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
  w.tiles_k = tiles_k; w.tiles_oc = tiles_oc; w.S = S; w.xcd_aware = g_tune.wgrad8p_xcd;
  const int nb = tiles_k * tiles_oc * S;
  hipLaunchKernelGGL(kern, dim3(8 * ((nb + 7) / 8)), dim3(512), 131072, st, w);
  int rc = check_launch("emrt_conv2d_wgrad(8-phase)");
  if (rc || !w.slab) return rc;
  hipLaunchKernelGGL(wgrad8p_reduce_kernel, dim3(64, tiles_k * tiles_oc), dim3(256), 0, st, (const float*)w.slab, a.dw, S, tiles_k, tiles_oc, a.KH * a.KW * a.C);
  return check_launch("emrt_conv2d_wgrad(8-phase reduce)");
}
