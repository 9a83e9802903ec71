// 256 x 256 implicit-GEMM convolution (forward / data gradient) for the LARGE layers, bf16 / fp16 -- included by conv.hip.
//
// Replaces, for UpHead's 3x3 convolutions at H/2 and H/4 and the cls_psp / big data-gradient GEMMs (paddle_EMRT.py:134-138,
// 164-180,201-209), the 128 x 128 register-staged tile: that one moves 64 B/clk/CU of operands at the MFMA rate -- the whole
// texture path -- and sat at 35 % MFMA busy (profiles/r2_pmc_conv_uphead128.txt).  Structure (CDNA4 guide, "8-phase" GEMM):
//   * block = 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = 4 x 2 MFMA 32x32 accumulators; BK = 64: half the operand bytes
//     per FLOP of the 128 x 128 tile;
//   * operands go global -> LDS directly (`buffer_load_dwordx4 ... lds`, 1 KiB per wave instruction), no register ring:
//     padding taps / stride holes / ragged M, OC tails are lanes whose byte offset has bit 31 set: out of the descriptor's
//     range, the DMA writes zeros for them;
//   * two 64 KiB k-tile buffers; a k-tile is cut into four 16 KiB UNITS by the phase that reads them
//       UA0 = A rows {0-63,128-191}  UB0 = B rows {64 wc + 0..31}   (phase 1)     UB1 = B rows {64 wc + 32..63} (phase 2)
//       UA1 = A rows {64-127,192-255}                                  (phase 3)     UB0 again                      (phase 4)
//     and every phase re-stages ONE unit (2 DMA instructions per wave) of a later k-tile into the slot whose last reader was
//     the previous phase:  ph2: UA0(t+2)  ph3: UB1(t+2)  ph4: UA1(t+2)  ph1(t+1): UB0(t+2);
//   * a phase = { LDS fragment reads + the unit's DMA issue ; s_waitcnt lgkmcnt(0) ; s_barrier ; 8 MFMAs (one 64 x 32 quadrant,
//     K = 64) ; s_barrier }, waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave issues MFMAs while its
//     partner reads LDS / issues DMA;
//   * the DMA is waited for ONCE per k-tile with a COUNTED s_waitcnt vmcnt(6) in phase 4 (three units stay in flight across
//     the barriers); raw s_barrier, never __syncthreads() in the loop (its fence would drain the DMA queue).
// Hazards (who may touch an LDS byte when), with I(g) = the barrier interval in which waves 0-3 run the load half of phase g
// (waves 4-7 run it in I(g) + 1):
//   WAR  a unit read in phase g is retired by the lgkmcnt(0) BEFORE that phase's first barrier, i.e. inside I(g) / I(g)+1; its
//        re-stage is issued in phase g+1, i.e. in I(g)+2 / I(g)+3: at least one barrier later for either group;
//   RAW  tile t+1's last unit is issued in ph1(t); every wave's vmcnt(6) in ph4(t) sits before that phase's first barrier and
//        leaves only the three units issued in ph2-4(t) (tile t+2) in flight; tile t+1 is first read in ph1(t+1), one barrier later.
// K tiles past the end are staged as all-out-of-range (zeros, never read): the loop has no special tail.
#pragma once

#ifndef EMRT_8P_ASM_DMA
#define EMRT_8P_ASM_DMA 1
#endif

typedef __attribute__((address_space(3))) void emrt_lds_void;

// one LDS-DMA wave instruction: 64 lanes x 16 bytes from (descriptor, per-lane byte offset) to LDS [lds_addr, lds_addr + 1 KiB)
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned lds_addr, unsigned char* smem) {
#if EMRT_8P_ASM_DMA
  // hidden from hipcc's wait-count bookkeeping on purpose: beside a builtin LDS-DMA the compiler guards later ds_reads with
  // vmcnt(0); the loop counts its DMA queue by hand.  M0 (the LDS destination) is written in the statement that uses it.
  (void)smem;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_addr), "s"(rs) : "memory");
#else
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (emrt_lds_void*)(smem + lds_addr), 16, (int)voff, 0, 0, 0);
#endif
}

// PROBE (timing experiments only, -DEMRT_8P_PROBES builds; results are WRONG): 1 = no DMA issue in the loop, 2 = no fragment reads,
// 4 = no MFMAs, 8 = no global stores / loads in the epilogue, 16 = no epilogue at all (kills most MFMAs too: only valid for the 1-k-tile
// shapes).  The production instantiation is PROBE = 0.
// Measured and not kept (round 3, tools/bench_conv.py big): a second schedule that keeps the b0 fragments in registers through phase 4
// (20 instead of 24 fragment reads per k-tile), re-stages every unit TWO phases after its last read (ph1: UB1(t+1), ph2: UA1(t+1),
// ph3: UA0(t+2), ph4: UB0(t+2), vmcnt(4)) and retires the reads AFTER the first barrier: within 1 % of this one on every shape
// (147.4 vs 146.6 us on UpHead conv_2) -- the loop is not bound by the fragment reads' latency.
template <class T, int MODE, int PROBE = 0>
__global__ __launch_bounds__(512, 2) void igemm8p_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // 2 x 64 KiB k-tile buffers; the epilogue reuses them
  static_assert(sizeof(T) == 2, "bf16 / fp16 only");
  constexpr int BM = 256, BN = 256;
  constexpr unsigned BUFB = 65536u, AB = 32768u;      // bytes per k-tile buffer; offset of the B tile inside it
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;            // SIMD partners differ in wr (waves w and w + 4 share a SIMD)
  const int tiles_n = (p.OC + BN - 1) / BN;
  int bid = (int)blockIdx.x;
  const int nblk = (int)gridDim.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);       // neighbouring tiles on one XCD (shared L2)
  const int bm = bid / tiles_n, bn = bid % tiles_n;
  const int OHW = p.OH * p.OW;
  const long long M = (long long)p.N * OHW;
  const int K = p.KH * p.KW * p.C;
  const int nkt = K >> 6;                              // the host guarantees C % 64 == 0: a k-tile never straddles a tap
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)BUF_RANGE, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)BUF_RANGE, 0x00020000);

  // ---- loader state: this thread's 4 A rows, 4 B rows and its (swizzled) 16-byte chunk of the k-tile ----------------------
  // LDS image: row r at r * 128 B, chunk c of the row at position c ^ ((r >> 1) & 7): a 32x32x16 fragment read (32 rows x one
  // chunk per half-wave) then covers all 16 slots of the 256-byte bank row once per 16-lane group.  The DMA writes lanes
  // linearly (lane l -> row l >> 3, position l & 7 of its piece), so the permutation is applied to the SOURCE chunk.
  const int lrow = lane >> 3;
  const unsigned chunk_b = (unsigned)(((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) * 16);   // byte offset inside the k-tile row
  // the weight operand first: its addresses need no division, so its first k-tile is in flight while the pixel decomposition below runs
  unsigned b_off[4];                                   // q = 2 j + u: unit UBu, piece j
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int n = bn * BN + (wave >> 2) * 64 + (wave & 3) * 8 + 128 * (q >> 1) + 32 * (q & 1) + lrow;
    b_off[q] = n < p.OC ? (unsigned)((long long)n * K * 2) : BUF_OOB;
  }
  // K order.  Default: tap-major (all channels of tap 0, then tap 1, ...: consecutive k-tiles read consecutive 128-byte pieces of the same
  // pixel rows and weight rows).  A/B knob igemm8p_cmajor (stride-1 problems): CHANNEL-BLOCK major, k-tile t = (64-channel block t / ntap,
  // tap t % ntap), so the KH*KW taps of one channel block are consecutive k-tiles and their re-reads of the same pixels hit L2.  Measured on
  // UpHead conv_2 (profiles/r3_pmc_conv_8phase.txt): FETCH_SIZE 146 -> 43 MB, i.e. 2 x FETCH + WRITE = 358 -> 151 MB = 2.6x -> 1.11x the
  // 135 MB of algorithmic bytes -- and 141 -> 147 us; with C = 1536 (cls_psp.0) 231 -> 272 us.  The tap-major order's extra fetches are
  // re-reads of a 67 MB tensor that sits in the 256 MB Infinity Cache (FETCH_SIZE counts fabric requests, MALL hits included), and its
  // sequential streams cost less than the strided ones that buy the L2 hits: the faster order stays the default.
  const int ntap = p.KH * p.KW;
  const bool cmajor = p.cmajor != 0 && p.stride == 1 && ntap <= 32;
  int bkt = 0, b_tap = 0, b_c0 = 0;
  auto advance_b = [&]() {
    ++bkt;
    if (cmajor) {
      ++b_tap;
      if (b_tap == ntap) { b_tap = 0; b_c0 += 64; }
    }
  };
  auto stage_b = [&](int u, int par) {                 // unit UBu of k-tile bkt -> buffer par
    if constexpr ((PROBE & 1) != 0) { if (bkt >= 2) return; }
    const unsigned kbad = bkt < nkt ? 0u : BUF_OOB;
    const unsigned kb = (cmajor ? (unsigned)(b_tap * p.C + b_c0) * 2u : (unsigned)bkt * 128u) + chunk_b;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row0 = (wave >> 2) * 64 + (wave & 3) * 8 + 128 * j + 32 * u;
      lds_dma16(rs_w, (b_off[2 * j + u] | kbad) + kb, (unsigned)par * BUFB + AB + (unsigned)row0 * 128u, smem);
    }
  };

  stage_b(0, 0); stage_b(1, 0);                        // prologue, part 1 (k-tile 0 of the weights)
  int a_h[4], a_w[4];
  unsigned a_base[4];
  bool a_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {                        // i = 0: UA0 rows 0-63, 1: UA1 64-127, 2: UA0 128-191, 3: UA1 192-255
    const unsigned m = (unsigned)bm * (unsigned)BM + (unsigned)(64 * i + 8 * wave + lrow);      // < 2^31 (host-checked): 32-bit divisions
    a_ok[i] = (long long)m < M;
    const unsigned mm = a_ok[i] ? m : 0u;
    const int nb = (int)(mm / (unsigned)OHW);
    const int r = (int)(mm - (unsigned)nb * (unsigned)OHW);
    const int oh = (int)((unsigned)r / (unsigned)p.OW), ow = r - oh * p.OW;
    a_base[i] = (unsigned)((long long)nb * p.in_bs * 2);
    if (MODE == 0) { a_h[i] = oh * p.stride - p.pad; a_w[i] = ow * p.stride - p.pad; }
    else { a_h[i] = oh + p.pad; a_w[i] = ow + p.pad; }
  }
  auto a_pixel = [&](int i, int kh, int kw) -> unsigned {
    int hi, wi;
    bool ok = a_ok[i];
    if (MODE == 0) { hi = a_h[i] + kh * p.dil; wi = a_w[i] + kw * p.dil; }
    else {
      const int th = a_h[i] - kh * p.dil, tw = a_w[i] - kw * p.dil;
      if (p.stride == 1) { hi = th; wi = tw; }
      else if (p.stride == 2) { hi = th >> 1; wi = tw >> 1; ok = ok && ((th | tw) & 1) == 0; }
      else {
        hi = th / p.stride; wi = tw / p.stride;
        ok = ok && th >= 0 && tw >= 0 && (hi * p.stride == th) && (wi * p.stride == tw);
      }
    }
    ok = ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
    const unsigned off = a_base[i] + (unsigned)((hi * p.W + wi) * p.ldin) * 2u;
    return ok ? off : BUF_OOB;
  };
  // cursors of the NEXT k-tile to stage, one for the A units and one for the B units (they advance at different phases)
  int akt = 0, a_c0 = 0, a_kh = 0, a_kw = 0;
  unsigned a_cur[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_cur[i] = a_pixel(i, 0, 0);
  const bool one_tap = ntap == 1;
  // channel-block-major order: the tap changes with every k-tile, so a row's tap offsets must be cheap: offset = rowbase + tapd (tapd is the
  // same for every row: scalar) where the tap is inside the map, a bit of vmask says where that is
  unsigned rowbase[4], vmask[4];
  if (cmajor) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      rowbase[i] = a_base[i] + (unsigned)((a_h[i] * p.W + a_w[i]) * p.ldin) * 2u;      // (wraps for border rows: only used where vmask says valid)
      unsigned vh = 0u, vw = 0u;
      for (int kh = 0; kh < p.KH; ++kh) vh |= ((unsigned)(MODE == 0 ? a_h[i] + kh * p.dil : a_h[i] - kh * p.dil) < (unsigned)p.H ? 1u : 0u) << kh;
      for (int kw = 0; kw < p.KW; ++kw) vw |= ((unsigned)(MODE == 0 ? a_w[i] + kw * p.dil : a_w[i] - kw * p.dil) < (unsigned)p.W ? 1u : 0u) << kw;
      unsigned vm = 0u;
      for (int kh = 0; kh < p.KH; ++kh) vm |= ((vh >> kh) & 1u) ? vw << (kh * p.KW) : 0u;
      vmask[i] = a_ok[i] ? vm : 0u;
    }
  }
  int a_tap = 0;
  auto advance_a = [&]() {
    ++akt;
    if (cmajor) {
      ++a_tap; ++a_kw;
      if (a_kw == p.KW) { a_kw = 0; ++a_kh; }
      if (a_tap == ntap) { a_tap = 0; a_kh = 0; a_kw = 0; a_c0 += 64; }
      if (!one_tap) {
        const int d = (a_kh * p.dil * p.W + a_kw * p.dil) * p.ldin * 2;
        const unsigned tapd = (unsigned)(MODE == 0 ? d : -d);
#pragma unroll
        for (int i = 0; i < 4; ++i) a_cur[i] = ((vmask[i] >> a_tap) & 1u) ? rowbase[i] + tapd : BUF_OOB;
      }
      return;
    }
    a_c0 += 64;
    if (a_c0 >= p.C) {
      a_c0 = 0;
      if (!one_tap) {
        ++a_kw;
        if (a_kw == p.KW) { a_kw = 0; ++a_kh; }
#pragma unroll
        for (int i = 0; i < 4; ++i) a_cur[i] = a_pixel(i, a_kh, a_kw);      // (past the last tap: masked by akt >= nkt)
      }
    }
  };
  auto stage_a = [&](int u, int par) {                 // unit UAu of k-tile akt -> buffer par
    if constexpr ((PROBE & 1) != 0) { if (akt >= 2) return; }
    const unsigned kbad = akt < nkt ? 0u : BUF_OOB;
    const unsigned cb = (unsigned)a_c0 * 2u + chunk_b;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int i = u + 2 * j;
      lds_dma16(rs_in, (a_cur[i] | kbad) + cb, (unsigned)par * BUFB + (unsigned)(64 * i + 8 * wave) * 128u, smem);
    }
  };
  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- fragment addresses: lane (frow, fh) reads row frow of a 32-row tile, chunk 2 s + fh of k-step s ----------------------
  const int frow = lane & 31, fh = lane >> 5;
  unsigned fpos[4];                                    // byte offset of (row frow, chunk 2 s + fh) inside a 32-row tile image
#pragma unroll
  for (int s = 0; s < 4; ++s) fpos[s] = (unsigned)(frow * 128 + (((2 * s + fh) ^ ((frow >> 1) & 7)) * 16));
  const unsigned a_tile0 = (unsigned)(wr * 128) * 128u;             // this wave's first A row
  const unsigned b_tile0 = AB + (unsigned)(wc * 64) * 128u;         // ... first B row
  uint4 fa[2][4], fb0[4], fb1[4];
  auto lds16 = [&](unsigned off) -> uint4 {
    if constexpr ((PROBE & 2) != 0) { uint4 v = make_uint4(off, off * 3u, 0x3f803f80u, 0x3f803f80u); asm volatile("" : "+v"(v.x), "+v"(v.y)); return v; }
    return *reinterpret_cast<const uint4*>(smem + off);
  };

  // ---- prologue: k-tile 0 whole, then the units of k-tile 1 that the last phases of "tile -1" would have issued ----------------
  stage_a(0, 0); stage_a(1, 0);                        // (the weights' half of k-tile 0 was issued above)
  advance_a(); advance_b();
  stage_a(0, 1); stage_b(1, 1); stage_a(1, 1);        // UB0(1) follows in phase 1 of tile 0 (bkt advances there)
  advance_a();
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // k-tile 0 has landed (this wave's pieces; the barrier covers the others)
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();          // waves 4-7 run one barrier behind from here on

  // one phase.  Q: quadrant phase 1..4, PAR: buffer of the k-tile being multiplied.
  auto phase = [&](auto Qc, auto PARc) {
    constexpr int Q = decltype(Qc)::value, PAR = decltype(PARc)::value;
    const unsigned buf = (unsigned)PAR * BUFB;
    if constexpr (Q == 1) {
#pragma unroll
      for (int s = 0; s < 4; ++s) fb0[s] = lds16(buf + b_tile0 + fpos[s]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 4; ++s) fa[i][s] = lds16(buf + a_tile0 + (unsigned)i * 4096u + fpos[s]);
      stage_b(0, PAR ^ 1);                              // UB0 of the NEXT k-tile (its slot's last reader: phase 4 of the tile before)
      advance_b();
    } else if constexpr (Q == 2) {
#pragma unroll
      for (int s = 0; s < 4; ++s) fb1[s] = lds16(buf + b_tile0 + 4096u + fpos[s]);
      stage_a(0, PAR);                                  // UA0 of the k-tile after next, into THIS tile's buffer (read in phase 1)
    } else if constexpr (Q == 3) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 4; ++s) fa[i][s] = lds16(buf + a_tile0 + 8192u + (unsigned)i * 4096u + fpos[s]);
      stage_b(1, PAR);                                  // UB1 (read in phase 2)
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) fb0[s] = lds16(buf + b_tile0 + fpos[s]);
      stage_a(1, PAR);                                  // UA1 (read in phase 3)
      advance_a();
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // everything but the last three units: the next k-tile is complete
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this phase's reads are retired BEFORE the barrier (WAR rule above)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
    constexpr int AH = (Q >= 3) ? 1 : 0, BH = (Q == 2 || Q == 3) ? 1 : 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if constexpr ((PROBE & 4) != 0) { asm volatile("" :: "v"(fa[i][s].x), "v"(fb0[s].x), "v"(fb1[s].x)); }
        else mma_chunk<T>(acc[AH * 2 + i][BH], fa[i][s], BH ? fb1[s] : fb0[s]);
      }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>;
  for (int kt = 0; kt < nkt; kt += 2) {                // (an odd tail multiplies one all-zero tile)
    phase(I1{}, I0{}); phase(I2{}, I0{}); phase(I3{}, I0{}); phase(I4{}, I0{});
    phase(I1{}, I1{}); phase(I2{}, I1{}); phase(I3{}, I1{}); phase(I4{}, I1{});
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();           // re-align the two wave groups
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the trailing (all-zero) stages must land before the epilogue reuses the LDS
  __syncthreads();

  if constexpr ((PROBE & 16) != 0) {
    if (acc[0][0][0] == 123.456f) ((float*)p.out)[tid] = acc[1][1][3] + acc[2][0][5] + acc[3][1][7];      // keeps the accumulators alive
    return;
  }
  // ---- epilogue: every wave on its own, through a PRIVATE 32 x 64 fp32 LDS tile, one M-tile (32 rows) at a time ---------------------
  // The C/D layout of the 32x32 MFMA gives a lane one output COLUMN; rows of 8 consecutive channels per lane (16-byte stores, 128-byte
  // runs per row) need a transposition, and doing it per wave needs no block barrier: LDS accesses of one wave execute in order, so the
  // eight waves drift apart and their LDS / store phases overlap (the first version ran four block-wide passes of 64 rows with two
  // __syncthreads() each: ~7 of the kernel's ~16 us of fixed cost).  Same arithmetic and order per element as igemm_body's row-vectorised
  // epilogue (the host only sends problems that satisfy its vector conditions): accumulate * scale + bias -> + residual -> ReLU -> mask ->
  // store -> statistics of the stored value.
  constexpr int WP = 68;                               // floats per row of the wave tile: 16-byte aligned rows, write side conflict-free
  float* tile = reinterpret_cast<float*>(smem) + wave * (32 * WP);
  const int cg = lane & 7, rl = lane >> 3;             // 8-channel group of the wave's 64 columns, row lane (8 rows per step)
  const int n0 = bn * BN + wc * 64 + cg * 8;
  const bool col_ok = n0 < p.OC;
  float bv[8], sv[8], ss[8], sq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { bv[e] = (p.bias && col_ok) ? p.bias[n0 + e] : 0.f; sv[e] = (p.scale && col_ok) ? p.scale[n0 + e] : 1.f; ss[e] = 0.f; sq[e] = 0.f; }
  const T* resp = (const T*)p.res;
  const T* ymask = (const T*)p.mask_y;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        tile[((r & 3) + 8 * (r >> 2) + 4 * fh) * WP + j * 32 + frow] = acc[i][j][r];
    // same wave: the LDS queue executes the reads below behind these writes; the wave barrier (no instruction) only keeps the COMPILER
    // from moving a lane's reads above other lanes' writes, which it cannot see as a dependence
    __builtin_amdgcn_wave_barrier();
#pragma unroll 2
    for (int st = 0; st < 4; ++st) {
      const int tr = st * 8 + rl;
      const unsigned m = (unsigned)bm * (unsigned)BM + (unsigned)(wr * 128 + i * 32 + tr);
      float v[8];
      {
        const float4 a = *reinterpret_cast<const float4*>(tile + tr * WP + cg * 8);
        const float4 b = *reinterpret_cast<const float4*>(tile + tr * WP + cg * 8 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      }
      if ((long long)m >= M || !col_ok) continue;
      const int e_nb = (int)(m / (unsigned)OHW);
      const int e_pix = (int)(m - (unsigned)e_nb * (unsigned)OHW);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], sv[e], bv[e]);
      if (resp && !(PROBE & 8)) {
        float w8[8];
        Vec8<T>::load(resp + (long long)e_nb * p.res_bs + (long long)e_pix * p.ldres + n0, w8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += w8[e];
      }
      if (p.relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      float second[8];
      if (ymask) {
        Vec8<T>::load(ymask + (long long)e_nb * p.y_bs + (long long)e_pix * p.ldy + n0, second);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = second[e] > 0.f ? v[e] * p.mask_scale : 0.f;
        if (p.stat_x) Vec8<T>::load((const T*)p.stat_x + (long long)e_nb * p.sx_bs + (long long)e_pix * p.ldsx + n0, second);
      }
      const long long obase = (long long)e_nb * p.out_bs + (long long)e_pix * p.ldout + n0;
      if constexpr ((PROBE & 8) != 0) {
        if (v[0] == 123.456f) Vec8<T>::store((T*)p.out + obase, v);
      } else if (p.out_f32) {
        Vec8<float>::store((float*)p.out + obase, v);
      } else {
        // NON-TEMPORAL: this kernel only runs on outputs of tens of MB (>= 160 blocks x 128 KiB) that every block writes at the same moment;
        // streaming them past L2 measured -4 % on UpHead conv_2 and -15 % on the kernel's fixed cost (profiles/r3_conv_8phase_probes.txt)
        typedef __attribute__((ext_vector_type(4))) unsigned int nt_u32x4;
        nt_u32x4 pk;
        if constexpr (std::is_same<T, bf16_t>::value) pk = nt_u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
        else pk = nt_u32x4{pack_f16x2(v[0], v[1]), pack_f16x2(v[2], v[3]), pack_f16x2(v[4], v[5]), pack_f16x2(v[6], v[7])};
        __builtin_nontemporal_store(pk, reinterpret_cast<nt_u32x4*>((T*)p.out + obase));
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = to_f32(from_f32<T>(v[e]));       // statistics of what the next kernel will read
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ss[e] += v[e];
        sq[e] = fmaf(v[e], ymask ? second[e] : v[e], sq[e]);
      }
    }
    __builtin_amdgcn_wave_barrier();                   // the next M-tile's writes stay behind this tile's reads
  }
  if (p.stats) {
    // column sums: the 8 row lanes of a column group by shuffles, the two wave rows (wr) through LDS, ONE atomic per column and statistic
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) { ss[e] += __shfl_xor(ss[e], o, 64); sq[e] += __shfl_xor(sq[e], o, 64); }
    }
    __syncthreads();                                   // every wave is done with its private tile
    float* red = reinterpret_cast<float*>(smem);      // [2 wr][2 statistics][256 columns]
    if (rl == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(wr * 2 + 0) * BN + wc * 64 + cg * 8 + e] = ss[e];
        red[(wr * 2 + 1) * BN + wc * 64 + cg * 8 + e] = sq[e];
      }
    }
    __syncthreads();
    const int which = tid >> 8, col = tid & 255;
    const int n = bn * BN + col;
    const float a = red[which * BN + col] + red[(2 + which) * BN + col];
    if (n < p.OC) atomicAdd(p.stats + (long long)(bm & 7) * 2 * p.OC + (long long)which * p.OC + n, (double)a);
  }
}

// host side: can this problem take the 256 x 256 kernel?  (vector operand path, whole k-tiles per tap, vector epilogue)
template <class T>
static bool igemm8p_ok(const ConvArgs& a) {
  if (sizeof(T) != 2 || a.drop_seed) return false;      // (the dropout epilogue lives in igemm_body only)
  if (a.C % 64 != 0 || a.OC % 8 != 0) return false;
  if (a.ldin % 8 || a.in_bs % 8 || ((uintptr_t)a.in) % 16 || ((uintptr_t)a.w) % 16) return false;
  const int eo = a.out_f32 ? 4 : 8;
  if (((uintptr_t)a.out) % 16 || a.ldout % eo || a.out_bs % eo) return false;
  if (a.res && (((uintptr_t)a.res) % 16 || a.ldres % 8 || a.res_bs % 8)) return false;
  if (a.mask_y && (((uintptr_t)a.mask_y) % 16 || a.ldy % 8 || a.y_bs % 8)) return false;
  if (a.stat_x && (((uintptr_t)a.stat_x) % 16 || a.ldsx % 8 || a.sx_bs % 8)) return false;
  return true;
}

template <class T, int MODE, int PROBE = 0>
static int launch_igemm8p(const ConvArgs& a, hipStream_t st) {
  const long long M = (long long)a.N * a.OH * a.OW;
  const long long grid = ((M + 255) / 256) * ((a.OC + 255) / 256);
#ifdef EMRT_8P_PROBES
  if constexpr (PROBE == 0 && MODE == 0 && std::is_same<T, bf16_t>::value) {
    switch (g_tune.igemm8p_probe) {
      case 1: return launch_igemm8p<T, MODE, 1>(a, st);
      case 2: return launch_igemm8p<T, MODE, 2>(a, st);
      case 3: return launch_igemm8p<T, MODE, 3>(a, st);
      case 4: return launch_igemm8p<T, MODE, 4>(a, st);
      case 6: return launch_igemm8p<T, MODE, 6>(a, st);
      case 7: return launch_igemm8p<T, MODE, 7>(a, st);
      case 8: return launch_igemm8p<T, MODE, 8>(a, st);
      case 16: return launch_igemm8p<T, MODE, 16>(a, st);
      default: break;
    }
  }
#endif
  auto kern = igemm8p_kernel<T, MODE, PROBE>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072) != hipSuccess)
      return fail("emrt_conv2d", "cannot raise the dynamic LDS limit to 128 KiB");
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), 131072, st, a);
  return check_launch("emrt_conv2d");
}
