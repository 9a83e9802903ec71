// Shared device/host helpers for the EMRT gfx950 kernels (internal; the public C-ABI is include/emrt_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define EMRT_F32 0
#define EMRT_BF16 1
#define EMRT_F16 2   /* IEEE half storage, fp32 accumulation: inference (forward) entry points only */

namespace emrt {

// ---- error plumbing (thread-local last-error string, see emrt_last_error) --------------------
extern thread_local char g_err[512];
inline int fail(const char* fn, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s: %s", fn, msg);
  return -1;
}
inline int check_launch(const char* fn) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: launch failed: %s", fn, hipGetErrorString(e));
    return -2;
  }
  return 0;
}
#define EMRT_REQUIRE(cond, msg) \
  do {                          \
    if (!(cond)) return emrt::fail(__func__, msg); \
  } while (0)

// ---- tuning knobs (developer / test aids) -----------------------------------------------------
// Read ONCE from the environment when the library is loaded (EMRT_<NAME IN CAPITALS>) and changeable afterwards through
// emrt_set_tuning(): the dispatchers only ever read these plain ints, no getenv() on a launch path.
struct Tuning {
  int conv_tile;        // 0 = the dispatcher's choice; 1..6 force an igemm tile (tools/bench_conv.py)
  int wgrad_split;      // 0 = cost model; > 0 forces the number of pixel-reduction slices
  int thin_cblk;        // thin classifier backward: channels per block (64)
  int thin_blocks;      // ... pixel chunks (128)
  int thin_ch;          // ... channels per thread (8)
  int no_thin_bwd;      // 1 = the classifier backward runs as two GEMMs
  int igemm64_nst;      // 3 (default) / 4: register ring depth of the 64x64-tile igemm loop (A/B knob)
  int wgrad_nst;        // 2 (default) / 3: register ring depth of the weight-gradient loop (A/B knob)
  int pair_max;         // largest dgrad grid that is paired with its wgrad in one launch (768)
  int msda_fwd_global;  // 1 = never use the LDS-staged MSDA forward
  int msda_bwd_global;  // 1 = never use the LDS-staged MSDA gradient kernel
  int msda_lds_min_pairs; // smallest B * M * Lq that takes the LDS-staged MSDA kernels (2048: the decoder's 110 queries too, 29 -> 13 us)
  int msda_bwd_dref_lds; // 1 (default): the LDS-staged gradient kernel also serves calls that want the reference-point gradient (the
                        // decoder): fp32 atomics into the caller-ZEROED dref; 0 = those calls take the global-gather kernel
  int msda_band_halo;   // band MSDA kernels (pyramids too large for one LDS slab): rows staged beyond a band (0 = 7, fewer if LDS is short)
  int msda_scatter_qsplit; // value-gradient scatter: -1 = never split a range's queries over several blocks, 0 = automatic, n > 0 = force n
  int msda_scatter_cuts; // value-gradient scatter: row ranges per level (0 = automatic: about two blocks per CU)
  int msda_fwd_chunks;  // LDS-staged MSDA forward: query chunks per (batch, head) slab (0 = automatic)
  int msda_fwd_threads; // ... threads per block (1024)
  int msda_fwd_probe;   // timing experiments only (results are WRONG): forward 1 = no gather, 2 = no staging, 4 = no preparation;
                        // value-gradient scatter 16 = no |g| scan, 32 = no sample loop (tools/exp/probe_msda_*.sh)
  int bn_block_kb;      // BatchNorm streaming kernels: KB of input per block (8)
  int bn_operand_blocks; // kernels that apply BatchNorm to their input operand (bn_operand.hpp): most blocks per launch (0 = 1024; the classifier kernel 512)
  int ln_bwd_rows;      // LayerNorm backward: rows per block (0 = 32) and most blocks (0 = 512): every block ends with 2C fp32 atomics
  int ln_bwd_max_blocks;
  int ln_bwd_threads;   // LayerNorm backward at C <= 256: 512 (default: 64 rows per block, <= 256 blocks) or 256 threads per block (32 rows, <= 512 blocks)
  int ln_atomic;        // 1 = LayerNorm / column-sum parameter gradients as atomics, 0 = partials + finalize launch
  int gn_group_blocks;  // 1 = multi-level GroupNorm with one block per (image, group) instead of the row-major stats + apply pair
  int gn_stat_rows;     // row-major GroupNorm: token rows per block of the forward statistics launch (32)
  int gn_bwd_stat_rows; // ... of the backward sums launch (32; each block also adds 2 C parameter-gradient atomics)
  int gn_apply_rows;    // ... of the apply / dx launches (8)
  int igemm8p_probe;    // -DEMRT_8P_PROBES builds only: which parts of the 256x256 kernel's loop are switched off (timing experiments)
  int igemm8p_cmajor;   // 256x256 kernel, stride-1 problems: 1 = channel-block-major k order (A/B knob; 1.11x instead of 2.6x the algorithmic HBM-side
                        // traffic, 4-18 % slower: igemm8p.hpp)
  int igemm8p_min_blocks; // smallest 256x256 grid that takes the LDS-DMA 8-phase kernel (160); 0 = never
  int wgrad8p_min_steps;  // 256x256 weight-gradient kernel: fewest 64-pixel steps per block worth its prologue / 256 KiB epilogue (8); 0 = never
  int wgrad8p_slab;       // 1 (default): partial tiles through the registered scratch + a reduce launch; 0: fp32 atomics into dW
  int wgrad8p_xcd;        // 1 (default): slices of the pixel reduction pinned to XCDs (shared L2); 0: launch order (A/B knob)
  int wgrad8p_force;      // tests: 1 = take the 256x256 kernel whenever the shape allows, whatever the grid size
  int no_ksplit128;     // 1 = never the 128x128 tile with two wave groups splitting K (A/B knob)
  int wgrad_no_overwrite; // 1 = weight gradients always accumulate with atomics, even into a dW the caller declared zero (A/B knob)
  int no_s2_dgrad;        // 1 = stride-2 data gradients through the generic kernels (A/B knob; tests compare the two bit for bit); -1 = the parity-class kernel for 1x1 kernels too
  int wgroup_blocks;      // batched weight gradients (emrt_conv2d_wgrad_group): blocks a launch aims for (1024 = 4 per CU)
  int wgroup_min_steps;   // ... fewest 64-pixel tiles per block (32: shorter blocks only buy fp32 atomic traffic)
  int wgroup_max;         // ... most problems per launch (0 = the kernel's limit, 24)
  int wgroup8;            // 1: the problems of a batch that fit the 256x256 LDS-DMA weight-gradient kernel's shape rules go out together on THAT kernel
                          // (wgrad8p_group_kernel) when they amount to wgroup8_min_work block-steps; 0 (default) = all on the 128x128 group kernel.  Measured
                          // (round 6, profiles/r6c_wgroup8.txt): an encoder layer's batch 186.7 -> 174.9 us and layer4's 51.0 -> 47.6 us alone, layer3 and the
                          // heads lose (52.7 -> 64.9, 93.6 -> 122.2 us); in the captured step 989.9 (off) vs 988.4 tiles/s (encoder batches only): not taken
  int wgroup8_blocks;     // ... 0 (default): the block length of the grouped 256x256 launch is chosen by simulating the schedule; n > 0: aim at n blocks (developer knob)
  int wgroup8_min_work;   // ... fewest (output tile x 64-pixel step) units worth a launch of its own (6000: the encoder batches)
  int msda_scatter_merge; // 1 = the value-gradient scatter adds consecutive points of a query with the same 2 x 2 footprint in registers first (A/B knob)
  int msda_scatter_mfma;  // bf16 value gradients of the deformable attention as a matrix product (msda_bwd_value_mfma_kernel): 1 (default) = when at most one level needs two
                          // row bands, 2 = whenever the maps are <= 512 pixels wide, 0 = always the LDS atomic scatter
  int msda_mf_bands;      // ... most pixels of a row band (0 = 512)
  int sgd_nt;             // 1 (default): the optimizer pass streams master weights / velocity / gradients with non-temporal accesses (A/B knob)
  int mha_bwd_split;      // 1 (default): the MFMA softmax-attention backward as two blocks per (batch, head) -- dq | dk, dv (A/B knob)
  int mha_valu;           // 1 = the decoder's softmax attention on the VALU kernels for every dtype (A/B knob; bf16 / fp16 default to the MFMA kernels)
  int no_bna;             // 1 = emrt_conv2d_bna_supported always answers 0: every BatchNorm between convolutions keeps its own emrt_bn_apply launch (A/B knob);
                          // -1 = also the long-k 3x3 layers the dispatcher leaves to the separate launch (tests)
  int memcpy_kernel;      // 1: emrt_memcpy is a copy kernel of the stream; 0 (default): hipMemcpyAsync (A/B knob, measured neutral)
  int xk;                 // cross-block K split of few-tile, long-K convolutions: 0 = the dispatcher's choice, -1 = never, n >= 2 = n copies whenever the shape allows
};
extern Tuning g_tune;

// scratch for partial sums, registered once per process by emrt_set_scratch (kernels of ONE stream use it one after the other)
struct Scratch {
  void* ptr; size_t bytes; void* stream;      // stream: the ONE stream whose launches may use it (partial tiles + reduce are not atomic across streams)
  unsigned* tick;                              // arrival counters of the cross-block K split (conv.hip: igemm_body XK), SCRATCH_TICKS words kept ZERO
                                               // between launches (the last block to arrive resets its tile's counter); nullptr when the registered
                                               // region was too small to carve them out
  float* xk_part;                              // the K split's partial tiles: SCRATCH_XK_BYTES of their OWN.  They are written and read with sc1
                                               // (write-through / L1-bypassing) accesses only, from every XCD; sharing addresses with the weight-gradient
                                               // slab -- plain stores and loads that leave dirty and clean lines in the per-XCD L2s -- gave wrong sums
                                               // under load (round 5: NaN gradients in 3 of 4 runs beside two other processes; never with either user alone)
};
constexpr size_t SCRATCH_TICK_BYTES = 65536;   // the tail of the registered region: 16384 counters
constexpr int SCRATCH_TICKS = (int)(SCRATCH_TICK_BYTES / 4);
constexpr size_t SCRATCH_XK_BYTES = 8u << 20;  // in front of the counters: 512 partial tiles of 64 x 64 fp32
extern Scratch g_scratch;

// ---- element types ---------------------------------------------------------------------------
struct bf16_t {
  unsigned short v;
};
struct f16_t {     // IEEE binary16 storage (the sliding-window inference configuration, src/api/infer.py:22-80 in fp16)
  unsigned short v;
};

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return __uint_as_float(((uint32_t)x.v) << 16); }
__device__ __forceinline__ float to_f32(f16_t x) { return (float)__builtin_bit_cast(_Float16, x.v); }
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t lo16) { return __uint_as_float(lo16 << 16); }
__device__ __forceinline__ float f16_bits_to_f32(uint32_t lo16) { return (float)__builtin_bit_cast(_Float16, (unsigned short)lo16); }

template <class T>
__device__ __forceinline__ T from_f32(float f);
template <>
__device__ __forceinline__ float from_f32<float>(float f) {
  return f;
}
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float f) {
  __bf16 b = (__bf16)f;  // hipcc: v_cvt_pk_bf16_f32, round-to-nearest-even, NaN stays NaN
  bf16_t r;
  r.v = __builtin_bit_cast(unsigned short, b);
  return r;
}
template <>
__device__ __forceinline__ f16_t from_f32<f16_t>(float f) {
  f16_t r;
  r.v = __builtin_bit_cast(unsigned short, (_Float16)f);   // v_cvt_f16_f32: round-to-nearest-even, overflow -> inf
  return r;
}
typedef __attribute__((ext_vector_type(2))) _Float16 emrt_half2_t;
typedef __attribute__((ext_vector_type(2))) float emrt_float2_t;
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
  emrt_half2_t h = __builtin_convertvector((emrt_float2_t){lo, hi}, emrt_half2_t);      // v_cvt_pk(rtz is NOT used: plain RNE converts)
  return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ void unpack_f16x2(uint32_t w, float& lo, float& hi) {
  const emrt_half2_t h = __builtin_bit_cast(emrt_half2_t, w);
  lo = (float)h.x;
  hi = (float)h.y;
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return (uint32_t)from_f32<bf16_t>(lo).v | ((uint32_t)from_f32<bf16_t>(hi).v << 16);
}

// 4-element vector access (the unit most memory-bound kernels work in): 16 B for f32, 8 B for bf16.
template <class T>
struct Vec4;
template <>
struct Vec4<float> {
  __device__ static __forceinline__ void load(const float* p, float (&o)[4]) {
    float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  __device__ static __forceinline__ void store(float* p, const float (&o)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  }
};
template <>
struct Vec4<bf16_t> {
  __device__ static __forceinline__ void load(const bf16_t* p, float (&o)[4]) {
    uint2 v = *reinterpret_cast<const uint2*>(p);
    o[0] = bf16_bits_to_f32(v.x & 0xffffu); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = bf16_bits_to_f32(v.y & 0xffffu); o[3] = __uint_as_float(v.y & 0xffff0000u);
  }
  __device__ static __forceinline__ void store(bf16_t* p, const float (&o)[4]) {
    uint2 v;
    v.x = pack_bf16x2(o[0], o[1]);
    v.y = pack_bf16x2(o[2], o[3]);
    *reinterpret_cast<uint2*>(p) = v;
  }
};

template <>
struct Vec4<f16_t> {
  __device__ static __forceinline__ void load(const f16_t* p, float (&o)[4]) {
    uint2 v = *reinterpret_cast<const uint2*>(p);
    unpack_f16x2(v.x, o[0], o[1]);
    unpack_f16x2(v.y, o[2], o[3]);
  }
  __device__ static __forceinline__ void store(f16_t* p, const float (&o)[4]) {
    uint2 v;
    v.x = pack_f16x2(o[0], o[1]);
    v.y = pack_f16x2(o[2], o[3]);
    *reinterpret_cast<uint2*>(p) = v;
  }
};

// 8-element vector (16 B of bf16, 32 B of f32)
template <class T>
struct Vec8;
template <>
struct Vec8<float> {
  __device__ static __forceinline__ void load(const float* p, float (&o)[8]) {
    float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
  }
  __device__ static __forceinline__ void store(float* p, const float (&o)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(o[4], o[5], o[6], o[7]);
  }
};
template <>
struct Vec8<bf16_t> {
  __device__ static __forceinline__ void unpack(const uint4& v, float (&o)[8]) {      // 16 raw bytes already in registers
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = bf16_bits_to_f32(w[i] & 0xffffu);
      o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  __device__ static __forceinline__ void load(const bf16_t* p, float (&o)[8]) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = bf16_bits_to_f32(w[i] & 0xffffu);
      o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  __device__ static __forceinline__ void store(bf16_t* p, const float (&o)[8]) {
    uint4 v;
    v.x = pack_bf16x2(o[0], o[1]); v.y = pack_bf16x2(o[2], o[3]);
    v.z = pack_bf16x2(o[4], o[5]); v.w = pack_bf16x2(o[6], o[7]);
    *reinterpret_cast<uint4*>(p) = v;
  }
};

template <>
struct Vec8<f16_t> {
  __device__ static __forceinline__ void unpack(const uint4& v, float (&o)[8]) {
    unpack_f16x2(v.x, o[0], o[1]); unpack_f16x2(v.y, o[2], o[3]);
    unpack_f16x2(v.z, o[4], o[5]); unpack_f16x2(v.w, o[6], o[7]);
  }
  __device__ static __forceinline__ void load(const f16_t* p, float (&o)[8]) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    unpack_f16x2(v.x, o[0], o[1]); unpack_f16x2(v.y, o[2], o[3]);
    unpack_f16x2(v.z, o[4], o[5]); unpack_f16x2(v.w, o[6], o[7]);
  }
  __device__ static __forceinline__ void store(f16_t* p, const float (&o)[8]) {
    uint4 v;
    v.x = pack_f16x2(o[0], o[1]); v.y = pack_f16x2(o[2], o[3]);
    v.z = pack_f16x2(o[4], o[5]); v.w = pack_f16x2(o[6], o[7]);
    *reinterpret_cast<uint4*>(p) = v;
  }
};

// ---- wave / block reductions (wave = 64 lanes on gfx950) ---------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// counter-based RNG for dropout: one 32-bit hash of (seed, salt, index).  keep iff u >= p.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float uniform01(uint64_t seed, uint32_t salt, uint64_t idx) {
  uint32_t a = mix32((uint32_t)idx ^ (uint32_t)seed);
  uint32_t b = mix32((uint32_t)(idx >> 32) ^ (uint32_t)(seed >> 32) ^ (salt * 0x9E3779B9U));
  uint32_t h = mix32(a ^ (b + 0x9E3779B9U + (a << 6) + (a >> 2)));
  return (float)(h >> 8) * (1.0f / 16777216.0f);
}

// Element-mode dropout (nn.Dropout on token tensors: the stand-alone kernels and the LayerNorm kernels that apply a branch's dropout themselves):
// the kernels move 4 consecutive elements per lane, so ONE 64-bit draw per aligned quad of the flat index gives their four 16-bit uniforms --
// 2 hash rounds per 4 elements where uniform01() needs 12 (the hash was most of the VALU work of the LayerNorm kernels: 14 + 14 launches a step).
// keep iff u16 >= p * 65536.  Every kernel that must agree on a mask calls these two functions with the same (seed, salt, quad index).
// The salt (one per dropout site) enters the FIRST round: every word of the draw -- hence every element of the quad -- differs between two sites
// that share the step's seed and the flat index (round 5 kept it out of h[0]: elements 0 and 1 of every quad had the same mask at every site).
__device__ __forceinline__ void drop_quad(uint64_t seed, uint32_t salt, uint64_t quad, uint32_t (&h)[2]) {
  h[0] = mix32(((uint32_t)quad ^ (uint32_t)seed) + salt * 0x9E3779B9U);
  h[1] = mix32(h[0] ^ (uint32_t)(quad >> 32) ^ (uint32_t)(seed >> 32) ^ (salt * 0x85EBCA6BU));
}
__device__ __forceinline__ bool drop_quad_keep(const uint32_t (&h)[2], int e, uint32_t thr) { return ((h[e >> 1] >> (16 * (e & 1))) & 0xffffu) >= thr; }
__device__ __forceinline__ uint32_t drop_thr16(float p) { return (uint32_t)(p * 65536.f); }

// Dropout inside a GEMM epilogue (conv.hip: linear1 -> ReLU -> Dropout of the FFN): the per-element hash above costs ~45 VALU instructions
// per element, which on a GEMM's critical path cost more than the separate dropout launch it replaced.  Here ONE group of
// 8 consecutive channels of a row draws four 32-bit words = eight 16-bit uniforms (~4 instructions per element); keep iff u16 >= p * 65536.
// The backward never re-derives this mask: the consumer's data gradient masks with (stored output > 0) (functional.conv2d: drop_rec).
__device__ __forceinline__ void drop_words8(uint64_t seed, uint32_t salt, uint32_t group, uint32_t (&h)[4]) {
  h[0] = mix32((group ^ (uint32_t)seed) + salt * 0x9E3779B9U);      // (the salt in the first round: see drop_quad)
  h[1] = mix32(h[0] ^ (uint32_t)(seed >> 32) ^ (salt * 0x85EBCA6BU));
  h[2] = mix32(h[1] + 0x9E3779B9U + (h[0] << 6));
  h[3] = mix32(h[2] ^ 0x85EBCA6BU ^ (h[1] >> 2));
}
__device__ __forceinline__ bool drop_keep8(const uint32_t (&h)[4], int e, uint32_t thr) { return ((h[e >> 1] >> (16 * (e & 1))) & 0xffffu) >= thr; }

// Flat index -> coordinates for the grid-stride elementwise kernels.  idx = ((i3 * D2 + i2) * D1 + i1) * D0 + i0.
// A 64-bit division is ~100 VALU instructions on gfx950 (no 64-bit divider, quarter-rate 32x32 multiplies): three of them per 16-byte vector
// made these "memory-bound" kernels VALU-bound.  Every EMRT tensor has fewer than 2^32 elements, so the 32-bit form runs (a 32-bit division is
// ~25 instructions); `small` is wave-uniform (idx < 2^32 for the whole grid), the 64-bit form stays for anything larger.
__device__ __forceinline__ void unravel2(long long idx, int D0, bool small, int& i0, long long& i1) {
  if (small) {
    const unsigned r = (unsigned)idx, q = r / (unsigned)D0;
    i0 = (int)(r - q * (unsigned)D0); i1 = (long long)q;
  } else {
    i1 = idx / D0; i0 = (int)(idx - i1 * D0);
  }
}
__device__ __forceinline__ void unravel3(long long idx, int D0, int D1, bool small, int& i0, int& i1, long long& i2) {
  if (small) {
    unsigned r = (unsigned)idx, q = r / (unsigned)D0;
    i0 = (int)(r - q * (unsigned)D0); r = q; q = r / (unsigned)D1;
    i1 = (int)(r - q * (unsigned)D1); i2 = (long long)q;
  } else {
    long long r = idx / D0; i0 = (int)(idx - r * D0);
    i2 = r / D1; i1 = (int)(r - i2 * D1);
  }
}
__device__ __forceinline__ void unravel4(long long idx, int D0, int D1, int D2, bool small, int& i0, int& i1, int& i2, int& i3) {
  if (small) {
    unsigned r = (unsigned)idx, q = r / (unsigned)D0;
    i0 = (int)(r - q * (unsigned)D0); r = q; q = r / (unsigned)D1;
    i1 = (int)(r - q * (unsigned)D1); r = q; q = r / (unsigned)D2;
    i2 = (int)(r - q * (unsigned)D2); i3 = (int)q;
  } else {
    long long r = idx / D0; i0 = (int)(idx - r * D0);
    long long q = r / D1; i1 = (int)(r - q * D1); r = q;
    q = r / D2; i2 = (int)(r - q * D2); i3 = (int)q;
  }
}

// entry points with a backward / training-only meaning accept f32 and bf16; fp16 (dtype 2) is inference-only
#define EMRT_REQUIRE_TRAIN_DTYPE(dtype) EMRT_REQUIRE((dtype) == EMRT_F32 || (dtype) == EMRT_BF16, "dtype must be 0 (f32) or 1 (bf16): fp16 (2) is inference-only")
#define EMRT_REQUIRE_FWD_DTYPE(dtype) EMRT_REQUIRE((dtype) == EMRT_F32 || (dtype) == EMRT_BF16 || (dtype) == EMRT_F16, "dtype must be 0 (f32), 1 (bf16) or 2 (f16)")

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace emrt
