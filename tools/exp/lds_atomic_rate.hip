// LDS atomic throughput on gfx950: ds_add_u32 against ds_add_u64 (no return), per wave instruction, for linear and for random (64-byte-row)
// addresses as the MSDA value-gradient scatter produces them.  Question: does packing two 32-bit fixed-point accumulators into one 64-bit add
// (exact with a sign-extended low half) halve the scatter's LDS-atomic time?        (developer probe, GPU box)
//   hipcc --offload-arch=gfx950 -O2 tools/exp/lds_atomic_rate.hip -o tools/exp/lds_atomic_rate.bin && tools/exp/lds_atomic_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>      // 0: u32 x8 per pixel visit, 1: u64 x4 per pixel visit (same bytes)
__global__ __launch_bounds__(1024) void k(unsigned* out, int iters, int random) {
  extern __shared__ unsigned char smem[];
  unsigned* s32 = (unsigned*)smem;
  unsigned long long* s64 = (unsigned long long*)smem;
  const int tid = threadIdx.x, lane = tid & 63, sub = lane & 3;
  for (int i = tid; i < 24576; i += 1024) s32[i] = 0;      // 96 KB slab
  __syncthreads();
  unsigned rng = tid * 2654435761u + 12345u;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    rng = rng * 1664525u + 1013904223u;
    // pixel row: 1536 rows of 64 bytes; a quad (4 lanes) covers one row, lane `sub` its 16 bytes
    const unsigned row = random ? ((rng >> 8) % 1536u) : (unsigned)(((tid >> 2) + it * 256) % 1536);
    if (MODE == 0) {
      unsigned* p = s32 + row * 16 + sub * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) __hip_atomic_fetch_add(p + e, rng + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      unsigned long long* p = s64 + row * 8 + sub * 2;
#pragma unroll
      for (int e = 0; e < 2; ++e) __hip_atomic_fetch_add(p + e, (unsigned long long)(rng + e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  const long long t1 = clock64();
  if (tid == 0) { out[blockIdx.x * 2] = (unsigned)(t1 - t0); out[blockIdx.x * 2 + 1] = s32[5]; }
}

int main() {
  unsigned* d;
  hipMalloc(&d, 4096);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  const int iters = 2000;
  for (int random = 0; random < 2; ++random)
    for (int mode = 0; mode < 2; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 98304, 0, d, iters, random);
        else hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 98304, 0, d, iters, random);
        hipDeviceSynchronize();
      }
      unsigned h[2];
      hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
      const double per_visit = (double)h[0] / iters;      // clock64 ticks (100 MHz?) per 16-byte-per-lane visit of 16 waves
      printf("%s addresses, %s: %u ticks for %d visits of 16 waves = %.3f ticks per visit (16 bytes per lane: %s)\n", random ? "random " : "linear ",
             mode == 0 ? "4 x ds_add_u32" : "2 x ds_add_u64", h[0], iters, per_visit, mode == 0 ? "64 wave instructions" : "32 wave instructions");
    }
  return 0;
}
