// Two LDS-atomic access patterns for the MSDA value-gradient scatter (developer probe, GPU box).
//   A (current kernel): the two half-waves of a wave work on two different samples; each adds its 32 channels to the 4 corner rows of ITS pixel
//     (row pitch 33 ints): 4 wave instructions per 2 samples, each touching two unrelated 128-byte rows.
//   B (candidate): all 64 lanes work on ONE sample; lanes 0-31 add to corner x0, lanes 32-63 to corner x0 + 1 (row pitch 32 ints: adjacent pixels
//     sit in opposite halves of the 64 banks): 2 wave instructions per sample, never a bank conflict.
//   hipcc --offload-arch=gfx950 -O2 tools/exp/lds_scatter_patterns.hip -o tools/exp/lds_scatter_patterns.bin && tools/exp/lds_scatter_patterns.bin
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned* out, int iters, int W) {
  extern __shared__ int slab[];
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, h = lane >> 5, wave = tid >> 6;
  constexpr int PITCH = MODE == 0 ? 33 : 32;
  const int NPIX = 1000;
  for (int i = tid; i < (NPIX + 2 * W + 4) * PITCH; i += 1024) slab[i] = 0;
  __syncthreads();
  unsigned rng = (wave * 2 + (MODE == 0 ? h : 0)) * 2654435761u + 777u;      // A: one stream per half-wave; B: one per wave
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      rng = rng * 1664525u + 1013904223u;
      const int f = (int)((rng >> 8) % (unsigned)NPIX);
      int* cell = slab + f * PITCH + c;
      const int v = (int)(rng & 255u) - 128;
      atomicAdd(cell, v); atomicAdd(cell + PITCH, v + 1); atomicAdd(cell + W * PITCH, v + 2); atomicAdd(cell + (W + 1) * PITCH, v + 3);
    } else {
#pragma unroll
      for (int s = 0; s < 2; ++s) {      // two samples per iteration: the same work as one iteration of A
        rng = rng * 1664525u + 1013904223u;
        const int f = (int)((rng >> 8) % (unsigned)NPIX);
        int* cell = slab + (f + h) * PITCH + c;
        const int v = (int)(rng & 255u) - 128;
        atomicAdd(cell, v + h); atomicAdd(cell + W * PITCH, v + 2 + h);
      }
    }
  }
  __syncthreads();
  const long long t1 = clock64();
  if (tid == 0) { out[0] = (unsigned)(t1 - t0); out[1] = (unsigned)slab[37]; }
}

int main() {
  unsigned* d;
  hipMalloc(&d, 64);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
  const int iters = 4000;
  for (int W : {32, 33, 16, 8})
    for (int mode = 0; mode < 2; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 150000, 0, d, iters, W);
        else hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 150000, 0, d, iters, W);
        hipDeviceSynchronize();
      }
      unsigned h[2];
      hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
      printf("W %2d  pattern %s: %.2f clock64 ticks per 2 samples per wave (16 waves per CU; 4 wave instructions)\n", W, mode == 0 ? "A (half-waves, pitch 33)" : "B (x0 | x0+1, pitch 32) ",
             (double)h[0] / iters / 16.0);
    }
  return 0;
}
