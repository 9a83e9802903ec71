// Probe (developer tool): what does `buffer_load_dwordx4 ... offen lds` (LDS-DMA) write for an out-of-range lane?
// The implicit-GEMM loaders zero-fill padding taps / ragged tails by giving such lanes a byte offset with bit 31 set
// (out of the descriptor's range).  For a register destination the hardware returns zeros; this checks the LDS destination.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void probe(const unsigned* g, unsigned* out, unsigned range) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* s32 = (unsigned*)smem;
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) s32[i] = 0xABABABABu;     // sentinel
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, (int)range, 0x00020000);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned off = (unsigned)(wave * 64 + lane) * 16u;
  if ((lane & 3) == 1) off |= 0x80000000u;                  // out of range: every 4th lane
  if ((lane & 3) == 2) off = range + 64u + lane * 16u;      // just past the end
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + wave * 1024), 16, off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) out[i] = s32[i];
}
int main() {
  const int n = 4096;
  unsigned *g, *o, h[4096], r[2048];
  for (int i = 0; i < n; ++i) h[i] = 0x1000000u + i;
  hipMalloc(&g, n * 4); hipMalloc(&o, 2048 * 4);
  hipMemcpy(g, h, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(128), 8192, 0, g, o, (unsigned)(n * 4));
  hipMemcpy(r, o, 2048 * 4, hipMemcpyDeviceToHost);
  int ok_in = 0, zero_oob = 0, sent_oob = 0, other = 0;
  for (int t = 0; t < 128; ++t) {
    const int lane = t & 63;
    for (int q = 0; q < 4; ++q) {
      const unsigned v = r[t * 4 + q];
      if ((lane & 3) == 1 || (lane & 3) == 2) { if (v == 0) ++zero_oob; else if (v == 0xABABABABu) ++sent_oob; else ++other; }
      else { if (v == 0x1000000u + t * 4 + q) ++ok_in; else ++other; }
    }
  }
  printf("LDS-DMA probe: in-range dwords correct %d/256, out-of-range dwords zero %d / untouched %d (of 256), other %d\n", ok_in, zero_oob, sent_oob, other);
  printf("lane 1 (bit31 OOB) dwords: %08x %08x %08x %08x; lane 2 (past end): %08x %08x %08x %08x\n", r[4], r[5], r[6], r[7], r[8], r[9], r[10], r[11]);
  return 0;
}
