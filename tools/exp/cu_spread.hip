// Where do the workgroups of a SMALL grid land?  (developer probe, GPU box)  Each block records its XCC / SE / SH / CU id; the host counts how
// many distinct CUs a grid of G blocks occupies for a few block sizes and LDS requests.  Question behind it: the step's small-M GEMM launches
// have 64..128 blocks for 256 CUs; if the dispatcher stacks two of them on one CU while others idle, asking for > 80 KB of LDS would spread them.
//   hipcc --offload-arch=gfx950 -O2 tools/exp/cu_spread.hip -o tools/exp/cu_spread.bin && tools/exp/cu_spread.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>

__global__ void probe(unsigned* out, int spin) {
  extern __shared__ unsigned char smem[];
  unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);      // HW_REG_HW_ID
  unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);     // HW_REG_XCC_ID
  // keep the block resident for a while so that the whole grid is in flight at once
  unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) { __builtin_amdgcn_s_sleep(8); }
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
    smem[0] = 1;
  }
}

int main() {
  unsigned* d;
  hipMalloc(&d, 2 * 4096 * sizeof(unsigned));
  const int grids[] = {32, 64, 128, 256, 512};
  const int threads[] = {256, 1024};
  const int ldss[] = {0, 36864, 73728, 98304};
  for (int t : threads)
    for (int lds : ldss) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      for (int g : grids) {
        hipLaunchKernelGGL(probe, dim3(g), dim3(t), lds, 0, d, 200000);
        std::vector<unsigned> h(2 * g);
        hipMemcpy(h.data(), d, 2 * g * sizeof(unsigned), hipMemcpyDeviceToHost);
        std::set<unsigned> cus, xccs;
        int per_cu_max = 0;
        std::vector<int> cnt(1 << 16, 0);
        for (int b = 0; b < g; ++b) {
          const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 15u;
          const unsigned cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
          const unsigned key = (xcc << 8) | (se << 5) | (sh << 4) | cu;
          cus.insert(key);
          xccs.insert(xcc);
          if (++cnt[key] > per_cu_max) per_cu_max = cnt[key];
        }
        printf("threads %4d  lds %6d  grid %4d: %3zu distinct CUs on %zu XCCs, at most %d blocks on one CU\n", t, lds, g, cus.size(), xccs.size(), per_cu_max);
      }
    }
  return 0;
}
