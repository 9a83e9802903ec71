// Experiment: what does a dependent kernel boundary cost on this box -- eager vs hipGraph (stream capture), empty vs small kernels,
// kernarg size, 1 vs 256 blocks?  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/launch_floor tools/exp/launch_floor.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <chrono>
struct Big { char pad[1024]; };
__global__ void k_empty(float* p) { if (p && threadIdx.x == 9999) p[0] = 1.f; }
__global__ void k_big(Big b, float* p) { if (p && threadIdx.x == 9999) p[0] = b.pad[3]; }
__global__ void k_touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <class F> static double run(hipStream_t st, F&& enqueue, int reps) {
  enqueue(); hipStreamSynchronize(st);
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int r = 0; r < reps; ++r) enqueue();
  hipStreamSynchronize(st);
  return std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / reps;
}
int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  float* p; CK(hipMalloc(&p, 64 << 20)); CK(hipMemset(p, 0, 64 << 20));
  const int N = 600;
  Big b{};
  struct Case { const char* name; int kind; int blocks; int n; } cases[] = {
    {"empty 1 block", 0, 1, 0}, {"empty 256 blocks", 0, 256, 0}, {"empty 2048 blocks", 0, 2048, 0}, {"1KB kernarg 256 blocks", 1, 256, 0},
    {"touch 64 KB", 2, 64, 16384}, {"touch 4 MB", 2, 4096, 1 << 20}, {"touch 32 MB", 2, 32768, 8 << 20}};
  for (auto& c : cases) {
    auto enq = [&]() {
      for (int i = 0; i < N; ++i) {
        if (c.kind == 0) hipLaunchKernelGGL(k_empty, dim3(c.blocks), dim3(256), 0, st, p);
        else if (c.kind == 1) hipLaunchKernelGGL(k_big, dim3(c.blocks), dim3(256), 0, st, b, p);
        else hipLaunchKernelGGL(k_touch, dim3(c.blocks), dim3(256), 0, st, p, c.n);
      }
    };
    double eager = run(st, enq, 5) / N;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal)); enq(); CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    double graph = run(st, [&]() { hipGraphLaunch(ge, st); }, 5) / N;
    printf("%-26s eager %.2f us/kernel   hipGraph %.2f us/kernel\n", c.name, eager, graph);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  return 0;
}
