// Experiment: is "conv epilogue -> grid barrier -> BatchNorm apply in the same kernel" cheaper than "conv epilogue | kernel boundary |
// bn_apply kernel"?  Models only the tail of the producer and the whole consumer: every block owns a 64x64 tile (bf16), adds its
// 64 column sums / sums of squares into one of 8 fp64 replicas with memory-side atomics (what igemm's epilogue does), then either
//   (a) ends, and a second kernel reads the statistics, re-reads the tile and writes the normalised tile, or
//   (b) arrives at a device-wide counter, spins until all blocks have arrived, reads the statistics with device-coherent loads and
//       writes the normalised tile from the registers it still holds.
// The spin is bounded (an error flag is set instead of hanging).  All launches replayed from one hipGraph of 200 layers.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/grid_barrier tools/exp/grid_barrier.hip
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdio.h>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int REPL = 8;

__device__ inline void tile_stats(const float (&v)[16], int col0, double* sums, int C, int blk) {
  // 256 threads: thread t owns 16 rows of column (t & 63), rows (t >> 6) * 16 ..; reduce over the 4 row groups through LDS
  __shared__ float red[2][4][64];
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { s += v[i]; q += v[i] * v[i]; }
  red[0][threadIdx.x >> 6][threadIdx.x & 63] = s;
  red[1][threadIdx.x >> 6][threadIdx.x & 63] = q;
  __syncthreads();
  if (threadIdx.x < 128) {
    const int w = threadIdx.x >> 6, c = threadIdx.x & 63;
    const float t = red[w][0][c] + red[w][1][c] + red[w][2][c] + red[w][3][c];
    atomicAdd(&sums[((blk & (REPL - 1)) * 2 + w) * C + col0 + c], (double)t);
  }
}

__device__ inline void load_tile(const __hip_bfloat16* x, int C, int row0, int col0, float (&v)[16]) {
  const int c = col0 + (threadIdx.x & 63), r0 = row0 + (threadIdx.x >> 6) * 16;
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = __bfloat162float(x[(size_t)(r0 + i) * C + c]);
}
__device__ inline void store_tile(__hip_bfloat16* y, int C, int row0, int col0, const float (&v)[16], float sc, float sh) {
  const int c = col0 + (threadIdx.x & 63), r0 = row0 + (threadIdx.x >> 6) * 16;
#pragma unroll
  for (int i = 0; i < 16; ++i) y[(size_t)(r0 + i) * C + c] = __float2bfloat16(fmaxf(v[i] * sc + sh, 0.f));
}

// producer tail: "accumulator" = the tile read from `src` (stands for the MFMA result), raw tile written, statistics added
__global__ __launch_bounds__(256) void k_producer(const __hip_bfloat16* src, __hip_bfloat16* x, double* sums, int C, int tiles_n) {
  const int row0 = (blockIdx.x / tiles_n) * 64, col0 = (blockIdx.x % tiles_n) * 64;
  float v[16];
  load_tile(src, C, row0, col0, v);
  const int c = col0 + (threadIdx.x & 63), r0 = row0 + (threadIdx.x >> 6) * 16;
#pragma unroll
  for (int i = 0; i < 16; ++i) x[(size_t)(r0 + i) * C + c] = __float2bfloat16(v[i]);
  tile_stats(v, col0, sums, C, blockIdx.x);
}
__device__ inline void channel_consts(const double* sums, int C, int c, double count, float& sc, float& sh, bool coherent) {
  double s = 0, q = 0;
#pragma unroll
  for (int r = 0; r < REPL; ++r) {
    if (coherent) {
      s += __hip_atomic_load(&sums[(r * 2 + 0) * C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      q += __hip_atomic_load(&sums[(r * 2 + 1) * C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else { s += sums[(r * 2 + 0) * C + c]; q += sums[(r * 2 + 1) * C + c]; }
  }
  const double mean = s / count, var = q / count - mean * mean;
  sc = (float)(1.0 / sqrt(var + 1e-5));
  sh = (float)(-mean) * sc;
}
__global__ __launch_bounds__(256) void k_apply(const __hip_bfloat16* x, __hip_bfloat16* y, const double* sums, int C, int tiles_n, double count) {
  const int row0 = (blockIdx.x / tiles_n) * 64, col0 = (blockIdx.x % tiles_n) * 64;
  float sc, sh;
  channel_consts(sums, C, col0 + (threadIdx.x & 63), count, sc, sh, false);
  float v[16];
  load_tile(x, C, row0, col0, v);
  store_tile(y, C, row0, col0, v, sc, sh);
}
__global__ __launch_bounds__(256) void k_fused(const __hip_bfloat16* src, __hip_bfloat16* x, __hip_bfloat16* y, double* sums, unsigned* counter,
                                               int* err, int C, int tiles_n, double count, int write_raw) {
  const int row0 = (blockIdx.x / tiles_n) * 64, col0 = (blockIdx.x % tiles_n) * 64;
  float v[16];
  load_tile(src, C, row0, col0, v);
  if (write_raw) {
    const int c = col0 + (threadIdx.x & 63), r0 = row0 + (threadIdx.x >> 6) * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) x[(size_t)(r0 + i) * C + c] = __float2bfloat16(v[i]);
  }
  tile_stats(v, col0, sums, C, blockIdx.x);
  __syncthreads();                                   // this block's atomics have been issued by threads 0..127
  if (threadIdx.x == 0) {
    __threadfence();
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1 << 22)) { *err = 1; break; }
    }
  }
  __syncthreads();
  float sc, sh;
  channel_consts(sums, C, col0 + (threadIdx.x & 63), count, sc, sh, true);
  store_tile(y, C, row0, col0, v, sc, sh);
}

template <class F> static double run(hipStream_t st, F&& enqueue, int reps) {
  enqueue(); hipStreamSynchronize(st);
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int r = 0; r < reps; ++r) enqueue();
  hipStreamSynchronize(st);
  return std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / reps;
}

int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const int LAYERS = 200;
  struct Case { int M, C; } cases[] = {{512, 2048}, {2048, 256}, {2048, 1024}, {8192, 128}, {8192, 512}, {32768, 64}, {32768, 128}};
  for (auto& cs : cases) {
    const int M = cs.M, C = cs.C, tiles_n = C / 64, blocks = (M / 64) * tiles_n;
    __hip_bfloat16 *src, *x, *y; double* sums; unsigned* counter; int* err;
    CK(hipMalloc(&src, (size_t)M * C * 2)); CK(hipMalloc(&x, (size_t)M * C * 2)); CK(hipMalloc(&y, (size_t)M * C * 2));
    const size_t stat_bytes = (size_t)LAYERS * REPL * 2 * C * 8;
    CK(hipMalloc(&sums, stat_bytes)); CK(hipMalloc(&counter, LAYERS * 4)); CK(hipMalloc(&err, 4));
    CK(hipMemset(src, 0x3c, (size_t)M * C * 2)); CK(hipMemset(err, 0, 4));
    double t[3];
    for (int mode = 0; mode < 3; ++mode) {
      auto enq = [&]() {
        hipMemsetAsync(sums, 0, stat_bytes, st); hipMemsetAsync(counter, 0, LAYERS * 4, st);
        for (int l = 0; l < LAYERS; ++l) {
          double* s = sums + (size_t)l * REPL * 2 * C;
          if (mode == 0) {
            hipLaunchKernelGGL(k_producer, dim3(blocks), dim3(256), 0, st, src, x, s, C, tiles_n);
            hipLaunchKernelGGL(k_apply, dim3(blocks), dim3(256), 0, st, x, y, s, C, tiles_n, (double)M);
          } else {
            hipLaunchKernelGGL(k_fused, dim3(blocks), dim3(256), 0, st, src, x, y, s, counter + l, err, C, tiles_n, (double)M, mode == 1 ? 1 : 0);
          }
        }
      };
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal)); enq(); CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      t[mode] = run(st, [&]() { hipGraphLaunch(ge, st); }, 5) / LAYERS;
      hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("M %6d C %5d blocks %5d: two kernels %.2f us   fused+barrier (raw tile also written) %.2f us   fused, no raw tile %.2f us%s\n", M, C, blocks,
           t[0], t[1], t[2], herr ? "   [SPIN LIMIT HIT]" : "");
    hipFree(src); hipFree(x); hipFree(y); hipFree(sums); hipFree(counter); hipFree(err);
  }
  return 0;
}
