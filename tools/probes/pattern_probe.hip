// Read-pattern probe: how fast does HBM deliver a [rows x 1536 B] table when every workgroup walks its
// 128-row tile in column chunks of P bytes per row (the access order of an LDS-staged GEMM tile), as a
// function of P and of the bytes in flight?  No LDS, no MFMA: only the loads.
//   build: hipcc --offload-arch=gfx950 -O3 -o pattern_probe pattern_probe.hip ; run: ./pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int P, int NT>  // P = bytes per row per chunk
__global__ __launch_bounds__(256) void walk(const f4* rows, uint32_t n_tiles, uint32_t pitch4, float* sink) {
  constexpr int LPR = P / 16, RPP = 256 / LPR, NA = 128 / RPP;
  const uint32_t srow = threadIdx.x / LPR, sq = threadIdx.x % LPR;
  const uint32_t chunks = pitch4 / LPR;
  if (threadIdx.x >= RPP * LPR) return;  // P = 1536: 192 of the 256 threads
  f4 acc = {0, 0, 0, 0};
  for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const f4* base = rows + (size_t)(t * 128 + srow) * pitch4 + sq;
    for (uint32_t c = 0; c < chunks; ++c) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const f4* p = base + (size_t)(RPP * i) * pitch4 + c * LPR;
        f4 v = NT ? __builtin_nontemporal_load(p) : *p;
        acc += v;
      }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

template <int P, int NT>
static void run(const f4* d, uint32_t n_tiles, uint32_t pitch4, float* sink, int wg_per_cu) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int grid = 256 * wg_per_cu;
  hipLaunchKernelGGL((walk<P, NT>), dim3(grid), dim3(256), 0, 0, d, n_tiles, pitch4, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((walk<P, NT>), dim3(grid), dim3(256), 0, 0, d, n_tiles, pitch4, sink);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
  printf("P=%4d nt=%d wg/cu=%d  %.3f ms  %.2f TB/s\n", P, NT, wg_per_cu, ms, (double)n_tiles * 128 * pitch4 * 16 / ms / 1e9);
  fflush(stdout);
}

int main() {
  const uint32_t pitch4 = 96, n_tiles = 78125;  // 10M x 384 fp32
  const size_t bytes = (size_t)n_tiles * 128 * pitch4 * 16;
  f4* d; float* sink;
  CK(hipMalloc(&d, bytes)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(d, 0, bytes));
  for (int w : {1, 2, 4, 8}) {
    run<128, 1>(d, n_tiles, pitch4, sink, w);
    run<256, 1>(d, n_tiles, pitch4, sink, w);
    run<512, 1>(d, n_tiles, pitch4, sink, w);
    run<1536, 1>(d, n_tiles, pitch4, sink, w);
  }
  run<128, 0>(d, n_tiles, pitch4, sink, 8);
  run<1536, 0>(d, n_tiles, pitch4, sink, 8);
  return 0;
}
