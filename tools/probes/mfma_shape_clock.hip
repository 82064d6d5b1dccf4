// Probe: which i8 MFMA shape holds the higher clock under load?  Bare loops (operands in registers, random bytes, 2 waves
// per SIMD, every CU busy), the same integer ops per iteration: 16 x v_mfma_i32_32x32x32_i8 vs 32 x v_mfma_i32_16x16x64_i8.
// Prints wall ms per launch, TOPS and the in-kernel clock (s_memtime / s_memrealtime).   (MI355X_MICROARCH.md, DVFS item 7,
// reports 1.12-1.15x for the bf16 pair.)     build: hipcc --offload-arch=gfx950 -O3 mfma_shape_clock.hip -o bin/mfma_shape_clock
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(512) void loop_kernel(const i32x4* src, int* sink, unsigned long long* stamps, int iters) {
  const int lane = threadIdx.x & 63;
  i32x4 a[2], b[16];
  for (int i = 0; i < 2; ++i) a[i] = src[(i * 64 + lane) % 4096];
  for (int i = 0; i < 16; ++i) b[i] = src[((i + 2) * 64 + lane + threadIdx.x) % 4096];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  int out = 0;
  if constexpr (SHAPE == 32) {
    i32x16 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = i32x16{0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[h], b[i + 8 * h], acc[i], 0, 0, 0);
      asm volatile("" : "+v"(a[0]), "+v"(a[1]));
    }
    for (int i = 0; i < 8; ++i)
      for (int r = 0; r < 16; ++r) out ^= acc[i][r];
  } else {
    i32x4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = i32x4{0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i * 2 + h] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[h], b[i], acc[i * 2 + h], 0, 0, 0);
      asm volatile("" : "+v"(a[0]), "+v"(a[1]));
    }
    for (int i = 0; i < 32; ++i)
      for (int r = 0; r < 4; ++r) out ^= acc[i][r];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (out == 0x12345678) sink[0] = out;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t1 - t0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000, launches = argc > 2 ? atoi(argv[2]) : 40;
  const int zero = argc > 3 ? atoi(argv[3]) : 0;
  std::vector<int> h(4096 * 4);
  srand(7);
  for (auto& v : h) v = zero ? 0 : (int)((unsigned)rand() * 2654435761u ^ (unsigned)rand());
  i32x4* src;
  int* sink;
  unsigned long long* st;
  hipMalloc(&src, h.size() * 4);
  hipMalloc(&sink, 4);
  hipMalloc(&st, 256 * 16);
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int round = 0; round < 3; ++round)
    for (int shape : {32, 16}) {
      auto fn = shape == 32 ? loop_kernel<32> : loop_kernel<16>;
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(fn, dim3(256), dim3(512), 0, 0, src, sink, st, iters);
      hipEventRecord(e0);
      for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(fn, dim3(256), dim3(512), 0, 0, src, sink, st, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      ms /= launches;
      std::vector<unsigned long long> s(512);
      hipMemcpy(s.data(), st, 512 * 8, hipMemcpyDeviceToHost);
      double clk = 0;
      for (int i = 0; i < 256; ++i) clk += (double)s[2 * i] / (double)s[2 * i + 1] * 0.1;
      const double ops = 256.0 * 8 * iters * 16 * 65536.0;
      printf("shape %2d  ms %.3f  TOPS %.0f  in-kernel clock %.3f GHz  cycles/iter %.1f\n", shape, ms, ops / ms / 1e9, clk / 256,
             (double)s[0] / iters);
    }
  return 0;
}
