// Probe: how a blocking call learns that its (tiny) kernel has finished -- hipStreamSynchronize, an event of its own
// (hipEventRecord + hipEventSynchronize, what wdbx_index_search does since round 4), or the host polling a word the kernel's
// last store writes into mapped host memory -- or the host spinning on hipEventQuery / hipStreamQuery.  Prints the median wall time of launch + wait for each, over a kernel that
// does ~nothing and over one that spins for ~20 us.   build: hipcc --offload-arch=gfx950 -O3 completion_probe.hip -o bin/completion_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void work_kernel(volatile unsigned* flag, unsigned seq, unsigned* out, int spin) {
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while ((int)(__builtin_amdgcn_s_memrealtime() - t0) < spin) {}
  out[threadIdx.x] = seq + threadIdx.x;  // "results" (mapped host memory)
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();
    *flag = seq;
  }
}

static double med(std::vector<double>& v) {
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

int main() {
  unsigned *h, *d;
  hipHostMalloc((void**)&h, 4096, hipHostMallocMapped);
  hipHostGetDevicePointer((void**)&d, h, 0);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t ev;
  hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  volatile unsigned* flag = h;
  for (int spin : {0, 2000}) {  // s_memrealtime ticks at 100 MHz: 2000 = 20 us
    for (int mode = 0; mode < 5; ++mode) {
      std::vector<double> t;
      for (unsigned i = 1; i <= 2200; ++i) {
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(work_kernel, dim3(1), dim3(64), 0, s, d, i, d + 64, spin);
        if (mode == 0) hipStreamSynchronize(s);
        else if (mode == 1) { hipEventRecord(ev, s); hipEventSynchronize(ev); }
        else if (mode == 2) { while (*flag != i) {} }
        else if (mode == 3) { hipEventRecord(ev, s); while (hipEventQuery(ev) == hipErrorNotReady) {} }
        else { while (hipStreamQuery(s) == hipErrorNotReady) {} }
        auto t1 = std::chrono::steady_clock::now();
        if (i > 200) t.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
        if (mode == 2 && (i % 64) == 0) hipStreamSynchronize(s);  // (keep the queue short)
      }
      hipStreamSynchronize(s);
      printf("spin %5d ticks  %-34s median %.1f us\n", spin, mode == 0 ? "hipStreamSynchronize" : mode == 1 ? "hipEventRecord + hipEventSynchronize" : mode == 2 ? "host polls a mapped word" : mode == 3 ? "hipEventRecord + hipEventQuery loop" : "hipStreamQuery loop", med(t));
    }
  }
  return 0;
}
