#!/usr/bin/env python3
"""Kernel-variant probe for the batched path's bf16 tile kernel.

Extracts gemm_bf16w8_kernel (+ the shared epilogue) from csrc/kernels_tiles.h into a standalone program with
a tiny host harness (random bf16 shadow rows / queries, thresholds at +inf so the epilogue appends nothing),
applies named text substitutions to build VARIANTS of the kernel, and compiles each to tools/probes/bin/.
Run the binaries on the GPU box; each prints ms, TB/s of shadow bytes and the bf16 MFMA fraction.
A variant is an experiment, not product code: what wins is ported back into kernels_tiles.h by hand.

  python3 tools/probes/gemm_probe.py            # build all variants
  ONLY=base,no_pin python3 tools/probes/gemm_probe.py
"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
SRC = (ROOT / "wdbx-py_amd" / "csrc" / "kernels_tiles.h").read_text()
PRE = r'''#include <hip/hip_runtime.h>
#include <type_traits>
#include <cstdint>
#include <cmath>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef float f4 __attribute__((ext_vector_type(4)));
#define WDBX_METRIC_COSINE 0
#define WDBX_METRIC_L2 1
__device__ __forceinline__ uint32_t orderable(float f){uint32_t u=__float_as_uint(f);return (u&0x80000000u)?~u:(u|0x80000000u);}
__device__ __forceinline__ u64 make_key(float s,uint32_t row){return ((u64)orderable(s)<<32)|(uint32_t)~row;}
'''
a = SRC.index("typedef float f16v")
b = SRC.index("// CT = 32-query column tiles per wave")
c = SRC.index("typedef __bf16 bh8")
d = SRC.index("// queries [nv, pitch] fp32 -> bf16")
BASE = PRE + SRC[a:b] + SRC[c:d]
MAIN = r'''
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ inline float rnd(size_t i) { unsigned x = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 13); x ^= x << 13; x ^= x >> 17; x ^= x << 5; return ((int)(x & 0xFFFFFF) - 0x800000) * (1.0f / 0x800000); }
__global__ void fill_bf16(__bf16* p, size_t n, float sc) { for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) p[i] = (__bf16)(rnd(i + 77) * sc); }
int main(int argc, char** argv) {
  const uint32_t n_rows = argc > 1 ? atoi(argv[1]) : 6000000, dim = argc > 2 ? atoi(argv[2]) : 384;
  const uint32_t pitch16 = (dim + 127) / 128 * 128, tiles = (n_rows + 255) / 256;
  __bf16 *rows, *qb; float* tau; uint32_t* count; u64* cand;
  CK(hipMalloc(&rows, (size_t)n_rows * pitch16 * 2));
  hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, rows, (size_t)n_rows * pitch16, 0.05f);
  CK(hipMalloc(&qb, 256 * pitch16 * 2));
  hipLaunchKernelGGL(fill_bf16, dim3(256), dim3(256), 0, 0, qb, (size_t)256 * pitch16, 0.05f);
  CK(hipMalloc(&tau, 1024)); CK(hipMemsetD32((hipDeviceptr_t)tau, 0x7F800000, 256));
  CK(hipMalloc(&count, 1024)); CK(hipMemset(count, 0, 1024));
  CK(hipMalloc(&cand, 256 * 64 * 8));
  CK(hipDeviceSynchronize());
  GemmArgs g = {};
  g.rows = (const f4*)rows; g.n_rows = n_rows; g.pitch4 = pitch16 / 8; g.num_tiles = tiles; g.tile_stride = 1; g.tau = tau;
  g.cand = cand; g.count = count; g.cap = 64; g.qb16 = qb; g.qb_pitch16 = pitch16 / 8;
  auto fn = gemm_bf16w8_kernel<1, false, 4, 0, true>;
  const int lds = (2 * 256 + 2 * 256) * (64 * 2 + 16);
  CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(fn, dim3(256), dim3(512), lds, 0, g); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(fn, dim3(256), dim3(512), lds, 0, g);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
  {
    u64 h[4]; CK(hipMemcpy(h, cand, sizeof(h), hipMemcpyDeviceToHost));
    if (h[0] == 0x636c6b) printf("  shader clock: %.0f MHz over the kernel (workgroup 0: %llu cycles, %llu ticks of the 100 MHz wall clock)\n",
                                 (double)h[1] / (double)h[2] * 100.0, h[1], h[2]);
  }
  printf("%-22s rows=%u d=%u  %.3f ms  %.2f TB/s  bf16 MFMA %.1f%%\n", VARIANT, n_rows, dim, ms, (double)n_rows * pitch16 * 2 / ms / 1e9,
         2.0 * 256 * pitch16 * n_rows / (ms * 1e-3) / 2.5e15 * 100);
  return 0;
}
'''
PIN = "        __builtin_amdgcn_sched_barrier(0);\n"
VARIANTS = {
    "base": [],
    # let the compiler schedule the staging steps around the matrix ops
    "no_pin": [(PIN, "")],
}


def build(name, subs, outdir):
    src = BASE
    for old, new in subs:
        assert old in src, (name, old[:60])
        src = src.replace(old, new)
    src = '#define VARIANT "%s"\n' % name + src + MAIN
    tmp = Path("/tmp/gemm_probe_%s.hip" % name)
    tmp.write_text(src)
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Rpass-analysis=kernel-resource-usage",
                        "-o", str(outdir / ("w8_" + name)), str(tmp)], capture_output=True, text=True)
    lines = r.stderr.splitlines()
    regs = []
    for i, l in enumerate(lines):  # resource usage of the tile kernel only
        if "Function Name" in l and "gemm_bf16w8" in l:
            regs = [x.split("remark:")[1].strip() for x in lines[i + 1:i + 10] if "remark:" in x and any(t in x for t in ("VGPRs:", "AGPRs:", "Spill", "Occupancy"))]
    print(name, "rc", r.returncode, "|", "; ".join(regs) if r.returncode == 0 else r.stderr[-800:])


if __name__ == "__main__":
    out = ROOT / "tools" / "probes" / "bin"
    out.mkdir(parents=True, exist_ok=True)
    only = os.environ.get("ONLY")
    extra = Path(__file__).with_name("gemm_probe_variants.py")
    if extra.exists():
        exec(extra.read_text())  # may add to VARIANTS
    for name, subs in VARIANTS.items():
        if only and name not in only.split(","):
            continue
        build(name, subs, out)
