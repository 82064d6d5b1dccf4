// One-kernel translation unit for iterating on an int8 tile kernel's code generation without the 4-minute device pass of the
// whole library:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iwdbx-py_amd/csrc -S --cuda-device-only \
//   -Rpass-analysis=kernel-resource-usage -DINST8='1, 8, 6, 384, 640' tools/probes/tile8_tu.hip -o /tmp/tile8.s
#include <hip/hip_runtime.h>
#include <climits>
#include <cmath>
#include <cstdint>
#include <type_traits>
typedef unsigned long long u64;
typedef float f4 __attribute__((ext_vector_type(4)));
#include "wdbx_hip.h"
#include "host_dispatch.h"
#include "kernels_common.h"
#include "kernels_scan.h"
#include "kernels_merge_select.h"
#include "kernels_tiles.h"
#include "kernels_scan8.h"
#include "kernels_tiles8.h"
#ifdef INST8
template __global__ void gemm_i8_kernel<INST8>(Gemm8Args);
#endif
#ifdef INSTM
template __global__ void merge_kernel<true>(MergeArgs);
#endif
#ifdef INSTS8
template __global__ void scan8_kernel<INSTS8>(Scan8Args);
#endif
