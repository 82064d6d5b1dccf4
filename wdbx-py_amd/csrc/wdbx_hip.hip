// wdbx_hip.hip -- MI355X (gfx950, CDNA4) implementation of the WDBX vector_search hot path.
//
// What the reference does on this path (paths under /root/reference):
//   FaissIndex.search   wdbx/core/indexing.py:983-1030  exact inner product of one unit query against
//                                                        every stored unit row, k best descending
//   FaissIndex.add      indexing.py:858-905, :921-968    rows normalised (:851-856) and appended
//   VectorStore.search  wdbx/core/vector_store.py:323-345 per-shard top-`limit`, concatenate, sort, cut
// Here: the corpus lives row-major fp32 in HBM; one streaming kernel computes every
// row's score and keeps a per-wavefront top-k; a small kernel merges the partial lists;
// across GPUs the per-shard lists are all-gathered with RCCL and merged again.
//
// Kernel inventory (DESIGN.md has the roofline arithmetic):
//   scan_kernel<L,QPL,METRIC,NT,MODE>  HBM-bound: reads N*pitch*4 bytes once; L lanes share a row,
//                                 16-byte non-temporal loads straight into VGPRs (no LDS round trip:
//                                 nothing is reused), query held in VGPRs, DPP tree for the L-lane sum;
//                                 MODE 1/0: per-wave sorted top-k list in registers (k <= 128) / LDS guarded
//                                 by a running threshold, 4 wave lists merged per workgroup;
//                                 MODE 2: one key per row to HBM for the radix select (k >= 200)
//   scan_kernel_generic<L,METRIC,MODE> any dimension (runtime loop, query staged in LDS)
//   merge_kernel<REG>             P sorted partial lists -> one sorted list (lane-per-list walk); also the
//                                 post-all-gather merge and the candidate selection of the batched path
//   radix_hist/pick/compact + sort_out   exact top-k of N keys, cost independent of k
//   gemm_topk_kernel<PHASE,KTAIL,CT,METRIC>   batched queries: exact fp32 MFMA tile + threshold filter
//   gemm_bf16w8_kernel<PHASE,KTAIL,CT,METRIC,SHADOW>   batched queries, default: bf16 MFMA tiles SELECT the
//                                 candidates (from the fp32 rows or their bf16 shadow copy), a rigorous error
//                                 margin keeps every true top-k row, rescore_kernel makes them exact fp32
//   row_sqnorm / tau_margin / rescore / rows_to_bf16 / queries_to_bf16   helpers of the batched path
//   fill_synthetic_kernel / normalize_rows_kernel / probe_read_kernel   ingest + measurement helpers
//
// Ordering everywhere is one total order on 64-bit keys:
//   key = (orderable(score) << 32) | ~row      (bigger key = better; 0 = empty slot)
// so "score descending, row ascending" is a single unsigned compare, ties are deterministic and a
// merged multi-shard result equals the single-shard result bit for bit.

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "wdbx_hip.h"

typedef unsigned long long u64;
typedef float f4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      int c_ = (e_ == hipErrorOutOfMemory) ? WDBX_E_NOMEM                                    \
               : (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? WDBX_E_NODEVICE   \
                                                                         : WDBX_E_HIP;       \
      return fail(c_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    }                                                                                        \
  } while (0)

#define NCCL_TRY(expr)                                                                        \
  do {                                                                                        \
    ncclResult_t r_ = (expr);                                                                 \
    if (r_ != ncclSuccess)                                                                    \
      return fail(WDBX_E_RCCL, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f2ord(float f) {
  uint32_t u = __float_as_uint(f);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
  uint32_t u = (o & 0x80000000u) ? (o ^ 0x80000000u) : ~o;
  return __uint_as_float(u);
}
__device__ __forceinline__ u64 make_key(float score, uint32_t row) {
  return ((u64)f2ord(score) << 32) | (u64)(~row);
}
__device__ __forceinline__ uint32_t key_row(u64 key) { return ~(uint32_t)(key & 0xFFFFFFFFull); }
__device__ __forceinline__ float key_score(u64 key) { return ord2f((uint32_t)(key >> 32)); }

__device__ __forceinline__ u64 readlane64(u64 v, int src) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
  return ((u64)hi << 32) | lo;
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// sum over aligned groups of L consecutive lanes; every lane of the group gets the sum
template <int L>
__device__ __forceinline__ float group_sum(float v) {
  if constexpr (L >= 2) v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  if constexpr (L >= 4) v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  if constexpr (L >= 8) v += dpp_mov<0x141>(v);  // row_half_mirror
  if constexpr (L >= 16) v += dpp_mov<0x140>(v); // row_mirror
  if constexpr (L >= 32) v += __shfl_xor(v, 16);
  if constexpr (L >= 64) v += __shfl_xor(v, 32);
  return v;
}

// Insert key c (c > current k-th) into the wave's sorted (descending) list of k keys in LDS.
// All 64 lanes cooperate; chunks are walked from the tail so a chunk only reads entries that are
// still original.  Returns the new k-th key (the wave's threshold).
__device__ __forceinline__ u64 list_insert(u64* list, int k, u64 c, int lane) {
  for (int base = ((k - 1) >> 6) << 6; base >= 0; base -= 64) {
    const int i = base + lane;
    u64 a = 0, ap = ~0ull;
    if (i < k) {
      a = list[i];
      if (i > 0) ap = list[i - 1];
    }
    const u64 b = (a > c) ? a : ((ap > c) ? c : ap);
    if (i < k) list[i] = b;
    // entries before this chunk are >= its first entry: if that one already beats c, nothing
    // further up moves
    const u64 first = readlane64(a, 0);
    if (first > c) break;
  }
  return list[k - 1];
}

// lane i <- lane i-1, lane 0 <- fill: v_mov_b32_dpp wave_shr:1 (lane 0 has no source and keeps `old`)
__device__ __forceinline__ u64 wave_shr1(u64 v, u64 fill) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)fill, (int)(uint32_t)v, 0x138, 0xF, 0xF, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(fill >> 32), (int)(uint32_t)(v >> 32), 0x138, 0xF, 0xF, false);
  return ((u64)hi << 32) | lo;
}

// A wave's sorted top-k list.  REG (k <= 128): entry i lives in lane i%64's register i/64 and an
// insert is a DPP shift + two compares per register, no LDS.  Otherwise the list lives in LDS
// (list_insert above).
template <bool REG>
struct TopList {
  u64* lds;
  u64 reg, reg1;  // entries 0..63 and 64..127
  int k;
  __device__ __forceinline__ void init(u64* p, int k_, int lane) {
    lds = p;
    k = k_;
    reg = 0;
    reg1 = 0;
    if constexpr (!REG)
      for (int i = lane; i < k; i += 64) lds[i] = 0;
  }
  __device__ __forceinline__ u64 insert(u64 c, int lane) {
    if constexpr (REG) {
      const u64 prev = wave_shr1(reg, ~0ull);
      if (k > 64) {
        const u64 carry = readlane64(reg, 63);  // the entry that may move from register 0 to register 1
        const u64 prev1 = wave_shr1(reg1, carry);
        reg1 = (reg1 > c) ? reg1 : ((prev1 > c) ? c : prev1);
      }
      reg = (reg > c) ? reg : ((prev > c) ? c : prev);
      return k > 64 ? readlane64(reg1, k - 65) : readlane64(reg, k - 1);
    } else {
      return list_insert(lds, k, c, lane);
    }
  }
  // offer every candidate lane's key; returns the new threshold (the k-th key)
  __device__ __forceinline__ u64 offer(u64 key, bool cand, u64 thr, int lane) {
    u64 m = __ballot(cand);
    while (m) {
      const int src = __builtin_ctzll(m);
      m &= m - 1;
      const u64 c = readlane64(key, src);
      if (c > thr) thr = insert(c, lane);
    }
    return thr;
  }
  // entry i (i = lane + 64*r), for i < k
  __device__ __forceinline__ u64 get(int i) const {
    if constexpr (REG)
      return i < 64 ? reg : reg1;
    else
      return lds[i];
  }
  // write entry i to dst[i * stride] for all i < k
  __device__ __forceinline__ void store(u64* dst, size_t stride, int lane) const {
    for (int i = lane; i < k; i += 64) dst[(size_t)i * stride] = get(i);
  }
};

// lane-per-list walk: lane owns list `p`, offers its current head while it beats the threshold
template <bool REG, typename Get>
__device__ __forceinline__ u64 walk_lists(Get get, bool owns, int len, TopList<REG>& top, u64 thr, int lane) {
  int ptr = 0;
  bool alive = owns;
  while (true) {
    const u64 key = (alive && ptr < len) ? get(ptr) : 0;
    const bool cand = key > thr;
    if (!__ballot(cand)) break;
    thr = top.offer(key, cand, thr, lane);
    alive = cand;  // lists are sorted: a head that lost cannot be followed by a winner
    ++ptr;
  }
  return thr;
}

template <bool NT>
__device__ __forceinline__ f4 ld16(const f4* p) {
  if constexpr (NT)
    return __builtin_nontemporal_load(p);
  else
    return *p;
}

template <int METRIC>
__device__ __forceinline__ f4 accum(f4 acc, f4 c, f4 q) {
  if constexpr (METRIC == WDBX_METRIC_COSINE) {
    acc.x = fmaf(c.x, q.x, acc.x);
    acc.y = fmaf(c.y, q.y, acc.y);
    acc.z = fmaf(c.z, q.z, acc.z);
    acc.w = fmaf(c.w, q.w, acc.w);
  } else {
    const float dx = c.x - q.x, dy = c.y - q.y, dz = c.z - q.z, dw = c.w - q.w;
    acc.x = fmaf(dx, dx, acc.x);
    acc.y = fmaf(dy, dy, acc.y);
    acc.z = fmaf(dz, dz, acc.z);
    acc.w = fmaf(dw, dw, acc.w);
  }
  return acc;
}

// "higher is better" ranking value from the accumulated lane-group sum
template <int METRIC>
__device__ __forceinline__ float rank_value(float s) {
  if constexpr (METRIC == WDBX_METRIC_L2) s = -s;
  return s + 0.0f;  // -0.0 -> +0.0 so equal scores have equal keys
}

struct ScanArgs {
  const f4* rows;       // [n_rows, pitch4] quads
  const f4* query;      // [pitch4]
  u64* partials;        // [k][P] sorted list per wave, transposed
  const uint32_t* mask; // optional row filter: bit r set = row r may be returned (metadata push-down)
  uint32_t n_rows;
  uint32_t pitch4;
  uint32_t groups;      // row groups in total
  uint32_t chunk;       // 0: waves interleave groups; else: groups per wave (contiguous)
  int k;
  int wg_merge;         // 1: one partial list per workgroup (4 wave lists merged here), 0: one per wave
  // repair launch of the shadow-selection path: do nothing unless *only_if_over > over_cap (the query's
  // candidate buffer overflowed, so its selection result is incomplete); null = always run
  const uint32_t* only_if_over;
  uint32_t over_cap;
};

// ------------------------------------------------------------------------------------------------
// scan kernel, specialised: L lanes per row, QPL quads (16 B) per lane per row, fully unrolled
// ------------------------------------------------------------------------------------------------
// MODE 0: per-wave top-k list in LDS, 1: in registers (k <= 128), 2: no list at all -- every row's key is
// written to a.partials[row] and the top-k is taken by the radix select below (large k)
// RAGGED: L*QPL > pitch4 -- the lane slots past the row end load the row's last quad again (always a
// valid address, the same cache line as a neighbour) and contribute zero, so ANY dimension up to
// 3072 floats runs on an unrolled instance (d = 100, 200, 300, 1000 ...) instead of the generic kernel
template <int L, int QPL, int METRIC, bool NT, int MODE, bool RAGGED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void scan_kernel(ScanArgs a) {
  if (a.only_if_over && *a.only_if_over <= a.over_cap) return;  // repair launch, nothing to repair (uniform)
  constexpr bool REG = MODE == 1;
  constexpr int R = 64 / L;                                              // rows per wave pass
  constexpr int U = (QPL >= 12) ? 1 : (QPL >= 6) ? 2 : (QPL >= 4) ? 3 : (QPL == 3) ? 4 : (QPL == 2) ? 6 : 8;  // passes in flight
  extern __shared__ u64 lds_lists[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane % L, g = lane / L;
  TopList<REG> top;
  if constexpr (MODE != 2) top.init(lds_lists + wave * a.k, a.k, lane);
  u64 thr = 0;

  // quad offsets of this lane inside a row; in a ragged instance the out-of-row slots are clamped to
  // the last quad and their query quad is zero (so they add exactly 0 for inner product; for L2 the
  // row value is zeroed too)
  f4 q[QPL];
  uint32_t qo[QPL];
#pragma unroll
  for (int i = 0; i < QPL; ++i) {
    const uint32_t o = j + i * L;
    qo[i] = RAGGED ? min(o, a.pitch4 - 1) : o;
    q[i] = a.query[qo[i]];
    if constexpr (RAGGED)
      if (o >= a.pitch4) q[i] = f4{0.f, 0.f, 0.f, 0.f};
  }

  const uint32_t W = gridDim.x * 4, wg = blockIdx.x * 4 + wave;
  uint32_t cur, end, stride;
  if (a.chunk) {
    cur = wg * a.chunk;
    end = min(cur + a.chunk, a.groups);
    stride = 1;
  } else {
    cur = wg;
    end = a.groups;
    stride = W;
  }
  const uint32_t last_row = a.n_rows - 1;

  for (; cur < end; cur += U * stride) {
    f4 v[U][QPL];
    uint32_t row[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t grp = cur + u * stride;
      row[u] = (grp < end) ? grp * R + g : 0xFFFFFFFFu;
      const uint32_t rc = min(row[u], last_row);  // clamp: tail lanes re-read the last row, masked below
      if constexpr (RAGGED) {
        const f4* p = a.rows + (size_t)rc * a.pitch4;
#pragma unroll
        for (int i = 0; i < QPL; ++i) {
          v[u][i] = ld16<NT>(p + qo[i]);
          if (j + i * L >= a.pitch4) v[u][i] = f4{0.f, 0.f, 0.f, 0.f};  // (also keeps Inf * 0 out of the sum)
        }
      } else {
        const f4* p = a.rows + (size_t)rc * a.pitch4 + j;
#pragma unroll
        for (int i = 0; i < QPL; ++i) v[u][i] = ld16<NT>(p + i * L);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < QPL; ++i) acc = accum<METRIC>(acc, v[u][i], q[i]);
      float s = (acc.x + acc.y) + (acc.z + acc.w);
      s = rank_value<METRIC>(group_sum<L>(s));
      const u64 key = make_key(s, row[u]);
      if constexpr (MODE == 2) {
        if (j == 0 && row[u] <= last_row) {
          bool ok = (s == s);
          if (a.mask && ok) ok = (a.mask[row[u] >> 5] >> (row[u] & 31)) & 1u;
          a.partials[row[u]] = ok ? key : 0ull;
        }
      } else {
        bool cand = (j == 0) && (row[u] <= last_row) && (s == s) && (key > thr);
        if (a.mask && cand) cand = (a.mask[row[u] >> 5] >> (row[u] & 31)) & 1u;  // only threshold-beaters look at the mask
        thr = top.offer(key, cand, thr, lane);
      }
    }
  }
  if constexpr (MODE != 2) {
    // the workgroup's 4 wave lists are merged here (wave 0 walks the other three), so the merge kernel
    // sees one list per workgroup instead of one per wave
    if (a.wg_merge) {
      if constexpr (REG) top.store(lds_lists + wave * a.k, 1, lane);
      __syncthreads();
      if (wave == 0) {
        const u64* other = lds_lists + (size_t)lane * a.k;
        walk_lists<REG>([&](int ptr) { return other[ptr]; }, lane >= 1 && lane < 4, a.k, top, thr, lane);
        top.store(a.partials + blockIdx.x, gridDim.x, lane);
      }
    } else {
      top.store(a.partials + wg, W, lane);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// scan kernel, generic: any pitch; L = min(8, pow2ceil(pitch4)) lanes per row chosen at launch,
// query staged in LDS, runtime loop with a predicated tail
// ------------------------------------------------------------------------------------------------
template <int L, int METRIC, int MODE>
__global__ __launch_bounds__(256) void scan_kernel_generic(ScanArgs a) {
  if (a.only_if_over && *a.only_if_over <= a.over_cap) return;  // repair launch, nothing to repair (uniform)
  constexpr bool REG = MODE == 1;
  constexpr int R = 64 / L;
  extern __shared__ u64 lds_lists[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane % L, g = lane / L;
  TopList<REG> top;
  const int klds = (MODE == 2) ? 0 : a.k;  // dump mode keeps no list
  if constexpr (MODE != 2) top.init(lds_lists + wave * a.k, a.k, lane);
  f4* qs = (f4*)(lds_lists + 4 * klds);  // 16-byte aligned: 4*k*8 is a multiple of 32
  for (uint32_t i = threadIdx.x; i < a.pitch4; i += 256) qs[i] = a.query[i];
  __syncthreads();
  u64 thr = 0;
  const uint32_t W = gridDim.x * 4, wg = blockIdx.x * 4 + wave;
  uint32_t cur, end, stride;
  if (a.chunk) {
    cur = wg * a.chunk;
    end = min(cur + a.chunk, a.groups);
    stride = 1;
  } else {
    cur = wg;
    end = a.groups;
    stride = W;
  }
  const uint32_t last_row = a.n_rows - 1;
  for (; cur < end; cur += stride) {
    const uint32_t row = cur * R + g;
    const uint32_t rc = min(row, last_row);
    const f4* p = a.rows + (size_t)rc * a.pitch4;
    f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    uint32_t i = j;
    if constexpr (L == 64) {
      // long rows (d > 3072): 8 non-temporal loads in flight per lane, as in the unrolled instances
      for (; i + 7 * L < a.pitch4; i += 8 * L) {
        f4 c[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) c[t] = __builtin_nontemporal_load(p + i + t * L);
#pragma unroll
        for (int t = 0; t < 8; t += 2) {
          acc0 = accum<METRIC>(acc0, c[t], qs[i + t * L]);
          acc1 = accum<METRIC>(acc1, c[t + 1], qs[i + (t + 1) * L]);
        }
      }
    }
    for (; i + L < a.pitch4; i += 2 * L) {
      const f4 c0 = p[i], c1 = p[i + L];
      acc0 = accum<METRIC>(acc0, c0, qs[i]);
      acc1 = accum<METRIC>(acc1, c1, qs[i + L]);
    }
    if (i < a.pitch4) acc0 = accum<METRIC>(acc0, p[i], qs[i]);
    float s = ((acc0.x + acc1.x) + (acc0.y + acc1.y)) + ((acc0.z + acc1.z) + (acc0.w + acc1.w));
    s = rank_value<METRIC>(group_sum<L>(s));
    const u64 key = make_key(s, row);
    if constexpr (MODE == 2) {
      if (j == 0 && row <= last_row) {
        bool ok = (s == s);
        if (a.mask && ok) ok = (a.mask[row >> 5] >> (row & 31)) & 1u;
        a.partials[row] = ok ? key : 0ull;
      }
    } else {
      bool cand = (j == 0) && (row <= last_row) && (s == s) && (key > thr);
      if (a.mask && cand) cand = (a.mask[row >> 5] >> (row & 31)) & 1u;
      thr = top.offer(key, cand, thr, lane);
    }
  }
  if constexpr (MODE != 2) {
    // the workgroup's 4 wave lists are merged here (wave 0 walks the other three), so the merge kernel
    // sees one list per workgroup instead of one per wave
    if (a.wg_merge) {
      if constexpr (REG) top.store(lds_lists + wave * a.k, 1, lane);
      __syncthreads();
      if (wave == 0) {
        const u64* other = lds_lists + (size_t)lane * a.k;
        walk_lists<REG>([&](int ptr) { return other[ptr]; }, lane >= 1 && lane < 4, a.k, top, thr, lane);
        top.store(a.partials + blockIdx.x, gridDim.x, lane);
      }
    } else {
      top.store(a.partials + wg, W, lane);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// merge kernel: one workgroup per query; P sorted lists of k keys -> one sorted list of k keys
// ------------------------------------------------------------------------------------------------
struct MergeArgs {
  const u64* in;          // entry i of list p of query q at in[q*q_stride + i*i_stride + p*p_stride]
  uint64_t q_stride, i_stride, p_stride;
  uint32_t P;
  const uint32_t* P_dev;  // optional per-query list count (clamped to P)
  int list_len;           // entries per input list (k for partial lists, 1 for unsorted candidates)
  int k;
  int metric;
  uint32_t row_base;      // added to rows when writing out_keys (local -> global rows)
  int64_t idx_base;       // added to rows when writing out_idx
  u64* out_keys;          // [nq, k] or null
  int64_t* out_idx;       // [nq, k] or null
  float* out_score;       // [nq, k] or null
  float* out_kth;         // [nq] ranking value of the k-th key, -inf when fewer than k keys; or null
  const uint32_t* only_if_over;  // [nq] or null: query q is merged only if only_if_over[q] > over_cap (see ScanArgs)
  uint32_t over_cap;
};

template <bool REG>
__global__ __launch_bounds__(1024) void merge_kernel(MergeArgs a) {
  if (a.only_if_over && a.only_if_over[blockIdx.x] <= a.over_cap) return;  // repair merge, nothing to repair
  extern __shared__ u64 lds_lists[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int k = a.k;
  TopList<REG> top;
  top.init(lds_lists + (size_t)wave * k, k, lane);
  const u64* in = a.in + (size_t)blockIdx.x * a.q_stride;
  const uint32_t P = a.P_dev ? min(a.P_dev[blockIdx.x], a.P) : a.P;
  u64 thr = 0;
  for (uint32_t p0 = wave * 64; p0 < P; p0 += nwaves * 64) {
    const uint32_t p = p0 + lane;
    const u64* mine = in + (size_t)p * a.p_stride;
    const uint64_t is = a.i_stride;
    thr = walk_lists<REG>([&](int ptr) { return mine[(size_t)ptr * is]; }, p < P, a.list_len, top, thr, lane);
  }
  if constexpr (REG) top.store(lds_lists + (size_t)wave * k, 1, lane);  // hand the register list over through LDS
  __syncthreads();
  if (wave == 0) {
    const u64* mine = lds_lists + (size_t)lane * k;
    TopList<REG> fin;
    fin.init(lds_lists + (size_t)nwaves * k, k, lane);
    const u64 kth = walk_lists<REG>([&](int ptr) { return mine[ptr]; }, lane < nwaves, k, fin, 0, lane);
    if (a.out_kth && lane == 0) a.out_kth[blockIdx.x] = kth ? key_score(kth) : -INFINITY;
    const size_t o = (size_t)blockIdx.x * k;
    for (int i = lane; i < k; i += 64) {
      const u64 key = fin.get(i);
      const uint32_t row = key_row(key);
      if (a.out_keys) a.out_keys[o + i] = key ? ((key & 0xFFFFFFFF00000000ull) | (u64)(~(row + a.row_base))) : 0;
      if (a.out_idx) a.out_idx[o + i] = key ? (int64_t)row + a.idx_base : -1;
      if (a.out_score) {
        float s = key_score(key);
        if (a.metric == WDBX_METRIC_L2) s = -s + 0.0f;
        a.out_score[o + i] = key ? s : 0.0f;
      }
    }
  }
}



// ------------------------------------------------------------------------------------------------
// large k: exact radix select over one key per row (the scan kernels' MODE 2 output).
//   8 passes of 8 bits, most significant first: histogram of the digit among keys that match the
//   prefix chosen so far -> pick the bucket holding the k-th largest -> narrow.  After the last pass the
//   prefix IS the k-th largest key (keys are unique); everything >= it is compacted and sorted.
//   Cost is independent of k (about 0.2 ms on 10 M rows) where the list kernels degrade (10 ms at k=1000).
// ------------------------------------------------------------------------------------------------
struct SelectState {
  u64 prefix;
  u64 mask;
  uint32_t need;
  uint32_t out_count;
  uint32_t hist[256];
};

__global__ void select_init_kernel(SelectState* st, uint32_t k) {
  if (threadIdx.x == 0) {
    st->prefix = 0;
    st->mask = 0;
    st->need = k;
    st->out_count = 0;
  }
  st->hist[threadIdx.x] = 0;
}

__global__ __launch_bounds__(256) void radix_hist_kernel(const u64* __restrict__ keys, u64 n, SelectState* st, int shift) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const u64 prefix = st->prefix, mask = st->mask;
  const int lane = threadIdx.x & 63;
  for (u64 i0 = (u64)blockIdx.x * 256; i0 < n; i0 += (u64)gridDim.x * 256) {
    const u64 i = i0 + threadIdx.x;
    const u64 key = (i < n) ? __builtin_nontemporal_load(keys + i) : 0ull;
    bool act = key != 0 && (key & mask) == prefix;
    const uint32_t digit = (uint32_t)(key >> shift) & 0xFFu;
    // wave-aggregated LDS atomics: scores cluster in a few buckets in the leading passes, where plain
    // per-lane atomics would serialise 64-deep on one address
    u64 todo = __ballot(act);
    while (todo) {
      const int src = __builtin_ctzll(todo);
      const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)digit, src);
      const u64 same = __ballot(act && digit == d0);
      if (lane == src) atomicAdd(&h[d0], (uint32_t)__builtin_popcountll(same));
      todo &= ~same;
    }
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&st->hist[threadIdx.x], h[threadIdx.x]);
}

__global__ void radix_pick_kernel(SelectState* st, int shift) {
  __shared__ uint32_t h[256];
  __shared__ uint32_t incl[256];  // incl[i] = sum of h[j], j >= i
  h[threadIdx.x] = st->hist[threadIdx.x];
  st->hist[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int i = 255; i >= 0; --i) {
      run += h[i];
      incl[i] = run;
    }
    uint32_t need = st->need;
    if (need > run) need = run;  // fewer valid keys than k: the smallest valid key becomes the cut
    int d = 0;
    if (need) {
      d = 255;
      while (d > 0 && incl[d] < need) --d;
      need -= incl[d] - h[d];  // keys in higher buckets are all taken
    }
    st->need = need;
    if (need) {
      st->prefix |= (u64)d << shift;
      st->mask |= 0xFFull << shift;
    } else {  // nothing to select (no valid key): make the cut unreachable
      st->prefix = ~0ull;
      st->mask = ~0ull;
    }
  }
}

__global__ __launch_bounds__(256) void radix_compact_kernel(const u64* __restrict__ keys, u64 n, SelectState* st, u64* out,
                                                            uint32_t k) {
  const u64 cut = st->prefix;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
    const u64 key = __builtin_nontemporal_load(keys + i);
    if (key != 0 && key >= cut) {
      const uint32_t pos = atomicAdd(&st->out_count, 1u);
      if (pos < k) out[pos] = key;
    }
  }
}

// one workgroup: bitonic sort (descending) of the <= k selected keys in LDS, then the usual outputs
__global__ __launch_bounds__(1024) void sort_out_kernel(const u64* sel, const SelectState* st, MergeArgs a, uint32_t npow2) {
  extern __shared__ u64 lds_lists[];
  const uint32_t have = min(st->out_count, (uint32_t)a.k);
  for (uint32_t i = threadIdx.x; i < npow2; i += blockDim.x) lds_lists[i] = (i < have) ? sel[i] : 0ull;
  __syncthreads();
  for (uint32_t size = 2; size <= npow2; size <<= 1)
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      for (uint32_t i = threadIdx.x; i < npow2 / 2; i += blockDim.x) {
        const uint32_t lo = (i / stride) * 2 * stride + (i % stride), hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const u64 x = lds_lists[lo], y = lds_lists[hi];
        if ((x < y) == desc) {
          lds_lists[lo] = y;
          lds_lists[hi] = x;
        }
      }
      __syncthreads();
    }
  for (uint32_t i = threadIdx.x; i < (uint32_t)a.k; i += blockDim.x) {
    const u64 key = lds_lists[i];
    const uint32_t row = key_row(key);
    if (a.out_keys) a.out_keys[i] = key ? ((key & 0xFFFFFFFF00000000ull) | (u64)(~(row + a.row_base))) : 0;
    if (a.out_idx) a.out_idx[i] = key ? (int64_t)row + a.idx_base : -1;
    if (a.out_score) {
      float s = key_score(key);
      if (a.metric == WDBX_METRIC_L2) s = -s + 0.0f;
      a.out_score[i] = key ? s : 0.0f;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// batched queries: scores[N, 256] = rows[N, d] . queries[256, d]^T on fp32 MFMA with a fused
// threshold filter (BASELINE config 4; extension, the reference is single-query: SURVEY F3).
//   workgroup tile 128 rows x 256 queries, K staged 32 floats at a time through LDS (double
//   buffered, rows padded to 36 floats: conflict-free ds_read_b128); 4 waves as 2 (rows) x 2
//   (queries), each 64 x 128 = 2 x 4 tiles of v_mfma_f32_32x32x2_f32 (exact fp32 fma chain).
//   PHASE 0 (sample tiles): per half-tile and query, the maximum score -> keys; the k-th largest of
//            them is a lower bound tau of the query's true k-th best score.
//   PHASE 1 (all tiles): every score >= tau is appended to the query's candidate buffer.
// The final top-k of the candidates is taken by merge_kernel (lists of length 1).
// ------------------------------------------------------------------------------------------------
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int GB_M = 128, GB_N = 256;  // GB_N: the largest query block (CT = 4)

struct GemmArgs {
  const f4* rows;
  const f4* queries;   // [64*CT, pitch4], rows beyond the valid queries are zero
  uint32_t n_rows, pitch4;
  uint32_t num_tiles;  // tiles this launch visits
  uint32_t tile_stride;
  u64* halfmax;        // PHASE 0: [256][2 * num_tiles]
  const float* tau;    // PHASE 1: [256]
  u64* cand;           // PHASE 1: [256][cap]
  uint32_t* count;     // PHASE 1: [256]
  uint32_t cap;
  const float* cn;     // L2 only: squared norm of every stored row
  const void* qb16;    // bf16 tile kernel: queries as bf16 [64*CT][qb_pitch16 * 8], zero padded
  uint32_t qb_pitch16; // its row pitch in 16-byte pieces (a whole number of 32-element chunks)
  uint32_t live;       // 0: every query of the block is live; else only queries < live (the rest neither
                       // report maxima nor append candidates: single-query passes use one column)
};

// Tile epilogue shared by the fp32 and bf16 tile kernels.  acc holds the wave's 64 rows x 32*CT queries in
// the 32x32 MFMA C layout: query = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// RW = 64-row wave groups per tile (tile rows = 64 * RW); PHASE 0 leaves one key per (query, tile, wave group).
template <int PHASE, int CT, int METRIC, int RW = 2>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, f16v (&acc)[2][CT], const float (&thr)[CT], uint32_t t,
                                              uint32_t trow0, int rh, int ch, int l31, int lh) {
  const uint32_t last_row = a.n_rows - 1;
  // epilogue: C layout of 32x32: query = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const uint32_t wrow0 = trow0 + rh * 64;
  const bool partial = trow0 + 64 * RW > a.n_rows;
  if constexpr (METRIC == WDBX_METRIC_L2) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t row = min(wrow0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, last_row);
        const float cn = a.cn[row];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct][r] = fmaf(2.0f, acc[rt][ct][r], -cn);
      }
  }
  if constexpr (PHASE == 0) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      float m = -INFINITY;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[rt][ct][r];
          if (partial) {
            const uint32_t row = wrow0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row >= a.n_rows) v = -INFINITY;
          }
          m = fmaxf(m, v);
        }
      m = fmaxf(m, __shfl_xor(m, 32));
      const uint32_t q = ch * (32 * CT) + ct * 32 + l31, ht = t * RW + rh;
      if (lh == 0 && (a.live == 0 || q < a.live)) {
        a.halfmax[(size_t)q * (RW * a.num_tiles) + ht] = (m == -INFINITY) ? 0ull : make_key(m + 0.0f, ht);
      }
    }
  } else {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const uint32_t q = ch * (32 * CT) + ct * 32 + l31;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        float m = acc[rt][ct][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[rt][ct][r]);
        if (m >= thr[ct]) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[rt][ct][r];
            const uint32_t row = wrow0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (v >= thr[ct] && row < a.n_rows) {
              const uint32_t pos = atomicAdd(&a.count[q], 1u);
              if (pos < a.cap) a.cand[(size_t)q * a.cap + pos] = make_key(v + 0.0f, row);
            }
          }
        }
      }
    }
  }
}

// CT = 32-query column tiles per wave: the workgroup covers GBN = 64*CT queries (256, 128 or 64), so a
// small batch does not pay for 256 columns (CT=1: about a quarter of the MFMA work of CT=4).
// (A BK=16 / two-workgroups-per-CU variant was measured slower, 16.1 vs 15.5 ms, and removed.)
// METRIC L2 ranks by  2 c.q - |c|^2  (= -|c-q|^2 + |q|^2, the query's own norm does not change the order);
// the candidates it selects are re-scored exactly by l2_rescore_kernel.
template <int PHASE, bool KTAIL, int CT, int METRIC>
__global__ __launch_bounds__(256) void gemm_topk_kernel(GemmArgs a) {
  constexpr int BK = 32;               // floats of K staged per chunk
  constexpr int GBN = 64 * CT;         // queries per workgroup tile
  constexpr int QPC = BK / 4;          // quads per row per chunk
  constexpr int LD = BK + 4;           // padded LDS row (floats): conflict-free ds_read_b128
  constexpr int S = BK / 8;            // MFMA sub-steps per chunk (8 k each)
  constexpr int RPP = 256 / QPC;       // rows staged per pass of the 256 threads
  constexpr int NA = GB_M / RPP, NB = GBN / RPP;
  extern __shared__ float lds_f[];
  float* As = lds_f;
  float* Bs = lds_f + 2 * GB_M * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rh = wave & 1, ch = wave >> 1, l31 = lane & 31, lh = lane >> 5;
  const uint32_t kchunks = (a.pitch4 + QPC - 1) / QPC;

  float thr[CT];
  if constexpr (PHASE == 1) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const uint32_t q = ch * (32 * CT) + ct * 32 + l31;
      thr[ct] = (a.live == 0 || q < a.live) ? a.tau[q] : INFINITY;
    }
  }
  // staging map: thread -> (tile row / query row = tid / QPC (+RPP per load), quad = tid % QPC)
  const uint32_t srow = tid / QPC, squad = tid % QPC;
  const uint32_t last_row = a.n_rows - 1;

  // The staging pipeline runs seamlessly ACROSS tiles: the loader has its own (tile, chunk) cursor one
  // step ahead of the compute cursor, so the first chunk of the next tile is already in LDS when a
  // tile's epilogue ends.
  f16v acc[2][CT];
  auto zero_acc = [&]() {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rt][ct][r] = 0.f;
  };
  zero_acc();
  f4 sa[NA], sb[NB];
  const f4* pa[NA];
  const f4* const pb = a.queries + (size_t)srow * a.pitch4;
  const size_t pb_step = (size_t)RPP * a.pitch4;
  uint32_t ld_tile = blockIdx.x, ld_kc = 0, kq = squad;  // loader cursor
  auto set_tile = [&](uint32_t tile) {
    const uint32_t r0 = tile * a.tile_stride * GB_M;
    // rows past the end are clamped to the last row (their scores are masked in the epilogue)
#pragma unroll
    for (int i = 0; i < NA; ++i) pa[i] = a.rows + (size_t)min(r0 + srow + RPP * i, last_row) * a.pitch4;
  };
  // K tail (pitch not a multiple of BK floats): quads past the row end re-read the row's last quad
  // (always inside the allocation) and are zeroed
  auto qoff = [&]() -> uint32_t { return KTAIL ? min(kq, a.pitch4 - 1) : kq; };
  auto gload_a = [&]() {
    const uint32_t o = qoff();
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      f4 v = __builtin_nontemporal_load(pa[i] + o);
      if constexpr (KTAIL)
        if (kq >= a.pitch4) v = f4{0.f, 0.f, 0.f, 0.f};
      sa[i] = v;
    }
  };
  auto gload_b = [&](int half) {
    const uint32_t o = qoff();
#pragma unroll
    for (int i = half * (NB / 2); i < (half + 1) * (NB / 2); ++i) {
      f4 v = pb[(size_t)i * pb_step + o];
      if constexpr (KTAIL)
        if (kq >= a.pitch4) v = f4{0.f, 0.f, 0.f, 0.f};
      sb[i] = v;
    }
  };
  auto gload_done = [&]() {  // advance the loader cursor
    kq += QPC;
    if (++ld_kc == kchunks) {
      ld_kc = 0;
      kq = squad;
      // past the last tile the loader simply re-reads it (valid memory, never consumed), which keeps the
      // main loop free of per-step branches
      if (ld_tile + gridDim.x < a.num_tiles) {
        ld_tile += gridDim.x;
        set_tile(ld_tile);
      }
    }
  };
  // single staging steps (compile-time index after unrolling): the main loop issues ONE of them in
  // the shadow of each MFMA pair, so their address arithmetic and issue never outlast a matrix op
  auto gload_one = [&](int j) {
    const uint32_t o = qoff();
    f4 v = (j < NA) ? __builtin_nontemporal_load(pa[j < NA ? j : 0] + o) : pb[(size_t)(j - NA) * pb_step + o];
    if constexpr (KTAIL)
      if (kq >= a.pitch4) v = f4{0.f, 0.f, 0.f, 0.f};
    if (j < NA)
      sa[j < NA ? j : 0] = v;
    else
      sb[j >= NA ? j - NA : 0] = v;
  };
  auto lstore_one = [&](int buf, int j) {
    if (j < NA)
      *(f4*)&As[(buf * GB_M + srow + RPP * j) * LD + squad * 4] = sa[j < NA ? j : 0];
    else
      *(f4*)&Bs[(buf * GBN + srow + RPP * (j - NA)) * LD + squad * 4] = sb[j >= NA ? j - NA : 0];
  };
  auto lstore_a = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NA; ++i) *(f4*)&As[(buf * GB_M + srow + RPP * i) * LD + squad * 4] = sa[i];
  };
  auto lstore_b = [&](int buf, int half) {
#pragma unroll
    for (int i = half * (NB / 2); i < (half + 1) * (NB / 2); ++i)
      *(f4*)&Bs[(buf * GBN + srow + RPP * i) * LD + squad * 4] = sb[i];
  };
  f4 af[2], bf[CT];
  auto frags = [&](int buf, int s) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
      af[rt] = *(const f4*)&As[(buf * GB_M + rh * 64 + rt * 32 + l31) * LD + (2 * s + lh) * 4];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
      bf[ct] = *(const f4*)&Bs[(buf * GBN + ch * (32 * CT) + ct * 32 + l31) * LD + (2 * s + lh) * 4];
  };
  auto mfma8 = [&](int e) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[rt][e], bf[ct][e], acc[rt][ct], 0, 0, 0);
  };

  if (blockIdx.x >= a.num_tiles) return;
  set_tile(ld_tile);
  gload_a();
  gload_b(0);
  gload_b(1);
  gload_done();
  lstore_a(0);
  lstore_b(0, 0);
  lstore_b(0, 1);
  __syncthreads();
  uint32_t it = 0;  // running chunk counter: LDS buffer = it & 1
  for (uint32_t t = blockIdx.x; t < a.num_tiles; t += gridDim.x) {
    const uint32_t trow0 = t * a.tile_stride * GB_M;
    for (uint32_t kc = 0; kc < kchunks; ++kc, ++it) {
      const int buf = it & 1;
      constexpr int MF = 8 * CT;                        // MFMAs per sub-step
      constexpr int NS = NA + NB;                       // staging steps per chunk
      constexpr int GAP = MF / NS > 0 ? MF / NS : 1;    // MFMAs between two staging steps
      // first sub-step: the next chunk's global loads, ONE per GAP MFMAs (pinned): a staging step and
      // its address arithmetic fit in the shadow of a 64-cycle matrix op, a clump of them does not
      // (hand-grouped clumps: 15.5 ms per 256-query batch, this: 14.85 ms)
      frags(buf, 0);
#pragma unroll
      for (int m = 0; m < MF; ++m) {
        const int e = m / (2 * CT), rt = (m / CT) & 1, ct = m % CT;
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[rt][e], bf[ct][e], acc[rt][ct], 0, 0, 0);
        if ((m + 1) % GAP == 0 && (m + 1) / GAP <= NS) {
          __builtin_amdgcn_sched_barrier(0);
          gload_one((m + 1) / GAP - 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      gload_done();
#pragma unroll
      for (int s = 1; s < S - 1; ++s) {
        frags(buf, s);
#pragma unroll
        for (int e = 0; e < 4; ++e) mfma8(e);
      }
      // last sub-step: the staged chunk's LDS stores, one per GAP MFMAs
      frags(buf, S - 1);
#pragma unroll
      for (int m = 0; m < MF; ++m) {
        const int e = m / (2 * CT), rt = (m / CT) & 1, ct = m % CT;
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[rt][e], bf[ct][e], acc[rt][ct], 0, 0, 0);
        if ((m + 1) % GAP == 0 && (m + 1) / GAP <= NS) {
          __builtin_amdgcn_sched_barrier(0);
          lstore_one(buf ^ 1, (m + 1) / GAP - 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();
    }

    gemm_epilogue<PHASE, CT, METRIC>(a, acc, thr, t, trow0, rh, ch, l31, lh);
    zero_acc();
  }
}


// ------------------------------------------------------------------------------------------------
// bf16 SELECTION tiles.  Same phases and epilogue as gemm_topk_kernel, but the products run on
// v_mfma_f32_32x32x16_bf16 (8x the fp32 MFMA rate), which turns a 256-query batch from a matrix-core-bound
// pass into a memory-bound one.  Their scores are approximations, used ONLY to select candidates: the
// threshold is lowered by a rigorous bound on the rounding error (tau_margin_kernel) so that no true
// top-k row can be filtered out, and every selected candidate is then re-scored in exact fp32
// (rescore_kernel).  The final ranking is therefore the exact fp32 ranking.
//   The query block is converted once per batch (queries_to_bf16_kernel).
// ------------------------------------------------------------------------------------------------
typedef __bf16 bh8 __attribute__((ext_vector_type(8)));
typedef __bf16 bh4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// 8-wave tile: 256 rows x 64*CT queries per workgroup of 512 threads, waves as 4 (row
// groups of 64) x 2 (query halves), one workgroup per CU = two waves per SIMD, so one wave's LDS and
// barrier waits are covered by the other's matrix ops.  Twice the rows per tile halves the query
// traffic from L2 per row byte and doubles the bytes a chunk keeps in flight.
//   SHADOW = false: rows are read as fp32 and rounded to bf16 on their way into LDS (32-element chunks);
//   SHADOW = true:  rows are read from the bf16 shadow copy of the corpus (64-element chunks, no
//                   conversion, half the HBM bytes; the shadow is zero padded to whole chunk pairs).
//   Register ring: two chunks of row loads, one of query loads (queries are L2 hits and are issued
//   first, so in-order completion never holds them behind younger row loads).  The chunk loop is
//   unrolled twice (ring slots and LDS buffers static); rows are padded to an even number of chunks.
// ------------------------------------------------------------------------------------------------
constexpr int GW_M = 256;  // rows per 8-wave tile

template <int PHASE, bool KTAIL, int CT, int METRIC, bool SHADOW>
__global__ __launch_bounds__(512) void gemm_bf16w8_kernel(GemmArgs a) {
  constexpr int BK = SHADOW ? 64 : 32;    // elements per chunk
  constexpr int GBN = 64 * CT;            // queries per workgroup tile
  constexpr int LDB = BK * 2 + 16;        // bytes per LDS row: BK bf16 + 16 bytes of padding (conflict-free ds_read_b128)
  constexpr int QPR = 8;                  // 16-byte pieces per row per chunk in GLOBAL memory (fp32: 32 el, bf16: 64 el)
  constexpr int ARP = 512 / QPR;          // rows staged per pass of the 512 threads
  constexpr int NA = GW_M / ARP;          // row loads per thread per chunk (4)
  constexpr int PPR = BK / 8;             // 16-byte bf16 pieces per query per chunk
  constexpr int BRP = 512 / PPR;          // queries staged per pass
  constexpr int NB = GBN / BRP;           // query loads per thread per chunk
  constexpr int STEPS = BK / 16;          // 16-deep MFMA steps per chunk
  static_assert(NB >= 1, "the 8-wave tile needs at least 128 queries with fp32 rows");
  extern __shared__ float lds_f[];
  char* const As = (char*)lds_f;
  char* const Bs = As + 2 * GW_M * LDB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rh = wave & 3, ch = wave >> 2, l31 = lane & 31, lh = lane >> 5;
  // chunks per row, rounded up to a pair (the surplus chunk is zeros on both sides)
  const uint32_t kchunks = ((a.pitch4 + QPR - 1) / QPR + 1) / 2 * 2;

  float thr[CT];
  if constexpr (PHASE == 1) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const uint32_t q = ch * (32 * CT) + ct * 32 + l31;
      thr[ct] = (a.live == 0 || q < a.live) ? a.tau[q] : INFINITY;
    }
  }
  const uint32_t srow = tid / QPR, squad = tid % QPR;   // A staging: tile row (+ARP per load), 16-byte piece of the chunk
  const uint32_t brow = tid / PPR, bpiece = tid % PPR;  // B staging: query (+BRP per load), 16-byte piece of the chunk
  const uint32_t last_row = a.n_rows - 1;

  f16v acc[2][CT];
  auto zero_acc = [&]() {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rt][ct][r] = 0.f;
  };
  zero_acc();

  f4 ra[2][NA];
  f4 rb[NB];
  const f4* pa[NA];
  const f4* const pb = (const f4*)a.qb16 + (size_t)brow * a.qb_pitch16 + bpiece;
  const size_t pb_step = (size_t)BRP * a.qb_pitch16;
  uint32_t ld_tile = blockIdx.x, ld_kc = 0;  // row loader cursor (tile, chunk)
  uint32_t lb_kc = 0;                         // query loader cursor (chunk; the queries are the same for every tile)
  auto set_tile = [&](uint32_t tile) {
    const uint32_t r0 = tile * a.tile_stride * GW_M;
    // rows past the end are clamped to the last row (their scores are masked in the epilogue)
#pragma unroll
    for (int i = 0; i < NA; ++i) pa[i] = a.rows + (size_t)min(r0 + srow + ARP * i, last_row) * a.pitch4;
  };
  auto gload_a = [&](int slot, int j) {
    const uint32_t kq = ld_kc * QPR + squad;
    // K tail (fp32 rows only): pieces past the row end re-read the row's last piece and are zeroed
    f4 v = __builtin_nontemporal_load(pa[j] + (KTAIL ? min(kq, a.pitch4 - 1) : kq));
    if constexpr (KTAIL)
      if (kq >= a.pitch4) v = f4{0.f, 0.f, 0.f, 0.f};
    ra[slot][j] = v;
  };
  auto gload_b = [&](int j) { rb[j] = pb[(size_t)j * pb_step + lb_kc * PPR]; };
  auto gload_a_done = [&]() {  // advance the row cursor
    if (++ld_kc == kchunks) {
      ld_kc = 0;
      // past the last tile the loader simply re-reads it (valid memory, never consumed)
      if (ld_tile + gridDim.x < a.num_tiles) {
        ld_tile += gridDim.x;
        set_tile(ld_tile);
      }
    }
  };
  auto gload_b_done = [&]() {
    if (++lb_kc == kchunks) lb_kc = 0;
  };
  auto lstore_a = [&](int buf, int slot, int j) {
    const f4 v = ra[slot][j];
    if constexpr (SHADOW) {
      *(f4*)(As + (buf * GW_M + srow + ARP * j) * LDB + squad * 16) = v;
    } else {
      bh4 h;
      h[0] = (__bf16)v.x;
      h[1] = (__bf16)v.y;
      h[2] = (__bf16)v.z;
      h[3] = (__bf16)v.w;
      *(bh4*)(As + (buf * GW_M + srow + ARP * j) * LDB + squad * 8) = h;
    }
  };
  auto lstore_b = [&](int buf, int j) { *(f4*)(Bs + (buf * GBN + brow + BRP * j) * LDB + bpiece * 16) = rb[j]; };
  bh8 af[2], bf[CT];
  auto frags = [&](int buf, int s) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) af[rt] = *(const bh8*)(As + (buf * GW_M + rh * 64 + rt * 32 + l31) * LDB + s * 32 + lh * 16);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
      bf[ct] = *(const bh8*)(Bs + (buf * GBN + ch * (32 * CT) + ct * 32 + l31) * LDB + s * 32 + lh * 16);
  };

  if (blockIdx.x >= a.num_tiles) return;
  set_tile(ld_tile);
  constexpr int NS = NA + NB;  // staging steps per chunk
  // prologue: chunk 0 into LDS buffer 0; query chunk 1 and row chunks 1, 2 in flight
#pragma unroll
  for (int j = 0; j < NB; ++j) gload_b(j);
  gload_b_done();
#pragma unroll
  for (int j = 0; j < NA; ++j) gload_a(0, j);
  gload_a_done();
#pragma unroll
  for (int j = 0; j < NA; ++j) lstore_a(0, 0, j);
#pragma unroll
  for (int j = 0; j < NB; ++j) lstore_b(0, j);
#pragma unroll
  for (int j = 0; j < NB; ++j) gload_b(j);
  gload_b_done();
#pragma unroll
  for (int c = 1; c <= 2; ++c) {
#pragma unroll
    for (int j = 0; j < NA; ++j) gload_a(c % 2, j);
    gload_a_done();
  }
  __syncthreads();

  constexpr int MF = 2 * CT;             // MFMAs per 16-deep step
  constexpr int MH = MF * STEPS / 2;     // MFMAs per half chunk
  // one chunk: the first half of its MFMAs shadows the LDS stores of the next chunk, the second half the
  // global loads into the registers just freed, one staging step at a time between matrix ops
  auto stage = [&](int buf, int slot, int j, bool store) {
    if (store) {
      if (j < NA) lstore_a(buf, slot, j < NA ? j : 0);
      else lstore_b(buf, j >= NA ? j - NA : 0);
    } else {
      if (j < NB) gload_b(j < NB ? j : 0);
      else gload_a(slot, j >= NB ? j - NB : 0);
    }
  };
  auto body = [&](auto S) {
    constexpr int s = decltype(S)::value;
    constexpr int slot = (s + 1) % 2;
    const int buf = s & 1;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      int done = 0;
#pragma unroll
      for (int mm = 0; mm < MH; ++mm) {
        const int st = half * (STEPS / 2) + mm / MF, m = mm % MF;
        if (m == 0) frags(buf, st);
        const int rt = m / CT, ct = m % CT;
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rt], bf[ct], acc[rt][ct], 0, 0, 0);
        const int upto = ((mm + 1) * NS) / MH;  // NS staging steps spread over MH matrix ops
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = done; j < upto; ++j) stage(buf ^ 1, slot, j, half == 0);
        __builtin_amdgcn_sched_barrier(0);
        done = upto;
      }
    }
    gload_b_done();
    gload_a_done();
    __syncthreads();
  };
  for (uint32_t t = blockIdx.x; t < a.num_tiles; t += gridDim.x) {
    for (uint32_t kc = 0; kc < kchunks; kc += 2) {
      body(std::integral_constant<int, 0>{});
      body(std::integral_constant<int, 1>{});
    }
    gemm_epilogue<PHASE, CT, METRIC, 4>(a, acc, thr, t, t * a.tile_stride * GW_M, rh, ch, l31, lh);
    zero_acc();
  }
}

// rows [r0, n) fp32 -> the bf16 shadow copy (round to nearest even), zero padded to its own pitch
__global__ __launch_bounds__(256) void rows_to_bf16_kernel(const float* rows, u64 r0, u64 n, uint32_t pitch, __bf16* out,
                                                           uint32_t pitch16) {
  const u64 total = (n - r0) * (pitch16 / 8);  // 8-element pieces
  for (u64 e = (u64)blockIdx.x * 256 + threadIdx.x; e < total; e += (u64)gridDim.x * 256) {
    const u64 r = r0 + e / (pitch16 / 8);
    const uint32_t c = (uint32_t)(e % (pitch16 / 8)) * 8;
    bh8 h;
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = (__bf16)((c + i < pitch) ? rows[r * pitch + c + i] : 0.0f);
    *(bh8*)(out + r * pitch16 + c) = h;
  }
}

// ------------------------------------------------------------------------------------------------
// int8 SELECTION scan for single queries.  The bytes a query has to read are the bound, so the rows are
// kept a third time as unsigned bytes u = round(c / s) + 128 with a per-row scale s = max|c| / 127
// (rows_to_u8_kernel; pitch rounded up to 128 bytes so every row starts a cache line), a quarter of the
// fp32 bytes.  scan8_kernel streams them like scan_kernel streams floats (L lanes per row, 16-byte
// non-temporal loads straight into VGPRs, query held in fp32 registers, DPP tree for the L-lane sum) and
// forms  w = s * (sum u_i q_i - 128 sum q_i)  ~  c.q  with the query in full fp32, so the ONLY error is
// the rows' quantisation:  |w - c.q| <= m = 0.51 s |q|_1  (0.5 s per element, the 0.01 covers the fp32
// roundings of the scale, of the quotient and of this kernel's own summation: gamma * 255 < 0.007).
//   PHASE 0 (sampled 64-row groups): per group the maximum of the LOWER bounds w - m; the k-th largest of
//            them, tau, is a lower bound of the query's true k-th best score (k distinct rows reach it).
//            One launch serves all queries of a round (grid.y): they sample the same rows, which then come
//            from L2 / Infinity Cache instead of HBM.
//   PHASE 1 (all rows): every row whose UPPER bound w + m reaches tau is appended to the candidate buffer.
// No true top-k row can be missed; rescore_kernel then computes the candidates' exact fp32 scores from the
// fp32 rows and merge_kernel ranks those.  L2 selects by 2 w - |c|^2 (cached fp32 norms; bound
// 2 m + 3e-5 |c|^2).  Rows with a non-finite element carry a NaN scale: never sampled, always candidates.
// ------------------------------------------------------------------------------------------------
typedef uint32_t u4v __attribute__((ext_vector_type(4)));

struct Scan8Args {
  const u4v* rows8;   // [n_rows][pieces] 16-byte pieces of u8
  const float* scale;   // [n_rows]
  const float* cn;      // L2: squared fp32 norm per row
  const f4* query;      // fp32 [pieces * 4] quads (zero padded by the caller's buffer pitch or by clamping)
  const float* qinfo;   // [0] |q|_1, [1] sum q
  const uint32_t* mask; // optional row filter (bit r set = row r may be returned), as in ScanArgs
  uint32_t n_rows, pieces, qquads;  // qquads: quads the query buffer really holds
  u64* halfmax;         // PHASE 0: one key per sampled 64-row group
  uint32_t num_tiles, tile_stride;  // PHASE 0: tiles of 256 rows = 4 groups, every tile_stride-th tile
  const float* tau;     // PHASE 1
  u64* cand;
  uint32_t* count;
  uint32_t cap;
};

__device__ __forceinline__ float u8_dot16(u4v v, const f4 (&q)[4], float acc) {
  float a0 = acc, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // (uint -> float of one byte: v_cvt_f32_ubyte0..3)
    const uint32_t w = v[i];
    a0 = fmaf((float)(w & 0xFFu), q[i].x, a0);
    a1 = fmaf((float)((w >> 8) & 0xFFu), q[i].y, a1);
    a2 = fmaf((float)((w >> 16) & 0xFFu), q[i].z, a2);
    a3 = fmaf((float)(w >> 24), q[i].w, a3);
  }
  return (a0 + a1) + (a2 + a3);
}

template <int L, int QPL, int METRIC, int PHASE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void scan8_kernel(Scan8Args a) {
  constexpr int R = 64 / L;  // rows per wave pass
  constexpr int U = (QPL >= 6) ? 2 : (QPL >= 4) ? 3 : (QPL == 3) ? 4 : (QPL == 2) ? 6 : 8;  // passes in flight
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane % L, g = lane / L;
  if constexpr (PHASE == 0) {  // one launch samples for every query of a round: blockIdx.y = query
    a.query += (size_t)blockIdx.y * a.qquads;
    a.qinfo += 2 * blockIdx.y;
    a.halfmax += (size_t)blockIdx.y * a.num_tiles * 4;
  }
  // this lane's share of the query: pieces j, j+L, ... = 16 floats each (quads past the buffer are zero)
  f4 q[QPL][4];
#pragma unroll
  for (int i = 0; i < QPL; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const uint32_t quad = (uint32_t)(j + i * L) * 4 + t;
      q[i][t] = quad < a.qquads ? a.query[quad] : f4{0.f, 0.f, 0.f, 0.f};
    }
  const float q1 = a.qinfo[0], qsum128 = 128.0f * a.qinfo[1];
  const uint32_t last_row = a.n_rows - 1;
  // w and the bound m of one row from the lane-group sum (scale and norm were loaded with the row)
  auto finish = [&](float s, float sc, float cn, float& m) -> float {
    float w = sc * (s - qsum128);
    m = 0.51f * sc * q1;
    if constexpr (METRIC == WDBX_METRIC_L2) {
      w = fmaf(2.0f, w, -cn);
      m = fmaf(2.0f, m, 3e-5f * cn);
    }
    return w;
  };

  if constexpr (PHASE == 0) {
    const uint32_t ngroups = a.num_tiles * 4;  // sampled 64-row groups, one per wave at a time
    for (uint32_t grp = blockIdx.x * 4 + wave; grp < ngroups; grp += gridDim.x * 4) {
      const uint32_t row0 = (grp >> 2) * a.tile_stride * 256 + (grp & 3) * 64;
      float best = -INFINITY;
#pragma unroll 1
      for (int p0 = 0; p0 < 64 / R; p0 += U) {
        u4v v[U][QPL];
        uint32_t row[U];
        float sc[U], cn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          row[u] = (p0 + u < 64 / R) ? row0 + (p0 + u) * R + g : 0xFFFFFFFFu;
          const uint32_t rc = min(row[u], last_row);
          const u4v* p = a.rows8 + (size_t)rc * a.pieces + j;
#pragma unroll
          for (int i = 0; i < QPL; ++i) v[u][i] = p[i * L];  // default cache policy: the round's other queries re-read these rows
          sc[u] = a.scale[rc];
          cn[u] = METRIC == WDBX_METRIC_L2 ? a.cn[rc] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          float s = 0.f;
#pragma unroll
          for (int i = 0; i < QPL; ++i) s = u8_dot16(v[u][i], q[i], s);
          s = group_sum<L>(s);
          if (row[u] <= last_row) {
            float m;
            const float w = finish(s, sc[u], cn[u], m);
            const float lo = w - m;
            // (NaN scale: not sampled; masked-out rows cannot vouch for the threshold either)
            if (lo == lo && (!a.mask || ((a.mask[row[u] >> 5] >> (row[u] & 31)) & 1u))) best = fmaxf(best, lo);
          }
        }
      }
      for (int o = 32; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o));
      if (lane == 0) a.halfmax[grp] = (best == -INFINITY) ? 0ull : make_key(best + 0.0f, grp);
    }
  } else {
    const float thr = a.tau[0];
    const uint32_t groups = (a.n_rows + R - 1) / R;
    const uint32_t W = gridDim.x * 4;
    for (uint32_t cur = blockIdx.x * 4 + wave; cur < groups; cur += U * W) {
      u4v v[U][QPL];
      uint32_t row[U];
      float sc[U], cn[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t grp = cur + u * W;
        row[u] = (grp < groups) ? grp * R + g : 0xFFFFFFFFu;
        const uint32_t rc = min(row[u], last_row);
        const u4v* p = a.rows8 + (size_t)rc * a.pieces + j;
#pragma unroll
        for (int i = 0; i < QPL; ++i) v[u][i] = __builtin_nontemporal_load(p + i * L);
        sc[u] = a.scale[rc];
        cn[u] = METRIC == WDBX_METRIC_L2 ? a.cn[rc] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < QPL; ++i) s = u8_dot16(v[u][i], q[i], s);
        s = group_sum<L>(s);
        if (j == 0 && row[u] <= last_row) {
          float m;
          const float w = finish(s, sc[u], cn[u], m);
          // !(w + m < thr): also true for a NaN bound, so rows with non-finite elements always go to the exact pass
          // only rows that clear the threshold look at their mask bit
          if (!(w + m < thr) && (!a.mask || ((a.mask[row[u] >> 5] >> (row[u] & 31)) & 1u))) {
            const uint32_t pos = atomicAdd(a.count, 1u);
            if (pos < a.cap) a.cand[pos] = make_key((w == w) ? w + 0.0f : INFINITY, row[u]);
          }
        }
      }
    }
  }
}

// rows [r0, n) fp32 -> u8 shadow + per-row scale, one wave per row
__global__ __launch_bounds__(256) void rows_to_u8_kernel(const float* rows, u64 r0, u64 n, uint32_t dim, uint32_t pitch,
                                                         uint8_t* out, uint32_t pitch8, float* scale) {
  const int lane = threadIdx.x & 63;
  const u64 wave = (u64)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (u64)gridDim.x * 4;
  for (u64 r = r0 + wave; r < n; r += nw) {
    const float* p = rows + r * pitch;
    float mx = 0.f;
    bool finite = true;
    for (uint32_t c = lane; c < dim; c += 64) {
      const float v = p[c];
      finite = finite && (fabsf(v) <= 3.4028235e38f);
      mx = fmaxf(mx, fabsf(v));
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    finite = __all(finite);
    const float sc = finite ? mx / 127.0f : NAN, inv = (finite && mx > 0.f) ? 127.0f / mx : 0.f;
    for (uint32_t c = lane; c < pitch8; c += 64) {
      float x = (c < dim && finite) ? rintf(p[c] * inv) : 0.f;
      x = fminf(fmaxf(x, -127.f), 127.f);
      out[r * pitch8 + c] = (uint8_t)((int)x + 128);
    }
    if (lane == 0) scale[r] = sc;
  }
}

// per query: |q|_1 and sum q (one wave per query)
__global__ void query_info_kernel(const float* queries, uint32_t pitch, int nv, float* qinfo) {
  const int qi = blockIdx.x, lane = threadIdx.x;
  if (qi >= nv) return;
  const float* p = queries + (size_t)qi * pitch;
  float s1 = 0.f, s = 0.f;
  for (uint32_t c = lane; c < pitch; c += 64) {
    s1 += fabsf(p[c]);
    s += p[c];
  }
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o);
    s += __shfl_xor(s, o);
  }
  if (lane == 0) {
    // |q|_1 is used as an upper bound: round it up past its own summation error
    qinfo[2 * qi] = s1 * (1.0f + 1e-5f);
    qinfo[2 * qi + 1] = s;
  }
}

// queries [nv, pitch] fp32 -> bf16 blocks of [gbn, kpad] (round to nearest even), zero padded in both
// directions; block b holds queries b*live ... b*live + live - 1 in its first rows (live = gbn: one block of
// up to gbn queries; live = 1: one query per block, for single-query passes)
__global__ __launch_bounds__(256) void queries_to_bf16_kernel(const float* q, uint32_t pitch, uint32_t nv, __bf16* out,
                                                              uint32_t kpad, uint32_t gbn, uint32_t live, uint32_t blocks) {
  const uint32_t total = blocks * gbn * kpad;
  for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const uint32_t row = e / kpad, c = e - row * kpad, blk = row / gbn, r = row - blk * gbn;
    const uint32_t src = blk * live + r;
    out[e] = (__bf16)((r < live && src < nv && c < pitch) ? q[(size_t)src * pitch + c] : 0.0f);
  }
}

// ------------------------------------------------------------------------------------------------
// L2 on the batched path: row norms, threshold margin, exact re-scoring of the selected candidates
// ------------------------------------------------------------------------------------------------
// cn[r] = sum c^2 (one wave per row) and the running maximum of cn (float bits of non-negative values
// order like unsigned integers)
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float* rows, u64 r0, u64 n, uint32_t pitch, float* cn,
                                                         uint32_t* cn_max_bits) {
  const int lane = threadIdx.x & 63;
  const u64 wave = (u64)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (u64)gridDim.x * 4;
  const uint32_t pitch4 = pitch / 4;
  uint32_t wmax = 0;  // this wave's running maximum: ONE atomic per wave at the end, not one per row
  for (u64 r = r0 + wave; r < n; r += nw) {
    const f4* p = (const f4*)(rows + r * pitch);
    float s = 0.f;
    for (uint32_t c = lane; c < pitch4; c += 64) {
      const f4 v = __builtin_nontemporal_load(p + c);
      s = fmaf(v.x, v.x, s);
      s = fmaf(v.y, v.y, s);
      s = fmaf(v.z, v.z, s);
      s = fmaf(v.w, v.w, s);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
      cn[r] = s;
      if (s == s) wmax = max(wmax, __float_as_uint(s));
    }
  }
  if (lane == 0 && wmax) atomicMax(cn_max_bits, wmax);
}

// tau[q] -= margin(q), a rigorous bound on the rounding error of the SELECTION scores, so that no true
// top-k row can fall below the threshold.  With u = 2^-24, gamma = n u / (1 - n u) for fp32 chains of n terms:
//   fp32 tiles, L2:      v = 2 c.q - |c|^2,  |v_fp32 - v| <= gamma (2 |c||q| + |c|^2)
//   bf16 tiles:          c and q are rounded to bf16 (relative error <= 2^-8 each), their products are exact
//                        in fp32, so |dot_bf16 - c.q| <= (2^-7 + 2^-16 + gamma) |c||q|  (Cauchy-Schwarz);
//                        cosine: v = dot;  L2: v = 2 dot - |c|^2 with the fp32 bound on the second term.
// Both the threshold (a maximum of such values) and every candidate carry that error, hence 2x.
// (fp32 tiles with the cosine metric need no margin: selection and final scores are the same numbers.)
__global__ void tau_margin_kernel(float* tau, const float* queries, uint32_t pitch, int nv, const uint32_t* cn_max_bits,
                                  int metric, int bf16) {
  const int q = blockIdx.x, lane = threadIdx.x;  // one wave per query
  if (q >= nv) return;
  const float* p = queries + (size_t)q * pitch;
  float s = 0.f;
  for (uint32_t c = lane; c < pitch; c += 64) s = fmaf(p[c], p[c], s);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) {
    const float cmax = __uint_as_float(*cn_max_bits);
    const float nu = (float)(pitch + 18) * 5.9604645e-08f;
    const float gamma = 1.02f * nu / (1.0f - nu);
    const float eps_dot = gamma + (bf16 ? 1.05f * 0.0078125f : 0.0f);
    const float cq = sqrtf(cmax * s) * 1.0001f;
    float margin = metric == WDBX_METRIC_L2 ? 2.0f * (2.0f * eps_dot * cq + gamma * cmax) : 2.0f * eps_dot * cq;
    margin *= 1.01f;
    if (!(margin == margin)) margin = INFINITY;  // NaN query: select everything, the exact pass decides
    if (tau[q] > -INFINITY) tau[q] -= margin;
  }
}

// every kept candidate of every query is re-scored exactly in fp32, one wave per candidate: cosine by the
// inner product, L2 by the direct form sum (c - q)^2 (no cancellation); its key becomes (score, row)
template <int METRIC>
__global__ __launch_bounds__(256) void rescore_kernel(const f4* rows, uint32_t pitch4, const f4* queries, u64* cand,
                                                      const uint32_t* count, uint32_t cap) {
  const int lane = threadIdx.x & 63;
  const uint32_t q = blockIdx.y;
  const uint32_t have = min(count[q], cap);
  const f4* qp = queries + (size_t)q * pitch4;
  for (uint32_t j = blockIdx.x * 4 + (threadIdx.x >> 6); j < have; j += gridDim.x * 4) {
    u64* slot = cand + (size_t)q * cap + j;
    const uint32_t row = key_row(*slot);
    const f4* cp = rows + (size_t)row * pitch4;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (uint32_t i = lane; i < pitch4; i += 64) acc = accum<METRIC>(acc, cp[i], qp[i]);
    float s = (acc.x + acc.y) + (acc.z + acc.w);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (METRIC == WDBX_METRIC_L2) s = -s;
    if (lane == 0) *slot = (s == s) ? make_key(s + 0.0f, row) : 0ull;
  }
}

// ------------------------------------------------------------------------------------------------
// ingest helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 splitmix64(u64 x) {
  u64 z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void fill_synthetic_kernel(float* dst, u64 n, uint32_t dim, uint32_t pitch,
                                                             u64 seed, u64 counter_row0) {
  const u64 total = n * pitch;
  for (u64 e = (u64)blockIdx.x * 256 + threadIdx.x; e < total; e += (u64)gridDim.x * 256) {
    const u64 r = e / pitch;
    const uint32_t c = (uint32_t)(e - r * pitch);
    float val = 0.f;
    if (c < dim) {
      const u64 h = splitmix64(seed ^ ((counter_row0 + r) * dim + c));
      val = (float)((int)(h >> 40) - (1 << 23)) * 1.1920928955078125e-07f;  // 2^-23, exact
    }
    dst[e] = val;
  }
}

// measurement aid: stream the stored rows with the scan kernel's load shape (16 B per lane,
// grid-stride) and nothing else -- the read ceiling the scan kernel is compared with
template <bool NT>
__global__ __launch_bounds__(256) void probe_read_kernel(const f4* p, u64 n_quads, float* sink) {
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  const u64 stride = (u64)gridDim.x * 256 * 8;
  for (u64 i = (u64)blockIdx.x * 256 * 8 + threadIdx.x; i < n_quads; i += stride) {
    f4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const u64 e = i + (u64)u * 256;
      v[u] = (e < n_quads) ? ld16<NT>(p + e) : acc;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc.x += v[u].x;
      acc.y += v[u].y;
      acc.z += v[u].z;
      acc.w += v[u].w;
    }
  }
  const float s = (acc.x + acc.y) + (acc.z + acc.w);
  if (s == 1.2345e38f) sink[0] = s;  // keeps the loads alive, practically never true
}

// one wave per row: x / sqrt(sum x^2) when the norm is > 0 (indexing.py:851-856)
__global__ __launch_bounds__(256) void normalize_rows_kernel(float* rows, u64 n, uint32_t pitch) {
  const int lane = threadIdx.x & 63;
  const u64 wave = (u64)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (u64)gridDim.x * 4;
  for (u64 r = wave; r < n; r += nw) {
    float* p = rows + r * pitch;
    float s = 0.f;
    for (uint32_t c = lane; c < pitch; c += 64) s = fmaf(p[c], p[c], s);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float nrm = sqrtf(s);
    if (nrm > 0.f)
      for (uint32_t c = lane; c < pitch; c += 64) p[c] = p[c] / nrm;
  }
}

// ------------------------------------------------------------------------------------------------
// host side: the handle
// ------------------------------------------------------------------------------------------------
struct EventPool {
  std::vector<hipEvent_t> ev;  // pairs
  size_t used = 0;
};

struct wdbx_index {
  int device = 0, dim = 0, pitch = 0, metric = 0;
  int cu_count = 256;
  uint64_t n = 0, cap = 0;
  float* d_rows = nullptr;
  hipStream_t stream = nullptr;
  std::mutex mu;
  // scratch (grown on demand, reused)
  u64* d_partials = nullptr;
  size_t partials_bytes = 0;
  u64* d_local_keys = nullptr;
  size_t local_keys_bytes = 0;
  u64* d_gathered = nullptr;
  size_t gathered_bytes = 0;
  float* d_q = nullptr;
  size_t q_bytes = 0;
  int64_t* d_oidx = nullptr;
  float* d_oscore = nullptr;
  size_t out_elems = 0;
  // communicator
  ncclComm_t comm = nullptr;
  int nranks = 1, rank = 0;
  uint64_t row_base = 0;
  // pinned, device-mapped staging for small blocking searches: the kernels read the query from and
  // write the result to host memory directly (no memcpy calls on the latency path)
  char* h_stage = nullptr;
  char* h_stage_dev = nullptr;
  u64* d_dump = nullptr;  // one key per row (large-k select)
  size_t dump_bytes = 0;
  u64* d_sel = nullptr;
  size_t sel_bytes = 0;
  SelectState* d_state = nullptr;
  size_t state_bytes = 0;
  uint32_t* d_mask = nullptr;
  size_t mask_bytes = 0;
  const uint32_t* active_mask = nullptr;  // set only for the duration of a masked search (under the mutex)
  // batched (GEMM) path scratch
  float* d_qblock = nullptr;
  size_t qblock_bytes = 0;
  u64* d_halfmax = nullptr;
  size_t halfmax_bytes = 0;
  float* d_tau = nullptr;
  size_t tau_bytes = 0;
  u64* d_cand = nullptr;
  size_t cand_bytes = 0;
  uint32_t* d_count = nullptr;
  size_t count_bytes = 0;
  uint32_t last_batch_nq = 0, last_batch_cap = 0;
  void* d_qb16 = nullptr;  // bf16 tiles: the query block as bf16
  size_t qb16_bytes = 0;
  float* d_cn = nullptr;  // L2 / bf16 batched path: squared row norms for rows [0, cn_rows), and their maximum
  size_t cn_bytes = 0;
  uint64_t cn_rows = 0;
  void* d_rows16 = nullptr;  // bf16 shadow copy of rows [0, shadow_rows), row pitch pitch16 elements (zero padded)
  size_t rows16_bytes = 0;
  uint64_t shadow_rows = 0;
  uint32_t pitch16 = 0;
  int last_gemm_mode = 0;  // tile kernel family the last batch ran on (GEMM_FP32 / GEMM_BF16 / GEMM_BF16_SHADOW)
  uint8_t* d_rows8 = nullptr;  // u8 shadow copy of rows [0, shadow8_rows) for the single-query selection scan, pitch8 bytes
  float* d_scale8 = nullptr;   // its per-row scales
  float* d_qinfo = nullptr;    // per query of a round: |q|_1, sum q
  size_t rows8_bytes = 0, scale8_bytes = 0, qinfo_bytes = 0;
  uint64_t shadow8_rows = 0;
  uint32_t pitch8 = 0;
  int last_single_path = 0;    // 0 fp32 scan, 1 bf16 tiles, 2 u8 scan (what the last single-query search ran on)
  uint32_t* d_cnmax = nullptr;
  size_t cnmax_bytes = 0;
  // profiling
  bool profile = false;
  EventPool scan_ev, merge_ev, gemm_ev, sample_ev;
  // options
  int64_t opt_lanes = 0, opt_blocks = 0, opt_nt = 1, opt_blocked = 0, opt_batch = 32, opt_generic = 0;
  int64_t opt_scan8_wgs = 2, opt_scan_shadow = 2, opt_gemm_bf16 = 2, opt_gemm_l2 = 1, opt_force_ragged = 0, opt_gemm_ct = 0, opt_wg_merge = 1, opt_zero_copy = 1, opt_lds_lists = 0, opt_select_min_k = 200, opt_gemm_min_nq = 4, opt_gemm_min_rows = 65536, opt_gemm_sample_div = 0;
};

struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) (void)hipSetDevice(dev);
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

static int grow(void** p, size_t* have, size_t need) {
  if (need <= *have) return WDBX_OK;
  if (*p) HIP_TRY(hipFree(*p));
  *p = nullptr;
  *have = 0;
  HIP_TRY(hipMalloc(p, need));
  *have = need;
  return WDBX_OK;
}

// ---- scan dispatch ------------------------------------------------------------------------------
typedef void (*scan_fn)(ScanArgs);
struct ScanChoice {
  scan_fn fn = nullptr;
  int L = 8;
  bool generic = false;
  size_t lds_extra = 0;  // bytes beyond the 4 lists
};

template <int L, int QPL, int METRIC, bool NT, bool RAGGED>
static scan_fn pick_mode(int mode) {
  switch (mode) {
    case 0: return scan_kernel<L, QPL, METRIC, NT, 0, RAGGED>;
    case 1: return scan_kernel<L, QPL, METRIC, NT, 1, RAGGED>;
    default: return scan_kernel<L, QPL, METRIC, NT, 2, RAGGED>;
  }
}

template <int L, int QPL, int METRIC>
static scan_fn pick_flags(bool nt, int mode, bool ragged) {
  if (ragged) return pick_mode<L, QPL, METRIC, true, true>(mode);  // ragged instances are non-temporal only
  return nt ? pick_mode<L, QPL, METRIC, true, false>(mode) : pick_mode<L, QPL, METRIC, false, false>(mode);
}

template <int L, int QPL>
static scan_fn pick_variant(int metric, bool nt, int reg, bool ragged) {
  return metric == WDBX_METRIC_COSINE ? pick_flags<L, QPL, WDBX_METRIC_COSINE>(nt, reg, ragged)
                                      : pick_flags<L, QPL, WDBX_METRIC_L2>(nt, reg, ragged);
}

template <int L>
static scan_fn pick_qpl(int qpl, int metric, bool nt, int reg, bool ragged) {
  if constexpr (L >= 16) {  // short rows on wide lane groups (d = 68 ... 256 when rows are not line aligned)
    if (qpl == 1) return pick_variant<L, 1>(metric, nt, reg, ragged);
    if (qpl == 2) return pick_variant<L, 2>(metric, nt, reg, ragged);
  }
  switch (qpl) {
    case 3: return pick_variant<L, 3>(metric, nt, reg, ragged);
    case 4: return pick_variant<L, 4>(metric, nt, reg, ragged);
    case 6: return pick_variant<L, 6>(metric, nt, reg, ragged);
    case 8: return pick_variant<L, 8>(metric, nt, reg, ragged);
    case 12: return pick_variant<L, 12>(metric, nt, reg, ragged);
    default: return nullptr;
  }
}

// short rows (d <= 48): L = 4 or 8 with 1..3 quads per lane, 4..8 passes in flight
template <int L>
static scan_fn pick_qpl_short(int qpl, int metric, bool nt, int reg, bool ragged) {
  switch (qpl) {
    case 1: return pick_variant<L, 1>(metric, nt, reg, ragged);
    case 2: return pick_variant<L, 2>(metric, nt, reg, ragged);
    case 3: return L == 4 ? pick_variant<4, 3>(metric, nt, reg, ragged) : nullptr;
    default: return nullptr;
  }
}

static scan_fn pick_specialised(int L, int qpl, int metric, bool nt, int reg, bool ragged) {
  if (L == 1 && qpl == 1) return pick_variant<1, 1>(metric, nt, reg, ragged);  // d <= 4: one lane per row
  if (L == 2 && qpl == 1) return pick_variant<2, 1>(metric, nt, reg, ragged);  // d <= 8
  if (L == 4) return pick_qpl_short<4>(qpl, metric, nt, reg, ragged);
  if (L == 8 && qpl <= 2) return pick_qpl_short<8>(qpl, metric, nt, reg, ragged);
  switch (L) {
    case 8: return pick_qpl<8>(qpl, metric, nt, reg, ragged);
    case 16: return pick_qpl<16>(qpl, metric, nt, reg, ragged);
    case 32: return pick_qpl<32>(qpl, metric, nt, reg, ragged);
    case 64: return pick_qpl<64>(qpl, metric, nt, reg, ragged);
    default: return nullptr;
  }
}

template <int L, int METRIC>
static scan_fn pick_generic_mode(int mode) {
  switch (mode) {
    case 0: return scan_kernel_generic<L, METRIC, 0>;
    case 1: return scan_kernel_generic<L, METRIC, 1>;
    default: return scan_kernel_generic<L, METRIC, 2>;
  }
}

template <int L>
static scan_fn pick_generic_metric(int metric, int mode) {
  return metric == WDBX_METRIC_COSINE ? pick_generic_mode<L, WDBX_METRIC_COSINE>(mode)
                                      : pick_generic_mode<L, WDBX_METRIC_L2>(mode);
}

static scan_fn pick_generic(int L, int metric, int reg) {
  switch (L) {
    case 64: return pick_generic_metric<64>(metric, reg);
    case 1: return pick_generic_metric<1>(metric, reg);
    case 2: return pick_generic_metric<2>(metric, reg);
    case 4: return pick_generic_metric<4>(metric, reg);
    default: return pick_generic_metric<8>(metric, reg);
  }
}

static bool use_select(const wdbx_index* ix, int k) { return ix->opt_select_min_k > 0 && k >= ix->opt_select_min_k; }

static ScanChoice choose_scan(const wdbx_index* ix, int k) {
  const int reg = use_select(ix, k) ? 2 : (k <= 128 && !ix->opt_lds_lists) ? 1 : 0;
  ScanChoice c;
  const int pitch4 = ix->pitch / 4;
  const bool nt = ix->opt_nt != 0;
  if (!ix->opt_generic && pitch4 < 16 && !ix->opt_lanes) {
    // short rows (d < 64): the smallest instance that holds the row, up to 3/8 of its slots idle
    const int cand[7][2] = {{1, 1}, {2, 1}, {4, 1}, {8, 1}, {4, 2}, {4, 3}, {8, 2}};
    for (int t = 0; t < 7; ++t) {
      const int slots = cand[t][0] * cand[t][1], waste = slots - pitch4;
      if (waste < 0 || waste * 8 > slots * 3) continue;
      scan_fn f = pick_specialised(cand[t][0], cand[t][1], ix->metric, nt, reg, waste != 0 || ix->opt_force_ragged);
      if (f) {
        c.fn = f;
        c.L = cand[t][0];
        return c;
      }
    }
  }
  if (!ix->opt_generic && pitch4 >= 16) {
    // unrolled instances exist for L in {8,16,32,64} x QPL in {3,4,6,8,12} (+ {1,2} for L >= 16); slots past the row end idle
    // (RAGGED form), at most a third of them.  Rows whose byte pitch is a multiple of 128 take the
    // instance with the fewest idle slots (smaller L on ties: all L measure alike there).  Rows that
    // do NOT start on cache-line boundaries take the LARGEST admissible L: a lane group then reads one
    // long contiguous span per instruction instead of many short ones that each straddle two lines
    // (d=300: L=8 5.35 TB/s, L=32 6.89 TB/s; d=200: L=8 5.87, L=16 6.84; profiles/r01/bench_dims.txt)
    const int Ls[4] = {8, 16, 32, 64}, Qs[7] = {1, 2, 3, 4, 6, 8, 12};
    const bool line_aligned = pitch4 % 8 == 0;
    int bestL = 0, bestQ = 0, best_waste = 1 << 30;
    // tier 0: QPL >= 3 (enough loads in flight per pass, several rows per pass).  tier 1, only for
    // misaligned short rows that tier 0 can serve with L = 8 at best: QPL 1 or 2 on L = 16 / 32
    // (d=100: (8,4) 5.63 TB/s, (32,1) 6.08 TB/s; for d=200 the wide short form is slower, 5.3 vs 6.6)
    for (int tier = 0; tier < 2; ++tier) {
      if (tier == 1 && (line_aligned || bestL >= 16)) break;
      for (int li = 0; li < 4; ++li) {
        if (ix->opt_lanes && Ls[li] != ix->opt_lanes) continue;
        if (tier == 1 && (Ls[li] < 16 || Ls[li] > 32)) continue;
        for (int qi = (tier == 0 ? 2 : 0); qi < (tier == 0 ? 7 : 2); ++qi) {
          const int slots = Ls[li] * Qs[qi], waste = slots - pitch4;
          if (waste < 0 || waste * 3 > slots) continue;
          const bool better = line_aligned ? waste < best_waste
                                           : (Ls[li] > bestL || (Ls[li] == bestL && waste < best_waste));
          if (better) {
            best_waste = waste;
            bestL = Ls[li];
            bestQ = Qs[qi];
          }
        }
      }
    }
    // a ragged instance may idle at most a third of its slots; beyond that the generic kernel is better
    if (bestL && best_waste * 3 <= bestL * bestQ) {
      scan_fn f = pick_specialised(bestL, bestQ, ix->metric, nt, reg, best_waste != 0 || ix->opt_force_ragged);
      if (f) {
        c.fn = f;
        c.L = bestL;
        return c;
      }
    }
  }
  int L = 1;
  while (L < 8 && L < pitch4) L <<= 1;
  if (pitch4 > 768) L = 64;  // rows longer than the largest unrolled instance
  c.fn = pick_generic(L, ix->metric, reg);
  c.L = L;
  c.generic = true;
  c.lds_extra = (size_t)pitch4 * 16;
  return c;
}

struct LaunchPlan {
  ScanChoice sc;
  uint32_t blocks = 0, P = 0, groups = 0, chunk = 0;
  bool wg_merge = false;
  size_t lds = 0;
};

static int plan_scan(wdbx_index* ix, int k, LaunchPlan* out) {
  LaunchPlan lp;
  lp.sc = choose_scan(ix, k);
  const int R = 64 / lp.sc.L;
  lp.groups = (uint32_t)((ix->n + R - 1) / R);
  lp.lds = (use_select(ix, k) ? 0 : (size_t)4 * k * sizeof(u64)) + lp.sc.lds_extra;
  if (lp.lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void*)lp.sc.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lp.lds));
  int per_cu = 0;
  HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)lp.sc.fn, 256, lp.lds));
  if (per_cu < 1) return fail(WDBX_E_INVALID, "scan kernel does not fit a CU at k=%d (LDS %zu B)", k, lp.lds);
  uint32_t blocks = (uint32_t)ix->cu_count * (uint32_t)std::min(per_cu, 2);  // 8 waves/CU x 8 KiB in flight: sweep in profiles/sweep_r01.txt
  if (ix->opt_blocks > 0) blocks = (uint32_t)ix->opt_blocks;
  // every wave should have a few passes of work; small corpora get a smaller grid
  const uint32_t min_groups_per_wave = 1;
  const uint32_t max_blocks = std::max<uint32_t>(1, (lp.groups + 4 * min_groups_per_wave - 1) / (4 * min_groups_per_wave));
  lp.blocks = std::max<uint32_t>(1, std::min(blocks, max_blocks));
  lp.wg_merge = ix->opt_wg_merge != 0;
  lp.P = lp.wg_merge ? lp.blocks : lp.blocks * 4;  // partial lists: one per workgroup or one per wave
  lp.chunk = ix->opt_blocked ? (lp.groups + lp.blocks * 4 - 1) / (lp.blocks * 4) : 0;
  *out = lp;
  return WDBX_OK;
}

static int merge_waves_for(int k) {
  const size_t budget = 128 * 1024;
  int nw = (int)(budget / ((size_t)k * sizeof(u64))) - 1;
  return std::max(1, std::min(16, nw));
}

static int record(EventPool& pool, bool enabled, hipStream_t s, bool start) {
  if (!enabled) return WDBX_OK;
  if (start) {
    if (pool.used + 2 > pool.ev.size()) {
      if (pool.ev.size() >= 2 * 65536) return WDBX_OK;  // pool exhausted: stop sampling
      for (int i = 0; i < 2; ++i) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        pool.ev.push_back(e);
      }
    }
    HIP_TRY(hipEventRecord(pool.ev[pool.used], s));
  } else if (pool.used + 2 <= pool.ev.size()) {
    HIP_TRY(hipEventRecord(pool.ev[pool.used + 1], s));
    pool.used += 2;
  }
  return WDBX_OK;
}

static int launch_merge(wdbx_index* ix, const MergeArgs& m, int nq) {
  const int nw = merge_waves_for(m.k);
  const size_t lds = (size_t)(nw + 1) * m.k * sizeof(u64);
  const bool reg = m.k <= 128 && !ix->opt_lds_lists;
  void (*fn)(MergeArgs) = reg ? merge_kernel<true> : merge_kernel<false>;
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int rc = record(ix->merge_ev, ix->profile, ix->stream, true);
  if (rc) return rc;
  hipLaunchKernelGGL(fn, dim3(nq), dim3(nw * 64), lds, ix->stream, m);
  HIP_TRY(hipGetLastError());
  return record(ix->merge_ev, ix->profile, ix->stream, false);
}

// all-gather this rank's key lists [b, k] (global rows) and merge the nranks lists per query
static int exchange_and_merge(wdbx_index* ix, int b, int k, int64_t* d_out_idx, float* d_out_score) {
  int rc = grow((void**)&ix->d_gathered, &ix->gathered_bytes, (size_t)ix->nranks * b * k * sizeof(u64));
  if (rc) return rc;
  // per-shard records [b, k] -> [nranks, b, k] on every rank (tiny: latency-bound)
  NCCL_TRY(ncclAllGather(ix->d_local_keys, ix->d_gathered, (size_t)b * k, ncclUint64, ix->comm, ix->stream));
  MergeArgs m = {};
  m.list_len = k;
  m.in = ix->d_gathered;
  m.q_stride = (uint64_t)k;
  m.i_stride = 1;
  m.p_stride = (uint64_t)b * k;
  m.P = (uint32_t)ix->nranks;
  m.k = k;
  m.metric = ix->metric;
  m.out_idx = d_out_idx;
  m.out_score = d_out_score;
  return launch_merge(ix, m, b);
}

// Enqueue nq searches.  Caller holds the handle's mutex and has made its device current.
// mode: 0 = final results of this shard alone; 1 = per-rank shard group (all-gather through the
// handle's communicator + second merge); 2 = only this shard's key list (global rows) into d_local_keys --
// the caller runs the exchange (in-process shard group, wdbx_group_search)
enum { SEARCH_FINAL = 0, SEARCH_SHARDED = 1, SEARCH_LOCAL_KEYS = 2 };

static int enqueue_search_gemm(wdbx_index* ix, const float* d_queries, int nq, int k, int64_t* d_out_idx, float* d_out_score,
                               int mode, int count_slot, u64* keys_out);
static bool shadow_single_eligible(const wdbx_index* ix, int k);
static bool u8_single_eligible(const wdbx_index* ix, int k);
static int enqueue_singles_u8(wdbx_index* ix, const float* d_queries, int nq, int k, int64_t* d_out_idx, float* d_out_score,
                              u64* keys_out);

static int enqueue_search(wdbx_index* ix, const float* d_queries, int nq, int k, int64_t* d_out_idx,
                          float* d_out_score, int mode) {
  const bool keys_only = mode == SEARCH_LOCAL_KEYS;
  const bool sharded = mode != SEARCH_FINAL;  // the local stage ends in keys with global rows
  if (nq <= 0) return WDBX_OK;
  if (k < 1 || k > WDBX_MAX_K) return fail(WDBX_E_INVALID, "k=%d outside [1, %d]", k, WDBX_MAX_K);
  if (!d_queries || (!keys_only && (!d_out_idx || !d_out_score))) return fail(WDBX_E_INVALID, "null device buffer");
  if (mode == SEARCH_SHARDED && !ix->comm) return fail(WDBX_E_STATE, "sharded search before wdbx_index_comm_init");
  if (ix->n >= 0xFFFFFF00ull) return fail(WDBX_E_INVALID, "shard holds too many rows for 32-bit row keys");

  const int batch = keys_only ? nq : (int)std::max<int64_t>(1, std::min<int64_t>(ix->opt_batch, 1024));
  int rc;
  if (ix->n == 0) {
    // empty shard: every local list is empty (the reference returns [] at indexing.py:998)
    rc = grow((void**)&ix->d_local_keys, &ix->local_keys_bytes, (size_t)batch * k * sizeof(u64));
    if (rc) return rc;
  }
  LaunchPlan lp;
  const bool select = ix->n && use_select(ix, k);
  if (ix->n) {
    rc = plan_scan(ix, k, &lp);
    if (rc) return rc;
    if (select) {
      if ((rc = grow((void**)&ix->d_dump, &ix->dump_bytes, (size_t)ix->n * sizeof(u64)))) return rc;
      if ((rc = grow((void**)&ix->d_sel, &ix->sel_bytes, (size_t)WDBX_MAX_K * sizeof(u64)))) return rc;
      if ((rc = grow((void**)&ix->d_state, &ix->state_bytes, sizeof(SelectState)))) return rc;
    } else {
      rc = grow((void**)&ix->d_partials, &ix->partials_bytes, (size_t)batch * k * lp.P * sizeof(u64));
      if (rc) return rc;
    }
  }
  if (sharded) {
    rc = grow((void**)&ix->d_local_keys, &ix->local_keys_bytes, (size_t)batch * k * sizeof(u64));
    if (rc) return rc;
    if (!keys_only) {
      rc = grow((void**)&ix->d_gathered, &ix->gathered_bytes, (size_t)ix->nranks * batch * k * sizeof(u64));
      if (rc) return rc;
    }
  }

  for (int q0 = 0; q0 < nq; q0 += batch) {
    const int b = std::min(batch, nq - q0);
    if (select) {
      // large k: per query  scan (key per row) -> radix select -> compact -> sort
      const uint32_t sgrid = (uint32_t)std::min<uint64_t>((ix->n + 255) / 256, (uint64_t)ix->cu_count * 16);
      uint32_t npow2 = 2;
      while (npow2 < (uint32_t)k) npow2 <<= 1;
      for (int q = 0; q < b; ++q) {
        ScanArgs sa = {};
        sa.rows = (const f4*)ix->d_rows;
        sa.query = (const f4*)(d_queries + (size_t)(q0 + q) * ix->pitch);
        sa.partials = ix->d_dump;
        sa.mask = ix->active_mask;
        sa.n_rows = (uint32_t)ix->n;
        sa.pitch4 = (uint32_t)(ix->pitch / 4);
        sa.groups = lp.groups;
        sa.chunk = lp.chunk;
        sa.k = k;
        sa.wg_merge = lp.wg_merge ? 1 : 0;
        if ((rc = record(ix->scan_ev, ix->profile, ix->stream, true))) return rc;
        hipLaunchKernelGGL(lp.sc.fn, dim3(lp.blocks), dim3(256), lp.lds, ix->stream, sa);
        HIP_TRY(hipGetLastError());
        if ((rc = record(ix->scan_ev, ix->profile, ix->stream, false))) return rc;
        if ((rc = record(ix->merge_ev, ix->profile, ix->stream, true))) return rc;
        hipLaunchKernelGGL(select_init_kernel, dim3(1), dim3(256), 0, ix->stream, ix->d_state, (uint32_t)k);
        for (int shift = 56; shift >= 0; shift -= 8) {
          hipLaunchKernelGGL(radix_hist_kernel, dim3(sgrid), dim3(256), 0, ix->stream, (const u64*)ix->d_dump, (u64)ix->n,
                             ix->d_state, shift);
          hipLaunchKernelGGL(radix_pick_kernel, dim3(1), dim3(256), 0, ix->stream, ix->d_state, shift);
        }
        hipLaunchKernelGGL(radix_compact_kernel, dim3(sgrid), dim3(256), 0, ix->stream, (const u64*)ix->d_dump, (u64)ix->n,
                           ix->d_state, ix->d_sel, (uint32_t)k);
        MergeArgs m = {};
        m.k = k;
        m.metric = ix->metric;
        m.row_base = (uint32_t)ix->row_base;
        m.out_keys = sharded ? ix->d_local_keys + (size_t)q * k : nullptr;
        m.out_idx = sharded ? nullptr : d_out_idx + (size_t)(q0 + q) * k;
        m.out_score = sharded ? nullptr : d_out_score + (size_t)(q0 + q) * k;
        hipLaunchKernelGGL(sort_out_kernel, dim3(1), dim3(1024), (size_t)npow2 * sizeof(u64), ix->stream,
                           (const u64*)ix->d_sel, (const SelectState*)ix->d_state, m, npow2);
        HIP_TRY(hipGetLastError());
        if ((rc = record(ix->merge_ev, ix->profile, ix->stream, false))) return rc;
      }
    } else if (ix->n) {
      // Single queries over the bf16 shadow: each query makes ITS OWN selection pass over the half-size copy
      // (threshold from a sample, candidates above threshold - error margin, exact fp32 re-scoring: the
      // batched path's pipeline with one query), which reads half the bytes of the fp32 scan.  The fp32
      // scan and its merge still follow, but as REPAIR launches that return at once unless that query's
      // candidate buffer overflowed (massive near-duplicates) -- the result is exact either way, without a
      // host round trip.
      const bool u8 = !keys_only && u8_single_eligible(ix, k);
      const bool shadow = u8 || (!keys_only && shadow_single_eligible(ix, k));
      ix->last_single_path = u8 ? 2 : shadow ? 1 : 0;
      if (shadow) {
        if ((rc = grow((void**)&ix->d_count, &ix->count_bytes, ((size_t)batch + 2 * GB_N) * sizeof(uint32_t)))) return rc;
        if (u8)  // the u8 selection scan: a quarter of the fp32 bytes per query
          rc = enqueue_singles_u8(ix, d_queries + (size_t)q0 * ix->pitch, b, k, d_out_idx + (size_t)q0 * k,
                                  d_out_score + (size_t)q0 * k, sharded ? ix->d_local_keys : nullptr);
        else     // the bf16 tile kernel with one live column: half the fp32 bytes
          rc = enqueue_search_gemm(ix, d_queries + (size_t)q0 * ix->pitch, b, k, d_out_idx + (size_t)q0 * k,
                                   d_out_score + (size_t)q0 * k, SEARCH_FINAL, 0, sharded ? ix->d_local_keys : nullptr);
        if (rc) return rc;
      }
      for (int q = 0; q < b; ++q) {
        ScanArgs sa = {};
        if (shadow) {
          sa.only_if_over = ix->d_count + q;
          sa.over_cap = ix->last_batch_cap;
        }
        sa.rows = (const f4*)ix->d_rows;
        sa.query = (const f4*)(d_queries + (size_t)(q0 + q) * ix->pitch);
        sa.partials = ix->d_partials + (size_t)q * k * lp.P;
        sa.mask = ix->active_mask;
        sa.n_rows = (uint32_t)ix->n;
        sa.pitch4 = (uint32_t)(ix->pitch / 4);
        sa.groups = lp.groups;
        sa.chunk = lp.chunk;
        sa.k = k;
        sa.wg_merge = lp.wg_merge ? 1 : 0;
        // (repair launches are not timed: they would read as scans of zero length)
        rc = shadow ? WDBX_OK : record(ix->scan_ev, ix->profile, ix->stream, true);
        if (rc) return rc;
        hipLaunchKernelGGL(lp.sc.fn, dim3(lp.blocks), dim3(256), lp.lds, ix->stream, sa);
        HIP_TRY(hipGetLastError());
        rc = shadow ? WDBX_OK : record(ix->scan_ev, ix->profile, ix->stream, false);
        if (rc) return rc;
      }
      MergeArgs m = {};
      if (shadow) {
        m.only_if_over = ix->d_count;
        m.over_cap = ix->last_batch_cap;
      }
      m.list_len = k;
      m.in = ix->d_partials;
      m.q_stride = (uint64_t)k * lp.P;
      m.i_stride = lp.P;
      m.p_stride = 1;
      m.P = lp.P;
      m.k = k;
      m.metric = ix->metric;
      m.row_base = (uint32_t)ix->row_base;
      m.idx_base = 0;
      m.out_keys = sharded ? ix->d_local_keys : nullptr;
      m.out_idx = sharded ? nullptr : d_out_idx + (size_t)q0 * k;
      m.out_score = sharded ? nullptr : d_out_score + (size_t)q0 * k;
      rc = launch_merge(ix, m, b);
      if (rc) return rc;
    } else if (sharded) {
      HIP_TRY(hipMemsetAsync(ix->d_local_keys, 0, (size_t)b * k * sizeof(u64), ix->stream));
    } else {
      // no rows: idx = -1 (all bits set), score = 0
      HIP_TRY(hipMemsetAsync(d_out_idx + (size_t)q0 * k, 0xFF, (size_t)b * k * sizeof(int64_t), ix->stream));
      HIP_TRY(hipMemsetAsync(d_out_score + (size_t)q0 * k, 0, (size_t)b * k * sizeof(float), ix->stream));
    }
    if (mode == SEARCH_SHARDED) {
      rc = exchange_and_merge(ix, b, k, d_out_idx + (size_t)q0 * k, d_out_score + (size_t)q0 * k);
      if (rc) return rc;
    }
  }
  return WDBX_OK;
}


// ---- batched queries on the MFMA path ----------------------------------------------------------
static bool gemm_eligible(const wdbx_index* ix, int nq, int k) {
  if (ix->metric == WDBX_METRIC_L2 && !ix->opt_gemm_l2) return false;
  return nq >= ix->opt_gemm_min_nq && (int64_t)ix->n >= ix->opt_gemm_min_rows && (uint64_t)k * 8 * GB_M <= ix->n;
}

// single queries take the shadow selection pipeline (see enqueue_search) when the bf16 shadow is in use, no row
// mask is active (the tiles do not read masks) and k is served by the list kernels (the repair launch)
static bool shadow_single_eligible(const wdbx_index* ix, int k) {
  if (ix->opt_scan_shadow <= 0 || ix->opt_gemm_bf16 < 2 || ix->active_mask || use_select(ix, k)) return false;
  if (ix->metric == WDBX_METRIC_L2 && !ix->opt_gemm_l2) return false;
  // the shadow pads rows to 128 elements: for short rows it is no smaller than the fp32 rows (d = 32: twice
  // the bytes, measured 0.54x; d = 64: 0.98x; d = 100: 1.4x) -- worth it from 0.8 of the fp32 bytes down
  const uint64_t pitch16 = ((uint64_t)ix->pitch + 127) / 128 * 128;
  if (pitch16 * 2 * 10 > (uint64_t)ix->pitch * 4 * 8) return false;
  return (int64_t)ix->n >= ix->opt_gemm_min_rows && (uint64_t)k * 8 * GB_M <= ix->n;
}

// ---- single queries on the u8 selection scan ---------------------------------------------------
// row shapes the scan8 kernel is instantiated for: pieces (16 bytes each) per row = L lanes x QPL loads
struct Scan8Shape { uint32_t pieces; int L, QPL; };
static const Scan8Shape kScan8Shapes[] = {{8, 8, 1},   {16, 8, 2},  {24, 8, 3},   {32, 16, 2},  {48, 16, 3},
                                           {64, 32, 2}, {96, 32, 3}, {128, 64, 2}, {192, 64, 3}, {256, 64, 4}};
// the smallest instantiated shape that holds a row of `dim` elements (its padded byte pitch = pieces * 16)
static const Scan8Shape* scan8_shape(uint32_t dim) {
  for (const Scan8Shape& sh : kScan8Shapes)
    if (sh.pieces * 16 >= dim) return &sh;
  return nullptr;
}

static bool u8_single_eligible(const wdbx_index* ix, int k) {
  if (ix->opt_scan_shadow < 2 || use_select(ix, k)) return false;  // (row masks are honoured by the u8 scan)
  const Scan8Shape* sh = scan8_shape((uint32_t)ix->dim);
  // worth it from 0.6 of the fp32 bytes down (d = 32 would read as many bytes as the fp32 row)
  if (!sh || (uint64_t)sh->pieces * 16 * 10 > (uint64_t)ix->pitch * 4 * 6) return false;
  return (int64_t)ix->n >= ix->opt_gemm_min_rows && (uint64_t)k * 8 * GB_M <= ix->n;
}

typedef void (*scan8_fn)(Scan8Args);
template <int PHASE, int METRIC>
static scan8_fn pick_scan8(int L, int QPL) {
  switch (L * 10 + QPL) {
    case 81: return scan8_kernel<8, 1, METRIC, PHASE>;
    case 82: return scan8_kernel<8, 2, METRIC, PHASE>;
    case 83: return scan8_kernel<8, 3, METRIC, PHASE>;
    case 162: return scan8_kernel<16, 2, METRIC, PHASE>;
    case 163: return scan8_kernel<16, 3, METRIC, PHASE>;
    case 322: return scan8_kernel<32, 2, METRIC, PHASE>;
    case 323: return scan8_kernel<32, 3, METRIC, PHASE>;
    case 642: return scan8_kernel<64, 2, METRIC, PHASE>;
    case 643: return scan8_kernel<64, 3, METRIC, PHASE>;
    case 644: return scan8_kernel<64, 4, METRIC, PHASE>;
  }
  return nullptr;
}

// nq single queries, each with its own sample pass + full pass over the u8 shadow; thresholds, re-scoring and
// the final top-k run once per round of 32 queries.  Candidate counters at d_count[0 .. nq) (sized by the caller).
static int enqueue_singles_u8(wdbx_index* ix, const float* d_queries, int nq, int k, int64_t* d_out_idx, float* d_out_score,
                              u64* keys_out) {
  const Scan8Shape* sh = scan8_shape((uint32_t)ix->dim);
  if (!sh) return fail(WDBX_E_STATE, "no u8 scan instance for dim %d", ix->dim);
  const bool l2 = ix->metric == WDBX_METRIC_L2;
  const uint32_t pitch8 = sh->pieces * 16;
  int rc;
  {  // the u8 shadow and scales of the rows added or overwritten since the last search
    const size_t need = (size_t)ix->cap * pitch8, need_s = (size_t)ix->cap * sizeof(float);
    if (ix->rows8_bytes < need || ix->scale8_bytes < need_s || ix->pitch8 != pitch8) {
      if (ix->d_rows8) (void)hipFree(ix->d_rows8);
      if (ix->d_scale8) (void)hipFree(ix->d_scale8);
      ix->d_rows8 = nullptr;
      ix->d_scale8 = nullptr;
      ix->rows8_bytes = ix->scale8_bytes = 0;
      ix->shadow8_rows = 0;
      HIP_TRY(hipMalloc((void**)&ix->d_rows8, need));
      ix->rows8_bytes = need;
      HIP_TRY(hipMalloc((void**)&ix->d_scale8, need_s));
      ix->scale8_bytes = need_s;
      ix->pitch8 = pitch8;
    }
    if (ix->shadow8_rows < ix->n) {
      const uint32_t blocks = (uint32_t)std::min<uint64_t>((ix->n - ix->shadow8_rows + 3) / 4, 65536);
      hipLaunchKernelGGL(rows_to_u8_kernel, dim3(blocks), dim3(256), 0, ix->stream, (const float*)ix->d_rows, (u64)ix->shadow8_rows,
                         (u64)ix->n, (uint32_t)ix->dim, (uint32_t)ix->pitch, ix->d_rows8, pitch8, ix->d_scale8);
      HIP_TRY(hipGetLastError());
      ix->shadow8_rows = ix->n;
    }
  }
  if (l2) {  // squared fp32 norms of the rows (the 2 w - |c|^2 form)
    if ((rc = grow((void**)&ix->d_cnmax, &ix->cnmax_bytes, sizeof(uint32_t)))) return rc;
    if (ix->cn_bytes < (size_t)ix->n * sizeof(float)) {
      if ((rc = grow((void**)&ix->d_cn, &ix->cn_bytes, (size_t)ix->cap * sizeof(float)))) return rc;
      ix->cn_rows = 0;
    }
    if (ix->cn_rows == 0) HIP_TRY(hipMemsetAsync(ix->d_cnmax, 0, sizeof(uint32_t), ix->stream));
    if (ix->cn_rows < ix->n) {
      const uint32_t blocks = (uint32_t)std::min<uint64_t>((ix->n - ix->cn_rows + 3) / 4, 65536);
      hipLaunchKernelGGL(row_sqnorm_kernel, dim3(blocks), dim3(256), 0, ix->stream, (const float*)ix->d_rows, (u64)ix->cn_rows,
                         (u64)ix->n, (uint32_t)ix->pitch, ix->d_cn, ix->d_cnmax);
      HIP_TRY(hipGetLastError());
      ix->cn_rows = ix->n;
    }
  }
  // sampled 256-row tiles (4 groups of 64 rows each), as on the tile path
  const uint32_t tiles = (uint32_t)((ix->n + 255) / 256);
  const uint32_t div = ix->opt_gemm_sample_div > 0 ? (uint32_t)ix->opt_gemm_sample_div : std::min(32u, std::max(4u, 1024u / (uint32_t)k));
  uint32_t sample_tiles = std::max<uint32_t>(tiles / div, (8u * k + 3) / 4);
  sample_tiles = std::max<uint32_t>(1, std::min(sample_tiles, tiles));
  const uint32_t stride = tiles / sample_tiles, ngroups = 4 * sample_tiles;
  if (ngroups < (uint32_t)k) return fail(WDBX_E_STATE, "corpus too small for the selection scan at k=%d", k);
  const uint64_t expect = (uint64_t)k * (tiles / sample_tiles + 1);
  const uint32_t cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(4096, expect * 32), 1u << 22);
  constexpr int ROUND = 32;
  if ((rc = grow((void**)&ix->d_halfmax, &ix->halfmax_bytes, (size_t)ROUND * ngroups * sizeof(u64)))) return rc;
  if ((rc = grow((void**)&ix->d_tau, &ix->tau_bytes, (size_t)GB_N * sizeof(float)))) return rc;
  if ((rc = grow((void**)&ix->d_cand, &ix->cand_bytes, (size_t)ROUND * cap * sizeof(u64)))) return rc;
  if ((rc = grow((void**)&ix->d_qinfo, &ix->qinfo_bytes, (size_t)ROUND * 2 * sizeof(float)))) return rc;
  if (ix->count_bytes < ((size_t)nq + GB_N) * sizeof(uint32_t)) return fail(WDBX_E_STATE, "candidate counters not sized by the caller");
  HIP_TRY(hipMemsetAsync(ix->d_count, 0, (size_t)nq * sizeof(uint32_t), ix->stream));
  ix->last_batch_nq = (uint32_t)nq;
  ix->last_batch_cap = cap;
  scan8_fn f0 = l2 ? pick_scan8<0, WDBX_METRIC_L2>(sh->L, sh->QPL) : pick_scan8<0, WDBX_METRIC_COSINE>(sh->L, sh->QPL);
  scan8_fn f1 = l2 ? pick_scan8<1, WDBX_METRIC_L2>(sh->L, sh->QPL) : pick_scan8<1, WDBX_METRIC_COSINE>(sh->L, sh->QPL);
  if (!f0 || !f1) return fail(WDBX_E_STATE, "no u8 scan instance for %d lanes x %d loads", sh->L, sh->QPL);
  const uint32_t R = 64u / (uint32_t)sh->L;
  const uint32_t groups1 = (uint32_t)((ix->n + R - 1) / R);
  const uint32_t wgs = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(ix->opt_scan8_wgs, 8));  // workgroups per CU
  const uint32_t grid1 = std::min<uint32_t>((groups1 + 3) / 4, (uint32_t)ix->cu_count * wgs);
  const uint32_t grid0 = std::min<uint32_t>((ngroups + 3) / 4, (uint32_t)ix->cu_count * 4);
  const size_t pitch4 = ix->pitch / 4;

  for (int q0 = 0; q0 < nq; q0 += ROUND) {
    const int nv = std::min(ROUND, nq - q0);
    const float* qsrc = d_queries + (size_t)q0 * ix->pitch;
    hipLaunchKernelGGL(query_info_kernel, dim3(nv), dim3(64), 0, ix->stream, qsrc, (uint32_t)ix->pitch, nv, ix->d_qinfo);
    HIP_TRY(hipGetLastError());
    Scan8Args a = {};
    a.rows8 = (const u4v*)ix->d_rows8;
    a.scale = ix->d_scale8;
    a.cn = ix->d_cn;
    a.n_rows = (uint32_t)ix->n;
    a.mask = ix->active_mask;
    a.pieces = sh->pieces;
    a.qquads = (uint32_t)pitch4;
    a.num_tiles = sample_tiles;
    a.tile_stride = stride;
    a.cap = cap;
    // phase 0, all queries of the round in one launch: maxima of the lower bounds over the sampled groups
    a.query = (const f4*)qsrc;
    a.qinfo = ix->d_qinfo;
    a.halfmax = ix->d_halfmax;
    if ((rc = record(ix->sample_ev, ix->profile, ix->stream, true))) return rc;
    hipLaunchKernelGGL(f0, dim3(grid0, nv), dim3(256), 0, ix->stream, a);
    HIP_TRY(hipGetLastError());
    if ((rc = record(ix->sample_ev, ix->profile, ix->stream, false))) return rc;
    MergeArgs m = {};
    m.in = ix->d_halfmax;
    m.q_stride = ngroups;
    m.i_stride = 0;
    m.p_stride = 1;
    m.P = ngroups;
    m.list_len = 1;
    m.k = k;
    m.metric = ix->metric;
    m.out_kth = ix->d_tau;  // = a rigorous lower bound of each query's true k-th best score
    if ((rc = launch_merge(ix, m, nv))) return rc;
    for (int i = 0; i < nv; ++i) {  // phase 1: every row whose upper bound reaches the threshold
      a.query = (const f4*)(qsrc + (size_t)i * ix->pitch);
      a.qinfo = ix->d_qinfo + 2 * i;
      a.tau = ix->d_tau + i;
      a.cand = ix->d_cand + (size_t)i * cap;
      a.count = ix->d_count + q0 + i;
      if ((rc = record(ix->gemm_ev, ix->profile, ix->stream, true))) return rc;
      hipLaunchKernelGGL(f1, dim3(grid1), dim3(256), 0, ix->stream, a);
      HIP_TRY(hipGetLastError());
      if ((rc = record(ix->gemm_ev, ix->profile, ix->stream, false))) return rc;
    }
    // exact fp32 scores for the candidates, from the fp32 rows
    hipLaunchKernelGGL(l2 ? rescore_kernel<WDBX_METRIC_L2> : rescore_kernel<WDBX_METRIC_COSINE>, dim3(256, nv), dim3(256), 0,
                       ix->stream, (const f4*)ix->d_rows, (uint32_t)pitch4, (const f4*)qsrc, ix->d_cand,
                       (const uint32_t*)(ix->d_count + q0), cap);
    HIP_TRY(hipGetLastError());
    MergeArgs f = {};
    f.in = ix->d_cand;
    f.q_stride = cap;
    f.i_stride = 0;
    f.p_stride = 1;
    f.P = cap;
    f.P_dev = ix->d_count + q0;
    f.list_len = 1;
    f.k = k;
    f.metric = ix->metric;
    if (keys_out) {
      f.row_base = (uint32_t)ix->row_base;
      f.out_keys = keys_out + (size_t)q0 * k;
    } else {
      f.out_idx = d_out_idx + (size_t)q0 * k;
      f.out_score = d_out_score + (size_t)q0 * k;
    }
    if ((rc = launch_merge(ix, f, nv))) return rc;
  }
  return WDBX_OK;
}

// tile kernel families of the batched path (option gemm_bf16): 0 = exact fp32 tiles, 1 = bf16 selection tiles
// reading the fp32 rows, 2 = bf16 selection tiles reading the bf16 shadow copy (falls back to 1 when the
// shadow does not fit in device memory)
enum { GEMM_FP32 = 0, GEMM_BF16 = 1, GEMM_BF16_SHADOW = 2 };
static inline int gemm_family(const wdbx_index* ix) {
  return ix->opt_gemm_bf16 <= 0 ? GEMM_FP32 : ix->opt_gemm_bf16 == 1 ? GEMM_BF16 : GEMM_BF16_SHADOW;
}
static inline uint32_t gemm_tile_rows(int family) { return family == GEMM_FP32 ? GB_M : GW_M; }

template <int PHASE, int CT, int METRIC>
static void (*pick_gemm_kernel(int family, bool ktail))(GemmArgs) {
  if constexpr (CT >= 2) {
    if (family == GEMM_BF16_SHADOW) return gemm_bf16w8_kernel<PHASE, false, CT, METRIC, true>;
    if (family == GEMM_BF16)
      return ktail ? gemm_bf16w8_kernel<PHASE, true, CT, METRIC, false> : gemm_bf16w8_kernel<PHASE, false, CT, METRIC, false>;
  }
  return ktail ? gemm_topk_kernel<PHASE, true, CT, METRIC> : gemm_topk_kernel<PHASE, false, CT, METRIC>;
}

template <int PHASE, int CT>
static int launch_gemm_ct(wdbx_index* ix, const GemmArgs& g, int family) {
  if (family != GEMM_FP32 && CT < 2) return fail(WDBX_E_STATE, "bf16 tiles need a query block of at least 128");
  const int bk = family == GEMM_BF16_SHADOW ? 64 : 32;  // elements per LDS chunk
  const uint32_t tile_rows = gemm_tile_rows(family);
  const size_t lds = family == GEMM_FP32 ? (size_t)(2 * GB_M + 2 * 64 * CT) * 36 * sizeof(float)
                                         : (size_t)(2 * tile_rows + 2 * 64 * CT) * (bk * 2 + 16);
  // K tail: the fp32 tiles step 8 quads at a time, the bf16 tiles a pair of chunks of 8 quads (never on the
  // shadow, which is padded)
  const bool ktail = family != GEMM_BF16_SHADOW && (g.pitch4 % (family == GEMM_FP32 ? 8u : 16u)) != 0;
  const bool l2 = ix->metric == WDBX_METRIC_L2;
  void (*fn)(GemmArgs) = l2 ? pick_gemm_kernel<PHASE, CT, WDBX_METRIC_L2>(family, ktail)
                            : pick_gemm_kernel<PHASE, CT, WDBX_METRIC_COSINE>(family, ktail);
  HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // fp32 tiles, CT = 1, 2: the tile's LDS footprint (55 / 74 KiB) lets two workgroups share a CU.
  // bf16 tiles: one 8-wave workgroup per CU (two waves per SIMD).
  const uint32_t per_cu = (family != GEMM_FP32 || CT == 4) ? 1 : 2;
  const uint32_t grid = std::min<uint32_t>(g.num_tiles, (uint32_t)ix->cu_count * per_cu);
  int rc = record(ix->gemm_ev, ix->profile, ix->stream, true);
  if (rc) return rc;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(family == GEMM_FP32 ? 256 : 512), lds, ix->stream, g);
  HIP_TRY(hipGetLastError());
  return record(ix->gemm_ev, ix->profile, ix->stream, false);
}

template <int PHASE>
static int launch_gemm(wdbx_index* ix, const GemmArgs& g, int ct, int family) {
  switch (ct) {
    case 1: return launch_gemm_ct<PHASE, 1>(ix, g, family);
    case 2: return launch_gemm_ct<PHASE, 2>(ix, g, family);
    default: return launch_gemm_ct<PHASE, 4>(ix, g, family);
  }
}

// Enqueue nq (any number) queries in blocks of 256 through the GEMM path.  Per query a counter of
// appended candidates is left in d_count[q]; a count above the capacity means that query's result
// may be incomplete and must be re-run on the scan path (wdbx_index_batch_status).
// count_slot >= 0: the per-query candidate counters live at d_count[count_slot ...] (sized by the caller) and
// only they are reset; keys_out != null: the final top-k is written there as keys with global rows instead
// of idx/score, and no exchange follows (the single-query caller batches its own).
static int enqueue_search_gemm(wdbx_index* ix, const float* d_queries, int nq, int k, int64_t* d_out_idx,
                               float* d_out_score, int mode = SEARCH_FINAL, int count_slot = -1, u64* keys_out = nullptr) {
  const bool sharded = mode == SEARCH_SHARDED;
  if (sharded && !ix->comm) return fail(WDBX_E_STATE, "sharded search before wdbx_index_comm_init");
  if (nq <= 0) return WDBX_OK;
  if (k < 1 || k > WDBX_MAX_K) return fail(WDBX_E_INVALID, "k=%d outside [1, %d]", k, WDBX_MAX_K);
  if (!d_queries || !d_out_idx || !d_out_score) return fail(WDBX_E_INVALID, "null device buffer");
  if (ix->n >= 0xFFFFFF00ull) return fail(WDBX_E_INVALID, "shard holds too many rows for 32-bit row keys");
  int rc;
  int family = gemm_family(ix);
  // short rows: the padded shadow would be no smaller than the fp32 rows, so the tiles read those
  if (family == GEMM_BF16_SHADOW && ((uint64_t)ix->pitch + 127) / 128 * 128 >= 2 * (uint64_t)ix->pitch) family = GEMM_BF16;
  if (family == GEMM_BF16_SHADOW) {  // the bf16 shadow copy of the rows added since the last batch
    const uint32_t pitch16 = (uint32_t)((ix->pitch + 127) / 128 * 128);  // whole pairs of 64-element chunks
    const size_t need = (size_t)ix->cap * pitch16 * 2;
    if (ix->rows16_bytes < need || ix->pitch16 != pitch16) {
      if (ix->d_rows16) (void)hipFree(ix->d_rows16);
      ix->d_rows16 = nullptr;
      ix->rows16_bytes = 0;
      ix->shadow_rows = 0;
      if (hipMalloc(&ix->d_rows16, need) == hipSuccess) {
        ix->rows16_bytes = need;
        ix->pitch16 = pitch16;
      } else {
        (void)hipGetLastError();  // no room for the shadow: the same tiles on the fp32 rows
        family = GEMM_BF16;
      }
    }
    if (family == GEMM_BF16_SHADOW && ix->shadow_rows < ix->n) {
      const u64 pieces = (ix->n - ix->shadow_rows) * (pitch16 / 8);
      hipLaunchKernelGGL(rows_to_bf16_kernel, dim3((uint32_t)std::min<u64>((pieces + 255) / 256, 1u << 20)), dim3(256), 0,
                         ix->stream, (const float*)ix->d_rows, (u64)ix->shadow_rows, (u64)ix->n, (uint32_t)ix->pitch,
                         (__bf16*)ix->d_rows16, pitch16);
      HIP_TRY(hipGetLastError());
      ix->shadow_rows = ix->n;
    }
  }
  ix->last_gemm_mode = family;
  const bool l2 = ix->metric == WDBX_METRIC_L2, bf16 = family != GEMM_FP32;
  const bool inexact = l2 || bf16;  // selection scores differ from the final ones: margin + exact re-scoring
  const uint32_t tile_rows = gemm_tile_rows(family), rw = tile_rows / 64;  // rw: PHASE 0 keys per tile and query
  const uint32_t tiles = (uint32_t)((ix->n + tile_rows - 1) / tile_rows);
  // sampled fraction 1/div: the bf16 tiles make the sample pass cheap and their error margin multiplies the
  // candidates, so a larger k gets a larger sample (a tighter threshold) there
  const uint32_t div = ix->opt_gemm_sample_div > 0 ? (uint32_t)ix->opt_gemm_sample_div
                       : bf16 ? std::min(32u, std::max(4u, 1024u / (uint32_t)k)) : 32u;
  uint32_t sample_tiles = std::max<uint32_t>(tiles / div, (8u * k + rw - 1) / rw);
  sample_tiles = std::max<uint32_t>(1, std::min(sample_tiles, tiles));
  const uint32_t stride = tiles / sample_tiles;
  if (rw * sample_tiles < (uint32_t)k) return fail(WDBX_E_STATE, "corpus too small for the batched path at k=%d", k);
  // expected candidates per query ~ k * tiles / sample_tiles; capacity leaves a wide margin
  const uint64_t expect = (uint64_t)k * (tiles / sample_tiles + 1);
  const uint32_t cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(4096, expect * (bf16 ? 32 : 8)), 1u << 22);
  const size_t pitch4 = ix->pitch / 4;
  if ((rc = grow((void**)&ix->d_qblock, &ix->qblock_bytes, (size_t)GB_N * ix->pitch * sizeof(float)))) return rc;
  if ((rc = grow((void**)&ix->d_halfmax, &ix->halfmax_bytes, (size_t)GB_N * rw * sample_tiles * sizeof(u64)))) return rc;
  if ((rc = grow((void**)&ix->d_tau, &ix->tau_bytes, (size_t)GB_N * sizeof(float)))) return rc;
  if ((rc = grow((void**)&ix->d_cand, &ix->cand_bytes, (size_t)GB_N * cap * sizeof(u64)))) return rc;
  if (count_slot < 0) {
    if ((rc = grow((void**)&ix->d_count, &ix->count_bytes, ((size_t)nq + GB_N) * sizeof(uint32_t)))) return rc;
  } else if (ix->count_bytes < ((size_t)count_slot + nq + GB_N) * sizeof(uint32_t)) {
    return fail(WDBX_E_STATE, "candidate counters not sized by the caller");
  }
  uint32_t* const d_count = ix->d_count + std::max(count_slot, 0);
  if (sharded && (rc = grow((void**)&ix->d_local_keys, &ix->local_keys_bytes, (size_t)GB_N * k * sizeof(u64)))) return rc;
  HIP_TRY(hipMemsetAsync(d_count, 0, (count_slot < 0 ? (size_t)nq + GB_N : (size_t)nq) * sizeof(uint32_t), ix->stream));
  // bf16 query block: zero padded to the K extent the tile kernel walks (a ring of 4 chunks / a pair of chunks)
  const uint32_t kring = family == GEMM_BF16_SHADOW ? 128u : 64u;
  const uint32_t kpad = (uint32_t)((ix->pitch + kring - 1) / kring * kring);
  if (bf16 && (rc = grow((void**)&ix->d_qb16, &ix->qb16_bytes, (size_t)GB_N * kpad * 2))) return rc;
  if (inexact) {  // squared norms of the rows added since the last such batch (L2 term, and the error margin)
    if ((rc = grow((void**)&ix->d_cnmax, &ix->cnmax_bytes, sizeof(uint32_t)))) return rc;
    if (ix->cn_bytes < (size_t)ix->n * sizeof(float)) {
      if ((rc = grow((void**)&ix->d_cn, &ix->cn_bytes, (size_t)ix->cap * sizeof(float)))) return rc;
      ix->cn_rows = 0;
    }
    if (ix->cn_rows == 0) HIP_TRY(hipMemsetAsync(ix->d_cnmax, 0, sizeof(uint32_t), ix->stream));
    if (ix->cn_rows < ix->n) {
      const uint32_t blocks = (uint32_t)std::min<uint64_t>((ix->n - ix->cn_rows + 3) / 4, 65536);
      hipLaunchKernelGGL(row_sqnorm_kernel, dim3(blocks), dim3(256), 0, ix->stream, (const float*)ix->d_rows, (u64)ix->cn_rows,
                         (u64)ix->n, (uint32_t)ix->pitch, ix->d_cn, ix->d_cnmax);
      HIP_TRY(hipGetLastError());
      ix->cn_rows = ix->n;
    }
  }
  ix->last_batch_nq = (uint32_t)nq;
  ix->last_batch_cap = cap;

  // per_query: the single-query caller.  Every query makes its OWN pass pair (a 128-column tile with one live
  // column), but the small kernels around the passes (conversion, thresholds, margins, re-scoring, final
  // top-k) run once per round of up to PQ_ROUND queries.
  const bool per_query = count_slot >= 0;
  constexpr int PQ_ROUND = 32;
  if (per_query && !bf16) return fail(WDBX_E_STATE, "single-query passes need the bf16 tiles");
  if (per_query && (rc = grow((void**)&ix->d_qb16, &ix->qb16_bytes, (size_t)PQ_ROUND * 128 * kpad * 2))) return rc;
  for (int q0 = 0; q0 < nq;) {
    // query block: 256, 128 or 64 wide -- a small batch does not pay for 256 columns (the bf16 tiles: 256 or 128)
    const int rem = nq - q0;
    int ct = (ix->opt_gemm_ct == 1 || ix->opt_gemm_ct == 2 || ix->opt_gemm_ct == 4) ? (int)ix->opt_gemm_ct
             : rem > 128 ? 4 : rem > 64 ? 2 : 1;
    if (bf16 && ct < 2) ct = 2;
    if (per_query) ct = 2;
    const int gbn = 64 * ct, nv = std::min(per_query ? PQ_ROUND : gbn, rem);
    const int passes = per_query ? nv : 1;  // tile kernel launches per phase this round
    const float* qsrc = d_queries + (size_t)q0 * ix->pitch;
    if (nv < gbn && !bf16) {  // zero-padded private copy of a partial block (the bf16 block is padded by its conversion)
      HIP_TRY(hipMemsetAsync(ix->d_qblock, 0, (size_t)gbn * ix->pitch * sizeof(float), ix->stream));
      HIP_TRY(hipMemcpyAsync(ix->d_qblock, qsrc, (size_t)nv * ix->pitch * sizeof(float), hipMemcpyDeviceToDevice, ix->stream));
      qsrc = ix->d_qblock;
    }
    // tau = +inf for padded queries so they never append
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)ix->d_tau, 0x7F800000, GB_N, ix->stream));
    GemmArgs g = {};
    g.rows = (const f4*)ix->d_rows;
    g.queries = (const f4*)qsrc;
    g.n_rows = (uint32_t)ix->n;
    g.pitch4 = (uint32_t)pitch4;
    if (family == GEMM_BF16_SHADOW) {
      g.rows = (const f4*)ix->d_rows16;
      g.pitch4 = ix->pitch16 / 8;
    }
    g.num_tiles = sample_tiles;
    g.tile_stride = stride;
    g.cn = ix->d_cn;
    g.live = per_query ? 1 : 0;
    const size_t qb_block = (size_t)gbn * kpad;  // bf16 elements per query block
    if (bf16) {
      hipLaunchKernelGGL(queries_to_bf16_kernel, dim3((uint32_t)((passes * qb_block + 255) / 256)), dim3(256), 0, ix->stream, qsrc,
                         (uint32_t)ix->pitch, (uint32_t)nv, (__bf16*)ix->d_qb16, kpad, (uint32_t)gbn,
                         (uint32_t)(per_query ? 1 : gbn), (uint32_t)passes);
      HIP_TRY(hipGetLastError());
      g.qb_pitch16 = kpad / 8;
    }
    for (int i = 0; i < passes; ++i) {  // phase 0: maxima of the sampled tiles
      g.qb16 = bf16 ? (const void*)((const __bf16*)ix->d_qb16 + (size_t)i * qb_block) : nullptr;
      g.halfmax = ix->d_halfmax + (size_t)i * rw * sample_tiles;
      if ((rc = launch_gemm<0>(ix, g, ct, family))) return rc;
    }
    MergeArgs m = {};
    m.in = ix->d_halfmax;
    m.q_stride = (u64)rw * sample_tiles;
    m.i_stride = 0;
    m.p_stride = 1;
    m.P = rw * sample_tiles;
    m.list_len = 1;
    m.k = k;
    m.metric = ix->metric;
    m.out_kth = ix->d_tau;
    if ((rc = launch_merge(ix, m, nv))) return rc;
    if (inexact) {  // rounding-error margin below the sampled threshold: no true top-k row can be filtered out
      hipLaunchKernelGGL(tau_margin_kernel, dim3(nv), dim3(64), 0, ix->stream, ix->d_tau, qsrc, (uint32_t)ix->pitch, nv,
                         (const uint32_t*)ix->d_cnmax, ix->metric, (int)bf16);
      HIP_TRY(hipGetLastError());
    }
    g.num_tiles = tiles;
    g.tile_stride = 1;
    g.halfmax = nullptr;
    g.cap = cap;
    for (int i = 0; i < passes; ++i) {  // phase 1: every score above the threshold becomes a candidate
      g.qb16 = bf16 ? (const void*)((const __bf16*)ix->d_qb16 + (size_t)i * qb_block) : nullptr;
      g.tau = ix->d_tau + i;
      g.cand = ix->d_cand + (size_t)i * cap;
      g.count = d_count + q0 + i;
      if ((rc = launch_gemm<1>(ix, g, ct, family))) return rc;
    }
    if (inexact) {  // exact fp32 scores for the selected candidates
      hipLaunchKernelGGL(l2 ? rescore_kernel<WDBX_METRIC_L2> : rescore_kernel<WDBX_METRIC_COSINE>, dim3(64, nv), dim3(256), 0,
                         ix->stream, (const f4*)ix->d_rows, (uint32_t)pitch4, (const f4*)qsrc, ix->d_cand,
                         (const uint32_t*)(d_count + q0), cap);
      HIP_TRY(hipGetLastError());
    }
    MergeArgs f = {};
    f.in = ix->d_cand;
    f.q_stride = cap;
    f.i_stride = 0;
    f.p_stride = 1;
    f.P = cap;
    f.P_dev = d_count + q0;
    f.list_len = 1;
    f.k = k;
    f.metric = ix->metric;
    if (keys_out) {  // keys with global rows for the caller's own exchange
      f.row_base = (uint32_t)ix->row_base;
      f.out_keys = keys_out + (size_t)q0 * k;
    } else if (sharded) {  // this shard's lists with global rows, then the exchange
      f.row_base = (uint32_t)ix->row_base;
      f.out_keys = ix->d_local_keys;
    } else {
      f.out_idx = d_out_idx + (size_t)q0 * k;
      f.out_score = d_out_score + (size_t)q0 * k;
    }
    if ((rc = launch_merge(ix, f, nv))) return rc;
    if (sharded && (rc = exchange_and_merge(ix, nv, k, d_out_idx + (size_t)q0 * k, d_out_score + (size_t)q0 * k))) return rc;
    q0 += nv;
  }
  return WDBX_OK;
}

static int launch_normalize(wdbx_index* ix, float* d, uint64_t n) {
  if (!n) return WDBX_OK;
  const uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 3) / 4, 65536);
  hipLaunchKernelGGL(normalize_rows_kernel, dim3(blocks), dim3(256), 0, ix->stream, d, (u64)n, (uint32_t)ix->pitch);
  HIP_TRY(hipGetLastError());
  return WDBX_OK;
}

static int launch_fill(wdbx_index* ix, float* d, uint64_t seed, uint64_t row0, uint64_t n, int normalize) {
  if (!n) return WDBX_OK;
  const uint64_t total = n * (uint64_t)ix->pitch;
  const uint32_t blocks = (uint32_t)std::min<uint64_t>((total + 255) / 256, 65536);
  hipLaunchKernelGGL(fill_synthetic_kernel, dim3(blocks), dim3(256), 0, ix->stream, d, (u64)n, (uint32_t)ix->dim,
                     (uint32_t)ix->pitch, (u64)seed, (u64)row0);
  HIP_TRY(hipGetLastError());
  if (normalize) return launch_normalize(ix, d, n);
  return WDBX_OK;
}

static int reserve_locked(wdbx_index* ix, uint64_t cap) {
  if (cap <= ix->cap) return WDBX_OK;
  float* nd = nullptr;
  const size_t bytes = (size_t)cap * ix->pitch * sizeof(float);
  HIP_TRY(hipMalloc((void**)&nd, bytes));
  if (ix->n) {
    hipError_t e = hipMemcpyAsync(nd, ix->d_rows, (size_t)ix->n * ix->pitch * sizeof(float), hipMemcpyDeviceToDevice,
                                  ix->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
    if (e != hipSuccess) {
      (void)hipFree(nd);
      return fail(WDBX_E_HIP, "row copy during reserve failed: %s", hipGetErrorString(e));
    }
  }
  if (ix->d_rows) HIP_TRY(hipFree(ix->d_rows));
  ix->d_rows = nd;
  ix->cap = cap;
  return WDBX_OK;
}

static int upload_rows(wdbx_index* ix, uint64_t first, const float* rows, uint64_t n, int normalize) {
  ix->cn_rows = std::min<uint64_t>(ix->cn_rows, first);  // cached squared norms from `first` on are stale
  ix->shadow_rows = std::min<uint64_t>(ix->shadow_rows, first);  // and so are the bf16 and u8 shadows
  ix->shadow8_rows = std::min<uint64_t>(ix->shadow8_rows, first);
  float* dst = ix->d_rows + (size_t)first * ix->pitch;
  if (ix->pitch == ix->dim) {
    HIP_TRY(hipMemcpyAsync(dst, rows, (size_t)n * ix->dim * sizeof(float), hipMemcpyHostToDevice, ix->stream));
  } else {
    HIP_TRY(hipMemsetAsync(dst, 0, (size_t)n * ix->pitch * sizeof(float), ix->stream));
    HIP_TRY(hipMemcpy2DAsync(dst, (size_t)ix->pitch * sizeof(float), rows, (size_t)ix->dim * sizeof(float),
                             (size_t)ix->dim * sizeof(float), n, hipMemcpyHostToDevice, ix->stream));
  }
  if (normalize) {
    int rc = launch_normalize(ix, dst, n);
    if (rc) return rc;
  }
  HIP_TRY(hipStreamSynchronize(ix->stream));
  return WDBX_OK;
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
static int drain(EventPool& pool, uint64_t* count, double* ms);

extern "C" {

int wdbx_hip_version(void) { return WDBX_HIP_ABI_VERSION; }

const char* wdbx_last_error(void) { return g_err.c_str(); }

int wdbx_device_count(int* out_count) {
  if (!out_count) return fail(WDBX_E_INVALID, "out_count is null");
  *out_count = 0;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(WDBX_E_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  *out_count = n;
  return WDBX_OK;
}

int wdbx_index_create(int device_id, int dim, int metric, uint64_t capacity_rows, wdbx_index** out) {
  if (!out) return fail(WDBX_E_INVALID, "out is null");
  *out = nullptr;
  if (dim < 1 || dim > (1 << 20)) return fail(WDBX_E_INVALID, "dim=%d outside [1, 2^20]", dim);
  if (metric != WDBX_METRIC_COSINE && metric != WDBX_METRIC_L2) return fail(WDBX_E_INVALID, "metric=%d unknown", metric);
  int ndev = 0;
  int rc = wdbx_device_count(&ndev);
  if (rc) return rc;
  if (ndev < 1) return fail(WDBX_E_NODEVICE, "no HIP device visible: the WDBX HIP backend needs an AMD GPU");
  if (device_id < 0 || device_id >= ndev) return fail(WDBX_E_NODEVICE, "device_id=%d but %d device(s) visible", device_id, ndev);
  wdbx_index* ix = new (std::nothrow) wdbx_index();
  if (!ix) return fail(WDBX_E_NOMEM, "host allocation failed");
  ix->device = device_id;
  ix->dim = dim;
  ix->pitch = (dim + 3) / 4 * 4;
  ix->metric = metric;
  DeviceGuard g(device_id);
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device_id);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete ix;
    return fail(WDBX_E_HIP, "device setup failed: %s", hipGetErrorString(e));
  }
  ix->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  const char* env;
  if ((env = getenv("WDBX_HIP_SCAN_LANES"))) ix->opt_lanes = atoll(env);
  if ((env = getenv("WDBX_HIP_SCAN_BLOCKS"))) ix->opt_blocks = atoll(env);
  if ((env = getenv("WDBX_HIP_SCAN_NT"))) ix->opt_nt = atoll(env);
  if ((env = getenv("WDBX_HIP_SCAN_BLOCKED"))) ix->opt_blocked = atoll(env);
  rc = reserve_locked(ix, std::max<uint64_t>(capacity_rows, 1));
  if (rc) {
    (void)hipStreamDestroy(ix->stream);
    delete ix;
    return rc;
  }
  *out = ix;
  return WDBX_OK;
}

void wdbx_index_destroy(wdbx_index* ix) {
  if (!ix) return;
  {
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    (void)hipStreamSynchronize(ix->stream);
    if (ix->comm) (void)ncclCommDestroy(ix->comm);
    for (hipEvent_t e : ix->scan_ev.ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : ix->merge_ev.ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : ix->gemm_ev.ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : ix->sample_ev.ev) (void)hipEventDestroy(e);
    void* bufs[] = {ix->d_rows, ix->d_partials, ix->d_local_keys, ix->d_gathered, ix->d_q, ix->d_oidx, ix->d_oscore,
                    ix->d_qblock, ix->d_halfmax, ix->d_tau, ix->d_cand, ix->d_count, ix->d_mask, ix->d_dump, ix->d_sel, ix->d_state, ix->d_cn, ix->d_cnmax, ix->d_qb16, ix->d_rows16, ix->d_rows8, ix->d_scale8, ix->d_qinfo};
    for (void* p : bufs)
      if (p) (void)hipFree(p);
    if (ix->h_stage) (void)hipHostFree(ix->h_stage);
    (void)hipStreamDestroy(ix->stream);
  }
  delete ix;
}

int wdbx_index_dim(const wdbx_index* ix) { return ix ? ix->dim : 0; }
int wdbx_index_row_pitch(const wdbx_index* ix) { return ix ? ix->pitch : 0; }

int wdbx_index_size(wdbx_index* ix, uint64_t* out_rows) {
  if (!ix || !out_rows) return fail(WDBX_E_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(ix->mu);
  *out_rows = ix->n;
  return WDBX_OK;
}

int wdbx_index_capacity(wdbx_index* ix, uint64_t* out_rows) {
  if (!ix || !out_rows) return fail(WDBX_E_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(ix->mu);
  *out_rows = ix->cap;
  return WDBX_OK;
}

int wdbx_index_reserve(wdbx_index* ix, uint64_t capacity_rows) {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  return reserve_locked(ix, capacity_rows);
}

int wdbx_index_clear(wdbx_index* ix) {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  HIP_TRY(hipStreamSynchronize(ix->stream));
  ix->n = 0;
  ix->cn_rows = 0;
  ix->shadow_rows = 0;
  ix->shadow8_rows = 0;
  return WDBX_OK;
}

static int ensure_room(wdbx_index* ix, uint64_t extra) {
  const uint64_t need = ix->n + extra;
  if (need <= ix->cap) return WDBX_OK;
  return reserve_locked(ix, std::max(need, ix->cap + ix->cap / 2));
}

int wdbx_index_add(wdbx_index* ix, const float* rows, uint64_t n, int normalize, uint64_t* first_row_out) {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  if (n && !rows) return fail(WDBX_E_INVALID, "rows is null");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  if (first_row_out) *first_row_out = ix->n;
  if (!n) return WDBX_OK;
  int rc = ensure_room(ix, n);
  if (rc) return rc;
  rc = upload_rows(ix, ix->n, rows, n, normalize);
  if (rc) return rc;
  ix->n += n;
  return WDBX_OK;
}

int wdbx_index_set_rows(wdbx_index* ix, uint64_t first_row, const float* rows, uint64_t n, int normalize) {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  if (n && !rows) return fail(WDBX_E_INVALID, "rows is null");
  std::lock_guard<std::mutex> lk(ix->mu);
  if (first_row > ix->n || n > ix->n - first_row)
    return fail(WDBX_E_INVALID, "rows [%llu, +%llu) outside the %llu stored rows", (u64)first_row, (u64)n, (u64)ix->n);
  if (!n) return WDBX_OK;
  DeviceGuard g(ix->device);
  return upload_rows(ix, first_row, rows, n, normalize);
}

int wdbx_index_get_rows(wdbx_index* ix, uint64_t first_row, uint64_t n, float* out_rows) {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  if (n && !out_rows) return fail(WDBX_E_INVALID, "out_rows is null");
  std::lock_guard<std::mutex> lk(ix->mu);
  if (first_row > ix->n || n > ix->n - first_row)
    return fail(WDBX_E_INVALID, "rows [%llu, +%llu) outside the %llu stored rows", (u64)first_row, (u64)n, (u64)ix->n);
  if (!n) return WDBX_OK;
  DeviceGuard g(ix->device);
  HIP_TRY(hipStreamSynchronize(ix->stream));
  const float* src = ix->d_rows + (size_t)first_row * ix->pitch;
  if (ix->pitch == ix->dim)
    HIP_TRY(hipMemcpy(out_rows, src, (size_t)n * ix->dim * sizeof(float), hipMemcpyDeviceToHost));
  else
    HIP_TRY(hipMemcpy2D(out_rows, (size_t)ix->dim * sizeof(float), src, (size_t)ix->pitch * sizeof(float),
                        (size_t)ix->dim * sizeof(float), n, hipMemcpyDeviceToHost));
  return WDBX_OK;
}

int wdbx_index_fill_synthetic(wdbx_index* ix, uint64_t seed, uint64_t counter_row0, uint64_t n, int normalize,
                              uint64_t* first_row_out) {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  if (first_row_out) *first_row_out = ix->n;
  if (!n) return WDBX_OK;
  int rc = ensure_room(ix, n);
  if (rc) return rc;
  rc = launch_fill(ix, ix->d_rows + (size_t)ix->n * ix->pitch, seed, counter_row0, n, normalize);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(ix->stream));
  ix->n += n;
  return WDBX_OK;
}

static int search_host(wdbx_index* ix, const float* queries, int nq, int k, int normalize_queries,
                       const uint32_t* mask_words, int64_t* out_idx, float* out_score) {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  if (nq < 0) return fail(WDBX_E_INVALID, "nq=%d", nq);
  if (nq == 0) return WDBX_OK;
  if (!queries || !out_idx || !out_score) return fail(WDBX_E_INVALID, "null buffer");
  if (k < 1 || k > WDBX_MAX_K) return fail(WDBX_E_INVALID, "k=%d outside [1, %d]", k, WDBX_MAX_K);
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  struct MaskScope {  // the mask applies to this call only
    wdbx_index* ix;
    ~MaskScope() { ix->active_mask = nullptr; }
  } scope{ix};
  int rc;
  if (mask_words && ix->n) {
    const size_t words = (size_t)((ix->n + 31) / 32);
    rc = grow((void**)&ix->d_mask, &ix->mask_bytes, words * sizeof(uint32_t));
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(ix->d_mask, mask_words, words * sizeof(uint32_t), hipMemcpyHostToDevice, ix->stream));
    ix->active_mask = ix->d_mask;
  }
  const size_t elems = (size_t)nq * k;
  const size_t q_bytes = (size_t)nq * ix->pitch * sizeof(float);
  constexpr size_t STAGE_Q = 256 << 10, STAGE_IDX = 256 << 10, STAGE_SCORE = 128 << 10;
  const bool gemm = !ix->active_mask && gemm_eligible(ix, nq, k);
  bool zero_copy = ix->opt_zero_copy && !gemm && q_bytes <= STAGE_Q && elems * sizeof(int64_t) <= STAGE_IDX;
  if (zero_copy && !ix->h_stage) {
    void* hp = nullptr;
    void* dp = nullptr;
    if (hipHostMalloc(&hp, STAGE_Q + STAGE_IDX + STAGE_SCORE, hipHostMallocMapped) == hipSuccess &&
        hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
      ix->h_stage = (char*)hp;
      ix->h_stage_dev = (char*)dp;
    } else {
      if (hp) (void)hipHostFree(hp);
      (void)hipGetLastError();
      ix->opt_zero_copy = 0;  // not available here: use the copy path from now on
      zero_copy = false;
    }
  }
  float* dq;
  int64_t* doidx;
  float* doscore;
  if (zero_copy) {
    float* hq = (float*)ix->h_stage;
    if (ix->pitch == ix->dim) {
      memcpy(hq, queries, q_bytes);
    } else {
      memset(hq, 0, q_bytes);
      for (int q = 0; q < nq; ++q) memcpy(hq + (size_t)q * ix->pitch, queries + (size_t)q * ix->dim, (size_t)ix->dim * sizeof(float));
    }
    dq = (float*)ix->h_stage_dev;
    doidx = (int64_t*)(ix->h_stage_dev + STAGE_Q);
    doscore = (float*)(ix->h_stage_dev + STAGE_Q + STAGE_IDX);
  } else {
    rc = grow((void**)&ix->d_q, &ix->q_bytes, q_bytes);
    if (rc) return rc;
    if (elems > ix->out_elems) {
      if (ix->d_oidx) HIP_TRY(hipFree(ix->d_oidx));
      if (ix->d_oscore) HIP_TRY(hipFree(ix->d_oscore));
      ix->d_oidx = nullptr;
      ix->d_oscore = nullptr;
      ix->out_elems = 0;
      HIP_TRY(hipMalloc((void**)&ix->d_oidx, elems * sizeof(int64_t)));
      HIP_TRY(hipMalloc((void**)&ix->d_oscore, elems * sizeof(float)));
      ix->out_elems = elems;
    }
    if (ix->pitch == ix->dim) {
      HIP_TRY(hipMemcpyAsync(ix->d_q, queries, q_bytes, hipMemcpyHostToDevice, ix->stream));
    } else {
      HIP_TRY(hipMemsetAsync(ix->d_q, 0, q_bytes, ix->stream));
      HIP_TRY(hipMemcpy2DAsync(ix->d_q, (size_t)ix->pitch * sizeof(float), queries, (size_t)ix->dim * sizeof(float),
                               (size_t)ix->dim * sizeof(float), nq, hipMemcpyHostToDevice, ix->stream));
    }
    dq = ix->d_q;
    doidx = ix->d_oidx;
    doscore = ix->d_oscore;
  }
  if (normalize_queries && ix->metric == WDBX_METRIC_COSINE) {
    rc = launch_normalize(ix, dq, nq);
    if (rc) return rc;
  }
  if (gemm) {
    rc = enqueue_search_gemm(ix, dq, nq, k, doidx, doscore);
    if (rc) return rc;
    std::vector<uint32_t> counts(nq);
    HIP_TRY(hipMemcpyAsync(counts.data(), ix->d_count, (size_t)nq * sizeof(uint32_t), hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    for (int q = 0; q < nq; ++q)  // a query whose candidate buffer overflowed is re-run exactly on the scan path
      if (counts[q] > ix->last_batch_cap) {
        const int64_t keep = ix->opt_scan_shadow;  // straight to the fp32 scan: the selection would overflow again
        ix->opt_scan_shadow = 0;
        rc = enqueue_search(ix, dq + (size_t)q * ix->pitch, 1, k, doidx + (size_t)q * k, doscore + (size_t)q * k, SEARCH_FINAL);
        ix->opt_scan_shadow = keep;
        if (rc) return rc;
      }
  } else {
    rc = enqueue_search(ix, dq, nq, k, doidx, doscore, SEARCH_FINAL);
    if (rc) return rc;
  }
  if (zero_copy) {
    HIP_TRY(hipStreamSynchronize(ix->stream));
    memcpy(out_idx, ix->h_stage + STAGE_Q, elems * sizeof(int64_t));
    memcpy(out_score, ix->h_stage + STAGE_Q + STAGE_IDX, elems * sizeof(float));
  } else {
    HIP_TRY(hipMemcpyAsync(out_idx, doidx, elems * sizeof(int64_t), hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(hipMemcpyAsync(out_score, doscore, elems * sizeof(float), hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(hipStreamSynchronize(ix->stream));
  }
  return WDBX_OK;
}

int wdbx_index_search(wdbx_index* ix, const float* queries, int nq, int k, int normalize_queries, int64_t* out_idx,
                      float* out_score) {
  return search_host(ix, queries, nq, k, normalize_queries, nullptr, out_idx, out_score);
}

int wdbx_index_search_masked(wdbx_index* ix, const float* queries, int nq, int k, int normalize_queries,
                             const uint32_t* mask_words, int64_t* out_idx, float* out_score) {
  if (!mask_words) return fail(WDBX_E_INVALID, "mask_words is null");
  return search_host(ix, queries, nq, k, normalize_queries, mask_words, out_idx, out_score);
}

int wdbx_device_alloc(wdbx_index* ix, uint64_t bytes, void** out_dev_ptr) {
  if (!ix || !out_dev_ptr) return fail(WDBX_E_INVALID, "null argument");
  *out_dev_ptr = nullptr;
  DeviceGuard g(ix->device);
  HIP_TRY(hipMalloc(out_dev_ptr, bytes ? bytes : 1));
  return WDBX_OK;
}

int wdbx_device_free(wdbx_index* ix, void* dev_ptr) {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  if (!dev_ptr) return WDBX_OK;
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  HIP_TRY(hipStreamSynchronize(ix->stream));
  HIP_TRY(hipFree(dev_ptr));
  return WDBX_OK;
}

int wdbx_device_upload(wdbx_index* ix, void* dev_dst, const void* host_src, uint64_t bytes) {
  if (!ix || (bytes && (!dev_dst || !host_src))) return fail(WDBX_E_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  HIP_TRY(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ix->stream));
  HIP_TRY(hipStreamSynchronize(ix->stream));
  return WDBX_OK;
}

int wdbx_device_download(wdbx_index* ix, void* host_dst, const void* dev_src, uint64_t bytes) {
  if (!ix || (bytes && (!host_dst || !dev_src))) return fail(WDBX_E_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  HIP_TRY(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ix->stream));
  HIP_TRY(hipStreamSynchronize(ix->stream));
  return WDBX_OK;
}

int wdbx_device_fill_synthetic(wdbx_index* ix, float* dev_dst, uint64_t seed, uint64_t counter_row0, uint64_t n,
                               int normalize) {
  if (!ix || (n && !dev_dst)) return fail(WDBX_E_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  int rc = launch_fill(ix, dev_dst, seed, counter_row0, n, normalize);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(ix->stream));
  return WDBX_OK;
}

int wdbx_index_search_device(wdbx_index* ix, const float* d_queries, int nq, int k, int64_t* d_out_idx,
                             float* d_out_score) {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  return enqueue_search(ix, d_queries, nq, k, d_out_idx, d_out_score, SEARCH_FINAL);
}

int wdbx_index_search_sharded_device(wdbx_index* ix, const float* d_queries, int nq, int k, int64_t* d_out_idx,
                                     float* d_out_score) {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  return enqueue_search(ix, d_queries, nq, k, d_out_idx, d_out_score, SEARCH_SHARDED);
}

int wdbx_index_search_sharded_batch_device(wdbx_index* ix, const float* d_queries, int nq, int k, int64_t* d_out_idx,
                                           float* d_out_score) {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  if (!gemm_eligible(ix, std::max(nq, (int)ix->opt_gemm_min_nq), k))
    return fail(WDBX_E_STATE, "batched MFMA path needs >= %lld rows and k*1024 <= rows on every rank",
                (long long)ix->opt_gemm_min_rows);
  return enqueue_search_gemm(ix, d_queries, nq, k, d_out_idx, d_out_score, SEARCH_SHARDED);
}

int wdbx_index_search_batch_device(wdbx_index* ix, const float* d_queries, int nq, int k, int64_t* d_out_idx,
                                   float* d_out_score) {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  if (!gemm_eligible(ix, std::max(nq, (int)ix->opt_gemm_min_nq), k))
    return fail(WDBX_E_STATE, "batched MFMA path needs >= %lld rows and k*1024 <= rows", (long long)ix->opt_gemm_min_rows);
  return enqueue_search_gemm(ix, d_queries, nq, k, d_out_idx, d_out_score);
}

int wdbx_index_batch_status(wdbx_index* ix, uint32_t* out_counts, int nq, uint32_t* out_capacity, int* out_overflowed) {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  if (nq < 0 || (uint32_t)nq > ix->last_batch_nq) return fail(WDBX_E_INVALID, "nq=%d but the last batch had %u queries", nq, ix->last_batch_nq);
  std::vector<uint32_t> counts((size_t)std::max(nq, 1));
  if (nq) HIP_TRY(hipMemcpyAsync(counts.data(), ix->d_count, (size_t)nq * sizeof(uint32_t), hipMemcpyDeviceToHost, ix->stream));
  HIP_TRY(hipStreamSynchronize(ix->stream));
  int over = 0;
  for (int q = 0; q < nq; ++q) {
    if (counts[q] > ix->last_batch_cap) ++over;
    if (out_counts) out_counts[q] = counts[q];
  }
  if (out_capacity) *out_capacity = ix->last_batch_cap;
  if (out_overflowed) *out_overflowed = over;
  return WDBX_OK;
}

int wdbx_index_profile_read_gemm(wdbx_index* ix, uint64_t* launches, double* ms_total) {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  HIP_TRY(hipStreamSynchronize(ix->stream));
  return drain(ix->gemm_ev, launches, ms_total);
}

int wdbx_index_profile_read_sample(wdbx_index* ix, uint64_t* launches, double* ms_total) {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  HIP_TRY(hipStreamSynchronize(ix->stream));
  return drain(ix->sample_ev, launches, ms_total);
}

int wdbx_index_synchronize(wdbx_index* ix) {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  HIP_TRY(hipStreamSynchronize(ix->stream));
  return WDBX_OK;
}

int wdbx_comm_unique_id(void* out_128_bytes) {
  if (!out_128_bytes) return fail(WDBX_E_INVALID, "null argument");
  static_assert(sizeof(ncclUniqueId) <= WDBX_UNIQUE_ID_BYTES, "unique id larger than the ABI slot");
  ncclUniqueId id;
  NCCL_TRY(ncclGetUniqueId(&id));
  memset(out_128_bytes, 0, WDBX_UNIQUE_ID_BYTES);
  memcpy(out_128_bytes, &id, sizeof id);
  return WDBX_OK;
}

int wdbx_index_comm_init(wdbx_index* ix, int nranks, int rank, const void* unique_id_128_bytes,
                         uint64_t global_row_base) {
  if (!ix || !unique_id_128_bytes) return fail(WDBX_E_INVALID, "null argument");
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(WDBX_E_INVALID, "rank %d of %d", rank, nranks);
  if (global_row_base >= 0xFFFFFFFFull) return fail(WDBX_E_INVALID, "global row base exceeds 32-bit row keys");
  std::lock_guard<std::mutex> lk(ix->mu);
  if (ix->comm) return fail(WDBX_E_STATE, "communicator already initialised");
  DeviceGuard g(ix->device);
  ncclUniqueId id;
  memcpy(&id, unique_id_128_bytes, sizeof id);
  NCCL_TRY(ncclCommInitRank(&ix->comm, nranks, id, rank));
  ix->nranks = nranks;
  ix->rank = rank;
  ix->row_base = global_row_base;
  return WDBX_OK;
}

int wdbx_index_comm_destroy(wdbx_index* ix) {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  if (!ix->comm) return WDBX_OK;
  DeviceGuard g(ix->device);
  HIP_TRY(hipStreamSynchronize(ix->stream));
  NCCL_TRY(ncclCommDestroy(ix->comm));
  ix->comm = nullptr;
  ix->nranks = 1;
  ix->rank = 0;
  ix->row_base = 0;
  return WDBX_OK;
}

int wdbx_index_probe_read(wdbx_index* ix, int nontemporal, int blocks, int reps, double* out_ms_per_pass) {
  if (!ix || !out_ms_per_pass) return fail(WDBX_E_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  if (!ix->n) return fail(WDBX_E_STATE, "empty index");
  int rc = grow((void**)&ix->d_tau, &ix->tau_bytes, (size_t)GB_N * sizeof(float));
  if (rc) return rc;
  const u64 quads = (u64)ix->n * (u64)(ix->pitch / 4);
  const uint32_t grid = blocks > 0 ? (uint32_t)blocks : (uint32_t)ix->cu_count * 8;
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  reps = std::max(1, reps);
  for (int i = -2; i < reps; ++i) {
    if (i == 0) HIP_TRY(hipEventRecord(e0, ix->stream));
    if (nontemporal)
      hipLaunchKernelGGL(probe_read_kernel<true>, dim3(grid), dim3(256), 0, ix->stream, (const f4*)ix->d_rows, quads, ix->d_tau);
    else
      hipLaunchKernelGGL(probe_read_kernel<false>, dim3(grid), dim3(256), 0, ix->stream, (const f4*)ix->d_rows, quads, ix->d_tau);
  }
  HIP_TRY(hipEventRecord(e1, ix->stream));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *out_ms_per_pass = ms / reps;
  return WDBX_OK;
}


// ------------------------------------------------------------------------------------------------
// in-process shard group: S shards on S devices driven by one process (the reference's
// VectorStore(num_shards=S) shape, vector_store.py:111-134, :323-345), RCCL communicators from
// ncclCommInitAll, contiguous row ranges (row r lives in shard r / cap_per_shard)
// ------------------------------------------------------------------------------------------------
struct wdbx_group {
  std::vector<wdbx_index*> shard;
  std::vector<ncclComm_t> comm;
  uint64_t cap_per_shard = 0;
  int dim = 0, metric = 0;
  std::mutex mu;
};

int wdbx_group_create(const int* device_ids, int n, int dim, int metric, uint64_t cap_per_shard, wdbx_group** out) {
  if (!out) return fail(WDBX_E_INVALID, "out is null");
  *out = nullptr;
  if (!device_ids || n < 1 || n > 64) return fail(WDBX_E_INVALID, "need 1..64 device ids");
  if (cap_per_shard < 1 || cap_per_shard * (uint64_t)n >= 0xFFFFFF00ull)
    return fail(WDBX_E_INVALID, "cap_per_shard * shards must stay below 2^32 rows");
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < i; ++j)
      if (device_ids[i] == device_ids[j]) return fail(WDBX_E_INVALID, "device %d listed twice (RCCL needs one rank per device)", device_ids[i]);
  wdbx_group* g = new (std::nothrow) wdbx_group();
  if (!g) return fail(WDBX_E_NOMEM, "host allocation failed");
  g->cap_per_shard = cap_per_shard;
  g->dim = dim;
  g->metric = metric;
  int rc = WDBX_OK;
  for (int i = 0; i < n && rc == WDBX_OK; ++i) {
    wdbx_index* ix = nullptr;
    rc = wdbx_index_create(device_ids[i], dim, metric, cap_per_shard, &ix);
    if (rc == WDBX_OK) {
      ix->row_base = (uint64_t)i * cap_per_shard;
      g->shard.push_back(ix);
    }
  }
  if (rc == WDBX_OK) {
    g->comm.resize(n);
    ncclResult_t r = ncclCommInitAll(g->comm.data(), n, device_ids);
    if (r != ncclSuccess) {
      g->comm.clear();
      rc = fail(WDBX_E_RCCL, "ncclCommInitAll failed: %s", ncclGetErrorString(r));
    }
  }
  if (rc != WDBX_OK) {
    const std::string keep = g_err;
    for (wdbx_index* ix : g->shard) wdbx_index_destroy(ix);
    delete g;
    g_err = keep;
    return rc;
  }
  *out = g;
  return WDBX_OK;
}

void wdbx_group_destroy(wdbx_group* g) {
  if (!g) return;
  for (size_t i = 0; i < g->shard.size(); ++i) {
    DeviceGuard dg(g->shard[i]->device);
    (void)hipStreamSynchronize(g->shard[i]->stream);
    if (i < g->comm.size() && g->comm[i]) (void)ncclCommDestroy(g->comm[i]);
  }
  for (wdbx_index* ix : g->shard) wdbx_index_destroy(ix);
  delete g;
}

int wdbx_group_size(wdbx_group* g, uint64_t* out_rows) {
  if (!g || !out_rows) return fail(WDBX_E_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(g->mu);
  uint64_t total = 0;
  for (wdbx_index* ix : g->shard) total += ix->n;
  *out_rows = total;
  return WDBX_OK;
}

// append rows; they fill shard 0 up to cap_per_shard, then shard 1, ... (contiguous global rows)
int wdbx_group_add(wdbx_group* g, const float* rows, uint64_t n, int normalize, uint64_t* first_row_out) {
  if (!g) return fail(WDBX_E_INVALID, "null handle");
  if (n && !rows) return fail(WDBX_E_INVALID, "rows is null");
  std::lock_guard<std::mutex> lk(g->mu);
  uint64_t total = 0;
  for (wdbx_index* ix : g->shard) total += ix->n;
  if (total + n > g->cap_per_shard * g->shard.size())
    return fail(WDBX_E_INVALID, "group is full: %llu + %llu rows > %llu", (u64)total, (u64)n, (u64)(g->cap_per_shard * g->shard.size()));
  if (first_row_out) *first_row_out = total;
  uint64_t done = 0;
  while (done < n) {
    const size_t s = (size_t)((total + done) / g->cap_per_shard);
    wdbx_index* ix = g->shard[s];
    const uint64_t room = g->cap_per_shard - ix->n, take = std::min(room, n - done);
    int rc = wdbx_index_add(ix, rows + (size_t)done * g->dim, take, normalize, nullptr);
    if (rc) return rc;
    done += take;
  }
  return WDBX_OK;
}

// blocking search over all shards: every shard scans its rows, the per-shard key lists are all-gathered
// (one ncclAllGather per shard inside a group call) and merged on shard 0's device
int wdbx_group_search(wdbx_group* g, const float* queries, int nq, int k, int normalize_queries, int64_t* out_idx,
                      float* out_score) {
  if (!g) return fail(WDBX_E_INVALID, "null handle");
  if (nq < 0) return fail(WDBX_E_INVALID, "nq=%d", nq);
  if (nq == 0) return WDBX_OK;
  if (!queries || !out_idx || !out_score) return fail(WDBX_E_INVALID, "null buffer");
  if (k < 1 || k > WDBX_MAX_K) return fail(WDBX_E_INVALID, "k=%d outside [1, %d]", k, WDBX_MAX_K);
  std::lock_guard<std::mutex> lk(g->mu);
  const int S = (int)g->shard.size();
  const int batch = 32;
  int rc;
  for (int s = 0; s < S; ++s) {  // queries to every device
    wdbx_index* ix = g->shard[s];
    std::lock_guard<std::mutex> li(ix->mu);
    DeviceGuard dg(ix->device);
    if ((rc = grow((void**)&ix->d_q, &ix->q_bytes, (size_t)nq * ix->pitch * sizeof(float)))) return rc;
    if (ix->pitch == ix->dim) {
      HIP_TRY(hipMemcpyAsync(ix->d_q, queries, (size_t)nq * ix->dim * sizeof(float), hipMemcpyHostToDevice, ix->stream));
    } else {
      HIP_TRY(hipMemsetAsync(ix->d_q, 0, (size_t)nq * ix->pitch * sizeof(float), ix->stream));
      HIP_TRY(hipMemcpy2DAsync(ix->d_q, (size_t)ix->pitch * sizeof(float), queries, (size_t)ix->dim * sizeof(float),
                               (size_t)ix->dim * sizeof(float), nq, hipMemcpyHostToDevice, ix->stream));
    }
    if (normalize_queries && ix->metric == WDBX_METRIC_COSINE && (rc = launch_normalize(ix, ix->d_q, nq))) return rc;
    if ((rc = grow((void**)&ix->d_gathered, &ix->gathered_bytes, (size_t)S * batch * k * sizeof(u64)))) return rc;
  }
  wdbx_index* root = g->shard[0];
  {
    std::lock_guard<std::mutex> li(root->mu);
    DeviceGuard dg(root->device);
    const size_t elems = (size_t)nq * k;
    if (elems > root->out_elems) {
      if (root->d_oidx) HIP_TRY(hipFree(root->d_oidx));
      if (root->d_oscore) HIP_TRY(hipFree(root->d_oscore));
      root->d_oidx = nullptr;
      root->d_oscore = nullptr;
      root->out_elems = 0;
      HIP_TRY(hipMalloc((void**)&root->d_oidx, elems * sizeof(int64_t)));
      HIP_TRY(hipMalloc((void**)&root->d_oscore, elems * sizeof(float)));
      root->out_elems = elems;
    }
  }
  for (int q0 = 0; q0 < nq; q0 += batch) {
    const int b = std::min(batch, nq - q0);
    for (int s = 0; s < S; ++s) {  // local stage on every device
      wdbx_index* ix = g->shard[s];
      std::lock_guard<std::mutex> li(ix->mu);
      DeviceGuard dg(ix->device);
      if ((rc = enqueue_search(ix, ix->d_q + (size_t)q0 * ix->pitch, b, k, nullptr, nullptr, SEARCH_LOCAL_KEYS))) return rc;
    }
    NCCL_TRY(ncclGroupStart());
    for (int s = 0; s < S; ++s) {
      wdbx_index* ix = g->shard[s];
      ncclResult_t r = ncclAllGather(ix->d_local_keys, ix->d_gathered, (size_t)b * k, ncclUint64, g->comm[s], ix->stream);
      if (r != ncclSuccess) {
        (void)ncclGroupEnd();
        return fail(WDBX_E_RCCL, "ncclAllGather failed: %s", ncclGetErrorString(r));
      }
    }
    NCCL_TRY(ncclGroupEnd());
    {
      std::lock_guard<std::mutex> li(root->mu);
      DeviceGuard dg(root->device);
      MergeArgs m = {};
      m.list_len = k;
      m.in = root->d_gathered;
      m.q_stride = (uint64_t)k;
      m.i_stride = 1;
      m.p_stride = (uint64_t)b * k;
      m.P = (uint32_t)S;
      m.k = k;
      m.metric = root->metric;
      m.out_idx = root->d_oidx + (size_t)q0 * k;
      m.out_score = root->d_oscore + (size_t)q0 * k;
      if ((rc = launch_merge(root, m, b))) return rc;
    }
  }
  {
    std::lock_guard<std::mutex> li(root->mu);
    DeviceGuard dg(root->device);
    const size_t elems = (size_t)nq * k;
    HIP_TRY(hipMemcpyAsync(out_idx, root->d_oidx, elems * sizeof(int64_t), hipMemcpyDeviceToHost, root->stream));
    HIP_TRY(hipMemcpyAsync(out_score, root->d_oscore, elems * sizeof(float), hipMemcpyDeviceToHost, root->stream));
    HIP_TRY(hipStreamSynchronize(root->stream));
  }
  for (int s = 1; s < S; ++s) {  // the other shards' streams only ran their local stage + all-gather
    DeviceGuard dg(g->shard[s]->device);
    HIP_TRY(hipStreamSynchronize(g->shard[s]->stream));
  }
  return WDBX_OK;
}

int wdbx_index_profile(wdbx_index* ix, int enable) {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  ix->profile = enable != 0;
  return WDBX_OK;
}

static int drain(EventPool& pool, uint64_t* count, double* ms) {
  double total = 0;
  for (size_t i = 0; i + 1 < pool.used; i += 2) {
    float t = 0;
    HIP_TRY(hipEventElapsedTime(&t, pool.ev[i], pool.ev[i + 1]));
    total += t;
  }
  if (count) *count = pool.used / 2;
  if (ms) *ms = total;
  pool.used = 0;
  return WDBX_OK;
}

int wdbx_index_profile_read(wdbx_index* ix, uint64_t* scan_launches, double* scan_ms_total, uint64_t* merge_launches,
                            double* merge_ms_total) {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  HIP_TRY(hipStreamSynchronize(ix->stream));
  int rc = drain(ix->scan_ev, scan_launches, scan_ms_total);
  if (rc) return rc;
  return drain(ix->merge_ev, merge_launches, merge_ms_total);
}

static int64_t* option_slot(wdbx_index* ix, const char* name) {
  if (!name) return nullptr;
  if (!strcmp(name, "scan_lanes")) return &ix->opt_lanes;
  if (!strcmp(name, "scan_blocks")) return &ix->opt_blocks;
  if (!strcmp(name, "scan_nt")) return &ix->opt_nt;
  if (!strcmp(name, "scan_blocked")) return &ix->opt_blocked;
  if (!strcmp(name, "scan_generic")) return &ix->opt_generic;
  if (!strcmp(name, "exchange_batch")) return &ix->opt_batch;
  if (!strcmp(name, "lds_lists")) return &ix->opt_lds_lists;
  if (!strcmp(name, "zero_copy")) return &ix->opt_zero_copy;
  if (!strcmp(name, "wg_merge")) return &ix->opt_wg_merge;
  if (!strcmp(name, "gemm_ct")) return &ix->opt_gemm_ct;
  if (!strcmp(name, "gemm_l2")) return &ix->opt_gemm_l2;
  if (!strcmp(name, "gemm_bf16")) return &ix->opt_gemm_bf16;
  if (!strcmp(name, "scan_shadow")) return &ix->opt_scan_shadow;
  if (!strcmp(name, "scan8_wgs")) return &ix->opt_scan8_wgs;
  if (!strcmp(name, "scan_force_ragged")) return &ix->opt_force_ragged;
  if (!strcmp(name, "select_min_k")) return &ix->opt_select_min_k;
  if (!strcmp(name, "gemm_min_queries")) return &ix->opt_gemm_min_nq;
  if (!strcmp(name, "gemm_min_rows")) return &ix->opt_gemm_min_rows;
  if (!strcmp(name, "gemm_sample_div")) return &ix->opt_gemm_sample_div;
  return nullptr;
}

int wdbx_index_set_option(wdbx_index* ix, const char* name, int64_t value) {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  int64_t* slot = option_slot(ix, name);
  if (!slot) return fail(WDBX_E_INVALID, "unknown option '%s'", name ? name : "(null)");
  *slot = value;
  return WDBX_OK;
}

int wdbx_index_get_option(wdbx_index* ix, const char* name, int64_t* value) {
  if (!ix || !value) return fail(WDBX_E_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(ix->mu);
  // read-only state of the batched path
  if (name && !strcmp(name, "last_gemm_family")) return *value = ix->last_gemm_mode, WDBX_OK;
  if (name && !strcmp(name, "shadow_rows")) return *value = (int64_t)ix->shadow_rows, WDBX_OK;
  if (name && !strcmp(name, "shadow_bytes")) return *value = (int64_t)ix->rows16_bytes, WDBX_OK;
  if (name && !strcmp(name, "shadow8_rows")) return *value = (int64_t)ix->shadow8_rows, WDBX_OK;
  if (name && !strcmp(name, "shadow8_bytes")) return *value = (int64_t)(ix->rows8_bytes + ix->scale8_bytes), WDBX_OK;
  if (name && !strcmp(name, "last_single_path")) return *value = ix->last_single_path, WDBX_OK;
  int64_t* slot = option_slot(ix, name);
  if (!slot) return fail(WDBX_E_INVALID, "unknown option '%s'", name ? name : "(null)");
  *value = *slot;
  return WDBX_OK;
}

}  // extern "C"
