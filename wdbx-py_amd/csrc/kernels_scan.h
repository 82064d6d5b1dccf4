// kernels_scan.h -- the fp32 scan kernels (one query, HBM-bound): unrolled instances and the generic form.
// Part of the single translation unit wdbx_hip.hip (included there, in order); not a standalone header.

// ------------------------------------------------------------------------------------------------
// scan kernel, specialised: L lanes per row, QPL quads (16 B) per lane per row, fully unrolled
// ------------------------------------------------------------------------------------------------
// MODE 0: per-wave top-k list in LDS, 1: in registers (k <= 128), 2: no list at all -- every row's key is
// written to a.partials[row] and the top-k is taken by the radix select below (large k)
// RAGGED: L*QPL > pitch4 -- the lane slots past the row end load the row's last quad again (always a
// valid address, the same cache line as a neighbour) and contribute zero, so ANY dimension up to
// 3072 floats runs on an unrolled instance (d = 100, 200, 300, 1000 ...) instead of the generic kernel
// Register budget: 4 waves per SIMD (128 VGPRs) lets the allocator keep a pass's 8-12 loads in flight instead of
// serialising them; the RAGGED forms of the widest instances (QPL >= 8: clamped offsets + zeroing of the idle slots
// on top of 12 quads of query and 12 of row) need more than that and take the 3-wave budget (168 VGPRs) rather
// than spill -- the grid keeps 2 waves per SIMD resident either way.
template <int L, int QPL, int METRIC, bool NT, int MODE, bool RAGGED>
__device__ __forceinline__ void scan_body(const ScanArgs& a) {
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

// Which query a workgroup of a plain launch scans: row y of the grid = query y of the launch (one query, or the repair launches
// of a round of single queries in ONE grid, each returning at once unless its query overflowed: only_if_over).
__device__ __forceinline__ bool scan_pick_query(ScanArgs& a) {
  if (blockIdx.y) {
    a.query += (size_t)blockIdx.y * a.pitch4;
    a.partials += (size_t)blockIdx.y * a.y_partials;
    if (a.only_if_over) a.only_if_over += blockIdx.y;
  }
  return !(a.only_if_over && *a.only_if_over <= a.over_cap);  // repair launch, nothing to repair (uniform)
}

template <int L, int QPL, int METRIC, bool NT, int MODE, bool RAGGED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((RAGGED && QPL >= 8) ? 3 : 4, 4))) void scan_kernel(ScanArgs a) {
  if (!scan_pick_query(a)) return;
  scan_body<L, QPL, METRIC, NT, MODE, RAGGED>(a);
}

// ------------------------------------------------------------------------------------------------
// scan kernel, generic: any pitch; L = min(8, pow2ceil(pitch4)) lanes per row chosen at launch,
// query staged in LDS, runtime loop with a predicated tail
// ------------------------------------------------------------------------------------------------
template <int L, int METRIC, int MODE>
__device__ __forceinline__ void scan_body_generic(const ScanArgs& a) {
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

template <int L, int METRIC, int MODE>
__global__ __launch_bounds__(256) void scan_kernel_generic(ScanArgs a) {
  if (!scan_pick_query(a)) return;
  scan_body_generic<L, METRIC, MODE>(a);
}

// Repairs behind a BATCH (up to 256 queries): a grid with one row per query is 130 k workgroups that start only to return --
// 32 us per batch of 256.  Here the grid has a few rows, over_list = {n, q_0 .. q_(n-1)} names the overflowed queries
// (mark_lost_kernel) and row y takes q_y, q_(y + rows), ...: nothing listed, nothing done, 4 us.  The generic body (any
// pitch, query staged in LDS): a loop around an unrolled instance made the widest ones spill (their 128-register budget has
// no room for what the compiler hoists out of the loop), and a repair is the rare path.
template <int L, int METRIC, int MODE>
__global__ __launch_bounds__(256) void scan_kernel_listed(ScanArgs a) {
  const uint32_t n = a.over_list[0];
  for (uint32_t y = blockIdx.y; y < n; y += gridDim.y) {
    const uint32_t qi = a.over_list[1 + y];
    ScanArgs b = a;
    b.query += (size_t)qi * a.pitch4;
    b.partials += (size_t)qi * a.y_partials;
    scan_body_generic<L, METRIC, MODE>(b);
    __syncthreads();  // (the next query's lists and staged query reuse the LDS)
  }
}
