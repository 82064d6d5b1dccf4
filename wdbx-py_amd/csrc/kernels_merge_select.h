// kernels_merge_select.h -- merge of partial top-k lists; exact radix select for large k.
// Part of the single translation unit wdbx_hip.hip (included there, in order); not a standalone header.

// ------------------------------------------------------------------------------------------------
// A threshold from up to R * blockDim.x ordered 32-bit values, R per thread in REGISTERS (0 = absent), by the whole
// workgroup: the largest prefix P -- bit by bit from the first bit in which the values differ down to bit KTH_LOW_BIT --
// with at least k values >= P.  So P <= the k-th largest value, short of it by less than 2^KTH_LOW_BIT (2^-15 relative for
// float keys): as a selection threshold it is as valid as the exact k-th largest and keeps a few ppm more rows.  One round =
// R compare + ballot per wave, one LDS atomic per wave, one barrier, one broadcast read: 0.2 us, whatever k is -- where the
// list kernels insert candidate after candidate into a sorted list (25 us for the 10 k sampled lower bounds of a 10 M-row
// shard, 18 rounds here).  Returns 0 when fewer than k values are present.
// s: KTH_SCRATCH words of LDS, ZEROED by the caller (barrier included).  Every thread must call it.
// (Measured on the way: values in LDS, one read per value and round each waited for: 1 us per round; per-wave counts summed
// by every thread with four 16-byte broadcast reads: 0.7 us per round, LDS-bandwidth bound.)
// ------------------------------------------------------------------------------------------------
constexpr int KTH_LOW_BIT = 8, KTH_SCRATCH = 40;
template <int R>
__device__ __forceinline__ uint32_t block_kth_threshold(const uint32_t (&v)[R], uint32_t k, uint32_t* s) {
  const int lane = threadIdx.x & 63;
  uint32_t mx = 0, mn_inv = 0, cnt = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    mx = max(mx, v[r]);
    mn_inv = max(mn_inv, v[r] ? ~v[r] : 0u);
    cnt += (uint32_t)__builtin_popcountll(__ballot(v[r] != 0u));
  }
  for (int o = 32; o > 0; o >>= 1) {
    mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
    mn_inv = max(mn_inv, (uint32_t)__shfl_xor((int)mn_inv, o));
  }
  if (lane == 0) {
    atomicMax(&s[0], mx);
    atomicMax(&s[1], mn_inv);
    atomicAdd(&s[2], cnt);
  }
  __syncthreads();
  if (s[2] < k) return 0u;
  mx = s[0];
  const uint32_t diff = mx ^ ~s[1];
  if (!diff) return mx;  // all values equal
  const int top = 31 - __builtin_clz(diff);
  uint32_t prefix = top == 31 ? 0u : (mx & ~((2u << top) - 1u));  // the bits every value shares
  for (int bit = top; bit >= KTH_LOW_BIT; --bit) {
    const uint32_t cand = prefix | (1u << bit);
    uint32_t c = 0;  // wave-uniform
#pragma unroll
    for (int r = 0; r < R; ++r) c += (uint32_t)__builtin_popcountll(__ballot(v[r] >= cand));
    if (lane == 0 && c) atomicAdd(&s[8 + bit], c);
    __syncthreads();
    if (s[8 + bit] >= k) prefix = cand;
  }
  return prefix;
}

// The same search by ONE wave over R values per lane (no LDS, no barrier): a lone query's full pass takes its threshold
// from the sampled lower bounds itself (scan8_kernel, tau_keys) -- every wave, redundantly, in about 2 us, where walking the
// keys through a sorted list cost each wave 8-10 us at the head of the launch.
template <int R>
__device__ __forceinline__ uint32_t wave_kth_threshold(const uint32_t (&v)[R], uint32_t k) {
  uint32_t mx = 0, mn_inv = 0, cnt = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    mx = max(mx, v[r]);
    mn_inv = max(mn_inv, v[r] ? ~v[r] : 0u);
    cnt += (uint32_t)__builtin_popcountll(__ballot(v[r] != 0u));
  }
  for (int o = 32; o > 0; o >>= 1) {
    mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
    mn_inv = max(mn_inv, (uint32_t)__shfl_xor((int)mn_inv, o));
  }
  if (cnt < k) return 0u;
  const uint32_t diff = mx ^ ~mn_inv;
  if (!diff) return mx;
  const int top = 31 - __builtin_clz(diff);
  uint32_t prefix = top == 31 ? 0u : (mx & ~((2u << top) - 1u));
  for (int bit = top; bit >= KTH_LOW_BIT; --bit) {
    const uint32_t cand = prefix | (1u << bit);
    uint32_t c = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) c += (uint32_t)__builtin_popcountll(__ballot(v[r] >= cand));
    if (c >= k) prefix = cand;
  }
  return prefix;
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
  // optional, for the blocking host path: over_out[q] = 1 if the query's candidate count P_dev[q] exceeded P (the host
  // then repairs it after its synchronisation, so no repair launches are queued), else 0
  uint32_t* over_out;
  int no_fast;            // 1: never the register path below (option merge_fast = 0, for A/B and the equality test)
  // a lone blocking query whose LAST kernel this is (one workgroup): done_seq goes into the mapped host word the caller polls
  uint32_t* done_flag;
  uint32_t done_seq;
  uint32_t* done_ticket;  // several workgroups (a small blocking batch): a device counter, zero between launches; the last ticket writes the word
};

// wave_stored: this wave wrote some of the results.  A fence waits for the EXECUTING wave's stores only, so every wave that
// stored makes its own stores visible to the host first; then the barrier; then the word.
__device__ __forceinline__ void merge_signal_done(const MergeArgs& a, bool wave_stored) {  // whole workgroup; after its result stores
  if (a.done_flag) {
    if (wave_stored) __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
      if (gridDim.x == 1) {
        *(volatile uint32_t*)a.done_flag = a.done_seq;
      } else if (atomicAdd(a.done_ticket, 1u) == gridDim.x - 1) {  // (every workgroup's results are visible to the host by now)
        *a.done_ticket = 0;
        __threadfence_system();
        *(volatile uint32_t*)a.done_flag = a.done_seq;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Small merges in registers.  The list walk below loads its keys ON DEMAND -- a dependent global load (~1 us) per step and
// lane -- which is what a merge of a few hundred to a few thousand keys costs (10 us for the 512 partial lists of a 10 k-row
// scan or the ~100 re-scored candidates of a lone query: profiles/r04/lone/).  When every key fits the workgroup's registers
// (at most MERGE_FAST_R per thread) and k is small, all keys are loaded at once, and the k best are extracted by k rounds of
// "wave-wide maximum, remove the winner": two 32-bit DPP reductions per round (score half, then row half among the lanes
// that hold the best score), no LDS, no barrier; the waves' k survivors meet in LDS and wave 0 repeats the rounds on them.
// Keys are unique (row bits) or 0 = absent, so "remove the winner" removes exactly one.
// ------------------------------------------------------------------------------------------------
constexpr int MERGE_FAST_R = 8;  // (MERGE_FAST_K: kernels_common.h)
constexpr int MERGE_MID_K = 512, MERGE_MID_CAP = 1024;  // the medium-k register path: k, and the keys its LDS sort holds

// (wave_max_u32, wave_top_k, MERGE_FAST_K: kernels_common.h)

// n = P * list_len keys of one query -> lane j < k of wave 0 returns the j-th best.  lds: [nwaves][k] keys.  Whole workgroup.
template <int R>
__device__ __forceinline__ u64 merge_small(const MergeArgs& a, const u64* in, uint32_t P, u64* lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int k = a.k;
  const uint32_t n = P * (uint32_t)a.list_len;
  uint32_t hi[R], lo[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t e = (uint32_t)r * blockDim.x + threadIdx.x;
    u64 key = 0;
    if (e < n) {
      // (consecutive threads along whichever index has the unit stride)
      const uint32_t p = a.p_stride == 1 ? e % P : e / (uint32_t)a.list_len;
      const uint32_t i = a.p_stride == 1 ? e / P : e % (uint32_t)a.list_len;
      key = in[(size_t)p * a.p_stride + (size_t)i * a.i_stride];
    }
    hi[r] = (uint32_t)(key >> 32);
    lo[r] = (uint32_t)key;
  }
  const u64 mine = wave_top_k<R>(hi, lo, k, lane);
  if (lane < k) lds[wave * k + lane] = mine;
  __syncthreads();
  u64 fin = 0;
  if (wave == 0) {
    constexpr int R2 = (16 * MERGE_FAST_K + 63) / 64;  // up to 16 waves x MERGE_FAST_K survivors
    uint32_t h2[R2], l2[R2];
#pragma unroll
    for (int r = 0; r < R2; ++r) {
      const int e = r * 64 + lane;
      const u64 key = e < nwaves * k ? lds[e] : 0;
      h2[r] = (uint32_t)(key >> 32);
      l2[r] = (uint32_t)key;
    }
    fin = wave_top_k<R2>(h2, l2, k, lane);
  }
  return fin;
}

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
  if (a.over_out && threadIdx.x == 0) a.over_out[blockIdx.x] = (a.P_dev && a.P_dev[blockIdx.x] > a.P) ? 1u : 0u;
  if (!a.no_fast && k <= MERGE_FAST_K && blockDim.x == 1024 && (u64)P * (u64)a.list_len <= (u64)MERGE_FAST_R * blockDim.x) {
    const uint32_t n = P * (uint32_t)a.list_len;  // (wave-uniform choice of the register count)
    const u64 key = n <= 2 * blockDim.x ? merge_small<2>(a, in, P, lds_lists) : merge_small<MERGE_FAST_R>(a, in, P, lds_lists);
    if (wave == 0) {
      const u64 kth = (u64)__shfl(key, k - 1);
      if (a.out_kth && lane == 0) a.out_kth[blockIdx.x] = kth ? key_score(kth) : -INFINITY;
      const size_t o = (size_t)blockIdx.x * k;
      if (lane < k) {
        const uint32_t row = key_row(key);
        if (a.out_keys) a.out_keys[o + lane] = key ? ((key & 0xFFFFFFFF00000000ull) | (u64)(~(row + a.row_base))) : 0;
        if (a.out_idx) a.out_idx[o + lane] = key ? (int64_t)row + a.idx_base : -1;
        if (a.out_score) {
          float sc = key_score(key);
          if (a.metric == WDBX_METRIC_L2) sc = -sc + 0.0f;
          a.out_score[o + lane] = key ? sc : 0.0f;
        }
      }
    }
    merge_signal_done(a, wave == 0);
    return;
  }
  // Medium k (17 .. 512) with every key in registers: a threshold P <= the k-th largest score from the bitwise search over the
  // keys' score halves (block_kth_threshold: short of the k-th largest by less than 2^-15 relative, so k plus a few keys pass
  // it), the keys >= P compacted into LDS and sorted there (bitonic), the first k written out.  The list walk costs 85-100 us
  // for the ~1 100 re-scored candidates of a k = 100 query (profiles/r04/c3_timeline/); this is its fixed ~10 us.  More than
  // MERGE_MID_CAP keys at or above P (massive exact ties): the list walk below, as before.
  if (!a.no_fast && k > MERGE_FAST_K && k <= MERGE_MID_K && blockDim.x == 1024 && (u64)P * (u64)a.list_len <= (u64)MERGE_FAST_R * blockDim.x) {
    __shared__ uint32_t s_k[KTH_SCRATCH];
    __shared__ uint32_t s_cnt;
    const uint32_t n = P * (uint32_t)a.list_len;
    uint32_t hi[MERGE_FAST_R], lo[MERGE_FAST_R];
#pragma unroll
    for (int r = 0; r < MERGE_FAST_R; ++r) {
      const uint32_t e = (uint32_t)r * blockDim.x + threadIdx.x;
      u64 key = 0;
      if (e < n) {
        const uint32_t p = a.p_stride == 1 ? e % P : e / (uint32_t)a.list_len;
        const uint32_t i = a.p_stride == 1 ? e / P : e % (uint32_t)a.list_len;
        key = in[(size_t)p * a.p_stride + (size_t)i * a.i_stride];
      }
      hi[r] = (uint32_t)(key >> 32);
      lo[r] = (uint32_t)key;
    }
    if (threadIdx.x < KTH_SCRATCH) s_k[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    uint32_t thr32 = block_kth_threshold<MERGE_FAST_R>(hi, (uint32_t)k, s_k);
    if (thr32 == 0u) thr32 = 1u;  // fewer than k keys: every key present (score half != 0) passes
#pragma unroll
    for (int r = 0; r < MERGE_FAST_R; ++r) {
      const bool ok = hi[r] >= thr32;
      const u64 m = __ballot(ok);
      uint32_t base = 0;
      if (m) {  // (wave-uniform)
        if (lane == 0) base = atomicAdd(&s_cnt, (uint32_t)__builtin_popcountll(m));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        const uint32_t pos = base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1));
        if (ok && pos < MERGE_MID_CAP) lds_lists[pos] = ((u64)hi[r] << 32) | lo[r];
      }
    }
    __syncthreads();
    const uint32_t total = s_cnt;
    if (total <= MERGE_MID_CAP) {
      uint32_t npow2 = 2;
      while (npow2 < total) npow2 <<= 1;
      for (uint32_t i = total + threadIdx.x; i < npow2; i += blockDim.x) lds_lists[i] = 0ull;
      __syncthreads();
      for (uint32_t size = 2; size <= npow2; size <<= 1)
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
          for (uint32_t i = threadIdx.x; i < npow2 / 2; i += blockDim.x) {
            const uint32_t l = (i / stride) * 2 * stride + (i % stride), h = l + stride;
            const bool desc = ((l & size) == 0);
            const u64 x = lds_lists[l], y = lds_lists[h];
            if ((x < y) == desc) {
              lds_lists[l] = y;
              lds_lists[h] = x;
            }
          }
          __syncthreads();
        }
      if (a.out_kth && threadIdx.x == 0) {
        const u64 kth = (uint32_t)k <= npow2 ? lds_lists[k - 1] : 0ull;
        a.out_kth[blockIdx.x] = kth ? key_score(kth) : -INFINITY;
      }
      const size_t o = (size_t)blockIdx.x * k;
      for (uint32_t i = threadIdx.x; i < (uint32_t)k; i += blockDim.x) {
        const u64 key = i < npow2 ? lds_lists[i] : 0ull;
        const uint32_t row = key_row(key);
        if (a.out_keys) a.out_keys[o + i] = key ? ((key & 0xFFFFFFFF00000000ull) | (u64)(~(row + a.row_base))) : 0;
        if (a.out_idx) a.out_idx[o + i] = key ? (int64_t)row + a.idx_base : -1;
        if (a.out_score) {
          float sc = key_score(key);
          if (a.metric == WDBX_METRIC_L2) sc = -sc + 0.0f;
          a.out_score[o + i] = key ? sc : 0.0f;
        }
      }
      merge_signal_done(a, wave * 64 < k || wave == 0);
      return;
    }
    __syncthreads();  // (the list walk reuses the LDS)
  }
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
  merge_signal_done(a, wave == 0);
}



// A threshold from the k-th largest SCORE among n keys per query (0 = no key): out_kth[q] <= that score, short of it by less
// than 2^-15 relative (block_kth_threshold); -inf when fewer than k keys.  For n up to KTH_R per thread.
struct KthArgs {
  const u64* keys;
  uint64_t q_stride;
  uint32_t n;
  int k;
  float* out_kth;
};
constexpr int KTH_R = 16;
__global__ __launch_bounds__(1024) void kth_score_kernel(KthArgs a) {
  __shared__ uint32_t s_k[KTH_SCRATCH];
  if (threadIdx.x < KTH_SCRATCH) s_k[threadIdx.x] = 0;
  const u64* in = a.keys + (size_t)blockIdx.x * a.q_stride;
  uint32_t v[KTH_R];
#pragma unroll
  for (int r = 0; r < KTH_R; ++r) {
    const uint32_t i = (uint32_t)r * blockDim.x + threadIdx.x;
    v[r] = i < a.n ? (uint32_t)(__builtin_nontemporal_load(in + i) >> 32) : 0u;
  }
  __syncthreads();
  const uint32_t ord = block_kth_threshold<KTH_R>(v, (uint32_t)a.k, s_k);
  if (threadIdx.x == 0) a.out_kth[blockIdx.x] = ord ? ord2f(ord) : -INFINITY;
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
  uint32_t total;  // valid keys seen (set by the first pick)
  uint32_t hist[256];
};

// Where a select chain reads its keys: fixed by the host (src == null in the kernels below) or chosen on the
// device (select_source_kernel) between a query's re-scored candidates and its full key-per-row dump.
struct SelectSrc {
  const u64* keys;
  u64 n;
};

// the selection scan's large-k epilogue: candidates if their buffer held them all, else the repair scan's dump
__global__ void select_source_kernel(SelectSrc* src, const uint32_t* count, uint32_t cap, const u64* cand, const u64* dump,
                                     u64 n_rows) {
  if (threadIdx.x == 0) {
    const bool over = *count > cap;
    src->keys = over ? dump : cand;
    src->n = over ? n_rows : (u64)*count;
  }
}

// k-th largest key of a finished chain -> its ranking value, or -inf when fewer than k valid keys were seen
__global__ void select_kth_value_kernel(const SelectState* st, uint32_t k, float* out) {
  if (threadIdx.x == 0) *out = (st->total >= k) ? key_score(st->prefix) : -INFINITY;
}

__global__ void select_init_kernel(SelectState* st, uint32_t k) {
  if (threadIdx.x == 0) {
    st->prefix = 0;
    st->mask = 0;
    st->need = k;
    st->out_count = 0;
    st->total = 0;
  }
  st->hist[threadIdx.x] = 0;
}

__global__ __launch_bounds__(256) void radix_hist_kernel(const u64* __restrict__ keys, u64 n, SelectState* st, int shift,
                                                         const SelectSrc* src) {
  if (src) {
    keys = src->keys;
    n = src->n;
  }
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
    if (shift == 56) st->total = run;
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
                                                            uint32_t k, const SelectSrc* src) {
  if (src) {
    keys = src->keys;
    n = src->n;
  }
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
