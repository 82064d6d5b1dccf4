// kernels_common.h -- device helpers shared by every kernel: 64-bit ordering keys, the per-wave top-k list, lane-group reductions.
// Part of the single translation unit wdbx_hip.hip (included there, in order); not a standalone header.

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
  uint32_t y_partials;  // grid rows > 1 (a round's repair launches in one grid): u64s between consecutive queries' partials
  // scan_kernel_listed (repairs behind a batch): {n, q_0 .. q_(n-1)} = the queries to scan; grid row y takes q_y, q_(y + rows), ...
  const uint32_t* over_list;
};

// ------------------------------------------------------------------------------------------------
// wave-wide top-k of keys held in registers (merge_small in kernels_merge_select.h)
// ------------------------------------------------------------------------------------------------
constexpr int MERGE_FAST_K = 16;  // the largest k of the register merges

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, false));   // quad_perm [1, 0, 3, 2]
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e, 0xf, 0xf, false));   // quad_perm [2, 3, 0, 1]
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false));  // row_ror:4
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false));  // row_ror:8: every lane its row's maximum
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));  // row_bcast:15 into rows 1 and 3
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));  // row_bcast:31 into rows 2 and 3
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// the wave's k largest keys among R per lane, in descending order: lane j < k returns the j-th (0 = fewer than j + 1 keys)
template <int R>
__device__ __forceinline__ u64 wave_top_k(uint32_t (&hi)[R], uint32_t (&lo)[R], int k, int lane) {
  u64 out = 0;
  for (int j = 0; j < k; ++j) {
    uint32_t mh = hi[0];
#pragma unroll
    for (int r = 1; r < R; ++r) mh = max(mh, hi[r]);
    const uint32_t H = wave_max_u32(mh);
    uint32_t ml = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) ml = max(ml, hi[r] == H ? lo[r] : 0u);
    const uint32_t L = wave_max_u32(ml);
    if ((H | L) == 0u) break;  // (wave-uniform) no key left
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool won = hi[r] == H && lo[r] == L;
      hi[r] = won ? 0u : hi[r];
      lo[r] = won ? 0u : lo[r];
    }
    if (lane == j) out = ((u64)H << 32) | L;
  }
  return out;
}

