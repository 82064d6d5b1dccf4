// host_index.h -- host side: the handle, kernel choice and the enqueue functions of every search path.
// Part of the single translation unit wdbx_hip.hip (included there, in order); not a standalone header.

// ------------------------------------------------------------------------------------------------
// host side: the handle
// ------------------------------------------------------------------------------------------------
struct EventPool {
  std::vector<hipEvent_t> ev;        // pairs
  std::vector<uint32_t> launches;    // kernel launches bracketed by each pair (back-to-back launches of ONE kernel share a pair:
  size_t used = 0;                   //  an event record between two kernels costs ~10 us of idle stream)
};

constexpr int STAGE_SLOTS = 4;

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
  uint64_t exchanges = 0;  // all-gather + merge steps this handle has enqueued (read-only option "exchanges")
  // pinned, device-mapped staging for small blocking searches: the kernels read the query from and
  // write the result to host memory directly (no memcpy calls on the latency path)
  char* h_stage = nullptr;
  char* h_stage_dev = nullptr;
  // STAGE_SLOTS staging blocks: a small blocking call owns one from its enqueue to the moment it has read its results, and
  // waits for the GPU on the slot's event with the handle's mutex released (search_host)
  SlotPool<4> slots;   // (host_dispatch.h; = STAGE_SLOTS)
  hipEvent_t slot_done[4] = {nullptr, nullptr, nullptr, nullptr};
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
  bool last_batch_repaired = false;  // the last batch's overflowed queries were repaired on the device (conditional launches)
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
  float* d_gmax = nullptr;     // largest squared row norm per 64-row group (selection bounds of the tile kernels)
  float* d_qn = nullptr;       // |q| per query of a block
  size_t gmax_bytes = 0, qn_bytes = 0;
  bool gmax_valid = false;
  bool cn_stats_dirty = false; // cached norms were refreshed in place: recompute their statistics before deciding
  bool group_bounds = false;   // per-group bounds (norms vary a lot) or one global bound (they do not: cheaper epilogue)
  void* d_selsrc = nullptr;    // device-side SelectSrc of the large-k selection epilogue
  size_t selsrc_bytes = 0;
  size_t rows8_bytes = 0, scale8_bytes = 0;
  uint64_t shadow8_rows = 0;
  uint32_t pitch8 = 0;
  uint64_t u8_no_room_cap = ~0ull;  // capacity at which the u8 shadow last failed to allocate
  uint32_t* defer_flag_dev = nullptr;  // non-null during a blocking call that repairs overflow itself (mapped host word)
  // a lone blocking call through a staging slot: the LAST kernel of its chain writes done_seq into done_flag_dev (a mapped
  // host word of the slot) and the caller polls that word instead of waiting on the runtime; done_signals counts the launches
  // that took the signal (exactly one, or the caller waits on its event as before)
  uint32_t* done_flag_dev = nullptr;
  uint32_t done_seq = 0, lone_seq = 0;
  int done_signals = 0;
  uint32_t* d_ticket = nullptr;        // merge_kernel's workgroup ticket for that signal when a small batch is merged (zero between launches)
  // a lone blocking query whose final top-k the HOST takes (search_host): the re-scored candidates' keys and their count go
  // to these mapped host locations and no final merge is launched; lone_cap_max = keys the host area holds
  u64* lone_keys_dev = nullptr;
  uint32_t* lone_count_dev = nullptr;
  uint32_t lone_cap_max = 0;
  bool lone_used = false;              // set by the enqueue when it took that form
  int last_single_path = 0;    // 0 fp32 scan, 1 bf16 tiles, 2 u8 scan (what the last single-query search ran on)
  int last_sample_qn = 1;      // queries per workgroup of the last u8 sample launch (1, 3 or 4: scan8_sample4_kernel)
  uint32_t* d_cnmax = nullptr;
  size_t cnmax_bytes = 0;
  // group-scaled i8 shadow copy of the rows for the int8 tiles (kernels_tiles8.h): rows [0, shadowg_rows) quantised
  int8_t* d_rows8g = nullptr;
  f4* d_groups8 = nullptr;       // per 64-row group {s_g, a_g, b_g, vouch}
  uint32_t* d_over_list = nullptr;  // {n, q_0 ..}: the overflowed queries of a batch block, for its repair scan (mark_lost_kernel)
  float* d_gref8 = nullptr;      // {a_ref, b_ref, sum a, sum b} + a counter: the prefilter epilogue's ordinary-group bounds
  bool gref_valid = false;       // (recomputed when the group table changed)
  u64* d_gbad8 = nullptr;        // per 64-row group: bits of the rows that hold a NaN (removed rows) or lie past the end
  size_t rows8g_bytes = 0, groups8_bytes = 0;
  uint64_t shadowg_rows = 0;
  uint64_t shadowg_tail_n = ~0ull;  // the row count for which the groups behind the last row (to the end of its 256-row tile) were last written
  uint32_t pitch8g = 0;
  uint64_t i8g_no_room_cap = ~0ull;
  int8_t* d_qb8 = nullptr;       // the query block as i8
  f4* d_qpar = nullptr;          // its per-query parameters
  size_t qb8_bytes = 0, qpar_bytes = 0;
  u64* d_pairs = nullptr;        // the tile waves' candidate (query, row) pairs and how many each wave produced
  uint32_t* d_pair_count = nullptr;
  size_t pairs_bytes = 0, pair_count_bytes = 0;
  // profiling
  bool profile = false;
  EventPool scan_ev, merge_ev, gemm_ev, sample_ev;
  // options
  int64_t opt_lanes = 0, opt_blocks = 0, opt_nt = 1, opt_blocked = 0, opt_batch = 32, opt_generic = 0;
  int64_t opt_group_bounds = -1, opt_single_min_rows = 131072, opt_scan8_wgs = -1, opt_scan8_per_query = -1,
          opt_scan8_ablate = 0, opt_batch_repair = 1, opt_scan_shadow = 2, opt_gemm_bf16 = 3, opt_gemm8_variant = 0,
          opt_gemm8_refine = 1, opt_scan8_sample4 = 1, opt_gemm_l2 = 1, opt_gemm_l2_i8 = 1, opt_force_ragged = 0,
          opt_gemm_ct = 0, opt_wg_merge = 1, opt_zero_copy = 1, opt_lone_host_select = 1, opt_lds_lists = 0,
          opt_merge_fast = 1, opt_poll_done = 1, opt_scan_one_grid = 1, opt_select_min_k = 200, opt_gemm_min_nq = 4, opt_gemm_min_rows = 65536, opt_gemm_min_work = 800000,
          opt_gemm_sample_div = 0;
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

// scratch buffers on the device: grown on demand, contents never preserved (bookkeeping: grow_with, host_dispatch.h)
static int grow(void** p, size_t* have, size_t need) {
  return grow_with(
      p, have, need,
      [](void** np, size_t bytes) -> int {
        HIP_TRY(hipMalloc(np, bytes));
        return WDBX_OK;
      },
      [](void* old) -> int {
        HIP_TRY(hipFree(old));
        return WDBX_OK;
      });
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

template <int L>
static scan_fn pick_listed_metric(int metric, int reg) {
  if (metric == WDBX_METRIC_COSINE) return reg ? scan_kernel_listed<L, WDBX_METRIC_COSINE, 1> : scan_kernel_listed<L, WDBX_METRIC_COSINE, 0>;
  return reg ? scan_kernel_listed<L, WDBX_METRIC_L2, 1> : scan_kernel_listed<L, WDBX_METRIC_L2, 0>;
}

// the listed repair scan (generic body): reg = 1 register lists (k <= 128), 0 LDS lists
static scan_fn pick_listed(int L, int metric, int reg) {
  switch (L) {
    case 64: return pick_listed_metric<64>(metric, reg);
    case 1: return pick_listed_metric<1>(metric, reg);
    case 2: return pick_listed_metric<2>(metric, reg);
    case 4: return pick_listed_metric<4>(metric, reg);
    default: return pick_listed_metric<8>(metric, reg);
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

// listed: the plan of the listed repair scan (scan_kernel_listed, generic body) instead of the handle's own scan kernel
static int plan_scan(wdbx_index* ix, int k, LaunchPlan* out, bool listed = false) {
  LaunchPlan lp;
  if (listed) {
    const int pitch4 = ix->pitch / 4;
    int L = 1;
    while (L < 8 && L < pitch4) L <<= 1;
    if (pitch4 > 768) L = 64;
    lp.sc.fn = pick_listed(L, ix->metric, (k <= 128 && !ix->opt_lds_lists) ? 1 : 0);
    lp.sc.L = L;
    lp.sc.generic = true;
    lp.sc.lds_extra = (size_t)pitch4 * 16;
  } else {
    lp.sc = choose_scan(ix, k);
  }
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

static int record(EventPool& pool, bool enabled, hipStream_t s, bool start, uint32_t launches = 1) {
  if (!enabled) return WDBX_OK;
  if (start) {
    if (pool.used + 2 > pool.ev.size()) {
      if (pool.ev.size() >= 2 * 65536) return WDBX_OK;  // pool exhausted: stop sampling
      for (int i = 0; i < 2; ++i) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        pool.ev.push_back(e);
      }
      pool.launches.resize(pool.ev.size() / 2, 1);
    }
    HIP_TRY(hipEventRecord(pool.ev[pool.used], s));
  } else if (pool.used + 2 <= pool.ev.size()) {
    HIP_TRY(hipEventRecord(pool.ev[pool.used + 1], s));
    pool.launches[pool.used / 2] = launches;
    pool.used += 2;
  }
  return WDBX_OK;
}

static int launch_merge(wdbx_index* ix, const MergeArgs& m_in, int nq) {
  MergeArgs m = m_in;
  m.no_fast = ix->opt_merge_fast ? 0 : 1;
  if (ix->done_flag_dev && (nq == 1 || ix->d_ticket) && (m.out_idx || m.out_score) && !m.only_if_over) {  // the call's final ranking
    m.done_flag = ix->done_flag_dev;
    m.done_seq = ix->done_seq;
    m.done_ticket = ix->d_ticket;
    ++ix->done_signals;
  }
  const int nw = merge_waves_for(m.k);
  size_t lds = (size_t)(nw + 1) * m.k * sizeof(u64);
  if (m.k > MERGE_FAST_K && m.k <= MERGE_MID_K) lds = std::max(lds, (size_t)MERGE_MID_CAP * sizeof(u64));  // (its LDS sort)
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
  ++ix->exchanges;
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
// handle's communicator + second merge); 2 = only this shard's key list (global rows) into keys_dst [nq, k] (a buffer
// of the caller, who runs the exchange: the in-process shard group, host_group.h) or, without one, d_local_keys
enum { SEARCH_FINAL = 0, SEARCH_SHARDED = 1, SEARCH_LOCAL_KEYS = 2 };

static int enqueue_search_gemm(wdbx_index* ix, const float* d_queries, int nq, int k, int64_t* d_out_idx, float* d_out_score,
                               int mode, int count_slot, u64* keys_out);
static int enqueue_search_gemm8(wdbx_index* ix, const float* d_queries, int nq, int k, int64_t* d_out_idx, float* d_out_score, int mode,
                                u64* keys_out);
static bool shadow_single_eligible(const wdbx_index* ix, int k, int nq_call);
static bool u8_single_eligible(const wdbx_index* ix, int k, int nq_call);
static int enqueue_singles_u8(wdbx_index* ix, const float* d_queries, int nq, int k, int64_t* d_out_idx, float* d_out_score,
                              u64* keys_out, bool candidates_only = false);
static bool prepare_u8_shadow(wdbx_index* ix);

static int enqueue_search(wdbx_index* ix, const float* d_queries, int nq, int k, int64_t* d_out_idx,
                          float* d_out_score, int mode, u64* keys_dst = nullptr) {
  const bool keys_only = mode == SEARCH_LOCAL_KEYS;
  const bool sharded = mode != SEARCH_FINAL;  // the local stage ends in keys with global rows
  if (nq <= 0) return WDBX_OK;
  if (k < 1 || k > WDBX_MAX_K) return fail(WDBX_E_INVALID, "k=%d outside [1, %d]", k, WDBX_MAX_K);
  if (!d_queries || (!keys_only && (!d_out_idx || !d_out_score))) return fail(WDBX_E_INVALID, "null device buffer");
  if (mode == SEARCH_SHARDED && !ix->comm) return fail(WDBX_E_STATE, "sharded search before wdbx_index_comm_init");
  if (ix->n >= 0xFFFFFF00ull) return fail(WDBX_E_INVALID, "shard holds too many rows for 32-bit row keys");

  const int batch = keys_only ? nq : (int)std::max<int64_t>(1, std::min<int64_t>(ix->opt_batch, 1024));
  int rc;
  if (ix->n == 0 && !keys_dst) {
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
  // SEARCH_SHARDED: the rounds of `batch` queries keep their size (the small kernels around the scans are shared per round),
  // but the all-gather + merge are issued once per EXCHANGE CHUNK of up to 1024 queries: an all-gather costs ~0.2 ms of
  // stream time whatever it carries (measured with one rank: 87.8 vs 81.0 us per query at 1.25 M rows when issued per 32)
  const int xch = keys_only ? nq : batch * std::max(1, 1024 / batch);
  if (sharded) {
    if (!keys_dst) {
      rc = grow((void**)&ix->d_local_keys, &ix->local_keys_bytes, (size_t)std::min(xch, std::max(nq, batch)) * k * sizeof(u64));
      if (rc) return rc;
    }
    if (!keys_only) {
      rc = grow((void**)&ix->d_gathered, &ix->gathered_bytes, (size_t)ix->nranks * std::min(xch, std::max(nq, batch)) * k * sizeof(u64));
      if (rc) return rc;
    }
  }

  u64* const lbase = keys_dst ? keys_dst : ix->d_local_keys;  // (keys_only: one batch, so offsets within it are offsets in keys_dst)
  for (int q0 = 0; q0 < nq; q0 += batch) {
    const int b = std::min(batch, nq - q0);
    u64* const lkeys = lbase ? lbase + (size_t)(q0 % xch) * k : nullptr;  // this round's lists inside the exchange chunk
    if (select) {
      // large k: per query  scan (key per row) -> radix select -> compact -> sort
      const uint32_t sgrid_rows = (uint32_t)std::min<uint64_t>((ix->n + 255) / 256, (uint64_t)ix->cu_count * 16);
      uint32_t npow2 = 2;
      while (npow2 < (uint32_t)k) npow2 <<= 1;
      // On the u8 selection scan the chain below ranks the query's re-scored CANDIDATES; the key-per-row fp32 scan
      // is then only the conditional repair of an overflowed candidate buffer, and a device-side descriptor picks
      // which of the two the chain reads.
      const bool u8 = u8_single_eligible(ix, k, nq) && prepare_u8_shadow(ix);  // (also for the in-process group's local stage)
      ix->last_single_path = u8 ? 2 : 0;
      SelectSrc* src = nullptr;
      if (u8) {
        if ((rc = grow((void**)&ix->d_count, &ix->count_bytes, ((size_t)batch + 2 * GB_N) * sizeof(uint32_t)))) return rc;
        if ((rc = grow((void**)&ix->d_selsrc, &ix->selsrc_bytes, sizeof(SelectSrc)))) return rc;
        src = (SelectSrc*)ix->d_selsrc;
      }
      for (int q = 0; q < b; ++q) {
        ScanArgs sa = {};
        if (u8) {
          rc = enqueue_singles_u8(ix, d_queries + (size_t)(q0 + q) * ix->pitch, 1, k, nullptr, nullptr, nullptr, true);
          if (rc) return rc;
          sa.only_if_over = ix->d_count;
          sa.over_cap = ix->last_batch_cap;
        }
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
        if (!u8 && (rc = record(ix->scan_ev, ix->profile, ix->stream, true))) return rc;
        hipLaunchKernelGGL(lp.sc.fn, dim3(lp.blocks), dim3(256), lp.lds, ix->stream, sa);
        HIP_TRY(hipGetLastError());
        if (!u8 && (rc = record(ix->scan_ev, ix->profile, ix->stream, false))) return rc;
        if ((rc = record(ix->merge_ev, ix->profile, ix->stream, true))) return rc;
        // (on the selection scan the chain's grid is sized for the candidate buffer; the rare repair walks the N keys
        // with the same, smaller grid)
        const uint32_t sgrid = u8 ? std::min<uint32_t>(sgrid_rows, (ix->last_batch_cap + 255) / 256) : sgrid_rows;
        if (u8)
          hipLaunchKernelGGL(select_source_kernel, dim3(1), dim3(64), 0, ix->stream, src, (const uint32_t*)ix->d_count,
                             ix->last_batch_cap, (const u64*)ix->d_cand, (const u64*)ix->d_dump, (u64)ix->n);
        hipLaunchKernelGGL(select_init_kernel, dim3(1), dim3(256), 0, ix->stream, ix->d_state, (uint32_t)k);
        for (int shift = 56; shift >= 0; shift -= 8) {
          hipLaunchKernelGGL(radix_hist_kernel, dim3(sgrid), dim3(256), 0, ix->stream, (const u64*)ix->d_dump, (u64)ix->n,
                             ix->d_state, shift, (const SelectSrc*)src);
          hipLaunchKernelGGL(radix_pick_kernel, dim3(1), dim3(256), 0, ix->stream, ix->d_state, shift);
        }
        hipLaunchKernelGGL(radix_compact_kernel, dim3(sgrid), dim3(256), 0, ix->stream, (const u64*)ix->d_dump, (u64)ix->n,
                           ix->d_state, ix->d_sel, (uint32_t)k, (const SelectSrc*)src);
        MergeArgs m = {};
        m.k = k;
        m.metric = ix->metric;
        m.row_base = (uint32_t)ix->row_base;
        m.out_keys = sharded ? lkeys + (size_t)q * k : nullptr;
        m.out_idx = sharded ? nullptr : d_out_idx + (size_t)(q0 + q) * k;
        m.out_score = sharded ? nullptr : d_out_score + (size_t)(q0 + q) * k;
        hipLaunchKernelGGL(sort_out_kernel, dim3(1), dim3(1024), (size_t)npow2 * sizeof(u64), ix->stream,
                           (const u64*)ix->d_sel, (const SelectState*)ix->d_state, m, npow2);
        HIP_TRY(hipGetLastError());
        if ((rc = record(ix->merge_ev, ix->profile, ix->stream, false))) return rc;
      }
    } else if (ix->n) {
      // Single queries on a reduced-precision shadow copy: each query makes ITS OWN selection pass (threshold from
      // a sample, every row whose score could reach it under a rigorous error bound becomes a candidate, the
      // candidates are re-scored in fp32 from the fp32 rows).  Preferred: the u8 scan (a quarter of the fp32 bytes,
      // query kept in fp32, per-row bounds, honours row masks); else the bf16 tile kernel with one live column
      // (half the bytes).  The fp32 scan and its merge still follow, but as REPAIR launches that return at once
      // unless that query's candidate buffer overflowed (massive near-duplicates) -- the result is exact either
      // way, without a host round trip (a blocking caller with one query does the check itself: defer_flag_dev).
      const bool u8 = u8_single_eligible(ix, k, nq) && prepare_u8_shadow(ix);  // (also for the in-process group's local stage)
      const bool shadow = u8 || (!keys_only && shadow_single_eligible(ix, k, nq));
      ix->last_single_path = u8 ? 2 : shadow ? 1 : 0;
      if (shadow) {
        if ((rc = grow((void**)&ix->d_count, &ix->count_bytes, ((size_t)batch + 2 * GB_N) * sizeof(uint32_t)))) return rc;
        if (u8)  // the u8 selection scan: a quarter of the fp32 bytes per query
          rc = enqueue_singles_u8(ix, d_queries + (size_t)q0 * ix->pitch, b, k, d_out_idx ? d_out_idx + (size_t)q0 * k : nullptr,
                                  d_out_score ? d_out_score + (size_t)q0 * k : nullptr, sharded ? lkeys : nullptr);
        else     // the bf16 tile kernel with one live column: half the fp32 bytes
          rc = enqueue_search_gemm(ix, d_queries + (size_t)q0 * ix->pitch, b, k, d_out_idx + (size_t)q0 * k,
                                   d_out_score + (size_t)q0 * k, SEARCH_FINAL, 0, sharded ? lkeys : nullptr);
        if (rc) return rc;
      }
      if (u8 && ix->defer_flag_dev) continue;  // the blocking caller repairs an overflow after its synchronisation
      // shadow paths: the fp32 scans below are REPAIR launches (they return at once unless the query's candidate buffer
      // overflowed), so the round's b of them go out as ONE grid of b rows (4.4 us per empty launch otherwise);
      // fp32 path: one timed launch per query -- or, for a round of several queries over a corpus of at most 1 GiB, ONE grid
      // with a row per query (blockIdx.y, as the repair launches): every query still makes its own pass over all rows, but the
      // passes run side by side out of the caches and the launches' fixed ~9 us are paid once per round, not per query
      // (10 k rows: 8 queries 90 -> 3x us, profiles/r04/small_batch/).  Larger corpora: a launch per query, as the u8 scan
      // does beyond 1 GiB (two passes streaming different regions at once cost more than the gaps between launches)
      const bool one_grid = !shadow && b > 1 && ix->opt_scan_one_grid && (uint64_t)ix->n * ix->pitch * sizeof(float) <= (1ull << 30);
      for (int q = 0; q < ((shadow || one_grid) ? 1 : b); ++q) {
        ScanArgs sa = {};
        if (shadow) {
          sa.only_if_over = ix->d_count + q;
          sa.over_cap = ix->last_batch_cap;
        }
        if (shadow || one_grid) sa.y_partials = (uint32_t)((size_t)k * lp.P);
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
        hipLaunchKernelGGL(lp.sc.fn, dim3(lp.blocks, (shadow || one_grid) ? b : 1), dim3(256), lp.lds, ix->stream, sa);
        HIP_TRY(hipGetLastError());
        rc = shadow ? WDBX_OK : record(ix->scan_ev, ix->profile, ix->stream, false, one_grid ? (uint32_t)b : 1u);
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
      m.out_keys = sharded ? lkeys : nullptr;
      m.out_idx = sharded ? nullptr : d_out_idx + (size_t)q0 * k;
      m.out_score = sharded ? nullptr : d_out_score + (size_t)q0 * k;
      rc = launch_merge(ix, m, b);
      if (rc) return rc;
    } else if (sharded) {
      HIP_TRY(hipMemsetAsync(lkeys, 0, (size_t)b * k * sizeof(u64), ix->stream));
    } else {
      // no rows: idx = -1 (all bits set), score = 0
      HIP_TRY(hipMemsetAsync(d_out_idx + (size_t)q0 * k, 0xFF, (size_t)b * k * sizeof(int64_t), ix->stream));
      HIP_TRY(hipMemsetAsync(d_out_score + (size_t)q0 * k, 0, (size_t)b * k * sizeof(float), ix->stream));
    }
    if (mode == SEARCH_SHARDED && ((q0 + b) % xch == 0 || q0 + b == nq)) {  // the chunk is complete: exchange it
      const int c0 = q0 / xch * xch, c = q0 + b - c0;
      rc = exchange_and_merge(ix, c, k, d_out_idx + (size_t)c0 * k, d_out_score + (size_t)c0 * k);
      if (rc) return rc;
    }
  }
  return WDBX_OK;
}


// The conditional repair of a batched block (the device-side form of what wdbx_index_search does on the host): for each
// of the block's nv queries whose candidate counter exceeds the capacity -- its selection result may be incomplete -- the
// exact fp32 scan + merge, as ONE grid of nv rows that returns at once for every other query (4 us when nothing
// overflowed).  Writes the same outputs the block's final merge wrote.  *done = false when this shape has no device-side
// repair (k in the radix-select range, or partial lists beyond 256 MiB): the caller then leaves it to the host, as before.
static int enqueue_batch_repair(wdbx_index* ix, const float* qsrc, int nv, int k, uint32_t* d_count_block, uint32_t cap,
                                int64_t* out_idx, float* out_score, u64* out_keys, bool* done, const uint32_t* d_lost = nullptr) {
  *done = false;
  int rc;
  LaunchPlan lp;
  bool go = ix->n && !use_select(ix, k) && ix->opt_batch_repair;
  size_t need = 0;
  if (go) {
    if ((rc = plan_scan(ix, k, &lp, true))) return rc;
    need = (size_t)nv * k * lp.P * sizeof(u64);
    if (need > ((size_t)256 << 20)) go = false;
  }
  if (go && !ix->d_over_list) HIP_TRY(hipMalloc((void**)&ix->d_over_list, (GB_N + 1) * sizeof(uint32_t)));
  // (after the exact passes consumed the buffers) a lost pair marks every query of the BLOCK as overflowed -- nobody knows
  // which of them lost a candidate; and the overflowed queries are listed for the repair scan
  if (go || d_lost) {
    hipLaunchKernelGGL(mark_lost_kernel, dim3(1), dim3(256), 0, ix->stream, d_count_block, (uint32_t)nv, d_lost, cap,
                       go ? ix->d_over_list : (uint32_t*)nullptr);
    HIP_TRY(hipGetLastError());
  }
  if (!go) return WDBX_OK;
  if ((rc = grow((void**)&ix->d_partials, &ix->partials_bytes, need))) return rc;
  ScanArgs sa = {};
  sa.over_list = ix->d_over_list;
  sa.y_partials = (uint32_t)((size_t)k * lp.P);
  sa.rows = (const f4*)ix->d_rows;
  sa.query = (const f4*)qsrc;
  sa.partials = ix->d_partials;
  sa.mask = nullptr;
  sa.n_rows = (uint32_t)ix->n;
  sa.pitch4 = (uint32_t)(ix->pitch / 4);
  sa.groups = lp.groups;
  sa.chunk = lp.chunk;
  sa.k = k;
  sa.wg_merge = lp.wg_merge ? 1 : 0;
  // a few grid rows share the listed queries (none listed: the grid returns at once)
  hipLaunchKernelGGL(lp.sc.fn, dim3(lp.blocks, std::min(nv, 4)), dim3(256), lp.lds, ix->stream, sa);
  HIP_TRY(hipGetLastError());
  MergeArgs m = {};
  m.only_if_over = d_count_block;
  m.over_cap = cap;
  m.list_len = k;
  m.in = ix->d_partials;
  m.q_stride = (uint64_t)k * lp.P;
  m.i_stride = lp.P;
  m.p_stride = 1;
  m.P = lp.P;
  m.k = k;
  m.metric = ix->metric;
  m.row_base = (uint32_t)ix->row_base;
  m.out_keys = out_keys;
  m.out_idx = out_idx;
  m.out_score = out_score;
  if ((rc = launch_merge(ix, m, nv))) return rc;
  *done = true;
  return WDBX_OK;
}

// ---- batched queries on the MFMA path ----------------------------------------------------------
static bool i8_tiles_eligible(const wdbx_index* ix);
static bool gemm_eligible(const wdbx_index* ix, int nq, int k) {
  if (ix->metric == WDBX_METRIC_L2 && !ix->opt_gemm_l2) return false;
  // From how many queries a call shares ONE pass: gemm_min_queries (4).  On the i8 tiles a small batch costs little more
  // than one scan (10 M x 384: 0.71 ms for up to 16 queries against 0.60 ms per single query), so on large shards two
  // coalesced callers already share a pass -- unless the option was raised to keep batching off.
  int64_t min_nq = ix->opt_gemm_min_nq;
  if (min_nq > 2 && min_nq <= 4 && ix->n >= 3000000 && ix->i8g_no_room_cap != ix->cap && i8_tiles_eligible(ix)) min_nq = 2;
  if (nq < min_nq || (uint64_t)k * 8 * GB_M > ix->n) return false;
  if ((int64_t)ix->n >= 2 * ix->opt_gemm_min_rows) return true;
  // Below 2 x gemm_min_rows the alternative is the fp32 scan with a round's queries as one grid, whose cost grows with
  // queries x rows (0.2-0.25 ns each; cheaper per row on the larger corpora) while a matrix-core batch costs its ~130-180 us of
  // launches whatever it holds: the tiles from queries x rows >= gemm_min_work below gemm_min_rows (measured, d = 384: 21 k rows
  // from 36 queries, 40 k from 18, 60 k from 12) and from 0.65 of it between gemm_min_rows and twice that (70 k rows from 8
  // queries, 100 k from 6): profiles/r04/small_batch/
  if (ix->opt_gemm_min_work <= 0) return (int64_t)ix->n >= ix->opt_gemm_min_rows;
  const uint64_t work = (int64_t)ix->n >= ix->opt_gemm_min_rows ? (uint64_t)ix->opt_gemm_min_work * 13 / 20 : (uint64_t)ix->opt_gemm_min_work;
  return (uint64_t)nq * ix->n >= work;
}

// single queries take the shadow selection pipeline (see enqueue_search) when the bf16 shadow is in use, no row
// mask is active (the tiles do not read masks) and k is served by the list kernels (the repair launch)
// rows from which a selection path pays: its 6 launches pipeline behind each other when a call carries several
// queries, but a lone query waits for each of them (measured, d = 384: 65 k rows 60 vs 41 us, equal at 262 k)
static inline int64_t selection_min_rows(const wdbx_index* ix, int nq_call) {
  return nq_call > 1 ? ix->opt_gemm_min_rows : std::max(ix->opt_gemm_min_rows, ix->opt_single_min_rows);
}

static bool shadow_single_eligible(const wdbx_index* ix, int k, int nq_call) {
  if (ix->opt_scan_shadow <= 0 || ix->opt_gemm_bf16 < 2 || ix->active_mask || use_select(ix, k)) return false;
  if (ix->metric == WDBX_METRIC_L2 && !ix->opt_gemm_l2) return false;
  // the shadow pads rows to 128 elements: for short rows it is no smaller than the fp32 rows (d = 32: twice
  // the bytes, measured 0.54x; d = 64: 0.98x; d = 100: 1.4x) -- worth it from 0.8 of the fp32 bytes down
  const uint64_t pitch16 = ((uint64_t)ix->pitch + 127) / 128 * 128;
  if (pitch16 * 2 * 10 > (uint64_t)ix->pitch * 4 * 8) return false;
  return (int64_t)ix->n >= selection_min_rows(ix, nq_call) && (uint64_t)k * 8 * GB_M <= ix->n;
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

static bool u8_single_eligible(const wdbx_index* ix, int k, int nq_call) {
  if (ix->opt_scan_shadow < 2) return false;  // (row masks and k >= 200 are served by the u8 scan too)
  const Scan8Shape* sh = scan8_shape((uint32_t)ix->dim);
  // worth it from 0.6 of the fp32 bytes down (d = 32 would read as many bytes as the fp32 row)
  if (!sh || (uint64_t)sh->pieces * 16 * 10 > (uint64_t)ix->pitch * 4 * 6) return false;
  return (int64_t)ix->n >= selection_min_rows(ix, nq_call) && (uint64_t)k * 8 * GB_M <= ix->n;
}

// Allocates / refreshes the u8 shadow and its scales for the rows added or overwritten since the last search.
// false = no room for it on the device (remembered per capacity): the caller stays on the other paths.
static bool prepare_u8_shadow(wdbx_index* ix) {
  const Scan8Shape* sh = scan8_shape((uint32_t)ix->dim);
  if (!sh) return false;
  const uint32_t pitch8 = sh->pieces * 16;
  const size_t need = ((size_t)ix->cap + TILE_PAD_ROWS) * pitch8, need_s = ((size_t)ix->cap + TILE_PAD_ROWS) * sizeof(float);
  if (ix->rows8_bytes < need || ix->scale8_bytes < need_s || ix->pitch8 != pitch8) {
    if (ix->u8_no_room_cap == ix->cap) return false;
    if (ix->d_rows8) (void)hipFree(ix->d_rows8);
    if (ix->d_scale8) (void)hipFree(ix->d_scale8);
    ix->d_rows8 = nullptr;
    ix->d_scale8 = nullptr;
    ix->rows8_bytes = ix->scale8_bytes = 0;
    ix->shadow8_rows = 0;
    if (hipMalloc((void**)&ix->d_rows8, need) != hipSuccess || hipMalloc((void**)&ix->d_scale8, need_s) != hipSuccess) {
      (void)hipGetLastError();
      if (ix->d_rows8) (void)hipFree(ix->d_rows8);
      ix->d_rows8 = nullptr;
      ix->u8_no_room_cap = ix->cap;
      return false;
    }
    ix->rows8_bytes = need;
    ix->scale8_bytes = need_s;
    ix->pitch8 = pitch8;
  }
  if (ix->shadow8_rows < ix->n) {
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((ix->n - ix->shadow8_rows + 3) / 4, 65536);
    hipLaunchKernelGGL(rows_to_u8_kernel, dim3(blocks), dim3(256), 0, ix->stream, (const float*)ix->d_rows, (u64)ix->shadow8_rows,
                       (u64)ix->n, (uint32_t)ix->dim, (uint32_t)ix->pitch, ix->d_rows8, pitch8, ix->d_scale8);
    if (hipGetLastError() != hipSuccess) return false;
    ix->shadow8_rows = ix->n;
  }
  return true;
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

// the several-queries-per-workgroup sample pass (scan8_sample4_kernel: 4 queries, 3 at QPL = 3)
template <int METRIC>
static scan8_fn pick_scan8_sample4(int L, int QPL) {
  switch (L * 10 + QPL) {
    case 81: return scan8_sample4_kernel<8, 1, METRIC>;
    case 82: return scan8_sample4_kernel<8, 2, METRIC>;
    case 83: return scan8_sample4_kernel<8, 3, METRIC>;
    case 162: return scan8_sample4_kernel<16, 2, METRIC>;
    case 163: return scan8_sample4_kernel<16, 3, METRIC>;
    case 322: return scan8_sample4_kernel<32, 2, METRIC>;
    case 323: return scan8_sample4_kernel<32, 3, METRIC>;
    case 642: return scan8_sample4_kernel<64, 2, METRIC>;
    case 643: return scan8_sample4_kernel<64, 3, METRIC>;
  }
  return nullptr;
}

// squared fp32 norms of the rows added since they were last computed (the L2 selection paths' |c|^2 term), and their
// running maximum / sum / count
static int ensure_row_norms(wdbx_index* ix) {
  int rc;
  if ((rc = grow((void**)&ix->d_cnmax, &ix->cnmax_bytes, 4 * sizeof(uint32_t)))) return rc;
  if (ix->cn_bytes < (size_t)ix->n * sizeof(float)) {
    if ((rc = grow((void**)&ix->d_cn, &ix->cn_bytes, (size_t)ix->cap * sizeof(float)))) return rc;
    ix->cn_rows = 0;
  }
  if (ix->cn_rows == 0) HIP_TRY(hipMemsetAsync(ix->d_cnmax, 0, 4 * sizeof(uint32_t), ix->stream));
  if (ix->cn_rows < ix->n) {
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((ix->n - ix->cn_rows + 3) / 4, 65536);
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3(blocks), dim3(256), 0, ix->stream, (const float*)ix->d_rows, (u64)ix->cn_rows,
                       (u64)ix->n, (uint32_t)ix->pitch, ix->d_cn, ix->d_cnmax);
    HIP_TRY(hipGetLastError());
    ix->cn_rows = ix->n;
    ix->gmax_valid = false;
  }
  return WDBX_OK;
}

// nq single queries, each with its own sample pass + full pass over the u8 shadow; thresholds, re-scoring and
// the final top-k run once per round of 32 queries.  Candidate counters at d_count[0 .. nq) (sized by the caller).
// candidates_only (k >= 200, one query per call): thresholds by radix select, no final top-k -- the re-scored
// candidates stay in d_cand[0 .. d_count[0]) for the caller's select chain.
static int enqueue_singles_u8(wdbx_index* ix, const float* d_queries, int nq, int k, int64_t* d_out_idx, float* d_out_score,
                              u64* keys_out, bool candidates_only) {
  const Scan8Shape* sh = scan8_shape((uint32_t)ix->dim);
  if (!sh) return fail(WDBX_E_STATE, "no u8 scan instance for dim %d", ix->dim);
  const bool l2 = ix->metric == WDBX_METRIC_L2;
  const uint32_t pitch8 = sh->pieces * 16;
  int rc;
  if (ix->rows8_bytes < ((size_t)ix->cap + TILE_PAD_ROWS) * pitch8 || ix->pitch8 != pitch8 || ix->shadow8_rows < ix->n)
    return fail(WDBX_E_STATE, "u8 shadow not prepared");
  if (l2 && (rc = ensure_row_norms(ix))) return rc;  // squared fp32 norms of the rows (the 2 w - |c|^2 form)
  // sampled 256-row tiles (4 groups of 64 rows each), as on the tile path
  const uint32_t tiles = (uint32_t)((ix->n + 255) / 256);
  const uint32_t div = ix->opt_gemm_sample_div > 0 ? (uint32_t)ix->opt_gemm_sample_div : std::min(32u, std::max(4u, 1024u / (uint32_t)k));
  uint32_t sample_tiles = std::max<uint32_t>(tiles / div, (8u * k + 3) / 4);
  sample_tiles = std::max<uint32_t>(1, std::min(sample_tiles, tiles));
  const uint32_t stride = tiles / sample_tiles, ngroups = 4 * sample_tiles;
  if (ngroups < (uint32_t)k) return fail(WDBX_E_STATE, "corpus too small for the selection scan at k=%d", k);
  const uint64_t expect = (uint64_t)k * (tiles / sample_tiles + 1);
  const uint32_t cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(4096, expect * 32), 1u << 22);
  constexpr int ROUND = 64;  // queries per round at most (the caller hands over `exchange_batch` = 32 queries at a time by default)
  if ((rc = grow((void**)&ix->d_halfmax, &ix->halfmax_bytes, (size_t)ROUND * ngroups * sizeof(u64)))) return rc;
  if ((rc = grow((void**)&ix->d_tau, &ix->tau_bytes, (size_t)GB_N * sizeof(float)))) return rc;
  if ((rc = grow((void**)&ix->d_cand, &ix->cand_bytes, (size_t)ROUND * cap * sizeof(u64)))) return rc;
  if (ix->count_bytes < ((size_t)nq + GB_N) * sizeof(uint32_t)) return fail(WDBX_E_STATE, "candidate counters not sized by the caller");
  ix->last_batch_nq = (uint32_t)nq;  // (the sample launch resets each query's candidate counter)
  ix->last_batch_cap = cap;
  scan8_fn f0 = l2 ? pick_scan8<0, WDBX_METRIC_L2>(sh->L, sh->QPL) : pick_scan8<0, WDBX_METRIC_COSINE>(sh->L, sh->QPL);
  scan8_fn f1 = l2 ? pick_scan8<1, WDBX_METRIC_L2>(sh->L, sh->QPL) : pick_scan8<1, WDBX_METRIC_COSINE>(sh->L, sh->QPL);
  if (!f0 || !f1) return fail(WDBX_E_STATE, "no u8 scan instance for %d lanes x %d loads", sh->L, sh->QPL);
  if (ix->opt_scan8_ablate == 1 && !l2 && sh->L == 8 && sh->QPL == 3) f1 = scan8_kernel<8, 3, WDBX_METRIC_COSINE, 1, 1>;  // timing only
  if (ix->opt_scan8_ablate == 2 && !l2 && sh->L == 8 && sh->QPL == 3) f1 = scan8_kernel<8, 3, WDBX_METRIC_COSINE, 1, 2>;  // all the arithmetic, no appends
  const uint32_t R = 64u / (uint32_t)sh->L;
  const uint32_t groups1 = (uint32_t)((ix->n + R - 1) / R);
  // workgroups per CU in the full pass's grid (2 fit at once: 8 waves per CU).  Shards of about 1 - 2.7 M rows at d = 384 -- the
  // strong-scaling shard sizes of 4 and 8 GPUs: a round's passes go out as one grid there -- take 6: with three times as many,
  // shorter workgroups per pass the tail of one query's pass overlaps the start of the next one's instead of the two resident
  // workgroups of every CU walking through prologue and tail in step (1.25 M rows 12 545 -> 12 782 q/s, 2.5 M 6 246 -> 6 512;
  // 400 k rows lose with more than 2, 5 M and 10 M rows do not care: profiles/r04/small_shard/summary.txt).  -1 = by size.
  const uint64_t shadow_bytes = (uint64_t)ix->n * pitch8;
  const int64_t wgs_auto = (nq > 1 && shadow_bytes >= (400ull << 20) && shadow_bytes <= (1ull << 30)) ? 6 : 2;  // (measured on rounds)
  const uint32_t wgs = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(ix->opt_scan8_wgs > 0 ? ix->opt_scan8_wgs : wgs_auto, 8));
  const uint32_t grid1 = std::min<uint32_t>((groups1 + 3) / 4, (uint32_t)ix->cu_count * wgs);
  const uint32_t grid0 = std::min<uint32_t>((ngroups + 3) / 4, (uint32_t)ix->cu_count * 4);
  const size_t pitch4 = ix->pitch / 4;

  for (int q0 = 0; q0 < nq; q0 += ROUND) {
    const int nv = std::min(ROUND, nq - q0);
    const float* qsrc = d_queries + (size_t)q0 * ix->pitch;
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
    a.sample_nt = (uint64_t)ngroups * 64 * (pitch8 + 4) > (48ull << 20);  // beyond ~48 MB the sample does not stay cached
    a.cap = cap;
    // phase 0, all queries of the round in one launch: maxima of the lower bounds over the sampled groups
    a.query = (const f4*)qsrc;
    a.halfmax = ix->d_halfmax;
    a.count = ix->d_count + q0;
    if ((rc = record(ix->sample_ev, ix->profile, ix->stream, true))) return rc;
    // a sample too large for the L2s (non-temporal loads) is read once per four (QPL = 3: three) queries of the round
    // (scan8_sample4_kernel).  Measured per round of 32 queries: 10 M x 384 top-10 (121 MB per query, served by the Infinity
    // Cache) 560 -> 363 us; 10 M x 768 L2 top-100 (772 MB per query, from HBM) about 3.9 -> 1.6 ms = 755 -> 797 q/s.
    scan8_fn f04 = ((a.sample_nt || ix->opt_scan8_sample4 == 2) && nv >= 4 && ix->opt_scan8_sample4)  // (2: also on a small sample, for tests)
                       ? (l2 ? pick_scan8_sample4<WDBX_METRIC_L2>(sh->L, sh->QPL) : pick_scan8_sample4<WDBX_METRIC_COSINE>(sh->L, sh->QPL))
                       : nullptr;
    ix->last_sample_qn = 1;
    if (f04) {
      const int qn = sh->QPL >= 3 ? 3 : 4;
      ix->last_sample_qn = qn;
      a.nq = (uint32_t)nv;
      hipLaunchKernelGGL(f04, dim3(grid0, (nv + qn - 1) / qn), dim3(256), 0, ix->stream, a);
    } else {
      hipLaunchKernelGGL(f0, dim3(grid0, nv), dim3(256), 0, ix->stream, a);
    }
    HIP_TRY(hipGetLastError());
    if ((rc = record(ix->sample_ev, ix->profile, ix->stream, false))) return rc;
    // a lone blocking query whose final top-k the host takes (3 dependent launches instead of 5): possible when its
    // candidate buffer fits the mapped host area; the threshold is then taken inside phase 1 when the sample is small
    const bool lone = ix->lone_keys_dev && nq == 1 && !candidates_only && !keys_out && cap <= ix->lone_cap_max;
    // (the in-kernel threshold also serves a lone query of the asynchronous entry points and of the shard group's local stage)
    const bool tau_in_kernel = nq == 1 && !candidates_only && k <= 128 && ngroups <= 1024 && ix->opt_lone_host_select;
    if (tau_in_kernel) {
      // (phase 1 takes the k-th largest sampled lower bound itself)
    } else if (candidates_only) {  // large k: the k-th largest sampled lower bound by radix select (the list kernels are insert-bound)
      if (nv != 1) return fail(WDBX_E_STATE, "large-k selection runs one query per call");
      if ((rc = grow((void**)&ix->d_state, &ix->state_bytes, sizeof(SelectState)))) return rc;
      const uint32_t hgrid = std::min<uint32_t>((ngroups + 255) / 256, (uint32_t)ix->cu_count * 4);
      hipLaunchKernelGGL(select_init_kernel, dim3(1), dim3(256), 0, ix->stream, ix->d_state, (uint32_t)k);
      for (int shift = 56; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(radix_hist_kernel, dim3(hgrid), dim3(256), 0, ix->stream, (const u64*)ix->d_halfmax, (u64)ngroups,
                           ix->d_state, shift, (const SelectSrc*)nullptr);
        hipLaunchKernelGGL(radix_pick_kernel, dim3(1), dim3(256), 0, ix->stream, ix->d_state, shift);
      }
      hipLaunchKernelGGL(select_kth_value_kernel, dim3(1), dim3(64), 0, ix->stream, (const SelectState*)ix->d_state, (uint32_t)k,
                         ix->d_tau);
      HIP_TRY(hipGetLastError());
    } else if (ngroups <= (uint32_t)KTH_R * 1024 && !ix->opt_lds_lists) {
      // a threshold from the k-th largest sampled lower bound (at most 2^-15 relative below it: kth_score_kernel)
      KthArgs ka = {ix->d_halfmax, (u64)ngroups, ngroups, k, ix->d_tau};
      if ((rc = record(ix->merge_ev, ix->profile, ix->stream, true))) return rc;
      hipLaunchKernelGGL(kth_score_kernel, dim3(nv), dim3(1024), 0, ix->stream, ka);
      HIP_TRY(hipGetLastError());
      if ((rc = record(ix->merge_ev, ix->profile, ix->stream, false))) return rc;
    } else {
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
    }
    // phase 1: every row whose upper bound reaches the threshold.  Every query makes its own pass over all rows; the
    // round's passes go out as ONE grid on small shards (grid.y = query: no launch gaps, ramps and tails overlap; option
    // scan8_per_query: -1 by size, 0 always one grid, 1 a launch per query).  One event pair around them: the profile
    // reports elapsed / passes.
    if ((rc = record(ix->gemm_ev, ix->profile, ix->stream, true))) return rc;
    a.query = (const f4*)qsrc;
    a.tau = ix->d_tau;
    a.cand = ix->d_cand;
    a.count = ix->d_count + q0;
    if (tau_in_kernel) {
      a.tau_keys = ix->d_halfmax;
      a.tau_n = ngroups;
      a.tau_k = k;
    }
    // (measured, d = 384: 1.25 M rows 84.5 vs 86.9 us per pass in one grid; 10 M rows 603 vs 591 us -- two passes streaming
    // different regions at once cost more there than the gaps between launches.  -1 = by size: one grid up to 1 GiB of shadow)
    const bool one_grid = ix->opt_scan8_per_query == 0 || (ix->opt_scan8_per_query < 0 && (uint64_t)ix->n * pitch8 <= (1ull << 30));
    if (one_grid && nv > 1) {
      hipLaunchKernelGGL(f1, dim3(grid1, nv), dim3(256), 0, ix->stream, a);
      HIP_TRY(hipGetLastError());
    } else {
      for (int i = 0; i < nv; ++i) {
        a.query = (const f4*)(qsrc + (size_t)i * ix->pitch);
        a.tau = ix->d_tau + i;
        a.cand = ix->d_cand + (size_t)i * cap;
        a.count = ix->d_count + q0 + i;
        hipLaunchKernelGGL(f1, dim3(grid1), dim3(256), 0, ix->stream, a);
        HIP_TRY(hipGetLastError());
      }
    }
    if ((rc = record(ix->gemm_ev, ix->profile, ix->stream, false, (uint32_t)nv))) return rc;
    // exact fp32 scores for the candidates, from the fp32 rows
    hipLaunchKernelGGL(l2 ? rescore_kernel<WDBX_METRIC_L2> : rescore_kernel<WDBX_METRIC_COSINE>, dim3(256, nv), dim3(256), 0,
                       ix->stream, (const f4*)ix->d_rows, (uint32_t)pitch4, (const f4*)qsrc, ix->d_cand,
                       (const uint32_t*)(ix->d_count + q0), cap, lone ? ix->lone_keys_dev : (u64*)nullptr,
                       lone ? ix->lone_count_dev : (uint32_t*)nullptr);
    HIP_TRY(hipGetLastError());
    if (candidates_only) continue;  // the caller's select chain ranks them
    if (lone) {  // the host ranks the keys after its synchronisation
      ix->lone_used = true;
      continue;
    }
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
    f.over_out = ix->defer_flag_dev ? ix->defer_flag_dev + q0 : nullptr;
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
enum { GEMM_FP32 = 0, GEMM_BF16 = 1, GEMM_BF16_SHADOW = 2, GEMM_I8 = 3 };
static inline int gemm_family(const wdbx_index* ix) {
  return ix->opt_gemm_bf16 <= 0 ? GEMM_FP32 : ix->opt_gemm_bf16 == 1 ? GEMM_BF16 : GEMM_BF16_SHADOW;
}

// ---- int8 tiles (option gemm_bf16 = 3, the default): inner product / cosine, rows whose i8 image is at most 1536 bytes
// (the query block stays resident in LDS: 64 * CT queries x pitch8g bytes <= 96 KiB), and not shorter than 0.8 of ...
static inline uint32_t i8g_pitch(const wdbx_index* ix) { return (uint32_t)((ix->dim + 127) / 128 * 128); }
static inline int i8g_max_ct(const wdbx_index* ix) {
  const uint32_t p = i8g_pitch(ix);
  return p * 256u <= (uint32_t)G8_LDS_B_MAX ? 4 : p * 128u <= (uint32_t)G8_LDS_B_MAX ? 2 : p * 64u <= (uint32_t)G8_LDS_B_MAX ? 1 : 0;
}
static bool i8_tiles_eligible(const wdbx_index* ix) {
  if (ix->opt_gemm_bf16 < 3 || ix->active_mask) return false;
  if (ix->metric == WDBX_METRIC_L2 && (!ix->opt_gemm_l2 || !ix->opt_gemm_l2_i8)) return false;
  if (i8g_max_ct(ix) == 0) return false;
  // short rows: the padded i8 image (128-byte multiples) must be clearly smaller than the fp32 row
  return (uint64_t)i8g_pitch(ix) * 10 <= (uint64_t)ix->pitch * 4 * 8;
}

// Allocates / refreshes shadow copy G for the rows added since the last batch (whole 64-row groups: the group that was
// only partly filled is re-quantised, its scale may have changed).  false = no room on the device (remembered per
// capacity): the caller stays on the bf16 tiles.
static bool prepare_i8g_shadow(wdbx_index* ix) {
  const uint32_t pitch8 = i8g_pitch(ix);
  const size_t need = (((size_t)ix->cap + G8_ROWS - 1) / G8_ROWS + 1) * G8_ROWS * pitch8;  // whole tiles (fragment order) + one of slack
  const size_t need_g = (((size_t)ix->cap + TILE_PAD_ROWS + 63) / 64 + 1) * sizeof(f4);
  if (ix->rows8g_bytes < need || ix->groups8_bytes < need_g || ix->pitch8g != pitch8) {
    if (ix->i8g_no_room_cap == ix->cap) return false;
    if (ix->d_rows8g) (void)hipFree(ix->d_rows8g);
    if (ix->d_groups8) (void)hipFree(ix->d_groups8);
    if (ix->d_gbad8) (void)hipFree(ix->d_gbad8);
    ix->d_rows8g = nullptr;
    ix->d_groups8 = nullptr;
    ix->d_gbad8 = nullptr;
    ix->rows8g_bytes = ix->groups8_bytes = 0;
    ix->shadowg_rows = 0;
    ix->shadowg_tail_n = ~0ull;
    if (hipMalloc((void**)&ix->d_rows8g, need) != hipSuccess || hipMalloc((void**)&ix->d_groups8, need_g) != hipSuccess ||
        hipMalloc((void**)&ix->d_gbad8, need_g / 2) != hipSuccess) {
      (void)hipGetLastError();
      if (ix->d_rows8g) (void)hipFree(ix->d_rows8g);
      if (ix->d_groups8) (void)hipFree(ix->d_groups8);
      ix->d_rows8g = nullptr;
      ix->d_groups8 = nullptr;
      ix->i8g_no_room_cap = ix->cap;
      return false;
    }
    // (the table entries of the pad groups are read by the last tile's waves and must not be garbage that traps: zero;
    // pad groups hold no row: all their bad-row bits set)
    if (hipMemsetAsync(ix->d_groups8, 0, need_g, ix->stream) != hipSuccess) return false;
    if (hipMemsetAsync(ix->d_gbad8, 0xFF, need_g / 2, ix->stream) != hipSuccess) return false;
    ix->rows8g_bytes = need;
    ix->groups8_bytes = need_g;
    ix->pitch8g = pitch8;
  }
  // Groups are (re)quantised up to the END OF THE LAST 256-ROW TILE of the current row count, not just up to the last row's
  // group: the tile kernels read whole tiles, and PHASE 0 trusts the bad-row bits alone to keep rows past the end out of the
  // threshold sample.  After wdbx_index_clear() + fewer rows, or a compaction that dropped live tail rows, the groups between
  // ceil(n / 64) and the tile's end would otherwise keep the OLD corpus's bytes, table entries and "good" bits, their lower
  // bounds could enter the k-th largest, and a threshold above the true k-th score silently drops true neighbours.
  if (ix->shadowg_rows < ix->n || ix->shadowg_tail_n != ix->n) {
    const u64 g0 = std::min<u64>(ix->shadowg_rows, ix->n) / 64;
    const u64 g1 = (ix->n + G8_ROWS - 1) / G8_ROWS * (G8_ROWS / 64);
    if (g1 > g0) {
      hipLaunchKernelGGL(rows_to_i8g_kernel, dim3((uint32_t)std::min<u64>(g1 - g0, 1u << 20)), dim3(256), 0, ix->stream,
                         (const float*)ix->d_rows, g0, g1, (u64)ix->n, (uint32_t)ix->dim, (uint32_t)ix->pitch, ix->d_rows8g, pitch8,
                         ix->d_groups8, ix->d_gbad8);
      if (hipGetLastError() != hipSuccess) return false;
    }
    ix->shadowg_rows = ix->n;
    ix->shadowg_tail_n = ix->n;
    ix->gref_valid = false;
  }
  return true;
}
static inline uint32_t gemm_tile_rows(int family) { return family == GEMM_FP32 ? GB_M : GW_M; }

template <int PHASE, int CT, int METRIC, bool GROUPB>
static void (*pick_gemm_kernel_gb(int family, bool ktail))(GemmArgs) {
  if constexpr (CT >= 2) {
    // (L2 with per-group bounds on the shadow tiles exists for query blocks of 128 only: the 256-wide form of that one
    // combination does not fit the register file without spilling; enqueue_search_gemm never asks for it)
    if constexpr (CT == 4 && GROUPB && METRIC == WDBX_METRIC_L2) {
      if (family == GEMM_BF16_SHADOW) return nullptr;
    } else if (family == GEMM_BF16_SHADOW) {
      return gemm_bf16w8_kernel<PHASE, false, CT, METRIC, true, GROUPB>;
    }
    if (family == GEMM_BF16)
      return ktail ? gemm_bf16w8_kernel<PHASE, true, CT, METRIC, false, GROUPB> : gemm_bf16w8_kernel<PHASE, false, CT, METRIC, false, GROUPB>;
  }
  if constexpr (GROUPB && METRIC == WDBX_METRIC_COSINE) return nullptr;  // exact fp32 inner-product tiles: nothing to bound
  else return ktail ? gemm_topk_kernel<PHASE, true, CT, METRIC, GROUPB> : gemm_topk_kernel<PHASE, false, CT, METRIC, GROUPB>;
}

template <int PHASE, int CT, int METRIC>
static void (*pick_gemm_kernel(int family, bool ktail, bool groupb))(GemmArgs) {
  return groupb ? pick_gemm_kernel_gb<PHASE, CT, METRIC, true>(family, ktail) : pick_gemm_kernel_gb<PHASE, CT, METRIC, false>(family, ktail);
}

template <int PHASE, int CT>
static int launch_gemm_ct(wdbx_index* ix, const GemmArgs& g, int family) {
  if (family != GEMM_FP32 && CT < 2) return fail(WDBX_E_STATE, "bf16 tiles need a query block of at least 128");
  const int bk = family == GEMM_BF16_SHADOW ? 64 : 32;  // elements per LDS chunk
  const uint32_t tile_rows = gemm_tile_rows(family);
  // (+ 64 floats per wave behind the tiles: the L2 epilogue's row norms)
  const size_t lds = (family == GEMM_FP32 ? (size_t)(2 * GB_M + 2 * 64 * CT) * 36 * sizeof(float) + 4 * 64 * sizeof(float)
                                          : (size_t)(2 * tile_rows + 2 * 64 * CT) * (bk * 2 + 16) + 8 * 64 * sizeof(float));
  // K tail: the fp32 tiles step 8 quads at a time, the bf16 tiles a pair of chunks of 8 quads (never on the
  // shadow, which is padded)
  const bool ktail = family != GEMM_BF16_SHADOW && (g.pitch4 % (family == GEMM_FP32 ? 8u : 16u)) != 0;
  const bool l2 = ix->metric == WDBX_METRIC_L2;
  void (*fn)(GemmArgs) = l2 ? pick_gemm_kernel<PHASE, CT, WDBX_METRIC_L2>(family, ktail, g.gmax != nullptr)
                            : pick_gemm_kernel<PHASE, CT, WDBX_METRIC_COSINE>(family, ktail, g.gmax != nullptr);
  if (!fn) return fail(WDBX_E_STATE, "no tile kernel for this family with per-group bounds");
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
  if (!d_queries || (!keys_out && (!d_out_idx || !d_out_score))) return fail(WDBX_E_INVALID, "null device buffer");
  if (ix->n >= 0xFFFFFF00ull) return fail(WDBX_E_INVALID, "shard holds too many rows for 32-bit row keys");
  int rc;
  ix->last_batch_repaired = false;
  if (count_slot < 0 && i8_tiles_eligible(ix) && prepare_i8g_shadow(ix))
    return enqueue_search_gemm8(ix, d_queries, nq, k, d_out_idx, d_out_score, mode, keys_out);
  int family = gemm_family(ix);
  // short rows: the padded shadow would be no smaller than the fp32 rows, so the tiles read those
  if (family == GEMM_BF16_SHADOW && ((uint64_t)ix->pitch + 127) / 128 * 128 >= 2 * (uint64_t)ix->pitch) family = GEMM_BF16;
  if (family == GEMM_BF16_SHADOW) {  // the bf16 shadow copy of the rows added since the last batch
    const uint32_t pitch16 = (uint32_t)((ix->pitch + 127) / 128 * 128);  // whole pairs of 64-element chunks
    const size_t need = ((size_t)ix->cap + TILE_PAD_ROWS) * pitch16 * 2;
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
    if ((rc = ensure_row_norms(ix))) return rc;
    // How the selection error is bounded.  Norms all alike (the reference normalises its rows): ONE bound from the
    // largest norm, subtracted from the thresholds (tau_margin_kernel) -- nothing in the tile epilogue.  Norms that
    // vary a lot (largest > 1.5 x mean: unnormalised data, outlier rows): a bound per 64-row group from that group's
    // largest norm, applied in the epilogue (+10 % kernel time), so an outlier only loosens its own group and the
    // batch does not degenerate into per-query repairs.  Decided when the norms change (one 8-byte read-back).
    if (!ix->gmax_valid) {
      if (ix->cn_stats_dirty) {  // rows were overwritten in place: rebuild maximum / sum / count from the cached norms
        HIP_TRY(hipMemsetAsync(ix->d_cnmax, 0, 4 * sizeof(uint32_t), ix->stream));
        hipLaunchKernelGGL(cn_stats_kernel, dim3((uint32_t)std::min<uint64_t>((ix->n + 255) / 256, 2048)), dim3(256), 0, ix->stream,
                           (const float*)ix->d_cn, (u64)ix->n, ix->d_cnmax);
        HIP_TRY(hipGetLastError());
        ix->cn_stats_dirty = false;
      }
      uint32_t words[3] = {0, 0, 0};
      HIP_TRY(hipMemcpyAsync(words, ix->d_cnmax, sizeof words, hipMemcpyDeviceToHost, ix->stream));
      HIP_TRY(hipStreamSynchronize(ix->stream));
      float cmax, csum;
      memcpy(&cmax, &words[0], 4);
      memcpy(&csum, &words[1], 4);
      const float mean = words[2] ? csum / (float)words[2] : 0.f;  // over the rows with a finite norm (removed rows are NaN)
      // an infinite norm (a row with an infinite or huge element) always takes the per-group instances: their epilogue
      // sends such a group's rows to the exact pass whatever the selection scores say (kernels_tiles.h)
      ix->group_bounds = ix->opt_group_bounds == 1 || !(cmax < INFINITY) || (ix->opt_group_bounds != 0 && !(cmax <= 1.5f * mean));
      if (ix->group_bounds) {
        if ((rc = grow((void**)&ix->d_gmax, &ix->gmax_bytes, ((size_t)ix->cap + 63) / 64 * sizeof(float)))) return rc;
        const uint32_t gblocks = (uint32_t)std::min<uint64_t>(((ix->n + 63) / 64 + 255) / 256, 4096);
        hipLaunchKernelGGL(group_max_kernel, dim3(gblocks), dim3(256), 0, ix->stream, (const float*)ix->d_cn, (u64)ix->n, ix->d_gmax);
        HIP_TRY(hipGetLastError());
      }
      ix->gmax_valid = true;
    }
    if ((rc = grow((void**)&ix->d_qn, &ix->qn_bytes, (size_t)2 * GB_N * sizeof(float)))) return rc;
  }
  const bool group_bounds = inexact && ix->group_bounds;
  ix->last_batch_nq = (uint32_t)nq;
  ix->last_batch_cap = cap;
  // rounding-error coefficients of the selection scores (kernels_tiles.h, GemmArgs): fp32 chains of n terms carry
  // gamma = n u / (1 - n u); bf16 roundings of both operands add 2^-7 (+ 2^-16) per product
  const float nu = (float)(ix->pitch + 18) * 5.9604645e-08f;
  const float gamma = 1.02f * nu / (1.0f - nu);
  const float sel_eps = (gamma + (bf16 ? 1.05f * 0.0078125f : 0.0f)) * 1.0102f, sel_gam = gamma * 1.01f;
  const float sel_floor = bf16 ? 7.5e-37f * sqrtf((float)ix->pitch) : 0.0f;  // 2^-120 sqrt(d): see GemmArgs::floor_abs

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
    if (family == GEMM_BF16_SHADOW && l2 && group_bounds && ct > 2) ct = 2;  // see pick_gemm_kernel_gb
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
    if (group_bounds) {
      const int nqn = per_query ? nv : gbn;
      hipLaunchKernelGGL(query_norm_kernel, dim3(nqn), dim3(64), 0, ix->stream, qsrc, (uint32_t)ix->pitch, nv, nqn, ix->d_qn);
      HIP_TRY(hipGetLastError());
      g.gmax = ix->d_gmax;
      g.qn = ix->d_qn;
      g.eps = sel_eps;
      g.gam = sel_gam;
      g.floor_abs = sel_floor;
    }
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
      if (group_bounds) g.qn = ix->d_qn + (per_query ? i : 0);
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
    if (inexact && !group_bounds) {  // one global rounding-error margin below the sampled threshold
      hipLaunchKernelGGL(tau_margin_kernel, dim3(nv), dim3(64), 0, ix->stream, ix->d_tau, qsrc, (uint32_t)ix->pitch, nv,
                         (const uint32_t*)ix->d_cnmax, ix->metric, (int)bf16, sel_floor);
      HIP_TRY(hipGetLastError());
    }  // (with per-group bounds the k-th largest reported LOWER bound is already a valid threshold)
    g.num_tiles = tiles;
    g.tile_stride = 1;
    g.halfmax = nullptr;
    g.cap = cap;
    for (int i = 0; i < passes; ++i) {  // phase 1: every score above the threshold becomes a candidate
      g.qb16 = bf16 ? (const void*)((const __bf16*)ix->d_qb16 + (size_t)i * qb_block) : nullptr;
      g.tau = ix->d_tau + i;
      g.cand = ix->d_cand + (size_t)i * cap;
      g.count = d_count + q0 + i;
      if (group_bounds) g.qn = ix->d_qn + (per_query ? i : 0);
      if ((rc = launch_gemm<1>(ix, g, ct, family))) return rc;
    }
    if (inexact) {  // exact fp32 scores for the selected candidates
      hipLaunchKernelGGL(l2 ? rescore_kernel<WDBX_METRIC_L2> : rescore_kernel<WDBX_METRIC_COSINE>, dim3(64, nv), dim3(256), 0,
                         ix->stream, (const f4*)ix->d_rows, (uint32_t)pitch4, (const f4*)qsrc, ix->d_cand,
                         (const uint32_t*)(d_count + q0), cap, (u64*)nullptr, (uint32_t*)nullptr);
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
    if (!per_query) {  // (the single-query caller queues its own repair launches)
      bool done = false;
      if ((rc = enqueue_batch_repair(ix, qsrc, nv, k, d_count + q0, cap, f.out_idx, f.out_score, f.out_keys, &done))) return rc;
      if (q0 == 0) ix->last_batch_repaired = done;
      else ix->last_batch_repaired = ix->last_batch_repaired && done;
    }
    if (sharded && (rc = exchange_and_merge(ix, nv, k, d_out_idx + (size_t)q0 * k, d_out_score + (size_t)q0 * k))) return rc;
    q0 += nv;
  }
  return WDBX_OK;
}

// ---- batched queries on the int8 tiles (kernels_tiles8.h) ----------------------------------------
// The instance for a block of 32 * CT8 queries: rows of 384 and 768 bytes (d <= 768 in steps that cover the reference's
// embedding sizes 384 and 768) get the compile-time pitch, everything else the run-time form.  Option gemm8_variant
// (experiments, tools/probes/c4_i8_ab.py): 1 = run-time pitch everywhere, 2 = row stream with the default cache policy, 3 = SIMD
// partners half a tile apart, 4 = two k-steps in flight instead of three, 5 = the tile epilogue inside the next tile's first
// k-step, 6 = the epilogue as one block + one branch, 7 = a query-fragment window of 4, 12 / 13 / 14 = prefilter epilogue with a branch per
// column group / the round-2 forms / prefilter in one block = the default (correct answers all: profiles/r03/c4_i8/); 8, 10, 11 = timing-only ablations (no epilogue; and no row
// stream / no query-fragment reads): wrong answers, never set outside the probe.
// L2 instances: the compile-time pitch for rows of 768 bytes (BASELINE config 3) and 384, the run-time form for the rest
template <int PHASE, int CT8>
static void (*pick_gemm8_l2(uint32_t pitch8, int ring))(Gemm8Args) {
  constexpr int M = WDBX_METRIC_L2;
  if constexpr (CT8 <= 4) {
    if (pitch8 == 384) return gemm_i8_kernel<PHASE, CT8, 6, 384, 0, M>;
    if (pitch8 == 768) return gemm_i8_kernel<PHASE, CT8, 6, 768, 0, M>;
    if (ring == 6) return gemm_i8_kernel<PHASE, CT8, 6, 0, 0, M>;
    if (ring == 4) return gemm_i8_kernel<PHASE, CT8, 4, 0, 0, M>;
    return gemm_i8_kernel<PHASE, CT8, 2, 0, 0, M>;
  }
  return nullptr;  // (L2 runs on query blocks of at most 128: its epilogue keeps 16 more values per lane)
}

template <int PHASE, int CT8>
static void (*pick_gemm8(uint32_t pitch8, int ring, int variant))(Gemm8Args) {
  // the full pass (PHASE 1) of every inner-product instance carries the prefilter epilogue (VAR bits 8 + 6) since round 3;
  // gemm8_variant = 13: the round-2 forms, for A/B
  constexpr int PV = PHASE == 1 ? 256 + 64 : 0;
  if (variant != 1) {
    if (pitch8 == 384) {
      if constexpr (CT8 == 8 && PHASE == 1) {  // the tile epilogue as one block + one branch (VAR bit 6)
        // (15 .. 18 = round 4's 4 x 2 wave split, measured 18-47 % slower and removed: profiles/r04/c4_split/)
        if (variant == 6) return gemm_i8_kernel<1, 8, 3, 384, 64>;
        // (21 .. 25 = round 4's four waves of 64 rows x 256 queries, one per SIMD with 256 + 256 registers: never ran clean
        // through the register allocator, and VALU ops cannot read the accumulator file -- profiles/r04/c4_wide/)
        // 27 .. 29: the row stream through buffer loads (VAR bit 9), which frees the registers a ring of 6 k-steps needs
        if (variant == 27) return gemm_i8_kernel<1, 8, 6, 384, 512 + 128>;       // ring 6, window 4, branch per column group
        if (variant == 28) return gemm_i8_kernel<1, 8, 6, 384, 512 + 128 + 64>;  // ring 6, window 4, one-block epilogue
        if (variant == 29) return gemm_i8_kernel<1, 8, 3, 384, 512 + 256 + 64>;  // the product form on buffer loads
        // timing only (wrong answers): 30 = matrix ops alone; 31 / 32 = no epilogue / product form over an L2-resident row stream
        if (variant == 30) return gemm_i8_kernel<1, 8, 3, 384, 4 + 8 + 16>;
        if (variant == 31) return gemm_i8_kernel<1, 8, 3, 384, 4 + 1024>;
        if (variant == 32) return gemm_i8_kernel<1, 8, 3, 384, 256 + 64 + 1024>;
        // default since round 3: the prefilter epilogue (VAR bit 8; -2 % against the round-2 form, identical candidates);
        // 13 = the round-2 product form, for A/B
        if (variant == 0 || variant == 14) return gemm_i8_kernel<1, 8, 3, 384, 256 + 64>;  // ... its 16 tests in one block, one branch
        if (variant == 12) return gemm_i8_kernel<1, 8, 3, 384, 256>;                       // ... with a branch per column group
        if (variant == 13) return gemm_i8_kernel<1, 8, 3, 384>;
        if (variant == 7) return gemm_i8_kernel<1, 8, 3, 384, 128>;   // query-fragment window of 4 instead of 8
        // (9 = ... and the 16 freed registers as two more k-steps of row ring: 0.891 vs 0.844 ms, 16 B of scratch; measured in
        // profiles/r03/c4_i8/ab_0_7_9.json and removed)
      }
      if constexpr (CT8 == 8)
        return variant == 2   ? gemm_i8_kernel<PHASE, 8, 3, 384, 1>
               : variant == 3 ? gemm_i8_kernel<PHASE, 8, 3, 384, 2>
               : variant == 4 ? gemm_i8_kernel<PHASE, 8, 2, 384>
               : variant == 5 ? gemm_i8_kernel<PHASE, 8, 3, 384, 32>
               : variant == 8 ? gemm_i8_kernel<PHASE, 8, 3, 384, 4>
               : variant == 10 ? gemm_i8_kernel<PHASE, 8, 3, 384, 12>
               : variant == 11 ? gemm_i8_kernel<PHASE, 8, 3, 384, 20>
                              : gemm_i8_kernel<PHASE, 8, 3, 384>;
      else return variant == 13 ? gemm_i8_kernel<PHASE, CT8, 6, 384> : gemm_i8_kernel<PHASE, CT8, 6, 384, PV>;
    }
    if constexpr (CT8 <= 4)
      if (pitch8 == 768) return variant == 13 ? gemm_i8_kernel<PHASE, CT8, 6, 768> : gemm_i8_kernel<PHASE, CT8, 6, 768, PV>;
  }
  const bool r2 = variant == 13 || variant == 1;  // (1 = run-time pitch everywhere, in its round-2 form)
  if constexpr (CT8 < 8) {  // (256-query blocks with run-time addressing have registers for 2 k-steps in flight, not more)
    if (ring == 6) return r2 ? gemm_i8_kernel<PHASE, CT8, 6> : gemm_i8_kernel<PHASE, CT8, 6, 0, PV>;
    if (ring == 4) return r2 ? gemm_i8_kernel<PHASE, CT8, 4> : gemm_i8_kernel<PHASE, CT8, 4, 0, PV>;
  }
  return r2 ? gemm_i8_kernel<PHASE, CT8, 2> : gemm_i8_kernel<PHASE, CT8, 2, 0, PV>;
}

template <int PHASE>
static int launch_gemm8(wdbx_index* ix, const Gemm8Args& g, int ct) {
  // k-steps (64 bytes of a row: two A fragments) in flight per wave: a divisor of the row's k-steps (pitch8 is a multiple of
  // 128).  256-query blocks leave room for 2 or 3 (128 accumulator + 32 query-fragment registers), narrower blocks for 6.
  const uint32_t steps = g.pitch8 / 64;
  const int ring = ct == 4 ? 2 : (steps % 6 == 0 ? 6 : steps % 4 == 0 ? 4 : 2);
  const int var = (int)ix->opt_gemm8_variant;
  void (*fn)(Gemm8Args) = ix->metric == WDBX_METRIC_L2
                              ? (ct == 2 ? pick_gemm8_l2<PHASE, 4>(g.pitch8, ring) : ct == 1 ? pick_gemm8_l2<PHASE, 2>(g.pitch8, ring) : nullptr)
                          : ct == 4 ? pick_gemm8<PHASE, 8>(g.pitch8, ring, var)
                          : ct == 2 ? pick_gemm8<PHASE, 4>(g.pitch8, ring, var)
                                    : pick_gemm8<PHASE, 2>(g.pitch8, ring, var);
  if (!fn) return fail(WDBX_E_STATE, "no int8 tile instance for this query block");
  const size_t lds = (size_t)64 * ct * g.pitch8 + (size_t)64 * ct * sizeof(f4);  // the query block + its parameters
  HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const uint32_t grid = std::min<uint32_t>(g.num_tiles, (uint32_t)ix->cu_count);  // one 8-wave workgroup per CU
  int rc = record(ix->gemm_ev, ix->profile, ix->stream, true);
  if (rc) return rc;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(512), lds, ix->stream, g);
  HIP_TRY(hipGetLastError());
  return record(ix->gemm_ev, ix->profile, ix->stream, false);
}

// nq (any number) queries in blocks of up to 256 through the int8 tiles.  Same contract as enqueue_search_gemm: per-query
// candidate counters in d_count[q] (a count above the capacity = that query must be re-run on the scan path), results are
// the exact fp32 ranking of the kept rows (rescore_kernel + merge_kernel).
static int enqueue_search_gemm8(wdbx_index* ix, const float* d_queries, int nq, int k, int64_t* d_out_idx, float* d_out_score,
                                int mode, u64* keys_out) {
  const bool sharded = mode == SEARCH_SHARDED;
  int rc;
  ix->last_gemm_mode = GEMM_I8;
  const uint32_t pitch8 = ix->pitch8g;
  const uint32_t tiles = (uint32_t)((ix->n + G8_ROWS - 1) / G8_ROWS), rw = G8_ROWS / 32;  // a lower bound per 32-row block
  const uint32_t div = ix->opt_gemm_sample_div > 0 ? (uint32_t)ix->opt_gemm_sample_div : std::min(32u, std::max(4u, 1024u / (uint32_t)k));
  uint32_t sample_tiles = std::max<uint32_t>(tiles / div, (8u * k + rw - 1) / rw);
  sample_tiles = std::max<uint32_t>(1, std::min(sample_tiles, tiles));
  const uint32_t stride = tiles / sample_tiles;
  if (rw * sample_tiles < (uint32_t)k) return fail(WDBX_E_STATE, "corpus too small for the batched path at k=%d", k);
  const uint64_t expect = (uint64_t)k * (tiles / sample_tiles + 1);
  const uint32_t cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(4096, expect * 32), 1u << 22);
  const size_t pitch4 = ix->pitch / 4;
  if ((rc = grow((void**)&ix->d_halfmax, &ix->halfmax_bytes, (size_t)GB_N * rw * sample_tiles * sizeof(u64)))) return rc;
  if ((rc = grow((void**)&ix->d_tau, &ix->tau_bytes, (size_t)GB_N * sizeof(float)))) return rc;
  if ((rc = grow((void**)&ix->d_cand, &ix->cand_bytes, (size_t)GB_N * cap * sizeof(u64)))) return rc;
  // (+ 1 word behind the counters: "a wave's pair list overflowed" -- see scatter_pairs_kernel)
  if ((rc = grow((void**)&ix->d_count, &ix->count_bytes, ((size_t)nq + GB_N + 1) * sizeof(uint32_t)))) return rc;
  uint32_t* const d_lost = ix->d_count + nq + GB_N;
  if ((rc = grow((void**)&ix->d_qb8, &ix->qb8_bytes, (size_t)GB_N * pitch8))) return rc;
  if ((rc = grow((void**)&ix->d_qpar, &ix->qpar_bytes, (size_t)GB_N * sizeof(f4)))) return rc;
  // per-wave candidate pair lists of the full pass: room for 4x the expected share of a wave, at least 2048 pairs
  const uint32_t nwaves = std::min<uint32_t>(tiles, (uint32_t)ix->cu_count) * 8;
  // (16384: two tiles' worth of "every row of the wave's group is a candidate for every query", so a few outlier groups
  // do not turn the call into per-query repairs)
  const uint32_t pair_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(16384, (uint64_t)GB_N * expect * 16 / nwaves), 1u << 16);
  if ((rc = grow((void**)&ix->d_pairs, &ix->pairs_bytes, (size_t)nwaves * pair_cap * sizeof(u64)))) return rc;
  if ((rc = grow((void**)&ix->d_pair_count, &ix->pair_count_bytes, (size_t)nwaves * sizeof(uint32_t)))) return rc;
  if (sharded && (rc = grow((void**)&ix->d_local_keys, &ix->local_keys_bytes, (size_t)GB_N * k * sizeof(u64)))) return rc;
  // (the counters of every block, tau and the lost flag are initialised by the block's queries_to_i8_kernel: no memsets)
  ix->last_batch_nq = (uint32_t)nq;
  ix->last_batch_cap = cap;
  const bool l2 = ix->metric == WDBX_METRIC_L2;
  if (l2 && (rc = ensure_row_norms(ix))) return rc;
  if (!ix->gref_valid) {  // the ordinary-group bounds of the prefilter epilogue (two passes over the group table, ~10 us)
    if (!ix->d_gref8) HIP_TRY(hipMalloc((void**)&ix->d_gref8, 8 * sizeof(float)));
    HIP_TRY(hipMemsetAsync(ix->d_gref8, 0, 8 * sizeof(float), ix->stream));
    const u64 ngroups = (ix->n + 63) / 64;
    const uint32_t blocks = (uint32_t)std::min<u64>((ngroups + 255) / 256, 1024);
    for (int pass = 0; pass < 2; ++pass)
      hipLaunchKernelGGL(group_ref_kernel, dim3(blocks), dim3(256), 0, ix->stream, (const f4*)ix->d_groups8, ngroups, ix->d_gref8,
                         (uint32_t*)(ix->d_gref8 + 4), pass);
    HIP_TRY(hipGetLastError());
    ix->gref_valid = true;
  }
  const int max_ct = l2 ? std::min(2, i8g_max_ct(ix)) : i8g_max_ct(ix);  // (L2: query blocks of at most 128)
  for (int q0 = 0; q0 < nq;) {
    const int rem = nq - q0;
    int ct = (ix->opt_gemm_ct == 1 || ix->opt_gemm_ct == 2 || ix->opt_gemm_ct == 4) ? (int)ix->opt_gemm_ct : rem > 128 ? 4 : rem > 64 ? 2 : 1;
    ct = std::min(ct, max_ct);
    const int gbn = 64 * ct, nv = std::min(gbn, rem);
    const float* qsrc = d_queries + (size_t)q0 * ix->pitch;
    hipLaunchKernelGGL(queries_to_i8_kernel, dim3((uint32_t)(gbn + 3) / 4), dim3(256), 0, ix->stream, qsrc, (uint32_t)ix->dim,
                       (uint32_t)ix->pitch, (uint32_t)nv, ix->d_qb8, pitch8, (uint32_t)gbn, ix->d_qpar, ix->d_tau, ix->d_count + q0, d_lost);
    HIP_TRY(hipGetLastError());
    Gemm8Args g = {};
    g.rows8 = ix->d_rows8g;
    g.groups = ix->d_groups8;
    g.cn = ix->d_cn;
    g.gbad = ix->d_gbad8;
    g.gref = ix->d_gref8;
    g.qb8 = ix->d_qb8;
    g.qpar = ix->d_qpar;
    g.n_rows = (uint32_t)ix->n;
    g.pitch8 = pitch8;
    g.num_tiles = sample_tiles;
    g.tile_stride = stride;
    g.halfmax = ix->d_halfmax;
    if ((rc = launch_gemm8<0>(ix, g, ct))) return rc;
    // the k-th largest of the groups' LOWER bounds: a valid threshold by itself (no margin)
    if (rw * sample_tiles <= KTH_R * 1024 && !ix->opt_lds_lists) {
      KthArgs ka = {ix->d_halfmax, (u64)rw * sample_tiles, rw * sample_tiles, k, ix->d_tau};
      if ((rc = record(ix->merge_ev, ix->profile, ix->stream, true))) return rc;
      hipLaunchKernelGGL(kth_score_kernel, dim3(nv), dim3(1024), 0, ix->stream, ka);
      HIP_TRY(hipGetLastError());
      if ((rc = record(ix->merge_ev, ix->profile, ix->stream, false))) return rc;
    } else {
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
    }
    g.num_tiles = tiles;
    g.tile_stride = 1;
    g.halfmax = nullptr;
    g.tau = ix->d_tau;
    g.pairs = ix->d_pairs;
    g.pair_count = ix->d_pair_count;
    g.pair_cap = pair_cap;
    if ((rc = launch_gemm8<1>(ix, g, ct))) return rc;
    hipLaunchKernelGGL(scatter_pairs_kernel, dim3((nwaves + SCATTER_LISTS - 1) / SCATTER_LISTS), dim3(1024), 0, ix->stream,
                       (const u64*)ix->d_pairs, (const uint32_t*)ix->d_pair_count, nwaves, pair_cap, ix->d_cand, ix->d_count + q0, cap,
                       d_lost);
    HIP_TRY(hipGetLastError());
    if (ix->opt_gemm8_refine) {
      // second selection stage (refine_pairs_kernel): the pairs' own integer dot products turn into per-row bounds, and the
      // rows that cannot be among the k best are dropped before the exact pass gathers any fp32 row
      RefineArgs r = {};
      r.cand = ix->d_cand;
      r.count = ix->d_count + q0;
      r.cap = cap;
      r.groups = ix->d_groups8;
      r.gbad = ix->d_gbad8;
      r.cn = ix->d_cn;
      r.qpar = ix->d_qpar;
      r.k = k;
      const bool reg = k <= 128 && !ix->opt_lds_lists;
      void (*fn)(RefineArgs) = l2 ? (reg ? refine_pairs_kernel<WDBX_METRIC_L2, true> : refine_pairs_kernel<WDBX_METRIC_L2, false>)
                                  : (reg ? refine_pairs_kernel<WDBX_METRIC_COSINE, true> : refine_pairs_kernel<WDBX_METRIC_COSINE, false>);
      r.n_lds = std::min<uint32_t>(cap, REFINE_R * 1024);
      hipLaunchKernelGGL(fn, dim3(nv), dim3(1024), std::max((size_t)r.n_lds * 4, (size_t)17 * k * sizeof(u64)), ix->stream, r);
      HIP_TRY(hipGetLastError());
    }
    // exact fp32 scores of the kept rows (L2: the direct form sum (c - q)^2)
    hipLaunchKernelGGL(l2 ? rescore_kernel<WDBX_METRIC_L2> : rescore_kernel<WDBX_METRIC_COSINE>, dim3(64, nv), dim3(256), 0, ix->stream,
                       (const f4*)ix->d_rows, (uint32_t)pitch4, (const f4*)qsrc, ix->d_cand, (const uint32_t*)(ix->d_count + q0), cap,
                       (u64*)nullptr, (uint32_t*)nullptr);
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
    if (keys_out) {  // keys with global rows for the caller's own exchange (the in-process shard group)
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
    {  // (marks the block's queries overflowed if a wave lost pairs, then the conditional repair launches)
      bool done = false;
      if ((rc = enqueue_batch_repair(ix, qsrc, nv, k, ix->d_count + q0, cap, f.out_idx, f.out_score, f.out_keys, &done, d_lost))) return rc;
      if (q0 == 0) ix->last_batch_repaired = done;
      else ix->last_batch_repaired = ix->last_batch_repaired && done;
    }
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
  if (cap >= 0xFFFFFF00ull) return fail(WDBX_E_INVALID, "capacity %llu exceeds the 2^32 - 256 rows one shard can number", (u64)cap);
  float* nd = nullptr;
  // (+ TILE_PAD_ROWS rows of slack: the 8-wave tile kernels read whole 256-row tiles and mask rows past the end)
  const size_t bytes = ((size_t)cap + TILE_PAD_ROWS) * ix->pitch * sizeof(float);
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
  // Derived copies (cached norms, bf16 and u8 shadows) of OVERWRITTEN rows are refreshed right here, for exactly
  // those rows: an update in the middle of a large corpus must not invalidate everything behind it.  Rows appended
  // past what a copy covers are picked up lazily by the next search, as before.
  const uint64_t end = first + n;
  if (first < ix->cn_rows && ix->d_cn) {
    const uint64_t e = std::min(end, ix->cn_rows);
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((uint32_t)std::min<uint64_t>((e - first + 3) / 4, 65536)), dim3(256), 0, ix->stream,
                       (const float*)ix->d_rows, (u64)first, (u64)e, (uint32_t)ix->pitch, ix->d_cn, (uint32_t*)nullptr);
    HIP_TRY(hipGetLastError());
    ix->gmax_valid = false;
    ix->cn_stats_dirty = true;  // the running maximum / sum would keep the overwritten rows' old norms
  }
  if (first < ix->shadow_rows && ix->d_rows16) {
    const uint64_t e = std::min(end, ix->shadow_rows);
    const u64 pieces = (e - first) * (ix->pitch16 / 8);
    hipLaunchKernelGGL(rows_to_bf16_kernel, dim3((uint32_t)std::min<u64>((pieces + 255) / 256, 1u << 20)), dim3(256), 0, ix->stream,
                       (const float*)ix->d_rows, (u64)first, (u64)e, (uint32_t)ix->pitch, (__bf16*)ix->d_rows16, ix->pitch16);
    HIP_TRY(hipGetLastError());
  }
  if (first < ix->shadowg_rows && ix->d_rows8g) {  // whole groups: a group's scale depends on all of its rows
    const uint64_t e = std::min(end, ix->shadowg_rows);
    const u64 g0 = first / 64, g1 = (e + 63) / 64;
    hipLaunchKernelGGL(rows_to_i8g_kernel, dim3((uint32_t)std::min<u64>(g1 - g0, 1u << 20)), dim3(256), 0, ix->stream,
                       (const float*)ix->d_rows, g0, g1, (u64)std::max<uint64_t>(ix->n, end), (uint32_t)ix->dim, (uint32_t)ix->pitch,
                       ix->d_rows8g, ix->pitch8g, ix->d_groups8, ix->d_gbad8);
    HIP_TRY(hipGetLastError());
    ix->gref_valid = false;
  }
  if (first < ix->shadow8_rows && ix->d_rows8) {
    const uint64_t e = std::min(end, ix->shadow8_rows);
    hipLaunchKernelGGL(rows_to_u8_kernel, dim3((uint32_t)std::min<uint64_t>((e - first + 3) / 4, 65536)), dim3(256), 0, ix->stream,
                       (const float*)ix->d_rows, (u64)first, (u64)e, (uint32_t)ix->dim, (uint32_t)ix->pitch, ix->d_rows8, ix->pitch8,
                       ix->d_scale8);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(ix->stream));
  return WDBX_OK;
}
