// host_group.h -- the in-process shard group: S shards driven by ONE process, one host thread per shard.
// Part of the single translation unit wdbx_hip.hip (included there, after host_index.h); not a standalone header.
//
// Reference shape being replaced (paths under /root/reference): VectorStore keeps one index object per shard
// (wdbx/core/vector_store.py:111-134) and searches them in a Python loop, concatenates the per-shard lists and sorts
// (:323-345).  Here every shard is a flat index in one GPU's HBM and a search of the group is:
//   1. local stage, all shards at once: each shard's chain of launches (selection scan, re-scoring, top-k: ~7 launches per
//      lone query) is enqueued by the shard's OWN persistent host thread with the shard's device current -- one thread for
//      S devices would serialise S x 7 launches in front of an 80 us scan;
//   2. exchange of the per-shard (row, score) key lists [nq, k]:
//        RCCL  one ncclAllGather per shard on the shard's stream (communicators from ncclCommInitAll; shards on distinct
//              devices), issued by the shard's thread right behind its local stage -- north_star's "RCCL all-gather of
//              per-shard (id, score) tuples over xGMI";
//        COPY  shards that share a device (no RCCL rank per shard possible), or RCCL unavailable: the root stream waits on
//              each shard's event and pulls its list with a device-to-device copy (SURVEY 8e "fallback"; wdbx_group_info
//              reports 0 RCCL ranks so the two can never be confused);
//   3. one merge_kernel launch on the root shard's stream for all nq queries (P = S lists of k -> k_out), results into the
//      group's own buffer on the root device.
// Every buffer the exchange touches between two enqueue steps (queries, key lists, gathered lists, results) belongs to the
// group, and every shard's handle mutex is held (in shard order) while the group enqueues, so a wdbx_index_search running
// concurrently on one of the handles can neither overwrite the group's queries nor read its results.

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <thread>

enum { GROUP_EXCHANGE_RCCL = 1, GROUP_EXCHANGE_COPY = 2 };
constexpr size_t GROUP_STAGE_Q = 256 << 10, GROUP_STAGE_IDX = 256 << 10, GROUP_STAGE_SCORE = 128 << 10;
// one "candidate buffer overflowed" word per shard (lone queries, up to 64 shards); word 64: the lone call's completion word
// (merge_signal_done)
constexpr size_t GROUP_STAGE_FLAGS = 80 * sizeof(uint32_t);

struct GroupShard {
  wdbx_index* ix = nullptr;
  ncclComm_t comm = nullptr;
  // group-owned buffers on this shard's device
  float* d_q = nullptr;       // resident queries [q_rows, pitch]
  size_t q_bytes = 0;
  u64* d_keys = nullptr;      // this shard's key lists of the current chunk [c, k], global rows
  size_t keys_bytes = 0;
  u64* d_gathered = nullptr;  // [S, c, k]: RCCL receive buffer on every shard; COPY: on the root only
  size_t gathered_bytes = 0;
  hipEvent_t ev = nullptr;    // COPY exchange: "this shard's key lists of the chunk are complete"
  char* stage_dev = nullptr;  // the group's mapped host staging area as this shard's device sees it (null: not mapped)
  bool writes_root = false;   // COPY exchange: this shard's kernels can write the root's gathered buffer directly
};

struct wdbx_group {
  std::vector<GroupShard> sh;
  Dispatcher disp;             // one persistent host thread per shard 1 .. S-1 (shard 0 runs on the calling thread): host_dispatch.h
  uint64_t cap_per_shard = 0;  // owned groups: rows per shard; attached groups: the row-number stride between shards
  int dim = 0, metric = 0, exchange = GROUP_EXCHANGE_COPY;
  bool owns_shards = true;     // false: wdbx_group_attach over handles that live on (the facade's per-shard indices)
  uint64_t q_rows = 0;         // resident queries held in every shard's d_q
  // results of the most recent search, on the root shard's device
  int64_t* d_oidx = nullptr;
  float* d_oscore = nullptr;
  size_t out_elems = 0;
  int last_nq = 0, last_k_out = 0;
  // pinned host memory mapped into every shard's device (small blocking searches: the kernels read the queries from and
  // the merge writes the results to host memory directly -- no memcpy calls on the latency path, as in search_host)
  char* h_stage = nullptr;
  hipEvent_t done_ev = nullptr;  // marks a lone staged call on the root's stream (the polled wait's fallback)
  uint32_t lone_seq = 0;
  uint64_t exchanges = 0;      // exchange + merge steps enqueued so far (one per chunk of a call: wdbx_group_stat "exchanges")
  // RCCL exchange: set by a shard whose ncclAllGather could not be enqueued.  Its peers' collectives are already on their
  // streams and can never complete, so the communicators are aborted and the group refuses every later search.
  std::atomic<bool> rccl_failed{false};
  bool unusable = false;
  std::mutex mu;
};

// Run job(s) for every shard s: shard 0 on the calling thread (its device current for the duration), the others on their
// workers, all at once.  Returns the first failure (its message becomes the caller's wdbx_last_error).
static int group_run(wdbx_group* g, const std::function<int(int)>& job) {
  const int dev0 = g->sh[0].ix->device;
  const std::function<int(int)> bound = [&](int s) -> int {
    if (s != 0) return job(s);
    DeviceGuard dg(dev0);
    return job(0);
  };
  return g->disp.run(bound);
}

static void group_bind_device(int device) { (void)hipSetDevice(device); }

static void group_stop_workers(wdbx_group* g) { g->disp.stop(); }

// communicators (when every shard has its own device) and worker threads; exchange_mode 0 = RCCL when possible, else COPY;
// GROUP_EXCHANGE_RCCL = RCCL or fail; GROUP_EXCHANGE_COPY = never RCCL
static int group_finish_setup(wdbx_group* g, int exchange_mode) {
  const int S = (int)g->sh.size();
  bool distinct = true;
  for (int i = 0; i < S; ++i)
    for (int j = 0; j < i; ++j)
      if (g->sh[i].ix->device == g->sh[j].ix->device) distinct = false;
  const char* env = getenv("WDBX_GROUP_EXCHANGE");
  if (exchange_mode == 0 && env && !strcmp(env, "copy")) exchange_mode = GROUP_EXCHANGE_COPY;
  if (exchange_mode == 0 && env && !strcmp(env, "rccl")) exchange_mode = GROUP_EXCHANGE_RCCL;
  g->exchange = GROUP_EXCHANGE_COPY;
  if (exchange_mode != GROUP_EXCHANGE_COPY) {
    if (!distinct) {
      if (exchange_mode == GROUP_EXCHANGE_RCCL)
        return fail(WDBX_E_INVALID, "shards share a device (RCCL needs one rank per device)");
    } else {
      std::vector<int> devs(S);
      std::vector<ncclComm_t> comms(S, nullptr);
      for (int i = 0; i < S; ++i) devs[i] = g->sh[i].ix->device;
      ncclResult_t r = ncclCommInitAll(comms.data(), S, devs.data());
      if (r == ncclSuccess) {
        for (int i = 0; i < S; ++i) g->sh[i].comm = comms[i];
        g->exchange = GROUP_EXCHANGE_RCCL;
      } else if (exchange_mode == GROUP_EXCHANGE_RCCL) {
        return fail(WDBX_E_RCCL, "ncclCommInitAll failed: %s", ncclGetErrorString(r));
      } else {
        fprintf(stderr, "wdbx_hip: ncclCommInitAll over %d devices failed (%s); the shard group exchanges by device copies\n", S,
                ncclGetErrorString(r));
      }
    }
  }
  for (int i = 0; i < S; ++i) {
    DeviceGuard dg(g->sh[i].ix->device);
    HIP_TRY(hipEventCreateWithFlags(&g->sh[i].ev, hipEventDisableTiming));
  }
  {  // mapped staging (optional: without it the blocking search copies)
    void* hp = nullptr;
    // (coherent whatever HIP_HOST_COHERENT says)
    if (hipHostMalloc(&hp, GROUP_STAGE_Q + GROUP_STAGE_IDX + GROUP_STAGE_SCORE + GROUP_STAGE_FLAGS,
                      hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
      bool ok = true;
      for (int i = 0; i < S && ok; ++i) {
        DeviceGuard dg(g->sh[i].ix->device);
        void* dp = nullptr;
        ok = hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess;
        g->sh[i].stage_dev = (char*)dp;
      }
      if (ok) {
        g->h_stage = (char*)hp;
      } else {
        (void)hipHostFree(hp);
        for (GroupShard& s : g->sh) s.stage_dev = nullptr;
      }
    }
    (void)hipGetLastError();
  }
  // COPY exchange: a shard on the root's device -- or on a peer that may write the root's memory -- lets its final top-k
  // kernel write the root's gathered buffer itself; otherwise the root pulls the list with a peer copy
  const int root_dev = g->sh[0].ix->device;
  for (int i = 1; i < S; ++i) {
    const int dev = g->sh[i].ix->device;
    if (dev == root_dev) {
      g->sh[i].writes_root = true;
    } else {
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, dev, root_dev) == hipSuccess && can) {
        DeviceGuard dg(dev);
        const hipError_t e = hipDeviceEnablePeerAccess(root_dev, 0);
        g->sh[i].writes_root = (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled);
      }
      (void)hipGetLastError();
    }
  }
  std::vector<int> worker_devices;
  for (int i = 1; i < S; ++i) worker_devices.push_back(g->sh[i].ix->device);
  g->disp.start(worker_devices, group_bind_device);
  return WDBX_OK;
}

static void group_free(wdbx_group* g) {
  group_stop_workers(g);
  for (size_t i = 0; i < g->sh.size(); ++i) {
    GroupShard& s = g->sh[i];
    if (!s.ix) continue;
    DeviceGuard dg(s.ix->device);
    (void)hipStreamSynchronize(s.ix->stream);
    if (s.comm) (void)ncclCommDestroy(s.comm);
    if (s.d_q) (void)hipFree(s.d_q);
    if (s.d_keys) (void)hipFree(s.d_keys);
    if (s.d_gathered) (void)hipFree(s.d_gathered);
    if (s.ev) (void)hipEventDestroy(s.ev);
    if (i == 0) {
      if (g->d_oidx) (void)hipFree(g->d_oidx);
      if (g->d_oscore) (void)hipFree(g->d_oscore);
      if (g->h_stage) (void)hipHostFree(g->h_stage);
      if (g->done_ev) (void)hipEventDestroy(g->done_ev);
    }
    if (!g->owns_shards) {
      std::lock_guard<std::mutex> li(s.ix->mu);
      s.ix->row_base = 0;
    }
  }
  if (g->owns_shards)
    for (GroupShard& s : g->sh) wdbx_index_destroy(s.ix);
  delete g;
}

// every shard's handle mutex, in shard order (the only multi-handle locker, so the order cannot deadlock)
struct GroupLocks : OrderedLocks {
  static std::vector<std::mutex*> of(wdbx_group* g) {
    std::vector<std::mutex*> mus;
    mus.reserve(g->sh.size());
    for (GroupShard& s : g->sh) mus.push_back(&s.ix->mu);
    return mus;
  }
  explicit GroupLocks(wdbx_group* g) : OrderedLocks(of(g)) {}
};

// Wait for everything the shards' streams hold.  After a failed enqueue: the shards that did enqueue still read the call's
// queries and write keys, overflow flags and results; the next call re-zeroes / rewrites the same areas.
static void group_drain(wdbx_group* g) {
  for (GroupShard& s : g->sh) {
    DeviceGuard dg(s.ix->device);
    (void)hipStreamSynchronize(s.ix->stream);
  }
  (void)hipGetLastError();
}

// error path of the blocking group searches: every shard's stream drained before the code travels up (the message survives)
static int group_fail_drained(wdbx_group* g, int rc) {
  const std::string keep = g_err;
  group_drain(g);
  g_err = keep;
  return rc;
}

// A shard failed to join a collective its peers have already enqueued: those all-gathers can never complete and every later
// hipStreamSynchronize on their streams would hang.  ncclCommAbort releases them; the group is unusable from then on.
static void group_abort_rccl(wdbx_group* g) {
  for (GroupShard& s : g->sh) {
    if (!s.comm) continue;
    DeviceGuard dg(s.ix->device);
    (void)ncclCommAbort(s.comm);
    s.comm = nullptr;
  }
  g->unusable = true;
}

// queries into every shard's resident query buffer: from the host (each shard's thread copies its own), or generated on
// each device (counter-based generator of BASELINE.md section 3)
static int group_load_queries(wdbx_group* g, const float* host, uint64_t seed, uint64_t row0, int nq, int normalize) {
  const std::function<int(int)> job = [&](int s) -> int {
    GroupShard& gs = g->sh[s];
    wdbx_index* ix = gs.ix;
    int rc;
    const size_t bytes = (size_t)nq * ix->pitch * sizeof(float);
    if ((rc = grow((void**)&gs.d_q, &gs.q_bytes, bytes))) return rc;
    if (host) {
      if (ix->pitch == ix->dim) {
        HIP_TRY(hipMemcpyAsync(gs.d_q, host, bytes, hipMemcpyHostToDevice, ix->stream));
      } else {
        HIP_TRY(hipMemsetAsync(gs.d_q, 0, bytes, ix->stream));
        HIP_TRY(hipMemcpy2DAsync(gs.d_q, (size_t)ix->pitch * sizeof(float), host, (size_t)ix->dim * sizeof(float),
                                 (size_t)ix->dim * sizeof(float), nq, hipMemcpyHostToDevice, ix->stream));
      }
      if (normalize && ix->metric == WDBX_METRIC_COSINE && (rc = launch_normalize(ix, gs.d_q, nq))) return rc;
    } else if ((rc = launch_fill(ix, gs.d_q, seed, row0, nq, normalize && ix->metric == WDBX_METRIC_COSINE))) {
      return rc;
    }
    return WDBX_OK;
  };
  int rc = group_run(g, job);
  if (rc == WDBX_OK) g->q_rows = (uint64_t)nq;
  return rc;
}

// Enqueue the search of resident queries [first, first + nq) on every shard, the exchange and the merge; results
// [nq, k_out] are left in the group's result buffer on the root device (stream-ordered on the root shard's stream).
// staged: the queries are the first nq rows of the mapped host staging area (not the resident buffers) and the results go
// to its result slots.  Caller holds g->mu and every shard's mutex.
// allow_batch: a call with enough queries may answer them with ONE batched matrix-core pass per shard (the blocking entry
// point, like wdbx_index_search); without it every query makes its own scan (the resident entry point, like
// wdbx_index_search_device -- what "one step = one single-query scan" of bench.py needs).
// masks: per shard, host mask words for this call (bit r of word r / 32 = row r may be returned; null entry = every row) or
// null: the metadata filter pushed down into every shard's scan.
// defer_repair (a staged lone query): the shards' selection scans skip their queued repair launches and leave an overflow
// word each in the staging area's flags; the caller looks at them after its synchronisation and re-runs the call if one is set.
static int group_enqueue_search(wdbx_group* g, int first, int nq, int k, int k_out, bool staged, bool allow_batch,
                                const uint32_t* const* masks = nullptr, bool defer_repair = false) {
  const int S = (int)g->sh.size();
  if (nq <= 0) return WDBX_OK;
  if (g->unusable) return fail(WDBX_E_STATE, "the group's RCCL communicators were aborted after a failed collective: destroy and re-create the group");
  if (k < 1 || k > WDBX_MAX_K) return fail(WDBX_E_INVALID, "k=%d outside [1, %d]", k, WDBX_MAX_K);
  if (k_out < k || k_out > WDBX_MAX_K || (int64_t)k_out > (int64_t)S * k)
    return fail(WDBX_E_INVALID, "k_out=%d outside [k=%d, min(%d, shards*k=%lld)]", k_out, k, WDBX_MAX_K, (long long)S * k);
  if (!staged && (first < 0 || (uint64_t)first + (uint64_t)nq > g->q_rows))
    return fail(WDBX_E_INVALID, "queries [%d, +%d) outside the %llu resident queries", first, nq, (u64)g->q_rows);
  for (int s = 0; s < S; ++s) {
    const wdbx_index* ix = g->sh[s].ix;
    if (!g->owns_shards && ix->n > g->cap_per_shard) return fail(WDBX_E_STATE, "shard %d outgrew the group's row-number stride", s);
    if (ix->row_base + ix->n >= 0xFFFFFF00ull) return fail(WDBX_E_INVALID, "shard %d: global rows exceed 32-bit row keys", s);
  }
  GroupShard& root = g->sh[0];
  int rc;
  int64_t* out_idx = staged ? (int64_t*)(root.stage_dev + GROUP_STAGE_Q) : nullptr;
  float* out_score = staged ? (float*)(root.stage_dev + GROUP_STAGE_Q + GROUP_STAGE_IDX) : nullptr;
  if (!staged) {
    DeviceGuard dg(root.ix->device);
    const size_t elems = (size_t)nq * k_out;
    if (elems > g->out_elems) {
      HIP_TRY(hipStreamSynchronize(root.ix->stream));
      if (g->d_oidx) HIP_TRY(hipFree(g->d_oidx));
      if (g->d_oscore) HIP_TRY(hipFree(g->d_oscore));
      g->d_oidx = nullptr;
      g->d_oscore = nullptr;
      g->out_elems = 0;
      HIP_TRY(hipMalloc((void**)&g->d_oidx, elems * sizeof(int64_t)));
      HIP_TRY(hipMalloc((void**)&g->d_oscore, elems * sizeof(float)));
      g->out_elems = elems;
    }
    out_idx = g->d_oidx;
    out_score = g->d_oscore;
  }
  // chunks keep the gathered lists (S * c * k keys) under 64 MiB whatever nq and k are
  const int chunk = (int)std::max<size_t>(32, std::min<size_t>((size_t)nq, ((size_t)64 << 20) / ((size_t)S * k * sizeof(u64))));
  for (int c0 = 0; c0 < nq; c0 += chunk) {
    const int c = std::min(chunk, nq - c0);
    const std::function<int(int)> local = [&](int s) -> int {
      GroupShard& gs = g->sh[s];
      wdbx_index* ix = gs.ix;
      int r;
      // COPY exchange: the list goes straight into its slot of the root's gathered buffer when this shard can write there
      const bool direct = g->exchange == GROUP_EXCHANGE_COPY && (s == 0 || gs.writes_root);
      // (RCCL: d_keys / d_gathered of EVERY shard were sized before the dispatch -- nothing below may fail between "the peers
      // have enqueued the collective" and "this shard enqueues it")
      if (g->exchange != GROUP_EXCHANGE_RCCL && !direct && (r = grow((void**)&gs.d_keys, &gs.keys_bytes, (size_t)c * k * sizeof(u64)))) return r;
      u64* const keys = direct ? g->sh[0].d_gathered + (size_t)s * c * k : gs.d_keys;
      const float* q = staged ? (const float*)gs.stage_dev + (size_t)c0 * ix->pitch : gs.d_q + (size_t)(first + c0) * ix->pitch;
      struct MaskScope {  // the mask applies to this enqueue only (the kernels take the pointer at launch)
        wdbx_index* ix;
        ~MaskScope() { ix->active_mask = nullptr; }
      } scope{ix};
      struct DeferScope {
        wdbx_index* ix;
        ~DeferScope() { ix->defer_flag_dev = nullptr; }
      } dscope{ix};
      if (defer_repair && staged && nq == 1 && gs.stage_dev)
        ix->defer_flag_dev = (uint32_t*)(gs.stage_dev + GROUP_STAGE_Q + GROUP_STAGE_IDX + GROUP_STAGE_SCORE) + s;
      const auto stage = [&]() -> int {  // this shard's local stage
        int r2;
        if (masks && masks[s] && ix->n) {
          const size_t words = (size_t)((ix->n + 31) / 32);
          if ((r2 = grow((void**)&ix->d_mask, &ix->mask_bytes, words * sizeof(uint32_t)))) return r2;
          HIP_TRY(hipMemcpyAsync(ix->d_mask, masks[s], words * sizeof(uint32_t), hipMemcpyHostToDevice, ix->stream));
          ix->active_mask = ix->d_mask;
        }
        if (allow_batch && ix->n && !ix->active_mask && !use_select(ix, k) && ix->opt_batch_repair && gemm_eligible(ix, c, k)) {
          // enough queries for ONE matrix-core pass over this shard (i8 / bf16 selection tiles + exact re-scoring, overflowed
          // queries repaired by conditional launches): the shard's lists come out as keys all the same
          return enqueue_search_gemm(ix, q, c, k, nullptr, nullptr, SEARCH_FINAL, -1, keys);
        }
        for (int b0 = 0; b0 < c; b0 += 32) {  // rounds of 32 queries share the small kernels around the scans
          const int b = std::min(32, c - b0);
          if ((r2 = enqueue_search(ix, q + (size_t)b0 * ix->pitch, b, k, nullptr, nullptr, SEARCH_LOCAL_KEYS, keys + (size_t)b0 * k)))
            return r2;
        }
        return WDBX_OK;
      };
      const int stage_rc = stage();
      const std::string stage_err = stage_rc ? g_err : std::string();
      // (RCCL: a shard whose local stage failed still joins the collective -- its buffers exist -- or the other shards'
      // all-gathers would wait for it for ever; the call then fails with this shard's error)
      if (g->exchange == GROUP_EXCHANGE_RCCL) {
        const ncclResult_t nr = ncclAllGather(gs.d_keys, gs.d_gathered, (size_t)c * k, ncclUint64, gs.comm, ix->stream);
        if (nr != ncclSuccess) {
          g->rccl_failed.store(true, std::memory_order_release);
          return fail(WDBX_E_RCCL, "shard %d: ncclAllGather failed: %s", s, ncclGetErrorString(nr));
        }
      } else if (s != 0 && stage_rc == WDBX_OK)
        HIP_TRY(hipEventRecord(gs.ev, ix->stream));
      if (stage_rc) g_err = stage_err;
      return stage_rc;
    };
    if (g->exchange == GROUP_EXCHANGE_COPY) {  // (the root's gathered buffer must exist before any shard writes into it)
      const size_t need = (size_t)S * c * k * sizeof(u64);
      if (need > root.gathered_bytes)  // growing frees the old one: no shard (a peer device's stream included) may still write it
        for (int s = 0; s < S; ++s) {
          DeviceGuard ds(g->sh[s].ix->device);
          HIP_TRY(hipStreamSynchronize(g->sh[s].ix->stream));
        }
      DeviceGuard dg(root.ix->device);
      if ((rc = grow((void**)&root.d_gathered, &root.gathered_bytes, need))) return rc;
    }
    if (g->exchange == GROUP_EXCHANGE_RCCL) {
      // every buffer the collective touches, on every shard, before any shard's thread starts (grow() frees the old
      // allocation, which waits for the device: no earlier chunk's collective still uses it)
      for (int s = 0; s < S; ++s) {
        GroupShard& gs = g->sh[s];
        DeviceGuard ds(gs.ix->device);
        if ((rc = grow((void**)&gs.d_keys, &gs.keys_bytes, (size_t)c * k * sizeof(u64)))) return rc;
        if ((rc = grow((void**)&gs.d_gathered, &gs.gathered_bytes, (size_t)S * c * k * sizeof(u64)))) return rc;
      }
    }
    rc = group_run(g, local);
    if (g->rccl_failed.load(std::memory_order_acquire)) {
      const std::string keep = g_err;
      group_abort_rccl(g);
      g_err = keep;
      return rc ? rc : fail(WDBX_E_RCCL, "a shard could not join the all-gather; the group's communicators were aborted");
    }
    if (rc) return rc;
    DeviceGuard dg(root.ix->device);
    if (g->exchange == GROUP_EXCHANGE_COPY) {
      const size_t bytes = (size_t)c * k * sizeof(u64);
      for (int s = 1; s < S; ++s) {
        GroupShard& gs = g->sh[s];
        HIP_TRY(hipStreamWaitEvent(root.ix->stream, gs.ev, 0));
        if (!gs.writes_root)
          HIP_TRY(hipMemcpyPeerAsync(root.d_gathered + (size_t)s * c * k, root.ix->device, gs.d_keys, gs.ix->device, bytes,
                                     root.ix->stream));
      }
    }
    MergeArgs m = {};
    m.list_len = k;
    m.in = root.d_gathered;
    m.q_stride = (uint64_t)k;
    m.i_stride = 1;
    m.p_stride = (uint64_t)c * k;
    m.P = (uint32_t)S;
    m.k = k_out;
    m.metric = root.ix->metric;
    m.out_idx = out_idx + (size_t)c0 * k_out;
    m.out_score = out_score + (size_t)c0 * k_out;
    if ((rc = launch_merge(root.ix, m, c))) return rc;
    ++g->exchanges;
    if (g->exchange == GROUP_EXCHANGE_COPY && S > 1) {
      // the next local stage of a shard (next chunk, next call) overwrites its key lists: it must wait for these copies
      HIP_TRY(hipEventRecord(root.ev, root.ix->stream));
      for (int s = 1; s < S; ++s) {
        DeviceGuard ds(g->sh[s].ix->device);
        HIP_TRY(hipStreamWaitEvent(g->sh[s].ix->stream, root.ev, 0));
      }
    }
  }
  if (!staged) {
    g->last_nq = nq;
    g->last_k_out = k_out;
  }
  return WDBX_OK;
}
