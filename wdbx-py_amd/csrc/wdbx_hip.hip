// wdbx_hip.hip -- MI355X (gfx950, CDNA4) implementation of the WDBX vector_search hot path.
//
// What the reference does on this path (paths under /root/reference):
//   FaissIndex.search   wdbx/core/indexing.py:983-1030  exact inner product of one unit query against
//                                                        every stored unit row, k best descending
//   FaissIndex.add      indexing.py:858-905, :921-968    rows normalised (:851-856) and appended
//   VectorStore.search  wdbx/core/vector_store.py:323-345 per-shard top-`limit`, concatenate, sort, cut
// Here: the corpus lives row-major fp32 in HBM (plus optional reduced-precision shadow copies used only to
// select candidates); every query makes one streaming pass; a small kernel merges partial lists or ranks the
// exactly re-scored candidates; across GPUs the per-shard lists are all-gathered with RCCL and merged again.
//
// One translation unit, in parts (included below in this order):
//   kernels_common.h        64-bit ordering keys, per-wave top-k list, lane-group reductions
//   kernels_scan.h          scan_kernel<L,QPL,METRIC,NT,MODE,RAGGED>, scan_kernel_generic: the fp32 scan of one
//                           query (HBM-bound, 16-byte non-temporal loads straight into VGPRs, query in VGPRs, DPP
//                           tree, per-wave threshold top-k; MODE 2: key per row for the radix select)
//   kernels_merge_select.h  merge_kernel<REG> (P lists -> 1; also thresholds and candidate top-k), radix select
//   kernels_tiles.h         batched queries on the matrix cores: gemm_topk_kernel (exact fp32 MFMA tiles),
//                           gemm_bf16w8_kernel (bf16 SELECTION tiles over the fp32 rows or the bf16 shadow copy),
//                           their epilogue (sampled maxima / candidate append), rows/queries -> bf16
//   kernels_scan8.h         scan8_kernel: single-query SELECTION scan over the u8 shadow copy (default path of
//                           single queries), rows -> u8 + scales, query norms
//   kernels_aux.h           row norms, threshold margins, rescore_kernel (exact fp32 scores of the candidates),
//                           synthetic fill / normalise, read probes
//   host_index.h            the handle, kernel choice, and the enqueue functions of every search path
//   host_group.h            the in-process shard group: per-shard host threads, exchange (RCCL all-gather / device copies), merge
// The selection paths never decide a result: they keep every row whose score could reach the true k-th best
// under a rigorous error bound, and the kept rows are re-scored in fp32 from the fp32 rows.
//
// Ordering everywhere is one total order on 64-bit keys:
//   key = (orderable(score) << 32) | ~row      (bigger key = better; 0 = empty slot)
// so "score descending, row ascending" is a single unsigned compare, ties are deterministic and a
// merged multi-shard result equals the single-shard result bit for bit.

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <functional>
#include <climits>
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

// error plumbing (g_err, fail), the exception barrier (WDBX_CATCH), the shard dispatcher, ordered locks, grow bookkeeping and
// the option table lookup live in host_dispatch.h: the device-free slice of the host side, which the CPU suite also builds
// with plain g++ under -fsanitize=thread / address,undefined (tests/test_host_dispatch_sanitizers.py)
#include "host_dispatch.h"

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      (void)hipGetLastError(); /* (the runtime's "last error" is sticky: a failed allocation must not fail the next launch check) */ \
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

#include "kernels_common.h"
#include "kernels_scan.h"
#include "kernels_merge_select.h"
#include "kernels_tiles.h"
#include "kernels_scan8.h"
#include "kernels_tiles8.h"
#include "kernels_aux.h"
#include "host_index.h"
#include "host_group.h"

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
static int drain(EventPool& pool, uint64_t* count, double* ms);

// every device allocation the handle holds: the fp32 rows, each shadow copy with its tables, and the scratch buffers
static uint64_t device_bytes_resident(const wdbx_index* ix) {
  uint64_t b = ((uint64_t)ix->cap + TILE_PAD_ROWS) * (uint64_t)ix->pitch * sizeof(float);
  b += ix->rows16_bytes + ix->rows8_bytes + ix->scale8_bytes + ix->rows8g_bytes + ix->groups8_bytes + ix->groups8_bytes / 2;
  b += ix->cn_bytes + ix->gmax_bytes + ix->partials_bytes + ix->local_keys_bytes + ix->gathered_bytes + ix->q_bytes;
  b += (uint64_t)ix->out_elems * 12 + ix->dump_bytes + ix->sel_bytes + ix->state_bytes + ix->mask_bytes + ix->qblock_bytes;
  b += ix->halfmax_bytes + ix->tau_bytes + ix->cand_bytes + ix->count_bytes + ix->qb16_bytes + ix->qn_bytes + ix->selsrc_bytes;
  b += ix->qb8_bytes + ix->qpar_bytes + ix->pairs_bytes + ix->pair_count_bytes + ix->cnmax_bytes;
  return b;
}

extern "C" {

int wdbx_hip_version(void) { return WDBX_HIP_ABI_VERSION; }

const char* wdbx_last_error(void) { return g_err.c_str(); }

int wdbx_device_count(int* out_count) try {
  if (!out_count) return fail(WDBX_E_INVALID, "out_count is null");
  *out_count = 0;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(WDBX_E_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  *out_count = n;
  return WDBX_OK;
} WDBX_CATCH

int wdbx_index_create(int device_id, int dim, int metric, uint64_t capacity_rows, wdbx_index** out) try {
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
} WDBX_CATCH

void wdbx_index_destroy(wdbx_index* ix) try {
  if (!ix) return;
  {
    std::unique_lock<std::mutex> lk(ix->mu);
    // (blocking searches that wait for their event outside the mutex still read their staging slot afterwards)
    ix->slots.wait_all_free(lk);
    DeviceGuard g(ix->device);
    (void)hipStreamSynchronize(ix->stream);
    if (ix->comm) (void)ncclCommDestroy(ix->comm);
    for (hipEvent_t e : ix->scan_ev.ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : ix->merge_ev.ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : ix->gemm_ev.ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : ix->sample_ev.ev) (void)hipEventDestroy(e);
    void* bufs[] = {ix->d_rows, ix->d_partials, ix->d_local_keys, ix->d_gathered, ix->d_q, ix->d_oidx, ix->d_oscore,
                    ix->d_qblock, ix->d_halfmax, ix->d_tau, ix->d_cand, ix->d_count, ix->d_ticket, ix->d_mask, ix->d_dump, ix->d_sel,
                    ix->d_state, ix->d_cn, ix->d_cnmax, ix->d_qb16, ix->d_rows16, ix->d_rows8, ix->d_scale8, ix->d_selsrc,
                    ix->d_gmax, ix->d_qn, ix->d_rows8g, ix->d_groups8, ix->d_gbad8, ix->d_gref8, ix->d_over_list, ix->d_qb8,
                    ix->d_qpar, ix->d_pairs, ix->d_pair_count};
    for (void* p : bufs)
      if (p) (void)hipFree(p);
    if (ix->h_stage) (void)hipHostFree(ix->h_stage);
    for (hipEvent_t e : ix->slot_done)
      if (e) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(ix->stream);
  }
  delete ix;
} WDBX_CATCH_VOID

int wdbx_index_dim(const wdbx_index* ix) { return ix ? ix->dim : 0; }
int wdbx_index_row_pitch(const wdbx_index* ix) { return ix ? ix->pitch : 0; }

int wdbx_index_size(wdbx_index* ix, uint64_t* out_rows) try {
  if (!ix || !out_rows) return fail(WDBX_E_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(ix->mu);
  *out_rows = ix->n;
  return WDBX_OK;
} WDBX_CATCH

int wdbx_index_capacity(wdbx_index* ix, uint64_t* out_rows) try {
  if (!ix || !out_rows) return fail(WDBX_E_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(ix->mu);
  *out_rows = ix->cap;
  return WDBX_OK;
} WDBX_CATCH

int wdbx_index_reserve(wdbx_index* ix, uint64_t capacity_rows) try {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  return reserve_locked(ix, capacity_rows);
} WDBX_CATCH

int wdbx_index_clear(wdbx_index* ix) try {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  HIP_TRY(hipStreamSynchronize(ix->stream));
  ix->n = 0;
  ix->cn_rows = 0;
  ix->cn_stats_dirty = false;
  ix->shadow_rows = 0;
  ix->shadow8_rows = 0;
  ix->shadowg_rows = 0;
  ix->shadowg_tail_n = ~0ull;
  return WDBX_OK;
} WDBX_CATCH

static int ensure_room(wdbx_index* ix, uint64_t extra) {
  const uint64_t need = ix->n + extra;
  if (need <= ix->cap) return WDBX_OK;
  return reserve_locked(ix, std::max(need, ix->cap + ix->cap / 2));
}

int wdbx_index_add(wdbx_index* ix, const float* rows, uint64_t n, int normalize, uint64_t* first_row_out) try {
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
} WDBX_CATCH

int wdbx_index_set_rows(wdbx_index* ix, uint64_t first_row, const float* rows, uint64_t n, int normalize) try {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  if (n && !rows) return fail(WDBX_E_INVALID, "rows is null");
  std::lock_guard<std::mutex> lk(ix->mu);
  if (first_row > ix->n || n > ix->n - first_row)
    return fail(WDBX_E_INVALID, "rows [%llu, +%llu) outside the %llu stored rows", (u64)first_row, (u64)n, (u64)ix->n);
  if (!n) return WDBX_OK;
  DeviceGuard g(ix->device);
  return upload_rows(ix, first_row, rows, n, normalize);
} WDBX_CATCH

int wdbx_index_get_rows(wdbx_index* ix, uint64_t first_row, uint64_t n, float* out_rows) try {
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
} WDBX_CATCH

int wdbx_index_fill_synthetic(wdbx_index* ix, uint64_t seed, uint64_t counter_row0, uint64_t n, int normalize,
                              uint64_t* first_row_out) try {
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
} WDBX_CATCH

// Keep exactly the rows src_rows[0 .. n_keep) (strictly increasing), moved down to rows 0 .. n_keep - 1 in that order:
// the compaction behind HipFlatIndex.optimize() (the reference's rebuild hook, indexing.py:1124-1149).  In place, chunk by
// chunk through a scratch buffer: the sources of a later chunk all lie at or behind that chunk's own destination rows, so
// writing an earlier chunk cannot overwrite them.  Derived copies (norms, shadows) are kept up to the first moved row and
// rebuilt lazily behind it.
int wdbx_index_compact(wdbx_index* ix, const uint64_t* src_rows, uint64_t n_keep) try {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  if (n_keep && !src_rows) return fail(WDBX_E_INVALID, "src_rows is null");
  std::lock_guard<std::mutex> lk(ix->mu);
  if (n_keep > ix->n) return fail(WDBX_E_INVALID, "%llu rows to keep of %llu stored", (u64)n_keep, (u64)ix->n);
  uint64_t first_moved = n_keep;
  for (uint64_t i = 0; i < n_keep; ++i) {
    if (src_rows[i] >= ix->n || (i && src_rows[i] <= src_rows[i - 1]))
      return fail(WDBX_E_INVALID, "src_rows must be strictly increasing row numbers below %llu (entry %llu)", (u64)ix->n, (u64)i);
    if (first_moved == n_keep && src_rows[i] != i) first_moved = i;
  }
  DeviceGuard g(ix->device);
  HIP_TRY(hipStreamSynchronize(ix->stream));
  if (first_moved < n_keep) {
    const uint64_t chunk = std::max<uint64_t>(1024, (256ull << 20) / ((uint64_t)ix->pitch * sizeof(float)));  // 256 MiB of rows at a time
    float* d_tmp = nullptr;
    u64* d_src = nullptr;
    const uint64_t c_max = std::min(chunk, n_keep - first_moved);
    HIP_TRY(hipMalloc((void**)&d_tmp, (size_t)c_max * ix->pitch * sizeof(float)));
    hipError_t e = hipMalloc((void**)&d_src, (size_t)c_max * sizeof(u64));
    for (uint64_t i0 = first_moved; e == hipSuccess && i0 < n_keep; i0 += chunk) {
      const uint64_t c = std::min(chunk, n_keep - i0);
      e = hipMemcpyAsync(d_src, src_rows + i0, (size_t)c * sizeof(u64), hipMemcpyHostToDevice, ix->stream);
      if (e != hipSuccess) break;
      hipLaunchKernelGGL(gather_rows_kernel, dim3((uint32_t)std::min<uint64_t>((c + 3) / 4, 65536)), dim3(256), 0, ix->stream,
                         (const f4*)ix->d_rows, (uint32_t)(ix->pitch / 4), (const u64*)d_src, (u64)c, (f4*)d_tmp);
      e = hipGetLastError();
      if (e == hipSuccess)
        e = hipMemcpyAsync(ix->d_rows + (size_t)i0 * ix->pitch, d_tmp, (size_t)c * ix->pitch * sizeof(float), hipMemcpyDeviceToDevice,
                           ix->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);  // (the host list and the scratch are reused by the next chunk)
    }
    (void)hipFree(d_tmp);
    if (d_src) (void)hipFree(d_src);
    if (e != hipSuccess) return fail(WDBX_E_HIP, "compaction failed: %s (the rows behind row %llu are undefined)", hipGetErrorString(e), (u64)first_moved);
  }
  ix->n = n_keep;
  ix->cn_rows = std::min(ix->cn_rows, first_moved);
  ix->shadow_rows = std::min(ix->shadow_rows, first_moved);
  ix->shadow8_rows = std::min(ix->shadow8_rows, first_moved);
  ix->shadowg_rows = std::min(ix->shadowg_rows, first_moved / 64 * 64);  // (whole 64-row groups: a group's scale depends on all its rows)
  ix->shadowg_tail_n = ~0ull;  // the groups behind the new last row still describe dropped rows: rewritten by the next batch
  ix->cn_stats_dirty = true;  // the running maximum / sum still hold the dropped rows' norms
  ix->gmax_valid = false;
  return WDBX_OK;
} WDBX_CATCH

// mask_word_count: how many words the caller's mask holds (checked against the row count under the handle's lock: a mask built
// before a concurrent add is refused instead of over-read); ~0 = the caller vouches for ceil(rows / 32) words
static int search_host(wdbx_index* ix, const float* queries, int nq, int k, int normalize_queries,
                       const uint32_t* mask_words, uint64_t mask_word_count, int64_t* out_idx, float* out_score) {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  if (nq < 0) return fail(WDBX_E_INVALID, "nq=%d", nq);
  if (nq == 0) return WDBX_OK;
  if (!queries || !out_idx || !out_score) return fail(WDBX_E_INVALID, "null buffer");
  if (k < 1 || k > WDBX_MAX_K) return fail(WDBX_E_INVALID, "k=%d outside [1, %d]", k, WDBX_MAX_K);
  // The handle's mutex covers everything that touches the handle's state: buffer growth, the enqueue of the call's whole
  // chain of launches, option reads.  It does NOT cover the wait for the GPU when the call's queries and results live in a
  // staging slot of its own (small calls: mapped host memory the kernels read and write directly) -- the next caller on
  // another thread enqueues behind this call on the stream while this one waits for its event, so N threads calling one
  // handle (the reference's 4-worker pools per index, indexing.py:692, :1045-1048) pipeline instead of taking turns at
  // wall-clock latency.  Scratch buffers are shared by consecutive calls: the stream runs them in order.
  std::unique_lock<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  struct MaskScope {  // the mask applies to this call only (such a call keeps the mutex to its end: d_mask is one buffer)
    wdbx_index* ix;
    bool set = false;
    ~MaskScope() {
      if (set) ix->active_mask = nullptr;
    }
  } scope{ix};
  int rc;
  const size_t elems = (size_t)nq * k;
  const size_t q_bytes = (size_t)nq * ix->pitch * sizeof(float);
  constexpr size_t STAGE_Q = 256 << 10, STAGE_IDX = 256 << 10, STAGE_SCORE = 128 << 10;
  constexpr size_t SLOT_BYTES = STAGE_Q + STAGE_IDX + STAGE_SCORE + 256;
  // a staging slot of its own for a small call.  Callers beyond STAGE_SLOTS in flight wait for one with the mutex RELEASED,
  // so the slot is taken FIRST, before this call has changed anything in the handle (a row mask set before the wait would
  // be seen by the calls that run meanwhile), and everything the decision rests on is looked at again after a wait.
  struct SlotHold {
    wdbx_index* ix;
    std::unique_lock<std::mutex>* lk;
    int slot = -1;
    ~SlotHold() {
      // (a call that kept the mutex -- a masked one -- keeps it until its mask is reset too: give_back leaves it as it is)
      if (slot >= 0) ix->slots.give_back(slot, *lk);
    }
  } hold{ix, &lk};
  bool gemm, zero_copy;
  for (;;) {
    gemm = !(mask_words && ix->n) && gemm_eligible(ix, nq, k);
    zero_copy = ix->opt_zero_copy && !gemm && q_bytes <= STAGE_Q && elems * sizeof(int64_t) <= STAGE_IDX;
    if (zero_copy && !ix->h_stage) {
      void* hp = nullptr;
      void* dp = nullptr;
      // (coherent whatever HIP_HOST_COHERENT says)
      bool ok = hipHostMalloc(&hp, STAGE_SLOTS * SLOT_BYTES, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
                hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess;
      for (int s = 0; ok && s < STAGE_SLOTS; ++s) ok = hipEventCreateWithFlags(&ix->slot_done[s], hipEventDisableTiming) == hipSuccess;
      if (ok) {
        ix->h_stage = (char*)hp;
        ix->h_stage_dev = (char*)dp;
      } else {
        for (int s = 0; s < STAGE_SLOTS; ++s)
          if (ix->slot_done[s]) {
            (void)hipEventDestroy(ix->slot_done[s]);
            ix->slot_done[s] = nullptr;
          }
        if (hp) (void)hipHostFree(hp);
        (void)hipGetLastError();
        ix->opt_zero_copy = 0;  // not available here: use the copy path from now on
        zero_copy = false;
      }
    }
    if (!zero_copy || hold.slot >= 0) break;
    if ((hold.slot = ix->slots.try_take()) >= 0) break;  // (taken without a wait: nothing can have changed)
    ix->slots.wait(lk);  // mutex released while waiting: decide again afterwards
  }
  if (mask_words && ix->n) {
    const size_t words = (size_t)((ix->n + 31) / 32);
    if (mask_word_count < words)
      return fail(WDBX_E_INVALID, "row mask of %llu words for %llu rows (%zu words needed)", (u64)mask_word_count, (u64)ix->n, words);
    rc = grow((void**)&ix->d_mask, &ix->mask_bytes, words * sizeof(uint32_t));
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(ix->d_mask, mask_words, words * sizeof(uint32_t), hipMemcpyHostToDevice, ix->stream));
    ix->active_mask = ix->d_mask;
    scope.set = true;
  }
  char* const hs = zero_copy ? ix->h_stage + (size_t)hold.slot * SLOT_BYTES : nullptr;      // this call's slot, host view
  char* const ds = zero_copy ? ix->h_stage_dev + (size_t)hold.slot * SLOT_BYTES : nullptr;  // ... and device view
  // wait for this call's launches: with a slot and no row mask, by its own event with the mutex RELEASED
  const bool narrow = zero_copy && !ix->active_mask;
  auto wait_for_gpu = [&]() -> int {
    if (narrow) {
      HIP_TRY(hipEventRecord(ix->slot_done[hold.slot], ix->stream));
      lk.unlock();
      HIP_TRY(hipEventSynchronize(ix->slot_done[hold.slot]));
    } else {
      HIP_TRY(hipStreamSynchronize(ix->stream));
    }
    return WDBX_OK;
  };
  // ... or, when the call's last kernel reports into a word of the slot (merge_signal_done): the event still marks the call on
  // the stream (and is how a failed launch would surface -- looked at every 4 096 spins); the wait itself is a poll of the word,
  // which the kernel wrote behind its results: 5 us less than the runtime's completion path (profiles/r04/poll/)
  auto wait_polled = [&](volatile uint32_t* word, uint32_t seq) -> int {
    HIP_TRY(hipEventRecord(ix->slot_done[hold.slot], ix->stream));
    hipEvent_t ev = ix->slot_done[hold.slot];
    lk.unlock();
    for (uint32_t spins = 1;; ++spins) {
      if (*word == seq) break;
      if ((spins & 0xFFFu) == 0) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) break;  // (the kernel has ended: its stores are visible)
        if (e != hipErrorNotReady) HIP_TRY(hipEventSynchronize(ev));
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);  // (the slot's contents are read after the word, not before)
    return WDBX_OK;
  };
  bool poll_tail = false;  // a small batch whose single merge launch took the signal: the common tail below polls
  uint32_t poll_seq = 0;
  float* dq;
  int64_t* doidx;
  float* doscore;
  if (zero_copy) {
    float* hq = (float*)hs;
    if (ix->pitch == ix->dim) {
      memcpy(hq, queries, q_bytes);
    } else {
      memset(hq, 0, q_bytes);
      for (int q = 0; q < nq; ++q) memcpy(hq + (size_t)q * ix->pitch, queries + (size_t)q * ix->dim, (size_t)ix->dim * sizeof(float));
    }
    dq = (float*)ds;
    doidx = (int64_t*)(ds + STAGE_Q);
    doscore = (float*)(ds + STAGE_Q + STAGE_IDX);
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
    // a query whose candidate buffer overflowed has been re-run exactly by the conditional repair launches queued behind
    // its block (enqueue_batch_repair); shapes without a device-side repair are re-run here
    for (int q = 0; q < nq && !ix->last_batch_repaired; ++q)
      if (counts[q] > ix->last_batch_cap) {
        const int64_t keep = ix->opt_scan_shadow;  // straight to the fp32 scan: the selection would overflow again
        ix->opt_scan_shadow = 0;
        rc = enqueue_search(ix, dq + (size_t)q * ix->pitch, 1, k, doidx + (size_t)q * k, doscore + (size_t)q * k, SEARCH_FINAL);
        ix->opt_scan_shadow = keep;
        if (rc) return rc;
      }
  } else {
    // Lone query through mapped memory: the u8 selection scan skips its queued repair launches and its final merge --
    // the re-scored candidates' keys and their count land in the call's slot and THIS thread ranks them after its wait
    // (an overflowed candidate buffer is repaired then, too); on small shards the threshold is taken inside the full pass.
    // 3 dependent launches instead of 5 (7 with the repairs).
    volatile uint32_t* const over = zero_copy ? (volatile uint32_t*)(hs + STAGE_Q + STAGE_IDX + STAGE_SCORE) : nullptr;
    const bool defer = zero_copy && nq == 1 && !use_select(ix, k);
    uint32_t done_seq = 0;
    if (defer) {
      over[0] = 0;
      over[1] = 0;
      over[2] = 0;
      ix->defer_flag_dev = (uint32_t*)(ds + STAGE_Q + STAGE_IDX + STAGE_SCORE);
      if (narrow && ix->opt_poll_done) {  // the chain's last kernel reports into over[2]; this thread polls it (wait_for_lone)
        if (++ix->lone_seq == 0) ++ix->lone_seq;
        done_seq = ix->lone_seq;
        ix->done_flag_dev = ix->defer_flag_dev + 2;
        ix->done_seq = done_seq;
        ix->done_signals = 0;
      }
      if (ix->opt_lone_host_select) {
        ix->lone_keys_dev = (u64*)(ds + STAGE_Q);
        ix->lone_count_dev = ix->defer_flag_dev + 1;
        ix->lone_cap_max = (uint32_t)((STAGE_IDX + STAGE_SCORE) / sizeof(u64));
      }
    }
    // a small batch (one round of up to 32 queries) on the plain fp32 scan ends in ONE merge launch of nq workgroups: the same
    // completion word, written by the last of them (a ticket)
    const bool batch_poll = !defer && narrow && nq > 1 && nq <= 32 && ix->opt_poll_done && !use_select(ix, k);
    if (batch_poll) {
      if (!ix->d_ticket) {
        HIP_TRY(hipMalloc((void**)&ix->d_ticket, sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(ix->d_ticket, 0, sizeof(uint32_t), ix->stream));
      }
      over[2] = 0;
      if (++ix->lone_seq == 0) ++ix->lone_seq;
      done_seq = ix->lone_seq;
      ix->done_flag_dev = (uint32_t*)(ds + STAGE_Q + STAGE_IDX + STAGE_SCORE) + 2;
      ix->done_seq = done_seq;
      ix->done_signals = 0;
    }
    ix->lone_used = false;
    rc = enqueue_search(ix, dq, nq, k, doidx, doscore, SEARCH_FINAL);
    ix->defer_flag_dev = nullptr;
    ix->lone_keys_dev = nullptr;
    ix->lone_count_dev = nullptr;
    // exactly one launch took the completion signal -- a final merge_kernel -- and it is the chain's last kernel on the plain
    // fp32 scan (0) and on the u8 scan (2) when its candidates are ranked on the device; the u8 scan's usual lone form ends in
    // rescore_kernel (hundreds of waves each storing one key: a host-visibility fence per storing wave costs more than the
    // runtime's path) and the bf16 single-query path (1) queues repair launches behind its merge: event wait as before
    const bool poll = ix->done_flag_dev && ix->done_signals == 1 &&
                      (ix->last_single_path == 0 || (ix->last_single_path == 2 && !batch_poll));  // (a u8 ROUND queues repairs behind its merge)
    ix->done_flag_dev = nullptr;
    if (rc) return rc;
    if (poll && batch_poll) {
      poll_tail = true;
      poll_seq = done_seq;
    }
    if (defer) {
      const bool lone_used = ix->lone_used;         // (handle state: read before the mutex may go)
      const uint32_t cap = ix->last_batch_cap;
      const int metric = ix->metric;
      if (poll) {
        if ((rc = wait_polled(over + 2, done_seq))) return rc;
      } else if ((rc = wait_for_gpu())) return rc;
      bool repair = false;
      if (lone_used) {
        const uint32_t cnt = over[1];
        if (cnt > cap) {
          repair = true;
        } else {  // the exact keys of the kept rows: the k largest, in key order = (score descending, row ascending)
          const u64* hk = (const u64*)(hs + STAGE_Q);
          std::vector<u64> keys(hk, hk + cnt);
          const size_t kk = std::min<size_t>((size_t)k, keys.size());
          if (kk < keys.size()) std::nth_element(keys.begin(), keys.begin() + kk, keys.end(), std::greater<u64>());  // O(n) ...
          std::sort(keys.begin(), keys.begin() + kk, std::greater<u64>());                                                // ... + k log k
          size_t o = 0;
          for (size_t i = 0; i < kk && keys[i]; ++i, ++o) {  // (a zero key = a NaN score: never a result)
            const uint32_t ord = (uint32_t)(keys[i] >> 32);
            const uint32_t u = (ord & 0x80000000u) ? (ord ^ 0x80000000u) : ~ord;
            float sc;
            memcpy(&sc, &u, sizeof sc);
            if (metric == WDBX_METRIC_L2) sc = -sc + 0.0f;
            out_idx[o] = (int64_t)(uint32_t)~(uint32_t)(keys[i] & 0xFFFFFFFFull);
            out_score[o] = sc;
          }
          for (; o < (size_t)k; ++o) {
            out_idx[o] = -1;
            out_score[o] = 0.0f;
          }
          return WDBX_OK;
        }
      } else if (over[0]) {
        repair = true;
      }
      if (repair) {  // (rare: back under the mutex, the exact scan into the same slot)
        if (!lk.owns_lock()) lk.lock();
        const int64_t keep = ix->opt_scan_shadow;
        ix->opt_scan_shadow = 0;
        rc = enqueue_search(ix, dq, nq, k, doidx, doscore, SEARCH_FINAL);
        ix->opt_scan_shadow = keep;
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(ix->stream));
      }
      memcpy(out_idx, hs + STAGE_Q, elems * sizeof(int64_t));
      memcpy(out_score, hs + STAGE_Q + STAGE_IDX, elems * sizeof(float));
      return WDBX_OK;
    }
  }
  if (zero_copy) {
    if (poll_tail) {
      if ((rc = wait_polled((volatile uint32_t*)(hs + STAGE_Q + STAGE_IDX + STAGE_SCORE) + 2, poll_seq))) return rc;
    } else if ((rc = wait_for_gpu())) return rc;
    memcpy(out_idx, hs + STAGE_Q, elems * sizeof(int64_t));
    memcpy(out_score, hs + STAGE_Q + STAGE_IDX, elems * sizeof(float));
  } else {
    HIP_TRY(hipMemcpyAsync(out_idx, doidx, elems * sizeof(int64_t), hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(hipMemcpyAsync(out_score, doscore, elems * sizeof(float), hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(hipStreamSynchronize(ix->stream));
  }
  return WDBX_OK;
}

int wdbx_index_search(wdbx_index* ix, const float* queries, int nq, int k, int normalize_queries, int64_t* out_idx,
                      float* out_score) try {
  return search_host(ix, queries, nq, k, normalize_queries, nullptr, 0, out_idx, out_score);
} WDBX_CATCH

int wdbx_index_search_masked(wdbx_index* ix, const float* queries, int nq, int k, int normalize_queries,
                             const uint32_t* mask_words, int64_t* out_idx, float* out_score) try {
  if (!mask_words) return fail(WDBX_E_INVALID, "mask_words is null");
  return search_host(ix, queries, nq, k, normalize_queries, mask_words, ~0ull, out_idx, out_score);
} WDBX_CATCH

int wdbx_index_search_masked_n(wdbx_index* ix, const float* queries, int nq, int k, int normalize_queries,
                               const uint32_t* mask_words, uint64_t mask_word_count, int64_t* out_idx, float* out_score) try {
  if (!mask_words) return fail(WDBX_E_INVALID, "mask_words is null");
  return search_host(ix, queries, nq, k, normalize_queries, mask_words, mask_word_count, out_idx, out_score);
} WDBX_CATCH

int wdbx_device_alloc(wdbx_index* ix, uint64_t bytes, void** out_dev_ptr) try {
  if (!ix || !out_dev_ptr) return fail(WDBX_E_INVALID, "null argument");
  *out_dev_ptr = nullptr;
  DeviceGuard g(ix->device);
  HIP_TRY(hipMalloc(out_dev_ptr, bytes ? bytes : 1));
  return WDBX_OK;
} WDBX_CATCH

int wdbx_device_free(wdbx_index* ix, void* dev_ptr) try {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  if (!dev_ptr) return WDBX_OK;
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  HIP_TRY(hipStreamSynchronize(ix->stream));
  HIP_TRY(hipFree(dev_ptr));
  return WDBX_OK;
} WDBX_CATCH

int wdbx_device_upload(wdbx_index* ix, void* dev_dst, const void* host_src, uint64_t bytes) try {
  if (!ix || (bytes && (!dev_dst || !host_src))) return fail(WDBX_E_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  HIP_TRY(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ix->stream));
  HIP_TRY(hipStreamSynchronize(ix->stream));
  return WDBX_OK;
} WDBX_CATCH

int wdbx_device_download(wdbx_index* ix, void* host_dst, const void* dev_src, uint64_t bytes) try {
  if (!ix || (bytes && (!host_dst || !dev_src))) return fail(WDBX_E_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  HIP_TRY(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ix->stream));
  HIP_TRY(hipStreamSynchronize(ix->stream));
  return WDBX_OK;
} WDBX_CATCH

int wdbx_device_fill_synthetic(wdbx_index* ix, float* dev_dst, uint64_t seed, uint64_t counter_row0, uint64_t n,
                               int normalize) try {
  if (!ix || (n && !dev_dst)) return fail(WDBX_E_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  int rc = launch_fill(ix, dev_dst, seed, counter_row0, n, normalize);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(ix->stream));
  return WDBX_OK;
} WDBX_CATCH

int wdbx_index_search_device(wdbx_index* ix, const float* d_queries, int nq, int k, int64_t* d_out_idx,
                             float* d_out_score) try {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  return enqueue_search(ix, d_queries, nq, k, d_out_idx, d_out_score, SEARCH_FINAL);
} WDBX_CATCH

int wdbx_index_search_sharded_device(wdbx_index* ix, const float* d_queries, int nq, int k, int64_t* d_out_idx,
                                     float* d_out_score) try {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  return enqueue_search(ix, d_queries, nq, k, d_out_idx, d_out_score, SEARCH_SHARDED);
} WDBX_CATCH

int wdbx_index_search_sharded_batch_device(wdbx_index* ix, const float* d_queries, int nq, int k, int64_t* d_out_idx,
                                           float* d_out_score) try {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  if (!gemm_eligible(ix, std::max(nq, (int)ix->opt_gemm_min_nq), k))
    return fail(WDBX_E_STATE, "batched MFMA path needs >= %lld rows and k*1024 <= rows on every rank",
                (long long)ix->opt_gemm_min_rows);
  return enqueue_search_gemm(ix, d_queries, nq, k, d_out_idx, d_out_score, SEARCH_SHARDED);
} WDBX_CATCH

int wdbx_index_search_batch_device(wdbx_index* ix, const float* d_queries, int nq, int k, int64_t* d_out_idx,
                                   float* d_out_score) try {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  if (!gemm_eligible(ix, std::max(nq, (int)ix->opt_gemm_min_nq), k))
    return fail(WDBX_E_STATE, "batched MFMA path needs >= %lld rows and k*1024 <= rows", (long long)ix->opt_gemm_min_rows);
  return enqueue_search_gemm(ix, d_queries, nq, k, d_out_idx, d_out_score);
} WDBX_CATCH

int wdbx_index_batch_status(wdbx_index* ix, uint32_t* out_counts, int nq, uint32_t* out_capacity, int* out_overflowed) try {
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
} WDBX_CATCH

int wdbx_index_profile_read_gemm(wdbx_index* ix, uint64_t* launches, double* ms_total) try {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  HIP_TRY(hipStreamSynchronize(ix->stream));
  return drain(ix->gemm_ev, launches, ms_total);
} WDBX_CATCH

int wdbx_index_profile_read_sample(wdbx_index* ix, uint64_t* launches, double* ms_total) try {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  HIP_TRY(hipStreamSynchronize(ix->stream));
  return drain(ix->sample_ev, launches, ms_total);
} WDBX_CATCH

int wdbx_index_synchronize(wdbx_index* ix) try {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  HIP_TRY(hipStreamSynchronize(ix->stream));
  return WDBX_OK;
} WDBX_CATCH

int wdbx_comm_unique_id(void* out_128_bytes) try {
  if (!out_128_bytes) return fail(WDBX_E_INVALID, "null argument");
  static_assert(sizeof(ncclUniqueId) <= WDBX_UNIQUE_ID_BYTES, "unique id larger than the ABI slot");
  ncclUniqueId id;
  NCCL_TRY(ncclGetUniqueId(&id));
  memset(out_128_bytes, 0, WDBX_UNIQUE_ID_BYTES);
  memcpy(out_128_bytes, &id, sizeof id);
  return WDBX_OK;
} WDBX_CATCH

int wdbx_index_comm_init(wdbx_index* ix, int nranks, int rank, const void* unique_id_128_bytes,
                         uint64_t global_row_base) try {
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
} WDBX_CATCH

int wdbx_index_comm_destroy(wdbx_index* ix) try {
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
} WDBX_CATCH

int wdbx_index_comm_info(wdbx_index* ix, int* out_nranks, int* out_rank, uint64_t* out_row_base) try {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  int n = 0, r = -1;
  if (ix->comm) {  // what RCCL itself says about the communicator, not what the host side asked for
    NCCL_TRY(ncclCommCount(ix->comm, &n));
    NCCL_TRY(ncclCommUserRank(ix->comm, &r));
  }
  if (out_nranks) *out_nranks = n;
  if (out_rank) *out_rank = r;
  if (out_row_base) *out_row_base = ix->row_base;
  return WDBX_OK;
} WDBX_CATCH

int wdbx_index_comm_set_row_base(wdbx_index* ix, uint64_t global_row_base) try {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  if (global_row_base >= 0xFFFFFFFFull) return fail(WDBX_E_INVALID, "global row base exceeds 32-bit row keys");
  std::lock_guard<std::mutex> lk(ix->mu);
  if (!ix->comm) return fail(WDBX_E_STATE, "no communicator on this handle");
  DeviceGuard g(ix->device);
  HIP_TRY(hipStreamSynchronize(ix->stream));
  ix->row_base = global_row_base;
  return WDBX_OK;
} WDBX_CATCH

int wdbx_index_probe_read(wdbx_index* ix, int nontemporal, int blocks, int reps, double* out_ms_per_pass) try {
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
} WDBX_CATCH


// ------------------------------------------------------------------------------------------------
// in-process shard group (host_group.h): S shards driven by one process, one host thread per shard -- the reference's
// VectorStore(num_shards=S) shape (vector_store.py:111-134, :323-345)
// ------------------------------------------------------------------------------------------------
int wdbx_group_create(const int* device_ids, int n, int dim, int metric, uint64_t cap_per_shard, wdbx_group** out) try {
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
      g->sh.emplace_back();
      g->sh.back().ix = ix;
    }
  }
  if (rc == WDBX_OK) rc = group_finish_setup(g, 0);
  if (rc != WDBX_OK) {
    const std::string keep = g_err;
    group_free(g);
    g_err = keep;
    return rc;
  }
  *out = g;
  return WDBX_OK;
} WDBX_CATCH

// A group over EXISTING shard handles (the facade's one index per shard, vector_store.py:111-134): the handles stay
// owned by the caller and keep growing through wdbx_index_add; the group adds the exchange (RCCL communicators from
// ncclCommInitAll when every shard has its own device, device copies otherwise) and gives shard s the row numbers
// [s * stride, (s + 1) * stride) in merged results, stride = (2^32 - 256) / n.  Shard order = row order, so ties come
// back in the order of the reference's stable sort over its shard loop (vector_store.py:323-330).
int wdbx_group_attach_ex(wdbx_index* const* shards, int n, int exchange_mode, wdbx_group** out) try {
  if (!out) return fail(WDBX_E_INVALID, "out is null");
  *out = nullptr;
  if (!shards || n < 1 || n > 64) return fail(WDBX_E_INVALID, "need 1..64 shard handles");
  if (exchange_mode < 0 || exchange_mode > 2) return fail(WDBX_E_INVALID, "exchange_mode=%d (0 auto, 1 RCCL, 2 device copies)", exchange_mode);
  for (int i = 0; i < n; ++i) {
    if (!shards[i]) return fail(WDBX_E_INVALID, "shard %d is null", i);
    if (shards[i]->dim != shards[0]->dim || shards[i]->metric != shards[0]->metric)
      return fail(WDBX_E_INVALID, "shard %d differs from shard 0 in dim or metric", i);
    if (shards[i]->comm) return fail(WDBX_E_STATE, "shard %d already belongs to a per-rank communicator", i);
    for (int j = 0; j < i; ++j)
      if (shards[i] == shards[j]) return fail(WDBX_E_INVALID, "shard handle %d listed twice", i);
  }
  wdbx_group* g = new (std::nothrow) wdbx_group();
  if (!g) return fail(WDBX_E_NOMEM, "host allocation failed");
  g->owns_shards = false;
  g->dim = shards[0]->dim;
  g->metric = shards[0]->metric;
  g->cap_per_shard = 0xFFFFFF00ull / (uint64_t)n;
  g->sh.resize(n);
  for (int i = 0; i < n; ++i) g->sh[i].ix = shards[i];
  int rc = group_finish_setup(g, exchange_mode);
  if (rc != WDBX_OK) {
    const std::string keep = g_err;
    group_free(g);
    g_err = keep;
    return rc;
  }
  for (int i = 0; i < n; ++i) {
    std::lock_guard<std::mutex> li(shards[i]->mu);
    shards[i]->row_base = (uint64_t)i * g->cap_per_shard;
  }
  *out = g;
  return WDBX_OK;
} WDBX_CATCH

int wdbx_group_attach(wdbx_index* const* shards, int n, wdbx_group** out) try {
  return wdbx_group_attach_ex(shards, n, 0, out);
} WDBX_CATCH

void wdbx_group_destroy(wdbx_group* g) try {
  if (!g) return;
  group_free(g);
} WDBX_CATCH_VOID

int wdbx_group_info(wdbx_group* g, int* out_shards, int* out_rccl_nranks, uint64_t* out_row_stride) try {
  if (!g) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(g->mu);
  int n = 0;  // what RCCL itself says; 0 = the group exchanges by device copies
  if (g->exchange == GROUP_EXCHANGE_RCCL && g->sh[0].comm) NCCL_TRY(ncclCommCount(g->sh[0].comm, &n));
  if (out_shards) *out_shards = (int)g->sh.size();
  if (out_rccl_nranks) *out_rccl_nranks = n;
  if (out_row_stride) *out_row_stride = g->cap_per_shard;
  return WDBX_OK;
} WDBX_CATCH

// counters of the group (names: "exchanges" = exchange + merge steps enqueued so far, one per chunk of a call; "dispatches" =
// jobs handed to the shards' threads; "unusable" = 1 after a failed collective aborted the communicators)
int wdbx_group_stat(wdbx_group* g, const char* name, int64_t* value) try {
  if (!g || !name || !value) return fail(WDBX_E_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(g->mu);
  if (!strcmp(name, "exchanges")) return *value = (int64_t)g->exchanges, WDBX_OK;
  if (!strcmp(name, "dispatches")) return *value = (int64_t)g->disp.dispatches, WDBX_OK;
  if (!strcmp(name, "unusable")) return *value = g->unusable ? 1 : 0, WDBX_OK;
  return fail(WDBX_E_INVALID, "unknown group statistic '%s'", name);
} WDBX_CATCH

int wdbx_group_size(wdbx_group* g, uint64_t* out_rows) try {
  if (!g || !out_rows) return fail(WDBX_E_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(g->mu);
  uint64_t total = 0;
  for (GroupShard& s : g->sh) {
    std::lock_guard<std::mutex> li(s.ix->mu);
    total += s.ix->n;
  }
  *out_rows = total;
  return WDBX_OK;
} WDBX_CATCH

// append rows; they fill shard 0 up to cap_per_shard, then shard 1, ... (contiguous global rows)
int wdbx_group_add(wdbx_group* g, const float* rows, uint64_t n, int normalize, uint64_t* first_row_out) try {
  if (!g) return fail(WDBX_E_INVALID, "null handle");
  if (n && !rows) return fail(WDBX_E_INVALID, "rows is null");
  std::lock_guard<std::mutex> lk(g->mu);
  if (!g->owns_shards) return fail(WDBX_E_STATE, "an attached group does not place rows: add them to the shard handles");
  uint64_t total = 0;
  for (GroupShard& s : g->sh) total += s.ix->n;
  if (total + n > g->cap_per_shard * g->sh.size())
    return fail(WDBX_E_INVALID, "group is full: %llu + %llu rows > %llu", (u64)total, (u64)n, (u64)(g->cap_per_shard * g->sh.size()));
  if (first_row_out) *first_row_out = total;
  uint64_t done = 0;
  while (done < n) {
    const size_t s = (size_t)((total + done) / g->cap_per_shard);
    wdbx_index* ix = g->sh[s].ix;
    const uint64_t room = g->cap_per_shard - ix->n, take = std::min(room, n - done);
    int rc = wdbx_index_add(ix, rows + (size_t)done * g->dim, take, normalize, nullptr);
    if (rc) return rc;
    done += take;
  }
  return WDBX_OK;
} WDBX_CATCH

// global row number of each shard's first row (default: shard * stride, or shard * cap_per_shard for owned groups); a
// caller that placed contiguous row ranges itself (bench.py) gives the ranges' first rows
int wdbx_group_set_row_bases(wdbx_group* g, const uint64_t* bases, int n) try {
  if (!g || !bases) return fail(WDBX_E_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(g->mu);
  if (n != (int)g->sh.size()) return fail(WDBX_E_INVALID, "%d bases for %d shards", n, (int)g->sh.size());
  for (int i = 0; i < n; ++i)
    if (bases[i] >= 0xFFFFFF00ull) return fail(WDBX_E_INVALID, "row base %llu exceeds 32-bit row keys", (u64)bases[i]);
  GroupLocks locks(g);
  for (int i = 0; i < n; ++i) {
    DeviceGuard dg(g->sh[i].ix->device);
    HIP_TRY(hipStreamSynchronize(g->sh[i].ix->stream));
    g->sh[i].ix->row_base = bases[i];
  }
  return WDBX_OK;
} WDBX_CATCH

// ---- resident queries: the group's own query buffer on every shard's device ----
int wdbx_group_queries_upload(wdbx_group* g, const float* queries, int nq, int normalize_queries) try {
  if (!g) return fail(WDBX_E_INVALID, "null handle");
  if (nq < 1 || !queries) return fail(WDBX_E_INVALID, "need at least one query");
  std::lock_guard<std::mutex> lk(g->mu);
  GroupLocks locks(g);
  return group_load_queries(g, queries, 0, 0, nq, normalize_queries);
} WDBX_CATCH

int wdbx_group_queries_synthetic(wdbx_group* g, uint64_t seed, uint64_t counter_row0, int nq, int normalize) try {
  if (!g) return fail(WDBX_E_INVALID, "null handle");
  if (nq < 1) return fail(WDBX_E_INVALID, "need at least one query");
  std::lock_guard<std::mutex> lk(g->mu);
  GroupLocks locks(g);
  return group_load_queries(g, nullptr, seed, counter_row0, nq, normalize);
} WDBX_CATCH

// asynchronous: enqueue the search of resident queries [first, first + nq) on every shard, the exchange and the merge
int wdbx_group_search_resident(wdbx_group* g, int first_query, int nq, int k, int k_out) try {
  if (!g) return fail(WDBX_E_INVALID, "null handle");
  if (nq < 0) return fail(WDBX_E_INVALID, "nq=%d", nq);
  std::lock_guard<std::mutex> lk(g->mu);
  GroupLocks locks(g);
  const int rc = group_enqueue_search(g, first_query, nq, k, k_out, false, false);
  return rc ? group_fail_drained(g, rc) : rc;
} WDBX_CATCH

int wdbx_group_synchronize(wdbx_group* g) try {
  if (!g) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(g->mu);
  for (GroupShard& s : g->sh) {
    DeviceGuard dg(s.ix->device);
    HIP_TRY(hipStreamSynchronize(s.ix->stream));
  }
  return WDBX_OK;
} WDBX_CATCH

// results [nq, k_out] of the most recent search (blocking: waits for the root shard's stream)
int wdbx_group_results(wdbx_group* g, int nq, int k_out, int64_t* out_idx, float* out_score) try {
  if (!g || !out_idx || !out_score) return fail(WDBX_E_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(g->mu);
  if (nq < 0 || nq > g->last_nq || k_out != g->last_k_out)
    return fail(WDBX_E_INVALID, "the last search left [%d, %d] results, [%d, %d] asked for", g->last_nq, g->last_k_out, nq, k_out);
  if (!nq) return WDBX_OK;
  wdbx_index* root = g->sh[0].ix;
  DeviceGuard dg(root->device);
  const size_t elems = (size_t)nq * k_out;
  HIP_TRY(hipMemcpyAsync(out_idx, g->d_oidx, elems * sizeof(int64_t), hipMemcpyDeviceToHost, root->stream));
  HIP_TRY(hipMemcpyAsync(out_score, g->d_oscore, elems * sizeof(float), hipMemcpyDeviceToHost, root->stream));
  HIP_TRY(hipStreamSynchronize(root->stream));
  return WDBX_OK;
} WDBX_CATCH

static int group_search_host(wdbx_group* g, const float* queries, int nq, int k, int k_out, int normalize_queries,
                             const uint32_t* const* masks, const uint64_t* mask_word_counts, int64_t* out_idx, float* out_score);

// every shard's top-k, merged into the k_out best of their union (k <= k_out <= shards * k): k_out = k is the plain
// search; k_out = shards * k returns the whole merged candidate list the reference's VectorStore.search sorts before
// its threshold / metadata post-filter / cut (vector_store.py:323-345).  out_idx / out_score are [nq, k_out].  Blocking.
int wdbx_group_search_merged(wdbx_group* g, const float* queries, int nq, int k, int k_out, int normalize_queries,
                             int64_t* out_idx, float* out_score) try {
  return group_search_host(g, queries, nq, k, k_out, normalize_queries, nullptr, nullptr, out_idx, out_score);
} WDBX_CATCH

// the same with a row filter per shard (metadata push-down through the group: vector_store.py:337-342 only post-filters):
// mask_words[s] = shard s's mask (uint32 words, bit r % 32 of word r / 32 = row r may be returned, ceil(rows / 32) words) or
// null for "every row of that shard"
int wdbx_group_search_merged_masked(wdbx_group* g, const float* queries, int nq, int k, int k_out, int normalize_queries,
                                    const uint32_t* const* mask_words, int64_t* out_idx, float* out_score) try {
  if (!mask_words) return fail(WDBX_E_INVALID, "mask_words is null");
  return group_search_host(g, queries, nq, k, k_out, normalize_queries, mask_words, nullptr, out_idx, out_score);
} WDBX_CATCH

// ... and with the number of words each mask holds (mask_word_counts[s]; ignored for a null mask): a mask shorter than
// ceil(rows of shard s / 32) words -- built before a concurrent add -- is refused under the locks instead of over-read
int wdbx_group_search_merged_masked_n(wdbx_group* g, const float* queries, int nq, int k, int k_out, int normalize_queries,
                                      const uint32_t* const* mask_words, const uint64_t* mask_word_counts, int64_t* out_idx,
                                      float* out_score) try {
  if (!mask_words || !mask_word_counts) return fail(WDBX_E_INVALID, "mask_words / mask_word_counts is null");
  return group_search_host(g, queries, nq, k, k_out, normalize_queries, mask_words, mask_word_counts, out_idx, out_score);
} WDBX_CATCH

static int group_search_host(wdbx_group* g, const float* queries, int nq, int k, int k_out, int normalize_queries,
                             const uint32_t* const* masks, const uint64_t* mask_word_counts, int64_t* out_idx, float* out_score) {
  if (!g) return fail(WDBX_E_INVALID, "null handle");
  if (nq < 0) return fail(WDBX_E_INVALID, "nq=%d", nq);
  if (nq == 0) return WDBX_OK;
  if (!queries || !out_idx || !out_score) return fail(WDBX_E_INVALID, "null buffer");
  if (k < 1 || k > WDBX_MAX_K) return fail(WDBX_E_INVALID, "k=%d outside [1, %d]", k, WDBX_MAX_K);
  std::lock_guard<std::mutex> lk(g->mu);
  GroupLocks locks(g);  // held until the results are on the host: the shards' own callers wait, as on any busy handle
  if (masks && mask_word_counts)
    for (size_t s = 0; s < g->sh.size(); ++s) {
      const uint64_t need = (g->sh[s].ix->n + 31) / 32;
      if (masks[s] && mask_word_counts[s] < need)
        return fail(WDBX_E_INVALID, "shard %zu: row mask of %llu words for %llu rows (%llu words needed)", s, (u64)mask_word_counts[s],
                    (u64)g->sh[s].ix->n, (u64)need);
    }
  wdbx_index* root = g->sh[0].ix;
  const size_t elems = (size_t)nq * k_out, pitch = (size_t)root->pitch, dim = (size_t)root->dim;
  int rc;
  // small calls (the facade's lone queries): queries and results through mapped host memory, no memcpy calls at all
  const bool cosine_norm = normalize_queries && root->metric == WDBX_METRIC_COSINE;
  if (g->h_stage && !cosine_norm && (size_t)nq * pitch * sizeof(float) <= GROUP_STAGE_Q && elems * sizeof(int64_t) <= GROUP_STAGE_IDX) {
    float* hq = (float*)g->h_stage;
    if (pitch == dim) {
      memcpy(hq, queries, (size_t)nq * dim * sizeof(float));
    } else {
      memset(hq, 0, (size_t)nq * pitch * sizeof(float));
      for (int q = 0; q < nq; ++q) memcpy(hq + (size_t)q * pitch, queries + (size_t)q * dim, dim * sizeof(float));
    }
    // a lone query: no repair launches are queued (two per shard); each shard leaves an overflow word instead, and the rare
    // call that finds one set is run again with the repairs in place
    volatile uint32_t* const flags = (volatile uint32_t*)(g->h_stage + GROUP_STAGE_Q + GROUP_STAGE_IDX + GROUP_STAGE_SCORE);
    const bool defer = nq == 1 && !use_select(root, k) && g->sh.size() <= 64;
    if (defer)
      for (size_t s = 0; s < g->sh.size(); ++s) flags[s] = 0;
    // (a failed enqueue: the shards that did enqueue still read the staged query and write keys, flags and results into the
    // staging area the next call reuses -- wait for them before returning the error)
    // a lone query: the group's final merge (one workgroup on the root's stream, behind the exchange and so behind every
    // shard's local stage) writes a sequence number behind its results and this thread polls it, as wdbx_index_search does
    uint32_t done_seq = 0;
    const bool poll = defer && root->opt_poll_done && g->sh[0].stage_dev;
    if (poll) {
      DeviceGuard dgr(root->device);
      if (!g->done_ev) HIP_TRY(hipEventCreateWithFlags(&g->done_ev, hipEventDisableTiming));
      if (++g->lone_seq == 0) ++g->lone_seq;
      done_seq = g->lone_seq;
      flags[64] = 0;
      root->done_flag_dev = (uint32_t*)(g->sh[0].stage_dev + GROUP_STAGE_Q + GROUP_STAGE_IDX + GROUP_STAGE_SCORE) + 64;
      root->done_seq = done_seq;
      root->done_signals = 0;
    }
    rc = group_enqueue_search(g, 0, nq, k, k_out, true, true, masks, defer);
    const bool polled = poll && root->done_signals == 1;
    root->done_flag_dev = nullptr;
    if (rc) return group_fail_drained(g, rc);
    DeviceGuard dg(root->device);
    // (the root stream's merge depends on every shard's local stage through the exchange: when it has drained, no
    // device reads the staged queries any more and the results are in host memory)
    if (polled) {
      HIP_TRY(hipEventRecord(g->done_ev, root->stream));
      for (uint32_t spins = 1;; ++spins) {
        if (flags[64] == done_seq) break;
        if ((spins & 0xFFFu) == 0) {
          const hipError_t e = hipEventQuery(g->done_ev);
          if (e == hipSuccess) break;
          if (e != hipErrorNotReady) HIP_TRY(hipEventSynchronize(g->done_ev));
        }
      }
      std::atomic_thread_fence(std::memory_order_acquire);
    } else {
      HIP_TRY(hipStreamSynchronize(root->stream));
    }
    if (defer) {
      bool over = false;
      for (size_t s = 0; s < g->sh.size(); ++s) over = over || flags[s] != 0;
      if (over) {
        if ((rc = group_enqueue_search(g, 0, nq, k, k_out, true, true, masks, false))) return group_fail_drained(g, rc);
        HIP_TRY(hipStreamSynchronize(root->stream));
      }
    }
    memcpy(out_idx, g->h_stage + GROUP_STAGE_Q, elems * sizeof(int64_t));
    memcpy(out_score, g->h_stage + GROUP_STAGE_Q + GROUP_STAGE_IDX, elems * sizeof(float));
    return WDBX_OK;
  }
  if ((rc = group_load_queries(g, queries, 0, 0, nq, normalize_queries))) return group_fail_drained(g, rc);  // (the caller's query buffer)
  if ((rc = group_enqueue_search(g, 0, nq, k, k_out, false, true, masks))) return group_fail_drained(g, rc);
  DeviceGuard dg(root->device);
  HIP_TRY(hipMemcpyAsync(out_idx, g->d_oidx, elems * sizeof(int64_t), hipMemcpyDeviceToHost, root->stream));
  HIP_TRY(hipMemcpyAsync(out_score, g->d_oscore, elems * sizeof(float), hipMemcpyDeviceToHost, root->stream));
  // (the root stream's merge depends on every shard's local stage through the exchange: when it has drained, the
  // caller's query buffer is no longer read by any device)
  HIP_TRY(hipStreamSynchronize(root->stream));
  return WDBX_OK;
}

// blocking search over all shards (k_out = k)
int wdbx_group_search(wdbx_group* g, const float* queries, int nq, int k, int normalize_queries, int64_t* out_idx,
                      float* out_score) try {
  return wdbx_group_search_merged(g, queries, nq, k, k, normalize_queries, out_idx, out_score);
} WDBX_CATCH

// host-level all-gather of small buffers through a handle's per-rank communicator (launcher-side plumbing of a
// torch-free multi-process run: barrier, max-reduction of a time, result cross-checks)
int wdbx_index_comm_allgather_host(wdbx_index* ix, const void* send, void* recv, uint64_t bytes) try {
  if (!ix || !send || !recv || !bytes) return fail(WDBX_E_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(ix->mu);
  if (!ix->comm) return fail(WDBX_E_STATE, "no communicator on this handle");
  DeviceGuard g(ix->device);
  int rc = grow((void**)&ix->d_gathered, &ix->gathered_bytes, (size_t)(ix->nranks + 1) * bytes);
  if (rc) return rc;
  char* base = (char*)ix->d_gathered;
  HIP_TRY(hipMemcpyAsync(base, send, bytes, hipMemcpyHostToDevice, ix->stream));
  NCCL_TRY(ncclAllGather(base, base + bytes, bytes, ncclUint8, ix->comm, ix->stream));
  HIP_TRY(hipMemcpyAsync(recv, base + bytes, (size_t)ix->nranks * bytes, hipMemcpyDeviceToHost, ix->stream));
  HIP_TRY(hipStreamSynchronize(ix->stream));
  return WDBX_OK;
} WDBX_CATCH

int wdbx_index_profile(wdbx_index* ix, int enable) try {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  ix->profile = enable != 0;
  return WDBX_OK;
} WDBX_CATCH

static int drain(EventPool& pool, uint64_t* count, double* ms) {
  double total = 0;
  uint64_t launches = 0;
  for (size_t i = 0; i + 1 < pool.used; i += 2) {
    float t = 0;
    HIP_TRY(hipEventElapsedTime(&t, pool.ev[i], pool.ev[i + 1]));
    total += t;
    launches += pool.launches[i / 2];
  }
  if (count) *count = launches;
  if (ms) *ms = total;
  pool.used = 0;
  return WDBX_OK;
}

int wdbx_index_profile_read(wdbx_index* ix, uint64_t* scan_launches, double* scan_ms_total, uint64_t* merge_launches,
                            double* merge_ms_total) try {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  HIP_TRY(hipStreamSynchronize(ix->stream));
  int rc = drain(ix->scan_ev, scan_launches, scan_ms_total);
  if (rc) return rc;
  return drain(ix->merge_ev, merge_launches, merge_ms_total);
} WDBX_CATCH

static const OptionDesc<wdbx_index> kOptions[] = {
    {"scan_lanes", &wdbx_index::opt_lanes},
    {"scan_blocks", &wdbx_index::opt_blocks},
    {"scan_nt", &wdbx_index::opt_nt},
    {"scan_blocked", &wdbx_index::opt_blocked},
    {"scan_generic", &wdbx_index::opt_generic},
    {"exchange_batch", &wdbx_index::opt_batch},
    {"lds_lists", &wdbx_index::opt_lds_lists},
    {"merge_fast", &wdbx_index::opt_merge_fast},
    {"poll_done", &wdbx_index::opt_poll_done},
    {"scan_one_grid", &wdbx_index::opt_scan_one_grid},
    {"zero_copy", &wdbx_index::opt_zero_copy},
    {"lone_host_select", &wdbx_index::opt_lone_host_select},
    {"wg_merge", &wdbx_index::opt_wg_merge},
    {"gemm_ct", &wdbx_index::opt_gemm_ct},
    {"gemm_l2", &wdbx_index::opt_gemm_l2},
    {"gemm_l2_i8", &wdbx_index::opt_gemm_l2_i8},
    {"gemm_bf16", &wdbx_index::opt_gemm_bf16},
    {"gemm8_variant", &wdbx_index::opt_gemm8_variant},
    {"gemm8_refine", &wdbx_index::opt_gemm8_refine},
    {"scan8_sample4", &wdbx_index::opt_scan8_sample4},
    {"scan_shadow", &wdbx_index::opt_scan_shadow},
    {"scan8_wgs", &wdbx_index::opt_scan8_wgs},
    {"scan8_per_query", &wdbx_index::opt_scan8_per_query},
    {"scan8_ablate", &wdbx_index::opt_scan8_ablate},
    {"batch_repair", &wdbx_index::opt_batch_repair},
    {"single_min_rows", &wdbx_index::opt_single_min_rows},
    {"group_bounds", &wdbx_index::opt_group_bounds},
    {"scan_force_ragged", &wdbx_index::opt_force_ragged},
    {"select_min_k", &wdbx_index::opt_select_min_k},
    {"gemm_min_queries", &wdbx_index::opt_gemm_min_nq},
    {"gemm_min_rows", &wdbx_index::opt_gemm_min_rows},
    {"gemm_min_work", &wdbx_index::opt_gemm_min_work},
    {"gemm_sample_div", &wdbx_index::opt_gemm_sample_div},
};

static int64_t* option_slot(wdbx_index* ix, const char* name) { return find_option(ix, kOptions, name); }

int wdbx_index_set_option(wdbx_index* ix, const char* name, int64_t value) try {
  if (!ix) return fail(WDBX_E_INVALID, "null handle");
  std::lock_guard<std::mutex> lk(ix->mu);
  int64_t* slot = option_slot(ix, name);
  if (!slot) return fail(WDBX_E_INVALID, "unknown option '%s'", name ? name : "(null)");
  *slot = value;
  if (!strcmp(name, "group_bounds")) ix->gmax_valid = false;  // re-decide (and rebuild the group maxima) at the next batch
  return WDBX_OK;
} WDBX_CATCH

int wdbx_index_get_option(wdbx_index* ix, const char* name, int64_t* value) try {
  if (!ix || !value) return fail(WDBX_E_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(ix->mu);
  // read-only state of the batched path
  if (name && !strcmp(name, "last_gemm_family")) return *value = ix->last_gemm_mode, WDBX_OK;
  if (name && !strcmp(name, "shadow_rows")) return *value = (int64_t)ix->shadow_rows, WDBX_OK;
  if (name && !strcmp(name, "shadow_bytes")) return *value = (int64_t)ix->rows16_bytes, WDBX_OK;
  if (name && !strcmp(name, "shadow8_rows")) return *value = (int64_t)ix->shadow8_rows, WDBX_OK;
  if (name && !strcmp(name, "shadow8_bytes")) return *value = (int64_t)(ix->rows8_bytes + ix->scale8_bytes), WDBX_OK;
  if (name && !strcmp(name, "shadowg_rows")) return *value = (int64_t)ix->shadowg_rows, WDBX_OK;
  if (name && !strcmp(name, "shadowg_bytes")) return *value = (int64_t)(ix->rows8g_bytes + ix->groups8_bytes), WDBX_OK;
  if (name && !strcmp(name, "last_single_path")) return *value = ix->last_single_path, WDBX_OK;
  if (name && !strcmp(name, "last_sample_qn")) return *value = ix->last_sample_qn, WDBX_OK;
  if (name && !strcmp(name, "last_batch_repaired")) return *value = ix->last_batch_repaired ? 1 : 0, WDBX_OK;
  if (name && !strcmp(name, "group_bounds_active")) return *value = ix->group_bounds ? 1 : 0, WDBX_OK;
  if (name && !strcmp(name, "exchanges")) return *value = (int64_t)ix->exchanges, WDBX_OK;
  if (name && !strcmp(name, "device_bytes_resident")) return *value = (int64_t)device_bytes_resident(ix), WDBX_OK;
  int64_t* slot = option_slot(ix, name);
  if (!slot) return fail(WDBX_E_INVALID, "unknown option '%s'", name ? name : "(null)");
  *value = *slot;
  return WDBX_OK;
} WDBX_CATCH

}  // extern "C"
