/*
 * wdbx_hip.h -- C ABI of the MI355X-native WDBX vector_search hot path.
 *
 * One shared library (libwdbx_hip.so, built from wdbx-py_amd/csrc/wdbx_hip.hip for
 * gfx950) replaces what the reference reaches through third-party native code on
 * the path WDBX.vector_search -> VectorStore.search -> VectorIndex.search:
 *
 *   reference interface (paths under /root/reference)            replaced by
 *   ------------------------------------------------------------------------------
 *   faiss.IndexFlatIP(dim)            wdbx/core/indexing.py:717   wdbx_index_create
 *   faiss.index_cpu_to_gpu(res,0,ix)  indexing.py:741-748         wdbx_index_create(device_id)
 *   index.add(rows[n,d])              indexing.py:890, :950       wdbx_index_add
 *   _normalize_vector at add          indexing.py:851-856,886,939 wdbx_index_add(normalize=1)
 *   hnswlib replace_vector            indexing.py:374, :431, :552 wdbx_index_set_rows
 *   index.search(q[1,d], k)           indexing.py:1013 (and :490) wdbx_index_search
 *   self.next_index / index.ntotal    indexing.py:998, :1005      wdbx_index_size
 *   _create_index() on clear          indexing.py:1098            wdbx_index_clear
 *   per-shard loop + list.sort merge  vector_store.py:323-345     wdbx_index_search_sharded_device
 *                                                                 (RCCL all-gather + merge)
 *
 * Conventions
 *   - every function returns 0 (WDBX_OK) or a negative WDBX_E_* code and never
 *     throws; wdbx_last_error() gives the calling thread's last message.
 *   - host buffers are caller-owned and only read/written during the call; device
 *     memory, streams and events are owned by the library.
 *   - rows are fp32, row-major [n, dim].  In HBM a row occupies
 *     wdbx_index_row_pitch() floats (dim rounded up to a multiple of 4, zero padded).
 *   - metric 0 = cosine as the reference does it: inner product over rows that
 *     were unit-normalised at add time; the query is normalised by the caller or
 *     with normalize_queries=1.  metric 1 = squared L2 (extension).
 *   - result order is total and deterministic: (score descending, row ascending)
 *     for cosine, (distance ascending, row ascending) for L2; unused result slots
 *     hold row -1 (like faiss, indexing.py:1023).  Rows whose score is NaN are
 *     never returned.
 *   - thread-safety: any thread may call into one handle (the reference calls search from
 *     4-worker pools, indexing.py:692, :1045-1048).  A handle mutex serialises everything that
 *     touches the handle's state, i.e. the ENQUEUE of a call's launches; a small blocking search
 *     (queries and results in one of 4 mapped staging slots) waits for the GPU on its own event
 *     with the mutex released, so concurrent callers pipeline on the handle's stream instead of
 *     taking turns at wall-clock latency.  Calls with a row mask, batched calls and ingest keep
 *     the mutex to their end.
 */
#ifndef WDBX_HIP_H
#define WDBX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WDBX_HIP_ABI_VERSION 1

#define WDBX_OK 0
#define WDBX_E_INVALID (-1)   /* bad argument */
#define WDBX_E_HIP (-2)       /* HIP runtime error (message has the hipError string) */
#define WDBX_E_NOMEM (-3)     /* device or host allocation failed */
#define WDBX_E_NODEVICE (-4)  /* no usable GPU */
#define WDBX_E_RCCL (-5)      /* RCCL error */
#define WDBX_E_STATE (-6)     /* call not valid in the handle's state */

#define WDBX_METRIC_COSINE 0
#define WDBX_METRIC_L2 1

#define WDBX_MAX_K 2048 /* largest k one search accepts */

typedef struct wdbx_index wdbx_index;

/* ---- library ---------------------------------------------------------------- */
int wdbx_hip_version(void);
const char* wdbx_last_error(void);
int wdbx_device_count(int* out_count);

/* ---- one shard = one flat index resident in one GPU's HBM -------------------- */
int wdbx_index_create(int device_id, int dim, int metric, uint64_t capacity_rows, wdbx_index** out);
void wdbx_index_destroy(wdbx_index* idx);
int wdbx_index_dim(const wdbx_index* idx);
int wdbx_index_row_pitch(const wdbx_index* idx); /* floats per stored row */
int wdbx_index_size(wdbx_index* idx, uint64_t* out_rows);
int wdbx_index_capacity(wdbx_index* idx, uint64_t* out_rows);
int wdbx_index_reserve(wdbx_index* idx, uint64_t capacity_rows); /* grow (copies rows device-to-device) */
int wdbx_index_clear(wdbx_index* idx);

/* append n host rows [n, dim]; normalize!=0 unit-normalises each row on the device
 * (a zero row stays zero, indexing.py:853-856); *first_row_out = index of the first
 * appended row (rows are numbered in append order, as faiss/next_index do). */
int wdbx_index_add(wdbx_index* idx, const float* rows, uint64_t n, int normalize, uint64_t* first_row_out);
/* overwrite stored rows [first_row, first_row+n) (replace_vector / zero-on-remove) */
int wdbx_index_set_rows(wdbx_index* idx, uint64_t first_row, const float* rows, uint64_t n, int normalize);
/* keep exactly the stored rows src_rows[0 .. n_keep) (strictly increasing row numbers), moved down to rows
 * 0 .. n_keep - 1 in that order; everything else is dropped.  The compaction behind the backend's optimize()
 * (the reference rebuilds its index there, indexing.py:1124-1149): removed rows are NaN tombstones that every scan
 * still streams.  Row order is preserved; shadow copies are rebuilt lazily from the first moved row on. */
int wdbx_index_compact(wdbx_index* idx, const uint64_t* src_rows, uint64_t n_keep);
/* read stored rows back (as stored, i.e. after normalisation) into out_rows[n, dim] */
int wdbx_index_get_rows(wdbx_index* idx, uint64_t first_row, uint64_t n, float* out_rows);
/* append n synthetic rows generated on the device: element (r, c) of counter row
 * r = counter_row0 + i is ((splitmix64(seed ^ (r*dim + c)) >> 40) - 2^23) * 2^-23
 * (BASELINE.md section 3); optional unit normalisation. */
int wdbx_index_fill_synthetic(wdbx_index* idx, uint64_t seed, uint64_t counter_row0, uint64_t n,
                              int normalize, uint64_t* first_row_out);

/* blocking search of nq host queries [nq, dim]: out_idx[nq, k] (row or -1),
 * out_score[nq, k] (inner product, or squared distance for L2).
 * Every result is the exact fp32 ranking.  How a single query is answered (option "scan_shadow"):
 *   2 (default) a selection scan over a u8 SHADOW COPY of the rows (per-row scales; +25 % device memory, built and
 *               refreshed lazily) keeps every row whose score could reach the k-th best under a rigorous
 *               quantisation bound; the kept rows are re-scored in fp32 from the fp32 rows.  Applies from
 *               65 536 rows (131 072 for a call with a single query), dim 54 ... 4096, any k, with or without a row
 *               mask; an overflowing candidate buffer is repaired on the device by the fp32 scan.
 *   1           the same selection on the bf16 tile kernel over the bf16 shadow (k < 200, no masks)
 *   0           the fp32 scan kernel
 * From "gemm_min_queries" (4) queries per call the batched path below answers them in one pass. */
int wdbx_index_search(wdbx_index* idx, const float* queries, int nq, int k, int normalize_queries,
                      int64_t* out_idx, float* out_score);

/* same with a row filter (metadata push-down; the reference only post-filters,
 * vector_store.py:337-342): bit r of mask_words (uint32 words, bit r%32 of word r/32,
 * ceil(size/32) words, host memory) says whether row r may be returned.  The result is the exact
 * top-k of the allowed rows. */
int wdbx_index_search_masked(wdbx_index* idx, const float* queries, int nq, int k, int normalize_queries,
                             const uint32_t* mask_words, int64_t* out_idx, float* out_score);
/* the same, with the number of words the mask holds: checked against the row count UNDER the handle's lock, so a mask
 * built before a concurrent add (the reference mutates its indices from pool threads, indexing.py:381-383, :407) is
 * refused with WDBX_E_INVALID instead of over-read.  What the Python binding calls. */
int wdbx_index_search_masked_n(wdbx_index* idx, const float* queries, int nq, int k, int normalize_queries,
                               const uint32_t* mask_words, uint64_t mask_word_count, int64_t* out_idx, float* out_score);

/* ---- device-resident path (inputs already in HBM; asynchronous) -------------- */
int wdbx_device_alloc(wdbx_index* idx, uint64_t bytes, void** out_dev_ptr);
int wdbx_device_free(wdbx_index* idx, void* dev_ptr);
int wdbx_device_upload(wdbx_index* idx, void* dev_dst, const void* host_src, uint64_t bytes);   /* blocking */
int wdbx_device_download(wdbx_index* idx, void* host_dst, const void* dev_src, uint64_t bytes); /* blocking */
/* fill a device query buffer [nq, row_pitch] with synthetic (optionally normalised) queries */
int wdbx_device_fill_synthetic(wdbx_index* idx, float* dev_dst, uint64_t seed, uint64_t counter_row0,
                               uint64_t n, int normalize);
/* enqueue nq single-query scans on the handle's stream and return at once.
 * d_queries is [nq, row_pitch] (zero padded beyond dim), d_out_idx [nq,k], d_out_score [nq,k]. */
int wdbx_index_search_device(wdbx_index* idx, const float* d_queries, int nq, int k,
                             int64_t* d_out_idx, float* d_out_score);
int wdbx_index_synchronize(wdbx_index* idx);

/* ---- batched queries (extension: the reference is single-query; BASELINE config 4) -- */
/* nq queries share ONE pass over the corpus, in blocks of up to 256 queries: a matrix-core pass computes
 * rows . queries^T tile by tile with a fused threshold filter, then a per-query top-k of the survivors.
 * Results are the exact fp32 ranking in every mode.  Tile family, option "gemm_bf16" (read back what the last batch
 * ran on with get_option("last_gemm_family")):
 *   3 (default) int8 tiles (v_mfma_i32_16x16x64_i8) over a GROUP-SCALED i8 SHADOW COPY of the rows (one scale per 64
 *               rows, stored in MFMA fragment order; +25 % device memory, built and refreshed lazily), only to SELECT
 *               candidates under a rigorous per-(group, query) quantisation bound; every candidate is re-scored in
 *               fp32 from the fp32 rows.  Cosine / inner product and (since round 3) L2, rows whose i8 image is at
 *               most 1536 bytes; other shapes and a shadow that does not fit fall to 2.
 *   2           bf16 tiles (v_mfma_f32_32x32x16_bf16) over a bf16 shadow copy (+50 % device memory), selection only,
 *               rigorous rounding-error margin, fp32 re-scoring.  Falls back to 1 when the shadow does not fit.
 *   1           the same bf16 selection reading the fp32 rows (no extra memory, twice the bytes per pass)
 *   0           exact fp32 MFMA tiles (v_mfma_f32_32x32x2_f32), no re-scoring for cosine
 * L2 selects by 2 c.q - |c|^2 on the pass and re-scores with the direct form.  wdbx_index_search() takes
 * this path by itself for nq >= 4 (2 on shards of 3 M rows and more) on corpora >= 65536 rows.  Asynchronous like
 * wdbx_index_search_device. */
int wdbx_index_search_batch_device(wdbx_index* idx, const float* d_queries, int nq, int k,
                                   int64_t* d_out_idx, float* d_out_score);
/* synchronises; per query the number of candidates the filter kept (out_counts[nq], may be null),
 * the buffer capacity, and how many queries exceeded it.  Since round 3 such queries are re-run exactly ON THE DEVICE by
 * conditional launches queued behind their block (get_option "last_batch_repaired" == 1; option "batch_repair"): the
 * results are exact either way and the count only tells how many needed it.  Shapes without a device-side repair (k in
 * the radix-select range) leave it to the caller as before (wdbx_index_search() does it by itself). */
int wdbx_index_batch_status(wdbx_index* idx, uint32_t* out_counts, int nq, uint32_t* out_capacity,
                            int* out_overflowed);
int wdbx_index_profile_read_gemm(wdbx_index* idx, uint64_t* launches, double* ms_total);
/* the sample launches of the single-query selection scan (one launch per round of up to 32 queries) */
int wdbx_index_profile_read_sample(wdbx_index* idx, uint64_t* launches, double* ms_total);

/* ---- shards across GPUs: one process per GPU, RCCL over xGMI ----------------- */
#define WDBX_UNIQUE_ID_BYTES 128
int wdbx_comm_unique_id(void* out_128_bytes); /* rank 0 creates, the host side distributes */
/* join the shard group; global_row_base = number of rows held by lower ranks
 * (contiguous row ranges, so the merged order equals the single-shard order). */
int wdbx_index_comm_init(wdbx_index* idx, int nranks, int rank, const void* unique_id_128_bytes,
                         uint64_t global_row_base);
int wdbx_index_comm_destroy(wdbx_index* idx);
/* what RCCL reports for the handle's communicator (ncclCommCount / ncclCommUserRank; 0 / -1 without one) and the
 * handle's global row base -- evidence for a scaling run that the exchange really spans N ranks */
int wdbx_index_comm_info(wdbx_index* idx, int* out_nranks, int* out_rank, uint64_t* out_row_base);
/* re-base this rank's rows (after the shard was cleared and refilled with another row range) without tearing the
 * communicator down */
int wdbx_index_comm_set_row_base(wdbx_index* idx, uint64_t global_row_base);
/* as wdbx_index_search_device, but every rank scans its own shard, the per-shard
 * (row, score) records are all-gathered with RCCL and merged on every rank:
 * identical global results on all ranks; rows are global row numbers. */
int wdbx_index_search_sharded_device(wdbx_index* idx, const float* d_queries, int nq, int k,
                                     int64_t* d_out_idx, float* d_out_score);
/* the same exchange around the batched MFMA path (every rank must call it with the same nq, k; every
 * rank's shard must be eligible: cosine, >= 65536 rows).  wdbx_index_batch_status reports this rank's
 * candidate overflow as for the unsharded call. */
int wdbx_index_search_sharded_batch_device(wdbx_index* idx, const float* d_queries, int nq, int k,
                                           int64_t* d_out_idx, float* d_out_score);
/* host-level all-gather of `bytes` bytes per rank through the handle's communicator (recv: nranks * bytes, rank order):
 * the launcher-side plumbing of a multi-process run without any other transport -- barrier, max-reduction of a time,
 * result cross-checks.  Blocking. */
int wdbx_index_comm_allgather_host(wdbx_index* idx, const void* send, void* recv, uint64_t bytes);

/* ---- shards across GPUs in ONE process (the reference's VectorStore(num_shards=S) shape,
 *      vector_store.py:111-134, :323-345): S flat indices, contiguous row ranges (global row r lives in shard
 *      r / cap_per_shard).  Every shard's launches are enqueued by the shard's own persistent host thread; the per-shard
 *      (row, score) key lists are exchanged with ONE ncclAllGather per shard (communicators from ncclCommInitAll) and
 *      merged on the first shard's device.  Shards that share a device cannot be RCCL ranks: such a group exchanges
 *      by device-to-device copies instead (same results; wdbx_group_info reports 0 RCCL ranks). ---- */
typedef struct wdbx_group wdbx_group;
int wdbx_group_create(const int* device_ids, int n, int dim, int metric, uint64_t cap_per_shard, wdbx_group** out);
void wdbx_group_destroy(wdbx_group* grp);
/* append rows: they fill shard 0 up to cap_per_shard, then shard 1, ... ; *first_row_out = global row */
int wdbx_group_add(wdbx_group* grp, const float* rows, uint64_t n, int normalize, uint64_t* first_row_out);
int wdbx_group_size(wdbx_group* grp, uint64_t* out_rows);
/* blocking: every shard scans its rows, per-shard (row, score) key lists are exchanged and merged on the first
 * shard's device; out_idx holds global rows.  Identical to a single-shard search. */
int wdbx_group_search(wdbx_group* grp, const float* queries, int nq, int k, int normalize_queries,
                      int64_t* out_idx, float* out_score);
/* The same fan-out over EXISTING shard handles -- the reference's VectorStore keeps one index object per shard
 * (vector_store.py:111-134) and loops over them (:323-327): the handles stay owned by the caller and keep growing
 * through wdbx_index_add; the group only adds the exchange.  In merged results shard s owns the row numbers
 * [s * stride, (s + 1) * stride), stride = (2^32 - 256) / n (wdbx_group_info): row = stride * shard + local row, and ties
 * come back in shard order = the order of the reference's stable sort (:330); wdbx_group_set_row_bases replaces that
 * numbering.  wdbx_group_add is not valid on such a group; wdbx_group_destroy leaves the handles alive.  While a group
 * call enqueues it holds every shard's handle mutex, and it works on buffers of its own, so the shards' own callers
 * (wdbx_index_search on the same handles from other threads) can run concurrently with it.
 * exchange_mode: 0 = RCCL when every shard has its own device and the communicators come up, device copies otherwise
 * (also: environment WDBX_GROUP_EXCHANGE=rccl|copy); 1 = RCCL or fail; 2 = device copies. */
int wdbx_group_attach(wdbx_index* const* shards, int n, wdbx_group** out);
int wdbx_group_attach_ex(wdbx_index* const* shards, int n, int exchange_mode, wdbx_group** out);
/* *out_rccl_nranks: what ncclCommCount says about the group's communicator; 0 = the group exchanges by device copies */
int wdbx_group_info(wdbx_group* grp, int* out_shards, int* out_rccl_nranks, uint64_t* out_row_stride);
/* counters of the group: "exchanges" = exchange (all-gather / device copies) + merge steps enqueued so far -- a call is cut
 * into chunks and each chunk has ONE, so a caller can say how many a timed region held; "dispatches" = jobs handed to the
 * shards' threads; "unusable" = 1 after a failed collective made the group abort its communicators */
int wdbx_group_stat(wdbx_group* grp, const char* name, int64_t* value);
/* global row number of each shard's first row (a caller that placed contiguous row ranges itself) */
int wdbx_group_set_row_bases(wdbx_group* grp, const uint64_t* bases, int n);
/* every shard's top-k, exchanged and merged into the k_out best of their union, k <= k_out <= min(shards * k,
 * WDBX_MAX_K); out_idx / out_score are [nq, k_out].  k_out = shards * k is the whole candidate list the reference sorts
 * before its threshold / metadata post-filter / cut (vector_store.py:329-345), so a post-filtered query sees exactly
 * the candidates the reference would.  Blocking. */
int wdbx_group_search_merged(wdbx_group* grp, const float* queries, int nq, int k, int k_out, int normalize_queries,
                             int64_t* out_idx, float* out_score);
/* the same with a row filter per shard (metadata push-down, SURVEY 8f row 2, through the group): mask_words[s] = shard s's
 * mask as in wdbx_index_search_masked (ceil(rows of that shard / 32) uint32 words, host memory), or null = every row */
int wdbx_group_search_merged_masked(wdbx_group* grp, const float* queries, int nq, int k, int k_out, int normalize_queries,
                                    const uint32_t* const* mask_words, int64_t* out_idx, float* out_score);
/* the same, with mask_word_counts[s] = words held by mask_words[s] (ignored for a null mask): a short mask is refused under
 * the group's locks (vector_store.py:337-342 filters a list that cannot change under it; a pushed-down mask can go stale) */
int wdbx_group_search_merged_masked_n(wdbx_group* grp, const float* queries, int nq, int k, int k_out, int normalize_queries,
                                      const uint32_t* const* mask_words, const uint64_t* mask_word_counts, int64_t* out_idx,
                                      float* out_score);
/* (wdbx_group_search_merged answers a call that carries enough queries -- 4 on shards of >= 65536 rows -- with ONE batched
 * matrix-core pass per shard, as wdbx_index_search does; the resident form below makes one scan per query on every shard,
 * as wdbx_index_search_device does.)
 * device-resident form (inputs already in HBM; asynchronous): queries are placed once in the group's query buffer on
 * EVERY shard's device (from the host, or generated there like wdbx_device_fill_synthetic); search_resident enqueues the
 * search of queries [first_query, first_query + nq) on all shards + exchange + merge and returns; the results [nq, k_out]
 * of the most recent search stay on the first shard's device until wdbx_group_results copies them out. */
int wdbx_group_queries_upload(wdbx_group* grp, const float* queries, int nq, int normalize_queries);
int wdbx_group_queries_synthetic(wdbx_group* grp, uint64_t seed, uint64_t counter_row0, int nq, int normalize);
int wdbx_group_search_resident(wdbx_group* grp, int first_query, int nq, int k, int k_out);
int wdbx_group_synchronize(wdbx_group* grp);
int wdbx_group_results(wdbx_group* grp, int nq, int k_out, int64_t* out_idx, float* out_score);

/* ---- measurement ------------------------------------------------------------- */
/* enable!=0: bracket every scan-kernel launch with HIP events on the handle's stream */
int wdbx_index_profile(wdbx_index* idx, int enable);
/* synchronises, then reports and resets: number of scan launches measured and their
 * summed duration (ms); likewise for the merge kernels. */
int wdbx_index_profile_read(wdbx_index* idx, uint64_t* scan_launches, double* scan_ms_total,
                            uint64_t* merge_launches, double* merge_ms_total);
/* measurement aid: time `reps` plain streaming reads of the stored rows (16 B per lane, no
 * arithmetic, no top-k) -- the read ceiling on this device that the scan kernel is compared with */
int wdbx_index_probe_read(wdbx_index* idx, int nontemporal, int blocks, int reps, double* out_ms_per_pass);
/* tuning knobs (name/value); unknown names return WDBX_E_INVALID.  Settable: scan_lanes, scan_blocks, scan_nt,
 * scan_blocked, scan_generic, scan_force_ragged, exchange_batch, lds_lists, merge_fast (1: merges whose keys fit the registers are ranked there, default; 0: always the list walk), scan_one_grid (1: a round of several queries on the fp32 scan over a corpus of at most 1 GiB is one grid with a row per query, default; 0: a launch per query), poll_done (1: a blocking call of up to 32 queries whose chain ends in a final merge polls a word that kernel writes into the mapped staging slot, default; 0: always waits on its event), zero_copy, wg_merge, select_min_k,
 * scan_shadow (2 u8 selection scan / 1 bf16 tiles / 0 fp32 scan), scan8_wgs, single_min_rows, gemm_bf16 (tile family
 * 3/2/1/0 as above), gemm_ct, gemm_l2, gemm_l2_i8, gemm8_variant, gemm8_refine (1: second selection stage of the i8 tiles, default), batch_repair, scan8_per_query, scan8_sample4 (1: a round's sample pass serves 3-4 queries per workgroup when the sample is too large for the L2s; 2: always; 0: never), gemm_min_queries, gemm_min_rows, gemm_min_work (below gemm_min_rows: the tiles from queries x rows >= this, default 800000; 0: never), gemm_sample_div, group_bounds.
 * get_option also answers the read-only names: last_gemm_family (0/1/2/3: what the last batch ran on),
 * last_single_path (0 fp32 scan / 1 bf16 tiles / 2 u8 selection scan), last_sample_qn (queries per workgroup of the last u8 sample launch: 1, 3 or 4), last_batch_repaired, shadow_rows + shadow_bytes (bf16 copy),
 * shadow8_rows + shadow8_bytes (u8 copy), shadowg_rows + shadowg_bytes (group-scaled i8 copy), group_bounds_active,
 * exchanges (all-gather + merge steps this handle's per-rank communicator has enqueued), device_bytes_resident (every device
 * allocation of the handle: fp32 rows, shadow copies and their tables, scratch). */
int wdbx_index_set_option(wdbx_index* idx, const char* name, int64_t value);
int wdbx_index_get_option(wdbx_index* idx, const char* name, int64_t* value);

#ifdef __cplusplus
}
#endif
#endif /* WDBX_HIP_H */
