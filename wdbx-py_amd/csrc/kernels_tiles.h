// kernels_tiles.h -- batched queries on the matrix cores: exact fp32 tiles, bf16 selection tiles, their shared epilogue and conversions.
// Part of the single translation unit wdbx_hip.hip (included there, in order); not a standalone header.

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
  // Rigorous selection bounds, per 64-row group and query (used by the GROUPB instances of the kernels only):
  //   |selection score - true score| <= bound(g, q) = eps * sqrt(gmax[g]) * qn[q]            (inner product)
  //                                                   2 eps * sqrt(gmax[g]) * qn[q] + gam * gmax[g]   (L2 form 2 c.q - |c|^2)
  // gmax[g] = largest squared norm among rows 64 g .. 64 g + 63, qn[q] = |q|.  PHASE 0 reports LOWER bounds (score -
  // bound), PHASE 1 keeps every row whose UPPER bound reaches the threshold, so the k-th largest reported value is
  // itself a valid threshold and one outlier row only loosens the bound of its own group.
  const float* gmax;
  const float* qn;
  float eps, gam;
  float floor_abs;     // absolute term: operands in the denormal range lose their RELATIVE precision when rounded to bf16
                       // (or are flushed); each costs at most 2^-126 per product, covered by floor_abs * (|c| + |q|)
};

// Tile epilogue shared by the fp32 and bf16 tile kernels.  acc holds the wave's 64 rows x 32*CT queries in
// the 32x32 MFMA C layout: query = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// RW = 64-row wave groups per tile (tile rows = 64 * RW); PHASE 0 leaves one key per (query, tile, wave group).
// L2: cn_pref = squared norm of row (wave's first row + lane), loaded by the caller before the tile's K loop (so the
// load's latency hides behind the loop), cn_wave = this wave's 64 floats of LDS to hand the norms to the lanes that
// need them -- 32 broadcast ds_reads instead of 32 global loads held in as many registers.
template <int PHASE, int CT, int METRIC, int RW = 2, bool GROUPB = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, f16v (&acc)[2][CT], const float (&thr)[CT], uint32_t t,
                                              uint32_t trow0, int rh, int ch, int l31, int lh, float cn_pref = 0.f,
                                              float* cn_wave = nullptr, float gm_pref = 0.f) {
  // epilogue: C layout of 32x32: query = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  // (the lane's query column is made opaque here: otherwise the per-query addresses of the rare append path and of
  // the norm hand-over are computed once at kernel entry and held -- spilled -- across the whole tile loop)
  asm volatile("" : "+v"(l31), "+v"(lh));
  const uint32_t wrow0 = trow0 + rh * 64;
  const bool partial = trow0 + 64 * RW > a.n_rows;
  if constexpr (METRIC == WDBX_METRIC_L2) {
    // lane i holds the norm of the wave's row i: every register's row norm comes by a lane read (ds_bpermute: the LDS
    // crossbar, no LDS memory and no per-lane LDS address to keep alive across the tile loop)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float cn = __shfl(cn_pref, rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct][r] = fmaf(2.0f, acc[rt][ct][r], -cn);
      }
  }
  // bound(ct) = bscale * qn[query of ct] + bconst (two registers; the query norms are re-read per column tile)
  // (GROUPB is a template parameter: the default kernels, whose bound is one global margin applied to the thresholds,
  // carry none of this)
  float bscale = 0.f, bconst = 0.f;
  if constexpr (GROUPB) {
    const float g = gm_pref;  // = a.gmax[this wave's 64-row group], loaded by the caller before the K loop
    const float sg = sqrtf(g), two = METRIC == WDBX_METRIC_L2 ? 2.0f : 1.0f;
    bscale = two * (a.eps * sg + a.floor_abs);
    bconst = (METRIC == WDBX_METRIC_L2 ? a.gam * g : 0.f) + two * a.floor_abs * sg;
  }
  auto bound = [&](int ct) -> float {
    if constexpr (GROUPB) return fmaf(bscale, a.qn[ch * (32 * CT) + ct * 32 + l31], bconst);
    return 0.f;
  };
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
      m = fmaxf(m, __shfl_xor(m, 32)) - bound(ct);  // lower bound of the group's best true score
      if (!(m == m)) m = -INFINITY;                 // (an infinite bound: the group vouches for nothing)
      const uint32_t q = ch * (32 * CT) + ct * 32 + l31, ht = t * RW + rh;
      if (lh == 0 && (a.live == 0 || q < a.live)) {
        a.halfmax[(size_t)q * (RW * a.num_tiles) + ht] = (m == -INFINITY) ? 0ull : make_key(m + 0.0f, ht);
      }
    }
  } else {
    if constexpr (GROUPB) {
      // A group that holds a row with an INFINITE norm (an infinite element, or finite elements beyond 1.8e19): its
      // selection scores say nothing (inf * 0, inf - inf; bf16 roundings clamped to the largest finite value) while
      // the exact fp32 score can be anything -- every row of the group goes to the exact pass, for every live query.
      // (Such rows force the per-group instances: the global bound's statistics see the infinite norm.  Rows with a
      // NaN element have a NaN norm, score NaN against every query and are never results: they need nothing here.)
      if (!(gm_pref < INFINITY)) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const uint32_t q = ch * (32 * CT) + ct * 32 + l31;
          if (!(thr[ct] < INFINITY)) continue;  // padded or idle column
#pragma unroll
          for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const uint32_t row = wrow0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
              if (row < a.n_rows) {
                const uint32_t pos = atomicAdd(&a.count[q], 1u);
                if (pos < a.cap) a.cand[(size_t)q * a.cap + pos] = make_key(INFINITY, row);
              }
            }
        }
        return;
      }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const uint32_t q = ch * (32 * CT) + ct * 32 + l31;
      const float cut = thr[ct] - bound(ct);  // score + bound >= thr  <=>  score >= cut
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        float m = acc[rt][ct][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[rt][ct][r]);
        if (m >= cut) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[rt][ct][r];
            const uint32_t row = wrow0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (v >= cut && row < a.n_rows) {
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
template <int PHASE, bool KTAIL, int CT, int METRIC, bool GROUPB = false>
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
  float* const cn_wave = Bs + 2 * GBN * LD + (threadIdx.x >> 6) * 64;  // L2: the wave's 64 row norms (epilogue)
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
    float cn_pref = 0.f, gm_pref = 0.f;
    if constexpr (METRIC == WDBX_METRIC_L2) cn_pref = a.cn[min(trow0 + rh * 64 + lane, last_row)];
    if constexpr (GROUPB) gm_pref = a.gmax[min(trow0 + rh * 64, last_row) >> 6];
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

    gemm_epilogue<PHASE, CT, METRIC, 2, GROUPB>(a, acc, thr, t, trow0, rh, ch, l31, lh, cn_pref, cn_wave, gm_pref);
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

// fp32 -> bf16 for SELECTION operands: round to nearest even, but a FINITE value never becomes infinite (values
// beyond the largest finite bf16, 3.3895e38, are clamped to it: still within the 2^-8 relative error the bounds
// assume).  NaN stays NaN, infinities stay infinite.
__device__ __forceinline__ __bf16 to_bf16_sel(float x) {
  const float lim = 3.3895313892515355e38f;  // (2 - 2^-7) * 2^127
  if (fabsf(x) <= 3.4028235e38f) x = fminf(fmaxf(x, -lim), lim);
  return (__bf16)x;
}

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
constexpr int TILE_PAD_ROWS = 256;  // rows of slack behind every row allocation the 8-wave tiles read (fp32 rows, shadows)

template <int PHASE, bool KTAIL, int CT, int METRIC, bool SHADOW, bool GROUPB = false>
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
  float* const cn_wave = (float*)(Bs + 2 * GBN * LDB) + (threadIdx.x >> 6) * 64;  // L2: the wave's 64 row norms (epilogue)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rh = wave & 3, ch = wave >> 2, l31 = lane & 31, lh = lane >> 5;
  // chunks per row, rounded up to a pair (the surplus chunk is zeros on both sides)
  const uint32_t kchunks = ((a.pitch4 + QPR - 1) / QPR + 1) / 2 * 2;

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
  // row addresses = the tile's first row (wave-uniform: scalar registers) + this thread's piece offset inside the tile +
  // a uniform step per staged row.  Rows past the end of the corpus are READ (the host pads every row allocation by
  // TILE_PAD_ROWS rows for this) and their scores are masked in the epilogue: no per-row clamping, one address register.
  const f4* tile_base = a.rows;
  const uint32_t po0 = srow * a.pitch4;
  const f4* const pb = (const f4*)a.qb16 + (size_t)brow * a.qb_pitch16 + bpiece;
  const size_t pb_step = (size_t)BRP * a.qb_pitch16;
  uint32_t ld_tile = blockIdx.x, ld_kc = 0;  // row loader cursor (tile, chunk)
  uint32_t lb_kc = 0;                         // query loader cursor (chunk; the queries are the same for every tile)
  auto set_tile = [&](uint32_t tile) {
    const uint32_t r0 = tile * a.tile_stride * GW_M;
    tile_base = a.rows + (size_t)r0 * a.pitch4;
  };
  auto gload_a = [&](int slot, int j) {
    const uint32_t kq = ld_kc * QPR + squad;
    // K tail (fp32 rows only): pieces past the row end re-read the row's last piece and are zeroed
    f4 v = __builtin_nontemporal_load(tile_base + (size_t)(j * ARP) * a.pitch4 + (po0 + (KTAIL ? min(kq, a.pitch4 - 1) : kq)));
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
      h[0] = to_bf16_sel(v.x);
      h[1] = to_bf16_sel(v.y);
      h[2] = to_bf16_sel(v.z);
      h[3] = to_bf16_sel(v.w);
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
    float cn_pref = 0.f, gm_pref = 0.f;
    // (row of this lane inside the wave's 64 = 32 * lh + l31, from the values the epilogue keeps alive anyway)
    if constexpr (METRIC == WDBX_METRIC_L2) cn_pref = a.cn[min(t * a.tile_stride * GW_M + rh * 64 + lh * 32 + l31, last_row)];
    for (uint32_t kc = 0; kc < kchunks; kc += 2) {
      body(std::integral_constant<int, 0>{});
      body(std::integral_constant<int, 1>{});
    }
    // (the group's largest norm is read after the K loop: one L2-resident word per tile, not a register held across it)
    if constexpr (GROUPB) gm_pref = a.gmax[min(t * a.tile_stride * GW_M + rh * 64, last_row) >> 6];
    // the queries' thresholds are (re)loaded per tile, behind an opaque copy of the lane's column so that they are
    // not held in registers across the K loop (4 L2 hits per tile against 4 registers of a full register file)
    float thr[CT];
    if constexpr (PHASE == 1) {
      int col = l31;
      asm volatile("" : "+v"(col));
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const uint32_t q = ch * (32 * CT) + ct * 32 + col;
        thr[ct] = (a.live == 0 || q < a.live) ? a.tau[q] : INFINITY;
      }
    }
    gemm_epilogue<PHASE, CT, METRIC, 4, GROUPB>(a, acc, thr, t, t * a.tile_stride * GW_M, rh, ch, l31, lh, cn_pref, cn_wave, gm_pref);
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
    for (int i = 0; i < 8; ++i) h[i] = to_bf16_sel((c + i < pitch) ? rows[r * pitch + c + i] : 0.0f);
    *(bh8*)(out + r * pitch16 + c) = h;
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
    out[e] = to_bf16_sel((r < live && src < nv && c < pitch) ? q[(size_t)src * pitch + c] : 0.0f);
  }
}
