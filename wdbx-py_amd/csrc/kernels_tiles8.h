// kernels_tiles8.h -- batched queries on the int8 matrix cores over a group-scaled i8 shadow copy (inner product / cosine).
// Part of the single translation unit wdbx_hip.hip (included there, in order); not a standalone header.

// ------------------------------------------------------------------------------------------------
// int8 SELECTION tiles (default batched path for inner product / cosine, rows up to 1536 bytes of i8 shadow).
// The bf16 selection tiles of kernels_tiles.h read 2 bytes per element and run v_mfma_f32_32x32x16_bf16; this kernel
// reads ONE byte per element and runs v_mfma_i32_16x16x64_i8 (twice the rate per clock).  Like every selection path
// it only SELECTS: exact fp32 re-scoring of the kept rows follows.
//
// Structure (profiles/r02/c4_i8_design_notes.md has the numbers of the forms measured on the way here):
//   * the QUERY BLOCK (up to 256 queries as signed bytes, 96 KiB at 384 bytes per row) is RESIDENT in LDS for the whole
//     launch, XOR-swizzled so the fragment reads are conflict-free;
//   * the ROWS never touch LDS: each of the 8 waves owns 32 rows of a 256-row tile (two 16-row halves against all 16
//     column groups of 16 queries = 128 accumulator registers), and its A fragments are laid out in HBM in fragment
//     order, so a wave streams them with plain 1 KiB coalesced 16-byte loads straight into a REGISTER RING a few k-steps
//     (about half a tile, 1 us) ahead of the matrix ops -- the streaming shape of the scan kernels;
//   * consequently there is NO barrier and no LDS-DMA in the main loop: waves drift freely, one wave's LDS / memory
//     waits are the other's matrix time.  (Ring-in-LDS forms: a barrier per 16 matrix ops per wave cost 0.22 ms of a 1.2 ms
//     pass and kept the SIMD partners in lockstep, so fragment-read latency and matrix time added up instead of overlapping.)
//   * the 16x16x64 shape, not 32x32x32: the same integer ops per cycle, but on random bytes the chip holds ~2.0 GHz under
//     the former and ~1.7 GHz under the latter (tools/probes/mfma_shape_clock.hip: 4.05 vs 3.48 POP/s in bare loops), and
//     this kernel is matrix-pipe / power bound as much as HBM bound.
//
// Shadow copy G (rows_to_i8g_kernel): rows as SIGNED bytes n = rint(c / s_g), one scale s_g per 64-ROW GROUP, so that a
// wave's integer dot products D = sum n_i m_i are directly comparable and the whole tile epilogue is integer compares
// against ONE threshold per query.  Queries are quantised the same way per query (queries_to_i8_kernel: m = rint(q / s_q)).
// With c_i = s_g (n_i + delta_i), q_i = s_q (m_i + eps_i):
//     c.q - s_g s_q D = s_g s_q sum (n_i eps_i + delta_i m_i + delta_i eps_i)
//     |c.q - s_g s_q D| <= a_r E_q + b_r M_q,   a_r = s_g |n_r|_2,  b_r = s_g |delta_r|_2,
//                                                E_q = s_q |eps|_2,  M_q = s_q (|m|_2 + |eps|_2)      (Cauchy-Schwarz)
// with delta, eps the ACTUAL residuals (computed at quantisation; |.|_2 <= sqrt(d)/2 but typically sqrt(d/12)).  The
// group table holds {s_g, a_g = max a_r, b_g = max b_r, vouch} per 64 rows (all rounded up).
//   PHASE 0 (sampled tiles): per (query, 32-row block) the LOWER bound s_g s_q Dmax - (a_g E_q + b_g M_q) of the block's
//            best true score; the k-th largest of them is a valid threshold tau (groups holding a non-finite row do not vouch).
//   PHASE 1 (all tiles): every row with s_g s_q D + a_g E_q + b_g M_q >= tau, i.e. D >= T(group, query), is appended.
// Rows with an infinite element cannot be quantised: their group has a_g = +inf and all its rows go to the exact
// pass.  Rows with a NaN element (removed rows) quantise to zeros; their exact score is NaN and is never a result.
//
// L2 (METRIC = 1) ranks by -|c - q|^2 = 2 c.q - |c|^2 - |q|^2, per query by  v(c) = 2 c.q - |c|^2.  The integer dot
// product bounds c.q as above, |c|^2 is the row's cached squared fp32 norm (taken 1e-4 relative to the safe side):
//   PHASE 0: max over the block's rows of  2 s_g s_q D_r - |c_r|^2,  minus twice the bound: a lower bound of the block's best v;
//   PHASE 1: keep row r iff 2 (s_g s_q D + bound) - |c_r|^2 >= tau  <=>  D - u_r w_q >= T,  T the inner-product threshold at
//            tau / 2,  u_r = |c_r|^2 / (2 s_g) per row (8 per lane and tile),  w_q = 1 / s_q per query: one convert and one
//            fma per accumulator more than the inner-product form.
// The kept rows are re-scored with the direct form sum (c - q)^2 (rescore_kernel), as on every L2 selection path.
//
// Layout of shadow copy G in HBM: FRAGMENT ORDER.  Block b (rows 32 b .. 32 b + 31) is pitch8 / 64 consecutive k-steps of
// 2 KiB; k-step s holds bytes 64 s .. 64 s + 63 of the block's rows as two 1 KiB MFMA A fragments (rows 0-15, rows 16-31):
// lane l = 16 kb + r of fragment h owns the 16 bytes [64 s + 16 kb, +16) of row 16 h + r, at offset 16 l.  (Any k order
// inside a k-step works as long as rows and queries use the same one: both fragments take bytes [64 s + 16 kb, +16).)
// ------------------------------------------------------------------------------------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

constexpr int G8_ROWS = 256;                        // rows per workgroup tile: 8 waves x 32 rows
constexpr uint32_t PAIR_D_UNKNOWN = 0x800000u;     // a pair's 24-bit dot-product field: "not representable" (= -2^23)
constexpr int G8_LDS_B_MAX = 96 * 1024;             // resident query block (the rest of the LDS: the queries' parameters)

// byte offset of (row r, column col) in the fragment-ordered shadow copy
__host__ __device__ __forceinline__ size_t g8_offset(u64 r, uint32_t col, uint32_t pitch8) {
  return (size_t)(r >> 5) * 32 * pitch8 + (size_t)(col >> 6) * 2048 + (size_t)((r >> 4) & 1) * 1024 +
         ((((col >> 4) & 3u) << 4) + (uint32_t)(r & 15)) * 16 + (col & 15u);
}

struct Gemm8Args {
  const int8_t* rows8;   // fragment-ordered blocks (see above), whole tiles
  const f4* groups;      // [ceil(rows / 64) + pad] {s_g, a_g, b_g, vouch}
  const float* cn;       // L2: squared fp32 norm per row
  const float* gref;     // {a_ref, b_ref}: bounds that hold for every ORDINARY group (group_ref_kernel); the prefilter epilogue
  const u64* gbad;       // [groups] bit r set = row r of the 64-row group holds a NaN (a removed row) or lies past the end:
                         // such a row is never a result and must not vouch for a threshold (PHASE 0)
  const int8_t* qb8;     // query block [32 * CT8][pitch8] signed bytes, zero padded (rows and columns)
  const f4* qpar;        // [32 * CT8] {s_q, E_q, M_q, 1 / s_q}; padded queries: all zero
  uint32_t n_rows, pitch8;
  uint32_t num_tiles, tile_stride;
  u64* halfmax;          // PHASE 0: [32 * CT8][8 * num_tiles]
  const float* tau;      // PHASE 1: [32 * CT8] (+inf for padded queries)
  // PHASE 1: every wave appends its candidates as (D24 << 40 | query << 32 | row) pairs to a list of ITS OWN -- plain stores at
  // positions from a wave-level prefix sum, no atomic whose return the row stream would have to be drained for;
  // scatter_pairs_kernel sorts them into the per-query candidate buffers afterwards.
  u64* pairs;            // [gridDim.x * 8 waves][pair_cap]
  uint32_t* pair_count;  // [gridDim.x * 8]: pairs each wave produced (beyond pair_cap: dropped, the call is flagged)
  uint32_t pair_cap;
};

// position (in 16-byte pieces) of piece c of query row r inside its LDS row: an XOR swizzle that makes the MFMA
// fragment reads (16 rows x the same piece per ds_read_b128 lane group) conflict-free for both parities of pitch8/128
__device__ __forceinline__ uint32_t g8_bswz(uint32_t c, uint32_t r, bool odd) {
  return odd ? ((c & ~7u) | ((c & 7u) ^ ((r >> 1) & 7u))) : ((c & ~15u) | ((c & 15u) ^ (r & 15u)));
}

// CT8 = 32-query units per wave (8, 4 or 2: query blocks of 256, 128, 64 = 2 CT8 column groups of 16); RING = k-steps
// (64 bytes of every row: two A fragments) in flight per wave, a divisor of pitch8 / 64; PITCH8 = the rows' bytes as a
// compile-time constant (384, 768: every LDS read address is then one of a few per-lane registers plus an immediate, no
// address arithmetic in the loop) or 0 = run time.
// VAR: experiment switches (option gemm8_variant; 0 = the product form):
//   bit 0: row stream with the default cache policy instead of non-temporal
//   bit 1: SIMD partners (waves w, w + 4) start half a tile apart (measured: no gain)
//   bit 2: TIMING ONLY, wrong answers: no tile epilogue
//   bit 5: the tile epilogue inside the next tile's first k-step instead of a block of its own (measured: slower)
//   bit 8: (PHASE 1, inner product) PREFILTER epilogue: per lane and column group a launch-constant U = A1 - a_ref E' - b_ref M'
//          (16 registers, paid for by a query-fragment window of 4); on a tile whose group is ordinary (a_g <= a_ref,
//          b_g <= b_ref) the test "max of 8 accumulators >= e_inv U - 1" needs no LDS read and one fma, and is never stricter
//          than the exact one, which runs -- unchanged -- only for the column groups that pass it
//   bit 3: TIMING ONLY: the row stream is not read inside the loop;  bit 4: TIMING ONLY: no query-fragment reads inside the loop
//   bit 10: TIMING ONLY: the row stream re-reads the workgroup's FIRST tile (an L2-resident stream: the vector-memory path without HBM)
//   bit 6: (PHASE 1) the tile epilogue as ONE straight-line block + one branch: all NJ (threshold, max of 8 accumulators,
//          compare) in a row with the hit masks kept in scalar registers, then -- in about two tiles of three -- the rows
//          of the column groups that had a hit.  (The product form branches per column group: each of its 16 blocks
//          waits out its own LDS read of the query parameters.)
template <int PHASE, int CT8, int RING, int PITCH8 = 0, int VAR = 0, int METRIC = WDBX_METRIC_COSINE>
__global__ __launch_bounds__(512) void gemm_i8_kernel(Gemm8Args a) {
  constexpr bool L2 = METRIC == WDBX_METRIC_L2;
  constexpr int GBN = 32 * CT8, NJ = 2 * CT8;  // NJ column groups of 16 queries
  constexpr bool PRE = PHASE == 1 && (VAR & 256) != 0 && METRIC == WDBX_METRIC_COSINE;
  constexpr int WMAX = ((VAR & 128) || PRE) ? 4 : 8;  // (bit 7, experiment: a window of 4 register sets instead of 8)
  constexpr int W = NJ < WMAX ? NJ : WMAX;     // query fragments in flight (a rolling window over the (k-step, group) sequence)
  extern __shared__ __attribute__((aligned(16))) char lds8[];
  const uint32_t pitch8 = PITCH8 ? (uint32_t)PITCH8 : a.pitch8;
  char* const Bs = lds8;                                       // [GBN][pitch8], pieces swizzled (g8_bswz)
  f4* const qp = (f4*)(lds8 + (size_t)GBN * pitch8);           // [GBN] the queries' parameters
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kb = lane >> 4;
  const uint32_t P = pitch8 / 16, steps = pitch8 / 64;
  const bool odd = ((pitch8 / 128) & 1) != 0;
  if (blockIdx.x >= a.num_tiles) return;

  // ---- the query block and its parameters: global -> LDS once ----
  for (uint32_t e = tid; e < (uint32_t)GBN * P; e += 512) {
    const uint32_t r = e / P, c = e - r * P;
    *(i32x4*)(Bs + (size_t)r * pitch8 + 16 * g8_bswz(c, r, odd)) = *(const i32x4*)(a.qb8 + (size_t)r * pitch8 + 16 * c);
  }
  for (uint32_t q = tid; q < (uint32_t)GBN; q += 512) {
    const f4 p = a.qpar[q];
    if constexpr (PHASE == 1) {
      // {A1, E', M', padded}: the threshold and the error norms in units of s_q, with the roundings' slack folded in
      // (see the epilogue).  A zero or non-finite query has s_q = 0 and all-zero bytes (every D = 0): its values stay
      // unscaled, which keeps "all rows or none" conservative.  Padded queries: A1 = +inf, never a candidate.
      // L2: the same at tau / 2 (v = 2 c.q - |c|^2), and the query's 1 / s_q, a little LOW, in the last slot for the rows'
      // norm term (padded queries are then told by A1 = +inf)
      const float tau = L2 ? 0.5f * a.tau[q] : a.tau[q];
      const bool padded = !(tau < INFINITY);
      const float w = p.w == 0.f ? 1.f : p.w;
      const float A = tau * w;
      qp[q] = f4{padded ? INFINITY : A - 2e-6f * fabsf(A), p.y * w * 1.000003f, p.z * w * 1.000003f,
                 L2 ? w * 0.999997f : (padded ? 1.f : 0.f)};
    } else {
      qp[q] = p;
    }
  }
  __syncthreads();

  // ---- B fragment read addresses: row = query 16 j + l15 of column group j, piece c = 4 s + kb, swizzled ----
  //   address(s, j) = l15 * pitch8 + 16 * ((4 s & ~mask) + ((4 s & mask) ^ bg)) + j * 16 * pitch8
  // (the swizzle key of row 16 j + l15 depends on l15 only; ((x | kb) ^ g) == x ^ (kb ^ g) for x a multiple of 4)
  const uint32_t bmask = odd ? 7u : 15u;
  const uint32_t bg = (odd ? (((uint32_t)l15 >> 1) & 7u) : (uint32_t)l15) ^ (uint32_t)kb;
  const uint32_t b_row0 = (uint32_t)l15 * pitch8;
  // compile-time pitch: the swizzled part takes NV values (one register each, per set of CPG column groups whose offsets
  // fit the 16-bit immediate); everything else is an immediate
  constexpr int NV = PITCH8 ? ((((PITCH8 / 128) & 1) ? 8 : 16) / 4) : 1;
  constexpr int P8 = PITCH8 ? PITCH8 : 64;  // (keeps the constant expressions below defined in the run-time form)
  constexpr int CPG = PITCH8 ? ((49152 / (16 * P8)) > 0 ? (49152 / (16 * P8)) : 1) : 1;
  constexpr int NSET = PITCH8 ? (NJ + CPG - 1) / CPG : 1;
  uint32_t voff[NSET][NV];
  if constexpr (PITCH8 != 0) {
#pragma unroll
    for (int g = 0; g < NSET; ++g)
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        uint32_t v = b_row0 + 16 * ((uint32_t)(4 * i) ^ bg) + (uint32_t)g * CPG * 16 * PITCH8;
        asm volatile("" : "+v"(v));  // (opaque: see b_read)
        voff[g][i] = v;
      }
  }
  // the fragment of k-step `step`, column group j.  (The query block never changes and the addresses repeat tile after
  // tile: were they transparent, the compiler would hoist ALL the block's fragments out of the tile loop, into registers it
  // does not have.)
  auto b_read = [&](uint32_t step, int j, uint32_t dyn_base) -> i32x4 {
    if constexpr (PITCH8 != 0) {
      const uint32_t piece = 4 * step;
      const uint32_t imm = 16 * (piece & ~bmask) + (uint32_t)(j % CPG) * 16 * PITCH8;
      return *(const i32x4*)(lds8 + voff[j / CPG][(piece & bmask) / 4] + imm);
    } else {
      return *(const i32x4*)(lds8 + dyn_base + j * 16 * pitch8);
    }
  };
  auto b_base = [&](uint32_t step) -> uint32_t {  // run-time pitch only
    if constexpr (PITCH8 != 0) return 0;
    const uint32_t piece = 4 * step;                                              // wave-uniform
    uint32_t addr = b_row0 + 16 * ((piece & ~bmask) + ((piece & bmask) ^ bg));
    asm volatile("" : "+v"(addr));
    return addr;
  };

  // ---- the row stream: this wave's block of the tile, k-step after k-step (2 KiB each), RING k-steps ahead ----
  const size_t blk_bytes = (size_t)32 * pitch8;
  const uint32_t gdim = gridDim.x;
  uint32_t ld_tile = blockIdx.x, ld_s = 0;   // loader cursor
  const int8_t* ld_p = a.rows8 + ((size_t)ld_tile * a.tile_stride * 8 + wave) * blk_bytes + lane * 16;
  // bit 9: the stream through BUFFER loads -- a descriptor of this wave's block of the tile in 4 scalar registers (rebuilt per
  // tile on the scalar unit), ONE vector register of lane offset, the k-step as scalar offset + immediate -- instead of a
  // 64-bit per-lane pointer and its carries: the registers that buys are what a ring of 6 k-steps needs to fit without scratch
  constexpr bool BUF = (VAR & 512) != 0;
  auto block_rsrc = [&](uint32_t tile) {
    const u64 blk = (u64)__builtin_amdgcn_readfirstlane(tile * a.tile_stride * 8 + wave);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(a.rows8 + blk * blk_bytes), (short)0, (int)blk_bytes, 0x00020000);
  };
  __amdgpu_buffer_rsrc_t ld_rsrc = block_rsrc(ld_tile);
  const uint32_t lane_off = (uint32_t)lane * 16;
  i32x4 ring[RING][2];
  auto load_next = [&](int j) {
    if constexpr (BUF) {
      ring[j][0] = (i32x4)__builtin_amdgcn_raw_buffer_load_b128(ld_rsrc, lane_off, ld_s * 2048, 2);          // (aux 2 = nt)
      ring[j][1] = (i32x4)__builtin_amdgcn_raw_buffer_load_b128(ld_rsrc, lane_off, ld_s * 2048 + 1024, 2);
      if (++ld_s == steps) {
        ld_s = 0;
        if (ld_tile + gdim < a.num_tiles) ld_tile += gdim;  // past the last tile: re-read it (valid memory, never used)
        ld_rsrc = block_rsrc(ld_tile);
      }
      return;
    }
    if constexpr (VAR & 1) {
      ring[j][0] = *(const i32x4*)ld_p;
      ring[j][1] = *(const i32x4*)(ld_p + 1024);
    } else {
      ring[j][0] = __builtin_nontemporal_load((const i32x4*)ld_p);
      ring[j][1] = __builtin_nontemporal_load((const i32x4*)(ld_p + 1024));
    }
    ld_p += 2048;
    if (++ld_s == steps) {
      ld_s = 0;
      // past the last tile: re-read it (valid memory, never used)
      if (!(VAR & 1024) && ld_tile + gdim < a.num_tiles) ld_tile += gdim;
      ld_p = a.rows8 + ((size_t)ld_tile * a.tile_stride * 8 + wave) * blk_bytes + lane * 16;
    }
  };
#pragma unroll
  for (int j = 0; j < RING; ++j) load_next(j);

  i32x4 acc[NJ][2];  // [column group][row half]: rows 16 h + 4 kb + i, query 16 j + l15
  // this lane's query parameters: qpl[16 j] (one address register + immediates; transparent, the compiler keeps NJ addresses)
  uint32_t qpl_off = (uint32_t)GBN * pitch8 + (uint32_t)l15 * 16;
  asm volatile("" : "+v"(qpl_off));
  const f4* const qpl = (const f4*)(lds8 + qpl_off);
  const uint32_t wave_id = blockIdx.x * 8 + wave;
  uint32_t npairs = 0;  // wave-uniform

  // The query fragments run W positions ahead of the matrix ops in the (k-step, column group) sequence, in a rolling window of
  // W register sets: the fragment W positions on is read into bf[j % W] right behind the two matrix ops that consumed it,
  // so every read has 2 (W - 1) matrix ops (and the SIMD partner's) to come back.  (Left to itself the compiler reads each
  // fragment one matrix op ahead of its use into two ping-pong registers: every matrix op then waits out the LDS latency.)
  i32x4 bf[W];
  uint32_t tb = b_base(0);
#pragma unroll
  for (int j = 0; j < W; ++j) bf[j] = b_read(0, j, tb);

  if constexpr ((VAR & 2) != 0) {
    if (wave >= 4)
      for (uint32_t i = 0; i < steps * NJ / 4; ++i) __builtin_amdgcn_s_sleep(1);  // steps * NJ matrix ops of 16 cycles = half a tile alone on the pipe
  }

  // ---- the tile epilogue, one column group at a time: this wave's 32 rows x 16 queries, one scale for all of them ----
  // e_* = the state of the tile whose accumulators are in the registers (wave-uniform).  Product form: a block of its own
  // behind the tile's last k-step.  (VAR bit 5 runs group j's epilogue inside the NEXT tile's first k-step, right before the
  // two matrix ops that restart group j's accumulators, so that its vector instructions issue under matrix ops: the branches
  // cut that k-step into 16 blocks and it came out slower, 0.82 vs 0.79 ms.)
  bool e_have = false;
  uint32_t e_wrow0 = 0, e_ht = 0;
  f4 e_gt = {0.f, 0.f, 0.f, 0.f};
  float e_inv = 0.f, e_ai = 0.f, e_bi = 0.f;
  // L2: per tile, this lane's 8 rows' norm terms.  PHASE 0: |c_r|^2 taken high.  PHASE 1: u_r = |c_r|^2 / (2 s_g) taken low;
  // a NaN norm (a removed row: never a result) becomes +inf = never kept, a norm that overflowed -inf = always kept.
  float e_u[L2 ? 8 : 1];
  uint32_t e_bad = 0;  // PHASE 0: the bad-row bits of this wave's 32 rows (wave-uniform)
  // PHASE 1: the lanes' kept rows (bits: which of the lane's 8 rows of column group j) go to the wave's pair list, one
  // ballot per trip (a lane rarely holds more than one).  A pair carries its integer dot product D in 24 bits
  // (|D| < 2^23 holds up to d = 520 at full-scale bytes; anything outside is stored as "unknown"): refine_pairs_kernel
  // turns it into the row's own score bounds without reading the row again.
  auto append_pairs = [&](uint32_t bits, uint32_t q, int j, uint32_t lrow0) __attribute__((always_inline)) {
    for (u64 mask = __ballot(bits != 0); mask; mask = __ballot(bits != 0)) {
      const uint32_t at = npairs + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1));
      if (bits) {
        const uint32_t r = (uint32_t)__builtin_ctz(bits);
        bits &= bits - 1;
        int d = acc[j][0][0];
#pragma unroll
        for (int rr = 1; rr < 8; ++rr) d = (r == (uint32_t)rr) ? acc[j][rr >> 2][rr & 3] : d;
        const uint32_t d24 = (d > -(1 << 23) && d < (1 << 23)) ? ((uint32_t)d & 0xFFFFFFu) : PAIR_D_UNKNOWN;
        if (at < a.pair_cap)
          a.pairs[(size_t)wave_id * a.pair_cap + at] = ((u64)d24 << 40) | ((u64)q << 32) | (lrow0 + 16 * (r >> 2) + (r & 3));
      }
      npairs += (uint32_t)__builtin_popcountll(mask);
    }
  };
  auto epilogue = [&](int j) {
    const uint32_t lrow0 = e_wrow0 + 4 * kb;  // this lane's rows: lrow0 + 16 h + i
    if constexpr (PHASE == 0) {
      uint32_t q = (uint32_t)l15;
      asm volatile("" : "+v"(q));  // (computed here, not kept in NJ registers across the launch)
      q += j * 16;
      const f4 p = qpl[j * 16];  // {s_q, E, M, 1 / s_q}
      float lb;
      bool none;
      if constexpr (L2) {
        const float ss = 2.0f * e_gt.x * p.x;
        float best = -INFINITY;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float d = (float)acc[j][r >> 2][r & 3];
          float v = fmaf(ss, d, -e_u[r]) - 8e-7f * fabsf(ss * d);  // (rounded down; a NaN norm makes it NaN: not taken)
          if ((e_bad >> (4 * kb + 16 * (r >> 2) + (r & 3))) & 1u) v = -INFINITY;
          best = (v > best) ? v : best;
        }
        best = fmaxf(best, __shfl_xor(best, 16));
        best = fmaxf(best, __shfl_xor(best, 32));
        none = best == -INFINITY;
        lb = best - 2.0f * (e_gt.y * p.y + e_gt.z * p.z) * 1.000001f;
      } else {
        int m = INT_MIN;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          int v = acc[j][r >> 2][r & 3];
          if ((e_bad >> (4 * kb + 16 * (r >> 2) + (r & 3))) & 1u) v = INT_MIN;  // a removed row, or past the end
          m = max(m, v);
        }
        m = max(m, __shfl_xor(m, 16));
        m = max(m, __shfl_xor(m, 32));
        none = m == INT_MIN;
        const float w = e_gt.x * p.x * (float)m;
        lb = w - (e_gt.y * p.y + e_gt.z * p.z) * 1.000001f - 4e-7f * fabsf(w);
      }
      // lower bound of the block's best true score (rounded down); groups with a non-finite row, blocks with no valid
      // row and infinite bounds vouch for nothing
      if (!(e_gt.w == 1.0f) || none || !(lb == lb)) lb = -INFINITY;
      if (kb == 0) a.halfmax[(size_t)q * (8 * a.num_tiles) + e_ht] = (lb == -INFINITY) ? 0ull : make_key(lb + 0.0f, e_ht);
    } else if constexpr (L2) {
      const f4 p = qpl[j * 16];  // {A1 (at tau / 2), E', M', ~1 / s_q}; padded <=> A1 = +inf
      const float T = fmaf(-e_bi, p.z, fmaf(-e_ai, p.y, fmaf(e_inv, p.x, -1.0f)));
      float f[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float d = (float)acc[j][r >> 2][r & 3];
        f[r] = fmaf(-e_u[r], p.w, d) + 2e-6f * fabsf(d);  // D - u_r w_q, a little HIGH (never fewer candidates)
      }
      // (NaN-aware maximum: a NaN -- an infinite bound meeting an infinite norm -- must reach the test, which keeps it)
      float m = f[0];
#pragma unroll
      for (int r = 1; r < 8; ++r) m = (f[r] > m || f[r] != f[r]) ? f[r] : m;
      const bool hit = !(m < T);
      if (__any(hit)) {
        uint32_t bits = 0, q = (uint32_t)l15;
        asm volatile("" : "+v"(q));
        q += j * 16;
        if (hit && p.x < INFINITY) {
#pragma unroll
          for (int r = 0; r < 8; ++r)
            if (!(f[r] < T) && lrow0 + 16 * (r >> 2) + (r & 3) < a.n_rows) bits |= 1u << r;
        }
        append_pairs(bits, q, j, lrow0);
      }
    } else {
      const f4 p = qpl[j * 16];  // {A1, E', M', padded}
      const float T = fmaf(-e_bi, p.z, fmaf(-e_ai, p.y, fmaf(e_inv, p.x, -1.0f)));
      int m = max(max(acc[j][0][0], acc[j][0][1]), max(acc[j][0][2], acc[j][0][3]));
      m = max(m, max(max(acc[j][1][0], acc[j][1][1]), max(acc[j][1][2], acc[j][1][3])));
      const bool hit = !((float)m < T);
      if (__any(hit)) {  // (wave-uniform: the ballots below need every lane.)  Rare: which of the lane's 8 rows, one per trip
        uint32_t bits = 0, q = (uint32_t)l15;
        asm volatile("" : "+v"(q));  // (computed here, not kept in NJ registers across the launch)
        q += j * 16;
        if (hit && p.w == 0.f) {
#pragma unroll
          for (int r = 0; r < 8; ++r)
            if (!((float)acc[j][r >> 2][r & 3] < T) && lrow0 + 16 * (r >> 2) + (r & 3) < a.n_rows) bits |= 1u << r;
        }
        append_pairs(bits, q, j, lrow0);
      }
    }
  };
  constexpr bool FUSED = (VAR & 32) != 0;   // (bit 5; measured slower than the epilogue as a block of its own: 0.82 vs 0.79 ms)
  constexpr bool NO_EPI = (VAR & 4) != 0;
  // ---- PRE: the launch-constant part of the prefilter threshold, per lane and column group ----
  float U[PRE ? NJ : 1];
  float g_aref = 0.f, g_bref = 0.f;
  if constexpr (PRE) {
    g_aref = *(const __attribute__((address_space(4))) float*)(a.gref);
    g_bref = *(const __attribute__((address_space(4))) float*)(a.gref + 1);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const f4 p = qpl[j * 16];  // {A1, E', M', padded}; A1 = +inf for padded queries: U = +inf, never a hit
      const float u = fmaf(-g_bref, p.z, fmaf(-g_aref, p.y, p.x));
      U[j] = u - 4e-6f * (fabsf(p.x) + g_aref * p.y + g_bref * p.z);  // (low by more than the two chains' roundings can differ)
    }
  }
  constexpr bool EPI1 = PHASE == 1 && (VAR & 64) != 0 && !L2 && !PRE;
  auto load_norm_terms = [&]() {  // L2: once per tile, before its epilogue
    if constexpr (L2) {
      const uint32_t lrow0 = e_wrow0 + 4 * kb;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const uint32_t row = lrow0 + 16 * (r >> 2) + (r & 3);
        const float cn = a.cn[row < a.n_rows ? row : a.n_rows - 1];
        if constexpr (PHASE == 0) e_u[r] = cn * 1.0001f;
        else e_u[r] = (cn != cn) ? INFINITY : (cn < INFINITY ? 0.5f * cn * 0.9999f * e_inv : -INFINITY);
      }
    }
  };
  auto epilogue_block = [&]() {  // (bit 6) every column group's test first, one branch, then the groups that had a hit
    const uint32_t lrow0 = e_wrow0 + 4 * kb;
    u64 hm[NJ], any = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const f4 p = qpl[j * 16];  // {A1, E', M', padded}
      const float T = fmaf(-e_bi, p.z, fmaf(-e_ai, p.y, fmaf(e_inv, p.x, -1.0f)));
      int m = max(max(acc[j][0][0], acc[j][0][1]), max(acc[j][0][2], acc[j][0][3]));
      m = max(m, max(max(acc[j][1][0], acc[j][1][1]), max(acc[j][1][2], acc[j][1][3])));
      hm[j] = __ballot(!((float)m < T) && p.w == 0.f);
      any |= hm[j];
    }
    if (any) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (!hm[j]) continue;
        const f4 p = qpl[j * 16];
        const float T = fmaf(-e_bi, p.z, fmaf(-e_ai, p.y, fmaf(e_inv, p.x, -1.0f)));  // (the same chain: the same value)
        uint32_t bits = 0, q = (uint32_t)l15;
        asm volatile("" : "+v"(q));
        q += j * 16;
        if ((hm[j] >> lane) & 1) {
#pragma unroll
          for (int r = 0; r < 8; ++r)
            if (!((float)acc[j][r >> 2][r & 3] < T) && lrow0 + 16 * (r >> 2) + (r & 3) < a.n_rows) bits |= 1u << r;
        }
        append_pairs(bits, q, j, lrow0);
      }
    }
  };

  for (uint32_t t = blockIdx.x; t < a.num_tiles; t += gridDim.x) {
    // RING k-steps per trip (ring slots static); the first k-step of a tile starts the accumulators from zero
    auto trip = [&](auto first_tag, uint32_t s0) {
#pragma unroll
      for (int jj = 0; jj < RING; ++jj) {
        const uint32_t sn = s0 + jj + 1 == steps ? 0 : s0 + jj + 1;                 // the next k-step (of the next tile at the end)
        const uint32_t tbn = b_base(sn);
        const i32x4 af0 = ring[jj][0], af1 = ring[jj][1];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (decltype(first_tag)::value && jj == 0) {
            if constexpr (FUSED && !NO_EPI)
              if (e_have) epilogue(j);  // the previous tile's column group j, before its accumulators restart
            const i32x4 zero = {0, 0, 0, 0};
            acc[j][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af0, bf[j % W], zero, 0, 0, 0);
            acc[j][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af1, bf[j % W], zero, 0, 0, 0);
          } else {
            acc[j][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af0, bf[j % W], acc[j][0], 0, 0, 0);
            acc[j][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af1, bf[j % W], acc[j][1], 0, 0, 0);
          }
          if constexpr ((VAR & 16) != 0) asm volatile("" : "+v"(bf[j % W]));
          else if (j + W < NJ) bf[j % W] = b_read(s0 + jj, j + W, tb);
          else bf[j % W] = b_read(sn, j + W - NJ, tbn);
        }
        // (keep that order: one fragment read behind each pair of matrix ops, not all reads in a clump behind the last one;
        // the k-step that carries the epilogue is cut into blocks by its branches and keeps program order anyway)
        if (!(decltype(first_tag)::value && jj == 0 && FUSED && !NO_EPI)) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
          }
          __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);    // VMEM read: the ring's loads stay HERE (not sunk to their use)
        }
        if constexpr ((VAR & 8) != 0) asm volatile("" : "+v"(ring[jj][0]), "+v"(ring[jj][1]));
        else load_next(jj);  // the k-step RING further down the stream takes the slot just consumed
        tb = tbn;
      }
    };
    if constexpr (PITCH8 != 0) {  // (fully unrolled: every k-step index is a constant)
      trip(std::true_type{}, 0);
#pragma unroll
      for (uint32_t s0 = RING; s0 < (uint32_t)(PITCH8 / 64); s0 += RING) trip(std::false_type{}, s0);
    } else {
      trip(std::true_type{}, 0);
      for (uint32_t s0 = RING; s0 < steps; s0 += RING) trip(std::false_type{}, s0);
    }

    if constexpr (NO_EPI) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(acc[j][0]), "v"(acc[j][1]));
      continue;
    }
    // this tile's accumulators are complete: its epilogue state
    e_wrow0 = (t * a.tile_stride * 8 + wave) * 32;
    e_ht = t * 8 + wave;
    {  // the group's {s_g, a_g, b_g, vouch}: a wave-uniform address, read on the scalar path
      const uint32_t gidx = __builtin_amdgcn_readfirstlane(e_wrow0 >> 6);
      e_gt = *(const __attribute__((address_space(4))) f4*)(a.groups + gidx);
      if constexpr (PHASE == 0) {
        const u64 bad = *(const __attribute__((address_space(4))) u64*)(a.gbad + gidx);
        e_bad = (uint32_t)(bad >> (e_wrow0 & 32u));
      }
    }
    if constexpr (PHASE == 1) {
      // keep row r for query q iff s_g s_q D + a_g E + b_g M >= tau  <=>  D >= (tau / s_q - (a_g E + b_g M) / s_q) / s_g.
      // T is that value taken a little LOW (more candidates, never fewer): minus 2e-6 of the magnitudes involved (the fp32
      // roundings of this chain, also when tau and the bound nearly cancel; folded into A1, E', M' per query) minus one unit
      // of D; three fmas per (group, query).  D is compared as fp32 (exact below 2^24; beyond, its rounding is inside the
      // slack).  T = NaN (an infinite bound) keeps everything.
      // An all-zero or vanishing group (1 / s_g overflows; every D is 0) is "all rows or none" by sign: the same chain with
      // 2^60 in place of 1 / s_g keeps the group for the queries with tau <= bound.
      const float rcp = __builtin_amdgcn_rcpf(e_gt.x);
      e_inv = rcp < INFINITY ? rcp : 0x1p60f;
      e_ai = e_gt.y * e_inv;
      e_bi = e_gt.z * e_inv;
    }
    e_have = true;
    load_norm_terms();
    if constexpr (EPI1) {
      epilogue_block();
      e_have = false;
    } else if constexpr (PRE) {
      // (wave-uniform; a NaN or infinite bound compares false: the exact epilogue for every column group)
      const bool ordinary = __builtin_amdgcn_readfirstlane((e_gt.y <= g_aref && e_gt.z <= g_bref) ? 1 : 0) != 0;
      if constexpr ((VAR & 64) != 0) {  // (experiment: all NJ prefilter tests in a row, one branch, then the groups that passed)
        if (ordinary) {
          u64 hm[NJ], any = 0;
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            int m = max(max(acc[j][0][0], acc[j][0][1]), max(acc[j][0][2], acc[j][0][3]));
            m = max(m, max(max(acc[j][1][0], acc[j][1][1]), max(acc[j][1][2], acc[j][1][3])));
            hm[j] = __ballot(!((float)m < fmaf(e_inv, U[PRE ? j : 0], -1.0f)));
            any |= hm[j];
          }
          if (any) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              if (hm[j]) epilogue(j);
          }
        } else {
#pragma unroll
          for (int j = 0; j < NJ; ++j) epilogue(j);
        }
      } else
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        bool go = !ordinary;
        if (ordinary) {
          int m = max(max(acc[j][0][0], acc[j][0][1]), max(acc[j][0][2], acc[j][0][3]));
          m = max(m, max(max(acc[j][1][0], acc[j][1][1]), max(acc[j][1][2], acc[j][1][3])));
          go = __any(!((float)m < fmaf(e_inv, U[PRE ? j : 0], -1.0f)));
        }
        if (go) epilogue(j);
      }
      e_have = false;
    } else if constexpr (!FUSED) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) epilogue(j);
      e_have = false;
    }
  }
  if constexpr (FUSED && !NO_EPI) {
    if (e_have) {  // the last tile's epilogue
#pragma unroll
      for (int j = 0; j < NJ; ++j) epilogue(j);
    }
  }
  if constexpr (PHASE == 1)
    if (lane == 0) a.pair_count[wave_id] = npairs;
}

// ------------------------------------------------------------------------------------------------
// shadow copy G: one workgroup per 64-row group.  Pass 1: the group's largest finite |element| over its finite rows
// -> s_g = max / 127.  Pass 2: every row as signed bytes n = rint(c / s_g) (non-finite rows: zeros), its |n|_2 and
// residual |delta|_2, and the group's maxima a_g = s_g max |n|_2, b_g = s_g max |delta|_2 (rounded up).
//   rows past n_rows inside the last group are written as zero rows.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rows_to_i8g_kernel(const float* rows, u64 g0, u64 g1, u64 n_rows, uint32_t dim, uint32_t pitch,
                                                          int8_t* out, uint32_t pitch8, f4* groups, u64* gbad) {
  __shared__ float red[4][4];
  __shared__ uint32_t redbad[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (u64 grp = g0 + blockIdx.x; grp < g1; grp += gridDim.x) {
    // pass 1: wave w takes rows 16 w .. 16 w + 15 of the group
    float mx = 0.f;
    bool any_inf = false, any_bad = false;
    uint32_t wbad = 0;  // this wave's 16 rows: NaN rows and rows past the end
    for (int i = 0; i < 16; ++i) {
      const u64 r = grp * 64 + wave * 16 + i;
      if (r >= n_rows) {
        wbad |= 0xFFFFu & ~((1u << i) - 1u);
        break;
      }
      const float* p = rows + r * pitch;
      float rmx = 0.f;
      bool nan = false, inf = false;
      for (uint32_t c = lane; c < dim; c += 64) {
        const float v = p[c];
        nan = nan || (v != v);
        inf = inf || (fabsf(v) > 3.4028235e38f);
        rmx = fmaxf(rmx, fabsf(v));
      }
      nan = __any(nan);
      inf = __any(inf);
      for (int o = 32; o > 0; o >>= 1) rmx = fmaxf(rmx, __shfl_xor(rmx, o));
      if (nan) wbad |= 1u << i;
      if (nan || inf) any_bad = true;       // the row cannot be quantised: zeros
      else mx = fmaxf(mx, rmx);
      if (inf && !nan) any_inf = true;      // ... and its exact score can be finite or infinite: the group goes to the exact pass
    }
    __syncthreads();
    if (lane == 0) {
      red[wave][0] = mx;
      red[wave][1] = any_inf ? 1.f : 0.f;
      red[wave][2] = any_bad ? 1.f : 0.f;
      redbad[wave] = wbad;
    }
    __syncthreads();
    mx = fmaxf(fmaxf(red[0][0], red[1][0]), fmaxf(red[2][0], red[3][0]));
    const bool g_inf = (red[0][1] + red[1][1] + red[2][1] + red[3][1]) > 0.f;
    (void)any_bad;  // (NaN rows: excluded row by row through the bad-row bits)
    // (a vanishing maximum would overflow 127 / max: everything quantises to 0 and the scale is widened so that the
    // residual bound still covers the rows: delta = c / s_g with |delta| <= 1/2)
    const bool vanishing = mx < 1.2e-30f;
    const float s_g = vanishing ? 2.0f * mx : mx / 127.0f;
    const float inv = vanishing ? 0.f : 127.0f / mx;
    // pass 2
    float n2max = 0.f, d2max = 0.f;
    for (int i = 0; i < 16; ++i) {
      const u64 r = grp * 64 + wave * 16 + i;
      const bool inside = r < n_rows;
      const float* p = rows + (inside ? r : 0) * pitch;
      bool bad = false;
      if (inside) {
        for (uint32_t c = lane; c < dim; c += 64) {
          const float v = p[c];
          bad = bad || !(fabsf(v) <= 3.4028235e38f);
        }
        bad = __any(bad);
      }
      float n2 = 0.f, d2 = 0.f;
      for (uint32_t c = lane; c < pitch8; c += 64) {
        float x = 0.f, res = 0.f;
        if (inside && !bad && c < dim) {
          const float v = p[c];
          x = fminf(fmaxf(rintf(v * inv), -127.f), 127.f);
          res = vanishing ? (mx > 0.f ? 0.5f : 0.f) : v * inv - x;  // (vanishing: n = 0, |delta| = |c| / (2 max) <= 1/2)
        }
        out[g8_offset(r, c, pitch8)] = (int8_t)(int)x;
        n2 = fmaf(x, x, n2);
        d2 = fmaf(res, res, d2);
      }
      for (int o = 32; o > 0; o >>= 1) {
        n2 += __shfl_xor(n2, o);
        d2 += __shfl_xor(d2, o);
      }
      n2max = fmaxf(n2max, n2);
      d2max = fmaxf(d2max, d2);
    }
    __syncthreads();
    if (lane == 0) {
      red[wave][0] = n2max;
      red[wave][1] = d2max;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      n2max = fmaxf(fmaxf(red[0][0], red[1][0]), fmaxf(red[2][0], red[3][0]));
      d2max = fmaxf(fmaxf(red[0][1], red[1][1]), fmaxf(red[2][1], red[3][1]));
      // rounded up: fp32 summation of d terms, the products v * inv (relative 2^-24 of up to 127 per element), s_g vs 1 / inv
      const float a_g = g_inf ? INFINITY : s_g * sqrtf(n2max) * 1.0002f;
      const float b_g = s_g * (sqrtf(d2max) * 1.0002f + 2e-5f * sqrtf((float)dim));
      // (NaN rows are excluded row by row through the bad-row bits; a group holding an INFINITE row vouches for nothing)
      groups[grp] = f4{s_g, a_g, b_g, g_inf ? 0.f : 1.f};
      gbad[grp] = (u64)redbad[0] | ((u64)redbad[1] << 16) | ((u64)redbad[2] << 32) | ((u64)redbad[3] << 48);
    }
  }
}

// queries [nv, pitch] fp32 -> the i8 query block [gbn][pitch8] (signed bytes, zero padded) and its parameters
// {s_q, E_q, M_q, 1 / s_q} (E, M rounded up; a non-finite query gets E = +inf: every row becomes a candidate and the exact
// pass decides); one wave per query
// Also the block's small initialisations, so that no memset launch sits in front of a batch: tau = +inf for every column
// (the threshold merge overwrites the real queries'; padded ones never append), the block's candidate counters and its
// "a wave lost pairs" flag = 0.
__global__ __launch_bounds__(256) void queries_to_i8_kernel(const float* q, uint32_t dim, uint32_t pitch, uint32_t nv, int8_t* out,
                                                            uint32_t pitch8, uint32_t gbn, f4* qpar, float* tau_init, uint32_t* count_zero,
                                                            uint32_t* lost_zero) {
  const int lane = threadIdx.x & 63;
  const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blockIdx.x == 0) {
    for (uint32_t i = threadIdx.x; i < gbn; i += 256) count_zero[i] = 0u;
    if (threadIdx.x == 0) *lost_zero = 0u;
  }
  if (r >= gbn) return;
  if (lane == 0) tau_init[r] = INFINITY;
  const float* p = q + (size_t)r * pitch;
  const bool real = r < nv;
  float mx = 0.f;
  bool bad = false;
  if (real)
    for (uint32_t c = lane; c < dim; c += 64) {
      const float v = p[c];
      bad = bad || !(fabsf(v) <= 3.4028235e38f);
      mx = fmaxf(mx, fabsf(v));
    }
  bad = __any(bad);
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  const bool vanishing = mx < 1.2e-30f;
  const float s_q = (!real || bad) ? 0.f : vanishing ? 2.0f * mx : mx / 127.0f;
  const float inv = (!real || bad || vanishing) ? 0.f : 127.0f / mx;
  float m2 = 0.f, e2 = 0.f;
  for (uint32_t c = lane; c < pitch8; c += 64) {
    float x = 0.f, res = 0.f;
    if (real && !bad && c < dim) {
      const float v = p[c];
      x = fminf(fmaxf(rintf(v * inv), -127.f), 127.f);
      res = vanishing ? (mx > 0.f ? 0.5f : 0.f) : v * inv - x;
    }
    out[(size_t)r * pitch8 + c] = (int8_t)(int)x;
    m2 = fmaf(x, x, m2);
    e2 = fmaf(res, res, e2);
  }
  for (int o = 32; o > 0; o >>= 1) {
    m2 += __shfl_xor(m2, o);
    e2 += __shfl_xor(e2, o);
  }
  if (lane == 0) {
    const float eps = sqrtf(e2) * 1.0002f + 2e-5f * sqrtf((float)dim), mm = sqrtf(m2) * 1.0002f;
    // E also absorbs the tiny mismatch between the scales used in the score (s = max / 127) and in the residuals (1 / inv)
    float E = s_q * (eps + 1e-6f * mm), M = s_q * (mm + eps);
    if (real && bad) E = INFINITY;
    qpar[r] = f4{s_q, E, M, s_q > 0.f ? 1.0f / s_q : 0.f};
  }
}

// the waves' candidate pairs -> the per-query candidate buffers the exact pass reads: cand[q][count[q]++] = key(row).
// One workgroup per 16 producing waves.  A wave that ran out of room (a tile whose every row is a candidate for every
// query fills 8192 pairs) scatters what it kept and raises *lost: some query lost candidates, nobody knows which, so
// mark_lost_kernel -- AFTER the exact pass has consumed the buffers -- marks every query of the call overflowed and the
// caller repairs them on the scan path, as for an overflowed candidate buffer.
constexpr int SCATTER_LISTS = 16;  // wave lists per workgroup (one per wave of the 1024-thread block)
__global__ __launch_bounds__(1024) void scatter_pairs_kernel(const u64* pairs, const uint32_t* pair_count, uint32_t nlists, uint32_t pair_cap,
                                                             u64* cand, uint32_t* count, uint32_t cap, uint32_t* lost) {
  // 256 counters take every pair of the call: one global atomic per pair serialises ~1000 deep per counter (measured 0.16 ms
  // for 320 k pairs).  So a workgroup first counts its 16 lists per query in LDS, reserves one range per query with ONE global
  // atomic, and hands out the positions inside the ranges from LDS again.
  __shared__ uint32_t hist[256], base[256];
  const uint32_t w = blockIdx.x * SCATTER_LISTS + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (threadIdx.x < 256) hist[threadIdx.x] = 0;
  uint32_t have = w < nlists ? pair_count[w] : 0;
  if (have > pair_cap) {
    if (lane == 0) *lost = 1u;
    have = pair_cap;
  }
  __syncthreads();
  const u64* mine = pairs + (size_t)w * pair_cap;
  for (uint32_t i = lane; i < have; i += 64) atomicAdd(&hist[(uint32_t)(mine[i] >> 32) & 255u], 1u);
  __syncthreads();
  if (threadIdx.x < 256) {
    const uint32_t c = hist[threadIdx.x];
    base[threadIdx.x] = c ? atomicAdd(&count[threadIdx.x], c) : 0u;
    hist[threadIdx.x] = 0;
  }
  __syncthreads();
  for (uint32_t i = lane; i < have; i += 64) {
    const u64 p = mine[i];
    const uint32_t q = (uint32_t)(p >> 32) & 255u, row = (uint32_t)p;
    const uint32_t pos = base[q] + atomicAdd(&hist[q], 1u);
    // candidate entry until the exact pass rewrites it as a key: the row where a key keeps it (~row in the low half) and
    // the pair's integer dot product above it (INT_MIN = unknown) for refine_pairs_kernel
    const int d = (int)(uint32_t)(p >> 32) >> 8;
    if (pos < cap) cand[(size_t)q * cap + pos] = ((u64)(uint32_t)(d == -(1 << 23) ? INT_MIN : d) << 32) | (u64)(~row);
  }
}

// One workgroup.  Also lists the block's overflowed queries for the repair scan: over_list = {n, q_0 .. q_(n-1)} (any
// order; null = no list wanted), so that its grid needs a few rows instead of one per query (scan_kernel_listed).
__global__ void mark_lost_kernel(uint32_t* count, uint32_t nq, const uint32_t* lost, uint32_t cap, uint32_t* over_list) {
  __shared__ uint32_t n_over;
  if (threadIdx.x == 0) n_over = 0;
  __syncthreads();
  const bool all = lost && *lost;
  for (uint32_t q = threadIdx.x; q < nq; q += blockDim.x) {
    uint32_t c = count[q];
    if (all && c <= cap) count[q] = c = cap + 1u;
    if (over_list && c > cap) over_list[1 + atomicAdd(&n_over, 1u)] = q;
  }
  __syncthreads();
  if (over_list && threadIdx.x == 0) over_list[0] = n_over;
}

// ------------------------------------------------------------------------------------------------
// second selection stage, between the scatter and the exact pass: one workgroup per query.  The full pass kept every row
// whose UPPER bound reaches tau, and tau came from a 1/div sample: ~k * div rows plus the bound's width, 1 200 per query at
// 10 M rows -- a 491 MB gather of fp32 rows per 256-query batch for the exact pass.  Each kept pair carries its integer dot
// product D, so the row's own bounds cost nothing:
//     lower(r) = s_g s_q D - (a_g E_q + b_g M_q)   (the PHASE 0 form),   upper(r) = s_g s_q D + (a_g E_q + b_g M_q)
// (L2: of v = 2 c.q - |c|^2 with the row's cached norm, as in the tile kernel).  tau2 = the k-th largest lower(r) among the
// query's candidates: k rows score at least tau2, so a row with upper(r) < tau2 is not among the k best (ties stay: the test
// is strict) and is dropped before any fp32 row is read -- typically 95 % of them.  Rows that cannot vouch (their group
// holds a non-finite row, an unknown D, a non-finite query) have lower = -inf, upper = +inf: always kept, never counted.
// Removed rows (NaN: never a result) are dropped here.  A query whose candidate buffer overflowed is left alone (the repair
// path re-runs it), and so is one with fewer than k vouching rows (tau2 = -inf).
// Candidate entry in: (D << 32) | ~row (scatter_pairs_kernel); out: the same, compacted in place; count[q] = rows kept.
// ------------------------------------------------------------------------------------------------
struct RefineArgs {
  u64* cand;
  uint32_t* count;
  uint32_t cap;
  const f4* groups;   // {s_g, a_g, b_g, vouch} per 64-row group
  const u64* gbad;    // removed / past-the-end rows, one bit per row
  const float* cn;    // L2: squared norms
  const f4* qpar;     // {s_q, E_q, M_q, 1 / s_q} per query of the block
  int k;
  uint32_t n_lds;     // candidates per query the launch's dynamic LDS holds (4 bytes each; <= REFINE_R * 1024)
};

// an entry's row and dot product, and what its bounds need from memory (all loads issued together)
struct RefineEntry {
  uint32_t row;
  int D;
  f4 gt;
  u64 bad;
  float cn;
};
template <int METRIC>
__device__ __forceinline__ RefineEntry refine_fetch(const RefineArgs& a, u64 e) {
  RefineEntry r;
  r.row = ~(uint32_t)e;
  r.D = (int)(uint32_t)(e >> 32);
  const uint32_t g = r.row >> 6;
  r.bad = a.gbad[g];
  r.gt = a.groups[g];
  r.cn = METRIC == WDBX_METRIC_L2 ? a.cn[r.row] : 0.f;
  return r;
}
// false = drop the entry (a removed row: its exact score is NaN, never a result)
template <int METRIC>
__device__ __forceinline__ bool refine_bounds(const RefineEntry& r, const f4 p, float& lb, float& ub) {
  lb = -INFINITY;
  ub = INFINITY;
  if ((r.bad >> (r.row & 63u)) & 1ull) return false;
  if (r.D == INT_MIN) return true;
  const f4 gt = r.gt;
  const float err = gt.y * p.y + gt.z * p.z;
  if constexpr (METRIC == WDBX_METRIC_L2) {
    const float ss = 2.0f * gt.x * p.x, w = ss * (float)r.D;
    lb = fmaf(ss, (float)r.D, -r.cn * 1.0001f) - 8e-7f * fabsf(w) - 2.0f * err * 1.000001f;
    ub = (w - r.cn * 0.9999f) + 2.0f * err * 1.00001f + 8e-6f * (fabsf(w) + err) + ss;
  } else {
    const float w = gt.x * p.x * (float)r.D;
    lb = w - err * 1.000001f - 4e-7f * fabsf(w);
    ub = w + err * 1.00001f + 4e-6f * (fabsf(w) + err) + gt.x * p.x;  // (+ one unit of D, as the tile kernel's threshold)
  }
  if (!(gt.w == 1.0f) || !(lb == lb)) lb = -INFINITY;
  if (!(ub == ub)) ub = INFINITY;
  return true;
}

// n_lds: entries whose upper bounds the dynamic LDS holds (4 bytes each; at most REFINE_R per thread) -- the usual case
// (about 1 200 candidates per query): tau2 by block_kth_threshold over lower bounds kept in registers.  Beyond it (large k,
// huge sample strides): the wave lists of the merge kernels.
constexpr int REFINE_R = 8;
template <int METRIC, bool REG>
__global__ __launch_bounds__(1024) void refine_pairs_kernel(RefineArgs a) {
  extern __shared__ u64 lds_lists[];
  __shared__ uint32_t s_wcnt[16], s_k[KTH_SCRATCH];
  __shared__ float s_tau;
  const uint32_t q = blockIdx.x;
  const uint32_t n = a.count[q];
  const int k = a.k;
  if (n > a.cap || n <= (uint32_t)k) return;  // overflowed: repaired later; k rows or fewer: nothing to drop (block-uniform)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  u64* const list = a.cand + (size_t)q * a.cap;
  const f4 p = a.qpar[q];
  const bool in_lds = n <= a.n_lds;
  float* const s_ub = (float*)lds_lists;
  float tau2;
  if (in_lds) {
    if (threadIdx.x < KTH_SCRATCH) s_k[threadIdx.x] = 0;
    uint32_t v[REFINE_R];
    // (two entries per trip, so that their dependent loads overlap)
#pragma unroll
    for (int r = 0; r < REFINE_R; r += 2) {
      const uint32_t i0 = (uint32_t)r * blockDim.x + threadIdx.x, i1 = i0 + blockDim.x;
      v[r] = 0u;
      v[r + 1] = 0u;
      if (r * blockDim.x >= n) continue;  // (block-uniform)
      const u64 e0 = i0 < n ? list[i0] : ~0ull, e1 = i1 < n ? list[i1] : ~0ull;  // (past the end: row 0's addresses, never used)
      const RefineEntry r0 = refine_fetch<METRIC>(a, e0), r1 = refine_fetch<METRIC>(a, e1);
      float lb, ub;
      if (i0 < n) {
        const bool ok = refine_bounds<METRIC>(r0, p, lb, ub);
        v[r] = (ok && lb > -INFINITY) ? f2ord(lb + 0.0f) : 0u;
        s_ub[i0] = ok ? ub : -INFINITY;  // (a dropped row: below every threshold)
      }
      if (i1 < n) {
        const bool ok = refine_bounds<METRIC>(r1, p, lb, ub);
        v[r + 1] = (ok && lb > -INFINITY) ? f2ord(lb + 0.0f) : 0u;
        s_ub[i1] = ok ? ub : -INFINITY;
      }
    }
    __syncthreads();
    const uint32_t ord = block_kth_threshold<REFINE_R>(v, (uint32_t)k, s_k);
    if (!ord) return;  // fewer than k vouching rows
    tau2 = ord2f(ord);  // (at most 2^-15 relative below the k-th largest lower bound: as valid, a few more rows kept)
  } else {
    TopList<REG> top;
    top.init(lds_lists + (size_t)wave * k, k, lane);
    u64 thr = 0;
    for (uint32_t i0 = (uint32_t)wave * 64; i0 < n; i0 += (uint32_t)nwaves * 64) {
      const uint32_t i = i0 + lane;
      float lb, ub;
      const RefineEntry r = refine_fetch<METRIC>(a, i < n ? list[i] : ~0ull);
      const bool ok = i < n && refine_bounds<METRIC>(r, p, lb, ub);
      const u64 key = (ok && lb > -INFINITY) ? make_key(lb + 0.0f, r.row) : 0ull;
      thr = top.offer(key, key > thr, thr, lane);
    }
    if constexpr (REG) top.store(lds_lists + (size_t)wave * k, 1, lane);
    __syncthreads();
    if (wave == 0) {
      const u64* mine = lds_lists + (size_t)lane * k;
      TopList<REG> fin;
      fin.init(lds_lists + (size_t)nwaves * k, k, lane);
      const u64 kth = walk_lists<REG>([&](int ptr) { return mine[ptr]; }, lane < nwaves, k, fin, 0, lane);
      if (lane == 0) s_tau = kth ? key_score(kth) : -INFINITY;
    }
    __syncthreads();
    tau2 = s_tau;
    if (tau2 == -INFINITY) return;  // fewer than k vouching rows
  }
  const u64 below = (1ull << lane) - 1;
  uint32_t kept = 0;  // block-uniform
  for (uint32_t base = 0; base < n; base += blockDim.x) {
    const uint32_t i = base + threadIdx.x;
    const u64 e = i < n ? list[i] : ~0ull;
    bool keep;
    if (in_lds) {
      keep = i < n && !(s_ub[i] < tau2);
    } else {
      float lb, ub;
      const RefineEntry r = refine_fetch<METRIC>(a, e);
      keep = i < n && refine_bounds<METRIC>(r, p, lb, ub) && !(ub < tau2);
    }
    const u64 m = __ballot(keep);
    if (lane == 0) s_wcnt[wave] = (uint32_t)__builtin_popcountll(m);
    __syncthreads();  // (every entry of this trip has been read: the writes below land at or before their own positions)
    uint32_t off = kept, total = 0;
    for (int w = 0; w < nwaves; ++w) {
      const uint32_t c = s_wcnt[w];
      off += w < wave ? c : 0u;
      total += c;
    }
    if (keep) list[off + (uint32_t)__builtin_popcountll(m & below)] = e;
    kept += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) a.count[q] = kept;
}

// a_ref, b_ref for the prefilter epilogue: the largest a_g, b_g among the ORDINARY groups -- finite bounds no larger than
// 1.5 x the mean over the finite ones (outlier groups, groups holding an infinite row and pad groups take the exact
// epilogue).  ref = {a_ref, b_ref, sum a, sum b} (floats; maxima as bit patterns of non-negative floats), cnt = finite groups.
// pass 0 accumulates the sums, pass 1 the maxima.
__global__ __launch_bounds__(256) void group_ref_kernel(const f4* groups, u64 n_groups, float* ref, uint32_t* cnt, int pass) {
  float mean_a = 0.f, mean_b = 0.f;
  if (pass == 1) {
    const uint32_t c = *cnt;
    mean_a = c ? ref[2] / (float)c : 0.f;
    mean_b = c ? ref[3] / (float)c : 0.f;
  }
  for (u64 g = (u64)blockIdx.x * 256 + threadIdx.x; g < n_groups; g += (u64)gridDim.x * 256) {
    const f4 t = groups[g];
    const bool fin = t.x > 0.f && t.y >= 0.f && t.y < INFINITY && t.z >= 0.f && t.z < INFINITY;
    if (!fin) continue;
    if (pass == 0) {
      atomicAdd(&ref[2], t.y);
      atomicAdd(&ref[3], t.z);
      atomicAdd(cnt, 1u);
    } else if (t.y <= 1.5f * mean_a && t.z <= 1.5f * mean_b) {
      atomicMax((uint32_t*)&ref[0], __float_as_uint(t.y));
      atomicMax((uint32_t*)&ref[1], __float_as_uint(t.z));
    }
  }
}
