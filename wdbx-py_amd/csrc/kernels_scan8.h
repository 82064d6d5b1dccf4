// kernels_scan8.h -- single queries over the u8 shadow copy: selection scan, quantisation, query norms.
// Part of the single translation unit wdbx_hip.hip (included there, in order); not a standalone header.

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
// 2 m + 3e-5 |c|^2).  Rows with an infinite element carry a NaN scale: never sampled, always candidates.  Rows with
// a NaN element score NaN against every query and are never returned (include/wdbx_hip.h; the Python layer
// overwrites REMOVED rows with NaN for exactly that reason): they carry a NEGATIVE scale and both phases skip them,
// so any number of removed rows costs no candidate slots.
// ------------------------------------------------------------------------------------------------
typedef uint32_t u4v __attribute__((ext_vector_type(4)));

struct Scan8Args {
  const u4v* rows8;   // [n_rows][pieces] 16-byte pieces of u8
  const float* scale;   // [n_rows]
  const float* cn;      // L2: squared fp32 norm per row
  const f4* query;      // fp32 [pieces * 4] quads (zero padded by the caller's buffer pitch or by clamping)
  const uint32_t* mask; // optional row filter (bit r set = row r may be returned), as in ScanArgs
  uint32_t n_rows, pieces, qquads;  // qquads: quads the query buffer really holds
  u64* halfmax;         // PHASE 0: one key per sampled 64-row group
  uint32_t num_tiles, tile_stride;  // PHASE 0: tiles of 256 rows = 4 groups, every tile_stride-th tile
  uint32_t sample_nt;   // PHASE 0: non-temporal loads (sample larger than the caches)
  const float* tau;     // PHASE 1
  // PHASE 1, a lone query on a small shard: the threshold is taken by every wave itself from the tau_k-th largest of the
  // tau_n <= 1024 sampled lower-bound keys (what the threshold launch would have computed; saves a dependent launch)
  const u64* tau_keys;
  uint32_t tau_n;
  int tau_k;
  u64* cand;
  uint32_t* count;
  uint32_t cap;
  uint32_t nq;          // scan8_sample4_kernel: queries of the launch
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

// ABLATE (timing only, wrong answers; option scan8_ablate): 1 = a quarter of the convert + fma work per 16 bytes
template <int L, int QPL, int METRIC, int PHASE, int ABLATE = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void scan8_kernel(Scan8Args a) {
  constexpr int R = 64 / L;  // rows per wave pass
  constexpr int U = (QPL >= 6) ? 2 : (QPL >= 4) ? 3 : (QPL == 3) ? 4 : (QPL == 2) ? 6 : 8;  // passes in flight
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane % L, g = lane / L;
  a.query += (size_t)blockIdx.y * a.qquads;  // both phases: one launch for every query of a round, blockIdx.y = query
  if constexpr (PHASE == 0) {
    a.halfmax += (size_t)blockIdx.y * a.num_tiles * 4;
    if (blockIdx.x == 0 && threadIdx.x == 0) a.count[blockIdx.y] = 0;  // the query's candidate counter, for its phase 1
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
  // |q|_1 (an upper bound: rounded up past its own summation error) and 128 * sum q, from the lanes' shares --
  // the same arithmetic in every wave of both phases, so the bounds agree everywhere
  float q1, qsum128;
  {
    float s1 = 0.f, ss = 0.f;
#pragma unroll
    for (int i = 0; i < QPL; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        s1 += (fabsf(q[i][t].x) + fabsf(q[i][t].y)) + (fabsf(q[i][t].z) + fabsf(q[i][t].w));
        ss += (q[i][t].x + q[i][t].y) + (q[i][t].z + q[i][t].w);
      }
    q1 = group_sum<L>(s1) * (1.0f + 1e-5f);
    qsum128 = 128.0f * group_sum<L>(ss);
  }
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
          for (int i = 0; i < QPL; ++i)  // a sample small enough for the caches is re-read from them by the round's other queries
            v[u][i] = a.sample_nt ? __builtin_nontemporal_load(p + i * L) : p[i * L];
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
            // (NaN or negative scale: not sampled; masked-out rows cannot vouch for the threshold either)
            if (sc[u] >= 0.f && lo == lo && (!a.mask || ((a.mask[row[u] >> 5] >> (row[u] & 31)) & 1u))) best = fmaxf(best, lo);
          }
        }
      }
      for (int o = 32; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o));
      if (lane == 0) a.halfmax[grp] = (best == -INFINITY) ? 0ull : make_key(best + 0.0f, grp);
    }
  } else {
    // one launch serves all queries of a round here too (blockIdx.y = query; every query still makes its own pass over
    // all rows): the launches' ramp-up and tail, ~10 us of an 83 us pass over 1.25 M rows, overlap with the neighbours' work
    a.tau += blockIdx.y;
    a.cand += (size_t)blockIdx.y * a.cap;
    a.count += blockIdx.y;
    float thr;
    if (a.tau_keys) {  // (tau_n <= 1024 keys: 16 ordered score values per lane, searched bit by bit by every wave itself)
      uint32_t tv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t p = (uint32_t)r * 64u + (uint32_t)lane;
        tv[r] = p < a.tau_n ? (uint32_t)(a.tau_keys[p] >> 32) : 0u;
      }
      const uint32_t ord = wave_kth_threshold<16>(tv, (uint32_t)a.tau_k);
      thr = ord ? ord2f(ord) : -INFINITY;  // fewer than k vouching groups: every row is a candidate
    } else {
      thr = a.tau[0];
    }
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
        for (int i = 0; i < QPL; ++i) {
          if constexpr (ABLATE == 1) {
            const u4v w = v[u][i];
            s = fmaf((float)((w.x ^ w.y ^ w.z ^ w.w) & 0xFFu), q[i][0].x, s);  // (every loaded byte still feeds the result)
            s = fmaf((float)(((w.x ^ w.y ^ w.z ^ w.w) >> 8) & 0xFFu), q[i][1].y, s);
            s = fmaf((float)(((w.x ^ w.y ^ w.z ^ w.w) >> 16) & 0xFFu), q[i][2].z, s);
            s = fmaf((float)((w.x ^ w.y ^ w.z ^ w.w) >> 24), q[i][3].w, s);
          } else {
            s = u8_dot16(v[u][i], q[i], s);
          }
        }
        s = group_sum<L>(s);
        if (j == 0 && row[u] <= last_row) {
          float m;
          const float w = finish(s, sc[u], cn[u], m);
          // !(w + m < thr): also true for a NaN bound, so rows with non-finite elements always go to the exact pass
          // only rows that clear the threshold look at their mask bit
          // (a negative scale marks a row with a NaN element: its score is NaN for every query, it is never a result)
          if constexpr (ABLATE != 0) {  // (timing only: nothing is appended, the arithmetic stays live)
            if (w + m == 3.0e38f) a.cand[0] = make_key(w, row[u]);
          } else
          if (!(sc[u] < 0.f) && !(w + m < thr) && (!a.mask || ((a.mask[row[u] >> 5] >> (row[u] & 31)) & 1u))) {
            const uint32_t pos = atomicAdd(a.count, 1u);
            if (pos < a.cap) a.cand[pos] = make_key((w == w) ? w + 0.0f : INFINITY, row[u]);
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// PHASE 0 for QN = 4 (QPL <= 2) or 3 (QPL = 3) queries of a round per workgroup (round 3).  The round's queries sample the SAME rows; on a large shard
// the sample does not stay cached (10 M x 384: 121 MB per query, 3.9 GB per round of 32 -- one more full pass per round,
// 3 % of the headline's time; 10 M x 768 at k = 100: 10 %).  Here a wave loads a sampled row once and takes its lower bound
// for QN queries: 16 converts + 16 QN fmas per 16 bytes -- at QN = 4 about what the memory system delivers per CU, so the
// pass runs at the rate of the stream with a quarter (a third) of the bytes.  Same arithmetic per query as
// scan8_kernel<PHASE 0> (same order of operations: identical lower bounds, identical thresholds).  QN query register sets of
// QPL x 4 quads each: 4 x 48 registers at QPL = 3 spill (92 bytes per lane), hence 3 there.
// a.nq = queries of the launch; blockIdx.y = query group.
// ------------------------------------------------------------------------------------------------
template <int L, int QPL, int METRIC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void scan8_sample4_kernel(Scan8Args a) {
  constexpr int QN = QPL >= 3 ? 3 : 4;
  constexpr int R = 64 / L;
  constexpr int U = 2;  // passes in flight
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane % L, g = lane / L;
  const uint32_t qbase = blockIdx.y * QN;
  if (blockIdx.x == 0 && threadIdx.x < QN && qbase + threadIdx.x < a.nq) a.count[qbase + threadIdx.x] = 0;
  f4 q[QN][QPL][4];
  float q1[QN], qsum128[QN];
#pragma unroll
  for (int t = 0; t < QN; ++t) {
    const bool live = qbase + t < a.nq;
    const f4* qp = a.query + (size_t)(live ? qbase + t : qbase) * a.qquads;
    float s1 = 0.f, ss = 0.f;
#pragma unroll
    for (int i = 0; i < QPL; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint32_t quad = (uint32_t)(j + i * L) * 4 + c;
        q[t][i][c] = quad < a.qquads ? qp[quad] : f4{0.f, 0.f, 0.f, 0.f};
        s1 += (fabsf(q[t][i][c].x) + fabsf(q[t][i][c].y)) + (fabsf(q[t][i][c].z) + fabsf(q[t][i][c].w));
        ss += (q[t][i][c].x + q[t][i][c].y) + (q[t][i][c].z + q[t][i][c].w);
      }
    q1[t] = group_sum<L>(s1) * (1.0f + 1e-5f);
    qsum128[t] = 128.0f * group_sum<L>(ss);
  }
  const uint32_t last_row = a.n_rows - 1;
  const uint32_t ngroups = a.num_tiles * 4;
  for (uint32_t grp = blockIdx.x * 4 + wave; grp < ngroups; grp += gridDim.x * 4) {
    const uint32_t row0 = (grp >> 2) * a.tile_stride * 256 + (grp & 3) * 64;
    float best[QN];
#pragma unroll
    for (int t = 0; t < QN; ++t) best[t] = -INFINITY;
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
        for (int i = 0; i < QPL; ++i) v[u][i] = __builtin_nontemporal_load(p + i * L);
        sc[u] = a.scale[rc];
        cn[u] = METRIC == WDBX_METRIC_L2 ? a.cn[rc] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool ok = row[u] <= last_row && sc[u] >= 0.f &&
                        (!a.mask || ((a.mask[min(row[u], last_row) >> 5] >> (row[u] & 31)) & 1u));
#pragma unroll
        for (int t = 0; t < QN; ++t) {
          float s = 0.f;
#pragma unroll
          for (int i = 0; i < QPL; ++i) s = u8_dot16(v[u][i], q[t][i], s);
          s = group_sum<L>(s);
          float w = sc[u] * (s - qsum128[t]);
          float m = 0.51f * sc[u] * q1[t];
          if constexpr (METRIC == WDBX_METRIC_L2) {
            w = fmaf(2.0f, w, -cn[u]);
            m = fmaf(2.0f, m, 3e-5f * cn[u]);
          }
          const float lo = w - m;
          if (ok && lo == lo) best[t] = fmaxf(best[t], lo);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < QN; ++t) {
      float b = best[t];
      for (int o = 32; o > 0; o >>= 1) b = fmaxf(b, __shfl_xor(b, o));
      if (lane == 0 && qbase + t < a.nq)
        a.halfmax[(size_t)(qbase + t) * ngroups + grp] = (b == -INFINITY) ? 0ull : make_key(b + 0.0f, grp);
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
    bool finite = true, has_nan = false;
    for (uint32_t c = lane; c < dim; c += 64) {
      const float v = p[c];
      finite = finite && (fabsf(v) <= 3.4028235e38f);
      has_nan = has_nan || (v != v);
      mx = fmaxf(mx, fabsf(v));
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    finite = __all(finite);
    has_nan = __any(has_nan);
    // rows of vanishing magnitude (127 / max would overflow): all elements quantise to 0 and the scale is set so that
    // the bound 0.51 s |q|_1 still covers the whole (negligible) score, |c.q| <= max|c| |q|_1
    const bool vanishing = mx < 1.2e-30f;
    const float sc = has_nan ? -1.0f : !finite ? NAN : vanishing ? 2.0f * mx : mx / 127.0f;
    const float inv = (finite && !vanishing) ? 127.0f / mx : 0.f;
    for (uint32_t c = lane; c < pitch8; c += 64) {
      float x = (c < dim && finite) ? rintf(p[c] * inv) : 0.f;
      x = fminf(fmaxf(x, -127.f), 127.f);
      out[r * pitch8 + c] = (uint8_t)((int)x + 128);
    }
    if (lane == 0) scale[r] = sc;
  }
}
