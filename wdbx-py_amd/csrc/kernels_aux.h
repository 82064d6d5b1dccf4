// kernels_aux.h -- row norms, threshold margins, exact re-scoring; ingest helpers; read probes.
// Part of the single translation unit wdbx_hip.hip (included there, in order); not a standalone header.

// ------------------------------------------------------------------------------------------------
// L2 on the batched path: row norms, threshold margin, exact re-scoring of the selected candidates
// ------------------------------------------------------------------------------------------------
// cn[r] = sum c^2 (one wave per row) and the running maximum of cn (float bits of non-negative values
// order like unsigned integers)
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float* rows, u64 r0, u64 n, uint32_t pitch, float* cn,
                                                         uint32_t* cn_max_bits) {
  const int lane = threadIdx.x & 63;
  const u64 wave = (u64)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (u64)gridDim.x * 4;
  const uint32_t pitch4 = pitch / 4;
  uint32_t wmax = 0, wcnt = 0;  // this wave's running maximum: ONE atomic per wave at the end, not one per row
  float wsum = 0.f;
  for (u64 r = r0 + wave; r < n; r += nw) {
    const f4* p = (const f4*)(rows + r * pitch);
    float s = 0.f;
    for (uint32_t c = lane; c < pitch4; c += 64) {
      const f4 v = __builtin_nontemporal_load(p + c);
      s = fmaf(v.x, v.x, s);
      s = fmaf(v.y, v.y, s);
      s = fmaf(v.z, v.z, s);
      s = fmaf(v.w, v.w, s);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
      cn[r] = s;
      if (s == s) wmax = max(wmax, __float_as_uint(s));
      if (s < INFINITY) {
        wsum += s;
        ++wcnt;
      }
    }
  }
  if (!cn_max_bits) return;  // in-place refresh of overwritten rows: the statistics are recomputed by cn_stats_kernel
  if (lane == 0 && wmax) atomicMax(cn_max_bits, wmax);
  // word [1]: running sum of the finite norms, word [2]: how many (for the mean: decides between one global bound and
  // per-group bounds)
  if (lane == 0 && wsum > 0.f) atomicAdd((float*)(cn_max_bits + 1), wsum);
  if (lane == 0 && wcnt) atomicAdd(cn_max_bits + 2, wcnt);
}

// the same statistics from the cached norms alone (after rows were overwritten in place: the running maximum and sum
// of row_sqnorm_kernel only ever grow, so they are rebuilt): stats[0] = largest non-NaN norm (float bits), [1] = sum of
// the finite norms, [2] = their count.  The caller zeroes the three words first.
__global__ __launch_bounds__(256) void cn_stats_kernel(const float* cn, u64 n, uint32_t* stats) {
  uint32_t wmax = 0, wcnt = 0;
  float wsum = 0.f;
  for (u64 r = (u64)blockIdx.x * 256 + threadIdx.x; r < n; r += (u64)gridDim.x * 256) {
    const float s = cn[r];
    if (s == s) wmax = max(wmax, __float_as_uint(s));
    if (s < INFINITY) {
      wsum += s;
      ++wcnt;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, o));
    wsum += __shfl_xor(wsum, o);
    wcnt += (uint32_t)__shfl_xor((int)wcnt, o);
  }
  if ((threadIdx.x & 63) == 0) {
    if (wmax) atomicMax(stats, wmax);
    if (wsum > 0.f) atomicAdd((float*)(stats + 1), wsum);
    if (wcnt) atomicAdd(stats + 2, wcnt);
  }
}

// gmax[g] = largest squared norm among rows 64 g .. 64 g + 63 (NaN norms skipped), one thread per group
__global__ __launch_bounds__(256) void group_max_kernel(const float* cn, u64 n, float* gmax) {
  const u64 groups = (n + 63) / 64;
  for (u64 g = (u64)blockIdx.x * 256 + threadIdx.x; g < groups; g += (u64)gridDim.x * 256) {
    float m = 0.f;
    const u64 e = min(n, g * 64 + 64);
    for (u64 r = g * 64; r < e; ++r) m = fmaxf(m, cn[r]);
    gmax[g] = m;
  }
}

// qn[q] = |q| rounded up (an upper bound), 0 for the padded queries of a block; one wave per query
__global__ void query_norm_kernel(const float* queries, uint32_t pitch, int nv, int gbn, float* qn) {
  const int q = blockIdx.x, lane = threadIdx.x;
  if (q >= gbn) return;
  float s = 0.f;
  if (q < nv) {
    const float* p = queries + (size_t)q * pitch;
    for (uint32_t c = lane; c < pitch; c += 64) s = fmaf(p[c], p[c], s);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  }
  if (lane == 0) qn[q] = sqrtf(s) * 1.0001f;
}

// tau[q] -= margin(q), a rigorous bound on the rounding error of the SELECTION scores, so that no true
// top-k row can fall below the threshold.  With u = 2^-24, gamma = n u / (1 - n u) for fp32 chains of n terms:
//   fp32 tiles, L2:      v = 2 c.q - |c|^2,  |v_fp32 - v| <= gamma (2 |c||q| + |c|^2)
//   bf16 tiles:          c and q are rounded to bf16 (relative error <= 2^-8 each), their products are exact
//                        in fp32, so |dot_bf16 - c.q| <= (2^-7 + 2^-16 + gamma) |c||q|  (Cauchy-Schwarz);
//                        cosine: v = dot;  L2: v = 2 dot - |c|^2 with the fp32 bound on the second term.
// Both the threshold (a maximum of such values) and every candidate carry that error, hence 2x.
// (fp32 tiles with the cosine metric need no margin: selection and final scores are the same numbers.)
__global__ void tau_margin_kernel(float* tau, const float* queries, uint32_t pitch, int nv, const uint32_t* cn_max_bits,
                                  int metric, int bf16, float floor_abs) {
  const int q = blockIdx.x, lane = threadIdx.x;  // one wave per query
  if (q >= nv) return;
  const float* p = queries + (size_t)q * pitch;
  float s = 0.f;
  for (uint32_t c = lane; c < pitch; c += 64) s = fmaf(p[c], p[c], s);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) {
    const float cmax = __uint_as_float(*cn_max_bits);
    const float nu = (float)(pitch + 18) * 5.9604645e-08f;
    const float gamma = 1.02f * nu / (1.0f - nu);
    const float eps_dot = gamma + (bf16 ? 1.05f * 0.0078125f : 0.0f);
    const float cq = sqrtf(cmax * s) * 1.0001f;
    float margin = metric == WDBX_METRIC_L2 ? 2.0f * (2.0f * eps_dot * cq + gamma * cmax) : 2.0f * eps_dot * cq;
    // + the absolute term for operands in the denormal range (see GemmArgs::floor_abs)
    margin = margin * 1.01f + 2.0f * (metric == WDBX_METRIC_L2 ? 2.0f : 1.0f) * floor_abs * (sqrtf(cmax) + sqrtf(s));
    if (!(margin == margin)) margin = INFINITY;  // NaN query: select everything, the exact pass decides
    if (tau[q] > -INFINITY) tau[q] -= margin;
  }
}

// every kept candidate of every query is re-scored exactly in fp32, one wave per candidate: cosine by the
// inner product, L2 by the direct form sum (c - q)^2 (no cancellation); its key becomes (score, row)
// host_keys / host_count (a lone blocking query): the exact keys and the candidate count also go to mapped host memory,
// where the caller ranks them after its synchronisation (no merge launch)
template <int METRIC>
__global__ __launch_bounds__(256) void rescore_kernel(const f4* rows, uint32_t pitch4, const f4* queries, u64* cand,
                                                      const uint32_t* count, uint32_t cap, u64* host_keys = nullptr,
                                                      uint32_t* host_count = nullptr) {
  const int lane = threadIdx.x & 63;
  const uint32_t q = blockIdx.y;
  const uint32_t have = min(count[q], cap);
  if (host_count && blockIdx.x == 0 && threadIdx.x == 0) *host_count = count[q];
  const f4* qp = queries + (size_t)q * pitch4;
  for (uint32_t j = blockIdx.x * 4 + (threadIdx.x >> 6); j < have; j += gridDim.x * 4) {
    u64* slot = cand + (size_t)q * cap + j;
    const uint32_t row = key_row(*slot);
    const f4* cp = rows + (size_t)row * pitch4;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (uint32_t i = lane; i < pitch4; i += 64) acc = accum<METRIC>(acc, cp[i], qp[i]);
    float s = (acc.x + acc.y) + (acc.z + acc.w);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (METRIC == WDBX_METRIC_L2) s = -s;
    if (lane == 0) {
      const u64 key = (s == s) ? make_key(s + 0.0f, row) : 0ull;
      *slot = key;
      if (host_keys) host_keys[j] = key;
    }
  }
}

// compaction (wdbx_index_compact): dst row i <- rows[src[i]], one wave per row, 16 bytes per lane and trip
__global__ __launch_bounds__(256) void gather_rows_kernel(const f4* rows, uint32_t pitch4, const u64* src, u64 n, f4* dst) {
  const int lane = threadIdx.x & 63;
  for (u64 i = (u64)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += (u64)gridDim.x * 4) {
    const f4* from = rows + (size_t)src[i] * pitch4;
    f4* to = dst + (size_t)i * pitch4;
    for (uint32_t c = lane; c < pitch4; c += 64) to[c] = from[c];
  }
}

// ------------------------------------------------------------------------------------------------
// ingest helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 splitmix64(u64 x) {
  u64 z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void fill_synthetic_kernel(float* dst, u64 n, uint32_t dim, uint32_t pitch,
                                                             u64 seed, u64 counter_row0) {
  const u64 total = n * pitch;
  for (u64 e = (u64)blockIdx.x * 256 + threadIdx.x; e < total; e += (u64)gridDim.x * 256) {
    const u64 r = e / pitch;
    const uint32_t c = (uint32_t)(e - r * pitch);
    float val = 0.f;
    if (c < dim) {
      const u64 h = splitmix64(seed ^ ((counter_row0 + r) * dim + c));
      val = (float)((int)(h >> 40) - (1 << 23)) * 1.1920928955078125e-07f;  // 2^-23, exact
    }
    dst[e] = val;
  }
}

// measurement aid: stream the stored rows with the scan kernel's load shape (16 B per lane,
// grid-stride) and nothing else -- the read ceiling the scan kernel is compared with
template <bool NT>
__global__ __launch_bounds__(256) void probe_read_kernel(const f4* p, u64 n_quads, float* sink) {
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  const u64 stride = (u64)gridDim.x * 256 * 8;
  for (u64 i = (u64)blockIdx.x * 256 * 8 + threadIdx.x; i < n_quads; i += stride) {
    f4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const u64 e = i + (u64)u * 256;
      v[u] = (e < n_quads) ? ld16<NT>(p + e) : acc;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc.x += v[u].x;
      acc.y += v[u].y;
      acc.z += v[u].z;
      acc.w += v[u].w;
    }
  }
  const float s = (acc.x + acc.y) + (acc.z + acc.w);
  if (s == 1.2345e38f) sink[0] = s;  // keeps the loads alive, practically never true
}

// one wave per row: x / sqrt(sum x^2) when the norm is > 0 (indexing.py:851-856)
__global__ __launch_bounds__(256) void normalize_rows_kernel(float* rows, u64 n, uint32_t pitch) {
  const int lane = threadIdx.x & 63;
  const u64 wave = (u64)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (u64)gridDim.x * 4;
  for (u64 r = wave; r < n; r += nw) {
    float* p = rows + r * pitch;
    float s = 0.f;
    for (uint32_t c = lane; c < pitch; c += 64) s = fmaf(p[c], p[c], s);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float nrm = sqrtf(s);
    if (nrm > 0.f)
      for (uint32_t c = lane; c < pitch; c += 64) p[c] = p[c] / nrm;
  }
}
