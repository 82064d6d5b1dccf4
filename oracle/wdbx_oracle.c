/* wdbx_oracle.c -- plain C restatement of the exact per-shard search.  TEST INFRASTRUCTURE ONLY
 * (same status and parity notes as oracle/wdbx_oracle.py; only tests/, smoke() and bench.py's
 * cpu_baseline leg may use it).
 *
 * Follows FaissIndex.search (/root/reference/wdbx/core/indexing.py:983-1030): every stored row is
 * scored against the query (inner product of unit rows = cosine; squared L2 as the build's
 * extension), the k best are returned best first; unused slots hold -1 (indexing.py:1023).
 * Unlike the BLAS-backed numpy restatement every row is summed in the SAME fixed order (one scalar
 * fp32 fmaf chain per row), so bit-equal rows get bit-equal scores and the total order
 * (score desc, row asc) is exact -- this is the oracle for tie-order tests.
 * NaN scores are never returned (faiss' heap comparisons are false for NaN).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static int better(float sa, int64_t ra, float sb, int64_t rb) { /* a ranks before b */
  if (sa > sb) return 1;
  if (sa < sb) return 0;
  return ra < rb;
}

/* rows [n, d] row-major fp32; metric 0 = inner product, 1 = squared L2; allowed = NULL or n bytes
 * (0 = row may not be returned).  out_idx/out_score have k slots.  Returns the number of hits. */
int wdbx_oracle_flat_search(const float* rows, int64_t n, int d, const float* query, int k, int metric,
                            const uint8_t* allowed, int64_t* out_idx, float* out_score) {
  int have = 0;
  for (int i = 0; i < k; ++i) {
    out_idx[i] = -1;
    out_score[i] = 0.0f;
  }
  for (int64_t r = 0; r < n; ++r) {
    if (allowed && !allowed[r]) continue;
    const float* c = rows + r * (int64_t)d;
    float acc = 0.0f;
    if (metric == 0) {
      for (int j = 0; j < d; ++j) acc = fmaf(c[j], query[j], acc);
    } else {
      for (int j = 0; j < d; ++j) {
        const float t = c[j] - query[j];
        acc = fmaf(t, t, acc);
      }
    }
    if (acc != acc) continue;
    const float rank = (metric == 0 ? acc : -acc) + 0.0f;
    /* insertion into the sorted result (k is small in the tests this serves) */
    if (have == k && !better(rank, r, metric == 0 ? out_score[k - 1] : -out_score[k - 1], out_idx[k - 1])) continue;
    int pos = have < k ? have : k - 1;
    while (pos > 0 && better(rank, r, metric == 0 ? out_score[pos - 1] : -out_score[pos - 1], out_idx[pos - 1])) {
      out_idx[pos] = out_idx[pos - 1];
      out_score[pos] = out_score[pos - 1];
      --pos;
    }
    out_idx[pos] = r;
    out_score[pos] = acc;
    if (have < k) ++have;
  }
  return have;
}

/* v / ||v||_2 in fp32 when the norm is > 0 (indexing.py:851-856), sequential sum */
void wdbx_oracle_normalize_rows(float* rows, int64_t n, int d) {
  for (int64_t r = 0; r < n; ++r) {
    float* c = rows + r * (int64_t)d;
    float s = 0.0f;
    for (int j = 0; j < d; ++j) s = fmaf(c[j], c[j], s);
    const float nrm = sqrtf(s);
    if (nrm > 0.0f)
      for (int j = 0; j < d; ++j) c[j] = c[j] / nrm;
  }
}

/* counter-based synthetic rows of BASELINE.md section 3 */
static uint64_t splitmix64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

void wdbx_oracle_synth_rows(uint64_t seed, uint64_t row0, int64_t n, int d, float* out) {
  for (int64_t r = 0; r < n; ++r)
    for (int j = 0; j < d; ++j) {
      const uint64_t h = splitmix64(seed ^ ((row0 + (uint64_t)r) * (uint64_t)d + (uint64_t)j));
      out[r * (int64_t)d + j] = (float)((int32_t)(h >> 40) - (1 << 23)) * 1.1920928955078125e-07f;
    }
}
