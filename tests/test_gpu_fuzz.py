"""Seeded differential fuzz on the GPU: random (rows, dim, k, metric, mask, kernel options, batch size)
against the oracle.  Small sizes, many shapes: exercises ragged/generic/unrolled instances, register /
LDS / radix-select top-k, masks, NaN rows, duplicates and the batched path in combinations no
hand-written case covers."""
import numpy as np
import pytest

import wdbx_oracle as O

import os

pytestmark = pytest.mark.gpu

# WDBX_FUZZ_SEEDS=N widens the campaign for a one-off soak (default: 96 scan + 16 batched cases)
_N_SCAN = int(os.environ.get("WDBX_FUZZ_SEEDS", "96"))
_N_BATCH = max(16, _N_SCAN // 6)


def _case(rng):
    d = int(rng.choice([1, 3, 4, 8, 13, 16, 31, 64, 65, 96, 100, 128, 200, 257, 384, 500, 768, 1000, 1536, 2049, 3072, 3100]))
    n = int(rng.choice([1, 2, 63, 64, 65, 500, 1023, 4096, 20_000]))
    if n * d > 30_000_000:
        n = 30_000_000 // d
    k = int(rng.choice([1, 2, 10, 63, 64, 65, 100, 199, 200, 201, 500, 2048]))
    metric = int(rng.integers(0, 2))
    return n, d, k, metric


@pytest.mark.parametrize("seed", range(_N_SCAN))
def test_fuzz_scan_paths(seed):
    from wdbx_amd import _native as native

    rng = np.random.default_rng(1000 + seed)
    n, d, k, metric = _case(rng)
    rows = rng.standard_normal((n, d)).astype(np.float32)
    if metric == 0:
        rows = O.normalize_rows_fast(rows)
    # a few pathologies
    if n > 10 and rng.random() < 0.5:
        rows[rng.integers(0, n)] = np.nan
    if n > 10 and rng.random() < 0.5:
        src = int(rng.integers(0, n))
        for dst in rng.integers(0, n, size=3):
            if not np.isnan(rows[src]).any():
                rows[int(dst)] = rows[src]
    if n > 5 and rng.random() < 0.3:
        rows[int(rng.integers(0, n))] = 0.0
    q = rng.standard_normal(d).astype(np.float32)
    if metric == 0:
        q = O.normalize_vector(q)
    allowed = None
    if rng.random() < 0.4:
        allowed = rng.random(n) < rng.choice([0.05, 0.5, 0.95])
    opts = {}
    for name, values in (("scan_generic", [0, 0, 1]), ("scan_blocked", [0, 1]), ("lds_lists", [0, 0, 1]),
                         ("select_min_k", [200, 200, 1, 0]), ("wg_merge", [1, 1, 0]), ("scan_blocks", [0, 0, 1, 7, 300]),
                         ("zero_copy", [1, 0])):
        opts[name] = int(rng.choice(values))
    with native.NativeIndex(d, metric=metric, capacity_rows=max(1, n // 3)) as ix:
        ix.add(rows[: n // 2])
        ix.add(rows[n // 2:])
        for name, v in opts.items():
            ix.set_option(name, v)
        mw = None if allowed is None else native.pack_row_mask(allowed)
        idx, score = ix.search(q, k, mask_words=mw)
    # oracle: exact fp64 ranking with the build's total order; fp32 score tolerance
    valid = ~np.isnan(rows).any(axis=1)
    if allowed is not None:
        valid &= allowed
    s64 = O.flat_scores_f64(np.nan_to_num(rows), q, metric)
    rank = np.where(valid, s64 if metric == 0 else -s64, -np.inf)
    order = np.lexsort((np.arange(n), -rank))
    nvalid = int(valid.sum())
    kk = min(k, nvalid)
    got = idx[0]
    assert np.all(got[kk:] == -1), (seed, n, d, k, metric, opts)
    exp = order[:kk]
    got_scores = score[0, :kk].astype(np.float64)
    exp_scores = s64[exp]
    tol = 1e-5 * max(1.0, float(np.max(np.abs(exp_scores))) if kk else 1.0) if metric == 1 else 1e-5
    np.testing.assert_allclose(got_scores, exp_scores, atol=tol, rtol=0, err_msg=str((seed, n, d, k, metric, opts)))
    # ids: equal to the fp64 order except where fp64 scores are closer than fp32 can tell apart
    g = got[:kk].tolist()
    if g != exp.tolist():
        assert sorted(g) == sorted(exp.tolist()) or abs(s64[g[-1]] - s64[exp[-1]]) <= tol, (seed, n, d, k, metric, opts)
        for p, (a, b) in enumerate(zip(g, exp.tolist())):
            if a != b:
                assert abs(s64[a] - s64[b]) <= tol, (seed, p, a, b, n, d, k, metric, opts)
    assert len(set(g)) == len(g)


@pytest.mark.parametrize("seed", range(_N_BATCH))
def test_fuzz_batched_path(seed):
    from wdbx_amd import _native as native

    rng = np.random.default_rng(5000 + seed)
    d = int(rng.choice([32, 100, 128, 384, 500, 768, 1000]))   # (i8 tiles: query blocks of 256 / 128 / 64, static and run-time pitch)
    n = int(rng.choice([70_000, 131_072, 200_001]))
    if n * d > 120_000_000:
        n = 70_000
    nq = int(rng.choice([4, 17, 64, 65, 129, 300]))
    k = int(rng.choice([1, 10, 40]))
    metric = int(rng.integers(0, 2))
    rows = O.normalize_rows_fast(rng.standard_normal((n, d)).astype(np.float32))
    queries = O.normalize_rows_fast(rng.standard_normal((nq, d)).astype(np.float32))
    scaled = metric == 0 and seed % 3 == 1   # inner product on rows of very different lengths: per-group scales and bounds
    if scaled:
        rows *= rng.lognormal(0.0, 0.7, size=(n, 1)).astype(np.float32)
    # pathologies: NaN rows, zero rows, a query that equals stored rows (exact ties, also across tiles)
    for r in rng.integers(0, n, size=5):
        rows[int(r)] = np.nan
    for r in rng.integers(0, n, size=5):
        rows[int(r)] = 0.0
    src = queries[0].copy()
    for r in (3, 127, 128, 129, n // 2, n - 1):
        rows[r] = src
    with native.NativeIndex(d, metric=metric, capacity_rows=n) as ix:
        ix.add(rows)
        ix.profile(True)
        idx, score = ix.search(queries, k)
        assert ix.profile_read_gemm()["gemm_launches"] >= 2
    tie_rows = [3, 127, 128, 129, n // 2, n - 1][: k]
    if not scaled:  # (longer rows can beat the exact copies on a plain inner product)
        assert idx[0, : len(tie_rows)].tolist() == tie_rows, (seed, idx[0])
    with np.errstate(invalid="ignore"):
        s = rows @ queries.T
    if metric == 1:  # unit rows and queries: |c - q|^2 = 2 - 2 c.q; rank by -distance, report distance
        with np.errstate(invalid="ignore"):
            s = np.float32(2.0) - np.float32(2.0) * s
        zero_rows = ~np.isnan(rows).any(axis=1) & (np.abs(rows).sum(axis=1) == 0)
        s[zero_rows, :] = 1.0  # |0 - q|^2 = 1
        for qi in range(nq):
            top = O._topk_desc(-s[:, qi], k)
            np.testing.assert_allclose(score[qi], s[top, qi], atol=2e-5, rtol=0)
            if idx[qi].tolist() != top.tolist():
                for a, b in zip(idx[qi].tolist(), top.tolist()):
                    assert a == b or abs(float(s[a, qi]) - float(s[b, qi])) <= 2e-5, (seed, qi)
        return
    for qi in range(nq):
        top = O._topk_desc(s[:, qi], k)
        np.testing.assert_allclose(score[qi], s[top, qi], atol=1e-5, rtol=1e-5 if scaled else 0)
        if idx[qi].tolist() != top.tolist():
            for a, b in zip(idx[qi].tolist(), top.tolist()):
                assert a == b or abs(float(s[a, qi]) - float(s[b, qi])) <= 2e-6 * max(1.0, abs(float(s[a, qi]))), (seed, qi)


_N_SEL = max(24, _N_SCAN // 4)


@pytest.mark.parametrize("seed", range(_N_SEL))
def test_fuzz_single_query_selection_paths(seed):
    """Single queries on the reduced-precision selection paths (u8 scan, bf16 tiles) against the fp32 scan on the
    same handle: random shapes, scales and pathologies; exactness must not depend on how well the rows quantise."""
    from wdbx_amd import _native as native

    rng = np.random.default_rng(9000 + seed)
    d = int(rng.choice([54, 64, 96, 100, 128, 200, 300, 384, 500, 768, 1000, 1536, 2048, 3000, 4096]))
    n = int(rng.choice([65_536, 70_001, 100_000, 150_000, 250_000]))
    if n * d > 40_000_000:
        n = max(65_536, 40_000_000 // d)
    k = int(rng.choice([1, 5, 10, 33, 64, 100, 150, 199, 200, 240]))   # (>= 200: radix-select epilogue, u8 path only)
    while k * 1024 > n:
        k //= 2
    metric = int(rng.integers(0, 2))
    path = int(rng.choice([2, 2, 1]))
    rows = rng.standard_normal((n, d)).astype(np.float32)
    style = int(rng.integers(0, 4))
    if style == 0:
        rows = O.normalize_rows_fast(rows)
    elif style == 1:
        rows *= rng.lognormal(0.0, 1.5, size=(n, 1)).astype(np.float32)          # wildly different row norms
    elif style == 2:
        rows[:, int(rng.integers(0, d))] *= 40.0                                  # one dominant column
    else:
        rows[rng.integers(0, n, size=n // 50), int(rng.integers(0, d))] += 25.0   # outlier elements in 2 % of the rows
    q = rng.standard_normal(d).astype(np.float32)
    if style == 0:
        q = O.normalize_vector(q)
    for r in rng.integers(0, n, size=3):
        rows[int(r)] = np.nan
    for r in rng.integers(0, n, size=3):
        rows[int(r)] = 0.0
    ties = sorted(set(int(r) for r in rng.integers(0, n, size=4)))
    for r in ties:                                                                # exact copies of the query: ties in row order
        rows[r] = q
    with native.NativeIndex(d, metric=metric, capacity_rows=n) as ix:
        ix.add(rows)
        ix.set_option("scan_shadow", path)
        ix.set_option("single_min_rows", 0)
        ix.profile(True)
        idx, score = ix.search(q, k)
        took = ix.get_option("last_single_path")
        ix.set_option("scan_shadow", 0)
        r_idx, r_score = ix.search(q, k)
    assert took in (path, 1 if path == 2 and d > 4096 else path, 0), (seed, took)
    ctx = (seed, n, d, k, metric, path, took, style)
    scale = max(1.0, float(np.max(np.abs(r_score[np.isfinite(r_score)])))) if np.isfinite(r_score).any() else 1.0
    np.testing.assert_allclose(score, r_score, rtol=2e-5, atol=2e-5 * scale, err_msg=str(ctx))
    g, e = idx[0].tolist(), r_idx[0].tolist()
    if g != e:  # only swaps between rows whose fp32 scores differ by summation-order noise
        for p, (a, b) in enumerate(zip(g, e)):
            if a != b:
                assert abs(float(score[0, p]) - float(r_score[0, p])) <= 2e-5 * scale, ctx + (p, a, b)
        assert len(set(g)) == len(g)
    if metric == 0 and style == 0:
        assert g[: min(k, len(ties))] == ties[: min(k, len(ties))], ctx


def test_stress_searches_from_threads_while_the_corpus_changes():
    """Round 4 narrowed the handle's mutex to the enqueue (a small blocking search waits for the GPU on its own event): a
    stress of exactly what that exposes.  8 searcher threads (lone queries, 3-query calls, masked calls, k = 250) run against
    ONE handle while a writer thread appends rows (re-allocating the row buffer and every shadow copy as capacity grows),
    overwrites rows, tombstones rows, compacts and clears + refills.  Nothing may crash or hang; every answer must be a
    well-formed result for SOME state of the corpus (in range, sorted by score, no duplicate rows); after the dust settles
    the handle answers exactly like a fresh one holding the same rows."""
    import threading
    import time

    from wdbx_amd import _native as native

    d, n0 = 96, 260_000
    rng = np.random.default_rng(77)
    base = O.normalize_rows_fast(O.synth_rows(O.SEED_CORPUS, 0, n0 + 120_000, d))
    queries = O.normalize_rows_fast(O.synth_rows(O.SEED_QUERY, 0, 32, d))
    errors, stop = [], threading.Event()
    counts = [0] * 8
    with native.NativeIndex(d, capacity_rows=n0 // 4) as ix:
        ix.add(base[:n0])

        def check(idx, score, k):
            for r_idx, r_score in zip(idx, score):
                live = r_idx[r_idx >= 0]
                assert len(set(live.tolist())) == len(live), "duplicate rows in one answer"
                assert np.all(live < n0 + 120_000)
                s = r_score[: len(live)]
                assert np.all(np.diff(s) <= 0), "scores not descending"
                assert np.all(r_idx[len(live):] == -1)

        def searcher(t):
            try:
                i = t
                while not stop.is_set():
                    kind = (i + t) % 4
                    q = queries[i % 32]
                    if kind == 0:
                        idx, score = ix.search(q, 10)
                    elif kind == 1:
                        idx, score = ix.search(queries[i % 29:i % 29 + 3], 7)
                    elif kind == 2:
                        nrows = ix.size()
                        try:
                            idx, score = ix.search(q, 10, mask_words=native.pack_row_mask((np.arange(nrows) % 2) == 0))
                        except (ValueError, native.HipBackendError):
                            i += 1                      # the corpus grew between size() and the call: refused, as documented
                            continue
                    else:
                        idx, score = ix.search(q, 250)
                    check(idx, score, idx.shape[1])
                    counts[t] += 1
                    i += 1
            except Exception as e:  # noqa: BLE001
                errors.append(e)
                stop.set()

        def writer():
            try:
                at = n0
                for step in range(12):
                    if stop.is_set():
                        return
                    ix.add(base[at:at + 10_000])
                    at += 10_000
                    ix.set_rows(int(rng.integers(0, n0)), base[int(rng.integers(0, n0))][None, :])
                    ix.set_rows(int(rng.integers(0, n0)), np.full((1, d), np.nan, np.float32))   # a tombstone
                    if step == 5:
                        keep = np.arange(0, ix.size(), dtype=np.uint64)
                        ix.compact(keep[keep % 10 != 3])                                         # drops a tenth, moves the rest
                    if step == 8:
                        ix.clear()
                        ix.add(base[:n0])
                        at = n0
                    time.sleep(0.05)
            except Exception as e:  # noqa: BLE001
                errors.append(e)
            finally:
                stop.set()

        threads = [threading.Thread(target=searcher, args=(t,)) for t in range(8)] + [threading.Thread(target=writer)]
        for th in threads:
            th.start()
        for th in threads:
            th.join(300)
        assert not any(th.is_alive() for th in threads), "a thread hangs"
        assert not errors, errors[:1]
        assert min(counts) > 5, counts
        # settled: the handle equals a fresh one with the same rows
        final = ix.get_rows(0, ix.size())
        with native.NativeIndex(d, capacity_rows=ix.size()) as fresh:
            fresh.add(final)
            for q in queries[:6]:
                a, b = ix.search(q, 10), fresh.search(q, 10)
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1], equal_nan=True)
            a, b = ix.search(queries[:8], 10), fresh.search(queries[:8], 10)
            assert np.array_equal(a[0], b[0])
