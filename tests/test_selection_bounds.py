"""CPU check of the error bounds the selection kernels rely on.

The GPU paths never rank by reduced-precision scores: they keep every row whose score COULD reach the k-th best
under a bound and re-score the kept rows in fp32.  These tests restate the kernels' quantisation / rounding in numpy
(float32 arithmetic, same formulas, citing the kernel lines they mirror) and check, against float64 ground truth, that
  (1) the per-row bound |w - c.q| <= 0.51 s |q|_1 of the u8 scan holds on benign and adversarial data,
  (2) the bf16 bound |dot_bf16 - c.q| <= (1.05 * 2^-7 + gamma) |c||q| + floor (|c| + |q|) of the tile kernels holds,
  (3) the two-phase selection built on such bounds (k-th largest sampled LOWER bound as threshold, keep every row
      whose UPPER bound reaches it) never loses a true top-k row, whatever the sample.
"""
import numpy as np
import pytest


def quantise_u8(rows):
    """kernels_scan8.h::rows_to_u8_kernel: s = max|c| / 127, u = rint(c * (127 / max|c|)) + 128 (float32)."""
    rows = rows.astype(np.float32)
    mx = np.max(np.abs(rows), axis=1).astype(np.float32)
    vanishing = mx < np.float32(1.2e-30)                      # 127 / max would overflow: quantise to 0, widen the scale
    sc = np.where(vanishing, np.float32(2.0) * mx, mx / np.float32(127.0)).astype(np.float32)
    with np.errstate(over="ignore", divide="ignore"):
        inv = np.where(vanishing, 0, np.float32(127.0) / np.where(vanishing, 1, mx)).astype(np.float32)
    n = np.clip(np.rint(rows * inv[:, None]).astype(np.float32), -127, 127)
    return (n + 128).astype(np.uint8), sc


def u8_score(u, sc, q):
    """kernels_scan8.h::scan8_kernel: w = s * (sum u_i q_i - 128 sum q_i), float32 accumulation."""
    q = q.astype(np.float32)
    s = (u.astype(np.float32) * q[None, :]).sum(axis=1, dtype=np.float32)
    return (sc * (s - np.float32(128.0) * q.sum(dtype=np.float32))).astype(np.float32)


def u8_bound(sc, q):
    q1 = np.float32(np.abs(q.astype(np.float32)).sum(dtype=np.float32)) * np.float32(1.0 + 1e-5)
    return (np.float32(0.51) * sc * q1).astype(np.float32)


def to_bf16(x):
    """round to nearest even to 8 significant bits (what v_cvt_pk_bf16_f32 / the (__bf16) cast do)."""
    b = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    b = (b + 0x7FFF + ((b >> 16) & 1)) & 0xFFFF0000
    return b.astype(np.uint32).view(np.float32)


def _datasets(rng, n, d):
    yield "normal", rng.standard_normal((n, d))
    yield "unit", (lambda x: x / np.linalg.norm(x, axis=1, keepdims=True))(rng.standard_normal((n, d)))
    yield "heavy_tail", rng.standard_t(1.5, size=(n, d))
    yield "one_outlier_element", np.concatenate([rng.standard_normal((n, d - 1)), 80.0 * rng.standard_normal((n, 1))], axis=1)
    yield "tiny", 1e-30 * rng.standard_normal((n, d))
    yield "denormal", 1e-41 * rng.standard_normal((n, d))
    yield "mixed_scales", rng.standard_normal((n, d)) * np.float32(10.0) ** rng.integers(-38, 30, size=(n, 1))
    yield "constant_rows", np.repeat(rng.standard_normal((n, 1)), d, axis=1)
    mid = (rng.integers(-126, 127, size=(n, d)) + 0.5) / 127.0   # every element on a rounding midpoint of its row's grid
    mid[:, 0] = 1.0
    yield "midpoints", mid
    yield "sparse", rng.standard_normal((n, d)) * (rng.random((n, d)) < 0.02)


@pytest.mark.parametrize("d", [54, 128, 384, 4096])
def test_u8_row_bound_holds(d):
    rng = np.random.default_rng(d)
    n = 400
    for qname, q in (("normal", rng.standard_normal(d)), ("ones", np.ones(d)), ("alternating", (-1.0) ** np.arange(d)),
                     ("one_hot", np.eye(d)[3]), ("huge", 1e6 * rng.standard_normal(d))):
        for name, rows in _datasets(rng, n, d):
            rows = rows.astype(np.float32)
            u, sc = quantise_u8(rows)
            w = u8_score(u, sc, q).astype(np.float64)
            exact = rows.astype(np.float64) @ q.astype(np.float32).astype(np.float64)
            m = u8_bound(sc, q).astype(np.float64)
            slack = m - np.abs(w - exact)
            assert np.all(slack >= 0), (d, qname, name, float(slack.min()), float(m.max()))


@pytest.mark.parametrize("d", [64, 384, 3072])
def test_bf16_dot_bound_holds(d):
    rng = np.random.default_rng(1000 + d)
    n = 300
    nu = (d + 18) * 2.0 ** -24
    gamma = 1.02 * nu / (1 - nu)
    eps = (gamma + 1.05 * 2.0 ** -7) * 1.0102          # host_index.h: sel_eps
    for name, rows in _datasets(rng, n, d):
        rows = rows.astype(np.float32)
        for q in (rng.standard_normal(d), np.ones(d), rng.standard_t(1.5, size=d)):
            q = q.astype(np.float32)
            rb, qb = to_bf16(rows), to_bf16(q)
            dot = np.zeros(n, np.float32)
            for i in range(0, d, 16):                    # fp32 accumulation in blocks, as the MFMA chain does
                dot = (dot + (rb[:, i:i + 16] * qb[None, i:i + 16]).sum(axis=1, dtype=np.float32)).astype(np.float32)
            exact = rows.astype(np.float64) @ q.astype(np.float64)
            cn, qn = np.linalg.norm(rows.astype(np.float64), axis=1), np.linalg.norm(q.astype(np.float64))
            floor_abs = 7.5e-37 * np.sqrt(d)                 # host_index.h: sel_floor (operands in the denormal range)
            bound = eps * cn * qn * 1.0001 + floor_abs * (cn + qn)
            assert np.all(np.abs(dot.astype(np.float64) - exact) <= bound + 1e-300), (d, name)


@pytest.mark.parametrize("seed", range(6))
def test_two_phase_selection_never_loses_a_top_k_row(seed):
    """PHASE 0 / PHASE 1 of scan8_kernel restated: whatever rows are sampled, every true top-k row is a candidate."""
    rng = np.random.default_rng(50 + seed)
    n, d, k = 20_000, 96, int(rng.choice([1, 10, 50]))
    style = seed % 3
    rows = rng.standard_normal((n, d)).astype(np.float32)
    if style == 1:
        rows *= rng.lognormal(0, 1.5, size=(n, 1)).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    if style == 2:                                          # a tight cluster around the query, far below one quantisation step apart
        idx = rng.choice(n, 300, replace=False)
        rows[idx] = (q * 2.0 + 1e-3 * rng.standard_normal((300, d))).astype(np.float32)
    u, sc = quantise_u8(rows)
    w, m = u8_score(u, sc, q).astype(np.float64), u8_bound(sc, q).astype(np.float64)
    exact = rows.astype(np.float64) @ q.astype(np.float64)
    top = np.argsort(-exact, kind="stable")[:k]
    for sample_frac in (1 / 32, 1 / 4, 1.0):
        groups = np.arange(n) // 64
        sampled = rng.random(groups.max() + 1) < max(sample_frac, 8 * k / (groups.max() + 1))
        lows = [np.max((w - m)[groups == g]) for g in np.flatnonzero(sampled)]
        tau = np.sort(lows)[-k] if len(lows) >= k else -np.inf
        candidates = np.flatnonzero(~((w + m) < tau))
        assert set(top.tolist()) <= set(candidates.tolist()), (seed, sample_frac, k)
        assert len(candidates) < n or tau == -np.inf or style != 0   # the bound is selective on benign data


# ------------------------------------------------------------------------------------------------
# int8 tiles (kernels_tiles8.h): rows quantised with one scale per 64-row group, queries per query; the bound uses the
# ACTUAL residual norms:  |c.q - s_g s_q D| <= a_g E_q + b_g M_q
# ------------------------------------------------------------------------------------------------
def quantise_i8_groups(rows, group=64):
    """kernels_tiles8.h::rows_to_i8g_kernel (float32 arithmetic): per group s_g = max|c| / 127 over its finite rows,
    n = rint(c * (127 / max)), residual delta = c * inv - n; a_g = s_g max|n|_2 * 1.0002, b_g = s_g (max|delta|_2 * 1.0002
    + 2e-5 sqrt(d)).  Returns n (int8), and per ROW the group's (s_g, a_g, b_g)."""
    rows = rows.astype(np.float32)
    nrow, d = rows.shape
    n_out = np.zeros((nrow, d), np.int8)
    s_r, a_r, b_r = (np.zeros(nrow, np.float32) for _ in range(3))
    for g0 in range(0, nrow, group):
        blk = rows[g0:g0 + group]
        mx = np.float32(np.max(np.abs(blk))) if blk.size else np.float32(0)
        vanishing = mx < np.float32(1.2e-30)
        s_g = np.float32(2.0) * mx if vanishing else mx / np.float32(127.0)
        with np.errstate(over="ignore", divide="ignore", invalid="ignore"):
            inv = np.float32(0) if vanishing else np.float32(127.0) / mx
            x = np.clip(np.rint(blk * inv), -127, 127).astype(np.float32)
            res = (np.full_like(blk, 0.5 if mx > 0 else 0.0) if vanishing else blk * inv - x).astype(np.float32)
        n2 = np.sqrt((x * x).sum(axis=1, dtype=np.float32).max(initial=0)).astype(np.float32)
        d2 = np.sqrt((res * res).sum(axis=1, dtype=np.float32).max(initial=0)).astype(np.float32)
        n_out[g0:g0 + group] = x.astype(np.int8)
        s_r[g0:g0 + group] = s_g
        a_r[g0:g0 + group] = s_g * n2 * np.float32(1.0002)
        b_r[g0:g0 + group] = s_g * (d2 * np.float32(1.0002) + np.float32(2e-5) * np.float32(np.sqrt(d)))
    return n_out, s_r, a_r, b_r


def quantise_i8_query(q):
    """kernels_tiles8.h::queries_to_i8_kernel: m = rint(q * (127 / max|q|)), E = s_q (|eps|_2 * 1.0002 + 2e-5 sqrt(d) + 1e-6
    |m|_2 * 1.0002), M = s_q (|m|_2 * 1.0002 + |eps|_2 * 1.0002 + 2e-5 sqrt(d))."""
    q = q.astype(np.float32)
    d = q.shape[0]
    mx = np.float32(np.max(np.abs(q)))
    vanishing = mx < np.float32(1.2e-30)
    s_q = np.float32(2.0) * mx if vanishing else mx / np.float32(127.0)
    with np.errstate(over="ignore", divide="ignore", invalid="ignore"):
        inv = np.float32(0) if vanishing else np.float32(127.0) / mx
        m = np.clip(np.rint(q * inv), -127, 127).astype(np.float32)
        eps_v = (np.full_like(q, 0.5 if mx > 0 else 0.0) if vanishing else q * inv - m).astype(np.float32)
    eps = np.float32(np.sqrt((eps_v * eps_v).sum(dtype=np.float32))) * np.float32(1.0002) + np.float32(2e-5) * np.float32(np.sqrt(d))
    mm = np.float32(np.sqrt((m * m).sum(dtype=np.float32))) * np.float32(1.0002)
    return m.astype(np.int8), s_q, s_q * (eps + np.float32(1e-6) * mm), s_q * (mm + eps)


@pytest.mark.parametrize("d", [40, 128, 384, 1536])
def test_i8_tile_bound_holds(d):
    rng = np.random.default_rng(7000 + d)
    n = 448  # 7 groups
    for qname, q in (("normal", rng.standard_normal(d)), ("ones", np.ones(d)), ("one_hot", np.eye(d)[3]),
                     ("heavy", rng.standard_t(1.5, size=d)), ("tiny", 1e-33 * rng.standard_normal(d)),
                     ("midpoints", (rng.integers(-126, 127, size=d) + 0.5) / 127.0)):
        m, s_q, E, M = quantise_i8_query(q)
        for name, rows in _datasets(rng, n, d):
            rows = rows.astype(np.float32)
            nq, s_g, a_g, b_g = quantise_i8_groups(rows)
            D = nq.astype(np.int64) @ m.astype(np.int64)                       # exact integer dot product (the MFMA's)
            w = (s_g.astype(np.float64) * np.float64(s_q)) * D
            exact = rows.astype(np.float64) @ q.astype(np.float32).astype(np.float64)
            bound = a_g.astype(np.float64) * np.float64(E) + b_g.astype(np.float64) * np.float64(M)
            slack = bound - np.abs(w - exact)
            assert np.all(slack >= 0), (d, qname, name, float(slack.min()), float(bound.max()))


@pytest.mark.parametrize("seed", range(4))
def test_i8_two_phase_selection_never_loses_a_top_k_row(seed):
    """PHASE 0 / PHASE 1 of gemm_i8_kernel restated: per sampled group the lower bound from the group's largest D, the
    k-th largest of them as threshold, every row with D >= T(group, query) kept."""
    rng = np.random.default_rng(90 + seed)
    n, d, k = 32_000, 96, int(rng.choice([1, 10, 40]))
    rows = rng.standard_normal((n, d)).astype(np.float32)
    if seed % 2:
        rows *= rng.lognormal(0, 1.0, size=(n, 1)).astype(np.float32)
    rows = rows / np.linalg.norm(rows, axis=1, keepdims=True).astype(np.float32) if seed < 2 else rows
    q = rng.standard_normal(d).astype(np.float32)
    idx = rng.choice(n, 200, replace=False)
    rows[idx] = (q / np.linalg.norm(q) + 1e-3 * rng.standard_normal((200, d))).astype(np.float32)
    m, s_q, E, M = quantise_i8_query(q)
    nq, s_g, a_g, b_g = quantise_i8_groups(rows)
    D = (nq.astype(np.int64) @ m.astype(np.int64)).astype(np.float64)
    w = s_g.astype(np.float64) * float(s_q) * D
    bound = a_g.astype(np.float64) * float(E) + b_g.astype(np.float64) * float(M)
    exact = rows.astype(np.float64) @ q.astype(np.float64)
    top = np.argsort(-exact, kind="stable")[:k]
    groups = np.arange(n) // 64
    for frac in (1 / 32, 1 / 4):
        sampled = np.flatnonzero(rng.random(groups.max() + 1) < max(frac, 8 * k / (groups.max() + 1)))
        lows = [np.max(w[groups == g]) - bound[g * 64] for g in sampled]
        tau = np.sort(lows)[-k] if len(lows) >= k else -np.inf
        candidates = np.flatnonzero(~((w + bound) < tau))
        assert set(top.tolist()) <= set(candidates.tolist()), (seed, frac, k)
        if seed < 2:
            assert len(candidates) < n // 10   # selective on normalised data
        # the kernel's fp32 threshold chain (gemm_i8_kernel PHASE 1: per-query values prepared once, three fmas per
        # (group, query), fp32 compare) must keep every row the exact inequality keeps, and hardly any more
        f32 = np.float32
        wq = f32(1.0) / f32(s_q)
        A = f32(f32(tau) * wq) if np.isfinite(tau) else f32(-np.inf)
        A1 = f32(A - f32(2e-6) * np.abs(A))
        E1, M1 = f32(f32(f32(E) * wq) * f32(1.000003)), f32(f32(f32(M) * wq) * f32(1.000003))
        inv = (f32(1.0) / s_g.astype(f32)).astype(f32)
        ai, bi = (a_g.astype(f32) * inv).astype(f32), (b_g.astype(f32) * inv).astype(f32)

        def fma(a, b, c):  # one rounding, as v_fma_f32
            return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)

        T = fma(-bi, M1, fma(-ai, E1, fma(inv, A1, f32(-1.0))))
        kept = np.flatnonzero(~(D.astype(f32) < T))
        assert set(candidates.tolist()) <= set(kept.tolist()), (seed, frac, k)
        assert len(kept) <= len(candidates) * 1.2 + 64


# ------------------------------------------------------------------------------------------------
# round 3: L2 on the int8 tiles, and the prefilter epilogue (kernels_tiles8.h)
# ------------------------------------------------------------------------------------------------
def _fma32(a, b, c):  # one rounding, as v_fma_f32
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(np.float32)


@pytest.mark.parametrize("seed", range(4))
def test_i8_l2_selection_never_loses_a_top_k_row(seed):
    """gemm_i8_kernel<METRIC = L2> restated: v(c) = 2 c.q - |c|^2 with the row's squared fp32 norm taken 1e-4 to the safe
    side.  PHASE 0: per sampled 32-row block  max_r (2 s_g s_q D_r - 1.0001 |c_r|^2) - 8e-7 |.| - 2 bound;  PHASE 1: keep row r
    iff D - u_r w_q + 2e-6 |D| >= T  with T the inner-product chain at tau / 2, u_r = 0.4999 |c_r|^2 / s_g, w_q = 0.999997 / s_q."""
    f32 = np.float32
    rng = np.random.default_rng(400 + seed)
    n, d, k = 16_384, 96, int(rng.choice([1, 10, 40]))
    rows = rng.standard_normal((n, d)).astype(f32)
    if seed % 2:
        rows *= rng.lognormal(0, 0.8, size=(n, 1)).astype(f32)      # norms differ a lot inside every 64-row group
    q = (rng.standard_normal(d) * (3.0 if seed >= 2 else 1.0)).astype(f32)
    idx = rng.choice(n, 100, replace=False)
    rows[idx] = (q + 1e-2 * rng.standard_normal((100, d))).astype(f32)   # near neighbours: the norm form cancels there
    m, s_q, E, M = quantise_i8_query(q)
    nq, s_g, a_g, b_g = quantise_i8_groups(rows)
    D = (nq.astype(np.int64) @ m.astype(np.int64)).astype(np.float64)
    cn = (rows.astype(f32) ** 2).sum(axis=1, dtype=f32)                   # row_sqnorm_kernel: fp32 sum of squares
    true_v = 2.0 * (rows.astype(np.float64) @ q.astype(np.float64)) - (rows.astype(np.float64) ** 2).sum(axis=1)
    top = np.argsort(-true_v, kind="stable")[:k]
    bound = (a_g.astype(f32) * f32(E) + b_g.astype(f32) * f32(M)).astype(f32)
    # PHASE 0 (every 8th block of 32 rows): lower bounds, rounded as the kernel does
    ss = (f32(2.0) * s_g.astype(f32) * f32(s_q)).astype(f32)
    v = (_fma32(ss, D.astype(f32), -(cn * f32(1.0001))) - f32(8e-7) * np.abs(ss * D.astype(f32))).astype(f32)
    lows = []
    for b0 in range(0, n, 32 * 8):
        blk = slice(b0, b0 + 32)
        lb = f32(np.max(v[blk])) - f32(2.0) * bound[b0] * f32(1.000001)
        assert lb <= np.max(true_v[blk]) + 1e-12 * abs(np.max(true_v[blk])), (seed, b0)   # a valid lower bound
        lows.append(lb)
    tau = f32(np.sort(lows)[-k]) if len(lows) >= k else f32(-np.inf)
    # PHASE 1: the kernel's chain
    wq = f32(1.0) / f32(s_q)
    A = f32(f32(0.5) * tau * wq) if np.isfinite(tau) else f32(-np.inf)
    A1 = f32(A - f32(2e-6) * np.abs(A))
    E1, M1 = f32(f32(f32(E) * wq) * f32(1.000003)), f32(f32(f32(M) * wq) * f32(1.000003))
    inv = (f32(1.0) / s_g.astype(f32)).astype(f32)
    T = _fma32(-(b_g.astype(f32) * inv), M1, _fma32(-(a_g.astype(f32) * inv), E1, _fma32(inv, A1, f32(-1.0))))
    u = (f32(0.5) * cn * f32(0.9999) * inv).astype(f32)
    f = (_fma32(-u, f32(wq * f32(0.999997)), D.astype(f32)) + f32(2e-6) * np.abs(D.astype(f32))).astype(f32)
    kept = np.flatnonzero(~(f < T))
    assert set(top.tolist()) <= set(kept.tolist()), (seed, k)
    must = np.flatnonzero(true_v >= float(tau))                             # every row that could matter is kept ...
    assert set(must.tolist()) <= set(kept.tolist())
    assert len(kept) < n // 4                                               # ... and the selection still selects


@pytest.mark.parametrize("seed", range(3))
def test_i8_prefilter_threshold_is_never_above_the_exact_one(seed):
    """The prefilter epilogue (VAR bit 8): U = A1 - a_ref E' - b_ref M' - 4e-6 (|A1| + a_ref E' + b_ref M') per query, test
    D >= e_inv U - 1 on ORDINARY groups (a_g <= a_ref, b_g <= b_ref).  It must pass every (row, query) the exact chain keeps."""
    f32 = np.float32
    rng = np.random.default_rng(77 + seed)
    n, d = 4096, 128
    rows = rng.standard_normal((n, d)).astype(f32)
    if seed == 0:
        rows /= np.linalg.norm(rows, axis=1, keepdims=True).astype(f32)
    nq, s_g, a_g, b_g = quantise_i8_groups(rows)
    fin_a, fin_b = a_g[::64], b_g[::64]
    a_ref = f32(fin_a[fin_a <= 1.5 * fin_a.mean()].max())                  # group_ref_kernel
    b_ref = f32(fin_b[fin_b <= 1.5 * fin_b.mean()].max())
    inv = (f32(1.0) / s_g.astype(f32)).astype(f32)
    ordinary = (a_g <= a_ref) & (b_g <= b_ref)
    assert ordinary.mean() > 0.9
    for _ in range(40):
        q = rng.standard_normal(d).astype(f32) * f32(10.0) ** rng.integers(-3, 3)
        m, s_q, E, M = quantise_i8_query(q)
        wq = f32(1.0) / f32(s_q)
        tau = f32(rng.standard_normal() * 0.3 * np.linalg.norm(q))
        A = f32(tau * wq)
        A1 = f32(A - f32(2e-6) * np.abs(A))
        E1, M1 = f32(f32(f32(E) * wq) * f32(1.000003)), f32(f32(f32(M) * wq) * f32(1.000003))
        T = _fma32(-(b_g.astype(f32) * inv), M1, _fma32(-(a_g.astype(f32) * inv), E1, _fma32(inv, A1, f32(-1.0))))
        U = f32(_fma32(-b_ref, M1, _fma32(-a_ref, E1, A1)) - f32(4e-6) * (np.abs(A1) + a_ref * E1 + b_ref * M1))
        Tp = _fma32(inv, U, f32(-1.0))
        assert np.all(Tp[ordinary] <= T[ordinary]), (seed, float((Tp - T)[ordinary].max()))



# ------------------------------------------------------------------------------------------------
# round 3: the second selection stage of the int8 tiles (kernels_tiles8.h::refine_pairs_kernel)
# ------------------------------------------------------------------------------------------------
def _refine_bounds(D, s_g, a_g, b_g, s_q, E, M, cn=None):
    """refine_bounds<METRIC> restated in float32: the row's own lower / upper bound from its integer dot product."""
    f32 = np.float32
    Df = D.astype(f32)
    err = (a_g.astype(f32) * f32(E) + b_g.astype(f32) * f32(M)).astype(f32)
    if cn is None:
        w = (s_g.astype(f32) * f32(s_q) * Df).astype(f32)
        lb = (w - err * f32(1.000001) - f32(4e-7) * np.abs(w)).astype(f32)
        ub = (w + err * f32(1.00001) + f32(4e-6) * (np.abs(w) + err) + s_g.astype(f32) * f32(s_q)).astype(f32)
    else:
        ss = (f32(2.0) * s_g.astype(f32) * f32(s_q)).astype(f32)
        w = (ss * Df).astype(f32)
        lb = (_fma32(ss, Df, -(cn * f32(1.0001))) - f32(8e-7) * np.abs(w) - f32(2.0) * err * f32(1.000001)).astype(f32)
        ub = ((w - cn * f32(0.9999)) + f32(2.0) * err * f32(1.00001) + f32(8e-6) * (np.abs(w) + err) + ss).astype(f32)
    return lb, ub


@pytest.mark.parametrize("metric", ["ip", "l2"])
@pytest.mark.parametrize("seed", range(4))
def test_i8_second_stage_bounds_hold_and_never_drop_a_top_k_row(seed, metric):
    """Every row's [lower, upper] holds its fp32 score (the value the exact pass computes, any summation order), so the
    k-th largest LOWER bound of a candidate set that holds the true top-k drops none of them -- ties included -- and on
    normalised data it drops most of what a sampled threshold lets through."""
    f32 = np.float32
    rng = np.random.default_rng(900 + seed)
    n, d, k = 16_384, 384 if seed == 0 else 96, int(rng.choice([1, 10, 40]))
    rows = rng.standard_normal((n, d)).astype(f32)
    if seed % 2:
        rows *= rng.lognormal(0, 0.8, size=(n, 1)).astype(f32)
    elif metric == "ip":
        rows /= np.linalg.norm(rows, axis=1, keepdims=True).astype(f32)
    q = rng.standard_normal(d).astype(f32)
    if metric == "ip" and seed % 2 == 0:
        q /= np.linalg.norm(q)
    idx = rng.choice(n, 60, replace=False)
    near = q if metric == "l2" else q / np.linalg.norm(q) * np.linalg.norm(rows[idx], axis=1).mean()
    rows[idx] = (near + 1e-3 * rng.standard_normal((60, d))).astype(f32)
    rows[idx[:5]] = rows[idx[5]]                                            # exact ties among the best
    m, s_q, E, M = quantise_i8_query(q)
    nq, s_g, a_g, b_g = quantise_i8_groups(rows)
    D = nq.astype(np.int64) @ m.astype(np.int64)
    assert np.abs(D).max() < 2 ** 23
    r64, q64 = rows.astype(np.float64), q.astype(np.float64)
    if metric == "ip":
        truth = r64 @ q64
        fp32_forms = [(rows * q).sum(axis=1, dtype=f32), (rows[:, ::-1] * q[::-1]).sum(axis=1, dtype=f32)]
        lb, ub = _refine_bounds(D, s_g, a_g, b_g, s_q, E, M)
    else:
        cn = (rows ** 2).sum(axis=1, dtype=f32)
        truth = 2.0 * (r64 @ q64) - (r64 ** 2).sum(axis=1)
        q2 = f32((q64 ** 2).sum())
        # the exact pass ranks by -sum (c - q)^2 in fp32 = v - |q|^2: compare in v-space
        fp32_forms = [-(((rows - q) ** 2).sum(axis=1, dtype=f32)) + q2]
        lb, ub = _refine_bounds(D, s_g, a_g, b_g, s_q, E, M, cn)
    tol = 1e-6 * (np.abs(truth) + 1.0) if metric == "l2" else 0.0         # (|q|^2 added back in fp32 above)
    for v in [truth] + [f.astype(np.float64) for f in fp32_forms]:
        assert np.all(lb.astype(np.float64) <= v + tol) and np.all(v - tol <= ub.astype(np.float64)), (seed, metric)
    top = np.argsort(-truth, kind="stable")[:k]
    # stage 1 let through everything above a loose (sampled) threshold; stage 2 takes tau2 from the candidates themselves
    loose = np.sort(truth)[-min(n, 32 * k)]
    cand = np.flatnonzero(ub.astype(np.float64) >= loose)
    assert set(top.tolist()) <= set(cand.tolist())
    tau2 = np.sort(lb[cand])[-k]
    kept = cand[~(ub[cand] < tau2)]
    assert set(top.tolist()) <= set(kept.tolist()), (seed, metric, k)
    kth = np.sort(truth)[-k]
    assert set(np.flatnonzero(truth >= kth).tolist()) <= set(kept.tolist())  # every tie of the k-th best stays
    if seed % 2 == 0 and metric == "ip":
        assert len(kept) <= max(4 * k, len(cand) // 3), (len(kept), len(cand))


def _block_kth_threshold(v, k, low_bit=8):
    """kernels_merge_select.h::block_kth_threshold restated: the largest prefix, bit by bit from the first bit in which the
    (non-zero) values differ down to `low_bit`, with at least k values >= it; 0 when fewer than k values are present."""
    v = np.asarray(v, np.uint64)
    nz = v[v != 0]
    if len(nz) < k:
        return 0
    mx, mn = int(nz.max()), int(nz.min())
    diff = mx ^ mn
    if diff == 0:
        return mx
    top = diff.bit_length() - 1
    prefix = 0 if top == 31 else mx & ~((2 << top) - 1)
    for bit in range(top, low_bit - 1, -1):
        cand = prefix | (1 << bit)
        if int((v >= cand).sum()) >= k:
            prefix = cand
    return prefix


def _f2ord(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return np.where(u >> 31, u ^ 0xFFFFFFFF, u ^ 0x80000000)


@pytest.mark.parametrize("seed", range(6))
def test_block_threshold_search_is_a_valid_and_tight_threshold(seed):
    """The threshold kernels (kth_score_kernel, refine_pairs_kernel) do not sort: they search the k-th largest ordered key
    bit by bit and stop at bit 8.  The result must have at least k values at or above it (a valid threshold), lie at most
    2^8 ulps below the true k-th largest, map back to a float that is not NaN, and be 0 exactly when fewer than k exist."""
    rng = np.random.default_rng(1234 + seed)
    n = int(rng.choice([5, 64, 1000, 9760]))
    k = int(rng.choice([1, 10, 100]))
    kind = seed % 3
    if kind == 0:
        x = (0.05 * rng.standard_normal(n) + 0.15).astype(np.float32)              # sampled lower bounds of unit vectors
    elif kind == 1:
        x = (rng.standard_normal(n) * 10.0 ** rng.integers(-30, 30, n)).astype(np.float32)   # both signs, every magnitude
    else:
        x = -np.abs(rng.standard_normal(n).astype(np.float32)) * np.float32(1e3)   # L2 ranking values: all negative
        x[: n // 3] = x[0]                                                          # many equal values
    v = _f2ord(x)
    v[rng.random(n) < 0.2] = 0                                                      # absent entries
    t = _block_kth_threshold(v, k)
    present = np.sort(v[v != 0])[::-1]
    if len(present) < k:
        assert t == 0
        return
    kth = int(present[k - 1])
    assert int((v >= t).sum()) >= k and t <= kth and kth - t < (1 << 8), (seed, n, k, kth, t)
    u = np.uint32(t ^ 0x80000000) if t & 0x80000000 else np.uint32(~np.uint32(t))
    f = np.array([u], np.uint32).view(np.float32)[0]
    assert not np.isnan(f) and f <= np.array([kth ^ 0x80000000 if kth & 0x80000000 else (~kth) & 0xFFFFFFFF], np.uint32).view(np.float32)[0]
