"""GPU parity tests of the individual HIP kernels, through the C ABI (-m gpu)."""
import numpy as np
import pytest

import lsspa_oracle as O

pytestmark = pytest.mark.gpu


def problem(seed, p, n, m):
    rng = np.random.default_rng(seed)
    Xa = rng.standard_normal((n, p))
    Xe = rng.standard_normal((m, p))
    w = rng.standard_normal(p)
    return Xa, Xe, Xa @ w + rng.standard_normal(n), Xe @ w + rng.standard_normal(m)


@pytest.mark.parametrize("f32", [False, True])
def test_mfma_lane_maps(engine, f32):
    """Pins the operand / result lane maps of v_mfma_f64_16x16x4_f64 and v_mfma_f32_16x16x4_f32
    (they differ in the result rows) with asymmetric small-integer data (exact in both types)."""
    rng = np.random.default_rng(1)
    A = rng.integers(-8, 9, (16, 4)).astype(np.float64)
    B = rng.integers(-8, 9, (4, 16)).astype(np.float64)
    D = engine.mfma_probe(A, B, f32=f32)
    np.testing.assert_array_equal(D, A @ B)


@pytest.mark.parametrize("p,n,m", [(12, 60, 50), (100, 400, 300), (200, 500, 260), (70, 300, 40)])
def test_gram_reduction(engine, p, n, m):
    Xa, Xe, ya, ye = problem(3, p, n, m)
    engine.load_data(Xa, Xe, ya, ye, 0.25)
    G, g, H, h = engine.gram()
    np.testing.assert_allclose(G, Xa.T @ Xa / n + 0.25 * np.eye(p), rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(g, Xa.T @ ya / n, rtol=1e-13, atol=1e-13)
    assert engine.tri == (m >= p)
    if m >= p:
        np.testing.assert_allclose(H, Xe.T @ Xe, rtol=1e-13, atol=1e-12)
        np.testing.assert_allclose(h, Xe.T @ ye, rtol=1e-13, atol=1e-12)
    assert abs(engine.y_norm_sq - ye @ ye) <= 1e-12 * (ye @ ye)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("p,n,m", [(1, 40, 33), (127, 300, 290), (131, 333, 301), (255, 700, 650), (257, 513, 511),
                                   (300, 1000, 17), (383, 450, 449)])
def test_gram_ragged_tiles_and_rows(engine, p, n, m, dt):
    """The guarded loads of the Gram kernel: odd p (a 16-byte vector straddles the end of the features or the y
    column), p + 1 a multiple of 128 (the y column is the last column of a full-width tile), tiles past the
    features, row counts that are no multiple of the 16-row chunk or of the slice, both data types.  The guarded
    chunks are loaded unconditionally (clamped rows, y for every column at or beyond p) and masked after the
    products -- every value that must read as zero is checked through the sums it would otherwise corrupt."""
    Xa, Xe, ya, ye = problem(23, p, n, m)
    Xa, Xe, ya, ye = (np.ascontiguousarray(a, dtype=dt) for a in (Xa, Xe, ya, ye))
    engine.load_data(Xa, Xe, ya, ye, 0.125)
    G, g, H, h = engine.gram()
    # the other workgroup -> unit map (developer flag 65536: workgroup id = unit, no XCD-contiguous ranges) computes
    # the same slabs: bit-identical Gram matrices
    try:
        engine.set_flags(65536)
        engine.load_data(Xa, Xe, ya, ye, 0.125)
        for a, b in zip(engine.gram(), (G, g, H, h)):
            np.testing.assert_array_equal(a, b)
    finally:
        engine.set_flags(0)
    A64, E64, a64, e64 = (a.astype(np.float64) for a in (Xa, Xe, ya, ye))
    np.testing.assert_allclose(G, A64.T @ A64 / n + 0.125 * np.eye(p), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(g, A64.T @ a64 / n, rtol=1e-12, atol=1e-12)
    if m >= p:
        np.testing.assert_allclose(H, E64.T @ E64, rtol=1e-12, atol=1e-11)
        np.testing.assert_allclose(h, E64.T @ e64, rtol=1e-12, atol=1e-11)
    assert abs(engine.y_norm_sq - e64 @ e64) <= 1e-12 * (e64 @ e64)


@pytest.mark.parametrize("p,n,m", [(12, 60, 50), (100, 400, 300), (200, 500, 260)])
def test_cholesky_factor(engine, p, n, m):
    Xa, Xe, ya, ye = problem(4, p, n, m)
    engine.load_data(Xa, Xe, ya, ye, 0.0)
    rng = np.random.default_rng(0)
    perm = rng.permutation(p)
    L, Lt, V = engine.debug_factor(perm)
    G = (Xa.T @ Xa / n)[np.ix_(perm, perm)]
    g = (Xa.T @ ya / n)[perm]
    Lref = np.linalg.cholesky(G)
    np.testing.assert_allclose(np.tril(L[:p, :p]), Lref, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(L[p, :p], np.linalg.solve(Lref, g), rtol=1e-10, atol=1e-12)
    H = (Xe.T @ Xe)[np.ix_(perm, perm)]
    Ltref = np.linalg.cholesky(H)
    np.testing.assert_allclose(np.tril(Lt[:p, :p]), Ltref, rtol=1e-11, atol=1e-11)
    Vref = np.linalg.solve(Lref, Ltref)
    np.testing.assert_allclose(V[:p, :p], Vref, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("p,n,m,anti", [(12, 60, 50, False), (12, 60, 50, True), (100, 400, 300, True),
                                        (200, 500, 260, True), (70, 300, 40, True), (130, 300, 131, False)])
def test_lift_batch_vs_oracle(engine, p, n, m, anti):
    Xa, Xe, ya, ye = problem(5, p, n, m)
    engine.load_data(Xa, Xe, ya, ye, 0.0)
    red = O.reduce(Xa, Xe, ya, ye, 0.0)
    yy = float(ye @ ye)
    rng = np.random.default_rng(2)
    perms = np.array([rng.permutation(p) for _ in range(9)])
    got = engine.run_batch(perms, anti, want_lifts=True, accumulate=False)
    want = np.array([O.sample_lift(*red, yy, o, anti) for o in perms])
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-11)


def test_full_fit_and_stats(engine):
    p, n, m = 100, 400, 300
    Xa, Xe, ya, ye = problem(6, p, n, m)
    engine.load_data(Xa, Xe, ya, ye, 0.1)
    theta, r2, info = engine.full_fit()
    assert info == 0
    G = Xa.T @ Xa / n + 0.1 * np.eye(p)
    th = np.linalg.solve(G, Xa.T @ ya / n)
    np.testing.assert_allclose(theta, th, rtol=1e-10, atol=1e-12)
    r2_ref = 1 - np.sum((ye - Xe @ th) ** 2) / (ye @ ye)
    assert abs(r2 - r2_ref) < 1e-11
    rng = np.random.default_rng(3)
    engine.reset_stats()
    all_l = []
    for b in range(3):
        perms = np.array([rng.permutation(p) for _ in range(7 + b)])
        all_l.append(engine.run_batch(perms, True, want_lifts=True, accumulate=True))
        engine.merge()
    all_l = np.concatenate(all_l)
    cnt, mean, cov = engine.stats()
    assert cnt == len(all_l)
    np.testing.assert_allclose(mean, all_l.mean(0), rtol=0, atol=1e-14)
    np.testing.assert_allclose(cov, np.cov(all_l, rowvar=False, bias=True), rtol=0, atol=1e-15)


# ---------------------------------------------------------------- fp32 mode
@pytest.mark.parametrize("p,n,m,anti,reg", [(12, 60, 50, True, 0.0), (100, 400, 300, True, 0.0),
                                            (200, 500, 260, False, 1e-2), (70, 300, 40, True, 0.0),
                                            (257, 900, 700, True, 1e-2)])
def test_fp32_lift_batch_vs_oracle(engine, p, n, m, anti, reg):
    """fp32 work matrices (fp64 Gram reduction, fp64 lift accumulation): stated tolerance 1e-4 absolute
    on well-conditioned data (SURVEY.md 8c for the fp32 configuration); observed ~1e-6."""
    Xa, Xe, ya, ye = problem(5, p, n, m)
    red = O.reduce(Xa, Xe, ya, ye, reg)
    yy = float(ye @ ye)
    rng = np.random.default_rng(2)
    perms = np.array([rng.permutation(p) for _ in range(9)])
    want = np.array([O.sample_lift(*red, yy, o, anti) for o in perms])
    engine.set_precision("float32")
    try:
        engine.load_data(Xa, Xe, ya, ye, reg)
        got = engine.run_batch(perms, anti, want_lifts=True, accumulate=False)
        theta, r2, info = engine.full_fit()
    finally:
        engine.set_precision("float64")
    assert info == 0
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-4)
    assert np.abs(got - want).max() < 2e-5 * max(1.0, np.abs(want).max() * p)
    G = Xa.T @ Xa / n + reg * np.eye(p)
    np.testing.assert_allclose(theta, np.linalg.solve(G, Xa.T @ ya / n), rtol=2e-3, atol=2e-4)
    # and the fp64 path is untouched after switching back
    engine.load_data(Xa, Xe, ya, ye, reg)
    np.testing.assert_allclose(engine.run_batch(perms, anti, want_lifts=True, accumulate=False), want,
                               rtol=0, atol=1e-11)


# ---------------------------------------------------------------- block-boundary shapes
@pytest.mark.parametrize("p", [1, 2, 9, 63, 64, 65, 127, 128, 129, 191, 192])
@pytest.mark.parametrize("m_rel", ["tri", "rect"])
def test_block_boundary_shapes(engine, p, m_rel):
    """p around the 64 / 128 tile edges (p_pad = round_up(p + 1, 64) changes at p = 63/64, the strip
    count at 128/129, an odd block count at 65..128), both device paths."""
    n = max(3 * p, 40)
    m = n if m_rel == "tri" else max(p - 3, 1)
    if m_rel == "rect" and p == 1:
        pytest.skip("M < p impossible for p = 1")
    Xa, Xe, ya, ye = problem(50 + p, p, n, m)
    engine.load_data(Xa, Xe, ya, ye, 0.0)
    assert engine.tri == (m >= p)
    red = O.reduce(Xa, Xe, ya, ye, 0.0)
    yy = float(ye @ ye)
    rng = np.random.default_rng(p)
    perms = np.array([np.arange(p), np.arange(p)[::-1]] + [rng.permutation(p) for _ in range(3)])
    for anti in (False, True):
        got = engine.run_batch(perms, anti, want_lifts=True, accumulate=False)
        want = np.array([O.sample_lift(*red, yy, o, anti) for o in perms])
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-10)
    theta, r2, info = engine.full_fit()
    assert info == 0
    np.testing.assert_allclose(theta, np.linalg.lstsq(Xa, ya, rcond=None)[0], rtol=1e-8, atol=1e-10)
    assert abs(r2 - want[0].sum()) < 1e-10      # every ordering's lifts sum to the full-model R^2


# ---------------------------------------------------------------- lift history, device-side error estimator
@pytest.mark.parametrize("p,n_batches,bsz", [(12, 3, 7), (100, 2, 16), (130, 5, 13), (300, 1, 9)])
def test_history_and_error_estimator(engine, p, n_batches, bsz):
    """The thin-form estimator on the device against the same arithmetic in numpy, same Xi.
    Sample counts that are not multiples of the 16-row chunk, p across the 128-column tile edge."""
    Xa, Xe, ya, ye = problem(8, p, 3 * p + 20, 2 * p + 11)
    engine.load_data(Xa, Xe, ya, ye, 0.0)
    engine.history_enable(16)   # smaller than what follows: the history has to grow
    rng = np.random.default_rng(4)
    lifts = []
    for _ in range(n_batches):
        perms = np.array([rng.permutation(p) for _ in range(bsz)])
        lifts.append(engine.run_batch(perms, True, want_lifts=True, accumulate=True))
        engine.merge()
    lifts = np.concatenate(lifts)
    n = len(lifts)
    assert engine.history_count() == n
    np.testing.assert_array_equal(engine.history(), lifts)
    _, mean, _ = engine.stats(want_cov=False)
    xi = rng.standard_normal((1024, n))
    engine.error_draws(xi, n)
    feat, tot = engine.error_quantiles()
    draws = (xi @ (lifts - mean)) / np.sqrt(n * (n - 1.0))
    np.testing.assert_allclose(feat, np.quantile(np.abs(draws), 0.95, axis=0), rtol=1e-10, atol=1e-18)
    np.testing.assert_allclose(tot, np.quantile(np.linalg.norm(draws, axis=1), 0.95), rtol=1e-11)
    # a second estimate after more samples reuses the buffers
    perms = np.array([rng.permutation(p) for _ in range(5)])
    more = engine.run_batch(perms, True, want_lifts=True, accumulate=True)
    with pytest.raises(Exception, match="merge"):
        engine.error_draws(rng.standard_normal((1024, n + 5)), n + 5)
    engine.merge()
    lifts = np.concatenate([lifts, more])
    n = len(lifts)
    _, mean, _ = engine.stats(want_cov=False)
    xi = rng.standard_normal((1024, n))
    engine.error_draws(xi, n)
    feat, tot = engine.error_quantiles()
    draws = (xi @ (lifts - mean)) / np.sqrt(n * (n - 1.0))
    np.testing.assert_allclose(feat, np.quantile(np.abs(draws), 0.95, axis=0), rtol=1e-10, atol=1e-18)
    np.testing.assert_allclose(tot, np.quantile(np.linalg.norm(draws, axis=1), 0.95), rtol=1e-11)
    with pytest.raises(Exception, match="n_local"):
        engine.error_draws(xi[:, :-1], n)
    engine.reset_stats()
    assert engine.history_count() == 0
    engine.history_enable(0)


@pytest.mark.parametrize("p,chunks,stride", [(12, (7, 9, 16), 1), (100, (16, 33), 2), (130, (13, 5, 64, 1), 3),
                                             (300, (9, 40), 1)])
def test_running_error_estimator(engine, p, chunks, stride):
    """The running form (lsspa_error_running_* / _advance / _quantiles_enqueue / _result): normals made on the device by
    Philox4x32-10 + Box-Muller against their NumPy restatement (tests/philox_ref.py, itself pinned by the generator's
    known-answer vectors), D = Xi L and s = Xi 1 accumulated chunk by chunk, and a check's quantiles against the same
    arithmetic in NumPy.  Chunk sizes that are not multiples of the 16-row GEMM chunk, p across the 128-column tile
    edge, sample ids with a stride (a rank's share of a dealt chunk)."""
    import philox_ref as P
    seed = 0x9E3779B97F4A7C15
    xi = engine.error_xi(seed, 5, stride, 37)
    want = P.normals(seed, 5 + stride * np.arange(37))
    np.testing.assert_allclose(xi, want, rtol=0, atol=1e-12)
    big = engine.error_xi(3, 2 ** 40, 1, 64)          # ids beyond 32 bits use the counter's second word
    np.testing.assert_allclose(big, P.normals(3, 2 ** 40 + np.arange(64)), rtol=0, atol=1e-12)
    Xa, Xe, ya, ye = problem(8, p, 3 * p + 20, 2 * p + 11)
    engine.load_data(Xa, Xe, ya, ye, 0.0)
    engine.reset_stats()
    engine.error_running_enable(seed)
    rng = np.random.default_rng(4)
    lifts, ids, nxt = [], [], 3
    for k, b in enumerate(chunks):
        perms = np.array([rng.permutation(p) for _ in range(b)])
        lifts.append(engine.run_batch(perms, True, want_lifts=True, accumulate=2))
        engine.error_advance(nxt, stride)
        ids.append(nxt + stride * np.arange(b))
        nxt += stride * b + 1
        L, I = np.concatenate(lifts), np.concatenate(ids)
        n = len(L)
        engine.error_running_draws(n)
        engine.error_quantiles_enqueue(k)
        feat, tot, mean, n_dev = engine.error_result(k, wait=True)
        assert n_dev == n
        np.testing.assert_allclose(mean, L.mean(0), rtol=0, atol=1e-14)
        draws = P.normals(seed, I) @ (L - L.mean(0)) / np.sqrt(n * (n - 1.0))
        np.testing.assert_allclose(feat, np.quantile(np.abs(draws), 0.95, axis=0), rtol=1e-9, atol=1e-16)
        np.testing.assert_allclose(tot, np.quantile(np.linalg.norm(draws, axis=1), 0.95), rtol=1e-10)
    # every earlier slot still holds its own check (nothing waits, nothing is overwritten)
    f0, t0, m0, n0 = engine.error_result(0, wait=False)
    assert n0 == chunks[0]
    np.testing.assert_allclose(m0, lifts[0].mean(0), rtol=0, atol=1e-14)
    # state out and back in (checkpoint / resume): the next check is unchanged
    D, s_vec = engine.error_state()
    Xi = P.normals(seed, I)
    np.testing.assert_allclose(D, Xi @ L, rtol=0, atol=1e-11 * max(1.0, np.abs(L).max()) * np.sqrt(n))
    np.testing.assert_allclose(s_vec, Xi.sum(1), rtol=0, atol=1e-11)
    engine.reset_stats()
    D0, s0 = engine.error_state()
    assert not D0.any() and not s0.any()
    engine.set_error_state(D, s_vec)
    D1, s1 = engine.error_state()
    np.testing.assert_array_equal(D1, D)
    np.testing.assert_array_equal(s1, s_vec)
    with pytest.raises(Exception, match="advance"):
        engine.run_batch(perms, True, want_lifts=False, accumulate=2)
        engine.error_running_draws(n + len(perms))
    engine.reset_stats()
    engine.history_enable(0)
    with pytest.raises(Exception, match="not enabled"):
        engine.error_advance(0, 1)


def test_stats_checkpoint_resume(engine):
    """lsspa_stats_set + lsspa_history_append: stop after two batches, restore into a fresh problem
    load, continue -- same statistics as the uninterrupted run."""
    p = 60
    Xa, Xe, ya, ye = problem(9, p, 200, 150)
    rng = np.random.default_rng(5)
    batches = [np.array([rng.permutation(p) for _ in range(8)]) for _ in range(3)]
    engine.load_data(Xa, Xe, ya, ye, 0.01)
    engine.history_enable(64)
    for b in batches:
        engine.run_batch(b, True, accumulate=True)
        engine.merge()
    n_ref, mean_ref, cov_ref = engine.stats()
    hist_ref = engine.history()

    engine.load_data(Xa, Xe, ya, ye, 0.01)
    engine.history_enable(64)
    for b in batches[:2]:
        engine.run_batch(b, True, accumulate=True)
        engine.merge()
    saved = engine.stats()
    saved_hist = engine.history()
    engine.load_data(Xa, Xe, ya, ye, 0.01)     # "new process"
    engine.history_enable(8)
    engine.set_stats(*saved)
    engine.history_append(saved_hist)
    engine.run_batch(batches[2], True, accumulate=True)
    engine.merge()
    n, mean, cov = engine.stats()
    assert n == n_ref == 24
    np.testing.assert_allclose(mean, mean_ref, rtol=0, atol=1e-15)
    np.testing.assert_allclose(cov, cov_ref, rtol=0, atol=1e-16)
    np.testing.assert_array_equal(engine.history(), hist_ref)
    engine.history_enable(0)


@pytest.mark.parametrize("p,n,m,prec", [(100, 400, 300, "float64"), (130, 400, 300, "float64"), (255, 900, 700, "float64"),
                                        (256, 900, 700, "float64"), (257, 900, 700, "float64"),
                                        (257, 900, 700, "float32"), (383, 1200, 1100, "float64"),
                                        (640, 2000, 1800, "float32"), (1000, 3000, 2500, "float64")])
def test_vt_tiles_agree_with_the_strip_kernel(p, n, m, prec):
    """Two independent computations of V = L^-1 L_t live in the library: as extra block rows of the training
    factorisation inside the panel launches (V^T, the shipped tri-mode path since round 4) and by the strip kernel
    (developer flag 128; the shipped path of rect mode).  Same factors in, same lifts to round-off -- and both meet
    the oracle.  The lift scan itself has two forms as well: fused into the X tiles (shipped) and as a kernel of its own
    over the stored V^T (flag 512).  The unpaired gather and the plain dispatch order of the panel kernel (developer
    flags of their own up to round 4) are what three orderings without antithetical partners take (six matrices: no
    multiple of eight): forward and reversed runs of them average to the paired, grouped result."""
    from ls_spa._engine import HipEngine
    Xa, Xe, ya, ye = problem(13, p, n, m)
    rng = np.random.default_rng(8)
    perms = np.array([rng.permutation(p) for _ in range(6)])
    f64 = prec == "float64"
    eng = HipEngine(0)
    try:
        eng.set_precision(prec)
        eng.load_data(Xa, Xe, ya, ye, 1e-3)
        eng.set_flags(1024)            # 1024: the general path also where the fused small-p kernel would run
        base = eng.run_batch(perms, True, want_lifts=True, accumulate=False)
        # 512: the lift kernel reads V^T back and scans it (the X tiles scan their own blocks otherwise)
        for flags in (128, 512, 128 | 512):
            eng.set_flags(flags | 1024)
            other = eng.run_batch(perms, True, want_lifts=True, accumulate=False)
            np.testing.assert_allclose(other, base, rtol=0, atol=5e-13 if f64 else 2e-5)
        eng.set_flags(1024)
        fwd = eng.run_batch(perms[:3], False, want_lifts=True, accumulate=False)
        rev = eng.run_batch(np.ascontiguousarray(perms[:3, ::-1]), False, want_lifts=True, accumulate=False)
        np.testing.assert_allclose(0.5 * (fwd + rev), base[:3], rtol=0, atol=5e-13 if f64 else 2e-5)
        assert eng.info() == 0
        eng.set_flags(1024)
        L, Lt, V = eng.debug_factor(perms[0])
        eng.set_flags(1024 | 128)
        L2, Lt2, V2 = eng.debug_factor(perms[0])
        np.testing.assert_array_equal(L, L2)
        np.testing.assert_array_equal(Lt, Lt2)
        scale = np.abs(V2[:p, :p]).max()
        np.testing.assert_allclose(V[:p, :p], V2[:p, :p], rtol=0, atol=(1e-12 if f64 else 1e-4) * scale)
    finally:
        eng.close()
    if p <= 400:
        red = O.reduce(Xa, Xe, ya, ye, 1e-3)
        want = np.array([O.sample_lift(*red, float(ye @ ye), o, True) for o in perms])
        np.testing.assert_allclose(base, want, rtol=0, atol=1e-11 if f64 else 1e-4)


def test_a_hand_over_that_never_comes_times_out_and_is_reported():
    """The fused lift scan waits, inside a panel launch, for row p of its panel from another workgroup of the same
    launch.  That wait must end whatever happens: with the flags muted (developer flag 4096) every X tile of the first two
    launches runs into the time-out, the launches finish, LSSPA_INFO_SCAN_WAIT (4) is set -- and the engine is fine
    afterwards.  Round 5: the scan then went on with whatever z and y~ it found, and ONLY the time-out said so; a flag
    that arrives with stale data would say nothing.  Every ordering's lifts sum to the full model's R^2
    (ls_spa/ls_spa.py:284-285), and once lsspa_full_fit has computed it every batch is checked against it on the device:
    LSSPA_INFO_SUM (8) catches the garbage on its own."""
    import time
    from ls_spa._engine import HipEngine
    p, n, m = 257, 900, 700
    Xa, Xe, ya, ye = problem(23, p, n, m)
    rng = np.random.default_rng(3)
    perms = np.array([rng.permutation(p) for _ in range(2)])
    eng = HipEngine(0)
    try:
        eng.load_data(Xa, Xe, ya, ye, 1e-3)
        good = eng.run_batch(perms, True, want_lifts=True, accumulate=False)
        assert eng.info() == 0
        assert eng.sum_deviation() == 0.0          # no R^2 known yet: nothing was checked
        _, r2, _ = eng.full_fit()
        eng.run_batch(perms, True, want_lifts=True, accumulate=False)
        assert eng.info() == 0 and 0.0 < eng.sum_deviation() + 1e-300 < 1e-12
        np.testing.assert_allclose(good.sum(1), r2, rtol=0, atol=1e-12)
        eng.set_flags(4096)
        t0 = time.perf_counter()
        bad = eng.run_batch(perms, True, want_lifts=True, accumulate=False)
        waited = time.perf_counter() - t0
        bits = eng.info()
        assert bits & 4 and bits & 8, bits
        assert eng.sum_deviation() == pytest.approx(np.abs(bad.sum(1) - r2).max(), rel=1e-9)
        assert waited < 20.0
        eng.set_flags(0)
        eng.load_data(Xa, Xe, ya, ye, 1e-3)      # (loading a problem clears the info word)
        again = eng.run_batch(perms, True, want_lifts=True, accumulate=False)
        assert eng.info() == 0
        np.testing.assert_array_equal(again, good)
        # the public call raises on either bit
        from ls_spa import ls_spa
        from ls_spa._native import LSSPANativeError

        class Muted(HipEngine):
            def reset_stats(self):
                super().reset_stats()
                self.set_flags(4096)
        e2 = Muted(0)
        try:
            with pytest.raises(LSSPANativeError, match="gave up waiting|did not sum"):
                ls_spa(Xa, Xe, ya, ye, reg=1e-3, perms=perms, batch_size=2, _engine=e2)
        finally:
            e2.close()
        # a source that can be drawn again (a QMC method here): the run is repeated once on the conservative path (the
        # lift kernel of its own, flag 512) and comes back with the healthy run's numbers and a warning
        class MutedOnce(HipEngine):
            muted = 0

            def reset_stats(self):
                super().reset_stats()
                if not self.muted:
                    self.set_flags(4096)
                self.muted += 1
        kq = dict(reg=1e-3, method="argsort", seed=3, batch_size=4, max_samples=8, tolerance=0.0)
        healthy = ls_spa(Xa, Xe, ya, ye, **kq)
        e3 = MutedOnce(0)
        try:
            with pytest.warns(RuntimeWarning, match="repeated"):
                again2 = ls_spa(Xa, Xe, ya, ye, _engine=e3, **kq)
            assert e3.muted == 2
            np.testing.assert_allclose(again2.attribution, healthy.attribution, rtol=0, atol=1e-12)
            np.testing.assert_allclose(again2.error_history, healthy.error_history, rtol=1e-6)
        finally:
            e3.close()
        # ... and the small-problem kernels are under the same check (the check itself, on a healthy engine)
        Xs, Xt, ys, yt = problem(5, 40, 300, 200)
        eng.load_data(Xs, Xt, ys, yt, 0.0)
        eng.full_fit()
        eng.reset_stats()
        eng.run_batch(np.array([rng.permutation(40) for _ in range(64)]), True, want_lifts=False, accumulate=2)
        assert eng.info() == 0 and eng.sum_deviation() < 1e-12
    finally:
        eng.close()


@pytest.mark.parametrize("anti", [False, True])
def test_batch_larger_than_the_workspace_is_split(engine, anti):
    """More orderings in one call than a launch sequence holds (4096): the batch is cut into sub-batches
    that share the pinned staging buffers; lifts and accumulated statistics must not notice."""
    p = 12
    Xa, Xe, ya, ye = problem(17, p, 80, 60)
    red = O.reduce(Xa, Xe, ya, ye, 0.0)
    yy = float(ye @ ye)
    rng = np.random.default_rng(9)
    n = 4500 if not anti else 2300
    perms = np.array([rng.permutation(p) for _ in range(n)])
    engine.load_data(Xa, Xe, ya, ye, 0.0)
    got = engine.run_batch(perms, anti, want_lifts=True, accumulate=True)
    engine.merge()
    idx = rng.choice(n, 60, replace=False)
    want = np.array([O.sample_lift(*red, yy, perms[i], anti) for i in idx])
    np.testing.assert_allclose(got[idx], want, rtol=0, atol=1e-11)
    cnt, mean, cov = engine.stats()
    assert cnt == n
    np.testing.assert_allclose(mean, got.mean(0), rtol=0, atol=1e-13)
    np.testing.assert_allclose(cov, np.cov(got, rowvar=False, bias=True), rtol=0, atol=1e-13)


def test_failed_allocation_leaves_the_context_usable():
    """A workspace growth that runs out of memory (second allocation fails: injected) raises MemoryError and must
    leave the context without a workspace -- not with the old capacity over freed buffers -- so that the next,
    smaller batch allocates afresh and computes the same lifts as before."""
    from ls_spa._engine import HipEngine
    d = O.gaussian_workload(40, 300, 200, seed=3)
    rng = np.random.default_rng(2)
    perms = np.array([rng.permutation(40) for _ in range(64)])
    eng = HipEngine(0)
    try:
        eng.load_data(*d, 0.0)
        want = eng.run_batch(perms[:4], True, want_lifts=True, accumulate=False)
        eng.debug_fail_alloc(2)
        with pytest.raises(MemoryError):
            eng.run_batch(perms, True, want_lifts=True, accumulate=False)
        got = eng.run_batch(perms[:4], True, want_lifts=True, accumulate=False)
        np.testing.assert_array_equal(got, want)
        eng.debug_fail_alloc(1)     # the lifts buffer's growth fails this time
        with pytest.raises(MemoryError):
            eng.run_batch(perms, True, want_lifts=True, accumulate=False)
        eng.debug_fail_alloc(0)
        full = eng.run_batch(perms, True, want_lifts=True, accumulate=False)
        np.testing.assert_array_equal(full[:4], want)
        # a new problem of another shape on the same context: the old workspace is released, not leaked
        d2 = O.gaussian_workload(70, 300, 200, seed=4)
        eng.load_data(*d2, 0.0)
        perms2 = np.array([rng.permutation(70) for _ in range(8)])
        got2 = eng.run_batch(perms2, False, want_lifts=True, accumulate=False)
        red = O.reduce(*d2, 0.0)
        want2 = np.array([O.ordering_lift(*red, float(d2[3] @ d2[3]), o) for o in perms2])
        np.testing.assert_allclose(got2, want2, rtol=0, atol=1e-10)
    finally:
        eng.close()


def test_two_lanes_same_results():
    """lsspa_set_lanes(2): successive batches alternate between two workspaces on two streams, staggered; the
    statistics stay in batch order.  Lift vectors and running statistics are those of one lane, bit for bit."""
    from ls_spa._engine import HipEngine
    d = O.gaussian_workload(150, 600, 500, seed=8)
    rng = np.random.default_rng(5)
    perms = np.array([rng.permutation(150) for _ in range(6 * 24)])
    res = {}
    for lanes in (1, 2):
        eng = HipEngine(0)
        try:
            eng.load_data(*d, 0.0)
            eng.set_lanes(lanes)
            lifts = []
            for k in range(6):
                lifts.append(eng.run_batch(perms[24 * k:24 * k + 24], True, want_lifts=(k % 2 == 0), accumulate=True))
                eng.merge()
            n, mean, cov = eng.stats()
            th, r2, info = eng.full_fit()
            res[lanes] = (n, mean, cov, [l for l in lifts if l is not None], th, r2)
            assert info == 0 and eng.info() == 0
        finally:
            eng.close()
    assert res[1][0] == res[2][0] == 144
    np.testing.assert_array_equal(res[1][1], res[2][1])
    np.testing.assert_array_equal(res[1][2], res[2][2])
    for a, b in zip(res[1][3], res[2][3]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(res[1][4], res[2][4])
    red = O.reduce(*d, 0.0)
    yy = float(d[3] @ d[3])
    want = np.array([O.sample_lift(*red, yy, o, True) for o in perms[:24]])
    np.testing.assert_allclose(res[2][3][0], want, rtol=0, atol=1e-10)


def test_launch_collect_discard():
    """The two halves of lsspa_lift_batch: a batch launched ahead and then discarded leaves no trace in the
    statistics; two may be in flight with two lanes, a third is refused; collect order = launch order."""
    from ls_spa._engine import HipEngine
    from ls_spa._native import LSSPANativeError
    d = O.gaussian_workload(60, 300, 200, seed=9)
    rng = np.random.default_rng(6)
    perms = np.array([rng.permutation(60) for _ in range(48)])
    eng = HipEngine(0)
    try:
        eng.load_data(*d, 0.0)
        ref_a = eng.run_batch(perms[:16], True, want_lifts=True, accumulate=False)
        ref_b = eng.run_batch(perms[16:32], True, want_lifts=True, accumulate=False)
        eng.set_lanes(2)
        t1 = eng.launch_batch(perms[:16], True)
        t2 = eng.launch_batch(perms[16:32], True)
        with pytest.raises(LSSPANativeError):
            eng.launch_batch(perms[32:], True)          # both lanes hold a batch
        a = eng.collect_batch(t1, want_lifts=True, accumulate=True)
        eng.merge()
        t3 = eng.launch_batch(perms[32:], True)         # lane of t1 is free again
        b = eng.collect_batch(t2, want_lifts=True, accumulate=True)
        eng.merge()
        eng.discard_batch(t3)
        with pytest.raises(LSSPANativeError):
            eng.collect_batch(t3)                       # nothing to collect any more
        np.testing.assert_array_equal(a, ref_a)
        np.testing.assert_array_equal(b, ref_b)
        n, mean, cov = eng.stats()
        both = np.concatenate([ref_a, ref_b])
        assert n == 32
        np.testing.assert_allclose(mean, both.mean(0), rtol=0, atol=1e-14)
        np.testing.assert_allclose(cov, np.cov(both, rowvar=False, bias=True), rtol=0, atol=1e-15)
        eng.set_lanes(1)
        c = eng.run_batch(perms[:16], True, want_lifts=True, accumulate=False)
        np.testing.assert_array_equal(c, ref_a)
    finally:
        eng.close()


def test_driver_lookahead_on_the_device():
    """Chunks of 8 samples launched four at a time: partial collects of one launched batch (lsspa_lift_collect with
    first / count), checks in the reference's order, the rest of the last group discarded at the stop."""
    from ls_spa import ls_spa
    d = O.gaussian_workload(40, 300, 200, seed=3)
    kw = dict(method="argsort", seed=11, max_samples=72, batch_size=8, tolerance=0.0, error_estimator="device")
    one = ls_spa(*d, **kw)
    four = ls_spa(*d, lookahead=4, **kw)
    np.testing.assert_array_equal(four.attribution, one.attribution)
    np.testing.assert_allclose(four.error_history, one.error_history, rtol=1e-12)
    waited = ls_spa(*d, _defer=0, **kw)          # every check waited for: the same numbers
    np.testing.assert_array_equal(waited.attribution, one.attribution)
    np.testing.assert_array_equal(waited.error_history, one.error_history)
    assert len(four.error_history) == 10      # 8 .. 64, 71, 72
    tol = float(one.error_history[2]) * 1.0000001
    if one.error_history[0] > tol and one.error_history[1] > tol:
        a = ls_spa(*d, **dict(kw, tolerance=tol))
        b = ls_spa(*d, lookahead=4, **dict(kw, tolerance=tol))
        assert len(a.error_history) == len(b.error_history) == 3
        np.testing.assert_array_equal(b.attribution, a.attribution)
    # attribution history: partial collects copy the chunk's lift vectors out
    h1 = ls_spa(*d, return_attribution_history=True, **dict(kw, error_estimator="reference"))
    h3 = ls_spa(*d, return_attribution_history=True, lookahead=3, **dict(kw, error_estimator="reference"))
    np.testing.assert_array_equal(h3.attribution_history, h1.attribution_history)
    # also through the lanes: two batches in flight
    from ls_spa._engine import HipEngine
    eng = HipEngine(0)
    try:
        eng.set_lanes(2)
        c = ls_spa(*d, lookahead=2, _engine=eng, **kw)
        np.testing.assert_array_equal(c.attribution, one.attribution)
    finally:
        eng.close()


@pytest.mark.parametrize("p,n,m", [(1, 30, 30), (2, 40, 30), (3, 50, 50), (9, 80, 60), (15, 90, 70), (16, 90, 70), (31, 150, 120), (47, 200, 160),
                                   (63, 300, 200), (79, 300, 250), (95, 350, 300), (100, 400, 300), (111, 420, 330),
                                   (112, 420, 330), (126, 500, 400), (127, 500, 400)])
def test_fused_small_p_kernel(p, n, m):
    """p + 1 <= 128: the fused per-ordering kernels (gather -> two Choleskys -> V -> lifts; p + 1 <= 112 with the
    matrices in registers, every block-row count from one to seven, beyond that in LDS) against the oracle, against each
    other (developer flag 16384 = the LDS kernel everywhere) and against the general multi-kernel path (flag 1024),
    single orderings and antithetical pairs."""
    from ls_spa._engine import HipEngine
    d = problem(20 + p, p, n, m)
    rng = np.random.default_rng(p)
    perms = np.array([rng.permutation(p) for _ in range(24)])
    perms[0] = np.arange(p)
    perms[1] = np.arange(p)[::-1]
    red = O.reduce(*d, 0.0)
    yy = float(d[3] @ d[3])
    eng = HipEngine(0)
    try:
        eng.load_data(*d, 0.0)
        eng.profile(True)
        single = eng.run_batch(perms, False, want_lifts=True, accumulate=False)
        paired = eng.run_batch(perms, True, want_lifts=True, accumulate=False)
        used = eng.profile_read()
        assert used["small_p"][1] == 2 and used["gather"][1] == 0, used      # the fused kernel ran, nothing else
        eng.profile(False)
        assert eng.info() == 0
        want1 = np.array([O.ordering_lift(*red, yy, o) for o in perms])
        want2 = np.array([O.sample_lift(*red, yy, o, True) for o in perms])
        np.testing.assert_allclose(single, want1, rtol=0, atol=1e-10)
        np.testing.assert_allclose(paired, want2, rtol=0, atol=1e-10)
        np.testing.assert_allclose(single.sum(1), single[0].sum(), rtol=0, atol=1e-11)   # every ordering: the full R^2
        eng.set_flags(1024)
        general = eng.run_batch(perms, True, want_lifts=True, accumulate=False)
        eng.set_flags(16384)
        in_lds = eng.run_batch(perms, True, want_lifts=True, accumulate=False)
        eng.set_flags(0)
        np.testing.assert_allclose(paired, general, rtol=0, atol=1e-12)
        np.testing.assert_allclose(paired, in_lds, rtol=0, atol=1e-12)
        # statistics through the fused path
        eng.reset_stats()
        eng.run_batch(perms, True, accumulate=True)
        eng.merge()
        nacc, mean, cov = eng.stats()
        assert nacc == 24
        np.testing.assert_allclose(mean, want2.mean(0), rtol=0, atol=1e-11)
        theta, r2, info = eng.full_fit()        # reads the factors back: takes the general path
        assert info == 0 and abs(r2 - single[0].sum()) < 1e-11
    finally:
        eng.close()


def test_fused_small_p_every_feature_count():
    """Every p from 1 to 127 once (block-row boundaries, ragged last blocks, the switch between the two fused kernels
    at p = 111 / 112 and to the general path at 127): antithetical pairs against the oracle."""
    from ls_spa._engine import HipEngine
    eng = HipEngine(0)
    try:
        for p in range(1, 128):
            d = problem(300 + p, p, 2 * p + 40, p + 50)
            rng = np.random.default_rng(1000 + p)
            perms = np.array([rng.permutation(p) for _ in range(4)])
            red = O.reduce(*d, 1e-3)
            yy = float(d[3] @ d[3])
            eng.load_data(*d, 1e-3)
            got = eng.run_batch(perms, True, want_lifts=True, accumulate=False)
            assert eng.info() == 0, p
            want = np.array([O.sample_lift(*red, yy, o, True) for o in perms])
            np.testing.assert_allclose(got, want, rtol=0, atol=1e-10, err_msg=f"p = {p}")
    finally:
        eng.close()


def test_fused_small_p_flags_collinear_features():
    from ls_spa._engine import HipEngine
    d = list(problem(5, 20, 200, 150))
    d[0] = d[0].copy()
    d[1] = d[1].copy()
    d[0][:, 7] = d[0][:, 3]       # feature 7 duplicates feature 3 in both sets
    d[1][:, 7] = d[1][:, 3]
    eng = HipEngine(0)
    try:
        eng.load_data(*d, 0.0)
        eng.run_batch(np.arange(20)[None, :], False, want_lifts=True, accumulate=False)
        assert eng.info() & 1      # LSSPA_INFO_NOT_PD, as on the general path
    finally:
        eng.close()


@pytest.mark.parametrize("p,bsz", [(12, 7), (100, 128), (100, 16), (127, 40), (130, 24)])
def test_accumulate_and_merge_at_once(p, bsz):
    """lsspa_lift_collect / lsspa_lift_batch with accumulate = 2 (one GPU: batch moments and Chan merge in one launch
    for p <= 128, the usual launches beyond) leaves the running statistics where accumulate = 1 followed by
    lsspa_stats_merge leaves them, batch after batch; a pending batch is refused."""
    from ls_spa._engine import HipEngine
    Xa, Xe, ya, ye = problem(31, p, 4 * p + 50, 3 * p + 40)
    rng = np.random.default_rng(5)
    two, one = HipEngine(0), HipEngine(0)
    try:
        for eng in (two, one):
            eng.load_data(Xa, Xe, ya, ye, 1e-3)
        for _ in range(4):
            perms = np.array([rng.permutation(p) for _ in range(bsz)])
            two.run_batch(perms, True, accumulate=True)
            two.merge()
            one.run_batch(perms, True, accumulate=2)
            n2, m2, c2 = two.stats()
            n1, m1, c1 = one.stats()
            assert n1 == n2
            np.testing.assert_allclose(m1, m2, rtol=0, atol=1e-15)
            # the two forms add the samples' outer products up in different orders, and the first batch's are taken
            # about a zero mean: rounding of the size of the squared lifts (<= 1), not of the covariance
            np.testing.assert_allclose(c1, c2, rtol=1e-13, atol=1e-15)
        # launch / collect in parts, the second part at once
        perms = np.array([rng.permutation(p) for _ in range(2 * bsz)])
        t2, t1 = two.launch_batch(perms, True), one.launch_batch(perms, True)
        for first in (0, bsz):
            two.collect_batch(t2, accumulate=True, first=first, count=bsz)
            two.merge()
            one.collect_batch(t1, accumulate=2, first=first, count=bsz)
        np.testing.assert_allclose(one.stats()[2], two.stats()[2], rtol=1e-13, atol=1e-15)
        assert one.stats()[0] == two.stats()[0] == 6 * bsz
        # against numpy on the lift vectors themselves
        one.reset_stats()
        lifts = one.run_batch(perms, True, want_lifts=True, accumulate=2)
        n, mean, cov = one.stats()
        assert n == 2 * bsz
        np.testing.assert_allclose(mean, lifts.mean(0), rtol=0, atol=1e-14)
        np.testing.assert_allclose(cov, np.cov(lifts, rowvar=False, bias=True), rtol=0, atol=1e-14)
        np.testing.assert_array_equal(cov, cov.T)       # exactly symmetric on the fused path too
        one.run_batch(perms[:3], True, accumulate=True)
        with pytest.raises(Exception, match="pending"):
            one.run_batch(perms[:3], True, accumulate=2)
        with pytest.raises(Exception, match="accumulate"):
            one.run_batch(perms[:3], True, accumulate=3)
        # the refused calls launched nothing and left no lane behind: the engine goes on as if they had not been made
        one.merge()
        more = one.run_batch(perms[3:8], True, want_lifts=True, accumulate=2)
        n, mean, cov = one.stats()
        assert n == 2 * bsz + 3 + 5
        allv = np.concatenate([lifts, lifts[:3], more])
        np.testing.assert_allclose(mean, allv.mean(0), rtol=0, atol=1e-14)
        np.testing.assert_allclose(cov, np.cov(allv, rowvar=False, bias=True), rtol=0, atol=1e-14)
    finally:
        two.close()
        one.close()


@pytest.mark.parametrize("p,chunk,n_chunks", [(7, 5, 3), (100, 16, 8), (100, 256, 8), (128, 33, 32), (61, 512, 2),
                                               (130, 16, 4), (100, 16, 33)])
def test_chunks_of_a_batch_folded_in_one_call(p, chunk, n_chunks):
    """lsspa_lift_collect_chunks: the parts of a launched batch folded one after the other by one call -- at p <= 128
    one statistics launch (stats_small_multi_kernel) that carries the running mean and n from part to part in
    registers -- leaves n, the mean and the covariance where part-by-part lsspa_lift_collect(accumulate = 2) leaves
    them, to the LAST BIT, with statistics already there and with the parts starting inside the batch; beyond the
    fused form's limits (p = 130, 33 parts) the call is the loop itself."""
    from ls_spa._engine import HipEngine
    Xa, Xe, ya, ye = problem(77, p, 3 * p + 60, 2 * p + 50)
    rng = np.random.default_rng(p + chunk)
    parts, whole = HipEngine(0), HipEngine(0)
    try:
        for eng in (parts, whole):
            eng.load_data(Xa, Xe, ya, ye, 1e-3)
        for rnd in range(2):                      # the second round starts from the first one's statistics
            lead = 0 if rnd == 0 else chunk       # ... and inside the batch: one part taken by itself first
            perms = np.array([rng.permutation(p) for _ in range(lead + chunk * n_chunks)])
            tp, tw = parts.launch_batch(perms, True), whole.launch_batch(perms, True)
            if lead:
                parts.collect_batch(tp, accumulate=2, first=0, count=lead)
                whole.collect_batch(tw, accumulate=2, first=0, count=lead)
            for c in range(n_chunks):
                parts.collect_batch(tp, accumulate=2, first=lead + c * chunk, count=chunk)
            whole.collect_chunks(tw, lead, chunk, n_chunks, accumulate=2)
            n1, m1, c1 = parts.stats()
            n2, m2, c2 = whole.stats()
            assert n1 == n2 == (rnd + 1) * chunk * n_chunks + lead * rnd
            np.testing.assert_array_equal(m2, m1)
            np.testing.assert_array_equal(c2, c1)
        # both lanes were given back: the engines go on
        perms = np.array([rng.permutation(p) for _ in range(4)])
        np.testing.assert_array_equal(whole.run_batch(perms, True, want_lifts=True, accumulate=2),
                                      parts.run_batch(perms, True, want_lifts=True, accumulate=2))
        np.testing.assert_array_equal(whole.stats()[2], parts.stats()[2])
    finally:
        parts.close()
        whole.close()


@pytest.mark.parametrize("p,bs,anti", [(24, 16, True), (100, 16, True), (126, 10, True), (12, 3, True), (13, 5, True),
                                       (61, 40, True), (40, 16, False), (127, 7, False)])
def test_a_group_of_checks_from_one_statistics_launch(p, bs, anti):
    """The public call on a small problem with the device estimator: a look-ahead group's chunks are folded by ONE
    statistics launch inside lsspa_group_collect and every chunk's check reads the mean and n as they stood after ITS
    chunk -- the error history, the attribution and its errors are those of the run that folds, merges and checks
    chunk by chunk (lookahead = 1, nothing deferred), to the last bit; also when the stop rule fires inside a group."""
    from ls_spa import ls_spa
    Xa, Xe, ya, ye = problem(5, p, 4 * p + 30, 3 * p + 20)
    # (chunks of 10, 3, 5 and 40 samples: the normals' blocks are padded to 16 columns; 21 chunks in groups of 8, 8, 5)
    kw = dict(reg=1e-3, method="argsort", seed=3, batch_size=bs, max_samples=bs * 21, error_estimator="device",
              antithetical=anti)
    ref = ls_spa(Xa, Xe, ya, ye, tolerance=0.0, lookahead=1, _defer=0, **kw)
    grp = ls_spa(Xa, Xe, ya, ye, tolerance=0.0, lookahead=8, **kw)
    assert len(ref.error_history) == len(grp.error_history) == 22          # 21 chunks and the check at max - 1
    auto = ls_spa(Xa, Xe, ya, ye, tolerance=0.0, **kw)                       # the default: groups of up to 16
    np.testing.assert_array_equal(auto.error_history, ref.error_history)
    np.testing.assert_array_equal(auto.attribution, ref.attribution)
    np.testing.assert_array_equal(grp.error_history, ref.error_history)
    np.testing.assert_array_equal(grp.attribution, ref.attribution)
    np.testing.assert_array_equal(grp.attribution_errors, ref.attribution_errors)
    tol = float(ref.error_history[10]) * 1.0000001
    if all(e > tol for e in ref.error_history[:10]):
        a = ls_spa(Xa, Xe, ya, ye, tolerance=tol, lookahead=1, _defer=0, **kw)
        b = ls_spa(Xa, Xe, ya, ye, tolerance=tol, lookahead=8, **kw)
        assert len(a.error_history) == len(b.error_history) <= 12
        np.testing.assert_array_equal(b.error_history, a.error_history)
        np.testing.assert_array_equal(b.attribution, a.attribution)


@pytest.mark.parametrize("p", [12, 100, 120, 300])
def test_every_batch_checks_its_sums(p):
    """Every ordering's lifts telescope to the full model's R^2 (ls_spa/ls_spa.py:284-285), and once that is known
    (lsspa_full_fit) every batch is checked against it: inside the register-resident small-problem kernel per ordering
    (p = 12, 100), by a launch of its own per sample otherwise (p = 120: the LDS-resident kernel; p = 300: the general
    path).  A healthy engine stays silent with a deviation at round-off; with the R^2 moved by 1e-3 (test hook) every
    form raises LSSPA_INFO_SUM and reports that deviation."""
    from ls_spa._engine import HipEngine
    Xa, Xe, ya, ye = problem(9, p, 3 * p + 100, 2 * p + 80)
    rng = np.random.default_rng(p)
    eng = HipEngine(0)
    try:
        eng.load_data(Xa, Xe, ya, ye, 1e-3)
        eng.full_fit()
        perms = np.array([rng.permutation(p) for _ in range(24)])
        for anti in (True, False):
            eng.reset_stats()
            lifts = eng.run_batch(perms, anti, want_lifts=True, accumulate=2)
            r2 = lifts.sum(1).mean()
            assert eng.info() == 0 and eng.sum_deviation() < 1e-11
        eng._check(eng._lib.lsspa_debug_set_r2(eng._h, float(r2) + 1e-3))
        for anti in (True, False):
            eng.reset_stats()
            eng.run_batch(perms, anti, want_lifts=False, accumulate=2)
            assert eng.info() & 8
            assert eng.sum_deviation() == pytest.approx(1e-3, rel=1e-6)
        eng._check(eng._lib.lsspa_debug_set_r2(eng._h, float(r2)))
        eng.reset_stats()
        eng.run_batch(perms, True, want_lifts=False, accumulate=2)
        assert eng.info() == 0
    finally:
        eng.close()


def test_collect_chunks_refuses_what_collect_refuses():
    """lsspa_lift_collect_chunks is lsspa_lift_collect part after part: parts out of turn, beyond the batch or of no
    samples are refused with the lane left as it was (the batch can still be collected), and the R^2 test hook needs a
    full fit first."""
    from ls_spa._engine import HipEngine
    Xa, Xe, ya, ye = problem(3, 30, 200, 150)
    rng = np.random.default_rng(0)
    eng = HipEngine(0)
    try:
        eng.load_data(Xa, Xe, ya, ye, 1e-3)
        with pytest.raises(Exception, match="full_fit"):
            eng._check(eng._lib.lsspa_debug_set_r2(eng._h, 0.5))
        perms = np.array([rng.permutation(30) for _ in range(40)])
        tk = eng.launch_batch(perms, True)
        with pytest.raises(Exception):
            eng.collect_chunks(tk, 8, 8, 2)            # not from the front
        with pytest.raises(Exception):
            eng.collect_chunks(tk, 0, 16, 3)           # 48 > 40 samples
        with pytest.raises(Exception):
            eng.collect_chunks(tk, 0, 0, 2)            # empty parts
        with pytest.raises(Exception):
            eng.collect_chunks(tk, 0, 8, 2, accumulate=3)
        eng.collect_chunks(tk, 0, 8, 5)                # the whole batch, five parts in one launch
        ref = HipEngine(0)
        try:
            ref.load_data(Xa, Xe, ya, ye, 1e-3)
            ref.run_batch(perms[:8], True, accumulate=2)
            for k in range(1, 5):
                ref.run_batch(perms[8 * k:8 * k + 8], True, accumulate=2)
            np.testing.assert_array_equal(eng.stats()[2], ref.stats()[2])
            np.testing.assert_array_equal(eng.stats()[1], ref.stats()[1])
        finally:
            ref.close()
    finally:
        eng.close()


def test_grouped_collect_survives_a_refused_allocation():
    """The grouped form of lsspa_group_collect makes its buffers before its first launch: with the next device allocation
    refused (test hook) the call fails, the lane still holds its batch and statistics and estimator have not moved --
    the same call again goes through and gives what chunk-by-chunk calls give."""
    from ls_spa._engine import HipEngine
    Xa, Xe, ya, ye = problem(4, 30, 220, 160)
    rng = np.random.default_rng(1)
    perms = np.array([rng.permutation(30) for _ in range(48)])
    args = ([0, 16, 32], [16, 16, 16], [0, 16, 32], 1, [16, 32, 48], [0, 1, 2])
    eng, ref = HipEngine(0), HipEngine(0)
    try:
        for e in (eng, ref):
            e.load_data(Xa, Xe, ya, ye, 1e-3)
            e.full_fit()
            e.error_running_enable(7)
        tk = eng.launch_batch(perms, True)
        eng.debug_fail_alloc(1)
        with pytest.raises(MemoryError):
            eng.group_collect(tk, *args)
        eng.debug_fail_alloc(0)
        assert eng.stats(want_cov=False)[0] == 0
        eng.group_collect(tk, *args)
        tr = ref.launch_batch(perms, True)
        for c in range(3):
            ref.collect_batch(tr, accumulate=2, first=16 * c, count=16)
            ref.error_advance(16 * c, 1)
            ref.error_check_enqueue(16 * (c + 1), c)
        for c in range(3):
            a, b = eng.error_result(c, wait=True), ref.error_result(c, wait=True)
            for x, y in zip(a, b):
                np.testing.assert_array_equal(np.asarray(x), np.asarray(y))
        np.testing.assert_array_equal(eng.stats()[2], ref.stats()[2])
    finally:
        eng.close()
        ref.close()
