"""Parity tests proper (-m gpu): the product path -- ls_spa.ls_spa() -> ctypes -> C ABI -> HIP kernels --
against (i) golden fixtures produced by the real reference, (ii) the CPU oracle on the same seeded
inputs, (iii) the reference's own unit tests re-expressed, (iv) size-independent properties at the
BASELINE sizes.

Stated fp64 tolerance: |lift_gpu - lift_ref| <= 1e-10 per ordering on benchmark-conditioned data
(observed ~1e-15); theta to 1e-9 relative; the error-estimator outputs are statistical pins (the
sample covariance is singular by construction, see test_oracle_golden.py).
"""
import numpy as np
import scipy.linalg as sla
import pytest

import lsspa_oracle as O
from ls_spa import (ShapleyResults, ls_spa, reduce_data, square_shapley)

pytestmark = pytest.mark.gpu
LIFT_TOL = dict(rtol=0, atol=1e-10)


def data_of(g):
    return [g[k] for k in ("X_train", "X_test", "y_train", "y_test")]


# ------------------------------------------------------------------ golden fixtures (reference outputs)
def test_toy_exact(golden):
    g = golden("toy")
    res = ls_spa(*data_of(g))
    np.testing.assert_allclose(res.attribution, g["attribution"], **LIFT_TOL)
    np.testing.assert_allclose(res.theta, g["theta"], rtol=1e-10)
    assert abs(res.r_squared - float(g["r_squared"])) < 1e-12
    assert res.overall_error == 0.0 and res.error_history.size == 0 and res.attribution_history is None
    assert repr(res) == str(g["repr"])


@pytest.mark.parametrize("p", [4, 8])
def test_exact_small_p(golden, p):
    g = golden(f"exact_p{p}")
    res = ls_spa(*data_of(g))
    np.testing.assert_allclose(res.attribution, g["attribution"], **LIFT_TOL)
    np.testing.assert_allclose(res.theta, g["theta"], rtol=1e-10)
    res2 = ls_spa(*data_of(g), method="exact")
    np.testing.assert_allclose(res2.attribution, g["attribution"], **LIFT_TOL)


@pytest.mark.parametrize("tag,reg", [("r0", 0.0), ("r1", 0.1)])
def test_helpers_reduce_and_square_shapley(golden, tag, reg):
    g = golden("p12")
    d = data_of(g)
    R, F, q, qt = reduce_data(*d, reg)
    Rg, Fg = g[f"{tag}_R_tr"], g[f"{tag}_F_te"]
    np.testing.assert_allclose(R.T @ R, Rg.T @ Rg, rtol=0, atol=1e-12)
    np.testing.assert_allclose(np.abs(R), np.abs(Rg), rtol=0, atol=1e-11)     # equal up to row signs
    np.testing.assert_allclose(R.T @ q, Rg.T @ g[f"{tag}_q_tr"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(F.T @ F, Fg.T @ Fg, rtol=0, atol=1e-10)
    np.testing.assert_allclose(F.T @ qt, Fg.T @ g[f"{tag}_q_te"], rtol=0, atol=1e-10)
    yy = np.linalg.norm(d[3]) ** 2
    for o, want in zip(g["orders"], g[f"{tag}_lifts"]):
        # the helper takes ANY valid factors: ours and the reference's (LAPACK signs) alike
        np.testing.assert_allclose(square_shapley(R, F, q, qt, yy, o), want, **LIFT_TOL)
        np.testing.assert_allclose(square_shapley(Rg, Fg, g[f"{tag}_q_tr"], g[f"{tag}_q_te"], yy, o), want,
                                   **LIFT_TOL)


@pytest.mark.parametrize("anti", [True, False])
def test_driver_injected_perms(golden, anti):
    g = golden("p12")
    pre = f"drv_anti{int(anti)}_"
    res = ls_spa(*data_of(g), perms=g["perms64"], batch_size=16, tolerance=0.0, antithetical=anti,
                 return_attribution_history=True)
    np.testing.assert_allclose(res.attribution, g[pre + "attribution"], **LIFT_TOL)
    np.testing.assert_allclose(res.attribution_history, g[pre + "attribution_history"], **LIFT_TOL)
    np.testing.assert_allclose(res.theta, g[pre + "theta"], rtol=1e-10)
    assert abs(res.r_squared - float(g[pre + "r_squared"])) < 1e-12
    assert len(res.error_history) == 4
    np.testing.assert_allclose(res.error_history, g[pre + "error_history"], rtol=0.15)
    np.testing.assert_allclose(res.attribution_errors, g[pre + "attribution_errors"], rtol=0.25)


def test_seed_path_first_batch(golden):
    g = golden("p12")
    res = ls_spa(*data_of(g), max_samples=40, batch_size=16, tolerance=0.0, seed=3,
                 return_attribution_history=True)
    np.testing.assert_allclose(res.attribution_history[:16], g["seedpath_attribution_history"][:16], **LIFT_TOL)
    assert len(res.error_history) == 4 and res.attribution_history.shape == (40, 12)


def test_correlated_generator_ill_conditioned(golden, engine):
    """The reference's own hard generator (cond ~ 1e3): pins the Gram/Cholesky route's accuracy."""
    g = golden("corr_p100")
    R, F, q, qt = g["R_tr"], g["F_te"], g["q_tr"], g["q_te"]
    yy = float(g["y_norm_sq"])
    # both device paths: rect (any factor) and tri (test Gram)
    engine.load_reduced(R.T @ R, R.T @ q, float(q @ q), yy, Ft=F.T.copy(), ytil=qt)
    rect = engine.run_batch(g["orders"], False, want_lifts=True, accumulate=False)
    engine.load_reduced(R.T @ R, R.T @ q, float(q @ q), yy, H=F.T @ F, h=F.T @ qt)
    tri = engine.run_batch(g["orders"], False, want_lifts=True, accumulate=False)
    np.testing.assert_allclose(rect, g["lifts"], **LIFT_TOL)
    np.testing.assert_allclose(tri, g["lifts"], **LIFT_TOL)
    assert engine.info() == 0


def test_m_less_than_p_and_float32(golden):
    g = golden("edge")
    res = ls_spa(*data_of(g), perms=g["perms"], batch_size=8, tolerance=0.0)
    np.testing.assert_allclose(res.attribution, g["mltp_attribution"], **LIFT_TOL)
    np.testing.assert_allclose(res.theta, g["mltp_theta"], rtol=1e-10)
    d32 = [a.astype(np.float32) for a in data_of(golden("p12"))]
    res = ls_spa(*d32, perms=g["perms"], batch_size=8, tolerance=0.0)
    assert res.attribution.dtype == np.float64
    np.testing.assert_allclose(res.attribution, g["f32_attribution"], rtol=0, atol=2e-5)


# ------------------------------------------------------------------ reference unit tests, re-expressed
@pytest.fixture(scope="module")
def ref_data():
    """setUp of the reference's TestLSSPA (test/test_ls_spa.py:48-72)."""
    rng = np.random.default_rng(128)
    n = 100
    A = rng.standard_normal((n, n))
    Qm, _ = np.linalg.qr(A)
    easy_X = Qm @ np.sqrt(np.diag(np.arange(1, n + 1)))
    easy_y = Qm[:, 0]
    w = rng.standard_normal(n)
    Xa = rng.multivariate_normal(np.zeros(n), A @ A.T, n)
    Xa_c = Xa - Xa.mean(0, keepdims=True)
    Xe = rng.multivariate_normal(np.zeros(n), A @ A.T, n)
    Xe_c = Xe - Xa.mean(0, keepdims=True)
    ya = Xa_c @ w + rng.standard_normal(n)
    ya_c = ya - ya.mean()
    ye = Xe_c @ w + rng.standard_normal(n)
    ye_c = ye - ye.mean()
    return dict(easy=(easy_X, easy_X.copy(), easy_y, easy_y.copy()), hard=(Xa_c, Xe_c, ya_c, ye_c))


def test_ref_return_type(ref_data):                                   # test_ls_spa.py:75-79
    assert isinstance(ls_spa(*ref_data["easy"]), ShapleyResults)


def test_ref_linear_regression(ref_data):                            # :82-96
    Xa, Xe, ya, ye = ref_data["easy"]
    th = np.linalg.lstsq(Xa, ya, rcond=None)[0]
    np.testing.assert_almost_equal(th, ls_spa(Xa, Xe, ya, ye, max_samples=4, batch_size=2).theta)
    Xa, Xe, ya, ye = ref_data["hard"]                                 # N = p, centred: rank p - 1
    th = np.linalg.lstsq(Xa, ya, rcond=None)[0]
    with pytest.warns(RuntimeWarning):
        got = ls_spa(Xa, Xe, ya, ye, max_samples=4, batch_size=2).theta
    np.testing.assert_almost_equal(th, got, decimal=6)


def test_ref_rsquared(ref_data):                                      # :99-109
    Xa, Xe, ya, ye = ref_data["hard"]
    th = np.linalg.lstsq(Xa, ya, rcond=None)[0]
    r2 = 1 - np.sum((ye - Xe @ th) ** 2) / np.sum(ye ** 2)
    with pytest.warns(RuntimeWarning):
        got = ls_spa(Xa, Xe, ya, ye, max_samples=4, batch_size=2).r_squared
    np.testing.assert_almost_equal(r2, got, decimal=6)


def test_ref_regularization(ref_data):                                # :112-124
    Xa, Xe, ya, ye = ref_data["hard"]
    N, p = Xa.shape
    Xr = np.vstack((Xa / np.sqrt(N), np.sqrt(0.1) * np.eye(p)))
    yr = np.concatenate((ya / np.sqrt(N), np.zeros(p)))
    th = np.linalg.lstsq(Xr, yr, rcond=None)[0]
    np.testing.assert_almost_equal(th, ls_spa(Xa, Xe, ya, ye, reg=0.1, max_samples=4, batch_size=2).theta)


def test_ref_seed_consistency(ref_data):                              # :127-135
    Xa, Xe, ya, ye = ref_data["hard"]
    a = ls_spa(Xa, Xe, ya, ye, reg=0.05, seed=42, max_samples=4, batch_size=2).attribution
    b = ls_spa(Xa, Xe, ya, ye, reg=0.05, seed=42, max_samples=4, batch_size=2).attribution
    np.testing.assert_array_equal(a, b)


def test_ref_correctness_easy(ref_data):                              # :138-160
    Xa, Xe, ya, ye = ref_data["easy"]
    p = Xa.shape[1]
    want = O.refit_lift(Xa, Xe, ya, ye, np.arange(p))    # orthogonal columns: any ordering gives the same lifts
    res = ls_spa(Xa, Xe, ya, ye, max_samples=256 * 256, batch_size=256)
    np.testing.assert_almost_equal(want, res.attribution)
    assert len(res.error_history) == 1                 # covariance ~ 0: stops at the first check, i = 256


# ------------------------------------------------------------------ seeded parity vs the oracle, mid sizes
@pytest.mark.parametrize("p,n,m,reg", [(33, 200, 150, 0.0), (128, 600, 500, 1e-2), (257, 700, 300, 0.0),
                                       (191, 500, 100, 0.0)])
def test_lifts_vs_oracle_various_shapes(engine, p, n, m, reg):
    Xa, Xe, ya, ye = O.gaussian_workload(p, n, m, seed=p)
    engine.load_data(Xa, Xe, ya, ye, reg)
    red = O.reduce(Xa, Xe, ya, ye, reg)
    yy = float(ye @ ye)
    rng = np.random.default_rng(p)
    perms = np.array([np.arange(p), np.arange(p)[::-1]] + [rng.permutation(p) for _ in range(6)])
    for anti in (False, True):
        got = engine.run_batch(perms, anti, want_lifts=True, accumulate=False)
        want = np.array([O.sample_lift(*red, yy, o, anti) for o in perms])
        np.testing.assert_allclose(got, want, **LIFT_TOL)


def test_bad_perms_rejected(engine):
    Xa, Xe, ya, ye = O.gaussian_workload(12, 60, 50, seed=1)
    engine.load_data(Xa, Xe, ya, ye, 0.0)
    bad = np.arange(12)[None, :].copy()
    bad[0, 3] = 4
    with pytest.raises(ValueError):
        engine.run_batch(bad, True)
    with pytest.raises(ValueError):
        engine.run_batch(np.arange(11)[None, :], True)


# ------------------------------------------------------------------ BASELINE sizes: properties
@pytest.mark.parametrize("p,N", [(100, 10000), (1000, 100000)])
def test_full_size_properties(p, N):
    """C2 / C3 shapes, data generated on the device.  Every ordering's lift vector sums to the
    full-model R^2; reversing twice is the identity; antithetical = mean of the
    two directions; a sampled sub-problem agrees with the oracle."""
    import torch
    from ls_spa._engine import HipEngine
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    Xa = torch.randn(N, p, dtype=torch.float64, device=dev)
    Xe = torch.randn(N, p, dtype=torch.float64, device=dev)
    w = torch.randn(p, dtype=torch.float64, device=dev)
    ya = Xa @ w + torch.randn(N, dtype=torch.float64, device=dev)
    ye = Xe @ w + torch.randn(N, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    eng = HipEngine(0)
    try:
        eng.load_device_data(Xa.data_ptr(), p, ya.data_ptr(), N, Xe.data_ptr(), p, ye.data_ptr(), N, p, 0.0)
        G, g, H, h = eng.gram()
        np.testing.assert_allclose(G, (Xa.T @ Xa / N).cpu().numpy(), rtol=0, atol=1e-12)
        np.testing.assert_allclose(h, (Xe.T @ ye).cpu().numpy(), rtol=1e-12)
        theta, r2, info = eng.full_fit()
        assert info == 0
        th = torch.linalg.solve(Xa.T @ Xa, Xa.T @ ya)
        np.testing.assert_allclose(theta, th.cpu().numpy(), rtol=1e-8, atol=1e-10)
        r2_ref = 1 - float(((ye - Xe @ th) ** 2).sum() / (ye ** 2).sum())
        assert abs(r2 - r2_ref) < 1e-10
        from scipy.stats.qmc import Sobol
        perms = np.argsort(Sobol(p, seed=42).random(16), axis=1)
        fwd = eng.run_batch(perms, False, want_lifts=True, accumulate=False)
        np.testing.assert_allclose(fwd.sum(1), r2, rtol=0, atol=1e-10)
        rev = eng.run_batch(perms[:, ::-1].copy(), False, want_lifts=True, accumulate=False)
        anti = eng.run_batch(perms, True, want_lifts=True, accumulate=False)
        np.testing.assert_allclose(anti, 0.5 * (fwd + rev), rtol=0, atol=1e-13)
        # oracle on the reduced problem (Cholesky factors of the device Grams are valid factors)
        R = np.linalg.cholesky(G).T
        F = np.linalg.cholesky(H).T
        q = np.linalg.solve(R.T, g)
        qt = np.linalg.solve(F.T, h)
        # every one of the 16 orderings, and the antithetical pairs, against the oracle's QR-based algorithm
        want = np.array([O.ordering_lift(R, F, q, qt, eng.y_norm_sq, o) for o in perms])
        np.testing.assert_allclose(fwd, want, **LIFT_TOL)
        want_rev = np.array([O.ordering_lift(R, F, q, qt, eng.y_norm_sq, o[::-1]) for o in perms[:4]])
        np.testing.assert_allclose(anti[:4], 0.5 * (want[:4] + want_rev), **LIFT_TOL)
    finally:
        eng.close()


def test_c5_shape_fp32_vs_fp64():
    """BASELINE config 5's shape (p = 5000, reg = 1e-2, argsort, fp32) at reduced N: the fp32 path
    against the fp64 path (itself parity-checked above) on the same orderings, plus the sum-to-R^2
    invariant.  Stated fp32 tolerance: 1e-4 absolute per lift."""
    import torch
    from scipy.stats.qmc import Sobol
    from ls_spa._engine import HipEngine
    p, N = 5000, 12000
    torch.manual_seed(1)
    dev = torch.device("cuda:0")
    Xa = torch.randn(N, p, dtype=torch.float32, device=dev)
    Xe = torch.randn(N, p, dtype=torch.float32, device=dev)
    w = torch.randn(p, dtype=torch.float32, device=dev) / 70.0
    ya = Xa @ w + torch.randn(N, dtype=torch.float32, device=dev)
    ye = Xe @ w + torch.randn(N, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    perms = np.argsort(Sobol(p, seed=42).random(4), axis=1)
    eng = HipEngine(0)
    try:
        out = {}
        for prec in ("float64", "float32"):
            eng.set_precision(prec)
            eng.load_device_data(Xa.data_ptr(), p, ya.data_ptr(), N, Xe.data_ptr(), p, ye.data_ptr(), N, p, 1e-2,
                                 f32=True)
            theta, r2, info = eng.full_fit()
            assert info == 0
            out[prec] = (eng.run_batch(perms, True, want_lifts=True, accumulate=False), theta, r2)
        l64, th64, r64 = out["float64"]
        l32, th32, r32 = out["float32"]
        np.testing.assert_allclose(l64.sum(1), r64, rtol=0, atol=1e-9)
        np.testing.assert_allclose(l32, l64, rtol=0, atol=1e-4)
        assert abs(r32 - r64) < 1e-4
        np.testing.assert_allclose(th32, th64, rtol=0, atol=1e-4)
    finally:
        eng.close()


def test_device_error_estimator_end_to_end():
    """ls_spa(error_estimator='device') on the product path: the running form (D = Xi L, s = Xi 1 in HBM, normals made
    on the device) against the same arithmetic in NumPy on the run's own lift vectors, and -- other normals, same
    distribution (ls_spa/ls_spa.py:321-341) -- statistically against 'lowrank' (host)."""
    import philox_ref as P
    d = O.correlated_workload(np.random.default_rng(21), 100, 2000, 1500)
    # (a QMC method: with the shared generator of method=None the orderings after the first check depend on how many
    # normals the estimator drew from it -- 'lowrank' draws, 'device' does not touch it)
    kw = dict(method="argsort", max_samples=512, batch_size=64, tolerance=0.0, seed=7)
    low = ls_spa(*d, error_estimator="lowrank", return_attribution_history=True, **kw)
    dev = ls_spa(*d, error_estimator="device", **kw)
    assert len(dev.error_history) == len(low.error_history) == 9       # 64 .. 512, and 511
    np.testing.assert_allclose(dev.attribution, low.attribution, rtol=0, atol=1e-14)
    np.testing.assert_allclose(dev.error_history, low.error_history, rtol=0.15)
    ratio = dev.attribution_errors / low.attribution_errors
    assert abs(np.median(ratio) - 1.0) < 0.1 and ratio.min() > 0.6 and ratio.max() < 1.6
    assert dev.overall_error == dev.error_history[-1]
    hist = low.attribution_history
    lifts = np.diff(np.vstack([np.zeros(100), hist * np.arange(1, 513)[:, None]]), axis=0)
    seed = int(np.random.SeedSequence(7).generate_state(1, np.uint64)[0])
    Xi = P.normals(seed, np.arange(512))
    for k, n in enumerate((64, 128, 192, 256, 320, 384, 448, 511, 512)):
        L = lifts[:n]
        x = Xi[:, :n] @ (L - L.mean(0)) / np.sqrt(n * (n - 1.0))
        assert dev.error_history[k] == pytest.approx(np.quantile(np.linalg.norm(x, axis=1), 0.95), rel=1e-7)
    # the stop rule on the deferred path: the numbers of the stopping check, as when every check is waited for
    tol = float(dev.error_history[3]) * 1.0000001
    kq = dict(method="argsort", max_samples=512, batch_size=64, seed=7)
    full = ls_spa(*d, tolerance=0.0, **kq)
    tol = float(full.error_history[3]) * 1.0000001
    if all(e > tol for e in full.error_history[:3]):
        a = ls_spa(*d, tolerance=tol, _defer=0, **kq)
        b = ls_spa(*d, tolerance=tol, **kq)
        assert len(a.error_history) == len(b.error_history) == 4
        np.testing.assert_array_equal(b.attribution, a.attribution)
        np.testing.assert_array_equal(b.error_history, a.error_history)
        np.testing.assert_array_equal(b.attribution_errors, a.attribution_errors)


@pytest.mark.parametrize("estimator", ["reference", "device"])
def test_checkpoint_resume_on_the_product_path(tmp_path, estimator):
    """Interrupt the HIP-backed run in its third batch and resume from the state file: the result is the
    uninterrupted run's (lsspa_stats_set / lsspa_error_state_set restore the device state)."""
    from ls_spa._engine import HipEngine

    class Dies(HipEngine):
        calls = 0

        def run_batch(self, *a, **k):
            self.calls += 1
            if self.calls == 3:
                raise KeyboardInterrupt
            return super().run_batch(*a, **k)

    d = O.gaussian_workload(30, 400, 300, seed=4)
    kw = dict(max_samples=160, batch_size=32, tolerance=0.0, seed=3, method="argsort", error_estimator=estimator,
              lookahead=1)
    full = ls_spa(*d, **kw)
    ck = str(tmp_path / "run.npz")
    eng = Dies(0)
    with pytest.raises(KeyboardInterrupt):
        ls_spa(*d, checkpoint=ck, _engine=eng, **kw)
    eng.close()
    with np.load(ck) as z:
        assert int(z["n"]) == 64
    res = ls_spa(*d, checkpoint=ck, **kw)
    np.testing.assert_allclose(res.attribution, full.attribution, rtol=0, atol=1e-14)
    np.testing.assert_allclose(res.error_history, full.error_history, rtol=1e-8)
    assert len(res.error_history) == len(full.error_history) == 6   # 32, 64, 96, 128, 159, 160


def test_integration_md_stub_runs(golden):
    """The ctypes stub printed in INTEGRATION.md is executed as written (only the library path is resolved to
    the in-tree build) and must reproduce the reference's toy attribution."""
    import os
    import re
    from ls_spa import _native
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n# ls_spa/_hip\.py.*?\n(.*?)```", text, re.S).group(1)
    _native.load()   # resolves the HIP runtime the way the package does
    code = code.replace('C.CDLL("liblsspa_hip.so")', f'C.CDLL({_native._LIB_PATH!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    g = golden("toy")
    d = data_of(g)
    eng = ns["Engine"](0)
    eng.reduce(*d, 0.0)
    import itertools
    perms = np.array(list(itertools.permutations(range(3))))
    lifts = eng.lifts(perms, False)
    np.testing.assert_allclose(lifts.mean(0), g["attribution"], rtol=0, atol=1e-12)
    n, mu, cov = eng.stats()
    assert n == 6
    np.testing.assert_allclose(mu, g["attribution"], rtol=0, atol=1e-12)
    th, r2 = eng.full_fit()
    np.testing.assert_allclose(th, g["theta"], rtol=1e-10)
    assert abs(r2 - float(g["r_squared"])) < 1e-12
    ns["_lib"].lsspa_destroy(eng.h)


def test_c4_end_to_end_from_host_arrays():
    """BASELINE config 4 end to end through the public call: 2 x 800 MB of host data (SURVEY.md 8d generator,
    p = 1000, N = M = 1e5) streamed into the reduction, permutohedron QMC orderings, antithetical pairs,
    two batches of 128.  Size-independent checks: the attribution sums to the out-of-sample R^2, theta
    solves the normal equations, both batches were checked, and the run is reproducible."""
    p, n = 1000, 100000
    rng = np.random.default_rng(0)
    Xa = rng.standard_normal((n, p))
    Xe = rng.standard_normal((n, p))
    w = rng.standard_normal(p)
    ya = Xa @ w + rng.standard_normal(n)
    ye = Xe @ w + rng.standard_normal(n)
    kw = dict(method="permutohedron", batch_size=128, num_batches=2, tolerance=0.0, seed=42,
              error_estimator="device")
    res = ls_spa(Xa, Xe, ya, ye, **kw)
    assert res.attribution.shape == (p,) and np.all(np.isfinite(res.attribution))
    assert abs(res.attribution.sum() - res.r_squared) < 1e-9
    G = Xa.T @ Xa
    np.testing.assert_allclose(G @ res.theta, Xa.T @ ya, rtol=1e-8, atol=1e-6)
    r2 = 1.0 - np.sum((ye - Xe @ res.theta) ** 2) / (ye @ ye)
    assert abs(res.r_squared - r2) < 1e-10
    assert len(res.error_history) == 3 and res.overall_error == res.error_history[-1]   # 128, 255, 256
    assert 0 < res.overall_error < 1e-3
    again = ls_spa(Xa, Xe, ya, ye, **kw)
    np.testing.assert_array_equal(again.attribution, res.attribution)


def test_c4_sampler_and_hip_path_against_the_oracle():
    """BASELINE config 4's sampler and the HIP path TOGETHER against the oracle (the full-size test above checks
    properties only): p = 1000, the row count cut to 3000 so that the oracle's per-ordering QR finishes in seconds,
    method='permutohedron' (MultivariateNormalQMC(zeros(p - 1), seed = 42, inv_transform = False), the basis of
    experiments/ground_truth_medium.py:56-67), antithetical, one batch of 16 -- the orderings the driver drew are
    the oracle's own sampler's, the attribution is the oracle's to 1e-10, the estimator's generator stream the same."""
    from scipy.stats.qmc import MultivariateNormalQMC
    p, n = 1000, 3000
    d = O.gaussian_workload(p, n, n, seed=4)
    res = ls_spa(*d, method="permutohedron", batch_size=16, num_batches=1, tolerance=0.0, seed=42,
                 return_attribution_history=True)
    orders = O.orderings_permutohedron(MultivariateNormalQMC(np.zeros(p - 1), seed=42, inv_transform=False), 16, p)
    want = O.estimate(*d, perms=orders, batch_size=16, tolerance=0.0, seed=42, return_attribution_history=True)
    np.testing.assert_allclose(res.attribution, want.attribution, rtol=0, atol=1e-10)
    np.testing.assert_allclose(res.attribution_history, want.attribution_history, rtol=0, atol=1e-10)
    np.testing.assert_allclose(res.theta, want.theta, rtol=1e-9, atol=1e-12)
    assert abs(res.r_squared - want.r_squared) < 1e-10
    # checks at i = max_samples - 1 = 15 and at 16 (the reference's trigger rule, ls_spa/ls_spa.py:222); the oracle, fed the
    # orderings through perms=, has no sample cap and checks at 16 only
    assert len(res.error_history) == 2 and len(want.error_history) == 1
    np.testing.assert_allclose(res.error_history[-1], want.error_history[-1], rtol=0.25)   # singular covariance: statistical pin


# ------------------------------------------------------------------ (f3) the experiment harness
def test_medium_experiment_harness(tmp_path):
    """experiments/medium_experiment.py at reduced size (rows 2000, ground truth from 2^10 samples, 512-sample
    convergence runs).  Reference: experiments/ground_truth_medium.py:108-119 (ground truth from injected random
    orderings drawn after the data from the SAME generator) and notebooks/medium_experiment.py:348-603 (six
    sampler runs, L2 error of the running attribution against the ground truth)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("medium_experiment",
                                                  os.path.join(root, "experiments", "medium_experiment.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lines = []
    gt, table, runs = mod.run(p=100, rows=2000, gt_log2=10, samples=512, out_dir=str(tmp_path), log=lines.append)
    # ground truth: every lift vector sums to the full model's R^2, so the mean does
    assert abs(gt.attribution.sum() - gt.r_squared) < 1e-10
    np.testing.assert_array_equal(np.load(tmp_path / "gt_Medium.npy"), gt.attribution)
    # the same experiment on the oracle: same generator stream -> same data, same injected orderings
    rng = np.random.default_rng(42)
    d = O.correlated_workload(rng, 100, 2000, 2000)
    perms = np.array([rng.permutation(100) for _ in range(2 ** 10)])
    want = O.estimate(*d, perms=perms, tolerance=0.0, batch_size=2 ** 12)
    np.testing.assert_allclose(gt.attribution, want.attribution, rtol=0, atol=1e-10)
    np.testing.assert_allclose(gt.theta, want.theta, rtol=1e-8, atol=1e-11)
    assert abs(gt.r_squared - want.r_squared) < 1e-10
    # six runs, full histories, the table and the CSV agree
    assert sorted(runs) == sorted((m, a) for m in ("random", "argsort", "permutohedron") for a in (False, True))
    for (method, anti), r in runs.items():
        assert r.attribution_history.shape == (512, 100)
        np.testing.assert_allclose(r.attribution_history[-1], r.attribution, rtol=0, atol=1e-13)
        err = np.linalg.norm(r.attribution_history - gt.attribution, axis=1)
        assert err[-1] < err[15]           # the estimate converges towards the ground truth
    assert len(table) == 6 * 6            # n = 16 .. 512 for each run
    csv = (tmp_path / "convergence.csv").read_text().strip().splitlines()
    assert csv[0] == "method,antithetical,samples,l2_error" and len(csv) == 1 + len(table)
    # one run against the oracle on its own sampler's orderings (QMC argsort, antithetical)
    from scipy.stats.qmc import Sobol
    orders = O.orderings_argsort(Sobol(100, seed=42), 512)
    ow = O.estimate(*d, perms=orders[:64], tolerance=0.0, batch_size=2 ** 8, return_attribution_history=True)
    np.testing.assert_allclose(runs[("argsort", True)].attribution_history[:64], ow.attribution_history,
                               rtol=0, atol=1e-10)


# ------------------------------------------------------------------ BASELINE config 5: p = 5000
def test_p5000_lifts_against_the_oracle():
    """One ordering at C5's feature count (p = 5000, reg = 1e-2), compared with the oracle's per-ordering
    algorithm (QR + triangular solve + GEMM: ls_spa/ls_spa.py:256-287) on the reduced problem the device
    holds: fp64 device path <= 1e-10, fp32 device path <= 1e-4 against the same oracle value."""
    import torch
    from ls_spa._engine import HipEngine
    p, n = 5000, 12000
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    Xa = torch.randn(n, p, dtype=torch.float64, device=dev, generator=gen)
    Xe = torch.randn(n, p, dtype=torch.float64, device=dev, generator=gen)
    w = torch.randn(p, dtype=torch.float64, device=dev, generator=gen)
    ya = Xa @ w + torch.randn(n, dtype=torch.float64, device=dev, generator=gen)
    ye = Xe @ w + torch.randn(n, dtype=torch.float64, device=dev, generator=gen)
    torch.cuda.synchronize()
    eng = HipEngine(0)
    try:
        eng.load_device_data(Xa.data_ptr(), p, ya.data_ptr(), n, Xe.data_ptr(), p, ye.data_ptr(), n, p, 1e-2)
        # the Gram reduction itself against fp64 GEMMs on the same device data
        G, g, H, h = eng.gram()
        np.testing.assert_allclose(G, (Xa.T @ Xa / n).cpu().numpy() + 1e-2 * np.eye(p), rtol=0, atol=1e-11)
        np.testing.assert_allclose(h, (Xe.T @ ye).cpu().numpy(), rtol=1e-12, atol=1e-8)
        del Xa, Xe
        torch.cuda.empty_cache()
        # reference-layout factors for the oracle (upper triangular, R^T R = G), from the device's Gram matrices
        R = np.linalg.cholesky(G).T
        F = np.linalg.cholesky(H).T
        q = np.linalg.solve(R.T, g)
        qt = np.linalg.solve(F.T, h)
        perm = np.random.default_rng(11).permutation(p)
        want = O.ordering_lift(R, F, q, qt, eng.y_norm_sq, perm)
        got64 = eng.run_batch(perm[None, :], False, want_lifts=True, accumulate=False)[0]
        assert eng.info() == 0
        np.testing.assert_allclose(got64, want, **LIFT_TOL)
        assert abs(got64.sum() - want.sum()) < 1e-10
        # antithetical pairing at this size: the pair's mean of (ordering, reversed ordering)
        rev64 = eng.run_batch(perm[None, ::-1].copy(), False, want_lifts=True, accumulate=False)[0]
        pair = eng.run_batch(perm[None, :], True, want_lifts=True, accumulate=False)[0]
        np.testing.assert_allclose(pair, 0.5 * (got64 + rev64), rtol=0, atol=1e-13)
        eng.set_precision("float32")
        got32 = eng.run_batch(perm[None, :], False, want_lifts=True, accumulate=False)[0]
        assert eng.info() == 0
        np.testing.assert_allclose(got32, want, rtol=0, atol=1e-4)
    finally:
        eng.close()


def test_c5_full_size_from_host_arrays():
    """BASELINE config 5 at its stated size through the public call: p = 5000, N = M = 200000, reg = 1e-2,
    method='argsort', float32 data (2 x 4 GB of host memory streamed into the fp64-accumulating Gram
    reduction), float32 per-ordering work, one batch of 128 antithetical samples (= 256 orderings of 5000
    features) with the device-side error estimate.  Size-independent checks (the oracle cannot run at this size;
    test_p5000_lifts_against_the_oracle pins the per-ordering arithmetic at this p)."""
    p, n, reg = 5000, 200000, 1e-2
    rng = np.random.default_rng(0)
    Xa = rng.standard_normal((n, p), dtype=np.float32)
    Xe = rng.standard_normal((n, p), dtype=np.float32)
    w = rng.standard_normal(p, dtype=np.float32)
    ya = Xa @ w + rng.standard_normal(n, dtype=np.float32)
    ye = Xe @ w + rng.standard_normal(n, dtype=np.float32)
    kw = dict(reg=reg, method="argsort", batch_size=128, num_batches=1, tolerance=0.0, seed=42,
              error_estimator="device", precision="float32")
    res = ls_spa(Xa, Xe, ya, ye, **kw)
    assert res.attribution.shape == (p,) and np.all(np.isfinite(res.attribution))
    # every lift vector sums to the full model's R^2 (fp32 per-ordering work: stated tolerance 1e-4)
    assert abs(res.attribution.sum() - res.r_squared) < 1e-4
    # theta solves the ridge normal equations (X^T X / N + reg I) theta = X^T y / N  (matrix-vector products only)
    th = res.theta
    resid = (Xa.T @ (Xa @ th.astype(np.float32) - ya)).astype(np.float64) / n + reg * th   # two fp32 GEMVs, no 8 GB upcast
    assert np.abs(resid).max() < 1e-4 * max(1.0, np.abs(th).max())
    pred = Xe @ th.astype(np.float32)
    r2 = 1.0 - float(np.sum((ye - pred).astype(np.float64) ** 2)) / float(ye.astype(np.float64) @ ye.astype(np.float64))
    assert abs(res.r_squared - r2) < 1e-4
    assert 0.9 < res.r_squared < 1.0
    assert len(res.error_history) == 2 and res.overall_error == res.error_history[-1]   # i = 127 (max_samples - 1), 128
    assert 0 < res.overall_error < 1e-2
    # relevant features carry the attribution: largest |theta| get the largest shares
    top = np.argsort(-np.abs(th))[:50]
    assert res.attribution[top].mean() > 5 * res.attribution.mean()


# ------------------------------------------------------------------ large p pinned to the REFERENCE itself
@pytest.mark.parametrize("name", ["large_p1000", "large_p5000"])
def test_large_p_against_reference_fixture(large_case, name):
    """The HIP path at the benchmark feature counts against lift vectors produced by the reference's own
    reduce_data + square_shapley (ls_spa/ls_spa.py:256-318; tests/golden/make_golden_large.py) -- not against the
    oracle: p = 1000 (C3 / C4; eight orderings + the reference driver's attribution on them) and p = 5000 (C5; one
    ordering and its reverse).  fp64 <= 1e-10, fp32 per-ordering work <= 1e-4, theta 1e-9 relative."""
    from ls_spa._engine import HipEngine
    g, d = large_case(name)
    p, reg = int(g["p"]), float(g["reg"])
    orders = g["orders"].astype(np.int32)
    eng = HipEngine(0)
    try:
        eng.load_data(*d, reg)
        assert eng.y_norm_sq == pytest.approx(float(g["y_norm_sq"]), rel=1e-13)
        got = eng.run_batch(orders, False, want_lifts=True, accumulate=False)
        np.testing.assert_allclose(got, g["lifts"], rtol=0, atol=1e-10)
        theta, r2, info = eng.full_fit()
        assert info == 0
        np.testing.assert_allclose(theta, g["theta"], rtol=1e-9, atol=1e-12)
        assert abs(r2 - float(g["r_squared"])) < 1e-11
        eng.set_precision("float32")
        got32 = eng.run_batch(orders, False, want_lifts=True, accumulate=False)
        np.testing.assert_allclose(got32, g["lifts"], rtol=0, atol=1e-4)
    finally:
        eng.close()
    if name == "large_p1000":
        res = ls_spa(*d, reg=reg, perms=g["orders"].astype(np.int64), batch_size=4, tolerance=0.0)
        np.testing.assert_allclose(res.attribution, g["attribution"], rtol=0, atol=1e-10)
        np.testing.assert_allclose(res.theta, g["drv_theta"], rtol=1e-9, atol=1e-12)
        assert abs(res.r_squared - float(g["drv_r_squared"])) < 1e-11
        # the reference driver's own error estimates on these eight samples (ls_spa/ls_spa.py:222-236, :321-341) as
        # statistical pins for all three estimators: they draw other normals (the reference's Cholesky-or-SVD branch
        # decides how many), the distribution is the same -- 0.95-quantiles of 1024 draws scatter by a few
        # per cent
        for est in ("reference", "lowrank", "device"):
            r = ls_spa(*d, reg=reg, perms=g["orders"].astype(np.int64), batch_size=4, tolerance=0.0, error_estimator=est)
            np.testing.assert_allclose(r.attribution, g["attribution"], rtol=0, atol=1e-10)
            assert len(r.error_history) == len(g["drv_error_history"]) == 2
            np.testing.assert_allclose(r.error_history, g["drv_error_history"], rtol=0.25)
            assert r.overall_error == pytest.approx(float(g["drv_overall_error"]), rel=0.25)
            ratio = r.attribution_errors / g["drv_attribution_errors"]
            assert abs(np.median(ratio) - 1.0) < 0.1 and ratio.min() > 0.6 and ratio.max() < 1.6, (est, ratio.min(), ratio.max())


def test_feature_count_beyond_64kb_of_gather_lds():
    """p = 6000: the gather's row + ordering staging is 72 KB, past the 64 KB a kernel gets by default (the limit
    that used to surface as hipErrorInvalidValue at the first batch; C5 sat 7 % under it).  One ordering against
    the oracle; a p beyond what 160 KB of LDS hold is refused by lsspa_reduce, by name."""
    from ls_spa._engine import HipEngine
    p, n = 6000, 6400
    d = O.gaussian_workload(p, n, n, seed=6)
    R, F, q, qt = O.reduce(*d, 1e-3)
    yy = float(np.linalg.norm(d[3]) ** 2)
    perm = np.random.default_rng(60).permutation(p)
    want = O.ordering_lift(R, F, q, qt, yy, perm)
    eng = HipEngine(0)
    try:
        eng.load_data(*d, 1e-3)
        got = eng.run_batch(perm[None, :].astype(np.int32), True, want_lifts=True, accumulate=False)[0]
        want_pair = 0.5 * (want + O.ordering_lift(R, F, q, qt, yy, perm[::-1]))
        np.testing.assert_allclose(got, want_pair, rtol=0, atol=1e-10)
        theta, _, info = eng.full_fit()            # the back-substitution keeps p doubles in LDS too
        assert info == 0
        np.testing.assert_allclose(theta, np.linalg.lstsq(R, q, rcond=None)[0], rtol=1e-8, atol=1e-11)
        # beyond the supported count: refused at the reduction, with a message that names p
        big = 32768
        z = np.zeros((big, big), dtype=np.float32)
        y = np.ones(big, dtype=np.float32)
        with pytest.raises(ValueError, match=f"p = {big} features exceeds"):
            eng.load_data(z, z, y, y, 0.0)
    finally:
        eng.close()


@pytest.mark.parametrize("estimator", ["reference", "device"])
def test_two_lanes_in_the_driver_change_nothing(estimator):
    """ls_spa(lanes=2): successive chunk groups on two workspaces and two streams, the next group launched before the
    current one's statistics are read back.  Same kernels on the same orderings, statistics in chunk order on one
    stream: every number of the one-lane run, bit for bit -- also when the stop rule drops a group launched ahead."""
    d = O.gaussian_workload(150, 900, 800, seed=21)
    kw = dict(reg=1e-3, method="argsort", seed=4, batch_size=16, max_samples=112, error_estimator=estimator)
    one = ls_spa(*d, tolerance=0.0, lanes=1, **kw)
    two = ls_spa(*d, tolerance=0.0, lanes=2, **kw)
    auto = ls_spa(*d, tolerance=0.0, lookahead=3, **kw)        # lanes='auto' -> 2 at p = 150, groups of three chunks
    for r in (two, auto):
        np.testing.assert_array_equal(r.attribution, one.attribution)
        np.testing.assert_array_equal(r.error_history, one.error_history)
        np.testing.assert_array_equal(r.attribution_errors, one.attribution_errors)
        assert r.r_squared == one.r_squared
    assert len(one.error_history) == 8          # checks at 16 .. 96, at max_samples - 1 and the trailing one
    tol = float(one.error_history[2]) * 1.0000001               # stops at the third check: a group in flight is dropped
    if one.error_history[0] > tol and one.error_history[1] > tol:
        a = ls_spa(*d, tolerance=tol, lanes=1, **kw)
        b = ls_spa(*d, tolerance=tol, lanes=2, **kw)
        assert len(a.error_history) == len(b.error_history) == 3
        np.testing.assert_array_equal(a.attribution, b.attribution)
    # chunks of 64 samples and more go to the two lanes as half-chunks: same orderings, same check indices; the
    # statistics are merged half-chunk by half-chunk, so the numbers agree to the rounding of that grouping
    kw = dict(reg=1e-3, method="argsort", seed=4, batch_size=80, max_samples=240, error_estimator=estimator, tolerance=0.0)
    a = ls_spa(*d, lanes=1, **kw)
    b = ls_spa(*d, lanes=2, **kw)
    assert len(a.error_history) == len(b.error_history) == 4      # 80, 160, 239, 240
    np.testing.assert_allclose(b.attribution, a.attribution, rtol=0, atol=1e-15)
    # ('reference': LAPACK's Cholesky-or-SVD branch on the singular sample covariance is decided by round-off, and with
    # it how many normals are drawn -- a statistical pin there)
    np.testing.assert_allclose(b.error_history, a.error_history, rtol=1e-9 if estimator == "device" else 0.3)
    np.testing.assert_allclose(b.theta, a.theta, rtol=0, atol=0)


@pytest.mark.slow
def test_feature_count_beyond_the_lds_of_a_cu():
    """p = 13700: a source row and the ordering (12 B a padded feature) no longer fit the 160 KB of LDS of a CU -- the
    ceiling of rounds 1-3 (p <= 13567; the reference has none, ls_spa/ls_spa.py:163).  The segmented gather and the
    back-substitution with its right-hand side in memory take over.  One ordering, fp32 per-ordering work, against
    the oracle's QR-based lift on the same reduced problem (minutes of host BLAS: marked slow)."""
    from ls_spa._engine import HipEngine
    p, n = 13700, 27400
    rng = np.random.default_rng(137)
    Xa = rng.standard_normal((n, p), dtype=np.float32)
    Xe = rng.standard_normal((n, p), dtype=np.float32)
    w = (rng.standard_normal(p) / np.sqrt(p)).astype(np.float32)
    ya = Xa @ w + rng.standard_normal(n, dtype=np.float32)
    ye = Xe @ w + rng.standard_normal(n, dtype=np.float32)
    perm = rng.permutation(p)
    eng = HipEngine(0)
    try:
        eng.set_precision("float32")
        eng.load_data(Xa, Xe, ya, ye, 1e-2)
        assert eng.tri
        got = eng.run_batch(perm[None, :].astype(np.int32), False, want_lifts=True, accumulate=False)[0]
        theta, r2, info = eng.full_fit()
        assert info == 0 and eng.info() == 0
        # the antithetical pair above the old ceiling, in both precisions (round 5: the segmented gather writes each
        # ordering of a pair by itself while the lift finish reads them as a pair): the pair's lift vector is the mean of
        # the ordering's and its reverse's, each evaluated alone; with the R^2 known, every batch is also sum-checked
        rev = eng.run_batch(np.ascontiguousarray(perm[None, ::-1]).astype(np.int32), False, want_lifts=True,
                            accumulate=False)[0]
        pair = eng.run_batch(perm[None, :].astype(np.int32), True, want_lifts=True, accumulate=False)[0]
        np.testing.assert_allclose(pair, 0.5 * (got + rev), rtol=0, atol=2e-6)
        eng.set_precision("float64")
        f64_f = eng.run_batch(perm[None, :].astype(np.int32), False, want_lifts=True, accumulate=False)[0]
        f64_r = eng.run_batch(np.ascontiguousarray(perm[None, ::-1]).astype(np.int32), False, want_lifts=True,
                              accumulate=False)[0]
        f64_p = eng.run_batch(perm[None, :].astype(np.int32), True, want_lifts=True, accumulate=False)[0]
        np.testing.assert_allclose(f64_p, 0.5 * (f64_f + f64_r), rtol=0, atol=1e-13)
        np.testing.assert_allclose(f64_f, got, rtol=0, atol=1e-4)
        assert eng.info() == 0 and eng.sum_deviation() < 1e-4
        G, g, H, h = eng.gram()           # the fp64 Gram reduction (pinned against numpy elsewhere) is the oracle's input
        yy = eng.y_norm_sq
    finally:
        eng.close()
    del Xa, Xe
    assert abs(got.sum() - r2) < 1e-4      # every ordering's lifts sum to the full model's R^2
    R, F = np.linalg.cholesky(G).T, np.linalg.cholesky(H).T
    q, qt = sla.solve_triangular(R, g, trans="T"), sla.solve_triangular(F, h, trans="T")
    want = O.ordering_lift(R, F, q, qt, yy, perm)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-4)
    np.testing.assert_allclose(theta, np.linalg.solve(G, g), rtol=2e-3, atol=2e-4)


# ------------------------------------------------------------------ streamed reduction: caller memory in awkward places
def test_streamed_reduction_host_memory_corner_cases():
    """lsspa_reduce from host arrays, through the C ABI, in the three situations the round-2 review listed for the
    abort once seen in this path: (i) X and y carved out of ONE buffer so that they share pages, each >= 8 MB;
    (ii) arrays the caller has already page-locked; (iii) a strided X (ld > p) whose last row ends exactly at the end
    of a memory mapping, so that n * ld elements would reach past it.  Each must give the Gram matrices of the plain
    call, and leave the caller's memory usable (still registered in (ii)).  The library reads the caller's pages
    through the runtime's ordinary copies only (the page-locking path of rounds 2-3 was removed in round 4)."""
    import ctypes as C
    import mmap
    from ls_spa import _native as N
    from ls_spa._engine import HipEngine
    import torch
    p, n = 640, 4200                      # X: 4200 x 640 x 8 B = 21.5 MB per side
    rng = np.random.default_rng(33)
    eng = HipEngine(0)
    lib = eng._lib

    def reduce_host(Xa, ya, Xe, ye, ld):
        eng._check(lib.lsspa_reduce(eng._h, Xa.ctypes.data, ld, ya.ctypes.data, n, Xe.ctypes.data, ld,
                                    ye.ctypes.data, n, p, 0.0, N.F64, N.HOST))
        eng._refresh_dims()
        return eng.gram()

    try:
        # reference result: well-separated, aligned arrays
        Xa, Xe = rng.standard_normal((n, p)), rng.standard_normal((n, p))
        ya, ye = rng.standard_normal(n), rng.standard_normal(n)
        base = reduce_host(Xa, ya, Xe, ye, p)
        np.testing.assert_allclose(base[0], Xa.T @ Xa / n, rtol=0, atol=1e-12)

        # (i) one buffer: [pad | X_train | y_train (>= 8 MB: padded with unused tail) | X_test | y_test], no alignment
        ny = (8 << 20) // 8 + 3            # y blocks of >= 8 MB of which the first n entries are used
        total = 5 + 2 * n * p + 2 * ny
        buf = np.empty(total + 1)[1:]      # odd offset: base not even 16-byte aligned
        o = 5
        Xa1 = buf[o:o + n * p].reshape(n, p); o += n * p
        ya1 = buf[o:o + ny]; o += ny
        Xe1 = buf[o:o + n * p].reshape(n, p); o += n * p
        ye1 = buf[o:o + ny]
        Xa1[:], Xe1[:], ya1[:n], ye1[:n] = Xa, Xe, ya, ye
        ya1[n:], ye1[n:] = 7.0, 7.0
        assert Xa1.ctypes.data % 4096 and (Xa1.ctypes.data + Xa1.nbytes) // 4096 == ya1.ctypes.data // 4096
        got = reduce_host(Xa1, ya1, Xe1, ye1, p)
        for a, b in zip(got, base):
            np.testing.assert_array_equal(a, b)
        assert ya1[n] == 7.0 and np.array_equal(Xa1, Xa)        # untouched

        # (ii) caller-pinned arrays (hipHostMalloc'ed by torch): left alone, still pinned afterwards
        tXa, tXe = torch.from_numpy(Xa).pin_memory(), torch.from_numpy(Xe).pin_memory()
        got = reduce_host(tXa.numpy(), ya, tXe.numpy(), ye, p)
        for a, b in zip(got, base):
            np.testing.assert_array_equal(a, b)
        assert tXa.is_pinned() and tXe.is_pinned()
        dev = tXa.to("cuda:0", non_blocking=True)               # the pinned block still serves its owner
        torch.cuda.synchronize()
        assert torch.equal(dev.cpu(), tXa)
        del dev, tXa, tXe

        # (iii) ld > p with the array's last element on the last bytes of a mapping
        ld = p + 24
        need = ((n - 1) * ld + p) * 8
        size = -(-need // mmap.PAGESIZE) * mmap.PAGESIZE
        maps = [mmap.mmap(-1, size + mmap.PAGESIZE) for _ in range(2)]
        views = []
        for m in maps:
            whole = np.frombuffer(m, dtype=np.uint8)
            # fence the page behind the array's end: those bytes are not ours to read, let alone to page-lock
            addr = whole.ctypes.data
            libc = C.CDLL(None, use_errno=True)
            assert libc.mprotect(C.c_void_p(addr + size), C.c_size_t(mmap.PAGESIZE), 0) == 0     # PROT_NONE
            flat = np.frombuffer(m, dtype=np.float64, count=size // 8)[(size - need) // 8:]
            views.append(np.lib.stride_tricks.as_strided(flat, shape=(n, p), strides=(ld * 8, 8)))
        views[0][:], views[1][:] = Xa, Xe
        got = reduce_host(views[0], ya, views[1], ye, ld)
        for a, b in zip(got, base):
            np.testing.assert_array_equal(a, b)
    finally:
        eng.close()
