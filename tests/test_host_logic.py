"""CPU tests of the host side: public surface, driver loop (against fixtures made by the real
reference), samplers, statistics helpers and the C-ABI export list.  The orderings are evaluated
by a test double backed by the oracle (tests/oracle_engine.py); the GPU runs of the same cases
live in test_gpu_parity.py."""
import ctypes
import inspect
import os
import re

import numpy as np
import pytest

import ls_spa as pkg
from ls_spa import (ShapleyResults, SizeIncompatible, error_estimates, ls_spa, merge_sample_cov,
                    merge_sample_mean, validate_data)
from ls_spa import _samplers as S
from oracle_engine import OracleEngine

TOL = dict(rtol=0, atol=1e-12)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def data_of(g):
    return [g[k] for k in ("X_train", "X_test", "y_train", "y_test")]


# ---------------------------------------------------------------- public surface
def test_public_names_and_signature():
    for name in ("ls_spa", "ShapleyResults", "SizeIncompatible", "validate_data", "merge_sample_mean",
                 "merge_sample_cov", "square_shapley", "reduce_data", "error_estimates"):
        assert hasattr(pkg, name), name
    sig = inspect.signature(ls_spa)
    names = list(sig.parameters)
    assert names[:12] == ["X_train", "X_test", "y_train", "y_test", "reg", "max_samples", "batch_size",
                          "tolerance", "seed", "perms", "antithetical", "return_attribution_history"]
    d = {k: v.default for k, v in sig.parameters.items()}
    assert (d["reg"], d["max_samples"], d["batch_size"], d["tolerance"], d["seed"], d["perms"],
            d["antithetical"], d["return_attribution_history"]) == (0., 2 ** 13, 2 ** 8, 1e-2, 42, None, True, False)
    for extra in ("method", "num_batches", "return_history"):
        assert sig.parameters[extra].kind is inspect.Parameter.KEYWORD_ONLY
    fields = list(ShapleyResults.__dataclass_fields__)
    assert fields == ["attribution", "theta", "overall_error", "attribution_errors", "r_squared",
                      "error_history", "attribution_history"]


def test_repr_matches_reference(golden):
    g = golden("toy")
    res = ls_spa(*data_of(g), _engine=OracleEngine())
    assert repr(res) == str(g["repr"])
    long = ShapleyResults(np.arange(7) / 3, np.arange(7) / 7, 1.5e-3, np.zeros(7), 0.5, np.zeros(0), None)
    text = repr(long)
    assert "(0.00, 0.33, 0.67, 1.00, 1.33, ...)" in text and "1.50E-03" in text and "p = 7" in text


def test_validate_messages():
    a, b = np.zeros((5, 3)), np.zeros((4, 3))
    cases = [
        ((a, np.zeros((4, 2)), np.zeros(5), np.zeros(4)), "same number of columns"),
        ((a, b, np.zeros(4), np.zeros(4)), "rows as y_train"),
        ((a, b, np.zeros(5), np.zeros(3)), "rows as y_test"),
        ((np.zeros((2, 3)), b, np.zeros(2), np.zeros(4)), "at most the number of observations"),
    ]
    for args, frag in cases:
        with pytest.raises(SizeIncompatible) as e:
            validate_data(*args)
        assert frag in e.value.message
    validate_data(a, b, np.zeros(5), np.zeros(4))
    with pytest.raises(SizeIncompatible):
        ls_spa(a, np.zeros((4, 2)), np.zeros(5), np.zeros(4), _engine=OracleEngine())
    with pytest.raises(ValueError):   # 2-D y, reference: ValueError from concatenate
        ls_spa(np.ones((5, 3)), np.ones((4, 3)), np.ones((5, 1)), np.ones(4), _engine=OracleEngine())


# ---------------------------------------------------------------- statistics helpers
def test_merge_helpers(golden):
    g = golden("merge")
    X = g["X"]
    a, b = X[:200], X[200:]
    np.testing.assert_allclose(merge_sample_mean(a.mean(0), b.mean(0), 200, 300), g["merged_mean"], atol=1e-14)
    got = merge_sample_cov(a.mean(0), b.mean(0), np.cov(a, rowvar=False, bias=True),
                           np.cov(b, rowvar=False, bias=True), 200, 300)
    np.testing.assert_allclose(got, g["merged_cov"], rtol=1e-13, atol=1e-12)
    # the reference's own assertions (test/test_ls_spa.py:20-44), 7 decimals
    np.testing.assert_almost_equal(X.mean(0), merge_sample_mean(a.mean(0), b.mean(0), 200, 300))
    np.testing.assert_almost_equal(np.cov(X, rowvar=False, bias=True), got)


def test_error_estimates_stream(golden):
    g = golden("edge")
    rng = np.random.default_rng(17)
    feat, total = error_estimates(rng, g["ee_full_cov"])
    np.testing.assert_allclose(feat, g["ee_full_feat"], rtol=1e-9)
    np.testing.assert_allclose(total, float(g["ee_full_total"]), rtol=1e-9)
    np.testing.assert_array_equal(rng.standard_normal(4), g["ee_full_next"])
    feat, total = error_estimates(np.random.default_rng(17), g["ee_low_cov"])
    np.testing.assert_allclose(total, float(g["ee_low_total"]), rtol=0.25)


def test_lowrank_estimator_matches_distribution():
    rng = np.random.default_rng(0)
    lifts = rng.standard_normal((200, 15)) @ rng.standard_normal((15, 15)) * 1e-2
    c = lifts - lifts.mean(0)
    cov = np.cov(lifts, rowvar=False) / len(lifts)
    f1, t1 = error_estimates(np.random.default_rng(1), cov)
    f2, t2 = pkg.error_estimates_lowrank(np.random.default_rng(2), c)
    assert abs(t1 - t2) / t1 < 0.1
    np.testing.assert_allclose(f1, f2, rtol=0.2)


# ---------------------------------------------------------------- samplers
def test_philox_known_answers():
    """The counter-based generator behind the device estimator's normals, against its published vectors."""
    import philox_ref as P
    for ctr, key, want in P.KAT:
        got = P.philox4x32_10(np.array([ctr], dtype=np.uint64), key)[0]
        assert [int(v) for v in got] == list(want)
    x = P.normals(7, np.arange(1000, 1400))
    from scipy import stats
    assert stats.kstest(x.ravel(), "norm").pvalue > 1e-3
    assert abs(x.mean()) < 0.01 and abs(x.std() - 1.0) < 0.01
    # a sample's column depends on (seed, sample id) only
    np.testing.assert_array_equal(P.normals(7, [1003])[:, 0], x[:, 3])
    assert not np.array_equal(P.normals(8, [1003])[:, 0], x[:, 3])


def test_device_estimator_is_the_running_form(golden):
    """'device' (test double): at every check the draws are Xi (L - 1 mean^T) / sqrt(n (n - 1)) with Xi the
    counter-based normals of the samples so far -- accumulated chunk by chunk as D = Xi L, s = Xi 1 -- and the
    reference's quantiles of them (ls_spa/ls_spa.py:337-340); the host generator is not touched; against the host
    low-rank form (other normals, same distribution) the pin is statistical."""
    import philox_ref as P
    g = golden("p12")
    d = [g[k] for k in ("X_train", "X_test", "y_train", "y_test")]
    kw = dict(perms=g["perms64"][:40], batch_size=16, tolerance=0.0)
    low = pkg.ls_spa(*d, error_estimator="lowrank", return_attribution_history=True, _engine=OracleEngine(), **kw)
    eng = OracleEngine()
    dev = pkg.ls_spa(*d, error_estimator="device", _engine=eng, **kw)
    np.testing.assert_allclose(dev.attribution, low.attribution, rtol=0, atol=1e-15)
    np.testing.assert_allclose(dev.error_history, low.error_history, rtol=0.15)
    assert eng.enqueued == 3
    # the lift vectors, from the running means of the history
    hist = low.attribution_history
    lifts = np.diff(np.vstack([np.zeros(12), hist * np.arange(1, 41)[:, None]]), axis=0)
    seed = int(np.random.SeedSequence(42).generate_state(1, np.uint64)[0])
    want = []
    for n in (16, 32, 40):
        L = lifts[:n]
        x = P.normals(seed, np.arange(n)) @ (L - L.mean(0)) / np.sqrt(n * (n - 1.0))
        want.append(np.quantile(np.linalg.norm(x, axis=1), 0.95))
        feat = np.quantile(np.abs(x), 0.95, axis=0)
    np.testing.assert_allclose(dev.error_history, want, rtol=1e-9)
    np.testing.assert_allclose(dev.attribution_errors, feat, rtol=1e-9)
    with pytest.raises(ValueError, match="error_estimator"):
        pkg.ls_spa(*d, error_estimator="gpu", _engine=OracleEngine(), **kw)


def test_deferred_checks_give_the_stopping_checks_results(golden):
    """QMC methods, device estimator: the stop rule of check k is evaluated when check k + 1 has been enqueued (the
    first check of a run is waited for).  A run that stops returns the numbers of the stopping check -- its own copy
    of the running mean -- exactly as the run that waits for every check; what was launched beyond is dropped."""
    g = golden("p12")
    d = [g[k] for k in ("X_train", "X_test", "y_train", "y_test")]
    base = dict(method="argsort", seed=5, max_samples=96, batch_size=16, lookahead=1)
    full = ls_spa(*d, tolerance=0.0, _engine=OracleEngine(), **base)
    wait = ls_spa(*d, tolerance=0.0, _engine=OracleEngine(), _defer=0, **base)
    np.testing.assert_array_equal(full.error_history, wait.error_history)
    np.testing.assert_array_equal(full.attribution, wait.attribution)
    assert len(full.error_history) == 7
    for k in (0, 1, 3, 5, 6):       # stop at the first check, in the middle, at the check before the last, at the last
        tol = float(full.error_history[k]) * 1.0000001
        assert all(e > tol for e in full.error_history[:k])
        e0, e1 = OracleEngine(), OracleEngine()
        a = ls_spa(*d, tolerance=tol, _engine=e0, _defer=0, **base)
        b = ls_spa(*d, tolerance=tol, _engine=e1, **base)
        np.testing.assert_array_equal(b.error_history, a.error_history)
        np.testing.assert_array_equal(b.attribution, a.attribution)
        np.testing.assert_array_equal(b.attribution_errors, a.attribution_errors)
        assert b.overall_error == a.overall_error and len(b.error_history) == k + 1
        # waited for: nothing beyond the stop; deferred: one chunk more was evaluated (none after the first check,
        # which is waited for, and none when the run's samples were used up anyway)
        extra = 0 if k in (0, 6) else (15 if k == 4 else 16)
        assert sum(e0.calls) == min(16 * (k + 1), 96 if k == 6 else 95 if k == 5 else 10 ** 9)
        assert sum(e1.calls) == sum(e0.calls) + (1 if k == 5 else extra)
    # the default look-ahead of this method and estimator ('auto': chunks of 16 samples at p = 12 go up to sixteen to a
    # launch, in groups that grow -- 1, 2, 4, ... chunks -- and as many checks as the group has may be outstanding): the
    # same numbers again; the stop at check 4 falls into the third group
    tol = float(full.error_history[3]) * 1.0000001
    e8 = OracleEngine()
    c = ls_spa(*d, tolerance=tol, _engine=e8, **dict(base, lookahead=None))
    a = ls_spa(*d, tolerance=tol, _engine=OracleEngine(), _defer=0, **base)
    np.testing.assert_array_equal(c.error_history, a.error_history)
    np.testing.assert_array_equal(c.attribution, a.attribution)
    assert e8.launched == 3 and len(c.error_history) == 4
    assert e8.group_calls == 3            # ... and every group's chunks went through ONE library call (lsspa_group_collect)
    assert e8.calls == [16, 16, 16, 16, 16, 15, 1]       # groups of 1, 2 and 4 chunks (the check at max_samples - 1 cuts one)
    e3 = OracleEngine()
    c3 = ls_spa(*d, tolerance=0.0, _engine=e3, **dict(base, lookahead=3))
    np.testing.assert_array_equal(c3.error_history, full.error_history)
    np.testing.assert_array_equal(c3.attribution, full.attribution)
    assert e3.group_calls == 3 and e3.calls == [16, 16, 16, 16, 16, 15, 1]
    # histories are cut at the stop as well
    tol = float(full.error_history[2]) * 1.0000001
    a = ls_spa(*d, tolerance=tol, return_attribution_history=True, _engine=OracleEngine(), _defer=0, **base)
    b = ls_spa(*d, tolerance=tol, return_attribution_history=True, _engine=OracleEngine(), **base)
    assert b.attribution_history.shape == a.attribution_history.shape == (48, 12)
    np.testing.assert_array_equal(b.attribution_history, a.attribution_history)


def test_default_estimator_fits_the_method(golden):
    """error_estimator=None: the reference's estimator wherever the reference's code has a behaviour to mirror (its
    shared generator makes the estimator's draws observable: ls_spa/ls_spa.py:168-175, :224) -- seed path, 'random',
    any perms= -- and the device-side thin form for the QMC methods its code does not implement."""
    from oracle_engine import OracleEngine
    g = golden("p12")
    d = [g[k] for k in ("X_train", "X_test", "y_train", "y_test")]
    base = dict(max_samples=48, batch_size=16, tolerance=0.0, seed=5)
    for method in ("argsort", "permutohedron"):
        eng = OracleEngine()
        auto = pkg.ls_spa(*d, method=method, _engine=eng, **base)
        assert eng.enqueued == 4                               # checks at 16, 32, 47 and the trailing one, on the engine
        dev = pkg.ls_spa(*d, method=method, error_estimator="device", _engine=OracleEngine(), **base)
        np.testing.assert_array_equal(auto.error_history, dev.error_history)
        np.testing.assert_array_equal(auto.attribution_errors, dev.attribution_errors)
    for kw in (dict(base), dict(base, method="random"), dict(perms=g["perms64"][:48], batch_size=16, tolerance=0.0)):
        eng = OracleEngine()
        auto = pkg.ls_spa(*d, _engine=eng, **kw)
        assert getattr(eng, "enqueued", 0) == 0
        ref = pkg.ls_spa(*d, error_estimator="reference", _engine=OracleEngine(), **kw)
        np.testing.assert_array_equal(auto.error_history, ref.error_history)
        np.testing.assert_array_equal(auto.attribution, ref.attribution)


@pytest.mark.parametrize("method,estimator", [(None, "reference"), ("argsort", "lowrank"),
                                              ("permutohedron", "device"), ("random", "device")])
def test_checkpoint_resume_continues_the_same_run(golden, tmp_path, method, estimator):
    """Kill the run in its fourth batch, resume: orderings, estimator draws and every number equal the
    uninterrupted run (generator state, QMC position, running moments and lift history restored)."""
    class Dies(OracleEngine):
        def run_batch(self, *a, **k):
            if len(self.calls) == 3:
                raise KeyboardInterrupt
            return super().run_batch(*a, **k)

    g = golden("p12")
    d = [g[k] for k in ("X_train", "X_test", "y_train", "y_test")]
    kw = dict(batch_size=16, tolerance=0.0, seed=11, method=method, error_estimator=estimator, lookahead=1)
    full = pkg.ls_spa(*d, max_samples=112, _engine=OracleEngine(), **kw)
    ck = str(tmp_path / "state.npz")
    with pytest.raises(KeyboardInterrupt):
        pkg.ls_spa(*d, max_samples=112, checkpoint=ck, _engine=Dies(), **kw)
    with np.load(ck) as z:
        assert int(z["n"]) == 48 and len(z["error_history"]) == 3
    second = pkg.ls_spa(*d, max_samples=112, checkpoint=ck, _engine=OracleEngine(), **kw)
    np.testing.assert_allclose(second.attribution, full.attribution, rtol=0, atol=1e-14)
    np.testing.assert_allclose(second.error_history, full.error_history, rtol=1e-9)
    np.testing.assert_allclose(second.attribution_errors, full.attribution_errors, rtol=1e-9)
    # a finished run resumes to itself without sampling again
    eng = OracleEngine()
    third = pkg.ls_spa(*d, max_samples=112, checkpoint=ck, _engine=eng, **kw)
    assert eng.calls == []
    np.testing.assert_allclose(third.attribution, full.attribution, rtol=0, atol=1e-14)
    with pytest.raises(ValueError, match="seed"):
        pkg.ls_spa(*d, max_samples=112, checkpoint=ck, _engine=OracleEngine(), **dict(kw, seed=12))


def test_samplers_match_fixtures(golden):
    g = golden("samplers_p12")
    np.testing.assert_allclose(S.helmert_rows(12), g["U"], atol=1e-15)
    src = S.ArgsortSource(12, 5, 32)
    got = np.concatenate([src.take(8), src.take(24), src.take(5)])   # chunked draws continue the sequence
    np.testing.assert_array_equal(got, g["argsort"])
    src = S.PermutohedronSource(12, 5, 32)
    got = np.concatenate([src.take(16), src.take(16)])
    np.testing.assert_array_equal(got, g["permutohedron"])
    assert len(src.take(4)) == 0


def test_sources_hand_out_a_ranks_share():
    """With several ranks ordering number g of the run belongs to rank g mod world (take_share).  The QMC sources do
    only their share's work -- the Sobol' points of a rank's orderings are computed directly from the engine's direction
    numbers (checked against SciPy's own output when the source is made) -- and must hand out exactly the rows of the
    full sequence, whatever the chunk sizes; the prefetching wrapper draws a rank's share in blocks of its own and must
    cut them at the consumer's chunk boundaries; every other source draws everything and keeps its share."""
    for cls in (S.ArgsortSource, S.PermutohedronSource):
        full = cls(23, 5, 1000).take(300)
        for world in (2, 3, 8):
            for rank in range(world):
                plain = cls(23, 5, 300)
                ahead = S.PrefetchedSource(cls(23, 5, 300), block=64, ahead=128, rank=rank, world=world)
                for src in (plain, ahead):
                    pos, rows, idx = 0, [], []
                    for cnt in (10, 100, 7, 120, 50, 50):
                        n, own = src.take_share(cnt, pos, rank, world)
                        idx.append(np.arange(pos + (rank - pos) % world, pos + n, world))
                        rows.append(own)
                        pos += n
                    assert pos == 300
                    np.testing.assert_array_equal(np.concatenate(rows), full[np.concatenate(idx)])
                ahead.close()
    rng = np.random.default_rng(3)
    for q in (2, 3, 12, 200):                       # the O(p) projection a rank's share goes through == the matrix product
        u = rng.standard_normal((5, q - 1))
        np.testing.assert_allclose(S.helmert_project(u), u @ S.helmert_rows(q), rtol=0, atol=1e-14)
    assert S.PermutohedronSource(23, 5, 10)._direct_normals() is not None, "direct QMC normals no longer match this SciPy"
    direct = S.ArgsortSource(23, 5, 10)._direct_points()
    assert direct is not None, "the direct Sobol' points no longer match this SciPy: the source fell back to drawing all"
    # a caller's iterable is consumed in full on every rank, as the one-process run consumes it
    src = S.IterableSource(iter(np.array([np.roll(np.arange(9), k) for k in range(20)])), 9)
    n, own = src.take_share(7, 0, 1, 3)
    assert n == 7 and [int(r[0]) for r in own] == [8, 5]           # orderings 1 and 4 of the first seven
    n, own = src.take_share(7, 7, 1, 3)
    assert n == 7 and [int(r[0]) for r in own] == [2, 8, 5]        # orderings 7, 10 and 13: dealt by number, not by place
    with pytest.raises(ValueError, match="out of turn"):
        S.PrefetchedSource(S.ArgsortSource(9, 1, 50), rank=0, world=2).take_share(4, 3, 0, 2)


def test_prefetched_source_in_one_process():
    """One process: the helper thread draws the orderings in blocks of its own ahead of the loop (PrefetchedSource);
    what comes out is the plain source's sequence, whatever the block size, the look-ahead limit and the sizes asked
    for -- also when the source ends inside a block, when it is asked for more than it has, and when it is closed with
    thousands of orderings never asked for."""
    for cls, p in ((S.ArgsortSource, 23), (S.PermutohedronSource, 23), (S.ArgsortSource, 140)):
        full = cls(p, 5, 1000).take(333)
        assert full.dtype == np.int32
        for block, ahead in ((7, 20), (64, 128), (100, 100), (400, 800)):
            src = S.PrefetchedSource(cls(p, 5, 333), block=block, ahead=ahead)
            rows, pos = [], 0
            for cnt in (1, 10, 100, 7, 120, 50, 80):
                n, own = src.take_share(cnt, pos, 0, 1)
                assert n == len(own) == min(cnt, 333 - pos)
                rows.append(own)
                pos += n
            assert pos == 333 and src.take_share(5, pos, 0, 1)[0] == 0
            np.testing.assert_array_equal(np.concatenate(rows), full)
            src.close()
        early = S.PrefetchedSource(cls(p, 5, 10 ** 6), block=256, ahead=4096)
        np.testing.assert_array_equal(early.take(20), full[:20])
        early.close()                      # thousands of orderings drawn ahead and never asked for
        assert not early._thread.is_alive() or early._thread.join(5) is None


def test_iterable_source_is_lazy():
    pulled = []

    def gen():
        for k in range(100):
            pulled.append(k)
            yield np.roll(np.arange(9), k)
    src = S.IterableSource(gen(), 9)
    assert src.take(4).shape == (4, 9) and len(pulled) == 4
    with pytest.raises(ValueError):
        S.IterableSource([[0, 1, 2]], 9).take(1)


# ---------------------------------------------------------------- driver against reference outputs
def test_toy_exact(golden):
    g = golden("toy")
    eng = OracleEngine()
    res = ls_spa(*data_of(g), _engine=eng)
    np.testing.assert_allclose(res.attribution, g["attribution"], **TOL)
    np.testing.assert_allclose(res.theta, g["theta"], **TOL)
    assert abs(res.r_squared - float(g["r_squared"])) < 1e-13
    assert res.overall_error == 0.0 and res.error_history.shape == (0,) and res.attribution_history is None
    np.testing.assert_array_equal(res.attribution_errors, np.zeros(3))
    assert sum(eng.calls) == 6          # 3! orderings, antithetical forced off


@pytest.mark.parametrize("p", [4, 8])
def test_exact_small_p(golden, p):
    g = golden(f"exact_p{p}")
    res = ls_spa(*data_of(g), _engine=OracleEngine())
    np.testing.assert_allclose(res.attribution, g["attribution"], **TOL)
    assert abs(res.attribution.sum() - res.r_squared) < 1e-12


@pytest.mark.parametrize("anti", [True, False])
def test_injected_perms_history_and_triggers(golden, anti):
    g = golden("p12")
    pre = f"drv_anti{int(anti)}_"
    eng = OracleEngine()
    res = ls_spa(*data_of(g), perms=g["perms64"], batch_size=16, tolerance=0.0, antithetical=anti,
                 return_attribution_history=True, _engine=eng)
    np.testing.assert_allclose(res.attribution, g[pre + "attribution"], **TOL)
    np.testing.assert_allclose(res.attribution_history, g[pre + "attribution_history"], **TOL)
    np.testing.assert_allclose(res.theta, g[pre + "theta"], **TOL)
    assert eng.calls == [16, 16, 16, 16]
    assert len(res.error_history) == len(g[pre + "error_history"]) == 4
    np.testing.assert_allclose(res.error_history, g[pre + "error_history"], rtol=0.15)
    np.testing.assert_allclose(res.overall_error, float(g[pre + "overall_error"]), rtol=0.15)


def test_perms_as_generator_and_progress_wrapper(golden):
    g = golden("p12")

    class Wrapper:            # stands in for marimo's progress bar: only __iter__ is used
        def __init__(self, rows):
            self.rows, self.n = rows, 0

        def __iter__(self):
            for r in self.rows:
                self.n += 1
                yield r
    w = Wrapper(list(g["perms64"]))
    res = ls_spa(*data_of(g), perms=w, batch_size=16, tolerance=0.0, _engine=OracleEngine())
    np.testing.assert_allclose(res.attribution, g["drv_anti1_attribution"], **TOL)
    assert w.n == 64


def test_seed_path_trigger_indices_and_stream(golden):
    """perms=None, p >= 9: checks at i = 16, 32, 39 (= max_samples - 1) and a trailing one at 40;
    the first 16 orderings are drawn before the estimator touches the shared generator."""
    g = golden("p12")
    eng = OracleEngine()
    res = ls_spa(*data_of(g), max_samples=40, batch_size=16, tolerance=0.0, seed=3,
                 return_attribution_history=True, _engine=eng)
    assert eng.calls == [16, 16, 7, 1]
    assert len(res.error_history) == 4
    np.testing.assert_allclose(res.attribution_history[:16], g["seedpath_attribution_history"][:16], **TOL)
    assert res.attribution_history.shape == (40, 12)
    if np.allclose(res.attribution_history[16:], g["seedpath_attribution_history"][16:], atol=1e-12):
        np.testing.assert_allclose(res.attribution, g["seedpath_attribution"], **TOL)


def test_tolerance_stops_at_batch_boundary(golden):
    g = golden("p12")
    eng = OracleEngine()
    res = ls_spa(*data_of(g), perms=g["perms64"], batch_size=16, tolerance=1e9, _engine=eng)
    assert eng.calls == [16] and len(res.error_history) == 1


def test_single_sample_gives_nan_error(golden):
    g = golden("p12")
    with np.errstate(all="ignore"):
        try:
            res = ls_spa(*data_of(g), perms=g["perms64"][:1], _engine=OracleEngine())
        except np.linalg.LinAlgError:
            return           # numpy builds that refuse to factor a NaN matrix: same as the reference there
    assert np.isnan(res.overall_error)


def test_m_less_than_p_and_float32(golden):
    g = golden("edge")
    res = ls_spa(*data_of(g), perms=g["perms"], batch_size=8, tolerance=0.0, _engine=OracleEngine())
    np.testing.assert_allclose(res.attribution, g["mltp_attribution"], **TOL)
    np.testing.assert_allclose(res.theta, g["mltp_theta"], **TOL)
    g12 = golden("p12")
    d32 = [a.astype(np.float32) for a in data_of(g12)]
    res = ls_spa(*d32, perms=g["perms"], batch_size=8, tolerance=0.0, _engine=OracleEngine())
    assert res.attribution.dtype == np.float64
    # the reference keeps the test side in float32; this build promotes both sides
    np.testing.assert_allclose(res.attribution, g["f32_attribution"], rtol=0, atol=2e-5)


def test_dataframe_inputs(golden):
    import pandas as pd
    g = golden("p12")
    Xa, Xe, ya, ye = data_of(g)
    res = ls_spa(pd.DataFrame(Xa), pd.DataFrame(Xe), pd.Series(ya), pd.Series(ye), perms=g["perms64"][:16],
                 batch_size=16, tolerance=0.0, _engine=OracleEngine())
    want = ls_spa(Xa, Xe, ya, ye, perms=g["perms64"][:16], batch_size=16, tolerance=0.0, _engine=OracleEngine())
    np.testing.assert_array_equal(res.attribution, want.attribution)


# ---------------------------------------------------------------- README dialect
def test_readme_keywords(golden):
    g = golden("p12")
    d = data_of(g)
    eng = OracleEngine()
    res = ls_spa(*d, method="argsort", batch_size=8, num_batches=3, tolerance=0.0, seed=5,
                 return_history=True, _engine=eng)
    assert eng.calls == [8, 8, 7, 1] and res.attribution_history.shape == (24, 12)
    from scipy.stats.qmc import Sobol
    want = ls_spa(*d, perms=np.argsort(Sobol(12, seed=5).random(32), axis=1)[:24], batch_size=8,
                  tolerance=0.0, _engine=OracleEngine())
    np.testing.assert_allclose(res.attribution, want.attribution, **TOL)
    res = ls_spa(*d, method="permutohedron", batch_size=8, num_batches=2, tolerance=0.0, seed=5,
                 _engine=OracleEngine())
    want = ls_spa(*d, perms=golden("samplers_p12")["permutohedron"][:16], batch_size=8, tolerance=0.0,
                  _engine=OracleEngine())
    np.testing.assert_allclose(res.attribution, want.attribution, **TOL)
    g4 = golden("exact_p4")
    res = ls_spa(*data_of(g4), method="exact", _engine=OracleEngine())
    np.testing.assert_allclose(res.attribution, g4["attribution"], **TOL)
    with pytest.raises(ValueError):
        ls_spa(*d, method="sobol", _engine=OracleEngine())
    with pytest.raises(ValueError):
        ls_spa(*d, method="argsort", perms=g["perms64"], _engine=OracleEngine())


# ---------------------------------------------------------------- C ABI export list
def test_library_exports_every_declared_symbol():
    from ls_spa import _native
    header = open(os.path.join(ROOT, "include", "lsspa.h")).read()
    declared = set(re.findall(r"\b(lsspa_[a-z0-9_]+)\s*\(", header)) - {"lsspa_ctx"}
    assert declared == set(_native.SIGNATURES), declared ^ set(_native.SIGNATURES)
    lib = ctypes.CDLL(_native.library_path())
    for name in declared:
        assert hasattr(lib, name), name
    assert _native.load().lsspa_abi_version() == 1


def test_batch_validation_fast_form_agrees_with_the_stamp_loop():
    """Every batch launch checks on the host that its rows are permutations (csrc/host_perms.cpp; the reference's
    orderings come from its own samplers, ls_spa/ls_spa.py:375-456, and are never checked).  The AVX2 form (a row as
    a 128-bit set, 8 <= p <= 128) and the stamp loop give the same verdict on permutations and on rows with a
    repeated entry, an entry just out of range, a negative one and one far out of range -- in any row, any column."""
    from ls_spa import _native
    lib = _native.load()
    rng = np.random.default_rng(17)

    def verdicts(q):
        q = np.ascontiguousarray(q, dtype=np.int32)
        return (lib.lsspa_debug_check_perms(_native.iptr(q), q.shape[0], q.shape[1], 0),
                lib.lsspa_debug_check_perms(_native.iptr(q), q.shape[0], q.shape[1], 1))

    for p in (1, 2, 7, 8, 9, 16, 23, 63, 64, 65, 100, 127, 128, 129, 300):
        B = 37
        perms = np.array([rng.permutation(p) for _ in range(B)], dtype=np.int32)
        assert verdicts(perms) == (1, 1), p
        if p == 1:
            assert verdicts(np.array([[1]])) == (0, 0)
            continue
        for t in range(60):
            q = perms.copy()
            s, j = rng.integers(B), rng.integers(p)
            q[s, j] = (q[s, (j + 1) % p], p, -1, p + 64 + int(rng.integers(1000)), 2 ** 31 - 1, -2 ** 31)[t % 6]
            assert verdicts(q) == (0, 0), (p, t, s, j)


def test_native_row_argsort_is_numpys():
    """The QMC sources' row argsort goes through the library's native threads (lsspa_host_argsort_rows; the reference:
    np.argsort of Sobol' points / projected normals, experiments/ground_truth_medium.py:56-71).  A row with all keys
    different has one argsort; rows with equal keys or a NaN are marked and sorted by numpy -- so every row is numpy's,
    whatever numpy does with ties."""
    import warnings
    from scipy.stats.qmc import Sobol
    from ls_spa import _native
    lib = _native.load()
    rng = np.random.default_rng(8)
    for p, n in ((2, 70), (12, 64), (100, 1024), (101, 333), (1000, 130)):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            keys = Sobol(p, seed=p).random(n) if p % 2 == 0 else rng.standard_normal((n, p))
        np.testing.assert_array_equal(S._argsort_rows(keys), np.argsort(keys, axis=1))
        spoilt = keys.copy()
        spoilt[3, 0] = spoilt[3, p - 1]          # equal keys at the two ends of a row
        spoilt[9, :] = 0.5                       # a constant row
        spoilt[17, p // 2] = np.nan
        spoilt[21, 0] = np.inf
        got = S._argsort_rows(spoilt)
        assert got.dtype == np.int32
        np.testing.assert_array_equal(got, np.argsort(spoilt, axis=1))
        out, redo, n_redo = np.empty((n, p), np.int32), np.empty(n, np.uint8), ctypes.c_int64()
        assert lib.lsspa_host_argsort_rows(_native.dptr(spoilt), n, p, _native.iptr(out),
                                           redo.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), 3, ctypes.byref(n_redo)) == 0
        marked = set(np.nonzero(redo)[0])
        assert {3, 9, 17} <= marked and n_redo.value == len(marked)          # (the row with +inf may sort or be left)
        for r in marked - {3, 9, 17, 21}:          # Sobol' points carry 30 bits: a row of 1000 has equal keys now and then
            assert len(np.unique(spoilt[r])) < p
    # below 64 rows the blocks stay with numpy
    few = rng.random((10, 30))
    np.testing.assert_array_equal(S._argsort_rows(few), np.argsort(few, axis=1))


def test_native_argsort_source_is_the_python_one():
    """method='argsort' draws its orderings on a thread of the library (lsspa_sampler_*: Sobol' points by SciPy's
    recurrence, read off a SciPy engine and checked against it, and their row argsort).  What it hands out is what the
    Python source hands out -- the reference's np.argsort(Sobol(p, seed).random(n), axis=1),
    experiments/ground_truth_medium.py:56-60 -- for one process and for every rank of several, whatever the sizes asked
    for, past the end of the source, after a skip, and in rows whose points have equal coordinates (30-bit points: a few
    in a thousand rows at p = 1000), which come back marked and are sorted by numpy."""
    for p, n, block in ((12, 300, 64), (100, 3000, 1024), (1000, 2500, 256)):
        full = S.ArgsortSource(p, 5, 10 ** 6).take(n)
        ties = sum(len(np.unique(r)) < p for r in S._DirectSobol.make(__import__("scipy.stats").stats.qmc.Sobol(p, seed=5))
                   .points(np.arange(n)))
        if p == 1000:
            assert ties > 0          # the marked-row path is exercised
        for world in (1, 2, 3, 8):
            for rank in range(world):
                src = S.NativeArgsortSource.make(p, 5, n, block=block, rank=rank, world=world)
                assert src is not None and src.usable()
                pos, rows, idx = 0, [], []
                for cnt in (1, 10, 100, 7, 120, 50, 10 ** 4):
                    m, own = src.take_share(cnt, pos, rank, world)
                    idx.append(np.arange(pos + (rank - pos) % world, pos + m, world))
                    rows.append(own)
                    pos += m
                assert pos == n and src.take_share(5, pos, rank, world)[0] == 0
                np.testing.assert_array_equal(np.concatenate(rows), full[np.concatenate(idx)])
                src.close()
                src.close()          # twice is once
    src = S.NativeArgsortSource.make(23, 7, 500, block=64)
    src.skip(123)
    np.testing.assert_array_equal(src.take(77), S.ArgsortSource(23, 7, 500).take(200)[123:])
    with pytest.raises(ValueError, match="out of turn"):
        src.take_share(4, 3, 0, 1)
    src.close()
    # the public call takes it for method='argsort' and gives the Python source's run to the last bit
    g = np.load(os.path.join(ROOT, "tests", "golden", "p12.npz"))
    d = [g[k] for k in ("X_train", "X_test", "y_train", "y_test")]
    kw = dict(method="argsort", seed=5, max_samples=96, batch_size=16, tolerance=0.0)
    from ls_spa import _driver
    made = []
    real = S.NativeArgsortSource.make.__func__

    def spy(cls, *a, **k):
        made.append(real(cls, *a, **k))
        return made[-1]
    S.NativeArgsortSource.make = classmethod(spy)
    try:
        nat = ls_spa(*d, _engine=OracleEngine(), **kw)
        assert len(made) == 1 and made[0] is not None and made[0]._fallback is None
        S.NativeArgsortSource.make = classmethod(lambda cls, *a, **k: None)
        py = ls_spa(*d, _engine=OracleEngine(), **kw)
    finally:
        S.NativeArgsortSource.make = classmethod(real)
    np.testing.assert_array_equal(nat.attribution, py.attribution)
    np.testing.assert_array_equal(nat.error_history, py.error_history)


def test_product_never_imports_oracle():
    pkg_dir = os.path.join(ROOT, "ls-spa_amd")
    for base, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(base, f)).read()
                assert "lsspa_oracle" not in text and "oracle_engine" not in text, f


def test_unique_id_rendezvous_three_ranks():
    """exchange_unique_id: rank 0 makes the id once and serves it, the others fetch it (started BEFORE the server is
    up, so they retry); a stray connection that does not speak the protocol is ignored."""
    import socket
    import threading
    from ls_spa._rccl import ID_BYTES, exchange_unique_id
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    made = []

    def make_id():
        made.append(1)
        return bytes(range(ID_BYTES))

    got = {}

    def run(rank, delay):
        import time
        time.sleep(delay)
        got[rank] = exchange_unique_id(rank, 3, "127.0.0.1", port, make_id, timeout=30.0)

    def stray():
        import time
        time.sleep(0.25)
        try:
            with socket.create_connection(("127.0.0.1", port), timeout=5.0) as c:
                c.sendall(b"GET / HTTP/1.0\r\n\r\n")
        except OSError:
            pass

    threads = [threading.Thread(target=run, args=(1, 0.0)), threading.Thread(target=run, args=(2, 0.4)),
               threading.Thread(target=stray), threading.Thread(target=run, args=(0, 0.2))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    assert len(made) == 1
    assert got[0] == got[1] == got[2] == bytes(range(ID_BYTES))
    assert exchange_unique_id(0, 1, "127.0.0.1", port, make_id) == bytes(range(ID_BYTES))   # world of one: no socket


def test_gather_ints_by_sum():
    from ls_spa._driver import gather_ints_by_sum

    class Two:
        world = 2

        def __init__(self, rank, other):
            self.rank, self.other = rank, other

        def sum_ints(self, flat):
            return [a + b for a, b in zip(flat, self.other)]

    # rank 1 contributes [7, -1] in its slot, rank 0 [3, 5]
    assert gather_ints_by_sum(Two(0, [0, 0, 7, -1]), [3, 5]) == [[3, 5], [7, -1]]


def test_lookahead_keeps_the_reference_order(golden):
    """lookahead=k: the orderings of k chunks of a QMC sampler are launched as one batch, accumulated and checked
    chunk by chunk, and dropped beyond a stop.  Same attribution, error history and stop index as lookahead=1; a
    user's iterable and the shared generator are never drawn ahead."""
    from oracle_engine import OracleEngine
    g = golden("p12")
    d = [g[k] for k in ("X_train", "X_test", "y_train", "y_test")]
    for method in ("argsort", "permutohedron"):
        kw = dict(method=method, seed=5, max_samples=96, batch_size=16, tolerance=0.0)
        e1, e3 = OracleEngine(), OracleEngine()
        one = ls_spa(*d, _engine=e1, **kw)
        three = ls_spa(*d, _engine=e3, lookahead=3, **kw)
        np.testing.assert_array_equal(three.attribution, one.attribution)
        np.testing.assert_array_equal(three.error_history, one.error_history)
        assert e3.calls == e1.calls == [16, 16, 16, 16, 16, 15, 1]     # checks at 16 .. 80, 95, 96
        assert e3.launched == 3 and e3.discarded == 0                   # 7 chunks in groups of 3, 3, 1
    # a tolerance that stops the run at the second check: the third chunk of the group (and nothing else) is dropped
    first = ls_spa(*d, method="argsort", seed=5, max_samples=96, batch_size=16, tolerance=0.0, _engine=OracleEngine())
    tol = float(first.error_history[1]) * 1.0000001
    assert first.error_history[0] > tol
    e1, e3 = OracleEngine(), OracleEngine()
    kw = dict(method="argsort", seed=5, max_samples=96, batch_size=16, tolerance=tol)
    one = ls_spa(*d, _engine=e1, _defer=0, **kw)
    three = ls_spa(*d, _engine=e3, lookahead=3, _defer=0, **kw)
    assert len(one.error_history) == len(three.error_history) == 2
    np.testing.assert_array_equal(three.attribution, one.attribution)
    assert e3.discarded == 1 and e3.launched == 1 and sum(e3.calls) == sum(e1.calls) == 32
    # the group ends exactly at a check.  Host-side estimator: the next group is launched before the rule is
    # evaluated, then dropped.  Device-side estimator with every check waited for (_defer=0): the next group is
    # launched after the decision -- nothing wasted.
    ref_hist = ls_spa(*d, method="argsort", seed=5, max_samples=96, batch_size=16, tolerance=0.0,
                      error_estimator="reference", _engine=OracleEngine()).error_history
    assert ref_hist[0] > ref_hist[1]
    e2 = OracleEngine()
    two = ls_spa(*d, _engine=e2, lookahead=2, error_estimator="reference",
                 **dict(kw, tolerance=float(ref_hist[1]) * 1.0000001))
    np.testing.assert_array_equal(two.attribution, one.attribution)
    assert e2.launched == 2 and e2.discarded == 1 and sum(e2.calls) == 32
    e2 = OracleEngine()
    two = ls_spa(*d, _engine=e2, lookahead=2, _defer=0, **kw)
    np.testing.assert_array_equal(two.attribution, one.attribution)
    assert e2.launched == 1 and e2.discarded == 0 and sum(e2.calls) == 32
    # the default of the QMC methods (round 5): as many checks as a group has chunks may be outstanding -- the decision
    # of check 2 is taken when check 4 has been enqueued: same results (the stopping check's own copy of the running
    # mean), two chunks more evaluated
    e2 = OracleEngine()
    two = ls_spa(*d, _engine=e2, lookahead=2, **kw)
    np.testing.assert_array_equal(two.attribution, one.attribution)
    np.testing.assert_array_equal(two.error_history, one.error_history)
    assert e2.launched == 2 and e2.discarded == 0 and sum(e2.calls) == 64
    # sources somebody else reads are not drawn ahead
    for kw in (dict(perms=iter(g["perms64"]), batch_size=16, tolerance=0.0),
               dict(max_samples=48, batch_size=16, tolerance=0.0, seed=3)):
        e4 = OracleEngine()
        ls_spa(*d, _engine=e4, lookahead=4, **kw)
        assert e4.launched == 0 and e4.discarded == 0
    with pytest.raises(ValueError):
        ls_spa(*d, lookahead=0, _engine=OracleEngine())
    # 'auto': small problems (p <= 127, one workgroup per ordering) go up to sixteen chunks to a launch up to 2048
    # samples -- in groups of 1, 2, 4, ... chunks --, chunks of 1024 samples and more one at a time
    ea, eb = OracleEngine(), OracleEngine()
    auto = ls_spa(*d, method="argsort", seed=5, max_samples=96, batch_size=16, tolerance=0.0, lookahead="auto", _engine=ea)
    np.testing.assert_array_equal(auto.attribution, first.attribution)
    assert ea.launched == 3                       # 7 chunks in groups of 1, 2 and 4: the automatic groups grow
    ls_spa(*d, method="argsort", seed=5, max_samples=2048, batch_size=1024, tolerance=0.0, lookahead="auto", _engine=eb)
    assert eb.launched == 0


def test_lookahead_with_history_and_checkpoint(golden, tmp_path):
    """lookahead composes with the other driver features: the attribution history is the one of lookahead=1, and a
    run interrupted inside a group resumes (the orderings launched ahead are drawn again) to the same numbers."""
    from oracle_engine import OracleEngine
    g = golden("p12")
    d = [g[k] for k in ("X_train", "X_test", "y_train", "y_test")]
    kw = dict(method="permutohedron", seed=9, max_samples=80, batch_size=16, tolerance=0.0, error_estimator="lowrank")
    one = ls_spa(*d, return_attribution_history=True, _engine=OracleEngine(), **kw)
    three = ls_spa(*d, return_attribution_history=True, lookahead=3, _engine=OracleEngine(), **kw)
    np.testing.assert_array_equal(three.attribution_history, one.attribution_history)
    np.testing.assert_array_equal(three.error_history, one.error_history)

    class Dies(OracleEngine):
        def collect_batch(self, *a, **k):
            if len(self.calls) == 4:          # second chunk of the second group of three
                raise KeyboardInterrupt
            return super().collect_batch(*a, **k)

    ck = str(tmp_path / "la.npz")
    with pytest.raises(KeyboardInterrupt):
        ls_spa(*d, checkpoint=ck, lookahead=3, _engine=Dies(), **kw)
    with np.load(ck) as z:
        assert int(z["n"]) == 64
    resumed = ls_spa(*d, checkpoint=ck, lookahead=3, _engine=OracleEngine(), **kw)
    np.testing.assert_allclose(resumed.attribution, one.attribution, rtol=0, atol=1e-14)
    np.testing.assert_allclose(resumed.error_history, one.error_history, rtol=1e-9)


def test_bench_helpers_follow_the_baseline_recipe():
    """bench.py's data generator is BASELINE.md section 3's (the same stream as the oracle's / the product's Gaussian
    workload, whatever the row blocking), its config labels are BASELINE.json's, and the algorithmic flop counts
    are SURVEY.md 8d's."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    got = bench.baseline_data(7, 50, "f64")
    import lsspa_oracle as O
    want = O.gaussian_workload(7, 50, 50, seed=0)
    for a, b in zip(got, want):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-13)      # y = X theta + noise: same draws, same order
    np.testing.assert_array_equal(got[0], want[0])
    f32 = bench.baseline_data(7, 50, "f32")
    assert all(a.dtype == np.float32 for a in f32)
    np.testing.assert_array_equal(f32[0], want[0].astype(np.float32))
    assert bench.config_label(100, 10000, "f64") == "C2" and bench.config_label(1000, 100000, "f64") == "C3"
    assert bench.config_label(5000, 200000, "f32") == "C5" and bench.config_label(1000, 100000, "f32") == "custom"
    p, n_ord = 1000, 256
    assert bench.algorithmic_flops("strip", p, n_ord, True, 1) == pytest.approx(p ** 3 / 3 * n_ord)
    assert bench.algorithmic_flops("chol_panel", p, n_ord, True, 7) == pytest.approx(p ** 3 / 3 * 2 * n_ord / 7)
    assert bench.algorithmic_flops("small_p", 100, n_ord, True, 1) == pytest.approx(100 ** 3 * n_ord)


def _bench_path():
    import os
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: the parent starts two ranks of itself with a
    consistent rendezvous environment (before it imports torch or touches HIP: the ranks stop at a stub that reports
    what they were given), relays rank 0's single line and exits 0; if a rank fails it exits non-zero."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(LSSPA_BENCH_STUB="1", LSSPA_BENCH_STUB_DIR=str(tmp_path))
    r = subprocess.run([sys.executable, _bench_path(), "--gpus", "2", "--steps", "3", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"stub": True, "world": 2}
    recs = [json.load(open(tmp_path / f"bench_stub_rank{k}.json")) for k in range(2)]
    assert [rec["RANK"] for rec in recs] == ["0", "1"] and [rec["LOCAL_RANK"] for rec in recs] == ["0", "1"]
    assert all(rec["WORLD_SIZE"] == "2" and rec["MASTER_ADDR"] == "127.0.0.1" and rec["gpus"] == 2 for rec in recs)
    assert recs[0]["MASTER_PORT"] == recs[1]["MASTER_PORT"] and 1024 < int(recs[0]["MASTER_PORT"]) < 65536 - 64
    assert not any(rec["torch_imported"] for rec in recs)
    # a failing rank ends the job with a non-zero status (and no JSON line is invented)
    env["LSSPA_BENCH_STUB_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, _bench_path(), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode != 0 and "rank 1 exited" in r.stderr
    # one GPU: no launcher (WORLD_SIZE unset, --gpus 1 runs in place: the stub sees no RANK)
    env.pop("LSSPA_BENCH_STUB_FAIL_RANK")
    r = subprocess.run([sys.executable, _bench_path(), "--gpus", "1"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0 and json.load(open(tmp_path / "bench_stub_rankNone.json"))["WORLD_SIZE"] is None


def test_checkpoint_ranks_agree_before_anyone_raises(golden, tmp_path):
    """A rank whose own file is foreign or unreadable reports it THROUGH the collective and then every rank raises;
    nothing raises before the exchange (the other ranks would hang in it).  A foreign .prev next to a valid file
    is ignored."""
    from ls_spa import _driver as D

    class TwoRanks:
        world = 2

        def __init__(self, rank, other_row):
            self.rank, self.other_row, self.calls = rank, other_row, 0

        def gather_ints(self, values):
            self.calls += 1
            rows = [None, None]
            rows[self.rank] = [int(v) for v in values]
            rows[1 - self.rank] = list(self.other_row)
            return rows

    ident = {"p": 3, "seed": 1}
    path = str(tmp_path / "ck.npz")

    def write(file, n, seed=1):
        np.savez(file, version=D._CKPT_VERSION, n=n, p=3, seed=seed)
        os.replace(file + ".npz", file) if os.path.exists(file + ".npz") else None

    mine = D._ckpt_path(path, TwoRanks(0, []))
    write(mine, 32)
    write(mine + ".prev", 16, seed=99)          # a leftover of another run: ignored, the target is valid
    c = TwoRanks(0, [32, 16, 0])
    st = D._load_checkpoint(path, c, ident)
    assert int(st["n"]) == 32 and c.calls == 1
    # this rank's target belongs to another run: the error surfaces only after the exchange, and names the rank
    write(mine, 32, seed=7)
    c = TwoRanks(0, [32, -1, 0])
    with pytest.raises(ValueError, match=r"rank\(s\) \[0\]"):
        D._load_checkpoint(path, c, ident)
    assert c.calls == 1
    # the OTHER rank's file is bad: this rank raises the same error although its own file is fine
    write(mine, 32)
    c = TwoRanks(0, [-1, -1, 1])
    with pytest.raises(ValueError, match=r"rank\(s\) \[1\]"):
        D._load_checkpoint(path, c, ident)
    # a corrupt archive is a problem like any other, not an exception before the collective
    with open(mine, "wb") as fh:
        fh.write(b"not a zip archive")
    c = TwoRanks(0, [32, -1, 0])
    with pytest.raises(ValueError, match=r"rank\(s\) \[0\]"):
        D._load_checkpoint(path, c, ident)
    assert c.calls == 1


def test_checkpoint_keeps_the_attribution_history(golden, tmp_path):
    """checkpoint= composes with return_attribution_history: the resumed run's history is the uninterrupted one's."""
    class Dies(OracleEngine):
        def run_batch(self, *a, **k):
            if len(self.calls) == 2:
                raise KeyboardInterrupt
            return super().run_batch(*a, **k)

    g = golden("p12")
    d = [g[k] for k in ("X_train", "X_test", "y_train", "y_test")]
    kw = dict(batch_size=16, tolerance=0.0, seed=4, method="argsort", max_samples=64, return_attribution_history=True,
              lookahead=1)
    full = ls_spa(*d, _engine=OracleEngine(), **kw)
    ck = str(tmp_path / "h.npz")
    with pytest.raises(KeyboardInterrupt):
        ls_spa(*d, checkpoint=ck, _engine=Dies(), **kw)
    resumed = ls_spa(*d, checkpoint=ck, _engine=OracleEngine(), **kw)
    assert resumed.attribution_history.shape == (64, 12)
    np.testing.assert_array_equal(resumed.attribution_history, full.attribution_history)
    np.testing.assert_allclose(resumed.attribution, full.attribution, rtol=0, atol=1e-14)
    with pytest.raises(ValueError, match="history"):      # a state file written without the history cannot supply it
        ls_spa(*d, checkpoint=ck, _engine=OracleEngine(), **dict(kw, return_attribution_history=False))
