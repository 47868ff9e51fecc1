"""RCCL behind the C ABI (lsspa_comm_* / lsspa_*_allreduce, ls_spa._rccl.NativeComm), exercised with ONE rank on
one GPU: two ranks cannot share a GPU under RCCL, so what runs here is the real library path -- dlopen of librccl,
ncclCommInitRank, ncclAllReduce / ncclAllGather on the engine's stream, the upper-triangle packing -- in a world of
one; the N > 1 dealing / merging logic is covered on CPU by test_dist_gloo.py and the rendezvous by
test_host_logic.py."""
import os
import subprocess
import sys

import numpy as np
import pytest

import lsspa_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stats_allreduce_plain_and_packed(golden):
    from ls_spa._engine import HipEngine
    from ls_spa._rccl import NativeComm
    g = golden("p12")
    d = [g[k] for k in ("X_train", "X_test", "y_train", "y_test")]
    out = {}
    for tag, pack_from in (("plain", 2048), ("packed", 1)):
        eng = HipEngine(0)
        comm = NativeComm(0, 1, force_collective=True)
        try:
            eng.load_data(*d, 0.0)
            comm.bind(eng)
            eng._check(eng._lib.lsspa_debug_pack_from(eng._h, pack_from))
            lifts = []
            for k in range(4):
                lifts.append(eng.run_batch(g["perms64"][16 * k:16 * k + 16], True, want_lifts=True, accumulate=True))
                comm.allreduce_pending(eng)      # ncclAllReduce on the engine's stream, in place
                eng.merge()
            n, mean, cov = eng.stats()
            all_l = np.concatenate(lifts)
            assert n == 64
            np.testing.assert_allclose(mean, all_l.mean(0), rtol=0, atol=1e-13)
            np.testing.assert_allclose(cov, np.cov(all_l, rowvar=False, bias=True), rtol=0, atol=1e-14)
            out[tag] = (mean, cov)
            assert comm.sum_ints([3, 4]) == [3, 4]
            assert comm.gather_ints([5, -1]) == [[5, -1]]
        finally:
            comm.close()
            eng.close()
    # the packed exchange (upper triangle of Q out, mirrored back) is exact
    np.testing.assert_array_equal(out["plain"][0], out["packed"][0])
    np.testing.assert_array_equal(out["plain"][1], out["packed"][1])


def test_driver_through_the_native_communicator(golden):
    from ls_spa import ls_spa
    from ls_spa._rccl import NativeComm
    g = golden("p12")
    d = [g[k] for k in ("X_train", "X_test", "y_train", "y_test")]
    kw = dict(perms=g["perms64"], batch_size=16, tolerance=0.0)
    plain = ls_spa(*d, return_attribution_history=True, **kw)
    via = ls_spa(*d, return_attribution_history=True, comm=NativeComm(0, 1, force_collective=True), **kw)
    np.testing.assert_array_equal(via.attribution, plain.attribution)
    np.testing.assert_array_equal(via.attribution_history, plain.attribution_history)
    np.testing.assert_allclose(via.attribution, g["drv_anti1_attribution"], rtol=0, atol=1e-10)   # the reference's
    # row-sharded reduction + device-side estimator: every collective of the product path in one run
    sharded = ls_spa(*d, row_sharded=True, error_estimator="device", comm=NativeComm(0, 1, force_collective=True), **kw)
    low = ls_spa(*d, error_estimator="device", **kw)
    np.testing.assert_allclose(sharded.attribution, plain.attribution, rtol=0, atol=1e-13)
    np.testing.assert_allclose(sharded.theta, plain.theta, rtol=1e-12)
    np.testing.assert_allclose(sharded.error_history, low.error_history, rtol=1e-9)


def test_multi_gpu_product_path_needs_no_torch(tmp_path):
    """A fresh interpreter runs the sharded driver with NativeComm.from_env() (world of one, collectives forced):
    system HIP runtime, system RCCL, and PyTorch is never imported."""
    code = f"""
import sys, os
import numpy as np
sys.path.insert(0, {os.path.join(ROOT, 'ls-spa_amd')!r})
from ls_spa import ls_spa
from ls_spa._rccl import NativeComm
g = np.load({os.path.join(ROOT, 'tests', 'golden', 'p12.npz')!r})
d = [g[k] for k in ("X_train", "X_test", "y_train", "y_test")]
comm = NativeComm.from_env(force_collective=True)
res = ls_spa(*d, perms=g["perms64"], batch_size=16, tolerance=0.0, device=comm.local_rank, comm=comm)
assert "torch" not in sys.modules, "the product path imported torch"
np.save({str(tmp_path / 'attr.npy')!r}, res.attribution)
print("ok")
"""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29671")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
    g = np.load(os.path.join(ROOT, "tests", "golden", "p12.npz"))
    np.testing.assert_allclose(np.load(tmp_path / "attr.npy"), g["drv_anti1_attribution"], rtol=0, atol=1e-10)


@pytest.mark.parametrize("p,n,m", [(40, 400, 300), (150, 900, 800)])
def test_group_collect_through_the_native_communicator(p, n, m):
    """lsspa_group_collect with a communicator on the context (world of one, collectives forced): per chunk collect,
    lsspa_stats_allreduce + merge, fold into the running estimator, lsspa_error_running_draws + lsspa_error_allreduce +
    quantiles -- the per-chunk tail a rank of an 8-GPU run executes in ONE library call -- against the plain one-GPU
    run (accumulate = 2, no collective, the draws never written): the same orderings, the same sample ids, the same
    normals, so every number agrees to round-off; also when the stop rule fires on the deferred path."""
    from ls_spa import ls_spa
    from ls_spa._engine import HipEngine
    from ls_spa._rccl import NativeComm
    rng = np.random.default_rng(200 + p)
    Xa, Xe = rng.standard_normal((n, p)), rng.standard_normal((m, p))
    th = rng.standard_normal(p)
    ya, ye = Xa @ th + rng.standard_normal(n), Xe @ th + rng.standard_normal(m)
    kw = dict(reg=1e-3, method="argsort", seed=11, batch_size=16, max_samples=96, lookahead=3)

    class Counting(HipEngine):
        groups = 0

        def group_collect(self, *a, **k):
            self.groups += 1
            return super().group_collect(*a, **k)

    plain_e, comm_e = Counting(0), Counting(0)
    cm = NativeComm(0, 1, force_collective=True)     # bound to comm_e by the first call, used again by the second
    try:
        plain = ls_spa(Xa, Xe, ya, ye, tolerance=0.0, _engine=plain_e, **kw)
        via = ls_spa(Xa, Xe, ya, ye, tolerance=0.0, _engine=comm_e, comm=cm, **kw)
        assert plain_e.groups == 3 and comm_e.groups == 3          # 7 chunks in groups of 3, 3, 1: the one-call path, both
        assert len(plain.error_history) == len(via.error_history) == 7
        np.testing.assert_allclose(via.attribution, plain.attribution, rtol=0, atol=1e-13)
        np.testing.assert_allclose(via.error_history, plain.error_history, rtol=1e-9)
        np.testing.assert_allclose(via.attribution_errors, plain.attribution_errors, rtol=1e-9)
        tol = float(plain.error_history[2]) * 1.0000001
        if all(e > tol for e in plain.error_history[:2]):
            a = ls_spa(Xa, Xe, ya, ye, tolerance=tol, _engine=plain_e, **kw)
            b = ls_spa(Xa, Xe, ya, ye, tolerance=tol, _engine=comm_e, comm=cm, **kw)
            assert len(a.error_history) == len(b.error_history) == 3
            np.testing.assert_allclose(b.attribution, a.attribution, rtol=0, atol=1e-13)
    finally:
        cm.close()
        plain_e.close()
        comm_e.close()
