"""world_size-2 test of the sharded driver on CPU (gloo), with the oracle-backed test double in
place of the HIP engine: the all-reduce of the pending moments must reproduce the single-process
result (attribution, covariance-driven error history, attribution history)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    for sub in ("ls-spa_amd", "oracle", "tests"):
        sys.path.insert(0, os.path.join(ROOT, sub))
    import torch.distributed as dist
    from ls_spa import ls_spa
    from ls_spa._dist import TorchComm
    from oracle_engine import OracleEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, "tests", "golden", "p12.npz"))
    d = [g[k] for k in ("X_train", "X_test", "y_train", "y_test")]
    eng = OracleEngine()
    res = ls_spa(*d, perms=g["perms64"][:50], batch_size=16, tolerance=0.0,
                 return_attribution_history=True, _engine=eng, comm=TorchComm())
    # thin-form estimator with the samples spread over the ranks: partial draws + one all-reduce
    eng2 = OracleEngine()
    dev = ls_spa(*d, perms=g["perms64"][:50], batch_size=16, tolerance=0.0, error_estimator="device",
                 _engine=eng2, comm=TorchComm())
    # row-sharded reduction: this rank holds every second row of both sets
    eng3 = OracleEngine()
    shard = ls_spa(d[0][rank::world], d[1][rank::world], d[2][rank::world], d[3][rank::world],
                   perms=g["perms64"][:20], batch_size=16, tolerance=0.0, row_sharded=True,
                   _engine=eng3, comm=TorchComm())
    # only the training rows sharded, every rank holds all test rows
    eng4 = OracleEngine()
    shard_tr = ls_spa(d[0][rank::world], d[1], d[2][rank::world], d[3], perms=g["perms64"][:20], batch_size=16,
                      tolerance=0.0, row_sharded="train", _engine=eng4, comm=TorchComm())
    # lookahead: three chunks of a QMC sampler launched as one batch per rank, accumulated / all-reduced / checked
    # chunk by chunk; a tolerance that stops the run inside a group (the rest of the group is dropped on both ranks)
    eng5 = OracleEngine()
    la = ls_spa(*d, method="argsort", seed=5, max_samples=96, batch_size=16, tolerance=0.0, lookahead=3,
                error_estimator="lowrank", _engine=eng5, comm=TorchComm())
    eng6 = OracleEngine()
    la_stop = ls_spa(*d, method="argsort", seed=5, max_samples=96, batch_size=16, tolerance=float(la.error_history[1]) * 1.0000001,
                     lookahead=3, error_estimator="lowrank", _engine=eng6, comm=TorchComm())
    np.savez(os.path.join(out_dir, f"l{rank}.npz"), attribution=la.attribution, err=la.error_history,
             calls=np.array(eng5.calls), launched=eng5.launched, stop_attr=la_stop.attribution,
             stop_checks=len(la_stop.error_history), stop_discarded=eng6.discarded, stop_calls=np.array(eng6.calls))
    # checkpoint / resume with two ranks: one state file per rank, killed in the third batch, resumed
    class Dies(OracleEngine):
        def run_batch(self, *a, **k):
            if len(self.calls) == 2:
                raise KeyboardInterrupt
            return super().run_batch(*a, **k)
    ck = os.path.join(out_dir, "state.npz")
    kw = dict(perms=None, method="argsort", seed=3, max_samples=80, batch_size=16, tolerance=0.0,
              error_estimator="device", lookahead=1)
    try:
        ls_spa(*d, checkpoint=ck, _engine=Dies(), comm=TorchComm(), **kw)
    except KeyboardInterrupt:
        pass
    # ... and between the two ranks' renames of the second check: rank 1 is left one generation behind (its newest
    # file is the older state); the ranks must agree on the state both hold (n = 16) instead of resuming apart
    dist.barrier()
    if rank == 1:
        os.replace(ck + ".rank1.prev", ck + ".rank1")
    dist.barrier()
    resumed = ls_spa(*d, checkpoint=ck, _engine=OracleEngine(), comm=TorchComm(), **kw)
    # a rank whose files are gone next to a rank that has them: every rank raises, nobody runs ahead alone
    ck2 = os.path.join(out_dir, "state2.npz")
    try:
        ls_spa(*d, checkpoint=ck2, _engine=Dies(), comm=TorchComm(), **kw)
    except KeyboardInterrupt:
        pass
    dist.barrier()
    if rank == 1:
        os.remove(ck2 + ".rank1")
        os.remove(ck2 + ".rank1.prev")
    dist.barrier()
    try:
        ls_spa(*d, checkpoint=ck2, _engine=OracleEngine(), comm=TorchComm(), **kw)
        lonely = "ran"
    except ValueError as exc:
        lonely = "refused" if "no common state" in str(exc) else str(exc)
    np.savez(os.path.join(out_dir, f"s{rank}.npz"), attribution=shard.attribution, theta=shard.theta,
             r2=shard.r_squared, attribution_tr=shard_tr.attribution, res_attr=resumed.attribution,
             res_err=resumed.error_history, ck_exists=os.path.exists(ck + f".rank{rank}"), lonely=lonely)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), attribution=res.attribution,
             history=res.attribution_history, err=res.error_history, calls=np.array(eng.calls),
             dev_err=dev.error_history, dev_feat=dev.attribution_errors, dev_checks=eng2.enqueued)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_match_single_process(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ls_spa import ls_spa
    from oracle_engine import OracleEngine
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g = np.load(os.path.join(ROOT, "tests", "golden", "p12.npz"))
    d = [g[k] for k in ("X_train", "X_test", "y_train", "y_test")]
    single = ls_spa(*d, perms=g["perms64"][:50], batch_size=16, tolerance=0.0,
                    return_attribution_history=True, _engine=OracleEngine())
    r0 = np.load(tmp_path / "r0.npz")
    r1 = np.load(tmp_path / "r1.npz")
    for r in (r0, r1):
        np.testing.assert_allclose(r["attribution"], single.attribution, rtol=0, atol=1e-13)
        np.testing.assert_allclose(r["history"], single.attribution_history, rtol=0, atol=1e-13)
        assert len(r["err"]) == len(single.error_history) == 4
    # both ranks take identical decisions and split every chunk
    np.testing.assert_array_equal(r0["err"], r1["err"])
    assert list(r0["calls"]) == [8, 8, 8, 1] and list(r1["calls"]) == [8, 8, 8, 1]
    # device-form estimator (running form): each rank folded half of the lift vectors into its own D = Xi L, s = Xi 1;
    # the normals are a function of the GLOBAL sample number, so the all-reduced draws -- hence every number -- are the
    # one-process run's; against the host low-rank form (other normals, same distribution) the pin is statistical
    one = ls_spa(*d, perms=g["perms64"][:50], batch_size=16, tolerance=0.0, error_estimator="device",
                 _engine=OracleEngine())
    low = ls_spa(*d, perms=g["perms64"][:50], batch_size=16, tolerance=0.0, error_estimator="lowrank",
                 _engine=OracleEngine())
    for r in (r0, r1):
        np.testing.assert_allclose(r["dev_err"], one.error_history, rtol=1e-10)
        np.testing.assert_allclose(r["dev_feat"], one.attribution_errors, rtol=1e-10)
        np.testing.assert_allclose(r["dev_err"], low.error_history, rtol=0.15)
        assert int(r["dev_checks"]) == 4
    # row-sharded reduction: same attribution as the run on the stacked rows
    full = ls_spa(*d, perms=g["perms64"][:20], batch_size=16, tolerance=0.0, _engine=OracleEngine())
    for rk in (0, 1):
        s = np.load(tmp_path / f"s{rk}.npz")
        np.testing.assert_allclose(s["attribution"], full.attribution, rtol=0, atol=1e-11)
        np.testing.assert_allclose(s["theta"], full.theta, rtol=1e-9)
        assert abs(float(s["r2"]) - full.r_squared) < 1e-11
        np.testing.assert_allclose(s["attribution_tr"], full.attribution, rtol=0, atol=1e-11)
        assert bool(s["ck_exists"])
        assert str(s["lonely"]) == "refused"
    # two-rank resume == uninterrupted single-process run with the same sampler and estimator
    straight = ls_spa(*d, method="argsort", seed=3, max_samples=80, batch_size=16, tolerance=0.0,
                      error_estimator="device", lookahead=1, _engine=OracleEngine())
    for rk in (0, 1):
        s = np.load(tmp_path / f"s{rk}.npz")
        np.testing.assert_allclose(s["res_attr"], straight.attribution, rtol=0, atol=1e-13)
        np.testing.assert_allclose(s["res_err"], straight.error_history, rtol=1e-9)
    # lookahead over two ranks == single process without it
    la1 = ls_spa(*d, method="argsort", seed=5, max_samples=96, batch_size=16, tolerance=0.0,
                 error_estimator="lowrank", _engine=OracleEngine())
    la1_stop = ls_spa(*d, method="argsort", seed=5, max_samples=96, batch_size=16, error_estimator="lowrank",
                      tolerance=float(la1.error_history[1]) * 1.0000001, _engine=OracleEngine())
    for rk in (0, 1):
        l = np.load(tmp_path / f"l{rk}.npz")
        np.testing.assert_allclose(l["attribution"], la1.attribution, rtol=0, atol=1e-13)
        np.testing.assert_allclose(l["err"], la1.error_history, rtol=1e-9)
        # chunks of 16, 16, 16, 16, 16, 15, 1 samples; sample number g of the run belongs to rank g mod 2 (round 5: dealt by
        # the sample's number, not by its place in the chunk), so the last sample, number 95, is rank 1's (a rank with
        # nothing in a chunk makes no call, and no launch for a group it has nothing in)
        assert list(l["calls"]) == ([8, 8, 8, 8, 8, 8] if rk == 0 else [8, 8, 8, 8, 8, 7, 1]) and int(l["launched"]) == 2 + rk
        np.testing.assert_allclose(l["stop_attr"], la1_stop.attribution, rtol=0, atol=1e-13)
        assert int(l["stop_checks"]) == len(la1_stop.error_history) == 2
        assert int(l["stop_discarded"]) == 1 and list(l["stop_calls"]) == [8, 8]
    # and the reference itself agrees (fixture made from it on the first 48... full 64 run differs)
