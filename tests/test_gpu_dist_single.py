"""The RCCL path of the sharded driver, exercised on ONE GPU (a one-rank "nccl" group): the pending
statistics buffer must be visible to torch.distributed as a zero-copy device tensor, and the full
driver must give the same answer through TorchComm as without it.  (Two ranks cannot share a GPU
under RCCL; the N > 1 logic is covered on CPU by test_dist_gloo.py.)"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_pending_buffer_is_a_zero_copy_device_tensor_and_allreduce_runs(golden):
    import torch
    import torch.distributed as dist
    from ls_spa import ls_spa
    from ls_spa._dist import TorchComm
    from ls_spa._engine import HipEngine

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        g = golden("p12")
        d = [g[k] for k in ("X_train", "X_test", "y_train", "y_test")]
        eng = HipEngine(0)
        eng.load_data(*d, 0.0)
        eng.reset_stats()
        lifts = eng.run_batch(g["perms64"][:16], True, want_lifts=True, accumulate=True)
        eng.synchronize()
        comm = TorchComm()
        t = comm._as_tensor(eng.pending_buffer(), eng)
        assert t.is_cuda and t.dtype == torch.float64 and t.numel() == 1 + 12 + 144
        host = t.cpu().numpy()
        assert host[0] == 16.0
        np.testing.assert_allclose(host[1:13], lifts.sum(0), rtol=0, atol=1e-13)   # running mean is 0 so far
        dist.all_reduce(t, op=dist.ReduceOp.SUM)          # RCCL on the engine's own buffer
        torch.cuda.synchronize()
        t[0] = 32.0                                        # a write through torch must land in the engine
        t[1:] *= 2.0                                       # = what a second identical rank would add
        torch.cuda.synchronize()
        eng.merge()
        n, mean, cov = eng.stats()
        assert n == 32
        np.testing.assert_allclose(mean, lifts.mean(0), rtol=0, atol=1e-13)
        np.testing.assert_allclose(cov, np.cov(lifts, rowvar=False, bias=True), rtol=0, atol=1e-14)
        eng.close()
        # shared-stream discipline: engine on a torch stream, collective forced in the world of one,
        # no host synchronisation between kernels, all-reduce and merge
        ts, raw = TorchComm.make_stream(torch.device("cuda", 0))
        eng2 = HipEngine(0, stream=raw)
        comm2 = TorchComm(stream=ts, force_collective=True)
        eng2.load_data(*d, 0.0)
        eng2.reset_stats()
        for k in range(4):
            eng2.run_batch(g["perms64"][16 * k:16 * k + 16], True, accumulate=True)
            comm2.allreduce_pending(eng2)
            eng2.merge()
        n2, mean2, cov2 = eng2.stats()
        all_l = HipEngine(0)
        all_l.load_data(*d, 0.0)
        ref_l = all_l.run_batch(g["perms64"], True, want_lifts=True, accumulate=False)
        assert n2 == 64
        np.testing.assert_allclose(mean2, ref_l.mean(0), rtol=0, atol=1e-13)
        np.testing.assert_allclose(cov2, np.cov(ref_l, rowvar=False, bias=True), rtol=0, atol=1e-14)
        eng2.close()
        all_l.close()
        # full driver through the communicator == without it
        a = ls_spa(*d, perms=g["perms64"], batch_size=16, tolerance=0.0, return_attribution_history=True,
                   comm=TorchComm())
        b = ls_spa(*d, perms=g["perms64"], batch_size=16, tolerance=0.0, return_attribution_history=True)
        np.testing.assert_array_equal(a.attribution, b.attribution)
        np.testing.assert_array_equal(a.attribution_history, b.attribution_history)
        # row-sharded reduction and the device-side estimator through the collective path (forced, world of 1)
        cf = TorchComm(force_collective=True)
        c = ls_spa(*d, perms=g["perms64"], batch_size=16, tolerance=0.0, row_sharded=True,
                   error_estimator="device", comm=cf)
        e = ls_spa(*d, perms=g["perms64"], batch_size=16, tolerance=0.0, error_estimator="device")
        np.testing.assert_allclose(c.attribution, b.attribution, rtol=0, atol=1e-13)
        np.testing.assert_allclose(c.theta, b.theta, rtol=1e-12)
        np.testing.assert_allclose(c.error_history, e.error_history, rtol=1e-9)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("p,n,m,shard_test", [(40, 333, 211, True), (130, 700, 90, False), (100, 400, 300, False)])
def test_row_sharded_reduction_two_engines_one_gpu(p, n, m, shard_test):
    """Two contexts play two ranks: each reduces its rows, the Gram sums are added the way the
    all-reduce would (through zero-copy torch views of lsspa_reduce_buffer), both finish -- same
    reduced problem as one context on the stacked rows."""
    import torch
    from ls_spa._engine import HipEngine
    rng = np.random.default_rng(3)
    Xa, Xe = rng.standard_normal((n, p)), rng.standard_normal((m, p))
    w = rng.standard_normal(p)
    ya, ye = Xa @ w + rng.standard_normal(n), Xe @ w + rng.standard_normal(m)
    ref = HipEngine(0)
    ref.load_data(Xa, Xe, ya, ye, 0.05)
    G0, g0, H0, h0 = ref.gram()
    cut_a, cut_e = n // 3, m // 2
    parts = [(Xa[:cut_a], ya[:cut_a], Xe[:cut_e], ye[:cut_e]), (Xa[cut_a:], ya[cut_a:], Xe[cut_e:], ye[cut_e:])]
    engs = [HipEngine(0), HipEngine(0)]
    keep = []
    for rank, (eng, (xa, y1, xe, y2)) in enumerate(zip(engs, parts)):
        if not shard_test:              # replicated test rows; only "rank 0" adds them when m >= p
            xe, y2 = Xe, ye
            m_loc = m if (m < p or rank == 0) else 0
        else:
            m_loc = len(xe)
        arrs = [np.ascontiguousarray(a) for a in (xa, y1, xe, y2)]
        keep.append(arrs)
        eng._check(eng._lib.lsspa_reduce_partial(eng._h, arrs[0].ctypes.data, p, arrs[1].ctypes.data, len(xa),
                                                 arrs[2].ctypes.data, p, arrs[3].ctypes.data, m_loc, m, p, 0, 0))
    eng0, eng1 = engs
    eng0.synchronize(), eng1.synchronize()
    t0 = torch.as_tensor(eng0.reduce_buffer(), device="cuda")
    t1 = torch.as_tensor(eng1.reduce_buffer(), device="cuda")
    t0 += t1
    t1.copy_(t0)
    torch.cuda.synchronize()
    for eng in engs:
        eng.reduce_finish(n, 0.05)
        G, g_, H, h = eng.gram()
        np.testing.assert_allclose(G, G0, rtol=1e-13, atol=1e-14)
        np.testing.assert_allclose(g_, g0, rtol=1e-12, atol=1e-14)
        assert eng.tri == ref.tri and abs(eng.y_norm_sq - ref.y_norm_sq) <= 1e-12 * ref.y_norm_sq
        if ref.tri:
            np.testing.assert_allclose(H, H0, rtol=1e-13, atol=1e-12)
            np.testing.assert_allclose(h, h0, rtol=1e-12, atol=1e-12)
    perms = np.array([rng.permutation(p) for _ in range(6)])
    want = ref.run_batch(perms, True, want_lifts=True, accumulate=False)
    for eng in engs:
        got = eng.run_batch(perms, True, want_lifts=True, accumulate=False)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-11)
        eng.close()
    ref.close()
    with pytest.raises(Exception):
        e3 = HipEngine(0)
        try:   # fewer than p test rows cannot be sharded
            e3._check(e3._lib.lsspa_reduce_partial(e3._h, keep[0][0].ctypes.data, p, keep[0][1].ctypes.data, 10,
                                                   keep[0][2].ctypes.data, p, keep[0][3].ctypes.data, 3, p - 1, p,
                                                   0, 0))
        finally:
            e3.close()
