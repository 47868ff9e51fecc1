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
        t = comm._as_tensor(eng.pending_buffer())
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
                   _comm=TorchComm())
        b = ls_spa(*d, perms=g["perms64"], batch_size=16, tolerance=0.0, return_attribution_history=True)
        np.testing.assert_array_equal(a.attribution, b.attribution)
        np.testing.assert_array_equal(a.attribution_history, b.attribution_history)
    finally:
        dist.destroy_process_group()
