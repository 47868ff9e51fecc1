"""Two ranks, two HIP engines, ONE GPU: the sharded driver with the real kernels on both sides of a collective.

RCCL refuses two ranks on one device, so the transport here is gloo with the engines' device buffers staged through
the host (TorchComm does that when the backend is not nccl); everything else -- the dealing of each chunk over the
ranks, look-ahead groups launched per rank, the all-reduce of the pending moments, the merge, the stop rule, the
device-side error estimator's partial draws, the row-sharded reduction -- is the code an N-GPU run executes.  The
two-rank results must equal the one-process results of the same calls."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
root = sys.argv[1]; out = sys.argv[2]
sys.path.insert(0, os.path.join(root, "ls-spa_amd"))
import torch, torch.distributed as dist
from ls_spa import ls_spa
from ls_spa._dist import TorchComm
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
res = {}
for tag, p, n, m in (("small", 40, 400, 300), ("general", 150, 900, 800)):
    rng = np.random.default_rng(100 + p)
    Xa = rng.standard_normal((n, p)); Xe = rng.standard_normal((m, p)); th = rng.standard_normal(p)
    ya = Xa @ th + rng.standard_normal(n); ye = Xe @ th + rng.standard_normal(m)
    comm = (lambda: TorchComm()) if world > 1 else (lambda: None)
    kw = dict(reg=1e-3, method="argsort", seed=11, batch_size=16, max_samples=96, tolerance=0.0, device=0)
    a = ls_spa(Xa, Xe, ya, ye, return_attribution_history=True, comm=comm(), **kw)
    b = ls_spa(Xa, Xe, ya, ye, lookahead=3, error_estimator="lowrank", comm=comm(), **kw)
    c = ls_spa(Xa, Xe, ya, ye, error_estimator="device", comm=comm(), **kw)
    if world > 1:
        d = ls_spa(Xa[rank::world], Xe[rank::world], ya[rank::world], ye[rank::world], row_sharded=True, comm=comm(), **kw)
    else:
        d = ls_spa(Xa, Xe, ya, ye, **kw)
    res.update({f"{tag}_attr": a.attribution, f"{tag}_hist": a.attribution_history, f"{tag}_theta": a.theta,
                f"{tag}_la": b.attribution, f"{tag}_la_err": np.array(b.error_history),
                f"{tag}_dev": c.attribution, f"{tag}_dev_err": np.array(c.error_history),
                f"{tag}_shard": d.attribution, f"{tag}_shard_r2": np.array(d.r_squared)})
if rank == 0:
    np.savez(out, **res)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
'''


def _run(world, out, port):
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER, ROOT, out], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=600)[0].decode(errors="replace") for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]


@pytest.mark.parametrize("world", [2, 3])   # 3: a chunk of 16 samples is dealt 6 / 5 / 5
def test_ranks_on_one_gpu_match_one_process(tmp_path, world):
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    _run(1, one, 29631)
    _run(world, two, 29633 + 2 * world)
    a, b = np.load(one), np.load(two)
    for tag in ("small", "general"):
        # same orderings, same kernels; only the order in which the two ranks' moments are summed differs
        np.testing.assert_allclose(b[f"{tag}_attr"], a[f"{tag}_attr"], rtol=0, atol=1e-13)
        np.testing.assert_allclose(b[f"{tag}_hist"], a[f"{tag}_hist"], rtol=0, atol=1e-13)
        np.testing.assert_allclose(b[f"{tag}_theta"], a[f"{tag}_theta"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(b[f"{tag}_la"], a[f"{tag}_la"], rtol=0, atol=1e-13)
        np.testing.assert_allclose(b[f"{tag}_la_err"], a[f"{tag}_la_err"], rtol=1e-9)
        np.testing.assert_allclose(b[f"{tag}_dev"], a[f"{tag}_dev"], rtol=0, atol=1e-13)
        np.testing.assert_allclose(b[f"{tag}_dev_err"], a[f"{tag}_dev_err"], rtol=1e-9)
        # row-sharded reduction: the Gram sums are added in another order -> round-off level differences
        np.testing.assert_allclose(b[f"{tag}_shard"], a[f"{tag}_shard"], rtol=0, atol=1e-10)
        np.testing.assert_allclose(b[f"{tag}_shard_r2"], a[f"{tag}_shard_r2"], rtol=1e-11)
