"""Multi-GPU sharding: one process per GPU, torch.distributed as the transport.

Orderings are independent given the (replicated) reduced problem, so each global chunk of
orderings is dealt round-robin over the ranks and the only data-path collective is ONE
all-reduce (SUM, fp64) per chunk of the packed pending statistics
[n_b, sum(l - mu), sum (l - mu)(l - mu)^T] -- the multi-device form of the reference's
merge_sample_mean / merge_sample_cov (cvxgrp/ls-spa ls_spa/ls_spa.py:103-119, :212-216).
With backend "nccl" (= RCCL on ROCm) the buffer is reduced in place in HBM over xGMI;
with "gloo" it is a host tensor (CPU tests) or a device buffer staged through the host (the two-ranks-on-one-GPU
test).  After the collective every rank holds the
same moments, merges them identically and therefore takes the same stop decision.

Stream discipline on the GPU: if the engine was created on a torch stream
(``TorchComm.make_stream()`` -> ``HipEngine(device, stream=...)``) the kernels, the collective
and the merge are ordered on that one stream and the host never blocks between them;
otherwise the engine's own stream and torch's are joined by host synchronisation.
"""
from __future__ import annotations

import numpy as np


class TorchComm:
    def __init__(self, group=None, stream=None, force_collective=False):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._torch, self._dist, self._group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self._on_gpu = dist.get_backend(group) == "nccl"
        self._stream = stream            # torch.cuda.Stream shared with the engine, or None
        self._force = force_collective   # tests: run the collective even in a world of one

    @staticmethod
    def make_stream(device):
        """A torch stream for the engine to run on; returns (torch_stream, raw hipStream_t)."""
        import torch
        s = torch.cuda.Stream(device=device)
        return s, s.cuda_stream

    def _as_tensor(self, buf, engine):
        torch = self._torch
        if isinstance(buf, np.ndarray):
            return torch.from_numpy(buf)            # shares memory with the test double's buffer
        # zero-copy view through __cuda_array_interface__, on the ENGINE's device: with device="cuda" torch takes
        # its current device and, if that is another GPU, silently copies the buffer there -- the collective would
        # then reduce the copy and the engine's own buffer would never see the other ranks' moments
        t = torch.as_tensor(buf, device=torch.device("cuda", engine.device))
        ptr = buf.__cuda_array_interface__["data"][0]
        if t.data_ptr() != ptr:
            raise RuntimeError(f"torch copied the engine's buffer (engine on cuda:{engine.device}, tensor on "
                               f"{t.device}): the all-reduce would not reach the engine")
        return t

    def allreduce_pending(self, engine):
        self._allreduce(engine, engine.pending_buffer)

    def allreduce_draws(self, engine):
        """Sum the ranks' partial error-estimator draws (device-side estimator, 1024 x p fp64)."""
        self._allreduce(engine, engine.draws_buffer)

    def allreduce_reduction(self, engine):
        """Sum the ranks' Gram sums (row-sharded reduction)."""
        self._allreduce(engine, engine.reduce_buffer)

    def sum_ints(self, values):
        # on the GPU backend the tensor lives on torch's current device: one process per GPU sets it
        # (torch.cuda.set_device(LOCAL_RANK)) before the process group is made, as RCCL itself requires
        t = self._torch.tensor([int(v) for v in values], dtype=self._torch.int64,
                               device=self._torch.device("cuda", self._torch.cuda.current_device())
                               if self._on_gpu else "cpu")
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self._group)
        return [int(v) for v in t.cpu().tolist()]

    def gather_ints(self, values):
        from ._driver import gather_ints_by_sum
        return gather_ints_by_sum(self, values)

    def _allreduce(self, engine, get_buffer):
        if self.world == 1 and not self._force:
            return
        t = self._as_tensor(get_buffer(), engine)
        if not self._on_gpu:
            if t.is_cuda:
                # a HIP engine under a CPU transport (gloo: several ranks sharing one GPU in the tests -- RCCL refuses
                # two ranks on one device): the buffer is staged through the host around the collective
                engine.synchronize()
                h = t.cpu()
                self._dist.all_reduce(h, op=self._dist.ReduceOp.SUM, group=self._group)
                t.copy_(h)
                self._torch.cuda.synchronize(t.device)
                return
            self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self._group)
            return
        if self._torch.cuda.current_device() != engine.device:
            raise RuntimeError(f"engine on cuda:{engine.device} but torch's current device is "
                               f"cuda:{self._torch.cuda.current_device()}: call torch.cuda.set_device first")
        if self._stream is not None:
            # engine kernels, collective and merge are all ordered on the shared stream
            with self._torch.cuda.stream(self._stream):
                self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self._group)
            return
        engine.synchronize()                        # engine stream -> torch stream hand-off
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self._group)
        self._torch.cuda.current_stream().synchronize()

    def gather_lifts(self, local, counts):
        """All ranks' per-sample lift vectors (only needed for attribution_history /
        the low-rank error estimator)."""
        if self.world == 1:
            return local
        torch = self._torch
        p = local.shape[1]
        dev = "cuda" if self._on_gpu else "cpu"
        width = max(counts)
        mine = torch.zeros((width, p), dtype=torch.float64, device=dev)
        if len(local):
            mine[: len(local)] = torch.from_numpy(np.ascontiguousarray(local)).to(dev)
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        self._dist.all_gather(parts, mine, group=self._group)
        return [parts[r][: counts[r]].cpu().numpy() for r in range(self.world)]
