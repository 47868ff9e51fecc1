#!/usr/bin/env python3
"""Benchmark of the LS-SPA hot path on MI355X (driver contract: one JSON line on rank 0).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Default workload = BASELINE.json configs[2] (C3): p = 1000 features, N = M = 100000 rows of the synthetic Gaussian
data of BASELINE.md section 3 (default_rng(0)), method='argsort' (Sobol, seed 42), batch_size = 128 antithetical
samples = 256 orderings per step, fp64.  Other configs:  --p 100 --rows 10000 (C2),
--p 5000 --rows 200000 --dtype f32 (C5: float32 data and per-ordering work, reg = 1e-2).

A step = one batch through the whole per-batch path: ordering upload -> permuted gather -> blocked Cholesky ->
strip solve -> lifts -> batch moments -> (all-reduce over the ranks: RCCL through the C ABI) -> merge.  The data
and its one-time Gram reduction are resident in HBM before the timed region; the reduction is timed and reported
separately.  --scaling weak (default): every rank evaluates its own batch_size samples per step (global batch =
N x batch_size).  --scaling strong: BASELINE config 4's semantics, the global batch of batch_size samples is dealt
over the ranks (batch_size / N samples each).  The only data-path collective is the all-reduce of the moments.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ls-spa_amd"))

import numpy as np  # noqa: E402

FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix = vector peak (vendor sheet)
FP32_PEAK_TFLOPS = 157.3  # MI355X fp32 matrix peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--p", type=int, default=1000)
    ap.add_argument("--rows", type=int, default=100000)
    ap.add_argument("--batch-size", type=int, default=128)
    ap.add_argument("--dtype", choices=("f64", "f32"), default="f64",
                    help="element type of the data and of the per-ordering factorisation work (C3: f64, C5: f32)")
    ap.add_argument("--reg", type=float, default=None, help="ridge term (default: 1e-2 for --dtype f32 as in C5, else 0)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--collective", choices=("native", "torch"), default="native",
                    help="transport of the all-reduce with several ranks: RCCL through the C ABI, or torch.distributed")
    ap.add_argument("--lookahead", type=int, default=0,
                    help="steps whose orderings are launched as one GPU batch and then accumulated / all-reduced / merged "
                         "step by step (what ls_spa(lookahead=k) does); 0 = auto: 8 for p <= 126, 4 when a rank's step "
                         "has <= 32 samples, else 1")
    ap.add_argument("--lanes", type=int, choices=(1, 2), default=1,
                    help="batches in flight on the engine (lsspa_set_lanes): 2 = the next step's kernels run beside this one's")
    ap.add_argument("--no-probe", action="store_true", help="skip the strong-scaling probe (clean rocprof averages)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ttt", action="store_true", help="skip the time-to-tolerance runs")
    return ap.parse_args()


def config_label(p, rows, dtype):
    if (p, rows, dtype) == (100, 10000, "f64"):
        return "C2"
    if (p, rows, dtype) == (1000, 100000, "f64"):
        return "C3"
    if (p, rows, dtype) == (5000, 200000, "f32"):
        return "C5"
    return "custom"


def baseline_data(p, rows, dtype):
    """BASELINE.md section 3 / SURVEY.md 8d: default_rng(0), X ~ N(0,1), theta ~ N(0,1)^p, y = X theta + N(0,1); cast
    to float32 for the fp32 config.  Rows are drawn in blocks (the generator fills element by element, so the
    stream -- hence the data -- is that of one (rows, p) call) to bound the float64 temporaries."""
    rng = np.random.default_rng(0)
    dt = np.float32 if dtype == "f32" else np.float64
    block = max(1, (256 << 20) // (8 * p))

    def matrix():
        out = np.empty((rows, p), dtype=dt)
        for r0 in range(0, rows, block):
            r1 = min(rows, r0 + block)
            out[r0:r1] = rng.standard_normal((r1 - r0, p))
        return out

    Xa, Xe = matrix(), matrix()
    theta = rng.standard_normal(p)
    ya = (Xa @ theta.astype(dt)).astype(np.float64) + rng.standard_normal(rows)
    ye = (Xe @ theta.astype(dt)).astype(np.float64) + rng.standard_normal(rows)
    return Xa, Xe, ya.astype(dt), ye.astype(dt)


def algorithmic_flops(kclass, p, n_ord, tri, launches_per_batch):
    """Useful flops of one launch of a kernel class (element granularity, no padding, no
    redundant tile work), so that 'achieved' cannot be inflated by wasted arithmetic."""
    n_mats = n_ord * (2 if tri else 1)
    if kclass == "strip":          # triangular-triangular solve V = L^-1 L_t (p^3/3)
        per = p ** 3 / 3.0 if tri else float(p) ** 3
        return per * n_ord / launches_per_batch
    if kclass == "chol_panel":     # the whole Cholesky, p^3/3 per matrix: the panel launches also factor the
        return (p ** 3 / 3.0) * n_mats / launches_per_batch   # diagonal blocks (all but block 0)
    if kclass == "chol_diag":      # stand-alone launch: block 0 only (factor + inverse)
        return (64 ** 3 / 3.0 * 2) * n_mats / launches_per_batch
    if kclass == "small_p":        # fused small-p kernel: the whole per-ordering work, ~p^3
        return float(p) ** 3 * n_ord / launches_per_batch
    return 0.0


def host_threads():
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def cpu_baseline(host, G, g, H, h, yy, p, reg, n_stop, n_checks, cov_at_stop, seconds=20.0):
    """The oracle (NumPy/SciPy restatement of the reference's algorithm, kind "port") on this host's cores:
    per-ordering rate on a bounded sample (QR + triangular solve + GEMM per ordering, ls_spa/ls_spa.py:256-287), the
    reduction timed once (:290-318; on a row sample, scaled, when the full QR would take minutes), one
    error_estimates call (:321-341) on the covariance the GPU run stopped on, and the CPU time to tolerance
    composed from them as BASELINE.md section 3 prescribes."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import lsspa_oracle as O
    from threadpoolctl import threadpool_limits
    ncpu = host_threads()
    R = np.linalg.cholesky(G).T
    F = np.linalg.cholesky(H).T
    q = np.linalg.solve(R.T, g)
    qt = np.linalg.solve(F.T, h)
    rng = np.random.default_rng(0)
    cand = sorted({t for t in (1, 2, 4, 8, 16, 32) if t <= ncpu})
    best_t, best_rate = 1, 0.0
    O.ordering_lift(R, F, q, qt, yy, rng.permutation(p))     # warm-up
    reps = 2 if p >= 1000 else 50
    for t in cand:
        with threadpool_limits(limits=t):
            t0 = time.perf_counter()
            for _ in range(reps):
                O.ordering_lift(R, F, q, qt, yy, rng.permutation(p))
            rate = reps / (time.perf_counter() - t0)
        if rate > best_rate:
            best_t, best_rate = t, rate
    with threadpool_limits(limits=best_t):
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds and n < 20000:
            O.ordering_lift(R, F, q, qt, yy, rng.permutation(p))
            n += 1
        dt = time.perf_counter() - t0
    rate = n / dt
    out = {"value": rate, "unit": "orderings/s", "cores": best_t, "blas_threads": best_t, "host_cores": ncpu,
           "host_cores_total": os.cpu_count(), "processes": 1, "kind": "port",
           "sample": f"{n} orderings of the same reduced problem (p={p}) in {dt:.1f} s; oracle ordering_lift = QR + "
                     f"trtrs + GEMM per ordering; one process, best BLAS thread count of {cand}"}
    # reduction: two Householder QRs with explicit Q, ~4 (N + M) p^2 flop
    Xa, Xe, ya, ye = host
    rows = Xa.shape[0]
    red_threads = min(ncpu, 32)
    full_cost = 4.0 * 2 * rows * p * p
    sub = rows if full_cost <= 2e12 else max(4 * p, int(rows * 2e12 / full_cost))
    with threadpool_limits(limits=red_threads):
        t0 = time.perf_counter()
        O.reduce(np.asarray(Xa[:sub], dtype=np.float64), np.asarray(Xe[:sub], dtype=np.float64),
                 np.asarray(ya[:sub], dtype=np.float64), np.asarray(ye[:sub], dtype=np.float64), reg)
        red_s = (time.perf_counter() - t0) * (rows / sub)
    out["reduction_s"] = red_s
    out["reduction_note"] = (f"oracle reduce (two QRs, ls_spa/ls_spa.py:309-317) on {sub} of {rows} rows per side"
                             + ("" if sub == rows else ", scaled linearly to all rows") + f", {red_threads} BLAS threads")
    # error estimate on the covariance of the stopping check
    est_s = None
    if cov_at_stop is not None and n_stop > 1:
        with threadpool_limits(limits=min(ncpu, 16)):
            t0 = time.perf_counter()
            O.error_quantiles(np.random.default_rng(1), cov_at_stop * n_stop / (n_stop - 1) / n_stop)
            est_s = time.perf_counter() - t0
        out["error_estimates_s"] = est_s
    if est_s is not None:
        out["time_to_tolerance_s"] = red_s + (2 * n_stop) / rate + n_checks * est_s
        out["time_to_tolerance_note"] = (f"reduction + {n_stop} samples x 2 orderings / rate + {n_checks} x error_estimates "
                                         "(BASELINE.md section 3; the stop index is the GPU run's)")
    return out


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from ls_spa import ls_spa
    from ls_spa._engine import HipEngine
    from ls_spa._driver import run_estimator, _Comm

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # developer switch: run the multi-rank code path (communicator, forced collective) in a world of one, to
    # rehearse on a single GPU exactly what the N > 1 launch executes
    rehearse = world == 1 and os.environ.get("LSSPA_BENCH_REHEARSE_DIST") == "1"
    multi = world > 1 or rehearse

    p, rows, B = args.p, args.rows, args.batch_size
    reg = args.reg if args.reg is not None else (1e-2 if args.dtype == "f32" else 0.0)
    label = config_label(p, rows, args.dtype)
    if args.scaling == "strong" and B % world:
        raise SystemExit("--scaling strong needs batch_size divisible by the number of ranks")
    B_rank = B // world if args.scaling == "strong" else B

    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # RCCL prints a version banner on stdout when its first communicator comes up; stdout is reserved
        # for the one JSON line, so park fd 1 on stderr until the communicators exist
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
    eng = HipEngine(local)
    comm, collective = _Comm(), "none"
    try:
        if multi:
            # torch.distributed serves the driver contract's barrier and the max-over-ranks of the timing; the data
            # path's all-reduce goes through the engine's own RCCL communicator on the engine's stream
            if rehearse:
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            else:
                dist.init_process_group("nccl", device_id=dev)
            warm = torch.zeros(1, device=dev)
            dist.all_reduce(warm)
            torch.cuda.synchronize()
            if args.collective == "native":
                try:
                    from ls_spa._rccl import NativeComm
                    comm = NativeComm.from_env(force_collective=rehearse)
                    comm.bind(eng)
                    collective = "rccl via the C ABI (lsspa_stats_allreduce, engine stream)"
                except Exception as exc:   # keep the measurement alive: fall back to the torch transport, say so
                    sys.stderr.write(f"[bench] native communicator failed ({exc}); using torch.distributed\n")
                    comm = None
            if args.collective == "torch" or comm is None:
                from ls_spa._dist import TorchComm
                comm = TorchComm(force_collective=rehearse)
                collective = "torch.distributed nccl (host-synchronised hand-off between the engine's and torch's stream)"
    finally:
        if multi:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    # the benchmark data (same on every rank), generated on the host as BASELINE.md prescribes, then moved to HBM
    t0 = time.perf_counter()
    host = baseline_data(p, rows, args.dtype)
    gen_s = time.perf_counter() - t0
    tdt = torch.float32 if args.dtype == "f32" else torch.float64
    dXa, dXe, dya, dye = (torch.from_numpy(a).to(dev) for a in host)
    assert dXa.dtype == tdt
    torch.cuda.synchronize()

    if args.dtype == "f32":
        eng.set_precision("float32")
    peak_tf = FP32_PEAK_TFLOPS if args.dtype == "f32" else FP64_PEAK_TFLOPS
    esz = 4 if args.dtype == "f32" else 8
    eng.profile(True)
    t0 = time.perf_counter()
    eng.load_device_data(dXa.data_ptr(), p, dya.data_ptr(), rows, dXe.data_ptr(), p, dye.data_ptr(), rows, p, reg,
                         f32=args.dtype == "f32")
    eng.synchronize()
    reduce_ms = 1e3 * (time.perf_counter() - t0)
    # the Gram kernels once more on the resident data: the first call is also the process's first GPU work (cold
    # clocks, code object load), which a roofline figure for the kernel should not carry
    eng.profile_reset()
    t0 = time.perf_counter()
    eng.load_device_data(dXa.data_ptr(), p, dya.data_ptr(), rows, dXe.data_ptr(), p, dye.data_ptr(), rows, p, reg,
                         f32=args.dtype == "f32")
    eng.synchronize()
    reduce_warm_ms = 1e3 * (time.perf_counter() - t0)
    gram_ms, gram_n = eng.profile_read()["gram"]
    del dXa, dXe, dya, dye
    torch.cuda.empty_cache()

    # orderings of every step, generated up front on the host (Sobol argsort); the global sequence is
    # dealt over the ranks: whole batches (weak) or round-robin inside each batch (strong, as the driver deals them)
    from ls_spa import _samplers as S
    src = S.ArgsortSource(p, 42, 2 ** 62)
    total_steps = args.warmup + args.steps
    if args.scaling == "strong":
        all_perms = src.take(total_steps * B).astype(np.int32).reshape(total_steps, B, p)
        my_perms = np.ascontiguousarray(all_perms[:, rank::world])
    else:
        all_perms = src.take(total_steps * B * world).astype(np.int32).reshape(total_steps, world, B, p)
        my_perms = np.ascontiguousarray(all_perms[:, rank])
    del all_perms
    n_ord = 2 * B_rank

    def barrier():
        eng.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # auto: a step that fills a fraction of the GPU (few samples per rank, or the one-workgroup-per-ordering kernel
    # of small problems) is launched in groups
    D = args.lookahead if args.lookahead > 0 else (8 if p + 1 <= 128 else (4 if B_rank <= 32 else 1))

    class Steps:
        """step(k) = one batch of B_rank samples into the statistics.  With D > 1 the kernels of D consecutive steps
        are launched together (one gather / factorisation / solve sequence over D * B_rank samples) and each step
        then folds its own B_rank lift vectors into the pending buffer, all-reduces and merges -- the reference's
        per-batch order, a fuller GPU.  Groups start at the first step of a region."""

        def __init__(self):
            self.tickets, self.base, self.end = {}, 0, 0

        def region(self, k0, k1):
            for tk in self.tickets.values():      # a group launched ahead across a region boundary
                eng.discard_batch(tk)
            self.tickets, self.base, self.end = {}, k0, k1

        def launch(self, g):
            lo = self.base + g * D
            if lo < self.end and g not in self.tickets:
                self.tickets[g] = eng.launch_batch(my_perms[lo:min(lo + D, self.end)].reshape(-1, p), True)

        def __call__(self, k):
            if D == 1:
                eng.run_batch(my_perms[k], True, want_lifts=False, accumulate=True)
            else:
                g, j = divmod(k - self.base, D)
                if j == 0:
                    self.launch(g)
                    if eng.lanes == 2:
                        self.launch(g + 1)       # the second lane holds the next group: its upload and kernels run
                                                 # while this group is accumulated step by step
                eng.collect_batch(self.tickets[g], want_lifts=False, accumulate=True, first=j * B_rank, count=B_rank)
                if j == D - 1 or k == self.end - 1:
                    del self.tickets[g]
            comm.allreduce_pending(eng)
            eng.merge()

    step = Steps()

    eng.set_lanes(args.lanes)
    # pass 1: the timed region proper (no events between the launches: an event record costs a
    # ~10 us bubble per kernel boundary)
    eng.profile(False)
    eng.reset_stats()
    step.region(0, args.warmup)
    for k in range(args.warmup):
        step(k)
    barrier()
    step.region(args.warmup, total_steps)
    t0 = time.perf_counter()
    for k in range(args.warmup, total_steps):
        step(k)
    barrier()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_seen, mean, _ = eng.stats(want_cov=False)
    # pass 2: the same K steps again with a HIP-event pair around every launch on the engine's
    # stream -> per-kernel durations for the roofline figures (one lane: each kernel alone on the GPU)
    eng.set_lanes(1)
    eng.profile(True)
    eng.profile_reset()
    step.region(args.warmup, total_steps)
    for k in range(args.warmup, total_steps):
        step(k)
    barrier()
    prof = eng.profile_read()
    eng.profile(False)

    # what one of 8 ranks runs per step under strong scaling (C4): batch_size / 8 samples on this GPU
    probe = None
    if world == 1 and B >= 16 and args.scaling == "weak" and not args.no_probe:
        eng.set_lanes(args.lanes)
        b8, dd, reps = B // 8, 4, 24
        pool = np.ascontiguousarray(my_perms[:2].reshape(-1, p)[: dd * b8])

        def probe_steps(group):
            for r in range(reps):
                j = r % group
                if group == 1:
                    eng.run_batch(pool[:b8], True, want_lifts=False, accumulate=True)
                else:
                    if j == 0:
                        tk = eng.launch_batch(pool[: group * b8], True)
                    eng.collect_batch(tk, want_lifts=False, accumulate=True, first=j * b8, count=b8)
                eng.merge()
            eng.synchronize()

        res = {}
        for group in (1, dd):
            probe_steps(group)
            t0 = time.perf_counter()
            probe_steps(group)
            res[group] = 1e3 * (time.perf_counter() - t0) / reps
        probe = {"samples_per_step": b8, "orderings_per_step": 2 * b8, "ms_per_step": res[dd],
                 "orderings_per_s": 2 * b8 / (res[dd] * 1e-3), "lookahead": dd, "ms_per_step_without_lookahead": res[1],
                 "note": "the per-rank step of an 8-GPU run with the global batch dealt over the ranks (--scaling strong): "
                         "kernels of 4 steps launched together, statistics / all-reduce / merge per step"}
        eng.set_lanes(1)

    out = None
    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        value = world * n_ord * args.steps / elapsed
        per_class = {k: {"ms_per_step": v[0] / args.steps, "launches_per_step": v[1] / args.steps,
                         "avg_launch_ms": (v[0] / v[1]) if v[1] else None}
                     for k, v in prof.items() if v[1]}
        mfma_classes = [k for k in ("strip", "chol_panel", "chol_diag", "small_p") if k in per_class]
        dom = max(mfma_classes, key=lambda k: per_class[k]["ms_per_step"])
        lpb = per_class[dom]["launches_per_step"]
        flops = algorithmic_flops(dom, p, n_ord, eng.tri, lpb)
        ach = flops / (per_class[dom]["avg_launch_ms"] * 1e-3) / 1e12
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                for rec in json.load(open(pmc)).get("runs", []):
                    if (rec.get("p"), rec.get("batch_size"), rec.get("dtype")) == (p, B_rank, args.dtype):
                        traffic = rec.get("hbm_bytes_per_launch", {}).get(dom)
            except Exception:
                traffic = None
        roofline = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": peak_tf,
                    "unit": "TFLOP/s", "frac": ach / peak_tf, "traffic": traffic,
                    "algorithmic_flops_per_launch": flops, "avg_launch_ms": per_class[dom]["avg_launch_ms"],
                    "traffic_gbps": (traffic / (per_class[dom]["avg_launch_ms"] * 1e-3) / 1e9) if traffic else None,
                    "note": "peak = vendor fp64 (fp32) matrix figure; on random operands at the steady-state clock the "
                            "isolated k-loop of these kernels sustains ~60 TFLOP/s fp64 fed from HBM and ~67 fed from the "
                            "caches, the vendor library's GEMM 69-71 at 8192^3 and 59-67 batched at 1024^3 "
                            "(profiles/r02_kloop_ceiling.log)"}
        out = {
            "metric": "orderings_per_sec", "value": value, "unit": "orderings/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{label} p={p} N=M={rows} reg={reg:g} method=argsort batch_size={B} antithetical "
                                   f"(={n_ord} orderings/step/GPU) "
                                   f"{'fp64' if args.dtype == 'f64' else 'fp32 data and per-ordering work / fp64 accumulation'}",
                       "p": p, "N": rows, "M": rows, "reg": reg, "batch_size": B, "global_batch": B * world if
                       args.scaling == "weak" else B, "orderings_per_step_per_gpu": n_ord,
                       "path": "tri" if eng.tri else "rect", "collective": collective, "lanes": args.lanes, "lookahead": D,
                       "data_generator": "BASELINE.md section 3: default_rng(0) on the host, moved to HBM before timing"},
            "roofline": roofline,
            "kernels": per_class,
            "reduction_ms": reduce_ms, "reduction_ms_second_call": reduce_warm_ms,
            "host_data_generation_s": gen_s,
            "check": {"samples": int(n_seen), "sum_attribution": float(mean.sum())},
            "timing_note": "value/ms_per_step: K steps without events; kernels/roofline: the same K steps "
                           "repeated with a HIP-event pair around every launch on the engine's stream",
        }
        if "gather" in per_class:
            g_bytes = 2.0 * p * p * esz * n_ord         # 2 p^2 s bytes per ordering
            g_ach = g_bytes / (per_class["gather"]["avg_launch_ms"] * 1e-3) / 1e9
            out["roofline_gather"] = {"bound": "hbm", "achieved": g_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": g_ach / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": g_bytes}
        gram_flops = 2.0 * rows * (p + 1) * (p + 2) / 2   # (N + M)(p + 1)(p + 2) over both sides -> per launch
        gram_avg = gram_ms / max(gram_n, 1)
        gram_ach = gram_flops / (gram_avg * 1e-3) / 1e12
        out["roofline_gram"] = {"bound": "mfma", "achieved": gram_ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": gram_ach / FP64_PEAK_TFLOPS, "algorithmic_flops_per_launch": gram_flops,
                                "avg_launch_ms": gram_avg, "launches": int(gram_n),
                                "note": "one launch per side = the Gram contraction kernel plus its fixed-order slab "
                                        "reduction, second call on the resident data; fp64 accumulation in both data types"}
        if probe is not None:
            probe["per_ordering_throughput_vs_full_step"] = probe["orderings_per_s"] / value
            out["strong_scaling_probe"] = probe

    # ---- time to tolerance (SURVEY 8d ii): the reference's own stopping rule, sharded over the ranks
    n_stop = n_checks = 0
    cov_at_stop = None
    if not args.no_ttt:
        def leg(estimator, reps=1):
            best = None
            for _ in range(reps):
                barrier()
                t0 = time.perf_counter()
                res = run_estimator(eng, p, max_samples=B * 128, batch_size=B, tolerance=1e-2, seed=42, perms=None,
                                    antithetical=True, return_attribution_history=False, method="argsort",
                                    error_estimator=estimator, comm=comm)
                barrier()
                dt = time.perf_counter() - t0
                best = (dt, res) if best is None or dt < best[0] else best
            return best

        legs = {}
        for name, estimator, reps in (("time_to_tolerance", "reference", 1), ("time_to_tolerance_lowrank", "lowrank", 1),
                                      ("time_to_tolerance_device", "device", 2)):
            try:
                dt, res = leg(estimator, reps)
                legs[name] = {"seconds_sampling_loop": dt, "seconds_incl_reduction": dt + reduce_ms * 1e-3,
                              "samples_at_stop": int(res[5]), "checks": int(len(res[3])),
                              "overall_error": float(res[2]), "tolerance": 1e-2, "error_estimator": estimator,
                              "h2d_included": False,
                              "note": "sampling loop on the HBM-resident reduced problem of the timed region"}
                if name == "time_to_tolerance":
                    n_stop, n_checks = int(res[5]), int(len(res[3]))
                    if rank == 0 and world == 1:
                        cov_at_stop = eng.stats(want_cov=True)[2]
            except Exception as exc:   # a failing leg must not cost the throughput line
                legs[name] = {"error": repr(exc)}
        # the whole public call on the host arrays: reduction over PCIe included.  One GPU (or the one-rank rehearsal)
        # only: with several ranks every call would make its own communicator, and a rank that fails to while the
        # others succeed would leave them waiting in a collective -- the sharded sampling loop above is the N > 1 figure
        try:
            if world > 1:
                raise RuntimeError("skipped with several ranks (see time_to_tolerance* for the sharded loop)")
            kw = dict(reg=reg, method="argsort", batch_size=B, num_batches=128, tolerance=1e-2, seed=42,
                      device=local, precision="float32" if args.dtype == "f32" else "float64")
            e2e = {}
            for name, estimator in (("reference", "reference"), ("device", "device")):
                times = []
                for _ in range(2):
                    c2 = None
                    if multi and collective.startswith("rccl"):
                        from ls_spa._rccl import NativeComm
                        c2 = NativeComm.from_env(force_collective=rehearse)
                        c2._port += 2 + len(times) + (10 if name == "device" else 0)
                    elif multi:
                        c2 = comm
                    barrier()
                    t0 = time.perf_counter()
                    r = ls_spa(*host, error_estimator=estimator, comm=c2, **kw)
                    times.append(time.perf_counter() - t0)
                e2e[name] = {"seconds": min(times), "first_call_seconds": times[0],
                             "samples_at_stop": int(128 * len(r.error_history)) if len(r.error_history) else 0,
                             "overall_error": float(r.overall_error), "error_estimator": estimator}
            legs["time_to_tolerance_e2e"] = dict(
                e2e, h2d_included=True, tolerance=1e-2,
                note="public ls_spa() on the BASELINE.md section 3 host arrays (default_rng(0)): engine creation, "
                     "streamed reduction over PCIe, sampling loop, estimator, final fit, teardown")
        except Exception as exc:
            legs["time_to_tolerance_e2e"] = {"error": repr(exc)}
        if out is not None:
            out.update(legs)

    if out is not None and world == 1 and not args.no_cpu_baseline:
        G, g, H, h = eng.gram()
        out["cpu_baseline"] = cpu_baseline(host, G, g, H, h, eng.y_norm_sq, p, reg, n_stop, n_checks, cov_at_stop)
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    if out is not None:
        print(json.dumps(out))
    if hasattr(comm, "close"):
        comm.close()
    eng.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
