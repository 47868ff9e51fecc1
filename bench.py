#!/usr/bin/env python3
"""Benchmark of the LS-SPA hot path on MI355X (driver contract: one JSON line on rank 0).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2]): p = 1000 features, N = M = 100000 synthetic Gaussian rows,
method='argsort' (Sobol), batch_size = 128 antithetical samples = 256 orderings per step, fp64.
A step = one batch through the whole per-batch path: ordering upload -> permuted gather ->
blocked Cholesky -> strip solve -> lifts -> batch moments -> (all-reduce over ranks) -> merge.
The data and its one-time Gram reduction are resident in HBM before the timed region; the
reduction is timed and reported separately.  With N ranks every rank evaluates its own 128
samples per step (weak scaling); the only collective is the all-reduce of the packed moments.
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
                    help="element type of the per-ordering factorisation work (BASELINE C3 is f64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ttt", action="store_true", help="skip the time-to-tolerance run")
    return ap.parse_args()


def algorithmic_flops(kclass, p, n_ord, tri, launches_per_batch):
    """Useful flops of one launch of a kernel class (element granularity, no padding, no
    redundant tile work), so that 'achieved' cannot be inflated by wasted arithmetic."""
    nblk = p // 64 + 1
    n_mats = n_ord * (2 if tri else 1)
    if kclass == "strip":          # triangular-triangular solve V = L^-1 L_t (p^3/3)
        per = p ** 3 / 3.0 if tri else float(p) ** 3
        return per * n_ord / launches_per_batch
    if kclass == "chol_panel":     # the whole Cholesky, p^3/3 per matrix: the panel launches also factor the
        return (p ** 3 / 3.0) * n_mats / launches_per_batch   # diagonal blocks (all but block 0)
    if kclass == "chol_diag":      # stand-alone launch: block 0 only (factor + inverse)
        return (64 ** 3 / 3.0 * 2) * n_mats / launches_per_batch
    return 0.0


def cpu_baseline(G, g, H, h, yy, p, seconds=20.0):
    """The oracle (numpy restatement of the reference's per-ordering algorithm: QR + triangular
    solve + GEMM) on this host's cores, on a bounded sample of the same reduced problem."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import lsspa_oracle as O
    R = np.linalg.cholesky(G).T
    F = np.linalg.cholesky(H).T
    q = np.linalg.solve(R.T, g)
    qt = np.linalg.solve(F.T, h)
    rng = np.random.default_rng(0)
    try:
        ncpu = len(os.sched_getaffinity(0))
    except Exception:
        ncpu = os.cpu_count() or 1
    try:
        from threadpoolctl import threadpool_limits
    except Exception:       # pragma: no cover
        threadpool_limits = None
    cand = sorted({t for t in (1, 2, 4, 8, 16, 32) if t <= ncpu})
    best_t, best_rate = 1, 0.0
    O.ordering_lift(R, F, q, qt, yy, rng.permutation(p))     # warm-up
    if threadpool_limits is not None:
        for t in cand:
            with threadpool_limits(limits=t):
                t0 = time.perf_counter()
                for _ in range(2):
                    O.ordering_lift(R, F, q, qt, yy, rng.permutation(p))
                rate = 2 / (time.perf_counter() - t0)
            if rate > best_rate:
                best_t, best_rate = t, rate
    ctx = threadpool_limits(limits=best_t) if threadpool_limits is not None else None
    if ctx is not None:
        ctx.__enter__()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds and n < 4096:
        O.ordering_lift(R, F, q, qt, yy, rng.permutation(p))
        n += 1
    dt = time.perf_counter() - t0
    if ctx is not None:
        ctx.__exit__(None, None, None)
    return {"value": n / dt, "unit": "orderings/s", "cores": best_t, "kind": "port",
            "sample": f"{n} orderings of the same reduced problem (p={p}) in {dt:.1f} s; "
                      f"oracle ordering_lift = QR + trtrs + GEMM; best of BLAS threads {cand}; "
                      f"{ncpu} CPUs available to this process ({os.cpu_count()} on the host)"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from ls_spa._engine import HipEngine
    from ls_spa._dist import TorchComm
    from ls_spa._driver import run_estimator, _Comm
    from ls_spa import _samplers as S

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    eng_stream = None
    # developer switch: run the multi-rank code path (RCCL group, shared stream, forced collective)
    # in a world of one, to rehearse on a single GPU exactly what the N > 1 launch executes
    rehearse = world == 1 and os.environ.get("LSSPA_BENCH_REHEARSE_DIST") == "1"
    if world > 1 or rehearse:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # RCCL prints a version banner on stdout when its first communicator comes up; stdout is reserved
        # for the one JSON line, so park fd 1 on stderr until the communicator exists
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if rehearse:
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            else:
                dist.init_process_group("nccl", device_id=dev)
            warm = torch.zeros(1, device=dev)
            dist.all_reduce(warm)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
        # engine kernels, the all-reduce and the merge share one torch stream: no host sync per step
        tstream, eng_stream = TorchComm.make_stream(dev)
        comm = TorchComm(stream=tstream, force_collective=rehearse)
    else:
        comm = _Comm()

    p, rows, B = args.p, args.rows, args.batch_size
    # synthetic Gaussian data of the benchmark shape, generated on the device (same on every rank)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)
    Xa = torch.randn(rows, p, dtype=torch.float64, device=dev, generator=gen)
    Xe = torch.randn(rows, p, dtype=torch.float64, device=dev, generator=gen)
    w = torch.randn(p, dtype=torch.float64, device=dev, generator=gen)
    ya = Xa @ w + torch.randn(rows, dtype=torch.float64, device=dev, generator=gen)
    ye = Xe @ w + torch.randn(rows, dtype=torch.float64, device=dev, generator=gen)
    torch.cuda.synchronize()

    eng = HipEngine(local, stream=eng_stream)
    if args.dtype == "f32":
        eng.set_precision("float32")
    peak_tf = FP32_PEAK_TFLOPS if args.dtype == "f32" else FP64_PEAK_TFLOPS
    esz = 4 if args.dtype == "f32" else 8
    eng.profile(True)
    t0 = time.perf_counter()
    eng.load_device_data(Xa.data_ptr(), p, ya.data_ptr(), rows, Xe.data_ptr(), p, ye.data_ptr(), rows, p, 0.0)
    eng.synchronize()
    reduce_ms = 1e3 * (time.perf_counter() - t0)
    gram_ms, gram_n = eng.profile_read()["gram"]
    del Xa, Xe, ya, ye
    torch.cuda.empty_cache()

    # orderings of every step, generated up front on the host (Sobol argsort);
    # each rank takes its own slice of the global sequence
    src = S.ArgsortSource(p, 42, 2 ** 62)
    total_steps = args.warmup + args.steps
    all_perms = src.take(total_steps * B * world).astype(np.int32).reshape(total_steps, world, B, p)
    my_perms = np.ascontiguousarray(all_perms[:, rank])
    n_ord = 2 * B

    def barrier():
        eng.synchronize()
        if world > 1 or rehearse:
            dist.barrier()
        torch.cuda.synchronize()

    def step(k):
        eng.run_batch(my_perms[k], True, want_lifts=False, accumulate=True)
        comm.allreduce_pending(eng)
        eng.merge()

    # pass 1: the timed region proper (no events between the launches: an event record costs a
    # ~10 us bubble per kernel boundary, 34 boundaries per step)
    eng.profile(False)
    eng.reset_stats()
    for k in range(args.warmup):
        step(k)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.warmup, total_steps):
        step(k)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1 or rehearse:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_seen, mean, _ = eng.stats(want_cov=False)
    # pass 2: the same K steps again with a HIP-event pair around every launch on the engine's
    # stream -> per-kernel durations for the roofline figures
    eng.profile(True)
    eng.profile_reset()
    for k in range(args.warmup, total_steps):
        step(k)
    barrier()
    prof = eng.profile_read()
    eng.profile(False)

    out = None
    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        value = world * n_ord * args.steps / elapsed
        per_class = {k: {"ms_per_step": v[0] / args.steps, "launches_per_step": v[1] / args.steps,
                         "avg_launch_ms": (v[0] / v[1]) if v[1] else None}
                     for k, v in prof.items() if v[1]}
        mfma_classes = [k for k in ("strip", "chol_panel", "chol_diag") if k in per_class]
        dom = max(mfma_classes, key=lambda k: per_class[k]["ms_per_step"])
        lpb = per_class[dom]["launches_per_step"]
        flops = algorithmic_flops(dom, p, n_ord, eng.tri, lpb)
        ach = flops / (per_class[dom]["avg_launch_ms"] * 1e-3) / 1e12
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if rec.get("p") == p and rec.get("batch_size") == B:
                    traffic = rec.get("hbm_bytes_per_launch", {}).get(dom)
            except Exception:
                traffic = None
        if args.dtype == "f32":
            traffic = None   # the committed PMC summary was collected on the f64 path
        roofline = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": peak_tf,
                    "unit": "TFLOP/s", "frac": ach / peak_tf, "traffic": traffic,
                    "algorithmic_flops_per_launch": flops, "avg_launch_ms": per_class[dom]["avg_launch_ms"],
                    "traffic_gbps": (traffic / (per_class[dom]["avg_launch_ms"] * 1e-3) / 1e9) if traffic else None,
                    "note": "peak = vendor fp64 matrix/vector figure; with random operands streaming from HBM the fp64 "
                            "matrix pipe sustains 48-54 TFLOP/s in the isolated k-loop of these kernels "
                            "(tools/mfma_bench5.hip: power bound, data dependent), which is the practical ceiling"}
        g_bytes = 2.0 * p * p * esz * n_ord         # 2 p^2 s bytes per ordering
        g_ach = g_bytes / (per_class["gather"]["avg_launch_ms"] * 1e-3) / 1e9
        gram_flops = 2.0 * rows * (p + 1) * (p + 2) / 2   # (N + M)(p + 1)(p + 2), both sides -> per launch
        gram_ach = gram_flops / ((gram_ms / max(gram_n, 1)) * 1e-3) / 1e12
        out = {
            "metric": "orderings_per_sec", "value": value, "unit": "orderings/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"C3 p={p} N=M={rows} method=argsort batch_size={B} antithetical "
                                   f"(={n_ord} orderings/step/GPU) {'fp64' if args.dtype == 'f64' else 'fp32 work / fp64 accumulation'}",
                       "p": p, "N": rows, "M": rows,
                       "batch_size": B, "orderings_per_step_per_gpu": n_ord, "path": "tri" if eng.tri else "rect"},
            "roofline": roofline,
            "roofline_gather": {"bound": "hbm", "achieved": g_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": g_ach / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": g_bytes},
            "roofline_gram": {"bound": "mfma", "achieved": gram_ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": gram_ach / FP64_PEAK_TFLOPS, "algorithmic_flops_per_launch": gram_flops,
                              "avg_launch_ms": gram_ms / max(gram_n, 1)},
            "kernels": per_class,
            "reduction_ms": reduce_ms,
            "check": {"samples": int(n_seen), "sum_attribution": float(mean.sum())},
            "timing_note": "value/ms_per_step: K steps without events; kernels/roofline: the same K steps "
                           "repeated with a HIP-event pair around every launch on the engine's stream",
        }

    # ---- time to tolerance (SURVEY 8d ii): the reference's own stopping rule, sharded over the ranks
    if not args.no_ttt:
        barrier()
        t0 = time.perf_counter()
        attribution, _, total_err, err_hist, _, n_stop = run_estimator(
            eng, p, max_samples=B * 128, batch_size=B, tolerance=1e-2, seed=42, perms=None, antithetical=True,
            return_attribution_history=False, method="argsort", error_estimator="reference", comm=comm)
        barrier()
        ttt = time.perf_counter() - t0
        if out is not None:
            out["time_to_tolerance"] = {"seconds_sampling_loop": ttt, "seconds_incl_reduction": ttt + reduce_ms * 1e-3,
                                        "samples_at_stop": int(n_stop), "overall_error": float(total_err),
                                        "tolerance": 1e-2, "error_estimator": "reference (host numpy)",
                                        "h2d_included": False}
        # same run with the statistically equivalent low-rank estimator (no p x p factorisation)
        barrier()
        t0 = time.perf_counter()
        _, _, total_err, _, _, n_stop = run_estimator(
            eng, p, max_samples=B * 128, batch_size=B, tolerance=1e-2, seed=42, perms=None, antithetical=True,
            return_attribution_history=False, method="argsort", error_estimator="lowrank", comm=comm)
        barrier()
        ttt = time.perf_counter() - t0
        if out is not None:
            out["time_to_tolerance_lowrank"] = {"seconds_sampling_loop": ttt,
                                                "seconds_incl_reduction": ttt + reduce_ms * 1e-3,
                                                "samples_at_stop": int(n_stop), "overall_error": float(total_err),
                                                "error_estimator": "lowrank (host numpy, O(1024 n p))"}
        # and with the same thin form evaluated on the GPU (lift vectors stay in HBM; with several ranks the
        # partial draws are summed by one extra all-reduce per check)
        for rep in range(2):   # first pass allocates the history / draws buffers
            barrier()
            t0 = time.perf_counter()
            _, _, total_err, _, _, n_stop = run_estimator(
                eng, p, max_samples=B * 128, batch_size=B, tolerance=1e-2, seed=42, perms=None, antithetical=True,
                return_attribution_history=False, method="argsort", error_estimator="device", comm=comm)
            barrier()
            ttt = time.perf_counter() - t0
        if out is not None:
            out["time_to_tolerance_device"] = {"seconds_sampling_loop": ttt,
                                               "seconds_incl_reduction": ttt + reduce_ms * 1e-3,
                                               "samples_at_stop": int(n_stop), "overall_error": float(total_err),
                                               "error_estimator": "device (thin form, HIP kernels)"}

    if out is not None and world == 1 and not args.no_cpu_baseline:
        G, g, H, h = eng.gram()
        out["cpu_baseline"] = cpu_baseline(G, g, H, h, eng.y_norm_sq, p)
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    if out is not None:
        print(json.dumps(out))
    eng.close()
    if world > 1 or rehearse:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
