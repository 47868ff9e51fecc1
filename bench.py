#!/usr/bin/env python3
"""Benchmark of the LS-SPA hot path on MI355X (driver contract: one JSON line on rank 0).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N --steps K --warmup W          # N > 1: starts its own N ranks (one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W   # ... or is started as one of them

Started plainly with --gpus N > 1 (no WORLD_SIZE in the environment) the process becomes a LAUNCHER: before it
imports PyTorch or touches HIP it starts N copies of itself as child processes with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR = 127.0.0.1 / MASTER_PORT set, relays rank 0's single JSON line, ends the others if one fails and exits
with the first non-zero status.  It never execs and never initialises the GPU itself.

Default workload = BASELINE.json configs[2] (C3): p = 1000 features, N = M = 100000 rows of the synthetic Gaussian
data of BASELINE.md section 3 (default_rng(0)), method='argsort' (Sobol, seed 42), batch_size = 128 antithetical
samples = 256 orderings per step, fp64.  Other configs:  --p 100 --rows 10000 (C2),
--p 5000 --rows 200000 --dtype f32 (C5: float32 data and per-ordering work, reg = 1e-2).

A step = one batch through the whole per-batch path: ordering upload -> permuted gather -> blocked Cholesky ->
V^T tiles inside the Cholesky's panel launches -> lifts -> batch moments -> (all-reduce over the ranks: RCCL through the C ABI) -> merge.  The data
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
    ap.add_argument("--warmup", type=int, default=5)
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
                         "step by step (what ls_spa(lookahead=k) does); 0 = auto: up to p = 800 what ls_spa(lookahead='auto') "
                         "takes (at 128 samples a step: 16 steps for p <= 127, 8 up to p = 250, 4 up to 800), beyond 4 when a "
                         "rank's step has <= 32 samples, else 1")
    ap.add_argument("--lanes", type=int, choices=(0, 1, 2), default=0,
                    help="batches in flight on the engine (lsspa_set_lanes): 2 = the next step's kernels start when this "
                         "step's are half way (two workspaces, two streams; statistics stay in batch order); 0 = auto: 2 on "
                         "the general path (p > 127), 1 for the fused small-p kernel")
    ap.add_argument("--split", type=int, default=0,
                    help="with two lanes: a step's batch goes to the engine as this many consecutive sub-batches (each its own "
                         "launch sequence, alternating lanes): a shorter pipeline, so less fill / drain inside a K-step region; "
                         "0 = auto")
    ap.add_argument("--flags", type=int, default=0, help="developer switches of the engine (include/lsspa.h, lsspa_set_flags)")
    ap.add_argument("--no-probe", action="store_true", help="skip the strong-scaling probe (clean rocprof averages)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 3 s sustained-rate region")
    ap.add_argument("--no-full-pass", action="store_true",
                    help="skip the third pass (full-batch launches on one lane): profiler runs, where every launch of a "
                         "kernel should have the timed region's size")
    ap.add_argument("--no-ttt", action="store_true", help="skip the time-to-tolerance runs")
    ap.add_argument("--no-full-run", action="store_true", help="skip the many-check run of the public call (full_run)")
    ap.add_argument("--full-run-batches", type=int, default=0,
                    help="checks of the full_run leg: ls_spa(method='argsort', batch_size=B, num_batches=this, tolerance=0); "
                         "0 = auto: 128 (BASELINE config 4's num_batches) up to p = 2000, 64 (the reference's default "
                         "max_samples = 8192 at batch 128) for p <= 127, 16 beyond p = 2000")
    ap.add_argument("--data", choices=("gaussian", "correlated"), default="gaussian",
                    help="gaussian: BASELINE.md section 3 (default_rng(0)), the data the metric is quoted on; correlated: "
                         "the reference's own generator (experiments/ground_truth_medium.py:74-106, seed 42) at the same "
                         "shape -- the harder, secondary workload of SURVEY.md 8(d)")
    ap.add_argument("--no-correlated-leg", action="store_true",
                    help="skip the time-to-tolerance run on the reference's correlated data (one GPU, default data only)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment
# ---------------------------------------------------------------------------------------------------------------
def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """Start n ranks of this script (one per GPU) and relay rank 0's stdout.  Runs BEFORE torch is imported or HIP
    is touched in this process; the children are plain subprocesses (no exec).  Returns the exit status."""
    import subprocess
    import threading
    port = _free_port()
    # the native communicator's own rendezvous listens on MASTER_PORT + 29 (ls_spa._rccl): keep that in range
    while port + 64 > 65535:
        port = _free_port()
    procs = []
    # LSSPA_BENCH_REHEARSE_WORLD=N (with --gpus N): the N ranks all open GPU 0 -- the real multi-rank code of this file
    # (dealing, agreement on the transport, strong-scaling region, max / min over ranks, sharded time-to-tolerance legs)
    # on a one-GPU box.  RCCL refuses several ranks on one device, so the process group is gloo and the engine's device
    # buffers are staged through the host around the all-reduce (ls_spa._dist.TorchComm).
    one_gpu = os.environ.get("LSSPA_BENCH_REHEARSE_WORLD") == str(n)
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0" if one_gpu else str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LSSPA_BENCH_LAUNCHED="1")
        if one_gpu:
            env["LSSPA_BENCH_ONE_GPU"] = "1"
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno()))
    lines = []

    def pump():     # rank 0's stdout is the contract's one JSON line
        for raw in procs[0].stdout:
            lines.append(raw.decode(errors="replace"))

    t = threading.Thread(target=pump, daemon=True)
    t.start()
    status, alive = 0, set(range(n))
    while alive:
        for r in sorted(alive):
            rc = procs[r].poll()
            if rc is None:
                continue
            alive.discard(r)
            if rc != 0 and status == 0:
                status = rc if rc > 0 else 1
                sys.stderr.write(f"[bench launcher] rank {r} exited with status {rc}; ending the other ranks\n")
                for q in alive:          # a rank that died leaves the others waiting in a collective
                    procs[q].terminate()
        if alive:
            time.sleep(0.05)
    for q in range(n):
        try:
            procs[q].wait(timeout=30)
        except Exception:
            procs[q].kill()
    t.join(timeout=5)
    sys.stdout.write("".join(lines))
    sys.stdout.flush()
    return status


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


def correlated_data(p, rows, dtype):
    """The reference's own "Medium" generator (experiments/ground_truth_medium.py:74-106; conditioning 20, signal to
    noise 5, centred by the training means) at the benchmark's shape, seed 42 -- the product's workloads.correlated,
    which tests/golden/corr_data_p100.npz pins to the reference's gen_data body."""
    from ls_spa.workloads import correlated
    Xa, Xe, ya, ye, _, _ = correlated(np.random.default_rng(42), p, rows, rows)
    dt = np.float32 if dtype == "f32" else np.float64
    return tuple(np.ascontiguousarray(a, dtype=dt) for a in (Xa, Xe, ya, ye))


def algorithmic_flops(kclass, p, n_ord, tri, launches_per_batch, vt=False):
    """Useful flops of one launch of a kernel class (element granularity, no padding, no
    redundant tile work), so that 'achieved' cannot be inflated by wasted arithmetic."""
    n_mats = n_ord * (2 if tri else 1)
    if kclass == "strip":          # triangular-triangular solve V = L^-1 L_t (p^3/3)
        per = p ** 3 / 3.0 if tri else float(p) ** 3
        return per * n_ord / launches_per_batch
    if kclass == "chol_panel":     # the whole Cholesky, p^3/3 per matrix: the panel launches also factor the
        # diagonal blocks (all but block 0); vt (tri mode since round 4): they also compute V^T = L_t^T L^-T, p^3/3 per
        # ordering -- SURVEY 8d's "~p^3 per ordering" in one kernel class
        return ((p ** 3 / 3.0) * n_mats + ((p ** 3 / 3.0) * n_ord if vt else 0.0)) / launches_per_batch
    if kclass == "chol_diag":      # stand-alone launch: block 0 only (factor + inverse)
        return (64 ** 3 / 3.0 * 2) * n_mats / launches_per_batch
    if kclass == "small_p":        # fused small-p kernel: the whole per-ordering work, ~p^3
        return float(p) ** 3 * n_ord / launches_per_batch
    return 0.0


def read_gpu_sysfs(index):
    """{'sclk_mhz', 'power_w'} of HIP device `index` from the amdgpu driver's sysfs files (hwmon freq1_input /
    power1_average, pp_dpm_sclk as a fallback): no child process, nothing executed."""
    import glob
    base = None
    try:
        import torch
        pr = torch.cuda.get_device_properties(index)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        if os.path.isdir(f"/sys/bus/pci/devices/{bdf}"):
            base = f"/sys/bus/pci/devices/{bdf}"
    except Exception:
        base = None
    if base is None:
        cards = []
        for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            try:
                if open(os.path.join(dev, "vendor")).read().strip() == "0x1002" and glob.glob(dev + "/hwmon/hwmon*"):
                    cards.append(dev)
            except OSError:
                pass
        base = cards[index]
    out = {}
    for hw in glob.glob(base + "/hwmon/hwmon*"):
        for name in ("power1_average", "power1_input"):
            f = os.path.join(hw, name)
            if "power_w" not in out and os.path.exists(f):
                out["power_w"] = int(open(f).read().strip()) / 1e6
        f = os.path.join(hw, "freq1_input")
        if os.path.exists(f):
            out["sclk_mhz"] = int(open(f).read().strip()) / 1e6
    if "sclk_mhz" not in out and os.path.exists(base + "/pp_dpm_sclk"):
        for ln in open(base + "/pp_dpm_sclk"):
            if "*" in ln:
                out["sclk_mhz"] = float(ln.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
    return out


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


def _keep_evidence():
    """Any abnormal end of a rank (a fatal signal inside the HIP runtime or a kernel fault that aborts the process)
    leaves a Python-level traceback of every thread in gpurun_out/bench_fault_rank<r>.log, beside whatever the
    runtime printed on stderr (the advisor's round-3 finding: the one abort of round 2 left no evidence)."""
    import faulthandler
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        fh = open(os.path.join(d, f"bench_fault_rank{os.environ.get('RANK', '0')}.log"), "w")
        faulthandler.enable(file=fh, all_threads=True)
        return fh
    except Exception:
        faulthandler.enable()
        return None


def main():
    args = parse()
    if (args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1
            and os.environ.get("LSSPA_BENCH_LAUNCHED") != "1"):
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if os.environ.get("LSSPA_BENCH_STUB") == "1":
        # test hook (tests/test_host_logic.py): a rank reports what it was started with and stops BEFORE torch or HIP
        env = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        if env["RANK"] is not None and os.environ.get("LSSPA_BENCH_STUB_FAIL_RANK") == env["RANK"]:
            raise SystemExit(3)
        import tempfile
        with open(os.path.join(os.environ.get("LSSPA_BENCH_STUB_DIR", tempfile.gettempdir()),
                               f"bench_stub_rank{env['RANK']}.json"), "w") as fh:
            json.dump(dict(env, torch_imported="torch" in sys.modules, gpus=args.gpus), fh)
        if env["RANK"] == "0":
            print(json.dumps({"stub": True, "world": int(env["WORLD_SIZE"])}))
        return
    _fault_log = _keep_evidence()     # (kept referenced: the handler writes to this file object)
    import torch
    import torch.distributed as dist
    from ls_spa import ls_spa
    from ls_spa._engine import HipEngine
    from ls_spa._driver import run_estimator, _Comm, SMALL_P_MAX, auto_lookahead

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # developer switch: run the multi-rank code path (communicator, forced collective) in a world of one, to
    # rehearse on a single GPU exactly what the N > 1 launch executes
    rehearse = world == 1 and os.environ.get("LSSPA_BENCH_REHEARSE_DIST") == "1"
    multi = world > 1 or rehearse
    # several ranks on ONE GPU (the launcher's LSSPA_BENCH_REHEARSE_WORLD): gloo group, moments staged through the host
    one_gpu = world > 1 and os.environ.get("LSSPA_BENCH_ONE_GPU") == "1"
    coll_dev = torch.device("cpu") if one_gpu else dev      # where the contract's own small collectives live
    fail_rank = os.environ.get("LSSPA_BENCH_FAIL_RANK")     # test hook: this rank dies after the warm-up steps

    p, rows, B = args.p, args.rows, args.batch_size
    if args.lanes == 0:
        args.lanes = 2 if p > SMALL_P_MAX else 1
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
            elif one_gpu:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)
            warm = torch.zeros(1, device=coll_dev)
            dist.all_reduce(warm)
            torch.cuda.synchronize()
            native_ok = 0
            if args.collective == "native" and not one_gpu:
                try:
                    from ls_spa._rccl import NativeComm
                    comm = NativeComm.from_env(force_collective=rehearse)
                    comm.bind(eng)
                    native_ok = 1
                except Exception as exc:   # keep the measurement alive: fall back to the torch transport, say so
                    sys.stderr.write(f"[bench] rank {rank}: native communicator failed ({exc})\n")
                # the ranks agree on the outcome (the torch group is up): one rank on torch.distributed beside others on
                # the engine's RCCL communicator would never meet them in a collective
                flag = torch.tensor([native_ok], dtype=torch.int32, device=coll_dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 0:
                    if native_ok and hasattr(comm, "close"):
                        comm.close()
                    comm = None
                    if rank == 0:
                        sys.stderr.write("[bench] a rank could not make the native communicator: ALL ranks use "
                                         "torch.distributed\n")
                else:
                    collective = "rccl via the C ABI (lsspa_stats_allreduce, engine stream)"
            if args.collective == "torch" or comm is None or one_gpu:
                from ls_spa._dist import TorchComm
                comm = TorchComm(force_collective=rehearse)
                collective = ("torch.distributed gloo, device buffers staged through the host (several ranks rehearsed on "
                              "one GPU: RCCL refuses that)" if one_gpu else
                              "torch.distributed nccl (host-synchronised hand-off between the engine's and torch's stream)")
    finally:
        if multi:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    # the benchmark data (same on every rank), generated on the host as BASELINE.md prescribes, then moved to HBM
    t0 = time.perf_counter()
    host = baseline_data(p, rows, args.dtype) if args.data == "gaussian" else correlated_data(p, rows, args.dtype)
    gen_s = time.perf_counter() - t0
    tdt = torch.float32 if args.dtype == "f32" else torch.float64
    dXa, dXe, dya, dye = (torch.from_numpy(a).to(dev) for a in host)
    assert dXa.dtype == tdt
    torch.cuda.synchronize()

    if args.dtype == "f32":
        eng.set_precision("float32")
    if args.flags:
        eng.set_flags(args.flags)
    peak_tf = FP32_PEAK_TFLOPS if args.dtype == "f32" else FP64_PEAK_TFLOPS
    esz = 4 if args.dtype == "f32" else 8
    eng.profile(True)
    t0 = time.perf_counter()
    eng.load_device_data(dXa.data_ptr(), p, dya.data_ptr(), rows, dXe.data_ptr(), p, dye.data_ptr(), rows, p, reg,
                         f32=args.dtype == "f32")
    eng.synchronize()
    reduce_ms = 1e3 * (time.perf_counter() - t0)
    # the Gram kernels again on the resident data: the first call is also the process's first GPU work (cold clocks,
    # code object load), which a roofline figure for the kernel should not carry, and two 2 ms launches after seconds
    # of host work do not bring the chip to the clock it holds under load either -- three more calls, the last one is
    # the one reported
    for _ in range(3):
        eng.profile_reset()
        t0 = time.perf_counter()
        eng.load_device_data(dXa.data_ptr(), p, dya.data_ptr(), rows, dXe.data_ptr(), p, dye.data_ptr(), rows, p, reg,
                             f32=args.dtype == "f32")
        eng.synchronize()
        reduce_warm_ms = 1e3 * (time.perf_counter() - t0)
    gram_ms, gram_n = eng.profile_read()["gram"]
    # theta and R^2 of the full model, as the public call computes them before its loop: from here on every batch of the
    # engine is checked on the device against "the lifts of a sample sum to R^2" (LSSPA_INFO_SUM, include/lsspa.h)
    _, r2_full, _ = eng.full_fit()
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
    D = args.lookahead if args.lookahead > 0 else (auto_lookahead(p, B_rank) if p <= 800 else (4 if B_rank <= 32 else 1))

    class Steps:
        """step(k) = one batch of b_rank samples into the statistics.  With d > 1 the kernels of d consecutive steps
        are launched together (one gather / factorisation / solve sequence over d * b_rank samples) and each step
        then folds its own b_rank lift vectors into the pending buffer, all-reduces and merges -- the reference's
        per-batch order, a fuller GPU.  Groups start at the first step of a region."""

        def __init__(self, perms, b_rank, d, split=1):
            self.perms, self.b_rank, self.d, self.split = perms, b_rank, d, split
            self.tickets, self.base, self.end = {}, 0, 0

        def region(self, k0, k1):
            for tk in self.tickets.values():      # a group launched ahead across a region boundary
                eng.discard_batch(tk)
            self.tickets, self.base, self.end = {}, k0, k1

        def launch(self, g):
            lo = self.base + g * self.d
            if lo < self.end and g not in self.tickets:
                self.tickets[g] = eng.launch_batch(self.perms[lo:min(lo + self.d, self.end)].reshape(-1, p), True)

        def __call__(self, k):
            acc = True if multi else 2      # one rank: fold and merge at once (lsspa_lift_collect, accumulate = 2)
            if self.d == 1 and self.split > 1:
                pk, h = self.perms[k], -(-self.b_rank // self.split)
                for s0 in range(0, self.b_rank, h):
                    eng.run_batch(pk[s0:s0 + h], True, want_lifts=False, accumulate=acc)
            elif self.d == 1:
                eng.run_batch(self.perms[k], True, want_lifts=False, accumulate=acc)
            else:
                g, j = divmod(k - self.base, self.d)
                if j == 0:
                    self.launch(g)
                    if eng.lanes == 2:
                        self.launch(g + 1)       # the second lane holds the next group: its upload and kernels run
                                                 # while this group is accumulated step by step
                if not multi:
                    # one rank: the group's steps are folded by ONE library call at its first step -- every step's
                    # batch still merged by itself, in order (lsspa_lift_collect_chunks; at p <= 128 one launch)
                    if j == 0:
                        n_g = min(self.d, self.end - (self.base + g * self.d))
                        eng.collect_chunks(self.tickets.pop(g), 0, self.b_rank, n_g, accumulate=2)
                else:
                    eng.collect_batch(self.tickets[g], want_lifts=False, accumulate=acc, first=j * self.b_rank,
                                      count=self.b_rank)
                    if j == self.d - 1 or k == self.end - 1:
                        del self.tickets[g]
            if multi:
                comm.allreduce_pending(eng)
                eng.merge()

    def timed_region(step, n_warm, n_total):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; returns this rank's
        seconds, the max and the min over the ranks."""
        eng.reset_stats()
        if step.d > 1:
            # the workspace of a whole look-ahead group before anything is timed (the warm-up steps may be fewer than a
            # group: a first full group inside the timed region would allocate there)
            eng.discard_batch(eng.launch_batch(step.perms[0:step.d].reshape(-1, p), True))
        step.region(0, n_warm)
        for k in range(n_warm):
            step(k)
        if fail_rank is not None and int(fail_rank) == rank:
            sys.stderr.write(f"[bench] rank {rank}: LSSPA_BENCH_FAIL_RANK set, leaving with status 3\n")
            os._exit(3)
        barrier()
        step.region(n_warm, n_total)
        t0 = time.perf_counter()
        for k in range(n_warm, n_total):
            step(k)
        barrier()
        mine = time.perf_counter() - t0
        hi = lo = mine
        if multi:
            t = torch.tensor([mine, -mine], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            hi, lo = float(t[0].item()), -float(t[1].item())
        return mine, hi, lo

    # auto: two half-batches per step on the two lanes once a half still fills the chip (measured at C3, 20 steps: one
    # launch sequence per step 6.37 ms, two 6.18, three 6.41, four 6.44; one lane 6.51)
    SPLIT = args.split if args.split > 0 else (2 if (args.lanes == 2 and D == 1 and B_rank >= 64) else 1)
    step = Steps(my_perms, B_rank, D, SPLIT)

    eng.set_lanes(args.lanes)
    # pass 1: the timed region proper (no events between the launches: an event record costs a
    # ~10 us bubble per kernel boundary)
    eng.profile(False)
    _, elapsed, elapsed_min = timed_region(step, args.warmup, total_steps)
    n_seen, mean, _ = eng.stats(want_cov=False)
    # sustained rate (orderings/s "sustained over the sampling loop"): the headline region is K steps --
    # 0.13 s at C3 -- and the chip needs ~0.4 s of back-to-back launches to settle on the clock it holds under load.  The
    # same step function over >= SUSTAIN_S seconds (the step count follows from the headline time, identical on every
    # rank), after the headline region so that `value` keeps its definition; the orderings are the region's, cyclically.
    # Rank 0 reads clock and power from the driver's sysfs files about a second into the region (helper thread; None if absent).
    sustained = None
    if not args.no_sustained:
        SUSTAIN_S = 3.0
        n_sus = max(args.steps, int(np.ceil(SUSTAIN_S / max(elapsed / args.steps, 1e-6))))
        n_sus = -(-n_sus // D) * D          # whole look-ahead groups

        class Cyclic:
            """perms[k] for k beyond the generated steps: the timed region's orderings again"""
            def __init__(self, base, lo, hi):
                self.base, self.lo, self.n = base, lo, hi - lo

            def __getitem__(self, k):
                if isinstance(k, slice):
                    return self.base[[self.lo + (i % self.n) for i in range(k.start, k.stop)]]
                return self.base[self.lo + (k % self.n)]

        smi = {}

        def ask_smi():
            # clock and power from sysfs, in-process (round 5): up to round 4 this started `rocm-smi`, a
            # `#!/usr/bin/env python3` script -- under rocprofv3 the child inherits the profiler's preload, which
            # initialises the GPU in `env`, and `env` then execs: the exec-after-GPU-init this pool forbids
            time.sleep(1.0)
            try:
                smi.update(read_gpu_sysfs(local))
            except Exception as exc:
                smi["error"] = repr(exc)

        import threading
        th = threading.Thread(target=ask_smi, daemon=True) if rank == 0 else None
        sus_step = Steps(Cyclic(my_perms, args.warmup, total_steps), B_rank, D, SPLIT)
        barrier()
        if th is not None:
            th.start()
        _, el_sus, el_sus_min = timed_region(sus_step, 0, n_sus)
        if th is not None:
            th.join(timeout=15)
        sustained = {"steps": n_sus, "seconds": el_sus, "ms_per_step": 1e3 * el_sus / n_sus,
                     "orderings_per_s": world * n_ord * n_sus / el_sus, "ms_per_step_min_rank": 1e3 * el_sus_min / n_sus,
                     "sclk_mhz": smi.get("sclk_mhz"), "power_w": smi.get("power_w"), "sysfs_error": smi.get("error"),
                     "note": "the same step function as the headline region over >= 3 s (max over ranks); clock and power "
                             "as the amdgpu driver's sysfs files (hwmon) report them about one second into the region"}
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
    # pass 3 (only when a step goes to the engine in several launch sequences): the same K steps once more as ONE launch
    # sequence per step on one lane -- the dominant kernel at its full batch size, alone on the GPU.  This is the shape
    # `rocprofv3 --stats -- python3 bench.py --lanes 1` shows (profiles/r0N_c3_one_lane_kernel_stats.csv): under the
    # default command the profiler's averages mix the timed region's launches, which overlap across the two streams and
    # are stretched by it, with this pass's.
    prof_full = None
    if SPLIT > 1 and not args.no_full_pass:
        eng.profile_reset()
        full = Steps(my_perms, B_rank, D, 1)
        full.region(args.warmup, total_steps)
        for k in range(args.warmup, total_steps):
            full(k)
        barrier()
        prof_full = eng.profile_read()
    eng.profile(False)

    # several ranks, weak line: the same K steps once more in BASELINE config 4's semantics -- the global batch of
    # batch_size samples dealt over the ranks (batch_size / N each), kernels of `lookahead` steps launched together
    strong = None
    if world > 1 and args.scaling == "weak" and B % world == 0:
        b_s = B // world
        d_s = args.lookahead if args.lookahead > 0 else (8 if p <= SMALL_P_MAX else max(1, min(8, 64 // b_s)))
        src_s = S.ArgsortSource(p, 42, 2 ** 62)
        perms_s = np.ascontiguousarray(src_s.take(total_steps * B).astype(np.int32)
                                       .reshape(total_steps, B, p)[:, rank::world])
        eng.set_lanes(args.lanes)
        _, el_s, el_s_min = timed_region(Steps(perms_s, b_s, d_s), args.warmup, total_steps)
        eng.set_lanes(1)
        strong = {"scaling": "strong", "global_batch": B, "samples_per_rank_per_step": b_s, "lookahead": d_s,
                  "value": 2 * B * args.steps / el_s, "unit": "orderings/s", "ms_per_step": 1e3 * el_s / args.steps,
                  "ms_per_step_min_rank": 1e3 * el_s_min / args.steps,
                  "note": "BASELINE config 4's partitioning: every step's batch_size samples are dealt round-robin over "
                          "the ranks, one all-reduce of the moments per step; value = global orderings / max-over-ranks time"}
        del perms_s

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
        vt = bool(eng.tri) and "strip" not in per_class      # V^T by the panel launches' X tiles (no strip launch)
        flops = algorithmic_flops(dom, p, n_ord, eng.tri, lpb, vt)
        ach = flops / (per_class[dom]["avg_launch_ms"] * 1e-3) / 1e12
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                for rec in json.load(open(pmc)).get("runs", []):
                    if (rec.get("p"), rec.get("batch_size"), rec.get("dtype")) == (p, B_rank, args.dtype):
                        traffic = rec.get("hbm_bytes_per_launch", {}).get(dom)
                        # the counters were collected with the launch sizes of the default pipeline: per launch of THIS
                        # run = per step there / launches per step here
                        rec_lps = rec.get("launches_per_step", {}).get(dom)
                        if traffic and rec_lps:
                            traffic = traffic * rec_lps / lpb
                        traffic_src = ("profiles/pmc_traffic.json (static): FETCH_SIZE x 2 + WRITE_SIZE from separate "
                                       "rocprofv3 --pmc passes of bench.py, collected when the profiles were last refreshed "
                                       "-- NOT measured in this run")
            except Exception:
                traffic = None
        roofline = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": peak_tf,
                    "unit": "TFLOP/s", "frac": ach / peak_tf, "traffic": traffic, "traffic_source": traffic_src,
                    "algorithmic_flops_per_launch": flops, "avg_launch_ms": per_class[dom]["avg_launch_ms"],
                    "launches_per_step": lpb,
                    "traffic_gbps": (traffic / (per_class[dom]["avg_launch_ms"] * 1e-3) / 1e9) if traffic else None,
                    "note": "peak = vendor fp64 (fp32) matrix figure, which the chip does sustain: a register-resident loop of "
                            "v_mfma_f64_16x16x4 runs at 77.6 TFLOP/s on all 256 CUs for as long as asked "
                            "(profiles/r04_mfma_cap_probe.log); the isolated k-loop of these kernels sustains ~60 TFLOP/s fp64 "
                            "fed from HBM and ~67 fed from the caches, the vendor library's GEMM 69-71 at 8192^3 and 59-67 "
                            "batched at 1024^3 (profiles/r02_kloop_ceiling.log)"}
        if prof_full is not None and prof_full.get(dom, (0, 0))[1]:
            ms_f, n_f = prof_full[dom]
            lpb_f = n_f / args.steps
            fl_f = algorithmic_flops(dom, p, n_ord, eng.tri, lpb_f, vt)
            roofline["full_batch_one_lane"] = {
                "avg_launch_ms": ms_f / n_f, "launches_per_step": lpb_f, "algorithmic_flops_per_launch": fl_f,
                "achieved": fl_f / (ms_f / n_f * 1e-3) / 1e12, "frac": fl_f / (ms_f / n_f * 1e-3) / 1e12 / peak_tf,
                "ms_per_step": ms_f / args.steps,
                "note": "the same kernel with a step's batch as ONE launch sequence (what --lanes 1 runs and what rocprofv3 "
                        "--stats of `bench.py --lanes 1` averages); the figures above are for the half-batch launches of "
                        "the default pipeline, each alone on the GPU in the event pass"}
        out = {
            "metric": "orderings_per_sec", "value": value, "unit": "orderings/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic gaussian" if args.data == "gaussian" else "synthetic correlated",
            "config": {"workload": f"{label} p={p} N=M={rows} reg={reg:g} method=argsort batch_size={B} antithetical "
                                   f"(={n_ord} orderings/step/GPU) "
                                   f"{'fp64' if args.dtype == 'f64' else 'fp32 data and per-ordering work / fp64 accumulation'}",
                       "p": p, "N": rows, "M": rows, "reg": reg, "batch_size": B, "global_batch": B * world if
                       args.scaling == "weak" else B, "orderings_per_step_per_gpu": n_ord,
                       "path": "tri" if eng.tri else "rect", "collective": collective, "lanes": args.lanes, "lookahead": D,
                       "launch_sequences_per_step": SPLIT,
                       "flags": args.flags,
                       "data_generator": ("BASELINE.md section 3: default_rng(0) on the host, moved to HBM before timing"
                                          if args.data == "gaussian" else
                                          "the reference's gen_data (experiments/ground_truth_medium.py:74-106), seed 42, "
                                          "generated on the host, moved to HBM before timing")},
            "roofline": roofline,
            "kernels": per_class,
            "reduction_ms": reduce_ms, "reduction_ms_warm_call": reduce_warm_ms,
            "host_data_generation_s": gen_s,
            # engine_info: LSSPA_INFO_* bits raised by any launch of this process so far (1: a pivot was not positive,
            # 4: a hand-over inside a panel launch timed out, 8: a sample's lifts did not sum to R^2) -- 0 on the benchmark data
            # sum_deviation_max: largest |sum of a sample's lifts - R^2| over the batches since the last reset (checked on
            # the device for every batch; beyond 1e-9 / 1e-4 (fp32 work) it raises bit 8)
            "check": {"samples": int(n_seen), "sum_attribution": float(mean.sum()), "r_squared": float(r2_full),
                      "engine_info": int(eng.info()), "sum_deviation_max": float(eng.sum_deviation())},
            "ms_per_step_min_rank": 1e3 * elapsed_min / args.steps,
            "sustained": sustained,
            "timing_note": "value/ms_per_step: K steps without events (two lanes: launch sequences overlap across two "
                           "streams); kernels/roofline: the same K steps repeated on ONE lane with a HIP-event pair around "
                           "every launch on the engine's stream, i.e. each kernel alone on the GPU",
        }
        if "gather" in per_class:
            # 2 p^2 s bytes per ordering; a step's orderings may go in several launch sequences
            g_bytes = 2.0 * p * p * esz * n_ord / max(per_class["gather"]["launches_per_step"], 1.0)
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
                                        "reduction, fourth call on the resident data; fp64 accumulation in both data types"}
        if probe is not None:
            probe["per_ordering_throughput_vs_full_step"] = probe["orderings_per_s"] / value
            out["strong_scaling_probe"] = probe
        if "roofline_gather" in out and os.path.exists(pmc):
            try:
                for rec in json.load(open(pmc)).get("runs", []):
                    if (rec.get("p"), rec.get("batch_size"), rec.get("dtype")) == (p, B_rank, args.dtype):
                        g_tr = rec.get("hbm_bytes_per_launch", {}).get("gather")
                        g_lps = rec.get("launches_per_step", {}).get("gather")
                        if g_tr and g_lps:
                            g_tr = g_tr * g_lps / max(per_class["gather"]["launches_per_step"], 1.0)
                        if g_tr:
                            g_ms = per_class["gather"]["avg_launch_ms"]
                            out["roofline_gather"].update(
                                traffic=g_tr, physical_gbps=g_tr / (g_ms * 1e-3) / 1e9,
                                physical_frac=g_tr / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                traffic_source="profiles/pmc_traffic.json (static; FETCH_SIZE x 2 + WRITE_SIZE of the last "
                                               "profile refresh, not measured in this run)")
            except Exception:
                pass
        if "roofline_gather" in out:
            out["roofline_gather"]["note"] = (
                "algorithmic bytes = SURVEY 8d's 2 p^2 s per ordering; by the PMC counters (profiles/*pmc_summary*) the kernel "
                "moves fewer: the sources stay on chip and a pair shares a row -- at C3 3.58 GB per step against 4.10 "
                "algorithmic, i.e. the physical rate (physical_frac, where the counters of this shape are on file) is 0.87 of "
                "the algorithmic one")
        if multi:
            # did RCCL see N ranks: answered by RCCL (ncclCommCount through lsspa_comm_info), not by our bookkeeping
            rccl_world = None
            try:
                import ctypes as C
                rk, wd = C.c_int32(), C.c_int32()
                if eng._lib.lsspa_comm_info(eng._h, C.byref(rk), C.byref(wd)) == 0:
                    rccl_world = int(wd.value)
            except Exception:
                rccl_world = None
            out["rccl_world"] = rccl_world
            out["torch_world"] = int(dist.get_world_size())
            if "comm" in per_class:
                out["allreduce_ms_per_step"] = per_class["comm"]["ms_per_step"]
                out["allreduce_bytes"] = 8 * (1 + p + p * p) if p < 2048 else 8 * (1 + p + p * (p + 1) // 2)
            if strong is not None:
                out["strong_scaling"] = strong
            out["scaling_note"] = ("value is the WEAK line (batch_size samples per rank per step); strong_scaling is the same "
                                   "K steps with the global batch dealt over the ranks" if args.scaling == "weak" else
                                   "value is the STRONG line (the global batch dealt over the ranks)")

    # ---- time to tolerance (SURVEY 8d ii): the reference's own stopping rule, sharded over the ranks
    n_stop = n_checks = 0
    cov_at_stop = None
    legs = {}

    def guarded(name, fn):
        """One GPU: a failing leg must not cost the throughput line.  Several ranks: an exception on one rank would
        leave the others waiting in a collective -- let it end the process (the launcher then ends the job)."""
        if world > 1:
            return fn()
        try:
            return fn()
        except Exception as exc:
            legs[name] = {"error": repr(exc)}
            return None

    # several ranks: the legs below run on the sharded loop as well (round 5: BASELINE's metric is "permutations/sec +
    # time-to-1e-2-tolerance ... 1/2/4/8 GPUs", and a scaling run that skipped them reported half of it).  A leg that
    # stalls in a collective must not hang the job: a watchdog ends THIS rank with a non-zero status once the legs
    # have taken longer than LSSPA_BENCH_TTT_LIMIT_S seconds (default 180) -- the launcher (or torch.distributed.run)
    # then ends the others; nothing is re-executed.  LSSPA_BENCH_TTT_MULTI=0 switches the legs off with several ranks.
    ttt_on = not args.no_ttt and (world == 1 or os.environ.get("LSSPA_BENCH_TTT_MULTI", "1") != "0")
    if not args.no_ttt and not ttt_on:
        legs["time_to_tolerance"] = {"skipped": "several ranks and LSSPA_BENCH_TTT_MULTI=0"}
    watchdog = None
    if ttt_on and world > 1:
        import threading
        limit_s = float(os.environ.get("LSSPA_BENCH_TTT_LIMIT_S", "180"))

        def give_up():
            sys.stderr.write(f"[bench] rank {rank}: the time-to-tolerance legs took longer than {limit_s:.0f} s "
                             "(a stalled collective?): leaving with status 5\n")
            sys.stderr.flush()
            os._exit(5)

        watchdog = threading.Timer(limit_s, give_up)
        watchdog.daemon = True
        watchdog.start()
    if ttt_on:
        # the legs below run whole batches on ONE lane; the timed region (two lanes, half-batches) left the lane's
        # workspace at half that size -- grown here, untimed, as any second call of a process finds it (the cold start
        # of a process is what time_to_tolerance_e2e measures)
        eng.set_lanes(1)
        eng.run_batch(my_perms[0], True, want_lifts=False, accumulate=False)
        eng.synchronize()

        def leg(estimator, reps=1):
            best = None
            for _ in range(reps):
                barrier()
                t0 = time.perf_counter()
                tm = {}
                res = run_estimator(eng, p, max_samples=B * 128, batch_size=B, tolerance=1e-2, seed=42, perms=None,
                                    antithetical=True, return_attribution_history=False, method="argsort",
                                    error_estimator=estimator, comm=comm, timings=tm,
                                    lookahead="auto" if world > 1 else 1)
                barrier()
                dt = time.perf_counter() - t0
                best = (dt, res, tm) if best is None or dt < best[0] else best
            return best

        for name, estimator, reps in (("time_to_tolerance", "reference", 1), ("time_to_tolerance_lowrank", "lowrank", 1),
                                      ("time_to_tolerance_device", "device", 2)):
            got = guarded(name, lambda: leg(estimator, reps))
            if got is None:
                continue
            dt, res, tm = got
            legs[name] = {"seconds_sampling_loop": dt, "seconds_incl_reduction": dt + reduce_ms * 1e-3,
                          "samples_at_stop": int(res[5]), "checks": int(len(res[3])),
                          "overall_error": float(res[2]), "tolerance": 1e-2, "error_estimator": estimator,
                          "h2d_included": False, "host_seconds": {k: round(v, 6) for k, v in tm.items()},
                          "note": "sampling loop on the HBM-resident reduced problem of the timed region"}
            if name == "time_to_tolerance":
                n_stop, n_checks = int(res[5]), int(len(res[3]))
                if rank == 0 and world == 1:
                    cov_at_stop = eng.stats(want_cov=True)[2]

        # the whole public call on the host arrays: reduction over PCIe included.  One GPU (or the one-rank rehearsal)
        # only: with several ranks every call would make its own communicator -- the sharded sampling loop above is
        # the N > 1 figure
        def e2e_runs(data, max_batches=128):
            kw = dict(reg=reg, method="argsort", batch_size=B, num_batches=max_batches, tolerance=1e-2, seed=42,
                      device=local, precision="float32" if args.dtype == "f32" else "float64")
            e2e = {}
            for name, estimator in (("reference", "reference"), ("device", "device")):
                runs = []
                for rep in range(2):
                    c2 = None
                    if multi and collective.startswith("rccl"):
                        from ls_spa._rccl import NativeComm
                        c2 = NativeComm.from_env(force_collective=rehearse)
                        c2._port += 2 + rep + (10 if name == "device" else 0)
                    elif multi:
                        c2 = comm
                    barrier()
                    tm = {}
                    t0 = time.perf_counter()
                    r = ls_spa(*data, error_estimator=estimator, comm=c2, _timings=tm, **kw)
                    runs.append((time.perf_counter() - t0, tm))
                secs, tm = min(runs, key=lambda v: v[0])
                tm = dict(tm)
                tm["unaccounted"] = secs - sum(tm.values())
                e2e[name] = {"seconds": secs, "first_call_seconds": runs[0][0],
                             "samples_at_stop": int(B * len(r.error_history)) if len(r.error_history) else 0,
                             "checks": int(len(r.error_history)),
                             "overall_error": float(r.overall_error), "error_estimator": estimator,
                             "e2e_breakdown": {k: round(v, 6) for k, v in tm.items()}}
            return e2e

        breakdown_note = ("e2e_breakdown: host seconds of the faster of two calls -- engine_create (context, stream), setup "
                          "(communicator, precision), sampler_start (generator + QMC constructor handed to a helper thread), "
                          "reduction_copy_gram / reduction_finalize (chunked H2D over PCIe under the Gram kernels; scaling and "
                          "sync -- the library's own timers, lsspa_reduce_timing; reduction_pin / reduction_unpin are 0 since "
                          "round 4: nothing of the caller's is page-locked) and reduction_host (what is left of the phase: coercion of the arrays, "
                          "buffer allocation, the call), sampler (drawing orderings), sampling (upload + kernels + "
                          "statistics of the loop, incl. workspace allocation on the first batch), estimator (statistics "
                          "read-back + error estimate), final_fit (theta, r^2), teardown (free, destroy)")
        if world == 1:
            got = guarded("time_to_tolerance_e2e", lambda: e2e_runs(host))
            if got is not None:
                legs["time_to_tolerance_e2e"] = dict(
                    got, h2d_included=True, tolerance=1e-2,
                    default_error_estimator="device (what ls_spa(method='argsort') uses when error_estimator is not given: "
                                            "the reference's code has no QMC method whose generator interleave could be "
                                            "mirrored; 'reference' is the default of the seed / perms= paths)",
                    note="public ls_spa() on the host arrays of the timed region: engine creation, streamed reduction "
                         "over PCIe, sampling loop, estimator, final fit, teardown; " + breakdown_note)
        else:
            legs["time_to_tolerance_e2e"] = {"skipped": "several ranks: see time_to_tolerance* for the sharded loop"}

        # The public call over a run of MANY checks (round 5): every time-to-tolerance leg above stops at check 1 or 2,
        # and the headline drives run_batch with orderings drawn beforehand -- neither shows what the sampler, the
        # estimator and the driver cost per batch.  tolerance = 0: the stop rule is evaluated at every check and never
        # fires.  One GPU: ls_spa() on the host arrays (reduction over PCIe included in `seconds`, not in the loop's
        # rate); several ranks: the sharded loop on the resident problem.
        if not args.no_full_run:
            nb = args.full_run_batches or (64 if p <= SMALL_P_MAX else (128 if p <= 2000 else 16))

            def full_run():
                reps = []
                for rep in range(2):
                    barrier()
                    tm = {"check_s": []}
                    t0 = time.perf_counter()
                    if world == 1:
                        r = ls_spa(*host, reg=reg, method="argsort", batch_size=B, num_batches=nb, tolerance=0.0, seed=42,
                                   device=local, precision="float32" if args.dtype == "f32" else "float64", _timings=tm)
                        checks, n_done = len(r.error_history), B * nb
                        sum_attr, r2 = float(r.attribution.sum()), float(r.r_squared)
                    else:
                        eng.set_lanes(args.lanes)
                        res = run_estimator(eng, p, max_samples=B * nb, batch_size=B, tolerance=0.0, seed=42, perms=None,
                                            antithetical=True, return_attribution_history=False, method="argsort",
                                            error_estimator="device", comm=comm, timings=tm, lookahead="auto")
                        eng.set_lanes(1)
                        checks, n_done = len(res[3]), int(res[5])
                        sum_attr, r2 = float(res[0].sum()), None
                    barrier()
                    reps.append((time.perf_counter() - t0, tm, checks, n_done, sum_attr, r2))
                secs, tm, checks, n_done, sum_attr, r2 = min(reps, key=lambda v: v[0])
                per_check = [1e3 * v for v in tm.pop("check_s")]
                loop_s = tm.get("sampler", 0.0) + tm.get("estimator", 0.0) + tm.get("sampling", 0.0)
                return {"seconds": secs, "seconds_sampling_loop": loop_s, "first_call_seconds": reps[0][0],
                        "samples": n_done, "orderings": 2 * n_done, "checks": checks,
                        "orderings_per_s": 2 * n_done / loop_s, "orderings_per_s_whole_call": 2 * n_done / secs,
                        "fraction_of_value": None, "tolerance": 0.0, "error_estimator": "device (default of the method)",
                        "host_seconds": {k: round(v, 6) for k, v in tm.items()},
                        "loop_ms_per_check": {
                            "first": per_check[0] if per_check else None, "last": per_check[-1] if per_check else None,
                            "median": float(np.median(per_check)) if per_check else None,
                            "max": max(per_check) if per_check else None},
                        "sum_attribution": sum_attr, "r_squared": r2,
                        "note": f"ls_spa(method='argsort', batch_size={B}, num_batches={nb}, tolerance=0.0): the public call "
                                "through sampler (SciPy's Sobol' stream and its argsort, drawn ahead by threads of the library), driver, device estimator "
                                "(running form, checks deferred by one) and statistics; orderings_per_s is over the "
                                "sampling loop (host_seconds sampler + estimator + sampling), seconds the whole call "
                                "(one GPU: engine, reduction over PCIe, final fit included); loop_ms_per_check = host "
                                "wall time per check of the loop -- its period, the wait for the GPU included: the same at "
                                "the first check and the last (what the estimator's kernels cost a check is estimator_gpu)"}
            got = guarded("full_run", full_run)
            if got is not None:
                legs["full_run"] = got

                def estimator_gpu_ms():
                    # what a check costs the GPU: eight checks of the same loop on one lane with a HIP-event pair around
                    # every launch; the estimator's launches (fold the chunk into D = Xi L / s = Xi 1; the two quantile
                    # kernels) are class "error".  Nothing in them depends on how many samples came before.
                    eng.set_lanes(1)
                    eng.profile(True)
                    eng.profile_reset()
                    run_estimator(eng, p, max_samples=B * 8, batch_size=B, tolerance=0.0, seed=42, perms=None,
                                  antithetical=True, return_attribution_history=False, method="argsort",
                                  error_estimator="device", comm=comm, lookahead=1, defer=0)
                    barrier()
                    ms, cnt = eng.profile_read()["error"]
                    eng.profile(False)
                    return {"ms_per_check": ms / 8.0, "launches_per_check": cnt / 8.0}
                g2 = guarded("full_run_estimator_gpu", estimator_gpu_ms)
                if g2 is not None:
                    legs["full_run"]["estimator_gpu"] = g2
                legs.pop("full_run_estimator_gpu", None)

        # SURVEY 8(d)'s secondary workload: the reference's own correlated generator at the same shape.  On the iid
        # Gaussian data the stop rule fires at the first check (error ~3e-5 against 1e-2: one batch); here the
        # attribution is spread over correlated features and the loop has to run
        # (p <= 2000: at C5's size the generator, and the host estimator's 5000 x 5000 SVD at every one of ~130 checks,
        # turn the leg into minutes of host work)
        if world == 1 and args.data == "gaussian" and not args.no_correlated_leg and rank == 0 and p <= 2000:
            def correlated_leg():
                t0 = time.perf_counter()
                data = correlated_data(p, rows, args.dtype)
                gen = time.perf_counter() - t0
                res = dict(e2e_runs(data), host_data_generation_s=gen, h2d_included=True, tolerance=1e-2)
                # the CPU's time for the same run, composed as BASELINE.md section 3 prescribes from the rate measured
                # on THIS reduced problem
                res["note"] = ("public ls_spa() on the reference's correlated data (experiments/ground_truth_medium.py:"
                               f"74-106; p={p}, N=M={rows}, seed 42, method='argsort', batch_size={B}, up to 128 batches)")
                return res
            got = guarded("time_to_tolerance_correlated", correlated_leg)
            if got is not None:
                legs["time_to_tolerance_correlated"] = got
    if watchdog is not None:
        watchdog.cancel()
    if out is not None:
        out.update(legs)
        if isinstance(out.get("full_run"), dict) and "orderings_per_s" in out["full_run"]:
            out["full_run"]["fraction_of_value"] = out["full_run"]["orderings_per_s"] / out["value"]

    if out is not None and world == 1 and not args.no_cpu_baseline:
        G, g, H, h = eng.gram()
        out["cpu_baseline"] = cpu_baseline(host, G, g, H, h, eng.y_norm_sq, p, reg, n_stop, n_checks, cov_at_stop)
        cb = out["cpu_baseline"]
        out["speedup_vs_cpu_baseline"] = out["value"] / cb["value"]
        out["speedup_vs_cpu_baseline_note"] = (
            f"against ONE host process on {cb['blas_threads']} BLAS threads of {cb['host_cores_total']} host cores (the "
            "reference's cost model: it is single-process); says nothing about kernel quality -- roofline.frac does")
        corr = out.get("time_to_tolerance_correlated")
        if isinstance(corr, dict) and "reference" in corr and cb.get("error_estimates_s") is not None:
            # per-ordering cost and the estimator's cost depend on p only, so the rate measured above composes the CPU
            # time of the correlated run as well (BASELINE.md section 3's recipe, the GPU run's stop index)
            ns, nc = corr["reference"]["samples_at_stop"], corr["reference"]["checks"]
            corr["cpu_time_to_tolerance_s"] = cb["reduction_s"] + 2 * ns / cb["value"] + nc * cb["error_estimates_s"]
            corr["cpu_composition"] = (f"reduction {cb['reduction_s']:.1f} s + {ns} samples x 2 orderings / "
                                       f"{cb['value']:.1f} per s + {nc} x error_estimates {cb['error_estimates_s']:.2f} s")
    if out is not None:
        print(json.dumps(out))
    if hasattr(comm, "close"):
        comm.close()
    eng.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
