"""The estimator driver: ``ls_spa(...)`` with the reference's signature and semantics,
running every ordering on the HIP engine in batches.

Reference: cvxgrp/ls-spa ``ls_spa/ls_spa.py:122-253``.  What is kept exactly:

* the 12 positional parameters, their order and defaults (:122-133);
* input coercion through ``np.array`` and ``validate_data`` (:158-162);
* ordering source selection (:169-177): ``perms is None`` and p < 9 -> every ordering,
  ``batch_size`` forced to 256, ``antithetical`` forced off; ``perms is None`` and p >= 9 ->
  ``rng.permutation`` drawn lazily from the SAME generator the error estimator uses;
  ``perms`` given -> consumed lazily, no sample cap;
* the error-check trigger indices ``i % batch_size == 0 or i == max_samples - 1``, the
  ``p >= 9`` guard, the tolerance break and the trailing estimate (:222-236).  Because the
  generator is only touched by the sampler between two checks, drawing a whole chunk of
  orderings up to the next check index and evaluating it as one GPU batch leaves the
  generator's call sequence -- hence every later ordering -- unchanged;
* results: running mean = attribution, biased covariance scaled as in :223-224.

README-dialect keywords (README.md:96-106) are accepted on top: ``method``,
``num_batches``, ``return_history``.
"""
from __future__ import annotations

import json
import os
import warnings

import numpy as np

from . import _samplers as S
from ._native import LSSPANativeError
from ._results import ShapleyResults, validate_data
from ._stats import error_estimates, error_estimates_lowrank

# problems up to this many features take the one-workgroup-per-ordering kernels (csrc/k_small.hip small_p_eligible:
# p + 1 <= 128): one lane, look-ahead groups of up to sixteen chunks, larger sampler blocks
SMALL_P_MAX = 127


def auto_lookahead(p, per_rank):
    """Chunks of `per_rank` samples launched as one batch when nobody says otherwise (lookahead='auto').

    A chunk of fewer than 64 samples per rank leaves most of an MI355X idle at p = 1000 -- as many
    chunks as make up 64, eight at most.  Smaller problems need more samples to fill the chip (the work of an ordering
    goes with p^3, its tiles with p^2; bench.py at 128 samples a step, one against the best look-ahead: p = 150 0.67 ->
    0.93 M orderings/s at 8, p = 200 0.65 -> 0.91 at 8, p = 300 0.39 -> 0.45 at 4, p = 500 0.21 -> 0.23 at 4, p = 700 +2 %,
    p = 1000 -1 %): 1024 samples up to p = 250, 512 up to p = 800.  Small problems (the one-workgroup-per-ordering
    kernels): a group costs the host one call and the GPU ~45 us of dependent launches around its lift kernel whatever
    its size (statistics, estimator and checks of all its chunks are five launches), against 25 us of kernel per chunk
    of 128 samples at p = 100 -- as many chunks as make up 2048 samples, sixteen at most; a chunk of 1024 samples fills
    the chip four times over by itself."""
    per_rank = max(int(per_rank), 1)
    if p <= SMALL_P_MAX:
        return 1 if per_rank >= 1024 else max(1, min(16, 2048 // per_rank))
    want = 1024 if p <= 250 else (512 if p <= 800 else 64)
    return max(1, min(8, want // per_rank))

_NO_CAP = 2 ** 100

# ---- engines kept between calls ------------------------------------------------------------------------------------
# ls_spa() needs a context on the GPU and, for large p, tens of GB of per-batch workspace (C5: 80 GB).  Creating and
# releasing that on every call cost 2.3 of a C5 call's 2.9 seconds (profiles/r03_bench_c5.json: first_call_seconds
# against 0.61 s of work).  One process per GPU is the deployment model and HBM is sized for it, so the engine of a
# device -- context, stream, workspace -- stays alive between calls of one process and the next call of the same shape
# finds its buffers in place (a different shape re-sizes them, as before).  ``release()`` frees everything at once;
# LSSPA_ENGINE_CACHE=0 restores an engine per call.  A kept engine that is busy (another thread inside ls_spa on the same
# device) is not shared: that call makes its own.  The reference keeps no state between calls either way: every
# call resets the statistics, the history and the flags it touches.
_engine_cache = {}
_engine_cache_lock = __import__("threading").Lock()     # guards the table itself (two threads, one new device)


def _acquire_engine(device):
    """(engine, lock or None): a kept engine of this device if it is free, else a fresh one (lock None: caller closes it)."""
    import threading
    from ._engine import HipEngine
    if os.environ.get("LSSPA_ENGINE_CACHE", "1") == "0":
        return HipEngine(device), None
    with _engine_cache_lock:
        slot = _engine_cache.get(device)
        if slot is None:
            if not _engine_cache:
                import atexit
                atexit.register(release)
            slot = _engine_cache[device] = {"engine": None, "lock": threading.Lock()}
    if not slot["lock"].acquire(blocking=False):
        return HipEngine(device), None
    try:
        if slot["engine"] is None or getattr(slot["engine"], "_h", None) is None:
            slot["engine"] = HipEngine(device)
        return slot["engine"], slot["lock"]
    except BaseException:
        slot["lock"].release()
        raise


def release(device=None):
    """Close the engines ls_spa() keeps between calls (all of them, or one device's): frees their HBM."""
    with _engine_cache_lock:
        slots = list(_engine_cache.items())
    for dev, slot in slots:
        if device is not None and dev != device:
            continue
        if slot["lock"].acquire(blocking=False):
            try:
                if slot["engine"] is not None:
                    slot["engine"].close()
                slot["engine"] = None
            finally:
                slot["lock"].release()


def _next_check(i, batch_size, max_samples):
    """Smallest index > i at which the reference evaluates the error estimate."""
    nxt = (i // batch_size + 1) * batch_size
    if i < max_samples - 1 < nxt:
        nxt = max_samples - 1
    return nxt


class _Comm:
    """Single-process communicator (world of one)."""
    rank, world = 0, 1

    def allreduce_pending(self, engine):
        return None

    def allreduce_draws(self, engine):
        return None

    def allreduce_reduction(self, engine):
        return None

    def sum_ints(self, values):
        return [int(v) for v in values]

    def gather_ints(self, values):
        return [[int(v) for v in values]]

    def gather_lifts(self, local, counts):
        return local


def gather_ints_by_sum(comm, values):
    """Every rank's integer vector, [world][k], through the one integer collective a communicator has to offer
    (sum_ints): each rank writes its own slot of a zero vector, the sum is the concatenation."""
    k = len(values)
    flat = [0] * (comm.world * k)
    flat[comm.rank * k:(comm.rank + 1) * k] = [int(v) for v in values]
    total = comm.sum_ints(flat)
    return [total[r * k:(r + 1) * k] for r in range(comm.world)]


def _min_norm_theta(G, g):
    """theta of minimal norm for a numerically singular Gram matrix (the reference gets it
    from lstsq on the triangular factor, ls_spa/ls_spa.py:240)."""
    from ._stats import host_blas_threads
    with host_blas_threads():
        w, Q = np.linalg.eigh(G)
    keep = w > w.max() * G.shape[0] * np.finfo(float).eps
    coef = np.zeros_like(w)
    coef[keep] = (Q.T @ g)[keep] / w[keep]
    return Q @ coef


# 3: the identity carries `history`, `precision` and `data`, the state `attribution_history` / `history_sum` (round 3);
# 4: the device estimator's state is its running sums (err_D, err_s) instead of the lift history (round 5);
# a file of another version is refused by name instead of failing on a missing key.  With return_attribution_history the
# whole n x p history is rewritten at every save (I/O quadratic in the run length): checkpoint long history runs sparsely.
_CKPT_VERSION = 4


def _ckpt_path(path, comm):
    return path if comm.world == 1 else f"{path}.rank{comm.rank}"


def _save_checkpoint(path, comm, state):
    """Atomic write (temp file + rename) of the estimator state; one file per rank.  The file it replaces is
    kept as ``<file>.prev``: the ranks write after the same check but not at the same instant, so a job killed
    between two ranks' renames leaves them one generation apart -- the older one is then the common state."""
    target = _ckpt_path(path, comm)
    tmp = target + ".tmp.npz"
    np.savez(tmp, **state)
    if os.path.exists(target):
        os.replace(target, target + ".prev")
    os.replace(tmp, target)


def _read_checkpoint(target, expect):
    if not os.path.exists(target):
        return None
    with np.load(target, allow_pickle=False) as z:
        st = {k: z[k] for k in z.files}
    if int(st["version"]) != _CKPT_VERSION:
        raise ValueError(f"checkpoint {target}: unsupported version {int(st['version'])}")
    for key, want in expect.items():
        have = st[key].item() if st[key].shape == () else st[key]
        if str(have) != str(want):
            raise ValueError(f"checkpoint {target} was written with {key}={have}, this run has {key}={want}")
    return st


def _load_checkpoint(path, comm, expect):
    """The state to resume from, or None.  With several ranks the ranks first agree on it: each reports the sample
    counts of the (at most two) generations it holds and all resume from the largest count EVERY rank holds;
    if there is none -- a rank without a file next to ranks with files, files of different runs -- every rank
    raises the same error instead of running chunks that no longer pair up in the collectives.

    Nothing raises before that exchange: a rank whose own file is unreadable or belongs to another run (other
    seed, data, sampler ...) reports it through the collective, and then ALL ranks raise -- a rank raising alone
    would leave the others waiting in the all-reduce.  A stale or foreign ``.prev`` next to a valid file is
    ignored (it is only ever the fallback generation)."""
    target = _ckpt_path(path, comm)
    gens, target_problem = [], None
    for k, f in enumerate((target, target + ".prev")):
        try:
            st = _read_checkpoint(f, expect)
        except Exception as exc:     # unreadable archive, missing key, version or ident mismatch
            st = None
            if k == 0:
                target_problem = f"{exc}"
        if st is not None:
            gens.append(st)
    if comm.world == 1:
        if target_problem is not None:
            raise ValueError(target_problem)
        return gens[0] if gens else None
    mine = ([int(st["n"]) for st in gens] + [-1, -1])[:2] + [0 if target_problem is None else 1]
    table = comm.gather_ints(mine)
    bad = [r for r, row in enumerate(table) if row[2]]
    if bad:
        raise ValueError(f"checkpoint {path}: the file of rank(s) {bad} is unreadable or was written by another run"
                         + (f" ({target_problem})" if target_problem is not None else "")
                         + "; remove the files to start over")
    held = [set(v for v in row[:2] if v >= 0) for row in table]
    if not any(held):
        return None
    common = set.intersection(*held)
    if not common:
        raise ValueError(f"checkpoint {path}: the ranks hold no common state (sample counts per rank: "
                         f"{[sorted(h) for h in held]}); remove the files to start over")
    n = max(common)
    return next(st for st in gens if int(st["n"]) == n)


def _data_fingerprint(engine):
    """A few numbers of the reduced problem that change with the data, the row count or reg."""
    G, g, _, _ = engine.gram()
    return f"{float(np.trace(G))!r}/{float(g @ g)!r}/{float(engine.y_norm_sq)!r}"


def prepare_sampling(p, *, max_samples, batch_size, seed, perms, antithetical, method, rank=0, world=1):
    """Generator and ordering source of a run, with the reference's overrides for small p applied
    (ls_spa/ls_spa.py:169-177).  Split from run_estimator so that ls_spa() can start it -- the QMC
    constructors work on a helper thread -- before the data reduction instead of after it."""
    rng = np.random.default_rng(seed)
    never_stop = False
    if perms is not None:
        if method is not None:
            raise ValueError("pass either perms= or method=, not both")
        source, max_samples = S.IterableSource(perms, p), _NO_CAP
    elif method is None:
        if p < 9:
            # the reference's loop runs over all p! orderings here; max_samples only ever enters
            # the (p >= 9)-guarded error check (:222), so it does not cap this case
            source, batch_size, antithetical, max_samples = S.exact_source(p), 2 ** 8, False, _NO_CAP
        else:
            source = S.RandomSource(rng, p, max_samples)
    elif method == "exact":
        source, batch_size, antithetical, never_stop = S.exact_source(p), 2 ** 8, False, True
        max_samples = _NO_CAP
    elif method == "random":
        source = S.RandomSource(rng, p, max_samples)
    elif method == "argsort":
        # a thread of the library draws the orderings ahead (no interpreter lock between it and this thread; SciPy's
        # engine still defines the stream); the Python source and its helper thread if that cannot be had
        source = (S.NativeArgsortSource.make(p, seed, max_samples, block=1024 if p <= SMALL_P_MAX else 256, rank=rank,
                                             world=world)
                  or S.ArgsortSource(p, seed, max_samples))
    elif method == "permutohedron":
        source = S.PermutohedronSource(p, seed, max_samples)
    else:
        raise ValueError(f"method must be one of {S.METHODS} or None")
    if source.independent:
        # the QMC samplers draw ahead of the loop on a helper thread (their stream is nobody else's), from now on --
        # in ls_spa() that is under the data reduction
        if not isinstance(source, S.NativeArgsortSource):
            source = S.PrefetchedSource(source, block=1024 if p <= SMALL_P_MAX else 256, rank=rank, world=world)
    return rng, source, batch_size, antithetical, max_samples, never_stop


def run_estimator(engine, p, *, max_samples, batch_size, tolerance, seed, perms, antithetical,
                  return_attribution_history, method, error_estimator, comm=None, chunk_cap=None,
                  checkpoint=None, prepared=None, lookahead=1, timings=None, defer=None):
    """The sampling loop on an engine whose problem is already loaded.  Returns
    (attribution, attribution_errors, overall_error, error_history, attribution_history, n).

    checkpoint: path of a state file.  Written after every error check (sample count, running mean and
    covariance, generator state, error history, the lift history the thin-form estimators need); if it
    exists when the call starts, the run continues from it -- the orderings, the estimator's draws and
    therefore every later number are those of the uninterrupted run.

    timings: optional dict; receives the host seconds spent drawing orderings ('sampler'), in the estimator calls
    ('estimator', reads of the statistics included) and in everything else of the loop ('sampling')."""
    import time as _time
    t_loop0 = _time.perf_counter()
    comm = comm or _Comm()
    if prepared is None:
        prepared = prepare_sampling(p, max_samples=max_samples, batch_size=batch_size, seed=seed, perms=perms,
                                    antithetical=antithetical, method=method, rank=comm.rank, world=comm.world)
    rng, source, batch_size, antithetical, max_samples, never_stop = prepared
    if batch_size < 1:
        raise ValueError("batch_size must be positive")
    try:
        return _run_estimator(engine, p, comm, rng, source, batch_size, antithetical, max_samples, never_stop,
                              tolerance=tolerance, seed=seed, return_attribution_history=return_attribution_history,
                              method=method, error_estimator=error_estimator, chunk_cap=chunk_cap,
                              checkpoint=checkpoint, lookahead=lookahead, timings=timings, defer=defer,
                              t_loop0=t_loop0)
    finally:
        if hasattr(source, "close"):
            source.close()


def _run_estimator(engine, p, comm, rng, source, batch_size, antithetical, max_samples, never_stop, *, tolerance, seed,
                   return_attribution_history, method, error_estimator, chunk_cap, checkpoint, lookahead, timings,
                   defer, t_loop0):
    import time as _time
    t_sampler = t_estimator = 0.0

    estimate = p >= 9
    keep_lifts = return_attribution_history or error_estimator == "lowrank"
    # one rank: nothing to all-reduce between a chunk's moments and their merge -- the engine folds the chunk into the
    # running statistics at once (accumulate = 2; one launch instead of two plus the merge call for small p)
    single = comm.world == 1 and not getattr(comm, "_force", False)
    acc_mode = 2 if single else True
    engine.reset_stats()
    on_device = error_estimator == "device" and estimate
    if on_device:
        # the running form (include/lsspa.h): D = Xi L and s = Xi 1 stay in HBM, Xi a function of (seed, sample, draw)
        engine.error_running_enable(int(np.random.SeedSequence(seed).generate_state(1, np.uint64)[0]))
    feat_err, total_err = np.zeros(p), 0.0
    err_hist, hist_parts, lift_parts = [], [], []
    hist_sum = np.zeros(p)
    i, pending, stop = 0, False, False
    mean = np.zeros(p)
    cov = None

    ident = {"p": p, "seed": seed, "method": str(method), "batch_size": batch_size,
             "antithetical": bool(antithetical), "error_estimator": error_estimator, "world": comm.world}
    if checkpoint is not None:
        ident["history"] = bool(return_attribution_history)
        ident["precision"] = str(getattr(engine, "precision", "float64"))
        ident["data"] = _data_fingerprint(engine)
        st = _load_checkpoint(checkpoint, comm, ident)
        if st is not None:
            i = int(st["n"])
            mean = st["mean"]
            engine.set_stats(i, mean, st["cov_biased"])
            rng.bit_generator.state = json.loads(str(st["rng_state"]))
            source.skip(i)
            err_hist = [float(v) for v in st["error_history"]]
            feat_err, total_err = st["feature_errors"], float(st["overall_error"])
            if return_attribution_history:
                hist_parts.append(st["attribution_history"])
                hist_sum = st["history_sum"]
            if error_estimator == "lowrank":
                lift_parts.append(st["lifts"])
            elif on_device:
                engine.set_error_state(st["err_D"], st["err_s"])
            if (i >= max_samples or (estimate and err_hist and total_err < tolerance and not never_stop)):
                stop = True   # the saved run had already finished

    def save_now(n):
        state = dict(ident, version=_CKPT_VERSION, n=n, rng_state=json.dumps(rng.bit_generator.state),
                     error_history=np.array(err_hist), feature_errors=feat_err, overall_error=total_err)
        _, state["mean"], state["cov_biased"] = engine.stats(want_cov=True)
        if return_attribution_history:
            state["attribution_history"] = np.concatenate(hist_parts) if hist_parts else np.zeros((0, p))
            state["history_sum"] = hist_sum
        if error_estimator == "lowrank":
            state["lifts"] = np.concatenate(lift_parts) if lift_parts else np.zeros((0, p))
        elif on_device:
            state["err_D"], state["err_s"] = engine.error_state()
        _save_checkpoint(checkpoint, comm, state)

    # lookahead > 1 (QMC samplers only: their stream is nobody else's): the orderings of several chunks are launched
    # as ONE batch -- a chunk of batch_size / world samples may fill a fraction of the GPU -- and then accumulated,
    # all-reduced and checked chunk by chunk in the reference's order (ls_spa/ls_spa.py:212-230).  When the stop
    # rule fires, the chunks launched beyond it are dropped: nothing of them ever reaches the statistics.  With a
    # host-side estimator the next group is launched AFTER the statistics of the group's last chunk have been read
    # back and BEFORE the host's estimate and decision, so the GPU works while the host computes (a stop then
    # wastes up to k chunks of GPU work, which only the final read-back waits for).  With the device-side estimator the
    # checks are enqueued behind the chunks' statistics and read late (below): the next group is launched when the
    # current one has been taken up, whatever its checks will say.
    ramp = None
    if lookahead == "auto":
        lookahead = auto_lookahead(p, -(-int(batch_size) // comm.world))
        # the automatic groups grow: 1, 2, 4, ... chunks up to the size above.  A run that stops at one of its first checks
        # -- the usual run to a tolerance -- then has its answer after about as many chunks as it needed, not after a
        # whole group (its kernels are one launch: none of its checks is known before all of its chunks have run); a long
        # run is at the full size after log2 of it groups.  An explicit lookahead = k is k from the first group on.
        ramp = [1]
    group = max(1, int(lookahead)) if (hasattr(engine, "launch_batch") and source.independent and not chunk_cap) else 1
    # The device estimator's checks are ENQUEUED, not waited for: x = (D - s mean^T) / sqrt(n (n - 1)), the all-reduce of
    # the per-rank x, the quantile kernels and a copy of (errors, running mean, n) into a pinned slot run on the context's
    # stream behind the chunk's statistics.  `defer` checks may be outstanding: the stop rule of check k is evaluated
    # when check k + defer has been enqueued -- the samples in between are already running and are dropped on a stop,
    # the results are those of check k (its own copy of the running mean) -- so the host never waits for the newest
    # work and the estimator is off the critical path.  The FIRST check of a run is always waited
    # for (a run on easy data ends there with nothing wasted), and a chunk that takes tens of milliseconds is too
    # (the round trip of a check is 0.2 ms: nothing to hide, and up to defer + 1 such chunks would run for nothing).
    # The point at which a check is resolved depends on counts only, never on timing: every rank takes the same decisions.
    outstanding = []     # (sample count, slot), oldest first
    slot_turn = [0]
    can_defer = on_device and checkpoint is None and source.independent and not chunk_cap
    if defer is None:
        per_rank = -(-min(int(batch_size), max_samples) // comm.world) * (2 if antithetical else 1)
        # a chunk's kernels, at 40 TFLOP/s, under 50 ms; the chunks of a look-ahead group are checked without waiting
        # in between (their kernels were one launch)
        defer = min(max(1, group), engine.RESULT_SLOTS - 2) if (can_defer and per_rank * float(p) ** 3 / 4e13 < 0.05) else 0
    defer = int(defer) if can_defer else 0
    if not 0 <= defer < engine.RESULT_SLOTS - 1 if on_device else False:
        raise ValueError("defer must be between 0 and the number of result slots - 2")

    def enqueue_check(n):
        slot = slot_turn[0]
        slot_turn[0] = (slot + 1) % engine.RESULT_SLOTS
        if single and hasattr(engine, "error_check_enqueue"):
            engine.error_check_enqueue(n, slot)      # one rank: nothing to all-reduce, the draws are never written
        else:
            engine.error_running_draws(n)
            comm.allreduce_draws(engine)
            engine.error_quantiles_enqueue(slot)
        outstanding.append((n, slot))

    def estimate_now(n, cov_b=None):
        nonlocal feat_err, total_err, t_estimator
        t_e0 = _time.perf_counter()
        with np.errstate(divide="ignore", invalid="ignore"):
            if error_estimator == "lowrank":
                centred = np.concatenate(lift_parts) - mean
                feat_err, total_err = error_estimates_lowrank(rng, centred, n)
            elif on_device:
                enqueue_check(n)
                feat_err, total_err, _, _ = engine.error_result(outstanding.pop()[1], wait=True)
            else:
                if cov_b is None:
                    _, _, cov_b = engine.stats(want_cov=True)
                feat_err, total_err = error_estimates(rng, cov_b * n / (n - 1) / n)
        err_hist.append(total_err)
        t_estimator += _time.perf_counter() - t_e0

    # Two lanes (engine.lanes == 2, set by ls_spa() on the general path for the QMC samplers): successive groups run on
    # two workspaces and two streams, the statistics on the context's own.  The next group is then launched as soon as
    # the current one's last chunk is taken up -- BEFORE its statistics are read back: the read-back waits for the
    # context's stream only, and the next group's kernels start when the current group's are half way (6.25 against
    # 6.45 ms a step at the C3 shape).  A stop wastes at most that one group, which is discarded.
    prefetch = (getattr(engine, "lanes", 1) == 2 and hasattr(engine, "launch_batch") and source.independent
                and not chunk_cap)
    # ... and a chunk goes to the engine as two half-chunks, each its own launch sequence on its own lane, once a half
    # still fills the chip (>= 32 samples = 64 orderings per rank): a finer-grained pipeline (6.18 against 6.37 ms a
    # step at the C3 shape).  The check indices are untouched -- a half-chunk that ends between two of them triggers
    # nothing -- and the statistics are folded in half-chunk by half-chunk, in order: the results differ from the
    # one-lane run's by the rounding of that grouping (1e-16 relative), nothing else.
    sub_cap = None
    if prefetch and group == 1 and -(-(int(batch_size) + 1) // 2 // comm.world) >= 32:
        sub_cap = (int(batch_size) + 1) // 2
    queue = []
    group_of = {}      # id(ticket) -> chunks launched with it: as many checks may be outstanding behind one of them

    def allowed(ticket):
        """Checks that may stay unread once a check of this ticket's group has been enqueued: the group's own (their
        kernels were one launch), `defer` at most."""
        return defer if (ramp is None or ticket is None) else max(min(defer, group_of.get(id(ticket), defer)), 1 if defer else 0)

    def refill(i_now):
        nonlocal t_sampler
        entries, cursor = [], i_now
        size = group
        if ramp is not None:
            size = min(group, ramp[0])
            ramp[0] *= 2
        for _ in range(size):
            if cursor >= max_samples:
                break
            target = _next_check(cursor, batch_size, max_samples)
            want = min(target, max_samples) - cursor
            if chunk_cap:
                want = min(want, chunk_cap)
            if sub_cap:
                want = min(want, sub_cap)
            t_s0 = _time.perf_counter()
            # sample number g of the run belongs to rank g mod world (round 5; up to round 4 the dealing restarted with
            # every chunk): a QMC source then draws -- ahead of the loop -- this rank's orderings only
            n_got, mine_rows = source.take_share(want, cursor, comm.rank, comm.world)
            t_sampler += _time.perf_counter() - t_s0
            if n_got == 0:
                break
            entries.append([n_got, mine_rows, want])
            cursor += n_got
            if n_got < want:
                break
        ticket = None
        if (group > 1 or prefetch) and entries:
            mine_all = np.concatenate([e[1] for e in entries])
            if len(mine_all):
                ticket = engine.launch_batch(mine_all, antithetical)
        first = 0
        for e in entries:
            e += [ticket, first]
            first += len(e[1])
        if ticket is not None:
            group_of[id(ticket)] = len(entries)
        queue.extend(entries)

    # the whole group in one library call: device estimator with its checks deferred by a group at least, nothing that
    # needs a chunk's lift vectors on the host, and moments that travel through the engine's own communicator (or not
    # at all)
    fast_group = (on_device and defer >= group and defer > 0 and not keep_lifts and checkpoint is None
                  and hasattr(engine, "group_collect")
                  and (single or getattr(comm, "_engine", None) is engine))
    stopped_at = None      # (mean, n) of the check whose stop rule fired (device estimator)
    resolved = [0]

    def resolve_due(limit):
        """Read the oldest outstanding checks until at most `limit` are left (the first check of a run is always read,
        whatever the limit); True if one of them stops the run."""
        nonlocal feat_err, total_err, stopped_at, t_estimator
        while outstanding and (len(outstanding) > limit or not resolved[0]):
            t_e0 = _time.perf_counter()
            n_k, slot = outstanding.pop(0)
            feat_err, total_err, mean_k, _ = engine.error_result(slot, wait=True)
            err_hist.append(total_err)
            resolved[0] += 1
            t_estimator += _time.perf_counter() - t_e0
            if total_err < tolerance and not never_stop:
                stopped_at = (mean_k, n_k)
                outstanding.clear()      # later checks: of samples the reference would never have drawn
                return True
        return False

    while not stop:
        if not queue:
            refill(i)
        if not queue:
            break
        n_new, mine, want, ticket, first = queue.pop(0)
        if fast_group and ticket is not None:
            # every chunk of the launched group in ONE call into the library (lsspa_group_collect): collect, all-reduce
            # and merge, fold into the estimator, enqueue the check -- per chunk, in the reference's order; at p = 100 a
            # chunk is 37 us of GPU work and this loop's own calls were what a run waited for
            members = [(n_new, mine, want, ticket, first)]
            while queue and queue[0][3] is ticket:
                members.append(tuple(queue.pop(0)))
            if (all(m[0] == m[2] for m in members)
                    and len(outstanding) + len(members) < engine.RESULT_SLOTS):
                t_g0 = _time.perf_counter()
                cursor = i
                if prefetch and not queue:
                    refill(i + sum(m[0] for m in members))      # the next group, on the other lane
                firsts, counts, ids, n_after, slots = [], [], [], [], []
                for n_ch, mn, _, _, fs in members:
                    firsts.append(fs)
                    counts.append(len(mn))
                    ids.append(cursor + (comm.rank - cursor) % comm.world)      # this rank's first sample of the chunk
                    cursor += n_ch
                    due = estimate and (cursor % batch_size == 0 or cursor == max_samples - 1)
                    n_after.append(cursor if due else 0)
                    slots.append(slot_turn[0] if due else 0)
                    if due:
                        slot_turn[0] = (slot_turn[0] + 1) % engine.RESULT_SLOTS
                engine.group_collect(ticket, firsts, counts, ids, comm.world, n_after, slots)
                outstanding.extend((n_a, sl) for n_a, sl in zip(n_after, slots) if n_a)
                i, pending = cursor, n_after[-1] == 0
                halt = resolve_due(allowed(ticket))
                if timings is not None and "check_s" in timings:
                    k = max(1, sum(1 for v in n_after if v))
                    timings["check_s"].extend([(_time.perf_counter() - t_g0) / k] * k)
                if halt or i >= max_samples:
                    break
                continue
            for m in reversed(members[1:]):      # a chunk cut short by a dry source: chunk by chunk below
                queue.insert(0, list(m))
        if prefetch and not queue and ticket is not None and n_new == want:
            refill(i + n_new)      # the next group, on the other lane
        local = None
        if len(mine):
            if ticket is not None:
                local = engine.collect_batch(ticket, want_lifts=keep_lifts, accumulate=acc_mode, first=first,
                                             count=len(mine))
            else:
                local = engine.run_batch(mine, antithetical, want_lifts=keep_lifts, accumulate=acc_mode)
            if on_device:
                # this rank's samples of the chunk: every world-th sample of the run from its first one at or after i
                engine.error_advance(i + (comm.rank - i) % comm.world, comm.world)
        if not single:
            comm.allreduce_pending(engine)
            engine.merge()
        if keep_lifts:
            offs = [(r - i) % comm.world for r in range(comm.world)]      # rank r's first position in this chunk
            counts = [len(range(offs[r], n_new, comm.world)) for r in range(comm.world)]
            parts = comm.gather_lifts(local if local is not None else np.empty((0, p)), counts)
            if comm.world > 1:
                full = np.empty((n_new, p))
                for r in range(comm.world):
                    full[offs[r]::comm.world] = parts[r]
            else:
                full = parts
            if return_attribution_history:
                run = hist_sum + np.cumsum(full, axis=0)
                hist_parts.append(run / np.arange(i + 1, i + n_new + 1)[:, None])
                hist_sum = run[-1]
            if error_estimator == "lowrank":
                lift_parts.append(full)
        i += n_new
        pending = True
        if n_new < want and not chunk_cap:
            stop = True  # the source ran dry inside a chunk
        check = estimate and (i % batch_size == 0 or i == max_samples - 1)
        if check and on_device:
            t_e0 = _time.perf_counter()
            enqueue_check(i)
            t_estimator += _time.perf_counter() - t_e0
            pending = False
            halt = resolve_due(allowed(ticket))
            if timings is not None and "check_s" in timings:
                timings["check_s"].append(_time.perf_counter() - t_e0)
            if halt:
                break
            if checkpoint is not None:       # (defer is 0 with a checkpoint: the engine's state is check i's)
                _, mean, _ = engine.stats(want_cov=False)
                save_now(i)
        elif check:
            t_e0 = _time.perf_counter()
            _, mean, cov_now = engine.stats(want_cov=error_estimator == "reference")
            t_estimator += _time.perf_counter() - t_e0
            if group > 1 and not queue and not stop and i < max_samples:
                refill(i)      # in flight while the host evaluates the stop rule below
            estimate_now(i, cov_now)
            pending = False
            if checkpoint is not None:
                save_now(i)
            if total_err < tolerance and not never_stop:
                break
        if i >= max_samples:
            break
    for tk in {id(e[3]): e[3] for e in queue if e[3] is not None}.values():
        engine.discard_batch(tk)     # launched, never accumulated

    if stopped_at is None:
        resolve_due(0)               # the run ended with checks still outstanding: they are read in order
    if stopped_at is not None:
        # the engine's statistics may have moved on by up to `defer` checks' samples: the run's results are the
        # stopping check's own copies
        mean, n = stopped_at
        if return_attribution_history and hist_parts:
            hist_parts = [np.concatenate(hist_parts)[:n]]
    else:
        n, mean, _ = engine.stats(want_cov=False)
        if estimate and pending and n > 0:
            estimate_now(n)
    history = None
    if return_attribution_history:
        history = np.concatenate(hist_parts) if hist_parts else np.zeros((0, p))
    if timings is not None:
        timings["sampler"] = timings.get("sampler", 0.0) + t_sampler
        timings["estimator"] = timings.get("estimator", 0.0) + t_estimator
        timings["sampling"] = timings.get("sampling", 0.0) + (_time.perf_counter() - t_loop0) - t_sampler - t_estimator
    return mean, feat_err, total_err, np.array(err_hist), history, n


def ls_spa(X_train, X_test, y_train, y_test, reg=0., max_samples=2 ** 13, batch_size=2 ** 8,
           tolerance=1e-2, seed=42, perms=None, antithetical=True, return_attribution_history=False, *,
           method=None, num_batches=None, return_history=None, device=0, error_estimator=None,
           precision="float64", row_sharded=False, checkpoint=None, comm=None, lookahead=None, lanes="auto",
           _engine=None, _comm=None, _timings=None, _defer=None):
    """Estimates the Shapley attribution of the out-of-sample R^2 of a least-squares fit.

    Positional parameters, defaults and behaviour follow cvxgrp/ls-spa
    (``ls_spa/ls_spa.py:122-253``).  One difference in what stays behind: the engine of the device -- its context, stream
    and per-batch workspace in HBM (gigabytes for large p) -- is kept alive between calls of a process, so that the
    next call of the same shape starts at once; ``ls_spa.release()`` (or ``release(device)``) frees it, and
    ``LSSPA_ENGINE_CACHE=0`` in the environment makes every call create and free its own.  Keyword-only additions:

    method:  None (reference behaviour), 'exact', 'random', 'argsort' or 'permutohedron'.
    num_batches:  if given, ``max_samples = batch_size * num_batches`` (README dialect).
    return_history:  alias of ``return_attribution_history``.
    device:  GPU index.
    error_estimator:  'reference' (host, same generator call order as the reference),
        'lowrank' (same distribution, O(n p) instead of an O(p^3) factorisation, on the host) or
        'device' (the low-rank form on the GPU: the lift vectors never leave HBM; same numbers as
        'lowrank' for the same seed up to summation order).  None (default) lets the method decide:
        'reference' wherever the reference's code has a behaviour to mirror -- ``method`` None / 'random' / 'exact' and
        every call with ``perms=``: there the estimator shares the generator with the ordering source and its
        normal draws are observable in what is drawn next (ls_spa/ls_spa.py:168-175, :224) -- and 'device' for the
        QMC methods 'argsort' / 'permutohedron', which the reference's code does not implement (only its README
        names them): the estimator is then statistically the reference's (same 0.95-quantile definition of draws
        with covariance C_unbiased / n, ls_spa/ls_spa.py:321-341) without its p x p factorisation on the host, which at
        p = 5000 is 7 of the call's 9 seconds.
    precision:  'float64' (default, the reference's arithmetic) or 'float32' for the per-ordering
        factorisation work (about half the time; lifts agree to ~1e-5 on well-conditioned data;
        the Gram reduction, lift accumulation and statistics stay float64).
    checkpoint:  path of a state file, written after every error check and resumed from if it exists
        (same data, seed and sampler required); with several ranks every rank keeps ``<path>.rank<r>``.
    lookahead:  QMC samplers ('argsort', 'permutohedron') only.  k > 1 launches the orderings of k chunks as one GPU
        batch (a chunk of batch_size / n_gpus samples may fill a fraction of the GPU), accumulates and checks them
        chunk by chunk in the reference's order and drops the chunks beyond a stop.  Same results; at most k
        chunks of wasted GPU work at the end of a run.  'auto' (auto_lookahead): as many chunks as make up 64 samples
        per rank (512 up to p = 800, 1024 up to p = 250: smaller problems need more samples to fill the chip), eight
        at most; 2048 and sixteen for p <= 127, the one-workgroup-per-ordering kernels.  Default:
        'auto' for the QMC methods with the device estimator (whose checks the loop does not wait for), else 1.
    lanes:  1, 2 or 'auto'.  2: successive chunk groups alternate between two workspaces on two HIP streams, the next
        group's kernels starting when the current group's are half way, its orderings drawn and uploaded before the
        current group's statistics are read back; a chunk of 64 samples or more per rank goes as two half-chunks (QMC
        samplers only, as for lookahead; same results up to the rounding of the half-chunk grouping of the statistics;
        a stop wastes at most one group of GPU work).  'auto': 2 for the QMC methods when p > 127 (the fused small-p
        kernel gains nothing from it), else 1.
    comm:  several GPUs, one process each: the communicator every rank passes -- ``NativeComm.from_env()``
        (RCCL through the C ABI, no PyTorch) or ``TorchComm()`` (torch.distributed: RCCL, or gloo on CPU in
        the tests).  Sample number g of the run belongs to rank g mod world (a QMC method then draws a rank's own
        orderings only); the only data-path collective is one all-reduce of the packed batch moments per chunk.
    row_sharded:  several ranks only (``comm``).  False: every rank passes the whole data set.
        True: the four arrays are this rank's ROWS of the training and test sets; the ranks reduce
        their rows and sum the Gram matrices with one all-reduce.  'train': only the training rows are
        sharded, every rank passes all test rows (needed when there are fewer than p test rows).
    """
    # the reference coerces with np.array (a copy, ls_spa/ls_spa.py:158-161); the inputs are never written here, so
    # asarray gives the same result without copying 1.6 GB at the C3 shape
    X_train, X_test = np.asarray(X_train), np.asarray(X_test)
    y_train, y_test = np.asarray(y_train), np.asarray(y_test)
    validate_data(X_train, X_test, y_train, y_test)
    if y_train.ndim != 1 or y_test.ndim != 1:
        raise ValueError("y_train and y_test must be one-dimensional")  # reference: concatenate error, :312
    p = X_train.shape[1]
    if return_history is not None:
        return_attribution_history = bool(return_history)
    if num_batches is not None:
        max_samples = int(batch_size) * int(num_batches)
    if error_estimator is None:
        error_estimator = "device" if (perms is None and method in ("argsort", "permutohedron")) else "reference"
    if error_estimator not in ("reference", "lowrank", "device"):
        raise ValueError("error_estimator must be None, 'reference', 'lowrank' or 'device'")

    import time as _time
    tm = _timings if _timings is not None else {}   # bench.py's e2e_breakdown: host seconds per phase of this call

    def lap(key, t0):
        tm[key] = tm.get(key, 0.0) + (_time.perf_counter() - t0)
        return _time.perf_counter()

    comm = comm if comm is not None else _comm
    engine = _engine
    owns = engine is None          # this call made (or borrowed) the engine: it also ends the communicator bound to it
    kept = None                    # the lock of a borrowed, kept engine
    # The ordering source first: a QMC source starts its helper thread here -- the first `import scipy.stats` of a process
    # (0.26 s; 1.3 s on a cold box), the Sobol' constructor (18 ms at p = 1000) and the first block of orderings then run
    # under the engine's creation and the data reduction instead of in front of the sampling loop.
    t0 = _time.perf_counter()
    share = dict(rank=comm.rank, world=comm.world) if comm is not None else {}
    prepared = prepare_sampling(p, max_samples=max_samples, batch_size=batch_size, seed=seed, perms=perms,
                                antithetical=antithetical, method=method, **share)
    t0 = lap("sampler_start", t0)
    ok = False
    try:
        if owns:
            engine, kept = _acquire_engine(device)
        t0 = lap("engine_create", t0)
        if comm is not None and hasattr(comm, "bind"):
            comm.bind(engine)      # RCCL communicator on this engine's GPU and stream (collective)
        if precision != "float64" or getattr(engine, "precision", "float64") != "float64":
            engine.set_precision(precision)
        if lookahead is None:
            # the device estimator's checks never make the loop wait: launching the chunks of a thin batch together costs
            # nothing but the samples already in flight at a stop
            lookahead = "auto" if (error_estimator == "device" and perms is None
                                   and method in ("argsort", "permutohedron")) else 1
        if lookahead != "auto" and int(lookahead) < 1:
            raise ValueError("lookahead must be >= 1 or 'auto'")
        if lanes == "auto":
            lanes = 2 if (perms is None and method in ("argsort", "permutohedron") and p > SMALL_P_MAX) else 1
        if int(lanes) not in (1, 2):
            raise ValueError("lanes must be 1, 2 or 'auto'")
        if hasattr(engine, "set_lanes") and getattr(engine, "lanes", 1) != int(lanes):
            engine.set_lanes(int(lanes))
        t0 = lap("setup", t0)
        if row_sharded:
            engine.load_data_sharded(X_train, X_test, y_train, y_test, reg, comm or _Comm(),
                                     shard_test=row_sharded != "train")
        else:
            engine.load_data(X_train, X_test, y_train, y_test, reg)
        t0 = lap("reduction_h2d_gram", t0)
        if _timings is not None and hasattr(engine, "reduce_timing"):
            # the library's own split of that phase: chunked copies + Gram kernels, finalize (the page-locking parts
            # are zero since round 4); what is left of the phase is host-side coercion and the call itself
            parts = engine.reduce_timing()
            whole = tm.pop("reduction_h2d_gram")
            tm["reduction_pin"], tm["reduction_copy_gram"] = parts["pin"], parts["h2d_gram"]
            tm["reduction_unpin"], tm["reduction_finalize"] = parts["unpin"], parts["finalize"]
            tm["reduction_host"] = whole - sum(parts.values())
        # theta and r^2 (ls_spa/ls_spa.py:240-243) depend on the reduced problem only: computed BEFORE the sampling loop,
        # so that the call does not end behind whatever the loop launched ahead of a stop and never collected
        t0 = _time.perf_counter()
        theta, r_squared, info = engine.full_fit()
        t0 = lap("final_fit", t0)
        def sampling_run(prep):
            out = run_estimator(
                engine, p, max_samples=max_samples, batch_size=batch_size, tolerance=tolerance, seed=seed,
                perms=perms, antithetical=antithetical, return_attribution_history=return_attribution_history,
                method=method, error_estimator=error_estimator, comm=comm, checkpoint=checkpoint, prepared=prep,
                lookahead=lookahead, timings=tm, defer=_defer)
            return out, info | (engine.info_collected() if hasattr(engine, "info_collected") else engine.info())

        (attribution, feat_err, total_err, err_hist, history, _), bits = sampling_run(prepared)
        t0 = _time.perf_counter()
        # LSSPA_INFO_SCAN_WAIT (4): a hand-over inside a panel launch timed out; LSSPA_INFO_SUM (8): a sample's lifts did
        # not sum to the R^2 of the full model (every ordering's must, ls_spa/ls_spa.py:284-285).  Either way the lift
        # vectors of the run are not valid (with a pivot that broke down, bit 1, they are meaningless anyway).
        fault = bool(bits & 12) and not bits & 1
        if fault and perms is None and checkpoint is None and hasattr(engine, "set_flags"):
            # An ordering source that can be drawn again (seed or QMC method; not the caller's iterable): the run is
            # repeated ONCE on the conservative path -- the lift kernel of its own reads V^T back, nothing is handed over
            # inside a launch (developer flag 512) -- instead of being lost.
            warnings.warn(f"engine fault in the fused lift scan (info bits {bits}): the run is repeated with the lift "
                          "kernel of its own", RuntimeWarning, stacklevel=2)
            engine.set_flags(512)
            prepared[1].close() if hasattr(prepared[1], "close") else None
            prepared = prepare_sampling(p, max_samples=max_samples, batch_size=batch_size, seed=seed, perms=perms,
                                        antithetical=antithetical, method=method, **share)
            (attribution, feat_err, total_err, err_hist, history, _), bits = sampling_run(prepared)
            fault = bool(bits & 12) and not bits & 1
        if fault:
            raise LSSPANativeError(
                ("the fused lift scan gave up waiting for a row of its panel" if bits & 4 else
                 "a sample's lifts did not sum to the R^2 of the full model")
                + f" (info bits {bits}): the lift vectors of this run are not valid (engine fault)")
        if bits & 1:
            warnings.warn("a permuted Gram matrix was not numerically positive definite; the attribution "
                          "of collinear features is not meaningful (the reference's is not either)",
                          RuntimeWarning, stacklevel=2)
        if info & 1:
            G, g, _, _ = engine.gram()
            theta = _min_norm_theta(G, g)
            yy = engine.y_norm_sq
            if engine.tri:      # from the Gram side: also right when the test rows are sharded
                _, _, H, h = engine.gram()
                r_squared = float((2.0 * (h @ theta) - theta @ H @ theta) / yy)
            else:
                pred = X_test.astype(np.float64) @ theta
                r_squared = float((2.0 * (pred @ y_test) - pred @ pred) / yy)
        t0 = lap("final_fit", t0)
        ok = True
    finally:
        t0 = _time.perf_counter()
        if prepared is not None and hasattr(prepared[1], "close"):
            prepared[1].close()      # the sampler's helper thread (already ended by a run that got as far as its loop)
        if owns and engine is not None:
            try:
                if comm is not None and hasattr(comm, "close"):
                    comm.close()       # the communicator lives on the engine's context
                if kept is None or not ok:
                    engine.close()     # a kept engine an exception went through is not trusted with another call
                else:
                    engine.set_flags(0)
                    engine.history_enable(0)      # (the lanes stay as they are: the next call sets what it needs)
            finally:
                if kept is not None:
                    kept.release()
        lap("teardown", t0)
    return ShapleyResults(attribution=attribution, theta=theta, overall_error=total_err,
                          attribution_errors=feat_err, r_squared=r_squared, error_history=err_hist,
                          attribution_history=history)


# ------------------------------------------------------------------------------------------
# helper-level surface of the reference (called by its notebooks and tests)
# ------------------------------------------------------------------------------------------
_helper_engine = None


def _helper():
    global _helper_engine
    if _helper_engine is None:
        from ._engine import HipEngine
        _helper_engine = HipEngine(0)
    return _helper_engine


def reduce_data(X_train, X_test, y_train, y_test, reg):
    """(R_tr, F_te, q_tr, q_te) with R_tr^T R_tr = X_tr^T X_tr / N + reg I, R_tr^T q_tr = X_tr^T y_tr / N,
    F_te^T F_te = X_te^T X_te, F_te^T q_te = X_te^T y_te -- the contract of the reference's
    ``reduce_data`` (ls_spa/ls_spa.py:290-318).  Here R_tr is the Cholesky factor of the MFMA Gram
    matrix (positive diagonal), so it equals LAPACK's QR factor up to row signs."""
    eng = _helper()
    eng.load_data(np.array(X_train), np.array(X_test), np.array(y_train), np.array(y_test), reg)
    return eng.factors()


def square_shapley(X_train, X_test, y_train, y_test, y_norm_sq, perm):
    """Lift vector of one ordering from reduced factors (ls_spa/ls_spa.py:256-287): arguments are
    the four outputs of ``reduce_data``, ||y_test||^2 of the raw labels and the ordering."""
    R, F = np.asarray(X_train, dtype=np.float64), np.asarray(X_test, dtype=np.float64)
    q, qt = np.asarray(y_train, dtype=np.float64), np.asarray(y_test, dtype=np.float64)
    eng = _helper()
    eng.load_reduced(R.T @ R, R.T @ q, float(q @ q), float(y_norm_sq), Ft=F.T.copy(), ytil=qt)
    return eng.run_batch(np.asarray(perm)[None, :], False, want_lifts=True, accumulate=False)[0]
