"""Ordering sources.

'exact' and 'random' are the two sources built into the reference driver
(cvxgrp/ls-spa ``ls_spa/ls_spa.py:170-175``); 'argsort' and 'permutohedron' are the
quasi-Monte-Carlo samplers of ``experiments/ground_truth_medium.py:56-71``, which the
reference README names as ``method=`` values.  They stay on the host (SciPy QMC
streams are part of the observable behaviour) and yield int orderings in chunks.
"""
from __future__ import annotations

import itertools
import threading
import warnings

import numpy as np

METHODS = ("exact", "random", "argsort", "permutohedron")


def helmert_rows(p):
    """(p-1) x p row-orthonormal basis of {x : sum x = 0}: row k = (1,..,1,-(k+1),0,..)/norm."""
    U = np.tril(np.ones((p - 1, p)))
    k = np.arange(p - 1)
    U[k, k + 1] = -(k + 1.0)
    return U / np.linalg.norm(U, axis=1, keepdims=True)


class OrderingSource:
    """Hands out up to ``count`` orderings at a time as an (n, p) integer array;
    an empty array means exhausted."""

    # True when drawing orderings ahead of time changes nothing anybody can observe (the source owns its stream).
    # False for the caller's iterable (the reference pulls it lazily, ls_spa/ls_spa.py:197) and for the shared
    # generator (the error estimator draws from it between batches).
    independent = False

    def take(self, count):  # pragma: no cover - interface
        raise NotImplementedError

    def take_share(self, count, first, rank, world):
        """(n, own): the next ``count`` orderings of the run are taken -- n of them exist --; ``own`` are those a rank
        evaluates: ordering number g of the run (``first`` is the number of the first one taken here) belongs to rank
        g mod world.  Default: draw them all, hand back the share (a caller's iterable and the shared generator are
        consumed in full on every rank, as the one-process run consumes them); the QMC sources do only their share's
        work."""
        rows = self.take(count)
        return len(rows), rows[(rank - first) % world::world]

    def skip(self, count):
        """Advance past ``count`` orderings already consumed before a checkpoint.  Default: draw and
        discard (exact for every deterministic source); subclasses do better where they can."""
        while count > 0:
            got = len(self.take(min(count, 4096)))
            if got == 0:
                break
            count -= got


class IterableSource(OrderingSource):
    """Any iterable of length-p index sequences (generator, list, ndarray rows, an object
    with ``__iter__`` such as a progress bar).  Pulled lazily, never exhausted eagerly."""

    def __init__(self, iterable, p):
        self._it = iter(iterable)
        self._p = p

    def take(self, count):
        rows = list(itertools.islice(self._it, count))
        if not rows:
            return np.empty((0, self._p), dtype=np.int32)
        out = np.asarray([np.asarray(r) for r in rows])
        if out.ndim != 2 or out.shape[1] != self._p:
            raise ValueError(f"every ordering must have length p = {self._p}")
        return out


class RandomSource(OrderingSource):
    """rng.permutation(p), one call per ordering, drawn only when asked for -- the shared
    generator is also consumed by the error estimator between batches."""

    def __init__(self, rng, p, limit):
        self._rng, self._p, self._left = rng, p, limit

    def take(self, count):
        n = int(min(count, self._left))
        self._left -= n
        if n <= 0:
            return np.empty((0, self._p), dtype=np.int32)
        return np.stack([self._rng.permutation(self._p) for _ in range(n)])

    def skip(self, count):
        # the orderings came out of the shared generator, whose state the checkpoint restores
        self._left -= int(min(count, self._left))


class _BackgroundBuild:
    """Constructs an object on a helper thread.  SciPy's Sobol constructor takes ~18 ms at p = 1000 -- as long as
    two batches on the GPU -- so it runs while the caller is inside the (GIL-free) data reduction."""

    def __init__(self, factory):
        self._value, self._error = None, None

        def work():
            try:
                self._value = factory()
            except BaseException as exc:   # re-raised in the caller's thread
                self._error = exc

        self._thread = threading.Thread(target=work, daemon=True)
        self._thread.start()

    def get(self):
        if self._thread is not None:
            self._thread.join()
            self._thread = None
        if self._error is not None:
            raise self._error
        return self._value


_sort_pool = None
_sort_pool_lock = threading.Lock()


def _pool():
    """A few sorting threads kept for the life of the process (making them costs 0.2-0.3 ms, a block's sort 0.2-2 ms)."""
    global _sort_pool
    if _sort_pool is None:
        with _sort_pool_lock:
            if _sort_pool is None:
                import os
                from concurrent.futures import ThreadPoolExecutor
                _sort_pool = ThreadPoolExecutor(max(1, min(4, (os.cpu_count() or 2) // 2)),
                                                thread_name_prefix="lsspa-argsort")
    return _sort_pool


_native_sort = False      # False: not looked for yet; None: not there; else the library's lsspa_host_argsort_rows


def _native_argsort(a):
    """np.argsort(a, axis=1) by the library's own threads (include/lsspa.h, lsspa_host_argsort_rows), or None if the
    library is not there.  A row whose keys are all different has one argsort; the rows the library marks -- equal keys,
    NaN: numpy's order among those is its sort's own business and the reference's results inherit it -- are sorted by
    numpy here, so the result is numpy's for every row."""
    global _native_sort
    if _native_sort is False:
        try:
            from . import _native
            _native_sort = (_native.load().lsspa_host_argsort_rows, _native)
        except Exception:
            _native_sort = None
    if _native_sort is None:
        return None
    fn, N = _native_sort
    import ctypes as C
    import os
    a = np.ascontiguousarray(a, dtype=np.float64)
    out = np.empty(a.shape, dtype=np.int32)
    redo = np.empty(len(a), dtype=np.uint8)
    n_redo = C.c_int64()
    threads = max(1, min(3, (os.cpu_count() or 2) // 4))
    rc = fn(N.dptr(a), len(a), a.shape[1], N.iptr(out), redo.ctypes.data_as(C.POINTER(C.c_uint8)), threads,
            C.byref(n_redo))
    if rc != 0:
        return None
    if n_redo.value:
        rows = np.nonzero(redo)[0]
        out[rows] = np.argsort(a[rows], axis=1)
    return out


def _argsort_rows(a):
    """np.argsort(a, axis=1) as int32 (what the engine uploads).  Blocks of 64 rows and more by the library's native
    threads (_native_argsort: the same result, row by row); without the library numpy, large blocks cut into row ranges
    sorted on a few threads (the sort releases the GIL).  (Python threads for the 1024 x 100 blocks of a small problem
    took the driver's own thread 1.6 ms of a 4 ms run -- measured, round 5.)"""
    n = len(a)
    if n >= 64 and a.ndim == 2 and a.shape[1] >= 2:
        got = _native_argsort(a)
        if got is not None:
            return got
    if a.size < (1 << 17) or n < 64:
        return np.argsort(a, axis=1).astype(np.int32)
    pool = _pool()
    workers = max(1, min(pool._max_workers, n // 32))
    if workers == 1:
        return np.argsort(a, axis=1).astype(np.int32)
    out = np.empty(a.shape, dtype=np.int32)
    cuts = np.linspace(0, n, workers + 1).astype(int)

    def one(k):
        out[cuts[k]:cuts[k + 1]] = np.argsort(a[cuts[k]:cuts[k + 1]], axis=1)
    list(pool.map(one, range(workers)))
    return out


class _DirectSobol:
    """Point number i of a SciPy Sobol' engine without drawing the points before it: the engine's state after i steps is
    its initial state XOR the direction numbers of the bits set in the Gray code of i (what its draw loop accumulates
    one lowest-zero-bit at a time).  With several GPUs a rank needs every world-th ordering only; drawing all of them
    on every rank made the sampler, not the GPUs, the bound of a dealt batch (p = 1000: 1.06 ms of points + 1.86 ms of
    argsort per 128 orderings against ~1 ms of kernels for a rank's 16).  Reads private attributes of the engine
    (``_sv``, ``_quasi``, ``_scale``): ``make`` checks the result against the engine's own output on a copy -- the first
    67 points and five beyond number 4096 -- and returns None (the caller then draws everything and keeps its share) if
    anything differs or is missing."""

    def __init__(self, engine):
        self.sv = np.array(engine._sv)                  # (d, bits) direction numbers (scrambled)
        self.q0 = np.array(engine._quasi)               # state before the first step (the scramble's shift)
        self.scale = float(engine._scale)
        self.bits = int(self.sv.shape[1])

    def points(self, idx):
        idx = np.asarray(idx, dtype=np.uint64)
        gray = idx ^ (idx >> np.uint64(1))
        q = np.broadcast_to(self.q0, (len(idx), len(self.q0))).copy()
        for b in range(self.bits):
            rows = np.nonzero((gray >> np.uint64(b)) & np.uint64(1))[0]
            if len(rows):
                q[rows] ^= self.sv[:, b]
        return q * self.scale

    @classmethod
    def make(cls, engine):
        import copy
        try:
            if engine.num_generated != 0:
                return None
            me = cls(engine)
            probe = copy.deepcopy(engine)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                head = probe.random(67)
                probe.fast_forward(4099 - 67)       # (a microsecond a point and dimension: kept short)
                far = probe.random(5)
            if not (np.array_equal(me.points(np.arange(67)), head)
                    and np.array_equal(me.points(np.arange(4099, 4104)), far)):
                return None
            return me
        except Exception:
            return None


class ArgsortSource(OrderingSource):
    independent = True

    def __init__(self, p, seed, limit):
        def build():
            # the import belongs to the helper thread as well: the first `import scipy.stats` of a process takes 0.26 s
            # (1.3 s on a cold box) -- of the caller's time, while it could be inside the data reduction
            from scipy.stats.qmc import Sobol
            return Sobol(p, seed=seed)
        self._build, self._p, self._left, self._pos = _BackgroundBuild(build), p, limit, 0
        self._direct = False      # False: not made yet (only a run with several ranks needs it); None: unavailable

    @property
    def _qmc(self):
        return self._build.get()

    def _direct_points(self):
        if self._direct is False:
            self._direct = _DirectSobol.make(self._qmc) if self._pos == 0 else None
        return self._direct

    def _points(self, n):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # Sobol balance warning for n not a power of two
            self._pos += n
            return self._qmc.random(n)

    def take(self, count):
        n = int(min(count, self._left))
        self._left -= n
        if n <= 0:
            return np.empty((0, self._p), dtype=np.int32)
        return _argsort_rows(self._points(n))

    def take_share(self, count, first, rank, world):
        if world == 1:
            rows = self.take(count)
            return len(rows), rows
        n = int(min(count, self._left))
        self._left -= n
        if n <= 0:
            return 0, np.empty((0, self._p), dtype=np.int32)
        off = (rank - first) % world
        direct = self._direct_points()
        if direct is None:
            return n, _argsort_rows(self._points(n)[off::world])
        own = np.arange(self._pos + off, self._pos + n, world)
        self._fast_forward(n)
        self._pos += n
        if len(own) == 0:
            return n, np.empty((0, self._p), dtype=np.int32)
        return n, _argsort_rows(direct.points(own))

    def skip(self, count):
        n = int(min(count, self._left))
        self._left -= n
        if n > 0:
            self._fast_forward(n)
            self._pos += n

    def _fast_forward(self, n):
        self._qmc.fast_forward(n)


class PermutohedronSource(ArgsortSource):
    def __init__(self, p, seed, limit):
        if p < 2:
            raise ValueError("permutohedron sampling needs p >= 2")

        def build():
            from scipy.stats.qmc import MultivariateNormalQMC
            return MultivariateNormalQMC(np.zeros(p - 1), seed=seed, inv_transform=False), helmert_rows(p)
        self._build = _BackgroundBuild(build)
        self._p, self._left, self._pos = p, limit, 0
        self._direct = False

    @property
    def _qmc(self):
        return self._build.get()[0]

    @property
    def _basis(self):
        return self._build.get()[1]

    def take(self, count):
        return self.take_share(count, 0, 0, 1)[1]

    @staticmethod
    def _normals_from(base, mvn):
        """What MultivariateNormalQMC(inv_transform=False) makes of its engine's points, row by row: Box-Muller on the
        coordinate pairs, the first d of them, then mean and correlation (scipy.stats._qmc: _standard_normal_samples,
        _correlate).  Checked against the generator's own output before it is relied on (_direct_normals)."""
        import math
        even = np.arange(0, base.shape[-1], 2)
        rs = np.sqrt(-2 * np.log(base[:, even]))
        thetas = 2 * math.pi * base[:, 1 + even]
        t = np.stack([rs * np.cos(thetas), rs * np.sin(thetas)], -1).reshape(len(base), -1)[:, :mvn._d]
        return mvn._correlate(t)

    def _direct_normals(self):
        """The direct Sobol' points of the generator's engine (see _DirectSobol), or None: a rank's share of the QMC
        normals without drawing everybody's.  Enabled only if a rank-like selection (every 8th of the first 256
        points, and five points beyond number 4096) reproduces the generator's own rows exactly."""
        if self._direct is False:
            self._direct = None
            try:
                import copy
                mvn = self._qmc
                direct = _DirectSobol.make(mvn.engine) if self._pos == 0 and not mvn._inv_transform else None
                if direct is not None:
                    probe = copy.deepcopy(mvn)
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        head = probe.random(256)
                        probe.engine.fast_forward(4099 - 256)
                        far = probe.random(5)
                    ok = all(np.array_equal(self._normals_from(direct.points(np.arange(r, 256, 8)), mvn), head[r::8])
                             for r in (0, 3, 7))
                    ok = ok and np.array_equal(self._normals_from(direct.points(np.arange(4099, 4104)), mvn), far)
                    self._direct = direct if ok else None
            except Exception:
                self._direct = None
        return self._direct

    def take_share(self, count, first, rank, world):
        n = int(min(count, self._left))
        self._left -= n
        if n <= 0:
            return 0, np.empty((0, self._p), dtype=np.int32)
        off = (rank - first) % world
        direct = self._direct_normals() if world > 1 else None
        if direct is None:
            # the QMC normals of every ordering are drawn; the projection on the basis and the argsort -- most of the
            # cost -- only for this rank's share
            pts = self._points(n)[off::world]
        else:
            own = np.arange(self._pos + off, self._pos + n, world)
            self._qmc.engine.fast_forward(n)
            self._pos += n
            pts = self._normals_from(direct.points(own), self._qmc) if len(own) else np.empty((0, self._p - 1))
        if len(pts) == 0:
            return n, np.empty((0, self._p), dtype=np.int32)
        pts = pts / np.linalg.norm(pts, axis=1, keepdims=True)
        if world == 1:
            return n, _argsort_rows(pts @ self._basis)       # the reference's own expression (its fixture pins it)
        return n, _argsort_rows(helmert_project(pts))


def helmert_project(u):
    """u @ helmert_rows(p) for u of shape (n, p - 1), in O(p) a row instead of O(p^2): row k of the basis is c_k on
    columns 0 .. k, -(k + 1) c_k on column k + 1 and zero beyond (c_k = 1 / sqrt((k + 1)(k + 2))), so
    x_j = sum_{k >= j} u_k c_k - j c_{j-1} u_{j-1}.  A rank's few rows of a dealt batch made the matrix product the
    sampler's largest cost (and a 16-row product on a many-threaded BLAS its slowest: 95 ms against 8 for 128 rows
    here).  Equal to the product to rounding (1e-16): the orderings can differ from the one-process run's only where
    two projected coordinates coincide to the last bits."""
    u = np.asarray(u, dtype=np.float64)
    n, q = u.shape
    k = np.arange(q)
    w = u / np.sqrt((k + 1.0) * (k + 2.0))
    x = np.zeros((n, q + 1))
    x[:, :q] = np.cumsum(w[:, ::-1], axis=1)[:, ::-1]
    x[:, 1:] -= (k + 1.0) * w
    return x

    def _fast_forward(self, n):
        # MultivariateNormalQMC has no fast_forward of its own: drawing n points advances the underlying
        # Sobol stream by exactly n
        left = n
        while left > 0:
            left -= len(self._points(min(left, 4096)))
        self._pos -= n      # (_points counted them; skip adds them again)


class PrefetchedSource(OrderingSource):
    """Draws an independent source's orderings ahead of the loop that consumes them, on a helper thread.

    SciPy's Sobol points and the row-wise argsort of a chunk take as long on the host as the chunk's kernels take on
    the GPU (p = 1000: 2.9 ms per 128 orderings against 6 ms; p = 100: 25 ms for the 8192 orderings the GPU evaluates
    in 2.4 ms), and between them the driver's thread is inside GIL-free library calls.  The helper starts when the
    source is made -- in ls_spa() that is before the data reduction -- and keeps up to ``ahead`` orderings ready, drawn
    ``block`` at a time (SciPy's generators and ``argsort`` release the GIL for most of their time).  With several
    ranks it draws this rank's share only (``take_share`` of the inner source; ordering number g of the run belongs to
    rank g mod world), so the blocks it draws need not line up with the chunks the loop asks for.  `take_share` hands
    out exactly what the inner source would have: a QMC sequence continues across calls whatever their sizes
    (tests/test_host_logic.py::test_samplers_match_fixtures).  Only for sources whose stream nobody else reads
    (``independent``)."""
    independent = True

    def __init__(self, inner, block=256, ahead=None, rank=0, world=1):
        assert inner.independent and 0 <= rank < world
        p = max(1, int(getattr(inner, "_p", 1)))
        if ahead is None:
            ahead = max(2 * block, min(8192, (64 << 20) // (8 * p)))     # at most 64 MB of orderings waiting
        self._inner, self._block, self._ahead, self._p = inner, int(block), int(ahead), p
        self._rank, self._world = int(rank), int(world)
        self._cv = threading.Condition()
        self._parts, self._ready = [], 0       # [first number, count, own rows] drawn and not yet handed out
        self._drawn = self._taken = 0          # orderings of the run drawn by the helper / handed to the consumer
        self._done = self._stop = False
        self._error = None
        # until the consumer has asked for the first time -- in ls_spa() that is while the data reduction streams the
        # caller's arrays through the host's memory system -- the helper stops at ONE block: a run that ends at its first
        # check needs no more, and drawing 8192 orderings beside the reduction cost it 12 ms of a 47 ms call (measured)
        # (small problems draw everything at once: a megabyte a block, no contention to speak of)
        self._asked = (8 * p * self._block) <= (1 << 20)
        self._thread = threading.Thread(target=self._work, daemon=True)
        self._thread.start()

    def skip(self, count):
        # the helper is already drawing: the orderings before a checkpoint's position are taken and dropped
        left = int(count)
        while left > 0:
            got = self.take_share(min(left, 4096), self._taken, self._rank, self._world)[0]
            if got == 0:
                break
            left -= got

    def _work(self):
        try:
            while True:
                with self._cv:
                    while self._ready >= (self._ahead if self._asked else self._block) and not self._stop:
                        self._cv.wait()
                    if self._stop:
                        return
                n, own = self._inner.take_share(self._block, self._drawn, self._rank, self._world)
                with self._cv:
                    if n:
                        self._parts.append([self._drawn, n, own])
                        self._ready += n
                        self._drawn += n
                    if n < self._block:
                        self._done = True
                    self._cv.notify_all()
                    if self._done:
                        return
        except BaseException as exc:      # handed to the consumer
            with self._cv:
                self._error, self._done = exc, True
                self._cv.notify_all()

    def _own_in(self, lo, hi):
        """How many orderings numbered lo <= g < hi belong to this rank."""
        first = lo + (self._rank - lo) % self._world
        return 0 if first >= hi else (hi - 1 - first) // self._world + 1

    def take_share(self, count, first, rank, world):
        if (rank, world) != (self._rank, self._world) or first != self._taken:
            raise ValueError("the prefetching source was made for another rank, or orderings were taken out of turn")
        out, need, got = [], int(count), 0
        with self._cv:
            if not self._asked:
                self._asked = True
                self._cv.notify_all()
            while need > 0:
                while not self._parts and not self._done:
                    self._cv.wait()
                if self._error is not None:
                    raise self._error
                if not self._parts:
                    break
                lo, n, rows = self._parts[0]
                use = min(n, need)
                k = self._own_in(lo, lo + use)
                out.append(rows[:k])
                if use == n:
                    self._parts.pop(0)
                else:
                    self._parts[0] = [lo + use, n - use, rows[k:]]
                need -= use
                got += use
                self._ready -= use
                self._cv.notify_all()
            self._taken += got
        if not out:
            return 0, np.empty((0, self._p), dtype=np.int32)
        return got, (out[0] if len(out) == 1 else np.concatenate(out))

    def take(self, count):
        if self._world != 1:
            raise ValueError("a rank's prefetching source hands out shares: use take_share")
        return self.take_share(count, self._taken, 0, 1)[1]

    def close(self):
        with self._cv:
            self._stop = True
            self._cv.notify_all()
        self._thread.join(timeout=5)


_sobol_numbers = {}       # (p, seed) -> _DirectSobol of SciPy's engine for them (checked against the engine when made)


class NativeArgsortSource(OrderingSource):
    """The 'argsort' source drawn ahead of the loop by a thread of the library (include/lsspa.h, lsspa_sampler_*):
    Sobol' points by SciPy's own recurrence and their row argsort with no interpreter in the way -- the Python helper of
    PrefetchedSource shares the interpreter lock with the driver's thread, and the public call of a small problem waited
    for orderings a third of its time.  SciPy's engine still defines the stream: it is built here (on a helper thread,
    under the caller's engine creation), its direction numbers, first state and scale are read off it (_DirectSobol)
    and checked against its own output before they are used; rows with equal keys come back marked and are sorted by
    numpy from their points, so every row is what ArgsortSource hands out
    (tests/test_host_logic.py::test_native_argsort_source_is_the_python_one).  ``make`` returns None -- the caller
    then takes PrefetchedSource(ArgsortSource) -- if the library or the engine's private attributes are not there."""
    independent = True

    def __init__(self, p, seed, limit, block, ahead, rank, world):
        self._p, self._rank, self._world, self._taken = int(p), int(rank), int(world), 0
        self._h = self._lib = self._N = self._direct = self._fallback = None
        self._args = (p, seed, limit, block, ahead)
        limit = int(min(limit, 2 ** 62))

        def build():
            import ctypes as C
            from . import _native as N
            lib = N.load()
            # SciPy's constructor (the scramble) takes 3-8 ms at p = 100 and tens at p = 1000; its numbers are a function
            # of (p, seed): a process that calls again with the same ones -- an experiment's runs, a benchmark's
            # repetitions -- reads them off the first call's engine
            key = (int(p), int(seed)) if isinstance(seed, (int, np.integer)) and not isinstance(seed, bool) else None
            direct = _sobol_numbers.get(key) if key is not None else None
            if direct is None:
                from scipy.stats.qmc import Sobol
                direct = _DirectSobol.make(Sobol(p, seed=seed))
                if direct is None:
                    raise RuntimeError("SciPy's Sobol engine does not match its direct form")
                if key is not None:
                    if len(_sobol_numbers) >= 8:
                        _sobol_numbers.pop(next(iter(_sobol_numbers)))
                    _sobol_numbers[key] = direct
            sv = np.ascontiguousarray(direct.sv, dtype=np.uint64)
            q0 = np.ascontiguousarray(direct.q0, dtype=np.uint64)
            import os
            threads = max(1, min(4, (os.cpu_count() or 2) // 4))      # producers: each draws and sorts whole blocks
            small = (8 * p * block) <= (1 << 20)          # as PrefetchedSource: one block until first asked, unless small
            h = C.c_void_p()
            rc = lib.lsspa_sampler_create(int(p), direct.bits, sv.ctypes.data_as(C.POINTER(C.c_uint64)),
                                          q0.ctypes.data_as(C.POINTER(C.c_uint64)), direct.scale, limit, int(block),
                                          int(ahead), int(ahead if small else block), threads, int(rank), int(world),
                                          C.byref(h))
            if rc != 0:
                raise RuntimeError("lsspa_sampler_create failed: " + (lib.lsspa_last_error(None) or b"").decode())
            return lib, N, direct, h

        self._build = _BackgroundBuild(build)

    @classmethod
    def make(cls, p, seed, limit, block=256, ahead=None, rank=0, world=1):
        try:
            from . import _native
            if not hasattr(_native.load(), "lsspa_sampler_create"):
                return None
        except Exception:
            return None
        if ahead is None:
            ahead = max(2 * block, min(8192, (64 << 20) // (8 * max(1, p))))
        return cls(p, seed, limit, block, ahead, rank, world)

    def _ready(self):
        if self._h is None:
            if self._build is None:
                raise ValueError("the source has been closed")
            self._lib, self._N, self._direct, self._h = self._build.get()
        return self._lib

    def usable(self):
        """False if the background build failed (the caller falls back to the Python source)."""
        try:
            self._ready()
            return True
        except Exception:
            return False

    def _python_source(self):
        """The Python source instead (SciPy's engine no longer matches its direct form, or the sampler could not be made):
        only ever chosen before the first ordering has been handed out."""
        if self._fallback is None:
            p, seed, limit, block, ahead = self._args
            self._fallback = PrefetchedSource(ArgsortSource(p, seed, limit), block=block, ahead=ahead, rank=self._rank,
                                              world=self._world)
        return self._fallback

    def take_share(self, count, first, rank, world):
        if (rank, world) != (self._rank, self._world) or first != self._taken:
            raise ValueError("the native source was made for another rank, or orderings were taken out of turn")
        import ctypes as C
        if self._fallback is None and self._h is None:
            try:
                self._ready()
            except ValueError:
                raise
            except Exception:
                self._python_source()
        if self._fallback is not None:
            n, own = self._fallback.take_share(count, first, rank, world)
            self._taken += n
            return n, own
        lib = self._lib
        count = int(count)
        cap = count // self._world + 2
        out = np.empty((cap, self._p), dtype=np.int32)
        pos, ids = np.empty(cap, dtype=np.int64), np.empty(cap, dtype=np.int64)
        n_taken, n_own, n_redo = C.c_int64(), C.c_int64(), C.c_int64()
        N = self._N
        rc = lib.lsspa_sampler_take(self._h, count, N.iptr(out), cap, C.byref(n_taken), C.byref(n_own),
                                    pos.ctypes.data_as(N._pi64), ids.ctypes.data_as(N._pi64), C.byref(n_redo))
        if rc != 0:
            raise RuntimeError("lsspa_sampler_take failed: " + (lib.lsspa_last_error(None) or b"").decode())
        k = n_redo.value
        if k:       # equal keys in a row: numpy's order among them is its own -- its argsort of the row's points
            out[pos[:k]] = np.argsort(self._direct.points(ids[:k]), axis=1)
        self._taken += n_taken.value
        return n_taken.value, out[:n_own.value]

    def take(self, count):
        if self._world != 1:
            raise ValueError("a rank's source hands out shares: use take_share")
        return self.take_share(count, self._taken, 0, 1)[1]

    def skip(self, count):
        left = int(count)
        while left > 0:
            got = self.take_share(min(left, 4096), self._taken, self._rank, self._world)[0]
            if got == 0:
                break
            left -= got

    def close(self):
        build, self._build = self._build, None
        if self._fallback is not None:
            self._fallback.close()
        if build is None:
            return                 # closed before
        try:
            if self._h is None:
                self._lib, self._N, self._direct, self._h = build.get()
        except Exception:
            return
        h, self._h = self._h, None
        if h is not None:
            self._lib.lsspa_sampler_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def exact_source(p):
    return IterableSource(itertools.permutations(range(p)), p)
