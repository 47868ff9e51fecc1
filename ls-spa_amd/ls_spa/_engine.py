"""Device engine: a thin object wrapper over the C ABI (one context = one GPU).

The driver (``_driver.py``) only talks to this interface:

    load_data / load_reduced      one-time reduction (a1)
    full_fit                      theta, r_squared (a7)
    run_batch                     lift vectors of a batch of orderings (a2, a3) and the
                                  batch's moments about the running mean (a4)
    pending_buffer / merge        the all-reduce target and the Chan merge (a4, multi-GPU)
    stats                         n, mean, biased covariance
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N


class DeviceArrayView:
    """Zero-copy view of a device buffer for consumers of ``__cuda_array_interface__``
    (torch.as_tensor on ROCm included): used to hand the pending-statistics buffer to
    torch.distributed without going through the host."""

    def __init__(self, ptr: int, count: int, owner):
        self._owner = owner  # keeps the context alive
        self.__cuda_array_interface__ = {
            "shape": (int(count),),
            "typestr": "<f8",
            "data": (int(ptr), False),
            "version": 2,
            "strides": None,
        }


class HipEngine:
    """One MI355X.  Raises LSSPANativeError when the HIP library or the GPU is missing."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self._lib = N.load()
        h = C.c_void_p()
        rc = self._lib.lsspa_create(int(device), C.byref(h))
        if rc != N.OK:
            msg = self._lib.lsspa_last_error(None)
            raise N.LSSPANativeError(f"lsspa_create(device={device}) failed: {msg.decode() if msg else rc}")
        self._h = h
        self.device = int(device)
        self.p = 0
        self.m = 0
        self.tri = False
        self.y_norm_sq = float("nan")
        self.precision = "float64"
        self.lanes = 1
        if stream is not None:
            self._check(self._lib.lsspa_set_stream(self._h, C.c_void_p(int(stream))))

    # ---- plumbing -------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != N.OK:
            msg = self._lib.lsspa_last_error(self._h)
            text = msg.decode() if msg else ""
            if rc == 1:
                raise ValueError(f"lsspa: {text}")
            if rc == 4:
                raise MemoryError(f"lsspa: {text}")
            raise N.LSSPANativeError(f"lsspa call failed (status {rc}): {text}")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lsspa_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self._check(self._lib.lsspa_synchronize(self._h))

    def _refresh_dims(self):
        p, m, tri, yy = C.c_int32(), C.c_int32(), C.c_int32(), C.c_double()
        self._check(self._lib.lsspa_get_problem(self._h, C.byref(p), C.byref(m), C.byref(tri), C.byref(yy)))
        self.p, self.m, self.tri, self.y_norm_sq = p.value, m.value, bool(tri.value), yy.value

    # ---- a1: reduction ---------------------------------------------------------------
    def load_data(self, X_train, X_test, y_train, y_test, reg: float):
        """Host ndarrays (float64 or float32, both sides alike) -> device Gram reduction."""
        dt = np.float32 if (X_train.dtype == np.float32 and X_test.dtype == np.float32) else np.float64
        Xa = np.ascontiguousarray(X_train, dtype=dt)
        Xe = np.ascontiguousarray(X_test, dtype=dt)
        ya = np.ascontiguousarray(y_train, dtype=dt)
        ye = np.ascontiguousarray(y_test, dtype=dt)
        n, p = Xa.shape
        mrows = Xe.shape[0]
        self._check(self._lib.lsspa_reduce(
            self._h, Xa.ctypes.data, p, ya.ctypes.data, n, Xe.ctypes.data, p, ye.ctypes.data, mrows, p,
            float(reg), N.F32 if dt == np.float32 else N.F64, N.HOST))
        self._refresh_dims()

    def reduce_timing(self):
        """Host seconds of the last reduction from host arrays: page-locking, streamed copies + Gram kernels,
        un-locking, finalize (include/lsspa.h, lsspa_reduce_timing)."""
        out = np.zeros(4)
        self._check(self._lib.lsspa_reduce_timing(self._h, N.dptr(out)))
        return dict(zip(("pin", "h2d_gram", "unpin", "finalize"), (float(v) for v in out)))

    def load_data_sharded(self, X_train, X_test, y_train, y_test, reg: float, comm, shard_test: bool = True):
        """Row-sharded reduction: the arrays are THIS rank's rows; one all-reduce of the Gram sums
        (2 (p+1)^2 fp64, padded) replaces moving the rows.  shard_test=False: every rank passes all the
        test rows (required when there are fewer than p of them) and rank 0 alone contributes them."""
        dt = np.float32 if (X_train.dtype == np.float32 and X_test.dtype == np.float32) else np.float64
        Xa = np.ascontiguousarray(X_train, dtype=dt)
        Xe = np.ascontiguousarray(X_test, dtype=dt)
        ya = np.ascontiguousarray(y_train, dtype=dt)
        ye = np.ascontiguousarray(y_test, dtype=dt)
        n_loc, p = Xa.shape
        m_loc = Xe.shape[0]
        n_tot, m_sum = comm.sum_ints([n_loc, m_loc])
        m_tot = m_sum if shard_test else m_loc
        if not shard_test and m_tot >= p and comm.rank != 0:
            m_loc = 0   # replicated test rows enter the Gram sum once
        self._check(self._lib.lsspa_reduce_partial(
            self._h, Xa.ctypes.data, p, ya.ctypes.data, n_loc, Xe.ctypes.data, p, ye.ctypes.data, m_loc, m_tot, p,
            N.F32 if dt == np.float32 else N.F64, N.HOST))
        comm.allreduce_reduction(self)
        self._check(self._lib.lsspa_reduce_finish(self._h, int(n_tot), float(reg)))
        self._refresh_dims()
        return n_tot, m_tot

    def reduce_partial_device(self, X_train_ptr, ld_train, y_train_ptr, n_local, X_test_ptr, ld_test, y_test_ptr,
                              m_local, m_total, p, f32=False):
        """Device-pointer form of the first step (rows already in HBM)."""
        self._check(self._lib.lsspa_reduce_partial(
            self._h, C.c_void_p(X_train_ptr), ld_train, C.c_void_p(y_train_ptr), n_local, C.c_void_p(X_test_ptr),
            ld_test, C.c_void_p(y_test_ptr), m_local, m_total, p, N.F32 if f32 else N.F64, N.DEVICE))

    def reduce_buffer(self) -> DeviceArrayView:
        ptr, cnt = C.c_void_p(), C.c_int64()
        self._check(self._lib.lsspa_reduce_buffer(self._h, C.byref(ptr), C.byref(cnt)))
        return DeviceArrayView(ptr.value, cnt.value, self)

    def reduce_finish(self, n_total: int, reg: float):
        self._check(self._lib.lsspa_reduce_finish(self._h, int(n_total), float(reg)))
        self._refresh_dims()

    def load_device_data(self, X_train_ptr, ld_train, y_train_ptr, n, X_test_ptr, ld_test, y_test_ptr, m_rows,
                         p, reg, f32=False):
        """Same, from device pointers (e.g. torch tensors' data_ptr())."""
        self._check(self._lib.lsspa_reduce(
            self._h, C.c_void_p(X_train_ptr), ld_train, C.c_void_p(y_train_ptr), n, C.c_void_p(X_test_ptr),
            ld_test, C.c_void_p(y_test_ptr), m_rows, p, float(reg), N.F32 if f32 else N.F64, N.DEVICE))
        self._refresh_dims()

    def load_reduced(self, G, g, aug_train, y_norm_sq, H=None, h=None, Ft=None, ytil=None):
        G = np.ascontiguousarray(G, dtype=np.float64)
        g = np.ascontiguousarray(g, dtype=np.float64)
        p = G.shape[0]
        if H is not None:
            H = np.ascontiguousarray(H, dtype=np.float64)
            h = np.ascontiguousarray(h, dtype=np.float64)
            self._check(self._lib.lsspa_set_reduced(self._h, p, N.dptr(G), N.dptr(g), float(aug_train), 1,
                                                    N.dptr(H), N.dptr(h), p, None, None, float(y_norm_sq)))
        else:
            Ft = np.ascontiguousarray(Ft, dtype=np.float64)
            ytil = np.ascontiguousarray(ytil, dtype=np.float64)
            self._check(self._lib.lsspa_set_reduced(self._h, p, N.dptr(G), N.dptr(g), float(aug_train), 0,
                                                    None, None, Ft.shape[1], N.dptr(Ft), N.dptr(ytil),
                                                    float(y_norm_sq)))
        self._refresh_dims()

    def gram(self):
        p = self.p
        G, g = np.empty((p, p)), np.empty(p)
        if self.tri:
            H, h = np.empty((p, p)), np.empty(p)
            self._check(self._lib.lsspa_get_gram(self._h, N.dptr(G), N.dptr(g), N.dptr(H), N.dptr(h)))
            return G, g, H, h
        self._check(self._lib.lsspa_get_gram(self._h, N.dptr(G), N.dptr(g), None, None))
        return G, g, None, None

    # ---- a7 ----------------------------------------------------------------------------
    def full_fit(self):
        theta = np.empty(self.p)
        r2, info = C.c_double(), C.c_int32()
        self._check(self._lib.lsspa_full_fit(self._h, N.dptr(theta), C.byref(r2), C.byref(info)))
        return theta, r2.value, info.value

    def factors(self):
        p, m = self.p, self.m
        R, q = np.empty((p, p)), np.empty(p)
        F, qt = np.empty((m, p)), np.empty(m)
        self._check(self._lib.lsspa_get_factors(self._h, N.dptr(R), N.dptr(q), N.dptr(F), N.dptr(qt)))
        return R, F, q, qt

    # ---- a2 / a3 / a4 ------------------------------------------------------------------
    def run_batch(self, perms, antithetical: bool, want_lifts: bool = False, accumulate: bool = True):
        perms = np.ascontiguousarray(perms, dtype=np.int32)
        if perms.ndim != 2 or perms.shape[1] != self.p:
            raise ValueError(f"perms must have shape (B, {self.p})")
        B = perms.shape[0]
        out = np.empty((B, self.p)) if want_lifts else None
        self._check(self._lib.lsspa_lift_batch(self._h, N.iptr(perms), B, int(bool(antithetical)),
                                               N.dptr(out), self._acc_mode(accumulate)))
        return out

    @staticmethod
    def _acc_mode(accumulate):
        """False / True / 2: nothing, into the pending buffer (all-reduce + merge() follow), or -- one GPU -- folded
        into the running statistics at once (no merge() call; include/lsspa.h)."""
        if isinstance(accumulate, (bool, np.bool_)):
            return int(accumulate)
        return int(accumulate)      # integers go through as they are: the library refuses anything but 0, 1, 2

    def set_lanes(self, n: int):
        """1: batches run one after the other (default).  2: successive batches alternate between two workspaces on
        two streams (include/lsspa.h, lsspa_set_lanes)."""
        self._check(self._lib.lsspa_set_lanes(self._h, int(n)))
        self.lanes = int(n)

    def launch_batch(self, perms, antithetical: bool):
        """Enqueue a batch up to its lift vectors; returns a ticket for collect_batch / discard_batch."""
        perms = np.ascontiguousarray(perms, dtype=np.int32)
        if perms.ndim != 2 or perms.shape[1] != self.p:
            raise ValueError(f"perms must have shape (B, {self.p})")
        t = C.c_int32()
        self._check(self._lib.lsspa_lift_launch(self._h, N.iptr(perms), perms.shape[0], int(bool(antithetical)),
                                                C.byref(t)))
        return (t.value, perms.shape[0])

    def collect_batch(self, ticket, want_lifts: bool = False, accumulate: bool = True, first: int = 0,
                      count: int | None = None):
        """Accumulate (and / or fetch) ``count`` samples of a launched batch starting at sample ``first`` (default:
        all of it).  Parts are taken front to back."""
        t, B = ticket
        count = B - first if count is None else int(count)
        out = np.empty((count, self.p)) if want_lifts else None
        self._check(self._lib.lsspa_lift_collect(self._h, t, int(first), count, N.dptr(out),
                                                 self._acc_mode(accumulate)))
        return out

    def collect_chunks(self, ticket, first: int, chunk: int, n_chunks: int, accumulate=2):
        """n_chunks consecutive parts of `chunk` samples of a launched batch, folded one after the other (include/lsspa.h,
        lsspa_lift_collect_chunks: one statistics launch for a small problem's parts, each still merged by itself)."""
        self._check(self._lib.lsspa_lift_collect_chunks(self._h, ticket[0], int(first), int(chunk), int(n_chunks),
                                                        self._acc_mode(accumulate)))

    def discard_batch(self, ticket):
        self._check(self._lib.lsspa_lift_discard(self._h, ticket[0]))

    def info(self) -> int:
        v = C.c_int32()
        self._check(self._lib.lsspa_get_info(self._h, C.byref(v)))
        return v.value

    def info_collected(self) -> int:
        """The info bits of every batch collected so far, without waiting for batches launched and dropped."""
        v = C.c_int32()
        self._check(self._lib.lsspa_get_info_collected(self._h, C.byref(v)))
        return v.value

    def sum_deviation(self) -> float:
        """Largest |sum of a sample's lifts - R^2| of all batches since the last reset (0 before full_fit)."""
        v = C.c_double()
        self._check(self._lib.lsspa_get_sum_deviation(self._h, C.byref(v)))
        return v.value

    def reset_stats(self):
        self._check(self._lib.lsspa_stats_reset(self._h))

    def pending_buffer(self) -> DeviceArrayView:
        ptr, cnt = C.c_void_p(), C.c_int64()
        self._check(self._lib.lsspa_stats_pending(self._h, C.byref(ptr), C.byref(cnt)))
        return DeviceArrayView(ptr.value, cnt.value, self)

    def merge(self):
        self._check(self._lib.lsspa_stats_merge(self._h))

    def stats(self, want_cov: bool = True):
        n = C.c_int64()
        mean = np.empty(self.p)
        cov = np.empty((self.p, self.p)) if want_cov else None
        self._check(self._lib.lsspa_stats_get(self._h, C.byref(n), N.dptr(mean), N.dptr(cov)))
        return n.value, mean, cov

    def set_stats(self, n: int, mean, cov_biased):
        """Restore running statistics saved from ``stats()`` (checkpoint / resume)."""
        mean = np.ascontiguousarray(mean, dtype=np.float64)
        cov = np.ascontiguousarray(cov_biased, dtype=np.float64)
        if mean.shape != (self.p,) or cov.shape != (self.p, self.p):
            raise ValueError("mean / covariance shapes do not match the loaded problem")
        self._check(self._lib.lsspa_stats_set(self._h, int(n), N.dptr(mean), N.dptr(cov)))

    # ---- lift history + device-side error estimator ---------------------------------------
    def history_enable(self, capacity: int):
        self._check(self._lib.lsspa_history_enable(self._h, int(capacity)))

    def history_count(self) -> int:
        n = C.c_int64()
        self._check(self._lib.lsspa_history_get(self._h, C.byref(n), None))
        return n.value

    def history(self):
        """All accumulated samples' lift vectors, (count, p)."""
        out = np.empty((self.history_count(), self.p))
        n = C.c_int64()
        self._check(self._lib.lsspa_history_get(self._h, C.byref(n), N.dptr(out)))
        return out

    def history_append(self, lifts):
        lifts = np.ascontiguousarray(lifts, dtype=np.float64)
        if lifts.ndim != 2 or lifts.shape[1] != self.p:
            raise ValueError("lifts must be (rows, p)")
        self._check(self._lib.lsspa_history_append(self._h, N.dptr(lifts), lifts.shape[0]))

    def error_draws(self, xi_local, n_total: int):
        """xi_local: (1024, n_local) standard normals for this engine's history rows."""
        xi_local = np.ascontiguousarray(xi_local, dtype=np.float64)
        if xi_local.ndim != 2 or xi_local.shape[0] != 1024:
            raise ValueError("xi must be (1024, n_local)")
        nl = xi_local.shape[1]
        self._check(self._lib.lsspa_error_draws(self._h, N.dptr(xi_local) if nl else None, max(nl, 1), nl,
                                                int(n_total)))

    def draws_buffer(self) -> DeviceArrayView:
        ptr, cnt = C.c_void_p(), C.c_int64()
        self._check(self._lib.lsspa_error_buffer(self._h, C.byref(ptr), C.byref(cnt)))
        return DeviceArrayView(ptr.value, cnt.value, self)

    def error_quantiles(self):
        feat, tot = np.empty(self.p), C.c_double()
        self._check(self._lib.lsspa_error_quantiles(self._h, N.dptr(feat), C.byref(tot)))
        return feat, tot.value

    # ---- running form of the device-side estimator (include/lsspa.h) ---------------------
    RESULT_SLOTS = 64
    SMALL_P_MAX = 127      # csrc/k_small.hip small_p_eligible: p + 1 <= 128 takes the one-workgroup-per-ordering kernels

    def error_running_enable(self, seed: int):
        """D = Xi L and s = Xi 1 stay in HBM; Xi is a function of (seed, sample id, draw)."""
        self._check(self._lib.lsspa_error_running_enable(self._h, int(seed) & (2 ** 64 - 1)))

    def error_advance(self, first_id: int, stride: int = 1):
        """Fold the samples accumulated since the last call in: they are samples first_id, first_id + stride, ..."""
        self._check(self._lib.lsspa_error_advance(self._h, int(first_id), int(stride)))

    def error_running_draws(self, n_total: int):
        self._check(self._lib.lsspa_error_running_draws(self._h, int(n_total)))

    def error_quantiles_enqueue(self, slot: int):
        self._check(self._lib.lsspa_error_quantiles_enqueue(self._h, int(slot)))

    def error_check_enqueue(self, n_total: int, slot: int):
        """One rank: draws and quantiles of a check in one call (the draws are evaluated as they are read)."""
        self._check(self._lib.lsspa_error_check_enqueue(self._h, int(n_total), int(slot)))

    def error_result(self, slot: int, wait: bool = True):
        """(feature_errors, overall_error, mean, n) of a slot, or None if wait is False and it is not there yet."""
        feat, mean = np.empty(self.p), np.empty(self.p)
        tot, n, ready = C.c_double(), C.c_int64(), C.c_int32()
        self._check(self._lib.lsspa_error_result(self._h, int(slot), int(bool(wait)), C.byref(ready), N.dptr(feat),
                                                 C.byref(tot), N.dptr(mean), C.byref(n)))
        if not ready.value:
            return None
        return feat, tot.value, mean, n.value

    def group_collect(self, ticket, first, count, first_id, stride, n_after, slot):
        """The chunks of a launched batch in one call (include/lsspa.h, lsspa_group_collect): per chunk collect,
        (all-reduce + merge), fold into the running estimator and -- where n_after > 0 -- enqueue its check."""
        first = np.ascontiguousarray(first, dtype=np.int32)
        count = np.ascontiguousarray(count, dtype=np.int32)
        first_id = np.ascontiguousarray(first_id, dtype=np.int64)
        n_after = np.ascontiguousarray(n_after, dtype=np.int64)
        slot = np.ascontiguousarray(slot, dtype=np.int32)
        k = len(first)
        assert len(count) == len(first_id) == len(n_after) == len(slot) == k
        self._check(self._lib.lsspa_group_collect(
            self._h, ticket[0], k, N.iptr(first), N.iptr(count), first_id.ctypes.data_as(N._pi64), int(stride),
            n_after.ctypes.data_as(N._pi64), N.iptr(slot)))

    def error_state(self):
        D, s = np.empty((1024, self.p)), np.empty(1024)
        self._check(self._lib.lsspa_error_state_get(self._h, N.dptr(D), N.dptr(s)))
        return D, s

    def set_error_state(self, D, s):
        D = np.ascontiguousarray(D, dtype=np.float64)
        s = np.ascontiguousarray(s, dtype=np.float64)
        if D.shape != (1024, self.p) or s.shape != (1024,):
            raise ValueError("D must be (1024, p) and s (1024,)")
        self._check(self._lib.lsspa_error_state_set(self._h, N.dptr(D), N.dptr(s)))

    def error_xi(self, seed: int, first_id: int, stride: int, count: int):
        """Test hook: the estimator's normals of `count` sample ids, (1024, count)."""
        out = np.empty((1024, int(count)))
        self._check(self._lib.lsspa_error_xi(self._h, int(seed) & (2 ** 64 - 1), int(first_id), int(stride), int(count),
                                             N.dptr(out)))
        return out

    # ---- profiling / test hooks ---------------------------------------------------------
    def profile(self, on: bool):
        self._check(self._lib.lsspa_profile_enable(self._h, int(on)))

    def profile_reset(self):
        self._check(self._lib.lsspa_profile_reset(self._h))

    def profile_read(self):
        out = {}
        for k, name in enumerate(N.KERNEL_CLASSES):
            ms, cnt = C.c_double(), C.c_int64()
            self._check(self._lib.lsspa_profile_get(self._h, k, C.byref(ms), C.byref(cnt)))
            out[name] = (ms.value, cnt.value)
        return out

    def set_flags(self, flags: int):
        self._check(self._lib.lsspa_set_flags(self._h, int(flags)))

    def set_precision(self, dtype):
        """'float64' (default) or 'float32' for the per-ordering factorisation work."""
        name = np.dtype(dtype).name
        if name not in ("float64", "float32"):
            raise ValueError("precision must be float64 or float32")
        self._check(self._lib.lsspa_set_precision(self._h, N.F32 if name == "float32" else N.F64))
        self.precision = name

    def debug_fail_alloc(self, nth: int):
        """Test hook: the nth device allocation from now fails with MemoryError (0 disarms)."""
        self._check(self._lib.lsspa_debug_fail_alloc(self._h, int(nth)))

    def mfma_probe(self, A, B, f32=False):
        A = np.ascontiguousarray(A, dtype=np.float64)
        B = np.ascontiguousarray(B, dtype=np.float64)
        D = np.empty((16, 16))
        self._check(self._lib.lsspa_mfma_probe(self._h, N.dptr(A), N.dptr(B), N.dptr(D), int(bool(f32))))
        return D

    def debug_factor(self, perm):
        pp, mp, vr = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self._lib.lsspa_debug_factor(self._h, None, None, None, None, C.byref(pp), C.byref(mp),
                                                 C.byref(vr)))
        perm = np.ascontiguousarray(perm, dtype=np.int32)
        L = np.empty((pp.value, pp.value))
        Lt = np.empty((pp.value, pp.value)) if self.tri else None
        V = np.empty((vr.value, mp.value))
        self._check(self._lib.lsspa_debug_factor(self._h, N.iptr(perm), N.dptr(L), N.dptr(Lt), N.dptr(V),
                                                 C.byref(pp), C.byref(mp), C.byref(vr)))
        return L, Lt, V
