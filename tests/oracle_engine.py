"""Test double with the HipEngine interface, backed by the CPU oracle.

Lets the host logic (driver loop, trigger indices, generator interleave, multi-rank
sharding) be tested without a GPU.  TESTS ONLY -- the product never imports this."""
import numpy as np

import lsspa_oracle as O
import philox_ref


class OracleEngine:
    def __init__(self):
        self.p = 0
        self.calls = []
        self.lanes = 1
        self.launched, self.discarded = 0, 0
        self._tickets = {}
        self._taken = {}

    # two-lane interface (lsspa_lift_launch / _collect / _discard): a launched batch is evaluated at once here;
    # what matters to the driver logic is WHEN it enters the statistics, and that a discarded one never does
    def set_lanes(self, n):
        self.lanes = int(n)

    def launch_batch(self, perms, antithetical):
        assert len(self._tickets) < 2, "both lanes busy"
        perms = np.asarray(perms)
        lifts = np.array([O.sample_lift(*self._red, self.y_norm_sq, o, antithetical) for o in perms])
        self.launched += 1
        t = self.launched
        self._tickets[t] = lifts
        return t

    def collect_chunks(self, ticket, first, chunk, n_chunks, accumulate=2):
        for c in range(n_chunks):
            self.collect_batch(ticket, want_lifts=False, accumulate=accumulate, first=first + c * chunk, count=chunk)

    def collect_batch(self, ticket, want_lifts=False, accumulate=True, first=0, count=None):
        assert ticket == min(self._tickets), "tickets are collected in launch order"
        done = self._taken.get(ticket, 0)
        assert first == done, "parts are taken front to back"
        lifts_all = self._tickets[ticket]
        count = len(lifts_all) - first if count is None else count
        lifts = lifts_all[first:first + count]
        self._taken[ticket] = first + count
        if first + count == len(lifts_all):
            del self._tickets[ticket]
        self.calls.append(len(lifts))
        if accumulate:
            self._accumulate(lifts)
            if accumulate is not True and accumulate == 2:
                self.merge()
        return lifts if want_lifts else None

    def discard_batch(self, ticket):
        self._tickets.pop(ticket)
        self.discarded += 1

    def load_data(self, Xa, Xe, ya, ye, reg):
        Xa, Xe, ya, ye = (np.asarray(a, dtype=np.float64) for a in (Xa, Xe, ya, ye))
        self._data = (Xa, Xe, ya, ye, reg)
        self._red = O.reduce(Xa, Xe, ya, ye, reg)
        self.p = Xa.shape[1]
        self.m = self._red[1].shape[0]
        self.tri = Xe.shape[0] >= self.p
        self.y_norm_sq = float(np.linalg.norm(ye) ** 2)
        self.reset_stats()

    def load_data_sharded(self, Xa, Xe, ya, ye, reg, comm, shard_test=True):
        """Gram sums of this rank's rows, one all-reduce, factors from the summed Gram (Cholesky)."""
        Xa, Xe, ya, ye = (np.asarray(a, dtype=np.float64) for a in (Xa, Xe, ya, ye))
        p = Xa.shape[1]
        n_tot, m_sum = comm.sum_ints([len(Xa), len(Xe)])
        m_tot = m_sum if shard_test else len(Xe)
        Za, Ze = np.column_stack([Xa, ya]), np.column_stack([Xe, ye])
        if not shard_test and comm.rank != 0:
            Ze = Ze[:0]
        self._cred = np.concatenate([(Za.T @ Za).ravel(), (Ze.T @ Ze).ravel()])
        comm.allreduce_reduction(self)
        Ca, Ce = self._cred.reshape(2, p + 1, p + 1)
        G = Ca[:p, :p] / n_tot + reg * np.eye(p)
        R = np.linalg.cholesky(G).T
        q = np.linalg.solve(R.T, Ca[:p, p] / n_tot)
        if m_tot >= p:
            F = np.linalg.cholesky(Ce[:p, :p]).T
            qt = np.linalg.solve(F.T, Ce[:p, p])
            self.y_norm_sq = float(Ce[p, p])
        else:
            F, qt = Xe, ye
            self.y_norm_sq = float(ye @ ye)
        self._red = (R, F, q, qt)
        self.p, self.m, self.tri = p, F.shape[0], m_tot >= p
        self.reset_stats()

    def reduce_buffer(self):
        return self._cred

    def gram(self):
        R, F, q, qt = self._red
        return R.T @ R, R.T @ q, F.T @ F, F.T @ qt

    def reset_stats(self):
        p = self.p
        self._n, self._mean, self._M2 = 0, np.zeros(p), np.zeros((p, p))
        self._pend = np.zeros(1 + p + p * p)
        self._hist = [] if getattr(self, "_hist_on", False) else None
        if getattr(self, "_run_on", False):
            self._D, self._s, self._staged = np.zeros((1024, p)), np.zeros(1024), []

    def run_batch(self, perms, antithetical, want_lifts=False, accumulate=True):
        perms = np.asarray(perms)
        self.calls.append(len(perms))
        lifts = np.array([O.sample_lift(*self._red, self.y_norm_sq, o, antithetical) for o in perms])
        if accumulate:
            self._accumulate(lifts)
            if accumulate is not True and accumulate == 2:
                self.merge()
        return lifts if want_lifts else None

    def _accumulate(self, lifts):
        p = self.p
        D = lifts - self._mean
        self._pend[0] += len(D)
        self._pend[1:1 + p] += D.sum(0)
        self._pend[1 + p:] += (D.T @ D).ravel()
        if self._hist is not None:
            self._hist.append(lifts)
        if getattr(self, "_run_on", False):
            self._staged.append(lifts)

    def pending_buffer(self):
        return self._pend

    def merge(self):
        p = self.p
        nb = self._pend[0]
        if nb > 0:
            delta = self._pend[1:1 + p] / nb
            Q = self._pend[1 + p:].reshape(p, p)
            n = self._n
            self._M2 += Q + (n * nb / (n + nb) - nb) * np.outer(delta, delta)
            self._mean = self._mean + self._pend[1:1 + p] / (n + nb)
            self._n = int(round(n + nb))
        self._pend[:] = 0

    def stats(self, want_cov=True):
        cov = self._M2 / self._n if (want_cov and self._n) else (self._M2.copy() if want_cov else None)
        return self._n, self._mean.copy(), cov

    def set_stats(self, n, mean, cov_biased):
        self._n, self._mean, self._M2 = int(n), np.array(mean, dtype=float), np.array(cov_biased) * n
        self._pend[:] = 0

    # lift history + thin-form error estimator (mirrors lsspa_history_* / lsspa_error_*)
    def history_enable(self, capacity):
        if capacity == 0:
            self._run_on = False
        self._hist_on = capacity > 0
        self._hist = [] if self._hist_on else None
        self._draws = np.zeros(1024 * self.p)

    def history(self):
        return np.concatenate(self._hist) if self._hist else np.zeros((0, self.p))

    def history_count(self):
        return len(self.history())

    def history_append(self, lifts):
        self._hist.append(np.array(lifts, dtype=float))

    def error_draws(self, xi_local, n_total):
        H = self.history()
        assert xi_local.shape == (1024, len(H))
        with np.errstate(divide="ignore", invalid="ignore"):
            d = (xi_local @ (H - self._mean)) / np.sqrt(n_total * (n_total - 1.0)) if len(H) else 0.0
        self._draws[:] = np.broadcast_to(d, (1024, self.p)).ravel()

    def draws_buffer(self):
        return self._draws

    def error_quantiles(self):
        d = self._draws.reshape(1024, self.p)
        return np.quantile(np.abs(d), 0.95, axis=0), float(np.quantile(np.linalg.norm(d, axis=1), 0.95))

    # running form (mirrors lsspa_error_running_* / _advance / _quantiles_enqueue / _result): D = Xi L, s = Xi 1 with the
    # counter-based normals of tests/philox_ref.py; a check's results are computed when it is enqueued and handed out
    # when its slot is read
    RESULT_SLOTS = 64

    def error_running_enable(self, seed):
        self._run_on, self._run_seed = True, int(seed)
        self._D, self._s = np.zeros((1024, self.p)), np.zeros(1024)
        self._staged, self._slots = [], {}
        self._draws = np.zeros(1024 * self.p)
        self.enqueued = 0

    def error_advance(self, first_id, stride=1):
        if not self._staged:
            return
        L = np.concatenate(self._staged)
        self._staged = []
        xi = philox_ref.normals(self._run_seed, first_id + stride * np.arange(len(L)))
        self._D += xi @ L
        self._s += xi.sum(axis=1)

    def error_running_draws(self, n_total):
        assert not self._staged, "advance first"
        with np.errstate(divide="ignore", invalid="ignore"):
            d = (self._D - np.outer(self._s, self._mean)) / np.sqrt(n_total * (n_total - 1.0))
        self._draws[:] = d.ravel()

    def group_collect(self, ticket, first, count, first_id, stride, n_after, slot):
        """lsspa_group_collect on one rank: per chunk collect + fold + (where due) enqueue the check."""
        self.group_calls = getattr(self, "group_calls", 0) + 1
        for f, c, fid, na, sl in zip(first, count, first_id, n_after, slot):
            if c > 0:
                self.collect_batch(ticket, want_lifts=False, accumulate=2, first=f, count=c)
                self.error_advance(fid, stride)
            if na > 0:
                self.error_running_draws(na)
                self.error_quantiles_enqueue(sl)

    def error_quantiles_enqueue(self, slot):
        feat, tot = self.error_quantiles()
        self._slots[slot] = (feat, tot, self._mean.copy(), self._n)
        self.enqueued += 1

    def error_result(self, slot, wait=True):
        return self._slots[slot]

    def error_state(self):
        return self._D.copy(), self._s.copy()

    def set_error_state(self, D, s):
        self._D, self._s = np.array(D, dtype=float), np.array(s, dtype=float)

    def full_fit(self):
        R, F, q, qt = self._red
        theta = np.linalg.lstsq(R, q, rcond=None)[0]
        r2 = (np.linalg.norm(qt) ** 2 - np.linalg.norm(qt - F @ theta) ** 2) / self.y_norm_sq
        return theta, float(r2), 0

    def info(self):
        return 0

    def info_collected(self):
        return 0

    def close(self):
        pass
