"""Generate the golden fixtures in this directory from the REAL reference.

Run only in the build container (the reference does not travel):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 OPENBLAS_NUM_THREADS=1 \
        python /root/repo/tests/golden/make_golden.py

It imports cvxgrp/ls-spa from /root/reference, feeds it seeded inputs and stores
inputs + outputs as small .npz files.  Fixtures are data only (arrays); no text of
the reference is stored.  Tests never import the reference.
"""
import os
import sys

import numpy as np
from scipy.stats.qmc import MultivariateNormalQMC, Sobol

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
import ls_spa as ref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path)} B")


def result_fields(r, prefix=""):
    out = {
        prefix + "attribution": r.attribution,
        prefix + "theta": r.theta,
        prefix + "overall_error": np.float64(r.overall_error),
        prefix + "attribution_errors": r.attribution_errors,
        prefix + "r_squared": np.float64(r.r_squared),
        prefix + "error_history": r.error_history,
    }
    if r.attribution_history is not None:
        out[prefix + "attribution_history"] = r.attribution_history
    return out


def small_problem(seed, p, n, m):
    rng = np.random.default_rng(seed)
    X_tr = rng.standard_normal((n, p))
    X_te = rng.standard_normal((m, p))
    w = rng.standard_normal(p)
    return X_tr, X_te, X_tr @ w + rng.standard_normal(n), X_te @ w + rng.standard_normal(m)


# (1) toy data: full result + all 6 per-ordering lift vectors
import itertools  # noqa: E402

toy = np.load("/root/reference/data/toy_data.npz")
Xa, Xe, ya, ye = (toy[k] for k in ("X_train", "X_test", "y_train", "y_test"))
res = ref.ls_spa(Xa, Xe, ya, ye)
red = ref.reduce_data(Xa, Xe, ya, ye, 0.0)
orders = np.array(list(itertools.permutations(range(3))))
lifts = np.array([ref.square_shapley(*red, np.linalg.norm(ye) ** 2, np.array(o)) for o in orders])
save("toy", X_train=Xa, X_test=Xe, y_train=ya, y_test=ye, orders=orders, lifts=lifts,
     repr=np.array(repr(res)), **result_fields(res))

# (2) exact mode p=4 and p=8
for p in (4, 8):
    d = small_problem(100 + p, p, 40, 30)
    r = ref.ls_spa(*d)
    save(f"exact_p{p}", X_train=d[0], X_test=d[1], y_train=d[2], y_test=d[3], **result_fields(r))

# (3)+(4) p=12: reduction, per-ordering lifts, full driver with injected perms
d12 = small_problem(7, 12, 60, 50)
rng = np.random.default_rng(70)
orders12 = np.array([np.arange(12), np.arange(12)[::-1]] + [rng.permutation(12) for _ in range(6)])
perms64 = np.array([rng.permutation(12) for _ in range(64)])
pack = dict(X_train=d12[0], X_test=d12[1], y_train=d12[2], y_test=d12[3], orders=orders12,
            perms64=perms64)
for tag, reg in (("r0", 0.0), ("r1", 0.1)):
    red = ref.reduce_data(*d12, reg)
    yn = np.linalg.norm(d12[3]) ** 2
    pack[f"{tag}_R_tr"], pack[f"{tag}_F_te"], pack[f"{tag}_q_tr"], pack[f"{tag}_q_te"] = red
    pack[f"{tag}_lifts"] = np.array([ref.square_shapley(*red, yn, o) for o in orders12])
for anti in (True, False):
    r = ref.ls_spa(*d12, perms=perms64, batch_size=16, tolerance=0.0, antithetical=anti,
                   return_attribution_history=True)
    pack.update(result_fields(r, f"drv_anti{int(anti)}_"))
# default seed path (lazy generator interleaved with the error estimator)
r = ref.ls_spa(*d12, max_samples=40, batch_size=16, tolerance=0.0, seed=3,
               return_attribution_history=True)
pack.update(result_fields(r, "seedpath_"))
save("p12", **pack)

# (5) + (6): gen_data / permutohedron_samples / argsort_samples live in the experiment SCRIPT, which runs its whole
# 2^19-ordering experiment when imported.  Their definitions are therefore lifted out of the script's syntax tree
# and executed here against the script's own module-level names (p, N, M, conditioning, STN_RATIO) -- the
# reference's code runs, nothing of it is stored, and the script body never starts.
import ast  # noqa: E402

_SCRIPT = "/root/reference/experiments/ground_truth_medium.py"


def script_functions(names, **module_globals):
    tree = ast.parse(open(_SCRIPT).read(), filename=_SCRIPT)
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(n.name for n in picked) == sorted(names), [n.name for n in picked]
    ns = {"np": np, **module_globals}
    exec(compile(ast.Module(body=picked, type_ignores=[]), _SCRIPT, "exec"), ns)
    return [ns[n] for n in names]


# (5) the reference's correlated generator at p=100, N=M=2000, seed 42 (ground_truth_medium.py:74-106)
p, N, M = 100, 2000, 2000
(gen_data,) = script_functions(["gen_data"], p=p, N=N, M=M, conditioning=20.0, STN_RATIO=5.0)
X_tr_c, X_te_c, y_tr_c, y_te_c, theta_c, cov_c = gen_data(np.random.default_rng(42))
dc = (X_tr_c, X_te_c, y_tr_c, y_te_c)
red = ref.reduce_data(*dc, 0.0)
yn = np.linalg.norm(dc[3]) ** 2
orders100 = np.array([np.random.default_rng(5).permutation(p) for _ in range(16)])
lifts100 = np.array([ref.square_shapley(*red, yn, o) for o in orders100])
rfull = ref.ls_spa(*dc, perms=orders100, batch_size=8, tolerance=0.0)
# store the reduced problem (4 x ~80 KB) instead of the 2000 x 100 raw data, plus Grams
save("corr_p100", R_tr=red[0], F_te=red[1], q_tr=red[2], q_te=red[3], y_norm_sq=np.float64(yn),
     orders=orders100, lifts=lifts100, attribution=rfull.attribution, theta=rfull.theta,
     r_squared=np.float64(rfull.r_squared), seed=np.int64(42), N=np.int64(N), M=np.int64(M))
# the generator's own output, pinned by slices and sums (the raw data would be 3.2 MB): what the product's
# ls_spa.workloads.correlated and the oracle's correlated_workload are checked against
save("corr_data_p100", seed=np.int64(42), p=np.int64(p), N=np.int64(N), M=np.int64(M),
     X_train_head=X_tr_c[:6], X_test_head=X_te_c[:6], X_train_tail=X_tr_c[-2:], X_test_tail=X_te_c[-2:],
     y_train_head=y_tr_c[:32], y_test_head=y_te_c[:32], theta_true=theta_c, cov_head=cov_c[:4],
     X_train_colsum=X_tr_c.sum(axis=0), X_test_colsum=X_te_c.sum(axis=0),
     X_train_sq=np.float64((X_tr_c ** 2).sum()), X_test_sq=np.float64((X_te_c ** 2).sum()),
     y_train_sq=np.float64(y_tr_c @ y_tr_c), y_test_sq=np.float64(y_te_c @ y_te_c))

# (6) sampler outputs of the script's own argsort_samples / permutohedron_samples (:56-71) at p = 12
p = 12
permutohedron_samples, argsort_samples = script_functions(["permutohedron_samples", "argsort_samples"], p=p)
exp = {}
exp["argsort"] = argsort_samples(Sobol(p, seed=5), 32)
exp["permutohedron"] = permutohedron_samples(MultivariateNormalQMC(np.zeros(p - 1), seed=5, inv_transform=False), 32)
# the projection basis is a local of permutohedron_samples: run the function once more with an identity "sample"
# and an argsort that hands its argument back, so that what comes out is  I @ U  as the reference builds it
import types  # noqa: E402


class _IdentityDraws:
    def random(self, n):
        return np.eye(n)


_np_passthrough = types.SimpleNamespace(**{k: getattr(np, k) for k in ("linalg", "tril", "ones", "diag", "arange")})
_np_passthrough.argsort = lambda a, axis=-1: a
(_probe,) = script_functions(["permutohedron_samples"], p=p)
_probe.__globals__["np"] = _np_passthrough
exp["U"] = np.array(_probe(_IdentityDraws(), p - 1))
save("samplers_p12", **exp)

# (7) merge formulas on the 200/300 split of test/test_ls_spa.py:7-44
rng = np.random.default_rng(128)
n = 20
A = rng.standard_normal((n, 3 * n))
X = rng.multivariate_normal(np.zeros(n), A @ A.T, 500)
b1, b2 = X[:200], X[200:]
mm = ref.merge_sample_mean(b1.mean(0), b2.mean(0), 200, 300)
mc = ref.merge_sample_cov(b1.mean(0), b2.mean(0), np.cov(b1, rowvar=False, bias=True),
                          np.cov(b2, rowvar=False, bias=True), 200, 300)
save("merge", X=X, merged_mean=mm, merged_cov=mc)

# (8) edge cases: M < p, float32 inputs, error_estimates on a fixed covariance
dm = small_problem(11, 12, 40, 8)
r = ref.ls_spa(*dm, perms=perms64[:16], batch_size=8, tolerance=0.0)
pack = dict(X_train=dm[0], X_test=dm[1], y_train=dm[2], y_test=dm[3], perms=perms64[:16])
pack.update(result_fields(r, "mltp_"))
d32 = tuple(a.astype(np.float32) for a in d12)
r = ref.ls_spa(*d32, perms=perms64[:16], batch_size=8, tolerance=0.0)
pack.update(result_fields(r, "f32_"))
rng = np.random.default_rng(9)
B = rng.standard_normal((12, 30))
cov_full = B @ B.T / 30 / 100
cov_low = B[:, :5] @ B[:, :5].T / 5 / 100
for tag, c in (("full", cov_full), ("low", cov_low)):
    g = np.random.default_rng(17)
    ae, oe = ref.error_estimates(g, c)
    pack[f"ee_{tag}_cov"], pack[f"ee_{tag}_feat"], pack[f"ee_{tag}_total"] = c, ae, np.float64(oe)
    pack[f"ee_{tag}_next"] = g.standard_normal(4)     # generator position after the call
save("edge", **pack)
