"""bench.py's real N > 1 path, executed: `LSSPA_BENCH_REHEARSE_WORLD=2 python bench.py --gpus 2` starts two ranks that
both open GPU 0 (gloo group, the engine's device buffers staged through the host around the all-reduce -- RCCL refuses
two ranks on one device).  Everything else is the code the 8-GPU scaling run executes: the launcher, the per-rank
dealing of the orderings, the strong-scaling second timed region, max / min over the ranks, the sharded
time-to-tolerance legs.  (The reference is single-process: ls_spa/ls_spa.py:197 is the loop that is dealt, :103-119 the
merge the all-reduce implements.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--p", "200", "--rows", "3000", "--batch-size", "16", "--steps", "4", "--warmup", "1", "--no-cpu-baseline",
        "--no-sustained", "--no-correlated-leg"]


def _bench(gpus, extra_env, timeout=900):
    env = dict(os.environ, **extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LSSPA_BENCH_LAUNCHED"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), *ARGS], env=env,
                       capture_output=True, text=True, timeout=timeout)
    return r


def test_two_ranks_of_bench_py_on_one_gpu():
    one = _bench(1, {})
    assert one.returncode == 0, one.stderr[-3000:]
    lines1 = [ln for ln in one.stdout.splitlines() if ln.strip()]
    assert len(lines1) == 1
    d1 = json.loads(lines1[0])

    # (no LSSPA_BENCH_TTT_MULTI: since round 5 the time-to-tolerance legs run with several ranks by default)
    two = _bench(2, {"LSSPA_BENCH_REHEARSE_WORLD": "2"})
    assert two.returncode == 0, two.stderr[-3000:]
    lines2 = [ln for ln in two.stdout.splitlines() if ln.strip()]
    assert len(lines2) == 1, two.stdout[-2000:]            # exactly one JSON line: rank 0's
    d2 = json.loads(lines2[0])
    assert d2["n_gpus"] == 2 and d2["torch_world"] == 2 and d2["scaling"] == "weak"
    assert "gloo" in d2["config"]["collective"]
    assert d2["config"]["global_batch"] == 2 * d1["config"]["global_batch"]
    # every rank folded its own batch into the shared statistics: twice the samples; every ordering's lifts sum to
    # the full model's R^2, so the mean attribution's sum is the one-rank run's to round-off whatever the sample set
    assert d2["check"]["samples"] == 2 * d1["check"]["samples"]
    assert abs(d2["check"]["sum_attribution"] - d1["check"]["sum_attribution"]) < 1e-12
    assert d2["ms_per_step_min_rank"] <= d2["ms_per_step"] + 1e-9
    assert d2["value"] > 0
    s = d2["strong_scaling"]                               # the second timed region: global batch dealt over the ranks
    assert s["scaling"] == "strong" and s["samples_per_rank_per_step"] == 8 and s["global_batch"] == 16
    assert s["ms_per_step_min_rank"] <= s["ms_per_step"] + 1e-9 and s["value"] > 0
    # the sharded sampling loop under the reference's stop rule ran on both ranks with each estimator
    for leg in ("time_to_tolerance", "time_to_tolerance_lowrank", "time_to_tolerance_device"):
        assert d2[leg]["samples_at_stop"] == d1[leg]["samples_at_stop"], leg
        assert abs(d2[leg]["overall_error"] - d1[leg]["overall_error"]) <= 0.3 * d1[leg]["overall_error"] + 1e-12, leg
    assert "skipped" in d2["time_to_tolerance_e2e"]
    # the many-check run: the public call on one GPU, the sharded loop with two ranks -- all 128 checks, same sample count
    f1, f2 = d1["full_run"], d2["full_run"]
    assert f1["checks"] == f2["checks"] == 129 and f1["samples"] == f2["samples"] == 16 * 128      # + the one at max - 1
    assert abs(f1["sum_attribution"] - f1["r_squared"]) < 1e-10
    assert abs(f2["sum_attribution"] - f1["sum_attribution"]) < 1e-11
    assert f1["orderings_per_s"] > 0 and f2["orderings_per_s"] > 0


def test_a_failing_rank_fails_the_job():
    r = _bench(2, {"LSSPA_BENCH_REHEARSE_WORLD": "2", "LSSPA_BENCH_FAIL_RANK": "1"}, timeout=600)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]     # no result line from a broken job
