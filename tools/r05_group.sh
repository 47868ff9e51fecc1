# the grouped statistics / estimator of small problems: tests, then the C2 public call and bench line
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_group; mkdir -p $O
timeout -k 10 800 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 200 python3 tools/full_run_probe.py 100 10000 64 > $O/full_run_c2.log 2>&1; cat $O/full_run_c2.log
timeout -k 10 300 python3 bench.py --steps 64 --warmup 8 --p 100 --rows 10000 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r05_group/bench_c2.json"))
f = d.get("full_run", {})
print("C2", d["value"], d["ms_per_step"], "full_run", f.get("orderings_per_s"), f.get("fraction_of_value"), d["roofline"]["avg_launch_ms"])
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -o c2 -- python3 bench.py --steps 64 --warmup 8 --p 100 --rows 10000 --no-probe --no-ttt --no-cpu-baseline --no-sustained --no-full-pass --no-full-run > $O/stats_c2.json 2> $O/stats_c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fr2 -o fr2 -- python3 tools/full_run_probe.py 100 10000 64 > $O/stats_fr2.log 2>&1
