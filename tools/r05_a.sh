#!/bin/bash
# round 5, first GPU pass: the running estimator's tests, then the bench lines with the full_run leg
set -o pipefail
O=gpurun_out/r05a
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "running_error or history_and_error or lookahead_on_the_device" > $O/t_kernels.log 2>&1 || { tail -30 $O/t_kernels.log; exit 1; }
tail -3 $O/t_kernels.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "device or two_lanes or large_p_against" > $O/t_parity.log 2>&1 || { tail -40 $O/t_parity.log; exit 1; }
tail -3 $O/t_parity.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err || { tail -20 $O/bench_c3.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05a/bench_c3.json'))
print('C3 value', d['value'], 'ms', d['ms_per_step'])
print('full_run', json.dumps(d.get('full_run'), indent=1))
for k in ('time_to_tolerance_e2e',):
    print(k, json.dumps(d.get(k), indent=1)[:1500])
PY
timeout -k 10 600 python bench.py --p 100 --rows 10000 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err || { tail -20 $O/bench_c2.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05a/bench_c2.json'))
print('C2 value', d['value'], 'ms', d['ms_per_step'])
print('full_run', json.dumps(d.get('full_run'), indent=1))
PY
