#!/bin/bash
set -o pipefail
O=gpurun_out/r05d
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "hand_over or vt_tiles or running_error" > $O/t_kernels.log 2>&1 || { tail -40 $O/t_kernels.log; exit 1; }
tail -2 $O/t_kernels.log
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "beyond_the_lds" > $O/t_parity.log 2>&1 || { tail -40 $O/t_parity.log; exit 1; }
tail -2 $O/t_parity.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err || { tail -20 $O/bench_c3.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05d/bench_c3.json'))
print('C3 value', d['value'], 'ms', d['ms_per_step'], 'sustained', d['sustained'] and (d['sustained']['ms_per_step'], d['sustained']['sclk_mhz'], d['sustained']['power_w'], d['sustained'].get('sysfs_error')))
print('check', d['check'])
f=d.get('full_run'); f.pop('note',None); print('full_run', json.dumps(f))
print('roofline', d['roofline']['frac'], d['roofline'].get('full_batch_one_lane',{}).get('frac'))
print('ttt e2e device', d['time_to_tolerance_e2e']['device']['seconds'])
PY
