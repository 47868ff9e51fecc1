#!/bin/bash
# the whole GPU suite, progress into a file (gpurun kills a run that is silent for 7 minutes)
O=gpurun_out/r05suite
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tee $O/gputest.log | tail -25
rc=${PIPESTATUS[0]}
if [ $rc -eq 0 ]; then
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err && python - <<'PY'
import json
d=json.load(open('gpurun_out/r05suite/bench_c3.json'))
print('C3', d['value'], d['ms_per_step'], 'full_run', d['full_run']['orderings_per_s'], d['full_run']['seconds'])
print('e2e device', json.dumps(d['time_to_tolerance_e2e']['device']))
PY
fi
exit $rc
