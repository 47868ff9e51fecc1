# Quick A/B pass on the GPU box: kernel + parity tests, then short C3 and C5 bench lines (per-kernel ms).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/ab; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q -m gpu > $O/t.log 2>&1 || (tail -40 $O/t.log; exit 1)
tail -2 $O/t.log
python3 bench.py --steps 20 --warmup 5 --no-ttt --no-cpu-baseline --no-probe > $O/c3.json 2> $O/c3.err
python3 bench.py --steps 5 --warmup 2 --no-ttt --no-cpu-baseline --no-probe --p 5000 --rows 200000 --dtype f32 > $O/c5.json 2> $O/c5.err
python3 - <<'PY'
import json
for f in ('c3','c5'):
    d=json.load(open(f'gpurun_out/ab/{f}.json')); print(f, round(d['value']), round(d['ms_per_step'],4), {k:round(v['ms_per_step'],4) for k,v in d['kernels'].items()})
PY
