set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_small; mkdir -p $O
python3 tools/host_overhead_probe.py 100 10000 128 | tail -6
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/t3.log 2>&1 || (tail -60 $O/t3.log; exit 1)
tail -2 $O/t3.log
python3 bench.py --steps 40 --warmup 5 --no-ttt --no-cpu-baseline --p 100 --rows 10000 > $O/c2.json 2> $O/c2.err
python3 bench.py --steps 40 --warmup 8 --no-ttt --no-cpu-baseline --no-probe --p 100 --rows 10000 --lookahead 8 > $O/c2_la8.json 2> $O/c2_la8.err
python3 bench.py --steps 20 --warmup 5 --no-ttt --no-cpu-baseline --no-probe > $O/c3.json 2> $O/c3.err
python3 - <<'PY'
import json,glob
for f in ('c2','c2_la8','c3'):
    d=json.load(open(f'gpurun_out/r02_small/{f}.json')); print(f, round(d['value']), round(d['ms_per_step'],4), {k:round(v['ms_per_step'],4) for k,v in d['kernels'].items()})
PY
