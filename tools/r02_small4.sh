set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_small; mkdir -p $O
true
true
python3 bench.py --steps 40 --warmup 5 --no-ttt --no-cpu-baseline --no-probe --p 100 --rows 10000 > $O/c2.json 2> $O/c2.err
python3 bench.py --steps 40 --warmup 8 --no-ttt --no-cpu-baseline --no-probe --p 100 --rows 10000 --lookahead 8 > $O/c2_la8.json 2> $O/c2_la8.err
python3 bench.py --steps 40 --warmup 8 --no-ttt --no-cpu-baseline --no-probe --p 100 --rows 10000 --lookahead 8 --lanes 2 > $O/c2_la8_l2.json 2> $O/c2_la8_l2.err
python3 bench.py --steps 48 --warmup 8 --no-ttt --no-cpu-baseline --no-probe --batch-size 16 --lookahead 4 --lanes 2 > $O/c3_b16_la4_l2.json 2> $O/c3_b16_la4_l2.err
python3 - <<'PY'
import json,glob
for f in ('c2','c2_la8','c2_la8_l2','c3_b16_la4_l2'):
    d=json.load(open(f'gpurun_out/r02_small/{f}.json')); print(f, round(d['value']), round(d['ms_per_step'],4), {k:round(v['ms_per_step'],4) for k,v in d['kernels'].items()})
PY
