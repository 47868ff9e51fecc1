set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_la
mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "lanes or launch_collect or lookahead" > $O/t.log 2>&1 || (tail -40 $O/t.log; exit 1)
tail -3 $O/t.log
python3 bench.py --steps 20 --warmup 5 --no-ttt --no-cpu-baseline > $O/c3.json 2> $O/c3.err
python3 bench.py --steps 40 --warmup 4 --no-ttt --no-cpu-baseline --no-probe --batch-size 16 > $O/c3_b16_la4.json 2> $O/c3_b16_la4.err
python3 bench.py --steps 40 --warmup 4 --no-ttt --no-cpu-baseline --no-probe --batch-size 16 --lookahead 1 > $O/c3_b16_la1.json 2> $O/c3_b16_la1.err
python3 bench.py --steps 40 --warmup 8 --no-ttt --no-cpu-baseline --no-probe --batch-size 16 --lookahead 8 > $O/c3_b16_la8.json 2> $O/c3_b16_la8.err
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02_la/*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), json.dumps(d.get('strong_scaling_probe',{}))[:400])
PY
