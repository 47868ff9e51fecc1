set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_small
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "small_p" > $O/t.log 2>&1 || (tail -60 $O/t.log; exit 1)
tail -3 $O/t.log
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/tall.log 2>&1 || (tail -60 $O/tall.log; exit 1)
tail -3 $O/tall.log
python3 bench.py --steps 40 --warmup 5 --no-ttt --no-cpu-baseline --p 100 --rows 10000 > $O/c2.json 2> $O/c2.err
python3 bench.py --steps 40 --warmup 8 --no-ttt --no-cpu-baseline --no-probe --p 100 --rows 10000 --lookahead 8 > $O/c2_la8.json 2> $O/c2_la8.err
python3 tools/perf_probe.py 100 10000 128 20 > $O/pp_b128.log 2>&1
python3 tools/perf_probe.py 100 10000 2048 5 > $O/pp_b2048.log 2>&1
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02_small/*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), {k:round(v['ms_per_step'],4) for k,v in d['kernels'].items()}, d['roofline']['kernel'], round(d['roofline']['frac'],4))
PY
tail -8 $O/pp_b128.log; tail -8 $O/pp_b2048.log
