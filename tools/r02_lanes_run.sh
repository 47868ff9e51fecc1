set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_lanes
mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "lanes or launch_collect or failed_alloc" > $O/t.log 2>&1 || (tail -30 $O/t.log; exit 1)
tail -3 $O/t.log
for L in 1 2; do
python3 bench.py --steps 20 --warmup 5 --no-ttt --no-cpu-baseline --lanes $L > $O/c3_l$L.json 2> $O/c3_l$L.err
python3 bench.py --steps 40 --warmup 5 --no-ttt --no-cpu-baseline --no-probe --batch-size 16 --lanes $L > $O/c3_b16_l$L.json 2> $O/c3_b16_l$L.err
python3 bench.py --steps 40 --warmup 5 --no-ttt --no-cpu-baseline --no-probe --p 100 --rows 10000 --lanes $L > $O/c2_l$L.json 2> $O/c2_l$L.err
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02_lanes/*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), d.get('strong_scaling_probe',{}).get('ms_per_step'))
PY
