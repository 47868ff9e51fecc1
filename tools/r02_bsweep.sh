cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02_bsweep
for B in 16 32 48 64 128; do
python3 bench.py --steps 30 --warmup 5 --no-ttt --no-cpu-baseline --no-probe --batch-size $B > gpurun_out/r02_bsweep/b$B.json 2>/dev/null
done
python3 tools/perf_probe.py 1000 100000 16 5 0,8,4,12 > gpurun_out/r02_bsweep/flags_b16.log 2>&1
python3 - <<'PY'
import json,glob
for B in (16,32,48,64,128):
    d=json.load(open(f'gpurun_out/r02_bsweep/b{B}.json')); print(B, round(d['value']), round(d['ms_per_step'],3), {k:round(v['ms_per_step'],3) for k,v in d['kernels'].items()})
PY
tail -12 gpurun_out/r02_bsweep/flags_b16.log
