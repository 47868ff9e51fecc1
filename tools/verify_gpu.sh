set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/verify
python -m pytest tests -x -q -m gpu > gpurun_out/verify/t.log 2>&1 || (tail -40 gpurun_out/verify/t.log; exit 1)
tail -2 gpurun_out/verify/t.log
python3 -c "import __graft_entry__ as g; g.build(); g.smoke()"
LSSPA_BENCH_REHEARSE_DIST=1 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/verify/rehearse.json 2> gpurun_out/verify/rehearse.err
LSSPA_BENCH_REHEARSE_DIST=1 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --collective torch > gpurun_out/verify/rehearse_torch.json 2> gpurun_out/verify/rehearse_torch.err
python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-ttt --lanes 2 --lookahead 2 > gpurun_out/verify/lanes2.json 2> gpurun_out/verify/lanes2.err
python3 - <<'PY'
import json
for f in ('rehearse','rehearse_torch','lanes2'):
    d=json.load(open(f'gpurun_out/verify/{f}.json')); print(f, round(d['value']), round(d['ms_per_step'],3), d['config']['collective'], {k:('err' if 'error' in v else round(v.get('seconds_sampling_loop',0),4)) for k,v in d.items() if k.startswith('time_to_tolerance') and isinstance(v,dict) and k!='time_to_tolerance_e2e'}, d.get('time_to_tolerance_e2e',{}).get('error'))
PY
