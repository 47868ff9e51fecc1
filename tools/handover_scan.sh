cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for r in 1 2; do for h in 3 1 2 4 5; do
LSSPA_HANDOVER=$h timeout -k 10 200 python3 bench.py --steps 40 --warmup 5 --no-probe --no-ttt --no-cpu-baseline --no-sustained --no-full-pass > gpurun_out/ho_$h.json 2>gpurun_out/ho_$h.err
python3 -c "import json;d=json.load(open('gpurun_out/ho_$h.json'));print('handover after launch $h: ms/step %.3f'%d['ms_per_step'])"
done; done
