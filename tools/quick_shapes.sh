cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P="--no-probe --no-ttt --no-cpu-baseline --no-sustained --no-full-pass"
python3 bench.py --steps 5 --warmup 2 --p 5000 --rows 200000 --dtype f32 $P > gpurun_out/q_c5.json 2> gpurun_out/q_c5.err
python3 bench.py --steps 20 --warmup 5 --dtype f32 --reg 0 $P > gpurun_out/q_f32.json 2> gpurun_out/q_f32.err
python3 bench.py --steps 64 --warmup 8 --p 100 --rows 10000 $P > gpurun_out/q_c2.json 2> gpurun_out/q_c2.err
python3 bench.py --steps 5 --warmup 2 --p 5000 --rows 200000 --dtype f32 --flags 512 $P > gpurun_out/q_c5_512.json 2> gpurun_out/q_c5_512.err
for f in q_c5 q_c5_512 q_f32 q_c2; do python3 -c "
import json;d=json.load(open('gpurun_out/$f.json'));print('$f', '%.3f ms/step'%d['ms_per_step'], '%.0f /s'%d['value'], ' '.join('%s=%.3f'%(k,v['ms_per_step']) for k,v in d['kernels'].items()))"; done
