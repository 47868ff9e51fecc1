# orderings/s and algorithmic TFLOP/s at feature counts around the C3 shape (developer: does the 8 MiB matrix stride of p_pad = 1024 cost anything?)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for p in ${PS:-1000 1023 1100 1151 896}; do
timeout -k 10 300 python3 bench.py --p $p --rows 20000 --steps 30 --warmup 5 --no-probe --no-ttt --no-cpu-baseline --no-sustained --no-full-pass > gpurun_out/psweep_$p.json 2> gpurun_out/psweep_$p.err || { tail -3 gpurun_out/psweep_$p.err; continue; }
python3 -c "
import json;d=json.load(open('gpurun_out/psweep_$p.json'));p=$p
v=d['value'];print('p=%d p_pad=%d  %.3f ms/step  %.0f orderings/s  %.1f algorithmic TFLOP/s (p+1)^3  panel %.3f ms'%(p,(p+1+127)//128*128,d['ms_per_step'],v,v*(p+1)**3*1e-12,d['kernels']['chol_panel']['ms_per_step']))"
done
