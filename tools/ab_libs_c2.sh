# A/B of two builds of the library on ONE box at the C2 shape: ls-spa_amd/lib/ab/{old,new}.so, alternating processes.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/ablib; mkdir -p $O
L=ls-spa_amd/lib
cp $L/liblsspa_hip.so $L/keep.so
for r in 1 2 3; do
  for v in old new; do
    cp $L/ab/$v.so $L/liblsspa_hip.so
    timeout -k 10 200 python3 bench.py --steps 80 --warmup 16 --p 100 --rows 10000 --no-ttt --no-cpu-baseline --no-probe > $O/c2_${v}_$r.json 2> $O/c2_${v}_$r.err || (tail -20 $O/c2_${v}_$r.err; exit 1)
    python3 -c "import json;d=json.load(open('$O/c2_${v}_$r.json'));print('$v',$r,round(d['value']),round(d['ms_per_step'],4),{k:round(v['avg_launch_ms'],4) for k,v in d['kernels'].items()})"
  done
done
cp $L/keep.so $L/liblsspa_hip.so
