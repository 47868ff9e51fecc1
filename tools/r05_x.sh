#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=ls-spa_amd/lib
cp $L/liblsspa_hip.so $L/keep.so
P="--no-probe --no-ttt --no-cpu-baseline --no-sustained --no-full-pass --no-full-run"
for r in 1 2; do
for p in 30 40 60 90 100 110; do
  for v in old new; do
    cp $L/ab/$v.so $L/liblsspa_hip.so
    python3 bench.py --steps 256 --warmup 32 --p $p --rows 4000 $P 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('p=$p $v', round(d['value']), round(d['ms_per_step'],5), {k: round(v['ms_per_step'],4) for k,v in d['kernels'].items()})"
  done
done
done
cp $L/keep.so $L/liblsspa_hip.so
