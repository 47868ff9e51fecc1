# A/B of two builds of the library on ONE box: ls-spa_amd/lib/ab/{old,new}.so, alternating processes.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/ablib; mkdir -p $O
L=ls-spa_amd/lib
cp $L/liblsspa_hip.so $L/keep.so
for r in 1 2 3; do
  for v in old new; do
    cp $L/ab/$v.so $L/liblsspa_hip.so
    timeout -k 10 200 python3 tools/perf_probe.py ${1:-1000} ${2:-100000} 128 4 > $O/${v}_$r.log 2>&1 || (tail -20 $O/${v}_$r.log; exit 1)
    echo $v $r $(grep "batch of" $O/${v}_$r.log | sed 's/.*: //') $(grep -E "^\s+(gather|chol_panel|strip|lift|chol_diag)" $O/${v}_$r.log | awk '{printf "%s=%s ", $1, $2}')
  done
done
cp $L/keep.so $L/liblsspa_hip.so
