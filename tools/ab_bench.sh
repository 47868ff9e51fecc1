# A/B of two builds of the library on ONE box (boxes of the pool differ by ~1.5 %): ls-spa_amd/lib/ab/{old,new}.so,
# alternating processes of bench.py (throughput line only).   bash tools/ab_bench.sh [bench.py arguments]
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_bench; mkdir -p $O
L=ls-spa_amd/lib
cp $L/liblsspa_hip.so $L/keep.so
for r in $(seq 1 ${AB_ROUNDS:-3}); do
  for v in old new; do
    cp $L/ab/$v.so $L/liblsspa_hip.so
    timeout -k 10 300 python3 bench.py --steps 40 --warmup 5 --no-probe --no-ttt --no-cpu-baseline --no-sustained "$@" > $O/${v}_$r.json 2> $O/${v}_$r.err || (tail -20 $O/${v}_$r.err; cp $L/keep.so $L/liblsspa_hip.so; exit 1)
    python3 - $v $r $O/${v}_$r.json <<'PY'
import json, sys
d = json.load(open(sys.argv[3]))
print(sys.argv[1], sys.argv[2], "ms/step %.4f" % d["ms_per_step"], " ".join("%s=%.4f" % (k, v["ms_per_step"]) for k, v in d["kernels"].items()))
PY
  done
done
cp $L/keep.so $L/liblsspa_hip.so
