# timing-only variants of the library (ls-spa_amd/lib/var/*.so, results are garbage), one bench process each, on ONE box
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/variants; mkdir -p $O
L=ls-spa_amd/lib
cp $L/liblsspa_hip.so $L/keep.so
for r in 1; do
for v in ${VARIANTS:-BASE}; do
  cp $L/var/$v.so $L/liblsspa_hip.so
  for lanes in 2 1; do
    timeout -k 10 200 python3 bench.py --steps 40 --warmup 5 --no-probe --no-ttt --no-cpu-baseline --no-sustained --no-full-pass --lanes $lanes > $O/${v}_l$lanes.json 2> $O/${v}_l$lanes.err || { tail -5 $O/${v}_l$lanes.err; cp $L/keep.so $L/liblsspa_hip.so; exit 1; }
    python3 - $v $lanes $O/${v}_l$lanes.json <<'PY'
import json, sys
d = json.load(open(sys.argv[3]))
print("%-9s lanes %s  ms/step %.3f " % (sys.argv[1], sys.argv[2], d["ms_per_step"]), " ".join("%s=%.3f" % (k, v["ms_per_step"]) for k, v in d["kernels"].items()), flush=True)
PY
  done
done
done
cp $L/keep.so $L/liblsspa_hip.so
