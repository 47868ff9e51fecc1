# timing-only variants at the C5 shape (p = 5000, fp32): tools/variants_bench.sh for the other benchmark configuration
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/variants_c5; mkdir -p $O
L=ls-spa_amd/lib
cp $L/liblsspa_hip.so $L/keep.so
for v in ${VARIANTS:-BASE}; do
  cp $L/var/$v.so $L/liblsspa_hip.so
  timeout -k 10 300 python3 bench.py --p 5000 --rows 200000 --dtype f32 --steps 4 --warmup 2 --no-probe --no-ttt --no-cpu-baseline --no-sustained --no-full-pass > $O/$v.json 2> $O/$v.err || { tail -5 $O/$v.err; cp $L/keep.so $L/liblsspa_hip.so; exit 1; }
  python3 - $v $O/$v.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print("%-9s ms/step %.1f " % (sys.argv[1], d["ms_per_step"]), " ".join("%s=%.1f" % (k, v["ms_per_step"]) for k, v in d["kernels"].items()), flush=True)
PY
done
cp $L/keep.so $L/liblsspa_hip.so
