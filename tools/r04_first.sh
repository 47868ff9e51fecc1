# GPU pass: the X-tile path against the strip kernel -- tests, then bench A/B by flag, then the probe
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_first; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "${TESTK:-vt_tiles or padding_tiles or block_boundary or lift_batch}" > $O/t_kernels.log 2>&1 || (tail -40 $O/t_kernels.log; exit 1)
tail -3 $O/t_kernels.log
P="--no-probe --no-ttt --no-cpu-baseline"
for r in 1 2; do
for f in ${FLAGSET:-0 128}; do
  timeout -k 10 300 python3 bench.py --steps 40 --warmup 5 $P --flags $f > $O/b_${f}_$r.json 2> $O/b_${f}_$r.err || (tail -20 $O/b_${f}_$r.err; exit 1)
  python3 - $f $O/b_${f}_$r.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print("flags", sys.argv[1], "ms/step %.4f" % d["ms_per_step"], " ".join("%s=%.4f" % (k, v["ms_per_step"]) for k, v in d["kernels"].items()))
PY
done
done
if [ -x tools/bin/xtile_probe ]; then timeout -k 10 120 tools/bin/xtile_probe > gpurun_out/xtile_probe.log 2>&1; grep "^Jo=" gpurun_out/xtile_probe.log; fi
