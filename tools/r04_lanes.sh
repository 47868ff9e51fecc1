set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_lanes; mkdir -p $O
P="--no-probe --no-ttt --no-cpu-baseline --no-sustained --steps 40 --warmup 6"
run() { # name, lanes, mid
  LSSPA_MID_LAUNCH=$3 timeout -k 10 300 python3 bench.py $P --lanes $2 > $O/$1.json 2> $O/$1.err || (tail -5 $O/$1.err; exit 1)
  python3 - $1 $O/$1.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], "ms/step %.4f" % d["ms_per_step"], d["check"])
PY
}
for r in 1 2; do
run lanes1 1 -1
for m in ${MIDS:-0 1 2 3 4}; do run lanes2_mid$m 2 $m; done
done
