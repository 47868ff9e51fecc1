set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_trace; mkdir -p $O
P="--no-probe --no-ttt --no-cpu-baseline"
for f in ${FLAGSET:-0 128}; do
rocprofv3 --kernel-trace --output-format csv -d $O/tr_$f -o t -- python3 bench.py --steps 12 --warmup 3 $P --flags $f > $O/tr_$f.json 2> $O/tr_$f.err
CSV=$(find $O/tr_$f -name "*kernel_trace.csv" | head -1)
echo "== flags $f"; python3 tools/panel_launches.py $CSV $([ $f = 0 ] && echo 8 || echo 7)
done
