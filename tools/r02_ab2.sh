set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/ab2; mkdir -p $O
timeout -k 10 300 python3 tools/flag_parity.py ${1:-4096} > $O/parity.log 2>&1 || (tail -30 $O/parity.log; exit 1)
cat $O/parity.log
timeout -k 10 300 python3 tools/perf_probe.py 1000 100000 128 3 0,${1:-4096} > $O/perf.log 2>&1 || (tail -30 $O/perf.log; exit 1)
grep "flags\|batch of" $O/perf.log
