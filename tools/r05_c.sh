#!/bin/bash
set -o pipefail
O=gpurun_out/r05c
mkdir -p $O
R=$PWD
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "running_error or history_and or lookahead_on_the_device" > $O/t_kernels.log 2>&1 || { tail -30 $O/t_kernels.log; exit 1; }
tail -2 $O/t_kernels.log
echo "--- C2 lanes auto"; timeout -k 10 300 python3 tools/full_run_probe.py 100 10000 64 || exit 1
echo "--- C2 lanes 2"; timeout -k 10 300 python3 tools/full_run_probe.py 100 10000 64 2 || exit 1
echo "--- C3"; timeout -k 10 300 python3 tools/full_run_probe.py 1000 100000 128 || exit 1
cd /tmp && export TMPDIR=/tmp
for v in 1 2; do
  timeout -k 10 300 rocprofv3 --kernel-trace -d $R/$O/l$v -o q -- python3 $R/tools/full_run_probe.py 100 10000 64 $v > $R/$O/l$v.log 2>&1 || { tail -5 $R/$O/l$v.log; exit 1; }
  echo "== lanes $v"; grep '"rep": 2' $R/$O/l$v.log
  python3 $R/tools/rocpd_summary.py $R/$O/l$v/q_results.db > $R/$O/l$v.summary 2>&1; sed -n 1,9p $R/$O/l$v.summary
done
