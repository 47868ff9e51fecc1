#!/bin/bash
set -o pipefail
O=gpurun_out/r05c
mkdir -p $O
R=$PWD
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "running_error or history_and or lookahead_on_the_device" > $O/t_kernels.log 2>&1 || { tail -30 $O/t_kernels.log; exit 1; }
tail -2 $O/t_kernels.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "device_error or large_p_against" > $O/t_parity.log 2>&1 || { tail -30 $O/t_parity.log; exit 1; }
tail -2 $O/t_parity.log
echo "--- C2"; timeout -k 10 300 python3 tools/full_run_probe.py 100 10000 64 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/$O/l1 -o q -- python3 $R/tools/full_run_probe.py 100 10000 64 > $R/$O/l1.log 2>&1 || { tail -5 $R/$O/l1.log; exit 1; }
python3 $R/tools/rocpd_summary.py $R/$O/l1/q_results.db > $R/$O/l1.summary 2>&1; sed -n 1,9p $R/$O/l1.summary
