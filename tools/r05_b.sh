#!/bin/bash
set -o pipefail
O=gpurun_out/r05b
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "running_error or lookahead_on_the_device" > $O/t_kernels.log 2>&1 || { tail -30 $O/t_kernels.log; exit 1; }
tail -2 $O/t_kernels.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "device_error or two_lanes" > $O/t_parity.log 2>&1 || { tail -40 $O/t_parity.log; exit 1; }
tail -2 $O/t_parity.log
echo "--- C2 lanes auto"; timeout -k 10 300 python3 tools/full_run_probe.py 100 10000 64 || exit 1
echo "--- C2 lanes 2"; timeout -k 10 300 python3 tools/full_run_probe.py 100 10000 64 2 || exit 1
echo "--- C3"; timeout -k 10 300 python3 tools/full_run_probe.py 1000 100000 128 || exit 1
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/$O/prof_c2 -o c2 -- python3 $R/tools/full_run_probe.py 100 10000 64 > $R/$O/prof_c2.log 2>&1 || { tail -5 $R/$O/prof_c2.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/$O/prof_c2l2 -o c2 -- python3 $R/tools/full_run_probe.py 100 10000 64 2 > $R/$O/prof_c2l2.log 2>&1 || { tail -5 $R/$O/prof_c2l2.log; exit 1; }
cd $R
python3 tools/rocpd_summary.py $O/prof_c2/c2_results.db 2>&1 | head -12
