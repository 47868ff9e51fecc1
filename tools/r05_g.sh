#!/bin/bash
O=gpurun_out/r05g; mkdir -p $O; R=$PWD
P="--no-probe --no-ttt --no-cpu-baseline --no-sustained --no-full-pass --no-full-run"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/$O/l2 -o t -- python3 $R/bench.py --steps 64 --warmup 16 --p 100 --rows 10000 --lanes 2 --lookahead 8 $P > $R/$O/l2.json 2> $R/$O/l2.err
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/$O/l1 -o t -- python3 $R/bench.py --steps 64 --warmup 16 --p 100 --rows 10000 --lanes 1 --lookahead 8 $P > $R/$O/l1.json 2> $R/$O/l1.err
cd $R
python3 tools/rocpd_summary.py $O/l2/t_results.db 200 60 | tail -62
