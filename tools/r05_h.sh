#!/bin/bash
O=gpurun_out/r05h; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_native_comm.py tests/test_gpu_bench_ranks.py -x -q -m gpu 2>&1 | tee $O/t.log | tail -15
