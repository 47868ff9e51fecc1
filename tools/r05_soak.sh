#!/bin/bash
O=gpurun_out/r05_soak; mkdir -p $O
timeout -k 10 900 python3 tools/soak_full_runs.py 60 2>&1 | grep -v amdgpu.ids | tee $O/soak.log
