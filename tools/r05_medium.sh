#!/bin/bash
O=gpurun_out/r05_medium; mkdir -p $O
python3 tools/first_call_probe.py 100 2>&1 | grep -v amdgpu.ids
timeout -k 10 900 python3 experiments/medium_experiment.py --out-dir $O 2>&1 | grep -v amdgpu.ids | tee $O/medium_run.log
