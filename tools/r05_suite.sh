#!/bin/bash
# the whole GPU suite, progress into a file (gpurun kills a run that is silent for 7 minutes)
O=gpurun_out/r05suite
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tee $O/gputest.log | tail -25
exit ${PIPESTATUS[0]}
