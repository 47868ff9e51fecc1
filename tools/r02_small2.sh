set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_small; mkdir -p $O
./tools/bin/small_probe 100 256 | tail -3
./tools/bin/small_probe 100 4096 | tail -2
./tools/bin/small_probe 126 256 | tail -2
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q -m gpu > $O/t2.log 2>&1 || (tail -60 $O/t2.log; exit 1)
tail -3 $O/t2.log
