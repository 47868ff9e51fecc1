set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_gram; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q -m gpu -k "gram or reduc or c4 or c5 or p5000 or helpers" > $O/t.log 2>&1 || (tail -40 $O/t.log; exit 1)
tail -2 $O/t.log
python3 tools/gram_time.py > $O/gram_time.log 2>&1 || true
tail -12 $O/gram_time.log
