#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gram" 2>&1 | tail -3
for shape in "1000 100000" "500 100000" "2000 100000" "1500 60000" "3000 50000" "777 100000" "5000 200000" "1000 100000"; do
  timeout -k 10 200 python3 tools/gram_variants.py $shape 2>/dev/null || echo "failed"
done
