set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02_base
./tools/bin/mfma_bench5 > gpurun_out/r02_base/mfma_bench5_zero.log 2>&1
./tools/bin/mfma_bench5 random > gpurun_out/r02_base/mfma_bench5_random.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/r02_base/prof_bench -o bench -- python3 bench.py --steps 20 --warmup 5 --no-ttt --no-cpu-baseline > gpurun_out/r02_base/bench_prof.json 2> gpurun_out/r02_base/bench_prof.err
python3 tools/perf_probe.py 1000 100000 16 10 > gpurun_out/r02_base/pp_1k_b16.log 2>&1
python3 tools/perf_probe.py 1000 100000 128 5 > gpurun_out/r02_base/pp_1k_b128.log 2>&1
python3 tools/perf_probe.py 100 10000 128 20 > gpurun_out/r02_base/pp_100_b128.log 2>&1
python3 tools/perf_probe.py 100 10000 2048 5 > gpurun_out/r02_base/pp_100_b2048.log 2>&1
echo done
