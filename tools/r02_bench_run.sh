set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02_b1
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r02_b1/bench_c3.json 2> gpurun_out/r02_b1/bench_c3.err
echo c3 done
LSSPA_BENCH_REHEARSE_DIST=1 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02_b1/bench_c3_rehearse.json 2> gpurun_out/r02_b1/bench_c3_rehearse.err
echo rehearse done
python3 bench.py --steps 20 --warmup 5 --p 100 --rows 10000 > gpurun_out/r02_b1/bench_c2.json 2> gpurun_out/r02_b1/bench_c2.err
echo c2 done
python3 bench.py --steps 5 --warmup 2 --p 5000 --rows 200000 --dtype f32 > gpurun_out/r02_b1/bench_c5.json 2> gpurun_out/r02_b1/bench_c5.err
echo c5 done
