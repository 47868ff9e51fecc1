# Round-2 measurement pass: bench lines, rocprofv3 kernel stats of bench.py itself, PMC traffic, MFMA ceiling.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_final; mkdir -p $O
./tools/bin/mfma_bench5 > $O/mfma_bench5_zero.log 2>&1
./tools/bin/mfma_bench5 random > $O/mfma_bench5_random.log 2>&1
# steady-state k-loop harnesses (random operands, >= 0.4 s of back-to-back launches per line) and the vendor's GEMMs
./tools/bin/mfma_bench6 > $O/mfma_bench6.log 2>&1
./tools/bin/mfma_bench7 > $O/mfma_bench7.log 2>&1
python3 tools/dgemm_probe.py > $O/dgemm_probe.log 2>&1
python3 tools/dgemm_probe2.py > $O/dgemm_probe2.log 2>&1
python3 tools/gram_time.py > $O/gram_time.log 2>&1
echo ceiling done
# the driver's command, in full (time to tolerance, CPU baseline)
python3 bench.py --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err
python3 bench.py --steps 40 --warmup 8 --p 100 --rows 10000 > $O/bench_c2.json 2> $O/bench_c2.err
python3 bench.py --steps 5 --warmup 2 --p 5000 --rows 200000 --dtype f32 > $O/bench_c5.json 2> $O/bench_c5.err
python3 bench.py --steps 20 --warmup 5 --dtype f32 --reg 0 --no-cpu-baseline > $O/bench_c3_f32.json 2> $O/bench_c3_f32.err
LSSPA_BENCH_REHEARSE_DIST=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c3_rehearse.json 2> $O/bench_c3_rehearse.err
LSSPA_BENCH_REHEARSE_DIST=1 python3 bench.py --steps 40 --warmup 8 --batch-size 16 --scaling strong --no-cpu-baseline --no-ttt > $O/bench_c3_strong16_rehearse.json 2> $O/bench_c3_strong16_rehearse.err
echo bench done
# the same command under the profiler (program directly after --), without the legs that launch other shapes
P="--no-probe --no-ttt --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -o c3 -- python3 bench.py --steps 20 --warmup 5 $P > $O/stats_c3.json 2> $O/stats_c3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -o c2 -- python3 bench.py --steps 40 --warmup 8 --p 100 --rows 10000 $P > $O/stats_c2.json 2> $O/stats_c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -o c5 -- python3 bench.py --steps 5 --warmup 2 --p 5000 --rows 200000 --dtype f32 $P > $O/stats_c5.json 2> $O/stats_c5.err
echo stats done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_c3 -o c3 -- python3 bench.py --steps 6 --warmup 2 $P > $O/fetch_c3.json 2> $O/fetch_c3.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_c3 -o c3 -- python3 bench.py --steps 6 --warmup 2 $P > $O/write_c3.json 2> $O/write_c3.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_c5 -o c5 -- python3 bench.py --steps 3 --warmup 1 --p 5000 --rows 200000 --dtype f32 $P > $O/fetch_c5.json 2> $O/fetch_c5.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_c5 -o c5 -- python3 bench.py --steps 3 --warmup 1 --p 5000 --rows 200000 --dtype f32 $P > $O/write_c5.json 2> $O/write_c5.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_c2 -o c2 -- python3 bench.py --steps 16 --warmup 8 --p 100 --rows 10000 $P > $O/fetch_c2.json 2> $O/fetch_c2.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_c2 -o c2 -- python3 bench.py --steps 16 --warmup 8 --p 100 --rows 10000 $P > $O/write_c2.json 2> $O/write_c2.err
echo pmc done
