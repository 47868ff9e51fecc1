# Round-5 measurement pass: bench lines, rocprofv3 kernel stats of bench.py itself, PMC traffic, MFMA ceiling.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_final; mkdir -p $O
PART=${1:-all}
P="--no-probe --no-ttt --no-cpu-baseline --no-sustained --no-full-pass --no-full-run"
if [ "$PART" = "all" ] || [ "$PART" = "a" ]; then
python3 tools/gram_time.py > $O/gram_time.log 2>&1
echo gram done
./tools/bin/lat_probe > $O/lat_probe.log 2>&1
./tools/bin/small_probe 100 2048 1 > $O/small_probe.log 2>&1
./tools/bin/small_probe 100 2048 0 >> $O/small_probe.log 2>&1
./tools/bin/small_probe 40 2048 0 >> $O/small_probe.log 2>&1
python3 tools/host_reduce_probe.py > $O/host_reduce_probe.log 2>&1
echo probes done
# the driver's command, in full (time to tolerance, CPU baseline)
python3 bench.py --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err
python3 bench.py --steps 20 --warmup 5 --data correlated --no-cpu-baseline --no-correlated-leg > $O/bench_c3_correlated.json 2> $O/bench_c3_correlated.err
python3 bench.py --steps 256 --warmup 16 --p 100 --rows 10000 > $O/bench_c2.json 2> $O/bench_c2.err
python3 bench.py --steps 5 --warmup 2 --p 5000 --rows 200000 --dtype f32 > $O/bench_c5.json 2> $O/bench_c5.err
python3 bench.py --steps 20 --warmup 5 --dtype f32 --reg 0 --no-cpu-baseline > $O/bench_c3_f32.json 2> $O/bench_c3_f32.err
python3 bench.py --steps 20 --warmup 5 --lanes 1 --no-cpu-baseline --no-ttt --no-probe > $O/bench_c3_one_lane.json 2> $O/bench_c3_one_lane.err
python3 bench.py --steps 80 --warmup 5 --no-cpu-baseline --no-ttt --no-probe > $O/bench_c3_80steps.json 2> $O/bench_c3_80steps.err
LSSPA_BENCH_REHEARSE_WORLD=2 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c3_two_ranks_one_gpu.json 2> $O/bench_c3_two_ranks_one_gpu.err
LSSPA_BENCH_REHEARSE_DIST=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c3_rehearse.json 2> $O/bench_c3_rehearse.err
LSSPA_BENCH_REHEARSE_DIST=1 python3 bench.py --steps 40 --warmup 8 --batch-size 16 --scaling strong --no-cpu-baseline --no-ttt > $O/bench_c3_strong16_rehearse.json 2> $O/bench_c3_strong16_rehearse.err
python3 tools/full_run_probe.py 100 10000 64 > $O/full_run_c2.log 2>&1
python3 tools/full_run_probe.py 1000 100000 128 > $O/full_run_c3.log 2>&1
echo bench done
fi
if [ "$PART" = "all" ] || [ "$PART" = "a" ] || [ "$PART" = "s" ]; then
# the same command under the profiler (program directly after --), without the legs that launch other shapes
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -o c3 -- python3 bench.py --steps 20 --warmup 5 $P > $O/stats_c3.json 2> $O/stats_c3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3l1 -o c3l1 -- python3 bench.py --steps 20 --warmup 5 --lanes 1 $P > $O/stats_c3l1.json 2> $O/stats_c3l1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -o c2 -- python3 bench.py --steps 256 --warmup 16 --p 100 --rows 10000 $P > $O/stats_c2.json 2> $O/stats_c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -o c5 -- python3 bench.py --steps 5 --warmup 2 --p 5000 --rows 200000 --dtype f32 $P > $O/stats_c5.json 2> $O/stats_c5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fr2 -o fr2 -- python3 tools/full_run_probe.py 100 10000 64 > $O/stats_fr2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fr3 -o fr3 -- python3 tools/full_run_probe.py 1000 100000 128 > $O/stats_fr3.log 2>&1
echo stats done
fi
if [ "$PART" = "all" ] || [ "$PART" = "b" ]; then
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_c3 -o c3 -- python3 bench.py --steps 6 --warmup 2 $P > $O/fetch_c3.json 2> $O/fetch_c3.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_c3 -o c3 -- python3 bench.py --steps 6 --warmup 2 $P > $O/write_c3.json 2> $O/write_c3.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_c5 -o c5 -- python3 bench.py --steps 3 --warmup 1 --p 5000 --rows 200000 --dtype f32 $P > $O/fetch_c5.json 2> $O/fetch_c5.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_c5 -o c5 -- python3 bench.py --steps 3 --warmup 1 --p 5000 --rows 200000 --dtype f32 $P > $O/write_c5.json 2> $O/write_c5.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_c2 -o c2 -- python3 bench.py --steps 16 --warmup 16 --p 100 --rows 10000 $P > $O/fetch_c2.json 2> $O/fetch_c2.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_c2 -o c2 -- python3 bench.py --steps 16 --warmup 16 --p 100 --rows 10000 $P > $O/write_c2.json 2> $O/write_c2.err
echo pmc done
# the Gram kernels' fabric traffic, XCD-contiguous unit map against the natural one
bash tools/gram_pmc.sh > $O/gram_pmc.log 2>&1
# per-dispatch hardware counters of a few C3 steps (matrix-pipe busy, clock, waits)
bash tools/pmc_diag.sh > $O/pmc_diag.log 2>&1 && python3 tools/pmc_diag_summary.py > $O/pmc_diag_summary.txt 2>&1
fi

if [ "$PART" = "c2" ]; then
# the C2 lines alone (after a change that touches only the small-problem path)
python3 bench.py --steps 256 --warmup 16 --p 100 --rows 10000 > $O/bench_c2.json 2> $O/bench_c2.err
python3 tools/full_run_probe.py 100 10000 64 > $O/full_run_c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -o c2 -- python3 bench.py --steps 256 --warmup 16 --p 100 --rows 10000 $P > $O/stats_c2.json 2> $O/stats_c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fr2 -o fr2 -- python3 tools/full_run_probe.py 100 10000 64 > $O/stats_fr2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_c2 -o c2 -- python3 bench.py --steps 16 --warmup 16 --p 100 --rows 10000 $P > $O/fetch_c2.json 2> $O/fetch_c2.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_c2 -o c2 -- python3 bench.py --steps 16 --warmup 16 --p 100 --rows 10000 $P > $O/write_c2.json 2> $O/write_c2.err
echo c2 done
fi
echo all done
