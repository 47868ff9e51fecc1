set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_pmc
mkdir -p $O
B="bench.py --steps 6 --warmup 2 --no-ttt --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -o c3 -- python3 $B > $O/stats_c3.json 2> $O/stats_c3.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_c3 -o c3 -- python3 $B > $O/fetch_c3.json 2> $O/fetch_c3.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_c3 -o c3 -- python3 $B > $O/write_c3.json 2> $O/write_c3.err
echo c3 pmc done
C5="bench.py --steps 3 --warmup 1 --no-ttt --no-cpu-baseline --p 5000 --rows 200000 --dtype f32"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -o c5 -- python3 $C5 > $O/stats_c5.json 2> $O/stats_c5.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_c5 -o c5 -- python3 $C5 > $O/fetch_c5.json 2> $O/fetch_c5.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_c5 -o c5 -- python3 $C5 > $O/write_c5.json 2> $O/write_c5.err
echo c5 pmc done
find $O -name "*.csv" | head -30
