# Per-dispatch hardware counters of a few C3 steps (developer diagnosis): what the panel launches are bound by.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_diag; rm -rf $O; mkdir -p $O
P="--steps 3 --warmup 1 --no-probe --no-ttt --no-cpu-baseline --no-sustained --no-full-pass --lanes 1"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/a -o a -- python3 bench.py $P > $O/a.json 2> $O/a.err
echo a done
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/b -o b -- python3 bench.py $P > $O/b.json 2> $O/b.err
echo b done
rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_DRAM TCC_HIT TCC_MISS --output-format csv -d $O/c -o c -- python3 bench.py $P > $O/c.json 2> $O/c.err
echo c done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $O/d -o d -- python3 bench.py $P > $O/d.json 2> $O/d.err
echo d done
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/e -o e -- python3 bench.py $P > $O/e.json 2> $O/e.err
echo e done
find $O -name "*counter_collection.csv" | xargs ls -la
