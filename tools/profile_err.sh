cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export BENCH_ERR=indel
OUT=gpurun_out/prof_err
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tests/bench/bench_mg.py 200000 1 ragged > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -- python3 tests/bench/bench_mg.py 200000 1 ragged > $OUT/pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/pmc2 -- python3 tests/bench/bench_mg.py 200000 1 ragged > $OUT/pmc2.log 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs head -12 | cut -c1-200
