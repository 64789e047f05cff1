cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_str
mkdir -p $OUT
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc1 -- python3 tests/bench/bench_strings.py 1000000 8 > $OUT/pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tests/bench/bench_strings.py 1000000 8 > $OUT/trace.log 2>&1
tail -2 $OUT/trace.log | cut -c1-300
