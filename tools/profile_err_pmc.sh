cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_err_pmc
mkdir -p $OUT
BENCH_ERR=indel timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_INSTS_SALU --output-format csv -d $OUT/pmc1 -- python3 tests/bench/bench_mg.py 200000 1 ragged > $OUT/pmc1.log 2>&1
echo done
