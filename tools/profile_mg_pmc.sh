# PMC counters of the glimmer-mg front half's kernels (gpurun): tools/profile_mg_pmc.sh <tag> [env assignments...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
OUT=gpurun_out/prof_mg_pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_INSTS_SALU --output-format csv -d $OUT/pmc1 -- python3 tests/bench/bench_mg.py 1000000 1 > $OUT/pmc1.log 2>&1
timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY --output-format csv -d $OUT/pmc2 -- python3 tests/bench/bench_mg.py 1000000 1 > $OUT/pmc2.log 2>&1
python3 tools/summarize_pmc.py $OUT k_mg_tile_starts
