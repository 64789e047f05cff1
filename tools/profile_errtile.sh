#!/bin/bash
# kernel trace + PMC passes of the error branch tile by tile: tools/profile_errtile.sh <tag> [indel|sub] [reads]  -> gpurun_out/prof_errtile_<tag>/
TAG=$1; MODE=${2:-indel}; N=${3:-1000000}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_errtile_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export BENCH_OWN_TABLE=1 BENCH_ERR=$MODE BENCH_NO_CPU=1
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 tests/bench/bench_mg.py $N 3 ragged > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
head -16 "$f" | cut -d, -f1-5 | cut -c1-150
timeout -k 10 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_SALU --output-format csv -d "$OUT/pmc1" -- python3 tests/bench/bench_mg.py $N 1 ragged > "$OUT/pmc1.log" 2>&1 || { tail -5 "$OUT/pmc1.log"; exit 1; }
timeout -k 10 240 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_SMEM --output-format csv -d "$OUT/pmc2" -- python3 tests/bench/bench_mg.py $N 1 ragged > "$OUT/pmc2.log" 2>&1 || { tail -5 "$OUT/pmc2.log"; exit 1; }
for k in k_mg_err_tile k_mg_walk_prefix "k_mg_err_level"; do echo "== $k"; python3 tools/summarize_pmc.py "$OUT" $k | grep -v "other kernel"; done
