#!/bin/bash
# PMC passes over the error branch's table kernels (one stream, so that every kernel runs alone): tools/wp_pmc.sh <tag>
set -u
TAG=${1:-wp}
OUT=gpurun_out/prof_pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GMG_MG_ONE_STREAM=1 BENCH_OWN_TABLE=1 BENCH_ERR=indel
i=1
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
  "FETCH_SIZE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
  "TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum" \
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_WRITE_sum"; do
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc$i" -- python3 tests/bench/bench_mg.py 1000000 1 ragged > "$OUT/pmc$i.log" 2>&1
  i=$((i+1))
done
for k in k_mg_walk_prefix k_mg_run_tables k_mg_quality "k_mg_err_level<false, 1"; do echo "== $k"; python3 tools/summarize_pmc.py "$OUT" "$k"; done > "$OUT/summary.txt" 2>&1
