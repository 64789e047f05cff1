#!/bin/bash
# Profile the six-frame kernels on the GPU box: kernel trace + PMC passes (counters in their own runs, as the
# MI355X guide prescribes).  Usage: tools/profile_frame6.sh <tag> [extra bench args]
# Writes gpurun_out/prof_<tag>/{trace,pmc1..6}/...csv and gpurun_out/prof_<tag>/summary.txt
set -u
TAG=${1:-run}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BENCH="python3 bench.py --steps 5 --warmup 2 --cpu-reads 0 --no-cli --no-extras $*"
# the kernel trace runs the DEFAULT bench command's timed region (steps 100, warmup 10; no CPU legs), so that its averages are the
# ones bench.py reports
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --cpu-reads 0 --no-cli --no-extras $* > "$OUT/trace.log" 2>&1
i=1
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM" \
  "FETCH_SIZE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
  "TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum"; do
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc$i" -- $BENCH > "$OUT/pmc$i.log" 2>&1
  i=$((i+1))
done
python3 tools/summarize_pmc.py "$OUT" > "$OUT/summary.txt" 2>&1
python3 tools/summarize_pmc.py "$OUT" k_frame6p > "$OUT/summary_k_frame6p.txt" 2>&1
cat "$OUT/summary.txt"
tail -1 "$OUT/trace.log"
