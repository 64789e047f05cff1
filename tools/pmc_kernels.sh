#!/bin/bash
# Kernel trace + four PMC passes of one command, summarised per kernel: tools/pmc_kernels.sh <tag> <kernel,kernel,...> <program> <args...>
#   -> gpurun_out/prof_<tag>/{trace,pmc1..4}/ and gpurun_out/prof_<tag>/summary_<kernel>.txt (tools/summarize_pmc.py)
# (counters in runs of their own, never beside a trace domain; FETCH_SIZE and WRITE_SIZE in separate passes: MI355X guide)
TAG=$1; KERNELS=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- "$@" > "$OUT/trace.log" 2>&1 || { tail -3 "$OUT/trace.log"; exit 1; }
i=1
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
  "FETCH_SIZE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc$i" -- "$@" > "$OUT/pmc$i.log" 2>&1 || tail -2 "$OUT/pmc$i.log"
  i=$((i+1))
done
for k in $(echo $KERNELS | tr , ' '); do
  python3 tools/summarize_pmc.py "$OUT" $k | grep -v "other kernel" > "$OUT/summary_$k.txt" 2>&1
  echo "== $k"; cat "$OUT/summary_$k.txt"
done
