#!/bin/bash
# Round-5 evidence for the error branch after the occupancy work (one gpurun call); everything under gpurun_out/r05f/
set -u
R=$GRAFT_REPO_ROOT/gpurun_out/r05f
mkdir -p $R
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
say() { echo "[$(date +%T)] $*" | tee -a $R/progress.log; }
kstats() { f=$(find gpurun_out/prof_$1 -name "*kernel_stats.csv" | xargs ls -t | head -1); cp "$f" $R/$2; }
say stamps;       for m in indel sub; do GMG_LIB_PATH=glimmer-mg_amd/lib/variants/libgmg_ewstamps.so timeout -k 10 300 python3 tools/ew_stamps.py $m >> $R/ew_stamps.txt 2>> $R/misc.err; done
say err-trace;    BENCH_OWN_TABLE=1 BENCH_ERR=indel bash tools/prof_kernels.sh r05ferr python3 tests/bench/bench_mg.py 1000000 3 ragged > $R/err_trace.log 2>&1
kstats r05ferr mgerr_indel_kernel_stats.csv; f=$(find gpurun_out/prof_r05ferr -name "*kernel_trace.csv" | head -1); python3 tools/mg_timeline.py $f > $R/mgerr_timeline_indel.txt
say sub-trace;    BENCH_OWN_TABLE=1 BENCH_ERR=sub bash tools/prof_kernels.sh r05fsub python3 tests/bench/bench_mg.py 1000000 3 ragged > $R/sub_trace.log 2>&1
kstats r05fsub mgerr_sub_kernel_stats.csv; f=$(find gpurun_out/prof_r05fsub -name "*kernel_trace.csv" | head -1); python3 tools/mg_timeline.py $f > $R/mgerr_timeline_sub.txt
say err-pmc;      GMG_MG_ONE_STREAM=1 BENCH_OWN_TABLE=1 BENCH_ERR=indel bash tools/pmc_kernels.sh r05ferrp "k_mg_err_wcount<false,k_mg_err_wcount<true" python3 tests/bench/bench_mg.py 1000000 2 ragged > $R/err_pmc.log 2>&1
cp "gpurun_out/prof_r05ferrp/summary_k_mg_err_wcount<false.txt" $R/mgerr_pmc_summary_k_mg_err_wcount_count.txt; cp "gpurun_out/prof_r05ferrp/summary_k_mg_err_wcount<true.txt" $R/mgerr_pmc_summary_k_mg_err_wcount_write.txt
say err-modes;    for e in indel sub; do for w in 1 2 0; do GMG_MG_ERR_WAVE=$w BENCH_OWN_TABLE=1 BENCH_ERR=$e timeout -k 10 300 python3 tests/bench/bench_mg.py 1000000 5 ragged >> $R/mgerr_modes.jsonl 2>> $R/misc.err; done; done
say done
