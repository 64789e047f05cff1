set -u
R=$GRAFT_REPO_ROOT/gpurun_out/r02mg
rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in "" ragged; do timeout -k 10 200 python3 tests/bench/bench_mg.py 1000000 5 $m >> $R/mg.jsonl 2>> $R/mg.err; BENCH_OWN_TABLE=1 timeout -k 10 200 python3 tests/bench/bench_mg.py 1000000 5 $m >> $R/mg_own.jsonl 2>> $R/mg.err; GMG_MG_FUSED=0 timeout -k 10 200 python3 tests/bench/bench_mg.py 1000000 5 $m >> $R/mg_sequential.jsonl 2>> $R/mg.err; done
GMG_MG_TIMING=1 timeout -k 10 200 python3 tests/bench/bench_mg.py 1000000 2 2> $R/mg_stage_timing.txt > /dev/null
bash tools/prof_kernels.sh r02mgk python3 tests/bench/bench_mg.py 1000000 3 > $R/mg_trace.txt 2>&1
f=$(find gpurun_out/prof_r02mgk -name "*kernel_trace.csv" | head -1); python3 tools/mg_timeline.py $f > $R/mg_timeline.txt
cp $(find gpurun_out/prof_r02mgk -name "*kernel_stats.csv" | head -1) $R/mg_kernel_stats.csv
bash tools/prof_kernels.sh r02mgr python3 tests/bench/bench_mg.py 1000000 3 ragged > $R/mg_trace_ragged.txt 2>&1
f=$(find gpurun_out/prof_r02mgr -name "*kernel_trace.csv" | head -1); python3 tools/mg_timeline.py $f > $R/mg_timeline_ragged.txt
bash tools/profile_mg_pmc.sh final > $R/mg_pmc_fused.txt 2>&1
echo done
