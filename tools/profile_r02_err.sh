# kernel trace of the error branch (-i and -s), 1M ragged reads: per-kernel times and the timeline of one call
set -u
R=$GRAFT_REPO_ROOT/gpurun_out/r02err
rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in indel sub; do
  BENCH_ERR=$m timeout -k 10 300 python3 tests/bench/bench_mg.py 1000000 3 ragged >> $R/bench.jsonl 2>> $R/bench.err
  export BENCH_ERR=$m
  bash tools/prof_kernels.sh r02err_$m python3 tests/bench/bench_mg.py 1000000 3 ragged > $R/trace_$m.txt 2>&1
  f=$(find gpurun_out/prof_r02err_$m -name "*kernel_trace.csv" | head -1); python3 tools/mg_timeline.py $f > $R/timeline_$m.txt
  cp $(find gpurun_out/prof_r02err_$m -name "*kernel_stats.csv" | head -1) $R/kernel_stats_$m.csv
done
echo done
