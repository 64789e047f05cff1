#!/bin/bash
# Round-4 evidence run on the GPU box (gpurun calls of <= 20 minutes: parts a, b, c); everything under gpurun_out/r04e/;
# tools/collect_r04.sh then copies the summaries into profiles/r04_*.
set -u
PART=${1:-a}
R=$GRAFT_REPO_ROOT/gpurun_out/r04e
mkdir -p $R
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GMG_EXPECT_REF=1
say() { echo "[$(date +%T)] $*" | tee -a $R/progress_$PART.log; }
kstats() { f=$(find gpurun_out/prof_$1 -name "*kernel_stats.csv" | xargs ls -t | head -1); cp "$f" $R/$2; }
if [ $PART = a ]; then
say bench;        timeout -k 10 400 python3 bench.py > $R/bench.json 2> $R/bench.err
say frame6-prof;  timeout -k 10 900 bash tools/profile_frame6.sh r04f6 > $R/profile_frame6.log 2>&1
cp gpurun_out/prof_r04f6/summary.txt $R/pmc_summary_k_frame6t.txt; cp gpurun_out/prof_r04f6/summary_k_frame6p.txt $R/pmc_summary_k_frame6p.txt
cp $(find gpurun_out/prof_r04f6/trace -name "*kernel_stats.csv" | head -1) $R/frame6_kernel_stats.csv
python3 tools/update_traffic.py gpurun_out/prof_r04f6 > $R/traffic_update.log 2>&1; cp profiles/traffic.json $R/traffic.json
say bench-again;  timeout -k 10 400 python3 bench.py --no-extras > $R/bench_with_traffic.json 2> $R/bench2.err
say done-a
fi
if [ $PART = b ]; then
say mg-pmc;       BENCH_OWN_TABLE=1 bash tools/pmc_kernels.sh r04mg k_mg_tile_starts,k_mg_find_orfs_ev,k_frame6t,k_frame6p python3 tests/bench/bench_mg.py 1000000 3 > $R/mg_pmc.log 2>&1
for k in k_mg_tile_starts k_mg_find_orfs_ev; do cp gpurun_out/prof_r04mg/summary_$k.txt $R/mg_pmc_summary_$k.txt; done
kstats r04mg mg_kernel_stats.csv; f=$(find gpurun_out/prof_r04mg/trace -name "*kernel_trace.csv" | head -1); python3 tools/mg_timeline.py $f > $R/mg_timeline.txt
say err-pmc;      BENCH_OWN_TABLE=1 BENCH_ERR=indel bash tools/pmc_kernels.sh r04err k_mg_err_level,k_mg_walk_prefix,k_mg_run_tables python3 tests/bench/bench_mg.py 1000000 3 ragged > $R/err_pmc.log 2>&1
for k in k_mg_err_level k_mg_walk_prefix k_mg_run_tables; do cp gpurun_out/prof_r04err/summary_$k.txt $R/mgerr_pmc_summary_$k.txt; done
kstats r04err mgerr_indel_kernel_stats.csv; f=$(find gpurun_out/prof_r04err/trace -name "*kernel_trace.csv" | head -1); python3 tools/mg_timeline.py $f > $R/mgerr_timeline_indel.txt
say errtile-pmc;  GMG_MG_ERR_TILE=1 BENCH_OWN_TABLE=1 BENCH_ERR=indel bash tools/pmc_kernels.sh r04et k_mg_err_tile python3 tests/bench/bench_mg.py 1000000 3 ragged > $R/errtile_pmc.log 2>&1
cp gpurun_out/prof_r04et/summary_k_mg_err_tile.txt $R/mgerr_tile_pmc_summary_k_mg_err_tile.txt
say orfs-pmc;     bash tools/pmc_kernels.sh r04orf k_orf_walk_sums,k_orf_events python3 tests/bench/bench_orfs.py 200000 3 > $R/orfs_pmc.log 2>&1
for k in k_orf_walk_sums k_orf_events; do cp gpurun_out/prof_r04orf/summary_$k.txt $R/orfs_pmc_summary_$k.txt; done
kstats r04orf orfs_kernel_stats.csv
say done-b
fi
if [ $PART = c ]; then
say err-ab;       for e in indel sub; do for t in 0 1; do GMG_MG_ERR_TILE=$t BENCH_OWN_TABLE=1 BENCH_ERR=$e timeout -k 10 300 python3 tests/bench/bench_mg.py 1000000 5 ragged >> $R/mgerr_level_vs_tile.jsonl 2>> $R/misc.err; done; done
say mg;           for m in "" ragged; do BENCH_OWN_TABLE=1 timeout -k 10 200 python3 tests/bench/bench_mg.py 1000000 5 $m >> $R/mg_own_table.jsonl 2>> $R/misc.err; done
say classes;      for m in 1 0 distinct relabel; do BENCH_PER_GROUP_CALLS=0 BENCH_SAME_MODEL=$m timeout -k 10 300 python3 tests/bench/bench_classes.py 1000000 64 100 7 >> $R/classes_bench.jsonl 2>> $R/misc.err; done
for m in 1 distinct relabel; do BENCH_ERR=indel BENCH_PER_GROUP_CALLS=0 BENCH_SAME_MODEL=$m timeout -k 10 300 python3 tests/bench/bench_classes.py 1000000 64 100 5 >> $R/classes_bench_indel.jsonl 2>> $R/misc.err; done
say multi-pmc;    bash tools/profile_multi_pmc.sh r04 relabel > $R/multi_pmc_relabel.txt 2>&1; bash tools/profile_multi_pmc.sh r04d distinct > $R/multi_pmc_trained.txt 2>&1
say strings;      for d in 0 1 relabel; do BENCH_DISTINCT=$d timeout -k 10 300 python3 tests/bench/bench_strings.py 1000000 64 >> $R/strings_bench.jsonl 2>> $R/misc.err; done
say orfs;         timeout -k 10 300 python3 tests/bench/bench_orfs.py 200000 5 > $R/orfs_bench.json 2>> $R/misc.err
say ingest;       timeout -k 10 300 python3 tests/bench/bench_ingest.py > $R/ingest_bench.json 2>> $R/misc.err
say cli;          timeout -k 10 600 python3 tests/bench/bench_cli.py 200000 > $R/cli_bench.json 2>> $R/misc.err
say tests;        timeout -k 10 1100 python3 -m pytest tests -m gpu -q -x > $R/pytest_gpu.txt 2>&1; tail -3 $R/pytest_gpu.txt
say done-c
fi
