#!/bin/bash
# Round-3 evidence run on the GPU box (one gpurun call): everything under gpurun_out/r03/; tools/collect_r03.sh then copies the
# summaries into profiles/r03_*.  Every step under its own timeout; progress appended as it comes (a silent box is taken for hung).
set -u
R=$GRAFT_REPO_ROOT/gpurun_out/r03
rm -rf $R; mkdir -p $R
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GMG_EXPECT_REF=1
say() { echo "[$(date +%T)] $*" | tee -a $R/progress.log; }
kstats() { f=$(find gpurun_out/prof_$1 -name "*kernel_stats.csv" | xargs ls -t | head -1); cp "$f" $R/$2; }
say bench;        timeout -k 10 400 python3 bench.py > $R/bench.json 2> $R/bench.err
say bench-genome; timeout -k 10 400 python3 bench.py --data genome --no-cli > $R/bench_genome.json 2> $R/bench_genome.err
say frame6-prof;  timeout -k 10 1200 bash tools/profile_frame6.sh r03f6 > $R/profile_frame6.log 2>&1
cp gpurun_out/prof_r03f6/summary.txt $R/pmc_summary_k_frame6t.txt; cp gpurun_out/prof_r03f6/summary_k_frame6p.txt $R/pmc_summary_k_frame6p.txt
cp $(find gpurun_out/prof_r03f6/trace -name "*kernel_stats.csv" | head -1) $R/frame6_kernel_stats.csv
say f6-data-ab;   timeout -k 10 300 python3 tools/f6_data_ab.py > $R/f6_data_ab.json 2>> $R/misc.err
say mg;           for m in "" ragged; do BENCH_OWN_TABLE=1 timeout -k 10 200 python3 tests/bench/bench_mg.py 1000000 5 $m >> $R/mg_own_table.jsonl 2>> $R/mg.err; timeout -k 10 200 python3 tests/bench/bench_mg.py 1000000 5 $m >> $R/mg_callers_table.jsonl 2>> $R/mg.err; BENCH_NULLS=100 timeout -k 10 200 python3 tests/bench/bench_mg.py 1000000 5 $m >> $R/mg_nulls.jsonl 2>> $R/mg.err; done
say mg-trace;     BENCH_OWN_TABLE=1 bash tools/prof_kernels.sh r03mg python3 tests/bench/bench_mg.py 1000000 3 > /dev/null 2>&1; kstats r03mg mg_kernel_stats.csv
f=$(find gpurun_out/prof_r03mg -name "*kernel_trace.csv" | head -1); python3 tools/mg_timeline.py $f > $R/mg_timeline.txt
say mg-err;       for e in indel sub; do BENCH_OWN_TABLE=1 BENCH_ERR=$e timeout -k 10 300 python3 tests/bench/bench_mg.py 1000000 5 ragged >> $R/mgerr.jsonl 2>> $R/mg.err; done
BENCH_OWN_TABLE=1 BENCH_ERR=indel bash tools/prof_kernels.sh r03err python3 tests/bench/bench_mg.py 1000000 3 ragged > /dev/null 2>&1; kstats r03err mgerr_indel_kernel_stats.csv
f=$(find gpurun_out/prof_r03err -name "*kernel_trace.csv" | head -1); python3 tools/mg_timeline.py $f > $R/mgerr_timeline_indel.txt
say classes;      BENCH_PER_GROUP_CALLS=1 timeout -k 10 300 python3 tests/bench/bench_classes.py 1000000 64 100 7 > $R/classes_bench.json 2>> $R/misc.err
BENCH_PER_GROUP_CALLS=0 BENCH_SAME_MODEL=0 timeout -k 10 300 python3 tests/bench/bench_classes.py 1000000 64 100 7 > $R/classes_bench_five_models.json 2>> $R/misc.err
BENCH_PER_GROUP_CALLS=0 bash tools/prof_kernels.sh r03cls python3 tests/bench/bench_classes.py 1000000 64 100 3 > /dev/null 2>&1; kstats r03cls classes_kernel_stats.csv
BENCH_ERR=indel BENCH_PER_GROUP_CALLS=0 timeout -k 10 300 python3 tests/bench/bench_classes.py 1000000 64 100 5 > $R/classes_bench_indel.json 2>> $R/misc.err
say cli-classes;  timeout -k 10 400 python3 tests/bench/bench_cli_classes.py 50000 > $R/cli_classes.json 2>> $R/misc.err; BENCH_CLI_FLAGS=-i timeout -k 10 400 python3 tests/bench/bench_cli_classes.py 50000 > $R/cli_classes_indel.json 2>> $R/misc.err
BENCH_CLI_DEV_OPTS="--shards 4" timeout -k 10 400 python3 tests/bench/bench_cli_classes.py 50000 > $R/cli_classes_shards4.json 2>> $R/misc.err
say strings;      timeout -k 10 300 python3 tests/bench/bench_strings.py 1000000 64 > $R/strings_bench.json 2>> $R/misc.err
bash tools/prof_kernels.sh r03str python3 tests/bench/bench_strings.py 1000000 8 > /dev/null 2>&1; kstats r03str strings_kernel_stats.csv
say orfs;         timeout -k 10 300 python3 tests/bench/bench_orfs.py 200000 5 > $R/orfs_bench.json 2>> $R/misc.err; timeout -k 10 400 python3 tests/bench/bench_orfs.py 1000000 3 > $R/orfs_bench_1M.json 2>> $R/misc.err
bash tools/prof_kernels.sh r03orf python3 tests/bench/bench_orfs.py 200000 3 > /dev/null 2>&1; kstats r03orf orfs_kernel_stats.csv
say ingest;       timeout -k 10 300 python3 tests/bench/bench_ingest.py > $R/ingest_bench.json 2>> $R/misc.err
say train;        timeout -k 10 300 python3 tests/bench/bench_train.py 1600 > $R/train_1600_bench.json 2>> $R/misc.err; timeout -k 10 400 python3 tests/bench/bench_train.py 64000 > $R/train_64000_bench.json 2>> $R/misc.err
bash tools/prof_kernels.sh r03trn python3 tests/bench/bench_train.py 64000 > /dev/null 2>&1; kstats r03trn train_kernel_stats.csv
say cli;          timeout -k 10 600 python3 tests/bench/bench_cli.py 200000 > $R/cli_bench.json 2>> $R/misc.err
say cli-shards;   BENCH_TMP=/dev/shm timeout -k 10 500 python3 tests/bench/bench_cli_shards.py 2000000 1 2 4 > $R/cli_shards_2M.json 2>> $R/misc.err
BENCH_TMP=/dev/shm timeout -k 10 900 python3 tests/bench/bench_cli_shards.py 10000000 1 2 4 > $R/cli_shards_10M.json 2>> $R/misc.err
say done
