#!/bin/bash
# Round-5 evidence run on the GPU box (gpurun calls of <= 20 minutes: parts a, b, c); everything under gpurun_out/r05e/;
# tools/collect_r05.sh then copies the summaries into profiles/r05_*.
set -u
PART=${1:-a}
R=$GRAFT_REPO_ROOT/gpurun_out/r05e
mkdir -p $R
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GMG_EXPECT_REF=1
say() { echo "[$(date +%T)] $*" | tee -a $R/progress_$PART.log; }
kstats() { f=$(find gpurun_out/prof_$1 -name "*kernel_stats.csv" | xargs ls -t | head -1); cp "$f" $R/$2; }
if [ $PART = a ]; then
say bench;        timeout -k 10 500 python3 bench.py > $R/bench.json 2> $R/bench.err
say frame6-prof;  timeout -k 10 900 bash tools/profile_frame6.sh r05f6 > $R/profile_frame6.log 2>&1
cp gpurun_out/prof_r05f6/summary.txt $R/pmc_summary_k_frame6t.txt; cp gpurun_out/prof_r05f6/summary_k_frame6p.txt $R/pmc_summary_k_frame6p.txt
cp $(find gpurun_out/prof_r05f6/trace -name "*kernel_stats.csv" | head -1) $R/frame6_kernel_stats.csv
python3 tools/update_traffic.py gpurun_out/prof_r05f6 > $R/traffic_update.log 2>&1; cp profiles/traffic.json $R/traffic.json
say bench-again;  timeout -k 10 400 python3 bench.py --no-extras > $R/bench_with_traffic.json 2> $R/bench2.err
say batches;      timeout -k 10 600 python3 bench.py --reads 12500000 --batches 13 --steps 3 --warmup 1 --no-extras --no-cli > $R/bench_configs2_one_gpu.json 2> $R/bench3.err
say done-a
fi
if [ $PART = b ]; then
say err-trace;    BENCH_OWN_TABLE=1 BENCH_ERR=indel bash tools/prof_kernels.sh r05err python3 tests/bench/bench_mg.py 1000000 3 ragged > $R/err_trace.log 2>&1
kstats r05err mgerr_indel_kernel_stats.csv; f=$(find gpurun_out/prof_r05err -name "*kernel_trace.csv" | head -1); python3 tools/mg_timeline.py $f > $R/mgerr_timeline_indel.txt
say err-pmc;      GMG_MG_ONE_STREAM=1 BENCH_OWN_TABLE=1 BENCH_ERR=indel bash tools/pmc_kernels.sh r05errp "k_mg_err_wcount<false,k_mg_err_wcount<true" python3 tests/bench/bench_mg.py 1000000 2 ragged > $R/err_pmc.log 2>&1
cp "gpurun_out/prof_r05errp/summary_k_mg_err_wcount<false.txt" $R/mgerr_pmc_summary_k_mg_err_wcount_count.txt; cp "gpurun_out/prof_r05errp/summary_k_mg_err_wcount<true.txt" $R/mgerr_pmc_summary_k_mg_err_wcount_write.txt
say sub-trace;    BENCH_OWN_TABLE=1 BENCH_ERR=sub bash tools/prof_kernels.sh r05sub python3 tests/bench/bench_mg.py 1000000 3 ragged > $R/sub_trace.log 2>&1
kstats r05sub mgerr_sub_kernel_stats.csv
say err-modes;    for e in indel sub; do for w in 1 2 0; do GMG_MG_ERR_WAVE=$w BENCH_OWN_TABLE=1 BENCH_ERR=$e timeout -k 10 300 python3 tests/bench/bench_mg.py 1000000 5 ragged >> $R/mgerr_modes.jsonl 2>> $R/misc.err; done; done
say mg-trace;     BENCH_OWN_TABLE=1 bash tools/prof_kernels.sh r05mg python3 tests/bench/bench_mg.py 1000000 3 > $R/mg_trace.log 2>&1
kstats r05mg mg_kernel_stats.csv; f=$(find gpurun_out/prof_r05mg -name "*kernel_trace.csv" | head -1); python3 tools/mg_timeline.py $f > $R/mg_timeline.txt
say mg-ab;        timeout -k 10 300 python3 tools/mg_ab.py glimmer-mg_amd/lib/variants/libgmg_r04.so glimmer-mg_amd/lib/libgmg.so > $R/mg_ab_r04_vs_r05.txt 2>&1; timeout -k 10 300 python3 tools/mg_ab.py ragged glimmer-mg_amd/lib/variants/libgmg_r04.so glimmer-mg_amd/lib/libgmg.so >> $R/mg_ab_r04_vs_r05.txt 2>&1
say orfs-trace;   BENCH_EXTRAS_LEGS=score_orfs bash tools/prof_kernels.sh r05orf python3 tests/bench/bench_extras.py 1000000 5 > $R/orfs_trace.log 2>&1
kstats r05orf orfs_kernel_stats.csv
say circular;     timeout -k 10 300 python3 tools/time_circular.py > $R/circular_genome.txt 2>&1
say done-b
fi
if [ $PART = c ]; then
say classes;      for m in 1 distinct; do BENCH_PER_GROUP_CALLS=0 BENCH_SAME_MODEL=$m timeout -k 10 300 python3 tests/bench/bench_classes.py 1000000 64 100 7 >> $R/classes_bench.jsonl 2>> $R/misc.err; done
for m in 1 distinct; do BENCH_ERR=indel BENCH_PER_GROUP_CALLS=0 BENCH_SAME_MODEL=$m timeout -k 10 300 python3 tests/bench/bench_classes.py 1000000 64 100 5 >> $R/classes_bench_indel.jsonl 2>> $R/misc.err; done
say cli;          timeout -k 10 600 python3 tests/bench/bench_cli.py 200000 > $R/cli_bench.json 2>> $R/misc.err
say stress;       timeout -k 10 400 python3 tests/bench/stress_mg.py 20000 5 > $R/stress_mg.txt 2>&1; timeout -k 10 300 python3 tests/bench/stress_mg_fused.py > $R/stress_mg_fused.txt 2>&1
say batches;      timeout -k 10 400 python3 bench.py --reads 12500000 --batches 13 --steps 3 --warmup 1 --no-extras --no-cli > $R/bench_configs2_one_gpu.json 2> $R/bench3.err
say done-c
fi
if [ $PART = d ]; then
say tests;        timeout -k 10 1150 python3 -m pytest tests -m gpu -q -x > $R/pytest_gpu.txt 2>&1; tail -3 $R/pytest_gpu.txt
say done-d
fi
if [ $PART = e ]; then
say lds-probe;    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_table_probe tools/probes/lds_table_probe.hip > $R/lds_probe_build.log 2>&1 && timeout -k 10 200 /tmp/lds_table_probe > $R/lds_table_probe.txt 2>&1
say ew-stamps;    for m in indel sub; do GMG_LIB_PATH=$PWD/glimmer-mg_amd/lib/variants/libgmg_ewstamps.so timeout -k 10 200 python3 tools/ew_stamps.py $m >> $R/ew_stamps.txt 2>&1; done
say orfbits;      timeout -k 10 300 python3 tools/mg_ab.py glimmer-mg_amd/lib/libgmg.so glimmer-mg_amd/lib/libgmg.so:mg_orfs_bits=1 > $R/mg_ab_orfs_bits.txt 2>&1
say done-e
fi
