#!/bin/bash
# gpurun_out/r04e/* (tools/profile_r04.sh a, b, c) -> profiles/r04_*
R=gpurun_out/r04e
for f in bench.json bench_with_traffic.json pmc_summary_k_frame6t.txt pmc_summary_k_frame6p.txt frame6_kernel_stats.csv \
         mg_pmc_summary_k_mg_tile_starts.txt mg_pmc_summary_k_mg_find_orfs_ev.txt mg_kernel_stats.csv mg_timeline.txt \
         mgerr_pmc_summary_k_mg_err_level.txt mgerr_pmc_summary_k_mg_walk_prefix.txt mgerr_pmc_summary_k_mg_run_tables.txt mgerr_indel_kernel_stats.csv mgerr_timeline_indel.txt \
         mgerr_tile_pmc_summary_k_mg_err_tile.txt orfs_pmc_summary_k_orf_walk_sums.txt orfs_pmc_summary_k_orf_events.txt orfs_kernel_stats.csv \
         mgerr_level_vs_tile.jsonl mg_own_table.jsonl classes_bench.jsonl classes_bench_indel.jsonl multi_pmc_relabel.txt multi_pmc_trained.txt \
         strings_bench.jsonl orfs_bench.json ingest_bench.json cli_bench.json pytest_gpu.txt; do
  [ -f $R/$f ] && cp $R/$f profiles/r04_$f
done
cp $R/traffic.json profiles/traffic.json
