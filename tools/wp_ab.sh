#!/bin/bash
# A/B of the error branch's table kernels, each alone on the device (mg_one_stream): tools/wp_ab.sh <variant> ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  if [ "$v" = default ]; then unset GMG_LIB_PATH; else export GMG_LIB_PATH=$GRAFT_REPO_ROOT/glimmer-mg_amd/lib/variants/libgmg_$v.so; fi
  echo "== $v"
  BENCH_OWN_TABLE=1 BENCH_ERR=indel timeout -k 10 300 python3 tests/bench/bench_mg.py 1000000 5 ragged | cut -c1-160
  GMG_MG_ONE_STREAM=1 BENCH_OWN_TABLE=1 BENCH_ERR=indel bash tools/prof_kernels.sh wp_$v python3 tests/bench/bench_mg.py 1000000 3 ragged > /dev/null 2>&1
  f=$(find gpurun_out/prof_wp_$v -name "*kernel_stats.csv" | head -1)
  grep -E "walk_prefix|run_tables|k_mg_quality|find_orfs|err_level|frame6" $f | cut -d, -f1-5 | cut -c1-150
done
