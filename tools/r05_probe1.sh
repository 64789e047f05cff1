#!/bin/bash
# probe: per-read-null path A/B across library builds (same box), PMC of the ORF passes alone
R=$GRAFT_REPO_ROOT/gpurun_out/r05p1; mkdir -p $R
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
V=glimmer-mg_amd/lib/variants
for rep in 1 2; do for l in $V/libgmg_r04.so $V/libgmg_r05a.so glimmer-mg_amd/lib/libgmg.so; do
  echo "$l" >> $R/prn_ab.jsonl
  GMG_LIB_PATH=$PWD/$l BENCH_PER_GROUP_CALLS=0 BENCH_SAME_MODEL=1 timeout -k 10 200 python3 tests/bench/bench_classes.py 1000000 64 100 7 >> $R/prn_ab.jsonl 2>> $R/err.txt
done; done
echo "[$(date +%T)] ab done"
GMG_MG_ONE_STREAM=1 BENCH_OWN_TABLE=1 bash tools/pmc_kernels.sh r05fo "k_mg_find_orfs_ev,k_mg_find_orfs<false" python3 tests/bench/bench_mg.py 1000000 3 > $R/fo_pmc.log 2>&1
cp gpurun_out/prof_r05fo/summary_* $R/
echo "[$(date +%T)] done"
