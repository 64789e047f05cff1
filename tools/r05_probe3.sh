#!/bin/bash
R=$GRAFT_REPO_ROOT/gpurun_out/r05p3; mkdir -p $R
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GMG_EXPECT_REF=1
timeout -k 10 600 python3 -m pytest tests/test_gpu_orfbits.py -m gpu -q -x > $R/pytest1.txt 2>&1; tail -5 $R/pytest1.txt
echo "[$(date +%T)] tests 1 done"
V=glimmer-mg_amd/lib/variants
timeout -k 10 300 python3 tools/mg_ab.py $V/libgmg_r04.so glimmer-mg_amd/lib/libgmg.so glimmer-mg_amd/lib/libgmg.so:mg_orfs_bits=0 > $R/mg_ab.txt 2>&1
cat $R/mg_ab.txt
GMG_MG_ONE_STREAM=1 BENCH_OWN_TABLE=1 bash tools/pmc_kernels.sh r05fb "k_mg_find_orfs_bits<false,k_mg_find_orfs_bits<true" python3 tests/bench/bench_mg.py 1000000 3 > $R/fb_pmc.log 2>&1
cp gpurun_out/prof_r05fb/summary_* $R/
cat $R/summary_*
echo "[$(date +%T)] done"
