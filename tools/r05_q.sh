#!/bin/bash
R=$GRAFT_REPO_ROOT/gpurun_out/r05q; mkdir -p $R; rm -f $R/*
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GMG_EXPECT_REF=1
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x > $R/pytest1.txt 2>&1; tail -3 $R/pytest1.txt
GMG_ORFS_WALK8=4 GMG_ORFS_Q_POISON=1 timeout -k 10 900 python3 -m pytest tests/test_gpu_dropin_cli.py -m gpu -q -x > $R/pytest2.txt 2>&1; tail -2 $R/pytest2.txt
for rep in 1 2 3; do for w in 1 4; do GMG_ORFS_WALK8=$w BENCH_EXTRAS_LEGS=score_orfs timeout -k 10 300 python3 tests/bench/bench_extras.py 1000000 10 2>> $R/err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('walk8=$w', d['legs']['score_orfs']['ms'], d['legs']['score_orfs'].get('check'))" ; done; done 2>&1 | tee $R/ab.txt
GMG_ORFS_WALK8=4 BENCH_EXTRAS_LEGS=score_orfs bash tools/prof_kernels.sh r05orf6 python3 tests/bench/bench_extras.py 1000000 5 > $R/orfs_trace.log 2>&1
f=$(find gpurun_out/prof_r05orf6 -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-110
echo done
