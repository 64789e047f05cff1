# kernel trace of the training counts (tests/bench/bench_train.py): 64,000 strings x ~1,000 bp, then one genome's worth
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_train
mkdir -p $OUT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/big -- python3 tests/bench/bench_train.py 64000 1000 3 > $OUT/big.json 2> $OUT/big.err &&
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/one -- python3 tests/bench/bench_train.py 1600 1000 5 > $OUT/one.json 2> $OUT/one.err
cat $OUT/big.json $OUT/one.json
find $OUT -name "*kernel_stats.csv" | head
