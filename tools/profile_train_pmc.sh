# SQ counters of the training kernels (tests/bench/bench_train.py, 64,000 strings): what the sorted count kernel waits on.
# Two --pmc passes of their own (never combined with a trace); summary: python3 tools/summarize_pmc.py gpurun_out/prof_train_pmc k_train_count_sorted
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_train_pmc
mkdir -p $OUT
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc1 -- python3 tests/bench/bench_train.py 64000 1000 2 > $OUT/pmc1.log 2>&1 &&
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc2 -- python3 tests/bench/bench_train.py 64000 1000 2 > $OUT/pmc2.log 2>&1
for k in k_train_count_sorted "k_train_level<true" "k_train_level<false, true"; do echo "== $k"; python3 tools/summarize_pmc.py $OUT "$k"; done > $OUT/summary.txt 2>&1
tail -60 $OUT/summary.txt
# HBM-side bytes of the same kernels (FETCH_SIZE / WRITE_SIZE in passes of their own, units of 1 KiB on gfx950 as in tools/profile_frame6.sh)
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 tests/bench/bench_train.py 64000 1000 2 > $OUT/pmc3.log 2>&1 &&
timeout 400 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc4 -- python3 tests/bench/bench_train.py 64000 1000 2 > $OUT/pmc4.log 2>&1
for k in k_train_count_sorted "k_train_level<true" "k_train_level<false, true" "radix_sort_onesweep"; do echo "== $k"; python3 tools/summarize_pmc.py $OUT "$k"; done > $OUT/summary.txt 2>&1
tail -80 $OUT/summary.txt
