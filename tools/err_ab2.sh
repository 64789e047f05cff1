# A/B of library builds on the error branch, one box, runs interleaved: tools/err_ab2.sh <indel|sub> <variant|-> ... ("-" = the product build)
mode=$1; shift
for rep in $(seq 1 ${REPS:-2}); do
for v in "$@"; do
  if [ "$v" = "-" ]; then L=glimmer-mg_amd/lib/libgmg.so; else L=glimmer-mg_amd/lib/variants/libgmg_$v.so; fi
  echo "== $mode $v"; GMG_LIB_PATH=$L BENCH_ERR=$mode timeout -k 10 300 python tests/bench/bench_mg.py 1000000 5 ragged 2>&1 | tail -1 | cut -c1-400
done
done
