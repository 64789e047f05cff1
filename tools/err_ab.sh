# A/B of library builds on the error branch (-i), one box: tools/err_ab.sh <variant names under lib/variants...> ("" = the product build)
for v in "$@"; do
  if [ "$v" = "-" ]; then L=glimmer-mg_amd/lib/libgmg.so; else L=glimmer-mg_amd/lib/variants/libgmg_$v.so; fi
  echo "== $v"; GMG_LIB_PATH=$L BENCH_ERR=indel timeout -k 10 300 python tests/bench/bench_mg.py 1000000 3 ragged 2>&1 | cut -c100-150
done
