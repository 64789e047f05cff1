# A/B of library builds on the error branch, one box: tools/et_ab.sh <mode indel|sub> <variant names under lib/variants...> ("-" = the product build)
M=$1; shift
for v in "$@"; do
  if [ "$v" = "-" ]; then L=glimmer-mg_amd/lib/libgmg.so; else L=glimmer-mg_amd/lib/variants/libgmg_$v.so; fi
  echo "== $M $v"; GMG_LIB_PATH=$L BENCH_OWN_TABLE=1 BENCH_ERR=$M timeout -k 10 300 python tests/bench/bench_mg.py 1000000 5 ragged 2>&1 | grep -o '"ms_all": [^]]*]\|Error.*\|error.*' | head -3
done
