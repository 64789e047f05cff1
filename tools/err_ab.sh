for v in "" B16 B8 B48; do
  if [ -z "$v" ]; then L=glimmer-mg_amd/lib/libgmg.so; else L=glimmer-mg_amd/lib/variants/libgmg_$v.so; fi
  echo "== ${v:-B32}"; GMG_LIB_PATH=$L BENCH_ERR=indel timeout -k 10 300 python tests/bench/bench_mg.py 1000000 3 ragged 2>&1 | cut -c100-150
done
