for round in 1 2; do
for v in w3 -; do
  if [ "$v" = "-" ]; then L=glimmer-mg_amd/lib/libgmg.so; else L=glimmer-mg_amd/lib/variants/libgmg_$v.so; fi
  for shape in uniform ragged; do
    A=""; [ $shape = ragged ] && A=ragged
    echo "== $v $shape nulls"; GMG_LIB_PATH=$L BENCH_NULLS=32 timeout -k 10 300 python tests/bench/bench_mg.py 1000000 7 $A 2>&1 | grep -o '"ms_all": [^]]*]\|Error.*\|error.*' | head -3
  done
done
done
