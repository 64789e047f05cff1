for r in 1 2; do for nw in 1 2 4; do for shape in uniform ragged; do A=""; [ $shape = ragged ] && A=ragged
echo "== nw=$nw $shape"; GMG_MG_TILE=$nw BENCH_OWN_TABLE=1 timeout -k 10 300 python tests/bench/bench_mg.py 1000000 7 $A 2>&1 | grep -o '"ms_all": [^]]*]\|Error.*' | head -2
done; done; done
