rm -f gpurun_out/mg_nulls.jsonl
for f in 1 0; do for m in "" ragged; do
  GMG_MG_FUSED=$f BENCH_NULLS=100 timeout -k 10 200 python tests/bench/bench_mg.py 1000000 5 $m 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); d['mg_fused']=$f; d['null_models']=100; print(json.dumps(d))" >> gpurun_out/mg_nulls.jsonl
done; done
BENCH_OWN_TABLE=1 timeout -k 10 200 python tests/bench/bench_mg.py 1000000 5 2>/dev/null >> gpurun_out/mg_nulls.jsonl
BENCH_OWN_TABLE=1 timeout -k 10 200 python tests/bench/bench_mg.py 1000000 5 ragged 2>/dev/null >> gpurun_out/mg_nulls.jsonl
python3 - <<'PY'
import json
for l in open('gpurun_out/mg_nulls.jsonl'):
    d=json.loads(l); print(d.get('null_models',1), d.get('mg_fused','-'), d['ragged'], round(d['ms'],2))
PY
