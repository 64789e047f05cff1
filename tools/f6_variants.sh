#!/bin/bash
# A/B runs of the six-frame kernel's build-time switches (glimmer-mg_amd/build.py build_variant): for every variant library
# the frame6 parity tests, then bench.py with and without the in-kernel patching.  Results: gpurun_out/f6_variants.txt
out=gpurun_out/f6_variants.txt
mkdir -p gpurun_out
: > $out
for v in "$@"; do
    lib=$PWD/glimmer-mg_amd/lib/variants/libgmg_$v.so
    GMG_LIB_PATH=$lib timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "frame6" > gpurun_out/f6_var_$v.pytest.log 2>&1
    echo "variant $v pytest rc=$? $(tail -1 gpurun_out/f6_var_$v.pytest.log)" >> $out
    for p in 1 0 1 0; do
        GMG_LIB_PATH=$lib GMG_F6_PATCH=$p timeout -k 10 200 python bench.py --steps 30 --warmup 5 --cpu-reads 2000 > gpurun_out/f6_var_$v.p$p.json 2> gpurun_out/f6_var_$v.p$p.err
        python -c "import json; j=json.load(open('gpurun_out/f6_var_$v.p$p.json')); print('variant $v patch $p', j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['frac'], j['check'])" >> $out 2>&1
    done
done
cat $out
