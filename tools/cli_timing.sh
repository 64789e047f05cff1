python3 - <<'PY'
import numpy as np
rng=np.random.default_rng(5)
b=np.frombuffer(b"acgt",np.uint8)[rng.integers(0,4,(200000,500))]
with open("/tmp/reads200k.fa","wb") as f:
    for i in range(200000):
        f.write(b">read%07d\n"%i); f.write(b[i].tobytes()); f.write(b"\n")
PY
for i in 1 2; do time env GMG_CLI_TIMING=1 integration/_build/glimmer-mg_gpu -m tests/golden/data/NC_000915.icm /tmp/reads200k.fa /tmp/out 2>&1 | tail -2; done
