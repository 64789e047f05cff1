#!/bin/bash
# One command for the multi-GPU numbers of a node: weak and strong scaling of bench.py over N = 1, 2, 4, 8 GPUs and the
# CLI driver with one forked shard per GPU.  Writes gpurun_out/scale/{weak,strong}_N.json, cli_N.json and table.md.
#     tools/scale_table.sh [max_gpus, default: all visible]
# (The driver's own SCALE run launches bench.py the same way; this script only puts the three curves side by side.)
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/scale; mkdir -p $OUT
MAXG=${1:-$(python3 -c 'import torch; print(torch.cuda.device_count())')}
export MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
PORT=29500
for N in 1 2 4 8; do
  [ "$N" -gt "$MAXG" ] && break
  for MODE in weak strong configs2; do
    EXTRA=""; [ "$MODE" = strong ] && EXTRA="--scaling strong --reads $((N * 1000000))"
    # BASELINE configs[2] at its stated size: 12.5M reads per GPU as 13 batches into one reused table (5 steps: 31 Gbases per rank)
    [ "$MODE" = configs2 ] && EXTRA="--reads 12500000 --batches 13 --steps 5 --warmup 1 --no-extras"
    if [ "$N" = 1 ]; then
      timeout -k 10 600 python3 bench.py --gpus 1 --no-cli $EXTRA > $OUT/${MODE}_$N.json 2> $OUT/${MODE}_$N.err
    else
      PORT=$((PORT + 1))
      timeout -k 10 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
        bench.py --gpus $N --no-cli $EXTRA > $OUT/${MODE}_$N.json 2> $OUT/${MODE}_$N.err
    fi
  done
  BENCH_GPUS=$N timeout -k 10 900 python3 tests/bench/bench_cli_shards.py 2000000 $N > $OUT/cli_$N.json 2> $OUT/cli_$N.err
done
python3 - <<'PY' > $OUT/table.md
import glob, json, os
out = "gpurun_out/scale"
def last_json(p):
    try:
        return json.loads([l for l in open(p).read().splitlines() if l.startswith("{")][-1])
    except Exception:
        return None
print("| GPUs | weak Mbases/s | x | strong Mbases/s (N x 1M reads) | configs[2]: 12.5M reads per GPU in 13 batches, Mbases/s | CLI 2 M reads, one shard per GPU: s | Mbases/s |")
print("|---|---|---|---|---|---|---|")
base = None
for n in (1, 2, 4, 8):
    w, s, c, c2 = (last_json(os.path.join(out, "%s_%d.json" % (k, n))) for k in ("weak", "strong", "cli", "configs2"))
    if not w:
        continue
    base = base or w["value"]
    run = c["runs"][0] if c else {}
    print("| %d | %.0f | %.2f | %s | %s | %s | %s |" % (n, w["value"], w["value"] / base, "%.0f" % s["value"] if s and s.get("value") else "-",
                                                   "%.0f" % c2["value"] if c2 and c2.get("value") else "-",
                                                   run.get("seconds", "-"), run.get("mbases_per_s", "-")))
PY
cat $OUT/table.md
