#!/usr/bin/env python3
"""bench.py -- Mbases/s scored (6-frame IMM) on synthetic reads, 1..8 MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (gmg_frame_score6, include/gmg.h) over one rank's shard of
synthetic reads already resident in HBM: 1M x 500 bp per GPU (BASELINE.json configs[1]); with N > 1
every rank scores its own shard (weak scaling, no collective in the data path -- reads shard
embarrassingly).  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline      dominant kernel (k_frame6) vs the HBM roof: achieved = 48.25 algorithmic bytes per base
                (0.25 B packed input + 6 x 8 B fp64 Frame_Scores) x bases per launch / mean
                launch duration measured with HIP events on the launch stream.
                roofline.measured_fill_GBps = what a plain fill of a same-sized buffer reaches on this GPU (a
                write-only stream's practical ceiling; reported beside, never instead of, the 8 TB/s peak).
                The same leg is the run's checker: `check` compares the XOR of all 6 x L x sample output doubles on the
                device with the CPU's.
  cpu_baseline  the same six-frame loop timed on this box's host cores on a bounded sample of the same
                reads: the real reference's ICM_t (oracle/_ref/ref_bench, "reference") when that build
                is present, else the plain-C oracle ("port").  Rank 0, N = 1 only.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_BASE = 0.25 + 6 * 8.0
HBM_PEAK_GBPS = 8000.0                    # MI355X spec peak, /opt/skills/guides/MI355X_MICROARCH.md
SEED = 20260101
MODEL = os.path.join(ROOT, "tests", "golden", "data", "NC_000915.icm")


# ----------------------------------------------------------------------------------------------
# distributed plumbing (backend-agnostic so that tests can drive it with gloo on CPU)
# ----------------------------------------------------------------------------------------------

def shard_plan(total_reads, world):
    """contiguous read ranges per rank (strong-scaling helper, also used by the tests)"""
    base, extra = divmod(total_reads, world)
    out, lo = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((lo, lo + n))
        lo += n
    return out


def timed_region(step, steps, warmup, sync, dist=None):
    """W untimed steps, then EXACTLY K steps bracketed by barrier + sync on both sides.
    Returns (seconds as MAX over ranks, per-rank seconds)."""
    for _ in range(warmup):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    mine = time.perf_counter() - t0
    worst = mine
    if dist is not None:
        import torch
        t = torch.tensor([mine], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        worst = float(t.item())
    return worst, mine


def aggregate(units_per_rank_per_step, world, steps, seconds):
    """whole-job throughput: units all ranks processed in the timed region / max-over-ranks time"""
    return units_per_rank_per_step * world * steps / seconds


# ----------------------------------------------------------------------------------------------
# CPU baseline (test infrastructure: oracle/_ref or the oracle port; never the product)
# ----------------------------------------------------------------------------------------------

def cpu_baseline(n_reads, L, seed, gc, packed):
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_bench")
    sample = "first %d of the rank-0 reads (%d x %d bp, same seed)" % (n_reads, n_reads, L)
    if os.access(ref, os.X_OK):
        try:
            res = subprocess.run([ref, MODEL, str(n_reads), str(L), str(seed), repr(float(gc))],
                                 check=True, stdout=subprocess.PIPE, timeout=600)
            j = json.loads(res.stdout)
            return {"value": round(j["mbases_per_s"], 4), "unit": "Mbases/s", "cores": 1, "kind": "reference",
                    "sample": sample, "seconds": j["seconds"], "xor": j["xor"]}
        except Exception as e:          # fall through to the port, but say why
            sys.stderr.write("bench: oracle/_ref/ref_bench failed (%s); timing the oracle port\n" % e)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_py
    import _gmg_pkg
    gmg = _gmg_pkg.load()
    orc = oracle_py.load()
    gene, indep = orc.read(MODEL), orc.indep(gc)
    ascii_all = gmg.synth.unpack_ascii(packed, 0, n_reads * L)
    t0 = time.perf_counter()
    out = orc.score_reads_6frame(gene, indep, ascii_all, n_reads, L)
    dt = time.perf_counter() - t0
    x = np.bitwise_xor.reduce(out.view(np.uint64).ravel())
    return {"value": round(n_reads * L / dt / 1e6, 4), "unit": "Mbases/s", "cores": 1, "kind": "port",
            "sample": sample, "seconds": dt, "xor": "%016x" % int(x)}


# ----------------------------------------------------------------------------------------------

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU")
    ap.add_argument("--length", type=int, default=500)
    ap.add_argument("--cpu-reads", type=int, default=20_000, help="reads in the CPU-baseline sample (0 = skip)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import _gmg_pkg
    gmg = _gmg_pkg.load()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world
    dist = None
    torch.cuda.set_device(local_rank)
    if world > 1 or os.environ.get("GMG_BENCH_FORCE_DIST"):     # (the switch lets a 1-GPU box exercise the RCCL plumbing)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    gmg.build.build_lib()
    gmg.init(local_rank)                      # raises if there is no gfx950 device: no fallback

    n, L = args.reads, args.length
    seed = SEED + rank                        # every rank has its own shard of the job
    packed, off = gmg.synth.packed_reads(n, L, seed)
    total = n * L
    # null-model GC the way Set_GC_Fraction computes it (glimmer_base.cc:2564-2595): count of g/c over all bases
    codes = np.unpackbits(packed[:(total + 15) // 16].view(np.uint8), bitorder="little").reshape(-1, 2)
    gc = float(np.count_nonzero(codes[:total, 0] != codes[:total, 1])) / total     # c=01, g=10 (LSB first)
    del codes
    gene = gmg.Icm.open(MODEL)
    indep = gmg.Icm.indep(gc)
    reads = gmg.Reads(packed, off)
    out = torch.empty(6 * total, dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream()
    sptr = stream.cuda_stream

    ev = []

    def step():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        gmg.frame_score6(gene, indep, reads, d_out=out.data_ptr(), stream=sptr)
        b.record(stream)
        ev.append((a, b))

    seconds, _ = timed_region(step, args.steps, args.warmup, torch.cuda.synchronize, dist)
    timed = ev[args.warmup:]
    kern_ms = sum(a.elapsed_time(b) for a, b in timed) / max(len(timed), 1)
    if dist is not None:
        t = torch.tensor([kern_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        kern_ms = float(t.item())

    # what a plain write stream into the same buffer reaches on this GPU (outside the timed region; rank 0 reports it):
    # the practical ceiling of a kernel that must write 48 of its 48.25 B/base
    fill_gbps = None
    if rank == 0:
        probe = torch.empty_like(out)
        best = None
        for _ in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            probe.fill_(1.0)
            b.record(stream)
            b.synchronize()
            ms = a.elapsed_time(b)
            best = ms if best is None or ms < best else best
        fill_gbps = probe.numel() * 8 / (best * 1e-3) / 1e9
        del probe

    check = None                                      # filled by the cpu_baseline leg below

    if rank == 0:
        value = aggregate(total, world, args.steps, seconds) / 1e6
        achieved = ALGO_BYTES_PER_BASE * total / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("%dx%d" % (n, L))
            except Exception:
                traffic = None
        line = {
            "metric": "Mbases/s scored (6-frame IMM)", "value": round(value, 2), "unit": "Mbases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(seconds / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d x %d bp synthetic reads per GPU, one 3-periodic ICM (NC_000915.icm), "
                                   "6-frame per-position scoring, fp64 Frame_Scores" % (n, L),
                       "reads_per_gpu": n, "read_len": L, "parallelism": "reads sharded, %d rank(s)" % world},
            "roofline": {"bound": "hbm", "kernel": "k_frame6", "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": traffic, "kernel_ms": round(kern_ms, 4),
                         "algorithmic_bytes_per_base": ALGO_BYTES_PER_BASE,
                         "measured_fill_GBps": round(fill_gbps, 1) if fill_gbps else None},
            "check": check,
        }
        if world == 1 and args.cpu_reads > 0:
            ns = min(args.cpu_reads, n)
            line["cpu_baseline"] = cpu_baseline(ns, L, seed, gc, packed)
            # the baseline leg doubles as the checker: XOR of all 6 x L x ns output doubles, device vs CPU
            host = out.view(6, total)[:, :ns * L].cpu().numpy()
            x = int(np.bitwise_xor.reduce(host.view(np.uint64).ravel()))
            same = "%016x" % x == line["cpu_baseline"]["xor"]
            line["check"] = ("bit-exact vs the CPU %s on %d reads (XOR of all output doubles)" % (line["cpu_baseline"]["kind"], ns)
                             if same else "MISMATCH vs the CPU %s" % line["cpu_baseline"]["kind"])
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
