#!/usr/bin/env python3
"""bench.py -- Mbases/s scored (6-frame IMM) on synthetic reads, 1..8 MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (gmg_frame_score6, include/gmg.h) over one rank's shard of
synthetic reads already resident in HBM: 1M x 500 bp per GPU (BASELINE.json configs[1]); with N > 1
every rank scores its own shard (weak scaling, no collective in the data path -- reads shard
embarrassingly).  `--batches B`: the rank's --reads reads as B batches into ONE reused table (BASELINE configs[2] at its stated
size: --reads 12500000 --batches 13; a step = all batches, the check runs on the last one).  `--scaling strong` keeps the job fixed instead: --reads reads in total, cut over
the ranks by gmg_shard_plan (contiguous ranges of equal base count).  Rank 0 prints ONE JSON line.

`--data genome`: SURVEY.md 8(d)'s second input distribution -- the same job shape with every read cut uniformly from
tests/golden/data/NC_000915.fna (the reference's sample genome), either strand; same JSON contract, same head + tail check
("data" says which).  The default (what the driver runs) is the synthetic stream.

Extra objects on that line:
  roofline      dominant kernels (k_frame6t + k_frame6p, one call) vs the HBM roof: achieved = 48.25 algorithmic
                bytes per base (0.25 B packed input + 6 x 8 B fp64 Frame_Scores) x bases per call /
                MEDIAN call duration over the timed steps, measured with HIP events on the launch stream.
                roofline.measured_fill_GBps = what a plain fill of a same-sized buffer reaches on this GPU (a
                write-only stream's practical ceiling; reported beside, never instead of, the 8 TB/s peak).
  check         the device table against the CPU on the FIRST and the LAST --cpu-reads reads of rank 0's shard: XOR
                of all output doubles and a position-weighted sum (a permuted or shifted table fails it).  A mismatch
                nulls `value` and the process exits 1.  With N > 1 rank 0 checks 2,000 + 2,000 reads after the timed region.
  cpu_baseline  the same six-frame loop timed on this box's host cores on a bounded sample of the same
                reads: the real reference's ICM_t (oracle/_ref/ref_bench, "reference") when that build is present,
                else the plain-C oracle ("port").  `value` is ONE core (the reference is single-threaded);
                `all_cores` runs one process per host core on disjoint slices (process sharding).  Rank 0, N = 1 only.
  cli_end_to_end  FASTA file -> .predict, the reference CLI against integration/glimmer-mg_gpu (N = 1, when both are built).
  extras        (N = 1, after the timed region, in a process of its own: tests/bench/bench_extras.py) the other device paths of
                SURVEY section 8(f) on 1M reads, 3 warm-ups + 10 calls each, every leg with its own `roofline` object and a sampled
                check against the oracle: gmg_mg_score_reads (500 bp, ragged, -i, -s), gmg_mg_score_groups (64 ICMs x 100 null
                models), gmg_score_reads_strings (64 ICMs), gmg_score_orfs, gmg_fasta_ingest.  --no-extras skips it.
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
REF_BENCH = os.path.join(ROOT, "oracle", "_ref", "ref_bench")
GENOME = os.path.join(ROOT, "tests", "golden", "data", "NC_000915.fna")


# ----------------------------------------------------------------------------------------------
# distributed plumbing (backend-agnostic so that tests can drive it with gloo on CPU)
# ----------------------------------------------------------------------------------------------

def shard_plan(total_reads, world):
    """contiguous read ranges per rank for reads of ONE length (the general, base-balanced plan is gmg_shard_plan)"""
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


def aggregate(units_per_step_all_ranks, steps, seconds):
    """whole-job throughput: units all ranks processed in the timed region / max-over-ranks time"""
    return units_per_step_all_ranks * steps / seconds


# ----------------------------------------------------------------------------------------------
# CPU baseline (test infrastructure: oracle/_ref or the oracle port; never the product)
# ----------------------------------------------------------------------------------------------

def ref_bench(first_read, n_reads, L, seed, gc):
    res = subprocess.run([REF_BENCH, MODEL, str(n_reads), str(L), str(seed), repr(float(gc)), str(first_read)],
                         check=True, stdout=subprocess.PIPE, timeout=900)
    return json.loads(res.stdout)


def port_bench(first_read, n_reads, L, seed, gc, packed):
    """the plain-C oracle on reads [first_read, first_read + n_reads) of the packed stream -> the same fields as ref_bench"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_py
    import _gmg_pkg
    gmg = _gmg_pkg.load()
    orc = oracle_py.load()
    gene, indep = orc.read(MODEL), orc.indep(gc)
    ascii_all = gmg.synth.unpack_ascii(packed, first_read * L, n_reads * L)
    t0 = time.perf_counter()
    out = orc.score_reads_6frame(gene, indep, ascii_all, n_reads, L)          # [n, 6, L]
    dt = time.perf_counter() - t0
    bits = np.ascontiguousarray(out.transpose(1, 0, 2)).view(np.uint64).ravel()   # [6][n * L]: the sample's table layout
    with np.errstate(over="ignore"):
        mix = int((bits * (2 * np.arange(bits.size, dtype=np.uint64) + 1)).sum(dtype=np.uint64))
    return {"bases": n_reads * L, "seconds": dt, "mbases_per_s": n_reads * L / dt / 1e6,
            "xor": "%016x" % int(np.bitwise_xor.reduce(bits)), "mix": "%016x" % mix}


def ref_bench_fasta(local_first_read, n_reads, L, gc, packed):
    """the real reference on reads that are not the synthetic stream: the sample goes over as a FASTA file"""
    import tempfile
    import _gmg_pkg
    gmg = _gmg_pkg.load()
    seq = gmg.synth.unpack_ascii(packed, local_first_read * L, n_reads * L)
    with tempfile.NamedTemporaryFile("wb", suffix=".fa", delete=False) as f:
        for r in range(n_reads):
            f.write(b">r%d\n" % r + seq[r * L:(r + 1) * L] + b"\n")
        path = f.name
    try:
        res = subprocess.run([REF_BENCH, MODEL, str(n_reads), str(L), "@" + path, repr(float(gc))], check=True,
                             stdout=subprocess.PIPE, timeout=900)
    finally:
        os.unlink(path)
    return json.loads(res.stdout)


def cpu_sample(stream_first_read, local_first_read, n_reads, L, seed, gc, packed, from_stream=True):
    """reads [stream_first_read, +n_reads) of the stream of `seed` = reads [local_first_read, +n_reads) of `packed`
    -> (result dict, kind)"""
    if os.access(REF_BENCH, os.X_OK):
        try:
            if not from_stream:
                return ref_bench_fasta(local_first_read, n_reads, L, gc, packed), "reference"
            return ref_bench(stream_first_read, n_reads, L, seed, gc), "reference"
        except Exception as e:          # fall through to the port, but say why
            sys.stderr.write("bench: oracle/_ref/ref_bench failed (%s); timing the oracle port\n" % e)
    return port_bench(local_first_read, n_reads, L, seed, gc, packed), "port"


def host_cores():
    """the cores this process may really use: affinity mask, cut by the cgroup CPU quota (a 1-GPU box's share of its host);
    GMG_BENCH_CORES overrides"""
    if os.environ.get("GMG_BENCH_CORES"):
        return max(1, int(os.environ["GMG_BENCH_CORES"]))
    n = len(os.sched_getaffinity(0))
    try:                                                    # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except Exception:
        try:                                                # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except Exception:
            pass
    return min(n, 64)                                       # (more processes than that only measure the scheduler)


def cpu_all_cores(n_reads_each, L, seed, gc):
    """one ref_bench process per host core, each on its own slice of the stream; wall time of all of them"""
    cores = host_cores()
    t0 = time.perf_counter()
    procs = [subprocess.Popen([REF_BENCH, MODEL, str(n_reads_each), str(L), str(seed), repr(float(gc)), str(c * n_reads_each)],
                              stdout=subprocess.PIPE) for c in range(cores)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    wall = time.perf_counter() - t0
    if any(p.returncode != 0 for p in procs):
        return None
    secs = [json.loads(o)["seconds"] for o in outs]         # each process times its scoring loop alone (no start-up, no model load)
    return {"value": round(cores * n_reads_each * L / max(secs) / 1e6, 3), "unit": "Mbases/s", "cores": cores,
            "sample": "%d concurrent processes x %d reads x %d bp, disjoint slices of the same stream; all bases / the slowest "
                      "process's scoring time" % (cores, n_reads_each, L),
            "slowest_scoring_seconds": round(max(secs), 3), "wall_seconds_with_startup": round(wall, 3)}


def device_digest(out, total, first_read, n_reads, L):
    """XOR and position-weighted sum of the sample's slice of the device table, computed on the device"""
    import torch
    # (the table of the call that ran last: 6 rows of `total` doubles at the buffer's start -- with --batches the buffer is sized by the largest batch)
    v = out[:6 * total].view(6, total)[:, first_read * L:(first_read + n_reads) * L].contiguous().view(torch.int64).ravel()
    w = 2 * torch.arange(v.numel(), dtype=torch.int64, device=v.device) + 1
    mix = int((v * w).sum().item()) & (2 ** 64 - 1)                       # int64 arithmetic wraps: mod 2^64
    x = v.cpu().numpy().view("uint64")
    import numpy as np
    return "%016x" % int(np.bitwise_xor.reduce(x)), "%016x" % mix


def extras(n_reads=1_000_000, reps=10):
    try:
        res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench", "bench_extras.py"), str(n_reads), str(reps)],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=max(240, int(120 + 12 * reps * n_reads / 1e6)))
        j = json.loads(res.stdout.decode().strip().splitlines()[-1])
        return j
    except Exception as e:
        return {"error": str(e)[:300]}


def cli_end_to_end(n_reads=20000):
    exe = os.path.join(ROOT, "integration", "_build", "glimmer-mg_gpu")
    ref = os.path.join(ROOT, "oracle", "_ref", "glimmer-mg")
    if not (os.access(exe, os.X_OK) and os.access(ref, os.X_OK)):
        return None
    try:
        res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench", "bench_cli.py"), str(n_reads)],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=dict(os.environ, BENCH_CLI_SKIP_G3="1"))
        j = json.loads(res.stdout.decode().strip().splitlines()[-1])
        return {k: j[k] for k in ("reads", "predict_identical", "reference_cli_s", "reference_cli_mbases_per_s",
                                  "device_front_half_cli_s", "device_front_half_cli_mbases_per_s")}
    except Exception as e:
        return {"error": str(e)[:200]}


# ----------------------------------------------------------------------------------------------

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU (weak) / in total (strong)")
    ap.add_argument("--length", type=int, default=500)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--cpu-reads", type=int, default=20_000, help="reads in each CPU sample (0 = no CPU legs, no check)")
    ap.add_argument("--no-cli", action="store_true", help="skip the CLI end-to-end leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the (f)-row legs (tests/bench/bench_extras.py)")
    ap.add_argument("--batches", type=int, default=1,
                    help="score a rank's --reads reads as this many batches (<= 1M reads each is the tested shape) into ONE reused table: "
                         "BASELINE configs[2]'s 12.5M reads per GPU = --reads 12500000 --batches 13; a step = all batches")
    ap.add_argument("--data", choices=("synthetic", "genome"), default="synthetic",
                    help="genome: reads cut uniformly from tests/golden/data/NC_000915.fna, both strands (weak scaling only)")
    args = ap.parse_args()

    # stdout carries ONE line, the JSON: whatever libraries write to file descriptor 1 on the way (RCCL prints a version banner
    # there when a process group is made) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

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

    L = args.length
    n_batches = max(1, args.batches)
    if n_batches > 1 and (args.scaling != "weak" or args.data != "synthetic"):
        sys.exit("bench.py: --batches runs with weak scaling on the synthetic stream")
    if args.scaling == "weak":
        n, seed, first_base = args.reads, SEED + rank, 0        # every rank has its own shard of the job
        if n_batches > 1:                                   # the LAST batch is the one kept on the host for the check; all are resident in HBM
            per = -(-args.reads // n_batches)
            sizes = [min(per, args.reads - b * per) for b in range(n_batches) if args.reads - b * per > 0]
            n_batches = len(sizes)
            n = sizes[-1]
            first_base = sum(sizes[:-1]) * L
        if args.data == "genome":
            packed, off = gmg.synth.genome_reads(GENOME, n, L, seed)
        elif n_batches > 1:                                 # one stream per rank, cut into the batches (the same bases as one 12.5M-read batch would hold)
            packed, off = gmg.synth.packed_reads_range(first_base, n * L, L, seed)
        else:
            packed, off = gmg.synth.packed_reads(n, L, seed)
        job_reads = args.reads * world
    else:                                     # one job of --reads reads, cut by gmg_shard_plan (equal base counts)
        if args.data != "synthetic":
            sys.exit("bench.py: --data genome runs with weak scaling only")
        job_off = np.arange(args.reads + 1, dtype=np.uint64) * np.uint64(L)
        plan = gmg.shard.shard_plan(job_off, world)
        lo, hi = int(plan[rank]), int(plan[rank + 1])
        n, seed, first_base = hi - lo, SEED, lo * L
        packed, off = gmg.synth.packed_reads_range(first_base, n * L, L, seed)
        job_reads = args.reads
    total = n * L
    # null-model GC the way Set_GC_Fraction computes it (glimmer_base.cc:2564-2595): count of g/c over all bases of the JOB:
    # per-shard {gc, total} counts, summed over the ranks (two integers each; shard.allreduce_counts)
    codes = np.unpackbits(packed[:(total + 15) // 16].view(np.uint8), bitorder="little").reshape(-1, 2)
    gc_count = int(np.count_nonzero(codes[:total, 0] != codes[:total, 1]))        # c=01, g=10 (LSB first)
    del codes
    if args.scaling == "strong":
        gcs, totals = gmg.shard.allreduce_counts(dist if world > 1 else None, gc_count, total)
        gc = gmg.shard.gc_fraction(gcs, totals, as_reference=False)
    else:
        gc = gc_count / max(total, 1)
    gene = gmg.Icm.open(MODEL)
    indep = gmg.Icm.indep(gc)
    reads = gmg.Reads(packed, off)
    batches = [reads]
    table_bases = total
    if n_batches > 1:                                       # the earlier batches: packed reads resident in HBM, no host copy kept
        batches, b0 = [], 0
        for nb in sizes[:-1]:
            pk, of = gmg.synth.packed_reads_range(b0 * L, nb * L, L, seed)
            batches.append(gmg.Reads(pk, of))
            b0 += nb
            del pk, of
        batches.append(reads)
        table_bases = max(sizes) * L
    out = torch.empty(6 * max(table_bases, 1), dtype=torch.float64, device="cuda")      # ONE table, reused by every batch
    stream = torch.cuda.current_stream()
    sptr = stream.cuda_stream

    ev = []

    def step():
        for bi, batch in enumerate(batches):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            gmg.frame_score6(gene, indep, batch, d_out=out.data_ptr(), stream=sptr)
            b.record(stream)
            if bi == 0:                                     # (the roofline's call: a full-size batch)
                ev.append((a, b))

    seconds, _ = timed_region(step, args.steps, args.warmup, torch.cuda.synchronize, dist)
    per_call = sorted(a.elapsed_time(b) for a, b in ev[args.warmup:])
    kern_ms = per_call[len(per_call) // 2] if per_call else 0.0               # MEDIAN of the timed calls
    kern_min, kern_max = (per_call[0], per_call[-1]) if per_call else (0.0, 0.0)
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

    ok = True
    if rank == 0:
        value = aggregate(job_reads * L, args.steps, seconds) / 1e6
        call_bases = (sizes[0] if n_batches > 1 else n) * L        # bases of the call the events bracket
        achieved = ALGO_BYTES_PER_BASE * call_bases / (kern_ms * 1e-3) / 1e9 if kern_ms else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and args.data == "synthetic":
            try:                                            # only while the kernels are the ones the counters were taken with
                import hashlib
                doc = json.load(open(tpath))
                src = os.path.join(ROOT, doc.get("source", "glimmer-mg_amd/csrc/gmg_frame6.hip"))
                if doc.get("source_sha256") == hashlib.sha256(open(src, "rb").read()).hexdigest():
                    traffic = doc.get("%dx%d" % (sizes[0] if n_batches > 1 else n, L))
                else:
                    sys.stderr.write("bench: profiles/traffic.json was measured with another gmg_frame6.hip; roofline.traffic = null "
                                     "(tools/profile_frame6.sh + tools/update_traffic.py renew it)\n")
            except Exception:
                traffic = None
        line = {
            "metric": "Mbases/s scored (6-frame IMM)", "value": round(value, 2), "unit": "Mbases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(seconds / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic" if args.data == "synthetic" else "genome: reads cut uniformly from NC_000915.fna (1.67 Mbp), both strands",
            "config": {"workload": "%s x %d bp %s reads %s%s, one 3-periodic ICM (NC_000915.icm), "
                                   "6-frame per-position scoring, fp64 Frame_Scores"
                                   % ("{:,}".format(args.reads) if n_batches > 1 else str(args.reads), L,
                                      "synthetic" if args.data == "synthetic" else "genome-sampled (NC_000915.fna)",
                                      "per GPU" if args.scaling == "weak" else "in total, sharded by gmg_shard_plan",
                                      " in %d batches of <= %s reads into one reused table" % (n_batches, "{:,}".format(max(sizes))) if n_batches > 1 else ""),
                       "reads_per_gpu": args.reads if args.scaling == "weak" else n, "read_len": L, "batches": n_batches,
                       "parallelism": "reads sharded, %d rank(s)" % world},
            "roofline": {"bound": "hbm", "kernel": "k_frame6t + k_frame6p (one gmg_frame_score6 call)", "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": traffic, "kernel_ms": round(kern_ms, 4), "kernel_ms_min_max": [round(kern_min, 4), round(kern_max, 4)],
                         "statistic": "median of the %d timed calls (HIP events on the launch stream)" % len(per_call),
                         "algorithmic_bytes_per_base": ALGO_BYTES_PER_BASE,
                         "measured_fill_GBps": round(fill_gbps, 1) if fill_gbps else None},
            "timed_region_s": round(seconds, 4),
            "check": None,
        }
        if args.cpu_reads > 0 and n > 0:
            ns = min(args.cpu_reads if world == 1 else 2000, n)
            first_read = first_base // L
            samples = [("first", 0), ("last", n - ns)] if n > ns else [("all", 0)]
            verdicts, head = [], None
            for name, r0 in samples:
                cpu, kind = cpu_sample(first_read + r0, r0, ns, L, seed, gc, packed, from_stream=args.data == "synthetic")
                if head is None:
                    head = (cpu, kind)
                x, mix = device_digest(out, total, r0, ns, L)
                same = x == cpu["xor"] and mix == cpu["mix"]
                ok = ok and same
                verdicts.append("%s %d reads: %s" % (name, ns, "bit-exact" if same else "MISMATCH (xor %s/%s, mix %s/%s)" % (x, cpu["xor"], mix, cpu["mix"])))
            line["check"] = "device table vs the CPU %s (XOR and position-weighted sum of all output doubles) -- %s" % (head[1], "; ".join(verdicts))
            if world == 1:
                cpu, kind = head
                line["cpu_baseline"] = {"value": round(cpu["mbases_per_s"], 4), "unit": "Mbases/s", "cores": 1, "kind": kind,
                                        "sample": "first %d of the rank-0 reads (%d x %d bp, same stream)" % (ns, ns, L),
                                        "seconds": round(cpu["seconds"], 3)}
                if kind == "reference" and args.data == "synthetic":
                    line["cpu_baseline"]["all_cores"] = cpu_all_cores(max(ns // 2, 1), L, seed, gc)
        if not ok:
            line["value"] = None
        if world == 1 and not args.no_cli and ok:
            line["cli_end_to_end"] = cli_end_to_end()
        if world == 1 and not args.no_extras and ok:
            del out                                         # (24 GB back to the device before the other paths run in their own process)
            torch.cuda.empty_cache()
            line["extras"] = extras()
            if isinstance(line["extras"], dict) and (line["extras"].get("error") or line["extras"].get("mismatch")):
                ok = False                                  # a wrong or crashed (f)-row leg fails the whole line
                line["value"] = None
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        sys.exit(1)


if __name__ == "__main__":
    main()
