"""The multi-GPU split, proven on one GPU (BASELINE configs[2] / configs[4]):
  * two gloo ranks that REALLY score on device 0: each ingests its byte range of seqs.fa, the job's GC comes from the
    all-reduced {gc, total} counts, the gathered records equal the single-batch bytes and the reference's goldens;
  * one GPU's share of "100M x 500 bp over 8 GPUs": 12.5M reads in 1M-read batches through the property checks;
  * the 32-bit index fields of the result records: a batch that would overflow them is refused (GMG_ETOOBIG), the
    same reads in planned batches go through."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import DATA, GOLD, ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch.distributed as dist
import _gmg_pkg
gmg = _gmg_pkg.load()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
gmg.init(0)                                             # both ranks on the one GPU of the box
data = open(sys.argv[2], "rb").read()
model = gmg.Icm.open(sys.argv[3])
sh = gmg.shard.MgShard(data, rank, world, piece_bytes=int(sys.argv[5]))
gcs, totals = gmg.shard.allreduce_counts(dist, sh.gc, sh.total)     # two integers per rank: the only exchange before scoring
gc = gmg.shard.gc_fraction(gcs, totals)
indep = gmg.Icm.indep(gc)
part = sh.score(model, indep)
res = gmg.shard.gather_results(dist, part, sh.n_reads)
hdrs = [None] * world if rank == 0 else None
dist.gather_object(sh.headers, hdrs, dst=0)
if rank == 0:
    orfs, starts, off = res
    np.savez(sys.argv[4], orfs=orfs, starts=starts, off=off, gc=gc, n_pieces=len(sh.pieces),
             headers=np.array([h for part in hdrs for h in part], dtype=object), counts=np.array([gcs, totals]))
dist.barrier()
dist.destroy_process_group()
'''


@pytest.fixture(scope="module")
def nc(gpu):
    return gpu.Icm.open(os.path.join(DATA, "NC_000915.icm"))


@pytest.mark.parametrize("piece_bytes", [1 << 28, 60_000])
def test_two_ranks_score_their_shards_and_concatenate_to_the_single_batch(gpu, nc, seqs_fa, tmp_path, piece_bytes):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = str(tmp_path / "gathered.npz")
    fasta = os.path.join(DATA, "seqs.fa")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, fasta, os.path.join(DATA, "NC_000915.icm"), out, str(piece_bytes)],
                              env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-3000:]
    g = np.load(out, allow_pickle=True)
    # the job's GC fraction is the whole file's (glimmer_base.cc:2564-2595), bit for bit
    gold = np.load(os.path.join(GOLD, "frames_nc.npz"))
    assert float(g["gc"]) == float(gold["gc"])
    assert int(g["counts"][1].sum()) == sum(len(s) for s in seqs_fa[1])
    assert [h.decode() if isinstance(h, bytes) else h for h in g["headers"]] == seqs_fa[0]
    if piece_bytes < 1 << 20:
        assert int(g["n_pieces"]) > 2                       # rank 0 really went through several batches
    # single batch, this process
    reads = gpu.Reads.from_strings(seqs_fa[1])
    orfs, starts, off = gpu.mg_score_reads(nc, gpu.Icm.indep(float(gold["gc"])), reads)
    assert g["orfs"].tobytes() == orfs.tobytes() and g["starts"].tobytes() == starts.tobytes() and np.array_equal(g["off"], off)
    # ... and the reference's own ORF lists / accepted genes for that file (tests/golden/mg_orfs_default.npz)
    ref = np.load(os.path.join(GOLD, "mg_orfs_default.npz"))
    got = np.stack([g["orfs"]["read"].astype(np.int32), g["orfs"]["frame"], g["orfs"]["stop_position"], g["orfs"]["gene_len"],
                    g["orfs"]["orf_len"]], 1)
    assert np.array_equal(got, ref["orfs"])
    accepted = np.zeros(len(got), bool)
    accepted[ref["gene_orf"]] = True
    assert np.array_equal(g["orfs"]["accepted"] != 0, accepted)


def test_result_index_fields_refuse_to_wrap(gpu, nc):
    """gmg_mg_orf.start_begin / n_orfs are 32-bit: a batch beyond them is an error (GMG_ETOOBIG), not a wrap; the batch plan
    of include/gmg.h cuts the same reads into batches that pass.  (The limit is lowered for the test: 2^31 entries need
    150M reads.)"""
    n, L = 20_000, 500
    packed, off = gpu.synth.packed_reads(n, L, 5)
    reads = gpu.Reads(packed, off)
    indep = gpu.Icm.indep(0.5)
    full = gpu.mg_score_reads(nc, indep, reads)
    n_orfs, n_starts = len(full[0]), len(full[1])
    with gpu.option("mg_max_entries", n_orfs - 1):
        with pytest.raises(gpu.GmgError) as e:
            gpu.mg_score_reads(nc, indep, reads)
        assert e.value.code == -7 and "split the batch" in str(e.value)
    with gpu.option("mg_max_entries", n_starts - 1):        # ORFs fit, starts do not
        with pytest.raises(gpu.GmgError) as e:
            gpu.mg_score_reads(nc, indep, reads)
        assert e.value.code == -7
        # planned batches of about a quarter of the bases each go through and concatenate to the full result
        plan = gpu.shard.batch_plan(off, n * L // 4 + 1)
        parts, counts = [], []
        for a, b in zip(plan[:-1], plan[1:]):
            a, b = int(a), int(b)
            sub = reads.select(np.arange(a, b))
            parts.append(gpu.mg_score_reads(nc, indep, sub))
            counts.append(b - a)
        cat = gpu.shard.concat_results(parts, counts)
    assert cat[0].tobytes() == full[0].tobytes() and cat[1].tobytes() == full[1].tobytes() and np.array_equal(cat[2], full[2])
    with gpu.option("mg_max_entries", 10):                  # glimmer3's batch entry guards the same way
        with pytest.raises(gpu.GmgError) as e:
            gpu.score_orfs(nc, indep, reads, [(0, 1, 400, 300)] * 11)
        assert e.value.code == -7


def test_one_gpus_share_of_configs2_in_batches(gpu, oracle, nc):
    """BASELINE configs[2]: 100M x 500 bp over 8 GPUs = 12.5M reads (6.25 Gbases) per GPU.  Its 300 GB Frame_Scores table
    cannot exist, so the share runs as the product does: base-balanced batches (gmg_shard_plan), each through
    gmg_mg_score_reads with only the accepted ORFs leaving the GPU.  Properties: every batch is deterministic (first and
    last batch scored twice), the per-batch record counts add up, the plan covers the share's 6.25e9 bases (beyond 2^32)
    exactly once, sampled reads of every batch -- incl. the very last read of the share -- equal the oracle."""
    n, L, per_batch = 12_500_000, 500, 1_000_000
    off_all = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    plan = gpu.shard.batch_plan(off_all, per_batch * L)
    assert len(plan) - 1 == 13 and int(plan[-1]) == n
    indep = gpu.Icm.indep(0.5)
    o_nc, o_indep, prm = oracle.read(os.path.join(DATA, "NC_000915.icm")), oracle.indep(0.5), oracle.mg_params()
    rng = np.random.default_rng(3)
    total_orfs = total_starts = 0
    for bi, (a, b) in enumerate(zip(plan[:-1], plan[1:])):
        a, b = int(a), int(b)
        # the job is ONE stream of bases (synth.packed_reads): batch bi holds bases [a*L, b*L) of the stream of seed 99
        packed, off = gpu.synth.packed_reads_range(a * L, (b - a) * L, L, 99)
        reads = gpu.Reads(packed, off)
        res = gpu.mg_score_reads(nc, indep, reads, accepted_only=True)
        if bi in (0, len(plan) - 2):
            again = gpu.mg_score_reads(nc, indep, reads, accepted_only=True)
            assert all(x.tobytes() == y.tobytes() for x, y in zip(res, again))
        orfs, starts, first = res
        assert len(orfs) == first[-1] and int(orfs["n_starts"].sum()) == len(starts) and np.all(orfs["accepted"] != 0)
        total_orfs += len(orfs)
        total_starts += len(starts)
        sample = [0, b - a - 1] + [int(x) for x in rng.integers(0, b - a, 3)]
        for r in sample:
            seq = gpu.synth.unpack_ascii(packed, r * L, L)
            want_orfs, scored = oracle.mg_read(o_nc, o_indep, seq, prm)
            mine = orfs[int(first[r]):int(first[r + 1])]
            want = [(wo, sc) for wo, sc in zip(want_orfs, scored) if sc[0].accepted]
            assert len(mine) == len(want)
            for o, (wo, (out, wst)) in zip(mine, want):
                assert (o["frame"], o["stop_position"], o["gene_len"], o["orf_len"]) == tuple(wo)
                st = starts[o["start_begin"]:o["start_begin"] + o["n_starts"]]
                assert [(s["j"], s["pos"], s["which"], s["score"]) for s in st] == [(w.j, w.pos, w.which, w.score) for w in wst]
        del reads, res, orfs, starts
    assert total_orfs > 0.02 * 7 * n * 0.5 and total_starts > total_orfs      # (random reads: ~2.4 % of ~7.6 ORFs per read accepted)


def test_bench_batches_one_reused_table(gpu):
    """bench.py --batches: a rank's reads scored as B batches into ONE reused table (how BASELINE configs[2]'s 12.5M reads per GPU
    run: --reads 12500000 --batches 13) -- here 30,001 reads in 4 batches (7,501 + 7,501 + 7,501 + 7,498: the last one shorter): the line's contract fields, the
    workload string, the check on the LAST batch (bit-exact against the CPU), value = all reads' bases over the timed region"""
    import json
    import subprocess
    import sys
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--reads", "30001", "--batches", "4", "--steps", "3", "--warmup", "1",
                          "--cpu-reads", "500", "--no-cli", "--no-extras"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    line = json.loads(res.stdout.decode().strip().splitlines()[-1])
    assert line["config"]["batches"] == 4 and line["config"]["reads_per_gpu"] == 30001 and "in 4 batches of <= 7,501 reads" in line["config"]["workload"]
    assert line["value"] and abs(line["value"] - 30001 * 500 / (line["ms_per_step"] * 1e-3) / 1e6) < 0.01 * line["value"]   # all four batches per step
    assert "MISMATCH" not in line["check"] and line["check"].count("bit-exact") == 2
    assert line["roofline"]["frac"] > 0 and line["scaling"] == "weak" and line["dtype"] == "f64"
