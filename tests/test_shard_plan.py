"""Host side of the multi-GPU split (include/gmg.h: gmg_shard_plan, gmg_fasta_shard_ranges, gmg_gc_fraction;
SURVEY.md 8e).  No GPU: these are pure host functions of the C-ABI library."""
import os

import numpy as np
import pytest

from conftest import DATA


def offsets_of(lengths):
    off = np.zeros(len(lengths) + 1, np.uint64)
    np.cumsum(np.asarray(lengths, np.uint64), out=off[1:])
    return off


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_shard_plan_is_contiguous_and_balanced_by_bases(gmg, seed, world):
    rng = np.random.default_rng(seed)
    lengths = np.concatenate([rng.integers(0, 900, 5000), [0, 0, 40_000, 1, 0]])      # ragged, empty reads, one giant
    rng.shuffle(lengths)
    off = offsets_of(lengths)
    plan = gmg.shard.shard_plan(off, world)
    assert plan[0] == 0 and plan[-1] == len(lengths) and np.all(np.diff(plan.astype(np.int64)) >= 0)
    total = int(off[-1])
    longest = int(lengths.max())
    for k in range(1, world):                               # every cut sits within one read of its ideal base position
        assert abs(int(off[int(plan[k])]) - k * total // world) <= longest
    # the nearest boundary, not just any: moving a cut by one read never gets closer
    for k in range(1, world):
        c, want = int(plan[k]), k * total // world
        here = abs(int(off[c]) - want)
        if c > 0:
            assert abs(int(off[c - 1]) - want) >= here or int(plan[k - 1]) > c - 1
        if c < len(lengths):
            assert abs(int(off[c + 1]) - want) >= here


def test_shard_plan_edge_cases(gmg):
    assert list(gmg.shard.shard_plan(offsets_of([]), 4)) == [0, 0, 0, 0, 0]
    assert list(gmg.shard.shard_plan(offsets_of([10]), 3))[0::3] == [0, 1]
    plan = gmg.shard.shard_plan(offsets_of([500] * 1000), 8)
    assert list(np.diff(plan.astype(np.int64))) == [125] * 8                          # uniform reads: equal shares
    plan = gmg.shard.shard_plan(offsets_of([5, 5, 5]), 8)                              # more shards than reads
    assert plan[0] == 0 and plan[-1] == 3 and np.all(np.diff(plan.astype(np.int64)) >= 0)
    with pytest.raises(gmg.GmgError):
        gmg.shard.shard_plan(offsets_of([1, 2]), 0)
    # batches inside a shard: the same plan, sized by a base budget
    bp = gmg.shard.batch_plan(offsets_of([500] * 1000), 100_000)
    assert len(bp) == 6 and list(np.diff(bp.astype(np.int64))) == [200] * 5


@pytest.mark.parametrize("name", ["seqs.fa", "nasty.fa"])
@pytest.mark.parametrize("world", [1, 2, 3, 7])
def test_fasta_shard_ranges_cut_at_record_starts(gmg, name, world):
    data = open(os.path.join(DATA, name), "rb").read()
    cuts = gmg.shard.fasta_shard_ranges(data, world)
    assert cuts[0] == 0 and cuts[-1] == len(data) and np.all(np.diff(cuts.astype(np.int64)) >= 0)
    for c in cuts[1:-1]:
        c = int(c)
        assert c == len(data) or (data[c:c + 1] == b">" and data[c - 1:c] == b"\n")
    # the shards parse to the file's records, in order (host parser with the reference's Fasta_Read semantics)
    import tempfile
    whole = gmg.read_fasta(os.path.join(DATA, name))
    hdrs, seqs = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        with tempfile.NamedTemporaryFile(suffix=".fa") as fh:
            fh.write(data[int(a):int(b)])
            fh.flush()
            h, s = gmg.read_fasta(fh.name)
        hdrs += h
        seqs += s
    if name == "seqs.fa":                                   # (nasty.fa has junk before its first record: shard 0 keeps it)
        assert hdrs == whole[0]
    assert seqs == whole[1]


def test_gc_fraction_sums_shards_and_mimics_the_reference_counters(gmg):
    f = gmg.shard.gc_fraction
    assert f([10, 30], [100, 100]) == 40 / 200
    assert f([1, 2, 3], [10, 10, 10], as_reference=False) == 6 / 30
    # Set_GC_Fraction counts in `unsigned int` (glimmer_base.cc:2570): beyond 2^32 bases its counters wrap
    gc, total = [2 ** 32 + 5, 7], [2 ** 33, 2 ** 32 + 100]
    assert f(gc, total, as_reference=True) == 12 / 100
    assert f(gc, total, as_reference=False) == (2 ** 32 + 12) / (3 * 2 ** 32 + 100)
    assert f([0], [0], as_reference=False) == 0.0
