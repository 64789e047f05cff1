"""One job over several GPUs and several batches: the Python face of include/gmg.h's gmg_shard_plan /
gmg_fasta_shard_ranges / gmg_gc_fraction (host side in glimmer-mg_amd/host/gmg_shard.cc), as bench.py and the
tests drive it.  One process per GPU; what crosses processes is two integers per shard ({gc, total} for the null
model's GC fraction, src/Glimmer/glimmer_base.cc:2564-2595) and, at the end, the compact result records -- never the
reads or the 48 B/base table.  The C++ driver integration/glimmer-mg_gpu.cc does the same with fork + pipes.
"""
import ctypes as C

import numpy as np

from . import api, capi


def shard_plan(offsets, n_shards):
    """gmg_shard_plan: contiguous read ranges of about equal BASE count -> read_begin uint64[n_shards + 1]"""
    offsets = np.ascontiguousarray(offsets, np.uint64)
    out = np.zeros(n_shards + 1, np.uint64)
    api._ck(capi.lib().gmg_shard_plan(api._ptr(offsets), len(offsets) - 1, int(n_shards), api._ptr(out)))
    return out


def batch_plan(offsets, max_bases):
    """batches inside one shard: the same plan with as many parts as keep a batch at about max_bases"""
    offsets = np.ascontiguousarray(offsets, np.uint64)
    total = int(offsets[-1] - offsets[0])
    return shard_plan(offsets, max(1, -(-total // int(max_bases))))


def fasta_shard_ranges(data, n_shards):
    """gmg_fasta_shard_ranges: byte cuts of an unparsed FASTA file at certain record starts -> uint64[n_shards + 1]"""
    out = np.zeros(n_shards + 1, np.uint64)
    api._ck(capi.lib().gmg_fasta_shard_ranges(data, len(data), int(n_shards), api._ptr(out)))
    return out


def gc_fraction(gc_counts, base_counts, as_reference=True):
    gc = np.ascontiguousarray(gc_counts, np.uint64)
    tot = np.ascontiguousarray(base_counts, np.uint64)
    fn = capi.lib().gmg_gc_fraction
    return float(fn(api._ptr(gc), api._ptr(tot), len(gc), int(bool(as_reference))))


def fasta_pieces(data, piece_bytes):
    """gmg_fasta_split: cut points for ingesting a shard in pieces of about piece_bytes"""
    max_pieces = max(2, len(data) // max(int(piece_bytes), 1) + 2)
    cuts = np.zeros(max_pieces + 1, np.uint64)
    n = capi.lib().gmg_fasta_split(data, len(data), int(piece_bytes), api._ptr(cuts), max_pieces)
    if n < 0:
        api._ck(n)
    return cuts[:n + 1]


class MgShard:
    """One rank's share of a glimmer-mg run on one FASTA file:
        sh = MgShard(data, rank, world, piece_bytes)     ingests bytes [cuts[rank], cuts[rank+1]) in pieces
        sh.gc, sh.total                                   this shard's counts -> sum over ranks -> gc_fraction
        sh.score(gene, indep, **options)                  per piece: gmg_mg_score_reads; results concatenated
    """

    def __init__(self, data, rank, world, piece_bytes=1 << 28):
        data = bytes(data)
        cuts = fasta_shard_ranges(data, world)
        self.byte_range = (int(cuts[rank]), int(cuts[rank + 1]))
        mine = data[self.byte_range[0]:self.byte_range[1]]
        self.pieces, self.headers = [], []
        self.gc = self.total = self.n_reads = 0
        if len(mine):
            pc = fasta_pieces(mine, piece_bytes)
            for a, b in zip(pc[:-1], pc[1:]):
                reads, hdrs, gc = api.Reads.from_fasta_bytes(mine[int(a):int(b)])
                self.pieces.append(reads)
                self.headers += hdrs
                self.gc += gc
                self.total += reads.total_bases
                self.n_reads += reads.n_reads

    def score(self, gene, indep, **kw):
        """-> (orfs, starts, read_orf_off) of the shard, reads numbered from 0 inside the shard, start_begin and
        read_orf_off rebased so that the pieces read as ONE batch"""
        return concat_results([api.mg_score_reads(gene, indep, r, **kw)[:3] for r in self.pieces],
                              [r.n_reads for r in self.pieces])


def concat_results(parts, n_reads_of):
    """results of consecutive batches -> one result, as if the batches had been one (read indices, start_begin and
    read_orf_off shifted by what precedes them)"""
    orfs, starts, offs = [], [], [np.zeros(1, np.uint64)]
    r0 = o0 = s0 = 0
    for (o, s, off), n in zip(parts, n_reads_of):
        o = o.copy()
        o["read"] += np.uint32(r0)
        o["start_begin"] += np.uint32(s0)
        orfs.append(o)
        starts.append(s)
        offs.append(off[1:] + np.uint64(o0))
        r0 += n
        o0 += len(o)
        s0 += len(s)
    if not orfs:
        return np.zeros(0, api.MG_ORF_DTYPE), np.zeros(0, api.START_DTYPE), np.zeros(1, np.uint64)
    return np.concatenate(orfs), np.concatenate(starts), np.concatenate(offs)


def allreduce_counts(dist, gc, total):
    """the job's {gc, total} from every rank's: two integers per rank through the process group (gloo or RCCL)"""
    if dist is None:
        return [gc], [total]
    import torch
    t = torch.zeros(2 * dist.get_world_size(), dtype=torch.int64)
    t[2 * dist.get_rank()], t[2 * dist.get_rank() + 1] = gc, total
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t)
    t = t.cpu().numpy()
    return t[0::2].astype(np.uint64), t[1::2].astype(np.uint64)


def gather_results(dist, part, n_reads):
    """rank 0 gets the shards' results concatenated in shard order (one host-side gather of compact records)"""
    if dist is None:
        return part
    objs = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object((part, n_reads), objs, dst=0)
    if dist.get_rank() != 0:
        return None
    return concat_results([p for p, _ in objs], [n for _, n in objs])
