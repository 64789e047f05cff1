"""64 DISTINCT models for the configs[3] / configs[4] shapes: the sample run's own model files plus models trained here
(gmg_icm_train = ICM_Training_t::Train_Model, the counts on the device) on disjoint slices of tests/golden/data/NC_000915.fna.
  period1_models(gmg, tmp_dir)   the 6 cluster-*.icm + 58 period-1 models (build-icm -p 1 shape: 12 / 7 / 1)   -> [(Icm, path)]
  gene_models(gmg, tmp_dir, n)   NC_000915.icm + the 4 sample .gicm + (n - 5) 3-periodic models (12 / 7 / 3)     -> [(Icm, path)]
Every trained model is also written to tmp_dir as an .icm file, so that the oracle (and anything else) reads the same tables.
  relabeled_models(gmg, tmp_dir, files, n)   n different tables with the VALUES of real models: each sample file with the four bases
                                 renamed by a permutation (children reordered, prob entries permuted): what differs from the
                                 trained sets is only the value range (small training sets give the trained tables probabilities
                                 of zero, which send them to the exact paths) -- the cache sees n different tables either way
Used by tests/test_gpu_strings.py, tests/test_gpu_classes.py and tests/bench/."""
import itertools
import os
import struct

import numpy as np

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data")
GENE_FILES = ["NC_000915.icm", "seqs.cluster-0.run1.filt.gicm", "seqs.cluster-2.run1.filt.gicm", "seqs.cluster-4.run1.filt.gicm",
              "seqs.cluster-5.run1.filt.gicm"]
_genome = None


def genome():
    global _genome
    if _genome is None:
        _genome = "".join(line.strip() for line in open(os.path.join(DATA, "NC_000915.fna")) if not line.startswith(">")).lower()
        _genome = "".join(c if c in "acgt" else "c" for c in _genome)        # (Filter: everything else is c)
    return _genome


def _slices(n, piece):
    """n disjoint slices of the genome, each cut into strings of `piece` bases"""
    g = genome()
    span = len(g) // n
    return [[g[k * span + i:k * span + i + piece] for i in range(0, span - piece + 1, piece)] for k in range(n)]


def period1_models(gmg, tmp_dir, n=64):
    out = [(gmg.Icm.open(os.path.join(DATA, "cluster-%d.icm" % i)), os.path.join(DATA, "cluster-%d.icm" % i)) for i in range(min(6, n))]
    for k, strings in enumerate(_slices(max(n - 6, 0), 1000)):
        m = gmg.Icm.train(strings, 12, 7, 1)
        path = os.path.join(str(tmp_dir), "p1_%02d.icm" % k)
        m.write(path)
        out.append((m, path))
    return out


def gene_models(gmg, tmp_dir, n=64):
    out = [(gmg.Icm.open(os.path.join(DATA, f)), os.path.join(DATA, f)) for f in GENE_FILES[:min(5, n)]]
    for k, strings in enumerate(_slices(max(n - 5, 0), 900)):          # 900 = whole codons: every string starts in frame 0
        m = gmg.Icm.train(strings, 12, 7, 3)
        path = os.path.join(str(tmp_dir), "g3_%02d.icm" % k)
        m.write(path)
        out.append((m, path))
    return out


def _read_icm(path):
    """binary .icm (icm.cc:614-726): 150 header bytes, six int32, records {int32 id, 4 float32 prob, int16 mip}, frames begin at id 0"""
    buf = open(path, "rb").read()
    ver, idl, W, D, P, N = struct.unpack_from("<6i", buf, 150)
    frames, off = [], 174
    while True:
        (nid,) = struct.unpack_from("<i", buf, off)
        off += 4
        if nid < 0:
            break
        if nid == 0:
            frames.append({})
        frames[-1][nid] = (struct.unpack_from("<4f", buf, off), struct.unpack_from("<h", buf, off + 16)[0])
        off += 18
    return buf[:174], frames


def _relabel(path_in, perm, path_out):
    """the model in which base b is called perm[b]: node (c1, .., ck) becomes (perm[c1], .., perm[ck]), prob'[perm[b]] = prob[b]"""
    head, frames = _read_icm(path_in)

    def new_id(nid):
        path = []
        while nid > 0:
            path.append((nid - 1) % 4)
            nid = (nid - 1) // 4
        out = 0
        for c in reversed(path):
            out = 4 * out + perm[c] + 1
        return out

    with open(path_out, "wb") as f:
        f.write(head)
        for fr in frames:
            recs = sorted((new_id(nid), prob, mip) for nid, (prob, mip) in fr.items())
            for nid, prob, mip in recs:
                q = [0.0] * 4
                for b in range(4):
                    q[perm[b]] = prob[b]
                f.write(struct.pack("<i4fh", nid, *q, mip))
        f.write(struct.pack("<i", -1))


def relabeled_models(gmg, tmp_dir, files, n=64):
    perms = [p for p in itertools.permutations(range(4))]
    out = []
    for k in range(n):
        src = os.path.join(DATA, files[k % len(files)])
        perm = perms[k // len(files)]
        if perm == (0, 1, 2, 3):
            out.append((gmg.Icm.open(src), src))
            continue
        path = os.path.join(str(tmp_dir), "relabel_%02d.icm" % k)
        _relabel(src, perm, path)
        out.append((gmg.Icm.open(path), path))
    return out
