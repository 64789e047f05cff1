#!/usr/bin/env python3
"""Generate tests/golden/* from the REAL reference (oracle/_ref, built by oracle/Makefile from
/root/reference).  Runs only in the build container; the GPU box and the tests use the committed
fixtures.  Test infrastructure only.

Inputs copied as DATA fixtures (the reference's own sample-run inputs/outputs):
  tests/golden/data/seqs.fa, NC_000915.icm, cluster-{0..5}.icm, seqs.cluster-4.run1.filt.gicm,
  icm-{0..5}.scores.tmp

Vectors produced by oracle/_ref/ref_dump (our driver over the reference's ICM_t):
  frames_nc.npz     Score_All_Frames (glimmer-mg.cc:1468-1510) for the first 48 reads + sha256 over all 999
  frames_gicm.npz   same with the small 3-periodic model seqs.cluster-4.run1.filt.gicm, first 16 reads
  sstring.npz       Score_String (icm.cc:864-903) 999 reads x frames 0,1,2 for NC_000915.icm, cluster-4.icm
  segs.npz          Cumulative_Score (icm.cc:354-405) gene+indep on ORF-style buffers (glimmer3.cc:1322-1347)
                    and the six Score_String values of All_Frame_Score (glimmer3.cc:346-354)
  windows.npz       Full_Window_Prob / Full_Window_Distrib (icm.cc:512-610) on 4096 random 12-mers x 3 frames
  partial.npz       Partial_Window_Prob (icm.cc:807-842) for every prefix position of 64 reads x 3 frames
  indep.npz         Build_Indep_WO_Stops tables (icm.cc:65-216) for GC x stop-codon sets
  predict/*.predict reference CLI outputs on seqs.fa (glimmer3, glimmer3 -X..., glimmer-mg, glimmer-mg -i)
  predict/NC_000915.run1.predict   the reference's own committed answer file (sample-run/glimmer3/results; scripts/g3-iterated.py:58),
                    copied as data and re-made here as a check; NC_000915.{step6,glimmer-mg,glimmer3.X_l}.predict: the same genome
                    through step 6's shape (-b motif -m gicm), glimmer-mg and glimmer3 -X -l
  revcodon.npz      Build_Reverse_Codon_WO_Stops (icm.cc:219-350) models + probe scores
  find_orfs_general.npz   Find_Orfs (glimmer_base.cc:638-817) with ignore regions and on circular sequences: every Orf_t of six genome
                    slices under six option sets (oracle/_ref/ref_orfs orfs | orfs-circular)
"""
import hashlib
import os
import shutil
import struct
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("GMG_REFERENCE", "/root/reference")
RB = os.path.join(HERE, "_ref")
GOLD = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(GOLD, "data")


def run(*args):
    return subprocess.run([os.path.join(RB, "ref_dump"), *map(str, args)], check=True,
                          stdout=subprocess.PIPE).stdout


def parse_icm_stream(buf):
    """Binary .icm stream -> (params, mip[P][N] int16, prob[P][N][4] float32); icm.cc:614-726."""
    ver, idl, W, D, P, N = struct.unpack_from("<6i", buf, 150)
    mip = np.zeros((P, N), np.int16)
    prob = np.zeros((P, N, 4), np.float32)
    seen = np.zeros((P, N), bool)
    off, period = 174, -1
    while True:
        (nid,) = struct.unpack_from("<i", buf, off)
        off += 4
        if nid < 0:
            break
        if nid == 0:
            period += 1
        prob[period, nid] = struct.unpack_from("<4f", buf, off)
        (mip[period, nid],) = struct.unpack_from("<h", buf, off + 16)
        seen[period, nid] = True
        off += 18
    mip[~seen] = -2
    return (W, D, P, N), mip, prob


def main():
    if not os.path.exists(os.path.join(RB, "ref_dump")):
        sys.exit("build oracle/_ref first:  make -C oracle ref")
    os.makedirs(DATA, exist_ok=True)
    os.makedirs(os.path.join(GOLD, "predict"), exist_ok=True)
    sr = os.path.join(REF, "sample-run")
    copies = {"seqs.fa": "glimmer-mg/seqs.fa", "NC_000915.icm": "glimmer3/results/NC_000915.icm",
              "seqs.cluster-4.run1.filt.gicm": "glimmer-mg/results/seqs.cluster-4.run1.filt.gicm"}
    for i in range(6):
        copies["cluster-%d.icm" % i] = "glimmer-mg/results/cluster-%d.icm" % i
        copies["icm-%d.scores.tmp" % i] = "glimmer-mg/results/icm-%d.scores.tmp" % i
    # the reference's whole-path known-answer case (scripts/g3-iterated.py:58: glimmer3 -u -12 -m NC_000915.icm NC_000915.fna):
    # the genome, the answer file it holds, and the second iteration's model + RBS matrix (step 6's inputs, g3-iterated.py:74)
    copies.update({"NC_000915.fna": "glimmer3/NC_000915.fna", "NC_000915.run1.gicm": "glimmer3/results/NC_000915.run1.gicm",
                   "NC_000915.run1.motif": "glimmer3/results/NC_000915.run1.motif"})
    for dst, src in copies.items():
        shutil.copyfile(os.path.join(sr, src), os.path.join(DATA, dst))
        os.chmod(os.path.join(DATA, dst), 0o644)
    shutil.copyfile(os.path.join(sr, "glimmer3/results/NC_000915.run1.predict"), os.path.join(GOLD, "predict", "NC_000915.run1.predict"))
    os.chmod(os.path.join(GOLD, "predict", "NC_000915.run1.predict"), 0o644)

    fa = os.path.join(DATA, "seqs.fa")
    nc = os.path.join(DATA, "NC_000915.icm")
    gicm = os.path.join(DATA, "seqs.cluster-4.run1.filt.gicm")
    c4 = os.path.join(DATA, "cluster-4.icm")
    L, NREADS = 500, 999

    (gc,) = struct.unpack("<d", run("gc", fa))

    # ---- frames
    nf = 48
    fr = np.frombuffer(run("frames", nc, fa, 0, nf, -1), "<f8").reshape(nf, 6, L)
    allf = run("frames", nc, fa, 0, NREADS, -1)
    np.savez_compressed(os.path.join(GOLD, "frames_nc.npz"), frames=fr, gc=gc, n_all=NREADS,
                        sha256_all=hashlib.sha256(allf).hexdigest())
    nf2 = 16
    fr2 = np.frombuffer(run("frames", gicm, fa, 0, nf2, 0.5, "taa,tag"), "<f8").reshape(nf2, 6, L)
    np.savez_compressed(os.path.join(GOLD, "frames_gicm.npz"), frames=fr2, gc=0.5, stops="taa,tag")

    # ---- Score_String on whole reads
    ss_nc = np.frombuffer(run("sstring", nc, fa), "<f8").reshape(NREADS, 3)
    ss_c4 = np.frombuffer(run("sstring", c4, fa), "<f8").reshape(NREADS, 3)
    np.savez_compressed(os.path.join(GOLD, "sstring.npz"), nc=ss_nc, cluster4=ss_c4)

    # ---- ORF-style segments: (read, lo, len, strand)
    rng = np.random.default_rng(20260101)
    segs = []
    for r in range(24):
        for _ in range(6):
            ln = int(rng.integers(1, L + 1))
            lo = int(rng.integers(0, L - ln + 1))
            segs.append((r, lo, ln, 1 if rng.integers(2) else -1))
    for ln in (1, 2, 10, 11, 12, 13, 500):        # edges around model_len-1 = 11
        segs.append((30, 0, ln, 1))
        segs.append((31, L - ln, ln, -1))
    segs = np.array(segs, np.int32)
    segfile = os.path.join(RB, "segs.txt")
    np.savetxt(segfile, segs, fmt="%d")
    raw = np.frombuffer(run("segs", nc, fa, segfile, -1), "<f8")
    gene_cum, indep_cum, off = [], [], 0
    for _, _, ln, _ in segs:
        gene_cum.append(raw[off:off + ln]); off += ln
        indep_cum.append(raw[off:off + ln]); off += ln
    assert off == raw.size
    af = np.frombuffer(run("allframe", nc, fa, segfile), "<f8").reshape(len(segs), 6)
    np.savez_compressed(os.path.join(GOLD, "segs.npz"), segs=segs, gene_cum=np.concatenate(gene_cum),
                        indep_cum=np.concatenate(indep_cum), allframe_raw=af, gc=gc)

    # ---- random windows
    nw = 4096
    raw = run("windows", nc, 12345, nw)
    rec = 12 + 3 * (8 + 16)
    wins = np.frombuffer(raw, np.uint8).reshape(nw, rec)
    wchars = wins[:, :12].copy()
    tail = wins[:, 12:].copy().reshape(nw, 3, 24)
    wprob = tail[:, :, :8].copy().view("<f8").reshape(nw, 3)
    wdist = tail[:, :, 8:].copy().view("<f4").reshape(nw, 3, 4)
    np.savez_compressed(os.path.join(GOLD, "windows.npz"), windows=wchars, prob=wprob, dist=wdist)

    # ---- partial windows
    npart = 64
    part = np.frombuffer(run("partial", nc, fa, npart), "<f8").reshape(npart, 3, 11)
    part_c4 = np.frombuffer(run("partial", c4, fa, npart), "<f8").reshape(npart, 1, 11)
    np.savez_compressed(os.path.join(GOLD, "partial.npz"), nc=part, cluster4=part_c4)

    # ---- null models
    out = {}
    for gcv in (0.25, 0.39, 0.5, 0.65, gc):
        for stops in ("taa,tag,tga", "taa,tag"):
            _, mip, prob = parse_icm_stream(run("indep", repr(float(gcv)), stops))
            key = "gc%.17g_%s" % (gcv, stops.replace(",", "-"))
            out[key + "_prob"] = prob
            # the writer drops nodes with mip < -1 only; calloc'ed nodes come back as written
            out[key + "_mip"] = mip
    np.savez_compressed(os.path.join(GOLD, "indep.npz"), seqs_gc=gc, **out)

    # ---- Build_Reverse_Codon_WO_Stops (icm.cc:219-350; public, no caller in the reference): seeded codon weights x stop sets
    rc = {}
    for seed, stops in ((1, "taa,tag,tga"), (20260105, "taa,tag"), (7, "tga")):
        rc["s%d_%s" % (seed, stops.replace(",", "_"))] = np.frombuffer(run("revcodon", seed, stops), np.uint8)
    np.savez_compressed(os.path.join(GOLD, "revcodon.npz"), **rc)

    # ---- writer round trip (icm.cc:729-803): reference Read -> Output must reproduce the file
    tmp = os.path.join(RB, "rewrite.icm")
    run("rewrite", nc, tmp)
    assert open(tmp, "rb").read() == open(nc, "rb").read(), "reference writer is not a fixed point?"

    # ---- CLI outputs
    clis = {
        "glimmer3.default": ["glimmer3", "-m", nc],
        "glimmer3.X": ["glimmer3", "-X", "-m", nc],
        "glimmer-mg.default": ["glimmer-mg", "-m", nc],
        "glimmer-mg.indel": ["glimmer-mg", "-i", "-m", nc],
        "glimmer-mg.g120": ["glimmer-mg", "-g", "120", "-m", nc],
        "glimmer-mg.Z2": ["glimmer-mg", "-Z", "taa,tag", "-m", nc],
        "glimmer-mg.sub": ["glimmer-mg", "-s", "-m", nc],
    }
    for name, cmd in clis.items():
        tag = os.path.join(RB, "cli_" + name)
        subprocess.run([os.path.join(RB, cmd[0]), *cmd[1:], fa, tag], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=RB)
        shutil.copyfile(tag + ".predict", os.path.join(GOLD, "predict", name + ".predict"))
    # ---- the whole path at genome scale: one 1.67 Mbp sequence (ORFs of several kb, ~100 k ORFs in one record list)
    fna = os.path.join(DATA, "NC_000915.fna")
    genome_clis = {
        "NC_000915.run1.rerun": ["glimmer3", "-u", "-12", "-m", nc],          # must reproduce the reference-held answer file
        "NC_000915.step6": ["glimmer3", "-b", os.path.join(DATA, "NC_000915.run1.motif"), "-m", os.path.join(DATA, "NC_000915.run1.gicm")],
        "NC_000915.glimmer-mg": ["glimmer-mg", "-m", nc],
        "NC_000915.glimmer3.X_l": ["glimmer3", "-X", "-l", "-m", nc],
    }
    for name, cmd in genome_clis.items():
        tag = os.path.join(RB, "cli_" + name)
        subprocess.run([os.path.join(RB, cmd[0]), *cmd[1:], fna, tag], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=RB)
        if name.endswith(".rerun"):
            assert open(tag + ".predict", "rb").read() == open(os.path.join(GOLD, "predict", "NC_000915.run1.predict"), "rb").read(), \
                "the reference built here does not reproduce sample-run/glimmer3/results/NC_000915.run1.predict"
        else:
            shutil.copyfile(tag + ".predict", os.path.join(GOLD, "predict", name + ".predict"))
    # ---- Find_Orfs in full (glimmer_base.cc:638-817): ignore regions (glimmer3 -i) and circular sequences (Wrap_Around_Back /
    #      Wrap_Through_Front), on slices of the sample genome -- the slices are named here and cut again by the tests
    genome = "".join(line.strip() for line in open(fna) if not line.startswith(">"))
    fo_slices = [(400_000, 30_000), (900_000, 4_000), (100_000, 12_001), (500_000, 900), (100, 95), (1_200_000, 20_002)]
    fo_fa = os.path.join(RB, "find_orfs_general.fa")
    with open(fo_fa, "w") as f:
        for k, (at, ln) in enumerate(fo_slices):
            f.write(">s%d\n%s\n" % (k, genome[at:at + ln]))
    fo_ign = os.path.join(RB, "find_orfs_general.ignore")
    with open(fo_ign, "w") as f:
        f.write("# lo hi\n1 40\n700 1300\n1200 1500\n2990 3005\n9000 8000\n11000 13000\n19990 20002 trailing words\n25000 25001\n")
    fo_ign2 = os.path.join(RB, "find_orfs_general.ignore2")         # (with the first file the reference itself aborts on a circular sequence:
    with open(fo_ign2, "w") as f:                                   #  Wrap_Around_Back's assert (pos > 0) behind a region that reaches the end)
        f.write("700 1300\n2990 3005\n")
    fo = {"slices": np.array(fo_slices, np.int64)}
    for name, mode, opts in (("ignore", "orfs", ["-i", fo_ign]), ("ignore_X_g60", "orfs", ["-X", "-g", "60", "-i", fo_ign]),
                             ("circular", "orfs-circular", []), ("circular_X_Z2", "orfs-circular", ["-X", "-Z", "taa,tag"]),
                             ("circular_ignore", "orfs-circular", ["-i", fo_ign2]), ("plain_g60", "orfs", ["-g", "60"])):
        txt = subprocess.run([os.path.join(RB, "ref_orfs"), mode, *opts, fo_fa, os.path.join(RB, "fo_tag")], check=True,
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, cwd=RB).stdout.decode()
        rows, regions, read = [], [], -1
        for line in txt.splitlines():
            p_ = line.split()
            if p_[0] == "I":
                regions.append((int(p_[1]), int(p_[2])))
            elif p_[0] == "R":
                read = int(p_[1])
            elif p_[0] == "O":
                rows.append((read, int(p_[1]), int(p_[2]), int(p_[3]), int(p_[4])))
        fo[name + "_orfs"] = np.array(rows, np.int32).reshape(-1, 5)          # sequence, frame, stop_position, gene_len, orf_len
        fo[name + "_regions"] = np.array(regions, np.int32).reshape(-1, 2)
        fo[name + "_opts"] = " ".join(o for o in opts if o not in (fo_ign, fo_ign2))
    np.savez_compressed(os.path.join(GOLD, "find_orfs_general.npz"), **fo)
    # ---- Score_Orfs inner loop (glimmer3.cc:1275-1552): ORFs from Find_Orfs + the start lists handed to Add_Events_*
    for name, flags in (("orfs_default", []), ("orfs_X", ["-X"]), ("orfs_g90_first", ["-g", "90", "-f", "x"])):
        txt = subprocess.run([os.path.join(RB, "ref_orfs"), "dump", *flags, "-m", nc, fa, os.path.join(RB, "orfs_tag")],
                             check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, cwd=RB).stdout.decode()
        orfs, genes, st_i, st_s = [], [], [], []
        read, base = -1, 0
        for line in txt.splitlines():
            p = line.split()
            if p[0] == "R":
                read, base = int(p[1]), len(orfs)
            elif p[0] == "O":
                orfs.append((read, int(p[1]), int(p[2]), int(p[3])))
            elif p[0] == "G":
                genes.append((base + int(p[1]), int(p[3]), int(p[4]), len(st_i), float(p[2])))
            elif p[0] == "S":
                st_i.append((int(p[1]), int(p[2]), int(p[4]), int(p[5]), int(p[6])))
                st_s.append(float.fromhex(p[3]))
        g = np.array([x[:4] for x in genes], np.int64)
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), flags=" ".join(flags),
                            orfs=np.array(orfs, np.int32),                       # read, frame, stop_position, orf_len
                            gene_orf=g[:, 0], gene_len=g[:, 1], gene_nstarts=g[:, 2], gene_start_begin=g[:, 3],
                            gene_score=np.array([x[4] for x in genes], np.float64),
                            start_int=np.array(st_i, np.int32),                  # j, pos, which, truncated, first
                            start_score=np.array(st_s, np.float64))
    # ---- glimmer-mg front half (Find_Orfs + Score_Orfs_Errors, glimmer-mg.cc:1605-1861): every ORF of Find_Orfs and
    #      the sorted start lists handed to Add_Events_*
    for name, flags in (("mg_orfs_default", []), ("mg_orfs_g120", ["-g", "120"]), ("mg_orfs_Z2", ["-Z", "taa,tag"])):
        txt = subprocess.run([os.path.join(RB, "ref_mg_orfs"), "dump", *flags, "-m", nc, fa, os.path.join(RB, "mg_tag")],
                             check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, cwd=RB).stdout.decode()
        orfs, genes, st_i, st_s = [], [], [], []
        read, base = -1, 0
        for line in txt.splitlines():
            p = line.split()
            if p[0] == "R":
                read, base = int(p[1]), len(orfs)
            elif p[0] == "O":
                orfs.append((read, int(p[1]), int(p[2]), int(p[3]), int(p[4])))
            elif p[0] == "G":
                genes.append((base + int(p[1]), int(p[2]), len(st_i)))
            elif p[0] == "S":
                st_i.append((int(p[1]), int(p[2]), int(p[4]), int(p[5]), int(p[6])))
                st_s.append(float.fromhex(p[3]))
        g = np.array(genes, np.int64)
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), flags=" ".join(flags),
                            orfs=np.array(orfs, np.int32),                 # read, frame, stop_position, gene_len, orf_len
                            gene_orf=g[:, 0], gene_nstarts=g[:, 1], gene_start_begin=g[:, 2],
                            start_int=np.array(st_i, np.int32),            # j, pos, which, truncated, first (sorted by pos)
                            start_score=np.array(st_s, np.float64))
    # ---- glimmer-mg's error branch (-i indels, -s substitutions, -q quality file; Score_Indels / Score_Orf_Starts,
    #      glimmer-mg.cc:1513-1861): the start lists in PUSH order (seen right before the reference's sort) with errors
    recs = open(fa).read().split(">")[1:]
    sub_fa, sub_q = os.path.join(GOLD, "data", "seqs80.fa"), os.path.join(GOLD, "data", "seqs80.qual")
    rng = np.random.default_rng(20260102)
    with open(sub_fa, "w") as f, open(sub_q, "w") as q:
        for rec in recs[:80]:
            hdr, seq = rec.split("\n", 1)
            seq = seq.replace("\n", "")
            f.write(">%s\n%s\n" % (hdr, seq))
            # Phred values, one in ten at or below the indel threshold, incl. 0 (Clean_Quality_454 lifts those to 1)
            vals = np.where(rng.random(len(seq)) < 0.1, rng.integers(0, 19, len(seq)), rng.integers(19, 41, len(seq)))
            q.write(">%s\n" % hdr)
            for a in range(0, len(vals), 25):
                q.write(" ".join(str(int(v)) for v in vals[a:a + 25]) + "\n")
    tag = os.path.join(RB, "cli_glimmer-mg.indel_q80")                  # -i with the quality file, through the whole CLI
    subprocess.run([os.path.join(RB, "glimmer-mg"), "-i", "-q", sub_q, "-m", nc, sub_fa, tag], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=RB)
    shutil.copyfile(tag + ".predict", os.path.join(GOLD, "predict", "glimmer-mg.indel_q80.predict"))
    for name, flags in (("mg_err_indel", ["-i"]), ("mg_err_sub", ["-s"]), ("mg_err_indel_q", ["-i", "-q", sub_q]),
                        ("mg_err_indel_g90", ["-i", "-g", "90", "-Z", "taa,tag"])):
        txt = subprocess.run([os.path.join(RB, "ref_mg_orfs"), "dump", *flags, "-m", nc, sub_fa, os.path.join(RB, "mg_tag")],
                             check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, cwd=RB).stdout.decode()
        orfs, genes, st_i, st_s = [], [], [], []
        read, base = -1, 0
        for line in txt.splitlines():
            p = line.split()
            if p[0] == "R":
                read, base = int(p[1]), len(orfs)
            elif p[0] == "O":
                orfs.append((read, int(p[1]), int(p[2]), int(p[3]), int(p[4])))
            elif p[0] == "G":
                genes.append((base + int(p[1]), int(p[2]), len(st_i)))
            elif p[0] == "S":
                ne = int(p[7])
                err = [int(x) for x in p[8:8 + 2 * ne]] + [0] * (4 - 2 * ne)
                st_i.append((int(p[1]), int(p[2]), int(p[4]), int(p[5]), int(p[6]), ne, *err))
                st_s.append(float.fromhex(p[3]))
        g = np.array(genes, np.int64)
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), flags=" ".join(os.path.basename(x) for x in flags),
                            orfs=np.array(orfs, np.int32),                 # read, frame, stop_position, gene_len, orf_len
                            gene_orf=g[:, 0], gene_nstarts=g[:, 1], gene_start_begin=g[:, 2],
                            # j, pos, which, truncated, first, n_errors, (pos, type) x 2 -- in push order
                            start_int=np.array(st_i, np.int32),
                            start_score=np.array(st_s, np.float64))
    # ---- FASTA ingest (Fasta_Read, fasta.cc:236-286): a deliberately awkward file, parsed by the reference
    nasty = (b"junk before the first record\n>  first read  with spaces \nACGTacgt\nNNRYKM\n\n  acgt \t ggg\r\n"
             b">second>has>gt in header\nacgtacgtacgtacgtacgtacgtacgtacgtacgt\n"
             b">empty\n>also empty\n\n\n>mid line gt\nacgt>weird  \naaa\nccc\n"
             b">iupac\nRYSWKMBDHVNrysw kmbdhvn*-.0123xXuU\n>crlf\r\nac\r\ngt\r\n>tabs\tin\theader\nT\tT\tT\vG\fG\n"
             b">long\n" + b"\n".join(bytes((b"acgt"[(i * 7 + j * 3) % 4]) for j in range(61)) for i in range(40)) +
             b"\n>last record without newline\nacgtn")
    fpath = os.path.join(GOLD, "data", "nasty.fa")
    open(fpath, "wb").write(nasty)
    txt = subprocess.run([os.path.join(RB, "ref_dump"), "fasta", fpath], check=True, stdout=subprocess.PIPE).stdout
    open(os.path.join(GOLD, "nasty.fasta_read.txt"), "wb").write(txt)
    txt = subprocess.run([os.path.join(RB, "ref_dump"), "fasta", fa], check=True, stdout=subprocess.PIPE).stdout
    open(os.path.join(GOLD, "seqs.fasta_read.sha256"), "w").write(hashlib.sha256(txt).hexdigest() + "\n")
    print("golden vectors written to", GOLD)


if __name__ == "__main__":
    main()
