#!/usr/bin/env python3
"""Golden vectors for glimmer-mg's classification mode (-c) from the REAL reference (oracle/_ref/ref_mg_classes: the
reference's own main() with ICM_dir pointed at the synthetic .genomeData tree of tests/golden/make_genome_data.py).
Runs only in the build container; the tests use the committed fixtures.  Test infrastructure only.

  tests/golden/predict/classes.<case>.predict     the reference's <tag>.predict, bytes
  tests/golden/classes_<case>.npz                 read by read IN THE REFERENCE'S PROCESSING ORDER:
      reads      header prefix of every processed read              icm / icm_files   its ICM file (index, names relative to ICM_dir)
      gc         Indep_GC_Frac of Update_Meta_Null_ICM (double)     isl               Ignore_Score_Len
      transl     Genbank_Xlate_Code of Update_Meta_Stop             n_stops, stops    Stop_Codon of the read ("taa,tag,tga")
      orf_off, orfs[frame, stop_position, gene_len, orf_len]        every ORF Find_Orfs produced
      acc_read, acc[frame, stop_position, n_starts]                 every ORF handed to Add_Events_* (reads < list_reads only)
      st_off, st_j, st_pos, st_score, st_which, st_trunc, st_first, st_nerr, st_epos[2], st_etype[2]
                                                                    its start list in the order Score_Orf_Starts pushed it
Cases: the sample-run's own seqs.class.txt (999 reads x 3 classes, 240 ICM files) in the default mode, with -i, with -g 90;
mixed.class.txt (1 - 3 classes per read, unclassified reads, ghost reads, a duplicate line) in chunks of 250 reads, default
mode and -s; and -m <icm> together with -c (glimmer-mg.py's --long-orfs path: one ICM, one GC, stop codons per read)."""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(GOLD, "data")
sys.path.insert(0, GOLD)
import make_genome_data  # noqa: E402

CASES = [
    # name, glimmer-mg flags, class file, chunk (None: the reference's 500000), reads whose start lists are kept
    ("default", [], "seqs.class.txt", None, 999),
    ("indel", ["-i"], "seqs.class.txt", None, 120),
    ("g90", ["-g", "90"], "seqs.class.txt", None, 999),
    ("mixed_chunks", [], "mixed.class.txt", 250, 999),
    ("mixed_sub", ["-s"], "mixed.class.txt", None, 400),
    ("user_icm", ["-m", os.path.join(DATA, "NC_000915.icm")], "seqs.class.txt", None, 999),
]


def parse_dump(text, icm_dir, list_reads):
    reads, gc, isl, transl, nstops, stops, icm_names, icm_idx = [], [], [], [], [], [], [], []
    orf_off, orfs = [0], []
    acc_read, acc, st_off = [], [], [0]
    st = dict(j=[], pos=[], score=[], which=[], trunc=[], first=[], nerr=[], epos=[], etype=[])
    names = {}
    for line in text.splitlines():
        f = line.split()
        if f[0] == "R":
            n_st = int(f[6])
            reads.append(f[1]); gc.append(float.fromhex(f[3])); isl.append(int(f[4])); transl.append(int(f[5]))
            nstops.append(n_st); stops.append(",".join(f[7:7 + n_st]))
            name = f[7 + n_st]
            name = name[len(icm_dir):] if name.startswith(icm_dir) else os.path.basename(name)
            icm_idx.append(names.setdefault(name, len(names)))
            orf_off.append(orf_off[-1])
        elif f[0] == "O":
            orfs.append([int(x) for x in f[1:5]])
            orf_off[-1] += 1
        elif f[0] == "G":
            keep = len(reads) - 1 < list_reads
            if keep:
                acc_read.append(len(reads) - 1); acc.append([int(x) for x in f[1:4]]); st_off.append(st_off[-1])
        elif f[0] == "S" and keep:
            st["j"].append(int(f[1])); st["pos"].append(int(f[2])); st["score"].append(float.fromhex(f[3]))
            st["which"].append(int(f[4])); st["trunc"].append(int(f[5])); st["first"].append(int(f[6]))
            ne = int(f[7]); st["nerr"].append(ne)
            e = [int(x) for x in f[8:8 + 2 * ne]] + [0] * (4 - 2 * ne)
            st["epos"].append([e[0], e[2]]); st["etype"].append([e[1], e[3]])
            st_off[-1] += 1
    icm_files = [n for n, _ in sorted(names.items(), key=lambda kv: kv[1])]
    return dict(reads=np.array(reads), gc=np.array(gc, np.float64), isl=np.array(isl, np.int64), transl=np.array(transl, np.int32),
                n_stops=np.array(nstops, np.int32), stops=np.array(stops), icm=np.array(icm_idx, np.int32), icm_files=np.array(icm_files),
                orf_off=np.array(orf_off, np.int64), orfs=np.array(orfs, np.int32).reshape(-1, 4),
                acc_read=np.array(acc_read, np.int32), acc=np.array(acc, np.int32).reshape(-1, 3), st_off=np.array(st_off, np.int64),
                st_j=np.array(st["j"], np.int32), st_pos=np.array(st["pos"], np.int32), st_score=np.array(st["score"], np.float64),
                st_which=np.array(st["which"], np.int32), st_trunc=np.array(st["trunc"], np.int8), st_first=np.array(st["first"], np.int8),
                st_nerr=np.array(st["nerr"], np.int8), st_epos=np.array(st["epos"], np.int32).reshape(-1, 2),
                st_etype=np.array(st["etype"], np.int8).reshape(-1, 2), list_reads=list_reads)


def main():
    exe = os.path.join(HERE, "_ref", "ref_mg_classes")
    tmp = tempfile.mkdtemp(prefix="gmg_classes_")
    try:
        mixed = make_genome_data.write_class_variants(tmp)
        icm_dir = os.path.join(tmp, ".genomeData")
        info, n_double = make_genome_data.build(icm_dir, [os.path.join(DATA, "seqs.class.txt"), mixed])
        shutil.copy(mixed, os.path.join(DATA, "mixed.class.txt"))
        os.makedirs(os.path.join(GOLD, "predict"), exist_ok=True)
        for name, flags, cls, chunk, list_reads in CASES:
            cls_path = mixed if cls == "mixed.class.txt" else os.path.join(DATA, cls)
            # the iteration order of ICM_Sequences depends on the hash of the whole file NAME, ICM_dir included: the goldens
            # are made with the relative name ".genomeData" (cwd = the directory that holds it), which every test can repeat
            env = dict(os.environ, GMG_REF_ICM_DIR=".genomeData")
            if chunk:
                env["GMG_REF_CHUNK"] = str(chunk)
            tag = os.path.join(tmp, name)
            out = subprocess.run([exe, *flags, "-c", cls_path, os.path.join(DATA, "seqs.fa"), tag], check=True, env=env, cwd=tmp,
                                 stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
            shutil.copy(tag + ".predict", os.path.join(GOLD, "predict", "classes.%s.predict" % name))
            d = parse_dump(out, ".genomeData" + os.sep, list_reads)
            np.savez_compressed(os.path.join(GOLD, "classes_%s.npz" % name), flags=" ".join(os.path.basename(x) for x in flags),
                                class_file=cls, chunk=chunk or 0, **d)
            print("%-14s %4d reads, %3d ICM files, %3d distinct GCs, %d stop sets, %6d ORFs, %5d accepted, %7d starts" % (
                name, len(d["reads"]), len(d["icm_files"]), len(set(d["gc"].tolist())), len(set(d["stops"].tolist())),
                len(d["orfs"]), len(d["acc"]), len(d["st_j"])))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
