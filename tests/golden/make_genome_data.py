"""A synthetic Phymm `.genomeData` tree for glimmer-mg's classification mode (-c), built from the reference's sample-run data.

glimmer-mg -c takes, for every class `<strain>|<NC>` of the class file, `<ICM_dir>/<strain>/<NC>.{gicm,gc.txt,motif,gbk,
lengths.*.txt,starts.*.txt,adj_orients.*.txt,adj_dist.*.txt}` (src/Glimmer/glimmer-mg.cc:473-515, 998-1420).  The real tree
is Phymm's 50 GB database (docs/notes.tex:80-81); this script lays out a small one that exercises every branch of that code:

  * every class of a class file gets a directory entry; its gene ICM is one of five sample-run models (symbolic links, so the
    tree stays small), chosen by a CRC of the class name -- reads of one class file therefore spread over many ICM *files*
    (the reference groups by file NAME) and five distinct tables;
  * `.gc.txt`: a CRC-derived value in [0.25, 0.75) with seven digits -- every class its own null model; classes whose CRC ends
    in 0x7 have NO `.gc.txt` (the reference warns and takes 0.5, glimmer-mg.cc:1412-1415);
  * `.gbk` with `/transl_table=4` (stop codons taa, tag only) for classes whose CRC % 11 == 3, `/transl_table=11` for
    CRC % 11 == 5, none for the rest (default 11, glimmer-mg.cc:1246-1248): the STOP CODONS change from read to read;
  * `.motif` for every class (Read_Meta_RBS exits without it, glimmer-mg.cc:1058), the length / start / adjacency
    distributions cut from the sample-run's feature files (two variants of each);
  * double ICMs `<strain1>/<NC1>_2/<strain2>/<NC2>.gicm` for the pairs (first class, later class) whose combined CRC % 3 == 0
    (Classes_ICM_File, glimmer-mg.cc:486-501).

Deterministic: the same class file gives the same tree byte for byte.  Test infrastructure; data only.
"""
import os
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "data")

GICMS = ["seqs.cluster-0.run1.filt.gicm", "seqs.cluster-2.run1.filt.gicm", "seqs.cluster-4.run1.filt.gicm",
         "seqs.cluster-5.run1.filt.gicm", "NC_000915.icm"]
MOTIFS = ["seqs.cluster-%d.run1.filt.motif" % k for k in range(6)]
FEATURES = ["NC_000915.run1.features.txt", "NC_000915.run2.features.txt"]
STARTS = ["seqs.cluster-%d.run1.filt.features.txt" % k for k in range(6)]

SECTION_FILE = {
    ("LENGTH", "GENE"): "lengths.genes.txt", ("LENGTH", "NON"): "lengths.non.txt",
    ("START", "GENE"): "starts.genes.txt", ("START", "NON"): "starts.non.txt",
    ("ADJACENT_ORIENTATION", "GENE"): "adj_orients.genes.txt", ("ADJACENT_ORIENTATION", "NON"): "adj_orients.non.txt",
    ("ADJACENT_DISTANCE_1_1", "GENE"): "adj_dist.1.1.genes.txt", ("ADJACENT_DISTANCE_1_1", "NON"): "adj_dist.1.1.non.txt",
    ("ADJACENT_DISTANCE_1_-1", "GENE"): "adj_dist.1.-1.genes.txt", ("ADJACENT_DISTANCE_1_-1", "NON"): "adj_dist.1.-1.non.txt",
    ("ADJACENT_DISTANCE_-1_1", "GENE"): "adj_dist.-1.1.genes.txt", ("ADJACENT_DISTANCE_-1_1", "NON"): "adj_dist.-1.1.non.txt",
}


def crc(s):
    return zlib.crc32(s.encode()) & 0xFFFFFFFF


def sections(path):
    """{(FEATURE, GENE|NON): text} of a glimmer feature file (Parse_Features, glimmer_base.cc:1197-1320)"""
    out, key, buf = {}, None, []
    for line in open(path):
        if line.startswith("DIST"):
            if key:
                out[key] = "".join(buf)
            _, a, b = line.split()
            key, buf = (a.upper(), b.upper()), []
        elif line.strip():
            buf.append(line)
    if key:
        out[key] = "".join(buf)
    return out


def parse_classes(path):
    """[(read, [class, ...])] in file order"""
    rows = []
    for line in open(path):
        a = line.split()
        if a:
            rows.append((a[0], a[1:]))
    return rows


def class_gc(name):
    """the text of <NC>.gc.txt, or None for a class without one"""
    h = crc(name)
    if (h & 0xF) == 0x7:
        return None
    return "%.7f\n" % (0.25 + 0.5 * ((h >> 8) % 100003) / 100003.0)


def class_transl(name):
    m = crc(name) % 11
    return 4 if m == 3 else 11 if m == 5 else None


def double_pair(a, b):
    return crc(a + "+" + b) % 3 == 0


def build(dest, class_files):
    """lay the tree out under dest for every class named in class_files; returns {class: info}"""
    feats = [sections(os.path.join(DATA, f)) for f in FEATURES]
    starts = [sections(os.path.join(DATA, f)) for f in STARTS]
    # every distinct distribution file once, the classes link to them
    pool = os.path.join(dest, "_dist")
    os.makedirs(pool, exist_ok=True)
    for tag, group in (("f", feats), ("s", starts)):
        for k, sec in enumerate(group):
            for key, fname in SECTION_FILE.items():
                if key in sec:
                    open(os.path.join(pool, "%s%d.%s" % (tag, k, fname)), "w").write(sec[key])
    info = {}
    rows = []
    for cf in class_files:
        rows += parse_classes(cf)
    names = sorted({c for _, cl in rows for c in cl})
    for name in names:
        strain, nc = name.split("|")
        d = os.path.join(dest, strain)
        os.makedirs(d, exist_ok=True)
        h = crc(name)
        gicm = GICMS[h % len(GICMS)]
        link = os.path.join(d, nc + ".gicm")
        if os.path.lexists(link):
            os.unlink(link)
        os.symlink(os.path.join(DATA, gicm), link)
        gc = class_gc(name)
        if gc is not None:
            open(os.path.join(d, nc + ".gc.txt"), "w").write(gc)
        tt = class_transl(name)
        if tt is not None:
            open(os.path.join(d, nc + ".gbk"), "w").write(
                "LOCUS       %s\nFEATURES             Location/Qualifiers\n     CDS             1..300\n"
                "                     /transl_table=%d\n                     /product=\"x\"\n" % (nc, tt))
        open(os.path.join(d, nc + ".motif"), "w").write(open(os.path.join(DATA, MOTIFS[(h >> 4) % len(MOTIFS)])).read())
        for key, fname in SECTION_FILE.items():
            src = "s%d" % ((h >> 9) % len(starts)) if key[0] == "START" else "f%d" % ((h >> 7) & 1)
            link = os.path.join(d, nc + "." + fname)
            if os.path.lexists(link):
                os.unlink(link)
            os.symlink(os.path.join(pool, src + "." + fname), link)
        info[name] = dict(gicm=gicm, gc=gc, transl=tt)
    n_double = 0
    for _, cl in rows:
        for other in cl[1:]:
            if other != cl[0] and double_pair(*sorted([cl[0], other])):
                a, b = sorted([cl[0], other])
                (s1, n1), (s2, n2) = a.split("|"), b.split("|")
                d = os.path.join(dest, s1, n1 + "_2", s2)
                os.makedirs(d, exist_ok=True)
                link = os.path.join(d, n2 + ".gicm")
                if not os.path.lexists(link):
                    os.symlink(os.path.join(DATA, GICMS[crc(a + "&" + b) % len(GICMS)]), link)
                    n_double += 1
    return info, n_double


def write_class_variants(dest_dir):
    """class files next to the sample-run's own (tests/golden/data/seqs.class.txt: 999 reads x 3 classes):
    mixed.class.txt -- the same reads with 1, 2 or 3 classes per line (tabs and spaces), a few reads missing, a few lines for
    reads that are in no FASTA file, and one read listed twice (the later line wins, glimmer-mg.cc:751)"""
    rows = parse_classes(os.path.join(DATA, "seqs.class.txt"))
    out = []
    for i, (read, cl) in enumerate(rows):
        k = crc(read) % 7
        if k == 0:
            continue                                    # a read without classification: never scored
        n = 1 if k in (1, 2) else 2 if k == 3 else 3
        sep = "\t" if i % 2 else " "
        out.append(read + "\t" + sep.join(cl[:n]) + "\n")
        if k == 5 and i % 3 == 0:
            out.append("ghost%d\t%s\n" % (i, cl[0]))   # classified, but not in the input
    out.append(rows[10][0] + "\t" + rows[500][1][0] + "\n")   # a second line for one read
    path = os.path.join(dest_dir, "mixed.class.txt")
    open(path, "w").write("".join(out))
    return path


if __name__ == "__main__":
    dest = sys.argv[1]
    os.makedirs(dest, exist_ok=True)
    mixed = write_class_variants(dest)
    info, n_double = build(os.path.join(dest, ".genomeData"), [os.path.join(DATA, "seqs.class.txt"), mixed])
    print("%d classes, %d double ICMs, %d without gc.txt, %d with a .gbk" % (
        len(info), n_double, sum(1 for v in info.values() if v["gc"] is None), sum(1 for v in info.values() if v["transl"])))
