"""Pins the oracle's restatement of Fasta_Read (src/Common/fasta.cc:236-286) + tolower (Filter (ch)) to the real
reference: tests/golden/nasty.fasta_read.txt is what oracle/_ref/ref_dump (linked against the reference's fasta.o /
gene.o) printed for tests/golden/data/nasty.fa -- junk before the first record, '>' in header lines and in the middle
of sequence lines, blank lines, CR LF, tabs, empty records, IUPAC and garbage letters, no newline at the end -- and
seqs.fasta_read.sha256 is the hash of the same dump for the 999 reads of seqs.fa.  CPU only."""
import hashlib
import os

from conftest import DATA, GOLD


def dump(records, gc):
    out = b"".join(b"H " + h + b"\nS " + s + b"\n" for h, s in records)
    return out + b"G %d %d\n" % (gc, sum(len(s) for _, s in records))


def test_fasta_read_nasty_file(oracle):
    data = open(os.path.join(DATA, "nasty.fa"), "rb").read()
    records, gc = oracle.fasta_records(data)
    assert dump(records, gc) == open(os.path.join(GOLD, "nasty.fasta_read.txt"), "rb").read()
    assert len(records) == 11 and records[2] == (b"empty", b"")


def test_fasta_read_seqs_fa(oracle):
    data = open(os.path.join(DATA, "seqs.fa"), "rb").read()
    records, gc = oracle.fasta_records(data)
    assert len(records) == 999
    want = open(os.path.join(GOLD, "seqs.fasta_read.sha256")).read().strip()
    assert hashlib.sha256(dump(records, gc)).hexdigest() == want
