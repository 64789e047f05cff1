// ref_bench.cc -- times the REAL reference ICM_t (icm.o from /root/reference, built by
// oracle/Makefile into oracle/_ref/) on the six-frame per-position loop of
// src/Glimmer/glimmer-mg.cc:1468-1510 (Score_All_Frames).  This file is ours; it calls only
// the reference's public interface (src/ICM/icm.hh:131-180).  Used by bench.py's
// cpu_baseline leg ("kind": "reference") and by tests as a cross-check of the oracle.
//
//   ref_bench <gene.icm> <n_reads> <L> <seed> <gc_frac> [first_read]
//   ref_bench <gene.icm> <n_reads> <L> @<reads.fa> <gc_frac>        the first n_reads records of a FASTA file (each L bases) instead
//                                                                  of the synthetic stream: bench.py --data genome hands its sample over this way
//
// Synthetic reads: the job is one stream of 2-bit bases; 64-bit word k of the stream is
// SplitMix64 output number k+1 of <seed> (z = mix(seed + (k+1)*0x9E3779B97F4A7C15)), base j of
// the word is (z >> 2j) & 3 -> "acgt".  Read r is bases [r*L, (r+1)*L) of the stream.  bench.py
// and the tests generate the same stream with numpy.
//
// Prints one JSON line: {"bases":..., "seconds":..., "mbases_per_s":..., "xor":"<hex>", "mix":"<hex>", "sum":...}
// where xor is the XOR of the bit patterns of every output double (order-independent checksum) and mix the
// position-sensitive one: sum over the sample's elements of bits * (2 * index + 1) mod 2^64, index = (f * n_reads + r) * L + i
// (the [6][n_reads * L] layout of the sample) -- a permuted or shifted table does not pass it.

#include "icm.hh"
#include "fasta.hh"
#include "gene.hh"
#include <string>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdint.h>
#include <time.h>

using namespace std;

static inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static inline char stream_base(uint64_t seed, uint64_t g)
{
    uint64_t z = mix64(seed + (g / 32 + 1) * 0x9E3779B97F4A7C15ULL);
    return "acgt"[(z >> (2 * (g % 32))) & 3];
}

static double now_s()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int main(int argc, char **argv)
{
    if (argc < 6) {
        fprintf(stderr, "usage: ref_bench <gene.icm> <n_reads> <L> <seed> <gc_frac> [first_read]\n");
        return 2;
    }
    ICM_t gene;
    gene.Read(argv[1]);
    long n_reads = atol(argv[2]);
    int L = atoi(argv[3]);
    uint64_t seed = strtoull(argv[4], NULL, 10);
    double gc = atof(argv[5]);
    long first = argc > 6 ? atol(argv[6]) : 0;

    ICM_t indep(3, 2, 3);
    vector<const char *> stops;
    stops.push_back("taa"); stops.push_back("tag"); stops.push_back("tga");   // glimmer_base.hh default stop codons
    indep.Build_Indep_WO_Stops(gc, stops);

    vector<string> reads(n_reads);
    if (argv[4][0] == '@') {                            // the sample as a FASTA file, read the way glimmer-mg reads its input
        FILE *fp = fopen(argv[4] + 1, "r");
        if (fp == NULL) { fprintf(stderr, "ref_bench: cannot open %s\n", argv[4] + 1); return 2; }
        string hdr;
        for (long r = 0; r < n_reads; r++) {
            if (!Fasta_Read(fp, reads[r], hdr) || (int)reads[r].length() != L) { fprintf(stderr, "ref_bench: record %ld of %s is not %d bases\n", r, argv[4] + 1, L); return 2; }
            for (int j = 0; j < L; j++) reads[r][j] = tolower(Filter(reads[r][j]));      // glimmer-mg.cc:381-382
        }
        fclose(fp);
    } else
    for (long r = 0; r < n_reads; r++) {
        reads[r].resize(L);
        for (int j = 0; j < L; j++) reads[r][j] = stream_base(seed, (uint64_t)(first + r) * L + j);
    }

    vector<vector<double> > fs(6);
    vector<double> g, z;
    string buff;
    uint64_t x = 0, mix = 0;
    double sum = 0.0;

    double t0 = now_s();
    for (long r = 0; r < n_reads; r++) {
        const string &S = reads[r];
        buff.assign(S.rbegin(), S.rend());                 // Reverse_Transfer(buff, Sequence, L-1, L)
        for (int f = 0; f < 3; f++) {
            gene.Frame_Score(buff, g, f);
            indep.Frame_Score(buff, z, f);
            fs[f].resize(L);
            for (int i = 0; i < L; i++) fs[f][i] = g[L - 1 - i] - z[L - 1 - i];
        }
        buff.resize(L);
        for (int i = 0; i < L; i++) buff[i] = Complement(S[i]);   // Complement_Transfer(buff, Sequence, 0, L)
        for (int f = 0; f < 3; f++) {
            gene.Frame_Score(buff, g, f);
            indep.Frame_Score(buff, z, f);
            fs[3 + f].resize(L);
            for (int i = 0; i < L; i++) fs[3 + f][i] = g[i] - z[i];
        }
        for (int f = 0; f < 6; f++)
            for (int i = 0; i < L; i++) {
                uint64_t b;
                memcpy(&b, &fs[f][i], 8);
                x ^= b;
                mix += b * (2 * (((uint64_t)f * n_reads + r) * L + i) + 1);
                sum += fs[f][i];
            }
    }
    double t1 = now_s();
    double bases = double(n_reads) * L;
    printf("{\"bases\": %.0f, \"seconds\": %.6f, \"mbases_per_s\": %.4f, \"xor\": \"%016llx\", \"mix\": \"%016llx\", \"sum\": %.17g}\n",
           bases, t1 - t0, bases / (t1 - t0) / 1e6, (unsigned long long)x, (unsigned long long)mix, sum);
    return 0;
}
