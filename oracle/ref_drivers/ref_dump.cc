// ref_dump.cc -- golden-vector dumper linked against the REAL reference objects
// (icm.o, gene.o, delcher.o, fasta.o, kelley.o built from /root/reference by
// oracle/Makefile into oracle/_ref/).  This file is ours; it only calls the
// reference's public ICM_t interface (src/ICM/icm.hh:131-180) and
// Fasta_Read / Filter / Complement (src/Common).  Test infrastructure only.
//
// Output: raw little-endian records on stdout; oracle/gen_golden.py frames them
// into tests/golden/*.npz.
//
//   ref_dump frames   <icm> <fasta> <first> <count> <gc|-1>   6*L doubles per read (Score_All_Frames semantics)
//   ref_dump sstring  <icm> <fasta>                            3 doubles per read  (Score_String frame 0,1,2)
//   ref_dump segs     <icm> <fasta> <segfile> <gc|-1>          per segment: gene cum[len], indep cum[len]
//   ref_dump allframe <icm> <fasta> <segfile>                  per segment: 6 Score_String values of All_Frame_Score order (unpermuted)
//   ref_dump windows  <icm> <seed> <count>                     per window x frame: double prob, 4 float dist
//   ref_dump partial  <icm> <fasta> <count>                    per read x frame x pos<min(W-1,L): double
//   ref_dump indep    <gc> <stop1,stop2,...>                   63 x {int16 mip, 4 float}
//   ref_dump gc       <fasta>                                  1 double (Set_GC_Fraction semantics)
//   ref_dump rewrite  <icm> <out.icm>                          ICM_t::Read then ::Output(binary)
//   ref_dump fasta    <fasta>                                  text: per record "H <hdr>" and "S <tolower(Filter(seq))>", then "G <gc count> <total>"
//   ref_dump cumstr   <icm> <fasta> <count>                    per read x frame: Cumulative_Score_String (icm.cc:409-452), len + 1 doubles
//   ref_dump text     <icm>                                    ICM_t::Output (fp, false): the text form (Output_Node, icm.cc:729-803)
//   ref_dump display  <icm>                                    ICM_t::Display (icm.cc:455-482)
//   ref_dump copy     <icm> <fasta> <count>                    ICM_t::Copy (icm.cc:1000-1007) into a second object, then Score_String on it
//   ref_dump revcodon <seed> <stop1,stop2,...> [model]         Build_Reverse_Codon_WO_Stops (icm.cc:219-350) on 64 seeded codon weights: the
//                                                              model through ::Output(binary), then 9 Score_String doubles of three probe strings
//
// The same file is built a second time against glimmer-mg_amd/host/icm.hh + libgmg.so (integration/Makefile: ref_dump_dropin): the
// C++ interface of the drop-in, method by method, against the reference's (tests/test_gpu_icm_class.py).

#include "delcher.hh"
#include "gene.hh"
#include "icm.hh"
#include "fasta.hh"
#include <string>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdint.h>
#include <unistd.h>

using namespace std;

static void put(const void *p, size_t n) { fwrite(p, 1, n, stdout); }
static void put_d(double x) { put(&x, 8); }

struct Read_Set {
    vector<string> seq, hdr;
};

// glimmer3.cc:270-271 / glimmer-mg.cc:381-382: Sequence[i] = tolower(Filter(Sequence[i]))
static void load_reads(const char *path, Read_Set &rs)
{
    FILE *fp = fopen(path, "r");
    if (!fp) { fprintf(stderr, "cannot open %s\n", path); exit(1); }
    string s, h;
    while (Fasta_Read(fp, s, h)) {
        for (size_t i = 0; i < s.length(); i++) s[i] = tolower(Filter(s[i]));
        rs.seq.push_back(s);
        rs.hdr.push_back(h);
    }
    fclose(fp);
}

// glimmer_base.cc:2564-2595 Set_GC_Fraction: Filter(tolower(c)), count g/c over all sequences
static double gc_fraction(const char *path)
{
    FILE *fp = fopen(path, "r");
    if (!fp) { fprintf(stderr, "cannot open %s\n", path); exit(1); }
    string s, h;
    unsigned ct = 0, total = 0;
    while (Fasta_Read(fp, s, h)) {
        total += s.length();
        for (size_t j = 0; j < s.length(); j++) {
            char c = Filter(tolower(s[j]));
            if (c == 'g' || c == 'c') ct++;
        }
    }
    fclose(fp);
    return double(ct) / total;
}

static void build_indep(ICM_t &indep, double gc, const char *stops_csv)
{
    static vector<string> keep;
    vector<const char *> stops;
    string csv = stops_csv ? stops_csv : "taa,tag,tga";
    size_t a = 0;
    keep.clear();
    while (a <= csv.size()) {
        size_t b = csv.find(',', a);
        if (b == string::npos) b = csv.size();
        keep.push_back(csv.substr(a, b - a));
        a = b + 1;
    }
    for (size_t i = 0; i < keep.size(); i++) stops.push_back(keep[i].c_str());
    indep.Build_Indep_WO_Stops(gc, stops);
}

struct Seg { int read, lo, len, strand; };

static void load_segs(const char *path, vector<Seg> &v)
{
    FILE *fp = fopen(path, "r");
    if (!fp) { fprintf(stderr, "cannot open %s\n", path); exit(1); }
    Seg s;
    while (fscanf(fp, "%d %d %d %d", &s.read, &s.lo, &s.len, &s.strand) == 4) v.push_back(s);
    fclose(fp);
}

// glimmer3.cc:1322-1343: forward ORF buffer = bases hi-1 .. lo (reversed, not complemented);
// reverse ORF buffer = complement of lo .. hi-1 (not reversed)
static string seg_buffer(const string &S, const Seg &g)
{
    string b(g.len, 'a');
    if (g.strand > 0)
        for (int j = 0; j < g.len; j++) b[j] = S[g.lo + g.len - 1 - j];
    else
        for (int j = 0; j < g.len; j++) b[j] = Complement(S[g.lo + j]);
    return b;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: ref_dump <cmd> ...\n"); return 2; }
    string cmd = argv[1];

    if (cmd == "fasta") {                               // Fasta_Read (fasta.cc:236-286) + the callers' filter + Set_GC_Fraction's count
        Read_Set rs;
        load_reads(argv[2], rs);
        unsigned long ct = 0, total = 0;
        for (size_t i = 0; i < rs.seq.size(); i++) {
            printf("H %s\nS %s\n", rs.hdr[i].c_str(), rs.seq[i].c_str());
            total += rs.seq[i].length();
            for (size_t k = 0; k < rs.seq[i].length(); k++) ct += (rs.seq[i][k] == 'g' || rs.seq[i][k] == 'c');
        }
        printf("G %lu %lu\n", ct, total);
        return 0;
    }
    if (cmd == "gc") {
        put_d(gc_fraction(argv[2]));
        return 0;
    }
    if (cmd == "indep") {
        ICM_t indep(3, 2, 3);
        build_indep(indep, atof(argv[2]), argc > 3 ? argv[3] : NULL);
        // dump through the public writer, then the caller parses the .icm stream
        indep.Output(stdout, true);
        return 0;
    }
    if (cmd == "revcodon") {                            // Build_Reverse_Codon_WO_Stops (icm.cc:219-350; no caller in the reference)
        // 64 codon weights from a seed (SplitMix64; the method normalises them), stop codons from the command line
        unsigned long long x = strtoull(argv[2], NULL, 10);
        double cp[64];
        for (int j = 0; j < 64; j++) {
            x += 0x9E3779B97F4A7C15ULL;
            unsigned long long z = x;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
            z ^= z >> 31;
            cp[j] = 0.001 + double(z >> 11) / 9007199254740992.0;
        }
        static vector<string> keep;
        vector<const char *> stops;
        string csv = argc > 3 ? argv[3] : "taa,tag,tga";
        for (size_t a = 0; a <= csv.size(); ) {
            size_t b = csv.find(',', a);
            if (b == string::npos) b = csv.size();
            if (b > a) keep.push_back(csv.substr(a, b - a));
            a = b + 1;
        }
        for (size_t i = 0; i < keep.size(); i++) stops.push_back(keep[i].c_str());
        ICM_t m(3, 2, 3);
        m.Build_Reverse_Codon_WO_Stops(cp, stops);
        m.Output(stdout, true);
        if (argc > 4 && !strcmp(argv[4], "model")) return 0;       // the tables alone (needs no device in the drop-in build)
        // and what the model then scores: three short strings in every frame
        const char *probe[3] = {"atggcgtaaacgtgatag", "ttagcatcacgcgcgcat", "acgtacgtacgtaacc"};
        for (int s = 0; s < 3; s++)
            for (int f = 0; f < 3; f++)
                put_d(m.Score_String(probe[s], strlen(probe[s]), f));
        return 0;
    }
    if (cmd == "rewrite") {
        ICM_t icm;
        icm.Read(argv[2]);
        FILE *fp = fopen(argv[3], "wb");
        icm.Output(fp, true);
        fclose(fp);
        return 0;
    }

    ICM_t gene;
    gene.Read(argv[2]);
    int W = gene.Get_Model_Len();
    int P = gene.Get_Periodicity();

    if (cmd == "windows") {
        uint64_t x = strtoull(argv[3], NULL, 10);
        int count = atoi(argv[4]);
        vector<char> w(W + 1, 0);
        for (int i = 0; i < count; i++) {
            for (int k = 0; k < W; k++) {
                // SplitMix64
                x += 0x9E3779B97F4A7C15ULL;
                uint64_t z = x;
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
                z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
                z ^= z >> 31;
                w[k] = "acgt"[z & 3];
            }
            put(&w[0], W);
            for (int f = 0; f < P; f++) {
                float dist[4];
                put_d(gene.Full_Window_Prob(&w[0], f));
                gene.Full_Window_Distrib(&w[0], f, dist);
                put(dist, 16);
            }
        }
        return 0;
    }

    if (cmd == "text") {
        gene.Output(stdout, false);
        return 0;
    }
    if (cmd == "display") {
        gene.Display(stdout);
        return 0;
    }

    Read_Set rs;
    load_reads(argv[3], rs);

    if (cmd == "cumstr") {
        int count = atoi(argv[4]);
        for (int r = 0; r < count && r < (int)rs.seq.size(); r++) {
            string S = rs.seq[r];
            int L = S.length();
            if (L < W - 1) continue;                    // (the reference scores the first model_len - 1 bases whatever len is)
            vector<double> cum(L + 1);
            for (int f = 0; f < P; f++) {
                gene.Cumulative_Score_String(&S[0], L, f, &cum[0]);
                put(&cum[0], 8 * (L + 1));
            }
        }
        return 0;
    }
    if (cmd == "copy") {
        // Copy shares the tables (the reference frees them twice if both objects die): the copy is never destroyed
        ICM_t *twin = new ICM_t;
        twin->Copy(gene);
        int count = atoi(argv[4]);
        for (int r = 0; r < count && r < (int)rs.seq.size(); r++)
            put_d(twin->Score_String(rs.seq[r].c_str(), rs.seq[r].length(), r % P));
        fflush(stdout);
        _exit(0);
    }

    if (cmd == "frames") {
        int first = atoi(argv[4]), count = atoi(argv[5]);
        double gc = atof(argv[6]);
        if (gc < 0) gc = gc_fraction(argv[3]);
        ICM_t indep(3, 2, 3);
        build_indep(indep, gc, argc > 7 ? argv[7] : NULL);
        vector<double> g, z;
        for (int r = first; r < first + count && r < (int)rs.seq.size(); r++) {
            const string &S = rs.seq[r];
            int L = S.length();
            string rev(S.rbegin(), S.rend()), comp(S);
            for (int i = 0; i < L; i++) comp[i] = Complement(S[i]);
            vector<double> row(L);
            for (int f = 0; f < 3; f++) {   // glimmer-mg.cc:1485-1494
                gene.Frame_Score(rev, g, f);
                indep.Frame_Score(rev, z, f);
                for (int i = 0; i < L; i++) row[i] = g[L - 1 - i] - z[L - 1 - i];
                put(&row[0], 8 * L);
            }
            for (int f = 0; f < 3; f++) {   // glimmer-mg.cc:1500-1509
                gene.Frame_Score(comp, g, f);
                indep.Frame_Score(comp, z, f);
                for (int i = 0; i < L; i++) row[i] = g[i] - z[i];
                put(&row[0], 8 * L);
            }
        }
        return 0;
    }
    if (cmd == "sstring") {
        for (size_t r = 0; r < rs.seq.size(); r++)
            for (int f = 0; f < 3; f++)
                put_d(gene.Score_String(rs.seq[r].c_str(), rs.seq[r].length(), P == 1 ? 0 : f % P));
        return 0;
    }
    if (cmd == "segs") {
        vector<Seg> segs;
        load_segs(argv[4], segs);
        double gc = atof(argv[5]);
        if (gc < 0) gc = gc_fraction(argv[3]);
        ICM_t indep(3, 2, 3);
        build_indep(indep, gc, NULL);
        vector<double> sc;
        for (size_t i = 0; i < segs.size(); i++) {
            string b = seg_buffer(rs.seq[segs[i].read], segs[i]);
            gene.Cumulative_Score(b, sc, 1);     // glimmer3.cc:1346
            put(&sc[0], 8 * sc.size());
            indep.Cumulative_Score(b, sc, 1);    // glimmer3.cc:1347
            put(&sc[0], 8 * sc.size());
        }
        return 0;
    }
    if (cmd == "allframe") {
        vector<Seg> segs;
        load_segs(argv[4], segs);
        for (size_t i = 0; i < segs.size(); i++) {
            string b = seg_buffer(rs.seq[segs[i].read], segs[i]);
            int len = b.length();
            string rc(len, 'a');
            for (int j = 0, k = len - 1; k >= 0; j++, k--) rc[j] = Complement(b[k]);
            // glimmer3.cc:346-354, before Permute_By_Frame
            put_d(gene.Score_String(b.c_str(), len, 1));
            put_d(gene.Score_String(b.c_str(), len, 2));
            put_d(gene.Score_String(b.c_str(), len, 0));
            put_d(gene.Score_String(rc.c_str(), len, 1));
            put_d(gene.Score_String(rc.c_str(), len, 0));
            put_d(gene.Score_String(rc.c_str(), len, 2));
        }
        return 0;
    }
    if (cmd == "partial") {
        int count = atoi(argv[4]);
        for (int r = 0; r < count && r < (int)rs.seq.size(); r++) {
            const string &S = rs.seq[r];
            int lim = (int)S.length() < W - 1 ? (int)S.length() : W - 1;
            for (int f = 0; f < P; f++)
                for (int i = 0; i < lim; i++)
                    put_d(gene.Partial_Window_Prob(i, S.c_str(), f));
        }
        return 0;
    }
    fprintf(stderr, "unknown command %s\n", cmd.c_str());
    return 2;
}
