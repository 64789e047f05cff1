// gmg_classes.cc -- glimmer-mg's classification mode (-c): which ICM scores which read, in which order, against which
// null model and with which stop codons (host only; the scoring itself is gmg_mg_score_reads, one call per group).
//
// With -c every read comes with the Phymm classes of a class file; the reference then
//   * names one gene-ICM FILE per read from its classes (Classes_ICM_File, src/Glimmer/glimmer-mg.cc:473-515) and collects
//     the reads of every file (Read_Meta_ICMs, :998-1027),
//   * reads the input in chunks of Chunk_Sequences reads and, per chunk, visits the ICM files in the iteration order of a
//     __gnu_cxx::hash_map and inside a file the reads in the order Read_Meta_ICMs met them -- itself the iteration order of
//     the hash_map of classifications (:334-375).  <tag>.predict is written in that order, so the order is part of the
//     byte-identical contract;
//   * rebuilds, for EVERY read, the stop codons from the translation table of its first class (Update_Meta_Stop,
//     :2185-2219; Read_Meta_Stops :1211-1250) and the null model from the mean GC of its classes (Update_Meta_Null_ICM,
//     :2050-2068; Read_Meta_GC :1389-1420), then Ignore_Score_Len (Set_Ignore_Score_Len, glimmer_base.cc:2597-2633).
// The order is whatever libstdc++'s SGI hash table makes of the insertions (string hash h = 5 h + c, prime bucket counts,
// insertion at the head of a bucket, rehashing that reverses chains): it is reproduced here by USING that container --
// <ext/hash_map> is part of the toolchain the reference itself needs -- not by imitating it.

#include "icm.hh"
#include "../../include/gmg.h"

#define _GLIBCXX_PERMIT_BACKWARD_HASH 1
#include <ext/hash_map>

#include <fstream>
#include <math.h>
#include <memory>
#include <new>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <vector>

extern int gmg_set_error(int code, const char *fmt, ...);

namespace {

// glimmer-mg.cc:150-160: strings hash through their c_str()
struct StrHash {
    size_t operator()(const std::string &x) const { return __gnu_cxx::hash<const char *>()(x.c_str()); }
};
typedef __gnu_cxx::hash_map<std::string, std::vector<std::string>, StrHash> ListMap;

// split on white space (src/Common/kelley.cc:34-53)
static std::vector<std::string> split_ws(const char *s, size_t n)
{
    std::vector<std::string> out;
    size_t i = 0;
    while (i < n) {
        while (i < n && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) i++;
        size_t b = i;
        while (i < n && !(s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) i++;
        if (i > b) out.push_back(std::string(s + b, i - b));
    }
    return out;
}

// "<strain>|<NC>" (kelley.cc:10-26 with '|'): false when there is no second field
static bool strain_nc(const std::string &cls, std::string &strain, std::string &nc)
{
    const size_t bar = cls.find('|');
    if (bar == std::string::npos) return false;
    strain = cls.substr(0, bar);
    const size_t bar2 = cls.find('|', bar + 1);
    nc = cls.substr(bar + 1, bar2 == std::string::npos ? std::string::npos : bar2 - bar - 1);
    return true;
}

}   // namespace

struct gmg_classes {
    std::string icm_dir;
    ListMap classifications;                             // header prefix -> classes           (glimmer-mg.cc:163-164)
    ListMap icm_sequences;                               // ICM file -> header prefixes        (:205-206)
    __gnu_cxx::hash_map<std::string, float, StrHash> gc; // class -> GC of its genome          (:196-197)
    __gnu_cxx::hash_map<std::string, int, StrHash> transl;   // class -> GenBank transl_table  (:199-200)
    std::vector<std::string> icm_files;                  // icm_sequences in iteration order
    std::vector<const std::vector<std::string> *> icm_reads;
    uint64_t n_missing_gc;
};

// Classes_ICM_File (glimmer-mg.cc:473-515)
static int classes_icm_file(const gmg_classes *c, const std::vector<std::string> &cl, std::string &out)
{
    std::string s1, n1, s2, n2;
    if (cl.size() >= 2) {
        for (size_t i = 1; i < cl.size(); i++) {
            const bool first_smaller = cl[0].compare(cl[i]) < 0;
            const std::string &a = first_smaller ? cl[0] : cl[i], &b = first_smaller ? cl[i] : cl[0];
            if (!strain_nc(a, s1, n1) || !strain_nc(b, s2, n2)) return -1;
            out = c->icm_dir + "/" + s1 + "/" + n1 + "_2/" + s2 + "/" + n2 + ".gicm";
            struct stat st;
            if (stat(out.c_str(), &st) == 0) return 0;   // the best existing double
        }
    }
    if (!strain_nc(cl[0], s1, n1)) return -1;
    out = c->icm_dir + "/" + s1 + "/" + n1 + ".gicm";
    return 0;
}

// (no C++ exception leaves the C ABI: the entry points below turn them into GMG_ENOMEM / GMG_EINVAL)
#define GMG_C_ABI_GUARD(name, call)                                                                              \
    try { return call; }                                                                                         \
    catch (const std::bad_alloc &) { return gmg_set_error(GMG_ENOMEM, name ": out of host memory"); }            \
    catch (const std::exception &e) { return gmg_set_error(GMG_EINVAL, name ": %s", e.what()); }

static int classes_load(const char *class_text, uint64_t n_bytes, const char *icm_dir, gmg_classes **out)
{
    if ((!class_text && n_bytes) || !icm_dir || !out) return gmg_set_error(GMG_EINVAL, "gmg_classes_load: NULL argument");
    std::unique_ptr<gmg_classes> holder(new gmg_classes);
    gmg_classes *c = holder.get();
    c->icm_dir = icm_dir;
    c->n_missing_gc = 0;
    // Parse_Classes (glimmer-mg.cc:726-758): one line per read, "<header prefix> <class> <class> ..."; a later line for the
    // same read replaces the earlier one
    uint64_t pos = 0, line_no = 0;
    while (pos < n_bytes) {
        const char *nl = (const char *)memchr(class_text + pos, '\n', n_bytes - pos);
        const uint64_t end = nl ? (uint64_t)(nl - class_text) : n_bytes;
        line_no++;
        std::vector<std::string> a = split_ws(class_text + pos, end - pos);
        pos = end + 1;
        // the reference indexes a[0] and, later, the first class of every read without a check (undefined behaviour on a blank
        // line or a read without classes): refused here
        if (a.size() < 2) {
            return gmg_set_error(GMG_EINVAL, "gmg_classes_load: line %llu of the classification file names no %s",
                                 (unsigned long long)line_no, a.empty() ? "read" : "class");
        }
        std::vector<std::string> v(a.begin() + 1, a.end());
        c->classifications[a[0]] = v;
    }
    // Read_Meta_ICMs (:998-1027), Read_Meta_GC (:1389-1420), Read_Meta_Stops (:1211-1250): all three walk the classifications
    // in the table's own order
    for (ListMap::const_iterator ci = c->classifications.begin(); ci != c->classifications.end(); ++ci) {
        const std::vector<std::string> &cl = ci->second;
        std::string icm_file, strain, nc;
        if (classes_icm_file(c, cl, icm_file) != 0) {
            const std::string who = ci->first;
            return gmg_set_error(GMG_EINVAL, "gmg_classes_load: a class of read %s is not of the form <strain>|<NC>", who.c_str());
        }
        ListMap::iterator isi = c->icm_sequences.find(icm_file);
        if (isi == c->icm_sequences.end()) {
            std::vector<std::string> seqs(1, ci->first);
            c->icm_sequences[icm_file] = seqs;
        } else
            isi->second.push_back(ci->first);
        for (size_t i = 0; i < cl.size(); i++) {
            if (c->gc.find(cl[i]) != c->gc.end()) continue;
            if (!strain_nc(cl[i], strain, nc)) {
                const std::string who = ci->first;
                    return gmg_set_error(GMG_EINVAL, "gmg_classes_load: a class of read %s is not of the form <strain>|<NC>", who.c_str());
            }
            const std::string gc_file = c->icm_dir + "/" + strain + "/" + nc + ".gc.txt";
            std::ifstream gc_open(gc_file.c_str());
            if (gc_open.good()) {
                std::string line;
                std::getline(gc_open, line);
                c->gc[cl[i]] = strtod(line.c_str(), NULL);   // double -> float, as Sequence_GC stores it
            } else {
                c->n_missing_gc++;                      // the reference warns on stderr and goes on with 0.5
                c->gc[cl[i]] = 0.5;
            }
        }
        if (c->transl.find(cl[0]) == c->transl.end()) {
            strain_nc(cl[0], strain, nc);
            const std::string gbk_file = c->icm_dir + "/" + strain + "/" + nc + ".gbk";
            std::ifstream gbk_in(gbk_file.c_str());
            std::string line;
            int code = 11;                              // no file, or no transl_table in it: the bacterial code
            while (std::getline(gbk_in, line)) {
                const size_t tt = line.find("transl_table=");
                if (tt != std::string::npos) { code = (int)strtol(line.substr(tt + 13).c_str(), NULL, 10); break; }
            }
            c->transl[cl[0]] = code;
        }
    }
    for (ListMap::const_iterator it = c->icm_sequences.begin(); it != c->icm_sequences.end(); ++it) {
        c->icm_files.push_back(it->first);
        c->icm_reads.push_back(&it->second);
    }
    *out = holder.release();
    return GMG_OK;
}
extern "C" int gmg_classes_load(const char *class_text, uint64_t n_bytes, const char *icm_dir, gmg_classes **out)
{
    GMG_C_ABI_GUARD("gmg_classes_load", classes_load(class_text, n_bytes, icm_dir, out))
}

extern "C" int gmg_classes_free(gmg_classes *c)
{
    delete c;
    return GMG_OK;
}

extern "C" int gmg_classes_info(const gmg_classes *c, uint64_t *n_reads, uint32_t *n_icms, uint32_t *n_classes, uint64_t *n_missing_gc)
{
    if (!c) return gmg_set_error(GMG_EINVAL, "gmg_classes_info: NULL handle");
    if (n_reads) *n_reads = c->classifications.size();
    if (n_icms) *n_icms = (uint32_t)c->icm_files.size();
    if (n_classes) *n_classes = (uint32_t)c->gc.size();
    if (n_missing_gc) *n_missing_gc = c->n_missing_gc;
    return GMG_OK;
}

extern "C" const char *gmg_classes_icm_file(const gmg_classes *c, uint32_t k)
{
    if (!c || k >= c->icm_files.size()) { gmg_set_error(GMG_EINVAL, "gmg_classes_icm_file: no such ICM"); return NULL; }
    return c->icm_files[k].c_str();
}

static int classes_plan(const gmg_classes *c, const char *const *hdr, const uint32_t *hdr_len, uint64_t n,
                        uint64_t *order, uint64_t *icm_begin, double *gc, int32_t *transl, uint64_t *n_order)
{
    if (!c || (!hdr && n) || (!hdr_len && n) || !order || !icm_begin || !n_order)
        return gmg_set_error(GMG_EINVAL, "gmg_classes_plan: NULL argument");
    // Read_Indexes (glimmer-mg.cc:328-352): prefix of the header line -> index in the chunk, the later read wins
    __gnu_cxx::hash_map<std::string, uint64_t, StrHash> read_index;
    for (uint64_t i = 0; i < n; i++) {
        uint32_t b = 0;
        const char *h = hdr[i];
        while (b < hdr_len[i] && (h[b] == ' ' || h[b] == '\t' || h[b] == '\n' || h[b] == '\r')) b++;
        uint32_t e = b;
        while (e < hdr_len[i] && !(h[e] == ' ' || h[e] == '\t' || h[e] == '\n' || h[e] == '\r')) e++;
        if (e == b) continue;                           // an empty header line: split (hdr)[0] does not exist in the reference
        read_index[std::string(h + b, e - b)] = i;
    }
    uint64_t k = 0;
    for (size_t f = 0; f < c->icm_files.size(); f++) {
        icm_begin[f] = k;
        const std::vector<std::string> &reads = *c->icm_reads[f];
        for (size_t r = 0; r < reads.size(); r++) {
            __gnu_cxx::hash_map<std::string, uint64_t, StrHash>::const_iterator it = read_index.find(reads[r]);
            if (it == read_index.end()) continue;
            if (k >= n) return gmg_set_error(GMG_EINVAL, "gmg_classes_plan: more processed reads than reads");   // cannot happen
            order[k] = it->second;
            const std::vector<std::string> &cl = c->classifications.find(reads[r])->second;
            if (gc) {                                   // Update_Meta_Null_ICM (:2058-2064): a double sum of floats, divided by a float
                const float num_classes = (float)cl.size();
                double g = 0.0;
                for (unsigned int s = 0; s < num_classes; s++) g += c->gc.find(cl[s])->second;
                g /= num_classes;
                gc[k] = g;
            }
            if (transl) transl[k] = c->transl.find(cl[0])->second;   // Update_Meta_Stop (:2196)
            k++;
        }
    }
    icm_begin[c->icm_files.size()] = k;
    *n_order = k;
    return GMG_OK;
}
extern "C" int gmg_classes_plan(const gmg_classes *c, const char *const *hdr, const uint32_t *hdr_len, uint64_t n,
                                uint64_t *order, uint64_t *icm_begin, double *gc, int32_t *transl, uint64_t *n_order)
{
    GMG_C_ABI_GUARD("gmg_classes_plan", classes_plan(c, hdr, hdr_len, n, order, icm_begin, gc, transl, n_order))
}

// Set_Stop_Codons_By_Code (src/Common/gene.cc:1560-1624)
extern "C" int gmg_stop_codons_by_code(int code, char stop_codon[8][4], int *n_stop_codons)
{
    if (!stop_codon || !n_stop_codons) return gmg_set_error(GMG_EINVAL, "gmg_stop_codons_by_code: NULL argument");
    const char *set = NULL;
    switch (code) {
    case 1: case 11: case 12: set = "taa,tag,tga"; break;
    case 2: set = "taa,tag,aga,agg"; break;
    case 3: case 4: case 5: case 9: case 10: case 13: case 21: set = "taa,tag"; break;
    case 6: set = "tga"; break;
    case 14: set = "tag"; break;
    case 15: case 16: set = "taa,tga"; break;
    case 22: set = "taa,tga,tca"; break;
    case 23: set = "taa,tag,tga,tta"; break;
    default:
        *n_stop_codons = 0;
        return gmg_set_error(GMG_EINVAL, "ERROR:  Unknown translation-table number = %d", code);
    }
    int n = 0;
    for (const char *p = set; *p; p += (p[3] == ',') ? 4 : 3, n++) {
        memcpy(stop_codon[n], p, 3);
        stop_codon[n][3] = 0;
    }
    *n_stop_codons = n;
    return GMG_OK;
}

// Set_Ignore_Score_Len (src/Glimmer/glimmer_base.cc:2597-2633): the longest ORF expected once at random in a million bases
extern "C" int gmg_ignore_score_len(double gc_frac, const char (*stop_codon)[4], int n_stop_codons, int32_t *out)
{
    if (!stop_codon || !out || n_stop_codons < 1) return gmg_set_error(GMG_EINVAL, "gmg_ignore_score_len: bad argument");
    double poisson_lambda = 0.0;
    for (int i = 0; i < n_stop_codons; i++) {
        double x = 1.0;
        for (int j = 0; j < 3; j++)
            if (stop_codon[i][j] == 'c' || stop_codon[i][j] == 'g') x *= gc_frac / 2.0;
            else x *= (1.0 - gc_frac) / 2.0;
        poisson_lambda += x;
    }
    if (poisson_lambda == 0.0) return gmg_set_error(GMG_EINVAL, "gmg_ignore_score_len: the stop codons have probability 0");
    *out = (int32_t)(long int)floor(3.0 * log(2.0 * 1000000 * poisson_lambda) / poisson_lambda);
    return GMG_OK;
}

// The null models of one ICM group: Indep_Model . Build_Indep_WO_Stops (gc, Stop_Codon) (glimmer-mg.cc:2066; icm.cc:65-216) for
// every GC value of the list, with the host ICM_t of this library (its libm, like the reference's), on a few host threads,
// and up to the device as one gmg_null_set.
static int null_set_build(const double *gc_frac, int n, const char (*stop_codon)[4], int n_stop_codons, gmg_null_set **out)
{
    if (!gc_frac || n < 1 || !stop_codon || n_stop_codons < 1 || n_stop_codons > 8 || !out)
        return gmg_set_error(GMG_EINVAL, "gmg_null_set_build: bad argument");
    std::vector<const char *> stops;
    for (int i = 0; i < n_stop_codons; i++) {
        if (strlen(stop_codon[i]) != 3) return gmg_set_error(GMG_EINVAL, "gmg_null_set_build: stop codon %d is not 3 letters", i);
        stops.push_back(stop_codon[i]);
    }
    std::vector<int16_t> mip((size_t)n * 63);
    std::vector<float> prob((size_t)n * 63 * 4);
    const int n_threads = n < 256 ? 1 : n < 4096 ? 4 : 16;
    std::vector<std::thread> pool;
    for (int t = 0; t < n_threads; t++)
        pool.push_back(std::thread([&, t]() {
            ICM_t m(3, 2, 3);
            std::vector<short> mp;
            std::vector<float> pp;
            for (int i = (int)((long long)n * t / n_threads); i < (int)((long long)n * (t + 1) / n_threads); i++) {
                m.Build_Indep_WO_Stops(gc_frac[i], stops);
                m.Export_Tables(mp, pp);
                memcpy(mip.data() + (size_t)i * 63, mp.data(), 63 * sizeof(int16_t));
                memcpy(prob.data() + (size_t)i * 252, pp.data(), 252 * sizeof(float));
            }
        }));
    for (size_t t = 0; t < pool.size(); t++) pool[t].join();
    return gmg_null_set_from_tables(mip.data(), prob.data(), n, out);
}
extern "C" int gmg_null_set_build(const double *gc_frac, int n, const char (*stop_codon)[4], int n_stop_codons, gmg_null_set **out)
{
    GMG_C_ABI_GUARD("gmg_null_set_build", null_set_build(gc_frac, n, stop_codon, n_stop_codons, out))
}
