// ref_mg_classes.cc -- golden vectors for glimmer-mg's classification mode (-c): the REAL reference main loop
// (src/Glimmer/glimmer-mg.cc:219-470) with Parse_Classes :726-758, Read_Meta_ICMs :998-1027, Classes_ICM_File :473-515,
// Read_Meta_GC :1389-1420, Read_Meta_Stops :1211-1250, Update_Meta_Stop :2185-2219 and Update_Meta_Null_ICM :2050-2068.
// Test infrastructure only; built by oracle/Makefile into oracle/_ref/.
//
// This file is ours.  It pulls the reference's glimmer-mg.cc translation unit in WHOLE (from /root/reference, via the
// include path; nothing is copied) with its main() renamed, sets the one thing the reference's installer patches into the
// source with sed (install_glimmer.py:121: the file-static `ICM_dir`, glimmer-mg.cc:147) from the environment variable
// GMG_REF_ICM_DIR, and then calls the reference's own main() with the command line it was given: option parsing, the
// chunked ICM-grouped loop, every Update_Meta_*, Find_Orfs, Score_Orfs_Errors, events, trace-back and <tag>.predict are
// the reference's code running unchanged.  (GMG_REF_CHUNK, optional, sets the other file-static constant of that loop,
// Chunk_Sequences = 500000 reads per chunk, glimmer-mg.cc:128, so that a 999-read file can span several chunks.)  Find_Orfs and Add_Events_Fwd / Add_Events_Rev are intercepted with the linker
// (--wrap) to see, read by read IN THE ORDER THE REFERENCE PROCESSES THEM, what the scoring path consumed and produced:
//
//   ref_mg_classes <glimmer-mg options incl. -c class.txt ...> <fasta> <tag>        text dump on stdout:
//        R <header prefix> <n_orfs> <Indep_GC_Frac %a> <Ignore_Score_Len> <Genbank_Xlate_Code> <n_stops> <stop>... <icm file>
//        O <frame> <stop_position> <gene_len> <orf_len>          every ORF Find_Orfs produced, in order
//        G <frame> <stop_position> <n_starts>                    every ORF Score_Orfs_Errors handed to Add_Events_*
//        S <j> <pos> <score %a> <which> <truncated> <first> <n_errors> {<pos> <type>}...
//   the S lines come in the order Score_Orf_Starts PUSHED them (the list as it was right before Score_Orfs_Errors' sort,
//   seen through a hook on that unqualified sort call, as in ref_mg_orfs.cc).

#include "glimmer-mg.hh"

static vector<Start_t> Presort_List;
static bool Quiet = false;      // GMG_REF_QUIET=1: no dump (and none of its per-read look-ups) -- the reference as it runs, for timing
template <class It> inline void gmg_hooked_sort(It a, It b) { std::sort(a, b); }
template <class It, class Cmp> inline void gmg_hooked_sort(It a, It b, Cmp c) { std::sort(a, b, c); }
inline void gmg_hooked_sort(vector<Start_t>::iterator a, vector<Start_t>::iterator b, bool (*c)(const Start_t &, const Start_t &))
{
    Presort_List.assign(a, b);
    std::sort(a, b, c);
}
#define sort(...) gmg_hooked_sort(__VA_ARGS__)
#define main glimmer_mg_reference_main
#include "glimmer-mg.cc"
#undef main
#undef sort

void real_Add_Events_Fwd(const Orf_t &, vector<Start_t> &, int &)
    asm("__real__Z14Add_Events_FwdRK5Orf_tRSt6vectorI7Start_tSaIS3_EERi");
void real_Add_Events_Rev(const Orf_t &, vector<Start_t> &, int &)
    asm("__real__Z14Add_Events_RevRK5Orf_tRSt6vectorI7Start_tSaIS3_EERi");
void wrap_Add_Events_Fwd(const Orf_t &, vector<Start_t> &, int &)
    asm("__wrap__Z14Add_Events_FwdRK5Orf_tRSt6vectorI7Start_tSaIS3_EERi");
void wrap_Add_Events_Rev(const Orf_t &, vector<Start_t> &, int &)
    asm("__wrap__Z14Add_Events_RevRK5Orf_tRSt6vectorI7Start_tSaIS3_EERi");
void real_Find_Orfs(vector<Orf_t> &) asm("__real__Z9Find_OrfsRSt6vectorI5Orf_tSaIS0_EE");
void wrap_Find_Orfs(vector<Orf_t> &) asm("__wrap__Z9Find_OrfsRSt6vectorI5Orf_tSaIS0_EE");

static void dump_list(const Orf_t &orf)
{
    const vector<Start_t> &sl = Presort_List;
    printf("G %d %d %d\n", orf.Get_Frame(), orf.Get_Stop_Position(), (int)sl.size());
    for (size_t s = 0; s < sl.size(); s++) {
        printf("S %d %d %a %d %d %d %d", sl[s].j, sl[s].pos, sl[s].score, (int)sl[s].which, (int)sl[s].truncated,
               (int)sl[s].first, (int)sl[s].errors.size());
        for (size_t e = 0; e < sl[s].errors.size(); e++) printf(" %d %d", sl[s].errors[e].pos, sl[s].errors[e].type);
        printf("\n");
    }
}

void wrap_Add_Events_Fwd(const Orf_t &orf, vector<Start_t> &sl, int &id)
{
    if (!Quiet) dump_list(orf);
    real_Add_Events_Fwd(orf, sl, id);
}
void wrap_Add_Events_Rev(const Orf_t &orf, vector<Start_t> &sl, int &id)
{
    if (!Quiet) dump_list(orf);
    real_Add_Events_Rev(orf, sl, id);
}

// Find_Orfs is called once per processed read, right after the Update_Meta_* calls (glimmer-mg.cc:398-425): the globals
// hold what the scoring of THIS read will use
void wrap_Find_Orfs(vector<Orf_t> &orf_list)
{
    real_Find_Orfs(orf_list);
    if (Quiet) return;
    // which ICM is loaded: the group loop's iterator is local to main, but a read belongs to exactly one group
    // (Read_Meta_ICMs puts every classified header into one vector), so look it up
    const string prefix = split(string(Fasta_Header))[0];
    string icm = "?";
    for (icm_reads_hash::const_iterator it = ICM_Sequences.begin(); it != ICM_Sequences.end(); it++) {
        bool mine = false;
        for (size_t r = 0; r < it->second.size() && !mine; r++) mine = it->second[r] == prefix;
        if (mine) { icm = it->first; break; }
    }
    printf("R %s %d %a %d %d %d", prefix.c_str(), (int)orf_list.size(), Indep_GC_Frac, Ignore_Score_Len, Genbank_Xlate_Code,
           (int)Stop_Codon.size());
    for (size_t s = 0; s < Stop_Codon.size(); s++) printf(" %s", Stop_Codon[s]);
    printf(" %s\n", icm.c_str());
    for (size_t o = 0; o < orf_list.size(); o++)
        printf("O %d %d %d %d\n", orf_list[o].Get_Frame(), orf_list[o].Get_Stop_Position(), orf_list[o].Get_Gene_Len(),
               orf_list[o].Get_Orf_Len());
}

int main(int argc, char **argv)
{
    const char *dir = getenv("GMG_REF_ICM_DIR");
    if (dir == NULL) { fprintf(stderr, "ref_mg_classes: set GMG_REF_ICM_DIR (the .genomeData directory)\n"); return 2; }
    Quiet = getenv("GMG_REF_QUIET") != NULL;
    ICM_dir = dir;                                      // what install_glimmer.py:121 writes into glimmer-mg.cc:147
    if (const char *c = getenv("GMG_REF_CHUNK")) Chunk_Sequences = atoi(c);   // glimmer-mg.cc:128 (500000): small chunks for the tests
    const int rc = glimmer_mg_reference_main(argc, argv);
    fflush(stdout);
    return rc;
}
