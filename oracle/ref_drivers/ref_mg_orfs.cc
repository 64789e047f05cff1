// ref_mg_orfs.cc -- golden vectors for glimmer-mg's front half (Find_Orfs, src/Glimmer/glimmer_base.cc:638-779;
// Score_Orfs_Errors / Score_Orf_Starts, src/Glimmer/glimmer-mg.cc:1605-1861) and the integration demo of
// gmg_mg_score_reads.  Test infrastructure only; built by oracle/Makefile into oracle/_ref/.
//
// This file is ours.  It pulls the reference's glimmer-mg.cc translation unit in WHOLE (from /root/reference,
// via the include path; nothing is copied) with its main() renamed, so that the file-static functions
// Parse_Command_Line / Score_Orfs_Errors / Trace_Back are callable, and it intercepts Add_Events_Fwd /
// Add_Events_Rev with the linker (--wrap) to see the start lists Score_Orfs_Errors hands over.
// Only the user-ICM mode (-m <icm>, no -c classifications) is driven; the error branch (-i indels, -s substitutions,
// -q quality file) is.
//
//   ref_mg_orfs dump  <glimmer-mg options...> <fasta> <tag>     text dump on stdout:
//        R <read index> <n_orfs>
//        O <frame> <stop_position> <gene_len> <orf_len>          every ORF Find_Orfs produced, in order
//        G <orf index> <n_starts>                                every ORF Score_Orfs_Errors accepted
//        S <j> <pos> <score %a> <which> <truncated> <first>      its start list as handed to Add_Events_* (sorted)
//   with -i / -s the S lines come in the order Score_Orf_Starts PUSHED them (the list as it was right before
//   Score_Orfs_Errors' sort, seen through a hook on that sort call) and carry the Error_t list:
//        S <j> <pos> <score %a> <which> <truncated> <first> <n_errors> {<pos> <type>}...
//   ref_mg_orfs batch <glimmer-mg options...> <fasta> <tag>     (built with -DGMG_BATCH, links libgmg.so)
//        same pipeline as glimmer-mg's main, but Find_Orfs + Score_Orfs_Errors of ALL reads are replaced by
//        ONE gmg_mg_score_reads call; writes <tag>.predict, which must equal the reference's byte for byte.

#include "glimmer-mg.hh"

// The reference sorts each start list with an unqualified sort(..., Start_Cmp) (glimmer-mg.cc:1659); seeing the list
// right before that call gives the push order of Score_Orf_Starts without touching the reference.
static vector<Start_t> Presort_List;
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

#include <map>

#ifdef GMG_BATCH
#include "gmg.h"
#endif

void real_Add_Events_Fwd(const Orf_t &, vector<Start_t> &, int &)
    asm("__real__Z14Add_Events_FwdRK5Orf_tRSt6vectorI7Start_tSaIS3_EERi");
void real_Add_Events_Rev(const Orf_t &, vector<Start_t> &, int &)
    asm("__real__Z14Add_Events_RevRK5Orf_tRSt6vectorI7Start_tSaIS3_EERi");
void wrap_Add_Events_Fwd(const Orf_t &, vector<Start_t> &, int &)
    asm("__wrap__Z14Add_Events_FwdRK5Orf_tRSt6vectorI7Start_tSaIS3_EERi");
void wrap_Add_Events_Rev(const Orf_t &, vector<Start_t> &, int &)
    asm("__wrap__Z14Add_Events_RevRK5Orf_tRSt6vectorI7Start_tSaIS3_EERi");

static bool Capture = false;
static vector<pair<Orf_t, vector<Start_t> > > Captured;
static bool Error_Mode = false;                         // -i / -s: capture the push order instead of the sorted list

void wrap_Add_Events_Fwd(const Orf_t &orf, vector<Start_t> &sl, int &id)
{
    if (Capture) Captured.push_back(make_pair(orf, Error_Mode ? Presort_List : sl));
    real_Add_Events_Fwd(orf, sl, id);
}
void wrap_Add_Events_Rev(const Orf_t &orf, vector<Start_t> &sl, int &id)
{
    if (Capture) Captured.push_back(make_pair(orf, Error_Mode ? Presort_List : sl));
    real_Add_Events_Rev(orf, sl, id);
}

// ---- the set-up steps of glimmer-mg's main for -m <icm> (glimmer-mg.cc:241-316), in the same order ----
static void setup_options(int argc, char **argv)
{
    Verbose = 0;
    Parse_Command_Line(argc, argv);
    Set_Start_And_Stop_Codons();
    if (Feature_File != NULL) Parse_Features(Feature_File);
    if (!User_ICM || !classifications.empty() || Detail_Log) {
        fprintf(stderr, "ref_mg_orfs: only -m <icm> without -c / detail log is driven here\n");
        exit(2);
    }
    Error_Mode = Allow_Indels || Allow_Subs;
}

// quality values of every read, as glimmer-mg's main reads them (glimmer-mg.cc:339-341)
static void read_qualities(vector<vector<int> > &qual_list, size_t n_seq)
{
    qual_list.assign(n_seq, vector<int>());
    if (Quality_File_Name == NULL) return;
    FILE *fp = File_Open(Quality_File_Name, "r", __FILE__, __LINE__);
    string header;
    for (size_t i = 0; i < n_seq; i++) Fasta_Qual_Vec_Read(fp, qual_list[i], header);
    fclose(fp);
}

// glimmer-mg.cc:384-392
static void load_quality(vector<vector<int> > &qual_list, int i)
{
    if (!Allow_Indels) return;
    Quality_Values = qual_list[i];
    if (Quality_File_Name == NULL) Set_Quality_454();
    else Clean_Quality_454();
}

static void setup_models(void)
{
    if (!GC_Frac_Set) Set_GC_Fraction();
    Indep_Model.Build_Indep_WO_Stops(Indep_GC_Frac, Stop_Codon);
    Set_Ignore_Score_Len();
    if (User_RBS) {
        LogOdds_PWM = Ribosome_PWM;
        LogOdds_PWM.Make_Log_Odds_WRT_GC(Indep_GC_Frac);
    }
    Gene_ICM.Read(ICM_File_Name);
}

static void load_sequence(const vector<string> &seq_list, const vector<string> &hdr_list, int i)
{
    Fasta_Header = hdr_list[i].c_str();
    Sequence = seq_list[i];
    Sequence_Len = Sequence.length();
    for (int k = 0; k < Sequence_Len; k++) Sequence[k] = tolower(Filter(Sequence[k]));   // glimmer-mg.cc:381-382
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: ref_mg_orfs dump|batch <glimmer-mg args>\n"); return 2; }
    string mode = argv[1];
    try {
        setup_options(argc - 1, argv + 1);
        vector<string> seq_list, hdr_list;
        vector<Orf_t> orf_list;

        if (mode == "dump") {
            setup_models();
            {
                FILE *fp = File_Open(Sequence_File_Name, "r", __FILE__, __LINE__);
                string s, h;
                while (Fasta_Read(fp, s, h)) { seq_list.push_back(s); hdr_list.push_back(h); }
                fclose(fp);
            }
            const int n_seq = seq_list.size();
            vector<vector<int> > qual_list;
            read_qualities(qual_list, n_seq);
            for (int i = 0; i < n_seq; i++) {
                load_sequence(seq_list, hdr_list, i);
                load_quality(qual_list, i);
                Initialize_Terminal_Events(First_Event, Final_Event, Best_Event, Last_Event);
                Find_Orfs(orf_list);
                printf("R %d %d\n", i, (int)orf_list.size());
                map<pair<int, int>, int> index_of;      // (frame, stop_position) -> position in orf_list
                for (size_t o = 0; o < orf_list.size(); o++) {
                    printf("O %d %d %d %d\n", orf_list[o].Get_Frame(), orf_list[o].Get_Stop_Position(),
                           orf_list[o].Get_Gene_Len(), orf_list[o].Get_Orf_Len());
                    index_of[make_pair(orf_list[o].Get_Frame(), orf_list[o].Get_Stop_Position())] = o;
                }
                Capture = true;
                Captured.clear();
                Score_Orfs_Errors(orf_list, NULL);
                Capture = false;
                for (size_t c = 0; c < Captured.size(); c++) {
                    const Orf_t &orf = Captured[c].first;
                    const vector<Start_t> &sl = Captured[c].second;
                    printf("G %d %d\n", index_of[make_pair(orf.Get_Frame(), orf.Get_Stop_Position())], (int)sl.size());
                    for (size_t s = 0; s < sl.size(); s++) {
                        printf("S %d %d %a %d %d %d", sl[s].j, sl[s].pos, sl[s].score, (int)sl[s].which,
                               (int)sl[s].truncated, (int)sl[s].first);
                        if (Error_Mode) {
                            printf(" %d", (int)sl[s].errors.size());
                            for (size_t e = 0; e < sl[s].errors.size(); e++) printf(" %d %d", sl[s].errors[e].pos, sl[s].errors[e].type);
                        }
                        printf("\n");
                    }
                }
                orf_list.clear();
                Clear_Events();
            }
            return 0;
        }
#ifdef GMG_BATCH
        if (mode == "batch") {
            const char *dev = getenv("GMG_DEVICE");
            if (gmg_init(dev ? atoi(dev) : 0) != GMG_OK) { fprintf(stderr, "%s\n", gmg_last_error()); return 1; }
            // the file's bytes go to the device as they are: Fasta_Read, tolower (Filter ()) and the 2-bit packing happen
            // there (gmg_fasta_ingest); the host keeps the header extents and gets the filtered bases back for the
            // event / DP code, which reads the global Sequence
            string file_bytes;
            {
                FILE *fp = File_Open(Sequence_File_Name, "rb", __FILE__, __LINE__);
                char buf[1 << 16];
                size_t got;
                while ((got = fread(buf, 1, sizeof buf, fp)) > 0) file_bytes.append(buf, got);
                fclose(fp);
            }
            gmg_reads *reads = NULL;
            gmg_fasta *fasta = NULL;
            if (gmg_fasta_ingest(file_bytes.data(), file_bytes.size(), &reads, &fasta) != GMG_OK) { fprintf(stderr, "%s\n", gmg_last_error()); return 1; }
            uint64_t n_ing = 0, total_bases = 0, gc_ct = 0;
            gmg_fasta_info(fasta, &n_ing, &total_bases, &gc_ct);
            const int n_seq = (int)n_ing;
            if (!GC_Frac_Set) {                         // Set_GC_Fraction (glimmer_base.cc:2564-2595) without reading the file again
                Indep_GC_Frac = double(gc_ct) / total_bases;
                GC_Frac_Set = true;
            }
            setup_models();
            seq_list.resize(n_seq);
            hdr_list.resize(n_seq);
            vector<uint64_t> hb(n_ing), he(n_ing), off(n_ing + 1);
            gmg_fasta_headers(fasta, hb.data(), he.data());
            vector<uint32_t> packed(gmg_packed_words(total_bases) + 1, 0);
            gmg_reads_download(reads, packed.data(), off.data());
            gmg_fasta_free(fasta);
            for (int i = 0; i < n_seq; i++) {            // replace what the host parser produced by what came back from the device
                hdr_list[i] = file_bytes.substr(hb[i], he[i] - hb[i]);
                string &sq = seq_list[i];
                sq.resize(off[i + 1] - off[i]);
                for (uint64_t k = 0; k < sq.size(); k++) { const uint64_t g = off[i] + k; sq[k] = "acgt"[(packed[g >> 4] >> (2 * (g & 15))) & 3]; }
            }
            // ONE call: Score_All_Frames + Find_Orfs + Score_Orf_Starts + the filter of Score_Orfs_Errors, all reads
            gmg_mg_params prm;
            memset(&prm, 0, sizeof prm);
            prm.min_gene_len = Min_Gene_Len;
            prm.allow_truncated = Allow_Truncated_Orfs;
            prm.ignore_score_len = Ignore_Score_Len;
            prm.start_threshold = Start_Threshold;
            prm.flags = GMG_MG_ACCEPTED_ONLY;           // only what Add_Events_* will see comes back
            vector<uint8_t> qual_all;
            if (Error_Mode) {                           // -i / -s: Score_Indels / the substitution branch run on the device too
                prm.flags |= Allow_Indels ? GMG_MG_ALLOW_INDELS : GMG_MG_ALLOW_SUBS;
                prm.min_indel_orf_len = Min_Indel_ORF_Len;
                prm.indel_quality_threshold = Indel_Quality_Threshold;
                prm.indel_max = Indel_Max;
                prm.indel_suffix_score_threshold = Indel_Suffix_Score_Threshold;
                if (Allow_Indels && Quality_File_Name != NULL) {       // the user's Phred values, one byte per base
                    vector<vector<int> > qual_list;
                    read_qualities(qual_list, n_seq);
                    qual_all.reserve(total_bases);
                    for (int i = 0; i < n_seq; i++) {
                        if (qual_list[i].size() != seq_list[i].size()) {   // Clean_Quality_454's check (glimmer-mg.cc:534-537)
                            fprintf(stderr, "ERROR:  %s sequence length does not match quality values length\n", hdr_list[i].c_str());
                            return 1;
                        }
                        for (size_t k = 0; k < qual_list[i].size(); k++) qual_all.push_back(qual_list[i][k] > 255 ? 255 : qual_list[i][k] < 0 ? 0 : qual_list[i][k]);
                    }
                    prm.quality = qual_all.data();
                }
            }
            prm.n_start_codons = Start_Codon.size();
            prm.n_stop_codons = Stop_Codon.size();
            for (size_t s = 0; s < Start_Codon.size() && s < 8; s++) memcpy(prm.start_codon[s], Start_Codon[s], 3);
            for (size_t s = 0; s < Stop_Codon.size() && s < 8; s++) memcpy(prm.stop_codon[s], Stop_Codon[s], 3);
            gmg_mg_result *res = NULL;
            if (gmg_mg_score_reads(Gene_ICM.Device_Model(), Indep_Model.Device_Model(), reads, &prm, NULL, &res, NULL) != GMG_OK) {
                fprintf(stderr, "%s\n", gmg_last_error());
                return 1;
            }
            uint64_t n_orfs = 0, n_starts = 0;
            gmg_mg_result_info(res, &n_orfs, &n_starts);
            vector<gmg_mg_orf> orfs(n_orfs ? n_orfs : 1);
            vector<gmg_start> starts(n_starts ? n_starts : 1);
            vector<uint64_t> read_orf_off(n_seq + 1);
            vector<gmg_start_errors> errs(Error_Mode ? (n_starts ? n_starts : 1) : 0);
            if (gmg_mg_result_fetch(res, orfs.data(), starts.data(), read_orf_off.data()) != GMG_OK) { fprintf(stderr, "%s\n", gmg_last_error()); return 1; }
            if (Error_Mode && gmg_mg_result_fetch_errors(res, errs.data()) != GMG_OK) { fprintf(stderr, "%s\n", gmg_last_error()); return 1; }
            gmg_mg_result_free(res);
            gmg_reads_free(reads);
            // events, DP and trace-back per read: host, unchanged reference code (glimmer-mg.cc:400-441, 1620-1685)
            string filename = Output_Tag;
            filename.append(".predict");
            FILE *predict_fp = File_Open(filename, "w", __FILE__, __LINE__);
            for (int i = 0; i < n_seq; i++) {
                load_sequence(seq_list, hdr_list, i);
                fprintf(predict_fp, ">%s\n", Fasta_Header);
                Initialize_Terminal_Events(First_Event, Final_Event, Best_Event, Last_Event);
                Meta_PWM_Save.resize(2 * Sequence_Len);                    // glimmer-mg.cc:1622-1627
                for (unsigned int si = 0; si < 2 * Sequence_Len; si++) Meta_PWM_Save[si] = pair<double, int>(0.0, 999);
                int id = 0;
                for (uint64_t o = read_orf_off[i]; o < read_orf_off[i + 1]; o++) {
                    const gmg_mg_orf &g = orfs[o];
                    if (!g.accepted) continue;
                    Orf_t orf;
                    orf.Set_Stop_Position(g.stop_position);
                    orf.Set_Frame(g.frame);
                    orf.Set_Gene_Len(g.gene_len);
                    orf.Set_Orf_Len(g.orf_len);
                    vector<Start_t> sl(g.n_starts);
                    for (uint32_t s = 0; s < g.n_starts; s++) {
                        const gmg_start &t = starts[g.start_begin + s];
                        sl[s].j = t.j; sl[s].pos = t.pos; sl[s].score = t.score; sl[s].rate = 0.0; sl[s].which = t.which;
                        sl[s].truncated = t.truncated; sl[s].first = t.first;
                        if (Error_Mode) {
                            const gmg_start_errors &e = errs[g.start_begin + s];
                            for (int k = 0; k < e.n; k++) sl[s].errors.push_back(Error_t(e.pos[k], e.type[k]));
                        }
                    }
                    std::sort(sl.begin(), sl.end(), Start_Cmp);            // glimmer-mg.cc:1659: same algorithm on the same push order
                    if (g.accepted == 2) {                                 // ties on pos: first_j is the sort's to decide (:1661-1666)
                        const int first_j = g.frame > 0 ? sl.front().j : sl.back().j;
                        if (first_j + 1 < Min_Gene_Len) continue;
                    }
                    if (g.frame > 0) real_Add_Events_Fwd(orf, sl, id);
                    else real_Add_Events_Rev(orf, sl, id);
                }
                Process_Events();
                Set_Final_Event(Final_Event, Best_Event, Sequence_Len);
                Trace_Back(predict_fp, Final_Event);
                Clear_Events();
            }
            fclose(predict_fp);
            return 0;
        }
#endif
        fprintf(stderr, "unknown mode %s\n", mode.c_str());
        return 2;
    } catch (std::exception &e) {
        cerr << "** Standard Exception **" << endl << e << endl;
        return 1;
    }
}
