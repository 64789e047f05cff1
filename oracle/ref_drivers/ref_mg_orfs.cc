// ref_mg_orfs.cc -- golden vectors for glimmer-mg's front half (Find_Orfs, src/Glimmer/glimmer_base.cc:638-779;
// Score_Orfs_Errors / Score_Orf_Starts, src/Glimmer/glimmer-mg.cc:1605-1861).  Test infrastructure only; built by oracle/Makefile into oracle/_ref/.
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
// (The drop-in driver that runs these loops on the GPU is product code: integration/glimmer-mg_gpu.cc.)

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
    if (argc < 4) { fprintf(stderr, "usage: ref_mg_orfs dump <glimmer-mg args>\n"); return 2; }
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
        fprintf(stderr, "unknown mode %s\n", mode.c_str());
        return 2;
    } catch (std::exception &e) {
        cerr << "** Standard Exception **" << endl << e << endl;
        return 1;
    }
}
