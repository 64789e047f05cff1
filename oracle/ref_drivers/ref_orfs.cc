// ref_orfs.cc -- golden vectors for the Score_Orfs inner loop (src/Glimmer/glimmer3.cc:1275-1552).
// Test infrastructure only; built by oracle/Makefile into oracle/_ref/.
//
// This file is ours.  It pulls the reference's glimmer3.cc translation unit in WHOLE (from
// /root/reference, via the include path; nothing is copied) with its main() renamed, so that the
// file-static functions Parse_Command_Line / Score_Orfs / Trace_Back are callable, and it intercepts
// Add_Events_Fwd / Add_Events_Rev with the linker (--wrap) to see the start lists Score_Orfs builds.
//
//   ref_orfs dump  <glimmer3 options...> <fasta> <tag>     text dump on stdout:
//        R <read index> <n_orfs>
//        O <frame> <stop_position> <orf_len>                 every ORF Find_Orfs produced, in order
//        G <orf index> <gene_score %.17g> <gene_len> <n_starts>   every ORF Score_Orfs accepted
//        S <j> <pos> <score %a> <which> <truncated> <first>  its start list as handed to Add_Events_*
//   ref_orfs orfs | orfs-circular  <glimmer3 options, e.g. -i regions.txt> <fasta> <tag>
//        I <lo> <hi> per ignore region as Get_Ignore_Regions leaves them; R <read index> <n_orfs>, then  O <frame> <stop_position> <gene_len> <orf_len>  for every ORF of Find_Orfs with the ignore
//        regions of -i and (orfs-circular) Genome_Is_Circular set
// (The drop-in driver that runs these loops on the GPU is product code: integration/glimmer3_gpu.cc.)

#define main glimmer3_reference_main
#include "glimmer3.cc"
#undef main

#include <map>

// ---- interception of Add_Events_* (ld --wrap on the mangled names) -------------------------------------
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

void wrap_Add_Events_Fwd(const Orf_t &orf, vector<Start_t> &sl, int &id)
{
    if (Capture) Captured.push_back(make_pair(orf, sl));
    real_Add_Events_Fwd(orf, sl, id);
}
void wrap_Add_Events_Rev(const Orf_t &orf, vector<Start_t> &sl, int &id)
{
    if (Capture) Captured.push_back(make_pair(orf, sl));
    real_Add_Events_Rev(orf, sl, id);
}

// ---- the set-up steps of glimmer3's main (glimmer3.cc:175-223), in the same order ---------------------
static void setup_options(int argc, char **argv)
{
    Verbose = 0;
    Parse_Command_Line(argc, argv);
    Set_Start_And_Stop_Codons();
    Prob_To_Logs(Start_Prob);
    if (Feature_File != NULL) Parse_Features(Feature_File);
}

static void setup_models(void)
{
    if (!GC_Frac_Set) Set_GC_Fraction();
    Indep_Model.Build_Indep_WO_Stops(Indep_GC_Frac, Stop_Codon);
    Set_Ignore_Score_Len();
    Gene_ICM.Read(ICM_File_Name);
    LogOdds_PWM = Ribosome_PWM;
    LogOdds_PWM.Make_Log_Odds_WRT_GC(Indep_GC_Frac);
}

static void load_sequence(const vector<string> &seq_list, const vector<string> &hdr_list, int i)
{
    Fasta_Header = hdr_list[i].c_str();
    Sequence = seq_list[i];
    Sequence_Len = Sequence.length();
    for (int k = 0; k < Sequence_Len; k++) Sequence[k] = tolower(Filter(Sequence[k]));   // glimmer3.cc:270-271
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: ref_orfs dump <glimmer3 args>\n"); return 2; }
    string mode = argv[1];
    try {
        setup_options(argc - 1, argv + 1);
        vector<string> seq_list, hdr_list;
        vector<Orf_t> orf_list;
        vector<Gene_t> gene_list;

        if (mode == "orfs" || mode == "orfs-circular") {
            // Find_Orfs alone (glimmer_base.cc:638-817), with the ignore regions of -i (Get_Ignore_Regions, glimmer3.cc:179-180) and, for
            // "orfs-circular", Genome_Is_Circular (glimmer-mg's -r; this glimmer3 has no option for it): every ORF with all four fields
            if (Ignore_File_Name != NULL) Get_Ignore_Regions();
            Genome_Is_Circular = mode == "orfs-circular";
            for (size_t k = 0; k < Ignore_Region.size(); k++) printf("I %d %d\n", Ignore_Region[k].lo, Ignore_Region[k].hi);
            FILE *fp = File_Open(Sequence_File_Name, "r", __FILE__, __LINE__);
            Read_Sequences(fp, seq_list, hdr_list, Sequence_Ct);
            fclose(fp);
            for (int i = 0; i < Sequence_Ct; i++) {
                load_sequence(seq_list, hdr_list, i);
                Find_Orfs(orf_list);
                printf("R %d %d\n", i, (int)orf_list.size());
                for (size_t o = 0; o < orf_list.size(); o++)
                    printf("O %d %d %d %d\n", orf_list[o].Get_Frame(), orf_list[o].Get_Stop_Position(), orf_list[o].Get_Gene_Len(),
                           orf_list[o].Get_Orf_Len());
                orf_list.clear();
            }
            return 0;
        }
        if (mode == "dump") {
            setup_models();
            FILE *fp = File_Open(Sequence_File_Name, "r", __FILE__, __LINE__);
            Read_Sequences(fp, seq_list, hdr_list, Sequence_Ct);
            fclose(fp);
            for (int i = 0; i < Sequence_Ct; i++) {
                load_sequence(seq_list, hdr_list, i);
                Initialize_Terminal_Events(First_Event, Final_Event, Best_Event, Last_Event);
                Find_Orfs(orf_list);
                printf("R %d %d\n", i, (int)orf_list.size());
                map<pair<int, int>, int> index_of;      // (frame, stop_position) -> position in orf_list
                for (size_t o = 0; o < orf_list.size(); o++) {
                    printf("O %d %d %d\n", orf_list[o].Get_Frame(), orf_list[o].Get_Stop_Position(),
                           orf_list[o].Get_Orf_Len());
                    index_of[make_pair(orf_list[o].Get_Frame(), orf_list[o].Get_Stop_Position())] = o;
                }
                Capture = true;
                Captured.clear();
                Score_Orfs(orf_list, gene_list, NULL);
                Capture = false;
                for (size_t c = 0; c < Captured.size(); c++) {
                    const Orf_t &orf = Captured[c].first;
                    const vector<Start_t> &sl = Captured[c].second;
                    printf("G %d %.17g %d %d\n", index_of[make_pair(orf.Get_Frame(), orf.Get_Stop_Position())],
                           gene_list[c].Get_Score(), gene_list[c].Get_Gene_Len(), (int)sl.size());
                    for (size_t s = 0; s < sl.size(); s++)
                        printf("S %d %d %a %d %d %d\n", sl[s].j, sl[s].pos, sl[s].score, (int)sl[s].which,
                               (int)sl[s].truncated, (int)sl[s].first);
                }
                gene_list.clear();
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
