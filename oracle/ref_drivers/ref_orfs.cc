// ref_orfs.cc -- golden vectors for the Score_Orfs inner loop (src/Glimmer/glimmer3.cc:1275-1552) and the
// batch-scoring integration demo.  Test infrastructure only; built by oracle/Makefile into oracle/_ref/.
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
//   ref_orfs batch <glimmer3 options...> <fasta> <tag>     (built with -DGMG_BATCH, links libgmg.so)
//        same pipeline as glimmer3's main, but the input is parsed on the device (gmg_fasta_ingest), Find_Orfs of all
//        reads is ONE gmg_find_orfs call and Score_Orfs ONE gmg_score_orfs call; events, DP and trace-back stay the
//        reference's host code; writes <tag>.predict, which must equal the reference's byte for byte.

#define main glimmer3_reference_main
#include "glimmer3.cc"
#undef main

#include <map>

#ifdef GMG_BATCH
#include "gmg.h"
#endif

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
    if (argc < 4) { fprintf(stderr, "usage: ref_orfs dump|batch <glimmer3 args>\n"); return 2; }
    string mode = argv[1];
    try {
        setup_options(argc - 1, argv + 1);
        vector<string> seq_list, hdr_list;
        vector<Orf_t> orf_list;
        vector<Gene_t> gene_list;

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
#ifdef GMG_BATCH
        if (mode == "batch") {
            // pass 1, on the device: the file's bytes are parsed there (gmg_fasta_ingest = Fasta_Read + tolower (Filter ()) +
            // packing + the g/c count of Set_GC_Fraction) and Find_Orfs runs for every read at once (gmg_find_orfs)
            const char *dev = getenv("GMG_DEVICE");
            if (gmg_init(dev ? atoi(dev) : 0) != GMG_OK) { fprintf(stderr, "%s\n", gmg_last_error()); return 1; }
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
            Sequence_Ct = (int)n_ing;
            if (!GC_Frac_Set) {                         // Set_GC_Fraction (glimmer_base.cc:2564-2595) without reading the file again
                Indep_GC_Frac = double(gc_ct) / total_bases;
                GC_Frac_Set = true;
            }
            setup_models();
            vector<uint64_t> hb(n_ing), he(n_ing), off(n_ing + 1);
            gmg_fasta_headers(fasta, hb.data(), he.data());
            vector<uint32_t> packed(gmg_packed_words(total_bases) + 1, 0);
            gmg_reads_download(reads, packed.data(), off.data());
            gmg_fasta_free(fasta);
            seq_list.resize(Sequence_Ct);
            hdr_list.resize(Sequence_Ct);
            for (int i = 0; i < Sequence_Ct; i++) {     // the event / DP code reads the global Sequence: filtered bases back from the device
                hdr_list[i] = file_bytes.substr(hb[i], he[i] - hb[i]);
                string &sq = seq_list[i];
                sq.resize(off[i + 1] - off[i]);
                for (uint64_t k = 0; k < sq.size(); k++) { const uint64_t g = off[i] + k; sq[k] = "acgt"[(packed[g >> 4] >> (2 * (g & 15))) & 3]; }
            }
            gmg_mg_params fprm;
            memset(&fprm, 0, sizeof fprm);
            fprm.min_gene_len = Min_Gene_Len;
            fprm.allow_truncated = Allow_Truncated_Orfs;
            fprm.n_start_codons = Start_Codon.size();
            fprm.n_stop_codons = Stop_Codon.size();
            for (size_t c = 0; c < Start_Codon.size() && c < 8; c++) memcpy(fprm.start_codon[c], Start_Codon[c], 3);
            for (size_t c = 0; c < Stop_Codon.size() && c < 8; c++) memcpy(fprm.stop_codon[c], Stop_Codon[c], 3);
            gmg_mg_result *found = NULL;
            if (gmg_find_orfs(reads, &fprm, &found, NULL) != GMG_OK) { fprintf(stderr, "%s\n", gmg_last_error()); return 1; }
            uint64_t n_found = 0;
            gmg_mg_result_info(found, &n_found, NULL);
            vector<gmg_mg_orf> frec(n_found ? n_found : 1);
            vector<uint64_t> first(Sequence_Ct + 1);
            if (gmg_mg_result_fetch(found, frec.data(), NULL, first.data()) != GMG_OK) { fprintf(stderr, "%s\n", gmg_last_error()); return 1; }
            gmg_mg_result_free(found);
            vector<vector<Orf_t> > all_orfs(Sequence_Ct);
            vector<gmg_orf> orfs;
            for (int i = 0; i < Sequence_Ct; i++)
                for (uint64_t o = first[i]; o < first[i + 1]; o++) {
                    Orf_t orf;
                    orf.Set_Stop_Position(frec[o].stop_position);
                    orf.Set_Frame(frec[o].frame);
                    orf.Set_Gene_Len(frec[o].gene_len);
                    orf.Set_Orf_Len(frec[o].orf_len);
                    all_orfs[i].push_back(orf);
                    gmg_orf g = {(uint32_t)i, frec[o].frame, frec[o].stop_position, frec[o].orf_len};
                    orfs.push_back(g);
                }
            // ONE batch call for the Score_Orfs inner loops of all reads
            gmg_orf_params prm;
            memset(&prm, 0, sizeof prm);
            prm.min_gene_len = Min_Gene_Len;
            prm.allow_truncated = Allow_Truncated_Orfs;
            prm.use_first_start = Use_First_Start_Codon;
            prm.ignore_score_len = Ignore_Score_Len;
            prm.start_threshold = Start_Threshold;
            prm.n_start_codons = Start_Codon.size();
            for (size_t s = 0; s < Start_Codon.size() && s < 8; s++) memcpy(prm.start_codon[s], Start_Codon[s], 3);
            vector<gmg_orf_result> res(orfs.size());
            uint64_t n_starts = 0;
            gmg_orf_batch *batch = NULL;
            if (gmg_orfs_upload(reads, orfs.data(), orfs.size(), &n_starts, &batch) != GMG_OK) { fprintf(stderr, "%s\n", gmg_last_error()); return 1; }
            vector<gmg_start> starts(n_starts);
            if (gmg_score_orfs(Gene_ICM.Device_Model(), Indep_Model.Device_Model(), reads, batch, &prm, res.data(),
                               starts.data(), NULL) != GMG_OK) { fprintf(stderr, "%s\n", gmg_last_error()); return 1; }
            // pass 2: events, DP and trace-back per read (host, unchanged reference code)
            string filename = Output_Tag;
            filename.append(".predict");
            FILE *predict_fp = File_Open(filename, "w", __FILE__, __LINE__);
            size_t o_base = 0;
            for (int i = 0; i < Sequence_Ct; i++) {
                load_sequence(seq_list, hdr_list, i);
                fprintf(predict_fp, ">%s\n", Fasta_Header);
                Initialize_Terminal_Events(First_Event, Final_Event, Best_Event, Last_Event);
                int id = 0;
                for (size_t o = 0; o < all_orfs[i].size(); o++) {
                    const gmg_orf_result &r = res[o_base + o];
                    if (!r.is_tentative_gene) continue;
                    vector<Start_t> sl(r.n_starts);
                    for (uint32_t s = 0; s < r.n_starts; s++) {
                        const gmg_start &g = starts[r.start_begin + s];
                        sl[s].j = g.j; sl[s].pos = g.pos; sl[s].score = g.score; sl[s].which = g.which;
                        sl[s].truncated = g.truncated; sl[s].first = g.first;
                    }
                    if (all_orfs[i][o].Get_Frame() > 0) real_Add_Events_Fwd(all_orfs[i][o], sl, id);
                    else real_Add_Events_Rev(all_orfs[i][o], sl, id);
                }
                o_base += all_orfs[i].size();
                Process_Events();
                Set_Final_Event(Final_Event, Best_Event, Sequence_Len);
                Trace_Back(predict_fp, Final_Event);
                Clear_Events();
            }
            fclose(predict_fp);
            gmg_orf_batch_free(batch);
            gmg_reads_free(reads);
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
