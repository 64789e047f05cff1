// glimmer3_gpu.cc -- glimmer3 with input parsing, Find_Orfs and the Score_Orfs inner loop on an MI355X.
//
// The reference's own glimmer3.cc is pulled in WHOLE from the reference tree at build time (main renamed; nothing is
// copied into this repository): option parsing, Add_Events_*, Process_Events, Trace_Back and the output format are
// the reference's code, unchanged.  What is replaced, for ALL sequences of the file in one call each:
//     Fasta_Read + tolower (Filter ()) + Set_GC_Fraction   -> gmg_fasta_ingest   (src/Common/fasta.cc:236-286, glimmer_base.cc:2564-2595)
//     Find_Orfs                                             -> gmg_find_orfs      (glimmer_base.cc:638-817)
//     Score_Orfs' scoring (buffers, two cumulative scores, start scan)  -> gmg_score_orfs  (glimmer3.cc:1275-1552)
//
//     glimmer3_gpu <glimmer3 options> <fasta> <tag>          (GMG_DEVICE selects the GPU)
// -M, -L and -i select loops that are not batched here: such a command line is handed to glimmer3_dropin (same directory).
//
// Output: <tag>.predict, byte-identical to the reference's (tests/test_gpu_dropin_cli.py).

#define main glimmer3_reference_main
#include "glimmer3.cc"
#undef main

#include "gmg.h"

#include <errno.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

// the set-up steps of glimmer3's main (glimmer3.cc:175-223), in the same order
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
}

// The modes whose loops are not batched here -- -M (every input sequence is one gene: Score_Separate_Input, glimmer3.cc:1555-1628),
// -L (ORFs from a coordinate file: Score_Orflist, :1177-1271) and -i (ignore regions in Find_Orfs, glimmer_base.cc:833-943) -- run
// in glimmer3_dropin beside this binary: the reference's own main() on the device-backed ICM_t (one launch per ICM_t call; same
// bytes).  Nothing has touched the GPU yet, so this process simply becomes that one.
static int run_dropin(const char *name, char **argv)
{
    char self[4096];
    const ssize_t n = readlink("/proc/self/exe", self, sizeof self - 1);
    string dir = n > 0 ? string(self, (size_t)n) : string(argv[0]);
    const size_t slash = dir.rfind('/');
    dir = slash == string::npos ? string(".") : dir.substr(0, slash);
    const string exe = dir + "/" + name;
    execv(exe.c_str(), argv);
    fprintf(stderr, "glimmer3_gpu: this option set needs %s (the reference's loop on the device-backed ICM_t), which could not be started: %s\n",
            exe.c_str(), strerror(errno));
    return 2;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: glimmer3_gpu <glimmer3 options> <fasta> <tag>\n"); return 2; }
    try {
        setup_options(argc, argv);
        if (Separate_Orf_Input || Orflist_File_Name != NULL || Ignore_File_Name != NULL || Genome_Is_Circular) return run_dropin("glimmer3_dropin", argv);
        vector<string> seq_list, hdr_list;
        // pass 1, on the device: the file's bytes are parsed there (gmg_fasta_ingest = Fasta_Read + tolower (Filter ()) +
        // packing + the g/c count of Set_GC_Fraction) and Find_Orfs runs for every read at once (gmg_find_orfs)
        const char *dev = getenv("GMG_DEVICE");
        if (gmg_init(dev ? atoi(dev) : 0) != GMG_OK) { fprintf(stderr, "%s\n", gmg_last_error()); return 1; }
        // the file, mapped (glimmer3 reads it with fgetc)
        const int fd = open(Sequence_File_Name, O_RDONLY);
        struct stat st;
        if (fd < 0 || fstat(fd, &st) != 0) { fprintf(stderr, "ERROR:  Could not open file  %s \n", Sequence_File_Name); return EXIT_FAILURE; }
        const size_t n_file = (size_t)st.st_size;
        const char *file_bytes = n_file ? (const char *)mmap(NULL, n_file, PROT_READ, MAP_PRIVATE, fd, 0) : "";
        if (file_bytes == MAP_FAILED) { perror("mmap"); return EXIT_FAILURE; }
        gmg_reads *reads = NULL;
        gmg_fasta *fasta = NULL;
        if (gmg_fasta_ingest(file_bytes, n_file, &reads, &fasta) != GMG_OK) { fprintf(stderr, "%s\n", gmg_last_error()); return 1; }
        uint64_t n_ing = 0, total_bases = 0, gc_ct = 0;
        gmg_fasta_info(fasta, &n_ing, &total_bases, &gc_ct);
        Sequence_Ct = (int)n_ing;
        if (!GC_Frac_Set) {                         // Set_GC_Fraction (glimmer_base.cc:2564-2595) without reading the file again
            Indep_GC_Frac = gmg_gc_fraction(&gc_ct, &total_bases, 1, 1);   // (the reference counts with unsigned int)
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
            hdr_list[i].assign(file_bytes + hb[i], he[i] - hb[i]);
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
        // (two steps: the start lists are a few per cent of the slots gmg_orfs_upload reserves -- only they get host memory)
        uint64_t n_used = 0;
        if (gmg_score_orfs_begin(Gene_ICM.Device_Model(), Indep_Model.Device_Model(), reads, batch, &prm, &n_used, NULL) != GMG_OK) {
            fprintf(stderr, "%s\n", gmg_last_error()); return 1; }
        vector<gmg_start> starts(n_used ? n_used : 1);
        if (gmg_score_orfs_fetch(batch, res.data(), starts.data(), NULL) != GMG_OK) { fprintf(stderr, "%s\n", gmg_last_error()); return 1; }
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
                if (all_orfs[i][o].Get_Frame() > 0) Add_Events_Fwd(all_orfs[i][o], sl, id);
                else Add_Events_Rev(all_orfs[i][o], sl, id);
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
        return 0;
    } catch (std::exception &e) {
        cerr << "** Standard Exception **" << endl << e << endl;
        return 1;
    }
}
