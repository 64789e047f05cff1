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
// -L (Score_Orflist, glimmer3.cc:1177-1271) and -M (Score_Separate_Input, :1555-1628) are batched too: every ORF of the coordinate
// list / every sequence of the file is one segment of ONE gmg_score_string call per model (score [m-4] of Cumulative_Score is
// Score_String of the buffer's first m-3 bases).  -i (ignore regions, glimmer_base.cc:689-731,833-943): the ORF lists come from the
// reference's own Find_Orfs on the host, the scoring of all of them is the same single gmg_score_orfs call.  Coordinate lists with
// wrap-around entries are handed to glimmer3_dropin (same directory).
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
#include <spawn.h>
#include <sys/wait.h>
extern char **environ;

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

// What is not batched here -- coordinate lists (-L) with entries that wrap around the sequence's end or leave it -- runs
// in glimmer3_dropin beside this binary: the reference's own main() on the device-backed ICM_t (one launch per ICM_t call; same
// bytes).  It is started as a CHILD (posix_spawn + waitpid) and its exit status handed on: under a profiler this process may
// have initialised the GPU before main(), and replacing a GPU-initialised process (exec*) is what this pool forbids.
static int run_dropin(const char *name, char **argv)
{
    char self[4096];
    const ssize_t n = readlink("/proc/self/exe", self, sizeof self - 1);
    string dir = n > 0 ? string(self, (size_t)n) : string(argv[0]);
    const size_t slash = dir.rfind('/');
    dir = slash == string::npos ? string(".") : dir.substr(0, slash);
    const string exe = dir + "/" + name;
    pid_t pid = 0;
    const int rc = posix_spawn(&pid, exe.c_str(), NULL, NULL, argv, environ);
    if (rc != 0) {
        fprintf(stderr, "glimmer3_gpu: this option set needs %s (the reference's loop on the device-backed ICM_t), which could not be started: %s\n",
                exe.c_str(), strerror(rc));
        return 2;
    }
    int status = 0;
    while (waitpid(pid, &status, 0) < 0)
        if (errno != EINTR) { perror("glimmer3_gpu: waitpid"); return 2; }
    return WIFEXITED(status) ? WEXITSTATUS(status) : 128 + (WIFSIGNALED(status) ? WTERMSIG(status) : 0);
}

// -L / -M on the device.  Per ORF the reference builds the buffer (Reverse_Transfer: forward strand, bases hi-1 downwards, not
// complemented; Complement_Transfer: reverse strand, bases lo upwards, complemented, not reversed), takes Cumulative_Score of both
// models from frame 1 and prints 100 * (score [m-4] - indep_score [m-4]) / (m - 3): the sums over the buffer's first m - 3 bases,
// i.e. Score_String (buff, m - 3, 1) with the window rule of a buffer that starts at the ORF's 3' end (partial windows for its
// first model_len - 1 bases).  Returns -1 when an entry is not a plain segment of the sequence (wrap-around, out of range, too
// short): the caller hands the run to glimmer3_dropin.
static int run_orf_scores_batched(char **argv)
{
    const char *dev = getenv("GMG_DEVICE");
    const int fd = open(Sequence_File_Name, O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0) { fprintf(stderr, "ERROR:  Could not open file  %s \n", Sequence_File_Name); return EXIT_FAILURE; }
    const size_t n_file = (size_t)st.st_size;
    const char *file_bytes = n_file ? (const char *)mmap(NULL, n_file, PROT_READ, MAP_PRIVATE, fd, 0) : "";
    if (file_bytes == MAP_FAILED) { perror("mmap"); return EXIT_FAILURE; }
    // the coordinates are checked against the sequence lengths on the host first (the first pass of Fasta_Read's state machine over
    // the headers and bases is the device's: lengths come back with the ingest) -- so the device is needed from here
    if (gmg_init(dev ? atoi(dev) : 0) != GMG_OK) { fprintf(stderr, "%s\n", gmg_last_error()); return 1; }
    gmg_reads *reads = NULL;
    gmg_fasta *fasta = NULL;
    if (gmg_fasta_ingest(file_bytes, n_file, &reads, &fasta) != GMG_OK) { fprintf(stderr, "%s\n", gmg_last_error()); return 1; }
    uint64_t n_ing = 0, total_bases = 0, gc_ct = 0;
    gmg_fasta_info(fasta, &n_ing, &total_bases, &gc_ct);
    Sequence_Ct = (int)n_ing;
    vector<uint64_t> hb(n_ing), he(n_ing), off(n_ing + 1);
    gmg_fasta_headers(fasta, hb.data(), he.data());
    vector<uint32_t> packed(gmg_packed_words(total_bases) + 1, 0);
    gmg_reads_download(reads, packed.data(), off.data());
    gmg_fasta_free(fasta);
    vector<gmg_segment> segs;
    vector<int> m_of;                                   // m = the buffer's length
    bool plain = true;
    if (Separate_Orf_Input) {                           // every sequence is one ORF in frame +1, its stop codon included (:1573-1574)
        for (uint64_t i = 0; i < n_ing && plain; i++) {
            const int64_t len = (int64_t)(off[i + 1] - off[i]) - 3;
            if (len < 4) { plain = false; break; }
            gmg_segment g = {(uint32_t)i, 3u, (uint32_t)(len - 3), (uint32_t)GMG_REVERSED};
            segs.push_back(g);
            m_of.push_back((int)len);
        }
    } else {                                            // the coordinate list against the FIRST sequence (:244-252)
        if (n_ing == 0) plain = false;
        const int64_t n = n_ing ? (int64_t)(off[1] - off[0]) : 0;
        for (size_t i = 0; i < Orf_Pos_List.size() && plain; i++) {
            const int64_t start = Orf_Pos_List[i].start, stop = Orf_Pos_List[i].stop;
            if (Orf_Pos_List[i].dir > 0) {              // buffer = bases stop-4 .. start-1 (0-based), downwards
                const int64_t len = 1 + stop - start - 3;
                if (start < 1 || stop - 3 > n || stop - 3 <= 0 || len < 4) { plain = false; break; }
                gmg_segment g = {0u, (uint32_t)(start + 2), (uint32_t)(len - 3), (uint32_t)GMG_REVERSED};
                segs.push_back(g);
                m_of.push_back((int)len);
            } else {                                    // buffer = complement of bases stop+2 .. (0-based), upwards
                const int64_t len = 1 + start - stop - 3;
                if (stop < 1 || start > n || stop + 2 >= n || len < 4) { plain = false; break; }
                gmg_segment g = {0u, (uint32_t)(stop + 2), (uint32_t)(len - 3), (uint32_t)GMG_COMPLEMENTED};
                segs.push_back(g);
                m_of.push_back((int)len);
            }
        }
    }
    if (!plain) { gmg_reads_free(reads); return -1; }
    if (!GC_Frac_Set) {                                 // Set_GC_Fraction (glimmer_base.cc:2564-2595) without reading the file again
        Indep_GC_Frac = gmg_gc_fraction(&gc_ct, &total_bases, 1, 1);
        GC_Frac_Set = true;
    }
    setup_models();
    const uint64_t ns = segs.size();
    vector<double> gene_sum(ns ? ns : 1), indep_sum(ns ? ns : 1);
    if (ns) {
        gmg_segments *dsegs = NULL;
        void *d_sums = NULL;
        uint64_t total_len = 0;
        if (gmg_segments_upload(reads, segs.data(), ns, NULL, &total_len, &dsegs) != GMG_OK || gmg_device_malloc(&d_sums, 2 * ns * sizeof(double)) != GMG_OK ||
            gmg_score_string(Gene_ICM.Device_Model(), reads, dsegs, 1, (double *)d_sums, NULL) != GMG_OK ||
            gmg_score_string(Indep_Model.Device_Model(), reads, dsegs, 1, (double *)d_sums + ns, NULL) != GMG_OK ||
            gmg_memcpy_d2h(gene_sum.data(), d_sums, ns * sizeof(double), NULL) != GMG_OK ||
            gmg_memcpy_d2h(indep_sum.data(), (double *)d_sums + ns, ns * sizeof(double), NULL) != GMG_OK) {
            fprintf(stderr, "%s\n", gmg_last_error());
            return 1;
        }
        gmg_device_free(d_sums);
        gmg_segments_free(dsegs);
    }
    string filename = Output_Tag;
    filename.append(".predict");
    FILE *predict_fp = File_Open(filename, "w", __FILE__, __LINE__);
    for (uint64_t i = 0; i < ns; i++) {
        const int m = m_of[i];
        const double gene_score = 100.0 * (gene_sum[i] - indep_sum[i]) / (m - 3);
        if (Separate_Orf_Input) {
            // the tag: the header's first token, or Seq%04d (:1575-1580)
            char line[MAX_LINE], tag[MAX_LINE];
            const size_t hl = he[i] - hb[i] < (uint64_t)MAX_LINE - 1 ? (size_t)(he[i] - hb[i]) : (size_t)MAX_LINE - 1;
            memcpy(line, file_bytes + hb[i], hl);
            line[hl] = 0;
            char *p = strtok(line, " \t\n");
            if (p == NULL) sprintf(tag, "Seq%04d", (int)i);
            else strcpy(tag, p);
            fprintf(predict_fp, "%-14s %8d %8d %+3d %8.2f\n", tag, 1, m, 1, gene_score);
        } else {
            const int start = Orf_Pos_List[i].start, stop = Orf_Pos_List[i].stop;
            const int frame = Orf_Pos_List[i].dir > 0 ? 1 + (stop % 3) : -((stop - 1) % 3) - 1;
            fprintf(predict_fp, "%-14s %8d %8d %+3d %8.2f\n", Orf_Pos_List[i].tag, start, stop, frame, gene_score);
        }
    }
    fclose(predict_fp);
    gmg_reads_free(reads);
    (void)argv;
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: glimmer3_gpu <glimmer3 options> <fasta> <tag>\n"); return 2; }
    try {
        setup_options(argc, argv);
        if (Genome_Is_Circular) return run_dropin("glimmer3_dropin", argv);      // (no option of this glimmer3 sets it: glimmer3.cc:53,874,959)
        if (Ignore_File_Name != NULL) Get_Ignore_Regions();                        // (glimmer3.cc:179-180)
        if (Separate_Orf_Input || Orflist_File_Name != NULL) {
            if (Orflist_File_Name != NULL && !Separate_Orf_Input) Get_Orf_Pos_List();      // (glimmer3.cc:182-183)
            const int rc = run_orf_scores_batched(argv);
            return rc >= 0 ? rc : run_dropin("glimmer3_dropin", argv);
        }
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
        vector<vector<Orf_t> > all_orfs(Sequence_Ct);
        vector<gmg_orf> orfs;
        {
        gmg_mg_params fprm;
        memset(&fprm, 0, sizeof fprm);
        fprm.min_gene_len = Min_Gene_Len;
        fprm.allow_truncated = Allow_Truncated_Orfs;
        fprm.n_start_codons = Start_Codon.size();
        fprm.n_stop_codons = Stop_Codon.size();
        for (size_t c = 0; c < Start_Codon.size() && c < 8; c++) memcpy(fprm.start_codon[c], Start_Codon[c], 3);
        for (size_t c = 0; c < Stop_Codon.size() && c < 8; c++) memcpy(fprm.stop_codon[c], Stop_Codon[c], 3);
        // -i: the ignore regions (Get_Ignore_Regions has sorted and merged them; the same regions screen every sequence,
        // glimmer_base.cc:844-847) go with the call: Find_Orfs' scan stops at a region and starts anew behind it on the device too
        vector<int32_t> ign_lo, ign_hi;
        for (size_t k = 0; k < Ignore_Region.size(); k++) { ign_lo.push_back(Ignore_Region[k].lo); ign_hi.push_back(Ignore_Region[k].hi); }
        fprm.n_ignore_regions = (int32_t)ign_lo.size();
        fprm.ignore_lo = ign_lo.empty() ? NULL : ign_lo.data();
        fprm.ignore_hi = ign_hi.empty() ? NULL : ign_hi.data();
        gmg_mg_result *found = NULL;
        if (gmg_find_orfs(reads, &fprm, &found, NULL) != GMG_OK) { fprintf(stderr, "%s\n", gmg_last_error()); return 1; }
        uint64_t n_found = 0;
        gmg_mg_result_info(found, &n_found, NULL);
        vector<gmg_mg_orf> frec(n_found ? n_found : 1);
        vector<uint64_t> first(Sequence_Ct + 1);
        if (gmg_mg_result_fetch(found, frec.data(), NULL, first.data()) != GMG_OK) { fprintf(stderr, "%s\n", gmg_last_error()); return 1; }
        gmg_mg_result_free(found);
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
