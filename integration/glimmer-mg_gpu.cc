// glimmer-mg_gpu.cc -- glimmer-mg with its front half on MI355X GPUs, one process per GPU.
//
// The reference's own glimmer-mg.cc is pulled in WHOLE from the reference tree at build time (main renamed; nothing
// is copied into this repository): option parsing, Add_Events_*, Process_Events, Trace_Back and the output format are
// the reference's code, unchanged.  What is replaced, for ALL reads of a batch in one call each:
//     Fasta_Read + tolower (Filter ()) + Set_GC_Fraction       -> gmg_fasta_ingest          (src/Common/fasta.cc:236-286,
//                                                                                             glimmer_base.cc:2564-2595)
//     Score_All_Frames + Find_Orfs + Score_Orfs_Errors          -> gmg_mg_score_reads        (glimmer-mg.cc:1468-1510,
//                                                                                             1605-1861; glimmer_base.cc:638-817)
// Only the user-ICM mode (-m <icm>, no -c classifications) is driven; -i / -s / -q are.
//
//     glimmer-mg_gpu [--shards N] [--gpus G] [--batch-bytes B] <glimmer-mg options> <fasta> <tag>
//
// --shards N   the file is cut into N byte ranges at record starts (gmg_fasta_shard_ranges); N child processes are
//              forked BEFORE anything touches a GPU, child k binds to device k mod G, ingests and scores its range and
//              writes <tag>.predict.part<k>.  The one run-wide quantity, the null model's GC fraction
//              (Set_GC_Fraction: a ratio of two counts over the whole file), is summed by the parent from the children's
//              {gc, total} (two integers per child through a pipe) and handed back; the parent concatenates the parts
//              in shard order.  No collective, no GPU-to-GPU traffic.
// --batch-bytes B   inside a shard the bytes are ingested and scored in pieces of about B bytes (gmg_fasta_split;
//              default 256 MiB = about 0.5 M reads of 500 bp): the 48 B/base table of a piece must fit the HBM, and a
//              piece's ORF / start counts the 32-bit fields of the result records.  The packed reads of all pieces stay
//              resident (0.25 B/base) between the counting pass and the scoring pass.
// Output: <tag>.predict, byte-identical to the reference's (tests/test_gpu_dropin_cli.py).

#include "glimmer-mg.hh"
#define main glimmer_mg_reference_main
#include "glimmer-mg.cc"
#undef main

#include "gmg.h"

#include <fcntl.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

static const uint64_t DEFAULT_BATCH_BYTES = 256ull << 20;

static void die_gmg(const char *what)
{
    fprintf(stderr, "glimmer-mg_gpu: %s: %s\n", what, gmg_last_error());
    exit(EXIT_FAILURE);
}

static bool full_write(int fd, const void *p, size_t n)
{
    const char *c = (const char *)p;
    while (n) { ssize_t w = write(fd, c, n); if (w <= 0) return false; c += w; n -= (size_t)w; }
    return true;
}

static bool full_read(int fd, void *p, size_t n)
{
    char *c = (char *)p;
    while (n) { ssize_t r = read(fd, c, n); if (r <= 0) return false; c += r; n -= (size_t)r; }
    return true;
}

// the set-up steps of glimmer-mg's main for -m <icm> (glimmer-mg.cc:241-316), in the same order
static void setup_options(int argc, char **argv)
{
    Verbose = 0;
    Parse_Command_Line(argc, argv);
    Set_Start_And_Stop_Codons();
    if (Feature_File != NULL) Parse_Features(Feature_File);
    if (!User_ICM || !classifications.empty() || Detail_Log) {
        fprintf(stderr, "glimmer-mg_gpu: only -m <icm> without -c / detail log is driven here\n");
        exit(2);
    }
}

static void setup_models(void)
{
    Indep_Model.Build_Indep_WO_Stops(Indep_GC_Frac, Stop_Codon);
    Set_Ignore_Score_Len();
    if (User_RBS) {
        LogOdds_PWM = Ribosome_PWM;
        LogOdds_PWM.Make_Log_Odds_WRT_GC(Indep_GC_Frac);
    }
    Gene_ICM.Read(ICM_File_Name);
}

struct Piece {
    gmg_reads *reads;
    uint64_t byte0, n_reads, total_bases;
    vector<uint64_t> hdr_begin, hdr_end;                // header extents, relative to the piece's first byte
};

// One shard: bytes [b0, b1) of the file.  up / down: pipes to / from the parent (-1: the shard is the whole job).
static int run_shard(const char *bytes, uint64_t b0, uint64_t b1, int device, uint64_t batch_bytes, int up, int down,
                     const string &out_name)
{
    if (gmg_init(device) != GMG_OK) die_gmg("gmg_init");
    const bool error_mode = Allow_Indels || Allow_Subs;

    // pass 1: every piece of the shard onto the device (parsed there); the shard's {gc, total}
    vector<Piece> pieces;
    uint64_t gc = 0, total = 0;
    if (b1 > b0) {
        const int max_pieces = (int)((b1 - b0) / (batch_bytes ? batch_bytes : 1)) + 2;
        vector<uint64_t> cuts(max_pieces + 1);
        const int n_pieces = gmg_fasta_split(bytes + b0, b1 - b0, batch_bytes, cuts.data(), max_pieces);
        if (n_pieces < 0) die_gmg("gmg_fasta_split");
        for (int p = 0; p < n_pieces; p++) {
            Piece pc;
            pc.byte0 = b0 + cuts[p];
            gmg_fasta *index = NULL;
            if (gmg_fasta_ingest(bytes + pc.byte0, cuts[p + 1] - cuts[p], &pc.reads, &index) != GMG_OK) die_gmg("gmg_fasta_ingest");
            uint64_t g = 0;
            gmg_fasta_info(index, &pc.n_reads, &pc.total_bases, &g);
            pc.hdr_begin.resize(pc.n_reads);
            pc.hdr_end.resize(pc.n_reads);
            if (pc.n_reads) gmg_fasta_headers(index, pc.hdr_begin.data(), pc.hdr_end.data());
            gmg_fasta_free(index);
            gc += g;
            total += pc.total_bases;
            pieces.push_back(pc);
        }
    }
    // the job's GC fraction (Set_GC_Fraction, glimmer_base.cc:2564-2595): the reference counts with `unsigned int`
    if (!GC_Frac_Set) {
        if (up >= 0) {
            const uint64_t mine[2] = {gc, total};
            double job_gc = 0.0;
            if (!full_write(up, mine, sizeof mine) || !full_read(down, &job_gc, sizeof job_gc)) {
                fprintf(stderr, "glimmer-mg_gpu: lost the parent process\n");
                return EXIT_FAILURE;
            }
            Indep_GC_Frac = job_gc;
        } else
            Indep_GC_Frac = gmg_gc_fraction(&gc, &total, 1, 1);
        GC_Frac_Set = true;
    }
    setup_models();

    gmg_mg_params prm;
    memset(&prm, 0, sizeof prm);
    prm.min_gene_len = Min_Gene_Len;
    prm.allow_truncated = Allow_Truncated_Orfs;
    prm.ignore_score_len = Ignore_Score_Len;
    prm.start_threshold = Start_Threshold;
    prm.flags = GMG_MG_ACCEPTED_ONLY;                   // only what Add_Events_* will see comes back
    if (error_mode) {                                   // -i / -s: Score_Indels / the substitution branch run on the device too
        prm.flags |= Allow_Indels ? GMG_MG_ALLOW_INDELS : GMG_MG_ALLOW_SUBS;
        prm.min_indel_orf_len = Min_Indel_ORF_Len;
        prm.indel_quality_threshold = Indel_Quality_Threshold;
        prm.indel_max = Indel_Max;
        prm.indel_suffix_score_threshold = Indel_Suffix_Score_Threshold;
    }
    prm.n_start_codons = Start_Codon.size();
    prm.n_stop_codons = Stop_Codon.size();
    for (size_t s = 0; s < Start_Codon.size() && s < 8; s++) memcpy(prm.start_codon[s], Start_Codon[s], 3);
    for (size_t s = 0; s < Stop_Codon.size() && s < 8; s++) memcpy(prm.stop_codon[s], Stop_Codon[s], 3);
    FILE *quality_fp = NULL;                            // -q: the values are read in file order, piece by piece
    if (Allow_Indels && Quality_File_Name != NULL) quality_fp = File_Open(Quality_File_Name, "r", __FILE__, __LINE__);

    // pass 2: piece by piece -- one gmg_mg_score_reads call, then events / DP / trace-back per read on the host
    FILE *predict_fp = File_Open(out_name, "w", __FILE__, __LINE__);
    for (size_t p = 0; p < pieces.size(); p++) {
        Piece &pc = pieces[p];
        const int n_seq = (int)pc.n_reads;
        vector<uint64_t> off(pc.n_reads + 1);
        vector<uint32_t> packed(gmg_packed_words(pc.total_bases) + 1, 0);
        if (gmg_reads_download(pc.reads, packed.data(), off.data()) != GMG_OK) die_gmg("gmg_reads_download");
        vector<uint8_t> qual_all;
        prm.quality = NULL;
        if (quality_fp) {                               // the user's Phred values, one byte per base
            qual_all.reserve(pc.total_bases);
            vector<int> q;
            string header;
            for (int i = 0; i < n_seq; i++) {
                Fasta_Qual_Vec_Read(quality_fp, q, header);
                if (q.size() != off[i + 1] - off[i]) {  // Clean_Quality_454's check (glimmer-mg.cc:534-537)
                    fprintf(stderr, "ERROR:  %s sequence length does not match quality values length\n", header.c_str());
                    return EXIT_FAILURE;
                }
                for (size_t k = 0; k < q.size(); k++) qual_all.push_back(q[k] > 255 ? 255 : q[k] < 0 ? 0 : q[k]);
            }
            prm.quality = qual_all.data();
        }
        gmg_mg_result *res = NULL;
        if (gmg_mg_score_reads(Gene_ICM.Device_Model(), Indep_Model.Device_Model(), pc.reads, &prm, NULL, &res, NULL) != GMG_OK)
            die_gmg("gmg_mg_score_reads");
        uint64_t n_orfs = 0, n_starts = 0;
        gmg_mg_result_info(res, &n_orfs, &n_starts);
        vector<gmg_mg_orf> orfs(n_orfs ? n_orfs : 1);
        vector<gmg_start> starts(n_starts ? n_starts : 1);
        vector<uint64_t> read_orf_off(pc.n_reads + 1);
        vector<gmg_start_errors> errs(error_mode ? (n_starts ? n_starts : 1) : 0);
        if (gmg_mg_result_fetch(res, orfs.data(), starts.data(), read_orf_off.data()) != GMG_OK) die_gmg("gmg_mg_result_fetch");
        if (error_mode && gmg_mg_result_fetch_errors(res, errs.data()) != GMG_OK) die_gmg("gmg_mg_result_fetch_errors");
        gmg_mg_result_free(res);
        gmg_reads_free(pc.reads);
        pc.reads = NULL;

        string hdr;
        for (int i = 0; i < n_seq; i++) {
            // what glimmer-mg.cc:376-382 prepares per read: header, filtered lower-case sequence (back from the device)
            hdr.assign(bytes + pc.byte0 + pc.hdr_begin[i], pc.hdr_end[i] - pc.hdr_begin[i]);
            Fasta_Header = hdr.c_str();
            Sequence.resize(off[i + 1] - off[i]);
            for (uint64_t k = 0; k < Sequence.size(); k++) {
                const uint64_t g = off[i] + k;
                Sequence[k] = "acgt"[(packed[g >> 4] >> (2 * (g & 15))) & 3];
            }
            Sequence_Len = Sequence.length();
            fprintf(predict_fp, ">%s\n", Fasta_Header);
            Initialize_Terminal_Events(First_Event, Final_Event, Best_Event, Last_Event);
            Meta_PWM_Save.resize(2 * Sequence_Len);                        // glimmer-mg.cc:1622-1627
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
                    if (error_mode) {
                        const gmg_start_errors &e = errs[g.start_begin + s];
                        for (int k = 0; k < e.n; k++) sl[s].errors.push_back(Error_t(e.pos[k], e.type[k]));
                    }
                }
                std::sort(sl.begin(), sl.end(), Start_Cmp);                // glimmer-mg.cc:1659: same algorithm on the same push order
                if (g.accepted == 2) {                                     // ties on pos: first_j is the sort's to decide (:1661-1666)
                    const int first_j = g.frame > 0 ? sl.front().j : sl.back().j;
                    if (first_j + 1 < Min_Gene_Len) continue;
                }
                if (g.frame > 0) Add_Events_Fwd(orf, sl, id);
                else Add_Events_Rev(orf, sl, id);
            }
            Process_Events();
            Set_Final_Event(Final_Event, Best_Event, Sequence_Len);
            Trace_Back(predict_fp, Final_Event);
            Clear_Events();
        }
    }
    fclose(predict_fp);
    if (quality_fp) fclose(quality_fp);
    return EXIT_SUCCESS;
}

int main(int argc, char **argv)
{
    int n_shards = 1, n_gpus = 1;
    uint64_t batch_bytes = DEFAULT_BATCH_BYTES;
    if (const char *e = getenv("GMG_GPUS")) n_gpus = atoi(e);
    // our own options come first; the rest is glimmer-mg's command line, untouched
    vector<char *> rest(1, argv[0]);
    int a = 1;
    for (; a + 1 < argc; a += 2) {
        if (strcmp(argv[a], "--shards") == 0) n_shards = atoi(argv[a + 1]);
        else if (strcmp(argv[a], "--gpus") == 0) n_gpus = atoi(argv[a + 1]);
        else if (strcmp(argv[a], "--batch-bytes") == 0) batch_bytes = strtoull(argv[a + 1], NULL, 10);
        else break;
    }
    for (; a < argc; a++) rest.push_back(argv[a]);
    if (rest.size() < 3 || n_shards < 1 || n_gpus < 1 || batch_bytes == 0 || batch_bytes >= 0x7fffffffull) {
        fprintf(stderr, "usage: glimmer-mg_gpu [--shards N] [--gpus G] [--batch-bytes B < 2^31] <glimmer-mg options> <fasta> <tag>\n");
        return 2;
    }
    try {
        setup_options((int)rest.size(), rest.data());
        if (n_shards > 1 && Quality_File_Name != NULL) {
            fprintf(stderr, "glimmer-mg_gpu: -q with --shards > 1 is not supported (the quality file is read in order)\n");
            return 2;
        }
        // the whole file, mapped once; the children inherit the mapping
        const int fd = open(Sequence_File_Name, O_RDONLY);
        struct stat st;
        if (fd < 0 || fstat(fd, &st) != 0) { fprintf(stderr, "ERROR:  Could not open file  %s \n", Sequence_File_Name); return EXIT_FAILURE; }
        const uint64_t n_bytes = (uint64_t)st.st_size;
        const char *bytes = n_bytes ? (const char *)mmap(NULL, n_bytes, PROT_READ, MAP_PRIVATE, fd, 0) : "";
        if (bytes == MAP_FAILED) { perror("mmap"); return EXIT_FAILURE; }
        const string out = string(Output_Tag) + ".predict";
        const int env_dev = getenv("GMG_DEVICE") ? atoi(getenv("GMG_DEVICE")) : 0;
        if (n_shards == 1) return run_shard(bytes, 0, n_bytes, env_dev, batch_bytes, -1, -1, out);

        vector<uint64_t> cuts(n_shards + 1);
        if (gmg_fasta_shard_ranges(bytes, n_bytes, n_shards, cuts.data()) != GMG_OK) die_gmg("gmg_fasta_shard_ranges");   // host only
        vector<pid_t> pid(n_shards);
        vector<int> up(n_shards), down(n_shards);
        for (int k = 0; k < n_shards; k++) {
            int pu[2], pd[2];
            if (pipe(pu) != 0 || pipe(pd) != 0) { perror("pipe"); return EXIT_FAILURE; }
            fflush(NULL);
            pid[k] = fork();                            // nothing in this process has touched a GPU
            if (pid[k] < 0) { perror("fork"); return EXIT_FAILURE; }
            if (pid[k] == 0) {
                close(pu[0]);
                close(pd[1]);
                for (int j = 0; j < k; j++) { close(up[j]); close(down[j]); }
                char part[32];
                snprintf(part, sizeof part, ".part%d", k);
                const int rc = run_shard(bytes, cuts[k], cuts[k + 1], (env_dev + k) % n_gpus, batch_bytes, pu[1], pd[0], out + part);
                fflush(NULL);
                _exit(rc);
            }
            close(pu[1]);
            close(pd[0]);
            up[k] = pu[0];
            down[k] = pd[1];
        }
        int rc = EXIT_SUCCESS;
        signal(SIGPIPE, SIG_IGN);                       // a child that died must not take the parent with it: the write fails instead
        if (!GC_Frac_Set) {                             // two integers up per child, one double down
            vector<uint64_t> gc(n_shards), total(n_shards);
            for (int k = 0; k < n_shards; k++) {
                uint64_t v[2] = {0, 0};
                if (!full_read(up[k], v, sizeof v)) { fprintf(stderr, "glimmer-mg_gpu: shard %d ended before reporting its counts\n", k); rc = EXIT_FAILURE; }
                gc[k] = v[0];
                total[k] = v[1];
            }
            const double job_gc = gmg_gc_fraction(gc.data(), total.data(), n_shards, 1);
            for (int k = 0; k < n_shards; k++) (void)full_write(down[k], &job_gc, sizeof job_gc);
        }
        for (int k = 0; k < n_shards; k++) {
            int status = 0;
            waitpid(pid[k], &status, 0);
            if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) { fprintf(stderr, "glimmer-mg_gpu: shard %d failed\n", k); rc = EXIT_FAILURE; }
        }
        if (rc != EXIT_SUCCESS) {                       // every child has been waited for: no part file survives a failed run
            for (int k = 0; k < n_shards; k++) {
                char part[32];
                snprintf(part, sizeof part, ".part%d", k);
                unlink((out + part).c_str());
            }
            return rc;
        }
        FILE *fo = File_Open(out, "w", __FILE__, __LINE__);      // the parts in shard order = the reads in file order
        vector<char> buf(1 << 20);
        for (int k = 0; k < n_shards; k++) {
            char part[32];
            snprintf(part, sizeof part, ".part%d", k);
            const string name = out + part;
            FILE *fi = File_Open(name, "r", __FILE__, __LINE__);
            size_t got;
            while ((got = fread(buf.data(), 1, buf.size(), fi)) > 0) fwrite(buf.data(), 1, got, fo);
            fclose(fi);
            unlink(name.c_str());
        }
        fclose(fo);
        return EXIT_SUCCESS;
    } catch (std::exception &e) {
        cerr << "** Standard Exception **" << endl << e << endl;
        return EXIT_FAILURE;
    }
}
