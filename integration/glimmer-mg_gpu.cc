// glimmer-mg_gpu.cc -- glimmer-mg with its front half on MI355X GPUs, one process per GPU.
//
// The reference's own glimmer-mg.cc is pulled in WHOLE from the reference tree at build time (main renamed; nothing
// is copied into this repository): option parsing, Add_Events_*, Process_Events, Trace_Back and the output format are
// the reference's code, unchanged.  What is replaced, for ALL reads of a batch in one call each:
//     Fasta_Read + tolower (Filter ()) + Set_GC_Fraction       -> gmg_fasta_ingest          (src/Common/fasta.cc:236-286,
//                                                                                             glimmer_base.cc:2564-2595)
//     Score_All_Frames + Find_Orfs + Score_Orfs_Errors          -> gmg_mg_score_reads        (glimmer-mg.cc:1468-1510,
//                                                                                             1605-1861; glimmer_base.cc:638-817)
// Both modes are driven: -m <icm> (one gene ICM, one null model) and -c <class file> (classification mode: the ICM, the null
// model and the stop codons follow every read's Phymm classes, reads are visited ICM by ICM -- glimmer-mg.cc:361-451 --;
// the bookkeeping is the library's gmg_classes_*, one gmg_mg_score_reads call per ICM group and stop-codon set), each with
// -i / -s / -q.
//
//     glimmer-mg_gpu [--shards N] [--gpus G] [--batch-bytes B] [--icm-dir DIR] [--chunk-reads N]
//                    <glimmer-mg options> <fasta> <tag>
//
// --icm-dir DIR   (-c) the Phymm .genomeData directory: the reference compiles it in as ICM_dir (glimmer-mg.cc:147,
//              patched by install_glimmer.py:121); default: $GMG_ICM_DIR.  The visiting order depends on the hash of the whole
//              file name, so the same string gives the same <tag>.predict as the reference built with it.
// --chunk-reads N (-c) Chunk_Sequences (glimmer-mg.cc:128; default 500000): every chunk of N reads is visited ICM by ICM.
// --shards N   the file is cut into N byte ranges at record starts (gmg_fasta_shard_ranges); N child processes are
//              forked BEFORE anything touches a GPU, child k binds to device k mod G, ingests and scores its range and
//              writes <tag>.predict.part<k>.  The one run-wide quantity, the null model's GC fraction
//              (Set_GC_Fraction: a ratio of two counts over the whole file), is summed by the parent from the children's
//              {gc, total} (two integers per child through a pipe) and handed back; the parent concatenates the parts
//              in shard order.  No collective, no GPU-to-GPU traffic.  With -q every child passes over the
//              quality records of the reads in front of its range and reads its own in order.
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

#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>
#include <spawn.h>
extern char **environ;

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
    if ((!User_ICM && classifications.empty()) || Detail_Log) {
        fprintf(stderr, "glimmer-mg_gpu: need -m <icm> or -c <class file>; the detail log is not driven here\n");
        exit(2);
    }
    if (!classifications.empty()) {                     // glimmer-mg.cc:268-279: the per-class feature models (host side, the reference's)
        if (!User_Length) Read_Meta_Lengths();
        if (!User_Start) Read_Meta_Starts();
        if (!User_Adj) { Read_Meta_AdjOr(); Read_Meta_AdjDist(); }
        if (!User_Stop) Read_Meta_Stops();
        if (!User_RBS) Read_Meta_RBS();                 // :313-316
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

// what one gmg_mg_score_reads call gives back, on the host
struct Scored {
    vector<gmg_mg_orf> orfs;
    vector<gmg_start> starts;
    vector<gmg_start_errors> errs;
    vector<uint64_t> read_orf_off;
};

static void fetch_result(gmg_mg_result *res, uint64_t n_reads, bool error_mode, Scored &sc);

static void score_batch(const gmg_model *gene, const gmg_model *nul, const gmg_reads *reads, uint64_t n_reads,
                        const gmg_mg_params &prm, bool error_mode, Scored &sc)
{
    gmg_mg_result *res = NULL;
    if (gmg_mg_score_reads(gene, nul, reads, &prm, NULL, &res, NULL) != GMG_OK) die_gmg("gmg_mg_score_reads");
    fetch_result(res, n_reads, error_mode, sc);
}

static void fetch_result(gmg_mg_result *res, uint64_t n_reads, bool error_mode, Scored &sc)
{
    uint64_t n_orfs = 0, n_starts = 0;
    gmg_mg_result_info(res, &n_orfs, &n_starts);
    sc.orfs.resize(n_orfs ? n_orfs : 1);
    sc.starts.resize(n_starts ? n_starts : 1);
    sc.read_orf_off.resize(n_reads + 1);
    sc.errs.resize(error_mode ? (n_starts ? n_starts : 1) : 0);
    if (gmg_mg_result_fetch(res, sc.orfs.data(), sc.starts.data(), sc.read_orf_off.data()) != GMG_OK) die_gmg("gmg_mg_result_fetch");
    if (error_mode && gmg_mg_result_fetch_errors(res, sc.errs.data()) != GMG_OK) die_gmg("gmg_mg_result_fetch_errors");
    gmg_mg_result_free(res);
}

// read i of a downloaded batch as glimmer-mg.cc:376-382 prepares it: the filtered lower-case sequence (back from the device)
static void load_sequence(const vector<uint32_t> &packed, const vector<uint64_t> &off, uint64_t i)
{
    Sequence.resize(off[i + 1] - off[i]);
    for (uint64_t k = 0; k < Sequence.size(); k++) {
        const uint64_t g = off[i] + k;
        Sequence[k] = "acgt"[(packed[g >> 4] >> (2 * (g & 15))) & 3];
    }
    Sequence_Len = Sequence.length();
}

// the reference's back half for the read in Fasta_Header / Sequence (glimmer-mg.cc:417-448): the accepted ORFs of read r of
// the scored batch go to Add_Events_* as Score_Orfs_Errors hands them over (:1656-1683), then the DP and the trace-back
static void back_half(FILE *predict_fp, const Scored &sc, uint64_t r, bool error_mode)
{
    Initialize_Terminal_Events(First_Event, Final_Event, Best_Event, Last_Event);
    Meta_PWM_Save.resize(2 * Sequence_Len);                                // glimmer-mg.cc:1622-1627
    for (unsigned int si = 0; si < 2 * Sequence_Len; si++) Meta_PWM_Save[si] = pair<double, int>(0.0, 999);
    int id = 0;
    for (uint64_t o = sc.read_orf_off[r]; o < sc.read_orf_off[r + 1]; o++) {
        const gmg_mg_orf &g = sc.orfs[o];
        if (!g.accepted) continue;
        Orf_t orf;
        orf.Set_Stop_Position(g.stop_position);
        orf.Set_Frame(g.frame);
        orf.Set_Gene_Len(g.gene_len);
        orf.Set_Orf_Len(g.orf_len);
        vector<Start_t> sl(g.n_starts);
        for (uint32_t s = 0; s < g.n_starts; s++) {
            const gmg_start &t = sc.starts[g.start_begin + s];
            sl[s].j = t.j; sl[s].pos = t.pos; sl[s].score = t.score; sl[s].rate = 0.0; sl[s].which = t.which;
            sl[s].truncated = t.truncated; sl[s].first = t.first;
            if (error_mode) {
                const gmg_start_errors &e = sc.errs[g.start_begin + s];
                for (int k = 0; k < e.n; k++) sl[s].errors.push_back(Error_t(e.pos[k], e.type[k]));
            }
        }
        std::sort(sl.begin(), sl.end(), Start_Cmp);                        // glimmer-mg.cc:1659: same algorithm on the same push order
        if (g.accepted == 2) {                                             // ties on pos: first_j is the sort's to decide (:1661-1666)
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

static void fill_params(gmg_mg_params &prm, bool error_mode)
{
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
}

struct Piece {
    gmg_reads *reads;
    uint64_t byte0, n_bytes, n_reads, total_bases;
    vector<uint64_t> hdr_begin, hdr_end;                // header extents, relative to the piece's first byte
};

// One shard: bytes [b0, b1) of the file.  up / down: pipes to / from the parent (-1: the shard is the whole job).
static int run_shard(const char *bytes, uint64_t b0, uint64_t b1, int device, uint64_t batch_bytes, int up, int down,
                     const string &out_name)
{
    // GMG_CLI_TIMING=1: where the wall time of this process goes, on stderr
    const bool cli_timing = getenv("GMG_CLI_TIMING") != NULL;
    struct CliClock {                                   // acc: device init, ingest, models, reads back to the host, scoring + fetch, events / DP / output
        struct timespec prev;
        double acc[6];
        CliClock() { clock_gettime(CLOCK_MONOTONIC, &prev); for (int k = 0; k < 6; k++) acc[k] = 0.0; }
        void lapse(int k)
        {
            struct timespec now;
            clock_gettime(CLOCK_MONOTONIC, &now);
            acc[k] += (double)(now.tv_sec - prev.tv_sec) + 1e-9 * (double)(now.tv_nsec - prev.tv_nsec);
            prev = now;
        }
    } clk;
    if (gmg_init(device) != GMG_OK) die_gmg("gmg_init");
    clk.lapse(0);
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
    clk.lapse(1);
    setup_models();
    clk.lapse(2);

    gmg_mg_params prm;
    fill_params(prm, error_mode);
    FILE *quality_fp = NULL;                            // -q: the values are read in file order, piece by piece
    if (Allow_Indels && Quality_File_Name != NULL) {
        quality_fp = File_Open(Quality_File_Name, "r", __FILE__, __LINE__);
        // a shard behind the first: the quality records of the reads in front of its byte range are passed over -- as many as Fasta_Read
        // (src/Common/fasta.cc:236-286) finds records in bytes [0, b0): a record begins at ANY '>' outside a header line (the rule
        // gmg_fasta_ingest follows as well), and they are passed over with the reference's own reader of the quality file
        uint64_t skip = 0;
        bool in_header = false;
        for (uint64_t i = 0; i < b0; i++) {
            if (in_header) in_header = bytes[i] != '\n';
            else if (bytes[i] == '>') { skip++; in_header = true; }
        }
        vector<int> q;
        string header;
        for (uint64_t k = 0; k < skip; k++)
            if (!Fasta_Qual_Vec_Read(quality_fp, q, header)) break;     // (fewer quality records than reads: the length check below reports it)
    }

    // pass 2: piece by piece -- one gmg_mg_score_reads call, then events / DP / trace-back per read on the host
    FILE *predict_fp = File_Open(out_name, "w", __FILE__, __LINE__);
    for (size_t p = 0; p < pieces.size(); p++) {
        Piece &pc = pieces[p];
        const int n_seq = (int)pc.n_reads;
        vector<uint64_t> off(pc.n_reads + 1);
        vector<uint32_t> packed(gmg_packed_words(pc.total_bases) + 1, 0);
        if (gmg_reads_download(pc.reads, packed.data(), off.data()) != GMG_OK) die_gmg("gmg_reads_download");
        clk.lapse(3);
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
        Scored sc;
        score_batch(Gene_ICM.Device_Model(), Indep_Model.Device_Model(), pc.reads, pc.n_reads, prm, error_mode, sc);
        gmg_reads_free(pc.reads);
        pc.reads = NULL;
        clk.lapse(4);

        string hdr;
        for (int i = 0; i < n_seq; i++) {
            hdr.assign(bytes + pc.byte0 + pc.hdr_begin[i], pc.hdr_end[i] - pc.hdr_begin[i]);
            Fasta_Header = hdr.c_str();
            load_sequence(packed, off, i);
            fprintf(predict_fp, ">%s\n", Fasta_Header);
            back_half(predict_fp, sc, i, error_mode);
        }
        clk.lapse(5);
    }
    fclose(predict_fp);
    if (quality_fp) fclose(quality_fp);
    if (cli_timing)
        fprintf(stderr, "glimmer-mg_gpu: device init %.3f s, ingest %.3f, models %.3f, reads back to the host %.3f, scoring + fetch %.3f, events / DP / output %.3f\n",
                clk.acc[0], clk.acc[1], clk.acc[2], clk.acc[3], clk.acc[4], clk.acc[5]);
    return EXIT_SUCCESS;
}

// ---- classification mode (-c) -------------------------------------------------------------------------------------------
// glimmer-mg.cc:326-451: the input is read in chunks of Chunk_Sequences reads; per chunk every ICM file is loaded in turn and
// scores the reads classified to it, each against the null model and the stop codons of ITS classes.  Here, per chunk:
// gmg_classes_plan gives the visiting order and every read's GC / translation table; the reads that share a stop-codon set are
// gathered on the device in visiting order (gmg_reads_select) and scored by ONE gmg_mg_score_groups call -- every ICM group a
// consecutive range of that batch under its own gene model, a null model per distinct GC (gmg_null_set_build,
// gmg_mg_params.read_null / read_ignore_score_len); the back half then runs read by read in the reference's order with the
// reference's own Update_Meta_* in between.  With -m AND -c (glimmer-mg.py's --long-orfs path)
// there is one group -- every read, in file order, under the user's ICM and the file's GC -- and only the stop codons vary.

static uint64_t next_record_start(const char *bytes, uint64_t from, uint64_t n_bytes)
{
    // the first '>' at or behind `from` that directly follows a newline: always a record start (Fasta_Read, fasta.cc:236-286)
    for (uint64_t q = from ? from : 1; q < n_bytes; q++) {
        const char *hit = (const char *)memchr(bytes + q, '>', n_bytes - q);
        if (hit == NULL) break;
        q = (uint64_t)(hit - bytes);
        if (bytes[q - 1] == '\n') return q;
    }
    return n_bytes;
}

// bytes [pos, *end) hold exactly `chunk_reads` reads (or all that are left): found by ingesting a guess and cutting at the
// header of read number chunk_reads
static void ingest_chunk(const char *bytes, uint64_t pos, uint64_t n_bytes, uint64_t chunk_reads, uint64_t guess_bytes, Piece &pc,
                         uint64_t &gc_out)
{
    uint64_t want = guess_bytes;
    for (;;) {
        uint64_t end = pos + want >= n_bytes ? n_bytes : next_record_start(bytes, pos + want, n_bytes);
        if (end - pos >= 0x7fffffffull) {
            fprintf(stderr, "glimmer-mg_gpu: a chunk of %llu reads exceeds 2 GiB of input; use a smaller --chunk-reads\n",
                    (unsigned long long)chunk_reads);
            exit(EXIT_FAILURE);
        }
        gmg_fasta *index = NULL;
        pc.byte0 = pos;
        if (gmg_fasta_ingest(bytes + pos, end - pos, &pc.reads, &index) != GMG_OK) die_gmg("gmg_fasta_ingest");
        gmg_fasta_info(index, &pc.n_reads, &pc.total_bases, &gc_out);
        pc.hdr_begin.resize(pc.n_reads);
        pc.hdr_end.resize(pc.n_reads);
        if (pc.n_reads) gmg_fasta_headers(index, pc.hdr_begin.data(), pc.hdr_end.data());
        gmg_fasta_free(index);
        if (pc.n_reads == chunk_reads || (pc.n_reads < chunk_reads && end == n_bytes)) { pc.n_bytes = end - pos; return; }
        if (pc.n_reads > chunk_reads) want = pc.hdr_begin[chunk_reads] - 1;    // the '>' of the first read that is one too many
        else want *= 2;
        gmg_reads_free(pc.reads);
        pc.reads = NULL;
        if (pc.n_reads > chunk_reads) {                 // exact now: [pos, pos + want) ends right in front of that '>'
            gmg_fasta *ix = NULL;
            if (gmg_fasta_ingest(bytes + pos, want, &pc.reads, &ix) != GMG_OK) die_gmg("gmg_fasta_ingest");
            gmg_fasta_info(ix, &pc.n_reads, &pc.total_bases, &gc_out);
            pc.hdr_begin.resize(pc.n_reads);
            pc.hdr_end.resize(pc.n_reads);
            if (pc.n_reads) gmg_fasta_headers(ix, pc.hdr_begin.data(), pc.hdr_end.data());
            gmg_fasta_free(ix);
            pc.n_bytes = want;
            return;
        }
    }
}

struct SubBatch {                                       // the reads of one ICM group that share a stop-codon set
    int code;                                           // translation table (0: the user's -z / -Z set)
    vector<uint64_t> member;                            // positions in the group's read list
    Scored sc;
};

// shard / n_shards: with --shards N every shard (one forked process per GPU) ingests the whole file (0.25 B/base per GPU and ~20 ms per
// 0.5 GB are cheap); the PLAN of a chunk -- the host-heavy part -- is made by shard 0 alone and read by the others from
// <out>.plan.c<chunk>; a shard takes the positions [n k / N, n (k + 1) / N) of every chunk's
// visiting order: reads are independent, so any consecutive run of the order is a valid piece of work; the pieces go to
// <out>.part<k>.c<chunk> and the parent concatenates them chunk by chunk, shard by shard: the reference's bytes.
static double wall_seconds()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int run_classes(const char *bytes, uint64_t n_bytes, int device, uint64_t batch_bytes, const char *class_file,
                       const string &icm_dir, const string &out_name, int shard, int n_shards)
{
    // GMG_CLI_TIMING=1: where the wall time of this process goes, one line on stderr at the end
    const bool timing = getenv("GMG_CLI_TIMING") != NULL;
    double t_sec[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // (6 .. 10: Update_Meta_RBS, _Length, _Start, _Adj, _Stop)
    //               // start-up, ingest, plan + device calls + fetch, Update_Meta_*, events / DP / trace-back, rest
    double t_mark = wall_seconds();
    const double t_begin = t_mark;
#define LAP(slot) do { if (timing) { const double t_ = wall_seconds(); t_sec[slot] += t_ - t_mark; t_mark = t_; } } while (0)
    if (gmg_init(device) != GMG_OK) die_gmg("gmg_init");
    const bool error_mode = Allow_Indels || Allow_Subs;
    const uint64_t chunk_reads = (uint64_t)Chunk_Sequences;

    // the class file once more, for the library's bookkeeping (the reference's own Parse_Classes filled `classifications`,
    // which its Update_Meta_* functions read)
    gmg_classes *cls = NULL;
    {
        FILE *fp = File_Open(class_file, "r", __FILE__, __LINE__);
        string text;
        char buf[1 << 16];
        size_t got;
        while ((got = fread(buf, 1, sizeof buf, fp)) > 0) text.append(buf, got);
        fclose(fp);
        if (gmg_classes_load(text.data(), text.size(), icm_dir.c_str(), &cls) != GMG_OK) die_gmg("gmg_classes_load");
    }
    uint32_t n_icms = 0;
    uint64_t missing_gc = 0;
    gmg_classes_info(cls, NULL, &n_icms, NULL, &missing_gc);
    if (missing_gc) fprintf(stderr, "WARNING: GC classification file unavailable for %llu classes (0.5 taken)\n", (unsigned long long)missing_gc);

    LAP(0);
    // every chunk onto the device (0.25 B/base stays resident); with -m the null model needs the whole file's GC first
    vector<Piece> chunks;
    uint64_t gc = 0, total = 0;
    for (uint64_t pos = 0; pos < n_bytes;) {
        Piece pc;
        uint64_t g = 0;
        ingest_chunk(bytes, pos, n_bytes, chunk_reads, batch_bytes, pc, g);
        gc += g;
        total += pc.total_bases;
        pos += pc.n_bytes;
        if (pc.n_reads == 0) { gmg_reads_free(pc.reads); continue; }
        chunks.push_back(pc);
    }
    if (User_ICM) {                                     // glimmer-mg.cc:283-296
        if (!GC_Frac_Set) { Indep_GC_Frac = gmg_gc_fraction(&gc, &total, 1, 1); GC_Frac_Set = true; }
        setup_models();
    } else if (User_RBS) {                              // :305-312 (-b without -m needs Indep_GC_Frac as well)
        if (!GC_Frac_Set) { Indep_GC_Frac = gmg_gc_fraction(&gc, &total, 1, 1); GC_Frac_Set = true; }
        LogOdds_PWM = Ribosome_PWM;
        LogOdds_PWM.Make_Log_Odds_WRT_GC(Indep_GC_Frac);
    }
    const double file_gc = Indep_GC_Frac;
    LAP(1);

    FILE *quality_fp = NULL;
    if (Allow_Indels && Quality_File_Name != NULL) quality_fp = File_Open(Quality_File_Name, "r", __FILE__, __LINE__);
    FILE *predict_fp = n_shards == 1 ? File_Open(out_name, "w", __FILE__, __LINE__) : NULL;
    vector<const char *> user_stops(Stop_Codon);         // -z / -Z: one set for every read
    std::map<string, ICM_t *> icm_cache;                // gene ICMs by file name, read once

    for (size_t c = 0; c < chunks.size(); c++) {
        Piece &pc = chunks[c];
        const uint64_t n = pc.n_reads;
        vector<uint64_t> off(n + 1);
        vector<uint32_t> packed(gmg_packed_words(pc.total_bases) + 1, 0);
        if (gmg_reads_download(pc.reads, packed.data(), off.data()) != GMG_OK) die_gmg("gmg_reads_download");
        vector<uint8_t> qual_all;                       // -q: the chunk's Phred values, reads back to back
        if (quality_fp) {
            qual_all.reserve(pc.total_bases);
            vector<int> q;
            string header;
            for (uint64_t i = 0; i < n; i++) {
                Fasta_Qual_Vec_Read(quality_fp, q, header);
                if (q.size() != off[i + 1] - off[i]) {
                    fprintf(stderr, "ERROR:  %s sequence length does not match quality values length\n", header.c_str());
                    return EXIT_FAILURE;
                }
                for (size_t k = 0; k < q.size(); k++) qual_all.push_back(q[k] > 255 ? 255 : q[k] < 0 ? 0 : q[k]);
            }
        }
        // the plan of the chunk
        vector<const char *> hdr(n);
        vector<uint32_t> hdr_len(n);
        for (uint64_t i = 0; i < n; i++) { hdr[i] = bytes + pc.byte0 + pc.hdr_begin[i]; hdr_len[i] = (uint32_t)(pc.hdr_end[i] - pc.hdr_begin[i]); }
        vector<uint64_t> order(n ? n : 1), icm_begin(n_icms + 1);
        vector<double> read_gc(n ? n : 1);
        vector<int32_t> read_tt(n ? n : 1);
        uint64_t n_order = 0;
        // --shards: shard 0 plans the chunk (the host-heavy part: a hash look-up and the class bookkeeping per read) and leaves the
        // plan in <out>.plan.c<chunk>; the other shards wait for that file instead of planning the same chunk again
        char plan_tag[64];
        snprintf(plan_tag, sizeof plan_tag, ".plan.c%zu", c);
        const string plan_name = out_name + plan_tag;
        if (n_shards > 1 && shard != 0) {
            FILE *fp = NULL;
            for (int waited_ms = 0; (fp = fopen(plan_name.c_str(), "rb")) == NULL; waited_ms++) {
                if (waited_ms > 3600 * 1000 || getppid() == 1) { fprintf(stderr, "glimmer-mg_gpu: shard %d: no plan of chunk %zu from shard 0\n", shard, c); return EXIT_FAILURE; }
                usleep(1000);
            }
            uint64_t hd[3] = {0, 0, 0};
            bool ok = fread(hd, 8, 3, fp) == 3 && hd[0] == n && hd[1] == n_icms && hd[2] <= n;
            n_order = hd[2];
            ok = ok && fread(order.data(), 8, n ? n : 1, fp) == (n ? n : 1) && fread(icm_begin.data(), 8, n_icms + 1, fp) == n_icms + 1 &&
                 fread(read_gc.data(), 8, n ? n : 1, fp) == (n ? n : 1) && fread(read_tt.data(), 4, n ? n : 1, fp) == (n ? n : 1);
            fclose(fp);
            if (!ok) { fprintf(stderr, "glimmer-mg_gpu: shard %d: the plan of chunk %zu does not fit this chunk\n", shard, c); return EXIT_FAILURE; }
        } else {
            if (gmg_classes_plan(cls, hdr.data(), hdr_len.data(), n, order.data(), icm_begin.data(), read_gc.data(), read_tt.data(), &n_order) != GMG_OK)
                die_gmg("gmg_classes_plan");
            if (n_shards > 1) {                         // written under another name and renamed: a reader sees the whole file or none
                const string tmp = plan_name + ".tmp";
                FILE *fp = File_Open(tmp, "wb", __FILE__, __LINE__);
                const uint64_t hd[3] = {n, n_icms, n_order};
                fwrite(hd, 8, 3, fp);
                fwrite(order.data(), 8, n ? n : 1, fp); fwrite(icm_begin.data(), 8, n_icms + 1, fp);
                fwrite(read_gc.data(), 8, n ? n : 1, fp); fwrite(read_tt.data(), 4, n ? n : 1, fp);
                if (fclose(fp) != 0 || rename(tmp.c_str(), plan_name.c_str()) != 0) { perror("glimmer-mg_gpu: writing the chunk's plan"); return EXIT_FAILURE; }
            }
        }
        uint32_t n_groups = n_icms;
        if (User_ICM) {                                 // one group: every read in file order; classes give the stop codons only
            vector<int32_t> tt_of(n, -1);
            for (uint64_t k = 0; k < n_order; k++) tt_of[order[k]] = read_tt[k];
            for (uint64_t i = 0; i < n; i++) {
                if (tt_of[i] < 0) {                     // the reference indexes the empty class list of such a read (glimmer-mg.cc:2196)
                    fprintf(stderr, "glimmer-mg_gpu: read %.*s has no line in the classification file (-m with -c needs one for every read)\n",
                            (int)hdr_len[i], hdr[i]);
                    return EXIT_FAILURE;
                }
                order[i] = i;
                read_tt[i] = tt_of[i];
                read_gc[i] = file_gc;
            }
            n_groups = 1;
            icm_begin.assign(2, 0);
            icm_begin[1] = n;
        }

        const uint64_t n_proc = icm_begin[n_groups];
        // this shard's run of the chunk's visiting order
        const uint64_t k_lo = n_shards == 1 ? 0 : n_proc * (uint64_t)shard / n_shards, k_hi = n_shards == 1 ? n_proc : n_proc * (uint64_t)(shard + 1) / n_shards;
        if (n_shards > 1) {
            char part[64];
            snprintf(part, sizeof part, ".part%d.c%zu", shard, c);
            predict_fp = File_Open(out_name + part, "w", __FILE__, __LINE__);
        }
        // the gene ICMs of the groups that have reads in this chunk (the reference reads every ICM of the class file for every
        // chunk, used or not, and again for the next chunk, glimmer-mg.cc:364; here a file is read once)
        vector<const gmg_model *> group_model(n_groups, (const gmg_model *)NULL);
        for (uint32_t f = 0; f < n_groups; f++) {
            if (icm_begin[f + 1] <= k_lo || icm_begin[f] >= k_hi || icm_begin[f + 1] == icm_begin[f]) continue;      // no read of this shard
            if (User_ICM) { group_model[f] = Gene_ICM.Device_Model(); continue; }
            const string name = gmg_classes_icm_file(cls, f);
            std::map<string, ICM_t *>::iterator it = icm_cache.find(name);
            if (it == icm_cache.end()) {
                ICM_t *m = new ICM_t();
                m->Read((char *)name.c_str());
                it = icm_cache.insert(make_pair(name, m)).first;
            }
            group_model[f] = it->second->Device_Model();
        }
        // one gmg_mg_score_groups call per stop-codon set: its reads in visiting order, its ICM groups as consecutive ranges
        vector<SubBatch> sub;
        vector<pair<uint32_t, uint64_t> > where(n_proc);                // visiting position -> (sub-batch, index in it)
        for (uint64_t k = k_lo; k < k_hi; k++) {
            const int code = User_Stop ? 0 : read_tt[k];
            size_t b = 0;
            while (b < sub.size() && sub[b].code != code) b++;
            if (b == sub.size()) { sub.push_back(SubBatch()); sub[b].code = code; }
            where[k] = make_pair((uint32_t)b, (uint64_t)sub[b].member.size());
            sub[b].member.push_back(k);
        }
        for (size_t b = 0; b < sub.size(); b++) {
            SubBatch &sb = sub[b];
            const uint64_t m = sb.member.size();
            char stops[8][4];
            int n_stops = 0;
            memset(stops, 0, sizeof stops);
            if (User_Stop) {
                n_stops = (int)user_stops.size();
                for (int t = 0; t < n_stops && t < 8; t++) memcpy(stops[t], user_stops[t], 3);
            } else if (gmg_stop_codons_by_code(sb.code, stops, &n_stops) != GMG_OK) {
                fprintf(stderr, "%s\n", gmg_last_error());           // Set_Stop_Codons_By_Code's message (gene.cc:1618)
                return EXIT_FAILURE;
            }
            // a null model per distinct GC, Ignore_Score_Len per read (Update_Meta_Null_ICM, glimmer-mg.cc:2050-2068)
            vector<double> gcs;
            vector<uint32_t> read_null(m);
            vector<int32_t> read_isl(m);
            vector<uint64_t> idx(m);
            vector<gmg_mg_group> groups;
            std::map<uint64_t, uint32_t> seen;
            uint32_t f = 0, f_last = 0xffffffffu;
            for (uint64_t r = 0; r < m; r++) {
                const uint64_t k = sb.member[r];
                while (icm_begin[f + 1] <= k) f++;                  // the ICM group of visiting position k
                if (f != f_last) {
                    gmg_mg_group g;
                    g.gene = group_model[f];
                    g.read_begin = r;
                    g.read_end = r;
                    groups.push_back(g);
                    f_last = f;
                }
                groups.back().read_end = r + 1;
                idx[r] = order[k];
                uint64_t bits;
                memcpy(&bits, &read_gc[k], 8);
                std::map<uint64_t, uint32_t>::iterator it = seen.find(bits);
                if (it == seen.end()) { it = seen.insert(make_pair(bits, (uint32_t)gcs.size())).first; gcs.push_back(read_gc[k]); }
                read_null[r] = it->second;
                if (gmg_ignore_score_len(read_gc[k], stops, n_stops, &read_isl[r]) != GMG_OK) die_gmg("gmg_ignore_score_len");
            }
            gmg_null_set *nulls = NULL;
            if (gmg_null_set_build(gcs.data(), (int)gcs.size(), stops, n_stops, &nulls) != GMG_OK) die_gmg("gmg_null_set_build");
            gmg_reads *batch = NULL;
            if (gmg_reads_select(pc.reads, idx.data(), m, &batch) != GMG_OK) die_gmg("gmg_reads_select");
            gmg_mg_params prm;
            fill_params(prm, error_mode);
            prm.n_stop_codons = n_stops;
            memcpy(prm.stop_codon, stops, sizeof stops);
            prm.nulls = nulls;
            prm.read_null = read_null.data();
            prm.read_ignore_score_len = read_isl.data();
            vector<uint8_t> qual;
            if (quality_fp) {
                for (uint64_t r = 0; r < m; r++) qual.insert(qual.end(), qual_all.begin() + off[idx[r]], qual_all.begin() + off[idx[r] + 1]);
                prm.quality = qual.data();
            }
            ICM_t any_null(3, 2, 3);                    // the null_model argument is not read when params.nulls is set, but must be a model
            {
                vector<const char *> sv;
                for (int t = 0; t < n_stops; t++) sv.push_back(stops[t]);
                any_null.Build_Indep_WO_Stops(gcs[0], sv);
            }
            gmg_mg_result *res = NULL;
            if (gmg_mg_score_groups(groups.data(), (int)groups.size(), any_null.Device_Model(), batch, &prm, &res, NULL) != GMG_OK)
                die_gmg("gmg_mg_score_groups");
            fetch_result(res, m, error_mode, sb.sc);
            gmg_reads_free(batch);
            gmg_null_set_free(nulls);
        }
        LAP(2);
        // the back half, read by read in the reference's order (glimmer-mg.cc:367-450)
        string hs;
        for (uint64_t k = k_lo; k < k_hi; k++) {
            const uint64_t i = order[k];
            hs.assign(hdr[i], hdr_len[i]);
            Fasta_Header = hs.c_str();
            load_sequence(packed, off, i);
            fprintf(predict_fp, ">%s\n", Fasta_Header);
            LAP(5);
            if (!User_RBS) Update_Meta_RBS();
            LAP(6);
            if (!User_Length) Update_Meta_Length();
            LAP(7);
            if (!User_Start) Update_Meta_Start();
            LAP(8);
            if (!User_Adj) Update_Meta_Adj();
            LAP(9);
            if (!User_Stop) Update_Meta_Stop();
            LAP(10);
            if (!User_ICM) {                            // what Update_Meta_Null_ICM leaves in the globals the back half reads
                Indep_GC_Frac = read_gc[k];
                int32_t isl = 0;
                char st[8][4];
                int ns = 0;
                memset(st, 0, sizeof st);
                for (size_t t = 0; t < Stop_Codon.size() && t < 8; t++, ns++) memcpy(st[t], Stop_Codon[t], 3);
                gmg_ignore_score_len(read_gc[k], st, ns, &isl);
                Ignore_Score_Len = isl;
            }
            LAP(3);
            back_half(predict_fp, sub[where[k].first].sc, where[k].second, error_mode);
            LAP(4);
        }
        gmg_reads_free(pc.reads);
        pc.reads = NULL;
        if (n_shards > 1) { fclose(predict_fp); predict_fp = NULL; }
    }
    if (predict_fp) fclose(predict_fp);
    if (quality_fp) fclose(quality_fp);
    gmg_classes_free(cls);
    LAP(5);
    if (timing)
        fprintf(stderr, "glimmer-mg_gpu timing (shard %d of %d): start-up %.3f s, ingest %.3f, plan + device calls + fetch %.3f, Update_Meta_* %.3f "
                        "(RBS %.3f, Length %.3f, Start %.3f, Adj %.3f, Stop %.3f), events / DP / trace-back %.3f, rest %.3f; total %.3f\n", shard, n_shards,
                t_sec[0], t_sec[1], t_sec[2], t_sec[3] + t_sec[6] + t_sec[7] + t_sec[8] + t_sec[9] + t_sec[10], t_sec[6], t_sec[7], t_sec[8], t_sec[9], t_sec[10],
                t_sec[4], t_sec[5], wall_seconds() - t_begin);
#undef LAP
    return EXIT_SUCCESS;
}

int main(int argc, char **argv)
{
    int n_shards = 1, n_gpus = 1;
    uint64_t batch_bytes = DEFAULT_BATCH_BYTES;
    string icm_dir = getenv("GMG_ICM_DIR") ? getenv("GMG_ICM_DIR") : "";
    if (const char *e = getenv("GMG_GPUS")) n_gpus = atoi(e);
    // our own options come first; the rest is glimmer-mg's command line, untouched
    vector<char *> rest(1, argv[0]);
    int a = 1;
    for (; a + 1 < argc; a += 2) {
        if (strcmp(argv[a], "--shards") == 0) n_shards = atoi(argv[a + 1]);
        else if (strcmp(argv[a], "--gpus") == 0) n_gpus = atoi(argv[a + 1]);
        else if (strcmp(argv[a], "--batch-bytes") == 0) batch_bytes = strtoull(argv[a + 1], NULL, 10);
        else if (strcmp(argv[a], "--icm-dir") == 0) icm_dir = argv[a + 1];
        else if (strcmp(argv[a], "--chunk-reads") == 0) Chunk_Sequences = atoi(argv[a + 1]);
        else break;
    }
    for (; a < argc; a++) rest.push_back(argv[a]);
    if (rest.size() < 3 || n_shards < 1 || n_gpus < 1 || batch_bytes == 0 || batch_bytes >= 0x7fffffffull) {
        fprintf(stderr, "usage: glimmer-mg_gpu [--shards N] [--gpus G] [--batch-bytes B < 2^31] <glimmer-mg options> <fasta> <tag>\n");
        return 2;
    }
    if (Chunk_Sequences < 1) { fprintf(stderr, "glimmer-mg_gpu: --chunk-reads must be positive\n"); return 2; }
    // -c <file> / -c<file> / --class <file> / --class=<file> (getopt_long, glimmer-mg.cc:777-801): the library parses the file too
    const char *class_file = NULL;
    for (size_t k = 1; k < rest.size(); k++) {
        const char *w = rest[k];
        if (strcmp(w, "--") == 0) break;
        if (strncmp(w, "--class=", 8) == 0) class_file = w + 8;
        else if (strcmp(w, "--class") == 0 && k + 1 < rest.size()) class_file = rest[++k];
        else if (w[0] == '-' && w[1] != '-' && w[1] != 0) {
            for (const char *o = w + 1; *o; o++) {
                if (strchr("bcfgmoPquzZ", *o) == NULL) continue;        // a flag without an argument: the next letter
                const char *val = o[1] ? o + 1 : (k + 1 < rest.size() ? rest[++k] : NULL);
                if (*o == 'c') class_file = val;
                break;
            }
        }
    }
    try {
        if (class_file != NULL) {
            if (icm_dir.empty()) { fprintf(stderr, "glimmer-mg_gpu: -c needs --icm-dir DIR (or GMG_ICM_DIR): the Phymm .genomeData directory\n"); return 2; }
            ICM_dir = icm_dir;                          // glimmer-mg.cc:147: what the reference's Read_Meta_* open
        }
        setup_options((int)rest.size(), rest.data());
        if (Genome_Is_Circular) {
            // -r (Find_Orfs with wrap-around, glimmer_base.cc:638-817) is not batched here: glimmer-mg_dropin beside this binary is the
            // reference's own main() on the device-backed ICM_t (same bytes, one launch per ICM_t call).  It runs as a CHILD
            // (posix_spawn + waitpid): under a profiler this process may have initialised the GPU before main(), and exec* from such
            // a process is what this pool forbids.  Not with -c: the drop-in is the reference's main() with ITS compiled-in ICM_dir
            // (glimmer-mg.cc:147), so --icm-dir could not reach it.
            if (class_file != NULL) {
                fprintf(stderr, "glimmer-mg_gpu: -r together with -c is not supported (circular sequences run in glimmer-mg_dropin, which has the reference's compiled-in ICM directory)\n");
                return 2;
            }
            char self[4096];
            const ssize_t n_self = readlink("/proc/self/exe", self, sizeof self - 1);
            string dir = n_self > 0 ? string(self, (size_t)n_self) : string(argv[0]);
            const size_t slash = dir.rfind('/');
            const string exe = (slash == string::npos ? string(".") : dir.substr(0, slash)) + "/glimmer-mg_dropin";
            rest.push_back(NULL);
            pid_t pid = 0;
            const int src = posix_spawn(&pid, exe.c_str(), NULL, NULL, rest.data(), environ);
            if (src != 0) {
                fprintf(stderr, "glimmer-mg_gpu: -r needs %s (the reference's loop on the device-backed ICM_t), which could not be started: %s\n", exe.c_str(), strerror(src));
                return 2;
            }
            int status = 0;
            while (waitpid(pid, &status, 0) < 0)
                if (errno != EINTR) { perror("glimmer-mg_gpu: waitpid"); return 2; }
            return WIFEXITED(status) ? WEXITSTATUS(status) : 128 + (WIFSIGNALED(status) ? WTERMSIG(status) : 0);
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
        if (!classifications.empty()) {
            if (n_shards == 1) return run_classes(bytes, n_bytes, env_dev, batch_bytes, class_file, icm_dir, out, 0, 1);
            vector<pid_t> cpid(n_shards);
            for (int k = 0; k < n_shards; k++) {
                fflush(NULL);
                cpid[k] = fork();                       // nothing in this process has touched a GPU
                if (cpid[k] < 0) { perror("fork"); return EXIT_FAILURE; }
                if (cpid[k] == 0) {
                    const int rc = run_classes(bytes, n_bytes, (env_dev + k) % n_gpus, batch_bytes, class_file, icm_dir, out, k, n_shards);
                    fflush(NULL);
                    _exit(rc);
                }
            }
            int rc = EXIT_SUCCESS;
            for (int k = 0; k < n_shards; k++) {
                int status = 0;
                waitpid(cpid[k], &status, 0);
                if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) { fprintf(stderr, "glimmer-mg_gpu: shard %d failed\n", k); rc = EXIT_FAILURE; }
            }
            // chunk by chunk, shard by shard (every shard wrote a part for every chunk, empty ones included)
            FILE *fo = rc == EXIT_SUCCESS ? File_Open(out, "w", __FILE__, __LINE__) : NULL;
            vector<char> buf(1 << 20);
            for (size_t c = 0;; c++) {
                bool any = false;
                for (int k = 0; k < n_shards; k++) {
                    char part[64];
                    snprintf(part, sizeof part, ".part%d.c%zu", k, c);
                    const string name = out + part;
                    FILE *fi = fopen(name.c_str(), "r");
                    if (fi == NULL) continue;
                    any = true;
                    size_t got;
                    while (fo && (got = fread(buf.data(), 1, buf.size(), fi)) > 0) fwrite(buf.data(), 1, got, fo);
                    fclose(fi);
                    unlink(name.c_str());
                }
                char plan_tag[64];
                snprintf(plan_tag, sizeof plan_tag, ".plan.c%zu", c);
                unlink((out + plan_tag).c_str());       // (shard 0's plan of the chunk, read by the other shards)
                if (!any) break;
            }
            if (fo) fclose(fo);
            return rc;
        }
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
