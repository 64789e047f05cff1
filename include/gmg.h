/*
 * gmg.h -- C ABI of the MI355X (gfx950) IMM scorer for Glimmer-MG.
 *
 * This is the drop-in boundary: a thin extern "C" layer over hand-written HIP
 * kernels.  Plain pointers and sizes only; no C++ or torch types.  The host
 * side (glimmer-mg_amd/host/icm.hh, an ICM_t with the reference's public
 * interface) and any FFI binding call exactly these entry points.
 *
 * Each entry point cites the reference interface it replaces
 * (paths relative to the reference root, davek44/Glimmer-MG).
 *
 * Conventions
 *   - every function returns 0 (GMG_OK) or a negative gmg_status; it never
 *     calls exit().  gmg_last_error() gives the text for the calling thread.
 *   - "d_" pointers are device (HBM) pointers owned by the caller (hipMalloc /
 *     torch); all other pointers are host memory, caller-owned.
 *   - handles (gmg_model, gmg_reads, gmg_segments) own their device memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *     launches are stream-ordered and asynchronous.
 *   - base code: a=0 c=1 g=2 t=3 (ALPHA_STRING, src/ICM/icm.hh:30), complement
 *     = 3 - code.  Packed reads hold 16 bases per uint32, base g of the job at
 *     bits [2*(g%16), 2*(g%16)+1] of word g/16, reads concatenated without
 *     padding; base_offsets[r] is the job-wide index of read r's first base
 *     (n_reads+1 entries).
 */
#ifndef GMG_H
#define GMG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gmg_status {
    GMG_OK = 0,
    GMG_EINVAL = -1,     /* bad argument (NULL handle, frame out of range, ...)          */
    GMG_ENODEV = -2,     /* no usable gfx950 device / HIP runtime failure at init         */
    GMG_ENOMEM = -3,     /* host or device allocation failed                              */
    GMG_EHIP = -4,       /* a HIP call failed; text in gmg_last_error()                   */
    GMG_EBADMODEL = -5,  /* model parameters outside what the kernels support             */
    GMG_ERANGE = -6,     /* a segment / window reaches outside its read                   */
    GMG_ETOOBIG = -7     /* the batch would hold more ORFs or starts than the 32-bit index fields of the result
                            records address (gmg_mg_orf.start_begin, gmg_orf_result.start_begin): nothing wraps,
                            the call refuses -- score the reads in smaller batches (gmg_shard_plan)              */
} gmg_status;

typedef struct gmg_model gmg_model;       /* device copy of one ICM_t table set  */
typedef struct gmg_reads gmg_reads;       /* device copy of a batch of reads     */
typedef struct gmg_segments gmg_segments; /* device copy of a list of segments   */

/* A scoring buffer cut from a read, in the orientation the reference scores it
 * (src/Glimmer/glimmer3.cc:1322-1343, src/Glimmer/glimmer-mg.cc:1482,1497):
 *   GMG_REVERSED     B[j] = S[lo+len-1-j]        Reverse_Transfer     (glimmer_base.cc:2505-2533)
 *   GMG_COMPLEMENTED B[j] = comp(S[lo+j])        Complement_Transfer  (glimmer_base.cc:410-434)
 *   GMG_FORWARD      B[j] = S[lo+j]              the string as given  (Score_String on a read)
 *   GMG_REVCOMP      B[j] = comp(S[lo+len-1-j])  Reverse_Complement_Transfer (glimmer_base.cc:2484-2501)
 * Context never crosses the start of the buffer: the first model_len-1 bases of
 * every segment use the partial-window rule (src/ICM/icm.cc:807-842). */
typedef enum gmg_orient {
    GMG_FORWARD = 0,
    GMG_REVERSED = 1,
    GMG_COMPLEMENTED = 2,
    GMG_REVCOMP = 3
} gmg_orient;

typedef struct gmg_segment {
    uint32_t read;    /* index into the gmg_reads batch            */
    uint32_t lo;      /* first base of the region, 0-based in read */
    uint32_t len;     /* number of bases                           */
    uint32_t orient;  /* gmg_orient                                */
} gmg_segment;

/* ---- library / device ---------------------------------------------------- */

/* Bind the calling process to HIP device `device` (one process per GPU).  Must
 * be called before anything else.  Fails with GMG_ENODEV when no GPU or the
 * device is not gfx950 -- there is no CPU fallback. */
int gmg_init(int device);
int gmg_device_count(void);
const char *gmg_last_error(void);
const char *gmg_version(void);
int gmg_synchronize(void *stream);
/* Tuning and test switches (nothing a caller needs for correct results): `key` is one of seg_plain, mg_tile, mg_one_stream,
 * mg_err_flat, mg_err_calls, mg_err_calls_grow, orfs_exact_path, train_sort_min, mg_max_entries, mg_timing, ingest_timing,
 * train_timing, strings_fused, mg_gene32, mg_fused, mg_err_skip, mg_orfs_events, diag.  gmg_init() reads GMG_<KEY IN UPPER CASE> from the
 * environment ONCE; no scoring call reads the environment. */
int gmg_set_option(const char *key, long long value);
int gmg_get_option(const char *key, long long *value);

/* ---- host-side packing helpers (no GPU needed) ---------------------------- */

/* 2-bit code of one character after the reference's load-time normalisation
 * tolower(Filter(ch)) (src/Glimmer/glimmer3.cc:270-271, src/Common/gene.cc:1139-1175)
 * followed by Subscript (src/ICM/icm.cc:2008-2027). */
int gmg_base_code(int ch);
/* Pack n characters starting at job-wide base index `first_base` into `packed`
 * (which must be zero-initialised or already hold the preceding bases). */
int gmg_pack_bases(const char *ascii, uint64_t n, uint64_t first_base, uint32_t *packed);
/* Number of uint32 words needed for total_bases bases (+ one guard word). */
uint64_t gmg_packed_words(uint64_t total_bases);

/* ---- models: replaces ICM_t's table (src/ICM/icm.hh:106-129) -------------- */

/* mip[p*num_nodes+n], prob4[(p*num_nodes+n)*4+b] exactly as ICM_t::Input builds
 * them (src/ICM/icm.cc:614-726) or Build_Indep_WO_Stops (icm.cc:65-216).
 * Supported: model_len <= 32, mut_info_pos in [-2, model_len-1].
 * The fast six-frame kernel additionally needs model_len <= 16, model_depth <= 8;
 * other shapes take the generic kernels (same results). */
int gmg_model_upload(const int16_t *mip, const float *prob4, int model_len, int model_depth,
                     int periodicity, int num_nodes, gmg_model **out);
int gmg_model_free(gmg_model *m);
int gmg_model_info(const gmg_model *m, int *model_len, int *model_depth, int *periodicity,
                   int *num_nodes);

/* ---- reads ------------------------------------------------------------------ */

int gmg_reads_upload(const uint32_t *packed2bit, const uint64_t *base_offsets, uint64_t n_reads,
                     gmg_reads **out);
/* Take packed reads that are ALREADY in HBM (e.g. generated on device): one
 * device-to-device copy into the library's guarded buffer; the offsets are used
 * in place and must stay valid until gmg_reads_free.  Neither buffer is freed. */
int gmg_reads_wrap_device(const uint32_t *d_packed2bit, const uint64_t *d_base_offsets,
                          uint64_t n_reads, uint64_t total_bases, gmg_reads **out);
int gmg_reads_free(gmg_reads *r);
int gmg_reads_info(const gmg_reads *r, uint64_t *n_reads, uint64_t *total_bases);
/* Copies the batch back to HOST buffers: packed2bit[gmg_packed_words(total_bases)], base_offsets[n_reads + 1]. */
int gmg_reads_download(const gmg_reads *reads, uint32_t *packed2bit, uint64_t *base_offsets);
/* A new batch made of reads idx[0..n) of `reads` (host indices; any order, repeats allowed), gathered on the device:
 * the grouping step of glimmer-mg's classification mode, where every ICM scores the reads classified to it with the
 * null model of their classes (src/Glimmer/glimmer-mg.cc:361-375, 2050-2068) -- one gmg_mg_score_reads call per group. */
int gmg_reads_select(const gmg_reads *reads, const uint64_t *idx, uint64_t n, gmg_reads **out);

/* ---- one string at a time ------------------------------------------------------
 * The ICM_t methods take ONE string per call (src/ICM/icm.hh:131-180).  A gmg_single keeps what such a call needs --
 * page-locked host staging, the packed read, its one segment and a result buffer on the device -- from call to call, so
 * that a call is one copy in, one launch and one copy out (no allocation).  One per host thread; glimmer-mg_amd/host/icm.cc
 * holds one per thread.  gmg_single_stage: the string `ascii` of n characters as a one-read batch with the whole read as
 * its segment in orientation `orient`; *reads / *segs are valid until the next stage call; *d_out has room for n + 16
 * doubles.  gmg_single_fetch copies n_doubles of it back (after the stream's work). */
typedef struct gmg_single gmg_single;
int gmg_single_create(gmg_single **out);
int gmg_single_stage(gmg_single *st, const char *ascii, uint64_t n, int orient, const gmg_reads **reads,
                     const gmg_segments **segs, double **d_out);
int gmg_single_fetch(gmg_single *st, double *dst, size_t n_doubles);
int gmg_single_free(gmg_single *st);
/* ICM_t::Full_Window_Prob / Full_Window_Distrib (src/ICM/icm.cc:512-610) for ONE window of model_len codes (0..3, one byte each)
 * under sub-model `frame`, on the staging of st: dist4 (4 floats, may be NULL), prob (may be NULL). */
int gmg_single_window(gmg_single *st, const gmg_model *m, const uint8_t *codes, int model_len, int frame, float *dist4, double *prob);

/* ---- segments --------------------------------------------------------------- */

/* Validates every segment against the read lengths (GMG_ERANGE otherwise).
 * out_total_len receives sum(len); per-position outputs of segment i start at
 * the exclusive prefix sum of the lengths (also returned in out_offsets if not NULL,
 * n_segments+1 entries). */
int gmg_segments_upload(const gmg_reads *reads, const gmg_segment *segs, uint64_t n_segments,
                        uint64_t *out_offsets, uint64_t *out_total_len, gmg_segments **out);
int gmg_segments_free(gmg_segments *s);

/* ---- scoring ---------------------------------------------------------------- */

/* Six-frame per-position scores of whole reads: replaces Score_All_Frames
 * (src/Glimmer/glimmer-mg.cc:1468-1510), i.e. 12 x ICM_t::Frame_Score
 * (src/ICM/icm.cc:485-509) + the gene - null subtraction in double.
 *   d_out[f*total_bases + base_offsets[r] + p],  f = 0..5, p = 0..len(r)-1
 * equals Frame_Scores[f][p] of read r bit for bit.  Both models need periodicity
 * >= 3 (Frame_Score asserts frame < periodicity, src/ICM/icm.cc:496); `null_model`
 * normally is the (3,2,3) Build_Indep_WO_Stops model. */
int gmg_frame_score6(const gmg_model *gene, const gmg_model *null_model, const gmg_reads *reads,
                     double *d_out, void *stream);
/* The same table with its six rows row_stride (>= total_bases) doubles apart:
 *   d_out[f*row_stride + base_offsets[r] + p].
 * The kernel stores two doubles per lane when every row starts on a 16-byte boundary (d_out 16-byte aligned and
 * row_stride even), one at a time otherwise (about 15% slower); a batch of ragged reads has an odd total_bases half of
 * the time, so a caller that owns the table should round the stride up (gmg_mg_score_reads does for its own table). */
int gmg_frame_score6_strided(const gmg_model *gene, const gmg_model *null_model, const gmg_reads *reads,
                             double *d_out, uint64_t row_stride, void *stream);

/* Per-read null models: a set of (3,2,3) Build_Indep_WO_Stops models on the device (1 KB each), and the six-frame table
 * with read r scored against null model read_null[r] (HOST array, n_reads entries) -- what Score_All_Frames gives inside
 * glimmer-mg's classification loop, where Update_Meta_Null_ICM (glimmer-mg.cc:2050-2068) rebuilds Indep_Model per read.
 * (Two passes: fp32 gene values, then the null model's part; gmg_mg_score_reads with gmg_mg_params.nulls applies the
 * null models where it builds its running sums, at no extra pass.) */
typedef struct gmg_null_set gmg_null_set;
int gmg_null_set_upload(const gmg_model *const *null_models, int n_models, gmg_null_set **out);
/* The same set from HOST tables: mip[n][3][21], prob4[n][3][21][4] as Build_Indep_WO_Stops fills (3,2,3) models; one copy. */
int gmg_null_set_from_tables(const int16_t *mip, const float *prob4, int n_models, gmg_null_set **out);
/* ... or from the GC values themselves: model i = Build_Indep_WO_Stops (gc_frac[i], stop_codon) (src/ICM/icm.cc:65-216,
 * the library's host ICM_t) -- what Update_Meta_Null_ICM (glimmer-mg.cc:2050-2068) builds per read. */
int gmg_null_set_build(const double *gc_frac, int n_models, const char (*stop_codon)[4], int n_stop_codons,
                       gmg_null_set **out);
int gmg_null_set_free(gmg_null_set *nulls);
int gmg_frame_score6_nulls(const gmg_model *gene, const gmg_null_set *nulls, const uint32_t *read_null,
                           const gmg_reads *reads, double *d_out, uint64_t row_stride, void *stream);

/* ICM_t::Frame_Score (src/ICM/icm.cc:485-509) on every segment: one fixed
 * sub-model `frame` for all positions, no sum.  d_out[offset(i)+j]. */
int gmg_segment_frame_score(const gmg_model *m, const gmg_reads *reads, const gmg_segments *segs,
                            int frame, double *d_out, void *stream);

/* ICM_t::Cumulative_Score (src/ICM/icm.cc:354-405) on every segment: running
 * double sum in reference order, sub-model of base 0 = frame0, cycling.
 * This is what Score_Orfs calls twice per ORF (src/Glimmer/glimmer3.cc:1346-1347). */
int gmg_segment_cumscore(const gmg_model *m, const gmg_reads *reads, const gmg_segments *segs,
                         int frame0, double *d_out, void *stream);

/* ICM_t::Score_String (src/ICM/icm.cc:864-903) on every segment: d_sums[i]. */
int gmg_score_string(const gmg_model *m, const gmg_reads *reads, const gmg_segments *segs,
                     int frame0, double *d_sums, void *stream);

/* Whole-read ICM_t::Score_String (src/ICM/icm.cc:864-903, frame 0) of every read AND of its reverse complement
 * under each of n_models models -- what Phymm's scoreReadsGlim.pl asks of `simple-score` for every genome's ICM
 * (scripts/scoreReadsGlim.pl:450,482; BASELINE configs[3]).  d_sums (device): [n_models][n_reads][2] doubles,
 * [..][0] = the read, [..][1] = its reverse complement.  Periodicity-1 models of the default shape (depth 7,
 * window <= 15) take a batched path at the six-frame kernel's rate; any other model goes through the exact
 * segment kernel.  Sums are sequential double additions in string order (bit-identical to the reference). */
int gmg_score_reads_strings(const gmg_model *const *models, int n_models, const gmg_reads *reads,
                            double *d_sums, void *stream);

/* ICM_t::Partial_Window_Prob (src/ICM/icm.cc:807-842) for the LAST base of every
 * segment (predict_pos = len-1), with the partial-window rule applied whatever
 * the length, as the reference does.  d_out[i]; 0.0 for an empty segment. */
int gmg_segment_partial_prob(const gmg_model *m, const gmg_reads *reads, const gmg_segments *segs,
                             int frame, double *d_out, void *stream);

/* All_Frame_Score (src/Glimmer/glimmer3.cc:328-359) on every segment: the
 * segment is the ORF buffer orientation used by Score_Orfs, `d_prefix_len[i]`
 * (device, uint32) is the number of leading buffer bases to score (best_j-2),
 * `d_frame[i]` (device, int32 in {1,2,3,-1,-2,-3}) selects Permute_By_Frame
 * (glimmer3.cc:1013-1088).  d_af[6*i + k]. */
int gmg_all_frame_score(const gmg_model *gene, const gmg_reads *reads, const gmg_segments *segs,
                        const uint32_t *d_prefix_len, const int32_t *d_frame, double *d_af,
                        void *stream);

/* ICM_t::Full_Window_Prob / Full_Window_Distrib (src/ICM/icm.cc:512-610) on
 * n_windows explicit windows.  d_windows: model_len codes (0..3) per window, one
 * byte each; d_frames: sub-model per window.  d_dist4[4*i+b] (may be NULL),
 * d_prob[i] (may be NULL) = (double) dist[code of last window char]. */
int gmg_window_distrib(const gmg_model *m, const uint8_t *d_windows, const int32_t *d_frames,
                       uint64_t n_windows, float *d_dist4, double *d_prob, void *stream);

/* ---- Score_Orfs inner loop (src/Glimmer/glimmer3.cc:1275-1552) ------------------- */

/* One Orf_t as Find_Orfs produced it (src/Common/gene.hh:101-139). */
typedef struct gmg_orf {
    uint32_t read;           /* index into the gmg_reads batch                                   */
    int32_t frame;           /* Orf_t::Get_Frame(): +1..+3 forward, -1..-3 reverse                */
    int32_t stop_position;   /* Orf_t::Get_Stop_Position(): first base of the stop codon, 1-based */
    int32_t orf_len;         /* Orf_t::Get_Orf_Len()                                              */
} gmg_orf;

/* The globals Score_Orfs reads (glimmer3.cc:23,61,71,122,148; glimmer_base.cc:2636-2712). */
typedef struct gmg_orf_params {
    int32_t min_gene_len;        /* Min_Gene_Len (default 75)                                    */
    int32_t allow_truncated;     /* Allow_Truncated_Orfs (-X)                                    */
    int32_t use_first_start;     /* Use_First_Start_Codon (-f)                                   */
    int32_t ignore_score_len;    /* Ignore_Score_Len (INT_MAX = never)                           */
    double start_threshold;      /* Start_Threshold (-6)                                         */
    int32_t n_start_codons;      /* <= 8                                                         */
    char start_codon[8][4];      /* Start_Codon strings, e.g. "atg","gtg","ttg" (IUPAC allowed)  */
} gmg_orf_params;

/* One Start_t (src/Glimmer/glimmer_base.hh:80-88), in the order Score_Orfs pushes them. */
typedef struct gmg_start {
    double score;            /* score[j-1] - indep_score[j-1], after the Ignore_Score_Len boost  */
    int32_t j, pos;
    int32_t which;           /* index of the matching start codon, -1 for a truncated start      */
    int16_t truncated, first;
} gmg_start;

typedef struct gmg_orf_result {
    double gene_score;       /* 100 * best_score / (best_j - 2)                  (glimmer3.cc:1489) */
    double best_score;
    uint32_t start_begin;    /* first entry of this ORF's start list in the starts array          */
    uint32_t n_starts;
    int32_t first_j, best_j; /* gene length = best_j + 1                         (glimmer3.cc:1499) */
    int32_t best_pos;
    int16_t is_tentative_gene;   /* first_j+1 >= Min_Gene_Len && best_score > Start_Threshold (:1468); only then
                                    does the reference add the gene and its events (:1494-1548)   */
    int16_t orf_is_truncated;
} gmg_orf_result;

typedef struct gmg_orf_batch gmg_orf_batch;

/* Validates the ORFs against their reads (linear sequences only: the reference's circular wrap-around
 * is refused with GMG_ERANGE) and reserves room for the start lists; *out_max_starts is the number
 * of gmg_start entries the caller must provide to gmg_score_orfs. */
int gmg_orfs_upload(const gmg_reads *reads, const gmg_orf *orfs, uint64_t n_orfs,
                    uint64_t *out_max_starts, gmg_orf_batch **out);
int gmg_orf_batch_free(gmg_orf_batch *b);

/* The scoring part of Score_Orfs for every ORF of the batch: ORF buffer (Reverse_Transfer /
 * Complement_Transfer), gene and null Cumulative_Score from frame 1, the start-codon scan from the
 * 3' end, first/best start, truncated starts, gene score and the tentative-gene test.  `results`
 * (n_orfs) and `starts` (room for out_max_starts entries) are HOST buffers.  The start lists are packed
 * back to back in ORF order on the device, so only sum(n_starts) entries are copied into `starts`
 * (typically ~5 % of out_max_starts); results[i].start_begin indexes that packed array.
 * Events / DP / trace-back stay host code (src/Glimmer/glimmer_base.cc). */
int gmg_score_orfs(const gmg_model *gene, const gmg_model *null_model, const gmg_reads *reads,
                   const gmg_orf_batch *orfs, const gmg_orf_params *params,
                   gmg_orf_result *results, gmg_start *starts, void *stream);

/* The same in two steps, for callers that do not want to reserve out_max_starts entries on the host (a slot per in-frame codon
 * of every ORF: ~100 per ORF, of which ~3 % are used): gmg_score_orfs_begin scores and leaves the results on the device,
 * *out_n_starts = the number of starts of all ORFs; gmg_score_orfs_fetch copies results[n_orfs] and starts[*out_n_starts]. */
int gmg_score_orfs_begin(const gmg_model *gene, const gmg_model *null_model, const gmg_reads *reads,
                         const gmg_orf_batch *orfs, const gmg_orf_params *params, uint64_t *out_n_starts, void *stream);
int gmg_score_orfs_fetch(const gmg_orf_batch *orfs, gmg_orf_result *results, gmg_start *starts, void *stream);

/* ---- glimmer-mg front half on the device -------------------------------------------
 * Everything between the reads and Add_Events_* for glimmer-mg's user-ICM mode (the -i / -s error branch: see the flags below):
 *   Score_All_Frames      src/Glimmer/glimmer-mg.cc:1468-1510   (gmg_frame_score6's kernels)
 *   Find_Orfs             src/Glimmer/glimmer_base.cc:638-779   (linear sequences, no ignore regions)
 *   Save_Prev_Stops       src/Glimmer/glimmer-mg.cc:675-729     (folded into the ORF scan: lo / hi per ORF)
 *   Score_Orf_Starts      src/Glimmer/glimmer-mg.cc:1693-1861   with Cumulative_Frame_Score :561-604
 *   Score_Orfs_Errors     src/Glimmer/glimmer-mg.cc:1632-1685   boost, first_j, best score, threshold
 * The 48 B/base Frame_Scores table never leaves HBM; what comes back is one record per ORF and the start
 * lists.  The caller sorts each accepted ORF's list with Start_Cmp (glimmer_base.hh:90) and hands it to
 * Add_Events_Fwd / Add_Events_Rev exactly as Score_Orfs_Errors does. */
/* gmg_mg_params.flags: return only the ORFs Score_Orfs_Errors would hand to Add_Events_* (accepted != 0) and their start
 * lists -- same order, packed on the device, read_orf_off counting the kept ORFs; typically a few per cent of all ORFs */
#define GMG_MG_ACCEPTED_ONLY 1
/* glimmer-mg's error branch (src/Glimmer/glimmer-mg.cc:1513-1602 Score_Indels, :1771-1806 the substitution branch of
 * Score_Orf_Starts; exclusive, :952): Find_Orfs also keeps every ORF of orf_len >= Min_Indel_ORF_Len
 * (glimmer_base.cc:494,528,806), Score_Orf_Starts recurses through frame shifts at low-quality bases (-i) or through
 * the previous stop codon (-s), and every start carries the Error_t list of its path (gmg_mg_result_fetch_errors). */
#define GMG_MG_ALLOW_INDELS 2    /* glimmer-mg -i / --indel */
#define GMG_MG_ALLOW_SUBS 4      /* glimmer-mg -s / --sub   */

typedef struct gmg_mg_params {
    int32_t min_gene_len;        /* Min_Gene_Len (>= 4)                                           */
    int32_t allow_truncated;     /* Allow_Truncated_Orfs (glimmer-mg default: true)               */
    int32_t ignore_score_len;    /* Ignore_Score_Len                                              */
    int32_t n_start_codons;      /* <= 8                                                          */
    int32_t n_stop_codons;       /* <= 8                                                          */
    int32_t flags;               /* GMG_MG_ACCEPTED_ONLY | GMG_MG_ALLOW_INDELS or GMG_MG_ALLOW_SUBS, or 0 */
    double start_threshold;      /* Start_Threshold                                               */
    char start_codon[8][4];      /* Start_Codon strings (IUPAC allowed)                           */
    char stop_codon[8][4];       /* Stop_Codon strings                                            */
    /* the error branch; read only when flags has GMG_MG_ALLOW_INDELS or GMG_MG_ALLOW_SUBS */
    int32_t min_indel_orf_len;   /* Min_Indel_ORF_Len (glimmer_base.cc:40: 15)                    */
    int32_t indel_quality_threshold;     /* Indel_Quality_Threshold (glimmer-mg.cc:136: 18)       */
    int32_t indel_max;           /* Indel_Max (glimmer-mg.cc:138: 2); 0..2                        */
    int32_t circular;            /* gmg_find_orfs only: Genome_Is_Circular (glimmer-mg -r): every sequence of the batch is circular
                                    (glimmer_base.cc:680-688, Wrap_Around_Back :2793-2850, Wrap_Through_Front :2854-2900)   */
    double indel_suffix_score_threshold; /* Indel_Suffix_Score_Threshold (glimmer-mg.cc:134: -12) */
    const uint8_t *quality;      /* HOST, one Phred value per base, reads back to back (total_bases): the user's quality
                                    file (-q), Clean_Quality_454 (glimmer-mg.cc:519-546) is applied on the device;
                                    NULL: Set_Quality_454 (:1865-1906, homopolymer runs).  Indels only: with -s the
                                    reference never loads the values (:384-392)                    */
    /* classification mode: glimmer-mg rebuilds Indep_Model and Ignore_Score_Len for EVERY read from the GC of its classes
     * (Update_Meta_Null_ICM, glimmer-mg.cc:2050-2068, inside the ICM-grouped loop :361-451).  One call scores a whole
     * ICM group: read r takes model read_null[r] of `nulls` (instead of the null_model argument) and
     * read_ignore_score_len[r] (instead of ignore_score_len above).  All three NULL: one null model for the batch. */
    const struct gmg_null_set *nulls;
    const uint32_t *read_null;           /* HOST [n_reads] */
    const int32_t *read_ignore_score_len;/* HOST [n_reads], or NULL with nulls set: ignore_score_len for every read */
    /* gmg_find_orfs only: glimmer3 -i, Ignore_Region as Get_Ignore_Regions leaves it (glimmer_base.cc:833-930): 0-based lo, hi one past
     * the last ignored base, sorted, overlaps merged; the same regions apply to EVERY sequence of the batch (:844-847) */
    int32_t n_ignore_regions;
    int32_t reserved2;
    const int32_t *ignore_lo, *ignore_hi;        /* HOST [n_ignore_regions] */
} gmg_mg_params;

/* the Error_t list (src/Common/gene.hh:138-146) of one start: type 0 insertion, 1 deletion, 2 substitution */
typedef struct gmg_start_errors {
    int32_t pos[2];
    int8_t type[2];
    int8_t n;                    /* 0..2 entries                                                  */
    int8_t reserved;
} gmg_start_errors;

typedef struct gmg_mg_orf {
    uint32_t read;               /* index into the gmg_reads batch                                */
    int32_t frame, stop_position, orf_len, gene_len;     /* the Orf_t Find_Orfs built             */
    int32_t lo, hi;              /* Score_Orf_Starts' bounds (glimmer-mg.cc:1730-1757)            */
    int32_t first_j;             /* j of the first start (after the sort: front / back)           */
    uint32_t start_begin, n_starts;      /* its start list: starts[start_begin .. +n_starts), in the
                                            order Score_Orf_Starts pushed them, boost applied      */
    int16_t accepted;            /* non-empty, first_j+1 >= Min_Gene_Len, best_score > Start_Threshold
                                    (glimmer-mg.cc:1656-1676): the ORF goes to Add_Events_*.  Error branch only: paths
                                    tie on pos with different j, so first_j is whichever entry the reference's unstable
                                    sort puts first (reported: the smallest).  2 = best_score passes but only some of those
                                    entries pass the length test, so the caller decides after ITS sort -- a guard: the
                                    reference's own filter (glimmer-mg.cc:1821) already makes every pushed j pass */
    int16_t orf_is_truncated;
    int32_t reserved;
    double best_score;           /* max over the boosted start scores, -DBL_MAX if none           */
} gmg_mg_orf;

typedef struct gmg_mg_result gmg_mg_result;

/* Runs the whole front half for every read of the batch.  d_frame_scores: device buffer of
 * 6 * total_bases doubles that receives the Frame_Scores table (the caller may want it), or NULL to let
 * the call allocate and release its own.  Needs a gene model of periodicity 3. */
int gmg_mg_score_reads(const gmg_model *gene, const gmg_model *null_model, const gmg_reads *reads,
                       const gmg_mg_params *params, double *d_frame_scores, gmg_mg_result **out,
                       void *stream);
/* The same for a batch whose reads come in consecutive GROUPS, every group under its own gene ICM: one chunk of glimmer-mg's
 * classification mode (the loop over ICM_Sequences, src/Glimmer/glimmer-mg.cc:361-451: Gene_ICM.Read per group, then every read
 * of the group against the null model and Ignore_Score_Len of ITS classes, Update_Meta_Null_ICM :2050-2068) in ONE call: the
 * reads gathered in visiting order (gmg_reads_select on the order of gmg_classes_plan), groups[k] = reads [read_begin, read_end)
 * of that batch (consecutive, from 0 to n_reads), params->nulls / read_null / read_ignore_score_len per read (required; one
 * stop-codon set per call).  The six-frame pass swaps the group's tables in LDS as it crosses from group to group -- one launch
 * whatever the number of groups (up to 2^27 - 1: a chunk may meet one ICM file per read); everything behind it never sees a gene
 * model.  Results as gmg_mg_score_reads', reads in batch order. */
typedef struct gmg_mg_group {
    const gmg_model *gene;
    uint64_t read_begin, read_end;
} gmg_mg_group;
int gmg_mg_score_groups(const gmg_mg_group *groups, int n_groups, const gmg_model *null_model, const gmg_reads *reads,
                        const gmg_mg_params *params, gmg_mg_result **out, void *stream);
/* Find_Orfs alone (src/Glimmer/glimmer_base.cc:638-817) for every read of the batch -- the ORF list glimmer3's Score_Orfs /
 * gmg_score_orfs and glimmer-mg's Score_Orfs_Errors start from.  Uses min_gene_len, allow_truncated and the codon lists of
 * `params`; the result holds the Orf_t fields (and lo / hi), no start lists (n_starts = 0).  With params->n_ignore_regions
 * (glimmer3 -i) or params->circular (glimmer-mg -r) the scan follows the reference's other two modes (lo = hi = 0 then);
 * GMG_EINVAL where the reference itself aborts (a circular sequence with a reverse frame that holds no stop codon). */
int gmg_find_orfs(const gmg_reads *reads, const gmg_mg_params *params, gmg_mg_result **out, void *stream);
/* Sizes of the result: ORFs of all reads (Find_Orfs order, read by read) and start entries. */
int gmg_mg_result_info(const gmg_mg_result *r, uint64_t *n_orfs, uint64_t *n_starts);
/* Copies the result to HOST buffers: orfs[n_orfs], starts[n_starts] and, if not NULL,
 * read_orf_off[n_reads + 1] (ORFs of read i are orfs[read_orf_off[i] .. read_orf_off[i+1])). */
int gmg_mg_result_fetch(const gmg_mg_result *r, gmg_mg_orf *orfs, gmg_start *starts, uint64_t *read_orf_off);
/* Error branch: errs[n_starts], parallel to starts (all-zero entries without the error flags). */
int gmg_mg_result_fetch_errors(const gmg_mg_result *r, gmg_start_errors *errs);
/* the same copies on `stream` (they then overlap the scoring of the next batch on another stream) */
int gmg_mg_result_fetch_on(const gmg_mg_result *r, gmg_mg_orf *orfs, gmg_start *starts, uint64_t *read_orf_off,
                           void *stream);
int gmg_mg_result_free(gmg_mg_result *r);
/* gmg_mg_* keeps released device buffers for the next call (allocation of GB-sized buffers is slow);
 * this returns the idle ones to the driver. */
int gmg_trim_cache(void);

/* ---- FASTA ingest on the device -----------------------------------------------------
 * Replaces the loop  while (Fasta_Read (fp, seq, hdr))  (src/Common/fasta.cc:236-286) + the per-base
 * tolower (Filter (ch)) of the callers (src/Glimmer/glimmer3.cc:270-271, glimmer-mg.cc:381-382) + the g/c count of
 * Set_GC_Fraction (src/Glimmer/glimmer_base.cc:2564-2595).  `bytes` is the whole file (host memory, < 2^31 bytes);
 * it is copied to the device once and parsed there.  Records follow Fasta_Read exactly: a '>' anywhere outside a
 * header line starts a record, the header runs to the end of that line, every non-isspace byte up to the next '>'
 * is a base, bytes in front of the first '>' are skipped.  *reads is ready for every scoring call;
 * *index keeps the counts and, per read, the extent of its header line in `bytes`. */
typedef struct gmg_fasta gmg_fasta;
int gmg_fasta_ingest(const char *bytes, uint64_t n_bytes, gmg_reads **reads, gmg_fasta **index);
/* the same, with every copy and kernel on `stream` (a pipeline ingests piece i+1 while piece i is scored) */
int gmg_fasta_ingest_on(const char *bytes, uint64_t n_bytes, gmg_reads **reads, gmg_fasta **index, void *stream);
/* Places where a big file can be cut into pieces of about chunk_bytes for separate gmg_fasta_ingest calls: a '>' that
 * directly follows a newline always starts a record.  cuts[0] = 0 ... cuts[pieces] = n_bytes; returns the number of pieces
 * (at most max_pieces; cuts needs max_pieces + 1 entries). */
int gmg_fasta_split(const char *bytes, uint64_t n_bytes, uint64_t chunk_bytes, uint64_t *cuts, int max_pieces);
/* n_reads, bases of all reads, and how many of them are g or c after filtering (Indep_GC_Frac = gc / total) */
int gmg_fasta_info(const gmg_fasta *index, uint64_t *n_reads, uint64_t *total_bases, uint64_t *gc_count);
/* hdr string of read i = bytes[hdr_begin[i] .. hdr_end[i])  (HOST arrays of n_reads entries) */
int gmg_fasta_headers(const gmg_fasta *index, uint64_t *hdr_begin, uint64_t *hdr_end);
int gmg_fasta_free(gmg_fasta *index);

/* ---- one job over several GPUs / several batches (host only) -----------------------------------
 * Reads shard embarrassingly: every sequence is scored on its own (src/Glimmer/glimmer3.cc:262-310,
 * glimmer-mg.cc:361-451); the one run-wide quantity is the null model's GC fraction, a ratio of two counts over all
 * bases (Set_GC_Fraction, src/Glimmer/glimmer_base.cc:2564-2595).  One process per GPU takes a contiguous range of
 * reads of about equal BASE count, the per-shard {gc, total} counts are summed on the host, results are concatenated
 * in shard order; inside a shard the same plan cuts batches that fit the 32-bit fields of the result records. */
/* read_begin[k] .. read_begin[k+1] = reads of shard k (n_shards + 1 entries); cut k at the read boundary nearest to
 * k * total_bases / n_shards. */
int gmg_shard_plan(const uint64_t *base_offsets, uint64_t n_reads, int n_shards, uint64_t *read_begin);
/* The same on the bytes of an unparsed FASTA file: cuts[k] = the first certain record start ('>' behind a newline) at
 * or behind k * n_bytes / n_shards; cuts[0] = 0, cuts[n_shards] = n_bytes.  Every shard is a valid input of
 * gmg_fasta_ingest / gmg_fasta_split. */
int gmg_fasta_shard_ranges(const char *bytes, uint64_t n_bytes, int n_shards, uint64_t *cuts);
/* Indep_GC_Frac from per-shard counts (gmg_fasta_info).  as_reference != 0: the reference's `unsigned int` counters,
 * which wrap beyond 2^32 bases (byte-identical output on such files); 0: 64-bit counts. */
double gmg_gc_fraction(const uint64_t *gc_counts, const uint64_t *base_counts, int n_shards, int as_reference);

/* ---- glimmer-mg's classification mode, -c (host only) --------------------------------------------
 * With -c every read is scored by the gene ICM its Phymm classes name, against a null model and with stop codons that
 * are rebuilt per read, and the reads are visited -- and <tag>.predict is written -- ICM by ICM in the iteration order of
 * the reference's hash tables, not in file order.  gmg_classes_* reproduce that bookkeeping; the scoring of one ICM's
 * reads is then ONE gmg_mg_score_reads call (gmg_reads_select, gmg_null_set_upload, gmg_mg_params.read_null /
 * read_ignore_score_len) per stop-codon set.  INTEGRATION.md shows the loop; integration/glimmer-mg_gpu.cc runs it. */
typedef struct gmg_classes gmg_classes;
/* Parse_Classes (src/Glimmer/glimmer-mg.cc:726-758) on the text of a classification file ("<read> <class> <class> ..."
 * per line, class = <strain>|<NC>), then Read_Meta_ICMs (:998-1027) with Classes_ICM_File (:473-515; looks for the
 * double ICMs under icm_dir with stat), Read_Meta_GC (:1389-1420: <icm_dir>/<strain>/<NC>.gc.txt, 0.5 when missing) and
 * Read_Meta_Stops (:1211-1250: transl_table of <NC>.gbk, 11 when missing).  icm_dir is the reference's ICM_dir (:147). */
int gmg_classes_load(const char *class_text, uint64_t n_bytes, const char *icm_dir, gmg_classes **out);
int gmg_classes_free(gmg_classes *c);
/* classified reads, distinct ICM files, distinct classes, classes without a .gc.txt (any pointer may be NULL) */
int gmg_classes_info(const gmg_classes *c, uint64_t *n_reads, uint32_t *n_icms, uint32_t *n_classes, uint64_t *n_missing_gc);
/* ICM file k (0 .. n_icms-1) in the order the reference's loop over ICM_Sequences loads them (glimmer-mg.cc:361-364) */
const char *gmg_classes_icm_file(const gmg_classes *c, uint32_t k);
/* The plan of ONE chunk of the input (the reference reads Chunk_Sequences = 500000 reads at a time, glimmer-mg.cc:128,
 * 326-356): hdr[i] / hdr_len[i] = header line of read i of the chunk (its first white-space-delimited token is the key;
 * when two reads share a key the later one is taken, as Read_Indexes does).  order[k], k < *n_order <= n: chunk index of
 * the k-th read the reference processes and prints; order[icm_begin[f] .. icm_begin[f+1]) are the reads of ICM file f
 * (icm_begin: n_icms + 1 entries).  Reads without a line in the class file are never processed (:369-371).  Per
 * processed read, parallel to order (either may be NULL): gc[k] = Indep_GC_Frac of Update_Meta_Null_ICM (:2050-2064: the
 * float GCs of its classes summed in double, divided by their float count), transl[k] = Genbank_Xlate_Code of
 * Update_Meta_Stop (:2196: the table of its FIRST class). */
int gmg_classes_plan(const gmg_classes *c, const char *const *hdr, const uint32_t *hdr_len, uint64_t n,
                     uint64_t *order, uint64_t *icm_begin, double *gc, int32_t *transl, uint64_t *n_order);
/* Set_Stop_Codons_By_Code (src/Common/gene.cc:1560-1624): the stop codons of a GenBank translation table, in the
 * reference's order (Build_Indep_WO_Stops depends on it); GMG_EINVAL for a table the reference does not know. */
int gmg_stop_codons_by_code(int code, char stop_codon[8][4], int *n_stop_codons);
/* Set_Ignore_Score_Len (src/Glimmer/glimmer_base.cc:2597-2633) */
int gmg_ignore_score_len(double gc_frac, const char (*stop_codon)[4], int n_stop_codons, int32_t *out);

/* ---- build-icm: training counts on the device ---------------------------------------
 * Replaces the counting of ICM_Training_t: Count_Char_Pairs for the roots (src/ICM/icm.cc:1841-1870, called from
 * Train_Model, icm.cc:1373-1390), Count_Char_Pairs_Restricted + Get_Training_Node for every deeper level
 * (icm.cc:1190-1256, called from Complete_Tree, icm.cc:1082-1083) and Count_Single_Chars (icm.cc:1874-1896).
 * `strings` = the training strings as a gmg_reads batch, in the orientation Train_Model gets them (build-icm -r
 * reverses them first); it must outlive the trainer.  The tree is built level by level, as in the reference: */
typedef struct gmg_trainer gmg_trainer;
int gmg_trainer_create(const gmg_reads *strings, int model_len, int model_depth, int periodicity, gmg_trainer **out);
/* The 4 x 4 pair tables of tree level `level` (0 = the roots).  Levels are taken in order 0, 1, .. model_depth.
 * For level >= 1, mip_prev[f * 4^(level-1) + k] = mut_info_pos the caller chose for node k of level - 1 in sub-model f
 * (< 0: the tree stops there, its windows count no further).  counts (HOST, periodicity * 4^level * npos * 16 int32,
 * npos = max(model_len - 1, 1)):  counts[((f * 4^level + k) * npos + i) * 16 + 4 * code(w[i]) + code(w[model_len-1])]
 * = number of complete windows w of sub-model f that reach node k of the level -- the reference's
 * train[f][first + k].count[i][pair] (src/ICM/icm.hh:88-101).  A window starting at offset s of its string belongs to
 * sub-model (model_len + s) mod periodicity (icm.cc:1203-1226).  With model_len = 1 only entry [last] of table 0 counts. */
int gmg_trainer_level_counts(gmg_trainer *t, int level, const int16_t *mip_prev, int32_t *counts);
int gmg_trainer_free(gmg_trainer *t);

/* ---- device memory helpers (for callers without their own allocator) -------- */
int gmg_device_malloc(void **d_ptr, size_t bytes);
int gmg_device_free(void *d_ptr);
int gmg_memcpy_h2d(void *d_dst, const void *src, size_t bytes, void *stream);
int gmg_memcpy_d2h(void *dst, const void *d_src, size_t bytes, void *stream);
/* Page-lock / release a host buffer the caller owns (file bytes, result arrays), so that the copies the library
 * makes from / to it run at PCIe speed. */
int gmg_host_register(void *ptr, size_t bytes);
int gmg_host_unregister(void *ptr);
/* Non-blocking streams for callers without a HIP binding of their own (every `void *stream` parameter takes one). */
int gmg_stream_create(void **stream);
int gmg_stream_destroy(void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GMG_H */
