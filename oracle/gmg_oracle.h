/*
 * gmg_oracle.h -- CPU ORACLE for the Glimmer-MG IMM scoring path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference
 * algorithm (davek44/Glimmer-MG, src/ICM/icm.cc scoring side and the six-frame
 * loops of src/Glimmer/glimmer3.cc / glimmer-mg.cc).  Only tests/, the
 * smoke() entry point and bench.py's cpu_baseline leg may link or call it --
 * and there only as the checker.  The product path (glimmer-mg_amd/) never
 * includes, links or falls back to anything in oracle/.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_*.py)
 *   - against the reference's own committed outputs
 *     (sample-run/glimmer-mg/results/icm-N.scores.tmp, 6 x 999 Score_String
 *     values, copied as data to tests/golden/), and
 *   - against golden vectors produced in the build container by the real
 *     reference objects (oracle/_ref, recipe in oracle/Makefile, generator
 *     oracle/gen_golden.py); the training side (orc_train_*) against .icm files
 *     written by the reference's build-icm compiled the same way.
 *
 * Every function cites the reference file:line it restates.  Paths are
 * relative to the reference root.
 */
#ifndef GMG_ORACLE_H
#define GMG_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One interpolated context model: src/ICM/icm.hh:106-129.
 * mip[p*num_nodes + n]     = mut_info_pos of node n in sub-model p
 * prob[(p*num_nodes+n)*4+b] = natural-log probability of base b at node n */
typedef struct orc_model {
    int model_len;    /* window width W (icm.hh:120)               */
    int model_depth;  /* max tree depth D (icm.hh:122)             */
    int periodicity;  /* number of cyclic sub-models P (icm.hh:125)*/
    int num_nodes;    /* nodes per sub-model (icm.hh:127)          */
    int16_t *mip;
    float   *prob;
} orc_model;

/* character helpers ------------------------------------------------------ */
int  orc_filter(int ch);       /* src/Common/gene.cc:1139-1175 */
int  orc_complement(int ch);   /* src/Common/gene.cc:15-19,1084-1091 */
int  orc_subscript(int ch);    /* src/ICM/icm.cc:2008-2027; -1 on bad char */

/* model construction / IO ------------------------------------------------ */
orc_model *orc_model_new(int model_len, int model_depth, int periodicity); /* icm.cc:24-44 */
void       orc_model_free(orc_model *m);                                   /* icm.cc:48-61 */
/* icm.cc:614-726 (Input) + 846-861 (Read).  Returns NULL and fills err on failure. */
orc_model *orc_model_read(const char *path, char *err, size_t errlen);
orc_model *orc_model_from_bytes(const unsigned char *buf, size_t n, char *err, size_t errlen);
/* icm.cc:729-803,961-998 (binary Output / Write_Header).  0 on success. */
int        orc_model_write(const orc_model *m, const char *path);
/* icm.cc:65-216.  stop_codon = n_stops NUL-terminated 3-letter strings.  The
 * model must be (3,2,3).  0 on success, -1 if incompatible. */
int        orc_build_indep_wo_stops(orc_model *m, double gc_frac,
                                    const char *const *stop_codon, int n_stops);

/* per-base scoring -------------------------------------------------------- */
double orc_full_window_prob(const orc_model *m, const char *s, int frame);     /* icm.cc:557-610 */
double orc_partial_window_prob(const orc_model *m, int predict_pos,
                               const char *s, int frame);                      /* icm.cc:807-842 */
void   orc_full_window_distrib(const orc_model *m, const char *s, int frame,
                               float dist[4]);                                 /* icm.cc:512-553 */

/* accumulations ----------------------------------------------------------- */
double orc_score_string(const orc_model *m, const char *s, int len, int frame);           /* icm.cc:864-903 */
void   orc_cumulative_score(const orc_model *m, const char *s, int n, double *score,
                            int frame);                                                   /* icm.cc:354-405 */
void   orc_cumulative_score_string(const orc_model *m, const char *s, int len, int frame,
                                   double *cum_score /* len+1 */);                        /* icm.cc:409-452 */
void   orc_frame_score(const orc_model *m, const char *s, int n, double *score,
                       int frame);                                                        /* icm.cc:485-509 */

/* six-frame loops --------------------------------------------------------- */
/* src/Glimmer/glimmer-mg.cc:1468-1510 (Score_All_Frames).  seq = filtered lower-case
 * read of length L.  out = 6 rows of L doubles, row-major (out[f*L+p]). */
void   orc_score_all_frames(const orc_model *gene, const orc_model *indep,
                            const char *seq, int L, double *out);
/* src/Glimmer/glimmer-mg.cc:561-604 (Cumulative_Frame_Score) on a 6xL table. */
void   orc_cumulative_frame_score(const double *frame_scores, int L, int frame,
                                  int lo, int hi, double *score /* hi-lo */);
/* src/Glimmer/glimmer3.cc:328-359 (All_Frame_Score) incl. Permute_By_Frame :1013-1088. */
void   orc_all_frame_score(const orc_model *gene, const char *s, int len, int frame,
                           double af[6]);
/* ORF buffer builders: glimmer_base.cc:2505-2533 (Reverse_Transfer, no wrap needed
 * when 0 <= start-len+1) and :410-434 (Complement_Transfer). */
void   orc_reverse_transfer(char *buff, const char *s, int n, int start, int len);
void   orc_complement_transfer(char *buff, const char *s, int n, int start, int len);

/* Score_Orfs inner loop for ONE ORF (src/Glimmer/glimmer3.cc:1301-1503): buffer, two cumulative
 * scores from frame 1, start-codon scan from the 3' end, first/best start, gene score.  seq is the
 * filtered lower-case sequence (linear).  Returns the number of starts written (<= cap), or -1 when the
 * ORF is skipped before the gene test (first_j + 1 < Min_Gene_Len, glimmer3.cc:1431). */
typedef struct orc_start { double score; int j, pos, which, truncated, first; } orc_start;
typedef struct orc_orf_params {
    int min_gene_len, allow_truncated, use_first_start, ignore_score_len;
    double start_threshold;
    int n_start_codons;
    const char *start_codon[8];
} orc_orf_params;
typedef struct orc_orf_out {
    double gene_score, best_score;
    int first_j, best_j, best_pos, is_tentative_gene, orf_is_truncated;
} orc_orf_out;
unsigned orc_ch_mask(int ch);                                 /* src/Common/gene.cc:954-995 */
int orc_score_orf(const orc_model *gene, const orc_model *indep, const char *seq, int seq_len,
                  int frame, int stop_position, int orf_len, const orc_orf_params *prm,
                  orc_start *starts, int cap, orc_orf_out *out);

/* ---- glimmer-mg front half: ORF discovery + start scan on the six-frame table ---------------------
 * Linear sequences, no ignore regions, no indel / substitution branch (Allow_Indels = Allow_Subs = false,
 * the default of glimmer-mg; src/Glimmer/glimmer-mg.cc:100-102). */
typedef struct orc_orf { int frame, stop_position, gene_len, orf_len; } orc_orf;
typedef struct orc_mg_params {
    int min_gene_len, allow_truncated, ignore_score_len;
    double start_threshold;
    int n_start_codons, n_stop_codons;
    const char *start_codon[8], *stop_codon[8];
} orc_mg_params;
/* Find_Orfs (src/Glimmer/glimmer_base.cc:638-779) with Do_Fwd_Stop_Codon :460-504, Do_Rev_Stop_Codon :506-537,
 * Finish_Orfs :783-817, Handle_First_Forward_Stop :946-985, Handle_First_Reverse_Stop :989-1015,
 * Handle_Last_Reverse_Stop :1019-1072.  Returns the number of ORFs (only the first cap are written). */
int orc_find_orfs(const char *seq, int n, const orc_mg_params *prm, orc_orf *orfs, int cap);
/* Find_Orfs in full: ignore regions (Ignore_Region as Get_Ignore_Regions leaves it, glimmer_base.cc:833-930) and circular sequences
 * (Wrap_Around_Back :2793-2850, Wrap_Through_Front :2854-2900); min_indel_orf_len < 0: Allow_Indels = Allow_Subs = false.
 * Returns the number of ORFs, -1 where the reference's assert in Wrap_Around_Back would fire. */
int orc_find_orfs_general(const char *seq, int len, const orc_mg_params *prm, int min_indel_orf_len, int circular,
                          const int *ign_lo, const int *ign_hi, int n_ignore, orc_orf *orfs, int cap);
/* Save_Prev_Stops (src/Glimmer/glimmer-mg.cc:675-729): fwd_prev[n], rev_next[n] */
void orc_save_prev_stops(const char *seq, int n, const orc_mg_params *prm, int *fwd_prev, int *rev_next);
typedef struct orc_mg_out { int lo, hi, first_j, accepted, orf_is_truncated; double best_score; } orc_mg_out;
/* Score_Orf_Starts without errors (glimmer-mg.cc:1693-1861) on frame_scores[6][n] (Cumulative_Frame_Score
 * :561-604), then the per-ORF part of Score_Orfs_Errors (:1632-1685): Ignore_Score_Len boost, first_j, best
 * score, threshold.  starts come back in the order Score_Orf_Starts pushed them (the reference then sorts
 * them by pos with std::sort, Start_Cmp glimmer_base.hh:90).  Returns the number of starts. */
int orc_mg_score_orf(const double *frame_scores, const char *seq, int n, const int *fwd_prev, const int *rev_next,
                     int frame, int stop_position, const orc_mg_params *prm, orc_start *starts, int cap,
                     orc_mg_out *out);

/* ---- glimmer-mg's error branch (-i indels / -s substitutions) -------------------------------------- */
#define ORC_MAX_ERRORS 4
typedef struct orc_mg_err_params {
    int allow_indels, allow_subs;                       /* Allow_Indels / Allow_Subs (glimmer-mg.cc:100-102), exclusive (:952) */
    int indel_quality_threshold, indel_max;             /* glimmer-mg.cc:136,138 (18, 2) */
    double indel_suffix_score_threshold;                /* glimmer-mg.cc:134 (-12) */
} orc_mg_err_params;
typedef struct orc_start_err { orc_start s; int n_errors; int err_pos[ORC_MAX_ERRORS], err_type[ORC_MAX_ERRORS]; } orc_start_err;
/* Find_Orfs with the error modes' extra ORFs: orf_len >= Min_Indel_ORF_Len also qualifies (glimmer_base.cc:494,528,806;
 * min_indel_orf_len < 0 switches that off) */
int orc_find_orfs_err(const char *seq, int n, const orc_mg_params *prm, int min_indel_orf_len, orc_orf *orfs, int cap);
void orc_set_quality_454(const char *seq, int n, int *q);                                  /* glimmer-mg.cc:1865-1906 */
void orc_clean_quality_454(const char *seq, int n, int *q, int indel_quality_threshold);   /* glimmer-mg.cc:519-546 */
/* Score_Orf_Starts with Score_Indels / the substitution branch (glimmer-mg.cc:1513-1602, 1693-1861, recursive) and
 * the per-ORF part of Score_Orfs_Errors.  quality: Quality_Values of the read (after Set_ / Clean_Quality_454) or
 * NULL.  Starts come back in push order with their Error_t lists.  out->accepted: 1 accepted, 0 not, 2 = accepted if
 * the reference's (unstable) sort puts an entry with j + 1 >= Min_Gene_Len first -- entries tie on pos.  */
int orc_mg_score_orf_errors(const double *frame_scores, const char *seq, int n, const int *fwd_prev, const int *rev_next,
                            const int *quality, int frame, int stop_position, const orc_mg_params *prm,
                            const orc_mg_err_params *ep, orc_start_err *starts, int cap, orc_mg_out *out);

/* ---- training (build-icm): ICM_Training_t, src/ICM/icm.cc:1010-1455, 1841-1955 ------------------------------- */
/* The 4 x 4 pair counts of one tree level.  level 0: Count_Char_Pairs per sub-model with Train_Model's offsets
 * (icm.cc:1373-1390, 1841-1870); level >= 1: Count_Char_Pairs_Restricted + Get_Training_Node (icm.cc:1190-1256) on
 * m->mip of the levels above.  strings are used as given (lower case; every character through orc_subscript).
 * counts[((f * 4^level + k) * (model_len - 1) + i) * 16 + 4 * code(w[i]) + code(w[model_len-1])], k = node - first
 * node of the level; the caller zeroes it.  model_len = 1: counts[f * 16 + code] of the predicted base alone
 * (Count_Single_Chars, icm.cc:1874-1896). */
void orc_train_level_counts(const orc_model *m, const char *const *strings, int n_strings, int level,
                            int32_t *counts);
/* Get_Mutual_Info (icm.cc:1900-1955) for one 4 x 4 table. */
double orc_mutual_info(const int32_t ct[16], int sum);
/* Train_Model (icm.cc:1356-1455) + Complete_Tree (icm.cc:1061-1186) + Interpolate_Probs (icm.cc:1260-1330) +
 * Take_Logs (icm.cc:1334-1352).  mut_info (may be NULL): [periodicity * num_nodes] floats, the value the reference
 * stores in ICM_Score_Node_t::mut_info (text output only). */
orc_model *orc_train_model(const char *const *strings, int n_strings, int model_len, int model_depth,
                           int periodicity, float *mut_info);

/* ---- glimmer-mg -c: which ICM file scores which read, in which order, with which null-model GC and stop codons ----
 * Parse_Classes (src/Glimmer/glimmer-mg.cc:726-758), Read_Meta_ICMs :998-1027 with Classes_ICM_File :473-515 (stat on the
 * double-ICM paths under icm_dir), Read_Meta_GC :1389-1420, Read_Meta_Stops :1211-1250.  The visiting order is the
 * iteration order of libstdc++'s SGI hash table (<ext/hash_map>, not part of the reference tree): restated in gmg_oracle.c.
 * NULL for a file the reference would crash on (a line without read or class, a class without '|'). */
typedef struct orc_classes orc_classes;
orc_classes *orc_classes_load(const char *text, long n, const char *icm_dir);
void orc_classes_free(orc_classes *c);
int orc_classes_n_icms(const orc_classes *c);
const char *orc_classes_icm_file(const orc_classes *c, int k);    /* in the order of the loop of glimmer-mg.cc:361 */
/* One chunk of the main loop (glimmer-mg.cc:326-375): hdr = header lines of the chunk's reads; order[k] = chunk index of the
 * k-th processed read, icm_begin[n_icms + 1], gc[k] = Indep_GC_Frac (Update_Meta_Null_ICM :2050-2064), transl[k] =
 * Genbank_Xlate_Code (Update_Meta_Stop :2196).  Returns the number of processed reads. */
long orc_classes_plan(const orc_classes *c, const char *const *hdr, long n, long *order, long *icm_begin, double *gc,
                      int *transl);
int orc_stop_codons_by_code(int code, const char *out[8]);                        /* src/Common/gene.cc:1560-1624 */
int orc_ignore_score_len(double gc_frac, const char *const *stop_codon, int n_stops);   /* glimmer_base.cc:2597-2633 */

/* Fasta_Read (src/Common/fasta.cc:236-286) on a memory buffer.  Starts at *pos; returns 0 at the end of the
 * input, else 1 with the header extent [*hdr_begin, *hdr_end) in buf, the raw sequence characters (every
 * non-isspace byte up to the next '>') in seq[0 .. *seq_len) (seq needs room for n - *pos bytes), *pos advanced. */
int orc_fasta_next(const char *buf, long n, long *pos, long *hdr_begin, long *hdr_end, char *seq, long *seq_len);

/* The ingest loop of the reference in one call (timing baseline): every record through orc_fasta_next, every base
 * through tolower (Filter (ch)) into out (may be NULL), g/c counted.  Returns the number of records. */
long orc_fasta_all(const char *buf, long n, char *out, long *n_bases, long *gc);

/* Whole-job helper used by bench.py's cpu_baseline leg: score n_reads reads of
 * fixed length L (concatenated, filtered lower-case) into out[read][6][L].
 * Returns number of bases scored.  Single-threaded like the reference. */
long   orc_score_reads_6frame(const orc_model *gene, const orc_model *indep,
                              const char *seqs, int n_reads, int L, double *out);

#ifdef __cplusplus
}
#endif
#endif
