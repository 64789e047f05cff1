// gmg_internal.h -- private structures shared by the HIP translation units.
// gfx950 (MI355X, CDNA4) only: 64-wide wavefronts, 160 KiB LDS per CU.
#ifndef GMG_INTERNAL_H
#define GMG_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gmg.h"

#define GMG_MAX_MODEL_LEN 32      // context of W-1 <= 31 bases fits one 64-bit register
#define GMG_FAST_MAX_LEN 16       // fast kernel: (W-1)*2 bits + predicted base in 32 bits
#define GMG_FAST_MAX_DEPTH 8
#define GMG_DENSE_MAX_LEN 6       // 4^6 = 4096-entry direct table for tiny (null) models
#define GMG_TILE 1024             // bases per tile_read entry
#define GMG_GUARD_WORDS 320        // zero words before and after the packed reads (>= one k_frame6s chunk + window)

// Device view of one model.  Pointers are HBM.
struct GmgDevModel {
    int W, D, P, N;
    const int8_t *mip;     // [P][N]   mut_info_pos, -2..W-1 (src/ICM/icm.hh:108)
    const float *prob;     // [P][N][4] ln-probabilities   (src/ICM/icm.hh:112)
    // --- flattened tables, built once at upload (gmg_model_upload) ---
    // "completed" tree: every early stop (mip -1 / -2) is expanded down to depth D so that a
    // full-window descent always takes exactly D steps and ends on a leaf whose row already is
    // the row the reference would have used (own row, or the parent's for a cut node).
    const uint8_t *cshift; // [P][cstride]  2*mip of the completed tree, levels 0..D-1, level l at (4^l-1)/3;
                           //               0 for nodes below an original stop
    const float *crow;     // [P][ctot][4]  row the reference uses when a descent ends at that completed-tree
                           //               node (own row; the parent's for a cut node; the stopping ancestor's
                           //               for nodes below a stop).  Leaves (level D) start at (4^D-1)/3.
    const float *chalf;    // [P][2][4^D][2]  the leaves' values once more, split for the six-frame kernel's LDS halves:
                           //               half h holds prob[2h], prob[2h+1] of every leaf row (gmg_frame6.hip)
    int cstride;           // bytes per sub-model in cshift (padded to 16)
    int ctot;              // (4^(D+1)-1)/3 nodes per sub-model in crow
    int has_fast;          // cshift/crow valid (W <= 16, D <= 8)
    // direct tables for tiny models (W <= 6), idx = sum_k code(w[k]) << 2k:
    const float *dense;      // [P][4^W]            full windows
    const float *dense_part; // [P][(4^W-4)/3]      partial windows: position j < W-1 at (4^(j+1)-4)/3,
                             //                     idx = sum_{i<=j} code(B[i]) << 2i
    int n_dense_part;
    int has_dense;
};

struct gmg_model {
    // what the values of the model allow (gmg_strings.hip: sums in any order): the smallest exponent field among the non-zero
    // probabilities' logarithms (1 .. 254; 255: all zero), the largest (0: all zero), and whether any value is positive, a NaN
    // or denormal
    int min_exp, max_exp, odd_values;
    GmgDevModel dev;
    void *d_blob;          // single allocation backing every table
    size_t blob_bytes;
};

struct gmg_reads {
    const uint32_t *d_packed;
    const uint64_t *d_off;       // n_reads + 1
    uint32_t *d_tile_read;       // [n_tiles + 1] read containing base t*GMG_TILE
    uint64_t n_reads;
    uint64_t total_bases;
    uint64_t n_tiles;
    uint64_t n_words;            // words addressable at d_packed[0 ..) incl. the trailing guard
    void *d_packed_alloc;        // d_packed - GMG_GUARD_WORDS: the allocation (always library-owned)
    int owns_off;                // d_off allocated by the library
    int uniform_len;             // > 0 when every read has this length (fast read lookup)
    uint64_t max_len, min_len;   // longest / shortest read of the batch
    uint64_t n_over_512;         // reads longer than 512 bases (tile shape of the mg running-sum kernel)
};

struct gmg_null_set {
    float *d_tab;                // [n][252]: [3][64] full windows, then [3][20] partial windows of each model
    int n;
    int min_exp, max_exp, odd_values;    // over all of its models, as in gmg_model
};

struct gmg_segments {
    gmg_segment *d_segs;
    uint64_t *d_out_off;         // n + 1, exclusive prefix of len
    uint64_t n;
    uint64_t total_len;
};

// error plumbing (gmg_api.hip)
int gmg_set_error(int code, const char *fmt, ...);
// first statement of every entry point that touches the device: gmg_init() done, and the calling host thread bound to
// the device it chose (HIP's current device is per thread)
int gmg_enter(const char *who);
#define GMG_HIP(call)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return gmg_set_error(GMG_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                                 __FILE__, __LINE__);                                         \
    } while (0)

// Tuning / test switches (gmg_set_option, include/gmg.h).  Set explicitly through the API, or once at gmg_init() from the
// environment variable GMG_<NAME IN UPPER CASE>; never read from the environment on a scoring path.
enum GmgOpt {
    GMG_OPT_SEG_PLAIN,           // segment kernels: plain descent on the original tables even when a completed tree exists
    GMG_OPT_MG_TILE,             // running-sum kernel: force the 512- or 1504-base tile (0 = by read length)
    GMG_OPT_MG_ONE_STREAM,       // glimmer-mg front half on the caller's stream only
    GMG_OPT_MG_ERR_FLAT,         // error branch on the per-ORF kernel (cross-check of the level kernels)
    GMG_OPT_MG_ERR_CALLS,        // error branch: entries per call array (0 = by batch size); tests force the overflow paths
    GMG_OPT_MG_ERR_CALLS_GROW,   // ... and allow the one retry with larger arrays although the size was forced
    GMG_OPT_ORFS_EXACT_PATH,     // gmg_score_orfs: the any-model path even for the default model shape
    GMG_OPT_TRAIN_SORT_MIN,      // training: smallest set (bases) that takes the sorted counting path (-1 = default)
    GMG_OPT_MG_MAX_ENTRIES,      // most ORFs / starts one gmg_mg_score_reads result may hold (its index fields are 32-bit)
    GMG_OPT_MG_TIMING,           // stage timings of gmg_mg_score_reads on stderr (synchronises between stages)
    GMG_OPT_INGEST_TIMING,       // stage timings of gmg_fasta_ingest on stderr
    GMG_OPT_TRAIN_TIMING,
    GMG_OPT_DIAG,                // ablation kernels; only in builds with -DGMG_ABLATIONS (their output is NOT valid)
    GMG_OPT_STRINGS_FUSED,       // gmg_score_reads_strings: 1 = sums folded into the main pass, 0 = value rows + summing kernel
    GMG_OPT_MG_GENE32,           // glimmer-mg front half, the call's own table as fp32 gene rows with the null model applied where the
                                 // running sums are built: 0 never, 1 with per-read null models and with tiles of two waves or more (default), 2 always
    GMG_OPT_MG_FUSED,            // glimmer-mg front half, default mode: 1 = running sums as a parallel scan + start lists in one kernel
                                 // when the models' values allow it (k_mg_tile_starts), 0 = always the sequential walks
    GMG_OPT_MG_ERR_SKIP,         // glimmer-mg's error branch: 1 = scores as differences of running sums, walks visit their events only (when the
                                 // models' values allow it), 0 = every walk adds up its own sum codon by codon
    GMG_OPT_MG_ORFS_EVENTS,      // glimmer-mg front half, the ORF scan's write pass (k_mg_find_orfs_ev): the reference's steps only at the codons that are in a
                                 // start or stop set -- 2 (default) = found as the set bits of four 32-codon masks in registers, 1 = queued in LDS
                                 // 64 positions at a time; 0 = the steps at every position (k_mg_find_orfs<write>)
    GMG_OPT_MG_ERR_TILE,         // glimmer-mg's error branch: 1 = tile by tile with the running sums in LDS, one lane per event (k_mg_err_tile; needs
                                 // mg_err_skip and sums that are exact in any order), 0 = the level kernels on the walk-order tables in HBM, -1
                                 // (default) = by the batch's size: the tile kernel up to 90 (-i) / 60 (-s) Mbases, the level kernels beyond
                                 // 
    GMG_OPT_MG_ERR_TILE_Q,       // ... tests: calls per level a work-group's slab holds (0 = ET_QCAP; a full slab sends the batch to the level kernels);
                                 // any value but 0 also starts with staging arrays of 64 entries (-1: only that: the kernel repeats with larger ones)
    GMG_OPT_MG_ERR_QONLY,        // glimmer-mg -s on the level kernels: 1 = the running-sum table holds one value per base and strand (16 B/base), 0 = three (48)
    GMG_OPT_MG_ERR_WAVE,         // glimmer-mg's error branch with one wave per (read, strand), running sums and masks in the wave's LDS (needs mg_err_skip
                                 // and sums that are exact in any order): 1 (default) = both passes breadth first without walks (k_mg_err_wcount),
                                 // 2 = both on the stack walker (k_mg_err_wave), 3 = count pass breadth first, write pass on the stack walker
                                 // (cross-checks), 0 = the tile / level kernels
    GMG_OPT_MG_ERR_WAVE_Q,       // ... tests: entries of a wave's call stack (0 = EW_QCAP; a full stack sends the batch to the level kernels)
    GMG_OPT_ORFS_WALK8,          // gmg_score_orfs, events path, the running sums Q: 4 (default) = only the values k_orf_events can ask for, back to back per unit of
                                 // 512 walk steps + a (prefix, need bits) pair per lane (k_orf_walk_sums8p<compact>); 1 = those values at their own bases
                                 // (k_orf_walk_sums8 + k_orf_mark_heads: scattered partial-sector stores, as slow as 2 = every base written); 3 = as 1 with
                                 // the next unit's loads in flight; 0 = k_orf_walk_sums (a lane on every 64th step)
    GMG_OPT_INGEST_SCANS,        // gmg_fasta_ingest: 1 = the first version (two hipcub scans over every byte + k_fa_pack), 0 = block summaries
    GMG_OPT_INGEST_PIECE_MIN,    // gmg_fasta_ingest: inputs of at least this many bytes are uploaded in 16 pieces, every piece parsed and packed as it arrives
    GMG_OPT_MG_ORFS_BITS,        // glimmer-mg front half: 1 = Find_Orfs on bit masks, a wave per window of reads, six lanes per read hopping from stop codon
                                 // to stop codon (k_mg_find_orfs_bits; batches without a read beyond 1,024 bases): bit-identical, an independent
                                 // cross-check -- and not faster; 0 (default) = one lane per read (k_mg_find_orfs / _ev)
    GMG_OPT_ORFS_Q_POISON,       // tests: gmg_score_orfs fills the running-sum array with NaNs before the sums are written
    GMG_OPT_COUNT
};
extern long long g_gmg_opt[GMG_OPT_COUNT];
static inline long long gmg_opt(int k) { return g_gmg_opt[k]; }

// cache of device blocks for scratch and result buffers (gmg_api.hip): hipMalloc / hipFree of GB-sized buffers cost up
// to hundreds of milliseconds now and then; a released block is handed to the next request it fits
void gmg_ingest_trim(void);                       // gmg_ingest.hip: frees its cached page-locked buffers
hipError_t gmg_pool_alloc(void **out, size_t bytes);
void gmg_pool_release(void *p);
void gmg_pool_release_after(void *p, hipStream_t s);   // ... once the work queued on s so far is done

// kernel launchers (gmg_kernels.hip)
int gmg_launch_tile_read(const uint64_t *d_off, uint64_t n_reads, uint64_t n_tiles, uint32_t *d_tile_read,
                         hipStream_t s);
int gmg_launch_frame6(const gmg_model *gene, const gmg_model *nul, const gmg_reads *reads, double *d_out,
                      hipStream_t s);
int gmg_launch_frame6_strided(const gmg_model *gene, const gmg_model *nul, const gmg_reads *reads, double *d_out,
                              uint64_t stride, hipStream_t s);
int gmg_launch_gene6(const gmg_model *gene, const gmg_reads *reads, float *d_gene, hipStream_t s);
int gmg_launch_gene6_full(const gmg_model *gene, const gmg_reads *reads, float *d_gene, uint64_t gstride, hipStream_t s);
// the same table for a batch made of consecutive GROUPS of reads, group g = reads [group_read[g], group_read[g+1]) (HOST array,
// n_groups + 1 entries, group_read[n_groups] = n_reads) scored by models[g]: one launch of the main pass for all groups
int gmg_launch_gene6_groups(const gmg_model *const *models, const uint64_t *group_read, int n_groups, const gmg_reads *reads,
                            float *d_gene, uint64_t gstride, hipStream_t s);
int gmg_launch_strings(const gmg_model *m, const gmg_reads *reads, float *d_vals, uint64_t *tail_start, hipStream_t s);
int gmg_launch_strings_sum(const gmg_model *m, const gmg_reads *reads, double *d_sums, uint64_t *tail_start, hipStream_t s);
int gmg_launch_seg_frame(const gmg_model *m, const gmg_reads *r, const gmg_segments *sg, int frame,
                         double *d_out, hipStream_t s);
int gmg_launch_seg_cum(const gmg_model *m, const gmg_reads *r, const gmg_segments *sg, int frame0,
                       double *d_out, double *d_sums, hipStream_t s);
int gmg_launch_seg_partial(const gmg_model *m, const gmg_reads *r, const gmg_segments *sg, int frame,
                           double *d_out, hipStream_t s);
int gmg_launch_all_frame(const gmg_model *m, const gmg_reads *r, const gmg_segments *sg,
                         const uint32_t *d_prefix, const int32_t *d_frame, double *d_af, hipStream_t s);
int gmg_launch_windows(const gmg_model *m, const uint8_t *d_windows, const int32_t *d_frames, uint64_t n,
                       float *d_dist4, double *d_prob, hipStream_t s);

#endif
