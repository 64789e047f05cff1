// gmg_mg.hip -- glimmer-mg's front half on gfx950: from packed reads to the start lists that
// Score_Orfs_Errors (src/Glimmer/glimmer-mg.cc:1605-1689) hands to Add_Events_*.  Linear sequences, no ignore
// regions.  Step 3 below is the default mode (Allow_Indels = Allow_Subs = false, glimmer-mg.cc:100-102);
// with -i / -s it is replaced by the error branch further down (k_mg_err_*: Score_Indels and the recursive
// Score_Orf_Starts, one lane per call, level by level; scores as differences of running sums when those are exact).
//
//   1. gmg_launch_frame6        Score_All_Frames (glimmer-mg.cc:1468-1510): Frame_Scores[6][total] in HBM
//   2. k_mg_find_orfs<count>    Find_Orfs (glimmer_base.cc:638-779), one lane per read: ORFs per read (beside step 1 on a second stream)
//      exclusive scan, k_mg_find_orfs_ev: the Orf_t records + lo / hi of Score_Orf_Starts + the number of starts each will push
//   3. k_mg_tile_starts         Cumulative_Frame_Score (glimmer-mg.cc:561-604) for EVERY possible ORF as three segmented parallel
//      scans per tile and strand, and Score_Orf_Starts on top of them: the start lists in push order (scores = the scan's value at
//      every start codon), boost, first_j, best score, threshold.  Needs sums that are exact in any order (checked per batch from
//      the models' exponent range); otherwise, and as the cross-check of the tests, the round-1 kernels:
//      k_mg_cum_tiled / k_mg_cum   one lane per stop-to-stop region (per (read, strand)): sequential sums, reset at the in-class stops
//      k_mg_starts<count / write>  one lane per ORF: its starts from the stored sums
// Integer / byte work except the one running sum; no table of Save_Prev_Stops (glimmer-mg.cc:675-729) is
// materialised -- the ORF scan already knows the previous / next in-frame stop of every ORF it emits:
//   forward ORF ended by the stop whose last base is i (class c = i % 3):
//       hi = end_point = i - 2;   lo = Fwd_Prev_Stops[i-3] + 1 = (last forward stop index of class c, or
//       {0, 1, -1}[c] when there is none) + 1
//   reverse ORF closed by the reverse stop whose last base is i:   lo = orf_stop + 3;
//       hi = Rev_Next_Stops[orf_stop+2] + 1 = (i - 2) + 1        (the closing stop starts at i-2, same class)
//   reverse ORF closed by the end of the read (Finish_Orfs):  e = orf_stop + 2;
//       hi = e + 1 if e >= n, else {n-1, n-2, n}[(n-1-e) % 3] + 1 (no further stop in that class)
// tests/test_gpu_mg.py checks these against the oracle's literal tables and the reference's own output.

#include "gmg_device.h"


#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <new>
#include <type_traits>
#include <vector>

struct gmg_mg_result {
    gmg_mg_orf *d_orfs;
    gmg_start *d_starts;
    gmg_start_errors *d_errs;    // error branch only: parallel to d_starts
    uint64_t *d_read_orf_off;    // [n_reads + 1]
    uint64_t n_reads, n_orfs, n_starts;
};

// One Score_Orf_Starts call waiting to be walked, 32 bytes (the count passes write and read 100 M + 77 M of them per 1 M reads
// with -i: as 56-byte structs with one field per value they were 30 GB of the call's traffic).  Reads of the level kernels are
// shorter than 2,040 bases, so every position fits 12 bits:
//   w[0] suffix_score;  w[1] key (39 bits: three 13-bit fields) | (end_point + 8) << 40 | suffix_j << 52;
//   w[2] the read's first base (40 bits) | its length << 40 | (level | forward strand << 2) << 51  -- 0: an empty entry;
//   w[3] orf | e0 << 32 | e1 << 46   (Error_t entries of the path: (pos + 8) << 2 | type, 14 bits each)
struct __attribute__((aligned(16))) MgCall { unsigned long long w[4]; };
struct MgOrfAgg { unsigned long long best, ext_a, ext_b; uint32_t cnt, m0; };

struct MgTile { uint64_t w0; uint32_t first, nfit, span, pad; };   // reads [first, first + nfit), bases [w0, w0 + span)

struct MgArgs {
    const uint32_t *packed;
    const uint64_t *read_off;
    const uint32_t *tile_read;   // read containing base t * GMG_TILE (gmg_reads)
    uint64_t n_reads, total;
    const double *fs;            // Frame_Scores [6][fs_stride]
    uint64_t fs_stride;          // = total for a caller's table; the call's own table pads its rows to 128-byte lines
    // GENE32 form of the table (the call's own, option mg_gene32): the gene model's values alone as fp32, [6][fs_stride]; the
    // null model's value is subtracted where the running sums are built -- which is also where a per-read null model costs
    // nothing (glimmer-mg -c: Update_Meta_Null_ICM, glimmer-mg.cc:2050-2068)
    const float *gene32;
    const float *null_tab;       // [n_null][252]: [3][64] full windows of a (3,2,3) null model, idx = sum code(w[k]) << 2k, then
                                 // [3][20] partial windows: position j < 2 at (4^(j+1)-4)/3
    const uint32_t *read_null;   // [n_reads] null model of every read, or NULL: model 0 for all
    const int32_t *read_isl;     // [n_reads] Ignore_Score_Len of every read (Set_Ignore_Score_Len per read, :2067), or NULL
    double *cum;                 // [2][total]: score[j-1] of the ORF for which this base is in frame (k_mg_cum*)
    // tiling of k_mg_cum_tiled: uniform batches take reads_per_tile reads per block; ragged batches the reads
    // that start inside the block's window of tile_window bases
    int uniform_len, reads_per_tile;
    int tile_reads_max;          // most reads a tile takes: MG_TILE_READS, or MT_NC when the fused kernel holds a null model per read in LDS
    uint32_t uniform_magic;      // ceil (2^32 / uniform_len): b / uniform_len = __umulhi (b, magic) for b < 2^16
    uint64_t tile_window, n_tiles;
    int lanes_only_unfit;        // k_mg_cum: skip the reads the tiled kernel has done
    const struct MgTile *tiles;  // ragged batches: the non-empty tiles, one entry each (k_mg_tile_table + select)
    const uint32_t *n_tiles_dev; // ... and how many there are (stays on the device: no host round trip before the launch)
    const struct MgTile *windows;// ragged batches: the tile of EVERY window (k_mg_tile_table), for the unfit test
    uint32_t *unfit;             // [n_reads] reads no tile took + their number in unfit_n (k_mg_unfit_list)
    uint32_t *unfit_n;
    int tile_cap;                // bases per tile of the tiled kernel
    // codon tests as 64-bit sets over idx6 = code(oldest) << 4 | code << 2 | code(newest)
    uint64_t fwd_start, rev_start, fwd_stop, rev_stop;
    uint64_t fwd_stop_nat, rev_stop_nat;     // the stop sets over v = code(first base) | code << 2 | code(last) << 4 (k_mg_tile_starts)
    int8_t which[64];            // index of the first matching start codon, -1 for none (Codon_t::Can_Be)
    int min_gene_len, allow_truncated, ignore_score_len;
    double start_threshold;
    uint32_t *read_cnt;          // [n_reads + 1] ORFs per read -> exclusive scan in read_orf_off
    const uint64_t *read_orf_off;
    gmg_mg_orf *orfs;
    uint64_t n_orfs;
    int count_starts;            // k_mg_find_orfs<write> also counts every ORF's starts (default mode, lowest j within 64 codons): no k_mg_starts<count>
    uint32_t *orf_cnt;           // [n_orfs + 1] starts per ORF
    const uint64_t *start_off;   // its exclusive scan
    gmg_start *starts;
    // the error branch (glimmer-mg -i / -s), k_mg_err_level / k_mg_err_flat
    int err_mode;                // 0 off, 1 indels, 2 substitutions
    int min_indel_orf_len, indel_q_thr, indel_max;
    double indel_suffix_thr;
    const uint8_t *qual;         // [total] Quality_Values after Set_ / Clean_Quality_454 (k_mg_quality)
    const double *pen;           // [256] Score_Indels' score_penalty by quality value (host libm, like the reference)
    double pass_stop[4];         // Pass_Stop_Penalty by (second base is a/t) * 2 + (third base is a/t)
    gmg_start_errors *errs;
    uint64_t *keys;              // [n_starts] order of a start inside its ORF's list (k_mg_err_level), ascending = push order
    uint8_t *read_fit;           // [n_reads] 1: the level kernels take the read, 0: k_mg_err_flat does (too long for the order keys)
    uint32_t *err_flag;          // set when a call array is full (the call then repeats on k_mg_err_flat)
    // level by level (k_mg_err_level): the calls of level 1 and 2 wait in two arrays
    struct MgCall *calls[2];
    unsigned long long *n_calls; // [2] entries used
    unsigned long long *tile_ctr;// [6] next tile of calls per (pass, level) (k_mg_err_level)
    uint64_t call_cap;           // entries per array
    struct MgOrfAgg *agg;        // [n_orfs] what the calls of an ORF add up to
    uint32_t *fill;              // [n_orfs] write pass: slots handed out inside the ORF's slice
    uint32_t *acc_bits;          // [n_orfs / 32 + 1] bit i: ORF i is accepted (k_mg_err_verdict): what the write passes of levels 1 and 2 ask per call
    const double *walk;          // [6][walk_stride] Frame_Scores in walking order (k_mg_walk_tables), or, pfx: their running sums
                                 // inside every read (k_mg_walk_prefix)
    int pfx;                     // the level kernels take score[j] as a difference of two running sums and skip the codons nothing happens at
    int qonly;                   // ... and the table holds ONE value per base and strand: the running sum of the base's own class over the steps
                                 // before it (-s: a walk needs score[j-1] at its start codons and the region's total, nothing inside a codon)
    const uint8_t *run_q, *run_n;// [2][walk_stride] by strand and walk index: codons from here on at which nothing can happen (k_mg_run_tables),
                                 // with / without the low-quality bases as events
    const uint8_t *walk_q;       // [total + 8] the qualities, last base first (forward walks; reverse walks read a.qual)
    uint64_t walk_stride;
    int ew_slack;                // k_mg_err_wave / _wcount: gene32 and qual have 64 spare entries on both sides (unpredicated loads)
    int q454;                    // ... Set_Quality_454 is computed in the kernel from the bases (no quality file): a.qual is not read
};

// Ch_Mask (src/Common/gene.cc:954-995)
static unsigned mg_ch_mask(int ch)
{
    switch (ch | 0x20) {
    case 'a': return 0x1; case 'c': return 0x2; case 'g': return 0x4; case 't': return 0x8;
    case 'r': return 0x5; case 'y': return 0xA; case 's': return 0x6; case 'w': return 0x9;
    case 'm': return 0x3; case 'k': return 0xC; case 'b': return 0xE; case 'd': return 0xD;
    case 'h': return 0xB; case 'v': return 0x7; case 'n': return 0xF;
    }
    return 0;
}

// sequential reader of 2-bit codes, one 32-bit word per 16 bases; DIR = +1 / -1
template <int DIR>
struct BaseStream {
    const uint32_t *packed;
    uint64_t g;
    uint32_t w;
    __device__ __forceinline__ void init(const uint32_t *p, uint64_t g0) { packed = p; g = g0; w = p[g0 >> 4]; }
    __device__ __forceinline__ int next()
    {
        const int code = (int)((w >> (2u * (unsigned)(g & 15))) & 3u);
        if (DIR > 0) { g++; if ((g & 15) == 0) w = packed[g >> 4]; }        // the guard words make one word past the end readable
        else { if ((g & 15) == 0) w = packed[(g >> 4) - 1]; g--; }
        return code;
    }
};

// ---------------------------------------------------------------------------------------------------
// Find_Orfs: one lane per read.  Class c = i % 3 is static in the unrolled loop, so the per-class state
// (glimmer_base.cc:647-652) lives in registers.
// ---------------------------------------------------------------------------------------------------
struct MgClass {
    int first_fwd_start, last_rev_start, prev_fwd_stop, prev_rev_stop;
    int fwd_last;                // Save_Prev_Stops' last_stops for the forward scan (glimmer-mg.cc:685-699)
    // the number of starts Score_Orf_Starts will push for the ORF that is open in this class (count_starts), counted while the
    // scan passes them.  Forward: position j is counted from the ORF's stop codon, which comes later -- the start codons of the
    // last k0 = 1 + lowest_j / 3 codons wait in a bit mask, older ones are counted.  Reverse: j is counted from the previous
    // stop: start codons from a threshold position on.
    unsigned long long fwd_recent;
    int fwd_older, rev_cnt, rev_from;
};

template <bool WRITE>
__global__ __launch_bounds__(256) void k_mg_find_orfs(MgArgs a)
{
    // (the counting write pass stores every ORF's number of starts; the extra element the scan wants is zeroed here -- a memset of its
    // own waited 0.16 ms for a slot beside the partial-window pass, on the critical path)
    if (WRITE && a.count_starts && blockIdx.x == 0 && threadIdx.x == 0) a.orf_cnt[a.n_orfs] = 0;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.n_reads; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t off = a.read_off[r];
        const int n = (int)(a.read_off[r + 1] - off);
        const int mgl = a.min_gene_len;
        const bool trunc = a.allow_truncated != 0;
        uint32_t cnt = 0;
        const uint64_t orf0 = WRITE ? a.read_orf_off[r] : 0;
        gmg_mg_orf *out = WRITE ? a.orfs + orf0 : nullptr;
        const bool counting = WRITE && a.count_starts;
        int j_lo = mgl - 3 > 1 ? mgl - 3 : 1;           // lowest j of a start, as in mg_starts_one
        j_lo = (j_lo + 2) / 3 * 3;
        const int k0 = 1 + j_lo / 3;                    // (<= 64 when count_starts is set)

        auto emit = [&](int stop_position, int frame, int gene_len, int orf_len, int lo, int hi, int n_real) __attribute__((always_inline)) {
            if (gene_len >= mgl || (a.err_mode && orf_len >= a.min_indel_orf_len)) {   // glimmer_base.cc:494,528,806
                if (WRITE) {
                    gmg_mg_orf o;
                    o.read = (uint32_t)r; o.frame = frame; o.stop_position = stop_position;
                    o.orf_len = orf_len; o.gene_len = gene_len; o.lo = lo; o.hi = hi;
                    o.first_j = 0; o.start_begin = 0; o.n_starts = 0; o.accepted = 0; o.orf_is_truncated = 0;
                    o.reserved = 0; o.best_score = -DBL_MAX;
                    out[cnt] = o;
                    if (counting) {                     // + the truncated start (mg_starts_one: has_trunc)
                        const int m = hi - lo;
                        const bool tr = trunc && (frame > 0 ? lo < 3 : n - (hi - 1) < 3);
                        const int jmax = m >= 1 ? (m - 1) / 3 * 3 : -1;
                        a.orf_cnt[orf0 + cnt] = (uint32_t)((m > 0 ? n_real : 0) + (tr && jmax >= j_lo ? 1 : 0));
                    }
                }
                cnt++;
            }
        };
        // Do_Fwd_Stop_Codon (glimmer_base.cc:460-504) + Handle_First_Forward_Stop, linear (:970-982)
        auto fwd_stop = [&](int i, MgClass &S, int cls) __attribute__((always_inline)) {
            int gene_len, orf_len;
            if (S.prev_fwd_stop == 0) {
                const int pos = i - 1;
                orf_len = pos - 1;                      // first_base = 1
                orf_len -= orf_len % 3;
                gene_len = S.first_fwd_start == INT_MAX ? 0 : pos - S.first_fwd_start;
                if (trunc && gene_len < mgl) gene_len = orf_len;
            } else {
                gene_len = (i - S.first_fwd_start) - 1;
                orf_len = i - S.prev_fwd_stop - 4;
            }
            emit(i - 1, 1 + (cls + 1) % 3, gene_len, orf_len, S.fwd_last + 1, i - 2, S.fwd_older);
            S.first_fwd_start = INT_MAX;
            S.prev_fwd_stop = i - 1;
            S.fwd_older = 0;
            S.fwd_recent = 0;
        };
        // Do_Rev_Stop_Codon (glimmer_base.cc:506-537) + Handle_First_Reverse_Stop (:989-1015)
        auto rev_stop = [&](int i, MgClass &S, int cls) __attribute__((always_inline)) {
            int gene_len, orf_stop = 0;
            if (S.prev_rev_stop == 0) {
                if (!trunc) gene_len = 0;
                else {
                    orf_stop = (i - 1) % 3;
                    if (orf_stop > 0) orf_stop -= 3;
                    gene_len = S.last_rev_start - orf_stop;
                }
            } else {
                orf_stop = S.prev_rev_stop;
                gene_len = S.last_rev_start - orf_stop;
            }
            emit(orf_stop, -1 - (cls + 1) % 3, gene_len, i - orf_stop - 4, orf_stop + 3, i - 1, S.rev_cnt);
            S.last_rev_start = 0;
            S.prev_rev_stop = i - 1;
            S.rev_cnt = 0;
            S.rev_from = (i - 1) + 4 + j_lo;            // j = i - 4 - orf_stop >= lowest j
        };

        if (n >= mgl) {                                 // glimmer_base.cc:676-677
            MgClass cs[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                cs[c].first_fwd_start = INT_MAX;
                cs[c].last_rev_start = cs[c].prev_fwd_stop = cs[c].prev_rev_stop = 0;
                cs[c].fwd_recent = 0;
                cs[c].fwd_older = cs[c].rev_cnt = 0;
                cs[c].rev_from = (c == 0 ? -1 : c == 1 ? 0 : -2) + 4 + j_lo;      // the virtual stop in front of the read (Handle_First_Reverse_Stop)
            }
            cs[0].fwd_last = 0; cs[1].fwd_last = 1; cs[2].fwd_last = -1;
            BaseStream<1> bs;
            bs.init(a.packed, off);
            uint32_t idx6 = 0;
            for (int i0 = 0; i0 < n; i0 += 3) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const int i = i0 + c;
                    if (i < n) {
                        idx6 = ((idx6 << 2) | (uint32_t)bs.next()) & 63u;
                        if (i >= 2) {                   // a Codon_t with an empty position matches nothing (gene.cc:56,85)
                            const uint64_t bit = 1ull << idx6;
                            if (counting) {             // the codon that is now k0 codons back leaves the mask.  (The stop codon that ends a
                                                        // region is no start, even when the start set holds it: "-A nnn")
                                cs[c].fwd_older += (int)((cs[c].fwd_recent >> (k0 - 1)) & 1ull);
                                cs[c].fwd_recent = (cs[c].fwd_recent << 1) | ((a.fwd_start & ~a.fwd_stop & bit) ? 1ull : 0ull);
                                if ((a.rev_start & ~a.rev_stop & bit) && i >= cs[c].rev_from) cs[c].rev_cnt++;
                            }
                            if ((a.fwd_start & bit) && cs[c].first_fwd_start == INT_MAX) cs[c].first_fwd_start = i - 1;
                            if (a.rev_start & bit) cs[c].last_rev_start = i - 1;
                            if (a.fwd_stop & bit) { fwd_stop(i, cs[c], c); cs[c].fwd_last = i; }
                            if (a.rev_stop & bit) rev_stop(i, cs[c], c);
                        }
                    }
                }
            }
            // Finish_Orfs (glimmer_base.cc:783-817) + Handle_Last_Reverse_Stop, linear (:1053-1066)
#pragma unroll
            for (int fr = 0; fr < 3; fr++) {
                const MgClass &S = cs[fr];
                const int orf_stop = S.prev_rev_stop == 0 ? (fr == 0 ? -1 : fr == 1 ? 0 : -2) : S.prev_rev_stop;
                int orf_len = n - orf_stop - 2;
                orf_len -= orf_len % 3;
                int gene_len = S.last_rev_start == 0 ? 0 : S.last_rev_start - orf_stop;
                if (trunc && gene_len < mgl) gene_len = orf_len;
                const int e = orf_stop + 2;             // Rev_Next_Stop (glimmer-mg.cc:1436-1445), no stop left in the class
                int hi;
                if (e >= n) hi = e + 1;
                else { const int rc = (n - 1 - e) % 3; hi = (rc == 0 ? n - 1 : rc == 1 ? n - 2 : n) + 1; }
                emit(orf_stop, -1 - (fr + 1) % 3, gene_len, orf_len, orf_stop + 3, hi, S.rev_cnt);
            }
            if (trunc)                                  // glimmer_base.cc:765-776: 3 bp past the end count as stops
                for (int i = n; i < n + 3; i++) {
                    const int c = i % 3;
                    if (counting) {                     // (the virtual stop takes a place in the mask like a real one; static
                                                        // indices, or the class states would live in scratch memory)
                        if (c == 0) { cs[0].fwd_older += (int)((cs[0].fwd_recent >> (k0 - 1)) & 1ull); cs[0].fwd_recent <<= 1; }
                        else if (c == 1) { cs[1].fwd_older += (int)((cs[1].fwd_recent >> (k0 - 1)) & 1ull); cs[1].fwd_recent <<= 1; }
                        else { cs[2].fwd_older += (int)((cs[2].fwd_recent >> (k0 - 1)) & 1ull); cs[2].fwd_recent <<= 1; }
                    }
                    if (c == 0) fwd_stop(i, cs[0], 0);
                    else if (c == 1) fwd_stop(i, cs[1], 1);
                    else fwd_stop(i, cs[2], 2);
                }
        }
        if (!WRITE) a.read_cnt[r] = cnt;
    }
}

// ---------------------------------------------------------------------------------------------------
// k_find_orfs_general: Find_Orfs in full (glimmer_base.cc:638-817) -- ignore regions (glimmer3 -i: the scan stops at a region,
// closes the reverse ORFs and starts anew behind it, :689-731) and circular sequences (glimmer-mg -r: two bases of overhang, the
// first forward stop of a frame looks back through the sequence's end, Wrap_Through_Front :2854-2900, the last reverse ORFs look
// on through its front, Wrap_Around_Back :2793-2850).  One lane per sequence, the reference's own order of steps: these are
// genome-scale inputs of a few sequences, nothing here is hot.  Records carry the Orf_t fields (lo / hi = 0: the start scan of
// glimmer-mg's front half does not take such ORFs).  fail: set where the reference's assert (pos > 0) in Wrap_Around_Back fires.
// ---------------------------------------------------------------------------------------------------
struct FgClass { int first_fwd_start, last_rev_start, prev_fwd_stop, prev_rev_stop; };

template <bool WRITE>
__global__ __launch_bounds__(64) void k_find_orfs_general(MgArgs a, const int circular, const int n_ign, const int32_t *ign_lo, const int32_t *ign_hi,
                                                          uint32_t *fail)
{
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.n_reads; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t off = a.read_off[r];
        const int L = (int)(a.read_off[r + 1] - off);
        const int mgl = a.min_gene_len;
        const bool trunc = a.allow_truncated != 0;
        uint32_t cnt = 0;
        gmg_mg_orf *out = WRITE ? a.orfs + a.read_orf_off[r] : nullptr;
        auto code = [&](int i) __attribute__((always_inline)) -> uint32_t {
            const uint64_t g = off + (uint64_t)(i < L ? i : i - L);
            return (a.packed[g >> 4] >> (2u * (unsigned)(g & 15))) & 3u;
        };
        auto emit = [&](int stop_position, int frame, int gene_len, int orf_len) __attribute__((always_inline)) {
            if (gene_len >= mgl || (a.err_mode && orf_len >= a.min_indel_orf_len)) {
                if (WRITE) {
                    gmg_mg_orf o;
                    o.read = (uint32_t)r; o.frame = frame; o.stop_position = stop_position;
                    o.orf_len = orf_len; o.gene_len = gene_len; o.lo = 0; o.hi = 0;
                    o.first_j = 0; o.start_begin = 0; o.n_starts = 0; o.accepted = 0; o.orf_is_truncated = 0;
                    o.reserved = 0; o.best_score = -DBL_MAX;
                    out[cnt] = o;
                }
                cnt++;
            }
        };
        if (L >= mgl) {
            const int n = L + (circular ? 2 : 0);
            bool hit_ignore = false, ignoring = false;
            int first_base = 1, ignore_sub = 0;
            int ignore_start = n_ign > 0 ? ign_lo[0] : INT_MAX, ignore_stop = n_ign > 0 ? ign_hi[0] : INT_MAX;
            FgClass cs[3];
#pragma unroll
            for (int c = 0; c < 3; c++) { cs[c].first_fwd_start = INT_MAX; cs[c].last_rev_start = cs[c].prev_fwd_stop = cs[c].prev_rev_stop = 0; }
            // Wrap_Through_Front: the forward ORF whose stop codon begins at pos (1-based) looked at backwards through the sequence's end
            auto wrap_front = [&](int pos, int &gene_len, int &orf_len) {
                int start_at = -1, s_ = (pos - 1) % 3, i;
                const int check_len = L + s_ - pos - 4;
                uint32_t idx6 = 0;
                for (i = 0; i < check_len; i += 3) {
                    for (int j = 0; j < 3; j++) {
                        s_--;
                        if (s_ < 0) s_ += L;
                        idx6 = (idx6 >> 2) | code(s_) << 4;          // Reverse_Shift_In: the new base becomes the codon's first
                    }
                    if ((a.fwd_stop >> idx6) & 1ull) break;
                    if ((a.fwd_start >> idx6) & 1ull) start_at = i + 3;
                }
                orf_len = i + 3 * ((pos - 1) / 3);
                gene_len = start_at == -1 ? 0 : start_at + 3 * ((pos - 1) / 3);
            };
            // Wrap_Around_Back: the reverse ORF behind the stop codon at pos goes on through the sequence's front, frame wfr there
            auto wrap_back = [&](int wfr, int pos, int &gene_len, int &orf_len) {
                int start_at = -1, orf_add = 0, frame = 0;
                if (pos <= 0) { atomicOr(fail, 1u); gene_len = orf_len = 0; return; }
                uint32_t idx6 = 0;
                for (int i = 0; i < pos - 1; i++) {
                    idx6 = ((idx6 << 2) | code(i)) & 63u;
                    const bool full = i >= 2;
                    if (frame == wfr) {
                        if (full && ((a.rev_stop >> idx6) & 1ull)) { orf_add = i - 2; break; }
                        orf_add = i + 1;
                    }
                    if (frame == wfr && full && ((a.rev_start >> idx6) & 1ull)) start_at = i + 1;
                    frame = frame == 2 ? 0 : frame + 1;
                }
                orf_len = orf_add + L - pos - 2;
                orf_len -= orf_len % 3;
                gene_len = start_at == -1 ? 0 : start_at + L - pos - 2;
            };
            auto fwd_stop = [&](int i, FgClass &S, int cls) {
                int gene_len, orf_len;
                if (S.prev_fwd_stop == 0) {             // Handle_First_Forward_Stop (:946-985)
                    const int pos = i - 1;
                    if (circular && !hit_ignore) {
                        wrap_front(pos, gene_len, orf_len);
                        if (gene_len == 0 && S.first_fwd_start != INT_MAX) gene_len = pos - S.first_fwd_start;
                    } else {
                        orf_len = pos - first_base;
                        orf_len -= orf_len % 3;
                        gene_len = S.first_fwd_start == INT_MAX ? 0 : pos - S.first_fwd_start;
                        if (trunc && gene_len < mgl) gene_len = orf_len;
                    }
                } else {
                    gene_len = i - S.first_fwd_start - 1;
                    orf_len = i - S.prev_fwd_stop - 4;
                }
                emit(i - 1, 1 + (cls + 1) % 3, gene_len, orf_len);
                S.first_fwd_start = INT_MAX;
                S.prev_fwd_stop = i - 1;
            };
            auto rev_stop = [&](int i, FgClass &S, int cls) {
                int gene_len, orf_stop = 0;
                if (S.prev_rev_stop == 0) {             // Handle_First_Reverse_Stop (:989-1015)
                    if (hit_ignore || !trunc) gene_len = 0;
                    else {
                        orf_stop = (i - 1) % 3;
                        if (orf_stop > 0) orf_stop -= 3;
                        gene_len = S.last_rev_start - orf_stop;
                    }
                } else {
                    orf_stop = S.prev_rev_stop;
                    gene_len = S.last_rev_start - orf_stop;
                }
                emit(orf_stop, -1 - (cls + 1) % 3, gene_len, i - orf_stop - 4);
                S.last_rev_start = 0;
                S.prev_rev_stop = i - 1;
            };
            auto finish = [&](bool use_wrap, int last_position) {       // Finish_Orfs (:783-817) + Handle_Last_Reverse_Stop (:1019-1072)
#pragma unroll
                for (int fr = 0; fr < 3; fr++) {
                    const FgClass &S = cs[fr];
                    const int orf_stop = S.prev_rev_stop == 0 ? (fr == 0 ? -1 : fr == 1 ? 0 : -2) : S.prev_rev_stop;
                    int gene_len, orf_len;
                    if (use_wrap) {
                        wrap_back((3 + fr - (L % 3)) % 3, S.prev_rev_stop, gene_len, orf_len);
                        if (gene_len == 0 && S.last_rev_start > 0) gene_len = S.last_rev_start - S.prev_rev_stop;
                    } else {
                        orf_len = last_position - orf_stop - 2;
                        orf_len -= orf_len % 3;
                        gene_len = S.last_rev_start == 0 ? 0 : S.last_rev_start - orf_stop;
                        if (trunc && gene_len < mgl) gene_len = orf_len;
                    }
                    emit(orf_stop, -1 - (fr + 1) % 3, gene_len, orf_len);
                }
            };
            uint32_t idx6 = 0;
            int fill = 0;                               // bases in the codon register (it is cleared behind an ignore region)
            for (int i0 = 0; i0 < n; i0 += 3) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const int i = i0 + c;
                    if (i >= n) continue;
                    if (i == ignore_start) {
                        finish(false, i);
                        hit_ignore = ignoring = true;
                    } else if (i == ignore_stop) {
#pragma unroll
                        for (int k = 0; k < 3; k++) { cs[k].first_fwd_start = INT_MAX; cs[k].last_rev_start = cs[k].prev_fwd_stop = cs[k].prev_rev_stop = 0; }
                        idx6 = 0; fill = 0;
                        first_base = i + 1;
                        ignoring = false;
                        ignore_sub++;
                        if (ignore_sub >= n_ign) ignore_start = ignore_stop = INT_MAX;
                        else { ignore_start = ign_lo[ignore_sub]; ignore_stop = ign_hi[ignore_sub]; }
                    }
                    if (!ignoring) {
                        idx6 = ((idx6 << 2) | code(i)) & 63u;
                        if (fill < 3) fill++;
                        if (fill == 3) {                // (a Codon_t with an empty position matches nothing)
                            const uint64_t bit = 1ull << idx6;
                            if ((a.fwd_start & bit) && cs[c].first_fwd_start == INT_MAX) cs[c].first_fwd_start = i - 1;
                            if (a.rev_start & bit) cs[c].last_rev_start = i - 1;
                            if (a.fwd_stop & bit) fwd_stop(i, cs[c], c);
                            if (a.rev_stop & bit) rev_stop(i, cs[c], c);
                        }
                    }
                }
            }
            finish(circular != 0, L);
            if (!circular && trunc)                     // :765-776: 3 bp past the end count as forward stops
                for (int i = n; i < n + 3; i++) {
                    if (ignoring) continue;
                    const int c = i % 3;
                    if (c == 0) fwd_stop(i, cs[0], 0);
                    else if (c == 1) fwd_stop(i, cs[1], 1);
                    else fwd_stop(i, cs[2], 2);
                }
        }
        if (!WRITE) a.read_cnt[r] = cnt;
    }
}

// ---------------------------------------------------------------------------------------------------
// k_mg_find_orfs_ev: the write pass of Find_Orfs in two phases per 64 positions.  In k_mg_find_orfs a wave pays at every position
// for the stop-codon code of whichever lane met one (88 vector instructions per position and wave; one codon in five is a start or
// a stop codon of either strand).  Here phase A only LOOKS -- four set bits per codon from a 64-entry table, the codons that
// have any queued in LDS (one column per lane) -- and phase B runs the reference's per-codon steps over the queue, every lane busy
// with an event of its own.  The class of an event is data there: its state is picked from the three classes' registers and put back.
// The start count of k_mg_find_orfs (count_starts) catches up with the codons between two events of a class in one step.
// Same records, same order; the count pass stays k_mg_find_orfs<false> (30 registers: it runs beside the six-frame kernel).
// ---------------------------------------------------------------------------------------------------
#define MG_EV_CH 64
#define MG_EV_LANES 64            // one-wave work-groups (8 KB of LDS: one of them fits beside the six-frame kernel's work-group on a CU)
// MASKS (the default): phase A without a queue -- the four tests of 32 consecutive codons as four 32-bit masks in registers (16 look-ups
// of a 256-entry table: four bases -> two codons x four tests; the word-aligned 32 bases that hold them), phase B over the set bits of
// their union.  A sixth of phase A's instructions, and 1 KB of LDS instead of 8: up to four of these waves fit beside the six-frame
// kernel's work-group on a CU (registers), so most of the pass runs in its shadow.
template <bool MASKS>
__global__ __launch_bounds__(MG_EV_LANES, 5) void k_mg_find_orfs_ev(MgArgs a)       // (five waves per SIMD = 96 registers: what the six-frame kernel leaves free)
{
    __shared__ uint16_t s_q[MASKS ? 1 : MG_EV_CH][MG_EV_LANES];     // events of the chunk: position in the chunk | set bits << 8
    __shared__ uint8_t s_evt[64];                       // by codon index: bit 0 forward start, 1 reverse start, 2 forward stop, 3 reverse stop
    __shared__ uint32_t s_tab[MASKS ? 256 : 1];         // four bases -> byte m: the two codons' membership in set m (fs, rs, ft, rt)
    if (threadIdx.x < 64) {
        const uint64_t bit = 1ull << threadIdx.x;
        s_evt[threadIdx.x] = (uint8_t)(((a.fwd_start & bit) ? 1 : 0) | ((a.rev_start & bit) ? 2 : 0) | ((a.fwd_stop & bit) ? 4 : 0) | ((a.rev_stop & bit) ? 8 : 0));
    }
    if (MASKS)
        for (uint32_t key = threadIdx.x; key < 256; key += MG_EV_LANES) {
            const uint32_t c0 = (key & 3u) << 4 | (key & 12u) | ((key >> 4) & 3u), k1 = key >> 2;
            const uint32_t c1 = (k1 & 3u) << 4 | (k1 & 12u) | ((k1 >> 4) & 3u);
            const uint64_t sets[4] = {a.fwd_start, a.rev_start, a.fwd_stop, a.rev_stop};
            uint32_t e = 0;
#pragma unroll
            for (int m = 0; m < 4; m++) e |= (uint32_t)(((sets[m] >> c0) & 1ull) | ((sets[m] >> c1) & 1ull) << 1) << (8 * m);
            s_tab[key] = e;
        }
    __syncthreads();
    const uint32_t lane = threadIdx.x;
    const int mgl = a.min_gene_len;
    const bool trunc = a.allow_truncated != 0;
    const bool counting = a.count_starts != 0;
    if (counting && blockIdx.x == 0 && threadIdx.x == 0) a.orf_cnt[a.n_orfs] = 0;       // (the scan's extra element: see k_mg_find_orfs)
    int j_lo = mgl - 3 > 1 ? mgl - 3 : 1;
    j_lo = (j_lo + 2) / 3 * 3;
    const int k0 = 1 + j_lo / 3;                        // (<= 64: mg_run)
    const unsigned long long k0_mask = k0 >= 64 ? ~0ull : (1ull << k0) - 1ull;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.n_reads; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t off = a.read_off[r];
        const int n = (int)(a.read_off[r + 1] - off);
        uint32_t cnt = 0;
        const uint64_t orf0 = a.read_orf_off[r];
        gmg_mg_orf *out = a.orfs + orf0;
        if (n < mgl) continue;                          // glimmer_base.cc:676-677

        auto emit = [&](int stop_position, int frame, int gene_len, int orf_len, int lo, int hi, int n_real) __attribute__((always_inline)) {
            if (gene_len >= mgl || (a.err_mode && orf_len >= a.min_indel_orf_len)) {   // glimmer_base.cc:494,528,806
                gmg_mg_orf o;
                o.read = (uint32_t)r; o.frame = frame; o.stop_position = stop_position;
                o.orf_len = orf_len; o.gene_len = gene_len; o.lo = lo; o.hi = hi;
                o.first_j = 0; o.start_begin = 0; o.n_starts = 0; o.accepted = 0; o.orf_is_truncated = 0;
                o.reserved = 0; o.best_score = -DBL_MAX;
                out[cnt] = o;
                if (counting) {
                    const int m = hi - lo;
                    const bool tr = trunc && (frame > 0 ? lo < 3 : n - (hi - 1) < 3);
                    const int jmax = m >= 1 ? (m - 1) / 3 * 3 : -1;
                    a.orf_cnt[orf0 + cnt] = (uint32_t)((m > 0 ? n_real : 0) + (tr && jmax >= j_lo ? 1 : 0));
                }
                cnt++;
            }
        };
        auto fwd_stop = [&](int i, MgClass &S, int cls) __attribute__((always_inline)) {
            int gene_len, orf_len;
            if (S.prev_fwd_stop == 0) {
                const int pos = i - 1;
                orf_len = pos - 1;
                orf_len -= orf_len % 3;
                gene_len = S.first_fwd_start == INT_MAX ? 0 : pos - S.first_fwd_start;
                if (trunc && gene_len < mgl) gene_len = orf_len;
            } else {
                gene_len = (i - S.first_fwd_start) - 1;
                orf_len = i - S.prev_fwd_stop - 4;
            }
            emit(i - 1, 1 + (cls + 1) % 3, gene_len, orf_len, S.fwd_last + 1, i - 2, S.fwd_older);
            S.first_fwd_start = INT_MAX;
            S.prev_fwd_stop = i - 1;
            S.fwd_older = 0;
            S.fwd_recent = 0;
        };
        auto rev_stop = [&](int i, MgClass &S, int cls) __attribute__((always_inline)) {
            int gene_len, orf_stop = 0;
            if (S.prev_rev_stop == 0) {
                if (!trunc) gene_len = 0;
                else {
                    orf_stop = (i - 1) % 3;
                    if (orf_stop > 0) orf_stop -= 3;
                    gene_len = S.last_rev_start - orf_stop;
                }
            } else {
                orf_stop = S.prev_rev_stop;
                gene_len = S.last_rev_start - orf_stop;
            }
            emit(orf_stop, -1 - (cls + 1) % 3, gene_len, i - orf_stop - 4, orf_stop + 3, i - 1, S.rev_cnt);
            S.last_rev_start = 0;
            S.prev_rev_stop = i - 1;
            S.rev_cnt = 0;
            S.rev_from = (i - 1) + 4 + j_lo;
        };
        // the forward start count of class S catches up to position i (the codons of the class since its last event hold no start)
        auto advance = [&](MgClass &S, int &last_i, int i, unsigned long long flag) __attribute__((always_inline)) {
            const int shift = (i - last_i) / 3;
            if (shift >= k0) { S.fwd_older += __popcll(S.fwd_recent); S.fwd_recent = 0; }
            else { S.fwd_older += __popcll(S.fwd_recent >> (k0 - shift)); S.fwd_recent = (S.fwd_recent << shift) & k0_mask; }
            S.fwd_recent |= flag;
            last_i = i;
        };

        MgClass cs[3];
        int last_i[3] = {0, 1, -1};                     // one codon in front of the class's first (positions 3, 4, 2)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            cs[c].first_fwd_start = INT_MAX;
            cs[c].last_rev_start = cs[c].prev_fwd_stop = cs[c].prev_rev_stop = 0;
            cs[c].fwd_recent = 0;
            cs[c].fwd_older = cs[c].rev_cnt = 0;
            cs[c].rev_from = (c == 0 ? -1 : c == 1 ? 0 : -2) + 4 + j_lo;
        }
        cs[0].fwd_last = 0; cs[1].fwd_last = 1; cs[2].fwd_last = -1;
        BaseStream<1> bs;
        bs.init(a.packed, off);
        uint32_t idx6 = 0;
        // the reference's steps at one codon that is in a start or stop set (position I_ of the read, B_: which sets).  A macro: as a lambda
        // that captures the three classes' states the compiler kept them in scratch memory (160 bytes per lane: the pass took 7 ms)
#define MG_EV_EVENT(I_, B_)                                                                                                        \
        {                                                                                                                        \
            const int i = (I_);                                                                                                  \
            const uint32_t bits = (B_);                                                                                          \
                const int c = (int)((uint32_t)i % 3u);                                                                           \
                MgClass S = c == 0 ? cs[0] : c == 1 ? cs[1] : cs[2];                                                             \
                int li = c == 0 ? last_i[0] : c == 1 ? last_i[1] : last_i[2];                                                    \
                if (counting) {                                                                                                  \
                    advance(S, li, i, (bits & 5u) == 1u ? 1ull : 0ull);                                                          \
                    if ((bits & 10u) == 2u && i >= S.rev_from) S.rev_cnt++;                                                      \
                }                                                                                                                \
                if ((bits & 1u) && S.first_fwd_start == INT_MAX) S.first_fwd_start = i - 1;                                      \
                if (bits & 2u) S.last_rev_start = i - 1;                                                                         \
                if (bits & 4u) { fwd_stop(i, S, c); S.fwd_last = i; }                                                            \
                if (bits & 8u) rev_stop(i, S, c);                                                                                \
                if (c == 0) { cs[0] = S; last_i[0] = li; } else if (c == 1) { cs[1] = S; last_i[1] = li; } else { cs[2] = S; last_i[2] = li; } \
        }
        if (MASKS) {
            // ---- 32 word-aligned bases at a time: the four tests of their codons as masks (bit b: the codon whose LAST base is base 32 w + b)
            const uint64_t wb = off & ~31ull;
            const int lead = (int)(off - wb);           // bases of the first word in front of the read
            for (uint64_t g0 = wb; g0 < off + (uint64_t)n; g0 += 32) {
                const uint32_t *pw = a.packed + (g0 >> 4);
                const uint32_t q0 = pw[-1], q1 = pw[0], q2 = pw[1];        // (guard words in front of the first read)
                const uint32_t s0 = q0 >> 28 | q1 << 4, s1 = q1 >> 28 | q2 << 4, s2 = q2 >> 28;
                uint32_t X[4];
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    uint32_t t[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int j = 4 * g + u;
                        uint32_t key;
                        if (j < 7) key = (s0 >> (4 * j)) & 255u;
                        else if (j == 7) key = (s0 >> 28 | s1 << 4) & 255u;
                        else if (j < 15) key = (s1 >> (4 * (j - 8))) & 255u;
                        else key = (s1 >> 28 | s2 << 4) & 255u;
                        t[u] = s_tab[key];
                    }
                    X[g] = t[0] | t[1] << 2 | t[2] << 4 | t[3] << 6;       // byte m: eight codons' membership in set m
                }
                uint32_t msk[4];
#pragma unroll
                for (int m = 0; m < 4; m++)
                    msk[m] = ((X[0] >> (8 * m)) & 255u) | ((X[1] >> (8 * m)) & 255u) << 8 | ((X[2] >> (8 * m)) & 255u) << 16 | ((X[3] >> (8 * m)) & 255u) << 24;
                // positions of the read with a whole codon behind them: i = 32 w + b - lead in [2, n)
                const int i0 = (int)(g0 - wb) - lead;   // position of bit 0
                uint32_t valid = ~0u;
                if (i0 < 2) valid = 2 - i0 >= 32 ? 0u : ~0u << (2 - i0);
                if (n - i0 < 32) valid &= (1u << (n - i0)) - 1u;
                uint32_t any = (msk[0] | msk[1] | msk[2] | msk[3]) & valid;
                while (any) {
                    const uint32_t b = (uint32_t)__ffs((int)any) - 1u;
                    any &= any - 1u;
                    MG_EV_EVENT(i0 + (int)b, ((msk[0] >> b) & 1u) | ((msk[1] >> b) & 1u) << 1 | ((msk[2] >> b) & 1u) << 2 | ((msk[3] >> b) & 1u) << 3)
                }
            }
        } else
        for (int chunk0 = 0; chunk0 < n; chunk0 += MG_EV_CH) {
            // ---- phase A: look
            const int kn = n - chunk0 < MG_EV_CH ? n - chunk0 : MG_EV_CH;
            uint32_t nq = 0;
            for (int k = 0; k < kn; k++) {
                idx6 = ((idx6 << 2) | (uint32_t)bs.next()) & 63u;
                const uint32_t bits = s_evt[idx6];
                s_q[nq][lane] = (uint16_t)((uint32_t)k | bits << 8);       // (overwritten by the next codon when this one has none)
                nq += bits != 0 && chunk0 + k >= 2 ? 1u : 0u;                 // a Codon_t with an empty position matches nothing (gene.cc:56,85)
            }
            // ---- phase B: the reference's steps at the queued codons
            for (uint32_t e = 0; e < nq; e++) {
                const uint32_t ev = s_q[e][lane];
                MG_EV_EVENT(chunk0 + (int)(ev & 255u), ev >> 8)
            }
        }
#undef MG_EV_EVENT
        // Finish_Orfs (glimmer_base.cc:783-817) + Handle_Last_Reverse_Stop, linear (:1053-1066)
#pragma unroll
        for (int fr = 0; fr < 3; fr++) {
            const MgClass &S = cs[fr];
            const int orf_stop = S.prev_rev_stop == 0 ? (fr == 0 ? -1 : fr == 1 ? 0 : -2) : S.prev_rev_stop;
            int orf_len = n - orf_stop - 2;
            orf_len -= orf_len % 3;
            int gene_len = S.last_rev_start == 0 ? 0 : S.last_rev_start - orf_stop;
            if (trunc && gene_len < mgl) gene_len = orf_len;
            const int e = orf_stop + 2;
            int hi;
            if (e >= n) hi = e + 1;
            else { const int rc = (n - 1 - e) % 3; hi = (rc == 0 ? n - 1 : rc == 1 ? n - 2 : n) + 1; }
            emit(orf_stop, -1 - (fr + 1) % 3, gene_len, orf_len, orf_stop + 3, hi, S.rev_cnt);
        }
        if (trunc)                                      // glimmer_base.cc:765-776: 3 bp past the end count as stops
            for (int i = n; i < n + 3; i++) {
                const int c = i % 3;
                if (c == 0) { if (counting) advance(cs[0], last_i[0], i, 0ull); fwd_stop(i, cs[0], 0); }
                else if (c == 1) { if (counting) advance(cs[1], last_i[1], i, 0ull); fwd_stop(i, cs[1], 1); }
                else { if (counting) advance(cs[2], last_i[2], i, 0ull); fwd_stop(i, cs[2], 2); }
            }
    }
}

// ---------------------------------------------------------------------------------------------------
// GENE32: the null model's part of Frame_Scores[(fwd ? 0 : 3) + f][si] of a read of n bases (glimmer-mg.cc:1485-1509):
// the (3,2,3) model's Frame_Score on the reversed read (fwd) or on the complemented read, at forward coordinate si.
//   fwd: buffer position j = n-1-si, window B[j-2..j] = S[si+2], S[si+1], S[si]
//   rev: buffer position j = si,     window B[j-2..j] = comp S[si-2], comp S[si-1], comp S[si]
// positions j < 2 take the partial-window tables (icm.cc:807-842), as the dense_part tables of gmg_model_upload.
// c0, c1, c2 = codes of S[si], S[si+1], S[si+2] (fwd) / S[si], S[si-1], S[si-2] (rev); what lies outside the read is not used.
// ---------------------------------------------------------------------------------------------------
#define MG_NULL_FLOATS 252       // one null model: [3][64] full windows, then [3][20] partial windows
// index of the f = 0 entry in that table and the distance to the entries of f = 1, 2 (no branches: the lanes of a wave sit at
// different distances from their reads' ends)
template <bool FWD>
__device__ __forceinline__ uint32_t mg_null_index(int si, int n, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t &stride)
{
    const int j = FWD ? n - 1 - si : si;
    const uint32_t x = FWD ? 0u : 3u;                   // complement
    const uint32_t b0 = c0 ^ x, b1 = c1 ^ x, b2 = c2 ^ x;          // B[j], B[j-1], B[j-2]
    const uint32_t full = b2 | b1 << 2 | b0 << 4, p1 = 192u + 4u + (b1 | b0 << 2), p0 = 192u + b0;
    stride = j >= 2 ? 64u : 20u;
    return j >= 2 ? full : j == 1 ? p1 : p0;
}

template <bool FWD>
__device__ __forceinline__ float mg_null_value(const float *tab, int f, int si, int n, uint32_t c0, uint32_t c1, uint32_t c2)
{
    uint32_t stride;
    const uint32_t i0 = mg_null_index<FWD>(si, n, c0, c1, c2, stride);
    return tab[i0 + (uint32_t)f * stride];
}

// ---------------------------------------------------------------------------------------------------
// Cumulative_Frame_Score (glimmer-mg.cc:561-604) for every ORF a read can have, in ONE walk per strand.
//
// A forward ORF with bounds (lo, hi) sums Frame_Scores[f][si] for si = hi-1, hi-2, ... with f = 1,2,0,...;
// a reverse ORF sums Frame_Scores[3+f][si] for si = lo-1, lo, ... .  Both start right behind an in-frame
// stop codon (a real one, or the virtual ones Find_Orfs / Save_Prev_Stops put around the read:
// glimmer_base.cc:765-776, 1035-1042; glimmer-mg.cc:685,708-710), and ORFs of one reading-frame class do
// not overlap.  So three running sums per strand, one per class, each set to zero when the walk steps
// over a stop codon of its class, hold score[j-1] of whichever ORF is open in that class -- with exactly
// the reference's sequence of double additions, because every sum restarts from 0 at the ORF's own first
// base.  Seen from the walk, the class that is in frame now (j % 3 == 0) takes row 1, the class in frame
// at the next step (j % 3 == 2) row 0, the third (j % 3 == 1) row 2; then the roles rotate.  At every
// base the in-frame class stores its sum BEFORE adding: that is score[j-1] of glimmer-mg.cc:1826.
// One lane per (read, strand); 64-byte loads per row and lane (8 positions), 64-byte stores.
// ---------------------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(8))) MgD4 { double v[4]; };
struct __attribute__((packed, aligned(8))) MgD3 { double v[3]; };
struct __attribute__((packed)) MgU4 { uint32_t v; };    // four quality values at any byte address

template <bool FWD, bool GENE32 = false>
__device__ __forceinline__ void mg_cum_one(const MgArgs &a, uint64_t r)
{
    const int64_t off = (int64_t)a.read_off[r];
    const int n = (int)((int64_t)a.read_off[r + 1] - off);
    if (n <= 0) return;
    const double *row0 = GENE32 ? nullptr : a.fs + (FWD ? 0 : 3) * a.fs_stride + off;
    const double *row1 = row0 + a.fs_stride, *row2 = row1 + a.fs_stride;
    double *ctab = a.cum + (FWD ? 0 : a.total) + off;
    const uint64_t stopmask = FWD ? a.fwd_stop : a.rev_stop;
    int64_t g = off + (FWD ? n - 1 : 0);
    uint32_t w = a.packed[g >> 4];
    uint32_t idx6 = 0;
    double A = 0.0, B = 0.0, C = 0.0;

    // one base: r0/r1/r2 = rows 0,1,2 (+3) at this base; returns what the in-frame class stores
    auto step = [&](bool warm, double r0, double r1, double r2) __attribute__((always_inline)) {
        const bool reset = warm || ((stopmask >> idx6) & 1ull);          // the three bases just passed are a stop of this class
        A = reset ? 0.0 : A;
        const double keep = A;
        A += r1; B += r0; C += r2;
        const uint32_t code = (w >> (2u * (unsigned)(g & 15))) & 3u;
        const int64_t g2 = g + (FWD ? -1 : 1);
        if ((g ^ g2) >> 4) w = a.packed[g2 >> 4];
        g = g2;
        // the last three bases passed, indexed as Find_Orfs indexes a codon (first base of the forward reading highest)
        idx6 = FWD ? (idx6 >> 2) | (code << 4) : ((idx6 << 2) | code) & 63u;
        const double tA = A; A = B; B = C; C = tA;
        return keep;
    };

    if (GENE32) {                                       // fp32 gene rows - the read's null model, position by position
        const float *g0 = a.gene32 + (uint64_t)(FWD ? 0 : 3) * a.fs_stride + off, *g1 = g0 + a.fs_stride, *g2 = g1 + a.fs_stride;
        const uint32_t ni = a.read_null ? a.read_null[r] : 0u;
        const float *nt = a.null_tab + (size_t)ni * MG_NULL_FLOATS;
        uint32_t c1 = 0, c2 = 0;
        for (int t = 0; t < n; t++) {
            const int si = FWD ? n - 1 - t : t;
            const uint32_t c0 = (uint32_t)dev_code(a.packed, (uint64_t)(off + si));
            const double r0 = (double)g0[si] - (double)mg_null_value<FWD>(nt, 0, si, n, c0, c1, c2);
            const double r1 = (double)g1[si] - (double)mg_null_value<FWD>(nt, 1, si, n, c0, c1, c2);
            const double r2 = (double)g2[si] - (double)mg_null_value<FWD>(nt, 2, si, n, c0, c1, c2);
            ctab[si] = step(t < 3, r0, r1, r2);
            c2 = c1;
            c1 = c0;
        }
        return;
    }
    int t = 0;
    for (; t + 24 <= n; t += 24) {
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const int64_t first = FWD ? (int64_t)n - 1 - (t + 8 * u) : (int64_t)(t + 8 * u);   // base of the group's first step
            const int64_t base = FWD ? first - 7 : first;
            MgD4 x0[2], x1[2], x2[2], y[2];
            x0[0] = *(const MgD4 *)(row0 + base); x0[1] = *(const MgD4 *)(row0 + base + 4);
            x1[0] = *(const MgD4 *)(row1 + base); x1[1] = *(const MgD4 *)(row1 + base + 4);
            x2[0] = *(const MgD4 *)(row2 + base); x2[1] = *(const MgD4 *)(row2 + base + 4);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int e = FWD ? 7 - k : k;
                y[e >> 2].v[e & 3] = step(t + 8 * u + k < 3, x0[e >> 2].v[e & 3], x1[e >> 2].v[e & 3], x2[e >> 2].v[e & 3]);
            }
            *(MgD4 *)(ctab + base) = y[0];
            *(MgD4 *)(ctab + base + 4) = y[1];
        }
    }
    for (; t < n; t++) {
        const int64_t si = FWD ? (int64_t)n - 1 - t : (int64_t)t;
        ctab[si] = step(t < 3, row0[si], row1[si], row2[si]);
    }
}

// ---------------------------------------------------------------------------------------------------
// k_mg_cum_tiled: the same sums with coalesced traffic.  A block stages the three rows of one strand of a
// few consecutive reads (<= MG_CAP bases) in LDS with contiguous loads, lists the places where a running
// sum starts -- behind a stop codon of its class, real or virtual -- and gives every such region (up to the
// next stop of the class) to one lane: one sum per lane, f = 1,2,0,..., on the staged rows.  The value the
// reference calls score[j-1] overwrites the row-1 entry it was about to consume (each entry is consumed
// exactly once), so row 1 of the tile becomes the output and leaves with contiguous stores.
// Which reads a tile takes is a pure function of the offsets (mg_tile_reads), so that k_mg_cum can pick up
// the reads that do not fit (longer than the tile, or more than MG_TILE_READS in one window).
// ---------------------------------------------------------------------------------------------------
#define MG_TILE_READS 64         // reads per tile

// first read r in [0, n_reads] with read_off[r] >= key; the table of the read at every 1024th base (gmg_reads)
// narrows the search to the reads of one 1024-base stretch
__device__ __forceinline__ uint64_t mg_lower_bound(const MgArgs &a, uint64_t key)
{
    if (key >= a.total) return key > a.total ? a.n_reads + 1 : a.n_reads;  // read_off[n_reads] == total
    const uint64_t t = key / GMG_TILE;
    uint64_t lo = a.tile_read[t];                                        // read_off[lo] <= t * 1024 <= key
    uint64_t hi = (t + 1) * GMG_TILE < a.total ? (uint64_t)a.tile_read[t + 1] + 1 : a.n_reads;   // read_off[hi] > key, or the end
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (a.read_off[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}

// reads [first, first + nfit) of tile k
__device__ __forceinline__ void mg_tile_reads(const MgArgs &a, uint64_t k, uint64_t &first, uint32_t &nfit, const uint32_t cap)
{
    uint64_t end;
    if (a.uniform_len > 0) {
        first = k * (uint64_t)a.reads_per_tile;
        end = first + (uint64_t)a.reads_per_tile < a.n_reads ? first + (uint64_t)a.reads_per_tile : a.n_reads;
        nfit = first < end ? (uint32_t)(end - first) : 0;
        return;
    }
    first = mg_lower_bound(a, k * a.tile_window);
    end = mg_lower_bound(a, (k + 1) * a.tile_window);
    if (end > a.n_reads) end = a.n_reads;
    nfit = 0;
    if (first >= a.n_reads) return;
    const uint64_t w0 = a.read_off[first];
    while (first + nfit < end && nfit < (uint32_t)a.tile_reads_max && a.read_off[first + nfit + 1] - w0 <= cap) nfit++;
}

// ragged batches: the tile of every window, computed once (the tiled kernel then needs ONE load per tile instead of a
// chain of searches); the empty ones are dropped with a select
__global__ __launch_bounds__(256) void k_mg_tile_table(MgArgs a, uint64_t n_windows, uint32_t cap, MgTile *tab)
{
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_windows; k += (uint64_t)gridDim.x * blockDim.x) {
        MgTile t;
        uint64_t first;
        mg_tile_reads(a, k, first, t.nfit, cap);
        t.first = (uint32_t)first;
        t.w0 = t.nfit ? a.read_off[first] : 0;
        t.span = t.nfit ? (uint32_t)(a.read_off[first + t.nfit] - t.w0) : 0;
        t.pad = 0;
        tab[k] = t;
    }
}

// ... the non-empty ones in their order: flags (k_mg_tile_flags), their exclusive sums (gmg_scan.h), then every such tile to its place
__global__ __launch_bounds__(256) void k_mg_tile_flags(const MgTile *tab, uint64_t n_windows, uint32_t *flag)
{
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k <= n_windows; k += (uint64_t)gridDim.x * blockDim.x)
        flag[k] = k < n_windows && tab[k].nfit != 0 ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_mg_tile_compact(const MgTile *tab, uint64_t n_windows, const uint32_t *pos, MgTile *tiles, uint32_t *n_tiles)
{
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_windows; k += (uint64_t)gridDim.x * blockDim.x)
        if (tab[k].nfit != 0) tiles[pos[k]] = tab[k];
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_tiles = pos[n_windows];
}

template <int MG_CAP, int BLOCK, bool GENE32 = false>
__global__ __launch_bounds__(BLOCK) void k_mg_cum_tiled(MgArgs a)
{
    constexpr int RS = MG_CAP + 8;                                      // row stride: 4 guard entries on both sides, so that
                                                                        // the walk needs no index clamps
    extern __shared__ __attribute__((aligned(16))) double s_fs[];       // [3][RS]: rows 0,1,2 of one strand, base b at 4 + b
    __shared__ uint16_t s_list[MG_CAP];                                 // tile bases where a running sum starts
    __shared__ uint8_t s_flag[RS + BLOCK];                              // the same as one byte per base (4 + b), ones around the tile
    __shared__ uint32_t s_packed[MG_CAP / 16 + 4];                      // one word in front of the tile's first: base b at bit pair 16 + shift + b
    __shared__ uint32_t s_roff[MG_TILE_READS + 1];                      // read starts relative to the tile
    __shared__ uint32_t s_nlist;
    constexpr int PER = (MG_CAP + BLOCK - 1) / BLOCK;                   // doubles per lane and row
    constexpr int PW = (MG_CAP / 16 + 4 + BLOCK - 1) / BLOCK;           // packed words per lane
    constexpr int PR = (MG_TILE_READS + 1 + BLOCK - 1) / BLOCK;         // read offsets per lane

    struct Tile { uint64_t first, w0; uint32_t nfit, span; };
    const uint64_t n_tiles = a.n_tiles_dev ? (uint64_t)*a.n_tiles_dev : a.n_tiles;
    typename std::conditional<GENE32, float, double>::type tmp[3][PER];
    uint32_t tpk[PW], tro[PR];                          // (raw: arithmetic on a loaded value where it is issued would wait for every load before it)
    // GENE32: the null models of the tile's first MG_NULL_CACHE reads sit in LDS (1 KB each; one model for the whole batch:
    // slot 0, loaded once); reads beyond them fetch their values through L1
    constexpr int NC = GENE32 ? (MG_CAP <= 512 ? 4 : 8) : 1;
    constexpr int PN = GENE32 ? (NC * MG_NULL_FLOATS + BLOCK - 1) / BLOCK : 1;
    __shared__ float s_null[GENE32 ? NC * MG_NULL_FLOATS : 1];
    float tnl[PN];
    if (GENE32 && !a.read_null) {
        for (uint32_t i = threadIdx.x; i < MG_NULL_FLOATS; i += BLOCK) s_null[i] = a.null_tab[i];
    }
    auto meta = [&](uint64_t k, Tile &t) __attribute__((always_inline)) {
        t.nfit = 0; t.span = 0; t.first = 0; t.w0 = 0;
        if (k >= 2 * n_tiles) return;
        if (a.tiles) {                                  // ragged batch: precomputed, non-empty
            const MgTile e = a.tiles[k >> 1];
            t.first = e.first; t.nfit = e.nfit; t.w0 = e.w0; t.span = e.span;
            return;
        }
        mg_tile_reads(a, k >> 1, t.first, t.nfit, MG_CAP);
        if (t.nfit == 0) return;
        t.w0 = t.first * (uint64_t)a.uniform_len;       // (uniform batches only come here: no load, nothing to wait for)
        t.span = t.nfit * (uint32_t)a.uniform_len;
    };
    // every global load of a tile is issued here, one tile ahead: they are in flight while the block works on the
    // tile before
    auto issue = [&](uint64_t k, const Tile &t) __attribute__((always_inline)) {
        if (t.nfit == 0) return;
        const bool fwd = (k & 1) == 0;
#pragma unroll
        for (int row = 0; row < 3; row++) {
            if (GENE32) {
                const float *src = a.gene32 + (uint64_t)((fwd ? 0 : 3) + row) * a.fs_stride + t.w0;
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const uint32_t i = threadIdx.x + (uint32_t)BLOCK * u;
                    tmp[row][u] = i < t.span ? src[i] : 0.0f;
                }
            } else {
                const double *src = a.fs + (uint64_t)((fwd ? 0 : 3) + row) * a.fs_stride + t.w0;
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const uint32_t i = threadIdx.x + (uint32_t)BLOCK * u;
                    tmp[row][u] = i < t.span ? src[i] : 0.0;
                }
            }
        }
        const uint32_t n_words = ((uint32_t)(t.w0 & 15) + t.span + 15) / 16 + 1;       // + the word in front (a guard word of gmg_reads at the batch's start)
#pragma unroll
        for (int u = 0; u < PW; u++) {
            const uint32_t i = threadIdx.x + (uint32_t)BLOCK * u;
            tpk[u] = i < n_words ? a.packed[(int64_t)(t.w0 >> 4) - 1 + i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < PR; u++) {
            const uint32_t i = threadIdx.x + (uint32_t)BLOCK * u;
            tro[u] = i <= t.nfit ? ((const uint32_t *)(a.read_off + t.first + i))[0] : 0u;        // (the low word is all that is needed)
        }
        if (GENE32 && a.read_null) {
#pragma unroll
            for (int u = 0; u < PN; u++) {
                const uint32_t i = threadIdx.x + (uint32_t)BLOCK * u, rl = i / MG_NULL_FLOATS;
                tnl[u] = rl < t.nfit && rl < (uint32_t)NC ? a.null_tab[(size_t)a.read_null[t.first + rl] * MG_NULL_FLOATS + (i - rl * MG_NULL_FLOATS)] : 0.0f;
            }
        }
    };

    uint64_t k = blockIdx.x;
    Tile cur;
    meta(k, cur);
    issue(k, cur);
    for (; k < 2 * n_tiles; k += gridDim.x) {           // (tile, strand)
        const bool fwd = (k & 1) == 0;
        const uint32_t nfit = cur.nfit, span = cur.span;
        const uint64_t w0 = cur.w0;
        __syncthreads();                                // the previous tile has left the LDS
        if (nfit) {
            if (!GENE32) {
#pragma unroll
                for (int row = 0; row < 3; row++)
#pragma unroll
                    for (int u = 0; u < PER; u++) {
                        const uint32_t i = threadIdx.x + (uint32_t)BLOCK * u;
                        if (i < span) s_fs[row * RS + 4 + i] = tmp[row][u];
                    }
            }
#pragma unroll
            for (int u = 0; u < PW; u++) {
                const uint32_t i = threadIdx.x + (uint32_t)BLOCK * u;
                if (i < MG_CAP / 16 + 4) s_packed[i] = tpk[u];
            }
#pragma unroll
            for (int u = 0; u < PR; u++) {
                const uint32_t i = threadIdx.x + (uint32_t)BLOCK * u;
                if (i <= nfit) s_roff[i] = tro[u] - (uint32_t)w0;
            }
            if (GENE32 && a.read_null) {
#pragma unroll
                for (int u = 0; u < PN; u++) {
                    const uint32_t i = threadIdx.x + (uint32_t)BLOCK * u;
                    if (i < (uint32_t)(NC * MG_NULL_FLOATS)) s_null[i] = tnl[u];
                }
            }
        }
        if (threadIdx.x == 0) s_nlist = 0;
        if (GENE32) {
            // Frame_Scores of the tile = (double) gene value - (double) the read's null-model value, built here from the fp32
            // rows in registers and the packed bases just staged (the reads' null models come through L1 / L2: 1 KB each)
            __syncthreads();
            if (nfit) {
                const uint32_t shift_g = 16u + (uint32_t)(w0 & 15);
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const uint32_t b = threadIdx.x + (uint32_t)BLOCK * u;
                    if (b < span) {
                        uint32_t rl;                    // the read of base b, relative to the tile's first
                        if (a.uniform_len > 0) rl = a.uniform_len == 1 ? b : __umulhi(b, a.uniform_magic);      // b / uniform_len (b < 2^16)
                        else {
                            uint32_t lo = 0, hi = nfit;
                            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_roff[mid] <= b) lo = mid; else hi = mid; }
                            rl = lo;
                        }
                        const int rs = a.uniform_len > 0 ? (int)(rl * (uint32_t)a.uniform_len) : (int)s_roff[rl];
                        const int n = a.uniform_len > 0 ? a.uniform_len : (int)s_roff[rl + 1] - rs;
                        const int si = (int)b - rs;
                        // codes of bases b-2 .. b+2 (five 2-bit fields, base b-2 lowest)
                        const uint32_t x = shift_g + b - 2u;
                        const uint64_t two = (uint64_t)s_packed[x >> 4] | (uint64_t)s_packed[(x >> 4) + 1] << 32;
                        const uint32_t five = (uint32_t)(two >> (2u * (x & 15u))) & 0x3ffu;
                        const uint32_t c0 = (five >> 4) & 3u;
                        const uint32_t c1 = fwd ? (five >> 6) & 3u : (five >> 2) & 3u;      // S[si+1] / S[si-1]
                        const uint32_t c2 = fwd ? (five >> 8) & 3u : five & 3u;             // S[si+2] / S[si-2]
                        uint32_t nstride;
                        const uint32_t i0 = fwd ? mg_null_index<true>(si, n, c0, c1, c2, nstride) : mg_null_index<false>(si, n, c0, c1, c2, nstride);
                        const uint32_t slot = a.read_null ? rl : 0u;
                        if (slot < (uint32_t)NC) {
                            const float *nt = s_null + slot * MG_NULL_FLOATS + i0;
#pragma unroll
                            for (int row = 0; row < 3; row++) s_fs[row * RS + 4 + b] = (double)tmp[row][u] - (double)nt[row * nstride];
                        } else {                        // (a tile of many short reads)
                            const float *nt = a.null_tab + (size_t)a.read_null[cur.first + rl] * MG_NULL_FLOATS + i0;
#pragma unroll
                            for (int row = 0; row < 3; row++) s_fs[row * RS + 4 + b] = (double)tmp[row][u] - (double)nt[row * nstride];
                        }
                    }
                }
            }
        }
        Tile nxt;
        meta(k + gridDim.x, nxt);
        issue(k + gridDim.x, nxt);
        __syncthreads();
        if (nfit) {
        const uint32_t shift = 16u + (uint32_t)(w0 & 15);     // tile base b is bit pair shift + b of s_packed (one word in front)
        auto codon = [&](uint32_t b) __attribute__((always_inline)) {      // tile bases b, b+1, b+2 as Find_Orfs indexes a codon
            const uint32_t x = shift + b;
            const uint64_t two = (uint64_t)s_packed[x >> 4] | (uint64_t)s_packed[(x >> 4) + 1] << 32;
            const uint32_t c = (uint32_t)(two >> (2u * (x & 15u))) & 63u;   // base b lowest
            return (c & 3u) << 4 | (c & 12u) | c >> 4;
        };
        // 1. where do running sums start?  forward strand: the walk goes down and starts below a forward stop codon
        //    (bases si+1..si+3) or below the virtual ones past the end of the read (glimmer_base.cc:765-776); reverse
        //    strand: it goes up and starts behind a reverse stop codon (bases si-3..si-1) or the virtual ones before
        //    the read (:1001-1003, 1035-1042).  One flag bit per tile base (bit 32 + b), all ones around the tile: the
        //    last three bases of a read (forward) / its first three (reverse) are always starts, so a walk can never
        //    step over a read boundary without meeting a flag.
        for (uint32_t b0 = 0; b0 < MG_CAP + 4; b0 += BLOCK) {
            const uint32_t b = b0 + threadIdx.x;
            bool st = true;
            if (b < span) {
                int rs, n;
                if (a.uniform_len > 0) { rs = (int)(a.uniform_len == 1 ? b : __umulhi(b, a.uniform_magic)) * a.uniform_len; n = a.uniform_len; }
                else {
                    uint32_t lo = 0, hi = nfit;         // last r with s_roff[r] <= b
                    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_roff[mid] <= b) lo = mid; else hi = mid; }
                    rs = (int)s_roff[lo];
                    n = (int)s_roff[lo + 1] - rs;
                }
                const int si = (int)b - rs;
                if (fwd) st = si + 3 >= n || ((a.fwd_stop >> codon((uint32_t)(rs + si + 1))) & 1ull);
                else st = si < 3 || ((a.rev_stop >> codon((uint32_t)(rs + si - 3))) & 1ull);
                if (st) s_list[atomicAdd(&s_nlist, 1u)] = (uint16_t)b;
            }
            if (b < MG_CAP + 4) s_flag[4 + b] = st ? 1 : 0;
        }
        if (threadIdx.x < 4) s_flag[threadIdx.x] = 1;
        __syncthreads();
        // 2. one lane per region: a single sum, f = 1,2,0,...; score[j-1] takes the place of the row-1 entry it
        //    precedes.  What a trip adds beyond the end of its read is never used: the next flag ends the walk.
        const uint32_t n_list = s_nlist;
        // one pointer per lane (q = &row0[b]) and compile-time offsets: ~12 instructions per codon.  This loop is what the
        // kernel's VALU time goes to (3.5e9 wave-instructions per 1M x 500 bp before the clamps and index arithmetic went)
        auto walk = [&](auto DIR_) __attribute__((always_inline)) {
            constexpr int DIR = decltype(DIR_)::value;
            for (uint32_t e = threadIdx.x; e < n_list; e += BLOCK) {
                const int b = (int)s_list[e];
                double *q = s_fs + 4 + b;
                const uint8_t *fl = s_flag + 4 + b + 3 * DIR;
                double cum = 0.0;
                for (;;) {
                    const double v1 = q[RS], v2 = q[2 * RS + DIR], v0 = q[2 * DIR];
                    const uint8_t f = *fl;
                    q[RS] = cum;
                    cum += v1; cum += v2; cum += v0;
                    if (f) break;                       // the next region belongs to another lane
                    q += 3 * DIR;
                    fl += 3 * DIR;
                }
            }
        };
        if (fwd) walk(std::integral_constant<int, -1>()); else walk(std::integral_constant<int, 1>());
        __syncthreads();
        double *dst = a.cum + (fwd ? 0 : a.total) + w0;
        for (uint32_t i = threadIdx.x; i < span; i += BLOCK) dst[i] = s_fs[RS + 4 + i];
        }
        cur = nxt;
    }
}

// the reads no tile took (longer than a tile, or more than MG_TILE_READS in one window): a short list, so that the per-lane
// kernel below runs over exactly those instead of testing every read
__global__ __launch_bounds__(256) void k_mg_unfit_list(MgArgs a)
{
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.n_reads; r += (uint64_t)gridDim.x * blockDim.x) {
        if (a.read_off[r + 1] == a.read_off[r]) continue;               // nothing to sum
        const MgTile t = a.windows[a.read_off[r] / a.tile_window];
        if (t.nfit && r >= t.first && r < (uint64_t)t.first + t.nfit) continue;
        a.unfit[atomicAdd(a.unfit_n, 1u)] = (uint32_t)r;
    }
}

// the per-lane walk: for the listed reads (lanes_only_unfit) or for every read
template <bool GENE32>
__global__ __launch_bounds__(256) void k_mg_cum(MgArgs a)
{
    const uint64_t n = a.lanes_only_unfit ? (uint64_t)*a.unfit_n : a.n_reads;
    // lanes [0, n): forward strand; [n, 2n): reverse strand (wave-uniform but for one wave)
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t e = i < n ? i : i - n;
        const uint64_t r = a.lanes_only_unfit ? a.unfit[e] : e;
        if (i < n) mg_cum_one<true, GENE32>(a, r);
        else mg_cum_one<false, GENE32>(a, r);
    }
}

// ---------------------------------------------------------------------------------------------------
// Score_Orf_Starts without errors (glimmer-mg.cc:1693-1861) + the per-ORF part of Score_Orfs_Errors
// (:1632-1685): one lane per ORF, ONE pass from buffer position 0 (the 3' end) upwards.
//   * running sum in reference order: score[j] = score[j-1] + Frame_Scores[f][si], f = 1,2,0,...
//   * the reference scans j downwards; "first" there is the LAST qualifying position here, so entries are
//     produced in ascending j and stored at mirrored slots: entry t of n goes to slot n-1-t, which is the
//     order Score_Orf_Starts pushed them (the count pass supplied n).
//   * a truncated ORF (glimmer-mg.cc:1741,1761) pushes its highest in-frame position unconditionally as a
//     truncated start (slot 0), followed by the real start at the same position if there is one.
// ---------------------------------------------------------------------------------------------------
template <bool WRITE>
__device__ __forceinline__ void mg_starts_one(const MgArgs &a, uint64_t i, const int8_t *s_which)
{
    gmg_mg_orf o = a.orfs[i];
    const bool fwd = o.frame > 0;                       // direction is data, not a code path: the lanes of a wave mix both strands
    const uint64_t off = a.read_off[o.read];
    const int n = (int)(a.read_off[o.read + 1] - off);
    const int lo = o.lo, hi = o.hi, m = hi - lo;
    const bool trunc = a.allow_truncated && (fwd ? lo < 3 : n - (hi - 1) < 3);
    const int mgl = a.min_gene_len;
    const int isl = a.read_isl ? a.read_isl[o.read] : a.ignore_score_len;      // Set_Ignore_Score_Len per read with -c (glimmer-mg.cc:2067)
    int j_lo = mgl - 3 > 1 ? mgl - 3 : 1;              // j >= lowest_j = Min (3, mgl-3), j >= 1 (j-1 is read), j+3 >= mgl
    j_lo = (j_lo + 2) / 3 * 3;
    const int jmax = m >= 1 ? (m - 1) / 3 * 3 : -1;     // highest in-frame position of the buffer
    const bool has_trunc = trunc && jmax >= j_lo;
    const uint32_t n_total = WRITE ? (uint32_t)(a.start_off[i + 1] - a.start_off[i]) : 0;
    gmg_start *out = WRITE ? a.starts + a.start_off[i] : nullptr;
    const int k_base = fwd ? lo - 1 + (m - 1) : hi + 1 - (m - 1);      // pos of j: k_base -/+ j (glimmer-mg.cc:1742,1762,1855-1858)
    const int64_t dir = fwd ? -1 : 1;
    const uint32_t comp = fwd ? 0u : 3u;                // Reverse_Transfer / Complement_Transfer (glimmer-mg.cc:1735,1756)

    uint32_t t = 0;                                     // real starts so far (ascending j)
    double best = -DBL_MAX, s_jmax = 0.0;
    int last_j = -1;
    if (m > 0) {
        int64_t g = (int64_t)off + (fwd ? hi - 1 : lo - 1);             // base of buffer position 0 (signed: steps to -1 at the very front)
        uint32_t w = a.packed[g >> 4];
        auto next_code = [&]() __attribute__((always_inline)) {
            const uint32_t c = ((w >> (2u * (unsigned)(g & 15))) & 3u) ^ comp;
            const int64_t g2 = g + dir;
            if ((g ^ g2) >> 4) w = a.packed[g2 >> 4];   // word -1 / one word past the end: the guard words of gmg_reads
            g = g2;
            return c;
        };
        // score[j-1] of every in-frame position j: k_mg_cum left it at the base of buffer position j
        const double *ctab = a.cum + (fwd ? 0 : a.total);
        const int64_t g0 = g;
        int j = 0;
        for (; j + 2 < m; j += 3) {                     // one whole codon per trip: in-frame position j and its two followers
            const uint32_t c0 = next_code(), c1 = next_code(), c2 = next_code();
            if (j >= j_lo) {
                const int which = s_which[c2 << 4 | c1 << 2 | c0];     // (buff[j+2], buff[j+1], buff[j]) as Codon_t holds them
                if (which >= 0) {
                    if (WRITE) {
                        const double pend = ctab[g0 + dir * j];        // score[j-1] - indep_score[j-1], indep_score == 0
                        const double sc = (j + 2 > isl && 0.0 > pend) ? 0.0 : pend;   // Max (0.0, score), :1644-1646
                        gmg_start st;
                        st.score = sc; st.j = j + 2; st.pos = fwd ? k_base - j : k_base + j;
                        st.which = which; st.truncated = 0; st.first = 0;
                        out[n_total - 1 - t] = st;
                        if (sc > best) best = sc;
                    }
                    t++;
                    last_j = j;
                }
            }
        }
        if (WRITE && has_trunc) s_jmax = ctab[g0 + dir * jmax];
    }
    const uint32_t n_starts = t + (has_trunc ? 1u : 0u);
    if (!WRITE) { a.orf_cnt[i] = n_starts; return; }

    int first_j = 0;
    if (has_trunc) {
        const double sc = (jmax + 2 > isl && 0.0 > s_jmax) ? 0.0 : s_jmax;
        gmg_start st;
        st.score = sc; st.j = jmax + 2; st.pos = fwd ? k_base - jmax : k_base + jmax;
        st.which = -1; st.truncated = 1; st.first = 1;
        out[0] = st;
        if (sc > best) best = sc;
        first_j = jmax + 2;
    } else if (t > 0) {
        out[0].first = 1;                               // the last real start found is the first the reference pushes
        first_j = last_j + 2;
    }
    o.orf_is_truncated = trunc;
    o.start_begin = (uint32_t)a.start_off[i];
    o.n_starts = n_starts;
    o.first_j = first_j;
    o.best_score = -DBL_MAX;
    o.accepted = 0;
    if (n_starts > 0 && first_j + 1 >= mgl) {           // glimmer-mg.cc:1656-1676
        o.best_score = best;
        o.accepted = best > a.start_threshold;
    }
    a.orfs[i] = o;
}

template <bool WRITE>
__global__ __launch_bounds__(256) void k_mg_starts(MgArgs a)
{
    __shared__ int8_t s_which[64];
    if (threadIdx.x < 64) s_which[threadIdx.x] = a.which[threadIdx.x];
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n_orfs; i += (uint64_t)gridDim.x * blockDim.x) {
        mg_starts_one<WRITE>(a, i, s_which);
    }
}

__device__ __forceinline__ uint64_t mg_ord(double x)    // order-preserving map double -> uint64
{
    const uint64_t u = (uint64_t)__double_as_longlong(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double mg_unord(uint64_t u)
{
    return __longlong_as_double((long long)((u >> 63) ? (u & 0x7fffffffffffffffull) : ~u));
}

// ---------------------------------------------------------------------------------------------------
// k_mg_tile_starts: steps 3 and 4 (the running sums and the start lists of the default mode) in ONE kernel per
// tile and strand, the sums as a parallel scan.  The running sums never reach HBM.
//
// Why the order of the additions may change.  Every Frame_Scores entry is (double) g - (double) n with g, n
// floats of the two models; all non-zero floats of both models are multiples of 2^(e_min - 150), and every sum
// of up to R of the differences is below 2^(ceil log2 R + e_max - 125) in magnitude (e_min / e_max: smallest /
// largest exponent field, gmg_model).  While ceil log2 R + e_max - e_min <= 28 all of these are exact doubles: the
// reference's sequential score[j] = score[j-1] + Frame_Scores[f][si] (glimmer-mg.cc:561-604) never rounds, and any
// other order of the same additions gives the same bits.  mg_run checks that with R = the longest read + 2 and
// falls back to the sequential walks (k_mg_cum_tiled / k_mg_cum + k_mg_starts) otherwise; a real model's values span
// about ten binades (NC_000915.icm: 0.16 .. 97).
//
// Walk coordinate u of a tile: u = b on the reverse strand, span-1-b on the forward strand (b = base in the tile),
// so every walk of k_mg_cum_tiled goes up in u, in steps of 3.  T[u] = the three entries the walk consumes at u.
// The three classes u % 3 are three independent segmented sums over the tile; a segment starts where
// k_mg_cum_tiled would start a lane (behind an in-class stop codon, real or virtual; read boundaries are such
// places), and the exclusive sum at u is score[j-1] of the ORF whose region holds u, j = u - (start of the
// segment).
//   stage 0  the staged tile goes to LDS: T (one double per base, formed from the three rows while they are
//            in registers), the packed bases, the read offsets
//   stage 1  one byte per base: segment start?  codon inside the read?  which start codon (Codon_t::Can_Be)?
//            one lane per ORF of the tile: its record's local index, region length and truncation flag go to
//            the LDS slot of the base where its region starts
//   stage 2  the scan.  Lane (class c, part p) owns MT_EL consecutive elements of class c; sum, number of start
//            codons so far and the segment's first u travel together: sequential inside the lane, then a segmented
//            scan over the lanes' totals with DPP moves (no LDS), then over the waves of the work-group.  Every start
//            codon inside an ORF's region at j >= lowest j is a start: its u goes into a queue.
//   stage 3  one lane per queued start: slot in the ORF's slice = n - 1 - (start codons before it in the region,
//            from the scan) -- the reference's push order, k_mg_starts -- score with the Ignore_Score_Len rule,
//            best score per ORF by LDS atomics
//   stage 4  one lane per ORF: the truncated start (glimmer-mg.cc:1741,1761), first_j, best score, verdict
// Reads no tile takes go to k_mg_cum + k_mg_starts_unfit as before.
// ---------------------------------------------------------------------------------------------------
#ifndef GMG_MT_STAMPS
#define GMG_MT_STAMPS 0          // diagnostic build: cycles per stage of k_mg_tile_starts, summed over all waves (tools/mt_stamps.py); not in the product
#endif
#if GMG_MT_STAMPS
__device__ unsigned long long g_mt_stamps[8];
extern "C" int gmg_debug_mt_stamps(unsigned long long *out, int reset)
{
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; return hipMemcpyToSymbol(HIP_SYMBOL(g_mt_stamps), z, sizeof z) == hipSuccess ? 0 : -1; }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mt_stamps), 64) == hipSuccess ? 0 : -1;
}
#define MT_STAMP(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); st_acc[i] += now_ - st_prev; st_prev = now_; } while (0)
#else
#define MT_STAMP(i) do { } while (0)
#endif

#ifndef MT_MIN_WAVES
#define MT_MIN_WAVES 4           // waves per SIMD the register allocation aims at where the LDS allows as many (GENE32 table, one null model)
#endif
#define MT_EL 9                  // chain elements per lane (the kernel's EL: 9, or 8 when 504 bases per wave hold the batch's reads as well)
#define MT_CL 21                 // lanes per class
#define MT_W (3 * MT_CL * MT_EL) // bases per wave: 567
#define MT_ORFS 64               // ORFs per pass of stages 3 and 4
#define MT_NC 6                  // per-read null models of a tile held in LDS = reads per tile in that mode
#define MT_REAL (1u << 24)       // scan word: start codons so far (bits 0-11), u of the segment start (12-23), "holds one" (24),
#define MT_BLK (1u << 25)        // "nothing flows in from the left" (25: a segment start, or the first lane of a class)

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void mt_scan_step(double &s, uint32_t &p)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(s);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, ROW_MASK, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    const uint32_t ps = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p, CTRL, ROW_MASK, 0xf, false);
    const double ss = __longlong_as_double((long long)((unsigned long long)hi << 32 | lo));   // (0.0, 0) where the pattern has no source
    const bool blk = (p & MT_BLK) != 0;
    s = blk ? s : ss + s;
    p = blk ? p : ps + p;
}

struct MtTile { uint64_t w0; uint32_t first, nfit, span, o0, o1, rn0; };    // reads [first, first + nfit), bases [w0, w0 + span), ORFs [o0, o1);
                                                                             // rn0: the null model of the tile's first read (GENE32, one per read)

// G32: the table is the GENE32 form (fp32 rows of the gene model's values alone, a.gene32): half the bytes to read.  T is formed
// from the widened floats, and stage 1, which has the bases of every position in a register anyway, subtracts the three null-model
// values of T from a 3 x 64 table in LDS (one null model for the batch).  A buffer's first two positions take the partial-window
// tables (icm.cc:807-842): the last two bases of a read on the forward strand, its first two on the reverse strand.
// EL: elements per lane: 9 (567 bases per wave), or 8 (504) when that holds the batch's reads -- 500-bp reads: a ninth less of every loop
// (nine elements: three waves per SIMD asked for -- 168 registers, three of them spilled, against 173 and two waves)
// Round 4: with one null model for the batch (PRN false: 6 KB less LDS) and the GENE32 table, FOUR waves per SIMD -- 127 / 128 registers with 4 / 14
// spilled, 20.3 KB (two waves, eight elements) / 40.9 KB (four waves, nine) of LDS = 8 / 4 work-groups per CU: 10.25 -> 9.15 ms per 1 M x 500 bp
// (the fp64-table form would spill 88 - 136 registers at 128: it keeps three / two waves; nine elements at 128 registers spill 14: 10.1 vs 9.9 ms on
// ragged reads, kept at three waves -- ragged batches whose reads fit take the eight-element form instead).  Tried with it: the rows staged with
// 16-byte loads (four consecutive bases per lane: 6 loads per lane and tile instead of 24) -- bit-exact, 9.27 vs 9.20 / 9.75 vs 9.62 ms: the
// unaligned wide loads cost more than the shorter queue saves; dropped.
// PRN: a null model per read (a.read_null; GENE32 only) -- the tables of the tile's reads live in LDS then (6 KB), one table for the batch otherwise (2 KB)
// NC: the per-read null tables held (PRN): MT_NC, or 2 when the batch's reads let a tile take two at most (uniform 500-bp reads in two-wave tiles:
// 4 KB less LDS = eight work-groups per CU there too)
template <int NW, bool G32, int EL, bool PRN, int NC = MT_NC>
#ifndef MT_PRN_WAVES
#define MT_PRN_WAVES MT_MIN_WAVES    // ... with a null model per read (the form spills 8 registers at four waves: 3 was measured, profiles/r05_prn_ab.txt)
#endif
__global__ __launch_bounds__(64 * NW, G32 && EL == 8 ? (PRN ? MT_PRN_WAVES : MT_MIN_WAVES) : EL == 9 ? 3 : 1) void k_mg_tile_starts(MgArgs a)
{
    constexpr int BLOCK = 64 * NW, WV = 3 * MT_CL * EL, CAP = WV * NW;
    constexpr bool DIST = !(G32 && PRN);                                // a base's distance to its read's ends from s_oinfo instead of followed read offsets
    constexpr int NPK = CAP / 16 + 7;                                   // packed words staged (two in front, the windows of stage 2 behind)
    constexpr int PW = (NPK + BLOCK - 1) / BLOCK, PR = (MG_TILE_READS + 1 + BLOCK - 1) / BLOCK;
    __shared__ __attribute__((aligned(16))) double s_val[CAP];          // T in walk order, then the running sums
    // at a region's first u: the ORF's index in the tile + 1 (bits 0-11; a tile's ORFs are fewer than its bases); for every u: how close its base lies
    // to its read's ends -- bits 12-13: 3 - min (si, 3), bits 14-15: 3 - min (n - 1 - si, 3) (0 = three or more away: only the six edge bases
    // of a read are marked) -- what stage 2 asks about a base's read (DIST; with a null model per read it follows the reads itself: it needs
    // their index).  Marks and ORF indices are OR-ed in with 32-bit LDS atomics: any order.
    __shared__ __attribute__((aligned(4))) uint16_t s_oinfo[CAP + 2];
    __shared__ uint32_t s_ch[CAP];                                      // start codons before u in its segment | the segment's first u << 12
    __shared__ uint16_t s_q[CAP];                                       // the starts found: u | (which + 1) << 12
    __shared__ uint32_t s_packed[NPK];
    __shared__ uint32_t s_roff[MG_TILE_READS + 1];
    __shared__ int32_t s_isl[MG_TILE_READS];
    // (a work-group works on ONE strand: the grid is even and k advances by it, so the strand is blockIdx.x's parity -- one copy of the strand's tables)
    __shared__ uint8_t s_wh[64];                                        // which + 1 of the codon field C (see stage 2) | 0x80: C is a stop codon, in the strand's order
    __shared__ unsigned long long s_obest[MT_ORFS];
    __shared__ uint32_t s_oso[MT_ORFS], s_ont[MT_ORFS];
    __shared__ uint16_t s_ofj[MT_ORFS];                                 // (j + 2 of an ORF's first start: below the tile's width)
    __shared__ double s_wsum[NW][3];
    __shared__ uint32_t s_wp[NW][3];
    __shared__ uint32_t s_nq;
    // G32: the null model's full-window values indexed with the read's own bases as they sit in the packed word -- s_nullf[f][v],
    // v = S[x] | S[x+1] << 2 | S[x+2] << 4 (forward strand: window of position x of the reversed read), s_nullr[f][v],
    // v = S[x-2] | S[x-1] << 2 | S[x] << 4 (complemented read) -- and the partial-window tables as they are
    // ([0] forward, [1] reverse, each followed by the partial-window tables).  With a null model per read (a.read_null) the tables of
    // the tile's reads (MT_NC at most in this mode) are fetched with the tile, in the order of the strand it works on.
    __shared__ double s_nulld[G32 && !PRN ? MG_NULL_FLOATS : 1];        // one null model: its values as doubles (no conversion, one table for the full and the partial windows)
    __shared__ float s_nullm[G32 && PRN ? NC : 1][G32 && PRN ? MG_NULL_FLOATS : 1];
    __shared__ uint32_t s_rnull[G32 && PRN ? NC + 2 : 1];                  // (a tile takes NC reads at most in that mode)

    const uint32_t tid = threadIdx.x;
    // entry e of a table in the strand's order <- entry of the (3,2,3) model's table as gmg_null_set / gmg_model_upload lay it out
    auto null_src = [](uint32_t e, bool fwd_order) __attribute__((always_inline)) {
        const uint32_t v = e & 63u;
        return e >= 192u ? e : (e & ~63u) + (fwd_order ? (v & 3u) << 4 | (v & 12u) | v >> 4      // window w[k]: B[j-2], B[j-1], B[j] = S[x+2], S[x+1], S[x]
                                                       : v ^ 63u);                               // ... = comp S[x-2], comp S[x-1], comp S[x]
    };
    const bool fwd_wg = (blockIdx.x & 1u) == 0;
    if (G32 && !PRN)
        for (uint32_t i = tid; i < MG_NULL_FLOATS; i += BLOCK) {
            s_nulld[i] = (double)a.null_tab[null_src(i, fwd_wg)];
        }
    const uint64_t n_tiles = a.n_tiles_dev ? (uint64_t)*a.n_tiles_dev : a.n_tiles;
    if (tid < 64) {
        // forward: the codon (b-2, b-1, b) as Codon_t holds it = the field with its pairs reversed; reverse: the complement of
        // (b+2, b+1, b) = the field itself, complemented
        const uint32_t wh = (uint32_t)(a.which[fwd_wg ? (tid & 3u) << 4 | (tid & 12u) | tid >> 4 : tid ^ 63u] + 1);
        s_wh[tid] = (uint8_t)(wh | (((fwd_wg ? a.fwd_stop_nat : a.rev_stop_nat) >> tid) & 1ull ? 0x80u : 0u));
    }
    const int mgl = a.min_gene_len;
    int j_lo = mgl - 3 > 1 ? mgl - 3 : 1;               // as in mg_starts_one
    j_lo = (j_lo + 2) / 3 * 3;

    // A tile's loads are issued in three steps, one iteration apart, and nothing is computed from a loaded value in the step
    // that loads it (the wait would be for every load issued before it, with the latency in the open):
    //   meta_a  which reads (ragged batches: the tile's entry is loaded; uniform batches: arithmetic)
    //   meta_b  the ORFs of those reads (two offsets)
    //   issue   the three rows, the packed bases, the read offsets, the lane's ORF record
    // The two meta steps read at wave-uniform addresses.  As scalar loads they would share the LDS counter (every wait for an LDS
    // read would wait for them), as uniform vector loads the compiler moves each result to a scalar register at once (a wait for
    // the load where it is issued).  So they are vector loads indexed with a zero the compiler cannot see through: the results
    // stay in vector registers, behind the tile's other loads in the queue, and become scalars where the tile is worked on.
    uint32_t vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    auto meta_a = [&](uint64_t k, MtTile &t) __attribute__((always_inline)) {
        t.nfit = 0; t.span = 0; t.first = 0; t.w0 = 0; t.o0 = 0; t.o1 = 0; t.rn0 = 0;
        if (k >= 2 * n_tiles) return;
        if (a.tiles) {                                  // ragged batch: precomputed, non-empty
            const MgTile *e = a.tiles + (k >> 1) + vzero;
            t.first = e->first; t.nfit = e->nfit; t.w0 = e->w0; t.span = e->span;
        } else {                                        // uniform batch: reads_per_tile whole reads (mg_tile_reads)
            const uint64_t f = (k >> 1) * (uint64_t)a.reads_per_tile;
            const uint64_t end = f + (uint64_t)a.reads_per_tile < a.n_reads ? f + (uint64_t)a.reads_per_tile : a.n_reads;
            t.first = (uint32_t)f;
            t.nfit = f < end ? (uint32_t)(end - f) : 0u;
            t.w0 = f * (uint64_t)a.uniform_len;
            t.span = t.nfit * (uint32_t)a.uniform_len;
        }
    };
    auto meta_b = [&](MtTile &t) __attribute__((always_inline)) {
        const uint32_t *roo = (const uint32_t *)(a.read_orf_off + vzero);      // (low words: a batch has less than 2^31 ORFs)
        t.o0 = roo[2 * (uint64_t)t.first];
        t.o1 = roo[2 * ((uint64_t)t.first + t.nfit)];
        if (G32 && PRN) t.rn0 = (a.read_null + vzero)[t.first];
    };
    // every global load of a tile is issued one tile ahead
    typename std::conditional<G32, float, double>::type tmp[3][EL];
    // (no arithmetic on a loaded value in there: it would wait for every load issued before it)
    // (and no load wider than what is used: a register half nobody reads is handed out again, and the write to it waits for the load)
    uint32_t tpk[PW], tro[PR], trn[PR];
    constexpr int PN1 = (MG_NULL_FLOATS + BLOCK - 1) / BLOCK;
    float tn1[G32 ? PN1 : 1];                           // GENE32, a null model per read: the table of the tile's first read, in the strand's order
    int32_t tis[PR];
    uint32_t po_read = 0, po_s0 = 0, po_s1 = 0;         // the lane's first ORF of the tile: read, frame / lo / hi, slice of the start array (low words)
    int32_t po_frame = 0, po_lo = 0, po_hi = 0;
    auto issue = [&](uint64_t k, const MtTile &t) __attribute__((always_inline)) {
        // (the tile's geometry is wave-uniform and has arrived an iteration ago: as scalars the rows' addresses are a scalar
        // base + a 32-bit lane offset)
        const uint32_t nfit = __builtin_amdgcn_readfirstlane(t.nfit);
        if (nfit == 0) return;
        const uint32_t span = __builtin_amdgcn_readfirstlane(t.span), first = __builtin_amdgcn_readfirstlane(t.first);
        const uint32_t o0 = __builtin_amdgcn_readfirstlane(t.o0), n_orf = __builtin_amdgcn_readfirstlane(t.o1) - o0;
        const uint64_t w0 = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(t.w0 >> 32)) << 32 | __builtin_amdgcn_readfirstlane((uint32_t)t.w0);
        uint32_t tid = threadIdx.x;
        asm volatile("" : "+v"(tid));                   // (index arithmetic is redone where it is used: kept across the stages it costs more registers than instructions)
        const bool fwd = (k & 1) == 0;
        typedef typename std::conditional<G32, float, double>::type row_t;
        const row_t *r0 = (G32 ? (const row_t *)a.gene32 : (const row_t *)a.fs) + (uint64_t)(fwd ? 0 : 3) * a.fs_stride;
        // T[b] = row 1 at b, row 2 at b -/+ 1, row 0 at b -/+ 2 (forward / reverse).  The neighbours of the batch's very first and
        // last bases are clamped into the table: what T holds beyond its read is never used
        const row_t *p1 = r0 + a.fs_stride + w0, *p2 = r0 + 2 * a.fs_stride + w0 - 2, *p0 = r0 + w0 - 2;       // (p2, p0: offset + 2 >= 0)
        const uint32_t lo_lim = w0 >= 2 ? 0u : 2u - (uint32_t)w0;
        const uint64_t left = a.total - w0;             // bases from the tile's first to the table's end (>= span)
        const uint32_t hi_lim = left + 1 < 0x7fffffffull ? (uint32_t)left + 1 : 0x7fffffffu;       // offset + 2 of the table's last entry
        const uint32_t d2 = fwd ? 1u : 3u, d0 = fwd ? 0u : 4u;                 // b -/+ 1 and b -/+ 2, + 2
        if (lo_lim == 0 && hi_lim >= (uint32_t)(EL * BLOCK) + 3u) {         // (every tile but the batch's first and last few)
            // no test against the span: a lane beyond it reads the next tile's entries (they are on their way anyway) and nobody uses them
            // (scalar row pointers + ONE 32-bit byte offset per lane + an immediate per element: no address arithmetic per load --
            // with 64-bit lane addresses every load cost four vector instructions, a tenth of the kernel's)
            const char *c1 = (const char *)p1, *c2 = (const char *)(p2 + d2), *c0 = (const char *)(p0 + d0);
            const uint32_t t4 = (uint32_t)sizeof(row_t) * tid;
#pragma unroll
            for (int i = 0; i < EL; i++) {
                const uint32_t bo = t4 + (uint32_t)(sizeof(row_t) * BLOCK * i);
                tmp[1][i] = *(const row_t *)(c1 + bo); tmp[2][i] = *(const row_t *)(c2 + bo); tmp[0][i] = *(const row_t *)(c0 + bo);
            }
        } else {
#pragma unroll
            for (int i = 0; i < EL; i++) {
                const uint32_t b = tid + (uint32_t)BLOCK * i;
                if (b < span) {
                    uint32_t x2 = b + d2, x0 = b + d0;
                    x2 = x2 < lo_lim ? lo_lim : x2 > hi_lim ? hi_lim : x2;
                    x0 = x0 < lo_lim ? lo_lim : x0 > hi_lim ? hi_lim : x0;
                    tmp[1][i] = p1[b];
                    tmp[2][i] = p2[x2];
                    tmp[0][i] = p0[x0];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < PW; u++) {                  // (packed reads have hundreds of guard words on both sides)
            const uint32_t i = tid + (uint32_t)BLOCK * u;
            tpk[u] = i < NPK ? (a.packed + ((int64_t)(w0 >> 4) - 2))[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < PR; u++) {
            const uint32_t i = tid + (uint32_t)BLOCK * u;
            tro[u] = i <= nfit ? ((const uint32_t *)(a.read_off + first))[2 * i] : 0u;        // (the low word is all that is needed)
            tis[u] = a.read_isl && i < nfit ? (a.read_isl + first)[i] : a.ignore_score_len;
            trn[u] = G32 && PRN && i < nfit ? (a.read_null + first)[i] : 0u;
        }
        if (G32 && PRN) {                       // (its index came with the tile's geometry: no wait here)
            const float *nt = a.null_tab + (size_t)__builtin_amdgcn_readfirstlane(t.rn0) * MG_NULL_FLOATS;
#pragma unroll
            for (int u = 0; u < PN1; u++) {
                const uint32_t e = tid + (uint32_t)BLOCK * u;
                tn1[u] = nt[null_src(e < MG_NULL_FLOATS ? e : 0u, fwd)];
            }
        }
        if (tid < n_orf) {
            const gmg_mg_orf *o = a.orfs + o0 + tid;
            po_read = o->read; po_frame = o->frame; po_lo = o->lo; po_hi = o->hi;
            po_s0 = ((const uint32_t *)(a.start_off + o0))[2 * tid];
            po_s1 = ((const uint32_t *)(a.start_off + o0))[2 * tid + 2];
        }
    };

    uint64_t k = blockIdx.x;
    MtTile cur, nxt, nx2;
    meta_a(k, cur);
    meta_a(k + gridDim.x, nxt);
    meta_a(k + 2 * (uint64_t)gridDim.x, nx2);
    meta_b(cur);
    meta_b(nxt);
    issue(k, cur);
#if GMG_MT_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = __builtin_readcyclecounter();
#endif
    for (; k < 2 * n_tiles; k += gridDim.x) {           // (tile, strand)
        const bool fwd = (k & 1) == 0;
        // (wave-uniform all of them; the loads they come from are older than the tile's rows, so this waits for nothing new)
        const uint32_t nfit = __builtin_amdgcn_readfirstlane(cur.nfit), span = __builtin_amdgcn_readfirstlane(cur.span);
        const uint32_t first = __builtin_amdgcn_readfirstlane(cur.first), o0 = __builtin_amdgcn_readfirstlane(cur.o0);
        const uint32_t n_orf = __builtin_amdgcn_readfirstlane(cur.o1) - o0;
        const uint32_t w0_lo = __builtin_amdgcn_readfirstlane((uint32_t)cur.w0);
        __syncthreads();                                // the previous tile has left the LDS
        MT_STAMP(6);
        // ---- stage 0
        if (nfit) {
            uint32_t tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
#pragma unroll
            for (int i = 0; i < EL; i++) {
                const uint32_t b = tid + (uint32_t)BLOCK * i;
                // (beyond the span T goes to its own padding slot: no branch)
                if (i < EL - 1 || b < CAP) {
                    s_val[b < span ? (fwd ? span - 1 - b : b) : b] = ((double)tmp[1][i] + (double)tmp[2][i]) + (double)tmp[0][i];
                    s_oinfo[b] = 0;
                }
            }
#pragma unroll
            for (int u = 0; u < PW; u++) {
                const uint32_t i = tid + (uint32_t)BLOCK * u;
                if (i < NPK) s_packed[i] = tpk[u];
            }
#pragma unroll
            for (int u = 0; u < PR; u++) {
                const uint32_t i = tid + (uint32_t)BLOCK * u;
                if (i <= nfit) s_roff[i] = tro[u] - w0_lo;
                if (i < nfit) s_isl[i] = tis[u];
                if (G32 && PRN && i < nfit && i < NC) s_rnull[i] = trn[u];
            }
            if (G32 && PRN) {
#pragma unroll
                for (int u = 0; u < PN1; u++) {
                    const uint32_t e = tid + (uint32_t)BLOCK * u;
                    if (e < MG_NULL_FLOATS) s_nullm[0][e] = tn1[u];
                }
            }
            if (tid == 0) s_nq = 0;
        }
        // the lane's first ORF of this tile stays in registers through the stages; the loads of the next tile overwrite po_*
        const uint32_t mo_read = po_read, mo_so = po_s0, mo_nt = po_s1 - po_s0;
        const int32_t mo_frame = po_frame, mo_lo = po_lo, mo_hi = po_hi;
        __syncthreads();
        MT_STAMP(0);                                    // stage 0 (waits for the tile's loads)
        // read of tile base b: index in the tile, first base, length
        auto read_of = [&](uint32_t b, uint32_t &rl, int &rs, int &n) __attribute__((always_inline)) {
            if (a.uniform_len > 0) {
                rl = a.uniform_len == 1 ? b : __umulhi(b, a.uniform_magic);
                rs = (int)(rl * (uint32_t)a.uniform_len);
                n = a.uniform_len;
            } else {
                uint32_t lo = 0, hi = nfit;             // last r with s_roff[r] <= b
                while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_roff[mid] <= b) lo = mid; else hi = mid; }
                rl = lo;
                rs = (int)s_roff[lo];
                n = (int)s_roff[lo + 1] - rs;
            }
        };
        // ORF e of the tile (e = tid: from the registers)
        auto orf_of = [&](uint32_t e, uint32_t &rd, int32_t &frame, int32_t &lo, int32_t &hi, uint32_t &so, uint32_t &nt) __attribute__((always_inline)) {
            if (e == tid) { rd = mo_read; frame = mo_frame; lo = mo_lo; hi = mo_hi; so = mo_so; nt = mo_nt; return; }
            const gmg_mg_orf *o = a.orfs + o0 + e;
            rd = o->read; frame = o->frame; lo = o->lo; hi = o->hi;
            const uint64_t s0 = a.start_off[o0 + e];
            so = (uint32_t)s0;
            nt = (uint32_t)(a.start_off[o0 + e + 1] - s0);
        };
        // geometry of an ORF of this strand: truncated?  region length, u of the region's first base
        auto orf_geo = [&](uint32_t rd, int32_t lo, int32_t hi, bool &trunc, int &m, uint32_t &uh) __attribute__((always_inline)) {
            const uint32_t rl = (uint32_t)(rd - first);
            const int rs = a.uniform_len > 0 ? (int)(rl * (uint32_t)a.uniform_len) : (int)s_roff[rl];
            const int n = a.uniform_len > 0 ? a.uniform_len : (int)s_roff[rl + 1] - rs;
            m = hi - lo;
            trunc = a.allow_truncated && (fwd ? lo < 3 : n - (hi - 1) < 3);
            const uint32_t bh = (uint32_t)(rs + (fwd ? hi - 1 : lo - 1));
            uh = fwd ? span - 1 - bh : bh;
        };

        // The null model's PARTIAL windows (a buffer's first two positions: the read's last two bases on the forward strand, its first
        // two on the reverse strand) are settled once per read instead of in a branch of the scan's element loop (a wave took it in
        // 2 - 3 of its 8 iterations: a tenth of the kernel's instructions): the scan subtracts the FULL-window values everywhere; the
        // two elements of the read that hold a partial-window term get (full - partial) added in front.  Exact: every quantity is a
        // multiple of the batch's grid and far below 2^53 of it (mg_run's test), in any order.
        // Element at base si, term t: sub-model (1, 2, 0)[t] at base x = si -/+ t, buffer position j = n - 1 - x / x.
        auto partial_fix = [&](uint32_t r, int rs, int n, auto tab) __attribute__((always_inline)) {
            (void)r;
#pragma unroll
            for (int e2 = 0; e2 < 2; e2++) {                // the read's last / first base, and the one beside it
                const int si = fwd ? n - 1 - e2 : e2;
                if (si < 0 || si >= n) continue;
                double fix = 0.0;
#pragma unroll
                for (int t = 0; t < 2 - e2; t++) {          // (e2 = 0: terms 0 and 1 sit at j = 0, 1; e2 = 1: term 0 at j = 1)
                    const int x = fwd ? si - t : si + t, j = e2 + t;
                    if (x < 0 || x >= n) continue;
                    const int fr = t == 0 ? 1 : 2;
                    // six bits from base y = x (forward) / x - 2 (reverse) of the tile, as stage 2's window holds them
                    const int y = rs + (fwd ? x : x - 2);
                    const uint32_t X = 2u * (uint32_t)(32 + (int)(w0_lo & 15u) + y);
                    const uint64_t two = (uint64_t)s_packed[(X >> 5) + 1] << 32 | s_packed[X >> 5];
                    const uint32_t v = (uint32_t)(two >> (X & 31u)) & 63u;
                    // S[x] and its neighbour towards the read's inside (forward S[x+1], reverse S[x-1]), as the buffer holds them
                    const uint32_t c0 = fwd ? v & 3u : (v >> 4) & 3u, c1 = (v >> 2) & 3u;
                    const uint32_t b0c = fwd ? c0 : c0 ^ 3u, b1c = fwd ? c1 : c1 ^ 3u;
                    const uint32_t part = 192u + (uint32_t)fr * 20u + (j == 1 ? 4u + (b1c | b0c << 2) : b0c);
                    fix += (double)tab[(uint32_t)fr * 64u + v] - (double)tab[part];
                }
                const uint32_t b = (uint32_t)(rs + si), u = fwd ? span - 1u - b : b;
                s_val[u] += fix;
            }
        };
        // ---- stage 1: one lane per ORF of the tile: where its region starts
        if (nfit) {
            for (uint32_t e = tid; e < n_orf; e += BLOCK) {
                uint32_t rd, so, nt, uh; int32_t frame, lo, hi; bool trunc; int m;
                orf_of(e, rd, frame, lo, hi, so, nt);
                if ((frame > 0) != fwd) continue;
                orf_geo(rd, lo, hi, trunc, m, uh);
                if (m > 0) atomicOr((uint32_t *)s_oinfo + (uh >> 1), (e + 1u) << (16u * (uh & 1u)));
            }
            if (DIST)                                   // the six edge bases of every read of the tile
                for (uint32_t r = tid; r < nfit; r += BLOCK) {
                    const int rs = (int)s_roff[r], n = (int)s_roff[r + 1] - rs;
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        const int si = k < 3 ? k : n - 1 - (k - 3);
                        if (si < 0 || si >= n) continue;
                        const uint32_t ds = si < 3 ? (uint32_t)si : 3u, de = n - 1 - si < 3 ? (uint32_t)(n - 1 - si) : 3u;
                        const uint32_t b = (uint32_t)(rs + si), u = fwd ? span - 1u - b : b;
                        atomicOr((uint32_t *)s_oinfo + (u >> 1), ((3u - ds) | (3u - de) << 2) << (12u + 16u * (u & 1u)));
                    }
                    if (G32 && !PRN) partial_fix(r, rs, n, s_nulld);
                }
            if (G32 && PRN) {
                const uint32_t nc = nfit < NC ? nfit : NC;
                for (uint32_t i = MG_NULL_FLOATS + tid; i < nc * MG_NULL_FLOATS; i += BLOCK) {       // (the first read's came with the tile)
                    const uint32_t rl = i / MG_NULL_FLOATS, e = i - rl * MG_NULL_FLOATS;
                    s_nullm[rl][e] = a.null_tab[(size_t)s_rnull[rl] * MG_NULL_FLOATS + null_src(e, fwd)];
                }
            }
        }
        __syncthreads();                                // (the null tables of the tile's reads, the edge marks: stage 2 reads both)
        MT_STAMP(2);                                    // stage 1
        // ---- stage 2: the scan.  Lane (class c, part jl) owns elements u = ub + 3 i, i < EL: every third base of 27
        // consecutive ones.  The ten codons of its class that surround them come out of ONE 64-bit window of the packed bases as
        // 6-bit fields C[k] = S[x] | S[x+1] << 2 | S[x+2] << 4, x = sb + 3 k; the stop-codon sets and the start-codon table are
        // indexed that way (MgArgs::*_nat, s_whf / s_whr).
        //   reverse strand (u = b):         element i is base sb + 3 + 3 i; a segment starts behind the stop codon C[i]
        //                                   (bases b-3 .. b-1), the start codon at it is C[i+1] (bases b .. b+2)
        //   forward strand (u = span-1-b):  element i is base sb + FW0 - 3 i (FW0 = 3 (EL-1) + 2); a segment starts below the stop
        //                                   codon C[EL-i] (bases b+1 .. b+3), the start codon at it is C[EL-1-i] (bases b-2 .. b)
        if (nfit) {
            uint32_t tid2 = threadIdx.x;
            asm volatile("" : "+v"(tid2));
            const uint32_t wv = tid2 >> 6, l = tid2 & 63u;
            const uint32_t c = l >= 2 * MT_CL ? 2u : l >= MT_CL ? 1u : 0u;
            const bool idle = l == 63;
            const uint32_t jl = idle ? MT_CL - 1 : l - MT_CL * c;          // (the idle lane repeats its neighbour's work and drops it)
            const uint32_t ub = WV * wv + 3 * EL * jl + c;
            double es[EL];
            uint32_t ep[EL];                         // scan word | start codon here << 26 | (which + 1) << 27
            double acc = 0.0;
            uint32_t p = 0;
            auto local = [&](auto FWD_) __attribute__((always_inline)) {
                constexpr bool FWD = decltype(FWD_)::value;
                const int b0 = FWD ? (int)span - 1 - (int)ub : (int)ub;            // base of element 0 (beyond the tile's reads: padding)
                constexpr int FW0 = 3 * (EL - 1) + 2;                             // forward strand: element 0 sits FW0 bases behind the window's first
                int sb = FWD ? b0 - FW0 : b0 - 3;
                if (sb < -FW0) sb = -FW0;                                            // (a lane of padding only: any window will do)
                const uint32_t X = 2u * (uint32_t)(32 + (int)(w0_lo & 15u) + sb);  // two words in front of the tile's first
                const uint32_t *pw = s_packed + (X >> 5);
                const uint32_t q0 = pw[0], q1 = pw[1], q2 = pw[2];
                const uint32_t wlo = __builtin_amdgcn_alignbit(q1, q0, X & 31u), whi = __builtin_amdgcn_alignbit(q2, q1, X & 31u);
                const uint64_t win = (uint64_t)whi << 32 | wlo;                    // base sb + t at bits 2t
                uint32_t rl = 0; int rs = 0, n = 0;
                if (!DIST && (uint32_t)b0 < span) read_of((uint32_t)b0, rl, rs, n);
                // the table bytes of the EL + 1 codon fields this lane's elements meet (element i: stop test on one, start test on the next)
                uint32_t wb[EL + 1];
#pragma unroll
                for (int t = 0; t <= EL; t++) wb[t] = s_wh[(uint32_t)(win >> (6 * t)) & 63u];
#pragma unroll
                for (int i = 0; i < EL; i++) {
                    const uint32_t u = ub + 3 * i;
                    const int b = FWD ? b0 - 3 * i : b0 + 3 * i;
                    const bool valid = (uint32_t)b < span;
                    if (!DIST && valid) {
                        if (FWD) while (b < rs) { rl--; rs = (int)s_roff[rl]; n = (int)s_roff[rl + 1] - rs; }
                        else while (b >= rs + n) { rl++; rs = (int)s_roff[rl]; n = (int)s_roff[rl + 1] - rs; }
                    }
                    const int si = b - rs;
                    // DIST: 3 - (distance to the read's first base, to its last base), both capped at 3; the walk runs away from `near`
                    const uint32_t dd = DIST ? (uint32_t)s_oinfo[u] >> 12 : 0u;
                    const uint32_t near3 = FWD ? dd >> 2 : dd & 3u, far3 = FWD ? dd & 3u : dd >> 2;
                    const uint32_t bs = wb[FWD ? EL - i : i], bw = wb[FWD ? EL - 1 - i : i + 1];
                    const bool st = !valid || (DIST ? near3 != 0 : (FWD ? si + 3 >= n : si < 3)) || (bs & 0x80u);
                    const bool geo = DIST ? far3 <= 1u : (FWD ? si >= 2 : si + 2 <= n - 1);
                    const uint32_t wh = bw & 0x7fu;
                    // (a codon may be in the start set AND in the stop set -- "-A nnn": the stop codon that ends a region is no start)
                    const uint32_t cand = valid && geo && bw - 1u < 0x7fu ? 1u : 0u;
                    double T = s_val[u];
                    if (G32 && valid) {
                        // T holds the gene model's three values; the null model's: sub-model 1 at x = b, 2 at x = b -/+ 1, 0 at
                        // x = b -/+ 2 (forward / reverse), buffer position j = n-1-x / x
                        const int bitb = FWD ? 2 * FW0 - 6 * i : 6 * i + 6;             // bit position of S[b] in the window
                        // entry `off` of the read's table in this strand's order
                        auto nullf = [&](uint32_t off) __attribute__((always_inline)) -> double {        // a full-window entry
                            if (!PRN) return s_nulld[off];
                            return (double)s_nullm[rl < NC ? rl : 0u][off];       // (mg_run lets a tile take MT_NC reads at most then)
                        };
                        // (one null model: the partial windows at the read's end were settled in front of the scan -- partial_fix: full
                        // windows everywhere here.  A null model per read keeps the branch: its tables arrive in LDS with stage 1, and
                        // a pass over the tile's reads behind that barrier costs more than the branch -- 11.36 against 10.73 ms per
                        // 1 M reads, the fix from the set in L2 11.64: profiles/r05_prn_ab.txt)
                        double nsum;
                        if (PRN && (FWD ? si + 2 >= n : si < 2)) {                  // one of them is a partial window
                            nsum = 0.0;
#pragma unroll
                            for (int t = 0; t < 3; t++) {                          // (no branches in here: selects)
                                const int fr = t == 0 ? 1 : t == 1 ? 2 : 0;
                                const int xs = FWD ? si - t : si + t;              // the term's base in its read
                                const int j = FWD ? n - 1 - xs : xs;
                                const int bx = bitb + (FWD ? -2 * t : 2 * t);      // bit position of S[x]
                                const uint32_t c0 = (uint32_t)(win >> bx) & 3u;
                                const uint32_t c1 = (uint32_t)(win >> (FWD ? bx + 2 : bx - 2)) & 3u;
                                const uint32_t b0c = FWD ? c0 : c0 ^ 3u, b1c = FWD ? c1 : c1 ^ 3u;
                                const uint32_t v = (uint32_t)(win >> (FWD ? bx : bx - 4)) & 63u;
                                const uint32_t off = j >= 2 ? fr * 64 + v : 192u + fr * 20 + (j == 1 ? 4u + (b1c | b0c << 2) : b0c);
                                const double nv = nullf(off);
                                nsum += xs >= 0 && xs < n ? nv : 0.0;              // (beyond the read: T is never used then)
                            }
                        } else {
                            const uint32_t v1 = (uint32_t)(win >> (FWD ? bitb : bitb - 4)) & 63u, v2 = (uint32_t)(win >> (bitb - 2)) & 63u;
                            const uint32_t v0 = (uint32_t)(win >> (FWD ? bitb - 4 : bitb)) & 63u;
                            nsum = (nullf(64 + v1) + nullf(128 + v2)) + nullf(v0);
                        }
                        T -= nsum;
                    }
                    if (st) { acc = 0.0; p = (u << 12) | MT_REAL | MT_BLK; }
                    es[i] = acc;
                    ep[i] = p | cand << 26 | wh << 27;
                    acc += T;
                    p += cand;
                }
            };
            if (fwd) local(std::integral_constant<bool, true>()); else local(std::integral_constant<bool, false>());
            // the totals of the lanes before this one in its class: shift by one lane, then an inclusive segmented scan
            double xs;
            uint32_t xp;
            {
                const unsigned long long bb = (unsigned long long)__double_as_longlong(acc);
                const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)bb, 0x138, 0xf, 0xf, false);      // wave_shr:1
                const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(bb >> 32), 0x138, 0xf, 0xf, false);
                xp = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p, 0x138, 0xf, 0xf, false);
                xs = __longlong_as_double((long long)((unsigned long long)hi << 32 | lo));
                if (jl == 0) { xs = 0.0; xp = MT_BLK; }
            }
            mt_scan_step<0x111, 0xf>(xs, xp);           // row_shr:1
            mt_scan_step<0x112, 0xf>(xs, xp);           // row_shr:2
            mt_scan_step<0x114, 0xf>(xs, xp);           // row_shr:4
            mt_scan_step<0x118, 0xf>(xs, xp);           // row_shr:8
            mt_scan_step<0x142, 0xa>(xs, xp);           // row_bcast:15 into rows 1 and 3
            mt_scan_step<0x143, 0xc>(xs, xp);           // row_bcast:31 into rows 2 and 3
            if (NW > 1) {
                // what the waves before this one leave open in each class
                if (!idle && jl == MT_CL - 1) {
                    const bool blk = (p & MT_BLK) != 0;
                    s_wsum[wv][c] = blk ? acc : xs + acc;
                    s_wp[wv][c] = blk ? p : xp + p;
                }
            }
            __syncthreads();                            // (the ORFs' region starts are in place too)
            if (NW > 1) {
                if (!(xp & MT_REAL)) {
                    double ws = 0.0;
                    uint32_t wp = 0;
                    for (uint32_t w2 = 0; w2 < wv; w2++) {
                        const double s2 = s_wsum[w2][c];
                        const uint32_t p2 = s_wp[w2][c];
                        const bool blk = (p2 & MT_REAL) != 0;            // (a wave's first lanes block without a segment start)
                        ws = blk ? s2 : ws + s2;
                        wp = blk ? p2 : wp + (p2 & 0xfffu);
                    }
                    xs = ws + xs;
                    xp = (wp & ~MT_BLK) + (xp & 0xfffu);
                }
            }
#pragma unroll
            for (int i = 0; i < EL; i++) {
                const uint32_t u = ub + 3 * i;
                const bool own = (ep[i] & MT_REAL) != 0;                   // the segment starts inside this lane's elements
                const double cum = own ? es[i] : xs + es[i];
                const uint32_t pp = own ? ep[i] : xp + (ep[i] & 0xfffu);
                const uint32_t ch = pp & 0xffffffu;
                if (!idle) {
                    s_val[u] = cum;
                    s_ch[u] = ch;
                    if ((ep[i] >> 26) & 1u) {           // a start codon inside the read: a start if an ORF's region holds it at j >= lowest j
                        const uint32_t hd = ch >> 12;
                        if ((s_oinfo[hd] & 0xfffu) != 0 && (int)(u - hd) >= j_lo) s_q[atomicAdd(&s_nq, 1u)] = (uint16_t)(u | (ep[i] >> 27) << 12);
                    }
                }
            }
        }
        // the next tile's loads: behind the scan (the stage with the most registers alive), in flight through stages 3, 4 and
        // whatever the other waves of the CU are doing
        MT_STAMP(3);                                    // stage 2
        MtTile nx3;
        meta_a(k + 3 * (uint64_t)gridDim.x, nx3);
        meta_b(nx2);
#ifndef MT_ISSUE_LATE
        issue(k + gridDim.x, nxt);
#endif
        __syncthreads();
        MT_STAMP(1);                                    // the next tile's loads issued
#ifdef GMG_MT_LOADS_ONLY                                // diagnostic build: the load pattern alone (results invalid)
        if (s_val[tid] != 1.2345e300) { cur = nxt; nxt = nx2; nx2 = nx3; continue; }
#endif
        // ---- stages 3 and 4, MT_ORFS ORFs of the tile at a time
        for (uint32_t e0 = 0; e0 < (nfit ? n_orf : 0u); e0 += MT_ORFS) {
            for (uint32_t i = tid; i < MT_ORFS; i += BLOCK) {
                const uint32_t e = e0 + i;
                s_obest[i] = 0;
                s_ofj[i] = 0;
                if (e < n_orf) {
                    uint32_t rd, so, nt; int32_t frame, lo, hi;
                    orf_of(e, rd, frame, lo, hi, so, nt);
                    s_oso[i] = so;
                    s_ont[i] = nt;
                }
            }
            __syncthreads();
            const uint32_t nq = s_nq;
            for (uint32_t qi = tid; qi < nq; qi += BLOCK) {
                const uint32_t qe = s_q[qi], u = qe & 0xfffu;
                const uint32_t ch = s_ch[u], hd = ch >> 12;
                const uint32_t info = s_oinfo[hd] & 0xfffu;
                const uint32_t e = info - 1u;
                if (e < e0 || e >= e0 + MT_ORFS) continue;
                const int j = (int)(u - hd);
                const uint32_t t = (ch & 0xfffu) - (s_ch[hd + (uint32_t)j_lo] & 0xfffu);    // start codons of the region at lower j >= lowest j
                const uint32_t nt = s_ont[e - e0];
                if (t >= nt) continue;                  // (cannot happen: the count pass found the same start codons; never write outside the slice)
                const uint32_t slot = nt - 1u - t;
                const uint32_t b = fwd ? span - 1 - u : u;
                uint32_t rl; int rs, n;
                read_of(b, rl, rs, n);
                const int si = (int)b - rs;
                const double pend = s_val[u];
                const double sc = (j + 2 > s_isl[rl] && 0.0 > pend) ? 0.0 : pend;          // glimmer-mg.cc:1644-1646
                gmg_start st;
                st.score = sc; st.j = j + 2; st.pos = fwd ? si - 1 : si + 3;
                st.which = (int32_t)(qe >> 12) - 1; st.truncated = 0; st.first = slot == 0 ? 1 : 0;
                a.starts[(uint64_t)s_oso[e - e0] + slot] = st;
                atomicMax(&s_obest[e - e0], (unsigned long long)mg_ord(sc));
                if (slot == 0) s_ofj[e - e0] = (uint16_t)(j + 2);
            }
            __syncthreads();
            for (uint32_t i = tid; i < MT_ORFS && e0 + i < n_orf; i += BLOCK) {
                const uint32_t e = e0 + i;
                uint32_t rd, so, nt, uh; int32_t frame, lo, hi; bool trunc; int m;
                orf_of(e, rd, frame, lo, hi, so, nt);
                if ((frame > 0) != fwd) continue;
                orf_geo(rd, lo, hi, trunc, m, uh);
                double best = s_obest[i] ? mg_unord(s_obest[i]) : -DBL_MAX;
                int first_j = (int)s_ofj[i];
                const int jmax = m >= 1 ? (m - 1) / 3 * 3 : -1;
                if (trunc && jmax >= j_lo && nt > 0) {  // the truncated start: slot 0
                    const uint32_t uj = uh + (uint32_t)jmax;
                    const uint32_t b = fwd ? span - 1 - uj : uj;
                    const uint32_t rl = (uint32_t)(rd - first);
                    const int rs = a.uniform_len > 0 ? (int)(rl * (uint32_t)a.uniform_len) : (int)s_roff[rl];
                    const int si = (int)b - rs;
                    const double pend = s_val[uj];
                    const double sc = (jmax + 2 > s_isl[rl] && 0.0 > pend) ? 0.0 : pend;
                    gmg_start st;
                    st.score = sc; st.j = jmax + 2; st.pos = fwd ? si - 1 : si + 3;
                    st.which = -1; st.truncated = 1; st.first = 1;
                    a.starts[so] = st;
                    if (sc > best) best = sc;
                    first_j = jmax + 2;
                }
                gmg_mg_orf *o = a.orfs + o0 + e;
                o->orf_is_truncated = trunc;
                o->start_begin = so;
                o->n_starts = nt;
                o->first_j = nt ? first_j : 0;
                const bool ok = nt > 0 && first_j + 1 >= mgl;               // glimmer-mg.cc:1656-1676
                o->best_score = ok ? best : -DBL_MAX;
                o->accepted = ok && best > a.start_threshold;
            }
            if (e0 + MT_ORFS < n_orf) __syncthreads();
        }
        MT_STAMP(4);                                    // stages 3 and 4
#ifdef MT_ISSUE_LATE
        issue(k + gridDim.x, nxt);
#endif
        cur = nxt;
        nxt = nx2;
        nx2 = nx3;
    }
#if GMG_MT_STAMPS
    if ((tid & 63u) == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&g_mt_stamps[i], st_acc[i]);
#endif
}

// the ORFs of the reads no tile took (k_mg_unfit_list; their running sums come from k_mg_cum): one lane per read
__global__ __launch_bounds__(256) void k_mg_starts_unfit(MgArgs a)
{
    __shared__ int8_t s_which[64];
    if (threadIdx.x < 64) s_which[threadIdx.x] = a.which[threadIdx.x];
    __syncthreads();
    const uint64_t n = *a.unfit_n;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = a.unfit[e];
        for (uint64_t i = a.read_orf_off[r]; i < a.read_orf_off[r + 1]; i++) mg_starts_one<true>(a, i, s_which);
    }
}

// ---------------------------------------------------------------------------------------------------
// The error branch: Score_Orf_Starts with Score_Indels (-i) or the substitution branch (-s),
// src/Glimmer/glimmer-mg.cc:1513-1602, 1693-1861.  The reference recurses: while it scans an ORF from the 5'
// end to the 3' end it branches, at every low-quality base (-i) or once through the previous stop codon (-s),
// into another Score_Orf_Starts that starts in another reading frame with the score of the suffix kept so far;
// every call pushes its starts onto ONE list.  Each call sums Frame_Scores sequentially from ITS OWN end point
// (Cumulative_Frame_Score, :561-604), so there are no shared running sums as in k_mg_cum.
// Two implementations, bit-identical (tests/test_gpu_mg_err.py runs both against the oracle and each other):
// k_mg_err_level (further down; the default) walks the call tree level by level with one lane per call and sorts
// each ORF's starts into the push order afterwards; k_mg_err_flat (next) lets one lane walk one ORF's whole tree
// and writes the exact slots -- slower (lopsided trees, every lane on its own rows), but it needs no order keys
// and no call arrays, so it is the fallback for reads too long for the keys and for a full call array.
//
// A lane walks every call from the 3' end to the 5' end (the direction of the sum; the end of the region is
// the first in-frame stop codon it meets, which is what the Fwd_Prev_Stops / Rev_Next_Stops tables hold,
// :675-729) -- the REVERSE of the reference's scan.  So it visits the call tree in exactly reversed order
// (per position: own start, insertion branch, deletion branch; the substitution branch last) and fills the
// ORF's slice of the start array from the back: entry by entry that reproduces the reference's push order,
// which the caller needs because the reference sorts the list with an unstable sort on pos alone
// (glimmer_base.hh:90) and paths tie on pos.  A count pass sizes the slices.
//   * an in-frame codon is handled one codon late, when it is known whether it was the last of the region: the
//     truncated start (:1818-1846) goes between the real start of that codon and its indel branches;
//   * "first" (:1833) is the 5'-most start of a call: the last one this walk emits, patched at the end.
// ---------------------------------------------------------------------------------------------------
#define MG_NO_SLOT 0xffffffffu

struct MgErrRun {                // what one ORF's walk accumulates
    uint32_t count, end;         // starts so far; WRITE: one past the next free slot (filled backwards)
    double best;                 // best boosted score
    int ext_pos, ext_jmin, ext_jmax;     // the extreme pos (lowest forward, highest reverse) and the j's seen there
    int m0, trunc0;              // the ORF's own call: region length and orf_is_truncated
};

// Set_Quality_454 (glimmer-mg.cc:1865-1906) / Clean_Quality_454 (:519-546).  One wave per read, 64 bases at a time: "this base equals
// the one before it" as a 64-bit mask, the length of the homopolymer run that ends at a base = the distance to the last clear bit at or
// below its own (a run that reaches the chunk's first base goes on with what the chunk before ended with) -- no search for the read of a
// base, no loop back over the run (one lane per base: 2.3 ms per 1M reads).
__global__ __launch_bounds__(256) void k_mg_quality(MgArgs a, const uint8_t *user, uint8_t *out, uint8_t *walk_q)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave; r < a.n_reads; r += n_waves) {
        const uint64_t b = a.read_off[r];
        const uint32_t n = (uint32_t)(a.read_off[r + 1] - b);
        uint32_t carry = 0;                             // the run that ends at the last base of the chunk before (<= 6)
        for (uint32_t t0 = 0; t0 < n; t0 += 64) {
            const uint32_t t = t0 + lane;
            const bool in = t < n;
            const uint64_t g = b + (in ? t : 0);
            const uint64_t x = dev_window_bits(a.packed, (int64_t)g - 1);         // bases g - 1, g, g + 1 in the low six bits
            const uint32_t cp = (uint32_t)x & 3u, c = (uint32_t)(x >> 2) & 3u, cn = (uint32_t)(x >> 4) & 3u;
            const uint64_t eq = __ballot(in && t > 0 && c == cp);
            const uint64_t below = ~eq & ((2ull << lane) - 1ull);                 // the clear bits at or below this lane
            uint32_t run = below ? lane - (63u - (uint32_t)__clzll((long long)below)) + 1u : lane + 1u + carry;
            if (run > 6u) run = 6u;
            const bool inside = in && t + 1 < n && cn == c;                       // not the last base of its homopolymer run
            int q;
            if (user) {
                q = user[g];
                if (q <= 0) q = 1;
                if (inside && q < a.indel_q_thr + 1) q = a.indel_q_thr + 1;
            } else if (inside) q = 31;
            else q = run < 6u ? 31 - 5 * (int)run : 6;
            if (in) {
                out[g] = (uint8_t)(q > 255 ? 255 : q);
                if (walk_q) walk_q[a.total - 1 - g] = (uint8_t)(q > 255 ? 255 : q);     // last base first: what a forward walk reads
            }
            carry = (uint32_t)__builtin_amdgcn_readlane((int)run, 63);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// k_mg_err_flat: one lane per ORF, the whole call tree in the visiting order above, without nested loops.  (With the
// recursion written as nested calls a lane that enters a branch runs the whole inner call while the other 63 lanes
// of its wave wait at the call site, at every step of every level.)  ONE loop whose every trip advances every lane
// by one buffer position of whatever call it is in; a branch pushes the caller's state (two levels at most ever
// wait: 0 and 1) onto a stack in LDS and a finished call pops it.
// ---------------------------------------------------------------------------------------------------
#define MG_ERR_BLOCK 256

struct MgErrStack {              // saved callers, [level][lane]
    double sum[2][MG_ERR_BLOCK], prev[2][MG_ERR_BLOCK], suffix_score[2][MG_ERR_BLOCK];
    int end_point[2][MG_ERR_BLOCK], tp[2][MG_ERR_BLOCK], suffix_j[2][MG_ERR_BLOCK];
    uint32_t bits[2][MG_ERR_BLOCK], last_own[2][MG_ERR_BLOCK];
};

template <bool WRITE>
__global__ __launch_bounds__(MG_ERR_BLOCK) void k_mg_err_flat(MgArgs a, const int accepted_only, const int only_unfit)
{
    __shared__ int8_t s_which[64];
    __shared__ MgErrStack st;
    if (threadIdx.x < 64) s_which[threadIdx.x] = a.which[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x;
    const int mgl = a.min_gene_len;
    const int lowest_j = mgl - 3 < 3 ? mgl - 3 : 3;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n_orfs; i += (uint64_t)gridDim.x * blockDim.x) {
        gmg_mg_orf o = a.orfs[i];
        if (only_unfit && a.read_fit[o.read]) continue;                  // the per-read kernel has this read
        if (WRITE && accepted_only && !o.accepted) continue;             // the count pass has judged it; nothing of it leaves the GPU
        const bool fwd = o.frame > 0;
        const int64_t off = (int64_t)a.read_off[o.read];
        const int n = (int)(a.read_off[o.read + 1] - a.read_off[o.read]);
        const int64_t dir = fwd ? -1 : 1;
        const uint32_t comp = fwd ? 0u : 3u;
        const double *row0 = a.fs + (uint64_t)(fwd ? 1 : 4) * a.fs_stride;      // rows of buffer positions j % 3 = 0, 1, 2 (:561-604)
        const double *row1 = a.fs + (uint64_t)(fwd ? 2 : 5) * a.fs_stride;
        const double *row2 = a.fs + (uint64_t)(fwd ? 0 : 3) * a.fs_stride;

        MgErrRun R;
        R.count = 0; R.end = WRITE ? (uint32_t)a.start_off[i + 1] : 0; R.best = -DBL_MAX; R.ext_pos = 0; R.ext_jmin = R.ext_jmax = 0;
        R.m0 = 0; R.trunc0 = 0;

        // the call being walked
        int level = 0, end_point = fwd ? o.stop_position - 1 : o.stop_position + 3, suffix_j = 0;
        double suffix_score = 0.0, sum = 0.0, prev = 0.0;
        int tp = 0, jj = 0, br = 0;                     // codon, position in it, next branch to try (0 insertion, 1 deletion, 2 none)
        uint32_t pidx = 0, nidx = 0, last_own = MG_NO_SLOT;
        bool is_last = false, trunc = false, first_done = false, walking = false;
        int e_pos[2] = {0, 0}, e_type[2] = {0, 0};
        int64_t g0 = 0, g = 0;
        uint32_t w = 0;
        int avail = 0;

        auto fetch = [&](int t, uint32_t &idx) __attribute__((always_inline)) -> bool {     // codon t of the buffer; true: the region ends before it
            if (avail - 3 * t < 3) { trunc = a.allow_truncated != 0; return true; }
            uint32_t c[3];
#pragma unroll
            for (int x = 0; x < 3; x++) {
                c[x] = ((w >> (2u * (unsigned)(g & 15))) & 3u) ^ comp;
                const int64_t g2 = g + dir;
                if ((g ^ g2) >> 4) w = a.packed[g2 >> 4];
                g = g2;
            }
            idx = c[2] << 4 | c[1] << 2 | c[0];
            return (a.fwd_stop >> idx) & 1;
        };
        auto enter = [&]() __attribute__((always_inline)) {          // start the call (end_point, suffix_score, suffix_j) at `level`
            sum = 0.0; prev = 0.0; tp = 0; jj = 0; br = 0; last_own = MG_NO_SLOT;
            is_last = false; trunc = false; first_done = false;
            const int anchor = end_point - 1;
            walking = false;
            if (anchor < 0 || anchor >= n) { avail = 0; return; }   // an empty region (Fwd_Prev_Stop / Rev_Next_Stop outside the read)
            avail = fwd ? anchor + 1 : n - anchor;
            g0 = off + anchor;
            g = g0;
            w = a.packed[g >> 4];
            if (fetch(0, pidx)) return;                 // m = 0
            walking = true;
        };
        auto emit = [&](double raw, int j_full, int pos, int which, int truncated, int first) __attribute__((always_inline)) -> uint32_t {
            const int isl = a.read_isl ? a.read_isl[o.read] : a.ignore_score_len;
            const double sc = (j_full > isl && 0.0 > raw) ? 0.0 : raw;
            if (R.count == 0 || (fwd ? pos < R.ext_pos : pos > R.ext_pos)) { R.ext_pos = pos; R.ext_jmin = R.ext_jmax = j_full; }
            else if (pos == R.ext_pos) { if (j_full < R.ext_jmin) R.ext_jmin = j_full; if (j_full > R.ext_jmax) R.ext_jmax = j_full; }
            if (sc > R.best) R.best = sc;
            R.count++;
            uint32_t slot = MG_NO_SLOT;
            if (WRITE) {
                slot = --R.end;
                gmg_start s1;
                s1.score = sc; s1.j = j_full; s1.pos = pos; s1.which = which; s1.truncated = (int16_t)truncated; s1.first = (int16_t)first;
                a.starts[slot] = s1;
                gmg_start_errors er;
                er.pos[0] = level > 0 ? e_pos[0] : 0; er.pos[1] = level > 1 ? e_pos[1] : 0;
                er.type[0] = (int8_t)(level > 0 ? e_type[0] : 0); er.type[1] = (int8_t)(level > 1 ? e_type[1] : 0);
                er.n = (int8_t)level; er.reserved = 0;
                a.errs[slot] = er;
                if (a.keys) a.keys[slot] = slot - (uint32_t)a.start_off[i];     // already in push order
            }
            return slot;
        };
        auto push = [&](int child_end, double child_score, int child_sj, int epos, int etype) __attribute__((always_inline)) {
            st.sum[level][lane] = sum; st.prev[level][lane] = prev; st.suffix_score[level][lane] = suffix_score;
            st.end_point[level][lane] = end_point; st.tp[level][lane] = tp; st.suffix_j[level][lane] = suffix_j;
            st.bits[level][lane] = (uint32_t)jj | (uint32_t)br << 2 | pidx << 4 | nidx << 10 | (uint32_t)is_last << 16 | (uint32_t)trunc << 17 |
                                   (uint32_t)first_done << 18 | (uint32_t)walking << 19;
            st.last_own[level][lane] = last_own;
            e_pos[level] = epos; e_type[level] = etype;
            level++;
            end_point = child_end; suffix_score = child_score; suffix_j = child_sj;
            enter();
        };
        auto pop = [&]() __attribute__((always_inline)) {
            level--;
            if (level < 0) return;
            sum = st.sum[level][lane]; prev = st.prev[level][lane]; suffix_score = st.suffix_score[level][lane];
            end_point = st.end_point[level][lane]; tp = st.tp[level][lane]; suffix_j = st.suffix_j[level][lane];
            const uint32_t b = st.bits[level][lane];
            jj = b & 3; br = (b >> 2) & 3; pidx = (b >> 4) & 63; nidx = (b >> 10) & 63;
            is_last = (b >> 16) & 1; trunc = (b >> 17) & 1; first_done = (b >> 18) & 1; walking = (b >> 19) & 1;
            last_own = st.last_own[level][lane];
            const int anchor = end_point - 1;           // (a caller always has a region: it branched from inside it)
            avail = fwd ? anchor + 1 : n - anchor;
            g0 = off + anchor;
            g = g0 + dir * 3 * (tp + 2);                // the stream stands behind the look-ahead codon
            w = a.packed[g >> 4];
        };

        enter();
        while (level >= 0) {
            if (walking) {
                const int j = 3 * tp + jj;
                const int64_t gj = g0 + dir * j;
                const int k = fwd ? end_point - 2 - j : end_point + 2 + j;
                if (br == 0) {                          // first visit of this position
                    if (jj == 0) is_last = fetch(tp + 1, nidx);      // is codon tp the last of the region?
                    prev = sum;
                    sum = prev + (jj == 0 ? row0 : jj == 1 ? row1 : row2)[gj];
                    if (jj == 0 && j >= lowest_j && j + 3 + suffix_j >= mgl) {
                        const int which = s_which[pidx];
                        const double raw = (prev - 0.0) + suffix_score;
                        if (which >= 0) last_own = emit(raw, j + 2 + suffix_j, k, which, 0, 0);
                        if (is_last && trunc) { emit(raw, j + 2 + suffix_j, k, -1, 1, 1); first_done = true; }
                    }
                }
                bool pushed = false;
                if (a.err_mode == 1 && level < 2 && level < a.indel_max && j >= lowest_j) {
                    const int q = a.qual[gj];
                    if (q <= a.indel_q_thr) {           // Score_Indels, reversed order: insertion, then deletion
                        const double pen = a.pen[q];
                        while (br < 2) {
                            const int b = br++;
                            const double es = ((suffix_score + (b == 0 ? prev : sum)) - 0.0) + pen;
                            if (es > a.indel_suffix_thr) {
                                int ep, epos;
                                if (b == 0) { ep = fwd ? k - (2 - jj) : k + 2 - jj; epos = fwd ? k + 2 : k - 2; }
                                else { ep = fwd ? k + jj : k - jj; epos = fwd ? k + 3 : k - 1; }
                                push(ep, es, suffix_j + j + 2 - jj, epos, b);
                                pushed = true;
                                break;
                            }
                        }
                    }
                }
                if (pushed) continue;
                br = 0;
                if (++jj == 3) {
                    jj = 0;
                    if (is_last) walking = false;       // the region is done; tp + 1 codons
                    else pidx = nidx;
                    tp++;
                }
                continue;
            }
            // the region is finished: m = 3 * tp (tp counts the codons walked; 0 for an empty region)
            if (br != 3) {
                const int m = 3 * tp;
                if (level == 0) { R.m0 = m; R.trunc0 = trunc; }
                br = 3;                                 // (marks "substitution branch done" for the way back)
                if (level == 0 && a.err_mode == 2) {    // mutate the previous stop codon (:1771-1806)
                    const int lo = fwd ? end_point - m : end_point, hi = fwd ? end_point : end_point + m;
                    const int eep = fwd ? lo - 3 : hi + 3;
                    if (end_point - 1 >= 0 && end_point - 1 < n && eep >= 0 && eep - 2 < n) {
                        auto base = [&](int x) { const int64_t y = off + x; return (a.packed[y >> 4] >> (2u * (unsigned)(y & 15))) & 3u; };
                        const uint32_t want = fwd ? 0u : 3u;
                        const int a1 = base(fwd ? lo - 2 : hi) == want, a2 = base(fwd ? lo - 1 : hi - 1) == want;
                        double es = suffix_score + a.pass_stop[a1 * 2 + a2];
                        if (m > 0) es += sum - 0.0;
                        push(eep, es, suffix_j + m, fwd ? lo - 2 : hi + 2, 2);
                        continue;
                    }
                }
            }
            if (WRITE && !first_done && last_own != MG_NO_SLOT) a.starts[last_own].first = 1;
            pop();
        }

        if (fwd) { o.hi = o.stop_position - 1; o.lo = o.hi - R.m0; }
        else { o.lo = o.stop_position + 3; o.hi = o.lo + R.m0; }
        o.orf_is_truncated = (int16_t)R.trunc0;
        o.n_starts = R.count;
        o.first_j = R.count ? R.ext_jmin : 0;
        o.best_score = -DBL_MAX;
        o.accepted = 0;
        if (R.count > 0 && R.ext_jmax + 1 >= a.min_gene_len) {          // glimmer-mg.cc:1656-1676
            o.best_score = R.best;
            if (R.best > a.start_threshold) o.accepted = R.ext_jmin + 1 >= a.min_gene_len ? 1 : 2;
        }
        if (!WRITE) {
            a.orf_cnt[i] = (accepted_only && !o.accepted) ? 0u : R.count;
            if (o.accepted && a.acc_bits) atomicOr(&a.acc_bits[i >> 5], 1u << (i & 31u));
            o.start_begin = 0;
        } else o.start_begin = (uint32_t)a.start_off[i];
        a.orfs[i] = o;
    }
}

// ---------------------------------------------------------------------------------------------------
// The error branch level by level (the default).  One lane per CALL: level 0 = the ORFs' own calls, and every
// branch a walk meets is appended to the array of the next level, which the next launch walks -- three launches
// per pass.  A lane does one plain walk of one region, a whole in-frame codon per trip: no stack, and the walking
// lanes of a wave stay in the same phase (a wave executes whatever ANY of its lanes needs -- that, not occupancy
// or bandwidth, is what the earlier versions paid for; DESIGN.md 4.7 has the numbers).
// The calls found by the count pass stay in their arrays; the write pass walks them again (only those of the
// accepted ORFs when that is all the caller wants).  Calls finish in no particular order, so a start cannot know
// its slot in the reference's push order; it carries the order instead: key = (position, kind) of each level of
// its path, most significant level first, each field inverted -- ascending keys are the push order (the reverse of
// the visiting order described above).  Slots inside the ORF's slice are handed out by a counter, and a segmented
// sort by key puts every slice in order afterwards (mg_run, step 5).  Per-ORF results (count, best score, the j's
// at the extreme pos) are merged with 64-bit atomics on MgOrfAgg when a call ends.  Reads too long for the key
// fields (>= 2040 bases) and a full call array are left to k_mg_err_flat.
// ---------------------------------------------------------------------------------------------------
// A walk that starts at global base ga reads Frame_Scores[1], [2], [0], [1], ... at ga, ga -/+ 1, ...: three rows, 24 bytes
// apart in time.  The table is rewritten once (96 B / base) in walking order: for every class c = ga % 3 one row in which
// consecutive steps of such a walk are consecutive doubles,
//   forward:  walk[c][total-1-g]  = Frame_Scores[((c - g) mod 3 + 1) % 3][g]        (the walk runs down the read: reversed)
//   reverse:  walk[3+c][g]        = Frame_Scores[3 + ((g - c) mod 3 + 1) % 3][g]
// so a lane streams the 24 bytes of a codon from one address.
__global__ __launch_bounds__(256) void k_mg_walk_tables(MgArgs a, double *walk)
{
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < a.total; g += (uint64_t)gridDim.x * blockDim.x) {
        double v[6];
#pragma unroll
        for (int f = 0; f < 6; f++) v[f] = a.fs[(uint64_t)f * a.fs_stride + g];
        const int m = (int)(g % 3);
        const uint64_t rg = a.total - 1 - g;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int tf = (c - m + 3) % 3, tr = (m - c + 3) % 3;
            const int rf = (tf + 1) % 3, rr = (tr + 1) % 3;
            walk[(uint64_t)c * a.walk_stride + rg] = rf == 0 ? v[0] : rf == 1 ? v[1] : v[2];
            walk[(uint64_t)(3 + c) * a.walk_stride + g] = rr == 0 ? v[3] : rr == 1 ? v[4] : v[5];
        }
    }
}

// The same rows as RUNNING SUMS inside every read (inclusive, in walking order): when every sum of the batch is exact in any order (see
// k_mg_tile_starts), score[j] of a call that starts at walk index w is I[w + j] - I[w - 1] -- the reference's sequential sum, bit for
// bit -- so a walk needs no sum of its own and may jump over the codons at which nothing happens (k_mg_run_tables).
// One wave per (read, strand): 64 walk steps at a time, the three class rows scanned with DPP moves, carries in scalar registers.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double mg_dpp_add(double x)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, ROW_MASK, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return x + __longlong_as_double((long long)((unsigned long long)hi << 32 | lo));      // (lanes without a source add 0)
}
__device__ __forceinline__ double mg_wave_scan(double x)               // inclusive sum over the lanes of a wave
{
    x = mg_dpp_add<0x111, 0xf>(x);                      // row_shr:1
    x = mg_dpp_add<0x112, 0xf>(x);                      // row_shr:2
    x = mg_dpp_add<0x114, 0xf>(x);                      // row_shr:4
    x = mg_dpp_add<0x118, 0xf>(x);                      // row_shr:8
    x = mg_dpp_add<0x142, 0xa>(x);                      // row_bcast:15 into rows 1 and 3
    x = mg_dpp_add<0x143, 0xc>(x);                      // row_bcast:31 into rows 2 and 3
    return x;
}
// G32: the rows come from the fp32 gene table and the read's null model (the values k_mg_apply_nulls would put into the fp64
// table: one exact subtraction of two widened floats, partial windows at the read's ends from the partial tables) -- the
// error branch then never builds or reads the 48 B/base table.
template <bool G32, bool QONLY = false>
__global__ __launch_bounds__(256) void k_mg_walk_prefix(MgArgs a, double *walk)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t it = wave; it < 2 * a.n_reads; it += n_waves) {
        const uint64_t r = it >> 1;
        const bool fwd = (it & 1) == 0;
        const uint64_t off = a.read_off[r];
        const uint32_t n = (uint32_t)(a.read_off[r + 1] - off);
        const float *nt = G32 ? a.null_tab + (size_t)(a.read_null ? a.read_null[r] : 0u) * MG_NULL_FLOATS : nullptr;
        const uint32_t off_m3 = (uint32_t)(off % 3);
        double carry[3] = {0.0, 0.0, 0.0};
        for (uint32_t t0 = 0; t0 < n; t0 += 64) {
            const uint32_t t = t0 + lane;
            const bool in = t < n;
            const uint64_t g = in ? (fwd ? off + n - 1 - t : off + t) : off;      // base of walk step t
            double v[3];
            if (G32) {
                const uint32_t five = (uint32_t)dev_window_bits(a.packed, (int64_t)g - 2) & 0x3ffu, c0 = (five >> 4) & 3u;
                const int si = (int)(g - off);
#pragma unroll
                for (int f = 0; f < 3; f++) {
                    const float nv = fwd ? mg_null_value<true>(nt, f, si, (int)n, c0, (five >> 6) & 3u, (five >> 8) & 3u)
                                         : mg_null_value<false>(nt, f, si, (int)n, c0, (five >> 2) & 3u, five & 3u);
                    v[f] = in ? (double)a.gene32[(uint64_t)((fwd ? 0 : 3) + f) * a.fs_stride + g] - (double)nv : 0.0;
                }
            } else {
#pragma unroll
            for (int f = 0; f < 3; f++) v[f] = in ? a.fs[(uint64_t)((fwd ? 0 : 3) + f) * a.fs_stride + g] : 0.0;
            }
            const int m = (int)((off_m3 + (uint32_t)(g - off)) % 3u);             // g % 3 without a 64-bit division per lane
            const uint64_t w = fwd ? a.total - 1 - g : g;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int row = ((fwd ? c - m + 3 : m - c + 3) % 3 + 1) % 3;          // as k_mg_walk_tables
                const double x = row == 0 ? v[0] : row == 1 ? v[1] : v[2];
                const double sc = mg_wave_scan(x) + carry[c];
                // (streaming stores: 19 GB written once -- they leave the L2 to the run-length kernel beside this one; -0.7 ms per call)
                if (QONLY) {                            // the base's own class (the codon that starts here: c == m): the sum BEFORE this step
                    if (in && c == m) __builtin_nontemporal_store(sc - x, walk + (uint64_t)(fwd ? 0 : 1) * a.walk_stride + w);
                } else
                if (in) __builtin_nontemporal_store(sc, walk + (uint64_t)((fwd ? 0 : 3) + c) * a.walk_stride + w);
                const unsigned long long top = (unsigned long long)__double_as_longlong(sc);
                carry[c] = __longlong_as_double((long long)((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(top >> 32), 63) << 32 |
                                                               (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)top, 63)));
            }
        }
    }
}

// run[strand][w] = the number of consecutive codons of a walk, from the one that starts at walk index w, at which nothing can
// happen: no start codon, no base of low quality (run_q only), and the codon after it is there and is no stop codon -- so it is not
// the last of its region either.  One wave per (read, strand), 64 walk steps at a time from the read's last to its first: the "nothing
// happens" bits of a chunk as a 64-bit mask, the length of the run of set bits i, i + 3, i + 6, ... by doubling (runs of >= 2, 4, 8, 16
// as masks), what a run that reaches the chunk's end finds behind it carried from the chunk before.
__device__ __forceinline__ uint32_t mg_run_len(uint64_t b, uint32_t lane, const uint32_t carry[3])
{
    // the positions lane, lane + 3, lane + 6, ... of the chunk: the run ends at the first clear one (count trailing zeros of the
    // inverted, thinned mask; x / 3 = x * 43 >> 7 for x <= 66) or goes on into the chunk behind (carry by the phase it leaves with).
    // (The first form doubled the run length through masks b & b >> 3, & >> 6, ...: 35 vector instructions per mask, this one 18.)
    const uint64_t y = ~(b >> lane) & (0x9249249249249249ull & (~0ull >> lane));      // (positions beyond the chunk do not count)
    const uint32_t lo = (uint32_t)y, hi = (uint32_t)(y >> 32);
    const uint32_t first = lo ? (uint32_t)__builtin_ctz(lo) : 32u + (uint32_t)__builtin_ctz(hi | 0x80000000u);   // (unused when y == 0)
    const uint32_t n_all = ((66u - lane) * 43u) >> 7;                   // positions from this lane to the chunk's end
    const uint32_t out = lane + 3u * n_all - 64u;                       // 0 .. 2: where the run enters the next chunk
    const uint32_t len = y ? (first * 43u) >> 7 : n_all + (out == 0u ? carry[0] : out == 1u ? carry[1] : carry[2]);
    return len < 254u ? len : 254u;
}
__global__ __launch_bounds__(256) void k_mg_run_tables(MgArgs a, uint8_t *run_q, uint8_t *run_n)
{
    __shared__ int8_t s_which[64];
    if (threadIdx.x < 64) s_which[threadIdx.x] = a.which[threadIdx.x];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t it = wave; it < 2 * a.n_reads; it += n_waves) {
        const uint64_t r = it >> 1;
        const bool fwd = (it & 1) == 0;
        const int64_t off = (int64_t)a.read_off[r];
        const int n = (int)((int64_t)a.read_off[r + 1] - off);
        if (n <= 0) continue;
        const uint64_t w_first = fwd ? a.total - 1 - (uint64_t)(off + n - 1) : (uint64_t)off;     // walk index of walk step 0
        const uint8_t *qp = a.err_mode == 1 ? (fwd ? a.walk_q + w_first : a.qual + off) : nullptr;  // quality of walk step t: qp[t]
        uint8_t *oq = run_q + (fwd ? 0 : a.walk_stride) + w_first, *on = run_n + (fwd ? 0 : a.walk_stride) + w_first;
        uint32_t cq[3] = {0, 0, 0}, cn[3] = {0, 0, 0};
        uint64_t lq_next = 0;                           // "quality at or below the threshold" of the chunk behind (its steps 0 and 1 count here)
        for (int t0 = (n - 1) / 64 * 64; t0 >= 0; t0 -= 64) {
            const int t = t0 + (int)lane;
            bool bn = false;
            // one quality per lane; a codon's three positions are this lane's bit and the two above it in the wave's mask
            const uint64_t lq = __ballot(qp && t < n && qp[t] <= a.indel_q_thr);
            const uint64_t low3 = lq | (lq >> 1 | lq_next << 63) | (lq >> 2 | lq_next << 62);
            lq_next = lq;
            if (t + 5 <= n - 1) {                       // (the codon after this one is there)
                // the six bases of walk steps t .. t + 5: forward strand bases g, g-1, .., g-5, reverse strand g, g+1, .., g+5 complemented
                const int64_t g = fwd ? off + n - 1 - t : off + t, gs = fwd ? g - 5 : g;
                const uint32_t w0 = a.packed[gs >> 4], w1 = a.packed[(gs >> 4) + 1];
                const uint32_t six = (uint32_t)(((uint64_t)w1 << 32 | w0) >> (2u * (unsigned)(gs & 15))) & 0xfffu;
                const uint32_t lo6 = six & 63u, hi6 = six >> 6;
                // codon index as the walks form it: code(step) | code(step + 1) << 2 | code(step + 2) << 4
                const uint32_t idx = fwd ? (hi6 & 3u) << 4 | (hi6 & 12u) | hi6 >> 4 : lo6 ^ 63u;
                const uint32_t nidx = fwd ? (lo6 & 3u) << 4 | (lo6 & 12u) | lo6 >> 4 : hi6 ^ 63u;
                bn = s_which[idx] < 0 && !((a.fwd_stop >> nidx) & 1ull);
            }
            const uint64_t mn = __ballot(bn), mq = mn & ~low3;
            const uint32_t vn = mg_run_len(mn, lane, cn), vq = mg_run_len(mq, lane, cq);
            if (t < n) { on[t] = (uint8_t)vn; oq[t] = (uint8_t)vq; }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                cn[k] = (uint32_t)__builtin_amdgcn_readlane((int)vn, k);
                cq[k] = (uint32_t)__builtin_amdgcn_readlane((int)vq, k);
            }
        }
    }
}

// (the ORFs' aggregates are initialised by the level-0 count pass as it takes them -- it has the record in hand; the slot
// counters and the bitmap of the accepted ORFs are zeroed by memsets)
__global__ __launch_bounds__(256) void k_mg_err_prepare(MgArgs a, const uint64_t fit_len)
{
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.n_reads; r += (uint64_t)gridDim.x * blockDim.x)
        a.read_fit[r] = a.read_off[r + 1] - a.read_off[r] < fit_len;
}

#ifndef MG_CALL_CHUNK
#define MG_CALL_CHUNK 256
#endif
// calls a wave works through before it asks for more (count pass per 1M reads: 64: 79 ms, 128: 62, 256: 59, 512: 61, 2048: 71,
// 8192: 84 -- small tiles keep a wave's lanes on neighbouring reads; the write pass skips most calls and wants fewer atomics: 8 ms at 2048, 12 at 256)
#ifndef MG_COUNT_TILE
#define MG_COUNT_TILE 512        // (count passes, 32-byte calls: 128: 56.9 ms per 1M reads with -i, 256: 51.3, 512: 50.6, 768: 50.6, 1024: 51.6, 2048: 52.3)
#endif
#define MG_LEVEL_TILE (WRITE ? 2048 : MG_COUNT_TILE)
#ifndef MG_LEVEL_BATCH
#define MG_LEVEL_BATCH 32        // lanes that wait before the wave runs the take / finish code (count pass per 1M reads: 4: 64 ms, 8: 61, 16: 60, 24 - 48: 58)
#endif

#ifndef MG_LEVEL_WAVES
#define MG_LEVEL_WAVES 4         // waves per SIMD the level kernels are compiled for: 128 VGPRs.  Level 1 wants 131 (3 waves:
                                 // 214 ms per 1M reads instead of 190); asking for 5 / 6 / 8 waves spills inside the walk: 457 / 688 / 1202 ms
#endif

// PFX: a.walk holds the running sums of k_mg_walk_prefix: score[j] = I[w + j] - I[w - 1] instead of a sum of its own, and a walk
// jumps over the codons at which nothing can happen (a.run_q / a.run_n) -- it visits its start codons, its low-quality bases and its
// last codon only.
template <bool WRITE, int LEVEL, bool PFX>
__global__ __launch_bounds__(256, MG_LEVEL_WAVES) void k_mg_err_level(MgArgs a, const int accepted_only)
{
    __shared__ int8_t s_which[64];
    __shared__ double s_pen[64];
    if (threadIdx.x < 64) { s_which[threadIdx.x] = a.which[threadIdx.x]; s_pen[threadIdx.x] = a.err_mode == 1 ? a.pen[threadIdx.x] : 0.0; }
    __syncthreads();
    const bool pen_lds = a.indel_q_thr < 64;
    const int lane = threadIdx.x & 63;
    const int mgl = a.min_gene_len;
    const int lowest_j = mgl - 3 < 3 ? mgl - 3 : 3;
    uint64_t n_in = LEVEL == 0 ? a.n_orfs : (uint64_t)a.n_calls[LEVEL - 1];
    if (LEVEL > 0 && n_in > a.call_cap) n_in = a.call_cap;
    uint64_t chunk_base = 0;                            // the wave's chunk of the next level's array (uniform over the wave)
    uint32_t chunk_used = MG_CALL_CHUNK;                // none yet
    // A wave owns MG_LEVEL_TILE consecutive calls at a time and its lanes take them one by one: a lane that has finished its
    // call starts the next one of the tile instead of waiting for the longest call of the wave (calls run from 0 to ~500
    // positions; with one call per lane per round 60 % of the lane-trips were idle).
    // Tiles are handed out by a counter (one atomic per tile): every wave stays busy until the calls run out, whatever its tiles cost.
    unsigned long long *tile_ctr = a.tile_ctr + (WRITE ? 3 : 0) + LEVEL;
    for (;;) {
        unsigned long long t_ = 0;
        if (lane == 0) t_ = atomicAdd(tile_ctr, 1ull);
        const uint64_t tile = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)t_) |
                               (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(t_ >> 32)) << 32) * MG_LEVEL_TILE;
        if (tile >= n_in) break;
        uint64_t next = tile;
        const uint64_t tile_end = tile + MG_LEVEL_TILE < n_in ? tile + MG_LEVEL_TILE : n_in;
        // the call a lane is walking
        uint32_t orf = 0;
        int end_point = 0, suffix_j = 0;
        double suffix_score = 0.0;
        uint64_t key = 0;
        uint32_t e0 = 0, e1 = 0;
        bool fwd = false;
        int64_t dir = 1, g = 0, off = 0;
        int avail = 0, n = 0;
        uint32_t comp = 0;
        const double *wp = a.walk;                      // the call's stream of Frame_Scores, four doubles at a time
        const uint8_t *qp = a.qual;
        const uint8_t *rp = a.run_n;                    // PFX: the call's stream of run lengths
        double p0 = 0.0;                                // PFX: the running sum in front of the call's first position
        uint32_t nskip = 0;                             // PFX: the run length at the codon the next trip starts with
        double s0 = 0.0, s1 = 0.0;                      // score[] inside the codon being walked
        uint32_t qw = 0;
        bool walking = false, finishing = false, is_last = false, trunc = false, first_done = false;
        int tp = 0, br = 0;                             // codon, next branch candidate of it (0..5; 6 none)
        uint32_t pidx = 0, nidx = 0, last_own = MG_NO_SLOT, cnt = 0;
        int last_pos = 0, last_j = 0;
        double sum = 0.0, prev = 0.0, best = -DBL_MAX;

        auto fetch = [&](int t, uint32_t &idx) __attribute__((always_inline)) -> bool {
            if (avail - 3 * t < 3) { trunc = a.allow_truncated != 0; return true; }
            // the three bases of codon t lie in at most two packed words: both are loaded, no branches
            const int64_t g0c = g, g2c = g + 2 * dir;
            const uint32_t wa = a.packed[g0c >> 4], wb = a.packed[g2c >> 4];
            const int64_t g1c = g + dir;
            const uint32_t c0 = ((wa >> (2u * (unsigned)(g0c & 15))) & 3u) ^ comp;
            const uint32_t c1 = ((((g1c >> 4) == (g0c >> 4) ? wa : wb) >> (2u * (unsigned)(g1c & 15))) & 3u) ^ comp;
            const uint32_t c2 = ((wb >> (2u * (unsigned)(g2c & 15))) & 3u) ^ comp;
            g += 3 * dir;
            idx = c2 << 4 | c1 << 2 | c0;
            return (a.fwd_stop >> idx) & 1;
        };
        auto emit = [&](double raw, int j_loc, int pos, int which, int truncated, int first, uint32_t kind) __attribute__((always_inline)) -> uint32_t {
            const int j_full = j_loc + 2 + suffix_j;
            const int isl = a.read_isl ? a.read_isl[a.orfs[orf].read] : a.ignore_score_len;      // (a start is rare: no lane state for it)
            const double sc = (j_full > isl && 0.0 > raw) ? 0.0 : raw;
            uint32_t slot = MG_NO_SLOT;
            if (WRITE) {
                slot = (uint32_t)a.start_off[orf] + atomicAdd(&a.fill[orf], 1u);
                gmg_start s1;
                s1.score = sc; s1.j = j_full; s1.pos = pos; s1.which = which; s1.truncated = (int16_t)truncated; s1.first = (int16_t)first;
                a.starts[slot] = s1;
                gmg_start_errors er;
                er.pos[0] = LEVEL > 0 ? (int)(e0 >> 2) - 8 : 0; er.pos[1] = LEVEL > 1 ? (int)(e1 >> 2) - 8 : 0;
                er.type[0] = (int8_t)(LEVEL > 0 ? (e0 & 3) : 0); er.type[1] = (int8_t)(LEVEL > 1 ? (e1 & 3) : 0);
                er.n = LEVEL; er.reserved = 0;
                a.errs[slot] = er;
                a.keys[slot] = key | (uint64_t)((uint32_t)(2047 - j_loc) << 2 | kind) << (26 - 13 * LEVEL);
            } else {
                // inside one call pos moves with j (forward: pos = end - 2 - j, reverse: end + 2 + j), so the call's entry at the
                // extreme pos is simply the last one it emits
                last_pos = pos; last_j = j_full;
                if (sc > best) best = sc;
                cnt++;
            }
            return slot;
        };

        for (;;) {
            // The trip of a walking lane is short; taking a new call and finishing one are long, and a wave executes whatever ANY of
            // its lanes needs.  So those two happen in batches: only when MG_LEVEL_BATCH lanes wait for them (or nobody walks).
            const uint64_t wm = __ballot(walking);
            const uint64_t fm = __ballot(finishing && !walking);
            const bool do_fin = fm && (__popcll(fm) >= MG_LEVEL_BATCH || !wm);
            const bool idle = !(walking || finishing);
            const uint64_t im = __ballot(idle);
            if (next < tile_end && (__popcll(im) >= MG_LEVEL_BATCH || !(wm | fm))) {   // idle lanes take the next calls of the tile
                const uint64_t left = tile_end - next;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(im >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)im, 0u));
                if (idle && rank < left) {
                    const uint64_t i = next + rank;
                    bool active = true;
                    suffix_j = 0; suffix_score = 0.0; key = 0; e0 = e1 = 0;
                    if (LEVEL == 0 && WRITE && accepted_only && !((a.acc_bits[i >> 5] >> (i & 31u)) & 1u)) {
                        active = false;                 // (not accepted, or a read of k_mg_err_flat: the record is not even loaded)
                    } else if (LEVEL == 0) {
                        const gmg_mg_orf rec = a.orfs[i];
                        orf = (uint32_t)i; end_point = rec.frame > 0 ? rec.stop_position - 1 : rec.stop_position + 3;
                        fwd = rec.frame > 0;
                        if (!WRITE) {                   // what the calls of this ORF will be merged into (every ORF is taken by exactly one lane)
                            MgOrfAgg g0;
                            g0.best = mg_ord(-DBL_MAX); g0.ext_a = g0.ext_b = fwd ? ~0ull : 0ull; g0.cnt = 0; g0.m0 = 0;
                            a.agg[i] = g0;
                        }
                        if (!a.read_fit[rec.read]) active = false;              // k_mg_err_flat has the read
                        if (WRITE && accepted_only && !rec.accepted) active = false;
                        off = (int64_t)a.read_off[rec.read];
                        n = (int)(a.read_off[rec.read + 1] - a.read_off[rec.read]);
                        // Only accepted ORFs leave the GPU: an ORF is accepted only if one of its starts has j + 1 >= Min_Gene_Len
                        // (glimmer-mg.cc:1655-1668), and no path can get further from the ORF's end than the read reaches that way (j counts
                        // the positions walked, an insertion per level adds one, the entry itself two): such ORFs are not expanded at all.
                        if (!WRITE && accepted_only && (fwd ? end_point : n - end_point + 1) + 12 < a.min_gene_len) active = false;
                    } else {
                        const MgCall c = a.calls[LEVEL - 1][i];
                        if ((c.w[2] >> 51) == 0) active = false;                // the unused end of a wave's chunk
                        else {
                            suffix_score = __longlong_as_double((long long)c.w[0]);
                            key = c.w[1] & 0xffffffffffull; end_point = (int)((c.w[1] >> 40) & 0xfffu) - 8; suffix_j = (int)(c.w[1] >> 52);
                            off = (int64_t)(c.w[2] & 0xffffffffffull); n = (int)((c.w[2] >> 40) & 0x7ffu); fwd = (c.w[2] >> 53) & 1;
                            orf = (uint32_t)c.w[3]; e0 = (uint32_t)(c.w[3] >> 32) & 0x3fffu; e1 = (uint32_t)(c.w[3] >> 46) & 0x3fffu;
                            if (WRITE && accepted_only && !((a.acc_bits[orf >> 5] >> (orf & 31u)) & 1u)) active = false;
                        }
                    }
                    if (active) {
                        dir = fwd ? -1 : 1; comp = fwd ? 0u : 3u;
                        const int anchor = end_point - 1;
                        avail = fwd ? anchor + 1 : n - anchor;
                        g = off + anchor;
                        if (anchor >= 0 && anchor < n) {
                            const uint64_t ga = (uint64_t)g;
                            const uint64_t w = fwd ? a.total - 1 - ga : ga;
                            wp = a.qonly ? a.walk + (uint64_t)(fwd ? 0 : 1) * a.walk_stride + w
                                         : a.walk + (uint64_t)((fwd ? 0 : 3) + (int)(ga % 3)) * a.walk_stride + w;
                            qp = fwd ? a.walk_q + w : a.qual + ga;
                            if (PFX) {
                                const bool with_q = LEVEL < 2 && !WRITE && a.err_mode == 1 && LEVEL < a.indel_max;     // (branches can start here)
                                rp = (with_q ? a.run_q : a.run_n) + (fwd ? 0 : a.walk_stride) + w;
                                // (the sums restart with every read; qonly: the entry at the anchor IS the sum in front of it)
                                p0 = a.qonly ? wp[0] : (fwd ? anchor == n - 1 : anchor == 0) ? 0.0 : wp[-1];
                            }
                        }
                        is_last = false; trunc = false; first_done = false; walking = false;
                        tp = 0; br = 0; last_own = MG_NO_SLOT; cnt = 0;
                        sum = 0.0; prev = 0.0; best = -DBL_MAX;
                        if (anchor >= 0 && anchor < n) walking = !fetch(0, pidx);
                        if (PFX && walking) nskip = rp[0];
                        finishing = true;               // the end-of-call work is still to do
                    }
                }
                const uint32_t n_idle = __popcll(im);
                next += n_idle < left ? n_idle : left;
            }
            if (!__ballot(walking || finishing)) {
                if (next >= tile_end) break;
                continue;                               // (the calls just taken were all empty entries)
            }
            {
            bool want_push = false;                     // a branch met in this trip: end point, suffix, error entry, key field
            int c_end = 0, c_sj = 0;
            uint32_t c_err = 0, c_field = 0;
            double c_score = 0.0;
            if (walking) {
                // one in-frame codon (three buffer positions) per trip: the lanes of a wave stay in the same phase of the codon
                if (PFX && br == 0) tp += (int)nskip;  // codons from the one due on at which nothing happens: on to the one behind them
                const int j0 = 3 * tp;
                if (br == 0) {                          // first visit of the codon
                    if (PFX) {
                        // the codon the trip works on (it is there and is no stop codon: the codon before it was not the last) and the one
                        // behind it: everything the trip reads is requested before any of it is used -- ONE memory latency per trip behind
                        // the run length, which was requested a trip ago
                        const int64_t ga_ = g + (3 * (int64_t)nskip - 3) * dir, gb_ = ga_ + 3 * dir;
                        const uint32_t wa0 = a.packed[ga_ >> 4], wa1 = a.packed[(ga_ + 2 * dir) >> 4];
                        const uint32_t wb0 = a.packed[gb_ >> 4], wb1 = a.packed[(gb_ + 2 * dir) >> 4];      // (guard words around the batch)
                        MgD4 d;
                        if (a.qonly) { d.v[0] = wp[j0]; d.v[1] = d.v[2] = d.v[3] = 0.0; }    // score[j0 - 1] = Q[j0] - Q[0]; nothing else is asked for
                        else d = *(const MgD4 *)(wp + j0 - 1);                      // (8 spare entries on both sides of the tables)
                        if (LEVEL < 2 && !WRITE && a.err_mode == 1) qw = ((const MgU4 *)(qp + j0))->v;
                        auto codon = [&](int64_t gx, uint32_t w0, uint32_t w1) __attribute__((always_inline)) {
                            const int64_t g1 = gx + dir, g2 = gx + 2 * dir;
                            const uint32_t c0 = ((w0 >> (2u * (unsigned)(gx & 15))) & 3u) ^ comp;
                            const uint32_t c1 = ((((g1 >> 4) == (gx >> 4) ? w0 : w1) >> (2u * (unsigned)(g1 & 15))) & 3u) ^ comp;
                            const uint32_t c2 = ((w1 >> (2u * (unsigned)(g2 & 15))) & 3u) ^ comp;
                            return c2 << 4 | c1 << 2 | c0;
                        };
                        pidx = codon(ga_, wa0, wa1);
                        if (avail - 3 * (tp + 1) < 3) { trunc = a.allow_truncated != 0; is_last = true; }
                        else { nidx = codon(gb_, wb0, wb1); is_last = (a.fwd_stop >> nidx) & 1; }
                        g = gb_ + 3 * dir;
                        prev = j0 ? d.v[0] - p0 : 0.0;  // score[j0 - 1]
                        s0 = d.v[1] - p0; s1 = d.v[2] - p0; sum = d.v[3] - p0;     // score[j0], [j0 + 1], [j0 + 2]
                    } else {
                        is_last = fetch(tp + 1, nidx);  // is it the last of the region?
                        if (LEVEL < 2 && !WRITE && a.err_mode == 1) qw = ((const MgU4 *)(qp + j0))->v;
                        const MgD3 d = *(const MgD3 *)(wp + j0);   // (the tables end in 8 spare entries)
                        prev = sum;                     // score[j0 - 1]
                        s0 = prev + d.v[0]; s1 = s0 + d.v[1]; sum = s1 + d.v[2];   // score[j0], [j0 + 1], [j0 + 2]
                    }
                    if (j0 >= lowest_j && j0 + 3 + suffix_j >= mgl) {
                        const int k = fwd ? end_point - 2 - j0 : end_point + 2 + j0;
                        const int which = s_which[pidx];
                        const double raw = (prev - 0.0) + suffix_score;
                        if (which >= 0) last_own = emit(raw, j0, k, which, 0, 0, 3u);
                        if (is_last && trunc) { emit(raw, j0, k, -1, 1, 1, 2u); first_done = true; }
                    }
                }
                if (LEVEL < 2 && !WRITE && a.err_mode == 1 && LEVEL < a.indel_max) {
                    // Score_Indels at the three positions, in reversed push order: per position insertion, then deletion -- six
                    // candidates.  All six are evaluated without branches (most fail on the quality alone, but a wave pays for the
                    // longest path of any lane: the early-exit loop cost more), then the first one at or after br is taken.
                    uint32_t pass = 0;
                    double es6[6];
#pragma unroll
                    for (int pj = 0; pj < 3; pj++) {
                        const int q = (int)((qw >> (8 * pj)) & 255u);
                        const bool low = j0 + pj >= lowest_j && q <= a.indel_q_thr;
                        const double pen = pen_lds ? s_pen[q & 63] : a.pen[q];
                        const double before = pj == 0 ? prev : pj == 1 ? s0 : s1, at = pj == 0 ? s0 : pj == 1 ? s1 : sum;
                        es6[2 * pj] = ((suffix_score + before) - 0.0) + pen;
                        es6[2 * pj + 1] = ((suffix_score + at) - 0.0) + pen;
                        if (low && es6[2 * pj] > a.indel_suffix_thr) pass |= 1u << (2 * pj);
                        if (low && es6[2 * pj + 1] > a.indel_suffix_thr) pass |= 2u << (2 * pj);
                    }
                    pass &= ~((1u << br) - 1u);
                    if (pass) {
                        const int c = __ffs((int)pass) - 1, pj = c >> 1, b = c & 1, j = j0 + pj;
                        br = c + 1;
                        const int k = fwd ? end_point - 2 - j : end_point + 2 + j;
                        int epos;
                        if (b == 0) { c_end = fwd ? k - (2 - pj) : k + 2 - pj; epos = fwd ? k + 2 : k - 2; }
                        else { c_end = fwd ? k + pj : k - pj; epos = fwd ? k + 3 : k - 1; }
                        c_score = c == 0 ? es6[0] : c == 1 ? es6[1] : c == 2 ? es6[2] : c == 3 ? es6[3] : c == 4 ? es6[4] : es6[5];
                        c_sj = suffix_j + j + 2 - pj;
                        c_err = (uint32_t)(epos + 8) << 2 | (uint32_t)b;
                        c_field = (uint32_t)(2047 - j) << 2 | (b == 0 ? 1u : 0u);
                        want_push = true;
                    }
                }
                if (!want_push) {
                    br = 0;
                    if (is_last) walking = false;
                    else pidx = nidx;
                    tp++;
                    if (PFX && !is_last) nskip = rp[3 * tp];        // (requested a trip ahead of its use)
                }
            } else if (finishing && do_fin) {
                finishing = false;
                const int m = 3 * tp;
                if (LEVEL == 0 && !WRITE) {
                    a.agg[orf].m0 = (uint32_t)m << 1 | (trunc ? 1u : 0u);
                    if (a.err_mode == 2) {              // the substitution branch (:1771-1806)
                        const int lo = fwd ? end_point - m : end_point, hi = fwd ? end_point : end_point + m;
                        const int eep = fwd ? lo - 3 : hi + 3;
                        const int anchor = end_point - 1;
                        if (anchor >= 0 && anchor < n && eep >= 0 && eep - 2 < n) {
                            auto base = [&](int x) { const int64_t y = off + x; return (a.packed[y >> 4] >> (2u * (unsigned)(y & 15))) & 3u; };
                            const uint32_t want = fwd ? 0u : 3u;
                            const int a1 = base(fwd ? lo - 2 : hi) == want, a2 = base(fwd ? lo - 1 : hi - 1) == want;
                            double es = suffix_score + a.pass_stop[a1 * 2 + a2];
                            if (m > 0) es += (a.qonly ? wp[m] - p0 : sum) - 0.0;       // (qonly: score[m - 1] = Q[m] - Q[0]; the stop codon behind the region is inside the read)
                            c_end = eep; c_score = es; c_sj = suffix_j + m;
                            c_err = (uint32_t)((fwd ? lo - 2 : hi + 2) + 8) << 2 | 2u;
                            c_field = 0;                    // before every position of the call
                            want_push = true;
                        }
                    }
                }
                if (WRITE) { if (!first_done && last_own != MG_NO_SLOT) a.starts[last_own].first = 1; }
                else if (cnt) {
                    MgOrfAgg *g2 = a.agg + orf;
                    atomicAdd(&g2->cnt, cnt);
                    atomicMax(&g2->best, (unsigned long long)mg_ord(best));
                    const unsigned long long pa = (unsigned long long)(uint32_t)(last_pos + 16) << 32 | (uint32_t)last_j,
                                             pb = (unsigned long long)(uint32_t)(last_pos + 16) << 32 | (0xffffffffu - (uint32_t)last_j);
                    if (fwd) { atomicMin(&g2->ext_a, pa); atomicMin(&g2->ext_b, pb); }
                    else { atomicMax(&g2->ext_a, pa); atomicMax(&g2->ext_b, pb); }
                }
            }
            // a call that cannot reach Min_Gene_Len before its read ends emits nothing (a start needs j + 3 + suffix_j >= Min_Gene_Len),
            // nor can a branch of it: it is not handed on
            if (LEVEL < 2 && !WRITE && want_push && c_sj + (fwd ? c_end : n - c_end + 1) + 12 < mgl) want_push = false;
            if (LEVEL < 2 && !WRITE) {
                // the branches of this trip go to the next level's array.  The wave owns a chunk of MG_CALL_CHUNK entries at a time
                // (ONE atomic on the shared counter per chunk -- one per trip made the counter the bottleneck: 73 -> 9 ms); what is
                // left of a chunk when it is given up is marked empty (level 0) and skipped by the next launch
                const uint64_t pm = __ballot(want_push);
                if (pm) {
                    const uint32_t np = __popcll(pm);
                    if (chunk_used + np > MG_CALL_CHUNK) {
                        for (uint32_t x = chunk_used + lane; x < MG_CALL_CHUNK; x += 64)
                            if (chunk_base + x < a.call_cap) a.calls[LEVEL][chunk_base + x].w[2] = 0;
                        unsigned long long base = 0;
                        if (lane == 0) base = atomicAdd(&a.n_calls[LEVEL], (unsigned long long)MG_CALL_CHUNK);
                        chunk_base = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base) |
                                     (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32;
                        chunk_used = 0;
                        if (chunk_base + MG_CALL_CHUNK > a.call_cap && lane == 0) atomicOr(a.err_flag, 1u);
                    }
                    if (want_push) {
                        const uint64_t slot = chunk_base + chunk_used + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                        if (slot < a.call_cap) {
                            MgCall child;
                            const uint32_t ce0 = LEVEL == 0 ? c_err : e0, ce1 = LEVEL == 1 ? c_err : 0u;
                            child.w[0] = (unsigned long long)__double_as_longlong(c_score);
                            child.w[1] = (key | (uint64_t)c_field << (26 - 13 * LEVEL)) | (uint64_t)(uint32_t)(c_end + 8) << 40 | (uint64_t)(uint32_t)c_sj << 52;
                            child.w[2] = (uint64_t)off | (uint64_t)(uint32_t)n << 40 | (uint64_t)((uint32_t)(LEVEL + 1) | (fwd ? 4u : 0u)) << 51;
                            child.w[3] = (uint64_t)orf | (uint64_t)ce0 << 32 | (uint64_t)ce1 << 46;
                            a.calls[LEVEL][slot] = child;
                        }
                    }
                    chunk_used += np;
                }
            }
            }
        }
    }
    if (LEVEL < 2 && !WRITE)                            // give the rest of the last chunk back as empty entries
        for (uint32_t x = chunk_used + lane; x < MG_CALL_CHUNK; x += 64)
            if (chunk_base + x < a.call_cap) a.calls[LEVEL][chunk_base + x].w[2] = 0;
}

// Score_Orfs_Errors' verdict per ORF (:1647-1683) from what its calls added up to
// accepted_only: only the accepted ORFs' records are ever read again (everything downstream asks the bitmap first), so an ORF
// without starts (more than half of them) is settled from its 32-byte aggregate alone and a rejected one is not written back.
// all_fit: no read is left to k_mg_err_flat (the usual case), so nothing here needs the record's read.
__global__ __launch_bounds__(256) void k_mg_err_verdict(MgArgs a, const int accepted_only, const int all_fit)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n_orfs; i += (uint64_t)gridDim.x * blockDim.x) {
        const MgOrfAgg g = a.agg[i];
        if (accepted_only && all_fit && g.cnt == 0) { a.orf_cnt[i] = 0; continue; }
        gmg_mg_orf rec = a.orfs[i];
        if (!all_fit && !a.read_fit[rec.read]) continue;
        const bool f = rec.frame > 0;
        const int m0 = (int)(g.m0 >> 1);
        if (f) { rec.hi = rec.stop_position - 1; rec.lo = rec.hi - m0; }
        else { rec.lo = rec.stop_position + 3; rec.hi = rec.lo + m0; }
        rec.orf_is_truncated = (int16_t)(g.m0 & 1);
        rec.n_starts = g.cnt;
        rec.first_j = 0; rec.best_score = -DBL_MAX; rec.accepted = 0;
        if (g.cnt) {
            const uint32_t ja = (uint32_t)g.ext_a, jb = 0xffffffffu - (uint32_t)g.ext_b;
            const int jmin = (int)(f ? ja : jb), jmax = (int)(f ? jb : ja);
            rec.first_j = jmin;
            if (jmax + 1 >= a.min_gene_len) {
                rec.best_score = mg_unord(g.best);
                if (rec.best_score > a.start_threshold) rec.accepted = jmin + 1 >= a.min_gene_len ? 1 : 2;
            }
        }
        a.orf_cnt[i] = (accepted_only && !rec.accepted) ? 0u : g.cnt;
        if (rec.accepted) atomicOr(&a.acc_bits[i >> 5], 1u << (i & 31u));      // (one ORF in a hundred)
        if (accepted_only && !rec.accepted) continue;
        rec.start_begin = 0;
        a.orfs[i] = rec;
    }
}

// (after the scan) where each ORF's slice begins
// (accepted_only: the other ORFs' records never leave the GPU -- the bitmap of the accepted ones, set for the reads the level
// kernels take, saves touching 99 % of the 56-byte records)
__global__ __launch_bounds__(256) void k_mg_err_begin(MgArgs a, const int accepted_only)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n_orfs; i += (uint64_t)gridDim.x * blockDim.x) {
        if (accepted_only ? !((a.acc_bits[i >> 5] >> (i & 31u)) & 1u) : !a.read_fit[a.orfs[i].read]) continue;
        a.orfs[i].start_begin = (uint32_t)a.start_off[i];
    }
}

#include "gmg_scan.h"
#include "gmg_mg_errtile.h"
#include "gmg_mg_errwave.h"
#include "gmg_mg_orfbits.h"

// Every ORF's slice of the start array into the reference's push order: ascending order key, ties by place (the level kernels and
// the wave kernels hand the starts out in whatever order their lanes came by).  Slices are short (2 - 3 starts per accepted ORF with
// -s, ~24 with -i, a few of several hundred), so a start's place is COUNTED, not sorted: the number of starts of its ORF with a
// smaller (key, place).  k_mg_order_starts: a wave per ORF -- up to 64 starts: the keys sit in the lanes, one pass of readlanes; more:
// the keys in the wave's LDS share 512 at a time, up to eight starts per lane and round.  Starts and Error_t lists move to their
// places in the same pass.  (Replaces a library segmented sort of (key, index) pairs + three helper launches.)
#define MG_ORDER_MID 512
__global__ __launch_bounds__(256) void k_mg_order_starts(const gmg_mg_orf *orfs, const uint64_t n_orfs, const uint64_t *keys, const gmg_start *s_in,
                                                         const gmg_start_errors *e_in, gmg_start *s_out, gmg_start_errors *e_out)
{
    __shared__ uint64_t s_keys[4][MG_ORDER_MID];
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t *sk = s_keys[threadIdx.x >> 6];
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t i = wave; i < n_orfs; i += n_waves) {
        const uint32_t b = orfs[i].start_begin, n = orfs[i].n_starts;
        if (n == 0) continue;
        if (n <= 64) {                                  // the keys sit in the lanes
            const uint64_t key = lane < n ? keys[b + lane] : ~0ull;
            // (a start = three 64-bit words, its Error_t list three 32-bit ones: as words they stay in registers -- the structs went through scratch)
            uint64_t sw[3] = {0, 0, 0};
            uint32_t ew[3] = {0, 0, 0};
            static_assert(sizeof(gmg_start) == 24 && sizeof(gmg_start_errors) == 12, "k_mg_order_starts moves them as words");
            if (lane < n) {
                const uint64_t *sp = (const uint64_t *)(const void *)(s_in + b + lane);
                const uint32_t *ep = (const uint32_t *)(const void *)(e_in + b + lane);
#pragma unroll
                for (int w = 0; w < 3; w++) { sw[w] = sp[w]; ew[w] = ep[w]; }
            }
            uint32_t rank = 0;
            for (uint32_t j = 0; j < n; j++) {
                const uint64_t kj = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, (int)j) |
                                    (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), (int)j) << 32;
                rank += (kj < key || (kj == key && j < lane)) ? 1u : 0u;
            }
            if (lane < n) {
                uint64_t *sp = (uint64_t *)(void *)(s_out + b + rank);
                uint32_t *ep = (uint32_t *)(void *)(e_out + b + rank);
#pragma unroll
                for (int w = 0; w < 3; w++) { sp[w] = sw[w]; ep[w] = ew[w]; }
            }
            continue;
        }
        // E starts per lane and round (start m0 + lane + 64 e), every key read once per round from LDS by all lanes
        auto mid = [&](auto E_) __attribute__((always_inline)) {
            constexpr int E = decltype(E_)::value;
            for (uint32_t m0 = 0; m0 < n; m0 += 64u * E) {
                uint64_t key[E];
                uint32_t rank[E];
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const uint32_t m = m0 + lane + 64u * (uint32_t)e;
                    key[e] = m < n ? keys[b + m] : ~0ull;
                    rank[e] = 0;
                }
                for (uint32_t c0 = 0; c0 < n; c0 += MG_ORDER_MID) {
                    const uint32_t cn = n - c0 < MG_ORDER_MID ? n - c0 : MG_ORDER_MID;
                    wcs_sync();                         // (the lanes are through with the keys before)
                    for (uint32_t t = lane; t < cn; t += 64) sk[t] = keys[b + c0 + t];
                    wcs_sync();
#pragma unroll 4
                    for (uint32_t t = 0; t < cn; t++) {
                        const uint64_t kj = sk[t];
#pragma unroll
                        for (int e = 0; e < E; e++) rank[e] += (kj < key[e] || (kj == key[e] && c0 + t < m0 + lane + 64u * (uint32_t)e)) ? 1u : 0u;
                    }
                }
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const uint32_t m = m0 + lane + 64u * (uint32_t)e;
                    if (m < n) { s_out[b + rank[e]] = s_in[b + m]; e_out[b + rank[e]] = e_in[b + m]; }
                }
            }
        };
        if (n <= 128) mid(std::integral_constant<int, 2>());
        else if (n <= 256) mid(std::integral_constant<int, 4>());
        else mid(std::integral_constant<int, 8>());
    }
}

// bits (may be NULL): bit i = ORF i is accepted (the error branch's bitmap): asked before a record is touched
#define MG_KEPT(bits, orfs, i) ((bits) ? (((bits)[(i) >> 5] >> ((i) & 31u)) & 1u) != 0 : (orfs)[i].accepted != 0)
// GMG_MG_ACCEPTED_ONLY: keep the ORFs that go to Add_Events_* and their start lists, packed, in the same order
__global__ __launch_bounds__(256) void k_mg_keep_counts(const gmg_mg_orf *orfs, const uint32_t *bits, uint64_t n, uint32_t *keep, uint32_t *keep_starts)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const bool k = MG_KEPT(bits, orfs, i);
        keep[i] = k ? 1u : 0u;
        if (keep_starts) keep_starts[i] = k ? orfs[i].n_starts : 0u;
    }
}

__global__ __launch_bounds__(256) void k_mg_keep_gather(const gmg_mg_orf *orfs, const uint32_t *bits, const gmg_start *starts, uint64_t n,
                                                        const uint64_t *new_orf, const uint64_t *new_start,
                                                        gmg_mg_orf *out_orfs, gmg_start *out_starts)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t base = wave * 64; base < n; base += n_waves * 64) {
        const uint64_t i = base + lane;
        uint32_t b = 0, cnt = 0;
        uint64_t dst = 0;
        bool acc = false;
        if (i < n && MG_KEPT(bits, orfs, i)) {
            gmg_mg_orf o = orfs[i];
            if (new_start) {                            // (NULL: the start array is packed already -- the records alone move)
                acc = true; b = o.start_begin; cnt = o.n_starts; dst = new_start[i];
                o.start_begin = (uint32_t)dst;
            }
            out_orfs[new_orf[i]] = o;
        }
        for (uint64_t m = __ballot(acc); m; m &= m - 1) {
            const int src = __ffsll((long long)m) - 1;
            const uint32_t b_ = (uint32_t)__shfl((int)b, src), cnt_ = (uint32_t)__shfl((int)cnt, src);
            const uint64_t dst_ = (uint64_t)(uint32_t)__shfl((int)(uint32_t)dst, src) | (uint64_t)(uint32_t)__shfl((int)(uint32_t)(dst >> 32), src) << 32;
            for (uint32_t t = lane; t < cnt_; t += 64) out_starts[dst_ + t] = starts[b_ + t];
        }
    }
}

// The same with the accepted ORFs as a bitmap (error branch; the start lists are packed already): the scan runs over the bitmap's WORDS
// (a 32nd of the entries), a record's new place is its word's offset + the set bits below it
__global__ __launch_bounds__(256) void k_mg_keep_words(const uint32_t *bits, uint64_t n_words, uint32_t *cnt)
{
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w <= n_words; w += (uint64_t)gridDim.x * blockDim.x)
        cnt[w] = w < n_words ? (uint32_t)__popc(bits[w]) : 0u;
}
__global__ __launch_bounds__(256) void k_mg_keep_gather_bits(const gmg_mg_orf *orfs, const uint32_t *bits, const uint32_t *word_off, uint64_t n_words, uint64_t n,
                                                             gmg_mg_orf *out_orfs)
{
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t m = bits[w], dst = word_off[w];
        while (m) {
            const uint64_t i = 32 * w + (uint32_t)__ffs((int)m) - 1u;
            m &= m - 1u;
            if (i < n) out_orfs[dst++] = orfs[i];
        }
    }
}
__global__ __launch_bounds__(256) void k_mg_keep_reads_bits(const uint64_t *read_orf_off, uint64_t n_reads, const uint32_t *bits, const uint32_t *word_off, uint64_t *out)
{
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= n_reads; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = read_orf_off[r];             // (<= n_orfs: inside the bitmap's last word, whose unused bits are zero)
        out[r] = (uint64_t)word_off[i >> 5] + (uint32_t)__popc(bits[i >> 5] & ((1u << (i & 31u)) - 1u));
    }
}

__global__ __launch_bounds__(256) void k_mg_keep_reads(const uint64_t *read_orf_off, uint64_t n_reads, const uint64_t *new_orf, uint64_t *out)
{
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= n_reads; r += (uint64_t)gridDim.x * blockDim.x)
        out[r] = new_orf[read_orf_off[r]];              // new_orf has n_orfs + 1 entries: the last one is the total
}

// Frame_Scores with one null model per read from the complete fp32 gene rows: out[row][g] = (double) gene - (double) null
// (gmg_frame_score6_nulls; the table the error branch walks in classification mode).  One lane per base.
__global__ __launch_bounds__(256) void k_mg_apply_nulls(MgArgs a, double *out, uint64_t out_stride)
{
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < a.total; g += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t r = a.tile_read[g / GMG_TILE];
        uint64_t r_end = a.read_off[r + 1];
        while (g >= r_end) { r++; r_end = a.read_off[r + 1]; }
        const uint64_t off = a.read_off[r];
        const int n = (int)(r_end - off), si = (int)(g - off);
        const uint32_t ni = a.read_null ? a.read_null[r] : 0u;
        const float *nt = a.null_tab + (size_t)ni * MG_NULL_FLOATS;
        const uint64_t x = dev_window_bits(a.packed, (int64_t)g - 2);          // bases g-2 .. g+2 in the low ten bits
        const uint32_t five = (uint32_t)x & 0x3ffu, c0 = (five >> 4) & 3u;
#pragma unroll
        for (int f = 0; f < 3; f++) {
            const float nf = mg_null_value<true>(nt, f, si, n, c0, (five >> 6) & 3u, (five >> 8) & 3u);
            const float nr = mg_null_value<false>(nt, f, si, n, c0, (five >> 2) & 3u, five & 3u);
            out[(uint64_t)f * out_stride + g] = (double)a.gene32[(uint64_t)f * a.fs_stride + g] - (double)nf;
            out[(uint64_t)(3 + f) * out_stride + g] = (double)a.gene32[(uint64_t)(3 + f) * a.fs_stride + g] - (double)nr;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
static unsigned grid_for(uint64_t n);

// gmg_mg_score_groups: consecutive groups of reads, each under its own gene model (NULL / 0: one model for the batch)
struct MgGroups {
    std::vector<const gmg_model *> models;
    std::vector<uint64_t> read_begin;                   // n + 1 entries
    int n = 0;
};

// the gene model's fp32 rows, complete (gmg_launch_gene6_full), for one model or for groups of reads under their own models
static int mg_gene6_full(const gmg_model *gene, const MgGroups *groups, const gmg_reads *reads, float *d_gene32, uint64_t gstride, hipStream_t s)
{
    if (groups && groups->n > 0)
        return gmg_launch_gene6_groups(groups->models.data(), groups->read_begin.data(), groups->n, reads, d_gene32, gstride, s);
    const int rc = gmg_launch_gene6_full(gene, reads, d_gene32, gstride, s);
    if (rc != GMG_EBADMODEL || gene->dev.P < 3) return rc;
    const uint64_t whole[2] = {0, reads->n_reads};      // a model of another shape: the any-shape kernel, as one group
    return gmg_launch_gene6_groups(&gene, whole, 1, reads, d_gene32, gstride, s);
}

// the table of gmg_frame_score6 with read r scored against null model d_read_null[r] (device array, or NULL: model 0)
static int mg_frame6_nulls(const gmg_model *gene, const float *d_null_tab, const uint32_t *d_read_null,
                           const gmg_reads *reads, double *d_out, uint64_t stride, hipStream_t s, const MgGroups *groups = nullptr)
{
    if (reads->total_bases == 0) return GMG_OK;
    const uint64_t gstride = (reads->total_bases + 15) & ~15ull;
    float *d_gene32 = nullptr;
    GMG_HIP(gmg_pool_alloc((void **)&d_gene32, (size_t)6 * gstride * sizeof(float)));
    int rc = mg_gene6_full(gene, groups, reads, d_gene32, gstride, s);
    if (rc) { gmg_pool_release(d_gene32); return gmg_set_error(rc, "per-read null models need a gene model of the default shape (depth 7, window <= 15, periodicity 3)"); }
    MgArgs a;
    memset(&a, 0, sizeof a);
    a.packed = reads->d_packed;
    a.read_off = reads->d_off;
    a.tile_read = reads->d_tile_read;
    a.n_reads = reads->n_reads;
    a.total = reads->total_bases;
    a.gene32 = d_gene32;
    a.fs_stride = gstride;
    a.null_tab = d_null_tab;
    a.read_null = d_read_null;
    hipLaunchKernelGGL(k_mg_apply_nulls, dim3(grid_for(a.total)), dim3(256), 0, s, a, d_out, stride);
    GMG_HIP(hipGetLastError());
    gmg_pool_release_after(d_gene32, s);
    return GMG_OK;
}

extern "C" int gmg_frame_score6_nulls(const gmg_model *gene, const gmg_null_set *nulls, const uint32_t *read_null,
                                      const gmg_reads *reads, double *d_out, uint64_t row_stride, void *stream)
{
    { int rc_enter = gmg_enter("gmg_frame_score6_nulls"); if (rc_enter) return rc_enter; }
    if (!gene || !nulls || !reads || (!read_null && reads->n_reads) || (!d_out && reads->total_bases))
        return gmg_set_error(GMG_EINVAL, "gmg_frame_score6_nulls: NULL argument");
    if (row_stride < reads->total_bases) return gmg_set_error(GMG_EINVAL, "gmg_frame_score6_nulls: row stride < total_bases");
    if (gene->dev.P < 3) return gmg_set_error(GMG_EBADMODEL, "gmg_frame_score6_nulls: periodicity must be >= 3");
    for (uint64_t r = 0; r < reads->n_reads; r++)
        if (read_null[r] >= (uint32_t)nulls->n)
            return gmg_set_error(GMG_ERANGE, "gmg_frame_score6_nulls: read %llu names null model %u of %d", (unsigned long long)r, read_null[r], nulls->n);
    if (reads->total_bases == 0) return GMG_OK;
    hipStream_t s = (hipStream_t)stream;
    uint32_t *d_rn = nullptr;
    GMG_HIP(gmg_pool_alloc((void **)&d_rn, reads->n_reads * 4));
    hipError_t e = hipMemcpyAsync(d_rn, read_null, reads->n_reads * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);              // (the caller's array may go away when the call returns)
    int rc = e == hipSuccess ? mg_frame6_nulls(gene, nulls->d_tab, d_rn, reads, d_out, row_stride, s)
                             : gmg_set_error(GMG_EHIP, "gmg_frame_score6_nulls: %s", hipGetErrorString(e));
    gmg_pool_release_after(d_rn, s);
    return rc;
}

static unsigned mg_codon_from(const char *s)            // Codon_t::Set_From (gene.cc:133-146)
{
    unsigned d = 0;
    for (int i = 0; i < 3 && s[i]; i++) d = ((d & 0xffu) << 4) | mg_ch_mask(s[i]);
    return d;
}
static unsigned mg_codon_revcomp(unsigned data)         // Codon_t::Reverse_Complement (gene.cc:96-113)
{
    unsigned x = 0;
    for (int i = 0; i < 12; i++) { x = (x << 1) | (data & 1u); data >>= 1; }
    return x;
}

// scratch and result buffers come from the library's cache of device blocks (gmg_pool_alloc, gmg_api.hip)
static unsigned grid_for(uint64_t n)
{
    const uint64_t blocks = (n + 255) / 256;
    return (unsigned)(blocks < 256 * 16 ? (blocks ? blocks : 1) : 256 * 16);
}

// exclusive sum of cnt[0..n] (cnt[n] = 0) into off[0..n] as 64-bit offsets; *total = off[n]
// d_off[i] = d_cnt[0] + ... + d_cnt[i-1], i <= n, and *total = d_off[n] (synchronises s): one launch (gmg_scan.h)
static int mg_scan(uint32_t *d_cnt, uint64_t *d_off, uint64_t n, uint64_t *total, hipStream_t s)
{
    hipError_t e = gmg_scan_excl<uint32_t, uint64_t>(d_cnt, d_off, n + 1, s);
    if (e == hipSuccess) e = hipMemcpyAsync(total, d_off + n, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return gmg_set_error(GMG_EHIP, "gmg_mg_score_reads: scan: %s", hipGetErrorString(e));
    return GMG_OK;
}

extern "C" int gmg_mg_result_free(gmg_mg_result *r)
{
    if (!r) return GMG_OK;
    void *ptrs[] = {r->d_orfs, r->d_starts, r->d_read_orf_off, r->d_errs};
    for (void *p : ptrs)
        if (p) gmg_pool_release(p);
    delete r;
    return GMG_OK;
}

// GMG_MG_TIMING=1: wall time of every stage on stderr (synchronises after each stage)
struct MgTimer {
    bool on;
    hipStream_t s;
    std::chrono::steady_clock::time_point t0;
    MgTimer(hipStream_t st) : on(gmg_opt(GMG_OPT_MG_TIMING) != 0), s(st), t0(std::chrono::steady_clock::now()) {}
    void lap(const char *what)
    {
        if (!on) return;
        (void)hipStreamSynchronize(s);
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[gmg_mg] %-28s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = std::chrono::steady_clock::now();
    }
};

#define MG_RETRY_NO_WAVE 1000    // (internal) the write pass of k_mg_err_wave ran out of stack: the call repeats without the wave kernels
static thread_local int tl_mg_no_wave = 0;
static int mg_run_once(const gmg_model *gene, const gmg_model *nul, const gmg_reads *reads, const gmg_mg_params *prm,
                       double *d_frame_scores, gmg_mg_result **out, void *stream, const bool find_only, const MgGroups *groups);
static int mg_run(const gmg_model *gene, const gmg_model *nul, const gmg_reads *reads, const gmg_mg_params *prm,
                  double *d_frame_scores, gmg_mg_result **out, void *stream, const bool find_only, const MgGroups *groups = nullptr)
{
    int rc = mg_run_once(gene, nul, reads, prm, d_frame_scores, out, stream, find_only, groups);
    if (rc == MG_RETRY_NO_WAVE) {
        tl_mg_no_wave = 1;
        rc = mg_run_once(gene, nul, reads, prm, d_frame_scores, out, stream, find_only, groups);
        tl_mg_no_wave = 0;
    }
    return rc;
}
static int mg_run_once(const gmg_model *gene, const gmg_model *nul, const gmg_reads *reads, const gmg_mg_params *prm,
                       double *d_frame_scores, gmg_mg_result **out, void *stream, const bool find_only, const MgGroups *groups)
{
    { int rc_enter = gmg_enter(find_only ? "gmg_find_orfs" : "gmg_mg_score_reads"); if (rc_enter) return rc_enter; }
    if ((!find_only && (!gene || !nul)) || !reads || !prm || !out) return gmg_set_error(GMG_EINVAL, "gmg_mg_score_reads: NULL argument");
    if (prm->n_start_codons < 0 || prm->n_start_codons > 8 || prm->n_stop_codons < 0 || prm->n_stop_codons > 8 ||
        prm->min_gene_len < 4)
        return gmg_set_error(GMG_EINVAL, "gmg_mg_score_reads: need 0..8 start / stop codons and min_gene_len >= 4");
    if (!find_only && (gene->dev.P != 3 || nul->dev.P != 3))
        return gmg_set_error(GMG_EBADMODEL, "gmg_mg_score_reads: Score_All_Frames needs models of periodicity 3");
    if (reads->n_reads >= 0x7fffffffull) return gmg_set_error(GMG_EINVAL, "gmg_mg_score_reads: batch too large");
    if (!find_only && prm->nulls) {                     // classification mode: one null model per read
        if (!prm->read_null && reads->n_reads) return gmg_set_error(GMG_EINVAL, "gmg_mg_score_reads: nulls without read_null");
        for (uint64_t r = 0; r < reads->n_reads; r++)
            if (prm->read_null[r] >= (uint32_t)prm->nulls->n)
                return gmg_set_error(GMG_ERANGE, "gmg_mg_score_reads: read %llu names null model %u of %d", (unsigned long long)r,
                                     prm->read_null[r], prm->nulls->n);
    } else if (!find_only && (prm->read_null || prm->read_ignore_score_len))
        return gmg_set_error(GMG_EINVAL, "gmg_mg_score_reads: read_null / read_ignore_score_len need a null set (gmg_mg_params.nulls)");
    const int err_mode = (prm->flags & GMG_MG_ALLOW_INDELS) ? 1 : (prm->flags & GMG_MG_ALLOW_SUBS) ? 2 : 0;
    if ((prm->flags & GMG_MG_ALLOW_INDELS) && (prm->flags & GMG_MG_ALLOW_SUBS))     // glimmer-mg.cc:952-955
        return gmg_set_error(GMG_EINVAL, "gmg_mg_score_reads: cannot use indels and substitutions simultaneously");
    if (err_mode && (prm->indel_max < 0 || prm->indel_max > 2 || prm->indel_quality_threshold < 0 || prm->indel_quality_threshold > 254))
        return gmg_set_error(GMG_EINVAL, "gmg_mg_score_reads: need indel_max in 0..2 and indel_quality_threshold in 0..254");
    // Find_Orfs' other two modes (ignore regions, circular sequences): gmg_find_orfs alone -- the start scan of the front half
    // indexes Frame_Scores inside one linear read
    const bool general = prm->circular != 0 || prm->n_ignore_regions != 0;
    if (general && !find_only)
        return gmg_set_error(GMG_EINVAL, "gmg_mg_score_reads: circular sequences / ignore regions are taken by gmg_find_orfs only");
    if (prm->n_ignore_regions < 0 || (prm->n_ignore_regions > 0 && (!prm->ignore_lo || !prm->ignore_hi)))
        return gmg_set_error(GMG_EINVAL, "gmg_find_orfs: n_ignore_regions without ignore_lo / ignore_hi");
    for (int k = 0; k < prm->n_ignore_regions; k++)     // as Get_Ignore_Regions leaves them: lo < hi, sorted, disjoint
        if (prm->ignore_lo[k] < 0 || prm->ignore_lo[k] >= prm->ignore_hi[k] || (k > 0 && prm->ignore_lo[k] < prm->ignore_hi[k - 1]))
            return gmg_set_error(GMG_EINVAL, "gmg_find_orfs: ignore region %d is not sorted / disjoint / lo < hi", k);
    hipStream_t s = (hipStream_t)stream;

    MgArgs a;
    memset(&a, 0, sizeof a);
    a.packed = reads->d_packed;
    a.read_off = reads->d_off;
    a.tile_read = reads->d_tile_read;
    a.n_reads = reads->n_reads;
    a.total = reads->total_bases;
    a.min_gene_len = prm->min_gene_len;
    a.allow_truncated = prm->allow_truncated;
    a.ignore_score_len = prm->ignore_score_len;
    a.start_threshold = prm->start_threshold;
    a.err_mode = err_mode;
    double pen_host[256];
    if (err_mode) {
        a.min_indel_orf_len = prm->min_indel_orf_len;
        a.indel_q_thr = prm->indel_quality_threshold;
        a.indel_max = prm->indel_max;
        a.indel_suffix_thr = prm->indel_suffix_score_threshold;
        for (int q = 0; q < 256; q++) {                 // Score_Indels (glimmer-mg.cc:1522-1523), the host's libm like the reference
            const double prob_err = pow(10.0, -(double)q / 10.0);
            pen_host[q] = log(prob_err / 2.0) - log(1.0 - prob_err);
        }
        for (int k = 0; k < 4; k++) {                   // Pass_Stop_Penalty (glimmer-mg.cc:961-995) without quality values
            const double default_p = 0.999;
            double p_stop = default_p;
            if (k & 2) p_stop *= 2.0 / 3.0 * default_p + 1.0 / 3.0; else p_stop *= default_p;
            if (k & 1) p_stop *= 2.0 / 3.0 * default_p + 1.0 / 3.0; else p_stop *= default_p;
            a.pass_stop[k] = log(1.0 - p_stop) - log(p_stop);
        }
    }
    {   // Set_Start_And_Stop_Codons (glimmer_base.cc:2683-2704) -> one bit / one byte per definite codon
        unsigned f_start[8], r_start[8], f_stop[8], r_stop[8];
        for (int p = 0; p < prm->n_start_codons; p++) { f_start[p] = mg_codon_from(prm->start_codon[p]); r_start[p] = mg_codon_revcomp(f_start[p]); }
        for (int p = 0; p < prm->n_stop_codons; p++) { f_stop[p] = mg_codon_from(prm->stop_codon[p]); r_stop[p] = mg_codon_revcomp(f_stop[p]); }
        for (unsigned idx = 0; idx < 64; idx++) {
            const unsigned data = (1u << ((idx >> 4) & 3)) << 8 | (1u << ((idx >> 2) & 3)) << 4 | (1u << (idx & 3));
            auto can_be = [&](const unsigned *pat, int np) {       // Codon_t::Can_Be (gene.cc:39-66)
                for (int p = 0; p < np; p++) { const unsigned x = data & pat[p]; if ((x & 0xf00) && (x & 0xf0) && (x & 0xf)) return p; }
                return -1;
            };
            auto must_be = [&](const unsigned *pat, int np) {      // Codon_t::Must_Be (gene.cc:70-92)
                for (int p = 0; p < np; p++) if ((data & pat[p]) == data) return true;
                return false;
            };
            a.which[idx] = (int8_t)can_be(f_start, prm->n_start_codons);
            if (a.which[idx] >= 0) a.fwd_start |= 1ull << idx;
            if (can_be(r_start, prm->n_start_codons) >= 0) a.rev_start |= 1ull << idx;
            if (must_be(f_stop, prm->n_stop_codons)) a.fwd_stop |= 1ull << idx;
            if (must_be(r_stop, prm->n_stop_codons)) a.rev_stop |= 1ull << idx;
        }
        for (unsigned v = 0; v < 64; v++) {
            const unsigned idx = (v & 3u) << 4 | (v & 12u) | v >> 4;
            if ((a.fwd_stop >> idx) & 1ull) a.fwd_stop_nat |= 1ull << v;
            if ((a.rev_stop >> idx) & 1ull) a.rev_stop_nat |= 1ull << v;
        }
    }

    gmg_mg_result *res = new (std::nothrow) gmg_mg_result();
    if (!res) return gmg_set_error(GMG_ENOMEM, "gmg_mg_score_reads: out of host memory");
    memset(res, 0, sizeof *res);
    res->n_reads = reads->n_reads;
    double *d_fs_own = nullptr;
    float *d_gene32 = nullptr;
    uint32_t *d_read_null = nullptr;
    int32_t *d_read_isl = nullptr;
    uint32_t *d_read_cnt = nullptr, *d_orf_cnt = nullptr;
    uint64_t *d_start_off = nullptr;
    double *d_cum = nullptr;
    bool fused_nc2 = false;
    int fused_nw = 0, fused_el = 9;                     // waves per tile of k_mg_tile_starts (0: the sequential kernels), elements per lane
    bool err_exact = false;                             // the batch's sums are exact in any order: the error branch may take differences of running sums
    bool err_tile = false;                              // ... and runs tile by tile with the sums in LDS (k_mg_err_tile)
    bool err_wave = false;                              // ... or with one wave per (read, strand), everything in the wave's LDS (k_mg_err_wave)
    uint32_t ew_cap = 0;
    uint8_t *d_item_flag = nullptr;
    uint8_t *d_run = nullptr;
    bool fused_rest = false;
    MgTile *d_tiles = nullptr, *d_all = nullptr;
    uint32_t *d_unfit = nullptr;
    uint32_t *d_ntiles = nullptr;
    void *d_sel_tmp = nullptr;
    uint8_t *d_qual = nullptr, *d_user_q = nullptr, *d_read_fit = nullptr;
    double *d_pen = nullptr;
    uint64_t *d_keys = nullptr;
    uint32_t *d_err_flag = nullptr, *d_fill = nullptr, *d_acc_bits = nullptr;
    MgCall *d_calls[2] = {nullptr, nullptr};
    MgOrfAgg *d_agg = nullptr;
    double *d_walk = nullptr;
    uint8_t *d_walk_q = nullptr;
    MgTile *d_et_tiles = nullptr;                       // the error branch tile by tile (k_mg_err_tile)
    MgCall *d_et_slabs = nullptr;
    EtEm *d_et_em = nullptr;
    gmg_start *d_st_s = nullptr;                        // ... its staging arrays: the kept ORFs' slices in the order the tiles finish
    gmg_start_errors *d_st_e = nullptr;
    uint64_t *d_st_k = nullptr;
    int rc = GMG_OK;
    auto fail = [&](int code) {
        (void)hipDeviceSynchronize();                   // nothing (either stream) may still use the blocks that go back to the cache
        if (d_fs_own) gmg_pool_release(d_fs_own);
        if (d_gene32) gmg_pool_release(d_gene32);
        if (d_read_null) gmg_pool_release(d_read_null);
        if (d_read_isl) gmg_pool_release(d_read_isl);
        if (d_read_cnt) gmg_pool_release(d_read_cnt);
        if (d_orf_cnt) gmg_pool_release(d_orf_cnt);
        if (d_start_off) gmg_pool_release(d_start_off);
        if (d_cum) gmg_pool_release(d_cum);
        if (d_tiles) gmg_pool_release(d_tiles);
        if (d_ntiles) gmg_pool_release(d_ntiles);
        if (d_all) gmg_pool_release(d_all);
        if (d_sel_tmp) gmg_pool_release(d_sel_tmp);
        if (d_unfit) gmg_pool_release(d_unfit);
        if (d_qual) gmg_pool_release(d_qual);
        if (d_user_q) gmg_pool_release(d_user_q);
        if (d_pen) gmg_pool_release(d_pen);
        if (d_read_fit) gmg_pool_release(d_read_fit);
        if (d_keys) gmg_pool_release(d_keys);
        if (d_err_flag) gmg_pool_release(d_err_flag);
        if (d_fill) gmg_pool_release(d_fill);
        if (d_acc_bits) gmg_pool_release(d_acc_bits);
        if (d_calls[0]) gmg_pool_release(d_calls[0]);
        if (d_calls[1]) gmg_pool_release(d_calls[1]);
        if (d_agg) gmg_pool_release(d_agg);
        if (d_walk) gmg_pool_release(d_walk);
        if (d_walk_q) gmg_pool_release(d_walk_q);
        if (d_run) gmg_pool_release(d_run);
        if (d_item_flag) gmg_pool_release(d_item_flag);
        if (d_et_tiles) gmg_pool_release(d_et_tiles);
        if (d_et_slabs) gmg_pool_release(d_et_slabs);
        if (d_et_em) gmg_pool_release(d_et_em);
        if (d_st_s) gmg_pool_release(d_st_s);
        if (d_st_e) gmg_pool_release(d_st_e);
        if (d_st_k) gmg_pool_release(d_st_k);
        gmg_mg_result_free(res);
        return code;
    };
#define MG_TRY(call)                                                                                            \
    do {                                                                                                        \
        hipError_t e_ = (call);                                                                                 \
        if (e_ != hipSuccess)                                                                                   \
            return fail(gmg_set_error(e_ == hipErrorOutOfMemory ? GMG_ENOMEM : GMG_EHIP, "gmg_mg_score_reads: %s: %s", \
                                      #call, hipGetErrorString(e_)));                                           \
    } while (0)

    MgTimer tm(s);
    // 1. Frame_Scores
    if (!find_only) {
    if (err_mode) {
        // the penalties go up FIRST: behind the six-frame kernels this 2 KB copy waited a millisecond for a free slot beside the
        // side streams' kernels, and the running sums behind it (profiles/r03_mgerr_timeline_indel.txt of the build before)
        MG_TRY(gmg_pool_alloc((void **)&d_pen, sizeof pen_host));
        MG_TRY(hipMemcpyAsync(d_pen, pen_host, sizeof pen_host, hipMemcpyHostToDevice, s));    // (pen_host lives until this call returns)
        a.pen = d_pen;
    }
    // classification mode: the per-read tables go to the device once per call (4 + 4 bytes per read)
    // (a (3,2,3) model's partial-window table follows its full-window table in the model blob: gmg_model_upload)
    const float *d_null_tab = nul->dev.dense;
    if (prm->nulls) {
        d_null_tab = prm->nulls->d_tab;
        if (a.n_reads) {
            MG_TRY(gmg_pool_alloc((void **)&d_read_null, a.n_reads * 4));
            MG_TRY(hipMemcpyAsync(d_read_null, prm->read_null, a.n_reads * 4, hipMemcpyHostToDevice, s));
            if (prm->read_ignore_score_len) {
                MG_TRY(gmg_pool_alloc((void **)&d_read_isl, a.n_reads * 4));
                MG_TRY(hipMemcpyAsync(d_read_isl, prm->read_ignore_score_len, a.n_reads * 4, hipMemcpyHostToDevice, s));
            }
        }
        a.read_null = d_read_null;
        a.read_isl = d_read_isl;
    }
    a.null_tab = d_null_tab;
    // The call's own table in the default mode is the GENE32 form: the gene model's fp32 rows (half the bytes to write, half to
    // read back), the null model applied where the running sums are built.  A caller's table, the error branch (its walks read
    // the table in place) and model shapes without the fast path keep the fp64 table of gmg_frame_score6.
    const bool nul_dense3 = nul->dev.has_dense && nul->dev.W == 3 && nul->dev.P == 3 && nul->dev.dense_part == nul->dev.dense + 192;
    // Measured (1M x 500 bp, profiles/r02_mg_*): with ONE null model the fp64 table wins (running sums 7.2 ms against 9.1 ms: the
    // conversion costs more vector work than the halved read saves); with per-read null models GENE32 saves the extra pass over
    // the table (13.3 ms against 16.6 ms for the table + sums).  Option mg_gene32: 0 never, 1 with per-read nulls and with tiles of two waves or more (default), 2 always.
    // The fused kernel (k_mg_tile_starts: sums as a parallel scan + start lists) when every sum of the batch is exact in any
    // order -- see there; R = the longest read + 2 terms.
    {
        const int n_min = prm->nulls ? prm->nulls->min_exp : nul->min_exp, n_max = prm->nulls ? prm->nulls->max_exp : nul->max_exp;
        const int n_odd = prm->nulls ? prm->nulls->odd_values : nul->odd_values;
        int g_min = gene->min_exp, g_max = gene->max_exp, g_odd = gene->odd_values;
        for (int k = 0; groups && k < groups->n; k++) {   // every group's model
            const gmg_model *m = groups->models[k];
            if (m->min_exp < g_min) g_min = m->min_exp;
            if (m->max_exp > g_max) g_max = m->max_exp;
            g_odd |= m->odd_values;
        }
        const int mn = g_min < n_min ? g_min : n_min, mx = g_max > n_max ? g_max : n_max;
        int clog = 0;
        const uint64_t longest_read = reads->max_len ? reads->max_len : reads->total_bases;   // (no lengths on the host: the batch's size is a bound)
        while ((1ull << clog) < longest_read + 2) clog++;
        const bool exact = !g_odd && !n_odd && (mx < mn || clog + mx - mn <= 28);
        err_exact = exact;
        const long long forced_tile = gmg_opt(GMG_OPT_MG_TILE);
        if (!err_mode && exact && gmg_opt(GMG_OPT_MG_FUSED) && a.n_reads && a.total) {
            if (forced_tile == 1 || forced_tile == 2 || forced_tile == 4) fused_nw = (int)forced_tile;
            else if (reads->uniform_len > 0) fused_nw = reads->uniform_len <= MT_W ? 1 : reads->uniform_len <= 2 * MT_W ? 2 : reads->uniform_len <= 4 * MT_W ? 4 : 0;
            else fused_nw = (reads->max_len <= MT_W || reads->n_over_512 * 10 <= reads->n_reads) ? 1 : reads->max_len <= 2 * MT_W ? 2 : 4;
            if (reads->uniform_len > (int)(MT_W * fused_nw)) fused_nw = 0;
            // Measured (1M reads, tests/bench/bench_mg.py with GMG_MG_TILE; profiles/r02_mg_tile_width.txt): ragged reads fill four-wave tiles
            // better than one-wave ones (~400 bp: 70 % of 567 bases, 88 % of 2,268) -- 12.5 -> 11.7 ms with one null model, 13.7 -> 10.95 ms with
            // a null model per read (one LDS table per read and tile: fewer, fuller tiles); two-wave tiles for uniform batches (two 500-bp reads
            // per tile): 12.3 -> 11.1 ms with a null model per read, 11.1 -> 10.9 ms with one (fp64 table), 11.5 -> 10.2 ms with the GENE32
            // table, which is why the call's own table takes that form whenever the tiles have two waves or more.
            // (A tile takes whole reads, at most tile_reads_max of them.)
            if (!(forced_tile == 1 || forced_tile == 2 || forced_tile == 4) && fused_nw) {
                const uint64_t mean = a.total / a.n_reads, per_tile = prm->nulls ? MT_NC : MG_TILE_READS;
                if (reads->uniform_len == 0 && mean * per_tile * 5 >= (uint64_t)4 * MT_W * 4) fused_nw = 4;
                else if (reads->uniform_len > 0 && fused_nw == 1 && (uint64_t)reads->uniform_len * per_tile >= (uint64_t)2 * MT_W) fused_nw = 2;
            }
            // eight elements per lane (504 bases per wave) when the reads of a uniform batch fill such tiles as well as the larger ones
            if (fused_nw && reads->uniform_len > 0) {
                const int l = reads->uniform_len, c8 = 504 * fused_nw, c9 = (int)MT_W * fused_nw;
                if (l <= c8 && (c8 / l) * 9 >= (c9 / l) * 8) fused_el = 8;
            }
            // ragged batches too when no read needs the wider tile: the eight-element form runs four waves per SIMD (9.76 -> 9.54 ms per 1 M x ~400 bp)
            else if (fused_nw && reads->max_len && reads->max_len <= (uint64_t)504 * fused_nw) fused_el = 8;
        }
    }
    // (the fused kernel reads whichever table there is.  Measured, 1M x 500 bp, one null model: the fp64 table 4.9 + 6.2 ms, the
    // GENE32 form 3.9 + 7.5 ms -- the kernel is bound by its vector instructions, not by the table's bytes, and the null-model
    // lookups add a quarter to them; profiles/r02_mg_pmc_*.txt)
    const long long g32_opt = gmg_opt(GMG_OPT_MG_GENE32);
    // (with tiles of two waves or more the GENE32 form wins with one null model as well: ragged 10.6 -> 10.0 ms per 1M x ~400 bp,
    // 500-bp reads 10.9 -> 10.2 ms)
    bool all_fast = gene->dev.has_fast && gene->dev.D == 7 && gene->dev.W >= 3 && gene->dev.W <= 15;
    for (int k = 0; groups && k < groups->n; k++) {
        const GmgDevModel &m = groups->models[k]->dev;
        all_fast = all_fast && m.has_fast && m.D == 7 && m.W == gene->dev.W;
    }
    // (groups of any-shape models: their gene rows come from the exact kernel, group by group -- still the GENE32 form)
    // the error branch on running sums (mg_err_skip) never reads the table itself, only the walk-order sums made from it: the
    // gene rows as fp32 (GENE32) then save the 48 B/base table's write and half of what the sums' kernel reads.  Not with reads the
    // level kernels cannot take (>= 2040 bases: k_mg_err_flat walks the table) or a forced per-ORF path; a call-array overflow
    // builds the table then (k_mg_apply_nulls) before it falls back.
    // the error branch tile by tile (k_mg_err_tile: the running sums in LDS) wants what the running-sum form wants; reads longer than
    // a tile go to k_mg_err_flat, which walks the fp64 table
    // Which of the two: measured on ragged ~400-bp reads (profiles/r04_errtile_crossover.txt), the tile kernel -- ONE launch, no table in
    // HBM -- wins on batches up to ~200,000 reads (-i: 1.1 vs 2.4 ms at 5,000 reads, 3.4 vs 4.7 at 50,000, 11.3 vs 11.6 at 200,000), the
    // level kernels from there on (22.1 vs 21.2 ms at 400,000, 54.3 vs 48.6 at 1M).  mg_err_tile: -1 (default) by the batch's size,
    // 1 always, 0 never.
    const long long tile_opt = gmg_opt(GMG_OPT_MG_ERR_TILE);
    const bool tile_wanted = tile_opt > 0 || (tile_opt < 0 && a.total <= (err_mode == 1 ? (uint64_t)MG_ET_AUTO_BASES_INDEL : (uint64_t)MG_ET_AUTO_BASES_SUB));
    err_tile = err_mode && err_exact && gmg_opt(GMG_OPT_MG_ERR_SKIP) && tile_wanted && !gmg_opt(GMG_OPT_MG_ERR_FLAT);
    // one wave per (read, strand) (k_mg_err_wave; the default whenever the sums are exact, unless the tile kernel is forced):
    // its LDS share is sized by the batch's longest read, reads beyond EW_MAX_CAP go to k_mg_err_flat
    // mg_err_wave: 1 (default), 2 = with the stack walker as the count pass too (cross-check), 0 = the tile / level kernels
    err_wave = err_mode && err_exact && gmg_opt(GMG_OPT_MG_ERR_SKIP) && !gmg_opt(GMG_OPT_MG_ERR_FLAT) && gmg_opt(GMG_OPT_MG_ERR_WAVE) > 0 &&
               !tl_mg_no_wave && tile_opt <= 0 && reads->max_len > 0 && a.total;
    if (err_wave) {
        err_tile = false;
        const uint64_t longest = reads->max_len < EW_MAX_CAP ? reads->max_len : EW_MAX_CAP;
        ew_cap = (uint32_t)((longest + 63) & ~63ull);
    }
    const uint64_t err_fit_len = err_wave ? (uint64_t)ew_cap + 1 : err_tile ? MG_ET_CAP + 1 : 2040;   // reads shorter than this are walked by the wave / tile / level kernels
    const bool err_g32 = err_mode && !d_frame_scores && a.total && g32_opt != 0 && nul_dense3 && all_fast && err_exact &&
                         gmg_opt(GMG_OPT_MG_ERR_SKIP) && !gmg_opt(GMG_OPT_MG_ERR_FLAT) && reads->max_len && reads->max_len < err_fit_len;
    const bool g32 = err_g32 || (!d_frame_scores && !err_mode && a.total &&
                     (g32_opt == 2 || (g32_opt == 1 && (prm->nulls || fused_nw >= 2))) && nul_dense3 &&
                     (all_fast || (groups && groups->n > 0)));
    if (prm->nulls && !nul_dense3) return fail(gmg_set_error(GMG_EBADMODEL, "gmg_mg_score_reads: per-read null models are (3,2,3) models"));
    a.fs_stride = a.total;
    if (g32) {
        a.fs_stride = (a.total + 31) & ~31ull;          // every fp32 row on a 128-byte line
        // (64 spare floats on both sides: the wave kernels of the error branch load their lanes' steps without predicates)
        MG_TRY(gmg_pool_alloc((void **)&d_gene32, ((size_t)6 * a.fs_stride + 128) * sizeof(float)));
        rc = mg_gene6_full(gene, groups, reads, d_gene32 + 64, a.fs_stride, s);
        if (rc) return fail(rc);
        a.gene32 = d_gene32 + 64;
        a.ew_slack = 1;
    } else {
        if (!d_frame_scores && a.total) {
            a.fs_stride = (a.total + 15) & ~15ull;      // our own table: every row on a 128-byte line
            MG_TRY(gmg_pool_alloc((void **)&d_fs_own, (size_t)6 * a.fs_stride * sizeof(double)));
            d_frame_scores = d_fs_own;
        }
        if (a.total) {
            if (prm->nulls) rc = mg_frame6_nulls(gene, d_null_tab, d_read_null, reads, d_frame_scores, a.fs_stride, s, groups);
            else rc = gmg_launch_frame6_strided(gene, nul, reads, d_frame_scores, a.fs_stride, s);
            if (rc) return fail(rc);
        }
        a.fs = d_frame_scores;
    }
    tm.lap("frame scores");
    if (err_mode) {                                     // the error branch sums per call; it needs the penalties and the qualities
        if (err_mode == 1 && a.total) {
            MG_TRY(gmg_pool_alloc((void **)&d_qual, a.total + 8 + 128));     // (+8: the level kernels read four values at a time; 64 spare bytes on both sides)
            if (prm->quality) MG_TRY(gmg_pool_alloc((void **)&d_user_q, a.total));
            a.qual = d_qual + 64;                       // filled below, beside Find_Orfs on the second stream
        }
    }
    // running sums of every reading-frame class (what Cumulative_Frame_Score would give any ORF)
    if (!err_mode && a.n_reads && a.total) {
        // tile shape of the sequential kernel: two waves and <= 512 bases (12 KB of LDS, many blocks per CU in different phases)
        // when the reads allow it, else eight waves and 1504 bases (39.8 KB, four blocks per CU); the fused kernel: 567 bases per wave
        const long long forced_tile = gmg_opt(GMG_OPT_MG_TILE);
        // (ragged batches: the few reads beyond 512 bases go to the per-lane kernel; measured 9.6 vs 10.6 ms on 1M x ~400 bp)
        const bool small = forced_tile ? forced_tile == 512 : (reads->max_len <= 512 || (reads->uniform_len == 0 && reads->n_over_512 * 10 <= reads->n_reads));
        const uint32_t cap = fused_nw ? (uint32_t)(3 * MT_CL * fused_el * fused_nw) : small ? 512 : 1504;
        a.tile_cap = (int)cap;
        a.tile_reads_max = fused_nw && g32 && prm->nulls ? MT_NC : MG_TILE_READS;
        // (two-wave tiles of eight elements over uniform reads of which two at most fit: the kernel form with two null tables)
        fused_nc2 = fused_nw == 2 && fused_el == 8 && g32 && prm->nulls && reads->uniform_len > 0 && cap / (uint32_t)reads->uniform_len <= 2;
        if (fused_nc2) a.tile_reads_max = 2;
        bool tiled = false, rest = true;
        if (reads->uniform_len > 0) {                  // every tile takes cap / L whole reads
            if ((uint32_t)reads->uniform_len <= cap) {
                a.uniform_len = reads->uniform_len;
                a.uniform_magic = (uint32_t)((0x100000000ull + (uint64_t)reads->uniform_len - 1) / (uint64_t)reads->uniform_len);
                a.reads_per_tile = (int)(cap / reads->uniform_len) < a.tile_reads_max ? cap / reads->uniform_len : a.tile_reads_max;
                a.n_tiles = (a.n_reads + a.reads_per_tile - 1) / a.reads_per_tile;
                tiled = true; rest = false;
            }
        } else {                                        // the reads that start inside a window of tile_window bases
            const uint64_t longest = reads->max_len < cap / 2 ? reads->max_len : cap / 2;
            a.tile_window = cap - longest;
            const uint64_t n_windows = a.total / a.tile_window + 1;
            tiled = true;
            rest = reads->max_len > longest || reads->min_len * (uint64_t)a.tile_reads_max < a.tile_window;
            if (n_windows >= 0x7fffffffull) return fail(gmg_set_error(GMG_EINVAL, "gmg_mg_score_reads: batch too large"));
            uint32_t *d_n = nullptr, *d_flag = nullptr;     // d_flag: [n_windows + 1] flags, then [n_windows + 1] their exclusive sums
            hipError_t e = gmg_pool_alloc((void **)&d_all, n_windows * sizeof(MgTile));
            if (e == hipSuccess) e = gmg_pool_alloc((void **)&d_tiles, n_windows * sizeof(MgTile));
            if (e == hipSuccess) e = gmg_pool_alloc((void **)&d_n, 4);
            if (e == hipSuccess) e = gmg_pool_alloc((void **)&d_flag, (2 * (n_windows + 4)) * 4);
            d_sel_tmp = d_flag;
            if (e == hipSuccess) {
                uint32_t *d_pos = d_flag + ((n_windows + 4) & ~3ull);
                hipLaunchKernelGGL(k_mg_tile_table, dim3(grid_for(n_windows)), dim3(256), 0, s, a, n_windows, cap, d_all);
                hipLaunchKernelGGL(k_mg_tile_flags, dim3(grid_for(n_windows + 1)), dim3(256), 0, s, d_all, n_windows, d_flag);
                e = gmg_scan_excl<uint32_t, uint32_t>(d_flag, d_pos, n_windows + 1, s);
                if (e == hipSuccess) hipLaunchKernelGGL(k_mg_tile_compact, dim3(grid_for(n_windows)), dim3(256), 0, s, d_all, n_windows, d_pos, d_tiles, d_n);
                if (e == hipSuccess) e = hipGetLastError();
            }
            d_ntiles = d_n;                                 // (all of these go back to the cache after the call's final synchronise)
            if (e != hipSuccess) return fail(gmg_set_error(GMG_EHIP, "gmg_mg_score_reads: tile table: %s", hipGetErrorString(e)));
            a.tiles = d_tiles;
            a.windows = d_all;
            a.n_tiles_dev = d_n;
            a.n_tiles = n_windows;                          // upper bound, for the grid
        }
        if (fused_nw && !tiled) fused_nw = 0;
        fused_rest = rest;
        if (!fused_nw || rest) {
            MG_TRY(gmg_pool_alloc((void **)&d_cum, (size_t)2 * a.total * sizeof(double)));
            a.cum = d_cum;
        }
        if (fused_nw) {
            // the kernel itself goes behind the ORF scan and the count pass (it writes the start lists); the reads no tile takes are
            // listed now
            if (rest) {
                a.lanes_only_unfit = 1;
                MG_TRY(gmg_pool_alloc((void **)&d_unfit, (a.n_reads + 1) * 4));
                a.unfit = d_unfit + 1;
                a.unfit_n = d_unfit;
                MG_TRY(hipMemsetAsync(d_unfit, 0, 4, s));
                hipLaunchKernelGGL(k_mg_unfit_list, dim3(grid_for(a.n_reads)), dim3(256), 0, s, a);
                if (g32) hipLaunchKernelGGL(k_mg_cum<true>, dim3(grid_for(2 * a.n_reads)), dim3(256), 0, s, a);
                else hipLaunchKernelGGL(k_mg_cum<false>, dim3(grid_for(2 * a.n_reads)), dim3(256), 0, s, a);
                MG_TRY(hipGetLastError());
            }
        } else {
        if (tiled && a.n_tiles) {
            const size_t lds = (size_t)3 * (cap + 8) * sizeof(double);
            const unsigned grid = (unsigned)(2 * a.n_tiles < 256 * 1024 ? 2 * a.n_tiles : 256 * 1024);
            if (small) {                                // two waves per tile: 7.1 ms; one: 8.0; four: 9.0
                if (g32) hipLaunchKernelGGL((k_mg_cum_tiled<512, 128, true>), dim3(grid), dim3(128), lds, s, a);
                else hipLaunchKernelGGL((k_mg_cum_tiled<512, 128, false>), dim3(grid), dim3(128), lds, s, a);
            } else {
                // eight waves per tile: 8.8 ms per 400k x 1000 bp; four: 9.7; two: 11.1; twelve: 13.3
                if (g32) {
                    MG_TRY(hipFuncSetAttribute((const void *)k_mg_cum_tiled<1504, 512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    hipLaunchKernelGGL((k_mg_cum_tiled<1504, 512, true>), dim3(grid), dim3(512), lds, s, a);
                } else {
                    MG_TRY(hipFuncSetAttribute((const void *)k_mg_cum_tiled<1504, 512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    hipLaunchKernelGGL((k_mg_cum_tiled<1504, 512, false>), dim3(grid), dim3(512), lds, s, a);
                }
            }
            MG_TRY(hipGetLastError());
        }
        if (rest) {
            a.lanes_only_unfit = tiled && a.windows != nullptr;
            if (a.lanes_only_unfit) {
                MG_TRY(gmg_pool_alloc((void **)&d_unfit, (a.n_reads + 1) * 4));
                a.unfit = d_unfit + 1;
                a.unfit_n = d_unfit;
                MG_TRY(hipMemsetAsync(d_unfit, 0, 4, s));
                hipLaunchKernelGGL(k_mg_unfit_list, dim3(grid_for(a.n_reads)), dim3(256), 0, s, a);
                MG_TRY(hipGetLastError());
            }
            if (g32) hipLaunchKernelGGL(k_mg_cum<true>, dim3(grid_for(2 * a.n_reads)), dim3(256), 0, s, a);
            else hipLaunchKernelGGL(k_mg_cum<false>, dim3(grid_for(2 * a.n_reads)), dim3(256), 0, s, a);
            MG_TRY(hipGetLastError());
        }
        }
    }
    tm.lap("running sums");
    }
    // 2. ORFs of every read.  Find_Orfs and the count pass of the start scan need only the packed reads: they run on a second
    //    stream beside the six-frame and running-sum kernels (light kernels without LDS, they fit next to the main pass's
    //    work-groups) and join the caller's stream before the start lists are written.
    static thread_local hipStream_t side_of[16] = {};   // one per device this host thread has used
    static thread_local hipEvent_t done_of[16] = {}, cum_of[16] = {};
    static thread_local hipStream_t side2_of[16] = {};  // error branch: the qualities and the run lengths beside the ORF scan and the six-frame table
    static thread_local hipEvent_t done2_of[16] = {};
    hipStream_t s2 = s, s3 = s;
    hipEvent_t side_done = nullptr, cum_done = nullptr, side2_done = nullptr;
    int dev_id = 0;
    MG_TRY(hipGetDevice(&dev_id));
    if (!find_only && !tm.on && !gmg_opt(GMG_OPT_MG_ONE_STREAM) && dev_id >= 0 && dev_id < 16) {
        // (the side streams get the lowest priority: their kernels fill the gaps of the caller's stream and must not keep its short
        // kernels waiting -- k_frame6p, 0.45 ms alone, took 5.9 ms beside the error branch's side kernels at equal priority)
        int prio_least = 0, prio_greatest = 0;
        MG_TRY(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        if (!side_of[dev_id]) {
            MG_TRY(hipStreamCreateWithPriority(&side_of[dev_id], hipStreamNonBlocking, prio_least));
            MG_TRY(hipEventCreateWithFlags(&done_of[dev_id], hipEventDisableTiming));
            MG_TRY(hipEventCreateWithFlags(&cum_of[dev_id], hipEventDisableTiming));
        }
        s2 = side_of[dev_id];
        side_done = done_of[dev_id];
        cum_done = cum_of[dev_id];
        s3 = s2;
        if (err_mode) {
            if (!side2_of[dev_id]) {
                MG_TRY(hipStreamCreateWithPriority(&side2_of[dev_id], hipStreamNonBlocking, prio_least));
                MG_TRY(hipEventCreateWithFlags(&done2_of[dev_id], hipEventDisableTiming));
            }
            s3 = side2_of[dev_id];
            side2_done = done2_of[dev_id];
        }
    }
    const uint64_t nr = a.n_reads;
    // (in front of the ORF scan's count pass: behind it the launch waited for the host to come back from the scan's total -- it then
    // ran beside the ORF write pass, both at half speed, 1 ms on the error branch's critical path; here it runs in the six-frame
    // kernel's shadow: no LDS, few registers)
    // The wave kernels compute Set_Quality_454 themselves from the bases they hold (no quality file, every read short enough for them):
    // no quality kernel, no 2 B/base of quality arrays written and read back; a fall-back to the level kernels makes them then
    auto build_qualities = [&](hipStream_t st) -> hipError_t {
        hipError_t e = hipSuccess;
        if (prm->quality) e = hipMemcpyAsync(d_user_q, prm->quality, a.total, hipMemcpyHostToDevice, st);
        if (e == hipSuccess && !d_walk_q) e = gmg_pool_alloc((void **)&d_walk_q, a.total + 8);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_mg_quality, dim3(grid_for(nr * 64)), dim3(256), 0, st, a, d_user_q, d_qual + 64, d_walk_q);
        a.walk_q = d_walk_q;
        return hipGetLastError();
    };
    a.q454 = err_mode == 1 && err_wave && !prm->quality && reads->max_len <= (uint64_t)ew_cap && gmg_opt(GMG_OPT_MG_ERR_WAVE) != 0 ? 1 : 0;
    if (err_mode == 1 && a.total && !find_only && !a.q454) {   // Set_Quality_454 / Clean_Quality_454: needs the reads only
        MG_TRY(build_qualities(s3));
        tm.lap("quality values");
    }
    MG_TRY(gmg_pool_alloc((void **)&d_read_cnt, (nr + 1) * 4));
    MG_TRY(hipMemsetAsync(d_read_cnt, 0, (nr + 1) * 4, s2));
    MG_TRY(gmg_pool_alloc((void **)&res->d_read_orf_off, (nr + 1) * 8));
    a.read_cnt = d_read_cnt;
    // Find_Orfs on bit masks (k_mg_find_orfs_bits): a wave per window of whole reads under its OB_WORDS x 32 bases, ten reads at a time.
    // Uniform batches: up to ten reads per window; ragged ones: the reads that begin in a stretch of about ten mean read lengths (and
    // no more than the window holds with the longest read at its end).  Not with reads beyond OB_MAX_LEN.
    const bool orf_bits = !general && nr && a.total && gmg_opt(GMG_OPT_MG_ORFS_BITS) && reads->max_len <= OB_MAX_LEN && reads->max_len > 0;
    uint32_t ob_win_bases = 0, ob_rpw = 0, ob_grid = 1;
    uint64_t ob_windows = 0;
    if (orf_bits) {
        if (reads->uniform_len > 0) {
            ob_rpw = (uint32_t)((OB_SPAN - 31) / reads->uniform_len);
            if (ob_rpw > OB_GROUP) ob_rpw = OB_GROUP;
            ob_windows = (nr + ob_rpw - 1) / ob_rpw;
        } else {
            const uint64_t room = OB_SPAN - 31 - reads->max_len, want = (a.total * 19 / 2) / nr;      // 9.5 mean lengths
            ob_win_bases = (uint32_t)(want < room ? (want > 0 ? want : 1) : room);
            ob_windows = (a.total + ob_win_bases - 1) / ob_win_bases;
        }
        const uint64_t blocks = (ob_windows + OB_WAVES - 1) / OB_WAVES;
        ob_grid = (uint32_t)(blocks < 256 * 32 ? blocks : 256 * 32);
    }
    int32_t *d_ign = nullptr;                           // general form: the regions' lo values, then their hi values; [2 n]: the failure flag
    const int n_ign = general ? prm->n_ignore_regions : 0;
    if (general) {
        MG_TRY(gmg_pool_alloc((void **)&d_ign, (size_t)(2 * n_ign + 1) * 4));
        if (n_ign) {
            MG_TRY(hipMemcpyAsync(d_ign, prm->ignore_lo, (size_t)n_ign * 4, hipMemcpyHostToDevice, s2));
            MG_TRY(hipMemcpyAsync(d_ign + n_ign, prm->ignore_hi, (size_t)n_ign * 4, hipMemcpyHostToDevice, s2));
        }
        MG_TRY(hipMemsetAsync(d_ign + 2 * n_ign, 0, 4, s2));
        if (nr) hipLaunchKernelGGL(k_find_orfs_general<false>, dim3(grid_for(nr)), dim3(64), 0, s2, a, prm->circular ? 1 : 0, n_ign, d_ign, d_ign + n_ign,
                                   (uint32_t *)(d_ign + 2 * n_ign));
    } else if (orf_bits) {
        hipLaunchKernelGGL(k_mg_find_orfs_bits<false>, dim3(ob_grid), dim3(64 * OB_WAVES), 0, s2, a, ob_windows, ob_win_bases, ob_rpw);
    } else if (nr) hipLaunchKernelGGL(k_mg_find_orfs<false>, dim3(grid_for(nr)), dim3(256), 0, s2, a);
    MG_TRY(hipGetLastError());
    rc = mg_scan(d_read_cnt, res->d_read_orf_off, nr, &res->n_orfs, s2);
    if (rc) return fail(rc);
    if (res->n_orfs > (uint64_t)gmg_opt(GMG_OPT_MG_MAX_ENTRIES))
        return fail(gmg_set_error(GMG_ETOOBIG, "gmg_mg_score_reads: %llu ORFs in one batch, the result's 32-bit fields hold %lld: split the batch",
                                  (unsigned long long)res->n_orfs, gmg_opt(GMG_OPT_MG_MAX_ENTRIES)));
    const uint64_t no = res->n_orfs;
    MG_TRY(gmg_pool_alloc((void **)&res->d_orfs, (no ? no : 1) * sizeof(gmg_mg_orf)));
    a.read_orf_off = res->d_read_orf_off;
    a.orfs = res->d_orfs;
    a.n_orfs = no;
    {   // default mode: the write pass of the ORF scan counts every ORF's starts as it goes (lowest j within the 64 codons its mask holds)
        const int j_lo = ((a.min_gene_len - 3 > 1 ? a.min_gene_len - 3 : 1) + 2) / 3 * 3;
        a.count_starts = !find_only && !err_mode && 1 + j_lo / 3 <= 64;
        if (!find_only) {
            MG_TRY(gmg_pool_alloc((void **)&d_orf_cnt, (no + 1) * 4));
            // (the counting write pass stores every ORF's count itself: only the scan's extra element needs a zero -- the 30 MB
            // memset sat 0.26 ms on the critical path, in front of the write pass)
            if (!a.count_starts || nr == 0) MG_TRY(hipMemsetAsync(d_orf_cnt, 0, (no + 1) * 4, s2));   // (else the write pass zeroes the last element)
            a.orf_cnt = d_orf_cnt;
        }
    }
    if (general) {
        if (nr) hipLaunchKernelGGL(k_find_orfs_general<true>, dim3(grid_for(nr)), dim3(64), 0, s2, a, prm->circular ? 1 : 0, n_ign, d_ign, d_ign + n_ign,
                                   (uint32_t *)(d_ign + 2 * n_ign));
        uint32_t failed = 0;
        MG_TRY(hipMemcpyAsync(&failed, d_ign + 2 * n_ign, 4, hipMemcpyDeviceToHost, s2));
        MG_TRY(hipStreamSynchronize(s2));
        gmg_pool_release(d_ign);
        d_ign = nullptr;
        if (failed)     // Wrap_Around_Back: assert (pos > 0) -- a circular sequence with a reverse frame that has no stop codon (behind the last ignore region)
            return fail(gmg_set_error(GMG_EINVAL, "gmg_find_orfs: a circular sequence has a reverse reading frame without a stop codon (the reference aborts: "
                                                  "Wrap_Around_Back, glimmer_base.cc:2793)"));
    } else if (orf_bits) {
        hipLaunchKernelGGL(k_mg_find_orfs_bits<true>, dim3(ob_grid), dim3(64 * OB_WAVES), 0, s2, a, ob_windows, ob_win_bases, ob_rpw);
    } else if (nr && gmg_opt(GMG_OPT_MG_ORFS_EVENTS)) {
        const uint64_t blocks = (nr + MG_EV_LANES - 1) / MG_EV_LANES;
        if (gmg_opt(GMG_OPT_MG_ORFS_EVENTS) == 1) hipLaunchKernelGGL(k_mg_find_orfs_ev<false>, dim3((unsigned)(blocks < 256 * 128 ? blocks : 256 * 128)), dim3(MG_EV_LANES), 0, s2, a);
        else hipLaunchKernelGGL(k_mg_find_orfs_ev<true>, dim3((unsigned)(blocks < 256 * 128 ? blocks : 256 * 128)), dim3(MG_EV_LANES), 0, s2, a);
    } else if (nr) hipLaunchKernelGGL(k_mg_find_orfs<true>, dim3(grid_for(nr)), dim3(256), 0, s2, a);
    MG_TRY(hipGetLastError());
    tm.lap("find orfs");
    // error branch, level by level: 0 (k_mg_err_level; the default), 1 = one lane per ORF with an explicit stack
    // (k_mg_err_flat: exact slots; the fallback of 0, and on its own with GMG_MG_ERR_FLAT=1 for A/B runs and cross-checks)
    int err_path = gmg_opt(GMG_OPT_MG_ERR_FLAT) ? 1 : 0;
    // the run lengths of the level kernels' walks (they need the reads and the qualities only)
    auto build_run_tables = [&](hipStream_t st) -> hipError_t {
        hipError_t e = gmg_pool_alloc((void **)&d_run, (size_t)4 * a.walk_stride);
        if (e != hipSuccess) return e;
        a.run_q = d_run; a.run_n = d_run + 2 * a.walk_stride;
        hipLaunchKernelGGL(k_mg_run_tables, dim3(grid_for(2 * nr * 64)), dim3(256), 0, st, a, d_run, d_run + 2 * a.walk_stride);
        return hipGetLastError();
    };
    // the walk-order rows of the level kernels (running sums, or the values themselves): behind the six-frame table on stream st
    auto build_walk_rows = [&](hipStream_t st) -> hipError_t {
        // -s on running sums: one value per base and strand is all its walks read (16 B/base instead of 48: a third of the table to
        // write, a third to keep in the caches)
        a.qonly = a.pfx && err_mode == 2 && gmg_opt(GMG_OPT_MG_ERR_QONLY) ? 1 : 0;
        hipError_t e = gmg_pool_alloc((void **)&d_walk, ((size_t)(a.qonly ? 2 : 6) * a.walk_stride + 8) * sizeof(double));
        if (e != hipSuccess) return e;
        if (a.qonly) {
            // (tried: the same table codon by codon -- a lane on three consecutive steps, ONE scan per class per 192 steps instead of
            // per 64: bit-exact, 23.5 against 22.0 ms per 1M reads with -s: the loads of a lane's three steps no longer coalesce)
            if (a.gene32) hipLaunchKernelGGL((k_mg_walk_prefix<true, true>), dim3(grid_for(2 * nr * 64)), dim3(256), 0, st, a, d_walk + 8);
            else hipLaunchKernelGGL((k_mg_walk_prefix<false, true>), dim3(grid_for(2 * nr * 64)), dim3(256), 0, st, a, d_walk + 8);
        } else if (a.pfx) {
            if (a.gene32) hipLaunchKernelGGL(k_mg_walk_prefix<true>, dim3(grid_for(2 * nr * 64)), dim3(256), 0, st, a, d_walk + 8);
            else hipLaunchKernelGGL(k_mg_walk_prefix<false>, dim3(grid_for(2 * nr * 64)), dim3(256), 0, st, a, d_walk + 8);
        } else
            hipLaunchKernelGGL(k_mg_walk_tables, dim3(grid_for(a.total)), dim3(256), 0, st, a, d_walk + 8);
        a.walk = d_walk + 8;                            // (8 spare entries in front: a call at the table's first entry looks one back)
        return hipGetLastError();
    };
    if (!find_only && res->n_orfs && err_mode && err_path == 0) {
        // the walk-order tables: the rows (running sums) need the six-frame table and go behind it on the caller's stream; the run
        // lengths need the reads and the qualities only and follow the quality kernel on a stream of their own, beside the ORF scan
        // (second stream), the six-frame kernel and the rows.  Tile by tile (k_mg_err_tile) the rows are built in LDS, tile by tile.
        a.walk_stride = ((a.total + 15) & ~15ull) + 16;
        a.pfx = err_exact && gmg_opt(GMG_OPT_MG_ERR_SKIP) ? 1 : 0;
        if (a.pfx && !err_tile && !err_wave) MG_TRY(build_run_tables(s3));   // running sums + run lengths: the walks visit their events only
        if (!err_tile && !err_wave) MG_TRY(build_walk_rows(s));
        MG_TRY(hipGetLastError());
        tm.lap("walk-order tables");
    }
    // 3. start lists
    if (!find_only) {
    MG_TRY(gmg_pool_alloc((void **)&d_start_off, (no + 1) * 8));
    // the level kernels' scratch: call arrays, per-ORF aggregates, slot counters
    // (what the level passes of this thread's last call handed on per base: a run's chunks are alike, and a count pass that finds its
    // arrays too small runs twice -- weakly trained models keep several times the branches of a real one alive)
    static thread_local double calls_per_base_hint = 0.0;
    auto alloc_level_scratch = [&]() -> hipError_t {
        a.call_cap = a.total / 2 > 65536 ? a.total / 2 : 65536;
        const uint64_t hinted = (uint64_t)(calls_per_base_hint * 1.15 * (double)a.total) + 65536;
        if (hinted > a.call_cap && hinted <= 8 * a.total + 65536) a.call_cap = hinted;
        if (gmg_opt(GMG_OPT_MG_ERR_CALLS) > 0) a.call_cap = (uint64_t)gmg_opt(GMG_OPT_MG_ERR_CALLS);     // (tests: force the fallback)
        hipError_t e = gmg_pool_alloc((void **)&d_calls[0], a.call_cap * sizeof(MgCall));
        if (e == hipSuccess) e = gmg_pool_alloc((void **)&d_calls[1], a.call_cap * sizeof(MgCall));
        if (e == hipSuccess) e = gmg_pool_alloc((void **)&d_agg, no * sizeof(MgOrfAgg));
        if (e == hipSuccess) e = gmg_pool_alloc((void **)&d_fill, no * 4);
        a.calls[0] = d_calls[0]; a.calls[1] = d_calls[1]; a.agg = d_agg; a.fill = d_fill;
        return e;
    };
    // k_mg_err_tile: work-groups (ET_WG_PER_CU per CU: sizeof (EtLds<MG_ET_CAP>) of LDS each), their slabs (calls per level, starts per batch of ORFs), the tile
    // list, the staging arrays
    unsigned et_grid = 0;
    const uint32_t et_qcap = gmg_opt(GMG_OPT_MG_ERR_TILE_Q) > 0 ? (uint32_t)gmg_opt(GMG_OPT_MG_ERR_TILE_Q) : (uint32_t)ET_QCAP;
    const uint32_t et_ecap = gmg_opt(GMG_OPT_MG_ERR_TILE_Q) > 0 ? (uint32_t)(4 * gmg_opt(GMG_OPT_MG_ERR_TILE_Q)) : (uint32_t)ET_ECAP;
    uint64_t et_stage_cap = 0;
    uint32_t *d_et_ntiles = nullptr;
    unsigned long long *d_et_items = nullptr, *d_et_stage_ctr = nullptr;
    const int err_acc_only = (prm->flags & GMG_MG_ACCEPTED_ONLY) ? 1 : 0;
    auto alloc_staging = [&](uint64_t entries) -> hipError_t {
        if (d_st_s) { gmg_pool_release(d_st_s); d_st_s = nullptr; }
        if (d_st_e) { gmg_pool_release(d_st_e); d_st_e = nullptr; }
        if (d_st_k) { gmg_pool_release(d_st_k); d_st_k = nullptr; }
        et_stage_cap = entries;
        hipError_t e = gmg_pool_alloc((void **)&d_st_s, entries * sizeof(gmg_start));
        if (e == hipSuccess) e = gmg_pool_alloc((void **)&d_st_e, entries * sizeof(gmg_start_errors));
        if (e == hipSuccess) e = gmg_pool_alloc((void **)&d_st_k, entries * 8);
        return e;
    };
    if (res->n_orfs && err_mode && err_path == 0) {
        MG_TRY(gmg_pool_alloc((void **)&d_read_fit, nr ? nr : 1));
        MG_TRY(gmg_pool_alloc((void **)&d_err_flag, 256));          // the flag + the two call counters + six tile counters; tile path: + the number of tiles, the item and staging counters; [32 ..]: the block queues of k_mg_err_wcount (count, write x length class)
        MG_TRY(hipMemsetAsync(d_err_flag, 0, 256, s3));             // (not behind the ORF write pass on the first side stream; both join the caller's below)
        a.read_fit = d_read_fit;
        a.err_flag = d_err_flag;
        a.n_calls = (unsigned long long *)(d_err_flag + 2);
        a.tile_ctr = (unsigned long long *)(d_err_flag + 6);
        d_et_ntiles = d_err_flag + 18;
        d_et_items = (unsigned long long *)(d_err_flag + 20);
        d_et_stage_ctr = (unsigned long long *)(d_err_flag + 22);
        MG_TRY(gmg_pool_alloc((void **)&d_acc_bits, (no / 32 + 1) * 4));
        a.acc_bits = d_acc_bits;
        if (err_tile) {
            int n_cu = 0;
            MG_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev_id));
            et_grid = (unsigned)(n_cu > 0 ? ET_WG_PER_CU * n_cu : ET_WG_PER_CU * 256);
            // (no more work-groups -- and slabs -- than (tile, strand) pairs: a tile that is closed before it is half full is closed by a
            // read that did not fit, by its 64th read or by its chunk's end)
            const uint64_t tiles_max = a.total / (MG_ET_CAP / 2) + nr / ET_MAXR + a.total / ((uint64_t)ET_CHUNK_TILES * MG_ET_CAP) + 2;
            if (2 * tiles_max < et_grid) et_grid = (unsigned)(2 * tiles_max);
            MG_TRY(gmg_pool_alloc((void **)&d_et_tiles, (nr + 1) * sizeof(MgTile)));
            MG_TRY(gmg_pool_alloc((void **)&d_et_slabs, (size_t)et_grid * 2 * et_qcap * sizeof(MgCall)));
            MG_TRY(gmg_pool_alloc((void **)&d_et_em, (size_t)et_grid * et_ecap * sizeof(EtEm)));
            // what leaves the tiles: the accepted ORFs' starts (measured: one per 52 bases of 454-like reads) or every start (one per 4)
            uint64_t want = (err_acc_only ? a.total / 16 : a.total / 3) + (1u << 20);
            if (gmg_opt(GMG_OPT_MG_ERR_TILE_Q) != 0) want = 64;         // (tests; -1: only this: the first pass finds the arrays too small)
            MG_TRY(alloc_staging(want));
        } else if (!err_wave) MG_TRY(alloc_level_scratch());
        if (err_wave) MG_TRY(gmg_pool_alloc((void **)&d_item_flag, 2 * nr + 64));
    }
    // k_mg_err_wave: as many one-wave work-groups per CU as their LDS shares allow (the grid strides over the (read, strand) pairs)
    const uint32_t ew_qcap = gmg_opt(GMG_OPT_MG_ERR_WAVE_Q) > 0 ? (uint32_t)gmg_opt(GMG_OPT_MG_ERR_WAVE_Q) : (uint32_t)EW_QCAP;
    auto launch_err_wave = [&](hipStream_t st, bool write) -> hipError_t {
        // Length classes, a launch each (a wave's LDS share is sized by its class: more waves per CU for the short reads): up to
        // 384 / 448 / 512 / 704 / EW_MAX_CAP bases.  Both passes on k_mg_err_wcount (breadth first, no walks); mg_err_wave = 2: the stack
        // walker (k_mg_err_wave) for both, 3: the stack walker as the write pass only (cross-checks), up to 512 / ew_cap bases.
        int n_cu = 0;
        hipError_t e = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev_id);
        if (e != hipSuccess) return e;
        const long long mode = gmg_opt(GMG_OPT_MG_ERR_WAVE);
        const bool wcount = mode == 1 || (mode == 3 && !write);
        const uint64_t n_blocks = wcount ? (2 * nr + EWC_ITEMS - 1) / EWC_ITEMS : (2 * nr + 63) / 64;
        uint32_t *st_ptr = tm.on && !write ? d_err_flag + 24 : (uint32_t *)nullptr;
        static const uint32_t bounds_c[6] = {0, 384, 448, 512, 704, EW_MAX_CAP}, bounds_w[3] = {0, 512, EW_MAX_CAP};
        const uint32_t *bounds = wcount ? bounds_c : bounds_w;
        const int n_cls = wcount ? 5 : 2;
        // the classes' launches go to streams of their own (forked from st, joined into it): a class's last work-groups run beside the
        // next class's first instead of holding the device for the launch's tail
        static thread_local hipStream_t cls_stream[16][4] = {};
        static thread_local hipEvent_t cls_fork[16] = {}, cls_done[16][4] = {};
        const bool forked = !tm.on && dev_id >= 0 && dev_id < 16 && !gmg_opt(GMG_OPT_MG_ONE_STREAM);
        if (forked) {
            if (!cls_fork[dev_id]) {
                e = hipEventCreateWithFlags(&cls_fork[dev_id], hipEventDisableTiming);
                for (int k = 0; k < 4 && e == hipSuccess; k++) {
                    e = hipStreamCreateWithFlags(&cls_stream[dev_id][k], hipStreamNonBlocking);
                    if (e == hipSuccess) e = hipEventCreateWithFlags(&cls_done[dev_id][k], hipEventDisableTiming);
                }
                if (e != hipSuccess) return e;
            }
            e = hipEventRecord(cls_fork[dev_id], st);
            if (e != hipSuccess) return e;
        }
        bool used[4] = {false, false, false, false};
        hipStream_t st0 = st;
        for (int cls = 0; cls < n_cls; cls++) {
            if (forked && cls > 0) {
                st = cls_stream[dev_id][cls - 1];
                if (!used[cls - 1]) { e = hipStreamWaitEvent(st, cls_fork[dev_id], 0); if (e != hipSuccess) return e; used[cls - 1] = true; }
            } else st = st0;
            const uint32_t lo = bounds[cls], hi = bounds[cls + 1] < ew_cap ? bounds[cls + 1] : ew_cap;
            if (hi <= lo || reads->max_len <= lo || reads->min_len > hi) continue;
            const uint32_t bytes = wcount ? ewc_layout(bounds[cls + 1], write, a.err_mode == 1).bytes : ew_layout(hi, ew_qcap, write).bytes;
            uint32_t per_cu = (uint32_t)((160u * 1024u) / (bytes + 1024u));
            if (per_cu > 16) per_cu = 16;
            if (per_cu < 1) per_cu = 1;
            uint64_t grid = (uint64_t)(n_cu > 0 ? n_cu : 256) * per_cu * 2;
            if (grid > n_blocks) grid = n_blocks;
            if (grid == 0) continue;
#define MG_EW_LAUNCH(W_, G_, K_) hipLaunchKernelGGL((k_mg_err_wave<W_, G_, K_>), dim3((unsigned)grid), dim3(EW_BLOCK), bytes, st, a, err_acc_only, lo, hi, ew_qcap, d_item_flag, st_ptr)
#define MG_EWC_LAUNCH_I(W_, G_, K_, I_) hipLaunchKernelGGL((k_mg_err_wcount<W_, G_, K_, I_>), dim3((unsigned)grid), dim3(EW_BLOCK), 0, st, a, err_acc_only, lo, hi, d_item_flag, st_ptr, \
                                                                d_err_flag + 32 + (write ? 8 : 0) + cls)
#define MG_EWC_LAUNCH(W_, G_, K_) do { if (a.err_mode == 1) MG_EWC_LAUNCH_I(W_, G_, K_, true); else MG_EWC_LAUNCH_I(W_, G_, K_, false); } while (0)
#define MG_EW_LAUNCH_K(W_, G_) do { if (cls == 0) MG_EW_LAUNCH(W_, G_, 8); else MG_EW_LAUNCH(W_, G_, 15); } while (0)
#define MG_EWC_LAUNCH_K(W_, G_) do { if (cls == 0) MG_EWC_LAUNCH(W_, G_, 6); else if (cls == 1) MG_EWC_LAUNCH(W_, G_, 7); else if (cls == 2) MG_EWC_LAUNCH(W_, G_, 8); else if (cls == 3) MG_EWC_LAUNCH(W_, G_, 11); else MG_EWC_LAUNCH(W_, G_, 15); } while (0)
            if (wcount && write) { if (a.gene32) MG_EWC_LAUNCH_K(true, true); else MG_EWC_LAUNCH_K(true, false); }
            else if (wcount) { if (a.gene32) MG_EWC_LAUNCH_K(false, true); else MG_EWC_LAUNCH_K(false, false); }
            else if (write) { if (a.gene32) MG_EW_LAUNCH_K(true, true); else MG_EW_LAUNCH_K(true, false); }
            else { if (a.gene32) MG_EW_LAUNCH_K(false, true); else MG_EW_LAUNCH_K(false, false); }
#undef MG_EWC_LAUNCH_K
#undef MG_EW_LAUNCH_K
#undef MG_EWC_LAUNCH
#undef MG_EWC_LAUNCH_I
#undef MG_EW_LAUNCH
            e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
        for (int k = 0; k < 4; k++)
            if (used[k]) {
                e = hipEventRecord(cls_done[dev_id][k], cls_stream[dev_id][k]);
                if (e == hipSuccess) e = hipStreamWaitEvent(st0, cls_done[dev_id][k], 0);
                if (e != hipSuccess) return e;
            }
        return hipSuccess;
    };
    const size_t et_lds = sizeof(EtLds<MG_ET_CAP>);
    auto launch_err_tile = [&](hipStream_t st) -> hipError_t {
        MgArgs at = a;                                  // the kernel's starts go to the staging arrays
        at.starts = d_st_s; at.errs = d_st_e; at.keys = d_st_k;
#define MG_ET_LAUNCH(G_)                                                                                                         \
        do {                                                                                                                     \
            hipError_t e_ = hipFuncSetAttribute((const void *)k_mg_err_tile<G_, MG_ET_CAP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)et_lds); \
            if (e_ != hipSuccess) return e_;                                                                                     \
            hipLaunchKernelGGL((k_mg_err_tile<G_, MG_ET_CAP>), dim3(et_grid), dim3(ET_BLOCK), et_lds, st, at, d_et_tiles, d_et_ntiles, d_et_items, \
                               d_et_slabs, et_qcap, d_et_em, et_ecap, d_et_stage_ctr, (unsigned long long)et_stage_cap, err_acc_only); \
        } while (0)
        if (a.gene32) MG_ET_LAUNCH(true); else MG_ET_LAUNCH(false);
#undef MG_ET_LAUNCH
        return hipGetLastError();
    };
    // (the wave kernels' zeroed arrays and the reads' fit flags need nothing of the six-frame table: on a side stream, beside the
    // partial-window pass and the ORF write pass -- they sat 0.3 ms between those and the count pass)
    bool wave_reset_done = false;
    if (no && err_mode && err_path == 0 && err_wave && s3 != s) {      // (the second side stream: the first one is busy with the ORF write pass)
        MG_TRY(hipMemsetAsync(d_acc_bits, 0, (no / 32 + 1) * 4, s3));
        MG_TRY(hipMemsetAsync(d_item_flag, 0, 2 * nr + 64, s3));
        hipLaunchKernelGGL(k_mg_err_prepare, dim3(grid_for(nr)), dim3(256), 0, s3, a, (uint64_t)ew_cap + 1);
        MG_TRY(hipGetLastError());
        wave_reset_done = true;
    }
    if (err_mode && s2 != s) {                          // the error branch needs the six-frame table from here on: one stream again
        MG_TRY(hipEventRecord(side_done, s2));
        MG_TRY(hipStreamWaitEvent(s, side_done, 0));
        if (s3 != s2) {
            MG_TRY(hipEventRecord(side2_done, s3));
            MG_TRY(hipStreamWaitEvent(s, side2_done, 0));
        }
        s2 = s;
    }
    int level_tries = 0;
    for (int attempt = 0; attempt < 6; attempt++) {
    const dim3 lvl_grid(256 * 16);
    const bool any_unfit = reads->max_len >= (err_wave ? (uint64_t)ew_cap + 1 : err_tile ? (uint64_t)MG_ET_CAP + 1 : 2040);
    if (no && err_mode && err_path == 0 && err_wave) {
        if (!wave_reset_done) {
            MG_TRY(hipMemsetAsync(d_acc_bits, 0, (no / 32 + 1) * 4, s2));
            MG_TRY(hipMemsetAsync(d_item_flag, 0, 2 * nr + 64, s2));
            hipLaunchKernelGGL(k_mg_err_prepare, dim3(grid_for(nr)), dim3(256), 0, s2, a, (uint64_t)ew_cap + 1);
        }
        wave_reset_done = false;                        // (a repeat of the call starts from zeroed arrays again)
        MG_TRY(launch_err_wave(s2, false));
        if (any_unfit) hipLaunchKernelGGL(k_mg_err_flat<false>, dim3(grid_for(no)), dim3(MG_ERR_BLOCK), 0, s2, a, err_acc_only, 1);
    } else if (no && err_mode && err_path == 0 && err_tile) {
        MG_TRY(hipMemsetAsync(d_acc_bits, 0, (no / 32 + 1) * 4, s2));
        const uint64_t chunk = (uint64_t)ET_CHUNK_TILES * MG_ET_CAP, n_chunks = a.total / chunk + 1;
        hipLaunchKernelGGL(k_et_tiles, dim3(grid_for(n_chunks)), dim3(256), 0, s2, a, (uint32_t)MG_ET_CAP, (uint32_t)MG_ET_CAP, chunk, n_chunks,
                           d_et_tiles, d_et_ntiles, d_read_fit);
        MG_TRY(launch_err_tile(s2));
        if (any_unfit) hipLaunchKernelGGL(k_mg_err_flat<false>, dim3(grid_for(no)), dim3(MG_ERR_BLOCK), 0, s2, a, err_acc_only, 1);
    } else if (no && err_mode && err_path == 0) {
        MG_TRY(hipMemsetAsync(d_fill, 0, no * 4, s2));
        MG_TRY(hipMemsetAsync(d_acc_bits, 0, (no / 32 + 1) * 4, s2));
        hipLaunchKernelGGL(k_mg_err_prepare, dim3(grid_for(nr)), dim3(256), 0, s2, a, (uint64_t)2040);
        if (a.pfx) hipLaunchKernelGGL((k_mg_err_level<false, 0, true>), dim3(grid_for(no)), dim3(256), 0, s2, a, err_acc_only);
        else hipLaunchKernelGGL((k_mg_err_level<false, 0, false>), dim3(grid_for(no)), dim3(256), 0, s2, a, err_acc_only);
        if (a.pfx) hipLaunchKernelGGL((k_mg_err_level<false, 1, true>), lvl_grid, dim3(256), 0, s2, a, err_acc_only);
        else hipLaunchKernelGGL((k_mg_err_level<false, 1, false>), lvl_grid, dim3(256), 0, s2, a, err_acc_only);
        if (a.pfx) hipLaunchKernelGGL((k_mg_err_level<false, 2, true>), lvl_grid, dim3(256), 0, s2, a, err_acc_only);
        else hipLaunchKernelGGL((k_mg_err_level<false, 2, false>), lvl_grid, dim3(256), 0, s2, a, err_acc_only);
        hipLaunchKernelGGL(k_mg_err_verdict, dim3(grid_for(no)), dim3(256), 0, s2, a, err_acc_only, any_unfit ? 0 : 1);
        if (any_unfit) hipLaunchKernelGGL(k_mg_err_flat<false>, dim3(grid_for(no)), dim3(MG_ERR_BLOCK), 0, s2, a, err_acc_only, 1);
    } else if (no && err_mode) hipLaunchKernelGGL(k_mg_err_flat<false>, dim3(grid_for(no)), dim3(MG_ERR_BLOCK), 0, s2, a, err_acc_only, 0);
    else if (no && !a.count_starts) hipLaunchKernelGGL(k_mg_starts<false>, dim3(grid_for(no)), dim3(256), 0, s2, a);
    MG_TRY(hipGetLastError());
    tm.lap("start lists: count");
    rc = mg_scan(d_orf_cnt, d_start_off, no, &res->n_starts, s2);
    if (rc) return fail(rc);
    if (no && err_mode && err_path == 0) {              // did a call array overflow?  (mg_scan has synchronised the stream)
        uint32_t st[32];
        MG_TRY(hipMemcpy(st, d_err_flag, 128, hipMemcpyDeviceToHost));
        if (tm.on && err_tile) fprintf(stderr, "[gmg_mg] k_mg_err_tile: %u tiles (%llu ORFs)\n", st[18], (unsigned long long)no);
        if (tm.on && err_wave)
            fprintf(stderr, "[gmg_mg] k_mg_err_wave: flag %u, deepest stack %u, most ORFs on a strand %u, %u trips and %u calls in %u (read, strand) pairs\n",
                    st[0], st[24], st[25], st[26], st[27], st[28]);
        if (!err_tile && !err_wave && !st[0] && a.total) {   // what this batch needed, for the next call's arrays
            unsigned long long handed[2];
            memcpy(handed, st + 2, 16);
            calls_per_base_hint = (double)(handed[0] > handed[1] ? handed[0] : handed[1]) / (double)a.total;
        }
        if (tm.on && !err_tile && !err_wave) {          // (mg_timing) how many calls the levels handed on
            unsigned long long handed[2];
            memcpy(handed, st + 2, 16);
            fprintf(stderr, "[gmg_mg] calls handed to level 1: %llu, to level 2: %llu (capacity %llu each; %llu ORFs)\n", handed[0], handed[1],
                    (unsigned long long)a.call_cap, (unsigned long long)no);
        }
        if (!(st[0] & 1u) && (st[0] & 2u) && err_tile) {
            // the staging arrays were too small: once more with what the kernel asked for (every batch of ORFs has added its wish)
            unsigned long long asked = 0;
            memcpy(&asked, st + 22, 8);
            MG_TRY(alloc_staging(asked + 1024));
            MG_TRY(hipMemsetAsync(d_err_flag, 0, 256, s2));
            MG_TRY(hipMemsetAsync(d_orf_cnt, 0, (no + 1) * 4, s2));
            continue;
        }
        if (st[0] && (err_tile || err_wave)) {
            // a work-group's call slab / a wave's call stack was full: the batch repeats on the level kernels (their call arrays grow with the batch)
            err_tile = err_wave = false;
            if (a.q454) { a.q454 = 0; MG_TRY(build_qualities(s2)); }
            if (a.pfx) MG_TRY(build_run_tables(s2));
            MG_TRY(build_walk_rows(s2));
            MG_TRY(alloc_level_scratch());
            MG_TRY(hipMemsetAsync(d_err_flag, 0, 256, s2));
            MG_TRY(hipMemsetAsync(d_orf_cnt, 0, (no + 1) * 4, s2));
            continue;
        }
        if (st[0]) {
            // once more with arrays of twice what was asked for (level 2 is only partly known when level 1 overflows); if that is
            // not enough either, or does not fit, everything runs on the per-ORF kernel
            unsigned long long asked[2];
            memcpy(asked, st + 2, 16);
            const uint64_t want = 2 * (asked[0] > asked[1] ? asked[0] : asked[1]) + 65536;
            gmg_pool_release(d_calls[0]); gmg_pool_release(d_calls[1]);
            d_calls[0] = d_calls[1] = nullptr;
            const bool may_grow = level_tries == 0 && want <= 8 * a.total + 65536 && (gmg_opt(GMG_OPT_MG_ERR_CALLS) <= 0 || gmg_opt(GMG_OPT_MG_ERR_CALLS_GROW));
            level_tries++;
            bool grown = false;
            if (may_grow && gmg_pool_alloc((void **)&d_calls[0], want * sizeof(MgCall)) == hipSuccess) {
                if (gmg_pool_alloc((void **)&d_calls[1], want * sizeof(MgCall)) == hipSuccess) grown = true;
                else { gmg_pool_release(d_calls[0]); d_calls[0] = nullptr; }
            }
            if (grown) { a.call_cap = want; a.calls[0] = d_calls[0]; a.calls[1] = d_calls[1]; }
            else {
                err_path = 1;
                a.acc_bits = nullptr;                   // (the verdicts come from the per-ORF kernel from here on: no bitmap)
                if (a.gene32 && !a.fs) {                // the per-ORF kernel walks the table itself: make it now
                    const uint64_t fstride = (a.total + 15) & ~15ull;
                    MG_TRY(gmg_pool_alloc((void **)&d_fs_own, (size_t)6 * fstride * sizeof(double)));
                    hipLaunchKernelGGL(k_mg_apply_nulls, dim3(grid_for(a.total)), dim3(256), 0, s2, a, d_fs_own, fstride);
                    MG_TRY(hipGetLastError());
                    MG_TRY(hipStreamSynchronize(s2));
                    a.fs = d_fs_own;
                    a.fs_stride = fstride;
                    a.gene32 = nullptr;
                }
            }
            MG_TRY(hipMemsetAsync(d_err_flag, 0, 256, s2));
            MG_TRY(hipMemsetAsync(d_orf_cnt, 0, (no + 1) * 4, s2));
            continue;
        }
    }
    if (res->n_starts > (uint64_t)gmg_opt(GMG_OPT_MG_MAX_ENTRIES))
        return fail(gmg_set_error(GMG_ETOOBIG, "gmg_mg_score_reads: %llu starts in one batch, gmg_mg_orf.start_begin holds %lld: split the batch",
                                  (unsigned long long)res->n_starts, gmg_opt(GMG_OPT_MG_MAX_ENTRIES)));
    MG_TRY(gmg_pool_alloc((void **)&res->d_starts, (res->n_starts ? res->n_starts : 1) * sizeof(gmg_start)));
    a.start_off = d_start_off;
    a.starts = res->d_starts;
    if (err_mode) {
        MG_TRY(gmg_pool_alloc((void **)&res->d_errs, (res->n_starts ? res->n_starts : 1) * sizeof(gmg_start_errors)));
        a.errs = res->d_errs;
        if (err_path == 0) {
            MG_TRY(gmg_pool_alloc((void **)&d_keys, (res->n_starts ? res->n_starts : 1) * 8));
            a.keys = d_keys;
        }
    }
    if (s2 != s) {                                      // (mg_scan has synchronised the side stream already; the event keeps
        MG_TRY(hipEventRecord(side_done, s2));          //  the ordering explicit)
        MG_TRY(hipStreamWaitEvent(s, side_done, 0));
    }
    if (no && err_mode && err_path == 0 && err_wave) {
        hipLaunchKernelGGL(k_mg_err_begin, dim3(grid_for(no)), dim3(256), 0, s, a, err_acc_only);
        MG_TRY(launch_err_wave(s, true));
        if (any_unfit) hipLaunchKernelGGL(k_mg_err_flat<true>, dim3(grid_for(no)), dim3(MG_ERR_BLOCK), 0, s, a, err_acc_only, 1);
    } else if (no && err_mode && err_path == 0 && err_tile) {
        hipLaunchKernelGGL(k_et_unstage, dim3(grid_for(no)), dim3(256), 0, s, a, d_st_s, d_st_e, d_st_k, err_acc_only);
        if (any_unfit) hipLaunchKernelGGL(k_mg_err_flat<true>, dim3(grid_for(no)), dim3(MG_ERR_BLOCK), 0, s, a, err_acc_only, 1);
    } else if (no && err_mode && err_path == 0) {
        hipLaunchKernelGGL(k_mg_err_begin, dim3(grid_for(no)), dim3(256), 0, s, a, err_acc_only);
        if (a.pfx) hipLaunchKernelGGL((k_mg_err_level<true, 0, true>), dim3(grid_for(no)), dim3(256), 0, s, a, err_acc_only);
        else hipLaunchKernelGGL((k_mg_err_level<true, 0, false>), dim3(grid_for(no)), dim3(256), 0, s, a, err_acc_only);
        if (a.pfx) hipLaunchKernelGGL((k_mg_err_level<true, 1, true>), lvl_grid, dim3(256), 0, s, a, err_acc_only);
        else hipLaunchKernelGGL((k_mg_err_level<true, 1, false>), lvl_grid, dim3(256), 0, s, a, err_acc_only);
        if (a.pfx) hipLaunchKernelGGL((k_mg_err_level<true, 2, true>), lvl_grid, dim3(256), 0, s, a, err_acc_only);
        else hipLaunchKernelGGL((k_mg_err_level<true, 2, false>), lvl_grid, dim3(256), 0, s, a, err_acc_only);
        if (any_unfit) hipLaunchKernelGGL(k_mg_err_flat<true>, dim3(grid_for(no)), dim3(MG_ERR_BLOCK), 0, s, a, err_acc_only, 1);
    } else if (no && err_mode) hipLaunchKernelGGL(k_mg_err_flat<true>, dim3(grid_for(no)), dim3(MG_ERR_BLOCK), 0, s, a, err_acc_only, 0);
    else if (no && fused_nw) {
        const unsigned grid = (unsigned)(2 * a.n_tiles < 64 * 1024 ? 2 * a.n_tiles : 64 * 1024);     // (even: a work-group keeps its strand)
        // the few reads no tile takes (one lane per read, long loops): beside the tile kernel on the side stream, behind their
        // running sums (k_mg_cum, queued on the caller's stream long ago)
        const bool unfit_aside = fused_rest && s2 != s;
        if (unfit_aside) {
            MG_TRY(hipEventRecord(cum_done, s));
            MG_TRY(hipStreamWaitEvent(s2, cum_done, 0));
            hipLaunchKernelGGL(k_mg_starts_unfit, dim3(grid_for(a.n_reads / 16 + 1)), dim3(256), 0, s2, a);
            MG_TRY(hipEventRecord(side_done, s2));
        }
#define MG_LAUNCH_TILE(NW_, G_, EL_) do { if (G_ && a.read_null && NW_ == 2 && EL_ == 8 && fused_nc2) hipLaunchKernelGGL((k_mg_tile_starts<2, G_, 8, G_, 2>), dim3(grid), dim3(128), 0, s, a); \
                                          else if (G_ && a.read_null) hipLaunchKernelGGL((k_mg_tile_starts<NW_, G_, EL_, G_>), dim3(grid), dim3(64 * NW_), 0, s, a); \
                                          else hipLaunchKernelGGL((k_mg_tile_starts<NW_, G_, EL_, false>), dim3(grid), dim3(64 * NW_), 0, s, a); } while (0)
#define MG_LAUNCH_TILE_EL(NW_, G_) do { if (fused_el == 8) MG_LAUNCH_TILE(NW_, G_, 8); else MG_LAUNCH_TILE(NW_, G_, 9); } while (0)
        if (a.gene32) {
            if (fused_nw == 1) MG_LAUNCH_TILE_EL(1, true); else if (fused_nw == 2) MG_LAUNCH_TILE_EL(2, true); else MG_LAUNCH_TILE_EL(4, true);
        } else {
            if (fused_nw == 1) MG_LAUNCH_TILE_EL(1, false); else if (fused_nw == 2) MG_LAUNCH_TILE_EL(2, false); else MG_LAUNCH_TILE_EL(4, false);
        }
#undef MG_LAUNCH_TILE_EL
#undef MG_LAUNCH_TILE
        if (unfit_aside) MG_TRY(hipStreamWaitEvent(s, side_done, 0));
        else if (fused_rest) hipLaunchKernelGGL(k_mg_starts_unfit, dim3(grid_for(a.n_reads / 16 + 1)), dim3(256), 0, s, a);
    } else if (no) hipLaunchKernelGGL(k_mg_starts<true>, dim3(grid_for(no)), dim3(256), 0, s, a);
    MG_TRY(hipGetLastError());
    break;
    }
    }
    if (!find_only && (prm->flags & GMG_MG_ACCEPTED_ONLY)) {
        // 4. only what Add_Events_* will see leaves the GPU: two prefix sums over the accepted flags, one gather
        uint32_t *d_keep = nullptr, *d_keep_st = nullptr;
        uint64_t *d_new_orf = nullptr, *d_new_st = nullptr, *d_new_first = nullptr;
        gmg_mg_orf *d_orfs2 = nullptr;
        gmg_start *d_starts2 = nullptr;
        uint64_t n_keep = 0, n_keep_st = 0;
        // Error branch: the write passes put an ORF's starts at the scan of the counts, and a rejected ORF counts 0 -- the start,
        // error and key arrays hold the accepted ORFs' lists alone, in order: packed already.  Only the records move.
        const bool starts_packed = err_mode != 0;
        hipError_t e = gmg_pool_alloc((void **)&d_keep, (no + 1) * 4);
        if (e == hipSuccess && !starts_packed) e = gmg_pool_alloc((void **)&d_keep_st, (no + 1) * 4);
        if (e == hipSuccess) e = gmg_pool_alloc((void **)&d_new_orf, (no + 1) * 8);
        if (e == hipSuccess && !starts_packed) e = gmg_pool_alloc((void **)&d_new_st, (no + 1) * 8);
        if (e == hipSuccess) e = gmg_pool_alloc((void **)&d_new_first, (nr + 1) * 8);
        if (e == hipSuccess) e = hipMemsetAsync(d_keep + no, 0, 4, s);
        if (e == hipSuccess && !starts_packed) e = hipMemsetAsync(d_keep_st + no, 0, 4, s);
        int rc2 = GMG_OK;
        // (error branch: the bitmap of the accepted ORFs is complete unless everything went to the per-ORF kernel)
        const uint32_t *kept_bits = (err_mode && err_path == 0) ? d_acc_bits : nullptr;
        if (e == hipSuccess && starts_packed && kept_bits && no) {
            const uint64_t n_words = no / 32 + 1;
            uint32_t *d_wc = nullptr, n_keep32 = 0;       // [n_words + 1] set bits per word, then [n_words + 1] their exclusive sums
            e = gmg_pool_alloc((void **)&d_wc, 2 * (n_words + 4) * 4);
            uint32_t *d_wo = d_wc ? d_wc + ((n_words + 4) & ~3ull) : nullptr;
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_mg_keep_words, dim3(grid_for(n_words + 1)), dim3(256), 0, s, kept_bits, n_words, d_wc);
                e = gmg_scan_excl<uint32_t, uint32_t>(d_wc, d_wo, n_words + 1, s);
            }
            if (e == hipSuccess) e = hipMemcpyAsync(&n_keep32, d_wo + n_words, 4, hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            n_keep = n_keep32;
            n_keep_st = res->n_starts;
            if (e == hipSuccess) e = gmg_pool_alloc((void **)&d_orfs2, (n_keep ? n_keep : 1) * sizeof(gmg_mg_orf));
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_mg_keep_gather_bits, dim3(grid_for(n_words)), dim3(256), 0, s, res->d_orfs, kept_bits, d_wo, n_words, no, d_orfs2);
                hipLaunchKernelGGL(k_mg_keep_reads_bits, dim3(grid_for(nr + 1)), dim3(256), 0, s, res->d_read_orf_off, nr, kept_bits, d_wo, d_new_first);
                e = hipGetLastError();
                if (e == hipSuccess) e = hipStreamSynchronize(s);
            }
            if (d_wc) gmg_pool_release(d_wc);
        } else if (e == hipSuccess) {
            if (no) hipLaunchKernelGGL(k_mg_keep_counts, dim3(grid_for(no)), dim3(256), 0, s, res->d_orfs, kept_bits, no, d_keep, d_keep_st);
            rc2 = mg_scan(d_keep, d_new_orf, no, &n_keep, s);
            if (starts_packed) n_keep_st = res->n_starts;
            else if (!rc2) rc2 = mg_scan(d_keep_st, d_new_st, no, &n_keep_st, s);
            if (!rc2) e = gmg_pool_alloc((void **)&d_orfs2, (n_keep ? n_keep : 1) * sizeof(gmg_mg_orf));
            if (!rc2 && e == hipSuccess && !starts_packed) e = gmg_pool_alloc((void **)&d_starts2, (n_keep_st ? n_keep_st : 1) * sizeof(gmg_start));
            if (!rc2 && e == hipSuccess) {
                if (no) hipLaunchKernelGGL(k_mg_keep_gather, dim3(grid_for(no)), dim3(256), 0, s, res->d_orfs, kept_bits, res->d_starts, no, d_new_orf,
                                           starts_packed ? (const uint64_t *)nullptr : d_new_st, d_orfs2, d_starts2);
                hipLaunchKernelGGL(k_mg_keep_reads, dim3(grid_for(nr + 1)), dim3(256), 0, s, res->d_read_orf_off, nr, d_new_orf, d_new_first);
                e = hipGetLastError();
                if (e == hipSuccess) e = hipStreamSynchronize(s);
            }
        }
        if (d_keep) gmg_pool_release(d_keep);
        if (d_keep_st) gmg_pool_release(d_keep_st);
        if (d_new_orf) gmg_pool_release(d_new_orf);
        if (d_new_st) gmg_pool_release(d_new_st);
        if (rc2 || e != hipSuccess) {
            if (d_orfs2) gmg_pool_release(d_orfs2);
            if (d_starts2) gmg_pool_release(d_starts2);
            if (d_new_first) gmg_pool_release(d_new_first);
            return fail(rc2 ? rc2 : gmg_set_error(GMG_EHIP, "gmg_mg_score_reads: packing the accepted ORFs: %s", hipGetErrorString(e)));
        }
        gmg_pool_release(res->d_orfs);
        gmg_pool_release(res->d_read_orf_off);
        if (!starts_packed) { gmg_pool_release(res->d_starts); res->d_starts = d_starts2; }     // (else: starts, errors and keys stay where they are)
        res->d_orfs = d_orfs2;
        res->d_read_orf_off = d_new_first;
        res->n_orfs = n_keep;
        res->n_starts = n_keep_st;
    }
    if (d_keys && res->n_starts) {
        // 5. error branch: every ORF's slice of the start array into the reference's push order (k_mg_order_starts)
        const uint64_t ns = res->n_starts, nseg = res->n_orfs;
        gmg_start *d_starts3 = nullptr;
        gmg_start_errors *d_errs3 = nullptr;
        hipError_t e = gmg_pool_alloc((void **)&d_starts3, ns * sizeof(gmg_start));
        if (e == hipSuccess) e = gmg_pool_alloc((void **)&d_errs3, ns * sizeof(gmg_start_errors));
        if (e == hipSuccess) {
            const uint64_t blocks = (nseg + 3) / 4;
            hipLaunchKernelGGL(k_mg_order_starts, dim3((unsigned)(blocks < 256 * 32 ? blocks : 256 * 32)), dim3(256), 0, s, res->d_orfs, nseg, d_keys, res->d_starts,
                               res->d_errs, d_starts3, d_errs3);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) {
            if (d_starts3) gmg_pool_release(d_starts3);
            if (d_errs3) gmg_pool_release(d_errs3);
            return fail(gmg_set_error(e == hipErrorOutOfMemory ? GMG_ENOMEM : GMG_EHIP, "gmg_mg_score_reads: ordering the start lists: %s", hipGetErrorString(e)));
        }
        gmg_pool_release(res->d_starts);
        gmg_pool_release(res->d_errs);
        res->d_starts = d_starts3;
        res->d_errs = d_errs3;
        tm.lap("start lists: push order");
    }
    MG_TRY(hipStreamSynchronize(s));
    tm.lap("start lists");
    if (err_wave && res->n_orfs && d_err_flag) {        // did a wave's stack overflow in the WRITE pass?  (the count pass was checked behind its scan)
        uint32_t flag = 0;
        MG_TRY(hipMemcpy(&flag, d_err_flag, 4, hipMemcpyDeviceToHost));
        if (flag) return fail(MG_RETRY_NO_WAVE);
    }
#undef MG_TRY
    if (d_fs_own) gmg_pool_release(d_fs_own);
    if (d_gene32) gmg_pool_release(d_gene32);
    if (d_read_null) gmg_pool_release(d_read_null);
    if (d_read_isl) gmg_pool_release(d_read_isl);
    gmg_pool_release(d_read_cnt);
    if (d_orf_cnt) gmg_pool_release(d_orf_cnt);
    if (d_start_off) gmg_pool_release(d_start_off);
    if (d_cum) gmg_pool_release(d_cum);
    if (d_tiles) gmg_pool_release(d_tiles);
    if (d_ntiles) gmg_pool_release(d_ntiles);
    if (d_all) gmg_pool_release(d_all);
    if (d_sel_tmp) gmg_pool_release(d_sel_tmp);
    if (d_unfit) gmg_pool_release(d_unfit);
    if (d_qual) gmg_pool_release(d_qual);
    if (d_user_q) gmg_pool_release(d_user_q);
    if (d_pen) gmg_pool_release(d_pen);
    if (d_read_fit) gmg_pool_release(d_read_fit);
    if (d_keys) gmg_pool_release(d_keys);
    if (d_err_flag) gmg_pool_release(d_err_flag);
    if (d_fill) gmg_pool_release(d_fill);
    if (d_acc_bits) gmg_pool_release(d_acc_bits);
    if (d_calls[0]) gmg_pool_release(d_calls[0]);
    if (d_calls[1]) gmg_pool_release(d_calls[1]);
    if (d_agg) gmg_pool_release(d_agg);
    if (d_walk) gmg_pool_release(d_walk);
    if (d_walk_q) gmg_pool_release(d_walk_q);
    if (d_run) gmg_pool_release(d_run);
    if (d_item_flag) gmg_pool_release(d_item_flag);
    if (d_et_tiles) gmg_pool_release(d_et_tiles);
    if (d_et_slabs) gmg_pool_release(d_et_slabs);
    if (d_et_em) gmg_pool_release(d_et_em);
    if (d_st_s) gmg_pool_release(d_st_s);
    if (d_st_e) gmg_pool_release(d_st_e);
    if (d_st_k) gmg_pool_release(d_st_k);
    tm.lap("free scratch");
    *out = res;
    return GMG_OK;
}

extern "C" int gmg_mg_score_reads(const gmg_model *gene, const gmg_model *nul, const gmg_reads *reads,
                                  const gmg_mg_params *prm, double *d_frame_scores, gmg_mg_result **out, void *stream)
{
    return mg_run(gene, nul, reads, prm, d_frame_scores, out, stream, false);
}

// glimmer-mg's classification mode: one call for a batch whose reads come in consecutive groups, every group under its own gene
// ICM (the loop over ICM_Sequences, glimmer-mg.cc:361-451), the null model and Ignore_Score_Len per read as in gmg_mg_score_reads
extern "C" int gmg_mg_score_groups(const gmg_mg_group *groups, int n_groups, const gmg_model *nul, const gmg_reads *reads,
                                   const gmg_mg_params *prm, gmg_mg_result **out, void *stream)
{
    if (!groups || n_groups < 1 || !reads || !prm) return gmg_set_error(GMG_EINVAL, "gmg_mg_score_groups: NULL argument");
    if (n_groups >= 1 << 27) return gmg_set_error(GMG_EINVAL, "gmg_mg_score_groups: at most 2^27 - 1 groups per call");
    if (!prm->nulls) return gmg_set_error(GMG_EINVAL, "gmg_mg_score_groups: needs the per-read null models (gmg_mg_params.nulls / read_null)");
    MgGroups g;
    g.n = n_groups;
    uint64_t next = 0;
    for (int k = 0; k < n_groups; k++) {
        if (!groups[k].gene || groups[k].read_begin != next || groups[k].read_end < groups[k].read_begin)
            return gmg_set_error(GMG_EINVAL, "gmg_mg_score_groups: group %d has no model, or the groups are not consecutive read ranges from 0", k);
        if (groups[k].gene->dev.P != 3) return gmg_set_error(GMG_EBADMODEL, "gmg_mg_score_groups: Score_All_Frames needs models of periodicity 3");
        g.models.push_back(groups[k].gene);
        g.read_begin.push_back(next);
        next = groups[k].read_end;
    }
    if (next != reads->n_reads) return gmg_set_error(GMG_EINVAL, "gmg_mg_score_groups: the groups end at read %llu of %llu",
                                                     (unsigned long long)next, (unsigned long long)reads->n_reads);
    g.read_begin.push_back(next);
    return mg_run(groups[0].gene, nul, reads, prm, nullptr, out, stream, false, &g);
}

extern "C" int gmg_find_orfs(const gmg_reads *reads, const gmg_mg_params *prm, gmg_mg_result **out, void *stream)
{
    return mg_run(nullptr, nullptr, reads, prm, nullptr, out, stream, true);
}

extern "C" int gmg_mg_result_info(const gmg_mg_result *r, uint64_t *n_orfs, uint64_t *n_starts)
{
    if (!r) return gmg_set_error(GMG_EINVAL, "gmg_mg_result_info: NULL result");
    if (n_orfs) *n_orfs = r->n_orfs;
    if (n_starts) *n_starts = r->n_starts;
    return GMG_OK;
}

extern "C" int gmg_mg_result_fetch_errors(const gmg_mg_result *r, gmg_start_errors *errs)
{
    { int rc_enter = gmg_enter("gmg_mg_result_fetch_errors"); if (rc_enter) return rc_enter; }
    if (!r || (r->n_starts && !errs)) return gmg_set_error(GMG_EINVAL, "gmg_mg_result_fetch_errors: NULL argument");
    if (!r->n_starts) return GMG_OK;
    if (!r->d_errs) { memset(errs, 0, r->n_starts * sizeof(gmg_start_errors)); return GMG_OK; }
    GMG_HIP(hipMemcpy(errs, r->d_errs, r->n_starts * sizeof(gmg_start_errors), hipMemcpyDeviceToHost));
    return GMG_OK;
}

extern "C" int gmg_mg_result_fetch(const gmg_mg_result *r, gmg_mg_orf *orfs, gmg_start *starts, uint64_t *read_orf_off)
{
    return gmg_mg_result_fetch_on(r, orfs, starts, read_orf_off, nullptr);
}

extern "C" int gmg_mg_result_fetch_on(const gmg_mg_result *r, gmg_mg_orf *orfs, gmg_start *starts, uint64_t *read_orf_off,
                                      void *stream)
{
    { int rc_enter = gmg_enter("gmg_mg_result_fetch_on"); if (rc_enter) return rc_enter; }
    if (!r || (r->n_orfs && !orfs) || (r->n_starts && !starts)) return gmg_set_error(GMG_EINVAL, "gmg_mg_result_fetch: NULL argument");
    hipStream_t s = (hipStream_t)stream;
    if (r->n_orfs) GMG_HIP(hipMemcpyAsync(orfs, r->d_orfs, r->n_orfs * sizeof(gmg_mg_orf), hipMemcpyDeviceToHost, s));
    if (r->n_starts) GMG_HIP(hipMemcpyAsync(starts, r->d_starts, r->n_starts * sizeof(gmg_start), hipMemcpyDeviceToHost, s));
    if (read_orf_off) GMG_HIP(hipMemcpyAsync(read_orf_off, r->d_read_orf_off, (r->n_reads + 1) * 8, hipMemcpyDeviceToHost, s));
    GMG_HIP(hipStreamSynchronize(s));
    return GMG_OK;
}
