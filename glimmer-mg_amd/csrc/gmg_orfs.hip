// gmg_orfs.hip -- the scoring part of glimmer3's Score_Orfs (src/Glimmer/glimmer3.cc:1275-1552) for a
// whole batch of ORFs: per ORF the reversed / complemented buffer (glimmer3.cc:1322-1343), the gene and
// null Cumulative_Score from frame 1 (:1346-1347; k_seg_cum, sequential double adds), then k_orf_scan:
// the start-codon scan from the 3' end (:1355-1421), first / best start, the Ignore_Score_Len boost
// (:1464-1466), the tentative-gene test (:1468) and the gene score (:1489).  Only the compact start
// lists leave the GPU; events, DP and trace-back stay host code (src/Glimmer/glimmer_base.cc).

#include "gmg_device.h"

#include "gmg_scan.h"

#include <float.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <vector>

struct OrfTmp { double s; int32_t which, pad; };   // one in-frame position of the ascending pass

struct gmg_orf_batch {
    double *d_walk;              // events path: running sums in walking order, [2][total_bases] (allocated on first use)
    uint32_t *d_heads;           // ... and one bit per base and strand: an ORF asks for the sum there (its HEAD position), [2][total_bases / 32 + 4]
    uint16_t *d_qpre;            // ... compact form: per lane of every unit the values in front of its own, [2][total_bases / 8 + n_reads + 64]
    uint8_t *d_qneed;            // ... and its need bits
    float *d_gene6;              // fused / events path: per-base gene values, [6][total_bases] (allocated on first use)
    OrfTmp *d_tmp;               // fused path: one slot per in-frame position, same offsets as the start lists
    gmg_orf *d_orfs;
    gmg_segments *segs;          // the ORF buffers as segments (REVERSED / COMPLEMENTED)
    uint64_t *d_start_off;       // [n+1] first slot of each ORF's start list
    double *d_score, *d_indep;   // cumulative scores, segs->total_len each
    gmg_orf_result *d_results;
    gmg_start *d_starts;         // one region of orf_len/3+2 slots per ORF, filled from its front
    uint32_t *d_nst, *d_coff;    // [n+1] starts per ORF and their exclusive prefix sum (compaction)
    void *d_scan_tmp;
    size_t scan_tmp_bytes;
    gmg_start *d_compact;        // the used slots back to back: what leaves the GPU (grown on demand)
    uint64_t compact_cap;
    uint64_t n, max_starts;
    uint64_t n_packed;           // starts of the last gmg_score_orfs_begin, back to back in d_compact
    int scored;
    const gmg_reads *reads;      // the batch the ORFs were validated against (gmg_orfs_upload): gmg_score_orfs takes no other
    uint64_t reads_total;
};

struct OrfScanArgs {
    const uint32_t *packed;
    const uint64_t *read_off;
    const gmg_orf *orfs;
    const uint64_t *cum_off;     // segment output offsets (exclusive prefix of orf_len)
    const uint64_t *start_off;
    const double *score, *indep;
    gmg_orf_result *results;
    gmg_start *starts;
    uint32_t *nst;
    uint64_t n;
    int min_gene_len, allow_truncated, use_first_start, ignore_score_len;
    double start_threshold;
    int n_pat;
    uint32_t pat[8];             // Codon_t patterns of the start codons, 4 bits per base (gene.cc:62-75)
};

// Ch_Mask (src/Common/gene.cc:954-995): one bit per base an IUPAC letter can stand for
static unsigned ch_mask(int ch)
{
    switch (ch | 0x20) {
    case 'a': return 0x1; case 'c': return 0x2; case 'g': return 0x4; case 't': return 0x8;
    case 'r': return 0x5; case 'y': return 0xA; case 's': return 0x6; case 'w': return 0x9;
    case 'm': return 0x3; case 'k': return 0xC; case 'b': return 0xE; case 'd': return 0xD;
    case 'h': return 0xB; case 'v': return 0x7; case 'n': return 0xF;
    }
    return 0;
}

// ---------------------------------------------------------------------------
// Fused path (default 12/7/3 gene model + width-3 null model): no per-base scratch at all.
//   k_frame6t<GENE_ONLY> (gmg_frame6.hip) first writes the gene model's per-base values of the whole
//   batch as fp32 rows [6][total] at the six-frame kernel's speed; buffer positions j >= W-1 of any ORF
//   see exactly the whole-read window, so their value is row (1+j)%3 (+3 for reverse ORFs) at that base.
//   k_orf_fused, one lane per ORF, walks the buffer ONCE from j = 0: the first W-1 values by the
//   partial-window descent (shift tables in LDS, row from crow), the rest from the rows; null values
//   from the 64-entry tables; both running sums in double in reference order (bit-identical to
//   Cumulative_Score).  At every in-frame position it parks score[j-1]-indep[j-1] and the start-codon
//   match in a small temp list, then replays that list from the 3' end with the reference's
//   first / best / truncated-start logic (glimmer3.cc:1355-1429).
// ---------------------------------------------------------------------------
struct OrfFusedArgs {
    OrfScanArgs sc;
    GmgDevModel gene, nul;
    const float *gene6;
    uint64_t total;
    OrfTmp *tmp;
};

__global__ __launch_bounds__(256) void k_orf_fused(OrfFusedArgs fa)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_lds[];
    const OrfScanArgs &a = fa.sc;
    const int cstride = fa.gene.cstride;
    uint8_t *s_shift = s_lds;                                       // [3][cstride]
    float *s_dense = (float *)(s_lds + 3 * cstride);                // [3][64]
    float *s_part = s_dense + 3 * 64;                               // [3][20]
    __shared__ int8_t s_which[64];                                  // start pattern index of a codon (buff[j+2], buff[j+1], buff[j]), -1: none
    for (int i = threadIdx.x * 16; i < 3 * cstride; i += 256 * 16) *(uint4 *)(s_shift + i) = *(const uint4 *)(fa.gene.cshift + i);
    for (int i = threadIdx.x; i < 3 * 64; i += 256) s_dense[i] = fa.nul.dense[i];
    for (int i = threadIdx.x; i < 3 * 20; i += 256) s_part[i] = fa.nul.dense_part[i];
    if (threadIdx.x < 64) {                                         // Codon_t::Can_Be (gene.cc:39-66) for every definite codon
        const uint32_t c = threadIdx.x, m = (1u << ((c >> 4) & 3u)) << 8 | (1u << ((c >> 2) & 3u)) << 4 | (1u << (c & 3u));
        int which = -1;
        for (int p = a.n_pat - 1; p >= 0; p--) {
            const uint32_t x = m & a.pat[p];
            if ((x & 0xf00u) && (x & 0xf0u) && (x & 0x0fu)) which = p;
        }
        s_which[c] = (int8_t)which;
    }
    __syncthreads();

    const int W = fa.gene.W, D = fa.gene.D;
    const uint32_t sh_top = 2u * (uint32_t)(W - 1);
    const uint32_t ctot = (uint32_t)fa.gene.ctot;

    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (uint64_t)gridDim.x * blockDim.x) {
        const gmg_orf orf = a.orfs[i];
        const uint64_t r_off = a.read_off[orf.read];
        const int L = (int)(a.read_off[orf.read + 1] - r_off);
        const int len = orf.orf_len;
        const bool fwd = orf.frame > 0;
        int lo, hi, k0;
        bool trunc;
        if (fwd) { hi = orf.stop_position - 1; lo = hi - len; trunc = lo < 3 && a.allow_truncated; k0 = orf.stop_position - len - 2; }
        else { lo = orf.stop_position + 2; hi = lo + len; trunc = L - hi < 3 && a.allow_truncated; k0 = orf.stop_position + len + 4; }
        OrfTmp *tmp = fa.tmp + a.start_off[i];
        const int lowest_j = a.min_gene_len - 3 < 3 ? a.min_gene_len - 3 : 3;
        const float *rows = fa.gene6 + (fwd ? 0 : 3) * fa.total;
        // in-frame positions that can carry a start: j % 3 == 0, j >= lowest_j (>= 1), j + 3 >= Min_Gene_Len
        int j_lo = lowest_j > 1 ? lowest_j : 1;
        if (j_lo + 3 < a.min_gene_len) j_lo = a.min_gene_len - 3;
        j_lo = (j_lo + 2) / 3 * 3;

        // ---- pass 1, ascending: running sums, one temp entry per in-frame position (entry t <-> j_lo + 3t)
        double gsum = 0.0, nsum = 0.0;
        int n_tmp = 0;
        const int64_t dirg = fwd ? -1 : 1;
        int64_t g = (int64_t)r_off + (fwd ? hi - 1 : lo);           // base of buffer position 0
        uint32_t w = a.packed[g >> 4];
        const uint32_t comp = fwd ? 0u : 3u;
        auto next_code = [&]() __attribute__((always_inline)) {     // codes stream from a register, one word per 16 bases
            const uint32_t c = ((w >> (2u * (unsigned)(g & 15))) & 3u) ^ comp;
            const int64_t g2 = g + dirg;
            if ((g ^ g2) >> 4) w = a.packed[g2 >> 4];   // word -1 / one past the end: the guard words of gmg_reads
            g = g2;
            return c;
        };
        // (a) the first W-1 positions: partial windows, descent on the completed tree
        uint32_t C = 0;              // window register: char k of the window ending at j in bits [2k, 2k+1]
        uint32_t n6 = 0;             // the last three buffer chars, newest in bits 4-5 (index of the null tables; with the
                                     //  chars of in-frame position j-2 .. j it is also the Codon_t of position j-2)
        int j = 0, f = 1;
        double pend = 0.0;           // score[j-1] - indep[j-1] of the in-frame position whose codon is not complete yet
        const int head = len < W - 1 ? len : W - 1;
        auto in_frame_bookkeeping = [&](int jj, uint32_t codon6) __attribute__((always_inline)) {
            const int m3 = jj % 3;
            if (m3 == 2 && jj - 2 >= j_lo) {                        // the codon of in-frame position jj-2 is complete
                tmp[n_tmp].s = pend;
                tmp[n_tmp].which = s_which[codon6];
                n_tmp++;
            }
            if (m3 == 0) pend = gsum - nsum;                        // score[jj-1] - indep_score[jj-1]
        };
        for (; j < head; j++) {
            const uint32_t code = next_code();
            C = (C >> 2) | (code << sh_top);
            n6 = (n6 >> 2) | (code << 4);
            in_frame_bookkeeping(j, n6);
            const uint8_t *tab = s_shift + f * cstride;
            const int thr2 = 2 * ((W - 1) - j);
            uint32_t idx = 0, lvl = 0, width = 1, node = 0xffffffffu;
            for (int l = 0; l < D; l++) {
                const uint32_t sh = tab[lvl + idx];
                if (node == 0xffffffffu && (int)sh < thr2) node = lvl + idx;
                idx = (idx << 2) + ((C >> sh) & 3u);
                lvl += width;
                width <<= 2;
            }
            if (node == 0xffffffffu) node = lvl + idx;
            const float gv = fa.gene.crow[((size_t)f * ctot + node) * 4 + code];
            // null value: last three buffer chars, partial tables for j < 2
            float nv;
            if (j >= 2) nv = s_dense[f * 64 + n6];
            else if (j == 1) nv = s_part[f * 20 + 4 + (n6 >> 2)];
            else nv = s_part[f * 20 + code];
            gsum += (double)gv;
            nsum += (double)nv;
            f = f == 2 ? 0 : f + 1;
        }
        // (b) everything else: the whole-read window lies inside the ORF, so the value is row f of the gene-only pass
        for (; j < len; j++) {
            const float gv = rows[(uint64_t)f * fa.total + (uint64_t)g];
            const uint32_t code = next_code();
            n6 = (n6 >> 2) | (code << 4);
            in_frame_bookkeeping(j, n6);
            gsum += (double)gv;
            nsum += (double)s_dense[f * 64 + n6];
            f = f == 2 ? 0 : f + 1;
        }
        // an in-frame position whose codon never completed (len % 3 != 0) still counts, without a start codon
        {
            const int last_in = len >= 1 ? (len - 1) / 3 * 3 : -1;
            if (last_in >= j_lo && last_in + 2 > len - 1) { tmp[n_tmp].s = pend; tmp[n_tmp].which = -1; n_tmp++; }
        }

        // ---- pass 2, from the 3' end: the reference's scan over the in-frame positions (glimmer3.cc:1355-1429)
        gmg_start *out = a.starts + a.start_off[i];
        uint32_t n_starts = 0;
        int first_pos = 0, best_pos = 0, first_j = 0, best_j = 0;
        bool first_trunc = false, best_trunc = false;
        double first_score = -DBL_MAX, best_score = -DBL_MAX;
        for (int t = n_tmp - 1; t >= 0; t--) {
            const int j = j_lo + 3 * t;
            const int k = fwd ? k0 + (len - 1 - j) : k0 - (len - 1 - j);
            const int which = tmp[t].which;
            if (which >= 0 || (first_pos == 0 && trunc)) {
                const double next_s = tmp[t].s;
                const double pushed = (j + 2 > a.ignore_score_len && next_s < 0.0) ? 0.0 : next_s;
                gmg_start st;
                st.score = pushed; st.j = j + 2; st.pos = k; st.first = (first_pos == 0);
                if (which >= 0 && first_pos == 0 && trunc) {
                    st.which = -1; st.truncated = 1;
                    out[n_starts++] = st;
                    st.first = 0;
                }
                st.which = which; st.truncated = (which < 0);
                out[n_starts++] = st;
                if (first_pos == 0) { first_score = next_s; first_pos = k; first_j = j + 2; first_trunc = (first_pos == 0 && trunc); }
                if (next_s > best_score) { best_score = next_s; best_pos = k; best_j = j + 2; best_trunc = st.truncated; }
            }
        }
        if (a.use_first_start) { best_score = first_score; best_pos = first_pos; best_j = first_j; best_trunc = first_trunc; }
        (void)best_trunc;
        gmg_orf_result res;
        res.start_begin = (uint32_t)a.start_off[i];
        res.first_j = first_j; res.best_j = best_j; res.best_pos = best_pos;
        res.best_score = best_score;
        res.orf_is_truncated = trunc;
        if (first_j + 1 < a.min_gene_len) { res.n_starts = 0; res.is_tentative_gene = 0; res.gene_score = 0.0; }
        else {
            res.n_starts = n_starts;
            res.is_tentative_gene = best_score > a.start_threshold;
            res.gene_score = 100.0 * best_score / (best_j - 2);
        }
        a.results[i] = res;
        a.nst[i] = res.n_starts;
    }
}

// ---------------------------------------------------------------------------
// Events path (round 3; default model shapes whose sums are exact in any order -- the test of gmg_mg.hip's fused kernel).
//
// score[j-1] - indep_score[j-1] of an ORF is a sum over its buffer positions 0 .. j-1 (glimmer3.cc:1346-1347, 1369).  Every term
// is (double) a float of the gene table - (double) a float of the null table, all multiples of 2^(e_min - 150); while
// ceil(log2 R) + e_max - e_min <= 28 (R = longest read + 2) every partial sum in every order is an exact double, so the
// reference's two sequential sums and their difference are exact too and ANY way of adding the same terms gives the same bits.
// Then an ORF need not be walked base by base:
//   * from buffer position HEAD = 12 on (>= W-1) a position's window lies inside the ORF and its value is the whole-read value of
//     the six-frame pass -- row (1 + j) % 3 of the strand's three rows.  All ORFs of one reading-frame class use the same row at
//     the same base, so ONE running sum per strand and class over the read serves them all: k_orf_walk_sums writes, per base p
//     and strand, Q[p] = the sum of the class of p over the bases the walk has passed before p (one double per base and strand:
//     a base is an in-frame position of exactly one class), and sum_{k = HEAD}^{j-1} = Q[p_j] - Q[p_HEAD];
//   * the first HEAD positions (partial windows: the ORF's own start) are HEAD descents per ORF;
//   * the scan over the in-frame codons (glimmer3.cc:1355-1421) is integer work on the packed bases; a start codon -- an EVENT --
//     costs one 8-byte read of Q.
// k_orf_fused (one lane per ORF, ~290 positions of dependent double additions behind a gather each) stays as the path for models
// that fail the test; both give the reference's start lists (tests/test_gpu_parity.py, tests/bench/bench_orfs.py compares bytes).
// ---------------------------------------------------------------------------
struct OrfWalkArgs {
    const uint32_t *packed;
    const uint64_t *read_off;
    uint64_t n_reads, total;
    const float *gene6;          // [6][total]: rows 0-2 reversed buffer (forward ORFs), rows 3-5 complemented buffer
    const float *null_dense;     // [3][64] full-window values of the (3,2,3) null model
    double *q;                   // [2][total]: forward-strand sums, reverse-strand sums
    // k_orf_walk_sums8, sparse form: Q is written only where k_orf_events can ask for it
    const uint32_t *heads;       // [2][head_words] bit g of strand s: an ORF's HEAD position (k_orf_mark_heads); NULL: every base is written
    uint64_t head_words;
    uint64_t start_set;          // codons (first char << 4 | second << 2 | third) that match a start pattern
    // compact form (k_orf_walk_sums8p<true>): the values a unit's lanes need are written back to back at the unit's first step;
    // per lane of every unit the number of values in front of its own and its eight need bits
    uint16_t *q_pre;             // [2][meta_stride]
    uint8_t *q_need;             // [2][meta_stride]
    uint64_t meta_stride;        // total / 8 + n_reads + 64: lane L of the unit (read r at off, first step t0) sits at (off >> 3) + r + (t0 >> 3) + L
};

constexpr int OW_EL = 4;         // walk steps per lane and trip: 256 per wave

// One wave per (read, strand), 256 walk steps per trip; wave_class_scan (gmg_device.h) keeps the loads and the store coalesced
// (a lane on every 64th step) and the additions cheap (a lane on 4 consecutive steps, one DPP scan of the lane totals per class).
//   forward ORFs (frame > 0) walk DOWN the read: step t <-> base p = L-1-t; at base p' an ORF whose in-frame positions are
//     p == c - 1 (mod 3) takes row f = (c - p') mod 3 of rows 0-2 and the null value of window (S[p'+2], S[p'+1], S[p']);
//   reverse ORFs walk UP: step t <-> p = t; an ORF with in-frame positions p == c (mod 3) takes row f = (1 + p' - c) mod 3 of rows
//     3-5 and the null value of (comp S[p'-2], comp S[p'-1], comp S[p']).
// Q[p] = the class-of-p sum over the steps before p's.  Windows that leave the read see the neighbouring read's bases (or the
// guard words): such terms are in both Q values an ORF subtracts, or in neither.
__global__ __launch_bounds__(256) void k_orf_walk_sums(OrfWalkArgs a)
{
    __shared__ double s_null[3 * 64 + 1];                           // [192]: 0 for the steps beyond the walk's end
    __shared__ __attribute__((aligned(16))) double s_scan[4 * wcs_lds_doubles<OW_EL>()];
    for (int i = threadIdx.x; i < 3 * 64 + 1; i += 256) s_null[i] = i < 192 ? (double)a.null_dense[i] : 0.0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    double *s_mine = s_scan + (threadIdx.x >> 6) * wcs_lds_doubles<OW_EL>();
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t it = wave; it < 2 * a.n_reads; it += n_waves) {
        const uint64_t r = it >> 1;
        const bool fwd = (it & 1) == 0;
        const uint64_t off = a.read_off[r];
        const uint32_t n = (uint32_t)(a.read_off[r + 1] - off);
        const float *rows = a.gene6 + (fwd ? 0 : 3) * a.total;
        double *q = a.q + (fwd ? 0 : a.total);
        double carry[3] = {0.0, 0.0, 0.0};
        for (uint32_t t0 = 0; t0 < n; t0 += 64 * OW_EL) {
            double x[OW_EL][3];
            uint32_t cls[OW_EL];
#pragma unroll
            for (int u = 0; u < OW_EL; u++) {
                const uint32_t t = t0 + 64 * u + lane;
                const bool in = t < n;
                const uint32_t p = in ? (fwd ? n - 1 - t : t) : 0;  // position in the read
                const uint64_t g = off + p;
                float v[3];
#pragma unroll
                for (int f = 0; f < 3; f++) v[f] = in ? rows[(uint64_t)f * a.total + g] : 0.0f;
                const uint32_t w3 = (uint32_t)dev_window_bits(a.packed, (int64_t)g - 2) & 0xfffu;      // bases g-2 .. g+3 at fields 0 .. 5
                // the three chars of the null window, oldest buffer char in the low bits
                const uint32_t n6 = fwd ? (((w3 >> 8) & 3u) | (((w3 >> 6) & 3u) << 2) | (((w3 >> 4) & 3u) << 4)) : ((w3 & 63u) ^ 63u);
                const uint32_t pm = p % 3u;
                cls[u] = fwd ? (p + 1u) % 3u : pm;
#pragma unroll
                for (int c = 0; c < 3; c++) {       // class c takes row (c - pm) mod 3 (forward) / (1 + pm - c) mod 3 (reverse)
                    const uint32_t f = fwd ? (uint32_t)(c + 3 - (int)pm) % 3u : (uint32_t)(1 + (int)pm + 3 - c) % 3u;
                    const float gv = f == 0 ? v[0] : f == 1 ? v[1] : v[2];
                    x[u][c] = (double)gv - s_null[in ? f * 64u + n6 : 192u];
                }
            }
            wave_class_scan<OW_EL, true>(s_mine, x, carry, lane);
#pragma unroll
            for (int u = 0; u < OW_EL; u++) {
                const uint32_t t = t0 + 64 * u + lane;
                if (t < n) q[off + (fwd ? n - 1 - t : t)] = cls[u] == 0 ? x[u][0] : cls[u] == 1 ? x[u][1] : x[u][2];
            }
        }
    }
}

// The same sums with a lane on EIGHT CONSECUTIVE walk steps (k_orf_walk_sums keeps a lane on every 64th step for coalescing and moves the
// values through LDS twice per 256 steps: 957 vector instructions per (read, strand) of 500 bases, the kernel's bound).  Here a
// trip is 512 steps: the lane's 8 x 3 gene values are two 16-byte loads per row (the 32 bytes a lane reads are its own), the class of
// a step is relabelled per lane so that the unrolled loop indexes statically (class c' takes row (c' + e) % 3 at the lane's step e), the
// lane sums its steps serially, ONE wave scan of the lane totals per class carries on, and the lane's eight Q values leave as four
// 16-byte stores.  Exact in any order (the events path's test), so the bits are k_orf_walk_sums' bits.
struct __attribute__((packed, aligned(4))) OwF4 { float v[4]; };
struct __attribute__((packed, aligned(8))) OwD2 { double v[2]; };

// What bounds it is not found.  4.1 - 4.3 ms per 1 M reads in every form (profiles/r05_orfs_walk8_ab.txt, r05_pmc_summary_k_orf_walk_sums8_*.txt):
//   * it reads its 12 GB at 2.9 TB/s; a probe with the same read pattern and no arithmetic reaches 6.0 (tools/probes/read_rows_probe.hip);
//   * Q written only where it can be read (7.8 -> 1.7 GB written): 4.33 -> 4.29 ms;
//   * 1.7e9 -> 1.0e9 vector instructions (this form: the trip's body once per strand, so that every register index is static and no value
//     of a step beyond the read needs a select; one load path; 32-bit window): 63 % -> 38 % of the vector pipes' time, 4.3 -> 4.1 ms;
//   * three, four (this form), six waves per SIMD (four steps per lane): no change; more waves by force (-DOW8_WAVES): spills, slower;
//   * the next unit's loads issued before this one is worked on (offsets two items ahead): 10.24 against 10.17 ms per call.
// Two traps on the way: a struct of the unit's registers handed to lambdas by reference, and `fwd ? gv[7 - e] : gv[e]` with the strand as
// data, both put the lane's 24 values into scratch memory (5.6 ms) -- hence the per-strand body.
#ifndef GMG_OW_STAMPS
#define GMG_OW_STAMPS 0          // diagnostic build: cycles per phase of k_orf_walk_sums8, summed over all waves (tools/ow_stamps.py); not in the product
#endif
#if GMG_OW_STAMPS
__device__ unsigned long long g_ow_stamps[8];
extern "C" int gmg_debug_ow_stamps(unsigned long long *out, int reset)
{
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; return hipMemcpyToSymbol(HIP_SYMBOL(g_ow_stamps), z, sizeof z) == hipSuccess ? 0 : -1; }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ow_stamps), 64) == hipSuccess ? 0 : -1;
}
#define OW_STAMP(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); st_acc[i] += now_ - st_prev; st_prev = now_; } while (0)
#else
#define OW_STAMP(i) do { } while (0)
#endif
#ifdef OW8_WAVES
__global__ __launch_bounds__(256, OW8_WAVES) void k_orf_walk_sums8(OrfWalkArgs a)
#else
__global__ __launch_bounds__(256) void k_orf_walk_sums8(OrfWalkArgs a)
#endif
{
    __shared__ double s_null[3 * 64];
    for (int i = threadIdx.x; i < 3 * 64; i += 256) s_null[i] = (double)a.null_dense[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
#if GMG_OW_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = __builtin_readcyclecounter();
#endif
    for (uint64_t it = wave; it < 2 * a.n_reads; it += n_waves) {
        const uint64_t r = it >> 1;
        const bool fwd_item = (it & 1) == 0;
        const uint64_t off = a.read_off[r];
        const uint32_t n = (uint32_t)(a.read_off[r + 1] - off);
        const float *rows = a.gene6 + (fwd_item ? 0 : 3) * a.total;
        double *q = a.q + (fwd_item ? 0 : a.total);
        double carry[3] = {0.0, 0.0, 0.0};              // by TRUE class
        // (the body once per strand: which of a lane's eight values belongs to which step is then known at compile time -- with the strand as
        // data the compiler indexed the lane's values dynamically, i.e. kept them in scratch memory)
        auto trip = [&](auto FWD_, const uint32_t t0) __attribute__((always_inline)) {
            constexpr bool fwd = decltype(FWD_)::value;
            OW_STAMP(0);                                // the item's offsets (a dependent load), loop overhead
            const uint32_t tb = t0 + 8u * lane;         // the lane's first step; its steps tb .. tb + 7 are bases p0, p0 -/+ 1, ..
            const bool any = tb < n;
            const uint32_t cnt = any ? (n - tb < 8u ? n - tb : 8u) : 0u;
            // the lowest base of the lane's eight (forward: the last step's), clamped into the read: what lies outside is masked below
            const int64_t p_first = fwd ? (int64_t)n - 1 - (int64_t)tb : (int64_t)tb;       // position of step tb in the read
            const int64_t p_lo = fwd ? p_first - 7 : p_first;
            const int64_t g_lo = (int64_t)off + p_lo;
            float gv[3][8];
            const bool whole = any && cnt == 8u && g_lo >= 0 && (uint64_t)g_lo + 8 <= a.total;
            // Always the eight values at g_lo .. g_lo + 7 of every row, also for the lane that holds a read's last steps and for the lanes
            // behind it: what lies beyond the read enters only sums that nobody reads -- the lane's own Q values BEHIND its last valid
            // step (not stored), the totals of the lanes behind it (they store nothing), the carry into a next trip (a read that has
            // one fills all lanes of this one).  One load path, no selects: the kernel is bound by its vector instructions (1.5e9
            // wave-instructions x 4 cycles on 1,024 SIMDs = 2.8 of its 4.3 ms; a wave-instruction takes four cycles on a SIMD).
            const bool oob = g_lo < 0 || (uint64_t)g_lo + 8 > a.total;         // (the batch's first / last read, or a lane without steps)
            {
                const float *rb = rows + (oob ? 0 : g_lo);
#pragma unroll
                for (int f = 0; f < 3; f++) {
                    const OwF4 lo4 = *(const OwF4 *)(rb + (uint64_t)f * a.total), hi4 = *(const OwF4 *)(rb + (uint64_t)f * a.total + 4);
#pragma unroll
                    for (int k = 0; k < 4; k++) { gv[f][k] = lo4.v[k]; gv[f][4 + k] = hi4.v[k]; }
                }
            }
            if (any && oob) {                           // (a handful of lanes of the whole grid: element by element, inside the array)
#pragma unroll
                for (int f = 0; f < 3; f++)
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const int64_t gk = g_lo + k;
                        gv[f][k] = gk >= 0 && (uint64_t)gk < a.total ? rows[(uint64_t)f * a.total + (uint64_t)gk] : 0.0f;
                    }
            }
            // bases p_lo - 2 .. p_lo + 9 as 2-bit fields (the null model's window reaches two bases beyond a step's own); E = 8: all in its low word
            const uint32_t win = (uint32_t)dev_window_bits(a.packed, any ? g_lo - 2 : 0);
#if GMG_OW_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            OW_STAMP(1);                                // the unit's loads issued and waited for
            // step e of the lane sits at index k = 7 - e (forward) / e (reverse) of the eight; its class: forward (p + 1) % 3, reverse p % 3.
            // Relabelled class c' = (true class - class of step 0 + ..): see below -- value row f of class c at base p is (c - p) % 3
            // forward and (1 + p - c) % 3 reverse (k_orf_walk_sums); with p = p_first -/+ e both become (c' + e) % 3 for
            // c' = (c - p_first) % 3 forward, (1 + p_first - c) % 3 reverse... reverse runs the other way: (c' - e) % 3
            const uint32_t pm = (uint32_t)(((p_first % 3) + 3) % 3);
            double acc[3] = {0.0, 0.0, 0.0}, Pq[8];     // Pq[e]: the relabelled class (1 - e) % 3 -- the class of step e's own base -- before step e
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int k = fwd ? 7 - e : e;
                // null window of the base at index k: forward (S[p+2], S[p+1], S[p]) oldest first, reverse the complement of (S[p-2], S[p-1], S[p])
                const uint32_t w3 = (win >> (2 * k)) & 0xfffu;                        // fields: bases p - 2 .. p + 3
                const uint32_t n6 = fwd ? (((w3 >> 8) & 3u) | (((w3 >> 6) & 3u) << 2) | (((w3 >> 4) & 3u) << 4)) : ((w3 & 63u) ^ 63u);
                double v[3];
#pragma unroll
                for (int f = 0; f < 3; f++) v[f] = (double)(fwd ? gv[f][7 - e] : gv[f][e]) - s_null[f * 64 + n6];
                // relabelled class cp takes row (cp + e) % 3 (forward) / (cp + 3 - e % 3) % 3 ... reverse: p grows with e: row = (1 + p - c) % 3
                Pq[e] = acc[((1 - e) % 3 + 3) % 3];     // the sum over the steps BEFORE this one
#pragma unroll
                for (int cp = 0; cp < 3; cp++) acc[cp] += v[(cp + e) % 3];
            }
            // relabelling: forward row f = (c - p) % 3 with p = p_first - e  ->  (c - pm + e) % 3: cp = (c - pm) % 3
            //              reverse row f = (1 + p - c) % 3 with p = p_first + e -> (1 + pm - c + e) % 3: cp = (1 + pm - c) % 3
            OW_STAMP(2);                                // the eight steps
            double tot[3], base[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const uint32_t cp = fwd ? ((uint32_t)c + 3u - pm) % 3u : (1u + pm + 3u - (uint32_t)c) % 3u;
                tot[c] = cp == 0u ? acc[0] : cp == 1u ? acc[1] : acc[2];
            }
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const double inc = wcs_wave_scan(tot[c]);
                base[c] = carry[c] + (inc - tot[c]);
                carry[c] += wcs_last_lane(inc);
            }
            // Q of step e: the sum of the class OF ITS BASE: forward class (p + 1) % 3, reverse p % 3 -- relabelled: forward
            // cp = (p + 1 - pm) % 3 with p = p_first - e -> (1 - e) % 3; reverse cp = (1 + pm - p) % 3 -> (1 - e) % 3: static
            OW_STAMP(3);                                // the three wave scans
            double qv[8], base_r[3];                    // base_r[cp]: what lies in front of the lane for relabelled class cp
#pragma unroll
            for (int cp = 0; cp < 3; cp++) {
                const uint32_t c = fwd ? ((uint32_t)cp + pm) % 3u : (1u + pm + 3u - (uint32_t)cp) % 3u;     // the true class of cp
                base_r[cp] = c == 0u ? base[0] : c == 1u ? base[1] : base[2];
            }
#pragma unroll
            for (int e = 0; e < 8; e++) Pq[e] += base_r[((1 - e) % 3 + 3) % 3];      // Q of step e
#pragma unroll
            for (int k = 0; k < 8; k++) qv[k] = fwd ? Pq[7 - k] : Pq[k];            // ... of the base at index k (static indices: registers)
            // Sparse form (a.heads): k_orf_events reads Q at an ORF's HEAD position (marked by k_orf_mark_heads), at a start codon --
            // forward: the codon that ENDS at the base, reverse: the reverse complement of the codon that BEGINS there -- and at the first
            // in-frame position of an ORF that runs into the read's end (within five bases of it).  Everything else is never read.
            uint32_t need = 0xffu;
            if (a.heads && any) {
                need = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const uint32_t f5 = (win >> (2 * k)) & 0x3ffu;                       // bases p - 2 .. p + 2 of the base at index k
                    const uint32_t c = fwd ? ((f5 & 3u) << 4 | (f5 & 12u) | ((f5 >> 4) & 3u))
                                           : ((((f5 >> 8) & 3u) << 4 | ((f5 >> 6) & 3u) << 2 | ((f5 >> 4) & 3u)) ^ 63u);
                    const int64_t p = p_lo + k;
                    const bool edge = p < 8 || p + 8 >= (int64_t)n;
                    need |= (((a.start_set >> c) & 1ull) || edge ? 1u : 0u) << k;
                }
                const uint64_t gb = (uint64_t)(g_lo < 0 ? 0 : g_lo);
                const uint32_t *hw = a.heads + (fwd ? 0 : a.head_words) + (gb >> 5);
                const uint64_t two = (uint64_t)hw[1] << 32 | hw[0];
                need |= (uint32_t)(two >> (gb & 31u)) & 0xffu;
            }
            OW_STAMP(4);                                // Q values, which of them are needed
            if (whole) {
#pragma unroll
                for (int k = 0; k < 8; k += 2)
                    if ((need >> k) & 3u) { OwD2 d; d.v[0] = qv[k]; d.v[1] = qv[k + 1]; *(OwD2 *)(q + (uint64_t)g_lo + k) = d; }
            } else if (any) {
                double *qb = q + ((int64_t)off + p_lo);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int64_t p = p_lo + k;
                    const uint32_t e = (uint32_t)(fwd ? 7 - k : k);
                    if (e < cnt && p >= 0 && p < (int64_t)n) qb[k] = qv[k];
                }
            }
            OW_STAMP(5);                                // stores issued
        };
        for (uint32_t t0 = 0; t0 < n; t0 += 512) {
            if (fwd_item) trip(std::integral_constant<bool, true>(), t0); else trip(std::integral_constant<bool, false>(), t0);
        }
    }
#if GMG_OW_STAMPS
    if ((threadIdx.x & 63u) == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&g_ow_stamps[i], st_acc[i]);
#endif
}

// k_orf_walk_sums8p: the same sums with the phases of consecutive units overlapped inside a wave.  The stamped build of k_orf_walk_sums8
// (tools/ow_stamps.py) shows a wave's time as four phases one after the other -- its item's offsets (a dependent load) 21 %, the unit's
// rows 30 %, arithmetic 27 %, stores 22 % -- and three to four waves per SIMD do not overlap them.  Here a wave's items are all of one
// strand (the grid has an even number of waves), the whole item loop runs once per strand, and the NEXT unit's loads are issued before
// this unit's arithmetic (its offsets two items ahead); a scheduling barrier keeps the compiler from moving them down again.
template <bool COMPACT>
__global__ __launch_bounds__(256) void k_orf_walk_sums8p(OrfWalkArgs a)
{
    __shared__ double s_null[3 * 64];
    for (int i = threadIdx.x; i < 3 * 64; i += 256) s_null[i] = (double)a.null_dense[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t n_items = 2 * a.n_reads;             // item it = (read it / 2, strand it % 2)
    if (wave >= n_items) return;

    auto run = [&](auto FWD_) __attribute__((always_inline)) {
        constexpr bool fwd = decltype(FWD_)::value;
        const float *rows = a.gene6 + (fwd ? 0 : 3) * a.total;
        double *q = a.q + (fwd ? 0 : a.total);
        const uint32_t *heads = a.heads ? a.heads + (fwd ? 0 : a.head_words) : nullptr;
        // where the lane's eight steps of a unit lie (k_orf_walk_sums8)
#define OWP_GEO(OFF, N, T0)                                                                                                       \
        const uint32_t tb = (T0) + 8u * lane;                                                                                    \
        const bool any = tb < (N);                                                                                               \
        const int64_t p_first = fwd ? (int64_t)(N) - 1 - (int64_t)tb : (int64_t)tb;                                              \
        const int64_t p_lo = fwd ? p_first - 7 : p_first;                                                                        \
        const int64_t g_lo = (int64_t)(OFF) + p_lo;
        // a unit's loads into gn / wn / hn (plain locals: a struct of them handed on by reference went to scratch memory)
#define OWP_LOAD(OFF, N, T0)                                                                                                      \
        {                                                                                                                        \
            OWP_GEO(OFF, N, T0)                                                                                                  \
            const bool oob = g_lo < 0 || (uint64_t)g_lo + 8 > a.total;                                                           \
            const float *rb = rows + (oob ? 0 : g_lo);                                                                           \
            _Pragma("unroll") for (int f = 0; f < 3; f++) {                                                                      \
                const OwF4 lo4 = *(const OwF4 *)(rb + (uint64_t)f * a.total), hi4 = *(const OwF4 *)(rb + (uint64_t)f * a.total + 4); \
                _Pragma("unroll") for (int k = 0; k < 4; k++) { gn[f][k] = lo4.v[k]; gn[f][4 + k] = hi4.v[k]; }                 \
            }                                                                                                                    \
            if (any && oob) {                                                                                                    \
                _Pragma("unroll") for (int f = 0; f < 3; f++)                                                                    \
                    _Pragma("unroll") for (int k = 0; k < 8; k++) {                                                              \
                        const int64_t gk = g_lo + k;                                                                             \
                        gn[f][k] = gk >= 0 && (uint64_t)gk < a.total ? rows[(uint64_t)f * a.total + (uint64_t)gk] : 0.0f;      \
                    }                                                                                                            \
            }                                                                                                                    \
            /* the window's and the head bits' WORDS only: shifting them here would wait for every load in front of them */       \
            const int64_t wfirst = any ? g_lo - 2 : 0;                                                                            \
            const uint32_t *pw = a.packed + (wfirst >> 4);       /* (arithmetic shift: floor for negatives; guard words) */       \
            wn0 = pw[0]; wn1 = pw[1]; wsh = 2u * (uint32_t)(wfirst & 15);                                                         \
            hn0 = 0; hn1 = 0; hsh = 0;                                                                                           \
            if (heads && any) {                                                                                                  \
                const uint64_t gb = (uint64_t)(g_lo < 0 ? 0 : g_lo);                                                             \
                const uint32_t *hw = heads + (gb >> 5);                                                                          \
                hn0 = hw[0]; hn1 = hw[1]; hsh = (uint32_t)(gb & 31u);                                                            \
            }                                                                                                                    \
        }
        uint64_t it = wave;
        uint64_t off = a.read_off[it >> 1];
        uint32_t n = (uint32_t)(a.read_off[(it >> 1) + 1] - off), t0 = 0;
        uint64_t off_b = 0, end_b = 0;                  // the offsets of the item after this one
        if (it + n_waves < n_items) { off_b = a.read_off[(it + n_waves) >> 1]; end_b = a.read_off[((it + n_waves) >> 1) + 1]; }
        float gn[3][8];
        uint32_t wn0, wn1, wsh, hn0, hn1, hsh;
        OWP_LOAD(off, n, t0)
        double carry[3] = {0.0, 0.0, 0.0};              // by TRUE class
        while (true) {
            float gv[3][8];
#pragma unroll
            for (int f = 0; f < 3; f++)
#pragma unroll
                for (int k = 0; k < 8; k++) gv[f][k] = gn[f][k];
            // bases p_lo - 2 .. p_lo + 13 as 2-bit fields (bits 0 .. 25 are used); the head bits of the lane's eight bases
            const uint32_t win = __builtin_amdgcn_alignbit(wn1, wn0, wsh), headbits = __builtin_amdgcn_alignbit(hn1, hn0, hsh) & 0xffu;
            const uint64_t it_of_unit = it;            // (it moves on below when the next unit is another item)
            // the unit after this one: the read's next 512 steps, or the wave's next item
            uint64_t n_off;
            uint32_t n_n, n_t0;
            bool have_next;
            if (t0 + 512u < n) { n_off = off; n_n = n; n_t0 = t0 + 512u; have_next = true; }
            else {
                it += n_waves;
                have_next = it < n_items;
                n_off = off_b; n_n = (uint32_t)(end_b - off_b); n_t0 = 0;
                if (it + n_waves < n_items) { off_b = a.read_off[(it + n_waves) >> 1]; end_b = a.read_off[((it + n_waves) >> 1) + 1]; }
            }
            if (have_next) OWP_LOAD(n_off, n_n, n_t0)
            __builtin_amdgcn_sched_barrier(0);          // (the loads stay in front of the arithmetic)
            if (t0 == 0) { carry[0] = 0.0; carry[1] = 0.0; carry[2] = 0.0; }
            {
                OWP_GEO(off, n, t0)
                const uint32_t cnt = any ? (n - tb < 8u ? n - tb : 8u) : 0u;
                const bool whole = any && cnt == 8u && g_lo >= 0 && (uint64_t)g_lo + 8 <= a.total;
                const uint32_t pm = (uint32_t)(((p_first % 3) + 3) % 3);
                double acc[3] = {0.0, 0.0, 0.0}, Pq[8];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int k = fwd ? 7 - e : e;
                    const uint32_t w3 = (win >> (2 * k)) & 0xfffu;
                    const uint32_t n6 = fwd ? (((w3 >> 8) & 3u) | (((w3 >> 6) & 3u) << 2) | (((w3 >> 4) & 3u) << 4)) : ((w3 & 63u) ^ 63u);
                    double v[3];
#pragma unroll
                    for (int f = 0; f < 3; f++) v[f] = (double)gv[f][k] - s_null[f * 64 + n6];
                    Pq[e] = acc[((1 - e) % 3 + 3) % 3];
#pragma unroll
                    for (int cp = 0; cp < 3; cp++) acc[cp] += v[(cp + e) % 3];
                }
                double tot[3], base[3];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const uint32_t cp = fwd ? ((uint32_t)c + 3u - pm) % 3u : (1u + pm + 3u - (uint32_t)c) % 3u;
                    tot[c] = cp == 0u ? acc[0] : cp == 1u ? acc[1] : acc[2];
                }
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const double inc = wcs_wave_scan(tot[c]);
                    base[c] = carry[c] + (inc - tot[c]);
                    carry[c] += wcs_last_lane(inc);
                }
                double qv[8], base_r[3];
#pragma unroll
                for (int cp = 0; cp < 3; cp++) {
                    const uint32_t c = fwd ? ((uint32_t)cp + pm) % 3u : (1u + pm + 3u - (uint32_t)cp) % 3u;
                    base_r[cp] = c == 0u ? base[0] : c == 1u ? base[1] : base[2];
                }
#pragma unroll
                for (int e = 0; e < 8; e++) qv[fwd ? 7 - e : e] = base_r[((1 - e) % 3 + 3) % 3] + Pq[e];
                uint32_t need = 0xffu;
                if (heads && any) {
                    need = headbits;
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const uint32_t f5 = (win >> (2 * k)) & 0x3ffu;
                        const uint32_t c = fwd ? ((f5 & 3u) << 4 | (f5 & 12u) | ((f5 >> 4) & 3u))
                                               : ((((f5 >> 8) & 3u) << 4 | ((f5 >> 6) & 3u) << 2 | ((f5 >> 4) & 3u)) ^ 63u);
                        const int64_t p = p_lo + k;
                        const bool edge = p < 8 || p + 8 >= (int64_t)n;
                        need |= (((a.start_set >> c) & 1ull) || edge ? 1u : 0u) << k;
                    }
                }
                if (COMPACT) {
                    // Scattered 16-byte stores of one value in seven cost what writing every base costs (the memory rewrites whole sectors:
                    // 2.1 of the kernel's 4.3 ms either way, profiles/r05_orfs_walk8_elimination.txt).  So the unit's needed values go
                    // back to back to the unit's first entries, in lane order and inside a lane by base, and k_orf_events finds a
                    // value by its rank: the values in front of the lane's (q_pre) + the need bits below its own (q_need).
                    need &= fwd ? (0xffu << (8u - cnt)) & 0xffu : (1u << cnt) - 1u;     // (only steps of the read)
                    const uint32_t mine = (uint32_t)__popc(need);
                    uint32_t incl = mine;
#define OWP_ADD(CTRL, RM) incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, CTRL, RM, 0xf, false)
                    OWP_ADD(0x111, 0xf); OWP_ADD(0x112, 0xf); OWP_ADD(0x114, 0xf); OWP_ADD(0x118, 0xf); OWP_ADD(0x142, 0xa); OWP_ADD(0x143, 0xc);
#undef OWP_ADD
                    const uint32_t excl = incl - mine;
                    if (any) {
                        double *qu = q + off + t0 + excl;
                        uint32_t slot = 0;
#pragma unroll
                        for (int k = 0; k < 8; k++)
                            if ((need >> k) & 1u) { qu[slot] = qv[k]; slot++; }
                        const uint64_t mi = (fwd ? 0 : a.meta_stride) + (off >> 3) + (it_of_unit >> 1) + (t0 >> 3) + lane;
                        a.q_pre[mi] = (uint16_t)excl;
                        a.q_need[mi] = (uint8_t)need;
                    }
                } else if (whole) {
#pragma unroll
                    for (int k = 0; k < 8; k += 2)
                        if ((need >> k) & 3u) { OwD2 d; d.v[0] = qv[k]; d.v[1] = qv[k + 1]; *(OwD2 *)(q + (uint64_t)g_lo + k) = d; }
                } else if (any) {
                    double *qb = q + ((int64_t)off + p_lo);
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const int64_t p = p_lo + k;
                        const uint32_t e = (uint32_t)(fwd ? 7 - k : k);
                        if (e < cnt && p >= 0 && p < (int64_t)n) qb[k] = qv[k];
                    }
                }
            }
            if (!have_next) break;
            off = n_off; n = n_n; t0 = n_t0;
        }
#undef OWP_LOAD
#undef OWP_GEO
    };
    if ((wave & 1) == 0) run(std::integral_constant<bool, true>()); else run(std::integral_constant<bool, false>());
}

// the HEAD position of every ORF (what k_orf_events reads as q_head), one bit per base and strand
__global__ __launch_bounds__(256) void k_orf_mark_heads(const gmg_orf *orfs, const uint64_t n, const uint64_t *read_off, const int W, uint32_t *heads,
                                                        const uint64_t head_words)
{
    const int HEAD = (W - 1 + 2) / 3 * 3;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const gmg_orf orf = orfs[i];
        if (orf.orf_len <= HEAD) continue;
        const uint64_t r_off = read_off[orf.read];
        const bool fwd = orf.frame > 0;
        const int64_t g_b0 = (int64_t)r_off + (fwd ? orf.stop_position - 1 - 1 : orf.stop_position + 2);      // (k_orf_events: hi - 1 / lo)
        const uint64_t g = (uint64_t)(g_b0 + (fwd ? -HEAD : HEAD));
        atomicOr(heads + (fwd ? 0 : head_words) + (g >> 5), 1u << (g & 31u));
    }
}

struct OrfEventArgs {
    OrfScanArgs sc;
    GmgDevModel gene, nul;
    const float *gene6;
    const double *q;
    uint64_t total;
    const uint16_t *q_pre;       // compact form of Q (k_orf_walk_sums8p<true>), or NULL: Q[strand][base]
    const uint8_t *q_need;
    uint64_t meta_stride;
};

// One lane per ORF: the HEAD positions by descent, then the codons from the 5' end down to the stop as the reference visits them
// (glimmer3.cc:1355-1421), pure bit work on the packed bases; only a start codon touches memory (Q, and the start it pushes).
__global__ __launch_bounds__(256) void k_orf_events(OrfEventArgs fa)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_lds[];
    const OrfScanArgs &a = fa.sc;
    const int cstride = fa.gene.cstride;
    uint8_t *s_shift = s_lds;                                       // [3][cstride]
    float *s_dense = (float *)(s_lds + 3 * cstride);                // [3][64]
    float *s_part = s_dense + 3 * 64;                               // [3][20]
    __shared__ int8_t s_which[64];                                  // start pattern index of a codon (buff[j+2], buff[j+1], buff[j]), -1: none
    for (int i = threadIdx.x * 16; i < 3 * cstride; i += 256 * 16) *(uint4 *)(s_shift + i) = *(const uint4 *)(fa.gene.cshift + i);
    for (int i = threadIdx.x; i < 3 * 64; i += 256) s_dense[i] = fa.nul.dense[i];
    for (int i = threadIdx.x; i < 3 * 20; i += 256) s_part[i] = fa.nul.dense_part[i];
    if (threadIdx.x < 64) {                                         // Codon_t::Can_Be (gene.cc:39-66) for every definite codon
        const uint32_t c = threadIdx.x, m = (1u << ((c >> 4) & 3u)) << 8 | (1u << ((c >> 2) & 3u)) << 4 | (1u << (c & 3u));
        int which = -1;
        for (int p = a.n_pat - 1; p >= 0; p--) {
            const uint32_t x = m & a.pat[p];
            if ((x & 0xf00u) && (x & 0xf0u) && (x & 0x0fu)) which = p;
        }
        s_which[c] = (int8_t)which;
    }
    __syncthreads();

    const int W = fa.gene.W, D = fa.gene.D;
    const int HEAD = (W - 1 + 2) / 3 * 3;                           // the smallest multiple of 3 >= W - 1 (12 for W = 12): <= 15
    const uint32_t sh_top = 2u * (uint32_t)(W - 1);
    const uint32_t ctot = (uint32_t)fa.gene.ctot;

    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (uint64_t)gridDim.x * blockDim.x) {
        const gmg_orf orf = a.orfs[i];
        const uint64_t r_off = a.read_off[orf.read];
        const int L = (int)(a.read_off[orf.read + 1] - r_off);
        const int len = orf.orf_len;
        const bool fwd = orf.frame > 0;
        int lo, hi, k0;
        bool trunc;
        if (fwd) { hi = orf.stop_position - 1; lo = hi - len; trunc = lo < 3 && a.allow_truncated; k0 = orf.stop_position - len - 2; }
        else { lo = orf.stop_position + 2; hi = lo + len; trunc = L - hi < 3 && a.allow_truncated; k0 = orf.stop_position + len + 4; }
        const int lowest_j = a.min_gene_len - 3 < 3 ? a.min_gene_len - 3 : 3;
        int j_lo = lowest_j > 1 ? lowest_j : 1;
        if (j_lo + 3 < a.min_gene_len) j_lo = a.min_gene_len - 3;
        j_lo = (j_lo + 2) / 3 * 3;
        const uint32_t comp = fwd ? 0u : 3u;
        const int64_t g_b0 = (int64_t)r_off + (fwd ? hi - 1 : lo);  // base of buffer position 0; position j at g_b0 -/+ j
        const int64_t dirg = fwd ? -1 : 1;

        // ---- the head: score[j-1] - indep[j-1] for j = 3, 6, .. HEAD, by the partial-window rule (icm.cc:807-842)
        double dh[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};             // dh[j / 3] (dh[0] unused: j >= 1)
        {
            double gsum = 0.0, nsum = 0.0;
            uint32_t C = 0, n6 = 0;
            int f = 1;
            const int head = len < HEAD ? len : HEAD;
            for (int j = 0; j < head; j++) {
                const int64_t g = g_b0 + dirg * j;
                const uint32_t code = (uint32_t)dev_code(a.packed, (uint64_t)g) ^ comp;
                C = (C >> 2) | (code << sh_top);
                n6 = (n6 >> 2) | (code << 4);
                const uint8_t *tab = s_shift + f * cstride;
                const int thr2 = 2 * ((W - 1) - j);
                uint32_t idx = 0, lvl = 0, width = 1, node = 0xffffffffu;
                for (int l = 0; l < D; l++) {
                    const uint32_t sh = tab[lvl + idx];
                    if (node == 0xffffffffu && (int)sh < thr2) node = lvl + idx;
                    idx = (idx << 2) + ((C >> sh) & 3u);
                    lvl += width;
                    width <<= 2;
                }
                if (node == 0xffffffffu) node = lvl + idx;
                const float gv = fa.gene.crow[((size_t)f * ctot + node) * 4 + code];
                float nv;
                if (j >= 2) nv = s_dense[f * 64 + n6];
                else if (j == 1) nv = s_part[f * 20 + 4 + (n6 >> 2)];
                else nv = s_part[f * 20 + code];
                gsum += (double)gv;
                nsum += (double)nv;
                if (j % 3 == 2) dh[(j + 1) / 3] = gsum - nsum;      // = score[j] - indep[j]: the value position j + 1 asks for
                f = f == 2 ? 0 : f + 1;
            }
        }
        // Q at base g of this read and strand: the array itself, or (compact form) the value's rank inside its unit of 512 walk steps
        auto q_at = [&](const int64_t g) __attribute__((always_inline)) -> double {
            if (!fa.q_pre) return fa.q[(fwd ? 0 : fa.total) + (uint64_t)g];
            const uint32_t p = (uint32_t)(g - (int64_t)r_off), t = fwd ? (uint32_t)L - 1u - p : p;     // the base's walk step
            const uint32_t t0 = t & ~511u, ln = (t & 511u) >> 3, k = fwd ? 7u - (t & 7u) : (t & 7u);
            const uint64_t mi = (fwd ? 0 : fa.meta_stride) + (r_off >> 3) + orf.read + (t0 >> 3) + ln;
            const uint32_t rank = (uint32_t)fa.q_pre[mi] + (uint32_t)__popc((uint32_t)fa.q_need[mi] & ((1u << k) - 1u));
            return fa.q[(fwd ? 0 : fa.total) + r_off + t0 + rank];
        };
        const double q_head = len > HEAD ? q_at(g_b0 + dirg * HEAD) : 0.0;

        // ---- the scan, from the 5' end (j = len - 1) down; only in-frame positions j >= j_lo can carry a start
        gmg_start *out = a.starts + a.start_off[i];
        uint32_t n_starts = 0;
        int first_pos = 0, best_pos = 0, first_j = 0, best_j = 0;
        bool first_trunc = false, best_trunc = false;
        double first_score = -DBL_MAX, best_score = -DBL_MAX;
        int j = len - 1;
        if (j >= j_lo) {
            int64_t g = g_b0 + dirg * j;                            // walks towards g_b0
            uint32_t w = a.packed[g >> 4];
            auto next_code = [&]() __attribute__((always_inline)) {
                const uint32_t c = ((w >> (2u * (unsigned)(g & 15))) & 3u) ^ comp;
                const int64_t g2 = g - dirg;
                if ((g ^ g2) >> 4) w = a.packed[g2 >> 4];
                g = g2;
                return c;
            };
            uint32_t n6 = 0, have = 0;                              // the last three chars shifted in (newest = lowest j in bits 0-1)
            for (; j % 3 != 0; j--) { n6 = ((n6 << 2) | next_code()) & 63u; have++; }     // the incomplete codon at the 5' end
            for (; j >= j_lo; j -= 3) {
                n6 = ((n6 << 2) | next_code()) & 63u;               // buff[j]: the codon (buff[j+2], buff[j+1], buff[j]) is complete if have >= 2
                const int which = have >= 2 ? (int)s_which[n6] : -1;
                have = 2;
                if (which >= 0 || (first_pos == 0 && trunc)) {
                    const int k = fwd ? k0 + (len - 1 - j) : k0 - (len - 1 - j);
                    double next_s;
                    if (j <= HEAD) next_s = dh[j / 3];
                    else next_s = dh[HEAD / 3] + (q_at(g_b0 + dirg * j) - q_head);
                    const double pushed = (j + 2 > a.ignore_score_len && next_s < 0.0) ? 0.0 : next_s;
                    gmg_start st;
                    st.score = pushed; st.j = j + 2; st.pos = k; st.first = (first_pos == 0);
                    if (which >= 0 && first_pos == 0 && trunc) {
                        st.which = -1; st.truncated = 1;
                        out[n_starts++] = st;
                        st.first = 0;
                    }
                    st.which = which; st.truncated = (which < 0);
                    out[n_starts++] = st;
                    if (first_pos == 0) { first_score = next_s; first_pos = k; first_j = j + 2; first_trunc = (first_pos == 0 && trunc); }
                    if (next_s > best_score) { best_score = next_s; best_pos = k; best_j = j + 2; best_trunc = st.truncated; }
                }
                if (j - 3 >= j_lo) { n6 = ((n6 << 2) | next_code()) & 63u; n6 = ((n6 << 2) | next_code()) & 63u; }    // buff[j-1], buff[j-2]
            }
        }
        if (a.use_first_start) { best_score = first_score; best_pos = first_pos; best_j = first_j; best_trunc = first_trunc; }
        (void)best_trunc;
        gmg_orf_result res;
        res.start_begin = (uint32_t)a.start_off[i];
        res.first_j = first_j; res.best_j = best_j; res.best_pos = best_pos;
        res.best_score = best_score;
        res.orf_is_truncated = trunc;
        if (first_j + 1 < a.min_gene_len) { res.n_starts = 0; res.is_tentative_gene = 0; res.gene_score = 0.0; }
        else {
            res.n_starts = n_starts;
            res.is_tentative_gene = best_score > a.start_threshold;
            res.gene_score = 100.0 * best_score / (best_j - 2);
        }
        a.results[i] = res;
        a.nst[i] = res.n_starts;
    }
}

// exact any-shape path: cumulative scores in scratch (k_seg_cum), then one lane per ORF scans them
__global__ __launch_bounds__(256) void k_orf_scan(OrfScanArgs a)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (uint64_t)gridDim.x * blockDim.x) {
        const gmg_orf orf = a.orfs[i];
        const uint64_t r_off = a.read_off[orf.read];
        const int L = (int)(a.read_off[orf.read + 1] - r_off);
        const int len = orf.orf_len;
        const bool fwd = orf.frame > 0;
        int lo, hi, k;
        bool trunc;
        if (fwd) {                                      // glimmer3.cc:1322-1332
            hi = orf.stop_position - 1;
            lo = hi - len;
            trunc = lo < 3 && a.allow_truncated;
            k = orf.stop_position - len - 2;
        } else {                                        // glimmer3.cc:1333-1343
            lo = orf.stop_position + 2;
            hi = lo + len;
            trunc = L - hi < 3 && a.allow_truncated;
            k = orf.stop_position + len + 4;
        }
        const double *score = a.score + a.cum_off[i];
        const double *indep = a.indep + a.cum_off[i];
        gmg_start *out = a.starts + a.start_off[i];
        uint32_t n_starts = 0;

        int first_pos = 0, best_pos = 0, first_j = 0, best_j = 0;
        bool first_trunc = false, best_trunc = false;
        double first_score = -DBL_MAX, best_score = -DBL_MAX;
        uint32_t codon = 0;
        const int lowest_j = a.min_gene_len - 3 < 3 ? a.min_gene_len - 3 : 3;
        int jm3 = (len - 1) % 3;
        for (int j = len - 1; j >= lowest_j && j >= 1; j--) {
            // buff[j]: forward ORF = S[hi-1-j] (Reverse_Transfer), reverse ORF = comp(S[lo+j]) (Complement_Transfer)
            const int code = fwd ? dev_code(a.packed, r_off + (uint64_t)(hi - 1 - j))
                                 : 3 - dev_code(a.packed, r_off + (uint64_t)(lo + j));
            codon = ((codon & 0xffu) << 4) | (1u << code);      // Codon_t::Shift_In (gene.cc:150-161)
            if (jm3 == 0) {
                int which = -1;                                 // Codon_t::Can_Be (gene.cc:39-66)
                for (int p = 0; p < a.n_pat; p++) {
                    const uint32_t x = codon & a.pat[p];
                    if ((x & 0xf00u) && (x & 0xf0u) && (x & 0x0fu)) { which = p; break; }
                }
                if ((which >= 0 || (first_pos == 0 && trunc)) && j + 3 >= a.min_gene_len) {
                    const double next_s = score[j - 1] - indep[j - 1];
                    // Ignore_Score_Len boost (glimmer3.cc:1464-1466) folded into the push
                    const double pushed = (j + 2 > a.ignore_score_len && next_s < 0.0) ? 0.0 : next_s;
                    gmg_start st;
                    st.score = pushed; st.j = j + 2; st.pos = k; st.first = (first_pos == 0);
                    if (which >= 0 && first_pos == 0 && trunc) {
                        st.which = -1; st.truncated = 1;
                        out[n_starts++] = st;
                        st.first = 0;
                    }
                    st.which = which; st.truncated = (which < 0);
                    out[n_starts++] = st;
                    if (first_pos == 0) {
                        first_score = next_s; first_pos = k; first_j = j + 2;
                        first_trunc = (first_pos == 0 && trunc);
                    }
                    if (next_s > best_score) { best_score = next_s; best_pos = k; best_j = j + 2; best_trunc = st.truncated; }
                }
            }
            k += fwd ? 1 : -1;
            jm3 = jm3 == 0 ? 2 : jm3 - 1;
        }
        if (a.use_first_start) { best_score = first_score; best_pos = first_pos; best_j = first_j; best_trunc = first_trunc; }
        (void)best_trunc;

        gmg_orf_result res;
        res.start_begin = (uint32_t)a.start_off[i];
        res.first_j = first_j; res.best_j = best_j; res.best_pos = best_pos;
        res.best_score = best_score;
        res.orf_is_truncated = trunc;
        if (first_j + 1 < a.min_gene_len) {             // glimmer3.cc:1431: the ORF is dropped
            res.n_starts = 0; res.is_tentative_gene = 0; res.gene_score = 0.0;
        } else {
            res.n_starts = n_starts;
            res.is_tentative_gene = best_score > a.start_threshold;     // glimmer3.cc:1468
            res.gene_score = 100.0 * best_score / (best_j - 2);        // glimmer3.cc:1489
        }
        a.results[i] = res;
        a.nst[i] = res.n_starts;
    }
}

// start lists -> back to back, in ORF order; start_begin now indexes the compact array
__global__ __launch_bounds__(256) void k_orf_compact(const gmg_start *sparse, const uint64_t *start_off, const uint32_t *coff,
                                                     gmg_orf_result *results, gmg_start *compact, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t cnt = results[i].n_starts, dst = coff[i];
        const gmg_start *src = sparse + start_off[i];
        for (uint32_t t = 0; t < cnt; t++) compact[dst + t] = src[t];
        results[i].start_begin = dst;
    }
}

extern "C" int gmg_orfs_upload(const gmg_reads *reads, const gmg_orf *orfs, uint64_t n, uint64_t *out_max_starts,
                               gmg_orf_batch **out)
{
    if (!reads || (!orfs && n) || !out) return gmg_set_error(GMG_EINVAL, "gmg_orfs_upload: NULL argument");
    { int rc0 = gmg_enter("gmg_orfs_upload"); if (rc0) return rc0; }
    std::vector<uint64_t> off(reads->n_reads + 1);
    GMG_HIP(hipMemcpy(off.data(), reads->d_off, off.size() * 8, hipMemcpyDeviceToHost));
    std::vector<gmg_segment> segs(n);
    std::vector<uint64_t> start_off(n + 1, 0);
    for (uint64_t i = 0; i < n; i++) {
        const gmg_orf &o = orfs[i];
        if (o.read >= reads->n_reads || o.frame == 0 || o.frame > 3 || o.frame < -3 || o.orf_len < 0)
            return gmg_set_error(GMG_ERANGE, "gmg_orfs_upload: ORF %llu: bad read / frame / length", (unsigned long long)i);
        const int64_t L = (int64_t)(off[o.read + 1] - off[o.read]);
        int64_t lo, hi;
        if (o.frame > 0) { hi = (int64_t)o.stop_position - 1; lo = hi - o.orf_len; }
        else { lo = (int64_t)o.stop_position + 2; hi = lo + o.orf_len; }
        if (lo < 0 || hi > L)
            return gmg_set_error(GMG_ERANGE, "gmg_orfs_upload: ORF %llu [%lld,%lld) leaves its read of length %lld "
                                 "(circular wrap-around is not supported)", (unsigned long long)i, (long long)lo,
                                 (long long)hi, (long long)L);
        segs[i].read = o.read;
        segs[i].lo = (uint32_t)lo;
        segs[i].len = (uint32_t)o.orf_len;
        segs[i].orient = o.frame > 0 ? GMG_REVERSED : GMG_COMPLEMENTED;     // glimmer3.cc:1328,1339
        start_off[i + 1] = start_off[i] + (uint64_t)o.orf_len / 3 + 2;     // one per in-frame codon + a truncated start
    }
    if (start_off[n] > (uint64_t)gmg_opt(GMG_OPT_MG_MAX_ENTRIES) || n > (uint64_t)gmg_opt(GMG_OPT_MG_MAX_ENTRIES))
        return gmg_set_error(GMG_ETOOBIG, "gmg_orfs_upload: %llu ORFs with room for %llu starts, gmg_orf_result.start_begin holds %lld: split the batch",
                             (unsigned long long)n, (unsigned long long)start_off[n], gmg_opt(GMG_OPT_MG_MAX_ENTRIES));
    gmg_orf_batch *b = new (std::nothrow) gmg_orf_batch();
    if (!b) return gmg_set_error(GMG_ENOMEM, "gmg_orfs_upload: out of host memory");
    memset(b, 0, sizeof *b);
    b->n = n;
    b->max_starts = start_off[n];
    b->reads = reads;
    b->reads_total = reads->total_bases;
    int rc = gmg_segments_upload(reads, segs.data(), n, nullptr, nullptr, &b->segs);
    if (rc) { delete b; return rc; }
    hipError_t e = hipMalloc((void **)&b->d_orfs, (n ? n : 1) * sizeof(gmg_orf));
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_start_off, (n + 1) * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_results, (n ? n : 1) * sizeof(gmg_orf_result));
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_starts, (b->max_starts ? b->max_starts : 1) * sizeof(gmg_start));
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_nst, (n + 1) * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_coff, (n + 1) * 4);
    if (e == hipSuccess) e = hipMemset(b->d_nst, 0, (n + 1) * 4);
    b->d_scan_tmp = nullptr;
    b->scan_tmp_bytes = 0;
    if (e == hipSuccess && n) e = hipMemcpy(b->d_orfs, orfs, n * sizeof(gmg_orf), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(b->d_start_off, start_off.data(), (n + 1) * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) { gmg_orf_batch_free(b); return gmg_set_error(GMG_ENOMEM, "gmg_orfs_upload: %s", hipGetErrorString(e)); }
    if (out_max_starts) *out_max_starts = b->max_starts;
    *out = b;
    return GMG_OK;
}

extern "C" int gmg_orf_batch_free(gmg_orf_batch *b)
{
    if (!b) return GMG_OK;
    if (b->segs) gmg_segments_free(b->segs);
    void *ptrs[] = {b->d_orfs, b->d_start_off, b->d_score, b->d_indep, b->d_results, b->d_starts, b->d_gene6, b->d_tmp, b->d_walk, b->d_heads, b->d_qpre, b->d_qneed,
                    b->d_nst, b->d_coff, b->d_scan_tmp, b->d_compact};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete b;
    return GMG_OK;
}

extern "C" int gmg_score_orfs(const gmg_model *gene, const gmg_model *nul, const gmg_reads *reads,
                              const gmg_orf_batch *b, const gmg_orf_params *prm, gmg_orf_result *results,
                              gmg_start *starts, void *stream)
{
    if ((b && b->n && !results) || (b && b->max_starts && !starts)) return gmg_set_error(GMG_EINVAL, "gmg_score_orfs: NULL argument");
    uint64_t n_starts = 0;
    const int rc = gmg_score_orfs_begin(gene, nul, reads, b, prm, &n_starts, stream);
    return rc ? rc : gmg_score_orfs_fetch(b, results, starts, stream);
}

extern "C" int gmg_score_orfs_fetch(const gmg_orf_batch *b, gmg_orf_result *results, gmg_start *starts, void *stream)
{
    if (!b || (b->n && !results) || (b->n_packed && !starts)) return gmg_set_error(GMG_EINVAL, "gmg_score_orfs_fetch: NULL argument");
    { int rc0 = gmg_enter("gmg_score_orfs_fetch"); if (rc0) return rc0; }
    if (b->n == 0) return GMG_OK;
    if (!b->scored) return gmg_set_error(GMG_EINVAL, "gmg_score_orfs_fetch: nothing scored yet (gmg_score_orfs_begin)");
    hipStream_t s = (hipStream_t)stream;
    GMG_HIP(hipMemcpyAsync(results, b->d_results, b->n * sizeof(gmg_orf_result), hipMemcpyDeviceToHost, s));
    if (b->n_packed) GMG_HIP(hipMemcpyAsync(starts, b->d_compact, (size_t)b->n_packed * sizeof(gmg_start), hipMemcpyDeviceToHost, s));
    GMG_HIP(hipStreamSynchronize(s));
    return GMG_OK;
}

extern "C" int gmg_score_orfs_begin(const gmg_model *gene, const gmg_model *nul, const gmg_reads *reads,
                                    const gmg_orf_batch *b, const gmg_orf_params *prm, uint64_t *out_n_starts, void *stream)
{
    if (!gene || !nul || !reads || !b || !prm || !out_n_starts)
        return gmg_set_error(GMG_EINVAL, "gmg_score_orfs: NULL argument");
    *out_n_starts = 0;
    if (prm->n_start_codons < 0 || prm->n_start_codons > 8 || prm->min_gene_len < 4)
        return gmg_set_error(GMG_EINVAL, "gmg_score_orfs: need 0..8 start codons and min_gene_len >= 4");
    // Cumulative_Score(buff, score, 1): frame 1 needs periodicity 1 or > 1 (src/ICM/icm.cc:367-369)
    if ((gene->dev.P != 1 && gene->dev.P < 2) || (nul->dev.P != 1 && nul->dev.P < 2))
        return gmg_set_error(GMG_EBADMODEL, "gmg_score_orfs: frame 1 outside the models' periodicity");
    if (b->reads != reads || b->reads_total != reads->total_bases)
        return gmg_set_error(GMG_EINVAL, "gmg_score_orfs: `reads` is not the batch these ORFs were uploaded against");
    { int rc0 = gmg_enter("gmg_score_orfs"); if (rc0) return rc0; }
    if (b->n == 0) return GMG_OK;
    hipStream_t s = (hipStream_t)stream;
    gmg_orf_batch *mb = const_cast<gmg_orf_batch *>(b);            // scratch is allocated on first use
    // option orfs_exact_path: 0 = the fastest path the models allow, 1 = the any-shape path, 2 = k_orf_fused even where the events path could run
    const bool fused = gene->dev.has_fast && gene->dev.D == 7 && gene->dev.P == 3 && gene->dev.W >= 3 && gene->dev.W <= 15 &&
                       nul->dev.has_dense && nul->dev.W == 3 && nul->dev.P == 3 && gmg_opt(GMG_OPT_ORFS_EXACT_PATH) != 1;
    // the events path: every sum of the batch exact in any order (the test of gmg_mg.hip's fused kernel; R = longest read + 2)
    bool events = false;
    if (fused && gmg_opt(GMG_OPT_ORFS_EXACT_PATH) == 0) {
        const int mn = gene->min_exp < nul->min_exp ? gene->min_exp : nul->min_exp, mx = gene->max_exp > nul->max_exp ? gene->max_exp : nul->max_exp;
        const uint64_t longest = reads->max_len ? reads->max_len : reads->total_bases;
        int clog = 0;
        while ((1ull << clog) < longest + 2) clog++;
        events = !gene->odd_values && !nul->odd_values && (mx < mn || clog + mx - mn <= 28) && nul->dev.dense_part == nul->dev.dense + 192;
    }
    if (events) {
        const uint64_t head_words = reads->total_bases / 32 + 4;
        const uint64_t meta_stride = reads->total_bases / 8 + reads->n_reads + 64;
        if (!mb->d_qpre) {
            uint16_t *qp = nullptr;
            uint8_t *qn = nullptr;
            hipError_t e = hipMalloc((void **)&qp, (size_t)2 * meta_stride * 2);
            if (e == hipSuccess) e = hipMalloc((void **)&qn, (size_t)2 * meta_stride);
            if (e != hipSuccess) { if (qp) (void)hipFree(qp); return gmg_set_error(GMG_ENOMEM, "gmg_score_orfs: scratch: %s", hipGetErrorString(e)); }
            mb->d_qpre = qp;
            mb->d_qneed = qn;
        }
        if (!mb->d_gene6 || !mb->d_walk || !mb->d_heads) {
            float *g6 = mb->d_gene6;
            double *wk = mb->d_walk;
            uint32_t *hd = mb->d_heads;
            hipError_t e = g6 ? hipSuccess : hipMalloc((void **)&g6, (size_t)6 * reads->total_bases * sizeof(float));
            if (e == hipSuccess && !wk) e = hipMalloc((void **)&wk, (size_t)2 * reads->total_bases * sizeof(double));
            if (e == hipSuccess && !hd) e = hipMalloc((void **)&hd, (size_t)2 * head_words * 4);
            if (e != hipSuccess) {
                if (g6 && !mb->d_gene6) (void)hipFree(g6);
                if (wk && !mb->d_walk) (void)hipFree(wk);
                return gmg_set_error(GMG_ENOMEM, "gmg_score_orfs: scratch: %s", hipGetErrorString(e));
            }
            mb->d_gene6 = g6;
            mb->d_walk = wk;
            mb->d_heads = hd;
        }
        int rc = gmg_launch_gene6(gene, reads, mb->d_gene6, s);
        if (rc) return gmg_set_error(rc, "gmg_score_orfs: gene-only pass refused the model");
        OrfWalkArgs wa;
        wa.packed = reads->d_packed;
        wa.read_off = reads->d_off;
        wa.n_reads = reads->n_reads;
        wa.total = reads->total_bases;
        wa.gene6 = mb->d_gene6;
        wa.null_dense = nul->dev.dense;
        wa.q = mb->d_walk;
        wa.heads = nullptr;
        wa.head_words = head_words;
        wa.start_set = 0;
        wa.q_pre = mb->d_qpre;
        wa.q_need = mb->d_qneed;
        wa.meta_stride = meta_stride;
        // orfs_walk8 = 1 (default): Q only where k_orf_events can ask for it -- the ORFs' HEAD positions (marked here), start codons, the reads' ends;
        // 2: every base (the form it is checked against); orfs_q_poison (tests): the array is filled with NaNs first, so that a read of an
        // entry that was not written cannot go unnoticed
        if (gmg_opt(GMG_OPT_ORFS_Q_POISON)) GMG_HIP(hipMemsetAsync(mb->d_walk, 0xff, (size_t)2 * reads->total_bases * sizeof(double), s));
        if (gmg_opt(GMG_OPT_ORFS_WALK8) == 1 || gmg_opt(GMG_OPT_ORFS_WALK8) >= 3) {
            for (uint32_t c = 0; c < 64; c++) {         // Codon_t::Can_Be (gene.cc:39-66) for every definite codon, as k_orf_events' s_which
                const uint32_t m = (1u << ((c >> 4) & 3u)) << 8 | (1u << ((c >> 2) & 3u)) << 4 | (1u << (c & 3u));
                for (int p = 0; p < prm->n_start_codons && p < 8; p++) {
                    uint32_t d = 0;
                    for (int k = 0; k < 3 && prm->start_codon[p][k]; k++) d = ((d & 0xffu) << 4) | ch_mask(prm->start_codon[p][k]);
                    const uint32_t x = m & d;
                    if ((x & 0xf00u) && (x & 0xf0u) && (x & 0x0fu)) wa.start_set |= 1ull << c;
                }
            }
            GMG_HIP(hipMemsetAsync(mb->d_heads, 0, (size_t)2 * head_words * 4, s));
            if (b->n) hipLaunchKernelGGL(k_orf_mark_heads, dim3((unsigned)((b->n + 255) / 256 < 256 * 16 ? (b->n + 255) / 256 : 256 * 16)), dim3(256), 0, s, b->d_orfs, b->n,
                                         reads->d_off, gene->dev.W, mb->d_heads, head_words);
            wa.heads = mb->d_heads;
        }
        const uint64_t waves = 2 * reads->n_reads, wblocks = (waves + 3) / 4;
        // (option orfs_walk8: the lane-on-eight-steps form, the default; 0: the lane-on-every-64th-step form it is checked against)
        if (gmg_opt(GMG_OPT_ORFS_WALK8) == 4) hipLaunchKernelGGL(k_orf_walk_sums8p<true>, dim3((unsigned)(wblocks < 256 * 64 ? wblocks : 256 * 64)), dim3(256), 0, s, wa);
        else if (gmg_opt(GMG_OPT_ORFS_WALK8) == 3) hipLaunchKernelGGL(k_orf_walk_sums8p<false>, dim3((unsigned)(wblocks < 256 * 64 ? wblocks : 256 * 64)), dim3(256), 0, s, wa);
        else if (gmg_opt(GMG_OPT_ORFS_WALK8)) hipLaunchKernelGGL(k_orf_walk_sums8, dim3((unsigned)(wblocks < 256 * 64 ? wblocks : 256 * 64)), dim3(256), 0, s, wa);
        else hipLaunchKernelGGL(k_orf_walk_sums, dim3((unsigned)(wblocks < 256 * 64 ? wblocks : 256 * 64)), dim3(256), 0, s, wa);
        GMG_HIP(hipGetLastError());
    } else if (fused) {
        if (!mb->d_gene6 || !mb->d_tmp) {                          // both or neither: a failed second allocation leaves nothing behind
            float *g6 = mb->d_gene6;
            OrfTmp *tmp = nullptr;
            hipError_t e = g6 ? hipSuccess : hipMalloc((void **)&g6, (size_t)6 * reads->total_bases * sizeof(float));
            if (e == hipSuccess) e = hipMalloc((void **)&tmp, (b->max_starts ? b->max_starts : 1) * sizeof(OrfTmp));
            if (e != hipSuccess) {
                if (g6 && !mb->d_gene6) (void)hipFree(g6);
                return gmg_set_error(GMG_ENOMEM, "gmg_score_orfs: scratch: %s", hipGetErrorString(e));
            }
            mb->d_gene6 = g6;
            mb->d_tmp = tmp;
        }
        int rc = gmg_launch_gene6(gene, reads, mb->d_gene6, s);
        if (rc) return gmg_set_error(rc, "gmg_score_orfs: gene-only pass refused the model");
    } else {
        const size_t tl = b->segs->total_len;
        if (!mb->d_score || !mb->d_indep) {
            double *sc = nullptr, *in = nullptr;
            hipError_t e = hipMalloc((void **)&sc, (tl ? tl : 1) * 8);
            if (e == hipSuccess) e = hipMalloc((void **)&in, (tl ? tl : 1) * 8);
            if (e != hipSuccess) {
                if (sc) (void)hipFree(sc);
                return gmg_set_error(GMG_ENOMEM, "gmg_score_orfs: scratch: %s", hipGetErrorString(e));
            }
            mb->d_score = sc;
            mb->d_indep = in;
        }
        int rc = gmg_launch_seg_cum(gene, reads, b->segs, gene->dev.P == 1 ? 0 : 1, b->d_score, nullptr, s);
        if (rc) return rc;
        rc = gmg_launch_seg_cum(nul, reads, b->segs, nul->dev.P == 1 ? 0 : 1, b->d_indep, nullptr, s);
        if (rc) return rc;
    }

    OrfScanArgs a;
    a.packed = reads->d_packed;
    a.read_off = reads->d_off;
    a.orfs = b->d_orfs;
    a.cum_off = b->segs->d_out_off;
    a.start_off = b->d_start_off;
    a.score = b->d_score;
    a.indep = b->d_indep;
    a.results = b->d_results;
    a.starts = b->d_starts;
    a.nst = b->d_nst;
    a.n = b->n;
    a.min_gene_len = prm->min_gene_len;
    a.allow_truncated = prm->allow_truncated;
    a.use_first_start = prm->use_first_start;
    a.ignore_score_len = prm->ignore_score_len;
    a.start_threshold = prm->start_threshold;
    a.n_pat = prm->n_start_codons;
    for (int p = 0; p < 8; p++) {
        uint32_t d = 0;                                 // Codon_t::Set_From (gene.cc:133-146)
        if (p < a.n_pat)
            for (int c = 0; c < 3 && prm->start_codon[p][c]; c++) d = ((d & 0xffu) << 4) | ch_mask(prm->start_codon[p][c]);
        a.pat[p] = d;
    }
    const uint64_t blocks = (b->n + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 256 * 16 ? blocks : 256 * 16);
    if (events) {
        OrfEventArgs ea;
        ea.sc = a;
        ea.gene = gene->dev;
        ea.nul = nul->dev;
        ea.gene6 = b->d_gene6;
        ea.q = b->d_walk;
        ea.q_pre = gmg_opt(GMG_OPT_ORFS_WALK8) == 4 ? b->d_qpre : nullptr;
        ea.q_need = b->d_qneed;
        ea.meta_stride = reads->total_bases / 8 + reads->n_reads + 64;
        ea.total = reads->total_bases;
        const size_t lds = (size_t)3 * gene->dev.cstride + (3 * 64 + 3 * 20) * sizeof(float);
        hipLaunchKernelGGL(k_orf_events, dim3(grid), dim3(256), lds, s, ea);
    } else if (fused) {
        OrfFusedArgs fa;
        fa.sc = a;
        fa.gene = gene->dev;
        fa.nul = nul->dev;
        fa.gene6 = b->d_gene6;
        fa.total = reads->total_bases;
        fa.tmp = b->d_tmp;
        const size_t lds = (size_t)3 * gene->dev.cstride + (3 * 64 + 3 * 20) * sizeof(float);
        hipLaunchKernelGGL(k_orf_fused, dim3(grid), dim3(256), lds, s, fa);
    } else {
        hipLaunchKernelGGL(k_orf_scan, dim3(grid), dim3(256), 0, s, a);
    }
    GMG_HIP(hipGetLastError());
    // only the used slots leave the GPU: prefix sum of the per-ORF counts, then pack the lists back to back
    GMG_HIP((gmg_scan_excl<uint32_t, uint32_t>(b->d_nst, b->d_coff, b->n + 1, s)));
    uint32_t total = 0;
    GMG_HIP(hipMemcpyAsync(&total, b->d_coff + b->n, 4, hipMemcpyDeviceToHost, s));
    GMG_HIP(hipStreamSynchronize(s));
    if (total > b->max_starts) return gmg_set_error(GMG_EHIP, "gmg_score_orfs: %u starts in %llu slots", total,
                                                    (unsigned long long)b->max_starts);
    if (total > b->compact_cap) {
        if (mb->d_compact) (void)hipFree(mb->d_compact);
        mb->d_compact = nullptr;
        mb->compact_cap = 0;
        const uint64_t cap = (uint64_t)total + total / 8 + 1024;
        hipError_t e = hipMalloc((void **)&mb->d_compact, cap * sizeof(gmg_start));
        if (e != hipSuccess) return gmg_set_error(GMG_ENOMEM, "gmg_score_orfs: start lists: %s", hipGetErrorString(e));
        mb->compact_cap = cap;
    }
    hipLaunchKernelGGL(k_orf_compact, dim3(grid), dim3(256), 0, s, b->d_starts, b->d_start_off, b->d_coff, b->d_results,
                       b->d_compact, b->n);
    GMG_HIP(hipGetLastError());
    mb->n_packed = total;
    mb->scored = 1;
    *out_n_starts = total;
    return GMG_OK;
}
