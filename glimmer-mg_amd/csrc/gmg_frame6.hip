// gmg_frame6.hip -- six-frame per-position gene - null scores of whole reads on gfx950.
// Replaces Score_All_Frames (src/Glimmer/glimmer-mg.cc:1468-1510) = 12 x ICM_t::Frame_Score
// (src/ICM/icm.cc:485-509) + the double subtraction, for every read of a batch.
//
// Output row f (0..2):   reversed read scored with sub-model f, stored at forward coordinates:
//     window chars w[k] = S[p+W-1-k], predicted base S[p]          (glimmer-mg.cc:1482-1494)
// Output row 3+f:        complemented read:
//     window chars w[k] = comp(S[p-(W-1)+k]), predicted comp(S[p]) (glimmer-mg.cc:1497-1509)
//
// The work is integer/byte arithmetic + table lookups (no MFMA); what must reach HBM is 48 B per
// base, so the goal is: every table access on-chip, every store a full line, and as few vector
// instructions per base as possible (profiles/r01_v5_*: the kernel is VALU-issue bound).
//
// Two launches per call, stream-ordered (the tail rides in the partial-window launch as extra blocks):
//   k_frame6t   main pass (LDS table swapping, below).  Treats the whole batch as ONE stream of bases and
//               scores every base with the full-window rule: no read boundaries, no branches.  Bases
//               whose window leaves their read (the first W-1 bases of either scoring buffer of each
//               read) get a meaningless value here.
//   k_frame6_generic  whole batches whose model shape has no fast path; the same exact code (f6_generic_range) scores
//               the last < 2048 bases of every batch as extra blocks of the k_frame6p launch (the main pass only
//               does full chunks).
//   k_frame6p   partial-window pass: one lane per (read, buffer position < W-1, strand) recomputes
//               exactly those bases with the reference's partial-window rule (icm.cc:807-842) and
//               overwrites them.  2(W-1) of every L bases (4.4 % at L = 500).
// DESIGN.md section 4.1 has the measurements that led here (gathers from L2 -> table swapping).

#include "gmg_device.h"
#include <stdlib.h>
#include <vector>

#ifndef GMG_F6_STAMPS
#define GMG_F6_STAMPS 0         // diagnostic build: per-wave cycle counts of the phases of a round (tools/f6_stamps.py); not in the product
#endif
#if GMG_F6_STAMPS
__device__ unsigned long long g_f6_stamps[1024 * 16 * 8];
extern "C" int gmg_debug_f6_stamps(unsigned long long *out, int n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_f6_stamps), (size_t)n * 8) == hipSuccess ? 0 : -1;
}
#define F6_STAMP(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); st_acc[i] += now_ - st_prev; st_prev = now_; } while (0)
#else
#define F6_STAMP(i) do { } while (0)
#endif

struct Frame6Args {
    GmgDevModel gene, nul;
    const uint32_t *packed;
    const uint64_t *off;
    const uint32_t *tile_read;
    uint64_t total, n_reads;
    uint64_t first, count;  // k_frame6_generic: range of bases to score
    uint32_t p_blocks;      // k_frame6p: blocks [0, p_blocks) do partial windows, the rest the tail [first, first+count)
    double *out;            // [6][stride] gene - null (the Frame_Scores table); stride = total for the public entry point
    uint64_t stride;        // distance between the rows of `out` in doubles (>= total)
    float *out_gene;        // gene-only mode (gmg_launch_gene6): [6][gstride] gene values as fp32
    uint64_t gstride;       // distance between the rows of out_gene in floats (>= total)
    // SUM mode (gmg_launch_strings_sum): per read and string the sum of the values whose window lies inside the read
    double *str_sums;       // [n_reads][2] (forward string, reverse complement), zeroed by the caller: atomically added to
    uint32_t uniform_len;   // > 0: every read has this length (read lookups by arithmetic)
    // MULTI mode (gmg_launch_gene6_groups; glimmer-mg -c: consecutive groups of reads, each under its own gene ICM)
    const struct F6Round *rounds = nullptr;     // k_frame6t: runs of <= K whole chunks that lie inside ONE group
    const GmgDevModel *gmodels = nullptr;       // [n_groups] the groups' models (all of one window width)
    const struct F6Range *ranges = nullptr;     // the bases no round covers: group edges inside a chunk, the batch tail
    const uint64_t *group_read = nullptr;       // [n_groups + 1] first read of every group (k_frame6p)
    uint32_t n_rounds = 0, n_ranges = 0, n_groups = 0;
};
struct F6Round { uint32_t chunk0; uint32_t ng; };      // ng = chunks of the round (1 .. 16) | its group << 5
#define F6_ROUND_N(r) ((r).ng & 31u)
#define F6_ROUND_GROUP(r) ((r).ng >> 5)
#define F6_MAX_GROUPS ((1 << 27) - 1)
struct F6Range { uint64_t first; uint32_t count, group; };

constexpr int f6_cstride(int dt) { return ((((1 << (2 * dt)) - 1) / 3) + 15) & ~15; }
constexpr int f6_level_base(int l) { return ((1 << (2 * l)) - 1) / 3; }

// SUM mode of k_frame6t: the four values one lane holds of a chunk -- the read itself (f0, f1) and its reverse complement (r0, r1)
// at bases g, g + 1 -- into the accumulators of their reads.  All positions are relative to the round's first base; span0 = the
// wave's first base.  roff: the reads' first bases (ragged batches; rel_a = read of the previous span of this wave, carried from
// chunk to chunk), or len > 0 for reads of one length (rem0 = the round's first base modulo len).  Reads are at least 86 bases long
// (the caller checks): a wave's 128 bases touch at most three of them.  Not inlined: sixteen copies of it in the unrolled second
// phase cost more registers than the call.
// sums over lanes: DPP moves of the two halves + one addition per step (no LDS traffic: the kernel's bottleneck is the LDS, a
// shuffle through ds_bpermute would add to it).  The order of the additions is whatever the butterfly makes it -- see SUM above
// for why that is allowed.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double f6_dpp_add(double x)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, ROW_MASK, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return x + __longlong_as_double((long long)((unsigned long long)hi << 32 | lo));      // (lanes the mask leaves out add 0)
}
// sum of x over the 16 lanes of a row, valid in every lane of the row
__device__ __forceinline__ double f6_row_sum(double x)
{
    x = f6_dpp_add<0xb1>(x);                            // quad_perm [1,0,3,2]
    x = f6_dpp_add<0x4e>(x);                            // quad_perm [2,3,0,1]
    x = f6_dpp_add<0x141>(x);                           // row_half_mirror
    x = f6_dpp_add<0x140>(x);                           // row_mirror
    return x;
}
__device__ __noinline__ void f6_sum_chunk_slow(double *sum, const int32_t *roff, uint32_t len, uint32_t len_magic, uint32_t rem0, int32_t wm1,
                                               uint32_t span0, uint32_t &rel_a, double f0, double r0, double f1, double r1)
{
    const uint32_t lane = threadIdx.x & 63u;
    int32_t a0, a1, a2, a3;
    if (len) {
        rel_a = __umulhi(span0 + rem0, len_magic);                  // (span0 + rem0) / len, < 2^17
        a0 = (int32_t)(rel_a * len) - (int32_t)rem0;
        a1 = a0 + (int32_t)len; a2 = a1 + (int32_t)len; a3 = a2 + (int32_t)len;
    } else {
        uint32_t ra = rel_a;
        while (roff[ra + 1] <= (int32_t)span0) ra++;
        rel_a = ra;
        a0 = roff[ra]; a1 = roff[ra + 1]; a2 = roff[ra + 2]; a3 = roff[ra + 3];
    }
    if ((int32_t)span0 >= a0 + wm1 && (int32_t)span0 + 128 <= a1 - wm1) {   // every value of the wave belongs to read A and counts
        // the four rows' sums go to the accumulator one by one: two butterfly steps (6 vector instructions) less per sum than one
        // sum of the wave (2.38 -> 2.30 ms per model)
        const double f = f6_row_sum(f0 + f1), r = f6_row_sum(r0 + r1);
        if ((lane & 15u) == 15u) { unsafeAtomicAdd(&sum[2 * rel_a], f); unsafeAtomicAdd(&sum[2 * rel_a + 1], r); }
        return;
    }
    // a read boundary, or a read's first / last W-1 bases, inside the span: what each lane's two bases give to read A and to
    // read B (the next one), summed over the wave; a third read (reads shorter than 128 bases) takes its values one by one
    double fa = 0.0, ra_ = 0.0, fb = 0.0, rb = 0.0;
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const int32_t g = (int32_t)(span0 + 2 * lane) + b;
        const uint32_t k = (g >= a1 ? 1u : 0u) + (g >= a2 ? 1u : 0u);
        const int32_t s0 = k == 0 ? a0 : k == 1 ? a1 : a2, s1 = k == 0 ? a1 : k == 1 ? a2 : a3;
        // the read itself: window = the W-1 bases in front; reverse complement at forward coordinates: the W-1 bases behind
        const double vf = g - s0 >= wm1 ? (b ? f1 : f0) : 0.0, vr = s1 - 1 - g >= wm1 ? (b ? r1 : r0) : 0.0;
        fa += k == 0 ? vf : 0.0; ra_ += k == 0 ? vr : 0.0;
        fb += k == 1 ? vf : 0.0; rb += k == 1 ? vr : 0.0;
        if (k == 2) {
            if (vf != 0.0) unsafeAtomicAdd(&sum[2 * (rel_a + 2)], vf);
            if (vr != 0.0) unsafeAtomicAdd(&sum[2 * (rel_a + 2) + 1], vr);
        }
    }
    fa = f6_row_sum(fa); ra_ = f6_row_sum(ra_); fb = f6_row_sum(fb); rb = f6_row_sum(rb);
    if ((lane & 15u) == 15u) {
        unsafeAtomicAdd(&sum[2 * rel_a], fa); unsafeAtomicAdd(&sum[2 * rel_a + 1], ra_);
        unsafeAtomicAdd(&sum[2 * rel_a + 2], fb); unsafeAtomicAdd(&sum[2 * rel_a + 3], rb);
    }
}

// Reads of one length: the common case -- the wave's 128 bases inside one read, away from its first / last W-1 -- stays in line
// (a wave-uniform test and two row sums); everything else goes through the call.
#ifndef GMG_F6_SUM_UNI
#define GMG_F6_SUM_UNI 1
#endif
// inclusive sums over the lanes 0 .. i of a row of 16
__device__ __forceinline__ double f6_row_scan(double x)
{
    x = f6_dpp_add<0x111>(x);                           // row_shr:1 (lanes without a source add 0)
    x = f6_dpp_add<0x112>(x);
    x = f6_dpp_add<0x114>(x);
    x = f6_dpp_add<0x118>(x);
    return x;
}
__device__ __forceinline__ void f6_sum_chunk(double *sum, const int32_t *roff, uint32_t len, uint32_t len_magic, uint32_t rem0, int32_t wm1,
                                             uint32_t span0, uint32_t &rel_a, double f0, double r0, double f1, double r1)
{
    if (GMG_F6_SUM_UNI && len >= 192) {
        // Reads of one length, long enough that a wave's 128 bases touch two of them at most: ONE path for every span, no call.
        // The values that do not count (a string's first W-1 positions) become 0, a row's lanes take running sums, and the row's
        // last lane adds the row's sum to the read of its last base; in the one row that holds the end of read A and the start of
        // read B the lane with A's last base moves A's share over: + share to A, - share to B (sums of at most 32 values: exact
        // under the caller's condition on the model's exponents, whatever the reads' totals come to).
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t ra = __umulhi(span0 + rem0, len_magic);
        rel_a = ra;
        const int32_t a1 = (int32_t)((ra + 1) * len) - (int32_t)rem0;          // first base of read B = A + 1
        const int32_t d0 = (int32_t)(span0 + 2 * lane) - a1, d1 = d0 + 1;      // < 0: read A
        const int32_t e0 = d0 < 0 ? d0 + (int32_t)len : d0, e1 = d1 < 0 ? d1 + (int32_t)len : d1;     // position in its read
        const int32_t hi = (int32_t)len - 1 - wm1;
        const double vf1 = e1 >= wm1 ? f1 : 0.0, vr1 = e1 <= hi ? r1 : 0.0;
        const double xf = (e0 >= wm1 ? f0 : 0.0) + vf1, xr = (e0 <= hi ? r0 : 0.0) + vr1;
        const double sf = f6_row_scan(xf), sr = f6_row_scan(xr);
        if ((lane & 15u) == 15u) {
            const uint32_t t = ra + (d1 >= 0 ? 1u : 0u);
            unsafeAtomicAdd(&sum[2 * t], sf); unsafeAtomicAdd(&sum[2 * t + 1], sr);
        }
        if (d0 < 0 && d1 >= -1 && d1 + 2 * (15 - (int32_t)(lane & 15u)) >= 0) {  // A's last base here, and bases of B further on in the row
            const double af = sf - (d1 >= 0 ? vf1 : 0.0), ar = sr - (d1 >= 0 ? vr1 : 0.0);
            unsafeAtomicAdd(&sum[2 * ra], af); unsafeAtomicAdd(&sum[2 * ra + 1], ar);
            unsafeAtomicAdd(&sum[2 * ra + 2], -af); unsafeAtomicAdd(&sum[2 * ra + 3], -ar);
        }
        return;
    }
    if (len) {
        const uint32_t ra = __umulhi(span0 + rem0, len_magic);
        const int32_t a0 = (int32_t)(ra * len) - (int32_t)rem0, a1 = a0 + (int32_t)len;
        if ((int32_t)span0 >= a0 + wm1 && (int32_t)span0 + 128 <= a1 - wm1) {
            rel_a = ra;
            const double f = f6_row_sum(f0 + f1), r = f6_row_sum(r0 + r1);
            if ((threadIdx.x & 15u) == 15u) { unsafeAtomicAdd(&sum[2 * ra], f); unsafeAtomicAdd(&sum[2 * ra + 1], r); }
            return;
        }
    }
    f6_sum_chunk_slow(sum, roff, len, len_magic, rem0, wm1, span0, rel_a, f0, r0, f1, r1);
}

// One descent in the completed tree of depth DT: C holds the window, tab the shift table in LDS.
// Returns the leaf index (0 .. 4^DT-1).
template <int DT>
__device__ __forceinline__ uint32_t f6_descend(const uint8_t *tab, uint32_t C, uint32_t shift0)
{
    uint32_t idx = __builtin_amdgcn_ubfe(C, shift0, 2);             // level 0: the root's shift is wave-uniform
#pragma unroll
    for (int l = 1; l < DT; l++) {
        const uint32_t sh = tab[f6_level_base(l) + idx];
        idx = (idx << 2) | __builtin_amdgcn_ubfe(C, sh, 2);
    }
    return idx;
}

// ---------------------------------------------------------------------------
// k_frame6t: the main pass with LDS table swapping (no gather from L2 at all).
//
// profiles/r01_v6_*: with ~41 % of the leaf lookups served by a 4-byte gather from L2, the per-CU
// vector-memory pipe (address coalescer: one divergent lane per cycle) and the L2 channels
// (1.3e9 requests per launch) are what the kernel waits for, and on gfx9 every load also waits for
// all older stores (one in-order vmcnt).  So this variant keeps ALL table accesses in LDS by
// holding one HALF of the sub-model's leaf values at a time (128 KiB of the 256 KiB).
//
// Which half: the one that holds the predicted base's values.  Half h = the two floats prob[2h], prob[2h+1] of every
// leaf row (gmg_model_upload lays the leaves out that way, GmgDevModel::chalf).  A lane's two scoring buffers predict
// S[p] (reversed buffer) and comp (S[p]) (complemented buffer) at the same base, and the complement flips the high
// bit of the 2-bit code (a 00 <-> t 11, c 01 <-> g 10): of the two values of a base EXACTLY ONE is in the resident
// half, whatever the read.  So per base one LDS read serves the pair in each phase (no dummy reads, no per-item
// flags: which of the two it was is a bit of the read itself), and both phases always carry the same load.
//
//   round = K chunks of this work-group (K x 2,048 bases; K = 16):
//     phase 1  (half h resident)   per chunk: contexts, four descents; per base the value of the buffer whose
//                                  predicted base is in h is read now, the other keeps its offset; 5 registers per
//                                  chunk and lane survive the phase (2 values, 2 offsets, 12 bits of read).
//     swap     barrier; the other half is streamed L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no registers in
//              between, so the round's state stays where it is); the packed words of the NEXT round are staged
//              into LDS, all requested before the first is waited for; barrier.
//     phase 2  (other half resident) per chunk: the missing values are read, everything is widened,
//                                  the null value subtracted, and the chunk is stored (full lines).
//   The next round starts with the half that is resident now, so there is ONE swap per round.
//
// Vector memory sees only coalesced traffic: the half reloads, the staged packed words and the
// output stores; phase 1 of the next round (pure LDS + VALU) overlaps the draining stores.
// Where a round's cycles go (tools/f6_stamps.py, profiles/r02_f6_stamps.txt): phase 1 52 %, waiting for the last wave
// of phase 1 19 %, swap 7 %, phase 2 22 % -- the kernel is bound by the LDS lookups and the vector instructions
// around them, not by HBM.
// ---------------------------------------------------------------------------
// GENE_ONLY: write the gene model's value alone, as fp32, to a.out_gene (input of the fused Score_Orfs scan,
// gmg_orfs.hip); the null model is not touched.
// STRINGS (with GENE_ONLY, periodicity-1 models): the two strings scoreReadsGlim / Score_String look at instead of
// the two scoring buffers of Score_All_Frames -- the read as it is (row 0: window = the W-1 bases in front of p,
// nothing complemented) and its reverse complement (row 1, stored at forward coordinates: window = the complements
// of the W-1 bases behind p).  Same geometry as the complemented / reversed buffers with the complement swapped;
// every work-group works on sub-model 0.
// SUM (with STRINGS): no per-base values leave the kernel.  Score_String adds its values one after the other (icm.cc:871-900);
// here a wave adds the 128 values it holds of one string by a shuffle tree, the work-group collects the reads' partial
// sums of a round in LDS and adds them to a.str_sums once per round.  The order differs from the reference's -- and the
// result does not, for every read whose sums the caller then accepts (k_string_finish, gmg_strings.hip): all values have
// one sign and are multiples of 2^(e_min - 150), so while |sum| < 2^(e_min - 150 + 53) every partial sum of every order is
// exact.  The first W-1 positions of either string (window outside the read) are left out here and added there.
// A worker's chunks are consecutive in this mode, so that a round is one run of 32,768 bases = a few dozen whole reads.
// MULTI (with GENE_ONLY): the batch is a sequence of read groups, each scored by its own gene model (glimmer-mg's classification
// mode loads one ICM per group of reads, glimmer-mg.cc:361-366).  The work is a list of rounds -- runs of at most K whole chunks
// that lie inside ONE group -- cut into consecutive shares, one per work-group; when a work-group's next round belongs to another
// group than its last it swaps the WHOLE model (both shift tables and the resident half: ~140 KiB from L2, once per group and
// work-group, against 128 KiB per round anyway).  One launch for any number of groups.
template <int BLOCK, int DT, int K, int DIAG, bool PAIR, bool GENE_ONLY, bool STRINGS = false, bool SUM = false, bool MULTI = false>
__global__ __launch_bounds__(BLOCK) void k_frame6t(Frame6Args a)
{
    constexpr int CS = f6_cstride(DT);
    constexpr uint32_t SPAN = 2 * BLOCK;                            // bases per chunk
    constexpr uint32_t RAWW = SPAN / 16 + 4;                        // packed words staged per chunk (1 before, 3 after)
    constexpr uint32_t LEAVES = 1u << (2 * DT);
    constexpr uint32_t HALF_BYTES = LEAVES * 8;                     // two floats per leaf
    extern __shared__ __attribute__((aligned(16))) uint8_t s_half[];   // [LEAVES][2] floats: first in LDS, offsets need no base
    __shared__ __attribute__((aligned(16))) uint8_t s_shr[CS];     // complemented buffer (rows 3+f): 2*mip
    __shared__ __attribute__((aligned(16))) uint8_t s_shf[CS];     // reversed buffer (rows f): 2*(W-1-mip)
    __shared__ __attribute__((aligned(16))) double s_nr[64];        // null model, complemented buffer
    __shared__ __attribute__((aligned(16))) double s_nf[64];        // null model, reversed buffer
    __shared__ __attribute__((aligned(16))) uint32_t s_raw[K * RAWW];   // packed words of the current round
    constexpr uint32_t NR_MAX = SUM ? 384 : 1;                      // reads of a round with an accumulator in LDS (more: global atomics)
    __shared__ double s_sum[SUM ? 2 * NR_MAX : 1];                  // [read of the round][string]
    __shared__ int32_t s_roff[SUM ? NR_MAX + 4 : 1];                // their first bases relative to the round's first (ragged batches)

    // MULTI: the work-groups of one XCD (blocks are dealt round-robin over the 8 XCDs: b % 8 labels it) take ONE consecutive share
    // of the rounds, all three sub-model types of them, and interleave inside it -- an XCD's L2 then holds the tables of the one
    // or two groups its work-groups are in (with a share per work-group all 64 groups are live at once: 64 MB against 4 MB of L2
    // per XCD, every half-table swap a miss: 4.16 against 3.72 ms per 1 M x 500 bp).  Types rotate with the XCD so that each type
    // gets the same number of work-groups overall.
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const int ftype = STRINGS ? 0 : MULTI ? (int)((slot + xcd) % 3u) : blockIdx.x % 3;
    const uint32_t worker = STRINGS ? blockIdx.x : blockIdx.x / 3, nworkers = STRINGS ? gridDim.x : gridDim.x / 3;
    uint32_t r_lo = 0, r_step = 1, r_count = 0;         // MULTI: rounds r_lo, r_lo + r_step, ... (r_count of them)
    if (MULTI) {
        auto cnt = [](uint32_t n, uint32_t c) { return n > c ? (n - c + 2u) / 3u : 0u; };     // s < n with s % 3 == c
        uint32_t before = 0, total = 0, team = 0;
        for (uint32_t x = 0; x < 8; x++) {
            const uint32_t slots = (gridDim.x + 7u - x) >> 3;       // blocks b < gridDim.x with b % 8 == x
            const uint32_t c = (uint32_t)(ftype + 3 * 8 - (int)x) % 3u, m = cnt(slots, c);
            if (x < xcd) before += m;
            if (x == xcd) team = m;
            total += m;
        }
        const uint32_t lo = (uint32_t)((uint64_t)a.n_rounds * before / total), hi = (uint32_t)((uint64_t)a.n_rounds * (before + team) / total);
        const uint32_t me = cnt(slot, (uint32_t)(ftype + 3 * 8 - (int)xcd) % 3u);                // my place in the team
        r_lo = lo + me;
        r_step = team;
        r_count = hi > r_lo ? (hi - r_lo + team - 1u) / team : 0u;
    }
    const int W = a.gene.W;                                         // (MULTI: every group's model has this width)
    const uint8_t *half_src = (const uint8_t *)(a.gene.chalf + (size_t)ftype * 2 * LEAVES * 2);   // [2][LEAVES][2] floats

    const uint64_t n_chunks = a.total / SPAN;                       // full chunks only
    // chunk j of this worker: worker + j * nworkers (neighbouring work-groups write neighbouring chunks), or, SUM mode, a
    // consecutive range of the batch
    const uint64_t per_worker = SUM ? (n_chunks + nworkers - 1) / nworkers : 0;
    const uint64_t chunk0 = SUM ? (uint64_t)worker * per_worker : worker, chunk_step = SUM ? 1 : nworkers;
    if (MULTI ? r_count == 0 : chunk0 >= n_chunks) return;
    const uint32_t n_mine = MULTI ? r_count * (uint32_t)K
                            : SUM ? (uint32_t)(n_chunks - chunk0 < per_worker ? n_chunks - chunk0 : per_worker)
                                  : (uint32_t)((n_chunks - worker + nworkers - 1) / nworkers);

    // every wave-instruction moves 1 KiB L2 -> LDS: lane l's 16 bytes land at the wave's LDS base + 16 l
    constexpr uint32_t NHALF = HALF_BYTES / 16 / BLOCK;             // 16-byte pieces per lane and half
    const uint32_t wbase = __builtin_amdgcn_readfirstlane(threadIdx.x & ~63u);
    auto half_issue = [&](uint32_t h) __attribute__((always_inline)) {
        const float4 *src = (const float4 *)(half_src + (size_t)h * HALF_BYTES);
#pragma unroll
        for (uint32_t i = 0; i < NHALF; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + i * BLOCK + threadIdx.x),
                                             (__attribute__((address_space(3))) void *)(s_half + (size_t)(i * BLOCK + wbase) * 16), 16, 0, 0);
    };
    auto half_landed = [&]() __attribute__((always_inline)) { __builtin_amdgcn_s_waitcnt(0x0F70); };   // vmcnt(0)
    // words [c*SPAN/16 - 1, c*SPAN/16 + RAWW - 1) of every chunk c of the round that starts at chunk j0 of this worker
    constexpr uint32_t NRAW = (K * RAWW + BLOCK - 1) / BLOCK;
    uint32_t raw_t[NRAW];
    auto raw_issue = [&](uint32_t j0) __attribute__((always_inline)) {
        F6Round nd = {0, 1};
        if (MULTI) nd = a.rounds[r_lo + (j0 / K) * r_step];
#pragma unroll
        for (uint32_t i = 0; i < NRAW; i++) {
            const uint32_t t = threadIdx.x + i * BLOCK;
            const uint32_t k = t / RAWW, w = t - k * RAWW;
            const uint32_t j = j0 + k < n_mine ? j0 + k : n_mine - 1;
            const uint64_t c = MULTI ? (uint64_t)nd.chunk0 + (k < F6_ROUND_N(nd) ? k : F6_ROUND_N(nd) - 1u) : chunk0 + (uint64_t)j * chunk_step;
            raw_t[i] = t < K * RAWW ? a.packed[c * (SPAN / 16) - 1 + w] : 0u;
        }
    };
    auto raw_commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (uint32_t i = 0; i < NRAW; i++) {
            const uint32_t t = threadIdx.x + i * BLOCK;
            if (t < K * RAWW) s_raw[t] = raw_t[i];
        }
    };

    // ---- fill LDS
    {
        const uint8_t *sh_src = a.gene.cshift + (size_t)ftype * a.gene.cstride;
        for (int i = threadIdx.x; i < CS && !MULTI; i += BLOCK) {   // (MULTI: with the first round's model, below)
            const uint8_t sh = sh_src[i];
            s_shr[i] = sh;
            s_shf[i] = (uint8_t)(2 * (W - 1) - sh);
        }
        if (threadIdx.x < 64) {
            const uint32_t i = threadIdx.x;
            const uint32_t mirrored = ((i & 3u) << 4) | (i & 12u) | (i >> 4);
            // both tables are indexed with plain (uncomplemented) read bases: s_nr[i] is the entry of the
            // complemented window i ^ 63, s_nf[i] the entry of the mirrored window
            s_nr[i] = GENE_ONLY ? 0.0 : (double)a.nul.dense[(size_t)ftype * 64 + (i ^ 63u)];
            s_nf[i] = GENE_ONLY ? 0.0 : (double)a.nul.dense[(size_t)ftype * 64 + mirrored];
        }
        raw_issue(0);
        if (!MULTI) half_issue(0);
        raw_commit();
        half_landed();
    }
    __syncthreads();
    uint32_t shift0_r = MULTI ? 0u : __builtin_amdgcn_readfirstlane((uint32_t)s_shr[0]);
    uint32_t shift0_f = MULTI ? 0u : __builtin_amdgcn_readfirstlane((uint32_t)s_shf[0]);
    uint32_t cur_group = 0xffffffffu;                               // MULTI: the group whose model is in LDS

    const uint32_t ctx_mask = (1u << (2 * W)) - 1u;                 // W <= 15
    const uint32_t sh_f = 2u * (uint32_t)(W - 1);                   // bit offset of S[p] in the window word
    const uint32_t lane_off = 2 * threadIdx.x;
    const uint32_t first_rel = lane_off + 16u - (uint32_t)(W - 1);
    const uint32_t wword = first_rel >> 4;
    const uint32_t wsh = 2u * (first_rel & 15u);
    double *const out_f = a.out + (uint64_t)ftype * a.stride;
    double *const out_r = a.out + (uint64_t)(3 + ftype) * a.stride;

    // ---- SUM mode: the reads of the round (kk * SPAN consecutive bases) and their accumulators
    uint64_t round_r0 = 0;                                          // read that holds the round's first base
    uint32_t round_rem0 = 0, sum_rel = 0;                           // uniform batches: that base's position in its read; ragged: see f6_sum_chunk
    const uint32_t L_u = a.uniform_len;
    const uint32_t L_magic = L_u > 1 ? (uint32_t)((0x100000000ull + L_u - 1) / L_u) : 0u;
    auto round_setup = [&](uint32_t j0) __attribute__((always_inline)) {       // all lanes; before the barrier that opens phase 2
        const uint64_t round_g0 = (chunk0 + j0) * SPAN;
        if (L_u) {
            round_r0 = round_g0 / L_u;
            round_rem0 = (uint32_t)(round_g0 - round_r0 * L_u);
        } else {
            uint64_t r = a.tile_read[round_g0 / GMG_TILE];          // read holding base round_g0 (a multiple of 1,024)
            while (a.off[r + 1] <= round_g0) r++;
            round_r0 = r;
            for (uint32_t i = threadIdx.x; i < NR_MAX + 4; i += BLOCK) {
                const uint64_t rr = r + i < a.n_reads ? r + i : a.n_reads;
                const int64_t rel = (int64_t)a.off[rr] - (int64_t)round_g0;
                s_roff[i] = rel > 0x3fffffff ? 0x3fffffff : (int32_t)rel;
            }
        }
        sum_rel = 0;
        for (uint32_t i = threadIdx.x; i < 2 * NR_MAX; i += BLOCK) s_sum[i] = 0.0;
    };
    auto sum_chunk = [&](uint32_t k, const double *v) __attribute__((always_inline)) {
        f6_sum_chunk(s_sum, s_roff, L_u, L_magic, round_rem0, W - 1, k * SPAN + (threadIdx.x & ~63u) * 2u, sum_rel, v[1], v[0], v[3], v[2]);
    };
    // the round's sums to the batch's (one wave-instruction per 32 reads: consecutive addresses)
    auto round_flush = [&]() __attribute__((always_inline)) {
        for (uint32_t i = threadIdx.x; i < 2 * NR_MAX; i += BLOCK) {
            const double x = s_sum[i];
            if (x != 0.0) unsafeAtomicAdd(&a.str_sums[2 * round_r0 + i], x);
        }
    };

#if GMG_F6_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = __builtin_readcyclecounter();
#endif
    uint32_t cur = 0;                                               // half resident in LDS
    for (uint32_t j0 = 0; j0 < n_mine; j0 += K) {
        uint32_t kk = n_mine - j0 < (uint32_t)K ? n_mine - j0 : (uint32_t)K;   // chunks in this round
        uint64_t round_chunk0 = 0;
        if (MULTI) {
            const F6Round rd = a.rounds[r_lo + (j0 / K) * r_step];
            kk = F6_ROUND_N(rd);
            round_chunk0 = rd.chunk0;
            if (F6_ROUND_GROUP(rd) != cur_group) {                            // another group: its model instead of the one in LDS
                cur_group = F6_ROUND_GROUP(rd);
                __syncthreads();                                    // every wave has left the previous round's second phase
                const GmgDevModel &gm = a.gmodels[cur_group];
                const uint8_t *sh_src = gm.cshift + (size_t)ftype * gm.cstride;
                for (int i = threadIdx.x; i < CS; i += BLOCK) {
                    const uint8_t sh = sh_src[i];
                    s_shr[i] = sh;
                    s_shf[i] = (uint8_t)(2 * (W - 1) - sh);
                }
                half_src = (const uint8_t *)(gm.chalf + (size_t)ftype * 2 * LEAVES * 2);
                half_issue(cur);
                half_landed();
                __syncthreads();
                shift0_r = __builtin_amdgcn_readfirstlane((uint32_t)s_shr[0]);
                shift0_f = __builtin_amdgcn_readfirstlane((uint32_t)s_shf[0]);
            }
        }
        // per chunk and base pair (g0, g0 + 1): the value that was in the resident half and the offset of the other
        uint32_t have[K][2], want[K][2];
        // per chunk the read bases S[g0-2 .. g0+3] (12 bits: the null-model windows, and which buffer of a base was resident:
        // the high bit of S[g0] / S[g0+1]); two chunks per register
        uint32_t meta[(K + 1) / 2];

        // ---- phase 1
#pragma unroll
        for (int k = 0; k < K; k++) {
            if ((uint32_t)k < kk) {
                const uint32_t *w = s_raw + k * RAWW + wword;
                const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
                // window of this lane: field i (bits 2i, 2i+1) = base g0-(W-1)+i, 32 fields
                const uint32_t xl = __builtin_amdgcn_alignbit(w1, w0, wsh);
                const uint32_t xh = __builtin_amdgcn_alignbit(w2, w1, wsh);
                uint32_t C[4];
                const uint32_t flip_r = STRINGS ? 0u : ctx_mask, flip_f = STRINGS ? ctx_mask : 0u;
                C[1] = (xl ^ flip_r) & ctx_mask;                                    // complemented buffer (STRINGS: the read itself)
                C[3] = (__builtin_amdgcn_alignbit(xh, xl, 2) ^ flip_r) & ctx_mask;
                C[0] = (__builtin_amdgcn_alignbit(xh, xl, sh_f) ^ flip_f) & ctx_mask;   // reversed buffer, natural order (STRINGS: complemented)
                C[2] = (__builtin_amdgcn_alignbit(xh, xl, sh_f + 2) ^ flip_f) & ctx_mask;
                uint32_t idx[4];
                idx[0] = f6_descend<DT>(s_shf, C[0], shift0_f);
                idx[1] = f6_descend<DT>(s_shr, C[1], shift0_r);
                idx[2] = f6_descend<DT>(s_shf, C[2], shift0_f);
                idx[3] = f6_descend<DT>(s_shr, C[3], shift0_r);
                // the six bases S[g0-2 .. g0+3] = fields W-3 .. W+2, uncomplemented: field 2 is S[g0], field 3 S[g0+1]
                const uint32_t six = __builtin_amdgcn_alignbit(xh, xl, sh_f - 4) & 0xfffu;
#pragma unroll
                for (int b = 0; b < 2; b++) {                       // base g0 + b: items 2b (reversed buffer) and 2b + 1 (complemented)
                    // predicted bases: C[2b] & 3 and C[2b+1] >> sh_f, complements of each other; byte offset of a value inside
                    // its half: leaf * 8 + (low bit of the predicted base) * 4
                    const uint32_t tf = (idx[2 * b] << 3) | ((C[2 * b] & 1u) << 2);
                    const uint32_t tr = (idx[2 * b + 1] << 3) | (((C[2 * b + 1] >> sh_f) & 1u) << 2);
                    const bool f_here = ((C[2 * b] >> 1) & 1u) == cur;             // the reversed buffer's value is resident
                    const uint32_t got = (DIAG & 2) ? tf : __float_as_uint(*(const float *)(s_half + (f_here ? tf : tr)));
                    have[k][b] = got;
                    want[k][b] = f_here ? tr : tf;
                }
                if (k & 1) meta[k >> 1] |= six << 16; else meta[k >> 1] = six;
            }
        }

        // ---- swap halves; stage the next round's packed words
        F6_STAMP(0);                                                // phase 1
        __syncthreads();
        F6_STAMP(1);                                                // wait at barrier A
        cur ^= 1u;
        if (j0 + K < n_mine) raw_issue(j0 + K);
        half_issue(cur);
        if (j0 + K < n_mine) raw_commit();
        if (SUM) round_setup(j0);
        half_landed();
        F6_STAMP(2);                                                // swap work
        __syncthreads();
        F6_STAMP(3);                                                // wait at barrier B

        // ---- phase 2
#pragma unroll
        for (int k = 0; k < K; k++) {
            if ((uint32_t)k < kk) {
                const uint32_t mt = meta[k >> 1];
                const int mb = (k & 1) * 16;                        // (a constant in the unrolled loop: folded into the bit-field offsets)
                double v[4];
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    const uint32_t got = (DIAG & 2) ? want[k][b] : __float_as_uint(*(const float *)(s_half + want[k][b]));
                    // which buffer was resident in phase 1: the high bit of S[g0 + b] (STRINGS: of its complement) against the half
                    // of then, i.e. the other one than now
                    const uint32_t hb = (mt >> (mb + 5 + 2 * b)) & 1u;
                    const bool f_was_here = (STRINGS ? hb ^ 1u : hb) != cur;
                    const float gf = __uint_as_float(f_was_here ? have[k][b] : got);
                    const float gr = __uint_as_float(f_was_here ? got : have[k][b]);
                    // item 2b: S[g0+b .. g0+b+2] reversed buffer; item 2b+1: S[g0+b-2 .. g0+b] complemented
                    if (GENE_ONLY) {                                // the gene model's value alone: no null-model lookups
                        v[2 * b] = (double)gf;
                        v[2 * b + 1] = (double)gr;
                    } else {
                        const double nf = s_nf[__builtin_amdgcn_ubfe(mt, mb + 4 + 2 * b, 6)];
                        const double nr = s_nr[__builtin_amdgcn_ubfe(mt, mb + 2 * b, 6)];
                        v[2 * b] = (double)gf - nf;                 // glimmer-mg.cc:1493,1508
                        v[2 * b + 1] = (double)gr - nr;
                    }
                }
                const uint64_t chunk = MULTI ? round_chunk0 + (uint64_t)k : chunk0 + (uint64_t)(j0 + k) * chunk_step;
                if (SUM) {
                    sum_chunk((uint32_t)k, v);
                } else if (GENE_ONLY) {
                    // v[c] = gene value exactly (the null tables hold zeros); rows of floats, 8-byte stores
                    float *gf = a.out_gene + (uint64_t)(STRINGS ? 1 : ftype) * a.gstride + chunk * SPAN;
                    float *gr = a.out_gene + (uint64_t)(STRINGS ? 0 : 3 + ftype) * a.gstride + chunk * SPAN;
                    if (PAIR) {
                        typedef float f2 __attribute__((ext_vector_type(2)));
                        const f2 x0 = {(float)v[0], (float)v[2]}, x1 = {(float)v[1], (float)v[3]};
                        __builtin_nontemporal_store(x0, (f2 *)(gf + lane_off));
                        __builtin_nontemporal_store(x1, (f2 *)(gr + lane_off));
                    } else {
                        gf[lane_off] = (float)v[0]; gf[lane_off + 1] = (float)v[2];
                        gr[lane_off] = (float)v[1]; gr[lane_off + 1] = (float)v[3];
                    }
                } else if (DIAG & 1) {
                    if (v[0] + v[1] + v[2] + v[3] == 1.2345e300) a.out[lane_off] = v[0];
                } else {
                    double *pf = out_f + chunk * SPAN, *pr = out_r + chunk * SPAN;   // wave-uniform bases + lane offset
                    // non-temporal (streaming) stores: the output is written once and never re-read here;
                    // measured -9 % on the whole call against plain stores (profiles/r01_v9_*)
                    if (PAIR) {
                        typedef double d2 __attribute__((ext_vector_type(2)));
                        const d2 x0 = {v[0], v[2]}, x1 = {v[1], v[3]};
                        __builtin_nontemporal_store(x0, (d2 *)(pf + lane_off));
                        __builtin_nontemporal_store(x1, (d2 *)(pr + lane_off));
                    } else {
                        __builtin_nontemporal_store(v[0], pf + lane_off);
                        __builtin_nontemporal_store(v[2], pf + lane_off + 1);
                        __builtin_nontemporal_store(v[1], pr + lane_off);
                        __builtin_nontemporal_store(v[3], pr + lane_off + 1);
                    }
                }
            }
        }
        if (SUM) {                                                  // (the next write to s_sum is behind two more barriers)
            __syncthreads();
            round_flush();
        }
        F6_STAMP(4);                                                // phase 2
    }
#if GMG_F6_STAMPS
    if ((threadIdx.x & 63u) == 0 && blockIdx.x < 1024)
        for (int i = 0; i < 8; i++) g_f6_stamps[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + i] = st_acc[i];
#endif
}

// ---------------------------------------------------------------------------
// k_frame6p: partial windows.  One lane per (read, z): z < W-1 is buffer position j = z of the
// complemented buffer (rows 3+f, base p = j); z >= W-1 is position j = z-(W-1) of the reversed
// buffer (rows f, base p = L-1-j).  Uses the partial-window rule of icm.cc:807-842: stop as soon as
// the context position named by a node lies before the buffer, i.e. mip < (W-1)-j, which in the
// completed tree reads "shift byte < 2*((W-1)-j)"; crow holds the right row for inner nodes too.
// ---------------------------------------------------------------------------

// Exact plain descent on the original tables for both models: bases [a.first, a.first + a.count), lane i0 of every n_lanes.
__device__ __forceinline__ void f6_generic_range(const Frame6Args &a, const GmgDevModel &gene, uint64_t first, uint64_t count,
                                                 uint64_t i0, uint64_t n_lanes)
{
    for (uint64_t i = i0; i < count; i += n_lanes) {
        const uint64_t g = first + i;
        uint64_t r = a.tile_read[g / GMG_TILE];
        uint64_t r_end = a.off[r + 1];
        while (g >= r_end) { r++; r_end = a.off[r + 1]; }
        const uint64_t r_off = a.off[r];
        const int L = (int)(r_end - r_off);
        const int p = (int)(g - r_off);
        DevBuf bf = dev_make_buf(a.packed, r_off, 0, (uint32_t)L, GMG_REVERSED);
        DevBuf br = dev_make_buf(a.packed, r_off, 0, (uint32_t)L, GMG_COMPLEMENTED);
        for (int f = 0; f < 3; f++) {
            if (a.out_gene) {
                a.out_gene[(uint64_t)f * a.gstride + g] = dev_score(gene, bf, L - 1 - p, f);
                a.out_gene[(uint64_t)(3 + f) * a.gstride + g] = dev_score(gene, br, p, f);
                continue;
            }
            a.out[(uint64_t)f * a.stride + g] =
                (double)dev_score(gene, bf, L - 1 - p, f) - (double)dev_score(a.nul, bf, L - 1 - p, f);
            a.out[(uint64_t)(3 + f) * a.stride + g] =
                (double)dev_score(gene, br, p, f) - (double)dev_score(a.nul, br, p, f);
        }
    }
}
__device__ __forceinline__ void f6_generic_range(const Frame6Args &a, uint64_t i0, uint64_t n_lanes)
{
    f6_generic_range(a, a.gene, a.first, a.count, i0, n_lanes);
}

// 32 lanes per read (z = lane & 31 < 2(W-1) <= 28 active): no division, read offsets broadcast.
// The kernel is a chain of dependent latencies (offsets -> packed words -> D LDS steps -> row gather ->
// store), so every lane keeps U reads in flight (U x 3 interleaved descents).
// DT > 0: depth known at compile time.
// GENE: the gene model's value alone, as fp32, into a.out_gene (the complete gene-only table of gmg_launch_gene6_full).
// MULTI (with GENE): groups of reads under their own models (see k_frame6t<.., MULTI>): block b takes the reads
// [n_reads b / p_blocks, n_reads (b + 1) / p_blocks) -- a consecutive share, so it meets one group, seldom two -- and loads the
// shift tables of the group it is in; the blocks behind p_blocks score the ranges no round of the main pass covers (eight blocks
// per range) with the exact any-shape code.
template <int DT, int U, bool GENE = false, bool MULTI = false>
__global__ __launch_bounds__(256) void k_frame6p(Frame6Args a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_shift[];   // [3][cstride] completed-tree shifts
    if (MULTI) {                                        // the blocks of the ranges come FIRST: their long dependent chains run under the rest
        if (blockIdx.x < 8 * a.n_ranges) {
            const F6Range rg = a.ranges[blockIdx.x / 8];
            f6_generic_range(a, a.gmodels[rg.group], rg.first, rg.count, (uint64_t)(blockIdx.x % 8) * blockDim.x + threadIdx.x, (uint64_t)8 * blockDim.x);
            return;
        }
    } else if (blockIdx.x >= a.p_blocks) {
        // the last < 2,048 bases of the batch (the main pass does whole chunks), in the same launch.  The partial-window
        // positions in there are written by both kinds of blocks, with identical bits.
        f6_generic_range(a, (uint64_t)(blockIdx.x - a.p_blocks) * blockDim.x + threadIdx.x, (uint64_t)(gridDim.x - a.p_blocks) * blockDim.x);
        return;
    }
    const int cstride = a.gene.cstride;
    const float *crow = a.gene.crow;
    int ctot = a.gene.ctot;
    if (!MULTI) {
        for (int i = threadIdx.x * 16; i < 3 * cstride; i += 256 * 16)
            *(uint4 *)(s_shift + i) = *(const uint4 *)(a.gene.cshift + i);
        __syncthreads();
    }

    const int W = a.gene.W, D = DT > 0 ? DT : a.gene.D, Wn = a.nul.W;
    const uint32_t Z = 2u * (uint32_t)(W - 1);
    const uint32_t ctx_mask = (W >= 16) ? 0xffffffffu : ((1u << (2 * W)) - 1u);
    const int n_dense = 1 << (2 * Wn);
    const int n_part = a.nul.n_dense_part;
    const uint32_t z = threadIdx.x & 31u;
    const bool rev_buf = z >= (uint32_t)(W - 1);                    // reversed buffer -> rows f
    const int j = rev_buf ? (int)z - (W - 1) : (int)z;              // position in the scoring buffer
    const int thr2 = 2 * ((W - 1) - j);                             // > 0: j < W-1
    const uint64_t reads_per_pass = (uint64_t)a.p_blocks * (256 / 32);

    // null-model slot of this lane's buffer position (same for every read)
    const float *ntab = (j >= Wn - 1) ? a.nul.dense : a.nul.dense_part;
    const int nstride = (j >= Wn - 1) ? n_dense : n_part;

    // the reads of this block: every reads_per_pass-th one from the block's own on, or (MULTI) its share group by group
    const uint32_t pb = MULTI ? blockIdx.x - 8 * a.n_ranges : blockIdx.x;
    const uint64_t share_lo = MULTI ? a.n_reads * pb / a.p_blocks : 0, share_hi = MULTI ? a.n_reads * (pb + 1) / a.p_blocks : a.n_reads;
    uint32_t grp = 0;
    if (MULTI) {
        uint32_t lo = 0, hi = a.n_groups;                           // the group that holds read share_lo
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (a.group_read[mid] <= share_lo) lo = mid; else hi = mid; }
        grp = lo;
    }
    for (uint64_t part_lo = share_lo; part_lo < share_hi;) {
    uint64_t part_hi = share_hi;
    if (MULTI) {
        while (a.group_read[grp + 1] <= part_lo) grp++;
        if (a.group_read[grp + 1] < part_hi) part_hi = a.group_read[grp + 1];
        const GmgDevModel &gm = a.gmodels[grp];
        __syncthreads();                                            // (the previous part's lookups are done)
        for (int i = threadIdx.x * 16; i < 3 * cstride; i += 256 * 16)
            *(uint4 *)(s_shift + i) = *(const uint4 *)(gm.cshift + i);
        __syncthreads();
        crow = gm.crow;
        ctot = gm.ctot;
    }
    const uint64_t r_step = MULTI ? 256 / 32 : reads_per_pass;
    for (uint64_t r0 = (MULTI ? part_lo : (uint64_t)blockIdx.x * (256 / 32)) + (threadIdx.x >> 5); r0 < part_hi; r0 += U * r_step) {
        bool live[U];
        uint64_t g[U];
        uint32_t C[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t r = r0 + u * r_step;
            const bool in = r < part_hi;
            const uint64_t rq = in ? r : part_hi - 1;
            const uint64_t r_off = a.off[rq];
            const int L = (int)(a.off[rq + 1] - r_off);
            live[u] = in && z < Z && j < L;                         // idle lanes; reads shorter than W-1
            const int p = rev_buf ? L - 1 - j : j;
            g[u] = r_off + (uint64_t)(live[u] ? p : 0);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t x = dev_window_bits(a.packed, (int64_t)g[u] - (W - 1));
            // window char k at bits 2k; chars that fall before the buffer are never looked at
            C[u] = rev_buf ? dev_reverse_fields((uint32_t)(x >> (2 * (W - 1))) & ctx_mask, W)
                           : (((uint32_t)x & ctx_mask) ^ ctx_mask);
        }
        uint32_t idx[U][3], node[U][3];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int f = 0; f < 3; f++) { idx[u][f] = 0; node[u][f] = 0xffffffffu; }
        uint32_t lvl = 0, width = 1;
#pragma unroll
        for (int l = 0; l < (DT > 0 ? DT : 12); l++) {
            if (DT == 0 && l >= D) break;
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int f = 0; f < 3; f++) {
                    const uint32_t sh = s_shift[f * cstride + lvl + idx[u][f]];
                    if (node[u][f] == 0xffffffffu && (int)sh < thr2) node[u][f] = lvl + idx[u][f];
                    idx[u][f] = (idx[u][f] << 2) + ((C[u] >> sh) & 3u);
                }
            lvl += width;
            width <<= 2;
        }
        float gv[U][3], nv[U][3];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t pred = (C[u] >> (2 * (W - 1))) & 3u;
            const uint32_t nslot = (j >= Wn - 1) ? (C[u] >> (2 * (W - Wn)))
                                                 : (C[u] >> (2 * (W - 1 - j))) + (((1u << (2 * (j + 1))) - 4u) / 3u);
#pragma unroll
            for (int f = 0; f < 3; f++) {
                if (node[u][f] == 0xffffffffu) node[u][f] = lvl + idx[u][f];
                gv[u][f] = crow[((size_t)f * ctot + node[u][f]) * 4 + pred];
                nv[u][f] = GENE ? 0.0f : ntab[(size_t)f * nstride + nslot];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            if (live[u])
#pragma unroll
                for (int f = 0; f < 3; f++) {
                    if (GENE)
                        a.out_gene[(uint64_t)((rev_buf ? 0 : 3) + f) * a.gstride + g[u]] = gv[u][f];
                    else
                        __builtin_nontemporal_store((double)gv[u][f] - (double)nv[u][f],
                                                    a.out + (uint64_t)((rev_buf ? 0 : 3) + f) * a.stride + g[u]);
                }
    }
    part_lo = part_hi;
    }
}

// ---------------------------------------------------------------------------
// Any-shape kernel: exact plain descent on the original tables for both models.  Scores bases
// [first, first+count) of the batch: the tail the main pass leaves, or everything when the gene
// model has no completed tree / the null model is not the width-3 one (same results, slower).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_frame6_generic(Frame6Args a)
{
    f6_generic_range(a, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, (uint64_t)gridDim.x * blockDim.x);
}

static int launch_generic(Frame6Args a, uint64_t first, uint64_t count, hipStream_t s)
{
    if (count == 0) return GMG_OK;
    a.first = first;
    a.count = count;
    const uint64_t blocks = (count + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 256 * 16 ? blocks : 256 * 16);
    hipLaunchKernelGGL(k_frame6_generic, dim3(grid), dim3(256), 0, s, a);
    GMG_HIP(hipGetLastError());
    return GMG_OK;
}

int gmg_launch_frame6(const gmg_model *gene, const gmg_model *nul, const gmg_reads *reads, double *d_out,
                      hipStream_t s)
{
    return gmg_launch_frame6_strided(gene, nul, reads, d_out, reads->total_bases, s);
}

// rows of the table `stride` doubles apart (the mg pipeline pads its own table so that every row starts on a 128-byte
// line whatever the number of bases: 16-byte stores, full lines)
int gmg_launch_frame6_strided(const gmg_model *gene, const gmg_model *nul, const gmg_reads *reads, double *d_out,
                              uint64_t stride, hipStream_t s)
{
    Frame6Args a;
    a.gene = gene->dev;
    a.nul = nul->dev;
    a.packed = reads->d_packed;
    a.off = reads->d_off;
    a.tile_read = reads->d_tile_read;
    a.total = reads->total_bases;
    a.n_reads = reads->n_reads;
    a.first = 0;
    a.count = 0;
    a.out = d_out;
    a.stride = stride;
    a.out_gene = nullptr;
    a.gstride = 0;
    a.str_sums = nullptr;
    a.uniform_len = 0;
    a.str_sums = nullptr;
    a.uniform_len = 0;

    // the specialised path: completed tree of depth 7 (DEFAULT_MODEL_DEPTH) and the width-3 null model
    const bool fast = gene->dev.has_fast && gene->dev.D == 7 && nul->dev.has_dense && nul->dev.W == 3 &&
                      gene->dev.W >= 3 && gene->dev.W <= 15;
    if (!fast) return launch_generic(a, 0, a.total, s);

    constexpr int BLOCK = 1024, DT = 7;
    constexpr uint32_t SPAN = 2 * BLOCK;
    constexpr int KR = 16;                                         // chunks per round (8 / 12 / 20 measured slower)
    const uint64_t n_chunks = a.total / SPAN;
    const int diag = (int)gmg_opt(GMG_OPT_DIAG);
    const bool pair = (a.stride & 1) == 0;          // every row 16-byte aligned
    if (n_chunks > 0) {
        int dev = 0, n_cu = 256;
        GMG_HIP(hipGetDevice(&dev));
        GMG_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        // persistent grid: one work-group per CU, a multiple of the 3 sub-model types
        unsigned nworkers = (unsigned)(n_cu / 3);
        if (nworkers < 1) nworkers = 1;
        if (nworkers > n_chunks) nworkers = (unsigned)n_chunks;
        const unsigned grid = 3 * nworkers;
        const size_t lds = ((size_t)1 << (2 * DT)) * 8;            // dynamic part: half of the leaf values
#define GMG_LAUNCH_F6T(DIAG_, P_)                                                                       \
    do {                                                                                                \
        GMG_HIP(hipFuncSetAttribute((const void *)k_frame6t<BLOCK, DT, KR, DIAG_, P_, false>,           \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));             \
        hipLaunchKernelGGL((k_frame6t<BLOCK, DT, KR, DIAG_, P_, false>), dim3(grid), dim3(BLOCK), lds, s, a); \
    } while (0)
        if (diag == 0) { if (pair) GMG_LAUNCH_F6T(0, true); else GMG_LAUNCH_F6T(0, false); }
#ifdef GMG_ABLATIONS
        else if (diag == 1 && pair) GMG_LAUNCH_F6T(1, true);
        else if (diag == 2 && pair) GMG_LAUNCH_F6T(2, true);
        else if (diag == 3 && pair) GMG_LAUNCH_F6T(3, true);
#endif
        else return gmg_set_error(GMG_EINVAL, "the ablation kernel diag=%d is not in this build", diag);
#undef GMG_LAUNCH_F6T
        GMG_HIP(hipGetLastError());
    }
    // partial windows of every read + the last, partial chunk: one launch
    if (a.n_reads > 0) {
        const uint64_t blocks = (a.n_reads + 7) / 8;                // 8 reads (32 lanes each) per block
        const unsigned grid = (unsigned)(blocks < 256 * 8 ? blocks : 256 * 8);
        const size_t lds_p = (size_t)3 * a.gene.cstride;
        a.first = n_chunks * SPAN;
        a.count = a.total - n_chunks * SPAN;
        a.p_blocks = grid;
        const unsigned tail_blocks = (unsigned)((a.count + 255) / 256);
        hipLaunchKernelGGL((k_frame6p<7, 4>), dim3(grid + tail_blocks), dim3(256), lds_p, s, a);
        GMG_HIP(hipGetLastError());
    }
    return GMG_OK;
}

// Gene-only per-position values of every base, fp32, rows [6][total] like Frame_Scores: the values
// ICM_t::Frame_Score(buffer, f) gives with the gene model alone for the reversed (rows 0-2) and the
// complemented (rows 3-5) read.  Bases whose window leaves their read hold a meaningless value
// (no partial-window pass here): the caller (gmg_score_orfs) never reads them.
// Returns GMG_EBADMODEL when the model shape has no fast path (the caller then takes its exact path).
static int gene6_impl(const gmg_model *gene, const gmg_reads *reads, float *d_gene, uint64_t gstride, bool full, hipStream_t s)
{
    if (!(gene->dev.has_fast && gene->dev.D == 7 && gene->dev.W >= 3 && gene->dev.W <= 15 && gene->dev.P >= 3))
        return GMG_EBADMODEL;
    Frame6Args a;
    a.gene = gene->dev;
    a.nul = gene->dev;                                 // not used in gene-only mode
    a.packed = reads->d_packed;
    a.off = reads->d_off;
    a.tile_read = reads->d_tile_read;
    a.total = reads->total_bases;
    a.n_reads = reads->n_reads;
    a.first = 0;
    a.count = 0;
    a.out = nullptr;
    a.stride = 0;
    a.out_gene = d_gene;
    a.gstride = gstride;
    a.str_sums = nullptr;
    a.uniform_len = 0;
    constexpr int BLOCK = 1024, DT = 7, KR = 16;
    constexpr uint32_t SPAN = 2 * BLOCK;
    const uint64_t n_chunks = a.total / SPAN;
    if (n_chunks > 0) {
        int dev = 0, n_cu = 256;
        GMG_HIP(hipGetDevice(&dev));
        GMG_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        unsigned nworkers = (unsigned)(n_cu / 3);
        if (nworkers < 1) nworkers = 1;
        if (nworkers > n_chunks) nworkers = (unsigned)n_chunks;
        const unsigned grid = 3 * nworkers;
        const size_t lds = ((size_t)1 << (2 * DT)) * 8;
        if ((gstride & 1) == 0) {
            GMG_HIP(hipFuncSetAttribute((const void *)k_frame6t<BLOCK, DT, KR, 0, true, true>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((k_frame6t<BLOCK, DT, KR, 0, true, true>), dim3(grid), dim3(BLOCK), lds, s, a);
        } else {
            GMG_HIP(hipFuncSetAttribute((const void *)k_frame6t<BLOCK, DT, KR, 0, false, true>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((k_frame6t<BLOCK, DT, KR, 0, false, true>), dim3(grid), dim3(BLOCK), lds, s, a);
        }
        GMG_HIP(hipGetLastError());
    }
    if (!full) return launch_generic(a, n_chunks * SPAN, a.total - n_chunks * SPAN, s);
    // the partial-window heads of every read and the batch tail, as in gmg_launch_frame6_strided
    if (a.n_reads > 0) {
        const uint64_t blocks = (a.n_reads + 7) / 8;
        const unsigned grid = (unsigned)(blocks < 256 * 8 ? blocks : 256 * 8);
        const size_t lds_p = (size_t)3 * a.gene.cstride;
        a.first = n_chunks * SPAN;
        a.count = a.total - n_chunks * SPAN;
        a.p_blocks = grid;
        const unsigned tail_blocks = (unsigned)((a.count + 255) / 256);
        hipLaunchKernelGGL((k_frame6p<7, 4, true>), dim3(grid + tail_blocks), dim3(256), lds_p, s, a);
        GMG_HIP(hipGetLastError());
    }
    return GMG_OK;
}

int gmg_launch_gene6(const gmg_model *gene, const gmg_reads *reads, float *d_gene, hipStream_t s)
{
    return gene6_impl(gene, reads, d_gene, reads->total_bases, false, s);
}

// The same rows complete: the first W-1 positions of either scoring buffer of every read by the partial-window rule
// (icm.cc:807-842), rows gstride floats apart.  Frame_Scores = (double) these - (double) the null model's values.
int gmg_launch_gene6_full(const gmg_model *gene, const gmg_reads *reads, float *d_gene, uint64_t gstride, hipStream_t s)
{
    return gene6_impl(gene, reads, d_gene, gstride, true, s);
}

__global__ void k_f6_gather_u64(const uint64_t *src, const uint64_t *idx, uint32_t n, uint64_t *dst)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

// The gene-only table of a batch whose reads come in consecutive groups, every group under its own gene model -- what
// glimmer-mg's classification mode scores ICM by ICM (glimmer-mg.cc:361-451) in ONE pass over the batch: the main pass takes a
// list of rounds (runs of <= 16 whole chunks inside one group) and swaps the model in LDS when a work-group crosses into another
// group, the partial-window pass takes its reads group by group, and the bases no round covers (the chunks that hold a group
// boundary, the batch tail) go to the exact any-shape code, all in two launches.  Models without the fast path (or of different
// window widths) make the whole batch take the any-shape kernel, group by group: same values.
int gmg_launch_gene6_groups(const gmg_model *const *models, const uint64_t *group_read, int n_groups, const gmg_reads *reads,
                            float *d_gene, uint64_t gstride, hipStream_t s)
{
    if (!models || !group_read || n_groups < 1 || n_groups > F6_MAX_GROUPS || group_read[0] != 0 || group_read[n_groups] != reads->n_reads)
        return gmg_set_error(GMG_EINVAL, "gmg_launch_gene6_groups: bad group list");
    for (int g = 0; g < n_groups; g++)
        if (!models[g] || group_read[g + 1] < group_read[g] || models[g]->dev.P < 3)
            return gmg_set_error(GMG_EINVAL, "gmg_launch_gene6_groups: group %d has no periodicity-3 model or an inverted read range", g);
    if (reads->total_bases == 0) return GMG_OK;
    constexpr int BLOCK = 1024, DT = 7, KR = 16;
    constexpr uint32_t SPAN = 2 * BLOCK;

    // the groups' first bases
    std::vector<uint64_t> base(n_groups + 1);
    {
        uint64_t *d_idx = nullptr, *d_val = nullptr;
        GMG_HIP(gmg_pool_alloc((void **)&d_idx, (size_t)(n_groups + 1) * 8));
        hipError_t e = gmg_pool_alloc((void **)&d_val, (size_t)(n_groups + 1) * 8);
        if (e != hipSuccess) {                          // (a block handed out and not given back stays busy for good)
            gmg_pool_release(d_idx);
            return gmg_set_error(e == hipErrorOutOfMemory ? GMG_ENOMEM : GMG_EHIP, "gmg_launch_gene6_groups: %s", hipGetErrorString(e));
        }
        e = hipMemcpyAsync(d_idx, group_read, (size_t)(n_groups + 1) * 8, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_f6_gather_u64, dim3((n_groups + 256) / 256), dim3(256), 0, s, reads->d_off, d_idx, (uint32_t)(n_groups + 1), d_val);
            e = hipMemcpyAsync(base.data(), d_val, (size_t)(n_groups + 1) * 8, hipMemcpyDeviceToHost, s);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        gmg_pool_release(d_idx);
        gmg_pool_release(d_val);
        if (e != hipSuccess) return gmg_set_error(GMG_EHIP, "gmg_launch_gene6_groups: %s", hipGetErrorString(e));
    }

    Frame6Args a;
    a.gene = models[0]->dev;
    a.nul = models[0]->dev;                            // not used in gene-only mode
    a.packed = reads->d_packed;
    a.off = reads->d_off;
    a.tile_read = reads->d_tile_read;
    a.total = reads->total_bases;
    a.n_reads = reads->n_reads;
    a.first = 0;
    a.count = 0;
    a.p_blocks = 0;
    a.out = nullptr;
    a.stride = 0;
    a.out_gene = d_gene;
    a.gstride = gstride;
    a.str_sums = nullptr;
    a.uniform_len = 0;

    bool fast = true;
    for (int g = 0; g < n_groups && fast; g++) {
        const GmgDevModel &m = models[g]->dev;
        fast = m.has_fast && m.D == 7 && m.W >= 3 && m.W <= 15 && m.W == models[0]->dev.W && m.cstride == models[0]->dev.cstride;
    }
    if (!fast) {                                       // any shapes: the exact kernel, group by group
        for (int g = 0; g < n_groups; g++) {
            a.gene = models[g]->dev;
            const int rc = launch_generic(a, base[g], base[g + 1] - base[g], s);
            if (rc) return rc;
        }
        return GMG_OK;
    }

    // rounds: whole chunks of the global 2,048-base grid that lie inside one group; ranges: everything else
    std::vector<F6Round> rounds;
    std::vector<F6Range> ranges;
    std::vector<GmgDevModel> gm(n_groups);
    auto add_range = [&](uint64_t first, uint64_t end, int g) {
        for (; first < end; first += 4 * SPAN) {        // pieces of at most four chunks: eight blocks each
            const uint64_t n = end - first < 4 * SPAN ? end - first : 4 * SPAN;
            ranges.push_back(F6Range{first, (uint32_t)n, (uint32_t)g});
        }
    };
    for (int g = 0; g < n_groups; g++) {
        gm[g] = models[g]->dev;
        const uint64_t b0 = base[g], b1 = base[g + 1];
        if (b1 == b0) continue;
        const uint64_t c0 = (b0 + SPAN - 1) / SPAN, c1 = b1 / SPAN;     // whole chunks [c0, c1)
        if (c1 <= c0) { add_range(b0, b1, g); continue; }
        if (c1 > 0xffffffffull) return gmg_set_error(GMG_ETOOBIG, "gmg_launch_gene6_groups: batch too large");
        add_range(b0, c0 * SPAN, g);
        for (uint64_t c = c0; c < c1; c += KR)
            rounds.push_back(F6Round{(uint32_t)c, (uint32_t)(c1 - c < KR ? c1 - c : KR) | (uint32_t)g << 5});
        add_range(c1 * SPAN, b1, g);
    }
    F6Round *d_rounds = nullptr;
    F6Range *d_ranges = nullptr;
    GmgDevModel *d_gm = nullptr;
    uint64_t *d_gread = nullptr;
    hipError_t e = gmg_pool_alloc((void **)&d_rounds, (rounds.size() + 1) * sizeof(F6Round));
    if (e == hipSuccess) e = gmg_pool_alloc((void **)&d_ranges, (ranges.size() + 1) * sizeof(F6Range));
    if (e == hipSuccess) e = gmg_pool_alloc((void **)&d_gm, gm.size() * sizeof(GmgDevModel));
    if (e == hipSuccess) e = gmg_pool_alloc((void **)&d_gread, (size_t)(n_groups + 1) * 8);
    if (e == hipSuccess && !rounds.empty()) e = hipMemcpyAsync(d_rounds, rounds.data(), rounds.size() * sizeof(F6Round), hipMemcpyHostToDevice, s);
    if (e == hipSuccess && !ranges.empty()) e = hipMemcpyAsync(d_ranges, ranges.data(), ranges.size() * sizeof(F6Range), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_gm, gm.data(), gm.size() * sizeof(GmgDevModel), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_gread, group_read, (size_t)(n_groups + 1) * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);               // (the host vectors go away with this call)
    auto done = [&](int rc) {
        gmg_pool_release_after(d_rounds, s);
        gmg_pool_release_after(d_ranges, s);
        gmg_pool_release_after(d_gm, s);
        gmg_pool_release_after(d_gread, s);
        return rc;
    };
    if (e != hipSuccess) return done(gmg_set_error(GMG_EHIP, "gmg_launch_gene6_groups: %s", hipGetErrorString(e)));
    a.rounds = d_rounds;
    a.n_rounds = (uint32_t)rounds.size();
    a.ranges = d_ranges;
    a.n_ranges = (uint32_t)ranges.size();
    a.gmodels = d_gm;
    a.group_read = d_gread;
    a.n_groups = (uint32_t)n_groups;

    if (!rounds.empty()) {
        int dev = 0, n_cu = 256;
        if ((e = hipGetDevice(&dev)) != hipSuccess || (e = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess)
            return done(gmg_set_error(GMG_EHIP, "gmg_launch_gene6_groups: %s", hipGetErrorString(e)));
        unsigned nworkers = (unsigned)(n_cu / 3);
        if (nworkers < 1) nworkers = 1;
        if (nworkers > rounds.size()) nworkers = (unsigned)rounds.size();
        const unsigned grid = 3 * nworkers;
        const size_t lds = ((size_t)1 << (2 * DT)) * 8;
        if ((gstride & 1) == 0) {
            e = hipFuncSetAttribute((const void *)k_frame6t<BLOCK, DT, KR, 0, true, true, false, false, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess) hipLaunchKernelGGL((k_frame6t<BLOCK, DT, KR, 0, true, true, false, false, true>), dim3(grid), dim3(BLOCK), lds, s, a);
        } else {
            e = hipFuncSetAttribute((const void *)k_frame6t<BLOCK, DT, KR, 0, false, true, false, false, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess) hipLaunchKernelGGL((k_frame6t<BLOCK, DT, KR, 0, false, true, false, false, true>), dim3(grid), dim3(BLOCK), lds, s, a);
        }
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) return done(gmg_set_error(GMG_EHIP, "gmg_launch_gene6_groups: %s", hipGetErrorString(e)));
    }
    // the partial-window heads of every read, group by group, and the ranges outside the rounds
    {
        const uint64_t blocks = (a.n_reads + 7) / 8;
        const unsigned grid = (unsigned)(blocks < 256 * 8 ? blocks : 256 * 8);
        const size_t lds_p = (size_t)3 * a.gene.cstride;
        a.p_blocks = grid;
        // (eight blocks per range: with one ICM per read -- up to 2^27 groups -- that would pass the grid limit only at launch time)
        if ((uint64_t)grid + 8ull * ranges.size() > 0x7fffffffull)
            return done(gmg_set_error(GMG_ETOOBIG, "gmg_launch_gene6_groups: %llu group edges in one batch: split the batch", (unsigned long long)ranges.size()));
        hipLaunchKernelGGL((k_frame6p<7, 4, true, true>), dim3(grid + 8 * (unsigned)ranges.size()), dim3(256), lds_p, s, a);
        e = hipGetLastError();
        if (e != hipSuccess) return done(gmg_set_error(GMG_EHIP, "gmg_launch_gene6_groups: %s", hipGetErrorString(e)));
    }
    return done(GMG_OK);
}

// Per-base values of the two strings scoreReadsGlim scores with a periodicity-1 ICM (the read, and its reverse
// complement stored at forward coordinates), fp32 rows [2][total], full-window rule, full chunks only: the caller
// (k_string_sum, gmg_strings.hip) recomputes the first W-1 positions of either string and the last < 2,048 bases
// of the batch itself.  *tail_start = first base the main pass did not write.
int gmg_launch_strings(const gmg_model *m, const gmg_reads *reads, float *d_vals, uint64_t *tail_start, hipStream_t s)
{
    if (!(m->dev.has_fast && m->dev.D == 7 && m->dev.W >= 3 && m->dev.W <= 15 && m->dev.P == 1)) return GMG_EBADMODEL;
    Frame6Args a;
    a.gene = m->dev;
    a.nul = m->dev;                                    // not used in gene-only mode
    a.packed = reads->d_packed;
    a.off = reads->d_off;
    a.tile_read = reads->d_tile_read;
    a.total = reads->total_bases;
    a.n_reads = reads->n_reads;
    a.first = 0;
    a.count = 0;
    a.out = nullptr;
    a.stride = 0;
    a.out_gene = d_vals;
    a.gstride = reads->total_bases;
    a.str_sums = nullptr;
    a.uniform_len = 0;
    constexpr int BLOCK = 1024, DT = 7, KR = 16;
    constexpr uint32_t SPAN = 2 * BLOCK;
    const uint64_t n_chunks = a.total / SPAN;
    *tail_start = n_chunks * SPAN;
    if (n_chunks == 0) return GMG_OK;
    int dev = 0, n_cu = 256;
    GMG_HIP(hipGetDevice(&dev));
    GMG_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    unsigned grid = (unsigned)n_cu;                    // one persistent work-group per CU, all on sub-model 0
    if (grid > n_chunks) grid = (unsigned)n_chunks;
    const size_t lds = ((size_t)1 << (2 * DT)) * 8;
    if ((a.total & 1) == 0) {
        GMG_HIP(hipFuncSetAttribute((const void *)k_frame6t<BLOCK, DT, KR, 0, true, true, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_frame6t<BLOCK, DT, KR, 0, true, true, true>), dim3(grid), dim3(BLOCK), lds, s, a);
    } else {
        GMG_HIP(hipFuncSetAttribute((const void *)k_frame6t<BLOCK, DT, KR, 0, false, true, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_frame6t<BLOCK, DT, KR, 0, false, true, true>), dim3(grid), dim3(BLOCK), lds, s, a);
    }
    GMG_HIP(hipGetLastError());
    return GMG_OK;
}

// The same pass with the values summed per read and string instead of stored (k_frame6t<.., SUM>): d_sums[n_reads][2] must be
// zero; on return (stream-ordered) it holds, per read and string, the sum of the values at the positions whose window lies
// inside the read and whose base is in front of *tail_start.  The caller (gmg_strings.hip) adds the first W-1 positions,
// decides per read whether the order of the additions could have mattered, and recomputes the reads where it could.
int gmg_launch_strings_sum(const gmg_model *m, const gmg_reads *reads, double *d_sums, uint64_t *tail_start, hipStream_t s)
{
    if (!(m->dev.has_fast && m->dev.D == 7 && m->dev.W >= 3 && m->dev.W <= 15 && m->dev.P == 1)) return GMG_EBADMODEL;
    Frame6Args a;
    a.gene = m->dev;
    a.nul = m->dev;                                    // not used in gene-only mode
    a.packed = reads->d_packed;
    a.off = reads->d_off;
    a.tile_read = reads->d_tile_read;
    a.total = reads->total_bases;
    a.n_reads = reads->n_reads;
    a.first = 0;
    a.count = 0;
    a.out = nullptr;
    a.stride = 0;
    a.out_gene = nullptr;
    a.gstride = 0;
    a.str_sums = d_sums;
    a.uniform_len = reads->uniform_len > 0 ? (uint32_t)reads->uniform_len : 0u;
    constexpr int BLOCK = 1024, DT = 7, KR = 14;      // chunks per round: 10 .. 14 take 1.77 - 1.80 ms per model, 16 (128 registers) 1.91, 8 1.84, 24 2.01
    constexpr uint32_t SPAN = 2 * BLOCK;
    const uint64_t n_chunks = a.total / SPAN;
    *tail_start = n_chunks * SPAN;
    if (n_chunks == 0) return GMG_OK;
    int dev = 0, n_cu = 256;
    GMG_HIP(hipGetDevice(&dev));
    GMG_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    unsigned grid = (unsigned)n_cu;                    // one persistent work-group per CU, all on sub-model 0
    if (grid > n_chunks) grid = (unsigned)n_chunks;
    const size_t lds = ((size_t)1 << (2 * DT)) * 8;
    GMG_HIP(hipFuncSetAttribute((const void *)k_frame6t<BLOCK, DT, KR, 0, true, true, true, true>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_frame6t<BLOCK, DT, KR, 0, true, true, true, true>), dim3(grid), dim3(BLOCK), lds, s, a);
    GMG_HIP(hipGetLastError());
    return GMG_OK;
}
