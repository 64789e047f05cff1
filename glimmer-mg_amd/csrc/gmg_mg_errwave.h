// gmg_mg_errwave.h -- glimmer-mg's error branch (-i / -s) with ONE WAVE PER (READ, STRAND); included by gmg_mg.hip.
// Reference: Score_Orfs_Errors / Score_Orf_Starts / Score_Indels, src/Glimmer/glimmer-mg.cc:1513-1602, 1605-1861.
//
// The level kernels above keep the running sums of every reading-frame class in HBM (48 B/base) and every event of every
// call fetches a cache line of that table for 32 bytes used: ~150 GB of traffic per 1M reads, two misses in three requests.
// But a call never leaves its read, nor its strand -- everything the calls of one (read, strand) ask for is
//     3 classes x n running sums (24 B/base: 10 KB for a 400-bp read), five bit masks over the walk steps, n quality bytes,
// which fits the LDS share of ONE wave.  So a wave takes a (read, strand), builds that in LDS from the fp32 gene rows (or the
// caller's table) and walks the whole call tree of every ORF of the strand out of it:
//   * the calls wait on a stack in the wave's LDS; a lane that has no call pops one, walks it (one EVENT per trip: the masks
//     say where the next start codon / low-quality base / last codon of the region is -- no run-length tables), pushes the
//     branches it meets and pops again.  No levels, no barriers: pushes and pops are wave-wide ballots, a call is 12 bytes;
//   * in walk-step coordinates both strands are the same program: an insertion at a codon that starts at step x + j0 sends the
//     child to step x + j0 + 4, a deletion to x + j0 + 2 (whichever of the codon's three bases is the bad one), the
//     substitution branch of -s to x + m + 3; only the reported coordinates know the strand;
//   * what an ORF's calls add up to (count, best score, the j's at the extreme pos) is merged with LDS atomics, and the verdict of
//     Score_Orfs_Errors (:1647-1683) is given by the same wave when its stack is empty -- no call arrays, no aggregates, no
//     verdict kernel, no 19 GB table, no run-length tables, no reversed quality copy in HBM.
// COUNT pass: every ORF (or, accepted_only, every ORF that can reach Min_Gene_Len), verdict + number of starts.  WRITE pass
// (after the scan of the counts): the kept ORFs again, starts / Error_t entries / order keys into their slices (the order inside a
// slice is restored by the segmented sort of mg_run step 5, as for the level kernels).
// Needs sums that are exact in any order (mg_run's test) and reads of at most EW_MAX_CAP bases; a full stack or more than
// EW_MAXO ORFs on one strand raise err_flag and the batch repeats on the level kernels.  Bit-identical to them and to the
// per-ORF kernel (tests/test_gpu_mg_err.py runs every path against the oracle).
#ifndef GMG_MG_ERRWAVE_H
#define GMG_MG_ERRWAVE_H

#define EW_BLOCK 64
#define EW_MAXO 64               // ORFs of one (read, strand)
#define EW_MAX_CAP 960           // longest read a wave takes (10 bits of a stack entry hold a walk step; a child may start 1 step behind the end)
#define EW_QCAP 192              // stack entries (the deepest stack of 1M 454-like reads: see DESIGN 4.7)
#define EW_THIN 0x9249249249249249ull     // every third bit: the codons of one reading frame in a 64-step window

struct EwLayout {                // byte offsets inside the wave's LDS
    uint32_t S, msk, q, st_ss, st_w, st_key, st_e, a_best, a_exa, a_exb, a_cnt, a_m0, gi, bytes;
    uint32_t srow, nw;           // doubles per class row, 64-step words per mask row
};
__host__ __device__ inline EwLayout ew_layout(uint32_t cap, uint32_t qcap, bool write)
{
    EwLayout L;
    L.srow = cap + 4;            // [0] = 0 in front of step 0; a child may be anchored one step behind the read's end
    L.nw = cap / 64 + 2;         // (+ the zero words a 64-step window runs into)
    uint32_t o = 0;
    L.S = o; o += 3 * L.srow * 8;
    L.msk = o; o += 5 * L.nw * 8;
    L.st_ss = o; o += qcap * 8;
    L.st_key = o; o += write ? qcap * 8 : 0;
    L.a_best = o; o += write ? 0 : EW_MAXO * 8;
    L.a_exa = o; o += write ? 0 : EW_MAXO * 8;
    L.a_exb = o; o += write ? 0 : EW_MAXO * 8;
    L.st_w = o; o += qcap * 4;
    L.st_e = o; o += write ? qcap * 4 : 0;
    L.a_cnt = o; o += EW_MAXO * 4;          // WRITE: slots handed out inside the ORF's slice
    L.a_m0 = o; o += EW_MAXO * 4;           // WRITE: where the slice begins
    L.gi = o; o += EW_MAXO * 4;
    L.q = o; o += (cap + 8 + 7) & ~7u;
    L.bytes = (o + 15) & ~15u;
    return L;
}

// the 64 steps from step t on of a mask row (rows end in zero words)
__device__ __forceinline__ uint64_t ew_window(const uint64_t *row, uint32_t t)
{
    const uint32_t wi = t >> 6, sh = t & 63u;
    const uint64_t lo = row[wi], hi = row[wi + 1];
    return (lo >> sh) | ((hi << 1) << (63u - sh));
}

template <bool WRITE, bool G32>
__global__ __launch_bounds__(EW_BLOCK) void k_mg_err_wave(MgArgs a, const int accepted_only, const uint32_t cap, const uint32_t qcap)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ew_lds[];
    __shared__ double s_pen[64];
    __shared__ int8_t s_which[64];
    const uint32_t lane = threadIdx.x;
    s_pen[lane] = a.err_mode == 1 ? a.pen[lane] : 0.0;
    s_which[lane] = a.which[lane];
    const EwLayout L = ew_layout(cap, qcap, WRITE);
    double *S = (double *)(ew_lds + L.S);
    uint64_t *msk = (uint64_t *)(ew_lds + L.msk);
    uint64_t *Eq = msk, *En = msk + L.nw, *Mstart = msk + 2 * L.nw, *Mstop = msk + 3 * L.nw, *Mlow = msk + 4 * L.nw;
    uint8_t *s_q = ew_lds + L.q;
    double *st_ss = (double *)(ew_lds + L.st_ss);
    uint32_t *st_w = (uint32_t *)(ew_lds + L.st_w);
    uint64_t *st_key = (uint64_t *)(ew_lds + L.st_key);
    uint32_t *st_e = (uint32_t *)(ew_lds + L.st_e);
    unsigned long long *a_best = (unsigned long long *)(ew_lds + L.a_best), *a_exa = (unsigned long long *)(ew_lds + L.a_exa),
                       *a_exb = (unsigned long long *)(ew_lds + L.a_exb);
    uint32_t *a_cnt = (uint32_t *)(ew_lds + L.a_cnt), *a_m0 = (uint32_t *)(ew_lds + L.a_m0), *s_gi = (uint32_t *)(ew_lds + L.gi);
    __syncthreads();
    const bool pen_lds = a.indel_q_thr < 64;
    const int mgl = a.min_gene_len;
    const int lowest_j = mgl - 3 < 3 ? mgl - 3 : 3;
    const bool indels = a.err_mode == 1;
    const uint32_t nw = L.nw, srow = L.srow;

    for (uint64_t it = blockIdx.x; it < 2 * a.n_reads; it += gridDim.x) {
        const uint64_t r = it >> 1;
        const bool fwd = (it & 1) == 0;
        const uint64_t off = a.read_off[r];
        const uint32_t n = (uint32_t)(a.read_off[r + 1] - off);
        const uint64_t ob = a.read_orf_off[r], oe = a.read_orf_off[r + 1];
        if (n == 0 || n > cap || ob == oe) continue;    // (longer reads: k_mg_err_flat, read_fit = 0)
        const int isl = a.read_isl ? a.read_isl[r] : a.ignore_score_len;
        const uint32_t off_m3 = (uint32_t)(off % 3);

        // ---- the ORFs of this strand: level-0 calls onto the stack
        uint32_t top = 0, nloc = 0;
        bool overflow = false;
        for (uint64_t o0 = ob; o0 < oe; o0 += 64) {
            const uint64_t i = o0 + lane;
            const bool have = i < oe;
            int frame = 0, stop_position = 0;
            if (have) { frame = a.orfs[i].frame; stop_position = a.orfs[i].stop_position; }
            const bool mine = have && (frame > 0) == fwd;
            const uint64_t mm = __ballot(mine);
            const uint32_t idx = nloc + __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
            nloc += (uint32_t)__popcll(mm);
            if (nloc > EW_MAXO) { overflow = true; break; }
            bool act = mine;
            const int end_point = fwd ? stop_position - 1 : stop_position + 3;
            const int xs = fwd ? (int)n - end_point : end_point - 1;               // the call's first walk step
            if (mine) {
                s_gi[idx] = (uint32_t)i;
                if (WRITE) {
                    a_cnt[idx] = 0;
                    a_m0[idx] = (uint32_t)a.start_off[i];
                    if (accepted_only && !((a.acc_bits[i >> 5] >> (i & 31u)) & 1u)) act = false;
                } else {
                    a_best[idx] = mg_ord(-DBL_MAX);
                    a_exa[idx] = a_exb[idx] = fwd ? ~0ull : 0ull;
                    a_cnt[idx] = 0; a_m0[idx] = 0;
                    // accepted_only: an ORF is kept only if one of its starts has j + 1 >= Min_Gene_Len (glimmer-mg.cc:1655-1668) and no
                    // path gets further from the ORF's end than the read reaches that way (as k_mg_err_level)
                    if (accepted_only && ((int)n - xs) + 12 < mgl) act = false;
                }
                if (xs < 0 || xs >= (int)n) act = false;                             // (nothing to walk, nothing to branch from; m0 = 0)
            }
            const uint64_t am = __ballot(act);
            if (act) {
                const uint32_t e = top + __builtin_amdgcn_mbcnt_hi((uint32_t)(am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am, 0u));
                if (e < qcap) {
                    st_ss[e] = 0.0;
                    st_w[e] = (uint32_t)xs | idx << 21;                              // step | suffix_j << 10 | ORF << 21 | level << 27
                    if (WRITE) { st_key[e] = 0; st_e[e] = 0; }
                }
            }
            top += (uint32_t)__popcll(am);
        }
        if (overflow || top > qcap) { if (lane == 0) atomicOr(a.err_flag, 1u); continue; }
        if (top == 0 && (WRITE || accepted_only)) continue;          // (every ORF's record is written otherwise: the verdict below)
        if (top) {
        // ---- the running sums of the three classes, the masks, the qualities
        const float *nt = G32 ? a.null_tab + (size_t)(a.read_null ? a.read_null[r] : 0u) * MG_NULL_FLOATS : nullptr;
        if (lane < 3) S[lane * srow] = 0.0;
        for (uint32_t w = lane; w < 5 * nw; w += 64) msk[w] = 0;
        wcs_sync();
        double carry[3] = {0.0, 0.0, 0.0};
        for (uint32_t t0 = 0; t0 < n; t0 += 64) {
            const uint32_t t = t0 + lane;
            const bool in = t < n;
            const uint64_t g = in ? (fwd ? off + n - 1 - t : off + t) : off;
            const uint32_t five = (uint32_t)dev_window_bits(a.packed, (int64_t)g - 2) & 0x3ffu, c0 = (five >> 4) & 3u;   // bases g - 2 .. g + 2
            double v[3];
            if (G32) {
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
            const int m = (int)((off_m3 + (uint32_t)(g - off)) % 3u);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int row = ((fwd ? c - m + 3 : m - c + 3) % 3 + 1) % 3;          // as k_mg_walk_prefix
                const double x = row == 0 ? v[0] : row == 1 ? v[1] : v[2];
                const double sc = mg_wave_scan(x) + carry[c];
                if (in) S[c * srow + t + 1] = sc;
                carry[c] = wcs_last_lane(sc);
            }
            // the codon that starts at this step, as the walks form it: code (step) | code (step + 1) << 2 | code (step + 2) << 4
            const uint32_t idx = fwd ? c0 | ((five >> 2) & 3u) << 2 | (five & 3u) << 4
                                     : (c0 | ((five >> 6) & 3u) << 2 | ((five >> 8) & 3u) << 4) ^ 63u;
            const bool codon = t + 2 < n;
            int q = 255;
            if (indels && in) q = a.qual[g];
            const uint64_t b_start = __ballot(codon && s_which[idx] >= 0), b_stop = __ballot(codon && ((a.fwd_stop >> idx) & 1ull)),
                           b_low = __ballot(in && q <= a.indel_q_thr);
            if (lane == 0) { Mstart[t0 >> 6] = b_start; Mstop[t0 >> 6] = b_stop; Mlow[t0 >> 6] = b_low; }
            if (indels && in) s_q[t] = (uint8_t)q;
        }
        wcs_sync();
        // events of a walk: a start codon, the last codon of a region (the next one is a stop codon or does not fit the read), and (Eq) a
        // codon with a base of low quality
        for (uint32_t w = lane; w < nw - 1; w += 64) {
            const uint64_t zs = Mstop[w], zs1 = Mstop[w + 1], lo = Mlow[w], lo1 = Mlow[w + 1];
            // steps t with t + 5 > n - 1, t < n
            uint64_t endm = 0;
            const int64_t b0 = (int64_t)64 * w;
            const int64_t e_lo = (int64_t)n - 5 > b0 ? (int64_t)n - 5 - b0 : 0, e_hi = (int64_t)n - b0;     // bits [e_lo, e_hi)
            if (e_hi > 0 && e_lo < 64) {
                const uint64_t upto = e_hi >= 64 ? ~0ull : ((1ull << e_hi) - 1ull);
                endm = upto & ~((1ull << e_lo) - 1ull);
            }
            const uint64_t en = Mstart[w] | (zs >> 3 | zs1 << 61) | endm;
            En[w] = en;
            Eq[w] = en | lo | (lo >> 1 | lo1 << 63) | (lo >> 2 | lo1 << 62);
        }
        wcs_sync();

        // ---- the call tree
        uint32_t state = 0;                             // 0: no call, 1: walking, 2: the call has ended
        uint32_t x = 0, sj = 0, lidx = 0, level = 0, j0 = 0, cnt = 0, m_end = 0, ee = 0, last_own = MG_NO_SLOT;
        double ss = 0.0, p0 = 0.0, best = -DBL_MAX, last_sum = 0.0;
        const double *Sc = S;
        uint64_t key = 0;
        int last_pos = 0, last_j = 0;
        bool trunc = false, first_done = false;
        for (;;) {
            const uint64_t im = __ballot(state == 0);
            if (top && im) {
                const uint32_t idle_n = (uint32_t)__popcll(im), take = idle_n < top ? idle_n : top;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(im >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)im, 0u));
                if (state == 0 && rank < take) {
                    const uint32_t e = top - 1 - rank;
                    ss = st_ss[e];
                    const uint32_t w = st_w[e];
                    x = w & 1023u; sj = (w >> 10) & 2047u; lidx = (w >> 21) & 63u; level = w >> 27;
                    if (WRITE) { key = st_key[e]; ee = st_e[e]; }
                    cnt = 0; best = -DBL_MAX; last_own = MG_NO_SLOT; first_done = false; trunc = false; j0 = 0; m_end = 0; last_sum = 0.0;
                    state = 2;
                    if (x < n) {
                        const uint32_t cls = (off_m3 + (fwd ? n - 1 - x : x)) % 3u;      // the class of the call's first base
                        Sc = S + cls * srow;
                        p0 = Sc[x];                     // the running sum in front of the call's first position
                        if (n - x < 3) trunc = a.allow_truncated != 0;
                        else if (!((Mstop[x >> 6] >> (x & 63u)) & 1ull)) state = 1;
                    }
                }
                top -= take;
                wcs_sync();
            }
            if (!__ballot(state != 0)) {
                if (!top) break;
                continue;
            }
            uint32_t pm = 0;                            // branches of this trip: bit 2 pj + b (b = 0 insertion, 1 deletion), bit 6 the substitution branch
            double es6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, es_sub = 0.0;
            uint32_t jt = 0;                            // the codon this trip works on (j0 moves on before the branches are pushed)
            if (state == 1) {
                uint32_t t = x + j0;
                const bool branching = indels && level < 2 && (int)level < a.indel_max;
                const uint64_t win = ew_window(branching ? Eq : En, t) & EW_THIN;
                const bool hop = win == 0;              // 22 codons without an event: on to the next window
                t += hop ? 66u : (uint32_t)__builtin_ctzll(win);
                j0 = t - x;
                jt = j0;
                if (!hop) {
                    const bool by_end = t + 5 > n - 1;
                    const bool is_last = by_end || ((Mstop[(t + 3) >> 6] >> ((t + 3) & 63u)) & 1ull);
                    if (by_end) trunc = a.allow_truncated != 0;
                    const bool st = (Mstart[t >> 6] >> (t & 63u)) & 1ull;
                    const double prev = Sc[t] - p0, s0 = Sc[t + 1] - p0, s1 = Sc[t + 2] - p0, sum = Sc[t + 3] - p0;   // score[j0 - 1 .. j0 + 2]
                    if ((int)j0 >= lowest_j && (int)(j0 + 3 + sj) >= mgl && (st || (is_last && trunc))) {
                        const int k = fwd ? (int)n - (int)x - 2 - (int)j0 : (int)x + (int)j0 + 3;
                        const double raw = (prev - 0.0) + ss;
                        const int j_full = (int)j0 + 2 + (int)sj;
                        const double sc = (j_full > isl && 0.0 > raw) ? 0.0 : raw;
                        int which = -1;
                        if (WRITE && st) {
                            const uint64_t g = fwd ? off + n - 1 - t : off + t;
                            const uint32_t five = (uint32_t)dev_window_bits(a.packed, (int64_t)g - 2) & 0x3ffu, c0 = (five >> 4) & 3u;
                            which = s_which[fwd ? c0 | ((five >> 2) & 3u) << 2 | (five & 3u) << 4 : (c0 | ((five >> 6) & 3u) << 2 | ((five >> 8) & 3u) << 4) ^ 63u];
                        }
                        // the real start of the codon first, then the truncated one (reversed push order: k_mg_err_level)
#pragma unroll
                        for (int e2 = 0; e2 < 2; e2++) {
                            if (e2 == 0 ? !st : !(is_last && trunc)) continue;
                            if (WRITE) {
                                const uint32_t slot = a_m0[lidx] + atomicAdd(&a_cnt[lidx], 1u);
                                gmg_start s1_;
                                s1_.score = sc; s1_.j = j_full; s1_.pos = k; s1_.which = e2 == 0 ? which : -1;
                                s1_.truncated = (int16_t)(e2 == 0 ? 0 : 1); s1_.first = (int16_t)(e2 == 0 ? 0 : 1);
                                a.starts[slot] = s1_;
                                gmg_start_errors er;
                                er.pos[0] = level > 0 ? (int)((ee & 0x3fffu) >> 2) - 8 : 0; er.pos[1] = level > 1 ? (int)((ee >> 14) >> 2) - 8 : 0;
                                er.type[0] = (int8_t)(level > 0 ? (ee & 3u) : 0); er.type[1] = (int8_t)(level > 1 ? ((ee >> 14) & 3u) : 0);
                                er.n = (int8_t)level; er.reserved = 0;
                                a.errs[slot] = er;
                                a.keys[slot] = key | (uint64_t)((uint32_t)(2047 - (int)j0) << 2 | (e2 == 0 ? 3u : 2u)) << (26 - 13 * (int)level);
                                if (e2 == 0) last_own = slot; else first_done = true;
                            } else {
                                last_pos = k; last_j = j_full;
                                if (sc > best) best = sc;
                                cnt++;
                            }
                        }
                    }
                    if (branching) {
                        const uint32_t lows = (uint32_t)ew_window(Mlow, t) & 7u;
                        if (lows) {
                            // Score_Indels at the codon's three positions (glimmer-mg.cc:1513-1602): insertion = the sum BEFORE the base,
                            // deletion = the sum AT it, each + the penalty of the base's quality
                            const uint32_t qw = (uint32_t)s_q[t] | (uint32_t)s_q[t + 1] << 8 | (uint32_t)s_q[t + 2] << 16;
                            const int c_sj = (int)sj + (int)j0 + 2;
                            // a call that cannot reach Min_Gene_Len before its read ends emits nothing, nor can a branch of it
                            const bool ins_ok = c_sj + ((int)n - (int)(x + j0 + 4)) + 12 >= mgl, del_ok = c_sj + ((int)n - (int)(x + j0 + 2)) + 12 >= mgl;
#pragma unroll
                            for (int pj = 0; pj < 3; pj++) {
                                const int q = (int)((qw >> (8 * pj)) & 255u);
                                const bool low = ((lows >> pj) & 1u) && (int)j0 + pj >= lowest_j;
                                const double pen = pen_lds ? s_pen[q & 63] : a.pen[q];
                                const double before = pj == 0 ? prev : pj == 1 ? s0 : s1, at = pj == 0 ? s0 : pj == 1 ? s1 : sum;
                                es6[2 * pj] = ((ss + before) - 0.0) + pen;
                                es6[2 * pj + 1] = ((ss + at) - 0.0) + pen;
                                if (low && ins_ok && es6[2 * pj] > a.indel_suffix_thr) pm |= 1u << (2 * pj);
                                if (low && del_ok && es6[2 * pj + 1] > a.indel_suffix_thr) pm |= 2u << (2 * pj);
                            }
                        }
                    }
                    last_sum = sum;
                    if (is_last) { m_end = j0 + 3; state = 2; }
                    else j0 += 3;
                }
            }
            if (state == 2) {                           // the end of the call (also of one that had nothing to walk)
                if (level == 0 && !WRITE) a_m0[lidx] = m_end << 1 | (trunc ? 1u : 0u);
                if (level == 0 && a.err_mode == 2 && x < n && x + m_end + 3 <= n) {
                    // the substitution branch (:1771-1806): through the stop codon behind the region (steps x + m .. x + m + 2)
                    const uint32_t sa = x + m_end;
                    const uint64_t g = fwd ? off + n - 1 - sa : off + sa;
                    const uint32_t five = (uint32_t)dev_window_bits(a.packed, (int64_t)g - 2) & 0x3ffu;
                    // code (step sa) == a and code (step sa + 1) == a in walk codes (reverse strand: the complement of t)
                    const int a2 = fwd ? ((five >> 4) & 3u) == 0u : ((five >> 4) & 3u) == 3u;
                    const int a1 = fwd ? ((five >> 2) & 3u) == 0u : ((five >> 6) & 3u) == 3u;
                    es_sub = ss + a.pass_stop[a1 * 2 + a2];
                    if (m_end > 0) es_sub += last_sum - 0.0;
                    if ((int)(sj + m_end) + ((int)n - (int)(x + m_end + 3)) + 12 >= mgl) pm |= 64u;
                }
                if (WRITE) { if (!first_done && last_own != MG_NO_SLOT) a.starts[last_own].first = 1; }
                else if (cnt) {
                    atomicAdd(&a_cnt[lidx], cnt);
                    atomicMax(&a_best[lidx], (unsigned long long)mg_ord(best));
                    const unsigned long long pa = (unsigned long long)(uint32_t)(last_pos + 16) << 32 | (uint32_t)last_j,
                                             pb = (unsigned long long)(uint32_t)(last_pos + 16) << 32 | (0xffffffffu - (uint32_t)last_j);
                    if (fwd) { atomicMin(&a_exa[lidx], pa); atomicMin(&a_exb[lidx], pb); }
                    else { atomicMax(&a_exa[lidx], pa); atomicMax(&a_exb[lidx], pb); }
                }
                state = 0;
            }
            // the branches of this trip onto the stack
            uint64_t wm;
            while ((wm = __ballot(pm != 0)) != 0) {
                if (pm) {
                    const int c = __ffs((int)pm) - 1;
                    pm &= pm - 1u;
                    const uint32_t e = top + __builtin_amdgcn_mbcnt_hi((uint32_t)(wm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wm, 0u));
                    if (e < qcap) {
                        if (c == 6) {
                            st_ss[e] = es_sub;
                            st_w[e] = (x + m_end + 3) | (sj + m_end) << 10 | lidx << 21 | 1u << 27;
                            if (WRITE) {
                                // Error_t (lo - 2 / hi + 2, substitution): forward lo = end_point - m = n - x - m, reverse hi = end_point + m = x + 1 + m
                                const int epos = fwd ? (int)n - (int)x - (int)m_end - 2 : (int)x + (int)m_end + 3;
                                st_key[e] = key;        // (field 0: before every position of the call)
                                st_e[e] = (uint32_t)(epos + 8) << 2 | 2u;
                            }
                        } else {
                            const int pj = c >> 1, b = c & 1, j = (int)jt + pj;
                            st_ss[e] = c == 0 ? es6[0] : c == 1 ? es6[1] : c == 2 ? es6[2] : c == 3 ? es6[3] : c == 4 ? es6[4] : es6[5];
                            st_w[e] = (x + jt + (b == 0 ? 4u : 2u)) | (sj + jt + 2u) << 10 | lidx << 21 | (level + 1u) << 27;
                            if (WRITE) {
                                const int k = fwd ? (int)n - (int)x - 2 - j : (int)x + 3 + j;
                                const int epos = b == 0 ? (fwd ? k + 2 : k - 2) : (fwd ? k + 3 : k - 1);
                                const uint32_t c_err = (uint32_t)(epos + 8) << 2 | (uint32_t)b;
                                st_key[e] = key | (uint64_t)((uint32_t)(2047 - j) << 2 | (b == 0 ? 1u : 0u)) << (26 - 13 * (int)level);
                                st_e[e] = level == 0 ? c_err : (ee & 0x3fffu) | c_err << 14;
                            }
                        }
                    }
                }
                top += (uint32_t)__popcll(wm);
            }
            if (top > qcap) { overflow = true; break; }
            wcs_sync();
        }
        }
        if (overflow) { if (lane == 0) atomicOr(a.err_flag, 1u); continue; }
        if (WRITE) continue;

        // ---- Score_Orfs_Errors' verdict per ORF (:1647-1683; as k_mg_err_verdict)
        wcs_sync();
        if (lane < nloc) {
            const uint32_t g_cnt = a_cnt[lane];
            const uint64_t i = s_gi[lane];
            if (!(accepted_only && g_cnt == 0)) {
                gmg_mg_orf rec = a.orfs[i];
                const uint32_t g_m0 = a_m0[lane];
                const int m0 = (int)(g_m0 >> 1);
                if (fwd) { rec.hi = rec.stop_position - 1; rec.lo = rec.hi - m0; }
                else { rec.lo = rec.stop_position + 3; rec.hi = rec.lo + m0; }
                rec.orf_is_truncated = (int16_t)(g_m0 & 1u);
                rec.n_starts = g_cnt;
                rec.first_j = 0; rec.best_score = -DBL_MAX; rec.accepted = 0;
                if (g_cnt) {
                    const uint32_t ja = (uint32_t)a_exa[lane], jb = 0xffffffffu - (uint32_t)a_exb[lane];
                    const int jmin = (int)(fwd ? ja : jb), jmax = (int)(fwd ? jb : ja);
                    rec.first_j = jmin;
                    if (jmax + 1 >= a.min_gene_len) {
                        rec.best_score = mg_unord(a_best[lane]);
                        if (rec.best_score > a.start_threshold) rec.accepted = jmin + 1 >= a.min_gene_len ? 1 : 2;
                    }
                }
                a.orf_cnt[i] = (accepted_only && !rec.accepted) ? 0u : g_cnt;
                if (rec.accepted) atomicOr(&a.acc_bits[i >> 5], 1u << (i & 31u));
                if (!(accepted_only && !rec.accepted)) { rec.start_begin = 0; a.orfs[i] = rec; }
            }
        }
        wcs_sync();
    }
}

#endif
