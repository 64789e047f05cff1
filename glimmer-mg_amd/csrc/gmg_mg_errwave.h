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
#define EW_QCAP 384              // stack entries (the deepest stack of 1M 454-like reads: see DESIGN 4.7)
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
    L.a_best = o; o += EW_MAXO * 8;          // (WRITE too: first_j and best_score of the kept ORFs)
    L.a_exa = o; o += EW_MAXO * 8;
    L.a_exb = o; o += EW_MAXO * 8;
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

// 64-bit value of lane `src` (uniform)
__device__ __forceinline__ uint64_t ew_readlane64(uint64_t v, uint32_t src)
{
    return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)src) |
           (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)src) << 32;
}
// the 2-bit fields of y in reversed order (nfields of them, held in the low bits)
__device__ __forceinline__ uint64_t ew_reverse_fields64(uint64_t y, uint32_t nfields)
{
    const uint64_t z = __brevll(y) >> (64u - 2u * nfields);
    return ((z & 0x5555555555555555ull) << 1) | ((z >> 1) & 0x5555555555555555ull);
}

// the value of the lane in front (lane 0: 0.0) -- DPP wave_shr:1
__device__ __forceinline__ double ew_wave_shr1(double x)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, 0x138, 0xf, 0xf, false),
                   hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), 0x138, 0xf, 0xf, false);
    return __longlong_as_double((long long)((unsigned long long)hi << 32 | lo));
}

// What a lane holds of its K consecutive walk steps between the loads and the sums
template <bool G32, int KMAX>
struct EwRegs {
    float gv[KMAX][3];           // G32: the gene model's values of the strand's three rows
    double fv[KMAX][3];          // else: the caller's Frame_Scores
    uint32_t qv[KMAX];           // qualities (255 without -i)
    uint64_t win;                // 32 bases from the lowest base the lane looks at
    uint64_t winq;               // Set_Quality_454 in the kernel: 32 bases from four bases lower (a homopolymer run is followed 5 bases back)
    int64_t g0;                  // base of the lane's first step
    uint32_t K, tb;              // steps per lane, the lane's first step
    float ntv[4];                // per-read null models: the lane's four floats of the read's table (copied to LDS before the sums)
};

// Everything a (read, strand) pair needs from HBM, asked for in one go: lane L takes the K = ceil (n / 64) consecutive steps from L K on
// (base of step t: forward off + n - 1 - t, reverse off + t)
template <bool G32, int KMAX>
__device__ __forceinline__ void ew_load(const MgArgs &a, const uint64_t off, const uint32_t n, const bool fwd, const bool indels, const uint32_t lane,
                                        EwRegs<G32, KMAX> &R)
{
    const uint32_t K = (n + 63u) >> 6, tb = lane * K;
    R.K = K; R.tb = tb;
    // the walk codes of steps tb - 2 .. tb + K + 1 come from one 32-base window
    const int64_t g_lo = fwd ? (int64_t)(off + n - 1) - (int64_t)(tb + K + 1) : (int64_t)(off + tb) - 2;
    if (indels && a.q454) {                             // (one load serves both: the lane's K + 4 fields begin four fields up)
        R.winq = dev_window_bits(a.packed, g_lo - 4);
        R.win = R.winq >> 8;
    } else R.win = dev_window_bits(a.packed, g_lo);
    const int64_t g0 = fwd ? (int64_t)(off + n - 1) - (int64_t)tb : (int64_t)(off + tb);
    R.g0 = g0;
    if (G32) {                                          // (a.ew_slack: the fp32 gene rows always come with their spare entries)
        // the call's own gene rows and quality bytes have 64 spare entries on both sides: no lane needs a predicate, every load is
        // the lane's pointer + a constant (what lies outside the read is masked where it is used)
        const float *p0 = a.gene32 + (uint64_t)(fwd ? 0 : 3) * a.fs_stride + g0, *p1 = p0 + a.fs_stride, *p2 = p1 + a.fs_stride;
        const uint8_t *pq_ = a.qual + g0;
        const bool ldq = indels && !a.q454;
        if (fwd) {
#pragma unroll
            for (int e = 0; e < KMAX; e++) {
                R.qv[e] = 255u;
                R.gv[e][0] = R.gv[e][1] = R.gv[e][2] = 0.0f;
                if ((uint32_t)e < K) {
                    R.gv[e][0] = p0[-e]; R.gv[e][1] = p1[-e]; R.gv[e][2] = p2[-e];
                    if (ldq) R.qv[e] = pq_[-e];
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < KMAX; e++) {
                R.qv[e] = 255u;
                R.gv[e][0] = R.gv[e][1] = R.gv[e][2] = 0.0f;
                if ((uint32_t)e < K) {
                    R.gv[e][0] = p0[e]; R.gv[e][1] = p1[e]; R.gv[e][2] = p2[e];
                    if (ldq) R.qv[e] = pq_[e];
                }
            }
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < KMAX; e++) {
        const bool in = (uint32_t)e < K && tb + (uint32_t)e < n;
        const int64_t g = in ? (fwd ? g0 - e : g0 + e) : (int64_t)off;
        R.qv[e] = 255u;
        if (G32) {
#pragma unroll
            for (int f = 0; f < 3; f++) R.gv[e][f] = in ? a.gene32[(uint64_t)((fwd ? 0 : 3) + f) * a.fs_stride + (uint64_t)g] : 0.0f;
        } else {
#pragma unroll
            for (int f = 0; f < 3; f++) R.fv[e][f] = in ? a.fs[(uint64_t)((fwd ? 0 : 3) + f) * a.fs_stride + (uint64_t)g] : 0.0;
        }
        if (indels && !a.q454 && in) R.qv[e] = a.qual[g];
    }
}

// The running sums of the three classes (S[c * srow + t + 1] = the sum through step t, [0] = 0), the masks over the walk steps (codon
// that starts here is a start / stop codon; base of low quality), the quality bytes.  A lane sums its K steps serially, ONE wave scan
// of the lanes' totals per class (exact in any order: mg_run's test).  zero / n_zero: the mask words to clear first.
// f_low_out: the lane's own low-quality bits (bit e = step tb + e).
template <bool G32, int KMAX, bool NTL = false>
__device__ __forceinline__ void ew_build(const MgArgs &a, const uint64_t r, const uint64_t off, const uint32_t n, const bool fwd, const bool indels,
                                         const uint32_t lane, const EwRegs<G32, KMAX> &R, double *S, const uint32_t srow, uint64_t *Mstart,
                                         uint64_t *Mstop, uint64_t *Mlow, const uint32_t nw, uint64_t *zero, const uint32_t n_zero, uint8_t *s_q,
                                         uint32_t &f_low_out, const float *nt_lds = nullptr, uint32_t *q_out = nullptr, const uint32_t ncw = 0)
{
    const uint32_t K = R.K, tb = R.tb;
    uint32_t Rq[KMAX];                                  // (the qualities computed here, for the caller's list of low-quality bases)
#pragma unroll
    for (int e = 0; e < KMAX; e++) Rq[e] = R.qv[e];
    // (the null model's table: the wave's copy in LDS when the batch has one null model, else the read's own in global memory)
    const float *nt = !G32 ? nullptr : NTL ? nt_lds : a.null_tab + (size_t)(a.read_null ? a.read_null[r] : 0u) * MG_NULL_FLOATS;
    if (lane < 3) S[lane * srow] = 0.0;
    for (uint32_t w = lane; w < n_zero; w += 64) zero[w] = 0;
    const uint64_t W = fwd ? ew_reverse_fields64(R.win, K + 4u) : ~R.win;    // field p = the walk code of step tb - 2 + p
    // the class of step t's base is (off + position) % 3; relabelled per lane so that the unrolled loop below indexes statically:
    // forward c' = (c - m0 + 1) % 3, reverse c' = (m0 - c + 1) % 3 (m0 = the class of the lane's first base): step e adds value
    // row (c' + e) % 3 to relabelled class c' (k_mg_walk_prefix's rows, rotated)
    const uint32_t m0 = (uint32_t)((uint64_t)R.g0 % 3ull);
    double P[KMAX][3];
    double acc3[3] = {0.0, 0.0, 0.0};
    uint32_t f_start = 0, f_stop = 0, f_low = 0;
    // Set_Quality_454 (glimmer-mg.cc:1865-1906) from the bases themselves: the last base of a homopolymer run of `run` bases gets
    // 31 - 5 run (6 from six on), every other base 31.  winq: base index k = the lane's lowest base - 6 + k; eq: bit 2 k set when
    // base k equals base k - 1; a run is followed down the READ (forward coordinates), whichever strand the wave walks.
    const bool q454 = indels && a.q454;
    uint64_t eq = 0;
    if (q454) {
        const uint64_t x = R.winq ^ (R.winq << 2);
        eq = ~(x | x >> 1) & 0x5555555555555554ull;
    }
#pragma unroll
    for (int e = 0; e < KMAX; e++) {
        if ((uint32_t)e < K) {                          // (uniform: the lanes' steps K .. KMAX - 1 do not exist)
            const uint32_t t = tb + (uint32_t)e;
            const bool in = t < n;
            // (Steps beyond the read -- in the lane that holds the read's last step and in the lanes behind it -- are not masked here:
            // what they add to the sums reaches only entries behind S[.][n], which nobody stores or reads; their mask bits go below.)
            double v[3];
            if (G32) {
                // the (3,2,3) null model's value at walk step t (mg_null_value in walk codes: the window is steps t - 2, t - 1, t)
                const uint32_t full = (uint32_t)(W >> (2 * e)) & 63u;
                uint32_t i0 = full, stride = 64u;
                if (e < 2 && t < 2u) {                      // (the read's first two steps: lane 0 alone)
                    const uint32_t b0 = full >> 4, b1 = (full >> 2) & 3u;
                    i0 = t == 1u ? 192u + 4u + (b1 | b0 << 2) : 192u + b0;
                    stride = 20u;
                }
#pragma unroll
                for (int f = 0; f < 3; f++) v[f] = (double)R.gv[e][f] - (double)nt[i0 + (uint32_t)f * stride];
            } else {
#pragma unroll
                for (int f = 0; f < 3; f++) v[f] = R.fv[e][f];
            }
#pragma unroll
            for (int c = 0; c < 3; c++) { acc3[c] += v[(c + e) % 3]; P[e][c] = acc3[c]; }
            const uint32_t idx = (uint32_t)(W >> (2 * e + 4)) & 63u;          // code (t) | code (t + 1) << 2 | code (t + 2) << 4
            f_start |= ((uint32_t)(a.fwd_start >> idx) & 1u) << e;
            f_stop |= ((uint32_t)(a.fwd_stop >> idx) & 1u) << e;
            uint32_t q = R.qv[e];
            if (q454) {
                const uint32_t kq = fwd ? 6u + K - 1u - (uint32_t)e : 6u + (uint32_t)e;       // the base's index in winq
                const uint32_t si = fwd ? n - 1u - t : t;                           // ... and its position in the read
                // equal pairs at k, k - 1, ..: the even bits from bit 2 k downwards, brought to the top of the word
                const uint64_t z = (eq << (62u - 2u * kq)) | 0xAAAAAAAAAAAAAAAAull;
                uint32_t run = 1u + (((uint32_t)__builtin_clzll(~z | 1ull) - 1u) >> 1);
                if (run > si + 1u) run = si + 1u;                                   // (not beyond the read's first base)
                const bool inside = si + 1u < n && ((eq >> (2u * kq + 2u)) & 1ull);   // the next base continues the run
                q = inside ? 31u : run < 6u ? 31u - 5u * run : 6u;
            }
            if (indels && q <= (uint32_t)a.indel_q_thr) f_low |= 1u << e;
            if (s_q && indels && in) s_q[t] = (uint8_t)q;
            if (q454) Rq[e] = q;
        }
    }
    {
        // the lane's steps inside the read, and those of them where a whole codon begins
        const uint32_t left = tb < n ? n - tb : 0u, n_in = left < K ? left : K, n_cod = left < 2u ? 0u : left - 2u < K ? left - 2u : K;
        f_start &= (1u << n_cod) - 1u;
        f_stop &= (1u << n_cod) - 1u;
        f_low &= (1u << n_in) - 1u;
    }
    f_low_out = f_low;
    if (q_out) {
#pragma unroll
        for (int e = 0; e < KMAX; e++) q_out[e] = Rq[e];
    }
    // lane totals -> true classes -> scan -> back
    double tot[3], basec[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {                   // true class c holds relabelled class cp
        const uint32_t cp = fwd ? ((uint32_t)c + 4u - m0) % 3u : (m0 + 4u - (uint32_t)c) % 3u;
        tot[c] = cp == 0u ? acc3[0] : cp == 1u ? acc3[1] : acc3[2];
    }
#pragma unroll
    for (int c = 0; c < 3; c++) basec[c] = ew_wave_shr1(mg_wave_scan(tot[c]));  // what the lanes in front add up to (not "- tot": see above)
    wcs_sync();                                     // (the zeroed masks before the ORs)
#pragma unroll
    for (int cp = 0; cp < 3; cp++) {
        const uint32_t c = fwd ? ((uint32_t)cp + m0 + 2u) % 3u : (m0 + 4u - (uint32_t)cp) % 3u;      // the true class of relabelled cp
        const double b = c == 0u ? basec[0] : c == 1u ? basec[1] : basec[2];
        double *row = S + c * srow + tb + 1;
#pragma unroll
        for (int e = 0; e < KMAX; e++)
            if ((uint32_t)e < K && tb + (uint32_t)e < n) row[e] = b + P[e][cp];
    }
    if (tb < n) {                                   // the lane's K bits of every mask, at bit tb
        const uint32_t wi = tb >> 6, sh = tb & 63u;
        const bool two = sh + K > 64u;
        if (ncw) {
            // compact form (k_mg_err_wcount): a row per phase, bit k of row p = the codon that starts at step 3 k + p (Mstart / Mstop
            // = row 0 of each kind, ncw words per row).  The lane's steps of phase p are every third from e0 on: consecutive codons.
            const uint32_t r3 = tb % 3u, q3 = tb / 3u;
#pragma unroll
            for (uint32_t p = 0; p < 3; p++) {
                const uint32_t e0 = (p + 3u - r3) % 3u, c0 = q3 + (r3 + e0) / 3u, cw = c0 >> 6, cs = c0 & 63u;
                const uint32_t bs = (f_start >> e0) & 0x1249u, bp = (f_stop >> e0) & 0x1249u;
                const uint32_t gs = (bs & 1u) | ((bs >> 2) & 2u) | ((bs >> 4) & 4u) | ((bs >> 6) & 8u) | ((bs >> 8) & 16u),
                               gp = (bp & 1u) | ((bp >> 2) & 2u) | ((bp >> 4) & 4u) | ((bp >> 6) & 8u) | ((bp >> 8) & 16u);
                if (gs) { atomicOr((unsigned long long *)&Mstart[p * ncw + cw], (unsigned long long)gs << cs); if (cs > 59u) atomicOr((unsigned long long *)&Mstart[p * ncw + cw + 1], (unsigned long long)gs >> (64u - cs)); }
                if (gp) { atomicOr((unsigned long long *)&Mstop[p * ncw + cw], (unsigned long long)gp << cs); if (cs > 59u) atomicOr((unsigned long long *)&Mstop[p * ncw + cw + 1], (unsigned long long)gp >> (64u - cs)); }
            }
        } else {
        if (f_start) { atomicOr((unsigned long long *)&Mstart[wi], (unsigned long long)f_start << sh); if (two) atomicOr((unsigned long long *)&Mstart[wi + 1], (unsigned long long)f_start >> (64u - sh)); }
        if (f_stop) { atomicOr((unsigned long long *)&Mstop[wi], (unsigned long long)f_stop << sh); if (two) atomicOr((unsigned long long *)&Mstop[wi + 1], (unsigned long long)f_stop >> (64u - sh)); }
        }
        if (f_low) { atomicOr((unsigned long long *)&Mlow[wi], (unsigned long long)f_low << sh); if (two) atomicOr((unsigned long long *)&Mlow[wi + 1], (unsigned long long)f_low >> (64u - sh)); }
    }
    wcs_sync();
}


// KMAX: walk steps per lane when the wave builds the running sums (a read of n bases: K = ceil (n / 64) consecutive steps per lane):
// 8 for reads up to 512 bases, 15 up to EW_MAX_CAP.  The grid strides over blocks of 64 (read, strand) pairs; a wave takes the pairs
// of its block whose read is longer than cap_lo and at most cap (the other length class has a launch of its own).
// item_flag [2 n_reads]: COUNT sets it for the pairs that hold an accepted ORF, WRITE (accepted_only) takes only those.
template <bool WRITE, bool G32, int KMAX>
__global__ __launch_bounds__(EW_BLOCK) void k_mg_err_wave(MgArgs a, const int accepted_only, const uint32_t cap_lo, const uint32_t cap, const uint32_t qcap,
                                                          uint8_t *item_flag, uint32_t *stats)
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
    const uint64_t n_items = 2 * a.n_reads;

    for (uint64_t blk = blockIdx.x; blk * 64 < n_items; blk += gridDim.x) {
    // what the block's 64 pairs are: one lane each
    uint64_t l_off = 0, l_ob = 0;
    uint32_t l_n = 0, l_no = 0;
    bool elig = false;
    {
        const uint64_t my = blk * 64 + lane;
        if (my < n_items) {
            const uint64_t r_ = my >> 1;
            l_off = a.read_off[r_];
            l_n = (uint32_t)(a.read_off[r_ + 1] - l_off);
            l_ob = a.read_orf_off[r_];
            const uint64_t no_ = a.read_orf_off[r_ + 1] - l_ob;
            l_no = no_ > 0xffffffffull ? 0xffffffffu : (uint32_t)no_;
            elig = l_n > cap_lo && l_n <= cap && l_no > 0;              // (longer reads: k_mg_err_flat, read_fit = 0)
            if (WRITE && accepted_only && elig) elig = item_flag[my] != 0;
        }
    }
    uint64_t todo = __ballot(elig);
    // what the NEXT pair needs from HBM (the lanes' K walk steps: bases, gene rows, qualities; the read's first 64 ORF records) is
    // asked for before this one is worked on and waits in registers
    EwRegs<G32, KMAX> Rn;
    int on_frame = 0, on_stop = 0;
    uint32_t on_acc = 1, on_sbeg = 0, src_n = 0;
    bool have_n = todo != 0;
    auto fetch_next = [&]() __attribute__((always_inline)) {
        src_n = (uint32_t)__builtin_ctzll(todo);
        todo &= todo - 1ull;
        const uint64_t it_ = blk * 64 + src_n;
        const uint64_t off_ = ew_readlane64(l_off, src_n), ob_ = ew_readlane64(l_ob, src_n);
        const uint32_t n_ = (uint32_t)__builtin_amdgcn_readlane((int)l_n, (int)src_n);
        const uint64_t oe_ = ob_ + (uint32_t)__builtin_amdgcn_readlane((int)l_no, (int)src_n);
        ew_load<G32, KMAX>(a, off_, n_, (it_ & 1) == 0, indels, lane, Rn);
        on_frame = 0; on_stop = 0; on_acc = 1; on_sbeg = 0;
        if (ob_ + lane < oe_) {
            on_frame = a.orfs[ob_ + lane].frame; on_stop = a.orfs[ob_ + lane].stop_position;
            if (WRITE) {
                on_sbeg = (uint32_t)a.start_off[ob_ + lane];
                if (accepted_only) on_acc = (a.acc_bits[(ob_ + lane) >> 5] >> ((ob_ + lane) & 31u)) & 1u;
            }
        }
    };
    if (have_n) fetch_next();
    while (have_n) {
        const uint32_t src = src_n;
        const EwRegs<G32, KMAX> R = Rn;
        const int o_frame = on_frame, o_stop = on_stop;
        const uint32_t o_acc = on_acc, o_sbeg = on_sbeg;
        have_n = todo != 0;
        if (have_n) fetch_next();
        const uint64_t it = blk * 64 + src;
        const uint64_t r = it >> 1;
        const bool fwd = (it & 1) == 0;
        const uint64_t off = ew_readlane64(l_off, src), ob = ew_readlane64(l_ob, src);
        const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)l_n, (int)src);
        const uint64_t oe = ob + (uint32_t)__builtin_amdgcn_readlane((int)l_no, (int)src);
        const int isl = a.read_isl ? a.read_isl[r] : a.ignore_score_len;

        // ---- the ORFs of this strand: level-0 calls onto the stack
        uint32_t top = 0, nloc = 0;
        bool overflow = false;
        for (uint64_t o0 = ob; o0 < oe; o0 += 64) {
            const uint64_t i = o0 + lane;
            const bool have = i < oe;
            int frame = o_frame, stop_position = o_stop;
            uint32_t acc = o_acc, sbeg = o_sbeg;
            if (o0 != ob && have) {                     // (a read with more than 64 ORFs: the next records)
                frame = a.orfs[i].frame; stop_position = a.orfs[i].stop_position;
                if (WRITE) {
                    sbeg = (uint32_t)a.start_off[i];
                    if (accepted_only) acc = (a.acc_bits[i >> 5] >> (i & 31u)) & 1u;
                }
            }
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
                a_best[idx] = mg_ord(-DBL_MAX);
                a_exa[idx] = a_exb[idx] = fwd ? ~0ull : 0ull;
                if (WRITE) {
                    a_cnt[idx] = 0;
                    a_m0[idx] = sbeg;
                    if (accepted_only && !acc) act = false;
                } else {
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
        uint32_t f_low_bits = 0;
        ew_build<G32, KMAX>(a, r, off, n, fwd, indels, lane, R, S, srow, Mstart, Mstop, Mlow, nw, msk, 5 * nw, s_q, f_low_bits);
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
        const uint32_t off_m3 = (uint32_t)(off % 3);

        // ---- the call tree
        uint32_t state = 0;                             // 0: no call, 1: walking, 2: the call has ended
        uint32_t x = 0, sj = 0, lidx = 0, level = 0, j0 = 0, cnt = 0, m_end = 0, ee = 0, last_own = MG_NO_SLOT;
        double ss = 0.0, p0 = 0.0, best = -DBL_MAX, last_sum = 0.0;
        const double *Sc = S;
        uint64_t key = 0;
        int last_pos = 0, last_j = 0;
        bool trunc = false, first_done = false;
        uint32_t st_trips = 0, st_calls = 0, st_deep = top;     // (mg_timing: trips, calls, deepest stack of this pair)
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
                st_calls += take;
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
                            }
                            last_pos = k; last_j = j_full;
                            if (sc > best) best = sc;
                            cnt++;
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
                if (cnt) {
                    if (!WRITE) atomicAdd(&a_cnt[lidx], cnt);
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
            st_trips++;
            if (top > st_deep) st_deep = top;
            if (top > qcap) { overflow = true; break; }
            wcs_sync();
        }
        if (stats && lane == 0) {
            atomicMax(&stats[0], st_deep); atomicMax(&stats[1], nloc); atomicAdd(&stats[2], st_trips); atomicAdd(&stats[3], st_calls);
            atomicAdd(&stats[4], 1u);
        }
        }
        if (overflow) { if (lane == 0) atomicOr(a.err_flag, 1u); continue; }
        if (WRITE) {
            // first_j and best_score of the ORFs whose starts were written (the walk-free count pass leaves them to this pass)
            wcs_sync();
            if (lane < nloc && a_cnt[lane]) {
                const uint64_t i = s_gi[lane];
                const uint32_t ja = (uint32_t)a_exa[lane], jb = 0xffffffffu - (uint32_t)a_exb[lane];
                const int jmin = (int)(fwd ? ja : jb), jmax = (int)(fwd ? jb : ja);
                a.orfs[i].first_j = jmin;
                if (jmax + 1 >= a.min_gene_len) a.orfs[i].best_score = mg_unord(a_best[lane]);
            }
            wcs_sync();
            continue;
        }

        // ---- Score_Orfs_Errors' verdict per ORF (:1647-1683; as k_mg_err_verdict)
        wcs_sync();
        bool kept = false;
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
                if (rec.accepted) { atomicOr(&a.acc_bits[i >> 5], 1u << (i & 31u)); kept = true; }
                if (!(accepted_only && !rec.accepted)) { rec.start_begin = 0; a.orfs[i] = rec; }
            }
        }
        if (__ballot(kept) && lane == 0) item_flag[it] = 1;            // (the write pass takes this pair)
        wcs_sync();
    }
    }
}

// ---------------------------------------------------------------------------------------------------
// k_mg_err_wcount -- the COUNT pass without walks.  What the count pass must deliver per ORF is the NUMBER of starts its call tree
// pushes and whether any of them scores above Start_Threshold (Score_Orfs_Errors' filter, glimmer-mg.cc:1647-1683; first_j and
// best_score of the kept ORFs come from the write pass, which meets every start anyway).  With the running sums and the masks in LDS
// neither needs a walk:
//   * a call (first step x, suffix score ss, D = x - suffix_j) pushes one start per start codon of its phase between
//     max (x + 3, D + Min_Gene_Len - 3) and the last codon in front of the next in-phase stop codon: a popcount of the start mask
//     (+ 1 for the truncated start when the region runs into the read's end); their scores are ss + S[t] - S[x]: a few look-ups
//     at the set bits (ew_own);
//   * a call branches only at bases of low quality: the (call, low-quality base) pairs are enumerated DENSELY, one lane each --
//     level 0: every low-quality base in each of the three phases belongs to at most one ORF's region (found from the stop mask
//     backwards); level 1: the bases between the call's first step and its region's end, by rank in the sorted list of the
//     read's low-quality bases.  A pair evaluates its two candidates (insertion, deletion) and appends the children to the next
//     level's list (wave-wide ballots); level 2 never branches: its calls are ew_own alone.
// The tree is processed breadth first, every trip with full lanes: ~10 trips per (read, strand) instead of the ~15 dependent
// event trips of the stack walker at 28 % of the lanes (mg_timing prints both kernels' counts).  -s: level 0 + one child per ORF.
// ---------------------------------------------------------------------------------------------------
#define EWC_CAP1 128             // level-1 calls waiting (worked off whenever the next 64 pairs' children would not fit: two each at most)
#define EWC_CAP2 128             // level-2 calls waiting (the same)
#define EWC_PCAP 256             // (level-1 call, low-quality base) pairs of one batch of 64 calls, taken that many at a time
#define EWC_PMAX 160             // low-quality bases of one read
#ifndef EWC_ITEMS
#define EWC_ITEMS 64              // (read, strand) pairs per block (measured with the queue below: 32 and 16 cost 1 and 4 ms with -i)
#endif

struct EwcLayout {
    uint32_t S, msk, l1_ss, l2_ss, l1_w, l2_w, l1_x, pcall, plist, pq, cum, orf_at, a_cnt, a_m0, gi, xs, acc, bytes;
    uint32_t l1_key, l2_key, l1_e, l2_e, a_best, a_exa, a_exb;   // WRITE only
    uint32_t srow, nw, ncw;      // doubles per class row; words per mask row over the steps / over a phase's codons (one guard word in front, zero words behind)
};
// indels = false (the substitution branch alone: one child per ORF, taken by the ORF's lane at once; no (call, low-quality base)
// pairs, no level 2): the level lists, the pair tables, the low-quality list and the ORF-at-step table are not laid out
__host__ __device__ constexpr EwcLayout ewc_layout(uint32_t cap, bool write = false, bool indels = true)
{
    EwcLayout L = {};
    L.srow = cap + 4;
    L.nw = cap / 64 + 3;
    L.ncw = (cap / 3 + 63) / 64 + 2;
    const uint32_t cap1 = indels ? EWC_CAP1 : 0u, cap2 = indels ? EWC_CAP2 : 0u, pcap = indels ? EWC_PCAP : 0u, pmax = indels ? EWC_PMAX : 0u;
    uint32_t o = 0;
    L.S = o; o += 3 * L.srow * 8;
    L.msk = o; o += (L.nw + 6 * L.ncw) * 8;         // low-quality bases by step; start and stop codons by phase and codon
    L.l1_ss = o; o += cap1 * 8;
    L.l2_ss = o; o += cap2 * 8;
    L.acc = o; o += 8;
    L.l1_key = o; o += write ? cap1 * 2 : 0;        // (a call's order key in 13-bit fields: one for level 1, two for level 2)
    L.l2_key = o; o += write ? cap2 * 4 : 0;
    L.a_best = o; o += write ? EW_MAXO * 8 : 0;
    L.a_exa = o; o += write ? EW_MAXO * 8 : 0;
    L.a_exb = o; o += write ? EW_MAXO * 8 : 0;
    L.l1_e = o; o += write ? cap1 * 4 : 0;
    L.l2_e = o; o += write ? cap2 * 4 : 0;
    L.l1_w = o; o += cap1 * 4;
    L.l2_w = o; o += cap2 * 4;
    L.a_cnt = o; o += EW_MAXO * 4;
    L.a_m0 = o; o += EW_MAXO * 4;
    L.gi = o; o += EW_MAXO * 4;
    L.xs = o; o += EW_MAXO * 2;
    L.pcall = o; o += pcap * 2;
    L.plist = o; o += pmax * 2;
    L.cum = o; o += indels ? ((L.nw + 1) * 2 + 3) & ~3u : 0u;
    L.pq = o; o += pmax;
    L.l1_x = o; o += indels ? cap1 : 0u;            // (a rank in the list of low-quality bases: < EWC_PMAX)
    L.orf_at = o; o += indels ? (cap + 8 + 7) & ~7u : 0u;
    L.bytes = (o + 15) & ~15u;
    return L;
}

// the 64 steps from step t on of a mask row; t may be negative (the row has a zero guard word in front)
__device__ __forceinline__ uint64_t ewc_window(const uint64_t *row, int t)
{
    const int wi = t >> 6;
    const uint32_t sh = (uint32_t)t & 63u;
    const uint64_t lo = row[wi], hi = row[wi + 1];
    return (lo >> sh) | ((hi << 1) << (63u - sh));
}
__device__ __forceinline__ uint32_t ewc_scan_u32(uint32_t x)       // inclusive sum over the lanes of a wave
{
#define EWC_DPP_ADD(CTRL, RM) x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, RM, 0xf, false)
    EWC_DPP_ADD(0x111, 0xf); EWC_DPP_ADD(0x112, 0xf); EWC_DPP_ADD(0x114, 0xf); EWC_DPP_ADD(0x118, 0xf);
    EWC_DPP_ADD(0x142, 0xa); EWC_DPP_ADD(0x143, 0xc);
#undef EWC_DPP_ADD
    return x;
}

struct EwOwn { uint32_t cnt, t_last, m_end; bool acc, has, trunc; };
// What one Score_Orf_Starts call pushes itself (glimmer-mg.cc:1812-1861), from the masks: the number of starts and whether one of
// them scores above the threshold; the last codon of its region (has: there is one).  Exactly the stack walker's emissions:
// a start codon at j0 = t - x counts when j0 >= lowest_j (j0 >= 3: j0 is a multiple of 3 and lowest_j >= 1) and
// j0 + 3 + suffix_j >= Min_Gene_Len, i.e. t + 3 - D >= Min_Gene_Len; the truncated start sits on the last codon of a region that
// runs into the read's end.
// Cstart / Cstop: the masks by phase and codon (ew_build's compact form: row p at + p ncw, bit k = the codon that starts at step
// 3 k + p): a 64-bit window is 64 codons of the call's phase -- 192 steps
__device__ __forceinline__ EwOwn ew_own(const double *S, const uint32_t srow, const uint64_t *Cstart, const uint64_t *Cstop, const uint32_t ncw,
                                        const uint32_t n, const bool fwd, const uint32_t off_m3, const uint32_t x, const double ss, const int D,
                                        const int mgl, const int isl, const double thr, const bool allow_trunc)
{
    EwOwn o;
    o.cnt = 0; o.t_last = 0; o.m_end = 0; o.acc = false; o.has = false; o.trunc = false;
    if (x >= n) return o;
    if (n - x < 3) { o.trunc = allow_trunc; return o; }
    const uint32_t ph = x % 3u, k0 = x / 3u, kc = (n - ph) / 3u;      // kc: the codons of this phase that lie inside the read
    const uint64_t *cstop = Cstop + ph * ncw, *cstart = Cstart + ph * ncw;
    uint32_t s = 0;
    bool found = false;
    for (uint32_t k = k0; k < kc; k += 64u) {           // the first stop codon of the call's phase at or behind x
        const uint64_t win = ewc_window(cstop, (int)k);
        if (win) { s = 3u * (k + (uint32_t)__builtin_ctzll(win)) + ph; found = true; break; }
    }
    if (found && s == x) return o;
    const uint32_t t_last = found ? s - 3u : x + (n - 3u - x) / 3u * 3u;
    o.has = true; o.t_last = t_last; o.trunc = !found && allow_trunc; o.m_end = t_last + 3u - x;
    int jmin = D + mgl - 3 - (int)x;
    if (jmin < 3) jmin = 3;
    jmin = (jmin + 2) / 3 * 3;
    const uint32_t tq = x + (uint32_t)jmin;
    if (tq > t_last) return o;
    const uint32_t cls = (off_m3 + (fwd ? n - 1u - x : x)) % 3u;
    const double *Sc = S + cls * srow;
    const double p0 = Sc[x];
    const uint32_t kl = t_last / 3u;
    for (uint32_t k = tq / 3u; k <= kl; k += 64u) {
        uint64_t win = ewc_window(cstart, (int)k);
        const uint32_t span = kl - k;
        if (span < 63u) win &= (2ull << span) - 1ull;
        o.cnt += (uint32_t)__popcll(win);
        while (win) {
            const uint32_t tt = 3u * (k + (uint32_t)__builtin_ctzll(win)) + ph;
            win &= win - 1ull;
            const double raw = ((Sc[tt] - p0) - 0.0) + ss;
            const int j_full = (int)tt + 2 - D;
            const double sc = (j_full > isl && 0.0 > raw) ? 0.0 : raw;
            if (sc > thr) o.acc = true;
        }
    }
    if (o.trunc) {
        o.cnt++;
        const double raw = ((Sc[t_last] - p0) - 0.0) + ss;
        const int j_full = (int)t_last + 2 - D;
        const double sc = (j_full > isl && 0.0 > raw) ? 0.0 : raw;
        if (sc > thr) o.acc = true;
    }
    return o;
}

// The same call in the WRITE pass: every start it pushes goes to a slot of its ORF's slice (handed out by a counter in LDS; the
// order key restores the reference's push order afterwards, as for k_mg_err_level), with the Error_t entries of its path; the
// call's best score and its entry at the extreme pos are merged into the ORF's aggregates (first_j / best_score of the record).
// Returns the region's last codon through t_last / has (what the branching needs).
__device__ __forceinline__ void ew_own_write(const MgArgs &a, const double *S, const uint32_t srow, const uint64_t *Cstart, const uint64_t *Cstop,
                                             const uint32_t ncw, const uint32_t n, const bool fwd, const uint64_t off, const uint32_t off_m3, const uint32_t x, const double ss,
                                             const int D, const int mgl, const int isl, const bool allow_trunc, const uint32_t lidx, const uint32_t level,
                                             const uint64_t key, const uint32_t ee, uint32_t *fill, const uint32_t *sbeg, unsigned long long *a_best,
                                             unsigned long long *a_exa, unsigned long long *a_exb, const int8_t *s_which, uint32_t &t_last_out, bool &has_out)
{
    has_out = false; t_last_out = 0;
    if (x >= n || n - x < 3) return;
    const uint32_t ph = x % 3u, k0 = x / 3u, kc = (n - ph) / 3u;
    const uint64_t *cstop = Cstop + ph * ncw, *cstart = Cstart + ph * ncw;
    uint32_t s = 0;
    bool found = false;
    for (uint32_t k = k0; k < kc; k += 64u) {
        const uint64_t win = ewc_window(cstop, (int)k);
        if (win) { s = 3u * (k + (uint32_t)__builtin_ctzll(win)) + ph; found = true; break; }
    }
    if (found && s == x) return;
    const uint32_t t_last = found ? s - 3u : x + (n - 3u - x) / 3u * 3u;
    const bool trunc = !found && allow_trunc;
    has_out = true; t_last_out = t_last;
    int jmin = D + mgl - 3 - (int)x;
    if (jmin < 3) jmin = 3;
    jmin = (jmin + 2) / 3 * 3;
    const uint32_t tq = x + (uint32_t)jmin;
    if (tq > t_last) return;
    const uint32_t cls = (off_m3 + (fwd ? n - 1u - x : x)) % 3u;
    const double *Sc = S + cls * srow;
    const double p0 = Sc[x];
    uint32_t cnt = 0, last_own = MG_NO_SLOT;
    double best = -DBL_MAX;
    int last_pos = 0, last_j = 0;
    auto put = [&](uint32_t tt, int which, bool truncated) __attribute__((always_inline)) -> uint32_t {
        const uint32_t j0 = tt - x;
        const int k = fwd ? (int)n - (int)x - 2 - (int)j0 : (int)x + (int)j0 + 3;
        const double raw = ((Sc[tt] - p0) - 0.0) + ss;
        const int j_full = (int)tt + 2 - D;
        const double sc = (j_full > isl && 0.0 > raw) ? 0.0 : raw;
        const uint32_t slot = sbeg[lidx] + atomicAdd(&fill[lidx], 1u);
        gmg_start st;
        st.score = sc; st.j = j_full; st.pos = k; st.which = which; st.truncated = (int16_t)(truncated ? 1 : 0); st.first = (int16_t)(truncated ? 1 : 0);
        a.starts[slot] = st;
        gmg_start_errors er;
        er.pos[0] = level > 0 ? (int)((ee & 0x3fffu) >> 2) - 8 : 0; er.pos[1] = level > 1 ? (int)((ee >> 14) >> 2) - 8 : 0;
        er.type[0] = (int8_t)(level > 0 ? (ee & 3u) : 0); er.type[1] = (int8_t)(level > 1 ? ((ee >> 14) & 3u) : 0);
        er.n = (int8_t)level; er.reserved = 0;
        a.errs[slot] = er;
        a.keys[slot] = key | (uint64_t)((uint32_t)(2047 - (int)j0) << 2 | (truncated ? 2u : 3u)) << (26 - 13 * (int)level);
        last_pos = k; last_j = j_full;
        if (sc > best) best = sc;
        cnt++;
        return slot;
    };
    const uint32_t kl = t_last / 3u;
    for (uint32_t k = tq / 3u; k <= kl; k += 64u) {
        uint64_t win = ewc_window(cstart, (int)k);
        const uint32_t span = kl - k;
        if (span < 63u) win &= (2ull << span) - 1ull;
        while (win) {
            const uint32_t tt = 3u * (k + (uint32_t)__builtin_ctzll(win)) + ph;
            win &= win - 1ull;
            // which start codon: the codon's index as the walks form it
            const uint64_t g = fwd ? off + n - 1 - tt : off + tt;
            const uint32_t five = (uint32_t)dev_window_bits(a.packed, (int64_t)g - 2) & 0x3ffu, c0 = (five >> 4) & 3u;
            const int which = s_which[fwd ? c0 | ((five >> 2) & 3u) << 2 | (five & 3u) << 4 : (c0 | ((five >> 6) & 3u) << 2 | ((five >> 8) & 3u) << 4) ^ 63u];
            last_own = put(tt, which, false);
        }
    }
    if (trunc) put(t_last, -1, true);                   // (behind the codon's real start: kind 2 < kind 3 in the key)
    else if (last_own != MG_NO_SLOT) a.starts[last_own].first = 1;
    if (cnt) {
        atomicMax(&a_best[lidx], (unsigned long long)mg_ord(best));
        const unsigned long long pa = (unsigned long long)(uint32_t)(last_pos + 16) << 32 | (uint32_t)last_j,
                                 pb = (unsigned long long)(uint32_t)(last_pos + 16) << 32 | (0xffffffffu - (uint32_t)last_j);
        if (fwd) { atomicMin(&a_exa[lidx], pa); atomicMin(&a_exb[lidx], pb); }
        else { atomicMax(&a_exa[lidx], pa); atomicMax(&a_exb[lidx], pb); }
    }
}

#ifndef GMG_EW_STAMPS
#define GMG_EW_STAMPS 0          // diagnostic build: cycles per phase of k_mg_err_wcount<count>, summed over all waves (tools/ew_stamps.py); not in the product
#endif
#if GMG_EW_STAMPS
__device__ unsigned long long g_ew_stamps[8];
extern "C" int gmg_debug_ew_stamps(unsigned long long *out, int reset)
{
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; return hipMemcpyToSymbol(HIP_SYMBOL(g_ew_stamps), z, sizeof z) == hipSuccess ? 0 : -1; }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ew_stamps), 64) == hipSuccess ? 0 : -1;
}
#define EW_STAMP(i) do { if (!WRITE) { const unsigned long long now_ = __builtin_readcyclecounter(); st_acc[i] += now_ - st_prev; st_prev = now_; } } while (0)
#else
#define EW_STAMP(i) do { } while (0)
#endif

template <bool WRITE, bool G32, int KMAX, bool INDELS>
__global__ __launch_bounds__(EW_BLOCK) void k_mg_err_wcount(MgArgs a, const int accepted_only, const uint32_t cap_lo, const uint32_t cap,
                                                            uint8_t *item_flag, uint32_t *stats, uint32_t *queue)
{
    // (the LDS share is sized by the length class, 64 KMAX bases: every offset a constant; reads of cap_lo < n <= cap are taken;
    // INDELS = (a.err_mode == 1): without it the arrays of the branching levels are not laid out and their code is not compiled)
    constexpr EwcLayout L = ewc_layout(64u * (uint32_t)KMAX, WRITE, INDELS);
    __shared__ __attribute__((aligned(16))) unsigned char ew_lds[L.bytes];
    __shared__ double s_pen[INDELS ? 32 : 1];          // (penalties of the qualities a low-quality base can have)
    __shared__ float s_nt[G32 ? MG_NULL_FLOATS + 4 : 4];
    __shared__ int8_t s_which[WRITE ? 64 : 1];              // (which start codon: the write pass's records)
    const uint32_t lane = threadIdx.x;
    if (INDELS && lane < 32) s_pen[lane] = a.pen[lane];
    if (WRITE) s_which[lane] = a.which[lane];
    if (G32 && !a.read_null)
        for (uint32_t k = lane; k < MG_NULL_FLOATS; k += 64) s_nt[k] = a.null_tab[k];
    double *S = (double *)(ew_lds + L.S);
    uint64_t *msk = (uint64_t *)(ew_lds + L.msk);
    uint64_t *Mlow = msk + 1, *Cstart = msk + L.nw + 1, *Cstop = Cstart + 3 * L.ncw;          // (row[-1]: the guard word)
    constexpr uint32_t ncw = L.ncw;
    double *l1_ss = (double *)(ew_lds + L.l1_ss), *l2_ss = (double *)(ew_lds + L.l2_ss);
    uint32_t *l1_w = (uint32_t *)(ew_lds + L.l1_w), *l2_w = (uint32_t *)(ew_lds + L.l2_w);
    uint8_t *l1_x = ew_lds + L.l1_x;
    uint16_t *pcall = (uint16_t *)(ew_lds + L.pcall), *plist = (uint16_t *)(ew_lds + L.plist),
             *cum = (uint16_t *)(ew_lds + L.cum), *s_xs = (uint16_t *)(ew_lds + L.xs);
    uint8_t *pq = ew_lds + L.pq, *orf_at = ew_lds + L.orf_at;
    uint32_t *a_cnt = (uint32_t *)(ew_lds + L.a_cnt), *a_m0 = (uint32_t *)(ew_lds + L.a_m0), *s_gi = (uint32_t *)(ew_lds + L.gi);
    unsigned long long *acc_mask = (unsigned long long *)(ew_lds + L.acc);
    // WRITE: the lists carry the order key and the Error_t entries of every call; a_cnt = slots handed out inside the ORF's slice,
    // a_m0 = where the slice begins; the ORFs' best score / entry at the extreme pos for first_j and best_score
    uint16_t *l1_key = (uint16_t *)(ew_lds + L.l1_key);
    uint32_t *l2_key = (uint32_t *)(ew_lds + L.l2_key);
    uint32_t *l1_e = (uint32_t *)(ew_lds + L.l1_e), *l2_e = (uint32_t *)(ew_lds + L.l2_e);
    unsigned long long *a_best = (unsigned long long *)(ew_lds + L.a_best), *a_exa = (unsigned long long *)(ew_lds + L.a_exa),
                       *a_exb = (unsigned long long *)(ew_lds + L.a_exb);
    __syncthreads();
    const bool pen_lds = a.indel_q_thr < 32;
    const int mgl = a.min_gene_len;
    const int lowest_j = mgl - 3 < 3 ? mgl - 3 : 3;
    constexpr bool indels = INDELS;
    constexpr uint32_t CAP1 = INDELS ? EWC_CAP1 : 0u, CAP2 = EWC_CAP2;
    const uint32_t nw = L.nw, srow = L.srow;
    const uint64_t n_items = 2 * a.n_reads;
    const bool trunc_ok = a.allow_truncated != 0;
    const double thr = a.start_threshold;

#if GMG_EW_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = __builtin_readcyclecounter();
#endif
    // With -i the blocks of EWC_ITEMS (read, strand) pairs are handed out by a counter (*queue, zero at the launch; the next block's
    // number is asked for while this one is worked on): the length classes' launches share the device, work-groups that become
    // resident late -- when another class has finished -- then simply take fewer blocks (with a fixed stride they had a full share
    // left: the classes' launches ended up to 6 ms apart; -0.5 ms over seven interleaved process pairs, -0.3 .. +1.1 each; the queue's last
    // eighth in blocks of a quarter did not add to that).  With -s the launches end within a millisecond of each other as
    // they are, and the queue costs 0.8 ms (the first launch takes the whole device, the classes run one after the other): fixed stride.
    uint32_t blk_next = blockIdx.x;
    if (INDELS) { blk_next = 0; if (lane == 0) blk_next = atomicAdd(queue, 1u); }
    for (;;) {
    const uint64_t blk = (uint32_t)__builtin_amdgcn_readfirstlane((int)blk_next);
    const uint64_t blk_base = blk * EWC_ITEMS;
    if (blk_base >= n_items) break;
    if (INDELS) { if (lane == 0) blk_next = atomicAdd(queue, 1u); }
    else blk_next = (uint32_t)blk + gridDim.x;
    uint64_t l_off = 0, l_ob = 0;
    uint32_t l_n = 0, l_no = 0;
    bool elig = false;
    {
        const uint64_t my = blk_base + lane;
        if (lane < EWC_ITEMS && my < n_items) {
            const uint64_t r_ = my >> 1;
            l_off = a.read_off[r_];
            l_n = (uint32_t)(a.read_off[r_ + 1] - l_off);
            l_ob = a.read_orf_off[r_];
            const uint64_t no_ = a.read_orf_off[r_ + 1] - l_ob;
            l_no = no_ > 0xffffffffull ? 0xffffffffu : (uint32_t)no_;
            elig = l_n > cap_lo && l_n <= cap && l_no > 0;
            if (WRITE && accepted_only && elig) elig = item_flag[my] != 0;
        }
    }
    uint64_t todo = __ballot(elig);
    // The pairs of the block one after the other; what the NEXT pair needs from HBM (its lanes' walk steps, its ORF records) is asked
    // for before this one is worked on and waits in registers: the pair's own phases then wait for LDS alone.
    EwRegs<G32, KMAX> Rn;
    int on_frame = 0, on_stop = 0;
    uint32_t on_acc = 1, on_sbeg = 0;
    uint32_t src_n = 0;
    bool have_n = todo != 0;
    auto fetch_next = [&]() __attribute__((always_inline)) {
        src_n = (uint32_t)__builtin_ctzll(todo);
        todo &= todo - 1ull;
        const uint64_t it_ = blk_base + src_n;
        const uint64_t off_ = ew_readlane64(l_off, src_n), ob_ = ew_readlane64(l_ob, src_n);
        const uint32_t n_ = (uint32_t)__builtin_amdgcn_readlane((int)l_n, (int)src_n);
        const uint64_t oe_ = ob_ + (uint32_t)__builtin_amdgcn_readlane((int)l_no, (int)src_n);
        ew_load<G32, KMAX>(a, off_, n_, (it_ & 1) == 0, indels, lane, Rn);
        if (G32 && a.read_null) {                       // the read's own null model: 252 floats, four per lane
            const float *nt_ = a.null_tab + (size_t)a.read_null[it_ >> 1] * MG_NULL_FLOATS;
#pragma unroll
            for (int k = 0; k < 4; k++) Rn.ntv[k] = 4u * lane + (uint32_t)k < MG_NULL_FLOATS ? nt_[4u * lane + (uint32_t)k] : 0.0f;
        }
        on_frame = 0; on_stop = 0; on_acc = 1; on_sbeg = 0;
        if (ob_ + lane < oe_) {
            on_frame = a.orfs[ob_ + lane].frame; on_stop = a.orfs[ob_ + lane].stop_position;
            if (WRITE) {
                on_sbeg = (uint32_t)a.start_off[ob_ + lane];
                if (accepted_only) on_acc = (a.acc_bits[(ob_ + lane) >> 5] >> ((ob_ + lane) & 31u)) & 1u;
            }
        }
    };
    if (have_n) fetch_next();
    EW_STAMP(0);                                        // the block's reads and ORF ranges, the first pair's loads issued
    while (have_n) {
        const uint32_t src = src_n;
        const EwRegs<G32, KMAX> R = Rn;
        const int o_frame = on_frame, o_stop = on_stop;
        const uint32_t o_acc = on_acc, o_sbeg = on_sbeg;
#if GMG_EW_STAMPS
        __builtin_amdgcn_s_waitcnt(0);                  // (diagnostic build: what the wave waits here for its own loads, apart from issuing the next ones)
        EW_STAMP(7);
#endif
        have_n = todo != 0;
        if (have_n) fetch_next();
        EW_STAMP(1);                                    // the next pair's loads issued
        const uint64_t it = blk_base + src;
        const uint64_t r = it >> 1;
        const bool fwd = (it & 1) == 0;
        const uint64_t off = ew_readlane64(l_off, src), ob = ew_readlane64(l_ob, src);
        const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)l_n, (int)src);
        const uint64_t oe = ob + (uint32_t)__builtin_amdgcn_readlane((int)l_no, (int)src);
        const int isl = a.read_isl ? a.read_isl[r] : a.ignore_score_len;
        const uint32_t off_m3 = (uint32_t)(off % 3);

        // ---- sums, masks, the sorted list of the low-quality bases
        uint32_t f_low = 0;
        if (G32 && a.read_null) {
#pragma unroll
            for (int k = 0; k < 4; k++) s_nt[4u * lane + (uint32_t)k] = R.ntv[k];       // (s_nt has four spare floats)
            wcs_sync();
        }
        uint32_t qv_lane[KMAX];
        ew_build<G32, KMAX, true>(a, r, off, n, fwd, indels, lane, R, S, srow, Cstart, Cstop, Mlow, nw - 1, msk, nw + 6 * ncw, (uint8_t *)nullptr, f_low, s_nt, qv_lane, ncw);
        uint32_t npos = 0;
        bool overflow = false;
        if (indels) {
            const uint32_t mine = (uint32_t)__popc(f_low), incl = ewc_scan_u32(mine);
            npos = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (npos > EWC_PMAX) overflow = true;
            else {
                uint32_t o = incl - mine;
#pragma unroll
                for (int e = 0; e < KMAX; e++)
                    if ((f_low >> e) & 1u) { plist[o] = (uint16_t)(R.tb + (uint32_t)e); pq[o] = (uint8_t)qv_lane[e]; o++; }
                // low-quality bases in front of every 64-step word
                const uint32_t pw = lane < nw - 1 ? (uint32_t)__popcll(Mlow[lane]) : 0u, iw = ewc_scan_u32(pw);
                if (lane < nw) cum[lane] = (uint16_t)(iw - pw);
            }
        }
        if (indels)
            for (uint32_t t = lane; 4u * t < n + 8u; t += 64) ((uint32_t *)orf_at)[t] = 0xffffffffu;
        if (lane == 0) *acc_mask = 0;
        wcs_sync();
        EW_STAMP(2);                                    // sums, masks, lists (waits for the pair's loads)

        // ---- level 0: the ORFs of this strand.  First every ORF's first step is put down (s_xs; with -i the step's table entry names the
        //      ORF that takes the branches of that step: ORFs that begin at the same step -- Find_Orfs gives the reverse frames
        //      without a stop codon in front the same virtual stop -- have the same call tree), then one lane per ORF
        uint32_t nloc = 0, n1 = 0;
        for (uint64_t o0 = ob; o0 < oe && !overflow; o0 += 64) {
            const uint64_t i = o0 + lane;
            const bool have = i < oe;
            int frame = o_frame, stop_position = o_stop;
            uint32_t acc = o_acc, sbeg = o_sbeg;
            if (o0 != ob && have) {
                frame = a.orfs[i].frame; stop_position = a.orfs[i].stop_position;
                if (WRITE) {
                    sbeg = (uint32_t)a.start_off[i];
                    if (accepted_only) acc = (a.acc_bits[i >> 5] >> (i & 31u)) & 1u;
                }
            }
            const bool mine = have && (frame > 0) == fwd;
            const uint64_t mm = __ballot(mine);
            const uint32_t idx = nloc + __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
            nloc += (uint32_t)__popcll(mm);
            if (nloc > EW_MAXO) { overflow = true; break; }
            if (mine) {
                const int end_point = fwd ? stop_position - 1 : stop_position + 3;
                const int xs = fwd ? (int)n - end_point : end_point - 1;           // the call's first walk step
                bool act = true;
                if (WRITE) { if (accepted_only && !acc) act = false; }
                else if (accepted_only && ((int)n - xs) + 12 < mgl) act = false;     // (as k_mg_err_level: cannot reach Min_Gene_Len)
                if (xs < 0 || xs >= (int)n) act = false;
                s_gi[idx] = (uint32_t)i;
                a_cnt[idx] = 0;
                a_m0[idx] = WRITE ? sbeg : 0u;
                if (WRITE) { a_best[idx] = mg_ord(-DBL_MAX); a_exa[idx] = a_exb[idx] = fwd ? ~0ull : 0ull; }
                s_xs[idx] = act ? (uint16_t)xs : (uint16_t)0xffffu;
                if (act && indels && a.indel_max >= 1) orf_at[xs] = (uint8_t)idx;
            }
        }
        wcs_sync();
        if (!overflow) {
            bool child = false;
            double es_sub = 0.0;
            uint32_t child_w = 0, child_e = 0;
            const uint32_t xs = lane < nloc ? (uint32_t)s_xs[lane] : 0xffffu;
            if (xs != 0xffffu) {
                uint32_t m_end = 0;
                bool reach = false;
                if (WRITE) {
                    // (the ORF the step's table entry names writes the tree; the others at that step copy its slice at the end)
                    if (!(indels && a.indel_max >= 1) || orf_at[xs] == lane) {
                        uint32_t t_last = 0;
                        bool has = false;
                        ew_own_write(a, S, srow, Cstart, Cstop, ncw, n, fwd, off, off_m3, xs, 0.0, (int)xs, mgl, isl, trunc_ok, lane, 0u, 0ull, 0u, a_cnt, a_m0,
                                     a_best, a_exa, a_exb, s_which, t_last, has);
                        m_end = has ? t_last + 3u - xs : 0u;
                        reach = true;
                    }
                } else {
                    const EwOwn o = ew_own(S, srow, Cstart, Cstop, ncw, n, fwd, off_m3, xs, 0.0, (int)xs, mgl, isl, thr, trunc_ok);
                    a_cnt[lane] = o.cnt;
                    a_m0[lane] = o.m_end << 1 | (o.trunc ? 1u : 0u);
                    if (o.acc) atomicOr(acc_mask, 1ull << lane);
                    m_end = o.m_end;
                    reach = true;
                }
                if (reach && a.err_mode == 2 && xs + m_end + 3u <= n) {
                    // the substitution branch (:1771-1806): through the stop codon behind the region (steps x + m .. x + m + 2)
                    const uint32_t sa = xs + m_end;
                    const uint64_t g = fwd ? off + n - 1 - sa : off + sa;
                    const uint32_t five = (uint32_t)dev_window_bits(a.packed, (int64_t)g - 2) & 0x3ffu;
                    const int a2 = fwd ? ((five >> 4) & 3u) == 0u : ((five >> 4) & 3u) == 3u;
                    const int a1 = fwd ? ((five >> 2) & 3u) == 0u : ((five >> 6) & 3u) == 3u;
                    es_sub = 0.0 + a.pass_stop[a1 * 2 + a2];
                    if (m_end > 0) {
                        const uint32_t cls = (off_m3 + (fwd ? n - 1u - xs : xs)) % 3u;
                        es_sub += (S[cls * srow + sa] - S[cls * srow + xs]) - 0.0;               // score[m - 1]
                    }
                    if ((int)m_end + ((int)n - (int)(sa + 3u)) + 12 >= mgl) {
                        child = true;
                        child_w = (sa + 3u) | (xs + 3u) << 10 | lane << 21;                      // step | D << 10 | ORF << 21
                        // Error_t (lo - 2 / hi + 2, substitution): forward lo = n - x - m, reverse hi = x + 1 + m
                        const int epos = fwd ? (int)n - (int)xs - (int)m_end - 2 : (int)xs + (int)m_end + 3;
                        child_e = (uint32_t)(epos + 8) << 2 | 2u;
                    }
                }
            }
            if (!INDELS) {
                // -s: the ORF's one child (level 1, never branches) by the ORF's own lane, at once: no list (there is as many a child
                // as there are ORF lanes: nothing to pack) -- the lists' 768 bytes buy a 14th wave per CU in the first length class
                if (child) {
                    const uint32_t x1 = child_w & 1023u;
                    const int D1 = (int)((child_w >> 10) & 2047u);
                    if (WRITE) {
                        uint32_t tl; bool hs;
                        ew_own_write(a, S, srow, Cstart, Cstop, ncw, n, fwd, off, off_m3, x1, es_sub, D1, mgl, isl, trunc_ok, lane, 1u, 0ull, child_e, a_cnt, a_m0,
                                     a_best, a_exa, a_exb, s_which, tl, hs);      // (key field 0: before every position of the call)
                    } else {
                        const EwOwn oc = ew_own(S, srow, Cstart, Cstop, ncw, n, fwd, off_m3, x1, es_sub, D1, mgl, isl, thr, trunc_ok);
                        if (oc.cnt) a_cnt[lane] += oc.cnt;
                        if (oc.acc) atomicOr(acc_mask, 1ull << lane);
                    }
                }
            } else {
            const uint64_t cm = __ballot(child);
            if (child) {
                const uint32_t e = n1 + __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0u));
                if (e < CAP1) {
                    l1_ss[e] = es_sub; l1_w[e] = child_w;
                    if (WRITE) { l1_key[e] = 0; l1_e[e] = child_e; }      // (key field 0: before every position of the call)
                }
            }
            n1 += (uint32_t)__popcll(cm);
            }
        }
        if (n1 > CAP1) overflow = true;
        wcs_sync();

        EW_STAMP(3);                                    // level 0
        // ---- levels 1 and 2.  The level-1 list holds CAP1 calls and the level-2 list CAP2: whenever the children of the next 64
        //      pairs would not fit, the calls listed so far are worked off first (level 1: their own starts, their (call, low-quality
        //      base) pairs, the children of those at level 2) and the same 64 pairs are evaluated again; the order in which calls are
        //      taken is free (COUNT adds up, WRITE carries order keys).  Short lists are what lets ten waves share a CU's LDS.
        uint32_t n2 = 0, n1_max = 0;
        auto drain2 = [&]() __attribute__((always_inline)) {
            wcs_sync();
            EW_STAMP(5);                                // (what came before: level 1 and the pairs)
            for (uint32_t b0 = 0; b0 < n2; b0 += 64) {
                const uint32_t i = b0 + lane;
                if (i < n2) {
                    const uint32_t w = l2_w[i];
                    if (WRITE) {
                        uint32_t tl; bool hs;
                        ew_own_write(a, S, srow, Cstart, Cstop, ncw, n, fwd, off, off_m3, w & 1023u, l2_ss[i], (int)((w >> 10) & 2047u), mgl, isl, trunc_ok, w >> 21, 2u,
                                     (uint64_t)l2_key[i] << 13, l2_e[i], a_cnt, a_m0, a_best, a_exa, a_exb, s_which, tl, hs);
                    } else {
                        const EwOwn o = ew_own(S, srow, Cstart, Cstop, ncw, n, fwd, off_m3, w & 1023u, l2_ss[i], (int)((w >> 10) & 2047u), mgl, isl, thr, trunc_ok);
                        if (o.cnt) atomicAdd(&a_cnt[w >> 21], o.cnt);
                        if (o.acc) atomicOr(acc_mask, 1ull << (w >> 21));
                    }
                }
            }
            n2 = 0;
            wcs_sync();
            EW_STAMP(4);                                // level 2: the calls' own starts
        };
        const bool expand1 = indels && a.indel_max >= 2;
        auto level1 = [&]() __attribute__((always_inline)) {
            wcs_sync();
            if (n1 > n1_max) n1_max = n1;
            for (uint32_t b0 = 0; b0 < n1 && !overflow; b0 += 64) {
                const uint32_t i = b0 + lane;
                uint32_t nq = 0;
                if (i < n1) {
                    const uint32_t w = l1_w[i], x1 = w & 1023u;
                    const int D1 = (int)((w >> 10) & 2047u);
                    uint32_t o_t_last = 0;
                    bool o_has = false;
                    if (WRITE)
                        ew_own_write(a, S, srow, Cstart, Cstop, ncw, n, fwd, off, off_m3, x1, l1_ss[i], D1, mgl, isl, trunc_ok, w >> 21, 1u, (uint64_t)l1_key[i] << 26, l1_e[i], a_cnt, a_m0,
                                     a_best, a_exa, a_exb, s_which, o_t_last, o_has);
                    else {
                        const EwOwn o = ew_own(S, srow, Cstart, Cstop, ncw, n, fwd, off_m3, x1, l1_ss[i], D1, mgl, isl, thr, trunc_ok);
                        if (o.cnt) atomicAdd(&a_cnt[w >> 21], o.cnt);
                        if (o.acc) atomicOr(acc_mask, 1ull << (w >> 21));
                        o_t_last = o.t_last; o_has = o.has;
                    }
                    if (expand1 && o_has) {
                        // the low-quality bases from the call's first step to the last base of its region, by rank
                        const uint32_t e = o_t_last + 3u;
                        const uint32_t r0 = cum[x1 >> 6] + (uint32_t)__popcll(Mlow[x1 >> 6] & ((1ull << (x1 & 63u)) - 1ull));
                        const uint32_t r1 = cum[e >> 6] + (uint32_t)__popcll(Mlow[e >> 6] & ((1ull << (e & 63u)) - 1ull));
                        nq = r1 - r0;
                        l1_x[i] = (uint8_t)r0;
                    }
                }
                if (!expand1) continue;
                const uint32_t incl = ewc_scan_u32(nq), T = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                for (uint32_t pb = 0; pb < T; pb += EWC_PCAP) {             // (the pair table holds EWC_PCAP pairs at a time)
                const uint32_t Tn = T - pb < EWC_PCAP ? T - pb : EWC_PCAP;
                for (uint32_t u = 0; __ballot(u < nq); u++) {
                    const uint32_t gq = incl - nq + u;
                    if (u < nq && gq >= pb && gq < pb + EWC_PCAP) pcall[gq - pb] = (uint16_t)(lane | u << 6);
                }
                wcs_sync();
                for (uint32_t q0 = 0; q0 < Tn;) {
                    const uint32_t qi = q0 + lane;
                    bool pi = false, pd = false;
                    double es_i = 0.0, es_d = 0.0;
                    uint32_t t = 0, lidx = 0, jj = 0, x1k = 0, e1 = 0;
                    uint32_t key1 = 0;
                    int D1 = 0;
                    if (qi < Tn) {
                        const uint32_t pc = pcall[qi], ci = b0 + (pc & 63u), u = pc >> 6;
                        if (WRITE) { key1 = l1_key[ci]; e1 = l1_e[ci]; }
                        const uint32_t w = l1_w[ci], x1 = w & 1023u;
                        D1 = (int)((w >> 10) & 2047u); lidx = w >> 21;
                        const double ss1 = l1_ss[ci];
                        const uint32_t k = (uint32_t)l1_x[ci] + u, p = plist[k];
                        const uint32_t pj = (p - x1) % 3u;
                        t = p - pj;
                        const uint32_t j0 = t - x1, j = j0 + pj;
                        jj = j; x1k = x1;
                        if ((int)j >= lowest_j) {
                            const uint32_t cls = (off_m3 + (fwd ? n - 1u - x1 : x1)) % 3u;
                            const double *Sc = S + cls * srow;
                            const double p0 = Sc[x1], before = Sc[p] - p0, at = Sc[p + 1] - p0;
                            const int q = pq[k];
                            const double pen = pen_lds ? s_pen[q & 31] : a.pen[q];
                            es_i = ((ss1 + before) - 0.0) + pen;
                            es_d = ((ss1 + at) - 0.0) + pen;
                            const int c_sj = ((int)x1 - D1) + (int)j0 + 2;
                            pi = c_sj + ((int)n - (int)(t + 4u)) + 12 >= mgl && es_i > a.indel_suffix_thr;
                            pd = c_sj + ((int)n - (int)(t + 2u)) + 12 >= mgl && es_d > a.indel_suffix_thr;
                        }
                    }
                    const uint64_t mi = __ballot(pi), md = __ballot(pd);
                    if (n2 + (uint32_t)__popcll(mi) + (uint32_t)__popcll(md) > CAP2) { drain2(); continue; }    // (the list is empty then: these lanes' children fit)
                    if (pi) {
                        const uint32_t e = n2 + __builtin_amdgcn_mbcnt_hi((uint32_t)(mi >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mi, 0u));
                        l2_ss[e] = es_i; l2_w[e] = (t + 4u) | (uint32_t)(D1 + 2) << 10 | lidx << 21;
                        if (WRITE) {
                            const int kj = fwd ? (int)n - (int)x1k - 2 - (int)jj : (int)x1k + 3 + (int)jj;
                            l2_key[e] = key1 << 13 | ((uint32_t)(2047 - (int)jj) << 2 | 1u);
                            l2_e[e] = (e1 & 0x3fffu) | ((uint32_t)((fwd ? kj + 2 : kj - 2) + 8) << 2 | 0u) << 14;
                        }
                    }
                    n2 += (uint32_t)__popcll(mi);
                    if (pd) {
                        const uint32_t e = n2 + __builtin_amdgcn_mbcnt_hi((uint32_t)(md >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)md, 0u));
                        l2_ss[e] = es_d; l2_w[e] = (t + 2u) | (uint32_t)D1 << 10 | lidx << 21;
                        if (WRITE) {
                            const int kj = fwd ? (int)n - (int)x1k - 2 - (int)jj : (int)x1k + 3 + (int)jj;
                            l2_key[e] = key1 << 13 | ((uint32_t)(2047 - (int)jj) << 2 | 0u);
                            l2_e[e] = (e1 & 0x3fffu) | ((uint32_t)((fwd ? kj + 3 : kj - 1) + 8) << 2 | 1u) << 14;
                        }
                    }
                    n2 += (uint32_t)__popcll(md);
                    q0 += 64;
                }
                wcs_sync();
                }
            }
            if (n2 && !overflow) drain2();
            n1 = 0;
            wcs_sync();
        };
        // level 0 -> 1: every (low-quality base, phase) pair
        const uint32_t n_ids = indels && a.indel_max >= 1 && !overflow ? 3u * npos : 0u;
        for (uint32_t i0 = 0;;) {
            const bool more = i0 < n_ids;
            const uint32_t id = i0 + lane;
            bool pi = false, pd = false;
            double es_i = 0.0, es_d = 0.0;
            uint32_t t = 0, xs = 0, lidx = 0, jj = 0;
            if (more && id < n_ids) {
                const uint32_t k = id / 3u, phi = id - 3u * k, p = plist[k];
                const uint32_t pj = (p + 3u - phi) % 3u;
                if (pj <= p && p - pj + 2u < n) {
                    t = p - pj;
                    const uint64_t *cstop = Cstop + phi * ncw;
                    const uint32_t kt = t / 3u;             // (t is a step of phase phi)
                    if (!((cstop[kt >> 6] >> (kt & 63u)) & 1ull)) {
                        // the nearest stop codon of the phase in front of t: the region's call begins behind it
                        int u = (int)kt - 1;
                        xs = phi;
                        while (u >= 0) {
                            const uint64_t m = ewc_window(cstop, u - 63);
                            if (m) { xs = 3u * (uint32_t)(u - __builtin_clzll(m)) + phi + 3u; break; }
                            u -= 64;
                        }
                        lidx = orf_at[xs];
                        const uint32_t j0 = t - xs, j = j0 + pj;
                        jj = j;
                        if (lidx != 255u && (int)j >= lowest_j) {
                            const uint32_t cls = (off_m3 + (fwd ? n - 1u - xs : xs)) % 3u;
                            const double *Sc = S + cls * srow;
                            const double p0 = Sc[xs], before = Sc[p] - p0, at = Sc[p + 1] - p0;
                            const int q = pq[k];
                            const double pen = pen_lds ? s_pen[q & 31] : a.pen[q];
                            es_i = ((0.0 + before) - 0.0) + pen;
                            es_d = ((0.0 + at) - 0.0) + pen;
                            const int c_sj = (int)j0 + 2;
                            pi = c_sj + ((int)n - (int)(t + 4u)) + 12 >= mgl && es_i > a.indel_suffix_thr;
                            pd = c_sj + ((int)n - (int)(t + 2u)) + 12 >= mgl && es_d > a.indel_suffix_thr;
                        }
                    }
                }
            }
            const uint64_t mi = __ballot(pi), md = __ballot(pd);
            if (!more || n1 + (uint32_t)__popcll(mi) + (uint32_t)__popcll(md) > CAP1) {
                level1();                               // (... and the list is empty)
                if (!more) break;
                continue;
            }
            if (pi) {
                const uint32_t e = n1 + __builtin_amdgcn_mbcnt_hi((uint32_t)(mi >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mi, 0u));
                l1_ss[e] = es_i; l1_w[e] = (t + 4u) | (xs + 2u) << 10 | lidx << 21;
                if (WRITE) {                            // Error_t of an insertion: k + 2 / k - 2 at the start's pos k of position j (Score_Indels)
                    const int kj = fwd ? (int)n - (int)xs - 2 - (int)jj : (int)xs + 3 + (int)jj;
                    l1_key[e] = (uint16_t)((uint32_t)(2047 - (int)jj) << 2 | 1u);
                    l1_e[e] = (uint32_t)((fwd ? kj + 2 : kj - 2) + 8) << 2 | 0u;
                }
            }
            n1 += (uint32_t)__popcll(mi);
            if (pd) {
                const uint32_t e = n1 + __builtin_amdgcn_mbcnt_hi((uint32_t)(md >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)md, 0u));
                l1_ss[e] = es_d; l1_w[e] = (t + 2u) | xs << 10 | lidx << 21;
                if (WRITE) {                            // ... of a deletion: k + 3 / k - 1
                    const int kj = fwd ? (int)n - (int)xs - 2 - (int)jj : (int)xs + 3 + (int)jj;
                    l1_key[e] = (uint16_t)((uint32_t)(2047 - (int)jj) << 2 | 0u);
                    l1_e[e] = (uint32_t)((fwd ? kj + 3 : kj - 1) + 8) << 2 | 1u;
                }
            }
            n1 += (uint32_t)__popcll(md);
            i0 += 64;
        }
        EW_STAMP(5);                                    // levels 0 -> 1, 1 and 2
        if (stats && lane == 0) { atomicMax(&stats[0], n1_max); atomicMax(&stats[1], nloc); atomicMax(&stats[2], npos); atomicAdd(&stats[4], 1u); }
        if (overflow) { if (lane == 0) atomicOr(a.err_flag, 1u); continue; }
        if (WRITE) {
            // first_j and best_score of the ORFs whose starts were written; an ORF that shares its first step with the one that wrote the
            // tree gets a copy of that slice (same tree, same order keys) and of its two values
            wcs_sync();
            if (lane < nloc && s_xs[lane] != 0xffffu) {
                uint32_t src_l = lane;
                if (indels && a.indel_max >= 1 && orf_at[s_xs[lane]] != lane) src_l = orf_at[s_xs[lane]];
                const uint32_t cnt_w = a_cnt[src_l];
                if (cnt_w) {
                    const uint64_t i = s_gi[lane];
                    const uint32_t ja = (uint32_t)a_exa[src_l], jb = 0xffffffffu - (uint32_t)a_exb[src_l];
                    const int jmin = (int)(fwd ? ja : jb), jmax = (int)(fwd ? jb : ja);
                    a.orfs[i].first_j = jmin;
                    if (jmax + 1 >= a.min_gene_len) a.orfs[i].best_score = mg_unord(a_best[src_l]);
                    if (src_l != lane) {
                        __threadfence();                // (the slice was written by other lanes of this wave: visible before it is read back)
                        const uint32_t from = a_m0[src_l], to = a_m0[lane];
                        for (uint32_t k = 0; k < cnt_w; k++) { a.starts[to + k] = a.starts[from + k]; a.errs[to + k] = a.errs[from + k]; a.keys[to + k] = a.keys[from + k]; }
                    }
                }
            }
            wcs_sync();
            continue;
        }

        // ---- Score_Orfs_Errors' verdict per ORF (:1647-1683): kept = one start above Start_Threshold (every pushed start passes
        //      the length test, glimmer-mg.cc:1821); first_j and best_score come from the write pass
        wcs_sync();
        bool kept = false;
        if (lane < nloc) {
            uint32_t g_cnt = a_cnt[lane], src_l = lane;
            const uint64_t i = s_gi[lane];
            {
                // ORFs that begin at the same step (Find_Orfs gives the reverse frames without a stop codon in front the same virtual
                // stop) have the same call tree; the branches were credited to the one the step's table entry names
                const uint32_t xs = s_xs[lane];
                if (xs != 0xffffu && indels && a.indel_max >= 1 && orf_at[xs] != lane) { src_l = orf_at[xs]; g_cnt = a_cnt[src_l]; }
            }
            const bool accepted = g_cnt && ((*acc_mask >> src_l) & 1ull);
            if (!(accepted_only && !accepted)) {        // (accepted_only: a rejected ORF's record is never read again, its count stays 0)
                gmg_mg_orf rec = a.orfs[i];
                const uint32_t g_m0 = a_m0[lane];
                const int m0 = (int)(g_m0 >> 1);
                if (fwd) { rec.hi = rec.stop_position - 1; rec.lo = rec.hi - m0; }
                else { rec.lo = rec.stop_position + 3; rec.hi = rec.lo + m0; }
                rec.orf_is_truncated = (int16_t)(g_m0 & 1u);
                rec.n_starts = g_cnt;
                rec.first_j = 0; rec.best_score = -DBL_MAX;
                rec.accepted = (int16_t)(accepted ? 1 : 0);
                a.orf_cnt[i] = g_cnt;
                if (accepted) { atomicOr(&a.acc_bits[i >> 5], 1u << (i & 31u)); kept = true; }
                rec.start_begin = 0;
                a.orfs[i] = rec;
            }
        }
        if (__ballot(kept) && lane == 0) item_flag[it] = 1;
        wcs_sync();
        EW_STAMP(6);                                    // verdicts
    }
    }
#if GMG_EW_STAMPS
    if (!WRITE && lane == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&g_ew_stamps[i], st_acc[i]);
#endif
}

#endif
