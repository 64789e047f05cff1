// gmg_mg_orfbits.h -- Find_Orfs (glimmer_base.cc:638-817, linear sequences without ignore regions) on bit masks: part of gmg_mg.hip.
//
// k_mg_find_orfs / k_mg_find_orfs_ev give a read to ONE lane, which follows the reference's scan base by base (500 dependent steps;
// the records of a wave's 64 reads leave as 64 scattered 56-byte stores).  Here a WAVE takes a window of whole reads (<= 5,120 bases:
// ten 500-bp reads) and works in two shapes:
//   1. a lane per 32 bases: the four codon tests (forward / reverse start, forward / reverse stop: MgArgs::fwd_start ...) of the lane's
//      32 codons come from 16 look-ups of a 256-entry table (four bases -> two codons x four tests) and land in four 32-bit masks in
//      LDS: bit b of word w = the codon whose LAST base is base 32 w + b of the window;
//   2. six lanes per read: the reference keeps its state per class (position mod 3) and strand (first_fwd_start, last_rev_start,
//      prev_*_stop, glimmer_base.cc:647-652) and the classes never meet -- so a lane takes ONE class of ONE strand of a read and hops
//      from stop codon to stop codon of it (the next set bit of every third bit of the stop mask); the first / last start codon of a
//      region and the number of start codons between two positions are a bit scan and popcounts of the start mask.  The lane ends with
//      its part of Finish_Orfs (:783-817: the open reverse ORF of its class) or with the virtual stop codon past the read's end
//      (:765-776);
//   3. the reference's order inside a read -- by position, forward before reverse, then Finish_Orfs -- is a popcount: every ORF sets
//      a bit at its stop codon's position in an emission mask; a record's place is the number of bits in front of it.
// The count pass leaves ORFs per read (-> scan); the write pass walks twice (marks, then records) and writes the same records as
// k_mg_find_orfs<true>, in the same order, with the same number of starts per ORF (count_starts).  Batches it takes: no read longer
// than OB_MAX_LEN (mg_run decides; the per-read kernels stay for the others and as the cross-check: option mg_orfs_bits = 0).
#pragma once

#define OB_WORDS 160             // words of 32 bases under a wave
#define OB_SPAN (32 * OB_WORDS)
#define OB_MAX_LEN 1024          // longest read of a batch this kernel takes
#define OB_WAVES 4               // (independent) waves per work-group
#define OB_GROUP 10              // reads a wave walks at a time (six lanes each)

struct ObParams {
    int mgl, trunc, err_mode, min_indel_orf_len, counting, k0, j_lo;
};
struct ObOrf {
    int stop_position, frame, gene_len, orf_len, lo, hi, n_real;
    bool emit;
};

__device__ __forceinline__ uint32_t ob_mod3(uint32_t x) { return x - 3u * ((x * 43691u) >> 17); }        // x < 2^16

// largest x' < x (INCL: <= x) of x's class (x' = x mod 3) with lo <= x' whose bit is set in M; -1: none
template <bool INCL>
__device__ __forceinline__ int ob_prev(const uint32_t *M, const int x, const int lo)
{
    if (x < lo) return -1;
    int w = x >> 5;
    const uint32_t b = (uint32_t)x & 31u;
    uint32_t k = ob_mod3(b);                            // the class's bits of word w: b' = k mod 3; one word down: k + 2
    uint32_t m = M[w] & (0x49249249u << k) & (INCL ? (2u << b) - 1u : (1u << b) - 1u);
    const int wlo = lo >> 5;
    while (m == 0 && w > wlo) {
        w--;
        k = k == 0 ? 2u : k - 1u;
        m = M[w] & (0x49249249u << k);
    }
    if (m == 0) return -1;
    const int xp = (w << 5) + 31 - __clz((int)m);
    return xp >= lo ? xp : -1;
}
// smallest x' in [lo, hi] with x' = c mod 3 whose bit is set in M; -1: none
__device__ __forceinline__ int ob_first(const uint32_t *M, const int lo, const int hi, const uint32_t c)
{
    if (lo > hi) return -1;
    int w = lo >> 5;
    const int whi = hi >> 5;
    uint32_t k = ob_mod3(c + (uint32_t)w);              // 32 w + b = c mod 3  <=>  b = c + w mod 3
    uint32_t m = M[w] & (0x49249249u << k) & ~((1u << ((uint32_t)lo & 31u)) - 1u);
    while (m == 0 && w < whi) {
        w++;
        k = k == 2 ? 0u : k + 1u;
        m = M[w] & (0x49249249u << k);
    }
    if (m == 0) return -1;
    const int xf = (w << 5) + __ffs((int)m) - 1;
    return xf <= hi ? xf : -1;
}
// how many x' in [lo, hi] with x' = c mod 3 have their bit set in M1 and not in M2
__device__ __forceinline__ int ob_count(const uint32_t *M1, const uint32_t *M2, const int lo, const int hi, const uint32_t c)
{
    if (lo > hi) return 0;
    int w = lo >> 5;
    const int whi = hi >> 5;
    uint32_t k = ob_mod3(c + (uint32_t)w);
    uint32_t range = ~((1u << ((uint32_t)lo & 31u)) - 1u);
    int cnt = 0;
    for (; w <= whi; w++) {
        if (w == whi) range &= (2u << ((uint32_t)hi & 31u)) - 1u;
        cnt += __popc(M1[w] & ~M2[w] & (0x49249249u << k) & range);
        range = ~0u;
        k = k == 2 ? 0u : k + 1u;
    }
    return cnt;
}
__device__ __forceinline__ bool ob_emits(const ObParams &P, const int gene_len, const int orf_len)
{
    return gene_len >= P.mgl || (P.err_mode && orf_len >= P.min_indel_orf_len);       // glimmer_base.cc:494,528,806
}

// One step of a lane's walk, for a read that begins at window base g0: the stop codon at window position x (fin: no stop codon --
// forward: the virtual one at i = n, n + 1 or n + 2, :765-776; reverse: Finish_Orfs for the class whose last position is x), prev = the
// class's previous stop codon (-1: none).
//   forward: Do_Fwd_Stop_Codon (glimmer_base.cc:460-504) + Handle_First_Forward_Stop, linear (:970-982)
//   reverse: Do_Rev_Stop_Codon (:506-537) + Handle_First_Reverse_Stop (:989-1015); Finish_Orfs (:783-817) + Handle_Last_Reverse_Stop (:1053-1066)
// FULL: the record's fields and the start count, else only o.emit
template <bool FULL>
__device__ __forceinline__ ObOrf ob_event(const uint32_t *FS, const uint32_t *RS, const uint32_t *FT, const uint32_t *RT, const int g0, const int n,
                                          const int x, const int prev, const bool fwd, const bool fin, const ObParams &P)
{
    ObOrf o;
    const int i = x - g0, lo2 = g0 + 2;                 // (a Codon_t with an empty position matches nothing: gene.cc:56,85)
    const uint32_t c = ob_mod3((uint32_t)(i + 3)), cx = ob_mod3((uint32_t)(x + 3));
    const int from = prev >= 0 ? prev + 1 : lo2, ip = prev - g0;
    o.emit = false;
    o.gene_len = o.n_real = 0;
    if (fwd) {
        int fwd_last;
        if (prev < 0) {
            o.orf_len = i - 2;
            o.orf_len -= o.orf_len % 3;
            fwd_last = c == 0 ? 0 : c == 1 ? 1 : -1;
        } else {
            fwd_last = ip;
            o.orf_len = i - ip - 3;
        }
        // (gene_len <= orf_len on this strand: a region too short for either test needs no look at its start codons)
        if (o.orf_len >= P.mgl || (P.err_mode && o.orf_len >= P.min_indel_orf_len)) {
            const int xf = ob_first(FS, from, fin ? x - 3 : x, cx);     // first_fwd_start + 1 (a stop's own codon may be a start: "-A")
            if (prev < 0) {
                o.gene_len = xf < 0 ? 0 : i - (xf - g0);
                if (P.trunc && o.gene_len < P.mgl) o.gene_len = o.orf_len;
            } else o.gene_len = xf < 0 ? (i - INT_MAX) - 1 : i - (xf - g0);
            o.emit = ob_emits(P, o.gene_len, o.orf_len);
        }
        if (FULL) {
            o.stop_position = i - 1;
            o.frame = 1 + (int)((c + 1u) % 3u);
            o.lo = fwd_last + 1;
            o.hi = i - 2;
            // count_starts: the start codons of the region that sit k0 codons or more in front of the stop (k_mg_find_orfs: fwd_older)
            o.n_real = P.counting && o.emit ? ob_count(FS, FT, from, x - 3 * P.k0, cx) : 0;
        }
        return o;
    }
    if (!fin && prev >= 0) {                            // (gene_len <= orf_len + 3 here: most regions are too short for either test)
        const int ol = i - ip - 3;
        if (ol + 3 < P.mgl && !(P.err_mode && ol >= P.min_indel_orf_len)) return o;
    }
    const int xs = ob_prev<true>(RS, x, from);          // last_rev_start + 1 (the stop's own codon may be a start)
    const int last_rev_start = xs < 0 ? 0 : (xs - g0) - 1;
    const int virt = c == 0 ? -1 : c == 1 ? 0 : -2;     // the virtual stop in front of the read
    int orf_stop;
    if (fin) {
        orf_stop = prev >= 0 ? ip - 1 : virt;
        o.orf_len = n - orf_stop - 2;
        o.orf_len -= o.orf_len % 3;
        o.gene_len = last_rev_start == 0 ? 0 : last_rev_start - orf_stop;
        if (P.trunc && o.gene_len < P.mgl) o.gene_len = o.orf_len;
        const int e = orf_stop + 2;                     // Rev_Next_Stop (glimmer-mg.cc:1436-1445), no stop left in the class
        if (e >= n) o.hi = e + 1;
        else { const int rc = (n - 1 - e) % 3; o.hi = (rc == 0 ? n - 1 : rc == 1 ? n - 2 : n) + 1; }
    } else {
        if (prev < 0) {
            if (!P.trunc) { orf_stop = 0; o.gene_len = 0; }
            else {
                orf_stop = (i - 1) % 3;
                if (orf_stop > 0) orf_stop -= 3;
                o.gene_len = last_rev_start - orf_stop;
            }
        } else {
            orf_stop = ip - 1;
            o.gene_len = last_rev_start - orf_stop;
        }
        o.orf_len = i - orf_stop - 4;
        o.hi = i - 1;
    }
    o.emit = ob_emits(P, o.gene_len, o.orf_len);
    if (FULL) {
        o.stop_position = orf_stop;
        o.frame = -1 - (int)((c + 1u) % 3u);
        o.lo = orf_stop + 3;
        // count_starts: the start codons from rev_from on (k_mg_find_orfs: rev_cnt), rev_from = previous stop (real or virtual) + 4 + lowest j
        const int rev_from = g0 + (prev >= 0 ? ip - 1 : virt) + 4 + P.j_lo;
        o.n_real = P.counting && o.emit ? ob_count(RS, RT, rev_from > from ? rev_from : from, x, cx) : 0;
    }
    return o;
}

__device__ __forceinline__ void ob_store(const MgArgs &a, const ObParams &P, const uint64_t slot, const uint32_t r, const int n, const ObOrf &o)
{
    gmg_mg_orf rec;
    rec.read = r; rec.frame = o.frame; rec.stop_position = o.stop_position;
    rec.orf_len = o.orf_len; rec.gene_len = o.gene_len; rec.lo = o.lo; rec.hi = o.hi;
    rec.first_j = 0; rec.start_begin = 0; rec.n_starts = 0; rec.accepted = 0; rec.orf_is_truncated = 0;
    rec.reserved = 0; rec.best_score = -DBL_MAX;
    a.orfs[slot] = rec;
    if (P.counting) {                                   // + the truncated start (mg_starts_one: has_trunc)
        const int m = o.hi - o.lo;
        const bool tr = P.trunc && (o.frame > 0 ? o.lo < 3 : n - (o.hi - 1) < 3);
        const int jmax = m >= 1 ? (m - 1) / 3 * 3 : -1;
        a.orf_cnt[slot] = (uint32_t)((m > 0 ? o.n_real : 0) + (tr && jmax >= P.j_lo ? 1 : 0));
    }
}

// win_bases > 0: ragged batch, window v = the reads that begin in [v win_bases, (v + 1) win_bases); else rpw reads each
template <bool WRITE>
__global__ __launch_bounds__(64 * OB_WAVES) void k_mg_find_orfs_bits(MgArgs a, const uint64_t n_windows, const uint32_t win_bases, const uint32_t rpw)
{
    __shared__ uint32_t s_tab[256];                     // four bases -> byte m: the two codons' membership in set m (fs, rs, ft, rt)
    __shared__ uint32_t s_msk[OB_WAVES][4][OB_WORDS];
    __shared__ uint32_t s_em[OB_WAVES][2][OB_WORDS + 1];        // an ORF was written out at this stop codon: forward, reverse
    __shared__ uint32_t s_pw[OB_WAVES][OB_WORDS + 1];           // ... how many in the words before
    {
        const uint32_t key = threadIdx.x;
        const uint32_t c0 = (key & 3u) << 4 | (key & 12u) | ((key >> 4) & 3u), k1 = key >> 2;
        const uint32_t c1 = (k1 & 3u) << 4 | (k1 & 12u) | ((k1 >> 4) & 3u);
        const uint64_t sets[4] = {a.fwd_start, a.rev_start, a.fwd_stop, a.rev_stop};
        uint32_t e = 0;
#pragma unroll
        for (int m = 0; m < 4; m++) e |= (uint32_t)(((sets[m] >> c0) & 1ull) | ((sets[m] >> c1) & 1ull) << 1) << (8 * m);
        s_tab[key] = e;
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t *FS = s_msk[wv][0], *RS = s_msk[wv][1], *FT = s_msk[wv][2], *RT = s_msk[wv][3];
    uint32_t *EMF = s_em[wv][0], *EMR = s_em[wv][1], *PW = s_pw[wv];
    ObParams P;
    P.mgl = a.min_gene_len; P.trunc = a.allow_truncated != 0; P.err_mode = a.err_mode; P.min_indel_orf_len = a.min_indel_orf_len;
    P.counting = WRITE && a.count_starts;
    P.j_lo = ((a.min_gene_len - 3 > 1 ? a.min_gene_len - 3 : 1) + 2) / 3 * 3;     // lowest j of a start, as in mg_starts_one
    P.k0 = 1 + P.j_lo / 3;
    if (WRITE && a.count_starts && blockIdx.x == 0 && threadIdx.x == 0) a.orf_cnt[a.n_orfs] = 0;      // (the scan's extra element: see k_mg_find_orfs)
    // the lane's part of a read: sub 0..2 forward class (i mod 3), 3..5 reverse class
    const uint32_t kk = lane / 6u, sub = lane - 6u * kk;
    const bool fwd = sub < 3u;
    const uint32_t cls = fwd ? sub : sub - 3u;

    for (uint64_t win = (uint64_t)blockIdx.x * OB_WAVES + wv; win < n_windows; win += (uint64_t)gridDim.x * OB_WAVES) {
        uint64_t ra, rb;
        if (win_bases) {
            ra = mg_lower_bound(a, win * win_bases);
            rb = mg_lower_bound(a, (win + 1) * win_bases);
            if (rb > a.n_reads) rb = a.n_reads;
        } else {
            ra = win * rpw;
            rb = ra + rpw < a.n_reads ? ra + rpw : a.n_reads;
        }
        if (ra >= rb) continue;
        const uint64_t G0 = a.read_off[ra], Wb = G0 & ~31ull;
        uint32_t nwords = (uint32_t)((a.read_off[rb] - Wb + 31) >> 5);
        if (nwords > OB_WORDS) nwords = OB_WORDS;       // (cannot happen: mg_run's test)
        wcs_sync();                                     // (the lanes are through with the window before)
        // ---- the four masks of every 32 codons
        for (uint32_t w = lane; w < nwords + 1u; w += 64) {
            EMF[w] = 0; EMR[w] = 0;
            if (w >= nwords) continue;
            const uint32_t *pw = a.packed + (Wb >> 4) + 2u * w;
            const uint32_t q0 = pw[-1], q1 = pw[0], q2 = pw[1];        // (guard words in front of the first read)
            const uint32_t s0 = q0 >> 28 | q1 << 4, s1 = q1 >> 28 | q2 << 4, s2 = q2 >> 28;
            uint32_t t[16];
#pragma unroll
            for (int j = 0; j < 16; j++) {
                uint32_t key;
                if (j < 7) key = (s0 >> (4 * j)) & 255u;
                else if (j == 7) key = (s0 >> 28 | s1 << 4) & 255u;
                else if (j < 15) key = (s1 >> (4 * (j - 8))) & 255u;
                else key = (s1 >> 28 | s2 << 4) & 255u;
                t[j] = s_tab[key];
            }
            uint32_t X[4];
#pragma unroll
            for (int g = 0; g < 4; g++) X[g] = t[4 * g] | t[4 * g + 1] << 2 | t[4 * g + 2] << 4 | t[4 * g + 3] << 6;
#pragma unroll
            for (int m = 0; m < 4; m++)
                s_msk[wv][m][w] = ((X[0] >> (8 * m)) & 255u) | ((X[1] >> (8 * m)) & 255u) << 8 | ((X[2] >> (8 * m)) & 255u) << 16 | ((X[3] >> (8 * m)) & 255u) << 24;
        }
        wcs_sync();
        const uint32_t *STOP = fwd ? FT : RT;
        // ---- ten reads at a time, six lanes each
        for (uint64_t k0 = ra; k0 < rb; k0 += OB_GROUP) {
            const uint64_t r = k0 + kk;
            const bool have = lane < 6u * OB_GROUP && r < rb;
            const uint64_t off = have ? a.read_off[r] : 0;
            const int n = have ? (int)(a.read_off[r + 1] - off) : 0;
            const int g0 = (int)(off - Wb);
            const bool act = have && n >= P.mgl;        // glimmer_base.cc:676-677
            const uint32_t cxw = ob_mod3((uint32_t)g0 + cls + 3u);         // the class's window positions mod 3
            // where the walk ends: forward at the virtual stop codon i = n, n + 1 or n + 2 of the class, reverse at the class's last position
            const int i_end = fwd ? n + (int)ob_mod3(cls + 3u - ob_mod3((uint32_t)n)) : n - 1 - (int)ob_mod3((uint32_t)(n - 1 + 3) - cls);
            const bool has_fin = fwd ? P.trunc != 0 : true;
            uint32_t fin_emit = 0;
            // one walk: MARK sets the emission bits, else the records are written
            auto walk = [&](auto MARK_, const uint64_t slot0, const uint32_t p_start, const uint32_t fin_slot) __attribute__((always_inline)) {
                constexpr bool MARK = decltype(MARK_)::value;
                int prev = -1, cur = g0 + 2;
                bool done = !act;
                while (__any(!done)) {
                    if (!done) {
                        const int xs = ob_first(STOP, cur, g0 + n - 1, cxw);
                        const bool fin = xs < 0;
                        const int x = fin ? g0 + i_end : xs;
                        if (!fin || has_fin) {
                            const ObOrf o = ob_event<!MARK>(FS, RS, FT, RT, g0, n, x, prev, fwd, fin, P);
                            if (o.emit) {
                                if (MARK) {
                                    if (fin) fin_emit = 1;
                                    else atomicOr(fwd ? &EMF[x >> 5] : &EMR[x >> 5], 1u << ((uint32_t)x & 31u));
                                } else {
                                    uint64_t slot;
                                    if (fin) slot = slot0 + fin_slot;
                                    else {
                                        const uint32_t w = (uint32_t)x >> 5, below = (1u << ((uint32_t)x & 31u)) - 1u;
                                        slot = slot0 + (PW[w] + (uint32_t)(__popc(EMF[w] & below) + __popc(EMR[w] & below)) - p_start) +
                                               (fwd ? 0u : (EMF[w] >> ((uint32_t)x & 31u)) & 1u);
                                    }
                                    ob_store(a, P, slot, (uint32_t)r, n, o);
                                }
                            }
                        }
                        if (fin) done = true;
                        else { prev = xs; cur = xs + 1; }
                    }
                }
            };
            walk(std::integral_constant<bool, true>(), 0, 0, 0);
            wcs_sync();
            // ---- ORFs in front of every word (the marks of the groups before are in: their reads lie in front)
            {
                uint32_t carry = 0;
                for (uint32_t w0 = 0; w0 < nwords + 1u; w0 += 64) {
                    const uint32_t w = w0 + lane;
                    const uint32_t c = w < nwords ? (uint32_t)(__popc(EMF[w]) + __popc(EMR[w])) : 0u;
                    const uint32_t incl = ewc_scan_u32(c);
                    if (w < nwords + 1u) PW[w] = carry + incl - c;
                    carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                }
            }
            wcs_sync();
            // the read's ORFs by position, then Finish_Orfs' (reverse classes 0, 1, 2), then the virtual stop codons' (i = n, n + 1, n + 2)
            auto upto = [&](int x) __attribute__((always_inline)) -> uint32_t {
                const uint32_t w = (uint32_t)x >> 5, below = (1u << ((uint32_t)x & 31u)) - 1u;
                return PW[w] + (uint32_t)(__popc(EMF[w] & below) + __popc(EMR[w] & below));
            };
            const uint32_t p_start = have ? upto(g0) : 0u, tot = have ? upto(g0 + n) - p_start : 0u;
            const uint64_t fl = __ballot(fin_emit != 0);
            const uint32_t grp = (uint32_t)(fl >> (6u * (kk < OB_GROUP ? kk : 0u))) & 63u;    // bit sub: the lane of that part has a last ORF
            // the parts that come before this lane's in that order
            uint32_t before = 0;
            if (fwd) {
                before = 0x38u;
                const uint32_t j = (uint32_t)(i_end - n);
                for (uint32_t j2 = 0; j2 < j; j2++) before |= 1u << ob_mod3((uint32_t)n + j2);
            } else before = ((1u << cls) - 1u) << 3;
            if (!WRITE) {
                if (act && sub == 0) a.read_cnt[r] = tot + (uint32_t)__popc(grp);
            } else {
                const uint64_t slot0 = act ? a.read_orf_off[r] : 0;
                walk(std::integral_constant<bool, false>(), slot0, p_start, tot + (uint32_t)__popc(grp & before));
            }
        }
    }
}
