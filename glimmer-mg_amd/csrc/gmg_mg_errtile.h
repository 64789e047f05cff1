// gmg_mg_errtile.h -- glimmer-mg's error branch (-i / -s: Score_Indels and the recursive Score_Orf_Starts,
// src/Glimmer/glimmer-mg.cc:1513-1602,1693-1861) with everything a walk reads held in LDS.  Part of gmg_mg.hip (included there,
// behind MgArgs and the level kernels whose walk it restates).
//
// The level kernels (k_mg_err_level) take score[j] of a call as a difference of two running sums and visit only the codons at
// which something can happen -- but the sums lived in HBM: 48 B/base written once (19 GB per 1M reads of ~400 bp) and then one
// 64-byte sector fetched per event for 8 - 32 useful bytes, the calls handed from level to level through HBM as well.  A read's
// calls never leave the read, and a strand's calls never leave the strand: so here a work-group takes a TILE -- a few consecutive
// whole reads, <= CAP bases, one strand -- builds in LDS what its walks will read
//     S[3][CAP]  the inclusive running sums of the three reading-frame classes along the strand's walk order, restarting with
//                every read (24 B/base, formed from the fp32 gene rows and the read's null model, or from the caller's fp64 table)
//     the bases in walk order (complemented on the reverse strand), the qualities, the run lengths of "nothing happens here"
// and then runs level 0 (the tile's ORFs), level 1 and level 2 one after the other with the calls of the next level queued in a
// slab of its own (global memory, a few KB per tile: it stays in the L2).  What an ORF's calls add up to is merged with LDS atomics
// and judged by the same work-group (Score_Orfs_Errors' verdict, :1647-1683): the 32-byte aggregates never exist in HBM either.
// Per 1M reads of ~400 bp the traffic of the count pass drops from ~80 GB (tables written and fetched, call records, aggregates)
// to the gene rows read once (9.6 GB) + the ORF records.
//
// Two launches as before: <count> (verdicts, starts per ORF), a scan, <write> (the kept ORFs' call trees are expanded again --
// the tables are rebuilt for the tiles that hold a kept ORF -- and every start goes to a slot of its ORF's slice with its order
// key; the segmented sort of mg_run puts the slices in push order).  Coordinates inside a tile: b = base - w0; walk index
// u = span - 1 - b on the forward strand (its walks run down the read), u = b on the reverse strand: every walk goes up in u.
// A call anchored at u0 belongs to class c = u0 % 3 and reads row c: score[j] = S[c][u0 + j] - S[c][u0 - 1]; at step u the row
// of class c adds Frame_Scores row ((u - c) % 3 + 1) % 3 of the strand (glimmer-mg.cc:561-604: f = 1, 2, 0, ...).
// Reads longer than CAP / 2 (and what a tile cannot take) keep read_fit = 0 and go to k_mg_err_flat as before; a full call slab
// raises err_flag and the whole batch repeats on the level kernels.

#ifndef MG_ET_CAP
#define MG_ET_CAP 2048           // bases per tile (68 KB of LDS: two work-groups per CU)
#endif
#define ET_BLOCK 256
#define ET_MAXR 64               // reads per tile
#define ET_MAXO 256              // ORF records (both strands) staged per batch of a tile
#ifndef ET_QCAP
#define ET_QCAP 8192             // calls per level a work-group's slab holds
#endif
#ifndef ET_BLK
#define ET_BLK 128               // calls a wave owns at a time
#endif
#ifndef ET_BATCH
#define ET_BATCH 16              // lanes that wait before a wave runs the take / finish code
#endif
static_assert(ET_MAXO == ET_BLOCK, "one lane per staged ORF record");
#define ET_CHUNK_TILES 16        // tile builder: one lane lays out the tiles of 16 x CAP bases

// the tiles: greedy runs of consecutive reads, restarted at every chunk of ET_CHUNK_TILES * cap bases (one lane per chunk: the
// chain of "where does the next tile begin" is sequential inside a chunk only).  read_fit[r] = 1 for the reads a tile takes.
__global__ __launch_bounds__(256) void k_et_tiles(MgArgs a, uint32_t cap, uint32_t longest, uint64_t chunk, uint64_t n_chunks,
                                                  MgTile *tiles, uint32_t *n_tiles, uint8_t *read_fit)
{
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_chunks; k += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t r0 = mg_lower_bound(a, k * chunk), r1 = mg_lower_bound(a, (k + 1) * chunk);
        if (r0 > a.n_reads) r0 = a.n_reads;
        if (r1 > a.n_reads) r1 = a.n_reads;
        if (k + 1 == n_chunks) r1 = a.n_reads;          // (empty reads at the very end start at `total`)
        uint32_t base = 0;
        for (int pass = 0; pass < 2; pass++) {
            uint32_t cnt = 0;
            MgTile cur;
            cur.nfit = 0; cur.w0 = 0; cur.first = 0; cur.span = 0; cur.pad = 0;
            uint64_t o0 = r0 < r1 ? a.read_off[r0] : 0;
            for (uint64_t r = r0; r < r1; r++) {
                const uint64_t o1 = a.read_off[r + 1];
                const uint64_t len = o1 - o0;
                if (len > longest) {
                    if (cur.nfit) { if (pass) tiles[base + cnt] = cur; cnt++; cur.nfit = 0; }
                    if (pass) read_fit[r] = 0;
                } else {
                    if (cur.nfit && (cur.nfit == ET_MAXR || o1 - cur.w0 > cap)) { if (pass) tiles[base + cnt] = cur; cnt++; cur.nfit = 0; }
                    if (!cur.nfit) { cur.w0 = o0; cur.first = (uint32_t)r; }
                    cur.nfit++;
                    cur.span = (uint32_t)(o1 - cur.w0);
                    if (pass) read_fit[r] = 1;
                }
                o0 = o1;
            }
            if (cur.nfit) { if (pass) tiles[base + cnt] = cur; cnt++; }
            if (pass == 0) base = cnt ? atomicAdd(n_tiles, cnt) : 0u;
        }
    }
}

template <int CAP>
struct EtLds {
    static constexpr int RS = CAP + 8;                  // row stride: 4 spare entries on both sides (a walk looks one back and three ahead)
    double S[3][RS];
    MgOrfAgg agg[ET_MAXO];
    double pen[64];
    uint32_t wpk[CAP / 16 + 4];                         // 2-bit codes in walk order (complemented on the reverse strand), 16 per word
    uint32_t q[(CAP + 16) / 4];                         // qualities in walk order, 4 per word
    uint32_t rq[(CAP + 16) / 4], rn[(CAP + 16) / 4];    // run lengths in walk order: with / without the low-quality bases as events
    uint32_t roff[ET_MAXR + 1];
    int32_t isl[ET_MAXR];
    uint32_t oinf[ET_MAXO];                             // ORF of the batch: index in the batch | read in the tile << 16
    int32_t oep[ET_MAXO];                               // its end_point
    uint32_t fill[ET_MAXO];                             // write pass: slots handed out inside the ORF's slice
    uint32_t n_o, qn[2], take[3], kept, item;
    int8_t which[64];
};

// one level of a tile's calls: the walk of k_mg_err_level<.., PFX = true> on the tables in LDS.  n_in calls: level 0 = the staged
// ORFs of the strand, else the entries of the level's queue.
template <bool WRITE, int LEVEL, int CAP>
__device__ __forceinline__ void et_level(const MgArgs &a, EtLds<CAP> &L, const bool fwd, const uint32_t span, const uint64_t orf_base,
                                         const uint32_t n_in, const MgCall *q_in, MgCall *q_out, const uint32_t qcap, const int accepted_only)
{
    constexpr int RS = EtLds<CAP>::RS;
    const int lane = threadIdx.x & 63;
    const int mgl = a.min_gene_len;
    const int lowest_j = mgl - 3 < 3 ? mgl - 3 : 3;
    const bool pen_lds = a.indel_q_thr < 64;
    const bool can_branch = LEVEL < 2 && a.err_mode == 1 && LEVEL < a.indel_max;          // Score_Indels may start calls from here
    const uint8_t *s_q = (const uint8_t *)L.q;
    const uint8_t *s_run = (const uint8_t *)(can_branch ? L.rq : L.rn);
    auto stopc = [&](uint32_t idx) __attribute__((always_inline)) { return (bool)((a.fwd_stop >> idx) & 1ull); };
    // the codon at walk index u (its three codes, first base lowest) and the one behind it: 12 bits of the walk-order stream
    auto codons = [&](uint32_t u) __attribute__((always_inline)) -> uint32_t {
        const uint32_t w = u >> 4;
        const uint64_t x = (uint64_t)L.wpk[w + 1] << 32 | L.wpk[w];
        return (uint32_t)(x >> (2u * (u & 15u))) & 0xfffu;
    };
    auto quals4 = [&](uint32_t u) __attribute__((always_inline)) -> uint32_t {
        const uint32_t w = u >> 2;
        const uint64_t x = (uint64_t)L.q[w + 1] << 32 | L.q[w];
        return (uint32_t)(x >> (8u * (u & 3u)));
    };
    for (;;) {
        uint32_t t_ = 0;
        if (lane == 0) t_ = atomicAdd(&L.take[LEVEL], (uint32_t)ET_BLK);
        const uint32_t blk = (uint32_t)__builtin_amdgcn_readfirstlane((int)t_);
        if (blk >= n_in) break;
        uint32_t next = blk;
        const uint32_t blk_end = blk + ET_BLK < n_in ? blk + ET_BLK : n_in;
        // the call a lane is walking
        uint32_t oe = 0, rl = 0;
        int end_point = 0, suffix_j = 0, n = 0, avail = 0;
        double suffix_score = 0.0;
        uint64_t key = 0;
        uint32_t e0 = 0, e1 = 0, u0 = 0;
        const double *Sc = &L.S[0][4];
        double p0 = 0.0;
        uint32_t nskip = 0;
        double s0 = 0.0, s1 = 0.0;
        uint32_t qw = 0;
        bool walking = false, finishing = false, is_last = false, trunc = false, first_done = false;
        int tp = 0, br = 0;
        uint32_t pidx = 0, last_own = MG_NO_SLOT, cnt = 0;
        int last_pos = 0, last_j = 0;
        double sum = 0.0, prev = 0.0, best = -DBL_MAX;

        auto emit = [&](double raw, int j_loc, int pos, int which, int truncated, int first, uint32_t kind) __attribute__((always_inline)) -> uint32_t {
            const int j_full = j_loc + 2 + suffix_j;
            const int isl = L.isl[rl];
            const double sc = (j_full > isl && 0.0 > raw) ? 0.0 : raw;
            uint32_t slot = MG_NO_SLOT;
            if (WRITE) {
                const uint64_t orf = orf_base + (L.oinf[oe] & 0xffffu);
                slot = (uint32_t)a.start_off[orf] + atomicAdd(&L.fill[oe], 1u);
                gmg_start s1_;
                s1_.score = sc; s1_.j = j_full; s1_.pos = pos; s1_.which = which; s1_.truncated = (int16_t)truncated; s1_.first = (int16_t)first;
                a.starts[slot] = s1_;
                gmg_start_errors er;
                er.pos[0] = LEVEL > 0 ? (int)(e0 >> 2) - 8 : 0; er.pos[1] = LEVEL > 1 ? (int)(e1 >> 2) - 8 : 0;
                er.type[0] = (int8_t)(LEVEL > 0 ? (e0 & 3) : 0); er.type[1] = (int8_t)(LEVEL > 1 ? (e1 & 3) : 0);
                er.n = LEVEL; er.reserved = 0;
                a.errs[slot] = er;
                a.keys[slot] = key | (uint64_t)((uint32_t)(2047 - j_loc) << 2 | kind) << (26 - 13 * LEVEL);
            } else {
                last_pos = pos; last_j = j_full;        // (inside one call pos moves with j: the entry at the extreme pos is the last one)
                if (sc > best) best = sc;
                cnt++;
            }
            return slot;
        };

        for (;;) {
            const uint64_t wm = __ballot(walking);
            const uint64_t fm = __ballot(finishing && !walking);
            const bool do_fin = fm && (__popcll(fm) >= ET_BATCH || !wm);
            const bool idle = !(walking || finishing);
            const uint64_t im = __ballot(idle);
            if (next < blk_end && (__popcll(im) >= ET_BATCH || !(wm | fm))) {              // idle lanes take the next calls of the block
                const uint32_t left = blk_end - next;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(im >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)im, 0u));
                if (idle && rank < left) {
                    const uint32_t i = next + rank;
                    bool active = true;
                    suffix_j = 0; suffix_score = 0.0; key = 0; e0 = e1 = 0;
                    if (LEVEL == 0) {
                        oe = i;
                        rl = L.oinf[i] >> 16;
                        end_point = L.oep[i];
                    } else {
                        const MgCall c = q_in[i];
                        suffix_score = __longlong_as_double((long long)c.w[0]);
                        key = c.w[1] & 0xffffffffffull; end_point = (int)((c.w[1] >> 40) & 0xfffu) - 8; suffix_j = (int)(c.w[1] >> 52);
                        rl = (uint32_t)c.w[2] & 0xffu;
                        oe = (uint32_t)c.w[3]; e0 = (uint32_t)(c.w[3] >> 32) & 0x3fffu; e1 = (uint32_t)(c.w[3] >> 46) & 0x3fffu;
                    }
                    const int rs = (int)L.roff[rl];
                    n = (int)L.roff[rl + 1] - rs;
                    // accepted ORFs only: an ORF whose end lies too close to the read's upstream end cannot reach Min_Gene_Len on any path
                    if (LEVEL == 0 && !WRITE && accepted_only && (fwd ? end_point : n - end_point + 1) + 12 < a.min_gene_len) active = false;
                    if (active) {
                        const int anchor = end_point - 1;
                        const bool inside = anchor >= 0 && anchor < n;
                        avail = fwd ? anchor + 1 : n - anchor;
                        is_last = false; trunc = false; first_done = false; walking = false;
                        tp = 0; br = 0; last_own = MG_NO_SLOT; cnt = 0;
                        sum = 0.0; prev = 0.0; best = -DBL_MAX;
                        if (inside) {
                            const uint32_t ba = (uint32_t)(rs + anchor);
                            u0 = fwd ? span - 1u - ba : ba;
                            Sc = &L.S[u0 % 3u][4];
                            p0 = (fwd ? anchor == n - 1 : anchor == 0) ? 0.0 : Sc[(int)u0 - 1];        // (the sums restart with every read)
                            if (avail < 3) trunc = a.allow_truncated != 0;
                            else {
                                pidx = codons(u0) & 63u;
                                walking = !stopc(pidx);
                            }
                            if (walking) nskip = s_run[u0];
                        }
                        finishing = true;
                    }
                }
                const uint32_t n_idle = __popcll(im);
                next += n_idle < left ? n_idle : left;
            }
            if (!__ballot(walking || finishing)) {
                if (next >= blk_end) break;
                continue;
            }
            bool want_push = false;
            int c_end = 0, c_sj = 0;
            uint32_t c_err = 0, c_field = 0;
            double c_score = 0.0;
            if (walking) {
                if (br == 0) tp += (int)nskip;          // over the codons at which nothing happens
                const int j0 = 3 * tp;
                const uint32_t u = u0 + (uint32_t)j0;
                if (br == 0) {
                    const uint32_t cc = codons(u);
                    const double *d = Sc + (int)u - 1;
                    const double d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3];
                    if (can_branch) qw = quals4(u);
                    pidx = cc & 63u;
                    if (avail - 3 * (tp + 1) < 3) { trunc = a.allow_truncated != 0; is_last = true; }
                    else is_last = stopc(cc >> 6);
                    prev = j0 ? d0 - p0 : 0.0;          // score[j0 - 1]
                    s0 = d1 - p0; s1 = d2 - p0; sum = d3 - p0;
                    if (j0 >= lowest_j && j0 + 3 + suffix_j >= mgl) {
                        const int k = fwd ? end_point - 2 - j0 : end_point + 2 + j0;
                        const int which = L.which[pidx];
                        const double raw = (prev - 0.0) + suffix_score;
                        if (which >= 0) last_own = emit(raw, j0, k, which, 0, 0, 3u);
                        if (is_last && trunc) { emit(raw, j0, k, -1, 1, 1, 2u); first_done = true; }
                    }
                }
                if (can_branch) {
                    // Score_Indels at the three positions, in reversed push order: per position insertion, then deletion
                    uint32_t pass = 0;
                    double es6[6];
#pragma unroll
                    for (int pj = 0; pj < 3; pj++) {
                        const int q = (int)((qw >> (8 * pj)) & 255u);
                        const bool low = j0 + pj >= lowest_j && q <= a.indel_q_thr;
                        const double pen = pen_lds ? L.pen[q & 63] : a.pen[q];
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
                    tp++;
                    if (!is_last) nskip = s_run[u0 + 3u * (uint32_t)tp];
                }
            } else if (finishing && do_fin) {
                finishing = false;
                const int m = 3 * tp;
                if (LEVEL == 0) {
                    if (!WRITE) L.agg[oe].m0 = (uint32_t)m << 1 | (trunc ? 1u : 0u);
                    if (a.err_mode == 2) {              // the substitution branch (:1771-1806)
                        const int lo = fwd ? end_point - m : end_point, hi = fwd ? end_point : end_point + m;
                        const int eep = fwd ? lo - 3 : hi + 3;
                        const int anchor = end_point - 1;
                        if (anchor >= 0 && anchor < n && eep >= 0 && eep - 2 < n) {
                            // the two bases behind the region in walk order: steps m and m + 1 from the anchor
                            const uint32_t ba = (uint32_t)((int)L.roff[rl] + anchor);
                            const uint32_t ua = (fwd ? span - 1u - ba : ba) + (uint32_t)m;
                            const uint32_t two = codons(ua);                    // (complemented on the reverse strand: "is it a/t" -> is it a)
                            // forward: bases lo-1 (step m), lo-2 (step m+1), wanted a = 0; reverse: bases hi-1 (step m), hi: wanted t, i.e. a after the complement
                            const int a_first = (two & 3u) == 0u, a_second = ((two >> 2) & 3u) == 0u;
                            const int a1 = a_second, a2 = a_first;          // (the stop codon's second and third base as the strand reads it)
                            double es = suffix_score + a.pass_stop[a1 * 2 + a2];
                            if (m > 0) es += sum - 0.0;
                            c_end = eep; c_score = es; c_sj = suffix_j + m;
                            c_err = (uint32_t)((fwd ? lo - 2 : hi + 2) + 8) << 2 | 2u;
                            c_field = 0;
                            want_push = true;
                        }
                    }
                }
                if (WRITE) { if (!first_done && last_own != MG_NO_SLOT) a.starts[last_own].first = 1; }
                else if (cnt) {
                    MgOrfAgg *g2 = &L.agg[oe];
                    atomicAdd(&g2->cnt, cnt);
                    atomicMax(&g2->best, (unsigned long long)mg_ord(best));
                    const unsigned long long pa = (unsigned long long)(uint32_t)(last_pos + 16) << 32 | (uint32_t)last_j,
                                             pb = (unsigned long long)(uint32_t)(last_pos + 16) << 32 | (0xffffffffu - (uint32_t)last_j);
                    if (fwd) { atomicMin(&g2->ext_a, pa); atomicMin(&g2->ext_b, pb); }
                    else { atomicMax(&g2->ext_a, pa); atomicMax(&g2->ext_b, pb); }
                }
            }
            if (LEVEL < 2) {
                // a call that cannot reach Min_Gene_Len before its read ends emits nothing, nor can a branch of it: not handed on
                if (want_push && c_sj + (fwd ? c_end : n - c_end + 1) + 12 < mgl) want_push = false;
                const uint64_t pm = __ballot(want_push);
                if (pm) {
                    uint32_t b_ = 0;
                    if (lane == 0) b_ = atomicAdd(&L.qn[LEVEL], (uint32_t)__popcll(pm));
                    const uint32_t qb = (uint32_t)__builtin_amdgcn_readfirstlane((int)b_);
                    if (want_push) {
                        const uint32_t slot = qb + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                        if (slot < qcap) {
                            MgCall child;
                            const uint32_t ce0 = LEVEL == 0 ? c_err : e0, ce1 = LEVEL == 1 ? c_err : 0u;
                            child.w[0] = (unsigned long long)__double_as_longlong(c_score);
                            child.w[1] = (key | (uint64_t)c_field << (26 - 13 * LEVEL)) | (uint64_t)(uint32_t)(c_end + 8) << 40 | (uint64_t)(uint32_t)c_sj << 52;
                            child.w[2] = (uint64_t)rl;
                            child.w[3] = (uint64_t)oe | (uint64_t)ce0 << 32 | (uint64_t)ce1 << 46;
                            q_out[slot] = child;
                        } else atomicOr(a.err_flag, 1u);
                    }
                }
            }
        }
    }
}

// mode: 0 count, 1 write.  tile_kept[item]: the count pass marks the (tile, strand) pairs that hold an ORF with starts to write.
template <bool WRITE, bool G32, int CAP>
__global__ __launch_bounds__(ET_BLOCK, 2) void k_mg_err_tile(MgArgs a, const MgTile *tiles, const uint32_t *n_tiles_dev, unsigned long long *item_ctr,
                                                             uint8_t *tile_kept, MgCall *slabs, const uint32_t qcap, const int accepted_only)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t et_lds_raw[];
    EtLds<CAP> &L = *reinterpret_cast<EtLds<CAP> *>(et_lds_raw);
    constexpr int RS = EtLds<CAP>::RS;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid < 64) { L.which[tid] = a.which[tid]; L.pen[tid] = a.err_mode == 1 ? a.pen[tid] : 0.0; }
    const uint64_t n_items = 2ull * (uint64_t)*n_tiles_dev;
    MgCall *q1 = slabs + (size_t)blockIdx.x * 2 * qcap, *q2 = q1 + qcap;
    const int max_level = a.err_mode == 1 ? (a.indel_max < 2 ? a.indel_max : 2) : 1;
    for (;;) {
        __syncthreads();                                // the tile before has left the LDS
        if (tid == 0) L.item = (uint32_t)atomicAdd(item_ctr, 1ull);
        __syncthreads();
        const uint64_t item = L.item;
        if (item >= n_items) break;
        if (WRITE && !tile_kept[item]) continue;
        const bool fwd = (item & 1) == 0;
        const MgTile t = tiles[item >> 1];
        const uint32_t span = t.span, nfit = t.nfit, first = t.first;
        const uint64_t w0 = t.w0;
        const uint64_t o0 = a.read_orf_off[first], o1 = a.read_orf_off[first + nfit];
        if (o0 == o1) continue;
        // ---- the tile's tables
        for (uint32_t i = tid; i <= nfit; i += ET_BLOCK) L.roff[i] = (uint32_t)(a.read_off[first + i] - w0);
        for (uint32_t i = tid; i < nfit; i += ET_BLOCK) L.isl[i] = a.read_isl ? a.read_isl[first + i] : a.ignore_score_len;
        for (uint32_t i = tid; i < (uint32_t)(CAP / 16 + 4); i += ET_BLOCK) {
            // walk indices 16 i .. 16 i + 15: forward strand bases span-1-16i downwards (the window [span-16-16i, span-16i) reversed),
            // reverse strand bases 16 i upwards, complemented.  (Beyond the span: other reads' bases or guard words, never used.)
            uint32_t x;
            if (fwd) x = dev_reverse_fields((uint32_t)dev_window_bits(a.packed, (int64_t)w0 + (int64_t)span - 16 - 16 * (int64_t)i), 16);
            else x = ~(uint32_t)dev_window_bits(a.packed, (int64_t)w0 + 16 * (int64_t)i);
            L.wpk[i] = x;
        }
        {
            // the byte tables in walk order are contiguous in the walk-order tables of the batch: forward strand at total - w0 - span + u
            const uint64_t wb = fwd ? a.total - w0 - span : w0;
            const uint8_t *gq = a.err_mode == 1 ? (fwd ? a.walk_q + wb : a.qual + wb) : nullptr;
            const uint8_t *grq = a.run_q + (fwd ? 0 : a.walk_stride) + wb, *grn = a.run_n + (fwd ? 0 : a.walk_stride) + wb;
            uint8_t *sq = (uint8_t *)L.q, *srq = (uint8_t *)L.rq, *srn = (uint8_t *)L.rn;
            for (uint32_t u = tid; u < span; u += ET_BLOCK) {
                if (gq) { sq[u] = gq[u]; srq[u] = grq[u]; }
                srn[u] = grn[u];
            }
            if (tid < 16) { sq[span + tid] = 255; }    // (a walk reads four qualities at a time)
        }
        // the running sums: one wave per read, 64 walk steps at a time (k_mg_walk_prefix)
        for (uint32_t rl = wave; rl < nfit; rl += ET_BLOCK / 64) {
            const uint64_t r = (uint64_t)first + rl;
            const uint32_t rs = (uint32_t)(a.read_off[r] - w0), n = (uint32_t)(a.read_off[r + 1] - a.read_off[r]);
            const float *nt = G32 ? a.null_tab + (size_t)(a.read_null ? a.read_null[r] : 0u) * MG_NULL_FLOATS : nullptr;
            const uint32_t ub = fwd ? span - rs - n : rs;                     // walk index of the read's first step
            double carry[3] = {0.0, 0.0, 0.0};
            for (uint32_t t0 = 0; t0 < n; t0 += 64) {
                const uint32_t tt = t0 + lane;
                const bool in = tt < n;
                const uint32_t si = in ? (fwd ? n - 1 - tt : tt) : 0u;          // base of walk step tt inside the read
                const uint64_t g = w0 + rs + si;
                double v[3];
                if (G32) {
                    const uint32_t five = (uint32_t)dev_window_bits(a.packed, (int64_t)g - 2) & 0x3ffu, c0 = (five >> 4) & 3u;
#pragma unroll
                    for (int f = 0; f < 3; f++) {
                        const float nv = fwd ? mg_null_value<true>(nt, f, (int)si, (int)n, c0, (five >> 6) & 3u, (five >> 8) & 3u)
                                             : mg_null_value<false>(nt, f, (int)si, (int)n, c0, (five >> 2) & 3u, five & 3u);
                        v[f] = in ? (double)a.gene32[(uint64_t)((fwd ? 0 : 3) + f) * a.fs_stride + g] - (double)nv : 0.0;
                    }
                } else {
#pragma unroll
                    for (int f = 0; f < 3; f++) v[f] = in ? a.fs[(uint64_t)((fwd ? 0 : 3) + f) * a.fs_stride + g] : 0.0;
                }
                const uint32_t u = ub + tt, m = u % 3u;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const uint32_t row = ((m + 3u - (uint32_t)c) % 3u + 1u) % 3u;
                    const double x = row == 0 ? v[0] : row == 1 ? v[1] : v[2];
                    const double sc = mg_wave_scan(x) + carry[c];
                    if (in) L.S[c][4 + u] = sc;
                    const unsigned long long top = (unsigned long long)__double_as_longlong(sc);
                    carry[c] = __longlong_as_double((long long)((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(top >> 32), 63) << 32 |
                                                                   (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)top, 63)));
                }
            }
        }
        // ---- the tile's ORFs, ET_MAXO records at a time
        bool any_kept = false;
        for (uint64_t eb = o0; eb < o1; eb += ET_MAXO) {
            __syncthreads();                            // tables complete / the batch before is done
            if (tid == 0) { L.n_o = 0; L.qn[0] = L.qn[1] = 0; L.take[0] = L.take[1] = L.take[2] = 0; L.kept = 0; }
            __syncthreads();
            const uint32_t nb = o1 - eb < (uint64_t)ET_MAXO ? (uint32_t)(o1 - eb) : (uint32_t)ET_MAXO;
            if (tid < nb) {
                const gmg_mg_orf *o = a.orfs + eb + tid;
                const int frame = o->frame;
                bool take = (frame > 0) == fwd;
                if (WRITE && take) take = a.orf_cnt[eb + tid] != 0;
                if (take) {
                    const uint32_t k = atomicAdd(&L.n_o, 1u);
                    L.oinf[k] = tid | (o->read - first) << 16;
                    L.oep[k] = frame > 0 ? o->stop_position - 1 : o->stop_position + 3;
                    if (WRITE) L.fill[k] = 0;
                    else {
                        MgOrfAgg g0;
                        g0.best = mg_ord(-DBL_MAX); g0.ext_a = g0.ext_b = fwd ? ~0ull : 0ull; g0.cnt = 0; g0.m0 = 0;
                        L.agg[k] = g0;
                    }
                }
            }
            __syncthreads();
            const uint32_t n_o = L.n_o;
            if (n_o == 0) continue;
            et_level<WRITE, 0, CAP>(a, L, fwd, span, eb, n_o, nullptr, q1, qcap, accepted_only);
            if (max_level >= 1) {
                __syncthreads();
                const uint32_t n1 = L.qn[0] < qcap ? L.qn[0] : qcap;
                // (the queue was written by other waves of this work-group: global memory, made visible by the barrier's release / acquire)
                __threadfence_block();
                if (n1) et_level<WRITE, 1, CAP>(a, L, fwd, span, eb, n1, q1, q2, qcap, accepted_only);
                if (max_level >= 2 && a.err_mode == 1) {
                    __syncthreads();
                    const uint32_t n2 = L.qn[1] < qcap ? L.qn[1] : qcap;
                    __threadfence_block();
                    if (n2) et_level<WRITE, 2, CAP>(a, L, fwd, span, eb, n2, q2, nullptr, qcap, accepted_only);
                }
            }
            __syncthreads();
            if (!WRITE && tid < n_o) {
                // Score_Orfs_Errors' verdict (:1647-1683) from what the ORF's calls added up to (k_mg_err_verdict)
                const MgOrfAgg g = L.agg[tid];
                const uint64_t i = eb + (L.oinf[tid] & 0xffffu);
                bool accepted = false;
                int acc = 0, jmin = 0;
                double best_score = -DBL_MAX;
                if (g.cnt) {
                    const uint32_t ja = (uint32_t)g.ext_a, jb = 0xffffffffu - (uint32_t)g.ext_b;
                    jmin = (int)(fwd ? ja : jb);
                    const int jmax = (int)(fwd ? jb : ja);
                    if (jmax + 1 >= a.min_gene_len) {
                        best_score = mg_unord(g.best);
                        if (best_score > a.start_threshold) { acc = jmin + 1 >= a.min_gene_len ? 1 : 2; accepted = true; }
                    }
                }
                const uint32_t n_keep = (accepted_only && !accepted) ? 0u : g.cnt;
                a.orf_cnt[i] = n_keep;
                if (n_keep) L.kept = 1;
                if (accepted) atomicOr(&a.acc_bits[i >> 5], 1u << (i & 31u));
                if (!accepted_only || accepted) {
                    gmg_mg_orf rec = a.orfs[i];
                    const int m0 = (int)(g.m0 >> 1);
                    if (fwd) { rec.hi = rec.stop_position - 1; rec.lo = rec.hi - m0; }
                    else { rec.lo = rec.stop_position + 3; rec.hi = rec.lo + m0; }
                    rec.orf_is_truncated = (int16_t)(g.m0 & 1);
                    rec.n_starts = g.cnt;
                    rec.first_j = g.cnt ? jmin : 0;
                    rec.best_score = best_score;
                    rec.accepted = (int16_t)acc;
                    rec.start_begin = 0;
                    a.orfs[i] = rec;
                }
            }
            if (!WRITE) {
                __syncthreads();
                any_kept = any_kept || L.kept != 0;
            }
        }
        if (!WRITE && tid == 0) tile_kept[item] = any_kept ? 1 : 0;
    }
}
