// gmg_mg_errtile.h -- glimmer-mg's error branch (-i / -s: Score_Indels and the recursive Score_Orf_Starts,
// src/Glimmer/glimmer-mg.cc:1513-1602,1693-1861) with everything a call reads held in LDS, one lane per EVENT.  Part of
// gmg_mg.hip (included there, behind MgArgs and the level kernels whose walk it restates).
//
// The level kernels (k_mg_err_level) take score[j] of a call as a difference of two running sums and visit only the codons at
// which something can happen -- but the sums lived in HBM: 48 B/base written once (19 GB per 1M reads of ~400 bp) and then one
// 64-byte sector fetched per event for 8 - 32 useful bytes, the calls handed from level to level through HBM as well, and a lane
// walked its call from event to event while the lanes with shorter calls waited.  Two observations:
//   * a read's calls never leave the read, a strand's calls never leave the strand: a work-group takes a TILE -- a few consecutive
//     whole reads, <= CAP bases, one strand -- and builds in LDS what its calls will read: S[3][CAP], the inclusive running sums of
//     the three reading-frame classes along the strand's walk order (24 B/base, from the fp32 gene rows and the read's null model, or
//     from the caller's fp64 table), the bases in walk order, the qualities;
//   * with running sums nothing is carried from one codon of a call to the next: score[j] = S[c][u0 + j] - S[c][u0 - 1] whatever
//     was visited before.  The codons at which something can happen -- a start codon, a base of low quality, the end of the region --
//     are the same for every call of a class: ONE sorted event list per class and tile (positions + flags, the next terminator and
//     the next start codon of every event).  A call is then (first event, number of events), its events are independent work
//     items, and a level of the call tree is: scan of the calls' event counts, one lane per (call, event) pair, children appended to
//     the next level's list.  No lane walks, no lane waits for a longer call.
// What an ORF's calls add up to is merged with LDS atomics and judged by the same work-group (Score_Orfs_Errors' verdict,
// :1647-1683); every start is also written, as it is found, to a slab of the work-group (global memory, read back at once: L2), and
// the starts of the ORFs that are kept move from there into the batch's staging arrays -- the slice of an ORF is reserved with one
// atomic per batch of ORFs.  After the kernel a scan of the counts gives the final slices and k_et_unstage moves the staged ones
// there; the segmented sort of mg_run puts every slice into push order by its keys, as for the level kernels.  ONE pass over the
// reads: the tables are built once.
//
// Coordinates inside a tile: b = base - w0; walk index u = span - 1 - b on the forward strand (its walks run down the read), u = b on
// the reverse strand: every call goes up in u.  A call anchored at u0 belongs to class c = u0 % 3 and reads row c; at step u the row
// of class c adds Frame_Scores row ((u - c) % 3 + 1) % 3 of the strand (glimmer-mg.cc:561-604: f = 1, 2, 0, ...).
// Events of position u (class u % 3), the codon = walk steps u, u + 1, u + 2:
//     END      the codon does not fit into the read any more                      (terminator; the region ends in front of it)
//     STOP     it is a stop codon                                                  (terminator; the region ends in front of it)
//     LASTFIT  it is the last codon of its class that fits into the read          (terminator; the region ends BEHIND it)
//     START    Codon_t::Can_Be says start;   LOWQ   one of its bases has quality <= Indel_Quality_Threshold (-i only)
// A call's region = the codons from u0 to its first terminator: m = 3 * codons; the terminator's item does what the walk did when
// it left the region (truncated start :1818-1846, level 0: the region's length and the substitution branch :1771-1806).
// Reads longer than CAP keep read_fit = 0 and go to k_mg_err_flat as before; a full slab raises err_flag bit 0 and the whole batch
// repeats on the level kernels; full staging arrays raise bit 1 and the kernel repeats with arrays of the size it asked for.

// Shapes measured on 1M reads of ~400 bp, -i / -s (profiles/r04_errtile_shapes.txt): 1024 bases x 3 work-groups per CU 68.1 / 41.3 ms,
// 1280 x 2: 57.4 / 32.8, 1536 x 2: 54.1 / 29.0 (the larger the tile the better while two work-groups fit: what an item costs
// is mostly independent of its size -- profiles/r04_errtile_stamps.txt), 2048 x 1 (512 lanes): 68.7 / 40.4, (1024 lanes): 78.3 / 47.8.
#ifndef MG_ET_CAP
#define MG_ET_CAP 1536           // bases per tile (79 KB of LDS: two work-groups per CU); also the longest read a tile takes
#endif
#ifndef ET_BLOCK
#define ET_BLOCK 256
#endif
#ifndef ET_WG_PER_CU
#define ET_WG_PER_CU 2           // work-groups per CU the LDS allows
#endif
#ifndef ET_WAVES_PER_SIMD
#define ET_WAVES_PER_SIMD 2      // (eight waves per CU: the register file is no limit)
#endif
#define ET_MAXR 64               // reads per tile
#ifndef ET_MAXO
#define ET_MAXO 192              // ORF records (both strands) staged per batch of a tile (256 would leave room for one work-group per CU)
#endif
#ifndef ET_CHUNK
#define ET_CHUNK 512             // calls whose event counts are scanned at a time
#endif
#ifndef ET_QCAP
#define ET_QCAP 8192             // calls per level a work-group's slab holds
#endif
#ifndef ET_ECAP
#define ET_ECAP 16384            // starts per batch of ORFs a work-group's slab holds
#endif
static_assert(ET_MAXO <= ET_BLOCK && ET_MAXO <= 256, "one lane per staged ORF record, its index in eight bits");
static_assert(ET_CHUNK <= 4 * ET_BLOCK, "four calls per lane in the scan");
static_assert(MG_ET_CAP <= 2048, "event entries hold the position in 11 bits");
#define MG_ET_AUTO_BASES_INDEL 90000000ull   // mg_err_tile = -1: batches up to this many bases take the tile kernel (-i) ...
#define MG_ET_AUTO_BASES_SUB 60000000ull     // ... (-s)
#define ET_QNI 512                // calls per queue whose event count is also kept in LDS
#define ET_CHUNK_TILES 16        // tile builder: one lane lays out the tiles of 16 x CAP bases

#ifndef GMG_ET_STAMPS
#define GMG_ET_STAMPS 0          // diagnostic build: cycles per stage of k_mg_err_tile, wave 0 of every work-group (tools/et_stamps.py); not in the product
#endif
#if GMG_ET_STAMPS
__device__ unsigned long long g_et_stamps[16];
extern "C" int gmg_debug_et_stamps(unsigned long long *out, int reset)
{
    if (reset) { unsigned long long z[16] = {0}; return hipMemcpyToSymbol(HIP_SYMBOL(g_et_stamps), z, sizeof z) == hipSuccess ? 0 : -1; }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_et_stamps), 128) == hipSuccess ? 0 : -1;
}
#define ET_STAMP(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); et_acc[i] += now_ - et_prev; et_prev = now_; } while (0)
#else
#define ET_STAMP(i) do { } while (0)
#endif

#define EV_STOP 1u
#define EV_END 2u
#define EV_LASTFIT 4u
#define EV_START 8u
#define EV_LOWQ 16u
#define EV_TERM (EV_STOP | EV_END | EV_LASTFIT)

// a start as it is found: record, error list, order key, ORF of the batch
struct __attribute__((aligned(8))) EtEm { gmg_start s; gmg_start_errors e; uint32_t oe; uint64_t key; };
static_assert(sizeof(EtEm) == 48, "EtEm layout");

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
    static constexpr int RS = CAP + 8;                  // row stride: 4 spare entries on both sides (an event looks one back and three ahead)
    static constexpr int EC = CAP / 3 + 4;              // events per class
    double S[3][RS];
    MgOrfAgg agg[ET_MAXO];
    double pen[64];
    unsigned long long stage_base;
    uint32_t wpk[CAP / 16 + 4];                         // 2-bit codes in walk order (complemented on the reverse strand), 16 per word
    uint32_t q[(CAP + 16) / 4];                         // qualities in walk order, 4 per word
    uint32_t scan[ET_CHUNK + 4];                        // exclusive sums of the event counts of a chunk of calls
    uint32_t roff[ET_MAXR + 1];
    int32_t isl[ET_MAXR];
    uint32_t o_inf[ET_MAXO];                            // level-0 call: record in the batch (8) | read in the tile << 8 (7) | p0 is zero << 15 | u0 << 16
    uint32_t o_ev[ET_MAXO];                             // ... its first event | events << 16
    int32_t o_ep[ET_MAXO];                              // ... its end_point
    uint32_t keep_cnt[ET_MAXO], keep_off[ET_MAXO];      // starts of the ORF that leave the tile, their place behind the batch's base
    uint32_t fill[ET_MAXO];                             // slots handed out inside the ORF's slice
    uint32_t wsum[ET_BLOCK / 64 + 1];
    uint32_t n_o, n_act, qn[2], n_em, stage_ok;
    // the item being worked on is read from nxt_* at the top of the loop; thread 0 refills it while the item is worked on
    unsigned long long nxt_item, nxt_o[2], wmask[ET_BLOCK / 64];
    MgTile nxt_tile;
    uint16_t o_act[ET_MAXO];                            // the staged ORFs that have events (the level-0 calls)
    uint16_t qni[2][ET_QNI];                            // event counts of the first calls of the two queues (the scan reads them here, not in the slab)
    // two event lists per class: [0] every event (the levels from which Score_Indels may branch), [1] without the events that
    // are LOWQ only (the last level: two events in three are of that kind, and there nothing happens at them)
    uint16_t fe[2][CAP + 8];                            // events of the class of u at positions below u = index of the first event at or behind u
    uint16_t ev[2][3][EC];                              // position | flags << 11
    uint16_t nt[2][3][EC];                              // the first terminator at or behind the event
    uint16_t ns[2][3][EC];                              // the first START event behind the event (n_ev: none)
    uint8_t fl[CAP + 8];                                // the flags of every position
    int8_t which[64];
};

static_assert(sizeof(EtLds<MG_ET_CAP>) * ET_WG_PER_CU <= 160 * 1024, "ET_WG_PER_CU work-groups must fit the CU's LDS");

// a call as the items see it
struct EtCall {
    double ss;                   // suffix_score
    uint64_t key;
    int ep, sj;                  // end_point, suffix_j
    uint32_t rl, oe, e0, e1, u0, i0, ni;
    bool p0z;
};

template <bool G32, int CAP>
__global__ __launch_bounds__(ET_BLOCK, ET_WAVES_PER_SIMD) void k_mg_err_tile(MgArgs a, const MgTile *tiles, const uint32_t *n_tiles_dev, unsigned long long *item_ctr,
                                                             MgCall *slabs, const uint32_t qcap, EtEm *em_slabs, const uint32_t ecap,
                                                             unsigned long long *stage_ctr, const unsigned long long stage_cap, const int accepted_only)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t et_lds_raw[];
    EtLds<CAP> &L = *reinterpret_cast<EtLds<CAP> *>(et_lds_raw);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid < 64) { L.which[tid] = a.which[tid]; L.pen[tid] = a.err_mode == 1 ? a.pen[tid] : 0.0; }
    const uint64_t n_items = 2ull * (uint64_t)*n_tiles_dev;
    MgCall *q1 = slabs + (size_t)blockIdx.x * 2 * qcap, *q2 = q1 + qcap;
    EtEm *em = em_slabs + (size_t)blockIdx.x * ecap;
    const int max_level = a.err_mode == 1 ? (a.indel_max < 2 ? a.indel_max : 2) : 1;
    const int mgl = a.min_gene_len;
    const int lowest_j = mgl - 3 < 3 ? mgl - 3 : 3;
    const bool pen_lds = a.indel_q_thr < 64;
    uint8_t *const s_q = (uint8_t *)L.q;
    // the event list of a level: the one without the LOWQ-only events where Score_Indels cannot branch (-i: the last level)
    const bool two_lists = a.err_mode == 1;
    auto list_of = [&](int level) __attribute__((always_inline)) -> int {
        return (two_lists && !(level < 2 && level < a.indel_max)) ? 1 : 0;
    };

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
    // exclusive sum over the work-group of one value per lane; returns the lane's offset, total = the sum (barriers inside)
    auto block_scan = [&](uint32_t v, uint32_t &total) __attribute__((always_inline)) -> uint32_t {
        uint32_t x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = (uint32_t)__shfl_up((int)x, d, 64);
            if ((int)lane >= d) x += y;
        }
        if (lane == 63) L.wsum[wave] = x;
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (uint32_t w = 0; w < ET_BLOCK / 64; w++) { const uint32_t s = L.wsum[w]; if (w < wave) before += s; all += s; }
        __syncthreads();
        total = all;
        return before + x - v;
    };

    // What the next item needs first -- its tile, its ORF range, the read offsets, the bases, the qualities -- is loaded while the
    // current item is worked on (a chain of three dependent loads and the tables' own: five memory latencies of the ~25 an item
    // had), and the lines of its gene rows are touched so that its stage B finds them in the L2.
    constexpr int PQ = (CAP + ET_BLOCK - 1) / ET_BLOCK;
    unsigned long long p_roff = 0;
    int32_t p_isl = 0;
    uint32_t p_w[3] = {0, 0, 0};
    uint8_t p_q[PQ];
    uint32_t p_touch[2] = {0, 0};
#pragma unroll
    for (int x = 0; x < PQ; x++) p_q[x] = 0;
    auto prefetch = [&]() __attribute__((always_inline)) {     // (all lanes) the registers above for the item in L.nxt_*
        const uint64_t item = L.nxt_item;
        if (item >= n_items) return;
        const MgTile t = L.nxt_tile;
        const bool fwd = (item & 1) == 0;
        if (tid <= t.nfit) p_roff = a.read_off[t.first + tid];
        if (tid < t.nfit) p_isl = a.read_isl ? a.read_isl[t.first + tid] : a.ignore_score_len;
        if (tid < (uint32_t)(CAP / 16 + 4)) {
            // walk indices 16 i .. 16 i + 15: forward strand bases span-1-16i downwards (the window [span-16-16i, span-16i) reversed),
            // reverse strand bases 16 i upwards, complemented.  (Beyond the span: other reads' bases or guard words, never used.)
            const int64_t fb = fwd ? (int64_t)t.w0 + (int64_t)t.span - 16 - 16 * (int64_t)tid : (int64_t)t.w0 + 16 * (int64_t)tid;
            const int64_t wi = fb >> 4;
            p_w[0] = a.packed[wi]; p_w[1] = a.packed[wi + 1]; p_w[2] = a.packed[wi + 2];
        }
        if (a.err_mode == 1) {
            // the qualities in walk order are contiguous in the batch's walk-order copy: forward strand at total - w0 - span + u
            const uint64_t wb = fwd ? a.total - t.w0 - t.span : t.w0;
            const uint8_t *gq = fwd ? a.walk_q + wb : a.qual + wb;
#pragma unroll
            for (int x = 0; x < PQ; x++) { const uint32_t u = tid + (uint32_t)ET_BLOCK * x; if (u < t.span) p_q[x] = gq[u]; }
        }
        if (wave >= 1) {                                // waves 1 - 3: one row each, one lane per 128-byte line
            const uint32_t per = G32 ? 32u : 16u;
            const uint32_t nl = t.span / per + 2u;
            const uint64_t lim = (uint64_t)6 * a.fs_stride - 1;
#pragma unroll
            for (int x = 0; x < 2; x++) {
                const uint32_t line = lane + 64u * (uint32_t)x;
                if (line < nl) {
                    const uint64_t e = (uint64_t)((fwd ? 0 : 3) + (wave - 1u)) * a.fs_stride + t.w0 + (uint64_t)line * per;
                    const uint64_t ee = e < lim ? e : lim;
                    p_touch[x] = G32 ? *(const volatile uint32_t *)(a.gene32 + ee) : *(const volatile uint32_t *)(a.fs + ee);
                }
            }
        }
    };
    if (tid == 0) {
        const unsigned long long it = atomicAdd(item_ctr, 1ull);
        L.nxt_item = it;
        if (it < n_items) {
            const MgTile t = tiles[it >> 1];
            L.nxt_tile = t;
            L.nxt_o[0] = a.read_orf_off[t.first]; L.nxt_o[1] = a.read_orf_off[t.first + t.nfit];
        }
    }
    __syncthreads();
    prefetch();
#if GMG_ET_STAMPS
    unsigned long long et_acc[16] = {0}, et_prev = __builtin_readcyclecounter();
#endif
    for (;;) {
        const uint64_t item = L.nxt_item;
        const MgTile t = L.nxt_tile;
        const uint64_t o0 = L.nxt_o[0], o1 = L.nxt_o[1];
        __syncthreads();                                // everyone has the item; the tile before has left the LDS
        if (item >= n_items) break;
        const bool fwd = (item & 1) == 0;
        const uint32_t span = t.span, nfit = t.nfit, first = t.first;
        const uint64_t w0 = t.w0;
        unsigned long long nx = 0;
        if (tid == 0) nx = atomicAdd(item_ctr, 1ull);   // (used behind stage B)
        // ---- stage A: offsets, bases and qualities in walk order, from the registers the item before has filled
        if (tid <= nfit) L.roff[tid] = (uint32_t)(p_roff - w0);
        if (tid < nfit) L.isl[tid] = p_isl;
        if (tid < (uint32_t)(CAP / 16 + 4)) {
            const int64_t fb = fwd ? (int64_t)w0 + (int64_t)span - 16 - 16 * (int64_t)tid : (int64_t)w0 + 16 * (int64_t)tid;
            const unsigned sh = 2u * (unsigned)(fb & 15);
            const uint64_t lo = (uint64_t)p_w[0] | ((uint64_t)p_w[1] << 32);
            const uint32_t win = (uint32_t)((lo >> sh) | (((uint64_t)p_w[2] << 1) << (63 - sh)));   // (dev_window_bits)
            L.wpk[tid] = fwd ? dev_reverse_fields(win, 16) : ~win;
        }
        if (a.err_mode == 1) {
#pragma unroll
            for (int x = 0; x < PQ; x++) { const uint32_t u = tid + (uint32_t)ET_BLOCK * x; if (u < span) s_q[u] = p_q[x]; }
            if (tid < 16) s_q[span + tid] = 255;        // (an event reads four qualities at a time)
        }
        __syncthreads();
        ET_STAMP(0);                                    // top of the item + stage A
        // ---- stage B: one wave per read, 64 walk steps at a time: the running sums (k_mg_walk_prefix) and the flags of every position.
        //      Every load of the read is issued before the first value is used (a trip's sums need the trip before: one memory
        //      latency per read instead of one per 64 steps).
        for (uint32_t rl = wave; rl < nfit; rl += ET_BLOCK / 64) {
            const uint64_t r = (uint64_t)first + rl;
            const uint32_t rs = L.roff[rl], n = L.roff[rl + 1] - rs;
            const float *nt = G32 ? a.null_tab + (size_t)(a.read_null ? a.read_null[r] : 0u) * MG_NULL_FLOATS : nullptr;
            const uint32_t ub = fwd ? span - rs - n : rs;                     // walk index of the read's first step
            constexpr int TG = 4;                       // trips whose loads are in flight together
            double carry[3] = {0.0, 0.0, 0.0};
            for (uint32_t t0 = 0; t0 < n; t0 += 64 * TG) {
                typename std::conditional<G32, float, double>::type gv[TG][3];
                float nv[G32 ? TG : 1][3];
#pragma unroll
                for (int k = 0; k < TG; k++) {
                    const uint32_t tt = t0 + 64u * (uint32_t)k + lane;
                    const bool in = tt < n;
                    const uint32_t si = in ? (fwd ? n - 1 - tt : tt) : 0u;      // base of walk step tt inside the read
                    const uint64_t g = w0 + rs + si;
#pragma unroll
                    for (int f = 0; f < 3; f++) {
                        if (G32) gv[k][f] = a.gene32[(uint64_t)((fwd ? 0 : 3) + f) * a.fs_stride + g];
                        else gv[k][f] = a.fs[(uint64_t)((fwd ? 0 : 3) + f) * a.fs_stride + g];
                    }
                    if (G32) {
                        const uint32_t five = (uint32_t)dev_window_bits(a.packed, (int64_t)g - 2) & 0x3ffu, c0 = (five >> 4) & 3u;
#pragma unroll
                        for (int f = 0; f < 3; f++)
                            nv[k][f] = fwd ? mg_null_value<true>(nt, f, (int)si, (int)n, c0, (five >> 6) & 3u, (five >> 8) & 3u)
                                           : mg_null_value<false>(nt, f, (int)si, (int)n, c0, (five >> 2) & 3u, five & 3u);
                    }
                }
#pragma unroll
                for (int k = 0; k < TG; k++) {
                    const uint32_t tt = t0 + 64u * (uint32_t)k + lane;
                    if (t0 + 64u * (uint32_t)k >= n) break;
                    const bool in = tt < n;
                    double v[3];
#pragma unroll
                    for (int f = 0; f < 3; f++) v[f] = in ? (G32 ? (double)gv[k][f] - (double)nv[G32 ? k : 0][f] : (double)gv[k][f]) : 0.0;
                    const uint32_t u = ub + tt, m3 = u % 3u;
                    if (in) {                           // the flags of position u
                        uint32_t f = 0;
                        if (tt + 2 >= n) f = EV_END;
                        else {
                            const uint32_t idx = codons(u) & 63u;
                            if ((a.fwd_stop >> idx) & 1ull) f = EV_STOP;
                            else {
                                if (tt + 5 >= n) f |= EV_LASTFIT;
                                if (L.which[idx] >= 0) f |= EV_START;
                                if (a.err_mode == 1) {
                                    const uint32_t qw = quals4(u);
                                    const uint32_t thr = (uint32_t)a.indel_q_thr;
                                    if ((qw & 255u) <= thr || ((qw >> 8) & 255u) <= thr || ((qw >> 16) & 255u) <= thr) f |= EV_LOWQ;
                                }
                            }
                        }
                        L.fl[u] = (uint8_t)f;
                    }
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const uint32_t row = ((m3 + 3u - (uint32_t)c) % 3u + 1u) % 3u;
                        const double x = row == 0 ? v[0] : row == 1 ? v[1] : v[2];
                        const double sc = mg_wave_scan(x) + carry[c];
                        if (in) L.S[c][4 + u] = sc;
                        const unsigned long long top = (unsigned long long)__double_as_longlong(sc);
                        carry[c] = __longlong_as_double((long long)((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(top >> 32), 63) << 32 |
                                                                       (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)top, 63)));
                    }
                }
            }
        }
        ET_STAMP(1);                                    // stage B, own work
        __syncthreads();
        ET_STAMP(2);                                    // ... waiting for the other waves
        MgTile tn;
        tn.w0 = 0; tn.first = 0; tn.nfit = 0; tn.span = 0; tn.pad = 0;
        if (tid == 0 && nx < n_items) tn = tiles[nx >> 1];     // (used behind stage C)
        // ---- stage C: the event lists of class c (wave c; both lists): positions c, c + 3, ... in order; then, from the last event to
        //      the first, every event's next terminator and next start codon
        if (wave < 3) {
            const uint32_t c = wave;
            for (int li = 0; li < (two_lists ? 2 : 1); li++) {
                const uint32_t keep = li == 0 ? ~0u : ~EV_LOWQ;
                uint32_t n_e = 0;
                for (uint32_t pb = c; pb < span; pb += 192) {
                    const uint32_t u = pb + 3u * lane;
                    const uint32_t f = u < span ? (uint32_t)L.fl[u] : 0u;
                    const uint64_t em_ = __ballot((f & keep) != 0);
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(em_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)em_, 0u));
                    if (u < span) L.fe[li][u] = (uint16_t)(n_e + rank);
                    if (f & keep) L.ev[li][c][n_e + rank] = (uint16_t)(u | f << 11);
                    n_e += (uint32_t)__popcll(em_);
                }
                uint32_t carry_t = n_e, carry_s = n_e;
                for (int eb_ = (int)((n_e + 63u) / 64u) * 64 - 64; eb_ >= 0; eb_ -= 64) {
                    const uint32_t e = (uint32_t)eb_ + lane;
                    const uint32_t f = e < n_e ? (uint32_t)L.ev[li][c][e] >> 11 : 0u;      // (written by this wave)
                    const uint64_t tm_ = __ballot((f & EV_TERM) != 0), sm_ = __ballot((f & EV_START) != 0);
                    const uint64_t t_here = tm_ >> lane, s_after = lane == 63 ? 0ull : sm_ >> (lane + 1);
                    const uint32_t nt_ = t_here ? e + (uint32_t)__builtin_ctzll(t_here) : carry_t;
                    const uint32_t ns_ = s_after ? e + 1u + (uint32_t)__builtin_ctzll(s_after) : carry_s;
                    if (e < n_e) { L.nt[li][c][e] = (uint16_t)nt_; L.ns[li][c][e] = (uint16_t)ns_; }
                    if (tm_) carry_t = (uint32_t)eb_ + (uint32_t)__builtin_ctzll(tm_);
                    if (sm_) carry_s = (uint32_t)eb_ + (uint32_t)__builtin_ctzll(sm_);
                }
            }
        }
        unsigned long long o0n = 0, o1n = 0;
        if (tid == 0) {
            L.nxt_item = nx;
            if (nx < n_items) { L.nxt_tile = tn; o0n = a.read_orf_off[tn.first]; o1n = a.read_orf_off[tn.first + tn.nfit]; }   // (used at the item's end)
        }
        ET_STAMP(3);                                    // stage C
        __syncthreads();
        prefetch();
        ET_STAMP(4);                                    // barrier + the next item's loads issued
        // a call's geometry from its end_point: anchor inside the read?  u0, first event, events up to and with the terminator
        auto call_geo = [&](int ep, uint32_t rl, int li, uint32_t &u0, uint32_t &i0, uint32_t &ni, bool &p0z) __attribute__((always_inline)) -> bool {
            const int rs = (int)L.roff[rl], n = (int)L.roff[rl + 1] - rs;
            const int anchor = ep - 1;
            if (anchor < 0 || anchor >= n) return false;
            const uint32_t ba = (uint32_t)(rs + anchor);
            u0 = fwd ? span - 1u - ba : ba;
            i0 = L.fe[li][u0];
            ni = (uint32_t)L.nt[li][u0 % 3u][i0] - i0 + 1u;
            p0z = fwd ? anchor == n - 1 : anchor == 0;  // (the sums restart with every read)
            return true;
        };
        // ---- the tile's ORFs, ET_MAXO records at a time
        for (uint64_t eb = o0; eb < o1; eb += ET_MAXO) {
            __syncthreads();                            // tables complete / the batch before is done
            if (tid == 0) { L.n_o = 0; L.n_act = 0; L.qn[0] = L.qn[1] = 0; L.n_em = 0; L.stage_ok = 1; }
            __syncthreads();
            const uint32_t nb = o1 - eb < (uint64_t)ET_MAXO ? (uint32_t)(o1 - eb) : (uint32_t)ET_MAXO;
            if (tid < nb) {
                const gmg_mg_orf *o = a.orfs + eb + tid;
                const int frame = o->frame;
                if ((frame > 0) == fwd) {
                    const uint32_t k = atomicAdd(&L.n_o, 1u);
                    const uint32_t rl = o->read - first;
                    const int ep = frame > 0 ? o->stop_position - 1 : o->stop_position + 3;
                    const int n = (int)(L.roff[rl + 1] - L.roff[rl]);
                    uint32_t u0 = 0, i0 = 0, ni = 0;
                    bool p0z = false;
                    bool ok = call_geo(ep, rl, list_of(0), u0, i0, ni, p0z);
                    // accepted ORFs only: an ORF whose end lies too close to the read's upstream end cannot reach Min_Gene_Len on any path
                    if (accepted_only && (fwd ? ep : n - ep + 1) + 12 < mgl) ok = false;
                    if (!ok) ni = 0;
                    else L.o_act[atomicAdd(&L.n_act, 1u)] = (uint16_t)k;
                    L.o_inf[k] = tid | rl << 8 | (p0z ? 1u << 15 : 0u) | u0 << 16;
                    L.o_ev[k] = i0 | ni << 16;
                    L.o_ep[k] = ep;
                    L.fill[k] = 0;
                    MgOrfAgg g0;
                    g0.best = mg_ord(-DBL_MAX); g0.ext_a = g0.ext_b = fwd ? ~0ull : 0ull; g0.cnt = 0; g0.m0 = 0;
                    L.agg[k] = g0;
                }
            }
            __syncthreads();
            ET_STAMP(5);                                // the batch's ORF records staged
            const uint32_t n_o = L.n_o;
            if (n_o == 0) continue;
            for (int level = 0; level <= max_level; level++) {
                const uint32_t n_calls = level == 0 ? L.n_act : (L.qn[level - 1] < qcap ? L.qn[level - 1] : qcap);
                const MgCall *q_in = level == 1 ? q1 : q2;
                MgCall *q_out = level == 0 ? q1 : q2;
                const bool can_branch = level < 2 && a.err_mode == 1 && level < a.indel_max;
                const int li = list_of(level);
                const int key_shift = 26 - 13 * level;
                auto fetch = [&](uint32_t k) __attribute__((always_inline)) -> EtCall {
                    EtCall c;
                    if (level == 0) {
                        const uint32_t ko = L.o_act[k];
                        const uint32_t inf = L.o_inf[ko], evw = L.o_ev[ko];
                        c.ss = 0.0; c.key = 0; c.ep = L.o_ep[ko]; c.sj = 0; c.rl = (inf >> 8) & 127u; c.oe = ko; c.e0 = c.e1 = 0;
                        c.u0 = inf >> 16; c.i0 = evw & 0xffffu; c.ni = evw >> 16; c.p0z = (inf >> 15) & 1u;
                    } else {
                        const MgCall r = q_in[k];
                        c.ss = __longlong_as_double((long long)r.w[0]);
                        c.key = r.w[1] & 0xffffffffffull; c.ep = (int)((r.w[1] >> 40) & 0xfffu) - 8; c.sj = (int)(r.w[1] >> 52);
                        c.rl = (uint32_t)r.w[2] & 0xffu; c.u0 = (uint32_t)(r.w[2] >> 8) & 0xfffu; c.i0 = (uint32_t)(r.w[2] >> 20) & 0xfffu;
                        c.ni = (uint32_t)(r.w[2] >> 32) & 0xfffu; c.p0z = (r.w[2] >> 44) & 1u;
                        c.oe = (uint32_t)r.w[3]; c.e0 = (uint32_t)(r.w[3] >> 32) & 0x3fffu; c.e1 = (uint32_t)(r.w[3] >> 46) & 0x3fffu;
                    }
                    return c;
                };
                for (uint32_t cb = 0; cb < n_calls; cb += ET_CHUNK) {
                    const uint32_t nc = n_calls - cb < (uint32_t)ET_CHUNK ? n_calls - cb : (uint32_t)ET_CHUNK;
                    // exclusive sums of the calls' event counts: four consecutive calls per lane
                    uint32_t v[4], tot4 = 0;
#pragma unroll
                    for (int x = 0; x < 4; x++) {
                        const uint32_t k = 4u * tid + (uint32_t)x;
                        v[x] = 0;
                        if (k < nc) v[x] = level == 0 ? L.o_ev[L.o_act[cb + k]] >> 16 : cb + k < (uint32_t)ET_QNI ? (uint32_t)L.qni[level - 1][cb + k] : (uint32_t)(q_in[cb + k].w[2] >> 32) & 0xfffu;
                        tot4 += v[x];
                    }
                    uint32_t total = 0;
                    uint32_t off = block_scan(tot4, total);
                    ET_STAMP(6);                        // counts fetched + scanned
#pragma unroll
                    for (int x = 0; x < 4; x++) {
                        const uint32_t k = 4u * tid + (uint32_t)x;
                        if (k <= nc) L.scan[k] = off;
                        off += v[x];
                    }
                    __syncthreads();
                    // A wave takes a contiguous share of the items, 64 at a time.  The call of an item = the last k with scan[k] <= i (every
                    // call has an item): found by bisection for the wave's first item, then carried -- the calls that begin inside the
                    // 64 items of a round mark their first item in a mask, and an item's call is the one before the round plus the marks
                    // up to it (the bisection per item was a quarter of the kernel's instructions).
                    const uint32_t share = (total + ET_BLOCK - 1) / ET_BLOCK * 64u;
                    uint32_t kA = 0;
                    for (uint32_t I = wave * share; I < (wave + 1u) * share && I < total; I += 64u) {
                        if (I == wave * share) {
                            uint32_t lo_ = 0, hi_ = nc;
                            while (hi_ - lo_ > 1) { const uint32_t mid = (lo_ + hi_) >> 1; if (L.scan[mid] <= I) lo_ = mid; else hi_ = mid; }
                            kA = lo_;
                        }
                        if (lane == 0) L.wmask[wave] = 0ull;
                        const uint32_t kk = kA + 1u + lane;
                        const uint32_t sk = kk < nc ? L.scan[kk] : 0xffffffffu;
                        if (sk - I < 64u) atomicOr(&L.wmask[wave], 1ull << (sk - I));      // (sk > I: kA is the last call that begins at or before I)
                        const uint64_t marks = L.wmask[wave];
                        const uint32_t i = I + lane;
                        const uint32_t lo = kA + (uint32_t)__popcll(marks & ((2ull << lane) - 1ull));
                        kA += (uint32_t)__popcll(marks);
                        if (i >= total) continue;
                        const EtCall c = fetch(cb + lo);
                        const uint32_t cls = c.u0 % 3u;
                        const uint32_t e = c.i0 + (i - L.scan[lo]), iT = c.i0 + c.ni - 1u;
                        const uint32_t evw = L.ev[li][cls][e], evT = L.ev[li][cls][iT];
                        const uint32_t p = evw & 2047u, f = evw >> 11, pT = evT & 2047u, fT = evT >> 11;
                        const bool is_tail = e == iT, inclusive = (fT & EV_LASTFIT) != 0;
                        const bool trunc = (fT & (EV_END | EV_LASTFIT)) && a.allow_truncated;
                        const double *Sc = &L.S[cls][4];
                        const double p0 = c.p0z ? 0.0 : Sc[(int)c.u0 - 1];
                        const int n = (int)(L.roff[c.rl + 1] - L.roff[c.rl]);
                        auto emit = [&](double raw, int j_loc, int pos, int which, int truncated, int first_, uint32_t kind) __attribute__((always_inline)) {
                            const int j_full = j_loc + 2 + c.sj;
                            const double sc = (j_full > L.isl[c.rl] && 0.0 > raw) ? 0.0 : raw;
                            MgOrfAgg *g2 = &L.agg[c.oe];
                            atomicAdd(&g2->cnt, 1u);
                            atomicMax(&g2->best, (unsigned long long)mg_ord(sc));
                            const unsigned long long pa = (unsigned long long)(uint32_t)(pos + 16) << 32 | (uint32_t)j_full,
                                                     pb = (unsigned long long)(uint32_t)(pos + 16) << 32 | (0xffffffffu - (uint32_t)j_full);
                            if (fwd) { atomicMin(&g2->ext_a, pa); atomicMin(&g2->ext_b, pb); }
                            else { atomicMax(&g2->ext_a, pa); atomicMax(&g2->ext_b, pb); }
                            const uint32_t slot = atomicAdd(&L.n_em, 1u);
                            if (slot < ecap) {
                                EtEm r;
                                r.s.score = sc; r.s.j = j_full; r.s.pos = pos; r.s.which = which; r.s.truncated = (int16_t)truncated; r.s.first = (int16_t)first_;
                                r.e.pos[0] = level > 0 ? (int)(c.e0 >> 2) - 8 : 0; r.e.pos[1] = level > 1 ? (int)(c.e1 >> 2) - 8 : 0;
                                r.e.type[0] = (int8_t)(level > 0 ? (c.e0 & 3) : 0); r.e.type[1] = (int8_t)(level > 1 ? (c.e1 & 3) : 0);
                                r.e.n = (int8_t)level; r.e.reserved = 0;
                                r.oe = c.oe;
                                r.key = c.key | (uint64_t)((uint32_t)(2047 - j_loc) << 2 | kind) << key_shift;
                                em[slot] = r;
                            }
                        };
                        auto push = [&](int c_end, double c_score, int c_sj, uint32_t c_err, uint32_t c_field) __attribute__((always_inline)) {
                            // a call that cannot reach Min_Gene_Len before its read ends emits nothing, nor can a branch of it: not handed on
                            if (c_sj + (fwd ? c_end : n - c_end + 1) + 12 < mgl) return;
                            uint32_t u0c = 0, i0c = 0, nic = 0;
                            bool p0zc = false;
                            if (!call_geo(c_end, c.rl, list_of(level + 1), u0c, i0c, nic, p0zc)) return;       // (an empty region: nothing to do for a call behind level 0)
                            const uint32_t slot = atomicAdd(&L.qn[level], 1u);
                            if (slot >= qcap) return;                                       // (seen after the level: err_flag)
                            MgCall child;
                            const uint32_t ce0 = level == 0 ? c_err : c.e0, ce1 = level == 1 ? c_err : 0u;
                            child.w[0] = (unsigned long long)__double_as_longlong(c_score);
                            child.w[1] = (c.key | (uint64_t)c_field << key_shift) | (uint64_t)(uint32_t)(c_end + 8) << 40 | (uint64_t)(uint32_t)c_sj << 52;
                            child.w[2] = (uint64_t)c.rl | (uint64_t)u0c << 8 | (uint64_t)i0c << 20 | (uint64_t)nic << 32 | (uint64_t)(p0zc ? 1u : 0u) << 44;
                            child.w[3] = (uint64_t)c.oe | (uint64_t)ce0 << 32 | (uint64_t)ce1 << 46;
                            q_out[slot] = child;
                            if (slot < (uint32_t)ET_QNI) L.qni[level][slot] = (uint16_t)nic;
                        };
                        if ((!is_tail || inclusive) && (f & (EV_START | EV_LOWQ))) {
                            // the codon at p: buffer positions j0, j0 + 1, j0 + 2 of the call
                            const int j0 = (int)p - (int)c.u0;
                            const double *d = Sc + (int)p - 1;
                            const double prev = j0 ? d[0] - p0 : 0.0;              // score[j0 - 1]
                            if ((f & EV_START) && j0 >= lowest_j && j0 + 3 + c.sj >= mgl) {
                                const int k = fwd ? c.ep - 2 - j0 : c.ep + 2 + j0;
                                const int which = L.which[codons(p) & 63u];
                                const uint32_t nse = L.ns[li][cls][e];
                                const bool later = nse < iT || (nse == iT && inclusive);      // another start codon inside the region
                                emit((prev - 0.0) + c.ss, j0, k, which, 0, (!later && !trunc) ? 1 : 0, 3u);
                            }
                            if (can_branch && (f & EV_LOWQ)) {
                                // Score_Indels at the three positions: per position insertion and deletion
                                const double s0 = d[1] - p0, s1 = d[2] - p0, sum = d[3] - p0;
                                const uint32_t qw = quals4(p);
#pragma unroll
                                for (int pj = 0; pj < 3; pj++) {
                                    const int q = (int)((qw >> (8 * pj)) & 255u);
                                    if (!(j0 + pj >= lowest_j && q <= a.indel_q_thr)) continue;
                                    const double pen = pen_lds ? L.pen[q & 63] : a.pen[q];
                                    const double before = pj == 0 ? prev : pj == 1 ? s0 : s1, at = pj == 0 ? s0 : pj == 1 ? s1 : sum;
                                    const int j = j0 + pj;
                                    const int k = fwd ? c.ep - 2 - j : c.ep + 2 + j;
#pragma unroll
                                    for (int b = 0; b < 2; b++) {
                                        const double es = ((c.ss + (b == 0 ? before : at)) - 0.0) + pen;
                                        if (!(es > a.indel_suffix_thr)) continue;
                                        int c_end, epos;
                                        if (b == 0) { c_end = fwd ? k - (2 - pj) : k + 2 - pj; epos = fwd ? k + 2 : k - 2; }
                                        else { c_end = fwd ? k + pj : k - pj; epos = fwd ? k + 3 : k - 1; }
                                        push(c_end, es, c.sj + j + 2 - pj, (uint32_t)(epos + 8) << 2 | (uint32_t)b, (uint32_t)(2047 - j) << 2 | (b == 0 ? 1u : 0u));
                                    }
                                }
                            }
                        }
                        if (is_tail) {
                            // what the walk did when it left the region: m bases walked
                            const int m = (int)(inclusive ? pT + 3u : pT) - (int)c.u0;
                            if (m > 0 && trunc) {
                                const int j0 = m - 3;
                                if (j0 >= lowest_j && j0 + 3 + c.sj >= mgl) {
                                    const int k = fwd ? c.ep - 2 - j0 : c.ep + 2 + j0;
                                    const double prev = j0 ? Sc[(int)c.u0 + j0 - 1] - p0 : 0.0;
                                    emit((prev - 0.0) + c.ss, j0, k, -1, 1, 1, 2u);
                                }
                            }
                            if (level == 0) {
                                L.agg[c.oe].m0 = (uint32_t)m << 1 | (trunc ? 1u : 0u);
                                if (a.err_mode == 2) {  // the substitution branch (:1771-1806)
                                    const int lo = fwd ? c.ep - m : c.ep, hi = fwd ? c.ep : c.ep + m;
                                    const int eep = fwd ? lo - 3 : hi + 3;
                                    if (eep >= 0 && eep - 2 < n) {
                                        // the stop codon behind the region in walk order: steps m, m + 1, m + 2 from the anchor; its third and
                                        // second base as the strand reads it are steps m and m + 1 ("is it a": the reverse strand is complemented)
                                        const uint32_t two = codons(c.u0 + (uint32_t)m);
                                        const int a2 = (two & 3u) == 0u, a1 = ((two >> 2) & 3u) == 0u;
                                        double es = c.ss + a.pass_stop[a1 * 2 + a2];
                                        if (m > 0) es += (Sc[(int)c.u0 + m - 1] - p0) - 0.0;
                                        push(eep, es, c.sj + m, (uint32_t)((fwd ? lo - 2 : hi + 2) + 8) << 2 | 2u, 0u);
                                    }
                                }
                            }
                        }
                    }
                    ET_STAMP(8 + level);                // the wave's items
                    __syncthreads();                    // the chunk's items are done: the scan array and the queue counters may be read / reused
                    ET_STAMP(7);                        // ... waiting for the other waves
                }
                if (level < 2 && L.qn[level] > qcap && tid == 0) atomicOr(a.err_flag, 1u);
            }
            __syncthreads();
            // ---- Score_Orfs_Errors' verdict (:1647-1683) from what the ORF's calls added up to (k_mg_err_verdict), the kept ORFs'
            //      places in the staging arrays
            uint32_t n_keep = 0;
            uint64_t orf_i = 0;
            bool write_rec = false;
            gmg_mg_orf rec;
            if (tid < n_o) {
                const MgOrfAgg g = L.agg[tid];
                orf_i = eb + (L.o_inf[tid] & 0xffu);
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
                n_keep = (accepted_only && !accepted) ? 0u : g.cnt;
                a.orf_cnt[orf_i] = n_keep;
                if (accepted) atomicOr(&a.acc_bits[orf_i >> 5], 1u << (orf_i & 31u));
                if (!accepted_only || accepted) {
                    write_rec = true;
                    rec = a.orfs[orf_i];
                    const int m0 = (int)(g.m0 >> 1);
                    if (fwd) { rec.hi = rec.stop_position - 1; rec.lo = rec.hi - m0; }
                    else { rec.lo = rec.stop_position + 3; rec.hi = rec.lo + m0; }
                    rec.orf_is_truncated = (int16_t)(g.m0 & 1);
                    rec.n_starts = g.cnt;
                    rec.first_j = g.cnt ? jmin : 0;
                    rec.best_score = best_score;
                    rec.accepted = (int16_t)acc;
                }
                L.keep_cnt[tid] = n_keep;
            }
            uint32_t keep_total = 0;
            const uint32_t koff = block_scan(n_keep, keep_total);
            if (tid < n_o) L.keep_off[tid] = koff;
            if (tid == 0) {
                unsigned long long base = 0;
                if (keep_total) {
                    base = atomicAdd(stage_ctr, (unsigned long long)keep_total);
                    if (base + keep_total > stage_cap) { L.stage_ok = 0; atomicOr(a.err_flag, 2u); }
                }
                L.stage_base = base;
                if (L.n_em > ecap) atomicOr(a.err_flag, 1u);
            }
            __syncthreads();
            const unsigned long long sbase = L.stage_base;
            if (write_rec) {
                rec.start_begin = (uint32_t)(sbase + koff);                 // (its place in the staging arrays: k_et_unstage moves the slice)
                a.orfs[orf_i] = rec;
            }
            if (keep_total && L.stage_ok) {
                const uint32_t n_em = L.n_em < ecap ? L.n_em : ecap;
                for (uint32_t i = tid; i < n_em; i += ET_BLOCK) {
                    const EtEm r = em[i];
                    if (!L.keep_cnt[r.oe]) continue;
                    const unsigned long long slot = sbase + L.keep_off[r.oe] + atomicAdd(&L.fill[r.oe], 1u);
                    a.starts[slot] = r.s;
                    a.errs[slot] = r.e;
                    a.keys[slot] = r.key;
                }
            }
        }
        ET_STAMP(11);                                   // verdict, staging
        if (tid == 0) { L.nxt_o[0] = o0n; L.nxt_o[1] = o1n; }
        asm volatile("" :: "v"(p_touch[0]), "v"(p_touch[1]));  // (the touching loads end here at the latest)
        __syncthreads();                                // the next item is complete in L.nxt_*
        ET_STAMP(12);
    }
#if GMG_ET_STAMPS
    if (tid == 0)
        for (int i = 0; i < 16; i++) atomicAdd(&g_et_stamps[i], et_acc[i]);
#endif
}

// the staged slices to their final places (the scan of the counts): one lane per ORF the tile kernel has kept
__global__ __launch_bounds__(256) void k_et_unstage(MgArgs a, const gmg_start *st_s, const gmg_start_errors *st_e, const uint64_t *st_k, const int accepted_only)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n_orfs; i += (uint64_t)gridDim.x * blockDim.x) {
        if (accepted_only) { if (!((a.acc_bits[i >> 5] >> (i & 31u)) & 1u)) continue; }
        gmg_mg_orf *o = a.orfs + i;
        if (!a.read_fit[o->read]) continue;             // (k_mg_err_flat writes its ORFs in place)
        const uint32_t n = a.orf_cnt[i];
        const uint64_t src = o->start_begin, dst = a.start_off[i];
        for (uint32_t x = 0; x < n; x++) {
            a.starts[dst + x] = st_s[src + x];
            a.errs[dst + x] = st_e[src + x];
            a.keys[dst + x] = st_k[src + x];
        }
        o->start_begin = (uint32_t)dst;
    }
}
