// gmg_frame6.hip -- k_frame6s: six-frame per-position gene - null scores of whole reads on gfx950.
// Replaces Score_All_Frames (src/Glimmer/glimmer-mg.cc:1468-1510) = 12 x ICM_t::Frame_Score
// (src/ICM/icm.cc:485-509) + the double subtraction, for every read of a batch.
//
// Output row f (0..2):   reversed read scored with sub-model f, stored at forward coordinates:
//     window chars w[k] = S[p+W-1-k], predicted base S[p]          (glimmer-mg.cc:1482-1494)
// Output row 3+f:        complemented read:
//     window chars w[k] = comp(S[p-(W-1)+k]), predicted comp(S[p]) (glimmer-mg.cc:1497-1509)
//
// The kernel is integer/byte work + gathers (no MFMA); what it must stream to HBM is 48 B per
// base, so the design goal is to keep every table access on-chip and every store a full line.
//
// Work-group specialisation.  The grid is persistent: 3 x nworkers work-groups of 1024 lanes,
// one per CU.  Work-group type f = blockIdx % 3 owns sub-model f, i.e. output rows f and 3+f.
// Its LDS (160 KiB, all of a CU) holds for that ONE sub-model
//   s_leaf   as many leaf rows of the completed tree as fit (16 B each, ~9,600 of 16,384 at D = 7)
//   s_shift  the completed-tree shift table (2*mip, one byte per node, levels 0..D-1)
//   s_dense / s_part   the null model as direct tables (full and partial windows).
// A descent step is  ds_read_u8 ; v_bfe_u32 ; v_lshl_add_u32 ; the leaf row comes from LDS when
// cached and from the L2-resident crow table otherwise (~41 % of lookups on uniform reads).
// Each lane owns two adjacent bases of one chunk and scores them on both strands (four
// independent descents in flight), then writes one 16-byte store per output row: a wave writes
// 1 KiB of consecutive doubles per row.
//
// Partial windows (the first W-1 bases of either scoring buffer, icm.cc:807-842): the reference
// stops descending as soon as the context position named by a node lies before the buffer,
// i.e. when mip < (W-1) - j.  In the completed tree that is "shift byte < 2*((W-1)-j)", so waves
// that contain such lanes run the same loop with one compare per step and remember where they
// stopped; crow holds the right row for inner nodes too.  Waves without such lanes skip that.

#include "gmg_device.h"
#include <stdlib.h>

struct Frame6Args {
    GmgDevModel gene, nul;
    const uint32_t *packed;
    const uint64_t *off;
    const uint32_t *tile_read;
    uint64_t total, n_words;
    double *out;
    int uniform_len;
    int n_cached;          // leaf rows of one sub-model held in LDS
    int leaf_off;          // byte offset of those rows in LDS
};

// reverse the order of `nfields` 2-bit fields held in the low bits of y
__device__ __forceinline__ uint32_t dev_reverse_fields(uint32_t y, int nfields)
{
    uint32_t z = __brev(y) >> (32 - 2 * nfields);
    return ((z & 0x55555555u) << 1) | ((z >> 1) & 0x55555555u);
}

// One descent in the completed tree.  DT > 0: depth known at compile time (fully unrolled, level
// bases fold into the ds_read_u8 offset field).  thr2 = 2 * ((W-1) - j) for partial windows,
// <= 0 for full ones.  Returns the crow node index.
template <int DT, bool PARTIAL>
__device__ __forceinline__ uint32_t dev_ctree_node(const uint8_t *tab, uint32_t C, int D, int thr2)
{
    uint32_t idx = 0, lvl = 0, width = 1;
    uint32_t stop_node = 0xffffffffu;
    const int depth = DT > 0 ? DT : D;
#pragma unroll
    for (int l = 0; l < depth; l++) {
        uint32_t sh = tab[lvl + idx];
        if (PARTIAL) {
            if (stop_node == 0xffffffffu && (int)sh < thr2) stop_node = lvl + idx;
        }
        idx = (idx << 2) + ((C >> sh) & 3u);
        lvl += width;
        width <<= 2;
    }
    uint32_t node = lvl + idx;
    if (PARTIAL && stop_node != 0xffffffffu) node = stop_node;
    return node;
}

// position of job-wide base g inside its read: p = g - off[r], to_end = off[r+1] - 1 - g
__device__ __forceinline__ void dev_locate(const Frame6Args &a, uint64_t g, int &p, int &to_end)
{
    uint64_t r = a.tile_read[g / GMG_TILE];
    uint64_t r_end = a.off[r + 1];
    while (g >= r_end) { r++; r_end = a.off[r + 1]; }
    p = (int)(g - a.off[r]);
    to_end = (int)(r_end - 1 - g);
}

template <int V> struct F6Int { static constexpr int value = V; };
constexpr int f6_cstride(int dt) { return ((((1 << (2 * dt)) - 1) / 3) + 15) & ~15; }

// values of one chunk between "issued" and "stored"
struct F6Pend { float l[4], g[4], n[4]; uint32_t miss; };

// DIAG != 0 builds are timing-only ablations (wrong results), selected with GMG_DIAG for profiling:
//   1 no output stores   2 no leaf-row fetch   4 no descent   8 no packed-read window loads
//
// DT > 0: gene depth DT and a width-3 null model (the reference's Indep_Model(3,2,3)) are compile-time
// facts; the small tables then sit in static LDS at addresses the compiler folds into the ds_read
// offsets.  DT == 0: any depth / null width, everything in dynamic LDS.
//
// The chunk loop is software-pipelined (unrolled by two so that no register is ever copied) to keep
// the two long-latency accesses off the critical path of the LDS descents:
//   stage A (chunk i+1): the three packed-read words of the next chunk are loaded an iteration early;
//   stage B (chunk i)  : contexts, four LDS descents, then the leaf-row / null-table reads are ISSUED;
//   stage C (chunk i-1): the values fetched one iteration ago are widened, subtracted and stored.
//
// UNIFORM: every read has the same length (no offset-table loads in the loop).  PAIR: total_bases is
// even, so every output row is 16-byte aligned and a lane's two doubles go out as one dwordx4 store.
// Both are template parameters so that the steady-state loop issues a FIXED number of vector-memory
// operations per iteration: the compiler can then wait with counted vmcnt(N) for last iteration's
// leaf rows while this iteration's loads stay in flight (a data-dependent count forces vmcnt(0)).
template <int BLOCK, int DT, int DIAG, bool UNIFORM, bool PAIR>
__global__ __launch_bounds__(BLOCK) void k_frame6s(Frame6Args a)
{
    constexpr bool STATIC = DT > 0;
    constexpr int CS = STATIC ? f6_cstride(DT > 0 ? DT : 1) : 16;
    __shared__ __attribute__((aligned(16))) uint8_t st_shift[CS];
    __shared__ __attribute__((aligned(16))) float st_dense[STATIC ? 64 : 4];
    __shared__ __attribute__((aligned(16))) float st_part[STATIC ? 24 : 4];
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

    const int ftype = blockIdx.x % 3;
    const uint32_t worker = blockIdx.x / 3, nworkers = gridDim.x / 3;

    const int W = a.gene.W, D = STATIC ? DT : a.gene.D, Wn = STATIC ? 3 : a.nul.W;
    const int cstride = a.gene.cstride;
    const int n_dense = 1 << (2 * Wn);
    const int n_part = a.nul.n_dense_part;
    const uint32_t n_cached = (uint32_t)a.n_cached;
    const uint32_t leaf_base = ((1u << (2 * D)) - 1u) / 3u;

    uint8_t *s_shift = STATIC ? st_shift : lds;
    float *s_dense = STATIC ? st_dense : (float *)(lds + cstride);
    float *s_part = STATIC ? st_part : (float *)(lds + cstride) + n_dense;
    float *s_leaf = (float *)(lds + a.leaf_off);                    // [n_cached][4], 16-byte aligned

    const float *__restrict__ crow_f = a.gene.crow + (size_t)ftype * a.gene.ctot * 4;
    {
        const float4 *src = (const float4 *)(crow_f + (size_t)leaf_base * 4);
        for (uint32_t i = threadIdx.x; i < n_cached; i += BLOCK) ((float4 *)s_leaf)[i] = src[i];
        const uint8_t *sh_src = a.gene.cshift + (size_t)ftype * cstride;
        for (int i = threadIdx.x * 16; i < cstride; i += BLOCK * 16) *(uint4 *)(s_shift + i) = *(const uint4 *)(sh_src + i);
        for (int i = threadIdx.x; i < n_dense; i += BLOCK) s_dense[i] = a.nul.dense[(size_t)ftype * n_dense + i];
        for (int i = threadIdx.x; i < n_part; i += BLOCK) s_part[i] = a.nul.dense_part[(size_t)ftype * n_part + i];
    }
    __syncthreads();

    const uint32_t ctx_mask = (W >= 16) ? 0xffffffffu : ((1u << (2 * W)) - 1u);
    constexpr uint32_t SPAN = 2 * BLOCK;                            // bases per chunk
    const uint64_t n_chunks = a.total / SPAN;                       // full chunks (pipelined loop)
    const uint32_t tail = (uint32_t)(a.total % SPAN);               // bases in the last, partial chunk
    const int L = a.uniform_len;
    double *const out_f = a.out + (uint64_t)ftype * a.total;
    double *const out_r = a.out + (uint64_t)(3 + ftype) * a.total;

    // lane constants: this lane's two bases sit at chunk*SPAN + lane_off (+1); its window starts
    // W-1 bases earlier, i.e. at word (chunk*SPAN/16 - 1) + wword, bit wsh  (the -1 keeps wword >= 0)
    const uint32_t lane_off = 2 * threadIdx.x;
    const int first_rel = (int)lane_off - (W - 1) + 16;             // >= 0 for W <= 16
    const uint32_t wword = (uint32_t)first_rel >> 4;
    const uint32_t wsh = 2u * ((uint32_t)first_rel & 15u);

    // uniform-length reads: track the lane's offset inside its read across chunks without dividing
    int pu = 0, step_mod = 0;
    if (UNIFORM) {
        pu = (int)(((uint64_t)worker * SPAN + lane_off) % (uint64_t)L);
        step_mod = (int)(((uint64_t)nworkers * SPAN) % (uint64_t)L);
    }

    // rings of three: words loaded one iteration ahead; values retired TWO iterations after issue
    uint32_t raw_a[3] = {0, 0, 0}, raw_b[3] = {0, 0, 0}, raw_c[3] = {0, 0, 0};     // stage A -> B
    F6Pend pend_a, pend_b, pend_c;                                                 // stage B -> C

    auto load_raw = [&](uint64_t chunk, uint32_t (&w)[3]) __attribute__((always_inline)) {         // stage A
        const uint32_t *base = a.packed + chunk * (SPAN / 16) - 1;  // wave-uniform
        if (DIAG & 8) { w[0] = (uint32_t)chunk * 0x9E3779B9u + lane_off; w[1] = w[0] * 0x85EBCA6Bu; w[2] = w[1] ^ w[0]; return; }
        w[0] = base[wword]; w[1] = base[wword + 1]; w[2] = base[wword + 2];
    };

    auto finish = [&](uint64_t chunk, const F6Pend &pd, uint32_t rem) __attribute__((always_inline)) {   // stage C
        double v[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            // bitwise select (keeps both values in registers): miss -> L2 row, hit -> LDS row
            const uint32_t mk = 0u - ((pd.miss >> c) & 1u);
            const float gv = __uint_as_float((__float_as_uint(pd.l[c]) & ~mk) | (__float_as_uint(pd.g[c]) & mk));
            v[c] = (double)gv - (double)pd.n[c];                    // glimmer-mg.cc:1493,1508
        }
        if (DIAG & 1) {
            if (v[0] + v[1] + v[2] + v[3] == 1.2345e300) a.out[lane_off] = v[0];
            return;
        }
        const uint64_t cbase = chunk * SPAN;                        // wave-uniform bases + 32-bit lane offset
        double *pf = out_f + cbase, *pr = out_r + cbase;
        if (rem == SPAN) {                                          // full chunk: unconditional stores
            if (PAIR) {
                *(double2 *)(pf + lane_off) = make_double2(v[0], v[2]);
                *(double2 *)(pr + lane_off) = make_double2(v[1], v[3]);
            } else {
                pf[lane_off] = v[0]; pf[lane_off + 1] = v[2];
                pr[lane_off] = v[1]; pr[lane_off + 1] = v[3];
            }
        } else {                                                    // the job's last, partial chunk
            if (lane_off < rem) { pf[lane_off] = v[0]; pr[lane_off] = v[1]; }
            if (lane_off + 1 < rem) { pf[lane_off + 1] = v[2]; pr[lane_off + 1] = v[3]; }
        }
    };

    auto step = [&](uint64_t chunk, bool have_prev, bool prefetch, const uint32_t (&w)[3], uint32_t (&w_next)[3],
                    F6Pend &pd, const F6Pend &pd_prev) __attribute__((always_inline)) {
        // stage A: always issued (a fixed count of loads per iteration); past the last full chunk it
        // re-reads this chunk's words, which are never used
        load_raw(prefetch ? chunk + nworkers : chunk, w_next);

        // ---- stage B: window bits of this lane: field i = base g0-(W-1)+i
        const uint64_t lo = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
        const uint64_t x = (lo >> wsh) | (((uint64_t)w[2] << 1) << (63 - wsh));

        int p[2], to_end[2];
        if (UNIFORM) {
            p[0] = pu;
            p[1] = (pu + 1 == L) ? 0 : pu + 1;
            to_end[0] = L - 1 - p[0];
            to_end[1] = L - 1 - p[1];
            pu += step_mod;
            if (pu >= L) pu -= L;
        } else {
            const uint64_t g0 = chunk * SPAN + lane_off;
            const uint64_t gq = g0 < a.total ? g0 : a.total - 1;    // idle tail lanes shadow the last base
            dev_locate(a, gq, p[0], to_end[0]);
            if (gq + 1 < a.total) dev_locate(a, gq + 1, p[1], to_end[1]);
            else { p[1] = p[0]; to_end[1] = to_end[0]; }
        }

        uint32_t C[4];
        int jj[4];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const uint64_t xq = x >> (2 * q);
            // fields 0..W-1 = S[p-(W-1)..p], fields W-1..2W-2 = S[p..p+W-1]
            C[2 * q + 0] = dev_reverse_fields((uint32_t)(xq >> (2 * (W - 1))) & ctx_mask, W);   // forward rows: S[p+W-1-k]
            C[2 * q + 1] = ((uint32_t)xq & ctx_mask) ^ ctx_mask;                               // reverse rows: comp(S[p-(W-1)+k])
            jj[2 * q + 0] = to_end[q];                                                         // index in the reversed buffer
            jj[2 * q + 1] = p[q];                                                              // index in the complemented buffer
        }
        const int jmin = min(min(jj[0], jj[1]), min(jj[2], jj[3]));
        const bool any_partial = __any(jmin < W - 1);               // wave-uniform

        uint32_t node[4];
        if (!any_partial) {
            // every window of this wave is full: fixed-depth descents, direct null table
#pragma unroll
            for (int c = 0; c < 4; c++)
                node[c] = (DIAG & 4) ? leaf_base + ((C[c] >> 3) & ((1u << (2 * D)) - 1u))
                                     : dev_ctree_node<DT, false>(s_shift, C[c], D, 0);
#pragma unroll
            for (int c = 0; c < 4; c++) pd.n[c] = s_dense[C[c] >> (2 * (W - Wn))];             // last Wn chars of the window
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++) node[c] = dev_ctree_node<DT, true>(s_shift, C[c], D, 2 * ((W - 1) - jj[c]));
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int j = jj[c];
                if (j >= Wn - 1) pd.n[c] = s_dense[C[c] >> (2 * (W - Wn))];
                else pd.n[c] = s_part[(C[c] >> (2 * (W - 1 - j))) + (((1u << (2 * (j + 1))) - 4u) / 3u)];   // B[0..j]
            }
        }

        // ---- issue the leaf-row reads of this chunk
        uint32_t miss = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t pred = (C[c] >> (2 * (W - 1))) & 3u;
            const uint32_t rel = node[c] - leaf_base;               // wraps for inner (stopped) nodes
            const bool m = rel >= n_cached;
            const uint32_t rel_c = m ? 0u : rel;
            pd.l[c] = (DIAG & 2) ? __uint_as_float(node[c] + pred) : s_leaf[rel_c * 4 + pred];
            // unconditional gather: hit lanes all read row 0 (one line), so the load count is fixed
            pd.g[c] = (DIAG & 2) ? 0.f : crow_f[(m ? node[c] : 0u) * 4 + pred];
            miss |= (m ? 1u : 0u) << c;
        }
        pd.miss = miss;

        // ---- stage C: retire the previous chunk
        if (have_prev) finish(chunk - 2 * (uint64_t)nworkers, pd_prev, SPAN);
    };

    if (worker < n_chunks) {
        load_raw(worker, raw_a);
        uint64_t chunk = worker;
        uint64_t k = 0;                                             // iterations done by this worker
        int phase = 0;
        while (true) {
            step(chunk, k >= 2, chunk + nworkers < n_chunks, raw_a, raw_b, pend_a, pend_b);
            k++; phase = 1; chunk += nworkers;
            if (chunk >= n_chunks) break;
            step(chunk, k >= 2, chunk + nworkers < n_chunks, raw_b, raw_c, pend_b, pend_c);
            k++; phase = 2; chunk += nworkers;
            if (chunk >= n_chunks) break;
            step(chunk, k >= 2, chunk + nworkers < n_chunks, raw_c, raw_a, pend_c, pend_a);
            k++; phase = 0; chunk += nworkers;
            if (chunk >= n_chunks) break;
        }
        // drain: the last two issued sets, oldest first.  `chunk` is one stride past the last chunk.
        const uint64_t c1 = chunk - nworkers, c2 = chunk - 2 * (uint64_t)nworkers;
        if (phase == 1) { if (k >= 2) finish(c2, pend_c, SPAN); finish(c1, pend_a, SPAN); }
        else if (phase == 2) { if (k >= 2) finish(c2, pend_a, SPAN); finish(c1, pend_b, SPAN); }
        else { if (k >= 2) finish(c2, pend_b, SPAN); finish(c1, pend_c, SPAN); }
    }

    // the job's last, partial chunk: one worker, no pipelining
    if (tail != 0 && (uint32_t)(n_chunks % nworkers) == worker) {
        if (UNIFORM) pu = (int)((n_chunks * SPAN + lane_off) % (uint64_t)L);
        load_raw(n_chunks, raw_a);
        step(n_chunks, false, false, raw_a, raw_b, pend_a, pend_b);
        finish(n_chunks, pend_a, tail);
    }
}

// Any-shape kernel: exact plain descent on the original tables for both models (used when the
// gene model has no completed tree or the null model no direct tables; same results, slower).
__global__ __launch_bounds__(256) void k_frame6_generic(Frame6Args a)
{
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < a.total;
         g += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t r = a.tile_read[g / GMG_TILE];
        uint64_t r_end = a.off[r + 1];
        while (g >= r_end) { r++; r_end = a.off[r + 1]; }
        const uint64_t r_off = a.off[r];
        const int L = (int)(r_end - r_off);
        const int p = (int)(g - r_off);
        DevBuf bf = dev_make_buf(a.packed, r_off, 0, (uint32_t)L, GMG_REVERSED);
        DevBuf br = dev_make_buf(a.packed, r_off, 0, (uint32_t)L, GMG_COMPLEMENTED);
        for (int f = 0; f < 3; f++) {
            a.out[(uint64_t)f * a.total + g] =
                (double)dev_score(a.gene, bf, L - 1 - p, f) - (double)dev_score(a.nul, bf, L - 1 - p, f);
            a.out[(uint64_t)(3 + f) * a.total + g] =
                (double)dev_score(a.gene, br, p, f) - (double)dev_score(a.nul, br, p, f);
        }
    }
}

int gmg_launch_frame6(const gmg_model *gene, const gmg_model *nul, const gmg_reads *reads, double *d_out,
                      hipStream_t s)
{
    Frame6Args a;
    a.gene = gene->dev;
    a.nul = nul->dev;
    a.packed = reads->d_packed;
    a.off = reads->d_off;
    a.tile_read = reads->d_tile_read;
    a.total = reads->total_bases;
    a.n_words = reads->n_words;
    a.out = d_out;
    a.uniform_len = reads->uniform_len;
    a.n_cached = 0;
    a.leaf_off = 0;

    const bool fast = gene->dev.has_fast && nul->dev.has_dense && nul->dev.W <= gene->dev.W;
    const size_t lds_max = 160 * 1024;
    size_t fixed = 0;
    // DT = 7 build: small tables in static LDS (5,472 + 256 + 96 bytes), leaf rows in dynamic LDS
    const bool is_static = fast && a.gene.D == 7 && a.nul.W == 3;
    const size_t static_lds = (size_t)f6_cstride(7) + 64 * 4 + 24 * 4;
    if (fast) fixed = (size_t)a.gene.cstride + ((size_t)1 << (2 * a.nul.W)) * 4 + (size_t)a.nul.n_dense_part * 4;
    if (!fast || fixed + 4096 > lds_max) {
        const uint64_t n_chunks = (a.total + 255) / 256;
        unsigned grid = (unsigned)(n_chunks < 256 * 16 ? n_chunks : 256 * 16);
        hipLaunchKernelGGL(k_frame6_generic, dim3(grid), dim3(256), 0, s, a);
        GMG_HIP(hipGetLastError());
        return GMG_OK;
    }
    constexpr int BLOCK = 1024;
    const size_t n_leaf = (size_t)1 << (2 * a.gene.D);
    size_t n_cached = (lds_max - ((fixed + 15) & ~(size_t)15)) / 16;
    if (is_static) n_cached = (lds_max - static_lds) / 16;
    if (n_cached > n_leaf) n_cached = n_leaf;
    if (const char *e = getenv("GMG_NCACHED")) {       // profiling aid: shrink the LDS leaf cache
        size_t v = (size_t)atol(e);
        if (v < n_cached) n_cached = v;
    }
    a.n_cached = (int)n_cached;
    a.leaf_off = is_static ? 0 : (int)((fixed + 15) & ~(size_t)15);
    const size_t lds = (size_t)a.leaf_off + n_cached * 16;      // dynamic part

    int dev = 0, n_cu = 256;
    GMG_HIP(hipGetDevice(&dev));
    GMG_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    // persistent grid: one work-group per CU, a multiple of the 3 sub-model types
    const uint64_t n_chunks = (a.total + 2 * BLOCK - 1) / (2 * BLOCK);
    unsigned nworkers = (unsigned)(n_cu / 3);
    if (nworkers < 1) nworkers = 1;
    if (nworkers > n_chunks) nworkers = (unsigned)n_chunks;
    const unsigned grid = 3 * nworkers;

    const char *env = getenv("GMG_DIAG");
    const int diag = env ? atoi(env) : 0;
    const bool uni = a.uniform_len > 0, pair = (a.total & 1) == 0;
#define GMG_LAUNCH_F6(DT_, DIAG_, U_, P_)                                                               \
    do {                                                                                                \
        GMG_HIP(hipFuncSetAttribute((const void *)k_frame6s<BLOCK, DT_, DIAG_, U_, P_>,                 \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));             \
        hipLaunchKernelGGL((k_frame6s<BLOCK, DT_, DIAG_, U_, P_>), dim3(grid), dim3(BLOCK), lds, s, a); \
    } while (0)
#define GMG_LAUNCH_F6_UP(DT_)                                                                           \
    do {                                                                                                \
        if (uni && pair) GMG_LAUNCH_F6(DT_, 0, true, true);                                             \
        else if (uni) GMG_LAUNCH_F6(DT_, 0, true, false);                                               \
        else if (pair) GMG_LAUNCH_F6(DT_, 0, false, true);                                              \
        else GMG_LAUNCH_F6(DT_, 0, false, false);                                                       \
    } while (0)
    if (diag != 0) {
        if (!(is_static && uni && pair))
            return gmg_set_error(GMG_EINVAL, "GMG_DIAG ablations exist only for depth 7, uniform even-sized batches");
        switch (diag) {
        case 1: GMG_LAUNCH_F6(7, 1, true, true); break;
        case 2: GMG_LAUNCH_F6(7, 2, true, true); break;
        case 3: GMG_LAUNCH_F6(7, 3, true, true); break;
        case 4: GMG_LAUNCH_F6(7, 4, true, true); break;
        case 8: GMG_LAUNCH_F6(7, 8, true, true); break;
        case 14: GMG_LAUNCH_F6(7, 14, true, true); break;
        case 15: GMG_LAUNCH_F6(7, 15, true, true); break;
        default: return gmg_set_error(GMG_EINVAL, "GMG_DIAG=%d is not a built ablation", diag);
        }
    } else if (is_static) {
        GMG_LAUNCH_F6_UP(7);
    } else {
        GMG_LAUNCH_F6_UP(0);
    }
#undef GMG_LAUNCH_F6_UP
#undef GMG_LAUNCH_F6
    GMG_HIP(hipGetLastError());
    return GMG_OK;
}
