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
};

// Window of packed bases: base `first`+i at bits [2i, 2i+1], 32 bases.  The packed buffer has
// GMG_GUARD_WORDS zero words on both sides, so `first` may be slightly negative or run past the data.
__device__ __forceinline__ uint64_t dev_window_bits(const uint32_t *__restrict__ packed, int64_t first)
{
    const int64_t w0 = first >> 4;                      // arithmetic: floor for negatives
    const unsigned sh = 2u * (unsigned)(first & 15);
    const uint64_t lo = (uint64_t)packed[w0] | ((uint64_t)packed[w0 + 1] << 32);
    const uint64_t hi = packed[w0 + 2];
    uint64_t x = lo >> sh;
    if (sh) x |= hi << (64 - sh);
    return x;
}

// reverse the order of `nfields` 2-bit fields held in the low bits of y
__device__ __forceinline__ uint32_t dev_reverse_fields(uint32_t y, int nfields)
{
    uint32_t z = __brev(y) >> (32 - 2 * nfields);
    return ((z & 0x55555555u) << 1) | ((z >> 1) & 0x55555555u);
}

// One descent in the completed tree.  DT > 0: depth known at compile time (fully unrolled).
// thr2 = 2 * ((W-1) - j) for partial windows, <= 0 for full ones.  Returns the crow node index.
template <int DT, bool PARTIAL>
__device__ __forceinline__ uint32_t dev_ctree_node(const uint8_t *__restrict__ tab, uint32_t C, int D, int thr2)
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

// DIAG != 0 builds are timing-only ablations (wrong results), selected with GMG_DIAG for profiling:
//   1 no output stores   2 no leaf-row fetch   4 no descent   8 no packed-read window loads
template <int BLOCK, int DT, int DIAG = 0>
__global__ __launch_bounds__(BLOCK) void k_frame6s(Frame6Args a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int ftype = blockIdx.x % 3;
    const uint32_t worker = blockIdx.x / 3, nworkers = gridDim.x / 3;

    const int W = a.gene.W, D = DT > 0 ? DT : a.gene.D, Wn = a.nul.W;
    const int cstride = a.gene.cstride;
    const int n_dense = 1 << (2 * Wn);
    const int n_part = a.nul.n_dense_part;
    const uint32_t n_cached = (uint32_t)a.n_cached;
    const uint32_t leaf_base = ((1u << (2 * D)) - 1u) / 3u;

    float *s_leaf = (float *)lds;                                   // [n_cached][4]
    uint8_t *s_shift = lds + (size_t)n_cached * 16;                 // [cstride]
    float *s_dense = (float *)(s_shift + cstride);                  // [n_dense]
    float *s_part = s_dense + n_dense;                              // [n_part]

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
    constexpr uint64_t SPAN = 2 * BLOCK;                            // bases per chunk
    const uint64_t n_chunks = (a.total + SPAN - 1) / SPAN;
    const uint64_t last_even = (a.total - 1) & ~1ull;
    const bool pair_store = (a.total & 1) == 0;                     // every row starts 16-byte aligned
    const int L = a.uniform_len;

    // uniform-length reads: track the lane's offset inside its read across chunks without dividing
    int pu = 0, step_mod = 0;
    if (L > 0) {
        pu = (int)(((uint64_t)worker * SPAN + 2 * threadIdx.x) % (uint64_t)L);
        step_mod = (int)(((uint64_t)nworkers * SPAN) % (uint64_t)L);
    }

    for (uint64_t chunk = worker; chunk < n_chunks; chunk += nworkers) {
        const uint64_t g0 = chunk * SPAN + 2 * threadIdx.x;         // this lane's bases: g0, g0+1
        const uint64_t gq = g0 <= last_even ? g0 : last_even;       // idle tail lanes shadow the last pair

        // ---- where the two bases sit in their reads
        int p[2], to_end[2];
        if (L > 0) {
            p[0] = pu;
            p[1] = (pu + 1 == L) ? 0 : pu + 1;
            to_end[0] = L - 1 - p[0];
            to_end[1] = L - 1 - p[1];
            pu += step_mod;
            if (pu >= L) pu -= L;
        } else {
            dev_locate(a, gq, p[0], to_end[0]);
            if (gq + 1 < a.total) dev_locate(a, gq + 1, p[1], to_end[1]);
            else { p[1] = p[0]; to_end[1] = to_end[0]; }
        }

        // ---- context registers.  x field i = base gq-(W-1)+i
        const uint64_t x = (DIAG & 8) ? (gq * 0x9E3779B97F4A7C15ull) : dev_window_bits(a.packed, (int64_t)gq - (W - 1));
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
        int thr2[4];
        bool part = false;
#pragma unroll
        for (int c = 0; c < 4; c++) { thr2[c] = 2 * ((W - 1) - jj[c]); part |= thr2[c] > 0; }
        const bool any_partial = __any(part);

        // ---- four descents
        uint32_t node[4];
        if (DIAG & 4) {
#pragma unroll
            for (int c = 0; c < 4; c++) node[c] = leaf_base + ((C[c] >> 3) & ((1u << (2 * D)) - 1u));
        } else if (any_partial) {
#pragma unroll
            for (int c = 0; c < 4; c++) node[c] = dev_ctree_node<DT, true>(s_shift, C[c], D, thr2[c]);
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++) node[c] = dev_ctree_node<DT, false>(s_shift, C[c], D, 0);
        }

        // ---- leaf rows, null model, difference
        double v[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t pred = (C[c] >> (2 * (W - 1))) & 3u;
            const uint32_t rel = node[c] - leaf_base;               // wraps for inner (stopped) nodes
            float gv;
            if (DIAG & 2) gv = __uint_as_float(node[c] + pred);
            else if (rel < n_cached) gv = s_leaf[rel * 4 + pred];
            else gv = crow_f[(size_t)node[c] * 4 + pred];
            const int j = jj[c];
            float nv;
            if (j >= Wn - 1) nv = s_dense[C[c] >> (2 * (W - Wn))];                 // last Wn chars of the window
            else nv = s_part[(C[c] >> (2 * (W - 1 - j))) + (((1u << (2 * (j + 1))) - 4u) / 3u)];   // B[0..j]
            v[c] = (double)gv - (double)nv;                         // glimmer-mg.cc:1493,1508
        }

        // ---- stores: rows f (forward) and 3+f (reverse), bases g0 and g0+1
        if (DIAG & 1) {
            if (v[0] + v[1] + v[2] + v[3] == 1.2345e300) a.out[g0] = v[0];
        } else {
            double *row_f = a.out + (uint64_t)ftype * a.total + g0;
            double *row_r = a.out + (uint64_t)(3 + ftype) * a.total + g0;
            if (g0 + 1 < a.total) {
                if (pair_store) {
                    *(double2 *)row_f = make_double2(v[0], v[2]);
                    *(double2 *)row_r = make_double2(v[1], v[3]);
                } else {
                    row_f[0] = v[0]; row_f[1] = v[2];
                    row_r[0] = v[1]; row_r[1] = v[3];
                }
            } else if (g0 < a.total) {
                row_f[0] = v[0];
                row_r[0] = v[1];
            }
        }
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

    const bool fast = gene->dev.has_fast && nul->dev.has_dense && nul->dev.W <= gene->dev.W;
    const size_t lds_max = 160 * 1024;
    size_t fixed = 0;
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
    size_t n_cached = (lds_max - fixed) / 16;
    if (n_cached > n_leaf) n_cached = n_leaf;
    a.n_cached = (int)n_cached;
    const size_t lds = n_cached * 16 + fixed;

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
#define GMG_LAUNCH_F6(DT_, DIAG_)                                                                       \
    do {                                                                                                \
        GMG_HIP(hipFuncSetAttribute((const void *)k_frame6s<BLOCK, DT_, DIAG_>,                         \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));             \
        hipLaunchKernelGGL((k_frame6s<BLOCK, DT_, DIAG_>), dim3(grid), dim3(BLOCK), lds, s, a);         \
    } while (0)
    if (a.gene.D == 7) {
        switch (diag) {
        case 0: GMG_LAUNCH_F6(7, 0); break;
        case 1: GMG_LAUNCH_F6(7, 1); break;
        case 2: GMG_LAUNCH_F6(7, 2); break;
        case 3: GMG_LAUNCH_F6(7, 3); break;
        case 4: GMG_LAUNCH_F6(7, 4); break;
        case 6: GMG_LAUNCH_F6(7, 6); break;
        case 7: GMG_LAUNCH_F6(7, 7); break;
        case 8: GMG_LAUNCH_F6(7, 8); break;
        case 14: GMG_LAUNCH_F6(7, 14); break;
        case 15: GMG_LAUNCH_F6(7, 15); break;
        default: return gmg_set_error(GMG_EINVAL, "GMG_DIAG=%d is not a built ablation", diag);
        }
    } else {
        GMG_LAUNCH_F6(0, 0);
    }
#undef GMG_LAUNCH_F6
    GMG_HIP(hipGetLastError());
    return GMG_OK;
}
