// gmg_frame6.hip -- k_frame6: six-frame per-position gene - null scores of whole reads on gfx950.
// Replaces Score_All_Frames (src/Glimmer/glimmer-mg.cc:1468-1510) = 12 x ICM_t::Frame_Score
// (src/ICM/icm.cc:485-509) + the double subtraction, for every read of a batch.
//
// Output row f (0..2):   reversed read scored with sub-model f, stored at forward coordinates:
//     window chars w[k] = S[p+W-1-k], predicted base S[p]          (glimmer-mg.cc:1482-1494)
// Output row 3+f:        complemented read:
//     window chars w[k] = comp(S[p-(W-1)+k]), predicted comp(S[p]) (glimmer-mg.cc:1497-1509)
//
// Mapping: one lane owns one base p of one read and produces its six doubles, so each store
// instruction writes 64 consecutive doubles of one output row (coalesced).  The kernel is
// integer/byte work + gathers, no MFMA; what it streams to HBM is 48 B per base.
//
// Tables (built at upload, gmg_api.hip):
//   cshift  completed-tree shift table (2*mip, one byte per node, levels 0..D-1) -> LDS.
//           One descent step is  ds_read_u8 ; v_bfe_u32 ; v_lshl_add_u32 .
//   crow    row used when a descent ends at a completed-tree node -> gathered from L2 (1 MB).
//   dense   null model as direct tables (full and partial windows) -> LDS.
// Context registers CF / CR hold window char k in bits [2k, 2k+1].
//
// Partial windows (the first W-1 bases of either scoring buffer, icm.cc:807-842): the reference
// stops descending as soon as the context position named by a node lies before the buffer,
// i.e. when mip < (W-1) - j.  In the completed tree that is "shift byte < 2*((W-1)-j)", so waves
// that contain such lanes run the same loop with one compare per step and remember where they
// stopped; crow holds the right row for inner nodes too.  Waves without such lanes (most) skip it.

#include "gmg_device.h"

struct Frame6Args {
    GmgDevModel gene, nul;
    const uint32_t *packed;
    const uint64_t *off;
    const uint32_t *tile_read;
    uint64_t total, n_words;
    double *out;
    int uniform_len;
};

// all 2W-1 bases around job-wide base g: base g-(W-1)+i at bits [2i, 2i+1]
__device__ __forceinline__ uint64_t dev_window_bits(const uint32_t *__restrict__ packed, uint64_t n_words,
                                                    int64_t first)
{
    // `first` may be negative or run past the data for lanes whose window leaves the read; those
    // lanes never use the missing bits, the clamps only keep the loads inside the buffer.
    int64_t fc = first < 0 ? 0 : first;
    unsigned deficit = (unsigned)(fc - first);          // bases missing before the start of the job
    uint64_t w0 = (uint64_t)fc >> 4;
    uint64_t last = n_words - 1;
    uint64_t i0 = w0 < last ? w0 : last, i1 = w0 + 1 < last ? w0 + 1 : last, i2 = w0 + 2 < last ? w0 + 2 : last;
    uint64_t lo = (uint64_t)packed[i0] | ((uint64_t)packed[i1] << 32);
    uint64_t hi = packed[i2];
    unsigned sh = 2u * (unsigned)(fc & 15);
    uint64_t x = lo >> sh;
    if (sh) x |= hi << (64 - sh);
    x <<= 2u * deficit;                                 // keep base g-(W-1)+i at field i
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

template <int BLOCK, int DT>
__global__ __launch_bounds__(BLOCK) void k_frame6(Frame6Args a)
{
    extern __shared__ uint8_t lds[];
    const int cstride = a.gene.cstride;
    const int shift_bytes = 3 * cstride;
    const int Wn = a.nul.W;
    const int n_dense = 1 << (2 * Wn);
    const int n_part = a.nul.n_dense_part;
    uint8_t *s_shift = lds;
    float *s_dense = (float *)(lds + shift_bytes);        // [3][n_dense]
    float *s_part = s_dense + 3 * n_dense;                // [3][n_part]

    for (int i = threadIdx.x * 16; i < shift_bytes; i += BLOCK * 16)
        *(uint4 *)(s_shift + i) = *(const uint4 *)(a.gene.cshift + i);
    for (int i = threadIdx.x; i < 3 * n_dense; i += BLOCK) s_dense[i] = a.nul.dense[i];
    for (int i = threadIdx.x; i < 3 * n_part; i += BLOCK) s_part[i] = a.nul.dense_part[i];
    __syncthreads();

    const int W = a.gene.W, D = a.gene.D;
    const uint32_t ctx_mask = (W >= 16) ? 0xffffffffu : ((1u << (2 * W)) - 1u);
    const uint32_t ctot = (uint32_t)a.gene.ctot;
    const float *__restrict__ crow = a.gene.crow;
    const uint64_t n_chunks = (a.total + BLOCK - 1) / BLOCK;

    for (uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const uint64_t g = chunk * BLOCK + threadIdx.x;
        const bool live = g < a.total;
        const uint64_t gq = live ? g : a.total - 1;       // idle tail lanes shadow the last base

        // ---- which read, where in it
        uint64_t r_off, r_end;
        if (a.uniform_len > 0) {
            uint64_t r = gq / (uint64_t)a.uniform_len;
            r_off = r * (uint64_t)a.uniform_len;
            r_end = r_off + (uint64_t)a.uniform_len;
        } else {
            uint64_t r = a.tile_read[gq / GMG_TILE];
            r_end = a.off[r + 1];
            while (gq >= r_end) { r++; r_end = a.off[r + 1]; }
            r_off = a.off[r];
        }
        const int L = (int)(r_end - r_off);
        const int p = (int)(gq - r_off);

        // ---- context registers
        const uint64_t x = dev_window_bits(a.packed, a.n_words, (int64_t)gq - (W - 1));
        // fields 0..W-1 of x = S[p-(W-1)..p], fields W-1..2W-2 = S[p..p+W-1]
        const uint32_t CR = ((uint32_t)x & ctx_mask) ^ ctx_mask;                                // comp(S[p-(W-1)+k])
        const uint32_t CF = dev_reverse_fields((uint32_t)(x >> (2 * (W - 1))) & ctx_mask, W);  // S[p+W-1-k]

#pragma unroll
        for (int strand = 0; strand < 2; strand++) {
            const uint32_t C = strand ? CR : CF;
            const int j = strand ? p : L - 1 - p;          // index in the scoring buffer
            const int thr2 = 2 * ((W - 1) - j);            // > 0  <=>  partial gene window
            const uint32_t pred = (C >> (2 * (W - 1))) & 3u;
            const bool any_partial = __any(thr2 > 0);

            // null model: direct tables.  Full window = last Wn chars of the gene window.
            uint32_t nul_slot;
            if (j >= Wn - 1) {
                nul_slot = C >> (2 * (W - Wn));
            } else {
                // B[0..j] are gene window chars W-1-j .. W-1; position j's table starts at (4^(j+1)-4)/3
                nul_slot = (C >> (2 * (W - 1 - j))) + (((1u << (2 * (j + 1))) - 4u) / 3u);
            }
            const float *nul_tab = (j >= Wn - 1) ? s_dense : s_part;
            const int nul_stride = (j >= Wn - 1) ? n_dense : n_part;

#pragma unroll
            for (int f = 0; f < 3; f++) {
                const uint8_t *tab = s_shift + f * cstride;
                uint32_t node;
                if (any_partial) node = dev_ctree_node<DT, true>(tab, C, D, thr2);
                else node = dev_ctree_node<DT, false>(tab, C, D, 0);
                const float gv = crow[((size_t)f * ctot + node) * 4 + pred];
                const float nv = nul_tab[f * nul_stride + nul_slot];
                // glimmer-mg.cc:1493,1508: double(gene) - double(null)
                if (live) a.out[(uint64_t)(strand * 3 + f) * a.total + g] = (double)gv - (double)nv;
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

    constexpr int BLOCK = 256;
    const uint64_t n_chunks = (a.total + BLOCK - 1) / BLOCK;
    const bool fast = gene->dev.has_fast && nul->dev.has_dense && nul->dev.W <= gene->dev.W;
    size_t lds = 0;
    if (fast) {
        lds = (size_t)3 * a.gene.cstride + (size_t)3 * ((size_t)1 << (2 * a.nul.W)) * 4 + (size_t)3 * a.nul.n_dense_part * 4;
    }
    if (!fast || lds > 150 * 1024) {
        unsigned grid = (unsigned)(n_chunks < 256 * 16 ? n_chunks : 256 * 16);
        hipLaunchKernelGGL(k_frame6_generic, dim3(grid), dim3(256), 0, s, a);
        GMG_HIP(hipGetLastError());
        return GMG_OK;
    }
    unsigned grid = (unsigned)(n_chunks < 256 * 8 ? n_chunks : 256 * 8);
    if (a.gene.D == 7) {
        if (lds > 64 * 1024)
            GMG_HIP(hipFuncSetAttribute((const void *)k_frame6<BLOCK, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_frame6<BLOCK, 7>), dim3(grid), dim3(BLOCK), lds, s, a);
    } else {
        if (lds > 64 * 1024)
            GMG_HIP(hipFuncSetAttribute((const void *)k_frame6<BLOCK, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_frame6<BLOCK, 0>), dim3(grid), dim3(BLOCK), lds, s, a);
    }
    GMG_HIP(hipGetLastError());
    return GMG_OK;
}
