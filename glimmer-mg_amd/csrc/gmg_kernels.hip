// gmg_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4).
//
// The work is table gather + integer index arithmetic (no MFMA): one base score is a
// data-dependent descent of a 4-ary tree followed by one fp32 lookup
// (src/ICM/icm.cc:557-610, 807-842).  Kernels:
//   k_tile_read       read containing the first base of every 1024-base tile
//   k_frame6          six-frame per-position gene - null scores of whole reads
//                     (src/Glimmer/glimmer-mg.cc:1468-1510): completed-tree shift table in LDS,
//                     leaf rows gathered from L2, coalesced double stores
//   k_seg_frame       ICM_t::Frame_Score on arbitrary segments          (icm.cc:485-509)
//   k_seg_cum         ICM_t::Cumulative_Score / Score_String, sequential double adds in
//                     reference order                                    (icm.cc:354-405, 864-903)
//   k_all_frame       All_Frame_Score incl. Permute_By_Frame            (glimmer3.cc:328-359,1013-1088)
//   k_windows         Full_Window_Prob / Full_Window_Distrib            (icm.cc:512-610)

#include "gmg_internal.h"

#define WAVE 64

// ---------------------------------------------------------------------------
// shared device helpers
// ---------------------------------------------------------------------------

__device__ __forceinline__ int dev_parent(int x) { return (x - 1) / 4; }   // icm.hh:84

// 2-bit code of job-wide base g
__device__ __forceinline__ int dev_code(const uint32_t *__restrict__ packed, uint64_t g)
{
    return (int)((packed[g >> 4] >> (2 * (unsigned)(g & 15))) & 3u);
}

// A scoring buffer B cut from a read (gmg_orient in include/gmg.h).
struct DevBuf {
    const uint32_t *packed;
    uint64_t base;     // job-wide index of S[lo]
    int len;
    int rev;           // B[j] reads S[lo+len-1-j]
    int comp;          // B[j] is complemented
    __device__ __forceinline__ int at(int j) const
    {
        int c = dev_code(packed, base + (uint64_t)(rev ? len - 1 - j : j));
        return comp ? 3 - c : c;
    }
};

__device__ __forceinline__ DevBuf dev_make_buf(const uint32_t *packed, uint64_t read_base, uint32_t lo,
                                               uint32_t len, uint32_t orient)
{
    DevBuf b;
    b.packed = packed;
    b.base = read_base + lo;
    b.len = (int)len;
    b.rev = (orient == GMG_REVERSED || orient == GMG_REVCOMP);
    b.comp = (orient == GMG_COMPLEMENTED || orient == GMG_REVCOMP);
    return b;
}

// Node whose row scores buffer position j under sub-model f: the plain descent on the ORIGINAL
// tables in HBM/L2.  Full window (icm.cc:568-595) when j >= W-1, else the partial-window rule
// (icm.cc:818-835): stop as soon as the context position named by the node is before the buffer.
__device__ int dev_descend(const GmgDevModel &m, const DevBuf &b, int j, int f)
{
    const int8_t *mip = m.mip + (size_t)f * m.N;
    const int start = j - (m.W - 1);
    int node = 0;
    if (start >= 0) {
        for (int i = 0; i < m.D; i++) {
            int pos = mip[node];
            if (pos == -1) break;
            if (pos < -1) { node = dev_parent(node); break; }
            node = 4 * node + b.at(start + pos) + 1;
        }
        if (mip[node] < -1) node = dev_parent(node);
    } else {
        for (int i = 0; i < m.D; i++) {
            int q = start + mip[node];
            if (q < 0) break;
            node = 4 * node + b.at(q) + 1;
        }
        if (mip[node] == -2) node = dev_parent(node);
    }
    return node;
}

__device__ __forceinline__ float dev_score(const GmgDevModel &m, const DevBuf &b, int j, int f)
{
    int node = dev_descend(m, b, j, f);
    return m.prob[4 * ((size_t)f * m.N + node) + b.at(j)];
}

// ---------------------------------------------------------------------------
// tile -> read table
// ---------------------------------------------------------------------------

__global__ void k_tile_read(const uint64_t *__restrict__ off, uint64_t n_reads, uint64_t n_tiles,
                            uint32_t *__restrict__ tile_read)
{
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tiles) return;
    uint64_t g = t * GMG_TILE;
    // largest r in [0, n_reads] with off[r] <= g; entry n_tiles may be n_reads
    uint64_t lo = 0, hi = n_reads;
    while (lo < hi) {
        uint64_t mid = (lo + hi + 1) >> 1;
        if (off[mid] <= g) lo = mid; else hi = mid - 1;
    }
    tile_read[t] = (uint32_t)lo;
}

int gmg_launch_tile_read(const uint64_t *d_off, uint64_t n_reads, uint64_t n_tiles, uint32_t *d_tile_read,
                         hipStream_t s)
{
    uint64_t n = n_tiles + 1;
    unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_tile_read, dim3(grid), dim3(256), 0, s, d_off, n_reads, n_tiles, d_tile_read);
    GMG_HIP(hipGetLastError());
    return GMG_OK;
}

// ---------------------------------------------------------------------------
// k_frame6: six-frame per-position scores of whole reads
// ---------------------------------------------------------------------------
//
// Output row f (0..2):   reversed read scored with sub-model f, stored at forward coordinates:
//     window chars w[k] = S[p+W-1-k], predicted base S[p]        (glimmer-mg.cc:1482-1494)
// Output row 3+f:        complemented read:
//     window chars w[k] = comp(S[p-(W-1)+k]), predicted comp(S[p]) (glimmer-mg.cc:1497-1509)
// Each lane owns one base p of one read and produces its six doubles, so every store instruction
// writes 64 consecutive doubles of one row (512 B, coalesced).
//
// Context registers: CF / CR hold window char k in bits [2k, 2k+1]; the completed-tree shift table
// (2*mip, one byte per node, levels 0..D-1) sits in LDS, so one descent step is
//     ds_read_u8 ; v_bfe_u32 ; v_lshl_add_u32
// and the leaf row is one 4-byte gather from the L2-resident cleaf table.
// Positions whose window leaves the read (the first W-1 bases of either scoring buffer) take the
// exact partial-window descent on the original tables.

struct Frame6Args {
    GmgDevModel gene, nul;
    const uint32_t *packed;
    const uint64_t *off;
    const uint32_t *tile_read;
    uint64_t n_reads, total, n_words;
    double *out;
    int gene_fast, nul_dense;
    int uniform_len;
};

// all (W-1+1+W-1) bases around job-wide base g, base g-(W-1)+i at bits [2i,2i+1]
__device__ __forceinline__ uint64_t dev_window_bits(const uint32_t *__restrict__ packed, uint64_t n_words,
                                                    int64_t first)
{
    // `first` may be negative or run past the data for lanes whose window leaves the read; those
    // lanes never use the bits, the clamps only keep the loads inside the buffer.
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

__device__ __forceinline__ uint32_t dev_reverse_fields(uint32_t y, int nfields)
{
    // reverse the order of `nfields` 2-bit fields held in the low bits of y
    uint32_t z = __brev(y) >> (32 - 2 * nfields);
    return ((z & 0x55555555u) << 1) | ((z >> 1) & 0x55555555u);
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_frame6(Frame6Args a)
{
    extern __shared__ uint8_t lds[];
    // LDS: gene completed-tree shift table for all sub-models, then the null dense table
    uint8_t *s_shift = lds;
    const int P = a.gene.P;
    const int cstride = a.gene.cstride;
    const int shift_bytes = a.gene_fast ? P * cstride : 0;
    float *s_dense = (float *)(lds + ((shift_bytes + 15) & ~15));
    const int n_dense = a.nul_dense ? (1 << (2 * a.nul.W)) : 0;

    for (int i = threadIdx.x * 16; i < shift_bytes; i += BLOCK * 16)
        *(uint4 *)(s_shift + i) = *(const uint4 *)(a.gene.cshift + i);
    for (int i = threadIdx.x; i < a.nul.P * n_dense; i += BLOCK) s_dense[i] = a.nul.dense[i];
    __syncthreads();

    const int W = a.gene.W, D = a.gene.D, Wn = a.nul.W;
    const uint32_t ctx_mask = (W >= 16) ? 0xffffffffu : ((1u << (2 * W)) - 1u);
    const size_t n_leaf = (size_t)1 << (2 * D);
    const uint64_t n_chunks = (a.total + BLOCK - 1) / BLOCK;

    for (uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const uint64_t g = chunk * BLOCK + threadIdx.x;
        if (g >= a.total) continue;

        // ---- which read, where in it
        uint64_t r, r_off, r_end;
        if (a.uniform_len > 0) {
            r = g / (uint64_t)a.uniform_len;
            r_off = r * (uint64_t)a.uniform_len;
            r_end = r_off + (uint64_t)a.uniform_len;
        } else {
            r = a.tile_read[g / GMG_TILE];
            r_end = a.off[r + 1];
            while (g >= r_end) { r++; r_end = a.off[r + 1]; }
            r_off = a.off[r];
        }
        const int L = (int)(r_end - r_off);
        const int p = (int)(g - r_off);
        const int jf = L - 1 - p;   // index of this base in the reversed buffer
        const int jr = p;           // index in the complemented buffer

        // ---- context registers
        const uint64_t x = dev_window_bits(a.packed, a.n_words, (int64_t)g - (W - 1));
        // fields 0..W-1 of x = S[p-(W-1)..p], fields W-1..2W-2 = S[p..p+W-1]
        const uint32_t CR = ((uint32_t)x & ctx_mask) ^ ctx_mask;                      // comp(S[p-(W-1)+k])
        const uint32_t CF = dev_reverse_fields((uint32_t)(x >> (2 * (W - 1))) & ctx_mask, W);   // S[p+W-1-k]

        DevBuf bf = dev_make_buf(a.packed, r_off, 0, (uint32_t)L, GMG_REVERSED);
        DevBuf br = dev_make_buf(a.packed, r_off, 0, (uint32_t)L, GMG_COMPLEMENTED);

#pragma unroll
        for (int strand = 0; strand < 2; strand++) {
            const uint32_t C = strand ? CR : CF;
            const int j = strand ? jr : jf;
            const DevBuf &b = strand ? br : bf;
            const bool gene_full = a.gene_fast && (j >= W - 1);
            const bool nul_full = a.nul_dense && (j >= Wn - 1);
            const uint32_t pred = (C >> (2 * (W - 1))) & 3u;
            // null window = last Wn chars of the gene window
            const uint32_t nidx = (Wn <= W) ? (C >> (2 * (W - Wn))) : 0;
#pragma unroll
            for (int f = 0; f < 3; f++) {
                const int fg = f, fn = f;
                float gv, nv;
                if (gene_full) {
                    const uint8_t *tab = s_shift + fg * cstride;
                    uint32_t idx = 0, lvl = 0, width = 1;
                    for (int l = 0; l < D; l++) {
                        uint32_t sh = tab[lvl + idx];
                        idx = (idx << 2) + ((C >> sh) & 3u);
                        lvl += width;
                        width <<= 2;
                    }
                    gv = a.gene.cleaf[((size_t)fg * n_leaf + idx) * 4 + pred];
                } else {
                    gv = dev_score(a.gene, b, j, fg);
                }
                if (nul_full && Wn <= W) {
                    nv = s_dense[fn * n_dense + nidx];
                } else {
                    nv = dev_score(a.nul, b, j, fn);
                }
                // glimmer-mg.cc:1493,1508: double(gene) - double(null)
                a.out[(uint64_t)(strand * 3 + f) * a.total + g] = (double)gv - (double)nv;
            }
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
    a.n_reads = reads->n_reads;
    a.total = reads->total_bases;
    a.n_words = reads->n_words;
    a.out = d_out;
    a.gene_fast = gene->dev.has_fast;
    a.nul_dense = nul->dev.has_dense && nul->dev.W <= gene->dev.W;
    a.uniform_len = reads->uniform_len;

    constexpr int BLOCK = 256;
    size_t lds = 0;
    if (a.gene_fast) lds += ((size_t)a.gene.P * a.gene.cstride + 15) & ~(size_t)15;
    if (a.nul_dense) lds += (size_t)a.nul.P * ((size_t)1 << (2 * a.nul.W)) * 4;
    if (lds > 150 * 1024) {   // absurdly deep fast table: keep correctness via the generic descent
        a.gene_fast = 0;
        lds = a.nul_dense ? (size_t)a.nul.P * ((size_t)1 << (2 * a.nul.W)) * 4 : 0;
    }
    if (lds > 64 * 1024)
        GMG_HIP(hipFuncSetAttribute((const void *)k_frame6<BLOCK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    uint64_t n_chunks = (a.total + BLOCK - 1) / BLOCK;
    unsigned grid = (unsigned)(n_chunks < 256 * 8 ? n_chunks : 256 * 8);
    hipLaunchKernelGGL(k_frame6<BLOCK>, dim3(grid), dim3(BLOCK), lds, s, a);
    GMG_HIP(hipGetLastError());
    return GMG_OK;
}

// ---------------------------------------------------------------------------
// segment kernels (generic descent, any model shape)
// ---------------------------------------------------------------------------

struct SegArgs {
    GmgDevModel m;
    const uint32_t *packed;
    const uint64_t *off;
    const gmg_segment *segs;
    const uint64_t *out_off;
    uint64_t n_segs, total_len;
};

// one lane per (segment, position): Frame_Score (icm.cc:485-509)
__global__ __launch_bounds__(256) void k_seg_frame(SegArgs a, int frame, double *__restrict__ out)
{
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < a.total_len;
         e += (uint64_t)gridDim.x * blockDim.x) {
        // segment containing output element e: largest i with out_off[i] <= e
        uint64_t lo = 0, hi = a.n_segs - 1;
        while (lo < hi) {
            uint64_t mid = (lo + hi + 1) >> 1;
            if (a.out_off[mid] <= e) lo = mid; else hi = mid - 1;
        }
        // skip zero-length segments that share the same offset
        while (a.out_off[lo + 1] <= e) lo++;
        const gmg_segment sg = a.segs[lo];
        DevBuf b = dev_make_buf(a.packed, a.off[sg.read], sg.lo, sg.len, sg.orient);
        int j = (int)(e - a.out_off[lo]);
        out[e] = (double)dev_score(a.m, b, j, frame);
    }
}

// one lane per segment, sequential double adds in reference order:
// Cumulative_Score (icm.cc:374-402) when out != NULL, Score_String (icm.cc:882-902) into sums.
__global__ __launch_bounds__(256) void k_seg_cum(SegArgs a, int frame0, double *__restrict__ out,
                                                 double *__restrict__ sums)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n_segs;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const gmg_segment sg = a.segs[i];
        DevBuf b = dev_make_buf(a.packed, a.off[sg.read], sg.lo, sg.len, sg.orient);
        const uint64_t o = a.out_off[i];
        double result = 0.0;
        int f = frame0;
        for (int j = 0; j < (int)sg.len; j++) {
            result += (double)dev_score(a.m, b, j, f);
            f = (f == a.m.P - 1) ? 0 : f + 1;
            if (out) out[o + j] = result;
        }
        if (sums) sums[i] = result;
    }
}

// Partial_Window_Prob (icm.cc:807-842) of the last base of each segment; the partial rule is
// used whatever the segment length.
__global__ __launch_bounds__(256) void k_seg_partial(SegArgs a, int frame, double *__restrict__ out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n_segs;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const gmg_segment sg = a.segs[i];
        if (sg.len == 0) { out[i] = 0.0; continue; }
        DevBuf b = dev_make_buf(a.packed, a.off[sg.read], sg.lo, sg.len, sg.orient);
        const int8_t *mip = a.m.mip + (size_t)frame * a.m.N;
        const int j = (int)sg.len - 1;
        const int start = j - (a.m.W - 1);
        int node = 0;
        for (int l = 0; l < a.m.D; l++) {
            int q = start + mip[node];
            if (q < 0) break;
            node = 4 * node + b.at(q) + 1;
        }
        if (mip[node] == -2) node = dev_parent(node);
        out[i] = (double)a.m.prob[4 * ((size_t)frame * a.m.N + node) + b.at(j)];
    }
}

// All_Frame_Score (glimmer3.cc:328-359): lane k of a segment computes one of the six
// Score_String values; Permute_By_Frame (glimmer3.cc:1013-1088) picks the output slot.
__global__ __launch_bounds__(256) void k_all_frame(SegArgs a, const uint32_t *__restrict__ prefix,
                                                   const int32_t *__restrict__ frame, double *__restrict__ af)
{
    // raw order of glimmer3.cc:346-354: {s,1} {s,2} {s,0} {rc,1} {rc,0} {rc,2}
    const int raw_frame[6] = {1, 2, 0, 1, 0, 2};
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < a.n_segs * 6;
         e += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = e / 6;
        const int k = (int)(e % 6);
        gmg_segment sg = a.segs[i];
        uint32_t n = prefix[i] < sg.len ? prefix[i] : sg.len;
        // the first n buffer bases as a segment of their own
        const bool rev = (sg.orient == GMG_REVERSED || sg.orient == GMG_REVCOMP);
        uint32_t lo = rev ? sg.lo + sg.len - n : sg.lo;
        uint32_t orient = sg.orient;
        if (k >= 3) {
            // reverse complement of that prefix (glimmer_base.cc:2484-2501): flips both properties
            orient = (sg.orient == GMG_FORWARD) ? GMG_REVCOMP
                   : (sg.orient == GMG_REVCOMP) ? GMG_FORWARD
                   : (sg.orient == GMG_REVERSED) ? GMG_COMPLEMENTED : GMG_REVERSED;
        }
        DevBuf b = dev_make_buf(a.packed, a.off[sg.read], lo, n, orient);
        int f = (a.m.P == 1) ? 0 : raw_frame[k];
        double result = 0.0;
        for (int j = 0; j < (int)n; j++) {
            result += (double)dev_score(a.m, b, j, f);
            f = (f + 1) % a.m.P;
        }
        // out[i] = raw[perm[i]]  <=>  raw k lands in slot inv[k]
        int slot = k;
        switch (frame[i]) {
        case 1:  { const int inv[6] = {1, 2, 0, 4, 5, 3}; slot = inv[k]; break; }
        case 2:  { const int inv[6] = {2, 0, 1, 5, 3, 4}; slot = inv[k]; break; }
        case -1: { const int inv[6] = {3, 5, 4, 0, 2, 1}; slot = inv[k]; break; }
        case -2: { const int inv[6] = {4, 3, 5, 1, 0, 2}; slot = inv[k]; break; }
        case -3: { const int inv[6] = {5, 4, 3, 2, 1, 0}; slot = inv[k]; break; }
        default: break;
        }
        af[6 * i + slot] = result;
    }
}

// Full_Window_Prob / Full_Window_Distrib on explicit windows (icm.cc:512-610)
__global__ __launch_bounds__(256) void k_windows(GmgDevModel m, const uint8_t *__restrict__ win,
                                                 const int32_t *__restrict__ frames, uint64_t n,
                                                 float *__restrict__ dist4, double *__restrict__ prob)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint8_t *w = win + i * m.W;
        int f = frames[i];
        if (f < 0 || f >= m.P) f = 0;
        const int8_t *mip = m.mip + (size_t)f * m.N;
        int node = 0;
        for (int l = 0; l < m.D; l++) {
            int pos = mip[node];
            if (pos == -1) break;
            if (pos < -1) { node = dev_parent(node); break; }
            node = 4 * node + (w[pos] & 3) + 1;
        }
        if (mip[node] < -1) node = dev_parent(node);
        const float *row = m.prob + 4 * ((size_t)f * m.N + node);
        if (dist4) {
            dist4[4 * i + 0] = row[0]; dist4[4 * i + 1] = row[1];
            dist4[4 * i + 2] = row[2]; dist4[4 * i + 3] = row[3];
        }
        if (prob) prob[i] = (double)row[w[m.W - 1] & 3];
    }
}

static unsigned grid_for(uint64_t n, unsigned block)
{
    uint64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    return (unsigned)(g > 256u * 16u ? 256u * 16u : g);
}

static SegArgs make_seg_args(const gmg_model *m, const gmg_reads *r, const gmg_segments *sg)
{
    SegArgs a;
    a.m = m->dev;
    a.packed = r->d_packed;
    a.off = r->d_off;
    a.segs = sg->d_segs;
    a.out_off = sg->d_out_off;
    a.n_segs = sg->n;
    a.total_len = sg->total_len;
    return a;
}

int gmg_launch_seg_frame(const gmg_model *m, const gmg_reads *r, const gmg_segments *sg, int frame,
                         double *d_out, hipStream_t s)
{
    SegArgs a = make_seg_args(m, r, sg);
    hipLaunchKernelGGL(k_seg_frame, dim3(grid_for(a.total_len, 256)), dim3(256), 0, s, a, frame, d_out);
    GMG_HIP(hipGetLastError());
    return GMG_OK;
}

int gmg_launch_seg_cum(const gmg_model *m, const gmg_reads *r, const gmg_segments *sg, int frame0,
                       double *d_out, double *d_sums, hipStream_t s)
{
    SegArgs a = make_seg_args(m, r, sg);
    hipLaunchKernelGGL(k_seg_cum, dim3(grid_for(a.n_segs, 256)), dim3(256), 0, s, a, frame0, d_out, d_sums);
    GMG_HIP(hipGetLastError());
    return GMG_OK;
}

int gmg_launch_seg_partial(const gmg_model *m, const gmg_reads *r, const gmg_segments *sg, int frame,
                           double *d_out, hipStream_t s)
{
    SegArgs a = make_seg_args(m, r, sg);
    hipLaunchKernelGGL(k_seg_partial, dim3(grid_for(a.n_segs, 256)), dim3(256), 0, s, a, frame, d_out);
    GMG_HIP(hipGetLastError());
    return GMG_OK;
}

int gmg_launch_all_frame(const gmg_model *m, const gmg_reads *r, const gmg_segments *sg,
                         const uint32_t *d_prefix, const int32_t *d_frame, double *d_af, hipStream_t s)
{
    SegArgs a = make_seg_args(m, r, sg);
    hipLaunchKernelGGL(k_all_frame, dim3(grid_for(a.n_segs * 6, 256)), dim3(256), 0, s, a, d_prefix, d_frame, d_af);
    GMG_HIP(hipGetLastError());
    return GMG_OK;
}

int gmg_launch_windows(const gmg_model *m, const uint8_t *d_windows, const int32_t *d_frames, uint64_t n,
                       float *d_dist4, double *d_prob, hipStream_t s)
{
    hipLaunchKernelGGL(k_windows, dim3(grid_for(n, 256)), dim3(256), 0, s, m->dev, d_windows, d_frames, n,
                       d_dist4, d_prob);
    GMG_HIP(hipGetLastError());
    return GMG_OK;
}
