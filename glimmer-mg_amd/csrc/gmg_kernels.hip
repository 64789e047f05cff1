// gmg_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4).
//
// The work is table gather + integer index arithmetic (no MFMA): one base score is a
// data-dependent descent of a 4-ary tree followed by one fp32 lookup
// (src/ICM/icm.cc:557-610, 807-842).  Kernels:
//   k_tile_read       read containing the first base of every 1024-base tile
//   (k_frame6, the six-frame whole-read kernel, lives in gmg_frame6.hip)
//   k_seg_frame       ICM_t::Frame_Score on arbitrary segments          (icm.cc:485-509)
//   k_seg_cum         ICM_t::Cumulative_Score / Score_String, sequential double adds in
//                     reference order                                    (icm.cc:354-405, 864-903)
//   k_all_frame       All_Frame_Score incl. Permute_By_Frame            (glimmer3.cc:328-359,1013-1088)
//   k_windows         Full_Window_Prob / Full_Window_Distrib            (icm.cc:512-610)

#include "gmg_internal.h"

#include "gmg_device.h"

#include <stdlib.h>

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

// The same for models with a completed tree (gmg_internal.h: W <= 16, D <= 8): the shift bytes of all sub-models sit in LDS,
// the window is a register that takes one 2-bit code per base (the codes stream from a packed word held in a register), and
// the only global access per base is the gather of the row entry.  The plain kernel above needs ~15 dependent global loads
// per base (mip and packed word at every level); this one 1.
__global__ __launch_bounds__(256) void k_seg_cum_fast(SegArgs a, int frame0, double *__restrict__ out,
                                                      double *__restrict__ sums)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_shift[];       // [P][cstride]
    const int P = a.m.P, cstride = a.m.cstride, W = a.m.W, D = a.m.D;
    for (int i = threadIdx.x * 16; i < P * cstride; i += 256 * 16) *(uint4 *)(s_shift + i) = *(const uint4 *)(a.m.cshift + i);
    __syncthreads();
    const uint32_t sh_top = 2u * (uint32_t)(W - 1);
    const uint32_t ctot = (uint32_t)a.m.ctot;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n_segs;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const gmg_segment sg = a.segs[i];
        const bool rev = sg.orient == GMG_REVERSED || sg.orient == GMG_REVCOMP;
        const uint32_t comp = (sg.orient == GMG_COMPLEMENTED || sg.orient == GMG_REVCOMP) ? 3u : 0u;
        const int len = (int)sg.len;
        const int64_t dirg = rev ? -1 : 1;
        int64_t g = (int64_t)a.off[sg.read] + sg.lo + (rev ? len - 1 : 0);     // base of buffer position 0
        const uint64_t o = a.out_off[i];
        double result = 0.0;
        if (len > 0) {
            uint32_t w = a.packed[g >> 4];
            uint32_t C = 0;                             // char k of the window that ends at j in bits [2k, 2k+1]
            int f = frame0;
            for (int j = 0; j < len; j++) {
                const uint32_t code = ((w >> (2u * (unsigned)(g & 15))) & 3u) ^ comp;
                const int64_t g2 = g + dirg;
                if ((g ^ g2) >> 4) w = a.packed[g2 >> 4];               // word -1 / one past the end: the guard words of gmg_reads
                g = g2;
                C = (C >> 2) | (code << sh_top);
                const uint8_t *tab = s_shift + f * cstride;
                // full window (icm.cc:568-595) for j >= W-1; before that the partial rule (icm.cc:818-835): stop at the first
                // node whose context position lies in front of the buffer, i.e. whose shift byte is < 2 ((W-1) - j)
                const int thr2 = j >= W - 1 ? 0 : 2 * ((W - 1) - j);
                uint32_t idx = 0, lvl = 0, width = 1, node = 0xffffffffu;
                for (int l = 0; l < D; l++) {
                    const uint32_t sh = tab[lvl + idx];
                    if (node == 0xffffffffu && (int)sh < thr2) node = lvl + idx;
                    idx = (idx << 2) + ((C >> sh) & 3u);
                    lvl += width;
                    width <<= 2;
                }
                if (node == 0xffffffffu) node = lvl + idx;
                result += (double)a.m.crow[((size_t)f * ctot + node) * 4 + code];
                f = (f == P - 1) ? 0 : f + 1;
                if (out) out[o + j] = result;
            }
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
    if (m->dev.has_fast && m->dev.W <= 15 && m->dev.D <= 8 && (size_t)m->dev.P * m->dev.cstride <= 96 * 1024 && !gmg_opt(GMG_OPT_SEG_PLAIN)) {
        const size_t lds = (size_t)m->dev.P * m->dev.cstride;
        if (lds > 48 * 1024) GMG_HIP(hipFuncSetAttribute((const void *)k_seg_cum_fast, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_seg_cum_fast, dim3(grid_for(a.n_segs, 256)), dim3(256), lds, s, a, frame0, d_out, d_sums);
        GMG_HIP(hipGetLastError());
        return GMG_OK;
    }
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
