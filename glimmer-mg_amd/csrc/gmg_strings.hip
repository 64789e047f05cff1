// gmg_strings.hip -- whole-read Score_String of many ICMs on both strands: what Phymm's scoreReadsGlim.pl asks of
// simple-score (scripts/scoreReadsGlim.pl:450,482: every read and its reverse complement against every genome's ICM;
// ICM_t::Score_String, src/ICM/icm.cc:864-903, frame 0, periodicity-1 models).  BASELINE configs[3].
//
// Per model (option strings_fused, the default):
//   k_frame6t<STRINGS, SUM> (gmg_frame6.hip)  the six-frame kernel's main pass on both strings, the values added per read
//                                        inside the kernel (wave shuffle tree -> LDS accumulators per round -> one atomic add
//                                        per read and round): nothing per base leaves the chip (0.25 B/base in, 16 B/read out)
//   k_string_heads                       the first W-1 positions of every string by the partial-window rule (icm.cc:807-842)
//   k_string_finish                      per (read, string): main sum + heads, and the proof that the ORDER of the additions
//                                        could not matter: every value of the model is <= 0 and a multiple of 2^(e_min - 150),
//                                        so while |sum| < 2^(e_min - 150 + 51) every partial sum of every order is an exact
//                                        double -- the result IS the reference's sequential sum (icm.cc:871-900).  Reads that
//                                        fail the test (or touch the last, partial chunk of the batch) are listed ...
//   k_string_exact                       ... and recomputed one lane per string in the reference's order (plain descent).
//   Models with a positive, denormal or non-finite value, or values so small that ordinary reads would fail the test, take
//   the two-pass form below from the start.
// Two-pass form (strings_fused = 0; 4 GB of fp32 values out and back per model and 1M x 500 bp):
//   k_frame6t<STRINGS>                   per-base values of both strings, fp32 rows [2][total]
//   k_string_sum                         one lane per (read, strand): heads first, then the rows streamed through LDS in
//                                        contiguous slabs; ONE running sum of doubles in string order.
// Models the fast pass does not cover (periodicity 3, other depths) go through k_seg_cum (exact, any shape).

#include "gmg_device.h"

#include <string.h>
#include <vector>

struct StringSumArgs {
    GmgDevModel m;
    const uint32_t *packed;
    const uint64_t *read_off;
    uint64_t n_reads, total, tail_start;
    const float *vals;           // [2][total]
    float *heads;                // [2][n_reads][16]: the first W-1 values of every string (k_string_heads)
    double *sums;                // [n_reads][2]: forward string, reverse complement
};

struct __attribute__((packed, aligned(4))) StrF4 { float v[4]; };

// The first W-1 positions of every string by the partial-window rule (icm.cc:883-888, 807-842), one lane per
// (strand, read, position), on the completed tree (shift bytes in LDS; "stop when the context position lies before the
// string" reads "shift byte < 2 ((W-1) - j)"; crow has the row for inner nodes too).  Its own kernel: inside k_string_sum
// the dependent global reads of this step stalled every wave for several microseconds and its 5.5 KB of LDS cost a
// third of the waves per CU (2.7 -> 1.x ms per model for the summing kernel).
__global__ __launch_bounds__(256) void k_string_heads(StringSumArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_shift[5472];       // completed-tree shifts of sub-model 0, depth 7
    const int W = a.m.W, D = a.m.D;
    for (int i = threadIdx.x * 16; i < a.m.cstride && i < 5472; i += 256 * 16) *(uint4 *)(s_shift + i) = *(const uint4 *)(a.m.cshift + i);
    __syncthreads();
    const uint32_t ctx_mask = (1u << (2 * W)) - 1u;
    const uint64_t per_strand = a.n_reads * (uint64_t)(W - 1);
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < 2 * per_strand; idx += (uint64_t)gridDim.x * blockDim.x) {
        const bool rc = idx >= per_strand;
        const uint64_t rem = rc ? idx - per_strand : idx;
        const uint64_t r = rem / (uint32_t)(W - 1);
        const int j = (int)(rem - r * (uint32_t)(W - 1));
        const uint64_t off_k = a.read_off[r];
        const int n_k = (int)(a.read_off[r + 1] - off_k);
        if (j >= n_k) continue;
        uint32_t C;                                     // window ending at string position j: char i at bits 2i
        if (!rc) C = (uint32_t)dev_window_bits(a.packed, (int64_t)off_k + j - (W - 1)) & ctx_mask;
        else C = (dev_reverse_fields((uint32_t)dev_window_bits(a.packed, (int64_t)off_k + n_k - 1 - j) & ctx_mask, W) ^ ctx_mask);
        const int thr2 = 2 * ((W - 1) - j);
        uint32_t tidx = 0, lvl = 0, width = 1, node = 0xffffffffu;
        for (int l = 0; l < D; l++) {
            const uint32_t sh = s_shift[lvl + tidx];
            if (node == 0xffffffffu && (int)sh < thr2) node = lvl + tidx;
            tidx = (tidx << 2) + ((C >> sh) & 3u);
            lvl += width;
            width <<= 2;
        }
        if (node == 0xffffffffu) node = lvl + tidx;
        a.heads[((rc ? a.n_reads : 0) + r) * 16 + j] = a.m.crow[(size_t)node * 4 + ((C >> (2 * (W - 1))) & 3u)];
    }
}

// One work-group = one wave = STR_READS consecutive reads of one strand.  Their values are ONE contiguous run of the
// row, so the wave streams it through LDS in slabs of STR_SLAB floats with fully coalesced 16-byte loads (memory order =
// load order: this is what keeps HBM pages open; 64-byte pieces per read ran at 1.1 TB/s), and the lane that owns a
// read adds the part of its read inside the slab, in string order, carrying its sum from slab to slab.  Forward
// strings walk the slabs upwards, reverse complements downwards.  Reads of any length work (a long read simply spans
// several slabs).
#ifndef STR_READS
#define STR_READS 16
#endif
#ifndef STR_SLAB
#define STR_SLAB 4096
#endif

__global__ __launch_bounds__(64) void k_string_sum(StringSumArgs a)
{
    __shared__ __attribute__((aligned(16))) float s_v[STR_SLAB];
    const uint32_t lane = threadIdx.x;
    const uint64_t groups = (a.n_reads + STR_READS - 1) / STR_READS;
    const int W = a.m.W;
    for (uint64_t blk = blockIdx.x; blk < 2 * groups; blk += gridDim.x) {
        const bool rc = blk >= groups;                  // reverse complement: string position q <-> read position n-1-q
        const uint64_t r0 = (rc ? blk - groups : blk) * STR_READS;
        const uint64_t r1 = r0 + STR_READS < a.n_reads ? r0 + STR_READS : a.n_reads;
        const uint64_t g_lo = a.read_off[r0], g_hi = a.read_off[r1];       // the group's run of the row
        const uint64_t r = r0 + lane;
        const bool live = lane < STR_READS && r < r1;
        const uint64_t off = live ? a.read_off[r] : 0;
        const int n = live ? (int)(a.read_off[r + 1] - off) : 0;
        const DevBuf b = dev_make_buf(a.packed, off, 0, (uint32_t)n, rc ? GMG_REVCOMP : GMG_FORWARD);
        const int head = n < W - 1 ? n : W - 1;
        double sum = 0.0;                               // icm.cc:871
        if (live) {                                     // the partial windows first (k_string_heads), in string order
            const float *hd = a.heads + ((rc ? a.n_reads : 0) + r) * 16;
            for (int q = 0; q < head; q++) sum += (double)hd[q];
        }
        int q = head;                                   // next string position of this lane's read
        const float *row = a.vals + (rc ? a.total : 0);
        const uint64_t n_slabs = (g_hi - g_lo + STR_SLAB - 1) / STR_SLAB;
        // slab [s_lo, s_hi) of the row, from the bottom (forward) or from the top (reverse complement)
        auto slab_lo = [&](uint64_t sl) __attribute__((always_inline)) {
            return rc ? (g_hi > (sl + 1) * STR_SLAB + g_lo ? g_hi - (sl + 1) * STR_SLAB : g_lo) : g_lo + sl * STR_SLAB;
        };
        auto slab_hi = [&](uint64_t sl) __attribute__((always_inline)) {
            return rc ? g_hi - sl * STR_SLAB : (g_lo + (sl + 1) * STR_SLAB < g_hi ? g_lo + (sl + 1) * STR_SLAB : g_hi);
        };
        StrF4 x[STR_SLAB / 256];
        auto issue = [&](uint64_t sl) __attribute__((always_inline)) {
            const uint64_t lo = slab_lo(sl), hi = slab_hi(sl);
#pragma unroll
            for (int i = 0; i < STR_SLAB / 256; i++) {
                const uint64_t g = lo + 4 * (lane + 64u * i);
                StrF4 t = {{0.f, 0.f, 0.f, 0.f}};
                if (g + 3 < hi) t = *(const StrF4 *)(row + g);
                else
                    for (int e = 0; e < 4; e++)
                        if (g + e < hi) t.v[e] = row[g + e];
                x[i] = t;
            }
        };
        if (n_slabs) issue(0);
        for (uint64_t sl = 0; sl < n_slabs; sl++) {
            const uint64_t s_lo = slab_lo(sl), s_hi = slab_hi(sl);
            __syncthreads();                            // the previous slab has been summed
#pragma unroll
            for (int i = 0; i < STR_SLAB / 256; i++) *(StrF4 *)(s_v + 4 * (lane + 64u * i)) = x[i];
            if (sl + 1 < n_slabs) issue(sl + 1);        // in flight while this slab is summed
            __syncthreads();
            if (live && q < n) {
                // string positions q .. q_end-1 of this read have their base inside the slab
                int q_end;
                if (!rc) q_end = off + (uint64_t)n <= s_hi ? n : (int)(s_hi - off);
                else q_end = off >= s_lo ? n : n - (int)(s_lo - off);
                const uint64_t g_first = off + (uint64_t)(rc ? n - 1 - q : q);
                if (q_end > q && g_first >= s_lo && g_first < s_hi) {
                    const uint64_t g_last = off + (uint64_t)(rc ? n - q_end : q_end - 1);
                    const uint64_t g_top = rc ? g_first : g_last;
                    if (g_top < a.tail_start) {         // the usual case: a tight loop of LDS reads and additions
                        const float *v = s_v + (g_first - s_lo);
                        const int cnt = q_end - q;
                        const int dir = rc ? -1 : 1;
                        int e = 0;
                        for (; e + 16 <= cnt; e += 16) {            // 16 LDS reads in flight, then the additions in order
                            float t[16];
#pragma unroll
                            for (int u = 0; u < 16; u++) t[u] = v[dir * (e + u)];
#pragma unroll
                            for (int u = 0; u < 16; u++) sum += (double)t[u];
                        }
                        for (; e < cnt; e++) sum += (double)v[dir * e];
                    } else {
                        // the last < 2,048 bases of the batch are not in the rows (the main pass does whole chunks): plain descent
                        for (int qq = q; qq < q_end; qq++) {
                            const uint64_t g = off + (uint64_t)(rc ? n - 1 - qq : qq);
                            sum += g < a.tail_start ? (double)s_v[g - s_lo] : (double)dev_score(a.m, b, qq, 0);
                        }
                    }
                    q = q_end;
                }
            }
        }
        if (live) a.sums[2 * r + (rc ? 1 : 0)] = sum;
    }
}

// Fused form: main sums (k_frame6t<.., SUM>) + heads -> final sums, or onto the list of strings to recompute
struct StringFinishArgs {
    StringSumArgs s;
    int min_exp;                 // smallest exponent field among the model's non-zero values
    uint32_t *n_redo;            // strings to recompute ...
    uint32_t *redo;              // ... 2 * read + string
};

__global__ __launch_bounds__(256) void k_string_finish(StringFinishArgs a)
{
    const int W = a.s.m.W;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * a.s.n_reads; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i >> 1;
        const bool rc = i & 1;
        const uint64_t off = a.s.read_off[r];
        const int n = (int)(a.s.read_off[r + 1] - off);
        double sum = a.s.sums[i];                        // the positions with a full window, added in no particular order
        const int head = n < W - 1 ? n : W - 1;
        const float *hd = a.s.heads + ((rc ? a.s.n_reads : 0) + r) * 16;
        for (int q = 0; q < head; q++) sum += (double)hd[q];
        // every partial sum of every order has |x| <= |sum| (one sign) and is a multiple of 2^(min_exp - 150): exact doubles
        // while ilogb |sum| + 2 - (min_exp - 150) <= 53 (one bit of margin for a sum that was itself rounded)
        const unsigned long long bits = (unsigned long long)__double_as_longlong(sum);
        const int e = (int)((bits >> 52) & 0x7ffull) - 1023;
        const bool exact = sum == 0.0 || e + 2 - (a.min_exp - 150) <= 53;
        const bool complete = off + (uint64_t)n <= a.s.tail_start;       // no base in the batch's last, partial chunk
        if (exact && complete) a.s.sums[i] = sum;
        else a.redo[atomicAdd(a.n_redo, 1u)] = (uint32_t)i;
    }
}

// the listed strings once more, the reference's way: plain descent, additions in string order.  One wave per string: 64
// positions' values at a time (the descents are chains of dependent loads: one lane alone would take a millisecond per
// read), then lane 0 adds them in order.
__global__ __launch_bounds__(64) void k_string_exact(StringFinishArgs a)
{
    __shared__ float s_v[64];
    const uint32_t n_redo = *a.n_redo;
    for (uint32_t k = blockIdx.x; k < n_redo; k += gridDim.x) {
        const uint32_t i = a.redo[k];
        const uint64_t r = i >> 1;
        const uint64_t off = a.s.read_off[r];
        const int n = (int)(a.s.read_off[r + 1] - off);
        const DevBuf b = dev_make_buf(a.s.packed, off, 0, (uint32_t)n, (i & 1) ? GMG_REVCOMP : GMG_FORWARD);
        double sum = 0.0;
        for (int q0 = 0; q0 < n; q0 += 64) {
            const int q = q0 + (int)threadIdx.x;
            __syncthreads();
            s_v[threadIdx.x] = q < n ? dev_score(a.s.m, b, q, 0) : 0.0f;
            __syncthreads();
            if (threadIdx.x == 0) {
                const int cnt = n - q0 < 64 ? n - q0 : 64;
                for (int e = 0; e < cnt; e++) sum += (double)s_v[e];
            }
        }
        if (threadIdx.x == 0) a.s.sums[i] = sum;
    }
}

extern "C" int gmg_score_reads_strings(const gmg_model *const *models, int n_models, const gmg_reads *reads,
                                       double *d_sums, void *stream)
{
    { int rc_enter = gmg_enter("gmg_score_reads_strings"); if (rc_enter) return rc_enter; }
    if (!models || n_models < 0 || !reads || (!d_sums && n_models && reads->n_reads))
        return gmg_set_error(GMG_EINVAL, "gmg_score_reads_strings: NULL argument");
    hipStream_t s = (hipStream_t)stream;
    const uint64_t nr = reads->n_reads;
    if (nr == 0 || n_models == 0) return GMG_OK;
    float *d_vals = nullptr, *d_heads = nullptr;
    uint32_t *d_redo = nullptr;                         // [1 + 2 nr]: counter, list
    gmg_segments *segs = nullptr;                       // built on demand for models without the fast pass
    int rc = GMG_OK;
    for (int k = 0; k < n_models && rc == GMG_OK; k++) {
        const gmg_model *m = models[k];
        if (!m) { rc = gmg_set_error(GMG_EINVAL, "gmg_score_reads_strings: model %d is NULL", k); break; }
        double *out = d_sums + (size_t)k * nr * 2;
        uint64_t tail_start = 0;
        // the fused form: values that are all <= 0 and not too small (ordinary reads then pass k_string_finish's test)
        // ... and reads of at least 86 bases: at most 384 of them then overlap a round of 32,768 bases (the accumulators in LDS) and
        // at most three a wave's 128 bases
        // ... and a bounded exponent range: the uniform-length path moves the share of the read that ends inside a row of 16 lanes
        // (<= 32 values) through the NEXT read's accumulator (+ row sum, - share; f6_sum_chunk).  That is exact only while such a
        // share cannot push the neighbour's partial sums past 53 bits: 32 values < 2^(max_exp - 121), the read's own sum passes
        // k_string_finish's test below 2^(min_exp - 99), so max_exp - min_exp <= 23 keeps every transient an exact double.  A model
        // with a zero probability (-FLT_MAX, exponent field 254; icm.cc:1345-1349) therefore takes the two-pass form.
        if (gmg_opt(GMG_OPT_STRINGS_FUSED) && !m->odd_values && m->min_exp >= 109 && m->max_exp - m->min_exp <= 23 &&
            reads->total_bases && reads->min_len >= 86) {
            hipError_t e = hipMemsetAsync(out, 0, nr * 2 * sizeof(double), s);
            if (e != hipSuccess) { rc = gmg_set_error(GMG_EHIP, "gmg_score_reads_strings: %s", hipGetErrorString(e)); break; }
            const int fused = gmg_launch_strings_sum(m, reads, out, &tail_start, s);
            if (fused == GMG_OK) {
                if (!d_heads) e = gmg_pool_alloc((void **)&d_heads, (size_t)2 * nr * 16 * sizeof(float));
                if (e == hipSuccess && !d_redo) e = gmg_pool_alloc((void **)&d_redo, (1 + 2 * nr) * sizeof(uint32_t));
                if (e == hipSuccess) e = hipMemsetAsync(d_redo, 0, 4, s);
                if (e != hipSuccess) { rc = gmg_set_error(GMG_ENOMEM, "gmg_score_reads_strings: %s", hipGetErrorString(e)); break; }
                StringFinishArgs f;
                f.s.m = m->dev;
                f.s.packed = reads->d_packed;
                f.s.read_off = reads->d_off;
                f.s.n_reads = nr;
                f.s.total = reads->total_bases;
                f.s.tail_start = tail_start;
                f.s.vals = nullptr;
                f.s.heads = d_heads;
                f.s.sums = out;
                f.min_exp = m->min_exp;
                f.n_redo = d_redo;
                f.redo = d_redo + 1;
                const uint64_t items = 2 * nr * (uint64_t)(m->dev.W - 1), hb = (items + 255) / 256;
                hipLaunchKernelGGL(k_string_heads, dim3((unsigned)(hb < 256 * 64 ? (hb ? hb : 1) : 256 * 64)), dim3(256), 0, s, f.s);
                const uint64_t fb = (2 * nr + 255) / 256;
                hipLaunchKernelGGL(k_string_finish, dim3((unsigned)(fb < 256 * 32 ? fb : 256 * 32)), dim3(256), 0, s, f);
                hipLaunchKernelGGL(k_string_exact, dim3(256 * 4), dim3(64), 0, s, f);
                e = hipGetLastError();
                if (e != hipSuccess) rc = gmg_set_error(GMG_EHIP, "gmg_score_reads_strings: %s", hipGetErrorString(e));
                continue;
            }
            if (fused != GMG_EBADMODEL) { rc = fused; break; }
        }
        if (!d_vals && reads->total_bases) {
            hipError_t e = gmg_pool_alloc((void **)&d_vals, (size_t)2 * reads->total_bases * sizeof(float));
            if (e != hipSuccess) { rc = gmg_set_error(GMG_ENOMEM, "gmg_score_reads_strings: %s", hipGetErrorString(e)); break; }
        }
        const int fast = gmg_launch_strings(m, reads, d_vals, &tail_start, s);
        if (fast == GMG_OK) {
            StringSumArgs a;
            a.m = m->dev;
            a.packed = reads->d_packed;
            a.read_off = reads->d_off;
            a.n_reads = nr;
            a.total = reads->total_bases;
            a.tail_start = tail_start;
            a.vals = d_vals;
            a.sums = out;
            if (!d_heads) {
                hipError_t eh = gmg_pool_alloc((void **)&d_heads, (size_t)2 * nr * 16 * sizeof(float));
                if (eh != hipSuccess) { rc = gmg_set_error(GMG_ENOMEM, "gmg_score_reads_strings: %s", hipGetErrorString(eh)); break; }
            }
            a.heads = d_heads;
            {
                const uint64_t items = 2 * nr * (uint64_t)(m->dev.W - 1), hb = (items + 255) / 256;
                hipLaunchKernelGGL(k_string_heads, dim3((unsigned)(hb < 256 * 64 ? (hb ? hb : 1) : 256 * 64)), dim3(256), 0, s, a);
            }
            const uint64_t blocks = 2 * ((nr + STR_READS - 1) / STR_READS);
            hipLaunchKernelGGL(k_string_sum, dim3((unsigned)(blocks < 256 * 256 ? blocks : 256 * 256)), dim3(64), 0, s, a);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) rc = gmg_set_error(GMG_EHIP, "gmg_score_reads_strings: %s", hipGetErrorString(e));
            continue;
        }
        if (fast != GMG_EBADMODEL) { rc = fast; break; }
        // any other model shape: the exact segment kernel on (read, FORWARD) and (read, REVCOMP) segments
        if (!segs) {
            std::vector<uint64_t> off(nr + 1);
            hipError_t e = hipMemcpy(off.data(), reads->d_off, (nr + 1) * 8, hipMemcpyDeviceToHost);
            if (e != hipSuccess) { rc = gmg_set_error(GMG_EHIP, "gmg_score_reads_strings: %s", hipGetErrorString(e)); break; }
            std::vector<gmg_segment> sg(2 * nr);
            for (uint64_t r = 0; r < nr; r++) {
                const uint32_t len = (uint32_t)(off[r + 1] - off[r]);
                sg[2 * r] = {(uint32_t)r, 0, len, GMG_FORWARD};
                sg[2 * r + 1] = {(uint32_t)r, 0, len, GMG_REVCOMP};
            }
            rc = gmg_segments_upload(reads, sg.data(), 2 * nr, nullptr, nullptr, &segs);
            if (rc) break;
        }
        rc = gmg_score_string(m, reads, segs, 0, out, s);
    }
    hipError_t e = hipStreamSynchronize(s);            // the scratch goes back to the cache: nothing may still use it
    if (d_vals) gmg_pool_release(d_vals);
    if (d_heads) gmg_pool_release(d_heads);
    if (d_redo) gmg_pool_release(d_redo);
    if (segs) gmg_segments_free(segs);
    if (rc == GMG_OK && e != hipSuccess) rc = gmg_set_error(GMG_EHIP, "gmg_score_reads_strings: %s", hipGetErrorString(e));
    return rc;
}
