// gmg_ingest.hip -- FASTA bytes -> packed reads, on the device.
//
// What it replaces: Fasta_Read (src/Common/fasta.cc:236-286) called in a loop, the callers' per-base
// tolower (Filter (ch)) (src/Glimmer/glimmer3.cc:270-271, glimmer-mg.cc:381-382; Filter = src/Common/gene.cc:1139-1175)
// and the g/c count of Set_GC_Fraction (src/Glimmer/glimmer_base.cc:2564-2595) -- a byte-at-a-time fgetc loop on the
// host, ~1 GB/s, which at the kernels' rates is the wall.  Here the raw file bytes are copied to HBM once and parsed there.
// Round 4: TWO passes over the bytes with a summary per 4 KiB block between them (k_fa_summ, k_fa_blocks, k_fa_pack2) instead of two
// hipcub scans over every byte (1 + 8 bytes written per input byte) and a pack pass: 6.8 -> 0.6 ms per 521 MB on the device; the
// header extents come back through a page-locked buffer.  The machine, and the first version's scans (kept as the fallback
// `ingest_scans` = 1 and as the cross-check of tests/test_gpu_ingest.py):
//
//   Fasta_Read as a 3-state machine over the bytes (PRE = before the first '>', HDR = in a header line, SEQ = in the
//   sequence part):   '>' : PRE,SEQ,HDR -> HDR ('>' inside a header line is text; anywhere else it starts a record, also
//   in the middle of a line);   '\n' : HDR -> SEQ;   anything else keeps the state.  Every byte is a function
//   state -> state (2 bits per input state); function composition is associative, so ONE inclusive scan gives the
//   state behind every byte (hipcub::DeviceScan with the composition as operator).
//   A byte is a sequence character iff the state before it is SEQ, it is not '>' and not isspace().
//   A second scan counts records and sequence characters in front of every byte (one 64-bit sum: records << 36 | bases).
//   k_fa_pack: 16 input bytes per lane -> 2-bit codes (gmg_base_code = tolower (Filter (ch))) OR-ed into the packed
//   words at their final positions, read offsets and header extents scattered by record number, g/c counted.
//
// Headers stay on the host: the caller gets, per read, the byte range of its header line in the input (leading blanks
// after '>' skipped, up to but excluding '\n', exactly the string Fasta_Read returns).
// Limits: < 2^31 bytes per call, < 2^28 reads, < 2^36 bases.

#include "gmg_device.h"

#include <hipcub/hipcub.hpp>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <mutex>
#include <new>
#include <vector>

struct gmg_fasta {
    uint64_t n_reads, total_bases, gc_count;
    // header extents: [n_reads] begin, then [n_reads] end, in ONE page-locked buffer from the library's small cache of them (a pageable
    // destination made the 16 MB of 1 M reads a 5.7 ms copy; this one takes 0.7) -- or in a vector when none could be had
    uint64_t *hdr = nullptr;
    size_t hdr_cap = 0;                                 // bytes; 0: `hdr` points into hdr_vec
    uint64_t hdr_stride = 0;                            // the ends begin at hdr[hdr_stride]
    std::vector<uint64_t> hdr_vec;
};

// page-locked buffers are slow to make (the pages are pinned one by one): the last few are kept
namespace {
struct PinnedCache {
    std::mutex mu;
    struct Item { void *p; size_t cap; };
    std::vector<Item> free_items;
    void *get(size_t bytes, size_t &cap)
    {
        {
            std::lock_guard<std::mutex> g(mu);
            for (size_t i = 0; i < free_items.size(); i++)
                if (free_items[i].cap >= bytes && free_items[i].cap <= 4 * bytes + (1u << 20)) {
                    void *p = free_items[i].p; cap = free_items[i].cap;
                    free_items.erase(free_items.begin() + (long)i);
                    return p;
                }
        }
        void *p = nullptr;
        cap = (bytes + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
        if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return p;
    }
    void put(void *p, size_t cap)
    {
        std::lock_guard<std::mutex> g(mu);
        if (free_items.size() >= 4) { (void)hipHostFree(free_items[0].p); free_items.erase(free_items.begin()); }
        free_items.push_back({p, cap});
    }
};
PinnedCache &pinned_cache() { static PinnedCache *c = new PinnedCache(); return *c; }
}  // namespace

void gmg_ingest_trim(void)
{
    PinnedCache &c = pinned_cache();
    std::lock_guard<std::mutex> g(c.mu);
    for (auto &it : c.free_items) (void)hipHostFree(it.p);
    c.free_items.clear();
}

namespace {

enum { ST_PRE = 0, ST_HDR = 1, ST_SEQ = 2 };

// a byte as a function on the three states: 2 bits per input state
__host__ __device__ inline uint8_t fa_func(uint8_t ch)
{
    if (ch == '>') return (uint8_t)(ST_HDR | ST_HDR << 2 | ST_HDR << 4);
    if (ch == '\n') return (uint8_t)(ST_PRE | ST_SEQ << 2 | ST_SEQ << 4);
    return (uint8_t)(ST_PRE | ST_HDR << 2 | ST_SEQ << 4);                   // identity
}

struct FaCompose {                                                          // first a, then b
    __host__ __device__ uint8_t operator()(uint8_t a, uint8_t b) const
    {
        const unsigned a0 = a & 3u, a1 = (a >> 2) & 3u, a2 = (a >> 4) & 3u;
        return (uint8_t)(((b >> (2 * a0)) & 3u) | ((b >> (2 * a1)) & 3u) << 2 | ((b >> (2 * a2)) & 3u) << 4);
    }
};

struct FaFuncOf {
    __host__ __device__ uint8_t operator()(uint8_t ch) const { return fa_func(ch); }
};

__device__ __forceinline__ bool fa_isspace(uint8_t ch) { return ch == ' ' || (ch >= 9 && ch <= 13); }    // C locale

// state before byte i = composition of bytes [0, i) applied to PRE
__device__ __forceinline__ unsigned fa_state_before(const uint8_t *func_incl, uint64_t i) { return i ? (func_incl[i - 1] & 3u) : ST_PRE; }

#define FA_REC_SHIFT 36

// (records << 36 | bases) contributed by byte i
struct FaCountOf {
    const uint8_t *bytes, *func_incl;
    __device__ uint64_t operator()(uint64_t i) const
    {
        const uint8_t ch = bytes[i];
        const unsigned st = fa_state_before(func_incl, i);
        const uint64_t rec = (ch == '>' && st != ST_HDR) ? 1ull : 0ull;
        const uint64_t seq = (st == ST_SEQ && ch != '>' && !fa_isspace(ch)) ? 1ull : 0ull;
        return rec << FA_REC_SHIFT | seq;
    }
};

// gmg_base_code on the device: tolower (Filter (ch)) as a 2-bit code, a0 c1 g2 t3 (src/Common/gene.cc:1139-1175)
__device__ __forceinline__ uint32_t fa_code(uint8_t ch)
{
    switch (ch | 0x20) {
    case 'a': return 0;
    case 'c': return 1;
    case 'g': return 2;
    case 't': return 3;
    case 'r': return 2; case 'y': return 1; case 's': return 1; case 'w': return 3; case 'm': return 1;
    case 'k': return 3; case 'b': return 1; case 'd': return 2; case 'h': return 1; case 'v': return 1;
    }
    return 1;                                                               // anything else -> 'c'
}

struct FaPackArgs {
    const uint8_t *bytes, *func_incl;
    const uint64_t *count_excl;      // records << 36 | bases in front of byte i
    uint64_t n_bytes;
    uint32_t *packed;                // zeroed
    uint64_t *read_off;              // [n_reads + 1]
    uint64_t *hdr_begin, *hdr_end;   // [n_reads]; hdr_end preset to n_bytes
    unsigned long long *gc_count;
};

__global__ __launch_bounds__(256) void k_fa_pack(FaPackArgs a)
{
    unsigned long long gc = 0;
    for (uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i0 < a.n_bytes; i0 += (uint64_t)gridDim.x * blockDim.x * 16) {
        const uint64_t c0 = a.count_excl[i0];
        uint64_t rec = c0 >> FA_REC_SHIFT, base = c0 & ((1ull << FA_REC_SHIFT) - 1);
        unsigned st = fa_state_before(a.func_incl, i0);
        uint64_t w_idx = base >> 4;
        uint32_t w_val = 0;
        const uint64_t end = i0 + 16 < a.n_bytes ? i0 + 16 : a.n_bytes;
        for (uint64_t i = i0; i < end; i++) {
            const uint8_t ch = a.bytes[i];
            if (ch == '>' && st != ST_HDR) {            // a record starts: its bases begin at `base`, its header behind the '>'
                a.read_off[rec] = base;
                uint64_t hb = i + 1;                    // Fasta_Read skips the blanks behind '>' (fasta.cc:258-260)
                while (hb < a.n_bytes && a.bytes[hb] == ' ') hb++;
                a.hdr_begin[rec] = hb;
                rec++;
            } else if (ch == '\n' && st == ST_HDR) {
                a.hdr_end[rec - 1] = i;                 // the header line of the record that is open
            } else if (st == ST_SEQ && ch != '>' && !fa_isspace(ch)) {
                const uint32_t code = fa_code(ch);
                gc += (code == 1 || code == 2);
                if ((base >> 4) != w_idx) { if (w_val) atomicOr(a.packed + w_idx, w_val); w_idx = base >> 4; w_val = 0; }
                w_val |= code << (2 * (unsigned)(base & 15));
                base++;
            }
            st = (a.func_incl[i] & 3u);
        }
        if (w_val) atomicOr(a.packed + w_idx, w_val);
    }
    // one atomic per wave
    for (int o = 32; o > 0; o >>= 1) gc += __shfl_down(gc, o);
    if ((threadIdx.x & 63) == 0 && gc) atomicAdd(a.gc_count, gc);
}

__global__ __launch_bounds__(256) void k_fa_fill(uint64_t *p, uint64_t n, uint64_t v)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = v;
}


// ---- the block-summary form.  A byte's effect on the state depends on the state in front of it, which a block does not know yet --
// but there are only three states: every lane follows its 16 bytes from all three, a scan composes the lanes' state functions, and a
// block's summary is its composed function + its (records << 36 | bases) for each of the three states it might be entered in.
#define FA_BLK 4096u
struct FaSumm { uint64_t cnt[3]; uint32_t func, pad; };

__device__ __forceinline__ uint32_t fa_apply(uint32_t f, uint32_t v) { return (f >> (2u * v)) & 3u; }
__device__ __forceinline__ uint32_t fa_comp(uint32_t a, uint32_t b)       // first a, then b
{
    return fa_apply(b, a & 3u) | fa_apply(b, (a >> 2) & 3u) << 2 | fa_apply(b, (a >> 4) & 3u) << 4;
}
#define FA_IDENT (uint32_t)(ST_PRE | ST_HDR << 2 | ST_SEQ << 4)

// the 16 bytes of a lane (zero-padded behind the input's end: '\0' is no sequence character... it IS for Fasta_Read, so `len` bounds the loops)
struct FaLane { uint8_t b[16]; int len; };
__device__ __forceinline__ FaLane fa_load16(const uint8_t *bytes, uint64_t n, uint64_t i0)
{
    FaLane L;
    L.len = i0 >= n ? 0 : (n - i0 < 16 ? (int)(n - i0) : 16);
    if (L.len == 16) { const uint4 q = *(const uint4 *)(bytes + i0); memcpy(L.b, &q, 16); }
    else for (int k = 0; k < 16; k++) L.b[k] = k < L.len ? bytes[i0 + k] : (uint8_t)'a';
    return L;
}
// the lane's state function
__device__ __forceinline__ uint32_t fa_lane_func(const FaLane &L)
{
    uint32_t s0 = ST_PRE, s1 = ST_HDR, s2 = ST_SEQ;
    for (int k = 0; k < L.len; k++) {
        const uint8_t ch = L.b[k];
        if (ch == '>') s0 = s1 = s2 = ST_HDR;
        else if (ch == '\n') { s0 = s0 == ST_HDR ? (uint32_t)ST_SEQ : s0; s1 = s1 == ST_HDR ? (uint32_t)ST_SEQ : s1; s2 = s2 == ST_HDR ? (uint32_t)ST_SEQ : s2; }
    }
    return s0 | s1 << 2 | s2 << 4;
}
// records << 36 | bases of the lane's bytes when it is entered in state st
__device__ __forceinline__ uint64_t fa_lane_count(const FaLane &L, uint32_t st)
{
    uint64_t c = 0;
    for (int k = 0; k < L.len; k++) {
        const uint8_t ch = L.b[k];
        if (ch == '>') { c += st != ST_HDR ? 1ull << FA_REC_SHIFT : 0ull; st = ST_HDR; }
        else if (ch == '\n') { if (st == ST_HDR) st = ST_SEQ; }
        else if (st == ST_SEQ && !fa_isspace(ch)) c++;
    }
    return c;
}
// exclusive composition of one function per lane over the 256 lanes of the work-group (s_w: 4 words of LDS); total = all of them
__device__ __forceinline__ uint32_t fa_wg_excl_func(uint32_t f, uint32_t *s_w, uint32_t &total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t x = f;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)x, d, 64); if ((int)lane >= d) x = fa_comp(y, x); }
    if (lane == 63) s_w[wave] = x;
    uint32_t ex = (uint32_t)__shfl_up((int)x, 1, 64);
    if (lane == 0) ex = FA_IDENT;
    __syncthreads();
    uint32_t before = FA_IDENT, all = FA_IDENT;
#pragma unroll
    for (uint32_t w = 0; w < 4; w++) { const uint32_t g = s_w[w]; if (w < wave) before = fa_comp(before, g); all = fa_comp(all, g); }
    __syncthreads();
    total = all;
    return fa_comp(before, ex);
}
__device__ __forceinline__ uint64_t fa_shfl_up64(uint64_t x, int d)
{
    return (uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(x >> 32), d, 64) << 32 | (uint32_t)__shfl_up((int)(uint32_t)x, d, 64);
}
// exclusive sum of one 64-bit value per lane over the work-group (s_c: 4 x 64 bits of LDS); total = the sum
__device__ __forceinline__ uint64_t fa_wg_excl_sum(uint64_t v, uint64_t *s_c, uint64_t &total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint64_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint64_t y = fa_shfl_up64(x, d); if ((int)lane >= d) x += y; }
    if (lane == 63) s_c[wave] = x;
    __syncthreads();
    uint64_t before = 0, all = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; w++) { const uint64_t g = s_c[w]; if (w < wave) before += g; all += g; }
    __syncthreads();
    total = all;
    return before + x - v;
}

// pass 1: one work-group per block of FA_BLK bytes -> its summary
__global__ __launch_bounds__(256) void k_fa_summ(const uint8_t *bytes, uint64_t n, uint64_t n_blocks, FaSumm *summ)
{
    __shared__ uint32_t s_w[4];
    __shared__ uint64_t s_c[4];
    for (uint64_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        const FaLane L = fa_load16(bytes, n, b * FA_BLK + (uint64_t)threadIdx.x * 16);
        const uint32_t f = fa_lane_func(L);
        uint32_t total_f;
        const uint32_t ex = fa_wg_excl_func(f, s_w, total_f);
        uint64_t tot[3];
#pragma unroll
        for (uint32_t v = 0; v < 3; v++) {               // the block entered in state v: this lane is entered in ex (v)
            uint64_t t_;
            (void)fa_wg_excl_sum(fa_lane_count(L, fa_apply(ex, v)), s_c, t_);
            tot[v] = t_;
        }
        if (threadIdx.x == 0) { FaSumm o; o.cnt[0] = tot[0]; o.cnt[1] = tot[1]; o.cnt[2] = tot[2]; o.func = total_f; o.pad = 0; summ[b] = o; }
    }
}

// pass 2: ONE work-group of 1,024 lanes over the summaries of blocks [b_lo, b_hi): the state every block is entered in and the counts
// in front of it.  A lane takes a run of consecutive blocks, follows it from all three states, the lanes' runs are composed like the
// lanes of a block.  carry[0] = the counts, carry[1] = the state in front of block b_lo (read and advanced: the pieces of a chunked
// upload are done one after the other, each behind its own first pass).
__global__ __launch_bounds__(1024) void k_fa_blocks(const FaSumm *summ, uint64_t b_lo, uint64_t b_hi, uint8_t *blk_state, uint64_t *blk_excl, uint64_t *carry)
{
    __shared__ uint32_t s_f[1024];
    __shared__ uint64_t s_cnt[3][1024];
    const uint32_t t = threadIdx.x;
    const uint64_t n_blocks = b_hi - b_lo;
    const uint64_t per = (n_blocks + 1023) / 1024, b0 = b_lo + (uint64_t)t * per, b1 = b0 + per < b_hi ? b0 + per : b_hi;
    uint32_t s[3] = {ST_PRE, ST_HDR, ST_SEQ};
    uint64_t c[3] = {0, 0, 0};
    for (uint64_t b = b0; b < b1; b++) {
        const FaSumm m = summ[b];
#pragma unroll
        for (int v = 0; v < 3; v++) { c[v] += s[v] == 0 ? m.cnt[0] : s[v] == 1 ? m.cnt[1] : m.cnt[2]; s[v] = fa_apply(m.func, s[v]); }
    }
    s_f[t] = s[0] | s[1] << 2 | s[2] << 4;
    s_cnt[0][t] = c[0]; s_cnt[1][t] = c[1]; s_cnt[2][t] = c[2];
    __syncthreads();
    if (t == 0) {                                       // 1,024 steps: the runs one after the other from the state in front of them
        uint32_t st = (uint32_t)carry[1];
        uint64_t acc = carry[0];
        for (uint32_t k = 0; k < 1024; k++) {
            const uint32_t f = s_f[k];
            const uint64_t add = s_cnt[st][k];
            s_f[k] = st;                                // the state run k is entered in
            s_cnt[0][k] = acc;                          // the counts in front of it
            acc += add;
            st = fa_apply(f, st);
        }
        carry[0] = acc;
        carry[1] = st;
    }
    __syncthreads();
    uint32_t st = s_f[t];
    uint64_t acc = s_cnt[0][t];
    for (uint64_t b = b0; b < b1; b++) {
        const FaSumm m = summ[b];
        blk_state[b] = (uint8_t)st;
        blk_excl[b] = acc;
        acc += st == 0 ? m.cnt[0] : st == 1 ? m.cnt[1] : m.cnt[2];
        st = fa_apply(m.func, st);
    }
}

struct FaPack2Args {
    const uint8_t *bytes;
    const uint8_t *blk_state;
    const uint64_t *blk_excl;
    uint64_t n_bytes, b_lo, b_hi;    // the file's length; the blocks of this launch
    uint64_t avail;                  // bytes that have arrived (a piece of a chunked upload: the blanks behind a '>' may run past them)
    uint64_t rec_cap;                // records the arrays hold (a launch ahead of the totals: the arrays are sized by a bound)
    uint32_t *packed;                // zeroed
    uint64_t *read_off;              // [n_reads + 1]
    uint64_t *hdr_begin, *hdr_end;   // [n_reads]; hdr_end preset to n_bytes
    unsigned long long *gc_count;
};

// pass 3: the same bytes again, every lane knowing the state and the counts in front of its 16 bytes: k_fa_pack's inner loop
__global__ __launch_bounds__(256) void k_fa_pack2(FaPack2Args a)
{
    __shared__ uint32_t s_w[4];
    __shared__ uint64_t s_c[4];
    unsigned long long gc = 0;
    for (uint64_t b = a.b_lo + blockIdx.x; b < a.b_hi; b += gridDim.x) {
        const uint64_t i0 = b * FA_BLK + (uint64_t)threadIdx.x * 16;
        const FaLane L = fa_load16(a.bytes, a.n_bytes, i0);
        uint32_t total_f;
        const uint32_t ex = fa_wg_excl_func(fa_lane_func(L), s_w, total_f);
        unsigned st = fa_apply(ex, a.blk_state[b]);
        uint64_t total_c;
        const uint64_t c0 = a.blk_excl[b] + fa_wg_excl_sum(fa_lane_count(L, st), s_c, total_c);
        uint64_t rec = c0 >> FA_REC_SHIFT, base = c0 & ((1ull << FA_REC_SHIFT) - 1);
        uint64_t w_idx = base >> 4;
        uint32_t w_val = 0;
        for (int k = 0; k < L.len; k++) {
            const uint8_t ch = L.b[k];
            const uint64_t i = i0 + (uint64_t)k;
            if (ch == '>' && st != ST_HDR) {            // a record starts: its bases begin at `base`, its header behind the '>'
                if (rec < a.rec_cap) {
                    a.read_off[rec] = base;
                    uint64_t hb = i + 1;                // Fasta_Read skips the blanks behind '>' (fasta.cc:258-260)
                    while (hb < a.avail && a.bytes[hb] == ' ') hb++;
                    // (blanks up to the last byte that has arrived, more to come: k_fa_hdr_fix goes on from there once all have)
                    a.hdr_begin[rec] = hb == a.avail && a.avail < a.n_bytes ? hb | 1ull << 63 : hb;
                }
                rec++;
            } else if (ch == '\n' && st == ST_HDR) {
                if (rec - 1 < a.rec_cap) a.hdr_end[rec - 1] = i;      // the header line of the record that is open
            } else if (st == ST_SEQ && ch != '>' && !fa_isspace(ch)) {
                const uint32_t code = fa_code(ch);
                gc += (code == 1 || code == 2);
                if ((base >> 4) != w_idx) { if (w_val) atomicOr(a.packed + w_idx, w_val); w_idx = base >> 4; w_val = 0; }
                w_val |= code << (2 * (unsigned)(base & 15));
                base++;
            }
            st = ch == '>' ? (unsigned)ST_HDR : (ch == '\n' && st == ST_HDR) ? (unsigned)ST_SEQ : st;
        }
        if (w_val) atomicOr(a.packed + w_idx, w_val);
    }
    for (int o = 32; o > 0; o >>= 1) gc += __shfl_down(gc, o);
    if ((threadIdx.x & 63) == 0 && gc) atomicAdd(a.gc_count, gc);
}

// header extents whose blanks ran up to the end of a piece of the upload: on from there, all bytes being there now
__global__ __launch_bounds__(256) void k_fa_hdr_fix(const uint8_t *bytes, uint64_t n_bytes, uint64_t *hdr_begin, uint64_t n_reads)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_reads; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t hb = hdr_begin[i];
        if (!(hb >> 63)) continue;
        hb &= ~(1ull << 63);
        while (hb < n_bytes && bytes[hb] == ' ') hb++;
        hdr_begin[i] = hb;
    }
}

// shortest / longest read and the reads over 512 bases (what gmg_reads keeps about a batch), from the offsets: stats[0] min, [1] max, [2] count
__global__ __launch_bounds__(256) void k_fa_stats(const uint64_t *off, uint64_t n_reads, unsigned long long *stats)
{
    unsigned long long mn = ~0ull, mx = 0, over = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_reads; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long l = off[i + 1] - off[i];
        mn = l < mn ? l : mn;
        mx = l > mx ? l : mx;
        over += l > 512;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long a = __shfl_xor(mn, o), b = __shfl_xor(mx, o);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
        over += __shfl_xor(over, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (mn != ~0ull) atomicMin(&stats[0], mn);
        if (mx) atomicMax(&stats[1], mx);
        if (over) atomicAdd(&stats[2], over);
    }
}

// the copy stream of the chunked upload and its events (one set per process: ingest calls of different threads take turns on it)
struct FaCopyLane {
    std::mutex mu;
    hipStream_t copy = nullptr;
    hipEvent_t ev[16] = {};
    hipEvent_t start = nullptr;
    bool ok = false, tried = false;
    bool init()
    {
        if (tried) return ok;
        tried = true;
        if (hipStreamCreateWithFlags(&copy, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return false; }
        for (int i = 0; i < 16; i++)
            if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (hipEventCreateWithFlags(&start, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
        ok = true;
        return true;
    }
};
FaCopyLane &copy_lane() { static FaCopyLane *c = new FaCopyLane(); return *c; }

struct DevFree {                                       // temporaries, from the library's cache of device blocks
    std::vector<void *> v;
    hipStream_t st = nullptr;
    ~DevFree() { (void)hipStreamSynchronize(st); for (void *p : v) gmg_pool_release(p); }
    template <class T> hipError_t alloc(T **p, size_t bytes) { hipError_t e = gmg_pool_alloc((void **)p, bytes); if (e == hipSuccess) v.push_back(*p); return e; }
};

}  // namespace

extern "C" int gmg_fasta_free(gmg_fasta *f)
{
    if (f && f->hdr_cap) pinned_cache().put(f->hdr, f->hdr_cap);
    delete f;
    return GMG_OK;
}

extern "C" int gmg_fasta_ingest(const char *bytes, uint64_t n_bytes, gmg_reads **out_reads, gmg_fasta **out_index)
{
    return gmg_fasta_ingest_on(bytes, n_bytes, out_reads, out_index, nullptr);
}

// Safe places to cut a big file into pieces for gmg_fasta_ingest: a '>' that directly follows a newline is never
// inside a header line, so it always starts a record.  cuts[0] = 0, then the first such position at or behind every
// multiple of chunk_bytes, at last n_bytes; returns the number of pieces (cuts has pieces + 1 entries).
extern "C" int gmg_fasta_split(const char *bytes, uint64_t n_bytes, uint64_t chunk_bytes, uint64_t *cuts, int max_pieces)
{
    if ((!bytes && n_bytes) || !cuts || max_pieces < 1 || chunk_bytes == 0) return gmg_set_error(GMG_EINVAL, "gmg_fasta_split: bad argument");
    int n = 0;
    cuts[0] = 0;
    uint64_t want = chunk_bytes;
    while (want < n_bytes && n + 1 < max_pieces) {
        uint64_t i = want;
        while (i < n_bytes && !(bytes[i] == '>' && bytes[i - 1] == '\n')) i++;
        if (i >= n_bytes) break;
        if (i > cuts[n]) cuts[++n] = i;
        want = i + chunk_bytes;
    }
    cuts[++n] = n_bytes;
    return n;
}

extern "C" int gmg_fasta_ingest_on(const char *bytes, uint64_t n_bytes, gmg_reads **out_reads, gmg_fasta **out_index, void *stream)
{
    { int rc_enter = gmg_enter("gmg_fasta_ingest_on"); if (rc_enter) return rc_enter; }
    hipStream_t st = (hipStream_t)stream;
    const bool timing = gmg_opt(GMG_OPT_INGEST_TIMING) != 0;         // wall time of every stage on stderr
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(st);
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[gmg_ingest] %-24s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = std::chrono::steady_clock::now();
    };
    if ((!bytes && n_bytes) || !out_reads || !out_index) return gmg_set_error(GMG_EINVAL, "gmg_fasta_ingest: NULL argument");
    if (n_bytes >= 0x7fffffffull) return gmg_set_error(GMG_EINVAL, "gmg_fasta_ingest: at most 2^31 - 2 bytes per call");
    gmg_fasta *idx = new (std::nothrow) gmg_fasta();
    if (!idx) return gmg_set_error(GMG_ENOMEM, "gmg_fasta_ingest: out of host memory");
    idx->n_reads = idx->total_bases = idx->gc_count = 0;
    // what goes to the gmg_reads: the packed words in their guarded buffer, the offsets (the guards are older than `dev`: on an early
    // return they let go of the blocks AFTER dev's destructor has waited for the stream)
    uint32_t *d_alloc = nullptr;
    uint64_t *d_off = nullptr;
    struct BufGuard { uint32_t *&p; ~BufGuard() { if (p) gmg_pool_release(p); } } alloc_guard = {d_alloc};   // until the reads own it
    struct OffGuard { uint64_t *&p; ~OffGuard() { if (p) gmg_pool_release(p); } } off_guard = {d_off};       // until then it is ours
    DevFree dev;
    dev.st = st;
    uint8_t *d_bytes = nullptr, *d_func = nullptr, *d_bstate = nullptr;
    FaSumm *d_summ = nullptr;
    uint64_t *d_bexcl = nullptr, *d_tot = nullptr, n_blocks = 0;
    uint64_t *d_count = nullptr, *d_hb = nullptr, *d_he = nullptr;
    uint32_t *d_packed = nullptr;
    unsigned long long *d_gc = nullptr;
    void *d_tmp = nullptr;
#define FA_TRY(call)                                                                                              \
    do {                                                                                                          \
        hipError_t e_ = (call);                                                                                   \
        if (e_ != hipSuccess) {                                                                                   \
            delete idx;                                                                                           \
            return gmg_set_error(e_ == hipErrorOutOfMemory ? GMG_ENOMEM : GMG_EHIP, "gmg_fasta_ingest: %s: %s", #call, \
                                 hipGetErrorString(e_));                                                          \
        }                                                                                                         \
    } while (0)
    const uint64_t n = n_bytes;
    uint64_t n_reads = 0, total = 0;
    // a chunked upload packs every piece as it arrives, into arrays sized by a bound: rec_cap records (16 file bytes per record; a file
    // with more takes the plain order once the totals are known), one base per byte
    bool spec = false;
    uint64_t rec_cap = 0;
    uint64_t *d_off_big = nullptr;
    if (n) {
        const bool scans = gmg_opt(GMG_OPT_INGEST_SCANS) != 0;       // the first version: two scans over every byte
        n_blocks = (n + FA_BLK - 1) / FA_BLK;
        FA_TRY(dev.alloc(&d_bytes, n + 16));
        if (scans) {
            FA_TRY(dev.alloc(&d_func, n));
            FA_TRY(dev.alloc(&d_count, (n + 1) * 8));
        } else {
            FA_TRY(dev.alloc(&d_summ, n_blocks * sizeof(FaSumm)));
            FA_TRY(dev.alloc(&d_bstate, n_blocks));
            FA_TRY(dev.alloc(&d_bexcl, n_blocks * 8));
            FA_TRY(dev.alloc(&d_tot, 16));
        }
        if (!scans) FA_TRY(hipMemsetAsync(d_tot, 0, 16, st));               // counts 0, state PRE in front of the first block
        lap("alloc");
        // the upload in up to 16 pieces on a stream of its own, the first pass over a piece as soon as it has arrived (the pass is
        // hidden behind the next piece's copy; small inputs, the scans' form and a busy copy lane take one plain copy)
        FaCopyLane &cl = copy_lane();
        const uint64_t piece = ((n / 16 + FA_BLK - 1) / FA_BLK + 1) * FA_BLK;          // a multiple of the block size
        bool piecewise = !scans && !timing && (long long)n >= gmg_opt(GMG_OPT_INGEST_PIECE_MIN) && cl.mu.try_lock();
        if (piecewise && !cl.init()) { cl.mu.unlock(); piecewise = false; }
        if (piecewise) {
            hipError_t pe = hipEventRecord(cl.start, st);                  // (the buffer is ours from here on in `st`'s order)
            if (pe == hipSuccess) pe = hipStreamWaitEvent(cl.copy, cl.start, 0);
            // (queued on `st` behind the event: they run beside the first piece's copy)
            rec_cap = n / 16 + 1024;
            const uint64_t words_cap = (n + 15) / 16;
            if (pe == hipSuccess) pe = gmg_pool_alloc((void **)&d_alloc, (words_cap + 2 * GMG_GUARD_WORDS + 1) * 4);
            if (pe == hipSuccess) pe = dev.alloc(&d_off_big, (rec_cap + 1) * 8);
            if (pe == hipSuccess) pe = dev.alloc(&d_hb, rec_cap * 8);
            if (pe == hipSuccess) pe = dev.alloc(&d_he, rec_cap * 8);
            if (pe == hipSuccess) pe = dev.alloc(&d_gc, 8);
            if (pe == hipSuccess) pe = hipMemsetAsync(d_alloc, 0, (words_cap + 2 * GMG_GUARD_WORDS + 1) * 4, st);
            if (pe == hipSuccess) pe = hipMemsetAsync(d_gc, 0, 8, st);
            if (pe == hipSuccess) {
                hipLaunchKernelGGL(k_fa_fill, dim3(4096), dim3(256), 0, st, d_he, rec_cap, n_bytes);
                pe = hipGetLastError();
                d_packed = d_alloc + GMG_GUARD_WORDS;
                spec = true;
            }
            int k = 0;
            for (uint64_t b0 = 0; pe == hipSuccess && b0 < n; b0 += piece, k++) {
                const uint64_t len = n - b0 < piece ? n - b0 : piece;
                pe = hipMemcpyAsync(d_bytes + b0, bytes + b0, len, hipMemcpyHostToDevice, cl.copy);
                if (pe == hipSuccess) pe = hipEventRecord(cl.ev[k], cl.copy);
                if (pe == hipSuccess) pe = hipStreamWaitEvent(st, cl.ev[k], 0);
                if (pe == hipSuccess) {
                    const uint64_t blk0 = b0 / FA_BLK, nb = (len + FA_BLK - 1) / FA_BLK;
                    hipLaunchKernelGGL(k_fa_summ, dim3((unsigned)(nb < 256 * 16 ? nb : 256 * 16)), dim3(256), 0, st, d_bytes + b0, len, nb, d_summ + blk0);
                    hipLaunchKernelGGL(k_fa_blocks, dim3(1), dim3(1024), 0, st, d_summ, blk0, blk0 + nb, d_bstate, d_bexcl, d_tot);
                    FaPack2Args pa = {d_bytes, d_bstate, d_bexcl, n, blk0, blk0 + nb, b0 + len, rec_cap, d_packed, d_off_big, d_hb, d_he, d_gc};
                    hipLaunchKernelGGL(k_fa_pack2, dim3((unsigned)(nb < 256 * 16 ? nb : 256 * 16)), dim3(256), 0, st, pa);
                    pe = hipGetLastError();
                }
            }
            if (pe != hipSuccess) (void)hipStreamSynchronize(cl.copy);
            cl.mu.unlock();
            FA_TRY(pe);
        } else
            FA_TRY(hipMemcpyAsync(d_bytes, bytes, n, hipMemcpyHostToDevice, st));
        lap("copy file to device");
        if (!scans) {
            if (!piecewise) {
                hipLaunchKernelGGL(k_fa_summ, dim3((unsigned)(n_blocks < 256 * 64 ? n_blocks : 256 * 64)), dim3(256), 0, st, d_bytes, n, n_blocks, d_summ);
                hipLaunchKernelGGL(k_fa_blocks, dim3(1), dim3(1024), 0, st, d_summ, (uint64_t)0, n_blocks, d_bstate, d_bexcl, d_tot);
            }
            FA_TRY(hipGetLastError());
            uint64_t tot[2] = {0, 0};
            FA_TRY(hipMemcpyAsync(tot, d_tot, 16, hipMemcpyDeviceToHost, st));
            FA_TRY(hipStreamSynchronize(st));
            lap("block summaries");
            n_reads = tot[0] >> FA_REC_SHIFT;
            total = tot[0] & ((1ull << FA_REC_SHIFT) - 1);
        } else {
        // 1. the state behind every byte
        hipcub::TransformInputIterator<uint8_t, FaFuncOf, const uint8_t *> func_in(d_bytes, FaFuncOf());
        size_t tmp_bytes = 0, tmp2 = 0;
        FA_TRY(hipcub::DeviceScan::InclusiveScan(nullptr, tmp_bytes, func_in, d_func, FaCompose(), (int)n));
        // 2. records and bases in front of every byte (entry n = the totals)
        hipcub::CountingInputIterator<uint64_t> pos(0);
        FaCountOf count_of = {d_bytes, d_func};
        hipcub::TransformInputIterator<uint64_t, FaCountOf, hipcub::CountingInputIterator<uint64_t>> count_in(pos, count_of);
        FA_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp2, count_in, d_count, (int)n));
        if (tmp2 > tmp_bytes) tmp_bytes = tmp2;
        FA_TRY(dev.alloc(&d_tmp, tmp_bytes));
        size_t tb = tmp_bytes;
        FA_TRY(hipcub::DeviceScan::InclusiveScan(d_tmp, tb, func_in, d_func, FaCompose(), (int)n, st));
        tb = tmp_bytes;
        FA_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp, tb, count_in, d_count, (int)n, st));
        uint64_t last_excl = 0;
        uint8_t last_func = 0, last_byte = 0, prev_func = 0;
        FA_TRY(hipMemcpyAsync(&last_excl, d_count + (n - 1), 8, hipMemcpyDeviceToHost, st));
        FA_TRY(hipMemcpyAsync(&last_func, d_func + (n - 1), 1, hipMemcpyDeviceToHost, st));
        if (n > 1) FA_TRY(hipMemcpyAsync(&prev_func, d_func + (n - 2), 1, hipMemcpyDeviceToHost, st));
        FA_TRY(hipStreamSynchronize(st));
        lap("two scans");
        last_byte = (uint8_t)bytes[n - 1];
        const unsigned st_last = n > 1 ? (prev_func & 3u) : (unsigned)ST_PRE;
        n_reads = (last_excl >> FA_REC_SHIFT) + ((last_byte == '>' && st_last != ST_HDR) ? 1 : 0);
        total = (last_excl & ((1ull << FA_REC_SHIFT) - 1)) +
                ((st_last == ST_SEQ && last_byte != '>' && !(last_byte == ' ' || (last_byte >= 9 && last_byte <= 13))) ? 1 : 0);
        (void)last_func;
        }
    }
    if (n_reads >= (1ull << 28)) { delete idx; return gmg_set_error(GMG_EINVAL, "gmg_fasta_ingest: too many records in one call"); }
    // 3. pack, offsets, header extents, g/c count
    // (the packed words are written where they stay: the gmg_reads' buffer with its guard words on both sides)
    const uint64_t data_words = (total + 15) / 16;
    unsigned long long *d_stats = nullptr;
    FA_TRY(dev.alloc(&d_stats, 32));
    hipError_t e2 = hipSuccess;
    if (spec && n_reads <= rec_cap) {
        // the pieces are packed already: the offsets move into an array of their size, the header extents that ran up to a piece's end are completed
        FA_TRY(gmg_pool_alloc((void **)&d_off, (n_reads + 1) * 8));           // goes to the gmg_reads
        if (n_reads) e2 = hipMemcpyAsync(d_off, d_off_big, n_reads * 8, hipMemcpyDeviceToDevice, st);
        if (e2 == hipSuccess) e2 = hipMemcpyAsync(d_off + n_reads, &total, 8, hipMemcpyHostToDevice, st);
        if (e2 == hipSuccess && n_reads) {
            const uint64_t blocks = (n_reads + 255) / 256;
            hipLaunchKernelGGL(k_fa_hdr_fix, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, d_bytes, n, d_hb, n_reads);
        }
        if (e2 != hipSuccess) { delete idx; return gmg_set_error(GMG_EHIP, "gmg_fasta_ingest: %s", hipGetErrorString(e2)); }
    } else {
        if (spec) {                                     // more records than the bound: the arrays again, in their real size
            gmg_pool_release(d_alloc);
            d_alloc = nullptr;
            d_hb = d_he = nullptr;                      // (the bound-sized ones stay with `dev` until the call ends)
            spec = false;
        }
        FA_TRY(gmg_pool_alloc((void **)&d_alloc, (data_words + 2 * GMG_GUARD_WORDS + 1) * 4));
        d_packed = d_alloc + GMG_GUARD_WORDS;
        FA_TRY(hipMemsetAsync(d_alloc, 0, (data_words + 2 * GMG_GUARD_WORDS + 1) * 4, st));
        FA_TRY(gmg_pool_alloc((void **)&d_off, (n_reads + 1) * 8));           // goes to the gmg_reads
        e2 = dev.alloc(&d_hb, n_reads * 8);
        if (e2 == hipSuccess) e2 = dev.alloc(&d_he, n_reads * 8);
        if (e2 == hipSuccess && !d_gc) e2 = dev.alloc(&d_gc, 8);
        if (e2 == hipSuccess) e2 = hipMemsetAsync(d_gc, 0, 8, st);
        if (e2 == hipSuccess) e2 = hipMemcpyAsync(d_off + n_reads, &total, 8, hipMemcpyHostToDevice, st);
        if (e2 != hipSuccess) { delete idx; return gmg_set_error(GMG_ENOMEM, "gmg_fasta_ingest: %s", hipGetErrorString(e2)); }
        if (n_reads) {
            const uint64_t blocks = (n_reads + 255) / 256;
            hipLaunchKernelGGL(k_fa_fill, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, d_he, n_reads, n_bytes);
        }
        if (n && d_summ) {
            FaPack2Args a = {d_bytes, d_bstate, d_bexcl, n, 0, n_blocks, n, n_reads, d_packed, d_off, d_hb, d_he, d_gc};
            hipLaunchKernelGGL(k_fa_pack2, dim3((unsigned)(n_blocks < 256 * 64 ? n_blocks : 256 * 64)), dim3(256), 0, st, a);
        } else if (n) {
            FaPackArgs a = {d_bytes, d_func, d_count, n, d_packed, d_off, d_hb, d_he, d_gc};
            const uint64_t blocks = (n / 16 + 255) / 256 + 1;
            hipLaunchKernelGGL(k_fa_pack, dim3((unsigned)(blocks < 256 * 32 ? blocks : 256 * 32)), dim3(256), 0, st, a);
        }
    }
    e2 = hipGetLastError();
    if (e2 == hipSuccess) e2 = hipStreamSynchronize(st);
    lap("pack kernel");
    unsigned long long gc = 0;
    if (n_reads) {
        idx->hdr = (uint64_t *)pinned_cache().get(2 * n_reads * 8, idx->hdr_cap);
        if (!idx->hdr) { idx->hdr_cap = 0; idx->hdr_vec.resize(2 * n_reads); idx->hdr = idx->hdr_vec.data(); }
    }
    if (e2 == hipSuccess && n_reads) e2 = hipMemcpyAsync(idx->hdr, d_hb, n_reads * 8, hipMemcpyDeviceToHost, st);
    if (e2 == hipSuccess && n_reads) e2 = hipMemcpyAsync(idx->hdr + n_reads, d_he, n_reads * 8, hipMemcpyDeviceToHost, st);
    if (e2 == hipSuccess) e2 = hipMemcpyAsync(&gc, d_gc, 8, hipMemcpyDeviceToHost, st);
    if (e2 == hipSuccess) e2 = hipStreamSynchronize(st);
    if (e2 != hipSuccess) { (void)gmg_fasta_free(idx); return gmg_set_error(GMG_EHIP, "gmg_fasta_ingest: %s", hipGetErrorString(e2)); }
    lap("headers to host");
    idx->n_reads = n_reads;                             // (begin at hdr[i], end at hdr[n_reads_as_copied + i]: kept for gmg_fasta_headers)
    const uint64_t n_copied = n_reads;
    // Fasta_Read: a record that is only "> <blanks> EOF" does not exist (the blanks were skipped on the device)
    if (n_reads && idx->hdr[n_reads - 1] == n_bytes && idx->hdr[n_copied + n_reads - 1] == n_bytes)
        n_reads--;                                      // fasta.cc:258-261: EOF while skipping the blanks -> return false
    // the gmg_reads around what is on the device already: the tile table, and the batch's length statistics from a reduction over
    // the offsets (gmg_reads_wrap_device copied the words once more and read 8 bytes per read back through pageable memory: 1.2 ms)
    gmg_reads *reads = new (std::nothrow) gmg_reads();
    if (!reads) { (void)gmg_fasta_free(idx); return gmg_set_error(GMG_ENOMEM, "gmg_fasta_ingest: out of host memory"); }
    memset(reads, 0, sizeof *reads);
    reads->n_reads = n_reads;
    reads->total_bases = total;
    reads->n_words = data_words + GMG_GUARD_WORDS;
    reads->n_tiles = (total + GMG_TILE - 1) / GMG_TILE;
    unsigned long long stats[3] = {~0ull, 0, 0};
    e2 = gmg_pool_alloc((void **)&reads->d_tile_read, (reads->n_tiles + 1) * sizeof(uint32_t));
    if (e2 == hipSuccess) e2 = hipMemcpyAsync(d_stats, stats, 24, hipMemcpyHostToDevice, st);
    int rc = 0;
    if (e2 == hipSuccess) rc = gmg_launch_tile_read(d_off, n_reads, reads->n_tiles, reads->d_tile_read, st);
    if (e2 == hipSuccess && !rc && n_reads) {
        const uint64_t blocks = (n_reads + 255) / 256;
        hipLaunchKernelGGL(k_fa_stats, dim3((unsigned)(blocks < 128 ? blocks : 128)), dim3(256), 0, st, d_off, n_reads, d_stats);      // (few waves: three atomics each)
        e2 = hipGetLastError();
    }
    if (e2 == hipSuccess && !rc) e2 = hipMemcpyAsync(stats, d_stats, 24, hipMemcpyDeviceToHost, st);
    if (e2 == hipSuccess && !rc) e2 = hipStreamSynchronize(st);
    if (e2 != hipSuccess || rc) {
        if (reads->d_tile_read) gmg_pool_release(reads->d_tile_read);
        delete reads;
        (void)gmg_fasta_free(idx);
        return rc ? rc : gmg_set_error(GMG_EHIP, "gmg_fasta_ingest: %s", hipGetErrorString(e2));
    }
    reads->d_packed_alloc = d_alloc;
    reads->d_packed = d_packed;
    reads->d_off = d_off;
    reads->owns_off = 1;                                // buffer and offsets now belong to the reads
    d_alloc = nullptr;
    d_off = nullptr;
    reads->min_len = n_reads ? stats[0] : 0;
    reads->max_len = stats[1];
    reads->n_over_512 = stats[2];
    reads->uniform_len = (n_reads && stats[0] == stats[1] && stats[0] > 0 && stats[0] < (1u << 30)) ? (int)stats[0] : 0;
    lap("reads object");
    idx->n_reads = n_reads;
    idx->hdr_stride = n_copied;
    idx->total_bases = total;
    idx->gc_count = gc;
    *out_reads = reads;
    *out_index = idx;
#undef FA_TRY
    return GMG_OK;
}

extern "C" int gmg_fasta_info(const gmg_fasta *f, uint64_t *n_reads, uint64_t *total_bases, uint64_t *gc_count)
{
    if (!f) return gmg_set_error(GMG_EINVAL, "gmg_fasta_info: NULL index");
    if (n_reads) *n_reads = f->n_reads;
    if (total_bases) *total_bases = f->total_bases;
    if (gc_count) *gc_count = f->gc_count;
    return GMG_OK;
}

extern "C" int gmg_fasta_headers(const gmg_fasta *f, uint64_t *hdr_begin, uint64_t *hdr_end)
{
    if (!f || ((!hdr_begin || !hdr_end) && f->n_reads)) return gmg_set_error(GMG_EINVAL, "gmg_fasta_headers: NULL argument");
    if (f->n_reads) {
        memcpy(hdr_begin, f->hdr, f->n_reads * 8);
        memcpy(hdr_end, f->hdr + f->hdr_stride, f->n_reads * 8);
    }
    return GMG_OK;
}
