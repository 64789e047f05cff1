// gmg_ingest.hip -- FASTA bytes -> packed reads, on the device.
//
// What it replaces: Fasta_Read (src/Common/fasta.cc:236-286) called in a loop, the callers' per-base
// tolower (Filter (ch)) (src/Glimmer/glimmer3.cc:270-271, glimmer-mg.cc:381-382; Filter = src/Common/gene.cc:1139-1175)
// and the g/c count of Set_GC_Fraction (src/Glimmer/glimmer_base.cc:2564-2595) -- a byte-at-a-time fgetc loop on the
// host, ~1 GB/s, which at the kernels' rates is the wall.  Here the raw file bytes are copied to HBM once and parsed by
// three data-parallel passes:
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
#include <new>
#include <vector>

struct gmg_fasta {
    uint64_t n_reads, total_bases, gc_count;
    std::vector<uint64_t> hdr_begin, hdr_end;
};

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

struct DevFree {                                       // temporaries, from the library's cache of device blocks
    std::vector<void *> v;
    hipStream_t st = nullptr;
    ~DevFree() { (void)hipStreamSynchronize(st); for (void *p : v) gmg_pool_release(p); }
    template <class T> hipError_t alloc(T **p, size_t bytes) { hipError_t e = gmg_pool_alloc((void **)p, bytes); if (e == hipSuccess) v.push_back(*p); return e; }
};

}  // namespace

extern "C" int gmg_fasta_free(gmg_fasta *f)
{
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
    DevFree dev;
    dev.st = st;
    uint8_t *d_bytes = nullptr, *d_func = nullptr;
    uint64_t *d_count = nullptr, *d_off = nullptr, *d_hb = nullptr, *d_he = nullptr;
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
    if (n) {
        FA_TRY(dev.alloc(&d_bytes, n));
        FA_TRY(dev.alloc(&d_func, n));
        FA_TRY(dev.alloc(&d_count, (n + 1) * 8));
        lap("alloc");
        FA_TRY(hipMemcpyAsync(d_bytes, bytes, n, hipMemcpyHostToDevice, st));
        lap("copy file to device");
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
    if (n_reads >= (1ull << 28)) { delete idx; return gmg_set_error(GMG_EINVAL, "gmg_fasta_ingest: too many records in one call"); }
    // 3. pack, offsets, header extents, g/c count
    const uint64_t n_words = gmg_packed_words(total);
    FA_TRY(dev.alloc(&d_packed, (n_words + 1) * 4));
    FA_TRY(hipMemsetAsync(d_packed, 0, (n_words + 1) * 4, st));
    FA_TRY(hipMalloc((void **)&d_off, (n_reads + 1) * 8));                   // goes to the gmg_reads
    struct OffGuard { uint64_t *&p; ~OffGuard() { if (p) (void)hipFree(p); } } off_guard = {d_off};   // until then it is ours
    hipError_t e2 = dev.alloc(&d_hb, n_reads * 8);
    if (e2 == hipSuccess) e2 = dev.alloc(&d_he, n_reads * 8);
    if (e2 == hipSuccess) e2 = dev.alloc(&d_gc, 8);
    if (e2 == hipSuccess) e2 = hipMemsetAsync(d_gc, 0, 8, st);
    if (e2 == hipSuccess) e2 = hipMemcpyAsync(d_off + n_reads, &total, 8, hipMemcpyHostToDevice, st);
    if (e2 != hipSuccess) { delete idx; return gmg_set_error(GMG_ENOMEM, "gmg_fasta_ingest: %s", hipGetErrorString(e2)); }
    if (n_reads) {
        const uint64_t blocks = (n_reads + 255) / 256;
        hipLaunchKernelGGL(k_fa_fill, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, d_he, n_reads, n_bytes);
    }
    if (n) {
        FaPackArgs a = {d_bytes, d_func, d_count, n, d_packed, d_off, d_hb, d_he, d_gc};
        const uint64_t blocks = (n / 16 + 255) / 256 + 1;
        hipLaunchKernelGGL(k_fa_pack, dim3((unsigned)(blocks < 256 * 32 ? blocks : 256 * 32)), dim3(256), 0, st, a);
    }
    e2 = hipGetLastError();
    if (e2 == hipSuccess) e2 = hipStreamSynchronize(st);
    lap("pack kernel");
    idx->hdr_begin.resize(n_reads);
    idx->hdr_end.resize(n_reads);
    unsigned long long gc = 0;
    if (e2 == hipSuccess && n_reads) e2 = hipMemcpyAsync(idx->hdr_begin.data(), d_hb, n_reads * 8, hipMemcpyDeviceToHost, st);
    if (e2 == hipSuccess && n_reads) e2 = hipMemcpyAsync(idx->hdr_end.data(), d_he, n_reads * 8, hipMemcpyDeviceToHost, st);
    if (e2 == hipSuccess) e2 = hipMemcpyAsync(&gc, d_gc, 8, hipMemcpyDeviceToHost, st);
    if (e2 == hipSuccess) e2 = hipStreamSynchronize(st);
    if (e2 != hipSuccess) { delete idx; return gmg_set_error(GMG_EHIP, "gmg_fasta_ingest: %s", hipGetErrorString(e2)); }
    lap("headers to host");
    // Fasta_Read: a record that is only "> <blanks> EOF" does not exist (the blanks were skipped on the device)
    if (n_reads && idx->hdr_begin[n_reads - 1] == n_bytes && idx->hdr_end[n_reads - 1] == n_bytes) {
        n_reads--;                                      // fasta.cc:258-261: EOF while skipping the blanks -> return false
        idx->hdr_begin.pop_back();
        idx->hdr_end.pop_back();
    }
    gmg_reads *reads = nullptr;
    int rc = gmg_reads_wrap_device(d_packed, d_off, n_reads, total, &reads);    // copies the words into the guarded buffer
    if (rc) { delete idx; return rc; }
    lap("wrap reads");
    reads->owns_off = 1;                                // the offsets now belong to the reads
    d_off = nullptr;
    idx->n_reads = n_reads;
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
        memcpy(hdr_begin, f->hdr_begin.data(), f->n_reads * 8);
        memcpy(hdr_end, f->hdr_end.data(), f->n_reads * 8);
    }
    return GMG_OK;
}
