// gmg_device.h -- device helpers shared by the HIP translation units: scoring buffers cut from
// packed reads and the plain tree descent on the original tables (exact for every model shape).
#ifndef GMG_DEVICE_H
#define GMG_DEVICE_H

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
__device__ inline int dev_descend(const GmgDevModel &m, const DevBuf &b, int j, int f)
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


// window of 32 packed bases starting at job-wide base `first` (may be slightly negative / past the
// end: the packed buffer has GMG_GUARD_WORDS zero words on both sides)
__device__ __forceinline__ uint64_t dev_window_bits(const uint32_t *__restrict__ packed, int64_t first)
{
    const int64_t w0 = first >> 4;                      // arithmetic shift: floor for negatives
    const unsigned sh = 2u * (unsigned)(first & 15);
    const uint64_t lo = (uint64_t)packed[w0] | ((uint64_t)packed[w0 + 1] << 32);
    const uint64_t hi = packed[w0 + 2];
    return (lo >> sh) | ((hi << 1) << (63 - sh));
}

// reverse the order of `nfields` 2-bit fields held in the low bits of y
__device__ __forceinline__ uint32_t dev_reverse_fields(uint32_t y, int nfields)
{
    uint32_t z = __brev(y) >> (32 - 2 * nfields);
    return ((z & 0x55555555u) << 1) | ((z >> 1) & 0x55555555u);
}


#endif
