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
// ---------------------------------------------------------------------------
// Running sums over the steps of a walk, three classes at once (the three reading-frame classes of a strand: gmg_mg.hip's error
// branch and gmg_orfs.hip's events path keep one running sum per class along a read).
//
// A wave takes 64 * EL consecutive steps per trip.  Loads and stores want a lane on every 64th step (coalesced), the sums want a
// lane on EL consecutive steps (a serial prefix in registers, then ONE scan of the lane totals per class): the values change
// owner through LDS in between.  With a lane per step and a wave scan per 64 steps the DPP scans were the bound (3 x 18
// instructions per 64 steps: k_orf_walk_sums 270 wave-instructions per 64 steps, 1.34 ms per 100 Mbases x 2 strands).
// ---------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double wcs_dpp_add(double x)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, ROW_MASK, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return x + __longlong_as_double((long long)((unsigned long long)hi << 32 | lo));      // (lanes without a source add 0)
}
__device__ __forceinline__ double wcs_wave_scan(double x)          // inclusive sum over the lanes of a wave
{
    x = wcs_dpp_add<0x111, 0xf>(x);                     // row_shr:1
    x = wcs_dpp_add<0x112, 0xf>(x);
    x = wcs_dpp_add<0x114, 0xf>(x);
    x = wcs_dpp_add<0x118, 0xf>(x);
    x = wcs_dpp_add<0x142, 0xa>(x);                     // row_bcast:15 into rows 1 and 3
    x = wcs_dpp_add<0x143, 0xc>(x);                     // row_bcast:31 into rows 2 and 3
    return x;
}
__device__ __forceinline__ double wcs_last_lane(double x)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return __longlong_as_double((long long)((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), 63) << 32 |
                                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, 63)));
}

template <int EL> constexpr int wcs_lds_doubles() { return 64 * (3 * EL + 2); }      // per wave
// the wave's lanes exchange values through LDS: the compiler must keep the accesses on either side in order (the hardware
// completes one wave's LDS operations in order)
__device__ __forceinline__ void wcs_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// x[u][c]: the term of class c at step 64 u + lane of the trip (0 for steps beyond the walk's end).  On return x[u][c] = the running
// sum of class c over the steps BEFORE that one (EXCL) or up to and including it, carried over from the trips before; carry moves on.
// s: the wave's own wcs_lds_doubles<EL>() doubles of LDS (16-byte aligned).  A lane's block is 3 EL + 2 doubles long: consecutive
// lanes' 16-byte reads fall on different banks.
template <int EL, bool EXCL>
__device__ __forceinline__ void wave_class_scan(double *s, double (&x)[EL][3], double (&carry)[3], uint32_t lane)
{
    constexpr int STRIDE = 3 * EL + 2;
#pragma unroll
    for (int u = 0; u < EL; u++) {
        const uint32_t t = 64u * (uint32_t)u + lane;                // step of the trip
        double *d = s + (t / EL) * STRIDE + (t % EL) * 3;
        d[0] = x[u][0]; d[1] = x[u][1]; d[2] = x[u][2];
    }
    wcs_sync();
    double y[EL][3];
    const double *mine = s + lane * STRIDE;
#pragma unroll
    for (int e = 0; e < EL; e++) { y[e][0] = mine[3 * e]; y[e][1] = mine[3 * e + 1]; y[e][2] = mine[3 * e + 2]; }
#pragma unroll
    for (int e = 1; e < EL; e++) { y[e][0] += y[e - 1][0]; y[e][1] += y[e - 1][1]; y[e][2] += y[e - 1][2]; }
    double base[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const double inc = wcs_wave_scan(y[EL - 1][c]);
        base[c] = carry[c] + (inc - y[EL - 1][c]);                  // what the lanes in front of this one and the trips before add up to
        carry[c] += wcs_last_lane(inc);
    }
    wcs_sync();
    double *out = s + lane * STRIDE;
#pragma unroll
    for (int e = 0; e < EL; e++)
#pragma unroll
        for (int c = 0; c < 3; c++)
            out[3 * e + c] = EXCL ? (e ? base[c] + y[e - 1][c] : base[c]) : base[c] + y[e][c];
    wcs_sync();
#pragma unroll
    for (int u = 0; u < EL; u++) {
        const uint32_t t = 64u * (uint32_t)u + lane;
        const double *d = s + (t / EL) * STRIDE + (t % EL) * 3;
        x[u][0] = d[0]; x[u][1] = d[1]; x[u][2] = d[2];
    }
    wcs_sync();
}

#endif
