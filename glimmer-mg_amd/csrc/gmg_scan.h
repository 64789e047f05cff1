// gmg_scan.h -- exclusive prefix sums of the counts the (f)-row paths produce (ORFs per read, starts per ORF, lengths of selected
// reads).  Three short launches, no spinning: the sum of every tile of 4,096 items (k_scan_tile_sums), the exclusive sums of those
// (k_scan_sums: one work-group), and the items again with their tile's offset (k_scan_apply) -- the input is read twice (from L2 the
// second time for these sizes) and the output written once; sums are 64-bit inside the kernels whatever the item types are.
// (Replaces a widening pass + the scan library's two launches: 3x the bytes.  A single-pass version with decoupled look-back was
// built first and measured 3x SLOWER than the library on 7.6 M items -- 0.24 against 0.087 ms: with a thousand tiles in flight a
// tile's look-back walks through hundreds of unresolved predecessors, one dependent L2 round trip per 64 of them.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SC_BLOCK 256
#define SC_CHUNKS 4                                     // 16-byte chunks per lane
#define SC_TILE (SC_BLOCK * SC_CHUNKS * 4)

__device__ __forceinline__ uint64_t sc_wave_incl(uint64_t x)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t y = __shfl_up((unsigned long long)x, d);
        if ((int)(threadIdx.x & 63u) >= d) x += y;
    }
    return x;
}
__device__ __forceinline__ uint64_t sc_wave_sum(uint64_t x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor((unsigned long long)x, o);
    return x;
}

// chunk j of lane t of a tile: items base + (j * SC_BLOCK + t) * 4 .. + 3 (16-byte loads; zeros beyond n)
template <typename TIn>
__device__ __forceinline__ void sc_load(const TIn *__restrict__ in, const uint64_t n, const uint64_t base, TIn (&v)[SC_CHUNKS][4])
{
#pragma unroll
    for (int j = 0; j < SC_CHUNKS; j++) {
        const uint64_t i0 = base + ((uint64_t)j * SC_BLOCK + threadIdx.x) * 4;
        if (i0 + 4 <= n) {
            if (sizeof(TIn) == 4) {
                const uint4 q = *(const uint4 *)(const void *)(in + i0);
                v[j][0] = (TIn)q.x; v[j][1] = (TIn)q.y; v[j][2] = (TIn)q.z; v[j][3] = (TIn)q.w;
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) v[j][k] = in[i0 + k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) v[j][k] = i0 + k < n ? in[i0 + k] : (TIn)0;
        }
    }
}

template <typename TIn>
__global__ __launch_bounds__(SC_BLOCK) void k_scan_tile_sums(const TIn *__restrict__ in, const uint64_t n, uint64_t *__restrict__ sums)
{
    __shared__ uint64_t s_w[SC_BLOCK / 64];
    const uint64_t n_tiles = (n + SC_TILE - 1) / SC_TILE;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        TIn v[SC_CHUNKS][4];
        sc_load(in, n, tile * SC_TILE, v);
        uint64_t x = 0;
#pragma unroll
        for (int j = 0; j < SC_CHUNKS; j++) x += (uint64_t)v[j][0] + v[j][1] + v[j][2] + v[j][3];
        x = sc_wave_sum(x);
        __syncthreads();
        if ((threadIdx.x & 63u) == 0) s_w[threadIdx.x >> 6] = x;
        __syncthreads();
        if (threadIdx.x == 0) sums[tile] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    }
}

// sums[0 .. m) -> their exclusive sums, in place; one work-group of four waves (it fits beside the six-frame kernel), four items per lane
static __global__ __launch_bounds__(256) void k_scan_sums(uint64_t *sums, const uint64_t m)
{
    __shared__ uint64_t s_w[4];
    uint64_t carry = 0;
    for (uint64_t b = 0; b < m; b += 1024) {
        const uint64_t i0 = b + 4ull * threadIdx.x;
        uint64_t x[4];
#pragma unroll
        for (int k = 0; k < 4; k++) x[k] = i0 + k < m ? sums[i0 + k] : 0ull;
        const uint64_t mine = x[0] + x[1] + x[2] + x[3], incl = sc_wave_incl(mine);
        __syncthreads();                                // (s_w of the round before is read)
        if ((threadIdx.x & 63u) == 63u) s_w[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint64_t front = carry, all = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4; w++) {
            if (w < (threadIdx.x >> 6)) front += s_w[w];
            all += s_w[w];
        }
        uint64_t p = front + incl - mine;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (i0 + k < m) sums[i0 + k] = p;
            p += x[k];
        }
        carry += all;
    }
}

// out[i] = in[0] + ... + in[i-1], i < n, given every tile's offset
template <typename TIn, typename TOut>
__global__ __launch_bounds__(SC_BLOCK) void k_scan_apply(const TIn *__restrict__ in, TOut *__restrict__ out, const uint64_t n, const uint64_t *__restrict__ sums)
{
    __shared__ uint64_t s_wsum[SC_CHUNKS][SC_BLOCK / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint64_t n_tiles = (n + SC_TILE - 1) / SC_TILE;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t base = tile * SC_TILE, tile_excl = sums[tile];
        TIn v[SC_CHUNKS][4];
        uint64_t incl[SC_CHUNKS];
        sc_load(in, n, base, v);
        __syncthreads();                                // (s_wsum of the tile before is read)
#pragma unroll
        for (int j = 0; j < SC_CHUNKS; j++) {
            incl[j] = sc_wave_incl((uint64_t)v[j][0] + v[j][1] + v[j][2] + v[j][3]);
            if (lane == 63) s_wsum[j][wv] = incl[j];
        }
        __syncthreads();
        uint64_t run = tile_excl;
#pragma unroll
        for (int j = 0; j < SC_CHUNKS; j++) {
            uint64_t front = 0;
#pragma unroll
            for (int w = 0; w < SC_BLOCK / 64; w++) {
                if ((uint32_t)w == wv) front = run;
                run += s_wsum[j][w];
            }
            const uint64_t i0 = base + ((uint64_t)j * SC_BLOCK + tid) * 4;
            uint64_t p = front + incl[j] - ((uint64_t)v[j][0] + v[j][1] + v[j][2] + v[j][3]);
            if (i0 + 4 <= n && sizeof(TOut) == 8) {
                ulonglong2 a, b;
                a.x = p; a.y = p + v[j][0]; b.x = a.y + v[j][1]; b.y = b.x + v[j][2];
                *(ulonglong2 *)(void *)(out + i0) = a;
                *(ulonglong2 *)(void *)(out + i0 + 2) = b;
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (i0 + k < n) out[i0 + k] = (TOut)p;
                    p += v[j][k];
                }
            }
        }
    }
}

// Exclusive sums of d_in[0 .. n) into d_out[0 .. n) on stream s (d_in and d_out 16-byte aligned); no synchronisation.
template <typename TIn, typename TOut>
static hipError_t gmg_scan_excl(const TIn *d_in, TOut *d_out, uint64_t n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    const uint64_t n_tiles = (n + SC_TILE - 1) / SC_TILE;
    uint64_t *d_sums = nullptr;
    hipError_t e = gmg_pool_alloc((void **)&d_sums, n_tiles * 8);
    if (e != hipSuccess) return e;
    const unsigned grid = (unsigned)(n_tiles < 256 * 16 ? n_tiles : 256 * 16);
    hipLaunchKernelGGL((k_scan_tile_sums<TIn>), dim3(grid), dim3(SC_BLOCK), 0, s, d_in, n, d_sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, s, d_sums, n_tiles);
    hipLaunchKernelGGL((k_scan_apply<TIn, TOut>), dim3(grid), dim3(SC_BLOCK), 0, s, d_in, d_out, n, (const uint64_t *)d_sums);
    e = hipGetLastError();
    gmg_pool_release_after(d_sums, s);
    return e;
}
