// How fast can the fp32 gene rows be READ the way k_orf_walk_sums8 / k_mg_tile_starts read them?  [6][total] floats (total = 5e8: 12 GB); a wave
// takes one (read, strand) of 500 bases: three rows, 2,000 contiguous bytes of each, 32 bytes per lane (two 16-byte loads at 4-byte alignment).
// Against: the same bytes as ONE stream, 16 bytes per lane, lanes contiguous.  Nothing is computed (an xor keeps the loads alive).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/read_rows_probe tools/probes/read_rows_probe.hip && /tmp/read_rows_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

struct __attribute__((packed, aligned(4))) F4 { float v[4]; };

__global__ __launch_bounds__(256) void k_rows(const float *g, uint64_t total, uint64_t n_reads, uint32_t L, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    uint32_t acc = 0;
    for (uint64_t it = wave; it < 2 * n_reads; it += n_waves) {
        const uint64_t off = (it >> 1) * L;
        const float *rows = g + ((it & 1) ? 3 : 0) * total;
        const uint32_t tb = 8u * lane;
        if (tb + 8 <= L) {
            const uint64_t g_lo = (it & 1) ? off + tb : off + L - 8 - tb;
#pragma unroll
            for (int f = 0; f < 3; f++) {
                const F4 a = *(const F4 *)(rows + (uint64_t)f * total + g_lo), b = *(const F4 *)(rows + (uint64_t)f * total + g_lo + 4);
                acc ^= __float_as_uint(a.v[0]) ^ __float_as_uint(a.v[3]) ^ __float_as_uint(b.v[1]) ^ __float_as_uint(b.v[2]);
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_stream(const uint4 *g, uint64_t n16, uint32_t *out)
{
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 v = g[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main()
{
    const uint64_t n_reads = 1000000, L = 500, total = n_reads * L;
    float *g; uint32_t *out;
    if (hipMalloc(&g, 6 * total * 4) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(g, 1, 6 * total * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {256 * 8, 256 * 16, 256 * 64}) {
        float best_r = 1e30f, best_s = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            float ms;
            hipEventRecord(e0); hipLaunchKernelGGL(k_rows, dim3(grid), dim3(256), 0, 0, g, total, n_reads, (uint32_t)L, out); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); if (ms < best_r) best_r = ms;
            hipEventRecord(e0); hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, 0, (const uint4 *)g, 6 * total / 4, out); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); if (ms < best_s) best_s = ms;
        }
        const double gb = 6.0 * total * 4 / 1e9;
        printf("grid %6d x 256: rows as (read, strand) items  %.3f ms = %.2f TB/s (%.1f of %.1f GB touched)   one stream  %.3f ms = %.2f TB/s\n", grid, best_r,
               gb * 496 / 500 / best_r, gb * 496 / 500, gb, best_s, gb / best_s);
    }
    return 0;
}
