// LDS byte look-ups at random addresses inside a table of S bytes (the shift tables of k_frame6t's tree levels: level 5 = 1,024
// entries, level 6 = 4,096; as bytes 1 KB / 4 KB per strand table, nibble-packed half of that): does a smaller footprint lower the
// bank conflicts?  64 lanes, 64 banks of 4 bytes: a look-up conflicts when two lanes hit DIFFERENT dwords of one bank.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_table_probe tools/probes/lds_table_probe.hip && /tmp/lds_table_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int NIBBLE>
__global__ __launch_bounds__(1024) void k_probe(uint32_t size_bytes, uint32_t iters, uint32_t *out)
{
    extern __shared__ uint8_t s_tab[];
    for (uint32_t i = threadIdx.x; i < size_bytes; i += blockDim.x) s_tab[i] = (uint8_t)(i * 37u + 11u);
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u, acc = 0;
    const uint32_t mask = (NIBBLE ? 2u * size_bytes : size_bytes) - 1u;      // entries: bytes, or two per byte
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            x = x * 1664525u + 1013904223u;
            const uint32_t idx = ((x >> 9) ^ acc) & mask;                        // (depends on the value read before: a descent)
            if (NIBBLE) acc = (s_tab[idx >> 1] >> ((idx & 1u) << 2)) & 15u;
            else acc = s_tab[idx];
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main()
{
    uint32_t *d_out;
    hipMalloc(&d_out, 256 * 8 * 1024 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const uint32_t iters = 20000;
    printf("entries  form    bytes   ns per wave look-up (1,024-lane work-groups, 256 x 2 of them, 8 dependent look-ups per step)\n");
    for (int entries = 64; entries <= 8192; entries *= 2)
        for (int nib = 0; nib < 2; nib++) {
            const uint32_t bytes = nib ? entries / 2 : entries;
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                if (nib) hipLaunchKernelGGL(k_probe<1>, dim3(512), dim3(1024), bytes, 0, bytes, iters, d_out);
                else hipLaunchKernelGGL(k_probe<0>, dim3(512), dim3(1024), bytes, 0, bytes, iters, d_out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            // look-ups per CU: 2 work-groups x 16 waves x iters x 8; a CU's LDS serves them one wave-instruction at a time
            const double per = best * 1e6 / (2.0 * 16.0 * iters * 8.0);
            printf("%7d  %-6s %6u   %.2f\n", entries, nib ? "nibble" : "byte", bytes, per);
        }
    return 0;
}
