// Microbenchmark: dependent random 4-byte loads (the inverse BWT's LF walk) as a function of
// how much table memory all concurrently running lanes touch together.  Every group of lanes
// chases pointers inside its own table of `tbytes`; `ntables` tables are live at once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

__global__ void fill(uint32_t* t, size_t n, uint32_t per_table) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        t[i] = (uint32_t)(((uint32_t)i * 1664525u + 1013904223u) % per_table);  // full-period LCG step: one cycle through the whole table
}

// lanes_per_table lanes share one table; xcd_local != 0 keeps all workgroups of one table on one XCD
// (workgroup w runs on XCD w % 8).
__global__ __launch_bounds__(256) void chase(const uint32_t* t, uint32_t per_table, uint32_t ntables, uint32_t wgs_per_table,
                                             int steps, int xcd_local, uint32_t* sink) {
    uint32_t wg = blockIdx.x, table;
    if (xcd_local) {
        const uint32_t x = wg & 7u, j = wg >> 3;
        table = (j / wgs_per_table) * 8u + x;
    } else {
        table = wg / wgs_per_table;
    }
    if (table >= ntables) return;
    const uint32_t* my = t + (size_t)table * per_table;
    uint32_t row = hash32(blockIdx.x * 256u + threadIdx.x) % per_table;
    uint32_t acc = 0;
    for (int s = 0; s < steps; ++s) {
        row = my[row];
        acc += row;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const size_t total = (size_t)4 << 30;  // 4 GiB of tables
    uint32_t* d; uint32_t* sink;
    CK(hipMalloc(&d, total)); CK(hipMalloc(&sink, 4));
    const uint32_t per_table = 1u << 18;  // 1 MiB tables (a 256 KiB block's LF table)
    fill<<<4096, 256>>>(d, total / 4, per_table);
    CK(hipDeviceSynchronize());
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int steps = 1024;
    const uint32_t wgs_total = 256 * 8;  // 2048 workgroups of 256 lanes: every CU full
    for (int xl = 0; xl < 2; ++xl)
        for (uint32_t wpt : {1u, 2u, 4u, 8u, 16u, 32u, 64u, 128u}) {
            const uint32_t ntables = wgs_total / wpt;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(a));
                chase<<<wgs_total, 256>>>(d, per_table, ntables, wpt, steps, xl, sink);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            }
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("xcd_local=%d tables live=%5u (%6.0f MiB) lanes/table=%6u : %7.2f ms  %7.1f G loads/s\n", xl, ntables,
                   ntables * (per_table * 4.0) / 1048576.0, wpt * 256, ms, (double)wgs_total * 256 * steps / ms * 1e-6);
        }
    return 0;
}
