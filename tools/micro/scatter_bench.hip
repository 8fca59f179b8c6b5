// Microbenchmark: cost of scattered vs coalesced stores/loads from a persistent workgroup,
// in the access shapes the BWT kernels use (8-byte elements into B bins that each advance
// sequentially; 4-byte and 1-byte random scatters inside a 1 MB window).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// mode 0: 8 B store, bin = hash (random bin per lane), bins B, each bin sequential
// mode 1: 8 B store fully coalesced
// mode 2: 4 B store random position in window
// mode 3: 1 B store random position in window
// mode 4: 4 B load random position in window (sum kept)
// mode 5: 8 B store, bin = hash, but lanes sorted by bin (runs) -> emulates LDS-staged scatter
template <int MODE>
__global__ __launch_bounds__(512) void k(uint8_t* base, size_t region, int iters, int bins, uint32_t* sink) {
    uint8_t* my = base + (size_t)blockIdx.x * region;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t n = (uint32_t)(region / 8);          // 8-byte slots in the region
    const uint32_t seg = n / bins;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t g = it * 512 + tid;              // element index this lane handles
        const uint32_t h = hash32(g * 2654435761u + blockIdx.x);
        if (MODE == 0) {
            const uint32_t d = h % bins;
            const uint32_t pos = d * seg + ((it * 8 + w) * 2 + (lane & 1)) % seg;   // sequential-ish within bin
            reinterpret_cast<uint64_t*>(my)[pos] = ((uint64_t)h << 32) | g;
        } else if (MODE == 1) {
            reinterpret_cast<uint64_t*>(my)[g % n] = ((uint64_t)h << 32) | g;
        } else if (MODE == 2) {
            reinterpret_cast<uint32_t*>(my)[h % (n * 2)] = g;
        } else if (MODE == 3) {
            my[h % (n * 8)] = (uint8_t)g;
        } else if (MODE == 4) {
            acc += reinterpret_cast<uint32_t*>(my)[h % (n * 2)];
        } else if (MODE == 5) {
            // 64 lanes write 8 runs of 8 consecutive slots: run r goes to bin (hash of (it,w,r))
            const uint32_t r = lane >> 3;
            const uint32_t d = hash32((it * 8 + w) * 8 + r + blockIdx.x * 977) % bins;
            const uint32_t pos = d * seg + ((it * 8 + w) * 8 + (lane & 7)) % seg;
            reinterpret_cast<uint64_t*>(my)[pos] = ((uint64_t)h << 32) | g;
        } else if (MODE >= 10) {
            // runs of RL = MODE-10 lanes: run r -> random bin, consecutive slots inside the run
            const uint32_t RL = MODE - 10;
            const uint32_t r = lane / RL;
            const uint32_t d = hash32((it * 8 + w) * 64 + r + blockIdx.x * 977) % bins;
            const uint32_t pos = d * seg + (hash32(it * 8 + w) * RL + (lane % RL)) % seg;
            reinterpret_cast<uint64_t*>(my)[pos] = ((uint64_t)h << 32) | g;
        }
    }
    if (MODE == 4 && acc == 0x12345678) sink[0] = acc;
}

template <int MODE>
int run(const char* name, uint8_t* d, size_t region, int grid, int iters, int bins, uint32_t* sink, int bytes_per_elem) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, d, region, iters, bins, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, d, region, iters, bins, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double elems = (double)grid * 512.0 * iters;
    printf("%-44s grid=%4d bins=%4d  %8.2f ms  %7.1f Gelem/s  %7.1f GB/s\n", name, grid, bins, ms, elems / ms / 1e6, elems * bytes_per_elem / ms / 1e6);
    return 0;
}

int main() {
    const size_t region = 4u << 20;   // 4 MB per workgroup (like the forward scratch of a 256 KiB block)
    const int maxgrid = 1024;
    uint8_t* d; uint32_t* sink;
    CK(hipMalloc(&d, region * maxgrid)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(d, 0, region * maxgrid));
    const int iters = 512;  // 512*512 = 262144 elements per workgroup
    for (int grid : {512, 768}) {
        run<1>("8B store coalesced", d, region, grid, iters, 1, sink, 8);
        run<11>("8B store runs of 1 lane", d, region, grid, iters, 512, sink, 8);
        run<12>("8B store runs of 2 lanes (16 B)", d, region, grid, iters, 512, sink, 8);
        run<14>("8B store runs of 4 lanes (32 B)", d, region, grid, iters, 512, sink, 8);
        run<18>("8B store runs of 8 lanes (64 B)", d, region, grid, iters, 512, sink, 8);
        run<26>("8B store runs of 16 lanes (128 B)", d, region, grid, iters, 512, sink, 8);
        run<42>("8B store runs of 32 lanes (256 B)", d, region, grid, iters, 512, sink, 8);
        run<4>("4B load random in 4 MB", d, region, grid, iters, 1, sink, 4);
    }
    for (int grid : {256, 512}) {
        run<4>("4B load random in 512 KB (L2/MALL resident)", d, 512u << 10, grid, iters, 1, sink, 4);
        run<4>("4B load random in 256 KB", d, 256u << 10, grid, iters, 1, sink, 4);
    }
    return 0;
}
