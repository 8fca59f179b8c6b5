// Times the grid-wide barrier of bwt_forward_grid_coop.hpp: N barriers with nothing between them, for several grid sizes.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ibwt_amd/csrc tools/microbench/coop_sync_bench.hip -o build/coop_sync_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include "bwt_forward_grid_coop.hpp"
using namespace bwtgpu;
__global__ __launch_bounds__(256) void sync_loop(GridCoopArgs c, int n, int fence_only) {
    __shared__ uint32_t ok;
    GridBarrier gb;
    if (!grid_census(c, gridDim.x, gb, &ok)) return;
    for (int i = 0; i < n; ++i) {
        if (fence_only) { __threadfence(); __syncthreads(); }
        else if (!grid_sync(c, gb, &ok)) return;
    }
}
int main() {
    uint32_t* d; hipMalloc(&d, 4096);
    GridCoopArgs c{}; c.bar = d; c.status = d + 16; c.state = d + 32;
    for (int fence_only = 0; fence_only < 2; ++fence_only)
    for (int nwg : {1, 8, 32, 64, 128, 256, 512}) {
        hipMemset(d, 0, 4096);
        int n = 1000;
        void* args[] = {&c, &n, &fence_only};
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchCooperativeKernel((const void*)sync_loop, dim3(nwg), dim3(256), args, 0, 0);  // warm-up
        hipMemset(d, 0, 4096);
        hipEventRecord(a);
        hipError_t e = hipLaunchCooperativeKernel((const void*)sync_loop, dim3(nwg), dim3(256), args, 0, 0);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%s nwg %4d: %.2f us per barrier (%s)\n", fence_only ? "fence+syncthreads only" : "grid barrier", nwg, ms * 1000 / n, hipGetErrorString(e));
    }
    return 0;
}
