// bwt_inverse_small.hpp -- inverse BWT of small blocks: one WAVE per block, everything in LDS.
//
// For blocks of at most kInvSmallCap bytes (delimiter included) the general path -- a workgroup
// per block to build the LF table in HBM, then a lane per block walking it through HBM -- is
// mostly launch and latency overhead.  Here each of a workgroup's waves takes its own blocks
// (every (number of waves)-th of the batch) and does the whole of the reference's bwt_inverse
// (src/inverse_bwt.cpp:13-63) without leaving LDS or meeting a workgroup barrier:
//   byte histogram and LAST delimiter row (:23-32), C[] by an exclusive scan (:35-45),
//   occurrence ranks by wave ballot matching in row order, LF[i] = C[L[i]] + occ[i];
//   one lane walks row = LF[row] from the delimiter row, emitting L[row] back to front until it
//   reaches a delimiter (:48-59); the wave then copies the text out, right-aligned in the
//   block's slot like the general kernels.
#pragma once
#include "bwt_device.hpp"
#include "bwt_inverse.hpp"

namespace bwtgpu {

constexpr int kInvSmallCap = 1025;  // bytes per transformed block: block_size <= 1024
constexpr int kInvSmallVecs = (kInvSmallCap + kWave - 1) / kWave;
constexpr int kInvSmallWaves = 4;

struct InvSmallWave {
    uint32_t hist[256];                         // counts, then running offsets C[c] + occurrences so far
    uint16_t lf[kInvSmallVecs * kWave];
    uint8_t L[kInvSmallVecs * kWave];
    uint8_t text[kInvSmallVecs * kWave];        // the walk's output, right-aligned at index n-1
};

__global__ __launch_bounds__(kInvSmallWaves* kWave) void bwt_inverse_small_kernel(InvArgs a) {
    __shared__ InvSmallWave shw[kInvSmallWaves];
    InvSmallWave& sh = shw[wave_id()];
    const int lane = lane_id();
    const uint64_t lt = lanemask_lt();
    // blocks are dealt to the waves round robin: small blocks cost about the same each, and one
    // queue counter for millions of blocks would serialise the whole launch on a single address
    const uint32_t nwaves = gridDim.x * (uint32_t)kInvSmallWaves;
    for (uint32_t b = blockIdx.x * (uint32_t)kInvSmallWaves + (uint32_t)wave_id(); b < a.nblocks; b += nwaves) {
        const uint32_t n = inv_block_len(a, b);
        const uint8_t* __restrict__ L = a.in + (uint64_t)b * ((uint64_t)a.block_size + 1u);
        uint8_t* const o = a.out + (uint64_t)b * a.out_stride;
        const int nv = (int)((n + kWave - 1u) / kWave);  // wave-uniform
        for (int i = lane; i < 256; i += kWave) sh.hist[i] = 0;
        compiler_fence();
        uint32_t drow = kNoRow;
        for (int v = 0; v < nv; ++v) {
            const uint32_t k = (uint32_t)(v * kWave + lane);
            const bool ok = k < n;
            const uint32_t c = ok ? L[k] : 0u;
            sh.L[k] = (uint8_t)c;
            if (ok) atomicAdd(&sh.hist[c], 1u);
            const uint64_t isd = __ballot(ok && c == a.delim);
            if (isd) drow = (uint32_t)(v * kWave) + 63u - (uint32_t)__clzll(isd);
        }
        compiler_fence();
        if (drow == kNoRow) {  // wave-uniform
            if (lane == 0) a.lens[b] = 0xFFFFFFFFu;
            continue;
        }
        {   // C[]: exclusive scan of the 256 counts, 4 per lane
            uint32_t c[4], sum = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                c[i] = sh.hist[4 * lane + i];
                sum += c[i];
            }
            uint32_t run = wave_inclusive_scan(sum) - sum;
            compiler_fence();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sh.hist[4 * lane + i] = run;
                run += c[i];
            }
        }
        compiler_fence();
        for (int v = 0; v < nv; ++v) {  // LF[i] = C[L[i]] + occurrences of L[i] before row i
            const uint32_t k = (uint32_t)(v * kWave + lane);
            const bool ok = k < n;
            const uint32_t c = sh.L[k];
            const uint64_t peers = match_digit<8>(c, ok);
            const uint32_t rk = __popcll(peers & lt);
            const uint32_t b0 = sh.hist[ok ? c : 0u];
            compiler_fence();  // every lane has read its offset before the leaders advance it
            if (ok && rk == 0) sh.hist[c] = b0 + (uint32_t)__popcll(peers);
            compiler_fence();
            sh.lf[k] = (uint16_t)(b0 + rk);
        }
        compiler_fence();
        uint32_t produced = 0;
        if (lane == 0) {  // the chain is serial by nature: one lane follows it through LDS
            uint32_t row = sh.lf[drow];
            uint32_t p = n - 1u;  // bytes are produced back to front; the text ends at index n-1 (exclusive)
            for (uint32_t step = 0; step < n && p > 0; ++step) {
                const uint32_t ch = sh.L[row];
                if (ch == a.delim) break;
                sh.text[--p] = (uint8_t)ch;
                row = sh.lf[row];
            }
            produced = n - 1u - p;
            a.lens[b] = produced;
        }
        produced = (uint32_t)__builtin_amdgcn_readfirstlane((int)produced);
        compiler_fence();
        const uint32_t first = n - 1u - produced;
        for (uint32_t i = first + lane; i < n - 1u; i += kWave) o[i] = sh.text[i];
        compiler_fence();
    }
}

}  // namespace bwtgpu
