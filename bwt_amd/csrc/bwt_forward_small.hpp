// bwt_forward_small.hpp -- forward BWT of small blocks: one WAVE suffix-sorts one block in LDS.
//
// The general kernel (bwt_forward.hpp) spends a workgroup, a dozen barriers and several
// dependent trips to HBM scratch on every block, which is what a 256 KiB block needs and far
// too much for a 128-byte one (the reference CLI's default block size, src/bwt.cpp:239-261).
// Here a block of up to kSmallCap symbols never leaves LDS: the 4 waves of a workgroup each
// take every (number of waves)-th block of the batch and run plain prefix doubling
// (src/bwt.cpp:50-99) wave-synchronously, without a single workgroup barrier:
//   rank[i] = byte value + 1 (the appended delimiter by its byte value, 0 = past the end);
//   round h: sort all suffixes by (rank[i], rank[i+h]) -- an in-place LSD radix sort of
//            22-bit keys, three 8-bit passes, stable ranks by wave ballot matching, every
//            lane holding its elements in registers across a pass;
//            new rank = 1 + sorted position of the first suffix with the same key;
//   until all ranks differ (or h >= n); then out[row] = the byte before the row's suffix, the
//   delimiter for suffix 0 (src/bwt.cpp:102-123).
// Same order as the reference (unsigned bytes, shorter suffix first), so the same bytes.
#pragma once
#include "bwt_device.hpp"
#include "bwt_forward.hpp"

namespace bwtgpu {

constexpr int kSmallCap = 1025;                        // symbols per block incl. the delimiter: block_size <= 1024
constexpr int kSmallVecs = (kSmallCap + kWave - 1) / kWave;  // 64-element vectors per block
constexpr int kSmallRankBits = 11;                     // ranks 0 .. 1025
constexpr int kSmallWaves = 4;                         // waves (= blocks in flight) per workgroup

struct SmallWave {
    uint32_t key[kSmallVecs * kWave];   // sort keys, in place
    uint16_t idx[kSmallVecs * kWave];   // the suffix each key belongs to
    uint16_t rank[kSmallVecs * kWave + 4];
    uint32_t hist[256];
    uint8_t txt[kSmallVecs * kWave + 4];
};

__global__ __launch_bounds__(kSmallWaves* kWave) void bwt_forward_small_kernel(FwdArgs a) {
    __shared__ SmallWave shw[kSmallWaves];
    SmallWave& sh = shw[wave_id()];
    const int lane = lane_id();
    const uint64_t lt = lanemask_lt();
    // blocks are dealt to the waves round robin: small blocks cost about the same each, and one
    // queue counter for millions of blocks would serialise the whole launch on a single address
    const uint32_t nwaves = gridDim.x * (uint32_t)kSmallWaves;
    for (uint32_t b = blockIdx.x * (uint32_t)kSmallWaves + (uint32_t)wave_id(); b < a.nblocks; b += nwaves) {
        const uint64_t off = (uint64_t)b * a.block_size;
        const uint64_t rem = a.total_len - off;
        const uint32_t len = rem < a.block_size ? (uint32_t)rem : a.block_size;
        const uint32_t n = len + 1u;
        const uint8_t* __restrict__ s = a.in + off;
        uint8_t* __restrict__ out = a.out + (uint64_t)b * ((uint64_t)a.block_size + 1u);
        const int nv = (int)((n + kWave - 1u) / kWave);  // wave-uniform
        // text and initial ranks
        for (uint32_t i = lane; i < (uint32_t)nv * kWave; i += kWave) {
            const uint32_t c = i < len ? s[i] : a.delim;
            sh.txt[i] = (uint8_t)c;
            sh.rank[i] = (uint16_t)(i < n ? c + 1u : 0u);
        }
        compiler_fence();
        uint32_t kk[kSmallVecs], ii[kSmallVecs];
        for (uint32_t h = 1;; h *= 2u) {
            // keys (rank[i], rank[i+h]) of all suffixes, in index order
#pragma unroll
            for (int v = 0; v < kSmallVecs; ++v) {
                kk[v] = 0;
                ii[v] = 0;
                if (v < nv) {
                    const uint32_t i = (uint32_t)(v * kWave + lane);
                    const uint32_t r1 = i < n ? sh.rank[i] : 0u;
                    const uint32_t r2 = i + h < n ? sh.rank[i + h] : 0u;
                    kk[v] = (r1 << kSmallRankBits) | r2;
                    ii[v] = i;
                }
            }
            // LSD radix sort of (key, index), 8 bits per pass; positions >= n hold key 0 only in
            // lanes that are masked out everywhere below
#pragma unroll 1
            for (int shift = 0; shift < 2 * kSmallRankBits; shift += 8) {
                for (int i = lane; i < 256; i += kWave) sh.hist[i] = 0;
                compiler_fence();
                uint32_t r[kSmallVecs];
#pragma unroll
                for (int v = 0; v < kSmallVecs; ++v) {
                    r[v] = 0;
                    if (v < nv) {
                        const bool ok = (uint32_t)(v * kWave + lane) < n;
                        const uint32_t d = (kk[v] >> shift) & 255u;
                        const uint64_t peers = match_digit<8>(d, ok);
                        const uint32_t rk = __popcll(peers & lt);
                        const uint32_t b0 = sh.hist[ok ? d : 0u];
                        compiler_fence();  // every lane has read the count before the leaders advance it
                        if (ok && rk == 0) sh.hist[d] = b0 + (uint32_t)__popcll(peers);
                        compiler_fence();
                        r[v] = b0 + rk;
                    }
                }
                {   // exclusive scan of the 256 counts: 4 per lane
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
#pragma unroll
                for (int v = 0; v < kSmallVecs; ++v) {
                    if (v < nv && (uint32_t)(v * kWave + lane) < n) {
                        const uint32_t pos = sh.hist[(kk[v] >> shift) & 255u] + r[v];
                        sh.key[pos] = kk[v];
                        sh.idx[pos] = (uint16_t)ii[v];
                    }
                }
                compiler_fence();
#pragma unroll
                for (int v = 0; v < kSmallVecs; ++v) {
                    if (v < nv) {
                        const uint32_t p = (uint32_t)(v * kWave + lane);
                        kk[v] = p < n ? sh.key[p] : 0u;
                        ii[v] = p < n ? sh.idx[p] : 0u;
                    }
                }
                compiler_fence();
            }
            // new ranks: 1 + position of the first element of the run of equal keys
            uint32_t carry = 0;      // position of the last run head seen so far
            bool all_distinct = true;
#pragma unroll
            for (int v = 0; v < kSmallVecs; ++v) {
                if (v < nv) {
                    const uint32_t p = (uint32_t)(v * kWave + lane);
                    const bool ok = p < n;
                    const uint32_t prev = p > 0 && ok ? sh.key[p - 1] : 0xFFFFFFFFu;
                    const bool head = ok && (p == 0 || prev != kk[v]);
                    const uint64_t H = __ballot(head);
                    const uint64_t hm = H & lanemask_le();
                    const uint32_t start = hm ? (uint32_t)(v * kWave) + 63u - (uint32_t)__clzll(hm) : carry;
                    if (H) carry = (uint32_t)(v * kWave) + 63u - (uint32_t)__clzll(H);
                    if (ok) sh.rank[ii[v]] = (uint16_t)(start + 1u);
                    if (__ballot(ok && !head)) all_distinct = false;
                }
            }
            compiler_fence();
            if (all_distinct || h >= n) break;
        }
        // the sorted order is the suffix array: emit the byte before every suffix
#pragma unroll
        for (int v = 0; v < kSmallVecs; ++v) {
            if (v < nv) {
                const uint32_t p = (uint32_t)(v * kWave + lane);
                if (p < n) out[p] = ii[v] ? sh.txt[ii[v] - 1u] : (uint8_t)a.delim;
            }
        }
        compiler_fence();
    }
}

}  // namespace bwtgpu
