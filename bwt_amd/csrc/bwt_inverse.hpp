// bwt_inverse.hpp -- batched inverse BWT by LF mapping.
//
// Replaces, for a whole batch of blocks, the serial loop of the reference's
// bwt_inverse_process_file (src/inverse_bwt.cpp:85-95) and bwt_inverse
// (src/inverse_bwt.cpp:13-63):
//   pass 1  (:23-32)  occ[i] = occurrences of L[i] in L[0..i), delimiter_row = LAST i with L[i] == delimiter
//   C[]     (:35-45)  exclusive prefix sum of the byte histogram in unsigned byte order
//   LF walk (:48-59)  row = delimiter_row; loop { row = C[L[row]] + occ[row]; ch = L[row];
//                     stop at the first ch == delimiter; else emit ch }, result reversed (:61-62)
//
// Kernel 1 (one workgroup per block) builds the LF table T[i] = (C[L[i]] + occ[i]) | L[i] << 24:
// an LDS byte histogram per wave chunk, an exclusive scan in (symbol, wave) order, then a
// stable per-symbol ranking of every position (wave-ballot match, the same primitive as the
// forward radix passes).  Packing the symbol next to the successor row makes the walk ONE
// dependent 4-byte load per output byte.
// Kernel 2 walks one independent chain per block, one lane per chain, emitting bytes back
// to front so no reversal pass is needed.
#pragma once
#include "bwt_device.hpp"

namespace bwtgpu {

constexpr uint32_t kNoRow = 0xFFFFFFFFu;
constexpr uint32_t kRowMask = 0x00FFFFFFu;

struct InvArgs {
    const uint8_t* in;    // block b reads in + b*(block_size+1)
    uint8_t* out;         // block b's text ends at out + b*block_size + len_b
    uint64_t total_len;   // bytes of `in` (texts + one delimiter per block)
    uint32_t block_size;
    uint32_t nblocks;
    uint32_t delim;
    uint32_t tstride;     // LF-table words per block (>= block_size+1)
    uint32_t* T;          // nblocks * tstride words
    uint32_t* drow;       // per block: delimiter row or kNoRow
    uint32_t* lens;       // per block: text length or BWTGPU_LEN_NO_DELIMITER
    uint32_t* queue;      // [0] next block index to hand out
};

template <int NT>
struct InvShared {
    static constexpr int W = NT / kWave;
    uint32_t hist[W * 256];
    uint32_t wsum[W];
    uint32_t drow[W];
    uint32_t blk;
};

__device__ __forceinline__ uint32_t inv_block_len(const InvArgs& a, uint32_t b) {
    const uint64_t off = (uint64_t)b * ((uint64_t)a.block_size + 1u);
    const uint64_t rem = a.total_len - off;
    return rem < (uint64_t)a.block_size + 1u ? (uint32_t)rem : a.block_size + 1u;
}

template <int NT>
__global__ __launch_bounds__(NT) void bwt_inverse_build_kernel(InvArgs a) {
    constexpr int W = NT / kWave;
    static_assert(NT >= 256, "one thread per symbol in the scan");
    __shared__ InvShared<NT> sh;
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const uint64_t lt = lanemask_lt();
    for (;;) {
        if (tid == 0) sh.blk = atomicAdd(&a.queue[0], 1u);
        __syncthreads();
        const uint32_t b = sh.blk;
        __syncthreads();
        if (b >= a.nblocks) break;
        const uint32_t n = inv_block_len(a, b);
        const uint8_t* __restrict__ L = a.in + (uint64_t)b * ((uint64_t)a.block_size + 1u);
        uint32_t* __restrict__ T = a.T + (uint64_t)b * a.tstride;

        for (int i = tid; i < W * 256; i += NT) sh.hist[i] = 0;
        __syncthreads();
        Chunking ck;
        ck.init(n, W);
        const uint32_t lo = (uint32_t)w * ck.chunk, hi = min(lo + ck.chunk, n);
        // pass 1: histogram per wave chunk + last delimiter row
        uint32_t last = kNoRow;
        for (uint32_t base = lo; base < hi; base += kWave) {
            const uint32_t k = base + lane;
            const bool valid = k < hi;
            const uint32_t c = valid ? L[k] : 0u;
            if (valid) atomicAdd(&sh.hist[w * 256 + c], 1u);
            const uint64_t isd = __ballot(valid && c == a.delim);
            if (isd) last = base + 63u - (uint32_t)__clzll(isd);
        }
        if (lane == 0) sh.drow[w] = last;
        __syncthreads();
        // C[] and per-wave start offsets: exclusive scan in (symbol major, wave minor) order
        {
            uint32_t tot = 0;
            if (tid < 256)
                for (int i = 0; i < W; ++i) tot += sh.hist[i * 256 + tid];
            uint32_t total;
            uint32_t run = block_exclusive_scan<NT>(tot, sh.wsum, total);
            if (tid < 256)
                for (int i = 0; i < W; ++i) {
                    const uint32_t v = sh.hist[i * 256 + tid];
                    sh.hist[i * 256 + tid] = run;
                    run += v;
                }
        }
        __syncthreads();
        uint32_t drow = kNoRow;  // rows grow with the wave index, so the last hit wins
#pragma unroll
        for (int i = 0; i < W; ++i)
            if (sh.drow[i] != kNoRow) drow = sh.drow[i];
        if (tid == 0) a.drow[b] = drow;
        // pass 2: LF table
        uint32_t* myoff = sh.hist + w * 256;
        for (uint32_t base = lo; base < hi; base += kWave) {
            const uint32_t k = base + lane;
            const bool valid = k < hi;
            const uint32_t c = valid ? L[k] : 0u;
            const uint64_t peers = match_digit<8>(c, valid);
            const uint32_t rk = __popcll(peers & lt);
            const uint32_t b0 = myoff[c];
            compiler_fence();  // every lane has read its offset before the leaders advance it
            if (valid && rk == 0) myoff[c] = b0 + (uint32_t)__popcll(peers);
            compiler_fence();
            if (valid) T[k] = (b0 + rk) | (c << 24);
        }
        __syncthreads();
    }
}

// One chain per block, one lane per chain.  Every step is a single dependent load; the
// loop is bounded by the block length, and a well-formed block ends it exactly there by
// walking back into the delimiter row.  Output bytes are produced back to front; they are
// packed in registers and leave as aligned 16-byte stores (a lane-private store per byte
// would be one scattered write per step).
__global__ __launch_bounds__(kWave) void bwt_inverse_walk_kernel(InvArgs a) {
    const uint32_t b = blockIdx.x * kWave + threadIdx.x;
    if (b >= a.nblocks) return;
    const uint32_t n = inv_block_len(a, b);
    const uint32_t len = n - 1;
    uint32_t row = a.drow[b];
    if (row == kNoRow) {
        a.lens[b] = 0xFFFFFFFFu;
        return;
    }
    const uint32_t* __restrict__ T = a.T + (uint64_t)b * a.tstride;
    uint8_t* const o = a.out + (uint64_t)b * a.block_size;
    uint8_t* addr = o + len;  // lowest address written so far (exclusive end of what is left)
    uint8_t* const floor_addr = o;
    uint32_t acc = 0, cnt = 0;            // bytes of the dword in progress: acc's low byte is *addr
    uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0, dcnt = 0;  // completed dwords, q0 lowest, at daddr + 4 j
    uint8_t* daddr = addr;
    row = T[row] & kRowMask;
    for (uint32_t step = 0; step < n; ++step) {
        const uint32_t wd = T[row];
        const uint32_t ch = wd >> 24;
        if (ch == a.delim || addr == floor_addr) break;
        row = wd & kRowMask;
        acc = (acc << 8) | ch;
        ++cnt;
        --addr;
        if ((reinterpret_cast<uintptr_t>(addr) & 3u) == 0) {
            if (cnt == 4) {
                q3 = q2; q2 = q1; q1 = q0; q0 = acc;
                daddr = addr;
                ++dcnt;
                if ((reinterpret_cast<uintptr_t>(addr) & 15u) == 0) {
                    if (dcnt == 4) {
                        *reinterpret_cast<uint4*>(addr) = make_uint4(q0, q1, q2, q3);
                    } else {
                        uint32_t* d = reinterpret_cast<uint32_t*>(addr);
                        d[0] = q0;
                        if (dcnt > 1) d[1] = q1;
                        if (dcnt > 2) d[2] = q2;
                    }
                    dcnt = 0;
                }
            } else {  // unaligned end of the block's slot: first dword is partial
                for (uint32_t j = 0; j < cnt; ++j) addr[j] = (uint8_t)(acc >> (8 * j));
            }
            acc = 0;
            cnt = 0;
        }
    }
    for (uint32_t j = 0; j < cnt; ++j) addr[j] = (uint8_t)(acc >> (8 * j));
    if (dcnt) {
        uint32_t* d = reinterpret_cast<uint32_t*>(daddr);
        d[0] = q0;
        if (dcnt > 1) d[1] = q1;
        if (dcnt > 2) d[2] = q2;
        if (dcnt > 3) d[3] = q3;
    }
    a.lens[b] = (uint32_t)((o + len) - addr);
}

}  // namespace bwtgpu
