// bwt_forward.hpp -- batched forward BWT: one workgroup suffix-sorts one block.
//
// Replaces, for a whole batch of blocks per launch, the reference's worker pool
// (src/bwt.cpp:143-156), bwt_forward (src/bwt.cpp:102-123) and build_suffix_array
// (src/bwt.cpp:50-99).  Output bytes are identical because the suffix array of a
// string is unique under the reference's order: unsigned byte values, and a suffix
// that runs past the end is smaller (src/bwt.cpp:66-80).
//
// Algorithm (per block, n = len + 1 symbols including the appended delimiter)
//   0. census: which byte values occur -> dense order-preserving codes 1..sigma
//      (0 is reserved for "past the end"), sbits = bits per code.
//   1. round 0: element i = (first m0 codes of suffix i, packed) << ibits | i;
//      LSD radix sort of all n elements on the packed key.
//   2. rerank: equal-key runs are groups; rank[i] = position of its group's first
//      element (so a final rank IS the suffix-array position); suffixes alone in
//      their group are finished and leave the working set (wave-ballot + prefix-
//      count compaction); survivors get a dense group id g.
//   3. round r >= 1 (prefix length h doubles): element = g | rank[i+h]+1 | i for the
//      survivors only; LSD radix sort on (g, rank[i+h]); rerank inside the old
//      groups; repeat until no survivor is left.
//   4. gather: out[rank[i]] = s[i-1] (the delimiter for i == 0, which lands on the
//      row of suffix 0 -- the block's primary index is that row).
//
// Every array of a block lives in a per-workgroup slice of HBM scratch (two u64
// element buffers and one u32 rank array); LDS holds the per-wave digit histograms.
#pragma once
#include "bwt_device.hpp"

namespace bwtgpu {

struct FwdArgs {
    const uint8_t* in;    // block b reads in + b*block_size
    uint8_t* out;         // block b writes out + b*(block_size+1)
    uint64_t total_len;
    uint32_t block_size;
    uint32_t nblocks;
    uint32_t ncap;        // elements per scratch array (>= block_size+1, multiple of 64)
    uint32_t delim;
    uint8_t* ws;          // workgroup w owns ws + w*ws_stride
    uint64_t ws_stride;
    uint32_t* queue;      // [0] next block index to hand out, [1] error flags
};

constexpr uint32_t kErrNotConverged = 1u;
constexpr uint32_t kErrBounds = 2u;

template <int NT>
struct FwdShared {
    static constexpr int W = NT / kWave;
    uint32_t histA[W * kBins];
    uint32_t histB[W * kBins];
    uint32_t sym[256];
    uint32_t wsum[W];
    uint32_t cntA[W], cntG[W], lastH[W], lastG[W];
    uint32_t blk;
};

template <int NT>
__device__ __forceinline__ void zero_hist(uint32_t* hist) {
    constexpr int W = NT / kWave;
    for (int i = threadIdx.x; i < W * kBins; i += NT) hist[i] = 0;
}

// Turns per-wave digit counts hist[w][d] into scatter offsets: exclusive prefix sum in
// (digit major, wave minor) order, which is what makes the pass stable.
template <int NT>
__device__ __forceinline__ void scan_hist(uint32_t* hist, uint32_t* s_wsum) {
    constexpr int W = NT / kWave;
    constexpr int PER = (kBins + NT - 1) / NT;  // bins per thread
    const int first = threadIdx.x * PER;
    uint32_t tot = 0;
    if (first < kBins) {
#pragma unroll
        for (int j = 0; j < PER; ++j)
            for (int w = 0; w < W; ++w) tot += hist[w * kBins + first + j];
    }
    uint32_t total;
    uint32_t run = block_exclusive_scan<NT>(tot, s_wsum, total);
    if (first < kBins) {
#pragma unroll
        for (int j = 0; j < PER; ++j)
            for (int w = 0; w < W; ++w) {
                const uint32_t v = hist[w * kBins + first + j];
                hist[w * kBins + first + j] = run;
                run += v;
            }
    }
    __syncthreads();
}

// One stable LSD pass: scatter src[0..m) to dst by digit (e >> shift) & mask, and count
// the NEXT pass's digit per destination wave chunk on the fly (so every pass reads and
// writes each element exactly once).
template <int NT>
__device__ __forceinline__ void radix_pass(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst,
                                           uint32_t m, const Chunking& ck, int shift, int bits,
                                           bool has_next, int nshift, int nbits, uint32_t* off,
                                           uint32_t* nhist, uint32_t* err) {
    const int lane = lane_id(), w = wave_id();
    const uint32_t lo = (uint32_t)w * ck.chunk;
    const uint32_t hi = min(lo + ck.chunk, m);
    volatile uint32_t* myoff = off + w * kBins;
    const uint32_t mask = (1u << bits) - 1u, nmask = (1u << nbits) - 1u;
    const uint64_t lt = lanemask_lt();
    for (uint32_t base = lo; base < hi; base += kWave) {
        const uint32_t k = base + lane;
        const bool valid = k < hi;
        const uint64_t e = valid ? src[k] : 0ull;
        const uint32_t d = (uint32_t)(e >> shift) & mask;
        const uint64_t peers = match_digit(d, bits, valid);
        const uint32_t rk = __popcll(peers & lt);
        const uint32_t cnt = __popcll(peers);
        uint32_t b0 = 0;
        if (valid) b0 = myoff[d];
        if (valid && rk == 0) myoff[d] = b0 + cnt;
        if (valid) {
            const uint32_t pos = b0 + rk;
            if (pos < m) {
                dst[pos] = e;
                if (has_next) {
                    const uint32_t nd = (uint32_t)(e >> nshift) & nmask;
                    atomicAdd(&nhist[ck.wave_of(pos) * kBins + nd], 1u);
                }
            } else {
                atomicOr(err, kErrBounds);  // unreachable unless the histogram is inconsistent
            }
        }
    }
}

// Sorts a[0..m) by the K key bits starting at bit key_shift.  On entry hA holds the
// per-wave counts of pass 0's digit (chunked by Chunking(m)).  Returns the buffer that
// holds the sorted elements (a or b).
template <int NT>
__device__ __forceinline__ uint64_t* radix_sort(uint64_t* a, uint64_t* b, uint32_t m, int key_shift,
                                                const DigitPlan& plan, uint32_t* hA, uint32_t* hB,
                                                uint32_t* s_wsum, uint32_t* err) {
    Chunking ck;
    ck.init(m, NT / kWave);
    for (int p = 0; p < plan.P; ++p) {
        const bool has_next = p + 1 < plan.P;
        if (has_next) zero_hist<NT>(hB);
        scan_hist<NT>(hA, s_wsum);  // its barriers also publish the zeroed hB
        radix_pass<NT>(a, b, m, ck, key_shift + plan.shift(p), plan.bits(p), has_next,
                       key_shift + plan.shift(has_next ? p + 1 : p), plan.bits(has_next ? p + 1 : p), hA, hB, err);
        __syncthreads();
        uint64_t* t = a; a = b; b = t;
        uint32_t* th = hA; hA = hB; hB = th;
    }
    return a;
}

struct RunFlags {
    uint64_t e;
    bool valid, head, tail, ghead;
};

// Flags of sorted position k: head/tail of its equal-key run, head of its old group.
template <bool ROUND0>
__device__ __forceinline__ RunFlags run_flags(const uint64_t* __restrict__ cur, uint32_t k, uint32_t hi,
                                              uint32_t m, int ibits, int gshift) {
    RunFlags f;
    f.valid = k < hi;
    uint64_t e = 0, ep = 0, en = 0;
    if (f.valid) {
        e = cur[k];
        if (k > 0) ep = cur[k - 1];
        if (k + 1 < m) en = cur[k + 1];
    }
    const uint64_t key = e >> ibits;
    f.head = f.valid && (k == 0 || (ep >> ibits) != key);
    f.tail = f.valid && (k + 1 == m || (en >> ibits) != key);
    f.ghead = f.valid && (k == 0 || (!ROUND0 && (ep >> gshift) != (e >> gshift)));
    f.e = e;
    return f;
}

// After a sort: assign positional ranks, drop finished suffixes, emit the survivors
// (dense group id << gshift | index) to nxt.  Returns survivors in m_next and their
// group count in g_next.
template <int NT, bool ROUND0>
__device__ __forceinline__ void rerank(const uint64_t* __restrict__ cur, uint64_t* __restrict__ nxt,
                                       uint32_t* __restrict__ rank, uint32_t m, int ibits, int gshift,
                                       FwdShared<NT>& sh, uint32_t& m_next, uint32_t& g_next) {
    constexpr int W = NT / kWave;
    const int lane = lane_id(), w = wave_id();
    Chunking ck;
    ck.init(m, W);
    const uint32_t lo = (uint32_t)w * ck.chunk;
    const uint32_t hi = min(lo + ck.chunk, m);
    const uint64_t lt = lanemask_lt(), le = lanemask_le();
    const uint64_t imask = (1ull << ibits) - 1ull;

    // phase A: per wave chunk, count survivors / surviving groups, find the last heads
    {
        uint32_t cA = 0, cG = 0, lH = 0, lG = 0;
        for (uint32_t base = lo; base < hi; base += kWave) {
            const RunFlags f = run_flags<ROUND0>(cur, base + lane, hi, m, ibits, gshift);
            const uint64_t H = __ballot(f.head), G = __ballot(f.ghead);
            cA += __popcll(__ballot(f.valid && !(f.head && f.tail)));
            cG += __popcll(__ballot(f.head && !f.tail));
            if (H) lH = base + 63u - (uint32_t)__clzll(H);
            if (G) lG = base + 63u - (uint32_t)__clzll(G);
        }
        if (lane == 0) {
            sh.cntA[w] = cA;
            sh.cntG[w] = cG;
            sh.lastH[w] = lH;
            sh.lastG[w] = lG;
        }
    }
    __syncthreads();
    // phase B: carries into this wave's chunk
    uint32_t baseA = 0, baseG = 0, carryH = 0, carryG = 0, totA = 0, totG = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const uint32_t a = sh.cntA[i], g = sh.cntG[i];
        if (i < w) {
            baseA += a;
            baseG += g;
            carryH = max(carryH, sh.lastH[i]);
            carryG = max(carryG, sh.lastG[i]);
        }
        totA += a;
        totG += g;
    }
    // phase C: ranks + compaction
    for (uint32_t base = lo; base < hi; base += kWave) {
        const RunFlags f = run_flags<ROUND0>(cur, base + lane, hi, m, ibits, gshift);
        const uint64_t H = __ballot(f.head), G = __ballot(f.ghead);
        const uint64_t hm = H & le, gm = G & le;
        const uint32_t khead = hm ? base + 63u - (uint32_t)__clzll(hm) : carryH;
        const uint32_t kg = gm ? base + 63u - (uint32_t)__clzll(gm) : carryG;
        if (H) carryH = base + 63u - (uint32_t)__clzll(H);
        if (G) carryG = base + 63u - (uint32_t)__clzll(G);
        const bool act = f.valid && !(f.head && f.tail);
        const uint64_t A = __ballot(act), NS = __ballot(f.head && !f.tail);
        const uint32_t idx = (uint32_t)(f.e & imask);
        if (f.valid) {
            if (ROUND0) {
                rank[idx] = khead;
            } else {
                const uint32_t delta = khead - kg;
                if (delta) rank[idx] += delta;
            }
        }
        if (act) {
            const uint32_t pos = baseA + __popcll(A & lt);
            const uint32_t g = baseG + __popcll(NS & le) - 1u;
            nxt[pos] = ((uint64_t)g << gshift) | idx;
        }
        baseA += __popcll(A);
        baseG += __popcll(NS);
    }
    m_next = totA;
    g_next = totG;
    __syncthreads();
}

template <int NT>
__device__ __forceinline__ void forward_block(const uint8_t* __restrict__ s, uint32_t len, uint32_t delim,
                                              uint8_t* __restrict__ out, uint64_t* bufA, uint64_t* bufB,
                                              uint32_t* rank, FwdShared<NT>& sh, uint32_t* queue) {
    constexpr int W = NT / kWave;
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const uint32_t n = len + 1;
    if (n == 1) {  // empty block: just the delimiter (reference tests/test_small.cpp:83)
        if (tid == 0) out[0] = (uint8_t)delim;
        return;
    }
    // ---- 0. census and codes ------------------------------------------------
    if (tid < 256) sh.sym[tid] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < len; i += NT) sh.sym[s[i]] = 1;  // all writers store the same value
    if (tid == 0) sh.sym[delim] = 1;
    __syncthreads();
    const uint32_t present = tid < 256 ? sh.sym[tid] : 0;
    uint32_t sigma;
    const uint32_t below = block_exclusive_scan<NT>(present, sh.wsum, sigma);
    if (tid < 256) sh.sym[tid] = present ? below + 1 : 0;
    const int sbits = bit_length(sigma);
    const int ibits = bit_length(n - 1);
    int m0 = (2 * ibits) / sbits;
    const int m0cap = (64 - ibits) / sbits;
    if (m0 > m0cap) m0 = m0cap;
    if (m0 < 1) m0 = 1;
    DigitPlan plan;
    plan.init(m0 * sbits);
    zero_hist<NT>(sh.histA);
    __syncthreads();

    // ---- 1. round 0: packed m0-symbol keys ---------------------------------------
    {
        Chunking ck;
        ck.init(n, W);
        const uint32_t lo = (uint32_t)w * ck.chunk, hi = min(lo + ck.chunk, n);
        const uint32_t dmask = (1u << plan.bits(0)) - 1u;
        for (uint32_t base = lo; base < hi; base += kWave) {
            const uint32_t i = base + lane;
            if (i < hi) {
                uint64_t key = 0;
                for (int j = 0; j < m0; ++j) {
                    const uint32_t p = i + j;
                    uint32_t c = 0;
                    if (p < n) c = sh.sym[p < len ? s[p] : delim];
                    key = (key << sbits) | c;
                }
                bufA[i] = (key << ibits) | i;
                atomicAdd(&sh.histA[w * kBins + ((uint32_t)key & dmask)], 1u);
            }
        }
    }
    __syncthreads();
    uint64_t* cur = radix_sort<NT>(bufA, bufB, n, ibits, plan, sh.histA, sh.histB, sh.wsum, &queue[1]);
    uint64_t* alt = (cur == bufA) ? bufB : bufA;
    // element layout for rounds >= 1: g | r2 | i, where r2 = rank[i+h] + 1, or 0 when the
    // suffix ends exactly at i+h (possible only if the delimiter also occurs inside the
    // block): "past the end" orders before every real rank, as in src/bwt.cpp:77-79
    const int rbits = bit_length(n);
    const int gshift = ibits + rbits;
    uint32_t m, G;
    rerank<NT, true>(cur, alt, rank, n, ibits, gshift, sh, m, G);

    // ---- 2. doubling rounds over the survivors -----------------------------------
    uint32_t h = (uint32_t)m0;
    for (int round = 1; m > 0; ++round) {
        if (round > 48) {  // cannot happen: h >= n is reached after at most 21 doublings
            if (tid == 0) atomicOr(&queue[1], kErrNotConverged);
            break;
        }
        const int gbits = max(1, bit_length(G - 1));
        plan.init(rbits + gbits);
        zero_hist<NT>(sh.histA);
        __syncthreads();
        {
            Chunking ck;
            ck.init(m, W);
            const uint32_t lo = (uint32_t)w * ck.chunk, hi = min(lo + ck.chunk, m);
            const uint32_t dmask = (1u << plan.bits(0)) - 1u;
            const uint64_t imask = (1ull << ibits) - 1ull;
            for (uint32_t base = lo; base < hi; base += kWave) {
                const uint32_t k = base + lane;
                if (k < hi) {
                    uint64_t e = alt[k];
                    const uint32_t idx = (uint32_t)(e & imask);
                    const uint32_t r2 = (idx + h < n) ? rank[idx + h] + 1u : 0u;
                    e |= (uint64_t)r2 << ibits;
                    alt[k] = e;
                    atomicAdd(&sh.histA[w * kBins + ((uint32_t)(e >> ibits) & dmask)], 1u);
                }
            }
        }
        __syncthreads();
        uint64_t* sorted = radix_sort<NT>(alt, cur, m, ibits, plan, sh.histA, sh.histB, sh.wsum, &queue[1]);
        uint64_t* other = (sorted == alt) ? cur : alt;
        uint32_t m2, G2;
        rerank<NT, false>(sorted, other, rank, m, ibits, gshift, sh, m2, G2);
        cur = sorted;
        alt = other;
        m = m2;
        G = G2;
        h = (h >= 0x40000000u) ? 0x80000000u : h * 2u;
    }
    __syncthreads();
    // ---- 3. gather --------------------------------------------------------------
    for (uint32_t i = tid; i < n; i += NT) {
        const uint32_t r = rank[i];
        if (r < n) out[r] = (i == 0) ? (uint8_t)delim : s[i - 1];
        else atomicOr(&queue[1], kErrBounds);
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void bwt_forward_kernel(FwdArgs a) {
    __shared__ FwdShared<NT> sh;
    uint64_t* bufA = reinterpret_cast<uint64_t*>(a.ws + (uint64_t)blockIdx.x * a.ws_stride);
    uint64_t* bufB = bufA + a.ncap;
    uint32_t* rank = reinterpret_cast<uint32_t*>(bufB + a.ncap);
    for (;;) {
        if (threadIdx.x == 0) sh.blk = atomicAdd(&a.queue[0], 1u);
        __syncthreads();
        const uint32_t b = sh.blk;
        __syncthreads();
        if (b >= a.nblocks) break;
        const uint64_t off = (uint64_t)b * a.block_size;
        const uint64_t rem = a.total_len - off;
        const uint32_t len = rem < a.block_size ? (uint32_t)rem : a.block_size;
        forward_block<NT>(a.in + off, len, a.delim, a.out + (uint64_t)b * ((uint64_t)a.block_size + 1u), bufA, bufB,
                          rank, sh, a.queue);
        __syncthreads();
    }
}

}  // namespace bwtgpu
