// bwt_forward.hpp -- batched forward BWT: one workgroup suffix-sorts one block.
//
// Replaces, for a whole batch of blocks per launch, the reference's worker pool
// (src/bwt.cpp:143-156), bwt_forward (src/bwt.cpp:102-123) and build_suffix_array
// (src/bwt.cpp:50-99).  Output bytes are identical because the suffix array of a
// string is unique under the reference's order: unsigned byte values, and a suffix
// that runs past the end is smaller (src/bwt.cpp:66-80).
//
// Algorithm (per block, n = len + 1 symbols including the appended delimiter)
//   0. census: which byte values occur in the text -> dense order-preserving codes
//      0..sigma-1, sbits = bits per code (the delimiter and the end get no code: round0_key).
//   1. round 0 sorts all n suffixes by their first m0 symbols.  Element i = key << (ibits+8) |
//      preceding byte << ibits | i, the key built by a rolling window over the input bytes.
//      A first sweep counts the keys per fine bin (top 13 key bits); the bins are grouped into
//      buckets of contiguous key ranges and about equal population; a second sweep generates
//      the elements again and partitions them into the buckets (one pass through HBM, runs of
//      consecutive stores staged in LDS); every bucket is then loaded into LDS once and sorted
//      there by the key bits its range leaves open (LSD, 4-bit digits, per-thread counters).
//   2. rerank, on the bucket while it is still in LDS: equal-key runs are groups; rank[i] =
//      row of its group's first element (so a final rank IS the suffix-array position);
//      suffixes alone in their group are finished and leave the working set (wave-ballot +
//      prefix-count compaction); survivors get a dense group id g.
//   3. round r >= 1 (prefix length h doubles): element = g | rank[i+h]+1 | i for the
//      survivors only, ordered inside each old group by rank[i+h]: chunk by chunk inside LDS
//      while the groups are small (cut_chunks / rerank_chunk), by LSD radix passes through HBM
//      with a separate rerank sweep otherwise; repeat until no survivor is left.  While few
//      suffixes survive round 0 their ranks are looked up by binary search in the sorted
//      round-0 array instead of being written for everybody.
//   4. output: a suffix that becomes alone in its group is finished; its rank is its row and
//      it stores the byte that precedes it there (the delimiter for suffix 0 -- that row is
//      the block's primary index).  Round-0 finishers carry that byte in their element and
//      store it coalesced; later finishers fetch it from the text.
//
// Every array of a block lives in a per-workgroup slice of HBM scratch (two u64 element
// buffers, one u32 rank array, a small bucket directory); LDS holds the staging tiles, the
// bucket being sorted and the counters.
#pragma once
#ifndef BWT_NOINLINE_SORT
#define BWT_NOINLINE_SORT 0
#endif
#include <type_traits>

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
    uint32_t* queue;      // [0] next block index to hand out, [1] error flags, [2] blocks put off, [3] next put-off
                          // block to hand out, [4] next entry of `normal` to hand out, [5] entries of `normal`,
                          // [6] next block to classify, [7] next block for the text-in-LDS launch, [16..] phase clocks
    uint32_t* deferred;   // nblocks words: the blocks the first launch put off for the second
    uint32_t* normal;     // nblocks words: with a classifying launch first, the blocks it leaves to the first launch
};

constexpr uint32_t kErrNotConverged = 1u;
constexpr uint32_t kErrBounds = 2u;
constexpr int kMaxM0 = 17;    // symbols per round-0 key (bounded by the 24-byte rolling window)
constexpr int kRun = 8;       // consecutive positions one lane rolls through in round 0
constexpr int kTile = kRun * kWave;  // positions per wave iteration in round 0
// Sizes that depend on the workgroup size.  256 threads (three workgroups per CU) is the batch
// configuration; 1024 threads (one workgroup per CU) serves inputs of only a few blocks, where a
// block's latency is all that matters.
template <int NT>
struct FwdCfg {
    // 64-element vectors a wave stages per tile in a radix pass (at least kRun: round 0 stages kRun vectors)
    static constexpr int kTileVecs = NT <= 256 ? 13 : 8;
    static constexpr int kTileElems = kTileVecs * kWave;
    // elements per thread in an LDS bucket sort (odd, see lds_sort_bucket)
    static constexpr int kSortVecs = NT <= 256 ? 17 : 5;
};
constexpr int kFineBits = 13;        // round 0: the top bits of the key that choose a fine bin
constexpr int kFine = 1 << kFineBits;
constexpr int kLdsDigitBits = 4;     // digit width of the LDS bucket sort
constexpr int kMaxBuckets = 256;     // round-0 buckets per block (bucket ids are bytes in the lookup table)
// after the rank array: the bucket directory and the first sorted row of every fine bin
constexpr size_t kBucketDirBytes = (2 * (kMaxBuckets + 1) + kFine + 1) * sizeof(uint32_t);
constexpr uint32_t kBucketTarget = 4100;  // elements per bucket the partition aims at (64 buckets cover 256 KiB + 1)

// Per-phase time accounting (100 MHz constant clock), summed over workgroups into
// queue[16..] as u64: 0 census, 1 round-0 fine-bin count, 2 round-0 partition, 3 round-0 rerank
// sweep (when not fused), 4 later key builds, 5 later sorts, 6 later reranks, 7 full-rank
// rebuilds, 8 blocks, 9 later rounds, 10 round-0 rerank of buckets in LDS (or their plain
// write-back), 11 round-0 bucket sorts through HBM, 12 round-0 bucket sorts in LDS, 13 / 14 later
// rounds' chunk sorts / chunk reranks in LDS, 15 periodic-run resolution (of which, for lists of
// many short groups: 16 the vote, 17 the next-break table, 18 the settling sweep; the rest of 15 is
// the compaction and the table variant), 19 the search for a whole-block period, 20 the expansion of
// a periodic block's output from its last periods, 21 / 22 blocks whose short stretch was given up for
// two full periods (a periodic-run step was due / two heads were still tied).
struct PhaseClock {
    unsigned long long t;
    unsigned long long* acc;
    __device__ __forceinline__ void start(uint32_t* queue) {
        acc = reinterpret_cast<unsigned long long*>(queue + 16);
        t = wall_clock64();
    }
    __device__ __forceinline__ void lap(int phase) {
        if (threadIdx.x == 0) {
            const unsigned long long now = wall_clock64();
            atomicAdd(&acc[phase], now - t);
            t = now;
        }
    }
    __device__ __forceinline__ void count(int slot, unsigned long long v) {
        if (threadIdx.x == 0) atomicAdd(&acc[slot], v);
    }
};

template <int NT>
struct FwdShared {
    static constexpr int W = NT / kWave;
    static constexpr int kTileElems = FwdCfg<NT>::kTileElems;
    static constexpr int kSortCap = NT * FwdCfg<NT>::kSortVecs;  // largest bucket the LDS sort takes
    static_assert(FwdCfg<NT>::kTileVecs >= kRun && kTileElems <= 1024, "round 0 stages kRun vectors; tile ranks have 10 bits");
    struct Staging {
        uint64_t tin[W][kTileElems];    // per-wave staging tile, elements in arrival order
        uint16_t trk[W][kTileElems];    // per element: digit << 10 | rank among the tile's elements with that digit
        uint16_t perm[W][kTileElems];   // tile position after grouping by digit -> arrival slot
        uint32_t tcnt[W][kBins];        // per-wave, per-tile digit counts, then tile-local run starts
        uint32_t tdel[W][kBins];        // per-wave: global slot of a run's first element minus its tile index
    };
    union {
        Staging st;                     // radix passes through HBM
        uint32_t fine[kFine];           // round 0: element count per fine bin
        uint64_t lsort[kSortCap];       // one bucket being sorted inside LDS
    } u;
    union {
        uint8_t lut[kFine];             // round 0: fine bin -> bucket id
        uint32_t cw[8 * NT + NT / 4];   // LDS bucket sort: 8 counter words per thread; word L of the (row, thread) order sits at L + L / 32
    } v;
    uint32_t hist[2][W * kBins];    // per-wave digit counts / scatter offsets, double buffered
    uint8_t sym[256];               // byte value -> code
    uint32_t wsum[W];
    uint32_t wsum2[2][W];  // the in-LDS sort alternates between the two (one barrier less per pass)
    uint32_t cntA[W], cntG[W], lastH[W], lastG[W];
    uint32_t chunk_bad;             // a doubling round's survivors cannot be cut into LDS-sized chunks of whole groups
    unsigned long long vote;        // periodic runs: members << 32 | step of the most populous long step
    uint32_t tailrow[kMaxM0 + 1];   // final rows of the suffixes that reach the delimiter within m0 symbols
    uint32_t blk;
    uint32_t period_cand;           // whole-block period: first place the block's first 16 bytes occur again
    uint32_t period_bad;            // ... and whether the text fails to repeat at that distance somewhere
};

template <int NT>
__device__ __forceinline__ void zero_hist(FwdShared<NT>& sh, int sel) {
    constexpr int W = NT / kWave;
    for (int i = threadIdx.x; i < W * kBins; i += NT) sh.hist[sel][i] = 0;
}

// Turns per-wave digit counts hist[w][d] into scatter offsets: exclusive prefix sum in
// (digit major, wave minor) order, which is what makes a pass stable.
template <int NT>
__device__ __forceinline__ void scan_hist(FwdShared<NT>& sh, int sel) {
    constexpr int W = NT / kWave;
    constexpr int PER = (kBins + NT - 1) / NT;  // bins per thread
    uint32_t* hist = sh.hist[sel];
    const int first = threadIdx.x * PER;
    uint32_t tot = 0;
    if (first < kBins) {
#pragma unroll
        for (int j = 0; j < PER; ++j)
#pragma unroll
            for (int w = 0; w < W; ++w) tot += hist[w * kBins + first + j];
    }
    uint32_t total;
    uint32_t run = block_exclusive_scan<NT>(tot, sh.wsum, total);
    if (first < kBins) {
#pragma unroll
        for (int j = 0; j < PER; ++j)
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const uint32_t v = hist[w * kBins + first + j];
                hist[w * kBins + first + j] = run;
                run += v;
            }
    }
    __syncthreads();
}

// Digit of an element for one pass over key bits: ((x >> kshift) - base) >> shift, masked.
// `base` lets a pass over a sub-range of the key space skip the bits all its keys share.
struct KeyDigit {
    int kshift, shift;
    uint32_t mask;
    uint64_t base;
    __device__ __forceinline__ uint32_t operator()(uint64_t x) const {
        return (uint32_t)(((x >> kshift) - base) >> shift) & mask;
    }
};

// Digit of a round-0 element for a pass over its bucket id (looked up from the key's fine bin).
struct BucketDigit {
    const uint8_t* lut;
    int fshift;  // element >> fshift = fine bin
    int shift;
    uint32_t mask;
    __device__ __forceinline__ uint32_t operator()(uint64_t x) const {
        return ((uint32_t)lut[x >> fshift] >> shift) & mask;
    }
};

// One wave scatters the tile it has staged in LDS (tin[64 v + lane], v < nvec, in arrival
// order; valid_of(v) tells whether this lane's element of vector v exists) by `digit`, stably:
//   1. stable rank of every element among the tile's elements with the same digit: ballot match
//      inside a vector, running per-digit counts (one LDS word per digit) across vectors;
//   2. exclusive scan of the 64 digit counts across the wave's 64 lanes -> where each digit's run
//      starts inside the tile; the wave's global per-digit cursors advance by the counts;
//   3. perm[run start + rank] = arrival slot: the tile's order after grouping by digit;
//   4. the tile leaves in that order: lane i's element goes to (global cursor of its digit) +
//      (i - run start), so consecutive lanes write consecutive slots -- whole runs (100 B on
//      average) instead of lane-scattered 8-byte stores.
// SHARED_CURSORS: all waves of the workgroup claim space from one cursor per digit
// (hist[sel][digit], atomically) instead of from per-wave cursors prepared by a counting pass;
// the result is grouped by digit but not stable across waves.
// The NEXT pass's digit is counted per destination wave chunk on the way out.
// All loops are rolled and everything lives in LDS, so the register footprint stays small.
template <int NT, bool SHARED_CURSORS, typename ValidOf, typename DigitOf, typename NextOf>
__device__ __forceinline__ void scatter_staged(int nvec, ValidOf valid_of, DigitOf digit_of, NextOf next_of,
                                               uint64_t* __restrict__ dst, uint32_t m, bool has_next,
                                               const Chunking& ck, FwdShared<NT>& sh, int sel, uint32_t* err) {
    const int lane = lane_id(), w = wave_id();
    uint32_t* myoff = sh.hist[sel] + (SHARED_CURSORS ? 0 : w * kBins);
    uint32_t* nhist = sh.hist[sel ^ 1];
    uint32_t* tcnt = sh.u.st.tcnt[w];
    uint32_t* tdel = sh.u.st.tdel[w];
    const uint64_t* tin = sh.u.st.tin[w];
    uint16_t* trk = sh.u.st.trk[w];
    uint16_t* perm = sh.u.st.perm[w];
    const uint64_t lt = lanemask_lt();
    tcnt[lane] = 0;
    compiler_fence();
    for (int v = 0; v < nvec; ++v) {
        const bool ok = valid_of(v);
        const uint32_t d = digit_of(tin[v * kWave + lane]);
        const uint64_t peers = match_digit<kMaxDigitBits>(d, ok);
        const uint32_t rk = __popcll(peers & lt);
        const uint32_t b0 = tcnt[ok ? d : 0u];
        compiler_fence();  // every lane has read the count before the leaders advance it
        if (ok && rk == 0) tcnt[d] = b0 + (uint32_t)__popcll(peers);
        compiler_fence();
        trk[v * kWave + lane] = (uint16_t)((d << 10) | (b0 + rk));
    }
    const uint32_t c = tcnt[lane];
    const uint32_t incl = wave_inclusive_scan(c);
    const uint32_t excl = incl - c;
    const uint32_t total = wave_read(incl, kWave - 1);
    uint32_t g0;
    if (SHARED_CURSORS) {
        g0 = c ? atomicAdd(&myoff[lane], c) : 0u;
    } else {
        g0 = myoff[lane];
        compiler_fence();
        myoff[lane] = g0 + c;
    }
    compiler_fence();
    tcnt[lane] = excl;
    tdel[lane] = g0 - excl;
    compiler_fence();
    for (int v = 0; v < nvec; ++v) {
        if (valid_of(v)) {
            const uint32_t t = trk[v * kWave + lane];
            perm[tcnt[t >> 10] + (t & 1023u)] = (uint16_t)(v * kWave + lane);
        }
    }
    compiler_fence();
    for (uint32_t i0 = 0; i0 < total; i0 += kWave) {
        const uint32_t i = i0 + lane;
        if (i < total) {
            const uint64_t x = tin[perm[i]];
            const uint32_t pos = tdel[digit_of(x)] + i;
            if (pos < m) {
                dst[pos] = x;
                if (has_next) atomicAdd(&nhist[ck.wave_of(pos) * kBins + next_of(x)], 1u);
            } else {
                atomicOr(err, kErrBounds);  // unreachable unless the histogram is inconsistent
            }
        }
    }
    compiler_fence();
}

// One stable LSD pass over src[0..m): each wave walks its contiguous chunk one staged tile
// (kTileVecs vectors) at a time.
template <int NT, typename DigitOf, typename NextOf>
__device__ __forceinline__ void radix_pass(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst,
                                           uint32_t m, const Chunking& ck, DigitOf digit_of, NextOf next_of,
                                           bool has_next, FwdShared<NT>& sh, int sel, uint32_t* err) {
    const int lane = lane_id(), w = wave_id();
    const uint32_t lo = (uint32_t)w * ck.chunk;
    const uint32_t hi = min(lo + ck.chunk, m);
    uint64_t* tin = sh.u.st.tin[w];
    constexpr int kTileVecs = FwdCfg<NT>::kTileVecs, kTileElems = FwdCfg<NT>::kTileElems;
    constexpr int kHalf = (kTileVecs + 1) / 2;
    for (uint32_t base = lo; base < hi; base += kTileElems) {
        const uint32_t cnt = min((uint32_t)kTileElems, hi - base);
        const int nvec = (int)((cnt + kWave - 1) / kWave);
        // stage the tile: half of its loads in flight per lane at a time, then LDS in arrival order
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint64_t e[kHalf];
#pragma unroll
            for (int j = 0; j < kHalf; ++j) {
                const int v = h * kHalf + j;
                const uint32_t i = (uint32_t)(v * kWave + lane);
                e[j] = (v < kTileVecs && i < cnt) ? src[base + i] : 0ull;
            }
#pragma unroll
            for (int j = 0; j < kHalf; ++j) {
                const int v = h * kHalf + j;
                if (v < kTileVecs) tin[v * kWave + lane] = e[j];
            }
        }
        compiler_fence();
        auto valid_of = [=](int v) { return (uint32_t)(v * kWave + lane) < cnt; };
        scatter_staged<NT, false>(nvec, valid_of, digit_of, next_of, dst, m, has_next, ck, sh, sel, err);
    }
}

// Sorts a[0..m) by the plan's key bits ((element >> key_shift) - base).  On entry hist[sel]
// holds the per-wave counts of pass `first_pass`'s digit (chunked by Chunking(m)); passes before
// first_pass have been done by the caller.  Returns the buffer holding the sorted elements.
template <int NT>
__device__ __forceinline__ uint64_t* radix_sort(uint64_t* a, uint64_t* b, uint32_t m, int key_shift, uint64_t base,
                                                const DigitPlan& plan, int first_pass, FwdShared<NT>& sh,
                                                int sel, uint32_t* err) {
    Chunking ck;
    ck.init(m, NT / kWave);
    for (int p = first_pass; p < plan.P; ++p) {
        const bool has_next = p + 1 < plan.P;
        if (has_next) zero_hist<NT>(sh, sel ^ 1);
        scan_hist<NT>(sh, sel);  // its barriers also publish the zeroed next histogram
        const int pn = has_next ? p + 1 : p;
        const KeyDigit dig{key_shift, plan.shift(p), (1u << plan.bits(p)) - 1u, base};
        const KeyDigit nxt{key_shift, plan.shift(pn), (1u << plan.bits(pn)) - 1u, base};
        radix_pass<NT>(a, b, m, ck, dig, nxt, has_next, sh, sel, err);
        __syncthreads();
        uint64_t* t = a; a = b; b = t;
        sel ^= 1;
    }
    return a;
}

// Counts digit_of(element) of src[0..m) per wave chunk into hist[sel] (which must be zero).
template <int NT, typename DigitOf>
__device__ __forceinline__ void count_digits(const uint64_t* __restrict__ src, uint32_t m, DigitOf digit_of,
                                             FwdShared<NT>& sh, int sel) {
    const int lane = lane_id(), w = wave_id();
    Chunking ck;
    ck.init(m, NT / kWave);
    const uint32_t lo = (uint32_t)w * ck.chunk, hi = min(lo + ck.chunk, m);
    uint32_t* myhist = sh.hist[sel] + w * kBins;
    for (uint32_t k = lo + lane; k < hi; k += kWave) atomicAdd(&myhist[digit_of(src[k])], 1u);
}

// Sorts the bucket g[0..cnt), cnt <= kSortCap, by ((element >> kshift) - base), which has at
// most `bits` bits, entirely inside LDS and leaves it in u.lsort[0..cnt) (published by a
// barrier); the caller writes it back.
// LSD passes with 4-bit digits.  The bucket is padded with all-ones elements to a multiple of
// kSortVecs; thread t owns the kSortVecs consecutive positions that start at kSortVecs * t
// (an odd stride, so a wave's 8-byte LDS reads spread over all banks) and keeps them in
// registers during a pass; threads past the padded end idle.  Ranking uses one private
// counter per (digit, thread): an LDS add-with-return per element gives its rank among the
// thread's elements with that digit, an exclusive scan of the counters in (digit, thread)
// order gives where each (digit, thread) run starts.  Counters are 16 bits wide, digits d and
// d+8 share a word and thread t's words sit at row*NT + t, so the lanes of a wave always hit
// different banks.  The padding has the largest digit in every pass and starts out behind all
// real elements, so stability keeps it behind them.
// While in LDS an element is kept shifted left so that its sort key ends at bit 63: when the
// key has at most 32 bits (NARROW) every digit is a bit field of the high dword.
#if BWT_NOINLINE_SORT == 2
#define BWT_SORT_PASSES_INLINE __noinline__
#else
#define BWT_SORT_PASSES_INLINE __forceinline__
#endif
#if BWT_NOINLINE_SORT == 1
#define BWT_SORT_BUCKET_INLINE __noinline__
#else
#define BWT_SORT_BUCKET_INLINE __forceinline__
#endif
template <int NT, bool NARROW>
__device__ BWT_SORT_PASSES_INLINE void lds_sort_passes(uint32_t nthr, int bits, int up, FwdShared<NT>& sh) {
    constexpr int V = FwdCfg<NT>::kSortVecs;
    const int tid = threadIdx.x;
    const bool active = (uint32_t)tid < nthr;
    const int P = (bits + kLdsDigitBits - 1) / kLdsDigitBits;
    const int db = (bits + P - 1) / P;
    // Counter word L = row * NT + thread lives at L + L / 32: a thread's own words (one per row) stay a
    // lane apart, and the 8 consecutive words a thread scans no longer start 32 bytes apart from its
    // neighbours' (an 8-way bank conflict on every scan access), but 33.
    constexpr int RS = NT + NT / 32;  // physical distance of two rows
    uint32_t* cw = sh.v.cw;
    uint32_t* mycw = cw + tid + tid / 32;
    uint32_t* scan8 = cw + 8 * tid + tid / 4;
    uint64_t* ls = sh.u.lsort;
    const uint64_t* mine = ls + (uint32_t)tid * (uint32_t)V;
    auto digit = [&](uint64_t y, int o) {  // o = bit offset of the digit inside the 64-bit word
        return NARROW ? __builtin_amdgcn_ubfe((uint32_t)(y >> 32), (uint32_t)(o - 32), (uint32_t)db)
                      : (uint32_t)(y >> o) & ((1u << db) - 1u);
    };
    for (int p = 0; p < P; ++p) {
        const int o = 64 - bits + p * db;
#pragma unroll
        for (int i = 0; i < 8; ++i) mycw[i * RS] = 0;
        __syncthreads();  // counters cleared, previous pass (or the staging) landed in ls
        uint64_t e[V];
        uint32_t old[V];
        uint32_t cwi[V], half[V];  // counter word of the element's digit and which half of it (0 / 16): kept for the scatter
        if (active) {
#pragma unroll
            for (int j = 0; j < V; ++j) e[j] = mine[j];
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const uint32_t d = digit(e[j], o);
                cwi[j] = (d & 7u) * RS;
                half[j] = (d << 1) & 16u;
                old[j] = atomicAdd(&mycw[cwi[j]], 1u << half[j]);
            }
        }
        __syncthreads();
        {   // exclusive scan of the counters in (digit, thread) order: the words in linear order
            // cover digits 0..7 in their low halves and 8..15 in their high halves
            uint32_t c[8], sum = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                c[i] = scan8[i];
                sum += c[i];
            }
            // exclusive scan over the workgroup; the wave sums of consecutive passes go to alternate
            // buffers, so nobody has to wait until everybody has read them
            constexpr int W = NT / kWave;
            uint32_t* ws = sh.wsum2[p & 1];
            const uint32_t incl = wave_inclusive_scan(sum);
            if (lane_id() == kWave - 1) ws[wave_id()] = incl;
            __syncthreads();
            uint32_t pre = 0, total = 0;
#pragma unroll
            for (int i = 0; i < W; ++i) {
                const uint32_t t = ws[i];
                if (i < wave_id()) pre += t;
                total += t;
            }
            uint32_t run = pre + incl - sum;
            run += (total & 0xFFFFu) << 16;  // every digit >= 8 comes after all digits < 8
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                scan8[i] = run;
                run += c[i];
            }
        }
        __syncthreads();
        if (active) {
            uint32_t off[V];
#pragma unroll
            for (int j = 0; j < V; ++j) off[j] = mycw[cwi[j]];
#pragma unroll
            for (int j = 0; j < V; ++j)  // run start and my rank inside it sit in the same half of their words
                ls[__builtin_amdgcn_ubfe(off[j] + old[j], half[j], 16u)] = e[j];
        }
    }
    __syncthreads();
    (void)up;
}

template <int NT>
__device__ BWT_SORT_BUCKET_INLINE void lds_sort_bucket(const uint64_t* __restrict__ g, uint32_t cnt, int kshift,
                                                uint64_t base, int bits, FwdShared<NT>& sh, int& up,
                                                uint64_t& bsub) {
    constexpr int V = FwdCfg<NT>::kSortVecs;
    const int tid = threadIdx.x;
    uint64_t* ls = sh.u.lsort;
    up = 0;
    bsub = 0;
    if (bits <= 0) {  // all keys equal
        for (uint32_t i = tid; i < cnt; i += NT) ls[i] = g[i];
        __syncthreads();
        return;
    }
    const uint32_t nthr = (cnt + (uint32_t)V - 1u) / (uint32_t)V;  // threads that own elements
    const uint32_t cntp = nthr * (uint32_t)V;
    up = 64 - kshift - bits;  // left shift that puts the key's top bit at bit 63
    bsub = base << kshift;
    for (uint32_t i0 = tid; i0 < cntp; i0 += 8 * NT) {  // eight loads in flight per thread
        uint64_t x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = load_global(g + min(i0 + (uint32_t)(u * NT), cnt - 1u));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t i = i0 + (uint32_t)(u * NT);
            if (i < cntp) ls[i] = i < cnt ? (x[u] - bsub) << up : ~0ull;
        }
    }
    if (bits <= 32) lds_sort_passes<NT, true>(nthr, bits, up, sh);
    else lds_sort_passes<NT, false>(nthr, bits, up, sh);
}

// Writes a bucket sorted in LDS back to g[0..cnt).
template <int NT>
__device__ __forceinline__ void store_bucket(uint64_t* __restrict__ g, uint32_t cnt, int up, uint64_t bsub,
                                             FwdShared<NT>& sh) {
    for (uint32_t i = threadIdx.x; i < cnt; i += NT) g[i] = (sh.u.lsort[i] >> up) + bsub;
    __syncthreads();
}

// Sorts the sub-range a[0..cnt) (too large for LDS) through HBM with b[0..cnt) as the other
// buffer; the result always ends in a.
template <int NT>
__device__ __forceinline__ void sort_range_global(uint64_t* a, uint64_t* b, uint32_t cnt, int kshift, uint64_t base,
                                                  int bits, FwdShared<NT>& sh, uint32_t* err) {
    if (bits <= 0) return;
    DigitPlan plan;
    plan.init(bits);
    zero_hist<NT>(sh, 0);
    __syncthreads();
    const KeyDigit d0{kshift, plan.shift(0), (1u << plan.bits(0)) - 1u, base};
    count_digits<NT>(a, cnt, d0, sh, 0);
    __syncthreads();
    uint64_t* sorted = radix_sort<NT>(a, b, cnt, kshift, base, plan, 0, sh, 0, err);
    if (sorted != a) {
        for (uint32_t i = threadIdx.x; i < cnt; i += NT) a[i] = b[i];
    }
    __syncthreads();
}

// ---- a range in which one key dominates -----------------------------------------------------------
// Periodic text and long runs give round-0 buckets that hold ONE key tens of thousands of times
// plus a handful of others (the suffixes near the block's end).  Radix-sorting such a bucket
// through HBM moves all of it four times over to put those few in place.  Instead: the dominant
// key is guessed from 64 samples; one sweep counts the elements below / above it; if they are
// few (<= kAsideCap) a second sweep lifts them out into LDS together with the dominant-key
// elements that sit where they belong (the first n_lt and the last n_gt slots), and then those
// fill the holes left in the middle while the others, ranked among themselves, take the two ends.
// Elements with equal keys may stand in any order (they are one group for the rerank).  Returns
// false, leaving the range untouched, when no key dominates like that.
constexpr uint32_t kAsideCap = 1024;

template <int NT>
__device__ __noinline__ bool settle_dominant_key(uint64_t* __restrict__ a, uint32_t cnt, int kshift,
                                                    FwdShared<NT>& sh) {
    constexpr int W = NT / kWave;
    static_assert(FwdShared<NT>::kSortCap >= 3 * (int)kAsideCap, "three lists of kAsideCap words in the sort buffer");
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    uint64_t* s_ne = sh.u.lsort;                     // the elements whose key is not the dominant one
    uint64_t* s_mv = sh.u.lsort + kAsideCap;         // dominant-key elements that stand in the two end regions
    uint32_t* s_pos = reinterpret_cast<uint32_t*>(sh.u.lsort + 2 * kAsideCap);  // where s_ne[i] stood
    uint32_t* s_hole = s_pos + kAsideCap;            // slots in the middle region that s_ne elements vacate
    const uint64_t lt = lanemask_lt();
    // the dominant key: the most frequent among 64 samples
    if (w == 0) {
        const uint64_t k = a[(uint32_t)(((uint64_t)lane * cnt) >> 6)] >> kshift;
        uint32_t same = 0;
        for (int i = 0; i < kWave; ++i) same += __shfl(k, i) == k ? 1u : 0u;
        uint32_t best = same;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = max(best, (uint32_t)__shfl_xor(best, o));
        const uint64_t pick = __ballot(same == best);
        const uint64_t kd = __shfl(k, __ffsll((unsigned long long)pick) - 1);
        if (lane == 0) {
            sh.u.lsort[3 * kAsideCap] = kd;
            sh.cntA[0] = sh.cntA[1] = sh.cntA[2] = sh.cntA[3] = 0;
        }
    }
    __syncthreads();
    const uint64_t kd = sh.u.lsort[3 * kAsideCap];
    // how many below, how many above
    {
        uint32_t nl = 0, ng = 0;
        for (uint32_t i = tid; i < cnt; i += NT) {
            const uint64_t k = a[i] >> kshift;
            nl += k < kd ? 1u : 0u;
            ng += k > kd ? 1u : 0u;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            nl += __shfl_xor(nl, o);
            ng += __shfl_xor(ng, o);
        }
        if (lane == 0 && (nl | ng)) {
            atomicAdd(&sh.cntA[0], nl);
            atomicAdd(&sh.cntA[1], ng);
        }
    }
    __syncthreads();
    const uint32_t n_lt = sh.cntA[0], n_gt = sh.cntA[1], n_ne = n_lt + n_gt;
    __syncthreads();
    if (n_ne == 0) return true;
    if (n_ne > kAsideCap) return false;
    if (tid == 0) sh.cntA[0] = sh.cntA[1] = 0;
    __syncthreads();
    // lift the others out, and the dominant-key elements that stand in the end regions
    const uint32_t mid_hi = cnt - n_gt;
    for (uint32_t base = 0; base < cnt; base += NT) {
        const uint32_t q = base + tid;
        const bool valid = q < cnt;
        const uint64_t e = valid ? a[q] : 0ull;
        const bool ne = valid && (e >> kshift) != kd;
        const bool edge = q < n_lt || q >= mid_hi;
        const uint64_t NE = __ballot(ne), MV = __ballot(valid && !ne && edge), HO = __ballot(ne && !edge);
        uint32_t b_ne = 0, b_mv = 0, b_ho = 0;
        if (lane == 0) {
            if (NE) b_ne = atomicAdd(&sh.cntA[0], (uint32_t)__popcll(NE));
            if (MV) b_mv = atomicAdd(&sh.cntA[1], (uint32_t)__popcll(MV));
            if (HO) b_ho = atomicAdd(&sh.cntA[2], (uint32_t)__popcll(HO));
        }
        b_ne = wave_read(b_ne, 0);
        b_mv = wave_read(b_mv, 0);
        b_ho = wave_read(b_ho, 0);
        if (ne) {
            const uint32_t i = b_ne + (uint32_t)__popcll(NE & lt);
            s_ne[i] = e;
            s_pos[i] = q;
            if (!edge) s_hole[b_ho + (uint32_t)__popcll(HO & lt)] = q;
        } else if (valid && edge) {
            s_mv[b_mv + (uint32_t)__popcll(MV & lt)] = e;
        }
    }
    __syncthreads();
    const uint32_t n_mv = sh.cntA[1];  // == number of holes: the end regions hold n_ne slots, n_ne - holes of them taken by the others
    // dominant-key elements into the holes; the others, ranked among themselves, to the two ends
    for (uint32_t i = tid; i < n_mv; i += NT) a[s_hole[i]] = s_mv[i];
    for (uint32_t i = tid; i < n_ne; i += NT) {
        const uint64_t e = s_ne[i], k = e >> kshift;
        uint32_t r = 0;
        for (uint32_t j = 0; j < n_ne; ++j) {
            const uint64_t kj = s_ne[j] >> kshift;
            r += (kj < k || (kj == k && j < i)) ? 1u : 0u;
        }
        a[r < n_lt ? r : cnt - n_ne + r] = e;
    }
    __syncthreads();
    return true;
}

// ---- round 0: rolling packed keys -------------------------------------------------------------
struct Round0 {
    const uint8_t* s;      // block text (len bytes), the delimiter is virtual at position len
    const uint8_t* g_lo;   // readable byte range of the whole batch buffer: [g_lo, g_hi)
    const uint8_t* g_hi;
    uint32_t len, n;
    uint32_t nm;           // suffixes 0 .. nm-1 have m0 text symbols ahead of them ("main"), the rest reach the delimiter ("tails")
    uint32_t dcode;        // what the delimiter position contributes to a tail's key
    uint32_t pad;          // what positions past the delimiter contribute
    uint32_t above;        // 1: the delimiter is larger than every text byte and does not fit a symbol field
    uint32_t delim;        // the delimiter byte itself
    int m0, sbits, ibits;
    int fshift;            // full key >> fshift = fine bin
    uint64_t kmask;        // m0*sbits low bits
};

// Round-0 keys.  Text symbols get dense order-preserving codes 0..sigma-1 (sbits bits each; the
// delimiter and "past the end" get none, which keeps e.g. ACGT at 2 bits per symbol).  K = the
// first m0 symbols packed.  A main suffix has key 2K+1.  A tail -- P (q < m0 text symbols), the
// delimiter, nothing -- sorts after every main suffix whose K is below V = pack(P, dr, 0..0) and
// before all others, where dr = number of text byte values below the delimiter (its code if it
// occurs in the text): that is src/bwt.cpp:66-80's order, the delimiter comparing by byte value
// and the end of the string comparing low.  So a tail gets the even key 2V.  When dr does not fit
// a symbol field (the delimiter is above all 2^sbits text symbols) the tail sorts after every
// main suffix with prefix P instead: key 2(pack(P, 1..1) + 1).  Tails with equal keys (a run of
// the smallest or largest symbol before the delimiter) are ordered by length in the next round.
__device__ __forceinline__ uint64_t round0_key(const Round0& r, uint64_t packed, uint32_t pos) {
    return pos < r.nm ? 2ull * packed + 1ull : 2ull * (packed + r.above);
}

// Tails whose round-0 keys tie are told apart in the next round by this value in place of a
// rank: the shorter suffix first when the delimiter orders low, the longer one when it orders
// above every symbol (see round0_key).  1 .. m0.
__device__ __forceinline__ uint32_t tail_order(const Round0& r, uint32_t idx) {
    return r.above ? idx - r.nm + 1u : r.n - idx;
}

__device__ __forceinline__ uint32_t load_dword_guarded(const uint8_t* p, const uint8_t* g_lo, const uint8_t* g_hi) {
    if (p >= g_lo && p + 4 <= g_hi) return load_global(reinterpret_cast<const uint32_t*>(p));
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (p + j >= g_lo && p + j < g_hi) v |= (uint32_t)load_global(p + j) << (8 * j);
    return v;
}

// Round-0 element: key << (ibits + 8) | preceding byte << ibits | position.  Carrying the byte
// that precedes the suffix through the sort lets every suffix that is already unique after
// round 0 emit its output byte with a coalesced store (no gather, no rank array).
// MODE 0: count the elements per fine bin (top bits of the key) into u.fine.
// MODE 1: generate the elements again and partition them by bucket id digit (`dig`; `nxt` is
//         the digit of a following pass, counted on the way out when has_next) into dst, the
//         waves sharing one cursor per digit in hist[sel].
// Lane l of a wave tile rolls through positions q0 .. q0+7 (q0 = tile + 8 l): it reads the 24
// bytes that start at q0 + m0 - 17, pushes one symbol code per step into a shift register
// and, from step 16 on, the register holds the key of position q0 + (step - 16).
template <int NT, int MODE>
__device__ __noinline__ void round0_sweep(const Round0 r, const BucketDigit dig, const BucketDigit nxt,
                                             bool has_next, uint64_t* __restrict__ dst, FwdShared<NT>& sh, int sel,
                                             uint32_t* err) {
    constexpr int W = NT / kWave;
    const int lane = lane_id(), w = wave_id();
    uint32_t chunk0 = (r.n + (uint32_t)W - 1u) / (uint32_t)W;
    chunk0 = (chunk0 + (uint32_t)kTile - 1u) / (uint32_t)kTile * (uint32_t)kTile;
    const uint32_t lo = (uint32_t)w * chunk0;
    const uint32_t hi = min(lo + chunk0, r.n);
    Chunking ck;
    ck.init(r.n, W);
    const int kshift = r.ibits + 8;
    const int lead = kMaxM0 - r.m0;  // steps whose symbols would be shifted out anyway
    // the global loads of a tile: 7 aligned dwords that cover the 24 window bytes at any alignment
    // and (MODE 1) 3 that cover the 8 bytes preceding the lane's positions
    uint32_t nraw[7], npw[3] = {0u, 0u, 0u};
    auto fetch = [&](uint32_t tile) {
        const uint32_t q0 = tile + (uint32_t)(kRun * lane);
        const uint8_t* wp = r.s + ((int64_t)q0 + r.m0 - kMaxM0);  // text position of window byte 0
        const uint8_t* ap = wp - (reinterpret_cast<uintptr_t>(wp) & 3u);
#pragma unroll
        for (int j = 0; j < 7; ++j) nraw[j] = load_dword_guarded(ap + 4 * j, r.g_lo, r.g_hi);
        if (MODE == 1) {
            const uint8_t* pp = r.s + (int64_t)q0 - 1;
            const uint8_t* pa = pp - (reinterpret_cast<uintptr_t>(pp) & 3u);
#pragma unroll
            for (int j = 0; j < 3; ++j) npw[j] = load_dword_guarded(pa + 4 * j, r.g_lo, r.g_hi);
        }
    };
    if (lo < hi) fetch(lo);
    for (uint32_t tile = lo; tile < hi; tile += kTile) {
        const uint32_t q0 = tile + (uint32_t)(kRun * lane);
        uint32_t raw[7], pw[3];
#pragma unroll
        for (int j = 0; j < 7; ++j) raw[j] = nraw[j];
#pragma unroll
        for (int j = 0; j < 3; ++j) pw[j] = npw[j];
        if (tile + kTile < hi) fetch(tile + kTile);  // the next tile's loads fly while this one is processed
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(r.s + ((int64_t)q0 + r.m0 - kMaxM0)) & 3u);
        uint32_t win[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) win[j] = __builtin_amdgcn_alignbyte(raw[j + 1], raw[j], mis);
        // the 8 bytes that precede positions q0 .. q0+7 (text positions q0-1 .. q0+6)
        uint32_t prevw[2] = {0u, 0u};
        if (MODE == 1) {
            const uint32_t pmis = (uint32_t)(reinterpret_cast<uintptr_t>(r.s + (int64_t)q0 - 1) & 3u);
            prevw[0] = __builtin_amdgcn_alignbyte(pw[1], pw[0], pmis);
            prevw[1] = __builtin_amdgcn_alignbyte(pw[2], pw[1], pmis);
        }
        uint64_t el[kRun];
        uint32_t valid_bits = 0;
        // The steps come in four variants: keys of at most 32 bits (ACGT: 24, four bytes of an
        // alphabet of up to 256 symbols: 32) roll in one register with a shift-or and an and per
        // symbol; and a tile whose windows end before the text does, and that is not the block's
        // first, needs none of the checks for the delimiter, the end and suffix 0: all its
        // positions are main suffixes of the wave's chunk.
        auto steps = [&](auto edge_tag, auto key0) {
            constexpr bool EDGE = decltype(edge_tag)::value;
            using KT = decltype(key0);
            KT key = 0;
            const KT kmask = (KT)r.kmask;
#pragma unroll
            for (int t = 0; t < 16 + kRun; ++t) {
                if (t >= lead) {  // wave-uniform
                    const uint32_t byte = (win[t >> 2] >> (8 * (t & 3))) & 255u;
                    uint32_t c = (uint32_t)sh.sym[byte];
                    if (EDGE) {
                        const uint32_t pos = q0 + (uint32_t)(t - lead);  // = wstart + t, never negative here
                        if (pos >= r.len) c = (pos == r.len) ? r.dcode : r.pad;
                    }
                    key = ((key << r.sbits) | (KT)c) & kmask;
                }
                if (t >= 16) {
                    const int j = t - 16;
                    const uint32_t pe = q0 + (uint32_t)j;
                    const bool valid = !EDGE || pe < hi;
                    if (MODE == 0) {  // main suffixes only (short window); the tails are counted by count_tails
                        if (valid && (!EDGE || pe < r.nm)) atomicAdd(&sh.u.fine[(uint32_t)((2ull * (uint64_t)key + 1ull) >> r.fshift)], 1u);
                    } else {
                        uint32_t prev = (prevw[j >> 2] >> (8 * (j & 3))) & 255u;
                        if (EDGE && pe == 0) prev = r.delim;  // the row of suffix 0 carries the delimiter (src/bwt.cpp:115-116)
                        const uint64_t k0 = EDGE ? round0_key(r, (uint64_t)key, pe) : 2ull * (uint64_t)key + 1ull;
                        el[j] = (k0 << kshift) | ((uint64_t)prev << r.ibits) | pe;
                        if (valid) valid_bits |= 1u << j;
                    }
                }
            }
        };
        const bool edge = tile == 0u || tile + (uint32_t)kTile + 24u > r.len;  // wave-uniform
        if (r.kmask <= 0xFFFFFFFFull) {
            if (edge) steps(std::true_type{}, (uint32_t)0);
            else steps(std::false_type{}, (uint32_t)0);
        } else {
            if (edge) steps(std::true_type{}, (uint64_t)0);
            else steps(std::false_type{}, (uint64_t)0);
        }
        if (MODE == 1) {
            // The partition needs no particular order inside a bucket: every element takes its
            // slot among the tile's elements of the same digit from an LDS counter, the tile is
            // laid out grouped by digit in LDS and leaves in runs of consecutive slots.
            uint32_t* tcnt = sh.u.st.tcnt[w];
            uint32_t* tdel = sh.u.st.tdel[w];
            uint64_t* tin = sh.u.st.tin[w];
            tcnt[lane] = 0;
            compiler_fence();
            uint32_t dj[kRun], rk[kRun];
#pragma unroll
            for (int j = 0; j < kRun; ++j) {
                dj[j] = dig(el[j]);
                rk[j] = 0;
                if ((valid_bits >> j) & 1u) rk[j] = atomicAdd(&tcnt[dj[j]], 1u);
            }
            compiler_fence();
            const uint32_t c = tcnt[lane];
            const uint32_t incl = wave_inclusive_scan(c);
            const uint32_t excl = incl - c;
            const uint32_t total = wave_read(incl, kWave - 1);
            const uint32_t g0 = c ? atomicAdd(&sh.hist[sel][lane], c) : 0u;
            compiler_fence();
            tcnt[lane] = excl;
            tdel[lane] = g0 - excl;
            compiler_fence();
#pragma unroll
            for (int j = 0; j < kRun; ++j)
                if ((valid_bits >> j) & 1u) tin[tcnt[dj[j]] + rk[j]] = el[j];
            compiler_fence();
            uint32_t* nhist = sh.hist[sel ^ 1];
            uint64_t x[kRun];
            uint32_t pos[kRun];
#pragma unroll
            for (int u = 0; u < kRun; ++u) x[u] = tin[u * kWave + lane];  // slots past `total` hold stale data, unused
#pragma unroll
            for (int u = 0; u < kRun; ++u) pos[u] = dig(x[u]);
#pragma unroll
            for (int u = 0; u < kRun; ++u) pos[u] = tdel[pos[u]] + (uint32_t)(u * kWave + lane);
            bool bad = false;
#pragma unroll
            for (int u = 0; u < kRun; ++u) {
                if ((uint32_t)(u * kWave + lane) < total) {
                    if (pos[u] < r.n) store_global(dst + pos[u], x[u]);
                    else bad = true;  // unreachable unless the histogram is inconsistent
                }
            }
            if (bad) atomicOr(err, kErrBounds);
            if (has_next) {
#pragma unroll
                for (int u = 0; u < kRun; ++u)
                    if ((uint32_t)(u * kWave + lane) < total && pos[u] < r.n)
                        atomicAdd(&nhist[ck.wave_of(pos[u]) * kBins + nxt(x[u])], 1u);
            }
            compiler_fence();
        }
    }
}

// Second step of the round-0 partition of a block with more than kBins buckets.  The sweep has
// grouped the elements by the high part of their bucket id; the elements [lo, hi) of one such group
// in src are partitioned here by the low part (`dig`) into dst, the waves sharing one cursor per
// digit in hist[sel] (absolute positions in dst, set by the caller).  The same tile partition as
// in round0_sweep: no order is kept inside a bucket, the elements take their slots from LDS
// counters, the tile is laid out grouped by digit and leaves in runs.
template <int NT>
__device__ __forceinline__ void partition_group(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, uint32_t lo,
                                             uint32_t hi, uint32_t n, const BucketDigit dig, FwdShared<NT>& sh, int sel,
                                             uint32_t* err) {
    constexpr int W = NT / kWave;
    const int lane = lane_id(), w = wave_id();
    uint32_t chunk = (hi - lo + (uint32_t)W - 1u) / (uint32_t)W;
    chunk = (chunk + (uint32_t)kTile - 1u) / (uint32_t)kTile * (uint32_t)kTile;
    const uint32_t wlo = min(lo + (uint32_t)w * chunk, hi), whi = min(wlo + chunk, hi);
    uint32_t* tcnt = sh.u.st.tcnt[w];
    uint32_t* tdel = sh.u.st.tdel[w];
    uint64_t* tin = sh.u.st.tin[w];
    uint64_t nel[kRun];
    auto fetch = [&](uint32_t tile) {
#pragma unroll
        for (int u = 0; u < kRun; ++u) {
            const uint32_t q = tile + (uint32_t)(u * kWave + lane);
            nel[u] = q < whi ? load_global(src + q) : 0ull;
        }
    };
    if (wlo < whi) fetch(wlo);
    for (uint32_t tile = wlo; tile < whi; tile += kTile) {
        uint64_t el[kRun];
        uint32_t valid_bits = 0;
#pragma unroll
        for (int u = 0; u < kRun; ++u) {
            el[u] = nel[u];
            if (tile + (uint32_t)(u * kWave + lane) < whi) valid_bits |= 1u << u;
        }
        if (tile + kTile < whi) fetch(tile + kTile);  // the next tile's loads fly while this one is processed
        tcnt[lane] = 0;
        compiler_fence();
        uint32_t dj[kRun], rk[kRun];
#pragma unroll
        for (int j = 0; j < kRun; ++j) {
            dj[j] = dig(el[j]);
            rk[j] = 0;
            if ((valid_bits >> j) & 1u) rk[j] = atomicAdd(&tcnt[dj[j]], 1u);
        }
        compiler_fence();
        const uint32_t c = tcnt[lane];
        const uint32_t incl = wave_inclusive_scan(c);
        const uint32_t excl = incl - c;
        const uint32_t total = wave_read(incl, kWave - 1);
        const uint32_t g0 = c ? atomicAdd(&sh.hist[sel][lane], c) : 0u;
        compiler_fence();
        tcnt[lane] = excl;
        tdel[lane] = g0 - excl;
        compiler_fence();
#pragma unroll
        for (int j = 0; j < kRun; ++j)
            if ((valid_bits >> j) & 1u) tin[tcnt[dj[j]] + rk[j]] = el[j];
        compiler_fence();
        bool bad = false;
#pragma unroll
        for (int u = 0; u < kRun; ++u) {
            const uint32_t slot = (uint32_t)(u * kWave + lane);
            if (slot < total) {  // slots past `total` hold stale data
                const uint64_t x = tin[slot];
                const uint32_t pos = tdel[dig(x)] + slot;
                if (pos < n) store_global(dst + pos, x);
                else bad = true;  // unreachable unless the directory is inconsistent
            }
        }
        if (bad) atomicOr(err, kErrBounds);
        compiler_fence();
    }
}

// All groups of 2^low_bits buckets, one after the other; start[k] = first slot of bucket k (start[nb] = n).
// Every thread of the workgroup must call.
template <int NT>
__device__ __noinline__ void partition_groups(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst,
                                              const uint32_t* start, uint32_t nb, int low_bits, uint32_t n,
                                              const BucketDigit dig, FwdShared<NT>& sh, uint32_t* err) {
    const int lane = lane_id(), w = wave_id();
    const uint32_t groups = (nb + (1u << low_bits) - 1u) >> low_bits;
#pragma unroll 1
    for (uint32_t gq = 0; gq < groups; ++gq) {
        const uint32_t k0 = gq << low_bits;
        if (w == 0) sh.hist[0][lane] = start[min(k0 + (uint32_t)lane, nb)];
        __syncthreads();
        partition_group<NT>(src, dst, start[k0], start[min(k0 + (1u << low_bits), nb)], n, dig, sh, 0, err);
        __syncthreads();
    }
}

// Where the round-0 buckets of a block start and which fine bins they cover; lives in the
// block's rank array until the ranks are written.  start[k+1] - start[k] = size of bucket k;
// flo[k] .. flo[k+1]-1 = the fine bins a non-empty bucket k may hold.
struct BucketDir {
    uint32_t* start;   // [nb + 1]
    uint32_t* flo;     // [nb + 1]
    uint32_t* fstart;  // [kFine + 1]: sorted row of the first element of every fine bin ([kFine] = n)
    uint32_t nb;       // number of bucket ids (some may be empty)
};

// Groups the fine bins (u.fine counts) into buckets of about `target` elements: a bin goes to
// bucket floor(first slot of the bin / target), so buckets are contiguous key ranges of at most
// target + (largest bin) elements, in key order.  Fills v.lut and the directory.
template <int NT>
__device__ __forceinline__ void assign_buckets(FwdShared<NT>& sh, uint32_t n, uint32_t target, int fbits,
                                               BucketDir& dir) {
    constexpr int PER = kFine / NT;
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const uint32_t nb = (n - 1u) / target + 1u;
    dir.nb = nb;
    uint32_t* tstart = sh.hist[0];            // temporary: first slot per bucket id
    uint32_t* tflo = sh.hist[0] + kMaxBuckets;  // temporary: first fine bin per bucket id
    static_assert(2 * kMaxBuckets <= 2 * (NT / kWave) * kBins, "bucket temporaries fit the two histograms");
    for (int k = tid; k < 2 * kMaxBuckets; k += NT) sh.hist[0][k] = 0xFFFFFFFFu;
    uint32_t sum = 0;
#pragma unroll 4
    for (int j = 0; j < PER; ++j) sum += sh.u.fine[tid * PER + j];
    uint32_t total;
    uint32_t run = block_exclusive_scan<NT>(sum, sh.wsum, total);  // its barriers publish the initialised temporaries
    Chunking dv;  // division by target
    dv.chunk = target;
    dv.magic = 0xFFFFFFFFu / target + 1u;
#pragma unroll 1
    for (int j = 0; j < PER; ++j) {
        const uint32_t f = (uint32_t)(tid * PER + j);
        const uint32_t c = sh.u.fine[f];
        uint32_t id = 0;
        if (c) {
            id = dv.wave_of(run);
            atomicMin(&tstart[id], run);
            atomicMin(&tflo[id], f);
        }
        sh.v.lut[f] = (uint8_t)id;
        dir.fstart[f] = run;
        run += c;
    }
    if (tid == 0) dir.fstart[kFine] = n;
    __syncthreads();
    if (w == 0) {  // suffix minima, so an empty bucket id reads as an empty range
        uint32_t cs = n, cf = 1u << fbits;
        for (int c = (int)(nb / kWave); c >= 0; --c) {
            const uint32_t k = (uint32_t)(c * kWave + lane);
            uint32_t sv = k < nb ? tstart[k] : 0xFFFFFFFFu;
            uint32_t fv = k < nb ? tflo[k] : 0xFFFFFFFFu;
#pragma unroll
            for (int o = 1; o < kWave; o <<= 1) {
                const uint32_t ys = __shfl_down(sv, o), yf = __shfl_down(fv, o);
                if (lane + o < kWave) {
                    sv = min(sv, ys);
                    fv = min(fv, yf);
                }
            }
            sv = min(sv, cs);
            fv = min(fv, cf);
            if (k <= nb) {
                dir.start[k] = sv;
                dir.flo[k] = fv;
            }
            cs = wave_read(sv, 0);
            cf = wave_read(fv, 0);
        }
    }
    __syncthreads();
}

struct RunFlags {
    uint64_t e;
    bool valid, head, tail, ghead;
};

// Walks sorted positions [lo, hi) of cur[0..m) one 64-element vector at a time and hands the
// body the flags of every position: head/tail of its equal-key run, head of its old group.
// One coalesced load per vector, four vectors in flight; the neighbours' keys come from lane
// shuffles (lane 0 / 63 take them from the previous / next vector).
template <bool ROUND0, typename Body>
__device__ __forceinline__ void for_each_run_vector(const uint64_t* __restrict__ cur, uint32_t lo, uint32_t hi,
                                                    uint32_t m, int kshift, int gshift, Body body) {
    constexpr int D = 4;
    const int lane = lane_id();
    if (lo >= hi) return;  // wave-uniform
    uint64_t e[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        e[j] = 0;
        if (lo + j * kWave + lane < m) e[j] = cur[lo + j * kWave + lane];
    }
    uint64_t prev_last = lo > 0 ? cur[lo - 1] : 0ull;  // same address in every lane
    for (uint32_t base = lo; base < hi; base += D * kWave) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const uint32_t vb = base + j * kWave;
            if (vb < hi) {  // wave-uniform
                const uint32_t k = vb + lane;
                const uint64_t ec = e[j];
                const uint64_t nxt0 = wave_read(e[(j + 1) % D], 0);
                const uint64_t ep = wave_from_below(ec, prev_last);
                const uint64_t en = wave_from_above(ec, nxt0);
                prev_last = wave_read(ec, kWave - 1);
                e[j] = 0;
                if (k + D * kWave < m) e[j] = cur[k + D * kWave];
                RunFlags f;
                f.valid = k < hi;
                f.e = ec;
                const uint64_t key = ec >> kshift;
                f.head = f.valid && (k == 0 || (ep >> kshift) != key);
                f.tail = f.valid && (k + 1 == m || (en >> kshift) != key);
                f.ghead = f.valid && (k == 0 || (!ROUND0 && (ep >> gshift) != (ec >> gshift)));
                body(vb, f);
            }
        }
    }
}

// What rerank needs to know about the block besides the sorted elements.
struct BlockIO {
    const uint8_t* s;   // block text
    uint8_t* out;       // block output (n bytes)
    uint32_t n;
    uint32_t delim;
};

// After a sort: find the equal-key runs (groups), give every suffix the position of its
// group's first element as rank, let suffixes that are alone in their group emit their
// output byte and leave the working set, and compact the survivors
// (dense group id << gshift | index) into nxt.  Returns the survivor count in m_next and their
// group count in g_next.
//
// ROUND0: cur holds all n round-0 elements.  Finished suffixes store their byte (carried in the
// element) at their sorted position -- a coalesced store.  When few suffixes survive
// (<= n/16, decided from the counting phase) the rank array is written for the survivors only
// (`full_out` = false): the next round then ranks suffix i+h by binary search in cur
// instead of reading rank[i+h].  `force_full` makes it write all ranks.
// `bail_at`: when at least that many suffixes survive round 0 nothing is written at all (the caller
// puts the block off).
// Later rounds: cur holds the m survivors sorted by (group, rank of the suffix h further on);
// ranks move up by the offset inside the old group; newly finished suffixes fetch their
// preceding byte from the text.
template <int NT, bool ROUND0>
__device__ __forceinline__ void rerank(const uint64_t* __restrict__ cur, uint64_t* __restrict__ nxt,
                                       uint32_t* __restrict__ rank, const BlockIO& io, uint32_t m, int ibits,
                                       int kshift, int gshift, bool force_full, uint32_t bail_at, FwdShared<NT>& sh,
                                       uint32_t& m_next, uint32_t& g_next, bool& full_out) {
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
        for_each_run_vector<ROUND0>(cur, lo, hi, m, kshift, gshift, [&](uint32_t base, const RunFlags& f) {
            const uint64_t H = __ballot(f.head), G = __ballot(f.ghead);
            cA += __popcll(__ballot(f.valid && !(f.head && f.tail)));
            cG += __popcll(__ballot(f.head && !f.tail));
            if (H) lH = base + 63u - (uint32_t)__clzll(H);
            if (G) lG = base + 63u - (uint32_t)__clzll(G);
        });
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
    const bool full = !ROUND0 || force_full || totA > m / 16u;
    if (ROUND0 && totA >= bail_at) {  // workgroup-uniform: the caller gives the block up if this many survive
        m_next = totA;
        g_next = totG;
        full_out = false;
        __syncthreads();
        return;
    }
    // phase C: ranks, output bytes of finished suffixes, compaction of the rest
    for_each_run_vector<ROUND0>(cur, lo, hi, m, kshift, gshift, [&](uint32_t base, const RunFlags& f) {
        const uint64_t H = __ballot(f.head), G = __ballot(f.ghead);
        const uint64_t hm = H & le, gm = G & le;
        const uint32_t khead = hm ? base + 63u - (uint32_t)__clzll(hm) : carryH;
        const uint32_t kg = gm ? base + 63u - (uint32_t)__clzll(gm) : carryG;
        if (H) carryH = base + 63u - (uint32_t)__clzll(H);
        if (G) carryG = base + 63u - (uint32_t)__clzll(G);
        const bool single = f.head && f.tail;
        const bool act = f.valid && !single;
        const uint64_t A = __ballot(act), NS = __ballot(f.head && !f.tail);
        const uint32_t idx = (uint32_t)(f.e & imask);
        if (ROUND0) {
            if (f.valid && (full || act)) rank[idx] = khead;
            if (single) io.out[base + lane] = (uint8_t)(f.e >> ibits);
        } else if (f.valid) {
            const uint32_t delta = khead - kg;
            if (delta || single) {
                const uint32_t rn = rank[idx] + delta;
                if (delta) rank[idx] = rn;
                if (single && rn < io.n) io.out[rn] = idx ? io.s[idx - 1] : (uint8_t)io.delim;
            }
        }
        if (act) {
            const uint32_t pos = baseA + __popcll(A & lt);
            const uint32_t g = baseG + __popcll(NS & le) - 1u;
            nxt[pos] = ((uint64_t)g << gshift) | idx;
        }
        baseA += __popcll(A);
        baseG += __popcll(NS);
    });
    m_next = totA;
    g_next = totG;
    full_out = full;
    __syncthreads();
}

// Round-0 rerank of one bucket while it is still sorted in LDS (u.lsort[0..cnt), global
// sorted positions s0 .. s0+cnt-1; a bucket is a key range, so runs never cross its ends):
// writes the bucket back to g, stores the output byte of every suffix that is alone in its run,
// and for the others writes rank (= global position of the run's first element) and appends
// (dense group id << gshift | index) to nxt (all_ranks: the finished suffixes get their rank -- their
// row -- as well).  m_tot / g_tot are the running survivor and group
// counts of the block and are advanced.  The elements in LDS are still shifted left by `up`
// and lack `bsub` (see lds_sort_bucket).  A first sweep finds the run heads and tails of every
// 64-element vector (kept as ballots in v.cw) and counts; the second one acts on them.
// `narrow` (fshift >= 0): instead of the bucket itself only the low fshift bits of every key are written
// back, as one 32-bit word per sorted row, to nk[s0 ..] -- together with the fine bins' first rows that
// is all a later rank lookup needs (ranks_by_fine_search).  A bucket in which more than a sixteenth
// of the suffixes stay tied then writes the ranks of its finished suffixes too (text: the block is
// going to need all ranks); `skipped` tells the caller whether any bucket did not.
template <int NT>
__device__ __forceinline__ void rerank_bucket(uint64_t* __restrict__ g, uint32_t* __restrict__ nk, int fshift,
                                              uint32_t s0, uint32_t cnt,
                                              uint64_t* __restrict__ nxt, uint32_t* __restrict__ rank,
                                              uint8_t* __restrict__ out, int ibits, int kshift, int gshift,
                                              int up, uint64_t bsub, bool all_ranks, FwdShared<NT>& sh,
                                              uint32_t& m_tot, uint32_t& g_tot, bool& skipped) {
    constexpr int W = NT / kWave;
    const int lane = lane_id(), w = wave_id();
    const uint64_t* ls = sh.u.lsort;
    Chunking ck;
    ck.init(cnt, W);
    const uint32_t lo = (uint32_t)w * ck.chunk;
    const uint32_t hi = min(lo + ck.chunk, cnt);
    const uint64_t lt = lanemask_lt(), le = lanemask_le();
    const uint64_t imask = (1ull << ibits) - 1ull;
    const int ks = kshift + up;
    constexpr int kVecsPerWave = FwdShared<NT>::kSortCap / NT + 5;  // a wave's chunk is at most 20 vectors; reads run up to 3 past it
    uint64_t* flags = reinterpret_cast<uint64_t*>(sh.v.cw) + (size_t)w * 2 * kVecsPerWave;
    static_assert(sizeof(sh.v.cw) >= (size_t)W * 2 * kVecsPerWave * sizeof(uint64_t), "room for two ballots per vector");
    uint32_t cA = 0, cG = 0, lH = 0;
    constexpr int U = 4;  // vectors whose LDS reads are issued together (the loop is latency bound otherwise)
    const uint32_t kmax = cnt - 1u;
    const bool narrow = fshift >= 0;
    const uint32_t fmask = narrow ? (uint32_t)((1ull << fshift) - 1ull) : 0u;
    // The usual bucket (narrow mode, few suffixes stay tied) is done in this one sweep: everything that
    // does not depend on the other waves' counts is written at once, and the few tied suffixes wait
    // in LDS (two words each: position; run head inside the bucket, or kCarry when the run began in an
    // earlier wave's share, and group number inside the wave's share + 1) until the counts are known.
    constexpr uint32_t kFlagWords = (uint32_t)W * 2u * (uint32_t)kVecsPerWave * 2u;
    constexpr uint32_t kStageCap = ((uint32_t)(sizeof(sh.v.cw) / sizeof(uint32_t)) - kFlagWords) / (uint32_t)W / 2u;
    constexpr uint32_t kCarry = 0xFFFFu;
    static_assert(kStageCap >= 64 && FwdShared<NT>::kSortCap < (int)kCarry, "room for some tied suffixes per wave; run heads fit 16 bits");
    uint32_t* const stage = sh.v.cw + kFlagWords + (uint32_t)w * kStageCap * 2u;
    const bool fast = narrow && !all_ranks;
    uint32_t nst = 0;
    bool seen_head = false;  // a run head in my share so far (wave-uniform)
    {
        uint32_t v = 0;
        for (uint32_t base = lo; base < hi; base += U * kWave, v += U) {
            uint64_t key[U], kp[U], kn[U], y[U];
            if (ks >= 32) {  // the key fits the high dword
                // one read per element; the neighbours' keys come from the neighbouring lanes, and only
                // the element before the first and after the last of the U vectors from LDS
                const uint32_t* hi32 = reinterpret_cast<const uint32_t*>(ls) + 1;
                const int sh32 = ks - 32;
                uint32_t k32[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {  // clamped indices: every read is in range, the flags below sort out validity
                    const uint32_t kc = min(base + (uint32_t)(u * kWave + lane), kmax);
                    if (fast) {
                        y[u] = ls[kc];
                        k32[u] = (uint32_t)(y[u] >> 32) >> sh32;
                    } else {
                        k32[u] = hi32[2 * kc] >> sh32;
                    }
                }
                const uint32_t before = hi32[2 * min(base - 1u, kmax)] >> sh32;
                const uint32_t after = hi32[2 * min(base + (uint32_t)(U * kWave), kmax)] >> sh32;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t first = u == 0 ? before : (uint32_t)__builtin_amdgcn_readlane((int)k32[u == 0 ? 0 : u - 1], kWave - 1);
                    const uint32_t last = u == U - 1 ? after : (uint32_t)__builtin_amdgcn_readfirstlane((int)k32[u == U - 1 ? u : u + 1]);
                    key[u] = k32[u];
                    kp[u] = wave_from_below(k32[u], first);
                    kn[u] = wave_from_above(k32[u], last);
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t k = base + (uint32_t)(u * kWave + lane);
                    const uint32_t kc = min(k, kmax), kl = min(k - 1u, kmax), kr = min(k + 1u, kmax);
                    y[u] = ls[kc];
                    key[u] = y[u] >> ks;
                    kp[u] = ls[kl] >> ks;
                    kn[u] = ls[kr] >> ks;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t vb = base + (uint32_t)(u * kWave);
                if (vb < hi) {  // wave-uniform
                    const uint32_t k = vb + lane;
                    const bool valid = k < hi;
                    const bool head = valid && (k == 0 || kp[u] != key[u]);
                    const bool tail = valid && (k + 1 == cnt || kn[u] != key[u]);
                    const uint64_t H = __ballot(head), T = __ballot(tail);
                    if (lane == 0) {
                        flags[2 * (v + u)] = H;
                        flags[2 * (v + u) + 1] = T;
                    }
                    const uint64_t A = __ballot(valid) & ~(H & T), NS = H & ~T;
                    if (fast) {
                        const uint64_t ec = (y[u] >> up) + bsub;
                        if (valid) nk[s0 + k] = (uint32_t)(ec >> kshift) & fmask;
                        if (head && tail) out[s0 + k] = (uint8_t)(ec >> ibits);
                        if (valid && !(head && tail)) {
                            const uint64_t hm = H & le;
                            const uint32_t kh = hm ? vb + 63u - (uint32_t)__clzll(hm) : seen_head ? lH : kCarry;
                            const uint32_t li = nst + (uint32_t)__popcll(A & lt);
                            if (li < kStageCap) {
                                stage[2u * li] = (uint32_t)(ec & imask);
                                stage[2u * li + 1u] = kh | ((cG + (uint32_t)__popcll(NS & le)) << 16);
                            }
                        }
                        nst += (uint32_t)__popcll(A);
                    }
                    cA += __popcll(A);
                    cG += __popcll(NS);
                    if (H) {
                        lH = vb + 63u - (uint32_t)__clzll(H);
                        seen_head = true;
                    }
                }
            }
        }
    }
    if (lane == 0) {
        sh.cntA[w] = cA;
        sh.cntG[w] = cG;
        sh.lastH[w] = lH;
        sh.lastG[w] = nst > kStageCap ? 1u : 0u;
    }
    __syncthreads();
    uint32_t baseA = m_tot, baseG = g_tot, carryH = 0, totA = 0, totG = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const uint32_t ca = sh.cntA[i], cg = sh.cntG[i];
        if (i < w) {
            baseA += ca;
            baseG += cg;
            carryH = max(carryH, sh.lastH[i]);
        }
        totA += ca;
        totG += cg;
    }
    if (narrow && !all_ranks) {
        all_ranks = totA * 16u > cnt;  // workgroup-uniform
        if (!all_ranks) skipped = true;
    }
    bool staged = fast && !all_ranks;  // the tied suffixes of every wave are waiting in LDS
#pragma unroll
    for (int i = 0; i < W; ++i) staged = staged && sh.lastG[i] == 0u;
    if (staged) {
        for (uint32_t i = (uint32_t)lane; i < nst; i += kWave) {
            const uint32_t idx = stage[2u * i], w1 = stage[2u * i + 1u];
            const uint32_t kh = (w1 & 0xFFFFu) == kCarry ? carryH : (w1 & 0xFFFFu);
            rank[idx] = s0 + kh;
            nxt[baseA + i] = ((uint64_t)(baseG + (w1 >> 16) - 1u) << gshift) | idx;
        }
    } else {
        uint32_t v = 0;
        for (uint32_t base0 = lo; base0 < hi; base0 += U * kWave, v += U) {
            uint64_t y[U], Hs[U], Ts[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                y[u] = ls[min(base0 + (uint32_t)(u * kWave + lane), kmax)];
                Hs[u] = flags[2 * (v + u)];
                Ts[u] = flags[2 * (v + u) + 1];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t base = base0 + (uint32_t)(u * kWave);
                if (base < hi) {  // wave-uniform
                    const uint32_t k = base + lane;
                    const bool valid = k < hi;
                    const uint64_t H = Hs[u], T = Ts[u];
                    const uint64_t ec = (y[u] >> up) + bsub;
                    const uint64_t hm = H & le;
                    const uint32_t khead = hm ? base + 63u - (uint32_t)__clzll(hm) : carryH;
                    if (H) carryH = base + 63u - (uint32_t)__clzll(H);
                    const bool head = (H >> lane) & 1u, tail = (T >> lane) & 1u;
                    const bool single = head && tail;
                    const bool act = valid && !single;
                    const uint64_t A = __ballot(valid) & ~(H & T), NS = H & ~T;
                    const uint32_t idx = (uint32_t)(ec & imask);
                    if (valid) {
                        if (narrow) nk[s0 + k] = (uint32_t)(ec >> kshift) & fmask;
                        else g[k] = ec;
                    }
                    if (single) out[s0 + k] = (uint8_t)(ec >> ibits);
                    if (all_ranks && single) rank[idx] = s0 + k;
                    if (act) {
                        rank[idx] = s0 + khead;
                        const uint32_t pos = baseA + __popcll(A & lt);
                        const uint32_t gid = baseG + __popcll(NS & le) - 1u;
                        nxt[pos] = ((uint64_t)gid << gshift) | idx;
                    }
                    baseA += __popcll(A);
                    baseG += __popcll(NS);
                }
            }
        }
    }
    m_tot += totA;
    g_tot += totG;
    __syncthreads();  // cntA/cntG/lastH, the ballots and the LDS bucket are free again
}

// Rank of suffix j by its first m0 symbols = number of round-0 elements with a smaller key:
// binary search in the sorted round-0 array (used while the rank array holds survivors only).
// U suffixes at a time; positions j >= n are tolerated (their result is meaningless).
template <int U>
__device__ __forceinline__ void suffix_keys(const Round0& r, const uint8_t* sym, const uint32_t (&j)[U],
                                            uint64_t (&key)[U]) {
    // the m0 <= 17 bytes at j[u] .. : six aligned dwords per suffix, all loads in flight together
    // (byte after byte, each waiting for the one before, was most of the time of a lookup)
    static_assert(kMaxM0 <= 20, "the window of a key fits five dwords after alignment");
    uint32_t win[U][5];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint8_t* p = r.s + min(j[u], r.len - 1u);  // len >= 1 here
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3u);
        const uint8_t* ap = p - mis;
        uint32_t raw[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) raw[k] = load_dword_guarded(ap + 4 * k, r.g_lo, r.g_hi);
#pragma unroll
        for (int k = 0; k < 5; ++k) win[u][k] = __builtin_amdgcn_alignbyte(raw[k + 1], raw[k], mis);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) key[u] = 0;
#pragma unroll
    for (int t = 0; t < kMaxM0; ++t) {
        if (t < r.m0) {  // workgroup-uniform
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t p = j[u] + (uint32_t)t;
                const uint32_t byte = (win[u][t >> 2] >> (8 * (t & 3))) & 255u;
                uint32_t c = (uint32_t)sym[byte];
                if (p >= r.len) c = (p == r.len) ? r.dcode : r.pad;
                key[u] = (key[u] << r.sbits) | c;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) key[u] = round0_key(r, key[u], j[u]);
}

template <int U>
__device__ __forceinline__ void ranks_by_search(const uint64_t* __restrict__ sorted0, const Round0& r,
                                                const uint8_t* sym, const uint32_t (&j)[U], uint32_t (&out)[U]) {
    uint64_t key[U];
    suffix_keys<U>(r, sym, j, key);
    const int kshift = r.ibits + 8;
    uint32_t lo[U], hi[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        lo[u] = 0;
        hi[u] = r.n;
    }
    for (int it = 0; it <= r.ibits; ++it) {  // the range halves every step: ibits + 1 steps empty it
        uint64_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = sorted0[min((lo[u] + hi[u]) >> 1, r.n - 1u)];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t mid = (lo[u] + hi[u]) >> 1;
            if (lo[u] < hi[u]) {
                if ((v[u] >> kshift) < key[u]) lo[u] = mid + 1;
                else hi[u] = mid;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) out[u] = lo[u];
}

// The same lookup without the sorted round-0 array: the key's top bits name a fine bin, whose rows
// fstart[f] .. fstart[f + 1] are known from the counting sweep, and inside the bin the rows are
// ordered by the low fshift key bits, which the fused rerank has left in nk[row] (one word per row).
// A dozen rows to search instead of the whole block, all in one or two cache lines.
struct FineIndex {
    const uint32_t* fstart;
    const uint32_t* nk;
    int fshift;
};

// out[u] = first row whose key is not below that of suffix j[u]; alone[u] = no other suffix shares the
// key (meaningful for j[u] < n, whose own element is in the array).
template <int U>
__device__ __forceinline__ void ranks_by_fine_search(const FineIndex& fx, const Round0& r, const uint8_t* sym,
                                                     const uint32_t (&j)[U], uint32_t (&out)[U], bool (&alone)[U]) {
    uint64_t key[U];
    suffix_keys<U>(r, sym, j, key);
    const uint32_t fmask = (uint32_t)((1ull << fx.fshift) - 1ull);
    uint32_t lo[U], hi[U], end[U], res[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t f = min((uint32_t)(key[u] >> fx.fshift), (uint32_t)kFine - 1u);
        lo[u] = fx.fstart[f];
        hi[u] = end[u] = fx.fstart[f + 1u];
        res[u] = (uint32_t)key[u] & fmask;
    }
    for (int it = 0; it <= r.ibits; ++it) {
        bool open = false;
        uint32_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            open = open || lo[u] < hi[u];
            v[u] = fx.nk[min((lo[u] + hi[u]) >> 1, r.n - 1u)];
        }
        if (__ballot(open) == 0ull) break;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t mid = (lo[u] + hi[u]) >> 1;
            if (lo[u] < hi[u]) {
                if (v[u] < res[u]) lo[u] = mid + 1;
                else hi[u] = mid;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        out[u] = lo[u];
        alone[u] = lo[u] + 1u >= end[u] || fx.nk[min(lo[u] + 1u, r.n - 1u)] != res[u];
    }
}

// Completes the rank array after a fused round 0 that wrote ranks only for the suffixes still tied
// (and for whole buckets of text-like data): every suffix that is alone with its round-0 key gets
// its row, found through the fine index; the others keep the ranks the rounds so far have given them.
template <int NT>
__device__ __noinline__ void fill_unique_ranks(const FineIndex& fx, const Round0& r, const uint8_t* sym,
                                               uint32_t* __restrict__ rank) {
    constexpr int U = 4;
    for (uint32_t base = 0; base < r.n; base += U * NT) {
        uint32_t j[U], rk[U];
        bool alone[U];
#pragma unroll
        for (int u = 0; u < U; ++u) j[u] = base + (uint32_t)(u * NT) + threadIdx.x;
        ranks_by_fine_search<U>(fx, r, sym, j, rk, alone);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (j[u] < r.n && alone[u]) rank[j[u]] = rk[u];
    }
    __syncthreads();
}

// ---- doubling rounds inside LDS ---------------------------------------------------------------
// The survivors of a round are sorted by group; a round only has to order each group by the
// rank of the suffix h further on.  When the groups are small the survivor list is cut, at
// group boundaries, into chunks that fit LDS, and every chunk is sorted (group, rank) and
// reranked there -- the work of 4-7 radix passes through HBM plus a rerank sweep.

// Cuts cur[0..m) (elements group << gshift | ..., groups ascending) into chunks of whole groups,
// at most `cap` elements each: chunk k = [start[k], start[k+1]).  Returns the number of chunks,
// or 0 when that is not possible (a group longer than the slack, or too many elements).
template <int NT>
__device__ __forceinline__ uint32_t cut_chunks(const uint64_t* __restrict__ cur, uint32_t m, int gshift,
                                               uint32_t cap, uint32_t* __restrict__ start, FwdShared<NT>& sh) {
    if (m <= cap) {  // workgroup-uniform: the whole list is one chunk, however long its groups are
        if (threadIdx.x == 0) {
            start[0] = 0;
            start[1] = m;
        }
        __syncthreads();
        return 1;
    }
    const uint32_t nominal = max(cap - cap / 6u, (m + (uint32_t)kMaxBuckets - 2u) / ((uint32_t)kMaxBuckets - 1u));
    if (nominal > cap) return 0;  // workgroup-uniform
    const uint32_t slack = cap - nominal;
    const uint32_t nchunks = (m + nominal - 1u) / nominal;
    if (threadIdx.x == 0) sh.chunk_bad = 0;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k <= nchunks; k += NT) {
        uint32_t sk = m;
        if (k == 0) sk = 0;
        else if (k < nchunks) {
            const uint32_t p = k * nominal;
            const uint64_t gk = cur[p] >> gshift;
            uint32_t lo = p - min(p, slack), hi = p;  // first element of gk's group lies in [lo, p] ...
            if ((cur[lo] >> gshift) == gk && lo > 0 && (cur[lo - 1] >> gshift) == gk) sh.chunk_bad = 1;  // ... or the group is too long
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if ((cur[mid] >> gshift) < gk) lo = mid + 1;
                else hi = mid;
            }
            sk = lo;
        }
        start[k] = sk;
    }
    __syncthreads();
    return sh.chunk_bad ? 0u : nchunks;
}

// Rerank of one chunk of a doubling round while it is sorted in LDS by (group, rank of the
// suffix h further on): u.lsort[0..cnt), shifted left by `up` with the chunk's base key removed
// (see lds_sort_bucket).  Equal keys stay one group; a suffix's rank moves up by the offset of
// its run inside its old group; suffixes alone in their run are finished and store the byte
// that precedes them at their row; the rest are appended to nxt with dense group ids.
template <int NT>
__device__ __forceinline__ void rerank_chunk(uint32_t cnt, uint64_t* __restrict__ nxt, uint32_t* __restrict__ rank,
                                             const BlockIO& io, int ibits, int gshift, int up, FwdShared<NT>& sh,
                                             uint32_t& m_tot, uint32_t& g_tot) {
    constexpr int W = NT / kWave;
    const int lane = lane_id(), w = wave_id();
    const uint64_t* ls = sh.u.lsort;
    Chunking ck;
    ck.init(cnt, W);
    const uint32_t lo = (uint32_t)w * ck.chunk;
    const uint32_t hi = min(lo + ck.chunk, cnt);
    const uint64_t lt = lanemask_lt(), le = lanemask_le();
    const uint64_t imask = (1ull << ibits) - 1ull;
    const int ks = ibits + up, gs = gshift + up;  // key = element >> ks, old group = element >> gs (64: one group)
    auto group_of = [=](uint64_t y) { return gs < 64 ? y >> gs : 0ull; };
    constexpr int kVecsPerWave = FwdShared<NT>::kSortCap / NT + 5;
    uint64_t* flags = reinterpret_cast<uint64_t*>(sh.v.cw) + (size_t)w * 3 * kVecsPerWave;
    static_assert(sizeof(sh.v.cw) >= (size_t)W * 3 * kVecsPerWave * sizeof(uint64_t), "room for three ballots per vector");
    constexpr int U = NT <= 256 ? 4 : 2;  // vectors whose LDS and rank reads are issued together (a wide workgroup's wave has only 5 vectors)
    const uint32_t kmax = cnt - 1u;
    uint32_t cA = 0, cG = 0, lH = 0, lG = 0;
    {
        uint32_t v = 0;
        for (uint32_t base = lo; base < hi; base += U * kWave, v += U) {
            // one read per element (clamped: the flags sort out validity); the neighbours come from the
            // neighbouring lanes, the elements before the first and after the last vector from LDS
            uint64_t y[U], yp[U], yn[U];
#pragma unroll
            for (int u = 0; u < U; ++u) y[u] = ls[min(base + (uint32_t)(u * kWave + lane), kmax)];
            const uint64_t before = ls[min(base - 1u, kmax)], after = ls[min(base + (uint32_t)(U * kWave), kmax)];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint64_t first = u == 0 ? before : wave_read(y[u == 0 ? 0 : u - 1], kWave - 1);
                const uint64_t last = u == U - 1 ? after : wave_read(y[u == U - 1 ? u : u + 1], 0);
                yp[u] = wave_from_below(y[u], first);
                yn[u] = wave_from_above(y[u], last);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t vb = base + (uint32_t)(u * kWave);
                if (vb < hi) {  // wave-uniform
                    const uint32_t k = vb + lane;
                    const bool valid = k < hi;
                    const bool head = valid && (k == 0 || (yp[u] >> ks) != (y[u] >> ks));
                    const bool tail = valid && (k + 1 == cnt || (yn[u] >> ks) != (y[u] >> ks));
                    const bool ghead = valid && (k == 0 || group_of(yp[u]) != group_of(y[u]));
                    const uint64_t H = __ballot(head), T = __ballot(tail), GH = __ballot(ghead);
                    if (lane == 0) {
                        flags[3 * (v + u)] = H;
                        flags[3 * (v + u) + 1] = T;
                        flags[3 * (v + u) + 2] = GH;
                    }
                    cA += __popcll(__ballot(valid) & ~(H & T));
                    cG += __popcll(H & ~T);
                    if (H) lH = vb + 63u - (uint32_t)__clzll(H);
                    if (GH) lG = vb + 63u - (uint32_t)__clzll(GH);
                }
            }
        }
    }
    if (lane == 0) {
        sh.cntA[w] = cA;
        sh.cntG[w] = cG;
        sh.lastH[w] = lH;
        sh.lastG[w] = lG;
    }
    __syncthreads();
    uint32_t baseA = m_tot, baseG = g_tot, carryH = 0, carryG = 0, totA = 0, totG = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const uint32_t ca = sh.cntA[i], cg = sh.cntG[i];
        if (i < w) {
            baseA += ca;
            baseG += cg;
            carryH = max(carryH, sh.lastH[i]);
            carryG = max(carryG, sh.lastG[i]);
        }
        totA += ca;
        totG += cg;
    }
    {
        uint32_t v = 0;
        for (uint32_t base0 = lo; base0 < hi; base0 += U * kWave, v += U) {
            uint64_t y[U], Hs[U], Ts[U], Gs[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                y[u] = ls[min(base0 + (uint32_t)(u * kWave + lane), kmax)];
                Hs[u] = flags[3 * (v + u)];
                Ts[u] = flags[3 * (v + u) + 1];
                Gs[u] = flags[3 * (v + u) + 2];
            }
            // first the rank reads of the whole batch, then their uses
            uint32_t idx[U], delta[U], rold[U];
            bool single[U], act[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t base = base0 + (uint32_t)(u * kWave);
                const uint32_t k = base + lane;
                const bool valid = k < hi;  // false for the vectors past the chunk as well
                const uint64_t H = Hs[u], G = Gs[u];
                const uint64_t hm = H & le, gm = G & le;
                const uint32_t khead = hm ? base + 63u - (uint32_t)__clzll(hm) : carryH;
                const uint32_t kg = gm ? base + 63u - (uint32_t)__clzll(gm) : carryG;
                if (base < hi) {
                    if (H) carryH = base + 63u - (uint32_t)__clzll(H);
                    if (G) carryG = base + 63u - (uint32_t)__clzll(G);
                }
                const bool head = (H >> lane) & 1u, tail = (Ts[u] >> lane) & 1u;
                single[u] = valid && head && tail;
                act[u] = valid && !single[u];
                idx[u] = (uint32_t)((y[u] >> up) & imask);
                delta[u] = valid ? khead - kg : 0u;
                rold[u] = (delta[u] || single[u]) ? rank[idx[u]] : 0u;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t base = base0 + (uint32_t)(u * kWave);
                if (base < hi) {  // wave-uniform
                    const uint64_t H = Hs[u], T = Ts[u];
                    const uint64_t A = __ballot(act[u] || single[u]) & ~(H & T), NS = H & ~T;
                    if (delta[u] || single[u]) {
                        const uint32_t rn = rold[u] + delta[u];
                        if (delta[u]) rank[idx[u]] = rn;
                        if (single[u] && rn < io.n) io.out[rn] = idx[u] ? io.s[idx[u] - 1u] : (uint8_t)io.delim;
                    }
                    if (act[u]) {
                        const uint32_t pos = baseA + __popcll(A & lt);
                        const uint32_t gid = baseG + __popcll(NS & le) - 1u;
                        nxt[pos] = ((uint64_t)gid << gshift) | idx[u];
                    }
                    baseA += __popcll(A);
                    baseG += __popcll(NS);
                }
            }
        }
    }
    m_tot += totA;
    g_tot += totG;
    __syncthreads();  // the counters, the ballots and the LDS chunk are free again
}

// next_break[x] (x in [0, n]) = the first y >= x, y >= P, with text[y] != text[y - P]; the position n
// (past the block's last symbol, the delimiter) always counts as one.  With it, "the text has period P
// over [lo, hi)" is next_break[lo + P] >= hi, and the place where a periodic stretch ends is one load.
template <int NT>
__device__ __noinline__ void build_next_break(uint32_t* __restrict__ next_break, uint32_t P, const BlockIO& io,
                                              FwdShared<NT>& sh) {
    constexpr int W = NT / kWave;
    const int lane = lane_id(), w = wave_id();
    const uint64_t lt = lanemask_lt();
    const uint32_t n = io.n, len = io.n - 1u;
    auto symbol = [&](uint32_t x) { return x < len ? (int)io.s[x] : (x == len ? (int)io.delim : -1); };
    auto breaks = [&](uint32_t y) { return y >= P && y <= n && (y == n || symbol(y) != symbol(y - P)); };
    Chunking ck;
    ck.init(n + 1u, W);
    const uint32_t lo = (uint32_t)w * ck.chunk, hi = min(lo + ck.chunk, n + 1u);
    constexpr int U = 4;  // vectors whose loads are issued together
    uint32_t firstb = 0xFFFFFFFFu;
#pragma unroll 1
    for (uint32_t base0 = lo; base0 < hi; base0 += U * kWave) {
        uint64_t F[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t y = base0 + (uint32_t)(u * kWave + lane);
            F[u] = __ballot(y < hi && breaks(y));
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (F[u] && firstb == 0xFFFFFFFFu) firstb = base0 + (uint32_t)(u * kWave) + (uint32_t)__ffsll((unsigned long long)F[u]) - 1u;
    }
    if (lane == 0) sh.lastH[w] = firstb;
    __syncthreads();
    uint32_t carry = n;
#pragma unroll
    for (int i = W - 1; i >= 0; --i)
        if (i > w && sh.lastH[i] != 0xFFFFFFFFu) carry = sh.lastH[i];  // ends with the nearest later chunk that has one
    if (lo < hi) {
        const uint32_t span = U * kWave;
#pragma unroll 1
        for (uint32_t base0 = lo + ((hi - lo - 1u) / span) * span;; base0 -= span) {
            uint64_t F[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t y = base0 + (uint32_t)(u * kWave + lane);
                F[u] = __ballot(y < hi && breaks(y));
            }
#pragma unroll
            for (int u = U - 1; u >= 0; --u) {
                const uint32_t base = base0 + (uint32_t)(u * kWave);
                const uint32_t y = base + lane;
                const uint64_t at_or_after = F[u] & ~lt;
                if (y < hi) next_break[y] = at_or_after ? base + (uint32_t)__ffsll((unsigned long long)at_or_after) - 1u : carry;
                if (F[u]) carry = base + (uint32_t)__ffsll((unsigned long long)F[u]) - 1u;
            }
            if (base0 == lo) break;
        }
    }
    __syncthreads();
}

// ---- periodic runs: whole groups resolved in one step ---------------------------------------------
// A group of suffixes tied over their first h symbols whose positions form an arithmetic
// progression i, i+p, i+2p, ... with p <= h lies inside a stretch of the text with period p (each
// member's h symbols overlap the next member's), and two neighbours i+ap, i+(a+1)p agree exactly
// up to the position E where that period breaks -- the SAME E for every pair.  So the comparison
// of text[E-p] with text[E] (the end of the block counting as lowest, src/bwt.cpp:77-79) orders
// all of them at once: ascending position if text[E-p] < text[E], descending otherwise.  Plain
// doubling needs log2(run length / h) more rounds for such a group, each sorting all of it --
// period-7 text in 256 KiB blocks: 15 rounds through HBM in which almost nothing splits.
//
// Works on the survivor list `list` (m elements, group << gshift | index, sorted by group; rank[]
// of a member = row of its group's first suffix): per-group minimum, maximum and count of the
// positions (a table in LDS, at most kTableGroups groups), the groups that are such progressions get their final
// ranks and output bytes, the others are copied to the front of `spare` in order.  Returns the
// number of suffixes left (== m: nothing was resolved and `list` is still the list; else the
// list is now in `spare`).
constexpr uint32_t kMaxPeriod = 256;  // the break of the period is searched by one lane per group
constexpr uint32_t kPeriodicMinBlock = 32768;  // wide workgroups (batches of few blocks): shorter blocks never take the periodic-run step
// Batches of many blocks (256-thread workgroups) take it from 8 KiB: a block of a few repeated bytes that goes through
// all its doubling rounds by radix passes over device memory holds its workgroup 10-15 times as long as other blocks
// (1 GiB of period-7 text in 16 KiB blocks: 361 -> 27 ms, in 8 KiB blocks 204 -> 86 ms).  In a batch of few blocks a
// put-off block is started over by a second launch while the chip idles, and a block of runs with exceptions (a scanned
// page) gains nothing from the step: there the bound stays.
template <int NT>
__device__ __forceinline__ constexpr uint32_t periodic_min_block() {
    return NT > 256 ? kPeriodicMinBlock : 8192u;
}
constexpr uint32_t kFewGroups = 256;     // a round through HBM with at most this many groups: periodic text, most likely
constexpr uint32_t kTableGroups = 2048;  // more groups than this: no table, one thread walks each (short) group
constexpr uint32_t kPeriodGood = 0x80000000u, kPeriodAsc = 0x40000000u, kPeriodMask = 0x00FFFFFFu;

template <int NT>
__device__ __noinline__ uint32_t resolve_periodic_groups(const uint64_t* __restrict__ list, uint64_t* spare,
                                                            uint32_t m, uint32_t G, uint32_t h,
                                                            uint32_t* __restrict__ rank, const BlockIO& io, int ibits,
                                                            int gshift, bool keep_ranks, FwdShared<NT>& sh) {
    constexpr int W = NT / kWave;
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    static_assert(FwdShared<NT>::kSortCap * 2 >= 4 * (int)kTableGroups, "the group table lives in the LDS sort buffer");
    if (G > kTableGroups) return m;  // workgroup-uniform
    uint32_t* tbl = reinterpret_cast<uint32_t*>(sh.u.lsort);  // per group: min, max, count, state
    const uint64_t imask = (1ull << ibits) - 1ull;
    const uint32_t n = io.n, len = io.n - 1u;
    for (uint32_t g = tid; g < G; g += NT) {
        tbl[4u * g + 0u] = 0xFFFFFFFFu;
        tbl[4u * g + 1u] = 0u;
        tbl[4u * g + 2u] = 0u;
        tbl[4u * g + 3u] = 0u;
    }
    if (tid == 0) sh.chunk_bad = 0;
    __syncthreads();
    Chunking ck;
    ck.init(m, W);
    const uint32_t lo = (uint32_t)w * ck.chunk, hi = min(lo + ck.chunk, m);
    // 1. minimum, maximum, count per group (a vector that lies inside one group: one update)
    constexpr int U = 4;  // vectors whose loads are issued together
#pragma unroll 1
    for (uint32_t base0 = lo; base0 < hi; base0 += U * kWave) {
        uint64_t es[U];
#pragma unroll
        for (int u = 0; u < U; ++u) es[u] = list[min(base0 + (uint32_t)(u * kWave + lane), m - 1u)];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t base = base0 + (uint32_t)(u * kWave);
            if (base >= hi) break;  // wave-uniform
            const bool valid = base + lane < hi;
            const uint32_t g = (uint32_t)(es[u] >> gshift), idx = (uint32_t)(es[u] & imask);
            const uint32_t g0 = wave_read(g, 0);
            if (__ballot(valid && g != g0) == 0ull) {
                uint32_t mn = valid ? idx : 0xFFFFFFFFu, mx = valid ? idx : 0u;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    mn = min(mn, (uint32_t)__shfl_xor(mn, o));
                    mx = max(mx, (uint32_t)__shfl_xor(mx, o));
                }
                const uint32_t members = (uint32_t)__popcll(__ballot(valid));
                if (lane == 0) {
                    atomicMin(&tbl[4u * g0 + 0u], mn);
                    atomicMax(&tbl[4u * g0 + 1u], mx);
                    atomicAdd(&tbl[4u * g0 + 2u], members);
                }
            } else if (valid) {
                atomicMin(&tbl[4u * g + 0u], idx);
                atomicMax(&tbl[4u * g + 1u], idx);
                atomicAdd(&tbl[4u * g + 2u], 1u);
            }
        }
    }
    __syncthreads();
    // 2. candidates: count - 1 equal steps between minimum and maximum, all h symbols inside the block
    for (uint32_t g = tid; g < G; g += NT) {
        const uint32_t mn = tbl[4u * g + 0u], mx = tbl[4u * g + 1u], cnt = tbl[4u * g + 2u];
        uint32_t state = 0;
        if (cnt >= 2u && (uint64_t)mx + h <= (uint64_t)len) {
            const uint32_t span = mx - mn, per = span / (cnt - 1u);
            if (per * (cnt - 1u) == span && per >= 1u) state = kPeriodGood | per;
        }
        tbl[4u * g + 3u] = state;
        if (state) sh.chunk_bad = 1;  // (cleared above; counts the settled suffixes from step 4 on)
    }
    __syncthreads();
    const bool any_candidate = sh.chunk_bad != 0u;
    __syncthreads();
    if (!any_candidate) return m;  // the usual outcome for text: nothing here is a progression
    if (tid == 0) sh.chunk_bad = 0;
    __syncthreads();
    // 3. every member on the progression?  (count distinct positions, all congruent, within the span: then it is the progression)
#pragma unroll 1
    for (uint32_t base0 = lo; base0 < hi; base0 += U * kWave) {
        uint64_t es[U];
#pragma unroll
        for (int u = 0; u < U; ++u) es[u] = list[min(base0 + (uint32_t)(u * kWave + lane), m - 1u)];
        uint32_t st[U], mn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t g = (uint32_t)(es[u] >> gshift);
            st[u] = tbl[4u * g + 3u];
            mn[u] = tbl[4u * g + 0u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t k = base0 + (uint32_t)(u * kWave + lane);
            const uint32_t g = (uint32_t)(es[u] >> gshift), idx = (uint32_t)(es[u] & imask);
            if (k < hi && (st[u] & kPeriodGood) && (idx - mn[u]) % (st[u] & kPeriodMask) != 0u) tbl[4u * g + 3u] = 0u;
        }
    }
    __syncthreads();
    // 4. where the period breaks, and which way that orders the group.  A step p <= h is a period of
    // the stretch by itself and its break lies within p symbols (one lane looks for it); a longer
    // step counts only if the text really has that period over the whole stretch, which
    // next_break answers -- for one such period per call, that of the first group that has one.
    const uint32_t short_max = min(h, kMaxPeriod);
    if (tid == 0) sh.vote = 0ull;
    __syncthreads();
    for (uint32_t g = tid; g < G; g += NT) {
        const uint32_t state = tbl[4u * g + 3u];
        if ((state & kPeriodGood) && (state & kPeriodMask) > short_max)
            atomicMax(&sh.vote, ((unsigned long long)tbl[4u * g + 2u] << 32) | (state & kPeriodMask));
    }
    __syncthreads();
    const uint32_t P = (uint32_t)sh.vote;  // the step of the most populous such group
    uint32_t* next_break = reinterpret_cast<uint32_t*>(spare);  // n + 1 words, dead before the compaction writes there
    const bool have_breaks = P != 0u;
    __syncthreads();
    if (have_breaks) build_next_break<NT>(next_break, P, io, sh);
    auto symbol = [&](uint32_t x) { return x < len ? (int)io.s[x] : (x == len ? (int)io.delim : -1); };
    for (uint32_t g = tid; g < G; g += NT) {
        const uint32_t state = tbl[4u * g + 3u];
        if (!(state & kPeriodGood)) continue;
        const uint32_t per = state & kPeriodMask, mn = tbl[4u * g + 0u], from = tbl[4u * g + 1u] + h;
        uint32_t E = 0xFFFFFFFFu;
        if (per <= short_max) {
            for (uint32_t t = 0; t <= per; ++t) {  // the break lies within one period of `from`, or the group would have another member
                const uint32_t x = from + t;
                if (x >= n || symbol(x - per) != symbol(x)) {
                    E = x;
                    break;
                }
            }
        } else if (have_breaks && per == P && next_break[mn + P] >= from) {
            E = next_break[from];
        }
        if (E != 0xFFFFFFFFu) {
            const bool asc = E < n && symbol(E - per) < symbol(E);
            tbl[4u * g + 3u] = state | (asc ? kPeriodAsc : 0u);
            atomicAdd(&sh.chunk_bad, tbl[4u * g + 2u]);
        } else {
            tbl[4u * g + 3u] = 0u;
        }
    }
    __syncthreads();
    const uint32_t resolved = sh.chunk_bad;
    __syncthreads();
    if (resolved == 0) return m;
    // 5. final ranks and output bytes for the resolved groups, the others move on in order.
    // With few groups the resolved ones are written out member by member in position order
    // (member a of a group sits at min + a p and goes to row first + a or first + count - 1 - a:
    // consecutive rows, evenly strided text and rank accesses) rather than in list order.
    const bool by_group = G <= 64u;
    const bool later = resolved < m || keep_ranks;  // suffixes remain: their next keys will look these ranks up (or the caller wants every final rank)
    if (by_group) {
#pragma unroll 1
        for (uint32_t g = 0; g < G; ++g) {
            const uint32_t state = tbl[4u * g + 3u];
            if (!(state & kPeriodGood)) continue;  // workgroup-uniform
            const uint32_t mn = tbl[4u * g + 0u], cnt = tbl[4u * g + 2u], per = state & kPeriodMask;
            const uint32_t first = rank[mn];
            __syncthreads();  // every thread has the group's first row before any member's rank changes
            for (uint32_t a = tid; a < cnt; a += NT) {
                const uint32_t idx = mn + a * per;
                const uint32_t r = first + ((state & kPeriodAsc) ? a : cnt - 1u - a);
                if (later) rank[idx] = r;
                if (r < n) io.out[r] = idx ? io.s[idx - 1u] : (uint8_t)io.delim;
            }
        }
    }
    if (by_group && resolved == m) {  // nothing to carry on
        __syncthreads();
        return 0;
    }
    {
        uint32_t kept = 0;
#pragma unroll 1
        for (uint32_t base = lo; base < hi; base += kWave) {
            const uint32_t k = base + lane;
            const bool keep = k < hi && !(tbl[4u * (uint32_t)(list[k] >> gshift) + 3u] & kPeriodGood);
            kept += (uint32_t)__popcll(__ballot(keep));
        }
        if (lane == 0) sh.cntA[w] = kept;
    }
    __syncthreads();
    uint32_t pos = 0;
#pragma unroll
    for (int i = 0; i < W; ++i)
        if (i < w) pos += sh.cntA[i];
    const uint64_t lt = lanemask_lt();
#pragma unroll 1
    for (uint32_t base = lo; base < hi; base += kWave) {
        const uint32_t k = base + lane;
        const bool valid = k < hi;
        const uint64_t e = list[valid ? k : base];
        const uint32_t g = (uint32_t)(e >> gshift), idx = (uint32_t)(e & imask);
        const uint32_t state = tbl[4u * g + 3u];
        const bool good = valid && (state & kPeriodGood);
        const uint64_t K = __ballot(valid && !good);
        if (good) {
            if (!by_group) {
                const uint32_t a = (idx - tbl[4u * g + 0u]) / (state & kPeriodMask), cnt = tbl[4u * g + 2u];
                const uint32_t r = rank[idx] + ((state & kPeriodAsc) ? a : cnt - 1u - a);
                if (later) rank[idx] = r;
                if (r < n) io.out[r] = idx ? io.s[idx - 1u] : (uint8_t)io.delim;
            }
        } else if (valid) {
            spare[pos + (uint32_t)__popcll(K & lt)] = e;
        }
        pos += (uint32_t)__popcll(K);
    }
    __syncthreads();
    return m - resolved;
}

// The same for a list of MANY SHORT groups (text repeated with a long period: every group is a
// handful of copies of one suffix, one per repetition), without a table: a wave looks at 64
// consecutive list elements at a time, vectors overlapping by kShortGroup, so that every group of
// at most kShortGroup members lies inside the vector that holds its first element among the
// first 64 - kShortGroup lanes.  The first lane of a group gathers minimum, maximum and count by
// shuffles, every member checks itself against the progression, and a period longer than h is
// accepted too if the text really has it over the whole stretch -- which, like the place where
// it breaks, is read from next_break (built once per call for ONE period: the step that most
// suffixes' groups have, found by a first sweep that only votes).  Resolved elements are marked
// in the list (lowest bit of the still empty rank field) and dropped by the compaction.  Returns
// the number of suffixes left (== m: nothing changed, the list is intact).
constexpr uint32_t kShortGroup = 16;
constexpr uint32_t kShortStride = kWave - kShortGroup;

struct ShortGroups {
    uint64_t e;          // my element
    uint32_t idx;        // its position in the text
    bool member;         // I belong to a group this vector is responsible for
    uint32_t head;       // lane of my group's first element
    uint32_t cnt, mn, per;  // the group's size, lowest position and step (per == 0: not a progression), in every member
    uint32_t mx;
};

// Loads the vector at `base` and works out its groups.
// The element of my lane in the vector at `base` and the element just before the vector.
struct ShortLoad {
    uint64_t e, before;
};
__device__ __forceinline__ ShortLoad short_load(const uint64_t* list, uint32_t base, uint32_t m) {
    ShortLoad l;
    l.e = list[min(base + (uint32_t)lane_id(), m - 1u)];
    l.before = list[min(base ? base - 1u : 0u, m - 1u)];  // same address in every lane
    return l;
}

__device__ __forceinline__ ShortGroups short_groups(const ShortLoad& ld, uint32_t base, uint32_t m, uint32_t h,
                                                    uint32_t len, uint64_t imask, int gshift) {
    const int lane = lane_id();
    ShortGroups r;
    const uint32_t k = base + (uint32_t)lane;
    const bool valid = k < m;
    r.e = ld.e;
    const uint64_t before = ld.before;
    r.idx = (uint32_t)(r.e & imask);
    const uint64_t g = r.e >> gshift;
    const uint64_t g_first = wave_read(g, 0);
    const uint64_t gp = wave_from_below(g, base ? before >> gshift : ~g_first);
    const uint64_t H = __ballot(valid && g != gp);  // first elements of groups
    const uint64_t upto = H & lanemask_le();
    r.head = upto ? 63u - (uint32_t)__clzll(upto) : 64u;  // 64: my group began before this vector
    const bool last_vec = base + kWave >= m;
    const uint32_t end = last_vec ? m - base : 64u;  // the list ends here / unknown beyond the vector
    uint64_t later = r.head < 63u ? H & ~((2ull << r.head) - 1ull) : 0ull;
    const uint32_t next = later ? (uint32_t)__ffsll((unsigned long long)later) - 1u : (last_vec ? end : 65u);
    r.member = valid && r.head < kShortStride && next <= 64u;  // (lanes < kShortStride of all vectors tile the list: one vector per group)
    r.cnt = r.member ? next - r.head : 0u;
    if (r.cnt < 2u || r.cnt > kShortGroup) r.member = false;
    // minimum and maximum position of the group: gathered by its first lane, handed to the members
    uint32_t mn = r.idx, mx = r.idx;
#pragma unroll
    for (int t = 1; t < (int)kShortGroup; ++t) {
        const uint32_t v = __shfl_down(r.idx, t);
        if ((uint32_t)t < r.cnt && (uint32_t)lane == r.head) {
            mn = min(mn, v);
            mx = max(mx, v);
        }
    }
    const int src = (int)(r.head & 63u);
    r.mn = __shfl(mn, src);
    r.mx = __shfl(mx, src);
    r.per = 0;
    if (r.member && (uint64_t)r.mx + h <= (uint64_t)len) {
        const uint32_t span = r.mx - r.mn, per = span / (r.cnt - 1u);
        if (per != 0u && per * (r.cnt - 1u) == span) r.per = per;
    }
    // every member on the progression?
    const bool off = r.member && (r.per == 0u || ((r.idx - r.mn) / r.per) * r.per != r.idx - r.mn);
    const uint64_t OFF = __ballot(off);
    if (r.member && ((OFF >> r.head) & ((r.cnt < 64u ? (1ull << r.cnt) : 0ull) - 1ull))) r.per = 0;
    return r;
}

template <int NT>
__device__ __noinline__ uint32_t resolve_short_periodic_groups(uint64_t* __restrict__ list, uint64_t* spare,
                                                               uint32_t m, uint32_t h, uint32_t* __restrict__ rank,
                                                               const BlockIO& io, int ibits, int gshift,
                                                               uint32_t& period_hint, PhaseClock& pc, FwdShared<NT>& sh) {
    constexpr int W = NT / kWave;
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const uint64_t imask = (1ull << ibits) - 1ull, lt = lanemask_lt();
    const uint32_t n = io.n, len = io.n - 1u;
    uint32_t* next_break = reinterpret_cast<uint32_t*>(spare);  // n + 1 words, dead before the compaction writes there
    auto symbol = [&](uint32_t x) { return x < len ? (int)io.s[x] : (x == len ? (int)io.delim : -1); };
    const uint32_t nvec = (m + kShortStride - 1u) / kShortStride;
    const uint32_t short_max = min(h, kMaxPeriod);
    // 1. the long period to prepare for: the one that worked last time, else the step that most
    // suffixes' groups have (a sweep that only votes, in a small LDS hash table: step, members; a
    // step that finds no slot within four probes does not run).  Few votes and no short steps: the
    // list is not (yet) made of such groups, and the call ends there.
    uint32_t P = period_hint;
    if (P == 0u) {
        constexpr uint32_t kVoteSlots = 256;
        uint32_t* vkey = sh.v.cw;
        uint32_t* vcnt = sh.v.cw + kVoteSlots;
        for (uint32_t i = tid; i < 2u * kVoteSlots; i += NT) sh.v.cw[i] = 0;
        if (tid == 0) {
            sh.vote = 0ull;
            sh.chunk_bad = 0;
        }
        __syncthreads();
        uint32_t shorts = 0;
        constexpr uint32_t kSample = 8;  // one vector in kSample votes: the list is either made of such groups or it is not
        ShortLoad ld = short_load(list, (uint32_t)w * kSample * kShortStride, m);
#pragma unroll 1
        for (uint32_t v = (uint32_t)w * kSample; v < nvec; v += W * kSample) {
            const ShortLoad nxt = short_load(list, min(v + W * kSample, nvec - 1u) * kShortStride, m);  // in flight while this one is worked on
            const ShortGroups sg = short_groups(ld, v * kShortStride, m, h, len, imask, gshift);
            ld = nxt;
            if (sg.member && (uint32_t)lane == sg.head && sg.per > short_max) {
                uint32_t slot = (sg.per * 2654435761u) >> 24;
                for (int probe = 0; probe < 4; ++probe, slot = (slot + 1u) & (kVoteSlots - 1u)) {
                    const uint32_t old = atomicCAS(&vkey[slot], 0u, sg.per);
                    if (old == 0u || old == sg.per) {
                        atomicAdd(&vcnt[slot], sg.cnt);
                        break;
                    }
                }
            }
            shorts += (uint32_t)__popcll(__ballot(sg.member && sg.per != 0u && sg.per <= short_max));
        }
        if (lane == 0 && shorts) atomicAdd(&sh.chunk_bad, shorts);
        __syncthreads();
        for (uint32_t i = tid; i < kVoteSlots; i += NT)
            if (vcnt[i]) atomicMax(&sh.vote, ((unsigned long long)vcnt[i] << 32) | vkey[i]);
        __syncthreads();
        const uint32_t votes = (uint32_t)(sh.vote >> 32), nshort = sh.chunk_bad;
        P = (uint32_t)sh.vote;
        __syncthreads();
        if (votes * kSample < m / 8u) P = 0u;
        pc.lap(16);
        if (P == 0u && nshort * kSample < m / 8u) return m;
    }
    // 2. next_break for P
    if (P) build_next_break<NT>(next_break, P, io, sh);
    if (tid == 0) sh.chunk_bad = 0;
    __syncthreads();
    pc.lap(17);
    // 3. settle the groups that are progressions inside a periodic stretch.  Every step is a chain
    // of dependent loads (list -> next_break twice -> two text bytes -> rank), so four vectors go
    // through it side by side.
    uint32_t settled = 0;
    constexpr int U = 4;
#pragma unroll 1
    for (uint32_t v0 = (uint32_t)w * U; v0 < nvec; v0 += W * U) {
        ShortLoad ld[U];
#pragma unroll
        for (int u = 0; u < U; ++u) ld[u] = short_load(list, min(v0 + (uint32_t)u, nvec - 1u) * kShortStride, m);
        ShortGroups sg[U];
        bool lead[U];  // my lane speaks for a candidate group
#pragma unroll
        for (int u = 0; u < U; ++u) {
            sg[u] = short_groups(ld[u], (v0 + (uint32_t)u) * kShortStride, m, h, len, imask, gshift);
            if (v0 + (uint32_t)u >= nvec) sg[u].member = false;
            lead[u] = sg[u].member && (uint32_t)lane == sg[u].head && sg[u].per != 0u;
        }
        uint32_t E[U], nb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            E[u] = 0xFFFFFFFFu;
            nb[u] = 0;
            if (lead[u] && sg[u].per > short_max && sg[u].per == P) nb[u] = next_break[sg[u].mn + P];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!lead[u]) continue;
            if (sg[u].per <= short_max) {
                const uint32_t from = sg[u].mx + h;
                for (uint32_t t = 0; t <= sg[u].per; ++t) {
                    const uint32_t x = from + t;
                    if (x >= n || symbol(x - sg[u].per) != symbol(x)) {
                        E[u] = x;
                        break;
                    }
                }
            } else if (sg[u].per == P && nb[u] >= sg[u].mx + h) {
                E[u] = next_break[sg[u].mx + h];
            }
        }
        int sa[U], sb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            sa[u] = sb[u] = 0;
            if (E[u] != 0xFFFFFFFFu) {
                sa[u] = symbol(E[u] - sg[u].per);
                sb[u] = E[u] < n ? symbol(E[u]) : -1;
            }
        }
        bool done[U];
        uint32_t dir[U], rk[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t d = 0;  // 0 unresolved, 1 descending, 2 ascending
            if (E[u] != 0xFFFFFFFFu) d = (E[u] < n && sa[u] < sb[u]) ? 2u : 1u;
            dir[u] = __shfl(d, (int)(sg[u].head & 63u));
            done[u] = sg[u].member && dir[u] != 0u;
            rk[u] = done[u] ? rank[sg[u].idx] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (done[u]) {
                const uint32_t a = (sg[u].idx - sg[u].mn) / sg[u].per;
                const uint32_t r = rk[u] + (dir[u] == 2u ? a : sg[u].cnt - 1u - a);
                rank[sg[u].idx] = r;
                if (r < n) io.out[r] = sg[u].idx ? io.s[sg[u].idx - 1u] : (uint8_t)io.delim;
                list[(v0 + (uint32_t)u) * kShortStride + (uint32_t)lane] = sg[u].e | (1ull << ibits);
            }
            settled += (uint32_t)__popcll(__ballot(done[u]));
        }
    }
    if (lane == 0 && settled) atomicAdd(&sh.chunk_bad, settled);
    __syncthreads();
    const uint32_t resolved = sh.chunk_bad;
    __syncthreads();
    period_hint = resolved ? P : 0u;  // a period that settled nothing is not tried again unasked
    pc.lap(18);
    if (resolved == 0) return m;
    // 4. the others move on in order
    Chunking ck;
    ck.init(m, W);
    const uint32_t lo = (uint32_t)w * ck.chunk, hi = min(lo + ck.chunk, m);
    {
        uint32_t kept = 0;
#pragma unroll 1
        for (uint32_t base = lo; base < hi; base += kWave) {
            const uint32_t k = base + lane;
            kept += (uint32_t)__popcll(__ballot(k < hi && !((list[k] >> ibits) & 1ull)));
        }
        if (lane == 0) sh.cntA[w] = kept;
    }
    __syncthreads();
    uint32_t pos = 0;
#pragma unroll
    for (int i = 0; i < W; ++i)
        if (i < w) pos += sh.cntA[i];
#pragma unroll 1
    for (uint32_t base = lo; base < hi; base += kWave) {
        const uint32_t k = base + lane;
        const uint64_t e = k < hi ? list[k] : 0ull;
        const bool keep = k < hi && !((e >> ibits) & 1ull);
        const uint64_t K = __ballot(keep);
        if (keep) spare[pos + (uint32_t)__popcll(K & lt)] = e;
        pos += (uint32_t)__popcll(K);
    }
    __syncthreads();
    return m - resolved;
}

// ---- blocks that are periodic as a whole ----------------------------------------------------------
// A block whose text has period P throughout (text[i] == text[i - P] for all i >= P; P the SMALLEST
// such distance, 3 P <= len) does not need all its suffixes sorted.  Let T2 be the block's last 2 P
// symbols plus the delimiter.  T2's suffixes ARE the block's suffixes at positions >= len - 2 P, so
// sorting T2 orders them.  Every other suffix i lies on the chain j, j - P, j - 2 P, ... of exactly
// one suffix j in T2's first period (len - 2 P <= j < len - P).  Two members of a chain agree until
// the later one reaches the delimiter, where the earlier one has the symbol text[len - P]: the whole
// chain is ordered by position, descending if the delimiter is the smaller of the two and ascending
// otherwise (the same for every chain).  And nothing else falls between two members: it would have
// to share their first P symbols, which -- P being the smallest period, so that no two of the P
// rotations of a period are equal -- only suffixes at the same position modulo P do.  So the block's
// suffix array is T2's with every first-period suffix replaced by its chain, and the output is
// T2's sorted order written out run by run: a chain's members all have the same preceding byte
// (except the suffix at position 0, which has the delimiter).
// period-7 text in 256 KiB blocks: 15 suffixes are sorted instead of 262 145.
//
// Two full periods are more than the argument needs.  Let T be the block's last P + tail symbols
// plus the delimiter (tail <= P; T2 is tail = P): its first P suffixes ("heads", one per position
// modulo P, each with more than `tail` symbols of text) stand for their chains, the others for
// themselves.  The chains stay contiguous as long as the P windows of `tail` symbols that start at
// the heads are pairwise different -- a suffix that shares a head's whole text then still has to
// be at the head's position modulo P.  That is checked while T is sorted: when the doubling
// reaches a depth beyond `tail`, no group of still tied suffixes may hold two heads (and no
// periodic-run step may have settled anything before); otherwise the block starts over with T2.
// With a long period T2 has the drawback that EVERY suffix of its first period has a twin in the
// second with which it stays tied for up to P symbols; the short T has only `tail` such pairs.
// Text tiled 6.9 times in a 1 MiB block (period 152 089): 156 K suffixes of plain text are sorted
// instead of 1 M tied seven-fold.
constexpr uint32_t kShortTail = 4096;       // symbols of T beyond one period when the period is long ...
constexpr uint32_t kShortTailMinPeriod = 8192;  // ... i.e. at least this (shorter periods: T2, which is small anyway)

// The smallest whole-block period P with 3 P <= len, or 0.  A period P repeats EVERY window of the
// text P places further on, so the candidates are the places where one chosen 16-byte window occurs
// again, tried in ascending order: the first one that holds for the whole block is the smallest
// period.  The window is the liveliest (most changes between neighbouring bytes) of sixteen near
// the block's start -- a window inside a run of blanks would occur again at every step --, an
// occurrence only counts if three more windows 1 KiB apart occur again with it, and a
// candidate is dropped at its first mismatch, so a wrong one costs little; after kPeriodTries of
// them the search gives up (the block is then sorted like any other).
constexpr uint32_t kPeriodTries = 16;

struct PeriodWords {  // the two LDS words the search needs (FwdShared has them too)
    uint32_t period_cand, period_bad;
};

// max_period: periods above it are not looked for (a caller that only wants the cheap cases -- the
// search for the window's next occurrence then reads that much of the block and no more).
template <int NT, typename Shared>
__device__ __noinline__ uint32_t detect_block_period(const uint8_t* __restrict__ s, uint32_t len, const uint8_t* g_lo,
                                                      const uint8_t* g_hi, Shared& sh, uint32_t max_period = 0xFFFFFFFFu) {
    const int tid = threadIdx.x, lane = lane_id();
    if (len < 16384u) return 0;  // workgroup-uniform (the windows below reach 4 KiB + len / 3 into the block)
    const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(s) & 3u);
    const uint8_t* ab = s - mis;  // dword-aligned; block position i is byte i + mis from here
    if (wave_id() == 0) {  // the window: position 64 t, t < 16, with the most changes between neighbouring bytes
        uint32_t score = 0;
        if (lane < 16)
            for (uint32_t k = 0; k < 15u; ++k) score += s[64u * lane + k] != s[64u * lane + k + 1u] ? 1u : 0u;
        uint32_t best = (score << 8) | (uint32_t)(63 - lane);  // ties: the earliest
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = max(best, (uint32_t)__shfl_xor(best, o));
        if (lane == 0) sh.period_cand = (64u * (63u - (best & 255u))) | ((best >> 8) == 0u ? 0x80000000u : 0u);
    }
    __syncthreads();
    const uint32_t i0 = sh.period_cand & 0x7FFFFFFFu;
    // (all sixteen windows inside runs: the window occurs again at every step of its run, and every such candidate but
    // the first fails where the run ends -- two tries, not sixteen sweeps that each find the next place at once)
    const uint32_t tries = (sh.period_cand >> 31) ? 2u : kPeriodTries;
    __syncthreads();
    const uint8_t* pat = s + i0;
    const uint32_t pat0 = (uint32_t)pat[0] | ((uint32_t)pat[1] << 8) | ((uint32_t)pat[2] << 16) | ((uint32_t)pat[3] << 24);
    const uint32_t limit = min(len / 3u, max_period);  // candidates 1 .. limit (window at i0 + candidate: i0 + len / 3 + 16 <= len)
    const uint32_t pat1 = (uint32_t)pat[4] | ((uint32_t)pat[5] << 8) | ((uint32_t)pat[6] << 16) | ((uint32_t)pat[7] << 24);
    constexpr int U = 4;  // dword pairs whose loads are issued together
    uint32_t from = 0;  // candidates up to here have been tried
#pragma unroll 1
    for (uint32_t attempt = 0; attempt < tries; ++attempt) {
        if (tid == 0) {
            sh.period_cand = 0xFFFFFFFFu;
            sh.period_bad = 0;
        }
        __syncthreads();
        // the first place after i0 + from where the window occurs again
        const uint32_t q_first = (i0 + from + 1u + mis) / 4u, q_end = (i0 + limit + mis) / 4u + 1u;
#pragma unroll 1
        for (uint32_t q0 = q_first + (uint32_t)tid; q0 < q_end; q0 += U * NT) {
            const uint32_t cand = *(volatile uint32_t*)&sh.period_cand;
            if (cand != 0xFFFFFFFFu && cand + i0 + mis < 4u * (q0 - (uint32_t)tid)) break;  // every place from here on lies behind one already found
            uint32_t w0[U], w1[U], w2[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t q = q0 + (uint32_t)(u * NT);
                w0[u] = load_dword_guarded(ab + 4u * (uint64_t)q, g_lo, g_hi);
                w1[u] = load_dword_guarded(ab + 4u * (uint64_t)q + 4u, g_lo, g_hi);
                w2[u] = load_dword_guarded(ab + 4u * (uint64_t)q + 8u, g_lo, g_hi);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t q = q0 + (uint32_t)(u * NT);
#pragma unroll
                for (uint32_t o = 0; o < 4u; ++o) {
                    // the window's first eight bytes (four would match every 256th place of a four-letter text)
                    const uint32_t v = o == 0 ? w0[u] : __builtin_amdgcn_alignbyte(w1[u], w0[u], o);
                    const uint32_t v1 = o == 0 ? w1[u] : __builtin_amdgcn_alignbyte(w2[u], w1[u], o);
                    const uint32_t at = 4u * q + o;  // byte offset from ab
                    if (v == pat0 && v1 == pat1 && at > i0 + from + mis && at - mis - i0 <= limit) {
                        const uint32_t i = at - mis;  // block position of the occurrence
                        // the window, and three more 1 KiB apart (a locally repetitive stretch -- a ruled
                        // line in a text -- repeats the one window many times, but not all four)
                        bool same = true;
                        for (uint32_t wv = 0; wv < 4u && same; ++wv)
                            for (uint32_t k = 0; k < 16u; ++k) same = same && s[i + 1024u * wv + k] == pat[1024u * wv + k];
                        if (same) atomicMin(&sh.period_cand, i - i0);
                    }
                }
            }
        }
        __syncthreads();
        const uint32_t P = sh.period_cand;
        if (P == 0xFFFFFFFFu) return 0;  // workgroup-uniform
        // does the text repeat at distance P everywhere?  aligned dwords of the upper copy against the
        // bytes P below them (two aligned dwords and a funnel shift: the misalignment is the same everywhere)
        const uint32_t m2 = (0u - P) & 3u;
        const uint32_t q_lo = (P + mis) / 4u, q_hi = (len + mis + 3u) / 4u;  // dwords that hold positions P .. len-1
        bool bad = false;
#pragma unroll 1
        for (uint32_t q0 = q_lo + (uint32_t)tid; q0 < q_hi; q0 += U * NT) {
            if (*(volatile uint32_t*)&sh.period_bad) break;  // somebody has found a mismatch
            uint32_t a[U], b0[U], b1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t q = q0 + (uint32_t)(u * NT);
                const int64_t bo = (int64_t)4 * q - (int64_t)P - (int64_t)m2;  // aligned byte offset of the lower dword pair
                a[u] = load_dword_guarded(ab + 4u * (uint64_t)q, g_lo, g_hi);
                b0[u] = load_dword_guarded(ab + bo, g_lo, g_hi);
                b1[u] = load_dword_guarded(ab + bo + 4, g_lo, g_hi);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t q = q0 + (uint32_t)(u * NT);
                if (q < q_hi) {
                    const uint32_t low = m2 == 0 ? b0[u] : (m2 == 1 ? __builtin_amdgcn_alignbyte(b1[u], b0[u], 1)
                                                         : (m2 == 2 ? __builtin_amdgcn_alignbyte(b1[u], b0[u], 2)
                                                                    : __builtin_amdgcn_alignbyte(b1[u], b0[u], 3)));
                    const uint32_t diff = a[u] ^ low;
#pragma unroll
                    for (uint32_t k = 0; k < 4u; ++k) {
                        const uint32_t at = 4u * q + k;  // byte offset from ab of byte k
                        if (at >= P + mis && at < len + mis && ((diff >> (8u * k)) & 255u)) bad = true;
                    }
                }
            }
            if (bad) *(volatile uint32_t*)&sh.period_bad = 1;
        }
        __syncthreads();
        const bool ok = sh.period_bad == 0;
        __syncthreads();
        if (ok) return P;
        from = P;
    }
    return 0u;
}

// Writes the output of a block with whole-block period P from the suffix order of T = its last
// P + tail symbols + delimiter: rank[x] (x = 0 .. P + tail) is the row of T's suffix x among T's
// suffixes.  sa2 / rowstart: scratch of P + tail + 2 words each.  Returns false if the rows do not
// add up (then rank[] was not a permutation: an inconsistent sort).
template <int NT>
__device__ __noinline__ bool expand_periodic_block(const uint8_t* __restrict__ text, uint32_t len, uint32_t P,
                                                    uint32_t tail, uint32_t delim, uint8_t* __restrict__ out,
                                                    const uint32_t* __restrict__ rank, uint32_t* __restrict__ sa2,
                                                    uint32_t* __restrict__ rowstart, FwdShared<NT>& sh) {
    constexpr int W = NT / kWave;
    constexpr int U = 4;  // independent elements per thread whose loads are issued together
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const uint32_t n2 = P + tail + 1u, base = len - P - tail;  // T's suffix x is the block's suffix base + x
    for (uint32_t x0 = tid; x0 < n2; x0 += U * NT) {
        uint32_t r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = rank[min(x0 + (uint32_t)(u * NT), n2 - 1u)];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t x = x0 + (uint32_t)(u * NT);
            if (x < n2 && r[u] < n2) sa2[r[u]] = x;
        }
    }
    __syncthreads();
    // rows per element of T2's order: a first-period suffix stands for its whole chain
    auto weight = [&](uint32_t x) { return x < P ? (base + x) / P + 1u : 1u; };
    Chunking ck;
    ck.init(n2, W);
    const uint32_t lo = (uint32_t)w * ck.chunk, hi = min(lo + ck.chunk, n2);
    {
        uint32_t sum = 0;
        for (uint32_t r = lo + lane; r < hi; r += kWave) sum += weight(sa2[r]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if (lane == 0) sh.cntA[w] = sum;
    }
    __syncthreads();
    uint32_t run = 0, total = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        if (i < w) run += sh.cntA[i];
        total += sh.cntA[i];
    }
    for (uint32_t r0 = lo; r0 < hi; r0 += kWave) {
        const uint32_t r = r0 + lane;
        const uint32_t wt = r < hi ? weight(sa2[r]) : 0u;
        const uint32_t incl = wave_inclusive_scan(wt);
        if (r < hi) rowstart[r] = run + incl - wt;
        run += wave_read(incl, kWave - 1);
    }
    __syncthreads();
    if (total != len + 1u) return false;  // workgroup-uniform
    // the chains run in ascending position when the delimiter is the larger symbol where they part
    const bool asc = delim > (uint32_t)text[len - P];
    if ((len + 1u) / n2 >= 32u) {
        // long runs (short period): a wave per run, 4 bytes per lane and step
        for (uint32_t r = (uint32_t)w; r < n2; r += W) {
            const uint32_t x = sa2[r], j = base + x, row0 = rowstart[r], c = weight(x);
            const uint32_t b = j ? (uint32_t)text[j - 1u] : delim;
            uint8_t* o = out + row0;
            const uint32_t head = min(c, (uint32_t)((4u - (uint32_t)(reinterpret_cast<uintptr_t>(o) & 3u)) & 3u));
            const uint32_t nd = (c - head) / 4u;
            if ((uint32_t)lane < head) o[lane] = (uint8_t)b;
            uint32_t* od = reinterpret_cast<uint32_t*>(o + head);
            for (uint32_t k = lane; k < nd; k += kWave) od[k] = b * 0x01010101u;
            const uint32_t done = head + 4u * nd;
            if (done + (uint32_t)lane < c) o[done + lane] = (uint8_t)b;
        }
    } else {
        for (uint32_t r0 = tid; r0 < n2; r0 += U * NT) {
            uint32_t x[U], row0[U];
            uint8_t b[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t r = min(r0 + (uint32_t)(u * NT), n2 - 1u);
                x[u] = sa2[r];
                row0[u] = rowstart[r];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t j = base + x[u];
                b[u] = j ? text[j - 1u] : (uint8_t)delim;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (r0 + (uint32_t)(u * NT) < n2) {
                    const uint32_t c = weight(x[u]);
                    for (uint32_t k = 0; k < c; ++k) out[row0[u] + k] = b[u];
                }
            }
        }
    }
    __syncthreads();
    // the suffix at position 0 is preceded by the delimiter: it is the far end of the chain of the
    // first-period suffix at the same position modulo P
    if (tid == 0) {
        const uint32_t x0 = (P - base % P) % P;  // (base + x0) % P == 0, x0 < P
        const uint32_t r = rank[x0], c = weight(x0);
        out[asc ? rowstart[r] : rowstart[r] + c - 1u] = (uint8_t)delim;
    }
    __syncthreads();
    return true;
}

// Whether some group of the survivor list (elements group << gshift | index, sorted by group) holds
// two heads, i.e. two suffixes with index below P.  Every head looks back through its group for
// another one; a group too long to look through (more than kHeadLookBack elements before the
// head) counts as a failure, which only costs the caller the short cut.
constexpr uint32_t kHeadLookBack = 32;

template <int NT>
__device__ __noinline__ bool two_heads_in_a_group(const uint64_t* __restrict__ list, uint32_t m, uint32_t P, int ibits,
                                                   int gshift, FwdShared<NT>& sh) {
    const uint64_t imask = (1ull << ibits) - 1ull;
    if (threadIdx.x == 0) sh.period_bad = 0;
    __syncthreads();
    bool bad = false;
    for (uint32_t k = threadIdx.x; k < m; k += NT) {
        const uint64_t e = list[k];
        if ((uint32_t)(e & imask) >= P) continue;
        const uint64_t g = e >> gshift;
        for (uint32_t t = 1; t <= k; ++t) {
            const uint64_t e2 = list[k - t];
            if ((e2 >> gshift) != g) break;
            if ((uint32_t)(e2 & imask) < P || t == kHeadLookBack) {
                bad = true;
                break;
            }
        }
    }
    if (bad) sh.period_bad = 1;
    __syncthreads();
    const bool r = sh.period_bad != 0u;
    __syncthreads();
    return r;
}

// HEAVY = false leaves out the machinery for blocks whose suffixes stay tied in long groups
// (periodic runs, keys that dominate a bucket: cold code that costs the common case registers):
// such a block is put off -- its index appended to a.deferred, nothing of it kept -- and done
// from the start by a second launch of the HEAVY = true instantiation.
template <int NT, bool HEAVY>
__device__ __forceinline__ void forward_block(const uint8_t* __restrict__ s_block, uint32_t len_block, uint32_t delim,
                                              uint8_t* __restrict__ out, uint64_t* bufA, uint64_t* bufB,
                                              uint32_t* rank, FwdShared<NT>& sh, const FwdArgs& a, uint32_t b) {
    auto put_off = [&]() {
        if (threadIdx.x == 0) a.deferred[atomicAdd(&a.queue[2], 1u)] = b;
    };
    constexpr int W = NT / kWave;
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    uint32_t* queue = a.queue;
    if (len_block == 0) {  // empty block: just the delimiter (reference tests/test_small.cpp:83)
        if (tid == 0) out[0] = (uint8_t)delim;
        return;
    }
    PhaseClock pc;
    pc.start(queue);
    // A block that is periodic as a whole: only its last two periods are sorted (T2, see
    // expand_periodic_block); everything below then works on that stretch, writes every rank, and the
    // block's output is expanded from the result at the end.
    const uint8_t* __restrict__ s = s_block;
    uint32_t len = len_block;
    uint32_t block_period = 0, period_tail = 0;
    if (HEAVY && len_block >= periodic_min_block<NT>()) {
        block_period = detect_block_period<NT>(s_block, len_block, a.in, a.in + a.total_len, sh);
        period_tail = block_period >= kShortTailMinPeriod ? kShortTail : block_period;
        pc.lap(19);
    }
restart:
    if (HEAVY && block_period) {
        s = s_block + (len_block - block_period - period_tail);
        len = block_period + period_tail;
    }
    const bool periodic = HEAVY && block_period != 0u;
    // the short T is only good if its heads prove to be told apart within `tail` symbols (checked below)
    bool heads_unchecked = periodic && period_tail < block_period;
    const uint32_t n = len + 1;
    // ---- 0. census and codes ------------------------------------------------
    if (tid < 256) sh.sym[tid] = 0;
    __syncthreads();
    {   // all writers store the same value; 16 bytes per load once the address is aligned
        const uint32_t head = min(len, (uint32_t)((16u - (uint32_t)(reinterpret_cast<uintptr_t>(s) & 15u)) & 15u));
        const uint32_t nvec = (len - head) / 16u;
        const uint4* sv = reinterpret_cast<const uint4*>(s + head);
        for (uint32_t i = tid; i < head; i += NT) sh.sym[s[i]] = 1;
        for (uint32_t i0 = tid; i0 < nvec; i0 += 4 * NT) {
            uint4 x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = sv[min(i0 + (uint32_t)(u * NT), nvec - 1u)];  // a repeated vector marks the same bytes again
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t wds[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
#pragma unroll
                for (int k = 0; k < 16; ++k) sh.sym[(wds[k >> 2] >> (8 * (k & 3))) & 255u] = 1;
            }
        }
        for (uint32_t i = head + 16u * nvec + tid; i < len; i += NT) sh.sym[s[i]] = 1;
    }
    __syncthreads();
    if (HEAVY && periodic && sh.sym[delim]) {
        // the delimiter occurs in the text (a caller's choice the file format never makes): the chain
        // argument needs it to differ from every symbol, so the block is sorted as a whole
        __syncthreads();
        block_period = period_tail = 0;
        s = s_block;
        len = len_block;
        goto restart;
    }
    const uint32_t present = tid < 256 ? sh.sym[tid] : 0;
    uint32_t sigma;  // distinct byte values in the text (the appended delimiter is not one of them)
    const uint32_t below = block_exclusive_scan<NT>(present, sh.wsum, sigma);
    if (tid < 256) sh.sym[tid] = (uint8_t)below;  // code of a text byte; for an absent byte: how many text bytes are below it
    __syncthreads();
    const int sbits = max(1, bit_length(sigma - 1u));
    const int ibits = bit_length(n - 1);
    // about 4/3 ibits bits of symbol information per key: sized as if the codes also had to
    // represent the delimiter and the end (bit_length(sigma + 1) bits per symbol)
    int m0 = (2 * ibits) / bit_length(sigma + 1u);
    const int m0cap = min((64 - 8 - ibits - 2) / sbits, kMaxM0);
    if (m0 > m0cap) m0 = m0cap;
    if (m0 < 1) m0 = 1;
    const uint32_t dr = sh.sym[delim];
    const uint32_t above = dr == (1u << sbits) ? 1u : 0u;
    const int kbits = m0 * sbits + 1 + (int)above;
    const int fbits = min(kbits, kFineBits);
    for (int i = tid; i < kFine; i += NT) sh.u.fine[i] = 0;
    __syncthreads();
    pc.lap(0);

    // ---- 1. round 0: sort all suffixes by their first m0 symbols ------------------------------
    // count per fine bin -> buckets of contiguous key ranges -> partition into the buckets
    // (keys generated on the fly both times) -> every bucket sorted inside LDS.
    Round0 r0;
    r0.s = s;
    r0.g_lo = a.in;
    r0.g_hi = a.in + a.total_len;
    r0.len = len;
    r0.n = n;
    r0.nm = len >= (uint32_t)m0 ? len - (uint32_t)m0 + 1u : 0u;
    r0.above = above;
    r0.dcode = above ? (1u << sbits) - 1u : dr;
    r0.pad = above ? (1u << sbits) - 1u : 0u;
    r0.delim = delim;
    r0.m0 = m0;
    r0.sbits = sbits;
    r0.ibits = ibits;
    r0.fshift = kbits - fbits;
    r0.kmask = (1ull << (m0 * sbits)) - 1ull;
    const int kshift0 = ibits + 8;
    BucketDigit dig{sh.v.lut, kshift0 + r0.fshift, 0, 0u}, dig2 = dig;
    {   // the fine bin of a main suffix only needs the first few symbols of its key: a shorter window
        Round0 rc = r0;
        rc.m0 = min(m0, (fbits - (int)above + sbits - 1) / sbits);
        rc.kmask = (1ull << (rc.m0 * sbits)) - 1ull;
        rc.fshift = rc.m0 * sbits + 1 - (fbits - (int)above);  // applied to 2 K + 1
        round0_sweep<NT, 0>(rc, dig, dig2, false, nullptr, sh, 0, &queue[1]);
        if (w == 0) {  // the tails (at most m0 <= 17): exact keys
            const uint32_t jt[1] = {r0.nm + (uint32_t)lane};
            if (jt[0] < n) {
                uint64_t kt[1];
                suffix_keys<1>(r0, sh.sym, jt, kt);
                atomicAdd(&sh.u.fine[(uint32_t)(kt[0] >> r0.fshift)], 1u);
            }
        }
    }
    __syncthreads();
    pc.lap(1);
    BucketDir dir;
    dir.start = rank + a.ncap;
    const uint32_t target = max(kBucketTarget, (n + (uint32_t)kMaxBuckets - 1u) / (uint32_t)kMaxBuckets);
    dir.flo = dir.start + (kMaxBuckets + 1);
    dir.fstart = dir.flo + (kMaxBuckets + 1);
    assign_buckets<NT>(sh, n, target, fbits, dir);
    bool all_fit = true;  // every bucket fits LDS
    uint32_t longest_bucket = 0;
    for (uint32_t k = 0; k < dir.nb; ++k) {
        const uint32_t cnt = dir.start[k + 1] - dir.start[k];
        all_fit = all_fit && cnt <= (uint32_t)FwdShared<NT>::kSortCap;
        longest_bucket = max(longest_bucket, cnt);
    }
    // a bucket with a sixteenth of the block under one key prefix: a dominant key (text has buckets
    // too long for LDS as well, but not of that size; those are sorted through HBM right here)
    if (!HEAVY && !all_fit && longest_bucket > n / 16u && n >= periodic_min_block<NT>()) {  // workgroup-uniform
        put_off();
        return;
    }
    // Will more than a sixteenth of the suffixes stay tied after this round?  Then the rounds to come read the rank of
    // any suffix, and a bucket that skipped the ranks of its finished suffixes (the fused rerank's choice per bucket,
    // right for uniform keys) leaves them to be looked up one by one afterwards -- 100 of 165 ms per GiB for four
    // symbols with frequencies 0.4 / 0.3 / 0.2 / 0.1.  An estimate from the fine bins: for independent symbols the
    // chance that two suffixes share their fine bin is S = sum of (bin / n)^2, the chance that they share the whole
    // key is S ^ (key bits / fine bits), and a suffix has n - 1 others to share it with.  Where that predicts ties in
    // plenty every bucket writes all its ranks (text, whose repeats no such estimate sees, is left to the buckets'
    // own counts as before; a wrong guess costs time, not correctness).
    bool tie_heavy;
    {
        float part = 0.0f;
        const float inv_n = 1.0f / (float)n;
        for (uint32_t f = (uint32_t)tid; f < (uint32_t)kFine; f += NT) {
            const float share = (float)(dir.fstart[f + 1u] - dir.fstart[f]) * inv_n;
            part += share * share;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if (lane == 0) sh.wsum[w] = __float_as_uint(part);
        __syncthreads();
        float S = 0.0f;
#pragma unroll
        for (int i = 0; i < W; ++i) S += __uint_as_float(sh.wsum[i]);
        __syncthreads();
        const float expo = (float)(m0 * sbits) / (float)max(fbits, 1);
        tie_heavy = (float)n * __powf(S, expo) > 1.0f / 16.0f;
    }
#ifdef BWT_SEPARATE_LONG_BUCKETS  // the earlier way: one bucket too long for LDS, and the whole block's rerank is a sweep of its own
    const bool fused = all_fit;
#else
    const bool fused = true;  // buckets too long for LDS go through the fused rerank piece by piece (rerank_long_bucket)
#endif
#ifdef BWT_OLD_TWO_PASS
    const bool two_pass = dir.nb > (uint32_t)kBins;
    const int b1 = two_pass ? 4 : kMaxDigitBits;
    dig.mask = (1u << b1) - 1u;
    dig2.shift = b1;
    dig2.mask = (1u << kMaxDigitBits) - 1u;
    // shared cursors of the partition pass: where each digit's elements start
    zero_hist<NT>(sh, 1);
    if (w == 0) {
        uint32_t c = 0;
        if (!two_pass) {
            c = (uint32_t)lane < dir.nb ? dir.start[lane + 1] - dir.start[lane] : 0u;
        } else if (lane < 16) {
            for (uint32_t k = (uint32_t)lane; k < dir.nb; k += 16u) c += dir.start[k + 1] - dir.start[k];
        }
        const uint32_t incl = wave_inclusive_scan(c);
        sh.hist[0][lane] = incl - c;
    }
    __syncthreads();
    round0_sweep<NT, 1>(r0, dig, dig2, two_pass, bufA, sh, 0, &queue[1]);
    __syncthreads();
    uint64_t* sorted0 = bufA;
    uint64_t* spare = bufB;
    if (two_pass) {
        Chunking ck;
        ck.init(n, W);
        scan_hist<NT>(sh, 1);
        radix_pass<NT>(bufA, bufB, n, ck, dig2, dig2, false, sh, 1, &queue[1]);
        __syncthreads();
        sorted0 = bufB;
        spare = bufA;
    }
#else
    // More than kBins buckets (blocks above 256 KiB): the sweep groups the elements by the high part
    // of the bucket id (groups of 16 buckets), and every group is then partitioned by the low 4 bits.
    const bool two_pass = dir.nb > (uint32_t)kBins;
    constexpr int kLowBits = 4;
    if (two_pass) {
        dig.shift = kLowBits;
        dig.mask = (1u << kMaxDigitBits) - 1u;
        dig2.mask = (1u << kLowBits) - 1u;
    } else {
        dig.mask = (1u << kMaxDigitBits) - 1u;
    }
    // shared cursors of the partition pass: where each digit's elements start
    zero_hist<NT>(sh, 1);
    if (w == 0) {
        const uint32_t first = two_pass ? (uint32_t)lane << kLowBits : (uint32_t)lane;
        sh.hist[0][lane] = dir.start[min(first, dir.nb)];
    }
    __syncthreads();
    round0_sweep<NT, 1>(r0, dig, dig2, false, bufA, sh, 0, &queue[1]);
    __syncthreads();
    uint64_t* sorted0 = bufA;
    uint64_t* spare = bufB;
    if (two_pass) {
        partition_groups<NT>(bufA, bufB, dir.start, dir.nb, kLowBits, n, dig2, sh, &queue[1]);
        sorted0 = bufB;
        spare = bufA;
    }
#endif
    pc.lap(2);
    // element layout for rounds >= 1: g | r2 | i, where r2 = rank[i+h] + 1, or 0 when the
    // suffix ends exactly at i+h (possible only if the delimiter also occurs inside the
    // block): "past the end" orders before every real rank, as in src/bwt.cpp:77-79
    const int rbits = bit_length(n);
    const int gshift = ibits + rbits;
    // When every bucket fits LDS the rerank is done on the sorted bucket while it is still
    // there (survivor ranks only); otherwise, or when many suffixes survive, a separate sweep does it.
    uint32_t m = 0, G = 0;
    // With the rerank fused, the sorted buckets are not written back: one word per row with the key
    // bits below the fine bin goes into the same buffer (row r at word r: always inside the part of the
    // buffer whose buckets have been consumed), which is what rank lookups need (FineIndex).
    const bool narrow = fused && r0.fshift <= 32;
    uint32_t* const nk = reinterpret_cast<uint32_t*>(sorted0);
    const FineIndex fx{dir.fstart, nk, r0.fshift};
    bool ranks_skipped = false;
    // A bucket too long for the LDS sort (text has such: a pair of frequent letters, a run of blanks) is put in order
    // through device memory and then goes through the same sort + rerank as the others piece by piece: pieces of whole
    // key runs of at most the sort's capacity, cut where one thread's binary searches find the run borders (sorting a
    // sorted piece again costs a few microseconds; what it buys is that the block keeps the fused rerank's narrow
    // write-back and is spared a separate sweep over all its suffixes).  A run of ONE key longer than a piece needs
    // neither order nor neighbours: its suffixes get the row of its first one and stay one group.
    for (uint32_t k = 0; k < dir.nb; ++k) {
        const uint32_t s0 = dir.start[k], cnt = dir.start[k + 1] - s0;
        if (cnt == 0) continue;  // workgroup-uniform
        const uint32_t f0 = dir.flo[k], f1 = dir.flo[k + 1];
        const uint64_t base = (uint64_t)f0 << r0.fshift;
        const int bits = r0.fshift + bit_length(f1 - f0 - 1u);
        constexpr uint32_t cap = (uint32_t)FwdShared<NT>::kSortCap;
        if (cnt > cap) {  // workgroup-uniform
            if (!fused) continue;
            uint64_t* const a = sorted0 + s0;
            if (!(HEAVY && settle_dominant_key<NT>(a, cnt, kshift0, sh))) sort_range_global<NT>(a, spare + s0, cnt, kshift0, base, bits, sh, &queue[1]);
            pc.lap(11);
        }
        uint32_t pos = 0;
        while (pos < cnt) {  // workgroup-uniform; once for a bucket that fits
            uint32_t end = cnt;
            bool equal = false;
            if (cnt - pos > cap) {
                const uint64_t* const a = sorted0 + s0;
                if (tid == 0) {
                    const uint32_t t = pos + cap;  // the first element that does not fit
                    const uint64_t kt = a[t] >> kshift0;
                    uint32_t lo = pos, hi = t;  // where kt's run starts, if that is behind pos
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if ((a[mid] >> kshift0) < kt) lo = mid + 1u;
                        else hi = mid;
                    }
                    uint32_t e2 = lo, eq = 0;
                    if (lo == pos) {  // one key from pos to beyond what fits: where its run ends
                        uint32_t l2 = t, h2 = cnt;
                        while (l2 < h2) {
                            const uint32_t mid = (l2 + h2) >> 1;
                            if ((a[mid] >> kshift0) <= kt) l2 = mid + 1u;
                            else h2 = mid;
                        }
                        e2 = l2;
                        eq = 1;
                    }
                    sh.period_cand = e2;
                    sh.period_bad = eq;
                }
                __syncthreads();
                end = sh.period_cand;
                equal = sh.period_bad != 0u;
                __syncthreads();
            }
            const uint32_t c = end - pos;
            if (!equal) {
                int up;
                uint64_t bsub;
                lds_sort_bucket<NT>(sorted0 + s0 + pos, c, kshift0, base, bits, sh, up, bsub);
                pc.lap(12);
                if (fused) rerank_bucket<NT>(sorted0 + s0 + pos, nk, narrow ? r0.fshift : -1, s0 + pos, c, spare, rank, out, ibits, kshift0, gshift, up, bsub, periodic || tie_heavy, sh, m, G, ranks_skipped);
                else store_bucket<NT>(sorted0 + s0 + pos, c, up, bsub, sh);
                pc.lap(10);
            } else {
                // (the narrow write-back of row r lands in the element of row r / 2: never ahead of the batch being read,
                // but possibly inside it -- so a batch is read by everybody before anybody writes)
                const uint64_t* const a = sorted0 + s0 + pos;
                const uint64_t imask = (1ull << ibits) - 1ull;
                for (uint32_t i0 = 0; i0 < c; i0 += NT) {
                    const uint32_t i = i0 + (uint32_t)tid;
                    const uint64_t e = i < c ? a[i] : 0ull;
                    __syncthreads();
                    if (i < c) {
                        const uint32_t idx = (uint32_t)(e & imask);
                        if (narrow) nk[s0 + pos + i] = (uint32_t)(e >> kshift0) & (uint32_t)((1ull << r0.fshift) - 1ull);
                        rank[idx] = s0 + pos;
                        spare[m + i] = ((uint64_t)G << gshift) | idx;
                    }
                }
                m += c;
                G += 1u;
                __syncthreads();
                pc.lap(10);
            }
            pos = end;
        }
    }
    if (!fused) {
        // The buckets too long for LDS (kept out of the loop above, which is the hot one): through
        // HBM -- unless one key makes up nearly all of the bucket (periodic text, long runs),
        // which takes one read to find out and a few moves to put right
#pragma unroll 1
        for (uint32_t k = 0; k < dir.nb; ++k) {
            const uint32_t s0 = dir.start[k], cnt = dir.start[k + 1] - s0;
            if (cnt <= (uint32_t)FwdShared<NT>::kSortCap) continue;  // workgroup-uniform
            const uint32_t f0 = dir.flo[k], f1 = dir.flo[k + 1];
            if (!(HEAVY && settle_dominant_key<NT>(sorted0 + s0, cnt, kshift0, sh)))
                sort_range_global<NT>(sorted0 + s0, spare + s0, cnt, kshift0, (uint64_t)f0 << r0.fshift,
                                      r0.fshift + bit_length(f1 - f0 - 1u), sh, &queue[1]);
            pc.lap(11);
        }
    }
    BlockIO io;
    io.s = s;
    io.out = out;
    io.n = n;
    io.delim = delim;
    bool full = false;
    __syncthreads();
    if (narrow && m > n / 16u) {
        // many suffixes still tied (text): the rounds to come read rank[] of any suffix.  The fused
        // rerank has written them all when every bucket held many tied suffixes; else the missing ones
        // are looked up.  (Nearly) every suffix with a twin is repeated text, which the common-case
        // kernel leaves to the other one.
        if (!HEAVY && n >= periodic_min_block<NT>() && m >= n - 64u) {  // workgroup-uniform
            put_off();
            return;
        }
        // Buckets with few ties have skipped the ranks of their finished suffixes.  Completing the array is a lookup per
        // suffix of the block; up to a quarter of the block tied (DNA with its repeats: 5-10 %), looking ranks up for
        // the survivors of every round is less -- the list halves or better from round to round.
        const bool by_lookup = ranks_skipped && !periodic && m <= n / 4u;
        if (ranks_skipped && !periodic && !by_lookup) fill_unique_ranks<NT>(fx, r0, sh.sym, rank);
        full = !by_lookup;
    } else if (!fused || m > n / 16u) {
        // (nearly) every suffix with a twin after round 0 is repeated text: the common-case kernel
        // learns that from the count and leaves the block to the other one before the costly sweep
        const uint32_t bail_at = !HEAVY && n >= periodic_min_block<NT>() ? n - 64u : 0xFFFFFFFFu;
        rerank<NT, true>(sorted0, spare, rank, io, n, ibits, kshift0, gshift, periodic, bail_at, sh, m, G, full);
        if (!HEAVY && m >= bail_at) {  // workgroup-uniform
            put_off();
            return;
        }
    } else if (periodic) {
        full = true;  // the fused rerank has written every rank
    }
    pc.lap(3);

    // Tails with equal round-0 keys are not told apart by their ranks yet although their first m0
    // symbols differ (round0_key); the first doubling round needs their true order when it looks
    // up rank[i + m0], so their final rows are worked out here: group start + the tied tails
    // ahead of each.
    if (w == 0) {
        const uint32_t jt[1] = {min(r0.nm + (uint32_t)lane, n - 1u)};
        const bool valid = r0.nm + (uint32_t)lane < n;
        uint64_t kt[1];
        suffix_keys<1>(r0, sh.sym, jt, kt);
        uint32_t ahead = 0, tied = 0;
        const uint32_t nt = n - r0.nm;
        for (uint32_t t = 0; t < nt; ++t) {
            const uint64_t other = __shfl(kt[0], (int)t);
            if (other == kt[0] && t != (uint32_t)lane) {
                ++tied;
                if (r0.above ? t < (uint32_t)lane : t > (uint32_t)lane) ++ahead;
            }
        }
        const bool any = __ballot(valid && tied) != 0ull;  // without ties the ranks are exact already
        if (any) {
            uint32_t rk[1];
            bool alone[1];
            if (narrow) ranks_by_fine_search<1>(fx, r0, sh.sym, jt, rk, alone);
            else ranks_by_search<1>(sorted0, r0, sh.sym, jt, rk);
            if (valid) sh.tailrow[lane] = rk[0] + ahead;
        }
        if (lane == 0) sh.tailrow[kMaxM0] = any ? 1u : 0u;
    }
    __syncthreads();
    const bool tail_fix = sh.tailrow[kMaxM0] != 0u;

    // ---- 2. doubling rounds over the survivors -----------------------------------
    // Buffers: while `full` is false the sorted round-0 array must stay intact (ranks are
    // looked up in it), so the few survivors ping-pong inside the two halves of `spare`.
    uint64_t* cur = full ? sorted0 : spare + a.ncap / 2;  // scratch side of the ping-pong
    uint64_t* alt = spare;                                 // holds the survivor list
    uint32_t h = (uint32_t)m0;
    uint32_t m_before = n;  // the list's length one round earlier
    uint32_t period_hint = 0;  // the long period that last settled groups of this block (0: none yet)
    for (int round = 1; m > 0; ++round) {
        if (round > 64) {  // cannot happen: h >= n is reached after at most 21 doublings
            if (tid == 0) atomicOr(&queue[1], kErrNotConverged);
            break;
        }
        const int gbits = max(1, bit_length(G - 1));
        const bool first = h == (uint32_t)m0;  // ranks still are round-0 ranks
        DigitPlan plan;
        plan.init(rbits + gbits);
        // small groups: the round runs chunk by chunk inside LDS (the bucket directory's slots hold the cuts)
        uint32_t nchunks = cut_chunks<NT>(alt, m, gshift, (uint32_t)FwdShared<NT>::kSortCap, dir.start, sh);
        // Long groups, or a list that has stopped shrinking: the groups that are periodic runs are settled at once
        const bool stalled = round >= 2 && m > n / 8u && (uint64_t)m * 4u > (uint64_t)m_before * 3u;
        m_before = m;
        // (long groups are common in the first rounds of plain text, but then there are thousands of groups)
        // ... and a text in which (nearly) every suffix still has a twin after round 0 is repeated text
        const bool all_twins = round == 1 && n - m <= 64u;
        // (short blocks go through their rounds inside LDS faster than the step's sweeps and barriers take)
        const bool suspect = n >= periodic_min_block<NT>() && full && (stalled || (nchunks == 0 && (G <= kFewGroups || all_twins)));
        if (!HEAVY && suspect) {  // workgroup-uniform
            put_off();
            return;
        }
        if (HEAVY && heads_unchecked && (suspect || 2u * h > period_tail)) {
            // the doubling is about to look beyond the `tail` symbols every head is sure to have (or a
            // periodic-run step is about to settle groups at any depth): no group may hold two heads
            const bool tied_heads = !suspect && two_heads_in_a_group<NT>(alt, m, block_period, ibits, gshift, sh);
            if (suspect || tied_heads) {  // workgroup-uniform
                pc.count(suspect ? 21 : 22, 1);
                period_tail = block_period;  // start over with two full periods
                goto restart;
            }
            heads_unchecked = false;
        }
        if (HEAVY && suspect) {
            const uint32_t left = G <= kTableGroups
                                      ? resolve_periodic_groups<NT>(alt, cur, m, G, h, rank, io, ibits, gshift, periodic, sh)
                                      : resolve_short_periodic_groups<NT>(alt, cur, m, h, rank, io, ibits, gshift, period_hint, pc, sh);
            pc.lap(15);
            if (left != m) {
                uint64_t* t = alt;
                alt = cur;
                cur = t;
                m = left;
                if (m == 0) break;
                nchunks = cut_chunks<NT>(alt, m, gshift, (uint32_t)FwdShared<NT>::kSortCap, dir.start, sh);
            }
        }
        zero_hist<NT>(sh, 0);
        __syncthreads();
        {
            Chunking ck;
            ck.init(m, W);
            const uint32_t lo = (uint32_t)w * ck.chunk, hi = min(lo + ck.chunk, m);
            const uint32_t dmask = (1u << plan.bits(0)) - 1u;
            const uint64_t imask = (1ull << ibits) - 1ull;
            uint32_t* myhist = sh.hist[0] + w * kBins;
            if (full) {
                // four elements per lane at a time: their reads of the list, then their lookups in the rank array, are
                // in flight together (one after the other, every lookup waited out a trip to the L2 on its own)
                constexpr int U = 4;
                for (uint32_t base = lo; base < hi; base += U * kWave) {
                    uint64_t e[U];
                    uint32_t r2[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) e[u] = alt[min(base + (uint32_t)(u * kWave + lane), m - 1u)];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const uint32_t j = (uint32_t)(e[u] & imask) + h;
                        r2[u] = j < n ? rank[j] + 1u : 0u;
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const uint32_t k = base + (uint32_t)(u * kWave + lane);
                        if (k < hi) {
                            const uint32_t idx = (uint32_t)(e[u] & imask);
                            const uint32_t j = idx + h;
                            uint32_t r = r2[u];
                            if (tail_fix && first && j >= r0.nm && j < n) r = sh.tailrow[j - r0.nm] + 1u;
                            if (idx >= r0.nm) r = tail_order(r0, idx);
                            const uint64_t x = e[u] | ((uint64_t)r << ibits);
                            alt[k] = x;
                            atomicAdd(&myhist[(uint32_t)(x >> ibits) & dmask], 1u);
                        }
                    }
                }
            } else {
                // ranks by binary search in the sorted round-0 array: four searches per lane run
                // side by side, so their dependent loads overlap
                constexpr int U = 4;
                for (uint32_t base = lo; base < hi; base += U * kWave) {
                    uint64_t e[U];
                    uint32_t j[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        e[u] = alt[min(base + (uint32_t)(u * kWave + lane), m - 1u)];
                        j[u] = (uint32_t)(e[u] & imask) + h;
                    }
                    uint32_t rk[U];
                    bool alone[U];
                    if (narrow) ranks_by_fine_search<U>(fx, r0, sh.sym, j, rk, alone);
                    else ranks_by_search<U>(sorted0, r0, sh.sym, j, rk);
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const uint32_t k = base + (uint32_t)(u * kWave + lane);
                        if (k < hi) {
                            // (after the first round a suffix that is not alone with its round-0 key has a rank of its
                            // own, as fine as this round needs it, where the lookup only knows its round-0 row)
                            uint32_t r2 = j[u] < n ? ((narrow && !first && !alone[u]) ? rank[j[u]] + 1u : rk[u] + 1u) : 0u;
                            if (tail_fix && j[u] >= r0.nm && j[u] < n) r2 = sh.tailrow[j[u] - r0.nm] + 1u;  // (the exact rows of the suffixes that reach the delimiter within m0 symbols)
                            if (j[u] - h >= r0.nm) r2 = tail_order(r0, j[u] - h);
                            const uint64_t x = e[u] | ((uint64_t)r2 << ibits);
                            alt[k] = x;
                            atomicAdd(&myhist[(uint32_t)(x >> ibits) & dmask], 1u);
                        }
                    }
                }
            }
        }
        __syncthreads();
        pc.lap(4);
        uint64_t* sorted;
        uint64_t* other;
        uint32_t m2 = 0, G2 = 0;
        if (nchunks) {
            for (uint32_t k = 0; k < nchunks; ++k) {
                const uint32_t c0 = dir.start[k], cnt = dir.start[k + 1] - c0;
                if (cnt == 0) continue;  // workgroup-uniform
                const uint64_t gf = alt[c0] >> gshift, gl = alt[c0 + cnt - 1u] >> gshift;
                int up;
                uint64_t bsub;
                lds_sort_bucket<NT>(alt + c0, cnt, ibits, gf << rbits, rbits + bit_length((uint32_t)(gl - gf)), sh, up,
                                    bsub);
                pc.lap(13);
                rerank_chunk<NT>(cnt, cur, rank, io, ibits, gshift, up, sh, m2, G2);
                pc.lap(14);
            }
            sorted = alt;  // its contents are spent; it is the scratch side of the next round
            other = cur;   // the new survivor list
        } else {
            sorted = radix_sort<NT>(alt, cur, m, ibits, 0ull, plan, 0, sh, 0, &queue[1]);
            pc.lap(5);
            other = (sorted == alt) ? cur : alt;
            bool unused;
            rerank<NT, false>(sorted, other, rank, io, m, ibits, ibits, gshift, true, 0xFFFFFFFFu, sh, m2, G2, unused);
            pc.lap(6);
        }
        pc.count(9, 1);
        if (!full) {
            if (m2 == 0) break;
            // survivors remain after the searched round (rare: long repeats in otherwise easy data):
            // the rounds to come read rank[] of any suffix
            if (narrow && !periodic) {
                // A handful of suffixes inside repeats (DNA is full of them): the doubling goes on by lookup.  A suffix
                // that shares its round-0 key with another has had its rank written by every round so far; one that is
                // alone with its key has its row from the lookup itself -- so nobody needs the rank array completed,
                // which is a lookup per suffix of the block (one repeat of 100 symbols in a 256 KiB block of DNA: 21 ms
                // against 2.8 ms for the block).
                cur = sorted;
                alt = other;
                m = m2;
                G = G2;
                h = (h >= 0x40000000u) ? 0x80000000u : h * 2u;
                continue;
            }
            if (narrow) {
                // the tied suffixes have their ranks of this round; the others get their rows looked up,
                // and the doubling goes on with the list at hand in the halves of `spare`
                fill_unique_ranks<NT>(fx, r0, sh.sym, rank);
                full = true;
                // from here on both sides of the ping-pong are whole buffers again (later steps use the
                // scratch side for tables of n + 1 words): the list moves to the start of `spare`
                if (other != spare) {
                    for (uint32_t i = tid; i < m2; i += NT) spare[i] = other[i];
                    __syncthreads();
                    other = spare;
                }
                sorted = sorted0;
                pc.lap(7);
            } else {
                // build the complete rank array from the round-0 order and redo this round with it
                rerank<NT, true>(sorted0, spare, rank, io, n, ibits, kshift0, gshift, true, 0xFFFFFFFFu, sh, m, G, full);
                cur = sorted0;
                alt = spare;
                pc.lap(7);
                continue;
            }
        }
        cur = sorted;
        alt = other;
        m = m2;
        G = G2;
        h = (h >= 0x40000000u) ? 0x80000000u : h * 2u;
    }
    __syncthreads();
    if (HEAVY && periodic) {
        // rank[] now holds the row of every suffix of T; the key buffers are free
        if (!expand_periodic_block<NT>(s_block, len_block, block_period, period_tail, delim, out, rank, reinterpret_cast<uint32_t*>(bufA),
                                       reinterpret_cast<uint32_t*>(bufB), sh) && tid == 0)
            atomicOr(&queue[1], kErrBounds);
        pc.lap(20);
    }
    pc.count(8, 1);
}

template <int NT, bool HEAVY>
__global__ __launch_bounds__(NT, NT <= 256 ? 3 : 4) void bwt_forward_kernel(FwdArgs a, int second) {
    __shared__ FwdShared<NT> sh;
    uint64_t* bufA = reinterpret_cast<uint64_t*>(a.ws + (uint64_t)blockIdx.x * a.ws_stride);
    uint64_t* bufB = bufA + a.ncap;
    uint32_t* rank = reinterpret_cast<uint32_t*>(bufB + a.ncap);
    for (;;) {
        // second == 0: all blocks in order; 1: the blocks that were put off (by the launch before, or by
        // the classifying launch); 2: the blocks the classifying launch has left to this one
        if (threadIdx.x == 0) sh.blk = atomicAdd(&a.queue[second == 1 ? 3 : second == 2 ? 4 : 0], 1u);
        __syncthreads();
        uint32_t b = sh.blk;
        __syncthreads();
        if (b >= (second == 1 ? a.queue[2] : second == 2 ? a.queue[5] : a.nblocks)) break;
        if (second == 1) b = a.deferred[b];
        else if (second == 2) b = a.normal[b];
        const uint64_t off = (uint64_t)b * a.block_size;
        const uint64_t rem = a.total_len - off;
        const uint32_t len = rem < a.block_size ? (uint32_t)rem : a.block_size;
        forward_block<NT, HEAVY>(a.in + off, len, a.delim, a.out + (uint64_t)b * ((uint64_t)a.block_size + 1u), bufA,
                                 bufB, rank, sh, a, b);
        __syncthreads();
    }
}

// ---- classifying launch (long blocks) ---------------------------------------------------------------
// A block that is periodic as a whole is only recognised by the common-case kernel after a full round 0
// (every suffix has a twin), which for long blocks is most of what sorting it would have cost.  For
// batches of long blocks a small launch therefore looks for whole-block periods first -- one sweep over
// a third of each block, a full comparison only for blocks that pass it -- and sorts the blocks into
// the two lists the sorting launches take their work from.
template <int NT>
__global__ __launch_bounds__(NT) void bwt_forward_classify_kernel(FwdArgs a) {
    __shared__ PeriodWords sh;
    __shared__ uint32_t s_blk;
    for (;;) {
        if (threadIdx.x == 0) s_blk = atomicAdd(&a.queue[6], 1u);
        __syncthreads();
        const uint32_t b = s_blk;
        __syncthreads();
        if (b >= a.nblocks) break;
        const uint64_t off = (uint64_t)b * a.block_size;
        const uint64_t rem = a.total_len - off;
        const uint32_t len = rem < a.block_size ? (uint32_t)rem : a.block_size;
        uint32_t P = 0;
        if (len >= kPeriodicMinBlock) P = detect_block_period<NT>(a.in + off, len, a.in, a.in + a.total_len, sh);
        if (threadIdx.x == 0) {
            if (P) a.deferred[atomicAdd(&a.queue[2], 1u)] = b;
            else a.normal[atomicAdd(&a.queue[5], 1u)] = b;
        }
        __syncthreads();
    }
}

}  // namespace bwtgpu
