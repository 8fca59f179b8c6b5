// bwt_forward_lds.hpp -- forward BWT of blocks whose packed text fits LDS.
//
// Replaces, for the blocks it takes, build_suffix_array and bwt_forward of the reference
// (src/bwt.cpp:50-99, :102-123) like bwt_forward.hpp does; the bytes are the same because the suffix
// array of a string is unique.  The general kernel (bwt_forward.hpp) moves an 8-byte element per
// suffix through HBM and sorts 64-bit words in LDS because the block's text stays in device memory:
// everything a later step wants to know about a suffix has to ride along in its element.  Here the
// text itself is resident in LDS -- two bits per symbol for blocks over at most four text symbols
// (256 KiB of DNA is 64 KB), whole bytes for blocks of at most 64 KiB -- so a suffix is just its
// position: any 32-bit window of the text (16 two-bit symbols, or 4 bytes) is two LDS reads and a
// funnel shift away, and the byte that precedes a suffix is one more read.
//
// One workgroup of 1024 lanes per block (the text fills the compute unit's LDS), blocks from a queue:
//   sample     16 bytes per lane spread over the block name the alphabet (2 bits per symbol if at most
//              four values occur, else bytes if the block is short enough, else the block is left to
//              the general kernel); a symbol the sample missed shows up in the pack step
//   pack       text -> sh.text, big-endian inside 32-bit words so that windows compare like strings;
//              behind the text follow "virtual" symbols that stand for the appended delimiter
//   count      every suffix's window -> fine bin (top 13 bits) -> LDS histogram
//   assign     prefix sum of the bins; consecutive bins form buckets of about 256 suffixes (at most
//              1024 buckets: lane t owns bucket t)
//   partition  tiles of 16384 positions: bucket and slot of every suffix (one LDS add-with-return),
//              the tile grouped by bucket in LDS, written to the block's slice of HBM scratch in runs
//   sort       a bucket is one wave's business, no workgroup barrier: its positions come back from
//              HBM, are counted into 512 sub-bins of the bucket's key range, scattered in LDS as
//              (key bits below the bucket's base << 9 | index), finished by a few odd-even
//              transposition passes (sub-bins hold a handful of elements), and lane d of the wave
//              writes the byte that precedes the d-th suffix of the bucket: the output is coalesced.
//              Suffixes whose kept key bits tie are compared window by window through the text.
//
// Order.  The reference compares suffixes of text + delimiter by unsigned byte value, a suffix that
// ends being smaller than any continuation (src/bwt.cpp:66-80); with a delimiter that does not occur
// in the text no comparison ever gets past the delimiter.  Windows see the delimiter as a virtual
// symbol at position len: the code of the smallest text symbol above it, followed by the smallest
// code (or, when the delimiter is above every text symbol, the largest code throughout).  Numeric
// order of windows then is the order of the strings wherever two windows differ, and two EQUAL windows
// of which one contains position len are told apart by position: the suffix nearer the end reaches
// the delimiter first, which makes it the smaller one (the larger one if the delimiter is above).
//
// A block is given up (left to the general kernel through the list it takes its work from) when the
// sample or the pack step meets a fifth symbol or the delimiter inside the text, when a bucket or a
// sub-bin is overfull (skewed keys: text, runs), or when two suffixes agree over kLdsDepth windows
// (long repeats); what it has written by then is overwritten.
#pragma once
#include "bwt_forward.hpp"

namespace bwtgpu {

constexpr int kLdsThreads = 1024;
constexpr int kLdsWaves = kLdsThreads / kWave;
constexpr uint32_t kLdsMaxLen2 = 262144u;        // two bits per symbol: 64 KB of packed text
constexpr uint32_t kLdsMaxLen8 = 65536u;         // bytes
// Shorter blocks than this go to the general kernels.  Measured on 1 GiB of ACGT text, this kernel / the general ones:
// 2 KiB blocks 44 / 60 ms, 4 KiB 22 / 38, 8 KiB 11.1 / 26.7, 16 KiB 7.8 / 20.3 (random bytes alike) -- a workgroup per
// block wins at every size; below 2 KiB the one-wave-per-block kernel for short blocks is near.
constexpr uint32_t kLdsMinLen = 2048u;
constexpr uint32_t kLdsTextWords = 16384u + 8u;  // the text, the virtual symbols behind it, and what a window there reads
constexpr int kLdsFineBits = 13;
constexpr uint32_t kLdsFine = 1u << kLdsFineBits;
constexpr int kLdsFineShift = 32 - kLdsFineBits;
constexpr uint32_t kLdsBuckets = kLdsThreads;    // lane t owns bucket t in the assign and partition steps
constexpr uint32_t kLdsTarget = 288;             // suffixes per bucket the assign step aims at when nothing fuller fits ...
constexpr uint32_t kLdsTargetMid = 352;          // ... after these two have been tried
constexpr uint32_t kLdsTargetHigh = 448;
constexpr uint32_t kLdsTargets = 5;             // attempts in all: the two above, kLdsTarget, and two more 64 and 128 below it
constexpr uint32_t kLdsCap = 512;                // ... and the most a bucket may hold
constexpr int kLdsSubBits = 9;
constexpr uint32_t kLdsSub = 1u << kLdsSubBits;  // sub-bins of a bucket
constexpr int kLdsFieldBits = 32 - kLdsSubBits;  // key bits an element keeps (beside its 9-bit index)
constexpr uint32_t kLdsMaxBin = 128;             // a fuller sub-bin gives the block up (the round counts of the settle step are checked up to 160)
constexpr int kLdsVecs = (int)(kLdsCap / 64u);   // 64-element vectors of the fullest bucket
constexpr uint32_t kLdsGroup = 16;               // consecutive positions one lane handles in a sweep
constexpr uint32_t kLdsTile = kLdsGroup * kLdsThreads;
constexpr int kLdsPosBits = 19;                  // positions 0 .. 262144
constexpr uint32_t kLdsPosMask = (1u << 19) - 1u;
constexpr int kLdsCodeShift = 29;                // a scratch word: position | bucket << 19 | (two-bit text) code of the symbol before << 29
constexpr int kLdsDepth = 32;                    // windows two suffixes are compared over before the block is given up (512 symbols of two-bit text, 128 bytes)
constexpr uint32_t kLdsMaxDups = 110;            // byte alphabets: colliding sample windows (of 1024) up to which a block is tried
constexpr uint32_t kLdsMaxFlat = 32;             // sample windows (of 1024) of one repeated byte up to which a block is tried
constexpr uint32_t kLdsMaxPeriod = 4096;         // whole-block periods up to here are looked for before anything else
constexpr int kLdsClockSlot = 32;                // phase clocks: u64 words from queue + 2 * kLdsClockSlot

struct LdsWaveArea {
    uint32_t sorted[kLdsCap + kWave];  // the bucket's elements in sub-bin order, then in order (+ a slot per lane for writes that count for nothing)
    uint32_t cnt[kLdsSub + 8];         // sub-bin counters (+ one for lanes without an element), then their starts; after the scatter: position by index
};

struct LdsShared {
    uint32_t text[kLdsTextWords];
    union {
        uint32_t fine[kLdsFine];                     // count, assign
        struct {
            uint32_t stage[kLdsTile];                // scratch words (position | bucket << 19 | ...), grouped by bucket
            uint16_t lut[kLdsFine];                  // fine bin -> bucket (beyond `fine`, which the assign step still reads)
            uint32_t tcnt[kLdsBuckets];              // per tile: elements per bucket
            uint32_t tstart[kLdsBuckets];            // per tile: where the bucket's run starts in `stage`
            uint32_t delta[kLdsBuckets];             // per tile: scratch slot of the run's first element minus its stage index
        } part;
        struct {
            LdsWaveArea wave[kLdsWaves];
            uint32_t dstart[kLdsBuckets + 1];        // first row of every bucket
            uint16_t dflo[kLdsBuckets + 2];          // first fine bin of every bucket
        } sort;
    } u;
    uint8_t present[256];
    uint8_t code[256];             // byte -> code 0..3; 0x80: not in the alphabet (the delimiter included)
    uint32_t quad[256];            // two-bit text: four two-bit fields of a dword of text -> its four codes (see the pack step)
    uint32_t wsum[kLdsWaves];
    uint64_t ball[4];
    uint32_t cb[4];                // code -> byte
    uint32_t blk, bad, next, dups, flat;
    uint32_t period_cand, period_bad;  // detect_block_period's words
};
static_assert(sizeof(LdsShared) <= 160u * 1024u, "one workgroup's LDS");

struct __attribute__((packed, aligned(1))) LdsBytes16 {
    uint32_t x, y, z, w;
};

__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
    v = max(v, dpp_or_zero<0x111, 0xF>(v));
    v = max(v, dpp_or_zero<0x112, 0xF>(v));
    v = max(v, dpp_or_zero<0x114, 0xF>(v));
    v = max(v, dpp_or_zero<0x118, 0xF>(v));
    v = max(v, dpp_or_zero<0x142, 0xA>(v));
    v = max(v, dpp_or_zero<0x143, 0xC>(v));
    return wave_read(v, kWave - 1);
}

// A value the compiler has to take as new at this point: address arithmetic that depends only on the
// thread index is otherwise computed once at the top of the kernel for every use in every phase, and
// what does not fit the registers by then is spilled and reloaded inside the hot loops.
__device__ __forceinline__ uint32_t opaque(uint32_t v) {
    asm volatile("" : "+v"(v));
    return v;
}

// Keeps the accesses of one wave's lanes to the wave's own LDS area in program order: lanes hand values
// to each other through that area without a workgroup barrier (a wave's LDS instructions execute in order).
__device__ __forceinline__ void wave_sync() {
    compiler_fence();
    __builtin_amdgcn_wave_barrier();
    compiler_fence();
}

template <int BITS>
struct LdsFmt {
    static constexpr int kPerWord = 32 / BITS;                  // symbols per 32-bit word
    static constexpr int kWordShift = BITS == 2 ? 4 : 2;        // log2(kPerWord)
    static constexpr int kGroupWords = (int)kLdsGroup / kPerWord;  // words that hold a lane's 16 positions
};

// The 32-bit window that starts at symbol p (p <= len: the words behind the text hold the virtual symbols).
template <int BITS>
__device__ __forceinline__ uint32_t lds_window(const uint32_t* text, uint32_t p) {
    constexpr int WS = LdsFmt<BITS>::kWordShift;
    const uint32_t w = p >> WS, s = (p & ((1u << WS) - 1u)) * (uint32_t)BITS;
    const uint64_t both = ((uint64_t)text[w] << 32) | text[w + 1u];
    return (uint32_t)((both << s) >> 32);
}

// The windows of the 16 positions 16 g .. 16 g + 15.
template <int BITS>
__device__ __forceinline__ void lds_group_windows(const uint32_t* text, uint32_t g, uint32_t (&key)[kLdsGroup]) {
    constexpr int GW = LdsFmt<BITS>::kGroupWords, PW = LdsFmt<BITS>::kPerWord;
    uint32_t w[GW + 1];
#pragma unroll
    for (int i = 0; i <= GW; ++i) w[i] = text[g * (uint32_t)GW + (uint32_t)i];
#pragma unroll
    for (int j = 0; j < (int)kLdsGroup; ++j) {
        const int q = j / PW, r = (j % PW) * BITS;
        key[j] = r ? __builtin_amdgcn_alignbit(w[q], w[q + 1], 32 - r) : w[q];
    }
}

// The byte before suffix p (the delimiter before suffix 0): what the suffix's row of the transform holds.
template <int BITS>
__device__ __forceinline__ uint32_t lds_prev_byte(const uint32_t* text, uint32_t p, uint32_t cbytes, uint32_t delim) {
    constexpr int WS = LdsFmt<BITS>::kWordShift;
    const uint32_t q = p - 1u;
    const uint32_t word = text[(p ? q : 0u) >> WS];
    const uint32_t sym = (word >> (32u - (uint32_t)BITS - (q & ((1u << WS) - 1u)) * (uint32_t)BITS)) & ((1u << BITS) - 1u);
    const uint32_t byte = BITS == 8 ? sym : (cbytes >> (8u * sym)) & 255u;
    return p ? byte : delim;
}

// Whether suffix p sorts before suffix q (p != q), window by window.  `above`: the delimiter is above
// every text symbol.  Sets *unresolved when kLdsDepth windows do not tell them apart.
template <int BITS>
__device__ __forceinline__ bool lds_suffix_less(const uint32_t* text, uint32_t p, uint32_t q, uint32_t len, bool above,
                                                bool* unresolved) {
    constexpr uint32_t K = LdsFmt<BITS>::kPerWord;
#pragma unroll 1
    for (int k = 0; k < kLdsDepth; ++k) {
        const uint32_t a = lds_window<BITS>(text, p), b = lds_window<BITS>(text, q);
        if (a != b) return a < b;
        if (max(p, q) + K > len) return above ? p < q : p > q;  // one of the equal windows holds the delimiter
        p += K;
        q += K;
    }
    *unresolved = true;
    return p < q;
}

// One bucket, by one wave: `pos` holds its cnt positions (element i in lane i % 64 of pos[i / 64]); E =
// elements per lane once they are sorted (cnt <= 64 E, E >= 2).  The bucket's keys lie in [base, base + 2^kb).
//   count    every element's sub-bin (the top 9 of the kb key bits below the base): LDS add-with-return
//   scan     sub-bin starts; the fullest sub-bin tells how far an element can be from its place
//   scatter  elements (key bits << 9 | index) into sub-bin order; the counters' words then hold the positions by index
//   settle   lane l takes elements E l .. E l + E - 1 into registers, and rounds of compare-exchanges between
//            neighbours -- inside the lane, then across the lane border through DPP -- finish the order.
//            Every round applies each neighbour pair once; for sub-bins of m elements (m + 1) / 2 rounds (E even) or
//            m - 1 - (m - 1) / 4 rounds (E odd) sort every input: the network has adjacent comparators only, so it
//            sorts everything if it sorts the reversed sequence, which was checked for all m <= 160, every E and
//            every place a sub-bin can start (E even is plain odd-even transposition, which needs m steps).
//   rows     element x of the bucket stands in row s0 + x: the lane stores the byte before each of its
//            suffixes; elements whose kept key bits tie are first put in order through the text.
template <int BITS, int E>
__device__ __forceinline__ void lds_sort_bucket(LdsShared& sh, LdsWaveArea& wa, const uint32_t (&pos)[kLdsVecs], int lane,
                                                uint32_t s0, uint32_t cnt, uint32_t base, int kb, uint32_t len, bool above,
                                                uint32_t cbytes, uint32_t delim, uint8_t* __restrict__ out, bool& give_up) {
    static_assert(E >= 2 && E <= kLdsVecs, "elements per lane");
    const int sub_shift = kb - kLdsSubBits;
    const int field_bits = min(kb, kLdsFieldBits);
    const int field_shift = kb - field_bits;
    uint4* c4 = reinterpret_cast<uint4*>(wa.cnt);
    c4[2 * lane] = make_uint4(0u, 0u, 0u, 0u);
    c4[2 * lane + 1] = make_uint4(0u, 0u, 0u, 0u);
    wave_sync();
    uint32_t el[E], slot[E];
    // (lanes without an element take part in everything, aimed at words nobody reads: no branches, so the
    // LDS reads of all vectors are in flight together)
#pragma unroll
    for (int v = 0; v < E; ++v) {
        const uint32_t i = (uint32_t)(v * kWave + lane);
        const uint32_t rel = lds_window<BITS>(sh.text, pos[v] & kLdsPosMask) - base;
        el[v] = ((rel >> field_shift) << kLdsSubBits) | i;
        slot[v] = atomicAdd(&wa.cnt[i < cnt ? rel >> sub_shift : kLdsSub], 1u);
    }
    wave_sync();
    uint32_t maxbin;
    {
        const uint4 a0 = c4[2 * lane], a1 = c4[2 * lane + 1];
        const uint32_t tot = a0.x + a0.y + a0.z + a0.w + a1.x + a1.y + a1.z + a1.w;
        const uint32_t m = max(max(max(a0.x, a0.y), max(a0.z, a0.w)), max(max(a1.x, a1.y), max(a1.z, a1.w)));
        uint32_t run = wave_inclusive_scan(tot) - tot;
        uint4 b0, b1;
        b0.x = run; run += a0.x;
        b0.y = run; run += a0.y;
        b0.z = run; run += a0.z;
        b0.w = run; run += a0.w;
        b1.x = run; run += a1.x;
        b1.y = run; run += a1.y;
        b1.z = run; run += a1.z;
        b1.w = run;
        c4[2 * lane] = b0;
        c4[2 * lane + 1] = b1;
        maxbin = wave_max(m);
    }
    if (maxbin > kLdsMaxBin) {  // wave-uniform: many suffixes under one key prefix
        give_up = true;
        return;
    }
    wave_sync();
#pragma unroll
    for (int v = 0; v < E; ++v) slot[v] += wa.cnt[min(el[v] >> field_bits, kLdsSub - 1u)];
    wave_sync();
#pragma unroll
    for (int v = 0; v < E; ++v) {
        const uint32_t i = (uint32_t)(v * kWave + lane);
        wa.sorted[i < cnt ? slot[v] : kLdsCap + (uint32_t)lane] = el[v];
        wa.cnt[i] = pos[v];
    }
    wave_sync();
    uint32_t a[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const uint32_t x = (uint32_t)(E * lane + j);
        const uint32_t t = wa.sorted[x];
        a[j] = x < cnt ? t : 0xFFFFFFFFu;
    }
    const uint32_t rounds = maxbin > 1u ? ((E & 1) ? maxbin - 1u - (maxbin - 1u) / 4u : (maxbin + 1u) / 2u) : 0u;
    for (uint32_t r = 0; r < rounds; ++r) {
#pragma unroll
        for (int j = 0; j + 1 < E; j += 2) {
            const uint32_t lo = min(a[j], a[j + 1]), hi = max(a[j], a[j + 1]);
            a[j] = lo;
            a[j + 1] = hi;
        }
#pragma unroll
        for (int j = 1; j + 1 < E; j += 2) {
            const uint32_t lo = min(a[j], a[j + 1]), hi = max(a[j], a[j + 1]);
            a[j] = lo;
            a[j + 1] = hi;
        }
        const uint32_t up = wave_from_above(a[0], 0xFFFFFFFFu), dn = wave_from_below(a[E - 1], 0u);
        a[E - 1] = min(a[E - 1], up);
        a[0] = max(a[0], dn);
    }
    // elements that share their kept key bits with a neighbour
    uint32_t tmask = 0;  // bit j: a[j] ties with its successor
    {
        const uint32_t nxt = wave_from_above(a[0], 0xFFFFFFFFu);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const uint32_t succ = j + 1 < E ? a[j + 1 < E ? j + 1 : j] : nxt;
            if ((((a[j] ^ succ) >> kLdsSubBits) == 0u) & ((uint32_t)(E * lane + j) + 1u < cnt)) tmask |= 1u << j;
        }
    }
    const uint32_t tied = tmask | (tmask << 1) | wave_from_below((tmask >> (E - 1)) & 1u, 0u);
    const bool any_tied = __ballot(tied != 0u) != 0ull;
    if (any_tied) {  // wave-uniform: the tied runs are looked at in LDS
#pragma unroll
        for (int j = 0; j < E; ++j) wa.sorted[E * lane + j] = a[j];
        wave_sync();
    }
    {
        uint32_t byte[E];
#pragma unroll
        for (int j = 0; j < E; ++j) {  // (a lane's padding elements read some word of the LDS)
            const uint32_t word = wa.cnt[a[j] & (kLdsCap - 1u)];
            if (BITS == 2) byte[j] = __builtin_amdgcn_perm(delim, cbytes, word >> kLdsCodeShift);  // codes 0 .. 3: a byte of cbytes; 4: the delimiter
            else byte[j] = lds_prev_byte<BITS>(sh.text, word & kLdsPosMask, cbytes, delim);
        }
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const uint32_t x = (uint32_t)(E * lane + j);
            if (x < cnt && !((tied >> j) & 1u)) out[s0 + x] = (uint8_t)byte[j];
        }
    }
    if (any_tied) {
        bool unresolved = false;
#pragma unroll 1
        for (uint32_t j = 0; j < (uint32_t)E; ++j) {
            const uint32_t d = (uint32_t)(E * lane) + j;
            if (((tied >> j) & 1u) && d < cnt) {
                const uint32_t e = wa.sorted[d];
                const uint32_t p = wa.cnt[e & (kLdsCap - 1u)] & kLdsPosMask;
                uint32_t lo = d, hi = d + 1u;
                while (lo > 0u && ((wa.sorted[lo - 1u] ^ e) >> kLdsSubBits) == 0u) --lo;
                while (hi < cnt && ((wa.sorted[hi] ^ e) >> kLdsSubBits) == 0u) ++hi;
                uint32_t row = lo;
                for (uint32_t x = lo; x < hi; ++x) {
                    const uint32_t q = wa.cnt[wa.sorted[x] & (kLdsCap - 1u)] & kLdsPosMask;
                    if (q != p && lds_suffix_less<BITS>(sh.text, q, p, len, above, &unresolved)) ++row;
                }
                out[s0 + row] = (uint8_t)lds_prev_byte<BITS>(sh.text, p, cbytes, delim);
            }
        }
        if (__ballot(unresolved) != 0ull) give_up = true;  // a repeat longer than the windows looked at
    }
    wave_sync();
}

// The pack step of one block (see lds_forward_block), a function of its own: the kernel around it sits at its register
// limit, and inlined, this step's temporaries moved the allocation of the sort phase for the worse (17.7 -> 18.8 ms).
template <int BITS>
__device__ __attribute__((noinline)) void lds_pack_block(const uint8_t* __restrict__ s, uint32_t len, uint32_t delim, LdsShared& sh,
                                                          uint32_t cbytes, uint32_t fs, uint32_t vd, uint32_t vfill, uint32_t nwords) {
    constexpr int PW = LdsFmt<BITS>::kPerWord;
    const int tid = threadIdx.x;
    for (uint32_t i = (uint32_t)tid; i < kLdsFine; i += kLdsThreads) sh.u.fine[i] = 0;
    {
        uint32_t flags = 0;
        constexpr uint32_t kWordsPer16 = 16u / (uint32_t)PW;  // words a 16-byte load fills: 1 or 4
        const uint32_t nvec = (nwords + kWordsPer16 - 1u) / kWordsPer16;
        constexpr int U = 4;
        for (uint32_t v0 = (uint32_t)tid; v0 < nvec; v0 += U * kLdsThreads) {
            uint32_t d[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t v = v0 + (uint32_t)u * kLdsThreads, p0 = 16u * v;
                d[u][0] = d[u][1] = d[u][2] = d[u][3] = 0;
                if (v < nvec && p0 + 16u <= len) {
                    const LdsBytes16 x = *reinterpret_cast<const LdsBytes16*>(s + p0);
                    d[u][0] = x.x; d[u][1] = x.y; d[u][2] = x.z; d[u][3] = x.w;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t v = v0 + (uint32_t)u * kLdsThreads, p0 = 16u * v;
                if (v >= nvec) continue;
                if (p0 + 16u > len) {  // the block's ragged end and the virtual symbols: byte by byte
#pragma unroll
                    for (uint32_t k = 0; k < 16u; ++k) {
                        const uint32_t p = p0 + k;
                        uint32_t byte;
                        if (p < len) {
                            byte = s[p];
                            if (BITS == 8 && byte == delim) flags |= 0x80u;
                        } else if (BITS == 8) {
                            byte = p == len ? vd : vfill;
                        } else {
                            byte = cbytes & 255u;  // any text symbol: replaced below
                        }
                        d[u][k >> 2] |= byte << (8u * (k & 3u));
                    }
                }
                if (BITS == 8) {
                    uint32_t wd[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        wd[k] = __builtin_bswap32(d[u][k]);
                        if (p0 + 16u <= len) {  // a zero byte of x ^ delimiter: the delimiter occurs in the text
                            const uint32_t x = d[u][k] ^ (delim * 0x01010101u);
                            flags |= ((x - 0x01010101u) & ~x & 0x80808080u) ? 0x80u : 0u;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (4u * v + (uint32_t)k < kLdsTextWords) sh.text[4u * v + (uint32_t)k] = wd[k];
                } else if (fs < 8u && p0 + 16u <= len) {
                    // four symbols at a time: bits fs, fs + 1 of the byte values tell the block's symbols apart, so the
                    // four fields of a dword index a table of its four codes -- as a byte-permute selector, which turns
                    // them back into bytes (a dword that does not come back as it was holds a byte outside the alphabet),
                    // and packed the way the text word wants them (bits 2-7, the last code being the selector's top byte)
                    uint32_t word = 0, diff = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t t = (d[u][j] >> fs) & 0x03030303u;
                        const uint32_t y = t | (t >> 6);
                        const uint32_t e = sh.quad[(y | (y >> 12)) & 255u];
                        diff |= __builtin_amdgcn_perm(0u, cbytes, e & 0x03030303u) ^ d[u][j];
                        word |= ((e & 0xFCu) | (e >> 24)) << (24 - 8 * j);
                    }
                    if (diff) flags |= 0x80u;
                    sh.text[v] = word;
                } else {
                    uint32_t word = 0;
#pragma unroll
                    for (uint32_t k = 0; k < 16u; ++k) {
                        uint32_t c = sh.code[(d[u][k >> 2] >> (8u * (k & 3u))) & 255u];
                        const uint32_t p = p0 + k;
                        if (p0 + 16u > len && p >= len) c = p == len ? vd : vfill;
                        flags |= c;
                        word |= (c & 3u) << (30u - 2u * k);
                    }
                    sh.text[v] = word;
                }
            }
        }
        if (flags & 0x80u) sh.bad = 1;
    }
}

// One block.  Returns false when the block is given up (workgroup-uniform).
template <int BITS>
__device__ __forceinline__ bool lds_forward_block(const uint8_t* __restrict__ s, uint32_t len, uint32_t delim,
                                                  uint8_t* __restrict__ out, uint32_t* sa, LdsShared& sh, uint32_t sigma,
                                                  uint32_t dr, uint32_t cbytes, uint32_t fs, PhaseClock& pc) {
    constexpr int PW = LdsFmt<BITS>::kPerWord, WS = LdsFmt<BITS>::kWordShift;
    int tid = (int)opaque(threadIdx.x);
    const int w = wave_id();
    const uint32_t n = len + 1u;
    const bool above = BITS == 2 && dr == sigma;
    const uint32_t top = (1u << BITS) - 1u;
    // the virtual symbols at position len and behind it
    const uint32_t vd = BITS == 8 ? delim : above ? top : dr, vfill = above ? top : 0u;
    const uint32_t nwords = (len >> WS) + 3u;  // a window at position len reads the word of len and the next one

    // ---- pack ----------------------------------------------------------------------------------
    lds_pack_block<BITS>(s, len, delim, sh, cbytes, fs, vd, vfill, nwords);
    __syncthreads();
    if (sh.bad) return false;  // workgroup-uniform
    pc.lap(kLdsClockSlot + 1);

    // ---- count: fine bin of every suffix ---------------------------------------------------------
    tid = (int)opaque((uint32_t)tid);
    const uint32_t ngroups = (n + kLdsGroup - 1u) / kLdsGroup;
    for (uint32_t g = (uint32_t)tid; g < ngroups; g += kLdsThreads) {
        uint32_t key[kLdsGroup];
        lds_group_windows<BITS>(sh.text, g, key);
        const uint32_t p0 = g * kLdsGroup;
        if (p0 + kLdsGroup <= n) {
#pragma unroll
            for (int j = 0; j < (int)kLdsGroup; ++j) atomicAdd(&sh.u.fine[key[j] >> kLdsFineShift], 1u);
        } else {
#pragma unroll
            for (int j = 0; j < (int)kLdsGroup; ++j)
                if (p0 + (uint32_t)j < n) atomicAdd(&sh.u.fine[key[j] >> kLdsFineShift], 1u);
        }
    }
    __syncthreads();
    pc.lap(kLdsClockSlot + 2);

    // ---- assign: fine bins -> buckets ------------------------------------------------------------
    // bin f goes to bucket (first slot of f) / target: buckets are runs of consecutive bins, hold less
    // than target + (their last bin) suffixes, and there are at most 1024 of them.
    tid = (int)opaque((uint32_t)tid);
    // The fuller the buckets, the fewer of them (a bucket's sort has fixed costs: counters, scan, hand-over), but a bucket
    // holds up to its target plus its last bin and must not exceed kLdsCap: the largest target first, which uniform keys
    // fit, then the more forgiving ones (measured on 4 GiB of DNA in 256 KiB blocks: target 288 22.2 ms, 352 20.9,
    // 384 19.3, 416 17.9, 448 17.6).
    uint32_t target = 0, nb = 0;
    uint32_t my_start = 0, my_end = 0, my_flo = 0;
#pragma unroll 1
    for (uint32_t attempt = 0; attempt < kLdsTargets; ++attempt) {
        // 448, 352, 288, 224, 160 -- but never so low that the block would need more than kLdsBuckets buckets (a 256 KiB
        // block: 257 at least, so its last two attempts repeat the third)
        const uint32_t want = attempt == 0u ? kLdsTargetHigh : attempt == 1u ? kLdsTargetMid : kLdsTarget - 64u * (attempt - 2u);
        target = max(want, (n + kLdsBuckets - 1u) / kLdsBuckets);
        nb = (n - 1u) / target + 1u;
        const uint32_t magic = 0xFFFFFFFFu / target + 1u;
        auto bucket_of = [&](uint32_t slot) {
            uint32_t q = __umulhi(slot, magic);
            if (q * target > slot) --q;
            return min(q, kLdsBuckets - 1u);
        };
        constexpr int B = (int)(kLdsFine / kLdsThreads);  // 8 bins per lane
        uint32_t c[B];
        uint32_t sum = 0;
#pragma unroll
        for (int i = 0; i < B; ++i) {
            c[i] = sh.u.fine[(uint32_t)tid * B + i];
            sum += c[i];
        }
        const uint32_t before = tid ? sh.u.fine[(uint32_t)tid * B - 1u] : 0u;
        uint32_t total;
        uint32_t run = block_exclusive_scan<kLdsThreads>(sum, sh.wsum, total);
        uint32_t bprev = tid ? bucket_of(run - before) : 0xFFFFFFFFu;
        sh.u.part.tcnt[tid] = 0;
#pragma unroll
        for (int i = 0; i < B; ++i) {
            const uint32_t f = (uint32_t)tid * B + i;
            const uint32_t bk = bucket_of(run);
            sh.u.part.lut[f] = (uint16_t)bk;
            for (uint32_t k = bprev + 1u; k <= bk && bk != bprev; ++k) {  // bucket bk, and the empty ones a long bin skips
                sh.u.part.tstart[k] = run;
                sh.u.part.delta[k] = f;
            }
            bprev = bk;
            run += c[i];
        }
        if (tid == kLdsThreads - 1)
            for (uint32_t k = bprev + 1u; k < nb; ++k) {  // nothing beyond the last bin that has suffixes
                sh.u.part.tstart[k] = n;
                sh.u.part.delta[k] = kLdsFine;
            }
        __syncthreads();
        my_start = (uint32_t)tid < nb ? sh.u.part.tstart[tid] : n;
        my_end = (uint32_t)tid + 1u < nb ? sh.u.part.tstart[tid + 1] : n;
        my_flo = (uint32_t)tid < nb ? sh.u.part.delta[tid] : kLdsFine;
        if (my_end - my_start > kLdsCap) sh.bad = 1;
        __syncthreads();
        const bool fits = sh.bad == 0u;  // workgroup-uniform
        __syncthreads();
        if (fits) break;
        if (attempt + 1u == kLdsTargets) return false;  // a key prefix too common for a wave's bucket
        if (tid == 0) sh.bad = 0;
        __syncthreads();
    }
    pc.lap(kLdsClockSlot + 3);

    // ---- partition: positions into their buckets' stretches of `sa`, tile by tile ----------------------
    {
        uint32_t cursor = my_start;
        const uint32_t ntiles = (n + kLdsTile - 1u) / kLdsTile;
        for (uint32_t tile = 0; tile < ntiles; ++tile) {
            tid = (int)opaque((uint32_t)tid);
            const uint32_t g = tile * kLdsThreads + (uint32_t)tid, p0 = g * kLdsGroup;
            // per position: tag = bucket << 19 (| for two-bit text: code of the symbol before it << 29, 4 = none:
            // position 0) and the slot among the tile's elements of that bucket; 0xFFFFFFFF: no position
            uint32_t tag[kLdsGroup], slot[kLdsGroup];
            if (p0 < n) {
                uint32_t key[kLdsGroup];
                lds_group_windows<BITS>(sh.text, g, key);
                uint32_t before = 0;  // the symbol before position p0
                if (BITS == 2) before = g ? (sh.text[g - 1u] & 3u) << kLdsCodeShift : 4u << kLdsCodeShift;
                if (p0 + kLdsGroup <= n) {  // all of a tile's lanes but for the block's last positions
#pragma unroll
                    for (int j = 0; j < (int)kLdsGroup; ++j) tag[j] = sh.u.part.lut[key[j] >> kLdsFineShift];
#pragma unroll
                    for (int j = 0; j < (int)kLdsGroup; ++j) {
                        slot[j] = atomicAdd(&sh.u.part.tcnt[tag[j]], 1u);
                        tag[j] <<= kLdsPosBits;
                        if (BITS == 2) tag[j] |= j ? (key[j ? j - 1 : 0] >> 30) << kLdsCodeShift : before;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < (int)kLdsGroup; ++j) {
                        tag[j] = 0xFFFFFFFFu;
                        slot[j] = 0;
                        if (p0 + (uint32_t)j < n) {
                            const uint32_t bk = sh.u.part.lut[key[j] >> kLdsFineShift];
                            slot[j] = atomicAdd(&sh.u.part.tcnt[bk], 1u);
                            tag[j] = bk << kLdsPosBits;
                            if (BITS == 2) tag[j] |= j ? (key[j ? j - 1 : 0] >> 30) << kLdsCodeShift : before;
                        }
                    }
                }
            }
            __syncthreads();
            const uint32_t c = sh.u.part.tcnt[tid];
            sh.u.part.tcnt[tid] = 0;
            uint32_t total;
            const uint32_t excl = block_exclusive_scan<kLdsThreads>(c, sh.wsum, total);
            sh.u.part.tstart[tid] = excl;
            sh.u.part.delta[tid] = cursor - excl;
            cursor += c;
            __syncthreads();
            if (p0 + kLdsGroup <= n) {
                uint32_t at[kLdsGroup];
#pragma unroll
                for (int j = 0; j < (int)kLdsGroup; ++j) at[j] = sh.u.part.tstart[(tag[j] >> kLdsPosBits) & (kLdsBuckets - 1u)] + slot[j];
#pragma unroll
                for (int j = 0; j < (int)kLdsGroup; ++j) sh.u.part.stage[at[j]] = tag[j] | (p0 + (uint32_t)j);
            } else if (p0 < n) {
#pragma unroll
                for (int j = 0; j < (int)kLdsGroup; ++j)
                    if (tag[j] != 0xFFFFFFFFu)
                        sh.u.part.stage[sh.u.part.tstart[(tag[j] >> kLdsPosBits) & (kLdsBuckets - 1u)] + slot[j]] = tag[j] | (p0 + (uint32_t)j);
            }
            __syncthreads();
            const uint32_t tcount = min(kLdsTile, n - tile * kLdsTile);
            if (tcount == kLdsTile) {  // workgroup-uniform
                uint32_t v[kLdsGroup], dl[kLdsGroup];
#pragma unroll
                for (int i = 0; i < (int)kLdsGroup; ++i) v[i] = sh.u.part.stage[(uint32_t)i * kLdsThreads + (uint32_t)tid];
#pragma unroll
                for (int i = 0; i < (int)kLdsGroup; ++i) dl[i] = sh.u.part.delta[(v[i] >> kLdsPosBits) & (kLdsBuckets - 1u)];
#pragma unroll
                for (int i = 0; i < (int)kLdsGroup; ++i) store_global(sa + (dl[i] + (uint32_t)i * kLdsThreads + (uint32_t)tid), v[i]);
            } else {
#pragma unroll
                for (int i = 0; i < (int)kLdsGroup; ++i) {
                    const uint32_t j = (uint32_t)i * kLdsThreads + (uint32_t)tid;
                    if (j < tcount) {
                        const uint32_t v = sh.u.part.stage[j];
                        store_global(sa + (sh.u.part.delta[(v >> kLdsPosBits) & (kLdsBuckets - 1u)] + j), v);
                    }
                }
            }
        }
        __syncthreads();
    }
    pc.lap(kLdsClockSlot + 4);

    // ---- sort: one wave per bucket ---------------------------------------------------------------
    tid = (int)opaque((uint32_t)tid);
    sh.u.sort.dstart[tid] = my_start;
    sh.u.sort.dflo[tid] = (uint16_t)my_flo;
    if (tid == 0) {
        sh.u.sort.dstart[kLdsBuckets] = n;
        sh.u.sort.dflo[kLdsBuckets] = (uint16_t)kLdsFine;
        sh.next = 0;
    }
    __syncthreads();
    LdsWaveArea& wa = sh.u.sort.wave[w];
    {
        const int lane = (int)opaque((uint32_t)tid & 63u);
        auto grab = [&]() {
            uint32_t k = 0;
            if (lane == 0) k = atomicAdd(&sh.next, 1u);
            return wave_read(k, 0);
        };
        // the bucket's positions: element i in lane i % 64 of p[i / 64]
        auto fetch = [&](uint32_t k, uint32_t (&p)[kLdsVecs]) {
            const uint32_t s0 = sh.u.sort.dstart[k], cnt = sh.u.sort.dstart[k + 1u] - s0;
#pragma unroll
            for (int v = 0; v < kLdsVecs; ++v) {
                p[v] = 0;
                if ((uint32_t)(v * kWave) < cnt) {  // wave-uniform
                    const uint32_t i = (uint32_t)(v * kWave + lane);
                    if (i < cnt) p[v] = load_global(sa + s0 + i);
                }
            }
        };
        uint32_t pos[kLdsVecs], ahead[kLdsVecs];
        uint32_t k = grab();
        if (k < nb) fetch(k, pos);
        while (k < nb && !__hip_atomic_load(&sh.bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {  // wave-uniform
            const uint32_t k2 = grab();
            if (k2 < nb) fetch(k2, ahead);  // in flight while this bucket is sorted
            const uint32_t s0 = sh.u.sort.dstart[k], cnt = sh.u.sort.dstart[k + 1u] - s0;
            if (cnt) {
                const uint32_t f0 = sh.u.sort.dflo[k], f1 = sh.u.sort.dflo[k + 1u];
                // the bucket's keys lie in [f0 << 19, f1 << 19): kb bits hold a key's offset from the base
                const uint32_t base = f0 << kLdsFineShift;
                const int kb = kLdsFineShift + bit_length(f1 - f0 - 1u);
                bool give_up = false;
                switch ((cnt + (uint32_t)kWave - 1u) >> 6) {
                    case 1:
                    case 2: lds_sort_bucket<BITS, 2>(sh, wa, pos, lane, s0, cnt, base, kb, len, above, cbytes, delim, out, give_up); break;
                    case 3: lds_sort_bucket<BITS, 3>(sh, wa, pos, lane, s0, cnt, base, kb, len, above, cbytes, delim, out, give_up); break;
                    case 4: lds_sort_bucket<BITS, 4>(sh, wa, pos, lane, s0, cnt, base, kb, len, above, cbytes, delim, out, give_up); break;
                    case 5: lds_sort_bucket<BITS, 5>(sh, wa, pos, lane, s0, cnt, base, kb, len, above, cbytes, delim, out, give_up); break;
                    case 6: lds_sort_bucket<BITS, 6>(sh, wa, pos, lane, s0, cnt, base, kb, len, above, cbytes, delim, out, give_up); break;
                    case 7: lds_sort_bucket<BITS, 7>(sh, wa, pos, lane, s0, cnt, base, kb, len, above, cbytes, delim, out, give_up); break;
                    default: lds_sort_bucket<BITS, 8>(sh, wa, pos, lane, s0, cnt, base, kb, len, above, cbytes, delim, out, give_up); break;
                }
                if (give_up) {  // wave-uniform
                    if (lane == 0) sh.bad = 1;
                    break;
                }
            }
            k = k2;
#pragma unroll
            for (int v = 0; v < kLdsVecs; ++v) pos[v] = ahead[v];
        }
    }
    __syncthreads();
    pc.lap(kLdsClockSlot + 5);
    return sh.bad == 0u;
}

// The sample every block starts with: 16 bytes per lane spread over the block mark the alphabet (sh.present), and two
// counts say what the bytes are like.  sh.dups: how many of the 1024 lanes' first four bytes fall on a value another lane
// has -- text and structured binary data repeat their four-byte strings (185-1023 lanes collide over the Canterbury files
// at any block size), uniform bytes over sixteen values or more do not (40-90, nearly all of them table slots shared by
// chance).  sh.flat: how many lanes read one byte value sixteen times -- long runs (scanned images, zero-filled tables)
// tie deeper than the in-LDS sort follows, whatever the alphabet.  Kept out of line: the kernel around it sits at its
// register limit, and the sample's temporaries moved its allocation for the worse.
__device__ __attribute__((noinline)) void lds_sample_block(const uint8_t* __restrict__ s, uint32_t len, LdsShared& sh) {
    const int tid = threadIdx.x;
    const uint32_t at = (uint32_t)(((uint64_t)tid * (len - 16u)) / (kLdsThreads - 1u));
    const LdsBytes16 x = *reinterpret_cast<const LdsBytes16*>(s + at);
    const uint32_t d[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int k = 0; k < 16; ++k) sh.present[(d[k >> 2] >> (8 * (k & 3))) & 255u] = 1;
    if (tid == 0) sh.dups = sh.flat = 0;
    for (uint32_t i = (uint32_t)tid; i < kLdsFine; i += kLdsThreads) sh.u.fine[i] = 0;
    __syncthreads();
    const uint32_t before = atomicAdd(&sh.u.fine[(d[0] * 0x9E3779B1u) >> (32 - kLdsFineBits)], 1u);
    const uint64_t hit = __ballot(before != 0u);
    if (lane_id() == 0 && hit) atomicAdd(&sh.dups, (uint32_t)__popcll(hit));
    const uint64_t run = __ballot(d[0] == (d[0] & 255u) * 0x01010101u && d[1] == d[0] && d[2] == d[0] && d[3] == d[0]);
    if (lane_id() == 0 && run) atomicAdd(&sh.flat, (uint32_t)__popcll(run));
}

// Blocks of kLdsMinLen .. kLdsMaxLen2 bytes from the queue (queue[7]); a block this kernel does not
// finish is appended to a.normal (queue[5] counts), which the general kernel's next launch works through.
__global__ __launch_bounds__(kLdsThreads) void bwt_forward_lds_kernel(FwdArgs a) {
    __shared__ LdsShared sh;
    const int tid = threadIdx.x, w = wave_id();
    uint32_t* sa = reinterpret_cast<uint32_t*>(a.ws + (uint64_t)blockIdx.x * a.ws_stride);
    PhaseClock pc;
    pc.start(a.queue);
    for (;;) {
        __syncthreads();  // everybody is done with the block before
        if (tid == 0) {
            sh.blk = atomicAdd(&a.queue[7], 1u);
            sh.bad = 0;
        }
        if (tid < 256) sh.present[tid] = 0;
        __syncthreads();
        const uint32_t b = sh.blk;
        if (b >= a.nblocks) break;
        const uint64_t off = (uint64_t)b * a.block_size;
        const uint64_t rem = a.total_len - off;
        const uint32_t len = rem < a.block_size ? (uint32_t)rem : a.block_size;
        const uint8_t* __restrict__ s = a.in + off;
        uint8_t* __restrict__ out = a.out + (uint64_t)b * ((uint64_t)a.block_size + 1u);
        bool done = false, tried = false;
        if (len >= kLdsMinLen && len <= kLdsMaxLen2) {  // workgroup-uniform
            lds_sample_block(s, len, sh);  // the alphabet, and the two counts that keep hopeless blocks away
            __syncthreads();
            const bool present = tid < 256 && sh.present[tid] != 0;
            const uint64_t bal = __ballot(present);
            if (w < 4 && lane_id() == 0) sh.ball[w] = bal;
            if (tid < 4) sh.cb[tid] = 0;
            __syncthreads();
            uint32_t sigma = 0, below = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t c = (uint32_t)__popcll(sh.ball[i]);
                sigma += c;
                if (i < w) below += c;
            }
            const bool delim_seen = ((sh.ball[a.delim >> 6] >> (a.delim & 63u)) & 1ull) != 0ull;
            if (!delim_seen && sigma == 1u) {
                // one byte value throughout, as far as the sample shows: a run, the business of the launch for those --
                // unless the batch is so small that every block has a workgroup to itself: then the two launches' slowest
                // blocks are all that counts, and one launch that starts them all at once does better (see below)
                tried = a.nblocks > gridDim.x;
            } else if (!delim_seen && sigma <= 4u && sh.flat > kLdsMaxFlat) {
                // long stretches of one value among other bytes: runs again, but left to the common-case launch, which puts off
                // what it has to -- a block like that takes its time wherever it goes, and blocks of this kind next to the ones
                // above would make two launches wait for a slow block each (a scanned page in 16 KiB blocks: 2.8 against 1.5 ms)
            } else if (!delim_seen && sigma <= 4u) {
                if (tid < 256) {
                    const uint32_t code = below + (uint32_t)__popcll(bal & lanemask_lt());
                    sh.code[tid] = present ? (uint8_t)code : (uint8_t)0x80;
                    if (present) sh.cb[code] = (uint32_t)tid;
                }
                __syncthreads();
                const uint32_t cbytes = sh.cb[0] | (sh.cb[1] << 8) | (sh.cb[2] << 16) | (sh.cb[3] << 24);
                uint32_t dr = 0;  // text symbols below the delimiter
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i) dr += (i < sigma && ((cbytes >> (8u * i)) & 255u) < a.delim) ? 1u : 0u;
                // a two-bit field of the byte values that tells the symbols apart (A C G T: bits 1 and 2), and the table
                // of the pack step's four-at-a-time path over it
                uint32_t fs = 8;  // none
                for (uint32_t f = 0; f < 7u && fs == 8u; ++f) {
                    uint32_t seen = 0;
                    for (uint32_t i = 0; i < sigma; ++i) seen |= 1u << (((cbytes >> (8u * i)) >> f) & 3u);
                    if ((uint32_t)__popc(seen) == sigma) fs = f;
                }
                if (fs < 8u && tid < 256) {
                    uint32_t word = 0;
                    for (uint32_t k = 0; k < 4u; ++k) {
                        const uint32_t field = ((uint32_t)tid >> (2u * k)) & 3u;
                        uint32_t code = 0;
                        for (uint32_t i = 0; i < sigma; ++i)
                            if ((((cbytes >> (8u * i)) >> fs) & 3u) == field) code = i;
                        word |= (code << (8u * k)) | (k < 3u ? code << (6u - 2u * k) : 0u);
                    }
                    sh.quad[tid] = word;
                }
                __syncthreads();
                pc.lap(kLdsClockSlot + 0);
                tried = true;
                done = lds_forward_block<2>(s, len, a.delim, out, sa, sh, sigma, dr, cbytes, fs, pc);
            } else if (!delim_seen && len <= kLdsMaxLen8 && sh.dups <= kLdsMaxDups) {
                // (bytes whose four-byte strings repeat would fill single sub-bins of the bucket sort beyond what it settles,
                // and the block would be given up after most of the work: left to the general kernels untried)
                pc.lap(kLdsClockSlot + 0);
                tried = true;
                done = lds_forward_block<8>(s, len, a.delim, out, sa, sh, 256u, a.delim, 0u, 8u, pc);
            }
        }
        // Where a block that is not finished goes.  One that was tried and given up has overfull buckets or
        // long repeats -- runs, periodic or repeated text: the launch with the machinery for such blocks
        // (a.deferred), which also looks for whole-block periods itself.  One that never qualified (too many
        // symbols for its length) goes to the common-case launch (a.normal) -- unless it repeats a short unit
        // throughout, which that launch would only notice after a wasted round: looked for here (reads up to
        // kLdsMaxPeriod + 1 KiB of the block; a full comparison only if the unit is found again).
        bool heavy = tried;
        if (!done && !tried && len >= (a.nblocks > gridDim.x ? periodic_min_block<256>() : kPeriodicMinBlock))  // (as the launches behind this one)
            heavy = detect_block_period<kLdsThreads>(s, len, a.in, a.in + a.total_len, sh, kLdsMaxPeriod) != 0u;
        if (tid == 0 && !done) {
            if (heavy) a.deferred[atomicAdd(&a.queue[2], 1u)] = b;
            else a.normal[atomicAdd(&a.queue[5], 1u)] = b;
        }
        pc.count(kLdsClockSlot + (done ? 6 : 7), 1);
    }
}

}  // namespace bwtgpu
