// bwt_inverse_packed.hpp -- inverse BWT of blocks over at most FOUR text symbols (DNA), inside LDS.
//
// The general inverse (bwt_inverse.hpp) spends one dependent 4-byte load from the L2 per output
// byte on the LF table T[i] = C[L[i]] + occ[i] (reference src/inverse_bwt.cpp:23-45 builds the same
// numbers as arrays).  A block whose text has at most four distinct byte values needs no table: the
// last column packs into two bits per row, and
//      LF(row) = C[s] + number of s among the rows above     (s = the row's symbol)
// is a sampled count plus a population count.  Per 64 text rows the workgroup keeps, in LDS,
//      { C[a] + #a above, C[b] + #b above, C[c] + #c above, C[d] + #d above }   (16 bytes)
//      { low bits of the 64 codes, high bits of the 64 codes }                     (16 bytes)
// -- 128 KB for a block of 256 KiB -- so a walk step is two LDS reads and a few dozen integer
// instructions instead of a trip to the L2.  The delimiter's row is taken OUT of the packed rows (the
// rows above it move down by one), so that counts over the packed rows are counts over text symbols.
//
// One workgroup of 1024 lanes per block, everything for that block in one go:
//   (bwt_inverse_packed_classify_kernel lists the blocks of which a sample of 64 bytes allows it)
//   sample   256 bytes spread over the block name the (at most four) text symbols; any other byte value met later,
//            more or fewer than one delimiter, or a block outside the size range leaves the block
//            to the general kernels (its mode stays != kModeDone)
//   pack     16 rows per lane and step: byte -> code through a 256-byte table, two bit planes
//   counts   per-group symbol counts -> exclusive scan -> the samples; C[] as in the reference
//            (unsigned byte order, the delimiter counted where its value falls)
//   walk     the chain is cut where q * A mod 2^k < ids (q = the packed row; the image is the segment's id) and at
//            the delimiter row; lanes take segment ids until none is left, record {length, segment run
//            into} and append every step's two-bit code to a stream of their own in device memory
//   order    pointer jumping over the records: bytes, and segments, from a segment's start to the text's
//            start -- which ranks the segments along the text
//   copy     lane per 16 bytes of a segment: codes from the streams, bytes to the text, stores coalesced
//   (-DBWT_PACK_STREAM=0 builds the earlier variant: walk 1 for the lengths only, the order step, then walk 2 --
//   the same walks again, now writing bytes to where the order step says they go: 22.1 against 18.1 ms on
//   4 GiB of DNA in 256 KiB blocks)
// The result is the chain the reference walks (src/inverse_bwt.cpp:48-62), byte for byte; a
// hand-made block whose LF permutation has several cycles yields the cycle through the delimiter
// row, like the reference, with the shorter length in lens[].
#pragma once
#include <cstddef>
#include "bwt_inverse.hpp"

namespace bwtgpu {

constexpr int kPackThreads = 1024;
constexpr uint32_t kPackMaxRows = 262144u + 1u;  // block_size + 1 at most: the bit planes of more rows do not fit
// Shorter blocks are left to the table kernels, which run several per compute unit: here a block has a
// compute unit to itself, and the end of the walk -- the last lanes finishing the longest segments --
// does not shrink with the block.  Measured on 1 GiB of ACGT text (this kernel / table kernels):
// 8 KiB blocks 24.8 / 11.3 ms, 16 KiB 15.3 / 11.2, 24 KiB 11.6 / 10.7, 32 KiB 9.7 / 11.0, 48 KiB 8.3 / 11.0,
// 64 KiB 7.0 / 11.0, 128 KiB 5.0 / 11.3, 256 KiB 4.0 / 10.8.
constexpr uint32_t kPackMinRows = 28672u;
constexpr uint32_t kPackGroups = (kPackMaxRows + 63u) / 64u;  // groups of 64 rows while the delimiter row is still in
constexpr uint32_t kPackIds = 7936u;             // segment ids (LDS holds one record each); id kPackIds = the chain's first segment
constexpr uint32_t kPackEnd = 8191u;             // "runs into the delimiter row"
constexpr uint32_t kPackNextBits = 13;
constexpr uint32_t kPackStretches = 8;           // walk 2 writes the text in this many stretches, last stretch first
constexpr uint32_t kPackBuckets = 2 * kPackStretches;  // ... and in each stretch the long segments before the short

struct PackShared {
    uint4 planes[kPackGroups];    // per 64 rows: low bits of the codes (x, y), high bits (z, w)
    uint4 samp[kPackGroups];      // per 64 rows: the row LF leads to from the group's first row, for each of the four symbols
    uint32_t seg[kPackIds + 1];   // walk 1, by segment id: bytes << 13 | id of the segment run into (kPackEnd: none)
                                  // walk 2, the segments on the chain in the order they are handed out:
                                  // bytes from the segment's start to the text's start << 13 | id
    uint8_t lut[256];             // byte value -> code 0..3, 4 = delimiter, 0x80 = not in this block's alphabet
    uint32_t wsum[kPackThreads / kWave];
    uint64_t ball[4];
    uint32_t cb[4];               // code -> byte value
    uint32_t blk, ctr, ndelim, drow, bad;
    uint32_t bucket_fill[kPackBuckets], bucket_base[kPackBuckets + 1];  // the order of walk 2
};
static_assert(sizeof(PackShared) <= 160u * 1024u, "one workgroup's LDS");
struct __attribute__((packed, aligned(1))) PackBytes16 {
    uint32_t x, y, z, w;
};
struct __attribute__((packed, aligned(1))) PackBytes4 {
    uint32_t x;
};
struct __attribute__((packed, aligned(4))) PackWords2 {
    uint32_t x, y;
};

__device__ __forceinline__ uint32_t pack_seg(uint32_t bytes, uint32_t next) { return (bytes << kPackNextBits) | next; }

#ifndef BWT_PACK_GRAB
#define BWT_PACK_GRAB 12
#endif
constexpr uint32_t kPackGrabBatch = BWT_PACK_GRAB;   // lanes of a wave that wait for a segment before the wave hands ids out
constexpr uint32_t kPackIdChunk = 64;    // ids a wave takes from the workgroup's counter at a time
constexpr uint32_t kPackLong = 48;       // walk 2 hands out the segments of at least this many bytes first

// Both walks.  EMIT = false: lengths and links into sh.seg.  EMIT = true: the bytes go to
// out_end[-(total - togo) - 1] downwards, togo from sh.seg.
// A lane's state is the PACKED row q it stands on (text rows only: q = row - (row > drow)).  The loop
// is flat on purpose -- every iteration is one step of every walking lane, nothing inside the step
// branches -- and lanes whose segment has ended wait until kPackGrabBatch of the wave's lanes do, so
// that the bookkeeping (the record or the last stores of the finished segment, the next id, its first
// row) is paid once for several lanes; ids come from a range of kPackIdChunk that the wave owns.
// Walk 2 takes the segments in the order the order step has laid out in sh.seg (`count` of them): the
// text stretch by stretch, so that what the workgroup's lanes write at any time lies within a few
// tens of KB and the dword stores of neighbouring segments meet in the L2 (handed out by id, they were
// scattered over the whole block and every line left the L2 in pieces: nine times the text in HBM
// writes); within a stretch the long segments first -- the workgroup is done when its last lane is,
// and a long segment started late would be that lane.
// Returns the iterations the wave has made.
template <bool EMIT>
__device__ __forceinline__ uint32_t packed_walk(PackShared& sh, uint32_t m, uint32_t drow, uint32_t row0, uint32_t mul,
                                                uint32_t inv, uint32_t mask, uint32_t ids, uint32_t cbytes,
                                                uint8_t* out_end, uint32_t total, uint32_t count) {
    const uint8_t* const planes = reinterpret_cast<const uint8_t*>(sh.planes);
    const uint8_t* const samp = reinterpret_cast<const uint8_t*>(sh.samp);
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    uint32_t id = kNone;  // the segment the lane walks, or has walked and not yet put on record
    uint32_t q = 0, len = 0, hit_next = 0;
    uint32_t walking = 0;  // lane mask: 0 or all ones
    // walk 2: bytes not yet stored (oldest in the highest bits), how many, one past the address of the oldest
    uint32_t acc_lo = 0, acc_hi = 0, cnt = 0;
    uint8_t* top = out_end;
    uint32_t wnext = 0, wend = 0;  // the wave's ids (wave-uniform)
    bool more = true;              // the workgroup's counter may have ids left (wave-uniform)
    const uint32_t id_end = EMIT ? count : ids;
    // No walk is longer than m steps and a lane that waits is served within a few iterations; the bound
    // only keeps a workgroup from spinning should its tables ever be inconsistent (the block is then
    // left to the general kernels).
    const uint32_t it_cap = 4u * m + 65536u;
    if (threadIdx.x == 0) {  // the chain's first segment starts where the delimiter row leads
        const uint32_t q0 = row0 - (row0 > drow ? 1u : 0u);
        const uint32_t img = __umul24(q0, mul) & mask;
        if (row0 == drow || img < ids) {  // ... which is a cut itself: an empty segment
            if (!EMIT) sh.seg[kPackIds] = pack_seg(0u, row0 == drow ? kPackEnd : img);
        } else {
            walking = ~0u;
            id = kPackIds;
            q = q0;
        }
    }
    auto store_dword = [&]() {  // the four oldest bytes
        const uint32_t sh8 = (cnt - 4u) * 8u;
        const uint64_t acc = ((uint64_t)acc_hi << 32) | acc_lo;
        reinterpret_cast<PackBytes4*>(top - 4)->x = (uint32_t)(acc >> sh8);
        top -= 4;
        cnt -= 4u;
    };
    for (uint32_t it = 0;; ++it) {
        const uint64_t idle = __ballot(walking == 0u);
        const uint32_t nidle = (uint32_t)__popcll(idle);
        if (nidle == (uint32_t)kWave || (more && nidle >= kPackGrabBatch)) {  // wave-uniform
            if (walking == 0u && id != kNone) {  // a finished segment
                if (EMIT) {
                    if (cnt >= 4u) store_dword();
                    for (uint32_t k = 0; k < cnt; ++k) top[-1 - (int)k] = (uint8_t)(acc_lo >> (8u * (cnt - 1u - k)));
                    cnt = 0;
                } else {
                    sh.seg[id] = pack_seg(len, hit_next);
                }
                id = kNone;
            }
            if (!more) return it;  // nobody walks, nothing left
            const uint32_t avail = wend - wnext;
            const uint32_t rank = (uint32_t)__popcll(idle & lanemask_lt());
            if (walking == 0u && rank < avail) {
                uint32_t cand = wnext + rank;
                bool ok = true;
                if (EMIT) {  // the cand-th segment of the order step's list
                    const uint32_t word = sh.seg[cand];
                    const uint32_t togo = word >> kPackNextBits;
                    ok = togo != 0u && togo <= total;
                    top = out_end - (total - togo);
                    cand = word & kPackEnd;
                }
                const uint32_t q1 = __umul24(cand, inv) & mask;
                ok = ok && q1 < m;
                if (ok) {
                    walking = ~0u;
                    id = cand;
                    q = q1;
                    len = 0;
                }
            }
            wnext += min(nidle, avail);
            if (wnext == wend) {  // a new range for the wave
                uint32_t v = 0;
                if (lane_id() == 0) v = atomicAdd(&sh.ctr, kPackIdChunk);
                wnext = wave_read(v, 0);
                wend = min(wnext + kPackIdChunk, id_end);
                if (wnext >= id_end) {
                    more = false;
                    wnext = wend = 0;
                }
            }
        }
        if (it > it_cap) {  // wave-uniform; never
            sh.bad = 1;
            return it;
        }
        if (walking) {
            const uint32_t off = (q >> 6) * 16u;
            const uint4 pl = *reinterpret_cast<const uint4*>(planes + off);
            const uint32_t up = 63u - (q & 63u);  // my row's bit goes to the top, the rows above me below it
            const uint64_t lo = (((uint64_t)pl.y << 32) | pl.x) << up, hi = (((uint64_t)pl.w << 32) | pl.z) << up;
            const uint32_t s0 = (uint32_t)((int32_t)(uint32_t)(lo >> 32) >> 31);  // all ones when the code's bit is set
            const uint32_t s1 = (uint32_t)((int32_t)(uint32_t)(hi >> 32) >> 31);
            // one word, read once the symbol is known: fetching all four beside the planes and choosing in
            // registers was slower (the loop is bound by LDS traffic)
            const uint32_t next_row = *reinterpret_cast<const uint32_t*>(samp + off + ((s0 & 4u) | (s1 & 8u)));
            // rows with my symbol: me, the rows above me, and -- for symbol 0 -- the `up` zeros shifted in
            const uint32_t same_lo = ~(((uint32_t)lo ^ s0) | ((uint32_t)hi ^ s1));
            const uint32_t same_hi = ~(((uint32_t)(lo >> 32) ^ s0) | ((uint32_t)(hi >> 32) ^ s1));
            const uint32_t extra = 1u + (up & ~(s0 | s1));
            const uint32_t row = next_row + (uint32_t)__popc(same_lo) + (uint32_t)__popc(same_hi) - extra;
            ++len;
            if (EMIT) {
                const uint32_t byte = (cbytes >> ((s0 & 8u) | (s1 & 16u))) & 255u;
                acc_hi = __builtin_amdgcn_alignbit(acc_hi, acc_lo, 24);
                acc_lo = (acc_lo << 8) | byte;
                ++cnt;
            }
            q = row - (row > drow ? 1u : 0u);
            const uint32_t img = __umul24(q, mul) & mask;
            const bool at_end = row == drow, cut = img < ids;
            hit_next = at_end ? kPackEnd : img;
            walking = (at_end | cut) ? 0u : ~0u;
        }
        if (EMIT && (it & 3u) == 3u && cnt >= 4u) store_dword();  // (it & 3) is wave-uniform: a lane holds at most 7 bytes
    }
}

// ---- one walk instead of two --------------------------------------------------------------------------
// Walk 2 repeats every step of walk 1 only because a byte's place in the text is not known before the
// lengths of all segments are.  With BWT_PACK_STREAM the chain is walked ONCE: a lane appends the bytes of
// the segments it walks, one after the other and back to front like the text itself, to a stream of its own
// in device memory (the block's slice of the LF-table area, which nothing else uses while the block is here),
// and notes per segment where its bytes end up.  The order step then also ranks the segments along the
// chain, which gives a table sorted by text position, and a copy pass -- one lane per 16 bytes of text,
// stores coalesced -- looks every piece up and fetches it from the streams.
#ifndef BWT_PACK_STREAM
#define BWT_PACK_STREAM 1
#endif
constexpr uint32_t kPackOffBytes = 32768u;  // head of the block's area: one word per segment id (where its bytes end), then the streams
static_assert((kPackIds + 1u) * 4u <= kPackOffBytes, "a word per segment");

// What the LDS of the rank structure holds once the walk is over.
struct PackPost {
    uint32_t cnt[kPackIds + 2];   // order: segments from this one to the chain's end; then by rank: 16-byte pieces of the segments before
    uint32_t pos[kPackIds + 2];   // by rank along the text: first text position of the segment; [segments] = the text's length
    uint32_t addr[kPackIds + 1];  // by rank: code address of the end of the segment's codes
    uint16_t owner[kPackMaxRows / 16u + kPackIds + 8u];  // by piece: the rank of the segment it belongs to
};
static_assert(sizeof(PackPost) <= 2u * sizeof(uint4) * kPackGroups + sizeof(uint32_t) * (kPackIds + 1u), "fits where planes, samples and records were");
static_assert(offsetof(PackShared, seg) == offsetof(PackShared, planes) + 2u * sizeof(uint4) * kPackGroups, "... which are one stretch");
static_assert(offsetof(PackShared, samp) == offsetof(PackShared, planes) + sizeof(uint4) * kPackGroups, "planes and samples are one stretch");

// The single walk: as packed_walk<false> (lengths and links into sh.seg), and every step's two-bit CODE goes
// to the lane's stream: area + kPackOffBytes + lane * lane_bytes .. + lane_bytes, filled from its top downwards,
// sixteen codes to a dword, the oldest in the highest bits -- so that, counting codes from the area's start
// (code address c = bits 2 (c % 16) .. of dword c / 16), a segment's codes lie in text order.
// segoff[id] = code address of the end of segment id's codes.
__device__ __forceinline__ uint32_t packed_walk_stream(PackShared& sh, uint32_t m, uint32_t drow, uint32_t row0, uint32_t mul,
                                                       uint32_t inv, uint32_t mask, uint32_t ids, uint8_t* area,
                                                       uint32_t* segoff, uint32_t lane_bytes) {
    const uint8_t* const planes = reinterpret_cast<const uint8_t*>(sh.planes);
    const uint8_t* const samp = reinterpret_cast<const uint8_t*>(sh.samp);
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    uint32_t id = kNone;
    uint32_t q = 0, first_ca = 0, hit_next = 0;  // first_ca: where the segment's codes end (its first code's address + 1)
    uint32_t walking = 0;
    uint32_t acc_lo = 0, acc_hi = 0, cnt = 0;  // codes not yet stored (the oldest in the highest bits), how many (at most 31)
    const uint32_t lane_lo = kPackOffBytes + (uint32_t)threadIdx.x * lane_bytes;  // byte offsets in the area
    uint32_t top = lane_lo + lane_bytes;       // one past the dword that holds the oldest code stored last
    uint32_t wnext = 0, wend = 0;
    bool more = true;
    const uint32_t it_cap = 4u * m + 65536u;
    if (threadIdx.x == 0) {  // the chain's first segment starts where the delimiter row leads
        const uint32_t q0 = row0 - (row0 > drow ? 1u : 0u);
        const uint32_t img = __umul24(q0, mul) & mask;
        segoff[kPackIds] = first_ca = top * 4u;
        if (row0 == drow || img < ids) {
            sh.seg[kPackIds] = pack_seg(0u, row0 == drow ? kPackEnd : img);
        } else {
            walking = ~0u;
            id = kPackIds;
            q = q0;
        }
    }
    auto store_dword = [&](uint32_t word) {
        if (top < lane_lo + 4u) {  // the lane's stream is full: the block is left to the general kernels
            sh.bad = 1;
            top = lane_lo + lane_bytes;
        }
        top -= 4u;
        *reinterpret_cast<uint32_t*>(area + top) = word;
    };
    for (uint32_t it = 0;;) {
#pragma unroll 1
      for (uint32_t k = 0; k < 16u; ++k, ++it) {  // sixteen steps, then the lanes that have sixteen codes store them
        const uint64_t idle = __ballot(walking == 0u);
        const uint32_t nidle = (uint32_t)__popcll(idle);
        if (nidle == (uint32_t)kWave || (more && nidle >= kPackGrabBatch)) {  // wave-uniform
            if (walking == 0u && id != kNone) {  // a finished segment
                sh.seg[id] = pack_seg(first_ca - (top * 4u - cnt), hit_next);  // (a code per step)
                id = kNone;
            }
            if (!more) {  // nobody walks, nothing left: the codes still in registers
                const uint64_t acc = ((uint64_t)acc_hi << 32) | acc_lo;
                if (cnt >= 16u) {
                    store_dword((uint32_t)(acc >> (2u * (cnt - 16u))));
                    cnt -= 16u;
                }
                if (cnt) store_dword(acc_lo << (32u - 2u * cnt));  // the rest in the dword's high bits, where they continue the row
                return it;
            }
            const uint32_t avail = wend - wnext;
            const uint32_t rank = (uint32_t)__popcll(idle & lanemask_lt());
            if (walking == 0u && rank < avail) {
                const uint32_t cand = wnext + rank;
                const uint32_t q1 = __umul24(cand, inv) & mask;
                if (q1 < m) {
                    walking = ~0u;
                    id = cand;
                    q = q1;
                    segoff[cand] = first_ca = top * 4u - cnt;
                }
            }
            wnext += min(nidle, avail);
            if (wnext == wend) {  // a new range for the wave
                uint32_t v = 0;
                if (lane_id() == 0) v = atomicAdd(&sh.ctr, kPackIdChunk);
                wnext = wave_read(v, 0);
                wend = min(wnext + kPackIdChunk, ids);
                if (wnext >= ids) {
                    more = false;
                    wnext = wend = 0;
                }
            }
        }
        if (walking) {
            const uint32_t off = (q >> 6) * 16u;
            const uint4 pl = *reinterpret_cast<const uint4*>(planes + off);
            const uint32_t up = 63u - (q & 63u);
            const uint64_t lo = (((uint64_t)pl.y << 32) | pl.x) << up, hi = (((uint64_t)pl.w << 32) | pl.z) << up;
            const uint32_t s0 = (uint32_t)((int32_t)(uint32_t)(lo >> 32) >> 31);
            const uint32_t s1 = (uint32_t)((int32_t)(uint32_t)(hi >> 32) >> 31);
            const uint32_t next_row = *reinterpret_cast<const uint32_t*>(samp + off + ((s0 & 4u) | (s1 & 8u)));
            const uint32_t same_lo = ~(((uint32_t)lo ^ s0) | ((uint32_t)hi ^ s1));
            const uint32_t same_hi = ~(((uint32_t)(lo >> 32) ^ s0) | ((uint32_t)(hi >> 32) ^ s1));
            const uint32_t extra = 1u + (up & ~(s0 | s1));
            const uint32_t row = next_row + (uint32_t)__popc(same_lo) + (uint32_t)__popc(same_hi) - extra;
            acc_hi = __builtin_amdgcn_alignbit(acc_hi, acc_lo, 30);
            acc_lo = (acc_lo << 2) | (s0 & 1u) | (s1 & 2u);
            ++cnt;
            q = row - (row > drow ? 1u : 0u);
            const uint32_t img = __umul24(q, mul) & mask;
            const bool at_end = row == drow, cut = img < ids;
            hit_next = at_end ? kPackEnd : img;
            walking = (at_end | cut) ? 0u : ~0u;
        }
      }
        if (it > it_cap) {  // wave-uniform; never
            sh.bad = 1;
            return it;
        }
        if (cnt >= 16u) {  // (a lane holds at most 31 codes)
            const uint64_t acc = ((uint64_t)acc_hi << 32) | acc_lo;
            store_dword((uint32_t)(acc >> (2u * (cnt - 16u))));
            cnt -= 16u;
        }
    }
}

// -DBWT_PACK_CLOCKS: thread 0 of the workgroup that gets block 0 prints what each step took (s_memtime ticks)
#ifdef BWT_PACK_CLOCKS
#define PACK_CLK(p)                                            \
    do {                                                       \
        __syncthreads();                                       \
        const uint64_t now_ = __builtin_amdgcn_s_memtime();    \
        clk_[p] = (uint32_t)(now_ - last_);                    \
        last_ = now_;                                          \
    } while (0)
#else
#define PACK_CLK(p)
#endif

// Which blocks are worth a try: one wave per block looks at 64 of the block's bytes and lists the
// block (queue[2] counts, the list is in a.drow, which the table kernels fill only later) when they
// hold at most four byte values besides the delimiter.  Random bytes and text are ruled out here
// for the price of one cache line; every block's mode is set to "not done".
constexpr int kPackClassifyThreads = 1024;
__global__ __launch_bounds__(kPackClassifyThreads) void bwt_inverse_packed_classify_kernel(InvArgs a) {
    constexpr int W = kPackClassifyThreads / kWave;
    __shared__ uint32_t s_listed[W];  // block index, or kNoRow
    __shared__ uint32_t s_base;
    const int lane = lane_id(), w = wave_id();
    const uint32_t b = blockIdx.x * (uint32_t)W + (uint32_t)w;
    bool listed = false;
    if (b < a.nblocks) {  // wave-uniform
        if (lane == 0) a.mode[b] = kModeNone;
        const uint32_t n = inv_block_len(a, b);
        if (n >= kPackMinRows && n <= kPackMaxRows) {
            // 64 bytes spread over the block: a transform keeps equal symbols together (its first rows may well
            // hold a single value whatever the text was), so a prefix would say little
            const uint32_t byte = a.in[(uint64_t)b * ((uint64_t)a.block_size + 1u) + (uint32_t)lane * (n / (uint32_t)kWave)];
            uint64_t rest = __ballot(byte != a.delim);
            uint32_t values = 0;
            while (rest && values <= 4u) {
                const uint32_t v = wave_read(byte, (int)__builtin_ctzll(rest));
                rest &= ~__ballot(byte == v);
                ++values;
            }
            listed = rest == 0ull && values >= 1u && values <= 4u;
        }
    }
    if (lane == 0) s_listed[w] = listed ? b : kNoRow;
    __syncthreads();
    // one counter update per workgroup (one per block was 16 K updates of a single word: 0.19 ms)
    if (threadIdx.x == 0) {
        uint32_t count = 0;
        for (int i = 0; i < W; ++i) count += s_listed[i] != kNoRow ? 1u : 0u;
        s_base = count ? atomicAdd(&a.queue[2], count) : 0u;
    }
    __syncthreads();
    if (lane == 0 && listed) {
        uint32_t at = s_base;
        for (int i = 0; i < w; ++i) at += s_listed[i] != kNoRow ? 1u : 0u;
        a.drow[at] = b;
    }
}

__global__ __launch_bounds__(kPackThreads) void bwt_inverse_packed_kernel(InvArgs a) {
    __shared__ PackShared sh;
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
#ifdef BWT_PACK_CLOCKS
    uint32_t clk_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t last_ = 0;
#endif
    for (;;) {
        __syncthreads();  // everybody is done with the block before
        if (tid == 0) {
            const uint32_t i = atomicAdd(&a.queue[1], 1u);  // the next listed block
            sh.blk = i < a.queue[2] ? a.drow[i] : a.nblocks;
            sh.ndelim = 0;
            sh.bad = 0;
            sh.drow = kNoRow;
        }
        if (tid < 256) sh.lut[tid] = 0;
        if (tid < 4) sh.cb[tid] = 0;
        if (tid < (int)kPackBuckets) sh.bucket_fill[tid] = 0;
        __syncthreads();
        const uint32_t b = sh.blk;
        if (b >= a.nblocks) break;
        const uint32_t n = inv_block_len(a, b);  // kPackMinRows .. kPackMaxRows: listed
        const uint8_t* __restrict__ L = a.in + (uint64_t)b * ((uint64_t)a.block_size + 1u);
#ifdef BWT_PACK_CLOCKS
        last_ = __builtin_amdgcn_s_memtime();
#endif

        // ---- the alphabet, from 256 bytes spread over the block (a symbol they miss shows up in the pack
        // step, which then leaves the block to the general kernels)
        if (tid < 256) sh.lut[L[(uint32_t)tid * (n / 256u)]] = 1;  // n >= kPackMinRows
        __syncthreads();
        const bool present = tid < 256 && sh.lut[tid] != 0 && (uint32_t)tid != a.delim;
        const uint64_t bal = __ballot(present);
        if (w < 4 && lane == 0) sh.ball[w] = bal;
        __syncthreads();
        uint32_t sigma = 0, below = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t c = (uint32_t)__popcll(sh.ball[i]);
            sigma += c;
            if (i < w) below += c;
        }
        if (sigma > 4u || sigma == 0u) continue;  // workgroup-uniform
        if (tid < 256) {
            const uint32_t code = below + (uint32_t)__popcll(bal & lanemask_lt());
            sh.lut[tid] = present ? (uint8_t)code : (uint32_t)tid == a.delim ? (uint8_t)4 : (uint8_t)0x80;
            if (present) sh.cb[code] = (uint32_t)tid;
        }
        __syncthreads();
        const uint32_t cbytes = sh.cb[0] | (sh.cb[1] << 8) | (sh.cb[2] << 16) | (sh.cb[3] << 24);

        // Four rows at a time where the alphabet allows it: if some two bits of the byte values (bits fs, fs + 1) tell
        // the block's symbols apart -- A C G T: bits 1 and 2 --, the four fields of a dword of rows index a table of 256
        // words: the four codes as a byte-permute selector (bits 0-1 of every byte), the codes' low bits (bits 4-7) and
        // high bits (bits 12-15).  The selector turns the codes back into bytes; a dword that does not come back as it
        // was holds the delimiter or a byte outside the alphabet, and its rows go the byte-by-byte way below.
        uint32_t fs = 8;  // no such field
        for (uint32_t f = 0; f < 7u && fs == 8u; ++f) {
            uint32_t seen = 0;
            for (uint32_t i = 0; i < sigma; ++i) seen |= 1u << (((cbytes >> (8u * i)) >> f) & 3u);
            if ((uint32_t)__popc(seen) == sigma) fs = f;
        }
        uint32_t* const quad = sh.seg;  // (the segment records are set up after this step)
        if (fs < 8u && tid < 256) {
            uint32_t word = 0;
            for (uint32_t k = 0; k < 4u; ++k) {
                const uint32_t field = ((uint32_t)tid >> (2u * k)) & 3u;
                uint32_t code = 0;
                for (uint32_t i = 0; i < sigma; ++i)
                    if ((((cbytes >> (8u * i)) >> fs) & 3u) == field) code = i;
                word |= (code << (8u * k)) | ((code & 1u) << (4u + k)) | ((code >> 1) << (12u + k));
            }
            quad[tid] = word;
        }
        __syncthreads();
        PACK_CLK(0);
        // ---- pack: 16 rows per lane and step, the delimiter row still in (code 0)
        const uint32_t groups = (n + 63u) / 64u;
        for (uint32_t p = (uint32_t)tid; p < 4u * groups; p += kPackThreads) {
            const uint32_t r0 = 16u * p;
            uint32_t lo = 0, hi = 0, fl = 0;
            if (r0 < n) {
                uint32_t d[4];
                if (r0 + 16u <= n) {
                    const PackBytes16 v = *reinterpret_cast<const PackBytes16*>(L + r0);
                    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                } else {  // the block's ragged end: padded with a text symbol
                    d[0] = d[1] = d[2] = d[3] = (cbytes & 255u) * 0x01010101u;
                    for (uint32_t k = 0; r0 + k < n; ++k)
                        d[k >> 2] = (d[k >> 2] & ~(255u << (8u * (k & 3u)))) | ((uint32_t)L[r0 + k] << (8u * (k & 3u)));
                }
                bool by_byte = true;
                if (fs < 8u) {
                    uint32_t e[4], diff = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t t = (d[j] >> fs) & 0x03030303u;
                        const uint32_t y = t | (t >> 6);
                        e[j] = quad[(y | (y >> 12)) & 255u];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        diff |= __builtin_amdgcn_perm(0u, cbytes, e[j] & 0x03030303u) ^ d[j];
                        lo |= ((e[j] >> 4) & 15u) << (4 * j);
                        hi |= ((e[j] >> 12) & 15u) << (4 * j);
                    }
                    by_byte = diff != 0u;
                }
                if (by_byte) {
                    lo = hi = 0;
#pragma unroll
                    for (uint32_t k = 0; k < 16u; ++k) {
                        const uint32_t c = sh.lut[(d[k >> 2] >> (8u * (k & 3u))) & 255u];
                        lo |= (c & 1u) << k;
                        hi |= ((c >> 1) & 1u) << k;
                        fl |= c;
                    }
                }
                if (fl & 0x84u) {  // rare: a delimiter, or a byte that is not in the alphabet
                    if (fl & 0x80u) sh.bad = 1;
                    if (fl & 4u)
                        for (uint32_t k = 0; k < 16u; ++k)
                            if (r0 + k < n && ((d[k >> 2] >> (8u * (k & 3u))) & 255u) == a.delim) {
                                atomicAdd(&sh.ndelim, 1u);
                                sh.drow = r0 + k;
                            }
                }
            }
            uint16_t* planes = reinterpret_cast<uint16_t*>(&sh.planes[p >> 2]);
            planes[p & 3u] = (uint16_t)lo;
            planes[4u + (p & 3u)] = (uint16_t)hi;
        }
        __syncthreads();
        if (sh.bad || sh.ndelim != 1u) continue;  // workgroup-uniform: left to the general kernels
        const uint32_t drow = sh.drow;
        PACK_CLK(1);

        // ---- take the delimiter row out: the rows above it move down by one
        {
            constexpr int K = (kPackGroups + kPackThreads - 1) / kPackThreads;
            const uint32_t gd = drow >> 6, rd = drow & 63u;
            uint64_t nlo[K], nhi[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t g = (uint32_t)tid + (uint32_t)k * kPackThreads;
                nlo[k] = nhi[k] = 0;
                if (g >= gd && g < groups) {
                    const uint4 c = sh.planes[g];
                    uint4 x = make_uint4(0u, 0u, 0u, 0u);
                    if (g + 1u < groups) x = sh.planes[g + 1u];
                    const uint64_t clo = ((uint64_t)c.y << 32) | c.x, chi = ((uint64_t)c.w << 32) | c.z;
                    const uint64_t keep = g == gd ? (1ull << rd) - 1ull : 0ull;  // rows of the group below the delimiter row stay
                    nlo[k] = (clo & keep) | ((clo >> 1) & ~keep) | ((uint64_t)(x.x & 1u) << 63);
                    nhi[k] = (chi & keep) | ((chi >> 1) & ~keep) | ((uint64_t)(x.z & 1u) << 63);
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t g = (uint32_t)tid + (uint32_t)k * kPackThreads;
                if (g >= gd && g < groups)
                    sh.planes[g] = make_uint4((uint32_t)nlo[k], (uint32_t)(nlo[k] >> 32), (uint32_t)nhi[k], (uint32_t)(nhi[k] >> 32));
            }
            __syncthreads();
        }

        PACK_CLK(2);
        // ---- counts per group -> samples.  Thread t owns K consecutive groups of the m = n - 1 text rows.
        const uint32_t m = n - 1u;
        const uint32_t tgroups = (m + 63u) / 64u;
        uint32_t row0;  // where the delimiter row leads: the delimiter's own place in the first column
        {
            constexpr int K = 4;  // tgroups <= 4096
            const uint32_t per = (tgroups + kPackThreads - 1u) / kPackThreads;  // 1..4
            uint32_t pre[K][4];
            uint32_t run[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t g = (uint32_t)tid * per + (uint32_t)k;
#pragma unroll
                for (int s = 0; s < 4; ++s) pre[k][s] = run[s];
                if ((uint32_t)k < per && g < tgroups) {
                    const uint4 c = sh.planes[g];
                    const uint32_t valid = min(64u, m - 64u * g);
                    const uint64_t vm = valid == 64u ? ~0ull : (1ull << valid) - 1ull;
                    const uint64_t clo = (((uint64_t)c.y << 32) | c.x), chi = (((uint64_t)c.w << 32) | c.z);
                    const uint32_t c1 = (uint32_t)__popcll(clo & ~chi & vm), c2 = (uint32_t)__popcll(~clo & chi & vm),
                                   c3 = (uint32_t)__popcll(clo & chi & vm);
                    run[0] += valid - c1 - c2 - c3;
                    run[1] += c1;
                    run[2] += c2;
                    run[3] += c3;
                }
            }
            uint32_t base[4], tot[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) base[s] = block_exclusive_scan<kPackThreads>(run[s], sh.wsum, tot[s]);
            // C[]: text symbols below, plus the delimiter when its value is below (reference src/inverse_bwt.cpp:35-45)
            uint32_t C[4];
            uint32_t acc = 0;
            row0 = 0;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bool real = (uint32_t)s < sigma;
                const bool dbelow = real && a.delim < ((cbytes >> (8 * s)) & 255u);
                C[s] = acc + (dbelow ? 1u : 0u);
                if (real && !dbelow) row0 += tot[s];
                acc += tot[s];
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t g = (uint32_t)tid * per + (uint32_t)k;
                if ((uint32_t)k < per && g < tgroups)
                    sh.samp[g] = make_uint4(C[0] + base[0] + pre[k][0], C[1] + base[1] + pre[k][1],
                                                C[2] + base[2] + pre[k][2], C[3] + base[3] + pre[k][3]);
            }
        }
        PACK_CLK(3);
        // ---- cuts: packed row q -> q * mul mod 2^kbits (2^kbits >= m), a cut where the image is below `ids`
        const uint32_t kbits = (uint32_t)bit_length(m - 1u);
        const uint32_t mask = (1u << kbits) - 1u;
        const uint32_t mul = ((uint32_t)(0.6180339887498949 * (double)(1u << kbits)) | 1u) & mask;
        uint32_t inv = mul;  // Newton: inverse of an odd number modulo 2^32, five steps from 3 correct bits
#pragma unroll
        for (int i = 0; i < 5; ++i) inv *= 2u - mul * inv;
        inv &= mask;
        const uint32_t ids = min(kPackIds, (1u << kbits) >> 3);
        for (uint32_t i = (uint32_t)tid; i <= kPackIds; i += kPackThreads) sh.seg[i] = pack_seg(0u, kPackEnd);
        if (tid == 0) sh.ctr = 0;
        __syncthreads();
        uint8_t* const out_end = a.out + (uint64_t)b * a.out_stride + m;  // one past the text's last byte

#if BWT_PACK_STREAM
        {
            uint8_t* const area = reinterpret_cast<uint8_t*>(a.T) + (uint64_t)b * a.tstride * 4u;
            uint32_t* const segoff = reinterpret_cast<uint32_t*>(area);
            const uint32_t lane_bytes = ((a.tstride * 4u - kPackOffBytes - 64u) / (uint32_t)kPackThreads) & ~3u;
            const uint32_t its = packed_walk_stream(sh, m, drow, row0, mul, inv, mask, ids, area, segoff, lane_bytes);
            __syncthreads();
            PACK_CLK(4);
            if (sh.bad) continue;  // workgroup-uniform: a stream ran full (or never: inconsistent tables)
            PackPost& post = *reinterpret_cast<PackPost*>(sh.planes);  // the rank structure has done its work

            // ---- order: every record becomes bytes, and segments, from its segment's start to the text's start.
            // Records of 8 bytes {bytes << 13 | next, segments} over cnt[] and pos[]; a thread keeps its own ones in registers.
            constexpr int E = (kPackIds + 1 + kPackThreads - 1) / kPackThreads;
            uint2* const rec = reinterpret_cast<uint2*>(&post);
            static_assert(sizeof(uint2) * (kPackIds + 1u) <= offsetof(PackPost, addr), "the records end before addr[]");
            uint2 mine[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint32_t i = (uint32_t)tid + (uint32_t)e * kPackThreads;
                mine[e] = make_uint2(i <= kPackIds ? sh.seg[i] : pack_seg(0u, kPackEnd), 1u);
                if (i <= kPackIds) rec[i] = mine[e];
            }
            __syncthreads();
            for (int round = 0; round < (int)kPackNextBits; ++round) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const uint32_t next = mine[e].x & kPackEnd;
                    if (next != kPackEnd) {
                        const uint2 w2 = rec[next];
                        mine[e] = make_uint2(pack_seg((mine[e].x >> kPackNextBits) + (w2.x >> kPackNextBits), w2.x & kPackEnd), mine[e].y + w2.y);
                    }
                }
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const uint32_t i = (uint32_t)tid + (uint32_t)e * kPackThreads;
                    if (i <= kPackIds) rec[i] = mine[e];
                }
                __syncthreads();
            }
            const uint2 first = rec[kPackIds];
            const uint32_t total = first.x >> kPackNextBits, nseg = first.y;
            if ((first.x & kPackEnd) != kPackEnd || total > m || total == 0u || nseg > kPackIds + 1u) continue;  // never for a permutation
            __syncthreads();  // everybody has read the first record: the tables go where the records are
            // The segments on the chain by their rank along the text (the chain's last segment holds the text's start):
            // the segment of rank r covers text positions pos[r] .. pos[r + 1], its codes end at code address addr[r].
            uint32_t placed = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint32_t i = (uint32_t)tid + (uint32_t)e * kPackThreads;
                const uint32_t togo = mine[e].x >> kPackNextBits;
                // (a record nobody walked still reads "nothing, chain's end": togo 0)
                if (i <= kPackIds && (mine[e].x & kPackEnd) == kPackEnd && togo != 0u && togo <= total && mine[e].y <= nseg) {
                    const uint32_t r = mine[e].y - 1u;
                    post.pos[r + 1u] = togo;
                    post.addr[r] = segoff[i];
                    ++placed;
                }
            }
            if (tid == 0) {
                post.pos[0] = 0;
                sh.ctr = 0;
            }
            __syncthreads();
            if (placed) atomicAdd(&sh.ctr, placed);
            __syncthreads();
            if (sh.ctr != nseg) continue;  // workgroup-uniform; never for a permutation: ranks would be missing from the table
            PACK_CLK(5);

            // ---- pieces: a segment of L bytes is copied as ceil(L / 16) pieces of 16 bytes (the last one moved back so
            // that it ends with the segment; a segment shorter than 16 bytes is one piece, stored in parts)
            {
                constexpr int R = (kPackIds + 1 + kPackThreads - 1) / kPackThreads;
                uint32_t np[R], sum = 0;
#pragma unroll
                for (int e = 0; e < R; ++e) {
                    const uint32_t r = (uint32_t)tid * R + (uint32_t)e;
                    np[e] = r < nseg ? (post.pos[r + 1u] - post.pos[r] + 15u) >> 4 : 0u;
                    sum += np[e];
                }
                uint32_t allp;
                uint32_t run = block_exclusive_scan<kPackThreads>(sum, sh.wsum, allp);
#pragma unroll
                // (owner[] reaches into the records' LDS, which the table above was the last to read: all threads are past it)
                for (int e = 0; e < R; ++e) {
                    const uint32_t r = (uint32_t)tid * R + (uint32_t)e;
                    if (r <= nseg) post.cnt[r] = run;
                    for (uint32_t j = 0; j < np[e]; ++j) post.owner[run + j] = (uint16_t)r;
                    run += np[e];
                }
                __syncthreads();
#ifdef BWT_PACK_CLOCKS
                PACK_CLK(7);
#endif
                // ---- copy: lane per piece; neighbouring lanes write neighbouring stretches of the text
                uint8_t* const text = out_end - total;  // (a hand-made block may spell less than m bytes: its text ends where a full one would)
                const uint32_t* const words = reinterpret_cast<const uint32_t*>(area);
                // All loads of a batch of pieces first, then all its stores: loads and stores share one in-order counter
                // of outstanding operations here, so a wait for a load issued after a store also sits out that store's
                // round trip -- once per batch this way, not once per piece.
                constexpr int U = 12;  // pieces of a batch per lane (a block has at most 24 per lane)
                for (uint32_t base = 0; base < allp; base += U * kPackThreads) {
                    uint32_t w0[U], w1[U], sh2[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const uint32_t pc = min(base + (uint32_t)u * kPackThreads + (uint32_t)tid, allp - 1u);
                        const uint32_t lo = post.owner[pc];
                        const uint32_t L = post.pos[lo + 1u] - post.pos[lo];
                        uint32_t start = 16u * (pc - post.cnt[lo]);
                        if (L >= 16u) start = min(start, L - 16u);
                        const uint32_t ca = post.addr[lo] - L + start;  // code address of the piece's first code
                        const PackWords2 ww = *reinterpret_cast<const PackWords2*>(words + (ca >> 4));  // one request for both words
                        w0[u] = ww.x;
                        w1[u] = ww.y;
                        sh2[u] = (ca & 15u) * 2u;
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const uint32_t p = base + (uint32_t)u * kPackThreads + (uint32_t)tid;
                        if (p >= allp) break;
                        const uint32_t lo = post.owner[p];
                        const uint32_t P = post.pos[lo], L = post.pos[lo + 1u] - P;
                        uint32_t start = 16u * (p - post.cnt[lo]);
                        if (L >= 16u) start = min(start, L - 16u);
                        uint8_t* const dst = text + P + start;
                        const uint32_t codes = __builtin_amdgcn_alignbit(w1[u], w0[u], sh2[u]);
                        uint32_t d[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {  // four codes -> the selector of a byte permute over the four byte values
                            const uint32_t c = (codes >> (8 * k)) & 255u;
                            const uint32_t sel = (c & 3u) | ((c & 0xCu) << 6) | ((c & 0x30u) << 12) | ((c & 0xC0u) << 18);
                            d[k] = __builtin_amdgcn_perm(0u, cbytes, sel);
                        }
                        if (L >= 16u) {
                            PackBytes16 o;
                            o.x = d[0]; o.y = d[1]; o.z = d[2]; o.w = d[3];
                            *reinterpret_cast<PackBytes16*>(dst) = o;
                        } else {
                            const uint32_t a8 = L & 8u, a4 = L & 4u;  // (no indexing of d[] by a run-time value: it would live in scratch)
                            const uint32_t x4 = a8 ? d[2] : d[0];
                            const uint32_t rest = a8 ? (a4 ? d[3] : d[2]) : (a4 ? d[1] : d[0]);
                            uint32_t at = a8;
                            if (a8) {
                                reinterpret_cast<PackBytes4*>(dst)->x = d[0];
                                reinterpret_cast<PackBytes4*>(dst + 4)->x = d[1];
                            }
                            if (a4) {
                                reinterpret_cast<PackBytes4*>(dst + at)->x = x4;
                                at += 4u;
                            }
                            if (L & 2u) {
                                dst[at] = (uint8_t)rest;
                                dst[at + 1u] = (uint8_t)(rest >> 8);
                            }
                            if (L & 1u) dst[at + (L & 2u)] = (uint8_t)(rest >> (8u * (L & 2u)));
                        }
                    }
                }
            }
            PACK_CLK(6);
            (void)its;
#ifdef BWT_PACK_CLOCKS
            if (tid == 0 && b == 0)
                printf("packed inverse, block 0 (%u rows): sample %u pack %u delete %u counts %u walk %u order %u pieces %u copy %u ticks; wave 0: %u iterations, %u segments\n", n,
                       clk_[0], clk_[1], clk_[2], clk_[3], clk_[4], clk_[5], clk_[7], clk_[6], its, nseg);
#endif
            if (tid == 0) {
                a.lens[b] = total;
                a.mode[b] = kModeDone;
                atomicAdd(&a.queue[3], 1u);
            }
            continue;
        }
#endif
        const uint32_t its1 = packed_walk<false>(sh, m, drow, row0, mul, inv, mask, ids, cbytes, out_end, 0u, 0u);
        __syncthreads();
        PACK_CLK(4);
        if (sh.bad) continue;  // workgroup-uniform; never

        // ---- order: every record becomes bytes from its segment's start to the text's start
        uint32_t total;
        {
            constexpr int E = (kPackIds + 1 + kPackThreads - 1) / kPackThreads;
            uint32_t own[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint32_t i = (uint32_t)tid + (uint32_t)e * kPackThreads;
                own[e] = i <= kPackIds ? sh.seg[i] >> kPackNextBits : 0u;
            }
            for (int round = 0; round < (int)kPackNextBits; ++round) {
                uint32_t nv[E];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const uint32_t i = (uint32_t)tid + (uint32_t)e * kPackThreads;
                    nv[e] = 0;
                    if (i <= kPackIds) {
                        const uint32_t word = sh.seg[i];
                        const uint32_t next = word & kPackEnd;
                        nv[e] = word;
                        if (next != kPackEnd) {
                            const uint32_t w2 = sh.seg[next];
                            nv[e] = pack_seg((word >> kPackNextBits) + (w2 >> kPackNextBits), w2 & kPackEnd);
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const uint32_t i = (uint32_t)tid + (uint32_t)e * kPackThreads;
                    if (i <= kPackIds) sh.seg[i] = nv[e];
                }
                __syncthreads();
            }
            const uint32_t first = sh.seg[kPackIds];
            total = first >> kPackNextBits;
            if ((first & kPackEnd) != kPackEnd || total > m) continue;  // never for a permutation; left to the general kernels
            __syncthreads();
            // for walk 2: the segments on the chain (the first one is thread 0's), grouped by the stretch of
            // the text they write -- the stretch at the text's end first -- and by long / short
            const uint32_t stretch_shift = (uint32_t)max(bit_length(total ? total - 1u : 0u), 3) - 3u;
            uint32_t word[E], bucket[E], slot[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint32_t i = (uint32_t)tid + (uint32_t)e * kPackThreads;
                word[e] = i < kPackIds ? sh.seg[i] : 0u;
                const uint32_t togo = word[e] >> kPackNextBits;
                const bool on = (word[e] & kPackEnd) == kPackEnd && own[e] != 0u && togo <= total && i < kPackIds;
                bucket[e] = kPackBuckets;
                slot[e] = 0;
                if (on) {
                    bucket[e] = min((total - togo) >> stretch_shift, kPackStretches - 1u) * 2u + (own[e] < kPackLong ? 1u : 0u);
                    slot[e] = atomicAdd(&sh.bucket_fill[bucket[e]], 1u);
                }
            }
            __syncthreads();
            if (tid == 0) {
                uint32_t run = 0;
                for (uint32_t k = 0; k < kPackBuckets; ++k) {
                    sh.bucket_base[k] = run;
                    run += sh.bucket_fill[k];
                }
                sh.bucket_base[kPackBuckets] = run;
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint32_t i = (uint32_t)tid + (uint32_t)e * kPackThreads;
                if (bucket[e] < kPackBuckets) sh.seg[sh.bucket_base[bucket[e]] + slot[e]] = pack_seg(word[e] >> kPackNextBits, i);
            }
        }
        PACK_CLK(5);
        if (tid == 0) sh.ctr = 0;
        __syncthreads();
        const uint32_t its2 = packed_walk<true>(sh, m, drow, row0, mul, inv, mask, ids, cbytes, out_end, total, sh.bucket_base[kPackBuckets]);
        __syncthreads();
        if (sh.bad) continue;  // workgroup-uniform; never
        PACK_CLK(6);
        (void)its1;
        (void)its2;
#ifdef BWT_PACK_CLOCKS
        if (tid == 0 && b == 0)
            printf("packed inverse, block 0 (%u rows): sample %u pack %u delete %u counts %u walk1 %u order %u walk2 %u ticks; wave 0: %u and %u iterations\n", n,
                   clk_[0], clk_[1], clk_[2], clk_[3], clk_[4], clk_[5], clk_[6], its1, its2);
#endif
        if (tid == 0) {
            a.lens[b] = total;
            a.mode[b] = kModeDone;
            atomicAdd(&a.queue[3], 1u);
        }
    }
}

}  // namespace bwtgpu
