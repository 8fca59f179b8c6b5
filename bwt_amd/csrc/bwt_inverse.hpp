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
// Kernel 1, bwt_inverse_build_kernel (one workgroup per block): the LF table
// T[i] = (C[L[i]] + occ[i]) | splitter flag << 23 | L[i] << 24 (TableWord) -- an LDS byte histogram per wave
// chunk, an exclusive scan in (symbol, wave) order, then a stable per-symbol ranking of every
// position (wave-ballot match, the same primitive as the forward radix passes).  Packing the
// symbol next to the successor row makes a walk step ONE dependent 4-byte load.
//
// The walk itself is a pointer chase of n dependent loads; one chain per block leaves the
// memory system idle, so a well-formed block (exactly one delimiter) is cut into segments:
// one row in 16 or 32 and the delimiter row are "splitters"; one lane per splitter walks until
// it reaches the next splitter (whose own byte still belongs to it).  Which rows: those whose
// image under row -> row * A mod n (A coprime to n, A / n close to the golden ratio) is below
// n / 16 -- the image is the segment's id.  Every 16th row would do for random text, but in the
// long runs of repetitive input the walk moves by a constant row distance d per step, and with
// d even it can go thousands of steps without meeting a multiple of 16; under the
// multiplication any such progression keeps meeting the splitter interval.
//   bwt_inverse_walk_segments_kernel  every segment's length, the splitter it runs into and its
//                                     bytes (into a 128-byte staging slot per segment; the few
//                                     longer ones chain chunks from a per-block pool)
//   bwt_inverse_stitch_kernel         orders the segments by following those links from the
//                                     delimiter row: where each segment's bytes go
//   bwt_inverse_place_kernel          copies every segment, reversed, to its place
// The lanes of a block are placed on ONE XCD (workgroup w runs on XCD w % 8), and with a
// splitter every 32 rows a block brings enough lanes that an XCD works on only a few blocks at
// a time, whose LF tables (4 bytes per row) then largely stay in that XCD's 4 MiB L2: a
// dependent random load costs an L2 hit instead of a 64-byte HBM transaction.  The staging
// stores bypass the L2 so they do not evict the tables.  The result is the chain the reference
// walks, byte for byte.
// Blocks with several delimiters (only possible for hand-made input) keep the reference's
// exact semantics -- start at the last delimiter row, stop at the first delimiter reached --
// through the single-chain kernel (bwt_inverse_walk_kernel), as do blocks too short to split.
#pragma once
#include "bwt_device.hpp"
#include "bwtgpu.h"

namespace bwtgpu {

constexpr uint32_t kNoRow = 0xFFFFFFFFu;
constexpr uint32_t kRowMask = 0x007FFFFFu;   // successor row in a 32-bit table word (blocks of at most 2^23 rows)
constexpr uint32_t kSplitFlag = 0x00800000u; // this row starts a segment
constexpr uint32_t kNarrowMaxRows = 1u << 21;  // longer blocks get 64-bit table words
constexpr uint32_t kMaxSegments = 1u << 17;    // regular segments per block, at most: longer blocks are cut more coarsely

// One word of the LF table: successor row, "this row starts a segment", the row's byte.  Blocks of
// up to kNarrowMaxRows rows pack it into 32 bits (one 4-byte load per walk step, twice the rows
// per cache line); longer blocks -- which only the tiled build handles -- use 64 bits.
template <typename TW>
struct TableWord;
template <>
struct TableWord<uint32_t> {
    static __device__ __forceinline__ uint32_t pack(uint32_t row, bool split, uint32_t c) {
        return row | (split ? kSplitFlag : 0u) | (c << 24);
    }
    static __device__ __forceinline__ uint32_t row(uint32_t w) { return w & kRowMask; }
    static __device__ __forceinline__ bool split(uint32_t w) { return (w & kSplitFlag) != 0u; }
    static __device__ __forceinline__ uint32_t byte(uint32_t w) { return w >> 24; }
};
template <>
struct TableWord<uint64_t> {
    static __device__ __forceinline__ uint64_t pack(uint32_t row, bool split, uint32_t c) {
        return (uint64_t)row | ((uint64_t)(split ? 1u : 0u) << 32) | ((uint64_t)c << 40);
    }
    static __device__ __forceinline__ uint32_t row(uint64_t w) { return (uint32_t)w; }
    static __device__ __forceinline__ bool split(uint64_t w) { return ((w >> 32) & 1u) != 0u; }
    static __device__ __forceinline__ uint32_t byte(uint64_t w) { return (uint32_t)(w >> 40) & 255u; }
};
constexpr uint32_t kMinSplitShift = 4;       // rows between regular splitters: 16 or 32 (InvArgs::split_shift),
constexpr uint32_t kMultiMinRows = 4096;     // shorter blocks are walked as one chain when the batch has many blocks
constexpr uint32_t kMultiMinRowsFew = 1280;  // ... and this is the bound when it has few: the chains are the whole cost then
constexpr uint32_t kFewBlocks = 2048;
#ifndef BWT_SEG_THREADS
#define BWT_SEG_THREADS 128
#endif
constexpr int kSegThreads = BWT_SEG_THREADS;  // workgroup size of the place kernel
                                             // staging slot per segment = 4 x that many bytes (InvArgs::slot)
constexpr uint32_t kNoPos = 0xFFFFFFFFu;
constexpr uint32_t kModeMulti = 0, kModeSingle = 1, kModeNone = 2;
constexpr uint32_t kModeDone = 3;  // the block's text has been written already (bwt_inverse_packed_kernel)

struct InvArgs {
    const uint8_t* in;    // block b reads in + b*(block_size+1)
    uint8_t* out;         // block b's text ends at out + b*block_size + len_b
    uint64_t total_len;   // bytes of `in` (texts + one delimiter per block)
    uint32_t block_size;
    uint32_t nblocks;
    uint32_t delim;
    uint32_t tstride;     // LF-table words per block (>= block_size+1)
    uint32_t seg_cap;     // segment slots per block (regular splitters + 1)
    uint32_t seg_wgs;     // workgroups of kSegThreads lanes that cover one block's segments
    uint32_t multi_min_rows;  // blocks of at least this many rows are cut into segments
    void* T;              // nblocks * tstride table words (TableWord: 32 or 64 bits)
    uint32_t* drow;       // per block: delimiter row or kNoRow
    uint32_t* mode;       // per block: kModeMulti / kModeSingle / kModeNone
    uint2* seg;           // nblocks * seg_cap: x = bytes the segment produces, y = id of the splitter it runs into
    uint2* seg_rec;       // nblocks * seg_cap: the segments in the order the chain visits them (= the text back to front):
                          // x = segment id, y = bytes between the segment's first byte and the text end
    uint2* chain;         // per block: x = segments on the chain, y = bytes they produce
    uint32_t split_shift; // log2 of the rows between regular splitters
    uint32_t slot;        // bytes of staging per segment (power of two); longer segments chain pool chunks of the same size
    uint8_t* stage;       // nblocks * stage_cap * slot: the first `slot` bytes of every segment, in walk order (stage_group)
    uint32_t stage_cap;   // seg_cap rounded up to a multiple of 64
    uint32_t* seg_ovf;    // nblocks * seg_cap: first pool chunk of a segment longer than a slot (valid only then)
    uint32_t chunk_cap;   // pool chunks per block
    uint32_t* pool_used;  // per block: chunks handed out
    uint32_t* chunk_next; // nblocks * chunk_cap: the chunk that follows (valid only if the segment goes on)
    uint8_t* pool;        // nblocks * chunk_cap * slot
    uint32_t* lens;       // per block: text length or BWTGPU_LEN_NO_DELIMITER
    uint32_t* queue;      // [0] next block index to hand out
    // splitter map row -> row * mul mod n and its inverse id -> id * inv mod n; [0] for full
    // blocks (n = block_size + 1), [1] for the last block of the batch
    uint32_t split_mul[2], split_inv[2];
    // Origin-pointer blocks (the container of reference util/bzip2_bwt_utils.cpp:137-198): the last
    // column of the sorted ROTATIONS of the text, no delimiter; orig_ptr[b] is the row of the text
    // itself.  Null for the delimiter format.  Such a block has n = text length rows; the kernels
    // treat row orig_ptr[b] as they treat the delimiter row (it starts and ends the chain and its byte
    // is left out), which yields the text's first n - 1 bytes; bwt_inverse_origin_finish_kernel adds
    // the last one.  `delim` is 256 then (no byte matches it).
    const uint32_t* orig_ptr;
    uint32_t out_stride;  // block b's text starts at out + b * out_stride (block_size; origin-pointer blocks: block_size + 1)
    uint32_t* left_hint;    // host memory the device can write: blocks of this batch that the in-LDS inverse did NOT finish
    uint32_t packed_first;  // bwt_inverse_packed_kernel has run over the batch: mode[] is valid, blocks in kModeDone are finished
};

template <int NT>
struct InvShared {
    static constexpr int W = NT / kWave;
    // Byte counts per wave, kept in kHistCopies copies that the lanes of a wave are spread over: with a
    // small alphabet (ACGT) all 64 lanes of a counting instruction hit the same few words, which the LDS
    // serialises.  After the scan the first W * 256 words hold the per-wave start offsets.
    static constexpr int kHistCopies = 4;  // measured: 2 copies 8.3 ms, 4 copies 7.8 ms, 8 copies 10.2 ms (occupancy) on 4 GiB of ACGT text
    uint32_t hist[W * kHistCopies * 256];
    uint32_t wsum[W];
    uint32_t drow[W];
    uint32_t ndelim;
    uint32_t blk;
    uint32_t sigma;       // distinct byte values in the block
    uint8_t code[256];    // byte value -> its index among the values that occur (ranking compares codes: fewer bits, fewer ballots)
};

__device__ __forceinline__ uint32_t inv_block_len(const InvArgs& a, uint32_t b) {
    const uint64_t off = (uint64_t)b * ((uint64_t)a.block_size + 1u);
    const uint64_t rem = a.total_len - off;
    return rem < (uint64_t)a.block_size + 1u ? (uint32_t)rem : a.block_size + 1u;
}

// Regular splitters of a block of n rows: one row in 2^split_shift.
__device__ __host__ __forceinline__ uint32_t regular_splitters(uint32_t n, uint32_t split_shift) {
    return (n + (1u << split_shift) - 1u) >> split_shift;
}
// Row -> splitter id (a splitter iff below regular_splitters) and id -> row.
__device__ __forceinline__ uint32_t mul_mod(uint32_t x, uint32_t m, uint32_t n) {
    return (uint32_t)(((uint64_t)x * m) % n);
}

// Staging of the first `slot` bytes of segment j of block b, in 16-byte groups.  The 64 segments that
// one wave walks fill their q-th groups in the same step, so those lie side by side (one 1 KiB run per
// step instead of 64 stores that each touch a line of their own): group q of segment j is at
// stage_group(a, b, j) + q * kStageGroupStride.
constexpr uint32_t kStageGroupStride = 64u * 16u;
__device__ __forceinline__ uint8_t* stage_group(const InvArgs& a, uint32_t b, uint32_t j) {
    return a.stage + ((uint64_t)b * a.stage_cap + (j & ~63u)) * a.slot + (j & 63u) * 16u;
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
        if (a.packed_first && a.mode[b] == kModeDone) continue;  // workgroup-uniform
        const uint32_t n = inv_block_len(a, b);
        const uint8_t* __restrict__ L = a.in + (uint64_t)b * ((uint64_t)a.block_size + 1u);
        uint32_t* __restrict__ T = (uint32_t*)a.T + (uint64_t)b * a.tstride;

        constexpr int HC = InvShared<NT>::kHistCopies;
        for (int i = tid; i < W * HC * 256; i += NT) sh.hist[i] = 0;
        __syncthreads();
        Chunking ck;
        ck.init(n, W);
        const uint32_t lo = (uint32_t)w * ck.chunk, hi = min(lo + ck.chunk, n);
        // pass 1: histogram per wave chunk + last delimiter row.  Order does not matter here, so the
        // chunk is read a dword per lane (the ranking pass below needs one byte per lane, in row order)
        constexpr int U = 8;  // vectors whose loads are issued together
        uint32_t last = kNoRow;
        {
            uint32_t lastp = 0;  // row + 1 of the last delimiter this lane has seen
            uint32_t* myh = sh.hist + (w * HC + (lane & (HC - 1))) * 256;
            auto take = [&](uint32_t c, uint32_t row) {
                atomicAdd(&myh[c], 1u);
                if (c == a.delim) lastp = max(lastp, row + 1u);
            };
            if (lo < hi) {
                const uint8_t* p = L + lo;
                const uint32_t n1 = hi - lo;
                const uint32_t head = min(n1, (uint32_t)((4u - (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3u)) & 3u));
                const uint32_t nd = (n1 - head) / 4u;
                const uint32_t* __restrict__ pd = reinterpret_cast<const uint32_t*>(p + head);
                if ((uint32_t)lane < head) take(p[lane], lo + (uint32_t)lane);
                for (uint32_t i0 = (uint32_t)lane; i0 < nd; i0 += U * kWave) {
                    uint32_t d[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) d[u] = pd[min(i0 + (uint32_t)(u * kWave), nd - 1u)];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const uint32_t i = i0 + (uint32_t)(u * kWave);
                        if (i < nd) {
                            const uint32_t row = lo + head + 4u * i;
                            take(d[u] & 255u, row);
                            take((d[u] >> 8) & 255u, row + 1u);
                            take((d[u] >> 16) & 255u, row + 2u);
                            take(d[u] >> 24, row + 3u);
                        }
                    }
                }
                const uint32_t done = head + 4u * nd;
                if (done + (uint32_t)lane < n1) take(p[done + (uint32_t)lane], lo + done + (uint32_t)lane);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) lastp = max(lastp, (uint32_t)__shfl_xor(lastp, o));
            if (lastp) last = lastp - 1u;
        }
        if (lane == 0) sh.drow[w] = last;
        __syncthreads();
        // C[] and per-wave start offsets: exclusive scan in (symbol major, wave minor) order
        {
            uint32_t tot = 0;
            if (tid < 256)
                for (int i = 0; i < W * HC; ++i) tot += sh.hist[i * 256 + tid];
            if (tid == (int)a.delim) sh.ndelim = tot;
            uint32_t total;
            uint32_t run = block_exclusive_scan<NT>(tot, sh.wsum, total);
            if (tid < 256)
                for (int i = 0; i < W; ++i) {  // row i is written after rows HC i .. have been read: never ahead of the reads
                    uint32_t v = 0;
#pragma unroll
                    for (int k = 0; k < HC; ++k) v += sh.hist[(i * HC + k) * 256 + tid];
                    sh.hist[i * 256 + tid] = run;
                    run += v;
                }
            // dense codes of the byte values that occur
            uint32_t sigma;
            const uint32_t below = block_exclusive_scan<NT>(tid < 256 && tot ? 1u : 0u, sh.wsum, sigma);
            if (tid < 256) sh.code[tid] = (uint8_t)below;
            if (tid == 0) sh.sigma = sigma;
        }
        __syncthreads();
        const int cbits = max(1, bit_length(sh.sigma - 1u));
        uint32_t drow = kNoRow;  // rows grow with the wave index, so the last hit wins
#pragma unroll
        for (int i = 0; i < W; ++i)
            if (sh.drow[i] != kNoRow) drow = sh.drow[i];
        // split the walk only when it is the plain single-delimiter case and long enough to pay
        uint32_t mode = drow == kNoRow ? kModeNone
                        : (sh.ndelim == 1u && n >= a.multi_min_rows) ? kModeMulti : kModeSingle;
        if (a.orig_ptr) {  // the chain starts at the given row; a row outside the block is refused
            drow = a.orig_ptr[b] < n ? a.orig_ptr[b] : kNoRow;
            mode = drow == kNoRow ? kModeNone : n >= a.multi_min_rows ? kModeMulti : kModeSingle;
        }
        if (tid == 0) {
            a.drow[b] = drow;
            a.mode[b] = mode;
            a.pool_used[b] = 0;
        }
        const bool split = mode == kModeMulti;
        // pass 2: LF table
        uint32_t* myoff = sh.hist + w * 256;
        const uint32_t nreg = regular_splitters(n, a.split_shift);
        const uint32_t mul = a.split_mul[b + 1u == a.nblocks ? 1 : 0];
        const uint32_t step = mul_mod(kWave, mul, n);          // image distance of rows 64 apart
        uint32_t image = mul_mod(lo + (uint32_t)lane, mul, n);  // of row base + lane
        for (uint32_t base0 = lo; base0 < hi; base0 += U * kWave) {
            uint32_t cs[U];
#pragma unroll
            for (int u = 0; u < U; ++u) cs[u] = L[min(base0 + (uint32_t)(u * kWave + lane), n - 1u)];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t base = base0 + (uint32_t)(u * kWave);
                if (base < hi) {  // wave-uniform
                    const uint32_t k = base + lane;
                    const bool valid = k < hi;
                    const uint32_t c = cs[u];
                    const uint64_t peers = match_digit_bits((uint32_t)sh.code[c], valid, cbits);
                    const uint32_t rk = __popcll(peers & lt);
                    const uint32_t b0 = myoff[c];
                    compiler_fence();  // every lane has read its offset before the leaders advance it
                    if (valid && rk == 0) myoff[c] = b0 + (uint32_t)__popcll(peers);
                    compiler_fence();
                    if (valid) T[k] = TableWord<uint32_t>::pack(b0 + rk, split && (image < nreg || k == drow), c);
                    image += step;
                    image -= image >= n ? n : 0u;
                }
            }
        }
        __syncthreads();
    }
}

// ---- LF table of FEW LARGE blocks: many workgroups per block --------------------------------------
// One workgroup per block leaves the chip idle when the batch has only a few blocks.  Here a
// block is cut into tiles of `tile_rows` rows; a first kernel counts the byte values of every
// tile (and finds the delimiter rows), the second gives every tile its own workgroup, which
// derives its start offsets from the counts of all tiles of its block (thread c adds up column
// c: C[c] plus the occurrences of c in the tiles before) and then ranks its rows exactly as
// bwt_inverse_build_kernel does.
struct InvTileArgs {
    uint32_t tile_rows;       // rows per tile (multiple of 64 * waves)
    uint32_t tiles_per_block;
    uint32_t* tile_hist;      // nblocks * tiles_per_block * 256
    uint32_t* dacc;           // per block: [0] last delimiter row + 1 (0 = none), [1] delimiters
};

template <int NT>
__global__ __launch_bounds__(NT) void bwt_inverse_tile_hist_kernel(InvArgs a, InvTileArgs ta) {
    __shared__ uint32_t s_h[256];
    __shared__ uint32_t s_last, s_cnt;
    const int tid = threadIdx.x;
    const uint32_t b = blockIdx.x / ta.tiles_per_block, t = blockIdx.x % ta.tiles_per_block;
    const uint32_t n = inv_block_len(a, b);
    const uint8_t* __restrict__ L = a.in + (uint64_t)b * ((uint64_t)a.block_size + 1u);
    for (int i = tid; i < 256; i += NT) s_h[i] = 0;
    if (tid == 0) s_last = s_cnt = 0;
    __syncthreads();
    const uint32_t lo = min(t * ta.tile_rows, n), hi = min(lo + ta.tile_rows, n);
    uint32_t last = 0, cnt = 0;
    for (uint32_t k = lo + tid; k < hi; k += NT) {
        const uint32_t c = L[k];
        atomicAdd(&s_h[c], 1u);
        if (c == a.delim) {
            last = k + 1u;
            ++cnt;
        }
    }
    if (cnt) {
        atomicMax(&s_last, last);
        atomicAdd(&s_cnt, cnt);
    }
    __syncthreads();
    uint32_t* __restrict__ h = ta.tile_hist + (uint64_t)blockIdx.x * 256u;
    for (int i = tid; i < 256; i += NT) h[i] = s_h[i];
    if (tid == 0 && s_cnt) {
        atomicMax(&ta.dacc[2u * b], s_last);
        atomicAdd(&ta.dacc[2u * b + 1u], s_cnt);
    }
}

template <int NT, typename TW>
__global__ __launch_bounds__(NT) void bwt_inverse_tile_build_kernel(InvArgs a, InvTileArgs ta) {
    constexpr int W = NT / kWave;
    static_assert(NT >= 256, "one thread per symbol in the scan");
    __shared__ InvShared<NT> sh;
    __shared__ uint32_t s_base[256];
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const uint64_t lt = lanemask_lt();
    const uint32_t b = blockIdx.x / ta.tiles_per_block, t = blockIdx.x % ta.tiles_per_block;
    const uint32_t n = inv_block_len(a, b);
    const uint8_t* __restrict__ L = a.in + (uint64_t)b * ((uint64_t)a.block_size + 1u);
    TW* __restrict__ T = (TW*)a.T + (uint64_t)b * a.tstride;
    const uint32_t tlo = min(t * ta.tile_rows, n), thi = min(tlo + ta.tile_rows, n);
    for (int i = tid; i < W * 256; i += NT) sh.hist[i] = 0;
    // where the rows of this tile with byte value c go: C[c] + occurrences in the tiles before
    {
        uint32_t before = 0, all = 0;
        if (tid < 256) {
            const uint32_t* __restrict__ h = ta.tile_hist + (uint64_t)b * ta.tiles_per_block * 256u + (uint32_t)tid;
            uint32_t i = 0;
            for (; i + 8u <= ta.tiles_per_block; i += 8u) {
                uint32_t v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = h[(i + (uint32_t)u) * 256u];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    all += v[u];
                    before += i + (uint32_t)u < t ? v[u] : 0u;
                }
            }
            for (; i < ta.tiles_per_block; ++i) {
                const uint32_t v = h[i * 256u];
                all += v;
                before += i < t ? v : 0u;
            }
        }
        uint32_t total;
        const uint32_t c0 = block_exclusive_scan<NT>(all, sh.wsum, total);
        if (tid < 256) s_base[tid] = c0 + before;
        uint32_t sigma;  // dense codes of the byte values that occur in the block, as in bwt_inverse_build_kernel
        const uint32_t below = block_exclusive_scan<NT>(tid < 256 && all ? 1u : 0u, sh.wsum, sigma);
        if (tid < 256) sh.code[tid] = (uint8_t)below;
        if (tid == 0) sh.sigma = sigma;
    }
    const uint32_t dlast = ta.dacc[2u * b], ndelim = ta.dacc[2u * b + 1u];
    uint32_t drow = dlast ? dlast - 1u : kNoRow;
    uint32_t mode = drow == kNoRow ? kModeNone : (ndelim == 1u && n >= a.multi_min_rows) ? kModeMulti : kModeSingle;
    if (a.orig_ptr) {
        drow = a.orig_ptr[b] < n ? a.orig_ptr[b] : kNoRow;
        mode = drow == kNoRow ? kModeNone : n >= a.multi_min_rows ? kModeMulti : kModeSingle;
    }
    if (t == 0 && tid == 0) {
        a.drow[b] = drow;
        a.mode[b] = mode;
        a.pool_used[b] = 0;
    }
    const bool split = mode == kModeMulti;
    __syncthreads();
    const int cbits = max(1, bit_length(sh.sigma - 1u));
    Chunking ck;
    ck.init(thi - tlo, W);
    const uint32_t lo = min(tlo + (uint32_t)w * ck.chunk, thi), hi = min(lo + ck.chunk, thi);
    constexpr int U = 4;
    // pass 1: byte counts of every wave's share of the tile
    for (uint32_t base0 = lo; base0 < hi; base0 += U * kWave) {
        uint32_t cs[U];
#pragma unroll
        for (int u = 0; u < U; ++u) cs[u] = L[min(base0 + (uint32_t)(u * kWave + lane), n - 1u)];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (base0 + (uint32_t)(u * kWave + lane) < hi) atomicAdd(&sh.hist[w * 256 + cs[u]], 1u);
    }
    __syncthreads();
    if (tid < 256) {
        uint32_t run = s_base[tid];
        for (int i = 0; i < W; ++i) {
            const uint32_t v = sh.hist[i * 256 + tid];
            sh.hist[i * 256 + tid] = run;
            run += v;
        }
    }
    __syncthreads();
    // pass 2: LF table
    uint32_t* myoff = sh.hist + w * 256;
    const uint32_t nreg = regular_splitters(n, a.split_shift);
    const uint32_t mul = a.split_mul[b + 1u == a.nblocks ? 1 : 0];
    const uint32_t step = mul_mod(kWave, mul, n);
    uint32_t image = mul_mod(lo + (uint32_t)lane, mul, n);
    for (uint32_t base0 = lo; base0 < hi; base0 += U * kWave) {
        uint32_t cs[U];
#pragma unroll
        for (int u = 0; u < U; ++u) cs[u] = L[min(base0 + (uint32_t)(u * kWave + lane), n - 1u)];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t base = base0 + (uint32_t)(u * kWave);
            if (base < hi) {  // wave-uniform
                const uint32_t k = base + lane;
                const bool valid = k < hi;
                const uint32_t c = cs[u];
                const uint64_t peers = match_digit_bits((uint32_t)sh.code[c], valid, cbits);
                const uint32_t rk = __popcll(peers & lt);
                const uint32_t b0 = myoff[c];
                compiler_fence();  // every lane has read its offset before the leaders advance it
                if (valid && rk == 0) myoff[c] = b0 + (uint32_t)__popcll(peers);
                compiler_fence();
                if (valid) T[k] = TableWord<TW>::pack(b0 + rk, split && (image < nreg || k == drow), c);
                image += step;
                image -= image >= n ? n : 0u;
            }
        }
    }
}

// Collects bytes that arrive back to front and stores them below `end`: whole aligned 16-byte
// groups where possible, single dwords and bytes at the two ragged ends, never a byte outside
// [end - count, end).
struct BackWriter {
    uint8_t* addr;   // lowest address produced so far
    uint8_t* daddr;  // where the completed dwords q0.. start
    uint32_t acc, cnt, q0, q1, q2, q3, dcnt;
    __device__ __forceinline__ void init(uint8_t* end) {
        addr = daddr = end;
        acc = cnt = q0 = q1 = q2 = q3 = dcnt = 0;
    }
    __device__ __forceinline__ void push(uint32_t ch) {
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
            } else {  // ragged upper end: the first dword is partial
                for (uint32_t k = 0; k < cnt; ++k) addr[k] = (uint8_t)(acc >> (8 * k));
            }
            acc = 0;
            cnt = 0;
        }
    }
    // Four bytes at once: v's low byte goes to the lowest address.  Only at a dword boundary.
    __device__ __forceinline__ void push_dword(uint32_t v) {
        addr -= 4;
        q3 = q2; q2 = q1; q1 = q0; q0 = v;
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
    }
    __device__ __forceinline__ void flush() {
        for (uint32_t k = 0; k < cnt; ++k) addr[k] = (uint8_t)(acc >> (8 * k));
        if (dcnt) {
            uint32_t* d = reinterpret_cast<uint32_t*>(daddr);
            d[0] = q0;
            if (dcnt > 1) d[1] = q1;
            if (dcnt > 2) d[2] = q2;
            if (dcnt > 3) d[3] = q3;
        }
    }
};

// ---- single chain (short blocks, blocks with several delimiters) -------------------------------
// One chain per block, one lane per chain.  Every step is a single dependent load; the
// loop is bounded by the block length, and a well-formed block ends it exactly there by
// walking back into the delimiter row.  Output bytes are produced back to front; they are
// packed in registers and leave as aligned 16-byte stores (a lane-private store per byte
// would be one scattered write per step).
template <typename TW>
__global__ __launch_bounds__(kWave) void bwt_inverse_walk_kernel(InvArgs a) {
    const uint32_t b = blockIdx.x * kWave + threadIdx.x;
    if (b >= a.nblocks) return;
    const uint32_t mode = a.mode[b];
    if (mode == kModeMulti || mode == kModeDone) return;
    const uint32_t n = inv_block_len(a, b);
    const uint32_t len = n - 1;
    uint32_t row = a.drow[b];
    if (mode == kModeNone || row == kNoRow) {
        a.lens[b] = 0xFFFFFFFFu;
        return;
    }
    const TW* __restrict__ T = (const TW*)a.T + (uint64_t)b * a.tstride;
    uint8_t* const o = a.out + (uint64_t)b * a.out_stride;
    BackWriter bw;
    bw.init(o + len);
    row = TableWord<TW>::row(T[row]);
    for (uint32_t step = 0; step < n; ++step) {
        const TW wd = T[row];
        const uint32_t ch = TableWord<TW>::byte(wd);
        if (ch == a.delim || bw.addr == o) break;
        row = TableWord<TW>::row(wd);
        bw.push(ch);
    }
    bw.flush();
    a.lens[b] = (uint32_t)((o + len) - bw.addr);
}

// ---- segmented walk -------------------------------------------------------------------------
// Lane -> (block, segment).  Workgroups are dealt to the XCDs round robin, so workgroup w of
// the grid serves block (w / 8 / seg_wgs) * 8 + w % 8: all workgroups of one block share an
// XCD, and an XCD works through its blocks in order.  (If the hardware placed workgroups
// differently only the cache locality would suffer.)
// Segment j < regular_splitters(n) starts at the row whose image is j, j == regular_splitters(n)
// starts at the delimiter row.  (A regular splitter that IS the delimiter row is left to the latter.)
template <int NT = kSegThreads>
__device__ __forceinline__ bool segment_of_lane(const InvArgs& a, uint32_t w, uint32_t& b, uint32_t& j) {
    const uint32_t wgs = (a.seg_cap + (uint32_t)NT - 1u) / (uint32_t)NT;  // workgroups that cover one block's segments
    const uint32_t x = w & 7u, q = w >> 3;
    b = (q / wgs) * 8u + x;
    j = (q % wgs) * (uint32_t)NT + threadIdx.x;
    // the mode is wave-uniform and not written by this kernel: a load the scalar unit can do
    return b < a.nblocks && __ldg(&a.mode[b]) == kModeMulti;
}
// The walk runs in workgroups of a single wave: a workgroup's slot on its compute unit is taken until
// its LONGEST segment is done (segment lengths are geometric, mean 16: the longest of 64 is about 76
// steps, of 256 about 98), and a slot that is given back earlier starts the next wave earlier.
constexpr int kWalkThreads = kWave;

// Walks every segment once: its length, the splitter it runs into, and its bytes -- in walk
// order, 16 at a time -- into the segment's staging slot (then chained pool chunks).
// Workgroup w of the grid the text above describes; the kernel below runs one or several of them.
template <typename TW>
__device__ __forceinline__ void walk_segments_unit(const InvArgs& a, uint32_t w) {
    uint32_t b, j;
    if (!segment_of_lane<kWalkThreads>(a, w, b, j)) return;
    const uint32_t n = inv_block_len(a, b);
    const uint32_t nreg = regular_splitters(n, a.split_shift);
    if (j > nreg) return;
    const uint32_t drow = a.drow[b];
    const uint64_t sb = (uint64_t)b * a.seg_cap;
    const int last = b + 1u == a.nblocks ? 1 : 0;
    const uint32_t start = j < nreg ? mul_mod(j, a.split_inv[last], n) : drow;
    if (j < nreg && start == drow) {
        a.seg[sb + j] = make_uint2(0u, j);
        return;
    }
    const TW* __restrict__ T = (const TW*)a.T + (uint64_t)b * a.tstride;
    const uint64_t pb = (uint64_t)b * a.chunk_cap;
    uint8_t* buf = stage_group(a, b, j);
    uint32_t gstride = kStageGroupStride;  // distance of the 16-byte groups: side by side in a pool chunk
    uint32_t prevc = 0;
    uint32_t row = TableWord<TW>::row(T[start]);
    uint32_t t = 0;  // bytes the segment produces
    uint32_t acc = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    uint32_t stop = start;
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    for (uint32_t step = 0; step < n; ++step) {
        const TW wd = T[row];
        const bool at_splitter = TableWord<TW>::split(wd);
        if (at_splitter && row == drow) {  // the delimiter row ends the text and has no byte
            stop = row;
            break;
        }
        if (t && (t & (a.slot - 1u)) == 0u) {  // the slot (or chunk) is full: chain a pool chunk
            const uint32_t c = atomicAdd(&a.pool_used[b], 1u);
            if (c >= a.chunk_cap) break;  // cannot happen: the pool holds n / slot + 8 chunks
            if (t == a.slot) a.seg_ovf[sb + j] = c;
            else a.chunk_next[pb + prevc] = c;
            prevc = c;
            buf = a.pool + (pb + c) * a.slot;
            gstride = 16u;
        }
        acc |= TableWord<TW>::byte(wd) << (8u * (t & 3u));
        ++t;
        if ((t & 3u) == 0u) {
            q0 = q1; q1 = q2; q2 = q3; q3 = acc;
            acc = 0;
            if ((t & 15u) == 0u) {
                const v4 val = {q0, q1, q2, q3};
                __builtin_nontemporal_store(val, reinterpret_cast<v4*>(buf + (((t - 16u) & (a.slot - 1u)) >> 4) * gstride));
            }
        }
        if (at_splitter) {  // a splitter's own byte belongs to the segment that reaches it
            stop = row;
            break;
        }
        row = TableWord<TW>::row(wd);
    }
    if (t & 15u) {  // flush the partial 16-byte group (the slot has room; bytes past t are never used)
        const uint32_t have = (t & 15u) >> 2;  // complete dwords in the group: q3 newest
        v4 val = {0u, 0u, 0u, 0u};
        if (have == 0) { val.x = acc; }
        else if (have == 1) { val.x = q3; val.y = acc; }
        else if (have == 2) { val.x = q2; val.y = q3; val.z = acc; }
        else { val.x = q1; val.y = q2; val.z = q3; val.w = acc; }
        __builtin_nontemporal_store(val, reinterpret_cast<v4*>(buf + (((t & ~15u) & (a.slot - 1u)) >> 4) * gstride));
    }
    a.seg[sb + j] = make_uint2(t, stop == drow ? nreg : mul_mod(stop, a.split_mul[last], n));
}
// `units` workgroups' worth of work, normally one workgroup per unit.  When the in-LDS inverse of
// small-alphabet blocks is expected to finish every block (it did so in the device's previous batch), the
// host launches a grid capped at kSparseGrid workgroups that strides over the units instead of
// dispatching millions of workgroups that return at once (the stride is a multiple of 8: a unit stays
// on the XCD its index names).  Blocks that do need this kernel are still done then, only more slowly:
// the stride gives up the order in which neighbouring workgroups share an XCD's L2.
constexpr uint32_t kSparseGrid = 16384;
template <typename TW, bool STRIDED>
__global__ __launch_bounds__(kWalkThreads) void bwt_inverse_walk_segments_kernel(InvArgs a, uint32_t units) {
    if (STRIDED) {
        for (uint32_t w = blockIdx.x; w < units; w += gridDim.x) walk_segments_unit<TW>(a, w);
    } else {
        walk_segments_unit<TW>(a, blockIdx.x);
    }
}

// Orders the segments of every split block: start at the delimiter row's segment, follow the
// next-splitter links until the chain returns to it.  The result is seg_rec: the segments in chain
// order, each with the bytes before it counted from the END of the text (the walk produces the
// text back to front).
// One workgroup per block.  Following thousands of links one after the other would take
// milliseconds, so every S-th segment (and the delimiter's) is a "super" node: each thread
// follows the links from its super node to the next one, summing the lengths (short independent
// chains), the workgroup then orders the super nodes, and each thread walks its chain again to
// hand out the ordinals and positions.  A chain ends at the first segment whose id is a multiple
// of S, and the ids along a chain are as good as random: the chains' lengths are geometric with
// mean S, and the longest of them, about S ln(segments / S), sets the pace of both walks.  So S
// is the smallest power of two, from 8 up, whose super nodes the workgroup's tables hold.
constexpr uint32_t kMinSuperShift = 3;
constexpr uint32_t kSuperNone = 0x1FFFu;  // "no next super node" in the packed word below
constexpr uint32_t kStitchSmallSegs = 8192;  // segments of a block, at most, for the 256-thread stitch kernel
static_assert(kMaxSegments < (1u << 18), "next super node in 13 bits, segment count above it");
// log2 of the super-node stride for a block of nreg regular segments and tables of `supers` entries
__device__ __forceinline__ uint32_t super_shift(uint32_t nreg, uint32_t supers) {
    uint32_t sh = kMinSuperShift;
    while (((nreg + (1u << sh) - 1u) >> sh) + 1u > supers) ++sh;
    return sh;
}

// What a super node's chain adds up to: bytes, segments, and the super node it ends at.
struct SuperChain {
    uint32_t bytes, segments, next;
};
// Follows the links from segment `id` to the next super node (bounded: a malformed block may link
// segments in a cycle).  seg_at(id) -> {bytes, next id}; super_of(id) -> super index or kNoPos.
template <typename SegAt, typename SuperOf>
__device__ __forceinline__ SuperChain follow_to_super(uint32_t id, uint32_t nreg, SegAt seg_at, SuperOf super_of) {
    SuperChain c{0u, 0u, kNoPos};
    for (uint32_t it = 0; it <= nreg; ++it) {
        const uint2 sg = seg_at(id);
        c.bytes += sg.x;
        ++c.segments;
        c.next = super_of(sg.y);
        if (c.next != kNoPos) break;
        id = min(sg.y, nreg);
    }
    return c;
}

// One thread orders the super nodes: s_sum[u] (bytes of u's chain) becomes the ordinal of its first
// segment on the block's chain, s_pos[u] the bytes before it; returns {segments, bytes} of the
// whole chain.  s_nc[u] = segments << 13 | next super (kSuperNone: none).
__device__ __forceinline__ uint2 order_supers(uint32_t last, uint32_t* s_sum, uint32_t* s_nc, uint32_t* s_pos) {
    uint32_t cur = last, bytes = 0, segs = 0;
    for (uint32_t it = 0; it <= last; ++it) {
        const uint32_t nc = s_nc[cur];
        s_pos[cur] = bytes;
        bytes += s_sum[cur];
        s_sum[cur] = segs;
        segs += nc >> 13;
        const uint32_t nx = nc & kSuperNone;
        if (nx == last || nx == kSuperNone) break;  // back at the delimiter row: the text is complete
        cur = nx;
    }
    return make_uint2(segs, bytes);
}

// The same by the whole workgroup when, as in every block the forward transform made, the super nodes
// form ONE chain from the delimiter's node through all of them: log2 rounds of pointer jumping give
// every node the bytes, segments and nodes from itself to the chain's end, and what lies before a node
// is the total minus that.  (One thread following some 500 links, each a dependent LDS access, was
// half the stitch kernel's time.)  Works in the three arrays themselves -- s_sum: bytes, s_pos:
// segments << 13 | nodes, s_nc: next node -- so a false return (the nodes do not form such a chain:
// hand-made blocks) leaves them overwritten: the caller fills them again and orders the nodes with
// order_supers.  All threads of the workgroup must call.
template <int NT, int K = 1>  // K: super nodes per thread (node u belongs to thread u % NT)
__device__ __forceinline__ bool order_supers_parallel(uint32_t last, uint32_t* s_sum, uint32_t* s_nc, uint32_t* s_pos,
                                                      uint2* whole) {
    if (last + 1u > (uint32_t)(NT * K) || last + 1u >= 8192u) return false;  // workgroup-uniform
    uint32_t bytes[K], segs[K], hops[K], nx[K];
    bool mine[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint32_t u = threadIdx.x + (uint32_t)(k * NT);
        mine[k] = u <= last;
        bytes[k] = segs[k] = 0;
        hops[k] = 1;
        nx[k] = kSuperNone;
        if (mine[k]) {
            bytes[k] = s_sum[u];
            segs[k] = s_nc[u] >> 13;
            nx[k] = s_nc[u] & kSuperNone;
            if (nx[k] == last) nx[k] = kSuperNone;  // back at the delimiter row: the text is complete
            s_pos[u] = (segs[k] << 13) | 1u;        // segments (at most 2^17 per block) above the node count (below 2^13)
            s_nc[u] = nx[k];
        }
    }
    __syncthreads();
    for (uint32_t span = 1; span <= last; span <<= 1) {
        uint32_t b2[K], h2[K], n2[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            b2[k] = h2[k] = 0;
            n2[k] = kSuperNone;
            if (mine[k] && nx[k] != kSuperNone) {
                b2[k] = s_sum[nx[k]];
                h2[k] = s_pos[nx[k]];
                n2[k] = s_nc[nx[k]];
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (mine[k] && nx[k] != kSuperNone) {
                const uint32_t u = threadIdx.x + (uint32_t)(k * NT);
                bytes[k] += b2[k];
                segs[k] += h2[k] >> 13;
                hops[k] += h2[k] & 8191u;
                nx[k] = n2[k];
                s_sum[u] = bytes[k];
                s_pos[u] = (segs[k] << 13) | (hops[k] & 8191u);
                s_nc[u] = nx[k];
            }
        }
        __syncthreads();
    }
    // one chain through all nodes: everybody has reached the end, and the head counts them all
    bool done = true;
#pragma unroll
    for (int k = 0; k < K; ++k) done = done && (!mine[k] || nx[k] == kSuperNone);
    const uint32_t head_hops = s_pos[last] & 8191u;
    const bool ok = __syncthreads_and(done) && head_hops == last + 1u;
    const uint32_t tot_bytes = s_sum[last], tot_segs = s_pos[last] >> 13;
    __syncthreads();
    if (!ok) return false;
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (mine[k]) {
            const uint32_t u = threadIdx.x + (uint32_t)(k * NT);
            s_pos[u] = tot_bytes - bytes[k];   // bytes before this node's chain
            s_sum[u] = tot_segs - segs[k];     // ordinal of its first segment
        }
    *whole = make_uint2(tot_segs, tot_bytes);
    __syncthreads();
    return true;
}

template <int NT>
__global__ __launch_bounds__(NT) void bwt_inverse_stitch_kernel(InvArgs a) {
    // 256 threads: blocks of at most kStitchSmallSegs segments (stride 8: 1025 super nodes);
    // 1024 threads: up to kMaxSegments (stride 32: 4097; 1 MiB blocks get stride 16)
    constexpr uint32_t kSupers = NT >= 1024 ? kMaxSegments / 32u + 2u : (kStitchSmallSegs >> kMinSuperShift) + 2u;
    constexpr int K = (int)((kSupers + NT - 1) / NT);  // super nodes per thread in the parallel ordering
    static_assert(kSupers < kSuperNone, "next super node in 13 bits");
    __shared__ uint32_t s_sum[kSupers], s_nc[kSupers], s_pos[kSupers];
    const uint32_t b = blockIdx.x;
    if (a.mode[b] != kModeMulti) return;  // workgroup-uniform
    const uint32_t n = inv_block_len(a, b);
    const uint32_t nreg = regular_splitters(n, a.split_shift);
    const uint64_t sb = (uint64_t)b * a.seg_cap;
    const uint32_t sshift = super_shift(nreg, kSupers), smask = (1u << sshift) - 1u;
    const uint32_t last = (nreg + smask) >> sshift;  // super index of the delimiter's segment
    auto super_of = [&](uint32_t id) {  // super index of segment id, or kNoPos
        if (id >= nreg) return last;
        return (id & smask) == 0u ? id >> sshift : kNoPos;
    };
    auto seg_at = [&](uint32_t id) { return a.seg[sb + id]; };
    auto sum_chains = [&]() {
        for (uint32_t u = threadIdx.x; u <= last; u += NT) {
            const SuperChain c = follow_to_super(u == last ? nreg : u << sshift, nreg, seg_at, super_of);
            s_sum[u] = c.bytes;
            s_nc[u] = (c.segments << 13) | (c.next == kNoPos ? kSuperNone : c.next);
            s_pos[u] = kNoPos;
        }
        __syncthreads();
    };
    sum_chains();
    uint2 whole;
    const bool chained = order_supers_parallel<NT, K>(last, s_sum, s_nc, s_pos, &whole);  // blocks of up to NT * K super nodes
    if (!chained) sum_chains();  // workgroup-uniform
    if (threadIdx.x == 0) {
        if (!chained) whole = order_supers(last, s_sum, s_nc, s_pos);
        a.chain[b] = whole;
        a.lens[b] = whole.y < n ? whole.y : n - 1;
    }
    __syncthreads();
    for (uint32_t u = threadIdx.x; u <= last; u += NT) {
        uint32_t pos = s_pos[u];
        if (pos == kNoPos) continue;  // a chain the delimiter's chain never reaches
        uint32_t id = u == last ? nreg : u << sshift, ord = s_sum[u];
        for (uint32_t it = 0; it <= nreg; ++it) {
            const uint2 sg = a.seg[sb + id];
            if (ord < a.seg_cap) a.seg_rec[sb + ord] = make_uint2(id, pos);
            ++ord;
            pos += sg.x;
            if (super_of(sg.y) != kNoPos) break;
            id = min(sg.y, nreg);
        }
    }
}

// The same ordering with the block's segment records held in LDS: following the links through
// global memory is one random 64-byte transaction per segment and phase, in LDS it is a few
// cycles.  For blocks of at most kStitchLdsSegs segments; one workgroup per CU.
constexpr uint32_t kStitchLdsSegs = 16640;
constexpr uint32_t kLdsSuperStride = 16;  // measured: 8 is slower (three nodes per thread in every round of the ordering), 32 too (longer chains)
constexpr uint32_t kStitchLdsSupers = kStitchLdsSegs / kLdsSuperStride + 2u;

// Debug builds (-DBWT_STITCH_CLOCKS) add up thread 0's clock over the kernel's steps in the workspace
// header (64-bit words from byte 64), which bench.py prints when they are not zero.
#ifdef BWT_STITCH_CLOCKS
#define STITCH_CLK(p)                                                                                \
    do {                                                                                             \
        if (threadIdx.x == 0) {                                                                      \
            const uint64_t t_ = __builtin_amdgcn_s_memtime();                                        \
            atomicAdd(reinterpret_cast<unsigned long long*>(a.queue) + 8 + (p), (unsigned long long)(t_ - clk_)); \
            clk_ = t_;                                                                               \
        }                                                                                            \
    } while (0)
#else
#define STITCH_CLK(p)
#endif

template <int NT>
__global__ __launch_bounds__(NT) void bwt_inverse_stitch_lds_kernel(InvArgs a) {
#ifdef BWT_STITCH_CLOCKS
    uint64_t clk_ = __builtin_amdgcn_s_memtime();
#endif
    __shared__ uint2 s_seg[kStitchLdsSegs];
    __shared__ uint32_t s_sum[kStitchLdsSupers], s_nc[kStitchLdsSupers], s_pos[kStitchLdsSupers];
    for (uint32_t b = blockIdx.x; b < a.nblocks; b += gridDim.x) {
        if (a.mode[b] != kModeMulti) continue;  // workgroup-uniform
        const uint32_t n = inv_block_len(a, b);
        const uint32_t nreg = regular_splitters(n, a.split_shift);
        const uint64_t sb = (uint64_t)b * a.seg_cap;
        const uint32_t last = (nreg + kLdsSuperStride - 1u) / kLdsSuperStride;
        auto super_of = [&](uint32_t id) {
            if (id >= nreg) return last;
            return (id % kLdsSuperStride) == 0u ? id / kLdsSuperStride : kNoPos;
        };
        auto seg_at = [&](uint32_t id) { return s_seg[id]; };
        STITCH_CLK(0);
        for (uint32_t i = threadIdx.x; i <= nreg; i += NT) s_seg[i] = a.seg[sb + i];
        __syncthreads();
        STITCH_CLK(1);
        auto sum_chains = [&]() {
            for (uint32_t u = threadIdx.x; u <= last; u += NT) {
                const SuperChain c = follow_to_super(u == last ? nreg : u * kLdsSuperStride, nreg, seg_at, super_of);
                s_sum[u] = c.bytes;
                s_nc[u] = (c.segments << 13) | (c.next == kNoPos ? kSuperNone : c.next);
                s_pos[u] = kNoPos;
            }
            __syncthreads();
        };
        sum_chains();
        STITCH_CLK(2);
        uint2 whole;
        constexpr int K = (int)((kStitchLdsSupers + NT - 1) / NT);
        const bool chained = order_supers_parallel<NT, K>(last, s_sum, s_nc, s_pos, &whole);
        if (!chained) sum_chains();  // workgroup-uniform
        if (threadIdx.x == 0) {
            if (!chained) whole = order_supers(last, s_sum, s_nc, s_pos);
            a.chain[b] = whole;
            a.lens[b] = whole.y < n ? whole.y : n - 1;
        }
        __syncthreads();
        STITCH_CLK(3);
        for (uint32_t u = threadIdx.x; u <= last; u += NT) {
            uint32_t pos = s_pos[u];
            if (pos == kNoPos) continue;
            uint32_t id = u == last ? nreg : u * kLdsSuperStride, ord = s_sum[u];
            for (uint32_t it = 0; it <= nreg; ++it) {
                const uint2 sg = s_seg[id];
                if (ord < a.seg_cap) a.seg_rec[sb + ord] = make_uint2(id, pos);
                ++ord;
                pos += sg.x;
                if (super_of(sg.y) != kNoPos) break;
                id = min(sg.y, nreg);
            }
        }
        __syncthreads();
        STITCH_CLK(4);
    }
}

// Moves every chained segment from its staging slot (and pool chunks) to its place: the bytes
// were produced back to front, so byte t of the segment lands t + 1 bytes below its upper end.
// Lane k takes the k-th segment ON THE CHAIN (seg_rec), not segment k, so the 256 lanes of a
// workgroup own one contiguous stretch of the text (about 4 KiB).
//
// place_segment_direct writes a segment straight to the text: an unaligned source dword (funnel
// shift) byte-swapped is an aligned destination dword; 16-byte stores where aligned, dword / byte
// stores at the ragged ends.  With every lane doing that, the pieces of a 64-byte line arrive from
// different lanes at different times and the pass wrote 2.6 times its payload to HBM.  So the
// workgroup first assembles its stretch in LDS (in walk order, which is the text reversed: a
// plain copy of every segment to its offset) and then writes it out as whole aligned 16-byte
// groups, each group byte-reversed; only the stretch's two ragged ends go byte by byte, and
// segments that reach beyond the LDS window (kPlaceWindow bytes: four times the average stretch)
// are written directly as before.
__device__ __forceinline__ void place_segment_direct(const InvArgs& a, uint32_t b, uint32_t j, uint32_t pos,
                                                     uint32_t len, uint32_t n, uint64_t sb) {
    const uint64_t pb = (uint64_t)b * a.chunk_cap;
    const uint8_t* src = stage_group(a, b, j);
    uint32_t gstride = kStageGroupStride;
    uint32_t c = 0;
    BackWriter bw;
    bw.init(a.out + (uint64_t)b * a.out_stride + (n - 1) - pos);
    // Source byte t goes to address end - 1 - t.  The bytes down to the first dword boundary go
    // one by one; from there every aligned destination dword is four consecutive source bytes
    // reversed: an unaligned source dword (funnel shift over two aligned ones) byte-swapped.
    const uint32_t head = min(len, (uint32_t)(reinterpret_cast<uintptr_t>(bw.addr) & 3u));
    const uint32_t nd = (len - head) >> 2;           // whole destination dwords
    const uint32_t tail = len - head - 4u * nd;      // bytes below the last whole dword
    const uint32_t last_group = (nd + 1u) >> 2;      // source dwords 0 .. nd+1 are consulted
    uint32_t prev = 0;
    for (uint32_t g = 0; g <= last_group; ++g) {
        if (g && (g & ((a.slot / 16u) - 1u)) == 0u && 16u * g < len) {  // the next chunk of a long segment
            c = g == (a.slot / 16u) ? a.seg_ovf[sb + j] : a.chunk_next[pb + c];
            if (c >= a.chunk_cap) break;  // never
            src = a.pool + (pb + c) * a.slot;
            gstride = 16u;
        }
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (16u * g < len) v = *reinterpret_cast<const uint4*>(src + (g & ((a.slot / 16u) - 1u)) * gstride);
        if (g == 0) {
            for (uint32_t k = 0; k < head; ++k) bw.push((v.x >> (8u * k)) & 255u);
        }
        const uint32_t lo[4] = {prev, v.x, v.y, v.z}, hi[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t d = 4u * g + (uint32_t)k - 1u;  // destination dword this pair yields (g = 0, k = 0: none)
            if (d <= nd && !(g == 0 && k == 0)) {
                uint32_t w = lo[k];  // source bytes head + 4 d ..
                if (head == 1) w = __builtin_amdgcn_alignbyte(hi[k], lo[k], 1);
                else if (head == 2) w = __builtin_amdgcn_alignbyte(hi[k], lo[k], 2);
                else if (head == 3) w = __builtin_amdgcn_alignbyte(hi[k], lo[k], 3);
                if (d < nd) {
                    bw.push_dword(__builtin_bswap32(w));
                } else {
                    for (uint32_t u = 0; u < tail; ++u) bw.push((w >> (8u * u)) & 255u);
                }
            }
        }
        prev = v.w;
    }
    bw.flush();
}

constexpr uint32_t kPlaceWindow = 64u * kSegThreads;  // bytes of a workgroup's stretch that are assembled in LDS

// Workgroup w of the place pass (s_buf: kPlaceWindow / 4 + 8 words, walk order with 16 bytes of slack for
// the alignment shift; s_wend: one word per wave).
__device__ __forceinline__ void place_unit(const InvArgs& a, uint32_t w, uint32_t* s_buf, uint32_t* s_wend) {
#ifdef BWT_STITCH_CLOCKS
    uint64_t clk_ = __builtin_amdgcn_s_memtime();
#endif
    // Everything the workgroup needs to know about its stretch is loaded at once and nothing is
    // exchanged before the copy: a workgroup's time is memory round trips in a row, and with the
    // block's mode, its chain, the lane's records and lane 0's position fetched one after the other
    // and two barriers to pass the latter round there were five of them before the first staged byte.
    const uint32_t wgs = (a.seg_cap + (uint32_t)kSegThreads - 1u) / (uint32_t)kSegThreads;
    const uint32_t b = ((w >> 3) / wgs) * 8u + (w & 7u);  // as in segment_of_lane
    const uint32_t k0 = ((w >> 3) % wgs) * (uint32_t)kSegThreads;  // the workgroup's first segment on the chain
    const uint32_t k = k0 + threadIdx.x;
    if (b >= a.nblocks || k0 >= a.seg_cap) return;  // workgroup-uniform
    // a block the in-LDS inverse has finished: out before anything else is fetched (a whole batch of such
    // blocks still dispatches this grid)
    if (a.packed_first && __ldg(&a.mode[b]) == kModeDone) return;
    const uint64_t sb = (uint64_t)b * a.seg_cap;
    const uint32_t mode = a.mode[b];
    const uint2 whole = a.chain[b];
    const uint32_t p0 = a.seg_rec[sb + k0].y;  // walk-order offset of the stretch's first byte (its LAST byte in the text)
    uint2 rec = make_uint2(0u, 0u);
    uint32_t pos1 = 0;
    if (k < a.seg_cap) rec = a.seg_rec[sb + k];
    if (k + 1u < a.seg_cap) pos1 = a.seg_rec[sb + k + 1u].y;
    if (mode != kModeMulti || k0 >= whole.x) return;  // workgroup-uniform: segments off the chain have no place
    const uint32_t n = inv_block_len(a, b);
    const bool mine = k < whole.x && k < a.seg_cap;
    uint32_t j = 0, pos = p0, len = 0;
    if (mine) {
        j = rec.x;
        pos = rec.y;
        len = (k + 1u < whole.x ? pos1 : whole.y) - pos;
        if ((uint64_t)pos + len > (uint64_t)(n - 1) || pos < p0) len = 0;  // never: the chain is at most n-1 bytes long
    }
    STITCH_CLK(5);
    // Walk-order offset o (from p0) <-> text address top - o, top = address of the stretch's last text byte.
    uint8_t* const top = a.out + (uint64_t)b * a.out_stride + (n - 2u) - p0;
    const uint32_t shift = (uint32_t)((16u - ((uint32_t)reinterpret_cast<uintptr_t>(top + 1) & 15u)) & 15u);  // LDS index = o + shift
    uint8_t* const eup = top + 1 + shift;  // 16-byte aligned: LDS index i <-> address eup - 1 - i
    const bool far = mine && (pos - p0) + len > kPlaceWindow;  // reaches beyond the window: written directly
    {
        // walk-order bytes [0, span) of the stretch go through LDS: up to the end of the last segment that
        // fits the window (positions grow with k, so the segments that do not fit are the last ones)
        uint32_t wend = (!far && len) ? pos - p0 + len : 0u;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) wend = max(wend, (uint32_t)__shfl_xor((int)wend, o));
        if ((threadIdx.x & (kWave - 1)) == 0) s_wend[threadIdx.x / kWave] = wend;
    }
    STITCH_CLK(6);
    if (far) {
        place_segment_direct(a, b, j, pos, len, n, sb);
    } else if (len) {
        // plain copy of the staged bytes to LDS index pos - p0 + shift ..: bytes up to the first dword
        // boundary, whole dwords (funnel shift over the aligned source), bytes at the end
        uint8_t* const lb = reinterpret_cast<uint8_t*>(s_buf);
        const uint32_t d0 = pos - p0 + shift;
        const uint32_t head = min(len, (4u - (d0 & 3u)) & 3u);
        const uint32_t nd = (len - head) >> 2, tail = len - head - 4u * nd;
        const uint64_t pb = (uint64_t)b * a.chunk_cap;
        const uint8_t* src = stage_group(a, b, j);
        uint32_t gstride = kStageGroupStride;
        uint32_t c = 0, prev = 0;
        const uint32_t last_group = (nd + 1u) >> 2;      // source dwords 0 .. nd+1 are consulted
        uint32_t* const ld = s_buf + ((d0 + head) >> 2);
        // The slot's first four groups (every slot has at least four) and the link to the first pool
        // chunk are loaded at once: fetched one after the other as the copy went along, a workgroup
        // waited for as many memory round trips in a row as its longest segment has groups.
        uint4 pre[4];
#pragma unroll
        for (uint32_t g = 0; g < 4u; ++g) {
            pre[g] = make_uint4(0u, 0u, 0u, 0u);
            if (16u * g < len) pre[g] = *reinterpret_cast<const uint4*>(src + g * kStageGroupStride);
        }
        uint32_t ovf = 0;
        if (len > a.slot) ovf = a.seg_ovf[sb + j];
        auto emit = [&](uint32_t g, const uint4 v) {  // source group g = walk-order bytes 16 g .. 16 g + 15 of the segment
            if (g == 0) {
                for (uint32_t t = 0; t < head; ++t) lb[d0 + t] = (uint8_t)(v.x >> (8u * t));
            }
            const uint32_t lo[4] = {prev, v.x, v.y, v.z}, hi[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t d = 4u * g + (uint32_t)q - 1u;  // destination dword this pair yields (g = 0, q = 0: none)
                if (d <= nd && !(g == 0 && q == 0)) {
                    uint32_t w = lo[q];  // source bytes head + 4 d ..
                    if (head == 1) w = __builtin_amdgcn_alignbyte(hi[q], lo[q], 1);
                    else if (head == 2) w = __builtin_amdgcn_alignbyte(hi[q], lo[q], 2);
                    else if (head == 3) w = __builtin_amdgcn_alignbyte(hi[q], lo[q], 3);
                    if (d < nd) {
                        ld[d] = w;
                    } else {
                        for (uint32_t t = 0; t < tail; ++t) lb[d0 + head + 4u * nd + t] = (uint8_t)(w >> (8u * t));
                    }
                }
            }
            prev = v.w;
        };
#pragma unroll
        for (uint32_t g = 0; g < 4u; ++g)
            if (g <= last_group) emit(g, pre[g]);
        for (uint32_t g = 4; g <= last_group; ++g) {
            if ((g & ((a.slot / 16u) - 1u)) == 0u && 16u * g < len) {  // the next chunk of a long segment
                c = g == (a.slot / 16u) ? ovf : a.chunk_next[pb + c];
                if (c >= a.chunk_cap) break;  // never
                src = a.pool + (pb + c) * a.slot;
                gstride = 16u;
            }
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (16u * g < len) v = *reinterpret_cast<const uint4*>(src + (g & ((a.slot / 16u) - 1u)) * gstride);
            emit(g, v);
        }
    }
    __syncthreads();
    STITCH_CLK(7);
    uint32_t span = 0;
#pragma unroll
    for (int w = 0; w < kSegThreads / kWave; ++w) span = max(span, s_wend[w]);
    // write the stretch out: LDS indices [shift, shift + span) <-> addresses eup - 1 - i.  Group c = indices
    // 16 c .. 16 c + 15 is one aligned 16-byte store, its bytes reversed.
    if (span == 0u) return;
    const uint32_t lo_i = shift, hi_i = shift + span;  // valid LDS indices
    const uint32_t groups = (hi_i + 15u) >> 4;
    const uint8_t* const lb = reinterpret_cast<const uint8_t*>(s_buf);
    for (uint32_t c = threadIdx.x; c < groups; c += kSegThreads) {
        uint8_t* const dst = eup - 16u * (c + 1u);
        if (16u * c >= lo_i && 16u * c + 16u <= hi_i) {
            const uint4 v = *reinterpret_cast<const uint4*>(s_buf + 4u * c);
            *reinterpret_cast<uint4*>(dst) = make_uint4(__builtin_bswap32(v.w), __builtin_bswap32(v.z),
                                                        __builtin_bswap32(v.y), __builtin_bswap32(v.x));
        } else {
            for (uint32_t t = 0; t < 16u; ++t) {
                const uint32_t i = 16u * c + t;
                if (i >= lo_i && i < hi_i) dst[15u - t] = lb[i];
            }
        }
    }
    STITCH_CLK(8);
}
// `units` workgroups' worth of work: one workgroup per unit, or a capped grid striding over them (see
// bwt_inverse_walk_segments_kernel).
template <bool STRIDED>
__global__ __launch_bounds__(kSegThreads) void bwt_inverse_place_kernel(InvArgs a, uint32_t units) {
    __shared__ uint32_t s_buf[kPlaceWindow / 4 + 8];
    __shared__ uint32_t s_wend[kSegThreads / kWave];
    if (STRIDED) {
        for (uint32_t w = blockIdx.x; w < units; w += gridDim.x) {
            place_unit(a, w, s_buf, s_wend);
            __syncthreads();  // the next unit reuses the buffers
        }
    } else {
        place_unit(a, blockIdx.x, s_buf, s_wend);
    }
}

// What the in-LDS inverse left to the kernels above, for the host to look at before its next batch
// (queue[3] counts the blocks that kernel has finished).
__global__ void bwt_inverse_left_kernel(InvArgs a) {
    __hip_atomic_store(a.left_hint, a.nblocks - a.queue[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- origin-pointer blocks: the last step --------------------------------------------------------
// The walk kernels have produced, below out + b * out_stride + n - 1, the bytes of the rows that follow
// orig_ptr[b] on its cycle of the LF permutation: c - 1 bytes, where c is the cycle's length
// (lens[b] = c - 1).  The text's last byte is the origin row's own (reference
// util/bzip2_bwt_utils.cpp:243-247 emits it first and reverses), and when the text is a shorter
// string repeated (c < n: the rotations tie and the LF permutation falls into n / c cycles) the walk of
// the reference goes round the cycle n / c times: the text is its last c bytes repeated.
__global__ __launch_bounds__(256) void bwt_inverse_origin_finish_kernel(InvArgs a) {
    const uint32_t b = blockIdx.x;
    const uint32_t n = inv_block_len(a, b);
    const uint32_t have = a.lens[b];
    if (a.mode[b] == kModeNone || have >= n) {  // workgroup-uniform
        if (threadIdx.x == 0) a.lens[b] = BWTGPU_LEN_NO_DELIMITER;
        return;
    }
    const uint8_t* __restrict__ L = a.in + (uint64_t)b * ((uint64_t)a.block_size + 1u);
    uint8_t* const o = a.out + (uint64_t)b * a.out_stride;
    const uint32_t c = have + 1u;
    if (threadIdx.x == 0) o[n - 1u] = L[a.drow[b]];
    __syncthreads();
    const uint8_t* __restrict__ cyc = o + (n - c);
    const uint32_t shift = (n - c) % c;  // position i < n - c equals position n - c + ((i - (n - c)) mod c)
    for (uint32_t i = threadIdx.x; i < n - c; i += blockDim.x) o[i] = cyc[(i % c + c - shift) % c];
    __syncthreads();
    if (threadIdx.x == 0) a.lens[b] = n;
}

}  // namespace bwtgpu
