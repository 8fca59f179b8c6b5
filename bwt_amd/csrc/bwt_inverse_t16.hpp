// bwt_inverse_t16.hpp -- inverse BWT of blocks of at most 64 KiB, any alphabet, inside LDS.
//
// The general inverse (bwt_inverse.hpp) keeps the LF table T[i] = C[L[i]] + occ[i] (the numbers the
// reference builds as arrays, src/inverse_bwt.cpp:23-45) in device memory, 4 bytes per row, and its walk
// pays one dependent load from the L2 per output byte.  A block of at most 65536 text bytes has at most
// 65537 rows: with the delimiter's row taken out (the walk starts from it and ends on it, it is never
// stepped FROM), the other rows are numbered 0 .. 65535 and the table is 65536 entries of 16 bits --
// 128 KB, which fits the LDS of a compute unit.  A walk step then is one 2-byte LDS read.
//
// The byte a step produces is L[row], which is not kept (another 64 KB); it is recovered from where the
// step leads: LF(row) lies in the stretch of rows whose first column holds L[row], so the symbol is the
// c with C[c] <= LF(row) < C[c + 1] -- found from a table of the first-column symbol at every 64th row
// and the stretch ends C1[] (one or two more LDS reads, off the chain of dependent ones).  Only the
// second walk, which writes bytes, needs that.
//
// One workgroup of 1024 lanes per block (one per compute unit), everything for a block in one go --
// the same structure as the inverse of small-alphabet blocks (bwt_inverse_packed.hpp):
//   count    byte histogram per wave chunk, the delimiter's row (exactly one, else the block is left
//            to the general kernels: its mode stays != kModeDone)
//   scan     C[] in unsigned byte order, the delimiter counted where its value falls
//            (src/inverse_bwt.cpp:35-45), per-wave start offsets, the symbol tables
//   table    stable per-symbol ranking by wave ballots -> T
//   walk 1   the chain is cut where q * A mod 2^16 < ids (q = the row's number; the image is the segment's
//            id) and at its end; lanes take segment ids until none is left and record {length, segment
//            run into}
//   order    pointer jumping over the records: bytes from a segment's start to the text's start
//   walk 2   the same walks again, now writing bytes (back to front, a dword every four steps), the
//            segments handed out stretch by stretch of the text, the long ones of a stretch first
// The result is the chain the reference walks (src/inverse_bwt.cpp:48-62), byte for byte; a hand-made
// block whose LF permutation has several cycles yields the cycle through the delimiter row, like the
// reference, with the shorter length in lens[].
#pragma once
#include "bwt_inverse.hpp"

namespace bwtgpu {

constexpr int kT16Threads = 1024;
constexpr int kT16Waves = kT16Threads / kWave;
constexpr uint32_t kT16MaxRows = 65536u + 1u;   // block_size + 1
constexpr uint32_t kT16MinRows = 49152u;        // shorter blocks: the table kernels run several per compute unit
constexpr uint32_t kT16Ids = 7168u;             // segment ids (LDS holds one record each); id kT16Ids = the chain's first segment
constexpr uint32_t kT16End = 8191u;             // "runs into the chain's end"
constexpr uint32_t kT16NextBits = 13;
constexpr uint32_t kT16Stretches = 8;
constexpr uint32_t kT16Buckets = 2 * kT16Stretches;
constexpr uint32_t kT16Long = 16;               // walk 2 hands out the segments of at least this many bytes first
constexpr uint32_t kT16GrabBatch = 16;
constexpr uint32_t kT16IdChunk = 64;
constexpr uint32_t kT16Groups = 65536u / 64u;

struct T16Shared {
    uint16_t T[65536];             // by row number: the number of the row LF leads to
    union {
        uint32_t seg[kT16Ids + 1]; // walk 1, by segment id: bytes << 13 | id of the segment run into (kT16End: none);
                                   // walk 2: the segments on the chain in the order they are handed out: bytes to the text's start << 13 | id
        uint32_t hist[kT16Waves * 256];  // building the table: byte counts per wave chunk, then the chunks' start offsets
    } u;
    uint32_t C1[256];              // row number at which the first-column stretch of symbol c ends
    uint8_t gsym[kT16Groups + 8];  // first-column symbol at row number 64 g
    uint8_t code[256];             // byte value -> its index among the values that occur
    uint32_t wsum[kT16Waves];
    uint32_t wdrow[kT16Waves];
    uint32_t blk, ctr, ndelim, bad, special, sigma;
    uint32_t bucket_fill[kT16Buckets], bucket_base[kT16Buckets + 1];
};
static_assert(sizeof(T16Shared) <= 160u * 1024u, "one workgroup's LDS");
struct __attribute__((packed, aligned(1))) T16Bytes4 {
    uint32_t x;
};

__device__ __forceinline__ uint32_t t16_seg(uint32_t bytes, uint32_t next) { return (bytes << kT16NextBits) | next; }

// Both walks.  EMIT = false: lengths and links into sh.u.seg.  EMIT = true: the bytes go to
// out_end[-(total - togo) - 1] downwards, togo from sh.u.seg.  A lane's state is the number q of the row it
// stands on.  The loop is flat -- every iteration is one step of every walking lane -- and lanes whose
// segment has ended wait until kT16GrabBatch of the wave's lanes do, so that the bookkeeping is paid
// once for several lanes; ids come from a range of kT16IdChunk that the wave owns.
// special: the row whose step leads to the delimiter's row (the chain's end); first_sym: the byte that step
// produces (the text's first).  Returns the iterations the wave has made.
template <bool EMIT>
__device__ __forceinline__ uint32_t t16_walk(T16Shared& sh, uint32_t m, bool empty, uint32_t q0, uint32_t special,
                                             uint32_t first_sym, uint32_t mul, uint32_t inv, uint32_t mask, uint32_t ids,
                                             uint8_t* out_end, uint32_t total, uint32_t count) {
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    uint32_t id = kNone;  // the segment the lane walks, or has walked and not yet put on record
    uint32_t q = 0, len = 0, hit_next = 0;
    uint32_t walking = 0;  // lane mask: 0 or all ones
    uint32_t acc_lo = 0, acc_hi = 0, cnt = 0;  // walk 2: bytes not yet stored (oldest in the highest bits), how many
    uint8_t* top = out_end;                    // ... and one past the address of the oldest
    uint32_t wnext = 0, wend = 0;  // the wave's ids (wave-uniform)
    bool more = true;              // the workgroup's counter may have ids left (wave-uniform)
    const uint32_t id_end = EMIT ? count : ids;
    const uint32_t it_cap = 4u * m + 65536u;  // never reached for consistent tables; keeps a workgroup from spinning
    if (threadIdx.x == 0) {  // the chain's first segment starts where the delimiter row leads
        const uint32_t img = __umul24(q0, mul) & mask;
        if (empty || img < ids) {  // ... which is a cut itself: an empty segment
            if (!EMIT) sh.u.seg[kT16Ids] = t16_seg(0u, empty ? kT16End : img);
        } else {
            walking = ~0u;
            id = kT16Ids;
            q = q0;
        }
    }
    auto store_dword = [&]() {  // the four oldest bytes
        const uint32_t sh8 = (cnt - 4u) * 8u;
        const uint64_t acc = ((uint64_t)acc_hi << 32) | acc_lo;
        reinterpret_cast<T16Bytes4*>(top - 4)->x = (uint32_t)(acc >> sh8);
        top -= 4;
        cnt -= 4u;
    };
    for (uint32_t it = 0;; ++it) {
        const uint64_t idle = __ballot(walking == 0u);
        const uint32_t nidle = (uint32_t)__popcll(idle);
        if (nidle == (uint32_t)kWave || (more && nidle >= kT16GrabBatch)) {  // wave-uniform
            if (walking == 0u && id != kNone) {  // a finished segment
                if (EMIT) {
                    if (cnt >= 4u) store_dword();
                    for (uint32_t k = 0; k < cnt; ++k) top[-1 - (int)k] = (uint8_t)(acc_lo >> (8u * (cnt - 1u - k)));
                    cnt = 0;
                } else {
                    sh.u.seg[id] = t16_seg(len, hit_next);
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
                    const uint32_t word = sh.u.seg[cand];
                    const uint32_t togo = word >> kT16NextBits;
                    ok = togo != 0u && togo <= total;
                    top = out_end - (total - togo);
                    cand = word & kT16End;
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
                if (lane_id() == 0) v = atomicAdd(&sh.ctr, kT16IdChunk);
                wnext = wave_read(v, 0);
                wend = min(wnext + kT16IdChunk, id_end);
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
            const uint32_t nq = sh.T[q];
            const bool at_end = q == special;
            ++len;
            if (EMIT) {
                uint32_t c = sh.gsym[nq >> 6];
                c += nq >= sh.C1[c] ? 1u : 0u;
                while (nq >= sh.C1[c]) ++c;  // (stretches shorter than 64 rows: rare symbols)
                const uint32_t byte = at_end ? first_sym : c;
                acc_hi = __builtin_amdgcn_alignbit(acc_hi, acc_lo, 24);
                acc_lo = (acc_lo << 8) | byte;
                ++cnt;
            }
            q = nq;
            const uint32_t img = __umul24(q, mul) & mask;
            const bool cut = img < ids;
            hit_next = at_end ? kT16End : img;
            walking = (at_end | cut) ? 0u : ~0u;
        }
        if (EMIT && (it & 3u) == 3u && cnt >= 4u) store_dword();  // (it & 3) is wave-uniform: a lane holds at most 7 bytes
    }
}

// Per-phase time accounting (100 MHz constant clock) of thread 0, summed over workgroups into the u64 words
// from queue + 64: 0 count, 1 scan, 2 table, 3 walk 1, 4 order, 5 walk 2, 6 blocks finished.
struct T16Clock {
    unsigned long long t;
    unsigned long long* acc;
    __device__ __forceinline__ void start(uint32_t* queue) {
        acc = reinterpret_cast<unsigned long long*>(queue + 64);
        t = wall_clock64();
    }
    __device__ __forceinline__ void lap(int phase) {
        if (threadIdx.x == 0) {
            const unsigned long long now = wall_clock64();
            atomicAdd(&acc[phase], now - t);
            t = now;
        }
    }
};

__global__ __launch_bounds__(kT16Threads) void bwt_inverse_t16_kernel(InvArgs a) {
    __shared__ T16Shared sh;
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const uint64_t lt = lanemask_lt();
    T16Clock clk;
    clk.start(a.queue);
    for (;;) {
        __syncthreads();  // everybody is done with the block before
        if (tid == 0) {
            sh.blk = atomicAdd(&a.queue[4], 1u);
            sh.ndelim = 0;
            sh.bad = 0;
            sh.special = kNoRow;
        }
        for (int i = tid; i < kT16Waves * 256; i += kT16Threads) sh.u.hist[i] = 0;
        if (tid < (int)kT16Buckets) sh.bucket_fill[tid] = 0;
        __syncthreads();
        const uint32_t b = sh.blk;
        if (b >= a.nblocks) break;
        if (a.mode[b] == kModeDone) continue;  // workgroup-uniform: inverted already (a small-alphabet block)
        const uint32_t n = inv_block_len(a, b);
        if (n < kT16MinRows || n > kT16MaxRows) continue;  // workgroup-uniform
        const uint8_t* __restrict__ L = a.in + (uint64_t)b * ((uint64_t)a.block_size + 1u);
        clk.lap(7);

        // ---- count: histogram per wave chunk, the delimiter's row
        Chunking ck;
        ck.init(n, kT16Waves);
        const uint32_t lo = (uint32_t)w * ck.chunk, hi = min(lo + ck.chunk, n);
        {
            uint32_t lastp = 0;  // row + 1 of the last delimiter this lane has seen
            uint32_t* myh = sh.u.hist + w * 256;
            auto take = [&](uint32_t c, uint32_t row) {
                atomicAdd(&myh[c], 1u);
                if (c == a.delim) lastp = max(lastp, row + 1u);
            };
            if (lo < hi) {
                constexpr int U = 4;
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
            lastp = wave_max(lastp);
            if (lane == 0) sh.wdrow[w] = lastp ? lastp - 1u : kNoRow;
        }
        __syncthreads();
        clk.lap(0);
        uint32_t drow = kNoRow;  // rows grow with the wave index, so the last hit wins
#pragma unroll
        for (int i = 0; i < kT16Waves; ++i)
            if (sh.wdrow[i] != kNoRow) drow = sh.wdrow[i];

        // ---- scan: C[] and per-wave start offsets in (symbol major, wave minor) order, the symbol tables
        uint32_t tot = 0, cstart = 0;
        {
            if (tid < 256)
                for (int i = 0; i < kT16Waves; ++i) tot += sh.u.hist[i * 256 + tid];
            if (tid == (int)a.delim) sh.ndelim = tot;
            uint32_t total;
            cstart = block_exclusive_scan<kT16Threads>(tot, sh.wsum, total);  // C[tid]
            if (tid < 256) {
                uint32_t run = cstart;
                for (int i = 0; i < kT16Waves; ++i) {
                    const uint32_t v = sh.u.hist[i * 256 + tid];
                    sh.u.hist[i * 256 + tid] = run;
                    run += v;
                }
            }
            uint32_t sigma;
            const uint32_t below = block_exclusive_scan<kT16Threads>(tid < 256 && tot ? 1u : 0u, sh.wsum, sigma);
            if (tid < 256) sh.code[tid] = (uint8_t)below;
            if (tid == 0) sh.sigma = sigma;
        }
        __syncthreads();
        if (sh.ndelim != 1u || drow == kNoRow) continue;  // workgroup-uniform: left to the general kernels
        const int cbits = max(1, bit_length(sh.sigma - 1u));
        const uint32_t m = n - 1u;  // numbered rows
        if (tid < 256) {
            // the stretch of first-column rows of symbol tid, as row numbers (the delimiter's row of the LAST
            // column has no number: the rows above it count one less)
            const uint32_t s0 = cstart - (cstart > drow ? 1u : 0u);
            const uint32_t e = cstart + tot, s1 = e - (e > drow ? 1u : 0u);
            sh.C1[tid] = s1;
            for (uint32_t g = (s0 + 63u) >> 6; (g << 6) < s1; ++g) sh.gsym[g] = (uint8_t)tid;
        }
        // where the delimiter row leads (the delimiter's own place in the first column), and the byte of
        // the step that closes the chain: the first-column symbol of the delimiter's row
        __shared__ uint32_t s_row0, s_first;
        if (tid < 256 && tot) {
            if (tid == (int)a.delim) s_row0 = cstart;
            if (cstart <= drow && drow < cstart + tot) s_first = (uint32_t)tid;
        }

        clk.lap(1);
        // ---- table: T[number of row k] = number of the row LF(k)
        {
            constexpr int U = 8;
            uint32_t* myoff = sh.u.hist + w * 256;
            for (uint32_t base0 = lo; base0 < hi; base0 += U * kWave) {
                uint32_t cs[U];
#pragma unroll
                for (int u = 0; u < U; ++u) cs[u] = L[min(base0 + (uint32_t)(u * kWave + lane), n - 1u)];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t base = base0 + (uint32_t)(u * kWave);
                    if (base < hi) {  // wave-uniform
                        const uint32_t k = base + (uint32_t)lane;
                        const bool valid = k < hi;
                        const uint32_t c = cs[u];
                        const uint64_t peers = match_digit_bits((uint32_t)sh.code[c], valid, cbits);
                        const uint32_t rk = __popcll(peers & lt);
                        const uint32_t b0 = myoff[c];
                        compiler_fence();  // every lane has read its offset before the leaders advance it
                        if (valid && rk == 0) myoff[c] = b0 + (uint32_t)__popcll(peers);
                        compiler_fence();
                        if (valid && k != drow) {
                            const uint32_t lf = b0 + rk;
                            const uint32_t q = k - (k > drow ? 1u : 0u);
                            if (lf == drow) sh.special = q;  // (its entry is never followed; any row number will do)
                            sh.T[q] = lf == drow ? (uint16_t)0 : (uint16_t)(lf - (lf > drow ? 1u : 0u));
                        }
                    }
                }
            }
        }
        __syncthreads();
        clk.lap(2);
        const uint32_t row0 = s_row0, first_sym = s_first, special = sh.special;
        const bool empty = row0 == drow;  // the delimiter row leads to itself: no text
        const uint32_t q0 = row0 - (row0 > drow ? 1u : 0u);

        // ---- cuts: row number q -> q * mul mod 2^kbits (2^kbits >= m), a cut where the image is below `ids`
        const uint32_t kbits = (uint32_t)bit_length(m - 1u);
        const uint32_t mask = (1u << kbits) - 1u;
        const uint32_t mul = ((uint32_t)(0.6180339887498949 * (double)(1u << kbits)) | 1u) & mask;
        uint32_t inv = mul;  // Newton: inverse of an odd number modulo 2^32, five steps from 3 correct bits
#pragma unroll
        for (int i = 0; i < 5; ++i) inv *= 2u - mul * inv;
        inv &= mask;
        const uint32_t ids = min(kT16Ids, (1u << kbits) >> 3);
        for (uint32_t i = (uint32_t)tid; i <= kT16Ids; i += kT16Threads) sh.u.seg[i] = t16_seg(0u, kT16End);  // (the offsets are spent)
        if (tid == 0) sh.ctr = 0;
        __syncthreads();
        uint8_t* const out_end = a.out + (uint64_t)b * a.out_stride + m;  // one past the text's last byte

        t16_walk<false>(sh, m, empty, q0, special, first_sym, mul, inv, mask, ids, out_end, 0u, 0u);
        __syncthreads();
        clk.lap(3);
        if (sh.bad) continue;  // workgroup-uniform; never

        // ---- order: every record becomes bytes from its segment's start to the text's start
        uint32_t total;
        {
            constexpr int E = (kT16Ids + 1 + kT16Threads - 1) / kT16Threads;
            uint32_t own[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint32_t i = (uint32_t)tid + (uint32_t)e * kT16Threads;
                own[e] = i <= kT16Ids ? sh.u.seg[i] >> kT16NextBits : 0u;
            }
            for (int round = 0; round < (int)kT16NextBits; ++round) {
                uint32_t nv[E];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const uint32_t i = (uint32_t)tid + (uint32_t)e * kT16Threads;
                    nv[e] = 0;
                    if (i <= kT16Ids) {
                        const uint32_t word = sh.u.seg[i];
                        const uint32_t next = word & kT16End;
                        nv[e] = word;
                        if (next != kT16End) {
                            const uint32_t w2 = sh.u.seg[next];
                            nv[e] = t16_seg((word >> kT16NextBits) + (w2 >> kT16NextBits), w2 & kT16End);
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const uint32_t i = (uint32_t)tid + (uint32_t)e * kT16Threads;
                    if (i <= kT16Ids) sh.u.seg[i] = nv[e];
                }
                __syncthreads();
            }
            const uint32_t first = sh.u.seg[kT16Ids];
            total = first >> kT16NextBits;
            if ((first & kT16End) != kT16End || total > m) continue;  // never for a permutation; left to the general kernels
            __syncthreads();
            // for walk 2: the segments on the chain (the first one is thread 0's), grouped by the stretch of
            // the text they write -- the stretch at the text's end first -- and by long / short
            const uint32_t stretch_shift = (uint32_t)max(bit_length(total ? total - 1u : 0u), 3) - 3u;
            uint32_t word[E], bucket[E], slot[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint32_t i = (uint32_t)tid + (uint32_t)e * kT16Threads;
                word[e] = i < kT16Ids ? sh.u.seg[i] : 0u;
                const uint32_t togo = word[e] >> kT16NextBits;
                const bool on = (word[e] & kT16End) == kT16End && own[e] != 0u && togo <= total && i < kT16Ids;
                bucket[e] = kT16Buckets;
                slot[e] = 0;
                if (on) {
                    bucket[e] = min((total - togo) >> stretch_shift, kT16Stretches - 1u) * 2u + (own[e] < kT16Long ? 1u : 0u);
                    slot[e] = atomicAdd(&sh.bucket_fill[bucket[e]], 1u);
                }
            }
            __syncthreads();
            if (tid == 0) {
                uint32_t run = 0;
                for (uint32_t k = 0; k < kT16Buckets; ++k) {
                    sh.bucket_base[k] = run;
                    run += sh.bucket_fill[k];
                }
                sh.bucket_base[kT16Buckets] = run;
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint32_t i = (uint32_t)tid + (uint32_t)e * kT16Threads;
                if (bucket[e] < kT16Buckets) sh.u.seg[sh.bucket_base[bucket[e]] + slot[e]] = t16_seg(word[e] >> kT16NextBits, i);
            }
        }
        if (tid == 0) sh.ctr = 0;
        __syncthreads();
        clk.lap(4);
        t16_walk<true>(sh, m, empty, q0, special, first_sym, mul, inv, mask, ids, out_end, total, sh.bucket_base[kT16Buckets]);
        __syncthreads();
        clk.lap(5);
        if (sh.bad) continue;  // workgroup-uniform; never
        if (tid == 0) {
            atomicAdd(&clk.acc[6], 1ull);
            a.lens[b] = total;
            a.mode[b] = kModeDone;
            atomicAdd(&a.queue[3], 1u);
        }
    }
}

}  // namespace bwtgpu
