// bwt_forward_grid.hpp -- forward BWT of a FEW LARGE blocks: the whole device sorts one batch.
//
// The batch kernel (bwt_forward.hpp) gives every block one workgroup, which fills the chip only
// when there are hundreds of blocks.  A file of one or a few blocks -- the reference's own
// benchmark runs the Canterbury files at block sizes up to 1 MiB, most of them a single block
// (tests/performance.cpp:300-308) -- would leave all but a few compute units idle.  Here the
// suffixes of ALL blocks of the batch form one array that every compute unit works on, one
// kernel launch per step, the host reading one word per round (the number of suffixes still
// tied) to size the next round and to know when to stop.
//
// Same algorithm as the reference's build_suffix_array (src/bwt.cpp:50-99), prefix doubling:
//   row space   block b owns rows [b (block_size+1), ...) -- the very bytes of `out` it writes;
//               a suffix is named by the row-space position g of its first symbol (the block's
//               text followed by its delimiter), so one 32-bit id serves all blocks;
//   round 0     key = (block, first m0 symbols as dense codes, 0 past the end); a device-wide
//               LSD radix sort of (key, g) leaves every suffix at its row up to ties;
//   rank        rank[g] = row of the first suffix with the same key (src/bwt.cpp:85-88 counts
//               the same thing from 0 per block); a suffix whose key is unique is done: its
//               output byte out[rank] = the byte before it, the delimiter for offset 0
//               (src/bwt.cpp:102-123), is written at once;
//   round h     only tied suffixes go on: key = (rank[g], rank[g+h] relative to the block + 1,
//               0 past the end -- src/bwt.cpp:66-82 orders a missing second rank first), sorted
//               again; ties of the new key share rank[g] + (offset of the tie inside the old
//               tie); h doubles until nothing is tied.
// The order is the reference's (unsigned bytes, shorter suffix first), so the bytes are the same.
//
// GridArgs::cyclic selects the other block format this library reads and writes, that of the reference's
// bzip2 comparison tool (util/bzip2_bwt_utils.cpp:104-136: the block's ROTATIONS sorted, their last column,
// and the row of rotation 0): no delimiter row (block b owns rows [b block_size, ...)), every symbol and
// every second rank read modulo the block's length, so nothing is ever "past the end".  Rotations that
// are equal (a block that is a shorter string repeated) stay tied for ever: doubling stops once
// h >= the block length, and the tied ones are ranked by position (grid_cyclic_settle_*), rotation 0
// first among its equals.
#pragma once
#include "bwt_device.hpp"

namespace bwtgpu {

constexpr int kGridThreads = 256;
constexpr int kGridWaves = kGridThreads / kWave;
constexpr int kGridVecs = 8;                           // 64-element vectors per wave in one staged chunk
constexpr int kGridChunk = kGridThreads * kGridVecs;   // elements a workgroup stages at a time
constexpr int kGridDigitBits = 8;
constexpr int kGridBins = 1 << kGridDigitBits;
constexpr uint32_t kGridMaxTiles = 1024;               // workgroups of a sort or rank step, at most
constexpr int kGridScanThreads = 1024;
constexpr int kGridRankVecs = 4;                       // consecutive elements per thread in the rank step
constexpr int kGridRankChunk = kGridThreads * kGridRankVecs;

struct GridArgs {
    const uint8_t* in;    // block b reads in + b*block_size
    uint8_t* out;         // block b writes out + b*(block_size+1)
    uint64_t total_len;   // bytes of `in`
    uint32_t block_size;
    uint32_t nblocks;
    uint32_t delim;
    uint32_t rows;        // total_len + nblocks: suffixes of the whole batch
    uint32_t* rank;       // rows words
    const uint32_t* presence;  // 256-bit mask of the byte values in `in`
    uint32_t* tile_hist;  // kGridBins * kGridMaxTiles words
    uint32_t* tile_rec;   // 4 * kGridMaxTiles words
    uint32_t* count;      // [0] suffixes still tied after the round, [1] suffixes settled as periodic runs since the host last looked
    uint32_t cyclic;      // 1: rotations instead of suffixes (no delimiter rows; rows = total_len)
};

// Row-space position -> block, offset in the block's text (len = the delimiter's position).
struct GridPos {
    uint32_t b, off, len;
};
__device__ __forceinline__ GridPos grid_pos(const GridArgs& a, uint32_t g) {
    GridPos p;
    const uint32_t unit = a.block_size + (a.cyclic ? 0u : 1u);
    p.b = g / unit;
    p.off = g - p.b * unit;
    const uint64_t rem = a.total_len - (uint64_t)p.b * a.block_size;
    p.len = rem < a.block_size ? (uint32_t)rem : a.block_size;
    return p;
}

// The output byte of the suffix (rotation) at p: the byte before it -- the delimiter (the block's last byte)
// for offset 0.
__device__ __forceinline__ uint8_t grid_prev_byte(const GridArgs& a, const GridPos& p) {
    const uint8_t* __restrict__ s = a.in + (uint64_t)p.b * a.block_size;
    if (p.off) return s[p.off - 1u];
    return a.cyclic ? s[p.len - 1u] : (uint8_t)a.delim;
}

// Exclusive running maximum of one value per thread over the workgroup (0 for thread 0);
// `total` receives the workgroup maximum.  s_w needs NT/64 words.  Contains two barriers.
template <int NT>
__device__ __forceinline__ uint32_t block_exclusive_max(uint32_t v, uint32_t* s_w, uint32_t& total) {
    constexpr int W = NT / kWave;
    const int lane = lane_id(), w = wave_id();
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const uint32_t y = __shfl_up(x, o);
        if (lane >= o) x = max(x, y);
    }
    if (lane == kWave - 1) s_w[w] = x;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const uint32_t t = s_w[i];
        if (i < w) pre = max(pre, t);
        tot = max(tot, t);
    }
    __syncthreads();
    total = tot;
    const uint32_t below = __shfl_up(x, 1);  // inclusive maximum of the lanes below mine
    return max(pre, lane ? below : 0u);
}

// ---- round 0 keys ---------------------------------------------------------------------------
// key[g] = block << (m0*sbits) | codes of the m0 symbols from g on, id[g] = g.  Codes are dense:
// 1 + the number of smaller byte values that occur in the batch (the delimiter counted among
// them), which keeps the byte order and leaves 0 for "past the end".
// s_code: 256 entries of LDS.  `first` / `nthreads`: this thread's index among all that share the work, and their number.
__device__ __forceinline__ void grid_keys0_body(const GridArgs& a, uint64_t* __restrict__ key, uint32_t* __restrict__ id,
                                                uint32_t m0, uint32_t sbits, uint16_t* s_code, uint32_t first,
                                                uint32_t nthreads) {
    {
        const uint32_t c = threadIdx.x;
        uint32_t below = 0;
#pragma unroll
        for (uint32_t w = 0; w < 8; ++w) {
            uint32_t m = a.presence[w];
            if (!a.cyclic && (a.delim >> 5) == w) m |= 1u << (a.delim & 31u);
            if (w < (c >> 5)) below += __popc(m);
            else if (w == (c >> 5)) below += __popc(m & ((1u << (c & 31u)) - 1u));
        }
        s_code[c] = (uint16_t)(below + 1u);
    }
    __syncthreads();
    const uint32_t dcode = s_code[a.delim];
    for (uint32_t g = first; g < a.rows; g += nthreads) {
        const GridPos p = grid_pos(a, g);
        const uint8_t* __restrict__ s = a.in + (uint64_t)p.b * a.block_size;
        uint64_t k = p.b;
        if (a.cyclic) {
            uint32_t q = p.off;  // m0 <= the block's length
            for (uint32_t i = 0; i < m0; ++i) {
                k = (k << sbits) | s_code[s[q]];
                q = q + 1u == p.len ? 0u : q + 1u;
            }
        } else {
            for (uint32_t i = 0; i < m0; ++i) {
                const uint32_t q = p.off + i;
                const uint32_t c = q < p.len ? s_code[s[q]] : q == p.len ? dcode : 0u;
                k = (k << sbits) | c;
            }
        }
        key[g] = k;
        id[g] = g;
    }
}
__global__ __launch_bounds__(kGridThreads) void grid_keys0_kernel(GridArgs a, uint64_t* __restrict__ key,
                                                                  uint32_t* __restrict__ id, uint32_t m0,
                                                                  uint32_t sbits) {
    __shared__ uint16_t s_code[256];
    grid_keys0_body(a, key, id, m0, sbits, s_code, blockIdx.x * kGridThreads + threadIdx.x, gridDim.x * kGridThreads);
}

// Second half of a later key: rank of the suffix h further on, relative to the block, + 1; 0 past the
// block's end.  Rotations wrap around instead.
__device__ __forceinline__ uint32_t grid_second_rank(const GridArgs& a, uint32_t g, const GridPos& p, uint64_t h) {
    const uint32_t base = g - p.off;
    if (a.cyclic) return a.rank[base + (uint32_t)(((uint64_t)p.off + h) % p.len)] - base + 1u;
    if ((uint64_t)p.off + h <= (uint64_t)p.len) return a.rank[g + (uint32_t)h] - base + 1u;
    return 0u;
}

// ---- later keys --------------------------------------------------------------------------------
// key[j] = rank[g] << b2 | (rank[g+h] - first row of the block + 1, or 0 past the block's end).
__device__ __forceinline__ void grid_keys_body(const GridArgs& a, const uint32_t* id, uint32_t m, uint64_t h, uint32_t b2,
                                               uint64_t* key, uint32_t first, uint32_t nthreads) {
    for (uint32_t j = first; j < m; j += nthreads) {
        const uint32_t g = id[j];
        const GridPos p = grid_pos(a, g);
        key[j] = ((uint64_t)a.rank[g] << b2) | grid_second_rank(a, g, p, h);
    }
}
__global__ __launch_bounds__(kGridThreads) void grid_keys_kernel(GridArgs a, const uint32_t* __restrict__ id,
                                                                 uint32_t m, uint64_t h, uint32_t b2,
                                                                 uint64_t* __restrict__ key) {
    grid_keys_body(a, id, m, h, b2, key, blockIdx.x * kGridThreads + threadIdx.x, gridDim.x * kGridThreads);
}

// ---- device-wide LSD radix pass: count, scatter -------------------------------------------------
// Tile t = elements [t*tile, (t+1)*tile); tile_hist[t * kGridBins + bin] = its count of digit `bin`.
__device__ __forceinline__ void grid_hist_body(const uint64_t* key, uint32_t m, uint32_t tile, uint32_t shift, uint32_t mask,
                                               uint32_t* tile_hist, uint32_t t, uint32_t* s_h) {
    s_h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lo = t * tile, hi = m - lo < tile ? m : lo + tile;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += kGridThreads)
        atomicAdd(&s_h[(uint32_t)(key[i] >> shift) & mask], 1u);
    __syncthreads();
    tile_hist[t * kGridBins + threadIdx.x] = s_h[threadIdx.x];
}
__global__ __launch_bounds__(kGridThreads) void grid_hist_kernel(const uint64_t* __restrict__ key, uint32_t m,
                                                                 uint32_t tile, uint32_t shift, uint32_t mask,
                                                                 uint32_t* __restrict__ tile_hist) {
    __shared__ uint32_t s_h[kGridBins];
    grid_hist_body(key, m, tile, shift, mask, tile_hist, blockIdx.x, s_h);
}

// Stable scatter of tile t by the digit.  Where the tile's elements of digit d start = the
// elements of smaller digits in all tiles + those of digit d in the tiles before t: thread d
// adds up its column of tile_hist (the whole table is a few hundred KiB at most and sits in the
// L2; a separate scan launch costs more than these loads).  Then, per chunk of kGridChunk
// elements: ranks by wave ballot match + per-wave counters (waves and vectors in element order),
// the chunk gathered by digit in LDS, and written out in runs of equal digit.
struct GridScatterShared {
    uint32_t cursor[kGridBins];             // where the next element of each digit goes
    uint32_t wcnt[kGridWaves][kGridBins];   // per wave: count so far, then chunk-local start
    uint32_t gbase[kGridBins];              // output position minus chunk-local position
    uint32_t wsum[kGridWaves];
    uint64_t key[kGridChunk];
    uint32_t id[kGridChunk];
};
__device__ __forceinline__ void grid_scatter_body(const uint64_t* key_in, const uint32_t* id_in, uint64_t* key_out,
                                                  uint32_t* id_out, uint32_t m, uint32_t tile, uint32_t ntiles,
                                                  uint32_t shift, uint32_t mask, const uint32_t* tile_hist, uint32_t t,
                                                  GridScatterShared& sh) {
    uint32_t* s_cursor = sh.cursor;
    uint32_t (*s_wcnt)[kGridBins] = sh.wcnt;
    uint32_t* s_gbase = sh.gbase;
    uint32_t* s_wsum = sh.wsum;
    uint64_t* s_key = sh.key;
    uint32_t* s_id = sh.id;
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const uint64_t lt = lanemask_lt();
    {
        uint32_t before = 0, all = 0;
        uint32_t i = 0;
        for (; i + 8u <= ntiles; i += 8u) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = tile_hist[(i + (uint32_t)u) * kGridBins + (uint32_t)tid];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                all += v[u];
                before += i + (uint32_t)u < t ? v[u] : 0u;
            }
        }
        for (; i < ntiles; ++i) {
            const uint32_t v = tile_hist[i * kGridBins + (uint32_t)tid];
            all += v;
            before += i < t ? v : 0u;
        }
        uint32_t total;
        s_cursor[tid] = block_exclusive_scan<kGridThreads>(all, s_wsum, total) + before;
    }
    const uint32_t lo = t * tile, hi = m - lo < tile ? m : lo + tile;
    for (uint32_t base = lo; base < hi; base += kGridChunk) {
        const uint32_t cnt = min(hi - base, (uint32_t)kGridChunk);
#pragma unroll
        for (int i = 0; i < kGridWaves; ++i) s_wcnt[i][tid] = 0;
        __syncthreads();
        uint64_t k[kGridVecs];
        uint32_t ids[kGridVecs], lr[kGridVecs];
#pragma unroll
        for (int v = 0; v < kGridVecs; ++v) {
            const uint32_t e = (uint32_t)((w * kGridVecs + v) * kWave + lane);
            const bool valid = e < cnt;
            k[v] = valid ? key_in[base + e] : 0ull;
            ids[v] = valid ? id_in[base + e] : 0u;
        }
#pragma unroll
        for (int v = 0; v < kGridVecs; ++v) {
            const uint32_t e = (uint32_t)((w * kGridVecs + v) * kWave + lane);
            const bool valid = e < cnt;
            const uint32_t d = (uint32_t)(k[v] >> shift) & mask;
            const uint64_t peers = match_digit<kGridDigitBits>(d, valid);
            const uint32_t rk = __popcll(peers & lt);
            const uint32_t c0 = s_wcnt[w][d];
            compiler_fence();  // every lane has read the count before the leaders advance it
            if (valid && rk == 0) s_wcnt[w][d] = c0 + (uint32_t)__popcll(peers);
            compiler_fence();
            lr[v] = c0 + rk;
        }
        __syncthreads();
        {   // thread d: chunk-local start of digit d, per wave, and the digit's place in the output
            uint32_t c[kGridWaves], tot = 0;
#pragma unroll
            for (int i = 0; i < kGridWaves; ++i) {
                c[i] = s_wcnt[i][tid];
                tot += c[i];
            }
            uint32_t total;
            const uint32_t start = block_exclusive_scan<kGridThreads>(tot, s_wsum, total);
            uint32_t run = start;
#pragma unroll
            for (int i = 0; i < kGridWaves; ++i) {
                s_wcnt[i][tid] = run;
                run += c[i];
            }
            s_gbase[tid] = s_cursor[tid] - start;
            s_cursor[tid] += tot;
        }
        __syncthreads();
#pragma unroll
        for (int v = 0; v < kGridVecs; ++v) {
            const uint32_t e = (uint32_t)((w * kGridVecs + v) * kWave + lane);
            if (e < cnt) {
                const uint32_t d = (uint32_t)(k[v] >> shift) & mask;
                const uint32_t p = s_wcnt[w][d] + lr[v];
                s_key[p] = k[v];
                s_id[p] = ids[v];
            }
        }
        __syncthreads();
        for (uint32_t i = tid; i < cnt; i += kGridThreads) {
            const uint64_t kk = s_key[i];
            const uint32_t pos = s_gbase[(uint32_t)(kk >> shift) & mask] + i;
            key_out[pos] = kk;
            id_out[pos] = s_id[i];
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(kGridThreads) void grid_scatter_kernel(const uint64_t* __restrict__ key_in,
                                                                    const uint32_t* __restrict__ id_in,
                                                                    uint64_t* __restrict__ key_out,
                                                                    uint32_t* __restrict__ id_out, uint32_t m,
                                                                    uint32_t tile, uint32_t ntiles, uint32_t shift,
                                                                    uint32_t mask,
                                                                    const uint32_t* __restrict__ tile_hist) {
    __shared__ GridScatterShared sh;
    grid_scatter_body(key_in, id_in, key_out, id_out, m, tile, ntiles, shift, mask, tile_hist, blockIdx.x, sh);
}

// ---- ranks from the sorted keys -----------------------------------------------------------------
// Over the sorted list, position j starts a key run if key[j] != key[j-1] and an r1 run if the
// old rank (key >> b2; the same for all in round 0) changes.  With s2 / s1 = start of j's key /
// r1 run, the new rank is r1 + (s2 - s1) -- in round 0 simply s2, the list being all rows in
// order.  A key run of length 1 is a finished suffix.  Three steps: per tile the last run starts
// and the number of unfinished suffixes (reduce), their running values over the tiles (scan),
// and the pass that writes ranks, output bytes and the list of unfinished suffixes (apply).
struct GridFlags {
    bool khead, rhead, single;
};
__device__ __forceinline__ GridFlags grid_flags(uint64_t prev, uint64_t cur, uint64_t next, uint32_t j, uint32_t m,
                                                uint32_t b2, bool round0) {
    GridFlags f;
    f.khead = j == 0 || cur != prev;
    f.rhead = j == 0 || (!round0 && (cur >> b2) != (prev >> b2));
    f.single = f.khead && (j + 1u == m || next != cur);
    return f;
}

struct GridReduceShared {
    uint32_t k[kGridWaves], r[kGridWaves], n[kGridWaves];
};
__device__ __forceinline__ void grid_rank_reduce_body(const uint64_t* key, uint32_t m, uint32_t tile, uint32_t b2, int round0,
                                                      uint32_t* tile_rec, uint32_t t, GridReduceShared& sh) {
    uint32_t* s_k = sh.k;
    uint32_t* s_r = sh.r;
    uint32_t* s_n = sh.n;
    const uint32_t lo = t * tile, hi = m - lo < tile ? m : lo + tile;
    uint32_t lastk = 0, lastr = 0, tied = 0;  // run starts as position + 1, 0 = none
    for (uint32_t j = lo + threadIdx.x; j < hi; j += kGridThreads) {
        const uint64_t cur = key[j];
        const uint64_t prev = j ? key[j - 1u] : 0ull, next = j + 1u < m ? key[j + 1u] : 0ull;
        const GridFlags f = grid_flags(prev, cur, next, j, m, b2, round0 != 0);
        if (f.khead) lastk = j + 1u;
        if (f.rhead) lastr = j + 1u;
        tied += f.single ? 0u : 1u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lastk = max(lastk, (uint32_t)__shfl_xor(lastk, o));
        lastr = max(lastr, (uint32_t)__shfl_xor(lastr, o));
        tied += __shfl_xor(tied, o);
    }
    if (lane_id() == 0) {
        s_k[wave_id()] = lastk;
        s_r[wave_id()] = lastr;
        s_n[wave_id()] = tied;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t k = 0, r = 0, n = 0;
#pragma unroll
        for (int i = 0; i < kGridWaves; ++i) {
            k = max(k, s_k[i]);
            r = max(r, s_r[i]);
            n += s_n[i];
        }
        tile_rec[4u * t + 0u] = k;
        tile_rec[4u * t + 1u] = r;
        tile_rec[4u * t + 2u] = n;
    }
}
__global__ __launch_bounds__(kGridThreads) void grid_rank_reduce_kernel(const uint64_t* __restrict__ key, uint32_t m,
                                                                        uint32_t tile, uint32_t b2, int round0,
                                                                        uint32_t* __restrict__ tile_rec) {
    __shared__ GridReduceShared sh;
    grid_rank_reduce_body(key, m, tile, b2, round0, tile_rec, blockIdx.x, sh);
}

// One workgroup, one thread per tile: what every tile inherits from the tiles before it.
__global__ __launch_bounds__(kGridScanThreads) void grid_rank_scan_kernel(uint32_t* __restrict__ tile_rec,
                                                                          uint32_t ntiles,
                                                                          uint32_t* __restrict__ count) {
    __shared__ uint32_t s_w[kGridScanThreads / kWave];
    const uint32_t t = threadIdx.x;
    const bool have = t < ntiles;
    const uint32_t k = have ? tile_rec[4u * t + 0u] : 0u, r = have ? tile_rec[4u * t + 1u] : 0u,
                   n = have ? tile_rec[4u * t + 2u] : 0u;
    uint32_t tk, tr, tn;
    const uint32_t pk = block_exclusive_max<kGridScanThreads>(k, s_w, tk);
    const uint32_t pr = block_exclusive_max<kGridScanThreads>(r, s_w, tr);
    const uint32_t pn = block_exclusive_scan<kGridScanThreads>(n, s_w, tn);
    if (have) {
        tile_rec[4u * t + 0u] = pk;
        tile_rec[4u * t + 1u] = pr;
        tile_rec[4u * t + 2u] = pn;
    }
    if (t == 0) count[0] = tn;
}

// cark / carr / carn: what tile t inherits from the tiles before it (last key-run start, last rank-run start,
// unfinished suffixes).
__device__ __forceinline__ void grid_rank_apply_body(const GridArgs& a, const uint64_t* key, const uint32_t* id, uint32_t m,
                                                     uint32_t tile, uint32_t b2, int round0, uint32_t cark, uint32_t carr,
                                                     uint32_t carn, uint32_t* id_out, uint32_t t, uint32_t* s_w) {
    const uint32_t lo = t * tile, hi = m - lo < tile ? m : lo + tile;
    for (uint32_t base = lo; base < hi; base += kGridRankChunk) {
        const uint32_t j0 = base + threadIdx.x * kGridRankVecs;
        uint64_t k[kGridRankVecs + 2];  // key[j0-1 .. j0+kGridRankVecs]
#pragma unroll
        for (int i = 0; i < kGridRankVecs + 2; ++i) {
            const uint32_t j = j0 + (uint32_t)i - 1u;
            k[i] = (j0 + (uint32_t)i >= 1u && j < m) ? key[j] : 0ull;
        }
        GridFlags f[kGridRankVecs];
        uint32_t lastk = 0, lastr = 0, tied = 0;
#pragma unroll
        for (int i = 0; i < kGridRankVecs; ++i) {
            const uint32_t j = j0 + (uint32_t)i;
            f[i] = grid_flags(k[i], k[i + 1], k[i + 2], j, m, b2, round0 != 0);
            if (j < hi) {
                if (f[i].khead) lastk = j + 1u;
                if (f[i].rhead) lastr = j + 1u;
                tied += f[i].single ? 0u : 1u;
            }
        }
        uint32_t tk, tr, tn;
        uint32_t curk = max(cark, block_exclusive_max<kGridThreads>(lastk, s_w, tk));
        uint32_t curr = max(carr, block_exclusive_max<kGridThreads>(lastr, s_w, tr));
        uint32_t slot = carn + block_exclusive_scan<kGridThreads>(tied, s_w, tn);
#pragma unroll
        for (int i = 0; i < kGridRankVecs; ++i) {
            const uint32_t j = j0 + (uint32_t)i;
            if (j < hi) {
                if (f[i].khead) curk = j + 1u;
                if (f[i].rhead) curr = j + 1u;
                const uint32_t r1 = round0 ? 0u : (uint32_t)(k[i + 1] >> b2);
                const uint32_t nr = r1 + (curk - curr);  // both are position + 1
                const uint32_t g = id[j];
                a.rank[g] = nr;
                if (f[i].single) {
                    a.out[nr] = grid_prev_byte(a, grid_pos(a, g));
                } else {
                    id_out[slot++] = g;
                }
            }
        }
        cark = max(cark, tk);
        carr = max(carr, tr);
        carn += tn;
    }
}
__global__ __launch_bounds__(kGridThreads) void grid_rank_apply_kernel(GridArgs a, const uint64_t* __restrict__ key,
                                                                       const uint32_t* __restrict__ id, uint32_t m,
                                                                       uint32_t tile, uint32_t b2, int round0,
                                                                       const uint32_t* __restrict__ tile_rec,
                                                                       uint32_t* __restrict__ id_out) {
    __shared__ uint32_t s_w[kGridWaves];
    const uint32_t t = blockIdx.x;
    grid_rank_apply_body(a, key, id, m, tile, b2, round0, tile_rec[4u * t + 0u], tile_rec[4u * t + 1u], tile_rec[4u * t + 2u],
                         id_out, t, s_w);
}

// ---- periodic runs: whole groups resolved in one step ------------------------------------------------
// A group of tied suffixes whose positions form an arithmetic progression i, i+p, i+2p, ... with
// p <= h lies inside a stretch of the text with period p (each member's h symbols overlap the next
// member's), and any two neighbours agree exactly up to the one position E where that period
// breaks; text[E-p] against text[E] (the end of the block lowest, src/bwt.cpp:77-79) therefore
// orders the whole group: ascending position, or descending.  Plain doubling needs
// log2(run length / h) more rounds for it (ptt5, a fax bitmap full of zero runs: 14 rounds).
// Run by the host when a round has left the list nearly as long as it was.  Groups are named by
// their rank (the row of their first suffix), which indexes two tables in buffers that are idle
// between rounds: {min, max} and {count, state} of the members' positions.  Five small kernels
// over the tied list: clear the groups' entries; gather min / max / count; every member checks that
// it lies on the progression; the first member of a group finds E (at most p <= kGridMaxPeriod
// steps: were the run longer the group would have another member) and notes the direction; the
// members take their final ranks and store their output bytes.  They stay in the list for one
// more round, where their now unique ranks make them drop out as finished.
constexpr uint32_t kGridMaxPeriod = 256;
constexpr uint32_t kGridRunBad = 1u, kGridRunDone = 2u, kGridRunAsc = 4u;

// The step of the progression a group with these statistics can be, or 0.
__device__ __forceinline__ uint32_t grid_run_step(const GridArgs& a, uint2 mm, uint32_t cnt, uint64_t h) {
    if (cnt < 2u) return 0u;
    const GridPos p = grid_pos(a, mm.y);           // of the highest member: all h symbols inside the block?
    if ((uint64_t)p.off + h > (uint64_t)p.len) return 0u;
    const uint32_t span = mm.y - mm.x, per = span / (cnt - 1u);
    if (per == 0u || per * (cnt - 1u) != span || (uint64_t)per > h || per > kGridMaxPeriod) return 0u;
    return per;
}

__device__ __forceinline__ void grid_runs_clear_body(const GridArgs& a, const uint32_t* id, uint32_t m, uint2* tmm, uint2* tcs, uint32_t first, uint32_t nthreads) {
    for (uint32_t j = first; j < m; j += nthreads) {
        const uint32_t r = a.rank[id[j]];
        tmm[r] = make_uint2(0xFFFFFFFFu, 0u);
        tcs[r] = make_uint2(0u, 0u);
    }
}
__global__ __launch_bounds__(kGridThreads) void grid_runs_clear_kernel(GridArgs a, const uint32_t* __restrict__ id,
                                                                       uint32_t m, uint2* __restrict__ tmm,
                                                                       uint2* __restrict__ tcs) {
    grid_runs_clear_body(a, id, m, tmm, tcs, blockIdx.x * kGridThreads + threadIdx.x, gridDim.x * kGridThreads);
}

__device__ __forceinline__ void grid_runs_stats_body(const GridArgs& a, const uint32_t* id, uint32_t m, uint2* tmm, uint2* tcs, uint32_t first, uint32_t nthreads) {
    const int lane = lane_id();
    for (uint32_t j0 = (first & ~63u); j0 < m; j0 += nthreads) {
        const uint32_t j = j0 + (uint32_t)lane;
        const bool valid = j < m;
        const uint32_t g = id[valid ? j : j0];
        const uint32_t r = a.rank[g];
        const uint32_t r0 = __shfl(r, 0);
        if (__ballot(valid && r != r0) == 0ull) {  // the vector lies inside one group: one update
            uint32_t mn = valid ? g : 0xFFFFFFFFu, mx = valid ? g : 0u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                mn = min(mn, (uint32_t)__shfl_xor(mn, o));
                mx = max(mx, (uint32_t)__shfl_xor(mx, o));
            }
            const uint32_t members = (uint32_t)__popcll(__ballot(valid));
            if (lane == 0) {
                atomicMin(&tmm[r0].x, mn);
                atomicMax(&tmm[r0].y, mx);
                atomicAdd(&tcs[r0].x, members);
            }
        } else if (valid) {
            atomicMin(&tmm[r].x, g);
            atomicMax(&tmm[r].y, g);
            atomicAdd(&tcs[r].x, 1u);
        }
    }
}
__global__ __launch_bounds__(kGridThreads) void grid_runs_stats_kernel(GridArgs a, const uint32_t* __restrict__ id,
                                                                       uint32_t m, uint2* __restrict__ tmm,
                                                                       uint2* __restrict__ tcs) {
    grid_runs_stats_body(a, id, m, tmm, tcs, blockIdx.x * kGridThreads + threadIdx.x, gridDim.x * kGridThreads);
}

__device__ __forceinline__ void grid_runs_check_body(const GridArgs& a, const uint32_t* id, uint32_t m, uint64_t h, const uint2* tmm, uint2* tcs, uint32_t first, uint32_t nthreads) {
    for (uint32_t j = first; j < m; j += nthreads) {
        const uint32_t g = id[j], r = a.rank[g];
        const uint2 mm = tmm[r];
        const uint32_t per = grid_run_step(a, mm, tcs[r].x, h);
        if (per && (g - mm.x) % per != 0u) atomicOr(&tcs[r].y, kGridRunBad);
    }
}
__global__ __launch_bounds__(kGridThreads) void grid_runs_check_kernel(GridArgs a, const uint32_t* __restrict__ id,
                                                                       uint32_t m, uint64_t h,
                                                                       const uint2* __restrict__ tmm,
                                                                       uint2* __restrict__ tcs) {
    grid_runs_check_body(a, id, m, h, tmm, tcs, blockIdx.x * kGridThreads + threadIdx.x, gridDim.x * kGridThreads);
}

__device__ __forceinline__ void grid_runs_break_body(const GridArgs& a, const uint32_t* id, uint32_t m, uint64_t h, const uint2* tmm, uint2* tcs, uint32_t first, uint32_t nthreads) {
    for (uint32_t j = first; j < m; j += nthreads) {
        const uint32_t g = id[j], r = a.rank[g];
        if (j && a.rank[id[j - 1u]] == r) continue;  // not the group's first member
        const uint2 mm = tmm[r];
        const uint2 cs = tcs[r];
        const uint32_t per = grid_run_step(a, mm, cs.x, h);
        if (per == 0u || (cs.y & kGridRunBad)) continue;
        const GridPos p = grid_pos(a, mm.y);
        const uint8_t* __restrict__ s = a.in + (uint64_t)p.b * a.block_size;
        auto symbol = [&](uint32_t off) { return off < p.len ? (int)s[off] : (off == p.len ? (int)a.delim : -1); };
        const uint32_t from = p.off + (uint32_t)h;  // offsets in the block from here on
        for (uint32_t t = 0; t <= per; ++t) {
            const uint32_t x = from + t;
            const int before = symbol(x - per), here = symbol(x);
            if (before != here) {
                tcs[r].y = kGridRunDone | (before < here ? kGridRunAsc : 0u) | (per << 8);
                break;
            }
            if (x > p.len) break;
        }
    }
}
__global__ __launch_bounds__(kGridThreads) void grid_runs_break_kernel(GridArgs a, const uint32_t* __restrict__ id,
                                                                       uint32_t m, uint64_t h,
                                                                       const uint2* __restrict__ tmm,
                                                                       uint2* __restrict__ tcs) {
    grid_runs_break_body(a, id, m, h, tmm, tcs, blockIdx.x * kGridThreads + threadIdx.x, gridDim.x * kGridThreads);
}

// (count[1] += the suffixes settled: the host reads it with the round's count and tries less often
// when it stays small)
__device__ __forceinline__ void grid_runs_apply_body(const GridArgs& a, const uint32_t* id, uint32_t m, const uint2* tmm, const uint2* tcs, uint32_t first, uint32_t nthreads) {
    uint32_t settled = 0;
    for (uint32_t j = first; j < m; j += nthreads) {
        const uint32_t g = id[j], r = a.rank[g];
        const uint2 cs = tcs[r];
        if (!(cs.y & kGridRunDone)) continue;
        ++settled;
        const uint32_t per = cs.y >> 8, k = (g - tmm[r].x) / per;
        const uint32_t nr = r + ((cs.y & kGridRunAsc) ? k : cs.x - 1u - k);
        a.rank[g] = nr;
        a.out[nr] = grid_prev_byte(a, grid_pos(a, g));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) settled += __shfl_xor(settled, o);
    if (lane_id() == 0 && settled) atomicAdd(&a.count[1], settled);
}
__global__ __launch_bounds__(kGridThreads) void grid_runs_apply_kernel(GridArgs a, const uint32_t* __restrict__ id,
                                                                       uint32_t m, const uint2* __restrict__ tmm,
                                                                       const uint2* __restrict__ tcs) {
    grid_runs_apply_body(a, id, m, tmm, tcs, blockIdx.x * kGridThreads + threadIdx.x, gridDim.x * kGridThreads);
}

// ---- the last rounds in one launch ----------------------------------------------------------------
// Once few suffixes are still tied, a round's dozen launches and the host's look at the count
// cost far more than its work.  One workgroup then finishes the job alone: the list stays in
// registers and LDS (kGridFinishVecs elements per thread), every round builds the keys, sorts
// them with the same ballot-match passes as grid_scatter_kernel, ranks them with the same run
// logic as grid_rank_apply_kernel, and keeps the still-tied suffixes for the next round, until
// none is left.  Ranks go through global memory (other suffixes' keys read them); a workgroup
// barrier orders those writes before the next round's reads.
constexpr int kGridFinishThreads = 1024;
constexpr int kGridFinishVecs = 8;
constexpr uint32_t kGridFinishCap = kGridFinishThreads * kGridFinishVecs;

struct GridFinishShared {
    uint64_t key[kGridFinishCap];
    uint32_t id[kGridFinishCap];
    uint32_t wcnt[kGridFinishThreads / kWave][kGridBins];
    uint32_t wsum[kGridFinishThreads / kWave];
};

__device__ __forceinline__ void grid_finish_body(const GridArgs& a, const uint32_t* id_in, uint32_t m, uint64_t h, uint32_t b1,
                                                 uint32_t b2, GridFinishShared& sh) {
    constexpr int NT = kGridFinishThreads, W = NT / kWave, V = kGridFinishVecs;
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const uint64_t lt = lanemask_lt();
    uint64_t k[V];
    uint32_t ids[V], lr[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const uint32_t e = (uint32_t)((w * V + v) * kWave + lane);
        ids[v] = e < m ? id_in[e] : 0u;
    }
    const uint64_t h_end = a.cyclic ? (uint64_t)min((uint64_t)a.block_size, a.total_len) : ~0ull;  // rotations tied at this depth are equal
    for (int round = 0; m && round < 64 && h < h_end; ++round, h *= 2) {
        // keys of the round
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const uint32_t e = (uint32_t)((w * V + v) * kWave + lane);
            k[v] = 0;
            if (e < m) {
                const uint32_t g = ids[v];
                k[v] = ((uint64_t)a.rank[g] << b2) | grid_second_rank(a, g, grid_pos(a, g), h);
            }
        }
        // LSD radix sort, the list going from registers through LDS back to registers every pass
#pragma unroll 1
        for (uint32_t shift = 0; shift < b1 + b2; shift += kGridDigitBits) {
            const uint32_t mask = (1u << min((uint32_t)kGridDigitBits, b1 + b2 - shift)) - 1u;
            for (int i = tid; i < W * kGridBins; i += NT) (&sh.wcnt[0][0])[i] = 0;
            __syncthreads();
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const uint32_t e = (uint32_t)((w * V + v) * kWave + lane);
                const bool valid = e < m;
                const uint32_t d = (uint32_t)(k[v] >> shift) & mask;
                const uint64_t peers = match_digit<kGridDigitBits>(d, valid);
                const uint32_t rk = __popcll(peers & lt);
                const uint32_t c0 = sh.wcnt[w][d];
                compiler_fence();  // every lane has read the count before the leaders advance it
                if (valid && rk == 0) sh.wcnt[w][d] = c0 + (uint32_t)__popcll(peers);
                compiler_fence();
                lr[v] = c0 + rk;
            }
            __syncthreads();
            {   // thread d < 256: where digit d starts, per wave
                uint32_t tot = 0;
                if (tid < kGridBins)
                    for (int i = 0; i < W; ++i) tot += sh.wcnt[i][tid];
                uint32_t total;
                uint32_t run = block_exclusive_scan<NT>(tot, sh.wsum, total);
                if (tid < kGridBins)
                    for (int i = 0; i < W; ++i) {
                        const uint32_t c = sh.wcnt[i][tid];
                        sh.wcnt[i][tid] = run;
                        run += c;
                    }
            }
            __syncthreads();
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const uint32_t e = (uint32_t)((w * V + v) * kWave + lane);
                if (e < m) {
                    const uint32_t p = sh.wcnt[w][(uint32_t)(k[v] >> shift) & mask] + lr[v];
                    sh.key[p] = k[v];
                    sh.id[p] = ids[v];
                }
            }
            __syncthreads();
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const uint32_t e = (uint32_t)((w * V + v) * kWave + lane);
                k[v] = e < m ? sh.key[e] : 0ull;
                ids[v] = e < m ? sh.id[e] : 0u;
            }
            __syncthreads();
        }
        // ranks: thread t takes the V consecutive positions from t * V of the sorted list in LDS
        const uint32_t j0 = (uint32_t)tid * V;
        uint64_t q[V + 2];
        uint32_t gs[V];
#pragma unroll
        for (int i = 0; i < V + 2; ++i) {
            const uint32_t j = j0 + (uint32_t)i - 1u;
            q[i] = (j0 + (uint32_t)i >= 1u && j < m) ? sh.key[j] : 0ull;
        }
#pragma unroll
        for (int i = 0; i < V; ++i) gs[i] = j0 + (uint32_t)i < m ? sh.id[j0 + (uint32_t)i] : 0u;
        GridFlags f[V];
        uint32_t lastk = 0, lastr = 0, tied = 0;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const uint32_t j = j0 + (uint32_t)i;
            f[i] = grid_flags(q[i], q[i + 1], q[i + 2], j, m, b2, false);
            if (j < m) {
                if (f[i].khead) lastk = j + 1u;
                if (f[i].rhead) lastr = j + 1u;
                tied += f[i].single ? 0u : 1u;
            }
        }
        uint32_t tk, tr, tn;
        uint32_t curk = block_exclusive_max<NT>(lastk, sh.wsum, tk);
        uint32_t curr = block_exclusive_max<NT>(lastr, sh.wsum, tr);
        uint32_t slot = block_exclusive_scan<NT>(tied, sh.wsum, tn);  // (the barriers inside: all of sh.id is read by now)
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const uint32_t j = j0 + (uint32_t)i;
            if (j < m) {
                if (f[i].khead) curk = j + 1u;
                if (f[i].rhead) curr = j + 1u;
                const uint32_t nr = (uint32_t)(q[i + 1] >> b2) + (curk - curr);
                const uint32_t g = gs[i];
                a.rank[g] = nr;
                if (f[i].single) {
                    a.out[nr] = grid_prev_byte(a, grid_pos(a, g));
                } else {
                    sh.id[slot++] = g;
                }
            }
        }
        __threadfence_block();
        __syncthreads();
        m = tn;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const uint32_t e = (uint32_t)((w * V + v) * kWave + lane);
            ids[v] = e < m ? sh.id[e] : 0u;
        }
        __syncthreads();
    }
    if (a.cyclic && m) {
        // Equal rotations: the list holds them group by group (ascending rank), each group in ascending
        // position (every sort and every compaction above is stable).  Final rank = the group's + the place in it.
        uint32_t* const rk = reinterpret_cast<uint32_t*>(sh.key);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const uint32_t e = (uint32_t)((w * V + v) * kWave + lane);
            if (e < m) rk[e] = a.rank[ids[v]];
        }
        __syncthreads();
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const uint32_t e = (uint32_t)((w * V + v) * kWave + lane);
            if (e < m) {
                const uint32_t r = rk[e];
                uint32_t lo = 0, hi = e;  // first list index with this rank
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (rk[mid] < r) lo = mid + 1u;
                    else hi = mid;
                }
                const uint32_t nr = r + (e - lo), g = ids[v];
                a.rank[g] = nr;
                a.out[nr] = grid_prev_byte(a, grid_pos(a, g));
            }
        }
    }
}

// Equal rotations left in a list too long for one workgroup: the same in two steps (the first reads every
// rank before the second changes any).  place[j] = list index of the first member of j's group.
__global__ __launch_bounds__(kGridThreads) void grid_cyclic_settle_find_kernel(GridArgs a, const uint32_t* __restrict__ id,
                                                                               uint32_t m, uint32_t* __restrict__ place) {
    for (uint32_t j = blockIdx.x * kGridThreads + threadIdx.x; j < m; j += gridDim.x * kGridThreads) {
        const uint32_t r = a.rank[id[j]];
        uint32_t lo = 0, hi = j;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (a.rank[id[mid]] < r) lo = mid + 1u;
            else hi = mid;
        }
        place[j] = lo;
    }
}
__global__ __launch_bounds__(kGridThreads) void grid_cyclic_settle_apply_kernel(GridArgs a, const uint32_t* __restrict__ id,
                                                                                uint32_t m, const uint32_t* __restrict__ place) {
    for (uint32_t j = blockIdx.x * kGridThreads + threadIdx.x; j < m; j += gridDim.x * kGridThreads) {
        const uint32_t g = id[j];
        const uint32_t nr = a.rank[g] + (j - place[j]);
        // (a.rank[g] of a group's first member does not change, and only its own entry is read here)
        a.rank[g] = nr;
        a.out[nr] = grid_prev_byte(a, grid_pos(a, g));
    }
}

// rows[b] = the row of block b's rotation 0 among the block's sorted rotations.
__global__ __launch_bounds__(kGridThreads) void grid_cyclic_origin_kernel(GridArgs a, uint32_t* __restrict__ rows) {
    for (uint32_t b = blockIdx.x * kGridThreads + threadIdx.x; b < a.nblocks; b += gridDim.x * kGridThreads)
        rows[b] = a.rank[b * a.block_size] - b * a.block_size;
}

__global__ __launch_bounds__(kGridFinishThreads) void grid_finish_kernel(GridArgs a, const uint32_t* __restrict__ id_in,
                                                                         uint32_t m, uint64_t h, uint32_t b1,
                                                                         uint32_t b2) {
    __shared__ GridFinishShared sh;
    grid_finish_body(a, id_in, m, h, b1, b2, sh);
}

}  // namespace bwtgpu
