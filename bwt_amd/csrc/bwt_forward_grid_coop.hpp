// bwt_forward_grid_coop.hpp -- the device-wide forward path (bwt_forward_grid.hpp) as ONE resident kernel.
//
// Driven from the host, that path costs a dozen launches per doubling round and, worse, a read-back of
// the number of suffixes still tied after every round: the call blocks its caller and a stream cannot
// run ahead.  Here every step is the same code (the *_body functions), but all of them run inside one
// kernel whose workgroups are resident together (a cooperative launch) and meet at grid-wide barriers;
// every workgroup derives the round's decisions -- how many suffixes are left, whether the periodic-run
// step is due, when the last-rounds kernel can take over -- from the same device memory, so nothing goes
// back to the host.  Two launches per batch: this kernel, and grid_finish_coop_kernel for the last
// rounds (one workgroup; it reads where to start from the state words this kernel leaves).
#pragma once
#include "bwt_forward_grid.hpp"

namespace bwtgpu {

constexpr uint32_t kCoopSpinCap = 1u << 23;  // polls of a barrier word before a workgroup gives the batch up (seconds)
constexpr uint32_t kErrGridBarrier = 4u;     // status flag: a grid barrier timed out / the sort did not converge

struct GridCoopArgs {
    GridArgs a;
    uint64_t* key[2];
    uint32_t* id[2];
    uint32_t* bar;     // 24 barrier words (see GridBarrier)
    uint32_t* state;   // for the last-rounds kernel: [0] suffixes still tied, [1] which id buffer lists them, [2..3] depth h
    uint32_t* status;  // the batch's error flags
};

// ---- grid-wide barrier ------------------------------------------------------------------------------------
// The eight XCDs of the chip have an L2 each, coherent with the others only through explicit write-backs
// and invalidations: a barrier in which every workgroup issues both (two __threadfence) serialises them --
// measured 20 us per barrier at 256 workgroups.  Here the write-back is issued ONCE per XCD: workgroups
// arrive at their own XCD's counter (their stores have drained to that XCD's L2 by then), the last one to
// arrive there writes the L2 back and arrives at the top counter, the last one there publishes the new
// generation; everybody polls the generation word and then drops its stale lines (an acquire fence).  Which
// XCD a workgroup runs on is read from the hardware (never assumed from its index), and how many workgroups
// each XCD holds is counted at the start of the kernel, behind one barrier of the plain kind.
// Barrier words (zeroed by the host before the launch): [0] arrivals at the top counter, [1] generation,
// [2] somebody has given up, [3] arrivals of the census barrier, [8..15] arrivals per XCD, [16..23] workgroups per XCD.
struct GridBarrier {
    uint32_t epoch;    // barriers passed so far
    uint32_t xcc;      // the XCD this workgroup runs on
    uint32_t mine;     // workgroups on that XCD
    uint32_t groups;   // XCDs that hold workgroups of this grid
};

__device__ __forceinline__ uint32_t coop_load(const uint32_t* p) {
    return __hip_atomic_load(const_cast<uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Waits until *word >= want.  False if the grid has been given up (here or elsewhere).
__device__ __forceinline__ bool coop_wait(const GridCoopArgs& c, const uint32_t* word, uint32_t want) {
    uint32_t polls = 0;
    while (coop_load(word) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (++polls > kCoopSpinCap || coop_load(&c.bar[2])) {
            __hip_atomic_store(&c.bar[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicOr(c.status, kErrGridBarrier);
            return false;
        }
    }
    return true;
}

// The first barrier of the kernel: counts the workgroups per XCD.  All threads get the same answer.
__device__ __forceinline__ bool grid_census(const GridCoopArgs& c, uint32_t nwg, GridBarrier& gb, uint32_t* s_ok) {
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    gb.xcc = xcc & 7u;
    gb.epoch = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&c.bar[16u + gb.xcc], 1u);
        __threadfence();
        bool ok = true;
        if (atomicAdd(&c.bar[3], 1u) + 1u == nwg) __hip_atomic_store(&c.bar[1], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        else ok = coop_wait(c, &c.bar[1], 1u);
        __threadfence();
        *s_ok = ok ? 1u : 0u;
    }
    __syncthreads();
    gb.epoch = 1;
    gb.mine = coop_load(&c.bar[16u + gb.xcc]);
    gb.groups = 0;
#pragma unroll
    for (uint32_t x = 0; x < 8u; ++x) gb.groups += coop_load(&c.bar[16u + x]) ? 1u : 0u;
    return *s_ok != 0u;
}

// Returns false if the grid has been given up (a wait timed out somewhere): the caller leaves the kernel.
// All threads of the workgroup get the same answer.
__device__ __forceinline__ bool grid_sync(const GridCoopArgs& c, GridBarrier& gb, uint32_t* s_ok) {
    __syncthreads();  // every wave's stores have drained to this XCD's L2
    ++gb.epoch;
    if (threadIdx.x == 0) {
        const uint32_t round = gb.epoch - 1u;  // barriers of this kind so far, this one included
        if (__hip_atomic_fetch_add(&c.bar[8u + gb.xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == round * gb.mine) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // the XCD's L2 written back, once
            if (__hip_atomic_fetch_add(&c.bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == round * gb.groups)
                __hip_atomic_store(&c.bar[1], gb.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const bool ok = coop_wait(c, &c.bar[1], gb.epoch);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // stale lines dropped: what the others wrote is read afresh
        *s_ok = ok ? 1u : 0u;
    }
    __syncthreads();
    return *s_ok != 0u;
}

struct GridCoopShared {
    union {
        GridScatterShared scatter;
        GridReduceShared reduce;
        uint32_t hist[kGridBins];
        uint16_t code[256];
        uint32_t mask[8];
    } u;
    uint32_t w[3][kGridWaves];
    uint32_t ok;
};

// (max, max, sum) of one triple per thread over the workgroup, for all threads.  Two barriers.
__device__ __forceinline__ void block_reduce_mms(uint32_t& k, uint32_t& r, uint32_t& n, uint32_t (*s_w)[kGridWaves]) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        k = max(k, (uint32_t)__shfl_xor(k, o));
        r = max(r, (uint32_t)__shfl_xor(r, o));
        n += (uint32_t)__shfl_xor(n, o);
    }
    if (lane_id() == 0) {
        s_w[0][wave_id()] = k;
        s_w[1][wave_id()] = r;
        s_w[2][wave_id()] = n;
    }
    __syncthreads();
    k = r = n = 0;
#pragma unroll
    for (int i = 0; i < kGridWaves; ++i) {
        k = max(k, s_w[0][i]);
        r = max(r, s_w[1][i]);
        n += s_w[2][i];
    }
    __syncthreads();
}

__device__ __host__ __forceinline__ uint32_t grid_tile_size(uint32_t m) {
    uint64_t tile = ((uint64_t)m + 255u) / 256u;
    if (tile < 1) tile = 1;
    tile = (tile + kGridChunk - 1) / kGridChunk * kGridChunk;
    if (tile > 65536) {
        tile = ((uint64_t)m + kGridMaxTiles - 1) / kGridMaxTiles;
        if (tile < 65536) tile = 65536;
        tile = (tile + kGridChunk - 1) / kGridChunk * kGridChunk;
    }
    return (uint32_t)tile;
}

__global__ __launch_bounds__(kGridThreads) void grid_forward_coop_kernel(GridCoopArgs c) {
    __shared__ GridCoopShared sh;
    const GridArgs& a = c.a;
    const int tid = threadIdx.x;
    const uint32_t wg = blockIdx.x, nwg = gridDim.x;
    const uint32_t first = wg * kGridThreads + (uint32_t)tid, nthreads = nwg * kGridThreads;
    GridBarrier gb;
    if (!grid_census(c, nwg, gb, &sh.ok)) return;
#define GRID_SYNC()                              \
    do {                                         \
        if (!grid_sync(c, gb, &sh.ok)) return;   \
    } while (0)

    // ---- byte values in the batch -> a.presence (the host has cleared it)
    {
        uint32_t m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        auto mark = [&](uint32_t b) {
#pragma unroll
            for (int j = 0; j < 8; ++j) m[j] |= ((b >> 5) == (uint32_t)j) ? (1u << (b & 31u)) : 0u;
        };
        const uint64_t len = a.total_len;
        const uint64_t mis = (4u - (uint32_t)(reinterpret_cast<uintptr_t>(a.in) & 3u)) & 3u;
        const uint64_t head = mis < len ? mis : len;
        const uint64_t nd = (len - head) / 4u;
        const uint32_t* __restrict__ words = reinterpret_cast<const uint32_t*>(a.in + head);
        for (uint64_t i = first; i < nd; i += nthreads) {
            const uint32_t v = words[i];
            mark(v & 255u);
            mark((v >> 8) & 255u);
            mark((v >> 16) & 255u);
            mark(v >> 24);
        }
        if (first < head) mark(a.in[first]);
        for (uint64_t i = head + 4u * nd + first; i < len; i += nthreads) mark(a.in[i]);
        if (tid < 8) sh.u.mask[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint32_t x = m[j];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) x |= (uint32_t)__shfl_xor(x, o);
            if (lane_id() == 0 && x) atomicOr(&sh.u.mask[j], x);
        }
        __syncthreads();
        uint32_t* presence = const_cast<uint32_t*>(a.presence);
        if (tid < 8 && sh.u.mask[tid]) atomicOr(&presence[tid], sh.u.mask[tid]);
    }
    GRID_SYNC();
    // ---- symbol width and how many symbols the first key holds (the same in every workgroup)
    uint32_t sigma = 0;
#pragma unroll
    for (uint32_t w = 0; w < 8; ++w) {
        uint32_t m = __hip_atomic_load(&a.presence[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((a.delim >> 5) == w) m |= 1u << (a.delim & 31u);
        sigma += (uint32_t)__popc(m);
    }
    const uint32_t sbits = (uint32_t)bit_length(sigma);  // codes 1 .. sigma
    const uint32_t bbits = (uint32_t)bit_length(a.nblocks - 1u);
    const uint64_t longest = (a.total_len < a.block_size ? a.total_len : (uint64_t)a.block_size) + 1u;  // symbols of the longest block, delimiter included
    uint32_t m0 = max(1u, (40u - bbits) / sbits);  // five 8-bit passes
    if ((uint64_t)m0 > longest) m0 = (uint32_t)longest;
    const uint32_t b1 = (uint32_t)bit_length(a.rows - 1u), b2 = (uint32_t)bit_length((uint32_t)longest);

    grid_keys0_body(a, c.key[0], c.id[0], m0, sbits, sh.u.code, first, nthreads);
    GRID_SYNC();
    int cur = 0;  // key[cur] / id[cur] hold the list the next step reads

    uint32_t m = a.rows;
    uint64_t h = m0;
    uint32_t m_before = a.rows;  // the list's length one round earlier
    bool runs_tried = false;
    uint32_t runs_m = 0, runs_wait = 0, runs_backoff = 1, runs_settled = 0;
    for (int round = -1;; ++round) {
        uint32_t bits;
        if (round < 0) {
            bits = bbits + m0 * sbits;  // the first sort: all rows by their first m0 symbols
        } else {
            if (m == 0u) break;
            if (round >= 64 || m > a.rows) {
                if (first == 0) atomicOr(c.status, kErrGridBarrier);
                m = 0;
                break;
            }
            // The periodic-run step: when a round has left the list nearly as long as it was (right after the
            // first sort: only if hardly anything is settled), and less and less often while it finds little.
            const bool stalled = round == 0 ? (uint64_t)m * 50u > (uint64_t)m_before * 49u : (uint64_t)m * 10u > (uint64_t)m_before * 9u;
            if (runs_wait) {
                --runs_wait;
            } else if (m > kGridFinishCap && stalled) {
                uint2* tmm = reinterpret_cast<uint2*>(c.key[cur]);  // both key buffers are idle between the rank step and the next keys
                uint2* tcs = reinterpret_cast<uint2*>(c.key[cur ^ 1]);
                grid_runs_clear_body(a, c.id[cur], m, tmm, tcs, first, nthreads);
                GRID_SYNC();
                grid_runs_stats_body(a, c.id[cur], m, tmm, tcs, first, nthreads);
                GRID_SYNC();
                grid_runs_check_body(a, c.id[cur], m, h, tmm, tcs, first, nthreads);
                GRID_SYNC();
                grid_runs_break_body(a, c.id[cur], m, h, tmm, tcs, first, nthreads);
                GRID_SYNC();
                grid_runs_apply_body(a, c.id[cur], m, tmm, tcs, first, nthreads);
                GRID_SYNC();
                runs_tried = true;
                runs_m = m;
            }
            m_before = m;
            if (m <= kGridFinishCap) break;  // few left: the last-rounds kernel takes over
            grid_keys_body(a, c.id[cur], m, h, b2, c.key[cur], first, nthreads);
            GRID_SYNC();
            bits = b1 + b2;
        }
        const uint32_t tile = grid_tile_size(m), ntiles = (m + tile - 1u) / tile;
        // ---- device-wide LSD radix sort of key[cur] / id[cur]
        for (uint32_t shift = 0; shift < bits; shift += kGridDigitBits) {
            const uint32_t mask = (1u << min((uint32_t)kGridDigitBits, bits - shift)) - 1u;
            for (uint32_t t = wg; t < ntiles; t += nwg) {
                grid_hist_body(c.key[cur], m, tile, shift, mask, a.tile_hist, t, sh.u.hist);
                __syncthreads();
            }
            GRID_SYNC();
            for (uint32_t t = wg; t < ntiles; t += nwg)
                grid_scatter_body(c.key[cur], c.id[cur], c.key[cur ^ 1], c.id[cur ^ 1], m, tile, ntiles, shift, mask, a.tile_hist, t,
                                  sh.u.scatter);
            GRID_SYNC();
            cur ^= 1;
        }
        // ---- ranks of the sorted list; the suffixes still tied go to id[cur ^ 1]
        const int round0 = round < 0 ? 1 : 0;
        for (uint32_t t = wg; t < ntiles; t += nwg) {
            grid_rank_reduce_body(c.key[cur], m, tile, b2, round0, a.tile_rec, t, sh.u.reduce);
            __syncthreads();
        }
        GRID_SYNC();
        uint32_t left;
        {   // the number still tied: the sum of all tiles' counts (every workgroup works it out for itself)
            uint32_t k = 0, r = 0, n = 0;
            for (uint32_t i = (uint32_t)tid; i < ntiles; i += kGridThreads) n += a.tile_rec[4u * i + 2u];
            block_reduce_mms(k, r, n, sh.w);
            left = n;
        }
        for (uint32_t t = wg; t < ntiles; t += nwg) {
            // what tile t inherits from the tiles before it
            uint32_t k = 0, r = 0, n = 0;
            for (uint32_t i = (uint32_t)tid; i < t; i += kGridThreads) {
                k = max(k, a.tile_rec[4u * i + 0u]);
                r = max(r, a.tile_rec[4u * i + 1u]);
                n += a.tile_rec[4u * i + 2u];
            }
            block_reduce_mms(k, r, n, sh.w);
            grid_rank_apply_body(a, c.key[cur], c.id[cur], m, tile, b2, round0, k, r, n, c.id[cur ^ 1], t, sh.w[0]);
            __syncthreads();
        }
        GRID_SYNC();
        cur ^= 1;  // the tied suffixes are the next list
        const uint32_t settled_before = runs_settled;
        runs_settled = __hip_atomic_load(&a.count[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        m = left;
        if (round >= 0) {
            if (runs_tried) {  // did it pay?  if not, wait twice as many rounds before the next try
                runs_tried = false;
                if (runs_settled - settled_before < runs_m / 16u) runs_wait = runs_backoff, runs_backoff *= 2;
                else runs_backoff = 1;
            }
            h *= 2;
        }
    }
#undef GRID_SYNC
    if (first == 0) {
        c.state[0] = m;
        c.state[1] = (uint32_t)cur;
        c.state[2] = (uint32_t)h;
        c.state[3] = (uint32_t)(h >> 32);
    }
}

// The last rounds (at most kGridFinishCap suffixes still tied), one workgroup, from the state the kernel above
// has left.
__global__ __launch_bounds__(kGridFinishThreads) void grid_finish_coop_kernel(GridCoopArgs c, uint32_t b1, uint32_t b2) {
    __shared__ GridFinishShared sh;
    const uint32_t m = c.state[0];
    if (m == 0u || *c.status) return;  // workgroup-uniform
    if (m > kGridFinishCap) {
        if (threadIdx.x == 0) atomicOr(c.status, kErrGridBarrier);
        return;
    }
    const uint64_t h = ((uint64_t)c.state[3] << 32) | c.state[2];
    grid_finish_body(c.a, c.id[c.state[1] & 1u], m, h, b1, b2, sh);
}

}  // namespace bwtgpu
