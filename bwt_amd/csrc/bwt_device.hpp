// bwt_device.hpp -- shared device-side helpers for the gfx950 BWT kernels.
//
// Wavefronts are 64 lanes wide on CDNA4; every cross-lane idiom here (ballot masks,
// shuffles, lane ids) is written for exactly that width.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bwtgpu {

constexpr int kWave = 64;
// LSD radix digit: 6 bits = 64 bins = one bin per lane of a wave.  Narrow digits keep the runs a
// staged tile writes per bin long (a 1024-element tile gives 16-element = 128-byte runs on
// average); measured on MI355X, 8-byte stores in runs of 64 B or more go ~4x faster per element
// than in runs of 32 B and ~14x faster than lane-scattered ones.
constexpr int kMaxDigitBits = 6;
constexpr int kBins = 1 << kMaxDigitBits;   // LDS histogram bins per wave

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }
__device__ __forceinline__ uint64_t lanemask_le() { return (2ull << lane_id()) - 1ull; }

// Bits needed to represent v (0 -> 0, 1 -> 1, 2..3 -> 2, ...).
__device__ __host__ __forceinline__ int bit_length(uint32_t v) {
    int b = 0;
    while (v) { ++b; v >>= 1; }
    return b;
}

// Lanes of this wavefront whose `valid` digit equals mine (bit i set <=> lane i is a
// peer).  One ballot per digit bit: the wave64 form of a match-any.  BITS is the digit
// width the caller guarantees (higher bits of d must be zero).
template <int BITS>
__device__ __forceinline__ uint64_t match_digit(uint32_t d, bool valid) {
    const uint64_t vm = __ballot(valid);
    uint32_t lo = (uint32_t)vm, hi = (uint32_t)(vm >> 32);
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
        const uint32_t x = (uint32_t)__builtin_amdgcn_sbfe((int)d, b, 1);  // all ones when my bit b is set
        const uint64_t bal = __ballot(x != 0u);
        lo &= ~((uint32_t)bal ^ x);  // keep lanes whose bit equals mine
        hi &= ~((uint32_t)(bal >> 32) ^ x);
    }
    return ((uint64_t)hi << 32) | lo;
}

// The same with a digit width known only at run time (wave-uniform): one ballot per bit that
// any digit can have set.
__device__ __forceinline__ uint64_t match_digit_bits(uint32_t d, bool valid, int bits) {
    const uint64_t vm = __ballot(valid);
    uint32_t lo = (uint32_t)vm, hi = (uint32_t)(vm >> 32);
#pragma unroll 1
    for (int b = 0; b < bits; ++b) {
        const uint32_t x = (uint32_t)__builtin_amdgcn_sbfe((int)d, b, 1);
        const uint64_t bal = __ballot(x != 0u);
        lo &= ~((uint32_t)bal ^ x);
        hi &= ~((uint32_t)(bal >> 32) ^ x);
    }
    return ((uint64_t)hi << 32) | lo;
}

// Keeps the compiler from moving memory operations across this point (used where lanes of
// one wave hand values to each other through LDS and the order of a read and a later write
// matters).
__device__ __forceinline__ void compiler_fence() { asm volatile("" ::: "memory"); }

// Loads and stores through a pointer that is known to point to device memory.  Inside a function
// that is not inlined into its kernel the compiler cannot tell, and a generic access is a FLAT
// instruction, which counts as an LDS operation as well: every wait for an LDS result then also
// waits for all such loads in flight, and loads issued ahead for the next tile hide nothing.
template <typename T>
__device__ __forceinline__ T load_global(const T* p) {
    return *(const __attribute__((address_space(1))) T*)p;
}
template <typename T>
__device__ __forceinline__ void store_global(T* p, T v) {
    *(__attribute__((address_space(1))) T*)p = v;
}

// Inclusive prefix sum across the 64 lanes of a wave (all of them active).  Six adds whose second
// operand comes through the data-parallel primitives of the vector ALU: inside each row of 16 lanes
// from 1, 2, 4 and 8 lanes below (a lane with nothing below it adds 0), then lane 15 of rows 0 and 2
// to the rows above them and lane 31 to rows 2 and 3.  (Through __shfl_up every step is an LDS
// crossbar instruction and a wait for it: a scan was 6 dependent LDS round trips.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
// The value one lane holds, for all lanes (`src` is the same in all of them): a read of the register
// through the scalar unit instead of __shfl's trip through the LDS crossbar.
__device__ __forceinline__ uint32_t wave_read(uint32_t v, int src) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, src);
}
__device__ __forceinline__ uint64_t wave_read(uint64_t v, int src) {
    return ((uint64_t)wave_read((uint32_t)(v >> 32), src) << 32) | wave_read((uint32_t)v, src);
}
// The value of the lane below (lane 0: `first`) / above (lane 63: `last`), all 64 lanes active.
__device__ __forceinline__ uint32_t wave_from_below(uint32_t v, uint32_t first) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138, 0xF, 0xF, false);  // wave_shr:1
}
__device__ __forceinline__ uint32_t wave_from_above(uint32_t v, uint32_t last) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)last, (int)v, 0x130, 0xF, 0xF, false);  // wave_shl:1
}
__device__ __forceinline__ uint64_t wave_from_below(uint64_t v, uint64_t first) {
    return ((uint64_t)wave_from_below((uint32_t)(v >> 32), (uint32_t)(first >> 32)) << 32) |
           wave_from_below((uint32_t)v, (uint32_t)first);
}
__device__ __forceinline__ uint64_t wave_from_above(uint64_t v, uint64_t last) {
    return ((uint64_t)wave_from_above((uint32_t)(v >> 32), (uint32_t)(last >> 32)) << 32) |
           wave_from_above((uint32_t)v, (uint32_t)last);
}
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    v += dpp_or_zero<0x111, 0xF>(v);  // row_shr:1
    v += dpp_or_zero<0x112, 0xF>(v);  // row_shr:2
    v += dpp_or_zero<0x114, 0xF>(v);  // row_shr:4
    v += dpp_or_zero<0x118, 0xF>(v);  // row_shr:8
    v += dpp_or_zero<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_or_zero<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}

// Exclusive prefix sum of one value per thread over the workgroup; `total` receives
// the workgroup sum.  s_wsum needs NT/64 words.  Contains two barriers.
template <int NT>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* s_wsum, uint32_t& total) {
    constexpr int W = NT / kWave;
    const int lane = lane_id(), w = wave_id();
    const uint32_t x = wave_inclusive_scan(v);
    if (lane == kWave - 1) s_wsum[w] = x;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const uint32_t t = s_wsum[i];
        if (i < w) pre += t;
        tot += t;
    }
    __syncthreads();
    total = tot;
    return pre + x - v;
}

// Splits m elements over the W waves of a workgroup in contiguous chunks whose
// length is a multiple of 64 (so every wave iteration is one aligned vector).
struct Chunking {
    uint32_t chunk;  // elements per wave
    uint32_t magic;  // ceil(2^32 / chunk)
    __device__ __forceinline__ void init(uint32_t m, int W) {
        uint32_t c = (m + (uint32_t)W - 1u) / (uint32_t)W;
        c = (c + 63u) & ~63u;
        if (c == 0) c = 64;
        chunk = c;
        magic = 0xFFFFFFFFu / c + 1u;
    }
    // pos / chunk without a hardware divide
    __device__ __forceinline__ uint32_t wave_of(uint32_t pos) const {
        uint32_t q = __umulhi(pos, magic);
        if (q * chunk > pos) --q;
        return q;
    }
};

// LSD digit plan for a K-bit key: P = ceil(K / kMaxDigitBits) passes of near-equal width.
struct DigitPlan {
    int P, base, rem;
    __device__ __forceinline__ void init(int K) {
        P = (K + kMaxDigitBits - 1) / kMaxDigitBits;
        if (P < 1) P = 1;
        base = K / P;
        rem = K % P;
    }
    __device__ __forceinline__ int bits(int p) const { return base + (p < rem ? 1 : 0); }
    __device__ __forceinline__ int shift(int p) const { return p * base + (p < rem ? p : rem); }
};

}  // namespace bwtgpu
