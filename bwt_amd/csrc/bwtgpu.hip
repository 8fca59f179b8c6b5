// bwtgpu.hip -- C ABI (include/bwtgpu.h) of the gfx950 BWT block codec: launch
// logic, host-buffer staging, delimiter search and the synthetic-input generator.
// The transform kernels themselves are in bwt_forward.hpp / bwt_inverse.hpp.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "bwt_forward.hpp"
#include "bwt_forward_grid.hpp"
#include "bwt_forward_grid_coop.hpp"
#include "bwt_forward_lds.hpp"
#include "bwt_forward_small.hpp"
#include "bwt_inverse.hpp"
#include "bwt_inverse_small.hpp"
#include "bwt_inverse_packed.hpp"
#include "bwt_inverse_t16.hpp"
#include "bwtgpu.h"

namespace {

using namespace bwtgpu;

constexpr int kFwdWideThreads = 1024;  // forward, few-block inputs: one workgroup per CU
constexpr size_t kWideMinBlock = 8192; // shorter blocks do not pay for the wide workgroup
constexpr int kFwdThreads = 256;   // forward: threads per workgroup (4 waves, each with an 8 KiB LDS staging tile)
constexpr int kInvThreads = 256;   // inverse table build: threads per workgroup
constexpr size_t kHeaderBytes = 512;  // workspace header: work queue, status words, phase clocks
constexpr size_t kBlockKernelMaxBlock = 2097151;  // forward: blocks above this only take the device-wide path
constexpr size_t kInvTiledMaxBlocks = 128;   // inverse: batches of at most this many blocks ...
constexpr size_t kInvTiledMinRows = 16384;   // ... of at least this many rows build their LF tables tile by tile

thread_local char g_err[512] = "";

// Scratch of bwtgpu_forward_batch_static, the entry point that allocates nothing: every GPU the
// library runs on holds its own copy of this array from the moment the code object is loaded until
// the process ends.  Host code never uses the array's name as an address -- it is a host-side
// stand-in there; ensure_device asks the runtime for each device's copy (hipGetSymbolAddress)
// and everything below goes through that pointer (Device::static_ws).
constexpr size_t kStaticScratchBytes = BWTGPU_STATIC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(BWTGPU_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                   \
    } while (0)

struct Device {
    int cus = 0;
    int fwd_wgs_per_cu = 1;
    int inv_wgs_per_cu = 1;
    int coop_wgs = 0;  // workgroups of grid_forward_coop_kernel that are resident together (0: no cooperative launches here)
    // grow-only buffers for the host-pointer entry points
    void* d_in = nullptr;   size_t in_cap = 0;
    void* d_out = nullptr;  size_t out_cap = 0;
    void* d_ws = nullptr;   size_t ws_cap = 0;
    void* d_aux = nullptr;  size_t aux_cap = 0;
    hipStream_t stream = nullptr;
    uint8_t* static_ws = nullptr;  // this device's copy of `scratch` (nullptr: the lookup failed, see static_ws_error)
    hipError_t static_ws_error = hipSuccess;
    // Pinned host words the device writes at the end of an inverse batch that went through the in-LDS inverse
    // first: the blocks that kernel did not finish.  0 = the table kernels found nothing to do, and the
    // next such batch launches them as small striding grids (a guess about the data: either way of
    // launching gives the same bytes).  kInvHints words, one per stream (by a hash of its handle), so that
    // two pipelines with different data on one device do not overturn each other's guess.
    uint32_t* inv_left = nullptr;
    uint32_t* inv_left_dev = nullptr;  // the same words as the device addresses them
    bool ready = false;
    std::mutex mu;          // one host-pointer call at a time per device
    std::mutex init_mu;     // guards first-use binding
};

constexpr int kInvHints = 16;  // (64 bytes of pinned memory)

std::mutex g_mu;
std::vector<Device*> g_dev;

// Binds device `device` on first use: properties, launch geometry, staging stream.
int ensure_device(int device);

// Fills the device table (g_mu held).  The table only grows here and is emptied by
// bwtgpu_shutdown, which must not run while another call is in flight.
int init_locked(int n_devices) {
    if (!g_dev.empty()) return (int)g_dev.size();
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(BWTGPU_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU path",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (n_devices > 0 && n_devices < count) count = n_devices;
    for (int i = 0; i < count; ++i) g_dev.push_back(new Device);
    return (int)g_dev.size();
}

int check_device(int device) {
    {
        std::lock_guard<std::mutex> lock(g_mu);
        if (g_dev.empty()) {
            int rc = init_locked(0);
            if (rc < 0) return rc;
        }
        if (device < 0 || device >= (int)g_dev.size())
            return fail(BWTGPU_ERR_NO_DEVICE, "device %d out of range (%d bound)", device, (int)g_dev.size());
    }
    return ensure_device(device);
}

int grow(void** p, size_t* cap, size_t need) {
    if (*cap >= need) return BWTGPU_OK;
    if (*p) HIP_TRY(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    need = (need + (1u << 20)) & ~(size_t)((1u << 20) - 1);
    HIP_TRY(hipMalloc(p, need));
    *cap = need;
    return BWTGPU_OK;
}

size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// A batch shorter than one block IS one block of its own length: block i lies at i * block_size
// whatever block_size is, and nothing else depends on it.  Shrinking block_size to the batch
// keeps any value the reference accepts (src/bwt.cpp:251 takes any positive number) usable.
size_t forward_block_size(size_t total_len, size_t block_size) {
    return std::min(block_size, std::max<size_t>(total_len, 1));
}
size_t inverse_block_size(size_t total_with_delims, size_t block_size) {
    return std::min(block_size, std::max<size_t>(total_with_delims, 2) - 1);
}

// ---- forward launch geometry ----------------------------------------------
struct FwdGeom {
    size_t nblocks;
    uint32_t ncap;
    size_t stride;   // bytes of scratch per workgroup
    uint32_t grid;
    bool wide;       // few blocks: 1024-thread workgroups, one per CU
    bool grid_path;  // few large blocks: all compute units work on the one batch
    size_t bytes;    // whole workspace
};

// ---- few large blocks: the device-wide path (bwt_forward_grid.hpp) --------------------------
// Measured on MI355X: the device-wide path takes about 0.15 ms plus 0.3 us per KiB of the whole
// batch (random bytes; text about three times that, and 0.1 ms for every further round), the
// batch kernel 0.2 ms plus 8.5 us per KiB of ONE block (text: 25 us) as long as every block has
// a workgroup of its own.  The former wins for up to a few dozen blocks unless they are short.
constexpr size_t kClassifyMinBlock = (size_t)512 << 10;  // batch kernels: blocks at least this long are checked for whole-block periods before they are sorted
// BWTGPU_CLASSIFY_MIN_BLOCK=<bytes> moves that threshold, and BWTGPU_NARROW=1 keeps batches of few blocks on the
// 256-thread kernels: how the tests send small batches through the classifying launch and both of its lists.
size_t classify_min_block() {
    const char* e = std::getenv("BWTGPU_CLASSIFY_MIN_BLOCK");
    if (e && *e) {
        const unsigned long long v = strtoull(e, nullptr, 10);
        if (v >= 32768) return (size_t)v;
    }
    return kClassifyMinBlock;
}
bool force_narrow() {
    const char* e = std::getenv("BWTGPU_NARROW");
    return e && !strcmp(e, "1");
}
constexpr size_t kGridMaxBlocks = 32;
constexpr size_t kGridMinBlock = 20000;      // bytes in the (longest) block
constexpr size_t kGridMaxRows = (size_t)1 << 31;

// BWTGPU_FORWARD_PATH=grid|block overrides the choice between the two forward paths (tests
// run every case through both); anything else, or unset, leaves it to the batch shape.
int forced_forward_path() {
    const char* e = std::getenv("BWTGPU_FORWARD_PATH");
    if (!e) return 0;
    if (!strcmp(e, "grid")) return 1;
    if (!strcmp(e, "block")) return 2;
    return 0;
}

// BWTGPU_FORWARD_LDS=0 keeps every block on the general batch kernels (tests run both ways).
bool forward_lds_enabled() {
    const char* e = std::getenv("BWTGPU_FORWARD_LDS");
    return !(e && !strcmp(e, "0"));
}

// BWTGPU_INVERSE_PATH=general keeps the inverse of small-alphabet blocks on the LF-table kernels,
// =packed lets the in-LDS inverse also take batches of few blocks (which the tiled table build serves
// otherwise); tests run every case all three ways.
int forced_inverse_path() {
    const char* e = std::getenv("BWTGPU_INVERSE_PATH");
    if (!e) return 0;
    if (!strcmp(e, "general")) return 1;
    if (!strcmp(e, "packed")) return 2;
    return 0;
}

struct GridGeom {
    size_t rows;
    size_t presence_off, count_off, hist_off, rec_off, rank_off, key_off[2], id_off[2], bytes;
};

GridGeom grid_geom(size_t total_len, size_t block_size, bool cyclic = false) {
    GridGeom g;
    g.rows = total_len + (cyclic ? 0 : (total_len + block_size - 1) / block_size);
    size_t off = kHeaderBytes;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off = round_up(off + bytes, 256);
        return at;
    };
    g.presence_off = take(32);
    g.count_off = take(8);
    g.hist_off = take((size_t)kGridBins * kGridMaxTiles * 4);
    g.rec_off = take((size_t)kGridMaxTiles * 16);
    g.rank_off = take(g.rows * 4);
    for (int i = 0; i < 2; ++i) g.key_off[i] = take(g.rows * 8);
    for (int i = 0; i < 2; ++i) g.id_off[i] = take(g.rows * 4);
    g.bytes = off;
    return g;
}

bool grid_shape(size_t total_len, size_t block_size) {
    const size_t nblocks = (total_len + block_size - 1) / block_size;
    return nblocks <= kGridMaxBlocks && std::min(total_len, block_size) >= kGridMinBlock;
}

FwdGeom fwd_geom(const Device& d, size_t total_len, size_t block_size) {
    FwdGeom g;
    g.nblocks = (total_len + block_size - 1) / block_size;
    const size_t longest = std::min(total_len, block_size);
    if (block_size > kBlockKernelMaxBlock) {  // only the device-wide path sorts blocks this long
        g.ncap = 0;
        g.stride = 0;
        g.grid = 0;
        g.wide = false;
        g.grid_path = total_len + g.nblocks <= kGridMaxRows;
        g.bytes = grid_geom(total_len, block_size).bytes;
        return g;
    }
    g.ncap = (uint32_t)round_up(longest + 1, 64);
    g.stride = round_up((size_t)g.ncap * 20 + bwtgpu::kBucketDirBytes, 256);
    const size_t max_wgs = (size_t)d.cus * (size_t)d.fwd_wgs_per_cu;
    g.grid = (uint32_t)std::max<size_t>(1, std::min(g.nblocks, max_wgs));
    // With fewer blocks than compute units most of the chip idles and a block's latency is the
    // whole cost: give every block a 1024-thread workgroup instead.
    g.wide = g.nblocks <= (size_t)d.cus && longest >= kWideMinBlock && !force_narrow();
    g.bytes = kHeaderBytes + g.stride * g.grid + 2 * round_up(g.nblocks * 4, 256);  // + the lists of blocks put off / left to the first launch
    // The device-wide path serves the batch itself or, when a caller sizes its workspace for a
    // full slab and then submits a shorter one, any prefix of it.
    const int forced = forced_forward_path();
    g.grid_path = total_len + g.nblocks <= kGridMaxRows && forced != 2 && (forced == 1 || grid_shape(total_len, block_size));
    if (forced != 2) {
        const size_t most = forced == 1 ? total_len : std::min(total_len, kGridMaxBlocks * block_size);
        if (forced == 1 || grid_shape(most, block_size)) g.bytes = std::max(g.bytes, grid_geom(most, block_size).bytes);
    }
    return g;
}

struct InvGeom {
    size_t nblocks;
    uint32_t tstride;
    uint32_t build_grid;
    uint32_t seg_cap, seg_wgs, chunk_cap, split_shift, slot, multi_min_rows;
    bool split;  // blocks are long enough for the segmented walk
    bool tiled;  // few large blocks: the LF tables are built by many workgroups per block
    bool wide;   // blocks of more than kNarrowMaxRows rows: 64-bit table words
    uint32_t tile_rows, tiles_per_block;
    size_t dacc_off, thist_off;
    uint32_t pack_wgs;  // workgroups of the in-LDS inverse of small-alphabet blocks (0: the batch's shape rules it out)
    size_t drow_off, mode_off, used_off, chain_off, seg_off, segrec_off, segovf_off, next_off, t_off, stage_off, pool_off, bytes;
};

// Multiplier of the splitter map row -> row * mul mod n of the inverse (bwt_inverse.hpp) and its
// inverse modulo n: mul / n near the golden ratio, coprime to n.
void splitter_map(uint32_t n, uint32_t* mul, uint32_t* inv) {
    *mul = *inv = 1;
    if (n < 3) return;
    uint64_t m = (uint64_t)((double)n * 0.6180339887498949);
    if (m < 1) m = 1;
    auto gcd = [](uint64_t x, uint64_t y) {
        while (y) {
            const uint64_t t = x % y;
            x = y;
            y = t;
        }
        return x;
    };
    while (gcd(m, n) != 1) ++m;  // reaches n - 1 at the latest
    // extended Euclid: m * x == 1 (mod n)
    int64_t r0 = n, r1 = (int64_t)m, x0 = 0, x1 = 1;
    while (r1) {
        const int64_t q = r0 / r1, r2 = r0 - q * r1, x2 = x0 - q * x1;
        r0 = r1; r1 = r2;
        x0 = x1; x1 = x2;
    }
    if (x0 < 0) x0 += n;
    *mul = (uint32_t)m;
    *inv = (uint32_t)x0;
}

InvGeom inv_geom(const Device& d, size_t total, size_t block_size) {
    InvGeom g;
    const size_t chunk = block_size + 1;
    g.nblocks = (total + chunk - 1) / chunk;
    const size_t longest = std::min(total, chunk);
    g.tstride = (uint32_t)round_up(longest, 64);
    const size_t max_wgs = (size_t)d.cus * (size_t)d.inv_wgs_per_cu;
    g.build_grid = (uint32_t)std::max<size_t>(1, std::min(g.nblocks, max_wgs));
    g.multi_min_rows = g.nblocks < kFewBlocks ? kMultiMinRowsFew : kMultiMinRows;
    g.split = longest >= g.multi_min_rows;
    // Splitter every 16 rows where the segment records of a block then still fit the LDS of the
    // stitch kernel and the block is long enough to pay for twice the segments (measured: 256 KiB
    // blocks gain, 64 KiB blocks lose); every 32 rows otherwise.
    g.split_shift = kMinSplitShift;
    if (longest < (size_t)192 << 10 || regular_splitters((uint32_t)longest, kMinSplitShift) + 1u > kStitchLdsSegs)
        g.split_shift = kMinSplitShift + 1u;
    // long blocks: no more segments than the stitch kernels hold super nodes for (and 64 K lanes
    // per block fill the chip anyway)
    while (regular_splitters((uint32_t)longest, g.split_shift) > kMaxSegments) ++g.split_shift;
    g.wide = longest > kNarrowMaxRows;
    g.slot = 4u << g.split_shift;
    g.seg_cap = g.split ? regular_splitters((uint32_t)longest, g.split_shift) + 1u : 1u;
    g.seg_wgs = (g.seg_cap + kSegThreads - 1) / kSegThreads;
    g.chunk_cap = g.split ? (uint32_t)(longest / g.slot) + 8u : 1u;
    // One workgroup builds a 1 MiB block's table in about 2 ms; with fewer blocks than compute
    // units the tables are built tile by tile instead.
    g.tiled = g.wide || (g.nblocks <= kInvTiledMaxBlocks && longest >= kInvTiledMinRows);
    const bool pack_range = longest >= kPackMinRows && longest <= kPackMaxRows;
    if (forced_inverse_path() == 2 && pack_range) g.tiled = false;
    g.tile_rows = (uint32_t)round_up(std::max<size_t>(4096, (longest + 255) / 256), 256);
    g.tiles_per_block = (uint32_t)((longest + g.tile_rows - 1) / g.tile_rows);
    size_t off = kHeaderBytes;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off = round_up(off + bytes, 256);
        return at;
    };
    g.dacc_off = take(g.tiled ? g.nblocks * 8 : 0);  // directly after the header: cleared with it
    g.thist_off = take(g.tiled ? g.nblocks * (size_t)g.tiles_per_block * 256 * 4 : 0);
    g.drow_off = take(g.nblocks * 4);
    g.mode_off = take(g.nblocks * 4);
    g.used_off = take(g.nblocks * 4);
    g.chain_off = take(g.nblocks * 8);
    g.seg_off = take(g.nblocks * g.seg_cap * 8);
    g.segrec_off = take(g.split ? g.nblocks * g.seg_cap * 8 : 0);
    g.segovf_off = take(g.nblocks * g.seg_cap * 4);
    g.next_off = take(g.nblocks * (size_t)g.chunk_cap * 4);
    g.stage_off = take(g.split ? g.nblocks * round_up(g.seg_cap, 64) * g.slot : 0);
    g.pool_off = take(g.split ? g.nblocks * (size_t)g.chunk_cap * g.slot : 0);
    g.t_off = take((size_t)g.tstride * (g.wide ? 8 : 4) * g.nblocks);
    g.pack_wgs = !g.tiled && pack_range && forced_inverse_path() != 1 ? (uint32_t)std::min<size_t>(g.nblocks, (size_t)d.cus) : 0u;
    g.bytes = off;
    return g;
}

// Workspace of a slab pipeline: enough for EVERY batch a slab of `slab_blocks` blocks can be
// submitted as.  inv_geom is not monotone in the number of blocks -- a batch of fewer than
// kFewBlocks blocks splits the walk of shorter blocks (segment records, staging and pool: about
// five times the block bytes), a batch of at most kInvTiledMaxBlocks blocks adds the tile counts --
// so the maximum is taken over the shapes on either side of each threshold.
size_t inv_slab_workspace(const Device& d, size_t slab_blocks, size_t block_size) {
    const size_t chunk = block_size + 1;
    size_t most = 0;
    for (size_t nb : {slab_blocks, std::min(slab_blocks, (size_t)kFewBlocks - 1), std::min(slab_blocks, kInvTiledMaxBlocks)})
        if (nb) most = std::max(most, inv_geom(d, nb * chunk, block_size).bytes);
    return most;
}

// ---- delimiter search --------------------------------------------------------
// 256-bit presence mask of the byte values in data[0..len): reference
// src/bwt.cpp:19-47 keeps the same information in an unordered_set.
__global__ __launch_bounds__(256) void presence_kernel(const uint8_t* __restrict__ data, size_t len,
                                                       uint32_t* __restrict__ presence) {
    __shared__ uint32_t s_mask[8];
    if (threadIdx.x < 8) s_mask[threadIdx.x] = 0;
    __syncthreads();
    uint32_t m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    // head bytes up to 16-byte alignment, then 16 bytes per lane, then the tail
    const size_t mis = (16 - ((uintptr_t)data & 15)) & 15;
    const size_t head = mis < len ? mis : len;
    const size_t nvec = (len - head) / 16;
    const uint4* v = reinterpret_cast<const uint4*>(data + head);
    auto mark = [&](uint32_t b) {
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] |= ((b >> 5) == (uint32_t)j) ? (1u << (b & 31)) : 0u;
    };
    for (size_t i = tid; i < nvec; i += nthreads) {
        const uint4 q = v[i];
        const uint32_t ws[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            mark(ws[j] & 255u);
            mark((ws[j] >> 8) & 255u);
            mark((ws[j] >> 16) & 255u);
            mark(ws[j] >> 24);
        }
    }
    const size_t tail0 = head + nvec * 16;
    for (size_t i = tid; i < head; i += nthreads) mark(data[i]);
    for (size_t i = tail0 + tid; i < len; i += nthreads) mark(data[i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        uint32_t x = m[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x |= __shfl_xor(x, o);
        if ((threadIdx.x & 63) == 0 && x) atomicOr(&s_mask[j], x);
    }
    __syncthreads();
    if (threadIdx.x < 8 && s_mask[threadIdx.x]) atomicOr(&presence[threadIdx.x], s_mask[threadIdx.x]);
}

// ---- synthetic inputs --------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix_word(uint64_t seed, uint64_t k) {
    uint64_t z = seed + (k + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint8_t synth_random255(uint64_t seed, uint64_t i) {
    const uint64_t w = splitmix_word(seed, i >> 2);
    return (uint8_t)((((w >> (16 * (i & 3))) & 0xFFFFull) * 255ull) >> 16);
}
__device__ __forceinline__ uint8_t synth_acgt(uint64_t seed, uint64_t i) {
    const uint64_t w = splitmix_word(seed, i >> 5);
    const uint32_t s = (uint32_t)(w >> (2 * (i & 31))) & 3u;
    return (uint8_t)((0x54474341u >> (8 * s)) & 255u);  // "ACGT"
}
__global__ __launch_bounds__(256) void synth_kernel(int kind, uint64_t seed, uint64_t offset,
                                                    uint8_t* __restrict__ out, size_t len,
                                                    const uint8_t* __restrict__ text, size_t text_len) {
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < len; j += nthreads) {
        const uint64_t i = offset + j;
        int k = kind;
        if (kind == 5) {
            const uint32_t which = (uint32_t)((i >> 20) & 3u);
            k = which == 0 ? 0 : which == 1 ? 1 : which == 2 ? 2 : 6;
        }
        uint8_t b;
        switch (k) {
            case 0: b = synth_random255(seed, i); break;
            case 1: b = synth_acgt(seed, i); break;
            case 2: b = (uint8_t)('a' + i % 7); break;
            case 3: b = (uint8_t)'a'; break;
            case 4: b = synth_random255(seed, i & 63); break;
            default: b = text[i % text_len]; break;
        }
        out[j] = b;
    }
}

uint32_t stream_grid(const Device& d, size_t items, int per_block) {
    const size_t want = (items + per_block - 1) / per_block;
    return (uint32_t)std::max<size_t>(1, std::min<size_t>(want, (size_t)d.cus * 8));
}

int ensure_device(int device) {
    Device* d = g_dev[device];
    std::lock_guard<std::mutex> lock(d->init_mu);
    if (d->ready) return BWTGPU_OK;
    hipDeviceProp_t prop;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    d->cus = prop.multiProcessorCount;
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, bwt_forward_kernel<kFwdThreads, false>, kFwdThreads, 0) ==
            hipSuccess && occ > 0)
        d->fwd_wgs_per_cu = occ;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, bwt_inverse_build_kernel<kInvThreads>, kInvThreads, 0) ==
            hipSuccess && occ > 0)
        d->inv_wgs_per_cu = occ;
    {
        int coop = 0;
        if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, device) == hipSuccess && coop &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, grid_forward_coop_kernel, kGridThreads, 0) == hipSuccess && occ > 0)
            d->coop_wgs = (int)std::min<size_t>((size_t)occ * (size_t)d->cus, kGridMaxTiles);
        (void)hipGetLastError();
    }
    HIP_TRY(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
    // this device's copy of the static scratch array (the device is current here)
    void* sp = nullptr;
    d->static_ws_error = hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch));
    d->static_ws = d->static_ws_error == hipSuccess ? (uint8_t*)sp : nullptr;
    if (d->static_ws_error != hipSuccess) (void)hipGetLastError();  // the failure stays with bwtgpu_forward_batch_static, which reports it
    if (hipHostMalloc((void**)&d->inv_left, 64, hipHostMallocPortable | hipHostMallocMapped) == hipSuccess &&
        hipHostGetDevicePointer((void**)&d->inv_left_dev, d->inv_left, 0) == hipSuccess) {
        for (int i = 0; i < kInvHints; ++i) d->inv_left[i] = 0xFFFFFFFFu;
    } else {
        if (d->inv_left) (void)hipHostFree(d->inv_left);
        d->inv_left = d->inv_left_dev = nullptr;
        (void)hipGetLastError();
    }
    d->ready = true;
    return BWTGPU_OK;
}

// ---- forward, device-wide path: one batch of few large blocks --------------------------------
// Pinned words the calling thread reads device results through (byte presence, suffixes left).
uint32_t* pinned_words() {
    thread_local uint32_t* p = nullptr;
    if (!p && hipHostMalloc((void**)&p, 64, hipHostMallocPortable) != hipSuccess) p = nullptr;
    return p;
}

// Elements per workgroup of a sort or rank step over m elements: one workgroup per compute unit
// while that keeps tiles short (every scatter workgroup reads one count per digit and tile
// before it starts), up to four per compute unit for long lists.
uint32_t grid_tile(uint32_t m) {
    size_t tile = round_up(std::max<size_t>(((size_t)m + 255) / 256, 1), kGridChunk);
    if (tile > 65536) tile = round_up(std::max<size_t>(65536, ((size_t)m + kGridMaxTiles - 1) / kGridMaxTiles), kGridChunk);
    return (uint32_t)tile;
}

// BWTGPU_GRID_SYNC=host keeps the device-wide path driven from the host (a launch per step, a read-back
// per round: the call then blocks); the default is the resident kernel (bwt_forward_grid_coop.hpp).
bool grid_host_driven() {
    const char* e = std::getenv("BWTGPU_GRID_SYNC");
    return e && !strcmp(e, "host");
}

// The device-wide path as two launches, nothing read back: returns BWTGPU_OK once they are enqueued, or
// 1 if this device cannot launch the resident kernel (the caller then drives the steps from the host).
int forward_grid_coop(const Device& d, const uint8_t* d_in, size_t total_len, size_t block_size, uint8_t delimiter,
                      uint8_t* d_out, void* d_workspace, hipStream_t s) {
    if (d.coop_wgs <= 0) return 1;
    const GridGeom g = grid_geom(total_len, block_size);
    uint8_t* ws = (uint8_t*)d_workspace;
    GridCoopArgs c;
    c.a.in = d_in;
    c.a.out = d_out;
    c.a.total_len = total_len;
    c.a.block_size = (uint32_t)block_size;
    c.a.nblocks = (uint32_t)((total_len + block_size - 1) / block_size);
    c.a.delim = delimiter;
    c.a.rows = (uint32_t)g.rows;
    c.a.rank = (uint32_t*)(ws + g.rank_off);
    c.a.presence = (uint32_t*)(ws + g.presence_off);
    c.a.tile_hist = (uint32_t*)(ws + g.hist_off);
    c.a.tile_rec = (uint32_t*)(ws + g.rec_off);
    c.a.count = (uint32_t*)(ws + g.count_off);
    c.a.cyclic = 0;
    for (int i = 0; i < 2; ++i) {
        c.key[i] = (uint64_t*)(ws + g.key_off[i]);
        c.id[i] = (uint32_t*)(ws + g.id_off[i]);
    }
    c.status = (uint32_t*)(ws + 4);   // the header (cleared by the caller): error flags, barrier words, state words
    c.bar = (uint32_t*)(ws + 384);
    c.state = (uint32_t*)(ws + 480);
    HIP_TRY(hipMemsetAsync(ws + g.presence_off, 0, 32, s));
    HIP_TRY(hipMemsetAsync(ws + g.count_off, 0, 8, s));
    const uint32_t tile = grid_tile_size(c.a.rows);
    size_t want = (c.a.rows + tile - 1) / tile;
    if (const char* e = std::getenv("BWTGPU_COOP_WGS")) want = std::min<size_t>(want, (size_t)std::max(1, atoi(e)));  // experiments
    const uint32_t nwg = (uint32_t)std::max<size_t>(1, std::min<size_t>((size_t)d.coop_wgs, want));
    void* args[] = {&c};
    const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(grid_forward_coop_kernel), dim3(nwg),
                                                    dim3(kGridThreads), args, 0, s);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return 1;
    }
    const size_t longest = std::min(total_len, block_size) + 1;
    hipLaunchKernelGGL(grid_finish_coop_kernel, dim3(1), dim3(kGridFinishThreads), 0, s, c, (uint32_t)bit_length(c.a.rows - 1u),
                       (uint32_t)bit_length((uint32_t)longest));
    HIP_TRY(hipGetLastError());
    return BWTGPU_OK;
}

// d_rows != nullptr: rotations instead of suffixes (no delimiter; d_out takes total_len bytes, d_rows the row of
// every block's rotation 0).  That variant is always driven from the host.
// may_block: the caller is going to wait for the result anyway (a host-pointer call, a timing call, a
// submit-and-collect): the host-driven sequence then serves it sooner than the resident kernel (kernel
// boundaries cost less than grid barriers).
int forward_grid(const Device& d, const uint8_t* d_in, size_t total_len, size_t block_size, uint8_t delimiter,
                 uint8_t* d_out, void* d_workspace, hipStream_t s, uint32_t* d_rows = nullptr, bool may_block = false) {
    const bool cyclic = d_rows != nullptr;
    if (!cyclic && !may_block && !grid_host_driven()) {
        const int rc = forward_grid_coop(d, d_in, total_len, block_size, delimiter, d_out, d_workspace, s);
        if (rc <= 0) return rc;
    }
    const GridGeom g = grid_geom(total_len, block_size, cyclic);
    uint8_t* ws = (uint8_t*)d_workspace;
    uint32_t* host = pinned_words();
    if (!host) return fail(BWTGPU_ERR_HIP, "hipHostMalloc of the result words failed");
    GridArgs a;
    a.in = d_in;
    a.out = d_out;
    a.total_len = total_len;
    a.block_size = (uint32_t)block_size;
    a.nblocks = (uint32_t)((total_len + block_size - 1) / block_size);
    a.delim = delimiter;
    a.rows = (uint32_t)g.rows;
    a.rank = (uint32_t*)(ws + g.rank_off);
    a.presence = (uint32_t*)(ws + g.presence_off);
    a.tile_hist = (uint32_t*)(ws + g.hist_off);
    a.tile_rec = (uint32_t*)(ws + g.rec_off);
    a.count = (uint32_t*)(ws + g.count_off);
    a.cyclic = cyclic ? 1u : 0u;
    uint64_t* key[2] = {(uint64_t*)(ws + g.key_off[0]), (uint64_t*)(ws + g.key_off[1])};
    uint32_t* id[2] = {(uint32_t*)(ws + g.id_off[0]), (uint32_t*)(ws + g.id_off[1])};

    // byte values in the batch -> symbol width and how many symbols the first key holds
    HIP_TRY(hipMemsetAsync(ws + g.presence_off, 0, 32, s));
    HIP_TRY(hipMemsetAsync(ws + g.count_off, 0, 8, s));
    hipLaunchKernelGGL(presence_kernel, dim3(stream_grid(d, total_len, 256 * 64)), dim3(256), 0, s, d_in, total_len,
                       (uint32_t*)(ws + g.presence_off));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(host, ws + g.presence_off, 32, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (!cyclic) host[delimiter >> 5] |= 1u << (delimiter & 31);
    uint32_t sigma = 0;
    for (int i = 0; i < 8; ++i) sigma += (uint32_t)__builtin_popcount(host[i]);
    const int sbits = bit_length(sigma);                       // codes 1 .. sigma
    const int bbits = bit_length(a.nblocks - 1u);
    const size_t longest = std::min(total_len, block_size) + (cyclic ? 0 : 1);  // symbols of the longest block, delimiter included
    int m0 = std::max(1, (40 - bbits) / sbits);                // five 8-bit passes
    m0 = (int)std::min<size_t>((size_t)m0, longest);
    const int b1 = bit_length(a.rows - 1u), b2 = bit_length((uint32_t)longest);

    int cur = 0;  // key[cur] / id[cur] hold the list the next step reads
    uint32_t runs_settled = 0;  // suffixes the periodic-run step has settled so far (device counter, read every round)
    auto sort = [&](uint32_t m, int bits) -> int {
        const uint32_t tile = grid_tile(m), ntiles = (m + tile - 1) / tile;
        for (int shift = 0; shift < bits; shift += kGridDigitBits) {
            const uint32_t mask = (1u << std::min(kGridDigitBits, bits - shift)) - 1u;
            hipLaunchKernelGGL(grid_hist_kernel, dim3(ntiles), dim3(kGridThreads), 0, s, key[cur], m, tile,
                               (uint32_t)shift, mask, a.tile_hist);
            hipLaunchKernelGGL(grid_scatter_kernel, dim3(ntiles), dim3(kGridThreads), 0, s, key[cur], id[cur],
                               key[cur ^ 1], id[cur ^ 1], m, tile, ntiles, (uint32_t)shift, mask, a.tile_hist);
            HIP_TRY(hipGetLastError());
            cur ^= 1;
        }
        return BWTGPU_OK;
    };
    // ranks of the sorted list in key[cur] / id[cur]; the suffixes still tied go to id[cur ^ 1]
    auto rerank = [&](uint32_t m, bool round0, uint32_t* left) -> int {
        const uint32_t tile = grid_tile(m), ntiles = (m + tile - 1) / tile;
        hipLaunchKernelGGL(grid_rank_reduce_kernel, dim3(ntiles), dim3(kGridThreads), 0, s, key[cur], m, tile,
                           (uint32_t)b2, round0 ? 1 : 0, a.tile_rec);
        hipLaunchKernelGGL(grid_rank_scan_kernel, dim3(1), dim3(kGridScanThreads), 0, s, a.tile_rec, ntiles, a.count);
        hipLaunchKernelGGL(grid_rank_apply_kernel, dim3(ntiles), dim3(kGridThreads), 0, s, a, key[cur], id[cur], m,
                           tile, (uint32_t)b2, round0 ? 1 : 0, a.tile_rec, id[cur ^ 1]);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(host, a.count, 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        *left = host[0];
        runs_settled = host[1];
        cur ^= 1;  // the tied suffixes are the next list
        return BWTGPU_OK;
    };

    int rc;
    uint32_t m = a.rows;
    hipLaunchKernelGGL(grid_keys0_kernel, dim3(stream_grid(d, m, kGridThreads * 4)), dim3(kGridThreads), 0, s, a,
                       key[cur], id[cur], (uint32_t)m0, (uint32_t)sbits);
    HIP_TRY(hipGetLastError());
    if ((rc = sort(m, bbits + m0 * sbits))) return rc;
    if ((rc = rerank(m, true, &m))) return rc;
    const bool trace = std::getenv("BWTGPU_TRACE") != nullptr;  // debugging aid: suffixes left after every round
    if (trace) fprintf(stderr, "bwtgpu grid forward: rows %u, %d symbols of %d bits first, %u tied\n", a.rows, m0, sbits, m);
    uint64_t h = (uint64_t)m0;
    uint32_t m_before = a.rows;  // the list's length one round earlier
    bool runs_tried = false;
    uint32_t runs_m = 0, runs_wait = 0, runs_backoff = 1;
    for (int round = 0; m; ++round) {
        if (round >= 64 || m > a.rows) return fail(BWTGPU_ERR_INTERNAL, "device-wide suffix sort did not converge");
        if (cyclic && h >= longest && m > kGridFinishCap) {  // what is still tied now are equal rotations: by position
            const dim3 grid(stream_grid(d, m, kGridThreads * 4)), block(kGridThreads);
            uint32_t* place = (uint32_t*)key[cur];  // idle between the rank step and the next keys
            hipLaunchKernelGGL(grid_cyclic_settle_find_kernel, grid, block, 0, s, a, id[cur], m, place);
            hipLaunchKernelGGL(grid_cyclic_settle_apply_kernel, grid, block, 0, s, a, id[cur], m, place);
            HIP_TRY(hipGetLastError());
            break;
        }
        // The periodic-run step: when a round has left the list nearly as long as it was (right after
        // the first sort: only if hardly anything is settled), and less and less often while it finds little.
        const bool stalled = round == 0 ? (uint64_t)m * 50 > (uint64_t)m_before * 49 : (uint64_t)m * 10 > (uint64_t)m_before * 9;
        if (runs_wait) --runs_wait;
        else if (m > kGridFinishCap && stalled && !cyclic) {  // (the run step reasons about a block's end, which rotations lack)
            const dim3 grid(stream_grid(d, m, kGridThreads * 4)), block(kGridThreads);
            uint2* tmm = (uint2*)key[cur];        // both key buffers are idle between the rank step and the next keys
            uint2* tcs = (uint2*)key[cur ^ 1];
            hipLaunchKernelGGL(grid_runs_clear_kernel, grid, block, 0, s, a, id[cur], m, tmm, tcs);
            hipLaunchKernelGGL(grid_runs_stats_kernel, grid, block, 0, s, a, id[cur], m, tmm, tcs);
            hipLaunchKernelGGL(grid_runs_check_kernel, grid, block, 0, s, a, id[cur], m, h, tmm, tcs);
            hipLaunchKernelGGL(grid_runs_break_kernel, grid, block, 0, s, a, id[cur], m, h, tmm, tcs);
            hipLaunchKernelGGL(grid_runs_apply_kernel, grid, block, 0, s, a, id[cur], m, tmm, tcs);
            HIP_TRY(hipGetLastError());
            runs_tried = true;
            runs_m = m;
        }
        m_before = m;
        if (m <= kGridFinishCap) {  // few left: one workgroup runs the remaining rounds by itself
            hipLaunchKernelGGL(grid_finish_kernel, dim3(1), dim3(kGridFinishThreads), 0, s, a, id[cur], m, h, (uint32_t)b1,
                               (uint32_t)b2);
            HIP_TRY(hipGetLastError());
            break;
        }
        hipLaunchKernelGGL(grid_keys_kernel, dim3(stream_grid(d, m, kGridThreads * 4)), dim3(kGridThreads), 0, s, a,
                           id[cur], m, h, (uint32_t)b2, key[cur]);
        HIP_TRY(hipGetLastError());
        if ((rc = sort(m, b1 + b2))) return rc;
        const uint32_t settled_before = runs_settled;
        if ((rc = rerank(m, false, &m))) return rc;
        if (runs_tried) {  // did it pay?  if not, wait twice as many rounds before the next try
            runs_tried = false;
            if (runs_settled - settled_before < runs_m / 16u) runs_wait = runs_backoff, runs_backoff *= 2;
            else runs_backoff = 1;
        }
        if (trace) fprintf(stderr, "bwtgpu grid forward: depth %llu, %u tied\n", (unsigned long long)(2 * h), m);
        h *= 2;
    }
    if (cyclic) {
        hipLaunchKernelGGL(grid_cyclic_origin_kernel, dim3(stream_grid(d, a.nblocks, kGridThreads)), dim3(kGridThreads), 0, s, a, d_rows);
        HIP_TRY(hipGetLastError());
    }
    return BWTGPU_OK;
}

// ---- forward, one workgroup per block: launches over a workspace laid out as header, g.grid
// slices of g.stride bytes, list of put-off blocks ---------------------------------------------
int forward_blocks(const Device& d, const FwdGeom& g, const uint8_t* d_in, size_t total_len, size_t block_size,
                   uint8_t delimiter, uint8_t* d_out, uint8_t* ws, hipStream_t s) {
    HIP_TRY(hipMemsetAsync(ws, 0, kHeaderBytes, s));
    FwdArgs a;
    a.in = d_in;
    a.out = d_out;
    a.total_len = total_len;
    a.block_size = (uint32_t)block_size;
    a.nblocks = (uint32_t)g.nblocks;
    a.ncap = g.ncap;
    a.delim = delimiter;
    a.ws = ws + kHeaderBytes;
    a.ws_stride = g.stride;
    a.queue = (uint32_t*)ws;
    a.deferred = (uint32_t*)(ws + kHeaderBytes + g.stride * g.grid);
    a.normal = a.deferred + round_up(g.nblocks * 4, 256) / 4;
    // Blocks whose packed text fits LDS (at most four text symbols up to 256 KiB, any bytes up to 64 KiB)
    // go through bwt_forward_lds_kernel first; the blocks it does not finish are listed in a.normal, which
    // the launches below then take their work from.
    const bool lds_first = forward_lds_enabled() && block_size >= kLdsMinLen && block_size <= kLdsMaxLen2;
    if (lds_first) {
        hipLaunchKernelGGL(bwt_forward_lds_kernel, dim3((uint32_t)std::min<size_t>(g.grid, (size_t)d.cus)), dim3(kLdsThreads), 0, s, a);
        HIP_TRY(hipGetLastError());
    }
    if (block_size + 1 <= (size_t)kSmallCap) {  // small blocks: one wave per block, everything in LDS
        const size_t wgs = (g.nblocks + kSmallWaves - 1) / kSmallWaves;
        const uint32_t grid = (uint32_t)std::max<size_t>(1, std::min(wgs, (size_t)d.cus * 3));
        hipLaunchKernelGGL(bwt_forward_small_kernel, dim3(grid), dim3(kSmallWaves * kWave), 0, s, a);
    } else if (g.wide) {
        hipLaunchKernelGGL((bwt_forward_kernel<kFwdWideThreads, false>), dim3(g.grid), dim3(kFwdWideThreads), 0, s, a, lds_first ? 2 : 0);
        hipLaunchKernelGGL((bwt_forward_kernel<kFwdWideThreads, true>), dim3(g.grid), dim3(kFwdWideThreads), 0, s, a, 1);
    } else {
        // the common case first; the blocks it puts off (periodic runs, dominant keys) in a second
        // launch of the full kernel, which finds an empty list otherwise
        static const bool heavy_only = std::getenv("BWTGPU_PHASES") && !strcmp(std::getenv("BWTGPU_PHASES"), "heavy_only");
        if (heavy_only) {  // debugging aid: every block through the full kernel
            hipLaunchKernelGGL((bwt_forward_kernel<kFwdThreads, true>), dim3(g.grid), dim3(kFwdThreads), 0, s, a, 0);
            HIP_TRY(hipGetLastError());
            return BWTGPU_OK;
        }
        // long blocks: whole-block periods are looked for up front (bwt_forward_classify_kernel)
        const bool classify = block_size >= classify_min_block() && !lds_first;
        if (classify) hipLaunchKernelGGL(bwt_forward_classify_kernel<256>, dim3((uint32_t)std::min<size_t>(g.nblocks, (size_t)d.cus * 8)), dim3(256), 0, s, a);
        hipLaunchKernelGGL((bwt_forward_kernel<kFwdThreads, false>), dim3(g.grid), dim3(kFwdThreads), 0, s, a, classify || lds_first ? 2 : 0);
        // debugging aid: BWTGPU_PHASES=second makes the phase clocks in the header cover the second launch only
        static const bool second_only = std::getenv("BWTGPU_PHASES") && !strcmp(std::getenv("BWTGPU_PHASES"), "second");
        if (second_only) HIP_TRY(hipMemsetAsync(ws + 64, 0, kHeaderBytes - 64, s));
        hipLaunchKernelGGL((bwt_forward_kernel<kFwdThreads, true>), dim3(g.grid), dim3(kFwdThreads), 0, s, a, 1);
    }
    HIP_TRY(hipGetLastError());
    return BWTGPU_OK;
}

// ---- slab pipeline ------------------------------------------------------------
struct PipeSlot {
    uint8_t* h_in = nullptr;
    uint8_t* h_out = nullptr;
    uint32_t* h_lens = nullptr;   // inverse: per-block text lengths; forward: [0..1] status words
    uint8_t* d_in = nullptr;
    uint8_t* d_out = nullptr;
    uint32_t* d_lens = nullptr;
    void* d_ws = nullptr;
    size_t ws_cap = 0;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    size_t in_bytes = 0;
    size_t resident = 0;  // bytes of h_in already uploaded by bwtgpu_pipeline_upload_presence
    hipEvent_t staged = nullptr;  // bwtgpu_pipeline_stage: the upload out of h_in has finished
    bool staging = false;
    bool busy = false;
};

}  // namespace

struct bwtgpu_pipeline {
    int device = 0;
    int direction = 0;
    size_t block_size = 0;
    size_t slab_blocks = 0;
    size_t in_cap = 0, out_cap = 0, ws_bytes = 0;
    uint8_t delimiter = 0;
    std::vector<PipeSlot> slots;
    // bwtgpu_pipeline_reserve: uploaded slabs kept on the device (slab k at d_resident + k * in_cap) and the
    // byte values seen in them so far
    uint8_t* d_resident = nullptr;
    size_t resident_slabs = 0;
    std::vector<size_t> resident_bytes;
    uint32_t* d_presence = nullptr;
};

namespace {

void pipeline_free(bwtgpu_pipeline* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    for (PipeSlot& s : p->slots) {
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        if (s.h_in) (void)hipHostFree(s.h_in);
        if (s.h_out) (void)hipHostFree(s.h_out);
        if (s.h_lens) (void)hipHostFree(s.h_lens);
        if (s.d_in) (void)hipFree(s.d_in);
        if (s.d_out) (void)hipFree(s.d_out);
        if (s.d_lens) (void)hipFree(s.d_lens);
        if (s.d_ws) (void)hipFree(s.d_ws);
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.staged) (void)hipEventDestroy(s.staged);
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    if (p->d_resident) (void)hipFree(p->d_resident);
    if (p->d_presence) (void)hipFree(p->d_presence);
    delete p;
}

int pipeline_alloc(bwtgpu_pipeline* p, int depth) {
    HIP_TRY(hipSetDevice(p->device));
    p->slots.resize((size_t)depth);
    for (PipeSlot& s : p->slots) {
        HIP_TRY(hipHostMalloc((void**)&s.h_in, p->in_cap, hipHostMallocDefault));
        HIP_TRY(hipHostMalloc((void**)&s.h_out, p->out_cap, hipHostMallocDefault));
        HIP_TRY(hipHostMalloc((void**)&s.h_lens, (p->slab_blocks + 8) * 4, hipHostMallocDefault));
        HIP_TRY(hipMalloc((void**)&s.d_in, p->in_cap));
        HIP_TRY(hipMalloc((void**)&s.d_out, p->out_cap));
        HIP_TRY(hipMalloc((void**)&s.d_lens, (p->slab_blocks + 8) * 4));
        HIP_TRY(hipMalloc(&s.d_ws, p->ws_bytes));
        s.ws_cap = p->ws_bytes;
        HIP_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s.staged, hipEventDisableTiming));
    }
    return BWTGPU_OK;
}

}  // namespace

// =============================================================================
extern "C" {

const char* bwtgpu_last_error(void) { return g_err; }

int bwtgpu_init(int n_devices) {
    int count;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        count = init_locked(n_devices);
    }
    // Bind every device now -- stream, pinned word, occupancy queries, the address of the static scratch --
    // so that no later entry point has anything left to create (bwtgpu_forward_batch_static promises to
    // allocate nothing, also as the first call after this one).
    for (int i = 0; i < count; ++i) {
        const int rc = ensure_device(i);
        if (rc) return rc;
    }
    return count;
}

void bwtgpu_shutdown(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    for (size_t i = 0; i < g_dev.size(); ++i) {
        Device* d = g_dev[i];
        if (hipSetDevice((int)i) != hipSuccess) continue;
        // a batch still queued on a caller's stream writes the pinned word below at its end
        if (d->ready) (void)hipDeviceSynchronize();
        if (d->d_in) (void)hipFree(d->d_in);
        if (d->d_out) (void)hipFree(d->d_out);
        if (d->d_ws) (void)hipFree(d->d_ws);
        if (d->d_aux) (void)hipFree(d->d_aux);
        if (d->stream) (void)hipStreamDestroy(d->stream);
        if (d->inv_left) (void)hipHostFree(d->inv_left);
        delete d;
    }
    g_dev.clear();
}

int bwtgpu_device_count(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    return (int)g_dev.size();
}

size_t bwtgpu_num_blocks(size_t total_len, size_t block_size) {
    return block_size ? (total_len + block_size - 1) / block_size : 0;
}

size_t bwtgpu_forward_output_bytes(size_t total_len, size_t block_size) {
    return total_len + bwtgpu_num_blocks(total_len, block_size);
}

size_t bwtgpu_forward_workspace_bytes(int device, size_t total_len, size_t block_size) {
    if (block_size) block_size = forward_block_size(total_len, block_size);
    if (check_device(device) != BWTGPU_OK || block_size == 0 || block_size > BWTGPU_MAX_BLOCK_SIZE) return 0;
    return fwd_geom(*g_dev[device], total_len, block_size).bytes;
}

size_t bwtgpu_inverse_workspace_bytes(int device, size_t total, size_t block_size) {
    if (block_size) block_size = inverse_block_size(total, block_size);
    if (check_device(device) != BWTGPU_OK || block_size == 0 || block_size > BWTGPU_MAX_BLOCK_SIZE) return 0;
    return inv_geom(*g_dev[device], total, block_size).bytes;
}

static int forward_device(int device, const uint8_t* d_in, size_t total_len, size_t block_size, uint8_t delimiter,
                          uint8_t* d_out, void* d_workspace, size_t workspace_bytes, void* stream, bool may_block);

int bwtgpu_forward_batch_device(int device, const uint8_t* d_in, size_t total_len, size_t block_size,
                                uint8_t delimiter, uint8_t* d_out, void* d_workspace, size_t workspace_bytes,
                                void* stream) {
    return forward_device(device, d_in, total_len, block_size, delimiter, d_out, d_workspace, workspace_bytes, stream, false);
}

static int forward_device(int device, const uint8_t* d_in, size_t total_len, size_t block_size, uint8_t delimiter,
                          uint8_t* d_out, void* d_workspace, size_t workspace_bytes, void* stream, bool may_block) {
    int rc = check_device(device);
    if (rc) return rc;
    if (block_size == 0) return fail(BWTGPU_ERR_INVALID, "block_size must be greater than 0");
    block_size = forward_block_size(total_len, block_size);
    if (block_size > BWTGPU_MAX_BLOCK_SIZE)
        return fail(BWTGPU_ERR_BLOCK_TOO_LARGE, "block_size %zu exceeds the supported maximum %zu", block_size,
                    (size_t)BWTGPU_MAX_BLOCK_SIZE);
    if (total_len == 0) return BWTGPU_OK;
    if (!d_in || !d_out || !d_workspace) return fail(BWTGPU_ERR_INVALID, "null device pointer");
    const Device& d = *g_dev[device];
    const FwdGeom g = fwd_geom(d, total_len, block_size);
    if (workspace_bytes < g.bytes)
        return fail(BWTGPU_ERR_WORKSPACE, "workspace of %zu bytes given, %zu needed", workspace_bytes, g.bytes);
    if (g.nblocks > 0xFFFFFFF0ull) return fail(BWTGPU_ERR_INVALID, "too many blocks in one batch");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(device));
    if (g.grid_path) {
        HIP_TRY(hipMemsetAsync(d_workspace, 0, kHeaderBytes, s));
        return forward_grid(d, d_in, total_len, block_size, delimiter, d_out, d_workspace, s, nullptr, may_block);
    }
    if (block_size > kBlockKernelMaxBlock)
        return fail(BWTGPU_ERR_INVALID, "a batch of blocks above %zu bytes may hold at most 2^31 bytes (%zu given): split it",
                    kBlockKernelMaxBlock, total_len);
    return forward_blocks(d, g, d_in, total_len, block_size, delimiter, d_out, (uint8_t*)d_workspace, s);
}

int bwtgpu_forward_batch_static(int device, const uint8_t* d_in, size_t total_len, size_t block_size,
                                uint8_t delimiter, uint8_t* d_out, uint32_t* d_status, void* stream) {
    int rc;
    {   // never binds a device itself (that creates a stream and pinned memory): bwtgpu_init has done it
        std::lock_guard<std::mutex> lock(g_mu);
        if (device < 0 || device >= (int)g_dev.size())
            return fail(BWTGPU_ERR_NO_DEVICE, "device %d is not bound (%d bound): call bwtgpu_init first", device, (int)g_dev.size());
        if (!g_dev[device]->ready)
            return fail(BWTGPU_ERR_NO_DEVICE, "device %d is not bound: call bwtgpu_init first", device);
    }
    if (block_size == 0) return fail(BWTGPU_ERR_INVALID, "block_size must be greater than 0");
    block_size = forward_block_size(total_len, block_size);
    if (block_size > kBlockKernelMaxBlock)
        return fail(BWTGPU_ERR_BLOCK_TOO_LARGE, "block_size %zu exceeds what the static scratch serves (%zu)", block_size,
                    kBlockKernelMaxBlock);
    const Device& d = *g_dev[device];
    if (!d.static_ws)
        return fail(BWTGPU_ERR_HIP, "hipGetSymbolAddress of the static scratch failed on device %d: %s", device,
                    hipGetErrorString(d.static_ws_error));
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(device));
    if (total_len == 0) {
        if (d_status) HIP_TRY(hipMemsetAsync(d_status, 0, sizeof(uint32_t), s));
        return BWTGPU_OK;
    }
    if (!d_in || !d_out) return fail(BWTGPU_ERR_INVALID, "null device pointer");
    // The one-workgroup-per-block kernels with as many workgroups as the fixed scratch holds slices
    // for (a slice is 20 bytes per block byte); the workgroups pull blocks from a queue, so fewer of
    // them only means less of the chip is used.
    FwdGeom g = fwd_geom(d, total_len, block_size);
    if (g.nblocks > 0xFFFFFFF0ull) return fail(BWTGPU_ERR_INVALID, "too many blocks in one batch");
    const size_t fixed = kHeaderBytes + 2 * round_up(g.nblocks * 4, 256);  // header and the two block lists
    const size_t slices = fixed < kStaticScratchBytes ? (kStaticScratchBytes - fixed) / g.stride : 0;
    if (slices == 0)
        return fail(BWTGPU_ERR_WORKSPACE,
                    "batch of %zu blocks of %zu bytes exceeds the static scratch of %zu bytes (needs %zu for its block list "
                    "and %zu per resident block)", g.nblocks, block_size, kStaticScratchBytes, fixed, g.stride);
    g.grid = (uint32_t)std::min<size_t>(g.grid, slices);
    g.wide = g.grid <= (uint32_t)d.cus && std::min(total_len, block_size) >= kWideMinBlock;
    g.grid_path = false;
    rc = forward_blocks(d, g, d_in, total_len, block_size, delimiter, d_out, d.static_ws, s);
    if (rc) return rc;
    if (d_status) HIP_TRY(hipMemcpyAsync(d_status, d.static_ws + 4, sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    return BWTGPU_OK;
}

}  // extern "C"

namespace {
// Both inverse entry points: delimiter blocks (d_orig_ptr null; chunks of block_size + 1 bytes) and
// origin-pointer blocks (chunks of block_size + 1 bytes as well: the caller passes the container's
// block size minus one, so that a chunk is one block's last column; `delimiter` is 256 then).
int inverse_launch(int device, const uint8_t* d_in, size_t total, size_t block_size, uint32_t delimiter,
                   const uint32_t* d_orig_ptr, uint8_t* d_out, uint32_t* d_out_lens, void* d_workspace,
                   size_t workspace_bytes, void* stream) {
    int rc = check_device(device);
    if (rc) return rc;
    if (block_size == 0) return fail(BWTGPU_ERR_INVALID, "block_size must be greater than 0");
    block_size = inverse_block_size(total, block_size);
    if (block_size > BWTGPU_MAX_BLOCK_SIZE)
        return fail(BWTGPU_ERR_BLOCK_TOO_LARGE, "block_size %zu exceeds the supported maximum %zu", block_size,
                    (size_t)BWTGPU_MAX_BLOCK_SIZE);
    if (total == 0) return BWTGPU_OK;
    if (!d_in || !d_out || !d_out_lens || !d_workspace) return fail(BWTGPU_ERR_INVALID, "null device pointer");
    const Device& d = *g_dev[device];
    const InvGeom g = inv_geom(d, total, block_size);
    if (workspace_bytes < g.bytes)
        return fail(BWTGPU_ERR_WORKSPACE, "workspace of %zu bytes given, %zu needed", workspace_bytes, g.bytes);
    if (g.nblocks > 0xFFFFFFF0ull) return fail(BWTGPU_ERR_INVALID, "too many blocks in one batch");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemsetAsync(d_workspace, 0, g.tiled ? g.dacc_off + g.nblocks * 8 : kHeaderBytes, s));
    InvArgs a;
    uint8_t* ws = (uint8_t*)d_workspace;
    a.in = d_in;
    a.out = d_out;
    a.total_len = total;
    a.block_size = (uint32_t)block_size;
    a.nblocks = (uint32_t)g.nblocks;
    a.delim = delimiter;
    a.orig_ptr = d_orig_ptr;
    a.out_stride = (uint32_t)block_size + (d_orig_ptr ? 1u : 0u);
    a.tstride = g.tstride;
    a.seg_cap = g.seg_cap;
    a.split_shift = g.split_shift;
    a.slot = g.slot;
    a.seg_wgs = g.seg_wgs;
    a.multi_min_rows = g.multi_min_rows;
    {
        const size_t unit = block_size + 1;
        splitter_map((uint32_t)std::min(total, unit), &a.split_mul[0], &a.split_inv[0]);
        splitter_map((uint32_t)(total - (g.nblocks - 1) * unit), &a.split_mul[1], &a.split_inv[1]);
    }
    a.T = ws + g.t_off;
    a.drow = (uint32_t*)(ws + g.drow_off);
    a.mode = (uint32_t*)(ws + g.mode_off);
    a.seg = (uint2*)(ws + g.seg_off);
    a.seg_rec = (uint2*)(ws + g.segrec_off);
    a.chain = (uint2*)(ws + g.chain_off);
    a.stage = ws + g.stage_off;
    a.stage_cap = (uint32_t)round_up(g.seg_cap, 64);
    a.seg_ovf = (uint32_t*)(ws + g.segovf_off);
    a.chunk_cap = g.chunk_cap;
    a.pool_used = (uint32_t*)(ws + g.used_off);
    a.chunk_next = (uint32_t*)(ws + g.next_off);
    a.pool = ws + g.pool_off;
    a.lens = d_out_lens;
    a.queue = (uint32_t*)d_workspace;
    if (block_size + 1 <= (size_t)kInvSmallCap && !d_orig_ptr) {  // small blocks: one wave per block, everything in LDS
        const size_t wgs = (g.nblocks + kInvSmallWaves - 1) / kInvSmallWaves;
        const uint32_t grid = (uint32_t)std::max<size_t>(1, std::min(wgs, (size_t)d.cus * 8));
        hipLaunchKernelGGL(bwt_inverse_small_kernel, dim3(grid), dim3(kInvSmallWaves * kWave), 0, s, a);
        HIP_TRY(hipGetLastError());
        return BWTGPU_OK;
    }
    // Blocks over at most four text symbols are inverted inside LDS first (bwt_inverse_packed.hpp); the
    // kernels below then skip the blocks that kernel has finished.
    a.packed_first = 0;
    a.left_hint = nullptr;
    bool sparse = false;  // the device's previous batch left nothing to the table kernels: launch them as small grids
    {
        if (g.pack_wgs && !d_orig_ptr) {
            a.packed_first = 1;
            const size_t hint = (((uintptr_t)s >> 4) ^ ((uintptr_t)s >> 12)) % (size_t)kInvHints;
            a.left_hint = d.inv_left_dev ? d.inv_left_dev + hint : nullptr;
            sparse = d.inv_left && __atomic_load_n(d.inv_left + hint, __ATOMIC_RELAXED) == 0u;
            hipLaunchKernelGGL(bwt_inverse_packed_classify_kernel,
                               dim3((uint32_t)((g.nblocks + kPackClassifyThreads / kWave - 1) / (kPackClassifyThreads / kWave))),
                               dim3(kPackClassifyThreads), 0, s, a);
            hipLaunchKernelGGL(bwt_inverse_packed_kernel, dim3(g.pack_wgs), dim3(kPackThreads), 0, s, a);
            // blocks of at most 64 KiB over any alphabet: their whole LF table fits LDS as 16-bit entries
            if (block_size + 1 <= (size_t)kT16MaxRows)
                hipLaunchKernelGGL(bwt_inverse_t16_kernel, dim3(g.pack_wgs), dim3(kT16Threads), 0, s, a);
            HIP_TRY(hipGetLastError());
        }
    }
    if (g.tiled) {
        InvTileArgs ta;
        ta.tile_rows = g.tile_rows;
        ta.tiles_per_block = g.tiles_per_block;
        ta.tile_hist = (uint32_t*)(ws + g.thist_off);
        ta.dacc = (uint32_t*)(ws + g.dacc_off);
        const uint32_t tiles = (uint32_t)(g.nblocks * g.tiles_per_block);
        hipLaunchKernelGGL(bwt_inverse_tile_hist_kernel<kInvThreads>, dim3(tiles), dim3(kInvThreads), 0, s, a, ta);
        if (g.wide)
            hipLaunchKernelGGL((bwt_inverse_tile_build_kernel<kInvThreads, uint64_t>), dim3(tiles), dim3(kInvThreads), 0, s, a, ta);
        else
            hipLaunchKernelGGL((bwt_inverse_tile_build_kernel<kInvThreads, uint32_t>), dim3(tiles), dim3(kInvThreads), 0, s, a, ta);
    } else {
        hipLaunchKernelGGL(bwt_inverse_build_kernel<kInvThreads>, dim3(g.build_grid), dim3(kInvThreads), 0, s, a);
    }
    HIP_TRY(hipGetLastError());
    const uint32_t per_block_grid = (uint32_t)((g.nblocks + kWave - 1) / kWave);
    if (g.wide) hipLaunchKernelGGL(bwt_inverse_walk_kernel<uint64_t>, dim3(per_block_grid), dim3(kWave), 0, s, a);
    else hipLaunchKernelGGL(bwt_inverse_walk_kernel<uint32_t>, dim3(per_block_grid), dim3(kWave), 0, s, a);
    HIP_TRY(hipGetLastError());
    if (g.split) {
        const uint64_t wgs = (uint64_t)((g.nblocks + 7) / 8 * 8) * g.seg_wgs;
        if (wgs > 0x7FFFFFFFull) return fail(BWTGPU_ERR_INVALID, "too many segments in one batch");
        const uint64_t walk_wgs = (uint64_t)((g.nblocks + 7) / 8 * 8) * ((g.seg_cap + kWalkThreads - 1) / kWalkThreads);
        if (walk_wgs > 0x7FFFFFFFull) return fail(BWTGPU_ERR_INVALID, "too many segments in one batch");
        const uint32_t walk_grid = sparse ? (uint32_t)std::min<uint64_t>(walk_wgs, kSparseGrid) : (uint32_t)walk_wgs;
        if (g.wide) hipLaunchKernelGGL((bwt_inverse_walk_segments_kernel<uint64_t, false>), dim3((uint32_t)walk_wgs), dim3(kWalkThreads), 0, s, a, (uint32_t)walk_wgs);
        else if (sparse) hipLaunchKernelGGL((bwt_inverse_walk_segments_kernel<uint32_t, true>), dim3(walk_grid), dim3(kWalkThreads), 0, s, a, (uint32_t)walk_wgs);
        else hipLaunchKernelGGL((bwt_inverse_walk_segments_kernel<uint32_t, false>), dim3(walk_grid), dim3(kWalkThreads), 0, s, a, (uint32_t)walk_wgs);
        HIP_TRY(hipGetLastError());
        // many segments per block, or few blocks (the chain following is then the whole cost): the
        // records go through LDS
        if (g.seg_cap <= kStitchLdsSegs && (g.split_shift == kMinSplitShift || g.nblocks <= (size_t)d.cus))
            hipLaunchKernelGGL(bwt_inverse_stitch_lds_kernel<1024>, dim3((uint32_t)std::min<size_t>(g.nblocks, (size_t)d.cus)),
                               dim3(1024), 0, s, a);
        else if (g.seg_cap > kStitchSmallSegs)  // more super nodes than the 256-thread kernel's tables hold
            hipLaunchKernelGGL(bwt_inverse_stitch_kernel<1024>, dim3((uint32_t)g.nblocks), dim3(1024), 0, s, a);
        else
            hipLaunchKernelGGL(bwt_inverse_stitch_kernel<256>, dim3((uint32_t)g.nblocks), dim3(256), 0, s, a);
        HIP_TRY(hipGetLastError());
        if (sparse) hipLaunchKernelGGL(bwt_inverse_place_kernel<true>, dim3((uint32_t)std::min<uint64_t>(wgs, kSparseGrid)), dim3(kSegThreads), 0, s, a, (uint32_t)wgs);
        else hipLaunchKernelGGL(bwt_inverse_place_kernel<false>, dim3((uint32_t)wgs), dim3(kSegThreads), 0, s, a, (uint32_t)wgs);
        HIP_TRY(hipGetLastError());
    }
    if (d_orig_ptr) {
        hipLaunchKernelGGL(bwt_inverse_origin_finish_kernel, dim3((uint32_t)g.nblocks), dim3(256), 0, s, a);
        HIP_TRY(hipGetLastError());
    }
    if (a.packed_first && a.left_hint) {
        hipLaunchKernelGGL(bwt_inverse_left_kernel, dim3(1), dim3(1), 0, s, a);
        HIP_TRY(hipGetLastError());
    }
    return BWTGPU_OK;
}
}  // namespace

extern "C" {

int bwtgpu_inverse_batch_device(int device, const uint8_t* d_in, size_t total, size_t block_size,
                                uint8_t delimiter, uint8_t* d_out, uint32_t* d_out_lens, void* d_workspace,
                                size_t workspace_bytes, void* stream) {
    return inverse_launch(device, d_in, total, block_size, delimiter, nullptr, d_out, d_out_lens, d_workspace,
                          workspace_bytes, stream);
}

size_t bwtgpu_inverse_origin_workspace_bytes(int device, size_t total_len, size_t block_size) {
    if (check_device(device) || block_size < 2 || total_len == 0) return 0;
    return inv_geom(*g_dev[device], total_len, std::min(block_size, total_len) - 1).bytes;
}

int bwtgpu_inverse_origin_batch_device(int device, const uint8_t* d_in, size_t total_len, size_t block_size,
                                       const uint32_t* d_orig_ptr, uint8_t* d_out, uint32_t* d_out_lens,
                                       void* d_workspace, size_t workspace_bytes, void* stream) {
    if (block_size == 0) return fail(BWTGPU_ERR_INVALID, "block_size must be greater than 0");
    if (total_len == 0) return BWTGPU_OK;
    if (!d_orig_ptr) return fail(BWTGPU_ERR_INVALID, "null origin-pointer array");
    block_size = std::min(block_size, total_len);
    if (block_size == 1) {  // one-byte blocks are their own transform (and have no room for a table row)
        int rc = check_device(device);
        if (rc) return rc;
        if (!d_in || !d_out || !d_out_lens) return fail(BWTGPU_ERR_INVALID, "null device pointer");
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipMemcpyAsync(d_out, d_in, total_len, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)d_out_lens, 1, total_len, (hipStream_t)stream));
        return BWTGPU_OK;
    }
    return inverse_launch(device, d_in, total_len, block_size - 1, 256u, d_orig_ptr, d_out, d_out_lens, d_workspace,
                          workspace_bytes, stream);
}

// ---- forward of origin-pointer blocks: rotations sorted by the device-wide path ------------------
size_t bwtgpu_forward_origin_workspace_bytes(int device, size_t total_len, size_t block_size) {
    if (check_device(device) != BWTGPU_OK || block_size == 0 || total_len == 0 || total_len > kGridMaxRows) return 0;
    return grid_geom(total_len, std::min(block_size, total_len), true).bytes;
}

int bwtgpu_forward_origin_batch_device(int device, const uint8_t* d_in, size_t total_len, size_t block_size,
                                       uint8_t* d_out, uint32_t* d_rows, void* d_workspace, size_t workspace_bytes,
                                       void* stream) {
    int rc = check_device(device);
    if (rc) return rc;
    if (block_size == 0) return fail(BWTGPU_ERR_INVALID, "block_size must be greater than 0");
    if (total_len == 0) return BWTGPU_OK;
    block_size = std::min(block_size, total_len);
    if (block_size > BWTGPU_MAX_BLOCK_SIZE)
        return fail(BWTGPU_ERR_BLOCK_TOO_LARGE, "block_size %zu exceeds the supported maximum %zu", block_size,
                    (size_t)BWTGPU_MAX_BLOCK_SIZE);
    if (total_len > kGridMaxRows) return fail(BWTGPU_ERR_INVALID, "at most 2^31 bytes per batch (%zu given): split it", total_len);
    if (!d_in || !d_out || !d_rows || !d_workspace) return fail(BWTGPU_ERR_INVALID, "null device pointer");
    const size_t need = grid_geom(total_len, block_size, true).bytes;
    if (workspace_bytes < need)
        return fail(BWTGPU_ERR_WORKSPACE, "workspace of %zu bytes given, %zu needed", workspace_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemsetAsync(d_workspace, 0, kHeaderBytes, s));
    return forward_grid(*g_dev[device], d_in, total_len, block_size, 0, d_out, d_workspace, s, d_rows);
}

int bwtgpu_forward_origin_batch(int device, const uint8_t* in, size_t total_len, size_t block_size, uint8_t* out,
                                uint32_t* rows) {
    int rc = check_device(device);
    if (rc) return rc;
    if (block_size == 0) return fail(BWTGPU_ERR_INVALID, "block_size must be greater than 0");
    if (total_len == 0) return BWTGPU_OK;
    if (!in || !out || !rows) return fail(BWTGPU_ERR_INVALID, "null host pointer");
    block_size = std::min(block_size, total_len);
    if (block_size > BWTGPU_MAX_BLOCK_SIZE) return fail(BWTGPU_ERR_BLOCK_TOO_LARGE, "block_size %zu too large", block_size);
    Device& d = *g_dev[device];
    std::lock_guard<std::mutex> lock(d.mu);
    HIP_TRY(hipSetDevice(device));
    // sub-batches of whole blocks bound the workspace (28 bytes per input byte); blocks are independent
    const size_t max_sub = std::max<size_t>(block_size, (size_t)256 << 20);
    const size_t blocks_per_sub = std::max<size_t>(1, max_sub / block_size);
    for (size_t off = 0; off < total_len;) {
        const size_t sub = std::min(total_len - off, blocks_per_sub * block_size);
        const size_t nb = (sub + block_size - 1) / block_size, b0 = off / block_size;
        if ((rc = grow(&d.d_in, &d.in_cap, sub))) return rc;
        if ((rc = grow(&d.d_out, &d.out_cap, sub))) return rc;
        if ((rc = grow(&d.d_ws, &d.ws_cap, grid_geom(sub, std::min(block_size, sub), true).bytes))) return rc;
        if ((rc = grow(&d.d_aux, &d.aux_cap, nb * 4))) return rc;
        HIP_TRY(hipMemcpyAsync(d.d_in, in + off, sub, hipMemcpyHostToDevice, d.stream));
        rc = bwtgpu_forward_origin_batch_device(device, (const uint8_t*)d.d_in, sub, block_size, (uint8_t*)d.d_out,
                                                (uint32_t*)d.d_aux, d.d_ws, d.ws_cap, d.stream);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(rows + b0, d.d_aux, nb * 4, hipMemcpyDeviceToHost, d.stream));
        HIP_TRY(hipMemcpyAsync(out + off, d.d_out, sub, hipMemcpyDeviceToHost, d.stream));
        HIP_TRY(hipStreamSynchronize(d.stream));
        off += sub;
    }
    return BWTGPU_OK;
}

int bwtgpu_forward_batch(int device, const uint8_t* in, size_t total_len, size_t block_size, uint8_t delimiter,
                         uint8_t* out) {
    int rc = check_device(device);
    if (rc) return rc;
    if (block_size == 0) return fail(BWTGPU_ERR_INVALID, "block_size must be greater than 0");
    block_size = forward_block_size(total_len, block_size);
    if (block_size > BWTGPU_MAX_BLOCK_SIZE)
        return fail(BWTGPU_ERR_BLOCK_TOO_LARGE, "block_size %zu exceeds the supported maximum %zu", block_size,
                    (size_t)BWTGPU_MAX_BLOCK_SIZE);
    if (total_len == 0) return BWTGPU_OK;
    if (!in || !out) return fail(BWTGPU_ERR_INVALID, "null host pointer");
    Device& d = *g_dev[device];
    std::lock_guard<std::mutex> lock(d.mu);
    HIP_TRY(hipSetDevice(device));
    // sub-batches bound the device footprint; blocks are independent so any split is exact
    const size_t max_sub = std::max<size_t>(block_size, (size_t)1 << 30);
    const size_t blocks_per_sub = std::max<size_t>(1, max_sub / block_size);
    for (size_t off = 0; off < total_len;) {
        const size_t sub = std::min(total_len - off, blocks_per_sub * block_size);
        const size_t nb = (sub + block_size - 1) / block_size;
        const FwdGeom g = fwd_geom(d, sub, block_size);
        if ((rc = grow(&d.d_in, &d.in_cap, sub))) return rc;
        if ((rc = grow(&d.d_out, &d.out_cap, sub + nb))) return rc;
        if ((rc = grow(&d.d_ws, &d.ws_cap, g.bytes))) return rc;
        HIP_TRY(hipMemcpyAsync(d.d_in, in + off, sub, hipMemcpyHostToDevice, d.stream));
        rc = forward_device(device, (const uint8_t*)d.d_in, sub, block_size, delimiter, (uint8_t*)d.d_out, d.d_ws, d.ws_cap,
                            d.stream, true);  // (this call waits for the result anyway)
        if (rc) return rc;
        const size_t out_off = (off / block_size) * (block_size + 1);
        HIP_TRY(hipMemcpyAsync(out + out_off, d.d_out, sub + nb, hipMemcpyDeviceToHost, d.stream));
        uint32_t status[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(status, d.d_ws, sizeof status, hipMemcpyDeviceToHost, d.stream));
        HIP_TRY(hipStreamSynchronize(d.stream));
        if (status[1]) return fail(BWTGPU_ERR_INTERNAL, "forward kernel reported status 0x%x", status[1]);
        off += sub;
    }
    return BWTGPU_OK;
}

int bwtgpu_inverse_batch(int device, const uint8_t* in, size_t total, size_t block_size, uint8_t delimiter,
                         uint8_t* out, size_t* out_len) {
    int rc = check_device(device);
    if (rc) return rc;
    if (block_size == 0) return fail(BWTGPU_ERR_INVALID, "block_size must be greater than 0");
    block_size = inverse_block_size(total, block_size);
    if (block_size > BWTGPU_MAX_BLOCK_SIZE)
        return fail(BWTGPU_ERR_BLOCK_TOO_LARGE, "block_size %zu exceeds the supported maximum %zu", block_size,
                    (size_t)BWTGPU_MAX_BLOCK_SIZE);
    if (out_len) *out_len = 0;
    if (total == 0) return BWTGPU_OK;
    if (!in || !out) return fail(BWTGPU_ERR_INVALID, "null host pointer");
    Device& d = *g_dev[device];
    std::lock_guard<std::mutex> lock(d.mu);
    HIP_TRY(hipSetDevice(device));
    const size_t chunk = block_size + 1;
    const size_t max_sub = std::max<size_t>(chunk, (size_t)1 << 30);
    const size_t blocks_per_sub = std::max<size_t>(1, max_sub / chunk);
    size_t written = 0;
    std::vector<uint32_t> lens;
    for (size_t off = 0; off < total;) {
        const size_t sub = std::min(total - off, blocks_per_sub * chunk);
        const InvGeom g = inv_geom(d, sub, block_size);
        if ((rc = grow(&d.d_in, &d.in_cap, sub))) return rc;
        if ((rc = grow(&d.d_out, &d.out_cap, sub))) return rc;
        if ((rc = grow(&d.d_ws, &d.ws_cap, g.bytes))) return rc;
        if ((rc = grow(&d.d_aux, &d.aux_cap, g.nblocks * 4))) return rc;
        HIP_TRY(hipMemcpyAsync(d.d_in, in + off, sub, hipMemcpyHostToDevice, d.stream));
        rc = bwtgpu_inverse_batch_device(device, (const uint8_t*)d.d_in, sub, block_size, delimiter,
                                         (uint8_t*)d.d_out, (uint32_t*)d.d_aux, d.d_ws, d.ws_cap, d.stream);
        if (rc) return rc;
        lens.resize(g.nblocks);
        HIP_TRY(hipMemcpyAsync(lens.data(), d.d_aux, g.nblocks * 4, hipMemcpyDeviceToHost, d.stream));
        HIP_TRY(hipStreamSynchronize(d.stream));
        // a block made by the forward transform yields len_b bytes; compact only if one did not
        bool exact = true;
        size_t text = 0;
        for (size_t b = 0; b < g.nblocks; ++b) {
            const size_t nb = std::min(chunk, sub - b * chunk);
            if (lens[b] == BWTGPU_LEN_NO_DELIMITER)
                return fail(BWTGPU_ERR_NO_DELIMITER, "block %zu holds no delimiter byte 0x%02x",
                            off / chunk + b, delimiter);
            if (lens[b] != nb - 1) exact = false;
            text += lens[b];
        }
        if (exact) {
            HIP_TRY(hipMemcpy(out + written, d.d_out, text, hipMemcpyDeviceToHost));
        } else {
            size_t w = written;
            for (size_t b = 0; b < g.nblocks; ++b) {
                const size_t nb = std::min(chunk, sub - b * chunk);
                if (lens[b])
                    HIP_TRY(hipMemcpy(out + w, (uint8_t*)d.d_out + b * block_size + (nb - 1 - lens[b]), lens[b],
                                      hipMemcpyDeviceToHost));
                w += lens[b];
            }
        }
        written += text;
        off += sub;
    }
    if (out_len) *out_len = written;
    return BWTGPU_OK;
}

int bwtgpu_inverse_origin_batch(int device, const uint8_t* in, size_t total_len, size_t block_size,
                                const uint32_t* orig_ptr, uint8_t* out) {
    int rc = check_device(device);
    if (rc) return rc;
    if (block_size == 0) return fail(BWTGPU_ERR_INVALID, "block_size must be greater than 0");
    if (total_len == 0) return BWTGPU_OK;
    if (!in || !out || !orig_ptr) return fail(BWTGPU_ERR_INVALID, "null host pointer");
    block_size = std::min(block_size, total_len);
    if (block_size > BWTGPU_MAX_BLOCK_SIZE) return fail(BWTGPU_ERR_BLOCK_TOO_LARGE, "block_size %zu too large", block_size);
    Device& d = *g_dev[device];
    std::lock_guard<std::mutex> lock(d.mu);
    HIP_TRY(hipSetDevice(device));
    const size_t max_sub = std::max<size_t>(block_size, (size_t)1 << 30);
    const size_t blocks_per_sub = std::max<size_t>(1, max_sub / block_size);
    std::vector<uint32_t> lens;
    for (size_t off = 0; off < total_len;) {
        const size_t sub = std::min(total_len - off, blocks_per_sub * block_size);
        const size_t nb = (sub + block_size - 1) / block_size, b0 = off / block_size;
        const size_t ws = block_size > 1 ? inv_geom(d, sub, std::min(block_size, sub) - 1).bytes : 256;
        if ((rc = grow(&d.d_in, &d.in_cap, sub))) return rc;
        if ((rc = grow(&d.d_out, &d.out_cap, sub))) return rc;
        if ((rc = grow(&d.d_ws, &d.ws_cap, ws))) return rc;
        if ((rc = grow(&d.d_aux, &d.aux_cap, nb * 8))) return rc;
        uint32_t* d_ptr = (uint32_t*)d.d_aux;
        uint32_t* d_lens = d_ptr + nb;
        HIP_TRY(hipMemcpyAsync(d.d_in, in + off, sub, hipMemcpyHostToDevice, d.stream));
        HIP_TRY(hipMemcpyAsync(d_ptr, orig_ptr + b0, nb * 4, hipMemcpyHostToDevice, d.stream));
        rc = bwtgpu_inverse_origin_batch_device(device, (const uint8_t*)d.d_in, sub, block_size, d_ptr, (uint8_t*)d.d_out,
                                                d_lens, d.d_ws, d.ws_cap, d.stream);
        if (rc) return rc;
        lens.resize(nb);
        HIP_TRY(hipMemcpyAsync(lens.data(), d_lens, nb * 4, hipMemcpyDeviceToHost, d.stream));
        HIP_TRY(hipMemcpyAsync(out + off, d.d_out, sub, hipMemcpyDeviceToHost, d.stream));
        HIP_TRY(hipStreamSynchronize(d.stream));
        for (size_t b = 0; b < nb; ++b)
            if (lens[b] == BWTGPU_LEN_NO_DELIMITER)
                return fail(BWTGPU_ERR_INVALID, "block %zu: origin pointer %u outside the block", b0 + b, orig_ptr[b0 + b]);
        off += sub;
    }
    return BWTGPU_OK;
}

int bwtgpu_time_resident(int device, const uint8_t* in, size_t total_len, size_t block_size, uint8_t delimiter,
                         int reps, double* forward_ms, double* inverse_ms) {
    int rc = check_device(device);
    if (rc) return rc;
    if (block_size == 0 || reps < 1) return fail(BWTGPU_ERR_INVALID, "block_size and reps must be greater than 0");
    if (!in || !forward_ms || !inverse_ms || total_len == 0) return fail(BWTGPU_ERR_INVALID, "null pointer or empty input");
    if (total_len > ((size_t)1 << 30)) return fail(BWTGPU_ERR_INVALID, "at most 1 GiB per timing call");
    block_size = forward_block_size(total_len, block_size);
    if (block_size > BWTGPU_MAX_BLOCK_SIZE) return fail(BWTGPU_ERR_BLOCK_TOO_LARGE, "block_size %zu too large", block_size);
    Device& d = *g_dev[device];
    std::lock_guard<std::mutex> lock(d.mu);
    HIP_TRY(hipSetDevice(device));
    const size_t nb = (total_len + block_size - 1) / block_size, fwd_bytes = total_len + nb;
    const FwdGeom fg = fwd_geom(d, total_len, block_size);
    const InvGeom ig = inv_geom(d, fwd_bytes, block_size);
    if ((rc = grow(&d.d_in, &d.in_cap, total_len))) return rc;
    if ((rc = grow(&d.d_out, &d.out_cap, fwd_bytes))) return rc;
    if ((rc = grow(&d.d_ws, &d.ws_cap, std::max(fg.bytes, ig.bytes)))) return rc;
    if ((rc = grow(&d.d_aux, &d.aux_cap, nb * 4))) return rc;
    uint8_t* d_back = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    auto cleanup = [&]() {
        for (hipEvent_t e : ev)
            if (e) (void)hipEventDestroy(e);
        if (d_back) (void)hipFree(d_back);
    };
    if (hipMalloc((void**)&d_back, total_len) != hipSuccess) return fail(BWTGPU_ERR_HIP, "hipMalloc of %zu bytes failed", total_len);
    for (hipEvent_t& e : ev)
        if (hipEventCreate(&e) != hipSuccess) {
            cleanup();
            return fail(BWTGPU_ERR_HIP, "hipEventCreate failed");
        }
    auto guarded = [&](int code) {
        if (code) cleanup();
        return code;
    };
    auto hip_ok = [&](hipError_t e) { return e == hipSuccess ? 0 : fail(BWTGPU_ERR_HIP, "%s", hipGetErrorString(e)); };
    if ((rc = guarded(hip_ok(hipMemcpyAsync(d.d_in, in, total_len, hipMemcpyHostToDevice, d.stream))))) return rc;
    float fwd = 0.f, inv = 0.f;
    for (int dir = 0; dir < 2; ++dir) {
        for (int it = 0; it <= reps; ++it) {  // pass 0 is the warm-up
            if (it == 1 && (rc = guarded(hip_ok(hipEventRecord(ev[0], d.stream))))) return rc;
            rc = dir == 0 ? forward_device(device, (const uint8_t*)d.d_in, total_len, block_size, delimiter,
                                           (uint8_t*)d.d_out, d.d_ws, d.ws_cap, d.stream, true)  // (what a caller that waits sees)
                          : bwtgpu_inverse_batch_device(device, (const uint8_t*)d.d_out, fwd_bytes, block_size, delimiter,
                                                        d_back, (uint32_t*)d.d_aux, d.d_ws, d.ws_cap, d.stream);
            if (guarded(rc)) return rc;
        }
        if ((rc = guarded(hip_ok(hipEventRecord(ev[1], d.stream))))) return rc;
        if ((rc = guarded(hip_ok(hipEventSynchronize(ev[1]))))) return rc;
        if ((rc = guarded(hip_ok(hipEventElapsedTime(dir == 0 ? &fwd : &inv, ev[0], ev[1]))))) return rc;
    }
    cleanup();
    *forward_ms = (double)fwd / reps;
    *inverse_ms = (double)inv / reps;
    return BWTGPU_OK;
}

int bwtgpu_lowest_clear_bit(const uint32_t presence[8]) {
    for (int v = 0; v < 256; ++v)
        if (!((presence[v >> 5] >> (v & 31)) & 1u)) return v;
    return fail(BWTGPU_ERR_NO_FREE_BYTE, "all 256 byte values occur: no delimiter can be chosen");
}

int bwtgpu_presence_device(int device, const uint8_t* d_data, size_t len, uint32_t* d_presence, void* stream) {
    int rc = check_device(device);
    if (rc) return rc;
    if (len == 0) return BWTGPU_OK;
    if (!d_data || !d_presence) return fail(BWTGPU_ERR_INVALID, "null device pointer");
    HIP_TRY(hipSetDevice(device));
    const Device& d = *g_dev[device];
    hipLaunchKernelGGL(presence_kernel, dim3(stream_grid(d, len, 256 * 64)), dim3(256), 0, (hipStream_t)stream,
                       d_data, len, d_presence);
    HIP_TRY(hipGetLastError());
    return BWTGPU_OK;
}

int bwtgpu_find_unused_byte(const uint8_t* data, size_t len, uint32_t presence_out[8]) {
    int rc = check_device(0);
    if (rc) return rc;
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (len) {
        if (!data) return fail(BWTGPU_ERR_INVALID, "null host pointer");
        Device& d = *g_dev[0];
        std::lock_guard<std::mutex> lock(d.mu);
        HIP_TRY(hipSetDevice(0));
        if ((rc = grow(&d.d_aux, &d.aux_cap, 64))) return rc;
        HIP_TRY(hipMemsetAsync(d.d_aux, 0, 32, d.stream));
        const size_t step = (size_t)1 << 30;
        for (size_t off = 0; off < len; off += step) {
            const size_t sub = std::min(step, len - off);
            if ((rc = grow(&d.d_in, &d.in_cap, sub))) return rc;
            HIP_TRY(hipMemcpyAsync(d.d_in, data + off, sub, hipMemcpyHostToDevice, d.stream));
            if ((rc = bwtgpu_presence_device(0, (const uint8_t*)d.d_in, sub, (uint32_t*)d.d_aux, d.stream))) return rc;
        }
        HIP_TRY(hipMemcpyAsync(mask, d.d_aux, 32, hipMemcpyDeviceToHost, d.stream));
        HIP_TRY(hipStreamSynchronize(d.stream));
    }
    if (presence_out) memcpy(presence_out, mask, sizeof mask);
    return bwtgpu_lowest_clear_bit(mask);
}

int bwtgpu_synth_device(int device, int kind, uint64_t seed, uint64_t offset, uint8_t* d_out, size_t len,
                        const uint8_t* d_text, size_t text_len, void* stream) {
    int rc = check_device(device);
    if (rc) return rc;
    if (kind < 0 || kind > 6) return fail(BWTGPU_ERR_INVALID, "unknown synthetic kind %d", kind);
    if (kind >= 5 && (!d_text || text_len == 0)) return fail(BWTGPU_ERR_INVALID, "kinds 5 and 6 need a text sample");
    if (len == 0) return BWTGPU_OK;
    if (!d_out) return fail(BWTGPU_ERR_INVALID, "null device pointer");
    HIP_TRY(hipSetDevice(device));
    const Device& d = *g_dev[device];
    hipLaunchKernelGGL(synth_kernel, dim3(stream_grid(d, len, 256 * 16)), dim3(256), 0, (hipStream_t)stream, kind,
                       seed, offset, d_out, len, d_text, text_len ? text_len : 1);
    HIP_TRY(hipGetLastError());
    return BWTGPU_OK;
}

bwtgpu_pipeline* bwtgpu_pipeline_create(int device, int direction, size_t block_size, size_t slab_blocks,
                                        int depth) {
    if (check_device(device)) return nullptr;
    if (block_size == 0 || slab_blocks == 0 || depth < 1 || (direction != BWTGPU_FORWARD && direction != BWTGPU_INVERSE)) {
        fail(BWTGPU_ERR_INVALID, "bad pipeline shape (block_size %zu, slab_blocks %zu, depth %d)", block_size,
             slab_blocks, depth);
        return nullptr;
    }
    if (block_size > BWTGPU_MAX_BLOCK_SIZE) {
        fail(BWTGPU_ERR_BLOCK_TOO_LARGE, "block_size %zu exceeds the supported maximum %zu", block_size,
             (size_t)BWTGPU_MAX_BLOCK_SIZE);
        return nullptr;
    }
    bwtgpu_pipeline* p = new bwtgpu_pipeline;
    p->device = device;
    p->direction = direction;
    p->block_size = block_size;
    p->slab_blocks = slab_blocks;
    const Device& d = *g_dev[device];
    if (direction == BWTGPU_FORWARD) {
        p->in_cap = slab_blocks * block_size;
        p->out_cap = slab_blocks * (block_size + 1);
        p->ws_bytes = fwd_geom(d, p->in_cap, block_size).bytes;
    } else {
        p->in_cap = slab_blocks * (block_size + 1);
        p->out_cap = slab_blocks * block_size;
        p->ws_bytes = inv_slab_workspace(d, slab_blocks, block_size);
    }
    if (pipeline_alloc(p, depth) != BWTGPU_OK) {
        pipeline_free(p);
        return nullptr;
    }
    return p;
}

uint8_t* bwtgpu_pipeline_input(bwtgpu_pipeline* p, int slot) {
    if (!p || slot < 0 || slot >= (int)p->slots.size()) return nullptr;
    return p->slots[(size_t)slot].h_in;
}

size_t bwtgpu_pipeline_input_capacity(const bwtgpu_pipeline* p) { return p ? p->in_cap : 0; }

namespace {
// The transform of `bytes` bytes into slot s's output: from d_src if given (already on the device), else from
// the slot's own input buffer, uploaded first unless bwtgpu_pipeline_upload_presence has done that.
int pipeline_run(bwtgpu_pipeline* p, int slot, const uint8_t* d_src, size_t bytes, uint8_t delimiter, bool may_block = false) {
    PipeSlot& s = p->slots[(size_t)slot];
    if (s.busy) return fail(BWTGPU_ERR_INVALID, "slot %d submitted again before it was collected", slot);
    if (bytes > p->in_cap) return fail(BWTGPU_ERR_INVALID, "%zu bytes exceed the slab capacity %zu", bytes, p->in_cap);
    HIP_TRY(hipSetDevice(p->device));
    s.in_bytes = bytes;
    p->delimiter = delimiter;
    const bool uploaded = d_src || (bytes && s.resident == bytes);
    if (!d_src) d_src = s.d_in;
    s.resident = 0;
    if (bytes) {
        // The slot's workspace was sized for every shape a slab can take; should a batch still ask
        // for more, the slot gets a larger one rather than failing.
        const size_t need = p->direction == BWTGPU_FORWARD ? bwtgpu_forward_workspace_bytes(p->device, bytes, p->block_size)
                                                           : bwtgpu_inverse_workspace_bytes(p->device, bytes, p->block_size);
        if (need > s.ws_cap) {
            HIP_TRY(hipStreamSynchronize(s.stream));
            HIP_TRY(hipFree(s.d_ws));
            s.d_ws = nullptr;
            s.ws_cap = 0;
            HIP_TRY(hipMalloc(&s.d_ws, need));
            s.ws_cap = need;
        }
        if (!uploaded) HIP_TRY(hipMemcpyAsync(s.d_in, s.h_in, bytes, hipMemcpyHostToDevice, s.stream));
        int rc;
        if (p->direction == BWTGPU_FORWARD) {
            rc = forward_device(p->device, d_src, bytes, p->block_size, delimiter, s.d_out, s.d_ws, s.ws_cap, s.stream, may_block);
            if (rc) return rc;
            const size_t nb = (bytes + p->block_size - 1) / p->block_size;
            HIP_TRY(hipMemcpyAsync(s.h_out, s.d_out, bytes + nb, hipMemcpyDeviceToHost, s.stream));
            HIP_TRY(hipMemcpyAsync(s.h_lens, s.d_ws, 8, hipMemcpyDeviceToHost, s.stream));
        } else {
            rc = bwtgpu_inverse_batch_device(p->device, d_src, bytes, p->block_size, delimiter, s.d_out, s.d_lens,
                                             s.d_ws, s.ws_cap, s.stream);
            if (rc) return rc;
            const size_t nb = (bytes + p->block_size) / (p->block_size + 1);
            if (bytes > nb) HIP_TRY(hipMemcpyAsync(s.h_out, s.d_out, bytes - nb, hipMemcpyDeviceToHost, s.stream));
            HIP_TRY(hipMemcpyAsync(s.h_lens, s.d_lens, nb * 4, hipMemcpyDeviceToHost, s.stream));
        }
    }
    HIP_TRY(hipEventRecord(s.done, s.stream));
    s.busy = true;
    return BWTGPU_OK;
}
}  // namespace

int bwtgpu_pipeline_submit(bwtgpu_pipeline* p, int slot, size_t bytes, uint8_t delimiter) {
    if (!p || slot < 0 || slot >= (int)p->slots.size()) return fail(BWTGPU_ERR_INVALID, "bad pipeline slot");
    return pipeline_run(p, slot, nullptr, bytes, delimiter);
}

int bwtgpu_pipeline_run(bwtgpu_pipeline* p, int slot, size_t bytes, uint8_t delimiter, const uint8_t** out, size_t* out_bytes) {
    if (!p || slot < 0 || slot >= (int)p->slots.size()) return fail(BWTGPU_ERR_INVALID, "bad pipeline slot");
    const int rc = pipeline_run(p, slot, nullptr, bytes, delimiter, true);
    return rc ? rc : bwtgpu_pipeline_collect(p, slot, out, out_bytes);
}

int bwtgpu_pipeline_reserve(bwtgpu_pipeline* p, size_t nslabs) {
    if (!p || p->direction != BWTGPU_FORWARD || nslabs == 0) return fail(BWTGPU_ERR_INVALID, "resident slabs: a forward pipeline and at least one slab");
    HIP_TRY(hipSetDevice(p->device));
    for (PipeSlot& s : p->slots) HIP_TRY(hipStreamSynchronize(s.stream));  // nothing may still read the old slabs
    if (nslabs > p->resident_slabs) {
        if (p->d_resident) HIP_TRY(hipFree(p->d_resident));
        p->d_resident = nullptr;
        p->resident_slabs = 0;
        if (hipMalloc((void**)&p->d_resident, nslabs * p->in_cap) != hipSuccess) {
            (void)hipGetLastError();
            return fail(BWTGPU_ERR_HIP, "no device memory for %zu resident slabs of %zu bytes", nslabs, p->in_cap);
        }
        p->resident_slabs = nslabs;
    }
    if (!p->d_presence) HIP_TRY(hipMalloc((void**)&p->d_presence, 32));
    HIP_TRY(hipMemsetAsync(p->d_presence, 0, 32, p->slots[0].stream));
    HIP_TRY(hipStreamSynchronize(p->slots[0].stream));
    p->resident_bytes.assign(p->resident_slabs, 0);
    for (PipeSlot& s : p->slots) s.staging = false;
    return BWTGPU_OK;
}

int bwtgpu_pipeline_stage(bwtgpu_pipeline* p, int slot, size_t k, size_t bytes) {
    if (!p || slot < 0 || slot >= (int)p->slots.size()) return fail(BWTGPU_ERR_INVALID, "bad pipeline slot");
    if (k >= p->resident_slabs) return fail(BWTGPU_ERR_INVALID, "resident slab %zu of %zu reserved", k, p->resident_slabs);
    PipeSlot& s = p->slots[(size_t)slot];
    if (s.busy) return fail(BWTGPU_ERR_INVALID, "slot %d is in flight", slot);
    if (bytes > p->in_cap) return fail(BWTGPU_ERR_INVALID, "%zu bytes exceed the slab capacity %zu", bytes, p->in_cap);
    HIP_TRY(hipSetDevice(p->device));
    p->resident_bytes[k] = bytes;
    if (bytes) {
        uint8_t* dst = p->d_resident + k * p->in_cap;
        HIP_TRY(hipMemcpyAsync(dst, s.h_in, bytes, hipMemcpyHostToDevice, s.stream));
        HIP_TRY(hipEventRecord(s.staged, s.stream));
        s.staging = true;
        const int rc = bwtgpu_presence_device(p->device, dst, bytes, p->d_presence, s.stream);  // (ORs into the mask)
        if (rc) return rc;
    }
    return BWTGPU_OK;
}

int bwtgpu_pipeline_stage_wait(bwtgpu_pipeline* p, int slot) {
    if (!p || slot < 0 || slot >= (int)p->slots.size()) return fail(BWTGPU_ERR_INVALID, "bad pipeline slot");
    PipeSlot& s = p->slots[(size_t)slot];
    if (!s.staging) return BWTGPU_OK;
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipEventSynchronize(s.staged));
    s.staging = false;
    return BWTGPU_OK;
}

int bwtgpu_pipeline_presence(bwtgpu_pipeline* p, uint32_t presence_out[8]) {
    if (!p || !presence_out || !p->d_presence) return fail(BWTGPU_ERR_INVALID, "no resident slabs reserved");
    HIP_TRY(hipSetDevice(p->device));
    for (PipeSlot& s : p->slots) {
        HIP_TRY(hipStreamSynchronize(s.stream));
        s.staging = false;
    }
    HIP_TRY(hipMemcpy(presence_out, p->d_presence, 32, hipMemcpyDeviceToHost));
    return BWTGPU_OK;
}

int bwtgpu_pipeline_submit_resident(bwtgpu_pipeline* p, int slot, size_t k, uint8_t delimiter) {
    if (!p || slot < 0 || slot >= (int)p->slots.size()) return fail(BWTGPU_ERR_INVALID, "bad pipeline slot");
    if (k >= p->resident_slabs) return fail(BWTGPU_ERR_INVALID, "resident slab %zu of %zu reserved", k, p->resident_slabs);
    return pipeline_run(p, slot, p->d_resident + k * p->in_cap, p->resident_bytes[k], delimiter);
}

int bwtgpu_pipeline_upload_presence(bwtgpu_pipeline* p, int slot, size_t bytes, uint32_t presence_out[8]) {
    if (!p || slot < 0 || slot >= (int)p->slots.size() || !presence_out) return fail(BWTGPU_ERR_INVALID, "bad pipeline slot");
    PipeSlot& s = p->slots[(size_t)slot];
    if (s.busy) return fail(BWTGPU_ERR_INVALID, "slot %d is in flight", slot);
    if (bytes > p->in_cap) return fail(BWTGPU_ERR_INVALID, "%zu bytes exceed the slab capacity %zu", bytes, p->in_cap);
    memset(presence_out, 0, 32);
    s.resident = 0;
    if (bytes == 0) return BWTGPU_OK;
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipMemcpyAsync(s.d_in, s.h_in, bytes, hipMemcpyHostToDevice, s.stream));
    HIP_TRY(hipMemsetAsync(s.d_lens, 0, 32, s.stream));
    int rc = bwtgpu_presence_device(p->device, s.d_in, bytes, s.d_lens, s.stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(s.h_lens, s.d_lens, 32, hipMemcpyDeviceToHost, s.stream));
    HIP_TRY(hipStreamSynchronize(s.stream));
    memcpy(presence_out, s.h_lens, 32);
    s.resident = bytes;
    return BWTGPU_OK;
}

int bwtgpu_pipeline_collect(bwtgpu_pipeline* p, int slot, const uint8_t** out, size_t* out_bytes) {
    if (!p || slot < 0 || slot >= (int)p->slots.size() || !out || !out_bytes)
        return fail(BWTGPU_ERR_INVALID, "bad pipeline slot");
    PipeSlot& s = p->slots[(size_t)slot];
    if (!s.busy) return fail(BWTGPU_ERR_INVALID, "slot %d collected without a submit", slot);
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipEventSynchronize(s.done));
    s.busy = false;
    *out = s.h_out;
    *out_bytes = 0;
    if (s.in_bytes == 0) return BWTGPU_OK;
    if (p->direction == BWTGPU_FORWARD) {
        if (s.h_lens[1]) return fail(BWTGPU_ERR_INTERNAL, "forward kernel reported status 0x%x", s.h_lens[1]);
        *out_bytes = s.in_bytes + (s.in_bytes + p->block_size - 1) / p->block_size;
        return BWTGPU_OK;
    }
    const size_t chunk = p->block_size + 1;
    const size_t nb = (s.in_bytes + chunk - 1) / chunk;
    size_t w = 0;
    for (size_t b = 0; b < nb; ++b) {
        const size_t n = std::min(chunk, s.in_bytes - b * chunk);
        const uint32_t len = s.h_lens[b];
        if (len == BWTGPU_LEN_NO_DELIMITER)
            return fail(BWTGPU_ERR_NO_DELIMITER, "block %zu of the slab holds no delimiter byte 0x%02x", b,
                        p->delimiter);
        const size_t from = b * p->block_size + (n - 1 - len);  // texts are right-aligned in their slot
        if (from != w && len) memmove(s.h_out + w, s.h_out + from, len);
        w += len;
    }
    *out_bytes = w;
    return BWTGPU_OK;
}

void bwtgpu_pipeline_destroy(bwtgpu_pipeline* p) { pipeline_free(p); }

void* bwtgpu_alloc_host(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        fail(BWTGPU_ERR_HIP, "hipHostMalloc(%zu) failed", bytes);
        return nullptr;
    }
    return p;
}

void bwtgpu_free_host(void* p) {
    if (p) (void)hipHostFree(p);
}

}  // extern "C"
