// slab_runner.hpp -- drives the C ABI's slab pipelines over a file, in order.
//
// This is what stands where the reference has its reader thread, BlockingQueue, worker
// pool, ReorderBuffer and writer thread (reference src/bwt.cpp:183-233): the calling thread
// reads whole slabs of blocks into pinned memory and submits them round-robin to one
// pipeline per GPU; each slot's upload, kernels and download run on its own HIP stream,
// and results are written back in submission order, so the file order is kept without a
// reorder buffer.  Blocks never interact, so how they are cut into slabs, and which GPU
// gets which slab, cannot change a single output byte.
#ifndef SLAB_RUNNER_HPP
#define SLAB_RUNNER_HPP

#include <cstdlib>
#include <deque>
#include <iostream>
#include <mutex>
#include <vector>

#include "bwtgpu.h"
#include "file_processor.hpp"

namespace bwt_host {

// GPUs to spread slabs over: BWT_DEVICES (default 1), capped by what the library bound.
inline int device_budget() {
    const int have = bwtgpu_init(0);
    if (have <= 0) return have;
    int want = 1;
    if (const char* e = std::getenv("BWT_DEVICES")) {
        const int v = std::atoi(e);
        if (v > 0) want = v;
    }
    return want < have ? want : have;
}

// Blocks per slab: enough blocks to fill every compute unit several times over, within a
// bounded pinned-memory footprint, and no more than the file needs.
inline size_t slab_blocks_for(size_t unit_bytes, size_t file_bytes) {
    const size_t kTargetBlocks = 2048, kMinBytes = (size_t)16 << 20, kMaxBytes = (size_t)512 << 20;
    size_t blocks = kTargetBlocks;
    if (blocks * unit_bytes < kMinBytes) blocks = (kMinBytes + unit_bytes - 1) / unit_bytes;
    if (blocks * unit_bytes > kMaxBytes) blocks = kMaxBytes / unit_bytes;
    if (blocks == 0) blocks = 1;
    if (file_bytes) {
        const size_t need = (file_bytes + unit_bytes - 1) / unit_bytes;
        if (need < blocks) blocks = need ? need : 1;
    }
    return blocks;
}

// Creating a pipeline allocates pinned host slabs and device buffers, which costs milliseconds --
// more than transforming a small file.  Callers that process many files (the performance
// benchmark, a server) mostly repeat the same geometry, so the most recently used pipelines are
// kept and handed out again (the operating system reclaims them at process exit).
class PipelineCache {
public:
    static PipelineCache& instance() {
        static PipelineCache* c = new PipelineCache;  // never destroyed: at exit the HIP runtime may be gone first
        return *c;
    }
    bwtgpu_pipeline* acquire(int dev, int direction, size_t block_size, size_t slab_blocks, int depth) {
        {
            std::lock_guard<std::mutex> lock(mu_);
            for (size_t i = 0; i < idle_.size(); ++i) {
                const Entry& e = idle_[i];
                if (e.dev == dev && e.direction == direction && e.block_size == block_size &&
                    e.slab_blocks == slab_blocks && e.depth == depth) {
                    bwtgpu_pipeline* p = e.p;
                    idle_.erase(idle_.begin() + (long)i);
                    return p;
                }
            }
        }
        return bwtgpu_pipeline_create(dev, direction, block_size, slab_blocks, depth);
    }
    void release(bwtgpu_pipeline* p, int dev, int direction, size_t block_size, size_t slab_blocks, int depth) {
        if (!p) return;
        bwtgpu_pipeline* evict = nullptr;
        {
            std::lock_guard<std::mutex> lock(mu_);
            idle_.push_back({p, dev, direction, block_size, slab_blocks, depth});
            if (idle_.size() > kKeep) {
                evict = idle_.front().p;
                idle_.erase(idle_.begin());
            }
        }
        bwtgpu_pipeline_destroy(evict);
    }
private:
    struct Entry {
        bwtgpu_pipeline* p;
        int dev, direction;
        size_t block_size, slab_blocks;
        int depth;
    };
    static constexpr size_t kKeep = 4;
    std::mutex mu_;
    std::vector<Entry> idle_;
};

// Streams `processor`'s remaining input through the transform and writes the results.
// unit = bytes one block occupies in the INPUT (block_size forward, block_size+1 inverse).
inline int run_slabs(FileProcessor& processor, int direction, size_t block_size, unsigned char delimiter,
                     size_t remaining_bytes) {
    if (processor.input_is_regular() && remaining_bytes == 0) return 0;  // nothing after the header
    const int ndev = device_budget();
    if (ndev <= 0) {
        std::cerr << "Error: " << bwtgpu_last_error() << std::endl;
        return 1;
    }
    const int depth = 2;
    const size_t unit = direction == BWTGPU_FORWARD ? block_size : block_size + 1;
    const size_t slab_blocks = slab_blocks_for(unit, remaining_bytes);
    std::vector<bwtgpu_pipeline*> pipes((size_t)ndev, nullptr);
    struct InFlight { int dev, slot; };
    std::deque<InFlight> inflight;
    int rc = 0;

    auto drain_one = [&]() {
        const InFlight f = inflight.front();
        inflight.pop_front();
        const uint8_t* out = nullptr;
        size_t n = 0;
        if (bwtgpu_pipeline_collect(pipes[(size_t)f.dev], f.slot, &out, &n) != BWTGPU_OK) {
            if (!rc) std::cerr << "Error: " << bwtgpu_last_error() << std::endl;
            rc = 1;
            return;
        }
        if (!rc && !processor.write_bytes(reinterpret_cast<const char*>(out), n)) {
            std::cerr << "Error: short write to the output file" << std::endl;
            rc = 1;
        }
    };

    for (size_t k = 0; !rc; ++k) {
        const int dev = (int)(k % (size_t)ndev);
        const int slot = (int)((k / (size_t)ndev) % (size_t)depth);
        if (inflight.size() == (size_t)ndev * (size_t)depth) drain_one();  // frees exactly (dev, slot)
        if (rc) break;
        if (!pipes[(size_t)dev]) {
            pipes[(size_t)dev] = PipelineCache::instance().acquire(dev, direction, block_size, slab_blocks, depth);
            if (!pipes[(size_t)dev]) {
                std::cerr << "Error: " << bwtgpu_last_error() << std::endl;
                rc = 1;
                break;
            }
        }
        bwtgpu_pipeline* p = pipes[(size_t)dev];
        const size_t cap = bwtgpu_pipeline_input_capacity(p);
        const size_t got = processor.read_bytes(reinterpret_cast<char*>(bwtgpu_pipeline_input(p, slot)), cap);
        if (got == 0) break;
        if (bwtgpu_pipeline_submit(p, slot, got, delimiter) != BWTGPU_OK) {
            std::cerr << "Error: " << bwtgpu_last_error() << std::endl;
            rc = 1;
            break;
        }
        inflight.push_back({dev, slot});
        if (got < cap) break;
    }
    while (!inflight.empty()) drain_one();
    for (size_t d = 0; d < pipes.size(); ++d) {
        if (rc) bwtgpu_pipeline_destroy(pipes[d]);  // do not reuse a pipeline that reported an error
        else PipelineCache::instance().release(pipes[d], (int)d, direction, block_size, slab_blocks, depth);
    }
    return rc;
}

}  // namespace bwt_host

#endif  // SLAB_RUNNER_HPP
