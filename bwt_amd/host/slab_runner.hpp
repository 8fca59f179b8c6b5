// slab_runner.hpp -- drives the C ABI's slab pipelines over a file, in order.
//
// This is what stands where the reference has its reader thread, BlockingQueue, worker
// pool, ReorderBuffer and writer thread (reference src/bwt.cpp:183-233): whole slabs of blocks
// are read into pinned memory and submitted round-robin to one pipeline per GPU; each slot's
// upload, kernels and download run on its own HIP stream, and results are written back in
// submission order, so the file order is kept without a reorder buffer.  Blocks never interact,
// so how they are cut into slabs, and which GPU gets which slab, cannot change a single output
// byte.
// A file of one slab is driven by the calling thread alone (run_slabs_serial: thread start-up
// would cost more than the transform).  Longer inputs get the reference's three roles back as
// threads (run_slabs_threaded): the calling thread reads, ONE thread per GPU submits that GPU's
// slabs (a submit that has to wait for the device -- the device-wide path for few large blocks
// reads intermediate counts back -- then holds up that GPU only), and one thread collects the
// results in file order and writes them, which is what frees a slot for the reader.
#ifndef SLAB_RUNNER_HPP
#define SLAB_RUNNER_HPP

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <iostream>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "bwtgpu.h"
#include "file_processor.hpp"

namespace bwt_host {

// BWT_HOST_TRACE=1: "bwt host: +<seconds since the program's first call here> <what>" on stderr
inline void trace_point(const char* what) {
    static const bool on = std::getenv("BWT_HOST_TRACE") != nullptr;
    static const auto t0 = std::chrono::steady_clock::now();
    if (on) std::fprintf(stderr, "bwt host: +%.3f s %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), what);
}

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

// Most bytes in a slab: BWT_SLAB_MIB, default 64.  A long file goes through many slabs of this size: the
// four pinned buffers of a pipeline (two slots, in and out) are then touched for the first time during
// the first two slabs only and stay mapped for the rest, and reading, transfers, kernels and writing of
// neighbouring slabs overlap.  (Slabs of 512 MiB made every write the first touch of its buffer:
// 0.24 s per slab in page faults.)
inline size_t slab_max_bytes() {
    static const size_t n = [] {
        const char* e = std::getenv("BWT_SLAB_MIB");
        const long v = e ? std::atol(e) : 0;
        return (size_t)(v >= 1 && v <= 4096 ? v : 64) << 20;
    }();
    return n;
}

// Blocks per slab: enough blocks to fill every compute unit several times over, within a
// bounded pinned-memory footprint, and no more than the file needs.
inline size_t slab_blocks_for(size_t unit_bytes, size_t file_bytes) {
    const size_t kTargetBlocks = 2048, kMinBytes = (size_t)16 << 20, kMaxBytes = slab_max_bytes();
    size_t blocks = kTargetBlocks;
    if (blocks * unit_bytes < kMinBytes) blocks = (kMinBytes + unit_bytes - 1) / unit_bytes;
    if (blocks * unit_bytes > kMaxBytes) blocks = kMaxBytes / unit_bytes;
    if (blocks == 0) blocks = 1;
    if (file_bytes) {
        size_t need = (file_bytes + unit_bytes - 1) / unit_bytes;
        if (need == 0) need = 1;
        // few distinct slab sizes, none tiny, so that files of different lengths share idle pipelines
        const size_t kSmallSlab = (size_t)4 << 20;
        size_t pow2 = 1;
        while (pow2 < need || pow2 * unit_bytes < kSmallSlab) pow2 *= 2;
        if (pow2 < blocks) blocks = pow2;
    }
    return blocks;
}

// Creating a pipeline allocates pinned host slabs and device buffers, which costs milliseconds --
// more than transforming a small file.  Callers that process many files (the performance
// benchmark, a server) mostly repeat a few geometries, so idle pipelines are kept and handed out
// again: any idle one for the same device, direction and block size that holds at least the
// slab asked for will do.  The most recent kKeepAlways stay whatever their size, older ones
// only while all idle slabs together stay within kIdleBytes (the operating system reclaims
// them at process exit).
class PipelineCache {
public:
    struct Lease {
        bwtgpu_pipeline* p = nullptr;
        size_t slab_blocks = 0;  // what the pipeline was created with (>= what was asked for)
    };
    static PipelineCache& instance() {
        static PipelineCache* c = new PipelineCache;  // never destroyed: at exit the HIP runtime may be gone first
        return *c;
    }
    Lease acquire(int dev, int direction, size_t block_size, size_t slab_blocks, int depth) {
        {
            std::lock_guard<std::mutex> lock(mu_);
            size_t best = idle_.size();
            for (size_t i = 0; i < idle_.size(); ++i) {
                const Entry& e = idle_[i];
                if (e.dev == dev && e.direction == direction && e.block_size == block_size && e.depth == depth &&
                    e.slab_blocks >= slab_blocks && (best == idle_.size() || e.slab_blocks < idle_[best].slab_blocks))
                    best = i;
            }
            if (best != idle_.size()) {
                Lease l;
                l.p = idle_[best].p;
                l.slab_blocks = idle_[best].slab_blocks;
                idle_.erase(idle_.begin() + (long)best);
                return l;
            }
        }
        Lease l;
        l.p = bwtgpu_pipeline_create(dev, direction, block_size, slab_blocks, depth);
        l.slab_blocks = slab_blocks;
        return l;
    }
    void release(const Lease& l, int dev, int direction, size_t block_size, int depth) {
        if (!l.p) return;
        std::vector<bwtgpu_pipeline*> evict;
        {
            std::lock_guard<std::mutex> lock(mu_);
            idle_.push_back({l.p, dev, direction, block_size, l.slab_blocks, depth});
            for (;;) {
                size_t bytes = 0;
                for (const Entry& e : idle_) bytes += e.slab_blocks * (e.block_size + 1) * (size_t)e.depth;
                if (idle_.size() <= kKeepAlways || (idle_.size() <= kKeepMost && bytes <= kIdleBytes)) break;
                evict.push_back(idle_.front().p);
                idle_.erase(idle_.begin());
            }
        }
        for (bwtgpu_pipeline* p : evict) bwtgpu_pipeline_destroy(p);
    }
private:
    struct Entry {
        bwtgpu_pipeline* p;
        int dev, direction;
        size_t block_size, slab_blocks;
        int depth;
    };
    static constexpr size_t kKeepAlways = 4, kKeepMost = 48;
    static constexpr size_t kIdleBytes = (size_t)256 << 20;
    std::mutex mu_;
    std::vector<Entry> idle_;
};

// Streams `processor`'s remaining input through the transform and writes the results.
// unit = bytes one block occupies in the INPUT (block_size forward, block_size+1 inverse).
inline int run_slabs_serial(FileProcessor& processor, int direction, size_t block_size, unsigned char delimiter,
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
    // BWT_HOST_TRACE=1: where the calling thread's time goes (seconds), on stderr
    const bool trace = std::getenv("BWT_HOST_TRACE") != nullptr;
    double t_acquire = 0, t_read = 0, t_submit = 0, t_wait = 0, t_write = 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    std::vector<PipelineCache::Lease> leases((size_t)ndev);
    std::vector<bwtgpu_pipeline*> pipes((size_t)ndev, nullptr);
    struct InFlight { int dev, slot; };
    std::deque<InFlight> inflight;
    int rc = 0;

    auto drain_one = [&]() {
        const InFlight f = inflight.front();
        inflight.pop_front();
        const uint8_t* out = nullptr;
        size_t n = 0;
        const double t0 = now();
        const int crc = bwtgpu_pipeline_collect(pipes[(size_t)f.dev], f.slot, &out, &n);
        t_wait += now() - t0;
        if (crc != BWTGPU_OK) {
            if (!rc) std::cerr << "Error: " << bwtgpu_last_error() << std::endl;
            rc = 1;
            return;
        }
        const double t1 = now();
        if (!rc && !processor.write_bytes(reinterpret_cast<const char*>(out), n)) {
            std::cerr << "Error: short write to the output file" << std::endl;
            rc = 1;
        }
        t_write += now() - t1;
    };

    for (size_t k = 0; !rc; ++k) {
        const int dev = (int)(k % (size_t)ndev);
        const int slot = (int)((k / (size_t)ndev) % (size_t)depth);
        if (inflight.size() == (size_t)ndev * (size_t)depth) drain_one();  // frees exactly (dev, slot)
        if (rc) break;
        if (!pipes[(size_t)dev]) {
            const double t0 = now();
            leases[(size_t)dev] = PipelineCache::instance().acquire(dev, direction, block_size, slab_blocks, depth);
            pipes[(size_t)dev] = leases[(size_t)dev].p;
            t_acquire += now() - t0;
            if (!pipes[(size_t)dev]) {
                std::cerr << "Error: " << bwtgpu_last_error() << std::endl;
                rc = 1;
                break;
            }
        }
        bwtgpu_pipeline* p = pipes[(size_t)dev];
        const size_t cap = bwtgpu_pipeline_input_capacity(p);
        const double t0 = now();
        const size_t got = processor.read_bytes(reinterpret_cast<char*>(bwtgpu_pipeline_input(p, slot)), cap);
        t_read += now() - t0;
        if (got == 0) break;
        const double t1 = now();
        const int src = bwtgpu_pipeline_submit(p, slot, got, delimiter);
        t_submit += now() - t1;
        if (src != BWTGPU_OK) {
            std::cerr << "Error: " << bwtgpu_last_error() << std::endl;
            rc = 1;
            break;
        }
        inflight.push_back({dev, slot});
        if (got < cap) break;
    }
    while (!inflight.empty()) drain_one();
    if (trace)
        std::fprintf(stderr, "bwt host: pipeline %.3f, read %.3f, submit %.3f, wait %.3f, write %.3f s (slabs of %zu blocks)\n",
                     t_acquire, t_read, t_submit, t_wait, t_write, slab_blocks);
    for (size_t d = 0; d < pipes.size(); ++d) {
        if (rc) bwtgpu_pipeline_destroy(pipes[d]);  // do not reuse a pipeline that reported an error
        else PipelineCache::instance().release(leases[d], (int)d, direction, block_size, depth);
    }
    return rc;
}


// The same with a reader (the calling thread), one submitting thread per GPU and a writer.
// Slab k goes to GPU k % G, slot (k / G) % depth; a slot is Free -> Filled (read) -> Submitted ->
// Free again (collected and written).
inline int run_slabs_threaded(FileProcessor& processor, int direction, size_t block_size, unsigned char delimiter,
                              size_t remaining_bytes) {
    const int ndev = device_budget();
    if (ndev <= 0) {
        std::cerr << "Error: " << bwtgpu_last_error() << std::endl;
        return 1;
    }
    const int depth = 2;
    const size_t unit = direction == BWTGPU_FORWARD ? block_size : block_size + 1;
    const size_t slab_blocks = slab_blocks_for(unit, remaining_bytes);
    const bool trace = std::getenv("BWT_HOST_TRACE") != nullptr;
    enum State { Free, Filled, Submitted };
    struct Slot {
        State st = Free;
        size_t bytes = 0;
    };
    const size_t kNone = (size_t)-1;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Slot> ring((size_t)ndev * (size_t)depth);
    std::vector<PipelineCache::Lease> leases((size_t)ndev);
    std::vector<bwtgpu_pipeline*> pipes((size_t)ndev, nullptr);
    size_t total = kNone;  // number of slabs, once the reader knows it
    bool failed = false;
    auto slot_of = [&](size_t k) -> Slot& { return ring[(k % (size_t)ndev) * (size_t)depth + (k / (size_t)ndev) % (size_t)depth]; };
    auto fail_with = [&](const std::string& msg) {
        std::lock_guard<std::mutex> lock(mu);
        if (!failed) std::cerr << "Error: " << msg << std::endl;
        failed = true;
        cv.notify_all();
    };

    auto submitter = [&](int dev) {
        for (size_t k = (size_t)dev;; k += (size_t)ndev) {
            size_t bytes;
            {
                std::unique_lock<std::mutex> lock(mu);
                cv.wait(lock, [&] { return failed || slot_of(k).st == Filled || (total != kNone && k >= total); });
                if (failed || slot_of(k).st != Filled) return;
                bytes = slot_of(k).bytes;
            }
            const int slot = (int)((k / (size_t)ndev) % (size_t)depth);
            if (bwtgpu_pipeline_submit(pipes[(size_t)dev], slot, bytes, delimiter) != BWTGPU_OK) {
                fail_with(bwtgpu_last_error());
                return;
            }
            std::lock_guard<std::mutex> lock(mu);
            slot_of(k).st = Submitted;
            cv.notify_all();
        }
    };
    double t_wait = 0, t_write = 0, t_read = 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto writer = [&] {
        for (size_t k = 0;; ++k) {
            {
                std::unique_lock<std::mutex> lock(mu);
                cv.wait(lock, [&] { return failed || slot_of(k).st == Submitted || (total != kNone && k >= total); });
                if (failed || slot_of(k).st != Submitted) return;
            }
            const int dev = (int)(k % (size_t)ndev), slot = (int)((k / (size_t)ndev) % (size_t)depth);
            const uint8_t* out = nullptr;
            size_t n = 0;
            const double t0 = now();
            if (bwtgpu_pipeline_collect(pipes[(size_t)dev], slot, &out, &n) != BWTGPU_OK) {
                fail_with(bwtgpu_last_error());
                return;
            }
            const double t1 = now();
            if (!processor.write_bytes(reinterpret_cast<const char*>(out), n)) {
                fail_with("short write to the output file");
                return;
            }
            t_wait += t1 - t0;
            t_write += now() - t1;
            if (trace) std::fprintf(stderr, "bwt host: slab %zu: %zu bytes, waited %.3f s, wrote in %.3f s\n", k, n, t1 - t0, now() - t1);
            std::lock_guard<std::mutex> lock(mu);
            slot_of(k).st = Free;
            cv.notify_all();
        }
    };

    std::vector<std::thread> threads;
    std::thread writer_thread(writer);
    for (size_t k = 0;; ++k) {
        const int dev = (int)(k % (size_t)ndev), slot = (int)((k / (size_t)ndev) % (size_t)depth);
        {
            std::unique_lock<std::mutex> lock(mu);
            cv.wait(lock, [&] { return failed || slot_of(k).st == Free; });
            if (failed) break;
        }
        if (!pipes[(size_t)dev]) {  // first slab for this GPU: its pipeline and its submitting thread
            leases[(size_t)dev] = PipelineCache::instance().acquire(dev, direction, block_size, slab_blocks, depth);
            pipes[(size_t)dev] = leases[(size_t)dev].p;
            if (!pipes[(size_t)dev]) {
                fail_with(bwtgpu_last_error());
                break;
            }
            trace_point("pipeline ready");
            threads.emplace_back(submitter, dev);
        }
        bwtgpu_pipeline* p = pipes[(size_t)dev];
        const size_t cap = bwtgpu_pipeline_input_capacity(p);
        const double t0 = now();
        const size_t got = processor.read_bytes(reinterpret_cast<char*>(bwtgpu_pipeline_input(p, slot)), cap);
        t_read += now() - t0;
        std::lock_guard<std::mutex> lock(mu);
        if (got == 0) {
            total = k;
        } else {
            slot_of(k).bytes = got;
            slot_of(k).st = Filled;
            if (got < cap) total = k + 1;
        }
        cv.notify_all();
        if (total != kNone) break;
    }
    {
        std::lock_guard<std::mutex> lock(mu);
        if (total == kNone) total = 0;  // only after a failure: lets the waiting threads go
        cv.notify_all();
    }
    for (std::thread& t : threads) t.join();
    writer_thread.join();
    if (trace)
        std::fprintf(stderr, "bwt host: %d GPU(s), read %.3f s (reader thread), wait %.3f, write %.3f s (writer thread), slabs of %zu blocks\n",
                     ndev, t_read, t_wait, t_write, slab_blocks);
    for (size_t d = 0; d < pipes.size(); ++d) {
        if (!pipes[d]) continue;
        if (failed) bwtgpu_pipeline_destroy(pipes[d]);  // do not reuse a pipeline that reported an error
        else PipelineCache::instance().release(leases[d], (int)d, direction, block_size, depth);
    }
    return failed ? 1 : 0;
}

// Most bytes of one file kept on the devices between the upload and the transform: BWT_RESIDENT_MIB per
// GPU, default 16384 (of 288 GB; longer files take the two-pass route).  0 turns the route off.
inline size_t resident_max_bytes() {
    static const size_t n = [] {
        const char* e = std::getenv("BWT_RESIDENT_MIB");
        const long v = e ? std::atol(e) : -1;
        return (size_t)(v >= 0 && v <= 262144 ? v : 16384) << 20;
    }();
    return n;
}

// Forward of a regular file in ONE pass over it: every slab is read and uploaded once, to the GPU that will
// transform it (slab k: GPU k % G), which looks at its byte values on the way; once all are up the masks
// give the delimiter (the reference's first pass over the file, src/bwt.cpp:19-47, reads it a second
// time for this), the header is written and the resident slabs are transformed, collected in file order
// and written.  `open_output(delimiter)` is called once the delimiter is known and returns the processor
// to write to (nullptr: failed, message printed).  Returns -1 if the route does not apply (the caller
// then takes the two-pass route), else the exit code.
template <typename OpenOutput>
inline int run_forward_resident(FileProcessor& reader, size_t block_size, OpenOutput open_output) {
    const size_t size = reader.input_size();
    if (!reader.input_is_regular() || size == 0) return -1;
    const int ndev = device_budget();
    if (ndev <= 0) return -1;
    if (size > resident_max_bytes() * (size_t)ndev) return -1;
    const int depth = 2;
    const size_t slab_blocks = slab_blocks_for(block_size, size);
    const size_t slab_bytes = slab_blocks * block_size;
    const size_t nslabs = (size + slab_bytes - 1) / slab_bytes;
    const size_t per_dev = (nslabs + (size_t)ndev - 1) / (size_t)ndev;
    std::vector<PipelineCache::Lease> leases((size_t)ndev);
    std::vector<bwtgpu_pipeline*> pipes((size_t)ndev, nullptr);
    int rc = 0;
    auto give_back = [&](bool reusable) {
        for (size_t d = 0; d < pipes.size(); ++d) {
            if (!pipes[d]) continue;
            if (reusable) PipelineCache::instance().release(leases[d], (int)d, BWTGPU_FORWARD, block_size, depth);
            else bwtgpu_pipeline_destroy(pipes[d]);
        }
    };
    auto gpu_error = [&] {
        std::cerr << "Error: " << bwtgpu_last_error() << std::endl;
        give_back(false);
        return 1;
    };
    const int used = (int)std::min<size_t>((size_t)ndev, nslabs);
    for (int d = 0; d < used; ++d) {
        leases[(size_t)d] = PipelineCache::instance().acquire(d, BWTGPU_FORWARD, block_size, slab_blocks, depth);
        pipes[(size_t)d] = leases[(size_t)d].p;
        if (!pipes[(size_t)d]) {  // (e.g. no memory for another pipeline): let the other route try
            give_back(true);
            return -1;
        }
        if (bwtgpu_pipeline_reserve(pipes[(size_t)d], per_dev) != BWTGPU_OK) {
            give_back(true);
            return -1;
        }
    }
    trace_point("pipelines ready");
    // pass 1: read, upload, look
    size_t total = 0, k = 0;
    for (; k < nslabs; ++k) {
        const int dev = (int)(k % (size_t)ndev), slot = (int)((k / (size_t)ndev) % (size_t)depth);
        bwtgpu_pipeline* p = pipes[(size_t)dev];
        if (bwtgpu_pipeline_stage_wait(p, slot) != BWTGPU_OK) return gpu_error();
        const size_t got = reader.read_bytes(reinterpret_cast<char*>(bwtgpu_pipeline_input(p, slot)), slab_bytes);
        if (got == 0) break;
        if (bwtgpu_pipeline_stage(p, slot, k / (size_t)ndev, got) != BWTGPU_OK) return gpu_error();
        total += got;
        if (got < slab_bytes) {
            ++k;
            break;
        }
    }
    char extra;
    if (total != size || reader.read_char(extra)) {  // the file changed under us
        give_back(true);
        return -1;
    }
    const size_t staged = k;
    uint32_t seen[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int d = 0; d < used; ++d) {
        uint32_t mask[8];
        if (bwtgpu_pipeline_presence(pipes[(size_t)d], mask) != BWTGPU_OK) return gpu_error();
        for (int i = 0; i < 8; ++i) seen[i] |= mask[i];
    }
    trace_point("file resident, delimiter known");
    const int delimiter_byte = bwtgpu_lowest_clear_bit(seen);
    if (delimiter_byte < 0) {
        std::cerr << "Error: Cannot find a unique delimiter (all 256 byte values appear in file)" << std::endl;
        give_back(true);
        return 1;
    }
    FileProcessor* writer = open_output(static_cast<char>(delimiter_byte));
    if (!writer) {
        give_back(true);
        return 1;
    }
    // pass 2: transform the resident slabs; up to `depth` per GPU in flight, collected and written in file order
    size_t submitted = 0, collected = 0;
    while (collected < staged && !rc) {
        while (submitted < staged && submitted - collected < (size_t)ndev * (size_t)depth) {
            const int dev = (int)(submitted % (size_t)ndev), slot = (int)((submitted / (size_t)ndev) % (size_t)depth);
            if (bwtgpu_pipeline_submit_resident(pipes[(size_t)dev], slot, submitted / (size_t)ndev, (uint8_t)delimiter_byte) != BWTGPU_OK)
                return gpu_error();
            ++submitted;
        }
        const int dev = (int)(collected % (size_t)ndev), slot = (int)((collected / (size_t)ndev) % (size_t)depth);
        const uint8_t* out = nullptr;
        size_t n = 0;
        if (bwtgpu_pipeline_collect(pipes[(size_t)dev], slot, &out, &n) != BWTGPU_OK) return gpu_error();
        if (!writer->write_bytes(reinterpret_cast<const char*>(out), n)) {
            std::cerr << "Error: short write to the output file" << std::endl;
            rc = 1;
        }
        ++collected;
    }
    if (rc) {  // slots may still be in flight: do not hand such a pipeline to the next caller
        give_back(false);
        return rc;
    }
    trace_point("slabs done");
    give_back(true);
    return 0;
}

inline int run_slabs(FileProcessor& processor, int direction, size_t block_size, unsigned char delimiter,
                     size_t remaining_bytes) {
    const size_t unit = direction == BWTGPU_FORWARD ? block_size : block_size + 1;
    const bool one_slab = processor.input_is_regular() && remaining_bytes <= slab_blocks_for(unit, remaining_bytes) * unit;
    if (one_slab || std::getenv("BWT_HOST_SERIAL")) return run_slabs_serial(processor, direction, block_size, delimiter, remaining_bytes);
    return run_slabs_threaded(processor, direction, block_size, delimiter, remaining_bytes);
}

}  // namespace bwt_host

#endif  // SLAB_RUNNER_HPP
