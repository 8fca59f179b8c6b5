#include "file_processor.hpp"

#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <iostream>
#include <thread>

namespace {

// A slab of hundreds of MiB moves between the page cache and pinned memory at the speed of one
// core's memcpy (4-5 GB/s) when a single fread / fwrite does it; that, not the GPU, bounds the
// file-to-file rate.  Bulk transfers on regular files are therefore split over a few threads
// that pread / pwrite disjoint ranges.
constexpr size_t kParallelMin = (size_t)8 << 20;   // below this one call is as fast
constexpr size_t kBytesPerThread = (size_t)4 << 20;
// threads a bulk read or write is split over: BWT_IO_THREADS, default 8
size_t max_io_threads() {
    static const size_t n = [] {
        const char* e = std::getenv("BWT_IO_THREADS");
        const long v = e ? std::atol(e) : 0;
        return (size_t)(v > 0 && v <= 64 ? v : 8);
    }();
    return n;
}

// Runs op(offset, length) -> bytes done over [0, n) in parallel; returns the length of the
// prefix that was fully transferred.
template <typename Op>
size_t parallel_ranges(size_t n, Op op) {
    const size_t nt = std::max<size_t>(1, std::min(max_io_threads(), n / kBytesPerThread));
    const size_t per = ((n + nt - 1) / nt + 4095) & ~(size_t)4095;
    std::vector<size_t> done(nt, 0);
    std::vector<std::thread> threads;
    for (size_t t = 1; t < nt; ++t)
        threads.emplace_back([&, t] {
            const size_t lo = std::min(n, t * per), hi = std::min(n, lo + per);
            done[t] = op(lo, hi - lo);
        });
    done[0] = op(0, std::min(n, per));
    for (std::thread& th : threads) th.join();
    size_t total = 0;
    for (size_t t = 0; t < nt; ++t) {
        const size_t lo = std::min(n, t * per), hi = std::min(n, lo + per);
        total += done[t];
        if (done[t] < hi - lo) break;
    }
    return total;
}

}  // namespace

FileProcessor::FileProcessor(const std::string& input_path, const std::string& output_path, size_t block_size)
    : block_size_(block_size) {
    in_ = std::fopen(input_path.c_str(), "rb");
    if (!in_) {
        std::cerr << "Error: Could not open input file " << input_path << std::endl;
    } else {
        struct stat st;
        if (::stat(input_path.c_str(), &st) == 0 && S_ISREG(st.st_mode)) {
            input_size_ = (size_t)st.st_size;
            regular_ = true;
        }
    }
    if (!output_path.empty()) {
        out_ = std::fopen(output_path.c_str(), "wb");
        if (!out_) {
            std::cerr << "Error: Could not open output file " << output_path << std::endl;
            if (in_) {
                std::fclose(in_);
                in_ = nullptr;
            }
        }
    }
    if (out_) {
        struct stat st;
        out_regular_ = ::fstat(::fileno(out_), &st) == 0 && S_ISREG(st.st_mode);
    }
    if (in_) std::setvbuf(in_, nullptr, _IOFBF, 1 << 20);
    if (out_) std::setvbuf(out_, nullptr, _IOFBF, 1 << 20);
    good_ = in_ != nullptr;
}

FileProcessor::~FileProcessor() { close(); }

bool FileProcessor::is_open() const {
    std::lock_guard<std::mutex> lock(mutex_);
    return in_ != nullptr;
}

bool FileProcessor::has_more_data() const {
    std::lock_guard<std::mutex> lock(mutex_);
    return good_;
}

size_t FileProcessor::read_bytes(char* dst, size_t max_bytes) {
    std::lock_guard<std::mutex> lock(mutex_);
    return read_bytes_locked(dst, max_bytes);
}

size_t FileProcessor::read_bytes_locked(char* dst, size_t max_bytes) {
    if (!good_ || max_bytes == 0) return 0;
    size_t got;
    const off_t pos = regular_ && max_bytes >= kParallelMin ? ::ftello(in_) : (off_t)-1;
    if (pos >= 0 && (size_t)pos < input_size_ && input_size_ - (size_t)pos >= kParallelMin) {
        const size_t want = std::min(max_bytes, input_size_ - (size_t)pos);
        const int fd = ::fileno(in_);
        got = parallel_ranges(want, [&](size_t off, size_t len) {
            size_t did = 0;
            while (did < len) {
                const ssize_t r = ::pread(fd, dst + off + did, len - did, pos + (off_t)(off + did));
                if (r <= 0) break;
                did += (size_t)r;
            }
            return did;
        });
        ::fseeko(in_, pos + (off_t)got, SEEK_SET);
    } else {
        got = std::fread(dst, 1, max_bytes, in_);
    }
    if (got < max_bytes) good_ = false;
    return got;
}

std::string FileProcessor::read_chunk() {
    // the lock covers the shared staging buffer as well as the stream (reference src/file_processor.cpp:38-53
    // holds its lock for the whole call)
    std::lock_guard<std::mutex> lock(mutex_);
    if (scratch_.size() < block_size_) scratch_.resize(block_size_);
    const size_t got = read_bytes_locked(scratch_.data(), block_size_);
    return std::string(scratch_.data(), got);
}

bool FileProcessor::read_char(char& c) { return read_bytes(&c, 1) == 1; }

bool FileProcessor::write_bytes(const char* src, size_t n) {
    std::lock_guard<std::mutex> lock(out_mutex_);
    if (!out_ || n == 0) return true;
    if (out_regular_ && n >= kParallelMin && std::fflush(out_) == 0) {
        const off_t pos = ::ftello(out_);
        if (pos >= 0) {
            const int fd = ::fileno(out_);
            const size_t put = parallel_ranges(n, [&](size_t off, size_t len) {
                size_t did = 0;
                while (did < len) {
                    const ssize_t r = ::pwrite(fd, src + off + did, len - did, pos + (off_t)(off + did));
                    if (r <= 0) break;
                    did += (size_t)r;
                }
                return did;
            });
            ::fseeko(out_, pos + (off_t)n, SEEK_SET);
            return put == n;
        }
    }
    return std::fwrite(src, 1, n, out_) == n;
}

void FileProcessor::write_chunk(const std::string& chunk) { (void)write_bytes(chunk.data(), chunk.size()); }

void FileProcessor::close() {
    std::lock_guard<std::mutex> lock(mutex_);
    std::lock_guard<std::mutex> out_lock(out_mutex_);
    if (in_) {
        std::fclose(in_);
        in_ = nullptr;
    }
    if (out_) {
        std::fclose(out_);
        out_ = nullptr;
    }
    good_ = false;
}

size_t FileProcessor::get_block_size() const { return block_size_; }

size_t FileProcessor::input_size() const { return input_size_; }

bool FileProcessor::input_is_regular() const { return regular_; }
