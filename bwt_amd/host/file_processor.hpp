// file_processor.hpp -- block-granular binary file reader/writer.
//
// Public interface and observable behaviour follow the reference's
// src/file_processor.hpp:9-46 / src/file_processor.cpp:4-83 (same constructor, same
// method names and results, same "Error: Could not open ..." messages, an empty output
// path meaning read-only).  Two bulk methods are added for the slab pipeline, which
// moves many blocks per call straight into and out of pinned host memory.
#ifndef FILE_PROCESSOR_HPP
#define FILE_PROCESSOR_HPP

#include <cstddef>
#include <cstdio>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

class FileProcessor {
public:
    FileProcessor(const std::string& input_path, const std::string& output_path, size_t block_size);
    ~FileProcessor();
    FileProcessor(const FileProcessor&) = delete;
    FileProcessor& operator=(const FileProcessor&) = delete;

    // True if the input is open (and, when an output path was given, the output too).
    bool is_open() const;
    // True until a read has hit the end of the input (like std::ifstream::good(): a file
    // whose size is an exact multiple of the block size needs one more, empty, read).
    bool has_more_data() const;
    // Next block: block_size bytes, fewer at the end of the file, "" when nothing is left.
    std::string read_chunk();
    // One byte; false at end of file.
    bool read_char(char& c);
    // Appends a block to the output (no-op without an output file or for an empty chunk).
    void write_chunk(const std::string& chunk);
    void close();
    size_t get_block_size() const;

    // ---- additions (not in the reference) ----
    // Reads up to `max_bytes` into `dst`; returns the count (0 at end of file).
    size_t read_bytes(char* dst, size_t max_bytes);
    // Appends `n` bytes to the output; false if the write failed.
    bool write_bytes(const char* src, size_t n);
    // Size of the input file in bytes (0 if it is not a regular file: see input_is_regular()).
    size_t input_size() const;
    bool input_is_regular() const;
    // Tells the writer how long the output will be (both transforms know it from the input's length).  A
    // regular output file is then set to that length and mapped, a helper thread allocates its pages
    // while the first results are still being computed, and write_bytes copies into the mapping from
    // several threads at once.  (Writes through the file descriptor take the file's lock one at a time.)
    // Without the call, or if it fails (false), write_bytes goes through the descriptor as before;
    // output beyond the reservation does too, and close() cuts the file to what was written.
    bool reserve_output(size_t total_bytes);

private:
    size_t read_bytes_locked(char* dst, size_t max_bytes);  // mutex_ held by the caller
    std::FILE* in_ = nullptr;
    std::FILE* out_ = nullptr;
    size_t block_size_;
    bool good_ = false;  // no read has come up short yet
    size_t input_size_ = 0;
    bool regular_ = false;
    bool out_regular_ = false;
    std::vector<char> scratch_;
    // reserve_output's mapping (null: none), its length, bytes written so far, the thread that allocates its pages
    char* map_ = nullptr;
    size_t map_len_ = 0, out_pos_ = 0;
    std::vector<std::thread> prefault_;
    std::atomic<bool> stop_prefault_{false};
    std::atomic<size_t> allocated_{0};  // bytes from the file's start whose pages exist (or one of two end marks)
    void drop_mapping_locked();  // out_mutex_ held: back to the descriptor, positioned after what was written
    // The reference guards both streams with one lock (src/file_processor.hpp:45); here each stream has
    // its own, so that the thread that reads slabs and the thread that writes results do not wait for
    // each other.  Every method is still safe to call from any thread.
    mutable std::mutex mutex_;      // input stream
    mutable std::mutex out_mutex_;  // output stream
};

#endif  // FILE_PROCESSOR_HPP
