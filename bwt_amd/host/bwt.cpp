// bwt.cpp -- host side of the forward transform: delimiter search, block framing and the
// `bwt` command-line tool.  All arithmetic happens on the GPU behind include/bwtgpu.h.
#include "bwt.hpp"

#include <cstdint>
#include <cstdio>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "bwtgpu.h"
#include "file_processor.hpp"
#include "slab_runner.hpp"

// Lowest byte value (0-255) that does not occur in the file, or -1 if all 256 occur or the
// file cannot be opened -- the contract of reference src/bwt.cpp:19-47.  The file is
// streamed in large slabs; each slab's 256-bit presence mask is computed on the GPU and
// the masks are OR-ed here.
int find_unique_char(const char* file_path) {
    const size_t kSlab = (size_t)64 << 20;
    FileProcessor processor(file_path, "", kSlab);
    if (!processor.is_open()) return -1;
    const size_t slab = processor.input_size() && processor.input_size() < kSlab ? processor.input_size() : kSlab;
    uint8_t* buf = static_cast<uint8_t*>(bwtgpu_alloc_host(slab));
    if (!buf) {
        std::cerr << "Error: " << bwtgpu_last_error() << std::endl;
        return -1;
    }
    uint32_t seen[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int rc = 0;
    while (processor.has_more_data()) {
        const size_t got = processor.read_bytes(reinterpret_cast<char*>(buf), slab);
        if (got == 0) break;
        uint32_t mask[8];
        rc = bwtgpu_find_unused_byte(buf, got, mask);
        if (rc < 0 && rc != BWTGPU_ERR_NO_FREE_BYTE) {
            std::cerr << "Error: " << bwtgpu_last_error() << std::endl;
            bwtgpu_free_host(buf);
            return -1;
        }
        for (int i = 0; i < 8; ++i) seen[i] |= mask[i];
    }
    bwtgpu_free_host(buf);
    processor.close();
    rc = bwtgpu_lowest_clear_bit(seen);
    return rc < 0 ? -1 : rc;
}

std::string bwt_forward(const std::string& input, char delimiter) {
    std::string out(input.size() + 1, delimiter);
    if (input.empty()) return out;
    const int rc = bwtgpu_forward_batch(0, reinterpret_cast<const uint8_t*>(input.data()), input.size(), input.size(),
                                        static_cast<uint8_t>(delimiter), reinterpret_cast<uint8_t*>(&out[0]));
    if (rc != BWTGPU_OK) {
        std::cerr << "Error: " << bwtgpu_last_error() << std::endl;
        return std::string();
    }
    return out;
}

// Whether the file can be opened for reading, without a message: the shortcuts below stay silent
// about a missing input so that the general route reports it exactly once, as the reference does
// (its find_unique_char opens the file first, src/bwt.cpp:23-26, and the constructor at
// src/file_processor.cpp:7-10 prints the message).
static bool readable(const char* path) {
    std::FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    std::fclose(f);
    return true;
}

// A regular file that fits one slab is read and uploaded ONCE: the delimiter search runs on the
// uploaded bytes and the transform then starts from them.  Returns -1 when the file does not
// qualify (the caller then takes the general two-pass route), else the result code.
static int forward_single_slab(const char* input_file, const char* output_file, size_t block_size) {
    if (block_size == 0 || block_size > BWTGPU_MAX_BLOCK_SIZE || !readable(input_file)) return -1;
    FileProcessor reader(input_file, "", block_size);
    if (!reader.is_open() || !reader.input_is_regular() || reader.input_size() == 0) return -1;
    const size_t size = reader.input_size();
    const size_t slab_blocks = bwt_host::slab_blocks_for(block_size, size);
    if (slab_blocks * block_size < size || bwt_host::device_budget() <= 0) return -1;
    const int depth = 2;
    bwt_host::PipelineCache& cache = bwt_host::PipelineCache::instance();
    bwt_host::PipelineCache::Lease lease = cache.acquire(0, BWTGPU_FORWARD, block_size, slab_blocks, depth);
    if (!lease.p) return -1;
    bwtgpu_pipeline* p = lease.p;
    int rc = 1;
    bool reusable = true;
    do {
        const size_t got = reader.read_bytes(reinterpret_cast<char*>(bwtgpu_pipeline_input(p, 0)),
                                             bwtgpu_pipeline_input_capacity(p));
        char extra;
        if (got != size || reader.read_char(extra)) {  // the file changed under us: let the general route handle it
            rc = -1;
            break;
        }
        uint32_t mask[8];
        if (bwtgpu_pipeline_upload_presence(p, 0, got, mask) != BWTGPU_OK) {
            std::cerr << "Error: " << bwtgpu_last_error() << std::endl;
            reusable = false;
            break;
        }
        const int delimiter_byte = bwtgpu_lowest_clear_bit(mask);
        if (delimiter_byte < 0) {
            std::cerr << "Error: Cannot find a unique delimiter (all 256 byte values appear in file)" << std::endl;
            break;
        }
        const char delimiter = static_cast<char>(delimiter_byte);
        FileProcessor writer(input_file, output_file, block_size);
        if (!writer.is_open()) break;
        writer.reserve_output(1 + got + (got + block_size - 1) / block_size);  // header, bytes, one delimiter per block
        const uint8_t* out = nullptr;
        size_t n = 0;
        if (bwtgpu_pipeline_run(p, 0, got, static_cast<uint8_t>(delimiter), &out, &n) != BWTGPU_OK) {
            std::cerr << "Error: " << bwtgpu_last_error() << std::endl;
            reusable = false;
            break;
        }
        if (!writer.write_bytes(&delimiter, 1) || !writer.write_bytes(reinterpret_cast<const char*>(out), n)) {
            std::cerr << "Error: short write to the output file" << std::endl;
            break;
        }
        writer.close();
        rc = 0;
    } while (false);
    if (reusable) cache.release(lease, 0, BWTGPU_FORWARD, block_size, depth);
    else bwtgpu_pipeline_destroy(p);
    return rc;
}

int bwt_forward_process_file(const char* input_file, const char* output_file, size_t block_size) {
    if (block_size > BWTGPU_MAX_BLOCK_SIZE && readable(input_file)) {
        // A block size beyond the file makes the file one block of its own length; only a file
        // that really has to be cut into blocks longer than the library's limit is refused (later).
        FileProcessor probe(input_file, "", 1);
        if (probe.is_open() && probe.input_is_regular() && probe.input_size() < block_size)
            block_size = probe.input_size() ? probe.input_size() : 1;
    }
    bwt_host::trace_point("start");
    const int fast = forward_single_slab(input_file, output_file, block_size);
    if (fast >= 0) return fast;
    if (block_size != 0 && block_size <= BWTGPU_MAX_BLOCK_SIZE && readable(input_file)) {
        // a regular file of several slabs that fits the devices: read and uploaded once (the delimiter is
        // found on the uploaded bytes), see run_forward_resident
        FileProcessor reader(input_file, "", block_size);
        std::unique_ptr<FileProcessor> writer;
        const int rc = !reader.is_open() ? -1 : bwt_host::run_forward_resident(reader, block_size, [&](char delimiter) -> FileProcessor* {
            writer.reset(new FileProcessor(input_file, output_file, block_size));
            if (!writer->is_open()) return nullptr;
            if (!writer->write_bytes(&delimiter, 1)) {
                std::cerr << "Error: short write to the output file" << std::endl;
                return nullptr;
            }
            const size_t n = reader.input_size();
            writer->reserve_output(1 + n + (n + block_size - 1) / block_size);
            return writer.get();
        });
        if (writer) {
            writer->close();
            bwt_host::trace_point("closed");
        }
        if (rc >= 0) return rc;
    }
    const int delimiter_byte = find_unique_char(input_file);
    bwt_host::trace_point("delimiter found");
    if (delimiter_byte == -1) {
        std::cerr << "Error: Cannot find a unique delimiter (all 256 byte values appear in file)" << std::endl;
        return 1;
    }
    const char delimiter = static_cast<char>(delimiter_byte);

    FileProcessor processor(input_file, output_file, block_size);
    if (!processor.is_open()) return 1;
    if (block_size == 0) {
        std::cerr << "Error: Block size must be greater than 0" << std::endl;
        return 1;
    }
    // header: the delimiter, then the transformed blocks in input order
    if (!processor.write_bytes(&delimiter, 1)) {
        std::cerr << "Error: short write to the output file" << std::endl;
        return 1;
    }
    if (processor.input_is_regular())
        processor.reserve_output(1 + processor.input_size() + (processor.input_size() + block_size - 1) / block_size);
    const int rc = bwt_host::run_slabs(processor, BWTGPU_FORWARD, block_size, static_cast<unsigned char>(delimiter),
                                       processor.input_size());
    bwt_host::trace_point("slabs done");
    processor.close();
    bwt_host::trace_point("closed");
    return rc;
}

#ifndef BUILD_TESTS
int main(int argc, char* argv[]) {
    if (argc < 3 || argc > 4) {
        std::cerr << "Usage: " << argv[0] << " <input_file> <output_file> [block_size]" << std::endl;
        std::cerr << "  block_size: size of each block in bytes (default: 128)" << std::endl;
        return 1;
    }
    size_t block_size = 128;
    if (argc == 4) {
        block_size = std::stoul(argv[3]);
        if (block_size == 0) {
            std::cerr << "Error: Block size must be greater than 0" << std::endl;
            return 1;
        }
    }
    return bwt_forward_process_file(argv[1], argv[2], block_size);
}
#endif  // BUILD_TESTS
