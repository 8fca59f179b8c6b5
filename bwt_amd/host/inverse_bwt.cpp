// inverse_bwt.cpp -- host side of the inverse transform and the `inverse_bwt` tool.
#include "inverse_bwt.hpp"

#include <cstdint>
#include <iostream>
#include <string>

#include "bwtgpu.h"
#include "file_processor.hpp"
#include "slab_runner.hpp"

std::string bwt_inverse(const std::string& bwt_str, char delimiter) {
    if (bwt_str.empty()) {
        std::cerr << "Error: an empty block holds no delimiter" << std::endl;
        return std::string();
    }
    std::string out(bwt_str.size(), '\0');
    size_t n = 0;
    const size_t block_size = bwt_str.size() > 1 ? bwt_str.size() - 1 : 1;
    const int rc = bwtgpu_inverse_batch(0, reinterpret_cast<const uint8_t*>(bwt_str.data()), bwt_str.size(), block_size,
                                        static_cast<uint8_t>(delimiter), reinterpret_cast<uint8_t*>(&out[0]), &n);
    if (rc != BWTGPU_OK) {
        std::cerr << "Error: " << bwtgpu_last_error() << std::endl;
        return std::string();
    }
    out.resize(n);
    return out;
}

int bwt_inverse_process_file(const char* input_file, const char* output_file, size_t block_size) {
    // every transformed block carries one extra byte (its delimiter)
    FileProcessor processor(input_file, output_file, block_size + 1);
    if (!processor.is_open()) return 1;
    char delimiter;
    if (!processor.read_char(delimiter)) {
        std::cerr << "Error: Empty input file" << std::endl;
        processor.close();
        return 1;
    }
    if (block_size == 0) {
        std::cerr << "Error: Block size must be greater than 0" << std::endl;
        return 1;
    }
    const size_t body = processor.input_size() ? processor.input_size() - 1 : 0;
    // a block size beyond the file: the body is one chunk of its own length
    if (block_size > BWTGPU_MAX_BLOCK_SIZE && processor.input_is_regular() && body >= 1 && body - 1 < block_size)
        block_size = body > 1 ? body - 1 : 1;
    if (processor.input_is_regular() && body > 0)  // every block of block_size + 1 bytes gives up its delimiter
        processor.reserve_output(body - (body + block_size) / (block_size + 1));
    bwt_host::trace_point("start");
    const int rc = bwt_host::run_slabs(processor, BWTGPU_INVERSE, block_size, static_cast<unsigned char>(delimiter), body);
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
    return bwt_inverse_process_file(argv[1], argv[2], block_size);
}
#endif  // BUILD_TESTS
