// bzip2_format.cpp -- see bzip2_format.hpp.
#include "bzip2_format.hpp"

#include <cstdint>
#include <iostream>

#include "bwtgpu.h"

bool Bzip2BWTProcessor::write_bzip2_block(std::ofstream& out, int origPtr,
                                          const std::vector<unsigned char>& bwt_output) {
    const unsigned char head[4] = {0xFF, (unsigned char)((origPtr >> 16) & 0xFF), (unsigned char)((origPtr >> 8) & 0xFF),
                                   (unsigned char)(origPtr & 0xFF)};
    out.write(reinterpret_cast<const char*>(head), 4);
    if (!out.good()) return false;
    out.write(reinterpret_cast<const char*>(bwt_output.data()), (std::streamsize)bwt_output.size());
    return out.good();
}

bool Bzip2BWTProcessor::read_bzip2_block(std::ifstream& in, size_t block_size, int& origPtr,
                                         std::vector<unsigned char>& bwt_output) {
    unsigned char marker = 0;
    in.read(reinterpret_cast<char*>(&marker), 1);
    if (in.gcount() == 0) return false;
    if (marker != 0xFF) {
        std::cerr << "Error: Invalid marker byte: 0x" << std::hex << (int)marker << std::dec << std::endl;
        return false;
    }
    unsigned char p[3];
    in.read(reinterpret_cast<char*>(p), 3);
    if (in.gcount() != 3) {
        std::cerr << "Error: Failed to read origPtr" << std::endl;
        return false;
    }
    origPtr = (p[0] << 16) | (p[1] << 8) | p[2];
    bwt_output.resize(block_size);
    in.read(reinterpret_cast<char*>(bwt_output.data()), (std::streamsize)block_size);
    const size_t got = (size_t)in.gcount();
    if (got == 0) return false;
    bwt_output.resize(got);
    if (origPtr < 0 || origPtr >= (int)got) {
        std::cerr << "Error: Invalid origPtr: " << origPtr << " (block size: " << got << ")" << std::endl;
        return false;
    }
    return true;
}

std::string Bzip2BWTProcessor::inverse_bwt(const std::vector<unsigned char>& bwt_str, int origPtr) {
    if (bwt_str.empty()) return "";
    std::string text(bwt_str.size(), '\0');
    const uint32_t row = (uint32_t)origPtr;
    if (bwtgpu_inverse_origin_batch(0, bwt_str.data(), bwt_str.size(), bwt_str.size(), &row,
                                    reinterpret_cast<uint8_t*>(&text[0])) != BWTGPU_OK) {
        std::cerr << "Error: " << bwtgpu_last_error() << std::endl;
        return "";
    }
    return text;
}

int Bzip2BWTProcessor::process_file_inverse(const char* input_file, const char* output_file, size_t block_size) {
    std::ifstream in(input_file, std::ios::binary);
    if (!in.is_open()) {
        std::cerr << "Error: Cannot open input file: " << input_file << std::endl;
        return 1;
    }
    std::ofstream out(output_file, std::ios::binary);
    if (!out.is_open()) {
        std::cerr << "Error: Cannot open output file: " << output_file << std::endl;
        return 1;
    }
    if (block_size == 0) return 0;  // the reference reads nothing per block then and stops at once
    // Well-formed blocks up to the end of the input or the first malformed one go to the GPU in batches of
    // bounded size (a batch ends early at a short block, which can only be the file's last), so that a
    // long file needs a batch's worth of host memory and not its own length twice over.
    const size_t kBatchBytes = (size_t)256 << 20;
    std::vector<unsigned char> batch, block, text;
    std::vector<uint32_t> rows;
    auto flush = [&]() -> bool {
        if (batch.empty()) return true;
        text.resize(batch.size());
        if (bwtgpu_inverse_origin_batch(0, batch.data(), batch.size(), block_size, rows.data(), text.data()) != BWTGPU_OK) {
            std::cerr << "Error: " << bwtgpu_last_error() << std::endl;
            return false;
        }
        out.write(reinterpret_cast<const char*>(text.data()), (std::streamsize)text.size());
        batch.clear();
        rows.clear();
        return out.good();
    };
    int origPtr = 0;
    bool full = true;  // only the last block may be short
    while (full && in.good() && read_bzip2_block(in, block_size, origPtr, block)) {
        batch.insert(batch.end(), block.begin(), block.end());
        rows.push_back((uint32_t)origPtr);
        full = block.size() == block_size;
        if (batch.size() + block_size > kBatchBytes && !flush()) return 1;
    }
    if (!flush()) return 1;
    return out.good() ? 0 : 1;
}
