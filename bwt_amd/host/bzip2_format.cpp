// bzip2_format.cpp -- see bzip2_format.hpp.
#include "bzip2_format.hpp"

#include <algorithm>
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

bool Bzip2BWTProcessor::forward_bwt(const unsigned char* data, size_t size, std::vector<unsigned char>& bwt_output,
                                    int& origPtr) {
    bwt_output.resize(size);
    origPtr = 0;
    if (size == 0) return true;
    uint32_t row = 0;
    if (bwtgpu_forward_origin_batch(0, data, size, size, bwt_output.data(), &row) != BWTGPU_OK) {
        std::cerr << "Error: " << bwtgpu_last_error() << std::endl;
        return false;
    }
    origPtr = (int)row;
    return true;
}

int Bzip2BWTProcessor::process_file_forward(const char* input_file, const char* output_file, size_t block_size) {
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
    if (block_size == 0) return 0;  // nothing is read per block then
    if (block_size > ((size_t)1 << 24)) {  // the container stores the row in three bytes
        std::cerr << "Error: Block size " << block_size << " exceeds maximum " << ((size_t)1 << 24) << std::endl;
        return 1;
    }
    // whole blocks in batches of bounded size, each batch one call
    const size_t kBatchBytes = (size_t)256 << 20;
    size_t per_batch = std::max<size_t>(1, kBatchBytes / block_size) * block_size;
    in.seekg(0, std::ios::end);
    const std::streamoff file_bytes = in.tellg();  // (-1: not seekable; the full batch buffer then)
    in.seekg(0, std::ios::beg);
    if (file_bytes >= 0 && (size_t)file_bytes < per_batch) per_batch = std::max<size_t>(1, (size_t)file_bytes);
    std::vector<unsigned char> text(per_batch), last, block;
    std::vector<uint32_t> rows;
    while (in.good()) {
        in.read(reinterpret_cast<char*>(text.data()), (std::streamsize)per_batch);
        const size_t got = (size_t)in.gcount();
        if (got == 0) break;
        const size_t nb = (got + block_size - 1) / block_size;
        last.resize(got);
        rows.resize(nb);
        if (bwtgpu_forward_origin_batch(0, text.data(), got, block_size, last.data(), rows.data()) != BWTGPU_OK) {
            std::cerr << "Error: " << bwtgpu_last_error() << std::endl;
            return 1;
        }
        for (size_t b = 0; b < nb; ++b) {
            const size_t lo = b * block_size, hi = std::min(got, lo + block_size);
            block.assign(last.begin() + (long)lo, last.begin() + (long)hi);
            if (!write_bzip2_block(out, (int)rows[b], block)) return 1;
        }
    }
    return 0;
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
