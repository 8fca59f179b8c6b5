#include "file_processor.hpp"

#include <sys/stat.h>

#include <iostream>

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
    if (!good_ || max_bytes == 0) return 0;
    const size_t got = std::fread(dst, 1, max_bytes, in_);
    if (got < max_bytes) good_ = false;
    return got;
}

std::string FileProcessor::read_chunk() {
    if (scratch_.size() < block_size_) scratch_.resize(block_size_);
    const size_t got = read_bytes(scratch_.data(), block_size_);
    return std::string(scratch_.data(), got);
}

bool FileProcessor::read_char(char& c) { return read_bytes(&c, 1) == 1; }

bool FileProcessor::write_bytes(const char* src, size_t n) {
    std::lock_guard<std::mutex> lock(mutex_);
    if (!out_ || n == 0) return true;
    return std::fwrite(src, 1, n, out_) == n;
}

void FileProcessor::write_chunk(const std::string& chunk) { (void)write_bytes(chunk.data(), chunk.size()); }

void FileProcessor::close() {
    std::lock_guard<std::mutex> lock(mutex_);
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
