// bzip2_format.hpp -- the block container of the reference's bzip2 comparison tool, inverse side.
//
// Mirrors the part of Bzip2BWTProcessor (reference util/bzip2_bwt_utils.hpp:21-125) that does not
// need bzip2's own sources: write_bzip2_block / read_bzip2_block (util/bzip2_bwt_utils.cpp:137-198),
// inverse_bwt (:229-253) and process_file_inverse (:316-349).  A block is a marker byte 0xFF, the
// row of the text among its sorted rotations ("origPtr") as three bytes big-endian, and the last
// column of the sorted rotations; there is no delimiter.  The inverse runs on the GPU
// (bwtgpu_inverse_origin_batch), all blocks of a file in one batch.  The forward half
// (process_block / process_file_forward) calls BZ2_blockSort and is not provided.
#pragma once
#include <cstddef>
#include <fstream>
#include <string>
#include <vector>

class Bzip2BWTProcessor {
public:
    // reference util/bzip2_bwt_utils.cpp:137-155
    static bool write_bzip2_block(std::ofstream& out, int origPtr, const std::vector<unsigned char>& bwt_output);
    // reference util/bzip2_bwt_utils.cpp:157-198: false at the end of the input or on a malformed block
    static bool read_bzip2_block(std::ifstream& in, size_t block_size, int& origPtr,
                                 std::vector<unsigned char>& bwt_output);
    // reference util/bzip2_bwt_utils.cpp:229-253: one block ("" for an empty one or a failed call)
    static std::string inverse_bwt(const std::vector<unsigned char>& bwt_str, int origPtr);
    // reference util/bzip2_bwt_utils.cpp:316-349: 0 on success, 1 on failure
    static int process_file_inverse(const char* input_file, const char* output_file, size_t block_size);
};
