// bzip2_format.hpp -- the block container of the reference's bzip2 comparison tool, inverse side.
//
// Mirrors the part of Bzip2BWTProcessor (reference util/bzip2_bwt_utils.hpp:21-125) that does not
// need bzip2's own sources: write_bzip2_block / read_bzip2_block (util/bzip2_bwt_utils.cpp:137-198),
// inverse_bwt (:229-253) and process_file_inverse (:316-349).  A block is a marker byte 0xFF, the
// row of the text among its sorted rotations ("origPtr") as three bytes big-endian, and the last
// column of the sorted rotations; there is no delimiter.  The inverse runs on the GPU
// (bwtgpu_inverse_origin_batch), all blocks of a file in one batch.  The forward half is here too:
// the reference's process_block hands the block to BZ2_blockSort (bzip2's sources, which its checkout
// does not hold); forward_bwt computes the same thing from its definition -- the block's rotations
// sorted, their last column, the row of rotation 0 -- on the GPU (bwtgpu_forward_origin_batch), and
// process_file_forward (reference :255-306) frames the blocks.  It takes the data instead of bzip2's
// EState, which is the one difference in the signature.
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
    // reference util/bzip2_bwt_utils.cpp:104-136 (process_block): one block -> last column and origin row
    static bool forward_bwt(const unsigned char* data, size_t size, std::vector<unsigned char>& bwt_output, int& origPtr);
    // reference util/bzip2_bwt_utils.cpp:255-306: 0 on success, 1 on failure
    static int process_file_forward(const char* input_file, const char* output_file, size_t block_size);
    // reference util/bzip2_bwt_utils.cpp:229-253: one block ("" for an empty one or a failed call)
    static std::string inverse_bwt(const std::vector<unsigned char>& bwt_str, int origPtr);
    // reference util/bzip2_bwt_utils.cpp:316-349: 0 on success, 1 on failure
    static int process_file_inverse(const char* input_file, const char* output_file, size_t block_size);
};
