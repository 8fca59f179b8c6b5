// bwt.hpp -- forward Burrows-Wheeler transform, GPU-backed.
//
// Same two entry points, argument meaning and return conventions as the reference's
// src/bwt.hpp:9,14; the bodies run on an MI355X through the C ABI in include/bwtgpu.h.
#ifndef BWT_HPP
#define BWT_HPP

#include <cstddef>
#include <string>

// Transforms one block.  The delimiter is appended to `input` before the suffix sort, so
// the result is input.size() + 1 bytes and the delimiter's position in it is the row of
// the original string (the block's primary index).  Safe to call from several threads.
std::string bwt_forward(const std::string& input, char delimiter = '~');

// Transforms a whole file in blocks of `block_size` bytes.  Output layout: one delimiter
// byte (the lowest byte value that does not occur in the file), then one
// (block length + 1)-byte transformed block per input block, in input order.
// Returns 0 on success, 1 on error (message on std::cerr).
int bwt_forward_process_file(const char* input_file, const char* output_file, size_t block_size);

#endif  // BWT_HPP
