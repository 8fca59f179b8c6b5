// inverse_bwt.hpp -- inverse Burrows-Wheeler transform, GPU-backed.
//
// Same two entry points, argument meaning and return conventions as the reference's
// src/inverse_bwt.hpp:8,13; the bodies run on an MI355X through include/bwtgpu.h.
#ifndef INVERSE_BWT_HPP
#define INVERSE_BWT_HPP

#include <cstddef>
#include <string>

// Recovers the text of one transformed block (text length + 1 bytes that contain
// `delimiter`).  Safe to call from several threads.  A block without the delimiter is
// an error: a message goes to std::cerr and the empty string is returned (the reference
// reads out of bounds in that case, src/inverse_bwt.cpp:19,49-52).
std::string bwt_inverse(const std::string& bwt_str, char delimiter = '~');

// Inverts a file written by bwt_forward_process_file with the same `block_size`.
// Returns 0 on success, 1 on error (message on std::cerr).
int bwt_inverse_process_file(const char* input_file, const char* output_file, size_t block_size);

#endif  // INVERSE_BWT_HPP
