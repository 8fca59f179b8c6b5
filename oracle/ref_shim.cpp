// ref_shim.cpp -- extern "C" doorway onto the UNMODIFIED reference, for checking only.
//
// TEST INFRASTRUCTURE ONLY.  oracle/Makefile compiles this file together with the
// reference's own src/bwt.cpp, src/inverse_bwt.cpp and src/file_processor.cpp
// (where they lie, read-only, with -DBUILD_TESTS so their main()s are compiled
// out) into oracle/_ref/libbwt_ref.so.  No reference source is copied into this
// repository; the only reference text used here are the public declarations the
// two headers below export.  The resulting library is used (a) to pin
// oracle/bwt_oracle.c and to generate tests/golden/, and (b) as the
// "reference" CPU baseline timed by bench.py.  The shipped library never loads it.
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <string>

#include "bwt.hpp"          // reference src/bwt.hpp:9,14
#include "inverse_bwt.hpp"  // reference src/inverse_bwt.hpp:8,13

int find_unique_char(const char* file_path);  // defined at reference src/bwt.cpp:19 (not in a header)

extern "C" {

// bwt_forward(input, delimiter): out receives len+1 bytes.
size_t ref_bwt_forward(const uint8_t* in, size_t len, uint8_t delimiter, uint8_t* out) {
    std::string r = bwt_forward(std::string(reinterpret_cast<const char*>(in), len),
                                static_cast<char>(delimiter));
    std::memcpy(out, r.data(), r.size());
    return r.size();
}

// bwt_inverse(bwt_str, delimiter): out must hold n bytes; returns the result length.
// The caller must pass a block that contains the delimiter (the reference has
// undefined behaviour otherwise, src/inverse_bwt.cpp:19,49-52).
size_t ref_bwt_inverse(const uint8_t* in, size_t n, uint8_t delimiter, uint8_t* out) {
    std::string r = bwt_inverse(std::string(reinterpret_cast<const char*>(in), n),
                                static_cast<char>(delimiter));
    std::memcpy(out, r.data(), r.size());
    return r.size();
}

int ref_forward_file(const char* in_path, const char* out_path, size_t block_size) {
    return bwt_forward_process_file(in_path, out_path, block_size);
}

int ref_inverse_file(const char* in_path, const char* out_path, size_t block_size) {
    return bwt_inverse_process_file(in_path, out_path, block_size);
}

int ref_find_unique_char(const char* path) { return find_unique_char(path); }

}  // extern "C"
