// origin_cli.hpp -- the argument handling shared by build/bzip2_bwt_extractor and
// build/bzip2_inverse_bwt_extractor (reference tests/bzip2_bwt_extractor.cpp:15-43 and
// tests/bzip2_inverse_bwt_extractor.cpp: <input_file> <output_file> [block_size], default 65536).
#pragma once
#include <cstddef>
#include <iostream>
#include <string>

template <typename Run>
int origin_cli_main(int argc, char** argv, Run run) {
    if (argc < 3 || argc > 4) {
        std::cerr << "Usage: " << argv[0] << " <input_file> <output_file> [block_size]" << std::endl;
        std::cerr << "  block_size: size of each block in bytes (default: 65536)" << std::endl;
        return 1;
    }
    size_t block_size = 65536;
    if (argc == 4) {
        block_size = std::stoul(argv[3]);
        if (block_size == 0) {
            std::cerr << "Error: Block size must be greater than 0" << std::endl;
            return 1;
        }
    }
    return run(argv[1], argv[2], block_size);
}
