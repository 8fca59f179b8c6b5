// bzip2_bwt_extractor -- file -> origin-pointer container (the block layout bzip2's sorter produces),
// computed on the GPU.  Same command line as reference tests/bzip2_bwt_extractor.cpp.
#include "bzip2_format.hpp"
#include "origin_cli.hpp"

int main(int argc, char** argv) { return origin_cli_main(argc, argv, Bzip2BWTProcessor::process_file_forward); }
