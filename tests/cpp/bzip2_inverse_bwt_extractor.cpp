// bzip2_inverse_bwt_extractor -- origin-pointer container -> file, inverted on the GPU.  Same command
// line as reference tests/bzip2_inverse_bwt_extractor.cpp.
#include "bzip2_format.hpp"
#include "origin_cli.hpp"

int main(int argc, char** argv) { return origin_cli_main(argc, argv, Bzip2BWTProcessor::process_file_inverse); }
