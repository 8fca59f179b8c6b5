// test_bzip2_format -- the origin-pointer container (reference util/bzip2_bwt_utils.cpp:137-253)
// through the C++ mirror of Bzip2BWTProcessor: containers are made here on the CPU from the textbook
// definition (sort the rotations of every block, last column, row of rotation 0 -- what
// BZ2_blockSort computes), written with write_bzip2_block, inverted on the GPU with
// process_file_inverse / inverse_bwt and compared with the original bytes; the GPU's own forward half
// (forward_bwt / process_file_forward) is checked against the same definition.
#include <algorithm>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <numeric>
#include <string>
#include <vector>

#include "bwtgpu.h"
#include "bzip2_format.hpp"
#include "host_util.hpp"

using namespace bwt_host;

static std::vector<unsigned char> slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

// last column of the sorted rotations of s; returns the row of rotation 0 (the first among equals)
static int rotation_transform(const std::vector<unsigned char>& s, std::vector<unsigned char>& last) {
    const size_t n = s.size();
    std::vector<size_t> rot(n);
    std::iota(rot.begin(), rot.end(), (size_t)0);
    std::stable_sort(rot.begin(), rot.end(), [&](size_t a, size_t b) {
        for (size_t k = 0; k < n; ++k) {
            const unsigned char ca = s[(a + k) % n], cb = s[(b + k) % n];
            if (ca != cb) return ca < cb;
        }
        return false;
    });
    last.resize(n);
    int orig = 0;
    for (size_t i = 0; i < n; ++i) {
        last[i] = s[(rot[i] + n - 1) % n];
        if (rot[i] == 0) orig = (int)i;
    }
    return orig;
}

static int failures = 0;
static void check(bool ok, const std::string& what) {
    std::cout << (ok ? "[PASS] " : "[FAIL] ") << what << std::endl;
    if (!ok) ++failures;
}

static void round_trip(const std::string& name, const std::vector<unsigned char>& data, size_t block_size) {
    const std::string fwd = "build/tmp/origin_temp.bz2bwt", back = "build/tmp/origin_recovered.bin";
    {
        std::ofstream out(fwd, std::ios::binary);
        std::vector<unsigned char> block, last;
        for (size_t off = 0; off < data.size(); off += block_size) {
            block.assign(data.begin() + (long)off, data.begin() + (long)std::min(data.size(), off + block_size));
            const int orig = rotation_transform(block, last);
            if (!Bzip2BWTProcessor::write_bzip2_block(out, orig, last)) {
                check(false, name + ": write_bzip2_block");
                return;
            }
        }
    }
    const int rc = Bzip2BWTProcessor::process_file_inverse(fwd.c_str(), back.c_str(), block_size);
    check(rc == 0 && slurp(back) == data, name + " @ " + format_size(block_size) + " (" + std::to_string(data.size()) + " bytes)");
    std::remove(fwd.c_str());
    std::remove(back.c_str());
}

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "data/canterbury_corpus";
    if (bwtgpu_init(1) < 1) {
        std::cerr << "Error: " << bwtgpu_last_error() << std::endl;
        return 1;
    }
    create_directory("build/tmp");
    {   // single blocks through inverse_bwt: the classic example, a repeated string, one byte, nothing
        std::vector<unsigned char> last;
        const std::string words[] = {"banana", "abababab", "z", "mississippi", "aaaaaaaaaaaaaaaa"};
        for (const std::string& w : words) {
            const std::vector<unsigned char> s(w.begin(), w.end());
            const int orig = rotation_transform(s, last);
            check(Bzip2BWTProcessor::inverse_bwt(last, orig) == w, "inverse_bwt(\"" + w + "\")");
        }
        check(Bzip2BWTProcessor::inverse_bwt({}, 0).empty(), "inverse_bwt(empty)");
        const std::vector<unsigned char> nn = {'n', 'n', 'b', 'a', 'a', 'a'};
        check(Bzip2BWTProcessor::inverse_bwt(nn, 3) == "banana", "inverse_bwt(\"nnbaaa\", 3) == \"banana\"");
    }
    std::vector<unsigned char> text = slurp(dir + "/alice29.txt");
    text.resize(std::min<size_t>(text.size(), 40000));
    for (size_t bs : {(size_t)1, (size_t)7, (size_t)512, (size_t)2000, (size_t)4096, (size_t)16384, (size_t)65536})
        round_trip("alice29.txt[0:40000]", text, bs);
    // repeated text: blocks that are a whole number of periods have equal rotations (the LF permutation
    // falls into several cycles), the others do not
    std::vector<unsigned char> periodic(6300);
    for (size_t i = 0; i < periodic.size(); ++i) periodic[i] = (unsigned char)('a' + i % 7);
    for (size_t bs : {(size_t)1000, (size_t)2100, (size_t)4200, (size_t)6300}) round_trip("period-7", periodic, bs);
    {   // the forward half on the GPU: same last column and row as the definition, and file to file and back
        std::vector<unsigned char> last, want;
        const std::string words[] = {"banana", "abababab", "z", "mississippi", "aaaaaaaaaaaaaaaa"};
        for (const std::string& w : words) {
            const std::vector<unsigned char> s(w.begin(), w.end());
            const int want_orig = rotation_transform(s, want);
            int orig = -1;
            check(Bzip2BWTProcessor::forward_bwt(s.data(), s.size(), last, orig) && last == want && orig == want_orig,
                  "forward_bwt(\"" + w + "\")");
        }
        const std::string src = "build/tmp/origin_src.bin", fwd = "build/tmp/origin_fwd.bz2bwt", back = "build/tmp/origin_back.bin";
        for (size_t bs : {(size_t)1, (size_t)7, (size_t)2000, (size_t)16384, (size_t)65536}) {
            std::ofstream(src, std::ios::binary).write(reinterpret_cast<const char*>(text.data()), (std::streamsize)text.size());
            const int rc1 = Bzip2BWTProcessor::process_file_forward(src.c_str(), fwd.c_str(), bs);
            const size_t nb = (text.size() + bs - 1) / bs;
            const bool sized = get_file_size(fwd) == text.size() + 4 * nb;
            const int rc2 = Bzip2BWTProcessor::process_file_inverse(fwd.c_str(), back.c_str(), bs);
            check(rc1 == 0 && rc2 == 0 && sized && slurp(back) == text, "process_file_forward / _inverse @ " + format_size(bs));
        }
        std::remove(src.c_str());
        std::remove(fwd.c_str());
        std::remove(back.c_str());
    }
    {   // a malformed container: reading stops at the bad block, what came before is kept
        const std::string fwd = "build/tmp/origin_bad.bz2bwt", back = "build/tmp/origin_bad.out";
        std::ofstream out(fwd, std::ios::binary);
        std::vector<unsigned char> last;
        const std::string w = "banana";
        const int orig = rotation_transform(std::vector<unsigned char>(w.begin(), w.end()), last);
        Bzip2BWTProcessor::write_bzip2_block(out, orig, last);
        Bzip2BWTProcessor::write_bzip2_block(out, 9, last);  // row outside the block
        out.close();
        const int rc = Bzip2BWTProcessor::process_file_inverse(fwd.c_str(), back.c_str(), 6);
        const std::vector<unsigned char> got = slurp(back);
        check(rc == 0 && std::string(got.begin(), got.end()) == "banana", "stops at a block whose origPtr is outside it");
        std::remove(fwd.c_str());
        std::remove(back.c_str());
    }
    bwtgpu_shutdown();
    std::cout << (failures ? "SOME TESTS FAILED" : "All tests passed!") << std::endl;
    return failures ? 1 : 0;
}
