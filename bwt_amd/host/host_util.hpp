// host_util.hpp -- small file / formatting helpers shared by the test and benchmark tools.
#ifndef BWT_HOST_UTIL_HPP
#define BWT_HOST_UTIL_HPP

#include <algorithm>
#include <cstdio>
#include <filesystem>
#include <fstream>
#include <iomanip>
#include <sstream>
#include <string>
#include <vector>

namespace bwt_host {

namespace fs = std::filesystem;

inline bool file_exists(const std::string& p) {
    std::error_code ec;
    return fs::is_regular_file(p, ec);
}
inline bool directory_exists(const std::string& p) {
    std::error_code ec;
    return fs::is_directory(p, ec);
}
inline bool create_directory(const std::string& p) {
    std::error_code ec;
    fs::create_directories(p, ec);
    return directory_exists(p);
}
inline size_t get_file_size(const std::string& p) {
    std::error_code ec;
    const auto n = fs::file_size(p, ec);
    return ec ? 0 : (size_t)n;
}
inline std::vector<std::string> list_files_in_directory(const std::string& dir) {
    std::vector<std::string> names;
    std::error_code ec;
    for (const auto& e : fs::directory_iterator(dir, ec))
        if (e.is_regular_file()) names.push_back(e.path().filename().string());
    std::sort(names.begin(), names.end());
    return names;
}
inline bool files_are_identical(const std::string& a, const std::string& b) {
    if (get_file_size(a) != get_file_size(b)) return false;
    std::ifstream fa(a, std::ios::binary), fb(b, std::ios::binary);
    if (!fa || !fb) return false;
    std::vector<char> ba(1 << 16), bb(1 << 16);
    while (fa && fb) {
        fa.read(ba.data(), (std::streamsize)ba.size());
        fb.read(bb.data(), (std::streamsize)bb.size());
        if (fa.gcount() != fb.gcount()) return false;
        if (!std::equal(ba.begin(), ba.begin() + fa.gcount(), bb.begin())) return false;
    }
    return true;
}

// "12.3456 ms"-style durations and "1.50 KB"-style sizes: the textual forms the reference's
// benchmark log uses (reference util/format_utils.cpp:5-33), so its plot script keeps parsing.
inline std::string format_time(double seconds) {
    std::ostringstream o;
    o << std::fixed << std::setprecision(4);
    if (seconds < 1e-3) o << seconds * 1e6 << " \xce\xbcs";
    else if (seconds < 1.0) o << seconds * 1e3 << " ms";
    else o << seconds << " s";
    return o.str();
}
inline std::string format_size(size_t bytes) {
    std::ostringstream o;
    o << std::fixed << std::setprecision(2);
    if (bytes < 1024) o << bytes << " B";
    else if (bytes < 1024 * 1024) o << bytes / 1024.0 << " KB";
    else o << bytes / (1024.0 * 1024.0) << " MB";
    return o.str();
}

struct FileTestCase {
    std::string name;
    std::string input_file;
    size_t block_size;
};

// One case per (file of `dir`, block size), files in sorted order
// (reference util/test_utils.cpp:5-36).
inline std::vector<FileTestCase> generate_file_test_cases(const std::string& dir, const std::vector<size_t>& block_sizes,
                                                          bool verbose_names) {
    std::vector<FileTestCase> cases;
    for (const std::string& fn : list_files_in_directory(dir))
        for (size_t bs : block_sizes) {
            std::string name = fn;
            if (verbose_names)
                name += " (" + (bs >= 1024 ? std::to_string(bs / 1024) + "KB" : std::to_string(bs)) + " blocks)";
            cases.push_back({name, dir + "/" + fn, bs});
        }
    return cases;
}

}  // namespace bwt_host

#endif  // BWT_HOST_UTIL_HPP
