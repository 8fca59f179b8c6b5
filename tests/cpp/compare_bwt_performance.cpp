// compare_bwt_performance -- the two block formats of this library side by side, file to file.
//
// Stands where the reference has tests/compare_bwt_performance.cpp (its own transform against bzip2's
// block sort, five trials at 64 / 128 / 256 KiB blocks: forward, inverse, round trip, one SUMMARY line per
// phase for scripts).  bzip2's sources are not part of the reference checkout, so the second column here
// is the SAME block layout bzip2's sorter produces (sorted rotations, origin row; Bzip2BWTProcessor in
// bzip2_format.hpp) computed on the GPU: the tool then measures what the delimiter costs or saves, and
// keeps the report's shape and the SUMMARY grammar
//   SUMMARY|<file>|<phase>|<delimiter ms>|<origin-pointer ms>|<ratio>|<winner>|<faster by %>
// Every trial checks both round trips against the input.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

#include "bwt.hpp"
#include "bwtgpu.h"
#include "bzip2_format.hpp"
#include "host_util.hpp"
#include "inverse_bwt.hpp"

using namespace bwt_host;

namespace {

const int kTrials = 5;
const size_t kBlockSizes[] = {65536, 131072, 262144};  // the reference's defaults

struct Series {
    std::vector<double> ms;
    double mean() const {
        double s = 0;
        for (double v : ms) s += v;
        return ms.empty() ? 0 : s / (double)ms.size();
    }
    double stddev() const {
        if (ms.size() < 2) return 0;
        const double m = mean();
        double s = 0;
        for (double v : ms) s += (v - m) * (v - m);
        return std::sqrt(s / (double)(ms.size() - 1));
    }
    double min() const { return ms.empty() ? 0 : *std::min_element(ms.begin(), ms.end()); }
    double max() const { return ms.empty() ? 0 : *std::max_element(ms.begin(), ms.end()); }
};

template <typename F>
bool timed(Series& into, F f) {
    const auto t0 = std::chrono::steady_clock::now();
    const bool ok = f();
    into.ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return ok;
}

void print_series(const char* who, const char* what, const Series& s) {
    std::cout << who << ":\n  " << what << ":\n" << std::fixed << std::setprecision(3)
              << "    Time:      " << s.mean() << " ms";
    if (s.ms.size() > 1) std::cout << " ± " << s.stddev() << " ms";
    std::cout << "\n    Min:       " << s.min() << " ms\n    Max:       " << s.max() << " ms" << std::endl;
}

void print_ratio(const std::string& file, const char* phase, const Series& delim, const Series& origin) {
    if (origin.mean() <= 0 || delim.mean() <= 0) return;
    const double ratio = origin.mean() / delim.mean();  // above 1: the delimiter format is faster
    const bool delim_wins = ratio >= 1.0;
    const double pct = (delim_wins ? ratio - 1.0 : 1.0 / ratio - 1.0) * 100.0;
    std::cout << "  Comparison:\n    Speedup:   " << std::fixed << std::setprecision(3) << ratio << "x ("
              << (delim_wins ? "delimiter format" : "origin-pointer format") << " is " << std::setprecision(1) << pct
              << "% faster)" << std::endl;
    std::cout << "SUMMARY|" << file << "|" << phase << "|" << std::setprecision(3) << delim.mean() << "|" << origin.mean()
              << "|" << ratio << "|" << (delim_wins ? "your_bwt" : "bzip2") << "|" << std::setprecision(1) << pct << std::endl;
}

}  // namespace

int main(int argc, char** argv) {
    std::cout << "\n" << std::string(80, '=') << "\nBWT Performance Comparison: delimiter format vs origin-pointer format (bzip2's block layout), both on the GPU\n"
              << "Testing: Forward BWT, Inverse BWT, and Round Trip\n" << std::string(80, '=') << std::endl;
    if (argc < 2) {
        std::cerr << "Usage: " << argv[0] << " <input_file>\n  input_file: File to test\n  Note: Using default block sizes: 64KB, 128KB, 256KB"
                  << std::endl;
        return 1;
    }
    const std::string input = argv[1];
    if (!file_exists(input)) {
        std::cerr << "Error: Input file not found: " << input << std::endl;
        return 1;
    }
    if (!create_directory("build/tmp")) {
        std::cerr << "Error: Failed to create build/tmp directory" << std::endl;
        return 1;
    }
    if (bwtgpu_init(0) < 1) {
        std::cerr << "Error: " << bwtgpu_last_error() << std::endl;
        return 1;
    }
    const std::string name = input.substr(input.find_last_of("/\\") == std::string::npos ? 0 : input.find_last_of("/\\") + 1);
    const std::string df = "build/tmp/your_forward.bwt", di = "build/tmp/your_inverse.txt", of = "build/tmp/bzip2_forward.bwt",
                      oi = "build/tmp/bzip2_inverse.txt";
    int bad = 0;
    for (size_t bs : kBlockSizes) {
        std::cout << "\n" << std::string(80, '=') << "\nBlock Size: " << format_size(bs) << "\n" << std::string(80, '=')
                  << "\nRunning " << kTrials << " trial(s)..." << std::endl;
        Series dfw, dinv, drt, ofw, oinv, ort;
        int good = 0;
        for (int t = 0; t < kTrials; ++t) {
            Series a, b, c, d;
            const bool ok = timed(a, [&] { return bwt_forward_process_file(input.c_str(), df.c_str(), bs) == 0; }) &&
                            timed(b, [&] { return Bzip2BWTProcessor::process_file_forward(input.c_str(), of.c_str(), bs) == 0; }) &&
                            timed(c, [&] { return bwt_inverse_process_file(df.c_str(), di.c_str(), bs) == 0; }) &&
                            timed(d, [&] { return Bzip2BWTProcessor::process_file_inverse(of.c_str(), oi.c_str(), bs) == 0; }) &&
                            files_are_identical(input, di) && files_are_identical(input, oi);
            if (!ok) {
                std::cerr << "Warning: Trial " << (t + 1) << " failed for " << name << std::endl;
                continue;
            }
            ++good;
            dfw.ms.push_back(a.ms[0]);
            ofw.ms.push_back(b.ms[0]);
            dinv.ms.push_back(c.ms[0]);
            oinv.ms.push_back(d.ms[0]);
            drt.ms.push_back(a.ms[0] + c.ms[0]);
            ort.ms.push_back(b.ms[0] + d.ms[0]);
        }
        std::cout << "\n" << std::string(80, '=') << "\nTest: " << name << "\nBlock Size: " << format_size(bs) << "\nFile Size: "
                  << format_size(get_file_size(input)) << "\nTrials: " << kTrials << " (successful: " << good << ")\n"
                  << std::string(80, '-') << std::endl;
        if (!good) {
            std::cout << "ERROR: No successful trials!\n" << std::string(80, '=') << std::endl;
            ++bad;
            continue;
        }
        std::cout << "\nFORWARD BWT:" << std::endl;
        print_series("Delimiter format", "Forward", dfw);
        std::cout << "  Output:    " << format_size(get_file_size(df)) << "\n" << std::endl;
        print_series("Origin-pointer format", "Forward", ofw);
        std::cout << "  Output:    " << format_size(get_file_size(of)) << std::endl;
        print_ratio(name, "forward", dfw, ofw);
        std::cout << "\nINVERSE BWT:" << std::endl;
        print_series("Delimiter format", "Inverse", dinv);
        std::cout << std::endl;
        print_series("Origin-pointer format", "Inverse", oinv);
        print_ratio(name, "inverse", dinv, oinv);
        std::cout << "\nROUND TRIP (Forward + Inverse):" << std::endl;
        print_series("Delimiter format", "Round Trip", drt);
        std::cout << std::endl;
        print_series("Origin-pointer format", "Round Trip", ort);
        print_ratio(name, "roundtrip", drt, ort);
        const double mib = (double)get_file_size(input) / (1024.0 * 1024.0);
        std::cout << "  Throughput:\n" << std::fixed << std::setprecision(2) << "    Delimiter format:      " << mib / (drt.mean() / 1000.0)
                  << " MB/s\n    Origin-pointer format: " << mib / (ort.mean() / 1000.0) << " MB/s\n" << std::string(80, '=') << std::endl;
    }
    for (const std::string& f : {df, di, of, oi}) std::remove(f.c_str());
    bwtgpu_shutdown();
    return bad ? 1 : 0;
}
