// performance [dataset_dir] [trials] -- file-to-file round-trip benchmark of the GPU path.
//
// Same command line, block-size sweep, timing method and log layout as the reference's
// benchmark (reference tests/performance.cpp:261-368; the "Test:", "Block Size:", "Mean:",
// "Throughput:" lines and the aggregate table are what reference util/plot_performance.py
// parses), so logs of the two implementations can be compared and plotted side by side.
// Times include file I/O, the delimiter search and host<->device copies, exactly like the
// reference's include its own I/O; throughput = input MiB / mean(forward + inverse).
//
// Beyond the reference's columns every test also reports the transform with the data resident
// in device memory (bwtgpu_time_resident: HIP-event time of the forward and inverse passes, GB/s
// and the share of the 8 TB/s HBM roofline that the round trip's algorithmic bytes -- read N,
// write N + blocks, and back -- amount to), the GPUs in use and the host's cores.  Those lines
// start with "GPU", which reference util/plot_performance.py:52-110 skips, and a second aggregate
// table follows the reference's own.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <iomanip>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "bwt.hpp"
#include "bwtgpu.h"
#include "host_util.hpp"
#include "inverse_bwt.hpp"
#include "slab_runner.hpp"

using namespace bwt_host;

struct Series {
    std::vector<double> v;
    double mean() const {
        double s = 0;
        for (double x : v) s += x;
        return v.empty() ? 0.0 : s / (double)v.size();
    }
    double stddev() const {
        if (v.size() < 2) return 0.0;
        const double m = mean();
        double s = 0;
        for (double x : v) s += (x - m) * (x - m);
        return std::sqrt(s / (double)(v.size() - 1));
    }
    double min() const { return *std::min_element(v.begin(), v.end()); }
    double max() const { return *std::max_element(v.begin(), v.end()); }
};

struct Result {
    Series forward, inverse, total;
    size_t input_size = 0, output_size = 0;
    double gpu_forward_ms = 0, gpu_inverse_ms = 0;  // data resident in device memory (0: not measured)
    size_t blocks = 0;
};

static const double kHbmPeak = 8.0e12;  // bytes per second, MI355X HBM3E
static int g_devices_bound = 0, g_devices_used = 0;

// Round trip with the file's bytes resident in device memory: mean HIP-event time of a forward and
// of an inverse pass (files above 1 GiB are skipped: the benchmark corpus is far below that).
static void measure_resident(const std::string& input, size_t block_size, int trials, Result& r) {
    if (r.input_size == 0 || r.input_size > ((size_t)1 << 30)) return;
    std::vector<uint8_t> data(r.input_size);
    FILE* f = std::fopen(input.c_str(), "rb");
    if (!f) return;
    const size_t got = std::fread(data.data(), 1, data.size(), f);
    std::fclose(f);
    if (got != data.size()) return;
    uint32_t mask[8];
    const int delim = bwtgpu_find_unused_byte(data.data(), data.size(), mask);
    if (delim < 0) return;
    double fwd = 0, inv = 0;
    if (bwtgpu_time_resident(0, data.data(), data.size(), block_size, (uint8_t)delim, trials, &fwd, &inv) != BWTGPU_OK) {
        std::cerr << "Device-resident timing failed: " << bwtgpu_last_error() << std::endl;
        return;
    }
    r.gpu_forward_ms = fwd;
    r.gpu_inverse_ms = inv;
    r.blocks = bwtgpu_num_blocks(data.size(), std::min(block_size, data.size()));
}

static double roundtrip_algorithmic_bytes(size_t n, size_t blocks) { return 4.0 * (double)n + 2.0 * (double)blocks; }

static double seconds_of(const std::chrono::high_resolution_clock::time_point& a,
                         const std::chrono::high_resolution_clock::time_point& b) {
    return std::chrono::duration<double>(b - a).count();
}

static Result measure(const std::string& input, size_t block_size, int trials) {
    Result r;
    r.input_size = get_file_size(input);
    const std::string fwd = "build/tmp/perf_temp.bwt", back = "build/tmp/perf_recovered.txt";
    for (int t = 0; t < trials; ++t) {
        std::remove(fwd.c_str());
        std::remove(back.c_str());
        const auto t0 = std::chrono::high_resolution_clock::now();
        const int frc = bwt_forward_process_file(input.c_str(), fwd.c_str(), block_size);
        const auto t1 = std::chrono::high_resolution_clock::now();
        if (frc != 0) {
            std::cerr << "Failed to process forward BWT" << std::endl;
            return r;
        }
        if (t == 0) r.output_size = get_file_size(fwd);
        const auto t2 = std::chrono::high_resolution_clock::now();
        const int irc = bwt_inverse_process_file(fwd.c_str(), back.c_str(), block_size);
        const auto t3 = std::chrono::high_resolution_clock::now();
        if (irc != 0) {
            std::cerr << "Failed to process inverse BWT" << std::endl;
            return r;
        }
        if (t == 0 && !files_are_identical(input, back)) {
            std::cerr << "Round trip did not reproduce " << input << std::endl;
            return r;
        }
        r.forward.v.push_back(seconds_of(t0, t1));
        r.inverse.v.push_back(seconds_of(t2, t3));
        r.total.v.push_back(seconds_of(t0, t1) + seconds_of(t2, t3));
    }
    std::remove(fwd.c_str());
    std::remove(back.c_str());
    measure_resident(input, block_size, trials, r);
    return r;
}

static void print_series(const char* title, const Series& s) {
    std::cout << title << std::endl;
    std::cout << "  Mean:   " << format_time(s.mean());
    if (s.v.size() > 1) std::cout << " \xc2\xb1 " << format_time(s.stddev());
    std::cout << std::endl;
    std::cout << "  Min:    " << format_time(s.min()) << std::endl;
    std::cout << "  Max:    " << format_time(s.max()) << std::endl;
}

static void print_result(const std::string& name, const Result& r, size_t block_size) {
    const std::string bar(70, '='), rule(70, '-');
    std::cout << "\n" << bar << "\nTest: " << name << "\nBlock Size: " << format_size(block_size) << std::endl;
    if (r.total.v.empty()) {
        std::cout << "Status: FAILED - No successful trials recorded.\n" << bar << std::endl;
        return;
    }
    std::cout << "Input Size: " << format_size(r.input_size) << "\nOutput Size: " << format_size(r.output_size)
              << std::endl;
    std::cout << "Compression Ratio: " << std::fixed << std::setprecision(4)
              << (r.input_size ? (double)r.output_size / (double)r.input_size : 0.0) << std::endl;
    std::cout << rule << "\nTrials: " << r.forward.v.size() << "\n" << rule << std::endl;
    print_series("Forward BWT:", r.forward);
    std::cout << std::endl;
    print_series("Inverse BWT:", r.inverse);
    std::cout << std::endl;
    print_series("Total Roundtrip:", r.total);
    std::cout << "\nThroughput: " << std::fixed << std::setprecision(2)
              << ((double)r.input_size / (1024.0 * 1024.0)) / r.total.mean() << " MB/s" << std::endl;
    if (r.gpu_forward_ms > 0 && r.gpu_inverse_ms > 0) {
        const double sec = (r.gpu_forward_ms + r.gpu_inverse_ms) * 1e-3;
        std::cout << rule << "\nGPU Forward (data resident in HBM): " << format_time(r.gpu_forward_ms * 1e-3)
                  << "\nGPU Inverse (data resident in HBM): " << format_time(r.gpu_inverse_ms * 1e-3) << std::endl;
        std::cout << "GPU Roundtrip Rate: " << std::fixed << std::setprecision(4) << (double)r.input_size / sec / 1e9
                  << " GB/s" << std::endl;
        std::cout << "GPU HBM Roofline: " << std::fixed << std::setprecision(4)
                  << 100.0 * roundtrip_algorithmic_bytes(r.input_size, r.blocks) / sec / kHbmPeak << " % of 8 TB/s"
                  << std::endl;
    }
    std::cout << "GPUs Used: " << g_devices_used << " of " << g_devices_bound
              << "\nGPU Host Cores: " << std::thread::hardware_concurrency() << std::endl;
    std::cout << bar << std::endl;
}

static void print_aggregate(const std::map<size_t, std::vector<Result>>& by_block) {
    const std::string bar(70, '=');
    std::cout << "\n" << bar << "\nAggregate Statistics by Block Size (Entire Dataset)\n" << bar << std::endl;
    std::cout << std::left << std::setw(12) << "Block Size" << std::setw(15) << "Total Size" << std::setw(15)
              << "Total Time" << std::setw(18) << "Throughput" << std::setw(12) << "Comp. Ratio" << std::endl;
    std::cout << std::string(72, '-') << std::endl;
    for (const auto& kv : by_block) {
        size_t in = 0, out = 0;
        double time = 0;
        for (const Result& r : kv.second) {
            if (r.total.v.empty()) continue;
            in += r.input_size;
            out += r.output_size;
            time += r.total.mean();
        }
        std::ostringstream tp;
        tp << std::fixed << std::setprecision(2) << (time > 0 ? ((double)in / (1024.0 * 1024.0)) / time : 0.0) << " MB/s";
        std::cout << std::left << std::setw(12) << format_size(kv.first) << std::setw(15) << format_size(in)
                  << std::setw(15) << format_time(time) << std::setw(18) << tp.str() << std::setw(12) << std::fixed
                  << std::setprecision(4) << (in ? (double)out / (double)in : 0.0) << std::endl;
    }
    std::cout << bar << std::endl;
    // the same dataset with the data resident in device memory
    std::cout << "\n" << bar << "\nGPU Aggregate by Block Size (data resident in HBM, " << g_devices_used << " of "
              << g_devices_bound << " GPUs, " << std::thread::hardware_concurrency() << " host cores)\n" << bar << std::endl;
    std::cout << std::left << std::setw(12) << "Block Size" << std::setw(15) << "Forward" << std::setw(15) << "Inverse"
              << std::setw(18) << "Roundtrip" << std::setw(14) << "% HBM roofline" << std::endl;
    std::cout << std::string(74, '-') << std::endl;
    for (const auto& kv : by_block) {
        size_t in = 0, blocks = 0;
        double f = 0, v = 0;
        for (const Result& r : kv.second) {
            if (r.gpu_forward_ms <= 0) continue;
            in += r.input_size;
            blocks += r.blocks;
            f += r.gpu_forward_ms * 1e-3;
            v += r.gpu_inverse_ms * 1e-3;
        }
        std::ostringstream tp, pc;
        tp << std::fixed << std::setprecision(4) << (f + v > 0 ? (double)in / (f + v) / 1e9 : 0.0) << " GB/s";
        pc << std::fixed << std::setprecision(4) << (f + v > 0 ? 100.0 * roundtrip_algorithmic_bytes(in, blocks) / (f + v) / kHbmPeak : 0.0);
        std::cout << std::left << std::setw(12) << format_size(kv.first) << std::setw(15) << format_time(f) << std::setw(15)
                  << format_time(v) << std::setw(18) << tp.str() << std::setw(14) << pc.str() << std::endl;
    }
    std::cout << bar << std::endl;
}

int main(int argc, char* argv[]) {
    const std::string bar(70, '=');
    std::cout << "\n" << bar << "\nBWT Performance Benchmark\n" << bar << std::endl;
    std::string dir = "data/canterbury_corpus";
    int trials = 5;
    if (argc > 1) dir = argv[1];
    if (argc > 2) {
        trials = std::atoi(argv[2]);
        if (trials < 1) {
            std::cerr << "Invalid number of trials. Using default: 5" << std::endl;
            trials = 5;
        }
    }
    const int devices = bwtgpu_init(0);
    if (devices <= 0) {
        std::cerr << "Error: " << bwtgpu_last_error() << std::endl;
        return 1;
    }
    g_devices_bound = devices;
    g_devices_used = bwt_host::device_budget();
    std::cout << "Dataset Directory: " << dir << "\nNumber of trials per test: " << trials << std::endl;
    std::cout << "Number of worker threads: 1 host thread driving " << bwt_host::device_budget() << " of " << devices
              << " GPU(s)" << std::endl;
    if (!create_directory("build/tmp")) {
        std::cerr << "Error: Failed to create build/tmp directory" << std::endl;
        return 1;
    }
    const std::vector<size_t> block_sizes = {512, 1024, 4096, 16384, 65536, 262144, 1048576};
    if (!directory_exists(dir)) {
        std::cerr << "Error: Data directory not found: " << dir << std::endl;
        return 1;
    }
    std::cout << "Scanning directory: " << dir << std::endl;
    const auto cases = generate_file_test_cases(dir, block_sizes, false);
    if (cases.empty()) {
        std::cerr << "Error: No test cases generated. Check if data directory contains files." << std::endl;
        return 1;
    }
    std::cout << "Found " << cases.size() / block_sizes.size() << " files\nTesting " << block_sizes.size()
              << " block sizes\nTotal test cases: " << cases.size() << "\n" << bar << std::endl;
    std::map<size_t, std::vector<Result>> by_block;
    size_t done = 0;
    for (const auto& c : cases) {
        ++done;
        std::cout << "\n[" << done << "/" << cases.size() << "] Running: " << c.name
                  << " (block size: " << format_size(c.block_size) << ")" << std::endl;
        const Result r = measure(c.input_file, c.block_size, trials);
        print_result(c.name, r, c.block_size);
        by_block[c.block_size].push_back(r);
    }
    std::cout << "\n" << bar << "\nPerformance Benchmark Complete!\nTotal tests completed: " << done << std::endl;
    print_aggregate(by_block);
    std::cout << bar << std::endl;
    bwtgpu_shutdown();
    return 0;
}
