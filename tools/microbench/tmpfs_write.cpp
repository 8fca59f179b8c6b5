// How fast can a new file in /dev/shm be filled?  Prints seconds for 1 GiB by method and thread count.
// build: g++ -O2 -pthread tools/microbench/tmpfs_write.cpp -o build/tmpfs_write
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <typename F>
static void par(int nt, size_t n, F f) {
    std::vector<std::thread> th;
    const size_t per = ((n + nt - 1) / nt + 4095) & ~(size_t)4095;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([=] {
            const size_t lo = std::min(n, t * per), hi = std::min(n, lo + per);
            if (hi > lo) f(lo, hi - lo);
        });
    for (auto& t : th) t.join();
}

int main(int argc, char** argv) {
    const size_t n = (size_t)(argc > 1 ? atoi(argv[1]) : 1024) << 20;
    const char* path = "/dev/shm/bwt_tmpfs_write_probe";
    char* src = static_cast<char*>(malloc(n));
    memset(src, 7, n);
    for (int nt : {1, 4, 8, 16}) {
        for (int method = 0; method < 4; ++method) {
            unlink(path);
            const int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0600);
            if (fd < 0) return 1;
            const double t0 = now();
            const char* name = "";
            if (method == 0) {
                name = "pwrite";
                par(nt, n, [&](size_t lo, size_t len) {
                    size_t did = 0;
                    while (did < len) {
                        const ssize_t r = pwrite(fd, src + lo + did, len - did, lo + did);
                        if (r <= 0) break;
                        did += r;
                    }
                });
            } else {
                if (ftruncate(fd, n) != 0) return 1;
                char* m = static_cast<char*>(mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
                if (m == MAP_FAILED) return 1;
                if (method == 1) {
                    name = "mmap+memcpy";
                    par(nt, n, [&](size_t lo, size_t len) { memcpy(m + lo, src + lo, len); });
                } else if (method == 2) {
                    name = "populate only";
                    par(nt, n, [&](size_t lo, size_t len) {
                        for (size_t at = 0; at < len; at += (size_t)16 << 20)
                            if (madvise(m + lo + at, std::min((size_t)16 << 20, len - at), MADV_POPULATE_WRITE)) perror("madvise");
                    });
                } else {
                    name = "fallocate then memcpy";
                    if (posix_fallocate(fd, 0, n) != 0) perror("fallocate");
                    const double t1 = now();
                    par(nt, n, [&](size_t lo, size_t len) { memcpy(m + lo, src + lo, len); });
                    printf("  (fallocate %.3f s, copy %.3f s)\n", t1 - t0, now() - t1);
                }
                munmap(m, n);
            }
            printf("%2d threads %-22s %.3f s  %.2f GB/s\n", nt, name, now() - t0, n / (now() - t0) / 1e9);
            close(fd);
        }
    }
    unlink(path);
    return 0;
}
