// CPU copy rates out of and into pinned host memory by allocation flag.
// build: hipcc -O2 -pthread tools/microbench/pinned_read.hip -o build/pinned_read
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <typename F>
static double par(int nt, size_t n, F f) {
    const double t0 = now();
    std::vector<std::thread> th;
    const size_t per = ((n + nt - 1) / nt + 4095) & ~(size_t)4095;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([=] {
            const size_t lo = std::min(n, t * per), hi = std::min(n, lo + per);
            if (hi > lo) f(lo, hi - lo);
        });
    for (auto& t : th) t.join();
    return now() - t0;
}

static void shm_fill(const char* when, const char* src, size_t n) {
    const char* path = "/dev/shm/bwt_pinned_probe";
    unlink(path);
    const int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0600);
    if (fd < 0 || ftruncate(fd, n) != 0) return;
    char* m = static_cast<char*>(mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
    const double t0 = now();
    if (posix_fallocate(fd, 0, n) != 0) perror("fallocate");
    const double t1 = now();
    const double c = par(8, n, [&](size_t lo, size_t len) { memcpy(m + lo, src + lo, len); });
    const double c2 = par(8, n, [&](size_t lo, size_t len) { memcpy(m + lo, src + lo, len); });
    printf("%s: new file in /dev/shm, %zu MiB: fallocate %.3f s, first copy (8 threads) %.3f s, second copy %.3f s\n", when, n >> 20, t1 - t0, c, c2);
    munmap(m, n);
    close(fd);
    unlink(path);
}

int main() {
    const size_t n = (size_t)512 << 20;
    char* plain = static_cast<char*>(malloc(n));
    memset(plain, 1, n);
    shm_fill("before the first HIP call", plain, n);
    (void)hipFree(nullptr);
    shm_fill("after HIP initialisation", plain, n);
    struct { const char* name; unsigned flags; } kinds[] = {
        {"default", hipHostMallocDefault}, {"non-coherent", hipHostMallocNonCoherent}, {"coherent", hipHostMallocCoherent},
        {"numa-user", hipHostMallocNumaUser}, {"registered malloc", 0xFFFFFFFFu}};
    for (auto& k : kinds) {
        char* p = nullptr;
        const double t0 = now();
        if (k.flags == 0xFFFFFFFFu) {
            p = static_cast<char*>(aligned_alloc(4096, n));
            memset(p, 0, n);
            if (hipHostRegister(p, n, hipHostRegisterDefault) != hipSuccess) { printf("register failed\n"); continue; }
        } else if (hipHostMalloc((void**)&p, n, k.flags) != hipSuccess) {
            printf("%s: alloc failed\n", k.name);
            (void)hipGetLastError();
            continue;
        }
        const double ta = now() - t0;
        {   // as a result slab lives: the device writes it first, then the CPU reads it
            char* d = nullptr;
            if (hipMalloc((void**)&d, n) == hipSuccess) {
                (void)hipMemset(d, 3, n);
                (void)hipDeviceSynchronize();
                double t1 = now();
                (void)hipMemcpy(p, d, n, hipMemcpyDeviceToHost);
                const double dn = now() - t1;
                const double r1 = par(8, n, [&](size_t lo, size_t len) { memcpy(plain + lo, p + lo, len); });
                const double r2 = par(8, n, [&](size_t lo, size_t len) { memcpy(plain + lo, p + lo, len); });
                t1 = now();
                (void)hipMemcpy(p, d, n, hipMemcpyDeviceToHost);
                const double dn2 = now() - t1;
                printf("%-18s untouched: download %.3f s, first CPU read (8 threads) %.3f s, second %.3f s, second download %.3f s\n", k.name, dn, r1, r2, dn2);
                (void)hipFree(d);
            }
        }
        for (int nt : {1, 8}) {
            const double w = par(nt, n, [&](size_t lo, size_t len) { memcpy(p + lo, plain + lo, len); });
            const double r = par(nt, n, [&](size_t lo, size_t len) { memcpy(plain + lo, p + lo, len); });
            printf("%-18s alloc %.3f s; %d threads: write into it %.2f GB/s, read out of it %.2f GB/s\n", k.name, ta, nt, n / w / 1e9, n / r / 1e9);
        }
        // device <-> host copy rate
        char* d = nullptr;
        if (hipMalloc((void**)&d, n) == hipSuccess) {
            (void)hipMemcpy(d, p, n, hipMemcpyHostToDevice);
            double t1 = now();
            (void)hipMemcpy(d, p, n, hipMemcpyHostToDevice);
            const double up = now() - t1;
            t1 = now();
            (void)hipMemcpy(p, d, n, hipMemcpyDeviceToHost);
            printf("%-18s upload %.1f GB/s, download %.1f GB/s\n", k.name, n / up / 1e9, n / (now() - t1) / 1e9);
            (void)hipFree(d);
        }
        if (k.flags == 0xFFFFFFFFu) (void)hipHostUnregister(p); else (void)hipHostFree(p);
    }
    shm_fill("after the pinned allocations", plain, n);
    return 0;
}
