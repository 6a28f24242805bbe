// Development tool: how fast can a FRESH pageable buffer be filled from a warm one (the last leg
// of the device->host download), with and without transparent huge pages.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
#include <thread>
#include <vector>

static double fill(size_t n, int threads, int advise, const char* src) {
    void* p = nullptr;
    if (posix_memalign(&p, 64, n)) return 0;
    auto t0 = std::chrono::steady_clock::now();
    if (advise) {
        uintptr_t a = (reinterpret_cast<uintptr_t>(p) + 4095) & ~uintptr_t(4095);
        size_t len = (reinterpret_cast<uintptr_t>(p) + n - a) & ~size_t(4095);
        madvise(reinterpret_cast<void*>(a), len, MADV_HUGEPAGE);
    }
    std::vector<std::thread> pool;
    for (int j = 0; j < threads; ++j)
        pool.emplace_back([=] {
            size_t lo = n / threads * j, hi = (j == threads - 1) ? n : n / threads * (j + 1);
            memcpy(static_cast<char*>(p) + lo, src + lo, hi - lo);
        });
    for (auto& t : pool) t.join();
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    free(p);
    return n / s / 1e9;
}

int main() {
    const size_t n = size_t(346) << 20;
    char* src = static_cast<char*>(malloc(n));
    memset(src, 1, n);
    FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
    char line[128] = "?";
    if (f) { if (!fgets(line, sizeof line, f)) line[0] = 0; fclose(f); }
    printf("THP: %s", line);
    for (int rep = 0; rep < 2; ++rep)
        for (int threads : {8, 16, 32})
            for (int advise : {0, 1})
                printf("threads %2d madvise %d: %.1f GB/s\n", threads, advise, fill(n, threads, advise, src));
    return 0;
}
