// measurement aid: what page-locking the CLI's arena costs -- one hipHostMalloc of the whole arena against one per block, serial and
// from several threads at once; and mmap + MADV_HUGEPAGE + hipHostRegister (hipcc -O2 -o pin_probe tools/pin_probe.cpp -lpthread)
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void copy_rate(const char* what, void* h, size_t bytes) {
    void* d = nullptr;
    hipMalloc(&d, 64u << 20);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const size_t piece = 32u << 20;
    hipMemcpyAsync(d, h, piece, hipMemcpyHostToDevice, s);
    hipStreamSynchronize(s);
    const double t0 = now();
    size_t moved = 0;
    for (size_t o = 0; o + piece <= bytes; o += piece) {
        hipMemcpyAsync(d, (char*)h + o, piece, hipMemcpyHostToDevice, s);
        moved += piece;
    }
    const double tsub = now() - t0;
    hipStreamSynchronize(s);
    printf("  %s: %.1f GB/s host to device over %zu MB (submissions %.2f ms)\n", what, moved / (now() - t0) * 1e-9, moved >> 20, tsub * 1e3);
    hipStreamDestroy(s);
    hipFree(d);
}
int main() {
    (void)hipSetDevice(0);
    (void)hipFree(0);
    const size_t block = 37u << 20;
    const int n = 20;
    void* p = nullptr;
    double t0 = now();
    (void)hipHostMalloc(&p, block * n, hipHostMallocDefault);
    printf("hipHostMalloc of %d x %zu MB: %.1f ms\n", n, block >> 20, (now() - t0) * 1e3);
    copy_rate("hipHostMalloc", p, block * n);
    t0 = now();
    (void)hipHostFree(p);
    printf("  freed in %.1f ms\n", (now() - t0) * 1e3);
    for (int huge = 0; huge < 2; huge++) {
        const size_t bytes = ((block * n) + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
        t0 = now();
        void* m = mmap(nullptr, bytes + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        char* a = (char*)(((size_t)m + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1));
        if (huge) madvise(a, bytes, MADV_HUGEPAGE);
        const double tmap = now() - t0;
        t0 = now();
        // touch from four threads
        std::vector<std::thread> th;
        for (int t = 0; t < 4; t++) th.emplace_back([&, t]() { memset(a + bytes / 4 * t, 0, bytes / 4); });
        for (auto& x : th) x.join();
        const double ttouch = now() - t0;
        t0 = now();
        hipError_t e = hipHostRegister(a, bytes, hipHostRegisterDefault);
        const double treg = now() - t0;
        printf("mmap%s %zu MB: map %.1f ms, first touch (4 threads) %.1f ms, hipHostRegister %.1f ms (%s)\n", huge ? " + MADV_HUGEPAGE" : "", bytes >> 20,
               tmap * 1e3, ttouch * 1e3, treg * 1e3, hipGetErrorString(e));
        if (e == hipSuccess) {
            copy_rate("registered", a, bytes);
            t0 = now();
            (void)hipHostUnregister(a);
            printf("  unregistered in %.1f ms\n", (now() - t0) * 1e3);
        }
        munmap(m, bytes + (2u << 20));
    }
    FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
    if (f) {
        char line[256] = {0};
        if (fgets(line, sizeof line, f)) printf("transparent_hugepage/enabled: %s", line);
        fclose(f);
    }
    return 0;
}
