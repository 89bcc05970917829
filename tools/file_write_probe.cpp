// measurement aid: how fast can ONE tmpfs file be filled?  modes: 0 one thread write(), 1 N threads pwrite at their offsets,
// 2 N threads memcpy into a shared mapping, 3 the same after posix_fallocate, 4 pwrite after posix_fallocate, 5 mapping with
// MADV_HUGEPAGE, 6 N threads each their OWN file (what the machine can do).  usage: file_write_probe <path> <mode> <threads> <MiB>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
using clk = std::chrono::steady_clock;
static double now() { return std::chrono::duration<double>(clk::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const char* path = argv[1];
    const int mode = atoi(argv[2]), nt = atoi(argv[3]);
    const size_t total = (size_t)atol(argv[4]) << 20, piece = 8u << 20;
    std::vector<char> src(piece);
    for (size_t i = 0; i < piece; i++) src[i] = "ACGT\n"[i % 5];
    unlink(path);
    int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    double t0 = now();
    if (mode == 0) { // one thread, write()
        for (size_t o = 0; o < total; o += piece) if (write(fd, src.data(), piece) != (ssize_t)piece) return 1;
    } else if (mode == 1 || mode == 4) { // nt threads, pwrite at own offsets (4: after fallocate)
        if (mode == 4) { if (posix_fallocate(fd, 0, total)) return 2; printf("fallocate %.2f s\n", now() - t0); }
        else if (ftruncate(fd, total)) return 2;
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++) th.emplace_back([&, t] {
            for (size_t o = (size_t)t * piece; o < total; o += (size_t)nt * piece) if (pwrite(fd, src.data(), piece, o) != (ssize_t)piece) abort();
        });
        for (auto& x : th) x.join();
    } else if (mode == 2 || mode == 3 || mode == 5) { // mmap + memcpy (3: after fallocate, 5: MADV_HUGEPAGE)
        if (mode == 3) { if (posix_fallocate(fd, 0, total)) return 2; printf("fallocate %.2f s\n", now() - t0); }
        else if (ftruncate(fd, total)) return 2;
        char* m = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) return 3;
        if (mode == 5) madvise(m, total, MADV_HUGEPAGE);
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++) th.emplace_back([&, t] {
            for (size_t o = (size_t)t * piece; o < total; o += (size_t)nt * piece) memcpy(m + o, src.data(), piece);
        });
        for (auto& x : th) x.join();
        munmap(m, total);
    } else if (mode == 6) { // nt threads, each its OWN file (reference point)
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++) th.emplace_back([&, t] {
            char p[256]; snprintf(p, sizeof p, "%s.%d", path, t);
            int f = open(p, O_RDWR | O_CREAT | O_TRUNC, 0644);
            for (size_t o = (size_t)t * piece; o < total; o += (size_t)nt * piece) if (write(f, src.data(), piece) != (ssize_t)piece) abort();
            close(f); unlink(p);
        });
        for (auto& x : th) x.join();
    }
    double dt = now() - t0;
    close(fd);
    printf("mode %d threads %d: %.2f s, %.2f GB/s\n", mode, nt, dt, total / dt / 1e9);
    unlink(path);
    return 0;
}
