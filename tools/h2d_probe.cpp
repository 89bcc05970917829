// measurement aid: the upload rate of 32 MB page-locked blocks (hipMemcpyAsync on one stream, three in flight) while T host threads copy
// text out of a file mapping into other page-locked blocks, as bin/fastplong_amd's chunk loaders do.
//   hipcc -O2 -o h2d_probe h2d_probe.cpp -lpthread && ./h2d_probe /dev/shm/file.fq
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main(int argc, char** argv) {
    const size_t blk = 37u << 20, n_blk = 19, copy = 32u << 20;
    char* arena;
    CK(hipHostMalloc((void**)&arena, blk * n_blk, hipHostMallocDefault));
    memset(arena, 1, blk * n_blk);
    char* dev[3];
    for (auto& d : dev) CK(hipMalloc((void**)&d, blk));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const char* map = nullptr;
    size_t map_n = 0;
    if (argc > 1) {
        int fd = open(argv[1], O_RDONLY);
        struct stat s;
        if (fd >= 0 && fstat(fd, &s) == 0) {
            map = (const char*)mmap(nullptr, s.st_size, PROT_READ, MAP_SHARED, fd, 0);
            map_n = s.st_size;
        }
    }
    for (int T : {0, 2, 4, 8, 16}) {
        std::atomic<bool> stop{false};
        std::atomic<size_t> copied{0};
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                size_t o = (size_t)t * copy, k = 0;
                while (!stop) {
                    const char* src = map ? map + (o % (map_n - copy)) : arena + ((t + 3) % n_blk) * blk;
                    memcpy(arena + ((8 + t) % n_blk) * blk, src, copy);
                    copied += copy;
                    o += (size_t)T * copy;
                    k++;
                }
            });
        const int reps = 200;
        CK(hipStreamSynchronize(st));
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; i++) CK(hipMemcpyAsync(dev[i % 3], arena + (i % 8) * blk, copy, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        stop = true;
        for (auto& x : th) x.join();
        printf("%2d host threads copying (%5.1f GB/s of memcpy): upload %.1f GB/s\n", T, copied / s / 1e9, reps * (double)copy / s / 1e9);
    }
    return 0;
}
