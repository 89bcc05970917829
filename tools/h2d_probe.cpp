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
    /* what the CALL costs the host thread, by size (page-locked source, idle stream / a stream with copies queued) */
    for (size_t mb : {8, 32, 128, 512}) {
        const size_t n = mb << 20;
        if (n > blk * n_blk) continue;
        char* big;
        CK(hipMalloc((void**)&big, n));
        CK(hipDeviceSynchronize());
        double first = 0, rest = 0;
        for (int i = 0; i < 6; i++) {
            auto a0 = std::chrono::steady_clock::now();
            CK(hipMemcpyAsync(big, arena, n, hipMemcpyHostToDevice, st));
            const double c = std::chrono::duration<double>(std::chrono::steady_clock::now() - a0).count();
            if (i == 0) first = c; else rest += c;
        }
        auto a0 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(st));
        const double drain = std::chrono::duration<double>(std::chrono::steady_clock::now() - a0).count();
        printf("hipMemcpyAsync of %4zu MB: the call takes %7.1f us on an idle stream, %7.1f us behind queued copies (6 copies drain in %.2f ms more)\n", mb, first * 1e6, rest / 5 * 1e6, drain * 1e3);
        CK(hipFree(big));
    }
    /* the same uploads with what a pipeline needs around them: an event behind every upload that a second stream waits for (with a
       small kernel behind the wait), on one copy stream and on two alternating ones */
    hipStream_t st2, sk;
    CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
    hipEvent_t ev[8];
    for (auto& evt : ev) CK(hipEventCreateWithFlags(&evt, hipEventDisableTiming));
    for (int mode = 0; mode < 3; mode++) {
        const int reps = 200;
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; i++) {
            hipStream_t s = mode == 2 ? ((i & 1) ? st2 : st) : st;
            CK(hipMemcpyAsync(dev[i % 3], arena + (i % 8) * blk, copy, hipMemcpyHostToDevice, s));
            if (mode >= 1) {
                CK(hipEventRecord(ev[i % 8], s));
                CK(hipStreamWaitEvent(sk, ev[i % 8], 0));
                CK(hipMemsetAsync(dev[i % 3], 0, 64, sk));
            }
        }
        CK(hipDeviceSynchronize());
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%s: upload %.1f GB/s\n", mode == 0 ? "bare uploads, one stream" : mode == 1 ? "event + dependent work behind every upload, one copy stream" : "the same on two alternating copy streams", reps * (double)copy / s / 1e9);
    }
    return 0;
}
