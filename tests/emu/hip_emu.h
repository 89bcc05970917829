/*
 * hip_emu.h -- TEST INFRASTRUCTURE ONLY.
 *
 * A tiny lock-step emulator of the HIP execution model, just enough to compile
 * fastplong_amd/csrc/kernels.h for the host and run its kernels on the CPU in the
 * `-m "not gpu"` test suite (this container has no GPU; every gpurun round trip costs GPU
 * minutes).  One OS thread per work-item; a wave is 64 threads that rendezvous on a
 * std::barrier at every cross-lane primitive; blocks run one after another so `static`
 * storage can stand in for __shared__.  It is never linked into the product library: the
 * shipped libfastplong_amd.so is built by hipcc only and has no CPU path.
 */
#ifndef FPL_HIP_EMU_H
#define FPL_HIP_EMU_H

#include <algorithm>
#include <atomic>
#include <barrier>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace emu {
struct Wave {
    std::barrier<> bar;
    uint64_t slot[64];
    bool active[64];
    explicit Wave() : bar(64) {
        for (int i = 0; i < 64; i++) active[i] = true, slot[i] = 0;
    }
};
struct Block {
    std::barrier<> bar;
    std::vector<std::unique_ptr<Wave>> waves;
    explicit Block(int nthreads) : bar(nthreads) {
        for (int i = 0; i < nthreads / 64; i++) waves.emplace_back(new Wave());
    }
};
inline thread_local Block* t_block = nullptr;
inline thread_local Wave* t_wave = nullptr;
inline thread_local int t_lane = 0;
}  // namespace emu

inline thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

inline void __syncthreads() { emu::t_block->bar.arrive_and_wait(); }
inline void emu_wave_barrier() { emu::t_wave->bar.arrive_and_wait(); }
using std::max;
using std::min;

/* exchange: every active lane publishes v, then reads what it needs */
template <class F>
inline auto emu_xchg(uint64_t v, F&& reader) {
    emu::Wave* w = emu::t_wave;
    w->slot[emu::t_lane] = v;
    w->bar.arrive_and_wait();
    auto r = reader(w);
    w->bar.arrive_and_wait();
    return r;
}

inline void emu_gather_u64(unsigned long long v, unsigned long long* out) {
    emu::Wave* w = emu::t_wave;
    w->slot[emu::t_lane] = v;
    w->bar.arrive_and_wait();
    for (int i = 0; i < 64; i++) out[i] = w->slot[i];
    w->bar.arrive_and_wait();
}

inline unsigned long long __ballot(int pred) {
    return emu_xchg(pred ? 1 : 0, [](emu::Wave* w) {
        unsigned long long m = 0;
        for (int i = 0; i < 64; i++)
            if (w->active[i] && w->slot[i]) m |= 1ull << i;
        return m;
    });
}
template <class T>
inline T emu_shfl_idx(T v, int src) {
    static_assert(sizeof(T) <= 8, "");
    uint64_t raw = 0;
    std::memcpy(&raw, &v, sizeof(T));
    int lane = emu::t_lane;
    uint64_t r = emu_xchg(raw, [src, lane](emu::Wave* w) {
        int s = (src < 0 || src > 63) ? lane : src; /* out of range: own value, like ds_bpermute semantics used here */
        return w->slot[s];
    });
    T out;
    std::memcpy(&out, &r, sizeof(T));
    return out;
}
template <class T>
inline T __shfl(T v, int src, int = 64) { return emu_shfl_idx(v, src & 63); }
template <class T>
inline T __shfl_up(T v, unsigned d, int = 64) {
    int s = emu::t_lane - (int)d;
    return emu_shfl_idx(v, s < 0 ? emu::t_lane : s);
}
template <class T>
inline T __shfl_down(T v, unsigned d, int = 64) {
    int s = emu::t_lane + (int)d;
    return emu_shfl_idx(v, s > 63 ? emu::t_lane : s);
}
template <class T>
inline T __shfl_xor(T v, int m, int = 64) { return emu_shfl_idx(v, emu::t_lane ^ m); }

/* a kernel's own consistency check failed (FPL_EMU-only code in csrc/kernels.h) */
[[noreturn]] inline void emu_fail(const char* what) {
    fprintf(stderr, "emulator: consistency check failed: %s\n", what);
    abort();
}
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
inline int __ffs(unsigned x) { return __builtin_ffs((int)x); }
inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }
inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
inline unsigned __brev(unsigned x) {
    unsigned r = 0;
    for (int i = 0; i < 32; i++) r |= ((x >> i) & 1u) << (31 - i);
    return r;
}

template <class T>
inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <class T>
inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <class T>
inline T atomicMin(T* p, T v) {
    T old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < old && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
template <class T>
inline T atomicMax(T* p, T v) {
    T old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v > old && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}

/* The work-item threads are created once per launch and walk through the blocks together (a thread start per
 * work-item and BLOCK made the kernel's clone / stack mmap traffic the bulk of the test time). */
template <class K, class... A>
inline void emu_launch(K kernel, dim3 grid, dim3 block, A... args) {
    const int nthreads = (int)block.x;
    if (nthreads % 64 != 0) throw "emu: blockDim.x must be a multiple of 64";
    const unsigned nblocks = grid.x * grid.y;
    if (nblocks == 0) return;
    std::unique_ptr<emu::Block> blk;
    std::barrier<> turn(nthreads); /* all work-items, between blocks */
    std::vector<std::thread> th;
    th.reserve(nthreads);
    for (int t = 0; t < nthreads; t++) {
        th.emplace_back([&, t]() {
            for (unsigned b = 0; b < nblocks; b++) {
                if (t == 0) blk.reset(new emu::Block(nthreads));
                turn.arrive_and_wait(); /* the block object is ready */
                threadIdx = dim3(t);
                blockIdx = dim3(b % grid.x, b / grid.x);
                blockDim = block;
                gridDim = grid;
                emu::t_block = blk.get();
                emu::t_wave = blk->waves[t / 64].get();
                emu::t_lane = t % 64;
                kernel(args...);
                emu::t_wave->active[emu::t_lane] = false;
                emu::t_wave->bar.arrive_and_drop();
                blk->bar.arrive_and_drop();
                turn.arrive_and_wait(); /* everyone has left the block before it is replaced */
            }
        });
    }
    for (auto& x : th) x.join();
}

#endif
