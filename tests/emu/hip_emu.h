/*
 * hip_emu.h -- TEST INFRASTRUCTURE ONLY.
 *
 * A tiny lock-step emulator of the HIP execution model, just enough to compile
 * fastplong_amd/csrc/kernels.h for the host and run its kernels on the CPU in the
 * `-m "not gpu"` test suite (this container has no GPU; every gpurun round trip costs GPU
 * minutes).  One OS thread per WAVE; its 64 work-items are fibers (own stacks, a dozen instructions of context switch) that the
 * thread resumes in lane order: a cross-lane primitive makes a lane yield, so a lane runs again only after every other live lane
 * of its wave has reached the same rendezvous -- a wave barrier costs 128 user-level switches instead of 64 futex sleeps and
 * wake-ups (the first form of this emulator: one OS thread per work-item on std::barrier; the suite spent four fifths of its
 * time in the kernel's scheduler).  __syncthreads is a real barrier between the block's wave threads, entered once all live
 * lanes of a wave stand at it.  Blocks run one after another so `static` storage can stand in for __shared__.  It is never
 * linked into the product library: the shipped libfastplong_amd.so is built by hipcc only and has no CPU path.
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
#include <sys/mman.h>
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

/* switch stacks: the callee-saved registers of the running context go on its stack, its stack pointer to *save_sp, and the context
   whose stack pointer is new_sp goes on (x86-64 System V; a fresh fiber's stack is laid out in emu::fiber_init to match) */
extern "C" void emu_switch(void** save_sp, void* new_sp);
asm(R"(
.text
.weak emu_switch
.type emu_switch, @function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_switch, .-emu_switch
)");

namespace emu {
struct Wave {
    uint64_t slot[64];
    bool active[64];
    explicit Wave() {
        for (int i = 0; i < 64; i++) active[i] = true, slot[i] = 0;
    }
};
struct Block {
    std::barrier<> bar; /* between the block's wave threads */
    std::vector<std::unique_ptr<Wave>> waves;
    explicit Block(int nthreads) : bar(nthreads / 64) {
        for (int i = 0; i < nthreads / 64; i++) waves.emplace_back(new Wave());
    }
};
constexpr size_t FIBER_STACK = 1u << 20; /* (address space: a page is only backed once a lane has touched it) */
enum Wait { WAIT_NONE = 0, WAIT_WAVE = 1, WAIT_BLOCK = 2, WAIT_DONE = 3 };
struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    int wait = WAIT_NONE;
};
inline thread_local Block* t_block = nullptr;
inline thread_local Wave* t_wave = nullptr;
inline thread_local int t_lane = 0;
inline thread_local Fiber* t_fiber = nullptr;     /* the lane that is running */
inline thread_local void* t_sched_sp = nullptr;   /* the wave thread's own context while a lane runs */
inline thread_local void (*t_entry)(void*) = nullptr; /* what a fresh lane runs */
inline thread_local void* t_entry_arg = nullptr;

/* a lane stops here until the wave thread resumes it */
inline void fiber_yield(int why) {
    Fiber* f = t_fiber;
    f->wait = why;
    emu_switch(&f->sp, t_sched_sp);
}
inline void fiber_main() {
    t_entry(t_entry_arg);
    fiber_yield(WAIT_DONE);
    abort(); /* (a finished lane is never resumed) */
}
inline void fiber_init(Fiber* f) {
    /* the stack as emu_switch leaves one: six registers, then the address `ret` goes to; above it one slot, so that
       fiber_main starts with the alignment a call would have given it */
    void** top = (void**)(f->stack + FIBER_STACK);
    top[-1] = nullptr;
    top[-2] = (void*)&fiber_main;
    for (int i = 3; i <= 8; i++) top[-i] = nullptr;
    f->sp = (void*)(top - 8);
    f->wait = WAIT_NONE;
}
}  // namespace emu

inline thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

inline void __syncthreads() { emu::fiber_yield(emu::WAIT_BLOCK); }
inline void emu_wave_barrier() { emu::fiber_yield(emu::WAIT_WAVE); }
using std::max;
using std::min;

/* exchange: every active lane publishes v, then reads what it needs */
template <class F>
inline auto emu_xchg(uint64_t v, F&& reader) {
    emu::Wave* w = emu::t_wave;
    w->slot[emu::t_lane] = v;
    emu_wave_barrier();
    auto r = reader(w);
    emu_wave_barrier();
    return r;
}

inline void emu_gather_u64(unsigned long long v, unsigned long long* out) {
    emu::Wave* w = emu::t_wave;
    w->slot[emu::t_lane] = v;
    emu_wave_barrier();
    for (int i = 0; i < 64; i++) out[i] = w->slot[i];
    emu_wave_barrier();
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

/* One OS thread per wave, created once per launch; the wave threads walk through the blocks together.  A wave thread resumes its
 * live lanes in lane order, pass after pass: every lane runs up to its next rendezvous (or to its end) per pass, so when lane i
 * is resumed every other live lane has been through the rendezvous lane i waited at.  When the lanes stand at __syncthreads the
 * thread joins the block's barrier once for all of them; a wave whose lanes have all returned leaves that barrier for good. */
template <class K, class... A>
inline void emu_launch(K kernel, dim3 grid, dim3 block, A... args) {
    const int nthreads = (int)block.x;
    if (nthreads % 64 != 0) throw "emu: blockDim.x must be a multiple of 64";
    const int nwaves = nthreads / 64;
    const unsigned nblocks = grid.x * grid.y;
    if (nblocks == 0) return;
    std::unique_ptr<emu::Block> blk;
    std::barrier<> turn(nwaves); /* all wave threads, between blocks */
    auto run_lane = [&]() { kernel(args...); };
    using Run = decltype(run_lane);
    std::vector<std::thread> th;
    th.reserve(nwaves);
    for (int w = 0; w < nwaves; w++) {
        th.emplace_back([&, w]() {
            char* stacks = (char*)mmap(nullptr, 64 * emu::FIBER_STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (stacks == (char*)MAP_FAILED) {
                fprintf(stderr, "emu: no address space for the lanes' stacks\n");
                abort();
            }
            emu::Fiber fib[64];
            for (int l = 0; l < 64; l++) fib[l].stack = stacks + (size_t)l * emu::FIBER_STACK;
            emu::t_entry = [](void* p) { (*(Run*)p)(); };
            emu::t_entry_arg = (void*)&run_lane;
            for (unsigned b = 0; b < nblocks; b++) {
                if (w == 0) blk.reset(new emu::Block(nthreads));
                turn.arrive_and_wait(); /* the block object is ready */
                blockIdx = dim3(b % grid.x, b / grid.x);
                blockDim = block;
                gridDim = grid;
                emu::t_block = blk.get();
                emu::t_wave = blk->waves[w].get();
                for (int l = 0; l < 64; l++) emu::fiber_init(&fib[l]);
                int live = 64;
                while (live) {
                    int at_block = 0;
                    for (int l = 0; l < 64; l++) {
                        emu::Fiber* f = &fib[l];
                        if (f->wait == emu::WAIT_DONE) continue;
                        threadIdx = dim3((unsigned)(w * 64 + l));
                        emu::t_lane = l;
                        emu::t_fiber = f;
                        emu_switch(&emu::t_sched_sp, f->sp);
                        if (f->wait == emu::WAIT_DONE) {
                            emu::t_wave->active[l] = false;
                            live--;
                        } else if (f->wait == emu::WAIT_BLOCK)
                            at_block++;
                    }
                    if (at_block) {
                        if (at_block != live) { /* (a kernel whose lanes disagree about where they wait would hang a GPU too) */
                            fprintf(stderr, "emu: %d of %d live lanes of a wave stand at __syncthreads\n", at_block, live);
                            abort();
                        }
                        blk->bar.arrive_and_wait();
                    }
                }
                blk->bar.arrive_and_drop();
                turn.arrive_and_wait(); /* everyone has left the block before it is replaced */
            }
            munmap(stacks, 64 * emu::FIBER_STACK);
        });
    }
    for (auto& x : th) x.join();
}

#endif
