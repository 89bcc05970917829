/*
 * pipeline.h -- the launch sequence of one batch, shared by the HIP library (fpl_hip.hip)
 * and by the test-only emulator driver (tests/emu/emu_driver.cpp) so that grid shapes and
 * kernel order are exercised on the CPU too.
 *
 *   1 k_trim_ends              reads  -> r1 window per read (+ polyX / adapter counters)
 *   2 k_cycle_stats<PRE>       original reads -> pre-filter per-cycle tables + k-mers
 *   3 k_scan                   r1 -> middle-adapter split, filter code, result records,
 *                              quality histograms / medians, passing-fragment list
 *   4 k_cycle_stats<POST>      passing fragments -> post-filter per-cycle tables + k-mers
 */
#ifndef FPL_PIPELINE_H
#define FPL_PIPELINE_H

#include "kernels.h"

namespace fpl {

#ifdef FPL_EMU
constexpr int KWAVES = 2; /* two waves per block keep the emulator's thread count low */
constexpr int SWAVES = 2;
#define FPL_LAUNCH(kernel, grid, block, stream, ...) emu_launch(kernel, grid, block, __VA_ARGS__)
typedef void* fpl_stream_t;
#else
constexpr int KWAVES = 4;
constexpr int SWAVES = 8; /* k_cycle_stats: 8 waves share one 68 KiB LDS table -> 16 waves per CU */
#define FPL_LAUNCH(kernel, grid, block, stream, ...) hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__)
typedef hipStream_t fpl_stream_t;
#endif

struct BatchArgs {
    const u8* seq;
    const u8* qual;
    const uint64_t* off;
    u32 n_reads;
    uint64_t n_bytes;
    u32 max_read_len;
    const DevConfig* cfg;
    const DevAdapter* ads;
    ReadState* state;
    fpl_read_result* results;
    uint64_t* frag_off; /* 2 * n_reads */
    u32* frag_len;      /* 2 * n_reads */
    long long* counters;
    u32 C;
    u32* work_ctr; /* zeroed before the batch */
    u32 n_cu;      /* compute units of the device (grid sizing) */
    int dbg = 0;   /* FPL_DEBUG_FLAGS ablation switches (profiling only) */
};

constexpr int N_STAGES = 4;
static const char* const STAGE_NAMES[N_STAGES] = {"k_trim_ends", "k_cycle_stats_pre", "k_scan", "k_cycle_stats_post"};

inline u32 cdiv(u32 a, u32 b) { return (a + b - 1) / b; }

/* slices of the item list for k_cycle_stats: enough blocks to fill the chip, bounded by the
 * 14-bit counter fields */
inline u32 stats_items_per_slice(u32 n_items, u32 n_tiles, u32 n_cu) {
    u32 want_blocks = 8 * n_cu;
    u32 slices = n_tiles ? cdiv(want_blocks, n_tiles) : 1;
    if (slices < 1) slices = 1;
    u32 per = cdiv(n_items ? n_items : 1, slices);
    per = (per + 63) / 64 * 64;
    if (per > CS_MAX_ITEMS_PER_SLICE / 64 * 64) per = CS_MAX_ITEMS_PER_SLICE / 64 * 64;
    if (per < 64) per = 64;
    return per;
}

template <class Mark>
inline void enqueue_batch(const BatchArgs& a, fpl_stream_t stream, Mark&& mark) {
    (void)stream;
    const dim3 block(KWAVES * 64);
    const u32 n = a.n_reads;
    mark(0);
    if (n == 0) {
        for (int i = 1; i <= N_STAGES; i++) mark(i);
        return;
    }
    /* 1: one wave per read, grid-stride; cap the grid so the LDS accumulators flush rarely */
    {
        u32 blocks = cdiv(n, KWAVES);
        const u32 cap = 16 * a.n_cu;
        if (blocks > cap) blocks = cap;
        FPL_LAUNCH((k_trim_ends<KWAVES>), dim3(blocks), block, stream, a.seq, a.qual, a.off, n, a.n_bytes, a.cfg, a.ads,
                   a.state, a.counters, a.C);
    }
    mark(1);
    const u32 n_tiles = cdiv(a.max_read_len ? a.max_read_len : 1, CS_T);
    {
        const u32 per = stats_items_per_slice(n, n_tiles, a.n_cu);
        FPL_LAUNCH((k_cycle_stats<SWAVES, true>), dim3(cdiv(n, per), n_tiles), dim3(SWAVES * 64), stream, a.seq, a.qual, a.n_bytes,
                   a.off, (const u32*)nullptr, n, per, a.counters + FPL_OFF_PRE(a.C), a.C, a.dbg);
    }
    mark(2);
    {
        u32 blocks = cdiv(n, KWAVES);
        const u32 cap = 3 * a.n_cu; /* registers / LDS admit three blocks per CU */
        if (blocks > cap) blocks = cap;
        u32 chunk = n / (blocks * KWAVES * 32u); /* ~32 dequeues per wave keep the tail short */
        if (chunk < 1) chunk = 1;
        if (chunk > 64) chunk = 64;
        FPL_LAUNCH((k_scan<KWAVES>), dim3(blocks), block, stream, a.seq, a.qual, a.off, n, a.n_bytes, a.cfg, a.ads,
                   (const ReadState*)a.state, a.results, a.frag_off, a.frag_len, a.counters, a.C, a.work_ctr, chunk);
    }
    mark(3);
    {
        const u32 items = 2 * n;
        const u32 per = stats_items_per_slice(items, n_tiles, a.n_cu);
        FPL_LAUNCH((k_cycle_stats<SWAVES, false>), dim3(cdiv(items, per), n_tiles), dim3(SWAVES * 64), stream, a.seq, a.qual,
                   a.n_bytes, (const uint64_t*)a.frag_off, (const u32*)a.frag_len, items, per,
                   a.counters + FPL_OFF_POST(a.C), a.C, a.dbg);
    }
    mark(4);
}

}  // namespace fpl
#endif
