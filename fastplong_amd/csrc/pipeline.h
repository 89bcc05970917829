/*
 * pipeline.h -- the launch sequence of one batch, shared by the HIP library (fpl_hip.hip)
 * and by the test-only emulator driver (tests/emu/emu_driver.cpp) so that grid shapes and
 * kernel order are exercised on the CPU too.
 *
 *   1 k_trim_ends              reads  -> r1 window per read (+ polyX / adapter counters)
 *   2 k_scan                   r1 -> quality histograms / medians, passFilter sums and code, Hamming argmins (ScanRec per read)
 *     k_resolve                lane = read: confirmations, middle-adapter split, result records, counters, statistics plan +
 *                              EXTRA fragment list; k_redo: the fragments of the reads that were split
 *   3 k_stats + k_stats_reduce reads -> pre- AND post-filter per-cycle tables + k-mers in one pass
 *   4 k_stats<EXTRA> + reduce  post-only fragments (split reads, far-trimmed reads)
 */
#ifndef FPL_PIPELINE_H
#define FPL_PIPELINE_H

#include <stdlib.h>
#include <string.h>

#include "kernels.h"

namespace fpl {

#ifdef FPL_EMU
constexpr int KWAVES = 2; /* two waves per block keep the emulator's thread count low */
constexpr int SWAVES = 2;
constexpr int RWAVES = 2;
#define FPL_LAUNCH(kernel, grid, block, stream, ...) emu_launch(kernel, grid, block, __VA_ARGS__)
#define FPL_MEMSET(ptr, bytes, stream) memset(ptr, 0, bytes)
typedef void* fpl_stream_t;
#define FPL_FORK_MARK(a, stream) (void)0
#define FPL_FORK(a, stream) (stream)
#define FPL_JOIN(a, stream) (void)0
#else
constexpr int KWAVES = 4;
constexpr int RWAVES = 16; /* k_resolve */
#ifndef FPL_SWAVES
#define FPL_SWAVES 16
#endif
constexpr int SWAVES = FPL_SWAVES; /* k_stats: 16 waves share one 80 KiB LDS table set -> 32 waves per CU (64 VGPRs each) */
#define FPL_LAUNCH(kernel, grid, block, stream, ...) hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__)
#define FPL_MEMSET(ptr, bytes, stream) (void)hipMemsetAsync(ptr, 0, bytes, stream)
typedef hipStream_t fpl_stream_t;
/* a side stream for work that does not depend on what the main stream does next (BatchArgs::aux; none: everything in order) */
#define FPL_FORK_MARK(a, stream) pipeline_fork_mark(a, stream)
#define FPL_FORK(a, stream) pipeline_fork(a, stream)
#define FPL_JOIN(a, stream) pipeline_join(a, stream)
#endif

/* Tuning / test hooks of the statistics passes (FPL_STATS_PER, FPL_STATS_EXTRA_PER, FPL_STATS_EXTRA_BLOCKS,
 * FPL_STATS_EXTRA_ACC, FPL_STATS_MIN_BUCKET, FPL_STATS_SORT_MIN, FPL_TRIM_BATCH_MIN, FPL_SCAN_CHUNK in the environment; 0 = built-in choice).  The library reads them ONCE, in fpl_create(); the
 * per-batch path only sees this struct. */
struct StatsTune {
    u32 per = 0, extra_per = 0, extra_blocks = 0, extra_acc = 0;
    u32 min_bucket = 0; /* FPL_STATS_MIN_BUCKET: reads that must share a front trim to get slices of their own (k_stats_sorted) */
    u32 sort_min = 0;   /* FPL_STATS_SORT_MIN: batches of fewer reads take the unsorted statistics pass */
    u32 hi_tile = 0;        /* FPL_STATS_HI_TILE: (value - 1) = the first cycle tile whose k_stats_sorted items are groups of slices */
    u32 group = 0;          /* FPL_STATS_GROUP: slices per group */
    u32 group_rows = 0;     /* FPL_STATS_GROUP_ROWS: rows a group's slab may take before it is handed over (test hook: small) */
    u32 redo_inline = 0;    /* FPL_REDO_INLINE=1: k_redo stays on the main stream (measurement aid) */
    u32 scan_chunk = 0;     /* FPL_SCAN_CHUNK: reads a k_scan wave takes per dequeue, whatever the batch size (the built-in rule gives small
                               batches chunks of one read: no wave then has a NEXT read whose head could ride in a last tile) */
    u32 trim_ahead_blocks = 0; /* FPL_TRIM_AHEAD_BLOCKS: blocks per CU of k_trim_ends_batched when it runs AHEAD of the main stream (beside the
                                  batch before): fewer than a CU holds leave most of the register file to the kernel it runs beside (0: two) */
    u32 trim_batch_min = 0; /* FPL_TRIM_BATCH_MIN: batches of fewer reads take k_trim_ends<1> (a wave per read) instead of
                               k_trim_ends_batched (64 reads per wave) */
};
inline StatsTune stats_tune_from_env() {
    auto get = [](const char* name) -> u32 {
        const char* e = getenv(name);
        return (e && atoi(e) > 0) ? (u32)atoi(e) : 0u;
    };
    StatsTune t;
    t.per = get("FPL_STATS_PER");
    t.extra_per = get("FPL_STATS_EXTRA_PER");
    t.extra_blocks = get("FPL_STATS_EXTRA_BLOCKS");
    t.extra_acc = get("FPL_STATS_EXTRA_ACC");
    t.min_bucket = get("FPL_STATS_MIN_BUCKET");
    t.sort_min = get("FPL_STATS_SORT_MIN");
    t.trim_batch_min = get("FPL_TRIM_BATCH_MIN");
    t.trim_ahead_blocks = get("FPL_TRIM_AHEAD_BLOCKS");
    t.scan_chunk = get("FPL_SCAN_CHUNK");
    t.redo_inline = get("FPL_REDO_INLINE");
    t.hi_tile = get("FPL_STATS_HI_TILE");
    t.group = get("FPL_STATS_GROUP");
    t.group_rows = get("FPL_STATS_GROUP_ROWS");
    return t;
}

constexpr int WORK_CTR_WORDS = 8;
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
    uint64_t* frag_off; /* the post-only (EXTRA) item list: 2 * n_reads entries, bm.item_cap with --break / --mask */
    u32* frag_len;
    u32* frag_cyc = nullptr; /* with --break / --mask: first cycle of the item | masked << 31 */
    BmLists bm = {nullptr, nullptr, 0, 0, 0, nullptr}; /* fragment / region lists of k_break_mask */
    bool defer = false;      /* DevConfig::defer on the host side */
    int trim_mode = 0;       /* DevConfig::trim_mode on the host side */
    int n_fasta = 0;         /* DevConfig::n_fasta on the host side */
    bool scan_short = false; /* DevConfig::scan_short on the host side */
    long long* counters;
    u32 C;
    u32* work_ctr; /* WORK_CTR_WORDS words zeroed before the batch: [0] k_scan work counter, [1] EXTRA fragment count,
                      [2] k_trim_ends_batched group counter, [3] REDO list: short items (from the back), [4] unused,
                      [5] REDO list: long items (from the front) */
    ScanRec* recs = nullptr;  /* n_reads: what k_scan leaves per read for k_resolve */
    ScanWin* wins = nullptr;  /* n_reads: ... and the bytes at the two Hamming argmins */
    RedoItem* redo = nullptr; /* n_reads: the reads a middle adapter splits (k_resolve -> k_redo) */
    u32* sort_ws = nullptr;       /* k_stats_sorted: sort_ws_words(stats_sorted_max_slices()) words */
    uint64_t* st_off = nullptr;   /* ... and (start, length, end of r1) of the reads in sorted order, n_reads each */
    u32* st_len = nullptr;
    u32* st_e = nullptr;
    u64* stats_scratch;   /* stats_scratch_slabs() x FS_SLAB u64 */
    u8* stats_flags;      /* n_tiles tile flags + one byte per slab, zeroed before each statistics pass */
    u64* extra_scratch = nullptr; /* the post-only (EXTRA) pass's own slabs and flags: it runs beside k_stats_sorted's reduce */
    u8* extra_flags = nullptr;
    fpl_stream_t trim_stream = nullptr; /* the end trims go here -- ahead of the main stream, beside the previous batch -- and the main
                                            stream waits for ev_trim_done in front of k_scan (device build only; set up by the caller) */
    void* ev_trim_done = nullptr;
    void* ev_stats_done = nullptr; /* recorded behind the statistics kernel: where the NEXT batch's end trims may start (what is left
                                      of this batch then -- the reduce, the post-only pass -- leaves most of the chip idle) */
    fpl_stream_t aux = nullptr;   /* side stream + the two events that tie it to the main stream (device build only) */
    void* ev_fork = nullptr;
    void* ev_join = nullptr;
    int sorted_form = -1; /* does this batch take the sorted statistics pass?  Decided ONCE per batch by whoever builds the arguments
                             (fpl_process_batch_device: the same answer sizes the side stream's slabs and counts the batch's forms);
                             -1: enqueue_batch asks stats_takes_sorted itself (the emulator driver) */
    u32 n_cu;      /* compute units of the device (grid sizing) */
    int dbg = 0;   /* FPL_DEBUG_FLAGS ablation switches (profiling only) */
    StatsTune tune; /* tuning / test hooks, read from the environment once by whoever builds the arguments */
};

#ifndef FPL_EMU
/* FORK_MARK: the point of the main stream the side stream's work depends on; FORK: the side stream, which waits for that point.
   What is enqueued on `stream` between the two is AHEAD of the side stream's work in the device's queues -- a kernel that fills
   the chip is resident before the side work asks for room -- but the side work does not wait for it. */
inline void pipeline_fork_mark(const BatchArgs& a, hipStream_t stream) {
    if (a.aux) (void)hipEventRecord((hipEvent_t)a.ev_fork, stream);
}
inline hipStream_t pipeline_fork(const BatchArgs& a, hipStream_t stream) {
    if (!a.aux) return stream;
    (void)hipStreamWaitEvent(a.aux, (hipEvent_t)a.ev_fork, 0);
    return a.aux;
}
/* `stream` goes on when the side stream's work is done */
inline void pipeline_join(const BatchArgs& a, hipStream_t stream) {
    if (!a.aux) return;
    (void)hipEventRecord((hipEvent_t)a.ev_join, a.aux);
    (void)hipStreamWaitEvent(stream, (hipEvent_t)a.ev_join, 0);
}
#endif

constexpr int N_STAGES = 7;
static const char* const STAGE_NAMES[N_STAGES] = {"k_trim_ends", "k_scan", "k_resolve", "k_stats_prep", "k_stats", "k_stats_reduce",
                                                  "k_stats_extra"};
/* k_resolve = k_resolve + k_redo (+ k_break_mask with --break / --mask; with the sorted statistics pass k_redo runs on the side stream,
   beside stage k_stats_prep, and its time shows there); k_stats_prep = the bucket kernels of the sorted pass;
   k_stats = k_stats_sorted (or the unsorted k_stats) alone; k_stats_extra = the post-only pass and its reduce -- when that pass
   runs on the context's side stream (the sorted pass with overlap on: the usual case for large batches) its kernel runs BESIDE
   stages k_stats / k_stats_reduce and only the join + its reduce are left in this stage; the events sit on the main stream, so
   the side kernel's own duration is rocprofv3's to report (profiles/: k_stats<.., true>), and FPL_NO_OVERLAP=1 puts it back in
   line.  k_scan and k_stats are single launches: their event times are kernel durations.  k_trim_ends is a kernel duration only
   when the trims run in line: with the end trims ahead of the main stream (fpl_assume_inputs_ready, or the asynchronous path's own
   copy events) the stage holds what is LEFT of them when the main stream gets there -- usually nothing -- and the wait for
   ev_trim_done lies between mark(1) and k_scan, i.e. in the k_scan stage; with a FASTA chain the stage is two launches
   (k_trim_ends_batched<8 words, chain> + k_trim_ends<2>).  A caller that wants the trims' own duration times a few batches with
   the promise withdrawn (bench.py: roofline.k_trim_ends_ms_in_line) or reads rocprofv3's table */

/* capacities of the lists k_break_mask appends to: every region is at least one window long, so an output
   read or a piece costs at least window + 1 bytes of input beyond the two fragments a read starts with */
inline void break_mask_caps(u32 n_reads, uint64_t n_bytes, int brk, int brk_w, int msk, int msk_w, u32& frags, u32& regs,
                            u32& items) {
    const uint64_t bw = brk && brk_w > 0 ? (uint64_t)brk_w : ~0ull >> 1, mw = msk && msk_w > 0 ? (uint64_t)msk_w : ~0ull >> 1;
    const uint64_t f = 2ull * n_reads + n_bytes / (bw + 1) + 16;
    const uint64_t r = f + n_bytes / (mw + 1) + 16; /* <= one region per window of a read, plus one per output read */
    const uint64_t i = f + 2 * r;
    const uint64_t cap = 0x7FFFFFF0ull;
    frags = (u32)(f < cap ? f : cap);
    regs = (u32)(r < cap ? r : cap);
    items = (u32)(i < cap ? i : cap);
}

inline u32 cdiv(u32 a, u32 b) { return (a + b - 1) / b; }

/* Slices of the item list for k_stats.  A (slice, tile) block is heavy only while its tile lies below the
 * typical item length, so the number of HEAVY blocks is about slices x (mean length / tile).  Every block pays
 * for zeroing its tables, the 70 KiB hand-over and the 5-mer flush, so slices should be as large as the
 * 14-bit counter fields allow; measured optimum: ~1.25 heavy blocks per block slot of the chip (2 per CU). */
inline u32 stats_items_per_slice(u32 n_items, u32 mean_len, u32 n_cu, const StatsTune& tune) {
    if (n_items == 0) return 64;
    if (tune.per) return tune.per;
    const u32 heavy_tiles = mean_len / FS_T + 1;
    const u32 slices = cdiv(5 * n_cu / 2, heavy_tiles);
    u32 per = cdiv(n_items, slices);
    per = (per + 63) / 64 * 64;
    if (per < 1024) per = 1024;
    if (per > CS_MAX_ITEMS_PER_SLICE / 64 * 64) per = CS_MAX_ITEMS_PER_SLICE / 64 * 64;
    return per;
}
/* EXTRA pass (post-only fragments; count known only on the device): a fixed number of blocks per tile walk
 * slices of FS_EXTRA_PER items and hand over one slab each. */
constexpr u32 FS_EXTRA_PER = 1024;
constexpr u32 FS_EXTRA_BLOCKS = 32; /* (64: 0.07 ms more on the 1 M-read batch: every block zeroes and hands over 70 KiB) */
inline u32 stats_extra_per(const StatsTune& tune) { return tune.extra_per ? tune.extra_per : FS_EXTRA_PER; }
/* heavy_tiles / n_cu given: the pass runs on the side stream, beside the reduce of k_stats_sorted -- fewer, longer blocks then (every
   block zeroes and hands over its tables whatever it counted), as long as the tiles that hold most of the items -- those below the
   mean read length -- still bring about five blocks per four CUs: 18 per tile on the bench batch (0.60 -> 0.56 ms for the reduce +
   what is left of the pass), 16 on configs[3] (2.03 -> 1.7), 32 for 2 kb reads (16 there: 0.28 -> 0.47).  Never more than
   FS_EXTRA_BLOCKS: what the scratch buffers are sized for. */
inline u32 stats_extra_blocks(u32 n_reads, const StatsTune& tune, u32 heavy_tiles = 0, u32 n_cu = 0) {
    const u32 b = cdiv(2 * (n_reads ? n_reads : 1), stats_extra_per(tune));
    u32 cap = tune.extra_blocks ? tune.extra_blocks : FS_EXTRA_BLOCKS;
    if (!tune.extra_blocks && heavy_tiles && n_cu) {
        cap = cdiv(5 * n_cu / 4, heavy_tiles);
        cap = cap < 8 ? 8 : cap;
    }
    /* never more than FS_EXTRA_BLOCKS: that is what the side stream's slabs are sized for (a batch of 150 000 .. 163 000 reads
       shorter than a cycle tile used to get 2 n / 1024 blocks here -- found by tests/test_gpu_parity.py::test_batch_forms_...) */
    const u32 lim = cap < (u32)FS_EXTRA_BLOCKS ? cap : (u32)FS_EXTRA_BLOCKS;
    return b < lim ? b : lim;
}
/* items a block may accumulate before it must empty its tables (test hook: force that path) */
inline u32 stats_extra_max_acc(const StatsTune& tune) { return tune.extra_acc ? tune.extra_acc : CS_MAX_ITEMS_PER_SLICE; }
/* k_stats_sorted: a front trim shared by fewer reads than this is not given slices of its own (every slice costs one slab
 * hand-over per cycle tile its reads reach) */
constexpr u32 FS_MIN_BUCKET = 256;
inline u32 stats_min_bucket(const StatsTune& tune) { return tune.min_bucket ? tune.min_bucket : FS_MIN_BUCKET; }
/* upper bound of the slices k_bucket_plan can make: sum over buckets of ceil(count / per) */
inline u32 stats_sorted_max_slices(u32 n_reads, u32 per, const StatsTune& tune) {
    const u32 own = n_reads / stats_min_bucket(tune) + 1; /* buckets with slices of their own, "not post" included */
    /* (odd: block x + y * slices runs on XCD (x + y * slices) % 8 -- with a multiple of 8 every slice would stay on one XCD
       for all its tiles, and the XCDs, which take their blocks in turn, wait for the one that got the larger slices) */
    return (n_reads / per + 1 + (own < (u32)FS_NB ? own : (u32)FS_NB)) | 1u;
}
/* The sorted pass pays for its bucket kernels and for one slab per (slice, tile) of every front trim: measured against the
 * unsorted pass it wins from about 200 k reads of 9 kb on (1 M: 6.7 -> 5.9 ms, 300 k: 2.19 -> 2.05, 100 k: 0.87 -> 1.00), and a
 * batch of a few thousand reads would hand nearly every read to the EXTRA pass (no front trim is shared by 256 of them).
 * A test hook that sets the bucket threshold asks for the sorted pass whatever the size. */
constexpr u32 FS_SORT_MIN_READS = FPL_FORM_STATS_SORTED_MIN;
inline bool stats_use_sorted(u32 n_reads, const StatsTune& tune) {
    if (tune.sort_min) return n_reads >= tune.sort_min;
    if (tune.min_bucket) return true;
    return n_reads >= FS_SORT_MIN_READS;
}
inline size_t sort_ws_words(u32 max_slices, u32 n_reads) {
    return (size_t)SW_SLICES + 6 * (size_t)max_slices + (size_t)FS_NB * cdiv(n_reads ? n_reads : 1, FS_SORT_READS); /* slices, groups, block counts */
}
/* slabs (tiles x slices) the scratch buffer must hold for a batch */
inline size_t stats_scratch_slabs(u32 n_reads, uint64_t n_bytes, u32 max_read_len, u32 n_cu, const StatsTune& tune) {
    const u32 n_tiles = cdiv(max_read_len ? max_read_len : 1, FS_T);
    const u32 mean_len = n_reads ? (u32)(n_bytes / n_reads) : 0;
    const u32 per = stats_items_per_slice(n_reads, mean_len, n_cu, tune);
    u32 slices = cdiv(n_reads ? n_reads : 1, per);
    if (FPL_OPT_SORTSTATS) slices = stats_sorted_max_slices(n_reads, per, tune);
    if (slices < FS_EXTRA_BLOCKS) slices = FS_EXTRA_BLOCKS;
    return (size_t)slices * n_tiles;
}

/* does a batch take the sorted statistics pass?  (the persistent blocks number their (tile, slice) items with 32 bits; a batch
 * beyond that -- hundreds of millions of reads next to a read of hundreds of megabases -- takes the plain walk; with --break /
 * --mask no read is counted post-filter by the sorted pass: the plain walk does) */
inline bool stats_takes_sorted(u32 n_reads, uint64_t n_bytes, u32 max_read_len, u32 n_cu, const StatsTune& tune, bool defer) {
    if (!FPL_OPT_SORTSTATS || defer || n_reads == 0 || !stats_use_sorted(n_reads, tune)) return false;
    const u32 n_tiles = cdiv(max_read_len ? max_read_len : 1, FS_T);
    const u32 per = stats_items_per_slice(n_reads, (u32)(n_bytes / n_reads), n_cu, tune);
    return (uint64_t)stats_sorted_max_slices(n_reads, per, tune) * n_tiles < 0xFFFFFFF0ull;
}

#ifndef FPL_OPT_BATCH
#define FPL_OPT_BATCH 1 /* the usual adapter set goes through k_trim_ends_batched (confirmations 64 reads at a time) */
#endif
/* do the end trims of a batch run in k_trim_ends_batched (the usual adapter set, 64 reads per wave)? */
constexpr u32 TRIM_BATCH_MIN_READS = FPL_FORM_TRIM_BATCHED_MIN;
inline bool trim_takes_batched(u32 n_reads, int trim_mode, const StatsTune& tune) {
    const u32 batch_min = tune.trim_batch_min ? tune.trim_batch_min : TRIM_BATCH_MIN_READS;
    return (trim_mode == 1 || trim_mode == 2) && FPL_OPT_BATCH && n_reads >= batch_min;
}
/* is a batch large enough for its end trims to be worth a stream of their own (two more event hand-overs per batch)? */
inline bool trim_worth_ahead(u32 n_reads, const StatsTune& tune) {
    return n_reads >= (tune.trim_batch_min ? tune.trim_batch_min : TRIM_BATCH_MIN_READS);
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
    /* (profiling only, tools/overlap_probe.py: FPL_DEBUG_FLAGS 0x1000 stops a batch behind k_resolve / k_redo, 0x2000 runs only
       what follows them, on the state an earlier batch of the context left) */
    const bool run_front = !(a.dbg & 0x2000), run_back = !(a.dbg & 0x1000);
    /* one answer per batch, wherever it is asked below */
    const bool takes_sorted = a.sorted_form >= 0 ? a.sorted_form != 0 : stats_takes_sorted(n, a.n_bytes, a.max_read_len, a.n_cu, a.tune, a.defer);
    /* (... and 0x4000: of the front half only the end trims, 0x8000: only the scan) */
    const bool run_trim = run_front && !(a.dbg & 0x8000), run_scan = run_front && !(a.dbg & 0x4000);
    /* 1: one wave per read, grid-stride; cap the grid so the LDS accumulators flush rarely */
    if (run_trim) {
        const fpl_stream_t ts = a.trim_stream ? a.trim_stream : stream;
        u32 blocks = cdiv(n, KWAVES);
#ifndef FPL_TRIM_BLOCKS_PER_CU
#define FPL_TRIM_BLOCKS_PER_CU 112 /* static grid-stride: more, shorter blocks even the load out (16: 3.80 ms, 112: 3.57 ms on the bench batch) */
#endif
        const u32 cap = FPL_TRIM_BLOCKS_PER_CU * a.n_cu;
        if (blocks > cap) blocks = cap;
        /* the usual adapter sets have their own, much smaller instantiations (DevConfig::trim_mode) */
        /* (a wave of the batched kernel walks its 64 reads one after the other through the per-read phases: 0.4 ms however few
           groups there are -- 2 000 reads: 0.42 ms against 0.03 for a wave per read; the two meet at about 100 k reads) */
        if (trim_takes_batched(n, a.trim_mode, a.tune)) {
            /* a wave takes 64 reads per round: enough waves to fill the chip, few enough to keep every wave a few rounds long */
            u32 gblocks = cdiv(cdiv(n, 64u), KWAVES);
            u32 gcap = FPL_TRIM_WAVES_PER_SIMD_BATCHED * a.n_cu; /* blocks of 4 waves a CU holds: waves per SIMD */
            /* AHEAD of the main stream the kernel runs beside k_scan of the batch before.  A wave of either holds 96 vector registers:
               five fill a SIMD, and k_scan's grid is five blocks per CU -- with its full grid the trim kernel takes places k_scan's
               persistent blocks then lack and k_scan loses what the trims gain; with two blocks per CU three or four k_scan blocks run
               beside them while they last (c3: 11.64 ms per step; two blocks 11.27-11.41, against 11.50-11.60 with the trims held
               back until the statistics kernel of the batch before is done; profiles/r06_v4/ab_trims_beside_scan.txt) */
            const u32 ahead_blocks = a.tune.trim_ahead_blocks ? a.tune.trim_ahead_blocks : 2u;
            if (a.trim_stream && ahead_blocks < FPL_TRIM_WAVES_PER_SIMD_BATCHED) gcap = ahead_blocks * a.n_cu;
            if (gblocks > gcap) gblocks = gcap;
            if (a.trim_mode == 1)
                FPL_LAUNCH((k_trim_ends_batched<KWAVES>), dim3(gblocks), block, ts, a.seq, a.qual, a.off, n, a.n_bytes, a.cfg,
                           a.ads, a.state, a.counters, a.C, a.work_ctr + 2);
            else if (a.n_fasta == 0) /* command-line adapters of 33..64 bases, no FASTA list: the lane-per-read kernel is all there is */
                FPL_LAUNCH((k_trim_ends_batched<KWAVES, 8, false>), dim3(gblocks), block, ts, a.seq, a.qual, a.off, n, a.n_bytes, a.cfg,
                           a.ads, a.state, a.counters, a.C, a.work_ctr + 2);
            else {
                /* a FASTA chain: trimAndCut, polyX and the two command-line adapters with lane = read, then the chain with a wave per
                   read from the state the first kernel left (c5: 4.3 of the chain kernel's 15 ms were those four steps) */
                FPL_LAUNCH((k_trim_ends_batched<KWAVES, 8, true>), dim3(gblocks), block, ts, a.seq, a.qual, a.off, n, a.n_bytes, a.cfg,
                           a.ads, a.state, a.counters, a.C, a.work_ctr + 2);
                if (FPL_OPT_ONEFP && a.n_fasta <= 64) /* one group of adapters: one Peq table per block */
                    FPL_LAUNCH((k_trim_ends<KWAVES, 2, true>), dim3(blocks), block, ts, a.seq, a.qual, a.off, n, a.n_bytes, a.cfg,
                               a.ads, a.state, a.counters, a.C, 1);
                else
                    FPL_LAUNCH((k_trim_ends<KWAVES, 2>), dim3(blocks), block, ts, a.seq, a.qual, a.off, n, a.n_bytes, a.cfg,
                               a.ads, a.state, a.counters, a.C, 1);
            }
        } else if (a.trim_mode == 1)
            FPL_LAUNCH((k_trim_ends<KWAVES, 1>), dim3(blocks), block, ts, a.seq, a.qual, a.off, n, a.n_bytes, a.cfg,
                       a.ads, a.state, a.counters, a.C, 0);
        else if (a.trim_mode == 2)
            FPL_LAUNCH((k_trim_ends<KWAVES, 2>), dim3(blocks), block, ts, a.seq, a.qual, a.off, n, a.n_bytes, a.cfg,
                       a.ads, a.state, a.counters, a.C, 0);
        else
            FPL_LAUNCH((k_trim_ends<KWAVES, 0>), dim3(blocks), block, ts, a.seq, a.qual, a.off, n, a.n_bytes, a.cfg,
                       a.ads, a.state, a.counters, a.C, 0);
#ifndef FPL_EMU
        if (a.trim_stream) {
            (void)hipEventRecord((hipEvent_t)a.ev_trim_done, a.trim_stream);
            (void)hipStreamWaitEvent(stream, (hipEvent_t)a.ev_trim_done, 0);
        }
#endif
    }
    mark(1);
    if (run_scan) {
        u32 blocks = cdiv(n, KWAVES);
        const u32 cap = SCAN_BLOCKS_PER_CU * a.n_cu; /* what registers / LDS admit */
        if (blocks > cap) blocks = cap;
        /* ~32 dequeues per wave keep the tail short, but never fewer than 4 reads per dequeue once there is
           that much work: the single work counter sustains only ~80 atomics/us */
        const u32 waves = blocks * KWAVES;
#ifndef FPL_SCAN_CHUNK_DIV
#define FPL_SCAN_CHUNK_DIV 32u
#endif
        u32 chunk = n / (waves * FPL_SCAN_CHUNK_DIV);
        if (chunk < 4) chunk = n >= 8 * waves ? 4 : (n >= 2 * waves ? 2 : 1);
        if (chunk > 64) chunk = 64;
        if (a.tune.scan_chunk) chunk = a.tune.scan_chunk < 64 ? a.tune.scan_chunk : 64;
        if (a.scan_short)
            FPL_LAUNCH((k_scan<KWAVES, true>), dim3(blocks), block, stream, a.seq, a.qual, a.off, n, a.n_bytes, a.cfg, a.ads,
                       (const ReadState*)a.state, a.recs, a.wins, a.counters, a.C, a.work_ctr, chunk);
        else
            FPL_LAUNCH((k_scan<KWAVES, false>), dim3(blocks), block, stream, a.seq, a.qual, a.off, n, a.n_bytes, a.cfg, a.ads,
                       (const ReadState*)a.state, a.recs, a.wins, a.counters, a.C, a.work_ctr, chunk);
    }
    mark(2);
    /* the sorted statistics pass starts from two zeroed buffers: cleared here, in front of k_resolve, where nothing runs beside the
       fill kernels (behind it they share the chip with k_redo on the side stream and take 0.1 ms to find room) */
    bool sorted_zeroed = false;
    auto zero_sorted_ws = [&]() {
        if (sorted_zeroed) return;
        const u32 tiles = cdiv(a.max_read_len ? a.max_read_len : 1, FS_T);
        const u32 ms = stats_sorted_max_slices(n, stats_items_per_slice(n, (u32)(a.n_bytes / n), a.n_cu, a.tune), a.tune);
        FPL_MEMSET(a.sort_ws, (size_t)SW_SLICES * sizeof(u32), stream);
        FPL_MEMSET(a.stats_flags, (size_t)ms * tiles + tiles, stream);
        sorted_zeroed = true;
    };
    if (run_front && run_back && !(a.dbg & 0xC000) && takes_sorted) zero_sorted_ws();
    if (run_front && !(a.dbg & 0xC000)) {
        /* lane = read: confirmations, gaps, records, counters, plan; the reads a middle adapter splits go on the REDO list */
        /* (sixteen waves per block and no more than two blocks per CU: every block ends with a few hundred global atomics on
           the same dozen cache lines -- its median histograms -- and those serialise: 977 blocks spent 0.1 ms on them) */
        u32 rblocks = cdiv(cdiv(n, 64u), RWAVES);
        if (rblocks > 2 * a.n_cu) rblocks = 2 * a.n_cu;
        FPL_LAUNCH((k_resolve<RWAVES>), dim3(rblocks), dim3(RWAVES * 64), stream, a.seq, a.off, n, a.n_bytes, a.cfg, a.ads, a.state,
                   (const ScanRec*)a.recs, (const ScanWin*)a.wins, a.results, a.frag_off, a.frag_len, a.work_ctr + 1, a.redo,
                   a.work_ctr + 3, a.counters, a.C);
        if (!a.defer) { /* (with --break / --mask k_break_mask scans the fragments) */
            /* the list's length is known on the device only: a grid that walks it (long reads first, REDO_LONG) */
#ifdef FPL_EMU
            constexpr int DW = KWAVES;
#else
            constexpr int DW = FPL_REDO_WAVES;
#endif
            u32 dblocks = cdiv(n, DW);
            if (dblocks > FPL_REDO_BLOCKS_PER_CU * a.n_cu) dblocks = FPL_REDO_BLOCKS_PER_CU * a.n_cu;
            /* the sorted statistics pass takes the split reads as the trim kernel's plan has them ("not post"): k_redo then writes
               no plan, the bucket kernels and k_stats_sorted need nothing of it, and -- when that pass forks the side stream for
               its post-only part anyway -- it runs there, in front of that part, beside them (0.15 ms of a mostly idle chip on
               the bench batch; the fragments it finds go on the EXTRA list, which only the post-only pass reads) */
            const bool sorted = run_back && takes_sorted;
            fpl_stream_t rs = stream;
            if (sorted && a.aux != nullptr && a.extra_scratch != nullptr && !a.tune.redo_inline) {
                FPL_FORK_MARK(a, stream);
                rs = FPL_FORK(a, stream);
            }
            if (sorted)
                FPL_LAUNCH((k_redo<DW, false>), dim3(dblocks), dim3(DW * 64), rs, a.seq, a.qual, a.off, n, a.n_bytes, a.cfg, a.state,
                           (const ScanRec*)a.recs, a.results, a.frag_off, a.frag_len, a.work_ctr + 1, (const RedoItem*)a.redo,
                           (const u32*)(a.work_ctr + 3), a.work_ctr + 4, a.counters, a.C);
            else
                FPL_LAUNCH((k_redo<DW, true>), dim3(dblocks), dim3(DW * 64), stream, a.seq, a.qual, a.off, n, a.n_bytes, a.cfg, a.state,
                           (const ScanRec*)a.recs, a.results, a.frag_off, a.frag_len, a.work_ctr + 1, (const RedoItem*)a.redo,
                           (const u32*)(a.work_ctr + 3), a.work_ctr + 4, a.counters, a.C);
        }
    }
    if (a.defer && run_front) {
        u32 blocks = cdiv(n, KWAVES);
        const u32 cap = 8 * a.n_cu;
        if (blocks > cap) blocks = cap;
        FPL_MEMSET(a.bm.counts, 3 * sizeof(u32), stream);
        FPL_LAUNCH((k_break_mask<KWAVES>), dim3(blocks), block, stream, a.seq, a.qual, a.off, n, a.cfg, a.results, a.bm,
                   a.frag_off, a.frag_len, a.frag_cyc, a.work_ctr + 1, a.counters, a.C);
    }
    mark(3);
    if (!run_back) {
        for (int i = 4; i <= N_STAGES; i++) mark(i);
        return;
    }
    const u32 n_tiles = cdiv(a.max_read_len ? a.max_read_len : 1, FS_T);
    /* the post-only pass over the EXTRA list (fragments of split reads, far-trimmed reads): the statistics kernel, then its reduce */
    bool extra_forked = false;
    auto launch_extra = [&](fpl_stream_t st, u64* scratch, u8* flags, bool reduce) {
        const u32 n_items = a.defer ? a.bm.item_cap : 2 * n; /* upper bound; the kernel reads the real count */
        const u32 gx = stats_extra_blocks(a.defer ? (n_items + 1) / 2 : n, a.tune, extra_forked ? (u32)(a.n_bytes / n) / FS_T + 1 : 0,
                                          a.n_cu); /* slabs per tile */
        if (!reduce) {
            FPL_MEMSET(flags, (size_t)gx * n_tiles + n_tiles, st);
            FPL_LAUNCH((k_stats<SWAVES, true>), dim3(gx, n_tiles), dim3(SWAVES * 64), st, a.seq, a.qual, a.n_bytes,
                       (const uint64_t*)a.frag_off, (const u32*)a.frag_len, (const u32*)a.frag_cyc, (const ReadState*)nullptr, n_items,
                       (const u32*)(a.work_ctr + 1), stats_extra_per(a.tune), gx, stats_extra_max_acc(a.tune), a.counters, scratch,
                       flags, a.C);
        } else {
            FPL_LAUNCH(k_stats_reduce, dim3(16 * FS_T / 256, n_tiles), dim3(256), st, (const u64*)scratch, (const u8*)flags, gx,
                       n_tiles, a.counters, a.C, 0);
        }
    };
    const u32 per_sorted = stats_items_per_slice(n, (u32)(a.n_bytes / n), a.n_cu, a.tune);
    if (takes_sorted) {
        const u32 per = per_sorted;
        const u32 max_slices = stats_sorted_max_slices(n, per, a.tune);
        zero_sorted_ws();
        const u32 nblk = cdiv(n, FS_SORT_READS);
        u32* const blkcnt = a.sort_ws + SW_SLICES + 6 * (size_t)max_slices;
        FPL_LAUNCH(k_bucket_count, dim3(nblk), dim3(FS_SORT_BLK), stream, (const ReadState*)a.state, n, blkcnt);
        FPL_LAUNCH(k_bucket_scan, dim3(FS_NB), dim3(256), stream, blkcnt, nblk, a.sort_ws);
        FPL_LAUNCH(k_bucket_plan, dim3(1), dim3(128), stream, a.sort_ws, per, stats_min_bucket(a.tune), max_slices,
                   a.tune.group ? a.tune.group : (u32)FS_GROUP);
        /* from this cycle tile on the items are groups of slices: two and a half times the mean read length -- beyond it a slice
           holds a few rows per tile (any value is correct: a group's rows are counted before they share a slab; measured flat
           between 1.5 and 3 times the mean, worse below: groups of full slices are items too heavy to balance) */
        const u32 hi_tile = a.tune.hi_tile ? a.tune.hi_tile - 1 : 5 * ((u32)(a.n_bytes / n) / FS_T) / 2 + 2;
        u32 max_rows = a.tune.group_rows ? a.tune.group_rows : CS_MAX_ITEMS_PER_SLICE;
        if (max_rows < per) max_rows = per; /* (one slice always fits) */
        if (max_rows > CS_MAX_ITEMS_PER_SLICE) max_rows = CS_MAX_ITEMS_PER_SLICE;
        FPL_LAUNCH(k_bucket_scatter, dim3(nblk), dim3(FS_SORT_BLK), stream, a.off, (const ReadState*)a.state, n, a.sort_ws,
                   (const u32*)blkcnt, a.st_off, a.st_len, a.st_e, a.frag_off, a.frag_len, a.work_ctr + 1);
        mark(4);
        /* the post-only (EXTRA) pass needs nothing of what follows: on the side stream its blocks fill the slots the persistent
           blocks of k_stats_sorted leave as they run out of items, and run on beside the reduce kernel (a bandwidth kernel) */
        extra_forked = a.aux != nullptr && a.extra_scratch != nullptr;
        if (extra_forked) FPL_FORK_MARK(a, stream);
        /* persistent blocks, two per CU (what the LDS tables allow) */
#ifndef FPL_STATS_BLOCKS_PER_CU
#define FPL_STATS_BLOCKS_PER_CU 2 /* persistent blocks of 16 waves and 80 KB of LDS: what a CU holds */
#endif
        FPL_LAUNCH((k_stats_sorted<SWAVES>), dim3(FPL_STATS_BLOCKS_PER_CU * a.n_cu), dim3(SWAVES * 64), stream, a.seq, a.qual, a.n_bytes,
                   (const uint64_t*)a.st_off, (const u32*)a.st_len, (const u32*)a.st_e, a.sort_ws, max_slices, n_tiles, a.counters,
                   a.stats_scratch, a.stats_flags, a.C, hi_tile, max_rows);
        if (extra_forked) launch_extra(FPL_FORK(a, stream), a.extra_scratch, a.extra_flags, false);
#ifndef FPL_EMU
        if (a.ev_stats_done) (void)hipEventRecord((hipEvent_t)a.ev_stats_done, stream);
#endif
        mark(5);
        FPL_LAUNCH(k_stats_reduce_sorted, dim3(16 * FS_T / 256, n_tiles), dim3(256), stream, (const u64*)a.stats_scratch,
                   (const u8*)a.stats_flags, (const u32*)a.sort_ws, max_slices, n_tiles, a.counters, a.C);
    } else {
        const u32 per = stats_items_per_slice(n, (u32)(a.n_bytes / n), a.n_cu, a.tune);
        const u32 n_slices = cdiv(n, per);
        FPL_MEMSET(a.stats_flags, (size_t)n_slices * n_tiles + n_tiles, stream);
        mark(4);
        FPL_LAUNCH((k_stats<SWAVES, false>), dim3(n_slices, n_tiles), dim3(SWAVES * 64), stream, a.seq, a.qual, a.n_bytes,
                   a.off, (const u32*)nullptr, (const u32*)nullptr, (const ReadState*)a.state, n, (const u32*)nullptr, per,
                   n_slices,
                   CS_MAX_ITEMS_PER_SLICE, a.counters, a.stats_scratch, a.stats_flags, a.C);
#ifndef FPL_EMU
        if (a.ev_stats_done) (void)hipEventRecord((hipEvent_t)a.ev_stats_done, stream);
#endif
        mark(5);
        FPL_LAUNCH(k_stats_reduce, dim3(16 * FS_T / 256, n_tiles), dim3(256), stream, (const u64*)a.stats_scratch,
                   (const u8*)a.stats_flags, n_slices, n_tiles, a.counters, a.C, 1);
    }
    mark(6);
    if (extra_forked) {
        FPL_JOIN(a, stream);
        launch_extra(stream, a.extra_scratch, a.extra_flags, true); /* (the reduce: it adds to the counters the reduce above owns) */
    } else {
        launch_extra(stream, a.stats_scratch, a.stats_flags, false);
        launch_extra(stream, a.stats_scratch, a.stats_flags, true);
    }
    mark(7);
}

}  // namespace fpl
#endif
