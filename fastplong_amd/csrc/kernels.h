/*
 * kernels.h -- the per-read hot path as CDNA4 (gfx950) kernels.  wave = 64 lanes.
 *
 *   k_trim_ends   : Filter::trimAndCut -> PolyX::trimPolyX -> trimBySequenceStart/End ->
 *                   trimByMultiSequences.  One wave per read, lanes = candidate positions.
 *   k_stats       : the per-cycle / k-mer part of Stats::statRead, pre- and post-filter tables in one
 *                   pass.  Block = (cycle tile, slice of reads); counters privatised in LDS.
 *   k_scan        : whole-read pass on r1: both middle-adapter Hamming scans
 *                   (findMiddleAdapters), the passFilter sums, the per-read quality histogram
 *                   (median, base-quality histogram); then resolves the read: Levenshtein
 *                   confirm, breakByGap, passFilter code, result record, counters.
 *
 * Reference behaviour restated here is cited per function (file:line under the reference's
 * src/).  All arithmetic is integer; `round(edMax*len)` arrives as the host-built table
 * DevConfig::thr.
 */
#ifndef FPL_KERNELS_H
#define FPL_KERNELS_H

#include "adapter_pick.h"
#include "dev_prims.h"
#include "dev_types.h"

namespace fpl {

/* Section timers, profiling builds only (-DFPL_PROF, tools/prof_sections.sh): every wave sums the cycles
   (s_memtime) it spends between PROF marks; fpl_debug_prof() reads the totals back. */
#if defined(FPL_PROF) && !defined(FPL_EMU)
__device__ unsigned long long g_fpl_prof[64];
#define PROF_INIT()                                     \
    unsigned long long prof_a[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; \
    unsigned long long prof_t = __builtin_amdgcn_s_memtime()
#define PROF(i)                                                     \
    {                                                               \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        prof_a[i] += t_ - prof_t;                                   \
        prof_t = t_;                                                \
    }
#define PROF_FLUSH(base)                                                                      \
    if (lane_id() == 0)                                                                       \
        for (int i_ = 0; i_ < 12; i_++) atomicAdd(&g_fpl_prof[(base) + i_], prof_a[i_])
#else
#define PROF_INIT()
#define PROF(i)
#define PROF_FLUSH(base)
#endif


/* compile-time A/B switches of individual optimisations (tools/ab_bench.py builds the variants; the defaults ship) */
#ifndef FPL_REDO_PREFETCH
#define FPL_REDO_PREFETCH 1 /* k_redo: the next tile's cache lines are touched one tile ahead */
#endif
#ifndef FPL_REDO_WAVES
#define FPL_REDO_WAVES 8 /* k_redo: waves per block; three such blocks fit a CU (LDS: 4.5 KiB per wave, 70 VGPRs) */
#endif
#ifndef FPL_REDO_BLOCKS_PER_CU
#define FPL_REDO_BLOCKS_PER_CU 3
#endif
#ifndef FPL_OPT_HIST
#define FPL_OPT_HIST 2 /* histogram counter address = (bin bits) | (4 KiB-aligned slice + lane copy), hist_bump.  2: written in
                          plain C, the compiler picks v_and_or_b32 (32 VALU fewer per tile, k_scan 1.4 % faster side by
                          side); 1: the same through inline asm -- 2.5 % SLOWER (profiles/r02_ab: the asm pins the counter
                          updates in place); 0: index arithmetic */
#endif
#ifndef FPL_OPT_SORTSTATS
#define FPL_OPT_SORTSTATS 1 /* the statistics pass walks the reads sorted by their front trim: one table update per base instead
                               of two (k_stats_sorted) */
#endif
#ifndef FPL_OPT_ACGT
#define FPL_OPT_ACGT 1 /* k_scan: full tiles made of A, C, G, T, N only take three code bit-planes through the scan (range_scan_fast).
                          2: the ragged last tile of a range too -- measured the same (6.756 vs 6.753 ms, 2 kb reads 10.62 vs
                          10.54 ms, profiles/r02_ab), so the ragged tile keeps the byte-masked variants */
#endif
#ifndef FPL_OPT_CSA16
#define FPL_OPT_CSA16 1 /* k_scan match counts: carry-save groups of sixteen adapter bases where the adapter has them */
#endif
#ifndef FPL_OPT_BCNT
#define FPL_OPT_BCNT 1 /* v_bcnt_u32_b32 with its addend in the passFilter sums (sums32) */
#endif
#ifndef FPL_OPT_NB6
#define FPL_OPT_NB6 1 /* six count planes instead of seven when both adapters have <= 32 bases */
#endif
#ifndef FPL_OPT_SCANBATCH
#define FPL_OPT_SCANBATCH 1 /* k_scan: the middle-adapter confirmations of up to 32 reads in one lane-parallel pass */
#endif
#ifndef FPL_OPT_FASTAFILTER
#define FPL_OPT_FASTAFILTER 2 /* k_trim_ends<2>: a lane-per-adapter Myers search pass over the two end windows decides which
                                 adapters of the FASTA list get the exact trims at all.  1: a 64-column run for the whole adapter
                                 and a 16-column run for the partial pattern (fasta_may_trim); 2: one 32-column run with two
                                 score taps (fasta_may_trim32) */
#endif
#ifndef FPL_OPT_FASTANEAR
#define FPL_OPT_FASTANEAR 1 /* k_trim_ends<2>: without a whole-adapter flag, partial-pattern hits only count next to the read's end */
#endif
#ifndef FPL_OPT_DPPPREV
#define FPL_OPT_DPPPREV 1 /* "the dword of the lane in front" (k_scan's predecessor byte, the 5-mer halo of the statistics kernels) through DPP
                             wave_shr:1 instead of ds_bpermute: one vector op, no trip through the LDS crossbar */
#endif
#ifndef FPL_OPT_VALADDC
#define FPL_OPT_VALADDC 1 /* sliced_max: the value bit by bit through add-with-carry */
#endif
#ifndef FPL_OPT_PADSCALAR
#define FPL_OPT_PADSCALAR 1 /* k_scan: the ragged last tile of a range is padded with wave-uniform byte masks (one lane is cut by
                               the end of the range, and which one is a scalar) instead of per-lane ones: 25 instead of 97 vector
                               instructions per read */
#endif
#ifndef FPL_OPT_VMFULL
#define FPL_OPT_VMFULL 1 /* k_scan: the mask of testable window positions is worked out only in the tiles where it is not all ones */
#endif
#ifndef FPL_RED_UNROLL
#define FPL_RED_UNROLL 4 /* k_stats_reduce_sorted: slabs whose cells a thread has in flight at a time */
#endif
#ifndef FPL_OPT_PACKRED
#define FPL_OPT_PACKRED 1 /* k_scan / k_redo: the wave reductions behind a range scan take two values each where the range's length allows */
#endif
#ifndef FPL_OPT_INCVALU
#define FPL_OPT_INCVALU 1 /* k_stats_sorted: a byte's packed increment built on the vector unit instead of read from a 256-entry LDS table:
                             the kernel's limit is the LDS array (24 LDS instructions per row of 512 bytes were 16 now), 5.10 -> 4.93 ms */
#endif
#ifndef FPL_OPT_KMER6
#define FPL_OPT_KMER6 1 /* k_stats_sorted: in the rows whose tile lies inside r1 the 5-mer updates of two neighbouring windows are ONE update of a
                           6-mer table (4096 bins: window k is the 6-mer's first five bases, window k + 1 its last five; unfolded at the
                           hand-over) -- four ds_add_u32 per 8 bytes and lane instead of eight, into four times the bins.  The table's 16 KB
                           come out of the per-cycle cells: the class rows 0 and 2 of the LDS tables (bytes whose low three bits are 000 or
                           010: no base letter of any case) hold the two 5-mer tables and the block's scalars, and such bytes are counted
                           with global atomics (exact, never taken by DNA) */
#endif
#ifndef FPL_OPT_KMERKEEP
#define FPL_OPT_KMERKEEP 1 /* k_stats_sorted (with FPL_OPT_KMER6): the 5-mer / 6-mer tables of a persistent block live through all its items --
                              zeroed once, flushed once -- instead of per (tile, slice) item: counts are sums, whichever item they came from */
#endif
#ifndef FPL_OPT_INCPERM
#define FPL_OPT_INCPERM 1 /* k_stats_sorted: the Q20 / Q30 half of a byte's packed increment through one v_perm per byte (20 instead of 32
                             vector instructions per row of 512 bytes) */
#endif
#ifndef FPL_OPT_STATSETUP
#define FPL_OPT_STATSETUP 1 /* k_stats_sorted: a row's 5-mer stream from v_dot4 packs and the NEIGHBOUR's finished pack (one DPP move), lane 0's
                               halo once per group of four rows -- instead of packing the neighbour's bytes a second time in every lane */
#endif
#ifndef FPL_OPT_PAIR
#define FPL_OPT_PAIR 1 /* k_scan (usual configuration): the head of the NEXT read of a wave's chunk rides in the lanes the last, ragged
                          tile of a read leaves empty (range_scan_fast<.., PAIR>) */
#endif
#ifndef FPL_PAIR_MIN_LANES
#define FPL_PAIR_MIN_LANES 12 /* ... when at least this many lanes are left for it (a packed tile costs ~100 instructions more) */
#endif
#ifndef FPL_OPT_TRIMTOUCH
#define FPL_OPT_TRIMTOUCH 1 /* k_trim_ends_batched: a phase asks for the cache lines its dependent loads will walk into -- both ends of the
                               qualities and of the bases in front of trimAndCut / polyX, the second line of a window in front of a window
                               scan -- all at once, so that a lane sits out ONE trip to memory per phase instead of one per line
                               (section timers, profiles/r05_trim: trimAndCut + polyX were a third of the kernel's wave cycles) */
#endif
#ifndef FPL_OPT_TRIMREG
#define FPL_OPT_TRIMREG 1 /* k_trim_ends_batched: the first 16 steps of trimAndCut's two window scans and the first 32 of polyX's tail scan
                             run out of 16-byte blocks loaded up front (a lane's step count is the wave's loop count, so the byte a step
                             needs sits at a compile-time place of the block) instead of one dependent byte load per step */
#endif
#ifndef FPL_OPT_PARTLANES
#define FPL_OPT_PARTLANES 1 /* k_trim_ends_batched: the partial-pattern searches with lane = read on the columns the search
                               pass leaves open (partial16_candidates / partial16_resolve_lanes) instead of a wave and 184
                               windows per flagged read */
#endif
#ifndef FPL_OPT_SGFILTER
#define FPL_OPT_SGFILTER 1 /* k_trim_ends_batched: a lane-parallel Myers search pass decides which reads need the
                              partial-pattern search at all (partial16_possible) */
#endif
#ifndef FPL_OPT_ONEHOT
#define FPL_OPT_ONEHOT 1 /* window Hamming scans of k_trim_ends on one-hot nibbles */
#endif

/* profiling-only ablation switches (FPL_DEBUG_FLAGS); compiled out unless -DFPL_ABLATE */
#ifdef FPL_ABLATE
#define FPL_DBG(x, bit) ((x) & (bit))
#else
#define FPL_DBG(x, bit) 0
#endif

/* =========================================================================================
 * Bit-parallel Levenshtein (Myers 1999 / Hyyro 2001, global-distance variant).  The
 * reference's edit_distance (src/editdistance.cpp:30-61,100-126) computes the same exact
 * distance; the pattern here is always a slice of an adapter, whose Peq vectors the host
 * prebuilt (DevAdapter), and the text is a window of the read.
 * ======================================================================================= */

/* pattern <= 16 columns in one 32-bit word; text of n <= 16 bytes.  The n table lookups are
 * issued up front (they do not depend on the recurrence), then the recurrence runs in registers. */
__device__ __forceinline__ int lev_bp32(const u32* __restrict__ peq, int m, const u8* __restrict__ text, int n) {
    if (m == 0) return n;
    u32 eq[16];
#pragma unroll
    for (int j = 0; j < 16; j++) eq[j] = j < n ? peq[text[j]] : 0u;
    u32 Pv = ~0u, Mv = 0;
    int score = m;
    const u32 top = 1u << (m - 1);
#pragma unroll
    for (int j = 0; j < 16; j++) {
        if (j < n) { /* wave-uniform */
            const u32 Eq = eq[j];
            const u32 Xv = Eq | Mv;
            const u32 Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
            u32 Ph = Mv | ~(Xh | Pv);
            u32 Mh = Pv & Xh;
            score += (Ph & top) ? 1 : ((Mh & top) ? -1 : 0);
            Ph = (Ph << 1) | 1u; /* D[0][j] = j */
            Mh <<= 1;
            Pv = Mh | ~(Xv | Ph);
            Mv = Ph & Xv;
        }
    }
    return score;
}

/* word w of (peq_full[c] >> shift); PW = words the table keeps per byte value (PEQ_WORDS, or 1 for the LDS copy of
   an adapter of <= 64 bases) */
template <int PW>
__device__ __forceinline__ u64 peq_word(const uint64_t (*__restrict__ peq)[PW], int c, int shift, int w) {
    int i = (shift >> 6) + w, r = shift & 63;
    u64 lo = i < PW ? peq[c][i] : 0;
    if (r == 0) return lo;
    u64 hi = (i + 1) < PW ? peq[c][i + 1] : 0;
    return (lo >> r) | (hi << (64 - r));
}

/* pattern = adapter[shift, shift+m), up to 4 x 64 columns (block-wise with horizontal carries) */
__device__ __forceinline__ int lev_bp64(const uint64_t (*__restrict__ peq)[PEQ_WORDS], int shift, int m,
                                        const u8* __restrict__ text, int n) {
    if (m == 0) return n;
    if (n == 0) return m;
    const int W = (m + 63) >> 6;
    constexpr int MW = PEQ_WORDS;
    u64 Pv[MW], Mv[MW];
#pragma unroll
    for (int b = 0; b < MW; b++) {
        Pv[b] = ~0ull;
        Mv[b] = 0;
    }
    int score = m;
    const u64 last_top = 1ull << ((m - 1) & 63);
    for (int j = 0; j < n; j++) {
        const int c = text[j];
        int hin = 1; /* D[0][j] - D[0][j-1] */
#pragma unroll
        for (int b = 0; b < MW; b++) {
            if (b < W) {
                u64 Eq = peq_word(peq, c, shift, b);
                const u64 pv = Pv[b], mv = Mv[b];
                const u64 Xv = Eq | mv;
                if (hin < 0) Eq |= 1ull;
                const u64 Xh = (((Eq & pv) + pv) ^ pv) | Eq;
                u64 Ph = mv | ~(Xh | pv);
                u64 Mh = pv & Xh;
                const u64 hb = (b == W - 1) ? last_top : (1ull << 63);
                int hout = 0;
                if (Ph & hb) hout = 1;
                else if (Mh & hb) hout = -1;
                Ph <<= 1;
                Mh <<= 1;
                if (hin < 0) Mh |= 1ull;
                else if (hin > 0) Ph |= 1ull;
                Pv[b] = Mh | ~(Xv | Ph);
                Mv[b] = Ph & Xv;
                hin = hout;
            }
        }
        score += hin;
    }
    return score;
}

/* Column loop of the wave-cooperative Levenshtein (see lev_wave): `pub` holds, per 64-column round,
 * the Peq words of the text bytes.  Returns the exact distance when it is <= thr, and some value
 * > thr otherwise: along the last DP row the value drops by at most 1 per column, so once
 * score - (columns left) > thr the answer is known and the loop stops.  Patterns of <= 32 columns
 * run on one 32-bit word (half the instructions of the 64-bit form). */
__device__ __forceinline__ bool lev_round32(const WaveVals64& pub, int cnt, int left_after, u32& Pv, u32& Mv, int& score,
                                            u32 top, int thr) {
    /* every value here is wave-uniform, so this loop runs on the scalar unit, which all four SIMDs of a CU
       share: keep it short -- the score moves by a bit-field of Ph / Mh, and the exit bound is tested once per
       two columns (it only grows: the bound at column t implies the one at t + 1 or an exact result) */
    const u32 tsh = (u32)__builtin_ctz(top);
    /* one column of the recurrence (everything in scalar registers) */
#define FPL_LEV_COL(tt)                                               \
    {                                                                 \
        const u32 Eq = (u32)pub.get(tt);                              \
        const u32 Xv = Eq | Mv;                                       \
        const u32 Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;                  \
        u32 Ph = Mv | ~(Xh | Pv);                                     \
        u32 Mh = Pv & Xh;                                             \
        score += (int)((Ph >> tsh) & 1u) - (int)((Mh >> tsh) & 1u);   \
        Ph = (Ph << 1) | 1u;                                          \
        Mh <<= 1;                                                     \
        Pv = Mh | ~(Xv | Ph);                                         \
        Mv = Ph & Xv;                                                 \
    }
    int t = 0;
#ifndef FPL_LEV_UNROLL
#define FPL_LEV_UNROLL 4 /* columns between two tests of the exit bound: the test and the loop bookkeeping cost as much
                            as a column on the scalar unit, which is what bounds k_trim_ends */
#endif
    for (; t + FPL_LEV_UNROLL <= cnt; t += FPL_LEV_UNROLL) {
#pragma unroll
        for (int u = 0; u < FPL_LEV_UNROLL; u++) FPL_LEV_COL(t + u)
        if (score - (left_after + cnt - FPL_LEV_UNROLL - t) > thr) return true; /* wave-uniform */
    }
    for (; t < cnt; t++) FPL_LEV_COL(t)
    if (score - left_after > thr) return true;
#undef FPL_LEV_COL
    return false;
}
template <int MW>
__device__ __forceinline__ bool lev_round64(const WaveVals64 (&pub)[MW], int W, int cnt, int left_after,
                                            u64 (&Pv)[MW], u64 (&Mv)[MW], int& score, u64 last_top, int thr) {
    for (int t = 0; t < cnt; t++) {
        int hin = 1; /* D[0][j] - D[0][j-1] */
#pragma unroll
        for (int b = 0; b < MW; b++) {
            if (b < W) {
                u64 Eq = pub[b].get(t);
                const u64 pv = Pv[b], mv = Mv[b];
                const u64 Xv = Eq | mv;
                if (hin < 0) Eq |= 1ull;
                const u64 Xh = (((Eq & pv) + pv) ^ pv) | Eq;
                u64 Ph = mv | ~(Xh | pv);
                u64 Mh = pv & Xh;
                const u64 hb = (b == W - 1) ? last_top : (1ull << 63);
                int hout = 0;
                if (Ph & hb) hout = 1;
                else if (Mh & hb) hout = -1;
                Ph <<= 1;
                Mh <<= 1;
                if (hin < 0) Mh |= 1ull;
                else if (hin > 0) Ph |= 1ull;
                Pv[b] = Mh | ~(Xv | Ph);
                Mv[b] = Ph & Xv;
                hin = hout;
            }
        }
        score += hin;
        if (score - (left_after + cnt - 1 - t) > thr) return true;
    }
    return false;
}

/* Global Levenshtein distance computed by the whole wave: lanes fetch text bytes and their Peq
 * words in parallel (64 columns per round), then every lane runs the identical recurrence on
 * v_readlane-broadcast words -- no serial chain of dependent memory loads, result wave-uniform.
 * BYTE(j) yields text byte j.  Exact when the distance is <= thr, otherwise some value > thr. */
/* MODE (see k_trim_ends): 0 = any pattern, 1 = the caller guarantees m <= 32, 2 = m <= 64 */
template <int MODE, int PW, class ByteFn>
__device__ __forceinline__ int lev_wave_core(const uint64_t (*__restrict__ peq)[PW], int shift, int m, int n,
                                             int thr, ByteFn&& BYTE) {
    if (m == 0) return n;
    if (n == 0) return m;
    const int W = (m + 63) >> 6;
    const int lane = lane_id();
    int score = m;
    if (MODE == 1 || m <= 32) { /* MODE 1: the multi-word code below is not even compiled */
        u32 Pv = ~0u, Mv = 0;
        const u32 top = 1u << (m - 1);
        for (int j0 = 0; j0 < n; j0 += 64) {
            u64 eq = 0;
            if (j0 + lane < n) eq = peq_word(peq, (int)BYTE(j0 + lane), shift, 0);
            const WaveVals64 pub = wave_publish(eq);
            const int cnt = min(64, n - j0);
            if (lev_round32(pub, cnt, n - j0 - cnt, Pv, Mv, score, top, thr)) return thr + 1;
        }
        return score;
    }
    constexpr int MW = MODE == 2 ? 1 : PEQ_WORDS; /* 64-column words the recurrence may need */
    u64 Pv[MW], Mv[MW];
#pragma unroll
    for (int b = 0; b < MW; b++) {
        Pv[b] = ~0ull;
        Mv[b] = 0;
    }
    const u64 last_top = 1ull << ((m - 1) & 63);
    for (int j0 = 0; j0 < n; j0 += 64) {
        u64 eqw[MW] = {};
        if (j0 + lane < n) {
            const int c = (int)BYTE(j0 + lane);
#pragma unroll
            for (int b = 0; b < MW; b++)
                if (b < W) eqw[b] = peq_word(peq, c, shift, b);
        }
        WaveVals64 pub[MW];
#pragma unroll
        for (int b = 0; b < MW; b++)
            if (b < W) pub[b] = wave_publish(eqw[b]);
        const int cnt = min(64, n - j0);
        if (lev_round64(pub, W, cnt, n - j0 - cnt, Pv, Mv, score, last_top, thr)) return thr + 1;
    }
    return score;
}
__device__ __forceinline__ int lev_wave(const uint64_t (*__restrict__ peq)[PEQ_WORDS], int shift, int m,
                                        const u8* __restrict__ text, int n, int thr) {
    return lev_wave_core<0>(peq, shift, m, n, thr, [&](int j) { return (u32)text[j]; });
}

/* run f() on lane 0 only and hand its int result to every lane */
#define FPL_LANE0_INT(expr) readlane_i32((lane_id() == 0) ? (expr) : 0, 0)

/* =========================================================================================
 * k_trim_ends
 * ======================================================================================= */

/* One array of a read (bases or qualities) whose first and last TRIM_WIN bytes sit in LDS (stage_ends): the sequential
 * scans of trimAndCut / polyX almost always end inside those windows, so a read costs ONE trip to memory instead of one
 * per scan round.  A round takes the LDS copy when its whole index range [lo, hi] lies in a window (a wave-uniform
 * test), the global array otherwise. */
constexpr int TRIM_WIN = 256; /* FPL_END_WINDOW rounded up, plus room for the few bases trimAndCut / polyX usually take */
struct EndsView {
    const u8* g;    /* the read in global memory */
    const u8* head; /* LDS: bytes [0, TRIM_WIN) */
    const u8* tail; /* LDS: bytes [tail0, tail0 + TRIM_WIN), tail0 = max(0, l - TRIM_WIN) */
    int tail0;
    __device__ __forceinline__ bool in_head(int lo, int hi) const { return lo >= 0 && hi < TRIM_WIN; }
    __device__ __forceinline__ bool in_tail(int lo, int hi) const { return lo >= tail0 && hi < tail0 + TRIM_WIN; }
};

/* Filter::trimAndCut, src/filter.cpp:130-232.  Returns false when the reference returns NULL.
 * [s,e) is the surviving window in coordinates of the original read (length l). */
__device__ inline bool trim_and_cut_wave(const EndsView& vs, const EndsView& vq, int l,
                                         const DevConfig* __restrict__ cfg, int& s_out, int& e_out) {
    const int lane = lane_id();
    const u8* __restrict__ sq = vs.g;
    const u8* __restrict__ ql = vq.g;
    int front = cfg->trim_front, tail = cfg->trim_tail;
    s_out = 0;
    e_out = l;
    if (front == 0 && tail == 0 && !cfg->cut_front && !cfg->cut_tail) return true; /* :133-134 */
    int rlen = l - front - tail;
    if (rlen < 0) return false; /* :137-139 */
    if (!cfg->cut_front && !cfg->cut_tail) { /* :141-151 */
        s_out = front;
        e_out = front + rlen;
        return true;
    }
    if (cfg->cut_front) { /* :159-189 */
        const int w = cfg->cut_front_w, thr = cfg->cut_front_thr;
        if (l - front - tail - w <= 0) return false;
        const int lim = l - tail - w; /* loop runs while s < lim */
        int s = lim;                  /* value of s when the loop ends without a break */
        for (int s0 = front; s0 < lim; s0 += 64) {
            const int sc = s0 + lane;
            int tot = 0;
            if (vq.in_head(s0, min(s0 + 63, lim - 1) + w - 1)) {
                if (sc < lim)
                    for (int i = 0; i < w; i++) tot += vq.head[sc + i];
            } else if (sc < lim) {
                for (int i = 0; i < w; i++) tot += ql[sc + i];
            }
            const u64 m = wave_ballot(sc < lim && tot >= thr);
            if (m) {
                s = s0 + __ffsll(m) - 1;
                break;
            }
        }
        if (s > 0) s = s + w - 1;
        for (;;) { /* while (s < l && seq[s] == 'N') s++ */
            const int p = s + lane;
            bool isn = false;
            if (vs.in_head(s, min(s + 63, l - 1))) isn = p < l && vs.head[p] == 'N';
            else if (p < l) isn = sq[p] == 'N';
            const u64 stop = wave_ballot(!isn);
            if (stop) {
                s += __ffsll(stop) - 1;
                break;
            }
            s += 64;
        }
        front = s;
        rlen = l - front - tail;
    }
    if (cfg->cut_tail) { /* :191-219 */
        const int w = cfg->cut_tail_w, thr = cfg->cut_tail_thr;
        if (l - front - tail - w <= 0) return false;
        const int tstart = l - tail - 1, tlow = front + w; /* loop runs while t >= tlow */
        int t = tlow - 1;                                  /* value of t when the loop ends without a break */
        for (int t0 = tstart; t0 >= tlow; t0 -= 64) {
            const int tc = t0 - lane;
            int tot = 0;
            if (vq.in_tail(max(t0 - 63, tlow) - (w - 1), t0)) {
                if (tc >= tlow)
                    for (int i = 0; i < w; i++) tot += vq.tail[tc - i - vq.tail0];
            } else if (tc >= tlow) {
                for (int i = 0; i < w; i++) tot += ql[tc - i];
            }
            const u64 m = wave_ballot(tc >= tlow && tot >= thr);
            if (m) {
                t = t0 - (__ffsll(m) - 1);
                break;
            }
        }
        if (t < l - 1) t = t - w + 1;
        for (;;) { /* while (t >= 0 && seq[t] == 'N') t-- */
            const int p = t - lane;
            bool isn = false;
            if (vs.in_tail(max(t - 63, 0), t)) isn = p >= 0 && vs.tail[p - vs.tail0] == 'N';
            else if (p >= 0) isn = sq[p] == 'N';
            const u64 stop = wave_ballot(!isn);
            if (stop) {
                t -= __ffsll(stop) - 1;
                break;
            }
            t -= 64;
        }
        rlen = t - front + 1;
    }
    if (rlen <= 0 || front >= l - 1) return false; /* :221-222 */
    s_out = front;
    e_out = front + rlen;
    return true;
}

/* PolyX::trimPolyX, src/polyx.cpp:11-78, on the window [s0, s0 + rlen) of the read.  Returns the new length
 * and, when a polyX was cut, poly (0..3 = A,T,C,G) and the number of bases trimmed. */
__device__ inline int trim_polyx_wave(const EndsView& vs, int s0, int rlen, int compareReq, int& poly_out, int& trimmed_out) {
    const int lane = lane_id();
    const u8* __restrict__ r = vs.g + s0;
    poly_out = -1;
    trimmed_out = 0;
    int carry[4] = {0, 0, 0, 0};
    int P = rlen; /* value of pos when the scan ends without a break */
    int cnt[4] = {0, 0, 0, 0};
    bool broke = false;
    for (int p0 = 0; p0 < rlen; p0 += 64) {
        const int pos = p0 + lane;
        const bool valid = pos < rlen;
        u32 oh = 0; /* one-hot, one byte per base: A | T<<8 | C<<16 | G<<24 */
        {
            /* this round looks at read bytes s0 + rlen - 1 - (p0 + 63) .. s0 + rlen - 1 - p0 */
            const int hi = s0 + rlen - 1 - p0;
            u8 c = 0;
            if (vs.in_tail(max(hi - 63, s0), hi)) {
                if (valid) c = vs.tail[hi - lane - vs.tail0];
            } else if (valid) {
                c = r[rlen - pos - 1];
            }
            if (valid) {
                if (c == 'A') oh = 0x00000001u;
                else if (c == 'T') oh = 0x00000100u;
                else if (c == 'C') oh = 0x00010000u;
                else if (c == 'G') oh = 0x01000000u;
                else if (c == 'N') oh = 0x01010101u;
            }
        }
        const u32 sc = wave_scan_incl_u32(oh); /* per-byte counts <= 64 */
        const int nA = carry[0] + (int)(sc & 0xFF), nT = carry[1] + (int)((sc >> 8) & 0xFF);
        const int nC = carry[2] + (int)((sc >> 16) & 0xFF), nG = carry[3] + (int)(sc >> 24);
        const int cmp = pos + 1;
        const int allowed = min(5, cmp / 8);
        const bool need = (cmp - nA > allowed) && (cmp - nT > allowed) && (cmp - nC > allowed) && (cmp - nG > allowed);
        const bool brk = valid && need && (pos >= 8 || pos + 1 >= compareReq - 1);
        const u64 m = wave_ballot(brk);
        const int src = m ? (__ffsll(m) - 1) : 63;
        /* counts at the break lane, or the running totals when the round ends without one */
        cnt[0] = readlane_i32(nA, src);
        cnt[1] = readlane_i32(nT, src);
        cnt[2] = readlane_i32(nC, src);
        cnt[3] = readlane_i32(nG, src);
        if (m) {
            P = p0 + src;
            broke = true;
            break;
        }
        carry[0] = cnt[0];
        carry[1] = cnt[1];
        carry[2] = cnt[2];
        carry[3] = cnt[3];
    }
    (void)broke;
    if (P + 1 < compareReq) return rlen; /* :57 */
    int poly = 0, maxc = -1;
#pragma unroll
    for (int b = 0; b < 4; b++)
        if (cnt[b] > maxc) {
            maxc = cnt[b];
            poly = b;
        }
    const u8 polyBase = poly == 0 ? 'A' : (poly == 1 ? 'T' : (poly == 2 ? 'C' : 'G'));
    /* :71  walk pos down from P until r[rlen-pos-1] == polyBase  <=>  first index >= max(0,rlen-P-1)
       holding polyBase; index -1 (P == rlen) never matches; pos = -1 when nothing matches */
    int idx0 = rlen - P - 1;
    if (idx0 < 0) idx0 = 0;
    int found = -1;
    for (int i0 = idx0; i0 < rlen; i0 += 64) {
        const int i = i0 + lane;
        bool hit = false;
        if (vs.in_tail(s0 + i0, s0 + min(i0 + 63, rlen - 1))) hit = i < rlen && vs.tail[s0 + i - vs.tail0] == polyBase;
        else if (i < rlen) hit = r[i] == polyBase;
        const u64 m = wave_ballot(hit);
        if (m) {
            found = i0 + __ffsll(m) - 1;
            break;
        }
    }
    const int pos = found >= 0 ? rlen - found - 1 : -1;
    poly_out = poly;
    trimmed_out = pos + 1;
    return rlen - pos - 1; /* Read::resize: a no-op when pos == -1 */
}

__device__ __forceinline__ int hamming_bytes(const u8* __restrict__ r, const u8* __restrict__ a, int alen) {
    int mm = 0;
    for (int i = 0; i < alen; i++) mm += (r[i] != a[i]);
    return mm;
}

/* Byte access to r1 for the end trims.  LDSWIN: an LDS copy of a <= 200-byte window of the read
 * (r1 byte j lives at window byte j - bias); reads are aligned ds_read_b32 + v_alignbyte, because
 * byte-granular or unaligned wide LDS reads stall the LDS pipe.  Otherwise: the read itself in
 * global memory (FASTA adapter chain, adapters longer than the window), byte loads. */
template <bool LDSWIN>
struct Win {
    const u8* r;  /* global: first base of r1 */
    const u32* w; /* LDS window */
    int bias;
    int rlen;
    const u32* w4 = nullptr; /* LDS: the same window as one-hot nibbles, 8 bases per dword (stage_window) */
    __device__ __forceinline__ u32 byte(int j) const {
        if (LDSWIN) {
            const int b = j - bias;
            return (w[b >> 2] >> (8 * (b & 3))) & 0xFFu;
        }
        return r[j];
    }
    /* d[k] = bytes [j + 4k, j + 4k + 4) for k < N; bytes at or beyond rlen are unspecified (LDS) / 0 (global) */
    template <int N>
    __device__ __forceinline__ void get(int j, u32 (&d)[N]) const {
        if (LDSWIN) {
            const int b = j - bias;
            const u32* p = w + (b >> 2);
            const u32 sh = (u32)b & 3u;
            u32 x[N + 1];
#pragma unroll
            for (int k = 0; k <= N; k++) x[k] = p[k];
#pragma unroll
            for (int k = 0; k < N; k++) d[k] = alignbyte(x[k + 1], x[k], sh);
        } else {
#pragma unroll
            for (int k = 0; k < N; k++) {
                u32 v = 0;
#pragma unroll
                for (int t = 0; t < 4; t++)
                    if (j + 4 * k + t < rlen) v |= (u32)r[j + 4 * k + t] << (8 * t);
                d[k] = v;
            }
        }
    }
};

/* Hamming distance between the adapter and r1[p, p + alen): 4 bytes per XOR + zero-byte test */
template <bool LDSWIN>
__device__ __forceinline__ int hamming_win(const Win<LDSWIN>& win, int p, const DevAdapter* __restrict__ ad) {
    const int alen = ad->len;
    const u32* __restrict__ ad32 = (const u32*)ad->seq;
    int mm = 0;
    for (int i0 = 0; i0 < alen; i0 += 16) {
        u32 d[4];
        win.template get<4>(p + i0, d);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int rem = alen - (i0 + 4 * k); /* wave-uniform */
            if (rem > 0) {
                u32 x = d[k] ^ ad32[(i0 >> 2) + k];
                if (rem < 4) x &= (1u << (8 * rem)) - 1u;
                const u32 z = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;
                mm += (int)popc32(z & 0x80808080u);
            }
        }
    }
    return mm;
}

/* The same for an adapter of <= 32 bases whose 8 dwords the caller keeps in registers (one scalar load per
 * trim instead of one per dword and position round) */
template <bool LDSWIN>
__device__ __forceinline__ int hamming_win32(const Win<LDSWIN>& win, int p, const u32 (&adw)[8], int alen) {
    u32 d[8];
    win.template get<8>(p, d);
    int mm = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int rem = alen - 4 * k; /* wave-uniform */
        if (rem > 0) {
            u32 x = d[k] ^ adw[k];
            if (rem < 4) x &= (1u << (8 * rem)) - 1u;
            const u32 z = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;
            mm += (int)popc32(z & 0x80808080u);
        }
    }
    return mm;
}

/* The same on one-hot nibbles (adapters of A / C / G / T only, DevAdapter::onehot): matches = popcount(window & adapter),
 * eight bases per dword.  NW = dwords that hold the adapter (4: <= 32 bases, 8: <= 64); the bits behind the adapter are
 * zero, so no tail mask is needed.  Anything that is not exactly A, C, G or T in the read has a zero nibble: a mismatch,
 * as in the byte comparison (the adapter has no such byte). */
template <int NW>
__device__ __forceinline__ int hamming_onehot(const u32* __restrict__ w4, int b, const u32 (&ad1h)[NW], int alen) {
    const u32* q = w4 + (b >> 3);
    const u32 sh = ((u32)b & 7u) * 4u;
    u32 x[NW + 1];
#pragma unroll
    for (int k = 0; k <= NW; k++) x[k] = q[k];
    u32 matches = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) matches += popc32(alignbit(x[k + 1], x[k], sh) & ad1h[k]);
    return alen - (int)matches;
}

/* 16-column Myers run on r1[p, p + n), n <= 16 */
/* FULL: m == n == 16 (every adapter of >= 16 bases), known at compile time: no per-column tests, one
 * straight-line block that the scheduler can interleave with the other chains of the lane */
template <bool FULL, bool LDSWIN, class PT>
__device__ __forceinline__ int lev16_win(const Win<LDSWIN>& win, int p, const PT* __restrict__ peq, int m_, int n_) {
    const int m = FULL ? 16 : m_, n = FULL ? 16 : n_;
    if (m == 0) return n;
    u32 d[4];
    win.template get<4>(p, d);
    u32 eq[16];
#pragma unroll
    for (int j = 0; j < 16; j++) eq[j] = (FULL || j < n) ? peq[(d[j >> 2] >> (8 * (j & 3))) & 0xFFu] : 0u;
    u32 Pv = ~0u, Mv = 0;
    int score = m;
    const u32 top = 1u << (m - 1);
#pragma unroll
    for (int j = 0; j < 16; j++) {
        if (FULL || j < n) { /* wave-uniform */
            const u32 Eq = eq[j];
            const u32 Xv = Eq | Mv;
            const u32 Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
            u32 Ph = Mv | ~(Xh | Pv);
            u32 Mh = Pv & Xh;
            score += (Ph & top) ? 1 : ((Mh & top) ? -1 : 0);
            Ph = (Ph << 1) | 1u;
            Mh <<= 1;
            Pv = Mh | ~(Xv | Ph);
            Mv = Ph & Xv;
        }
    }
    return score;
}

/* lev_wave with the text taken from a Win */
template <int MODE, bool LDSWIN, int PW>
__device__ __forceinline__ int lev_wave_win(const uint64_t (*__restrict__ peq)[PW], int shift, int m,
                                            const Win<LDSWIN>& win, int p, int n, int thr) {
    return lev_wave_core<MODE>(peq, shift, m, n, thr, [&](int j) { return win.byte(p + j); });
}

/* AdapterTrimmer::trimBySequenceStart, src/adaptertrimmer.cpp:168-236 (searchAdapter in its
 * asRightAsPossible mode, :109-131, inlined).  rd = first base of the original read; [s,e) is
 * updated; returns the reference's return value; keylen = cmplen handed to addAdapterTrimmed. */
/* r = first base of r1 -- the global read or a copy of its first 200 bytes in LDS; peq16 / peqf = the
 * adapter's Myers tables, global or LDS copies. */
template <int MODE, bool LDSWIN, class PT, int PW>
__device__ __forceinline__ int trim_start_wave(const Win<LDSWIN>& win, int& s, int& e, const DevAdapter* __restrict__ ad,
                                               const PT* __restrict__ peq16,
                                               const uint64_t (*__restrict__ peqf)[PW],
                                               const DevConfig* __restrict__ cfg, int& keylen,
                                               bool may_full = true, bool may_part = true, int hint_lo = 0, int hint_hi = 0x3fffffff) {
    /* may_full / may_part (wave-uniform): false when the caller has PROOF that the whole-adapter search / the partial-
       pattern search finds nothing here (fasta_may_trim32) -- the search, and the Levenshtein run on its best window, are
       then left out.  [hint_lo, hint_hi] (wave-uniform): the only positions the partial-pattern search can accept
       (fasta_hint_range); a range of <= 64 positions is ONE round of windows instead of three */
    const int lane = lane_id();
    const int rlen = e - s;
    keylen = 0;
    if (rlen < FPL_PATTERN_LEN) return 0;
    const int alen = ad->len, plen = ad->plen, ext = cfg->ext;
    const int thrA = cfg->thr[alen];
    /* the adapter in scalar registers: one-hot nibbles for the reduced instantiations (A / C / G / T only), else its
       first 32 bytes (zero padded) */
    constexpr int NW1H = MODE == 1 ? 4 : 8;
    u32 adw[8], ad1h[NW1H];
#pragma unroll
    for (int k = 0; k < 8; k++) adw[k] = (MODE == 0 || !FPL_OPT_ONEHOT) ? uniform_u32(((const u32*)ad->seq)[k]) : 0u;
#pragma unroll
    for (int k = 0; k < NW1H; k++) ad1h[k] = (MODE != 0 && FPL_OPT_ONEHOT) ? uniform_u32(ad->onehot[k]) : 0u;
    int mpos = -1;
    const int searchEnd = min(rlen, FPL_END_WINDOW);
    if (may_full && alen <= rlen && searchEnd > alen) {
        const int npos = searchEnd - alen + 1; /* p = searchEnd-alen .. 0 */
        int hit = -1;
        u64 best = ~0ull;
        for (int p0 = 0; p0 < npos; p0 += 64) {
            const int p = p0 + lane;
            int mm = 0x7fffffff;
            if (p < npos && !FPL_DBG(cfg->dbg, 256)) {
                if (MODE != 0 && FPL_OPT_ONEHOT) mm = hamming_onehot<NW1H>(win.w4, p - win.bias, ad1h, alen);
                else mm = (MODE == 1 || alen <= 32) ? hamming_win32(win, p, adw, alen) : hamming_win(win, p, ad);
            }
            const u64 m = wave_ballot(p < npos && mm <= thrA);
            if (m) hit = p0 + 63 - __clzll(m); /* rightmost hit so far */
            if (p < npos) {
                const u64 k = ((u64)(u32)mm << 32) | (u32)p; /* ties: leftmost (descending scan, <=) */
                best = k < best ? k : best;
            }
        }
        if (hit >= 0) mpos = hit; /* returned at once, no edit-distance check (:121-124) */
        else {
            best = wave_min_u64(best);
            if (best != ~0ull) {
                const int pos = (int)(u32)best;
                const int ed = FPL_DBG(cfg->dbg, 64) ? 999 : lev_wave_win<MODE>(peqf, 0, alen, win, pos, alen, thrA);
                if (ed <= thrA) mpos = pos;
            }
        }
    }
    if (mpos >= 0) { /* :185-193 */
        mpos = min(mpos + ext, rlen - alen);
        keylen = alen;
        s += min(rlen - 1, mpos + alen); /* Read::trimFront */
        return mpos + alen;
    }
    /* partial match of the last plen adapter bases, :202-216: first minimum among ed <= thr */
    const int lim = may_part ? min(rlen - plen, FPL_END_WINDOW - plen) : 0;
    const int thrP = cfg->thr[plen];
    u64 best = ~0ull;
    /* lim <= 184: three positions per lane, evaluated as three independent chains in one straight-line
       block (positions past lim are clamped for the loads and masked afterwards) */
    if (MODE == 2 && lim > 0 && hint_hi - hint_lo < 64) { /* (MODE 2: plen == 16) */
        const int p = hint_lo + lane;
        const int ed = lev16_win<true>(win, min(p, lim - 1), peq16, 16, 16);
        if (p <= hint_hi && p < lim && ed <= thrP) best = ((u64)(u32)ed << 32) | (u32)p;
    } else
    for (int p0 = 0; p0 < lim && !FPL_DBG(cfg->dbg, 128); p0 += 192) {
        int pp[3], ed[3];
#pragma unroll
        for (int u = 0; u < 3; u++) pp[u] = p0 + 64 * u + lane;
        if (MODE != 0 || plen == 16) { /* (MODE 1 / 2: adapters of >= 16 bases, so plen == 16) */
#pragma unroll
            for (int u = 0; u < 3; u++) ed[u] = lev16_win<true>(win, min(pp[u], lim - 1), peq16, 16, 16);
        } else {
#pragma unroll
            for (int u = 0; u < 3; u++) ed[u] = lev16_win<false>(win, min(pp[u], lim - 1), peq16, plen, plen);
        }
#pragma unroll
        for (int u = 0; u < 3; u++)
            if (pp[u] < lim && ed[u] <= thrP) {
                const u64 k = ((u64)(u32)ed[u] << 32) | (u32)pp[u];
                best = k < best ? k : best;
            }
    }
    best = wave_min_u64(best);
    if (best != ~0ull) { /* :218-233 */
        int pos = (int)(u32)best;
        const int cmplen = min(pos + plen, alen);
        const int ed = lev_wave_win<MODE>(peqf, alen - cmplen, cmplen, win, pos + plen - cmplen, cmplen, cfg->thr[cmplen]);
        if (ed <= cfg->thr[cmplen]) {
            pos = min(pos + ext, rlen - alen);
            keylen = cmplen;
            const int n = min(rlen - 1, pos + plen); /* Read::trimFront; negative erases everything */
            if (n < 0) s = e;
            else s += n;
            return pos + plen;
        }
    }
    return 0;
}

/* AdapterTrimmer::trimBySequenceEnd, src/adaptertrimmer.cpp:238-302 (searchAdapter in its
 * asLeftAsPossible mode, :84-107, inlined). */
/* r = first base of r1 as an address: only its last 200 bytes are dereferenced, so r may point
 * 200 - rlen bytes in front of an LDS copy of that tail. */
template <int MODE, bool LDSWIN, class PT, int PW>
__device__ __forceinline__ int trim_end_wave(const Win<LDSWIN>& win, int& s, int& e, const DevAdapter* __restrict__ ad,
                                             const PT* __restrict__ peq16,
                                             const uint64_t (*__restrict__ peqf)[PW],
                                             const DevConfig* __restrict__ cfg, int& keylen,
                                             bool may_full = true, bool may_part = true, int hint_lo = 0,
                                             int hint_hi = 0x3fffffff) { /* (see trim_start_wave) */
    const int lane = lane_id();
    const int rlen = e - s;
    keylen = 0;
    if (rlen < FPL_PATTERN_LEN) return 0;
    const int alen = ad->len, plen = ad->plen, ext = cfg->ext;
    const int thrA = cfg->thr[alen];
    constexpr int NW1H = MODE == 1 ? 4 : 8; /* (see trim_start_wave) */
    u32 adw[8], ad1h[NW1H];
#pragma unroll
    for (int k = 0; k < 8; k++) adw[k] = (MODE == 0 || !FPL_OPT_ONEHOT) ? uniform_u32(((const u32*)ad->seq)[k]) : 0u;
#pragma unroll
    for (int k = 0; k < NW1H; k++) ad1h[k] = (MODE != 0 && FPL_OPT_ONEHOT) ? uniform_u32(ad->onehot[k]) : 0u;
    const int ss = max(0, rlen - FPL_END_WINDOW);
    int mpos = -1;
    if (may_full && ss + alen <= rlen) {
        const int pend = rlen - alen; /* p in [ss, pend) : the last position is never tested */
        int hit = -1;
        u64 best = ~0ull;
        for (int p0 = ss; p0 < pend; p0 += 64) {
            const int p = p0 + lane;
            int mm = 0x7fffffff;
            if (p < pend) {
                if (MODE != 0 && FPL_OPT_ONEHOT) mm = hamming_onehot<NW1H>(win.w4, p - win.bias, ad1h, alen);
                else mm = (MODE == 1 || alen <= 32) ? hamming_win32(win, p, adw, alen) : hamming_win(win, p, ad);
            }
            const u64 m = wave_ballot(p < pend && mm <= thrA);
            if (m) {
                hit = p0 + __ffsll(m) - 1; /* leftmost hit, returned at once (:98-101) */
                break;
            }
            if (p < pend) {
                const u64 k = ((u64)(u32)mm << 32) | (u32)(0xFFFFFFFFu - (u32)p); /* ties: rightmost (<=) */
                best = k < best ? k : best;
            }
        }
        if (hit >= 0) mpos = hit;
        else {
            best = wave_min_u64(best);
            if (best != ~0ull) {
                const int pos = (int)(0xFFFFFFFFu - (u32)best);
                const int ed = lev_wave_win<MODE>(peqf, 0, alen, win, pos, alen, thrA);
                if (ed <= thrA) mpos = pos;
            }
        }
    }
    if (mpos >= 0) { /* :256-264 */
        mpos = max(0, mpos - ext);
        keylen = alen;
        e = s + mpos; /* Read::resize */
        return rlen - mpos;
    }
    /* partial match of the first plen adapter bases walking in from the tail, :273-286:
       qualifying positions in ascending p; stop at the first increase; ties take the later */
    const int lim = may_part ? min(rlen - plen, FPL_END_WINDOW - plen) : 0;
    const int thrP = cfg->thr[plen];
    int pos = -1, mined = -1;
    bool stop = false;
    int ed3[3];
    /* (hinted: the qualifying positions all lie in [hint_lo, hint_lo + 64): one round starting there -- the walk below only
       ever looks at qualifying positions, in ascending order) */
    const bool one_round = MODE == 2 && lim > 0 && hint_hi - hint_lo < 64;
    const int first = one_round ? hint_lo : 0, last = one_round ? min(lim, hint_hi + 1) : lim;
    if (one_round) {
        ed3[0] = lev16_win<true>(win, rlen - 16 - min(first + lane, lim - 1), peq16, 16, 16);
        ed3[1] = ed3[2] = 0x7fffffff;
    } else if (MODE != 0 || plen == 16) {
#pragma unroll
        for (int u = 0; u < 3; u++) /* lim <= 184 = three rounds; independent chains, one straight-line block */
            ed3[u] = lim > 0 ? lev16_win<true>(win, rlen - 16 - min(64 * u + lane, lim - 1), peq16, 16, 16) : 0x7fffffff;
    } else {
#pragma unroll
        for (int u = 0; u < 3; u++)
            ed3[u] = lim > 0 ? lev16_win<false>(win, rlen - plen - min(64 * u + lane, lim - 1), peq16, plen, plen) : 0x7fffffff;
    }
    for (int p0 = first; p0 < last && !stop; p0 += 64) {
        const int p = p0 + lane;
        /* (an adapter of fewer than 8 bases has up to 194 positions: the ones behind the three precomputed rounds
           are evaluated here; the reduced instantiations only see partial patterns of 16 columns, 184 positions) */
        int edr = p0 == first ? ed3[0] : (p0 == 64 ? ed3[1] : ed3[2]);
        if (MODE == 0 && p0 >= 192) edr = lev16_win<false>(win, rlen - plen - min(p0 + lane, lim - 1), peq16, plen, plen);
        const int ed = p < last ? edr : 0x7fffffff;
        u64 q = wave_ballot(p < last && ed <= thrP);
        while (q && !stop) {
            const int b = __ffsll(q) - 1;
            q &= q - 1;
            const int edb = readlane_i32(ed, b);
            if (pos < 0) {
                pos = p0 + b;
                mined = edb;
            } else if (edb > mined) {
                stop = true;
            } else {
                pos = p0 + b;
                mined = edb;
            }
        }
    }
    if (pos > 0) { /* :288 strict */
        const int cmplen = min(pos + plen, alen);
        const int ed = lev_wave_win<MODE>(peqf, 0, cmplen, win, rlen - plen - pos, cmplen, cfg->thr[cmplen]);
        if (ed <= cfg->thr[cmplen]) {
            pos = min(pos + ext, rlen - plen);
            keylen = cmplen;
            e = s + (rlen - plen - pos); /* Read::resize */
            return pos + plen;
        }
    }
    return 0;
}

/* LDS accumulator of one k_trim_ends block: FilterResult scalars + the key histogram of the
 * two command-line adapters (the FASTA slots go straight to global atomics). */
template <bool SLIM = false> /* SLIM: the chain-only instantiation, which trims with neither command-line adapter */
struct TrimBlockAcc {
    u64 fr[FPL_FR_LEN];
    u32 key[SLIM ? 1 : 2 * 2 * FPL_KEY_STRIDE];
};

/* LDS copies for the two command-line adapters: their 16-column Peq tables, their full Peq tables,
 * and per wave the first / last 200 bases of the read being trimmed.  Everything the 2 x 184 Myers
 * runs and the window Hamming scans touch is then an LDS read instead of a dependent global load. */
#ifndef FPL_TRIM_WAVES_PER_SIMD
#define FPL_TRIM_WAVES_PER_SIMD 7 /* 13.6 KB of LDS and <= 72 VGPRs per 4-wave block */
#endif
#ifndef FPL_TRIM_WAVES_PER_SIMD_SHORT
#define FPL_TRIM_WAVES_PER_SIMD_SHORT 7 /* (8 would cap it at 64 VGPRs: spills) */
#endif
template <int WAVES, bool SLIM = false> /* SLIM (chain only): no command-line adapter tables, no quality windows */
struct TrimLds {
    uint16_t peq16[SLIM ? 1 : 2][SLIM ? 1 : 256]; /* [0] = start adapter's peq16_start, [1] = end adapter's peq16_end (16 columns) */
    uint64_t peqf[SLIM ? 1 : 2][SLIM ? 1 : 256][1]; /* word 0 of the full Peq tables: adapters of <= 64 bases (longer ones use the global tables) */
    /* per wave: the first / last TRIM_WIN bytes of the read being trimmed (stage_ends), [0] bases at the head, [1] bases
       at the tail, [2] / [3] the qualities; + 8 zero dwords the dword-aligned reads of the last positions run into */
    u32 win[WAVES][SLIM ? 2 : 4][TRIM_WIN / 4 + 8];
    u32 win4[WAVES][2][TRIM_WIN / 8 + 8]; /* the two base windows as one-hot nibbles (+ what a shifted read runs into) */
    uint16_t peq16w[WAVES][256];   /* per wave: the 16-column Peq table of the FASTA adapter being tried */
};

/* copy bytes [from, from + n) of a read (n <= TRIM_WIN) into this wave's window, 4 bytes per lane */
/* ... and, when dst4 is given, the same bytes as one-hot nibbles (A 1, C 2, G 4, T 8, anything else 0), 16 bits per lane */
/* dstr (with dst4): per byte the row of the adapter filter's Peq table it selects -- its code when it is exactly A / C / T / G,
   else 4, the all-zero row (fasta_may_trim32 would otherwise work that out on the scalar unit, column by column, for every
   group of adapters and every refresh) */
__device__ __forceinline__ void stage_window(u32* __restrict__ dst, const u8* __restrict__ src, int n,
                                             const u8* __restrict__ seq_end, u32* __restrict__ dst4 = nullptr,
                                             u32* __restrict__ dstr = nullptr) {
    const int lane = lane_id();
    wave_sync();
    if (lane < TRIM_WIN / 4) {
        u32 w = (4 * lane < n) ? load4_guard(src + 4 * lane, seq_end) : 0u;
        if (4 * lane + 4 > n && 4 * lane < n) w &= (1u << (8 * (n - 4 * lane))) - 1u; /* bytes behind the window read as 0 */
        dst[lane] = w;
        if (dst4) {
            const u32 code = (w >> 1) & 0x03030303u;            /* A0 C1 T2 G3 */
            const u32 t = perm_lo(0x47544341u, code) ^ w;       /* zero byte <=> exactly that letter */
            const u32 ok = (~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t) >> 7) & 0x01010101u;
            const u32 nib = perm_lo(0x04080201u, code) & (ok * 15u);
            const u32 h = (nib | (nib >> 4)) & 0x00FF00FFu;
            ((uint16_t*)dst4)[lane] = (uint16_t)((h | (h >> 8)) & 0xFFFFu);
            if (dstr) dstr[lane] = (code & (ok * 3u)) | ((ok ^ 0x01010101u) << 2);
        }
    }
    if (dst4 && lane < 16) dst4[TRIM_WIN / 8 + (lane & 7)] = 0u; /* (what a shifted read of the last positions touches) */
    wave_sync();
}

/* The one trip to memory of a read in k_trim_ends: lane i fetches dword i of the first and of the last TRIM_WIN bytes of
 * the bases and of the qualities (bytes behind the read are zeroed in the base windows), the base windows also go out
 * as one-hot nibbles when the adapter scans want them.  tail0 = max(0, l - TRIM_WIN). */
template <bool ONEHOT>
__device__ __forceinline__ void stage_ends(u32* __restrict__ hs, u32* __restrict__ ts, u32* __restrict__ hq, u32* __restrict__ tq,
                                           u32* __restrict__ h4, u32* __restrict__ t4, const u8* __restrict__ sq,
                                           const u8* __restrict__ ql, int l, int tail0, const u8* __restrict__ seq_end,
                                           const u8* __restrict__ qual_end) {
    const int lane = lane_id();
    auto keep = [](u32 w, int n) -> u32 { return n >= 4 ? w : (n <= 0 ? 0u : (w & ((1u << (8 * n)) - 1u))); };
    const int hn = l - 4 * lane, tn = l - (tail0 + 4 * lane); /* bytes of the read at / behind this lane's dword */
    u32 a = 0, b = 0, c = 0, d = 0;
    if (hn > 0) {
        a = load4_guard(sq + 4 * lane, seq_end);
        c = load4_guard(ql + 4 * lane, qual_end);
    }
    if (tn > 0) {
        b = load4_guard(sq + tail0 + 4 * lane, seq_end);
        d = load4_guard(ql + tail0 + 4 * lane, qual_end);
    }
    a = keep(a, hn);
    b = keep(b, tn);
    auto onehot16 = [](u32 w) -> u32 {
        const u32 code = (w >> 1) & 0x03030303u;            /* A0 C1 T2 G3 */
        const u32 t = perm_lo(0x47544341u, code) ^ w;       /* zero byte <=> exactly that letter */
        const u32 ok = (~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t) >> 7) & 0x01010101u;
        const u32 nib = perm_lo(0x04080201u, code) & (ok * 15u);
        const u32 h = (nib | (nib >> 4)) & 0x00FF00FFu;
        return (h | (h >> 8)) & 0xFFFFu;
    };
    wave_sync();
    hs[lane] = a;
    ts[lane] = b;
    hq[lane] = c;
    tq[lane] = d;
    if (lane < 8) hs[TRIM_WIN / 4 + lane] = ts[TRIM_WIN / 4 + lane] = 0u;
    if (ONEHOT) {
        ((uint16_t*)h4)[lane] = (uint16_t)onehot16(a);
        ((uint16_t*)t4)[lane] = (uint16_t)onehot16(b);
        if (lane < 8) h4[TRIM_WIN / 8 + lane] = t4[TRIM_WIN / 8 + lane] = 0u;
    }
    wave_sync();
}

/* The FASTA chain (trimByMultiSequences, src/adaptertrimmer.cpp:42-57) tries every adapter at both ends of every read;
 * with 64 adapters nearly all of those 128 trims find nothing, at full price.  Lane = adapter: ONE pass of Myers' search
 * recurrence over the <= 200 window bytes (the same byte for every lane, each lane its own Peq words) gives, per adapter,
 * the smallest edit distance of (a) the whole adapter and (b) its 16-base partial pattern against ANY substring of the
 * window.  A trim needs a window position whose GLOBAL distance (full adapter: Hamming <= thr, or the candidate's edit
 * distance <= thr, :84-131; partial pattern: :202-216 / :273-286) is within the threshold, and no global distance is
 * smaller than the search distance at the window's last byte -- so an adapter whose two minima both stay above their
 * thresholds cannot trim this end, and the chain skips it.  Anything else gets the exact code, unchanged.
 * FastaPeqLds: the Peq words of a group of 64 adapters, [letter code | 4 = any other byte: zero][field][lane]. */
#ifdef FPL_EMU_FILTER_STATS
static unsigned long long g_filter_stats[8]; /* emulator only: mask refreshes, adapters flagged at the start / at the end, of those: whole-adapter flags, exact trims called, trims that moved r1 */
#endif
struct FastaPeqLds {
    u32 w[5][4][64]; /* field 0 / 1: whole adapter, low / high word; 2: last 16 bases; 3: first 16 bases */
};
/* this lane's adapter into the table (a_ok: the lane has one) */
__device__ __forceinline__ void fasta_peq_store(FastaPeqLds* __restrict__ t, const DevAdapter* __restrict__ ad, bool a_ok) {
    const int lane = lane_id();
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const uint64_t f = a_ok ? ad->peq4_full[c] : 0ull;
        t->w[c][0][lane] = (u32)f;
        t->w[c][1][lane] = (u32)(f >> 32);
        t->w[c][2][lane] = a_ok ? ad->peq4_s16[c] : 0u;
        t->w[c][3][lane] = a_ok ? ad->peq4_e16[c] : 0u;
    }
#pragma unroll
    for (int f = 0; f < 4; f++) t->w[4][f][lane] = 0u;
#if FPL_OPT_FASTAFILTER == 2
#pragma unroll
    for (int c = 0; c < 4; c++) { /* (the cheaper form only needs fields 2 and 3) */
        t->w[c][2][lane] = a_ok ? ad->peq4_s32r[c] : 0u;
        t->w[c][3][lane] = a_ok ? ad->peq4_e32[c] : 0u;
    }
#endif
}
/* The cheaper form of the test below (FPL_OPT_FASTAFILTER 2): ONE 32-column semi-global Myers run per end with two score
   taps.  If the whole adapter matches somewhere within thrA edits, so does any prefix or suffix of it, and if the 16-base
   partial pattern matches a window within thrP edits, the search variant finds a substring at least that close: the end
   trim searches for the adapter's first min(32, len) bases (tap at their last column) and reads the partial pattern's
   score off column 15 of the same run; the start trim does the same with the adapter's last bases REVERSED over the
   window read backwards (an edit script read backwards is an edit script), so that its partial pattern -- the adapter's
   last 16 bases -- is again the first 16 columns.  18 vector instructions per window byte instead of 49; what it lets
   through that the two-run form would have stopped only costs an exact trim that finds nothing. */
template <bool START>
__device__ __forceinline__ u32 fasta_may_trim32(const FastaPeqLds* __restrict__ t, const u8* __restrict__ rowc, int boff, int n,
                                                 int alen, int thrA, int thrP, bool a_ok) { /* rowc: stage_window's dstr */
    const int lane = lane_id();
    const int m = min(alen, 32);
    u32 Pv = m >= 32 ? ~0u : ((1u << m) - 1u), Mv = 0;
    int scF = m, scP = 16, bestF = m;
    u32 blocks = 0; /* bit b: the partial pattern's score got down to thrP somewhere in columns [32 b, 32 b + 32) */
    const u32 topF = (u32)(m - 1);
    for (int j0 = 0; j0 < n; j0 += 32) { /* (n <= 200: seven blocks at most) */
        int blkP = 16 + 32;
        const int j1 = min(n, j0 + 32);
        /* eight columns at a time: their window bytes (the same byte for every lane) and then their Peq words are fetched
           together -- one byte, one word and two trips to LDS per column kept the wave waiting most of the time */
        for (int jj = j0; jj < j1; jj += 8) {
            u32 cb[8], Eqs[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int j = min(jj + u, n - 1);
                cb[u] = (u32)rowc[(START ? n - 1 - j : j) + boff];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) Eqs[u] = t->w[uniform_u32(cb[u])][START ? 2 : 3][lane];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (jj + u < j1) { /* wave-uniform */
                    const u32 Eq = Eqs[u];
                    const u32 Xv = Eq | Mv;
                    const u32 Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
                    u32 Ph = Mv | ~(Xh | Pv);
                    u32 Mh = Pv & Xh;
                    scF += (int)((Ph >> topF) & 1u) - (int)((Mh >> topF) & 1u);
                    scP += (int)((Ph >> 15) & 1u) - (int)((Mh >> 15) & 1u);
                    Ph <<= 1; /* (no "| 1": a match may start anywhere) */
                    Mh <<= 1;
                    Pv = Mh | ~(Xv | Ph);
                    Mv = Ph & Xv;
                    bestF = min(bestF, scF);
                    blkP = min(blkP, scP);
                }
            }
        }
        blocks |= (blkP <= thrP ? 1u : 0u) << (j0 >> 5);
    }
    /* A partial-pattern hit only trims when its confirmation passes (src/adaptertrimmer.cpp:218-233 / :288-299): the adapter's
       last / first cmplen = min(p + 16, alen) bases against the read around the hit, within thr[cmplen].  For a hit at p >= alen - 16
       that is the WHOLE adapter within thrA -- which the other tap of this very run would have seen (a whole adapter within thrA
       has its first / last 32 columns within thrA of some substring).  So without that flag only the hits next to the read's end
       count: p < alen - 16, i.e. columns j = n - 1 - p >= n - alen + 16, i.e. the blocks from (n - alen + 16) / 32 on (the few
       columns of that block in front of the bound ride along).  Random 16-mers within 4 edits of SOME stretch of 200 bases are
       what 8 of the 10 flags per read were (emulator, c5-like reads); nearly all of them sit further in. */
    const bool fullF = bestF <= thrA;
    bool partF = blocks != 0;
    if (FPL_OPT_FASTANEAR && !fullF) {
        /* (only WHETHER the search runs is decided here: when it does it must see every hit -- the walk over the hits picks its
           position among all of them, and one further in can take the place of one next to the end) */
        const int jn = n - alen + 16;
        partF = (jn <= 0 ? blocks : (blocks & (~0u << (jn >> 5)))) != 0;
    }
    /* bit 0: the whole adapter may match, bit 1: its partial pattern; bits 8..: where (blocks of 32 columns) */
    return a_ok ? ((fullF ? 1u : 0u) | (partF ? 2u : 0u) | ((partF ? blocks : 0u) << 8)) : 0u;
}
/* fasta_may_trim32 with fewer instructions per window byte (22 -> 18), for adapters of 23 bases and more (every lane of the wave):
   the same verdicts.
   - The two scores live in ONE register as 7-bit fields, each with a bias that puts "score > threshold" into the field's top bit
     (score + 63 - thr: 0 .. 127 for scores up to 64): the partial pattern's at bits 0..6, the adapter's from bit topF - 15 (>= 7) on,
     so that one shift by 15 and one AND with a per-lane mask bring both taps of Ph (of Mh) to their fields' low bits; a column
     adds the one and subtracts the other (no field ever leaves 0 .. 127: a score moves by one and is an edit distance).  ANDing
     the register over the columns keeps a field's top bit exactly when the score stayed above its threshold in all of them:
     seven instructions where two bit-field extracts, an add / subtract and a minimum per score took nine.
   - Eight columns whose bytes all exist are stepped without a test per column, their window bytes come from one address
     register with constant offsets, and a byte's table row goes into the address arithmetic as the (wave-uniform) vector
     value it is loaded as instead of through a scalar register. */
#ifndef FPL_OPT_ONEFP
#define FPL_OPT_ONEFP 1 /* k_trim_ends<.., 2, true>: one filter table per block when there is one group of FASTA adapters */
#endif
#ifndef FPL_OPT_FILTPACK
#define FPL_OPT_FILTPACK 1
#endif
template <bool START>
__device__ __forceinline__ u32 fasta_may_trim32p(const FastaPeqLds* __restrict__ t, const u8* __restrict__ rowc, int boff, int n,
                                                  int alen, int thrA, int thrP, bool a_ok) {
    const int lane = lane_id();
    const int m = min(alen, 32); /* >= 23 */
    u32 Pv = m >= 32 ? ~0u : ((1u << m) - 1u), Mv = 0;
    const u32 sF = (u32)(m - 1) - 15u;       /* the adapter's field: bits sF .. sF + 6 */
    const u32 MK = 1u | (1u << sF);          /* the two taps behind the shift by 15 */
    u32 sc = (u32)(16 + 63 - thrP) | ((u32)(m + 63 - thrA) << sF);
    u32 acc = sc | 0x7Fu; /* (the adapter's score before the first column counts, the partial pattern's does not: as fasta_may_trim32) */
    u32 blocks = 0;
    const u32* const wf = &t->w[0][START ? 2 : 3][lane];
/* (the recurrence in nine instructions: Xh = ((Eq & Pv) + Pv) ^ Pv | Eq and Xv = Eq | Mv never exist on their own -- with s = (Eq & Pv) + Pv,
   Ph = Mv | ~(s | Pv | Eq), Mh = Pv & ((s ^ Pv) | Eq), and behind the shift Pv' = (Mh << 1) | ~(Eq | Mv | Ph'), Mv' = Ph' & (Eq | Mv) are
   three-input functions, v_bitop3, and the shift of Mh rides in a v_lshl_or) */
#define FPL_FILT_COL(Eq_)                                                                 \
    {                                                                                     \
        const u32 Eq = (Eq_);                                                             \
        const u32 s = (Eq & Pv) + Pv;                                                     \
        u32 Ph = Mv | bitop3<0x01>(s, Pv, Eq);        /* Mv | ~(s | Pv | Eq) */           \
        const u32 Mh = bitop3<0x8C>(s, Pv, Eq);       /* Pv & ((s ^ Pv) | Eq) */          \
        sc = sc + ((Ph >> 15) & MK) - ((Mh >> 15) & MK);                                  \
        acc &= sc;                                                                        \
        Ph <<= 1;                                                                         \
        Pv = lshl_or<1>(Mh, bitop3<0x01>(Eq, Mv, Ph)); /* (Mh << 1) | ~(Eq | Mv | Ph) */  \
        Mv = bitop3<0xE0>(Ph, Eq, Mv);                 /* Ph & (Eq | Mv) */               \
    }
    for (int j0 = 0; j0 < n; j0 += 32) { /* (n <= 200: seven blocks at most) */
        const int j1 = min(n, j0 + 32);
        for (int jj = j0; jj < j1; jj += 8) {
            u32 Eqs[8];
            if (jj + 8 <= j1) { /* wave-uniform: eight columns of the window */
                const u8* const pb = rowc + boff + (START ? n - 8 - jj : jj);
                u32 cb[8];
#pragma unroll
                for (int u = 0; u < 8; u++) cb[u] = (u32)pb[START ? 7 - u : u];
#pragma unroll
                for (int u = 0; u < 8; u++) Eqs[u] = wf[cb[u] * (4u * 64u)]; /* (the row: the same value in every lane) */
#pragma unroll
                for (int u = 0; u < 8; u++) FPL_FILT_COL(Eqs[u])
            } else {
                u32 cb[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int j = min(jj + u, n - 1);
                    cb[u] = (u32)rowc[(START ? n - 1 - j : j) + boff];
                }
#pragma unroll
                for (int u = 0; u < 8; u++) Eqs[u] = wf[cb[u] * (4u * 64u)];
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (jj + u < j1) FPL_FILT_COL(Eqs[u]) /* wave-uniform */
            }
        }
        blocks |= ((~acc >> 6) & 1u) << (j0 >> 5);
        acc |= 0x7Fu;
    }
#undef FPL_FILT_COL
    const bool fullF = ((acc >> (sF + 6u)) & 1u) == 0u;
    bool partF = blocks != 0;
    if (FPL_OPT_FASTANEAR && !fullF) {
        const int jn = n - alen + 16;
        partF = (jn <= 0 ? blocks : (blocks & (~0u << (jn >> 5)))) != 0;
    }
    return a_ok ? ((fullF ? 1u : 0u) | (partF ? 2u : 0u) | ((partF ? blocks : 0u) << 8)) : 0u;
}
/* The positions p of a partial-pattern search (window of 16 bases at p, src/adaptertrimmer.cpp:202-216 / :273-286) that the
   filter's verdict leaves open, [lo, hi]: column j of the run is the LAST byte of the end trim's window p = n - 1 - j (its
   windows end at byte n - 1 - p of the staged tail), and -- the start trim's run walks its window backwards -- the FIRST byte
   of the start trim's window p = n - 1 - j.  The semi-global score at a column bounds the global distance of the 16-base
   window that ends there from below, so every p the exact search could accept lies in the range. */
__device__ __forceinline__ void fasta_hint_range(u32 verdict, int n, int& lo, int& hi) {
    const u32 blocks = verdict >> 8;
    if (!blocks) {
        lo = 1;
        hi = 0;
        return;
    }
    const int jlo = 32 * (__ffs((int)blocks) - 1), jhi = min(n - 1, 32 * (31 - __clz((int)blocks)) + 31);
    lo = n - 1 - jhi;
    hi = n - 1 - jlo;
}
/* can this lane's adapter (length alen in 16..64, thresholds thrA / thrP) trim at this end?  win = the window bytes in
   LDS (window byte j at win[j + boff]), n of them; START: the start trim (partial pattern = the adapter's last 16 bases) */
template <bool START>
__device__ __forceinline__ bool fasta_may_trim(const FastaPeqLds* __restrict__ t, const u8* __restrict__ win, int boff, int n,
                                               int alen, int thrA, int thrP, bool a_ok) {
    const int lane = lane_id();
    u32 PvL = ~0u, PvH = ~0u, MvL = 0, MvH = 0; /* whole adapter: 64 columns in two words */
    u32 Pv = 0xFFFFu, Mv = 0;                   /* partial pattern: 16 columns */
    int scF = alen, scP = 16, bestF = alen, bestP = 16;
    const u32 topF = (u32)(alen - 1); /* bit of the adapter's last column, 15..63 */
    for (int j = 0; j < n; j++) {
        const u32 c = uniform_u32((u32)win[j + boff]); /* the same byte for every lane */
        const u32 code = (c >> 1) & 3u;
        const u32 row = (((0x47544341u >> (8 * code)) & 0xFFu) == c) ? code : 4u; /* exactly A / C / T / G, else the zero row */
        const u32 EqL = t->w[row][0][lane], EqH = t->w[row][1][lane], Eq = t->w[row][START ? 2 : 3][lane];
        { /* 64 columns; the row above the pattern is all zero (a match may start anywhere) */
            const u32 XvL = EqL | MvL, XvH = EqH | MvH;
            const u64 sum = (((u64)(EqH & PvH) << 32) | (EqL & PvL)) + (((u64)PvH << 32) | PvL);
            const u32 XhL = ((u32)sum ^ PvL) | EqL, XhH = ((u32)(sum >> 32) ^ PvH) | EqH;
            u32 PhL = MvL | ~(XhL | PvL), PhH = MvH | ~(XhH | PvH);
            u32 MhL = PvL & XhL, MhH = PvH & XhH;
            const u64 ph = ((u64)PhH << 32) | PhL, mh = ((u64)MhH << 32) | MhL;
            scF += (int)((ph >> topF) & 1ull) - (int)((mh >> topF) & 1ull);
            PhH = (PhH << 1) | (PhL >> 31);
            PhL <<= 1;
            MhH = (MhH << 1) | (MhL >> 31);
            MhL <<= 1;
            PvL = MhL | ~(XvL | PhL);
            PvH = MhH | ~(XvH | PhH);
            MvL = PhL & XvL;
            MvH = PhH & XvH;
            bestF = min(bestF, scF);
        }
        { /* 16 columns */
            const u32 Xv = Eq | Mv;
            const u32 Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
            u32 Ph = Mv | ~(Xh | Pv);
            u32 Mh = Pv & Xh;
            scP += (int)((Ph >> 15) & 1u) - (int)((Mh >> 15) & 1u);
            Ph <<= 1;
            Mh <<= 1;
            Pv = Mh | ~(Xv | Ph);
            Mv = Ph & Xv;
            bestP = min(bestP, scP);
        }
    }
    return a_ok && (bestF <= thrA || bestP <= thrP);
}

/* MODE (DevConfig::trim_mode, chosen by the host): 0 = anything; 1 = no FASTA adapters and command-line adapters of
   16..32 bases; 2 = every adapter (command-line and FASTA) has 16..64 bases.  Modes 1 and 2 leave the global-memory
   paths, the short-pattern variants and the multi-word Levenshtein out (mode 1 also the FASTA chain): a fraction of
   the code, fewer scalar registers to spill */
#ifndef FPL_TRIM_WPS_ONEFP
#define FPL_TRIM_WPS_ONEFP 7 /* (c5 side by side: 4 waves per SIMD 10.64 ms, 5 9.93, 6 9.64, 7 9.48, 8 9.61) */
#endif
/* ONEFP (MODE 2, at most 64 FASTA adapters: one group of lanes; launched for the chain alone, behind k_trim_ends_batched): the filter's
   Peq table is the same for every read and wave, so the block holds ONE copy, filled once, instead of one per wave (20 of the
   block's 40 KB of LDS), and an instantiation that only ever runs the chain needs neither the command-line adapters' tables, nor
   the quality windows, nor their key histograms (11 KB more) or the registers of the code in front of the chain: 13 KB of LDS and
   72 vector registers, seven waves per SIMD instead of four -- the exact trims are strings of dependent LDS round trips that more
   waves hide (k_trim_ends<2> on c5 10.64 -> 9.48 ms, profiles/r05_v3/ab_filter_packed.txt) */
template <int WAVES, int MODE, bool ONEFP = false>
/* (MODE 2 with the adapter filter holds two 64-column Myers states per lane next to the exact trims' registers: 4 waves
   per SIMD, up to 128 VGPRs) */
__global__ void __launch_bounds__(WAVES * 64, MODE == 1 ? FPL_TRIM_WAVES_PER_SIMD_SHORT
                                              : (MODE == 2 && FPL_OPT_FASTAFILTER) ? (ONEFP ? FPL_TRIM_WPS_ONEFP : 4) : FPL_TRIM_WAVES_PER_SIMD)
k_trim_ends(const u8* __restrict__ seq, const u8* __restrict__ qual, const uint64_t* __restrict__ off, u32 n_reads,
            uint64_t n_bytes, const DevConfig* __restrict__ cfg, const DevAdapter* __restrict__ ads,
            ReadState* __restrict__ state, long long* __restrict__ counters, u32 C, int from_state) {
    constexpr bool CO = ONEFP; /* this instantiation only ever runs the chain (launched with from_state != 0) */
    __shared__ TrimBlockAcc<CO> acc;
    __shared__ TrimLds<WAVES, CO> lds;
    constexpr bool FILT = MODE == 2 && FPL_OPT_FASTAFILTER != 0;
    static_assert(!ONEFP || FILT, "ONEFP is a form of the filtered chain");
    __shared__ FastaPeqLds fpeq[FILT && !ONEFP ? WAVES : 1]; /* per wave: the Peq words of the FASTA adapters being filtered */
    if (ONEFP && wave_in_block() == 0) { /* (n_fasta <= 64: lane = adapter, once per block; the barrier below publishes it) */
        const bool a_ok = lane_id() < cfg->n_fasta;
        fasta_peq_store(&fpeq[0], &ads[2 + (a_ok ? lane_id() : 0)], a_ok);
    }
    const int lane = lane_id();
    for (u32 i = threadIdx.x; i < FPL_FR_LEN; i += blockDim.x) acc.fr[i] = 0;
    for (u32 i = threadIdx.x; !CO && i < 2 * 2 * FPL_KEY_STRIDE; i += blockDim.x) acc.key[CO ? 0 : i] = 0;
    for (u32 i = threadIdx.x; !CO && i < 256; i += blockDim.x) {
        lds.peq16[0][CO ? 0 : i] = (uint16_t)ads[0].peq16_start[i];
        lds.peq16[CO ? 0 : 1][CO ? 0 : i] = (uint16_t)ads[1].peq16_end[i];
        lds.peqf[0][CO ? 0 : i][0] = ads[0].peq_full[i][0];
        lds.peqf[CO ? 0 : 1][CO ? 0 : i][0] = ads[1].peq_full[i][0];
    }
    __syncthreads();
    const u8* seq_end = seq + n_bytes;
    const u8* qual_end = qual + n_bytes;
    FastaPeqLds* const fp = &fpeq[FILT && !ONEFP ? wave_in_block() : 0];
    int fp_group = ONEFP ? 0 : -1; /* the group of 64 FASTA adapters whose words fp holds */
    u32* const win_s = lds.win[wave_in_block()][0];  /* head of the read, bases (later: the start trim's window) */
    u32* const win_e = lds.win[wave_in_block()][1];  /* tail of the read, bases (later: the end trim's window) */
    u32* const win_hq = lds.win[wave_in_block()][CO ? 0 : 2]; /* head / tail, qualities (chain only: never touched) */
    u32* const win_tq = lds.win[wave_in_block()][CO ? 1 : 3];
    u32* const win4_s = MODE != 0 ? lds.win4[wave_in_block()][0] : nullptr; /* (only the reduced instantiations scan nibbles) */
    u32* const win4_e = MODE != 0 ? lds.win4[wave_in_block()][1] : nullptr;
    __shared__ u32 winr[FILT ? WAVES : 1][2][TRIM_WIN / 4 + 8]; /* the two windows as Peq-table rows (stage_window's dstr) */
    u32* const winr_s = FILT ? winr[wave_in_block()][0] : nullptr;
    u32* const winr_e = FILT ? winr[wave_in_block()][1] : nullptr;

    const u32 wave_global = blockIdx.x * WAVES + wave_in_block();
    const u32 n_waves = gridDim.x * WAVES;
    long long* keyh = counters + FPL_OFF_KEYHIST(C);
    PROF_INIT();
    /* chain only (MODE 2, from_state != 0: wave-uniform): trimAndCut, polyX and the two command-line adapters are done --
       k_trim_ends_batched<.., 8, true> left r1 and the bases they took in state[] (lane = read there: a fraction of what the
       wave-per-read forms below cost for them) -- and this kernel runs the FASTA chain from there */
    const bool chain_only = CO || (MODE == 2 && from_state != 0);
    bool pq_zeroed = false; /* wave-uniform */
    for (u32 ri = wave_global; ri < n_reads; ri += n_waves) {
        const uint64_t o0 = off[ri];
        const int l = (int)(off[ri + 1] - o0);
        const u8* sq = seq + o0;
        const u8* ql = qual + o0;
        int s = 0, e = 0;
        PROF(0)
        /* the read's one trip to memory: its first and last TRIM_WIN bytes, bases and qualities, into LDS */
        const int tail0 = max(0, l - TRIM_WIN);
        bool alive;
        int trimmed_pre = 0;
        if (chain_only) {
            const ReadState ps = state[ri];
            s = (int)uniform_u32(ps.s);
            e = (int)uniform_u32(ps.e);
            alive = uniform_u32(ps.dropped) == 0;
            trimmed_pre = (int)uniform_u32(ps.pad);
        } else {
            stage_ends<MODE != 0>(win_s, win_e, win_hq, win_tq, win4_s, win4_e, sq, ql, l, tail0, seq_end, qual_end);
        }
        const EndsView vs = {sq, (const u8*)win_s, (const u8*)win_e, tail0};
        const EndsView vq = {ql, (const u8*)win_hq, (const u8*)win_tq, tail0};
        if (!chain_only) alive = trim_and_cut_wave(vs, vq, l, cfg, s, e);
        PROF(1)
        if (!chain_only && alive && cfg->polyx) { /* src/seprocessor.cpp:198-201 */
            int poly, tl;
            const int nl = trim_polyx_wave(vs, s, e - s, cfg->polyx_min_len, poly, tl);
            e = s + nl;
            if (poly >= 0 && lane == 0) {
                atomicAdd(&acc.fr[FPL_FR_POLYX_READS + poly], (u64)1);
                atomicAdd(&acc.fr[FPL_FR_POLYX_BASES + poly], (u64)tl);
            }
        }
        PROF(2) /* polyX */
        if (alive && cfg->adapter_enabled) { /* src/seprocessor.cpp:205-216 */
            int trimmed = trimmed_pre, kl;
            if (chain_only) {
                /* (the command-line adapters have had their turn) */
            } else if (cfg->has_start && (MODE != 0 || ads[0].len <= 64)) {
                /* the start trim only looks at r1[0, 200): still inside the staged head of the read unless trimAndCut
                   took more than TRIM_WIN - 200 bases (then the window is fetched again) */
                const int wl = min(e - s, FPL_END_WINDOW);
                int bias = -s; /* r1 byte j lives at window byte j - bias */
                if (s + wl > TRIM_WIN) {
                    stage_window(win_s, sq + s, wl, seq_end, win4_s);
                    bias = 0;
                }
                const Win<true> wn = {nullptr, win_s, bias, e - s, win4_s};
                trimmed += trim_start_wave<MODE>(wn, s, e, &ads[0], lds.peq16[0], lds.peqf[0], cfg, kl);
                if (!CO && kl > 0 && lane == 0) atomicAdd(&acc.key[CO ? 0 : (0 * 2 + 0) * FPL_KEY_STRIDE + kl], 1u);
            } else if (MODE == 0 && cfg->has_start) {
                const Win<false> wn = {sq + s, nullptr, 0, e - s};
                trimmed += trim_start_wave<0>(wn, s, e, &ads[0], ads[0].peq16_start, ads[0].peq_full, cfg, kl);
                if (!CO && kl > 0 && lane == 0) atomicAdd(&acc.key[CO ? 0 : (0 * 2 + 0) * FPL_KEY_STRIDE + kl], 1u);
            }
            PROF(3) /* start adapter */
            if (chain_only) {
            } else if (cfg->has_end && (MODE != 0 || ads[1].len <= 64)) {
                /* the end trim only looks at the last 200 bases of r1 */
                const int rlen = e - s, wl = min(rlen, FPL_END_WINDOW);
                int bias = tail0 - s; /* (the staged tail of the read starts at byte tail0) */
                if (e - wl < tail0) {
                    stage_window(win_e, sq + e - wl, wl, seq_end, win4_e);
                    bias = rlen - wl;
                }
                const Win<true> wn = {nullptr, win_e, bias, rlen, win4_e};
                trimmed += trim_end_wave<MODE>(wn, s, e, &ads[1], lds.peq16[CO ? 0 : 1], lds.peqf[CO ? 0 : 1], cfg, kl);
                if (!CO && kl > 0 && lane == 0) atomicAdd(&acc.key[CO ? 0 : (1 * 2 + 1) * FPL_KEY_STRIDE + kl], 1u);
            } else if (MODE == 0 && cfg->has_end) {
                const Win<false> wn = {sq + s, nullptr, 0, e - s};
                trimmed += trim_end_wave<0>(wn, s, e, &ads[1], ads[1].peq16_end, ads[1].peq_full, cfg, kl);
                if (!CO && kl > 0 && lane == 0) atomicAdd(&acc.key[CO ? 0 : (1 * 2 + 1) * FPL_KEY_STRIDE + kl], 1u);
            }
            PROF(4) /* end adapter */
            /* trimByMultiSequences, src/adaptertrimmer.cpp:42-57: every FASTA adapter at both ends, in order.  The
               two 200-base windows stay in LDS across adapters and are staged again only after a trim moved r1;
               each adapter's 16-column Peq table is copied next to them (4 loads per lane) */
            bool stale_s = true, stale_e = true;
            uint16_t* const pq = lds.peq16w[wave_in_block()];
            if (FILT && !pq_zeroed) { /* (with the filter only the four letters' words are ever written: the rest stays zero) */
                for (int i = lane; i < 256; i += 64) pq[i] = 0;
                pq_zeroed = true;
            }
            /* which adapters of the current group of 64 can trim the start / the end of r1 as it is now (fasta_may_trim) */
            u64 may_s = ~0ull, may_e = ~0ull;
            u64 full_s = ~0ull, part_s = ~0ull, full_e = ~0ull, part_e = ~0ull; /* ... and which of its two searches could succeed */
            u32 verd_s = 0, verd_e = 0; /* this lane's adapter: the filter's verdicts (fasta_may_trim32), incl. where the partial pattern may sit */
            int fn_s = 0, fn_e = 0;     /* window bytes the verdicts were made on */
            bool masks_ok = false;
            bool dirty_s = true, dirty_e = true; /* which end's verdicts are out of date */
            auto refresh_masks = [&](int a) {
                const int g = a >> 6, ai = g * 64 + lane;
                const int rlen = e - s, wl = min(rlen, FPL_END_WINDOW);
                masks_ok = true;
                if (rlen < FPL_PATTERN_LEN) { /* (no trim looks at an r1 this short) */
                    may_s = may_e = 0;
                    return;
                }
                if (stale_s) stage_window(win_s, sq + s, wl, seq_end, win4_s, winr_s);
                if (stale_e) stage_window(win_e, sq + e - wl, wl, seq_end, win4_e, winr_e);
                stale_s = stale_e = false;
                const bool a_ok = ai < cfg->n_fasta;
                const DevAdapter* la = &ads[2 + (a_ok ? ai : a)];
                if (g != fp_group) {
                    wave_sync();
                    fasta_peq_store(fp, la, a_ok);
                    wave_sync();
                    fp_group = g;
                    dirty_s = dirty_e = true;
                }
                const int alen = la->len;
                const int thrA = cfg->thr[alen], thrP = cfg->thr[FPL_PATTERN_LEN];
                if (FPL_DBG(cfg->dbg, 2048)) { /* (timing experiment: no filter, no exact trims) */
                    may_s = may_e = 0;
                    return;
                }
#if FPL_OPT_FASTAFILTER == 2
                /* (a trim at one end leaves the other end's window -- and with it that end's verdicts -- as they were, unless r1
                   has become shorter than the window) */
                /* (wave-uniform: every adapter of the group -- a lane without one borrows adapter a's length -- has the 23 bases
                   the packed form of the filter needs) */
                const bool packed = FPL_OPT_FILTPACK && wave_ballot(alen < 23 || (u32)thrA > 63u || (u32)thrP > 63u) == 0; /* (thresholds: the bias 63 - thr) */
                if (dirty_s) {
                    verd_s = packed ? fasta_may_trim32p<true>(fp, (const u8*)winr_s, 0, wl, alen, thrA, thrP, a_ok)
                                    : fasta_may_trim32<true>(fp, (const u8*)winr_s, 0, wl, alen, thrA, thrP, a_ok);
                    fn_s = wl;
#ifdef FPL_EMU /* (the emulator holds the packed form against the plain one, verdict by verdict) */
                    if (packed && verd_s != fasta_may_trim32<true>(fp, (const u8*)winr_s, 0, wl, alen, thrA, thrP, a_ok)) emu_fail("fasta_may_trim32p<start>");
#endif
                    full_s = wave_ballot((verd_s & 1u) != 0);
                    part_s = wave_ballot((verd_s & 2u) != 0);
                    may_s = full_s | part_s;
                }
                if (dirty_e) {
                    verd_e = packed ? fasta_may_trim32p<false>(fp, (const u8*)winr_e, 0, wl, alen, thrA, thrP, a_ok)
                                    : fasta_may_trim32<false>(fp, (const u8*)winr_e, 0, wl, alen, thrA, thrP, a_ok);
                    fn_e = wl;
#ifdef FPL_EMU
                    if (packed && verd_e != fasta_may_trim32<false>(fp, (const u8*)winr_e, 0, wl, alen, thrA, thrP, a_ok)) emu_fail("fasta_may_trim32p<end>");
#endif
                    full_e = wave_ballot((verd_e & 1u) != 0);
                    part_e = wave_ballot((verd_e & 2u) != 0);
                    may_e = full_e | part_e;
                }
                dirty_s = dirty_e = false;
#else
                may_s = wave_ballot(fasta_may_trim<true>(fp, (const u8*)win_s, 0, wl, alen, thrA, thrP, a_ok));
                may_e = wave_ballot(fasta_may_trim<false>(fp, (const u8*)win_e, 0, wl, alen, thrA, thrP, a_ok));
#endif
                if (FPL_DBG(cfg->dbg, 1024)) may_s = may_e = 0; /* (timing experiment: the filter, but no exact trims) */
#ifdef FPL_EMU_FILTER_STATS
                if (lane == 0) {
                    __atomic_fetch_add(&g_filter_stats[0], 1ull, __ATOMIC_RELAXED);
                    __atomic_fetch_add(&g_filter_stats[1], (unsigned long long)__builtin_popcountll(may_s), __ATOMIC_RELAXED);
                    __atomic_fetch_add(&g_filter_stats[2], (unsigned long long)__builtin_popcountll(may_e), __ATOMIC_RELAXED);
                    __atomic_fetch_add(&g_filter_stats[3], (unsigned long long)(__builtin_popcountll(full_s) + __builtin_popcountll(full_e)), __ATOMIC_RELAXED);
                }
#endif
            };
            /* r1 moved from [s0, e0) to [s, e): which windows -- and verdicts -- that leaves standing */
            auto moved = [&](int s0, int e0) {
                const int rl = e - s;
                stale_s = stale_s || s != s0 || rl < FPL_END_WINDOW;
                stale_e = stale_e || e != e0 || rl < FPL_END_WINDOW;
                dirty_s = dirty_s || s != s0 || rl < FPL_END_WINDOW;
                dirty_e = dirty_e || e != e0 || rl < FPL_END_WINDOW;
                masks_ok = false; /* what can trim r1 has to be asked again */
            };
            for (int a = 0; MODE != 1 && a < cfg->n_fasta; a++) {
                const DevAdapter* ad = &ads[2 + a];
                if (MODE == 0 && ad->len > FPL_END_WINDOW) { /* longer than the window: work on the read in global memory */
                    const Win<false> ws = {sq + s, nullptr, 0, e - s};
                    trimmed += trim_start_wave<0>(ws, s, e, ad, ad->peq16_start, ad->peq_full, cfg, kl);
                    if (kl > 0 && lane == 0) atomicAdd((u64*)&keyh[((2 + a) * 2 + 0) * FPL_KEY_STRIDE + kl], (u64)1);
                    const Win<false> we = {sq + s, nullptr, 0, e - s};
                    trimmed += trim_end_wave<0>(we, s, e, ad, ad->peq16_end, ad->peq_full, cfg, kl);
                    if (kl > 0 && lane == 0) atomicAdd((u64*)&keyh[((2 + a) * 2 + 1) * FPL_KEY_STRIDE + kl], (u64)1);
                    stale_s = stale_e = true;
                    continue;
                }
                if (FILT && (!masks_ok || (a >> 6) != fp_group)) refresh_masks(a);
                if (!FILT || ((may_s >> (a & 63)) & 1ull)) {
                    if (stale_s) stage_window(win_s, sq + s, min(e - s, FPL_END_WINDOW), seq_end, win4_s, winr_s);
                    stale_s = false;
                    wave_sync();
                    if (FILT) { /* (A / C / G / T adapters: the four words are in the filter's table -- the start trim's reversed) */
                        if (lane < 4) pq[(0x47544341u >> (8 * lane)) & 0xFFu] = (uint16_t)(FPL_OPT_FASTAFILTER == 2 ? __brev(fp->w[lane][2][a & 63]) >> 16 : fp->w[lane][2][a & 63]);
                    } else {
                        for (int i = lane; i < 256; i += 64) pq[i] = (uint16_t)ad->peq16_start[i];
                    }
                    wave_sync();
                    const int s0 = s, e0 = e;
                    const Win<true> wn = {nullptr, win_s, 0, e - s, win4_s};
                    int hlo = 0, hhi = 0x3fffffff;
                    if (FILT && FPL_OPT_FASTAFILTER == 2) fasta_hint_range(readlane_u32(verd_s, a & 63), fn_s, hlo, hhi);
                    trimmed += trim_start_wave<MODE>(wn, s, e, ad, pq, ad->peq_full, cfg, kl, ((full_s >> (a & 63)) & 1ull) != 0,
                                                     ((part_s >> (a & 63)) & 1ull) != 0, hlo, hhi);
                    if (kl > 0 && lane == 0) atomicAdd((u64*)&keyh[((2 + a) * 2 + 0) * FPL_KEY_STRIDE + kl], (u64)1);
#ifdef FPL_EMU_FILTER_STATS
                    if (lane == 0) {
                        __atomic_fetch_add(&g_filter_stats[4], 1ull, __ATOMIC_RELAXED);
                        if (s != s0 || e != e0) __atomic_fetch_add(&g_filter_stats[5], 1ull, __ATOMIC_RELAXED);
                    }
#endif
                    if (s != s0 || e != e0) moved(s0, e0);
                }
                if (FILT && !masks_ok) refresh_masks(a);
                if (!FILT || ((may_e >> (a & 63)) & 1ull)) {
                    const int rlen = e - s, wl = min(rlen, FPL_END_WINDOW);
                    if (stale_e) stage_window(win_e, sq + e - wl, wl, seq_end, win4_e, winr_e);
                    stale_e = false;
                    wave_sync();
                    if (FILT) {
                        if (lane < 4) pq[(0x47544341u >> (8 * lane)) & 0xFFu] = (uint16_t)fp->w[lane][3][a & 63];
                    } else {
                        for (int i = lane; i < 256; i += 64) pq[i] = (uint16_t)ad->peq16_end[i];
                    }
                    wave_sync();
                    const int s0 = s, e0 = e;
                    const Win<true> wn = {nullptr, win_e, rlen - wl, rlen, win4_e};
                    int hlo = 0, hhi = 0x3fffffff;
                    if (FILT && FPL_OPT_FASTAFILTER == 2) fasta_hint_range(readlane_u32(verd_e, a & 63), fn_e, hlo, hhi);
                    trimmed += trim_end_wave<MODE>(wn, s, e, ad, pq, ad->peq_full, cfg, kl, ((full_e >> (a & 63)) & 1ull) != 0,
                                                   ((part_e >> (a & 63)) & 1ull) != 0, hlo, hhi);
                    if (kl > 0 && lane == 0) atomicAdd((u64*)&keyh[((2 + a) * 2 + 1) * FPL_KEY_STRIDE + kl], (u64)1);
#ifdef FPL_EMU_FILTER_STATS
                    if (lane == 0) {
                        __atomic_fetch_add(&g_filter_stats[4], 1ull, __ATOMIC_RELAXED);
                        if (s != s0 || e != e0) __atomic_fetch_add(&g_filter_stats[5], 1ull, __ATOMIC_RELAXED);
                    }
#endif
                    if (s != s0 || e != e0) moved(s0, e0);
                }
            }
            if (trimmed > 0 && lane == 0) { /* FilterResult::addReadTrimmed */
                atomicAdd(&acc.fr[FPL_FR_ADAPTER_READS], (u64)1);
                atomicAdd(&acc.fr[FPL_FR_ADAPTER_BASES], (u64)trimmed);
            }
        }
        if (lane == 0) {
            ReadState st;
            st.s = alive ? (u32)s : 0;
            st.e = alive ? (u32)e : 0;
            st.dropped = alive ? 0 : 1;
            st.pad = 0;
            state[ri] = st;
        }
        PROF(5)
    }
    PROF_FLUSH(16);
    __syncthreads();
    long long* fr = counters + FPL_OFF_FR(C);
    for (u32 i = threadIdx.x; i < FPL_FR_LEN; i += blockDim.x)
        if (acc.fr[i]) atomicAdd((u64*)&fr[i], acc.fr[i]);
    for (u32 i = threadIdx.x; !CO && i < 2 * 2 * FPL_KEY_STRIDE; i += blockDim.x)
        if (acc.key[CO ? 0 : i]) atomicAdd((u64*)&keyh[i], (u64)acc.key[CO ? 0 : i]);
}

/* -----------------------------------------------------------------------------------------
 * k_trim_ends_batched: the same result as k_trim_ends<MODE 1> (no FASTA list, command-line adapters of 16..32 bases,
 * A / C / G / T only) with the edit-distance confirmations taken out of the per-read stream.
 *
 * k_trim_ends runs every confirmation (src/adaptertrimmer.cpp:126-131 / :100-107 / :218-233 / :288-299) as a wave-wide
 * Myers recurrence on broadcast words: all of it wave-uniform, so it lands on the CU's one scalar unit -- ~30 scalar
 * instructions per column, up to four confirmations of 16..32 columns per read, and the scalar unit is what bounds the
 * kernel (1.6e9 scalar against 1.5e9 vector wave-instructions on the bench batch).  Here a wave takes 64 reads at a time
 * and alternates between per-read steps (lanes = candidate positions, as before) and per-lane steps (lane j = read j of
 * the group): the per-read steps only FIND the window a confirmation is about and park it in lane j; one lane-parallel
 * Myers pass (lev_lanes32: every lane its own text, pattern slice and threshold) then confirms 64 of them at once, and
 * the arithmetic that follows a search (trimming extension, key length, Read::trimFront / resize) is per-lane work too.
 *
 *   per read   P1  stage head + tail, trimAndCut, polyX, start adapter: window Hamming scan -> hit | candidate
 *   per lane   P2  confirm the candidates                                   (:126-131)
 *   per read   P3  (no full match) partial-pattern search at the start      (:202-216)
 *   per lane   P4  confirm the partial matches, finish the start trim       (:185-193, :218-233)
 *   per read   P5  end adapter: window Hamming scan -> hit | candidate      (:84-107)
 *   per lane   P6  confirm
 *   per read   P7  (no full match) partial-pattern search at the end        (:273-286)
 *   per lane   P8  confirm, finish the end trim, write the ReadState records (:256-264, :288-299)
 * --------------------------------------------------------------------------------------- */

/* ---- Filter::trimAndCut and PolyX::trimPolyX with LANE = READ (k_trim_ends_batched).
 * Both are short sequential scans with an early exit -- a sliding window that stops at the first good window (a handful of
 * positions in), a tail scan that stops once no base dominates (nine positions in when there is no poly-X tail).  With
 * lanes = positions (trim_and_cut_wave, trim_polyx_wave) a wave evaluates 64 candidates to use the first few, once per read;
 * here every lane walks the reference's own loop on its own read and the wave loops for as long as its slowest lane -- a
 * few dozen rounds for 64 reads.  A lane whose scan outlasts LANE_SCAN_CAP rounds gives up (slow = true): the caller sends
 * that read through the wave-per-read forms, so that one read with a long bad stretch cannot hold 63 others. */
constexpr int LANE_SCAN_CAP = 160;
constexpr int LANE_SCAN_WMAX = 16; /* windows beyond this take the wave-per-read form for every read (the warm-up sum is a loop) */

/* byte K (0..15, compile time) of a 16-byte block */
template <int K>
__device__ __forceinline__ u32 byte16(const u32x4& v) {
    const u32 d = K < 4 ? v.x : (K < 8 ? v.y : (K < 12 ? v.z : v.w));
    return (d >> (8 * (K & 3))) & 0xFFu;
}
/* the sum of the first n (0..16, wave-uniform) bytes of a block / of its last n */
__device__ __forceinline__ u32 sum_first_bytes(const u32x4& v, int n) {
    const u32 d[4] = {v.x, v.y, v.z, v.w};
    u32 t = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int c = n - 4 * k;
        const u32 bm = c >= 4 ? ~0u : (c <= 0 ? 0u : ((1u << (8 * c)) - 1u));
        t = sum_bytes(d[k] & bm, t);
    }
    return t;
}
__device__ __forceinline__ u32 sum_last_bytes(const u32x4& v, int n) {
    const u32 d[4] = {v.w, v.z, v.y, v.x};
    u32 t = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int c = n - 4 * k;
        const u32 bm = c >= 4 ? ~0u : (c <= 0 ? 0u : ~((1u << (8 * (4 - c))) - 1u));
        t = sum_bytes(d[k] & bm, t);
    }
    return t;
}

/* Filter::trimAndCut, src/filter.cpp:130-232, one read per lane: sq / ql = this lane's read (length l).
   Out: alive (false: the reference returns NULL), [s, e) in coordinates of the read, slow (see above). */
__device__ __forceinline__ void trim_and_cut_lanes(const u8* __restrict__ sq, const u8* __restrict__ ql, int l, bool valid,
                                                   const DevConfig* __restrict__ cfg, int& s_out, int& e_out, bool& alive,
                                                   bool& slow) {
    int front = cfg->trim_front, tail = cfg->trim_tail;
    const bool cf = cfg->cut_front != 0, ct = cfg->cut_tail != 0;
    s_out = 0;
    e_out = l;
    alive = valid;
    slow = false;
    if (front == 0 && tail == 0 && !cf && !ct) return; /* :133-134 (wave-uniform) */
    int rlen = l - front - tail;
    if (rlen < 0) alive = false; /* :137-139 */
    if (!cf && !ct) { /* :141-151 */
        s_out = front;
        e_out = front + rlen;
        return;
    }
    u32 touch = 0;
    if (FPL_OPT_TRIMTOUCH && valid && l > 0) {
        /* the four lines the scans below (and polyX behind them) start in: requested together, consumed one after the other */
        const int h = min(front, l - 1), t = max(l - 1 - tail, 0);
        touch = (u32)ql[h] + (u32)ql[t] + (u32)sq[h] + (u32)sq[t];
    }
    if (cf) { /* :159-189 */
        const int w = cfg->cut_front_w, thr = cfg->cut_front_thr;
        if (l - front - tail - w <= 0) alive = false;
        const int lim = l - tail - w; /* the loop runs while s < lim */
        int s = front, total = 0, it = 0;
        bool fin = false; /* this lane's scan is over (the first 16 steps below may end it) */
        /* the first 16 steps out of two blocks: step k adds qual[front + w - 1 + k] and takes qual[front + k - 1] out again -- byte k
           of A, byte k - 1 of S, whichever lane (front and w are the same for all; the lanes that stop drop out) */
        const bool reg = FPL_OPT_TRIMREG && alive && w <= 16 && front + w + 15 <= l;
        if (FPL_OPT_TRIMREG && wave_ballot(reg)) { /* (wave-uniform) */
            u32x4 A = {0, 0, 0, 0}, S = {0, 0, 0, 0};
            if (reg) {
                A = load16(ql + front + w - 1);
                S = load16(ql + front);
                total = (int)sum_first_bytes(S, w - 1);
            }
            bool run = reg && s < lim;
            fin = reg && !run;
#define FPL_CF_STEP(K)                                                                  \
    if (wave_ballot(run)) {                                                             \
        const int t2 = total + (int)byte16<K>(A) - (K > 0 ? (int)byte16<(K > 0 ? K - 1 : 0)>(S) : 0); \
        total = run ? t2 : total;                                                       \
        const bool stop = run && total >= thr;                                          \
        fin = fin || stop;                                                              \
        run = run && !stop;                                                             \
        s += run ? 1 : 0;                                                               \
        const bool out = run && s >= lim;                                               \
        fin = fin || out;                                                               \
        run = run && !out;                                                              \
    }
            FPL_CF_STEP(0) FPL_CF_STEP(1) FPL_CF_STEP(2) FPL_CF_STEP(3) FPL_CF_STEP(4) FPL_CF_STEP(5) FPL_CF_STEP(6) FPL_CF_STEP(7)
            FPL_CF_STEP(8) FPL_CF_STEP(9) FPL_CF_STEP(10) FPL_CF_STEP(11) FPL_CF_STEP(12) FPL_CF_STEP(13) FPL_CF_STEP(14) FPL_CF_STEP(15)
#undef FPL_CF_STEP
            it = reg ? 16 : 0; /* (a lane still running has made 16 steps; for the others the count no longer matters) */
        }
        if (alive && !reg)
            for (int i = 0; i < w - 1; i++) total += ql[front + i];
        while (alive && !slow && !fin && s < lim) {
            total += ql[s + w - 1];
            if (s > front) total -= ql[s - 1];
            if (total >= thr) break; /* total / w >= 33 + q */
            s++;
            if (++it > LANE_SCAN_CAP) slow = true;
        }
        if (s > 0) s = s + w - 1;
        it = 0;
        while (alive && !slow && s < l && sq[s] == 'N') {
            s++;
            if (++it > LANE_SCAN_CAP) slow = true;
        }
        front = s;
        rlen = l - front - tail;
    }
    if (ct) { /* :191-219 */
        const int w = cfg->cut_tail_w, thr = cfg->cut_tail_thr;
        if (l - front - tail - w <= 0) alive = false;
        int t = l - tail - 1, total = 0, it = 0;
        bool fin = false;
        /* the mirror image: step k adds qual[T0 - w + 1 - k] and takes qual[T0 + 1 - k] out again (T0 = l - tail - 1, a lane's own):
           byte 15 - k of the block that ENDS at T0 - w + 1, byte 16 - k of the block that ends at T0 */
        const bool reg = FPL_OPT_TRIMREG && alive && !slow && w <= 16 && t >= w + 14;
        if (FPL_OPT_TRIMREG && wave_ballot(reg)) { /* (wave-uniform) */
            u32x4 A = {0, 0, 0, 0}, S = {0, 0, 0, 0};
            if (reg) {
                A = load16(ql + t - w + 1 - 15);
                S = load16(ql + t - 15);
                total = (int)sum_last_bytes(S, w - 1); /* qual[t - w + 2 .. t] */
            }
            bool run = reg && t - w >= front;
            fin = reg && !run;
#define FPL_CT_STEP(K)                                                                  \
    if (wave_ballot(run)) {                                                             \
        const int t2 = total + (int)byte16<15 - K>(A) - (K > 0 ? (int)byte16<(K > 0 ? 16 - K : 0)>(S) : 0); \
        total = run ? t2 : total;                                                       \
        const bool stop = run && total >= thr;                                          \
        fin = fin || stop;                                                              \
        run = run && !stop;                                                             \
        t -= run ? 1 : 0;                                                               \
        const bool out = run && t - w < front;                                          \
        fin = fin || out;                                                               \
        run = run && !out;                                                              \
    }
            FPL_CT_STEP(0) FPL_CT_STEP(1) FPL_CT_STEP(2) FPL_CT_STEP(3) FPL_CT_STEP(4) FPL_CT_STEP(5) FPL_CT_STEP(6) FPL_CT_STEP(7)
            FPL_CT_STEP(8) FPL_CT_STEP(9) FPL_CT_STEP(10) FPL_CT_STEP(11) FPL_CT_STEP(12) FPL_CT_STEP(13) FPL_CT_STEP(14) FPL_CT_STEP(15)
#undef FPL_CT_STEP
            it = reg ? 16 : 0;
        }
        if (alive && !slow && !reg)
            for (int i = 0; i < w - 1; i++) total += ql[t - i]; /* qual[t - w + 2 .. t] */
        while (alive && !slow && !fin && t - w >= front) {
            total += ql[t - w + 1];
            if (t < l - tail - 1) total -= ql[t + 1];
            if (total >= thr) break;
            t--;
            if (++it > LANE_SCAN_CAP) slow = true;
        }
        if (t < l - 1) t = t - w + 1;
        it = 0;
        while (alive && !slow && t >= 0 && sq[t] == 'N') {
            t--;
            if (++it > LANE_SCAN_CAP) slow = true;
        }
        rlen = t - front + 1;
    }
    if (rlen <= 0 || front >= l - 1) alive = false; /* :221-222 */
    s_out = front;
    e_out = front + rlen;
#if !defined(FPL_EMU)
    if (FPL_OPT_TRIMTOUCH) asm volatile("" ::"v"(touch)); /* (keeps the touch loads alive) */
#else
    (void)touch;
#endif
}

/* PolyX::trimPolyX, src/polyx.cpp:11-78, one read per lane: r = first base of r1 (rlen bases).  Returns the new
   length; poly (0..3 = A, T, C, G; -1: nothing cut) and the bases cut; slow as above. */
__device__ __forceinline__ int trim_polyx_lanes(const u8* __restrict__ r, int rlen, bool active, int compareReq, int& poly_out,
                                                int& trimmed_out, bool& slow) {
    poly_out = -1;
    trimmed_out = 0;
    int cA = 0, cT = 0, cC = 0, cG = 0;
    int P = rlen; /* value of pos when the scan ends without a break */
    int pos = 0, it = 0;
    bool fin = false;
    /* the first 32 steps out of the read's last 32 bytes, loaded up front: step k looks at byte 31 - k of them, whichever lane */
    const bool reg = FPL_OPT_TRIMREG && active && !slow && rlen >= 32;
    if (FPL_OPT_TRIMREG && wave_ballot(reg)) { /* (wave-uniform) */
        u32x4 R0 = {0, 0, 0, 0}, R1 = {0, 0, 0, 0};
        if (reg) {
            R0 = load16(r + rlen - 16);
            R1 = load16(r + rlen - 32);
        }
        bool run = reg; /* (rlen >= 32: no lane runs out of read here) */
#define FPL_PX_STEP(K)                                                                                          \
    if (wave_ballot(run)) {                                                                                     \
        const u32 c = K < 16 ? byte16<15 - (K & 15)>(R0) : byte16<15 - (K & 15)>(R1);                            \
        const int isn = c == 'N' ? 1 : 0;                                                                       \
        cA += run ? ((c == 'A' ? 1 : 0) + isn) : 0;                                                             \
        cT += run ? ((c == 'T' ? 1 : 0) + isn) : 0;                                                             \
        cC += run ? ((c == 'C' ? 1 : 0) + isn) : 0;                                                             \
        cG += run ? ((c == 'G' ? 1 : 0) + isn) : 0;                                                             \
        constexpr int cmp = K + 1, allowed = cmp / 8 < 5 ? cmp / 8 : 5;                                         \
        const bool need = (cmp - cA > allowed) && (cmp - cT > allowed) && (cmp - cC > allowed) && (cmp - cG > allowed); \
        const bool stop = run && need && (K >= 8 || K + 1 >= compareReq - 1);                                   \
        P = stop ? K : P;                                                                                       \
        fin = fin || stop;                                                                                      \
        run = run && !stop;                                                                                     \
        pos += run ? 1 : 0;                                                                                     \
    }
        FPL_PX_STEP(0) FPL_PX_STEP(1) FPL_PX_STEP(2) FPL_PX_STEP(3) FPL_PX_STEP(4) FPL_PX_STEP(5) FPL_PX_STEP(6) FPL_PX_STEP(7)
        FPL_PX_STEP(8) FPL_PX_STEP(9) FPL_PX_STEP(10) FPL_PX_STEP(11) FPL_PX_STEP(12) FPL_PX_STEP(13) FPL_PX_STEP(14) FPL_PX_STEP(15)
        FPL_PX_STEP(16) FPL_PX_STEP(17) FPL_PX_STEP(18) FPL_PX_STEP(19) FPL_PX_STEP(20) FPL_PX_STEP(21) FPL_PX_STEP(22) FPL_PX_STEP(23)
        FPL_PX_STEP(24) FPL_PX_STEP(25) FPL_PX_STEP(26) FPL_PX_STEP(27) FPL_PX_STEP(28) FPL_PX_STEP(29) FPL_PX_STEP(30) FPL_PX_STEP(31)
#undef FPL_PX_STEP
        it = reg ? 32 : 0;
    }
    while (active && !slow && !fin && pos < rlen) {
        const u32 c = r[rlen - pos - 1];
        cA += (c == 'A' || c == 'N');
        cT += (c == 'T' || c == 'N');
        cC += (c == 'C' || c == 'N');
        cG += (c == 'G' || c == 'N');
        const int cmp = pos + 1, allowed = min(5, cmp / 8);
        const bool need = (cmp - cA > allowed) && (cmp - cT > allowed) && (cmp - cC > allowed) && (cmp - cG > allowed);
        if (need && (pos >= 8 || pos + 1 >= compareReq - 1)) {
            P = pos;
            break;
        }
        pos++;
        if (++it > LANE_SCAN_CAP) slow = true;
    }
    if (!active || slow || P + 1 < compareReq) return rlen; /* :57 */
    int poly = 0, maxc = cA; /* the first maximum in A, T, C, G order */
    if (cT > maxc) { maxc = cT; poly = 1; }
    if (cC > maxc) { maxc = cC; poly = 2; }
    if (cG > maxc) { maxc = cG; poly = 3; }
    const u32 polyBase = poly == 0 ? 'A' : (poly == 1 ? 'T' : (poly == 2 ? 'C' : 'G'));
    /* :71  walk pos down from P until r[rlen - pos - 1] == polyBase: the first index >= max(0, rlen - P - 1) holding
       polyBase; index -1 (P == rlen) never matches; pos = -1 when nothing matches */
    int i = rlen - P - 1;
    if (i < 0) i = 0;
    it = 0;
    while (i < rlen && r[i] != polyBase) {
        i++;
        if (++it > LANE_SCAN_CAP) {
            slow = true;
            return rlen;
        }
    }
    const int p2 = i < rlen ? rlen - i - 1 : -1;
    poly_out = poly;
    trimmed_out = p2 + 1;
    return rlen - p2 - 1; /* Read::resize: a no-op when pos == -1 */
}

/* the four bases of a dword as one-hot nibbles (A 1, C 2, G 4, T 8; any other byte 0), base k in bits 4k..4k+3 */
__device__ __forceinline__ u32 onehot4(u32 w) {
    const u32 code = (w >> 1) & 0x03030303u;            /* A0 C1 T2 G3 */
    const u32 t = perm_lo(0x47544341u, code) ^ w;       /* zero byte <=> exactly that letter */
    const u32 ok = (~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t) >> 7) & 0x01010101u;
    const u32 nib = perm_lo(0x04080201u, code) & (ok * 15u);
    const u32 h = (nib | (nib >> 4)) & 0x00FF00FFu;
    return (h | (h >> 8)) & 0xFFFFu;
}
/* searchAdapter's window scan (src/adaptertrimmer.cpp:84-131) with LANE = READ: every lane slides a 32-base window of one-hot
 * nibbles along its own read -- four bases enter per dword load -- and counts the adapter's matches with four AND + popcount.
 * r1 / rlen: this lane's trimmed read.  START: asRightAsPossible over p in [0, min(rlen, 200) - alen]: hit = the LARGEST p
 * within thr (the reference scans downwards and returns at once), else cand = the smallest p among the minima (ties: "<="
 * in a descending scan).  END: asLeftAsPossible over p in [max(0, rlen - 200), rlen - alen): hit = the SMALLEST p within thr,
 * else cand = the largest p among the minima.  hit / cand are -1 when there is none.  A few hundred vector instructions per
 * 64 reads and end where the scan with lanes = positions (one read at a time) took as many per read. */
/* NW: words of one-hot nibbles the window holds -- 4 for adapters of <= 32 bases, 8 for <= 64 */
template <bool START, int NW = 4>
__device__ __forceinline__ void ham_scan_lanes(const u8* __restrict__ r1, int rlen, bool active, const u32 (&ad1h)[NW], int alen,
                                               int thr, const u8* __restrict__ seq_end, int& hit, int& cand) {
    hit = -1;
    cand = -1;
    int p_lo = 0, np = 0; /* first position and number of positions of this lane */
    if (START) {
        const int searchEnd = min(rlen, FPL_END_WINDOW);
        if (active && alen <= rlen && searchEnd > alen) np = searchEnd - alen + 1;
    } else {
        const int ss = max(0, rlen - FPL_END_WINDOW);
        if (active && ss + alen <= rlen) {
            p_lo = ss;
            np = rlen - alen - ss; /* the last position is never tested */
        }
    }
    const int npmax = (int)wave_max_u32((u32)max(np, 0));
    if (npmax == 0) return; /* wave-uniform */
    const u8* base = r1 + p_lo;
    const int navail = np > 0 ? rlen - p_lo : 0; /* bytes of the read from `base` on */
    /* dword k of the stream: bases 4k .. 4k + 3 behind `base`; bytes past the read count as no base */
    auto nibbles = [&](int k) -> u32 {
        const int left = navail - 4 * k;
        if (left <= 0) return 0u;
        u32 w = load4_guard(base + 4 * k, seq_end);
        if (left < 4) w &= (1u << (8 * left)) - 1u;
        return onehot4(w);
    };
    u32 touch = 0;
    if (FPL_OPT_TRIMTOUCH && navail > 64) /* the window's second (and third) cache line, asked for now */
        touch = (u32)*(base + min(navail - 1, 112)) + (u32)*(base + min(navail - 1, 207));
    u32 W[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) W[k] = nibbles(2 * k) | (nibbles(2 * k + 1) << 16);
    int bestmm = 0x7fffffff;
    u32 nxt = nibbles(2 * NW);
    for (int i0 = 0; i0 < npmax; i0 += 4) { /* wave-uniform trip count */
        const u32 cur = nxt;
        nxt = nibbles(i0 / 4 + 2 * NW + 1); /* (requested one step ahead of its use) */
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u;
            u32 matches = popc32(W[0] & ad1h[0]);
#pragma unroll
            for (int k = 1; k < NW; k++) matches = popc_acc(W[k] & ad1h[k], matches);
            const int mm = alen - (int)matches;
            const bool in = i < np;
            const int p = p_lo + i;
            if (START) {
                hit = (in && mm <= thr) ? p : hit;         /* ascending: the last one stands */
                const bool better = in && mm < bestmm;     /* the first of the minima */
                cand = better ? p : cand;
                bestmm = better ? mm : bestmm;
            } else {
                hit = (in && hit < 0 && mm <= thr) ? p : hit; /* the first one stands */
                const bool better = in && mm <= bestmm;       /* the last of the minima */
                cand = better ? p : cand;
                bestmm = better ? mm : bestmm;
            }
            /* slide by one base: the next nibble of the stream enters at the top */
#pragma unroll
            for (int k = 0; k + 1 < NW; k++) W[k] = alignbit(W[k + 1], W[k], 4);
            W[NW - 1] = (W[NW - 1] >> 4) | (((cur >> (4 * u)) & 15u) << 28);
        }
    }
    if (hit >= 0) cand = -1;
#if !defined(FPL_EMU)
    if (FPL_OPT_TRIMTOUCH) asm volatile("" ::"v"(touch));
#else
    (void)touch;
#endif
}

/* Global edit distance <= thr? between the adapter slice [shift, shift + m) (m <= 32; peqf = word 0 of the adapter's Peq
 * table, in LDS) and the m text bytes at `text`, one problem per lane (need = this lane has one).  The exact distance
 * as the reference's edit_distance computes it (src/editdistance.cpp:30-61), compared per lane. */
__device__ __forceinline__ bool lev_lanes32(const u8* __restrict__ text, int m, int shift, int thr, bool need,
                                            const uint64_t (*__restrict__ peqf)[1], const u8* __restrict__ seq_end) {
    u32 w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (need) {
        const u32x4 a = load16_guard(text, seq_end), b = load16_guard(text + 16, seq_end);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
        w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
    }
    const int mm = need ? m : 0;
    const u32 mask = mm >= 32 ? ~0u : ((1u << mm) - 1u);
    const u32 topsh = mm > 0 ? (u32)(mm - 1) : 0u;
    const int mmax = (int)wave_max_u32((u32)mm);
    u32 Pv = ~0u, Mv = 0;
    int score = mm;
#pragma unroll
    for (int t = 0; t < 32; t++) {
        if (t < mmax) { /* wave-uniform */
            const u32 c = (w[t >> 2] >> (8 * (t & 3))) & 0xFFu;
            const u32 Eq = (u32)(peqf[c][0] >> shift) & mask;
            const u32 Xv = Eq | Mv;
            const u32 Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
            u32 Ph = Mv | ~(Xh | Pv);
            u32 Mh = Pv & Xh;
            const int sc = score + (int)((Ph >> topsh) & 1u) - (int)((Mh >> topsh) & 1u);
            Ph = (Ph << 1) | 1u;
            Mh <<= 1;
            const bool act = t < mm; /* this lane's text has a column t */
            score = act ? sc : score;
            const u32 nPv = Mh | ~(Xv | Ph), nMv = Ph & Xv;
            Pv = act ? nPv : Pv;
            Mv = act ? nMv : Mv;
        }
    }
    return need && score <= thr;
}

/* ... the same for pattern slices of up to 64 columns (adapters of 33..64 bases): 64-bit columns, the text fetched 16 bytes at a time */
__device__ __forceinline__ bool lev_lanes64(const u8* __restrict__ text, int m, int shift, int thr, bool need,
                                            const uint64_t (*__restrict__ peqf)[1], const u8* __restrict__ seq_end) {
    const int mm = need ? m : 0;
    const u64 mask = mm >= 64 ? ~0ull : ((1ull << mm) - 1ull);
    const u32 topsh = mm > 0 ? (u32)(mm - 1) : 0u;
    const int mmax = (int)wave_max_u32((u32)mm);
    u64 Pv = ~0ull, Mv = 0;
    int score = mm;
    for (int t0 = 0; t0 < mmax; t0 += 16) { /* wave-uniform */
        u32 w[4] = {0, 0, 0, 0};
        if (need && t0 < mm) {
            const u32x4 a = load16_guard(text + t0, seq_end);
            w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
        }
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int t = t0 + u;
            const u32 c = (w[u >> 2] >> (8 * (u & 3))) & 0xFFu;
            const u64 Eq = (peqf[c][0] >> shift) & mask;
            const u64 Xv = Eq | Mv;
            const u64 Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
            u64 Ph = Mv | ~(Xh | Pv);
            u64 Mh = Pv & Xh;
            const int sc = score + (int)((Ph >> topsh) & 1ull) - (int)((Mh >> topsh) & 1ull);
            Ph = (Ph << 1) | 1ull;
            Mh <<= 1;
            const bool act = t < mm; /* this lane's text has a column t */
            score = act ? sc : score;
            const u64 nPv = Mh | ~(Xv | Ph), nMv = Ph & Xv;
            Pv = act ? nPv : Pv;
            Mv = act ? nMv : Mv;
        }
    }
    return need && score <= thr;
}
/* the one the adapter length of the instantiation asks for */
template <int NW>
__device__ __forceinline__ bool lev_lanes_nw(const u8* __restrict__ text, int m, int shift, int thr, bool need,
                                             const uint64_t (*__restrict__ peqf)[1], const u8* __restrict__ seq_end) {
    if (NW <= 4) return lev_lanes32(text, m, shift, thr, need, peqf, seq_end);
    return lev_lanes64(text, m, shift, thr, need, peqf, seq_end);
}

/* Can the 16-base partial pattern (peq16 = its Peq table, in LDS) match ANY 16-byte window of the n text bytes at
 * `text` with an edit distance <= thr?  One problem per lane.  Myers' search recurrence (the first DP row is all zero:
 * a match may start anywhere) gives, for every text position j, the best distance of the pattern against any substring
 * that ENDS at j; the global distance of the pattern against the window [j - 16, j) -- what the partial-pattern searches
 * of trimBySequenceStart / End compute per window, src/adaptertrimmer.cpp:202-216, 273-286 -- cannot be smaller.  So
 * "no j reaches thr" proves that the search finds nothing, at 1/12 of its cost and for 64 reads at once; "some j does"
 * proves nothing, and the read takes the exact search. */
__device__ __forceinline__ bool partial16_possible(const u8* __restrict__ text, int n, int thr, bool need,
                                                   const uint16_t* __restrict__ peq16, const u8* __restrict__ seq_end) {
    const int nn = need ? n : 0;
    const int nmax = (int)wave_max_u32((u32)nn);
    u32 Pv = 0xFFFFu, Mv = 0;
    int score = 16, best = 16;
    for (int c0 = 0; c0 < nmax; c0 += 16) { /* wave-uniform */
        u32 w[4] = {0, 0, 0, 0};
        if (c0 < nn) {
            const u32x4 a = load16_guard(text + c0, seq_end);
            w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
        }
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const u32 c = (w[t >> 2] >> (8 * (t & 3))) & 0xFFu;
            const u32 Eq = peq16[c];
            const u32 Xv = Eq | Mv;
            const u32 Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
            u32 Ph = Mv | ~(Xh | Pv);
            u32 Mh = Pv & Xh;
            const int sc = score + (int)((Ph >> 15) & 1u) - (int)((Mh >> 15) & 1u);
            Ph <<= 1; /* (no "| 1": the row above the pattern is zero everywhere) */
            Mh <<= 1;
            const bool act = c0 + t < nn;
            const u32 nPv = Mh | ~(Xv | Ph), nMv = Ph & Xv; /* (what gathers above bit 15 never comes back down) */
            score = act ? sc : score;
            Pv = act ? nPv : Pv;
            Mv = act ? nMv : Mv;
            best = min(best, score);
        }
    }
    return need && best <= thr;
}

#ifdef FPL_EMU_TRIM_STATS
static unsigned long long g_trim_stats[10]; /* emulator only: groups, lanes handed to P1b, P2 / P3 wants / P3 may, P6 / P7 wants / P7 may, lane-parallel partial searches / their rounds */
#define FPL_TRIM_STAT(i, n) (g_trim_stats[i] += (unsigned long long)(n))
#else
#define FPL_TRIM_STAT(i, n) ((void)0)
#endif
/* partial16_possible that also says WHERE (FPL_OPT_PARTLANES): bit (j & 31) of cand[j >> 5][lane] = the search variant's score
 * at text column j is <= thr -- the only columns a window accepted by the exact search can END at (its global distance is
 * no smaller than that score).  Returns, per lane, which of the seven words hold a bit (0: the search finds nothing).
 * The running value is score - thr - 1, so "reached thr" is its sign bit and one v_alignbit per column collects the bits:
 * the same instruction count as the minimum that partial16_possible keeps. */
constexpr int PART_WORDS = 7; /* ceil(FPL_END_WINDOW / 32) */
__device__ __forceinline__ u32 partial16_candidates(const u8* __restrict__ text, int n, int thr, bool need,
                                                    const uint16_t* __restrict__ peq16, const u8* __restrict__ seq_end,
                                                    u32 (*__restrict__ cand)[64]) {
    const int lane = lane_id();
    const int nn = need ? n : 0;
    const int nmax = (int)wave_max_u32((u32)nn);
    u32 Pv = 0xFFFFu, Mv = 0;
    int x = 16 - (thr + 1);
    u32 nz = 0, lo = 0;
    for (int c0 = 0; c0 < nmax; c0 += 16) { /* wave-uniform */
        u32 w[4] = {0, 0, 0, 0};
        if (c0 < nn) {
            const u32x4 a = load16_guard(text + c0, seq_end);
            w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
        }
        u32 bits = 0;
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const u32 c = (w[t >> 2] >> (8 * (t & 3))) & 0xFFu;
            const u32 Eq = peq16[c];
            /* (the recurrence as nine instructions, see fasta_may_trim32p.  A lane whose text has ended runs on: what it computes
               behind its last column is masked out of h below and nothing of it comes back into a column that counts) */
            const u32 s = (Eq & Pv) + Pv;
            u32 Ph = Mv | bitop3<0x01>(s, Pv, Eq);  /* Mv | ~(s | Pv | Eq) */
            const u32 Mh = bitop3<0x8C>(s, Pv, Eq); /* Pv & ((s ^ Pv) | Eq) */
            x += (int)((Ph >> 15) & 1u) - (int)((Mh >> 15) & 1u);
            Ph <<= 1;
            Pv = lshl_or<1>(Mh, bitop3<0x01>(Eq, Mv, Ph)); /* (Mh << 1) | ~(Eq | Mv | Ph) */
            Mv = bitop3<0xE0>(Ph, Eq, Mv);                 /* Ph & (Eq | Mv) */
            bits = alignbit(bits, (u32)x, 31); /* (bits << 1) | sign(x): column c0 + t ends up at bit 15 - t */
        }
        const int nv = nn - c0; /* columns of this block that exist in this lane's text */
        const u32 h = (brev32(bits) >> 16) & (nv >= 16 ? 0xFFFFu : (nv <= 0 ? 0u : ((1u << nv) - 1u)));
        if (c0 & 16) { /* wave-uniform */
            const u32 word = lo | (h << 16);
            cand[c0 >> 5][lane] = word;
            nz |= (word ? 1u : 0u) << (c0 >> 5);
        } else {
            lo = h;
        }
    }
    if (nmax > 0 && !((nmax - 1) & 16)) { /* the last block was the low half of its word */
        const int k = (nmax - 1) >> 5;
        cand[k][lane] = lo;
        nz |= (lo ? 1u : 0u) << k;
    }
    return need ? nz : 0u; /* (only the words flagged in nz are ever read) */
}

/* the global edit distance between the 16-base pattern (peq16, in LDS) and the 16 text bytes at `text`, one problem per lane:
 * lev16_win<true> with lane = read */
__device__ __forceinline__ int lev16_lanes(const u8* __restrict__ text, bool need, const uint16_t* __restrict__ peq16,
                                           const u8* __restrict__ seq_end) {
    u32 w[4] = {0, 0, 0, 0};
    if (need) {
        const u32x4 a = load16_guard(text, seq_end);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
    }
    u32 eq[16];
#pragma unroll
    for (int t = 0; t < 16; t++) eq[t] = peq16[(w[t >> 2] >> (8 * (t & 3))) & 0xFFu];
    u32 Pv = ~0u, Mv = 0;
    int score = 16;
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const u32 Eq = eq[t];
        const u32 Xv = Eq | Mv;
        const u32 Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
        u32 Ph = Mv | ~(Xh | Pv);
        u32 Mh = Pv & Xh;
        score += (int)((Ph >> 15) & 1u) - (int)((Mh >> 15) & 1u);
        Ph = (Ph << 1) | 1u;
        Mh <<= 1;
        Pv = Mh | ~(Xv | Ph);
        Mv = Ph & Xv;
    }
    return score;
}

/* The partial-pattern searches of trimBySequenceStart / End (src/adaptertrimmer.cpp:202-216, :273-286) with lane = read:
 * every lane walks ITS candidate columns (partial16_candidates) in the order the reference walks the windows and takes
 * the exact 16 x 16 distance of each -- usually one to three per flagged read instead of 184 windows.  START: windows
 * r1[p, p + 16), p = column - 15, ascending, the smallest distance wins and the smallest p among equals.  !START: windows
 * r1[rlen - 16 - p, rlen - p), p = n - 1 - column ascending (columns descending), the reference's "stop at the first
 * increase" walk over the qualifying windows.  A lane that still has candidates after PART_CAND_CAP rounds reports over =
 * true and takes the wave-per-read search instead.  r1w = the first byte of window p = 0's text row: r1 (START) / the byte
 * 16 before the end of r1 (!START), so window p starts at r1w + p / r1w - p. */
constexpr int PART_CAND_CAP = 10;
template <bool START>
__device__ __forceinline__ int partial16_resolve_lanes(const u8* __restrict__ r1w, int n, int lim, int thrP, u32 nz,
                                                       const uint16_t* __restrict__ peq16, const u8* __restrict__ seq_end,
                                                       u32 (*__restrict__ cand)[64], bool& over) {
    const int lane = lane_id();
    u32 cur = 0;
    int k = 0;
    int pos = -1, mined = 0x7fffffff;
    bool stop = false;
    over = false;
    FPL_TRIM_STAT(8, 1);
    for (int it = 0;; it++) { /* wave-uniform */
        if (cur == 0 && nz != 0) { /* this lane's next non-empty word */
            k = START ? (int)__ffs((int)nz) - 1 : 31 - (int)__clz((int)nz);
            nz &= ~(1u << k);
            cur = cand[k][lane];
        }
        const bool has = cur != 0;
        if (!wave_ballot(has)) break;
        FPL_TRIM_STAT(9, 1);
        if (it == PART_CAND_CAP) {
            over = has;
            break;
        }
        const int b = has ? (START ? (int)__ffs((int)cur) - 1 : 31 - (int)__clz((int)cur)) : 0;
        cur &= ~(1u << b);
        const int j = 32 * k + b;
        const int p = START ? j - 15 : n - 1 - j;
        const bool ok = has && p >= 0 && p < lim;
        const int ed = lev16_lanes(START ? r1w + p : r1w - p, ok, peq16, seq_end);
        if (ok && ed <= thrP) {
            if (START) {
                if (ed < mined) { /* (ascending p: the first of equals stays) */
                    mined = ed;
                    pos = p;
                }
            } else if (pos < 0 || ed <= mined) {
                pos = p;
                mined = ed;
            } else {
                stop = true;
            }
        }
        if (stop) {
            cur = 0;
            nz = 0;
        }
    }
    return pos;
}

/* the value lane j holds, as a wave-uniform value / set lane j's value */
__device__ __forceinline__ int lane_get(int v, int j) { return readlane_i32(v, j); }
__device__ __forceinline__ void lane_set(int& v, int j, int x) { v = lane_id() == j ? x : v; }

#ifndef FPL_TRIM_WAVES_PER_SIMD_BATCHED
#define FPL_TRIM_WAVES_PER_SIMD_BATCHED 5 /* (the lane-per-read phases hold a read's state and a sliding window per lane: 72 registers
                                             spilled 51 of them -- 1.80 ms per million reads at 7 waves per SIMD, 1.55 at 6, 1.48 at 5) */
#endif
/* NW: words of one-hot nibbles per window scan -- 4: both command-line adapters have <= 32 bases (DevConfig::trim_mode 1), 8: <= 64
   (trim_mode 2).  CHAIN: a FASTA chain follows (k_trim_ends<.., 2> with `pre` = the ReadState records written here): the bases the
   two command-line adapters took ride along in ReadState::pad and FilterResult::addReadTrimmed is left to the chain kernel, which
   knows the read's total (src/seprocessor.cpp:205-216) */
template <int WAVES, int NW = 4, bool CHAIN = false>
__global__ void __launch_bounds__(WAVES * 64, FPL_TRIM_WAVES_PER_SIMD_BATCHED)
k_trim_ends_batched(const u8* __restrict__ seq, const u8* __restrict__ qual, const uint64_t* __restrict__ off, u32 n_reads,
                    uint64_t n_bytes, const DevConfig* __restrict__ cfg, const DevAdapter* __restrict__ ads,
                    ReadState* __restrict__ state, long long* __restrict__ counters, u32 C, u32* __restrict__ group_ctr) {
    __shared__ TrimBlockAcc acc;
    __shared__ TrimLds<WAVES> lds;
    __shared__ int thr_lds[72]; /* DevConfig::thr[0..64] */
    __shared__ u32 cand_lds[FPL_OPT_PARTLANES ? WAVES : 1][PART_WORDS][64]; /* per lane: the columns its partial-pattern search may end at */
    PROF_INIT();
    const int lane = lane_id();
    u32(*const cand)[64] = cand_lds[FPL_OPT_PARTLANES ? wave_in_block() : 0];
    for (u32 i = threadIdx.x; i < FPL_FR_LEN; i += blockDim.x) acc.fr[i] = 0;
    for (u32 i = threadIdx.x; i < 2 * 2 * FPL_KEY_STRIDE; i += blockDim.x) acc.key[i] = 0;
    for (u32 i = threadIdx.x; i < 256; i += blockDim.x) {
        lds.peq16[0][i] = (uint16_t)ads[0].peq16_start[i];
        lds.peq16[1][i] = (uint16_t)ads[1].peq16_end[i];
        lds.peqf[0][i][0] = ads[0].peq_full[i][0];
        lds.peqf[1][i][0] = ads[1].peq_full[i][0];
    }
    for (u32 i = threadIdx.x; i < 72; i += blockDim.x) thr_lds[i] = i <= 64 ? cfg->thr[i] : 0;
    __syncthreads();
    const u8* seq_end = seq + n_bytes;
    const u8* qual_end = qual + n_bytes;
    u32* const win_s = lds.win[wave_in_block()][0];
    u32* const win_e = lds.win[wave_in_block()][1];
    u32* const win_hq = lds.win[wave_in_block()][2];
    u32* const win_tq = lds.win[wave_in_block()][3];
    u32* const win4_s = lds.win4[wave_in_block()][0];
    u32* const win4_e = lds.win4[wave_in_block()][1];
    const bool do_ad = cfg->adapter_enabled != 0;
    const bool do_start = do_ad && cfg->has_start, do_end = do_ad && cfg->has_end;
    const int ext = cfg->ext;
    const int alen0 = ads[0].len, alen1 = ads[1].len;
    constexpr int plen = FPL_PATTERN_LEN; /* (both adapters have >= 16 bases here) */
    const int thrA0 = cfg->thr[alen0], thrA1 = cfg->thr[alen1], thrP = cfg->thr[plen];
    u32 ad1h0[NW], ad1h1[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) {
        ad1h0[k] = uniform_u32(ads[0].onehot[k]);
        ad1h1[k] = uniform_u32(ads[1].onehot[k]);
    }

    /* groups of 64 reads are handed out through one counter (zeroed before the batch): the grid is what the chip holds
       at once, and no wave idles while another still has rounds to go */
    const u32 n_groups = (n_reads + 63) / 64;
    for (;;) {
        u32 grp = 0;
        if (lane == 0) grp = atomicAdd(group_ctr, 1u);
        grp = readlane_u32(grp, 0);
        if (grp >= n_groups) break;
        const u32 g0 = grp * 64;
        const int gn = (int)min(64u, n_reads - g0);
        FPL_TRIM_STAT(0, 1);
        /* lane j = read g0 + j */
        uint64_t v_o0 = 0;
        int v_l = 0;
        if (lane < gn) {
            v_o0 = off[g0 + lane];
            v_l = (int)(off[g0 + lane + 1] - v_o0);
        }
        int v_s = 0, v_e = 0, v_alive = 0, v_trim = 0;
        int v_mpos = -1; /* full match decided at this r1 position */
        int v_cand = -1; /* candidate of the window scan that still needs its edit distance */

        PROF(0) /* dequeue, offsets */
        /* ---- P1a: trimAndCut and polyX, lane = read */
        bool v_slow = false;
        {
            bool alive = false, slow = false;
            const bool wide = (cfg->cut_front && cfg->cut_front_w > LANE_SCAN_WMAX) || (cfg->cut_tail && cfg->cut_tail_w > LANE_SCAN_WMAX);
            if (!wide) {
                trim_and_cut_lanes(seq + v_o0, qual + v_o0, v_l, lane < gn, cfg, v_s, v_e, alive, slow);
                if (cfg->polyx) { /* src/seprocessor.cpp:198-201 */
                    int poly, tl;
                    const int nl = trim_polyx_lanes(seq + v_o0 + v_s, v_e - v_s, alive && !slow, cfg->polyx_min_len, poly, tl, slow);
                    if (!slow) {
                        v_e = v_s + nl;
                        if (poly >= 0) {
                            atomicAdd(&acc.fr[FPL_FR_POLYX_READS + poly], (u64)1);
                            atomicAdd(&acc.fr[FPL_FR_POLYX_BASES + poly], (u64)tl);
                        }
                    }
                }
            } else {
                slow = lane < gn;
            }
            v_alive = alive ? 1 : 0;
            v_slow = slow;
        }
        PROF(1) /* trimAndCut + polyX, lane = read */
        /* ---- P1b: the reads whose scans ran long (and every read when a cut window is wide), lanes = positions */
        for (u64 todo = wave_ballot(v_slow); todo;) {
            const int j = __ffsll(todo) - 1;
            todo &= todo - 1;
            FPL_TRIM_STAT(1, 1);
            const uint64_t o0 = readlane_u64(v_o0, j);
            const int l = lane_get(v_l, j);
            const u8* sq = seq + o0;
            const u8* ql = qual + o0;
            const int tail0 = max(0, l - TRIM_WIN);
            stage_ends<true>(win_s, win_e, win_hq, win_tq, win4_s, win4_e, sq, ql, l, tail0, seq_end, qual_end);
            const EndsView vs = {sq, (const u8*)win_s, (const u8*)win_e, tail0};
            const EndsView vq = {ql, (const u8*)win_hq, (const u8*)win_tq, tail0};
            int s, e;
            const bool alive = trim_and_cut_wave(vs, vq, l, cfg, s, e);
            if (alive && cfg->polyx) {
                int poly, tl;
                const int nl = trim_polyx_wave(vs, s, e - s, cfg->polyx_min_len, poly, tl);
                e = s + nl;
                if (poly >= 0 && lane == 0) {
                    atomicAdd(&acc.fr[FPL_FR_POLYX_READS + poly], (u64)1);
                    atomicAdd(&acc.fr[FPL_FR_POLYX_BASES + poly], (u64)tl);
                }
            }
            lane_set(v_s, j, s);
            lane_set(v_e, j, e);
            lane_set(v_alive, j, alive ? 1 : 0);
        }
        PROF(2) /* their wave-per-read fallback */
        /* ---- P1c: start adapter window scan, lane = read (searchAdapter, asRightAsPossible, :109-131) */
        if (do_start) {
            const bool act = v_alive && (v_e - v_s) >= FPL_PATTERN_LEN;
            ham_scan_lanes<true, NW>(seq + v_o0 + v_s, v_e - v_s, act, ad1h0, alen0, thrA0, seq_end, v_mpos, v_cand);
        }
        PROF(3) /* window scan, start */
        /* ---- P2: the candidates' edit distance, 64 reads at once */
        if (do_start) {
            const bool need = v_cand >= 0;
            FPL_TRIM_STAT(2, __popcll(wave_ballot(need)));
            if (wave_ballot(need)) {
                const bool ok = lev_lanes_nw<NW>(seq + v_o0 + v_s + v_cand, alen0, 0, thrA0, need, lds.peqf[0], seq_end);
                v_mpos = ok ? v_cand : v_mpos;
            }
        }
        PROF(4) /* candidate confirmation, start */
        /* ---- P3: partial-pattern search at the start for the reads without a full match (:202-216) */
        int v_ppos = -1;
        if (do_start) {
            /* (the windows the search looks at are r1[p, p + 16) for p < lim: the first lim + 15 bytes of r1) */
            const int rl = v_e - v_s;
            const bool wants = v_alive && v_mpos < 0 && rl >= FPL_PATTERN_LEN;
            bool may = wants;
            if (FPL_OPT_PARTLANES) {
                may = false;
                if (wave_ballot(wants)) {
                    const int n = min(rl, FPL_END_WINDOW);
                    const u32 nzc = partial16_candidates(seq + v_o0 + v_s, n, thrP, wants, lds.peq16[0], seq_end, cand);
                    if (wave_ballot(nzc != 0))
                        v_ppos = partial16_resolve_lanes<true>(seq + v_o0 + v_s, n, min(rl - plen, FPL_END_WINDOW - plen), thrP, nzc,
                                                               lds.peq16[0], seq_end, cand, may);
                }
            } else if (FPL_OPT_SGFILTER && wave_ballot(wants))
                may = partial16_possible(seq + v_o0 + v_s, min(rl, FPL_END_WINDOW), thrP, wants, lds.peq16[0], seq_end);
            u64 todo = wave_ballot(may);
            FPL_TRIM_STAT(3, __popcll(wave_ballot(wants)));
            FPL_TRIM_STAT(4, __popcll(todo));
            while (todo) {
                const int j = __ffsll(todo) - 1;
                todo &= todo - 1;
                const uint64_t o0 = readlane_u64(v_o0, j);
                const int s = lane_get(v_s, j), e = lane_get(v_e, j), rlen = e - s;
                const u8* sq = seq + o0;
                stage_window(win_s, sq + s, min(rlen, FPL_END_WINDOW), seq_end, nullptr);
                const Win<true> win = {nullptr, win_s, 0, rlen, nullptr};
                const int lim = min(rlen - plen, FPL_END_WINDOW - plen);
                u64 best = ~0ull;
                for (int p0 = 0; p0 < lim; p0 += 192) {
                    int pp[3], ed[3];
#pragma unroll
                    for (int u = 0; u < 3; u++) pp[u] = p0 + 64 * u + lane;
#pragma unroll
                    for (int u = 0; u < 3; u++) ed[u] = lev16_win<true>(win, min(pp[u], lim - 1), lds.peq16[0], 16, 16);
#pragma unroll
                    for (int u = 0; u < 3; u++)
                        if (pp[u] < lim && ed[u] <= thrP) {
                            const u64 k = ((u64)(u32)ed[u] << 32) | (u32)pp[u];
                            best = k < best ? k : best;
                        }
                }
                best = wave_min_u64(best);
                lane_set(v_ppos, j, best != ~0ull ? (int)(u32)best : -1);
            }
        }
        PROF(5) /* partial-pattern search, start */
        /* ---- P4: confirm the partial matches; finish the start trim (:185-193, :218-233) */
        if (do_start) {
            const int rlen = v_e - v_s;
            const int cmplen = min(v_ppos + plen, alen0);
            const bool need = v_ppos >= 0;
            bool pok = false;
            if (wave_ballot(need))
                pok = lev_lanes_nw<NW>(seq + v_o0 + v_s + v_ppos + plen - cmplen, cmplen, alen0 - cmplen, thr_lds[need ? cmplen : 0], need,
                                  lds.peqf[0], seq_end);
            int kl = 0, got = 0;
            if (v_mpos >= 0) {
                const int mp = min(v_mpos + ext, rlen - alen0);
                kl = alen0;
                v_s += min(rlen - 1, mp + alen0); /* Read::trimFront */
                got = mp + alen0;
            } else if (pok) {
                const int pos = min(v_ppos + ext, rlen - alen0);
                kl = cmplen;
                const int n = min(rlen - 1, pos + plen); /* Read::trimFront; negative erases everything */
                if (n < 0) v_s = v_e;
                else v_s += n;
                got = pos + plen;
            }
            v_trim += got;
            if (kl > 0) atomicAdd(&acc.key[(0 * 2 + 0) * FPL_KEY_STRIDE + kl], 1u);
        }
        PROF(6) /* partial confirmation, start */
        /* ---- P5: end adapter window scan, lane = read (searchAdapter, asLeftAsPossible, :84-107) */
        v_mpos = -1;
        v_cand = -1;
        if (do_end) {
            const bool act = v_alive && (v_e - v_s) >= FPL_PATTERN_LEN;
            ham_scan_lanes<false, NW>(seq + v_o0 + v_s, v_e - v_s, act, ad1h1, alen1, thrA1, seq_end, v_mpos, v_cand);
        }
        PROF(7) /* window scan, end */
        /* ---- P6 */
        if (do_end) {
            const bool need = v_cand >= 0;
            FPL_TRIM_STAT(5, __popcll(wave_ballot(need)));
            if (wave_ballot(need)) {
                const bool ok = lev_lanes_nw<NW>(seq + v_o0 + v_s + v_cand, alen1, 0, thrA1, need, lds.peqf[1], seq_end);
                v_mpos = ok ? v_cand : v_mpos;
            }
        }
        PROF(8) /* candidate confirmation, end */
        /* ---- P7: partial-pattern search walking in from the tail (:273-286) */
        v_ppos = -1;
        if (do_end) {
            /* (the windows are r1[rlen - 16 - p, rlen - p) for p < lim: the last lim + 15 bytes of r1) */
            const int rl = v_e - v_s, wlf = min(rl, FPL_END_WINDOW);
            const bool wants = v_alive && v_mpos < 0 && rl >= FPL_PATTERN_LEN;
            bool may = wants;
            if (FPL_OPT_PARTLANES) {
                may = false;
                if (wave_ballot(wants)) {
                    const u32 nzc = partial16_candidates(seq + v_o0 + v_e - wlf, wlf, thrP, wants, lds.peq16[1], seq_end, cand);
                    if (wave_ballot(nzc != 0)) {
                        const int pos = partial16_resolve_lanes<false>(seq + v_o0 + v_e - 16, wlf, min(rl - plen, FPL_END_WINDOW - plen), thrP,
                                                                       nzc, lds.peq16[1], seq_end, cand, may);
                        v_ppos = pos > 0 ? pos : -1; /* :288 strict */
                    }
                }
            } else if (FPL_OPT_SGFILTER && wave_ballot(wants))
                may = partial16_possible(seq + v_o0 + v_e - wlf, wlf, thrP, wants, lds.peq16[1], seq_end);
            u64 todo = wave_ballot(may);
            FPL_TRIM_STAT(6, __popcll(wave_ballot(wants)));
            FPL_TRIM_STAT(7, __popcll(todo));
            while (todo) {
                const int j = __ffsll(todo) - 1;
                todo &= todo - 1;
                const uint64_t o0 = readlane_u64(v_o0, j);
                const int s = lane_get(v_s, j), e = lane_get(v_e, j), rlen = e - s;
                const u8* sq = seq + o0;
                const int wl = min(rlen, FPL_END_WINDOW);
                stage_window(win_e, sq + e - wl, wl, seq_end, nullptr);
                const Win<true> win = {nullptr, win_e, rlen - wl, rlen, nullptr};
                const int lim = min(rlen - plen, FPL_END_WINDOW - plen);
                int pos = -1, mined = -1;
                bool stop = false;
                int ed3[3];
#pragma unroll
                for (int u = 0; u < 3; u++)
                    ed3[u] = lim > 0 ? lev16_win<true>(win, rlen - 16 - min(64 * u + lane, lim - 1), lds.peq16[1], 16, 16) : 0x7fffffff;
                for (int p0 = 0; p0 < lim && !stop; p0 += 64) {
                    const int p = p0 + lane;
                    const int edr = p0 == 0 ? ed3[0] : (p0 == 64 ? ed3[1] : ed3[2]);
                    const int ed = p < lim ? edr : 0x7fffffff;
                    u64 q = wave_ballot(p < lim && ed <= thrP);
                    while (q && !stop) {
                        const int b = __ffsll(q) - 1;
                        q &= q - 1;
                        const int edb = readlane_i32(ed, b);
                        if (pos < 0) {
                            pos = p0 + b;
                            mined = edb;
                        } else if (edb > mined) {
                            stop = true;
                        } else {
                            pos = p0 + b;
                            mined = edb;
                        }
                    }
                }
                lane_set(v_ppos, j, pos > 0 ? pos : -1); /* :288 strict */
            }
        }
        PROF(9) /* partial-pattern search, end */
        /* ---- P8: confirm; finish the end trim (:256-264, :288-299); the records */
        if (do_end) {
            const int rlen = v_e - v_s;
            const int cmplen = min(v_ppos + plen, alen1);
            const bool need = v_ppos >= 0;
            bool pok = false;
            if (wave_ballot(need))
                pok = lev_lanes_nw<NW>(seq + v_o0 + v_s + (rlen - plen - v_ppos), cmplen, 0, thr_lds[need ? cmplen : 0], need, lds.peqf[1],
                                  seq_end);
            int kl = 0, got = 0;
            if (v_mpos >= 0) {
                const int mp = max(0, v_mpos - ext);
                kl = alen1;
                v_e = v_s + mp; /* Read::resize */
                got = rlen - mp;
            } else if (pok) {
                const int pos = min(v_ppos + ext, rlen - plen);
                kl = cmplen;
                v_e = v_s + (rlen - plen - pos); /* Read::resize */
                got = pos + plen;
            }
            v_trim += got;
            if (kl > 0) atomicAdd(&acc.key[(1 * 2 + 1) * FPL_KEY_STRIDE + kl], 1u);
        }
        if (do_ad && !CHAIN) { /* FilterResult::addReadTrimmed */
            const u64 nt = wave_ballot(v_trim > 0);
            /* (only the reads whose total is positive are booked: with an adapter beyond 32 bases a trim at a read shorter than the
               adapter returns a NEGATIVE count -- pos = min(pos + ext, rlen - alen) -- as the reference's does, src/adaptertrimmer.cpp:224-232) */
            const u32 tb = wave_sum_u32(v_trim > 0 ? (u32)v_trim : 0u);
            if (nt && lane == 0) {
                atomicAdd(&acc.fr[FPL_FR_ADAPTER_READS], (u64)__popcll(nt));
                atomicAdd(&acc.fr[FPL_FR_ADAPTER_BASES], (u64)tb);
            }
        }
        if (lane < gn) {
            ReadState st;
            st.s = v_alive ? (u32)v_s : 0;
            st.e = v_alive ? (u32)v_e : 0;
            st.dropped = v_alive ? 0 : 1;
            st.pad = CHAIN ? (u32)v_trim : 0u; /* (what the two command-line adapters took: the chain kernel adds its own and books the read) */
            state[g0 + lane] = st;
        }
        PROF(10) /* partial confirmation, end; counters; state */
    }
    PROF_FLUSH(32);
    __syncthreads();
    long long* fr = counters + FPL_OFF_FR(C);
    long long* keyh = counters + FPL_OFF_KEYHIST(C);
    for (u32 i = threadIdx.x; i < FPL_FR_LEN; i += blockDim.x)
        if (acc.fr[i]) atomicAdd((u64*)&fr[i], acc.fr[i]);
    for (u32 i = threadIdx.x; i < 2 * 2 * FPL_KEY_STRIDE; i += blockDim.x)
        if (acc.key[i]) atomicAdd((u64*)&keyh[i], (u64)acc.key[i]);
}

/* =========================================================================================
 * k_stats: the per-cycle tables and the 5-mer counts of Stats::statRead (src/stats.cpp:265-347),
 * PRE-filter and POST-filter in ONE pass over the batch.
 *
 * The post-filter statistics of a read that survives unsplit are the statistics of its window
 * r1 = [s, e) with the cycle re-based to s (src/seprocessor.cpp:277); k_scan has already decided
 * s, e and pass/fail, so a single walk over the original read can feed both tables: a base at
 * position c goes to pre cycle c and, when s <= c < e, to post cycle c - s; a 5-mer window ending
 * at c goes to the pre table and, when it lies inside r1 (c - 4 >= s, c < e), to the post table.
 * Class / quality extraction, the packed increment and the k-mer index are computed once.
 *
 * Block (x = slice of items, y = tile of FS_T pre cycles).  LDS holds the packed 64-bit counters
 * (bits 0..21 sum of raw quality bytes, 22..35 count, 36..49 count(q>='5'), 50..63 count(q>='?');
 * a slice has <= 16383 items so no field overflows) of FS_T pre cycles and of the FS_T + FS_SMAX post
 * cycles [tile_start - FS_SMAX, tile_start + FS_T) they can map to (reads whose front trim exceeds
 * FS_SMAX, and the fragments of split reads, come through the EXTRA list instead: post only, cycle =
 * position).  A lane takes 8 consecutive bytes; the LDS slot of local cycle x is (x%8)*(n/8) + x/8, so
 * the 64 lanes of one ds_add_u64 hit 64 consecutive slots.  At the end the block stores its tables as
 * one slab of the scratch buffer (plain coalesced stores); k_stats_reduce sums the slabs per tile.
 * ======================================================================================= */
constexpr int FS_T = 512;
constexpr int FS_SMAX = 96;
constexpr int FS_PT = FS_T + FS_SMAX;          /* post cycles per slab */
constexpr int FS_SLAB = 8 * FS_T + 8 * FS_PT;  /* u64 per slab: pre table, then post table */
/* In LDS the two tables are interleaved per base class -- [cls][FS_T pre cells | FS_PT post cells] -- so that one
   multiply-add per byte gives the pre address and the post address is a wave-uniform offset away from it. */
constexpr int FS_STRIDE = FS_T + FS_PT;
__device__ __forceinline__ u32 fs_pre_cell(u32 i) { return (i / FS_T) * FS_STRIDE + (i % FS_T); }           /* i = cls * FS_T + x */
__device__ __forceinline__ u32 fs_post_cell(u32 j) { return (j / FS_PT) * FS_STRIDE + FS_T + (j % FS_PT); } /* j = cls * FS_PT + y */
constexpr u32 CS_MAX_ITEMS_PER_SLICE = 16383;
#ifndef FPL_ABL
#define FPL_ABL 0 /* profiling only (-DFPL_ABL=bits): 1 no pre-table, 2 no post-table, 4 no 5-mer updates in k_stats */
#endif
constexpr int CS_GROUP = 4; /* items whose loads are in flight together, per wave */
constexpr u32 PLAN_TO_POST = 1u; /* ReadState::pad bit: the read has ONE output read, it passes, and it starts <= FS_SMAX bases into the read
                                    (ReadState::s / e then hold that output read's window) */

/* 2-bit base codes of the four bytes of d, one per byte (Stats::base2val: A0 T/U1 C2 G3; other letters give
   some code and are caught by the validity mask) */
__device__ __forceinline__ u32 kmer_codes(u32 d) { return (d & 0x02020202u) | ((d >> 2) & 0x01010101u); }
/* byte 3 of the result = the four codes packed v0<<6 | v1<<4 | v2<<2 | v3 (v0 = lowest byte = earliest base) */
__device__ __forceinline__ u32 kmer_pack(u32 v) {
    const u32 x = lshl_or<10>(v, v);
    return lshl_or<20>(x, x);
}
/* the same four codes packed into the LOW byte, by one v_dot4 (weights 64, 16, 4, 1) */
__device__ __forceinline__ u32 kmer_pack_dot(u32 v) { return udot4(v, 0x01041040u, 0u); }
/* A lane's 5-mer stream: the packed codes of the four bases in front of its eight (bits 16..23), of its first four (8..15) and
   of its last four (0..7); the window that ends at byte k is the 10-bit field at bit 2 * (7 - k).  The four bases in front are
   the previous lane's last four: its finished pack comes over with one DPP move (lane 0: pack_h0, wave-uniform).  Bits 24..31
   hold the previous lane's first pack, which no window reads. */
__device__ __forceinline__ u32 kmer_stream(u32 v0, u32 v1, u32 pack_h0) {
    u32 p0 = kmer_pack_dot(v0), p1 = kmer_pack_dot(v1);
    dot_settle(p0, p1);
    const u32 W01 = lshl_or<8>(p0, p1);
    return lshl_or<16>(wave_prev_u32(W01, pack_h0), W01);
}
/* d = four bases, mapped = the letters their 2-bit codes stand for (A T C G): bit i of the result is set when base i is
   not one of A, T, U, C, G.  A mapped T also admits U (0x54 ^ 0x55 = 1; T is the only mapped letter with bit 4). */
__device__ __forceinline__ u32 invalid_nibble(u32 mapped, u32 d) {
    const u32 x = (mapped ^ d) & ~((mapped >> 4) & 0x01010101u);
    const u32 nz = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u; /* bit 7 of every non-zero byte */
    const u32 y = lshl_or<7>(nz, nz);                                      /* gather bits 7,15,23,31 into 28..31 */
    return lshl_or<14>(y, y) >> 28;
}
/* bits [lo, hi) of a byte, lo/hi clamped to 0..8 */
__device__ __forceinline__ u32 range_mask8(int lo, int hi) {
    lo = lo < 0 ? 0 : (lo > 8 ? 8 : lo);
    hi = hi < 0 ? 0 : (hi > 8 ? 8 : hi);
    return hi > lo ? (((1u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
}

struct u32x2 {
    u32 x, y;
};
__device__ __forceinline__ u32x2 load8_guard(const u8* p, const u8* end) {
    u32x2 v = {0, 0};
    if (p + 8 <= end) {
        __builtin_memcpy(&v, p, 8);
        return v;
    }
    v.x = load4_guard(p, end);
    v.y = load4_guard(p + 4, end);
    return v;
}

/* (two blocks of 16 waves per CU need <= 64 VGPRs: 8 waves per SIMD) */
/* packed counter -> its four fields */
__device__ __forceinline__ void fs_unpack_add(u64 v, u64& qsum, u64& cnt, u64& q20, u64& q30) {
    qsum += v & 0x3FFFFF;
    cnt += (v >> 22) & 0x3FFF;
    q20 += (v >> 36) & 0x3FFF;
    q30 += v >> 50;
}
/* Hand the packed tables over as they are: plain coalesced stores into one slab of the scratch buffer;
   k_stats_reduce sums the slabs of a tile and unpacks them.  (Flushing with global atomics instead cost
   more than the counting itself.)  The 5-mer counts are few: atomics. */
template <bool EXTRA>
__device__ __forceinline__ void fs_hand_over(const u64* tbl, const u32* kpre, const u32* kpost,
                                             u64* __restrict__ scratch, u8* __restrict__ flags, size_t slab,
                                             long long* kg0, long long* kg1) {
    u64* dst = scratch + slab * FS_SLAB;
    if (!EXTRA)
        for (u32 i = threadIdx.x; i < 8 * FS_T; i += blockDim.x) dst[i] = tbl[fs_pre_cell(i)];
    for (u32 i = threadIdx.x; i < 8 * FS_PT; i += blockDim.x) dst[8 * FS_T + i] = tbl[fs_post_cell(i)];
    if (threadIdx.x == 0) {
        flags[gridDim.y + slab] = 1;
        flags[blockIdx.y] = 1; /* this tile has at least one slab */
    }
    for (u32 i = threadIdx.x; i < 1024 && !(FPL_ABL & 64); i += blockDim.x) {
        const u32 both = kpost[i], pre_only = EXTRA ? 0u : kpre[i]; /* (EXTRA: kpost is post-only) */
        if (!EXTRA && both + pre_only) atomicAdd((u64*)&kg0[i], (u64)both + pre_only);
        if (both) atomicAdd((u64*)&kg1[i], (u64)both);
    }
}

template <int WAVES, bool EXTRA>
__global__ void __launch_bounds__(WAVES * 64, (2 * WAVES + 3) / 4)
k_stats(const u8* __restrict__ seq, const u8* __restrict__ qual, uint64_t n_bytes,
        const uint64_t* __restrict__ item_off, const u32* __restrict__ item_len, const u32* __restrict__ item_cyc,
        const ReadState* __restrict__ plan,
        u32 n_items, const u32* __restrict__ n_items_dev, u32 items_per_slice, u32 n_slices, u32 max_acc,
        long long* __restrict__ counters, u64* __restrict__ scratch, u8* __restrict__ flags, u32 C) {
    /* main pass: items are the reads (CSR offsets + plan).  EXTRA: items are the post-only fragment list
       k_scan built (count in n_items_dev), cycle = position in the fragment. */
    /* one LDS array carved by hand: the 5-mer tables come first so that their (data-dependent) addresses
       take the table choice as an instruction offset */
    __shared__ u64 lds_all[1024 + 256 + 8 * FS_STRIDE];
    static_assert(sizeof(u64) * (1024 + 256 + 8 * FS_STRIDE) <= 81920, "two blocks per CU");
    u32* const kmer = (u32*)lds_all; /* [0,1024): 5-mers counted pre-filter only; [1024,2048): pre- AND post-filter */
    u64* const inc_of = lds_all + 1024; /* the packed increment of every quality byte: one LDS read instead of two compares,
                                          two selects and two ORs per base */
    u64* const tbl = inc_of + 256;      /* [8][FS_STRIDE] */
    for (u32 q = threadIdx.x; q < 256; q += blockDim.x)
        inc_of[q] = (u64)q | (1ull << 22) | ((u64)(q >= '5') << 36) | ((u64)(q >= '?') << 50);
    u32* const kpre = kmer;
    u32* const kpost = kmer + 1024;
    u32& any_work = kpost[0]; /* (2 x 81920 bytes of LDS per CU: no room for one more word) */
    const int lane = lane_id();
    if (EXTRA) n_items = *n_items_dev;
    if ((FPL_ABL & 32) && !EXTRA && blockIdx.y >= 48) return;
    const u32 tile_start = blockIdx.y * FS_T;
    const u8* seq_end = seq + n_bytes;
    const u8* qual_end = qual + n_bytes;
    const u32 c0 = tile_start + 8 * lane; /* position of this lane's first byte */
    /* main pass: one slice per block (gridDim.x == n_slices), one slab per (tile, slice).
       EXTRA pass: the list is usually tiny and its length is only known on the device, so a fixed grid of
       blocks walks short slices, keeps accumulating in the same tables (while the 14-bit fields allow:
       max_acc items) and hands over ONE slab per (tile, blockIdx.x) at the end. */
    long long* kg0 = counters + FPL_OFF_PRE(C) + FPL_ST_KMER(C);
    long long* kg1 = counters + FPL_OFF_POST(C) + FPL_ST_KMER(C);
    bool ready = false; /* tables zeroed and in use (block-uniform) */
    u32 acc = 0;        /* items the tables may already hold */
    for (u32 slice = blockIdx.x; EXTRA || slice < n_slices; slice += gridDim.x) {
    const u32 i_begin = slice * items_per_slice;
    if (i_begin >= n_items) break;
    const u32 i_end = min(n_items, i_begin + items_per_slice);
    /* most (slice, tile) blocks beyond the typical item length have nothing to count: find out before
       paying for the tables */
    u32 aw;
    if (EXTRA && ready) {
        aw = 1; /* any_work aliases a live counter now; the tables are paid for anyway */
    } else {
        if (threadIdx.x == 0) any_work = 0;
        __syncthreads();
        bool mine = false;
        for (u32 it = i_begin + threadIdx.x; it < i_end; it += blockDim.x) {
            u32 L = EXTRA ? item_len[it] : (u32)(item_off[it + 1] - item_off[it]);
            if (EXTRA && item_cyc) L += item_cyc[it] & 0x7FFFFFFFu;
            mine = mine || (L > tile_start);
        }
        if (wave_ballot(mine) && lane == 0) any_work = 1;
        __syncthreads();
        aw = any_work; /* the word is about to be reused as a counter: read, then barrier */
        __syncthreads();
    }
    if (!aw) continue;
    if (EXTRA && ready && acc + items_per_slice > max_acc) {
        /* the packed fields could overflow: empty the tables with atomics (never seen outside tests) */
        __syncthreads();
        for (u32 i = threadIdx.x; i < 8 * FS_PT; i += blockDim.x) {
            const u64 v = tbl[fs_post_cell(i)];
            if (!v) continue;
            const u32 cls = i / FS_PT, slot = i % FS_PT;
            const u32 pl = (slot % (FS_PT / 8)) * 8 + slot / (FS_PT / 8);
            const u32 c = tile_start + pl - FS_SMAX; /* EXTRA items start at cycle 0: pl >= FS_SMAX */
            if (c >= C) continue;
            u64 qsum = 0, cnt = 0, q20 = 0, q30 = 0;
            fs_unpack_add(v, qsum, cnt, q20, q30);
            long long* st = counters + FPL_OFF_POST(C);
            atomicAdd((u64*)&st[FPL_ST_CYC(c, 0, cls)], cnt);
            atomicAdd((u64*)&st[FPL_ST_CYC(c, 1, cls)], qsum - 33 * cnt);
            atomicAdd((u64*)&st[FPL_ST_CYC(c, 2, cls)], q20);
            atomicAdd((u64*)&st[FPL_ST_CYC(c, 3, cls)], q30);
        }
        for (u32 i = threadIdx.x; i < 1024; i += blockDim.x)
            if (kpost[i]) atomicAdd((u64*)&kg1[i], (u64)kpost[i]);
        __syncthreads();
        ready = false;
    }
    if (!ready) {
        if (!EXTRA)
            for (u32 i = threadIdx.x; i < 8 * FS_T; i += blockDim.x) tbl[fs_pre_cell(i)] = 0;
        for (u32 i = threadIdx.x; i < 8 * FS_PT; i += blockDim.x) tbl[fs_post_cell(i)] = 0;
        for (u32 i = threadIdx.x; i < 1024; i += blockDim.x) kpre[i] = kpost[i] = 0;
        __syncthreads();
        ready = true;
        acc = 0;
    }
    acc += items_per_slice;

    for (u32 ib = i_begin + 64 * wave_in_block(); ib < i_end; ib += 64 * WAVES) {
        const u32 it = ib + lane;
        u32 L = 0, S = 0, E = 0, TP = 0;
        uint64_t st = 0;
        if (it < i_end) {
            st = item_off[it];
            if (EXTRA) {
                /* a post-only item starts at cycle cyc0 (non-zero for the pieces of a --mask'ed fragment): treat it
                   as a virtual read that begins cyc0 bytes earlier and whose body is [cyc0, cyc0 + len); TP bit 1
                   marks a piece whose bases were overwritten with N */
                const u32 cy = item_cyc ? item_cyc[it] : 0u;
                S = cy & 0x7FFFFFFFu;
                L = item_len[it] + S;
                E = L;
                TP = 1u | ((cy >> 31) << 1);
                st -= S;
            } else {
                L = (u32)(item_off[it + 1] - st);
                const ReadState ps = plan[it];
                S = ps.s;
                E = ps.e;
                TP = ps.pad & PLAN_TO_POST;
            }
        }
        u64 m = wave_ballot(L > tile_start && (!EXTRA || S < tile_start + FS_T));
        /* groups of CS_GROUP items: all their loads are issued before the first one is counted */
        while (m) {
            u32x2 svG[CS_GROUP], qvG[CS_GROUP];
            u32 haloG[CS_GROUP], LG[CS_GROUP], SG[CS_GROUP], EG[CS_GROUP], TG[CS_GROUP];
#pragma unroll
            for (int g = 0; g < CS_GROUP; g++) {
                svG[g] = {0, 0};
                qvG[g] = {0, 0};
                haloG[g] = 0;
                LG[g] = SG[g] = EG[g] = TG[g] = 0;
                if (m) { /* wave-uniform; readlane keeps the item's fields in scalar registers */
                    const int bit = __ffsll(m) - 1;
                    m &= m - 1;
                    LG[g] = readlane_u32(L, bit);
                    SG[g] = readlane_u32(S, bit);
                    EG[g] = readlane_u32(E, bit);
                    TG[g] = readlane_u32(TP, bit);
                    uint64_t start = readlane_u64(st, bit);
                    if (FPL_ABL & 8) start &= ~(uint64_t)127; /* (timing experiment: rows on cache-line boundaries) */
                    if ((FPL_ABL & 16) ? lane == 0 : LG[g] > c0 && (!EXTRA || c0 + 8 > SG[g])) {
                        svG[g] = load8_guard(seq + start + c0, seq_end);
                        qvG[g] = load8_guard(qual + start + c0, qual_end);
                    }
                    /* (EXTRA: the four bases in front of the tile matter as soon as ONE of them belongs to the body --
                       a body that starts 1..3 bases before a tile boundary still has windows reaching across it) */
                    if (lane == 0 && tile_start >= 4 && (!EXTRA || tile_start > SG[g]))
                        haloG[g] = load4_guard(seq + start + tile_start - 4, seq_end);
                }
            }
#pragma unroll
            for (int g = 0; g < CS_GROUP; g++) {
                const u32 itemL = uniform_u32(LG[g]);
                if (itemL <= tile_start) continue; /* wave-uniform: empty slot of the last group */
                const u32 tflags = uniform_u32(TG[g]);
                const bool tp = (tflags & 1u) != 0;
                const bool allN = EXTRA && (tflags & 2u); /* Read::maskRegionWithN: the bases read as N, qualities stay */
                const u32 sw[2] = {allN ? 0x4E4E4E4Eu : svG[g].x, allN ? 0x4E4E4E4Eu : svG[g].y};
                const u32 qw[2] = {qvG[g].x, qvG[g].y};
                const int s = (int)uniform_u32(SG[g]), e = (int)uniform_u32(EG[g]);
                const int nvalid = itemL > c0 ? (int)min(8u, itemL - c0) : 0; /* bytes of the item in this lane */
                /* the four bases in front of this lane's chunk: previous lane's last dword */
                const u32 up = FPL_OPT_DPPPREV ? wave_prev_u32(sw[1], 0u) : shfl_up_u32(sw[1], 1);
                const bool have_halo = lane > 0 || tile_start >= 4;
                const u32 halo = allN ? 0x4E4E4E4Eu : (lane > 0 ? up : haloG[g]);
                /* 5-mers, twelve bases at once: 2-bit codes (Stats::base2val: A0 T1 C2 G3) packed earliest base
                   highest, so the window that ends at byte k is a 10-bit field of W */
                const u32 vh = kmer_codes(halo), v0 = kmer_codes(sw[0]), v1 = kmer_codes(sw[1]);
                const u32 W = perm_b32(kmer_pack(vh), perm_b32(kmer_pack(v0), kmer_pack(v1), 0x0c0c0703u), 0x0c070100u);
                /* okmask bit k: the window ending at byte k holds five valid bases.  All-ACGT rows (the rule)
                   are recognised by mapping the codes back to letters; anything else takes the exact count */
                u32 okmask;
                {
                    const u32 m0 = perm_lo(0x47435441u, v0), m1 = perm_lo(0x47435441u, v1), mh = perm_lo(0x47435441u, vh);
                    u32 bad = (m0 ^ sw[0]) | (m1 ^ sw[1]);
                    if (have_halo) bad |= mh ^ halo;
                    if (!wave_ballot(nvalid > 0 && bad != 0)) {
                        okmask = have_halo ? 0xFFu : 0xF0u;
                    } else {
                        /* the exact rule (Stats::base2val: A, T, U, C, G are the valid bases) without a per-byte
                           chain: bit j of inv = base j of [halo | this lane's eight] is invalid; a window is counted
                           when none of its five bases is */
                        const u32 ih = have_halo ? invalid_nibble(mh, halo) : 0xFu;
                        const u32 inv = lshl_or<8>(invalid_nibble(m1, sw[1]), lshl_or<4>(invalid_nibble(m0, sw[0]), ih));
                        const u32 r = inv | (inv >> 1) | (inv >> 2) | (inv >> 3) | (inv >> 4);
                        okmask = ~r & 0xFFu;
                    }
                    okmask &= (1u << nvalid) - 1u;
                }
                /* post cell of local byte k: x = 8*lane + k + (FS_SMAX - s); slot(x) = (x%8)*(FS_PT/8) + x/8
                   (EXTRA items keep their cycle: the body starts at s but is not re-based) */
                const int u0 = EXTRA ? FS_SMAX : FS_SMAX - s;
                const int p0 = (int)c0; /* position of byte 0 of this lane */
                /* one byte.  DO_PRE / DO_POST / FULL are compile-time; FULL: all 8 bytes of every lane belong to
                   the item and (with DO_POST) to r1, 5-mer windows included.  Otherwise bit k of bodymask /
                   kbodymask says whether byte k (its window) lies inside r1. */
/* the increment of byte k + 1 is fetched before the updates of byte k are issued: LDS operations of a wave finish in
   order, so a table read issued AFTER three atomics (and waited for) would stall the wave for all of them, eight times
   per row */
#define FPL_FS_Q(k) ((qw[(k) >> 2] >> (8 * ((k)&3))) & 0xFF)
                u64 inc_n = inc_of[FPL_FS_Q(0)];
#define FPL_FS_BYTE(k, DO_PRE, DO_POST, FULL)                                                                     \
    if (FULL || (k) < nvalid) {                                                                                   \
        const u32 bb = (sw[(k) >> 2] >> (8 * ((k)&3))) & 0xFF;                                                    \
        const u32 cell = mad_u24(bb & 7u, FS_STRIDE, lane);                                                       \
        if (DO_PRE && !(FPL_ABL & 1)) atomicAdd(&tbl[cell + (k)*64], inc);                                        \
        if (DO_POST && !(FPL_ABL & 2)) {                                                                          \
            const int uu = u0 + (k); /* wave-uniform */                                                           \
            if (FULL || ((bodymask >> (k)) & 1u))                                                                 \
                atomicAdd(&tbl[cell + (FS_T + (uu & 7) * (FS_PT / 8) + (uu >> 3))], inc);                         \
        }                                                                                                         \
        /* ONE 5-mer update per byte: a window counted post-filter is also counted pre-filter, so kpost   \
           holds the windows inside r1 (both tables) and kpre the pre-only rest */                              \
        const u32 kidx = (W >> (2 * (7 - (k)))) & 0x3FFu;                                                         \
        const u32 kval = (okmask >> (k)) & 1u;                                                                    \
        if (FPL_ABL & 4) {                                                                                        \
        } else if (DO_PRE && DO_POST)                                                                             \
            atomicAdd(&kmer[(FULL ? 1024u : (((kbodymask >> (k)) & 1u) << 10)) + kidx], kval);                    \
        else if (DO_PRE)                                                                                          \
            atomicAdd(&kmer[kidx], kval);                                                                         \
        else                                                                                                      \
            atomicAdd(&kmer[1024u + kidx], FULL ? kval : (kval & (kbodymask >> (k))));                            \
    }
                if (EXTRA) {
                    /* post only; cycle = position in the (virtual) item; a window needs 4 predecessors inside
                       the body [s, itemL) */
                    const u32 bodymask = range_mask8(s - p0, e - p0), kbodymask = range_mask8(s + 4 - p0, e - p0);
                    if ((int)tile_start >= s + 4 && itemL >= tile_start + FS_T) {
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            const u64 inc = inc_n;
                            if (k < 7) inc_n = inc_of[FPL_FS_Q(k + 1)];
                            FPL_FS_BYTE(k, false, true, true)
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            const u64 inc = inc_n;
                            if (k < 7) inc_n = inc_of[FPL_FS_Q(k + 1)];
                            FPL_FS_BYTE(k, false, true, false)
                        }
                    }
                } else if (tp && (int)tile_start >= s + 4 && (int)(tile_start + FS_T) <= e) {
                    /* wave-uniform: the whole tile lies inside r1 (and inside the read) */
                    const u32 bodymask = 0xFFu, kbodymask = 0xFFu;
                    (void)bodymask;
                    (void)kbodymask;
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                            const u64 inc = inc_n;
                            if (k < 7) inc_n = inc_of[FPL_FS_Q(k + 1)];
                            FPL_FS_BYTE(k, true, true, true)
                        }
                } else if (tp) { /* a tile that straddles an end of r1 */
                    const u32 bodymask = range_mask8(s - p0, e - p0), kbodymask = range_mask8(s + 4 - p0, e - p0);
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                            const u64 inc = inc_n;
                            if (k < 7) inc_n = inc_of[FPL_FS_Q(k + 1)];
                            FPL_FS_BYTE(k, true, true, false)
                        }
                } else { /* pre only (dropped, failed, split or far-trimmed read) */
                    const u32 bodymask = 0, kbodymask = 0;
                    (void)bodymask;
                    (void)kbodymask;
                    if (itemL >= tile_start + FS_T) {
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            const u64 inc = inc_n;
                            if (k < 7) inc_n = inc_of[FPL_FS_Q(k + 1)];
                            FPL_FS_BYTE(k, true, false, true)
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            const u64 inc = inc_n;
                            if (k < 7) inc_n = inc_of[FPL_FS_Q(k + 1)];
                            FPL_FS_BYTE(k, true, false, false)
                        }
                    }
                }
#undef FPL_FS_BYTE
#undef FPL_FS_Q
            }
        }
    }
    if (EXTRA) continue;
    __syncthreads();
    fs_hand_over<EXTRA>(tbl, kpre, kpost, scratch, flags, (size_t)blockIdx.y * n_slices + slice, kg0, kg1);
    __syncthreads(); /* (the tables would be reused by a next slice) */
    ready = false;
    }
    if (EXTRA && ready) {
        __syncthreads();
        fs_hand_over<EXTRA>(tbl, kpre, kpost, scratch, flags, (size_t)blockIdx.y * n_slices + blockIdx.x, kg0,
                            kg1);
    }
}

/* Sum the per-(tile, slice) slabs of one k_stats launch into the per-cycle counters.  Block (x = chunk
 * of 256 cells, y = tile); cells 0 .. 8*FS_T-1 are the tile's pre cycles, the next 8*FS_T its post cycles
 * [tile*FS_T, (tile+1)*FS_T) -- which receive contributions from this tile's slabs and from the next
 * tile's (whose post window starts FS_SMAX cycles earlier).  Every counter has exactly one owner, so
 * the update is a plain read-modify-write. */
__global__ void __launch_bounds__(256)
k_stats_reduce(const u64* __restrict__ scratch, const u8* __restrict__ flags, u32 n_slices, u32 n_tiles,
               long long* __restrict__ counters, u32 C, int with_pre) {
    const u32 tile = blockIdx.y;
    const u32 cell = blockIdx.x * 256 + threadIdx.x; /* [0, 8*FS_T): pre, [8*FS_T, 16*FS_T): post */
    const bool is_post = cell >= 8 * FS_T;
    const u8* slab_flags = flags + n_tiles;
    const bool here = flags[tile] != 0, next = tile + 1 < n_tiles && flags[tile + 1] != 0;
    if (!here && !next) return; /* block-uniform */
    const u32 cc = is_post ? cell - 8 * FS_T : cell;
    /* (threads in slot order, as in k_stats_reduce_sorted: a wave's loads of a slab are consecutive bytes, pre and post) */
    const u32 cls = cc / FS_T, tj = cc % FS_T;
    const u32 x = (tj & 63u) * 8u + (tj >> 6); /* x = cycle within the tile */
    u64 qsum = 0, cnt = 0, q20 = 0, q30 = 0;
    if (!is_post) {
        if (!with_pre || !here) return;
        const u32 slot = (x & 7) * 64 + (x >> 3);
        for (u32 sl = 0; sl < n_slices; sl++) {
            const size_t slab = (size_t)tile * n_slices + sl;
            if (!slab_flags[slab]) continue;
            fs_unpack_add(scratch[slab * FS_SLAB + cls * FS_T + slot], qsum, cnt, q20, q30);
        }
    } else {
        /* post cycle p = tile*FS_T + x: local index x + FS_SMAX in this tile's slabs, x + FS_SMAX - FS_T in the next tile's */
        const u32 pl0 = x + FS_SMAX;
        const u32 slot0 = (pl0 & 7) * (FS_PT / 8) + (pl0 >> 3);
        for (u32 sl = 0; here && sl < n_slices; sl++) {
            const size_t slab = (size_t)tile * n_slices + sl;
            if (!slab_flags[slab]) continue;
            fs_unpack_add(scratch[slab * FS_SLAB + 8 * FS_T + cls * FS_PT + slot0], qsum, cnt, q20, q30);
        }
        if (x + FS_SMAX >= FS_T && next) {
            const u32 pl1 = x + FS_SMAX - FS_T;
            const u32 slot1 = (pl1 & 7) * (FS_PT / 8) + (pl1 >> 3);
            for (u32 sl = 0; sl < n_slices; sl++) {
                const size_t slab = (size_t)(tile + 1) * n_slices + sl;
                if (!slab_flags[slab]) continue;
                fs_unpack_add(scratch[slab * FS_SLAB + 8 * FS_T + cls * FS_PT + slot1], qsum, cnt, q20, q30);
            }
        }
    }
    const u32 c = tile * FS_T + x;
    if (cnt && c < C) {
        long long* st = counters + (is_post ? FPL_OFF_POST(C) : FPL_OFF_PRE(C));
        st[FPL_ST_CYC(c, 0, cls)] += (long long)cnt;
        st[FPL_ST_CYC(c, 1, cls)] += (long long)qsum - 33ll * (long long)cnt; /* += qual - 33 */
        st[FPL_ST_CYC(c, 2, cls)] += (long long)q20;
        st[FPL_ST_CYC(c, 3, cls)] += (long long)q30;
    }
}

/* =========================================================================================
 * k_scan
 * ======================================================================================= */
#ifndef FPL_HIST_COPIES
#define FPL_HIST_COPIES 8
#endif
constexpr int HIST_COPIES = FPL_HIST_COPIES; /* per-wave quality histogram: 128 bins x this many lane-copies */
/* k_scan blocks (4 waves) a CU holds: LDS-bound, 10 KiB (16 copies) or 6 KiB (8 copies) per wave */
#ifndef FPL_SCAN_BLOCKS
#define FPL_SCAN_BLOCKS (FPL_HIST_COPIES == 16 ? 3 : 5)
#endif
constexpr int SCAN_BLOCKS_PER_CU = FPL_SCAN_BLOCKS;
constexpr int SC_CHUNK = 32;    /* bases per lane per tile in the bit-sliced scan */
constexpr int SC_FBUF = 32;     /* fragments a wave gathers per reservation in the global fragment list */
constexpr int SC_LANES_HAM = 62; /* lanes 62/63 only provide the plane words the last windows reach into */

/* per-wave LDS of k_scan */
/* The quality histogram of the range being scanned (128 bins x HIST_COPIES lane-copies, all zero between uses) lives
 * in its own array, one 4 KiB-aligned slice per wave (k_scan: hist_all): with the bin index in address bits 5..11 and
 * the lane's copy in bits 2..4, the address of a byte's counter is (shifted byte & 0xfe0) | lane base -- one
 * v_and_or_b32 after the shift instead of an and and an add. */
struct alignas(16) ScanWaveLds {
    u32 planes[5][64];            /* letter bit-planes of the current tile: [A,C,T,G][chunk]; row 4 stays zero */
    u32 ehist[128];               /* quality histogram of the trimmed-off ends of the read; all zero between uses */
    uint64_t fbuf_off[SC_FBUF];   /* passing fragments waiting for a slot in the global list */
    u32 fbuf_len[SC_FBUF];
};

struct ScanBlockAcc {
    u64 bqh[2][128];   /* mBaseQualHistogram        pre / post */
    u64 medh[2][128];  /* mMedianReadQualHistogram  pre / post */
    u64 medb[2][128];  /* mMedianReadQualBases      pre / post */
    u64 reads[2], lensum[2];
    u64 fr[FPL_FILTER_RESULT_TYPES];
};

struct RangeSums {
    u32 lowq, nn, totq, diff;
};

__device__ __forceinline__ void hist_zero(u32* __restrict__ h) {
    const int lane = lane_id();
    wave_sync();
    for (int i = lane; i < 128 * HIST_COPIES; i += 64) h[i] = 0;
    wave_sync();
}

/* totals of bins 2*lane and 2*lane+1 over the lane copies; leaves the histogram zeroed for its next
 * user.  A lane owns 2 * HIST_COPIES consecutive words = HIST_COPIES / 2 slots of 16 bytes and visits them in
 * an order rotated by its lane number, so the 64 lanes of one ds_read_b128 / ds_write_b128 spread over the banks. */
__device__ __forceinline__ void hist_totals(u32* __restrict__ h, u32& t0, u32& t1) {
    static_assert(HIST_COPIES == 8 || HIST_COPIES == 16, "slot arithmetic below");
    constexpr int NSLOT = HIST_COPIES / 2; /* 16-byte slots per lane: the first half is bin 2*lane, the rest 2*lane+1 */
    const int lane = lane_id();
    t0 = 0;
    t1 = 0;
    wave_sync();
    u32x4* row = (u32x4*)(h + 2 * HIST_COPIES * lane);
#pragma unroll
    for (int k = 0; k < NSLOT; k++) {
        const int slot = (k + lane) & (NSLOT - 1);
        const u32x4 v = row[slot];
        const u32x4 z = {0, 0, 0, 0};
        row[slot] = z;
        const u32 sum = v.x + v.y + v.z + v.w;
        t0 += slot < NSLOT / 2 ? sum : 0u;
        t1 += slot < NSLOT / 2 ? 0u : sum;
    }
    wave_sync(); /* later atomics of other lanes must not overtake these accesses */
}

/* the two passFilter sums over the qualities of a range from its histogram (t0 / t1 = this lane's bins 2 * lane and
   2 * lane + 1): bases with a quality below qq, and the sum of the qualities (src/filter.cpp:27-39) */
__device__ __forceinline__ void hist_quality_sums(u32 t0, u32 t1, int qq, u32& lowq, u32& totq) {
    const int b0 = 2 * lane_id();
    lowq = wave_sum_u32((b0 < qq ? t0 : 0u) + (b0 + 1 < qq ? t1 : 0u));
    totq = wave_sum_u32(t0 * (u32)b0 + t1 * (u32)(b0 + 1));
}

/* median as Stats::statRead computes it, src/stats.cpp:352-363: smallest q with
 * cumulative count > len/2.  t0/t1 = this lane's two bins; len > 0. */
__device__ __forceinline__ int hist_median(u32 t0, u32 t1, u32 len) {
    const u32 v = t0 + t1;
    const u32 incl = wave_scan_incl_u32(v);
    const u32 excl = incl - v;
    const u32 half = len >> 1;
    const u64 m = wave_ballot(incl > half);
    const int L = __ffsll(m) - 1; /* m != 0 because the total is len > half */
    const int mine = 2 * lane_id() + ((excl + t0 > half) ? 0 : 1);
    return readlane_i32(mine, L);
}

/* Hamming distances between the adapter and the 16 windows starting at this lane's 16 bytes.
 * p points at the lane's first byte, cur holds its 16 bytes; acc = 16 packed byte counters. */
__device__ __forceinline__ void hamming16(const u8* __restrict__ p, const u8* __restrict__ end, u32x4 cur,
                                          const u32* __restrict__ adw, int alen, u32 acc[4]) {
    acc[0] = acc[1] = acc[2] = acc[3] = 0;
    for (int m0 = 0; m0 < alen; m0 += 16) {
        const u32x4 nxt = load16_guard(p + m0 + 16, end);
        const u32 w[8] = {cur.x, cur.y, cur.z, cur.w, nxt.x, nxt.y, nxt.z, nxt.w};
#pragma unroll
        for (int t = 0; t < 16; t++) {
            if (m0 + t < alen) { /* wave-uniform */
                const u32 a = adw[m0 + t];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const u32 x = (t & 3) ? alignbyte(w[(t >> 2) + j + 1], w[(t >> 2) + j], t & 3) : w[(t >> 2) + j];
                    acc[j] += nonzero_bytes01(x ^ a);
                }
            }
        }
        cur = nxt;
    }
}

/*
 * One pass over bytes [a, b) of a read (rb/qb = first base / quality of the original read):
 *   - quality histogram into h (LDS, this wave's),
 *   - the sums Filter::passFilter needs (src/filter.cpp:27-39, 67-81), when SUMS,
 *   - when NADS == 2, the Hamming argmin of both middle-adapter scans
 *     (searchAdapter default mode, src/adaptertrimmer.cpp:133-151) over positions
 *     p in [0, (b-a) - alen), as keys (mismatches << 32 | p), ~0 when nothing was tested.
 */
template <bool SUMS, bool HAM>
__device__ inline void range_scan_bytes(const u8* __restrict__ rb, const u8* __restrict__ qb, int a, int b,
                                  const u8* __restrict__ seq_end, const u8* __restrict__ qual_end,
                                  u32* __restrict__ h, int qualified_qual, RangeSums& sums,
                                  const DevAdapter* __restrict__ ad0, const DevAdapter* __restrict__ ad1,
                                  u64& key0, u64& key1) {
    const int lane = lane_id();
    const int blen = b - a;
    u32 lowq = 0, nn = 0, totq = 0, diff = 0;
    u32 bmm0 = 0xFFFFFFFFu, bp0 = 0xFFFFFFFFu, bmm1 = 0xFFFFFFFFu, bp1 = 0xFFFFFFFFu;
    const int npos0 = HAM ? blen - ad0->len : 0, npos1 = HAM ? blen - ad1->len : 0;
    u32 prev_last = 0; /* last byte of the previous tile */
    for (int t0 = 0; t0 < blen; t0 += 16 * 64) {
        const int j0 = t0 + 16 * lane;
        const int nvalid = blen > j0 ? min(16, blen - j0) : 0;
        u32x4 sv = {0, 0, 0, 0}, qv = {0, 0, 0, 0};
        if (nvalid > 0) {
            sv = load16_guard(rb + a + j0, seq_end);
            qv = load16_guard(qb + a + j0, qual_end);
        }
        u32 prevb = shfl_up_u32(sv.w >> 24, 1);
        if (lane == 0) prevb = prev_last;
        prev_last = readlane_u32(sv.w >> 24, 63);
        const u32 sw[4] = {sv.x, sv.y, sv.z, sv.w};
        const u32 qw[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (k < nvalid) {
                const u32 bb = (sw[k >> 2] >> (8 * (k & 3))) & 0xFF;
                const u32 q = (qw[k >> 2] >> (8 * (k & 3))) & 0xFF;
                atomicAdd(&h[(q & 127u) * HIST_COPIES + (lane & (HIST_COPIES - 1))], 1u); /* q < 128 in FASTQ */
                if (SUMS) {
                    lowq += ((int)q < qualified_qual);
                    nn += (bb == 'N');
                    totq += q;
                    diff += (bb != prevb) && (j0 + k > 0);
                    prevb = bb;
                }
            }
        }
        if (HAM) {
            u32 acc[4];
            if (npos0 > t0) { /* wave-uniform: some position of this tile is tested */
                hamming16(rb + a + j0, seq_end, sv, ad0->seqw, ad0->len, acc);
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const u32 mm = (acc[j >> 2] >> (8 * (j & 3))) & 0xFF;
                    if (j0 + j < npos0 && mm < bmm0) {
                        bmm0 = mm;
                        bp0 = (u32)(j0 + j);
                    }
                }
            }
            if (npos1 > t0) {
                hamming16(rb + a + j0, seq_end, sv, ad1->seqw, ad1->len, acc);
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const u32 mm = (acc[j >> 2] >> (8 * (j & 3))) & 0xFF;
                    if (j0 + j < npos1 && mm < bmm1) {
                        bmm1 = mm;
                        bp1 = (u32)(j0 + j);
                    }
                }
            }
        }
    }
    if (SUMS) {
        sums.lowq = wave_sum_u32(lowq);
        sums.nn = wave_sum_u32(nn);
        sums.totq = wave_sum_u32(totq);
        sums.diff = wave_sum_u32(diff);
    }
    if (HAM) {
        key0 = wave_min_u64(((u64)bmm0 << 32) | bp0);
        key1 = wave_min_u64(((u64)bmm1 << 32) | bp1);
    }
}


/* ---- bit-sliced scan ------------------------------------------------------------------
 * Letter bit-planes of 32 bases held as 8 dwords: bit j of P_X = (base j == X), X in A C T G,
 * compared as raw bytes (lower case, N, U, ... match nothing).  byte -> bit compaction with
 * v_dot4_u32_u8 against power-of-two weights: 4 bases per instruction and plane. */
/* three bit-planes of 32 bases that are all exactly A, C, G, T or N: bit j of L / H / N = ASCII bit 1 / 2 / 3 of base j
   (A 000, C 001 -- L set --, T 010, G 011; bit 3 is set for N = 0x4E alone, whose L and H bits read like G's) */
__device__ __forceinline__ void code_planes(const u32 s[8], u32& L, u32& H, u32& N) {
    /* the masked bit itself is the multiplicand (2, 4 or 8 per base instead of 1): no shift per dword and plane */
    L = 0;
    H = 0;
    N = 0;
#pragma unroll
    for (int pr = 0; pr < 4; pr++) {
        u32 lb = 0, hb = 0, nb = 0;
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const u32 w = s[2 * pr + hh];
            const u32 wt = hh ? 0x80402010u : 0x08040201u;
            lb = udot4(w & 0x02020202u, wt, lb);
            hb = udot4(w & 0x04040404u, wt, hb);
            nb = udot4(w & 0x08080808u, wt, nb);
        }
        /* lb = 2 * (8 plane bits), hb = 4 * ..., nb = 8 * ...: the factor goes away with the shift that parks the byte */
        L |= pr ? lb << (8 * pr - 1) : lb >> 1;
        H |= pr ? hb << (8 * pr - 2) : hb >> 2;
        N |= pr ? nb << (8 * pr - 3) : nb >> 3;
    }
}
/* non-zero when one of the 32 bytes is not exactly A, C, G, T or N: ASCII bits 1..3 pick the letter the byte would have
   to be out of a table (codes 4..6 pick a zero byte, which no base is) */
__device__ __forceinline__ u32 not_acgtn(const u32 s[8]) {
    u32 bad = 0;
#pragma unroll
    for (int d = 0; d < 8; d++) bad |= perm_b32(0x4E000000u, 0x47544341u, (s[d] >> 1) & 0x07070707u) ^ s[d];
    return bad;
}
/* sums32 for a full chunk of 32 bases that are all A, C, G, T or N (L, H, N = code_planes): the N count is one
   popcount, and "differs from its predecessor" (Filter::passLowComplexityFilter, src/filter.cpp:66-81) one XOR of each
   plane with itself shifted by one position -- 16 instead of 104 vector instructions.  The predecessor of base 0 (the
   byte in front of the chunk) may be any byte: that one comparison is made on the bytes. */
/* MASKED: only the first nvalid (1..32) bases of the chunk count */
/* QS: also the two sums over the qualities (the main scan leaves them to the histogram: hist_quality_sums) */
template <bool MASKED, bool QS = true>
__device__ __forceinline__ void sums32_acgtn(const u32 s0, const u32 q[8], int nvalid, u32 L, u32 H, u32 N, u32 prev_dword,
                                             u32 qqrep, u32& lowq, u32& nn, u32& totq, u32& diff) {
#pragma unroll
    for (int d = 0; QS && d < 8; d++) {
        u32 bm = ~0u, fm = 0x80808080u;
        if (MASKED) {
            const int c = nvalid - 4 * d;
            bm = c >= 4 ? ~0u : (c <= 0 ? 0u : ((1u << (8 * c)) - 1u));
            fm &= bm;
        }
        const u32 t = (q[d] | 0x80808080u) - qqrep;
        lowq = FPL_OPT_BCNT ? popc_acc(~t & fm, lowq) : lowq + popc32(~t & fm);
        totq = sum_bytes(q[d] & bm, totq);
    }
    const u32 vm = (!MASKED || nvalid >= 32) ? ~0u : ((1u << nvalid) - 1u);
    nn = FPL_OPT_BCNT ? popc_acc(N & vm, nn) : nn + popc32(N & vm);
    const u32 dm = ((L ^ (L << 1)) | (H ^ (H << 1)) | (N ^ (N << 1))) & ~1u & vm;
    const u32 d0 = ((s0 & 0xFFu) != (prev_dword >> 24)) ? 1u : 0u;
    diff = (FPL_OPT_BCNT ? popc_acc(dm, diff) : diff + popc32(dm)) + d0;
}
/* not_acgtn over the first nvalid (0..32) bytes only */
__device__ __forceinline__ u32 not_acgtn_masked(const u32 s[8], int nvalid) {
    u32 bad = 0;
#pragma unroll
    for (int d = 0; d < 8; d++) {
        const int c = nvalid - 4 * d;
        const u32 bm = c >= 4 ? ~0u : (c <= 0 ? 0u : ((1u << (8 * c)) - 1u));
        bad |= (perm_b32(0x4E000000u, 0x47544341u, (s[d] >> 1) & 0x07070707u) ^ s[d]) & bm;
    }
    return bad;
}
__device__ __forceinline__ void build_planes(const u32 s[8], u32& PA, u32& PC, u32& PT, u32& PG) {
    u32 L = 0, H = 0, X = 0;
#pragma unroll
    for (int pr = 0; pr < 4; pr++) {
        u32 lb = 0, hb = 0, xb = 0;
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const u32 w = s[2 * pr + hh];
            const u32 wt = hh ? 0x80402010u : 0x08040201u;
            const u32 w1 = w >> 1;
            lb = udot4(w1 & 0x01010101u, wt, lb);        /* ASCII bit 1: A0 C1 T0 G1 */
            hb = udot4((w >> 2) & 0x01010101u, wt, hb);  /* ASCII bit 2: A0 C0 T1 G1 */
            const u32 e = perm_lo(0x47544341u, w1 & 0x03030303u); /* the letter that code stands for */
            const u32 t = e ^ w;                                   /* zero byte <=> exactly that letter */
            const u32 nz = ((((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t) >> 7) & 0x01010101u;
            xb = udot4(nz, wt, xb);
        }
        L |= lb << (8 * pr);
        H |= hb << (8 * pr);
        X |= xb << (8 * pr);
    }
    const u32 V = ~X;
    PA = bitop3<0x10>(V, H, L); /* V & ~H & ~L */
    PC = bitop3<0x20>(V, H, L); /* V & ~H &  L */
    PT = bitop3<0x40>(V, H, L); /* V &  H & ~L */
    PG = bitop3<0x80>(V, H, L); /* V &  H &  L */
}

/* Number of adapter bases that match at each of this lane's 32 positions, as 7 bit-planes
 * (count <= 64).  plane_lane = &planes[0][lane] in LDS; term i fetches the plane of adapter letter
 * i shifted by i positions (two neighbouring words + v_alignbit); Harley-Seal carry-save adders
 * (v_bitop3 majority / parity) sum eight 1-bit planes with seven CSAs. */
template <int NB>
__device__ __forceinline__ void match_counts(const u32* __restrict__ plane_lane, const DevAdapter* __restrict__ ad, u32 (&B)[NB]) {
    const int alen = ad->len;
#ifndef FPL_EMU
    /* LDS byte address of lane 0's plane word, minus what the instruction adds for this lane */
    const u32 planes_m0 = uniform_u32((u32)(size_t)plane_lane - 4u * (u32)lane_id());
#endif
#pragma unroll
    for (int b = 0; b < NB; b++) B[b] = 0;
    /* the eights carry of eight terms (B[0..2] take their ones, twos and fours) */
    auto group8 = [&](int i0) -> u32 {
        u32 m[8], tw[8];
#pragma unroll
        for (int k = 0; k < 8; k++) tw[k] = ad->term[i0 + k]; /* terms past alen point at the all-zero plane row */
#ifdef FPL_EMU
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const u32* p = plane_lane + (tw[k] >> 10);
            m[k] = alignbit(p[1], p[0], tw[k] & 31u);
        }
#else
        /* ds_read_addtid_b32: LDS address = M0 + offset + 4 * lane, so a term costs two scalar ops and no
           address arithmetic on the vector unit (the compiler has no intrinsic for it) */
        u32 lo[8], hi[8], m0_keep;
        asm volatile(
            "s_mov_b32 %16, m0\n"
            "s_lshr_b32 m0, %17, 8\n s_add_u32 m0, m0, %25\n s_nop 0\n ds_read_addtid_b32 %0 offset:0\n ds_read_addtid_b32 %8 offset:4\n"
            "s_lshr_b32 m0, %18, 8\n s_add_u32 m0, m0, %25\n s_nop 0\n ds_read_addtid_b32 %1 offset:0\n ds_read_addtid_b32 %9 offset:4\n"
            "s_lshr_b32 m0, %19, 8\n s_add_u32 m0, m0, %25\n s_nop 0\n ds_read_addtid_b32 %2 offset:0\n ds_read_addtid_b32 %10 offset:4\n"
            "s_lshr_b32 m0, %20, 8\n s_add_u32 m0, m0, %25\n s_nop 0\n ds_read_addtid_b32 %3 offset:0\n ds_read_addtid_b32 %11 offset:4\n"
            "s_lshr_b32 m0, %21, 8\n s_add_u32 m0, m0, %25\n s_nop 0\n ds_read_addtid_b32 %4 offset:0\n ds_read_addtid_b32 %12 offset:4\n"
            "s_lshr_b32 m0, %22, 8\n s_add_u32 m0, m0, %25\n s_nop 0\n ds_read_addtid_b32 %5 offset:0\n ds_read_addtid_b32 %13 offset:4\n"
            "s_lshr_b32 m0, %23, 8\n s_add_u32 m0, m0, %25\n s_nop 0\n ds_read_addtid_b32 %6 offset:0\n ds_read_addtid_b32 %14 offset:4\n"
            "s_lshr_b32 m0, %24, 8\n s_add_u32 m0, m0, %25\n s_nop 0\n ds_read_addtid_b32 %7 offset:0\n ds_read_addtid_b32 %15 offset:4\n"
            "s_waitcnt lgkmcnt(0)\n s_mov_b32 m0, %16"
            : "=&v"(lo[0]), "=&v"(lo[1]), "=&v"(lo[2]), "=&v"(lo[3]), "=&v"(lo[4]), "=&v"(lo[5]), "=&v"(lo[6]), "=&v"(lo[7]),
              "=&v"(hi[0]), "=&v"(hi[1]), "=&v"(hi[2]), "=&v"(hi[3]), "=&v"(hi[4]), "=&v"(hi[5]), "=&v"(hi[6]), "=&v"(hi[7]),
              "=&s"(m0_keep)
            : "s"(tw[0]), "s"(tw[1]), "s"(tw[2]), "s"(tw[3]), "s"(tw[4]), "s"(tw[5]), "s"(tw[6]), "s"(tw[7]), "s"(planes_m0)
            : "scc", "memory"); /* M0 is restored: the compiler does not know it was touched */
        (void)m0_keep;
#pragma unroll
        for (int k = 0; k < 8; k++) m[k] = alignbit(hi[k], lo[k], tw[k]);
#endif
        u32 t1, t2, t3, t4, f1, f2, e;
        csa(t1, B[0], B[0], m[0], m[1]);
        csa(t2, B[0], B[0], m[2], m[3]);
        csa(f1, B[1], B[1], t1, t2);
        csa(t3, B[0], B[0], m[4], m[5]);
        csa(t4, B[0], B[0], m[6], m[7]);
        csa(f2, B[1], B[1], t3, t4);
        csa(e, B[2], B[2], f1, f2);
        return e;
    };
    /* ripple a carry of weight 2^lvl upwards */
    auto ripple = [&](u32 c, int lvl) {
#pragma unroll
        for (int b = 3; b < NB; b++) {
            if (b < lvl) continue;
            const u32 t = B[b] & c;
            B[b] ^= c;
            c = t;
        }
    };
    int i0 = 0;
#if FPL_OPT_CSA16
    for (; i0 + 8 < alen; i0 += 16) { /* sixteen terms: the two eights carries go through one more adder before they ripple */
        const u32 ea = group8(i0);
        const u32 eb = group8(i0 + 8);
        u32 c16;
        csa(c16, B[3], B[3], ea, eb);
        ripple(c16, 4);
    }
#endif
    for (; i0 < alen; i0 += 8) ripple(group8(i0), 3);
}

/* largest count among the positions in `cand` (non-zero) and the first position holding it */
template <int NB>
__device__ __forceinline__ void sliced_max(const u32 (&B)[NB], u32 cand, int& val, int& first) {
    val = 0;
#pragma unroll
    for (int b = NB - 1; b >= 0; b--) { /* branch-free: selects, no exec-mask juggling */
        const u32 t = cand & B[b];
        const bool nz = t != 0;
        cand = nz ? t : cand;
#if FPL_OPT_VALADDC && !defined(FPL_EMU)
        { /* val = 2 val + nz in one op: the compare's lane mask is the carry of an add-with-carry */
            u64 c = wave_ballot(nz);
            /* (s_nop: the mask comes from a vector compare, and nothing tells the scheduler that this instruction reads it) */
            asm("s_nop 1\n\tv_addc_co_u32_e64 %0, %1, %0, %0, %1" : "+v"(val), "+s"(c));
        }
#elif FPL_OPT_VALADDC
        val = val + val + (nz ? 1 : 0);
#else
        val |= nz ? (1 << b) : 0;
#endif
    }
    first = __ffs(cand) - 1;
}

/* passFilter sums over 32 bytes with SWAR byte tricks (quality and base bytes < 128):
 *   lowq: bytes with q < qq          totq: sum of q (v_sad_u8)
 *   nn  : bytes == 'N'               diff: bytes that differ from their predecessor (pw = byte before s[0])
 * MASKED: only the first nvalid bytes count (the ragged last tile of a range). */
template <bool MASKED, bool QS = true>
__device__ __forceinline__ void sums32(const u32 s[8], const u32 q[8], int nvalid, u32 prev_dword, u32 qqrep, u32& lowq,
                                       u32& nn, u32& totq, u32& diff) {
    u32 pd = prev_dword;
#pragma unroll
    for (int d = 0; d < 8; d++) {
        u32 bm = ~0u, fm = 0x80808080u; /* bytes / flag bits of this dword that count */
        if (MASKED) {
            const int c = nvalid - 4 * d;
            bm = c >= 4 ? ~0u : (c <= 0 ? 0u : ((1u << (8 * c)) - 1u));
            fm &= bm;
        }
        if (QS) {
            const u32 t = (q[d] | 0x80808080u) - qqrep; /* per byte, no borrow: bit 7 survives iff q >= qq */
            lowq = FPL_OPT_BCNT ? popc_acc(~t & fm, lowq) : lowq + popc32(~t & fm);
            totq = sum_bytes(q[d] & bm, totq);
        }
        const u32 x = s[d] ^ 0x4E4E4E4Eu;
        const u32 zx = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;
        nn = FPL_OPT_BCNT ? popc_acc(~zx & fm, nn) : nn + popc32(~zx & fm);
        const u32 y = s[d] ^ alignbyte(s[d], pd, 3);
        const u32 zy = ((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y;
        diff = FPL_OPT_BCNT ? popc_acc(zy & fm, diff) : diff + popc32(zy & fm);
        pd = s[d];
    }
}
/* This lane's way into its wave's histogram slice: a typed pointer to its copy of bin 0 and the same as an LDS byte
   address (see ScanWaveLds) */
struct HistLane {
    u32* p;   /* &hist[lane & (HIST_COPIES - 1)] */
    u32 addr; /* the LDS byte address of p: bits 5..11 zero (HIST_COPIES == 8) */
};
__device__ __forceinline__ HistLane hist_lane(u32* __restrict__ h) {
    HistLane hl;
    hl.p = h + (lane_id() & (HIST_COPIES - 1));
    hl.addr = (u32)(size_t)hl.p;
    return hl;
}
/* count byte K of the dword qd (a quality < 128) */
template <int K>
__device__ __forceinline__ void hist_bump(const HistLane& hl, u32 qd) {
#if defined(FPL_EMU) || FPL_HIST_COPIES != 8 || !defined(__HIP_DEVICE_COMPILE__) || !FPL_OPT_HIST
    atomicAdd(&hl.p[((qd >> (8 * K)) & 0x7Fu) * HIST_COPIES], 1u);
#else
    typedef __attribute__((address_space(3))) u32 lds_u32;
    const u32 sh = K == 0 ? (qd << 5) : (qd >> (8 * K - 5));
#if FPL_OPT_HIST == 3
    /* two full-rate instructions (v_and_b32 + v_add_u32) instead of one v_and_or_b32: a gfx950 SIMD runs the plain two-operand
       integer ops of two waves side by side, every three-operand op (v_and_or, v_lshl_or, v_perm ...) takes the whole SIMD and
       first waits until both halves are free ("the issue model") */
    const u32 a = (sh & 0xfe0u) + hl.addr;
#elif FPL_OPT_HIST == 2
    const u32 a = (sh & 0xfe0u) | hl.addr; /* (plain C: the compiler picks v_and_or_b32 and keeps its freedom to schedule) */
#else
    const u32 a = and_or(sh, 0xfe0u, hl.addr);
#endif
    __hip_atomic_fetch_add((lds_u32*)a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}
/* the 32 quality bytes of a lane into the wave's histogram.  MASKED: the bytes past nvalid are counted in bin 0 --
   the caller takes them out of that bin's total again (hist_dumped): a quality byte of 0 does not occur in FASTQ,
   and if it does, it still counts exactly */
template <bool MASKED>
__device__ __forceinline__ void hist32(const HistLane& hl, const u32 q[8], int nvalid) {
#pragma unroll
    for (int d = 0; d < 8; d++) {
        u32 qd = q[d];
        if (MASKED) {
            const int c = nvalid - 4 * d;
            const u32 bm = c >= 4 ? ~0u : (c <= 0 ? 0u : ((1u << (8 * c)) - 1u));
            qd &= bm;
        }
        hist_bump<0>(hl, qd);
        hist_bump<1>(hl, qd);
        hist_bump<2>(hl, qd);
        hist_bump<3>(hl, qd);
    }
}
/* bytes a masked scan of a range of blen bytes parks in bin 0: only the lane that holds the end of the range has a
   partly filled chunk */
__device__ __forceinline__ u32 hist_dumped(int blen) { return blen > 0 ? (u32)((SC_CHUNK - (blen % SC_CHUNK)) % SC_CHUNK) : 0u; }

/* Pair packing (north_star: "length-bucketed ... for coalesced loads"; what bucketing is for is that no lane idles): the reads of
 * a wave's chunk lie one behind the other, and the last tile of a read is ragged -- on average half its lanes hold nothing.
 * Lanes hb..61 of that tile take the first 32 * (62 - hb) bytes of the NEXT read's r1 (lanes 62 / 63: the plane words its last
 * windows reach into), so that read starts its own tiles that much further in.  Everything in the tile loop is lane-local but
 * four things: the predecessor byte (the head's first byte has none), the per-lane running sums and best windows (the lanes of
 * the head still hold the first read's: the head gets accumulators of its own, handed on as PairCarry), the masks of testable
 * windows (all ones in a head), and the quality histogram -- one per wave, so the head's quality bytes are counted once the first
 * read's totals are out of it (k_scan fetches them again for that: they are in the cache). */
#ifdef FPL_EMU_PAIR_STATS
static unsigned long long g_pair_stats[2]; /* emulator only: last tiles that hosted a head, bytes of heads */
#endif
struct PairCarry { /* the state of a read whose head was scanned in its predecessor's last tile */
    int t0;        /* wave-uniform: bytes of r1 that are done (0: none -- a plain scan) */
    u32 nn, diff;  /* per lane: the head's partial N count / complexity sum */
    int bm0, bp0, bm1, bp1; /* per lane: best match count / position so far, per adapter */
    u32 prev_last; /* wave-uniform: the dword whose top byte is the head's last byte */
};
struct PairIO {
    PairCarry in;  /* what the predecessor's last tile did of THIS range */
    /* the next read of the chunk as its metadata loads deliver it (per-lane copies of the same values: made wave-uniform where
       they are used -- in the last tile, so that the loads have the whole scan to come back) */
    bool allow;    /* wave-uniform: this read may host a head at all (and there is a next read) */
    uint64_t nx_o0;
    u32 nx_s, nx_e, nx_dropped;
    const u8* seq;
    const u8* qual;
    PairCarry out; /* out.t0 > 0: that much of the NEXT read's r1 was scanned in this range's last tile */
    int hb;        /* ... in lanes hb .. 61; their quality bytes are still to be counted (the caller loads them again and runs hist32
                      once the wave's histogram is free: holding them in registers through the window search costs spills) */
};

/*
 * One pass over bytes [a, b) of a read, 32 bases per lane and tile:
 *   - quality histogram into h (this wave's slice, see ScanWaveLds),
 *   - the passFilter sums (src/filter.cpp:27-39, 67-81) when SUMS,
 *   - when HAM, the Hamming argmin of both middle-adapter scans (searchAdapter default mode,
 *     src/adaptertrimmer.cpp:133-151: positions p in [0, (b-a) - alen), first global minimum)
 *     by the bit-sliced method above; keys (mismatches << 32 | p), ~0 when nothing was tested.
 *     Requires ACGT-only adapters of <= 64 bases (DevConfig::ham_fast).
 */
/* LEAN: the rare callers (long trimmed ends, the fragments of a split read) take the byte-masked variants for every
 * tile, full or ragged: half the code of an inlined copy, and the kernel's instruction footprint is what they cost. */
/* NB: bit-planes the match counts need (6 when both adapters have <= 32 bases, else 7).  h = this wave's histogram
 * slice.  Returns the number of bytes the masked tiles parked in bin 0 of the histogram (hist_dumped): the caller takes
 * them out of that bin's total. */
template <bool SUMS, bool HAM, bool LEAN = false, int NB = 7, bool PREFETCH = false, bool PAIR = false>
__device__ __forceinline__ u32 range_scan_fast(const u8* __restrict__ rb, const u8* __restrict__ qb, int a, int b,
                                               const u8* __restrict__ seq_end, const u8* __restrict__ qual_end,
                                               ScanWaveLds* __restrict__ w, u32* __restrict__ h, int qualified_qual, RangeSums& sums,
                                               const DevAdapter* __restrict__ ad0, const DevAdapter* __restrict__ ad1,
                                               u64& key0, u64& key1, bool do_ham = true, PairIO* __restrict__ pio = nullptr) {
    static_assert(!PAIR || (SUMS && HAM && !LEAN), "pair packing: the main scan only");
    /* do_ham (wave-uniform): false turns a HAM instance into a plain scan -- no window is tested, the keys come out
       as ~0 -- so that one inlined copy of the loop can serve reads with and without an adapter search */
    const int lane = lane_id();
    const int blen = b - a;
    constexpr int ACTIVE = HAM ? SC_LANES_HAM : 64;
    constexpr int ADV = ACTIVE * SC_CHUNK;
    const HistLane hl = hist_lane(h);
    u32 lowq = 0, nn = 0, totq = 0, diff = 0;
    int bm0 = -1, bp0 = 0, bm1 = -1, bp1 = 0; /* best match count / its position, per lane */
    const u32 act_mask = lane < ACTIVE ? 0xFFFFFFFFu : 0u;
    (void)act_mask;
    const int npos0 = (HAM && do_ham) ? blen - ad0->len : 0, npos1 = (HAM && do_ham) ? blen - ad1->len : 0;
    const int dbg = qualified_qual >> 8; /* ablation switches ride in the high bits */
    qualified_qual &= 0xFF;
    const u32 qqrep = 0x01010101u * (u32)(qualified_qual & 0x7F);
    u32 prev_tile_last = 0;
    /* the last byte of the range (the same in every lane; the ragged last tile pads with it) */
    u32 last_v = 0; /* (left in its vector register until the last tile: nothing waits for this load up front) */
    if (FPL_OPT_PADSCALAR && !LEAN && blen > 0) last_v = (u32)rb[a + blen - 1];
    /* what the lanes of a head find (they still hold THIS range's running sums and best windows: kept apart) */
    u32 h_nn = 0, h_diff = 0;
    int h_bm0 = -1, h_bp0 = 0, h_bm1 = -1, h_bp1 = 0;
    int t_first = 0;
    if (PAIR) {
        if (pio->in.t0 > 0) { /* (wave-uniform) the head of this range is done: go on from there with what its lanes found */
            t_first = pio->in.t0;
            nn = pio->in.nn;
            diff = pio->in.diff;
            bm0 = pio->in.bm0;
            bp0 = pio->in.bp0;
            bm1 = pio->in.bm1;
            bp1 = pio->in.bp1;
            prev_tile_last = pio->in.prev_last;
        }
        pio->out.t0 = 0;
    }
    for (int t0 = t_first; t0 < blen; t0 += ADV) {
        /* A wave consumes its tile as soon as the loads are back, so it sits out one trip to HBM per tile.  One byte
           of every 128-byte line of the NEXT tile (lanes 0..31 the bases, 32..63 the qualities), requested before this
           tile's own loads, brings that tile into the XCD's L2 meanwhile. */
        u32 pf = 0;
        if (PREFETCH && !LEAN) {
            const int line = 128 * (lane & 31), nx = t0 + ADV + line;
            if (line < ADV + 128 && nx < blen) pf = (u32)(lane < 32 ? rb : qb)[a + nx];
        }
        /* pair packing: does the next read's head ride in this tile, and from which lane on (hb; 64: no) */
        int hb = 64;
        const u8* nrb = nullptr;
        const u8* nqb = nullptr;
        if (PAIR && t0 + ADV >= blen && pio->allow && do_ham) { /* wave-uniform: the last tile of a read that may host */
            const int la = (blen - t0 + SC_CHUNK - 1) / SC_CHUNK; /* lanes this range still needs */
            if (ACTIVE - la >= FPL_PAIR_MIN_LANES) {
                const int ns = (int)uniform_u32(pio->nx_s), ne = (int)uniform_u32(pio->nx_e);
                /* every lane of the head, halo lanes included, loads 32 bytes of the next read's r1, and every position of
                   its active lanes starts a tested window (adapters of <= 32 bases here) */
                if (uniform_u32(pio->nx_dropped) == 0 && ne - ns >= SC_CHUNK * (64 - la) + 64) {
                    hb = la;
                    const uint64_t no0 = uniform_u64(pio->nx_o0);
                    nrb = pio->seq + no0 + ns;
                    nqb = pio->qual + no0 + ns;
                }
            }
        }
        const bool packed = PAIR && hb < 64;            /* wave-uniform */
        const bool head = PAIR && lane >= hb;           /* this lane holds bytes of the NEXT read */
        const int j0 = head ? SC_CHUNK * (lane - hb) : t0 + SC_CHUNK * lane; /* (the head's positions count from ITS first byte) */
        const int navail = head ? SC_CHUNK : (blen > j0 ? min(SC_CHUNK, blen - j0) : 0); /* bytes of the range in this chunk */
        const int nstat = lane < ACTIVE ? navail : 0;                 /* bytes this lane accounts for */
        u32 s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        /* a lane loads 32 bytes wherever its chunk starts inside the range: only the chunks of the batch's very last bytes
           can reach past the buffers (wave-uniform test on the tile's last possible byte), and only those pay for guards */
        if (packed) {
            /* (every lane has bytes, and the head lies in front of at least 64 more bytes of its read: no load leaves the batch) */
            const u8* const ps = head ? nrb + j0 : rb + a + j0;
            const u8* const pq = head ? nqb + j0 : qb + a + j0;
            const u32x4 s0 = load16(ps), s1 = load16(ps + 16);
            s[0] = s0.x; s[1] = s0.y; s[2] = s0.z; s[3] = s0.w;
            s[4] = s1.x; s[5] = s1.y; s[6] = s1.z; s[7] = s1.w;
            if (nstat > 0) {
                const u32x4 q0 = load16(pq), q1 = load16(pq + 16);
                q[0] = q0.x; q[1] = q0.y; q[2] = q0.z; q[3] = q0.w;
                q[4] = q1.x; q[5] = q1.y; q[6] = q1.z; q[7] = q1.w;
            }
        } else if (LEAN || rb + a + min(blen, t0 + 64 * SC_CHUNK) + SC_CHUNK > seq_end) {
            if (navail > 0) {
                const u32x4 s0 = load16_guard(rb + a + j0, seq_end), s1 = load16_guard(rb + a + j0 + 16, seq_end);
                s[0] = s0.x; s[1] = s0.y; s[2] = s0.z; s[3] = s0.w;
                s[4] = s1.x; s[5] = s1.y; s[6] = s1.z; s[7] = s1.w;
            }
            if (nstat > 0) {
                const u32x4 q0 = load16_guard(qb + a + j0, qual_end), q1 = load16_guard(qb + a + j0 + 16, qual_end);
                q[0] = q0.x; q[1] = q0.y; q[2] = q0.z; q[3] = q0.w;
                q[4] = q1.x; q[5] = q1.y; q[6] = q1.z; q[7] = q1.w;
            }
        } else {
            if (navail > 0) {
                const u32x4 s0 = load16(rb + a + j0), s1 = load16(rb + a + j0 + 16);
                s[0] = s0.x; s[1] = s0.y; s[2] = s0.z; s[3] = s0.w;
                s[4] = s1.x; s[5] = s1.y; s[6] = s1.z; s[7] = s1.w;
            }
            if (nstat > 0) {
                const u32x4 q0 = load16(qb + a + j0), q1 = load16(qb + a + j0 + 16);
                q[0] = q0.x; q[1] = q0.y; q[2] = q0.z; q[3] = q0.w;
                q[4] = q1.x; q[5] = q1.y; q[6] = q1.z; q[7] = q1.w;
            }
        }
        /* The last tile of a range is ragged: some lane holds fewer than 32 bytes of it.  Those lanes pad themselves -- the
           bases with copies of their last base, the qualities with zeros -- and the tile then takes the same code as every
           full tile: a padding base equals its predecessor (nothing for the complexity sum), a padding quality adds nothing
           to the histogram but to bin 0 (hist_dumped: the caller takes it out; the two sums over the qualities come from
           the histogram), and what a padding N adds to the N count is taken out of this lane's partial sum right here.  No
           tested window reaches a padding base (positions p < length - alen).  A lane without any byte of the range pads
           with 'A'. */
        if (!LEAN && t0 + 64 * SC_CHUNK > blen) { /* wave-uniform: some lane holds fewer than 32 bytes of the range */
#if FPL_OPT_PADSCALAR
            /* only ONE lane is cut by the end of the range -- lane lb, which keeps its first nb bytes -- and both numbers are
               wave-uniform: the eight byte masks are scalar values, the lanes in front of lb stay as they are, lane lb takes
               one v_bfi / v_and per dword, the lanes behind it (no byte of the range: nothing was loaded) become 'A's */
            const int rem = blen - t0; /* 1 .. 64 * 32 - 1 */
            const int lb = rem >> 5, nb = rem & 31;
            const u32 last_byte = uniform_u32(last_v);
            const u32 rep = 0x01010101u * last_byte;
            if (lane >= lb && !head) { /* (a head starts right behind the lane the range ends in) */
                if (lane == lb && nb != 0) {
#pragma unroll
                    for (int d = 0; d < 8; d++) {
                        const int c = nb - 4 * d; /* (wave-uniform) */
                        const u32 bm = c >= 4 ? ~0u : (c <= 0 ? 0u : ((1u << (8 * c)) - 1u));
                        s[d] = (s[d] & bm) | (rep & ~bm);
                        q[d] &= bm;
                    }
                    if (SUMS && lane < ACTIVE && last_byte == (u32)'N') nn -= (u32)(SC_CHUNK - nb);
                } else {
#pragma unroll
                    for (int d = 0; d < 8; d++) s[d] = 0x41414141u;
                }
            }
#else
            if (navail < SC_CHUNK) {
                const int li = navail > 0 ? navail - 1 : 0;
                u32 lw = s[0];
#pragma unroll
                for (int d = 1; d < 8; d++) lw = (li >> 2) == d ? s[d] : lw;
                const u32 last = navail > 0 ? ((lw >> (8 * (li & 3))) & 0xFFu) : (u32)'A';
                const u32 rep = 0x01010101u * last;
#pragma unroll
                for (int d = 0; d < 8; d++) {
                    const int c = navail - 4 * d;
                    const u32 bm = c >= 4 ? ~0u : (c <= 0 ? 0u : ((1u << (8 * c)) - 1u));
                    s[d] = (s[d] & bm) | (rep & ~bm);
                    q[d] &= bm;
                }
                if (SUMS && nstat > 0 && last == (u32)'N') nn -= (u32)(SC_CHUNK - nstat);
            }
#endif
        }
        /* predecessor of this chunk's first byte: last dword of the previous lane / previous tile */
        u32 prevd = FPL_OPT_DPPPREV ? wave_prev_u32(s[7], prev_tile_last) : shfl_up_u32(s[7], 1);
        if (!FPL_OPT_DPPPREV && lane == 0) prevd = prev_tile_last;
        prev_tile_last = readlane_u32(s[7], ACTIVE - 1);
        if (j0 == 0) prevd = s[0] << 24; /* the first byte of the range has no predecessor */
        /* a tile whose bytes are all exactly A, C, G, T or N -- nearly every tile -- is scanned on three code bit-planes:
           no per-byte validity test, the N count and the complexity sum from the planes.  Any other tile (a lower-case
           letter, a U, ...) takes the byte-wise sums and builds its letter planes with a validity test, whole wave
           (wave-uniform choice).  LEAN instances take the byte-masked variants for every tile. */
        bool acgt = false;
        u32 cL = 0, cH = 0, cN = 0;
        if (FPL_OPT_ACGT && !LEAN && (SUMS || HAM)) {
            acgt = !wave_ballot(not_acgtn(s) != 0);
            if (acgt) code_planes(s, cL, cH, cN);
        }
        if (!LEAN) {
            if (nstat > 0) {
                if (!FPL_DBG(dbg, 1) && !head) hist32<false>(hl, q, SC_CHUNK); /* (the head's bytes are counted by the caller, later) */
                if (SUMS && !FPL_DBG(dbg, 2)) { /* (N count and complexity sum; lowq / totq: hist_quality_sums) */
                    if (packed) { /* (wave-uniform) the head's lanes add to sums of their own */
                        u32 t_nn = head ? 0u : nn, t_diff = head ? 0u : diff;
                        if (acgt) sums32_acgtn<false, false>(s[0], q, SC_CHUNK, cL, cH, cN, prevd, qqrep, lowq, t_nn, totq, t_diff);
                        else sums32<false, false>(s, q, SC_CHUNK, prevd, qqrep, lowq, t_nn, totq, t_diff);
                        h_nn = head ? t_nn : 0u;
                        h_diff = head ? t_diff : 0u;
                        nn = head ? nn : t_nn;
                        diff = head ? diff : t_diff;
                    } else if (acgt) sums32_acgtn<false, false>(s[0], q, SC_CHUNK, cL, cH, cN, prevd, qqrep, lowq, nn, totq, diff);
                    else sums32<false, false>(s, q, SC_CHUNK, prevd, qqrep, lowq, nn, totq, diff);
                }
                if (FPL_DBG(dbg, 4)) totq += s[0] + s[3] + s[4] + s[7] + q[0] + q[3] + q[4] + q[7]; /* keep the loads alive */
            }
        } else if (nstat > 0) {
            hist32<true>(hl, q, nstat);
            if (SUMS) sums32<true>(s, q, nstat, prevd, qqrep, lowq, nn, totq, diff);
        }
        if (HAM) {
            if ((npos0 > t0 || npos1 > t0 || packed) && !FPL_DBG(dbg, 16)) { /* wave-uniform: some window of this tile is tested */
                u32 PA, PC, PT, PG;
                if (acgt) {
                    PA = ~(cH | cL); /* (an N has both bits set) */
                    PC = ~cH & cL;
                    PT = cH & ~cL;
                    PG = cH & cL & ~cN;
                } else {
                    build_planes(s, PA, PC, PT, PG);
                }
                wave_sync(); /* previous tile's plane reads are done */
                w->planes[0][lane] = PA;
                w->planes[1][lane] = PC;
                w->planes[2][lane] = PT;
                w->planes[3][lane] = PG;
                wave_sync();
                /* lanes 62/63 hold halo words only; their (clamped) plane reads are never used */
                const u32* plane_lane = &w->planes[0][lane < ACTIVE ? lane : 0];
                u32 B[NB];
                if ((npos0 > t0 || packed) && !FPL_DBG(dbg, 8)) {
                    match_counts(plane_lane, ad0, B);
                    u32 vm = act_mask; /* every position of every active lane is a window start ... */
                    if (!FPL_OPT_VMFULL || npos0 - t0 < ACTIVE * SC_CHUNK) { /* ... except in the last tile(s) (wave-uniform) */
                        const int nv = npos0 - j0;
                        vm = (lane >= ACTIVE || nv <= 0) ? 0u : (nv >= 32 ? 0xFFFFFFFFu : ((1u << nv) - 1u));
                    }
                    if (head) vm = act_mask; /* (a head is followed by more of its read than any window is long) */
                    if (vm) {
                        int val, first;
                        sliced_max(B, vm, val, first);
                        if (packed && head) { /* (its first and only tile so far) */
                            h_bm0 = val;
                            h_bp0 = j0 + first;
                        } else if (val > bm0) {
                            bm0 = val;
                            bp0 = j0 + first;
                        }
                    }
                }
                if ((npos1 > t0 || packed) && !FPL_DBG(dbg, 8)) {
                    match_counts(plane_lane, ad1, B);
                    u32 vm = act_mask; /* every position of every active lane is a window start ... */
                    if (!FPL_OPT_VMFULL || npos1 - t0 < ACTIVE * SC_CHUNK) { /* ... except in the last tile(s) (wave-uniform) */
                        const int nv = npos1 - j0;
                        vm = (lane >= ACTIVE || nv <= 0) ? 0u : (nv >= 32 ? 0xFFFFFFFFu : ((1u << nv) - 1u));
                    }
                    if (head) vm = act_mask; /* (a head is followed by more of its read than any window is long) */
                    if (vm) {
                        int val, first;
                        sliced_max(B, vm, val, first);
                        if (packed && head) { /* (its first and only tile so far) */
                            h_bm1 = val;
                            h_bp1 = j0 + first;
                        } else if (val > bm1) {
                            bm1 = val;
                            bp1 = j0 + first;
                        }
                    }
                }
            }
        }
#if !defined(FPL_EMU)
        if (PREFETCH && !LEAN) asm volatile("" ::"v"(pf)); /* (keeps the touch load alive; it has long returned) */
#else
        (void)pf;
#endif
        if (packed) { /* (wave-uniform; the last tile) what the head's lanes found goes to the next read */
            pio->out.t0 = SC_CHUNK * (ACTIVE - hb);
            pio->out.nn = h_nn;
            pio->out.diff = h_diff;
            pio->out.bm0 = h_bm0;
            pio->out.bp0 = h_bp0;
            pio->out.bm1 = h_bm1;
            pio->out.bp1 = h_bp1;
            pio->out.prev_last = prev_tile_last; /* (lane 61's last dword: the head's last byte on top) */
            pio->hb = hb;
#ifdef FPL_EMU_PAIR_STATS
            if (lane == 0) {
                __atomic_fetch_add(&g_pair_stats[0], 1ull, __ATOMIC_RELAXED);
                __atomic_fetch_add(&g_pair_stats[1], (unsigned long long)pio->out.t0, __ATOMIC_RELAXED);
            }
#endif
        }
    }
    if (SUMS) {
        /* (the main scan leaves lowq / totq to its caller: hist_quality_sums on the histogram totals) */
        sums.lowq = LEAN ? wave_sum_u32(lowq) : 0u;
        sums.totq = LEAN ? wave_sum_u32(totq) : 0u;
        if (FPL_OPT_PACKRED && !LEAN && blen < 65536) { /* (wave-uniform) neither sum exceeds the range's length: one reduction for the two */
            const u32 x = wave_sum_u32(nn | (diff << 16));
            sums.nn = x & 0xFFFFu;
            sums.diff = x >> 16;
        } else {
            sums.nn = wave_sum_u32(nn);
            sums.diff = wave_sum_u32(diff);
        }
    }
    if (HAM) {
        /* the first position with the fewest mismatches = the largest match count, then the smallest position holding it:
           two 32-bit reductions per adapter (the (mismatches, position) pair as one 64-bit key costs three times that) */
        key0 = key1 = ~0ull;
        if (do_ham && FPL_OPT_PACKRED && blen < (1 << 24)) { /* wave-uniform */
            /* (match count + 1) above the position counted down from 2^24 - 1: ONE maximum per adapter gives the largest count and,
               among the lanes holding it, the smallest position (a lane that tested nothing holds 0 above 2^24 - 1: below any real key) */
            const u32 k0 = wave_max_u32(((u32)(bm0 + 1) << 24) | (0xFFFFFFu - (u32)bp0));
            const u32 k1 = wave_max_u32(((u32)(bm1 + 1) << 24) | (0xFFFFFFu - (u32)bp1));
            const u32 m0 = k0 >> 24, m1 = k1 >> 24;
            if (m0) key0 = ((u64)(u32)(ad0->len - (int)(m0 - 1)) << 32) | (0xFFFFFFu - (k0 & 0xFFFFFFu));
            if (m1) key1 = ((u64)(u32)(ad1->len - (int)(m1 - 1)) << 32) | (0xFFFFFFu - (k1 & 0xFFFFFFu));
        } else if (do_ham) { /* wave-uniform */
            const u32 m0 = wave_max_u32((u32)(bm0 + 1)), m1 = wave_max_u32((u32)(bm1 + 1)); /* 0: no window was tested */
            const u32 p0 = wave_min_u32((u32)(bm0 + 1) == m0 ? (u32)bp0 : ~0u);
            const u32 p1 = wave_min_u32((u32)(bm1 + 1) == m1 ? (u32)bp1 : ~0u);
            if (m0) key0 = ((u64)(u32)(ad0->len - (int)(m0 - 1)) << 32) | p0;
            if (m1) key1 = ((u64)(u32)(ad1->len - (int)(m1 - 1)) << 32) | p1;
        }
    }
    return hist_dumped(blen);
}

/* Filter::passFilter + passLowComplexityFilter from the sums, src/filter.cpp:12-81.  The
 * reference's double comparisons are equivalent to these integer cross-multiplications
 * ("float <-> integer equivalences"). */
__device__ __forceinline__ int filter_code(const DevConfig* __restrict__ cfg, int len, const RangeSums& sm) {
    if (len == 0) return FPL_FAIL_LENGTH;
    if (cfg->qual_filter) {
        const int totalQual = (int)sm.totq - 33 * len;
        if ((long long)sm.lowq * 100 > (long long)cfg->unqual_pct * len) return FPL_FAIL_QUALITY;
        else if (cfg->avg_qual_req > 0 && (totalQual / len) < cfg->avg_qual_req) return FPL_FAIL_QUALITY;
        else if ((long long)sm.nn * 100 > (long long)len * cfg->n_pct_limit) return FPL_FAIL_N_BASE;
        else if (cfg->n_base_limit != 1000000 && (int)sm.nn > cfg->n_base_limit) return FPL_FAIL_N_BASE;
    }
    if (cfg->length_filter) {
        if (len < cfg->required_length) return FPL_FAIL_LENGTH;
        if (cfg->max_length > 0 && len > cfg->max_length) return FPL_FAIL_TOO_LONG;
    }
    if (cfg->complexity) {
        if (len <= 1) return FPL_FAIL_COMPLEXITY;
        if (!((long long)sm.diff * 100 >= (long long)cfg->complexity_pct * (len - 1))) return FPL_FAIL_COMPLEXITY;
    }
    return FPL_PASS_FILTER;
}


/* add a block's accumulators to the counter buffer (block-wide; call after a __syncthreads) */
__device__ __forceinline__ void scan_acc_flush(ScanBlockAcc& acc, long long* __restrict__ counters, u32 C) {
    for (int k = 0; k < 2; k++) {
        long long* st = counters + (k == 0 ? FPL_OFF_PRE(C) : FPL_OFF_POST(C));
        for (u32 i = threadIdx.x; i < 128; i += blockDim.x) {
            if (acc.bqh[k][i]) atomicAdd((u64*)&st[FPL_ST_BASE_QUAL_HIST(C) + i], acc.bqh[k][i]);
            if (acc.medh[k][i]) atomicAdd((u64*)&st[FPL_ST_MEDIAN_HIST(C) + i], acc.medh[k][i]);
            if (acc.medb[k][i]) atomicAdd((u64*)&st[FPL_ST_MEDIAN_BASES(C) + i], acc.medb[k][i]);
        }
        if (threadIdx.x == 0) {
            if (acc.reads[k]) atomicAdd((u64*)&st[FPL_ST_READS(C)], acc.reads[k]);
            if (acc.lensum[k]) atomicAdd((u64*)&st[FPL_ST_LENGTH_SUM(C)], acc.lensum[k]);
        }
    }
    long long* fr = counters + FPL_OFF_FR(C);
    for (u32 i = threadIdx.x; i < FPL_FILTER_RESULT_TYPES; i += blockDim.x)
        if (acc.fr[i]) atomicAdd((u64*)&fr[FPL_FR_FILTER + i], acc.fr[i]);
}

/* The trimmed-off ends of a read (head [0, s), tail [e, l)) only feed the pre-filter quality histogram and
 * are short: one byte per lane, two rounds each (SC_END_PF bytes), loaded early so that the trip to HBM
 * overlaps the body scan; anything longer (rare) goes through a range scan. */
constexpr int SC_END_PF = 128;
struct EndBytes {
    u32 q[4]; /* head round 0 / 1, tail round 0 / 1 */
};
__device__ __forceinline__ EndBytes ends_prefetch(const u8* __restrict__ qb, int s, int e, int l) {
    const int lane = lane_id();
    EndBytes eb;
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int jh = 64 * r + lane, jt = e + 64 * r + lane;
        eb.q[r] = jh < s ? (u32)qb[jh] : 0u;
        eb.q[2 + r] = jt < l ? (u32)qb[jt] : 0u;
    }
    return eb;
}
__device__ __forceinline__ void ends_apply(u32* __restrict__ eh, const EndBytes& eb, int s, int e, int l) {
    const int lane = lane_id();
#pragma unroll
    for (int r = 0; r < 2; r++) {
        if (64 * r + lane < s) atomicAdd(&eh[eb.q[r] & 127u], 1u);
        if (e + 64 * r + lane < l) atomicAdd(&eh[eb.q[2 + r] & 127u], 1u);
    }
}

/* Levenshtein confirmation of both middle-adapter candidates at once (ACGT-only adapters of <= 32 bases).
 * The two text windows are fetched early (lev_pair32_fetch, right after the scan that found them, so the
 * trip to memory overlaps the histogram work); their Peq words come from a 4-entry LDS table per adapter;
 * lane 0 runs the column recurrence of adapter 0 and lane 1 that of adapter 1 in the same VALU
 * instructions (lane j holds the Peq word of column j of both windows; v_readlane + one select feed the
 * two lanes).  edX = the distance when it is <= thrX, some value > thrX otherwise (see lev_round32);
 * windows that need no confirmation (needX false) are skipped. */
struct LevPairText {
    u32 b0, b1;
};
__device__ __forceinline__ LevPairText lev_pair32_fetch(const u8* __restrict__ t0, int m0, bool need0,
                                                        const u8* __restrict__ t1, int m1, bool need1) {
    const int lane = lane_id();
    LevPairText x;
    x.b0 = (need0 && lane < m0) ? (u32)t0[lane] : 0u;
    x.b1 = (need1 && lane < m1) ? (u32)t1[lane] : 0u;
    return x;
}
__device__ __forceinline__ u32 peq4_lookup(const u32* __restrict__ tbl, u32 b) {
    const u32 code = (b >> 1) & 3u; /* A0 C1 T2 G3 */
    return ((0x47544341u >> (8 * code)) & 0xFFu) == b ? tbl[code] : 0u;
}
__device__ __forceinline__ void lev_pair32_run(const u32 (*__restrict__ peq4)[4], const LevPairText& x, int m0, int thr0,
                                               bool need0, int m1, int thr1, bool need1, int& ed0, int& ed1) {
    const int lane = lane_id();
    const WaveVals64 pub0 = wave_publish((u64)peq4_lookup(peq4[0], x.b0));
    const WaveVals64 pub1 = wave_publish((u64)peq4_lookup(peq4[1], x.b1));
    const bool second = lane == 1;
    const int m = second ? m1 : m0, thr = second ? thr1 : thr0;
    const bool active = (lane == 0 && need0 && m0 > 0) || (second && need1 && m1 > 0);
    u32 Pv = ~0u, Mv = 0;
    int score = m, fin = m; /* fin: the score after the lane's last column (m == 0: the distance is the text length) */
    bool bad = false;       /* the early-exit bound fired at some column */
    const u32 topsh = m > 0 ? (u32)(m - 1) : 0u;
    const int mmax = max(need0 ? m0 : 0, need1 ? m1 : 0);
    for (int t0 = 0; t0 < mmax; t0 += 2) {
#pragma unroll
        for (int u = 0; u < 2; u++) { /* straight-line: the exit test below runs once per two columns */
            const int t = t0 + u;
            const u32 e0 = (u32)pub0.get(t & 63), e1 = (u32)pub1.get(t & 63);
            const u32 Eq = second ? e1 : e0;
            const u32 Xv = Eq | Mv;
            const u32 Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
            u32 Ph = Mv | ~(Xh | Pv);
            u32 Mh = Pv & Xh;
            score += (int)((Ph >> topsh) & 1u) - (int)((Mh >> topsh) & 1u);
            Ph = (Ph << 1) | 1u;
            Mh <<= 1;
            Pv = Mh | ~(Xv | Ph);
            Mv = Ph & Xv;
            /* columns past the lane's own text (t >= m) change nothing that is kept */
            bad = bad || (t < m && score - (m - 1 - t) > thr);
            fin = t == m - 1 ? score : fin;
        }
        if (!(wave_ballot(active && !bad && t0 + 2 < m) & 3ull)) break; /* wave-uniform */
    }
    const int res = !active ? ((lane == 0 && need0) ? m0 : ((second && need1) ? m1 : 0)) : (bad ? thr + 1 : fin);
    ed0 = readlane_i32(res, 0);
    ed1 = readlane_i32(res, 1);
}

/* One Levenshtein confirmation per lane: every lane its own window (the m text bytes in w, m = the adapter's length <= 32,
 * A / C / G / T only; peq4row = the adapter's four Peq words, in LDS).  True when the global edit distance is <= thr. */
__device__ __forceinline__ bool lev_lanes32_acgt_w(const u32 (&w)[8], int m, int thr, bool need, const u32* __restrict__ peq4row) {
    const int mm = need ? m : 0;
    const u32 topsh = mm > 0 ? (u32)(mm - 1) : 0u;
    const int mmax = (int)wave_max_u32((u32)mm);
    u32 Pv = ~0u, Mv = 0;
    int score = mm;
    bool alive = need; /* the window can still end within thr: every remaining column lowers the score by one at most */
#pragma unroll
    for (int t = 0; t < 32; t++) {
        if (t < mmax) { /* wave-uniform */
            const u32 c = (w[t >> 2] >> (8 * (t & 3))) & 0xFFu;
            const u32 Eq = peq4_lookup(peq4row, c);
            const u32 Xv = Eq | Mv;
            const u32 Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
            u32 Ph = Mv | ~(Xh | Pv);
            u32 Mh = Pv & Xh;
            const int sc = score + (int)((Ph >> topsh) & 1u) - (int)((Mh >> topsh) & 1u);
            Ph = (Ph << 1) | 1u;
            Mh <<= 1;
            const bool act = t < mm;
            score = act ? sc : score;
            const u32 nPv = Mh | ~(Xv | Ph), nMv = Ph & Xv;
            Pv = act ? nPv : Pv;
            Mv = act ? nMv : Mv;
            if ((t & 3) == 3) { /* most windows are hopeless after a dozen columns: leave when every lane's is */
                alive = alive && score - (mm - 1 - t) <= thr;
                if (!wave_ballot(alive)) break;
            }
        }
    }
    return alive && score <= thr;
}
/* ... the window fetched from the read (32 bytes at `text`) */
__device__ __forceinline__ bool lev_lanes32_acgt(const u8* __restrict__ text, int m, int thr, bool need,
                                                 const u32* __restrict__ peq4row, const u8* __restrict__ seq_end) {
    u32 w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (need) {
        const u32x4 a = load16_guard(text, seq_end), b = load16_guard(text + 16, seq_end);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
        w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
    }
    return lev_lanes32_acgt_w(w, m, thr, need, peq4row);
}
/* ... and for adapters of up to 64 bases (A / C / G / T only): 64-bit columns, the window fetched 16 bytes at a time */
__device__ __forceinline__ u64 peq4_lookup64(const u64* __restrict__ tbl, u32 b) {
    const u32 code = (b >> 1) & 3u; /* A0 C1 T2 G3 */
    return ((0x47544341u >> (8 * code)) & 0xFFu) == b ? tbl[code] : 0ull;
}
__device__ __forceinline__ bool lev_lanes64_acgt(const u8* __restrict__ text, int m, int thr, bool need,
                                                 const u64* __restrict__ peq4row, const u8* __restrict__ seq_end) {
    const int mm = need ? m : 0;
    const u32 topsh = mm > 0 ? (u32)(mm - 1) : 0u;
    const int mmax = (int)wave_max_u32((u32)mm);
    u64 Pv = ~0ull, Mv = 0;
    int score = mm;
    bool alive = need;
    for (int t0 = 0; t0 < mmax; t0 += 16) { /* wave-uniform */
        u32 w[4] = {0, 0, 0, 0};
        if (need && t0 < mm) {
            const u32x4 a = load16_guard(text + t0, seq_end);
            w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
        }
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int t = t0 + u;
            const u32 c = (w[u >> 2] >> (8 * (u & 3))) & 0xFFu;
            const u64 Eq = peq4_lookup64(peq4row, c);
            const u64 Xv = Eq | Mv;
            const u64 Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
            u64 Ph = Mv | ~(Xh | Pv);
            u64 Mh = Pv & Xh;
            const int sc = score + (int)((Ph >> topsh) & 1ull) - (int)((Mh >> topsh) & 1ull);
            Ph = (Ph << 1) | 1ull;
            Mh <<= 1;
            const bool act = t < mm;
            score = act ? sc : score;
            const u64 nPv = Mh | ~(Xv | Ph), nMv = Ph & Xv;
            Pv = act ? nPv : Pv;
            Mv = act ? nMv : Mv;
            if ((u & 3) == 3) alive = alive && (t >= mm || score - (mm - 1 - t) <= thr); /* (columns behind the window count for nothing) */
        }
        if (!wave_ballot(alive)) break;
    }
    return alive && score <= thr;
}

/* What k_scan leaves per read for k_resolve: the read's wave has scanned r1 once and reduced what every lane saw; everything
 * that follows from these numbers -- the Levenshtein confirmations, breakByGap, the result record, the FilterResult and
 * median counters, the plan for the statistics passes -- is per-read arithmetic that k_resolve runs with LANE = READ. */
struct alignas(16) ScanRec {
    u32 lowq, nn, totq, diff; /* the passFilter sums of r1 (src/filter.cpp:27-39, 67-81) */
    u32 pos0, pos1;           /* Hamming argmin of the two middle-adapter scans, in r1 coordinates */
    u32 mm;                   /* mismatches there: adapter 0 | adapter 1 << 8; median quality of the read << 16 | of r1 << 24 */
    u32 flags;                /* SR_TESTED0 / SR_TESTED1: the scan tested at least one window; bits 8..15: passFilter code of r1
                                 as ONE output read (valid unless the read was dropped or --break / --mask defer the filter) */
};
constexpr u32 SR_TESTED0 = 1u, SR_TESTED1 = 2u;
/* ... and, when both command-line adapters have <= 32 bases, the 32 bytes at each argmin: k_scan still has them in its
   XCD's L2, k_resolve would have to fetch two scattered cache lines per read */
struct alignas(16) ScanWin {
    u32 w[2][8];
};
/* a read whose r1 a middle adapter splits: k_resolve found the gap, k_redo scans the fragments */
struct alignas(16) RedoItem {
    u32 ri, gs, glen, pad;
};
/* the REDO list (n_reads slots) is filled from both ends: reads whose r1 is longer than this from the front (their count in
   the word two behind the short items' count), the others from the back */
constexpr int REDO_LONG = 16384;

/* the base-quality histograms of one wave of k_scan (bins 2 * lane and 2 * lane + 1, pre- / post-filter), kept in registers and
   handed to the block's LDS accumulators when they might overflow and when the wave ends */
struct BqRegs {
    u32 pre0, pre1, post0, post1;
    u32 bases; /* bases counted since the last hand-over (wave-uniform) */
};
struct ScanBqAcc {
    u64 bqh[2][128];
};
__device__ __forceinline__ void bq_hand_over(BqRegs& r, ScanBqAcc& acc) {
    const int lane = lane_id();
    if (r.pre0) atomicAdd(&acc.bqh[0][2 * lane], (u64)r.pre0);
    if (r.pre1) atomicAdd(&acc.bqh[0][2 * lane + 1], (u64)r.pre1);
    if (r.post0) atomicAdd(&acc.bqh[1][2 * lane], (u64)r.post0);
    if (r.post1) atomicAdd(&acc.bqh[1][2 * lane + 1], (u64)r.post1);
    r.pre0 = r.pre1 = r.post0 = r.post1 = 0;
    r.bases = 0;
}

/* k_scan: ONE pass over r1 of every read -- the quality histogram (medians, base-quality histogram), the passFilter sums,
 * both middle-adapter Hamming scans -- and nothing else: a wave reduces what its lanes saw to a 32-byte ScanRec and goes on
 * to the next read.
 * SHORT: adapter trimming is on and both command-line adapters are ACGT-only and <= 32 bases (DevConfig::scan_short):
 * the byte-wise scan is left out of that instantiation */
template <int WAVES, bool SHORT>
__global__ void __launch_bounds__(WAVES * 64, SCAN_BLOCKS_PER_CU * WAVES / 4)
k_scan(const u8* __restrict__ seq, const u8* __restrict__ qual, const uint64_t* __restrict__ off, u32 n_reads,
       uint64_t n_bytes, const DevConfig* __restrict__ cfg, const DevAdapter* __restrict__ ads,
       const ReadState* __restrict__ state, ScanRec* __restrict__ recs, ScanWin* __restrict__ wins,
       long long* __restrict__ counters, u32 C, u32* __restrict__ work_ctr, u32 chunk) {
    __shared__ alignas(4096) u32 hist_all[WAVES][128 * HIST_COPIES]; /* 4 KiB per wave when HIST_COPIES == 8: hist_bump */
    __shared__ ScanWaveLds wlds[WAVES];
    __shared__ ScanBqAcc acc;
    const bool pair32 = SHORT || (cfg->ham_fast && ads[0].len <= 32 && ads[1].len <= 32); /* (k_resolve: the same test) */
    const int lane = lane_id();
    ScanWaveLds* const wl = &wlds[wave_in_block()];
    u32* const h = hist_all[wave_in_block()];
    constexpr int NB = (SHORT && FPL_OPT_NB6) ? 6 : 7; /* count planes: both adapters <= 32 bases / <= 64 */
    wl->planes[4][lane] = 0;
    hist_zero(h); /* from here on every user of the histograms leaves them zeroed */
    wl->ehist[lane] = 0;
    wl->ehist[64 + lane] = 0;
    {
        u64* z = (u64*)&acc;
        for (u32 i = threadIdx.x; i < sizeof(ScanBqAcc) / 8; i += blockDim.x) z[i] = 0;
    }
    __syncthreads();
    const u8* seq_end = seq + n_bytes;
    const u8* qual_end = qual + n_bytes;
    const int qq = cfg->qualified_qual | (cfg->dbg << 8);
    const bool defer = cfg->defer != 0;
    BqRegs bq = {0, 0, 0, 0, 0};

    /* dynamic work distribution in chunks: one device-scope atomic serves `chunk` reads (a single
       hot counter sustains only ~80 atomics/us, which a per-read dequeue would saturate) */
    u32 chunk_next = 0, chunk_end = 0;
    bool have_next = false;
    constexpr bool PAIR = SHORT && FPL_OPT_PAIR != 0;
    PairIO pio;
    pio.in.t0 = 0;
    pio.out.t0 = 0;
    pio.allow = false;
    pio.seq = seq;
    pio.qual = qual;
    PROF_INIT();
    uint64_t nx_o0 = 0, nx_o1 = 0;
    ReadState nx_st = {0, 0, 0, 0};
    for (;;) {
        if (chunk_next >= chunk_end) {
            u32 base = 0;
            if (lane == 0) base = atomicAdd(work_ctr, chunk);
            base = readlane_u32(base, 0);
            if (base >= n_reads) break;
            chunk_next = base;
            chunk_end = min(n_reads, base + chunk);
            have_next = false;
        }
        const u32 ri = chunk_next++;
        /* this read's metadata: loaded while the previous read was being processed, when possible */
        uint64_t o0, o1;
        ReadState st;
        if (have_next) {
            o0 = nx_o0;
            o1 = nx_o1;
            st = nx_st;
        } else {
            o0 = off[ri];
            o1 = off[ri + 1];
            st = state[ri];
        }
        /* every lane holds the same values: keep them (and all the range arithmetic and branching derived
           from them) on the scalar unit */
        o0 = uniform_u64(o0);
        o1 = uniform_u64(o1);
        st.s = uniform_u32(st.s);
        st.e = uniform_u32(st.e);
        st.dropped = uniform_u32(st.dropped);
        have_next = chunk_next < chunk_end;
        if (have_next) { /* the next read of the chunk */
            /* (scalar loads, as the compiler makes them: forced through the vector path -- so that the tile loop's lgkmcnt waits do not
               wait for them -- k_scan got 1.3 % slower at 8 kb and 6 % at 2 kb, round 5) */
            nx_o0 = o1;
            nx_o1 = off[ri + 2];
            nx_st = state[ri + 1];
        }
        if (PAIR) {
            pio.nx_o0 = nx_o0;
            pio.nx_s = nx_st.s;
            pio.nx_e = nx_st.e;
            pio.nx_dropped = nx_st.dropped;
        }
        const int l = (int)(o1 - o0);
        const u8* rb = seq + o0;
        const u8* qb = qual + o0;
        const int s = (int)st.s, e = (int)st.e;
        const bool dropped = st.dropped != 0;
        const int blen = e - s;
        /* the trimmed-off ends only feed the pre-filter histogram: start their loads now, use them
           after the body scan */
        const EndBytes eb = ends_prefetch(qb, s, e, l);
        PROF(0) /* dequeue, metadata, prefetch issue */

        /* ---- r1 body: histogram + filter sums + (adapters enabled) both Hamming scans */
        RangeSums sm = {0, 0, 0, 0};
        u64 key0 = ~0ull, key1 = ~0ull;
        const bool ham = !dropped && cfg->adapter_enabled;
        u32 dumped = 0; /* bytes the padded last tile of the body scan parked in bin 0 (hist_dumped) */
        bool qsums = true; /* lowq / totq still to come (from the histogram) */
        if (PAIR) {
            /* the next read's head may ride in this read's last tile when the histogram is not needed again before this
               read's totals are taken: no trimmed end so long that it takes a scan of its own (below) */
            pio.allow = have_next && ham && !defer && s <= SC_END_PF && l - e <= SC_END_PF;
            dumped = range_scan_fast<true, true, false, NB, false, true>(rb, qb, s, e, seq_end, qual_end, wl, h, qq, sm, &ads[0],
                                                                                          &ads[1], key0, key1, ham, &pio);
        } else if (SHORT || (ham && cfg->ham_fast)) /* (SHORT: also the reads without an adapter search, through do_ham) */
            dumped = range_scan_fast<true, true, false, NB>(rb, qb, s, e, seq_end, qual_end, wl, h, qq, sm, &ads[0], &ads[1], key0, key1, ham);
        else if (ham) { /* adapters with bytes outside ACGT or longer than 64: byte-wise SWAR scan */
            range_scan_bytes<true, true>(rb, qb, s, e, seq_end, qual_end, h, qq, sm, &ads[0], &ads[1], key0, key1);
            qsums = false;
        } else
            dumped = range_scan_fast<true, false>(rb, qb, s, e, seq_end, qual_end, wl, h, qq, sm, nullptr, nullptr, key0, key1);
        /* the 32 bytes at the two argmins, for k_resolve's edit distances: dword lane & 7 of window lane >> 3 */
        u32 wv = 0;
        const bool wsave = pair32 && ham && (key0 != ~0ull || key1 != ~0ull); /* wave-uniform */
        if (wsave && lane < 16) {
            const u64 k = lane < 8 ? key0 : key1;
            if (k != ~0ull) wv = load4_guard(rb + s + (int)(u32)k + 4 * (lane & 7), seq_end);
        }
        PROF(2) /* body scan */
        /* (pair packing) the qualities of the next read's head, lanes hb .. 61: fetched again -- they are in the cache -- rather
           than carried through the window search in registers; counted once this read's totals are out of the histogram */
        u32 hq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const bool head_q = PAIR && pio.out.t0 > 0 && lane >= pio.hb && lane < SC_LANES_HAM;
        if (PAIR && pio.out.t0 > 0) { /* wave-uniform */
            if (head_q) {
                const u8* const pq = qual + uniform_u64(pio.nx_o0) + uniform_u32(pio.nx_s) + SC_CHUNK * (lane - pio.hb);
                const u32x4 q0 = load16(pq), q1 = load16(pq + 16);
                hq[0] = q0.x; hq[1] = q0.y; hq[2] = q0.z; hq[3] = q0.w;
                hq[4] = q1.x; hq[5] = q1.y; hq[6] = q1.z; hq[7] = q1.w;
            }
        }
        u32 hb0, hb1;
        hist_totals(h, hb0, hb1);
        if (lane == 0) hb0 -= dumped;
        if (PAIR) {
            if (pio.out.t0 > 0 && head_q) hist32<false>(hist_lane(h), hq, SC_CHUNK);
            pio.in = pio.out; /* what the next read starts from (t0 == 0: from scratch) */
        }
        if (qsums && !dropped && !defer) hist_quality_sums(hb0, hb1, qq & 0x7F, sm.lowq, sm.totq); /* (wave-uniform) */
        PROF(3)
        /* ---- the ends: their first SC_END_PF bytes from the prefetched registers into the small histogram,
           any rest (rare) by a scan; pre-filter totals = body + ends */
        u32 ht0 = hb0, ht1 = hb1;
        if (s > 0 || e < l) { /* wave-uniform */
            u32* const eh = wl->ehist;
            ends_apply(eh, eb, s, e, l);
            if (s > SC_END_PF || l - e > SC_END_PF) { /* wave-uniform */
                RangeSums dummy;
                u64 d0, d1;
                u32 dmp = 0;
                for (int part = 0; part < 2; part++) { /* (one inlined copy of the scan loop for both ends) */
                    const int a = part == 0 ? SC_END_PF : e + SC_END_PF, b = part == 0 ? s : l;
                    if (b > a) dmp += range_scan_fast<false, false, true>(rb, qb, a, b, seq_end, qual_end, wl, h, qq, dummy, nullptr, nullptr, d0, d1);
                }
                u32 x0, x1;
                hist_totals(h, x0, x1);
                if (lane == 0) x0 -= dmp;
                ht0 += x0;
                ht1 += x1;
            }
            wave_sync();
            ht0 += eh[2 * lane];
            ht1 += eh[2 * lane + 1];
            eh[2 * lane] = 0;
            eh[2 * lane + 1] = 0;
            wave_sync();
        }
        PROF(4) /* ends */
        /* ---- medians (src/stats.cpp:352-363), the filter code of r1 as one output read, base-quality histograms */
        if ((u32)l > 0x7FFFFFFFu - bq.bases) bq_hand_over(bq, acc); /* (wave-uniform; the register bins are 32 bits wide) */
        bq.bases += (u32)l;
        bq.pre0 += ht0;
        bq.pre1 += ht1;
        int med_pre = 0, med_body = 0;
        if (l > 0) med_pre = hist_median(ht0, ht1, (u32)l);
        int code = 0;
        if (!dropped && !defer) {
            code = uniform_i32(filter_code(cfg, blen, sm));
            if (code == FPL_PASS_FILTER) { /* (blen > 0 when passing) */
                /* booked as if r1 stayed in one piece; k_redo takes it back for the few reads a middle adapter splits */
                med_body = (s == 0 && e == l) ? med_pre : hist_median(hb0, hb1, (u32)blen);
                bq.post0 += hb0;
                bq.post1 += hb1;
            }
        }
        PROF(6)
        if (lane == 0) {
            ScanRec r;
            r.lowq = sm.lowq; r.nn = sm.nn; r.totq = sm.totq; r.diff = sm.diff;
            r.pos0 = (u32)key0; r.pos1 = (u32)key1;
            r.mm = ((u32)(key0 >> 32) & 0xFFu) | (((u32)(key1 >> 32) & 0xFFu) << 8) | ((u32)med_pre << 16) | ((u32)med_body << 24);
            r.flags = (key0 != ~0ull ? SR_TESTED0 : 0u) | (key1 != ~0ull ? SR_TESTED1 : 0u) | ((u32)code << 8);
            recs[ri] = r;
        }
        if (wsave && lane < 16) wins[ri].w[lane >> 3][lane & 7] = wv;
        PROF(9) /* record */
    }
    PROF_FLUSH(0);
    bq_hand_over(bq, acc);
    __syncthreads();
#ifndef FPL_ABL_NOFLUSH /* (timing experiment: what the blocks' closing atomics cost) */
    for (int k = 0; k < 2; k++) {
        long long* st = counters + (k == 0 ? FPL_OFF_PRE(C) : FPL_OFF_POST(C));
        for (u32 i = threadIdx.x; i < 128; i += blockDim.x)
            if (acc.bqh[k][i]) atomicAdd((u64*)&st[FPL_ST_BASE_QUAL_HIST(C) + i], acc.bqh[k][i]);
    }
#endif
}

/* One Levenshtein confirmation per lane, any adapter: the rare configurations (an adapter beyond 32 bases or with bytes
   outside A / C / G / T) take the wave-cooperative Myers run, one flagged lane after the other.  True in the lanes whose
   window is within thr of the adapter. */
__device__ __forceinline__ bool lev_lanes_any(const DevAdapter* __restrict__ ad, const u8* __restrict__ text, int thr, bool need) {
    bool ok = false;
    u64 m = wave_ballot(need);
    const int alen = ad->len;
    while (m) { /* wave-uniform */
        const int j = __ffsll((long long)m) - 1;
        m &= m - 1;
        const u8* t = (const u8*)readlane_u64((u64)(size_t)text, j);
        const int ed = lev_wave(ad->peq_full, 0, alen, t, alen, thr);
        if (lane_id() == j) ok = ed <= thr;
    }
    return ok;
}

/* append (offset, length) to the post-only EXTRA list: one device atomic per wave */
__device__ __forceinline__ void extra_append(bool want, uint64_t o, u32 len, uint64_t* __restrict__ frag_off, u32* __restrict__ frag_len,
                                             u32* __restrict__ frag_count) {
    const u64 wm = wave_ballot(want);
    if (!wm) return;
    u32 base = 0;
    if (lane_id() == 0) base = atomicAdd(frag_count, (u32)__popcll(wm));
    base = readlane_u32(base, 0);
    if (want) {
        const u32 slot = base + (u32)__popcll(wm & ((1ull << lane_id()) - 1ull));
        frag_off[slot] = o;
        frag_len[slot] = len;
    }
}

/* k_resolve: LANE = READ.  From the ScanRec of a read: the Levenshtein confirmation of the two Hamming argmins
 * (findMiddleAdapters, src/adaptertrimmer.cpp:13-40; searchAdapter's final check, :152-165), the gap; a read that stays in
 * one piece gets its result record, its FilterResult / median counters (src/seprocessor.cpp:265-281) and its plan for the
 * statistics passes right here; a read that is split goes on the REDO list (k_redo scans its fragments), or -- with --break /
 * --mask, where k_break_mask decides the fragments' fate -- gets its record with the two fragments. */
#ifndef FPL_RESOLVE_WPS
#define FPL_RESOLVE_WPS 4 /* waves per SIMD the register allocation must leave room for (16-wave blocks: 4 = one block per CU) */
#endif
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64, FPL_RESOLVE_WPS)
k_resolve(const u8* __restrict__ seq, const uint64_t* __restrict__ off, u32 n_reads, uint64_t n_bytes,
          const DevConfig* __restrict__ cfg, const DevAdapter* __restrict__ ads, ReadState* __restrict__ state,
          const ScanRec* __restrict__ recs, const ScanWin* __restrict__ wins, fpl_read_result* __restrict__ results,
          uint64_t* __restrict__ frag_off, u32* __restrict__ frag_len, u32* __restrict__ frag_count, RedoItem* __restrict__ redo,
          u32* __restrict__ redo_count, long long* __restrict__ counters, u32 C) {
    __shared__ ScanBlockAcc acc;
    __shared__ u64 peq4[2][4]; /* Peq words of A, C, T, G for the two command-line adapters (lev_lanes32 / 64_acgt) */
    __shared__ u32 peq4lo[2][4];
    const int lane = lane_id();
    if (threadIdx.x < 8) {
        const u64 v = ads[threadIdx.x >> 2].peq_full[(0x47544341u >> (8 * (threadIdx.x & 3))) & 0xFFu][0];
        peq4[threadIdx.x >> 2][threadIdx.x & 3] = v;
        peq4lo[threadIdx.x >> 2][threadIdx.x & 3] = (u32)v;
    }
    {
        u64* z = (u64*)&acc;
        for (u32 i = threadIdx.x; i < sizeof(ScanBlockAcc) / 8; i += blockDim.x) z[i] = 0;
    }
    __syncthreads();
    const u8* seq_end = seq + n_bytes;
    const bool adapters = cfg->adapter_enabled != 0, defer = cfg->defer != 0;
    const int al0 = ads[0].len, al1 = ads[1].len;
    const int thr0 = cfg->thr[al0], thr1 = cfg->thr[al1];
    const bool pair32 = cfg->ham_fast && al0 <= 32 && al1 <= 32;
    const int ext = cfg->ext;
    const u32 n_groups = (n_reads + 63) / 64;
    /* Places on the three lists (EXTRA, REDO short / long) are reserved ONCE per block and round: device atomics on one word
       run at ~80 per microsecond, and one per wave and list (31 000 of them for a million reads) were two thirds of this
       kernel's time.  wcnt[w][k]: wave w's entries for list k, then -- after thread 0's turn -- the index of its first. */
    __shared__ u32 wcnt[WAVES][3];
    for (u32 g0 = blockIdx.x * WAVES; g0 < n_groups; g0 += gridDim.x * WAVES) { /* block-uniform */
        const u32 g = g0 + (u32)wave_in_block();
        const u32 ri = g * 64 + (u32)lane;
        const bool live = g < n_groups && ri < n_reads;
        uint64_t o0 = 0, o1 = 0;
        ReadState st = {0, 0, 1, 0};
        ScanRec rec = {0, 0, 0, 0, 0, 0, 0, 0};
        if (live) {
            o0 = off[ri];
            o1 = off[ri + 1];
            st = state[ri];
            rec = recs[ri];
        }
        const int l = (int)(o1 - o0);
        const int s = (int)st.s, e = (int)st.e, blen = e - s;
        const bool dropped = st.dropped != 0;
        const int med_pre = (int)((rec.mm >> 16) & 0xFFu), med_body = (int)(rec.mm >> 24);
        /* pre-filter Stats scalars, src/stats.cpp:265-271,352-374 */
        if (live) {
            if (l > 0) {
                atomicAdd(&acc.medh[0][med_pre], (u64)1);
                atomicAdd(&acc.medb[0][med_pre], (u64)l);
            }
        }
        {
            const u32 nl = (u32)__popcll(wave_ballot(live));
            const u32 lo = wave_sum_u32(live ? ((u32)l & 0xFFFFu) : 0u), hi = wave_sum_u32(live ? ((u32)l >> 16) : 0u);
            if (lane == 0) {
                atomicAdd(&acc.reads[0], (u64)nl);
                atomicAdd(&acc.lensum[0], (u64)lo + ((u64)hi << 16));
            }
        }
        /* ---- findMiddleAdapters: an argmin within the threshold stands (edit distance <= Hamming distance), any other
           needs its edit distance */
        const bool ham = live && !dropped && adapters;
        const bool t0 = ham && (rec.flags & SR_TESTED0), t1 = ham && (rec.flags & SR_TESTED1);
        const int mm0 = (int)(rec.mm & 0xFFu), mm1 = (int)((rec.mm >> 8) & 0xFFu);
        const bool need0 = t0 && mm0 > thr0, need1 = t1 && mm1 > thr1;
        bool ok0 = false, ok1 = false;
        const u8* w0 = seq + o0 + (uint64_t)s + (need0 ? rec.pos0 : 0u);
        const u8* w1 = seq + o0 + (uint64_t)s + (need1 ? rec.pos1 : 0u);
        if (wave_ballot(need0 || need1)) { /* wave-uniform */
            if (pair32) { /* the windows k_scan saved */
                u32 x0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, x1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                if (need0 || need1) {
                    const u32x4* wp = (const u32x4*)&wins[ri];
                    const u32x4 a = wp[0], b = wp[1], c = wp[2], d = wp[3];
                    x0[0] = a.x; x0[1] = a.y; x0[2] = a.z; x0[3] = a.w; x0[4] = b.x; x0[5] = b.y; x0[6] = b.z; x0[7] = b.w;
                    x1[0] = c.x; x1[1] = c.y; x1[2] = c.z; x1[3] = c.w; x1[4] = d.x; x1[5] = d.y; x1[6] = d.z; x1[7] = d.w;
                }
                ok0 = lev_lanes32_acgt_w(x0, al0, thr0, need0, peq4lo[0]);
                ok1 = lev_lanes32_acgt_w(x1, al1, thr1, need1, peq4lo[1]);
            } else if (cfg->ham_fast) { /* A / C / G / T only, <= 64 bases: a 64-bit column per lane */
                ok0 = lev_lanes64_acgt(w0, al0, thr0, need0, peq4[0], seq_end);
                ok1 = lev_lanes64_acgt(w1, al1, thr1, need1, peq4[1], seq_end);
            } else {
                ok0 = lev_lanes_any(&ads[0], w0, thr0, need0);
                ok1 = lev_lanes_any(&ads[1], w1, thr1, need1);
            }
        }
        const int sp = (t0 && (!need0 || ok0)) ? (int)rec.pos0 : -1;
        const int ep = (t1 && (!need1 || ok1)) ? (int)rec.pos1 : -1;
        bool split = false;
        int gs = 0, glen = 0;
        if (sp >= 0 && ep >= 0) {
            const int gstart = max(0, min(sp, ep) - ext), gend = min(blen, max(sp + al0, ep + al1) + ext);
            gs = gstart;
            glen = gend - gstart;
            split = true;
        } else if (sp >= 0) {
            gs = max(0, sp - ext);
            glen = min(blen, sp + al0 + ext) - gs;
            split = true;
        } else if (ep >= 0) {
            gs = max(0, ep - ext);
            glen = min(blen, ep + al1 + ext) - gs;
            split = true;
        }
        /* ---- the read stays in one piece (or was dropped): everything is known */
        const bool whole = live && !dropped && !split;
        const int code = (int)((rec.flags >> 8) & 0xFFu);
        const bool pass = whole && !defer && code == FPL_PASS_FILTER;
        if (whole && !defer) {
            atomicAdd(&acc.fr[code], (u64)1);
            if (pass) {
                atomicAdd(&acc.medh[1][med_body], (u64)1);
                atomicAdd(&acc.medb[1][med_body], (u64)blen);
                atomicAdd(&acc.reads[1], (u64)1);
                atomicAdd(&acc.lensum[1], (u64)blen);
            }
        }
        /* one passing output read that starts within FS_SMAX bases of the read's start: the single statistics pass counts it
           post-filter as the window [start, start + len) of the read; any other passing output read goes on the EXTRA list */
        const bool to_post = pass && (u32)s <= (u32)FS_SMAX;
        if (live && !(split && !defer)) {
            fpl_read_result res;
            res.r1_start = dropped ? 0 : (u32)s;
            res.r1_len = dropped ? 0 : (u32)blen;
            res.frag_start[0] = res.frag_start[1] = 0;
            res.frag_len[0] = res.frag_len[1] = 0;
            res.code[0] = res.code[1] = 0;
            res.kind[0] = res.kind[1] = 0;
            res.median_q_post[0] = res.median_q_post[1] = 0;
            int nf = 0;
            if (whole) {
                res.frag_start[0] = (u32)s;
                res.frag_len[0] = (u32)blen;
                res.code[0] = defer ? 0 : (u8)code;
                res.median_q_post[0] = pass ? (u8)med_body : 0;
                nf = 1;
            } else if (!dropped) { /* split, --break / --mask: Read::breakByGap, src/read.cpp:192-215 */
                const int len1 = gs, len2 = blen - gs - glen;
                if (len1 > 0) {
                    res.frag_start[nf] = (u32)s;
                    res.frag_len[nf] = (u32)len1;
                    res.kind[nf++] = 1;
                }
                if (len2 > 0) {
                    res.frag_start[nf] = (u32)(s + gs + glen);
                    res.frag_len[nf] = (u32)len2;
                    res.kind[nf++] = 2;
                }
            }
            res.n_frag = (u8)nf;
            res.dropped = dropped ? 1 : 0;
            res.median_q_pre = (u8)med_pre;
            res.reserved[0] = res.reserved[1] = res.reserved[2] = 0;
            results[ri] = res;
            state[ri].pad = to_post ? PLAN_TO_POST : 0u;
        }
        /* ---- the lists: passing reads the single statistics pass cannot count post-filter (EXTRA); split reads, whose
           fragments k_redo scans -- the few long ones go to the FRONT of the REDO list, the others fill it from its far end,
           so that the long ones are started first instead of ending a wave's string of items */
        const bool wantx = pass && !to_post;
        const bool wants = live && split && !dropped && !defer && blen <= REDO_LONG;
        const bool wantl = live && split && !dropped && !defer && blen > REDO_LONG;
        const u64 mx = wave_ballot(wantx), ms = wave_ballot(wants), ml = wave_ballot(wantl);
        if (lane == 0) {
            wcnt[wave_in_block()][0] = (u32)__popcll(mx);
            wcnt[wave_in_block()][1] = (u32)__popcll(ms);
            wcnt[wave_in_block()][2] = (u32)__popcll(ml);
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            const u32 k = threadIdx.x;
            u32 tot = 0;
            for (int w = 0; w < WAVES; w++) tot += wcnt[w][k];
            u32 base = 0;
            if (tot) base = atomicAdd(k == 0 ? frag_count : (k == 1 ? redo_count : redo_count + 2), tot);
            for (int w = 0; w < WAVES; w++) {
                const u32 c = wcnt[w][k];
                wcnt[w][k] = base;
                base += c;
            }
        }
        __syncthreads();
        const u64 below = (1ull << lane) - 1ull;
        if (wantx) {
            const u32 slot = wcnt[wave_in_block()][0] + (u32)__popcll(mx & below);
            frag_off[slot] = o0 + (uint64_t)s;
            frag_len[slot] = (u32)blen;
        }
        if (wants || wantl) {
            RedoItem it = {ri, (u32)gs, (u32)glen, 0u};
            if (wants) redo[n_reads - 1u - (wcnt[wave_in_block()][1] + (u32)__popcll(ms & below))] = it;
            else redo[wcnt[wave_in_block()][2] + (u32)__popcll(ml & below)] = it;
        }
        __syncthreads(); /* (wcnt is reused by the next round) */
    }
    __syncthreads();
    scan_acc_flush(acc, counters, C);
}

/* k_redo: one wave per read that a middle adapter splits (a few per cent of the reads at most).  Scans the fragments
 * (passFilter sums, quality histogram -> code, median; src/seprocessor.cpp:265-281) and the gap between them -- the three
 * histograms add up to the one k_scan booked post-filter when r1 passed as a whole, which is taken back here -- and writes the
 * read's record, counters and plan. */
/* PLAN: the read's plan for the statistics pass is written here too (a split read with ONE passing output read next to its start is
 * counted post-filter by the single pass).  Without it -- the sorted pass, whose bucket kernels then need nothing of this kernel and
 * run beside it -- state[] is only read: a split read keeps the plan "not post" that the trim kernel left, and every passing
 * fragment goes on the EXTRA list. */
template <int WAVES, bool PLAN>
__global__ void __launch_bounds__(WAVES * 64)
k_redo(const u8* __restrict__ seq, const u8* __restrict__ qual, const uint64_t* __restrict__ off, u32 n_reads, uint64_t n_bytes,
       const DevConfig* __restrict__ cfg, ReadState* __restrict__ state, const ScanRec* __restrict__ recs,
       fpl_read_result* __restrict__ results, uint64_t* __restrict__ frag_off, u32* __restrict__ frag_len,
       u32* __restrict__ frag_count, const RedoItem* __restrict__ redo, const u32* __restrict__ redo_count,
       u32* __restrict__ redo_next, long long* __restrict__ counters, u32 C) {
    __shared__ alignas(4096) u32 hist_all[WAVES][128 * HIST_COPIES];
    __shared__ ScanWaveLds wlds[WAVES];
    __shared__ ScanBlockAcc acc;
    __shared__ uint64_t fb_off[WAVES][SC_FBUF]; /* passing fragments waiting for a place on the EXTRA list, per wave */
    __shared__ u32 fb_len[WAVES][SC_FBUF];
    const int lane = lane_id();
    ScanWaveLds* const wl = &wlds[wave_in_block()];
    u32* const h = hist_all[wave_in_block()];
    hist_zero(h);
    {
        u64* z = (u64*)&acc;
        for (u32 i = threadIdx.x; i < sizeof(ScanBlockAcc) / 8; i += blockDim.x) z[i] = 0;
    }
    __syncthreads();
    const u8* seq_end = seq + n_bytes;
    const u8* qual_end = qual + n_bytes;
    const int qq = cfg->qualified_qual;
    const u32 n_long = redo_count[2], n_items = n_long + redo_count[0]; /* (long reads first) */
    u32 nbuf = 0; /* entries in this wave's buffer of EXTRA-list entries (wave-uniform) */
    /* The list is walked with a fixed stride, long reads first (REDO_LONG).  Taking the items off a device counter
       (the dequeue k_scan uses, one item at a time; round 4, since removed) evens the hundredfold spread of the items' lengths out no
       better than the ordering does and costs one same-address atomic per wave and item: 6 144 waves + 14 000 items on one word
       are 0.24 ms of a 0.16 ms kernel (c3, measured side by side in round 4).  Round 3 saw a build of that loop that never
       ended on the GPU; its code had a second loop header BEHIND the dequeue that tested the stale lane-0 value again.  Today's
       build of either form has the atomic in the loop header (tools/dequeue_isa.py checks that for every work-counter loop of
       the library, tests/test_isa_dequeue.py runs it). */
    (void)redo_next;
    for (u32 it = blockIdx.x * WAVES + wave_in_block(); it < n_items; it += gridDim.x * WAVES) {
        const RedoItem item = redo[it < n_long ? it : n_reads - 1u - (it - n_long)];
        const u32 ri = uniform_u32(item.ri);
        const int gs = uniform_i32((int)item.gs), glen = uniform_i32((int)item.glen);
        const uint64_t o0 = uniform_u64(off[ri]);
        const ReadState st = state[ri];
        const int s = uniform_i32((int)st.s), e = uniform_i32((int)st.e), blen = e - s;
        const ScanRec rec = recs[ri];
        const u32 rflags = uniform_u32(rec.flags), rmm = uniform_u32(rec.mm);
        const u8* rb = seq + o0;
        const u8* qb = qual + o0;
        /* Read::breakByGap, src/read.cpp:192-215 */
        const int len1 = gs, len2 = blen - gs - glen;
        /* per-fragment outcome in scalars (an indexed array would be demoted to scratch): left (L), right (R) */
        u32 codeL = 0, codeR = 0, medL = 0, medR = 0;
        bool passL = false, passR = false;
        /* what k_scan booked post-filter for r1 as a whole comes off again: its histogram is the sum of the three below */
        const bool undo = ((rflags >> 8) & 0xFFu) == FPL_PASS_FILTER;
        u32 d0 = 0, d1 = 0; /* this lane's two bins: (passing fragments) - (r1 when it was booked), modulo 2^32 */
#pragma unroll
        for (int f = 0; f < 3; f++) { /* left fragment, right fragment, the gap */
            const int fa = f == 0 ? s : (f == 1 ? s + gs + glen : s + gs);
            const int flen = f == 0 ? len1 : (f == 1 ? len2 : glen);
            if (f == 2 ? (!undo || flen <= 0) : flen <= 0) continue; /* wave-uniform */
            RangeSums fs = {0, 0, 0, 0};
            u64 k0, k1;
            u32 t0, t1;
            /* (a wave alone with its read: the next tile's lines are requested one tile ahead) */
            const u32 dmp = range_scan_fast<true, false, false, 7, FPL_REDO_PREFETCH != 0>(rb, qb, fa, fa + flen, seq_end, qual_end, wl, h, qq, fs, nullptr, nullptr, k0, k1);
            hist_totals(h, t0, t1);
            if (lane == 0) t0 -= dmp;
            if (f != 2) hist_quality_sums(t0, t1, qq & 0x7F, fs.lowq, fs.totq);
            if (undo) {
                d0 -= t0;
                d1 -= t1;
            }
            if (f == 2) continue;
            const int code = uniform_i32(filter_code(cfg, flen, fs));
            const bool pass = code == FPL_PASS_FILTER;
            int med = 0;
            if (pass) {
                med = hist_median(t0, t1, (u32)flen);
                d0 += t0;
                d1 += t1;
            }
            if (f == 0) { codeL = (u32)code; medL = (u32)med; passL = pass; }
            else { codeR = (u32)code; medR = (u32)med; passR = pass; }
            if (lane == 0) {
                atomicAdd(&acc.fr[code], (u64)1);
                if (pass) {
                    atomicAdd(&acc.medh[1][med], (u64)1);
                    atomicAdd(&acc.medb[1][med], (u64)flen);
                    atomicAdd(&acc.reads[1], (u64)1);
                    atomicAdd(&acc.lensum[1], (u64)flen);
                }
            }
        }
        /* the output reads in order: the left fragment when it has bases, then the right one */
        const bool hasL = len1 > 0, hasR = len2 > 0;
        const int nf = (hasL ? 1 : 0) + (hasR ? 1 : 0);
        const u32 fsL = (u32)s, fsR = (u32)(s + gs + glen);
        u32 r_fs[2], r_fl[2], r_code[2], r_kind[2], r_med[2];
        bool r_pass[2];
        r_fs[0] = hasL ? fsL : (hasR ? fsR : 0u);
        r_fl[0] = hasL ? (u32)len1 : (hasR ? (u32)len2 : 0u);
        r_code[0] = hasL ? codeL : (hasR ? codeR : 0u);
        r_kind[0] = hasL ? 1u : (hasR ? 2u : 0u);
        r_med[0] = hasL ? medL : (hasR ? medR : 0u);
        r_pass[0] = hasL ? passL : (hasR && passR);
        r_fs[1] = (hasL && hasR) ? fsR : 0u;
        r_fl[1] = (hasL && hasR) ? (u32)len2 : 0u;
        r_code[1] = (hasL && hasR) ? codeR : 0u;
        r_kind[1] = (hasL && hasR) ? 2u : 0u;
        r_med[1] = (hasL && hasR) ? medR : 0u;
        r_pass[1] = hasL && hasR && passR;
        if (d0) atomicAdd(&acc.bqh[1][2 * lane], (u64)(long long)(int)d0);
        if (d1) atomicAdd(&acc.bqh[1][2 * lane + 1], (u64)(long long)(int)d1);
        /* one passing output read that starts within FS_SMAX bases of the read's start (Read::breakByGap drops an empty
           side): the statistics pass counts it post-filter as a window of the read, like an unsplit r1 */
        const bool to_post = PLAN && nf == 1 && r_pass[0] && r_fs[0] <= (u32)FS_SMAX;
        if (lane == 0) {
            fpl_read_result res;
            res.r1_start = (u32)s;
            res.r1_len = (u32)blen;
            res.frag_start[0] = r_fs[0]; res.frag_start[1] = r_fs[1];
            res.frag_len[0] = r_fl[0]; res.frag_len[1] = r_fl[1];
            res.n_frag = (u8)nf;
            res.dropped = 0;
            res.code[0] = (u8)r_code[0]; res.code[1] = (u8)r_code[1];
            res.kind[0] = (u8)r_kind[0]; res.kind[1] = (u8)r_kind[1];
            res.median_q_pre = (u8)((rmm >> 16) & 0xFFu);
            res.median_q_post[0] = (u8)r_med[0]; res.median_q_post[1] = (u8)r_med[1];
            res.reserved[0] = res.reserved[1] = res.reserved[2] = 0;
            results[ri] = res;
            if (PLAN) {
                ReadState ns = st;
                ns.pad = to_post ? PLAN_TO_POST : 0u;
                if (to_post) { /* the window the statistics pass works on is the fragment, not r1 */
                    ns.s = r_fs[0];
                    ns.e = r_fs[0] + r_fl[0];
                }
                state[ri] = ns;
            }
            /* passing fragments wait in this wave's buffer for a place on the EXTRA list */
            u32 slot = nbuf;
            if (r_pass[0] && !to_post) {
                fb_off[wave_in_block()][slot] = o0 + r_fs[0];
                fb_len[wave_in_block()][slot] = r_fl[0];
                slot++;
            }
            if (nf > 1 && r_pass[1]) {
                fb_off[wave_in_block()][slot] = o0 + r_fs[1];
                fb_len[wave_in_block()][slot] = r_fl[1];
            }
        }
        nbuf += (r_pass[0] && !to_post ? 1u : 0u) + (nf > 1 && r_pass[1] ? 1u : 0u); /* (wave-uniform) */
        if (nbuf > SC_FBUF - 2) { /* the buffer is full (a wave with more than fifteen items): one atomic for its entries */
            u32 base = 0;
            if (lane == 0) base = atomicAdd(frag_count, nbuf);
            base = readlane_u32(base, 0);
            wave_sync();
            if ((u32)lane < nbuf) {
                frag_off[base + lane] = fb_off[wave_in_block()][lane];
                frag_len[base + lane] = fb_len[wave_in_block()][lane];
            }
            wave_sync();
            nbuf = 0;
        }
    }
    /* what is left in the waves' buffers: ONE place reservation for the block (a device atomic per item -- 25 000 on one word
       for a million reads -- was most of this kernel's time) */
    __shared__ u32 wbase[WAVES];
    if (lane == 0) wbase[wave_in_block()] = nbuf;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 tot = 0;
        for (int w = 0; w < WAVES; w++) tot += wbase[w];
        u32 base = tot ? atomicAdd(frag_count, tot) : 0u;
        for (int w = 0; w < WAVES; w++) {
            const u32 c = wbase[w];
            wbase[w] = base;
            base += c;
        }
    }
    __syncthreads();
    if ((u32)lane < nbuf) {
        frag_off[wbase[wave_in_block()] + lane] = fb_off[wave_in_block()][lane];
        frag_len[wbase[wave_in_block()] + lane] = fb_len[wave_in_block()][lane];
    }
    scan_acc_flush(acc, counters, C);
}

/* =========================================================================================
 * The statistics pass over reads SORTED BY THEIR FRONT TRIM (FPL_OPT_SORTSTATS).
 *
 * k_stats above spends two table updates per base: pre cycle p and post cycle p - s.  The second one exists only because
 * s differs from read to read.  Statistics are sums, so the reads may be taken in any order and in any grouping: when
 * every read of a slice has the SAME front trim s, the slice's post-filter table is its pre-filter table shifted by s,
 * minus the bases behind the reads' ends e -- so a block keeps the pre table (one update per base, as before) and a
 * "not post" table that only the bases at p >= e touch, and k_stats_reduce_sorted computes
 *      pre[c]  += slab.pre[c]           post[c - s] += slab.pre[c] - slab.notpost[c]   (c >= s)
 * per slab.  Buckets: front trim 0..FS_SMAX (a read k_scan marked PLAN_TO_POST), and one for the reads that are not
 * counted post-filter by this pass at all (dropped, failed, split or far-trimmed reads).  A front trim that too few reads
 * of the batch share (StatsTune::min_bucket) is not worth a slab per cycle tile: those reads are filed under "not post"
 * and handed to the EXTRA pass instead, like the far-trimmed ones.
 *   k_bucket_count    reads per (block of 256 reads, bucket)
 *   k_bucket_scan     their prefix sums along the blocks, and the buckets' totals
 *   k_bucket_plan     bucket -> range of the sorted order, ranges -> slices of <= `per` reads (device-side table)
 *   k_bucket_scatter  (start, length, e) of every read into its bucket's range (order inside a bucket: arbitrary)
 *   k_stats_sorted    block = (slice, cycle tile)
 *   k_stats_reduce_sorted
 * ======================================================================================= */
constexpr int FS_NB = FS_SMAX + 2;        /* buckets */
constexpr int FS_B_NOPOST = FS_SMAX + 1;  /* the bucket of the reads this pass does not count post-filter */
/* sort workspace (u32 words; the first SW_SLICES are zeroed before every batch) */
constexpr int SW_CNT = 0;                /* [FS_NB] reads per bucket */
constexpr int SW_CUR = FS_NB;            /* [FS_NB] next free position of the range of bucket b */
constexpr int SW_MAP = 2 * FS_NB;        /* [FS_NB] the bucket that bucket b is filed under: b, or FS_B_NOPOST when too small */
constexpr int SW_NSLICES = 3 * FS_NB;    /* slices in the table */
constexpr int SW_WORK = 3 * FS_NB + 1;   /* k_stats_sorted: the next (tile, slice) / (tile, group) item */
constexpr int SW_NGROUPS = 3 * FS_NB + 2; /* groups in the table behind the slices */
constexpr int SW_SLICES = 3 * FS_NB + 3; /* [slices][4]: begin, end (positions of the sorted order), front trim + 1 (0: not
                                            post-filter), unused; behind them [groups][2]: first slice, one past the last */
/* Beyond the typical read length a (tile, slice) item holds a handful of rows -- the few reads of the slice that are that long --
   but still costs its block the zeroing of 80 KB of tables and a slab for the reduce kernel to read (configs[3], 300 b ... 200 kb:
   1 % of the rows, most of the slabs).  From cycle tile `hi_tile` on the items are therefore (tile, GROUP of up to FS_GROUP
   consecutive slices of one front trim): one table set, one slab -- filed under the group's first slice that has rows -- for as
   long as the rows counted so far fit the 14-bit fields (counted, not assumed: a group whose rows would not fit is handed over in
   pieces). */
#ifndef FPL_STATS_GROUP
#define FPL_STATS_GROUP 16
#endif
constexpr int FS_GROUP = FPL_STATS_GROUP;
constexpr int FS_BSTRIDE = 2 * FS_T;     /* LDS cells per base class: [FS_T pre | FS_T not-post] */

__device__ __forceinline__ u32 plan_bucket(const ReadState& st) {
    return (st.pad & PLAN_TO_POST) ? st.s : (u32)FS_B_NOPOST; /* (PLAN_TO_POST implies s <= FS_SMAX) */
}

/* one block per FS_SORT_READS consecutive reads (FS_SORT_PER per thread); blkcnt[b][block] = the block's reads of bucket b */
constexpr int FS_SORT_BLK = 256;
#ifndef FPL_SORT_PER
#define FPL_SORT_PER 8 /* (4 M reads of 2 kb: 0.82 ms of bucket kernels with 1, 0.24 with 4, 0.18 with 8, 0.17 with 16) */
#endif
constexpr int FS_SORT_PER = FPL_SORT_PER; /* (one read per thread: four times the blocks and four times the prefix-sum rows -- 0.8 ms of bucket
                                  kernels for a batch of four million short reads) */
constexpr int FS_SORT_READS = FS_SORT_BLK * FS_SORT_PER;
__global__ void __launch_bounds__(FS_SORT_BLK)
k_bucket_count(const ReadState* __restrict__ plan, u32 n_reads, u32* __restrict__ blkcnt) {
    __shared__ u32 h[FS_NB];
    if (threadIdx.x < FS_NB) h[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < FS_SORT_PER; k++) {
        const u32 it = (blockIdx.x * FS_SORT_PER + k) * FS_SORT_BLK + threadIdx.x;
        if (it < n_reads) atomicAdd(&h[plan_bucket(plan[it])], 1u);
    }
    __syncthreads();
    if (threadIdx.x < FS_NB) blkcnt[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = h[threadIdx.x];
}

/* blkcnt[b][*] -> its exclusive prefix sums: where, inside bucket b's range, the reads of every block go.  The sorted
   order is then ascending in the input inside every bucket -- a slice walks the batch front to back like the unsorted
   pass does, instead of hopping between whatever blocks happened to reserve their places one after another (measured
   with 18 GB batches: the hopping costs more address-translation misses than the saved table updates are worth) */
__global__ void __launch_bounds__(256)
k_bucket_scan(u32* __restrict__ blkcnt, u32 nblk, u32* __restrict__ sw) {
    __shared__ u32 wsum[4];
    u32* row = blkcnt + (size_t)blockIdx.x * nblk;
    const u32 chunk = (nblk + 255u) / 256u;
    const u32 lo = min(nblk, threadIdx.x * chunk), hi = min(nblk, lo + chunk);
    u32 sum = 0;
    for (u32 i = lo; i < hi; i++) sum += row[i];
    const u32 incl = wave_scan_incl_u32(sum);
    if (lane_id() == 63) wsum[wave_in_block()] = incl;
    __syncthreads();
    u32 run = incl - sum;
    for (int w = 0; w < wave_in_block(); w++) run += wsum[w];
    if (threadIdx.x == 255) sw[SW_CNT + blockIdx.x] = run + sum; /* reads in the bucket */
    for (u32 i = lo; i < hi; i++) {
        const u32 v = row[i];
        row[i] = run;
        run += v;
    }
}

/* one thread: 98 buckets */
__global__ void __launch_bounds__(128)
k_bucket_plan(u32* __restrict__ sw, u32 per, u32 min_bucket, u32 max_slices, u32 group) {
    __shared__ u32 cnt[FS_NB];
    for (u32 b = threadIdx.x; b < (u32)FS_NB; b += blockDim.x) cnt[b] = sw[SW_CNT + b];
    __syncthreads();
    if (threadIdx.x != 0) return;
    u32 nopost = cnt[FS_B_NOPOST];
    for (int b = 0; b <= FS_SMAX; b++) {
        const u32 c = cnt[b];
        const bool own = c >= min_bucket && c > 0;
        sw[SW_MAP + b] = own ? (u32)b : (u32)FS_B_NOPOST;
        if (!own) {
            nopost += c;
            cnt[b] = 0;
        }
    }
    sw[SW_MAP + FS_B_NOPOST] = FS_B_NOPOST;
    cnt[FS_B_NOPOST] = nopost;
    u32 pos = 0, ns = 0, ng = 0;
    u32* const gt = sw + SW_SLICES + 4 * (size_t)max_slices; /* groups: runs of <= group consecutive slices of one bucket */
    for (int b = 0; b < FS_NB; b++) {
        const u32 c = cnt[b];
        sw[SW_CUR + b] = pos;
        if (c) {
            const u32 k = (c + per - 1) / per;
            const u32 pb = (((c + k - 1) / k) + 63u) / 64u * 64u; /* <= per: per is a multiple of 64 */
            const u32 ns0 = ns;
            for (u32 i = 0; i < k; i++) {
                const u32 begin = pos + i * pb, end = min(pos + c, begin + pb);
                if (begin < end && ns < max_slices) {
                    u32* e = sw + SW_SLICES + 4 * ns++;
                    e[0] = begin;
                    e[1] = end;
                    e[2] = b == FS_B_NOPOST ? 0u : (u32)b + 1u;
                    e[3] = 0;
                }
            }
            for (u32 g0 = ns0; g0 < ns; g0 += group) {
                gt[2 * ng] = g0;
                gt[2 * ng + 1] = min(ns, g0 + group);
                ng++;
            }
        }
        pos += c;
    }
    sw[SW_NSLICES] = ns;
    sw[SW_NGROUPS] = ng;
}

__global__ void __launch_bounds__(FS_SORT_BLK)
k_bucket_scatter(const uint64_t* __restrict__ off, const ReadState* __restrict__ plan, u32 n_reads, u32* __restrict__ sw,
                 const u32* __restrict__ blkoff,
                 uint64_t* __restrict__ st_off, u32* __restrict__ st_len, u32* __restrict__ st_e,
                 uint64_t* __restrict__ frag_off, u32* __restrict__ frag_len, u32* __restrict__ frag_count) {
    __shared__ u32 h[FS_NB], base[FS_NB], nextra, xbase;
    if (threadIdx.x < FS_NB) h[threadIdx.x] = 0;
    if (threadIdx.x == 0) nextra = 0;
    __syncthreads();
    ReadState st[FS_SORT_PER];
    u32 b[FS_SORT_PER], rank[FS_SORT_PER], xr[FS_SORT_PER];
    bool handed[FS_SORT_PER]; /* a passing read whose front trim is too rare for a slice of its own: post-filter through EXTRA */
#pragma unroll
    for (int k = 0; k < FS_SORT_PER; k++) {
        const u32 it = (blockIdx.x * FS_SORT_PER + k) * FS_SORT_BLK + threadIdx.x;
        st[k] = {0, 0, 0, 0};
        b[k] = rank[k] = xr[k] = 0;
        handed[k] = false;
        if (it < n_reads) {
            st[k] = plan[it];
            const u32 b0 = plan_bucket(st[k]);
            b[k] = sw[SW_MAP + b0];
            rank[k] = atomicAdd(&h[b[k]], 1u);
            handed[k] = b0 != (u32)FS_B_NOPOST && b[k] == (u32)FS_B_NOPOST;
            if (handed[k]) xr[k] = atomicAdd(&nextra, 1u);
        }
    }
    __syncthreads();
    /* a bucket with slices of its own: this block's place in the range (k_bucket_scan); "not post" collects reads of
       several source buckets and hands its places out as the blocks come (few reads, any order) */
    if (threadIdx.x < FS_NB && h[threadIdx.x])
        base[threadIdx.x] = threadIdx.x == FS_B_NOPOST ? atomicAdd(&sw[SW_CUR + FS_B_NOPOST], h[threadIdx.x])
                                                       : sw[SW_CUR + threadIdx.x] + blkoff[(size_t)threadIdx.x * gridDim.x + blockIdx.x];
    if (threadIdx.x == 0 && nextra) xbase = atomicAdd(frag_count, nextra);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < FS_SORT_PER; k++) {
        const u32 it = (blockIdx.x * FS_SORT_PER + k) * FS_SORT_BLK + threadIdx.x;
        if (it < n_reads) {
            const u32 pos = base[b[k]] + rank[k];
            const uint64_t o = off[it];
            st_off[pos] = (FPL_ABL & 8) ? (o & ~(uint64_t)127) : o; /* (timing experiment: rows on cache-line boundaries) */
            st_len[pos] = (u32)(off[it + 1] - o);
            st_e[pos] = st[k].e;
            if (handed[k]) {
                frag_off[xbase + xr[k]] = o + st[k].s;
                frag_len[xbase + xr[k]] = st[k].e - st[k].s;
            }
        }
    }
}

#ifdef FPL_PROF_BLOCKS
/* profiling only: when every block of k_stats_sorted ran, and where (tools/block_timeline.py) */
__device__ unsigned long long g_blockprof[1 << 17][2];
#endif
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64, (2 * WAVES + 3) / 4)
k_stats_sorted(const u8* __restrict__ seq, const u8* __restrict__ qual, uint64_t n_bytes,
               const uint64_t* __restrict__ st_off, const u32* __restrict__ st_len, const u32* __restrict__ st_e,
               u32* __restrict__ sw, u32 max_slices, u32 n_tiles, long long* __restrict__ counters,
               u64* __restrict__ scratch, u8* __restrict__ flags, u32 C, u32 hi_tile, u32 max_rows) {
    (void)C;
    (void)counters;
    constexpr int N_INC = FPL_OPT_INCVALU ? 0 : 256;
#if FPL_OPT_KMER6
    /* 80 KB to the byte: [8][FS_T pre | FS_T not-post] cells (64 KB) + the 6-mer table (16 KB).  Class row 0 of the cells is the two
       5-mer tables, class row 2 the block's scalars (FPL_OPT_KMER6 above) */
    static_assert(FPL_OPT_INCVALU && FPL_OPT_STATSETUP, "FPL_OPT_KMER6 has no room for the increment table and builds on the v_dot4 row set-up");
    __shared__ u64 lds_all[8 * FS_BSTRIDE + 2048];
    static_assert(sizeof(u64) * (8 * FS_BSTRIDE + 2048) <= 81920, "two blocks per CU");
    u32* const k6 = (u32*)lds_all;   /* 6-mers of window pairs counted pre- AND post-filter (in front: a ds offset holds 16 bits) */
    u64* const tbl = lds_all + 2048;
    u64* const inc_of = lds_all; /* (unused) */
    u32* const kmer = (u32*)tbl;  /* class row 0: [0,1024) 5-mers counted pre-filter only; [1024,2048): pre- AND post-filter */
    u32* const scal = (u32*)(tbl + 2 * FS_BSTRIDE);   /* class row 2 */
    u32& any_work = scal[0];
    u32& cur_item = scal[1];
    u32& cls_mask = scal[2];
    constexpr u32 LDS_ROWS = 0xFAu; /* the class rows that are cells: 1, 3 .. 7 */
#else
    __shared__ u64 lds_all[1024 + N_INC + 8 * FS_BSTRIDE];
    static_assert(sizeof(u64) * (1024 + N_INC + 8 * FS_BSTRIDE) <= 81920, "two blocks per CU");
    u32* const kmer = (u32*)lds_all; /* [0,1024): 5-mers counted pre-filter only; [1024,2048): pre- AND post-filter */
    u64* const inc_of = lds_all + 1024; /* (without FPL_OPT_INCVALU: the packed increment of every quality byte) */
    u64* const tbl = inc_of + N_INC;    /* [8][FS_T pre | FS_T not-post] */
    __shared__ u32 any_work, cur_item, cls_mask;
    constexpr u32 LDS_ROWS = 0xFFu;
#endif
    u32* const kpre = kmer;
    u32* const kpost = kmer + 1024;
    const int lane = lane_id();
    const u32 lane8 = 8u * (u32)lane;
    static_assert(8 * FS_BSTRIDE == 8192, "a class row of the LDS tables is 8192 bytes apart from the next: FPL_FS_CELL");
    const u8* seq_end = seq + n_bytes;
    const u8* qual_end = qual + n_bytes;
    long long* kg0 = counters + FPL_OFF_PRE(C) + FPL_ST_KMER(C);
    long long* kg1 = counters + FPL_OFF_POST(C) + FPL_ST_KMER(C);
    for (u32 q = threadIdx.x; q < (u32)N_INC; q += blockDim.x)
        inc_of[q] = (u64)q | (1ull << 22) | ((u64)(q >= '5') << 36) | ((u64)(q >= '?') << 50);
    const u32 n_slices = uniform_u32(sw[SW_NSLICES]);
    const u32 n_groups = uniform_u32(sw[SW_NGROUPS]);
    const u32* const gtab = sw + SW_SLICES + 4 * (size_t)max_slices;
    const u32 lo_tiles = min(hi_tile, n_tiles);                 /* tiles whose items are single slices */
    const u32 lo_items = lo_tiles * n_slices;
    const u32 n_items = lo_items + (n_tiles - lo_tiles) * n_groups;
    /* The blocks are persistent (two per CU) and take (tile, slice) items -- (tile, group of slices) from tile hi_tile on -- off
       one counter, tile by tile -- the heavy low tiles first.  A grid of one block per item leaves a fifth of the chip idle: the
       hardware hands blocks to the XCDs in turn and in order, so every XCD waits for the one whose slots are all taken by long
       blocks (block timeline in profiles/r02_ab) */
    auto kmer_flush = [&]() { /* the 5-mer tables (and what the 6-mer table holds of them) into the counters */
        for (u32 i = threadIdx.x; i < 1024 && !(FPL_ABL & 64); i += blockDim.x) {
            u32 both = kpost[i];
            const u32 pre_only = kpre[i];
#if FPL_OPT_KMER6
            /* 5-mer i is the first five bases of the 6-mers 4 i .. 4 i + 3 and the last five of the 6-mers i + 1024 a */
            both += k6[4 * i] + k6[4 * i + 1] + k6[4 * i + 2] + k6[4 * i + 3] + k6[i] + k6[i + 1024] + k6[i + 2048] + k6[i + 3072];
#endif
            if (both + pre_only) atomicAdd((u64*)&kg0[i], (u64)both + pre_only);
            if (both) atomicAdd((u64*)&kg1[i], (u64)both);
        }
    };
#if FPL_OPT_KMER6 && FPL_OPT_KMERKEEP
    for (u32 i = threadIdx.x; i < 2048 + FS_BSTRIDE; i += blockDim.x) lds_all[i] = 0; /* the 6-mer table and class row 0 (the 5-mer tables): once */
#endif
    for (;;) {
    __syncthreads(); /* (everybody is done with the previous item's tables and cur_item) */
    if (threadIdx.x == 0) cur_item = atomicAdd(&sw[SW_WORK], 1u);
    __syncthreads();
    const u32 item = cur_item;
    if (item >= n_items) break; /* block-uniform */
    u32 tile, sl0, sl1;
    if (item < lo_items) {
        tile = item / n_slices;
        sl0 = item - tile * n_slices;
        sl1 = sl0 + 1;
    } else {
        const u32 j = item - lo_items;
        tile = lo_tiles + j / n_groups;
        const u32 g = j - (tile - lo_tiles) * n_groups;
        sl0 = uniform_u32(gtab[2 * g]);
        sl1 = uniform_u32(gtab[2 * g + 1]);
    }
#ifdef FPL_PROF_BLOCKS
    const unsigned long long prof_t0 = wall_clock64();
#endif
    const u32 sp1 = uniform_u32(sw[SW_SLICES + 4 * sl0 + 2]); /* (one front trim per group) */
    const bool tp = sp1 != 0;            /* the slice's reads pass unsplit, all with ... */
    const int s = tp ? (int)sp1 - 1 : 0; /* ... this front trim */
    const u32 tile_start = tile * FS_T;
    const u32 c0 = tile_start + 8 * lane;
    /* hand-over: the two tables as one slab (pre cells, then not-post cells, both in LDS slot order) -- the rows of the base
       classes that occur: a byte's class is its low three bits, so DNA fills four or five of the eight rows (A 1, C 3, T 4,
       G 7, N 6), and the slab's flag byte says which; k_stats_reduce_sorted reads no others.  Filed under slice `leader`. */
    auto hand_over = [&](u32 leader) {
        __syncthreads(); /* (every wave is done with its rows) */
        if (threadIdx.x == 0) cls_mask = 0;
        __syncthreads();
        {
            u32 m = 0;
            for (u32 i = threadIdx.x; i < 8 * FS_T; i += blockDim.x)
                if (((LDS_ROWS >> (i / FS_T)) & 1u) && tbl[(i / FS_T) * FS_BSTRIDE + (i % FS_T)]) m |= 1u << (i / FS_T);
            if (m) atomicOr(&cls_mask, m);
        }
        __syncthreads();
        const u32 cmask = cls_mask;
        const size_t slab = (size_t)tile * max_slices + leader;
        u64* dst = scratch + slab * FS_SLAB;
        for (u32 i = threadIdx.x; i < 8 * FS_T; i += blockDim.x) {
            if (!((cmask >> (i / FS_T)) & 1u)) continue;
            const u32 cell = (i / FS_T) * FS_BSTRIDE + (i % FS_T);
            dst[i] = tbl[cell];
            if (tp) dst[8 * FS_T + i] = tbl[cell + FS_T];
        }
        if (threadIdx.x == 0) {
            flags[n_tiles + slab] = (u8)cmask;
            flags[tile] = 1;
#ifdef FPL_PROF_BLOCKS
            if (item < (1u << 17)) {
                g_blockprof[item][0] = prof_t0;
                g_blockprof[item][1] = (wall_clock64() << 8) | (__builtin_amdgcn_s_getreg(6164) & 0xF);
            }
#endif
        }
        if (!(FPL_OPT_KMER6 && FPL_OPT_KMERKEEP)) kmer_flush();
        __syncthreads(); /* (the tables may be zeroed again) */
    };
    bool open = false; /* block-uniform: the tables are zeroed and may hold rows */
    u32 held = 0, leader = 0;
    for (u32 slice = sl0; slice < sl1; slice++) { /* block-uniform */
    const u32 i_begin = uniform_u32(sw[SW_SLICES + 4 * slice]), i_end = uniform_u32(sw[SW_SLICES + 4 * slice + 1]);
    /* the rows of this slice in this tile (most (tile, slice) pairs beyond the typical read length have none: find out before
       paying for the tables) */
    __syncthreads();
    if (threadIdx.x == 0) any_work = 0;
    __syncthreads();
    {
        u32 mine = 0;
        for (u32 it = i_begin + threadIdx.x; it < i_end; it += blockDim.x) mine += st_len[it] > tile_start ? 1u : 0u;
        const u32 w = wave_sum_u32(mine);
        if (w && lane == 0) atomicAdd(&any_work, w);
    }
    __syncthreads();
    const u32 rows = any_work;
    if (!rows) continue; /* block-uniform */
    if (open && held + rows > max_rows) { /* the packed 14-bit fields hold max_rows rows: what is there goes out first */
        hand_over(leader);
        open = false;
    }
    if (!open) {
#if FPL_OPT_KMER6
        /* (class row 0 = the 5-mer tables, zeroed as cells; class row 2 = the scalars: left alone) */
#if FPL_OPT_KMERKEEP
        for (u32 i = threadIdx.x + FS_BSTRIDE; i < 8 * FS_BSTRIDE; i += blockDim.x) /* the cells: class rows 1, 3 .. 7 */
            if (i / FS_BSTRIDE != 2) tbl[i] = 0;
#else
        for (u32 i = threadIdx.x; i < 8 * FS_BSTRIDE + 2048; i += blockDim.x)
            if (i / FS_BSTRIDE != 4) lds_all[i] = 0; /* (words 4096 .. 5119 of the array = class row 2: the scalars) */
#endif
#else
        for (u32 i = threadIdx.x; i < 8 * FS_BSTRIDE; i += blockDim.x) tbl[i] = 0;
        for (u32 i = threadIdx.x; i < 2048; i += blockDim.x) kmer[i] = 0;
#endif
        __syncthreads();
        open = true;
        held = 0;
        leader = slice;
    }
    held += rows;

    for (u32 ib = i_begin + 64 * wave_in_block(); ib < i_end; ib += 64 * WAVES) {
        const u32 it = ib + lane;
        u32 L = 0, E = 0;
        uint64_t st = 0;
        if (it < i_end) {
            st = st_off[it];
            L = st_len[it];
            E = st_e[it];
        }
        u64 m = wave_ballot(L > tile_start);
        while (m) {
            u32x2 svG[CS_GROUP], qvG[CS_GROUP];
            u32 haloG[CS_GROUP], LG[CS_GROUP], EG[CS_GROUP];
            u32 haloAll = 0; /* lane g: the four bases in front of row g's tile */
#if FPL_OPT_KMER6
            int bitG[CS_GROUP];
            u32 odd_rows = 0; /* wave-uniform: (lane of the read + 1) of the group's rows that hold a byte of class 0 or 2, a byte each */
#endif
#pragma unroll
            for (int g = 0; g < CS_GROUP; g++) {
                /* (lanes behind the end of the read load nothing: with the set-up below they hold 'A's, so that they do not
                   send the row's validity test down the exact path; none of their bytes is counted either way) */
                svG[g] = FPL_OPT_STATSETUP ? u32x2{0x41414141u, 0x41414141u} : u32x2{0, 0};
                qvG[g] = {0, 0};
                haloG[g] = 0;
                LG[g] = EG[g] = 0;
#if FPL_OPT_KMER6
                bitG[g] = 0;
#endif
                if (m) {
                    const int bit = __ffsll(m) - 1;
                    m &= m - 1;
#if FPL_OPT_KMER6
                    bitG[g] = bit;
#endif
                    LG[g] = readlane_u32(L, bit);
                    EG[g] = readlane_u32(E, bit);
                    const uint64_t start = readlane_u64(st, bit);
                    if (LG[g] > c0) {
                        /* (one wave-uniform bounds test per row instead of one per lane and load measured 9 % SLOWER, round 3) */
                        svG[g] = load8_guard(seq + start + c0, seq_end);
                        qvG[g] = load8_guard(qual + start + c0, qual_end);
                    }
                    if (FPL_OPT_STATSETUP) {
                        if (lane == g && tile_start >= 4) haloAll = load4_guard(seq + start + tile_start - 4, seq_end);
                    } else if (lane == 0 && tile_start >= 4) {
                        haloG[g] = load4_guard(seq + start + tile_start - 4, seq_end);
                    }
                }
            }
            /* the four bases in front of the tile, of all rows of the group at once (lane g: row g): their packed codes and
               which of them are no bases -- once per group instead of once per row */
            u32 packAll = 0, invAll = 0;
            if (FPL_OPT_STATSETUP) {
                const u32 vA = kmer_codes(haloAll);
                packAll = kmer_pack_dot(vA);
                invAll = invalid_nibble(perm_lo(0x47435441u, vA), haloAll);
            }
#pragma unroll
            for (int g = 0; g < CS_GROUP; g++) {
                const u32 itemL = uniform_u32(LG[g]);
                if (itemL <= tile_start) continue; /* wave-uniform: empty slot of the last group */
                const u32 sw2[2] = {svG[g].x, svG[g].y};
                const u32 qw[2] = {qvG[g].x, qvG[g].y};
                const int e = (int)uniform_u32(EG[g]);
#if FPL_OPT_STATSETUP
                /* (bytes of the row in this lane: only the rows that hold an end of the read or of r1 ask -- FPL_FB_ROW(.., false)) */
#define FPL_FS_NVALID (itemL > c0 ? (int)min(8u, itemL - c0) : 0)
#else
                const int nvalid = itemL > c0 ? (int)min(8u, itemL - c0) : 0;
#define FPL_FS_NVALID nvalid
#endif
                const bool have_halo = lane > 0 || tile_start >= 4;
                u32 W, okmask;
                bool allok = false; /* wave-uniform */
#if FPL_OPT_STATSETUP
                {
                    /* 5-mers: the 2-bit codes of a dword packed by one v_dot4 (weights 64, 16, 4, 1); the twelve bases a lane
                       needs are its own two packs and the previous lane's -- the neighbour's finished packs through one DPP move
                       (lane 0: the group's halo register), not the neighbour's bytes packed again.  Bits 24.. of W hold the
                       neighbour's first pack, which no window reads */
                    const u32 v0 = kmer_codes(sw2[0]), v1 = kmer_codes(sw2[1]);
                    W = kmer_stream(v0, v1, readlane_u32(packAll, g));
                    const u32 m0 = perm_lo(0x47435441u, v0), m1 = perm_lo(0x47435441u, v1);
                    u32 bad = (m0 ^ sw2[0]) | (m1 ^ sw2[1]);
                    opaque_u32(bad); /* (one compare for the ballot: the compiler would test the two halves apart and merge the masks) */
                    /* (a lane's halo is its neighbour's second dword, which that lane tests itself: only lane 0's is extra) */
                    const u32 inv_h0 = tile_start >= 4 ? readlane_u32(invAll, g) : 0xFu; /* wave-uniform */
                    if (!wave_ballot(bad != 0) && !(tile_start >= 4 && inv_h0 != 0)) {
                        okmask = have_halo ? 0xFFu : 0xF0u;
                        allok = tile_start >= 4; /* every window of every lane counts: the 5-mer updates add a constant */
                    } else {
                        const u32 i1 = invalid_nibble(m1, sw2[1]);
                        const u32 ih = wave_prev_u32(i1, inv_h0);
                        const u32 inv = lshl_or<8>(i1, lshl_or<4>(invalid_nibble(m0, sw2[0]), ih));
                        const u32 r = inv | (inv >> 1) | (inv >> 2) | (inv >> 3) | (inv >> 4);
                        okmask = ~r & 0xFFu;
                    }
                    /* (no mask of the row's bytes on top: a window is only looked at for a byte of the row) */
                }
#else
                const u32 up = FPL_OPT_DPPPREV ? wave_prev_u32(sw2[1], 0u) : shfl_up_u32(sw2[1], 1);
                const u32 halo = lane > 0 ? up : haloG[g];
                const u32 vh = kmer_codes(halo), v0 = kmer_codes(sw2[0]), v1 = kmer_codes(sw2[1]);
                W = perm_b32(kmer_pack(vh), perm_b32(kmer_pack(v0), kmer_pack(v1), 0x0c0c0703u), 0x0c070100u);
                {
                    const u32 m0 = perm_lo(0x47435441u, v0), m1 = perm_lo(0x47435441u, v1), mh = perm_lo(0x47435441u, vh);
                    u32 bad = (m0 ^ sw2[0]) | (m1 ^ sw2[1]);
                    if (have_halo) bad |= mh ^ halo;
                    if (!wave_ballot(nvalid > 0 && bad != 0)) {
                        okmask = have_halo ? 0xFFu : 0xF0u;
                        allok = tile_start >= 4; /* every window of every lane counts: the 5-mer updates add a constant */
                    } else {
                        const u32 ih = have_halo ? invalid_nibble(mh, halo) : 0xFu;
                        const u32 inv = lshl_or<8>(invalid_nibble(m1, sw2[1]), lshl_or<4>(invalid_nibble(m0, sw2[0]), ih));
                        const u32 r = inv | (inv >> 1) | (inv >> 2) | (inv >> 3) | (inv >> 4);
                        okmask = ~r & 0xFFu;
                    }
                    okmask &= (1u << nvalid) - 1u;
                }
#endif
                const int p0 = (int)c0;
#if FPL_OPT_KMER6
                if (!allok) { /* wave-uniform.  (A row of A, C, G, T holds no byte of the classes 0 and 2; elsewhere ask) */
                    const u32 z = ~((sw2[0] | (sw2[0] >> 2)) & (sw2[1] | (sw2[1] >> 2))) & 0x01010101u; /* a byte whose bits 0 and 2 are clear */
                    if (wave_ballot(z != 0)) {
                        odd_rows = (odd_rows << 8) | (u32)(bitG[g] + 1); /* wave-uniform: taken up behind the group */
                        continue;
                    }
                }
#endif
                /* one byte.  NPM: bit k of npmask says whether byte k lies behind the end of r1 (it then also goes to the
                   not-post table); KM: the byte's 5-mer window is counted 0 pre-filter only, 1 pre- and post-filter, 2 as
                   bit k of kbodymask says */
                /* (the increment of byte k + 1 is fetched before the updates of byte k are issued, as in k_stats.  Building it
                   on the vector unit instead -- bit 7 of q + 75 / q + 65 as the Q20 / Q30 tests, two v_perm per byte --
                   measured the same: this kernel issues 20 vector instructions per 64 bytes and the vector unit is what it
                   waits for, profiles/r02_ab) */
#define FPL_FS_Q(k) ((qw[(k) >> 2] >> (8 * ((k)&3))) & 0xFF)
                /* (the cell through one v_perm per byte on a per-dword high byte instead: four instructions fewer, 1.4 % slower, round 4) */
#define FPL_FS_CELL(k) mad_u24((sw2[(k) >> 2] >> (8 * ((k)&3))) & 7u, 8 * FS_BSTRIDE, lane8)
#if FPL_OPT_INCVALU
                /* the packed increment of a byte on the vector unit instead of out of the LDS table: the Q20 / Q30 bits of four
                   qualities at once (bit 7 of q + 75 / q + 65: qualities are < 128), moved to where two of the four need them
                   (bits 4 / 18 of the high word) */
#if FPL_OPT_INCPERM
                /* ... and the high word of byte k's increment as ONE v_perm: byte 0 from p20 (0x10 where q >= '5'), byte 2 from p30
                   (0x04 where q >= '?') */
                u32 p20[2], p30[2];
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    p20[j] = ((qw[j] + 0x4B4B4B4Bu) & 0x80808080u) >> 3;
                    p30[j] = ((qw[j] + 0x41414141u) & 0x80808080u) >> 5;
                }
#define FPL_FS_INC(k) (((u64)perm_b32(p30[(k) >> 2], p20[(k) >> 2], 0x0c000c00u | ((4u + ((k)&3)) << 16) | ((k)&3)) << 32) | (FPL_FS_Q(k) | (1u << 22)))
#else
                u32 dq[4];
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const u32 t20 = (qw[j] + 0x4B4B4B4Bu) & 0x80808080u, t30 = (qw[j] + 0x41414141u) & 0x80808080u;
                    dq[2 * j] = (t20 >> 3) | (t30 << 11);
                    dq[2 * j + 1] = (t20 >> 19) | (t30 >> 5);
                }
#define FPL_FS_INC(k) (((u64)((dq[(k) >> 1] >> (8 * ((k)&1))) & 0x40010u) << 32) | (FPL_FS_Q(k) | (1u << 22)))
#endif
                u64 inc_n = FPL_FS_INC(0);
#else
#define FPL_FS_INC(k) inc_of[FPL_FS_Q(k)]
                u64 inc_n = inc_of[FPL_FS_Q(0)];
#endif
#define FPL_FB_BYTE(k, NPM, KM, FULL)                                                                             \
    if (FULL || (k) < nv_row) {                                                                                   \
        u64* const cellp = (u64*)((char*)tbl + FPL_FS_CELL(k)); /* (byte offset of the (class, lane) cell) */      \
        atomicAdd(&cellp[(k)*64], inc);                                                                           \
        if (NPM && ((npmask >> (k)) & 1u)) atomicAdd(&cellp[(k)*64 + FS_T], inc);                                 \
        const u32 kidx = (W >> (2 * (7 - (k)))) & 0x3FFu;                                                         \
        const u32 kval = (FULL && allok) ? 1u : ((okmask >> (k)) & 1u);                                           \
        if (KM == 1)                                                                                              \
            atomicAdd(&kmer[1024u + kidx], kval);                                                                 \
        else if (KM == 0)                                                                                         \
            atomicAdd(&kmer[kidx], kval);                                                                         \
        else                                                                                                      \
            atomicAdd(&kmer[(((kbodymask >> (k)) & 1u) << 10) + kidx], kval);                                     \
    }
#define FPL_FB_ROW(NPM, KM, FULL)                                                                                 \
    {                                                                                                             \
        const int nv_row = FULL ? 8 : FPL_FS_NVALID;                                                              \
        (void)nv_row;                                                                                             \
        _Pragma("unroll") for (int k = 0; k < 8; k++) {                                                           \
            const u64 inc = inc_n;                                                                                \
            if (k < 7) inc_n = FPL_FS_INC(k + 1);                                                                 \
            FPL_FB_BYTE(k, NPM, KM, FULL)                                                                         \
        }                                                                                                         \
    }
                if (tp && (int)tile_start >= s + 4 && (int)(tile_start + FS_T) <= e) {
                    /* wave-uniform: the whole tile lies inside r1 (and inside the read) */
#if FPL_OPT_KMER6
                    /* the 5-mer windows two at a time: the pair that ends at byte k (odd) is the 6-mer at bit 2 (7 - k) of the stream;
                       it counts when both of its windows do, a window that counts alone goes to the 5-mer table */
#define FPL_FB_ROW6(ALLOK)                                                                                        \
    _Pragma("unroll") for (int k = 0; k < 8; k++) {                                                               \
        const u64 inc = inc_n;                                                                                    \
        if (k < 7) inc_n = FPL_FS_INC(k + 1);                                                                     \
        u64* const cellp = (u64*)((char*)tbl + FPL_FS_CELL(k));                                                   \
        atomicAdd(&cellp[(k)*64], inc);                                                                           \
        if (k & 1) atomicAdd(&k6[(W >> (2 * (7 - k))) & 0xFFFu], (ALLOK) ? 1u : ((ok2 >> (k - 1)) & 1u));         \
    }
                    if (allok) {
                        const u32 ok2 = 0;
                        (void)ok2;
                        FPL_FB_ROW6(true)
                    } else {
                        const u32 ok2 = okmask & (okmask >> 1);
                        FPL_FB_ROW6(false)
                        u32 alone = okmask & ~((ok2 & 0x55u) * 3u) & 0xFFu;
                        if (wave_ballot(alone != 0)) { /* wave-uniform; the lanes next to an N */
                            while (alone) {
                                const int k = __ffsll((unsigned long long)alone) - 1;
                                alone &= alone - 1;
                                atomicAdd(&kpost[(W >> (2 * (7 - k))) & 0x3FFu], 1u);
                            }
                        }
                    }
#undef FPL_FB_ROW6
#else
                    const u32 npmask = 0, kbodymask = 0xFFu;
                    (void)npmask;
                    (void)kbodymask;
                    if (allok) {
                        FPL_FB_ROW(false, 1, true)
                    } else {
                        FPL_FB_ROW(false, 1, true)
                    }
#endif
                } else if (tp) { /* a tile that holds an end of r1 */
                    const u32 npmask = ~range_mask8(-1, e - p0) & 0xFFu, kbodymask = range_mask8(s + 4 - p0, e - p0);
                    FPL_FB_ROW(true, 2, false)
                } else { /* not counted post-filter */
                    const u32 npmask = 0, kbodymask = 0;
                    (void)npmask;
                    (void)kbodymask;
                    if (itemL >= tile_start + FS_T) {
                        FPL_FB_ROW(false, 0, true)
                    } else {
                        FPL_FB_ROW(false, 0, false)
                    }
                }
#undef FPL_FB_ROW
#undef FPL_FB_BYTE
#undef FPL_FS_Q
#undef FPL_FS_CELL
#undef FPL_FS_INC
#undef FPL_FS_NVALID
            }
#if FPL_OPT_KMER6
            /* The rows that hold a byte of class 0 or 2 (no letter: the cells of those classes gave their LDS to the 5-mer tables):
               read again and walked byte by byte, every test spelled out -- cells in LDS for the other classes, global atomics on the
               counters for these two; the 5-mer windows one by one.  Right for any row (inside r1, across an end of r1, not counted
               post-filter; whole or ragged); taken by none that holds DNA. */
            while (odd_rows) { /* wave-uniform */
                const int bit = (int)(odd_rows & 0xFFu) - 1;
                odd_rows >>= 8;
                const u32 itemL = readlane_u32(L, bit);
                const int e = (int)readlane_u32(E, bit);
                const u8* const rs = seq + readlane_u64(st, bit);
                const u8* const rq = qual + readlane_u64(st, bit);
                /* (this path must cost the rows of DNA nothing: its addresses are worked out here, from values the optimiser cannot
                   trace to the loop around it -- hoisted, they took registers away from every row) */
                u32 c0s = c0, l8s = lane8;
                opaque_u32(c0s);
                opaque_u32(l8s);
                if (itemL <= c0s) continue; /* (per lane from here on: no wave-wide operation below) */
                const int nv = (int)min(8u, itemL - c0s);
                const u32x2 sv = load8_guard(rs + c0s, seq_end), qv = load8_guard(rq + c0s, qual_end);
                const u32 hv = c0s >= 4 ? load4_guard(rs + c0s - 4, seq_end) : 0u;
                const u32 vh = kmer_codes(hv), v0 = kmer_codes(sv.x), v1 = kmer_codes(sv.y);
                const u32 Wr = (kmer_pack_dot(vh) << 16) | (kmer_pack_dot(v0) << 8) | kmer_pack_dot(v1);
                const u32 ih = c0s >= 4 ? invalid_nibble(perm_lo(0x47435441u, vh), hv) : 0xFu;
                const u32 inv = (invalid_nibble(perm_lo(0x47435441u, v1), sv.y) << 8) | (invalid_nibble(perm_lo(0x47435441u, v0), sv.x) << 4) | ih;
                const u32 okm = ~(inv | (inv >> 1) | (inv >> 2) | (inv >> 3) | (inv >> 4)) & 0xFFu;
                long long* const pre = counters + FPL_OFF_PRE(C);
                long long* const post = counters + FPL_OFF_POST(C);
#pragma unroll 1
                for (int k = 0; k < nv; k++) {
                    const u32 sh = 8u * (u32)(k & 3);
                    const u32 b = ((k < 4 ? sv.x : sv.y) >> sh) & 0xFFu, q = ((k < 4 ? qv.x : qv.y) >> sh) & 0xFFu;
                    const u32 cls = b & 7u;
                    const int p = (int)c0s + k;
                    if (cls == 0u || cls == 2u) {
                        if ((u32)p < C) {
                            atomicAdd((u64*)&pre[FPL_ST_CYC(p, 0, cls)], 1ull);
                            atomicAdd((u64*)&pre[FPL_ST_CYC(p, 1, cls)], (u64)(long long)((int)q - 33));
                            if (q >= '5') atomicAdd((u64*)&pre[FPL_ST_CYC(p, 2, cls)], 1ull);
                            if (q >= '?') atomicAdd((u64*)&pre[FPL_ST_CYC(p, 3, cls)], 1ull);
                            if (tp && p >= s && p < e) {
                                const int pc = p - s;
                                atomicAdd((u64*)&post[FPL_ST_CYC(pc, 0, cls)], 1ull);
                                atomicAdd((u64*)&post[FPL_ST_CYC(pc, 1, cls)], (u64)(long long)((int)q - 33));
                                if (q >= '5') atomicAdd((u64*)&post[FPL_ST_CYC(pc, 2, cls)], 1ull);
                                if (q >= '?') atomicAdd((u64*)&post[FPL_ST_CYC(pc, 3, cls)], 1ull);
                            }
                        }
                    } else {
                        const u64 inc = (u64)q | (1ull << 22) | ((u64)(q >= '5') << 36) | ((u64)(q >= '?') << 50);
                        u64* const cellp = (u64*)((char*)tbl + cls * (8u * FS_BSTRIDE) + l8s);
                        atomicAdd(&cellp[k * 64], inc);
                        if (tp && p >= e) atomicAdd(&cellp[k * 64 + FS_T], inc); /* behind the end of r1: not counted post-filter */
                    }
                    if ((okm >> k) & 1u) {
                        const bool body = tp && p - 4 >= s && p < e; /* the window lies inside r1 */
                        atomicAdd(&kmer[(body ? 1024u : 0u) + ((Wr >> (2 * (7 - k))) & 0x3FFu)], 1u);
                    }
                }
            }
#endif
        }
    }
    } /* slices of the item */
    if (open) hand_over(leader);
    }
#if FPL_OPT_KMER6 && FPL_OPT_KMERKEEP
    __syncthreads(); /* (the last item's rows are in) */
    kmer_flush();
#endif
}

/* Sum the slabs of one k_stats_sorted launch into the per-cycle counters (grid as k_stats_reduce: x = chunk of 256 cells,
 * y = tile).  A post cycle c of tile t takes, from every slice with a front trim s, cycle c + s of the slice's slab of tile
 * t or t + 1: pre minus not-post. */
__global__ void __launch_bounds__(256)
k_stats_reduce_sorted(const u64* __restrict__ scratch, const u8* __restrict__ flags, const u32* __restrict__ sw,
                      u32 max_slices, u32 n_tiles, long long* __restrict__ counters, u32 C) {
    /* the slices that handed over a slab for this tile or the next one, gathered once per block (256 at a time): a thread
       that asks flag byte after flag byte whether a slab exists, and only then for its cell, spends its time waiting for
       one load after the other; from the list its loads go out four at a time */
    __shared__ u32 a_sl[256], a_sp1[256], a_n;
    __shared__ u8 a_fh[256], a_fn[256];
    const u32 tile = blockIdx.y;
    const u32 cell = blockIdx.x * 256 + threadIdx.x;
    const bool is_post = cell >= 8 * FS_T;
    const u8* slab_flags = flags + n_tiles;
    const bool here = flags[tile] != 0, next = tile + 1 < n_tiles && flags[tile + 1] != 0;
    if (!here && !next) return; /* block-uniform */
    const u32 n_slices = sw[SW_NSLICES];
    const u32 cc = is_post ? cell - 8 * FS_T : cell;
    /* threads in SLOT order: thread j of a class row owns the cycle whose slab slot is j (slot(x) = (x % 8) * 64 + x / 8), so a
       wave's 64 loads of a slab are 512 consecutive bytes -- with threads in cycle order they were eight pieces of 64 bytes; a
       front trim moves every thread's slot by the same amount, so the shifted post reads stay consecutive too */
    const u32 cls = cc / FS_T, tj = cc % FS_T;
    const u32 x = (tj & 63u) * 8u + (tj >> 6);
    const u32 slot_pre = tj;
    u64 qsum = 0, cnt = 0, q20 = 0, q30 = 0;
    u64 nsum = 0, ncnt = 0, n20 = 0, n30 = 0; /* post: what lies behind the reads' ends */
    for (u32 base = 0; base < n_slices; base += 256) { /* block-uniform */
        __syncthreads();
        if (threadIdx.x == 0) a_n = 0;
        __syncthreads();
        const u32 sl = base + threadIdx.x;
        if (sl < n_slices) {
            const u8 fh = here ? slab_flags[(size_t)tile * max_slices + sl] : (u8)0;
            const u8 fn = next ? slab_flags[(size_t)(tile + 1) * max_slices + sl] : (u8)0;
            if (fh | fn) {
                const u32 i = atomicAdd(&a_n, 1u);
                a_sl[i] = sl;
                a_sp1[i] = sw[SW_SLICES + 4 * sl + 2];
                a_fh[i] = fh;
                a_fn[i] = fn;
            }
        }
        __syncthreads();
        const u32 na = a_n;
        for (u32 k0 = 0; k0 < na; k0 += FPL_RED_UNROLL) {
            u64 v[FPL_RED_UNROLL] = {}, nv[FPL_RED_UNROLL] = {};
#pragma unroll
            for (u32 u = 0; u < FPL_RED_UNROLL; u++) {
                const u32 k = k0 + u;
                if (k >= na) continue;
                const u32 s_l = a_sl[k];
                if (!is_post) { /* (block-uniform) */
                    if ((a_fh[k] >> cls) & 1u) v[u] = scratch[((size_t)tile * max_slices + s_l) * FS_SLAB + cls * FS_T + slot_pre];
                } else {
                    /* post cycle x of this tile takes, from a slice with front trim s, cycle x + s of its slab of this tile or
                       the next: pre minus not-post */
                    const u32 sp1 = a_sp1[k];
                    if (!sp1) continue;
                    const u32 xs = x + (sp1 - 1);
                    const bool nx = xs >= (u32)FS_T;
                    if (!(((nx ? a_fn[k] : a_fh[k]) >> cls) & 1u)) continue;
                    const u32 xx = xs & (u32)(FS_T - 1);
                    const u32 slot = (xx & 7) * 64 + (xx >> 3);
                    const u64* sb = scratch + ((size_t)(tile + (nx ? 1u : 0u)) * max_slices + s_l) * FS_SLAB;
                    v[u] = sb[cls * FS_T + slot];
                    nv[u] = sb[8 * FS_T + cls * FS_T + slot];
                }
            }
#pragma unroll
            for (u32 u = 0; u < FPL_RED_UNROLL; u++) {
                fs_unpack_add(v[u], qsum, cnt, q20, q30);
                fs_unpack_add(nv[u], nsum, ncnt, n20, n30);
            }
        }
    }
    if (!is_post && !here) return;
    qsum -= nsum;
    cnt -= ncnt;
    q20 -= n20;
    q30 -= n30;
    const u32 c = tile * FS_T + x;
    if (cnt && c < C) {
        long long* st = counters + (is_post ? FPL_OFF_POST(C) : FPL_OFF_PRE(C));
        if (is_post) {
            /* (the post-only pass may run beside this kernel on the side stream, and a block of it whose packed fields fill up
               empties its table into these very counters with atomics: so does this kernel -- one owner per cell here, the
               atomic costs no more than the store) */
            atomicAdd((u64*)&st[FPL_ST_CYC(c, 0, cls)], cnt);
            atomicAdd((u64*)&st[FPL_ST_CYC(c, 1, cls)], qsum - 33ull * cnt);
            atomicAdd((u64*)&st[FPL_ST_CYC(c, 2, cls)], q20);
            atomicAdd((u64*)&st[FPL_ST_CYC(c, 3, cls)], q30);
        } else {
            st[FPL_ST_CYC(c, 0, cls)] += (long long)cnt;
            st[FPL_ST_CYC(c, 1, cls)] += (long long)qsum - 33ll * (long long)cnt;
            st[FPL_ST_CYC(c, 2, cls)] += (long long)q20;
            st[FPL_ST_CYC(c, 3, cls)] += (long long)q30;
        }
    }
}

/* =========================================================================================
 * k_count_end_kmers: the counting loops of the adapter auto-detection (Evaluator::evalAdapterAndReadNum,
 * src/evaluator.cpp:300-345).  For every read of the evaluation prefix (<= 64 Ki reads) the 10-mers that start at the first
 * 128 positions (side 0) or at the last 129 positions in front of the skipped tail (side 1) are counted: counts[key]++,
 * position_acc[key] += pos (side 0) / rlen - pos (side 1), total++ -- a key is valid when its ten bases are A, T/U, C, G
 * (the reference rolls the key and starts over behind an invalid base: the same set of keys).  One wave per read,
 * lanes = positions; the 4^10 counters live in HBM (the host keeps getTopKey / extendKeyToAdapter, which walk them).
 * ======================================================================================= */
__global__ void __launch_bounds__(256)
k_count_end_kmers(const u8* __restrict__ seq, const uint64_t* __restrict__ off, u32 n_reads, int side, int shift_tail,
                  u32* __restrict__ counts, unsigned long long* __restrict__ position_acc, unsigned long long* __restrict__ total) {
    const int lane = lane_id();
    u32 mine = 0;
    for (u32 ri = blockIdx.x * (blockDim.x / 64) + wave_in_block(); ri < n_reads; ri += gridDim.x * (blockDim.x / 64)) {
        const uint64_t o = off[ri];
        const long long rlen = (long long)(off[ri + 1] - o);
        long long first, end;
        if (!pick::key_window(rlen, side, shift_tail, first, end)) continue;
        const u8* data = seq + o;
        for (long long p0 = first; p0 <= end; p0 += 64) {
            const long long pos = p0 + lane;
            if (pos > end) continue;
            u32 key;
            if (!pick::key_at(data + pos, key)) continue;
            atomicAdd(&counts[key], 1u);
            atomicAdd(&position_acc[key], (unsigned long long)(side == 0 ? pos : rlen - pos));
            mine++;
        }
    }
    const u32 t = wave_sum_u32(mine);
    if (lane == 0 && t) atomicAdd(total, (unsigned long long)t);
}

/* =========================================================================================
 * k_pick_adapter: what the detection does with the counters (adapter_pick.h), on the device, so that the 12 MB of tables
 * stay where they were counted: ONE block; every thread walks a stride of the 2^20 keys for the masked arg-max (the seed:
 * the admissible key with the largest count, the smallest key among equals) and the number of keys seen at all; thread 0
 * then grows the seed in both directions (a chain of <= 54 dependent look-ups of four counters each).
 * ======================================================================================= */
__global__ void __launch_bounds__(1024)
k_pick_adapter(const u32* __restrict__ counts, const unsigned long long* __restrict__ position_acc, int is_rna,
               pick::Pick* __restrict__ out) {
    __shared__ u64 best[1024];
    __shared__ u32 seen[1024];
    u64 mine = 0;
    u32 nseen = 0;
    for (u32 k = threadIdx.x; k < pick::NKEYS; k += blockDim.x) {
        const u32 val = counts[k];
        nseen += val > 0 ? 1u : 0u;
        if (val > 0 && pick::key_admissible(k) && pick::count_digits_vary(val)) {
            const u64 r = pick::seed_rank(val, k);
            mine = r > mine ? r : mine;
        }
    }
    best[threadIdx.x] = mine;
    seen[threadIdx.x] = nseen;
    __syncthreads();
    for (u32 d = blockDim.x / 2; d > 0; d >>= 1) {
        if (threadIdx.x < d) {
            if (best[threadIdx.x + d] > best[threadIdx.x]) best[threadIdx.x] = best[threadIdx.x + d];
            seen[threadIdx.x] += seen[threadIdx.x + d];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        pick::Pick p;
        p.key = best[0] ? (int32_t)~(u32)best[0] : -1;
        p.count = (u32)(best[0] >> 32);
        p.total_key = seen[0];
        p.len = 0;
        p.seq[0] = 0;
        if (p.key >= 0)
            pick::grow(p, is_rna != 0, [&](u32 k) { return k ? counts[k] : 0u; }, [&](u32 k) { return (u64)position_acc[k]; });
        *out = p;
    }
}

/* =========================================================================================
 * k_break_mask: --break and --mask, src/seprocessor.cpp:234-281.
 *
 * Runs only when one of the two options is on (DevConfig::defer); k_scan then stops after the
 * middle-adapter split and leaves <= 2 preliminary fragments per read in the result record.  One
 * wave per read walks them:
 *   --break: Filter::detectLowQualityRegions (src/filter.cpp:83-128) on the fragment, then
 *            Read::breakByRegions (src/read.cpp:227-262): the stretches between regions become the
 *            output reads, numbered "r<i>-";
 *   --mask : detectLowQualityRegions on every output read, Read::maskRegionWithN over each region;
 *   then Filter::passFilter on the (masked) output read, the FilterResult / post-Stats scalars, a
 *   fpl_fragment record, its fpl_region list, and -- when it passes -- its pieces (unmasked / masked,
 *   each with the cycle it starts at) on the post-only list that k_stats<EXTRA> counts.
 *
 * The reference's region detector rolls a window sum along the read with two sequential scans per
 * region ("first window below the threshold", then "first window back above it").  Its rolling sum
 * is, at every step, a window sum W(s) = q[s] + .. + q[s+w-1] plus a constant that only depends on
 * where the scan started (the warm-up loop's absolute bound makes the first scan start from
 * W(0) - q[w-1] and all later ones from 0), so both scans are "first s with W(s) + bias </>= thr":
 * 64 candidate positions per round from one prefix scan of q[s+w] - q[s], ballot + ffs.
 * ======================================================================================= */
struct BmLists {
    fpl_fragment* frags;
    fpl_region* regs;
    u32 frag_cap, reg_cap, item_cap;
    u32* counts; /* [0] fragments, [1] regions, [2] set when a list was too small */
};
struct BmWaveLds {
    u32 hist[129 * HIST_COPIES];
};

__device__ __forceinline__ int bm_window_sum(const u8* __restrict__ q, int s, int w) {
    u32 acc = 0;
    for (int j = lane_id(); j < w; j += 64) acc += q[s + j];
    return (int)wave_sum_u32(acc);
}
/* first s in [s0, s1) with W(s) + bias < thr (LT) or >= thr, -1 when there is none; s1 + w <= L + 1 */
template <bool LT>
__device__ __forceinline__ int bm_search(const u8* __restrict__ q, int L, int w, int s0, int s1, int bias, int thr) {
    if (s0 >= s1) return -1;
    const int lane = lane_id();
    int wbase = bm_window_sum(q, s0, w); /* W(b) of the round's first position */
    for (int b = s0; b < s1; b += 64) {
        const int s = b + lane;
        u32 d = 0; /* W(s + 1) - W(s) */
        if (s + w < L) d = (u32)q[s + w] - (u32)q[s];
        const u32 incl = wave_scan_incl_u32(d);
        const int ws = wbase + (int)(incl - d);
        const bool hit = s < s1 && (LT ? (ws + bias < thr) : (ws + bias >= thr));
        const u64 m = wave_ballot(hit);
        if (m) return b + (int)__ffsll((long long)m) - 1;
        wbase += (int)readlane_u32(incl, 63);
    }
    return -1;
}
struct BmGen {
    int start;
    bool first;
};
/* the next region (first, last inclusive) detectLowQualityRegions reports on q[0, L) */
__device__ __forceinline__ bool bm_next_region(const u8* __restrict__ q, int L, int w, int thr, BmGen& g, int& r_first,
                                               int& r_last) {
    if (w <= 0 || g.start + w > L) return false;
    /* the rolling sum at scan position s is W(s) + bias */
    const int bias = g.first ? -(int)q[w - 1] : -bm_window_sum(q, g.start, w);
    const int ws = bm_search<true>(q, L, w, g.start, L - w, bias, thr);
    if (ws < 0) return false;
    const int hit = bm_search<false>(q, L, w, ws + 1, L - w + 1, bias, thr); /* the update at e looks at W(e + 1) */
    const int e = hit < 0 ? L - w : hit - 1;
    r_first = ws;
    r_last = e + w - 1;
    g.start = e + w;
    g.first = false;
    return true;
}

/* passFilter sums and quality histogram of bytes [a, b) of the read, masked (every base reads as N) or not;
   prev = the effective base in front of a, 256 when a starts the output read.  Per-lane partial sums. */
__device__ __forceinline__ void bm_accumulate(const u8* __restrict__ rb, const u8* __restrict__ qb, int a, int b, bool masked,
                                              u32 prev, int qualified_qual, u32* __restrict__ h, RangeSums& part) {
    const int lane = lane_id();
    for (int j = a + lane; j < b; j += 64) {
        const u32 q = qb[j];
        const u32 cur = masked ? (u32)'N' : (u32)rb[j];
        const u32 pv = j == a ? prev : (masked ? (u32)'N' : (u32)rb[j - 1]);
        atomicAdd(&h[(q & 127u) * HIST_COPIES + (lane & (HIST_COPIES - 1))], 1u);
        part.lowq += ((int)q < qualified_qual);
        part.totq += q;
        part.nn += (cur == 'N');
        part.diff += (pv != 256u && cur != pv);
    }
}

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
k_break_mask(const u8* __restrict__ seq, const u8* __restrict__ qual, const uint64_t* __restrict__ off, u32 n_reads,
             const DevConfig* __restrict__ cfg, fpl_read_result* __restrict__ results, BmLists lists,
             uint64_t* __restrict__ item_off, u32* __restrict__ item_len, u32* __restrict__ item_cyc,
             u32* __restrict__ item_count, long long* __restrict__ counters, u32 C) {
    __shared__ BmWaveLds wlds[WAVES];
    __shared__ ScanBlockAcc acc;
    const int lane = lane_id();
    u32* const h = wlds[wave_in_block()].hist;
    hist_zero(h);
    {
        u64* z = (u64*)&acc;
        for (u32 i = threadIdx.x; i < sizeof(ScanBlockAcc) / 8; i += blockDim.x) z[i] = 0;
    }
    __syncthreads();
    const int qq = cfg->qualified_qual;
    const u32 wave_global = blockIdx.x * WAVES + wave_in_block(), n_waves = gridDim.x * WAVES;
    for (u32 ri = wave_global; ri < n_reads; ri += n_waves) {
        const uint64_t o0 = off[ri];
        const u8* rb = seq + o0;
        const u8* qb = qual + o0;
        const fpl_read_result pre = results[ri];
        const int npre = pre.dropped ? 0 : (int)pre.n_frag;
        u32 n_out = 0; /* output reads of this read so far (= seq_no of the next one) */

        /* one output read [os, os + ol) of the original read: mask, passFilter, counters, records */
        auto emit = [&](int os, int ol, u32 kind, u32 break_no) {
            const u32 seq_no = n_out++;
            RangeSums part = {0, 0, 0, 0};
            u32 n_reg = 0, n_piece = 0;
            /* pass 1: walk the pieces (unmasked stretch, masked region, ...) for the sums */
            {
                BmGen g = {0, true};
                int pos = 0, rf, rl;
                u32 prev = 256u;
                while (cfg->msk && bm_next_region(qb + os, ol, cfg->msk_w, cfg->msk_thr, g, rf, rl)) {
                    int st = rf, ln = rl - rf + 1; /* maskRegionWithN(first, last - first + 1), src/read.cpp:217-225 */
                    if (st < 0 || ln <= 0 || st >= ol) continue;
                    if (st + ln > ol) ln = ol - st;
                    if (st > pos) {
                        bm_accumulate(rb, qb, os + pos, os + st, false, prev, qq, h, part);
                        prev = rb[os + st - 1];
                        n_piece++;
                    }
                    bm_accumulate(rb, qb, os + st, os + st + ln, true, prev, qq, h, part);
                    prev = 'N';
                    n_piece++;
                    n_reg++;
                    pos = st + ln;
                }
                if (pos < ol) {
                    bm_accumulate(rb, qb, os + pos, os + ol, false, prev, qq, h, part);
                    n_piece++;
                }
            }
            RangeSums sm;
            sm.lowq = wave_sum_u32(part.lowq);
            sm.nn = wave_sum_u32(part.nn);
            sm.totq = wave_sum_u32(part.totq);
            sm.diff = wave_sum_u32(part.diff);
            u32 t0, t1;
            hist_totals(h, t0, t1);
            const int code = filter_code(cfg, ol, sm);
            const bool pass = code == FPL_PASS_FILTER;
            int med = 0;
            if (pass) med = hist_median(t0, t1, (u32)ol);
            if (lane == 0) atomicAdd(&acc.fr[code], (u64)1);
            if (pass) {
                if (t0) atomicAdd(&acc.bqh[1][2 * lane], (u64)t0);
                if (t1) atomicAdd(&acc.bqh[1][2 * lane + 1], (u64)t1);
                if (lane == 0) {
                    atomicAdd(&acc.medh[1][med], (u64)1);
                    atomicAdd(&acc.medb[1][med], (u64)ol);
                    atomicAdd(&acc.reads[1], (u64)1);
                    atomicAdd(&acc.lensum[1], (u64)ol);
                }
            }
            /* reserve the record, its regions and (when it passes) its pieces */
            u32 fi = 0, rbase = 0, ibase = 0;
            if (lane == 0) {
                fi = atomicAdd(&lists.counts[0], 1u);
                if (n_reg) rbase = atomicAdd(&lists.counts[1], n_reg);
                if (pass && n_piece) ibase = atomicAdd(item_count, n_piece);
            }
            fi = readlane_u32(fi, 0);
            rbase = readlane_u32(rbase, 0);
            ibase = readlane_u32(ibase, 0);
            const bool fits = fi < lists.frag_cap && rbase + n_reg <= lists.reg_cap && (!pass || ibase + n_piece <= lists.item_cap);
            if (!fits) {
                if (lane == 0) lists.counts[2] = 1u;
                return;
            }
            if (lane == 0) {
                fpl_fragment f;
                f.read = ri;
                f.seq_no = seq_no;
                f.start = (u32)os;
                f.len = (u32)ol;
                f.region_first = rbase;
                f.region_count = n_reg;
                f.break_no = (uint16_t)break_no;
                f.code = (u8)code;
                f.kind = (u8)kind;
                f.median_q = (u8)med;
                f.reserved[0] = f.reserved[1] = f.reserved[2] = 0;
                lists.frags[fi] = f;
            }
            if (n_reg == 0 && !pass) return;
            /* pass 2: the same walk writes the regions and the pieces */
            {
                BmGen g = {0, true};
                int pos = 0, rf, rl;
                u32 ir = rbase, ii = ibase;
                auto piece = [&](int a, int b, bool masked) {
                    if (pass && lane == 0) {
                        item_off[ii] = o0 + (uint64_t)(os + a);
                        item_len[ii] = (u32)(b - a);
                        item_cyc[ii] = (u32)a | (masked ? 0x80000000u : 0u);
                    }
                    ii++;
                };
                while (cfg->msk && bm_next_region(qb + os, ol, cfg->msk_w, cfg->msk_thr, g, rf, rl)) {
                    int st = rf, ln = rl - rf + 1;
                    if (st < 0 || ln <= 0 || st >= ol) continue;
                    if (st + ln > ol) ln = ol - st;
                    if (st > pos) piece(pos, st, false);
                    piece(st, st + ln, true);
                    if (lane == 0) {
                        lists.regs[ir].start = (u32)(os + st);
                        lists.regs[ir].len = (u32)ln;
                    }
                    ir++;
                    pos = st + ln;
                }
                if (pos < ol) piece(pos, ol, false);
            }
        };

        for (int f = 0; f < npre; f++) {
            const int fs = (int)pre.frag_start[f], fl = (int)pre.frag_len[f];
            const u32 kind = pre.kind[f];
            bool broken = false;
            if (cfg->brk) { /* :235-252 */
                BmGen g = {0, true};
                int rf, rl, last_end = -1, j = 0;
                while (bm_next_region(qb + fs, fl, cfg->brk_w, cfg->brk_thr, g, rf, rl)) {
                    broken = true;
                    j++; /* Read::breakByRegions, src/read.cpp:230-250: i + 1 */
                    int st = rf, en = rl;
                    if (st < 0) st = 0;
                    if (en >= fl) en = fl - 1;
                    if (st > en || st >= fl) continue;
                    if (st > last_end + 1) emit(fs + last_end + 1, st - last_end - 1, kind, (u32)j);
                    last_end = en;
                }
                if (broken && last_end < fl - 1) emit(fs + last_end + 1, fl - last_end - 1, kind, (u32)(j + 1));
            }
            if (!broken) emit(fs, fl, kind, 0u);
        }
        if (lane == 0) { /* the per-read record keeps r1 / dropped / median_q_pre; fragments are in the list */
            fpl_read_result r = pre;
            r.n_frag = (u8)(n_out > 255u ? 255u : n_out);
            r.frag_start[0] = r.frag_start[1] = r.frag_len[0] = r.frag_len[1] = 0;
            r.code[0] = r.code[1] = r.kind[0] = r.kind[1] = r.median_q_post[0] = r.median_q_post[1] = 0;
            results[ri] = r;
        }
    }
    __syncthreads();
    scan_acc_flush(acc, counters, C);
}

}  // namespace fpl
#endif
