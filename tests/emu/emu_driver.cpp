/*
 * emu_driver.cpp -- TEST INFRASTRUCTURE ONLY.
 * Compiles the product kernels (fastplong_amd/csrc/kernels.h + pipeline.h) for the host on top
 * of hip_emu.h and runs one batch through the same launch sequence the HIP library uses.
 * Built as tests/emu/libfpl_emu.so by tests/emu/build.py; loaded only by tests.
 */
#define FPL_EMU 1
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../fastplong_amd/csrc/pipeline.h"

using namespace fpl;

/* fragments / regions of the last emu_process_batch call with --break / --mask (sorted like fpl_get_fragments) */
static std::vector<fpl_fragment> g_frags;
static std::vector<fpl_region> g_regs;
extern "C" uint32_t emu_fragment_count() { return (uint32_t)g_frags.size(); }
extern "C" uint32_t emu_region_count() { return (uint32_t)g_regs.size(); }
extern "C" void emu_get_fragments(fpl_fragment* f, fpl_region* r) {
    if (!g_frags.empty()) memcpy(f, g_frags.data(), g_frags.size() * sizeof(fpl_fragment));
    if (!g_regs.empty()) memcpy(r, g_regs.data(), g_regs.size() * sizeof(fpl_region));
}

extern "C" int emu_process_batch(const fpl_options* opt, const char* start, int start_len, const char* end,
                                 int end_len, const fpl_adapter* fasta, int n_fasta, const uint8_t* seq,
                                 const uint8_t* qual, const uint64_t* off, uint32_t n_reads, int64_t* counters,
                                 uint32_t C, fpl_read_result* results, uint32_t n_cu) {
    DevConfig cfg;
    build_config(&cfg, opt, start_len, end_len, n_fasta);
    std::vector<DevAdapter> ads(2 + n_fasta);
    build_adapter(&ads[0], start, start_len);
    build_adapter(&ads[1], end, end_len);
    for (int i = 0; i < n_fasta; i++) build_adapter(&ads[2 + i], fasta[i].seq, fasta[i].len);
    cfg.ham_fast = ads[0].acgt_only && ads[1].acgt_only;
    {
        std::vector<int> lens(2 + n_fasta), acgt(2 + n_fasta);
        for (int i = 0; i < 2 + n_fasta; i++) lens[i] = ads[i].len, acgt[i] = ads[i].acgt_only;
        cfg.trim_mode = trim_mode_of(lens.data(), acgt.data(), 2 + n_fasta);
    }
    if (getenv("FPL_EMU_NO_HAM_FAST")) cfg.ham_fast = 0;
    cfg.scan_short = cfg.adapter_enabled && cfg.ham_fast && ads[0].len <= 32 && ads[1].len <= 32;

    uint64_t n_bytes = n_reads ? off[n_reads] : 0;
    uint32_t max_len = 0;
    for (uint32_t i = 0; i < n_reads; i++) {
        uint32_t l = (uint32_t)(off[i + 1] - off[i]);
        if (l > max_len) max_len = l;
    }
    if (max_len > C) return FPL_ERR_CAPACITY;
    std::vector<ReadState> state(n_reads ? n_reads : 1);
    u32 frag_cap = 0, reg_cap = 0, item_cap = 2 * n_reads + 2;
    if (cfg.defer) break_mask_caps(n_reads, n_bytes, cfg.brk, cfg.brk_w, cfg.msk, cfg.msk_w, frag_cap, reg_cap, item_cap);
    std::vector<uint64_t> frag_off((size_t)item_cap + 2, 0);
    std::vector<uint32_t> frag_len((size_t)item_cap + 2, 0), frag_cyc((size_t)item_cap + 2, 0);
    std::vector<fpl_fragment> frags((size_t)frag_cap + 1);
    std::vector<fpl_region> regs((size_t)reg_cap + 1);
    u32 bm_counts[4] = {0, 0, 0, 0};
    uint32_t work_ctr[WORK_CTR_WORDS] = {0};
    /* the kernels never read past n_bytes, but give the buffers an end guard anyway */
    BatchArgs a;
    a.seq = seq;
    a.qual = qual;
    a.off = off;
    a.n_reads = n_reads;
    a.n_bytes = n_bytes;
    a.max_read_len = max_len;
    a.cfg = &cfg;
    a.ads = ads.data();
    a.state = state.data();
    a.results = results;
    a.frag_off = frag_off.data();
    a.frag_len = frag_len.data();
    a.frag_cyc = frag_cyc.data();
    a.bm = BmLists{frags.data(), regs.data(), frag_cap, reg_cap, item_cap, bm_counts};
    a.defer = cfg.defer != 0;
    a.trim_mode = cfg.trim_mode;
    a.n_fasta = cfg.n_fasta;
    a.scan_short = cfg.scan_short != 0;
    a.counters = (long long*)counters;
    a.C = C;
    a.work_ctr = work_ctr;
    std::vector<ScanRec> recs(n_reads ? n_reads : 1);
    std::vector<RedoItem> redo(n_reads ? n_reads : 1);
    a.recs = recs.data();
    std::vector<ScanWin> wins(n_reads ? n_reads : 1);
    a.wins = wins.data();
    a.redo = redo.data();
    a.n_cu = n_cu ? n_cu : 2;
    a.tune = stats_tune_from_env(); /* (the tests set the hooks per case) */
    const size_t slabs = stats_scratch_slabs(n_reads, n_bytes, max_len, a.n_cu, a.tune);
    std::vector<u64> scratch(slabs * (size_t)FS_SLAB + 1);
    std::vector<u8> sflags(slabs + slabs / 8 + 4096);
    a.stats_scratch = scratch.data();
    const u32 per_s = stats_items_per_slice(n_reads, n_reads ? (u32)(n_bytes / n_reads) : 0, a.n_cu, a.tune);
    std::vector<u32> sort_ws(sort_ws_words(stats_sorted_max_slices(n_reads, per_s, a.tune), n_reads) + 1, 0xA5A5A5A5u);
    std::vector<uint64_t> st_off(n_reads + 1);
    std::vector<u32> st_len(n_reads + 1), st_e(n_reads + 1);
    a.sort_ws = sort_ws.data();
    a.st_off = st_off.data();
    a.st_len = st_len.data();
    a.st_e = st_e.data();
    a.stats_flags = sflags.data();
    enqueue_batch(a, nullptr, [](int) {});
#ifdef FPL_EMU_FILTER_STATS
    fprintf(stderr, "emu: filter refreshes %llu, flagged start %llu, end %llu, whole-adapter flags %llu, exact trims run %llu, of which moved r1 %llu (reads %u)\n",
            fpl::g_filter_stats[0], fpl::g_filter_stats[1], fpl::g_filter_stats[2], fpl::g_filter_stats[3], fpl::g_filter_stats[4],
            fpl::g_filter_stats[5], n_reads);
#endif
#ifdef FPL_EMU_PAIR_STATS
    fprintf(stderr, "emu: k_scan pair packing: %llu last tiles hosted a head, %llu head bytes (reads %u)\n", fpl::g_pair_stats[0],
            fpl::g_pair_stats[1], n_reads);
#endif
#ifdef FPL_EMU_TRIM_STATS
    fprintf(stderr, "emu: k_trim_ends_batched groups %llu, P1b lanes %llu, P2 %llu, P3 wants %llu may %llu, P6 %llu, P7 wants %llu may %llu, lane searches %llu rounds %llu (all x 64 lanes; reads %u)\n",
            fpl::g_trim_stats[0], fpl::g_trim_stats[1], fpl::g_trim_stats[2], fpl::g_trim_stats[3], fpl::g_trim_stats[4], fpl::g_trim_stats[5],
            fpl::g_trim_stats[6], fpl::g_trim_stats[7], fpl::g_trim_stats[8], fpl::g_trim_stats[9], n_reads);
#endif
    if (getenv("FPL_EMU_DEBUG_PLAN")) { /* how the reads were planned for the statistics passes */
        u32 tp = 0;
        for (uint32_t i = 0; i < n_reads; i++) tp += state[i].pad & 1u;
        fprintf(stderr, "emu: %u reads, %u planned to-post, EXTRA list %u, sort slices %u\n", n_reads, tp, work_ctr[1], sort_ws[SW_NSLICES]);
    }
    g_frags.clear();
    g_regs.clear();
    if (cfg.defer) {
        if (bm_counts[2]) return FPL_ERR_CAPACITY;
        g_frags.assign(frags.begin(), frags.begin() + bm_counts[0]);
        g_regs.assign(regs.begin(), regs.begin() + bm_counts[1]);
        std::sort(g_frags.begin(), g_frags.end(), [](const fpl_fragment& x, const fpl_fragment& y) {
            return x.read != y.read ? x.read < y.read : x.seq_no < y.seq_no;
        });
    }
    return 0;
}

/* k_count_end_kmers (the counting of the adapter auto-detection) on the emulator: counts / position_acc of 4^10 entries */
extern "C" void emu_count_end_kmers(const uint8_t* seq, const uint64_t* off, uint32_t n_reads, int side, int shift_tail,
                                    uint32_t* counts, unsigned long long* position_acc, unsigned long long* total) {
    memset(counts, 0, sizeof(uint32_t) << 20);
    memset(position_acc, 0, sizeof(unsigned long long) << 20);
    *total = 0;
    emu_launch(k_count_end_kmers, dim3(3), dim3(256), seq, off, n_reads, side, shift_tail, counts, position_acc, total);
}

/* direct hooks for unit tests of the bit-parallel Levenshtein code */
extern "C" int emu_lev_bp64(const char* adapter, int alen, int shift, int m, const char* text, int n) {
    static DevAdapter ad;
    build_adapter(&ad, adapter, alen);
    return lev_bp64(ad.peq_full, shift, m, (const u8*)text, n);
}
/* the wave-cooperative, thresholded form: one 64-thread block, result from lane 0 */
static DevAdapter g_ad;
static int g_res;
static void k_lev_wave(int shift, int m, const u8* text, int n, int thr) {
    int r = lev_wave(g_ad.peq_full, shift, m, text, n, thr);
    if (lane_id() == 0) g_res = r;
}
extern "C" int emu_lev_wave(const char* adapter, int alen, int shift, int m, const char* text, int n, int thr) {
    build_adapter(&g_ad, adapter, alen);
    emu_launch(k_lev_wave, dim3(1), dim3(64), shift, m, (const u8*)text, n, thr);
    return g_res;
}
extern "C" int emu_lev_bp32_start(const char* adapter, int alen, const char* text, int n) {
    static DevAdapter ad;
    build_adapter(&ad, adapter, alen);
    return lev_bp32(ad.peq16_start, ad.plen, (const u8*)text, n);
}
extern "C" int emu_lev_bp32_end(const char* adapter, int alen, const char* text, int n) {
    static DevAdapter ad;
    build_adapter(&ad, adapter, alen);
    return lev_bp32(ad.peq16_end, ad.plen, (const u8*)text, n);
}

/* k_pick_adapter on the emulator: seed + grown adapter from host tables (out: >= 72 bytes); returns the seed key */
extern "C" int emu_pick_adapter(const uint32_t* counts, const uint64_t* position_acc, int is_rna, uint32_t* count, uint32_t* total_key,
                                char* out) {
    pick::Pick p;
    emu_launch(k_pick_adapter, dim3(1), dim3(128), counts, (const unsigned long long*)position_acc, is_rna, &p);
    if (count) *count = p.count;
    if (total_key) *total_key = p.total_key;
    memcpy(out, p.seq, (size_t)p.len + 1);
    return p.key;
}
