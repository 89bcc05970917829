/*
 * dev_types.h -- plain-old-data shared by the kernels and the host side of the C-ABI, plus the
 * host-only code that fills it (thresholds in double, Myers Peq bit-vectors per adapter).
 */
#ifndef FPL_DEV_TYPES_H
#define FPL_DEV_TYPES_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/fastplong_amd.h"

namespace fpl {

constexpr int PEQ_WORDS = 4; /* 4 x 64 columns >= FPL_MAX_ADAPTER_LEN */

/* One adapter slot in device memory. */
struct DevAdapter {
    int32_t len;  /* alen */
    int32_t plen; /* min(PATTERN_LEN, alen), reference src/adaptertrimmer.cpp:181,251 */
    uint8_t seq[256];
    uint32_t seqw[256];          /* seq[i] replicated into the 4 bytes of a dword (SWAR compare) */
    uint32_t peq16_start[256];   /* Myers Peq of the LAST plen bytes  (start trim, :203)  */
    uint32_t peq16_end[256];     /* Myers Peq of the FIRST plen bytes (end trim,   :274)  */
    uint64_t peq_full[256][PEQ_WORDS]; /* Myers Peq of the whole adapter, bit j <-> seq[j]  */
    /* bit-sliced middle-adapter scan (k_scan): for adapter offset i, (plane BYTE offset << 8) | shift,
       plane byte offset = 4 * (code(seq[i]) * 64 + (i >> 5)), shift = i & 31, code A0 C1 T2 G3 */
    uint32_t term[64];
    int32_t acgt_only; /* every byte is one of A C G T and len <= 64 */
    /* one-hot nibbles of the first 64 bases (A 1, C 2, G 4, T 8; base i in bits 4i..4i+3 of word i / 8), zero behind the
       adapter: the window Hamming scans of k_trim_ends count matches as popcount(text nibbles & these) */
    uint32_t onehot[8];
    /* the Peq words of the letters A, C, T, G (code = (ASCII >> 1) & 3) in one place, for the lane-per-adapter filter of
       the FASTA chain (k_trim_ends, fasta_may_trim): the whole adapter (first 64 columns), its last 16 bases (start
       trim's partial pattern) and its first 16 (end trim's) */
    uint64_t peq4_full[4];
    uint32_t peq4_s16[4];
    uint32_t peq4_e16[4];
    /* ... and for its cheaper form (FPL_OPT_FASTAFILTER 2): the first min(32, len) bases in reading order (end trim) and the
       last min(32, len) bases in REVERSE order (start trim, whose window is then walked backwards): in both the trim's
       16-base partial pattern is the first 16 columns */
    uint32_t peq4_e32[4];
    uint32_t peq4_s32r[4];
};

/* Options as the kernels consume them: integers only. */
struct DevConfig {
    int32_t trim_front, trim_tail;
    int32_t cut_front, cut_tail;
    int32_t cut_front_w, cut_front_thr; /* thr = (33 + quality) * window: total >= thr  <=>  mean >= 33+q */
    int32_t cut_tail_w, cut_tail_thr;
    int32_t polyx, polyx_min_len;
    int32_t adapter_enabled, ext;
    int32_t has_start, has_end, n_fasta;
    int32_t trim_mode; /* host-side dispatch of k_trim_ends<MODE>: 1 = no FASTA adapters, command-line adapters of 16..32 bases
                          (or none); 2 = every adapter 16..64 bases; else 0 (set by trim_mode_of once the adapters are known) */
    int32_t qual_filter, qualified_qual, unqual_pct, n_base_limit, n_pct_limit, avg_qual_req;
    int32_t length_filter, required_length, max_length;
    int32_t complexity, complexity_pct;
    /* --break / --mask (src/seprocessor.cpp:234-262): window and (33 + quality) * window; with either one on,
       k_scan leaves the fragments' fate to k_break_mask (defer) */
    int32_t brk, brk_w, brk_thr, msk, msk_w, msk_thr, defer;
    int32_t dbg;      /* FPL_DEBUG_FLAGS: ablation switches for profiling (wrong results when set) */
    int32_t ham_fast; /* both command-line adapters are ACGT-only and <= 64 long: bit-sliced scan */
    int32_t scan_short; /* host-side dispatch: adapter trimming on, ham_fast and both adapters <= 32 bases -> k_scan<SHORT> */
    int32_t thr[FPL_MAX_ADAPTER_LEN + 1]; /* (int)round(ed_max * len), computed in double on the host */
};

/* r1 after the end trims, in coordinates of the original read */
struct ReadState {
    uint32_t s, e;
    uint32_t dropped;
    uint32_t pad;
};

inline void build_adapter(DevAdapter* a, const char* seq, int len) {
    memset(a, 0, sizeof(*a));
    a->len = len;
    a->plen = len < FPL_PATTERN_LEN ? len : FPL_PATTERN_LEN;
    for (int i = 0; i < len; i++) {
        uint8_t c = (uint8_t)seq[i];
        a->seq[i] = c;
        a->seqw[i] = 0x01010101u * c;
        a->peq_full[c][i >> 6] |= 1ull << (i & 63);
    }
    a->acgt_only = len <= 64;
    for (int i = 0; i < 64; i++) a->term[i] = (4u * 4u * 64u) << 8; /* padding: the all-zero plane row */
    for (int i = 0; i < len && i < 64; i++) {
        const uint8_t c = (uint8_t)seq[i];
        if (c != 'A' && c != 'C' && c != 'G' && c != 'T') a->acgt_only = 0;
        const uint32_t code = (c >> 1) & 3u; /* A0 C1 T2 G3 */
        a->term[i] = ((4u * (code * 64u + (uint32_t)(i >> 5))) << 8) | (uint32_t)(i & 31);
    }
    for (int i = 0; i < len && i < 64; i++) {
        const uint8_t c = (uint8_t)seq[i];
        const uint32_t nib = c == 'A' ? 1u : (c == 'C' ? 2u : (c == 'G' ? 4u : (c == 'T' ? 8u : 0u)));
        a->onehot[i >> 3] |= nib << (4 * (i & 7));
    }
    for (int j = 0; j < a->plen; j++) {
        a->peq16_start[(uint8_t)seq[len - a->plen + j]] |= 1u << j;
        a->peq16_end[(uint8_t)seq[j]] |= 1u << j;
    }
    {
        static const uint8_t letters[4] = {'A', 'C', 'T', 'G'};
        for (int c = 0; c < 4; c++) {
            a->peq4_full[c] = a->peq_full[letters[c]][0];
            a->peq4_s16[c] = a->peq16_start[letters[c]];
            a->peq4_e16[c] = a->peq16_end[letters[c]];
            a->peq4_e32[c] = 0;
            a->peq4_s32r[c] = 0;
            const int m = len < 32 ? len : 32;
            for (int i = 0; i < m; i++) {
                if ((uint8_t)seq[i] == letters[c]) a->peq4_e32[c] |= 1u << i;
                if ((uint8_t)seq[len - 1 - i] == letters[c]) a->peq4_s32r[c] |= 1u << i;
            }
        }
    }
}

/* DevConfig::trim_mode from the adapters (slot order: start, end, FASTA...; a length of 0 = slot unused): the reduced
   instantiations take adapters of A / C / G / T only (their window scans work on one-hot nibbles) */
inline int trim_mode_of(const int* lens, const int* acgt_only, int n_adapters) {
    bool short_ok = n_adapters == 2, mid_ok = true;
    for (int i = 0; i < n_adapters; i++) {
        const int l = lens[i];
        if (l == 0 && i < 2) continue; /* command-line adapter not given */
        if (l < 16 || l > 32 || !acgt_only[i]) short_ok = false;
        if (l < 16 || l > 64 || !acgt_only[i]) mid_ok = false;
    }
    return short_ok ? 1 : (mid_ok ? 2 : 0);
}
inline void build_config(DevConfig* c, const fpl_options* o, int start_len, int end_len, int n_fasta) {
    memset(c, 0, sizeof(*c));
    c->trim_front = o->trim_front;
    c->trim_tail = o->trim_tail;
    c->cut_front = o->cut_front != 0;
    c->cut_tail = o->cut_tail != 0;
    c->cut_front_w = o->cut_front_window;
    c->cut_front_thr = (33 + o->cut_front_quality) * o->cut_front_window;
    c->cut_tail_w = o->cut_tail_window;
    c->cut_tail_thr = (33 + o->cut_tail_quality) * o->cut_tail_window;
    c->polyx = o->polyx != 0;
    c->polyx_min_len = o->polyx_min_len;
    c->adapter_enabled = o->adapter_enabled != 0;
    c->ext = o->trimming_extension;
    c->has_start = start_len > 0;
    c->has_end = end_len > 0;
    c->n_fasta = n_fasta;
    c->trim_mode = 0;
    c->qual_filter = o->qual_filter != 0;
    c->qualified_qual = o->qualified_qual;
    c->unqual_pct = o->unqualified_percent_limit;
    c->n_base_limit = o->n_base_limit;
    c->n_pct_limit = o->n_base_percent_limit;
    c->avg_qual_req = o->avg_qual_req;
    c->length_filter = o->length_filter != 0;
    c->required_length = o->required_length;
    c->max_length = o->max_length;
    c->complexity = o->complexity_filter != 0;
    int y = o->complexity_percent; /* src/main.cpp:219 clamps -Y to 0..100 */
    c->complexity_pct = y < 0 ? 0 : (y > 100 ? 100 : y);
    c->brk = o->break_enabled != 0;
    c->brk_w = o->break_window;
    c->brk_thr = (33 + o->break_quality) * o->break_window;
    c->msk = o->mask_enabled != 0;
    c->msk_w = o->mask_window;
    c->msk_thr = (33 + o->mask_quality) * o->mask_window;
    c->defer = c->brk || c->msk;
    for (int l = 0; l <= FPL_MAX_ADAPTER_LEN; l++) c->thr[l] = (int)round(o->ed_max * l);
}

}  // namespace fpl
#endif
