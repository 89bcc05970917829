/*
 * adapter_pick.h -- what the adapter auto-detection does with the 4^10 end-k-mer counters once they are counted
 * (k_count_end_kmers): pick the seed key and grow it into an adapter.  One implementation for the device (k_pick_adapter,
 * kernels.h) and for the host (host/evaluator.cpp, the path of tests and of FPLH_HOST_KMERS).
 *
 * Behaviour restated from the reference -- Evaluator::getTopKey (src/evaluator.cpp:268-326) and
 * Evaluator::extendKeyToAdapter (src/evaluator.cpp:328-404) -- but not their loops:
 *   - which keys may be a seed depends on the key alone, except for one test the reference applies to the key's COUNT; both
 *     are evaluated on all ten 2-bit digits at once (XOR against a replicated digit, fold the bit pairs, popcount), so that
 *     a scan over the 2^20 counters is a masked arg-max -- on the device one pass of 1024 threads;
 *   - of the four keys that extend the current one by a base at most one can hold 70 % of their counts, so the walk looks
 *     at the largest of the four only; each direction is walked once from the seed.
 * Parity status: UNPINNED beyond the reference's own known-answer test for the key coding (test/evaluator_test.cpp):
 * src/evaluator.cpp cannot be built in this image (it includes the FASTQ reader -> ISA-L headers).  tests/ cross-check this
 * file against a literal restatement of the two reference functions that lives with the test infrastructure.
 */
#ifndef FPL_ADAPTER_PICK_H
#define FPL_ADAPTER_PICK_H

#include <stdint.h>

#if defined(__HIPCC__) && !defined(FPL_EMU)
#define FPL_HD __host__ __device__
#else
#define FPL_HD
#endif

namespace fpl {
namespace pick {

constexpr int KEYLEN = 10;                /* bases per key, two bits each: A 0, T/U 1, C 2, G 3; first base = top digit */
constexpr uint32_t NKEYS = 1u << (2 * KEYLEN);
constexpr int MAX_ADAPTER = 64;           /* src/evaluator.cpp:334 */
constexpr uint32_t DIGITS = 0x55555u;     /* the low bit of each of the ten digits */

FPL_HD inline int popc(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(x);
#else
    return __builtin_popcount(x);
#endif
}
/* The window of key positions the detection looks at in a read of rlen bases (src/evaluator.cpp:166-183, :207-225): the first
   128 positions (side 0) or the last 129 in front of the skipped tail (side 1), a key never reaching into the last shift_tail
   bases.  false: the read is too short to hold a key. */
FPL_HD inline bool key_window(long long rlen, int side, int shift_tail, long long& first, long long& last) {
    last = rlen - KEYLEN - shift_tail;
    if (last < 0) return false;
    first = side == 0 ? 0 : (last - 128 > 0 ? last - 128 : 0);
    if (side == 0 && last > 127) last = 127;
    return true;
}
/* the key of the ten bases at data[0..10): false when one of them is not A, T/U, C, G (the reference rolls its key and starts
   over behind such a base: the same set of (position, key) pairs) */
FPL_HD inline bool key_at(const uint8_t* data, uint32_t& key) {
    uint32_t k = 0, bad = 0;
    for (int i = 0; i < KEYLEN; i++) {
        const uint32_t c = data[i];
        const uint32_t code = c == 'A' ? 0u : ((c == 'T' || c == 'U') ? 1u : (c == 'C' ? 2u : (c == 'G' ? 3u : 4u)));
        bad |= code;
        k = (k << 2) | (code & 3u);
    }
    key = k;
    return bad < 4u;
}
/* how many of the ten digits of k equal d */
FPL_HD inline int digits_equal(uint32_t k, uint32_t d) {
    const uint32_t x = k ^ (DIGITS * d); /* 00 where the digit is d */
    return KEYLEN - popc((x | (x >> 1)) & DIGITS);
}
/* may key k be the seed of an adapter?  (src/evaluator.cpp:275-292, :301-311: a base that makes up six or more of the ten,
   two bases absent, two identical halves, eight or more G / C, a GGGG head, and poly-A are out) */
FPL_HD inline bool key_admissible(uint32_t k) {
    if (k == 0) return false;
    const int a = digits_equal(k, 0), t = digits_equal(k, 1), c = digits_equal(k, 2), g = digits_equal(k, 3);
    if (a >= KEYLEN - 4 || t >= KEYLEN - 4 || c >= KEYLEN - 4 || g >= KEYLEN - 4) return false;
    if ((a == 0) + (t == 0) + (c == 0) + (g == 0) >= 2) return false;
    if ((k >> KEYLEN) == (k & ((1u << KEYLEN) - 1u))) return false;
    if (c + g >= KEYLEN - 2) return false;
    if ((k >> 12) == 0xFFu) return false;
    return true;
}
/* the test the reference makes on the COUNT of a key (src/evaluator.cpp:293-300 walks `val`, not the key): read as 2-bit digits,
   at least three of the nine neighbouring pairs among digits 1..10 of the count must differ */
FPL_HD inline bool count_digits_vary(uint32_t val) {
    const uint32_t x = (val ^ (val >> 2)) >> 2; /* digit j: digit j + 1 of val against digit j + 2 */
    return popc((x | (x >> 1)) & 0x15555u) >= 3;
}

struct Pick {
    int32_t key;        /* the seed, -1 when no key qualifies */
    uint32_t count;     /* its count */
    uint32_t total_key; /* keys that were seen at all (poly-A included) */
    int32_t len;        /* bases in seq (0 without a seed) */
    char seq[MAX_ADAPTER + 8];
};

/* (count, smaller key wins) as one comparable word */
FPL_HD inline uint64_t seed_rank(uint32_t val, uint32_t k) { return ((uint64_t)val << 32) | (uint32_t)~k; }

/* Grow the seed in one direction (left: towards the read's start) while one of the four one-base extensions of the current
   key is clearly THE continuation: it holds at least 70 % of the four keys' counts and half the seed's count, and its mean
   position sits next to the current key's (src/evaluator.cpp:344-385).  The comparisons are the reference's, in double.
   CNT(k) / POS(k): the counters (poly-A reads as zero, src/evaluator.cpp:191).  Writes the new bases in walking order and returns
   how many; room = bases the adapter may still take. */
template <class CNT, class POS>
FPL_HD inline int walk(uint32_t seed, bool left, int room, bool is_rna, CNT cnt, POS pos, char* out) {
    const uint32_t mask = NKEYS - 1u;
    const double seed_count = (double)cnt(seed);
    uint32_t cur = seed;
    int n = 0;
    while (n < room) {
        uint32_t best = 0, best_c = 0, sum = 0;
        int best_b = -1;
        for (uint32_t b = 0; b < 4; b++) {
            const uint32_t nk = left ? ((b << (2 * (KEYLEN - 1))) | (cur >> 2)) : (b | ((cur << 2) & mask));
            const uint32_t c = cnt(nk);
            sum += c;
            if (c > best_c) { /* (a 70 % share cannot be tied) */
                best_c = c;
                best = nk;
                best_b = (int)b;
            }
        }
        if (best_b < 0) break; /* none of the four was ever seen */
        if ((double)best_c / (double)sum < 0.7) break;
        if ((double)best_c / seed_count < 0.5) break;
        const double shift = (double)pos(best) / (double)best_c - (double)pos(cur) / (double)cnt(cur);
        if (shift > 2 || shift < -4) break;
        cur = best;
        out[n++] = best_b == 0 ? 'A' : (best_b == 1 ? (is_rna ? 'U' : 'T') : (best_b == 2 ? 'C' : 'G'));
    }
    return n;
}

/* the adapter around a seed: left walk, then right walk with what room is left (src/evaluator.cpp:336-403) */
template <class CNT, class POS>
FPL_HD inline void grow(Pick& p, bool is_rna, CNT cnt, POS pos) {
    char lbuf[MAX_ADAPTER], rbuf[MAX_ADAPTER];
    const int nl = walk((uint32_t)p.key, true, MAX_ADAPTER - KEYLEN, is_rna, cnt, pos, lbuf);
    const int nr = walk((uint32_t)p.key, false, MAX_ADAPTER - KEYLEN - nl, is_rna, cnt, pos, rbuf);
    int n = 0;
    for (int i = nl - 1; i >= 0; i--) p.seq[n++] = lbuf[i];
    for (int i = KEYLEN - 1; i >= 0; i--) {
        const uint32_t d = ((uint32_t)p.key >> (2 * i)) & 3u;
        p.seq[n++] = d == 0 ? 'A' : (d == 1 ? (is_rna ? 'U' : 'T') : (d == 2 ? 'C' : 'G'));
    }
    for (int i = 0; i < nr; i++) p.seq[n++] = rbuf[i];
    p.seq[n] = 0;
    p.len = n;
}

}  // namespace pick
}  // namespace fpl
#endif
