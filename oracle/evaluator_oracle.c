/*
 * evaluator_oracle.c -- TEST INFRASTRUCTURE ONLY (part of oracle/liboracle.so).
 *
 * Literal restatement, loop by loop, of the two functions of the reference's adapter auto-detection that decide on the
 * counters: Evaluator::getTopKey (src/evaluator.cpp:268-326) and Evaluator::extendKeyToAdapter (src/evaluator.cpp:328-404).
 * The product computes the same thing in another form (fastplong_amd/csrc/adapter_pick.h: a masked arg-max and two
 * directional walks, on the device); tests/test_host_evaluator.py compares the two on seeded counter tables.
 *
 * Also here: the counting loops in front of them (evalAdapterAndReadNum, src/evaluator.cpp:166-183 and :207-225) with the
 * rolling key coder they call (Evaluator::seq2int, src/evaluator.cpp:503-560) -- what fpl_count_end_kmers / fpl_pick_adapter are
 * checked against on the GPU.
 *
 * Parity status: UNPINNED.  src/evaluator.cpp includes the FASTQ reader and through it ISA-L headers this image lacks, so the
 * real object cannot be compiled here; the reference's only known-answer test for this file (test/evaluator_test.cpp) covers
 * int2seq / seq2int, not these two functions.  What this file gives is an independent SECOND reading of the reference.
 */
#include <stdint.h>
#include <string.h>

/* Evaluator::getTopKey, src/evaluator.cpp:268-326 (its diff test reads the COUNT's bits, `val`, not the key) */
int orc_eval_top_key(const uint32_t* counts, int keylen) {
    const int size = 1 << (keylen * 2);
    int topkey = -1;
    unsigned int top_count = 0;
    for (int k = 0; k < size; k++) {
        const unsigned int val = counts[k];
        int atcg[4] = {0, 0, 0, 0};
        for (int i = 0; i < keylen; i++) atcg[(k >> (i * 2)) & 0x03]++;
        int low_complexity = 0;
        int zero_num = 0;
        for (int b = 0; b < 4; b++) {
            if (atcg[b] >= keylen - 4) low_complexity = 1;
            if (atcg[b] == 0) zero_num++;
        }
        if (zero_num >= 2) low_complexity = 1;
        if ((k >> keylen) == (k & ((0x01 << keylen) - 1))) low_complexity = 1; /* :287 repetitive */
        int diff = 0;
        for (int s = 0; s < keylen - 1; s++) { /* :293-299 */
            const int cur = (val >> ((keylen - s) * 2)) & 0x03;
            const int last = (val >> ((keylen - s - 1) * 2)) & 0x03;
            if (cur != last) diff++;
        }
        if (diff < 3) continue;
        if (low_complexity) continue;
        if (atcg[2] + atcg[3] >= keylen - 2) continue; /* :305 */
        if ((k >> 12) == 0xff) continue;               /* :309 */
        if (k == 0) continue;
        if (val > top_count) {
            top_count = val;
            topkey = k;
        }
    }
    return topkey;
}

/* Evaluator::extendKeyToAdapter, src/evaluator.cpp:328-404.  out: >= 65 bytes.  Returns the length. */
int orc_eval_extend_key(int key, const uint32_t* counts, const uint64_t* position_acc, int keylen, int is_rna, int left_first,
                        char* out) {
    char buf[160];
    int lo = 80, hi = 80; /* the adapter is buf[lo, hi) */
    char bases[4] = {'A', 'T', 'C', 'G'};
    if (is_rna) bases[1] = 'U';
    for (int i = 0; i < keylen; i++) buf[hi++] = bases[(key >> (2 * (keylen - 1 - i))) & 3]; /* int2seq, :485-497 */
    const int mask = (1 << (keylen * 2)) - 1;
    const int MAX_LEN = 64;
    int left_finished = 0, right_finished = 0;
    int extending_left = left_first;
    while (1) {
        int curkey = key;
        while (hi - lo < MAX_LEN) {
            int total_count = 0;
            int extended = 0;
            for (int b = 0; b < 4; b++) {
                const int newkey = extending_left ? ((b << ((keylen - 1) * 2)) | (curkey >> 2)) : (b | (mask & (curkey << 2)));
                total_count += (int)counts[newkey];
            }
            for (int b = 0; b < 4; b++) {
                const int newkey = extending_left ? ((b << ((keylen - 1) * 2)) | (curkey >> 2)) : (b | (mask & (curkey << 2)));
                if (counts[newkey] == 0) continue;
                const double offset = (double)position_acc[newkey] / counts[newkey] - (double)position_acc[curkey] / counts[curkey];
                if ((double)counts[newkey] / (double)total_count < 0.7) continue;
                if ((double)counts[newkey] / (double)counts[key] < 0.5) continue;
                if (offset > 2 || offset < -4) continue; /* :371 */
                curkey = newkey;
                extended = 1;
                if (extending_left) buf[--lo] = bases[b];
                else buf[hi++] = bases[b];
                break;
            }
            if (!extended) {
                if (extending_left) left_finished = 1;
                else right_finished = 1;
                break;
            }
            if (hi - lo == MAX_LEN) {
                left_finished = 1;
                right_finished = 1;
                break;
            }
        }
        extending_left = !extending_left;
        if (left_finished && right_finished) break;
    }
    memcpy(out, buf + lo, (size_t)(hi - lo));
    out[hi - lo] = 0;
    return hi - lo;
}

/* Evaluator::seq2int, src/evaluator.cpp:503-560: rolls the previous key when there is one, else codes keylen bases afresh;
   -1 on anything but A, T/U, C, G (this is the coder the reference's KAT test/evaluator_test.cpp round-trips) */
int orc_eval_seq2int(const char* seq, int pos, int keylen, int last_val) {
    if (last_val >= 0) {
        const int mask = (1 << (keylen * 2)) - 1;
        int key = (last_val << 2) & mask;
        switch (seq[pos + keylen - 1]) {
            case 'A': key += 0; break;
            case 'T':
            case 'U': key += 1; break;
            case 'C': key += 2; break;
            case 'G': key += 3; break;
            default: return -1;
        }
        return key;
    }
    int key = 0;
    for (int i = pos; i < keylen + pos; i++) {
        key <<= 2;
        switch (seq[i]) {
            case 'A': key += 0; break;
            case 'T':
            case 'U': key += 1; break;
            case 'C': key += 2; break;
            case 'G': key += 3; break;
            default: return -1;
        }
    }
    return key;
}

/* the two counting loops of Evaluator::evalAdapterAndReadNum: side 0 = read start (src/evaluator.cpp:166-183), side 1 = read
   end (src/evaluator.cpp:207-225); reads as one CSR batch.  counts / position_acc: 4^10 entries, zeroed here (:163-164). */
void orc_eval_count_end_kmers(const uint8_t* seq, const uint64_t* off, uint32_t n_reads, int side, int shift_tail,
                              uint32_t* counts, uint64_t* position_acc, uint64_t* total_out) {
    const int keylen = 10;
    const int size = 1 << (keylen * 2);
    long total = 0;
    memset(counts, 0, sizeof(uint32_t) * (size_t)size);
    memset(position_acc, 0, sizeof(uint64_t) * (size_t)size);
    for (uint32_t i = 0; i < n_reads; i++) {
        const char* data = (const char*)seq + off[i];
        const int length = (int)(off[i + 1] - off[i]);
        int key = -1;
        if (side == 0) {
            for (int pos = 0; pos <= length - keylen - shift_tail && pos < 128; pos++) {
                key = orc_eval_seq2int(data, pos, keylen, key);
                if (key >= 0) {
                    counts[key]++;
                    position_acc[key] += (uint64_t)pos;
                    total++;
                }
            }
        } else {
            int startpos = length - keylen - shift_tail - 128;
            if (startpos < 0) startpos = 0;
            for (int pos = startpos; pos <= length - keylen - shift_tail; pos++) {
                key = orc_eval_seq2int(data, pos, keylen, key);
                if (key >= 0) {
                    counts[key]++;
                    position_acc[key] += (uint64_t)(length - pos);
                    total++;
                }
            }
        }
    }
    *total_out = (uint64_t)total;
}
