/*
 * text_stand_in.h -- TEST / MEASUREMENT INFRASTRUCTURE ONLY (tests/stub, tools/nulldev): what fpl_process_text_async's device
 * kernels decide, restated on the CPU from the contract in include/fastplong_amd.h so that the stand-in libraries can play
 * a device that parses text.  REGULAR text: records of four lines, every line ended by "\n" or "\r\n" (the last one too),
 * '@' in front of a non-empty name, '+' in front of the third line, as many qualities as bases; at most n_bytes / 64 + 16
 * records.  Never part of the product.
 */
#ifndef FPL_TEXT_STAND_IN_H
#define FPL_TEXT_STAND_IN_H
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../../include/fastplong_amd.h"

struct StandInText {
    fpl_text_result info;
    std::vector<uint32_t> line;    /* 4 per record */
    std::vector<uint64_t> off;     /* CSR */
    std::vector<uint8_t> seq, qual;
};
/* gather: also build the CSR arrays (the test stub needs them for the oracle; the null device does not) */
static inline void stand_in_parse(const uint8_t* t, uint64_t n, bool gather, StandInText& o) {
    memset(&o.info, 0, sizeof o.info);
    o.info.bad_record = ~0ull;
    o.line.clear();
    o.off.assign(1, 0);
    o.seq.clear();
    o.qual.clear();
    std::vector<uint32_t> nl;
    bool irregular = false;
    for (const uint8_t* p = t; p < t + n;) {
        const uint8_t* q = (const uint8_t*)memchr(p, '\n', (size_t)(t + n - p));
        if (!q) break;
        nl.push_back((uint32_t)(q - t));
        p = q + 1;
    }
    for (const uint8_t* p = t; p < t + n;) { /* a '\r' that is not followed by '\n' */
        const uint8_t* q = (const uint8_t*)memchr(p, '\r', (size_t)(t + n - p));
        if (!q) break;
        if (q + 1 >= t + n || q[1] != '\n') irregular = true;
        p = q + 1;
    }
    o.info.n_lines = (uint32_t)nl.size();
    if (nl.size() % 4 != 0 || (n > 0 && t[n - 1] != '\n')) irregular = true;
    const uint64_t n_rec = nl.size() / 4, cap = n / 64 + 16;
    const bool too_many = n_rec > cap;
    for (uint64_t r = 0; r < n_rec && r < cap; r++) {
        uint32_t L[5];
        L[0] = r ? nl[4 * r - 1] + 1 : 0;
        for (int j = 0; j < 4; j++) L[j + 1] = nl[4 * r + j] + 1;
        uint32_t ll[4];
        for (int j = 0; j < 4; j++) {
            uint32_t e = L[j + 1] - 1;
            if (e > L[j] && t[e - 1] == '\r') e--;
            ll[j] = e - L[j];
            o.line.push_back(L[j]);
        }
        const bool good = ll[0] > 0 && t[L[0]] == '@' && ll[2] > 0 && t[L[2]] == '+' && ll[1] == ll[3];
        if (!good) {
            irregular = true;
            if (r < o.info.bad_record) o.info.bad_record = r;
        }
        o.off.push_back(o.off.back() + ll[1]);
        if (ll[1] > o.info.max_read_len) o.info.max_read_len = ll[1];
        if (gather && good) {
            o.seq.insert(o.seq.end(), t + L[1], t + L[1] + ll[1]);
            o.qual.insert(o.qual.end(), t + L[3], t + L[3] + ll[1]);
        }
    }
    o.info.status = irregular ? FPL_TEXT_IRREGULAR : too_many ? FPL_TEXT_TOO_MANY : FPL_TEXT_OK;
    if (o.info.status == FPL_TEXT_OK) {
        o.info.n_reads = (uint32_t)n_rec;
        o.info.n_bases = o.off.back();
    } else {
        o.info.max_read_len = 0;
    }
}
#endif
