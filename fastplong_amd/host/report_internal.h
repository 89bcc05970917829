/* report_internal.h -- pieces the JSON and the HTML report writers share (not part of any interface). */
#ifndef FPLH_REPORT_INTERNAL_H
#define FPLH_REPORT_INTERNAL_H

#include <stdint.h>

#include <string>

#include "fastplong_amd.h"

namespace fplh {
namespace detail {

/* per-cycle accessors on one Stats block, cls = base ASCII & 7 */
struct StatsBlock {
    const int64_t* st;
    uint32_t C;
    long cyc(uint32_t c, int kind, int cls) const { return st[FPL_ST_CYC(c, kind, cls)]; }
    long total_base(uint32_t c) const {
        long t = 0;
        for (int b = 0; b < 8; b++) t += cyc(c, 0, b);
        return t;
    }
    long total_qual(uint32_t c) const {
        long t = 0;
        for (int b = 0; b < 8; b++) t += cyc(c, 1, b);
        return t;
    }
    long base_qual_hist(int q) const { return st[FPL_ST_BASE_QUAL_HIST(C) + q]; }
    long kmer(int i) const { return st[FPL_ST_KMER(C) + i]; }
    long reads() const { return st[FPL_ST_READS(C)]; }
    long length_sum() const { return st[FPL_ST_LENGTH_SUM(C)]; }
};

inline std::string kmer3(int val, bool is_rna) { /* Stats::kmer3 / kmer2, src/stats.cpp:826-845 */
    const char bases[4] = {'A', is_rna ? 'U' : 'T', 'C', 'G'};
    std::string ret(3, ' ');
    ret[0] = bases[(val & 0x30) >> 4];
    ret[1] = bases[(val & 0x0C) >> 2];
    ret[2] = bases[(val & 0x03)];
    return ret;
}
inline std::string kmer2(int val, bool is_rna) {
    const char bases[4] = {'A', is_rna ? 'U' : 'T', 'C', 'G'};
    std::string ret(2, ' ');
    ret[0] = bases[(val & 0x0C) >> 2];
    ret[1] = bases[(val & 0x03)];
    return ret;
}

}  // namespace detail
}  // namespace fplh
#endif
