/*
 * report.h -- host side of the path after the counters come back: what the reference does in
 * Stats::summarize (src/stats.cpp:150-256), Stats::reportJson (:473-548),
 * FilterResult::report*Json (src/filterresult.cpp:120-204) and JsonReporter::report
 * (src/jsonreporter.cpp:11-94), working from the flat int64 counter buffer of
 * include/fastplong_amd.h instead of the reference's per-thread Stats/FilterResult objects.
 */
#ifndef FPLH_REPORT_H
#define FPLH_REPORT_H

#include <stdint.h>

#include <string>
#include <vector>

#include "fastplong_amd.h"

namespace fplh {

struct ReportInputs {
    const int64_t* counters = nullptr; /* FPL_COUNTERS_LEN(C, n_adapters) */
    uint32_t C = 0;
    std::vector<std::string> adapters; /* slot order: start, end, FASTA... */
    bool adapter_enabled = true;       /* Options::adapter.enabled */
    bool polyx = false;                /* Options::polyXTrim.enabled */
    bool complexity = false;           /* Options::complexityFilter.enabled */
    bool length_filter = true;
    int max_length = 0;
    bool is_rna = false;               /* Options::isRNA (U instead of T in curve names) */
    std::string command;               /* src/main.cpp:252-256 */
};

/* One summarized Stats block (the members JSON / stderr need). */
struct StatsSummary {
    long reads = 0, bases = 0, q20 = 0, q30 = 0, length_sum = 0, gc = 0;
    int cycles = 0;
    int mean_length() const { return reads == 0 ? 0 : (int)(length_sum / reads); }
};

StatsSummary summarize(const int64_t* stats, uint32_t C);
bool write_json(const std::string& path, const ReportInputs& in);
/* the "Before filtering / After filtering / Filtering result" text the reference prints to
 * stderr (src/seprocessor.cpp:129-137, Stats::print, FilterResult::print) */
std::string summary_text(const ReportInputs& in);

}  // namespace fplh

extern "C" {
int fplh_write_json(const char* path, const int64_t* counters, uint32_t C, int n_adapters, const char* const* adapters,
                    const int* adapter_lens, int adapter_enabled, int polyx, int complexity, int is_rna,
                    const char* command);
}
#endif
