/*
 * report.h -- host side of the path after the counters come back: what the reference does in
 * Stats::summarize (src/stats.cpp:150-256), Stats::reportJson (:473-548),
 * FilterResult::report*Json (src/filterresult.cpp:120-204) and JsonReporter::report
 * (src/jsonreporter.cpp:11-94) and HtmlReporter::report (src/htmlreporter.cpp), working from the flat int64 counter buffer of
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

/* What Stats::statRead keeps PER READ beyond the counters (mLengthVec and mQualLength, src/stats.cpp:262,352-368);
 * only the HTML report reads it.  `worker` says which of the reference's workers would have seen the read: it deals
 * packs of PACK_SIZE = 16 reads round-robin to its workers (src/seprocessor.cpp:343-378) and merges the workers'
 * lists in worker order (src/stats.cpp:1066-1075), which fixes the order of the density plot's points. */
struct ReadLists {
    std::vector<uint8_t> worker; /* worker_of(input index of the read) */
    std::vector<int32_t> len;
    std::vector<uint8_t> median; /* quality char; ignored when len == 0 */
    static uint8_t worker_of(uint64_t read_index, int threads) { return (uint8_t)((read_index / 16) % (uint64_t)threads); }
    void add(uint8_t w, int32_t l, uint8_t m) {
        worker.push_back(w);
        len.push_back(l);
        median.push_back(m);
    }
};
struct HtmlInputs {
    ReadLists pre, post;
    int threads = 1;                        /* Options::thread after validate(): 1..16 */
    std::string title = "fastplong report"; /* -R */
    std::string timestamp;                  /* empty: now, in HtmlReporter::getCurrentSystemTime's format */
};
/* HtmlReporter::report (src/htmlreporter.cpp:72-176) with the Stats / FilterResult HTML pieces it calls */
bool write_html(const std::string& path, const ReportInputs& in, const HtmlInputs& h);

}  // namespace fplh

extern "C" {
int fplh_write_json(const char* path, const int64_t* counters, uint32_t C, int n_adapters, const char* const* adapters,
                    const int* adapter_lens, int adapter_enabled, int polyx, int complexity, int is_rna,
                    const char* command);
/* test hook: lists as parallel arrays (read index, length, median quality char) */
int fplh_write_html(const char* path, const int64_t* counters, uint32_t C, int n_adapters, const char* const* adapters,
                    const int* adapter_lens, int adapter_enabled, int polyx, int complexity, int is_rna,
                    int length_filter, int max_length, const char* command, int threads, const char* title,
                    const char* timestamp, uint64_t n_pre, const uint32_t* pre_read, const int32_t* pre_len,
                    const uint8_t* pre_median, uint64_t n_post, const uint32_t* post_read, const int32_t* post_len,
                    const uint8_t* post_median);
}
#endif
