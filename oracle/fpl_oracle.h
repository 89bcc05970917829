/*
 * fpl_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the per-read hot path of OpenGene/fastplong v0.4.1
 * (SingleEndProcessor::processSingleEnd, reference src/seprocessor.cpp:180-329, and what it
 * calls).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product (fastplong_amd/) never links, imports or executes it.
 *
 * Parity status: PINNED.
 *   - every known-answer test the reference holds for this path (reference test dir and
 *     editdistance_test(), src/editdistance.cpp:141-172) is replayed in
 *     tests/test_oracle_kat.py;
 *   - edit_distance, trimAndCut, trimPolyX, passFilter, statRead/summarize/reportJson,
 *     Read::trimFront/resize/breakByGap/appendToString*, FilterResult JSON and
 *     JsonReporter::report are cross-checked against the real reference objects compiled
 *     into oracle/_ref (see oracle/Makefile) in tests/test_oracle_vs_ref.py;
 *   - the --break / --mask stage (Filter::detectLowQualityRegions, Read::breakByRegions,
 *     Read::maskRegionWithN as processSingleEnd chains them, src/seprocessor.cpp:234-262) is
 *     cross-checked the same way (harness commands LQR / BRK);
 *   - AdapterTrimmer (src/adaptertrimmer.cpp) includes Google Highway, which this image
 *     does not have, so that one file is unbuildable here: for it the restatement is pinned
 *     by the reference's own four known-answer tests only (test/adaptertrimmer_test.cpp).
 */
#ifndef FPL_ORACLE_H
#define FPL_ORACLE_H

#include <stdint.h>
#include "../include/fastplong_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* A read (or fragment) as a window on immutable seq/qual bytes: the reference mutates
 * std::string in place; the window [start, start+len) on the original bytes is equivalent. */
typedef struct orc_read {
    const char* seq; /* original read bases  */
    const char* qual; /* original read quals */
    int start;
    int len;
} orc_read;

typedef struct orc_config {
    fpl_options opt;
    const char* start_adapter;
    int start_len;
    const char* end_adapter;
    int end_len;
    const fpl_adapter* fasta;
    int n_fasta; /* hasFasta = n_fasta > 0 */
} orc_config;

unsigned orc_edit_distance(const char* a, unsigned asize, const char* b, unsigned bsize);
int orc_search_adapter(const char* seq, int rlen, const char* adapter, int alen, double edMax,
                       int searchStart, int searchLen, int asLeft, int asRight);
/* returns 0 if the read survives (window updated), -1 if the reference returns NULL */
int orc_trim_and_cut(orc_read* r, const fpl_options* opt, int* frontTrimmed);
/* poly (0..3) and trimmed length are reported through the pointers when a polyX was cut;
 * returns 1 if FilterResult::addPolyXTrimmed would have been called */
int orc_trim_polyx(orc_read* r, int compareReq, int* poly, int* trimmedLen);
/* key_side/key_len describe the string handed to FilterResult::addAdapterTrimmed
 * (key_len = 0: no call) */
int orc_trim_start(orc_read* r, const char* adapter, int alen, double edMax, int ext, int* key_len);
int orc_trim_end(orc_read* r, const char* adapter, int alen, double edMax, int ext, int* key_len);
int orc_find_middle(const orc_read* r, const char* sa, int salen, const char* ea, int ealen,
                    double edMax, int ext, int* start, int* len);
int orc_pass_filter(const orc_read* r, const fpl_options* opt);
/* stats: int64 block laid out as FPL_STATS_LEN(C) of include/fastplong_amd.h */
void orc_stat_read(int64_t* stats, uint32_t C, const orc_read* r, uint8_t* median_out);

/* Filter::detectLowQualityRegions, src/filter.cpp:83-128, on the window of r: writes up to cap (first, last)
 * pairs (positions relative to the window, last inclusive) and returns how many the reference finds. */
int orc_detect_low_quality_regions(const orc_read* r, int window, int quality, int* first, int* last, int cap);

/* Growing lists the extended flow appends to (plain realloc; test infrastructure) */
typedef struct orc_fraglist {
    fpl_fragment* frag;
    uint32_t n_frag, cap_frag;
    fpl_region* reg;
    uint32_t n_reg, cap_reg;
} orc_fraglist;
void orc_fraglist_free(orc_fraglist* l);

/* The whole of processSingleEnd for one read.  counters: FPL_COUNTERS_LEN(C, nad) int64.
 * With opt.break_enabled / mask_enabled the fragments are appended to `list` (must not be NULL then) with
 * read = read_index, and res follows the convention of fpl_fragment in include/fastplong_amd.h. */
void orc_process_read_ex(const orc_config* cfg, const char* seq, const char* qual, int len,
                         int64_t* counters, uint32_t C, fpl_read_result* res, uint32_t read_index, orc_fraglist* list);
void orc_process_batch_ex(const orc_config* cfg, const uint8_t* seq, const uint8_t* qual,
                          const uint64_t* off, uint32_t n_reads, int64_t* counters, uint32_t C,
                          fpl_read_result* res, orc_fraglist* list);
/* The whole of processSingleEnd for one read.  counters: FPL_COUNTERS_LEN(C, nad) int64. */
void orc_process_read(const orc_config* cfg, const char* seq, const char* qual, int len,
                      int64_t* counters, uint32_t C, fpl_read_result* res);
void orc_process_batch(const orc_config* cfg, const uint8_t* seq, const uint8_t* qual,
                       const uint64_t* off, uint32_t n_reads, int64_t* counters, uint32_t C,
                       fpl_read_result* res);

#ifdef __cplusplus
}
#endif
#endif
