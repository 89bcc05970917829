/* report.cpp -- see report.h.  Formatting goes through std::ostream exactly like the reference
 * (default precision, `endl`, tabs), so the bytes match as long as the numbers do. */
#include "report.h"
#include "report_internal.h"

#include <charconv>
#include <cmath>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <thread>
#include <utility>
#include <vector>

using namespace std;

namespace fplh {

namespace {
using detail::StatsBlock;
using detail::kmer2;
using detail::kmer3;

/* adapter strings ordered by (length, then lexicographic): struct classcomp, src/filterresult.h:14-23 */
struct ByLenThenLex {
    bool operator()(const string& a, const string& b) const {
        if (a.length() != b.length()) return a.length() < b.length();
        return a < b;
    }
};

/* Stats::reportJson, src/stats.cpp:473-548 (curves from Stats::summarize, :204-244) */
/* the eleven per-cycle lists of one Stats block as text (five quality curves, six content curves) */
struct CurveTexts {
    string t[11];
};
CurveTexts curve_texts(const StatsBlock& s, const StatsSummary& sm, bool is_rna) {
    CurveTexts out;
    string* const curveText = out.t;
    const int cycles = sm.cycles;
    vector<double> mean(cycles);
    for (int c = 0; c < cycles; c++) mean[c] = (double)s.total_qual(c) / (double)s.total_base(c);
    auto qual_curve = [&](char base, int c) {
        int b = base & 0x07;
        long n = s.cyc(c, 0, b);
        return n == 0 ? mean[c] : (double)s.cyc(c, 1, b) / (double)n;
    };
    auto content_curve = [&](char base, int c) { return (double)s.cyc(c, 0, base & 0x07) / (double)s.total_base(c); };
    const char t_or_u = is_rna ? 'U' : 'T';

    string qualNames[5] = {"A", string(1, t_or_u), "C", "G", "mean"};
    string contentNames[6] = {"A", string(1, t_or_u), "C", "G", "N", "GC"};
    /* the eleven per-cycle lists are hundreds of thousands of numbers each for long reads: one thread per list,
       every number still through a default-formatted ostream */
    {
        vector<thread> th;
        for (int k = 0; k < 11; k++)
            th.emplace_back([&, k]() {
                /* (a default-formatted ostream prints a double as printf's %g with six digits, and so does to_chars(general, 6) -- by
                   its definition, and on 3 000 000 random ratios -- at a third of the time; anything not finite keeps the stream) */
                string& o = curveText[k];
                o.reserve((size_t)cycles * 9);
                char buf[48];
                auto put = [&](double v) {
                    if (std::isfinite(v)) {
                        o.append(buf, (size_t)(std::to_chars(buf, buf + sizeof buf, v, std::chars_format::general, 6).ptr - buf));
                    } else {
                        ostringstream t;
                        t << v;
                        o += t.str();
                    }
                };
                for (int c = 0; c < cycles; c++) {
                    if (k < 4) put(qual_curve(qualNames[k][0], c));
                    else if (k == 4) put(mean[c]);
                    else if (k < 10) put(content_curve(contentNames[k - 5][0], c));
                    else put((double)(s.cyc(c, 0, 'G' & 0x07) + s.cyc(c, 0, 'C' & 0x07)) / (double)s.total_base(c));
                    if (c != cycles - 1) o += ',';
                }
            });
        for (auto& t : th) t.join();
    }
    return out;
}
void stats_json(ofstream& ofs, const string& padding, const StatsBlock& s, const StatsSummary& sm, bool is_rna, const CurveTexts& curves) {
    ofs << "{" << endl;
    ofs << padding << "\t" << "\"total_reads\": " << sm.reads << "," << endl;
    ofs << padding << "\t" << "\"total_bases\": " << sm.bases << "," << endl;
    ofs << padding << "\t" << "\"q20_bases\": " << sm.q20 << "," << endl;
    ofs << padding << "\t" << "\"q30_bases\": " << sm.q30 << "," << endl;
    ofs << padding << "\t" << "\"total_cycles\": " << sm.cycles << "," << endl;
    const string* const curveText = curves.t;
    const char t_or_u = is_rna ? 'U' : 'T';
    string qualNames[5] = {"A", string(1, t_or_u), "C", "G", "mean"};
    string contentNames[6] = {"A", string(1, t_or_u), "C", "G", "N", "GC"};
    ofs << padding << "\t" << "\"quality_curves\": {" << endl;
    for (int i = 0; i < 5; i++) {
        ofs << padding << "\t\t" << "\"" << qualNames[i] << "\":[" << curveText[i] << "]";
        if (i != 5 - 1) ofs << ",";
        ofs << endl;
    }
    ofs << padding << "\t" << "}," << endl;

    ofs << padding << "\t" << "\"content_curves\": {" << endl;
    for (int i = 0; i < 6; i++) {
        ofs << padding << "\t\t" << "\"" << contentNames[i] << "\":[" << curveText[5 + i] << "]";
        if (i != 6 - 1) ofs << ",";
        ofs << endl;
    }
    ofs << padding << "\t" << "}," << endl;

    ofs << padding << "\t" << "\"kmer_count\": {" << endl;
    for (int i = 0; i < 64; i++) {
        string first = kmer3(i, is_rna);
        for (int j = 0; j < 16; j++) {
            int target = (i << 4) + j;
            ofs << padding << "\t\t\"" << first << kmer2(j, is_rna) << "\":" << s.kmer(target);
            if (j != 16 - 1) ofs << ",";
        }
        if (i != 64 - 1) ofs << "," << endl;
        else ofs << endl;
    }
    ofs << padding << "\t" << "}" << endl;
    ofs << padding << "}," << endl;
}

/* the reference's map<string,long,classcomp> mAdapter rebuilt from the index histogram */
map<string, long, ByLenThenLex> adapter_map(const ReportInputs& in) {
    map<string, long, ByLenThenLex> m;
    const int nad = (int)in.adapters.size();
    const int64_t* kh = in.counters + FPL_OFF_KEYHIST(in.C);
    for (int a = 0; a < nad; a++) {
        const string& ad = in.adapters[a];
        const int alen = (int)ad.length();
        for (int side = 0; side < 2; side++)
            for (int k = 1; k <= alen && k <= FPL_MAX_ADAPTER_LEN; k++) {
                long n = kh[((size_t)a * 2 + side) * FPL_KEY_STRIDE + k];
                if (n == 0) continue;
                /* side 0: adapterseq.substr(alen - cmplen, cmplen) (src/adaptertrimmer.cpp:225),
                   side 1: adapterseq.substr(0, cmplen) (:293); cmplen == alen is the whole adapter */
                string key = side == 0 ? ad.substr(alen - k, k) : ad.substr(0, k);
                if (key.empty()) continue; /* FilterResult::addAdapterTrimmed ignores "" */
                m[key] += n;
            }
    }
    return m;
}

string read_adapter_name(const string& s) { /* Options::getReadStartAdapter, src/options.cpp:247-259 */
    return (s.empty() || s == "auto") ? "unspecified" : s;
}

}  // namespace

/* Stats::summarize, src/stats.cpp:150-202 */
StatsSummary summarize(const int64_t* stats, uint32_t C) {
    StatsBlock s{stats, C};
    StatsSummary sm;
    sm.reads = s.reads();
    sm.length_sum = s.length_sum();
    sm.cycles = (int)C;
    for (uint32_t c = 0; c < C; c++) {
        long t = s.total_base(c);
        if (t == 0) {
            sm.cycles = (int)c;
            break;
        }
        sm.bases += t;
    }
    for (int c = 0; c < sm.cycles; c++) sm.gc += s.cyc(c, 0, 'G' & 0x07) + s.cyc(c, 0, 'C' & 0x07);
    long q40 = 0;
    for (int c = 40; c < 127 - 33; c++) q40 += s.base_qual_hist(c + 33);
    sm.q30 = q40;
    for (int c = 30; c < 40; c++) sm.q30 += s.base_qual_hist(c + 33);
    sm.q20 = sm.q30;
    for (int c = 20; c < 30; c++) sm.q20 += s.base_qual_hist(c + 33);
    return sm;
}

bool write_json(const string& path, const ReportInputs& in) {
    ofstream ofs;
    ofs.open(path, ifstream::out);
    if (!ofs.is_open()) return false;
    const uint32_t C = in.C;
    const int64_t* pre = in.counters + FPL_OFF_PRE(C);
    const int64_t* post = in.counters + FPL_OFF_POST(C);
    const int64_t* fr = in.counters + FPL_OFF_FR(C);
    StatsSummary a = summarize(pre, C), b = summarize(post, C);
    /* (the two blocks' per-cycle lists are put together side by side, the second beside the writing of everything in front of it) */
    CurveTexts curvesPre, curvesPost;
    thread curvesPostMaker([&]() { curvesPost = curve_texts(StatsBlock{post, C}, b, in.is_rna); });
    curvesPre = curve_texts(StatsBlock{pre, C}, a, in.is_rna);
    const string start = in.adapters.size() > 0 ? in.adapters[0] : "", end = in.adapters.size() > 1 ? in.adapters[1] : "";

    /* JsonReporter::report, src/jsonreporter.cpp:11-94 */
    ofs << "{" << endl;
    ofs << "\t" << "\"summary\": {" << endl;
    ofs << "\t\t" << "\"fastplong_version\": \"" << "0.4.1" << "\"," << endl;
    const StatsSummary* two[2] = {&a, &b};
    const char* names[2] = {"before_filtering", "after_filtering"};
    for (int k = 0; k < 2; k++) {
        const StatsSummary& s = *two[k];
        ofs << "\t\t" << "\"" << names[k] << "\": {" << endl;
        ofs << "\t\t\t" << "\"total_reads\":" << s.reads << "," << endl;
        ofs << "\t\t\t" << "\"total_bases\":" << s.bases << "," << endl;
        ofs << "\t\t\t" << "\"q20_bases\":" << s.q20 << "," << endl;
        ofs << "\t\t\t" << "\"q30_bases\":" << s.q30 << "," << endl;
        ofs << "\t\t\t" << "\"q20_rate\":" << (s.bases == 0 ? 0.0 : (double)s.q20 / (double)s.bases) << "," << endl;
        ofs << "\t\t\t" << "\"q30_rate\":" << (s.bases == 0 ? 0.0 : (double)s.q30 / (double)s.bases) << "," << endl;
        ofs << "\t\t\t" << "\"read_mean_length\":" << s.mean_length() << "," << endl;
        ofs << "\t\t\t" << "\"gc_content\":" << (s.bases == 0 ? 0.0 : (double)s.gc / (double)s.bases) << endl;
        if (k == 0) ofs << "\t\t" << "}," << endl;
        else ofs << "\t\t" << "}";
    }
    ofs << endl;
    ofs << "\t" << "}," << endl;

    /* FilterResult::reportJson, src/filterresult.cpp:120-132 */
    ofs << "\t" << "\"filtering_result\": ";
    {
        const string padding = "\t";
        ofs << "{" << endl;
        ofs << padding << "\t" << "\"passed_filter_reads\": " << fr[FPL_FR_FILTER + FPL_PASS_FILTER] << "," << endl;
        ofs << padding << "\t" << "\"low_quality_reads\": " << fr[FPL_FR_FILTER + FPL_FAIL_QUALITY] << "," << endl;
        ofs << padding << "\t" << "\"too_many_N_reads\": " << fr[FPL_FR_FILTER + FPL_FAIL_N_BASE] << "," << endl;
        if (in.complexity)
            ofs << padding << "\t" << "\"low_complexity_reads\": " << fr[FPL_FR_FILTER + FPL_FAIL_COMPLEXITY] << "," << endl;
        ofs << padding << "\t" << "\"too_short_reads\": " << fr[FPL_FR_FILTER + FPL_FAIL_LENGTH] << "," << endl;
        ofs << padding << "\t" << "\"too_long_reads\": " << fr[FPL_FR_FILTER + FPL_FAIL_TOO_LONG] << endl;
        ofs << padding << "}," << endl;
    }
    /* adapterCuttingEnabled, src/options.cpp:27-33; reportAdapterJson, src/filterresult.cpp:171-185 */
    if (in.adapter_enabled && (!start.empty() || !end.empty())) {
        const string padding = "\t";
        ofs << "\t" << "\"adapter_cutting\": ";
        ofs << "{" << endl;
        ofs << padding << "\t" << "\"adapter_trimmed_reads\": " << fr[FPL_FR_ADAPTER_READS] << "," << endl;
        ofs << padding << "\t" << "\"adapter_trimmed_bases\": " << fr[FPL_FR_ADAPTER_BASES] << "," << endl;
        ofs << padding << "\t" << "\"read_start_adapter\": \"" << read_adapter_name(start) << "\"," << endl;
        ofs << padding << "\t" << "\"read_end_adapter\": \"" << read_adapter_name(end) << "\"," << endl;
        ofs << padding << "\t" << "\"read_adapter_counts\": " << "{";
        { /* what outputAdaptersJson prints (src/filterresult.cpp:134-169): the keys that hold at least 1 % of all trimmed
             reads, in map order, then the rest as "others" -- here the items are collected first and joined afterwards */
            const auto keys = adapter_map(in);
            long all = 0;
            for (const auto& kv : keys) all += kv.second;
            vector<std::pair<string, long>> items;
            long rest = all;
            for (const auto& kv : keys)
                if (all != 0 && !(kv.second / (double)all < 0.01)) {
                    items.emplace_back(kv.first, kv.second);
                    rest -= kv.second;
                }
            if (all != 0 && rest > 0) items.emplace_back("others", rest);
            for (size_t i = 0; i < items.size(); i++)
                ofs << (i ? ", " : "") << "\"" << items[i].first << "\":" << items[i].second;
        }
        ofs << "}";
        ofs << endl;
        ofs << padding << "}," << endl;
    }
    /* reportPolyXTrimJson, src/filterresult.cpp:187-204 (the padding before "{" is the reference's) */
    if (in.polyx) {
        const string padding = "\t";
        const char ATCG[4] = {'A', 'T', 'C', 'G'};
        ofs << "\t" << "\"polyx_trimming\": ";
        ofs << padding << "{" << endl;
        for (int part = 0; part < 2; part++) {
            const int64_t* counts = fr + (part == 0 ? FPL_FR_POLYX_READS : FPL_FR_POLYX_BASES);
            const string key = part == 0 ? "polyx_trimmed_reads" : "polyx_trimmed_bases";
            long total = counts[0] + counts[1] + counts[2] + counts[3];
            ofs << padding << "\t\"total_" << key << "\": " << total << "," << endl;
            ofs << padding << "\t\"" << key << "\":{";
            for (int bb = 0; bb < 4; bb++) {
                if (bb > 0) ofs << ", ";
                ofs << "\"" << ATCG[bb] << "\": " << counts[bb];
            }
            ofs << "}";
            if (part == 0) ofs << "," << endl;
        }
        ofs << endl << padding << "}," << endl;
    }
    ofs << "\t" << "\"read_before_filtering\": ";
    stats_json(ofs, "\t", StatsBlock{pre, C}, a, in.is_rna, curvesPre);
    ofs << "\t" << "\"" << "read_after_filtering" << "\": ";
    curvesPostMaker.join();
    stats_json(ofs, "\t", StatsBlock{post, C}, b, in.is_rna, curvesPost);
    ofs << "\t\"command\": " << "\"" << in.command << "\"" << endl;
    ofs << "}";
    return ofs.good();
}

string summary_text(const ReportInputs& in) {
    const uint32_t C = in.C;
    StatsSummary a = summarize(in.counters + FPL_OFF_PRE(C), C), b = summarize(in.counters + FPL_OFF_POST(C), C);
    const int64_t* fr = in.counters + FPL_OFF_FR(C);
    ostringstream o;
    const StatsSummary* two[2] = {&a, &b};
    const char* heads[2] = {"Before filtering:", "After filtering:"};
    for (int k = 0; k < 2; k++) { /* Stats::print, src/stats.cpp:463-471 */
        const StatsSummary& s = *two[k];
        o << heads[k] << endl;
        o << "total reads: " << s.reads << endl;
        o << "total bases: " << s.bases << endl;
        o << "Q20 bases: " << s.q20 << "(" << (s.q20 * 100.0) / s.bases << "%)" << endl;
        o << "Q30 bases: " << s.q30 << "(" << (s.q30 * 100.0) / s.bases << "%)" << endl;
        o << endl;
    }
    o << "Filtering result:" << endl; /* FilterResult::print, src/filterresult.cpp:98-118 */
    o << "reads passed filter: " << fr[FPL_FR_FILTER + FPL_PASS_FILTER] << endl;
    o << "reads failed due to low quality: " << fr[FPL_FR_FILTER + FPL_FAIL_QUALITY] << endl;
    o << "reads failed due to too many N: " << fr[FPL_FR_FILTER + FPL_FAIL_N_BASE] << endl;
    if (in.length_filter) {
        o << "reads failed due to too short: " << fr[FPL_FR_FILTER + FPL_FAIL_LENGTH] << endl;
        if (in.max_length > 0) o << "reads failed due to too long: " << fr[FPL_FR_FILTER + FPL_FAIL_TOO_LONG] << endl;
    }
    if (in.complexity) o << "reads failed due to low complexity: " << fr[FPL_FR_FILTER + FPL_FAIL_COMPLEXITY] << endl;
    if (in.adapter_enabled) {
        o << "reads with adapter trimmed: " << fr[FPL_FR_ADAPTER_READS] << endl;
        o << "bases trimmed due to adapters: " << fr[FPL_FR_ADAPTER_BASES] << endl;
    }
    if (in.polyx) {
        long r = 0, bs = 0;
        for (int i = 0; i < 4; i++) r += fr[FPL_FR_POLYX_READS + i], bs += fr[FPL_FR_POLYX_BASES + i];
        o << "reads with polyX in 3' end: " << r << endl;
        o << "bases trimmed in polyX tail: " << bs << endl;
    }
    return o.str();
}

}  // namespace fplh

extern "C" int fplh_write_json(const char* path, const int64_t* counters, uint32_t C, int n_adapters,
                               const char* const* adapters, const int* adapter_lens, int adapter_enabled, int polyx,
                               int complexity, int is_rna, const char* command) {
    if (!path || !counters) return -1;
    fplh::ReportInputs in;
    in.counters = counters;
    in.C = C;
    for (int i = 0; i < n_adapters; i++) in.adapters.emplace_back(adapters[i] ? adapters[i] : "", (size_t)adapter_lens[i]);
    in.adapter_enabled = adapter_enabled != 0;
    in.polyx = polyx != 0;
    in.complexity = complexity != 0;
    in.is_rna = is_rna != 0;
    in.command = command ? command : "";
    return fplh::write_json(path, in) ? 0 : -2;
}
