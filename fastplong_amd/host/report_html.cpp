/* report_html.cpp -- fastplong.html from the counter buffer and the per-read (length, median quality) lists:
 * the page HtmlReporter::report lays out (src/htmlreporter.cpp:72-176) with the pieces Stats and FilterResult
 * contribute (src/stats.cpp:589-1011, src/filterresult.cpp:227-242).  The page is data: its fixed text lives in
 * the PAGE_* literals below, the six two-column sections in one table, and every number goes through the same
 * formatting route as in the reference (to_string for "%f", a default ostream for "%g"), so the bytes match
 * whenever the numbers do.  Checked byte for byte against the real HtmlReporter in tests/test_report_html.py. */
#include <time.h>

#include <algorithm>
#include <fstream>
#include <memory>
#include <charconv>
#include <sstream>
#include <thread>
#include <type_traits>

#include "fastq.h"
#include "report.h"
#include "report_internal.h"

using namespace std;

namespace fplh {

namespace {

using detail::StatsBlock;

const char* const PAGE_SCRIPTS = /* HtmlReporter::printJS */
    "<script src='https://opengene.org/plotly-1.2.0.min.js'></script>\n"
    "\n<script type='text/javascript'>\n"
    "    window.Plotly || document.write('<script src=\"https://cdn.plot.ly/plotly-1.2.0.min.js\"><\\/script>')\n"
    "</script>\n"
    "\n<script type=\"text/javascript\">\n"
    "    function showOrHide(divname) {\n"
    "        div = document.getElementById(divname);\n"
    "        if(div.style.display == 'none')\n"
    "            div.style.display = 'block';\n"
    "        else\n"
    "            div.style.display = 'none';\n"
    "    }\n"
    "</script>\n";

const char* const PAGE_STYLE[] = { /* HtmlReporter::printCSS, one rule per line */
    "td {border:1px solid #dddddd;padding:5px;font-size:12px;}",
    "table {border:1px solid #999999;padding:2x;border-collapse:collapse;width:100%}",
    ".col1 {width:240px; font-weight:bold;}",
    ".adapter_col {width:500px; font-size:10px;}",
    "img {padding:30px;}",
    "#menu {font-family:Consolas, 'Liberation Mono', Menlo, Courier, monospace;}",
    "#menu a {color:#0366d6; font-size:18px;font-weight:600;line-height:28px;text-decoration:none;font-family:-apple-system, "
    "BlinkMacSystemFont, 'Segoe UI', Helvetica, Arial, sans-serif, 'Apple Color Emoji', 'Segoe UI Emoji', 'Segoe UI Symbol'}",
    "a:visited {color: #999999}",
    ".alignleft {text-align:left;}",
    ".alignright {text-align:right;}",
    ".figure {width:680px;height:600px;}",
    ".header {color:#ffffff;padding:1px;height:20px;background:#000000;}",
    ".section_title {color:#ffffff;font-size:20px;padding:5px;text-align:left;background:#663355; margin-top:10px;}",
    ".section_table {width:100%;}",
    ".subsection_title {font-size:16px;padding:5px;margin-top:10px;text-align:left;color:#663355}",
    "#container {text-align:center;padding:3px 3px 3px 10px;font-family:Arail,'Liberation Mono', Menlo, Courier, monospace;}",
    ".menu_item {text-align:left;padding-top:5px;font-size:18px;}",
    ".highlight {text-align:left;padding-top:30px;padding-bottom:30px;font-size:20px;line-height:35px;}",
    "#helper {text-align:left;border:1px dotted #fafafa;color:#777777;font-size:12px;}",
    "#footer {text-align:left;padding:15px;color:#ffffff;font-size:10px;background:#663355;font-family:Arail,'Liberation Mono', "
    "Menlo, Courier, monospace;}",
    ".kmer_table {text-align:center;font-size:8px;padding:2px;}",
    ".kmer_table td{text-align:center;font-size:8px;padding:0px;color:#ffffff}",
    ".sub_section_tips {color:#999999;font-size:10px;padding-left:5px;padding-bottom:3px;}",
};
const char* const PROJECT_URL = "https://github.com/OpenGene/fastplong";
const char* const VERSION = "0.4.1";

string now_text() { /* HtmlReporter::getCurrentSystemTime */
    time_t tt = time(NULL);
    struct tm* p = localtime(&tt);
    char date[60] = {0};
    snprintf(date, sizeof(date), "%d-%02d-%02d      %02d:%02d:%02d", p->tm_year + 1900, p->tm_mon + 1, p->tm_mday, p->tm_hour,
             p->tm_min, p->tm_sec);
    return date;
}

/* HtmlReporter::formatNumber / getPercents / outputRow (src/htmlreporter.cpp:14-43) */
string human(long number) {
    static const char* unit[6] = {"", "K", "M", "G", "T", "P"};
    double num = (double)number;
    int order = 0;
    while (num > 1000.0) {
        order++;
        num /= 1000.0;
    }
    return order == 0 ? to_string(number) : to_string(num) + " " + unit[order];
}
string percents(long numerator, long denominator) {
    return denominator == 0 ? string("0.0") : to_string((double)numerator * 100.0 / (double)denominator);
}
void row(ostream& o, const string& key, const string& v) {
    o << "<tr><td class='col1'>" << key << "</td><td class='col2'>" << v << "</td></tr>\n";
}
string div_name(const string& subsection, bool colons_too) { /* replace(s, " ", "_") [and ":" too], src/util.h:98-114 */
    string r = subsection;
    for (char& c : r)
        if (c == ' ' || (colons_too && c == ':')) c = '_';
    return r;
}
template <class T>
string joined(const vector<T>& v) { /* Stats::list2string(T*, long) */
    if constexpr (std::is_integral<T>::value) { /* (the density plots list two numbers per read: operator<< costs 0.2 s per million reads) */
        string out;
        out.resize(v.size() * 21 + 1);
        char* p = &out[0];
        for (size_t i = 0; i < v.size(); i++) {
            p = std::to_chars(p, p + 21, (long long)v[i]).ptr;
            if (i + 1 < v.size()) *p++ = ',';
        }
        out.resize((size_t)(p - out.data()));
        return out;
    }
    stringstream ss;
    for (size_t i = 0; i < v.size(); i++) {
        ss << v[i];
        if (i + 1 < v.size()) ss << ",";
    }
    return ss.str();
}

/* One side (before / after filtering) of the report: Stats after merge() + summarize() + calcLengthHistogram(). */
struct Side {
    StatsBlock s;
    StatsSummary sm;
    const ReadLists* lists;
    int threads;
    bool is_rna;
    string label; /* "Before filtering" / "After filtering" */
    long q5 = 0, q7 = 0, q10 = 0, q15 = 0, q20 = 0, q30 = 0, q40 = 0;
    int min_len = 0, max_len = 0, median_len = 0, n50_len = 0;
    long base_contents[8] = {0};
    vector<long> xs; /* sampled cycle coordinates of the two curve plots */

    Side(const int64_t* st, uint32_t C, const ReadLists* l, int threads_, bool rna, const string& lab)
        : s{st, C}, sm(summarize(st, C)), lists(l), threads(threads_), is_rna(rna), label(lab) {
        /* Stats::summarize, src/stats.cpp:176-202: the Q totals come from the base-quality histogram */
        auto span = [&](int lo, int hi) {
            long t = 0;
            for (int c = lo; c < hi; c++) t += s.base_qual_hist(c + 33);
            return t;
        };
        q40 = span(40, 127 - 33);
        q30 = q40 + span(30, 40);
        q20 = q30 + span(20, 30);
        q15 = q20 + span(15, 20);
        q10 = q15 + span(10, 15);
        q7 = q10 + span(7, 10);
        q5 = q7 + span(5, 7);
        for (int b = 0; b < 8; b++)
            for (int c = 0; c < sm.cycles; c++) base_contents[b] += s.cyc(c, 0, b);
        length_stats();
        sample_cycles();
    }

    /* Stats::calcLengthHistogram, src/stats.cpp:377-409, run lengths of the sorted list standing in for the
       map<int,int>; `first * second` is an int product there and is kept one */
    void length_stats() {
        const vector<int32_t>& len = lists->len;
        if (len.empty()) return;
        int32_t lo = len[0], hi = len[0];
        for (int32_t x : len) {
            lo = min(lo, x);
            hi = max(hi, x);
        }
        min_len = lo;
        max_len = hi;
        long totalBase = 0;
        int readnum = 0;
        const size_t n = len.size();
        auto step = [&](int first, int second) { /* one (length, count) pair of the map, ascending; true = done */
            totalBase += (int)((unsigned)first * (unsigned)second);
            if (n50_len == 0 && totalBase > sm.length_sum / 2) n50_len = first;
            readnum += second;
            if (median_len == 0 && (size_t)readnum > n / 2) median_len = first;
            return median_len > 0 && n50_len > 0;
        };
        if (lo >= 0 && (uint64_t)hi - (uint64_t)lo < (64u << 20)) {
            /* a histogram by length instead of sorting a million lengths (0.08 s per side) */
            vector<uint32_t> hist((size_t)(hi - lo) + 1, 0u);
            for (int32_t x : len) hist[(size_t)(x - lo)]++;
            for (size_t k = 0; k < hist.size(); k++)
                if (hist[k] && step((int)(lo + (int32_t)k), (int)hist[k])) break;
            return;
        }
        vector<int32_t> v(len);
        sort(v.begin(), v.end());
        for (size_t i = 0; i < n;) {
            size_t j = i;
            while (j < n && v[j] == v[i]) j++;
            if (step(v[i], (int)(j - i))) break;
            i = j;
        }
    }

    bool long_reads() const { return sm.cycles > 300; } /* Stats::isLongRead */

    /* the x coordinates of reportHtmlQuality / reportHtmlContents, src/stats.cpp:868-899: every cycle up to 300
       cycles, else the first 40 and then a geometric walk (x 1.05) that ends on the last cycle */
    void sample_cycles() {
        const int cycles = sm.cycles;
        if (!long_reads()) {
            for (int i = 0; i < cycles; i++) xs.push_back(i + 1);
            return;
        }
        const int fullSampling = 40;
        for (int i = 0; i < fullSampling && i < cycles; i++) xs.push_back(i + 1);
        double pos = fullSampling;
        for (;;) {
            pos *= 1.05;
            if (pos >= cycles) break;
            xs.push_back((int)pos);
        }
        if (xs.back() != cycles) xs.push_back(cycles);
    }

    /* Stats::list2string(double*, long, long*): the mean of the curve over (x[i-1], x[i]] */
    template <class F>
    string binned(F curve) const {
        stringstream ss;
        for (size_t i = 0; i < xs.size(); i++) {
            const long start = i > 0 ? xs[i - 1] : 0, end = xs[i];
            double total = 0.0;
            for (long k = start; k < end; k++) total += curve((int)k);
            if (end == start) ss << "0.0";
            else ss << total / (end - start);
            if (i + 1 < xs.size()) ss << ",";
        }
        return ss.str();
    }

    double mean_qual(int c) const { return (double)s.total_qual(c) / (double)s.total_base(c); }
    double base_qual(int cls, int c) const { /* Stats::summarize, src/stats.cpp:222-226 */
        const long n = s.cyc(c, 0, cls);
        return n == 0 ? mean_qual(c) : (double)s.cyc(c, 1, cls) / (double)n;
    }
    double base_content(int cls, int c) const { return (double)s.cyc(c, 0, cls) / (double)s.total_base(c); }
    double gc_content(int c) const {
        return (double)(s.cyc(c, 0, 'G' & 0x07) + s.cyc(c, 0, 'C' & 0x07)) / (double)s.total_base(c);
    }

    /* ---- the six blocks, in page order ---- */

    void basic(ostream& o) const { /* Stats::reportHtmlBasicInfo, src/stats.cpp:719-745 */
        o << "<div class='subsection_title'>" << label << ": Basic statistics</div>\n<table>\n";
        const long bases = sm.bases;
        row(o, "total reads:", human(sm.reads));
        row(o, "total bases:", human(bases));
        row(o, "minimum length:", human(min_len));
        row(o, "maximum length:", human(max_len));
        row(o, "median length:", human(median_len));
        row(o, "mean length:", human(sm.mean_length()));
        row(o, "N50 length:", human(n50_len));
        row(o, "GC content:", percents(sm.gc, bases) + "%");
        const struct {
            const char* key;
            long v;
        } q[7] = {{"Q5", q5}, {"Q7", q7}, {"Q10", q10}, {"Q15", q15}, {"Q20", q20}, {"Q30", q30}, {"Q40", q40}};
        for (auto& e : q) row(o, string(e.key) + " bases:", human(e.v) + " (" + percents(e.v, bases) + "%)");
        o << "</table>\n";
    }

    void median_hist(ostream& o) const { /* Stats::reporHtmlMedianQualHist, src/stats.cpp:589-668 */
        const string subsection = label + ": Read median quality statistics";
        const string plot = "plot_median_qual_hist_" + div_name(subsection, false);
        o << "<div class='subsection_title'>" << subsection << "</div>\n";
        const int64_t* hist = s.st + FPL_ST_MEDIAN_HIST(s.C);
        const int64_t* bases = s.st + FPL_ST_MEDIAN_BASES(s.C);
        int minVal = 0, maxVal = 0;
        for (int i = 0; i < 127 - 33; i++) {
            if (bases[i + 33] != 0) break;
            minVal++;
        }
        for (int i = 127 - 33; i >= 0; i--)
            if (bases[i + 33] > 0) {
                maxVal = i;
                break;
            }
        const int offset = max(0, minVal - 1);
        const int total = max(0, min(127 - 33, maxVal - minVal + 2));
        vector<long> x(total);
        vector<double> pr(total), pb(total);
        for (int i = 0; i < total; i++) {
            x[i] = i + offset;
            pr[i] = (double)hist[i + offset + 33] * 100.0 / (double)sm.reads;
            pb[i] = (double)bases[i + offset + 33] * 100.0 / (double)sm.bases;
        }
        o << "<div id='mean_qual_length_histogram_figure'>\n<div class='figure' id='" << plot
          << "' style='height:400px;'></div>\n</div>\n";
        o << "\n<script type=\"text/javascript\">" << endl;
        const string xt = joined(x);
        o << "var readNum={x:[" << xt << "],y:[" << joined(pr)
          << "],name: '% reads',type:'bar',line:{color:'rgba(128,0,128,1.0)', width:1}\n};\n";
        o << "var baseNum={x:[" << xt << "],y:[" << joined(pb)
          << "],name: '% accumulated bases',type:'bar',line:{color:'rgba(128,128,0,1.0)', width:1}\n};\n";
        o << "var data = [readNum, baseNum];;\n";
        o << "var layout={legend: {x: 0, y: 1.0},title:'Read median quality distribution', xaxis:{title:'read median quality "
             "score'}, yaxis:{title:'Percent (%)'}};\n";
        o << "Plotly.newPlot('" << plot << "', data, layout);\n";
        o << "</script>" << endl;
    }

    /* Stats::reporHtmlMedianQualLengthDensity, src/stats.cpp:670-716: one point per read, grouped by median quality
       (the map's key), inside a group worker by worker (Stats::merge) and in input order inside a worker.  The
       reference sizes the arrays by mReads but fills one slot per NON-EMPTY read: with empty reads in the input
       its tail is uninitialised memory; zeros here. */
    void density(ostream& o) const {
        const string subsection = label + ": Density plot of read median quality and read length";
        const string plot = "plot_median_qual_length_density_" + div_name(subsection, false);
        o << "<div class='subsection_title'>" << subsection << "</div>\n";
        const size_t n = lists->len.size();
        vector<uint64_t> start((size_t)128 * threads + 1, 0);
        auto group = [&](size_t i) { return (size_t)(lists->median[i] & 127) * threads + min<int>(lists->worker[i], threads - 1); };
        size_t filled = 0;
        for (size_t i = 0; i < n; i++)
            if (lists->len[i] > 0) start[group(i) + 1]++, filled++;
        for (size_t g = 0; g + 1 < start.size(); g++) start[g + 1] += start[g];
        vector<short> x((size_t)sm.reads, 0);
        vector<int> y((size_t)sm.reads, 0);
        for (size_t i = 0; i < n; i++)
            if (lists->len[i] > 0) {
                const uint64_t at = start[group(i)]++;
                if (at < x.size()) x[at] = (short)((char)lists->median[i] - 33), y[at] = lists->len[i];
            }
        (void)filled;
        o << "<div id='mean_qual_length_density_figure'>\n<div class='figure' id='" << plot
          << "' style='height:400px;'></div>\n</div>\n";
        o << "\n<script type=\"text/javascript\">" << endl;
        string xt, yt; /* (a number per read each: put together side by side) */
        if (x.size() > 100000) {
            thread ty([&]() { yt = joined(y); });
            xt = joined(x);
            ty.join();
        } else {
            xt = joined(x);
            yt = joined(y);
        }
        o << "var density={x:[" << xt << "],y:[" << yt
          << "],name: '% reads',type:'histogram2dcontour',line:{color:'rgba(128,0,128,1.0)', width:1}\n};\n";
        o << "var data = [density];\n";
        o << "var layout={legend: {x: 0, y: 1.0},title:' Density plot of read median quality and read length', "
             "xaxis:{title:'read median quality score'}, yaxis:{title:'Read length', type:'log'}};\n";
        o << "Plotly.newPlot('" << plot << "', data, layout);\n";
        o << "</script>" << endl;
    }

    void curves_head(ostream& o, const string& subsection, const string& name) const {
        o << "<div class='subsection_title'>" << subsection << "</div>\n";
        o << "<div id='" << name << "'>\n";
        o << "<div class='sub_section_tips'>Value of each position will be shown on mouse over.</div>\n";
        o << "<div class='figure' id='plot_" << name << "'></div>\n</div>\n";
        o << "\n<script type=\"text/javascript\">" << endl;
    }
    void curves_tail(ostream& o, const string& name, const char* ytitle) const {
        o << "];\nvar layout={title:'', xaxis:{title:'position'" << (long_reads() ? ",type:'log'" : "") << "}, yaxis:{title:'"
          << ytitle << "'}};\n";
        o << "Plotly.newPlot('plot_" << name << "', data, layout);\n";
        o << "</script>" << endl;
    }

    void quality(ostream& o) const { /* Stats::reportHtmlQuality, src/stats.cpp:847-925 */
        const string subsection = label + ": quality", name = div_name(subsection, true);
        curves_head(o, subsection, name);
        const string names[5] = {"A", is_rna ? "U" : "T", "C", "G", "mean"};
        const char* colors[5] = {"rgba(128,128,0,1.0)", "rgba(128,0,128,1.0)", "rgba(0,255,0,1.0)", "rgba(0,0,255,1.0)",
                                 "rgba(20,20,20,1.0)"};
        const string xt = joined(xs);
        o << "var data=[";
        for (int b = 0; b < 5; b++) {
            const int cls = names[b][0] & 0x07;
            o << "{x:[" << xt << "],y:["
              << (b == 4 ? binned([&](int c) { return mean_qual(c); }) : binned([&](int c) { return base_qual(cls, c); }))
              << "],name: '" << names[b] << "',mode:'lines',line:{color:'" << colors[b] << "', width:1}\n},";
        }
        curves_tail(o, name, "quality");
    }

    void contents(ostream& o) const { /* Stats::reportHtmlContents, src/stats.cpp:927-1011 */
        const string subsection = label + ": base contents", name = div_name(subsection, true);
        curves_head(o, subsection, name);
        const string names[6] = {"A", is_rna ? "U" : "T", "C", "G", "N", "GC"};
        const char* colors[6] = {"rgba(128,128,0,1.0)", "rgba(128,0,128,1.0)", "rgba(0,255,0,1.0)", "rgba(0,0,255,1.0)",
                                 "rgba(255, 0, 0, 1.0)", "rgba(20,20,20,1.0)"};
        const string xt = joined(xs);
        o << "var data=[";
        for (int b = 0; b < 6; b++) {
            const int cls = names[b][0] & 0x07;
            const long count = b == 5 ? base_contents['G' & 0x07] + base_contents['C' & 0x07] : base_contents[cls];
            string percentage = to_string((double)count * 100.0 / sm.bases);
            if (percentage.length() > 5) percentage = percentage.substr(0, 5);
            o << "{x:[" << xt << "],y:["
              << (b == 5 ? binned([&](int c) { return gc_content(c); }) : binned([&](int c) { return base_content(cls, c); }))
              << "],name: '" << names[b] << "(" << percentage << "%)',mode:'lines',line:{color:'" << colors[b] << "', width:1}\n},";
        }
        curves_tail(o, name, "base content ratios");
    }

    /* Stats::reportHtmlKMER + makeKmerTD, src/stats.cpp:747-824: 64 x 16 cells shaded by count / mean count */
    void kmers(ostream& o) const {
        const string subsection = label + ": KMER counting";
        o << "<div class='subsection_title'>" << subsection << "</div>\n";
        o << "<div  id='" << div_name(subsection, true) << "'>\n";
        o << "<div class='sub_section_tips'>Darker background means larger counts. The count will be shown on mouse over.</div>\n";
        o << "<table class='kmer_table' style='width:680px;'>\n<tr><td></td>";
        for (int h = 0; h < 16; h++) o << "<td style='color:#333333'>" << detail::kmer2(h, is_rna) << "</td>";
        o << "</tr>\n";
        const double meanBases = (double)(sm.bases + 1) / 2048; /* mKmerBufLen = 2 << (KMER_LEN * 2) */
        for (int i = 0; i < 64; i++) {
            const string first = detail::kmer3(i, is_rna);
            o << "<tr><td style='color:#333333'>" << first << "</td>";
            for (int j = 0; j < 16; j++) {
                const long val = s.kmer((i << 4) + j);
                const string kmer = first + detail::kmer2(j, is_rna);
                const double prop = val / meanBases;
                int r, g, b;
                if (prop <= 0.3) {
                    const double frac = prop * 2.0;
                    b = 255 - 256 * frac;
                    g = 255 * frac;
                    r = b * frac;
                } else if (prop > 3.0) {
                    const double frac = 2.0 / prop;
                    r = 255 - 128 * frac;
                    g = 128 * frac;
                    b = r * frac;
                } else {
                    r = g = b = 196;
                }
                stringstream ss;
                ss << "<td style='background:#";
                for (int ch : {r, g, b}) {
                    if (ch < 16) ss << "0";
                    ss << hex << ch;
                }
                ss << dec << "' title='" << kmer << ": " << val << "\n" << prop << " times as mean value'>" << kmer << "</td>";
                o << ss.str();
            }
            o << "</tr>\n";
        }
        o << "</table>\n</div>\n";
    }
};

struct Section {
    const char* id;
    const char* title;
    void (Side::*render)(ostream&) const;
};
const Section SECTIONS[] = { /* HtmlReporter::report, src/htmlreporter.cpp:80-173 */
    {"basic_stat", "Basic statistics", &Side::basic},
    {"median_qual_stat", "Median qual histogram", &Side::median_hist},
    {"median_qual_length_density", "Median qual length density", &Side::density},
    {"quality_stat", "Quality statistics", &Side::quality},
    {"contents_stat", "Base contents statistics", &Side::contents},
    {"kmer_stat", "k-mer statistics", &Side::kmers},
};

}  // namespace

bool write_html(const string& path, const ReportInputs& in, const HtmlInputs& h) {
    ofstream o;
    o.open(path, ifstream::out);
    if (!o.is_open()) return false;
    const uint32_t C = in.C;
    const int threads = max(1, min(16, h.threads));
    /* the two sides (length statistics, curve sampling) are prepared side by side (rendering the sections into strings on
       several threads and writing those was slower than streaming them: the density lists are 8 MB each) */
    std::unique_ptr<Side> sides[2];
    parallel_run(2, [&](int i) {
        sides[i].reset(i == 0 ? new Side(in.counters + FPL_OFF_PRE(C), C, &h.pre, threads, in.is_rna, "Before filtering")
                              : new Side(in.counters + FPL_OFF_POST(C), C, &h.post, threads, in.is_rna, "After filtering"));
    });
    const Side& pre = *sides[0];
    const Side& post = *sides[1];
    const int64_t* fr = in.counters + FPL_OFF_FR(C);
    const string stamp = h.timestamp.empty() ? now_text() : h.timestamp;

    /* printHeader */
    o << "<html><head><meta http-equiv=\"content-type\" content=\"text/html;charset=utf-8\" />";
    o << "<title>fastplong report at " << stamp << " </title>";
    o << PAGE_SCRIPTS;
    o << "<style type=\"text/css\">" << endl;
    for (const char* rule : PAGE_STYLE) o << rule << endl;
    o << "</style>" << endl;
    o << "</head><body><div id='container'>";

    /* printSummary + FilterResult::reportHtml (src/filterresult.cpp:227-242) */
    o << endl;
    o << "<h3 style='text-align:left;'><a href='" << PROJECT_URL << "' target='_blank' style='color:#663355;text-decoration:none;'>"
      << h.title << "</a><a href='" << PROJECT_URL << "' target='_blank' style='font-size:-2;text-decoration:none;'>(fastplong version v"
      << VERSION << ")</a></h3>" << endl;
    o << "<div class='section_div'>\n";
    o << "<div class='section_title' onclick=showOrHide('summary')><a name='summary'>Summary</a> </div>\n";
    o << "<div id='summary'>\n<div class='subsection_title'>Filtering result</div>\n<div id='filtering_result'>\n";
    {
        const double total = (double)pre.sm.reads;
        auto line = [&](const char* key, int code) {
            const long n = fr[FPL_FR_FILTER + code];
            row(o, key, human(n) + " (" + to_string(n * 100.0 / total) + "%)");
        };
        o << "<table class='summary_table'>\n";
        line("reads passed filters:", FPL_PASS_FILTER);
        line("reads with low quality:", FPL_FAIL_QUALITY);
        line("reads with too many N:", FPL_FAIL_N_BASE);
        if (in.length_filter) {
            line("reads too short:", FPL_FAIL_LENGTH);
            if (in.max_length > 0) line("reads too long:", FPL_FAIL_TOO_LONG);
        }
        if (in.complexity) line("reads with low complexity:", FPL_FAIL_COMPLEXITY);
        o << "</table>\n";
    }
    o << "</div>\n</div>\n</div>\n";

    /* the density plots list two numbers per read: the second side's is put together beside everything in front of it */
    string densityPost;
    thread densityPostMaker([&]() {
        ostringstream t;
        post.density(t);
        densityPost = t.str();
    });
    for (const Section& sec : SECTIONS) {
        o << "<div class='section_div'>\n";
        o << "<div class='section_title' onclick=showOrHide('" << sec.id << "')><a name='summary'>" << sec.title << "</a></div>\n";
        o << "<table id='" << sec.id << "' class='section_table'>\n<tr><td>\n";
        (pre.*sec.render)(o);
        o << "</td><td>\n";
        if (sec.render == &Side::density) {
            densityPostMaker.join();
            o << densityPost;
        } else {
            (post.*sec.render)(o);
        }
        o << "</td></tr>\n</table>\n</div>\n";
    }

    /* printFooter */
    o << "\n</div>" << endl;
    o << "<div id='footer'> <p>" << in.command << "</p>fastplong " << VERSION << ", at " << stamp << " </div></body></html>";
    return o.good();
}

}  // namespace fplh

extern "C" int fplh_write_html(const char* path, const int64_t* counters, uint32_t C, int n_adapters,
                               const char* const* adapters, const int* adapter_lens, int adapter_enabled, int polyx,
                               int complexity, int is_rna, int length_filter, int max_length, const char* command, int threads,
                               const char* title, const char* timestamp, uint64_t n_pre, const uint32_t* pre_read,
                               const int32_t* pre_len, const uint8_t* pre_median, uint64_t n_post, const uint32_t* post_read,
                               const int32_t* post_len, const uint8_t* post_median) {
    if (!path || !counters) return -1;
    fplh::ReportInputs in;
    in.counters = counters;
    in.C = C;
    for (int i = 0; i < n_adapters; i++) in.adapters.emplace_back(adapters[i] ? adapters[i] : "", (size_t)adapter_lens[i]);
    in.adapter_enabled = adapter_enabled != 0;
    in.polyx = polyx != 0;
    in.complexity = complexity != 0;
    in.is_rna = is_rna != 0;
    in.length_filter = length_filter != 0;
    in.max_length = max_length;
    in.command = command ? command : "";
    fplh::HtmlInputs h;
    h.threads = std::max(1, std::min(16, threads));
    h.title = title ? title : "fastplong report";
    h.timestamp = timestamp ? timestamp : "";
    for (uint64_t i = 0; i < n_pre; i++) h.pre.add(fplh::ReadLists::worker_of(pre_read[i], h.threads), pre_len[i], pre_median[i]);
    for (uint64_t i = 0; i < n_post; i++)
        h.post.add(fplh::ReadLists::worker_of(post_read[i], h.threads), post_len[i], post_median[i]);
    return fplh::write_html(path, in, h) ? 0 : -2;
}
