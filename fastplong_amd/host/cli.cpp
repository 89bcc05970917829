/*
 * cli.cpp -- `fastplong_amd`: fastplong's command line (flag names, short forms, defaults and
 * validation messages of reference src/main.cpp:27-103 and src/options.cpp:68-207) in front of
 * the MI355X hot path.  The host keeps what the reference's host keeps -- FASTQ reader/writer,
 * batching, report writers -- and calls fpl_process_batch() where the reference's workers call
 * SingleEndProcessor::processSingleEnd() (src/seprocessor.cpp:440).
 *
 * Batches are cut in input order and dealt round-robin to --gpus devices (one fpl_ctx and one
 * host thread per device); outputs are written back in input order; at the end the per-device
 * counter buffers are summed with one RCCL all-reduce (the replacement of Stats::merge /
 * FilterResult::merge, src/seprocessor.cpp:108-121) and rank 0's copy feeds the reports.
 *
 * --split / --split_by_lines replay what the reference's workers do with their private writers
 * (src/threadconfig.cpp:72-120) in the one writer thread: see SplitOutput.
 */
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>
#include <zlib.h>
#include <atomic>

#include <algorithm>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "fastplong_amd.h"
#include "evaluator.h"
#include "fastq.h"
#include "report.h"

using namespace std;

static void error_exit(const string& msg) { /* src/util.h:270-273 */
    cerr << "ERROR: " << msg << endl;
    /* the reference's exit(-1) status without its static destructors: a thread of this process may still be inside
       fpl_comm_init (ncclCommInitAll) or a device call when an error ends the run, and tearing the library's statics down under it
       can crash or hang at exit */
    fflush(NULL);
    _exit(255);
}

struct Flag {
    const char* name;
    char shortc;
    bool has_value;
    const char* def;
};
/* the flag table of src/main.cpp:27-103, plus --gpus / --batch_mbases of this host */
static const Flag FLAGS[] = {
    {"in", 'i', true, ""}, {"out", 'o', true, ""}, {"failed_out", 0, true, ""}, {"compression", 'z', true, "4"},
    {"stdin", 0, false, ""}, {"stdout", 0, false, ""}, {"reads_to_process", 0, true, "0"}, {"dont_overwrite", 0, false, ""},
    {"verbose", 'V', false, ""}, {"disable_adapter_trimming", 'A', false, ""}, {"start_adapter", 's', true, "auto"},
    {"end_adapter", 'e', true, "auto"}, {"adapter_fasta", 'a', true, ""}, {"distance_threshold", 'd', true, "0.25"},
    {"trimming_extension", 0, true, "10"}, {"trim_front", 'f', true, "0"}, {"trim_tail", 't', true, "0"},
    {"trim_poly_x", 'x', false, ""}, {"poly_x_min_len", 0, true, "10"}, {"cut_front", '5', false, ""},
    {"cut_tail", '3', false, ""}, {"cut_window_size", 'W', true, "4"}, {"cut_mean_quality", 'M', true, "20"},
    {"cut_front_window_size", 0, true, "4"}, {"cut_front_mean_quality", 0, true, "20"},
    {"cut_tail_window_size", 0, true, "4"}, {"cut_tail_mean_quality", 0, true, "20"}, {"mask", 'N', false, ""},
    {"mask_window_size", 0, true, "50"}, {"mask_mean_quality", 0, true, "10"}, {"break", 'b', false, ""},
    {"break_window_size", 0, true, "100"}, {"break_mean_quality", 0, true, "10"},
    {"disable_quality_filtering", 'Q', false, ""}, {"qualified_quality_phred", 'q', true, "15"},
    {"unqualified_percent_limit", 'u', true, "40"}, {"n_base_limit", 0, true, "1000000"},
    {"n_percent_limit", 'n', true, "10"}, {"mean_qual", 'm', true, "0"}, {"disable_length_filtering", 'L', false, ""},
    {"length_required", 'l', true, "20"}, {"length_limit", 0, true, "0"}, {"low_complexity_filter", 'y', false, ""},
    {"complexity_threshold", 'Y', true, "30"}, {"json", 'j', true, "fastplong.json"}, {"html", 'h', true, "fastplong.html"},
    {"report_title", 'R', true, "fastplong report"}, {"thread", 'w', true, "3"}, {"split", 0, true, "0"},
    {"split_by_lines", 0, true, "0"}, {"split_prefix_digits", 0, true, "4"},
    {"gpus", 0, true, "1"}, {"batch_mbases", 0, true, "256"}, {"batch_reads", 0, true, "0"},
    {"reader_threads", 0, true, "0"}, {"chunk_mb", 0, true, "32"}, {"gz_stream", 0, false, ""}, {"device_parse", 0, false, ""}, {"host_parse", 0, false, ""},
};

struct Args {
    map<string, string> val;
    map<string, bool> seen;
    bool exist(const string& k) const { return seen.count(k) > 0; }
    string str(const string& k) const { return val.at(k); }
    int i(const string& k) const { return atoi(val.at(k).c_str()); }
    long l(const string& k) const { return atol(val.at(k).c_str()); }
    double d(const string& k) const { return atof(val.at(k).c_str()); }
};

static Args parse(int argc, char** argv) {
    Args a;
    for (const Flag& f : FLAGS) a.val[f.name] = f.def;
    for (int i = 1; i < argc; i++) {
        string t = argv[i];
        const Flag* fl = nullptr;
        string inline_val;
        bool has_inline = false;
        if (t.rfind("--", 0) == 0) {
            string name = t.substr(2);
            size_t eq = name.find('=');
            if (eq != string::npos) {
                inline_val = name.substr(eq + 1);
                name = name.substr(0, eq);
                has_inline = true;
            }
            for (const Flag& f : FLAGS)
                if (name == f.name) fl = &f;
            if (!fl) error_exit("undefined option: --" + name);
        } else if (t.size() == 2 && t[0] == '-') {
            for (const Flag& f : FLAGS)
                if (f.shortc && t[1] == f.shortc) fl = &f;
            if (!fl) error_exit("undefined short option: " + t);
        } else {
            error_exit("unexpected argument: " + t);
        }
        a.seen[fl->name] = true;
        if (fl->has_value) {
            if (has_inline) a.val[fl->name] = inline_val;
            else {
                if (i + 1 >= argc) error_exit(string("option needs value: --") + fl->name);
                a.val[fl->name] = argv[++i];
            }
        }
    }
    return a;
}

/* Sequence::reverseComplement, src/sequence.cpp:29-77: A<->T, C<->G (either case), else N */
static string reverse_complement(const string& s) {
    string r(s.rbegin(), s.rend());
    for (char& c : r) {
        switch (c) {
            case 'A': case 'a': c = 'T'; break;
            case 'T': case 't': c = 'A'; break;
            case 'C': case 'c': c = 'G'; break;
            case 'G': case 'g': c = 'C'; break;
            default: c = 'N';
        }
    }
    return r;
}

struct Device {
    fpl_ctx* ctx = nullptr;
};

/* One batch on its way through the host pipeline:
 *   reader thread (parse into CSR) -> one thread per device (fpl_process_batch, then the output text on a few
 *   helper threads) -> the main thread (writes the pieces in input order).
 * Where the reference's workers hand strings to WriterThread (src/seprocessor.cpp:283-313), the stages here
 * hand whole batches; a small pool of Work objects bounds what is in flight. */
struct Work {
    uint64_t seq_no = 0;
    fplh::Batch batch;
    vector<fpl_read_result> res;
    fplh::FragmentList frags; /* --break / --mask */
    vector<string> outs, faileds;
    vector<struct iovec> gather; /* --out as a gather list over the batch's own arrays (plain output, see build_gather) */
    string gather_text;          /* the few bytes of it that exist nowhere yet: names with a split prefix */
    int rc = 0;
    string err;
    std::atomic<int> holders{0}; /* --split*: the per-worker writer threads that still read this batch (+ the in-order thread) */
    bool verdict_done = false;   /* --device_parse: this batch's verdict is published (a chunk that came back from the host's reader is submitted a second time) */
};
/* The passing reads of a batch as they go to --out (Read::appendToString, src/read.cpp:119-143), NOT copied together: every
 * line is a slice of what the batch already holds -- names and '+' lines in Batch::text, bases and qualities in the
 * page-locked arrays -- so the writer hands the kernel a gather list (writev) instead of a second copy of the data.
 * Formatting 18 GB of output text was 4.5 of the pipeline's 9 CPU-seconds, and the CPU quota is what bounds it.
 * (Plain --out only: gzip members, --failed_out, --split* and --break / --mask output go through format_batch_parallel.) */
static void build_gather(const fplh::Batch& b, const fpl_read_result* res, vector<struct iovec>& iov, string& text,
                         uint32_t first = 0, uint32_t last = ~0u) {
    static const char* prefix[3] = {"", "split-by-adapter-left-", "split-by-adapter-right-"}; /* src/read.cpp:199,208 */
    static const char nl_byte = '\n';
    const uint32_t n = min(last, b.n());
    iov.clear();
    text.clear();
    size_t need = 0; /* bytes of prefixed names: reserved up front, the list points into the string */
    for (uint32_t i = first; i < n; i++) {
        const fpl_read_result& r = res[i];
        if (r.dropped) continue;
        for (int f = 0; f < r.n_frag; f++)
            if (r.code[f] == FPL_PASS_FILTER && r.kind[f] >= 1 && r.kind[f] <= 2 && b.name_len[i] > 0)
                need += b.name_len[i] + strlen(prefix[r.kind[f]]);
    }
    text.reserve(need + 1);
    auto put = [&](const void* p, size_t len) {
        struct iovec v;
        v.iov_base = const_cast<void*>(p);
        v.iov_len = len;
        iov.push_back(v);
    };
    for (uint32_t i = first; i < n; i++) {
        const fpl_read_result& r = res[i];
        if (r.dropped) continue;
        const char* name = b.name_ptr(i);
        const uint32_t nl = b.name_len[i], sl = b.strand_len[i];
        const char* strand = b.strand_ptr(i);
        const uint8_t* sq = b.seq_ptr(i);
        const uint8_t* ql = b.qual_ptr(i);
        for (int f = 0; f < r.n_frag; f++) {
            if (r.code[f] != FPL_PASS_FILTER) continue;
            const char* pf = prefix[r.kind[f] <= 2 ? r.kind[f] : 0];
            if (*pf && nl > 0) { /* name->insert(1, prefix) */
                const size_t at = text.size();
                text.append(name, 1);
                text.append(pf);
                text.append(name + 1, nl - 1);
                put(text.data() + at, text.size() - at);
            } else {
                put(name, nl);
            }
            put(&nl_byte, 1);
            put(sq + r.frag_start[f], r.frag_len[f]);
            put(&nl_byte, 1);
            put(strand, sl);
            put(&nl_byte, 1);
            put(ql + r.frag_start[f], r.frag_len[f]);
            put(&nl_byte, 1);
        }
    }
}
/* all of a gather list to fd (writev takes 1024 entries and about 2 GiB at a time, and may stop short) */
static bool write_gather(int fd, vector<struct iovec>& iov) {
    size_t k = 0;
    while (k < iov.size()) {
        const int cnt = (int)min<size_t>(1024, iov.size() - k);
        ssize_t w = writev(fd, iov.data() + k, cnt);
        if (w < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        while (w > 0 && k < iov.size()) { /* skip what went out, trim the entry it stopped in */
            if ((size_t)w >= iov[k].iov_len) {
                w -= (ssize_t)iov[k].iov_len;
                k++;
            } else {
                iov[k].iov_base = (char*)iov[k].iov_base + w;
                iov[k].iov_len -= (size_t)w;
                w = 0;
            }
        }
        while (k < iov.size() && iov[k].iov_len == 0) k++;
    }
    return true;
}

template <class T>
class Channel {
   public:
    void push(T v) {
        { lock_guard<mutex> g(m_); q_.push_back(v); }
        cv_.notify_one();
    }
    T pop() { /* blocks */
        unique_lock<mutex> g(m_);
        cv_.wait(g, [&] { return !q_.empty(); });
        T v = q_.front();
        q_.pop_front();
        return v;
    }
    bool try_pop(T& v) {
        lock_guard<mutex> g(m_);
        if (q_.empty()) return false;
        v = q_.front();
        q_.pop_front();
        return true;
    }
   private:
    mutex m_;
    condition_variable cv_;
    deque<T> q_;
};

#include "split.h"
using fplh::gzip_into;
using fplh::gzip_member;
using fplh::SplitOutput;

int main(int argc, char* argv[]) {
    /* (before any other thread exists: a context drives five streams, the runtime's default is four hardware queues per device and
       two streams on one queue run in submission order -- fpl_create asks for the same, but setenv belongs where one thread runs) */
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    if (argc == 1) {
        cerr << "fastplong_amd: fastplong's per-read hot path on MI355X" << endl << "version 0.4.1-compatible" << endl;
        return 0;
    }
    if (argc == 2 && (strcmp(argv[1], "-v") == 0 || strcmp(argv[1], "--version") == 0)) {
        cout << "fastplong 0.4.1" << endl;
        return 0;
    }
    Args cmd = parse(argc, argv);

    string in = cmd.str("in"), out = cmd.str("out"), failedOut = cmd.str("failed_out");
    const bool fromStdin = cmd.exist("stdin"), toStdout = cmd.exist("stdout");
    const int readsToProcess = cmd.i("reads_to_process");
    if (fromStdin) in = "/dev/stdin";

    fpl_options o;
    fpl_options_default(&o);
    o.adapter_enabled = !cmd.exist("disable_adapter_trimming");
    string startAd = cmd.str("start_adapter"), endAd = cmd.str("end_adapter");
    o.ed_max = cmd.d("distance_threshold");
    o.trimming_extension = cmd.i("trimming_extension");
    if (startAd != "auto" && endAd == "auto") endAd = reverse_complement(startAd); /* src/main.cpp:138-140 */
    vector<string> fasta;
    if (!cmd.str("adapter_fasta").empty()) {
        string err;
        if (!fplh::load_fasta_adapters(cmd.str("adapter_fasta"), fasta, &cerr, err)) error_exit(err);
    }
    o.trim_front = cmd.i("trim_front");
    o.trim_tail = cmd.i("trim_tail");
    o.polyx = cmd.exist("trim_poly_x");
    o.polyx_min_len = cmd.i("poly_x_min_len");
    o.cut_front = cmd.exist("cut_front");
    o.cut_tail = cmd.exist("cut_tail");
    const int wShared = cmd.i("cut_window_size"), qShared = cmd.i("cut_mean_quality");
    o.cut_front_window = cmd.exist("cut_front_window_size") ? cmd.i("cut_front_window_size") : wShared;
    o.cut_front_quality = cmd.exist("cut_front_mean_quality") ? cmd.i("cut_front_mean_quality") : qShared;
    o.cut_tail_window = cmd.exist("cut_tail_window_size") ? cmd.i("cut_tail_window_size") : wShared;
    o.cut_tail_quality = cmd.exist("cut_tail_mean_quality") ? cmd.i("cut_tail_mean_quality") : qShared;
    if (!o.cut_front && !o.cut_tail &&
        (cmd.exist("cut_window_size") || cmd.exist("cut_mean_quality") || cmd.exist("cut_front_window_size") ||
         cmd.exist("cut_front_mean_quality") || cmd.exist("cut_tail_window_size") || cmd.exist("cut_tail_mean_quality")))
        cerr << "WARNING: you specified the options for cutting by quality, but forgot to enable any of "
                "cut_front/cut_tail/cut_right. This will have no effect." << endl;
    o.qual_filter = !cmd.exist("disable_quality_filtering");
    o.qualified_qual = 33 + cmd.i("qualified_quality_phred"); /* num2qual */
    o.unqualified_percent_limit = cmd.i("unqualified_percent_limit");
    o.avg_qual_req = cmd.i("mean_qual");
    o.n_base_percent_limit = cmd.i("n_percent_limit");
    o.n_base_limit = cmd.i("n_base_limit");
    o.length_filter = !cmd.exist("disable_length_filtering");
    o.required_length = cmd.i("length_required");
    o.max_length = cmd.i("length_limit");
    o.complexity_filter = cmd.exist("low_complexity_filter");
    o.complexity_percent = min(100, max(0, cmd.i("complexity_threshold")));
    o.mask_enabled = cmd.exist("mask"); /* src/main.cpp:207-215 */
    o.mask_window = cmd.i("mask_window_size");
    o.mask_quality = cmd.i("mask_mean_quality");
    o.break_enabled = cmd.exist("break");
    o.break_window = cmd.i("break_window_size");
    o.break_quality = cmd.i("break_mean_quality");
    if ((o.mask_enabled && o.mask_window <= 0) || (o.break_enabled && o.break_window <= 0))
        error_exit("the window size of --mask / --break must be positive");
    const bool fragmentMode = o.mask_enabled || o.break_enabled;
    /* src/main.cpp:225-250 */
    const bool splitEnabled = cmd.exist("split") || cmd.exist("split_by_lines");
    const int splitDigits = cmd.i("split_prefix_digits");
    int splitNumber = 0;
    long splitSize = 0;
    bool splitByNumber = false, splitByLines = false;
    if (cmd.exist("split") && cmd.exist("split_by_lines"))
        error_exit("You cannot set both splitting by file number (--split) and splitting by file lines (--split_by_lines), please choose either.");
    if (cmd.exist("split")) {
        splitNumber = cmd.i("split");
        splitByNumber = true;
    }
    if (cmd.exist("split_by_lines")) {
        const long lines = cmd.l("split_by_lines");
        if (lines % 4 != 0) error_exit("Line number (--split_by_lines) should be a multiple of 4");
        splitSize = lines / 4; /* 4 lines per record */
        splitByLines = true;
    }
    if ((fromStdin || in == "/dev/stdin") && splitByNumber) error_exit("Splitting by file number is not supported in STDIN mode");
    const string jsonFile = cmd.str("json"), htmlFile = cmd.str("html");
    int workers = cmd.i("thread"); /* Options::validate, src/options.cpp:120-125: the HTML report's point order and --split see it */
    if (workers < 1) workers = 1;
    else if (workers > 16) {
        cerr << "WARNING: fastp uses up to 16 threads although you specified " << workers << endl;
        workers = 16;
    }
    const int nGpus = max(1, cmd.i("gpus"));
    const uint64_t batchBases = (uint64_t)max(1L, cmd.l("batch_mbases")) * 1000000ull;
    const uint32_t batchReads = cmd.l("batch_reads") > 0 ? (uint32_t)cmd.l("batch_reads") : 0x3FFFFFFFu;

    stringstream ss; /* src/main.cpp:252-256 */
    for (int i = 0; i < argc; i++) ss << argv[i] << " ";
    const string command = ss.str();
    time_t t1 = time(NULL);

    /* Options::validate, src/options.cpp:68-207 (the checks that concern this path) */
    if (in.empty()) error_exit("read input should be specified by --in, or enable --stdin if you want to read STDIN");
    if (toStdout && !out.empty()) {
        cerr << "In STDOUT mode, ignore the output filename " << out << endl;
        out = "";
    }
    { /* --dont_overwrite, src/options.cpp:90-112 */
        const bool keep = cmd.exist("dont_overwrite");
        auto exists = [](const string& f) { return !f.empty() && access(f.c_str(), F_OK) == 0; };
        const string why = " already exists and you have set to not rewrite output files by --dont_overwrite";
        if (keep && exists(out)) error_exit(out + why);
        if (keep && exists(failedOut)) error_exit(failedOut + why);
        if (!failedOut.empty() && failedOut == out) error_exit("--failed_out and --out shouldn't have same file name");
        if (keep && exists(cmd.str("json"))) error_exit(cmd.str("json") + why);
        if (keep && exists(cmd.str("html"))) error_exit(cmd.str("html") + why);
    }
    if (toStdout && splitEnabled) error_exit("splitting mode cannot work with stdout mode");
    if (splitEnabled) { /* src/options.cpp:151-168 */
        if (splitDigits < 0 || splitDigits > 10)
            error_exit("you have enabled splitting output to multiple files, the digits number of file name prefix (--split_prefix_digits) should be 0 ~ 10.");
        if (splitByNumber) {
            if (splitNumber < 2 || splitNumber >= 1000)
                error_exit("you have enabled splitting output by file number, the number of files (--split) should be 2 ~ 999.");
            if (workers > splitNumber) workers = splitNumber; /* thread number cannot be more than the number of file to split */
        }
        if (splitByLines && splitSize < 1000 / 4)
            error_exit("you have enabled splitting output by file lines, the file lines (--split_by_lines) should be >= 1000.");
    }
    if (readsToProcess < 0) error_exit("the number of reads to process (--reads_to_process) cannot be negative");
    if (o.trim_front < 0) error_exit("trim_front1 (--trim_front1) should be >0, suggest 0 ~ 100");
    if (o.trim_tail < 0) error_exit("trim_tail1 (--trim_tail1) should be >0, suggest 0 ~ 100");
    if (o.qualified_qual - 33 < 0 || o.qualified_qual - 33 > 93)
        error_exit("qualitified phred (--qualified_quality_phred) should be 0 ~ 93, suggest 3 ~ 20");
    if (o.avg_qual_req < 0 || o.avg_qual_req > 93)
        error_exit("average quality score requirement (--mean_qual) should be 0 ~ 93, suggest 5 ~ 30");
    if (o.unqualified_percent_limit < 0 || o.unqualified_percent_limit > 100)
        error_exit("unqualified percent limit (--unqualified_percent_limit) should be 0 ~ 100, suggest 20 ~ 60");
    if (o.n_base_percent_limit < 0 || o.n_base_percent_limit > 100)
        error_exit("N base percent limit (--n_percent_limit) should be 0 ~ 100, suggest 5 ~ 20");
    if (o.n_base_limit < 0 || o.n_base_limit > 1000000) error_exit("N base number limit (--n_base_limit) should be 0 ~ 1000000");
    if (o.required_length < 0) error_exit("length requirement (--length_required) should be >0, suggest >50");
    if (o.cut_front || o.cut_tail) {
        if (wShared < 1 || wShared > 1000) error_exit("the sliding window size for cutting by quality (--cut_window_size) should be between 1~1000.");
        if (qShared < 1 || qShared > 30) error_exit("the mean quality requirement for cutting by quality (--cut_mean_quality) should be 1 ~ 30, suggest 15 ~ 20.");
        if (o.cut_front_window < 1 || o.cut_front_window > 1000) error_exit("the sliding window size for cutting by quality (--cut_front_window_size) should be between 1~1000.");
        if (o.cut_front_quality < 1 || o.cut_front_quality > 30) error_exit("the mean quality requirement for cutting by quality (--cut_front_mean_quality) should be 1 ~ 30, suggest 15 ~ 20.");
        if (o.cut_tail_window < 1 || o.cut_tail_window > 1000) error_exit("the sliding window size for cutting by quality (--cut_tail_window_size) should be between 1~1000.");
        if (o.cut_tail_quality < 1 || o.cut_tail_quality > 30) error_exit("the mean quality requirement for cutting by quality (--cut_tail_mean_quality) should be 1 ~ 30, suggest 13 ~ 20.");
    }
    if (startAd != "auto" && !startAd.empty()) {
        if (startAd.length() <= 3) error_exit("the sequence of <adapter_sequence> should be longer than 3");
        for (char c : startAd)
            if (c != 'A' && c != 'T' && c != 'C' && c != 'G')
                error_exit("the adapter <adapter_sequence> can only have bases in {A, T, C, G}, but the given sequence is: " + startAd);
    }
    if (o.ed_max < 0 || o.ed_max > 1.0) error_exit("the adapter <distance_threshold> should be 0.0 ~ 1.0, suggest 0.1 ~ 0.3");
    if (o.trimming_extension < 0 || o.trimming_extension > 100) error_exit("the adapter <trimming_extension> should be 0 ~ 100, suggest 5 ~ 30");

    auto clk = []() { return chrono::duration<double>(chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tMain = clk();
    auto since_launch = [&]() -> double { /* measurement hook: FPLH_T0 = the launcher's time.time() */
        const char* e = getenv("FPLH_T0");
        return e ? chrono::duration<double>(chrono::system_clock::now().time_since_epoch()).count() - atof(e) : -1.0;
    };
    const double launchToMain = since_launch();
    /* Evaluator::evaluateSeqLenAndCheckRNA, src/evaluator.cpp:16-61: U vs T in the first 100 reads */
    bool isRNA = false;
    if (!fromStdin && in != "/dev/stdin") {
        fplh::FastqReader ev(in);
        if (!ev.ok()) error_exit("Failed to open file: " + in);
        fplh::Batch b;
        ev.fill(b, ~0ull, 100);
        long numT = 0, numU = 0;
        for (uint8_t c : b.seq) {
            numT += c == 'T';
            numU += c == 'U';
        }
        if (numT > 0 && numU > 0) error_exit("This data contains both U and T");
        if (numU > 0) {
            isRNA = true;
            cerr << "RNA direct sequencing data" << endl;
        }
    }
    /* adapter auto-detection, src/main.cpp:270-277 (an undetected "auto" stays literal, as in the reference) */
    long readNum = 0;
    if (o.adapter_enabled && (startAd == "auto" || endAd == "auto")) {
        if (fromStdin || in == "/dev/stdin") cerr << "Adapter auto-detection is disabled for STDIN mode" << endl;
        else {
            /* counting, seed and growth of the detection run on the first device (fpl_pick_adapter; device 0: the first of
               --gpus); the host keeps the verdict.  (FPLH_HOST_KMERS: everything on the host -- test / measurement hook) */
            if (!getenv("FPLH_HOST_KMERS"))
                fplh::set_adapter_picker([&](const uint8_t* sq, const uint64_t* of, uint32_t n, int side, int shift, bool rna,
                                             fplh::AdapterVerdict& v) {
                    fpl_adapter_pick p;
                    if (fpl_pick_adapter(0, sq, of, n, side, shift, rna ? 1 : 0, &p) != FPL_OK) return false;
                    v.key = p.key;
                    v.count = p.count;
                    v.total_key = p.total_key;
                    v.total = p.total;
                    v.adapter.assign(p.seq, (size_t)(p.len > 0 ? p.len : 0));
                    return true;
                });
            fplh::detect_adapters(in, o.trim_tail, isRNA, startAd, endAd, &readNum);
            cerr << endl;
        }
    }
    if (splitByNumber) { /* src/main.cpp:282-293: the evaluator's guess of the read count decides the file size */
        if (readNum == 0) readNum = fplh::evaluate_read_num(in);
        splitSize = readNum / splitNumber;
        if (splitSize <= 0) { /* one record per file at least */
            splitSize = 1;
            cerr << "WARNING: the input file has less reads than the number of files to split" << endl;
        }
    }

    /* one context + one host thread per device */
    const double tEval = clk();
    vector<fpl_adapter> fa(fasta.size());
    for (size_t i = 0; i < fasta.size(); i++) fa[i] = fpl_adapter{fasta[i].data(), (int32_t)fasta[i].size()};
    vector<Device> dev(nGpus);
    {
        /* (a context costs a tenth of a second -- streams, events, tables, the device's first allocations: the devices' contexts
           are made side by side, a node's eight in the time of one) */
        vector<int> rcs((size_t)nGpus, FPL_OK);
        auto make = [&](int d) {
            rcs[(size_t)d] = fpl_create(&dev[(size_t)d].ctx, &o, startAd.data(), (int32_t)startAd.size(), endAd.data(), (int32_t)endAd.size(),
                                        fa.data(), (int32_t)fa.size(), d, 65536);
        };
        vector<thread> makers;
        for (int d = 1; d < nGpus; d++) makers.emplace_back(make, d);
        make(0);
        for (auto& t : makers) t.join();
        for (int d = 0; d < nGpus; d++) {
            if (rcs[(size_t)d] == FPL_ERR_NO_DEVICE)
                error_exit("fastplong_amd needs " + to_string(nGpus) + " HIP device(s); there is no CPU path");
            if (rcs[(size_t)d] != FPL_OK) error_exit(string("fpl_create: ") + fpl_strerror(rcs[(size_t)d]));
        }
    }

    /* the communicators of the closing merge, made while the batches run (a thread of its own: ncclCommInitAll over several devices
       takes longer than many a run's whole pipeline; a failure here is not one yet -- the merge then makes its own and reports) */
    thread commMaker;
    {
        vector<fpl_ctx*> ctxs;
        for (auto& D : dev) ctxs.push_back(D.ctx);
        /* FPL_NO_COMM_PREINIT=1: no thread here, the merge makes the communicators itself (the round-3 order) */
        if ((nGpus > 1 || getenv("FPL_RCCL_FORCE")) && !getenv("FPL_NO_COMM_PREINIT")) commMaker = thread([ctxs]() mutable { (void)fpl_comm_init(ctxs.data(), (int32_t)ctxs.size()); });
    }
    const double tCreate = clk();
    if (cmd.exist("verbose"))
        cerr << "start-up: input evaluation " << tEval - tMain << " s, device contexts " << tCreate - tEval << " s" << endl;
    /* the CSR arrays of every batch are page-locked (fpl_host_alloc), so the DMA engines read them in place */
    if (!getenv("FPLH_NO_PIN")) /* (measurement hook: pageable batches, the runtime stages the copies) */
        fplh::ByteBuf::set_allocator(fpl_host_alloc, fpl_host_free);
    const int hw = max(1, fplh::effective_cpus()); /* (what the scheduler lets this process use: affinity and cgroup quota) */
    /* How the input is read.  A regular uncompressed file is cut into chunks of --chunk_mb that --reader_threads
       workers parse at the same time (FastqReader::parse_chunk: each worker reads its chunk from the page cache,
       locates the records and copies their lines into a page-locked batch); the sequencer below puts the chunks
       back in order and checks every chunk's guessed start against its predecessor.  Everything else -- gzip,
       pipes, --reads_to_process -- goes through the one sequential reader. */
    uint64_t chunkBytes = (uint64_t)max(1L, cmd.l("chunk_mb")) << 20;
    if (const char* e = getenv("FPLH_CHUNK_BYTES")) /* test hook: tiny chunks put every cut inside some record */
        if (atol(e) > 0) chunkBytes = (uint64_t)atol(e);
    int chunkFd = -1;
    const char* chunkMem = nullptr; /* the input's text in memory (a mapping of the file / inflated gzip members) instead of a descriptor */
    bool chunkMemMapped = false;    /* ... a file mapping: its pages go back to the kernel as the reader passes them */
    uint64_t chunkFileSize = 0;
    if (!fromStdin && in != "/dev/stdin" && readsToProcess == 0 && !getenv("FPLH_NO_CHUNKS")) {
        const int fd = open(in.c_str(), O_RDONLY);
        struct stat st;
        unsigned char magic[2] = {0, 0};
        if (fd >= 0 && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0 && pread(fd, magic, 2, 0) == 2 &&
            !(magic[0] == 0x1f && magic[1] == 0x8b)) {
            chunkFd = fd;
            chunkFileSize = (uint64_t)st.st_size;
            /* the parsers take the file's bytes in place from a mapping (18 GB of page cache: pipeline 0.76 -> 0.58 s
               against pread into per-thread windows, and no first-touch penalty on a file this process has not read before);
               the pages are handed back as the sequencer passes them.  (FPLH_NO_MMAP_INPUT: measurement hook.)  A file cut
               short under the mapping raises SIGBUS where pread would have returned an error: same message, same exit code */
            if (!getenv("FPLH_NO_MMAP_INPUT") && chunkFileSize > chunkBytes) { /* (a file of one chunk goes through the sequential reader) */
                void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
                if (m != MAP_FAILED) {
                    madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
                    chunkMem = (const char*)m;
                    chunkMemMapped = true;
                    struct sigaction sa;
                    memset(&sa, 0, sizeof(sa));
                    sa.sa_handler = [](int) {
                        static const char msg[] = "ERROR: reading the input failed (file truncated while it was being read?)\n";
                        ssize_t r = write(2, msg, sizeof(msg) - 1);
                        (void)r;
                        _exit(1);
                    };
                    sigaction(SIGBUS, &sa, nullptr);
                }
            }
        } else if (fd >= 0) {
            const bool gz_file = S_ISREG(st.st_mode) && magic[0] == 0x1f && magic[1] == 0x8b;
            close(fd);
            /* a gzip file made of several members (bgzip, a `cat` of per-chunk files, what fastp / fastplong / this host
               write): the members are inflated side by side into anonymous memory, which the chunk parsers then take like
               a mapped file.  One deflate stream, or more text than a third of the machine's memory: the sequential
               reader and its stream.  (FPLH_NO_GZ_EXPAND: measurement / test hook) */
            /* The whole text sits in memory until the parsers have passed it: it may take what the process can still get --
               MemAvailable and the cgroup's limit, whichever is smaller -- less the page-locked arena and the batches in
               flight (2 GiB), and of that no more than half; anything larger is streamed.  --gz_stream (or FPLH_NO_GZ_EXPAND)
               forces the stream. */
            if (gz_file && !cmd.exist("gz_stream") && !getenv("FPLH_NO_GZ_EXPAND") && (uint64_t)st.st_size > chunkBytes / 8) { /* (small inputs: the stream) */
                const double t0 = clk();
                const uint64_t budget = fplh::memory_budget(), hold = 2ull << 30;
                const uint64_t cap = budget > hold ? (budget - hold) / 2 : 0;
                uint64_t sz = 0, reserved = 0;
                chunkMem = cap ? fplh::gunzip_members_to_memory(in, max(4, min(64, hw)), cap, &sz, &reserved) : nullptr;
                if (chunkMem && sz <= chunkBytes) { /* one chunk of text: not worth the parsers */
                    munmap((void*)chunkMem, (size_t)reserved);
                    chunkMem = nullptr;
                }
                if (!chunkMem && cmd.exist("verbose"))
                    cerr << "input: gzip text not expanded in memory (" << clk() - t0 << " s spent finding out): the sequential reader streams it" << endl;
                if (chunkMem) { /* (the mapping lives until the process ends) */
                    chunkFileSize = sz;
                    if (cmd.exist("verbose"))
                        cerr << "input: gzip members inflated into memory: " << sz << " bytes of text in " << clk() - t0 << " s" << endl;
                }
            }
        }
    }
    int readerThreads = cmd.i("reader_threads");
    /* (half of the CPUs parse, the rest formats, copies and writes; sixteen parsers feed one device's PCIe link with room to
       spare -- 4.7 GB/s of text each -- so several devices get sixteen each, as far as the CPUs go) */
    if (readerThreads <= 0) readerThreads = max(2, min(16 * nGpus, hw / 2));
    const bool chunked = (chunkFd >= 0 || chunkMem) && chunkFileSize > chunkBytes;
    /* the chunk parsers cut the batches: one per --chunk_mb of text; --batch_mbases / --batch_reads only size the batches
       of the sequential reader (pipes, streamed gzip, --reads_to_process) */
    if (chunked && (cmd.exist("batch_mbases") || cmd.exist("batch_reads")))
        cerr << "WARNING: --batch_mbases / --batch_reads do not apply to this input: its batches are the chunks of --chunk_mb ("
             << (chunkBytes >> 20) << " MB of text each); lower --chunk_mb for smaller batches" << endl;
    fplh::FastqReader* reader = nullptr;
    /* Work objects bound what is in flight: one per parser, FPL_MAX_IN_FLIGHT per device in the copy / kernel stage,
       one per device being formatted, two waiting for the writer */
    auto gz_name = [](const string& p) { return p.size() > 3 && p.compare(p.size() - 3, 3, ".gz") == 0; };
    const bool gzOut = !splitEnabled && (gz_name(out) || gz_name(failedOut)); /* (then up to four batches are formatted at a time) */
    /* (+ FPLH_EXTRA_WORK, default 6: with exactly as many as the stages can hold, a parser waits for a Work object while the writer
       or a formatter still holds one, and the device thread finds its queue empty -- the link then idles between two uploads) */
    const int extraWork = getenv("FPLH_EXTRA_WORK") ? atoi(getenv("FPLH_EXTRA_WORK")) : 6;
    const int nWork = (chunked ? readerThreads : 1) + (FPL_MAX_IN_FLIGHT + 1) * nGpus + 2 + (gzOut ? 3 : 0) + (chunked ? max(0, extraWork) : 0);
    /* --device_parse: the chunk parsers only LOAD the file's bytes (page-locked), the device finds the records
       (fpl_process_text_async); --break / --mask keep the host's reader (their fragment lists come back batch by batch through
       the CSR entry points), and so do inputs that are not cut into chunks (pipes, a streamed gzip, a small file) */
    /* (the default wherever it applies; --host_parse keeps the host's parsers, --device_parse only says so out loud) */
    if (cmd.exist("device_parse") && cmd.exist("host_parse")) error_exit("--device_parse and --host_parse exclude each other");
    const bool textMode = !cmd.exist("host_parse") && !getenv("FPLH_HOST_PARSE") && chunked && !cmd.exist("break") && !cmd.exist("mask") &&
                          chunkBytes < (3ull << 30);
    if (cmd.exist("device_parse") && !textMode && cmd.exist("verbose"))
        cerr << "input: --device_parse does not apply (it needs an uncompressed file or multi-member gzip cut into chunks, --chunk_mb below 3072, no --break / --mask): the host parses" << endl;
    if (chunked && textMode) /* one block holds a chunk's text and the stretch behind it that the last record may run into */
        fplh::ByteBuf::set_arena((size_t)(chunkBytes + (5u << 20)), (size_t)nWork);
    else if (chunked) /* a chunk holds about half its bytes in bases: one block each for the bases and the qualities of a batch */
        fplh::ByteBuf::set_arena((size_t)(chunkBytes / 2 + chunkBytes / 16 + (2u << 20)), 2 * (size_t)nWork);
    if (!chunked) {
        reader = new fplh::FastqReader(in);
        if (!reader->ok()) error_exit("Failed to open file: " + in);
        /* threads of the reader's refill / locate / copy phases (FPLH_PARSE_THREADS overrides) */
        const char* e = getenv("FPLH_PARSE_THREADS");
        reader->set_copy_threads(e && atoi(e) > 0 ? atoi(e) : max(1, min(8, hw / 2)));
    }
    /* Outputs are plain files; a name ending in .gz gets gzip members (-z level), one per formatted slice,
       deflated on the formatter threads and concatenated by the writer: any gzip reader takes that as one stream */
    struct OutFile {
        FILE* f = nullptr;
        bool gz = false;
        bool wrote = false;
        /* FPLH_PARALLEL_WRITE (measurement hook): the pieces of a batch written side by side at their offsets (pwrite from
           the worker pool; nothing goes through the FILE's buffer then).  Measured on the GPU box into tmpfs: no gain -- one
           thread copies into the page cache at 5.5-6 GB/s, fifteen pwrite()s side by side, or fifteen memcpy()s into a mapped
           window of the file, fill it at the same 5.5-6 GB/s: what bounds a single output file is the kernel's insertion of
           fresh pages into that file's page cache, not the copy.  So the plain path stays the default. */
        bool positional = false;
        uint64_t pos = 0;
        explicit operator bool() const { return f != nullptr; }
    };
    auto open_out = [](const string& path) -> OutFile {
        OutFile o;
        if (path.empty()) return o;
        o.gz = path.size() > 3 && path.compare(path.size() - 3, 3, ".gz") == 0;
        o.f = fopen(path.c_str(), "wb");
        if (!o.f) error_exit("Failed to write: " + path);
        struct stat st;
        o.positional = getenv("FPLH_PARALLEL_WRITE") && fstat(fileno(o.f), &st) == 0 && S_ISREG(st.st_mode);
        return o;
    };
    /* with --split* the reference never calls initOutput (src/seprocessor.cpp:65-67): no single --out file and no
       --failed_out either; the workers' private writers take the passing reads */
    OutFile fout = open_out(splitEnabled ? string() : out), ffail = open_out(splitEnabled ? string() : failedOut);
    if (toStdout) fout.f = stdout, fout.gz = false, fout.positional = false;
    const int gzLevel = min(9, max(1, cmd.i("compression")));
    SplitOutput* split = splitEnabled ? new SplitOutput(out, splitDigits, workers, splitByLines, splitNumber, splitSize, gzLevel) : nullptr;
    auto gzip_pieces = [&](vector<string>& pieces) { /* in parallel; pieces stay below 4 GiB (one slice of a batch) */
        fplh::parallel_run((int)pieces.size(), [&](int i) {
            if (!pieces[i].empty()) gzip_into(pieces[i], gzLevel, pieces[i]);
        });
    };
    auto write_pieces = [](OutFile& o, const vector<string>& pieces) {
        if (o.positional) { /* input order by construction: the offsets are the running sum of the pieces' sizes */
            vector<uint64_t> at(pieces.size());
            for (size_t i = 0; i < pieces.size(); i++) {
                at[i] = o.pos;
                o.pos += pieces[i].size();
                if (!pieces[i].empty()) o.wrote = true;
            }
            std::atomic<bool> bad{false};
            const int fd = fileno(o.f);
            fplh::parallel_run((int)pieces.size(), [&](int i) {
                const char* p = pieces[i].data();
                size_t left = pieces[i].size();
                uint64_t off = at[i];
                while (left > 0) {
                    const ssize_t w = pwrite(fd, p, left, (off_t)off);
                    if (w < 0 && errno == EINTR) continue;
                    if (w <= 0) {
                        bad = true;
                        return;
                    }
                    p += w;
                    off += (uint64_t)w;
                    left -= (size_t)w;
                }
            });
            if (bad) error_exit("write failed");
            return;
        }
        for (auto& piece : pieces)
            if (!piece.empty()) {
                if (fwrite(piece.data(), 1, piece.size(), o.f) != piece.size()) error_exit("write failed");
                o.wrote = true;
            }
    };

    long readsLeft = readsToProcess > 0 ? readsToProcess : -1;
    /* slices a batch's output is formatted in (one worker each); gzip outputs are deflated per slice, which is compute-
       bound, so they get more, smaller slices */
    const bool anyGz = (fout && fout.gz) || (ffail && ffail.gz);
    /* Plain --out alone that is NOT a regular file -- a pipe into an aligner or a compressor (--stdout, /dev/stdout), /dev/null --
       is written as gather lists over the batches' own arrays (build_gather): nothing is formatted.  Into a regular file the
       one writer thread's copy into the page cache is the bottleneck either way (18 GB: 2.9 s from formatted pieces, 3.6 s
       from eight small entries per read), so files keep the pieces the formatter threads put together side by side.
       FPLH_NO_GATHER / FPLH_GATHER_FILES: measurement hooks */
    bool gatherOut = fout && !fout.gz && !ffail && !fragmentMode && !split && !getenv("FPLH_NO_GATHER");
    if (gatherOut && !getenv("FPLH_GATHER_FILES")) {
        struct stat ost;
        if (fstat(fileno(fout.f), &ost) == 0 && S_ISREG(ost.st_mode)) gatherOut = false;
    }
    if (gatherOut) fflush(fout.f); /* (from here on the descriptor is written directly) */
    /* formatter stage threads: one per device -- or four when the output is deflated, each with a quarter of the helpers:
       a batch of one chunk (32 MB of text) cut into 64 members keeps 64 helpers busy for a few milliseconds between two
       thread hand-offs (measured: 25 ms per batch, 1.3 GB/s), four batches side by side in 16 members each do not wait
       for one another */
    const int nFmt = anyGz ? max(nGpus, 4) : nGpus;
    const int fmtThreads = max(1, min(anyGz ? max(8, 64 / nFmt * nGpus) : 16, hw / max(1, nGpus) - 1));
    vector<Work> pool(nWork);
    Channel<Work*> freeq, fmtq, doneq;
    vector<Channel<Work*>> devq(nGpus);
    for (auto& w : pool) freeq.push(&w);
    uint64_t nBatches = 0;
    /* --verbose: where the wall time of the host pipeline goes (busy seconds per stage) */
    auto now = []() { return chrono::duration<double>(chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tStart = now();
    double tParse = 0, tWrite = 0, tRedo = 0;
    uint64_t nRedo = 0;
    vector<double> tGpu(nGpus, 0), tFormat(nFmt, 0);
    /* --verbose, per device thread: seconds inside the submissions, seconds with nothing in flight and nothing parsed (starved),
       how often the queue was empty when there was room for another batch, and how deep the submissions found the pipeline */
    vector<double> tSubmit(nGpus, 0), tStarved(nGpus, 0);
    vector<uint64_t> nMiss(nGpus, 0), nSubmit(nGpus, 0), depthSum(nGpus, 0);
    std::atomic<uint64_t> nTextBatches{0}, nTextFallbacks{0}; /* --device_parse: chunks the device parsed / chunks handed back to the host's reader */
    /* --device_parse: the reference stops READING at a malformed record (FastqReader::read returns NULL, src/fastqreader.cpp:326-341),
       so nothing behind one may be counted -- but a chunk's verdict comes from its device, and the chunks of several devices are
       under way side by side.  Every chunk's verdict is published here (fpl_peek_text: the parse only, nothing counted yet), and a
       chunk's per-read kernels are let go (fpl_wait_text; a CSR batch: its submission) only when every chunk in front of it was
       good; a chunk behind a malformed record is dropped (fpl_cancel_text).  Waits only ever look at smaller sequence numbers. */
    struct Verdicts {
        mutex m;
        condition_variable cv;
        vector<uint8_t> v; /* 0 unknown, 1 good, 2 holds a malformed record */
        uint64_t frontier = 0, bad = ~0ull;
        string bad_text;
        void publish(uint64_t j, bool good, const string& text = string()) {
            {
                lock_guard<mutex> g(m);
                if (v.size() <= j) v.resize(j + 1, 0);
                v[j] = good ? 1 : 2;
                if (!good && j < bad) {
                    bad = j;
                    bad_text = text;
                }
                while (frontier < v.size() && v[frontier] == 1) frontier++;
            }
            cv.notify_all();
        }
        bool wait_before(uint64_t j) { /* true: a chunk in front of j holds a malformed record -- j is not part of the input */
            unique_lock<mutex> g(m);
            cv.wait(g, [&] { return frontier >= j || bad < j; });
            return bad < j;
        }
    } verdicts;
    std::atomic<bool> stopInput{false};
    string inputError; /* a malformed record: reported the way the sequential reader does, the input ends there */
    string ioError;    /* the input could not be read / decompressed to its end: the run fails (src/fastqreader.cpp:92-137) */

    /* ---- stage 1: batches in input order -> devq (round-robin over the devices) */
    thread readerThread([&]() {
        if (!chunked) {
            for (;;) {
                uint32_t maxReads = batchReads;
                if (readsLeft >= 0) maxReads = (uint32_t)min<long>(readsLeft, maxReads);
                if (maxReads == 0) break;
                Work* w = freeq.pop();
                w->batch.clear();
                const double t0 = now();
                const uint32_t got = reader->fill(w->batch, batchBases, maxReads);
                tParse += now() - t0;
                if (got == 0) {
                    freeq.push(w);
                    break;
                }
                if (readsLeft >= 0) readsLeft -= w->batch.n();
                w->seq_no = nBatches++;
                devq[w->seq_no % nGpus].push(w); /* batches are dealt round-robin in input order */
            }
            if (reader->input_error()) ioError = reader->input_error_text();
        } else {
            /* the parsers take their batches from the Work pool; next() puts the chunks back in input order */
            auto acquire = [&]() {
                fplh::ChunkedReader::Item it;
                Work* w = freeq.pop();
                w->verdict_done = false;
                it.batch = &w->batch;
                it.token = w;
                return it;
            };
            auto release = [&](fplh::ChunkedReader::Item it) { freeq.push((Work*)it.token); };
            fplh::ChunkedReader cr(chunkFd, chunkFileSize, chunkBytes, readerThreads, acquire, release, chunkMem, textMode);
            fplh::ChunkedReader::Item it;
            uint64_t unmapped = 0;
            while (cr.next(it)) {
                Work* w = (Work*)it.token;
                if (stopInput.load()) { /* (--device_parse: a device found a malformed record in an earlier chunk) */
                    freeq.push(w);
                    break;
                }
                w->seq_no = nBatches++;
                devq[w->seq_no % nGpus].push(w);
                if (chunkMem) { /* pages no parser looks at again (unmapping 18 GB at exit costs 0.2 s of process time) */
                    const uint64_t dead = cr.dead_below() & ~(uint64_t)((2u << 20) - 1);
                    if (dead > unmapped) {
                        if (chunkMemMapped) munmap((void*)(chunkMem + unmapped), (size_t)(dead - unmapped));
                        else madvise((void*)(chunkMem + unmapped), (size_t)(dead - unmapped), MADV_DONTNEED);
                        unmapped = dead;
                    }
                }
            }
            inputError = cr.malformed_text();
            ioError = cr.io_error_text();
            nRedo = cr.chunks_parsed_again();
            tRedo = cr.redo_seconds();
            tParse = cr.busiest_parser_seconds();
        }
        for (int d = 0; d < nGpus; d++) devq[d].push(nullptr);
    });
    /* ---- stage 2, one thread per device: copies and kernels, FPL_MAX_IN_FLIGHT batches deep.
       A text batch (the device parses) goes through three calls: its submission (upload + parse), its APPROVAL (the parse's
       verdict, published for the other devices' threads; then fpl_start_text: the per-read kernels) and its wait.  The loop
       approves batch k + 1 before it waits for batch k, so the device's queue holds the next batch's kernels while this thread
       sits in the wait, and the thread's own time in the runtime (some thirty calls per batch) overlaps the device's. */
    vector<thread> devThreads;
    for (int d = 0; d < nGpus; d++)
        devThreads.emplace_back([&, d]() {
            enum { TEXT_PENDING, TEXT_STARTED, TEXT_DROPPED, TEXT_HANDED_BACK, CSR };
            struct Flight {
                Work* w;
                int state;
            };
            deque<Flight> inflight; /* in the order of submission: the library's slots are a FIFO */
            deque<Work*> redo;      /* chunks the host's reader took: in again as CSR batches */
            bool open = true;
            fpl_ctx* const ctx = dev[d].ctx;
            auto fail = [&](Work* w, int rc) {
                w->rc = rc;
                if (w->err.empty()) w->err = string(fpl_strerror(rc)) + " " + fpl_last_error(ctx);
            };
            auto make_empty = [](Work* w) { /* an empty batch, as the reader makes them */
                w->batch.clear();
                w->batch.off.push_back(0);
                w->batch.name_off.push_back(0);
                w->res.clear();
            };
            /* the oldest text batch that is still pending: verdict first, then its kernels -- or not.  false: none is pending */
            auto approve_next = [&]() -> bool {
                Flight* f = nullptr;
                for (auto& x : inflight)
                    if (x.state == TEXT_PENDING) {
                        f = &x;
                        break;
                    }
                if (!f) return false;
                const double t0 = now();
                Work* w = f->w;
                fplh::Batch& b = w->batch;
                fpl_text_result tr;
                bool good = true, to_csr = false;
                string bad_text;
                int rc = fpl_peek_text(ctx, &tr);
                if (rc == FPL_OK && tr.status != FPL_TEXT_OK) {
                    /* irregular text (blank lines, a lone \r, no line break at the end, a record the reference would stop at):
                       nothing of it was counted -- the host's reader takes the chunk, by the reference's rules */
                    rc = fpl_cancel_text(ctx);
                    fplh::FastqReader::ChunkInfo ci;
                    vector<char> window;
                    const uint64_t len = b.raw_len;
                    const char* base = (const char*)b.raw.data() + b.raw_begin;
                    b.text_backed = false;
                    fplh::FastqReader::parse_chunk(-1, len, 0, len, true, window, b, ci, 1, base);
                    nTextFallbacks++;
                    to_csr = true;
                    if (ci.status == 3) { /* the input ends at this record, as with the host's reader; what the chunk holds in front of it counts */
                        good = false;
                        bad_text = ci.err;
                        stopInput = true;
                    }
                }
                verdicts.publish(w->seq_no, good, bad_text);
                const bool drop = verdicts.wait_before(w->seq_no);
                if (rc != FPL_OK) { /* (the run fails with this batch's error) */
                    if (!to_csr) (void)fpl_cancel_text(ctx);
                    fail(w, rc);
                    f->state = TEXT_DROPPED;
                } else if (drop) { /* behind a malformed record: not part of the input */
                    if (!to_csr) rc = fpl_cancel_text(ctx);
                    make_empty(w);
                    if (rc != FPL_OK) fail(w, rc);
                    f->state = TEXT_DROPPED;
                } else if (to_csr) {
                    if (b.n() > 0) { /* in again, as a CSR batch; the cancelled slot stays in the FIFO until its turn */
                        w->res.resize(b.n());
                        w->verdict_done = true;
                        redo.push_back(w);
                        f->w = nullptr;
                        f->state = TEXT_HANDED_BACK;
                    } else {
                        make_empty(w);
                        f->state = TEXT_DROPPED;
                    }
                } else {
                    rc = fpl_start_text(ctx);
                    if (rc != FPL_OK) fail(w, rc);
                    f->state = TEXT_STARTED;
                }
                tGpu[d] += now() - t0;
                return true;
            };
            auto finish_oldest = [&]() {
                if (inflight.front().state == TEXT_PENDING) approve_next(); /* (the oldest pending batch is this one) */
                const Flight f = inflight.front();
                inflight.pop_front();
                Work* w = f.w;
                const double t0 = now();
                if (f.state == CSR) {
                    if (w->rc == FPL_OK) {
                        const int rc = fpl_wait(ctx);
                        if (rc != FPL_OK) fail(w, rc);
                    }
                    if (w->rc == FPL_OK && fragmentMode) { /* any number of output reads per read: fetch the list */
                        uint32_t nf = 0, nr = 0;
                        int rc = fpl_fragment_counts(ctx, &nf, &nr);
                        if (rc == FPL_OK) {
                            w->frags.frags.resize(nf);
                            w->frags.regs.resize(nr);
                            rc = fpl_get_fragments(ctx, w->frags.frags.data(), nf, w->frags.regs.data(), nr);
                            w->frags.index(w->batch.n());
                        }
                        if (rc != FPL_OK) fail(w, rc);
                    }
                } else {
                    fpl_text_result tr;
                    const fpl_read_result* rr = nullptr;
                    const uint32_t* ls = nullptr;
                    const int rc = fpl_wait_text(ctx, &tr, &rr, &ls);
                    if (f.state == TEXT_STARTED && w->rc == FPL_OK) {
                        if (rc != FPL_OK) fail(w, rc);
                        else if (tr.status != FPL_TEXT_OK) fail(w, FPL_ERR_STATE); /* (the verdict was "good") */
                        else {
                            w->res.assign(rr, rr + tr.n_reads);
                            w->batch.adopt_lines(ls, tr.n_reads);
                            nTextBatches++;
                        }
                    } else if (w && rc != FPL_OK && w->rc == FPL_OK) {
                        fail(w, rc);
                    }
                }
                tGpu[d] += now() - t0;
                if (w) fmtq.push(w);
            };
            const size_t depth = fragmentMode ? 1 : FPL_MAX_IN_FLIGHT;
            auto n_pending = [&]() {
                size_t n = 0;
                for (auto& x : inflight) n += x.state == TEXT_PENDING;
                return n;
            };
            while (open || !inflight.empty() || !redo.empty()) {
                /* 1. fill the pipeline: every free slot gets a batch if one is parsed (uploads queue up behind one another) */
                bool starved = false;
                while (inflight.size() < depth) {
                    Work* w = nullptr;
                    bool got = false;
                    if (!redo.empty()) {
                        w = redo.front();
                        redo.pop_front();
                        got = true;
                    } else if (open) {
                        if (inflight.empty()) {
                            const double ts = now();
                            w = devq[d].pop();
                            tStarved[d] += now() - ts;
                            got = true;
                        } else {
                            got = devq[d].try_pop(w); /* nothing parsed yet: go on with what is in flight meanwhile */
                            if (!got) nMiss[d]++;
                        }
                    }
                    if (got && !w) open = false;
                    if (!got || !w) {
                        starved = true;
                        break;
                    }
                    w->res.resize(w->batch.n());
                    w->err.clear();
                    w->rc = FPL_OK;
                    if (textMode && !w->batch.text_backed && !w->verdict_done) {
                        /* a CSR batch in a run whose chunks the device parses (a chunk the sequencer parsed itself): its kernels
                           are enqueued by the submission, so it waits for the verdicts in front of it first -- with nothing of
                           this thread in flight, whose verdicts nobody else could publish */
                        while (!inflight.empty()) finish_oldest();
                        w->verdict_done = true;
                        verdicts.publish(w->seq_no, true);
                        if (verdicts.wait_before(w->seq_no)) {
                            make_empty(w);
                            fmtq.push(w);
                            continue;
                        }
                    }
                    depthSum[d] += inflight.size() + 1;
                    const double t0 = now();
                    const bool text = w->batch.text_backed;
                    if (text)
                        w->rc = fpl_process_text_async(ctx, w->batch.raw.data() + w->batch.raw_begin, w->batch.raw_len);
                    else
                        w->rc = fpl_process_batch_async(ctx, w->batch.seq.data(), w->batch.qual.data(), w->batch.off.data(), w->batch.n(),
                                                        w->res.data());
                    tGpu[d] += now() - t0;
                    tSubmit[d] += now() - t0;
                    nSubmit[d]++;
                    if (w->rc != FPL_OK) { /* nothing was enqueued: hand the error on in order */
                        fail(w, w->rc);
                        if (textMode && text) verdicts.publish(w->seq_no, true); /* (nobody may wait for this chunk's verdict for ever) */
                        while (!inflight.empty()) finish_oldest();
                        fmtq.push(w);
                        continue;
                    }
                    inflight.push_back(Flight{w, text ? (int)TEXT_PENDING : (int)CSR});
                }
                if (inflight.empty()) continue;
                /* 2. the oldest pending batch's verdict and kernels -- while another upload is queued behind it (or nothing more is
                   to come): the wait inside is for ITS upload, and the link must not run dry meanwhile */
                /* (all but the newest pending batch: the batch this thread is about to wait for was then started an iteration ago,
                   and the next one's kernels sit in the device's queue behind its) */
                while (n_pending() >= 2) approve_next();
                if (n_pending() == 1 && starved) approve_next();
                /* 3. the oldest batch's results, when the pipeline is full or has nothing else to do */
                if (inflight.size() >= depth || starved) finish_oldest();
            }
            fmtq.push(nullptr);
        });
    /* ---- stage 3: the output text of a batch, on helper threads (the writer below only writes) */
    vector<thread> fmtStage;
    std::atomic<int> devEnded{0};
    for (int f = 0; f < nFmt; f++)
        fmtStage.emplace_back([&, f]() {
            for (;;) {
                Work* w = fmtq.pop();
                if (!w) {
                    /* one end marker per device thread; the formatter that sees the last one wakes the others */
                    if (devEnded.load() >= nGpus) break;
                    if (++devEnded == nGpus) {
                        for (int i = 0; i + 1 < nFmt; i++) fmtq.push(nullptr);
                        break;
                    }
                    continue;
                }
                const double t1 = now();
                if (w->rc == FPL_OK && !split && gatherOut) {
                    build_gather(w->batch, w->res.data(), w->gather, w->gather_text);
                } else if (w->rc == FPL_OK && !split) { /* (--split* output is cut per pack of 16 reads by the writer) */
                    fplh::format_batch_parallel(w->batch, w->res.data(), fmtThreads, w->outs, ffail ? &w->faileds : nullptr,
                                                fragmentMode ? &w->frags : nullptr);
                    if (fout && fout.gz) gzip_pieces(w->outs);
                    if (ffail && ffail.gz) gzip_pieces(w->faileds);
                }
                tFormat[f] += now() - t1;
                doneq.push(w);
            }
            doneq.push(nullptr);
        });
    fplh::HtmlInputs page; /* per-read lengths and median qualities: what Stats keeps beyond the counters */
    page.threads = workers;
    page.title = cmd.str("report_title");
    uint64_t readBase = 0;
    auto note_reads = [&](const Work& w) {
        const uint32_t n = w.batch.n();
        for (uint32_t i = 0; i < n; i++) {
            const uint8_t wk = fplh::ReadLists::worker_of(readBase + i, workers);
            const fpl_read_result& r = w.res[i];
            page.pre.add(wk, (int32_t)(w.batch.off[i + 1] - w.batch.off[i]), r.median_q_pre);
            if (!fragmentMode)
                for (int f = 0; f < r.n_frag; f++)
                    if (r.code[f] == FPL_PASS_FILTER) page.post.add(wk, (int32_t)r.frag_len[f], r.median_q_post[f]);
        }
        if (fragmentMode)
            for (const fpl_fragment& fr : w.frags.frags)
                if (fr.code == FPL_PASS_FILTER)
                    page.post.add(fplh::ReadLists::worker_of(readBase + fr.read, workers), (int32_t)fr.len, fr.median_q);
        readBase += n;
    };
    long packReads = 0, packPassed = 0; /* the pack of 16 input reads under way (it may straddle two batches) */
    /* --split*: this thread only PLANS -- which reads of the batch go to which worker's writer, and after which of them the
       worker's ThreadConfig::markProcessed is due (with what count); the workers' own threads (SplitOutput::start_threads)
       put the text together and write it, every worker into its own file.  A pack of 16 reads belongs to worker
       (index / 16) % workers (src/seprocessor.cpp:343-378); one that straddles two batches is marked with the second. */
    struct PackRange {
        uint32_t first, last;
        long mark; /* -1: the pack goes on in the next batch */
    };
    const bool splitThreads = split && !getenv("FPLH_SPLIT_ONE_THREAD"); /* (test hook: the replay on this thread) */
    if (splitThreads) split->start_threads();
    auto release = [&](Work* w) {
        if (--w->holders == 0) freeq.push(w);
    };
    auto split_reads = [&](Work* wp) { /* before note_reads: readBase is the index of the batch's first read */
        const Work& w = *wp;
        const uint32_t n = w.batch.n();
        const fplh::FragmentList* fl = fragmentMode ? &w.frags : nullptr;
        vector<vector<PackRange>> plan((size_t)workers);
        for (uint32_t i = 0; i < n;) {
            const uint64_t g = readBase + i;
            const uint32_t j = (uint32_t)min<uint64_t>(n, i + (16 - g % 16));
            const int wk = (int)((g / 16) % (uint64_t)workers);
            for (uint32_t k = i; k < j; k++) { /* `passed`, src/seprocessor.cpp:264-276: any output read of the read passes */
                bool passed = false;
                if (fl) {
                    for (uint32_t x = fl->first[k]; x < fl->first[k + 1]; x++) passed |= fl->frags[x].code == FPL_PASS_FILTER;
                } else {
                    for (int f = 0; f < w.res[k].n_frag; f++) passed |= w.res[k].code[f] == FPL_PASS_FILTER;
                }
                packPassed += passed;
            }
            packReads += j - i;
            long mark = -1;
            if ((readBase + j) % 16 == 0) { /* the pack is complete: ThreadConfig::markProcessed */
                mark = splitByLines ? packPassed : packReads;
                packReads = packPassed = 0;
            }
            plan[(size_t)wk].push_back({i, j, mark});
            i = j;
        }
        for (int wk = 0; wk < workers; wk++) {
            if (plan[(size_t)wk].empty()) continue;
            auto job = [&, wp, wk, fl, ranges = std::move(plan[(size_t)wk])]() {
                const bool gather = !fl && !split->gzipped();
                vector<struct iovec> iov;
                string text;
                for (const PackRange& r : ranges) {
                    if (gather) { /* the worker's getWriter1()->writeString(outstr), as a gather list over the batch's arrays */
                        build_gather(wp->batch, wp->res.data(), iov, text, r.first, r.last);
                        split->write_gather(wk, iov.data(), iov.size());
                    } else {
                        text.clear();
                        fplh::format_range(wp->batch, wp->res.data(), r.first, r.last, text, nullptr, fl);
                        split->write(wk, text);
                    }
                    if (r.mark >= 0) split->mark(wk, r.mark);
                }
                if (splitThreads) release(wp);
            };
            if (splitThreads) {
                wp->holders++;
                split->post(wk, std::move(job));
            } else {
                job();
            }
        }
    };
    { /* writer: this thread, in input order */
        map<uint64_t, Work*> ready;
        uint64_t next = 0;
        int live = nFmt;
        while (live > 0) {
            Work* w = doneq.pop();
            if (!w) {
                live--;
                continue;
            }
            ready[w->seq_no] = w;
            while (!ready.empty() && ready.begin()->first == next) {
                Work* r = ready.begin()->second;
                ready.erase(ready.begin());
                if (r->rc != FPL_OK) error_exit("fpl_process_batch: " + r->err);
                const double t0 = now();
                if (fout && gatherOut) {
                    if (!r->gather.empty()) {
                        if (!write_gather(fileno(fout.f), r->gather)) error_exit("write failed");
                        fout.wrote = true;
                    }
                } else if (fout) write_pieces(fout, r->outs);
                if (ffail) write_pieces(ffail, r->faileds);
                r->holders = 1; /* this thread's own hold, until note_reads is done with the batch */
                if (split) split_reads(r);
                note_reads(*r);
                tWrite += now() - t0;
                next++;
                release(r);
            }
        }
    }
    if (split) {
        if (packReads > 0) { /* the last, short pack */
            const int wk = (int)(((readBase - 1) / 16) % (uint64_t)workers);
            const long cnt = splitByLines ? packPassed : packReads;
            if (splitThreads) split->post(wk, [&, wk, cnt]() { split->mark(wk, cnt); });
            else split->mark(wk, cnt);
        }
        const double t0 = now();
        split->close(); /* (threaded: waits for the workers' writers) */
        tWrite += now() - t0;
        delete split;
    }
    readerThread.join();
    for (auto& t : devThreads) t.join();
    for (auto& t : fmtStage) t.join();
    if (inputError.empty() && verdicts.bad != ~0ull) inputError = verdicts.bad_text; /* (--device_parse: the record a device's chunk came back with) */
    if (!inputError.empty()) cerr << inputError; /* (the sequential reader printed it when it met the record) */
    if (!ioError.empty()) error_exit(ioError);
    if (cmd.exist("verbose")) {
        double g = 0, f = 0;
        for (int d = 0; d < nGpus; d++) g = max(g, tGpu[d]);
        for (int d = 0; d < nFmt; d++) f = max(f, tFormat[d]);
        cerr << "host pipeline: " << nBatches << " batches, wall " << now() - tStart << " s; busy: parse " << tParse
             << " s" << (chunked ? " (busiest of " + to_string(readerThreads) + " chunk parsers; " + to_string(nRedo) + " chunks parsed again, " + to_string(tRedo) + " s)" : string())
             << ", copies + kernels (waits) " << g << " s, format (" << fmtThreads << " threads) " << f << " s, write " << tWrite
             << " s" << endl;
    }
    if (cmd.exist("verbose"))
        for (int d = 0; d < nGpus; d++)
            cerr << "device thread " << d << ": " << nSubmit[d] << " submissions " << tSubmit[d] << " s (mean depth behind them "
                 << (nSubmit[d] ? (double)depthSum[d] / (double)nSubmit[d] : 0.0) << "), queue empty with room for a batch " << nMiss[d]
                 << " times, nothing in flight and nothing parsed " << tStarved[d] << " s" << endl;
    if (cmd.exist("verbose") && textMode)
        cerr << "device parse: " << nTextBatches.load() << " chunks parsed on the device, " << nTextFallbacks.load()
             << " handed back to the host's reader (irregular text)" << endl;
    if (cmd.exist("verbose")) { /* which kernel forms the batches took: the library picks by batch size (csrc/pipeline.h) */
        uint64_t f[6] = {0, 0, 0, 0, 0, 0};
        for (auto& D : dev) {
            uint64_t g[6] = {0, 0, 0, 0, 0, 0};
            if (fpl_get_batch_forms(D.ctx, g) == FPL_OK) {
                for (int i = 0; i < 4; i++) f[i] += g[i];
                f[4] = max(f[4], g[4]);
            }
        }
        if (f[0])
            cerr << "kernel forms: " << f[0] << " batches, mean " << f[1] / f[0] << " reads (largest " << f[4] << "); end trims: " << f[2]
                 << " through k_trim_ends_batched (64 reads per wave, from " << FPL_FORM_TRIM_BATCHED_MIN << " reads on), " << f[0] - f[2]
                 << " one wave per read; statistics: " << f[3] << " through k_stats_sorted (from " << FPL_FORM_STATS_SORTED_MIN
                 << " reads on), " << f[0] - f[3] << " through the two-update k_stats" << endl;
    }
    for (OutFile* o : {&fout, &ffail})
        if (*o) {
            if (o->gz && !o->wrote) { /* an empty .gz still has to be a gzip stream */
                const string e = gzip_member(string(), gzLevel);
                if (fwrite(e.data(), 1, e.size(), o->f) != e.size()) error_exit("write failed");
            }
            /* the buffered tail goes out here: a full disk shows up as a failing flush / close */
            if (o->f == stdout ? (fflush(stdout) != 0 || ferror(stdout)) : (fclose(o->f) != 0)) error_exit("write failed");
        }

    /* merge: agree on the per-cycle capacity, then ONE all-reduce (sum, int64) over RCCL -- behind the C-ABI */
    {
        vector<fpl_ctx*> ctxs;
        for (auto& D : dev) ctxs.push_back(D.ctx);
        const double tJ0 = now();
        const bool commMade = commMaker.joinable();
        if (commMade) commMaker.join();
        /* (what the end of the run waited for the communicators: the first use of RCCL in a process takes seconds, a short run
           is over before it is) */
        if (cmd.exist("verbose") && commMade) cerr << "counter merge: waited " << now() - tJ0 << " s for fpl_comm_init after the last batch" << endl;
        const double tM0 = now();
        const int rc = fpl_allreduce_counters(ctxs.data(), (int32_t)ctxs.size());
        if (cmd.exist("verbose") && fpl_rccl_library()[0]) cerr << "counter merge: " << now() - tM0 << " s" << endl;
        if (rc != FPL_OK) error_exit(string("fpl_allreduce_counters: ") + fpl_strerror(rc) + " " + fpl_last_error(ctxs[0]));
        if (cmd.exist("verbose") && fpl_rccl_library()[0])
            cerr << "counter merge: one all-reduce over " << ctxs.size() << " device(s), RCCL from " << fpl_rccl_library() << endl;
        if (commMade) (void)fpl_comm_init(nullptr, 0); /* the kept communicators go back before any context does */
    }
    const uint32_t C = fpl_max_cycles(dev[0].ctx);
    const size_t ncnt = fpl_counters_len(dev[0].ctx);
    vector<int64_t> counters(ncnt);
    if (fpl_get_counters(dev[0].ctx, counters.data(), ncnt) != FPL_OK) error_exit("fpl_get_counters failed");
    /* (the contexts, the page-locked arena and the HIP runtime are not torn down piece by piece: the process is about
       to end -- see the _exit at the bottom -- and unpinning a gigabyte of staging costs tenths of a second) */

    fplh::ReportInputs ri;
    ri.counters = counters.data();
    ri.C = C;
    ri.adapters.push_back(startAd);
    ri.adapters.push_back(endAd);
    for (auto& s : fasta) ri.adapters.push_back(s);
    ri.adapter_enabled = o.adapter_enabled;
    ri.polyx = o.polyx;
    ri.complexity = o.complexity_filter;
    ri.length_filter = o.length_filter;
    ri.max_length = o.max_length;
    ri.is_rna = isRNA;
    ri.command = command;
    cerr << fplh::summary_text(ri);
    const double tRep0 = now();
    /* nothing reads a batch any more: the page-locked arena is unpinned (0.09 s for 1.4 GB) while the reports are written,
       instead of by the kernel when the process exits */
    thread arenaRelease([]() { fplh::ByteBuf::release_arena(); });
    { /* the two report writers only read the counters: side by side */
        bool jsonOk = true;
        double tJson = 0;
        thread jt([&]() {
            jsonOk = fplh::write_json(jsonFile, ri);
            tJson = now() - tRep0;
        });
        const bool htmlOk = fplh::write_html(htmlFile, ri, page);
        const double tHtml = now() - tRep0;
        jt.join();
        if (!jsonOk) error_exit("Failed to write: " + jsonFile);
        if (!htmlOk) error_exit("Failed to write: " + htmlFile);
        if (cmd.exist("verbose"))
            cerr << "reports: json " << tJson << " s beside html " << tHtml << " s; since start " << now() - tStart << " s" << endl;
    }

    arenaRelease.join();
    time_t t2 = time(NULL);
    cerr << endl << "JSON report: " << jsonFile << endl;
    cerr << "HTML report: " << htmlFile << endl;
    cerr << endl << command << endl;
    cerr << "fastplong v0.4.1 (fastplong_amd), time used: " << (t2) - t1 << " seconds" << endl;
    if (cmd.exist("verbose") && launchToMain >= 0)
        cerr << "since launch: main() entered at " << launchToMain << " s, returning at " << since_launch() << " s" << endl;
    if (getenv("FPLH_TEARDOWN_TIMING")) { /* measurement hook: what the explicit teardown would cost */
        const double a = now();
        for (auto& d : dev) fpl_destroy(d.ctx);
        cerr << "teardown: contexts " << now() - a << " s" << endl;
    }
    /* every output has been written, flushed and closed above; skip the static destructors (worker pool, HIP runtime) */
    fflush(NULL);
    if (getenv("FPLH_NORMAL_EXIT")) exit(0); /* (measurement hook: a profiler's atexit handlers must run -- rocprofv3 writes its tables there) */
    _exit(0);
}
