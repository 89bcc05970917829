#include "fastq.h"

#include <fcntl.h>
#include <immintrin.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <iostream>
#include <thread>

using namespace std;

namespace fplh {

/* FAILED_TYPES, src/common.h:55-64 */
static const char* failed_type(int code) {
    switch (code) {
        case 0: return "passed";
        case 4: return "failed_polyx_filter";
        case 8: return "failed_bad_overlap";
        case 12: return "failed_too_many_n_bases";
        case 16: return "failed_too_short";
        case 17: return "failed_too_long";
        case 20: return "failed_quality_filter";
        case 24: return "failed_low_complexity";
        default: return "";
    }
}

void Batch::clear() {
    seq.clear();
    qual.clear();
    off.clear();
    text.clear();
    name_off.clear();
    name_len.clear();
    strand_len.clear();
}

/* index of the first '\n' or '\r' in p[0, n), or n: one pass for both terminators */
static size_t find_eol_sse2(const char* p, size_t n) {
    const __m128i nl = _mm_set1_epi8('\n'), cr = _mm_set1_epi8('\r');
    size_t i = 0;
    for (; i + 16 <= n; i += 16) {
        const __m128i v = _mm_loadu_si128((const __m128i*)(p + i));
        const int m = _mm_movemask_epi8(_mm_or_si128(_mm_cmpeq_epi8(v, nl), _mm_cmpeq_epi8(v, cr)));
        if (m) return i + (size_t)__builtin_ctz((unsigned)m);
    }
    for (; i < n; i++)
        if (p[i] == '\n' || p[i] == '\r') return i;
    return n;
}
__attribute__((target("avx2"))) static size_t find_eol_avx2(const char* p, size_t n) {
    const __m256i nl = _mm256_set1_epi8('\n'), cr = _mm256_set1_epi8('\r');
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
        const __m256i a = _mm256_loadu_si256((const __m256i*)(p + i)), b = _mm256_loadu_si256((const __m256i*)(p + i + 32));
        const unsigned ma = (unsigned)_mm256_movemask_epi8(_mm256_or_si256(_mm256_cmpeq_epi8(a, nl), _mm256_cmpeq_epi8(a, cr)));
        const unsigned mb = (unsigned)_mm256_movemask_epi8(_mm256_or_si256(_mm256_cmpeq_epi8(b, nl), _mm256_cmpeq_epi8(b, cr)));
        if (ma | mb) return i + (size_t)__builtin_ctzll((unsigned long long)ma | ((unsigned long long)mb << 32));
    }
    return i + find_eol_sse2(p + i, n - i);
}
static size_t find_eol(const char* p, size_t n) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    return avx2 ? find_eol_avx2(p, n) : find_eol_sse2(p, n);
}

FastqReader::FastqReader(const string& path) {
    size_t cap = 32u << 20;
    bool allow_map = true;
    if (const char* e = getenv("FPLH_READ_WINDOW")) /* test hook: tiny windows exercise the refill paths */
        if (atol(e) > 0) {
            cap = (size_t)atol(e);
            allow_map = false;
        }
    if (path != "/dev/stdin" && allow_map) { /* a regular file that is not gzip: map it, no copies into a window */
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd >= 0) {
            struct stat st;
            unsigned char magic[2] = {0, 0};
            if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0 && pread(fd, magic, 2, 0) == 2 &&
                !(magic[0] == 0x1f && magic[1] == 0x8b)) {
                void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m != MAP_FAILED) {
                    madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
                    madvise(m, (size_t)st.st_size, MADV_WILLNEED);
                    map_ = m;
                    map_len_ = (size_t)st.st_size;
                    win_ = (const char*)m;
                    len_ = map_len_;
                    eof_ = true;
                    fp_ = this;
                }
            }
            close(fd);
        }
    }
    if (!map_) {
        /* gzopen reads plain files transparently */
        fp_ = path == "/dev/stdin" ? (void*)gzdopen(0, "rb") : (void*)gzopen(path.c_str(), "rb");
        if (fp_) gzbuffer((gzFile)fp_, 1 << 20);
        buf_.resize(cap);
        win_ = buf_.data();
    }
}

FastqReader::~FastqReader() {
    if (map_) munmap(map_, map_len_);
    else if (fp_) gzclose((gzFile)fp_);
}

bool FastqReader::pull() {
    if (eof_ || !fp_ || map_) return false;
    if (pos_ > 0) {
        memmove(buf_.data(), buf_.data() + pos_, len_ - pos_);
        len_ -= pos_;
        pos_ = 0;
    } else if (len_ == buf_.size()) {
        buf_.resize(buf_.size() * 2); /* one record is larger than the window */
    }
    win_ = buf_.data();
    while (len_ < buf_.size()) { /* gzread returns short counts on pipes */
        const size_t want = min<size_t>(buf_.size() - len_, 1u << 30);
        const int n = gzread((gzFile)fp_, buf_.data() + len_, (unsigned)want);
        if (n <= 0) {
            eof_ = true;
            break;
        }
        len_ += (size_t)n;
    }
    return true;
}

/* FastqReader::getLine, src/fastqreader.cpp:219-312: a line ends at '\r' or '\n', "\r\n" counts once */
int FastqReader::scan_line(size_t& pos, Line& ln) const {
    if (pos >= len_) return eof_ ? -1 : 0;
    const char* b = win_ + pos;
    const size_t avail = len_ - pos;
    const size_t e = find_eol(b, avail);
    if (e == avail) { /* no terminator in the window */
        if (!eof_) return 0;
        ln = Line{b, avail};
        pos = len_;
        return 1;
    }
    size_t next = pos + e + 1;
    if (b[e] == '\r') { /* swallow the '\n' of "\r\n": needs the byte after it */
        if (next >= len_ && !eof_) return 0;
        if (next < len_ && win_[next] == '\n') next++;
    }
    ln = Line{b, e};
    pos = next;
    return 1;
}

/* append the located records to the batch: offsets first, then the line copies on copy_threads_ threads */
void FastqReader::copy_records(Batch& b, const vector<Rec>& recs) const {
    const size_t n0 = b.n(), nr = recs.size();
    if (nr == 0) return;
    const size_t base0 = b.seq.size(), text0 = b.text.size();
    uint64_t bases = base0, text = text0;
    b.off.reserve(n0 + nr + 1);
    for (const Rec& r : recs) {
        bases += r.seq.n;
        text += r.name.n + r.strand.n;
        b.off.push_back(bases);
        b.name_off.push_back(text);
        b.name_len.push_back((uint32_t)r.name.n);
        b.strand_len.push_back((uint32_t)r.strand.n);
    }
    b.seq.resize_uninit(bases);
    b.qual.resize_uninit(bases);
    b.text.resize(text);
    auto work = [&](size_t first, size_t last) {
        for (size_t i = first; i < last; i++) {
            const Rec& r = recs[i];
            const uint64_t o = b.off[n0 + i], t = b.name_off[n0 + i];
            memcpy(b.seq.data() + o, r.seq.p, r.seq.n);
            memcpy(b.qual.data() + o, r.qual.p, r.qual.n);
            memcpy(b.text.data() + t, r.name.p, r.name.n);
            memcpy(b.text.data() + t + r.name.n, r.strand.p, r.strand.n);
        }
    };
    const int T = (bases - base0) < (8u << 20) ? 1 : copy_threads_;
    if (T <= 1) {
        work(0, nr);
        return;
    }
    vector<std::thread> th;
    size_t first = 0;
    for (int t = 0; t < T; t++) { /* slices of about equal numbers of bases */
        const uint64_t want = base0 + (bases - base0) / T * (t + 1);
        size_t last = t == T - 1 ? nr : (size_t)(std::lower_bound(b.off.begin() + n0 + 1, b.off.begin() + n0 + 1 + nr, want) -
                                                 (b.off.begin() + n0 + 1)) + 1;
        last = min(max(last, first), nr);
        th.emplace_back(work, first, last);
        first = last;
    }
    for (auto& x : th) x.join();
}

uint32_t FastqReader::fill(Batch& b, uint64_t max_bases, uint32_t max_reads) {
    if (b.off.empty()) {
        b.off.push_back(0);
        b.name_off.push_back(0);
    }
    uint32_t added = 0;
    const Line none = {nullptr, 0};
    vector<Rec> recs;
    uint64_t bases = b.seq.size();
    uint32_t reads = b.n();
    bool end = false;
    while (!end && !malformed_ && bases < max_bases && reads < max_reads) {
        /* locate the records of the current window (mapped file: of the next stretch of it) */
        recs.clear();
        bool need_more = false;
        while (bases < max_bases && reads < max_reads) {
            /* one record = the next line that starts with '@' (src/fastqreader.cpp:316-319) and the three lines
               after it; lines missing at the end of the input read as empty, as getLine() does */
            Rec rc = {none, none, none, none};
            size_t p = pos_;
            int r;
            for (;;) {
                r = scan_line(p, rc.name);
                if (r <= 0) break;
                if (rc.name.n > 0 && rc.name.p[0] == '@') break;
                pos_ = p; /* a skipped line is consumed for good */
            }
            if (r == 0) {
                need_more = true;
                break;
            }
            if (r < 0) {
                end = true;
                break;
            }
            Line* rest[3] = {&rc.seq, &rc.strand, &rc.qual};
            for (int k = 0; k < 3 && !need_more; k++) {
                r = scan_line(p, *rest[k]);
                if (r == 0) need_more = true;
                else if (r < 0) *rest[k] = none;
            }
            if (need_more) break; /* the record continues beyond the window: restart it after reading more */
            if (rc.strand.n == 0 || rc.strand.p[0] != '+') {
                cerr << string(rc.name.p, rc.name.n) << endl
                     << "Expected '+', got " << string(rc.strand.p ? rc.strand.p : "", rc.strand.n) << endl
                     << "Your FASTQ may be invalid, please check the tail of your FASTQ file" << endl;
                malformed_ = true;
                break;
            }
            if (rc.qual.n != rc.seq.n) {
                cerr << "ERROR: sequence and quality have different length:" << endl << string(rc.name.p, rc.name.n) << endl
                     << string(rc.seq.p ? rc.seq.p : "", rc.seq.n) << endl << string(rc.strand.p, rc.strand.n) << endl
                     << string(rc.qual.p ? rc.qual.p : "", rc.qual.n) << endl
                     << "Your FASTQ may be invalid, please check the tail of your FASTQ file" << endl;
                malformed_ = true;
                break;
            }
            pos_ = p;
            recs.push_back(rc);
            bases += rc.seq.n;
            reads++;
        }
        copy_records(b, recs); /* before the window moves */
        added += (uint32_t)recs.size();
        if (need_more && !pull()) end = true;
    }
    return added;
}

void format_batch(const Batch& b, const fpl_read_result* res, string& out, string* failed) {
    format_range(b, res, 0, b.n(), out, failed);
}

void FragmentList::index(uint32_t n_reads) {
    first.assign((size_t)n_reads + 1, 0);
    for (const fpl_fragment& f : frags)
        if (f.read < n_reads) first[f.read + 1]++;
    for (uint32_t i = 0; i < n_reads; i++) first[i + 1] += first[i];
}

void format_batch_parallel(const Batch& b, const fpl_read_result* res, int threads, vector<string>& outs,
                           vector<string>* faileds, const FragmentList* fl) {
    const uint32_t n = b.n();
    if (threads < 1) threads = 1;
    outs.assign(threads, string());
    if (faileds) faileds->assign(threads, string());
    /* slices of about equal numbers of bases */
    vector<uint32_t> cut(threads + 1, n);
    cut[0] = 0;
    const uint64_t total = n ? b.off[n] : 0;
    for (int t = 1; t < threads; t++) {
        const uint64_t want = total / threads * t;
        cut[t] = (uint32_t)(std::lower_bound(b.off.begin(), b.off.begin() + n, want) - b.off.begin());
    }
    vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t]() {
            outs[t].reserve((size_t)((b.off[cut[t + 1]] - b.off[cut[t]]) * 2 + (uint64_t)(cut[t + 1] - cut[t]) * 128 + 64));
            format_range(b, res, cut[t], cut[t + 1], outs[t], faileds ? &(*faileds)[t] : nullptr, fl);
        });
    for (auto& x : th) x.join();
}

/* bases [start, start + len) of a read with the regions Read::maskRegionWithN overwrote (src/read.cpp:217-225) */
static void append_masked(string& out, const uint8_t* s, uint32_t start, uint32_t len, const fpl_region* regs, uint32_t n_regs) {
    const size_t at = out.size();
    out.append((const char*)s + start, len);
    for (uint32_t k = 0; k < n_regs; k++) {
        if (regs[k].start < start || regs[k].start - start >= len) continue;
        const uint32_t a = regs[k].start - start, l = std::min(regs[k].len, len - a);
        memset(&out[at + a], 'N', l);
    }
}

void format_range(const Batch& b, const fpl_read_result* res, uint32_t first, uint32_t last, string& out,
                  string* failed, const FragmentList* fl) {
    static const char* prefix[3] = {"", "split-by-adapter-left-", "split-by-adapter-right-"}; /* src/read.cpp:199,208 */
    for (uint32_t i = first; i < last; i++) {
        const fpl_read_result& r = res[i];
        if (r.dropped) continue;
        const char* name = b.text.data() + b.name_off[i];
        const uint32_t nl = b.name_len[i], sl = b.strand_len[i];
        const char* strand = name + nl;
        const uint8_t* s = b.seq.data() + b.off[i];
        const uint8_t* q = b.qual.data() + b.off[i];
        if (fl) { /* --break / --mask: any number of output reads, src/seprocessor.cpp:234-281 */
            const uint32_t f0 = fl->first[i], f1 = fl->first[i + 1];
            for (uint32_t k = f0; k < f1; k++) {
                const fpl_fragment& f = fl->frags[k];
                const fpl_region* rg = fl->regs.data() + f.region_first;
                if (f.code == FPL_PASS_FILTER) {
                    /* the name went through breakByGap's insert(1, "split-..") and then breakByRegions'
                       insert(1, "r<i>-") (src/read.cpp:199,208,244,256) */
                    if (nl > 0) out.append(name, 1);
                    if (f.break_no) {
                        out.push_back('r');
                        out.append(std::to_string(f.break_no));
                        out.push_back('-');
                    }
                    out.append(prefix[f.kind <= 2 ? f.kind : 0]);
                    if (nl > 1) out.append(name + 1, nl - 1);
                    out.push_back('\n');
                    append_masked(out, s, f.start, f.len, rg, f.region_count);
                    out.push_back('\n');
                    out.append(strand, sl);
                    out.push_back('\n');
                    out.append((const char*)q + f.start, f.len);
                    out.push_back('\n');
                } else if (failed && f1 - f0 == 1) {
                    /* or1 with its tag; it shows the N only when the one output read IS r1 (masked in place) */
                    const bool in_place = f.kind == 0 && f.break_no == 0;
                    failed->append(name, nl);
                    failed->push_back(' ');
                    failed->append(failed_type(f.code));
                    failed->push_back('\n');
                    append_masked(*failed, s, r.r1_start, r.r1_len, rg, in_place ? f.region_count : 0);
                    failed->push_back('\n');
                    failed->append(strand, sl);
                    failed->push_back('\n');
                    failed->append((const char*)q + r.r1_start, r.r1_len);
                    failed->push_back('\n');
                }
            }
            continue;
        }
        for (int f = 0; f < r.n_frag; f++) {
            if (r.code[f] == FPL_PASS_FILTER) { /* Read::appendToString, src/read.cpp:119-143 */
                const char* pf = prefix[r.kind[f] <= 2 ? r.kind[f] : 0];
                if (*pf && nl > 0) { /* name->insert(1, prefix) */
                    out.append(name, 1);
                    out.append(pf);
                    out.append(name + 1, nl - 1);
                } else {
                    out.append(name, nl);
                }
                out.push_back('\n');
                out.append((const char*)s + r.frag_start[f], r.frag_len[f]);
                out.push_back('\n');
                out.append(strand, sl);
                out.push_back('\n');
                out.append((const char*)q + r.frag_start[f], r.frag_len[f]);
                out.push_back('\n');
            } else if (failed && r.n_frag == 1) { /* or1->appendToStringWithTag: the trimmed r1, src/read.cpp:145-173 */
                failed->append(name, nl);
                failed->push_back(' ');
                failed->append(failed_type(r.code[f]));
                failed->push_back('\n');
                failed->append((const char*)s + r.r1_start, r.r1_len);
                failed->push_back('\n');
                failed->append(strand, sl);
                failed->push_back('\n');
                failed->append((const char*)q + r.r1_start, r.r1_len);
                failed->push_back('\n');
            }
        }
    }
}

}  // namespace fplh

extern "C" {
void* fplh_batch_read(const char* path, uint64_t max_bases, uint32_t max_reads) {
    fplh::FastqReader rd(path);
    if (!rd.ok()) return nullptr;
    fplh::Batch* b = new fplh::Batch();
    rd.fill(*b, max_bases, max_reads);
    if (b->off.empty()) {
        b->off.push_back(0);
        b->name_off.push_back(0);
    }
    return b;
}
uint32_t fplh_batch_n(void* b) { return ((fplh::Batch*)b)->n(); }
uint64_t fplh_batch_bytes(void* b) { return ((fplh::Batch*)b)->seq.size(); }
const uint8_t* fplh_batch_seq(void* b) { return ((fplh::Batch*)b)->seq.data(); }
const uint8_t* fplh_batch_qual(void* b) { return ((fplh::Batch*)b)->qual.data(); }
const uint64_t* fplh_batch_off(void* b) { return ((fplh::Batch*)b)->off.data(); }
void fplh_batch_free(void* b) { delete (fplh::Batch*)b; }
int fplh_format_batch(void* bv, const fpl_read_result* res, char** out, uint64_t* out_len, char** failed,
                      uint64_t* failed_len) {
    fplh::Batch* b = (fplh::Batch*)bv;
    std::string o, f;
    fplh::format_batch(*b, res, o, failed ? &f : nullptr);
    *out = (char*)malloc(o.size() + 1);
    memcpy(*out, o.data(), o.size());
    *out_len = o.size();
    if (failed) {
        *failed = (char*)malloc(f.size() + 1);
        memcpy(*failed, f.data(), f.size());
        *failed_len = f.size();
    }
    return 0;
}
int fplh_format_batch_fragments(void* bv, const fpl_read_result* res, const fpl_fragment* frags, uint32_t n_frags,
                                const fpl_region* regs, uint32_t n_regs, int threads, char** out, uint64_t* out_len,
                                char** failed, uint64_t* failed_len) {
    fplh::Batch* b = (fplh::Batch*)bv;
    fplh::FragmentList fl;
    fl.frags.assign(frags, frags + n_frags);
    fl.regs.assign(regs, regs + n_regs);
    fl.index(b->n());
    std::vector<std::string> o, f;
    fplh::format_batch_parallel(*b, res, threads, o, failed ? &f : nullptr, &fl);
    std::string oo, ff;
    for (auto& x : o) oo += x;
    for (auto& x : f) ff += x;
    *out = (char*)malloc(oo.size() + 1);
    memcpy(*out, oo.data(), oo.size());
    *out_len = oo.size();
    if (failed) {
        *failed = (char*)malloc(ff.size() + 1);
        memcpy(*failed, ff.data(), ff.size());
        *failed_len = ff.size();
    }
    return 0;
}
int fplh_format_batch_parallel(void* bv, const fpl_read_result* res, int threads, char** out, uint64_t* out_len,
                               char** failed, uint64_t* failed_len) {
    fplh::Batch* b = (fplh::Batch*)bv;
    std::vector<std::string> o, f;
    fplh::format_batch_parallel(*b, res, threads, o, failed ? &f : nullptr);
    std::string oo, ff;
    for (auto& x : o) oo += x;
    for (auto& x : f) ff += x;
    *out = (char*)malloc(oo.size() + 1);
    memcpy(*out, oo.data(), oo.size());
    *out_len = oo.size();
    if (failed) {
        *failed = (char*)malloc(ff.size() + 1);
        memcpy(*failed, ff.data(), ff.size());
        *failed_len = ff.size();
    }
    return 0;
}
void fplh_free(void* p) { free(p); }
}
