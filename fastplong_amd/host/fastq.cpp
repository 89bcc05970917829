#include "fastq.h"

#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <iostream>

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

FastqReader::FastqReader(const string& path) {
    /* gzopen reads plain files transparently */
    fp_ = path == "/dev/stdin" ? (void*)gzdopen(0, "rb") : (void*)gzopen(path.c_str(), "rb");
    if (fp_) gzbuffer((gzFile)fp_, 1 << 20);
    buf_.resize(8 << 20); /* FQ_BUF_SIZE of the reference is 8 MiB as well */
}

FastqReader::~FastqReader() {
    if (fp_) gzclose((gzFile)fp_);
}

bool FastqReader::refill() {
    if (eof_ || !fp_) return false;
    int n = gzread((gzFile)fp_, buf_.data(), (unsigned)buf_.size());
    pos_ = 0;
    len_ = n > 0 ? (size_t)n : 0;
    if (n <= 0) eof_ = true;
    return n > 0;
}

/* one line without its terminator; false at end of input with nothing read */
bool FastqReader::getline(string& line) {
    line.clear();
    bool any = false;
    for (;;) {
        if (pos_ >= len_ && !refill()) return any;
        any = true;
        size_t e = pos_;
        while (e < len_ && buf_[e] != '\r' && buf_[e] != '\n') e++;
        line.append(buf_.data() + pos_, e - pos_);
        if (e < len_) {
            const char term = buf_[e];
            pos_ = e + 1;
            if (term == '\r') { /* swallow the '\n' of "\r\n" */
                if (pos_ >= len_) refill();
                if (pos_ < len_ && buf_[pos_] == '\n') pos_++;
            }
            return true;
        }
        pos_ = len_;
    }
}

uint32_t FastqReader::fill(Batch& b, uint64_t max_bases, uint32_t max_reads) {
    if (b.off.empty()) {
        b.off.push_back(0);
        b.name_off.push_back(0);
    }
    uint32_t added = 0;
    string name, seq, strand, qual;
    while (!malformed_ && b.seq.size() < max_bases && b.n() < max_reads) {
        bool got = getline(name);
        while (got && (name.empty() || name[0] != '@')) got = getline(name); /* src/fastqreader.cpp:316-319 */
        if (!got) break;
        getline(seq);
        getline(strand);
        getline(qual);
        if (strand.empty() || strand[0] != '+') {
            cerr << name << endl << "Expected '+', got " << strand << endl
                 << "Your FASTQ may be invalid, please check the tail of your FASTQ file" << endl;
            malformed_ = true;
            break;
        }
        if (qual.length() != seq.length()) {
            cerr << "ERROR: sequence and quality have different length:" << endl << name << endl << seq << endl
                 << strand << endl << qual << endl
                 << "Your FASTQ may be invalid, please check the tail of your FASTQ file" << endl;
            malformed_ = true;
            break;
        }
        b.seq.insert(b.seq.end(), seq.begin(), seq.end());
        b.qual.insert(b.qual.end(), qual.begin(), qual.end());
        b.off.push_back(b.seq.size());
        b.text.insert(b.text.end(), name.begin(), name.end());
        b.text.insert(b.text.end(), strand.begin(), strand.end());
        b.name_off.push_back(b.text.size());
        b.name_len.push_back((uint32_t)name.size());
        b.strand_len.push_back((uint32_t)strand.size());
        added++;
    }
    return added;
}

void format_batch(const Batch& b, const fpl_read_result* res, string& out, string* failed) {
    static const char* prefix[3] = {"", "split-by-adapter-left-", "split-by-adapter-right-"}; /* src/read.cpp:199,208 */
    const uint32_t n = b.n();
    for (uint32_t i = 0; i < n; i++) {
        const fpl_read_result& r = res[i];
        if (r.dropped) continue;
        const char* name = b.text.data() + b.name_off[i];
        const uint32_t nl = b.name_len[i], sl = b.strand_len[i];
        const char* strand = name + nl;
        const uint8_t* s = b.seq.data() + b.off[i];
        const uint8_t* q = b.qual.data() + b.off[i];
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
void fplh_free(void* p) { free(p); }
}
