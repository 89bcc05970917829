#include "split.h"

#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <ctype.h>
#include <iostream>

using namespace std;

namespace fplh {

static void error_exit(const string& msg) { /* src/util.h:270-273 */
    cerr << "ERROR: " << msg << endl;
    exit(-1);
}

/* The deflated bytes go through a buffer the calling thread keeps (the pool's workers are persistent): dozens of
   threads allocating and releasing multi-megabyte strings per slice spend their time in the kernel's address-space
   lock instead. */
void gzip_into(const string& in, int level, string& out) {
    static thread_local vector<char> scratch;
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) error_exit("deflateInit2 failed");
    const size_t bound = deflateBound(&zs, (uLong)in.size()) + 64;
    if (scratch.size() < bound) scratch.resize(bound);
    zs.next_in = (Bytef*)in.data();
    zs.avail_in = (uInt)in.size();
    zs.next_out = (Bytef*)scratch.data();
    zs.avail_out = (uInt)bound;
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) error_exit("deflate failed");
    const size_t n = zs.total_out;
    deflateEnd(&zs);
    out.assign(scratch.data(), n); /* (when out is the input itself: shrinks inside its own allocation) */
}
string gzip_member(const string& in, int level) {
    string o;
    gzip_into(in, level, o);
    return o;
}

SplitOutput::SplitOutput(const string& out, int digits, int workers, bool by_lines, int number, long size, int gz_level)
    : out_(out), digits_(digits), T_(workers), by_lines_(by_lines), number_(number), size_(size), level_(gz_level), w_(workers) {
    for (int t = 0; t < T_; t++) {
        w_[t].working = t; /* mWorkingSplit = threadId */
        open(w_[t]);
    }
}
void SplitOutput::write(int t, const string& text) { /* config->getWriter1()->writeString(outstr), src/seprocessor.cpp:297-301 */
    if (out_.empty()) return;
    Worker& w = w_[t];
    w.pending += text;
    if (w.pending.size() >= (4u << 20)) flush(w);
}
void SplitOutput::mark(int t, long reads) { /* ThreadConfig::markProcessed, src/threadconfig.cpp:89-110 */
    Worker& w = w_[t];
    w.current += reads;
    if (w.current >= size_ && (by_lines_ || w.working + T_ < number_)) {
        w.working += T_;
        open(w);
        w.current = 0;
    }
}
void SplitOutput::close() { /* ThreadConfig::cleanup: files a short input never reached still have to exist */
    for (Worker& w : w_) {
        if (!by_lines_)
            while (w.working + T_ < number_) {
                w.working += T_;
                open(w);
            }
        shut(w);
    }
}
void SplitOutput::flush(Worker& w) {
    if (w.pending.empty() || !w.f) return;
    const string& bytes = w.gz ? gzip_member(w.pending, level_) : w.pending;
    if (fwrite(bytes.data(), 1, bytes.size(), w.f) != bytes.size()) error_exit("write failed");
    w.wrote = true;
    w.pending.clear();
}
void SplitOutput::shut(Worker& w) {
    if (!w.f) return;
    flush(w);
    if (w.gz && !w.wrote) {
        const string e = gzip_member(string(), level_);
        if (fwrite(e.data(), 1, e.size(), w.f) != e.size()) error_exit("write failed");
    }
    if (fclose(w.f) != 0) error_exit("write failed");
    w.f = nullptr;
}
void SplitOutput::open(Worker& w) { /* ThreadConfig::initWriterForSplit: 1-based number, zero-padded, in front of the base name */
    if (out_.empty()) return;
    shut(w);
    string num = to_string(w.working + 1);
    while ((int)num.size() < digits_) num = "0" + num;
    const size_t slash = out_.find_last_of('/');
    const string dir = slash == string::npos ? "./" : out_.substr(0, slash + 1);
    const string base = slash == string::npos ? out_ : out_.substr(slash + 1);
    const string path = dir + num + "." + base;
    w.f = fopen(path.c_str(), "wb");
    if (!w.f) error_exit("Failed to write: " + path);
    w.gz = path.size() > 3 && path.compare(path.size() - 3, 3, ".gz") == 0;
    w.wrote = false;
    names.push_back(path);
}

bool load_fasta_contigs(const string& path, map<string, string>& contigs, string& err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        err = "There is a problem with the provided fasta file: could NOT read " + path;
        return false;
    }
    string data;
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) data.append(buf, n);
    fclose(f);
    /* FastaReader's constructor + readNext + readAll, src/fastareader.cpp:5-101, restated on the bytes of the file:
       the constructor skips to the first '>' (wherever it is); from then on a record ends where a LINE starts with
       '>' -- a '>' inside a header or a sequence line is an ordinary character.  Of every line the first character
       is taken by get() (upper-cased, otherwise as it is -- even a line feed, when the line is empty) and the rest by
       getline(), which goes to the header for the first line and through str_keep_valid_sequence (upper case,
       letters / '-' / '*' only) for the others. */
    size_t i = data.find('>');
    bool eof = i == string::npos;
    if (!eof) i++;
    while (!eof) {
        string header, seq;
        bool foundHeader = false;
        for (;;) {
            if (i >= data.size()) {
                eof = true;
                break;
            }
            char c = data[i++];
            if (c == '>') break;
            if (foundHeader) {
                if (c >= 'a' && c <= 'z') c -= ('a' - 'A');
                seq += c;
            } else {
                header += c;
            }
            const size_t e = data.find('\n', i);
            const string line = data.substr(i, (e == string::npos ? data.size() : e) - i);
            i = e == string::npos ? data.size() : e + 1;
            if (!foundHeader) {
                header += line;
                foundHeader = true;
            } else {
                for (char ch : line) {
                    if (ch >= 'a' && ch <= 'z') ch -= ('a' - 'A');
                    if (isalpha((unsigned char)ch) || ch == '-' || ch == '*') seq += ch;
                }
            }
        }
        contigs[header] = seq;
    }
    return true;
}

bool load_fasta_adapters(const string& path, vector<string>& adapters, ostream* log, string& err) {
    map<string, string> contigs;
    if (!load_fasta_contigs(path, contigs, err)) return false;
    for (auto& kv : contigs) { /* Options::loadFastaAdapters, src/options.cpp:50-59 */
        if (kv.second.length() >= 6) adapters.push_back(kv.second);
        else if (log) *log << "skip too short adapter sequence in " << path << " (6bp required): " << kv.second << endl;
    }
    return true;
}

}  // namespace fplh

extern "C" int fplh_load_fasta(const char* path, char** out, unsigned long long* out_len) {
    std::map<std::string, std::string> contigs;
    std::string err;
    if (!fplh::load_fasta_contigs(path, contigs, err)) return -1;
    std::string o;
    for (auto& kv : contigs) o += kv.first + "\t" + kv.second + "\n";
    *out = (char*)malloc(o.size() + 1);
    memcpy(*out, o.data(), o.size());
    *out_len = o.size();
    return (int)contigs.size();
}

extern "C" int fplh_split_replay(const char* out, int digits, int workers, int by_lines, int number, long size, int gz_level,
                                 unsigned n_packs, const int* worker, const long* reads, const long* passed,
                                 const char* const* texts) {
    fplh::SplitOutput so(out ? out : "", digits, workers, by_lines != 0, number, size, gz_level);
    for (unsigned k = 0; k < n_packs; k++) {
        so.write(worker[k], texts[k]);
        so.mark(worker[k], by_lines ? passed[k] : reads[k]);
    }
    so.close();
    return (int)so.names.size();
}
