#include "split.h"

#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <stdlib.h>
#include <unistd.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>

#include <ctype.h>
#include <iostream>

using namespace std;

namespace fplh {

static void error_exit(const string& msg) { /* src/util.h:270-273 */
    cerr << "ERROR: " << msg << endl;
    exit(-1);
}

/* libdeflate, loaded on first use (its API is a handful of C functions; declared here, the image ships the library
   without a header on the default include path) */
namespace {
struct Deflate {
    void* lib = nullptr;
    void* (*alloc_compressor)(int) = nullptr;
    size_t (*gzip_compress)(void*, const void*, size_t, void*, size_t) = nullptr;
    size_t (*gzip_compress_bound)(void*, size_t) = nullptr;
    void (*free_compressor)(void*) = nullptr;
    void* (*alloc_decompressor)() = nullptr;
    int (*gzip_decompress_ex)(void*, const void*, size_t, void*, size_t, size_t*, size_t*) = nullptr;
    void (*free_decompressor)(void*) = nullptr;
    Deflate() {
        if (getenv("FPLH_NO_LIBDEFLATE")) return; /* test hook: the zlib paths */
        for (const char* name : {"libdeflate.so.0", "libdeflate.so", "/usr/lib/x86_64-linux-gnu/libdeflate.so.0"}) {
            lib = dlopen(name, RTLD_NOW);
            if (lib) break;
        }
        if (!lib) return;
        alloc_compressor = (decltype(alloc_compressor))dlsym(lib, "libdeflate_alloc_compressor");
        gzip_compress = (decltype(gzip_compress))dlsym(lib, "libdeflate_gzip_compress");
        gzip_compress_bound = (decltype(gzip_compress_bound))dlsym(lib, "libdeflate_gzip_compress_bound");
        free_compressor = (decltype(free_compressor))dlsym(lib, "libdeflate_free_compressor");
        alloc_decompressor = (decltype(alloc_decompressor))dlsym(lib, "libdeflate_alloc_decompressor");
        gzip_decompress_ex = (decltype(gzip_decompress_ex))dlsym(lib, "libdeflate_gzip_decompress_ex");
        free_decompressor = (decltype(free_decompressor))dlsym(lib, "libdeflate_free_decompressor");
        if (!alloc_compressor || !gzip_compress || !gzip_compress_bound || !free_compressor || !alloc_decompressor ||
            !gzip_decompress_ex || !free_decompressor)
            lib = nullptr;
    }
};
const Deflate& deflate_lib() {
    static const Deflate d;
    return d;
}
/* one compressor / decompressor per thread and level (they are not thread-safe, and allocating one costs more than a
   small member) */
struct ThreadCodec {
    void* comp = nullptr;
    int level = -1;
    void* decomp = nullptr;
    ~ThreadCodec() {
        const Deflate& d = deflate_lib();
        if (comp) d.free_compressor(comp);
        if (decomp) d.free_decompressor(decomp);
    }
};
}  // namespace

bool have_libdeflate() { return deflate_lib().lib != nullptr; }

/* The deflated bytes go through a buffer the calling thread keeps (the pool's workers are persistent): dozens of
   threads allocating and releasing multi-megabyte strings per slice spend their time in the kernel's address-space
   lock instead. */
void gzip_into(const string& in, int level, string& out) {
    static thread_local vector<char> scratch;
    const Deflate& d = deflate_lib();
    if (d.lib) {
        static thread_local ThreadCodec tc;
        if (!tc.comp || tc.level != level) {
            if (tc.comp) d.free_compressor(tc.comp);
            tc.comp = d.alloc_compressor(level);
            tc.level = level;
            if (!tc.comp) error_exit("libdeflate_alloc_compressor failed");
        }
        const size_t bound = d.gzip_compress_bound(tc.comp, in.size());
        if (scratch.size() < bound) scratch.resize(bound);
        const size_t n = d.gzip_compress(tc.comp, in.data(), in.size(), scratch.data(), bound);
        if (n == 0) error_exit("libdeflate_gzip_compress failed");
        out.assign(scratch.data(), n);
        return;
    }
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

int gunzip_member_into(const unsigned char* in, size_t in_len, char* out, size_t out_cap, size_t* consumed, size_t* produced) {
    const Deflate& d = deflate_lib();
    if (!d.lib) return -1;
    static thread_local ThreadCodec tc;
    if (!tc.decomp) tc.decomp = d.alloc_decompressor();
    if (!tc.decomp) return -1;
    size_t used = 0, made = 0;
    const int rc = d.gzip_decompress_ex(tc.decomp, in, in_len, out, out_cap, &used, &made);
    if (rc == 0) {
        if (consumed) *consumed = used;
        if (produced) *produced = made;
        return 1;
    }
    return rc == 3 ? 2 : 0;
}

int gunzip_member(const unsigned char* in, size_t in_len, RawBuf& out, size_t cap, size_t* consumed, size_t hint) {
    const Deflate& d = deflate_lib();
    out.clear();
    if (d.lib) {
        static thread_local ThreadCodec tc;
        if (!tc.decomp) tc.decomp = d.alloc_decompressor();
        if (!tc.decomp) return 0;
        /* the member's own trailer says how long it inflates to (mod 2^32), but where the member ends is what is being
           found out: start from the caller's guess and grow (a wrong guess costs one more pass over the member) */
        /* (untouched pages of a generous buffer cost nothing, a second pass over the member does) */
        size_t guess = min<size_t>(cap, max<size_t>(64u << 20, hint));
        for (;;) {
            out.reserve(guess);
            size_t used = 0, produced = 0;
            const int rc = d.gzip_decompress_ex(tc.decomp, in, in_len, out.p, guess, &used, &produced);
            if (rc == 0) {
                out.n = produced;
                if (consumed) *consumed = used;
                return 1;
            }
            if (rc != 3) { /* bad data / truncated */
                out.release();
                return 0;
            }
            if (guess >= cap) { /* LIBDEFLATE_INSUFFICIENT_SPACE at the cap */
                out.release();
                return 2;
            }
            guess = min(cap, guess * 2);
        }
    }
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, 15 + 16) != Z_OK) return 0;
    zs.next_in = (Bytef*)in;
    size_t in_left = in_len;
    out.reserve(min<size_t>(max<size_t>(4u << 20, hint), cap));
    size_t produced = 0;
    int state = 0;
    for (;;) {
        if (zs.avail_in == 0 && in_left > 0) {
            zs.avail_in = (uInt)min<size_t>(in_left, 1u << 30);
            in_left -= zs.avail_in;
        }
        if (produced == out.cap) {
            if (out.cap >= cap) {
                state = 2;
                break;
            }
            out.reserve(min(cap, out.cap * 2));
        }
        zs.next_out = (Bytef*)out.p + produced;
        zs.avail_out = (uInt)min<size_t>(out.cap - produced, 1u << 30);
        const uInt before = zs.avail_out;
        const int rc = inflate(&zs, Z_NO_FLUSH);
        produced += before - zs.avail_out;
        if (rc == Z_STREAM_END) {
            state = 1;
            if (consumed) *consumed = (size_t)((const unsigned char*)zs.next_in - in);
            break;
        }
        if (rc != Z_OK || (zs.avail_in == 0 && in_left == 0 && zs.avail_out != 0)) break; /* bad data / truncated */
    }
    inflateEnd(&zs);
    if (state == 1) out.n = produced;
    else out.release();
    return state;
}
string gzip_member(const string& in, int level) {
    string o;
    gzip_into(in, level, o);
    return o;
}

SplitOutput::SplitOutput(const string& out, int digits, int workers, bool by_lines, int number, long size, int gz_level)
    : out_(out), digits_(digits), T_(workers), by_lines_(by_lines), number_(number), size_(size), level_(gz_level), w_(workers) {
    gz_ = out_.size() > 3 && out_.compare(out_.size() - 3, 3, ".gz") == 0;
    for (int t = 0; t < T_; t++) {
        w_[t].working = t; /* mWorkingSplit = threadId */
        open(w_[t]);
    }
}
SplitOutput::~SplitOutput() {
    if (!closed_) close();
}
void SplitOutput::write(int t, const string& text) { /* config->getWriter1()->writeString(outstr), src/seprocessor.cpp:297-301 */
    if (out_.empty()) return;
    Worker& w = w_[t];
    w.pending += text;
    if (w.pending.size() >= (4u << 20)) flush(w);
}
void SplitOutput::write_gather(int t, struct iovec* iov, size_t cnt) {
    if (out_.empty() || cnt == 0) return;
    Worker& w = w_[t];
    if (gz_) { /* (a gzip member needs the text in one piece) */
        for (size_t k = 0; k < cnt; k++) w.pending.append((const char*)iov[k].iov_base, iov[k].iov_len);
        if (w.pending.size() >= (4u << 20)) flush(w);
        return;
    }
    flush(w);
    size_t k = 0;
    while (k < cnt) { /* writev takes 1024 entries at a time and may stop short */
        const int c = (int)min<size_t>(1024, cnt - k);
        ssize_t n = writev(w.fd, iov + k, c);
        if (n < 0) {
            if (errno == EINTR) continue;
            error_exit("write failed");
        }
        while (n > 0 && k < cnt) {
            if ((size_t)n >= iov[k].iov_len) {
                n -= (ssize_t)iov[k].iov_len;
                k++;
            } else {
                iov[k].iov_base = (char*)iov[k].iov_base + n;
                iov[k].iov_len -= (size_t)n;
                n = 0;
            }
        }
        while (k < cnt && iov[k].iov_len == 0) k++;
    }
    w.wrote = true;
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
void SplitOutput::finish(Worker& w) { /* ThreadConfig::cleanup: files a short input never reached still have to exist */
    if (!by_lines_)
        while (w.working + T_ < number_) {
            w.working += T_;
            open(w);
        }
    shut(w);
}
void SplitOutput::close() {
    if (closed_) return;
    closed_ = true;
    if (threaded()) {
        for (Worker& w : w_) {
            { lock_guard<mutex> g(w.m); w.stop = true; }
            w.cv.notify_one();
        }
        for (thread& th : threads_) th.join();
        threads_.clear();
    } else {
        for (Worker& w : w_) finish(w);
    }
    for (Worker& w : w_) names.insert(names.end(), w.opened.begin(), w.opened.end());
}
void SplitOutput::start_threads() {
    if (threaded() || closed_) return;
    for (int t = 0; t < T_; t++)
        threads_.emplace_back([this, t]() {
            Worker& w = w_[t];
            for (;;) {
                function<void()> job;
                {
                    unique_lock<mutex> g(w.m);
                    w.cv.wait(g, [&] { return !w.q.empty() || w.stop; });
                    if (w.q.empty()) break; /* (stop, and nothing left) */
                    job = std::move(w.q.front());
                    w.q.pop_front();
                }
                job();
            }
            finish(w);
        });
}
void SplitOutput::post(int t, function<void()> job) {
    Worker& w = w_[t];
    { lock_guard<mutex> g(w.m); w.q.push_back(std::move(job)); }
    w.cv.notify_one();
}
void SplitOutput::put(Worker& w, const char* p, size_t n) {
    while (n > 0) {
        const ssize_t k = ::write(w.fd, p, n);
        if (k < 0) {
            if (errno == EINTR) continue;
            error_exit("write failed");
        }
        p += k;
        n -= (size_t)k;
    }
}
void SplitOutput::flush(Worker& w) {
    if (w.pending.empty() || w.fd < 0) return;
    if (gz_) {
        const string bytes = gzip_member(w.pending, level_);
        put(w, bytes.data(), bytes.size());
    } else {
        put(w, w.pending.data(), w.pending.size());
    }
    w.wrote = true;
    w.pending.clear();
}
void SplitOutput::shut(Worker& w) {
    if (w.fd < 0) return;
    flush(w);
    if (gz_ && !w.wrote) {
        const string e = gzip_member(string(), level_);
        put(w, e.data(), e.size());
    }
    if (::close(w.fd) != 0) error_exit("write failed");
    w.fd = -1;
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
    w.fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
    if (w.fd < 0) error_exit("Failed to write: " + path);
    w.wrote = false;
    w.opened.push_back(path);
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
