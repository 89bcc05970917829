#include "fastq.h"
#include "split.h"

#include <fcntl.h>
#include <immintrin.h>
#include <stdlib.h>
#include <string.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <iostream>
#include <map>
#include <mutex>
#include <thread>

using namespace std;

namespace fplh {

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static const bool g_timing = getenv("FPLH_TIMING") != nullptr;
static std::atomic<uint64_t> g_chunk_us[3]; /* FPLH_TIMING: microseconds the chunk parsers spent reading / locating / copying */

int effective_cpus() {
    static const int cached = []() {
        if (const char* e = getenv("FPLH_CPUS"))
            if (atoi(e) > 0) return atoi(e);
        int n = max(1, (int)std::thread::hardware_concurrency());
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) n = min(n, (int)CPU_COUNT(&set));
        auto quota = [](const char* path, bool v2) -> double {
            FILE* f = fopen(path, "r");
            if (!f) return 0;
            char a[64] = {0}, b[64] = {0};
            double q = 0;
            if (v2) { /* "max 100000" or "<quota> <period>" */
                if (fscanf(f, "%63s %63s", a, b) == 2 && strcmp(a, "max") != 0 && atof(b) > 0) q = atof(a) / atof(b);
            } else if (fscanf(f, "%63s", a) == 1) {
                q = atof(a); /* microseconds per period, -1 = none */
            }
            fclose(f);
            return q;
        };
        double q = quota("/sys/fs/cgroup/cpu.max", true);
        if (q <= 0) {
            const double us = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", false), per = quota("/sys/fs/cgroup/cpu/cpu.cfs_period_us", false);
            if (us > 0 && per > 0) q = us / per;
        }
        if (q > 0) n = min(n, max(1, (int)(q + 0.5)));
        return n;
    }();
    return cached;
}

/* Bytes of memory this process may still take: the smaller of what the machine has available (MemAvailable) and what its
   cgroup leaves (memory.max - memory.current, v2; limit_in_bytes - usage_in_bytes, v1) -- a container's limit is usually far
   below the node's RAM, and going over it is a kill, not an error.  FPLH_MEM_BYTES overrides (tests). */
uint64_t memory_budget() {
    if (const char* e = getenv("FPLH_MEM_BYTES"))
        if (atoll(e) > 0) return (uint64_t)atoll(e);
    auto number = [](const char* path, uint64_t& v) -> bool {
        FILE* f = fopen(path, "r");
        if (!f) return false;
        char a[64] = {0};
        const bool got = fscanf(f, "%63s", a) == 1;
        fclose(f);
        if (!got || a[0] < '0' || a[0] > '9') return false; /* "max": no limit */
        v = strtoull(a, nullptr, 10);
        return true;
    };
    uint64_t best = (uint64_t)sysconf(_SC_PHYS_PAGES) * (uint64_t)sysconf(_SC_PAGE_SIZE);
    if (FILE* f = fopen("/proc/meminfo", "r")) {
        char line[256];
        while (fgets(line, sizeof(line), f)) {
            unsigned long long kb = 0;
            if (sscanf(line, "MemAvailable: %llu kB", &kb) == 1) best = min<uint64_t>(best, (uint64_t)kb << 10);
        }
        fclose(f);
    }
    uint64_t lim = 0, use = 0;
    if (number("/sys/fs/cgroup/memory.max", lim) || number("/sys/fs/cgroup/memory/memory.limit_in_bytes", lim)) {
        if (!number("/sys/fs/cgroup/memory.current", use)) number("/sys/fs/cgroup/memory/memory.usage_in_bytes", use);
        if (lim < (1ull << 60)) best = min<uint64_t>(best, lim > use ? lim - use : 0);
    }
    return best;
}

namespace {
class Pool {
   public:
    Pool() {
        const int hw = effective_cpus();
        int n = min(64, max(1, hw - 1));
        if (const char* e = getenv("FPLH_POOL_THREADS"))
            if (atoi(e) >= 0) n = atoi(e);
        for (int i = 0; i < n; i++) workers_.emplace_back([this]() { work(); });
    }
    ~Pool() {
        {
            lock_guard<mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    void run(int tasks, const function<void(int)>& fn) {
        if (tasks <= 0) return;
        if (tasks == 1 || workers_.empty()) {
            for (int i = 0; i < tasks; i++) fn(i);
            return;
        }
        Job job{&fn, tasks, 0, {0}};
        {
            lock_guard<mutex> g(m_);
            jobs_.push_back(&job);
        }
        cv_.notify_all();
        for (;;) { /* the caller works too */
            int i;
            {
                lock_guard<mutex> g(m_);
                i = claim(&job);
            }
            if (i < 0) break;
            fn(i);
            job.done.fetch_add(1);
        }
        unique_lock<mutex> g(m_);
        done_cv_.wait(g, [&]() { return job.done.load() == tasks; });
    }

   private:
    struct Job {
        const function<void(int)>* fn;
        int n, next;
        atomic<int> done;
    };
    /* next index of job j, or -1 when all are handed out (m_ held); a job leaves the queue with its last index, so
       nobody looks at it once its caller may have returned */
    int claim(Job* j) {
        if (j->next >= j->n) return -1;
        const int i = j->next++;
        if (j->next == j->n) jobs_.erase(std::find(jobs_.begin(), jobs_.end(), j));
        return i;
    }
    void work() {
        unique_lock<mutex> g(m_);
        for (;;) {
            cv_.wait(g, [&]() { return stop_ || !jobs_.empty(); });
            if (stop_) return;
            Job* j = jobs_.front();
            const int i = claim(j);
            if (i < 0) continue;
            const function<void(int)>* fn = j->fn;
            const int n = j->n;
            g.unlock();
            (*fn)(i);
            const bool last = j->done.fetch_add(1) + 1 == n; /* j may be gone right after this */
            g.lock();
            if (last) done_cv_.notify_all();
        }
    }
    mutex m_;
    condition_variable cv_, done_cv_;
    deque<Job*> jobs_;
    vector<std::thread> workers_;
    bool stop_ = false;
};
}  // namespace

void parallel_run(int tasks, const function<void(int)>& fn) {
    static Pool pool;
    pool.run(tasks, fn);
}

/* ---- gzip input made of several members ---------------------------------------------------------------
 * One deflate stream cannot be inflated in parallel, but a gzip FILE is often a concatenation of members: bgzip
 * blocks, `cat` of the per-chunk files sequencers write, the 4 MiB flushes of fastp / fastplong, the slices of this
 * host's own writer.  Members start with 1f 8b 08 and a flag byte whose top three bits are zero; that pattern
 * also occurs inside compressed data, so a candidate only counts once a member that starts there has been inflated
 * to its end with a good CRC (zlib checks it) AND the chain of members starting at offset 0 lands on it.  Batches
 * of candidates are inflated speculatively on the worker pool; the chain walk then keeps what lines up and drops
 * the rest.  A member that inflates to more than 512 MiB (a plain `gzip` of a whole run) is not buffered:
 * from there on the file is streamed through zlib as before. */
class GzMembers {
   public:
    static std::atomic<uint64_t> delivered; /* members handed to the parser since the last fplh_gz_members() (test hook) */
    static GzMembers* open(const string& path, int threads) {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return nullptr;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 64) {
            close(fd);
            return nullptr;
        }
        void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) {
            close(fd);
            return nullptr;
        }
        GzMembers* g = new GzMembers;
        g->fd_ = fd;
        g->base_ = (const unsigned char*)m;
        g->size_ = (size_t)st.st_size;
        g->threads_ = max(1, threads);
        if (const char* e = getenv("FPLH_GZ_MEMBER_CAP")) /* test hook */
            if (atol(e) > 0) g->cap_ = (size_t)atol(e);
        g->find_candidates();
        if (g->cands_.size() < 2 || g->cands_[0] != 0) { /* one member (or not gzip): nothing to gain */
            delete g;
            return nullptr;
        }
        return g;
    }
    ~GzMembers() {
        if (stream_) gzclose(stream_);
        if (base_) munmap((void*)base_, size_);
        if (fd_ >= 0 && !stream_) close(fd_); /* (gzclose closes the descriptor it was given) */
    }
    /* next bytes of the inflated stream; 0 = end of input */
    size_t read(char* dst, size_t n) {
        size_t got = 0;
        while (got < n) {
            if (stream_) {
                const int r = gzread(stream_, dst + got, (unsigned)min<size_t>(n - got, 1u << 30));
                if (r <= 0) {
                    int errnum = Z_OK;
                    gzerror(stream_, &errnum);
                    if (r < 0 || (errnum != Z_OK && errnum != Z_STREAM_END)) err_ = errnum == Z_OK ? Z_ERRNO : errnum;
                    break;
                }
                got += (size_t)r;
                continue;
            }
            if (cur_off_ < cur_.size()) {
                const size_t k = min(n - got, cur_.size() - cur_off_);
                memcpy(dst + got, cur_.data() + cur_off_, k);
                cur_off_ += k;
                got += k;
                continue;
            }
            if (!next_member()) break;
        }
        return got;
    }
    /* The next members of the chain, inflated (at most `threads` of them at a time), in file order; false when the chain
       cannot be followed this way any further -- the end of the input (`*at_end`), or a member that is too large to
       buffer / damaged (the caller goes back to the stream) */
    bool next_group(vector<RawBuf>& out, bool* at_end) {
        out.clear();
        *at_end = false;
        if (stream_) return false;
        if (pos_ >= size_ || size_ - pos_ < 18 || !looks_like_header(base_ + pos_)) {
            *at_end = true;
            return false;
        }
        vector<size_t> todo;
        for (auto c = std::lower_bound(cands_.begin(), cands_.end(), pos_); c != cands_.end() && (int)todo.size() < threads_; ++c)
            if (!done_.count(*c)) todo.push_back(*c);
        if (!done_.count(pos_) && (todo.empty() || todo[0] != pos_)) todo.insert(todo.begin(), pos_);
        vector<Result> res(todo.size());
        parallel_run((int)todo.size(), [&](int i) { inflate_at(todo[i], res[i]); });
        for (size_t i = 0; i < todo.size(); i++) done_[todo[i]] = std::move(res[i]);
        for (;;) {
            auto it = done_.find(pos_);
            if (it == done_.end()) break;
            if (it->second.state != 1) return !out.empty(); /* (the next call reports the member that cannot be taken) */
            const size_t end = it->second.end;
            out.emplace_back(std::move(it->second.out));
            for (auto d = done_.begin(); d != done_.end();)
                d = d->first < end ? done_.erase(d) : std::next(d);
            pos_ = end;
            n_parallel_++;
            delivered++;
        }
        if (out.empty()) { /* pos_ is there and cannot be taken */
            auto it = done_.find(pos_);
            if (it != done_.end() && it->second.state != 1) return false;
        }
        return !out.empty();
    }
    uint64_t members_inflated_in_parallel() const { return n_parallel_; }
    int error() const { return err_; } /* zlib's code when the stream turned out damaged or truncated, else 0 */

   private:
    struct Result {
        RawBuf out;
        size_t end = 0; /* file offset behind the member's trailer */
        int state = 0;  /* 1 = a whole member, 2 = too large to buffer, -1 = not a member */
    };
    static bool looks_like_header(const unsigned char* p) { return p[0] == 0x1f && p[1] == 0x8b && p[2] == 8 && (p[3] & 0xE0) == 0; }
    void find_candidates() {
        const int T = (int)max<size_t>(1, min<size_t>((size_t)threads_, size_ / (4u << 20)));
        vector<vector<size_t>> found(T);
        parallel_run(T, [&](int t) {
            const size_t lo = size_ / T * t, hi = t == T - 1 ? size_ : size_ / T * (t + 1);
            const unsigned char* p = base_ + lo;
            const unsigned char* e = base_ + min(hi, size_ - 18); /* header 10 + trailer 8 at least */
            while (p < e) {
                p = (const unsigned char*)memchr(p, 0x1f, (size_t)(e - p));
                if (!p) break;
                if (looks_like_header(p)) found[t].push_back((size_t)(p - base_));
                p++;
            }
        });
        for (auto& v : found) cands_.insert(cands_.end(), v.begin(), v.end());
    }
    void inflate_at(size_t off, Result& r) const { /* (libdeflate when the system has it, else zlib: split.cpp) */
        size_t used = 0;
        /* a first guess of the inflated size: four times the distance to the next candidate header */
        auto nx = std::upper_bound(cands_.begin(), cands_.end(), off);
        const size_t span = (nx == cands_.end() ? size_ : *nx) - off;
        const int st = gunzip_member(base_ + off, size_ - off, r.out, cap_, &used, span * 4 + (64u << 10));
        r.state = st == 1 ? 1 : (st == 2 ? 2 : -1);
        r.end = off + used;
    }
    /* make the member at pos_ current; false at the end of the input */
    bool next_member() {
        cur_.clear();
        cur_off_ = 0;
        for (;;) {
            if (pos_ >= size_ || size_ - pos_ < 18 || !looks_like_header(base_ + pos_)) return false; /* end, or trailing bytes zlib ignores too */
            auto it = done_.find(pos_);
            if (it == done_.end()) { /* inflate the next candidates at and behind pos_ that are not there yet */
                vector<size_t> todo;
                for (auto c = std::lower_bound(cands_.begin(), cands_.end(), pos_); c != cands_.end() && (int)todo.size() < threads_; ++c)
                    if (!done_.count(*c)) todo.push_back(*c);
                if (todo.empty() || todo[0] != pos_) todo.insert(todo.begin(), pos_);
                vector<Result> res(todo.size());
                parallel_run((int)todo.size(), [&](int i) { inflate_at(todo[i], res[i]); });
                for (size_t i = 0; i < todo.size(); i++) done_[todo[i]] = std::move(res[i]);
                it = done_.find(pos_);
            }
            Result& r = it->second;
            if (r.state == 2 || r.state == -1) {
                /* too large to buffer (or damaged: let zlib report it the usual way): stream the rest */
                done_.clear();
                if (lseek(fd_, (off_t)pos_, SEEK_SET) < 0) return false;
                stream_ = gzdopen(fd_, "rb");
                if (stream_) gzbuffer(stream_, 1 << 20);
                else err_ = Z_ERRNO;
                return stream_ != nullptr;
            }
            cur_.swap(r.out);
            const size_t end = r.end;
            for (auto d = done_.begin(); d != done_.end();) /* speculative results the chain has passed */
                d = d->first < end ? done_.erase(d) : std::next(d);
            pos_ = end;
            n_parallel_++;
            delivered++;
            if (!cur_.empty()) return true; /* (an empty member: go on to the next) */
        }
    }
    int fd_ = -1;
    const unsigned char* base_ = nullptr;
    size_t size_ = 0, pos_ = 0, cur_off_ = 0;
    int threads_ = 1;
    size_t cap_ = 512ull << 20; /* largest inflated member that is buffered */
    vector<size_t> cands_;
    std::map<size_t, Result> done_;
    RawBuf cur_;
    gzFile stream_ = nullptr;
    uint64_t n_parallel_ = 0;
    int err_ = 0;
};

std::atomic<uint64_t> GzMembers::delivered{0};

/* A gzip file made of several members -> its inflated text in anonymous memory, so that the chunk-parallel reader can
 * take it like a mapped file (the members are inflated on `threads` workers and copied into place side by side; address
 * space for `max_bytes` is reserved up front, pages are only touched as the text arrives).  nullptr when the file is not
 * of that kind, a member cannot be buffered or checked, or the text would take more than `max_bytes`: the caller then
 * reads the input through the sequential stream as before.  The caller owns the mapping (`*reserved` bytes). */
/* A gzip file that is ONE member (a plain `gzip` of a whole run): no two workers can share a deflate stream, but libdeflate
 * inflates a whole member 2.3 times faster than zlib streams it, and the text can then be parsed by all the chunk parsers.
 * The member's trailer gives its inflated size modulo 4 GiB; the candidates size, size + 4 GiB, ... are tried in turn (a
 * wrong one fails with "no space" at the end of the output).  nullptr: not a single clean member, no libdeflate, or more
 * text than max_bytes. */
static char* gunzip_single_to_memory(const string& path, uint64_t max_bytes, uint64_t* size_out, uint64_t* reserved) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return nullptr;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 18) {
        close(fd);
        return nullptr;
    }
    const size_t fsize = (size_t)st.st_size;
    const unsigned char* in = (const unsigned char*)mmap(nullptr, fsize, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (in == (const unsigned char*)MAP_FAILED) return nullptr;
    char* result = nullptr;
    if (in[0] == 0x1f && in[1] == 0x8b && in[2] == 8) {
        /* where the member ends: at the end of the file -- or, when zero bytes trail it (zlib ignores padding behind the last
           member: so does this), 0..3 bytes behind the last non-zero byte (the size field itself may end in zero bytes).  The
           likeliest end is tried first -- the file's own when fewer than four zero bytes trail it, else the last non-zero
           byte's (a size field whose top byte is zero means a text within 16 MiB of a multiple of 4 GiB) -- and a member that
           is followed by another one ends the attempt */
        size_t tail = fsize;
        while (tail > 18 && in[tail - 1] == 0 && (fsize < 4096 || tail > fsize - 4096)) tail--;
        size_t ends[5];
        int n_ends = 0;
        const bool padded = fsize - tail >= 4; /* four zero bytes at the very end: padding, or a text of k * 4 GiB */
        if (!padded) ends[n_ends++] = fsize;
        for (int pad = 0; pad < 4 && tail < fsize; pad++)
            if (tail + (size_t)pad < fsize) ends[n_ends++] = tail + (size_t)pad;
        if (padded) ends[n_ends++] = fsize;
        /* The candidate ends only say how much text to make room for (the size field in front of them).  The inflate itself
           always gets the whole file: it stops where the member really ends (`used`, trailer checked) -- so a member with
           padding behind it is accepted from the FIRST attempt that had room for its text, instead of being inflated again for
           every guess of where the padding starts.  At most four attempts in all: every one is a full pass over the file. */
        uint64_t wants[12];
        int n_wants = 0;
        for (int e = 0; e < n_ends; e++) {
            const size_t end = ends[e];
            if (end < 18) continue;
            const uint64_t isize = (uint64_t)in[end - 4] | ((uint64_t)in[end - 3] << 8) | ((uint64_t)in[end - 2] << 16) | ((uint64_t)in[end - 1] << 24);
            for (uint64_t want = isize; want <= max_bytes && n_wants < 12; want += 1ull << 32) {
                bool seen = want == 0;
                for (int k = 0; k < n_wants; k++) seen = seen || wants[k] == want;
                if (!seen) wants[n_wants++] = want;
                if (want - isize >= (1ull << 32)) break; /* (one wrap per candidate: a text beyond 8 GiB of a guess is the next guess's) */
            }
        }
        int attempts = 0;
        for (int k = 0; k < n_wants && !result && attempts < 4; k++) {
            const uint64_t want = wants[k];
            const uint64_t span = want + (4u << 20);
            char* base = (char*)mmap(nullptr, (size_t)span, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (base == (char*)MAP_FAILED) break;
            madvise(base, (size_t)span, MADV_HUGEPAGE); /* (one fault per 2 MiB instead of per 4 KiB as the text arrives) */
            size_t used = 0, made = 0;
            attempts++;
            const int rc = gunzip_member_into(in, fsize, base, (size_t)want, &used, &made);
            if (rc == 1 && made > 0) {
                /* a whole member.  Zero padding may follow (zlib ignores it: so does this); anything else is another member --
                   not for this lane */
                size_t z = used;
                while (z < fsize && in[z] == 0) z++;
                if (z == fsize) {
                    result = base;
                    *size_out = made;
                    *reserved = span;
                    break;
                }
                munmap(base, (size_t)span);
                break;
            }
            munmap(base, (size_t)span);
            if (rc != 2) break; /* damaged, or no libdeflate: the streaming reader reports it / takes over */
            /* rc == 2: more text than this guess made room for: the next one */
        }
    }
    munmap((void*)in, fsize);
    return result;
}

char* gunzip_members_to_memory(const string& path, int threads, uint64_t max_bytes, uint64_t* size_out, uint64_t* reserved) {
    /* The size field at the end of the file belongs to its LAST member.  When it says "at least as much text as the whole file
       has bytes", the file is almost certainly one member: that lane first (bytes that look like a member header inside the
       compressed data would otherwise send it through the member chain, whose size guesses for a member this large cost
       several passes).  Otherwise the chain first, the single-member lane if the chain finds only one. */
    bool single_first = false;
    {
        const int fd = ::open(path.c_str(), O_RDONLY);
        struct stat st;
        unsigned char t[4];
        if (fd >= 0 && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size >= 18 && pread(fd, t, 4, st.st_size - 4) == 4) {
            const uint64_t isize = (uint64_t)t[0] | ((uint64_t)t[1] << 8) | ((uint64_t)t[2] << 16) | ((uint64_t)t[3] << 24);
            single_first = isize >= (uint64_t)st.st_size;
        }
        if (fd >= 0) close(fd);
    }
    if (single_first)
        if (char* one = gunzip_single_to_memory(path, max_bytes, size_out, reserved)) return one;
    GzMembers* g = GzMembers::open(path, threads);
    if (!g) return single_first ? nullptr : gunzip_single_to_memory(path, max_bytes, size_out, reserved);
    const uint64_t span = max_bytes + (4u << 20);
    char* base = (char*)mmap(nullptr, (size_t)span, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (base == (char*)MAP_FAILED) {
        delete g;
        return nullptr;
    }
    madvise(base, (size_t)span, MADV_HUGEPAGE); /* (one fault per 2 MiB instead of per 4 KiB as the text arrives) */
    uint64_t total = 0;
    bool ok = true, at_end = false;
    vector<RawBuf> group;
    while (ok && g->next_group(group, &at_end)) {
        vector<uint64_t> at(group.size());
        for (size_t i = 0; i < group.size(); i++) {
            at[i] = total;
            total += group[i].n;
        }
        if (total > max_bytes) {
            ok = false;
            break;
        }
        parallel_run((int)group.size(), [&](int i) {
            memcpy(base + at[i], group[i].p, group[i].n);
            group[i].release();
        });
    }
    if (!at_end || g->error()) ok = false;
    delete g;
    if (!ok || total == 0) {
        munmap(base, (size_t)span);
        return nullptr;
    }
    *size_out = total;
    *reserved = span;
    return base;
}

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

static std::atomic<uint64_t> g_alloc_seconds_x1000{0}, g_alloc_bytes{0}; /* microseconds / bytes spent in the page-locked allocator */
static ByteBuf::AllocFn g_alloc = nullptr;
static ByteBuf::FreeFn g_free = nullptr;
void ByteBuf::set_allocator(AllocFn a, FreeFn f) {
    g_alloc = a;
    g_free = f;
}
namespace {
struct Arena {
    uint8_t* base = nullptr;
    size_t block = 0, n = 0;
    vector<uint8_t*> free_blocks;
    mutex mu;
    bool released = false;
    bool owns(const uint8_t* p) const { return base && p >= base && p < base + block * n; }
} g_arena;
}  // namespace
void ByteBuf::set_arena(size_t block_bytes, size_t n_blocks) {
    if (!g_alloc || g_arena.base || block_bytes == 0 || n_blocks == 0) return;
    block_bytes = (block_bytes + 4095) & ~(size_t)4095;
    const double t0 = now_s();
    g_arena.base = (uint8_t*)g_alloc(block_bytes * n_blocks);
    g_alloc_seconds_x1000.fetch_add((uint64_t)((now_s() - t0) * 1e6));
    if (!g_arena.base) return; /* (buffers then come from the allocator one by one) */
    g_alloc_bytes.fetch_add(block_bytes * n_blocks);
    g_arena.block = block_bytes;
    g_arena.n = n_blocks;
    for (size_t i = n_blocks; i-- > 0;) g_arena.free_blocks.push_back(g_arena.base + i * block_bytes);
}
void ByteBuf::release_arena() {
    lock_guard<mutex> g(g_arena.mu);
    if (g_arena.base && g_free && !g_arena.released) g_free(g_arena.base);
    g_arena.released = true; /* (owns() stays true: a buffer of the arena that is destroyed later is simply dropped) */
    g_arena.free_blocks.clear();
}
static void buf_release(uint8_t* p) {
    if (g_arena.owns(p)) {
        lock_guard<mutex> g(g_arena.mu);
        if (!g_arena.released) g_arena.free_blocks.push_back(p); /* (a released arena hands nothing out again) */
    } else if (g_free) {
        g_free(p);
    } else {
        free(p);
    }
}
ByteBuf::~ByteBuf() {
    if (p_) buf_release(p_);
}
void ByteBuf::reserve(size_t c) {
    if (c <= cap_) return;
    uint8_t* np = nullptr;
    size_t nc = 0;
    if (g_arena.base && !g_arena.released && c <= g_arena.block) {
        lock_guard<mutex> g(g_arena.mu);
        if (!g_arena.free_blocks.empty()) {
            np = g_arena.free_blocks.back();
            g_arena.free_blocks.pop_back();
            nc = g_arena.block;
        }
    }
    if (!np) {
        nc = cap_ ? cap_ : 4096;
        while (nc < c) nc += nc / 2 + 4096; /* (page-locked memory is not cheap: grow by halves, not by doubling) */
        if (g_alloc) {
            const double t0 = now_s();
            np = (uint8_t*)g_alloc(nc);
            g_alloc_seconds_x1000.fetch_add((uint64_t)((now_s() - t0) * 1e6));
            g_alloc_bytes.fetch_add(nc);
            if (!np) {
                cerr << "ERROR: cannot allocate " << nc << " bytes of page-locked host memory" << endl;
                exit(-1);
            }
        } else {
            np = (uint8_t*)malloc(nc);
            if (!np) {
                cerr << "ERROR: out of memory" << endl;
                exit(-1);
            }
        }
    }
    if (n_) memcpy(np, p_, n_);
    if (p_) buf_release(p_);
    p_ = np;
    cap_ = nc;
}

void Batch::clear() {
    seq.clear();
    qual.clear();
    off.clear();
    text.clear();
    name_off.clear();
    name_len.clear();
    strand_len.clear();
    raw.clear();
    raw_begin = raw_len = 0;
    line.clear();
    text_backed = false;
}

void Batch::adopt_lines(const uint32_t* ls, uint32_t n_records) {
    const uint8_t* t = raw.data();
    const uint32_t base = (uint32_t)raw_begin;
    line.resize(4 * (size_t)n_records);
    off.resize((size_t)n_records + 1);
    name_len.resize(n_records);
    strand_len.resize(n_records);
    uint64_t run = 0;
    for (uint32_t i = 0; i < n_records; i++) {
        uint32_t L[5];
        for (int j = 0; j < 4; j++) L[j] = line[4 * (size_t)i + j] = ls[4 * (size_t)i + j] + base;
        L[4] = i + 1 < n_records ? ls[4 * (size_t)i + 4] + base : base + (uint32_t)raw_len;
        uint32_t ll[4];
        for (int j = 0; j < 4; j++) { /* a line ends with "\n" or "\r\n" (regular text: the device checked) */
            uint32_t e = L[j + 1] - 1;
            if (e > L[j] && t[e - 1] == '\r') e--;
            ll[j] = e - L[j];
        }
        name_len[i] = ll[0];
        strand_len[i] = ll[2];
        off[i] = run;
        run += ll[1];
    }
    off[n_records] = run;
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

/* the reference's words for a gzip stream that ends early / does not decode (src/fastqreader.cpp:92-137) */
static string gz_error_text(int zerr, const string& path) {
    return zerr == Z_BUF_ERROR ? string("igzip: unexpected eof") : "igzip: encountered while decompressing file: " + path;
}

FastqReader::FastqReader(const string& path) {
    path_ = path;
    size_t cap = 32u << 20;
    bool allow_map = true;
    if (const char* e = getenv("FPLH_READ_WINDOW")) /* test hook: tiny windows exercise the refill paths */
        if (atol(e) > 0) {
            cap = (size_t)atol(e);
            allow_map = false;
        }
    if (const char* e = getenv("FPLH_PARSE_MIN"))
        if (atol(e) > 0) parse_min_ = (size_t)atol(e);
    if (const char* e = getenv("FPLH_PARSE_THREADS"))
        if (atol(e) > 0) copy_threads_ = (int)atol(e);
    if (path != "/dev/stdin" && allow_map) { /* a regular file that is not gzip: parallel pread into the window */
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd >= 0) {
            struct stat st;
            unsigned char magic[2] = {0, 0};
            if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0 && pread(fd, magic, 2, 0) == 2 &&
                !(magic[0] == 0x1f && magic[1] == 0x8b)) {
                fd_ = fd;
                file_size_ = (uint64_t)st.st_size;
                fp_ = this;
            } else {
                close(fd);
            }
        }
    }
    if (fd_ < 0 && path != "/dev/stdin" && allow_map && !getenv("FPLH_NO_GZ_MEMBERS")) {
        members_ = GzMembers::open(path, max(copy_threads_, 16)); /* inflate is compute-bound: more workers than the memory-bound phases use */
        if (members_) fp_ = this;
    }
    if (fd_ < 0 && !members_) {
        /* gzopen reads plain files transparently */
        fp_ = path == "/dev/stdin" ? (void*)gzdopen(0, "rb") : (void*)gzopen(path.c_str(), "rb");
        if (fp_) gzbuffer((gzFile)fp_, 1 << 20);
    }
    buf_.resize(cap);
    win_ = buf_.data();
}

FastqReader::FastqReader(const char* data, size_t len, bool at_eof) {
    mem_ = true;
    fp_ = this;
    win_ = data;
    len_ = len;
    pulled_ = len;
    eof_ = at_eof;
}

FastqReader::~FastqReader() {
    if (mem_) return;
    if (fd_ >= 0) close(fd_);
    else if (members_) delete members_;
    else if (fp_) gzclose((gzFile)fp_);
}

bool FastqReader::pull() {
    if (eof_ || !fp_ || mem_) return false;
    if (pos_ > 0) {
        memmove(buf_.data(), buf_.data() + pos_, len_ - pos_);
        len_ -= pos_;
        pos_ = 0;
    } else if (len_ == buf_.size()) {
        buf_.resize(buf_.size() * 2); /* one record is larger than the window */
    }
    win_ = buf_.data();
    if (fd_ >= 0) { /* regular file: every thread preads its slice of the free part of the window */
        const size_t want = (size_t)min<uint64_t>(buf_.size() - len_, file_size_ - file_pos_);
        const int T = (int)max<size_t>(1, min<size_t>((size_t)copy_threads_, want / (4u << 20)));
        std::atomic<bool> ok{true};
        parallel_run(T, [&](int t) {
            size_t a = want / T * t, e = t == T - 1 ? want : want / T * (t + 1);
            while (a < e) {
                const ssize_t n = pread(fd_, buf_.data() + len_ + a, e - a, (off_t)(file_pos_ + a));
                if (n <= 0) {
                    ok = false;
                    return;
                }
                a += (size_t)n;
            }
        });
        if (!ok) {
            eof_ = true; /* (truncated underneath us: stop with what was read so far, and say so) */
            io_error_ = "reading " + path_ + " failed (file truncated while it was being read?)";
            return true;
        }
        len_ += want;
        pulled_ += want;
        file_pos_ += want;
        if (file_pos_ >= file_size_) eof_ = true;
        return true;
    }
    if (members_) {
        const size_t n = members_->read(buf_.data() + len_, buf_.size() - len_);
        if (n == 0) eof_ = true;
        if (members_->error()) {
            eof_ = true;
            io_error_ = gz_error_text(members_->error(), path_);
        }
        len_ += n;
        pulled_ += n;
        return true;
    }
    while (len_ < buf_.size()) { /* gzread returns short counts on pipes */
        const size_t want = min<size_t>(buf_.size() - len_, 1u << 30);
        const int n = gzread((gzFile)fp_, buf_.data() + len_, (unsigned)want);
        if (n <= 0) {
            eof_ = true;
            int errnum = Z_OK;
            gzerror((gzFile)fp_, &errnum);
            if (n < 0 || (errnum != Z_OK && errnum != Z_STREAM_END)) io_error_ = gz_error_text(errnum == Z_OK ? Z_ERRNO : errnum, path_);
            break;
        }
        len_ += (size_t)n;
        pulled_ += (uint64_t)n;
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
    vector<size_t> cut(T + 1, nr);
    cut[0] = 0;
    for (int t = 0; t < T - 1; t++) { /* slices of about equal numbers of bases */
        const uint64_t want = base0 + (bases - base0) / T * (t + 1);
        const size_t last = (size_t)(std::lower_bound(b.off.begin() + n0 + 1, b.off.begin() + n0 + 1 + nr, want) -
                                     (b.off.begin() + n0 + 1)) + 1;
        cut[t + 1] = min(max(last, cut[t]), nr);
    }
    parallel_run(T, [&](int t) { work(cut[t], cut[t + 1]); });
}

/* Locate records starting at `pos` (a line start) while they START before `start_limit` and the running totals
 * stay below the caps; `pos` ends behind the last record taken (skipped non-'@' lines in front of a taken record
 * are consumed too).  Returns 0 = stopped at a cap / the limit, 1 = the window ran out inside a record (stream
 * mode: pull and call again), 2 = end of input, 3 = malformed record at `pos` (message in `err`). */
int FastqReader::scan_records(size_t& pos, size_t start_limit, uint64_t& bases, uint64_t max_bases, uint32_t& reads,
                              uint32_t max_reads, vector<Rec>& recs, string& err) const {
    const Line none = {nullptr, 0};
    while (bases < max_bases && reads < max_reads) {
        /* one record = the next line that starts with '@' (src/fastqreader.cpp:316-319) and the three lines after
           it; lines missing at the end of the input read as empty, as getLine() does */
        Rec rc = {none, none, none, none, 0};
        size_t p = pos, rec_start = pos;
        int r;
        for (;;) {
            rec_start = p;
            r = scan_line(p, rc.name);
            if (r <= 0) break;
            if (rc.name.n > 0 && rc.name.p[0] == '@') break;
            pos = p; /* a skipped line is consumed for good */
        }
        if (r == 0) return 1;
        if (r < 0) return 2;
        if (rec_start >= start_limit) return 0; /* belongs to the next stretch */
        Line* rest[3] = {&rc.seq, &rc.strand, &rc.qual};
        for (int k = 0; k < 3; k++) {
            r = scan_line(p, *rest[k]);
            if (r == 0) return 1; /* the record continues beyond the window: restart it after reading more */
            if (r < 0) *rest[k] = none;
        }
        if (rc.strand.n == 0 || rc.strand.p[0] != '+') {
            err = string(rc.name.p, rc.name.n) + "\nExpected '+', got " + string(rc.strand.p ? rc.strand.p : "", rc.strand.n) +
                  "\nYour FASTQ may be invalid, please check the tail of your FASTQ file\n";
            return 3;
        }
        if (rc.qual.n != rc.seq.n) {
            err = "ERROR: sequence and quality have different length:\n" + string(rc.name.p, rc.name.n) + "\n" +
                  string(rc.seq.p ? rc.seq.p : "", rc.seq.n) + "\n" + string(rc.strand.p, rc.strand.n) + "\n" +
                  string(rc.qual.p ? rc.qual.p : "", rc.qual.n) +
                  "\nYour FASTQ may be invalid, please check the tail of your FASTQ file\n";
            return 3;
        }
        pos = p;
        rc.end = p;
        recs.push_back(rc);
        bases += rc.seq.n;
        reads++;
    }
    return 0;
}

static std::atomic<uint64_t> g_parallel_records{0}; /* records taken from the multi-threaded scan (test hook) */

/* first position >= from that starts a line beginning with '@' (what the sequential scan would take next) */
size_t FastqReader::next_at_line(size_t from) const {
    size_t p = from;
    Line ln;
    for (;;) {
        const size_t at = p;
        const int r = scan_line(p, ln);
        if (r <= 0) return len_;
        if (ln.n > 0 && ln.p[0] == '@') return at;
    }
}

/* Regular files: the stretch of the window that should hold the rest of the batch is cut into one piece per thread.
 * A thread starts at the first line in its piece that looks like a record header ('@' line whose third line
 * starts with '+' and whose second and fourth lines are equally long) and locates the records that start in its
 * piece.  The pieces are then joined in order, but only while thread k's first record is exactly the '@' line the
 * sequential scan would have taken after thread k-1's last record: anything else (a quality line that passed for
 * a header, junk between records, a malformed record) ends the join there and the sequential scan carries on, so
 * the result never differs from the one-thread reader's. */
void FastqReader::scan_parallel(uint64_t& bases, uint64_t max_bases, uint32_t& reads, uint32_t max_reads, vector<Rec>& recs) {
    const uint64_t want = max_bases - bases;
    size_t stretch = (size_t)min<uint64_t>(len_ - pos_, want * 2 + want / 8 + (1u << 20));
    const int T = (int)min<size_t>((size_t)copy_threads_, stretch / parse_min_);
    if (T < 2) return;
    const size_t piece = stretch / T;
    struct Part {
        vector<Rec> recs;
        size_t first = 0, end = 0; /* start of the first record, position behind the last one */
        uint64_t bases = 0;
        uint32_t reads = 0;
        int rc = 0;
    };
    vector<Part> parts(T);
    parallel_run(T, [&](int k) {
        {
            Part& pt = parts[k];
            const size_t lo = pos_ + (size_t)k * piece, hi = k == T - 1 ? pos_ + stretch : lo + piece;
            size_t p = lo;
            if (k > 0) { /* find a header that validates */
                Line ln;
                size_t q = lo;
                if (lo > 0 && win_[lo - 1] != '\n' && win_[lo - 1] != '\r') scan_line(q, ln); /* finish the line we fell into */
                for (;;) {
                    const size_t cand = next_at_line(q);
                    if (cand >= hi) {
                        pt.first = pt.end = hi;
                        pt.rc = -1; /* no record starts in this piece */
                        return;
                    }
                    size_t t = cand;
                    Line l0, l1, l2, l3;
                    const bool ok = scan_line(t, l0) == 1 && scan_line(t, l1) == 1 && scan_line(t, l2) == 1 && scan_line(t, l3) == 1 &&
                                    l2.n > 0 && l2.p[0] == '+' && l1.n == l3.n;
                    if (ok) {
                        p = cand;
                        break;
                    }
                    q = cand;
                    scan_line(q, ln); /* not a header: move past this line */
                }
            }
            pt.first = next_at_line(p);
            size_t pos = p;
            string err;
            pt.rc = scan_records(pos, hi, pt.bases, ~0ull, pt.reads, 0xFFFFFFFFu, pt.recs, err);
            pt.end = pos;
        }
    });
    /* join in order while the pieces line up with the sequential scan */
    for (int k = 0; k < T; k++) {
        Part& pt = parts[k];
        if (pt.rc == -1) continue; /* (empty piece: the next one must still line up with pos_) */
        if (next_at_line(pos_) != pt.first || pt.recs.empty()) return;
        for (const Rec& r : pt.recs) {
            if (bases >= max_bases || reads >= max_reads) return;
            recs.push_back(r);
            g_parallel_records++;
            bases += r.seq.n;
            reads++;
            pos_ = r.end;
        }
        if (pt.rc != 0) return; /* end of input or a malformed record: the sequential scan reports it */
    }
}

bool FastqReader::parse_chunk(int fd, uint64_t file_size, uint64_t a, uint64_t b, bool exact, vector<char>& window,
                              Batch& out, ChunkInfo& info, int threads, const char* mem) {
    info = ChunkInfo();
    if (out.off.empty()) {
        out.off.push_back(0);
        out.name_off.push_back(0);
    }
    if (a >= file_size) {
        info.status = 2;
        return true;
    }
    if (b > file_size) b = file_size;
    const uint64_t w0 = (exact || a == 0) ? a : a - 1; /* (a guessing chunk looks at the byte in front of the cut) */
    if (threads < 1) threads = 1;
    for (uint64_t slack = 4u << 20;; slack *= 4) { /* the last record may run past b: read on, more if it has to be */
        const double t_begin = now_s();
        const uint64_t w1 = min<uint64_t>(file_size, b + slack);
        const size_t n = (size_t)(w1 - w0);
        const char* wp = mem ? mem + w0 : nullptr; /* the bytes [w0, w1): in place when the input is in memory */
        if (!mem) {
            if (window.size() < n) window.resize(n);
            wp = window.data();
            const int T = (int)max<size_t>(1, min<size_t>((size_t)threads, n / (8u << 20)));
            std::atomic<bool> ok{true};
            parallel_run(T, [&](int t) {
                size_t x = n / T * t, e = t == T - 1 ? n : n / T * (t + 1);
                while (x < e) {
                    const ssize_t r = pread(fd, window.data() + x, e - x, (off_t)(w0 + x));
                    if (r <= 0) {
                        ok = false;
                        return;
                    }
                    x += (size_t)r;
                }
            });
            if (!ok) {
                info.status = 4;
                info.err = "reading the input failed (file truncated while it was being read?)";
                return false;
            }
        }
        const double t_read = now_s();
        FastqReader m(wp, n, w1 >= file_size);
        m.copy_threads_ = threads;
        size_t pos = (size_t)(a - w0);
        if (!exact && a > 0) { /* the first header at or behind the cut that validates (see scan_parallel) */
            Line ln;
            if (wp[pos - 1] != '\n' && wp[pos - 1] != '\r') m.scan_line(pos, ln); /* finish the line we fell into */
            for (;;) {
                const size_t cand = m.next_at_line(pos);
                if (cand >= n) {
                    pos = n;
                    break;
                }
                size_t t = cand;
                Line l0, l1, l2, l3;
                const bool good = m.scan_line(t, l0) == 1 && m.scan_line(t, l1) == 1 && m.scan_line(t, l2) == 1 &&
                                  m.scan_line(t, l3) == 1 && l2.n > 0 && l2.p[0] == '+' && l1.n == l3.n;
                if (good || cand >= (size_t)(b - w0)) { /* (beyond the chunk nothing is taken anyway) */
                    pos = cand;
                    break;
                }
                pos = cand;
                m.scan_line(pos, ln); /* not a header: move past this line */
            }
        }
        vector<Rec> recs;
        uint64_t bases = 0;
        uint32_t reads = 0;
        string err;
        const int rc = m.scan_records(pos, (size_t)(b - w0), bases, ~0ull, reads, 0xFFFFFFFFu, recs, err);
        if (rc == 1) continue; /* the window ends inside a record although the file goes on */
        if (!recs.empty()) info.first = w0 + (uint64_t)(recs[0].name.p - wp);
        const double t_scan = now_s();
        m.copy_records(out, recs);
        if (g_timing) {
            const double t_copy = now_s();
            g_chunk_us[0].fetch_add((uint64_t)((t_read - t_begin) * 1e6));
            g_chunk_us[1].fetch_add((uint64_t)((t_scan - t_read) * 1e6));
            g_chunk_us[2].fetch_add((uint64_t)((t_copy - t_scan) * 1e6));
        }
        if (rc == 0) info.next = w0 + pos;
        else if (rc == 2) info.status = 2;
        else {
            info.status = 3;
            info.err = err;
        }
        return true;
    }
}

bool FastqReader::load_chunk_text(int fd, uint64_t file_size, uint64_t a, uint64_t b, uint64_t chunk_bytes, Batch& out, ChunkInfo& info,
                                  const char* mem) {
    info = ChunkInfo();
    out.text_backed = true;
    out.raw_begin = out.raw_len = 0;
    if (a >= file_size) {
        info.status = 2;
        return true;
    }
    if (b > file_size) b = file_size;
    const uint64_t w0 = a == 0 ? 0 : a - 1; /* (the guess looks at the byte in front of the cut) */
    for (uint64_t slack = 4u << 20;; slack *= 4) {
        const uint64_t w1 = min<uint64_t>(file_size, b + slack);
        const size_t n = (size_t)(w1 - w0);
        out.raw.resize_uninit(n);
        char* wp = (char*)out.raw.data();
        if (mem) {
            memcpy(wp, mem + w0, n);
        } else {
            size_t x = 0;
            while (x < n) {
                const ssize_t r = pread(fd, wp + x, n - x, (off_t)(w0 + x));
                if (r <= 0) {
                    info.status = 4;
                    info.err = "reading the input failed (file truncated while it was being read?)";
                    return false;
                }
                x += (size_t)r;
            }
        }
        FastqReader m(wp, n, w1 >= file_size);
        /* The first header at or behind `pos` that VALIDATES (its third line starts with '+', its second and fourth are equally
           long).  parse_chunk's guess stops at the first candidate beyond its chunk, valid or not, because it takes nothing from
           there anyway and the sequencer puts a wrong announcement right; here the position IS the end of this chunk's text and the
           start of the next one's, so both are searched to the end -- through a read of any length (the window grows).
           short_of_window: the search ran into the end of the window although the file goes on */
        bool short_of_window = false;
        auto find_header = [&](size_t pos) -> size_t {
            Line ln;
            if (pos > 0 && wp[pos - 1] != '\n' && wp[pos - 1] != '\r') m.scan_line(pos, ln); /* finish the line we fell into */
            for (;;) {
                const size_t cand = m.next_at_line(pos);
                if (cand >= n) {
                    if (w1 < file_size) short_of_window = true;
                    return n;
                }
                size_t t = cand;
                Line l0, l1, l2, l3;
                const int r0 = m.scan_line(t, l0), r1 = r0 == 1 ? m.scan_line(t, l1) : r0, r2 = r1 == 1 ? m.scan_line(t, l2) : r1,
                          r3 = r2 == 1 ? m.scan_line(t, l3) : r2;
                if (r0 == 0 || r1 == 0 || r2 == 0 || r3 == 0) { /* the window ends inside these four lines and the file goes on */
                    short_of_window = true;
                    return n;
                }
                if (r3 == 1 && l2.n > 0 && l2.p[0] == '+' && l1.n == l3.n) return cand;
                pos = cand;
                m.scan_line(pos, ln); /* not a header: move past this line */
            }
        };
        size_t first = (size_t)(a - w0), next = n;
        if (a > 0) first = find_header(first);
        if (!short_of_window && b < file_size) next = first >= (size_t)(b - w0) ? first : find_header((size_t)(b - w0));
        if (short_of_window) continue;
        if (first > next) first = next;
        if (first >= (size_t)(b - w0) && b < file_size) first = next; /* no record STARTS in this chunk */
        out.raw_begin = first;
        out.raw_len = next - first;
        if (out.raw_len > 0) info.first = w0 + first;
        if (b >= file_size) info.status = 2; /* the end of the input */
        else info.next = w0 + next;
        return true;
    }
}

static double g_t_pull = 0, g_t_scan = 0, g_t_copy = 0;
struct TimingDump {
    ~TimingDump() {
        if (!g_timing) return;
        fprintf(stderr, "reader phases: refill %.3f s, locate %.3f s, copy %.3f s\n", g_t_pull, g_t_scan, g_t_copy);
        fprintf(stderr, "chunk parsers (summed over threads): read %.3f s, locate %.3f s, copy %.3f s; page-locked allocations %.3f s for %.2f GB\n",
                g_chunk_us[0].load() * 1e-6, g_chunk_us[1].load() * 1e-6, g_chunk_us[2].load() * 1e-6,
                g_alloc_seconds_x1000.load() * 1e-6, g_alloc_bytes.load() * 1e-9);
    }
} g_timing_dump;

uint32_t FastqReader::fill(Batch& b, uint64_t max_bases, uint32_t max_reads) {
    if (b.off.empty()) {
        b.off.push_back(0);
        b.name_off.push_back(0);
    }
    uint32_t added = 0;
    vector<Rec> recs;
    uint64_t bases = b.seq.size();
    uint32_t reads = b.n();
    bool end = false;
    while (!end && !malformed_ && bases < max_bases && reads < max_reads) {
        recs.clear();
        if (fd_ >= 0 && max_bases < (1ull << 40) && !eof_) { /* have the stretch this batch needs in the window */
            const uint64_t want = max_bases - bases;
            const size_t stretch = (size_t)min<uint64_t>(file_size_ - (file_pos_ - (len_ - pos_)), want * 2 + want / 8 + (1u << 20));
            if (len_ - pos_ < stretch) {
                if (buf_.size() < stretch + (16u << 20)) { /* grow the window once (it is reused for every batch) */
                    vector<char> nb(stretch + (16u << 20));
                    memcpy(nb.data(), buf_.data() + pos_, len_ - pos_);
                    len_ -= pos_;
                    pos_ = 0;
                    buf_.swap(nb);
                    win_ = buf_.data();
                }
                const double t0 = now_s();
                pull();
                g_t_pull += now_s() - t0;
            }
        }
        const double t1 = now_s();
        /* (only when the batch is cut by bases: with a small read cap the stretch to scan cannot be sized) */
        if (fd_ >= 0 && copy_threads_ > 1 && max_bases < (1ull << 40) && max_reads - reads >= (1u << 24))
            scan_parallel(bases, max_bases, reads, max_reads, recs);
        /* locate the (remaining) records of the current window sequentially */
        string err;
        const int rc = scan_records(pos_, len_, bases, max_bases, reads, max_reads, recs, err);
        const double t2 = now_s();
        copy_records(b, recs); /* before the window moves */
        g_t_scan += t2 - t1;
        g_t_copy += now_s() - t2;
        added += (uint32_t)recs.size();
        if (rc == 3) {
            cerr << err;
            malformed_ = true;
        } else if (rc == 2) {
            end = true;
        } else if (rc == 1 && !pull()) {
            end = true;
        }
    }
    return added;
}

struct ChunkedReader::Impl {
    int fd;
    const char* mem = nullptr;
    uint64_t file_size, chunk_bytes, n_chunks;
    std::function<Item()> acquire;
    std::function<void(Item)> release;
    struct Parsed {
        Item item;
        FastqReader::ChunkInfo info;
    };
    mutex take_mu, parsed_mu;
    condition_variable parsed_cv;
    map<uint64_t, Parsed> parsed;
    uint64_t next_chunk = 0, seq_chunk = 0, expected = 0;
    bool stop = false, done = false;
    vector<std::thread> threads;
    vector<double> busy;
    vector<char> window; /* of the calling thread, for chunks that are parsed again */
    bool as_text = false;
};

ChunkedReader::ChunkedReader(int fd, uint64_t file_size, uint64_t chunk_bytes, int threads, std::function<Item()> acquire,
                             std::function<void(Item)> release, const char* mem, bool as_text) {
    d_ = new Impl;
    d_->fd = fd;
    d_->mem = mem;
    d_->as_text = as_text;
    d_->file_size = file_size;
    d_->chunk_bytes = chunk_bytes ? chunk_bytes : 1;
    d_->n_chunks = (file_size + d_->chunk_bytes - 1) / d_->chunk_bytes;
    d_->acquire = acquire;
    d_->release = release;
    if (threads < 1) threads = 1;
    d_->busy.assign(threads, 0.0);
    for (int t = 0; t < threads; t++)
        d_->threads.emplace_back([this, t]() {
            Impl& D = *d_;
            vector<char> window;
            for (;;) {
                /* a batch first, then the chunk number: chunk numbers are only ever handed to threads that already
                   hold a batch, so the lowest chunk not yet parsed never waits behind later chunks for one (the
                   batches of the chunks in front of it are on their way through the caller's pipeline and come back) */
                Impl::Parsed ps;
                uint64_t k;
                {
                    lock_guard<mutex> g(D.take_mu);
                    if (D.stop || D.next_chunk >= D.n_chunks) break;
                }
                ps.item = D.acquire(); /* may block; not under the lock */
                if (!ps.item.batch) break;
                {
                    lock_guard<mutex> g(D.take_mu);
                    if (D.stop || D.next_chunk >= D.n_chunks) {
                        D.release(ps.item);
                        break;
                    }
                    k = D.next_chunk++;
                }
                ps.item.batch->clear();
                const auto t0 = std::chrono::steady_clock::now();
                if (D.as_text)
                    FastqReader::load_chunk_text(D.fd, D.file_size, k * D.chunk_bytes, (k + 1) * D.chunk_bytes, D.chunk_bytes,
                                                 *ps.item.batch, ps.info, D.mem);
                else
                    FastqReader::parse_chunk(D.fd, D.file_size, k * D.chunk_bytes, (k + 1) * D.chunk_bytes, false, window,
                                             *ps.item.batch, ps.info, 1, D.mem);
                D.busy[t] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                {
                    lock_guard<mutex> g(D.parsed_mu);
                    D.parsed[k] = std::move(ps);
                }
                D.parsed_cv.notify_all();
            }
        });
}

ChunkedReader::~ChunkedReader() {
    {
        lock_guard<mutex> g(d_->take_mu);
        d_->stop = true;
    }
    for (auto& t : d_->threads) t.join();
    for (auto& kv : d_->parsed) d_->release(kv.second.item);
    delete d_;
}

uint64_t ChunkedReader::dead_below() const {
    return d_->seq_chunk >= 2 ? (d_->seq_chunk - 2) * d_->chunk_bytes : 0; /* (seq_chunk - 1 was taken last; one chunk of margin) */
}

double ChunkedReader::busiest_parser_seconds() const {
    double m = 0;
    for (double x : d_->busy) m = max(m, x);
    return m;
}

bool ChunkedReader::next(Item& out) {
    Impl& D = *d_;
    while (!D.done && D.seq_chunk < D.n_chunks) {
        const uint64_t k = D.seq_chunk++;
        Impl::Parsed ps;
        {
            unique_lock<mutex> g(D.parsed_mu);
            D.parsed_cv.wait(g, [&] { return D.parsed.count(k) > 0; });
            ps = std::move(D.parsed[k]);
            D.parsed.erase(k);
        }
        const uint64_t a = k * D.chunk_bytes, b = (k + 1) * D.chunk_bytes;
        const bool has = ps.info.first != FastqReader::ChunkInfo::NONE;
        /* the chunk's first record must be the one its predecessor announced */
        const bool ok = k == 0 || (has ? ps.info.first == D.expected : (ps.info.status == 0 && ps.info.next == D.expected));
        if (!ok) { /* the guess was wrong (or there was nothing to find): parse again from the known start */
            const double t0 = now_s();
            ps.item.batch->clear();
            if (D.expected >= b) { /* the record in front runs across this whole chunk */
                ps.info = FastqReader::ChunkInfo();
                ps.info.next = D.expected;
                ps.item.batch->off.push_back(0);
                ps.item.batch->name_off.push_back(0);
            } else {
                const int hw = effective_cpus();
                FastqReader::parse_chunk(D.fd, D.file_size, max(a, D.expected), b, true, D.window, *ps.item.batch, ps.info,
                                         max(1, min(8, hw / 2)), D.mem);
            }
            t_redo_ += now_s() - t0;
            n_redo_++;
        }
        D.expected = ps.info.next;
        if (ps.info.status == 3) malformed_ = ps.info.err;
        if (ps.info.status == 4) io_error_ = ps.info.err;
        if (ps.info.status != 0) { /* end of input, or a malformed record / a read error ends it */
            D.done = true;
            lock_guard<mutex> g(D.take_mu);
            D.stop = true;
        }
        if (ps.item.batch->has_records()) {
            out = ps.item;
            return true;
        }
        D.release(ps.item);
    }
    return false;
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
    /* the pieces keep their capacity from batch to batch (the Work objects are recycled): fresh multi-megabyte strings
       would be mapped, faulted in page by page and unmapped again for every batch */
    outs.resize(threads);
    for (auto& o : outs) o.clear();
    if (faileds) {
        faileds->resize(threads);
        for (auto& o : *faileds) o.clear();
    }
    /* slices of about equal numbers of bases */
    vector<uint32_t> cut(threads + 1, n);
    cut[0] = 0;
    const uint64_t total = n ? b.off[n] : 0;
    for (int t = 1; t < threads; t++) {
        const uint64_t want = total / threads * t;
        cut[t] = (uint32_t)(std::lower_bound(b.off.begin(), b.off.begin() + n, want) - b.off.begin());
    }
    parallel_run(threads, [&](int t) {
        const size_t want = (size_t)((b.off[cut[t + 1]] - b.off[cut[t]]) * 2 + (uint64_t)(cut[t + 1] - cut[t]) * 128 + 64);
        if (outs[t].capacity() < want) outs[t].reserve(want + want / 4);
        format_range(b, res, cut[t], cut[t + 1], outs[t], faileds ? &(*faileds)[t] : nullptr, fl);
    });
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
        const char* name = b.name_ptr(i);
        const uint32_t nl = b.name_len[i], sl = b.strand_len[i];
        const char* strand = b.strand_ptr(i);
        const uint8_t* s = b.seq_ptr(i);
        const uint8_t* q = b.qual_ptr(i);
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
/* test hook: the whole file through repeated fill() calls of the given caps, concatenated */
void* fplh_batch_read_all(const char* path, uint64_t max_bases, uint32_t max_reads) {
    fplh::FastqReader rd(path);
    if (!rd.ok()) return nullptr;
    fplh::Batch* all = new fplh::Batch();
    all->off.push_back(0);
    all->name_off.push_back(0);
    for (;;) {
        fplh::Batch t;
        if (rd.fill(t, max_bases, max_reads) == 0) break;
        const size_t o = all->seq.size();
        all->seq.resize_uninit(o + t.seq.size());
        all->qual.resize_uninit(o + t.seq.size());
        memcpy(all->seq.data() + o, t.seq.data(), t.seq.size());
        memcpy(all->qual.data() + o, t.qual.data(), t.seq.size());
        for (uint32_t i = 0; i < t.n(); i++) all->off.push_back(o + t.off[i + 1]);
    }
    return all;
}
void* fplh_batch_read_chunked(const char* path, uint64_t chunk_bytes, int threads, uint64_t* chunks_parsed_again) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return nullptr;
    struct stat st;
    if (fstat(fd, &st) != 0) {
        close(fd);
        return nullptr;
    }
    fplh::Batch* all = new fplh::Batch();
    all->off.push_back(0);
    all->name_off.push_back(0);
    {
        /* a small pool of batches, as the CLI's Work objects are */
        std::mutex mu;
        std::condition_variable cv;
        std::vector<fplh::Batch*> pool;
        for (int i = 0; i < threads + 2; i++) pool.push_back(new fplh::Batch());
        std::vector<fplh::Batch*> owned = pool;
        auto acquire = [&]() {
            std::unique_lock<std::mutex> g(mu);
            cv.wait(g, [&] { return !pool.empty(); });
            fplh::ChunkedReader::Item it;
            it.batch = pool.back();
            pool.pop_back();
            return it;
        };
        auto release = [&](fplh::ChunkedReader::Item it) {
            {
                std::lock_guard<std::mutex> g(mu);
                pool.push_back(it.batch);
            }
            cv.notify_all();
        };
        {
            /* FPLH_CHUNK_MEM (test hook): the parsers take the file's bytes from a mapping, as they take inflated gzip members */
            const char* mem = nullptr;
            if (getenv("FPLH_CHUNK_MEM") && st.st_size > 0) {
                void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m != MAP_FAILED) mem = (const char*)m;
            }
            fplh::ChunkedReader cr(mem ? -1 : fd, (uint64_t)st.st_size, chunk_bytes, threads, acquire, release, mem);
            fplh::ChunkedReader::Item it;
            while (cr.next(it)) {
                const fplh::Batch& t = *it.batch;
                const size_t o = all->seq.size();
                all->seq.resize_uninit(o + t.seq.size());
                all->qual.resize_uninit(o + t.seq.size());
                memcpy(all->seq.data() + o, t.seq.data(), t.seq.size());
                memcpy(all->qual.data() + o, t.qual.data(), t.seq.size());
                const size_t to = all->text.size();
                all->text.insert(all->text.end(), t.text.begin(), t.text.end());
                for (uint32_t i = 0; i < t.n(); i++) {
                    all->off.push_back(o + t.off[i + 1]);
                    all->name_off.push_back(to + t.name_off[i + 1]);
                    all->name_len.push_back(t.name_len[i]);
                    all->strand_len.push_back(t.strand_len[i]);
                }
                release(it);
            }
            if (chunks_parsed_again) *chunks_parsed_again = cr.chunks_parsed_again();
        }
        for (fplh::Batch* b : owned) delete b;
    }
    close(fd);
    return all;
}
/* test hook: the whole (regular, uncompressed) file through the chunk LOADER (text-backed batches, ChunkedReader as_text): the file
   offsets [begin, end) of every chunk's records, in input order, into ranges[2 k], ranges[2 k + 1]; returns the number of chunks
   that hold records (-1: the file could not be read; more than `cap` chunks: only the first `cap` are stored) */
int64_t fplh_text_chunk_ranges(const char* path, uint64_t chunk_bytes, int threads, uint64_t* ranges, uint64_t cap) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return -1;
    struct stat st;
    if (fstat(fd, &st) != 0) {
        close(fd);
        return -1;
    }
    int64_t n = 0;
    {
        std::mutex mu;
        std::condition_variable cv;
        std::vector<fplh::Batch*> pool;
        for (int i = 0; i < threads + 2; i++) pool.push_back(new fplh::Batch());
        std::vector<fplh::Batch*> owned = pool;
        auto acquire = [&]() {
            std::unique_lock<std::mutex> g(mu);
            cv.wait(g, [&] { return !pool.empty(); });
            fplh::ChunkedReader::Item it;
            it.batch = pool.back();
            pool.pop_back();
            return it;
        };
        auto release = [&](fplh::ChunkedReader::Item it) {
            {
                std::lock_guard<std::mutex> g(mu);
                pool.push_back(it.batch);
            }
            cv.notify_all();
        };
        {
            const char* mem = nullptr;
            if (getenv("FPLH_CHUNK_MEM") && st.st_size > 0) {
                void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m != MAP_FAILED) mem = (const char*)m;
            }
            fplh::ChunkedReader cr(mem ? -1 : fd, (uint64_t)st.st_size, chunk_bytes, threads, acquire, release, mem, true);
            fplh::ChunkedReader::Item it;
            uint64_t at = 0; /* (file offset of a chunk's text: where the one in front of it ended -- checked by the caller) */
            while (cr.next(it)) {
                const fplh::Batch& t = *it.batch;
                /* the loader keeps the window's bytes [w0, w1): raw_begin counts from w0, which the batch does not say; the text
                   itself does -- compare it with the file at the running offset (the ranges must be contiguous for a regular file) */
                uint64_t found = ~0ull;
                if (t.raw_len > 0) {
                    std::vector<char> buf(t.raw_len);
                    /* chunks follow one another: try the running offset first, then look ahead (junk lines between records) */
                    for (uint64_t o = at; o + t.raw_len <= (uint64_t)st.st_size && found == ~0ull; o++) {
                        if (pread(fd, buf.data(), t.raw_len, (off_t)o) != (ssize_t)t.raw_len) break;
                        if (memcmp(buf.data(), t.raw.data() + t.raw_begin, t.raw_len) == 0) found = o;
                        if (o - at > (1u << 16)) break;
                    }
                }
                if ((uint64_t)n < cap) {
                    ranges[2 * n] = found;
                    ranges[2 * n + 1] = found == ~0ull ? ~0ull : found + t.raw_len;
                }
                if (found != ~0ull) at = found + t.raw_len;
                n++;
                release(it);
            }
        }
        for (fplh::Batch* b : owned) delete b;
    }
    close(fd);
    return n;
}
/* bench / test helper: a CSR batch as a FASTQ file ("@<prefix><i>" names, "+" strand lines); the text is composed on
   `threads` threads, slice by slice, and written in order.  0 on success. */
int fplh_write_fastq(const char* path, const uint8_t* seq, const uint8_t* qual, const uint64_t* off, uint32_t n,
                     const char* prefix, int threads) {
    return fplh_write_fastq_ex(path, seq, qual, off, n, prefix, threads, 0);
}
int fplh_write_fastq_ex(const char* path, const uint8_t* seq, const uint8_t* qual, const uint64_t* off, uint32_t n,
                        const char* prefix, int threads, int append) {
    FILE* f = fopen(path, append ? "ab" : "wb");
    if (!f) return -1;
    if (threads < 1) threads = 1;
    const std::string pre = prefix ? prefix : "r";
    const uint32_t per_round = 65536u * (uint32_t)threads; /* bounds the text held in memory */
    int rc = 0;
    for (uint32_t r0 = 0; r0 < n && rc == 0; r0 += per_round) {
        const uint32_t r1 = (uint32_t)std::min<uint64_t>(n, (uint64_t)r0 + per_round);
        std::vector<std::string> parts((size_t)threads);
        fplh::parallel_run(threads, [&](int t) {
            const uint32_t a = r0 + (uint32_t)((uint64_t)(r1 - r0) * t / threads), b = r0 + (uint32_t)((uint64_t)(r1 - r0) * (t + 1) / threads);
            std::string& s = parts[t];
            s.reserve((size_t)(2 * (off[b] - off[a]) + (uint64_t)(b - a) * (pre.size() + 20)));
            for (uint32_t i = a; i < b; i++) {
                s += '@';
                s += pre;
                s += std::to_string(i);
                s += '\n';
                s.append((const char*)seq + off[i], (size_t)(off[i + 1] - off[i]));
                s += "\n+\n";
                s.append((const char*)qual + off[i], (size_t)(off[i + 1] - off[i]));
                s += '\n';
            }
        });
        for (auto& s : parts)
            if (!s.empty() && fwrite(s.data(), 1, s.size(), f) != s.size()) rc = -2;
    }
    if (fclose(f) != 0) rc = -2;
    return rc;
}
/* test hook: read the whole file; 1 (and the message) when the input could not be read / decompressed to its end */
int fplh_read_error(const char* path, char* msg, int msg_len) {
    fplh::FastqReader rd(path);
    if (!rd.ok()) return -1;
    for (;;) {
        fplh::Batch t;
        if (rd.fill(t, 64u << 20, 0x3FFFFFFFu) == 0) break;
    }
    if (!rd.input_error()) return 0;
    if (msg && msg_len > 0) snprintf(msg, (size_t)msg_len, "%s", rd.input_error_text().c_str());
    return 1;
}
uint64_t fplh_parallel_records(void) { return fplh::g_parallel_records.exchange(0); }
uint64_t fplh_gz_members(void) { return fplh::GzMembers::delivered.exchange(0); }
/* test hook: descriptor of the in-memory file with the inflated text of a multi-member gzip file, or -1 */
char* fplh_gunzip_to_memory(const char* path, int threads, uint64_t max_bytes, uint64_t* size_out, uint64_t* reserved) {
    return fplh::gunzip_members_to_memory(path, threads, max_bytes, size_out, reserved);
}
int fplh_have_libdeflate(void) { return fplh::have_libdeflate() ? 1 : 0; }
void fplh_gunzip_release(char* base, uint64_t reserved) {
    if (base) munmap(base, (size_t)reserved);
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
