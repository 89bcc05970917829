/*
 * split.h -- gzip members for the output files and the --split / --split_by_lines writer
 * (reference src/threadconfig.cpp:72-120, src/seprocessor.cpp:297-316), shared by the CLI and the host tests.
 */
#ifndef FPLH_SPLIT_H
#define FPLH_SPLIT_H

#include <stdio.h>
#include <stdlib.h>

#include <sys/uio.h>

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <ostream>
#include <string>
#include <thread>
#include <vector>

namespace fplh {

/* one complete gzip member holding `in` (any gzip reader takes a concatenation of members as one stream).  Whole-buffer
 * work: libdeflate does it (the library the reference's Writer uses, src/writer.cpp:110-133; loaded at run time when
 * the system has libdeflate.so.0), zlib otherwise. */
void gzip_into(const std::string& in, int level, std::string& out);
std::string gzip_member(const std::string& in, int level);
/* bytes without std::vector's zero fill (an inflate target is overwritten anyway, and its size is a guess) */
struct RawBuf {
    char* p = nullptr;
    size_t n = 0, cap = 0;
    RawBuf() = default;
    RawBuf(const RawBuf&) = delete;
    RawBuf& operator=(const RawBuf&) = delete;
    RawBuf(RawBuf&& o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr, o.n = o.cap = 0; }
    RawBuf& operator=(RawBuf&& o) noexcept {
        swap(o);
        return *this;
    }
    ~RawBuf() { free(p); }
    void swap(RawBuf& o) {
        std::swap(p, o.p);
        std::swap(n, o.n);
        std::swap(cap, o.cap);
    }
    void reserve(size_t c) {
        if (c > cap) {
            p = (char*)realloc(p, c);
            cap = c;
        }
    }
    void release() {
        free(p);
        p = nullptr;
        n = cap = 0;
    }
    char* data() { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    void clear() { n = 0; }
};
/* the gzip member that starts at in[0]: inflated into out, *consumed = its compressed length.  hint = a guess of the
   inflated size (0 = none).  1 = a whole member, 0 = not a (complete, undamaged) member, 2 = it inflates to more than
   cap bytes */
int gunzip_member(const unsigned char* in, size_t in_len, RawBuf& out, size_t cap, size_t* consumed, size_t hint = 0);
bool have_libdeflate();
/* one whole member straight into caller-owned memory (libdeflate only): 1 = done (*consumed input bytes, *produced output
   bytes), 2 = out_cap is too small, 0 = damaged / truncated, -1 = libdeflate is not there */
int gunzip_member_into(const unsigned char* in, size_t in_len, char* out, size_t out_cap, size_t* consumed, size_t* produced);

/* --split / --split_by_lines.  Each of the reference's workers owns a writer and walks through the file numbers
 * t, t + T, t + 2T, ... as its current file fills up (ThreadConfig::initWriterForSplit / markProcessed /
 * writeEmptyFilesForSplitting, src/threadconfig.cpp:72-120); packs of PACK_SIZE = 16 reads reach the workers
 * round-robin (src/seprocessor.cpp:343-378), so which file a read lands in is a function of its input index.  The
 * one writer thread of this host replays that, pack by pack: write(t, text) is the worker's
 * getWriter1()->writeString(outstr), mark(t, n) its markProcessed(n), close() the ThreadConfig destructors.
 * Pinned against the real ThreadConfig / Writer objects of the reference (tests/test_host_split.py).
 *
 * What the pin leaves open is timing, not logic: with --split, a worker whose files are used up while
 * number % threads != 0 and id >= number % threads gets mCanBeStopped (src/threadconfig.cpp:103-107) and leaves
 * its loop the next time it finds its input queue EMPTY (src/seprocessor.cpp:432-441) -- packs that reach it later are
 * never processed, so how many reads the reference loses there depends on how far its reader thread is ahead.  This
 * writer reproduces the reference run in which the reader stays ahead (the worker never sees an empty queue before
 * the input ends): nothing is lost and the worker's last file takes the rest, exactly what the real objects do when
 * every pack is handed to them (the S_* commands of the test harness). */
class SplitOutput {
   public:
    SplitOutput(const std::string& out, int digits, int workers, bool by_lines, int number, long size, int gz_level);
    ~SplitOutput();
    void write(int t, const std::string& text);
    /* the same bytes as a gather list (plain files only): what is pending for worker t goes out first, then the list, by
       writev -- no copy of the text in user space */
    void write_gather(int t, struct iovec* iov, size_t cnt);
    void mark(int t, long reads);
    void close();
    bool gzipped() const { return gz_; }
    std::vector<std::string> names; /* the files opened, in the order they were opened (threaded: per worker, then merged) */

    /* One thread per worker -- the reference's writers ARE per worker (ThreadConfig owns its Writer, src/threadconfig.cpp:72-87),
     * and one file on tmpfs takes 6-9 GB/s whoever writes it while fifteen files take 60 (tools/file_write_probe.cpp).
     * start_threads() once; post(t, job) hands worker t's thread a job that calls write / write_gather / mark for worker t ONLY
     * (jobs of one worker run in the order they were posted, so every file gets the bytes the one-thread replay gives it);
     * close() drains the queues, runs the workers' clean-up and joins. */
    void start_threads();
    void post(int t, std::function<void()> job);
    bool threaded() const { return !threads_.empty(); }

   private:
    struct Worker {
        int working = 0;
        long current = 0;
        int fd = -1;
        bool wrote = false;
        std::string pending;
        std::vector<std::string> opened;
        /* threaded mode */
        std::mutex m;
        std::condition_variable cv;
        std::deque<std::function<void()>> q;
        bool stop = false;
    };
    void flush(Worker& w);
    void shut(Worker& w);
    void open(Worker& w);
    void finish(Worker& w);
    void put(Worker& w, const char* p, size_t n);
    std::string out_;
    int digits_, T_;
    bool by_lines_;
    int number_;
    long size_;
    int level_;
    bool gz_ = false;
    std::vector<Worker> w_;
    std::vector<std::thread> threads_;
    bool closed_ = false;
};

/* --adapter_fasta: FastaReader + Options::loadFastaAdapters (src/fastareader.cpp:5-101, src/options.cpp:39-66).
 * load_fasta_contigs restates the reader byte for byte (pinned against the real FastaReader, tests/test_host_split.py);
 * load_fasta_adapters keeps the sequences of >= 6 characters in header order -- the order trimByMultiSequences visits
 * them in -- and reports the skipped ones on `log` like the reference.  false + err when the file cannot be read. */
bool load_fasta_contigs(const std::string& path, std::map<std::string, std::string>& contigs, std::string& err);
bool load_fasta_adapters(const std::string& path, std::vector<std::string>& adapters, std::ostream* log, std::string& err);

}  // namespace fplh

extern "C" {
/* test hook: "header\tsequence\n" for every contig in map order; malloc'ed, free with fplh_free */
int fplh_load_fasta(const char* path, char** out, unsigned long long* out_len);
/* test hook: replay n_packs packs (worker, reads, passing reads, text) through a SplitOutput; returns the number of
   files it opened */
int fplh_split_replay(const char* out, int digits, int workers, int by_lines, int number, long size, int gz_level,
                      unsigned n_packs, const int* worker, const long* reads, const long* passed, const char* const* texts);
}
#endif
