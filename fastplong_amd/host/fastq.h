/*
 * fastq.h -- the host keeps the FASTQ reader/writer role of the reference
 * (src/fastqreader.cpp:219-347, src/writer.cpp, Read::appendToString* src/read.cpp:119-173) but
 * re-shaped for the device: records are parsed straight into CSR batches (all bases
 * concatenated, all qualities concatenated, byte offsets), names and strand lines stay on the
 * host, and output is formatted from the per-read result records.
 */
#ifndef FPLH_FASTQ_H
#define FPLH_FASTQ_H

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <functional>
#include <string>
#include <vector>

#include "fastplong_amd.h"

namespace fplh {

/* Persistent worker threads for the short parallel phases of the host pipeline (window refill, record location,
 * line copies, output formatting, gzip members): a phase lasts a few milliseconds, so starting threads for it costs
 * as much as the work.  run(n, fn) executes fn(0) .. fn(n-1) on the workers and the calling thread and returns when
 * all are done; any number of threads may call it at the same time. */
void parallel_run(int tasks, const std::function<void(int)>& fn);
/* CPUs this process may actually use: the hardware threads, cut down to the scheduler affinity mask and to the cgroup's
   CPU bandwidth quota (cpu.max / cpu.cfs_quota_us) -- a container on a 256-thread node with a 16-CPU quota is throttled,
   not sped up, by 64 busy threads (FPLH_CPUS overrides) */
int effective_cpus();
uint64_t memory_budget(); /* bytes this process may still take: MemAvailable and the cgroup's limit */

/* growable byte array without the zero fill of std::vector::resize (batches are hundreds of megabytes and
 * every byte is overwritten by the parser's copy threads).  The memory comes from a process-wide allocator pair the
 * host may replace ONCE, before the first batch exists: the CLI installs fpl_host_alloc / fpl_host_free, so that the
 * CSR arrays are page-locked and the GPU's DMA engines read them in place (no staging copy). */
class ByteBuf {
   public:
    typedef void* (*AllocFn)(size_t);
    typedef void (*FreeFn)(void*);
    static void set_allocator(AllocFn a, FreeFn f);
    /* ... and carve the usual buffers out of ONE allocation of n_blocks x block_bytes made right away (page-locking
       memory is slow and does not scale over threads: 24 parser threads allocating their first batches spent 13 s in
       it for 5 GB); a buffer that needs more than a block, or finds none free, falls back to the allocator */
    static void set_arena(size_t block_bytes, size_t n_blocks);
    static void release_arena(); /* give the arena back (no buffer of it may be used afterwards) */
    ByteBuf() = default;
    ByteBuf(const ByteBuf&) = delete;
    ByteBuf& operator=(const ByteBuf&) = delete;
    ~ByteBuf();
    uint8_t* data() { return p_; }
    const uint8_t* data() const { return p_; }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    void clear() { n_ = 0; }
    const uint8_t* begin() const { return p_; }
    const uint8_t* end() const { return p_ + n_; }
    void reserve(size_t c);
    void resize_uninit(size_t n) {
        reserve(n);
        n_ = n;
    }
   private:
    uint8_t* p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
};

struct Batch {
    ByteBuf seq, qual;                /* CSR payload handed to fpl_process_batch */
    std::vector<uint64_t> off;        /* n + 1 */
    std::vector<char> text;           /* name and strand lines, back to back */
    std::vector<uint64_t> name_off;   /* n + 1 offsets into text for names   */
    std::vector<uint32_t> name_len, strand_len; /* strand line follows the name in `text` */
    /* A TEXT-BACKED batch (--device_parse): `raw` holds a stretch of the file as it lies there, raw[raw_begin, raw_begin +
       raw_len) are whole records; the DEVICE finds them (fpl_process_text_async) and the caller then fills off / name_len /
       strand_len and `line` (four per read: where its name, bases, '+' line and qualities start in raw) from what comes
       back -- no base is copied on the host, the output is formatted out of raw. */
    ByteBuf raw;
    uint64_t raw_begin = 0, raw_len = 0;
    std::vector<uint32_t> line;
    bool text_backed = false;
    uint32_t n() const { return off.empty() ? 0 : (uint32_t)(off.size() - 1); }
    bool has_records() const { return n() > 0 || (text_backed && raw_len > 0); }
    /* the four lines of read i, whichever form the batch has */
    const char* name_ptr(uint32_t i) const { return text_backed ? (const char*)raw.data() + line[4 * (size_t)i] : text.data() + name_off[i]; }
    const char* strand_ptr(uint32_t i) const {
        return text_backed ? (const char*)raw.data() + line[4 * (size_t)i + 2] : text.data() + name_off[i] + name_len[i];
    }
    const uint8_t* seq_ptr(uint32_t i) const { return text_backed ? raw.data() + line[4 * (size_t)i + 1] : seq.data() + off[i]; }
    const uint8_t* qual_ptr(uint32_t i) const { return text_backed ? raw.data() + line[4 * (size_t)i + 3] : qual.data() + off[i]; }
    /* text-backed: off / name_len / strand_len from the line starts the device found (n records, offsets relative to
       raw_begin as fpl_wait_text hands them out) */
    void adopt_lines(const uint32_t* line_starts, uint32_t n_records);
    void clear();
};

/* A gzip file that is a concatenation of members (bgzip, `cat` of per-chunk files as sequencers write them, the
 * outputs of fastp / fastplong / this host) inflated on several threads: see fastq.cpp. */
class GzMembers;

/* Plain or gzip FASTQ (zlib).  Line splitting follows FastqReader::getLine: a line ends at
 * '\r' or '\n', "\r\n" counts once; records whose header does not start with '@' are skipped
 * line by line; a malformed record ends the input (src/fastqreader.cpp:326-341). */
class FastqReader {
   public:
    explicit FastqReader(const std::string& path);
    /* the same reader over bytes already in memory (parse_chunk): no refill; at_eof = the bytes end the input */
    FastqReader(const char* data, size_t len, bool at_eof);
    ~FastqReader();

    /* Chunk-parallel reading of a regular, uncompressed file: the file is cut at fixed byte offsets and a record
     * belongs to the chunk its '@' falls into, so any number of threads can parse different chunks at the same
     * time.  A chunk that does not start the file has to GUESS where its first record begins (the first '@' line at
     * or behind the cut whose third line starts with '+' and whose second and fourth lines are equally long); the
     * caller checks every guess against what the chunk in front of it reports as the next record (ChunkInfo::next)
     * and parses the chunk again with exact = true when they differ, so the records are always those the
     * sequential reader finds.  Reads [a, b) of the file behind fd (plus the tail of the last record) into
     * `window`, appends the records to `out`. */
    struct ChunkInfo {
        static constexpr uint64_t NONE = ~0ull;
        uint64_t first = NONE; /* file offset of the first record taken, NONE when the chunk holds none */
        uint64_t next = NONE;  /* file offset at which the next record starts: the first '@' line behind the last
                                  record taken that lies at or behind b (NONE after the end of input / a bad record) */
        int status = 0;        /* 0 = more input follows, 2 = end of input reached, 3 = malformed record, 4 = read error (message in err) */
        std::string err;
    };
    /* (mem != nullptr: the input's bytes [0, file_size) are in memory there and are parsed in place; fd is not used) */
    static bool parse_chunk(int fd, uint64_t file_size, uint64_t a, uint64_t b, bool exact, std::vector<char>& window,
                            Batch& out, ChunkInfo& info, int threads, const char* mem = nullptr);
    /* The same chunk as TEXT for the device to parse (Batch::raw): the bytes [a - 1, b + slack) are read into out.raw and only
     * the two ends are looked at -- the first record of this chunk and the first record of the next, both by the guess
     * parse_chunk makes -- so the records of the text are exactly those parse_chunk(.., exact = false) would take.  No line
     * in between is scanned and no base copied a second time. */
    static bool load_chunk_text(int fd, uint64_t file_size, uint64_t a, uint64_t b, uint64_t chunk_bytes, Batch& out, ChunkInfo& info,
                                const char* mem = nullptr);
    bool ok() const { return fp_ != nullptr; }
    /* append records until the batch holds >= max_bases bases or max_reads reads; returns the
     * number of records appended (0 at end of input) */
    uint32_t fill(Batch& b, uint64_t max_bases, uint32_t max_reads);
    bool malformed() const { return malformed_; }
    /* decompression / read errors end the input like the end of the file does, but are remembered: the reference
       aborts there (error_exit in src/fastqreader.cpp:92-137), so the caller must not report success */
    bool input_error() const { return !io_error_.empty(); }
    const std::string& input_error_text() const { return io_error_; }
    /* offset, in the (uncompressed) input, of the byte behind the last record handed out */
    uint64_t consumed() const { return pulled_ - (len_ - pos_); }

    void set_copy_threads(int t) { copy_threads_ = t < 1 ? 1 : t; }

   private:
    struct Line {
        const char* p;
        size_t n;
    };
    struct Rec {
        Line name, seq, strand, qual;
        size_t end; /* window position behind the record's last line terminator */
    };
    /* The input is scanned in place (one SIMD pass for both terminators) in one large window that is refilled
     * behind the unconsumed tail: by gzread for gzip and pipes, by parallel pread for a regular uncompressed
     * file (whose window grows to hold a whole batch, so that the scan can be split over threads too).  fill() first only LOCATES the records of a stretch of the window, then copies their
     * lines into the batch on copy_threads_ threads.  scan_line: 1 = a line, 0 = the window ends inside the
     * line and more input exists, -1 = end of input with nothing left. */
    int scan_line(size_t& pos, Line& ln) const;
    int scan_records(size_t& pos, size_t start_limit, uint64_t& bases, uint64_t max_bases, uint32_t& reads,
                     uint32_t max_reads, std::vector<Rec>& recs, std::string& err) const;
    size_t next_at_line(size_t from) const;
    void scan_parallel(uint64_t& bases, uint64_t max_bases, uint32_t& reads, uint32_t max_reads, std::vector<Rec>& recs);
    bool pull(); /* stream mode: keep [pos_, len_), read more behind it (growing the window when a record fills it) */
    void copy_records(Batch& b, const std::vector<Rec>& recs) const;
    void* fp_ = nullptr;       /* gzFile (gzip, pipes) or a non-null token (regular file, fd_ / members_) */
    GzMembers* members_ = nullptr; /* multi-member gzip file: parallel inflate */
    const char* win_ = nullptr; /* the window: buf_.data() */
    std::vector<char> buf_;
    int fd_ = -1;              /* regular uncompressed file: refilled with parallel pread */
    uint64_t file_size_ = 0, file_pos_ = 0, pulled_ = 0;
    size_t pos_ = 0, len_ = 0;
    int copy_threads_ = 1;
    size_t parse_min_ = 8u << 20; /* smallest piece worth a scanning thread (FPLH_PARSE_MIN: test hook) */
    bool eof_ = false, malformed_ = false;
    bool mem_ = false; /* memory mode: the window is the caller's */
    std::string io_error_, path_;
};

/* The chunk-parallel reader built on FastqReader::parse_chunk: `threads` parser threads take chunks of
 * `chunk_bytes` in file order, each into a Batch it obtains from `acquire` (blocking; the caller's pool bounds how far
 * the parsers run ahead); next() hands the non-empty batches back in input order after checking every chunk's guessed
 * start against its predecessor (a chunk whose guess was wrong is parsed again from the known offset, on the calling
 * thread), and returns the batches it does not pass on through `release`.  The records are exactly those the
 * sequential FastqReader finds (tests/test_host_io.py::test_chunked_reader_equals_sequential). */
class ChunkedReader {
   public:
    struct Item {
        Batch* batch = nullptr;
        void* token = nullptr; /* the caller's handle for the batch (e.g. the Work object it lives in) */
    };
    /* as_text: the parsers only LOAD their chunks (FastqReader::load_chunk_text, text-backed batches) */
    ChunkedReader(int fd, uint64_t file_size, uint64_t chunk_bytes, int threads, std::function<Item()> acquire,
                  std::function<void(Item)> release, const char* mem = nullptr, bool as_text = false);
    ~ChunkedReader();
    /* the next batch in input order; false at the end of the input (or behind a malformed record / read error) */
    bool next(Item& out);
    const std::string& malformed_text() const { return malformed_; } /* "" or the reference's message for a bad record */
    const std::string& io_error_text() const { return io_error_; }
    /* input offset below which no parser will look again (the chunks in front of the one the sequencer took last): the
       owner of an in-memory input may give those pages back */
    uint64_t dead_below() const;
    uint64_t chunks_parsed_again() const { return n_redo_; }
    double redo_seconds() const { return t_redo_; }
    double busiest_parser_seconds() const;

   private:
    struct Impl;
    Impl* d_;
    std::string malformed_, io_error_;
    uint64_t n_redo_ = 0;
    double t_redo_ = 0;
};

/* multi-member gzip -> the inflated text in anonymous memory (fastq.cpp); nullptr when that does not apply */
char* gunzip_members_to_memory(const std::string& path, int threads, uint64_t max_bytes, uint64_t* size_out, uint64_t* reserved);

/* --break / --mask: the outcome list of a batch (fpl_get_fragments: sorted by read, then seq_no) and where
 * each read's records start */
struct FragmentList {
    std::vector<fpl_fragment> frags;
    std::vector<fpl_region> regs;
    std::vector<uint32_t> first; /* n + 1: fragments of read i are [first[i], first[i+1]) */
    void index(uint32_t n_reads);
};

/* Serialize what src/seprocessor.cpp:265-281 writes for one batch: passing fragments to `out`
 * (name with the split prefix when the read was broken), and -- when failed != nullptr -- the
 * trimmed read with its tag for reads that produced exactly one fragment and failed. */
void format_batch(const Batch& b, const fpl_read_result* res, std::string& out, std::string* failed);
/* the same for reads [first, last) only (appends) */
void format_range(const Batch& b, const fpl_read_result* res, uint32_t first, uint32_t last, std::string& out,
                  std::string* failed, const FragmentList* fl = nullptr);
/* the same on `threads` host threads: piece t of outs / faileds holds the text of the t-th slice of the
 * batch, so writing the pieces in order gives format_batch's output */
void format_batch_parallel(const Batch& b, const fpl_read_result* res, int threads, std::vector<std::string>& outs,
                           std::vector<std::string>* faileds, const FragmentList* fl = nullptr);

}  // namespace fplh

extern "C" {
/* test hooks: parse a FASTQ file into CSR arrays; format a batch from result records */
void* fplh_batch_read(const char* path, uint64_t max_bases, uint32_t max_reads);
void* fplh_batch_read_all(const char* path, uint64_t max_bases, uint32_t max_reads);
/* test hook: the whole (regular, uncompressed) file through the chunk-parallel reader, concatenated */
void* fplh_batch_read_chunked(const char* path, uint64_t chunk_bytes, int threads, uint64_t* chunks_parsed_again);
/* test hook: the file through the chunk LOADER (text-backed batches): file offsets [begin, end) of every chunk's records */
int64_t fplh_text_chunk_ranges(const char* path, uint64_t chunk_bytes, int threads, uint64_t* ranges, uint64_t cap);
int fplh_read_error(const char* path, char* msg, int msg_len);
int fplh_write_fastq(const char* path, const uint8_t* seq, const uint8_t* qual, const uint64_t* off, uint32_t n,
                     const char* prefix, int threads);
/* append != 0: the records go behind what the file holds (bench.py builds its N-GPU input out of N copies of a batch, each
   with a prefix of its own) */
int fplh_write_fastq_ex(const char* path, const uint8_t* seq, const uint8_t* qual, const uint64_t* off, uint32_t n,
                        const char* prefix, int threads, int append);
uint64_t fplh_parallel_records(void); /* records the multi-threaded scan contributed since the last call */
uint64_t fplh_gz_members(void);       /* gzip members inflated on the worker pool since the last call */
char* fplh_gunzip_to_memory(const char* path, int threads, uint64_t max_bytes, uint64_t* size_out, uint64_t* reserved);
void fplh_gunzip_release(char* base, uint64_t reserved);
int fplh_have_libdeflate(void);
uint32_t fplh_batch_n(void* b);
uint64_t fplh_batch_bytes(void* b);
const uint8_t* fplh_batch_seq(void* b);
const uint8_t* fplh_batch_qual(void* b);
const uint64_t* fplh_batch_off(void* b);
void fplh_batch_free(void* b);
/* returns malloc'ed buffers the caller frees with fplh_free */
int fplh_format_batch(void* b, const fpl_read_result* res, char** out, uint64_t* out_len, char** failed,
                      uint64_t* failed_len);
/* the same with a --break / --mask fragment list (n_frags records sorted by read / seq_no, their regions) */
int fplh_format_batch_fragments(void* b, const fpl_read_result* res, const fpl_fragment* frags, uint32_t n_frags,
                                const fpl_region* regs, uint32_t n_regs, int threads, char** out, uint64_t* out_len,
                                char** failed, uint64_t* failed_len);
int fplh_format_batch_parallel(void* b, const fpl_read_result* res, int threads, char** out, uint64_t* out_len,
                               char** failed, uint64_t* failed_len);
void fplh_free(void* p);
}
#endif
