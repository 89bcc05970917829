/*
 * fastplong_amd.h -- C-ABI of the MI355X-native per-read hot path.
 *
 * The reference (OpenGene/fastplong v0.4.1) has no plugin/FFI interface; the seam this
 * library replaces is the C++ call
 *     bool SingleEndProcessor::processSingleEnd(ReadPack*, ThreadConfig*)
 *         (reference src/seprocessor.h:30, body src/seprocessor.cpp:180-329)
 * together with the per-thread accumulators it updates (Stats x2 + FilterResult,
 * src/threadconfig.h:16-59) and the serial merge that follows the join
 * (Stats::merge src/stats.cpp:1013-1082, FilterResult::merge src/filterresult.cpp:28-61).
 *
 * Shape of the replacement: the caller hands over a *batch* of variable-length reads in CSR
 * form (all bases concatenated, all quality bytes concatenated, n+1 byte offsets) and gets
 * back one fixed-size record per read (bounds of the surviving fragment(s) + filter code)
 * while all additive counters stay resident on the device until fpl_get_counters().
 * Bases are never written back: the host slices its own copy using the returned offsets.
 *
 * Conventions: plain C types only; the caller owns every buffer it passes in; functions
 * return 0 or a negative FPL_ERR_* code and never call exit(); one context per device, one
 * batch in flight per context; contexts on different devices may be driven from different
 * threads.  There is no CPU fallback: fpl_create() fails with FPL_ERR_NO_DEVICE when no
 * gfx950 device is usable.
 */
#ifndef FASTPLONG_AMD_H
#define FASTPLONG_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FPL_ABI_VERSION 7

/* limits */
#define FPL_MAX_ADAPTER_LEN 255 /* longest adapter the device path accepts            */
#define FPL_MAX_ADAPTERS 1024   /* start + end + FASTA adapters                      */
#define FPL_END_WINDOW 200      /* WINDOW,      reference src/adaptertrimmer.cpp:169,239 */
#define FPL_PATTERN_LEN 16      /* PATTERN_LEN, reference src/adaptertrimmer.cpp:170,240 */

/* filter result codes, reference src/common.h:43-50 */
#define FPL_PASS_FILTER 0
#define FPL_FAIL_N_BASE 12
#define FPL_FAIL_LENGTH 16
#define FPL_FAIL_TOO_LONG 17
#define FPL_FAIL_QUALITY 20
#define FPL_FAIL_COMPLEXITY 24
#define FPL_FILTER_RESULT_TYPES 32

/* error codes */
#define FPL_OK 0
#define FPL_ERR_ARG (-1)       /* null pointer / inconsistent argument                  */
#define FPL_ERR_NO_DEVICE (-2) /* no usable HIP device: this library has no CPU path    */
#define FPL_ERR_HIP (-3)       /* a HIP runtime call failed (see fpl_last_error)        */
#define FPL_ERR_ADAPTER (-4)   /* adapter longer than FPL_MAX_ADAPTER_LEN / too many    */
#define FPL_ERR_CAPACITY (-5)  /* read longer than the context's max_cycles capacity    */
#define FPL_ERR_STATE (-6)     /* call not valid in the current state                   */

/*
 * Options the path reads.  Field-for-field mirror of the parts of the reference's `Options`
 * that processSingleEnd and its callees consult (src/options.h:56-184); defaults in
 * fpl_options_default() are the reference CLI defaults (src/main.cpp:27-103).
 */
typedef struct fpl_options {
    /* TrimmingOptions, src/options.h:149-161: -f / -t */
    int32_t trim_front;
    int32_t trim_tail;
    /* QualityCutOptions, src/options.h:69-98: --cut_front/--cut_tail and their windows */
    int32_t cut_front;        /* enabledFront */
    int32_t cut_tail;         /* enabledTail  */
    int32_t cut_front_window; /* windowSizeFront 1..1000 */
    int32_t cut_front_quality; /* qualityFront, phred (not +33) */
    int32_t cut_tail_window;
    int32_t cut_tail_quality;
    /* PolyXTrimmerOptions, src/options.h:57-66: -x / --poly_x_min_len */
    int32_t polyx;
    int32_t polyx_min_len;
    /* AdapterOptions, src/options.h:126-147: -A, -d, --trimming_extension */
    int32_t adapter_enabled;
    double ed_max;            /* distance_threshold: round(ed_max*len) is done in double on the host */
    int32_t trimming_extension;
    /* QualityFilteringOptions, src/options.h:163-184: -Q -q -u -n --n_base_limit -m */
    int32_t qual_filter;      /* enabled */
    int32_t qualified_qual;   /* qualifiedQual as the raw ASCII char (phred+33), '0' = Q15 */
    int32_t unqualified_percent_limit;
    int32_t n_base_limit;     /* 1000000 means "no limit" (src/filter.cpp:46) */
    int32_t n_base_percent_limit;
    int32_t avg_qual_req;     /* -m, 0 = off */
    /* ReadLengthFilteringOptions, src/options.h:186-200: -L -l --length_limit */
    int32_t length_filter;    /* enabled */
    int32_t required_length;
    int32_t max_length;       /* 0 = no limit */
    /* LowComplexityFilterOptions, src/options.h:43-53: -y -Y (integer percent 0..100;
       the reference stores percent/100.0 and compares doubles -- equivalent) */
    int32_t complexity_filter;
    int32_t complexity_percent;
    /* LowQualityBreakOptions / MaskOptions, src/options.h:20-44: -b --break_window_size --break_mean_quality,
       -N --mask_window_size --mask_mean_quality (src/main.cpp:65-73,207-215); qualities are phred (not +33).
       With either enabled the per-read record cannot hold the outcome any more: see fpl_fragment. */
    int32_t break_enabled;
    int32_t break_window;
    int32_t break_quality;
    int32_t mask_enabled;
    int32_t mask_window;
    int32_t mask_quality;
} fpl_options;

typedef struct fpl_adapter {
    const char* seq; /* raw bytes, not NUL-terminated */
    int32_t len;
} fpl_adapter;

/*
 * One record per input read, written in input order.  Everything a writer needs to do what
 * src/seprocessor.cpp:265-281 does (serialize passing fragments; tag failed reads) without
 * touching the bases again.  All positions are byte offsets relative to the start of the
 * ORIGINAL read.  36 bytes.
 *
 *  dropped = 1 : Filter::trimAndCut returned NULL (src/filter.cpp:137-139,165,197,221): the
 *                read is counted in the pre-filter statistics and nowhere else.
 *  r1_start/r1_len : the read after trimAndCut + polyX + adapter end trims ("r1"); this is
 *                what --failed_out receives when exactly one fragment exists and it fails.
 *  n_frag 0..2 : fragments after the middle-adapter split (Read::breakByGap,
 *                src/read.cpp:192-215); kind 0 = unsplit r1, 1 = "split-by-adapter-left-",
 *                2 = "split-by-adapter-right-".
 *  code[i]     : Filter::passFilter result for fragment i (FPL_PASS_FILTER / FPL_FAIL_*).
 *  median_q_pre: per-read median quality char of the original read (Stats::statRead,
 *                src/stats.cpp:352-363); 0 when the read is empty.
 *  median_q_post[i]: same for fragment i, valid only when code[i] == FPL_PASS_FILTER.
 */
typedef struct fpl_read_result {
    uint32_t r1_start;
    uint32_t r1_len;
    uint32_t frag_start[2];
    uint32_t frag_len[2];
    uint8_t n_frag;
    uint8_t dropped;
    uint8_t code[2];
    uint8_t kind[2];
    uint8_t median_q_pre;
    uint8_t median_q_post[2];
    uint8_t reserved[3];
} fpl_read_result;

/*
 * --break / --mask (src/seprocessor.cpp:234-262) turn a read into any number of output reads, some of whose
 * bases are replaced by N.  When fpl_options.break_enabled or mask_enabled is set, fpl_read_result keeps
 * r1_start / r1_len / dropped / median_q_pre, n_frag saturates at 255 and its fragment fields are zero;
 * the fragments come as a list of these records (fpl_fragment_counts / fpl_get_fragments after each batch),
 * sorted by (read, seq_no) = the order the reference writes them.  32 bytes.
 *
 *  start/len    : window on the ORIGINAL read.
 *  kind         : 0 r1 itself or cut from r1, 1 / 2 the left / right part of a middle-adapter split.
 *  break_no     : 0 = not a product of Read::breakByRegions; i >= 1: that function named it by inserting
 *                 "r<i>-" after the first character of the (possibly split-prefixed) name (src/read.cpp:244,256).
 *  region_first/count : slice of the fpl_region list: stretches of the fragment that Read::maskRegionWithN
 *                 overwrote with 'N' (original-read coordinates, ascending, disjoint).
 *  code         : Filter::passFilter of the fragment (after masking); median_q valid when code == FPL_PASS_FILTER.
 * The reference writes --failed_out only for a read with exactly one fragment that fails; it then prints r1,
 * masked only if that fragment IS r1 (kind == 0 and break_no == 0: masking was done in place).
 */
typedef struct fpl_fragment {
    uint32_t read;
    uint32_t seq_no;
    uint32_t start;
    uint32_t len;
    uint32_t region_first;
    uint32_t region_count;
    uint16_t break_no;
    uint8_t code;
    uint8_t kind;
    uint8_t median_q;
    uint8_t reserved[3];
} fpl_fragment;

typedef struct fpl_region {
    uint32_t start;
    uint32_t len;
} fpl_region;

/*
 * Flat int64 counter buffer ("what the RCCL all-reduce sums").  C = max_cycles capacity.
 *
 *   [ pre-filter Stats | post-filter Stats | FilterResult | adapter-key histogram ]
 *
 * One Stats block (FPL_STATS_LEN(C) entries), replacing the members of the reference's
 * class Stats that statRead updates (src/stats.h:70-110):
 *   cyc[c][kind*8 + cls], c < C, cycle-major so that growing C appends:
 *        kind 0 mCycleBaseContents, 1 mCycleBaseQual, 2 mCycleQ20Bases, 3 mCycleQ30Bases,
 *        cls = base ASCII & 7 (A1 C3 T4 U5 N6 G7, src/stats.h:60-69);
 *        mCycleTotalBase / mCycleTotalQual are the sums over cls and are not stored.
 *   base_qual_hist[128]   mBaseQualHistogram
 *   median_hist[128]      mMedianReadQualHistogram
 *   median_bases[128]     mMedianReadQualBases
 *   kmer[1024]            mKmer (the reference allocates 2048, indices are 10 bit)
 *   reads, length_sum     mReads, mLengthSum
 *
 * FilterResult block (FPL_FR_LEN entries), src/filterresult.h:57-64:
 *   filter_read_stats[32], adapter_trimmed_reads, adapter_trimmed_bases,
 *   polyx_reads[4], polyx_bases[4]      (A,T,C,G order of ATCG_BASES, src/common.h:26)
 *
 * Adapter-key histogram: key_hist[a][side][cmplen], a < n_adapters, side 0 = trimmed at
 * read start (key = adapter.substr(alen-cmplen, cmplen)), side 1 = trimmed at read end
 * (key = adapter.substr(0, cmplen)), cmplen <= FPL_MAX_ADAPTER_LEN; cmplen == alen is the
 * full adapter.  The host turns indices into the strings of the reference's
 * map<string,long> mAdapter (src/filterresult.cpp:68-76).
 * Adapter slot order: 0 = start adapter, 1 = end adapter, 2.. = FASTA adapters in the order
 * given (the caller sorts them by FASTA header like src/options.cpp:50-59).
 */
#define FPL_CYC_STRIDE 32
#define FPL_STATS_TAIL (128 * 3 + 1024 + 2)
#define FPL_STATS_LEN(C) ((size_t)(C) * FPL_CYC_STRIDE + FPL_STATS_TAIL)
#define FPL_ST_CYC(c, kind, cls) ((size_t)(c) * FPL_CYC_STRIDE + (size_t)(kind) * 8 + (size_t)(cls))
#define FPL_ST_BASE_QUAL_HIST(C) ((size_t)(C) * FPL_CYC_STRIDE)
#define FPL_ST_MEDIAN_HIST(C) (FPL_ST_BASE_QUAL_HIST(C) + 128)
#define FPL_ST_MEDIAN_BASES(C) (FPL_ST_BASE_QUAL_HIST(C) + 256)
#define FPL_ST_KMER(C) (FPL_ST_BASE_QUAL_HIST(C) + 384)
#define FPL_ST_READS(C) (FPL_ST_BASE_QUAL_HIST(C) + 384 + 1024)
#define FPL_ST_LENGTH_SUM(C) (FPL_ST_READS(C) + 1)
#define FPL_FR_LEN 42
#define FPL_FR_FILTER 0
#define FPL_FR_ADAPTER_READS 32
#define FPL_FR_ADAPTER_BASES 33
#define FPL_FR_POLYX_READS 34
#define FPL_FR_POLYX_BASES 38
#define FPL_KEY_STRIDE (FPL_MAX_ADAPTER_LEN + 1)
#define FPL_KEYHIST_LEN(nad) ((size_t)(nad) * 2 * FPL_KEY_STRIDE)
#define FPL_COUNTERS_LEN(C, nad) (2 * FPL_STATS_LEN(C) + FPL_FR_LEN + FPL_KEYHIST_LEN(nad))
#define FPL_OFF_PRE(C) ((size_t)0)
#define FPL_OFF_POST(C) (FPL_STATS_LEN(C))
#define FPL_OFF_FR(C) (2 * FPL_STATS_LEN(C))
#define FPL_OFF_KEYHIST(C) (2 * FPL_STATS_LEN(C) + FPL_FR_LEN)

typedef struct fpl_ctx fpl_ctx;

/* Reference CLI defaults (src/main.cpp:27-103, src/options.h ctors). */
void fpl_options_default(fpl_options* opt);

/*
 * Create a context on HIP device `device` (>= 0).
 *  start/end  : adapter strings of --start_adapter / --end_adapter after the CLI resolved
 *               them (src/main.cpp:131-140); len 0 = empty string.
 *  fasta      : sequences of --adapter_fasta in the order trimByMultiSequences must visit
 *               them (src/adaptertrimmer.cpp:42-57); n_fasta = 0 means hasFasta = false.
 *  max_cycles : initial capacity C of the per-cycle tables (grown on demand).
 */
int fpl_create(fpl_ctx** out, const fpl_options* opt, const char* start_adapter, int32_t start_len,
               const char* end_adapter, int32_t end_len, const fpl_adapter* fasta, int32_t n_fasta,
               int32_t device, uint32_t max_cycles);
void fpl_destroy(fpl_ctx* ctx);

/*
 * Process one batch whose buffers already live in device memory (HBM).
 *  d_seq, d_qual : n_bytes bytes each; read i occupies [d_off[i], d_off[i+1]).
 *  d_off         : n_reads + 1 uint64 offsets, non-decreasing, d_off[n_reads] <= n_bytes.
 *  max_read_len  : upper bound of the read lengths in the batch (the caller knows it from
 *                  its offsets); used to size the launch and to check capacity.
 *  d_results     : n_reads records, device memory.
 *  stream        : hipStream_t (NULL = default stream).  Asynchronous: returns after the
 *                  launches are enqueued.
 */
int fpl_process_batch_device(fpl_ctx* ctx, const uint8_t* d_seq, const uint8_t* d_qual,
                             const uint64_t* d_off, uint32_t n_reads, uint64_t n_bytes,
                             uint32_t max_read_len, fpl_read_result* d_results, void* stream);

/*
 * Same, from host buffers (pinned recommended): copies in, processes, copies the records
 * back and synchronizes.  This is the call a host worker loop makes in place of
 * processSingleEnd().
 */
int fpl_process_batch(fpl_ctx* ctx, const uint8_t* seq, const uint8_t* qual, const uint64_t* off,
                      uint32_t n_reads, fpl_read_result* results);

/*
 * The same without the wait: the pipelined form a host thread uses to keep the PCIe link and the
 * kernels busy together (what the reference gets from its producer/consumer queues,
 * src/seprocessor.cpp:331-485).  fpl_process_batch_async() enqueues H2D copies on a copy stream,
 * the kernels on the context's compute stream behind them, the D2H of the records on a third
 * stream, and returns; fpl_wait() blocks until the OLDEST batch in flight is complete and its
 * records are in the `results` array given at submission.  Up to FPL_MAX_IN_FLIGHT batches may be
 * in flight per context (one set of device staging buffers per batch in flight): the copies of batch k+1 overlap
 * the kernels of batch k.  seq / qual / off must stay valid until the batch has been waited for;
 * they should come from fpl_host_alloc() (pinned) -- pageable memory works but the runtime then
 * stages the copy on the calling thread.  With break_enabled / mask_enabled, a submission first
 * drains the batch in flight (its fragment list lives in buffers the next batch reuses), and
 * fpl_get_fragments() refers to the batch most recently waited for.
 * Errors of the asynchronous part are reported by fpl_wait().
 */
#define FPL_MAX_IN_FLIGHT 3
int fpl_process_batch_async(fpl_ctx* ctx, const uint8_t* seq, const uint8_t* qual, const uint64_t* off,
                            uint32_t n_reads, fpl_read_result* results);
int fpl_wait(fpl_ctx* ctx);
int fpl_in_flight(const fpl_ctx* ctx);

/*
 * (ABI v7) The reader's work on the device: FASTQ TEXT in, records out.  Replaces, for the per-read path, what
 * FastqReader::read / ::getLine do per record on the reference's reader thread (src/fastqreader.cpp:219-347): the host
 * hands over a chunk of the file's bytes as they lie there -- it must start at the '@' of a record and end behind the line
 * break of a record's quality line, nothing else is asked of it -- and the device finds the line breaks, checks every record
 * the way FastqReader::read does ('@', '+', as many qualities as bases, :312-341), lays bases and qualities out as the
 * CSR batch the kernels take and runs the batch.  No base is copied on the host: the link carries the same 2.02 bytes per
 * base as with fpl_process_batch_async.
 *
 *   fpl_process_text_async  uploads `text` (page-locked memory recommended) and starts the parse; returns at once.  `text`
 *                           must stay valid until the batch has been waited for.  Shares the FPL_MAX_IN_FLIGHT slots and the
 *                           FIFO order of fpl_process_batch_async.
 *   fpl_peek_text           the parse's verdict for the NEXT PENDING text batch -- the oldest one in flight that has been neither
 *                           started nor cancelled -- as soon as it is in: the upload and the parse kernels, not the per-read
 *                           kernels, so a batch that has only been peeked at is in no counter yet.  Fills *out like fpl_wait_text.
 *   fpl_start_text          enqueues the per-read kernels of the next pending text batch (waits for its parse, not for them).
 *                           Optional: fpl_wait_text starts a batch that was not.  A host that starts batch k + 1 before it waits
 *                           for batch k keeps the device's queue filled while it sits in the wait.
 *   fpl_cancel_text         the next pending text batch will not run; its fpl_wait_text reports FPL_TEXT_CANCELLED.
 *                           Together: a host with several devices publishes every chunk's verdict and lets a chunk run only when
 *                           all chunks in front of it were good -- the reference stops READING at a malformed record
 *                           (src/fastqreader.cpp:326-341), so nothing behind one may be counted (bin/fastplong_amd does this).
 *   fpl_wait_text           blocks until the OLDEST batch in flight (which must be a text batch) is complete.  *out says what
 *                           the chunk held; results[i] is the record of read i and line_starts[4 i + j] the offset, in
 *                           `text`, of line j of record i (0 name, 1 bases, 2 '+', 3 qualities) -- so the caller formats its
 *                           output from the text it still holds.  Both arrays are the library's (page-locked) and stay valid
 *                           until the second submission after this one.
 *
 * status FPL_TEXT_IRREGULAR: the chunk is not "four lines per record, every line ended by \n or \r\n, '@' and '+' in place,
 * equal lengths" (blank lines, a lone \r, a missing final line break, a malformed record: bad_record is the first) -- NOTHING
 * of it was processed or counted; the caller parses the chunk with its own reader (the reference's rules for such text are
 * the sequential reader's: skipped lines, the error texts of :326-341) and submits it through fpl_process_batch_async.
 * FPL_TEXT_TOO_MANY: more than n_bytes / 64 + 16 records (reads shorter than 30 bases on average): same treatment.
 * A text batch is in the counters (fpl_get_counters, fpl_counters_device_ptr) once fpl_wait_text has returned for it: its per-read
 * kernels are enqueued by fpl_start_text or by that call (the uploads of the next chunks, submitted before, run beside them).  n_bytes < 4 GiB (line
 * positions are 32 bits).
 */
#define FPL_TEXT_OK 0
#define FPL_TEXT_IRREGULAR 1
#define FPL_TEXT_TOO_MANY 2
#define FPL_TEXT_CANCELLED 3
typedef struct fpl_text_result {
    uint32_t n_reads;      /* records of the chunk (0 unless status is FPL_TEXT_OK) */
    uint32_t status;       /* FPL_TEXT_* */
    uint64_t n_bases;      /* bases of all reads */
    uint64_t bad_record;   /* FPL_TEXT_IRREGULAR: index of the first record that failed a check, ~0 when the structure did */
    uint32_t max_read_len; /* longest read */
    uint32_t n_lines;      /* line breaks found */
} fpl_text_result;
int fpl_process_text_async(fpl_ctx* ctx, const uint8_t* text, uint64_t n_bytes);
int fpl_peek_text(fpl_ctx* ctx, fpl_text_result* out);
int fpl_start_text(fpl_ctx* ctx);
int fpl_cancel_text(fpl_ctx* ctx);
int fpl_wait_text(fpl_ctx* ctx, fpl_text_result* out, const fpl_read_result** results, const uint32_t** line_starts);

/* Page-locked host memory for the arrays handed to fpl_process_batch[_async] / fpl_process_text_async: the DMA engines read it
 * directly.  Blocks of 8 MB and more are anonymous memory on transparent huge pages, touched and registered with the runtime
 * (hipHostRegister, portable across devices) -- locking 4 KB pages goes at 4 GB/s, 370 huge pages take 13 ms for 740 MB; smaller
 * blocks, and any failure on that way, come from hipHostMalloc.  NULL when the allocation fails.  Thread-safe. */
void* fpl_host_alloc(size_t bytes);
void fpl_host_free(void* p);

/*
 * Fragment list of the LAST batch (contexts created with break_enabled or mask_enabled; otherwise the counts
 * are zero).  fpl_get_fragments synchronizes, copies the records to the host and sorts them by (read, seq_no).
 */
int fpl_fragment_counts(fpl_ctx* ctx, uint32_t* n_fragments, uint32_t* n_regions);
int fpl_get_fragments(fpl_ctx* ctx, fpl_fragment* fragments, uint32_t n_fragments, fpl_region* regions,
                      uint32_t n_regions);

/* Counter buffer: capacity, number of adapter slots, length, device pointer, host copy. */
uint32_t fpl_max_cycles(const fpl_ctx* ctx);
int32_t fpl_n_adapters(const fpl_ctx* ctx);
size_t fpl_counters_len(const fpl_ctx* ctx);
/* Grow the per-cycle capacity (all ranks must agree on C before an all-reduce). */
int fpl_reserve_cycles(fpl_ctx* ctx, uint32_t max_cycles);
/* int64 device buffer of fpl_counters_len() entries; valid until the next grow/destroy. A
   multi-GPU host all-reduces (sum) this buffer in place over RCCL. */
void* fpl_counters_device_ptr(fpl_ctx* ctx);
int fpl_get_counters(fpl_ctx* ctx, int64_t* host_buf, size_t n);
/*
 * The merge that follows the join in the reference (Stats::merge src/stats.cpp:1013-1082,
 * FilterResult::merge src/filterresult.cpp:28-61), for the contexts of ONE process (one per device):
 * agrees on the per-cycle capacity (max over the contexts, fpl_reserve_cycles), then sums the counter
 * buffers in place with one grouped RCCL all-reduce (int64, sum) over xGMI, so that afterwards every
 * context holds the totals.  n == 1 only reserves.  librccl is loaded on first use (dlopen), so hosts
 * that never call this do not need it.  Hosts that run one PROCESS per device all-reduce
 * fpl_counters_device_ptr() themselves (bench.py does, through torch.distributed) after agreeing on C.
 */
int fpl_allreduce_counters(fpl_ctx** ctxs, int32_t n);
/* (ABI v5) Path of the RCCL library fpl_allreduce_counters runs on: "" until a merge has needed it.  A librccl the process has
 * mapped already is taken first (PyTorch-ROCm's own), then the loader's search path, ROCm's directory and the directory of the
 * HIP runtime in use.  With FPL_RCCL_FORCE=1 in the environment a merge of ONE context goes through RCCL as well (a one-rank
 * communicator; the buffer comes out unchanged) -- the way to exercise the collective on a box with a single GPU. */
const char* fpl_rccl_library(void);
/* (ABI v5) Optional: make the RCCL communicators of these contexts' devices AHEAD of the merge and keep them -- ncclCommInitAll over
 * a node's eight devices takes far longer than the all-reduce of a few megabytes that follows, and a host can pay for it on a thread
 * of its own while its batches run (bin/fastplong_amd does).  fpl_allreduce_counters uses kept communicators when they were made
 * for exactly its devices (same order) and makes its own otherwise.  fpl_comm_init(NULL, 0) gives them back.  Thread-safe; may run
 * beside fpl_process_batch* on the same contexts. */
int fpl_comm_init(fpl_ctx** ctxs, int32_t n);

/* The counting loops of the adapter auto-detection, Evaluator::evalAdapterAndReadNum (src/evaluator.cpp:300-345), on the
 * device: for the reads of the evaluation prefix (host CSR arrays; the reference looks at <= 64 Ki reads / 512 Mbases) count
 * the 10-mers starting at the first 128 positions (side 0) or at the last 129 positions in front of the `shift_tail` skipped
 * bases (side 1).  counts / position_acc: host arrays of 4^10 entries, overwritten; *total = keys counted.  What the reference
 * does with the counters (getTopKey :268-326, extendKeyToAdapter :328-404) stays with the caller.  No context is needed: the
 * detection runs before the adapters -- and with them the contexts -- exist (src/main.cpp:270-277). */
int fpl_count_end_kmers(int32_t device, const uint8_t* seq, const uint64_t* off, uint32_t n_reads, int32_t side, int32_t shift_tail,
                        uint32_t* counts, uint64_t* position_acc, uint64_t* total);
/* ... and the whole decision on the device (fastplong_amd/csrc/adapter_pick.h): the counters stay in HBM, one more launch
 * picks the seed -- the admissible key with the largest count, Evaluator::getTopKey (src/evaluator.cpp:268-326) -- and grows it
 * into an adapter (Evaluator::extendKeyToAdapter, :328-404).  What comes back is what Evaluator::evalAdapterAndReadNum
 * (:191-222) needs for its verdict: the seed and its count, the number of keys seen, the keys counted, the grown sequence.
 * key = -1 (len 0) when no key qualifies.  is_rna: T is spelled U. */
typedef struct fpl_adapter_pick {
    int32_t key;
    uint32_t count;
    uint32_t total_key;
    int32_t len;
    uint64_t total;
    char seq[72]; /* NUL-terminated, <= 64 bases */
} fpl_adapter_pick;
int fpl_pick_adapter(int32_t device, const uint8_t* seq, const uint64_t* off, uint32_t n_reads, int32_t side, int32_t shift_tail,
                     int32_t is_rna, fpl_adapter_pick* out);
int fpl_reset_counters(fpl_ctx* ctx);
int fpl_synchronize(fpl_ctx* ctx);

/*
 * Which kernel forms the batches of a context took since fpl_create() / fpl_reset_counters() (ABI v6).  The library picks
 * per batch, by the batch's size: the end trims run in k_trim_ends_batched (64 reads per wave) from FPL_FORM_TRIM_BATCHED_MIN
 * reads on and one wave per read below, the statistics pass is k_stats_sorted (one table update per base) from
 * FPL_FORM_STATS_SORTED_MIN reads on and the two-update k_stats below (csrc/pipeline.h) -- a host that flushes small batches
 * never runs the forms a large resident batch is timed on, and this call says so.  The reference has no counterpart (its
 * unit of work is a pack of 16 reads, src/common.h:33).
 *   out[0] batches   out[1] reads   out[2] batches through k_trim_ends_batched   out[3] batches through k_stats_sorted
 *   out[4] reads of the largest batch   out[5] batches whose end trims started ahead of the main stream (fpl_assume_inputs_ready /
 *   the asynchronous path's own copy events), beside the scan of the batch before
 */
/*
 * A promise about fpl_process_batch_device (ABI v6): yes != 0 says that d_seq / d_qual / d_off of every later call are COMPLETE
 * on the device when the call is made -- resident data, or copies the caller has already waited for -- not merely ordered on
 * the call's stream.  The library then starts the end trims of a batch beside the kernels of the batch before it (they are
 * bound by memory latency, the scan and the statistics pass by instruction issue).  Without the promise only the library's own
 * asynchronous path (fpl_process_batch_async: it knows when its copies are in) does that.  Results are the same either way.
 * The promise holds until it is withdrawn (yes == 0): a caller that later hands in inputs which are only ORDERED on the call's
 * stream -- written by a kernel or a copy still in flight there -- must withdraw it first, or the trims may read them early.
 */
int fpl_assume_inputs_ready(fpl_ctx* ctx, int yes);

#define FPL_FORM_TRIM_BATCHED_MIN 65536
#define FPL_FORM_STATS_SORTED_MIN 150000
int fpl_get_batch_forms(const fpl_ctx* ctx, uint64_t out[6]);

/*
 * Per-kernel timing, measured with HIP events recorded on the stream the kernels are launched
 * on.  fpl_enable_timing(ctx, 1) starts a measurement window; every later
 * fpl_process_batch_device() records one event set (a ring of 128).  fpl_get_kernel_times()
 * waits for the recorded events and returns, per kernel, the time SUMMED over the batches of
 * the window (n_batches of them); names[i] are static strings.  The stages are those of csrc/pipeline.h (STAGE_NAMES); a stage
 * whose kernels run on one of the context's side streams -- the end trims of a batch whose inputs were known to be in (see
 * fpl_assume_inputs_ready), the post-only statistics pass of a large batch -- shows what is left of them IN LINE, not their
 * duration: withdraw the promise / set FPL_NO_OVERLAP=1 for in-line timings, or use rocprofv3's kernel table.
 */
#define FPL_MAX_KERNEL_TIMES 16
int fpl_enable_timing(fpl_ctx* ctx, int enable);
int fpl_get_kernel_times(fpl_ctx* ctx, float* ms, const char** names, int* n, int* n_batches);

const char* fpl_strerror(int code);
const char* fpl_last_error(const fpl_ctx* ctx);
int fpl_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FASTPLONG_AMD_H */
