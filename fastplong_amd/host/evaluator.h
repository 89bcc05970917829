/*
 * evaluator.h -- adapter auto-detection, the host side of the reference's Evaluator (src/evaluator.cpp:105-266; int2seq /
 * seq2int :485-560).  The seed choice and the growth of the adapter (getTopKey :268-326, extendKeyToAdapter :328-404) are
 * csrc/adapter_pick.h, shared with the device kernel that runs them for the CLI.
 * It runs once before the batches flow, on the first 64 Ki reads / 512 Mbases of the input, when
 * --start_adapter / --end_adapter are left at "auto" (src/main.cpp:270-277).
 *
 * Parity status: UNPINNED beyond the reference's own known-answer test (test/evaluator_test.cpp,
 * replayed in tests/test_host_evaluator.py): src/evaluator.cpp includes the FASTQ reader, whose ISA-L
 * header this image lacks, so the real object cannot be built in place as a cross-check.
 */
#ifndef FPLH_EVALUATOR_H
#define FPLH_EVALUATOR_H

#include <cstdint>
#include <functional>
#include <string>

namespace fplh {

/* The counting loops of the detection (src/evaluator.cpp:300-345) over the first / last 128 positions of every read of the
 * evaluation prefix: 4^10 counters and position sums, overwritten.  count_end_kmers_host is the host form; set_kmer_counter
 * plugs in another one (the CLI hands over the C-ABI's fpl_count_end_kmers, which counts on the GPU) -- it returns false when it
 * could not run, and the host form takes over. */
void count_end_kmers_host(const uint8_t* seq, const uint64_t* off, uint32_t n_reads, int side, int shift_tail, uint32_t* counts,
                          uint64_t* position_acc, uint64_t* total);
using KmerCounter = std::function<bool(const uint8_t*, const uint64_t*, uint32_t, int, int, uint32_t*, uint64_t*, uint64_t*)>;
void set_kmer_counter(KmerCounter f);
/* ... or the whole decision for one read end -- counting, seed, growth -- in one call (the CLI: the C-ABI's fpl_pick_adapter):
 * what Evaluator::evalAdapterAndReadNum (src/evaluator.cpp:191-222) bases its verdict on */
struct AdapterVerdict {
    int32_t key = -1;       /* the seed key, -1: none */
    uint32_t count = 0;     /* its count */
    uint32_t total_key = 0; /* keys seen at all */
    uint64_t total = 0;     /* keys counted */
    std::string adapter;    /* the seed grown in both directions */
};
using AdapterPicker = std::function<bool(const uint8_t*, const uint64_t*, uint32_t, int side, int shift_tail, bool is_rna, AdapterVerdict&)>;
void set_adapter_picker(AdapterPicker f);

std::string int2seq(unsigned int val, int seqlen, bool is_rna = false);
int seq2int(const char* seq, int rlen, int pos, int keylen, int last_val = -1);

/* Replaces `start` / `end` when they are "auto" and a sequence is detected; prints the reference's progress
 * lines to stderr.  trim_tail = -t (the evaluation skips max(1, trim_tail) bases at the end of every read). */
void detect_adapters(const std::string& path, int trim_tail, bool is_rna, std::string& start, std::string& end,
                     long* read_num = nullptr);

/* Evaluator::evaluateReadNum, src/evaluator.cpp:62-103: how many reads the input holds -- exact when the first
 * 512 Ki reads / 77 Mbases reach the end of the file, else file size x 1.01 / (bytes per read so far), where
 * "bytes so far" is what the reference's reader had PULLED from the file (FastqReader::getBytes,
 * src/fastqreader.cpp:190-200: 8 MiB buffers of a plain file, compressed bytes behind each 8 MiB inflate call of a
 * gzip file).  detect_adapters produces the same estimate from its own 64 Ki reads / 512 Mbases when asked.
 * Only --split reads the number (src/main.cpp:282-293). */
long evaluate_read_num(const std::string& path);

}  // namespace fplh

extern "C" {
/* test hooks */
int fplh_seq2int(const char* seq, int rlen, int pos, int keylen, int last_val);
void fplh_int2seq(unsigned int val, int seqlen, int is_rna, char* out);
/* out_start / out_end: buffers of >= 128 bytes, NUL-terminated results ("auto" when nothing was detected) */
void fplh_detect_adapters(const char* path, int trim_tail, int is_rna, char* out_start, char* out_end);
long fplh_evaluate_read_num(const char* path);
void fplh_count_end_kmers_host(const uint8_t* seq, const uint64_t* off, uint32_t n_reads, int side, int shift_tail, uint32_t* counts,
                               uint64_t* position_acc, uint64_t* total);
long fplh_detect_read_num(const char* path); /* the estimate detect_adapters gives */
int fplh_pick_adapter(const uint32_t* counts, const uint64_t* position_acc, int is_rna, uint32_t* count, uint32_t* total_key, char* out);
}
#endif
