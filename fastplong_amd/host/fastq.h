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

#include <string>
#include <vector>

#include "fastplong_amd.h"

namespace fplh {

struct Batch {
    std::vector<uint8_t> seq, qual;   /* CSR payload handed to fpl_process_batch */
    std::vector<uint64_t> off;        /* n + 1 */
    std::vector<char> text;           /* name and strand lines, back to back */
    std::vector<uint64_t> name_off;   /* n + 1 offsets into text for names   */
    std::vector<uint32_t> name_len, strand_len; /* strand line follows the name in `text` */
    uint32_t n() const { return off.empty() ? 0 : (uint32_t)(off.size() - 1); }
    void clear();
};

/* Plain or gzip FASTQ (zlib).  Line splitting follows FastqReader::getLine: a line ends at
 * '\r' or '\n', "\r\n" counts once; records whose header does not start with '@' are skipped
 * line by line; a malformed record ends the input (src/fastqreader.cpp:326-341). */
class FastqReader {
   public:
    explicit FastqReader(const std::string& path);
    ~FastqReader();
    bool ok() const { return fp_ != nullptr; }
    /* append records until the batch holds >= max_bases bases or max_reads reads; returns the
     * number of records appended (0 at end of input) */
    uint32_t fill(Batch& b, uint64_t max_bases, uint32_t max_reads);
    bool malformed() const { return malformed_; }

   private:
    bool getline(std::string& line);
    bool refill();
    void* fp_ = nullptr; /* gzFile */
    std::vector<char> buf_;
    size_t pos_ = 0, len_ = 0;
    bool eof_ = false, malformed_ = false;
};

/* Serialize what src/seprocessor.cpp:265-281 writes for one batch: passing fragments to `out`
 * (name with the split prefix when the read was broken), and -- when failed != nullptr -- the
 * trimmed read with its tag for reads that produced exactly one fragment and failed. */
void format_batch(const Batch& b, const fpl_read_result* res, std::string& out, std::string* failed);

}  // namespace fplh

extern "C" {
/* test hooks: parse a FASTQ file into CSR arrays; format a batch from result records */
void* fplh_batch_read(const char* path, uint64_t max_bases, uint32_t max_reads);
uint32_t fplh_batch_n(void* b);
uint64_t fplh_batch_bytes(void* b);
const uint8_t* fplh_batch_seq(void* b);
const uint8_t* fplh_batch_qual(void* b);
const uint64_t* fplh_batch_off(void* b);
void fplh_batch_free(void* b);
/* returns malloc'ed buffers the caller frees with fplh_free */
int fplh_format_batch(void* b, const fpl_read_result* res, char** out, uint64_t* out_len, char** failed,
                      uint64_t* failed_len);
void fplh_free(void* p);
}
#endif
