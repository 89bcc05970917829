#include "evaluator.h"

#include <stdio.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <iostream>
#include <utility>
#include <vector>

#include "../csrc/adapter_pick.h"
#include "fastq.h"

using namespace std;

namespace fplh {

/* Evaluator::int2seq, src/evaluator.cpp:485-497 */
string int2seq(unsigned int val, int seqlen, bool is_rna) {
    char bases[4] = {'A', 'T', 'C', 'G'};
    if (is_rna) bases[1] = 'U';
    string ret(seqlen, 'N');
    for (int done = 0; done < seqlen; done++) {
        ret[seqlen - done - 1] = bases[val & 0x03];
        val >>= 2;
    }
    return ret;
}

static int base_code(char b) {
    switch (b) {
        case 'A': return 0;
        case 'T':
        case 'U': return 1;
        case 'C': return 2;
        case 'G': return 3;
        default: return -1; /* N or anything else */
    }
}

/* Evaluator::seq2int, src/evaluator.cpp:503-560: rolling when the previous key is valid */
int seq2int(const char* seq, int rlen, int pos, int keylen, int last_val) {
    (void)rlen;
    if (last_val >= 0) {
        const int mask = (1 << (keylen * 2)) - 1;
        const int c = base_code(seq[pos + keylen - 1]);
        if (c < 0) return -1;
        return ((last_val << 2) & mask) + c;
    }
    int key = 0;
    for (int i = pos; i < keylen + pos; i++) {
        const int c = base_code(seq[i]);
        if (c < 0) return -1;
        key = (key << 2) + c;
    }
    return key;
}

/* The seed and the adapter grown from it out of the counters (csrc/adapter_pick.h: the masked arg-max and the two walks the
   device runs in k_pick_adapter, here over host tables).  poly-A (key 0) reads as never seen, src/evaluator.cpp:191. */
fpl::pick::Pick pick_adapter_host(const uint32_t* counts, const uint64_t* position_acc, bool is_rna) {
    namespace pk = fpl::pick;
    pk::Pick p;
    p.key = -1;
    p.count = 0;
    p.total_key = 0;
    p.len = 0;
    p.seq[0] = 0;
    uint64_t best = 0;
    for (uint32_t k = 0; k < pk::NKEYS; k++) {
        const uint32_t val = counts[k];
        if (val == 0) continue;
        p.total_key++;
        if (!pk::key_admissible(k) || !pk::count_digits_vary(val)) continue;
        const uint64_t r = pk::seed_rank(val, k);
        if (r > best) best = r;
    }
    if (best) {
        p.key = (int32_t)~(uint32_t)best;
        p.count = (uint32_t)(best >> 32);
        pk::grow(p, is_rna, [&](uint32_t k) { return k ? counts[k] : 0u; }, [&](uint32_t k) { return position_acc[k]; });
    }
    return p;
}

namespace {

const uint64_t REF_BUF = 1ull << 23; /* FQ_BUF_SIZE, src/fastqreader.cpp:30 */

/* FastqReader::getBytes' bytesRead at the moment the reference's reader has handed out the record that ends in
 * front of uncompressed offset u (u >= 1): it refills only when a line runs off the end of its buffer. */
struct PulledBytes {
    bool gz = false;
    uint64_t size = 0;
    vector<pair<uint64_t, uint64_t>> steps; /* gzip: (uncompressed bytes out, compressed bytes in) after each inflate call */
    PulledBytes(const string& path, uint64_t upto) {
        gz = path.size() > 3 && path.compare(path.size() - 3, 3, ".gz") == 0;
        FILE* f = fopen(path.c_str(), "rb");
        if (!f) return;
        fseeko(f, 0, SEEK_END);
        size = (uint64_t)ftello(f);
        fseeko(f, 0, SEEK_SET);
        if (gz) { /* replay the inflate calls: 8 MiB of output each, a call also ends with its gzip member */
            z_stream zs;
            memset(&zs, 0, sizeof(zs));
            if (inflateInit2(&zs, 15 + 16) == Z_OK) {
                vector<unsigned char> in(1u << 20), out(REF_BUF);
                uint64_t tin = 0, tout = 0;
                bool more = true;
                while (more && tout < upto) {
                    zs.next_out = out.data();
                    zs.avail_out = (uInt)out.size();
                    int rc = Z_OK;
                    while (zs.avail_out > 0) {
                        if (zs.avail_in == 0) {
                            zs.next_in = in.data();
                            zs.avail_in = (uInt)fread(in.data(), 1, in.size(), f);
                            if (zs.avail_in == 0) {
                                more = false;
                                break;
                            }
                        }
                        const uInt before = zs.avail_in;
                        rc = inflate(&zs, Z_NO_FLUSH);
                        tin += before - zs.avail_in;
                        if (rc == Z_STREAM_END) {
                            inflateReset(&zs);
                            break;
                        }
                        if (rc != Z_OK) {
                            more = false;
                            break;
                        }
                    }
                    const uint64_t got = out.size() - zs.avail_out;
                    tout += got;
                    if (got > 0) steps.emplace_back(tout, tin);
                }
                inflateEnd(&zs);
            }
        }
        fclose(f);
    }
    uint64_t at(uint64_t u) const {
        if (!gz) return min(size, ((u - 1) / REF_BUF + 1) * REF_BUF);
        for (auto& st : steps)
            if (st.first >= u) return st.second;
        return steps.empty() ? 0 : steps.back().second;
    }
};

/* the tail of evaluateReadNum / evalAdapterAndReadNum (src/evaluator.cpp:93-102, :141-150) */
long read_num_from(const string& path, long records, bool reached_eof, uint64_t first_end, uint64_t last_end) {
    if (reached_eof) return records;
    if (records <= 0) return 0;
    const PulledBytes pb(path, last_end);
    const double bytesPerRead = (double)(pb.at(last_end) - pb.at(first_end)) / (double)records;
    const double est = (double)pb.size * 1.01 / bytesPerRead;
    /* everything inside one buffer: the reference converts +inf to long (LONG_MIN on x86-64) */
    if (!(est < 9.2e18)) return (long)0x8000000000000000ull;
    return (long)est;
}

/* the reference's loop `while(records < READ_LIMIT && bases < BASE_LIMIT) read()` with the position bookkeeping */
long read_prefix(FastqReader& reader, Batch& b, long read_limit, long base_limit, const string& path) {
    reader.fill(b, ~0ull, 1);
    const uint64_t first_end = reader.consumed();
    if (b.n() == 1) reader.fill(b, (uint64_t)base_limit, (uint32_t)read_limit);
    const long records = b.n();
    const bool reached_eof = records < read_limit && (long)b.seq.size() < base_limit;
    return read_num_from(path, records, reached_eof, first_end, reader.consumed());
}

}  // namespace

long evaluate_read_num(const string& path) {
    FastqReader reader(path);
    if (!reader.ok()) return 0;
    Batch b;
    return read_prefix(reader, b, 512 * 1024, 151L * 512 * 1024, path);
}

/* Evaluator::evalAdapterAndReadNum, src/evaluator.cpp:105-266 */
static KmerCounter g_kmer_counter;
void set_kmer_counter(KmerCounter f) { g_kmer_counter = std::move(f); }
static AdapterPicker g_adapter_picker;
void set_adapter_picker(AdapterPicker f) { g_adapter_picker = std::move(f); }

/* The end-k-mer counters of the evaluation prefix on the host, for when no device call is plugged in (tests, FPLH_HOST_KMERS):
   the windows and the key coder are the ones k_count_end_kmers uses (csrc/adapter_pick.h), one independent key per position. */
void count_end_kmers_host(const uint8_t* seq, const uint64_t* off, uint32_t n_reads, int side, int shift_tail, uint32_t* counts,
                          uint64_t* position_acc, uint64_t* total) {
    namespace pk = fpl::pick;
    fill(counts, counts + pk::NKEYS, 0u);
    fill(position_acc, position_acc + pk::NKEYS, (uint64_t)0);
    uint64_t seen = 0;
    for (uint32_t r = 0; r < n_reads; r++) {
        const long long rlen = (long long)(off[r + 1] - off[r]);
        long long first, last;
        if (!pk::key_window(rlen, side, shift_tail, first, last)) continue;
        const uint8_t* data = seq + off[r];
        for (long long pos = first; pos <= last; pos++) {
            uint32_t key;
            if (!pk::key_at(data + pos, key)) continue;
            counts[key]++;
            position_acc[key] += (uint64_t)(side == 0 ? pos : rlen - pos);
            seen++;
        }
    }
    *total = seen;
}

void detect_adapters(const string& path, int trim_tail, bool is_rna, string& start, string& end, long* read_num) {
    if (start != "auto" && end != "auto") return;
    const long READ_LIMIT = 64 * 1024;
    const long BASE_LIMIT = 8192 * READ_LIMIT;
    FastqReader reader(path);
    if (!reader.ok()) return;
    Batch b;
    const long rn = read_prefix(reader, b, READ_LIMIT, BASE_LIMIT, path);
    if (read_num) *read_num = rn;
    const long records = b.n();
    if (records < 100) return; /* we need at least 100 valid records to evaluate */
    const int shift_tail = max(1, trim_tail);
    const double FOLD_THRESHOLD = 100.0;
    for (int side = 0; side < 2; side++) {
        string& target = side == 0 ? start : end;
        if (target != "auto") continue;
        cerr << (side == 0 ? "Trying to detect adapter sequence at read start" : "Trying to detect adapter sequence at read end") << endl;
        /* counting, seed and growth: on the device when the caller has plugged it in (fpl_pick_adapter), else here.
           (the start adapter is spelled in DNA letters whatever the input, src/evaluator.cpp:205,246) */
        const bool rna = side == 0 ? false : is_rna;
        AdapterVerdict v;
        bool have = g_adapter_picker && g_adapter_picker(b.seq.data(), b.off.data(), (uint32_t)records, side, shift_tail, rna, v);
        if (!have) {
            vector<uint32_t> counts(fpl::pick::NKEYS);
            vector<uint64_t> position_acc(fpl::pick::NKEYS);
            uint64_t t = 0;
            if (!(g_kmer_counter && g_kmer_counter(b.seq.data(), b.off.data(), (uint32_t)records, side, shift_tail, counts.data(),
                                                   position_acc.data(), &t)))
                count_end_kmers_host(b.seq.data(), b.off.data(), (uint32_t)records, side, shift_tail, counts.data(),
                                     position_acc.data(), &t);
            const fpl::pick::Pick p = pick_adapter_host(counts.data(), position_acc.data(), rna);
            v.key = p.key;
            v.count = p.count;
            v.total_key = p.total_key;
            v.total = t;
            v.adapter.assign(p.seq, (size_t)p.len);
        }
        const long count = v.key >= 0 ? (long)v.count : 0; /* (the reference indexes counts[-1] when nothing qualifies) */
        if (v.key >= 0 && count > 10 && (double)(count * (long)v.total_key) > (double)(long)v.total * FOLD_THRESHOLD) {
            const string& adapter = v.adapter;
            if (adapter.length() > 16) {
                cerr << "Detected: " << adapter << endl;
                target = adapter;
            } else {
                cerr << "Found possible adapter sequence, but it's too short: " << adapter << ", specify "
                     << (side == 0 ? "-s " : "-e ") << adapter << " to force trimming using this adapter" << endl;
            }
        } else {
            cerr << "Not detected" << endl;
        }
    }
}

}  // namespace fplh

extern "C" {
int fplh_seq2int(const char* seq, int rlen, int pos, int keylen, int last_val) {
    return fplh::seq2int(seq, rlen, pos, keylen, last_val);
}
void fplh_int2seq(unsigned int val, int seqlen, int is_rna, char* out) {
    const std::string s = fplh::int2seq(val, seqlen, is_rna != 0);
    memcpy(out, s.c_str(), s.size() + 1);
}
long fplh_evaluate_read_num(const char* path) { return fplh::evaluate_read_num(path); }
void fplh_count_end_kmers_host(const uint8_t* seq, const uint64_t* off, uint32_t n_reads, int side, int shift_tail, uint32_t* counts,
                               uint64_t* position_acc, uint64_t* total) {
    fplh::count_end_kmers_host(seq, off, n_reads, side, shift_tail, counts, position_acc, total);
}
/* test hook: seed + grown adapter from host tables (out: >= 72 bytes); returns the seed key */
int fplh_pick_adapter(const uint32_t* counts, const uint64_t* position_acc, int is_rna, uint32_t* count, uint32_t* total_key, char* out) {
    const fpl::pick::Pick p = fplh::pick_adapter_host(counts, position_acc, is_rna != 0);
    if (count) *count = p.count;
    if (total_key) *total_key = p.total_key;
    memcpy(out, p.seq, (size_t)p.len + 1);
    return p.key;
}
long fplh_detect_read_num(const char* path) {
    std::string s = "auto", e = "auto";
    long n = 0;
    fplh::detect_adapters(path, 0, false, s, e, &n);
    return n;
}
void fplh_detect_adapters(const char* path, int trim_tail, int is_rna, char* out_start, char* out_end) {
    std::string s = "auto", e = "auto";
    fplh::detect_adapters(path, trim_tail, is_rna != 0, s, e);
    strncpy(out_start, s.c_str(), 127);
    out_start[127] = 0;
    strncpy(out_end, e.c_str(), 127);
    out_end[127] = 0;
}
}
