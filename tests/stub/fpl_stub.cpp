/*
 * fpl_stub.cpp -- TEST INFRASTRUCTURE ONLY (tests/stub/libfastplong_amd.so, built by tests/stub/build.py).
 *
 * A stand-in for the C-ABI library on a box without GPUs, so that the HOST side of a multi-device run -- the CLI's
 * round-robin of batches over `--gpus N` device threads, two batches in flight per device, the formatter stage, the
 * in-order writer, the counter merge -- runs somewhere before hardware does.  FPL_STUB_DEVICES (default 1) "devices";
 * every context computes its batches with the oracle (oracle/fpl_oracle.c, the checker -- allowed here because this file is
 * test infrastructure under tests/ and never ships), lazily in fpl_wait() so that submissions really are outstanding, and
 * on a thread of its own per context like a device would.  FPL_STUB_LOG=<file>: one line per batch
 * "<device> <n_reads> <first 16 bytes of the batch's bases as hex>" in the order the batches were SUBMITTED.
 * The product never links or loads this library: the CLI finds it only through LD_LIBRARY_PATH in tests.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <deque>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fastplong_amd.h"
#include "../../oracle/fpl_oracle.h"
#include "text_stand_in.h"

struct Pending {
    const uint8_t *seq, *qual;
    const uint64_t* off;
    uint32_t n;
    fpl_read_result* res;
    const uint8_t* text = nullptr; /* a text batch (fpl_process_text_async): parsed in fpl_wait_text */
    uint64_t text_bytes = 0;
    bool is_text = false;
    bool started = false, cancelled = false; /* text: fpl_start_text / fpl_cancel_text */
};
struct fpl_ctx {
    int device = 0;
    fpl_options opt;
    std::string start, end;
    std::vector<std::string> fasta_s;
    std::vector<fpl_adapter> fasta;
    uint32_t C = 0;
    std::vector<int64_t> counters;
    std::deque<Pending> q;
    std::string err;
    orc_fraglist last = {nullptr, 0, 0, nullptr, 0, 0};
    StandInText text_slot[FPL_MAX_IN_FLIGHT + 1]; /* what fpl_wait_text hands out stays valid for two more submissions */
    std::vector<fpl_read_result> text_res[FPL_MAX_IN_FLIGHT + 1];
    unsigned text_no = 0;
    int n_adapters() const { return 2 + (int)fasta.size(); }
};

static std::mutex g_log_m;
/* FPL_STUB_COMM_LOG=<file>: one line per call of the two merge entry points, in the order the host made them */
static void comm_log(const char* what, fpl_ctx** ctxs, int32_t n) {
    const char* lf = getenv("FPL_STUB_COMM_LOG");
    if (!lf) return;
    std::lock_guard<std::mutex> g(g_log_m);
    if (FILE* f = fopen(lf, "a")) {
        fprintf(f, "%s %d", what, n);
        for (int i = 0; ctxs && i < n; i++) fprintf(f, " %d", ctxs[i] ? ctxs[i]->device : -1);
        fputc('\n', f);
        fclose(f);
    }
}
static int stub_devices() {
    const char* e = getenv("FPL_STUB_DEVICES");
    return e && atoi(e) > 0 ? atoi(e) : 1;
}

static void relayout(fpl_ctx* c, uint32_t newC) { /* the per-cycle tables sit in front of each Stats block: move the tails */
    if (newC <= c->C && !c->counters.empty()) return;
    const int nad = c->n_adapters();
    std::vector<int64_t> nb(FPL_COUNTERS_LEN(newC, nad), 0);
    if (!c->counters.empty()) {
        const uint32_t oldC = c->C;
        for (int k = 0; k < 2; k++) {
            const int64_t* so = c->counters.data() + (k ? FPL_OFF_POST(oldC) : FPL_OFF_PRE(oldC));
            int64_t* sn = nb.data() + (k ? FPL_OFF_POST(newC) : FPL_OFF_PRE(newC));
            memcpy(sn, so, sizeof(int64_t) * (size_t)oldC * FPL_CYC_STRIDE);
            memcpy(sn + (size_t)newC * FPL_CYC_STRIDE, so + (size_t)oldC * FPL_CYC_STRIDE, sizeof(int64_t) * FPL_STATS_TAIL);
        }
        memcpy(nb.data() + FPL_OFF_FR(newC), c->counters.data() + FPL_OFF_FR(oldC),
               sizeof(int64_t) * (FPL_FR_LEN + FPL_KEYHIST_LEN(nad)));
    }
    c->counters.swap(nb);
    c->C = newC;
}

extern "C" {

int fpl_abi_version(void) { return FPL_ABI_VERSION; }
const char* fpl_strerror(int code) {
    switch (code) {
        case FPL_OK: return "ok";
        case FPL_ERR_ARG: return "invalid argument";
        case FPL_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU path)";
        case FPL_ERR_CAPACITY: return "read longer than the per-cycle capacity";
        default: return "stub error";
    }
}
const char* fpl_last_error(const fpl_ctx* ctx) { return ctx ? ctx->err.c_str() : ""; }

void fpl_options_default(fpl_options* o) { /* (tests/test_cli_multi_device_stub.py compares this with the real library's) */
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->cut_front_window = o->cut_tail_window = 4;
    o->cut_front_quality = o->cut_tail_quality = 20;
    o->polyx_min_len = 10;
    o->adapter_enabled = 1;
    o->ed_max = 0.25;
    o->trimming_extension = 10;
    o->qual_filter = 1;
    o->qualified_qual = '0';
    o->unqualified_percent_limit = 40;
    o->n_base_limit = 1000000;
    o->n_base_percent_limit = 10;
    o->length_filter = 1;
    o->required_length = 20;
    o->complexity_percent = 30;
    o->break_window = 100;
    o->break_quality = 10;
    o->mask_window = 50;
    o->mask_quality = 10;
}

int fpl_create(fpl_ctx** out, const fpl_options* opt, const char* start_adapter, int32_t start_len, const char* end_adapter,
               int32_t end_len, const fpl_adapter* fasta, int32_t n_fasta, int32_t device, uint32_t max_cycles) {
    if (!out || !opt) return FPL_ERR_ARG;
    *out = nullptr;
    if (device < 0 || device >= stub_devices()) return FPL_ERR_NO_DEVICE;
    fpl_ctx* c = new fpl_ctx();
    c->device = device;
    c->opt = *opt;
    c->start.assign(start_adapter ? start_adapter : "", (size_t)start_len);
    c->end.assign(end_adapter ? end_adapter : "", (size_t)end_len);
    for (int i = 0; i < n_fasta; i++) c->fasta_s.emplace_back(fasta[i].seq, (size_t)fasta[i].len);
    for (auto& s : c->fasta_s) c->fasta.push_back(fpl_adapter{s.data(), (int32_t)s.size()});
    relayout(c, max_cycles ? max_cycles : 1);
    *out = c;
    return FPL_OK;
}
void fpl_destroy(fpl_ctx* ctx) {
    if (!ctx) return;
    orc_fraglist_free(&ctx->last);
    delete ctx;
}

int fpl_process_batch_async(fpl_ctx* ctx, const uint8_t* seq, const uint8_t* qual, const uint64_t* off, uint32_t n_reads,
                            fpl_read_result* results) {
    if (!ctx) return FPL_ERR_ARG;
    if (ctx->q.size() >= FPL_MAX_IN_FLIGHT) return FPL_ERR_STATE;
    if (const char* lf = getenv("FPL_STUB_LOG")) {
        std::lock_guard<std::mutex> g(g_log_m);
        if (FILE* f = fopen(lf, "a")) {
            fprintf(f, "%d %u ", ctx->device, n_reads);
            const uint64_t nb = n_reads ? off[n_reads] - off[0] : 0;
            for (uint64_t i = 0; i < 16 && i < nb; i++) fprintf(f, "%02x", seq[off[0] + i]);
            fputc('\n', f);
            fclose(f);
        }
    }
    ctx->q.push_back(Pending{seq, qual, off, n_reads, results});
    return FPL_OK;
}
int fpl_in_flight(const fpl_ctx* ctx) { return ctx ? (int)ctx->q.size() : 0; }

int fpl_wait(fpl_ctx* ctx) {
    if (!ctx || ctx->q.empty()) return FPL_ERR_STATE;
    if (ctx->q.front().is_text) return FPL_ERR_STATE;
    const Pending p = ctx->q.front();
    ctx->q.pop_front();
    uint32_t maxlen = 0;
    for (uint32_t i = 0; i < p.n; i++) {
        const uint64_t l = p.off[i + 1] - p.off[i];
        if (l > maxlen) maxlen = (uint32_t)l;
    }
    if (maxlen > ctx->C) relayout(ctx, maxlen + maxlen / 4);
    orc_config cfg;
    cfg.opt = ctx->opt;
    cfg.start_adapter = ctx->start.data();
    cfg.start_len = (int)ctx->start.size();
    cfg.end_adapter = ctx->end.data();
    cfg.end_len = (int)ctx->end.size();
    cfg.fasta = ctx->fasta.data();
    cfg.n_fasta = (int)ctx->fasta.size();
    orc_fraglist_free(&ctx->last);
    ctx->last = orc_fraglist{nullptr, 0, 0, nullptr, 0, 0};
    if (ctx->opt.break_enabled || ctx->opt.mask_enabled)
        orc_process_batch_ex(&cfg, p.seq, p.qual, p.off, p.n, ctx->counters.data(), ctx->C, p.res, &ctx->last);
    else
        orc_process_batch(&cfg, p.seq, p.qual, p.off, p.n, ctx->counters.data(), ctx->C, p.res);
    return FPL_OK;
}
int fpl_process_batch(fpl_ctx* ctx, const uint8_t* seq, const uint8_t* qual, const uint64_t* off, uint32_t n_reads,
                      fpl_read_result* results) {
    int rc = fpl_process_batch_async(ctx, seq, qual, off, n_reads, results);
    return rc == FPL_OK ? fpl_wait(ctx) : rc;
}

int fpl_process_text_async(fpl_ctx* ctx, const uint8_t* text, uint64_t n_bytes) {
    if (!ctx || (n_bytes && !text)) return FPL_ERR_ARG;
    if (ctx->q.size() >= FPL_MAX_IN_FLIGHT) return FPL_ERR_STATE;
    if (ctx->opt.break_enabled || ctx->opt.mask_enabled) return FPL_ERR_STATE;
    if (const char* lf = getenv("FPL_STUB_LOG")) {
        std::lock_guard<std::mutex> g(g_log_m);
        if (FILE* f = fopen(lf, "a")) {
            fprintf(f, "%d text %llu\n", ctx->device, (unsigned long long)n_bytes);
            fclose(f);
        }
    }
    Pending p{nullptr, nullptr, nullptr, 0, nullptr};
    p.text = text;
    p.text_bytes = n_bytes;
    p.is_text = true;
    ctx->q.push_back(p);
    return FPL_OK;
}
static Pending* text_pending(fpl_ctx* ctx) { /* the oldest text batch that is neither started nor cancelled */
    for (auto& p : ctx->q)
        if (p.is_text && !p.started && !p.cancelled) return &p;
    return nullptr;
}
int fpl_peek_text(fpl_ctx* ctx, fpl_text_result* out) {
    if (!ctx || !out) return FPL_ERR_ARG;
    Pending* p = text_pending(ctx);
    if (!p) return FPL_ERR_STATE;
    StandInText t; /* (parsed again by the wait: the stand-in keeps no state between the two) */
    stand_in_parse(p->text, p->text_bytes, false, t);
    *out = t.info;
    return FPL_OK;
}
int fpl_start_text(fpl_ctx* ctx) {
    if (!ctx) return FPL_ERR_ARG;
    Pending* p = text_pending(ctx);
    if (!p) return FPL_ERR_STATE;
    p->started = true; /* (the stand-in computes in the wait) */
    return FPL_OK;
}
int fpl_cancel_text(fpl_ctx* ctx) {
    if (!ctx) return FPL_ERR_ARG;
    Pending* p = text_pending(ctx);
    if (!p) return FPL_ERR_STATE;
    p->cancelled = true;
    if (const char* lf = getenv("FPL_STUB_LOG")) {
        std::lock_guard<std::mutex> g(g_log_m);
        if (FILE* f = fopen(lf, "a")) {
            fprintf(f, "%d cancel\n", ctx->device);
            fclose(f);
        }
    }
    return FPL_OK;
}
int fpl_wait_text(fpl_ctx* ctx, fpl_text_result* out, const fpl_read_result** results, const uint32_t** line_starts) {
    if (!ctx || !out || ctx->q.empty() || !ctx->q.front().is_text) return ctx && out ? FPL_ERR_STATE : FPL_ERR_ARG;
    const Pending p = ctx->q.front();
    ctx->q.pop_front();
    if (results) *results = nullptr;
    if (line_starts) *line_starts = nullptr;
    if (p.cancelled) {
        memset(out, 0, sizeof *out);
        out->status = FPL_TEXT_CANCELLED;
        out->bad_record = ~0ull;
        return FPL_OK;
    }
    const unsigned k = ctx->text_no++ % (FPL_MAX_IN_FLIGHT + 1);
    StandInText& t = ctx->text_slot[k];
    stand_in_parse(p.text, p.text_bytes, true, t);
    *out = t.info;
    if (results) *results = nullptr;
    if (line_starts) *line_starts = nullptr;
    if (t.info.status != FPL_TEXT_OK || t.info.n_reads == 0) return FPL_OK;
    ctx->text_res[k].resize(t.info.n_reads);
    /* the batch itself: through the CSR entry points above (same queue discipline: nothing else is in front of it now) */
    std::deque<Pending> rest;
    rest.swap(ctx->q);
    int rc = fpl_process_batch_async(ctx, t.seq.data(), t.qual.data(), t.off.data(), t.info.n_reads, ctx->text_res[k].data());
    if (rc == FPL_OK) rc = fpl_wait(ctx);
    ctx->q.swap(rest);
    if (rc != FPL_OK) return rc;
    if (results) *results = ctx->text_res[k].data();
    if (line_starts) *line_starts = t.line.data();
    return FPL_OK;
}

void* fpl_host_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
void fpl_host_free(void* p) { free(p); }

int fpl_fragment_counts(fpl_ctx* ctx, uint32_t* nf, uint32_t* nr) {
    if (!ctx || !nf || !nr) return FPL_ERR_ARG;
    *nf = ctx->last.n_frag;
    *nr = ctx->last.n_reg;
    return FPL_OK;
}
int fpl_get_fragments(fpl_ctx* ctx, fpl_fragment* fr, uint32_t nf, fpl_region* rg, uint32_t nr) {
    if (!ctx || nf < ctx->last.n_frag || nr < ctx->last.n_reg) return FPL_ERR_ARG;
    if (ctx->last.n_frag) memcpy(fr, ctx->last.frag, sizeof(fpl_fragment) * ctx->last.n_frag);
    if (ctx->last.n_reg) memcpy(rg, ctx->last.reg, sizeof(fpl_region) * ctx->last.n_reg);
    return FPL_OK;
}

uint32_t fpl_max_cycles(const fpl_ctx* ctx) { return ctx ? ctx->C : 0; }
int32_t fpl_n_adapters(const fpl_ctx* ctx) { return ctx ? ctx->n_adapters() : 0; }
size_t fpl_counters_len(const fpl_ctx* ctx) { return ctx ? ctx->counters.size() : 0; }
int fpl_reserve_cycles(fpl_ctx* ctx, uint32_t c) {
    if (!ctx) return FPL_ERR_ARG;
    relayout(ctx, c);
    return FPL_OK;
}
int fpl_get_counters(fpl_ctx* ctx, int64_t* buf, size_t n) {
    if (!ctx || !buf || n < ctx->counters.size()) return FPL_ERR_ARG;
    memcpy(buf, ctx->counters.data(), sizeof(int64_t) * ctx->counters.size());
    return FPL_OK;
}
/* Stats::merge / FilterResult::merge: agree on the capacity, then every context holds the sums */
int fpl_allreduce_counters(fpl_ctx** ctxs, int32_t n) {
    if (!ctxs || n < 1) return FPL_ERR_ARG;
    comm_log("allreduce", ctxs, n);
    uint32_t C = 0;
    for (int i = 0; i < n; i++) {
        if (!ctxs[i] || !ctxs[i]->q.empty()) return FPL_ERR_STATE;
        if (ctxs[i]->C > C) C = ctxs[i]->C;
    }
    for (int i = 0; i < n; i++) relayout(ctxs[i], C);
    std::vector<int64_t> sum(ctxs[0]->counters.size(), 0);
    for (int i = 0; i < n; i++)
        for (size_t k = 0; k < sum.size(); k++) sum[k] += ctxs[i]->counters[k];
    for (int i = 0; i < n; i++) ctxs[i]->counters = sum;
    return FPL_OK;
}
const char* fpl_rccl_library(void) { return ""; }
int fpl_comm_init(fpl_ctx** ctxs, int32_t n) {
    comm_log("comm_init", ctxs, n);
    return FPL_OK;
}
int fpl_count_end_kmers(int32_t, const uint8_t*, const uint64_t*, uint32_t, int32_t, int32_t, uint32_t*, uint64_t*, uint64_t*) {
    return FPL_ERR_NO_DEVICE; /* (the tests give -s / -e, or set FPLH_HOST_KMERS) */
}
int fpl_pick_adapter(int32_t, const uint8_t*, const uint64_t*, uint32_t, int32_t, int32_t, int32_t, fpl_adapter_pick*) {
    return FPL_ERR_NO_DEVICE;
}
int fpl_reset_counters(fpl_ctx* ctx) {
    if (!ctx) return FPL_ERR_ARG;
    std::fill(ctx->counters.begin(), ctx->counters.end(), 0);
    return FPL_OK;
}
int fpl_synchronize(fpl_ctx*) { return FPL_OK; }
int fpl_assume_inputs_ready(fpl_ctx* ctx, int) { return ctx ? FPL_OK : FPL_ERR_ARG; }
int fpl_get_batch_forms(const fpl_ctx* ctx, uint64_t out[6]) { /* (no kernels here: nothing to report but zeros) */
    if (!ctx || !out) return FPL_ERR_ARG;
    for (int i = 0; i < 6; i++) out[i] = 0;
    return FPL_OK;
}
}
