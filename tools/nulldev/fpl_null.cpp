/*
 * fpl_null.cpp -- MEASUREMENT INFRASTRUCTURE ONLY (tools/nulldev/libfastplong_amd.so, built by tools/nulldev/build.py).
 *
 * A NULL device behind the C-ABI: FPL_NULL_DEVICES (default 8) "devices" whose batches take no time at all --
 * fpl_process_batch_async returns at once, fpl_wait fills the records with "one passing fragment = the whole read" and counts
 * reads and bases, every other counter stays zero.  No kernel, no copy, no oracle.  What is left when bin/fastplong_amd runs
 * against it is the HOST side of a run over N devices -- chunk parsers, the batch round-robin over N device threads, the
 * gather lists / formatters, the writer(s), the merge, the reports -- i.e. the ceiling the host pipeline puts on an N-GPU
 * run, measured on a box with one GPU or none (bench.py's e2e.host_ceiling).  Differences from the real
 * library that make the ceiling optimistic: the CSR arenas are plain pages (no page-locking: fpl_host_alloc is malloc), there
 * is no PCIe traffic beside the parsers' memory traffic, and the records come back the moment they are asked for.
 * The product never links or loads this library: the CLI finds it only through LD_LIBRARY_PATH.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <deque>
#include <string>
#include <vector>

#include "../../include/fastplong_amd.h"
#include "../../tests/stub/text_stand_in.h" /* (the text entry points: a CPU line scan stands in for the device's -- its cost counts against the host here) */

struct Pending {
    const uint64_t* off;
    uint32_t n;
    fpl_read_result* res;
    const uint8_t* text = nullptr;
    uint64_t text_bytes = 0;
    bool is_text = false;
    bool started = false, cancelled = false;
};
struct fpl_ctx {
    int device = 0;
    int n_adapters = 2;
    uint32_t C = 1;
    std::vector<int64_t> counters;
    std::deque<Pending> q;
    uint64_t reads = 0, bases = 0;
    StandInText text_slot[FPL_MAX_IN_FLIGHT + 1];
    std::vector<fpl_read_result> text_res[FPL_MAX_IN_FLIGHT + 1];
    unsigned text_no = 0;
};

static int null_devices() {
    const char* e = getenv("FPL_NULL_DEVICES");
    return e && atoi(e) > 0 ? atoi(e) : 8;
}
static void relayout(fpl_ctx* c, uint32_t newC) { /* (nothing but the two totals is ever non-zero: written at read-out) */
    if (newC > c->C || c->counters.empty()) {
        c->C = newC > c->C ? newC : c->C;
        c->counters.assign(FPL_COUNTERS_LEN(c->C, c->n_adapters), 0);
    }
}
static void fill_totals(fpl_ctx* c) {
    std::fill(c->counters.begin(), c->counters.end(), 0);
    for (int k = 0; k < 2; k++) {
        int64_t* st = c->counters.data() + (k ? FPL_OFF_POST(c->C) : FPL_OFF_PRE(c->C));
        st[FPL_ST_READS(c->C)] = (int64_t)c->reads;
        st[FPL_ST_LENGTH_SUM(c->C)] = (int64_t)c->bases;
    }
    c->counters[FPL_OFF_FR(c->C) + FPL_FR_FILTER + FPL_PASS_FILTER] = (int64_t)c->reads;
}

extern "C" {

int fpl_abi_version(void) { return FPL_ABI_VERSION; }
const char* fpl_strerror(int code) { return code == FPL_OK ? "ok" : (code == FPL_ERR_NO_DEVICE ? "no such null device" : "null-device error"); }
const char* fpl_last_error(const fpl_ctx*) { return ""; }
void fpl_options_default(fpl_options* o) {
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
int fpl_create(fpl_ctx** out, const fpl_options* opt, const char*, int32_t, const char*, int32_t, const fpl_adapter*, int32_t n_fasta,
               int32_t device, uint32_t max_cycles) {
    if (!out || !opt) return FPL_ERR_ARG;
    *out = nullptr;
    if (device < 0 || device >= null_devices()) return FPL_ERR_NO_DEVICE;
    fpl_ctx* c = new fpl_ctx();
    c->device = device;
    c->n_adapters = 2 + n_fasta;
    relayout(c, max_cycles ? max_cycles : 1);
    *out = c;
    return FPL_OK;
}
void fpl_destroy(fpl_ctx* ctx) { delete ctx; }
int fpl_process_batch_async(fpl_ctx* ctx, const uint8_t*, const uint8_t*, const uint64_t* off, uint32_t n_reads, fpl_read_result* results) {
    if (!ctx) return FPL_ERR_ARG;
    if (ctx->q.size() >= FPL_MAX_IN_FLIGHT) return FPL_ERR_STATE;
    ctx->q.push_back(Pending{off, n_reads, results});
    return FPL_OK;
}
int fpl_in_flight(const fpl_ctx* ctx) { return ctx ? (int)ctx->q.size() : 0; }
int fpl_process_text_async(fpl_ctx* ctx, const uint8_t* text, uint64_t n_bytes) {
    if (!ctx || (n_bytes && !text)) return FPL_ERR_ARG;
    if (ctx->q.size() >= FPL_MAX_IN_FLIGHT) return FPL_ERR_STATE;
    Pending p{nullptr, 0, nullptr};
    p.text = text;
    p.text_bytes = n_bytes;
    p.is_text = true;
    ctx->q.push_back(p);
    return FPL_OK;
}
static Pending* text_pending(fpl_ctx* ctx) {
    for (auto& p : ctx->q)
        if (p.is_text && !p.started && !p.cancelled) return &p;
    return nullptr;
}
int fpl_peek_text(fpl_ctx* ctx, fpl_text_result* out) { /* (a null device's text is always taken: the verdict costs nothing here) */
    if (!ctx || !out) return FPL_ERR_ARG;
    if (!text_pending(ctx)) return FPL_ERR_STATE;
    memset(out, 0, sizeof *out);
    out->bad_record = ~0ull;
    return FPL_OK;
}
int fpl_start_text(fpl_ctx* ctx) {
    if (!ctx) return FPL_ERR_ARG;
    Pending* p = text_pending(ctx);
    if (!p) return FPL_ERR_STATE;
    p->started = true;
    return FPL_OK;
}
int fpl_cancel_text(fpl_ctx* ctx) {
    if (!ctx) return FPL_ERR_ARG;
    Pending* p = text_pending(ctx);
    if (!p) return FPL_ERR_STATE;
    p->cancelled = true;
    return FPL_OK;
}
int fpl_wait_text(fpl_ctx* ctx, fpl_text_result* out, const fpl_read_result** results, const uint32_t** line_starts) {
    if (!ctx || !out) return FPL_ERR_ARG;
    if (ctx->q.empty() || !ctx->q.front().is_text) return FPL_ERR_STATE;
    const Pending p = ctx->q.front();
    ctx->q.pop_front();
    if (p.cancelled) {
        memset(out, 0, sizeof *out);
        out->status = FPL_TEXT_CANCELLED;
        out->bad_record = ~0ull;
        if (results) *results = nullptr;
        if (line_starts) *line_starts = nullptr;
        return FPL_OK;
    }
    const unsigned k = ctx->text_no++ % (FPL_MAX_IN_FLIGHT + 1);
    StandInText& t = ctx->text_slot[k];
    stand_in_parse(p.text, p.text_bytes, false, t);
    *out = t.info;
    if (results) *results = nullptr;
    if (line_starts) *line_starts = nullptr;
    if (t.info.status != FPL_TEXT_OK || t.info.n_reads == 0) return FPL_OK;
    std::vector<fpl_read_result>& rr = ctx->text_res[k];
    rr.resize(t.info.n_reads);
    for (uint32_t i = 0; i < t.info.n_reads; i++) {
        const uint32_t l = (uint32_t)(t.off[i + 1] - t.off[i]);
        fpl_read_result r;
        memset(&r, 0, sizeof r);
        r.r1_len = l;
        r.frag_len[0] = l;
        r.n_frag = 1;
        r.code[0] = FPL_PASS_FILTER;
        r.median_q_pre = r.median_q_post[0] = 'I';
        rr[i] = r;
    }
    ctx->reads += t.info.n_reads;
    ctx->bases += t.info.n_bases;
    if (t.info.max_read_len > ctx->C) relayout(ctx, t.info.max_read_len + t.info.max_read_len / 4);
    if (results) *results = rr.data();
    if (line_starts) *line_starts = t.line.data();
    return FPL_OK;
}
int fpl_wait(fpl_ctx* ctx) {
    if (!ctx || ctx->q.empty() || ctx->q.front().is_text) return FPL_ERR_STATE;
    const Pending p = ctx->q.front();
    ctx->q.pop_front();
    uint32_t maxlen = 0;
    for (uint32_t i = 0; i < p.n; i++) {
        const uint32_t l = (uint32_t)(p.off[i + 1] - p.off[i]);
        fpl_read_result r;
        memset(&r, 0, sizeof r);
        r.r1_len = l;
        r.frag_len[0] = l;
        r.n_frag = 1;
        r.code[0] = FPL_PASS_FILTER;
        r.median_q_pre = r.median_q_post[0] = 'I';
        p.res[i] = r;
        if (l > maxlen) maxlen = l;
    }
    ctx->reads += p.n;
    ctx->bases += p.n ? p.off[p.n] - p.off[0] : 0;
    if (maxlen > ctx->C) relayout(ctx, maxlen + maxlen / 4);
    return FPL_OK;
}
int fpl_process_batch(fpl_ctx* ctx, const uint8_t* seq, const uint8_t* qual, const uint64_t* off, uint32_t n_reads, fpl_read_result* results) {
    const int rc = fpl_process_batch_async(ctx, seq, qual, off, n_reads, results);
    return rc == FPL_OK ? fpl_wait(ctx) : rc;
}
void* fpl_host_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
void fpl_host_free(void* p) { free(p); }
int fpl_fragment_counts(fpl_ctx* ctx, uint32_t* nf, uint32_t* nr) {
    if (!ctx || !nf || !nr) return FPL_ERR_ARG;
    *nf = *nr = 0;
    return FPL_OK;
}
int fpl_get_fragments(fpl_ctx* ctx, fpl_fragment*, uint32_t, fpl_region*, uint32_t) { return ctx ? FPL_OK : FPL_ERR_ARG; }
uint32_t fpl_max_cycles(const fpl_ctx* ctx) { return ctx ? ctx->C : 0; }
int32_t fpl_n_adapters(const fpl_ctx* ctx) { return ctx ? ctx->n_adapters : 0; }
size_t fpl_counters_len(const fpl_ctx* ctx) { return ctx ? ctx->counters.size() : 0; }
int fpl_reserve_cycles(fpl_ctx* ctx, uint32_t c) {
    if (!ctx) return FPL_ERR_ARG;
    relayout(ctx, c);
    return FPL_OK;
}
int fpl_get_counters(fpl_ctx* ctx, int64_t* buf, size_t n) {
    if (!ctx || !buf || n < ctx->counters.size()) return FPL_ERR_ARG;
    fill_totals(ctx);
    memcpy(buf, ctx->counters.data(), sizeof(int64_t) * ctx->counters.size());
    return FPL_OK;
}
int fpl_allreduce_counters(fpl_ctx** ctxs, int32_t n) { /* every context ends up with the sums, as after the real merge */
    if (!ctxs || n < 1) return FPL_ERR_ARG;
    uint32_t C = 0;
    uint64_t reads = 0, bases = 0;
    for (int i = 0; i < n; i++) {
        if (!ctxs[i] || !ctxs[i]->q.empty()) return FPL_ERR_STATE;
        C = ctxs[i]->C > C ? ctxs[i]->C : C;
        reads += ctxs[i]->reads;
        bases += ctxs[i]->bases;
    }
    for (int i = 0; i < n; i++) {
        relayout(ctxs[i], C);
        ctxs[i]->reads = reads;
        ctxs[i]->bases = bases;
    }
    return FPL_OK;
}
const char* fpl_rccl_library(void) { return ""; }
int fpl_comm_init(fpl_ctx**, int32_t) { return FPL_OK; }
int fpl_count_end_kmers(int32_t, const uint8_t*, const uint64_t*, uint32_t, int32_t, int32_t, uint32_t*, uint64_t*, uint64_t*) { return FPL_ERR_NO_DEVICE; }
int fpl_pick_adapter(int32_t, const uint8_t*, const uint64_t*, uint32_t, int32_t, int32_t, int32_t, fpl_adapter_pick*) { return FPL_ERR_NO_DEVICE; }
int fpl_reset_counters(fpl_ctx* ctx) {
    if (!ctx) return FPL_ERR_ARG;
    ctx->reads = ctx->bases = 0;
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
