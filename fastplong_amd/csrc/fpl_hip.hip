/*
 * fpl_hip.hip -- the C-ABI of include/fastplong_amd.h on top of the gfx950 kernels.
 * Built by hipcc only (--offload-arch=gfx950); there is no CPU path in this library.
 */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> /* types only: the library is loaded with dlopen in fpl_allreduce_counters */
#include <dlfcn.h>
#include <sys/mman.h>

#include <new>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <thread>
#include <mutex>
#include <string>
#include <vector>

#include "pipeline.h"
#include "text_parse.h"

using namespace fpl;

struct fpl_ctx {
    int device = -1;
    u32 n_cu = 256;
    int dbg = 0;
    bool probe_primed = false;
    int n_adapters = 2;
    u32 C = 0;
    DevConfig* d_cfg = nullptr;
    DevAdapter* d_ads = nullptr;
    long long* d_counters = nullptr;
    /* per-batch workspace, grown on demand */
    u32 ws_reads = 0;
    ReadState* d_state = nullptr;
    ScanRec* d_recs = nullptr;  /* k_scan -> k_resolve */
    ScanWin* d_wins = nullptr;
    RedoItem* d_redo = nullptr; /* k_resolve -> k_redo */
    uint64_t* d_frag_off = nullptr;
    u32* d_frag_len = nullptr;
    u32* d_work_ctr = nullptr;
    /* --break / --mask (DevConfig::defer): lists k_break_mask appends to, sized per batch */
    DevConfig hcfg;
    u32* d_frag_cyc = nullptr;
    BmLists bm = {nullptr, nullptr, 0, 0, 0, nullptr};
    size_t scratch_slabs = 0;
    u32* d_sort_ws = nullptr;       /* k_stats_sorted: bucket counters and the slice table */
    size_t sort_ws_cap = 0;         /* words */
    uint64_t* d_st_off = nullptr;   /* the reads in sorted order (ws_reads each) */
    u32* d_st_len = nullptr;
    u32* d_st_e = nullptr;
    u64* d_stats_scratch = nullptr;
    u8* d_stats_flags = nullptr;
    u64* d_extra_scratch = nullptr; /* the post-only pass's own slabs / flags (it runs on s_aux beside the reduce of k_stats_sorted) */
    u8* d_extra_flags = nullptr;
    size_t extra_slabs = 0;
    /* The end trims of batch k + 1 beside the kernels of batch k ("trim ahead"): the trim kernel is the first of a batch, needs
       nothing of the batch before, and is bound by memory latency where k_scan / k_stats_sorted are bound by instruction issue
       -- 0.5 ms of a 12.5 ms step when two whole batches run side by side (round 4, tools/overlap_probe.py).  It writes
       ReadState[] and takes its groups off a work counter: both exist twice, batches alternate.  A batch qualifies when its
       inputs are known to be complete on the device before its predecessor is done: the asynchronous path (its own H2D
       event), or a caller's promise (fpl_assume_inputs_ready). */
    ReadState* d_state2 = nullptr;
    hipStream_t s_trim = nullptr;
    hipEvent_t ev_trim_done = nullptr, ev_batch_done[2] = {nullptr, nullptr}, ev_stats_done[2] = {nullptr, nullptr};
    int ahead_gate = 0;             /* FPL_TRIM_AHEAD_GATE: 0 the trims of batch k + 1 start as soon as batch k - 1 is done -- beside k_scan of
                                       batch k, two of their blocks per CU (pipeline.h) --, 1 when the statistics kernel
                                       of batch k is done (beside its reduce / post-only tail: the default until round 6) */
    uint64_t batch_no = 0;          /* batches enqueued (parity picks the buffers) */
    bool trim_ahead = true;         /* FPL_NO_TRIM_AHEAD=1 (read in fpl_create) turns it off */
    bool inputs_ready = false;      /* fpl_assume_inputs_ready */
    hipEvent_t next_inputs_event = nullptr; /* (set by the asynchronous path around its call of fpl_process_batch_device) */
    hipStream_t s_aux = nullptr;    /* owned: the side stream of a batch (pipeline.h: FPL_FORK / FPL_JOIN) */
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool overlap = true;            /* FPL_NO_OVERLAP=1 (read in fpl_create): everything on the one stream */
    /* staging for the host-pointer entry points: FPL_MAX_IN_FLIGHT slots, so that the copies of one batch
       overlap the kernels of the previous one */
    struct Slot {
        uint64_t st_bytes = 0;
        u32 st_reads = 0;
        u32 h_res_cap = 0; /* records h_results holds (a text slot sizes it by the records its chunk really has) */
        u8* d_seq = nullptr;
        u8* d_qual = nullptr;
        uint64_t* d_off = nullptr;
        fpl_read_result* d_results = nullptr;
        fpl_read_result* h_results = nullptr; /* pinned: the D2H copy never waits for a pageable destination */
        u32 h_reads = 0;
        hipEvent_t ev_h2d = nullptr, ev_kern = nullptr, ev_done = nullptr;
        fpl_read_result* user_results = nullptr;
        u32 n_reads = 0;
        int rc = FPL_OK; /* error met while enqueueing, reported by fpl_wait */
        /* a TEXT batch (fpl_process_text_async): the chunk's bytes, its line breaks, the records' line starts and lengths; stage 1
           (copy + parse + the header's way back) is enqueued at submission, stage 2 (the per-read kernels, the records' and line
           starts' way back) once the header is in -- by the next submission or by the wait, whichever comes first */
        int kind = 0;          /* 0 CSR batch, 1 text batch */
        bool cancelled = false; /* text: fpl_cancel_text -- never run, reported by its wait */
        int stage = 0;         /* text: 1 parse enqueued, 2 batch enqueued (or nothing to enqueue) */
        uint64_t text_cap = 0; /* bytes d_text holds */
        u32 rec_cap = 0;       /* records d_line / d_len / h_line hold */
        u8* d_text = nullptr;
        u32* d_nl = nullptr;
        u32* d_blk = nullptr;
        u32* d_line = nullptr;
        u32* d_len = nullptr;
        TextHeader* d_hdr = nullptr;
        TextHeader* h_hdr = nullptr; /* pinned */
        u32* h_line = nullptr;       /* pinned */
        u32 h_line_cap = 0;
        uint64_t text_bytes = 0;
        hipEvent_t ev_parsed = nullptr;
    };
    Slot slot[FPL_MAX_IN_FLIGHT];
    u32 submitted = 0, waited = 0; /* batches handed to / collected from the asynchronous path */
    hipStream_t stream = nullptr;  /* owned: the compute stream of the host-pointer entry points */
    hipStream_t s_h2d = nullptr, s_d2h = nullptr; /* owned: copy streams */
    hipStream_t s_parse = nullptr; /* owned: the text-parse kernels of a chunk (behind its upload, beside the upload of the next) */
    StatsTune tune; /* FPL_STATS_* tuning hooks, read once in fpl_create */
    /* timing */
    int timing = 0;
    static constexpr int EV_RING = 128;
    hipEvent_t ev[EV_RING][N_STAGES + 1] = {};
    int ev_calls = 0; /* batches recorded since fpl_enable_timing() */
    bool ev_ready = false; /* the whole event ring exists */
    uint64_t forms[6] = {0, 0, 0, 0, 0, 0}; /* fpl_get_batch_forms */
    std::string err;
};

#define FPL_HIP(call)                                                                         \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess) {                                                              \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e__);                    \
            return FPL_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)

extern "C" {

int fpl_abi_version(void) { return FPL_ABI_VERSION; }

const char* fpl_strerror(int code) {
    switch (code) {
        case FPL_OK: return "ok";
        case FPL_ERR_ARG: return "invalid argument";
        case FPL_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU path)";
        case FPL_ERR_HIP: return "HIP runtime error";
        case FPL_ERR_ADAPTER: return "adapter too long or too many adapters";
        case FPL_ERR_CAPACITY: return "read longer than the per-cycle capacity";
        case FPL_ERR_STATE: return "invalid state";
        default: return "unknown error";
    }
}

const char* fpl_last_error(const fpl_ctx* ctx) { return ctx ? ctx->err.c_str() : ""; }

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
    o->break_window = 100; /* src/main.cpp:72-73 */
    o->break_quality = 10;
    o->mask_window = 50;   /* src/main.cpp:67-68 */
    o->mask_quality = 10;
}

static int alloc_counters(fpl_ctx* ctx, u32 C, long long** out) {
    size_t n = FPL_COUNTERS_LEN(C, ctx->n_adapters);
    FPL_HIP(hipMalloc((void**)out, n * sizeof(long long)));
    FPL_HIP(hipMemset(*out, 0, n * sizeof(long long)));
    return FPL_OK;
}

int fpl_create(fpl_ctx** out, const fpl_options* opt, const char* start_adapter, int32_t start_len,
               const char* end_adapter, int32_t end_len, const fpl_adapter* fasta, int32_t n_fasta,
               int32_t device, uint32_t max_cycles) {
    if (!out || !opt || start_len < 0 || end_len < 0 || n_fasta < 0 || (start_len && !start_adapter) ||
        (end_len && !end_adapter) || (n_fasta && !fasta))
        return FPL_ERR_ARG;
    *out = nullptr;
    if (start_len > FPL_MAX_ADAPTER_LEN || end_len > FPL_MAX_ADAPTER_LEN || 2 + n_fasta > FPL_MAX_ADAPTERS)
        return FPL_ERR_ADAPTER;
    for (int i = 0; i < n_fasta; i++)
        if (fasta[i].len < 0 || fasta[i].len > FPL_MAX_ADAPTER_LEN || (fasta[i].len && !fasta[i].seq)) return FPL_ERR_ADAPTER;
    int ndev = 0;
    /* (a context of the host-pointer path drives five streams -- kernels, two copy streams, two side streams; with the runtime's
       default of four hardware queues per device two of them share one and run in submission order.  The HOST asks for more --
       GPU_MAX_HW_QUEUES=8 in the environment before its first HIP call, as bin/fastplong_amd and bench.py do;
       the library does not touch the process's environment: setenv races with getenv on the host's other threads and does
       nothing once the runtime is up) */
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return FPL_ERR_NO_DEVICE;
    fpl_ctx* ctx = new (std::nothrow) fpl_ctx();
    if (!ctx) return FPL_ERR_ARG;
    ctx->device = device;
    ctx->n_adapters = 2 + n_fasta;
    int rc = [&]() -> int {
        FPL_HIP(hipSetDevice(device));
        hipDeviceProp_t prop;
        FPL_HIP(hipGetDeviceProperties(&prop, device));
        ctx->n_cu = prop.multiProcessorCount > 0 ? (u32)prop.multiProcessorCount : 256;
        /* (the side streams first, the three streams of the host-pointer path when that path is first used -- ensure_host_streams:
           the runtime deals its hardware queues out to the streams in turn, four by default, and two streams that share a queue
           run in submission order, i.e. not beside each other.  A process that only hands over device pointers has the caller's
           stream, s_aux and s_trim: three queues.) */
        FPL_HIP(hipStreamCreateWithFlags(&ctx->s_aux, hipStreamNonBlocking));
        FPL_HIP(hipStreamCreateWithFlags(&ctx->s_trim, hipStreamNonBlocking));
        FPL_HIP(hipEventCreateWithFlags(&ctx->ev_trim_done, hipEventDisableTiming));
        FPL_HIP(hipEventCreateWithFlags(&ctx->ev_batch_done[0], hipEventDisableTiming));
        FPL_HIP(hipEventCreateWithFlags(&ctx->ev_batch_done[1], hipEventDisableTiming));
        FPL_HIP(hipEventCreateWithFlags(&ctx->ev_stats_done[0], hipEventDisableTiming));
        FPL_HIP(hipEventCreateWithFlags(&ctx->ev_stats_done[1], hipEventDisableTiming));
        if (const char* e = getenv("FPL_TRIM_AHEAD_GATE")) ctx->ahead_gate = atoi(e);
        if (const char* e = getenv("FPL_NO_TRIM_AHEAD")) ctx->trim_ahead = atoi(e) == 0;
        FPL_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        FPL_HIP(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
        if (const char* e = getenv("FPL_NO_OVERLAP")) ctx->overlap = atoi(e) == 0;
        for (auto& sl : ctx->slot) {
            FPL_HIP(hipEventCreateWithFlags(&sl.ev_h2d, hipEventDisableTiming));
            FPL_HIP(hipEventCreateWithFlags(&sl.ev_kern, hipEventDisableTiming));
            FPL_HIP(hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming));
            FPL_HIP(hipEventCreateWithFlags(&sl.ev_parsed, hipEventDisableTiming));
        }
        DevConfig cfg;
        build_config(&cfg, opt, start_len, end_len, n_fasta);
        std::vector<DevAdapter> ads(ctx->n_adapters);
        build_adapter(&ads[0], start_adapter, start_len);
        build_adapter(&ads[1], end_adapter, end_len);
        for (int i = 0; i < n_fasta; i++) build_adapter(&ads[2 + i], fasta[i].seq, fasta[i].len);
        cfg.ham_fast = ads[0].acgt_only && ads[1].acgt_only;
        {
            std::vector<int> lens(2 + n_fasta), acgt(2 + n_fasta);
            for (int i = 0; i < 2 + n_fasta; i++) lens[i] = ads[i].len, acgt[i] = ads[i].acgt_only;
            cfg.trim_mode = trim_mode_of(lens.data(), acgt.data(), 2 + n_fasta);
        }
        cfg.scan_short = cfg.adapter_enabled && cfg.ham_fast && ads[0].len <= 32 && ads[1].len <= 32;
        if (const char* e = getenv("FPL_DEBUG_FLAGS")) cfg.dbg = atoi(e); /* the environment is read here and nowhere else */
        ctx->tune = stats_tune_from_env();
        ctx->dbg = cfg.dbg;
        if ((cfg.brk && cfg.brk_w <= 0) || (cfg.msk && cfg.msk_w <= 0)) {
            ctx->err = "break / mask window size must be positive";
            return FPL_ERR_ARG; /* (the caller below destroys the context) */
        }
        ctx->hcfg = cfg;
        FPL_HIP(hipMalloc((void**)&ctx->d_cfg, sizeof(DevConfig)));
        FPL_HIP(hipMemcpy(ctx->d_cfg, &cfg, sizeof(cfg), hipMemcpyHostToDevice));
        FPL_HIP(hipMalloc((void**)&ctx->d_ads, sizeof(DevAdapter) * ads.size()));
        FPL_HIP(hipMemcpy(ctx->d_ads, ads.data(), sizeof(DevAdapter) * ads.size(), hipMemcpyHostToDevice));
        FPL_HIP(hipMalloc((void**)&ctx->d_work_ctr, 2 * WORK_CTR_WORDS * sizeof(u32))); /* (two sets: batches alternate) */
        ctx->C = max_cycles ? max_cycles : 1;
        int r = alloc_counters(ctx, ctx->C, &ctx->d_counters);
        if (r != FPL_OK) return r;
        /* (the ring of timing events -- a thousand of them -- is made when timing is first asked for: a command-line run never does) */
        return FPL_OK;
    }();
    if (rc != FPL_OK) {
        fprintf(stderr, "fpl_create: %s (%s)\n", fpl_strerror(rc), ctx->err.c_str());
        fpl_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return FPL_OK;
}

void fpl_destroy(fpl_ctx* ctx) {
    if (!ctx) return;
    if (ctx->device >= 0) (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    void* ptrs[] = {ctx->d_cfg, ctx->d_ads, ctx->d_counters, ctx->d_state, ctx->d_frag_off, ctx->d_frag_len,
                    ctx->d_work_ctr, ctx->d_stats_scratch, ctx->d_extra_scratch, ctx->d_extra_flags,
                    ctx->d_stats_flags, ctx->d_frag_cyc, ctx->bm.frags, ctx->bm.regs, ctx->bm.counts,
                    ctx->d_sort_ws, ctx->d_st_off, ctx->d_st_len, ctx->d_st_e, ctx->d_recs, ctx->d_redo, ctx->d_wins};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (auto& sl : ctx->slot) {
        void* sp[] = {sl.d_seq, sl.d_qual, sl.d_off, sl.d_results, sl.d_text, sl.d_nl, sl.d_blk, sl.d_line, sl.d_len, sl.d_hdr};
        for (void* p : sp)
            if (p) (void)hipFree(p);
        if (sl.h_results) (void)hipHostFree(sl.h_results);
        if (sl.h_hdr) (void)hipHostFree(sl.h_hdr);
        if (sl.h_line) (void)hipHostFree(sl.h_line);
        if (sl.ev_parsed) (void)hipEventDestroy(sl.ev_parsed);
        if (sl.ev_h2d) (void)hipEventDestroy(sl.ev_h2d);
        if (sl.ev_kern) (void)hipEventDestroy(sl.ev_kern);
        if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
    }
    if (ctx->s_h2d) (void)hipStreamDestroy(ctx->s_h2d);
    if (ctx->s_d2h) (void)hipStreamDestroy(ctx->s_d2h);
    if (ctx->s_parse) (void)hipStreamDestroy(ctx->s_parse);
    if (ctx->s_aux) (void)hipStreamDestroy(ctx->s_aux);
    if (ctx->s_trim) (void)hipStreamDestroy(ctx->s_trim);
    if (ctx->ev_trim_done) (void)hipEventDestroy(ctx->ev_trim_done);
    for (hipEvent_t e : ctx->ev_batch_done)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev_stats_done)
        if (e) (void)hipEventDestroy(e);
    if (ctx->d_state2) (void)hipFree(ctx->d_state2);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    for (int r = 0; r < fpl_ctx::EV_RING; r++)
        for (int i = 0; i <= N_STAGES; i++)
            if (ctx->ev[r][i]) (void)hipEventDestroy(ctx->ev[r][i]);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

uint32_t fpl_max_cycles(const fpl_ctx* ctx) { return ctx ? ctx->C : 0; }
int32_t fpl_n_adapters(const fpl_ctx* ctx) { return ctx ? ctx->n_adapters : 0; }
size_t fpl_counters_len(const fpl_ctx* ctx) { return ctx ? FPL_COUNTERS_LEN(ctx->C, ctx->n_adapters) : 0; }
void* fpl_counters_device_ptr(fpl_ctx* ctx) { return ctx ? ctx->d_counters : nullptr; }

int fpl_synchronize(fpl_ctx* ctx) {
    if (!ctx) return FPL_ERR_ARG;
    FPL_HIP(hipSetDevice(ctx->device));
    FPL_HIP(hipDeviceSynchronize());
    return FPL_OK;
}

/* cycle-major layout: growing C moves the two Stats tails and appends zero cycles */
int fpl_reserve_cycles(fpl_ctx* ctx, uint32_t max_cycles) {
    if (!ctx) return FPL_ERR_ARG;
    if (max_cycles <= ctx->C) return FPL_OK;
    FPL_HIP(hipSetDevice(ctx->device));
    FPL_HIP(hipDeviceSynchronize());
    long long* nw = nullptr;
    const u32 Co = ctx->C, Cn = max_cycles;
    int r = alloc_counters(ctx, Cn, &nw);
    if (r != FPL_OK) return r;
    for (int k = 0; k < 2; k++) {
        const long long* so = ctx->d_counters + (size_t)k * FPL_STATS_LEN(Co);
        long long* sn = nw + (size_t)k * FPL_STATS_LEN(Cn);
        FPL_HIP(hipMemcpy(sn, so, (size_t)Co * FPL_CYC_STRIDE * sizeof(long long), hipMemcpyDeviceToDevice));
        FPL_HIP(hipMemcpy(sn + (size_t)Cn * FPL_CYC_STRIDE, so + (size_t)Co * FPL_CYC_STRIDE,
                          FPL_STATS_TAIL * sizeof(long long), hipMemcpyDeviceToDevice));
    }
    FPL_HIP(hipMemcpy(nw + FPL_OFF_FR(Cn), ctx->d_counters + FPL_OFF_FR(Co),
                      (FPL_FR_LEN + FPL_KEYHIST_LEN(ctx->n_adapters)) * sizeof(long long), hipMemcpyDeviceToDevice));
    FPL_HIP(hipFree(ctx->d_counters));
    ctx->d_counters = nw;
    ctx->C = Cn;
    return FPL_OK;
}

int fpl_get_counters(fpl_ctx* ctx, int64_t* host_buf, size_t n) {
    if (!ctx || !host_buf || n != fpl_counters_len(ctx)) return FPL_ERR_ARG;
    FPL_HIP(hipSetDevice(ctx->device));
    FPL_HIP(hipDeviceSynchronize());
    FPL_HIP(hipMemcpy(host_buf, ctx->d_counters, n * sizeof(int64_t), hipMemcpyDeviceToHost));
    return FPL_OK;
}

/* RCCL, loaded on first use: a host that never merges across devices does not need the library at all */
namespace {
struct Rccl {
    void* lib = nullptr;
    std::string path; /* what dlopen took */
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string& err) {
        if (lib) return true;
        /* a librccl the process has mapped already (PyTorch-ROCm carries its own under torch/lib, beside its HIP runtime) is THE
           one to use: a second copy would bring a second set of communicator state.  Else the loader's search path, ROCm's
           directory, and the directory of the HIP runtime this library itself resolved to. */
        std::vector<std::string> names;
        if (FILE* maps = fopen("/proc/self/maps", "r")) {
            char line[4096];
            while (fgets(line, sizeof line, maps)) {
                const char* path = strchr(line, '/');
                if (!path || !strstr(path, "librccl.so")) continue;
                std::string s(path);
                while (!s.empty() && (s.back() == '\n' || s.back() == ' ')) s.pop_back();
                names.push_back(s);
                break;
            }
            fclose(maps);
        }
        names.insert(names.end(), {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"});
        Dl_info hip_at;
        if (dladdr((void*)&hipGetDeviceCount, &hip_at) && hip_at.dli_fname) {
            std::string dir(hip_at.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                names.push_back(dir.substr(0, slash) + "/librccl.so.1");
                names.push_back(dir.substr(0, slash) + "/librccl.so");
            }
        }
        for (const std::string& name : names) {
            lib = dlopen(name.c_str(), RTLD_NOW | RTLD_GLOBAL);
            if (lib) {
                path = name;
                break;
            }
        }
        if (!lib) {
            err = std::string("dlopen(librccl): ") + dlerror();
            return false;
        }
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !AllReduce) {
            err = "librccl lacks an expected symbol";
            lib = nullptr;
            return false;
        }
        return true;
    }
};
Rccl g_rccl;
}  // namespace

/* communicators made ahead of the merge (fpl_comm_init), kept for the devices they were made for */
namespace {
struct CommCache {
    std::mutex m;
    std::vector<int> devs;
    std::vector<ncclComm_t> comms;
    bool matches(fpl_ctx** ctxs, int n) const {
        if ((int)devs.size() != n || n == 0) return false;
        for (int i = 0; i < n; i++)
            if (devs[(size_t)i] != ctxs[i]->device) return false;
        return true;
    }
    void drop() { /* (caller holds m) */
        for (ncclComm_t c : comms)
            if (c && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c);
        comms.clear();
        devs.clear();
    }
};
CommCache g_comms;
bool rccl_forced() {
    const char* force = getenv("FPL_RCCL_FORCE");
    return force && atoi(force) > 0;
}
int check_merge_args(fpl_ctx** ctxs, int32_t n) {
    if (!ctxs || n < 1) return FPL_ERR_ARG;
    for (int i = 0; i < n; i++) {
        if (!ctxs[i] || ctxs[i]->n_adapters != ctxs[0]->n_adapters) return FPL_ERR_ARG;
        for (int j = 0; j < i; j++)
            if (ctxs[j]->device == ctxs[i]->device) return FPL_ERR_ARG; /* one context per device */
    }
    return FPL_OK;
}
}  // namespace

int fpl_comm_init(fpl_ctx** ctxs, int32_t n) {
    if (!ctxs && n == 0) { /* give the kept communicators back */
        std::lock_guard<std::mutex> g(g_comms.m);
        g_comms.drop();
        return FPL_OK;
    }
    const int rc0 = check_merge_args(ctxs, n);
    if (rc0 != FPL_OK) return rc0;
    if (n == 1 && !rccl_forced()) return FPL_OK; /* (one context: the merge needs no communicator) */
    std::lock_guard<std::mutex> g(g_comms.m);
    if (g_comms.matches(ctxs, n)) return FPL_OK;
    std::string err;
    if (!g_rccl.load(err)) return FPL_ERR_STATE; /* (fpl_allreduce_counters will say why) */
    g_comms.drop();
    std::vector<int> devs((size_t)n);
    for (int i = 0; i < n; i++) devs[(size_t)i] = ctxs[i]->device;
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    if (g_rccl.CommInitAll(comms.data(), n, devs.data()) != ncclSuccess) return FPL_ERR_HIP;
    g_comms.devs = devs;
    g_comms.comms = comms;
    return FPL_OK;
}

int fpl_allreduce_counters(fpl_ctx** ctxs, int32_t n) {
    const int rc0 = check_merge_args(ctxs, n);
    if (rc0 != FPL_OK) return rc0;
    fpl_ctx* ctx = ctxs[0]; /* (FPL_HIP reports through this one) */
    u32 C = 0;
    for (int i = 0; i < n; i++) {
        if (ctxs[i]->submitted != ctxs[i]->waited) return FPL_ERR_STATE;
        C = std::max(C, ctxs[i]->C);
    }
    for (int i = 0; i < n; i++) {
        const int r = fpl_reserve_cycles(ctxs[i], C);
        if (r != FPL_OK) return r;
        FPL_HIP(hipSetDevice(ctxs[i]->device));
        FPL_HIP(hipDeviceSynchronize());
    }
    /* one context: nothing to merge.  FPL_RCCL_FORCE=1 (a test hook) runs the collective all the same -- a one-rank communicator,
       the in-place sum on the context's stream -- so that the loader, the communicator set-up and the call are exercised on a
       box with a single GPU; the buffer must come out unchanged. */
    if (n == 1 && !rccl_forced()) return FPL_OK;
    /* (the loader, the communicator cache and the library's path are all behind g_comms.m: a host may merge while a thread of
       its own is still inside fpl_comm_init) */
    std::lock_guard<std::mutex> keep(g_comms.m);
    if (!g_rccl.load(ctx->err)) return FPL_ERR_STATE;
#define FPL_NCCL(call)                                                                                   \
    do {                                                                                                 \
        const ncclResult_t r__ = (call);                                                                 \
        if (r__ != ncclSuccess) {                                                                        \
            ctx->err = std::string(#call) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r__) : "rccl error"); \
            rc = FPL_ERR_HIP;                                                                            \
        }                                                                                                \
    } while (0)
    int rc = FPL_OK;
    /* the communicators fpl_comm_init made for exactly these devices, else a set of this call's own */
    const bool kept = g_comms.matches(ctxs, n);
    std::vector<ncclComm_t> own;
    if (!kept) {
        own.assign((size_t)n, nullptr);
        std::vector<int> devs((size_t)n);
        for (int i = 0; i < n; i++) devs[(size_t)i] = ctxs[i]->device;
        FPL_NCCL(g_rccl.CommInitAll(own.data(), n, devs.data()));
        if (rc != FPL_OK) return rc;
    }
    const std::vector<ncclComm_t>& comms = kept ? g_comms.comms : own;
    const size_t len = FPL_COUNTERS_LEN(C, ctx->n_adapters);
    FPL_NCCL(g_rccl.GroupStart());
    for (int i = 0; i < n && rc == FPL_OK; i++) {
        if (hipSetDevice(ctxs[i]->device) != hipSuccess) {
            ctx->err = "hipSetDevice failed inside the all-reduce group";
            rc = FPL_ERR_HIP;
            break;
        }
        FPL_NCCL(g_rccl.AllReduce(ctxs[i]->d_counters, ctxs[i]->d_counters, len, ncclInt64, ncclSum, comms[(size_t)i], ctxs[i]->s_aux));
    }
    FPL_NCCL(g_rccl.GroupEnd());
    for (int i = 0; i < n; i++) {
        if (hipSetDevice(ctxs[i]->device) != hipSuccess || hipStreamSynchronize(ctxs[i]->s_aux) != hipSuccess) {
            if (rc == FPL_OK) ctx->err = "synchronizing the all-reduce failed";
            rc = FPL_ERR_HIP;
        }
    }
    for (ncclComm_t c : own)
        if (c) FPL_NCCL(g_rccl.CommDestroy(c));
#undef FPL_NCCL
    return rc;
}

const char* fpl_rccl_library(void) {
    /* a copy taken under the lock (the loader may be running on another thread); it stays valid until the next call on this thread */
    static thread_local std::string copy;
    std::lock_guard<std::mutex> g(g_comms.m);
    copy = g_rccl.path;
    return copy.c_str();
}

int fpl_assume_inputs_ready(fpl_ctx* ctx, int yes) {
    if (!ctx) return FPL_ERR_ARG;
    ctx->inputs_ready = yes != 0;
    return FPL_OK;
}

int fpl_get_batch_forms(const fpl_ctx* ctx, uint64_t out[6]) {
    if (!ctx || !out) return FPL_ERR_ARG;
    for (int i = 0; i < 6; i++) out[i] = ctx->forms[i];
    return FPL_OK;
}

int fpl_reset_counters(fpl_ctx* ctx) {
    if (!ctx) return FPL_ERR_ARG;
    for (int i = 0; i < 6; i++) ctx->forms[i] = 0;
    FPL_HIP(hipSetDevice(ctx->device));
    FPL_HIP(hipDeviceSynchronize());
    FPL_HIP(hipMemset(ctx->d_counters, 0, fpl_counters_len(ctx) * sizeof(long long)));
    return FPL_OK;
}

static int ensure_scratch(fpl_ctx* ctx, u32 n_reads, uint64_t n_bytes, u32 max_read_len) {
    const size_t slabs = stats_scratch_slabs(n_reads, n_bytes, max_read_len, ctx->n_cu, ctx->tune);
    if (slabs <= ctx->scratch_slabs) return FPL_OK;
    FPL_HIP(hipDeviceSynchronize());
    if (ctx->d_stats_scratch) (void)hipFree(ctx->d_stats_scratch);
    if (ctx->d_stats_flags) (void)hipFree(ctx->d_stats_flags);
    ctx->d_stats_scratch = nullptr;
    ctx->d_stats_flags = nullptr;
    ctx->scratch_slabs = 0;
    const size_t cap = slabs + slabs / 4;
    FPL_HIP(hipMalloc((void**)&ctx->d_stats_scratch, cap * (size_t)FS_SLAB * sizeof(u64)));
    FPL_HIP(hipMalloc((void**)&ctx->d_stats_flags, 2 * cap + 64)); /* slab flags + tile flags: tiles <= slabs, whatever the shape */
    ctx->scratch_slabs = cap;
    return FPL_OK;
}

/* slabs of the post-only pass when it has a stream of its own: FS_EXTRA_BLOCKS per cycle tile */
static int ensure_extra_scratch(fpl_ctx* ctx, u32 n_reads, u32 max_read_len, bool sorted) {
    if (!ctx->overlap || ctx->hcfg.defer) return FPL_OK;
    /* only the sorted pass forks the post-only pass onto the side stream; a batch that takes the plain walk needs none of this */
    if (!sorted) return FPL_OK;
    const u32 n_tiles = cdiv(max_read_len ? max_read_len : 1, FS_T);
    const size_t slabs = (size_t)stats_extra_blocks(n_reads, ctx->tune) * n_tiles;
    if (slabs <= ctx->extra_slabs) return FPL_OK;
    FPL_HIP(hipDeviceSynchronize());
    if (ctx->d_extra_scratch) (void)hipFree(ctx->d_extra_scratch);
    if (ctx->d_extra_flags) (void)hipFree(ctx->d_extra_flags);
    ctx->d_extra_scratch = nullptr;
    ctx->d_extra_flags = nullptr;
    ctx->extra_slabs = 0;
    const size_t cap = slabs + slabs / 4;
    FPL_HIP(hipMalloc((void**)&ctx->d_extra_scratch, cap * (size_t)FS_SLAB * sizeof(u64)));
    FPL_HIP(hipMalloc((void**)&ctx->d_extra_flags, 2 * cap + 64)); /* (slab flags + tile flags) */
    ctx->extra_slabs = cap;
    return FPL_OK;
}

static int ensure_sort_ws(fpl_ctx* ctx, u32 n_reads, uint64_t n_bytes) {
    const u32 per = stats_items_per_slice(n_reads, n_reads ? (u32)(n_bytes / n_reads) : 0, ctx->n_cu, ctx->tune);
    const size_t words = sort_ws_words(stats_sorted_max_slices(n_reads, per, ctx->tune), n_reads);
    if (words <= ctx->sort_ws_cap) return FPL_OK;
    FPL_HIP(hipDeviceSynchronize());
    if (ctx->d_sort_ws) (void)hipFree(ctx->d_sort_ws);
    ctx->d_sort_ws = nullptr;
    ctx->sort_ws_cap = 0;
    const size_t cap = words + words / 4;
    FPL_HIP(hipMalloc((void**)&ctx->d_sort_ws, cap * sizeof(u32)));
    ctx->sort_ws_cap = cap;
    return FPL_OK;
}

/* the fragment / region / piece lists of k_break_mask: capacities grow (25 % headroom) and never shrink, so that a
   run whose batches differ a little in size does not reallocate -- and wait for the device -- on every batch */
static int ensure_break_mask(fpl_ctx* ctx, u32 n_reads, uint64_t n_bytes) {
    if (!ctx->hcfg.defer) return FPL_OK;
    if (!ctx->bm.counts) FPL_HIP(hipMalloc((void**)&ctx->bm.counts, 4 * sizeof(u32)));
    u32 need_f = 0, need_r = 0, need_i = 0;
    break_mask_caps(n_reads, n_bytes, ctx->hcfg.brk, ctx->hcfg.brk_w, ctx->hcfg.msk, ctx->hcfg.msk_w, need_f, need_r, need_i);
    if (ctx->bm.frags && need_f <= ctx->bm.frag_cap && need_r <= ctx->bm.reg_cap && need_i <= ctx->bm.item_cap) return FPL_OK;
    FPL_HIP(hipDeviceSynchronize());
    if (ctx->bm.frags) (void)hipFree(ctx->bm.frags);
    if (ctx->bm.regs) (void)hipFree(ctx->bm.regs);
    if (ctx->d_frag_cyc) (void)hipFree(ctx->d_frag_cyc);
    if (ctx->d_frag_off) (void)hipFree(ctx->d_frag_off);
    if (ctx->d_frag_len) (void)hipFree(ctx->d_frag_len);
    ctx->bm.frags = nullptr;
    ctx->bm.regs = nullptr;
    ctx->d_frag_cyc = nullptr;
    ctx->d_frag_off = nullptr;
    ctx->d_frag_len = nullptr;
    auto grow = [](u32 have, u32 need) -> u32 {
        const uint64_t want = (uint64_t)need + need / 4 + 64;
        const uint64_t cap = want < 0x7FFFFFF0ull ? want : 0x7FFFFFF0ull;
        return have > cap ? have : (u32)cap;
    };
    ctx->bm.frag_cap = grow(ctx->bm.frag_cap, need_f);
    ctx->bm.reg_cap = grow(ctx->bm.reg_cap, need_r);
    ctx->bm.item_cap = grow(ctx->bm.item_cap, need_i);
    FPL_HIP(hipMalloc((void**)&ctx->bm.frags, sizeof(fpl_fragment) * (size_t)ctx->bm.frag_cap));
    FPL_HIP(hipMalloc((void**)&ctx->bm.regs, sizeof(fpl_region) * (size_t)ctx->bm.reg_cap));
    FPL_HIP(hipMalloc((void**)&ctx->d_frag_off, sizeof(uint64_t) * (size_t)ctx->bm.item_cap));
    FPL_HIP(hipMalloc((void**)&ctx->d_frag_len, sizeof(u32) * (size_t)ctx->bm.item_cap));
    FPL_HIP(hipMalloc((void**)&ctx->d_frag_cyc, sizeof(u32) * (size_t)ctx->bm.item_cap));
    return FPL_OK;
}

static int ensure_workspace(fpl_ctx* ctx, u32 n_reads) {
    if (n_reads <= ctx->ws_reads) return FPL_OK;
    /* 25 % headroom, as the other workspaces: a host that cuts its input by BYTES hands in batches whose read counts wander by a few
       per cent, and every new record used to cost a device-wide wait, a dozen hipFree and as many hipMalloc -- 3 to 9 ms each, five or
       six times in the first 60 ms of a run (rocprofv3 timeline of the CLI, tools/cli_timeline.sh) */
    {
        const uint64_t want = (uint64_t)n_reads + n_reads / 4 + 1024;
        n_reads = want > 0xFFFFFFF0ull ? n_reads : (u32)want;
    }
    FPL_HIP(hipDeviceSynchronize());
    if (ctx->d_state) (void)hipFree(ctx->d_state);
    if (ctx->d_state2) (void)hipFree(ctx->d_state2);
    ctx->d_state = ctx->d_state2 = nullptr;
    ctx->ws_reads = 0;
    FPL_HIP(hipMalloc((void**)&ctx->d_state, sizeof(ReadState) * (size_t)n_reads));
    FPL_HIP(hipMalloc((void**)&ctx->d_state2, sizeof(ReadState) * (size_t)n_reads));
    if (ctx->d_recs) (void)hipFree(ctx->d_recs);
    if (ctx->d_redo) (void)hipFree(ctx->d_redo);
    if (ctx->d_wins) (void)hipFree(ctx->d_wins);
    ctx->d_recs = nullptr;
    ctx->d_redo = nullptr;
    ctx->d_wins = nullptr;
    FPL_HIP(hipMalloc((void**)&ctx->d_wins, sizeof(ScanWin) * (size_t)n_reads));
    FPL_HIP(hipMalloc((void**)&ctx->d_recs, sizeof(ScanRec) * (size_t)n_reads));
    FPL_HIP(hipMalloc((void**)&ctx->d_redo, sizeof(RedoItem) * (size_t)n_reads));
    if (ctx->d_st_off) (void)hipFree(ctx->d_st_off);
    if (ctx->d_st_len) (void)hipFree(ctx->d_st_len);
    if (ctx->d_st_e) (void)hipFree(ctx->d_st_e);
    ctx->d_st_off = nullptr;
    ctx->d_st_len = ctx->d_st_e = nullptr;
    FPL_HIP(hipMalloc((void**)&ctx->d_st_off, sizeof(uint64_t) * (size_t)n_reads));
    FPL_HIP(hipMalloc((void**)&ctx->d_st_len, sizeof(u32) * (size_t)n_reads));
    FPL_HIP(hipMalloc((void**)&ctx->d_st_e, sizeof(u32) * (size_t)n_reads));
    if (!ctx->hcfg.defer) { /* (with --break / --mask the item list is sized by ensure_break_mask) */
        if (ctx->d_frag_off) (void)hipFree(ctx->d_frag_off);
        if (ctx->d_frag_len) (void)hipFree(ctx->d_frag_len);
        ctx->d_frag_off = nullptr;
        ctx->d_frag_len = nullptr;
        FPL_HIP(hipMalloc((void**)&ctx->d_frag_off, sizeof(uint64_t) * 2 * (size_t)n_reads));
        FPL_HIP(hipMalloc((void**)&ctx->d_frag_len, sizeof(u32) * 2 * (size_t)n_reads));
    }
    ctx->ws_reads = n_reads;
    return FPL_OK;
}

int fpl_process_batch_device(fpl_ctx* ctx, const uint8_t* d_seq, const uint8_t* d_qual, const uint64_t* d_off,
                             uint32_t n_reads, uint64_t n_bytes, uint32_t max_read_len, fpl_read_result* d_results,
                             void* stream_v) {
    if (!ctx) return FPL_ERR_ARG;
    if (n_reads && (!d_seq || !d_qual || !d_off || !d_results)) return FPL_ERR_ARG;
    if (n_reads > 0x7FFFFFFFu / 2) return FPL_ERR_ARG;
    hipStream_t stream = (hipStream_t)stream_v;
    FPL_HIP(hipSetDevice(ctx->device));
    if (max_read_len > ctx->C) {
        /* (a quarter more than asked for: the longest read so far is a record that keeps being broken by a little, and every
           growth waits for the device and moves the counters) */
        const uint64_t want = (uint64_t)max_read_len + max_read_len / 4;
        int r = fpl_reserve_cycles(ctx, want > 0x7FFFFFFFull ? max_read_len : (u32)want);
        if (r != FPL_OK) return r;
    }
    /* which statistics pass the batch takes: asked ONCE -- the side stream's slabs, the launch sequence and the form counters all
       follow this one answer (a drift between separate askings would size the slabs for one form and launch the other) */
    const bool sorted_form = n_reads && stats_takes_sorted(n_reads, n_bytes, max_read_len, ctx->n_cu, ctx->tune, ctx->hcfg.defer != 0);
    if (n_reads) {
        int r = ensure_workspace(ctx, n_reads);
        if (r != FPL_OK) return r;
        r = ensure_scratch(ctx, n_reads, n_bytes, max_read_len);
        if (r != FPL_OK) return r;
        r = ensure_extra_scratch(ctx, n_reads, max_read_len, sorted_form);
        if (r != FPL_OK) return r;
        r = ensure_sort_ws(ctx, n_reads, n_bytes);
        if (r != FPL_OK) return r;
        r = ensure_break_mask(ctx, n_reads, n_bytes);
        if (r != FPL_OK) return r;
    }
    /* which of the two ReadState[] / work-counter sets this batch takes, and whether its end trims start ahead of the main stream */
    const int par = (int)(ctx->batch_no & 1);
    u32* const work_ctr = ctx->d_work_ctr + par * WORK_CTR_WORDS;
    hipEvent_t inputs_ev = ctx->next_inputs_event;
    ctx->next_inputs_event = nullptr;
    const bool ahead = n_reads && ctx->trim_ahead && ctx->overlap && !ctx->dbg && !ctx->hcfg.defer && ctx->batch_no > 0 &&
                       (inputs_ev || ctx->inputs_ready) && trim_worth_ahead(n_reads, ctx->tune);
    if (n_reads) {
        if (ahead) {
            /* the set was last used two batches ago; the trims also wait for this batch's inputs when an event says when they are in */
            FPL_HIP(hipStreamWaitEvent(ctx->s_trim, ctx->ev_batch_done[par], 0));
            if (ctx->ahead_gate) FPL_HIP(hipStreamWaitEvent(ctx->s_trim, ctx->ev_stats_done[par ^ 1], 0)); /* (the batch before this one) */
            if (inputs_ev) FPL_HIP(hipStreamWaitEvent(ctx->s_trim, inputs_ev, 0));
            FPL_HIP(hipMemsetAsync(work_ctr, 0, WORK_CTR_WORDS * sizeof(u32), ctx->s_trim));
        } else {
            FPL_HIP(hipMemsetAsync(work_ctr, 0, WORK_CTR_WORDS * sizeof(u32), stream));
        }
    }
    if (ctx->hcfg.defer && ctx->bm.counts) FPL_HIP(hipMemsetAsync(ctx->bm.counts, 0, 4 * sizeof(u32), stream));
    BatchArgs a;
    a.seq = d_seq;
    a.qual = d_qual;
    a.off = d_off;
    a.n_reads = n_reads;
    a.n_bytes = n_bytes;
    a.max_read_len = max_read_len;
    a.cfg = ctx->d_cfg;
    a.ads = ctx->d_ads;
    a.state = par ? ctx->d_state2 : ctx->d_state;
    if (ctx->trim_ahead && ctx->overlap) a.ev_stats_done = (void*)ctx->ev_stats_done[par];
    if (ahead) {
        a.trim_stream = ctx->s_trim;
        a.ev_trim_done = (void*)ctx->ev_trim_done;
    }
    a.results = d_results;
    a.frag_off = ctx->d_frag_off;
    a.frag_len = ctx->d_frag_len;
    a.frag_cyc = ctx->d_frag_cyc;
    a.bm = ctx->bm;
    a.defer = ctx->hcfg.defer != 0;
    a.trim_mode = ctx->hcfg.trim_mode;
    a.n_fasta = ctx->hcfg.n_fasta;
    a.scan_short = ctx->hcfg.scan_short != 0;
    a.counters = ctx->d_counters;
    a.C = ctx->C;
    a.work_ctr = work_ctr;
    a.recs = ctx->d_recs;
    a.wins = ctx->d_wins;
    a.redo = ctx->d_redo;
    a.sort_ws = ctx->d_sort_ws;
    a.st_off = ctx->d_st_off;
    a.st_len = ctx->d_st_len;
    a.st_e = ctx->d_st_e;
    a.stats_scratch = ctx->d_stats_scratch;
    a.stats_flags = ctx->d_stats_flags;
    if (ctx->overlap && ctx->d_extra_scratch) {
        a.extra_scratch = ctx->d_extra_scratch;
        a.extra_flags = ctx->d_extra_flags;
        a.aux = ctx->s_aux;
        a.ev_fork = (void*)ctx->ev_fork;
        a.ev_join = (void*)ctx->ev_join;
    }
    a.n_cu = ctx->n_cu;
    a.sorted_form = sorted_form ? 1 : 0;
    a.dbg = ctx->dbg;
    if ((a.dbg & 0xA000) && !ctx->probe_primed) { /* (profiling only: the first batch of a back-only / scan-only context runs whole) */
        a.dbg &= ~0xB000;
        ctx->probe_primed = true;
    }
    a.tune = ctx->tune;
    if (n_reads) { /* which forms this batch takes (the same predicates enqueue_batch asks) */
        ctx->forms[0]++;
        ctx->forms[1] += n_reads;
        ctx->forms[2] += trim_takes_batched(n_reads, a.trim_mode, a.tune) ? 1 : 0;
        ctx->forms[3] += sorted_form ? 1 : 0;
        if (n_reads > ctx->forms[4]) ctx->forms[4] = n_reads;
        ctx->forms[5] += ahead ? 1 : 0;
    }
    const bool timing = ctx->timing != 0;
    const int slot = ctx->ev_calls % fpl_ctx::EV_RING;
    hipError_t ev_err = hipSuccess;
    enqueue_batch(a, stream, [&](int i) {
        if (timing) {
            hipError_t e = hipEventRecord(ctx->ev[slot][i], stream);
            if (e != hipSuccess) ev_err = e;
        }
    });
    FPL_HIP(hipGetLastError());
    FPL_HIP(ev_err);
    if (n_reads) {
        FPL_HIP(hipEventRecord(ctx->ev_batch_done[par], stream));
        ctx->batch_no++;
    }
    if (timing) ctx->ev_calls++;
    return FPL_OK;
}

/* device staging of one slot for a batch of this size (grown with 25 % headroom; a grow waits for the device) */
static int ensure_host_streams(fpl_ctx* ctx) {
    if (ctx->stream) return FPL_OK;
    FPL_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    FPL_HIP(hipStreamCreateWithFlags(&ctx->s_h2d, hipStreamNonBlocking));
    FPL_HIP(hipStreamCreateWithFlags(&ctx->s_d2h, hipStreamNonBlocking));
    FPL_HIP(hipStreamCreateWithFlags(&ctx->s_parse, hipStreamNonBlocking));
    return FPL_OK;
}

static int ensure_host_results(fpl_ctx* ctx, fpl_ctx::Slot& sl, u32 n_reads) {
    if (n_reads <= sl.h_res_cap && sl.h_results) return FPL_OK;
    if (sl.h_results) (void)hipHostFree(sl.h_results);
    sl.h_results = nullptr;
    sl.h_res_cap = 0;
    const u32 cap = n_reads + n_reads / 4 + 1024;
    FPL_HIP(hipHostMalloc((void**)&sl.h_results, sizeof(fpl_read_result) * (size_t)cap, hipHostMallocDefault));
    sl.h_res_cap = cap;
    return FPL_OK;
}
/* host_results: false for a text slot -- its device arrays are sized by the most records its bytes COULD hold (one per 64 bytes),
   the page-locked host copy of the records by what the chunk turns out to have (text_continue): locking 19 MB of pages per slot
   for the 1 900 records of a 32 MB chunk of long reads was 3 ms of the link standing still, three times at the start of a run */
static int ensure_slot(fpl_ctx* ctx, fpl_ctx::Slot& sl, u32 n_reads, uint64_t n_bytes, bool host_results = true) {
    if (n_bytes > sl.st_bytes || !sl.d_seq) {
        FPL_HIP(hipDeviceSynchronize());
        if (sl.d_seq) (void)hipFree(sl.d_seq);
        if (sl.d_qual) (void)hipFree(sl.d_qual);
        sl.d_seq = sl.d_qual = nullptr;
        sl.st_bytes = 0;
        const uint64_t cap = n_bytes + n_bytes / 4 + 64;
        FPL_HIP(hipMalloc((void**)&sl.d_seq, cap));
        FPL_HIP(hipMalloc((void**)&sl.d_qual, cap));
        sl.st_bytes = cap;
    }
    if (n_reads > sl.st_reads) {
        FPL_HIP(hipDeviceSynchronize());
        if (sl.d_off) (void)hipFree(sl.d_off);
        if (sl.d_results) (void)hipFree(sl.d_results);
        sl.d_off = nullptr;
        sl.d_results = nullptr;
        sl.st_reads = 0;
        const u32 cap = n_reads + n_reads / 4 + 16;
        FPL_HIP(hipMalloc((void**)&sl.d_off, sizeof(uint64_t) * ((size_t)cap + 1)));
        FPL_HIP(hipMalloc((void**)&sl.d_results, sizeof(fpl_read_result) * (size_t)cap));
        sl.st_reads = cap;
    }
    if (host_results) return ensure_host_results(ctx, sl, n_reads);
    return FPL_OK;
}

/* ---- FASTQ text in (ABI v7): csrc/text_parse.h ---- */
static int ensure_text_slot(fpl_ctx* ctx, fpl_ctx::Slot& sl, uint64_t n_bytes) {
    const u32 rec_cap = (u32)(n_bytes / 64 + 16);
    int r = ensure_slot(ctx, sl, rec_cap, n_bytes / 2 + 64, false);
    if (r != FPL_OK) return r;
    if (!sl.d_hdr) {
        FPL_HIP(hipMalloc((void**)&sl.d_hdr, sizeof(TextHeader)));
        FPL_HIP(hipHostMalloc((void**)&sl.h_hdr, sizeof(TextHeader), hipHostMallocDefault));
    }
    if (n_bytes > sl.text_cap) {
        FPL_HIP(hipDeviceSynchronize());
        void* old[] = {sl.d_text, sl.d_nl, sl.d_blk, sl.d_line, sl.d_len};
        for (void* p : old)
            if (p) (void)hipFree(p);
        sl.d_text = nullptr;
        sl.d_nl = sl.d_blk = sl.d_line = sl.d_len = nullptr;
        sl.text_cap = 0;
        sl.rec_cap = 0;
        const uint64_t cap = n_bytes + n_bytes / 4 + 4096;
        const u32 rc = (u32)(cap / 64 + 16);
        FPL_HIP(hipMalloc((void**)&sl.d_text, cap + 16));
        FPL_HIP(hipMalloc((void**)&sl.d_nl, sizeof(u32) * 4 * (size_t)rc));
        FPL_HIP(hipMalloc((void**)&sl.d_blk, sizeof(u32) * (size_t)(cap / TP_BLOCK_BYTES + 2)));
        FPL_HIP(hipMalloc((void**)&sl.d_line, sizeof(u32) * 4 * (size_t)rc));
        FPL_HIP(hipMalloc((void**)&sl.d_len, sizeof(u32) * (size_t)rc));
        sl.text_cap = cap;
        sl.rec_cap = rc;
    }
    return FPL_OK;
}

/* stage 2 of a text batch: the header is in -- enqueue the per-read kernels and the way back of the records and line starts */
/* (called by fpl_wait_text only: a submission never waits for a parse, so the next chunk's copy goes out behind this one's at
   once -- no round trip to the host between two chunks on the link -- and a batch that has only been peeked at is in no counter) */
static int text_continue(fpl_ctx* ctx, fpl_ctx::Slot& sl) {
    if (sl.kind != 1 || sl.stage != 1) return FPL_OK;
    sl.stage = 2;
    FPL_HIP(hipEventSynchronize(sl.ev_parsed));
    const TextHeader h = *sl.h_hdr;
    sl.n_reads = 0;
    if (h.status != 0 || h.n_records == 0) return FPL_OK; /* nothing to run: fpl_wait_text reports */
    const u32 n = h.n_records;
    if (n > sl.h_line_cap) {
        if (sl.h_line) (void)hipHostFree(sl.h_line);
        sl.h_line = nullptr;
        sl.h_line_cap = 0;
        const u32 cap = n + n / 4 + 16;
        FPL_HIP(hipHostMalloc((void**)&sl.h_line, sizeof(u32) * 4 * (size_t)cap, hipHostMallocDefault));
        sl.h_line_cap = cap;
    }
    {
        const int rh = ensure_host_results(ctx, sl, n);
        if (rh != FPL_OK) return rh;
    }
    FPL_HIP(hipStreamWaitEvent(ctx->stream, sl.ev_parsed, 0));
    ctx->next_inputs_event = sl.ev_parsed; /* (the end trims may start beside the batch before) */
    const int rd = fpl_process_batch_device(ctx, sl.d_seq, sl.d_qual, sl.d_off, n, h.n_bases, h.max_len, sl.d_results, ctx->stream);
    ctx->next_inputs_event = nullptr;
    if (rd != FPL_OK) return rd;
    FPL_HIP(hipEventRecord(sl.ev_kern, ctx->stream));
    FPL_HIP(hipStreamWaitEvent(ctx->s_d2h, sl.ev_kern, 0));
    FPL_HIP(hipMemcpyAsync(sl.h_results, sl.d_results, sizeof(fpl_read_result) * (size_t)n, hipMemcpyDeviceToHost, ctx->s_d2h));
    FPL_HIP(hipMemcpyAsync(sl.h_line, sl.d_line, sizeof(u32) * 4 * (size_t)n, hipMemcpyDeviceToHost, ctx->s_d2h));
    FPL_HIP(hipEventRecord(sl.ev_done, ctx->s_d2h));
    sl.n_reads = n;
    return FPL_OK;
}
int fpl_process_text_async(fpl_ctx* ctx, const uint8_t* text, uint64_t n_bytes) {
    if (!ctx || (n_bytes && !text)) return FPL_ERR_ARG;
    if (n_bytes > 0xFFFFFFF0ull) return FPL_ERR_ARG; /* (line positions are 32 bits wide: cut the file in smaller chunks) */
    if (ctx->submitted - ctx->waited >= FPL_MAX_IN_FLIGHT) return FPL_ERR_STATE;
    FPL_HIP(hipSetDevice(ctx->device));
    if (ctx->hcfg.defer) return FPL_ERR_STATE; /* (--break / --mask read their fragment lists batch by batch: the CSR entry points) */
    int r = FPL_OK;
    fpl_ctx::Slot& sl = ctx->slot[ctx->submitted % FPL_MAX_IN_FLIGHT];
    sl.kind = 1;
    sl.stage = 2;
    sl.cancelled = false;
    sl.n_reads = 0;
    sl.rc = FPL_OK;
    sl.text_bytes = n_bytes;
    r = ensure_host_streams(ctx);
    if (r != FPL_OK) return r;
    r = ensure_text_slot(ctx, sl, n_bytes);
    if (r != FPL_OK) return r;
    if (n_bytes == 0) {
        memset(sl.h_hdr, 0, sizeof(TextHeader));
        sl.h_hdr->bad_record = ~0ull;
        ctx->submitted++;
        return FPL_OK;
    }
    auto enqueue = [&]() -> int {
        /* the upload on the copy stream, the parse on a stream of its own behind it: the NEXT chunk's upload starts the moment this
           one's is done (with the parse on the copy stream the link sat idle for 140 us between two uploads of 590) */
        FPL_HIP(hipMemcpyAsync(sl.d_text, text, n_bytes, hipMemcpyHostToDevice, ctx->s_h2d));
        FPL_HIP(hipEventRecord(sl.ev_h2d, ctx->s_h2d));
        hipStream_t st = ctx->s_parse;
        FPL_HIP(hipStreamWaitEvent(st, sl.ev_h2d, 0));
        FPL_HIP(hipMemsetAsync(sl.d_hdr, 0, sizeof(TextHeader), st));
        FPL_HIP(hipMemsetAsync(&sl.d_hdr->bad_record, 0xFF, sizeof(u64), st));
        const u32 nblk = (u32)((n_bytes + TP_BLOCK_BYTES - 1) / TP_BLOCK_BYTES);
        const u32 rec_cap = (u32)(n_bytes / 64 + 16);
        hipLaunchKernelGGL(k_text_count, dim3(nblk), dim3(TP_THREADS), 0, st, (const u8*)sl.d_text, (u64)n_bytes, sl.d_blk, sl.d_hdr);
        hipLaunchKernelGGL(k_text_scan, dim3(1), dim3(1024), 0, st, sl.d_blk, nblk, sl.d_hdr);
        hipLaunchKernelGGL(k_text_fill, dim3(nblk), dim3(TP_THREADS), 0, st, (const u8*)sl.d_text, (u64)n_bytes, (const u32*)sl.d_blk, sl.d_nl,
                           4 * rec_cap);
        const u32 rblk = std::min<u32>(std::max<u32>(1u, (rec_cap + 255u) / 256u), 4u * ctx->n_cu);
        hipLaunchKernelGGL(k_text_records, dim3(rblk), dim3(256), 0, st, (const u8*)sl.d_text, (u64)n_bytes, (const u32*)sl.d_nl, rec_cap,
                           sl.d_hdr, sl.d_line, sl.d_len);
        hipLaunchKernelGGL(k_text_offsets, dim3(1), dim3(1024), 0, st, (const u32*)sl.d_len, rec_cap, sl.d_hdr, sl.d_off);
        hipLaunchKernelGGL(k_text_gather, dim3(8 * ctx->n_cu), dim3(256), 0, st, (const u8*)sl.d_text, (const u32*)sl.d_line,
                           (const u32*)sl.d_len, (const uint64_t*)sl.d_off, (const TextHeader*)sl.d_hdr, rec_cap, sl.d_seq, sl.d_qual);
        FPL_HIP(hipGetLastError());
        FPL_HIP(hipMemcpyAsync(sl.h_hdr, sl.d_hdr, sizeof(TextHeader), hipMemcpyDeviceToHost, st));
        FPL_HIP(hipEventRecord(sl.ev_parsed, st));
        return FPL_OK;
    };
    r = enqueue();
    if (r != FPL_OK) {
        if (ctx->s_h2d) (void)hipStreamSynchronize(ctx->s_h2d);
        if (ctx->s_parse) (void)hipStreamSynchronize(ctx->s_parse);
        return r;
    }
    sl.stage = 1;
    ctx->submitted++;
    return FPL_OK;
}

static void text_info(const fpl_ctx::Slot& sl, fpl_text_result* out) {
    const TextHeader& h = *sl.h_hdr;
    memset(out, 0, sizeof(*out));
    out->n_lines = h.n_lines;
    out->bad_record = h.bad_record;
    out->status = (h.status & 1u) ? FPL_TEXT_IRREGULAR : (h.status & 2u) ? FPL_TEXT_TOO_MANY : FPL_TEXT_OK;
    if (out->status == FPL_TEXT_OK) {
        out->n_reads = h.n_records;
        out->n_bases = h.n_bases;
        out->max_read_len = h.max_len;
    }
}

/* the oldest text batch in flight that is neither started nor cancelled: what fpl_peek_text / fpl_start_text / fpl_cancel_text act on */
static fpl_ctx::Slot* text_pending(fpl_ctx* ctx) {
    for (u32 k = ctx->waited; k != ctx->submitted; k++) {
        fpl_ctx::Slot& sl = ctx->slot[k % FPL_MAX_IN_FLIGHT];
        if (sl.kind == 1 && sl.stage == 1 && !sl.cancelled) return &sl;
    }
    return nullptr;
}

int fpl_peek_text(fpl_ctx* ctx, fpl_text_result* out) {
    if (!ctx || !out) return FPL_ERR_ARG;
    fpl_ctx::Slot* sl = text_pending(ctx);
    if (!sl) return FPL_ERR_STATE;
    memset(out, 0, sizeof(*out));
    if (sl->rc != FPL_OK) return sl->rc;
    FPL_HIP(hipSetDevice(ctx->device));
    FPL_HIP(hipEventSynchronize(sl->ev_parsed));
    text_info(*sl, out);
    return FPL_OK;
}

int fpl_start_text(fpl_ctx* ctx) {
    if (!ctx) return FPL_ERR_ARG;
    fpl_ctx::Slot* sl = text_pending(ctx);
    if (!sl) return FPL_ERR_STATE;
    if (sl->rc != FPL_OK) return sl->rc;
    FPL_HIP(hipSetDevice(ctx->device));
    const int r = text_continue(ctx, *sl);
    if (r != FPL_OK) sl->rc = r;
    return r;
}

int fpl_cancel_text(fpl_ctx* ctx) {
    if (!ctx) return FPL_ERR_ARG;
    fpl_ctx::Slot* sl = text_pending(ctx);
    if (!sl) return FPL_ERR_STATE;
    FPL_HIP(hipSetDevice(ctx->device));
    if (sl->rc == FPL_OK) FPL_HIP(hipEventSynchronize(sl->ev_parsed)); /* (its copy and parse read the caller's text) */
    sl->cancelled = true;
    sl->n_reads = 0;
    return FPL_OK;
}

int fpl_wait_text(fpl_ctx* ctx, fpl_text_result* out, const fpl_read_result** results, const uint32_t** line_starts) {
    if (!ctx || !out) return FPL_ERR_ARG;
    if (ctx->submitted == ctx->waited) return FPL_ERR_STATE;
    fpl_ctx::Slot& sl = ctx->slot[ctx->waited % FPL_MAX_IN_FLIGHT];
    if (sl.kind != 1) return FPL_ERR_STATE; /* (a CSR batch: fpl_wait) */
    memset(out, 0, sizeof(*out));
    if (results) *results = nullptr;
    if (line_starts) *line_starts = nullptr;
    FPL_HIP(hipSetDevice(ctx->device));
    if (sl.cancelled) {
        ctx->waited++;
        out->status = FPL_TEXT_CANCELLED;
        out->bad_record = ~0ull;
        return FPL_OK;
    }
    if (sl.rc == FPL_OK) {
        const int r = text_continue(ctx, sl); /* (no-op when fpl_start_text did it) */
        if (r != FPL_OK) sl.rc = r;
    }
    ctx->waited++;
    if (sl.rc != FPL_OK) return sl.rc;
    text_info(sl, out);
    if (out->status != FPL_TEXT_OK || sl.n_reads == 0) return FPL_OK;
    FPL_HIP(hipEventSynchronize(sl.ev_done));
    if (results) *results = sl.h_results;
    if (line_starts) *line_starts = sl.h_line;
    return FPL_OK;
}

int fpl_in_flight(const fpl_ctx* ctx) { return ctx ? (int)(ctx->submitted - ctx->waited) : 0; }

int fpl_wait(fpl_ctx* ctx) {
    if (!ctx) return FPL_ERR_ARG;
    if (ctx->submitted == ctx->waited) return FPL_ERR_STATE;
    fpl_ctx::Slot& sl = ctx->slot[ctx->waited % FPL_MAX_IN_FLIGHT];
    if (sl.kind != 0) return FPL_ERR_STATE; /* (a text batch: fpl_wait_text) */
    ctx->waited++;
    if (sl.rc != FPL_OK) return sl.rc; /* nothing was enqueued behind the failure */
    if (sl.n_reads == 0) return FPL_OK;
    FPL_HIP(hipSetDevice(ctx->device));
    FPL_HIP(hipEventSynchronize(sl.ev_done));
    memcpy(sl.user_results, sl.h_results, sizeof(fpl_read_result) * (size_t)sl.n_reads);
    return FPL_OK;
}

int fpl_process_batch_async(fpl_ctx* ctx, const uint8_t* seq, const uint8_t* qual, const uint64_t* off,
                            uint32_t n_reads, fpl_read_result* results) {
    if (!ctx) return FPL_ERR_ARG;
    if (n_reads && (!seq || !qual || !off || !results)) return FPL_ERR_ARG;
    if (ctx->submitted - ctx->waited >= FPL_MAX_IN_FLIGHT) return FPL_ERR_STATE;
    FPL_HIP(hipSetDevice(ctx->device));
    /* --break / --mask: the fragment lists of the batch in flight live in buffers this batch's kernels reuse */
    if (ctx->hcfg.defer && ctx->submitted != ctx->waited) return FPL_ERR_STATE;
    /* (a text batch in flight keeps waiting for ITS wait: the kernels of this batch go first -- the order of the kernels is free,
       the slots are collected in the order of submission) */
    fpl_ctx::Slot& sl = ctx->slot[ctx->submitted % FPL_MAX_IN_FLIGHT];
    sl.kind = 0;
    sl.n_reads = n_reads;
    sl.user_results = results;
    sl.rc = FPL_OK;
    if (n_reads == 0) {
        ctx->submitted++;
        return FPL_OK;
    }
    const uint64_t n_bytes = off[n_reads];
    u32 max_len = 0;
    for (u32 i = 0; i < n_reads; i++) {
        if (off[i + 1] < off[i] || off[i + 1] - off[i] > 0x7FFFFFFFull) return FPL_ERR_ARG;
        const u32 l = (u32)(off[i + 1] - off[i]);
        if (l > max_len) max_len = l;
    }
    int r = ensure_host_streams(ctx);
    if (r != FPL_OK) return r;
    r = ensure_slot(ctx, sl, n_reads, n_bytes);
    if (r != FPL_OK) return r;
    /* (the slot's previous batch has been waited for -- FPL_MAX_IN_FLIGHT slots, FIFO -- so its buffers are free) */
    auto enqueue = [&]() -> int {
        if (n_bytes) {
            FPL_HIP(hipMemcpyAsync(sl.d_seq, seq, n_bytes, hipMemcpyHostToDevice, ctx->s_h2d));
            FPL_HIP(hipMemcpyAsync(sl.d_qual, qual, n_bytes, hipMemcpyHostToDevice, ctx->s_h2d));
        }
        FPL_HIP(hipMemcpyAsync(sl.d_off, off, sizeof(uint64_t) * ((size_t)n_reads + 1), hipMemcpyHostToDevice, ctx->s_h2d));
        FPL_HIP(hipEventRecord(sl.ev_h2d, ctx->s_h2d));
        FPL_HIP(hipStreamWaitEvent(ctx->stream, sl.ev_h2d, 0));
        ctx->next_inputs_event = sl.ev_h2d; /* (the end trims may start as soon as the copies are in: beside the previous batch) */
        const int rd = fpl_process_batch_device(ctx, sl.d_seq, sl.d_qual, sl.d_off, n_reads, n_bytes, max_len, sl.d_results, ctx->stream);
        ctx->next_inputs_event = nullptr;
        if (rd != FPL_OK) return rd;
        /* the records leave on their own stream, so that they do not queue behind the next batch's input copies */
        FPL_HIP(hipEventRecord(sl.ev_kern, ctx->stream));
        FPL_HIP(hipStreamWaitEvent(ctx->s_d2h, sl.ev_kern, 0));
        FPL_HIP(hipMemcpyAsync(sl.h_results, sl.d_results, sizeof(fpl_read_result) * (size_t)n_reads, hipMemcpyDeviceToHost,
                               ctx->s_d2h));
        FPL_HIP(hipEventRecord(sl.ev_done, ctx->s_d2h));
        return FPL_OK;
    };
    r = enqueue();
    if (r != FPL_OK) {
        /* "nothing is in flight" is what the caller reads into an error here: it recycles the host arrays at once.  Copies or
           kernels that did get enqueued before the failing call may still read them (and the slot): wait them out first. */
        if (ctx->s_h2d) (void)hipStreamSynchronize(ctx->s_h2d);
        if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
        if (ctx->s_d2h) (void)hipStreamSynchronize(ctx->s_d2h);
        return r;
    }
    ctx->submitted++;
    return FPL_OK;
}

int fpl_process_batch(fpl_ctx* ctx, const uint8_t* seq, const uint8_t* qual, const uint64_t* off, uint32_t n_reads,
                      fpl_read_result* results) {
    if (!ctx) return FPL_ERR_ARG;
    if (ctx->submitted != ctx->waited) return FPL_ERR_STATE; /* (collect the asynchronous batches first) */
    if (n_reads == 0) return FPL_OK;
    const int r = fpl_process_batch_async(ctx, seq, qual, off, n_reads, results);
    if (r != FPL_OK) return r;
    return fpl_wait(ctx);
}

/* Large blocks (a host's batch arenas: hundreds of megabytes) are anonymous memory on transparent huge pages, touched once from a few
   threads and then registered with the runtime: page-locking goes page by page, and hipHostMalloc locks 4 KB pages at 4 GB/s -- 0.18 s
   for the CLI's 740 MB arena, every run, before the first byte is read; 370 huge pages are touched in 11 ms and registered in 1.5 ms,
   and the DMA engines read them at the same 56 GB/s (tools/pin_probe.cpp).  Without huge pages (THP off) the same path costs what
   hipHostMalloc costs.  Small blocks, and any failure on the way, take hipHostMalloc.  FPL_NO_HUGE_PIN: measurement hook. */
namespace {
struct HugeBlocks {
    std::mutex mu;
    std::map<void*, std::pair<void*, size_t>> m; /* registered address -> (mapping, its length) */
};
HugeBlocks* huge_blocks() {
    static HugeBlocks* h = new HugeBlocks; /* (never destroyed: a buffer may be freed from a static's destructor) */
    return h;
}
constexpr size_t HUGE_PAGE = 2u << 20;
constexpr size_t HUGE_MIN = 8u << 20;
}  // namespace
void* fpl_host_alloc(size_t bytes) {
    if (bytes >= HUGE_MIN && !getenv("FPL_NO_HUGE_PIN")) {
        const size_t len = (bytes + HUGE_PAGE - 1) & ~(HUGE_PAGE - 1);
        void* const m = mmap(nullptr, len + HUGE_PAGE, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m != MAP_FAILED) {
            char* const a = (char*)(((size_t)m + HUGE_PAGE - 1) & ~(HUGE_PAGE - 1));
            (void)madvise(a, len, MADV_HUGEPAGE);
            /* first touch (the kernel clears a huge page per fault): a few threads side by side, one byte per small page */
            const int nt = (int)std::min<size_t>(4, len / (64u << 20) + 1);
            auto touch = [a, len, nt](int t) {
                const size_t lo = len / HUGE_PAGE * (size_t)t / (size_t)nt * HUGE_PAGE, hi = len / HUGE_PAGE * (size_t)(t + 1) / (size_t)nt * HUGE_PAGE;
                for (size_t o = lo; o < hi; o += 4096) ((volatile char*)a)[o] = 0;
            };
            std::vector<std::thread> th;
            for (int t = 1; t < nt; t++) th.emplace_back(touch, t);
            touch(0);
            for (auto& x : th) x.join();
            if (hipHostRegister(a, len, hipHostRegisterPortable) == hipSuccess) {
                HugeBlocks& h = *huge_blocks();
                std::lock_guard<std::mutex> g(h.mu);
                h.m[a] = std::make_pair(m, len + HUGE_PAGE);
                return a;
            }
            (void)hipGetLastError();
            munmap(m, len + HUGE_PAGE);
        }
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void fpl_host_free(void* p) {
    if (!p) return;
    {
        HugeBlocks& h = *huge_blocks();
        std::unique_lock<std::mutex> g(h.mu);
        auto it = h.m.find(p);
        if (it != h.m.end()) {
            const std::pair<void*, size_t> mp = it->second;
            h.m.erase(it);
            g.unlock();
            (void)hipHostUnregister(p);
            munmap(mp.first, mp.second);
            return;
        }
    }
    (void)hipHostFree(p);
}

int fpl_enable_timing(fpl_ctx* ctx, int enable) {
    if (!ctx) return FPL_ERR_ARG;
    if (enable && !ctx->ev_ready) { /* all or nothing: a ring with holes would hand null events to hipEventRecord later */
        FPL_HIP(hipSetDevice(ctx->device));
        hipError_t bad = hipSuccess;
        for (int r = 0; r < fpl_ctx::EV_RING && bad == hipSuccess; r++)
            for (int i = 0; i <= N_STAGES && bad == hipSuccess; i++) bad = hipEventCreate(&ctx->ev[r][i]);
        if (bad != hipSuccess) {
            for (int r = 0; r < fpl_ctx::EV_RING; r++)
                for (int i = 0; i <= N_STAGES; i++) {
                    if (ctx->ev[r][i]) (void)hipEventDestroy(ctx->ev[r][i]);
                    ctx->ev[r][i] = nullptr;
                }
            ctx->timing = 0;
            FPL_HIP(bad);
        }
        ctx->ev_ready = true;
    }
    ctx->timing = enable ? 1 : 0;
    ctx->ev_calls = 0;
    return FPL_OK;
}

int fpl_get_kernel_times(fpl_ctx* ctx, float* ms, const char** names, int* n, int* n_batches) {
    if (!ctx || !ms || !n) return FPL_ERR_ARG;
    if (ctx->ev_calls <= 0) return FPL_ERR_STATE;
    FPL_HIP(hipSetDevice(ctx->device));
    const int calls = ctx->ev_calls < fpl_ctx::EV_RING ? ctx->ev_calls : fpl_ctx::EV_RING;
    for (int i = 0; i < N_STAGES; i++) ms[i] = 0.f;
    for (int c = 0; c < calls; c++) {
        const int slot = (ctx->ev_calls - 1 - c) % fpl_ctx::EV_RING;
        FPL_HIP(hipEventSynchronize(ctx->ev[slot][N_STAGES]));
        for (int i = 0; i < N_STAGES; i++) {
            float t = 0.f;
            FPL_HIP(hipEventElapsedTime(&t, ctx->ev[slot][i], ctx->ev[slot][i + 1]));
            ms[i] += t;
        }
    }
    for (int i = 0; i < N_STAGES; i++)
        if (names) names[i] = STAGE_NAMES[i];
    *n = N_STAGES;
    if (n_batches) *n_batches = calls;
    return FPL_OK;
}

} /* extern "C" */

extern "C" int fpl_fragment_counts(fpl_ctx* ctx, uint32_t* n_fragments, uint32_t* n_regions) {
    if (!ctx || !n_fragments || !n_regions) return FPL_ERR_ARG;
    *n_fragments = *n_regions = 0;
    if (!ctx->hcfg.defer || !ctx->bm.counts) return FPL_OK;
    FPL_HIP(hipSetDevice(ctx->device));
    FPL_HIP(hipDeviceSynchronize());
    u32 c[4] = {0, 0, 0, 0};
    FPL_HIP(hipMemcpy(c, ctx->bm.counts, sizeof(c), hipMemcpyDeviceToHost));
    if (c[2]) {
        ctx->err = "break/mask lists overflowed their capacity";
        return FPL_ERR_CAPACITY;
    }
    *n_fragments = c[0];
    *n_regions = c[1];
    return FPL_OK;
}

extern "C" int fpl_get_fragments(fpl_ctx* ctx, fpl_fragment* fragments, uint32_t n_fragments, fpl_region* regions,
                                 uint32_t n_regions) {
    if (!ctx || (n_fragments && !fragments) || (n_regions && !regions)) return FPL_ERR_ARG;
    uint32_t nf = 0, nr = 0;
    int r = fpl_fragment_counts(ctx, &nf, &nr);
    if (r != FPL_OK) return r;
    if (n_fragments < nf || n_regions < nr) return FPL_ERR_ARG;
    if (nf) FPL_HIP(hipMemcpy(fragments, ctx->bm.frags, sizeof(fpl_fragment) * (size_t)nf, hipMemcpyDeviceToHost));
    if (nr) FPL_HIP(hipMemcpy(regions, ctx->bm.regs, sizeof(fpl_region) * (size_t)nr, hipMemcpyDeviceToHost));
    std::sort(fragments, fragments + nf, [](const fpl_fragment& a, const fpl_fragment& b) {
        return a.read != b.read ? a.read < b.read : a.seq_no < b.seq_no;
    });
    return FPL_OK;
}

#ifdef FPL_PROF
/* profiling builds only: read (and clear) the section timers the kernels accumulate */
extern "C" int fpl_debug_prof(unsigned long long* out, int n) {
    unsigned long long tmp[64];
    if (hipMemcpyFromSymbol(tmp, HIP_SYMBOL(fpl::g_fpl_prof), sizeof(tmp)) != hipSuccess) return -1;
    for (int i = 0; i < n && i < 64; i++) out[i] = tmp[i];
    memset(tmp, 0, sizeof(tmp));
    if (hipMemcpyToSymbol(HIP_SYMBOL(fpl::g_fpl_prof), tmp, sizeof(tmp)) != hipSuccess) return -1;
    return 0;
}
#endif

/* the counting of the detection: tables in device memory (the caller frees what `bufs` lists) */
struct KmerTables {
    u8* d_seq = nullptr;
    uint64_t* d_off = nullptr;
    u32* d_counts = nullptr;
    unsigned long long *d_pos = nullptr, *d_total = nullptr;
    void release() {
        void* ptrs[] = {d_seq, d_off, d_counts, d_pos, d_total};
        for (void* q : ptrs)
            if (q) (void)hipFree(q);
    }
};
static int count_end_kmers_device(int32_t device, const uint8_t* seq, const uint64_t* off, uint32_t n_reads, int32_t side,
                                  int32_t shift_tail, KmerTables& t) {
    if (!off || (n_reads && !seq) || side < 0 || side > 1 || shift_tail < 0) return FPL_ERR_ARG;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0 || device < 0 || device >= n_dev) return FPL_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return FPL_ERR_NO_DEVICE;
    const size_t n_keys = (size_t)pick::NKEYS;
    const uint64_t n_bytes = n_reads ? off[n_reads] : 0;
    int rc = FPL_OK;
    auto ok = [&](hipError_t e) {
        if (e != hipSuccess && rc == FPL_OK) rc = FPL_ERR_HIP;
        return e == hipSuccess;
    };
    if (ok(hipMalloc((void**)&t.d_seq, n_bytes ? n_bytes : 1)) && ok(hipMalloc((void**)&t.d_off, sizeof(uint64_t) * ((size_t)n_reads + 1))) &&
        ok(hipMalloc((void**)&t.d_counts, sizeof(u32) * n_keys)) && ok(hipMalloc((void**)&t.d_pos, sizeof(unsigned long long) * n_keys)) &&
        ok(hipMalloc((void**)&t.d_total, sizeof(unsigned long long)))) {
        ok(hipMemcpy(t.d_seq, seq, n_bytes, hipMemcpyHostToDevice));
        ok(hipMemcpy(t.d_off, off, sizeof(uint64_t) * ((size_t)n_reads + 1), hipMemcpyHostToDevice));
        ok(hipMemset(t.d_counts, 0, sizeof(u32) * n_keys));
        ok(hipMemset(t.d_pos, 0, sizeof(unsigned long long) * n_keys));
        ok(hipMemset(t.d_total, 0, sizeof(unsigned long long)));
        if (rc == FPL_OK && n_reads) {
            u32 blocks = (n_reads + 3) / 4;
            if (blocks > 8192) blocks = 8192;
            hipLaunchKernelGGL(k_count_end_kmers, dim3(blocks), dim3(256), 0, 0, (const u8*)t.d_seq, (const uint64_t*)t.d_off, n_reads,
                               (int)side, (int)shift_tail, t.d_counts, t.d_pos, t.d_total);
            ok(hipGetLastError());
        }
    }
    return rc;
}

int fpl_count_end_kmers(int32_t device, const uint8_t* seq, const uint64_t* off, uint32_t n_reads, int32_t side, int32_t shift_tail,
                        uint32_t* counts, uint64_t* position_acc, uint64_t* total) {
    if (!counts || !position_acc || !total) return FPL_ERR_ARG;
    KmerTables t;
    int rc = count_end_kmers_device(device, seq, off, n_reads, side, shift_tail, t);
    if (rc == FPL_OK) {
        const size_t n_keys = (size_t)pick::NKEYS;
        if (hipMemcpy(counts, t.d_counts, sizeof(u32) * n_keys, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(position_acc, t.d_pos, sizeof(unsigned long long) * n_keys, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(total, t.d_total, sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
            rc = FPL_ERR_HIP;
    }
    t.release();
    return rc;
}

static_assert(sizeof(fpl_adapter_pick) >= sizeof(pick::Pick) && sizeof(((fpl_adapter_pick*)0)->seq) >= sizeof(((pick::Pick*)0)->seq),
              "the ABI record holds what the kernel writes");
int fpl_pick_adapter(int32_t device, const uint8_t* seq, const uint64_t* off, uint32_t n_reads, int32_t side, int32_t shift_tail,
                     int32_t is_rna, fpl_adapter_pick* out) {
    if (!out) return FPL_ERR_ARG;
    KmerTables t;
    pick::Pick* d_pick = nullptr;
    int rc = count_end_kmers_device(device, seq, off, n_reads, side, shift_tail, t);
    if (rc == FPL_OK && hipMalloc((void**)&d_pick, sizeof(pick::Pick)) != hipSuccess) rc = FPL_ERR_HIP;
    if (rc == FPL_OK) {
        hipLaunchKernelGGL(k_pick_adapter, dim3(1), dim3(1024), 0, 0, (const u32*)t.d_counts, (const unsigned long long*)t.d_pos,
                           (int)(is_rna != 0), d_pick);
        pick::Pick p;
        unsigned long long total = 0;
        if (hipGetLastError() != hipSuccess || hipMemcpy(&p, d_pick, sizeof(p), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(&total, t.d_total, sizeof(total), hipMemcpyDeviceToHost) != hipSuccess)
            rc = FPL_ERR_HIP;
        else {
            memset(out, 0, sizeof(*out));
            out->key = p.key;
            out->count = p.count;
            out->total_key = p.total_key;
            out->len = p.len;
            out->total = total;
            memcpy(out->seq, p.seq, sizeof(p.seq));
        }
    }
    if (d_pick) (void)hipFree(d_pick);
    t.release();
    return rc;
}

#ifdef FPL_PROF_BLOCKS
extern "C" int fpl_debug_read_blockprof(void* dst, size_t bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(fpl::g_blockprof), bytes) == hipSuccess ? 0 : -1;
}
#endif
