// Host side of the C ABI declared in include/genomad_nn.h.
#include <algorithm>
#include <map>
#include <mutex>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "gnn_common.h"

namespace gnn {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }

// Debug / measurement switches read from the environment (GNN_ASYNC_EVENT_WAIT, GNN_NO_BACKEND_OVERLAP, GNN_BACKEND_OVERLAP, GNN_DEBUG_POISON,
// GNN_NO_PAD_SKIP, GNN_NO_TIME_SPLIT, GNN_LOGITS_F32): each is read ONCE per process, and a switch that is set says so on stderr -
// a stray variable in a user's environment must not silently change ordering or arithmetic.
bool debug_switch(const char* name) {
    static std::mutex mu;
    static std::map<std::string, bool> seen;
    std::lock_guard<std::mutex> lock(mu);
    auto it = seen.find(name);
    if (it != seen.end()) return it->second;
    const bool on = std::getenv(name) != nullptr;
    if (on)
        std::fprintf(stderr, "libgenomad_nn_hip: debug switch %s is set: NOT the production path (ordering or arithmetic may differ)\n", name);
    seen[name] = on;
    return on;
}

struct ProfScope {
    gnn_ctx* ctx;
    int id;
    hipEvent_t a = nullptr, b = nullptr;
    ProfScope(gnn_ctx* c, int kid) : ctx(c), id(kid) {
        if (!ctx->profile) return;
        auto take = [&]() {
            hipEvent_t e = nullptr;
            if (!ctx->event_pool.empty()) {
                e = ctx->event_pool.back();
                ctx->event_pool.pop_back();
            } else if (hipEventCreate(&e) != hipSuccess) {
                e = nullptr;
            }
            return e;
        };
        a = take();
        b = take();
        if (a) (void)hipEventRecord(a, ctx->stream);
    }
    ~ProfScope() {
        if (!ctx->profile || !a || !b) return;
        (void)hipEventRecord(b, ctx->stream);
        ctx->prof[id].pending.emplace_back(a, b);
    }
};

template <typename Tp>
static int upload(gnn_ctx* ctx, const Tp* host, size_t count, Tp** dev) {
    void* p = nullptr;
    GNN_HIP(hipMalloc(&p, count * sizeof(Tp)));
    ctx->owned.push_back(p);
    GNN_HIP(hipMemcpy(p, host, count * sizeof(Tp), hipMemcpyHostToDevice));
    *dev = static_cast<Tp*>(p);
    return GNN_OK;
}

static int dev_buffer(gnn_ctx* ctx, size_t bytes, void** out) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();      // the runtime keeps the failure as its "last error": a caller that retries with a smaller size would
                                      // otherwise read THIS out-of-memory behind its next, successful kernel launch
        set_error("hipMalloc of " + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e));
        return GNN_ERR_NOMEM;
    }
    *out = p;
    return GNN_OK;
}

static void free_ws(Workspace& ws) {
    auto f = [](auto*& p) {
        if (p) (void)hipFree(p);
        p = nullptr;
    };
    f(ws.tokens);
    f(ws.x[0]);
    f(ws.x[1]);
    f(ws.x[2]);
    f(ws.mp);
    f(ws.m);
    f(ws.yp);
    f(ws.logits);
    f(ws.alpha);
    f(ws.feat);
    ws.chunk = 0;
    ws.x_chunk = 0;
}

// bytes of workspace per window of a fused launch (mp, m, yp, logits, alpha, feat): 0.86 MB
constexpr size_t WS_BYTES_PER_WINDOW = ((size_t)2 * NPAIR + 2 * NP + (size_t)2 * POOLED * C + 4 * POOLED + FEAT) * sizeof(float);
constexpr int64_t MIN_CHUNK = 256;       // one round of workgroups

// Make sure the workspace holds `chunk` windows (and the f32 activation buffers `x_chunk`).
static int ensure_ws(gnn_ctx* ctx, Workspace& ws, int64_t chunk, int64_t x_chunk) {
    if (ws.chunk < chunk || ws.x_chunk < x_chunk) {      // growing: nothing may still be reading the old buffers, on either stream
        if (ctx->stream2) GNN_HIP(hipStreamSynchronize(ctx->stream2));
        ctx->back_pending[0] = ctx->back_pending[1] = false;
    }
    if (ws.chunk < chunk) {
        GNN_HIP(hipStreamSynchronize(ctx->stream));
        auto re = [&](auto*& p, size_t bytes) -> int {
            if (p) (void)hipFree(p);
            p = nullptr;
            void* q = nullptr;
            int rc = dev_buffer(ctx, bytes, &q);
            p = static_cast<std::remove_reference_t<decltype(p)>>(q);
            return rc;
        };
        // a failed allocation leaves NO workspace behind (chunk = x_chunk = 0), never a half-grown one whose
        // stale size would let a later, smaller call launch kernels on null pointers
        int rc = GNN_OK;
        if (!rc) rc = re(ws.mp, (size_t)chunk * 2 * NPAIR * sizeof(float));
        if (!rc) rc = re(ws.m, (size_t)chunk * 2 * NP * sizeof(float));
        if (!rc) rc = re(ws.yp, (size_t)chunk * 2 * POOLED * C * sizeof(float));
        if (!rc) rc = re(ws.logits, (size_t)chunk * 2 * POOLED * sizeof(float));
        if (!rc) rc = re(ws.alpha, (size_t)chunk * 2 * POOLED * sizeof(float));
        if (!rc) rc = re(ws.feat, (size_t)chunk * FEAT * sizeof(float));
        if (rc) {
            free_ws(ws);
            return rc;
        }
        ws.chunk = chunk;
    }
    if (ws.x_chunk < x_chunk) {
        GNN_HIP(hipStreamSynchronize(ctx->stream));
        ws.x_chunk = 0;
        for (int i = 0; i < 3; ++i) {
            if (ws.x[i]) (void)hipFree(ws.x[i]);
            ws.x[i] = nullptr;
        }
        if (ws.tokens) (void)hipFree(ws.tokens);
        ws.tokens = nullptr;
        for (int i = 0; i < 3; ++i) {
            void* q = nullptr;
            int rc = dev_buffer(ctx, (size_t)x_chunk * T * C * sizeof(float), &q);
            if (rc) {
                free_ws(ws);
                return rc;
            }
            ws.x[i] = static_cast<float*>(q);
        }
        void* q = nullptr;
        int rc = dev_buffer(ctx, (size_t)x_chunk * T * sizeof(uint16_t), &q);
        if (rc) {
            free_ws(ws);
            return rc;
        }
        ws.tokens = static_cast<uint16_t*>(q);
        ws.x_chunk = x_chunk;
    }
    return GNN_OK;
}

int flush_backend(gnn_ctx* ctx) {
    for (int i = 0; i < 2; ++i)
        if (ctx->back_pending[i]) {
            GNN_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_back[i], 0));
            ctx->back_pending[i] = false;
        }
    return GNN_OK;
}

int finish_pending(gnn_ctx* ctx) {
    if (ctx->back_pending[0] || ctx->back_pending[1]) {
        GNN_HIP(hipStreamSynchronize(ctx->stream2));
        ctx->back_pending[0] = ctx->back_pending[1] = false;
    }
    return GNN_OK;
}

// Every entry point but gnn_classify_dev_async: what an earlier asynchronous classification left on the second stream is
// finished before this call does anything.  A host-side wait, not an event wait on ctx->stream: these entry points copy
// between host and device, allocate and free, and read the workspaces - none of them is on a path where the few hundred
// microseconds matter, and nothing they do can then depend on how a cross-stream dependency is resolved
// (profiles/history/r02c6_async_flake.md).  Callers that never use the asynchronous entry point never have anything pending here.
static int check_ctx(gnn_ctx* ctx, bool flush = true) {
    if (!ctx) {
        set_error("ctx is NULL");
        return GNN_ERR_ARG;
    }
    GNN_HIP(hipSetDevice(ctx->device));
    if (!flush) return GNN_OK;
    // GNN_ASYNC_EVENT_WAIT=1 (scripts/async_hunt.py only): the ordering round 2 first shipped, an event wait on ctx->stream
    static const bool event_wait = debug_switch("GNN_ASYNC_EVENT_WAIT");
    return event_wait ? flush_backend(ctx) : finish_pending(ctx);
}

// conv1 pair tables.  conv1 on the one-hot input is sum_k W1[k][tok[t-5+k]] (model.py:11 +
// igloo.py:45-47).  Tokens of adjacent positions share 3 bases, so the pair (tok[s], tok[s+1])
// has only 1795 possible values (PAIR_ROWS, see pair_row() in gnn_fused_common.h); tabulating
// W1[2j][a] + W1[2j+1][b] for the three tap pairs j halves the rows the fused kernels gather.
void build_conv1_pair_tables(const float* k1, std::vector<float>& pt) {
    pt.assign((size_t)3 * PAIR_ROWS * C, 0.f);
    auto row = [&](int j, int r) { return &pt[((size_t)j * PAIR_ROWS + r) * C]; };
    auto add = [&](float* dst, int k, int tok) {
        const float* src = k1 + ((size_t)k * GNN_DEPTH + tok) * C;
        for (int c = 0; c < C; ++c) dst[c] += src[c];
    };
    for (int j = 0; j < 3; ++j) {
        for (int code5 = 0; code5 < 1024; ++code5) {          // both 4-mers valid: a 5-mer
            add(row(j, code5), 2 * j, 1 + (code5 >> 2));
            add(row(j, code5), 2 * j + 1, 1 + (code5 & 255));
        }
        for (int b = 1; b <= 256; ++b) {                       // first 4-mer has an N (token 0)
            add(row(j, 1024 + b - 1), 2 * j, 0);
            add(row(j, 1024 + b - 1), 2 * j + 1, b);
        }
        for (int a = 1; a <= 256; ++a) {                       // second 4-mer has an N
            add(row(j, 1280 + a - 1), 2 * j, a);
            add(row(j, 1280 + a - 1), 2 * j + 1, 0);
        }
        add(row(j, 1536), 2 * j, 0);                           // both have an N
        add(row(j, 1536), 2 * j + 1, 0);
        // row 1537: both positions before the window start (causal zero padding) -> zeros
        for (int b = 0; b <= 256; ++b) add(row(j, 1538 + b), 2 * j + 1, b);   // only the first is absent
    }
}

// ---- persistent staging of the host-buffer entry points (gnn_classify, gnn_debug_forward)
constexpr int64_t STAGE_MAX_WINDOWS = 32768;          // windows per slab: 197 MB of bases on the device at most
constexpr size_t PIN_BYTES = (size_t)8 << 20;         // one bounce buffer

void free_stage(gnn_ctx* ctx) {
    if (ctx->stage_bases) (void)hipFree(ctx->stage_bases);
    if (ctx->stage_scores) (void)hipFree(ctx->stage_scores);
    if (ctx->stage_scores_host) (void)hipHostFree(ctx->stage_scores_host);
    ctx->stage_bases = nullptr;
    ctx->stage_scores = nullptr;
    ctx->stage_scores_host = nullptr;
    ctx->stage_windows = 0;
    for (int i = 0; i < 2; ++i) {
        if (ctx->pin[i]) (void)hipHostFree(ctx->pin[i]);
        if (ctx->pin_ev[i]) (void)hipEventDestroy(ctx->pin_ev[i]);
        ctx->pin[i] = nullptr;
        ctx->pin_ev[i] = nullptr;
        ctx->pin_busy[i] = false;
    }
}

static int ensure_stage(gnn_ctx* ctx, int64_t windows) {
    for (int i = 0; i < 2; ++i)
        if (!ctx->pin[i]) {
            GNN_HIP(hipHostMalloc(&ctx->pin[i], PIN_BYTES, hipHostMallocDefault));
            GNN_HIP(hipEventCreateWithFlags(&ctx->pin_ev[i], hipEventDisableTiming));
        }
    if (ctx->stage_windows >= windows) return GNN_OK;
    GNN_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t want = std::max<int64_t>(windows, std::min<int64_t>(2 * ctx->stage_windows, STAGE_MAX_WINDOWS));
    if (ctx->stage_bases) (void)hipFree(ctx->stage_bases);
    if (ctx->stage_scores) (void)hipFree(ctx->stage_scores);
    if (ctx->stage_scores_host) (void)hipHostFree(ctx->stage_scores_host);
    ctx->stage_bases = nullptr;
    ctx->stage_scores = nullptr;
    ctx->stage_scores_host = nullptr;
    ctx->stage_windows = 0;
    void *b = nullptr, *sc = nullptr, *sh = nullptr;
    int rc = dev_buffer(ctx, (size_t)want * W, &b);
    if (!rc) rc = dev_buffer(ctx, (size_t)want * GNN_CLASSES * sizeof(float), &sc);
    if (!rc && hipHostMalloc(&sh, (size_t)want * GNN_CLASSES * sizeof(float), hipHostMallocDefault) != hipSuccess) {
        set_error("hipHostMalloc of the score landing buffer failed");
        rc = GNN_ERR_NOMEM;
    }
    if (rc) {
        if (b) (void)hipFree(b);
        if (sc) (void)hipFree(sc);
        return rc;
    }
    ctx->stage_bases = static_cast<uint8_t*>(b);
    ctx->stage_scores = static_cast<float*>(sc);
    ctx->stage_scores_host = static_cast<float*>(sh);
    ctx->stage_windows = want;
    return GNN_OK;
}

// host windows -> the device slab, through the two bounce buffers (pageable source: the runtime would otherwise pin or
// stage it anew on every call)
static int stage_upload(gnn_ctx* ctx, const uint8_t* src, size_t bytes) {
    int j = 0;
    for (size_t off = 0; off < bytes; off += PIN_BYTES, j ^= 1) {
        const size_t len = std::min(PIN_BYTES, bytes - off);
        if (ctx->pin_busy[j]) GNN_HIP(hipEventSynchronize(ctx->pin_ev[j]));
        std::memcpy(ctx->pin[j], src + off, len);
        GNN_HIP(hipMemcpyAsync(ctx->stage_bases + off, ctx->pin[j], len, hipMemcpyHostToDevice, ctx->stream));
        GNN_HIP(hipEventRecord(ctx->pin_ev[j], ctx->stream));
        ctx->pin_busy[j] = true;
    }
    return GNN_OK;
}

// One pass of the hot path over n windows whose bases are on the device.
int classify_chunks(gnn_ctx* ctx, const uint8_t* bases_dev, int64_t n, int precision, float* scores_dev, bool defer_last) {
    if (!ctx->has_weights) {
        set_error("gnn_load_weights has not been called");
        return GNN_ERR_STATE;
    }
    if (precision == GNN_PREC_BF16 || precision == GNN_PREC_F16C8) {
        set_error(std::string(precision == GNN_PREC_BF16 ? "GNN_PREC_BF16" : "GNN_PREC_F16C8") +
                  " was removed in round 6 (it fails the 1e-4 score tolerance; the enum value is kept so that old callers get this error)");
        return GNN_ERR_STATE;
    }
    if (precision != GNN_PREC_F32 && precision != GNN_PREC_BF16X3 && precision != GNN_PREC_F16X3 && precision != GNN_PREC_F16C6 &&
        precision != GNN_PREC_F16X3TC && precision != GNN_PREC_F16X3TK) {
        set_error("unknown precision " + std::to_string(precision));
        return GNN_ERR_ARG;
    }
    if (precision == GNN_PREC_F16X3TK && !(ctx->w.tk_x2_tbl && ctx->w.tk_mpa_tbl)) {
        set_error("GNN_PREC_F16X3TK needs the k-mer tables: call gnn_build_kmer_tables first (it answers GNN_ERR_NOMEM on a device without ~190 GB free)");
        return GNN_ERR_STATE;
    }
    const bool f32 = precision == GNN_PREC_F32;
    int64_t chunk = std::min<int64_t>(f32 ? ctx->chunk_f32 : ctx->chunk_fused, std::max<int64_t>(n, 1));
    // Workspace growth of a fused launch shape (WS_BYTES_PER_WINDOW each, 13 GB at the default 16384): the library default is
    // clamped to a quarter of the device memory that is free right now (an integrator on a shared or partitioned GPU never asked
    // for 13 GB), and whatever size is asked for is halved and retried when the allocation fails.  The size that worked becomes
    // the ctx's chunk, so later calls (and the taps limit of gnn_debug_forward) see one consistent value.
    if (!f32 && ctx->ws.chunk < chunk) {
        if (!ctx->chunk_explicit) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                const int64_t cap = std::max<int64_t>(MIN_CHUNK, (int64_t)(free_b / 4 / WS_BYTES_PER_WINDOW) / MIN_CHUNK * MIN_CHUNK);
                if (cap < chunk) {
                    std::fprintf(stderr, "libgenomad_nn_hip: %lld windows per launch need %.1f GB of workspace, %.1f GB of device memory are free: "
                                 "using launches of %lld windows (gnn_set_chunk overrides)\n", (long long)chunk,
                                 chunk * (double)WS_BYTES_PER_WINDOW / 1e9, free_b / 1e9, (long long)cap);
                    chunk = cap;
                }
            }
        }
    }
    int rc = ensure_ws(ctx, ctx->ws, chunk, f32 ? chunk : 0);
    while (rc == GNN_ERR_NOMEM && !f32 && chunk > MIN_CHUNK) {
        chunk = std::max<int64_t>(MIN_CHUNK, chunk / 2);
        std::fprintf(stderr, "libgenomad_nn_hip: workspace allocation failed, retrying with launches of %lld windows\n", (long long)chunk);
        rc = ensure_ws(ctx, ctx->ws, chunk, 0);
    }
    if (rc) return rc;
    if (!f32 && chunk < std::min<int64_t>(ctx->chunk_fused, std::max<int64_t>(n, 1))) ctx->chunk_fused = chunk;
    // An asynchronous call (gnn_classify_dev_async, or whatever one left pending): the back end of chunk i (five small, mostly
    // HBM-bound kernels, 4 % of the time) is enqueued on a second stream and runs beside the front end of chunk i+1 (of this call
    // or of the next one); two workspaces alternate.  A synchronous multi-chunk call no longer does that (rounds 2-4 did): beside
    // the power-bound default kernel the overlapped back end costs the front end more than it saves - 184.6 vs 180.4 k windows/s
    // at 8192 windows per launch, 185.9 vs 183.5 k at 16384 (profiles/r04/backend_overlap_ab.txt) - and the second workspace
    // is not allocated.  GNN_BACKEND_OVERLAP=1 brings the old policy back for A/B runs, GNN_NO_BACKEND_OVERLAP=1 serialises
    // the asynchronous path too.
    static const bool allow_overlap = !debug_switch("GNN_NO_BACKEND_OVERLAP");
    static const bool chunk_overlap = debug_switch("GNN_BACKEND_OVERLAP");
    const bool pending = ctx->back_pending[0] || ctx->back_pending[1];
    if (pending && (f32 || !allow_overlap)) {
        if ((rc = flush_backend(ctx))) return rc;
    }
    bool overlap = allow_overlap && !f32 && ((chunk_overlap && n > chunk) || defer_last || pending);
    // a second workspace of `chunk` windows did not fit before: do not try again (up to six failing hipMallocs and a synchronisation
    // of both streams per call) until a smaller launch shape is asked for - the asynchronous path runs in order on one workspace
    if (overlap && ctx->alt_failed_chunk > 0 && chunk >= ctx->alt_failed_chunk && ctx->ws_alt.chunk < chunk) {
        if ((rc = flush_backend(ctx))) return rc;
        overlap = false;
    }
    if (overlap && (rc = ensure_ws(ctx, ctx->ws_alt, chunk, 0)) == GNN_ERR_NOMEM) {
        // no room for the second workspace (the asynchronous path costs 2 x the workspace): run in order on one
        std::fprintf(stderr, "libgenomad_nn_hip: no device memory for the second workspace of %lld windows: asynchronous calls run in order "
                     "(back end not overlapped) from here on\n", (long long)chunk);
        ctx->alt_failed_chunk = chunk;
        if ((rc = flush_backend(ctx))) return rc;
        overlap = false;
    } else if (rc) {
        return rc;
    }
    if (overlap) {
        if (!ctx->stream2) {
            GNN_HIP(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
            for (int i = 0; i < 2; ++i) {
                GNN_HIP(hipEventCreateWithFlags(&ctx->ev_front[i], hipEventDisableTiming));
                GNN_HIP(hipEventCreateWithFlags(&ctx->ev_back[i], hipEventDisableTiming));
            }
        }
    }
    struct StreamGuard {           // launchers read ctx->stream / ctx->ws: both are switched per phase and restored on every path
        gnn_ctx* c;
        hipStream_t main;
        ~StreamGuard() { c->stream = main; }
    } guard{ctx, ctx->stream};
    for (int64_t a = 0; a < n; a += chunk) {
        const int64_t m = std::min(chunk, n - a);
        const uint8_t* b = bases_dev + a * W;
        // A window buffer that is not 4-byte aligned (possible only for device pointers a caller offsets by hand; W is a multiple of
        // 4, so every chunk of it is misaligned alike): the streaming kernels fetch bases as aligned dwords, so the chunk goes through
        // one aligned staging copy (6 KB per window, on the stream the front end runs on) and then through the SAME kernel - the
        // scores do not depend on where the caller's buffer starts (nn_classification.py:316-317: any batch, any offset).
        if (!f32 && (reinterpret_cast<uintptr_t>(b) & 3u)) {
            if (ctx->align_windows < m) {
                GNN_HIP(hipStreamSynchronize(guard.main));
                if (ctx->align_buf) (void)hipFree(ctx->align_buf);
                ctx->align_buf = nullptr;
                ctx->align_windows = 0;
                void* q = nullptr;
                if ((rc = dev_buffer(ctx, (size_t)chunk * W, &q))) return rc;
                ctx->align_buf = static_cast<uint8_t*>(q);
                ctx->align_windows = chunk;
            }
            GNN_HIP(hipMemcpyAsync(ctx->align_buf, b, (size_t)m * W, hipMemcpyDeviceToDevice, guard.main));
            b = ctx->align_buf;
        }
        const int buf = ctx->buf_cur;
        if (overlap && ctx->back_pending[buf]) {      // the back end that last used this workspace
            GNN_HIP(hipStreamWaitEvent(guard.main, ctx->ev_back[buf], 0));
            ctx->back_pending[buf] = false;
        }
        // GNN_DEBUG_POISON=1 (debug aid): every workspace tensor is filled with NaN bit patterns before the front end runs,
        // so a kernel that reads something this launch has not written yet turns the scores into NaN instead of reading the
        // previous launch's values (which are the right ones whenever the same windows are classified again)
        static const bool poison = debug_switch("GNN_DEBUG_POISON");
        if (poison) {
            // ctx->ws is the workspace THIS chunk's front end writes (the swap happens after the launch); every tensor was sized
            // for ws.chunk >= m windows by ensure_ws above, for the f32 path as for the fused ones
            const Workspace& ws = ctx->ws;
            const size_t mm = (size_t)std::min<int64_t>(m, ws.chunk);
            auto fill = [&](float* p, size_t per_window) -> int {
                if (p && mm) GNN_HIP(hipMemsetAsync(p, 0xFF, mm * per_window * sizeof(float), guard.main));
                return GNN_OK;
            };
            if ((rc = fill(ws.mp, 2 * NPAIR)) || (rc = fill(ws.m, 2 * NP)) || (rc = fill(ws.yp, (size_t)2 * POOLED * C)) ||
                (rc = fill(ws.logits, 2 * POOLED)) || (rc = fill(ws.alpha, 2 * POOLED)) || (rc = fill(ws.feat, FEAT)))
                return rc;
        }
        if (f32) {
            ProfScope ps(ctx, GNN_K_F32_FRONT);
            if ((rc = launch_front_f32(ctx, b, m))) return rc;
        } else {
            ProfScope ps(ctx, GNN_K_FUSED);
            rc = precision == GNN_PREC_F16X3TK ? launch_front_tk(ctx, b, m)
                 : precision == GNN_PREC_F16X3TC ? launch_front_tc(ctx, b, m)
                 : precision == GNN_PREC_F16C6 ? launch_front_c6(ctx, b, m)
                                               : launch_front_x3(ctx, b, m, precision);       // GNN_PREC_F16X3 / GNN_PREC_BF16X3
            if (rc) return rc;
        }
        if (overlap) {
            GNN_HIP(hipEventRecord(ctx->ev_front[buf], guard.main));
            ctx->stream = ctx->stream2;
            GNN_HIP(hipStreamWaitEvent(ctx->stream2, ctx->ev_front[buf], 0));
        }
        {
            ProfScope ps(ctx, GNN_K_BACKEND);
            // the Toom-Cook front end feeds the back end of the default arithmetic
            rc = launch_backend(ctx, m, (precision == GNN_PREC_F16X3TC || precision == GNN_PREC_F16X3TK) ? GNN_PREC_F16X3 : precision, scores_dev + a * GNN_CLASSES);
        }
        if (overlap) {
            if (!rc) {
                GNN_HIP(hipEventRecord(ctx->ev_back[buf], ctx->stream2));
                ctx->back_pending[buf] = true;
            }
            ctx->stream = guard.main;
            std::swap(ctx->ws, ctx->ws_alt);          // the next chunk (of this or of the next call) takes the other workspace
            ctx->buf_cur ^= 1;
        }
        if (rc) return rc;
    }
    return defer_last ? GNN_OK : flush_backend(ctx);     // the caller synchronises ctx->stream only
}

}  // namespace gnn

using namespace gnn;

extern "C" {

const char* gnn_last_error(void) { return g_last_error.c_str(); }

int gnn_version(void) { return 300; }

int gnn_debug_pack_c6(const float* w, int k, int n, uint32_t* out, size_t out_words, size_t* need_words) {
    std::vector<uint32_t> v;
    const int rc = c6_pack_matrix(w, k, n, v);
    if (rc) {
        set_error("gnn_debug_pack_c6: w is NULL or K is not a multiple of 128 or N not a multiple of 32");
        return rc;
    }
    if (need_words) *need_words = v.size();
    if (out) {
        if (out_words < v.size()) {
            set_error("gnn_debug_pack_c6: output buffer too small");
            return GNN_ERR_ARG;
        }
        std::memcpy(out, v.data(), v.size() * sizeof(uint32_t));
    }
    return GNN_OK;
}

int gnn_debug_set_pad_skip(gnn_ctx* ctx, int on) {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    ctx->c6_pad_skip = on != 0;
    return GNN_OK;
}

int gnn_debug_set_time_split(gnn_ctx* ctx, int on) {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    ctx->time_split = on != 0;
    return GNN_OK;
}

int gnn_debug_last_split(gnn_ctx* ctx, int* workgroups_per_window) {
    if (!ctx || !workgroups_per_window) {
        set_error("bad argument to gnn_debug_last_split");
        return GNN_ERR_ARG;
    }
    *workgroups_per_window = ctx->last_split;
    return GNN_OK;
}

int gnn_fused_rows_per_step(int precision) {
    switch (precision) {
        case GNN_PREC_F32: return 0;
        case GNN_PREC_F16C6: return c6_rows_per_step();
        case GNN_PREC_F16X3TC: case GNN_PREC_F16X3TK: return 96;
        case GNN_PREC_BF16X3: case GNN_PREC_F16X3: return FT;
        default: return GNN_ERR_ARG;
    }
}

// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) of a host buffer — the checksum of the
// TFRecord framing the reference writes its encoded windows with (nn_classification.py:43-52).
// Host-only utility (slicing-by-8); no GPU involved.
uint32_t gnn_crc32c(const void* data, size_t n) {
    struct Table {
        uint32_t t[8][256];
        Table() {
            for (uint32_t i = 0; i < 256; ++i) {
                uint32_t c = i;
                for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u)));
                t[0][i] = c;
            }
            for (uint32_t i = 0; i < 256; ++i)
                for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFF];
        }
    };
    static const Table table;            // thread-safe one-time initialisation (C++11 magic static)
    const uint32_t(*tab)[256] = table.t;
    const unsigned char* p = static_cast<const unsigned char*>(data);
    uint32_t c = 0xFFFFFFFFu;
    while (n >= 8) {
        uint32_t lo, hi;
        std::memcpy(&lo, p, 4);
        std::memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = tab[7][lo & 0xFF] ^ tab[6][(lo >> 8) & 0xFF] ^ tab[5][(lo >> 16) & 0xFF] ^ tab[4][lo >> 24] ^
            tab[3][hi & 0xFF] ^ tab[2][(hi >> 8) & 0xFF] ^ tab[1][(hi >> 16) & 0xFF] ^ tab[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = (c >> 8) ^ tab[0][(c ^ *p++) & 0xFF];
    return c ^ 0xFFFFFFFFu;
}

int gnn_device_count(int* count) {
    if (!count) {
        set_error("count is NULL");
        return GNN_ERR_ARG;
    }
    GNN_HIP(hipGetDeviceCount(count));
    return GNN_OK;
}

int gnn_create(int device, gnn_ctx** out) {
    if (!out) {
        set_error("out is NULL");
        return GNN_ERR_ARG;
    }
    *out = nullptr;
    int count = 0;
    GNN_HIP(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) {
        set_error("device " + std::to_string(device) + " out of range (" + std::to_string(count) + " visible)");
        return GNN_ERR_ARG;
    }
    GNN_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    GNN_HIP(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        set_error(std::string("this library is built for gfx950 only; device is ") + prop.gcnArchName);
        return GNN_ERR_HIP;
    }
    gnn_ctx* ctx = new gnn_ctx();
    ctx->device = device;
    ctx->c6_pad_skip = !debug_switch("GNN_NO_PAD_SKIP");
    ctx->time_split = !debug_switch("GNN_NO_TIME_SPLIT");
    ctx->cu_count = prop.multiProcessorCount;
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_error(std::string("hipStreamCreate failed: ") + hipGetErrorString(e));
        delete ctx;
        return GNN_ERR_HIP;
    }
    *out = ctx;
    return GNN_OK;
}

int gnn_destroy(gnn_ctx* ctx) {
    if (!ctx) return GNN_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);     // back ends an asynchronous call left pending
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm || ctx->comm_scratch) (void)gnn_comm_destroy(ctx);
    free_contig_ws(ctx);
    free_stage(ctx);
    if (ctx->align_buf) (void)hipFree(ctx->align_buf);
    free_kmer_tables(ctx);
    free_ws(ctx->ws);
    free_ws(ctx->ws_alt);
    if (ctx->stream2) {
        for (int i = 0; i < 2; ++i) {
            if (ctx->ev_front[i]) (void)hipEventDestroy(ctx->ev_front[i]);
            if (ctx->ev_back[i]) (void)hipEventDestroy(ctx->ev_back[i]);
        }
        (void)hipStreamDestroy(ctx->stream2);
    }
    for (void* p : ctx->owned) (void)hipFree(p);
    for (auto& s : ctx->prof)
        for (auto& pr : s.pending) {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return GNN_OK;
}

int gnn_sync(gnn_ctx* ctx) {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    GNN_HIP(hipStreamSynchronize(ctx->stream));
    return GNN_OK;
}

int gnn_device_info(gnn_ctx* ctx, char* name, size_t name_len, int* cus, int64_t* hbm_bytes) {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    hipDeviceProp_t prop;
    GNN_HIP(hipGetDeviceProperties(&prop, ctx->device));
    if (name && name_len) {
        std::string s = std::string(prop.name) + " (" + prop.gcnArchName + ")";
        std::strncpy(name, s.c_str(), name_len - 1);
        name[name_len - 1] = 0;
    }
    if (cus) *cus = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return GNN_OK;
}

int gnn_device_pci_bus_id(gnn_ctx* ctx, char* out, size_t out_len) {
    int rc = check_ctx(ctx, false);
    if (rc) return rc;
    if (!out || out_len < 16) {
        set_error("gnn_device_pci_bus_id: out must hold at least 16 bytes");
        return GNN_ERR_ARG;
    }
    GNN_HIP(hipDeviceGetPCIBusId(out, (int)out_len, ctx->device));
    return GNN_OK;
}

int gnn_device_mem_info(gnn_ctx* ctx, int64_t* free_bytes, int64_t* total_bytes) {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    size_t f = 0, t = 0;
    GNN_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return GNN_OK;
}

int gnn_set_chunk(gnn_ctx* ctx, int64_t windows_per_chunk) {
    if (!ctx || windows_per_chunk < 1) {
        set_error("bad chunk");
        return GNN_ERR_ARG;
    }
    ctx->chunk_fused = windows_per_chunk;
    ctx->chunk_explicit = true;
    return GNN_OK;
}

int gnn_load_weights(gnn_ctx* ctx, const gnn_weights* w) {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!w) {
        set_error("weights is NULL");
        return GNN_ERR_ARG;
    }
    const void* req[] = {w->conv1_kernel, w->conv1_bias, w->conv2_kernel, w->conv2_bias, w->conv3_kernel,
                         w->conv3_bias,   w->out_kernel, w->out_bias,     w->enc.kernel,  w->enc.bias,
                         w->enc.gamma,    w->enc.beta,   w->enc.mean,     w->enc.var,     w->head.kernel,
                         w->head.bias,    w->head.gamma, w->head.beta,    w->head.mean,   w->head.var};
    for (const void* p : req)
        if (!p) {
            set_error("a weight pointer is NULL");
            return GNN_ERR_ARG;
        }
    if (ctx->has_weights) {
        set_error("weights already loaded into this ctx (create a new ctx)");
        return GNN_ERR_STATE;
    }
    DeviceWeights& d = ctx->w;
    if ((rc = upload(ctx, w->conv1_kernel, (size_t)KS * GNN_DEPTH * C, &d.conv1_k))) return rc;
    if ((rc = upload(ctx, w->conv1_bias, (size_t)C, &d.conv1_b))) return rc;
    const float* ck[2] = {w->conv2_kernel, w->conv3_kernel};
    const float* cb[2] = {w->conv2_bias, w->conv3_bias};
    for (int i = 0; i < 2; ++i) {
        if ((rc = upload(ctx, ck[i], (size_t)KS * C * C, &d.conv_k[i]))) return rc;
        if ((rc = upload(ctx, cb[i], (size_t)C, &d.conv_b[i]))) return rc;
    }
    const gnn_igloo_weights* ig[2] = {&w->igloo_a, &w->igloo_b};
    for (int h = 0; h < 2; ++h) {
        const gnn_igloo_weights* g = ig[h];
        if (!g->patches || !g->w_mult || !g->w_summer || !g->w_bias || !g->w_qk || !g->w_v) {
            set_error("an IGLOO weight pointer is NULL");
            return GNN_ERR_ARG;
        }
        // sort the (patch, slot) pairs by position (stable: ties keep pair order)
        std::vector<int32_t> order(NPAIR);
        for (int i = 0; i < NPAIR; ++i) {
            const int32_t t = g->patches[i];
            if (t < 0 || t >= T) {
                set_error("patch index out of range [0,5997)");
                return GNN_ERR_WEIGHTS;
            }
            order[i] = i;
        }
        std::stable_sort(order.begin(), order.end(),
                         [&](int32_t a, int32_t b) { return g->patches[a] < g->patches[b]; });
        // W_eff[p,j,c] = w_mult[0,p,j,c] * w_summer[0, j*128+c, 0]  (igloo.py:195-204 folded)
        std::vector<float> weff((size_t)NPAIR * C);
        std::vector<int32_t> pos(NPAIR), slot(NPAIR), ptr(FSTEPS + 1, 0);
        for (int e = 0; e < NPAIR; ++e) {
            const int pair = order[e], j = pair % PS;
            pos[e] = g->patches[pair];
            slot[pair] = e;
            ptr[pos[e] / FT + 1] += 1;
            for (int c = 0; c < C; ++c)
                weff[(size_t)e * C + c] = g->w_mult[(size_t)pair * C + c] * g->w_summer[j * C + c];
        }
        for (int s2 = 0; s2 < FSTEPS; ++s2) ptr[s2 + 1] += ptr[s2];
        if ((rc = upload(ctx, weff.data(), weff.size(), &d.weff_sorted[h]))) return rc;
        if ((rc = upload(ctx, pos.data(), pos.size(), &d.pos_sorted[h]))) return rc;
        if ((rc = upload(ctx, slot.data(), slot.size(), &d.slot[h]))) return rc;
        if ((rc = upload(ctx, ptr.data(), ptr.size(), &d.bucket_ptr[h]))) return rc;
        if ((rc = upload(ctx, g->w_bias, (size_t)NP, &d.w_bias[h]))) return rc;
        if ((rc = upload(ctx, g->w_qk, (size_t)NP * POOLED, &d.w_qk[h]))) return rc;
        if ((rc = upload(ctx, g->w_v, (size_t)C * C, &d.w_v[h]))) return rc;
    }
    {
        std::vector<float> pt;
        build_conv1_pair_tables(w->conv1_kernel, pt);
        if ((rc = upload(ctx, pt.data(), pt.size(), &d.conv1_pairs))) return rc;
    }
    // Fold BatchNormalization (inference statistics) into the preceding Dense (model.py:28-30, 40-42):
    // y = gamma * (x@K + b - mean) / sqrt(var + eps) + beta = x@(K*s) + ((b - mean)*s + beta)
    auto fold = [&](const gnn_dense_bn& l, int in, float** dk, float** db, uint16_t** dfrag) -> int {
        std::vector<float> k((size_t)in * HID), b(HID);
        for (int o = 0; o < HID; ++o) {
            const float s = l.gamma[o] / std::sqrt(l.var[o] + BN_EPS);
            b[o] = (l.bias[o] - l.mean[o]) * s + l.beta[o];
            for (int i = 0; i < in; ++i) k[(size_t)i * HID + o] = l.kernel[(size_t)i * HID + o] * s;
        }
        int r = upload(ctx, k.data(), k.size(), dk);
        if (r) return r;
        if ((r = upload(ctx, b.data(), b.size(), db))) return r;
        const std::vector<uint16_t> fr = pack_frags(k.data(), in, HID, true);   // the same folded kernel, f16 hi / lo limbs, for dense_mfma_kernel
        void* p = nullptr;
        GNN_HIP(hipMalloc(&p, fr.size() * sizeof(uint16_t)));
        ctx->owned.push_back(p);
        GNN_HIP(hipMemcpy(p, fr.data(), fr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        *dfrag = static_cast<uint16_t*>(p);
        return GNN_OK;
    };
    if ((rc = fold(w->enc, FEAT, &d.d1_k, &d.d1_b, &d.d1_frag))) return rc;
    if ((rc = fold(w->head, HID, &d.d2_k, &d.d2_b, &d.d2_frag))) return rc;
    if ((rc = upload(ctx, w->out_kernel, (size_t)HID * GNN_CLASSES, &d.d3_k))) return rc;
    if ((rc = upload(ctx, w->out_bias, (size_t)GNN_CLASSES, &d.d3_b))) return rc;
    if ((rc = pack_fused_weights(ctx, w))) return rc;
    if ((rc = pack_fused_c6_weights(ctx, w))) return rc;
    if ((rc = pack_fused_x3_consts(ctx))) return rc;
    if ((rc = pack_fused_tc_weights(ctx, w))) return rc;
    ctx->has_weights = true;
    return GNN_OK;
}

// k-mer tables of GNN_PREC_F16X3TK (gnn_fused_tk.hip).  reserve_bytes < 0: the default reserve = two workspaces of the ctx's launch
// size (the asynchronous entry point alternates two) + 8 GiB for the caller's own buffers.
int gnn_build_kmer_tables(gnn_ctx* ctx, int64_t reserve_bytes) {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!ctx->has_weights) {
        set_error("gnn_load_weights has not been called");
        return GNN_ERR_STATE;
    }
    if ((rc = finish_pending(ctx))) return rc;
    const size_t reserve = reserve_bytes >= 0 ? (size_t)reserve_bytes : 2 * (size_t)ctx->chunk_fused * WS_BYTES_PER_WINDOW + ((size_t)8 << 30);
    return build_kmer_tables(ctx, reserve);
}

int gnn_has_kmer_tables(gnn_ctx* ctx) {
    return ctx && ctx->w.tk_x2_tbl && ctx->w.tk_mpa_tbl && ctx->w.tk_pt_tbl && ctx->w.tk_x1t_tbl && ctx->w.tk_yp_const ? 1 : 0;
}

int gnn_drop_kmer_tables(gnn_ctx* ctx) {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if ((rc = finish_pending(ctx))) return rc;
    GNN_HIP(hipStreamSynchronize(ctx->stream));
    free_kmer_tables(ctx);
    return GNN_OK;
}

int64_t gnn_kmer_tables_bytes(void) { return (int64_t)kmer_tables_bytes(); }

// test aid: row `row` of the 14-mer table (128 floats) / entry (e = row >> 32, 9-mer = row & 0xffffffff) of head A's table (1 float)
int gnn_debug_kmer_table_row(gnn_ctx* ctx, int which, uint64_t row, float* out_host) {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!gnn_has_kmer_tables(ctx) || !out_host) {
        set_error("gnn_debug_kmer_table_row: no tables / NULL output");
        return GNN_ERR_STATE;
    }
    if (which == 0) {
        if (row > ((uint64_t)1 << 28)) { set_error("14-mer row out of range"); return GNN_ERR_ARG; }
        GNN_HIP(hipMemcpy(out_host, ctx->w.tk_x2_tbl + (size_t)row * C, C * sizeof(float), hipMemcpyDeviceToHost));
    } else {
        const uint64_t e = row >> 32, n = row & 0xffffffffu;
        if (e >= (uint64_t)NPAIR || n > ((uint64_t)1 << 18)) { set_error("(entry, 9-mer) out of range"); return GNN_ERR_ARG; }
        GNN_HIP(hipMemcpy(out_host, ctx->w.tk_mpa_tbl + (size_t)e * (((size_t)1 << 18) + 1) + n, sizeof(float), hipMemcpyDeviceToHost));
    }
    return GNN_OK;
}

int gnn_dev_alloc(gnn_ctx* ctx, size_t bytes, void** dev_ptr) {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!dev_ptr) {
        set_error("dev_ptr is NULL");
        return GNN_ERR_ARG;
    }
    return dev_buffer(ctx, bytes ? bytes : 1, dev_ptr);
}

int gnn_dev_free(gnn_ctx* ctx, void* dev_ptr) {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    GNN_HIP(hipStreamSynchronize(ctx->stream));
    if (dev_ptr) GNN_HIP(hipFree(dev_ptr));
    return GNN_OK;
}

int gnn_memcpy_h2d(gnn_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    GNN_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    GNN_HIP(hipStreamSynchronize(ctx->stream));
    return GNN_OK;
}

int gnn_memcpy_d2h(gnn_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    GNN_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    GNN_HIP(hipStreamSynchronize(ctx->stream));
    return GNN_OK;
}

int gnn_tokenize_dev(gnn_ctx* ctx, const uint8_t* bases_dev, int64_t n, uint16_t* tokens_dev) {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!bases_dev || !tokens_dev))) {
        set_error("bad argument to gnn_tokenize_dev");
        return GNN_ERR_ARG;
    }
    if (n == 0) return GNN_OK;
    return launch_tokenize(ctx, bases_dev, n, tokens_dev);
}

int gnn_tokenize(gnn_ctx* ctx, const uint8_t* bases_host, int64_t n, uint16_t* tokens_host) {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!bases_host || !tokens_host))) {
        set_error("bad argument to gnn_tokenize");
        return GNN_ERR_ARG;
    }
    if (n == 0) return GNN_OK;
    void *b = nullptr, *t = nullptr;
    if ((rc = dev_buffer(ctx, (size_t)n * W, &b))) return rc;
    if ((rc = dev_buffer(ctx, (size_t)n * T * sizeof(uint16_t), &t))) {
        (void)hipFree(b);
        return rc;
    }
    rc = gnn_memcpy_h2d(ctx, b, bases_host, (size_t)n * W);
    if (!rc) rc = launch_tokenize(ctx, (const uint8_t*)b, n, (uint16_t*)t);
    if (!rc) rc = gnn_memcpy_d2h(ctx, tokens_host, t, (size_t)n * T * sizeof(uint16_t));
    (void)hipFree(b);
    (void)hipFree(t);
    return rc;
}

int gnn_onehot_dev(gnn_ctx* ctx, const uint8_t* bases_dev, int64_t n, int dtype, void* out) {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!bases_dev || !out)) || dtype < GNN_OH_U8 || dtype > GNN_OH_F32) {
        set_error("bad argument to gnn_onehot_dev");
        return GNN_ERR_ARG;
    }
    if (n == 0) return GNN_OK;
    ProfScope ps(ctx, GNN_K_ENCODER);
    return launch_onehot(ctx, bases_dev, n, dtype, out);
}

int gnn_synth_windows_dev(gnn_ctx* ctx, uint64_t seed, int64_t first, int64_t n, uint8_t* bases_dev) {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (n < 0 || first < 0 || (n > 0 && !bases_dev)) {
        set_error("bad argument to gnn_synth_windows_dev");
        return GNN_ERR_ARG;
    }
    if (n == 0) return GNN_OK;
    return launch_synth(ctx, seed, first, n, bases_dev);
}

int gnn_classify_dev(gnn_ctx* ctx, const uint8_t* bases_dev, int64_t n, int precision, float* scores_dev) {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!bases_dev || !scores_dev))) {
        set_error("bad argument to gnn_classify_dev");
        return GNN_ERR_ARG;
    }
    if (n == 0) return GNN_OK;
    return classify_chunks(ctx, bases_dev, n, precision, scores_dev);
}

int gnn_classify_dev_async(gnn_ctx* ctx, const uint8_t* bases_dev, int64_t n, int precision, float* scores_dev) {
    int rc = check_ctx(ctx, false);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!bases_dev || !scores_dev))) {
        set_error("bad argument to gnn_classify_dev_async");
        return GNN_ERR_ARG;
    }
    if (n == 0) return GNN_OK;
    return classify_chunks(ctx, bases_dev, n, precision, scores_dev, true);
}

int gnn_classify_flush(gnn_ctx* ctx) { return check_ctx(ctx); }

int gnn_classify(gnn_ctx* ctx, const uint8_t* bases_host, int64_t n, int precision, float* scores_host) {
    return gnn_debug_forward(ctx, bases_host, n, precision, scores_host, nullptr);
}

int gnn_debug_forward(gnn_ctx* ctx, const uint8_t* bases_host, int64_t n, int precision,
                      float* scores_host, const gnn_taps* taps) {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!bases_host || !scores_host))) {
        set_error("bad argument to gnn_classify");
        return GNN_ERR_ARG;
    }
    if (n == 0) return GNN_OK;
    if (taps) {
        // taps are copied out of the workspace, so the whole batch must be one chunk
        const int64_t lim = precision == GNN_PREC_F32 ? ctx->chunk_f32 : ctx->chunk_fused;
        if (n > lim) {
            set_error("gnn_debug_forward with taps: n_windows exceeds one chunk (" + std::to_string(lim) + ")");
            return GNN_ERR_ARG;
        }
        if ((taps->x1 || taps->x2 || taps->x3) && precision != GNN_PREC_F32) {
            set_error("x1/x2/x3 taps exist only on the f32 path (the fused path keeps them in LDS)");
            return GNN_ERR_ARG;
        }
    }
    // windows go up and scores come back through the ctx's persistent staging, a slab of at most STAGE_MAX_WINDOWS at a time
    // (the reference calls predict() once per batch of 128 windows, nn_classification.py:316-317: nothing is allocated here)
    const int64_t slab = std::min<int64_t>(n, STAGE_MAX_WINDOWS);
    if ((rc = ensure_stage(ctx, slab))) return rc;
    for (int64_t a0 = 0; a0 < n && !rc; a0 += slab) {
        const int64_t m = std::min(slab, n - a0);
        rc = stage_upload(ctx, bases_host + a0 * W, (size_t)m * W);
        if (!rc) rc = classify_chunks(ctx, ctx->stage_bases, m, precision, ctx->stage_scores);
        if (!rc) {
            GNN_HIP(hipMemcpyAsync(ctx->stage_scores_host, ctx->stage_scores, (size_t)m * GNN_CLASSES * sizeof(float),
                                   hipMemcpyDeviceToHost, ctx->stream));
            GNN_HIP(hipStreamSynchronize(ctx->stream));
            std::memcpy(scores_host + a0 * GNN_CLASSES, ctx->stage_scores_host, (size_t)m * GNN_CLASSES * sizeof(float));
        }
    }
    if (!rc && taps && precision != GNN_PREC_F32 && n > ctx->chunk_fused) {
        // the launch size was lowered inside this call (free-memory clamp or halve-and-retry): the workspace holds the last launch only
        set_error("gnn_debug_forward with taps: the launch size was reduced to " + std::to_string(ctx->chunk_fused) +
                  " windows for lack of device memory; the taps of " + std::to_string(n) + " windows do not fit one launch");
        return GNN_ERR_STATE;
    }
    if (!rc && taps) {
        const Workspace& ws = ctx->ws;
        auto out = [&](float* dst, const float* src, size_t count) {
            if (dst && !rc) rc = gnn_memcpy_d2h(ctx, dst, src, count * sizeof(float));
        };
        out(taps->x1, ws.x[0], (size_t)n * T * C);
        out(taps->x2, ws.x[1], (size_t)n * T * C);
        out(taps->x3, ws.x[2], (size_t)n * T * C);
        out(taps->feat, ws.feat, (size_t)n * FEAT);
        // per-head strided tensors: copy the (n,2,...) buffers and de-interleave on the host
        auto out2 = [&](float* dst_a, float* dst_b, const float* src, size_t per) {
            if ((!dst_a && !dst_b) || rc) return;
            std::vector<float> tmp((size_t)n * 2 * per);
            rc = gnn_memcpy_d2h(ctx, tmp.data(), src, tmp.size() * sizeof(float));
            if (rc) return;
            for (int64_t i = 0; i < n; ++i) {
                if (dst_a) std::memcpy(dst_a + i * per, tmp.data() + (i * 2 + 0) * per, per * sizeof(float));
                if (dst_b) std::memcpy(dst_b + i * per, tmp.data() + (i * 2 + 1) * per, per * sizeof(float));
            }
        };
        out2(taps->yp_a, taps->yp_b, ws.yp, (size_t)POOLED * C);
        out2(taps->alpha_a, taps->alpha_b, ws.alpha, (size_t)POOLED);
        out2(taps->m_a, taps->m_b, ws.m, (size_t)NP);
    }
    return rc;
}

// Copy the span table to the device; checks every span against the rules of the caller's contract.
static int upload_spans(gnn_ctx* ctx, const int64_t* starts_host, const int32_t* lens_host, int64_t n,
                        int64_t** starts_dev, int32_t** lens_dev) {
    for (int64_t i = 0; i < n; ++i)
        if (starts_host[i] < 0 || lens_host[i] < 0 || lens_host[i] > W) {
            set_error("span " + std::to_string(i) + " has a negative start or a length outside [0, 6000]");
            return GNN_ERR_ARG;
        }
    void *ps = nullptr, *pl = nullptr;
    int rc = dev_buffer(ctx, (size_t)n * sizeof(int64_t), &ps);
    if (rc) return rc;
    if ((rc = dev_buffer(ctx, (size_t)n * sizeof(int32_t), &pl))) {
        (void)hipFree(ps);
        return rc;
    }
    rc = gnn_memcpy_h2d(ctx, ps, starts_host, (size_t)n * sizeof(int64_t));
    if (!rc) rc = gnn_memcpy_h2d(ctx, pl, lens_host, (size_t)n * sizeof(int32_t));
    if (rc) {
        (void)hipFree(ps);
        (void)hipFree(pl);
        return rc;
    }
    *starts_dev = static_cast<int64_t*>(ps);
    *lens_dev = static_cast<int32_t*>(pl);
    return GNN_OK;
}

int gnn_span_byte_count(gnn_ctx* ctx, const uint8_t* seq_dev, const int64_t* starts_host, const int32_t* lens_host,
                        int64_t n, int byte, int32_t* counts_host) {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (n < 0 || byte < 0 || byte > 255 || (n > 0 && (!seq_dev || !starts_host || !lens_host || !counts_host))) {
        set_error("bad argument to gnn_span_byte_count");
        return GNN_ERR_ARG;
    }
    if (n == 0) return GNN_OK;
    int64_t* ds = nullptr;
    int32_t* dl = nullptr;
    if ((rc = upload_spans(ctx, starts_host, lens_host, n, &ds, &dl))) return rc;
    void* dc = nullptr;
    rc = dev_buffer(ctx, (size_t)n * sizeof(int32_t), &dc);
    if (!rc) rc = launch_span_count(ctx, seq_dev, ds, dl, n, byte, (int32_t*)dc);
    if (!rc) rc = gnn_memcpy_d2h(ctx, counts_host, dc, (size_t)n * sizeof(int32_t));
    (void)hipFree(ds);
    (void)hipFree(dl);
    if (dc) (void)hipFree(dc);
    return rc;
}

int gnn_classify_spans(gnn_ctx* ctx, const uint8_t* seq_dev, const int64_t* starts_host, const int32_t* lens_host,
                       int64_t n, int precision, float* scores_host) {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!seq_dev || !starts_host || !lens_host || !scores_host))) {
        set_error("bad argument to gnn_classify_spans");
        return GNN_ERR_ARG;
    }
    if (n == 0) return GNN_OK;
    int64_t* ds = nullptr;
    int32_t* dl = nullptr;
    if ((rc = upload_spans(ctx, starts_host, lens_host, n, &ds, &dl))) return rc;
    // windows are materialised one slab at a time (6 KB each), never the whole table
    const int64_t slab = std::min<int64_t>(n, 4 * std::max<int64_t>(ctx->chunk_fused, 1));
    void *db = nullptr, *dsc = nullptr;
    rc = dev_buffer(ctx, (size_t)slab * W, &db);
    if (!rc) rc = dev_buffer(ctx, (size_t)n * GNN_CLASSES * sizeof(float), &dsc);
    for (int64_t a = 0; a < n && !rc; a += slab) {
        const int64_t m = std::min(slab, n - a);
        rc = launch_materialize(ctx, seq_dev, ds + a, dl + a, m, (uint8_t*)db);
        if (!rc) rc = classify_chunks(ctx, (const uint8_t*)db, m, precision, (float*)dsc + a * GNN_CLASSES);
    }
    if (!rc) rc = gnn_memcpy_d2h(ctx, scores_host, dsc, (size_t)n * GNN_CLASSES * sizeof(float));
    (void)hipFree(ds);
    (void)hipFree(dl);
    if (db) (void)hipFree(db);
    if (dsc) (void)hipFree(dsc);
    return rc;
}

int gnn_profile_enable(gnn_ctx* ctx, int on) {
    if (!ctx) {
        set_error("ctx is NULL");
        return GNN_ERR_ARG;
    }
    ctx->profile = on != 0;
    return GNN_OK;
}

static int drain_profile(gnn_ctx* ctx) {
    GNN_HIP(hipStreamSynchronize(ctx->stream));
    for (auto& s : ctx->prof) {
        for (auto& pr : s.pending) {
            float ms = 0.f;
            GNN_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
            s.total_ms += ms;
            s.launches += 1;
            ctx->event_pool.push_back(pr.first);
            ctx->event_pool.push_back(pr.second);
        }
        s.pending.clear();
    }
    return GNN_OK;
}

// Debug aid (not part of the hot path): run the fused kernel's instrumented build and return the
// per-phase cycle sums of wave 0 of every workgroup.  on=1 allocates/zeroes, on=0 frees.
int gnn_phase_cycles(gnn_ctx* ctx, int on, unsigned long long* out16) {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    GNN_HIP(hipStreamSynchronize(ctx->stream));
    if (out16 && ctx->phase_cycles)
        GNN_HIP(hipMemcpy(out16, ctx->phase_cycles, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (on) {
        if (!ctx->phase_cycles) GNN_HIP(hipMalloc((void**)&ctx->phase_cycles, 16 * sizeof(unsigned long long)));
        GNN_HIP(hipMemset(ctx->phase_cycles, 0, 16 * sizeof(unsigned long long)));
    } else if (ctx->phase_cycles) {
        (void)hipFree(ctx->phase_cycles);
        ctx->phase_cycles = nullptr;
    }
    return GNN_OK;
}

int gnn_profile_reset(gnn_ctx* ctx) {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if ((rc = drain_profile(ctx))) return rc;
    for (auto& s : ctx->prof) {
        s.total_ms = 0.0;
        s.launches = 0;
    }
    return GNN_OK;
}

int gnn_profile_get(gnn_ctx* ctx, int kernel_id, double* total_ms, int64_t* launches) {
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (kernel_id < 0 || kernel_id >= GNN_K_COUNT) {
        set_error("bad kernel id");
        return GNN_ERR_ARG;
    }
    if ((rc = drain_profile(ctx))) return rc;
    if (total_ms) *total_ms = ctx->prof[kernel_id].total_ms;
    if (launches) *launches = ctx->prof[kernel_id].launches;
    return GNN_OK;
}

}  // extern "C"
