// Contig front end as ONE entry point: packed raw contig bytes + offsets -> per-contig
// class scores.  Replaces, for a whole packed buffer, generate_data (window cutting, the N-content rule,
// upper-casing, padding, tokenising: nn_classification.py:54-82), the predict loop and
// tf.math.segment_mean (:316-320) without a per-window host object and without the window scores ever
// leaving the device.
//
// Everything after the span table is asynchronous: the sequence bytes (when they start on the host) go up
// in pieces on a copy stream while earlier pieces are classified on the ctx stream; the N-content rule is
// evaluated on the device and applied as a MASK of the segment mean — every candidate window is
// classified, windows the rule drops (window_n > 0 and more than 4000 literal 'N') just do not enter their
// contig's mean, so no host round trip sits between the rule and the classification.  (Dropped windows are
// rare: they cost one window's work each.)  All buffers are persistent and grow-only (gnn_destroy frees them).
#include <algorithm>
#include <cstring>

#include "gnn_common.h"

namespace gnn {

constexpr int MIN_TAIL = 2500;        // nn_classification.py:68  seq_windows(seq, 6000, 2500, ...)
constexpr int MAX_N = 4000;           // nn_classification.py:70  window_n > 0 and count("N") > 4000 -> skip
constexpr int64_t PIECE = 64ll << 20; // bytes per host->device piece of the sequence buffer

struct ContigWorkspace {
    hipStream_t copy_stream = nullptr;
    std::vector<hipEvent_t> piece_done;
    // host side of the span table
    std::vector<int64_t> starts, ids;
    std::vector<int32_t> lens, window_n, counts;
    // device side (capacities in elements / bytes)
    uint8_t* seq = nullptr;
    size_t seq_cap = 0;
    int64_t* d_starts = nullptr;
    int64_t* d_ids = nullptr;
    int32_t* d_lens = nullptr;
    int32_t* d_window_n = nullptr;
    int32_t* d_counts = nullptr;
    float* d_scores = nullptr;
    size_t span_cap = 0;
    uint8_t* d_bases = nullptr;
    size_t bases_cap = 0;
    float* d_out = nullptr;
    size_t out_cap = 0;
};

template <typename Tp>
static int grow(Tp*& p, size_t& cap, size_t need, size_t elem = sizeof(Tp)) {
    if (cap >= need && p) return GNN_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    void* q = nullptr;
    const size_t want = need + need / 4 + 64;          // some head-room: grow-only, amortised
    hipError_t e = hipMalloc(&q, want * elem);
    if (e != hipSuccess) {
        set_error("hipMalloc of " + std::to_string(want * elem) + " bytes failed: " + hipGetErrorString(e));
        return GNN_ERR_NOMEM;
    }
    p = static_cast<Tp*>(q);
    cap = want;
    return GNN_OK;
}

void free_contig_ws(gnn_ctx* ctx) {
    ContigWorkspace* w = ctx->contig_ws;
    if (!w) return;
    for (hipEvent_t e : w->piece_done) (void)hipEventDestroy(e);
    if (w->copy_stream) (void)hipStreamDestroy(w->copy_stream);
    for (void* p : {(void*)w->seq, (void*)w->d_starts, (void*)w->d_ids, (void*)w->d_lens, (void*)w->d_window_n,
                    (void*)w->d_counts, (void*)w->d_scores, (void*)w->d_bases, (void*)w->d_out})
        if (p) (void)hipFree(p);
    delete w;
    ctx->contig_ws = nullptr;
}

// tf.math.segment_mean over the windows the N-content rule keeps (nn_classification.py:70-71, :320): one thread
// per (contig, class); ids are sorted, the contig's windows are a contiguous run summed in window order —
// the same order and arithmetic as segment_mean_kernel, so the two agree bit for bit on the kept windows.
__global__ void masked_segment_mean_kernel(const float* __restrict__ scores, const int64_t* __restrict__ ids,
                                           const int32_t* __restrict__ window_n, const int32_t* __restrict__ counts,
                                           int64_t n, int64_t n_seg, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_seg * GNN_CLASSES) return;
    const int64_t seg = i / GNN_CLASSES;
    const int cl = (int)(i % GNN_CLASSES);
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (ids[mid] < seg) lo = mid + 1; else hi = mid;
    }
    float s = 0.f;
    int kept = 0;
    for (int64_t k = lo; k < n && ids[k] == seg; ++k)
        if (window_n[k] == 0 || counts[k] <= MAX_N) {
            s += scores[k * GNN_CLASSES + cl];
            ++kept;
        }
    out[i] = kept ? s / (float)kept : 0.f;
}

}  // namespace gnn

using namespace gnn;

extern "C" int gnn_classify_contigs(gnn_ctx* ctx, const uint8_t* seq, int seq_on_host, int64_t seq_bytes,
                                    const int64_t* offsets_host, int64_t n_contigs, int single_window, int precision,
                                    float* contig_scores_host, int64_t* window_ids_host, int64_t ids_capacity,
                                    int64_t* n_windows_out) {
    if (!ctx) {
        set_error("ctx is NULL");
        return GNN_ERR_ARG;
    }
    GNN_HIP(hipSetDevice(ctx->device));
    {
        const int frc = finish_pending(ctx);
        if (frc) return frc;
    }
    if (n_contigs < 0 || seq_bytes < 0 || !offsets_host || !n_windows_out || (n_contigs > 0 && !contig_scores_host) ||
        (seq_bytes > 0 && !seq)) {
        set_error("bad argument to gnn_classify_contigs");
        return GNN_ERR_ARG;
    }
    if (offsets_host[0] < 0 || offsets_host[n_contigs] > seq_bytes) {
        set_error("contig offsets outside the sequence buffer");
        return GNN_ERR_ARG;
    }
    if (!ctx->contig_ws) ctx->contig_ws = new ContigWorkspace();
    ContigWorkspace& w = *ctx->contig_ws;

    // ---- candidate windows: seq_windows(seq, 6000, 2500, max_windows) for every contig (sequence.py:150-167)
    w.starts.clear(), w.lens.clear(), w.ids.clear(), w.window_n.clear();
    for (int64_t c = 0; c < n_contigs; ++c) {
        const int64_t a = offsets_host[c], b = offsets_host[c + 1];
        if (b < a) {
            set_error("contig offsets are not non-decreasing");
            return GNN_ERR_ARG;
        }
        const int64_t len = b - a;
        for (int64_t k = 0; k * W < len; ++k) {
            const int64_t l = std::min<int64_t>(W, len - k * W);
            if (l < MIN_TAIL && k > 0) break;             // a short tail is dropped, a short first window kept
            w.starts.push_back(a + k * W);
            w.lens.push_back((int32_t)l);
            w.ids.push_back(c);
            w.window_n.push_back((int32_t)k);
            if (l < MIN_TAIL || (single_window && k == 0)) break;
        }
    }
    const int64_t n = (int64_t)w.starts.size();
    *n_windows_out = 0;
    if (n_contigs) std::memset(contig_scores_host, 0, (size_t)n_contigs * GNN_CLASSES * sizeof(float));
    if (n == 0) return GNN_OK;
    if (!window_ids_host || ids_capacity < n) {
        set_error("window_ids_host holds " + std::to_string(ids_capacity) + " entries, " + std::to_string(n) + " candidate windows");
        return GNN_ERR_ARG;
    }

    // ---- device buffers
    int rc = GNN_OK;
    size_t cap = w.span_cap;
    if (cap < (size_t)n) {
        GNN_HIP(hipStreamSynchronize(ctx->stream));
        size_t c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0;
        if (!rc) rc = grow(w.d_starts, c1, (size_t)n);
        if (!rc) rc = grow(w.d_ids, c2, (size_t)n);
        if (!rc) rc = grow(w.d_lens, c3, (size_t)n);
        if (!rc) rc = grow(w.d_window_n, c4, (size_t)n);
        if (!rc) rc = grow(w.d_counts, c5, (size_t)n);
        if (!rc) rc = grow(w.d_scores, c6, (size_t)n * GNN_CLASSES);
        w.span_cap = rc ? 0 : c1;
        if (rc) return rc;
    }
    const int64_t slab = std::min<int64_t>(n, 4 * std::max<int64_t>(ctx->chunk_fused, 1));
    if ((rc = grow(w.d_bases, w.bases_cap, (size_t)slab * W))) return rc;
    if ((rc = grow(w.d_out, w.out_cap, (size_t)n_contigs * GNN_CLASSES))) return rc;
    const uint8_t* seq_dev = seq;
    int64_t n_pieces = 0;
    if (seq_on_host) {
        if ((rc = grow(w.seq, w.seq_cap, (size_t)seq_bytes))) return rc;
        seq_dev = w.seq;
        if (!w.copy_stream) GNN_HIP(hipStreamCreateWithFlags(&w.copy_stream, hipStreamNonBlocking));
        n_pieces = (seq_bytes + PIECE - 1) / PIECE;
        while ((int64_t)w.piece_done.size() < n_pieces) {
            hipEvent_t e = nullptr;
            GNN_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            w.piece_done.push_back(e);
        }
        // the previous call's kernels may still read w.seq: order the copies behind them
        hipEvent_t& first = w.piece_done[0];
        GNN_HIP(hipEventRecord(first, ctx->stream));
        GNN_HIP(hipStreamWaitEvent(w.copy_stream, first, 0));
    }
    GNN_HIP(hipMemcpyAsync(w.d_starts, w.starts.data(), (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    GNN_HIP(hipMemcpyAsync(w.d_lens, w.lens.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    GNN_HIP(hipMemcpyAsync(w.d_ids, w.ids.data(), (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    GNN_HIP(hipMemcpyAsync(w.d_window_n, w.window_n.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));

    // ---- slabs of windows: upload what they read (copy stream), count N, materialise, classify (ctx stream)
    int64_t uploaded = 0;       // pieces issued so far
    for (int64_t a = 0; a < n; a += slab) {
        const int64_t m = std::min(slab, n - a);
        if (seq_on_host) {
            const int64_t need = w.starts[a + m - 1] + w.lens[a + m - 1];       // spans are in buffer order
            const int64_t upto = std::min<int64_t>(n_pieces, (need + PIECE - 1) / PIECE);
            for (; uploaded < upto; ++uploaded) {
                const int64_t off = uploaded * PIECE, len = std::min(PIECE, seq_bytes - off);
                GNN_HIP(hipMemcpyAsync(w.seq + off, seq + off, (size_t)len, hipMemcpyHostToDevice, w.copy_stream));
                GNN_HIP(hipEventRecord(w.piece_done[uploaded], w.copy_stream));
            }
            if (upto > 0) GNN_HIP(hipStreamWaitEvent(ctx->stream, w.piece_done[upto - 1], 0));
        }
        if ((rc = launch_span_count(ctx, seq_dev, w.d_starts + a, w.d_lens + a, m, 'N', w.d_counts + a))) return rc;
        if ((rc = launch_materialize(ctx, seq_dev, w.d_starts + a, w.d_lens + a, m, w.d_bases))) return rc;
        if ((rc = classify_chunks(ctx, w.d_bases, m, precision, w.d_scores + a * GNN_CLASSES))) return rc;
    }
    const int64_t threads = n_contigs * GNN_CLASSES;
    hipLaunchKernelGGL(masked_segment_mean_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream,
                       w.d_scores, w.d_ids, w.d_window_n, w.d_counts, n, n_contigs, w.d_out);
    GNN_HIP(hipGetLastError());
    w.counts.resize((size_t)n);
    GNN_HIP(hipMemcpyAsync(contig_scores_host, w.d_out, (size_t)threads * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    GNN_HIP(hipMemcpyAsync(w.counts.data(), w.d_counts, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    GNN_HIP(hipStreamSynchronize(ctx->stream));
    int64_t kept = 0;
    for (int64_t i = 0; i < n; ++i)
        if (w.window_n[i] == 0 || w.counts[i] <= MAX_N) window_ids_host[kept++] = w.ids[i];
    *n_windows_out = kept;
    return GNN_OK;
}
