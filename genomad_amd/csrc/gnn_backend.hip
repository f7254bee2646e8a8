// Back end shared by both front ends: from the per-window IGLOO partials
//   mp (n,2,8400)   pair dot products        yp (n,2,749,128)  max-pooled y @ w_v
// to class scores.  Everything here is f32 (the softmax over 749 pooled positions is driven by a
// 2100-term dot product and is the precision-sensitive part of the network).
//
//   m[p]      = w_bias[p] + sum_j mp[slot[p*4+j]]                igloo.py:204-206  (m_kernel)
//   logits[q] = sum_p m[p] * w_qk[p,q] ; alpha = softmax(logits)  igloo.py:211-212
//   feat      = sum_q alpha[q] * yp[q,:]   (both heads, concat)   igloo.py:213-214, :83
//   h1 = relu(BN(feat@D1+d1)); h2 = relu(BN(h1@D2+d2)); scores = softmax(h2@D3+d3)   model.py:28-44
//   contig score = segment mean of window scores                  nn_classification.py:320
#include "gnn_common.h"

namespace gnn {

// m[w][h][p] = w_bias[p] + sum_j mp[w][h][slot[p*4+j]]: one block per (window, head) stages the
// window's 8400 pair products (written in position order by the front end) in LDS and gathers
// the four slots of every patch from there, so global traffic stays coalesced.
__global__ __launch_bounds__(256) void m_kernel(const float* __restrict__ mp, const float* __restrict__ w_bias0,
                                                const float* __restrict__ w_bias1,
                                                const int32_t* __restrict__ slot0,
                                                const int32_t* __restrict__ slot1, float* __restrict__ m) {
    __shared__ float s[NPAIR];
    const int wi = blockIdx.x, h = blockIdx.y;
    const float* w_bias = h ? w_bias1 : w_bias0;
    const int32_t* slot = h ? slot1 : slot0;
    const float4* src = reinterpret_cast<const float4*>(mp + ((size_t)wi * 2 + h) * NPAIR);
    for (int i = threadIdx.x; i < NPAIR / 4; i += 256) reinterpret_cast<float4*>(s)[i] = src[i];
    __syncthreads();
    for (int p = threadIdx.x; p < NP; p += 256) {
        const int4 sl = reinterpret_cast<const int4*>(slot)[p];
        m[((size_t)wi * 2 + h) * NP + p] = w_bias[p] + s[sl.x] + s[sl.y] + s[sl.z] + s[sl.w];
    }
}

// logits[w][h][q] = sum_p m[w][h][p] * w_qk[h][p][q].  Block tile: 32 windows x 64 q, K chunk 32.
// Each window's sum runs over p in ascending order independent of the batch it is in, so results
// do not depend on how windows are sharded.
constexpr int LW = 32, LQ = 64, LK = 32;

__global__ __launch_bounds__(256) void logits_kernel(const float* __restrict__ m,
                                                     const float* __restrict__ w_qk0,
                                                     const float* __restrict__ w_qk1, int n,
                                                     float* __restrict__ logits) {
    __shared__ float ms[LK][LW + 1];
    __shared__ float qs[LK][LQ];
    const int h = blockIdx.z;
    const float* w_qk = h ? w_qk1 : w_qk0;
    const int w0 = blockIdx.y * LW;
    const int q0 = blockIdx.x * LQ;
    const int tw = threadIdx.x >> 4;   // 16 groups x 2 windows
    const int tq = threadIdx.x & 15;   // 16 groups x 4 q
    float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int p0 = 0; p0 < NP; p0 += LK) {
        // m tile: 32 p x 32 windows (thread -> one (window, p))
        for (int i = threadIdx.x; i < LK * LW; i += 256) {
            const int p = i % LK, w = i / LK;
            float v = 0.f;
            if (p0 + p < NP && w0 + w < n) v = m[((size_t)(w0 + w) * 2 + h) * NP + p0 + p];
            ms[p][w] = v;
        }
        for (int i = threadIdx.x; i < LK * LQ; i += 256) {
            const int q = i % LQ, p = i / LQ;
            qs[p][q] = (p0 + p < NP && q0 + q < POOLED) ? w_qk[(size_t)(p0 + p) * POOLED + q0 + q] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int p = 0; p < LK; ++p) {
            const float a0 = ms[p][tw * 2], a1 = ms[p][tw * 2 + 1];
            const float4 b = *reinterpret_cast<const float4*>(&qs[p][tq * 4]);
            acc[0][0] = fmaf(a0, b.x, acc[0][0]);
            acc[0][1] = fmaf(a0, b.y, acc[0][1]);
            acc[0][2] = fmaf(a0, b.z, acc[0][2]);
            acc[0][3] = fmaf(a0, b.w, acc[0][3]);
            acc[1][0] = fmaf(a1, b.x, acc[1][0]);
            acc[1][1] = fmaf(a1, b.y, acc[1][1]);
            acc[1][2] = fmaf(a1, b.z, acc[1][2]);
            acc[1][3] = fmaf(a1, b.w, acc[1][3]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int w = w0 + tw * 2 + i;
        if (w >= n) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = q0 + tq * 4 + j;
            if (q < POOLED) logits[((size_t)w * 2 + h) * POOLED + q] = acc[i][j];
        }
    }
}

// One block per (window, head): softmax over 749 logits, then feat = alpha @ yp.
__global__ __launch_bounds__(256) void attn_kernel(const float* __restrict__ logits,
                                                   const float* __restrict__ yp,
                                                   float* __restrict__ alpha_out,
                                                   float* __restrict__ feat) {
    __shared__ float a[POOLED + 3];
    __shared__ float red[256];
    const int wi = blockIdx.x, h = blockIdx.y;
    const float* lg = logits + ((size_t)wi * 2 + h) * POOLED;
    float mx = -INFINITY;
    for (int q = threadIdx.x; q < POOLED; q += 256) {
        const float v = lg[q];
        a[q] = v;
        mx = fmaxf(mx, v);
    }
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    mx = red[0];
    __syncthreads();
    float sum = 0.f;
    for (int q = threadIdx.x; q < POOLED; q += 256) {
        const float e = expf(a[q] - mx);
        a[q] = e;
        sum += e;
    }
    red[threadIdx.x] = sum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const float inv = 1.f / red[0];
    __syncthreads();
    for (int q = threadIdx.x; q < POOLED; q += 256) {
        const float v = a[q] * inv;
        a[q] = v;
        alpha_out[((size_t)wi * 2 + h) * POOLED + q] = v;
    }
    __syncthreads();
    // 256 threads = 2 q-halves x 128 channels; rows of yp are 512 B, read coalesced
    const int c = threadIdx.x & 127, half = threadIdx.x >> 7;
    const float* y = yp + ((size_t)wi * 2 + h) * POOLED * C;
    float acc = 0.f;
    for (int q = half; q < POOLED; q += 2) acc = fmaf(a[q], y[(size_t)q * C + c], acc);
    red[threadIdx.x] = acc;
    __syncthreads();
    if (half == 0) feat[(size_t)wi * FEAT + h * C + c] = red[c] + red[128 + c];
}

// Dense stack for DW windows per block, 512 threads = one per hidden unit.
constexpr int DW = 8;

__global__ __launch_bounds__(512) void dense_kernel(const float* __restrict__ feat,
                                                    const float* __restrict__ d1k, const float* __restrict__ d1b,
                                                    const float* __restrict__ d2k, const float* __restrict__ d2b,
                                                    const float* __restrict__ d3k, const float* __restrict__ d3b,
                                                    int n, float* __restrict__ scores) {
    __shared__ float f[DW][FEAT];
    __shared__ float h1[DW][HID];
    __shared__ float h2[DW][HID];
    __shared__ float lg[DW][4];
    const int w0 = blockIdx.x * DW;
    const int j = threadIdx.x;
    for (int i = j; i < DW * FEAT; i += 512) {
        const int w = i / FEAT, k = i % FEAT;
        f[w][k] = (w0 + w < n) ? feat[(size_t)(w0 + w) * FEAT + k] : 0.f;
    }
    __syncthreads();
    float acc[DW];
#pragma unroll
    for (int w = 0; w < DW; ++w) acc[w] = d1b[j];
    for (int k = 0; k < FEAT; ++k) {
        const float wv = d1k[(size_t)k * HID + j];
#pragma unroll
        for (int w = 0; w < DW; ++w) acc[w] = fmaf(f[w][k], wv, acc[w]);
    }
#pragma unroll
    for (int w = 0; w < DW; ++w) h1[w][j] = fmaxf(acc[w], 0.f);
    __syncthreads();
#pragma unroll
    for (int w = 0; w < DW; ++w) acc[w] = d2b[j];
    for (int k = 0; k < HID; ++k) {
        const float wv = d2k[(size_t)k * HID + j];
#pragma unroll
        for (int w = 0; w < DW; ++w) acc[w] = fmaf(h1[w][k], wv, acc[w]);
    }
#pragma unroll
    for (int w = 0; w < DW; ++w) h2[w][j] = fmaxf(acc[w], 0.f);
    __syncthreads();
    // output layer: DW*3 dot products of length 512; one wave per (window, class) pair, 8 waves
    const int wave = j >> 6, lane = j & 63;
    for (int o = wave; o < DW * GNN_CLASSES; o += 8) {
        const int w = o / GNN_CLASSES, cl = o % GNN_CLASSES;
        float s = 0.f;
        for (int k = lane; k < HID; k += 64) s = fmaf(h2[w][k], d3k[(size_t)k * GNN_CLASSES + cl], s);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) lg[w][cl] = s + d3b[cl];
    }
    __syncthreads();
    if (j < DW && w0 + j < n) {
        const float a = lg[j][0], b = lg[j][1], c = lg[j][2];
        const float mx = fmaxf(a, fmaxf(b, c));
        const float ea = expf(a - mx), eb = expf(b - mx), ec = expf(c - mx);
        const float inv = 1.f / (ea + eb + ec);
        float* o = scores + (size_t)(w0 + j) * GNN_CLASSES;
        o[0] = ea * inv;
        o[1] = eb * inv;
        o[2] = ec * inv;
    }
}

int launch_backend(gnn_ctx* ctx, int64_t n, float* scores_dev) {
    const DeviceWeights& d = ctx->w;
    Workspace& ws = ctx->ws;
    hipLaunchKernelGGL(m_kernel, dim3((unsigned)n, 2), dim3(256), 0, ctx->stream, ws.mp, d.w_bias[0], d.w_bias[1],
                       d.slot[0], d.slot[1], ws.m);
    hipLaunchKernelGGL(logits_kernel, dim3((POOLED + LQ - 1) / LQ, (unsigned)((n + LW - 1) / LW), 2), dim3(256), 0,
                       ctx->stream, ws.m, d.w_qk[0], d.w_qk[1], (int)n, ws.logits);
    hipLaunchKernelGGL(attn_kernel, dim3((unsigned)n, 2), dim3(256), 0, ctx->stream, ws.logits, ws.yp, ws.alpha,
                       ws.feat);
    hipLaunchKernelGGL(dense_kernel, dim3((unsigned)((n + DW - 1) / DW)), dim3(512), 0, ctx->stream, ws.feat,
                       d.d1_k, d.d1_b, d.d2_k, d.d2_b, d.d3_k, d.d3_b, (int)n, scores_dev);
    GNN_HIP(hipGetLastError());
    return GNN_OK;
}

// tf.math.segment_mean: one thread per (segment, class); ids are sorted, so the segment is the
// half-open range found by binary search and is summed in window order.
__global__ void segment_mean_kernel(const float* __restrict__ scores, const int64_t* __restrict__ ids,
                                    int64_t n, int64_t n_seg, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_seg * GNN_CLASSES) return;
    const int64_t seg = i / GNN_CLASSES;
    const int cl = (int)(i % GNN_CLASSES);
    int64_t lo = 0, hi = n;   // first index with ids >= seg
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (ids[mid] < seg) lo = mid + 1; else hi = mid;
    }
    const int64_t a = lo;
    hi = n;                    // first index with ids > seg
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (ids[mid] <= seg) lo = mid + 1; else hi = mid;
    }
    float s = 0.f;
    for (int64_t k = a; k < lo; ++k) s += scores[k * GNN_CLASSES + cl];
    out[i] = lo > a ? s / (float)(lo - a) : 0.f;
}

}  // namespace gnn

using namespace gnn;

extern "C" int gnn_segment_mean(gnn_ctx* ctx, const float* scores_host, const int64_t* ids_host, int64_t n,
                                int64_t n_segments, float* out_host) {
    if (!ctx || n < 0 || n_segments < 0 || (n > 0 && (!scores_host || !ids_host)) ||
        (n_segments > 0 && !out_host)) {
        set_error("bad argument to gnn_segment_mean");
        return GNN_ERR_ARG;
    }
    if (n_segments == 0) return GNN_OK;
    for (int64_t i = 1; i < n; ++i)
        if (ids_host[i] < ids_host[i - 1]) {
            set_error("segment ids are not sorted");
            return GNN_ERR_ARG;
        }
    if (n > 0 && (ids_host[0] < 0 || ids_host[n - 1] >= n_segments)) {
        set_error("segment id out of range");
        return GNN_ERR_ARG;
    }
    GNN_HIP(hipSetDevice(ctx->device));
    float *ds = nullptr, *dout = nullptr;
    int64_t* di = nullptr;
    int rc = GNN_OK;
    auto step = [&](hipError_t e, const char* what) {
        if (rc == GNN_OK && e != hipSuccess) {
            set_error(std::string(what) + " failed: " + hipGetErrorString(e));
            rc = GNN_ERR_HIP;
        }
    };
    step(hipMalloc((void**)&ds, std::max<size_t>(1, (size_t)n * GNN_CLASSES * sizeof(float))), "hipMalloc scores");
    step(hipMalloc((void**)&di, std::max<size_t>(1, (size_t)n * sizeof(int64_t))), "hipMalloc ids");
    step(hipMalloc((void**)&dout, (size_t)n_segments * GNN_CLASSES * sizeof(float)), "hipMalloc out");
    if (n > 0 && rc == GNN_OK) {
        step(hipMemcpyAsync(ds, scores_host, (size_t)n * GNN_CLASSES * sizeof(float), hipMemcpyHostToDevice, ctx->stream), "copy scores");
        step(hipMemcpyAsync(di, ids_host, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream), "copy ids");
    }
    if (rc == GNN_OK) {
        const int64_t threads = n_segments * GNN_CLASSES;
        hipLaunchKernelGGL(segment_mean_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream,
                           ds, di, n, n_segments, dout);
        step(hipGetLastError(), "segment_mean launch");
        step(hipMemcpyAsync(out_host, dout, (size_t)n_segments * GNN_CLASSES * sizeof(float), hipMemcpyDeviceToHost, ctx->stream), "copy out");
        step(hipStreamSynchronize(ctx->stream), "sync");
    }
    (void)hipFree(ds);
    (void)hipFree(di);
    (void)hipFree(dout);
    return rc;
}
