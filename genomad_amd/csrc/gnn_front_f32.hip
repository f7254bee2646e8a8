// Unfused f32 front end (GNN_PREC_F32): the straightforward, exact-f32 statement of the
// encoder part of the network with every activation tensor in HBM.  It is the parity anchor
// for the fused path and the source of the x1/x2/x3 debug taps; it is not the fast path.
//
//   tokens -> x1 = lrelu(conv1) -> x2 = lrelu(conv2) -> x3 = lrelu(conv3)      igloo.py:45-72
//   per head h (A on x1, B on x3):                                              igloo.py:190-214
//     mp[pair]  = sum_c x[P[pair], c] * Weff[pair, c]
//     yp[q, :]  = max_{r<8} (x[8q+r, :] @ w_v)
#include "gnn_common.h"

namespace gnn {

__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : v * LRELU; }

// conv1 on the one-hot input == sum of 6 gathered kernel rows (model.py:11 + igloo.py:45-48).
// One thread per (position, 4 channels).
__global__ __launch_bounds__(256) void conv1_gather_kernel(const uint16_t* __restrict__ tokens,
                                                           const float* __restrict__ k1,
                                                           const float* __restrict__ b1, int64_t n,
                                                           float* __restrict__ x1) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t pos = idx >> 5;             // window * T + t
    const int c4 = (int)(idx & 31) * 4;
    if (pos >= n * T) return;
    const int64_t wi = pos / T;
    const int t = (int)(pos - wi * T);
    const uint16_t* tok = tokens + wi * T;
    float4 acc = *reinterpret_cast<const float4*>(b1 + c4);
#pragma unroll
    for (int k = 0; k < KS; ++k) {
        const int s = t + k - (KS - 1);
        if (s >= 0) {
            const float4 r = *reinterpret_cast<const float4*>(k1 + ((size_t)k * GNN_DEPTH + tok[s]) * C + c4);
            acc.x += r.x;
            acc.y += r.y;
            acc.z += r.z;
            acc.w += r.w;
        }
    }
    acc.x = lrelu(acc.x);
    acc.y = lrelu(acc.y);
    acc.z = lrelu(acc.z);
    acc.w = lrelu(acc.w);
    *reinterpret_cast<float4*>(x1 + pos * C + c4) = acc;
}

// Causal Conv1D(128, 6) + LeakyReLU in f32.  Block = 64 positions x 128 channels, 256 threads,
// thread = 4 positions x 8 channels; the input rows t0-5 .. t0+63 are staged in LDS (row padded
// to 129 floats against bank conflicts), the kernel streams from L2.
constexpr int CT = 64;
constexpr int CROW = C + 1;

template <int TAPS, bool POOL_OUT>
__global__ __launch_bounds__(256) void conv_f32_kernel(const float* __restrict__ xin,
                                                       const float* __restrict__ kern,   // (TAPS,128,128)
                                                       const float* __restrict__ bias,   // (128,) or null
                                                       float* __restrict__ out, int out_stride_w) {
    __shared__ float xs[(CT + TAPS - 1) * CROW];
    float* pool = xs;   // POOL_OUT: the input tile is dead once the products are done
    const int wi = blockIdx.y;
    const int t0 = blockIdx.x * CT;
    const float* xw = xin + (size_t)wi * T * C;
    for (int i = threadIdx.x; i < (CT + TAPS - 1) * C; i += 256) {
        const int r = i / C, c = i % C;
        const int t = t0 + r - (TAPS - 1);
        xs[r * CROW + c] = (t >= 0 && t < T) ? xw[(size_t)t * C + c] : 0.f;
    }
    __syncthreads();
    const int tp = threadIdx.x >> 4;   // 16 groups of 4 positions
    const int tc = threadIdx.x & 15;   // 16 groups of 8 channels
    float acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
    for (int k = 0; k < TAPS; ++k) {
        for (int ci = 0; ci < C; ++ci) {
            const float4 w0 = *reinterpret_cast<const float4*>(kern + ((size_t)k * C + ci) * C + tc * 8);
            const float4 w1 = *reinterpret_cast<const float4*>(kern + ((size_t)k * C + ci) * C + tc * 8 + 4);
            const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float xv = xs[(tp * 4 + i + k) * CROW + ci];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = fmaf(xv, wv[j], acc[i][j]);
            }
        }
    }
    if constexpr (!POOL_OUT) {
        float* ow = out + (size_t)wi * T * C;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = t0 + tp * 4 + i;
            if (t < T) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = lrelu(acc[i][j] + bias[tc * 8 + j]);
                *reinterpret_cast<float4*>(ow + (size_t)t * C + tc * 8) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(ow + (size_t)t * C + tc * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
        }
    } else {
        // y @ w_v followed by MaxPool1D(8) (igloo.py:208-210): rows of this block -> 8 pooled rows
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) pool[(tp * 4 + i) * C + tc * 8 + j] = acc[i][j];
        __syncthreads();
        for (int i = threadIdx.x; i < (CT / GNN_POOL) * C; i += 256) {
            const int q = i / C, c = i % C;
            const int qg = t0 / GNN_POOL + q;
            if (qg < POOLED) {
                float m = pool[(q * GNN_POOL) * C + c];
#pragma unroll
                for (int r = 1; r < GNN_POOL; ++r) m = fmaxf(m, pool[(q * GNN_POOL + r) * C + c]);
                out[(size_t)wi * out_stride_w + (size_t)qg * C + c] = m;
            }
        }
    }
}

// mp[w][h][pair] = sum_c x[w][pos[pair]][c] * Weff[pair][c]; 32 lanes per pair.
__global__ __launch_bounds__(256) void mpart_f32_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ weff,
                                                        const int32_t* __restrict__ pos_sorted,
                                                        float* __restrict__ mp, int head) {
    const int wi = blockIdx.y;
    const int pair = blockIdx.x * 8 + (threadIdx.x >> 5);   // entry index in bucket order
    const int l = threadIdx.x & 31;
    if (pair >= NPAIR) return;
    const int t = pos_sorted[pair];
    const float4 a = *reinterpret_cast<const float4*>(x + ((size_t)wi * T + t) * C + l * 4);
    const float4 b = *reinterpret_cast<const float4*>(weff + (size_t)pair * C + l * 4);
    float s = a.x * b.x;
    s = fmaf(a.y, b.y, s);
    s = fmaf(a.z, b.z, s);
    s = fmaf(a.w, b.w, s);
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) s += __shfl_xor(s, off, 32);
    if (l == 0) mp[((size_t)wi * 2 + head) * NPAIR + pair] = s;
}

int launch_front_f32(gnn_ctx* ctx, const uint8_t* bases, int64_t n) {
    const DeviceWeights& d = ctx->w;
    Workspace& ws = ctx->ws;
    int rc = launch_tokenize(ctx, bases, n, ws.tokens);
    if (rc) return rc;
    {
        const int64_t threads = n * T * 32;
        hipLaunchKernelGGL(conv1_gather_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                           ctx->stream, ws.tokens, d.conv1_k, d.conv1_b, n, ws.x[0]);
    }
    const dim3 cgrid((T + CT - 1) / CT, (unsigned)n);
    for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL((conv_f32_kernel<KS, false>), cgrid, dim3(256), 0, ctx->stream, ws.x[i],
                           d.conv_k[i], d.conv_b[i], ws.x[i + 1], 0);
    for (int h = 0; h < 2; ++h) {
        const float* xh = h == 0 ? ws.x[0] : ws.x[2];
        hipLaunchKernelGGL(mpart_f32_kernel, dim3(NPAIR / 8, (unsigned)n), dim3(256), 0, ctx->stream, xh,
                           d.weff_sorted[h], d.pos_sorted[h], ws.mp, h);
        hipLaunchKernelGGL((conv_f32_kernel<1, true>), cgrid, dim3(256), 0, ctx->stream, xh, d.w_v[h],
                           (const float*)nullptr, ws.yp + (size_t)h * POOLED * C, 2 * POOLED * C);
    }
    GNN_HIP(hipGetLastError());
    return GNN_OK;
}

}  // namespace gnn
