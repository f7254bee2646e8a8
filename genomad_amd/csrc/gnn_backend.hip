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
#include <cstdlib>

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

// logits[w][h][q] = sum_p m[w][h][p] * w_qk[h][p][q].  Block tile: 64 windows x 128 q, K chunk 16;
// thread tile 4 windows x 8 q (32 FMAs per 3 LDS reads).  Each (window, q) sum runs over p in
// ascending order with one fmaf per term, independent of the batch the window is in, so results do
// not depend on how windows are sharded or batched.
constexpr int LW = 64, LQ = 128, LK = 16;

__global__ __launch_bounds__(256) void logits_kernel(const float* __restrict__ m,
                                                     const float* __restrict__ w_qk0,
                                                     const float* __restrict__ w_qk1, int n,
                                                     float* __restrict__ logits) {
    __shared__ __attribute__((aligned(16))) float ms[LK][LW + 4];
    __shared__ __attribute__((aligned(16))) float qs[LK][LQ];
    const int h = blockIdx.z;
    const float* w_qk = h ? w_qk1 : w_qk0;
    const int w0 = blockIdx.y * LW;
    const int q0 = blockIdx.x * LQ;
    const int tw = threadIdx.x >> 4;   // 16 groups x 4 windows
    const int tq = threadIdx.x & 15;   // 16 groups x (4 + 4) q: columns tq*4.. and 64 + tq*4..
    float acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[i][k] = 0.f;
    for (int p0 = 0; p0 < NP; p0 += LK) {
        // m tile: 16 p x 64 windows; thread -> (window, 4 consecutive p) as one float4 (NP % 4 == 0)
        {
            const int w = threadIdx.x >> 2, pq = (threadIdx.x & 3) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (w0 + w < n && p0 + pq < NP)
                v = *reinterpret_cast<const float4*>(m + ((size_t)(w0 + w) * 2 + h) * NP + p0 + pq);
            ms[pq][w] = v.x;
            ms[pq + 1][w] = v.y;
            ms[pq + 2][w] = v.z;
            ms[pq + 3][w] = v.w;
        }
        for (int i = threadIdx.x; i < LK * LQ; i += 256) {
            const int q = i % LQ, p = i / LQ;
            qs[p][q] = (p0 + p < NP && q0 + q < POOLED) ? w_qk[(size_t)(p0 + p) * POOLED + q0 + q] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int p = 0; p < LK; ++p) {
            const float4 a = *reinterpret_cast<const float4*>(&ms[p][tw * 4]);
            const float4 b0 = *reinterpret_cast<const float4*>(&qs[p][tq * 4]);
            const float4 b1 = *reinterpret_cast<const float4*>(&qs[p][64 + tq * 4]);
            const float av[4] = {a.x, a.y, a.z, a.w};
            const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[i][k] = fmaf(av[i], bv[k], acc[i][k]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int w = w0 + tw * 4 + i;
        if (w >= n) continue;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q = q0 + (k >> 2) * 64 + tq * 4 + (k & 3);
            if (q < POOLED) logits[((size_t)w * 2 + h) * POOLED + q] = acc[i][k];
        }
    }
}

// The same GEMM on the matrix pipe for the bf16x3 / bf16 precisions (6.3 MFLOP per window: at f32
// VALU rates it was a third of the back end's time).  A = m (windows x 2100, f32 in global, split to
// bf16 hi/lo in registers), B = w_qk pre-split and packed in MFMA fragment order (zero padded to
// 2112 x 768), D = A B accumulated in f32: rows = windows, so a window's logits never depend on which
// other windows share its tile.  Wave tile 64 windows x 64 q (2 x 2 MFMA blocks), block = 4 waves =
// 64 windows x 256 q; operands of k-step k+1 are fetched while the MFMAs of k-step k issue.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct QkA {
    float4 v[2][2];     // [m-block][half]: 8 consecutive patches of one window
};
struct QkB {
    uint4 v[2][2];      // [n-block][plane hi, lo]
};

__device__ __forceinline__ void qk_load(QkA& a, QkB& b, const float* __restrict__ mrow0, const float* __restrict__ mrow1,
                                        const uint4* __restrict__ bfrag, int ks, int lane) {
    const int k = ks * 16 + (lane >> 5) * 8;
    const float* rows[2] = {mrow0, mrow1};
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        if (k + 8 <= NP) {
            a.v[mb][0] = *reinterpret_cast<const float4*>(rows[mb] + k);
            a.v[mb][1] = *reinterpret_cast<const float4*>(rows[mb] + k + 4);
        } else {                                   // last k-step: patches >= 2100 do not exist
            float t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = k + e < NP ? rows[mb][k + e] : 0.f;
            a.v[mb][0] = make_float4(t[0], t[1], t[2], t[3]);
            a.v[mb][1] = make_float4(t[4], t[5], t[6], t[7]);
        }
    }
    const uint4* bp = bfrag + (size_t)ks * (QK_NBLK * 2 * 64);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        b.v[nb][0] = bp[nb * 2 * 64];
        b.v[nb][1] = bp[nb * 2 * 64 + 64];
    }
}

// F16 = false: split-bf16 limbs (8 + 8 significant bits, any f32 range); F16 = true: split-f16 limbs (11 + 11 bits: the three
// products then carry 22 bits, f32 class - the logits GEMM of GNN_PREC_F16X3, whose operands (m: |.| < 1e2, w_qk: ~1e-2) are
// far inside the f16 range)
template <int PASSES, bool F16>
__device__ __forceinline__ void qk_mfma(const QkA& a, const QkB& b, f32x16 (&acc)[2][2]) {
    typedef float f32x8 __attribute__((ext_vector_type(8)));
    typedef _Float16 f16x8q __attribute__((ext_vector_type(8)));
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const f32x8 x = {a.v[mb][0].x, a.v[mb][0].y, a.v[mb][0].z, a.v[mb][0].w,
                         a.v[mb][1].x, a.v[mb][1].y, a.v[mb][1].z, a.v[mb][1].w};
        if constexpr (F16) {
            const f16x8q ah = __builtin_convertvector(x, f16x8q);
            const f16x8q al = __builtin_convertvector(x - __builtin_convertvector(ah, f32x8), f16x8q);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const f16x8q bh = __builtin_bit_cast(f16x8q, b.v[nb][0]);
                if constexpr (PASSES == 3) {
                    const f16x8q bl = __builtin_bit_cast(f16x8q, b.v[nb][1]);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[mb][nb], 0, 0, 0);
                }
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[mb][nb], 0, 0, 0);
            }
        } else {
            const bf16x8 ah = __builtin_convertvector(x, bf16x8);
            const bf16x8 al = __builtin_convertvector(x - __builtin_convertvector(ah, f32x8), bf16x8);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const bf16x8 bh = __builtin_bit_cast(bf16x8, b.v[nb][0]);
                if constexpr (PASSES == 3) {
                    const bf16x8 bl = __builtin_bit_cast(bf16x8, b.v[nb][1]);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[mb][nb], 0, 0, 0);
                }
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[mb][nb], 0, 0, 0);
            }
        }
    }
}

template <int PASSES, bool F16 = false>
__global__ __launch_bounds__(256) void logits_mfma_kernel(const float* __restrict__ m, const uint4* __restrict__ frag0,
                                                          const uint4* __restrict__ frag1, int n,
                                                          float* __restrict__ logits) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = blockIdx.z;
    const int w0 = blockIdx.y * 64;
    const int nb0 = blockIdx.x * 8 + wave * 2;                 // first of this wave's two n-blocks
    const uint4* bfrag = (h ? frag1 : frag0) + (size_t)nb0 * 2 * 64 + lane;
    // rows past the end are clamped (computed, never stored)
    const float* mrow0 = m + ((size_t)min(w0 + (lane & 31), n - 1) * 2 + h) * NP;
    const float* mrow1 = m + ((size_t)min(w0 + 32 + (lane & 31), n - 1) * 2 + h) * NP;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    QkA a0, a1;
    QkB b0, b1;
    qk_load(a0, b0, mrow0, mrow1, bfrag, 0, lane);
#pragma unroll 1
    for (int ks = 0; ks < QK_KSTEPS; ks += 2) {                // QK_KSTEPS is even
        qk_load(a1, b1, mrow0, mrow1, bfrag, ks + 1, lane);
        qk_mfma<PASSES, F16>(a0, b0, acc);
        qk_load(a0, b0, mrow0, mrow1, bfrag, min(ks + 2, QK_KSTEPS - 1), lane);
        qk_mfma<PASSES, F16>(a1, b1, acc);
    }
    // C/D layout: column = lane&31 (q), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (window)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int w = w0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (w >= n) continue;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const int q = (nb0 + nb) * 32 + (lane & 31);
                if (q < POOLED) logits[((size_t)w * 2 + h) * POOLED + q] = acc[mb][nb][r];
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
    // 256 threads = 8 q-groups x 32 channel quads: a row of yp (512 B) is one float4 per lane of a
    // 32-lane group, four rows in flight per thread; partial sums are combined in a fixed order
    const int cq = threadIdx.x & 31, g = threadIdx.x >> 5;
    const float4* y = reinterpret_cast<const float4*>(yp + ((size_t)wi * 2 + h) * POOLED * C) + cq;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int q = g;
    for (; q + 24 < POOLED; q += 32) {
        // non-temporal: a pooled row is read once, by this block, and never again (12.6 GB per 16 384-window launch; -4 % of the back end's time)
        typedef float ntf4 __attribute__((ext_vector_type(4)));
        const ntf4* yn = reinterpret_cast<const ntf4*>(y);
        const ntf4 v0 = __builtin_nontemporal_load(yn + (size_t)q * (C / 4)), v1 = __builtin_nontemporal_load(yn + (size_t)(q + 8) * (C / 4)),
                   v2 = __builtin_nontemporal_load(yn + (size_t)(q + 16) * (C / 4)), v3 = __builtin_nontemporal_load(yn + (size_t)(q + 24) * (C / 4));
        const float a0 = a[q], a1 = a[q + 8], a2 = a[q + 16], a3 = a[q + 24];
        acc.x = fmaf(a0, v0.x, acc.x); acc.y = fmaf(a0, v0.y, acc.y); acc.z = fmaf(a0, v0.z, acc.z); acc.w = fmaf(a0, v0.w, acc.w);
        acc.x = fmaf(a1, v1.x, acc.x); acc.y = fmaf(a1, v1.y, acc.y); acc.z = fmaf(a1, v1.z, acc.z); acc.w = fmaf(a1, v1.w, acc.w);
        acc.x = fmaf(a2, v2.x, acc.x); acc.y = fmaf(a2, v2.y, acc.y); acc.z = fmaf(a2, v2.z, acc.z); acc.w = fmaf(a2, v2.w, acc.w);
        acc.x = fmaf(a3, v3.x, acc.x); acc.y = fmaf(a3, v3.y, acc.y); acc.z = fmaf(a3, v3.z, acc.z); acc.w = fmaf(a3, v3.w, acc.w);
    }
    for (; q < POOLED; q += 8) {
        const float4 v = y[(size_t)q * (C / 4)];
        const float aq = a[q];
        acc.x = fmaf(aq, v.x, acc.x); acc.y = fmaf(aq, v.y, acc.y); acc.z = fmaf(aq, v.z, acc.z); acc.w = fmaf(aq, v.w, acc.w);
    }
    __shared__ float4 part[8][32];
    part[g][cq] = acc;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int c = threadIdx.x;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += reinterpret_cast<const float*>(&part[k][c >> 2])[c & 3];
        feat[(size_t)wi * FEAT + h * C + c] = s;
    }
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
    // ReLU written so that NaN stays NaN (fmaxf would turn it into 0 and hide an overflow of the f16-operand
    // front ends behind plausible finite scores; main() relies on non-finite scores to fall back to bf16x3)
    for (int w = 0; w < DW; ++w) h1[w][j] = acc[w] < 0.f ? 0.f : acc[w];
    __syncthreads();
#pragma unroll
    for (int w = 0; w < DW; ++w) acc[w] = d2b[j];
    for (int k = 0; k < HID; ++k) {
        const float wv = d2k[(size_t)k * HID + j];
#pragma unroll
        for (int w = 0; w < DW; ++w) acc[w] = fmaf(h1[w][k], wv, acc[w]);
    }
#pragma unroll
    for (int w = 0; w < DW; ++w) h2[w][j] = acc[w] < 0.f ? 0.f : acc[w];
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

// The same dense stack on the matrix pipe (model.py:28-30, 40-44 with BN folded) for the f16-operand fast modes (f16c6,
// f16c8): 64 windows per block, D = X W with X split to f16 hi/lo in registers (11 + 11 significant bits) and W pre-split
// in MFMA fragment order (pack_frags), three products per k16 step -> f32-class accuracy (a split-bf16 version cost
// 1.3e-5 of the 1e-4 score tolerance, which these modes do not have to spare).  Inputs beyond the f16 range give
// non-finite scores, which is what these modes do everywhere else too (main() then falls back to bf16x3, whose dense
// stack is the exact f32 kernel above).  Wave w owns columns
// 128 w .. 128 w + 127 of the 512 hidden units for all 64 windows (2 x 4 accumulator blocks); the first layer's output
// (bias + ReLU) goes through LDS (64 x 516 f32, rows padded so that the row groups of a C/D register land on
// different banks) because every wave needs all of it as the second layer's A operand.  Dense3 + softmax as above.
constexpr int DM_MB = 1;                 // 32-window blocks per workgroup: 4096 windows = 128 workgroups (with 2: 64, half the chip idle and 0.13 ms)
constexpr int DM_ROWS = 32 * DM_MB;
constexpr int DM_STRIDE = HID + 4;

template <int KSTEPS, bool FROM_LDS>
__device__ __forceinline__ void dense_mfma_layer(const float* __restrict__ a0, int row_stride, const uint4* __restrict__ frag,
                                                 int wave, int lane, f32x16 (&acc)[DM_MB][4]) {
    typedef float f32x8 __attribute__((ext_vector_type(8)));
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
    for (int i = 0; i < DM_MB; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll 2
    for (int ks = 0; ks < KSTEPS; ++ks) {
        const int k = ks * 16 + (lane >> 5) * 8;
        f16x8 ah[DM_MB], al[DM_MB];
#pragma unroll
        for (int mb = 0; mb < DM_MB; ++mb) {
            const float4 lo4 = *reinterpret_cast<const float4*>(a0 + mb * row_stride + k), hi4 = *reinterpret_cast<const float4*>(a0 + mb * row_stride + k + 4);
            const f32x8 x = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
            ah[mb] = __builtin_convertvector(x, f16x8);
            al[mb] = __builtin_convertvector(x - __builtin_convertvector(ah[mb], f32x8), f16x8);
        }
        const uint4* bp = frag + ((size_t)ks * (HID / 32) + wave * 4) * 2 * 64 + lane;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const f16x8 bh = __builtin_bit_cast(f16x8, bp[nb * 2 * 64]), bl = __builtin_bit_cast(f16x8, bp[nb * 2 * 64 + 64]);
#pragma unroll
            for (int mb = 0; mb < DM_MB; ++mb) {
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mb], bl, acc[mb][nb], 0, 0, 0);
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mb], bh, acc[mb][nb], 0, 0, 0);
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mb], bh, acc[mb][nb], 0, 0, 0);
            }
        }
    }
}

// bias + ReLU (NaN stays NaN, see dense_kernel) of a wave's 64 x 128 tile into the LDS rows
__device__ __forceinline__ void dense_mfma_store(float* __restrict__ hs, const float* __restrict__ bias, int wave, int lane,
                                                 const f32x16 (&acc)[DM_MB][4]) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int col = (wave * 4 + nb) * 32 + (lane & 31);
        const float b = bias[col];
#pragma unroll
        for (int mb = 0; mb < DM_MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);   // C/D layout of 32x32
                const float v = acc[mb][nb][r] + b;
                hs[row * DM_STRIDE + col] = v < 0.f ? 0.f : v;
            }
    }
}

__global__ __launch_bounds__(256) void dense_mfma_kernel(const float* __restrict__ feat, const uint4* __restrict__ f1,
                                                         const float* __restrict__ d1b, const uint4* __restrict__ f2,
                                                         const float* __restrict__ d2b, const float* __restrict__ d3k,
                                                         const float* __restrict__ d3b, int n, float* __restrict__ scores) {
    __shared__ __attribute__((aligned(16))) float hs[DM_ROWS * DM_STRIDE];
    __shared__ float lg[DM_ROWS][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w0 = blockIdx.x * DM_ROWS;
    f32x16 acc[DM_MB][4];
    // rows past the end are clamped (computed, never stored); the launch guarantees whole 32-row blocks exist for mb > 0 or clamps too
    static_assert(DM_MB == 1, "the first layer's row clamp is written for one 32-window block per workgroup");
    dense_mfma_layer<FEAT / 16, false>(feat + (size_t)min(w0 + (lane & 31), n - 1) * FEAT, 32 * FEAT, f1, wave, lane, acc);
    dense_mfma_store(hs, d1b, wave, lane, acc);
    __syncthreads();
    dense_mfma_layer<HID / 16, true>(hs + (lane & 31) * DM_STRIDE, 32 * DM_STRIDE, f2, wave, lane, acc);
    __syncthreads();                                   // every wave has read h1: the rows are reused for h2
    dense_mfma_store(hs, d2b, wave, lane, acc);
    __syncthreads();
    for (int o = wave; o < DM_ROWS * GNN_CLASSES; o += 4) {
        const int w = o / GNN_CLASSES, cl = o % GNN_CLASSES;
        float s = 0.f;
        for (int k = lane; k < HID; k += 64) s = fmaf(hs[w * DM_STRIDE + k], d3k[(size_t)k * GNN_CLASSES + cl], s);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) lg[w][cl] = s + d3b[cl];
    }
    __syncthreads();
    const int j = threadIdx.x;
    if (j < DM_ROWS && w0 + j < n) {
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

int launch_backend(gnn_ctx* ctx, int64_t n, int precision, float* scores_dev) {
    const DeviceWeights& d = ctx->w;
    Workspace& ws = ctx->ws;
    hipLaunchKernelGGL(m_kernel, dim3((unsigned)n, 2), dim3(256), 0, ctx->stream, ws.mp, d.w_bias[0], d.w_bias[1],
                       d.slot[0], d.slot[1], ws.m);
    static_assert(QK_KSTEPS % 2 == 0 && QK_NBLK % 8 == 0, "logits_mfma_kernel tiling");
    static const bool logits_f32 = debug_switch("GNN_LOGITS_F32");      // A/B measurements only
    const dim3 qk_grid(QK_NBLK / 8, (unsigned)((n + 63) / 64), 2);
    const uint4* qf0 = reinterpret_cast<const uint4*>(d.wqk_frag[0]);
    const uint4* qf1 = reinterpret_cast<const uint4*>(d.wqk_frag[1]);
    if (precision == GNN_PREC_BF16X3 || precision == GNN_PREC_F16C8 || precision == GNN_PREC_F16C6)   // the logits GEMM keeps split-bf16 x 3
        hipLaunchKernelGGL(logits_mfma_kernel<3>, qk_grid, dim3(256), 0, ctx->stream, ws.m, qf0, qf1, (int)n, ws.logits);
    else if (precision == GNN_PREC_BF16)
        hipLaunchKernelGGL(logits_mfma_kernel<1>, qk_grid, dim3(256), 0, ctx->stream, ws.m, qf0, qf1, (int)n, ws.logits);
    else if (precision == GNN_PREC_F16X3 && !logits_f32)
        // the default arithmetic: split-f16 x 3 on the matrix pipe (22 significant bits, f32 accumulate: f32 class; 0.25 instead of
        // 0.64 ms per 4096 windows for the exact-f32 VALU kernel, which GNN_LOGITS_F32=1 selects for A/B runs)
        hipLaunchKernelGGL((logits_mfma_kernel<3, true>), qk_grid, dim3(256), 0, ctx->stream, ws.m,
                           reinterpret_cast<const uint4*>(d.wqk_frag_h[0]), reinterpret_cast<const uint4*>(d.wqk_frag_h[1]), (int)n, ws.logits);
    else    // GNN_PREC_F32 (and GNN_PREC_F16X3 with GNN_LOGITS_F32=1): exact f32 FMAs
        hipLaunchKernelGGL(logits_kernel, dim3((POOLED + LQ - 1) / LQ, (unsigned)((n + LW - 1) / LW), 2), dim3(256), 0,
                           ctx->stream, ws.m, d.w_qk[0], d.w_qk[1], (int)n, ws.logits);
    hipLaunchKernelGGL(attn_kernel, dim3((unsigned)n, 2), dim3(256), 0, ctx->stream, ws.logits, ws.yp, ws.alpha,
                       ws.feat);
    if (precision != GNN_PREC_F16C6 && precision != GNN_PREC_F16C8)    // exact f32 FMAs (f32 range and accuracy)
        hipLaunchKernelGGL(dense_kernel, dim3((unsigned)((n + DW - 1) / DW)), dim3(512), 0, ctx->stream, ws.feat,
                           d.d1_k, d.d1_b, d.d2_k, d.d2_b, d.d3_k, d.d3_b, (int)n, scores_dev);
    else                                                                // split-f16 x 3 on the matrix pipe
        hipLaunchKernelGGL(dense_mfma_kernel, dim3((unsigned)((n + DM_ROWS - 1) / DM_ROWS)), dim3(256), 0, ctx->stream, ws.feat,
                           reinterpret_cast<const uint4*>(d.d1_frag), d.d1_b, reinterpret_cast<const uint4*>(d.d2_frag), d.d2_b,
                           d.d3_k, d.d3_b, (int)n, scores_dev);
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
