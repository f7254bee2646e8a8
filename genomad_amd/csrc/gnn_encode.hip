// Integer kernels: synthetic windows, 4-mer tokenizer, stand-alone byte -> one-hot encoder.
//
// tokenizer  == closed form of genomad/sequence.py:170-193 (tokenize_dna(seq, 4)) on padded,
//               upper-cased 6000-byte windows (nn_classification.py:72-73):
//               tok[i] = 0 if any of seq[i..i+3] is not A/C/G/T, else 1 + 4-mer code (first base
//               most significant, A0 C1 G2 T3).
// one-hot    == tf.one_hot(tokens, depth=257) (genomad/neural_network/model.py:9-11), written as
//               u8 / bf16 / f32.  HBM-write bound: 5997*257 elements out per 6000 bytes in.
#include "gnn_common.h"

namespace gnn {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// genomad_amd/synthetic.py::synth_windows, one thread per 4 output bytes.
__global__ __launch_bounds__(256) void synth_kernel(uint64_t seed, int64_t first, int64_t n,
                                                    uint32_t* __restrict__ out) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;   // 4-byte group index
    if (g >= n * (W / 4)) return;
    const int64_t wi = g / (W / 4);
    const int p0 = (int)(g - wi * (W / 4)) * 4;
    const uint64_t i = (uint64_t)(first + wi);
    // per-window composition thresholds (window_thresholds)
    const uint64_t r = splitmix64(seed + i + (1ull << 25));
    const uint64_t gc = 16384 + (r & 0x7FFF);
    const uint64_t sA = 24576 + ((r >> 16) & 0x3FFF);
    const uint64_t sC = 24576 + ((r >> 32) & 0x3FFF);
    const uint64_t tA = ((65536 - gc) * sA) >> 16;
    const uint64_t tC = tA + ((gc * sC) >> 16);
    const uint64_t tG = tA + gc;
    int64_t L = W;
    if (i % 16 == 5) L = 2500 + (int64_t)(splitmix64(seed + i) % 3501);
    int64_t run_off = -1, run_end = -1;
    if (i % 64 == 9) {
        const uint64_t rr = splitmix64(seed + i + (1ull << 24));
        const uint64_t len = 1 + rr % 200;
        run_off = (int64_t)((rr >> 16) % (W - len));
        run_end = run_off + (int64_t)len;
    }
    uint32_t word = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = p0 + k;
        const uint64_t u = splitmix64(seed ^ (i * (uint64_t)W + (uint64_t)p)) >> 48;
        uint32_t ch = u < tA ? 'A' : (u < tC ? 'C' : (u < tG ? 'G' : 'T'));
        if (p >= L || (p >= run_off && p < run_end)) ch = 'N';
        word |= ch << (8 * k);
    }
    out[g] = word;
}

__device__ __forceinline__ int base_code(uint32_t b) {
    // A=65 C=67 G=71 T=84 -> 0..3, anything else -> -1 (sequence.py:178-188)
    return b == 65 ? 0 : (b == 67 ? 1 : (b == 71 ? 2 : (b == 84 ? 3 : -1)));
}

__device__ __forceinline__ uint32_t token_at(const uint8_t* __restrict__ w, int t) {
    const int c0 = base_code(w[t]), c1 = base_code(w[t + 1]), c2 = base_code(w[t + 2]),
              c3 = base_code(w[t + 3]);
    if ((c0 | c1 | c2 | c3) < 0) return 0;
    return 1u + (uint32_t)(c0 * 64 + c1 * 16 + c2 * 4 + c3);
}

// One thread per 4 consecutive tokens of a window (reads 7 bytes, writes 8 bytes).
__global__ __launch_bounds__(256) void tokenize_kernel(const uint8_t* __restrict__ bases, int64_t n,
                                                       uint16_t* __restrict__ tokens) {
    constexpr int GROUPS = (T + 3) / 4;   // 1500 groups of 4 per window (last has 1 token)
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n * GROUPS) return;
    const int64_t wi = g / GROUPS;
    const int t0 = (int)(g - wi * GROUPS) * 4;
    const uint8_t* w = bases + wi * W;
    uint16_t* o = tokens + wi * T;
    int code[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) code[k] = (t0 + k < W) ? base_code(w[t0 + k]) : -1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (t0 + k < T) {
            const int bad = (code[k] | code[k + 1] | code[k + 2] | code[k + 3]) < 0;
            o[t0 + k] = bad ? 0 : (uint16_t)(1 + code[k] * 64 + code[k + 1] * 16 + code[k + 2] * 4 + code[k + 3]);
        }
    }
}

// Stand-alone encoder.  The output (n, 5997, 257) is treated as one flat array of elements;
// each thread produces VEC consecutive elements = one 16-byte store (u8: 16, bf16: 8, f32: 4), so
// a wave writes 1 KiB contiguous.  A 16-byte group spans at most two one-hot rows (257 > 16).
template <typename OutT, int VEC>
__global__ __launch_bounds__(256) void onehot_kernel(const uint8_t* __restrict__ bases, int64_t n,
                                                     OutT* __restrict__ out, OutT one) {
    const int64_t total = n * (int64_t)T * GNN_DEPTH;
    const int64_t e0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (e0 >= total) return;
    const int64_t row = e0 / GNN_DEPTH;                // global row = window * 5997 + t
    const int d0 = (int)(e0 - row * GNN_DEPTH);
    const int64_t wi = row / T;
    const int t = (int)(row - wi * T);
    const uint32_t tok0 = token_at(bases + wi * W, t);
    uint32_t tok1 = 0;
    if (d0 + VEC > GNN_DEPTH) {                        // group spills into the next row
        const int64_t row1 = row + 1;
        if (row1 < n * (int64_t)T) {
            const int64_t w1 = row1 / T;
            tok1 = token_at(bases + w1 * W, (int)(row1 - w1 * T));
        }
    }
    OutT v[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        const int d = d0 + k;
        const bool hit = d < GNN_DEPTH ? ((uint32_t)d == tok0) : ((uint32_t)(d - GNN_DEPTH) == tok1);
        v[k] = hit ? one : (OutT)0;
    }
    if (e0 + VEC <= total) {
        *reinterpret_cast<uint4*>(out + e0) = *reinterpret_cast<const uint4*>(v);
    } else {
        for (int k = 0; k < VEC && e0 + k < total; ++k) out[e0 + k] = v[k];
    }
}

int launch_synth(gnn_ctx* ctx, uint64_t seed, int64_t first, int64_t n, uint8_t* bases) {
    const int64_t groups = n * (W / 4);
    const int64_t blocks = (groups + 255) / 256;
    hipLaunchKernelGGL(synth_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, seed, first, n,
                       reinterpret_cast<uint32_t*>(bases));
    GNN_HIP(hipGetLastError());
    return GNN_OK;
}

int launch_tokenize(gnn_ctx* ctx, const uint8_t* bases, int64_t n, uint16_t* tokens) {
    const int64_t groups = n * ((T + 3) / 4);
    const int64_t blocks = (groups + 255) / 256;
    hipLaunchKernelGGL(tokenize_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, bases, n, tokens);
    GNN_HIP(hipGetLastError());
    return GNN_OK;
}

int launch_onehot(gnn_ctx* ctx, const uint8_t* bases, int64_t n, int dtype, void* out) {
    const int64_t total = n * (int64_t)T * GNN_DEPTH;
    if (dtype == GNN_OH_U8) {
        const int64_t blocks = ((total + 15) / 16 + 255) / 256;
        hipLaunchKernelGGL((onehot_kernel<uint8_t, 16>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                           bases, n, (uint8_t*)out, (uint8_t)1);
    } else if (dtype == GNN_OH_BF16) {
        const int64_t blocks = ((total + 7) / 8 + 255) / 256;
        hipLaunchKernelGGL((onehot_kernel<uint16_t, 8>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                           bases, n, (uint16_t*)out, (uint16_t)0x3F80);   // bf16 1.0
    } else {
        const int64_t blocks = ((total + 3) / 4 + 255) / 256;
        hipLaunchKernelGGL((onehot_kernel<uint32_t, 4>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                           bases, n, (uint32_t*)out, (uint32_t)0x3F800000);  // f32 1.0
    }
    GNN_HIP(hipGetLastError());
    return GNN_OK;
}

}  // namespace gnn
