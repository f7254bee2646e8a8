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

// Stand-alone encoder (HBM-write bound).  The output (n, 5997, 257) is one flat array; a block owns
// RB = 128 consecutive one-hot rows, i.e. 128*257*sizeof(OutT) bytes = a whole number of 16-byte
// chunks starting on a 16-byte boundary.  The block first puts its 129 tokens into LDS (one 64-bit
// division per token, not per chunk), then every thread produces 16-byte chunks with 32-bit index
// math: a chunk is all zero except at most two elements (257 > 16: it touches at most two rows), so
// it is built as four dwords with at most two OR-ed in, and written with one coalesced 16-byte store
// (a wave writes 1 KiB contiguous).
constexpr int RB = 128;

template <int ISZ>   // element size in bytes: 1 (u8), 2 (bf16), 4 (f32); `one` is the element's bit pattern
__global__ __launch_bounds__(256) void onehot_kernel(const uint8_t* __restrict__ bases, int64_t n_rows,
                                                     uint4* __restrict__ out, uint32_t one) {
    constexpr int VEC = 16 / ISZ;                            // elements per 16-byte chunk
    constexpr int CHUNKS = RB * GNN_DEPTH * ISZ / 16;        // chunks per full block (2056 / 4112 / 8224)
    __shared__ uint32_t tok[RB + 1];
    const int64_t row0 = (int64_t)blockIdx.x * RB;
    if (threadIdx.x <= RB) {
        const int64_t row = row0 + threadIdx.x;              // global row = window * 5997 + t
        uint32_t v = 0xFFFFu;                                // past the end: matches no column
        if (row < n_rows) {
            const int64_t wi = row / T;
            v = token_at(bases + wi * W, (int)(row - wi * T));
        }
        tok[threadIdx.x] = v;
    }
    __syncthreads();
    const int rows_here = (int)min((int64_t)RB, n_rows - row0);
    const int elems_here = rows_here * GNN_DEPTH;
    uint4* dst = out + (size_t)blockIdx.x * CHUNKS;
    for (int c = threadIdx.x; c * VEC < elems_here; c += 256) {
        const int e0 = c * VEC;                              // element offset inside the block
        const int lrow = (int)(((uint32_t)e0 * 65281u) >> 24);   // e0 / 257 for e0 < 2^16 (65281 = ceil(2^24/257))
        const int d0 = e0 - lrow * GNN_DEPTH;
        // element index (inside this chunk) of the hot column of row lrow and of row lrow+1
        const int p0 = (int)tok[lrow] - d0;
        const int p1 = (int)tok[lrow + 1] + GNN_DEPTH - d0;
        uint32_t dw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int p = h ? p1 : p0;
            if (p >= 0 && p < VEC) {
                const int byte = p * ISZ;
                const uint32_t val = one << ((byte & 3) * 8);
#pragma unroll
                for (int i = 0; i < 4; ++i) dw[i] |= (byte >> 2) == i ? val : 0u;
            }
        }
        if (e0 + VEC <= elems_here) {
            dst[c] = make_uint4(dw[0], dw[1], dw[2], dw[3]);
        } else {                                             // ragged tail of the very last block
            const unsigned char* src = reinterpret_cast<const unsigned char*>(dw);
            unsigned char* o = reinterpret_cast<unsigned char*>(dst + c);
            for (int k = 0; k < (elems_here - e0) * ISZ; ++k) o[k] = src[k];
        }
    }
}

// Contig front end: a window is a span (start, len <= 6000) of one packed buffer of raw contig bytes.
// One wave per span counts the bytes equal to `byte` (the reference counts literal upper-case "N" on
// the raw string, sequence.py:38-39 / nn_classification.py:70-71).
__global__ __launch_bounds__(256) void span_count_kernel(const uint8_t* __restrict__ seq,
                                                         const int64_t* __restrict__ starts,
                                                         const int32_t* __restrict__ lens, int64_t n,
                                                         uint32_t byte, int32_t* __restrict__ counts) {
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int lane = threadIdx.x & 63;
    const uint8_t* p = seq + starts[i];
    const int len = lens[i];
    int c = 0;
    for (int k = lane; k < len; k += 64) c += p[k] == byte;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
    if (lane == 0) counts[i] = c;
}

// Span -> padded window: upper-case (Sequence.seq_ascii, sequence.py:35-36) and right-pad with 'N' to
// 6000 bytes (nn_classification.py:72).  One thread per 4 output bytes.
__global__ __launch_bounds__(256) void materialize_kernel(const uint8_t* __restrict__ seq,
                                                          const int64_t* __restrict__ starts,
                                                          const int32_t* __restrict__ lens, int64_t n,
                                                          uint32_t* __restrict__ out) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n * (W / 4)) return;
    const int64_t wi = g / (W / 4);
    const int p0 = (int)(g - wi * (W / 4)) * 4;
    const uint8_t* p = seq + starts[wi];
    const int len = lens[wi];
    uint32_t word = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t ch = 'N';
        if (p0 + k < len) {
            ch = p[p0 + k];
            if (ch >= 'a' && ch <= 'z') ch -= 32;      // str.upper() on ASCII
        }
        word |= ch << (8 * k);
    }
    out[g] = word;
}

int launch_span_count(gnn_ctx* ctx, const uint8_t* seq, const int64_t* starts, const int32_t* lens, int64_t n,
                      int byte, int32_t* counts) {
    hipLaunchKernelGGL(span_count_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, seq, starts, lens,
                       n, (uint32_t)byte, counts);
    GNN_HIP(hipGetLastError());
    return GNN_OK;
}

int launch_materialize(gnn_ctx* ctx, const uint8_t* seq, const int64_t* starts, const int32_t* lens, int64_t n,
                       uint8_t* bases) {
    const int64_t groups = n * (W / 4);
    hipLaunchKernelGGL(materialize_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, ctx->stream, seq,
                       starts, lens, n, reinterpret_cast<uint32_t*>(bases));
    GNN_HIP(hipGetLastError());
    return GNN_OK;
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
    const int64_t n_rows = n * (int64_t)T;
    const unsigned blocks = (unsigned)((n_rows + RB - 1) / RB);
    uint4* o = reinterpret_cast<uint4*>(out);
    if (dtype == GNN_OH_U8)
        hipLaunchKernelGGL((onehot_kernel<1>), dim3(blocks), dim3(256), 0, ctx->stream, bases, n_rows, o, 1u);
    else if (dtype == GNN_OH_BF16)
        hipLaunchKernelGGL((onehot_kernel<2>), dim3(blocks), dim3(256), 0, ctx->stream, bases, n_rows, o, 0x3F80u);
    else
        hipLaunchKernelGGL((onehot_kernel<4>), dim3(blocks), dim3(256), 0, ctx->stream, bases, n_rows, o, 0x3F800000u);
    GNN_HIP(hipGetLastError());
    return GNN_OK;
}

}  // namespace gnn
