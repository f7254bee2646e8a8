// Pieces shared by the fused front-end kernels (gnn_fused.hip: split-bf16, gnn_fused_c8.hip: f16 + fp8
// corrections): LDS geometry, kernel arguments, the tokenizer in closed form and the conv1 gather.
#pragma once
#include "gnn_common.h"

namespace gnn {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ROWB = 528;                    // LDS row stride in bytes
constexpr int LO_OFF = 256;                  // lo plane offset inside a row
constexpr int CARRY = KS - 1;                // 5 rows carried from the previous step
constexpr int BUF_ROWS = CARRY + FT;         // 133
constexpr int BUF_BYTES = BUF_ROWS * ROWB;   // 70224
constexpr int TOK_OFF = 2 * BUF_BYTES;       // u16 tokens of the whole window, toks[j] = position j-5
constexpr int TOK_COUNT = ((FSTEPS * FT + CARRY + KS) + 7) / 8 * 8;   // 6032
constexpr int SMEM_BYTES = TOK_OFF + ((TOK_COUNT * 2 + 15) / 16) * 16;
constexpr int FRAG_U4 = 64;                  // one fragment = 64 lanes x uint4

// v_max_f32 as is: hipcc wraps fmaxf of values it cannot prove quiet (MFMA results, loads) in a canonicalising v_max per operand,
// two more instructions per value in the VALU-bound row producers.  NaN stays NaN when both operands are NaN (LeakyReLU below).
__device__ __forceinline__ float vmax_raw(float a, float b) {
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// x - (float)(one f16 half of the packed word h) in ONE instruction: v_fma_mix_f32 reads the f16 half directly (h * -1.0 + x; exact,
// the value a v_cvt_f32_f16 + v_sub_f32 pair gives) - the f16 rounding residual of the two-limb splits and of the f16c6 rows
__device__ __forceinline__ float sub_f16_lo(float x, uint32_t h) {
    float d;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(x));
    return d;
}
__device__ __forceinline__ float sub_f16_hi(float x, uint32_t h) {
    float d;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(x));
    return d;
}
// LeakyReLU(0.1) = max(v, 0.1 v) since the slope is < 1: two VALU ops instead of mul + compare + select
#ifdef GNN_LRELU_FMAXF      // measurement variant: the compiler's fmaxf (with its canonicalising v_max)
__device__ __forceinline__ float lrelu_f(float v) { return fmaxf(v, v * LRELU); }
#else
__device__ __forceinline__ float lrelu_f(float v) { return vmax_raw(v, v * LRELU); }
#endif

// Maximum that PROPAGATES NaN (IEEE 754-2019 maximum; v_maximum3_f32 on gfx950), for the max-pool of y @ w_v: fmaxf returns
// the other operand when one is NaN, which would let the pooling hide an overflow of the f16-operand arithmetic behind
// finite scores; callers rely on non-finite scores to detect it (nn_classification.py: fallback to bf16x3).  Bit-identical
// to fmaxf on everything else.
__device__ __forceinline__ float max_nan(float a, float b) { return __builtin_elementwise_maximum(a, b); }

__device__ __forceinline__ int base_code_f(uint32_t b) {
    return b == 65 ? 0 : (b == 67 ? 1 : (b == 71 ? 2 : (b == 84 ? 3 : -1)));
}

struct FusedArgs {
    const uint8_t* bases;
    const float* conv1_k;        // (3, PAIR_ROWS, 128) f32 conv1 pair tables
    const float* conv1_b;
    const uint4* conv_frag[2];
    const float* conv_b[2];
    const uint4* wv_frag[2];
    const float* weff[2];
    const int32_t* pos_sorted[2];
    const int32_t* bucket_ptr[2];
    float* mp;
    float* yp;
    unsigned long long* cycles;   // PROF builds: 10 phase counters, summed over workgroups
};

// PROF instrumentation: s_memtime deltas, MFMA wave 0 -> counters 0..7, helper wave 4 -> 8, 9:
// 0 w_v+pool A, 1 conv2 loop, 2 wait B1, 3 conv2 epilogue+B2, 4 conv3 loop, 5 wait B3,
// 6 conv3 epilogue+B4, 7 w_v+pool B + wait B0, 8 helper m-partials (B+A), 9 helper conv1 gather.
#define GNN_TICK(i)                                                   \
    if constexpr (PROF) {                                             \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        cyc[i] += now_ - tick_;                                       \
        tick_ = now_;                                                 \
    }

// Token state of position t: -1 = before the window start (zero padding of the one-hot input),
// 0 = the 4-mer touches a non-ACGT byte (or lies past the last token), 1..256 = 4-mer code + 1
// (sequence.py:170-193, closed form).
__device__ __forceinline__ int token_state(const uint8_t* __restrict__ bases, int t) {
    if (t < 0) return -1;
    if (t >= T) return 0;
    const int c0 = base_code_f(bases[t]), c1 = base_code_f(bases[t + 1]), c2 = base_code_f(bases[t + 2]),
              c3 = base_code_f(bases[t + 3]);
    return (c0 | c1 | c2 | c3) < 0 ? 0 : 1 + c0 * 64 + c1 * 16 + c2 * 4 + c3;
}

// Row of the conv1 pair tables for the tokens (a, b) of two adjacent positions (layout built by
// gnn_load_weights): adjacent 4-mers overlap in 3 bases, so a valid pair is a 5-mer.
__device__ __forceinline__ uint32_t pair_row(int a, int b) {
    if (a > 0 && b > 0) return (uint32_t)((a - 1) * 4 + ((b - 1) & 3));
    if (a == 0 && b > 0) return 1024u + (uint32_t)(b - 1);
    if (a > 0 && b == 0) return 1280u + (uint32_t)(a - 1);
    if (a == 0) return 1536u;            // (N, N)
    if (b < 0) return 1537u;             // both before the window start: zero row
    return 1538u + (uint32_t)b;          // only the first one is before the window start
}

// conv1 + LeakyReLU for the 128 positions starting at t0, split to bf16 hi/lo, into rows 5..132 of
// xbuf.  conv1 on a one-hot input is a 6-row gather-sum of its kernel (model.py:11 + igloo.py:45-48);
// with the pair tables it is 3 rows: taps (0,1), (2,3), (4,5) of position t read the pairs starting
// at t-5, t-3, t-1.  prow[j] is the pair row of positions (j-5, j-4).  256 helper threads:
// thread = 4 channels x 16 CONSECUTIVE positions, so the 20 pair rows it needs are 40 contiguous
// bytes of LDS, fetched with three wide reads up front: the LDS pipe is busy feeding the matrix
// waves, and per-position index reads in the dependency chain of every load batch were what made
// the gather 4x slower beside the MFMA loops than alone.  Positions P0..P1 (of 16) are produced.
// Store::put(row, cq, v) applies LeakyReLU to the 4 channels cq*4.. of one position and writes them in
// the operand format of the calling kernel (split bf16 hi/lo, or f16 + fp8 corrections).
#ifndef GNN_GATHER_EARLY
#define GNN_GATHER_EARLY 4      // positions (of 16 per thread) of the next step's gather done between B1 and B2
#endif
#ifndef GNN_GATHER_BATCH
#define GNN_GATHER_BATCH 4      // positions whose 3 table loads each are in flight together
#endif
template <int P0, int P1, typename Store, int NBATCH = 0>
__device__ __forceinline__ void conv1_gather(unsigned char* __restrict__ xbuf, const uint16_t* __restrict__ prow,
                                             const float* __restrict__ pt, const float* __restrict__ b1,
                                             int t0, int ht) {
    const int cq = ht & 31, ug = ht >> 5;
    const f32x4 b = *reinterpret_cast<const f32x4*>(b1 + cq * 4);
    const unsigned char* pr = reinterpret_cast<const unsigned char*>(prow + t0 + ug * 16);   // 32-B aligned
    const uint4 r0 = *reinterpret_cast<const uint4*>(pr), r1 = *reinterpret_cast<const uint4*>(pr + 16);
    const uint2 r2 = *reinterpret_cast<const uint2*>(pr + 32);
    const uint32_t rw[10] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y};
    const float* ptc = pt + cq * 4;
    // positions per load batch (NBATCH must divide P1 - P0 when given)
    constexpr int NB = NBATCH > 0 ? NBATCH : ((P1 - P0) % GNN_GATHER_BATCH == 0 ? GNN_GATHER_BATCH : 4);
    static_assert((P1 - P0) % NB == 0, "gather batch must divide the position range");
#pragma unroll
    for (int i0 = P0; i0 < P1; i0 += NB) {
        f32x4 v[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            v[i] = b;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int e = i0 + i + 2 * j;                           // static after unrolling
                const uint32_t r = (rw[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu;
                v[i] += *reinterpret_cast<const f32x4*>(ptc + ((size_t)j * PAIR_ROWS + r) * C);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)      // LeakyReLU(0.1) + split into the MFMA operand planes of the row
            Store::put(xbuf + (CARRY + ug * 16 + i0 + i) * ROWB, cq, v[i]);
    }
}

}  // namespace gnn
