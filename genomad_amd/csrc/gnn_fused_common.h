// Pieces shared by the streaming fused front-end kernels (gnn_fused_c6.hip, gnn_fused_x3.hip, gnn_fused_tc.hip): vector types, raw
// VALU forms, the tokenizer in closed form, the pair-table row of a token pair, the PROF tick.
#pragma once
#include "gnn_common.h"

namespace gnn {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CARRY = KS - 1;                // 5 rows carried from the previous step

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

// PROF instrumentation: cycle-counter deltas into the kernel's `cyc` array (matrix wave 0 -> counters 0..7, helper wave 4 -> 8..15;
// what each counter means is listed next to the kernel that fills it)
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

}  // namespace gnn
