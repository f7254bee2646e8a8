// Fused front end (GNN_PREC_BF16X3 / GNN_PREC_BF16): bases -> tokens -> conv1 -> conv2 -> conv3
// and both IGLOO heads' partials, with every activation kept in LDS.
//
// One workgroup (8 waves) streams ONE window through time in steps of FT = 128 positions.  The
// convolutions are causal (igloo.py:45-47,66: padding="causal"), so step s only needs the last 5
// rows of the previous step, which stay in LDS ("carry" rows) — no halo recompute.
//
//   LDS  bufX : rows 0..4 carry | rows 5..132 = x1 of this step
//        bufY : rows 0..4 x2 carry | rows 5..132 = x2 of this step, later overwritten by x3
//        row  = 128 ch bf16 hi (256 B) | 128 ch bf16 lo (256 B) | 16 B pad  (528 B stride keeps the
//               16-lane groups of ds_read_b128 on 16 distinct 16-B slots: 528/4 mod 64 = 4)
//
// Contractions (conv2, conv3: M=128 rows, K=768, N=128; y@w_v: K=128) run on
// v_mfma_f32_32x32x16_bf16 with f32 accumulation.  bf16x3: every f32 operand is split as
// hi = bf16(x), lo = bf16(x - hi) and x*w ~= hi*hi + hi*lo + lo*hi (3 MFMA passes): measured
// max |dscore| 1.5e-5 against the fp64 oracle, inside the 1e-4 tolerance that single-pass bf16
// (7e-3) and fp16 (1e-3) miss.  Weights are pre-split and pre-shuffled on the host into MFMA
// fragment order, so a wave loads a fragment as one coalesced 1 KiB global_load_dwordx4 straight
// into VGPRs (L2 resident); wave w owns output channels 32w..32w+31 for all 128 rows.
//
// F16 = true (GNN_PREC_F16X3) runs the same three passes on v_mfma_f32_32x32x16_f16 with f16 limbs:
// hi = f16(x), lo = f16(x - hi) carry 11 + 11 significant bits instead of 8 + 8, which puts the result
// in f32 class (emulation: max |dscore| 6.5e-7 against 1.6e-5 for the bf16 split, profiles/
// r02_precision_study.json) at the same cost; the price is the f16 range (|activation| < 65504).
#include <cstring>

#include "gnn_fused_common.h"

namespace gnn {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// one 32x32x16 MFMA on 16-bit operands given as raw 128-bit fragments
template <bool F16>
__device__ __forceinline__ f32x16 mma16(uint4 a, uint4 b, f32x16 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// One GEMM tile of the wave: 4 m-blocks (128 rows of the LDS buffer) x 1 n-block (32 columns),
// K = NTAPS * 128.  SWAP: D = W^T X^T (columns of D are positions; used by the convs so that a
// lane ends up with 4 consecutive channels of one position -> 8-byte LDS writes).  !SWAP: D = X W
// (rows are positions; used by y@w_v so that the 8-row max-pool is 4 registers + one lane swap).
//
// Software pipeline (one matrix wave per SIMD, so nothing else hides its latency): weight fragments
// are fetched from L2 three k-steps ahead (ring of four), activation fragments from LDS one k-step
// ahead (ring of two); sched_group_barriers interleave the loads 1:1 with the MFMAs and the
// sched_barrier at the end of a k-step keeps hipcc from sinking loads back down to their first use
// (left alone it emits load; s_waitcnt vmcnt(0); mfma — every L2 round trip exposed).
template <int PASSES>
struct WFrag {
    uint4 v[PASSES == 3 ? 2 : 1];
};
template <int PASSES>
struct XFrag {
    uint4 v[4][PASSES == 3 ? 2 : 1];
};

template <int PASSES>
__device__ __forceinline__ void load_w(WFrag<PASSES>& f, const uint4* __restrict__ wfrag, int ks) {
    const uint4* wp = wfrag + (size_t)ks * (4 * 2 * FRAG_U4);
    f.v[0] = wp[0];
    if constexpr (PASSES == 3) f.v[1] = wp[FRAG_U4];
}

template <int PASSES>
__device__ __forceinline__ void load_x(XFrag<PASSES>& f, const unsigned char* __restrict__ xl, int ks) {
    const unsigned char* xp = xl + (ks >> 3) * ROWB + (ks & 7) * 32;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        f.v[mb][0] = *reinterpret_cast<const uint4*>(xp + mb * 32 * ROWB);
        if constexpr (PASSES == 3) f.v[mb][1] = *reinterpret_cast<const uint4*>(xp + mb * 32 * ROWB + LO_OFF);
    }
}

template <bool SWAP, int PASSES, bool F16>
__device__ __forceinline__ void mfma_block(const WFrag<PASSES>& w, const XFrag<PASSES>& x, f32x16 (&acc)[4]) {
    const uint4 wh = w.v[0];
    if constexpr (PASSES == 3) {
        const uint4 wl = w.v[1];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[mb] = SWAP ? mma16<F16>(wl, x.v[mb][0], acc[mb]) : mma16<F16>(x.v[mb][0], wl, acc[mb]);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[mb] = SWAP ? mma16<F16>(wh, x.v[mb][1], acc[mb]) : mma16<F16>(x.v[mb][1], wh, acc[mb]);
    }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) acc[mb] = SWAP ? mma16<F16>(wh, x.v[mb][0], acc[mb]) : mma16<F16>(x.v[mb][0], wh, acc[mb]);
}

// One k-step region: the MFMAs of k-step k (fragments WCUR, XCUR) with the loads of later k-steps
// spread between them (activations of k+1 from LDS, weights of k+3 from L2), so that the matrix
// pipe never waits for a block of loads to issue.  The sched_group_barriers spell the interleave.
template <bool SWAP, int PASSES, bool F16>
__device__ __forceinline__ void gemm_region(const WFrag<PASSES>& wcur, WFrag<PASSES>& wload, const XFrag<PASSES>& xcur,
                                            XFrag<PASSES>& xload, const unsigned char* __restrict__ xl,
                                            const uint4* __restrict__ wfrag, int ks_x, int ks_w, f32x16 (&acc)[4]) {
    // GNN_ABL_NOX / GNN_ABL_NOW: measurement-only ablations (scripts/mkvariant.sh) that compile the
    // operand loads out — wrong results, used to show the launch is power- rather than cycle-bound
#ifndef GNN_ABL_NOX
    load_x(xload, xl, ks_x);
#endif
#ifndef GNN_ABL_NOW
    load_w(wload, wfrag, ks_w);
#endif
    mfma_block<SWAP, PASSES, F16>(wcur, xcur, acc);
    if constexpr (PASSES == 3) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <bool SWAP, int NTAPS, int PASSES, bool F16>
__device__ __forceinline__ void gemm_tile(const unsigned char* __restrict__ xbuf,   // row of u=0, tap 0
                                          const uint4* __restrict__ wfrag,           // + nblk*2*64 + lane
                                          f32x16 (&acc)[4], int lane) {
    constexpr int NK = NTAPS * (C / 16);         // k-steps of 16, a multiple of 4
    const unsigned char* xl = xbuf + (lane & 31) * ROWB + (lane >> 5) * 16;
    WFrag<PASSES> w0, w1, w2, w3;                // ring: W(k) lives in w[k%4], loaded 3 k-steps ahead
    XFrag<PASSES> xa, xb;                        // X(k) lives in x[k%2], loaded 1 k-step ahead
    load_w(w0, wfrag, 0);
    load_w(w1, wfrag, 1);
    load_w(w2, wfrag, 2);
    load_x(xa, xl, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int ks = 0; ks < NK; ks += 4) {
        // clamped indices: the prefetches past the end are harmless re-reads
        gemm_region<SWAP, PASSES, F16>(w0, w3, xa, xb, xl, wfrag, ks + 1, min(ks + 3, NK - 1), acc);
        gemm_region<SWAP, PASSES, F16>(w1, w0, xb, xa, xl, wfrag, ks + 2, min(ks + 4, NK - 1), acc);
        gemm_region<SWAP, PASSES, F16>(w2, w1, xa, xb, xl, wfrag, ks + 3, min(ks + 5, NK - 1), acc);
        gemm_region<SWAP, PASSES, F16>(w3, w2, xb, xa, xl, wfrag, min(ks + 4, NK - 1), min(ks + 6, NK - 1), acc);
    }
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// LeakyReLU + split of two values into packed 16-bit hi / lo words.  Written on pairs so that hipcc
// selects the packed forms (v_pk_mul_f32, v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 on both halves,
// v_pk_add_f32).  hi = round16(x) (RNE), lo = round16(x - hi).
template <bool F16>
__device__ __forceinline__ void lrelu_split2(f32x2 v, uint32_t& hi, uint32_t& lo) {
    const f32x2 s = v * LRELU;
    v = f32x2{fmaxf(v[0], s[0]), fmaxf(v[1], s[1])};
    if constexpr (F16) {
        const f16x2 h = __builtin_convertvector(v, f16x2);
        hi = __builtin_bit_cast(uint32_t, h);
        const f32x2 back = {(float)h[0], (float)h[1]};
        lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(v - back, f16x2));
    } else {
        hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
        const f32x2 back = {__uint_as_float(hi << 16), __uint_as_float(hi & 0xFFFF0000u)};
        lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(v - back, bf16x2));
    }
}

// conv1 gather epilogue of this kernel: 4 channels of one position -> 16-bit hi / lo planes
template <bool F16>
struct StoreSplit16 {
    static __device__ __forceinline__ void put(unsigned char* __restrict__ row, int cq, f32x4 v) {
        uint2 h, l;
        lrelu_split2<F16>(f32x2{v[0], v[1]}, h.x, l.x);
        lrelu_split2<F16>(f32x2{v[2], v[3]}, h.y, l.y);
        *reinterpret_cast<uint2*>(row + cq * 8) = h;
        *reinterpret_cast<uint2*>(row + cq * 8 + LO_OFF) = l;
    }
};

// accumulators start at the bias (the bias add of the epilogue, for free): in the D = W^T X^T layout
// register rg*4+e of a lane is output channel wave*32 + rg*8 + (lane>>5)*4 + e for every m-block
__device__ __forceinline__ void acc_init_bias(f32x16 (&acc)[4], const float* __restrict__ bias, int wave, int lane) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias + wave * 32 + rg * 8 + (lane >> 5) * 4);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[mb][rg * 4 + e] = b[e];
    }
}

// conv epilogue: LeakyReLU (the bias is already in the accumulators), split to bf16 hi/lo, write
// rows 5..132 of the output buffer.
// C/D layout of 32x32 MFMA: column = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
template <bool F16>
__device__ __forceinline__ void conv_epilogue(unsigned char* __restrict__ obuf, const f32x16 (&acc)[4], int wave, int lane) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const int f0 = wave * 32 + rg * 8 + (lane >> 5) * 4;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            uint2 h, l;
            lrelu_split2<F16>(f32x2{acc[mb][rg * 4], acc[mb][rg * 4 + 1]}, h.x, l.x);
            lrelu_split2<F16>(f32x2{acc[mb][rg * 4 + 2], acc[mb][rg * 4 + 3]}, h.y, l.y);
            unsigned char* o = obuf + (CARRY + mb * 32 + (lane & 31)) * ROWB + f0 * 2;
            *reinterpret_cast<uint2*>(o) = h;
            *reinterpret_cast<uint2*>(o + LO_OFF) = l;
        }
    }
}

// y @ w_v on the current 128 rows + MaxPool1D(8) -> yp rows (igloo.py:208-210)
template <int PASSES, bool F16>
__device__ __forceinline__ void wv_pool(const unsigned char* __restrict__ xbuf, const uint4* __restrict__ wfrag,
                                        float* __restrict__ yp_w, int t0, int wave, int lane) {
    f32x16 acc[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
    gemm_tile<false, 1, PASSES, F16>(xbuf + CARRY * ROWB, wfrag, acc, lane);
    // rows 8rg..8rg+3 of a 32-row block sit in lanes 0-31, rows 8rg+4..8rg+7 in lanes 32-63: the
    // 8-row max is 4 registers + one exchange with lane^32 (v_permlane32_swap, no LDS round trip).
    // All 16 pooled values are reduced first, then stored under one predicate.
    float m[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int mb = i >> 2, rg = i & 3;
        const float v = max_nan(max_nan(acc[mb][rg * 4], acc[mb][rg * 4 + 1]), max_nan(acc[mb][rg * 4 + 2], acc[mb][rg * 4 + 3]));
        const unsigned bits = __float_as_uint(v);
        const auto sw = __builtin_amdgcn_permlane32_swap(bits, bits, false, false);
        m[i] = max_nan(v, __uint_as_float(lane < 32 ? sw[1] : sw[0]));
    }
    const int q0 = t0 / GNN_POOL;
    const int nq = min(16, POOLED - q0);              // pooled rows of this step that exist (q < 749)
    if (lane < 32) {
        float* dst = yp_w + (size_t)q0 * C + wave * 32 + lane;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < nq) dst[(size_t)i * C] = m[i];
    }
}

// pair dot products of this step (igloo.py:192-204, w_mult*w_summer folded): entries e_begin..e_end
// of the position-sorted pair list all read rows of the current step.  16 lanes per entry, 8
// channels per lane; 4 entries per lane group are in flight at once (their loads are independent:
// weights stream linearly in entry order, rows come from LDS) so the L2 latency is paid once per
// batch.  Results are written in entry order (contiguous), the back end un-permutes.
template <bool F16>
__device__ __forceinline__ void m_partials(const unsigned char* __restrict__ xbuf, const float* __restrict__ weff,
                                           const int32_t* __restrict__ pos, int t0, int e_begin, int e_end,
                                           float* __restrict__ mp_w, int wave, int lane) {
    constexpr int MB = 4;
    const int sub = lane & 15;
    // the 16 lanes of an entry share e, so a lane group enters/leaves the loop together and the
    // width-16 shuffles below only ever read lanes that are active
    for (int e = e_begin + wave * 4 + (lane >> 4); e < e_end; e += 16 * MB) {
        int ei[MB], u[MB];
        float4 w0[MB], w1[MB];
        uint4 h[MB], l[MB];
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            ei[i] = min(e + 16 * i, e_end - 1);      // clamped duplicates are computed but not stored
            u[i] = pos[ei[i]] - t0;
        }
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            w0[i] = *reinterpret_cast<const float4*>(weff + (size_t)ei[i] * C + sub * 8);
            w1[i] = *reinterpret_cast<const float4*>(weff + (size_t)ei[i] * C + sub * 8 + 4);
        }
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const unsigned char* xr = xbuf + (CARRY + u[i]) * ROWB + sub * 16;
            h[i] = *reinterpret_cast<const uint4*>(xr);
            l[i] = *reinterpret_cast<const uint4*>(xr + LO_OFF);
        }
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const uint32_t hv[4] = {h[i].x, h[i].y, h[i].z, h[i].w}, lv[4] = {l[i].x, l[i].y, l[i].z, l[i].w};
            const float wv[8] = {w0[i].x, w0[i].y, w0[i].z, w0[i].w, w1[i].x, w1[i].y, w1[i].z, w1[i].w};
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float x0, x1;
                if constexpr (F16) {
                    const f16x2 hh = __builtin_bit_cast(f16x2, hv[k]), ll = __builtin_bit_cast(f16x2, lv[k]);
                    x0 = (float)hh[0] + (float)ll[0];
                    x1 = (float)hh[1] + (float)ll[1];
                } else {
                    x0 = __uint_as_float(hv[k] << 16) + __uint_as_float(lv[k] << 16);
                    x1 = __uint_as_float(hv[k] & 0xFFFF0000u) + __uint_as_float(lv[k] & 0xFFFF0000u);
                }
                s = fmaf(x0, wv[2 * k], s);
                s = fmaf(x1, wv[2 * k + 1], s);
            }
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) s += __shfl_xor(s, off, 16);
            if (sub == 0 && e + 16 * i < e_end) mp_w[e + 16 * i] = s;
        }
    }
}

// Warp-specialised workgroup of 8 waves streaming one window through 47 steps of 128 positions:
//   waves 0-3 ("MFMA waves", one per SIMD): y@w_v + pool of head A, conv2, conv3, y@w_v + pool of head B
//   waves 4-7 ("helpers", the second wave of each SIMD): everything that is memory-latency bound —
//     the conv1 gather of the NEXT step and the IGLOO pair dot products of both heads — so that it
//     overlaps with the MFMA waves instead of serialising with them.
// Per step (B0..B4 = workgroup barriers):
//   B0  MFMA: w_v A, conv2 loop  [read bufX]      helpers: m-partials B(s-1) [bufY], m-partials A(s) [bufX]
//   B1  MFMA: conv2 epilogue -> bufY (x2)          helpers: x1 carry rows -> bufX rows 0..4
//   B2  MFMA: conv3 loop [read bufY]               helpers: conv1 gather of step s+1 -> bufX rows 5..132
//   B3  MFMA: conv3 epilogue -> bufY (x3)          helpers: x2 carry rows -> bufY rows 0..4
//   B4  MFMA: w_v B [read bufY]
template <int PASSES, bool PROF, bool F16>
__global__ __launch_bounds__(512, 2) void fused_front_kernel(FusedArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];
    unsigned char* bufX = smem;
    unsigned char* bufY = smem + BUF_BYTES;
    uint16_t* toks = reinterpret_cast<uint16_t*>(smem + TOK_OFF);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool helper = wave >= 4;
    const int hw = wave & 3;                 // index within the role
    const int ht = tid & 255;
    const int64_t wi = blockIdx.x;
    const uint8_t* bases = a.bases + wi * W;
    float* mp_w[2] = {a.mp + (wi * 2 + 0) * NPAIR, a.mp + (wi * 2 + 1) * NPAIR};
    float* yp_w[2] = {a.yp + (wi * 2 + 0) * (size_t)POOLED * C, a.yp + (wi * 2 + 1) * (size_t)POOLED * C};
    const uint4* cfrag[2] = {a.conv_frag[0] + hw * 2 * FRAG_U4 + lane, a.conv_frag[1] + hw * 2 * FRAG_U4 + lane};
    const uint4* vfrag[2] = {a.wv_frag[0] + hw * 2 * FRAG_U4 + lane, a.wv_frag[1] + hw * 2 * FRAG_U4 + lane};

    // causal zero padding: carry rows of both buffers start at zero
    for (int i = tid; i < CARRY * ROWB / 16; i += 512) {
        reinterpret_cast<uint4*>(bufX)[i] = make_uint4(0, 0, 0, 0);
        reinterpret_cast<uint4*>(bufY)[i] = make_uint4(0, 0, 0, 0);
    }
    // pair rows of the whole window: toks[j] = conv1 pair-table row of positions (j-5, j-4)
    for (int j = tid; j < TOK_COUNT; j += 512) {
        const int t = j - CARRY;
        toks[j] = (uint16_t)pair_row(token_state(bases, t), token_state(bases, t + 1));
    }
    __syncthreads();
    if (helper) conv1_gather<0, FT / 8, StoreSplit16<F16>>(bufX, toks, a.conv1_k, a.conv1_b, 0, ht);
    // static priority for the matrix waves (measured neutral against no priority and against
    // prioritising the helpers; kept so the matrix pipe never loses an issue slot to a helper)
    else __builtin_amdgcn_s_setprio(2);

    unsigned long long cyc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tick_ = 0;

#pragma unroll 1
    for (int step = 0; step < FSTEPS; ++step) {
        const int t0 = step * FT;
        __syncthreads();                                                         // ---- B0
        if constexpr (PROF) tick_ = step == 0 ? __builtin_readcyclecounter() : tick_;
        if (!helper) {
            GNN_TICK(7)
            wv_pool<PASSES, F16>(bufX, vfrag[0], yp_w[0], t0, hw, lane);
            GNN_TICK(0)
            f32x16 acc[4];
            acc_init_bias(acc, a.conv_b[0], hw, lane);
            gemm_tile<true, KS, PASSES, F16>(bufX, cfrag[0], acc, lane);
            GNN_TICK(1)
            __syncthreads();                                                     // ---- B1
            GNN_TICK(2)
            conv_epilogue<F16>(bufY, acc, hw, lane);
            __syncthreads();                                                     // ---- B2
            GNN_TICK(3)
            acc_init_bias(acc, a.conv_b[1], hw, lane);
            gemm_tile<true, KS, PASSES, F16>(bufY, cfrag[1], acc, lane);
            GNN_TICK(4)
            __syncthreads();                                                     // ---- B3
            GNN_TICK(5)
            conv_epilogue<F16>(bufY, acc, hw, lane);
            __syncthreads();                                                     // ---- B4
            GNN_TICK(6)
            wv_pool<PASSES, F16>(bufY, vfrag[1], yp_w[1], t0, hw, lane);
        } else {
            if (step > 0)
                m_partials<F16>(bufY, a.weff[1], a.pos_sorted[1], t0 - FT, a.bucket_ptr[1][step - 1], a.bucket_ptr[1][step],
                           mp_w[1], hw, lane);
            m_partials<F16>(bufX, a.weff[0], a.pos_sorted[0], t0, a.bucket_ptr[0][step], a.bucket_ptr[0][step + 1], mp_w[0], hw, lane);
            uint4 carry = make_uint4(0, 0, 0, 0);
            const int cr = ht >> 5, cc = ht & 31;        // 5 rows x 32 chunks of 16 B (hi+lo = 512 B)
            if (ht < CARRY * 32) carry = *reinterpret_cast<const uint4*>(bufX + (FT + cr) * ROWB + cc * 16);
            GNN_TICK(8)
            __syncthreads();                                                     // ---- B1
            if (ht < CARRY * 32) *reinterpret_cast<uint4*>(bufX + cr * ROWB + cc * 16) = carry;
            // bufX is free from B1 on: the first part of the next step's gather runs while the matrix
            // waves are in their conv2 epilogue (no MFMA traffic to compete with), the rest beside conv3
            if (step + 1 < FSTEPS) conv1_gather<0, GNN_GATHER_EARLY, StoreSplit16<F16>>(bufX, toks, a.conv1_k, a.conv1_b, t0 + FT, ht);
            __syncthreads();                                                     // ---- B2
            if constexpr (PROF) tick_ = __builtin_readcyclecounter();
            if (step + 1 < FSTEPS) conv1_gather<GNN_GATHER_EARLY, FT / 8, StoreSplit16<F16>>(bufX, toks, a.conv1_k, a.conv1_b, t0 + FT, ht);
            if (ht < CARRY * 32) carry = *reinterpret_cast<const uint4*>(bufY + (FT + cr) * ROWB + cc * 16);
            GNN_TICK(9)
            __syncthreads();                                                     // ---- B3
            if (ht < CARRY * 32) *reinterpret_cast<uint4*>(bufY + cr * ROWB + cc * 16) = carry;
            __syncthreads();                                                     // ---- B4
            if constexpr (PROF) tick_ = __builtin_readcyclecounter();
        }
    }
    if (helper)
        m_partials<F16>(bufY, a.weff[1], a.pos_sorted[1], (FSTEPS - 1) * FT, a.bucket_ptr[1][FSTEPS - 1], a.bucket_ptr[1][FSTEPS],
                   mp_w[1], hw, lane);
    if constexpr (PROF) {
        if (tid == 0)
            for (int i = 0; i < 8; ++i) atomicAdd(a.cycles + i, cyc[i]);
        if (tid == 256)
            for (int i = 8; i < 10; ++i) atomicAdd(a.cycles + i, cyc[i]);
    }
}

// ---------------------------------------------------------------------------------- host side
static inline uint16_t bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// Wmat (K x N, row major) -> [kstep Kpad/16][nblk Npad/32][plane hi,lo][lane 64][8] bf16 in the operand
// layout of v_mfma_f32_32x32x16_bf16: lane l holds W[kstep*16 + (l>>5)*8 + e][nblk*32 + (l&31)];
// rows >= K and columns >= N are zero.
static uint16_t f16_bits(float f) {
    const _Float16 h = (_Float16)f;          // round to nearest even
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}
static float f16_value(uint16_t u) {
    _Float16 h;
    memcpy(&h, &u, 2);
    return (float)h;
}

std::vector<uint16_t> pack_frags(const float* wmat, int K, int N, bool f16) {
    const int ksteps = (K + 15) / 16, nblks = (N + 31) / 32;
    std::vector<uint16_t> out((size_t)ksteps * nblks * 2 * 64 * 8, 0);
    for (int ks = 0; ks < ksteps; ++ks)
        for (int nb = 0; nb < nblks; ++nb)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    const int k = ks * 16 + (l >> 5) * 8 + e, n = nb * 32 + (l & 31);
                    if (k >= K || n >= N) continue;
                    const float v = wmat[(size_t)k * N + n];
                    const uint16_t hi = f16 ? f16_bits(v) : bf16_rne(v);
                    const uint16_t lo = f16 ? f16_bits(v - f16_value(hi)) : bf16_rne(v - bf16_to_f32(hi));
                    const size_t base = ((size_t)(ks * nblks + nb) * 2) * 64 * 8;
                    out[base + (size_t)l * 8 + e] = hi;
                    out[base + 64 * 8 + (size_t)l * 8 + e] = lo;
                }
    return out;
}

template <typename Tp>
static int upload_vec(gnn_ctx* ctx, const std::vector<Tp>& v, Tp** dev) {
    void* p = nullptr;
    GNN_HIP(hipMalloc(&p, v.size() * sizeof(Tp)));
    ctx->owned.push_back(p);
    GNN_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(Tp), hipMemcpyHostToDevice));
    *dev = static_cast<Tp*>(p);
    return GNN_OK;
}

int pack_fused_weights(gnn_ctx* ctx, const gnn_weights* w) {
    DeviceWeights& d = ctx->w;
    int rc;
    const float* ck[2] = {w->conv2_kernel, w->conv3_kernel};
    const gnn_igloo_weights* ig[2] = {&w->igloo_a, &w->igloo_b};
    for (int i = 0; i < 2; ++i) {
        if ((rc = upload_vec(ctx, pack_frags(ck[i], KS * C, C), &d.conv_frag[i]))) return rc;
        if ((rc = upload_vec(ctx, pack_frags(ig[i]->w_v, C, C), &d.wv_frag[i]))) return rc;
        if ((rc = upload_vec(ctx, pack_frags(ig[i]->w_qk, NP, POOLED), &d.wqk_frag[i]))) return rc;
        if ((rc = upload_vec(ctx, pack_frags(ig[i]->w_qk, NP, POOLED, true), &d.wqk_frag_h[i]))) return rc;
        if ((rc = upload_vec(ctx, pack_frags(ck[i], KS * C, C, true), &d.conv_frag_h[i]))) return rc;
        if ((rc = upload_vec(ctx, pack_frags(ig[i]->w_v, C, C, true), &d.wv_frag_h[i]))) return rc;
    }
    return GNN_OK;
}

int launch_front_fused(gnn_ctx* ctx, const uint8_t* bases, int64_t n, int precision) {
    const DeviceWeights& d = ctx->w;
    const bool f16 = precision == GNN_PREC_F16X3;
    FusedArgs a;
    a.bases = bases;
    a.conv1_k = d.conv1_pairs;
    a.conv1_b = d.conv1_b;
    for (int i = 0; i < 2; ++i) {
        a.conv_frag[i] = reinterpret_cast<const uint4*>(f16 ? d.conv_frag_h[i] : d.conv_frag[i]);
        a.conv_b[i] = d.conv_b[i];
        a.wv_frag[i] = reinterpret_cast<const uint4*>(f16 ? d.wv_frag_h[i] : d.wv_frag[i]);
        a.weff[i] = d.weff_sorted[i];
        a.pos_sorted[i] = d.pos_sorted[i];
        a.bucket_ptr[i] = d.bucket_ptr[i];
    }
    a.mp = ctx->ws.mp;
    a.yp = ctx->ws.yp;
    a.cycles = ctx->phase_cycles;
    const bool prof = ctx->phase_cycles != nullptr;
    const dim3 grid((unsigned)n), block(512);
    if (f16) {
        if (prof) hipLaunchKernelGGL((fused_front_kernel<3, true, true>), grid, block, 0, ctx->stream, a);
        else hipLaunchKernelGGL((fused_front_kernel<3, false, true>), grid, block, 0, ctx->stream, a);
    } else if (precision == GNN_PREC_BF16X3) {
        if (prof) hipLaunchKernelGGL((fused_front_kernel<3, true, false>), grid, block, 0, ctx->stream, a);
        else hipLaunchKernelGGL((fused_front_kernel<3, false, false>), grid, block, 0, ctx->stream, a);
    } else {
        if (prof) hipLaunchKernelGGL((fused_front_kernel<1, true, false>), grid, block, 0, ctx->stream, a);
        else hipLaunchKernelGGL((fused_front_kernel<1, false, false>), grid, block, 0, ctx->stream, a);
    }
    GNN_HIP(hipGetLastError());
    return GNN_OK;
}

}  // namespace gnn
