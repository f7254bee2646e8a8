// Device-side pieces of the Toom-Cook front ends (gnn_fused_tc.hip: conv2 + conv3 on the matrix pipe; gnn_fused_tk.hip: conv3 only,
// x2 gathered from a table of all 14-mers): LDS geometry, the weight ring and the conv loop of the matrix waves, the y @ w_v tile,
// head A's 9-mer table (WvaTable), epilogues, the helpers' input transform, the conv1 gather and the IGLOO pair passes.  Moved out of
// gnn_fused_tc.hip in round 6 without a change (the device code of that kernel is the same to the byte).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "gnn_fused_helpers.h"

namespace gnn {
namespace tc {

constexpr int NMB = 3;                       // 32-row blocks per step (w_v tiles); one 32-tile block per transform point
constexpr int FTT = 32 * NMB;                // 96 rows per step
constexpr int STEPST = (T + FTT - 1) / FTT;  // 63
constexpr int NXI = 8;
constexpr int ROWX = 528, LOX = 256;
constexpr int BUF_ROWS = CARRY + FTT;        // 101
constexpr int BUF_BYTES = BUF_ROWS * ROWX;
constexpr int VRING_OFF = 2 * BUF_BYTES;
constexpr int VSLOT = NXI * 2 * 1024, VRING = 3;
constexpr int PROW_OFF = VRING_OFF + VRING * VSLOT;
constexpr int PROW_N = FTT + 4;
constexpr int PROW_BYTES = ((PROW_N * 2 + 15) / 16) * 16;
constexpr int BIAS_OFF = PROW_OFF + 2 * PROW_BYTES;
constexpr int LAST_OFF = BIAS_OFF + 2 * C * 4;
constexpr int SMEMT = LAST_OFF + 16;
constexpr int ROW_U4 = ROWX / 16;
constexpr int WNBLK_B = 2048;                // weight bytes per (unit, [xi,] n-block): hi fragment | lo fragment
constexpr int WUNIT_B = 4 * WNBLK_B;         // per k16 unit (w_v) / per (k16 unit, xi) (convs)
constexpr int RINGV = 8;                     // w_v tile: all 8 k16 units of its weights are loaded up front
constexpr int RINGT = 8;                     // convs: weight ring slots of one (unit, xi); 7 in flight ahead of the MFMAs
static_assert(SMEMT <= 160 * 1024, "LDS budget");
static_assert(RINGV <= RINGT, "the w_v tile's weights live in the conv loops' ring registers");
static_assert(T == 3 * 1999, "the tiles tile the window exactly");

struct Args {
    const uint8_t* bases;
    const float* conv1_k;             // pair tables in the gather's lane order (DeviceWeights::conv1_pairs6)
    const unsigned char* tcw[2];      // transformed conv weights: [k16 unit 8][xi 8][nblk 4][hi | lo] x 1 KiB, scaled by 1 / inv_s
    float inv_s[2];                   // power of two that A^T absorbs
    const float* conv_b[2];
    const unsigned char* wv_w[2];     // [k16 unit 8][nblk 4][hi | lo] x 1 KiB (pack_frags, f16 limbs)
    const float* weff[2];
    const int32_t* pos_sorted[2];
    const int32_t* bucket_ptr[2];     // (STEPST + 1,) entry ranges per 96-row step
    const unsigned char* wva_tbl;     // head A's y @ w_v per 9-mer (WvaTable below)
    float* mp;
    float* yp;
    const float* yp_c;                // outputs of an all-N window (padding skip), nullptr = compute everything
    const float* mp_c;
    unsigned long long* cycles;
    int split;
};

struct WU {
    uint4 h, l;
};

// -DTC_JITTER (test builds only, scripts/tc_jitter_check.py): every wave sleeps a pseudo-random time (0 .. ~2 000 cycles, a hash of wave,
// step-local counter and lane-uniform salt) behind every barrier.  Results must not change by a bit: a producer / consumer pair of LDS
// data that is not ordered by a barrier shows up as a mismatch against the normal build.
#ifdef TC_JITTER
__device__ __forceinline__ void tc_jitter(unsigned& state) {
    state = state * 1664525u + 1013904223u;
    const unsigned n = __builtin_amdgcn_readfirstlane((state >> 24) & 31u);
    for (unsigned i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
}
#define TC_JITTER_HERE() tc_jitter(jitter_state)
#else
#define TC_JITTER_HERE()
#endif
#define TC_BARRIER() asm volatile("s_barrier" ::: "memory"); TC_JITTER_HERE()
// The helpers outrank the matrix waves (priority 2) while the conv loops wait for their chunks, and yield beside the w_v tiles,
// where the matrix waves are the critical path and the helpers have time to spare.
#define TC_HPRIO_LOW() __builtin_amdgcn_s_setprio(1)
#define TC_HPRIO_HIGH() __builtin_amdgcn_s_setprio(3)
// helper-side barrier with the PROF counters around it: `work` collects the time since the last tick, `wait` the time in the barrier
#define HBAR_W(work, wait) GNN_TICK(work) TC_BARRIER_W(); GNN_TICK(wait)
#define HBAR(work, wait) GNN_TICK(work) TC_BARRIER(); GNN_TICK(wait)
// this wave stored to LDS since the last barrier: the stores must have landed before the others are released
#define TC_BARRIER_W() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); TC_JITTER_HERE()

__device__ __forceinline__ f32x16 mma(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
#ifndef TC_WAUX
#define TC_WAUX 0     // cache policy bits of the weight stream (bit 0 sc0, bit 1 nt, bit 4 sc1): A/B in profiles/r04/tc_ablation.txt
#endif
__device__ __forceinline__ void load_wu(WU& w, wrsrc_t r, uint32_t l16, int soff) {
    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, l16, soff, TC_WAUX);
    const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(r, l16 + 1024, soff, TC_WAUX);
    w.h = make_uint4(a[0], a[1], a[2], a[3]);
    w.l = make_uint4(b[0], b[1], b[2], b[3]);
}

// ---------------------------------------------------------------- IGLOO pair products: the 16 FMAs of one 16-byte slice
// (hi + lo) * w as two v_fma_mix_f32 per value (the f16 halves are read in place: no conversion, no addition; hi * w and lo * w are exact
// in f32 up to one rounding each, like (hi + lo) * w).  Four independent accumulators; slice i of a lane's 32-channel block uses the
// weights w[2 i], w[2 i + 1].  Shared by the helpers' PairCompute::run and the matrix waves' staged form below: the same instructions on
// the same operands in the same order per accumulator, so who computes an entry does not change a bit of it.
__device__ __forceinline__ void pair_fma16(const uint4& hx, const uint4& lx, const float4& wlo, const float4& whi, float& s0, float& s1, float& s2, float& s3) {
    const uint32_t hv[4] = {hx.x, hx.y, hx.z, hx.w}, lv[4] = {lx.x, lx.y, lx.z, lx.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float4 w0 = (k >> 1) ? whi : wlo;
        const float wa = (k & 1) ? w0.z : w0.x, wb = (k & 1) ? w0.w : w0.y;
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(s0) : "v"(hv[k]), "v"(wa));
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(s1) : "v"(lv[k]), "v"(wa));
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(s2) : "v"(hv[k]), "v"(wb));
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(s3) : "v"(lv[k]), "v"(wb));
    }
}


// ---------------------------------------------------------------- y @ w_v tiles (direct, 3 row blocks, as gnn_fused_x3.hip)
struct XU {
    uint4 h[NMB], l[NMB];
};
template <int OFF>
__device__ __forceinline__ void load_xu(XU& f, const unsigned char* __restrict__ xh) {
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
        f.h[mb] = *reinterpret_cast<const uint4*>(xh + OFF + mb * 32 * ROWX);
        f.l[mb] = *reinterpret_cast<const uint4*>(xh + OFF + LOX + mb * 32 * ROWX);
    }
}
// one k16 unit of the tile: 9 MFMAs (3 row blocks x 3 limb products), the next unit's six row reads interleaved 1:1 behind the first six
template <bool LX, int OFFN>
__device__ __forceinline__ void wv_unit(const WU& wc, const XU& xc, XU& xl, const unsigned char* __restrict__ xh, f32x16 (&acc)[NMB]) {
    if constexpr (LX) load_xu<OFFN>(xl, xh);
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
        if (mb % 2 == 0) {
            acc[mb] = mma(xc.h[mb], wc.l, acc[mb]);
            acc[mb] = mma(xc.h[mb], wc.h, acc[mb]);
            acc[mb] = mma(xc.l[mb], wc.h, acc[mb]);
        } else {
            acc[mb] = mma(xc.l[mb], wc.h, acc[mb]);
            acc[mb] = mma(xc.h[mb], wc.h, acc[mb]);
            acc[mb] = mma(xc.h[mb], wc.l, acc[mb]);
        }
    }
#pragma unroll
    for (int i = 0; i < 3 * NMB; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (LX && i < 2 * NMB) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    GNN_REGION_END();
}
// D = X W over the 96 rows that start at buffer row CARRY of `xoff`: a lane ends up with 16 rows of one channel per row block.
// All 8 weight units are in the ring (prime_wv, requested before the conv3 epilogue): the tile issues no memory request - it runs
// while head A's table rows of the next step travel (requested right in front of it), and a wave's request behind 96 missing lines
// waits at issue until the vector L1 has room.
__device__ __forceinline__ void wv_tile(const unsigned char* __restrict__ smem, int xoff, const WU (&ring)[RINGT], f32x16 (&acc)[NMB], int lane) {
    uint32_t rowoff = (uint32_t)xoff + (uint32_t)(lane & 31) * ROWX + (uint32_t)(lane >> 5) * 16u;
    asm volatile("" : "+v"(rowoff));
    const unsigned char* xh = smem + rowoff;
    XU xa, xb;
    load_xu<0>(xa, xh);
    GNN_REGION_END();
    static_for(std::make_integer_sequence<int, 8>{}, [&](auto kc) {
        constexpr int k = decltype(kc)::value, kn = k + 1;
        constexpr int OFFN = kn * 32;
        constexpr bool LX = kn < 8;
        if constexpr (k % 2 == 0) wv_unit<LX, OFFN>(ring[k], xa, xb, xh, acc);
        else wv_unit<LX, OFFN>(ring[k], xb, xa, xh, acc);
    });
}
__device__ __forceinline__ void prime_wv(WU (&ring)[RINGT], wrsrc_t wr, int lane) {
    const uint32_t l16 = (uint32_t)lane * 16u;
#pragma unroll
    for (int u = 0; u < RINGV; ++u) load_wu(ring[u], wr, l16, u * WUNIT_B);
    asm volatile("" ::: "memory");
}

// MaxPool1D(8) of the y @ w_v tile -> yp rows (igloo.py:209-210); gnn_fused_x3.hip, wv_pool_store
template <int AUX = 0>      // cache policy bits of the stores (bit 1 = nt: the rows are not read again by this kernel)
__device__ __forceinline__ void wv_pool_store(const f32x16 (&acc)[NMB], wrsrc_t yp_w, int head_off, int t0, int wave, int lane) {
    float m[4 * NMB];
#pragma unroll
    for (int i = 0; i < 4 * NMB; ++i) {
        const int mb = i >> 2, rg = i & 3;
        const float v = max_nan(max_nan(acc[mb][rg * 4], acc[mb][rg * 4 + 1]), max_nan(acc[mb][rg * 4 + 2], acc[mb][rg * 4 + 3]));
        const unsigned bits = __float_as_uint(v);
        const auto sw = __builtin_amdgcn_permlane32_swap(bits, bits, false, false);
        m[i] = max_nan(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    const int q0 = t0 / GNN_POOL;
    const int nq = min(4 * NMB, POOLED - q0);
    if (lane < 32) {
        const uint32_t voff = (uint32_t)(wave * 32 + lane) * 4u;
#pragma unroll
        for (int i = 0; i < 4 * NMB; ++i)
            if (i < nq) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m[i]), yp_w, voff, head_off + (q0 + i) * (C * 4), AUX);
    }
}

// ---------------------------------------------------------------- head A's y @ w_v as a table lookup (round 6)
// x1[t] = LeakyReLU(conv1) (model.py:11, igloo.py:45-48) is a function of the tokens t-5 .. t, i.e. of the NINE bases t-5 .. t+3, so
// head A's y @ w_v row (igloo.py:208) is too: the matrix waves no longer compute it (288 of a step's 2 112 MFMAs per CU, 8 % of the
// launch's energy, profiles/r05/MODEL.md) but gather it - one 512-byte row per position - from tables that gnn_load_weights builds on
// the device (x1 exactly as the gather below makes it, the 128 x 128 product accumulated in f64 and rounded once: closer to the
// reference's f32 than the three f16 products were) and take the 8-row maximum (igloo.py:209-210) in registers:
//   D4  4^9 rows     all nine bases in ACGT, t >= 5: index = the 9-mer, first base most significant (128 MiB: the rows every window reads)
//   S5  5^4 .. 5^8   the first five positions of a window (tokens before the window start are absent, not N): bases 0 .. t+3 in base 5
//   D5  5^9 rows     t >= 5 with a non-ACGT base among the nine (digit 4)
// One allocation [D4 | S5 | D5] of 2 703 394 rows = 1.38 GB, one buffer resource; a window's 5 992 pooled positions read 3.07 MB of it.
struct WvaTable {
    static constexpr uint32_t D4_ROWS = 262144u, S5_ROWS = 625u + 3125u + 15625u + 78125u + 390625u, D5_ROWS = 1953125u;
    static constexpr uint32_t S5_OFF = D4_ROWS, D5_OFF = D4_ROWS + S5_ROWS, ROWS = D4_ROWS + S5_ROWS + D5_ROWS;
    static constexpr uint32_t ROW_BYTES = C * 4;
    __host__ __device__ static constexpr uint32_t s5_off(int t) { return S5_OFF + (t == 0 ? 0u : t == 1 ? 625u : t == 2 ? 3750u : t == 3 ? 19375u : 97500u); }
};
constexpr int WVA_ROWS_PER_WAVE = FTT / 4;    // 24 rows = 3 pooled rows per matrix wave and step
struct WvaBytes {
    uint32_t x0, x1, x2;       // the 12 aligned bytes that hold bases t-5 .. t+3 of the lane's row
};
// lane l < 24 of matrix wave hw owns row t0 + 24 hw + l; rows past the last token (the last step's tail) are never pooled: clamped
__device__ __forceinline__ int wva_row(int t0, int hw, int lane) { return min(t0 + WVA_ROWS_PER_WAVE * hw + min(lane, WVA_ROWS_PER_WAVE - 1), T - 1); }
__device__ __forceinline__ void wva_fetch(WvaBytes& b, const uint8_t* __restrict__ bases, int t) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(bases + (max(t - 5, 0) & ~3));      // <= W - 12: windows start 4-byte aligned
    b.x0 = src[0];
    b.x1 = src[1];
    b.x2 = src[2];
}
// table row of position t.  Branch-free: digit of a byte = ((b >> 1) & 3) ^ ((b >> 2) & 1) for A, C, G, T (65, 67, 71, 84 -> 0, 1, 2, 3:
// the order of sequence.py:170-193), 4 for every other byte
__device__ __forceinline__ uint32_t wva_digit(uint32_t byte) {
    const uint32_t x = byte - 65u, c = (byte >> 1) & 3u;
    const bool acgt = x < 20u && ((0x80045u >> (x & 31u)) & 1u);
    return acgt ? (c ^ (c >> 1)) : 4u;
}
__device__ __forceinline__ uint32_t wva_index(const WvaBytes& b, int t) {
    const int q = max(t - 5, 0), sh = q & 3, np = min(t, 5) + 4;            // np bases are present: all 9 from position 5 on
    const uint32_t w0 = __builtin_amdgcn_alignbyte(b.x1, b.x0, (uint32_t)sh), w1 = __builtin_amdgcn_alignbyte(b.x2, b.x1, (uint32_t)sh),
                   w2 = b.x2 >> (8 * sh);
    uint32_t i4 = 0, i5 = 0, worst = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const uint32_t d = wva_digit(((k < 4 ? w0 : k < 8 ? w1 : w2) >> (8 * (k & 3))) & 255u);
        i4 = i4 * 4u + (d & 3u);
        const uint32_t n5 = i5 * 5u + d;
        i5 = k < np ? n5 : i5;
        worst = max(worst, d);
    }
    // t < 5: the bytes behind the np present ones are ordinary bases of the window; they only decide `worst`, which is not used then
    return t < 5 ? WvaTable::s5_off(t) + i5 : (worst < 4u ? i4 : WvaTable::D5_OFF + i5);
}
// Row indices of a step for a matrix wave (lane l < 24: row 24 hw + l).  The pair rows the helpers keep in LDS for the conv1 gather
// already hold the 9-mer: prow[r] is the 5-mer of the bases t-5 .. t-1 and prow[r + 4] that of t-1 .. t+3 whenever both are < 1024 (all
// nine bases in ACGT, none before the window start) - two LDS reads and three integer instructions.  Only a wave that sees another
// pair row (a non-ACGT base, the first five positions of a window) reads the bases themselves and walks wva_index; that round trip
// is exposed, on the few steps that have one.
__device__ __forceinline__ uint32_t wva_step_index(const uint16_t* __restrict__ prow, const uint8_t* __restrict__ bases, int t0, int hw, int lane) {
    const int r = WVA_ROWS_PER_WAVE * hw + min(lane, WVA_ROWS_PER_WAVE - 1);
    const uint32_t p0 = prow[r], p4 = prow[r + 4];
    uint32_t idx = (p0 << 8) | (p4 & 255u);
    if (__builtin_amdgcn_ballot_w64(p0 >= 1024u || p4 >= 1024u)) {
        const int t = wva_row(t0, hw, lane);
        WvaBytes b;
        wva_fetch(b, bases, t);
        idx = wva_index(b, t);
    }
    return idx;
}
struct WvaRows {
    u32x4 v[WVA_ROWS_PER_WAVE / 2];     // 24 table rows: request i holds row 2 i in lanes 0..31 and row 2 i + 1 in lanes 32..63, four channels per lane
};
// all 24 rows of a wave in 12 requests of 64 lanes x 16 B (two 512-byte rows each): behind E the vector memory pipe carries the w_v
// weights, these rows, the helpers' x2 rows and pair weights, and a request of 16 B per lane costs what one of 8 B does (one row per
// request, 24 of them: w_v B + pool 5.5 k -> 4.4 k cycles per step, profiles/r06/ab/ab14_wva_rows_16B.txt)
template <int I0 = 0, int I1 = WVA_ROWS_PER_WAVE>
__device__ __forceinline__ void wva_issue(WvaRows& r, wrsrc_t tbl, uint32_t my_row, int lane) {
    static_assert(I0 % 2 == 0 && I1 % 2 == 0, "two rows per request");
    const uint32_t l16 = (uint32_t)(lane & 31) * 16u;
#pragma unroll
    for (int i = I0 / 2; i < I1 / 2; ++i) {
        const uint32_t r0 = __builtin_amdgcn_readlane(my_row, 2 * i), r1 = __builtin_amdgcn_readlane(my_row, 2 * i + 1);
        r.v[i] = __builtin_amdgcn_raw_buffer_load_b128(tbl, (lane < 32 ? r0 : r1) * WvaTable::ROW_BYTES + l16, 0, 0);
    }
}
// MaxPool1D(8) over the gathered rows -> 3 pooled rows of yp (igloo.py:209-210): the even rows' maximum in lanes 0..31, the odd rows' in
// lanes 32..63 (max is exact and commutative: any order gives the same bits), one half-swap, lanes 0..31 store the row
template <int AUX = 0>
__device__ __forceinline__ void wva_pool_store(const WvaRows& r, wrsrc_t yp_w, int t0, int hw, int lane) {
    const int q0 = t0 / GNN_POOL + 3 * hw;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        u32x4 out;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float m = __uint_as_float(r.v[4 * p][c]);
#pragma unroll
            for (int j = 1; j < 4; ++j) m = max_nan(m, __uint_as_float(r.v[4 * p + j][c]));
            const unsigned bits = __float_as_uint(m);
            const auto sw = __builtin_amdgcn_permlane32_swap(bits, bits, false, false);
            out[c] = __float_as_uint(max_nan(__uint_as_float(sw[0]), __uint_as_float(sw[1])));
        }
        if (lane < 32 && q0 + p < POOLED) __builtin_amdgcn_raw_buffer_store_b128(out, yp_w, (uint32_t)lane * 16u, (q0 + p) * (C * 4), AUX);
    }
}

// ---------------------------------------------------------------- Toom-Cook conv: the matrix waves' side
struct XV {
    uint4 h, l;
};
template <int VOFFN>
__device__ __forceinline__ void xi_mma(const WU& wc, WU& wl, const XV& vc, XV& vl, const unsigned char* __restrict__ vb, wrsrc_t wr, int wnext,
                                       uint32_t l16, f32x16& acc) {
    vl.h = *reinterpret_cast<const uint4*>(vb + VOFFN);
    vl.l = *reinterpret_cast<const uint4*>(vb + VOFFN + 1024);
    load_wu(wl, wr, l16, wnext);
    acc = mma(wc.l, vc.h, acc);            // D = U^T V: a lane ends up with 16 channels of one tile
    acc = mma(wc.h, vc.h, acc);
    acc = mma(wc.h, vc.l, acc);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    GNN_REGION_END();
}
__device__ __forceinline__ void prime_tc(WU (&ring)[RINGT], wrsrc_t wr, int woff, int lane) {
    const uint32_t l16 = (uint32_t)lane * 16u;
#pragma unroll
    for (int u = 0; u < RINGT - 1; ++u) load_wu(ring[u], wr, l16, woff + u * WUNIT_B);
    asm volatile("" ::: "memory");
}
// 8 k16 units x 8 points; barrier b_c in front of unit c (the helpers' chunk c + 1 is complete, the slot of chunk c - 1 is free).
// The weight requests of the last 7 (unit, xi) wrap onto the conv's first ones (in-bounds, unused).
__device__ __forceinline__ void conv_tc(const unsigned char* __restrict__ smem, wrsrc_t wr, int woff, WU (&ring)[RINGT], f32x16 (&acc)[NXI],
                                        int lane, unsigned& jitter_state) {
    const uint32_t l16 = (uint32_t)lane * 16u;
    uint32_t voff = (uint32_t)VRING_OFF + l16;
    asm volatile("" : "+v"(voff));
    const unsigned char* vb = smem + voff;
    XV va, vc;
    TC_BARRIER();                                                            // b_0
    va.h = *reinterpret_cast<const uint4*>(vb);
    va.l = *reinterpret_cast<const uint4*>(vb + 1024);
    vc = va;
    GNN_REGION_END();
    static_for(std::make_integer_sequence<int, 64>{}, [&](auto kc) {
        constexpr int k = decltype(kc)::value, kn = (k + 1) % 64;             // k = unit * 8 + xi
        constexpr int VOFFN = ((kn / 8) % VRING) * VSLOT + (kn % 8) * 2048;
        constexpr int kw = (k + RINGT - 1) % 64;
        if constexpr (k % 8 == 0 && k > 0) TC_BARRIER();                      // b_1 .. b_7
        if constexpr (k % 2 == 0)
            xi_mma<VOFFN>(ring[k % RINGT], ring[(k + RINGT - 1) % RINGT], va, vc, vb, wr, woff + kw * WUNIT_B, l16, acc[k % 8]);
        else
            xi_mma<VOFFN>(ring[k % RINGT], ring[(k + RINGT - 1) % RINGT], vc, va, vb, wr, woff + kw * WUNIT_B, l16, acc[k % 8]);
    });
}

template <bool F16>
__device__ __forceinline__ void split2(f32x2 v, uint32_t& hi, uint32_t& lo) {
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{sub_f16_lo(v[0], hi), sub_f16_hi(v[1], hi)}, f16x2));
}

// A^T of F(3,6) (oracle/toomcook.py) on TWO neighbouring accumulator registers of the 8 points at a time, then scale, bias,
// LeakyReLU:
//   y0 = m0 + (m1 + m2) + (m3 + m4) + (m5 + m6);  y1 = (m1 - m2) + 2 (m3 - m4) + (m5 - m6) / 2;
//   y2 = (m1 + m2) + 4 (m3 + m4) + (m5 + m6) / 4 + m7
// The epilogues run while no MFMA is in flight on the SIMD, where the packed f32 forms (v_pk_add / v_pk_fma / v_pk_mul_f32) issue
// at full rate: 26 instead of 46 instructions per register pair (the file is compiled without SLP packing - the helpers' transform
// runs beside the MFMA stream, where packed f32 is an anti-lever -, so the pairs are spelled out with 2-vectors here).
__device__ __forceinline__ f32x2 pair_of(const f32x16& a, int r) { return f32x2{a[r], a[r + 1]}; }
__device__ __forceinline__ void inverse3(const f32x16 (&acc)[NXI], int r, float inv_s, f32x2 bias, f32x2 (&y)[3]) {
    const f32x2 m0 = pair_of(acc[0], r), m1 = pair_of(acc[1], r), m2 = pair_of(acc[2], r), m3 = pair_of(acc[3], r);
    const f32x2 m4 = pair_of(acc[4], r), m5 = pair_of(acc[5], r), m6 = pair_of(acc[6], r), m7 = pair_of(acc[7], r);
    const f32x2 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4, s56 = m5 + m6, d56 = m5 - m6;
    const f32x2 y0 = ((m0 + s12) + s34) + s56;
    const f32x2 y1 = __builtin_elementwise_fma(d56, f32x2{0.5f, 0.5f}, __builtin_elementwise_fma(d34, f32x2{2.f, 2.f}, d12));
    const f32x2 y2 = __builtin_elementwise_fma(s56, f32x2{0.25f, 0.25f}, __builtin_elementwise_fma(s34, f32x2{4.f, 4.f}, s12)) + m7;
    const f32x2 sc = {inv_s, inv_s}, lr = {LRELU, LRELU};
    const f32x2 v0 = __builtin_elementwise_fma(y0, sc, bias), v1 = __builtin_elementwise_fma(y1, sc, bias), v2 = __builtin_elementwise_fma(y2, sc, bias);
    const f32x2 w0 = v0 * lr, w1 = v1 * lr, w2 = v2 * lr;
    y[0] = f32x2{vmax_raw(v0[0], w0[0]), vmax_raw(v0[1], w0[1])};
    y[1] = f32x2{vmax_raw(v1[0], w1[0]), vmax_raw(v1[1], w1[1])};
    y[2] = f32x2{vmax_raw(v2[0], w2[0]), vmax_raw(v2[1], w2[1])};
}
// conv2 epilogue: x2 rows as f32 (only the input transform of conv3 reads them).  Register r of lane l = channel
// 8 (r >> 2) + 4 (l >> 5) + (r & 3) of tile l & 31: one 16-B store per (row of the tile, register group)
__device__ __forceinline__ void epilogue_f32(unsigned char* __restrict__ obuf, const f32x16 (&acc)[NXI], float inv_s,
                                             const float* __restrict__ bias, int wave, int lane) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const int f0 = wave * 32 + rg * 8 + (lane >> 5) * 4;
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias + f0);
        f32x2 ya[3], yb[3];
        inverse3(acc, rg * 4, inv_s, f32x2{b[0], b[1]}, ya);
        inverse3(acc, rg * 4 + 2, inv_s, f32x2{b[2], b[3]}, yb);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            *reinterpret_cast<f32x4*>(obuf + (CARRY + 3 * (lane & 31) + i) * ROWX + f0 * 4) = f32x4{ya[i][0], ya[i][1], yb[i][0], yb[i][1]};
    }
}
// conv3 epilogue: x3 rows as hi | lo planes (y @ w_v of head B and its pair products read them)
__device__ __forceinline__ void epilogue_x3(unsigned char* __restrict__ obuf, const f32x16 (&acc)[NXI], float inv_s, const float* __restrict__ bias,
                                            int wave, int lane) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const int f0 = wave * 32 + rg * 8 + (lane >> 5) * 4;
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias + f0);
        f32x2 ya[3], yb[3];
        inverse3(acc, rg * 4, inv_s, f32x2{b[0], b[1]}, ya);
        inverse3(acc, rg * 4 + 2, inv_s, f32x2{b[2], b[3]}, yb);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            uint2 h, l;
            split2<true>(ya[i], h.x, l.x);
            split2<true>(yb[i], h.y, l.y);
            unsigned char* o = obuf + (CARRY + 3 * (lane & 31) + i) * ROWX + f0 * 2;
            *reinterpret_cast<uint2*>(o) = h;
            *reinterpret_cast<uint2*>(o + LOX) = l;
        }
    }
}

// ---------------------------------------------------------------- Toom-Cook conv: the helper waves' side
// One helper wave = 16 (tile, k half) combinations x 4 channel pairs of a k16 unit: the 4 lanes of a combination cover its 8
// channels = the 16 bytes of one B-fragment lane, so the row reads (4 lanes = 16 consecutive bytes, tiles 1 584 B apart) and the
// fragment stores (a wave = 256 consecutive bytes per point and limb) are bank-conflict free.
struct HLane {
    const unsigned char* rows;    // first input row of the lane's tile (buffer row 3 * tile), at the lane's channel pair
    unsigned char* frag;          // the lane's dword of the ring slot's fragments
};
__device__ __forceinline__ HLane hlane(unsigned char* smem, int buf_off, int hw, int lane) {
    const int pr = lane & 3, th = hw * 16 + (lane >> 2), tile = th & 31, half = th >> 5;
    HLane h;
    h.rows = smem + buf_off + (3 * tile) * ROWX + (half * 8 + pr * 2) * 4;
    h.frag = smem + VRING_OFF + (hw * 64 + lane) * 4;
    return h;
}
struct Raw16 {   // 8 rows x 2 channels as stored: f32 pairs (x1 and x2 rows alike since round 6)
    uint32_t a[8], b[8];
};
__device__ __forceinline__ void load_rows(Raw16& r, const HLane& h, int unit) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint2 v = *reinterpret_cast<const uint2*>(h.rows + j * ROWX + unit * 64);
        r.a[j] = v.x;
        r.b[j] = v.y;
    }
}
// B^T of F(3,6) on the 8 rows of one channel (oracle/toomcook.py: rows of B^T in this order)
__device__ __forceinline__ void bt8(const float (&d)[8], float (&v)[8]) {
    v[0] = fmaf(d[2] - d[4], 5.25f, d[6] - d[0]);
    const float t1 = fmaf(d[4], -4.25f, d[2] + d[6]), t2 = fmaf(d[3], -4.25f, d[1] + d[5]);
    v[1] = t1 + t2;
    v[2] = t1 - t2;
    const float t3 = fmaf(d[4], -1.25f, fmaf(d[2], 0.25f, d[6])), t4 = fmaf(d[5], 2.f, fmaf(d[3], -2.5f, d[1] * 0.5f));
    v[3] = t3 + t4;
    v[4] = t3 - t4;
    const float t5 = fmaf(d[4], -5.f, fmaf(d[2], 4.f, d[6])), t6 = fmaf(d[5], 0.5f, fmaf(d[3], -2.5f, d[1] * 2.f));
    v[5] = t5 + t6;
    v[6] = t5 - t6;
    v[7] = fmaf(d[3] - d[5], 5.25f, d[7] - d[1]);
}
__device__ __forceinline__ void transform_store(const Raw16& r, const HLane& h, int slot) {
    float d0[8], d1[8], v0[8], v1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        d0[j] = __uint_as_float(r.a[j]);
        d1[j] = __uint_as_float(r.b[j]);
    }
    bt8(d0, v0);
    bt8(d1, v1);
    unsigned char* o = h.frag + slot * VSLOT;
#pragma unroll
    for (int xi = 0; xi < NXI; ++xi) {
        uint32_t hi, lo;
        split2<true>(f32x2{v0[xi], v1[xi]}, hi, lo);
        *reinterpret_cast<uint32_t*>(o + xi * 2048) = hi;
        *reinterpret_cast<uint32_t*>(o + xi * 2048 + 1024) = lo;
    }
}

// conv1 gather (model.py:11 + igloo.py:45-48 on the pair tables of gnn_load_weights): one lane = 16 consecutive channels of ONE row,
// the 8 lanes of a row read one 128-B line per load; 96 rows x 8 = 768 items = exactly 3 per helper lane and step, so the four helper
// waves carry the same load (the lane-pair scheme of gnn_fused_x3.hip handles 64 rows per round: 1.5 rounds here, two of them on two
// of the four waves, and 48 lane-parity selects + 16 DPP moves per item that the single-row form does not need).
struct GRow {
    f32x4 v[3][4];      // [table][i]: channels 16 pq + 4 i ..
};
__device__ __forceinline__ void grow_issue(GRow& g, const uint16_t* __restrict__ prow, const float* __restrict__ pt, int row, int pq) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const uint32_t r = prow[row + 2 * j];
        const float* src = pt + ((size_t)j * PAIR_ROWS + r) * C + pq * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) g.v[j][i] = *reinterpret_cast<const f32x4*>(src + i * 32);
    }
}
struct GOut {         // a finished item: 16 channels of one x1 row (f32), waiting for bufX to become writable
    f32x4 v[4];       // channels 16 pq + 4 i ..
};
__device__ __forceinline__ void grow_compute(GOut& o, const GRow& g) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 t = g.v[0][i] + g.v[1][i] + g.v[2][i];           // the bias is folded into table 0
        o.v[i] = f32x4{vmax_raw(t[0], t[0] * LRELU), vmax_raw(t[1], t[1] * LRELU), vmax_raw(t[2], t[2] * LRELU), vmax_raw(t[3], t[3] * LRELU)};
    }
}
__device__ __forceinline__ void grow_store(const GOut& o, unsigned char* __restrict__ xbuf, int row, int pq) {
    unsigned char* d = xbuf + (CARRY + row) * ROWX + pq * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(d + i * 16) = o.v[i];
}
__device__ __forceinline__ void grow_finish(const GRow& g, unsigned char* __restrict__ xbuf, int row, int pq) {
    GOut o;
    grow_compute(o, g);
    grow_store(o, xbuf, row, pq);
}

// dot product of an entry's 32 folded weights with block p of row u, summed over the entry's 4 lanes.  Two row formats: x3 (head B) is
// stored as f16 hi | lo planes (the y @ w_v tile's MFMA operands; x = hi + lo), x1 (head A) as f32 since round 6 (nothing multiplies it on
// the matrix pipe any more: conv2 reads it through the input transform, head A's y @ w_v comes from the table)
struct PairCompute {
    static __device__ __forceinline__ void run(const PairW& w, const PairJob& jb, int e, int u, int p) {
        const unsigned char* xr = jb.xbuf + (CARRY + u - jb.t0) * ROWX + p * 64;
        uint4 hx[4], lx[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            hx[i] = *reinterpret_cast<const uint4*>(xr + i * 16);
            lx[i] = *reinterpret_cast<const uint4*>(xr + LOX + i * 16);
        }
        // four independent accumulators: one chain of 64 dependent FMAs is latency-bound on a wave that has the SIMD's leftover issue slots
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) pair_fma16(hx[i], lx[i], w.w[2 * i], w.w[2 * i + 1], s0, s1, s2, s3);
        float s = (s0 + s2) + (s1 + s3);
        s += dpp_xor1(s);
        s += dpp_xor2(s);
        if (p == 0) jb.mp[e] = s;
    }
};
struct PairComputeF32 {
    static __device__ __forceinline__ void run(const PairW& w, const PairJob& jb, int e, int u, int p) {
        const unsigned char* xr = jb.xbuf + (CARRY + u - jb.t0) * ROWX + p * 128;
        f32x4 x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = *reinterpret_cast<const f32x4*>(xr + i * 16);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            s0 = fmaf(x[i][0], w.w[i].x, s0);
            s1 = fmaf(x[i][1], w.w[i].y, s1);
            s2 = fmaf(x[i][2], w.w[i].z, s2);
            s3 = fmaf(x[i][3], w.w[i].w, s3);
        }
        float s = (s0 + s2) + (s1 + s3);
        s += dpp_xor1(s);
        s += dpp_xor2(s);
        if (p == 0) jb.mp[e] = s;
    }
};

// Pair products of one head and step, one PASS (64 entries: 4 lanes per entry, 16 entries per wave) at a time, the loads of a pass
// requested long before they are used: a step holds ~134 entries per head = 3 passes, and the pass loop of gnn_fused_helpers.h
// (weights one pass ahead) would expose an L2 round trip per head and step here, where the helpers are the critical path.
struct PairPass {
    PairW w;
    int u;
};
__device__ __forceinline__ void pass_issue(PairPass& pp, const PairJob& jb, int k, int wave, int lane) {
    const int e = jb.e + wave * 16 + (lane >> 2) + 64 * k;
    pp.u = jb.t0;
    if (e < jb.e_end) {
        pp.u = jb.pos[e];
        pair_load_w(pp.w, jb, e, lane & 3);
    }
}
template <class Compute = PairCompute>
__device__ __forceinline__ void pass_compute(const PairPass& pp, const PairJob& jb, int k, int wave, int lane) {
    const int e = jb.e + wave * 16 + (lane >> 2) + 64 * k;
    if (e < jb.e_end) Compute::run(pp.w, jb, e, pp.u, lane & 3);
    GNN_REGION_END();        // keeps the scheduler from hoisting the next pass's 8 row reads (32 registers) above this pass
}
// a crowded step (more than 3 passes; rare): the remaining passes one by one, loads not hidden
template <class Compute = PairCompute>
__device__ __forceinline__ void pass_rest(PairPass& pp, const PairJob& jb, int k0, int wave, int lane) {
    for (int e = jb.e + wave * 16 + (lane >> 2) + 64 * k0; e < jb.e_end; e += 64) {
        pair_load_w(pp.w, jb, e, lane & 3);
        Compute::run(pp.w, jb, e, jb.pos[e], lane & 3);
    }
}


}  // namespace tc
}  // namespace gnn
