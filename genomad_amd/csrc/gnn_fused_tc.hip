// Fused front end, GNN_PREC_F16X3TC (the default since round 4): the f16x3 arithmetic of gnn_fused_x3.hip with conv2 and conv3
// (igloo.py:65-67) evaluated by Toom-Cook minimal filtering F(3,6) over the time axis - VERDICT r03 item 1.
//
//   y[3t + i] = sum_xi AT[i][xi] * M_xi[t],   M_xi[t] = sum_c V_xi[t][c] * U_xi[c][n],   V_xi[t] = sum_j BT[xi][j] x[3t - 5 + j],
//   U_xi = s * sum_k G[xi][k] w[k]            (oracle/toomcook.py: exact matrices; points 0, +-1, +-2, +-1/2, inf)
//
// so a step of 96 rows = 32 tiles = ONE 32-column MFMA block per transform point: 8 GEMMs of K = 128 instead of 6 taps x 3 row
// blocks of K = 128: 192 MFMAs per wave, conv and step instead of 432 (0.444x).  U and V are split into f16 hi / lo limbs and
// multiplied with the three products of the f16x3 arithmetic; transforms and accumulators are f32.
//
// What it costs (measured before it was built: scripts/probe_tc_loop.hip, profiles/r04/): a transformed weight fragment feeds 3
// MFMAs (one tile block) where a direct one feeds 12 (four row blocks), so the L2 -> CU weight stream is 4x denser per MFMA
// (the conv loops run at the chip's L2 ceiling, ~57 B/clk/CU), and the input transform is ~100 VALU instructions per (tile, 2
// channels) on the helper waves, which get ~6 issue slots per MFMA of the matrix wave they share a SIMD with.  Result: 21.7 vs
// 26.7 ms per 4096 windows against the direct form (gnn_fused_x3.hip) on one box; DESIGN.md section 4.0 has the accounting.
//
// Round 6: head A's y @ w_v (igloo.py:208 on x1) is no longer computed.  x1[t] is a function of the nine bases t-5 .. t+3, so the
// row is gathered from a table of all 9-mers (WvaTable below: built on the device by gnn_load_weights, f64 accumulation) and the
// 8-row max is taken in registers: 288 of a step's 2 112 MFMAs per CU and their weight stream are gone, and since nothing multiplies
// x1 on the matrix pipe any more it is stored as f32 rows (no limb split in the gather, no reconstruction in conv2's input transform,
// half the FMAs in head A's pair products).
//
// Structure: the streaming structure of gnn_fused_x3.hip (one workgroup = one window, both activation buffers in LDS with 5 carry
// rows, 4 matrix waves + 4 helper waves) with steps of 96 rows and a ring of 3 x 16 KB in LDS through which the transformed
// activations of ONE k16 unit (8 points x hi | lo x 64 lanes x 16 B, MFMA B-fragment order) reach the matrix waves:
//
//   matrix : [b0 it0 | b1 it1 | ... | b7 it7] conv2 -> inverse transform -> x2 (f32 rows) -> bufY | B1 | head A's 24 table rows per
//            wave requested, V3 chunk 1 [bufY], next step's row indices, 8-row max of the table rows -> yp A |
//            [b0' .. b7'] conv3 -> inverse transform -> x3 (hi | lo rows) -> bufY | B0 | w_v B(s) [bufY], V2(s+1) chunk 1 [bufX] -> step s+1
//   helpers: beside conv2 units 0..5: V2 chunks 2..7 [bufX]; beside units 6, 7 and the conv2 epilogue: head A's pair products [bufX],
//            gather round 0 of x1(s+1) (into registers) | B1 | V3 chunk 0 [bufY], head A's last pass, gather round 1 (registers) |
//            beside conv3 units 0..5: V3 chunks 2..7, the held gather rounds -> bufX; units 6, 7 and the conv3 epilogue: gather round 2,
//            carry rows, pair rows | B0 | head B's pair products [bufY], V2(s+1) chunk 0
//   (tests/test_kernel_schedule.py is an executable model of this schedule: every LDS producer / consumer pair is ordered by a barrier)
//
// 18 workgroup barriers per step (bare s_barrier: a __syncthreads() would drain the matrix waves' weight loads in flight).  The
// helpers run two chunks ahead of the matrix waves: before barrier b_c chunks <= c + 1 are complete, during unit c chunk c + 2
// is written into slot (c + 2) % 3, which the matrix waves read last in unit c - 1.  Chunks 0 and 1 of a conv are made in the
// interval in front of its loop, chunk 0 by the helpers and chunk 1 by the matrix waves (same lane mapping: wave w of either role
// owns the same 16 (tile, k half) combinations), so that neither role waits for the other there.
//
// LDS: bufX 101 rows x 528 B (x1 as f32 rows), bufY 101 x 528 (x2 as f32 rows, then x3 as hi | lo planes), ring 3 x 16 KB,
// pair rows, biases: 157.3 KB.
//
// Compile-time switches: -DTC_JITTER only (libgenomad_nn_hip_jitter.so, a test build: random sleeps behind every barrier, results
// must not move).  The ablation / probe / emulation switches of rounds 4 and 5 (TC_ABL_*, TC_PROBE_*, TC_EMU_ONEBUF, TC_PAIRS_MATRIX,
// ..., several of them wrong by construction) were removed from this file in round 6 (scripts/strip_switches.py; the device code of
// the default build did not change by a byte); the builds behind profiles/r04 and profiles/r05 are those of commit ad410a6.
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
            if (i < nq) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m[i]), yp_w, voff, head_off + (q0 + i) * (C * 4), 0);
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
    u32x2 v[WVA_ROWS_PER_WAVE];     // 24 table rows, two channels per lane
};
// all 24 row requests of a wave: 64 lanes x 8 B = one 512-byte row per instruction, the row's byte offset in an SGPR
template <int I0 = 0, int I1 = WVA_ROWS_PER_WAVE>
__device__ __forceinline__ void wva_issue(WvaRows& r, wrsrc_t tbl, uint32_t my_row, int lane) {
    const uint32_t l8 = (uint32_t)lane * 8u;
#pragma unroll
    for (int i = I0; i < I1; ++i) {
        const uint32_t row = __builtin_amdgcn_readlane(my_row, i);
        r.v[i] = __builtin_amdgcn_raw_buffer_load_b64(tbl, l8, row * WvaTable::ROW_BYTES, 0);
    }
}
// MaxPool1D(8) over the gathered rows -> 3 pooled rows of yp (igloo.py:209-210)
__device__ __forceinline__ void wva_pool_store(const WvaRows& r, wrsrc_t yp_w, int t0, int hw, int lane) {
    const int q0 = t0 / GNN_POOL + 3 * hw;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        float m0 = __uint_as_float(r.v[8 * p][0]), m1 = __uint_as_float(r.v[8 * p][1]);
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            m0 = max_nan(m0, __uint_as_float(r.v[8 * p + j][0]));
            m1 = max_nan(m1, __uint_as_float(r.v[8 * p + j][1]));
        }
        if (q0 + p < POOLED) __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(m0), __float_as_uint(m1)}, yp_w, (uint32_t)lane * 8u, (q0 + p) * (C * 4), 0);
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

template <bool PROF>
__global__ __launch_bounds__(512, 2) void fused_front_tc_kernel(Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEMT];
    unsigned char* bufX = smem;
    unsigned char* bufY = smem + BUF_BYTES;
    auto prow2 = [&](int parity) { return reinterpret_cast<uint16_t*>(smem + PROW_OFF + parity * PROW_BYTES); };
    float* bias_s = reinterpret_cast<float*>(smem + BIAS_OFF);
    int* s_last = reinterpret_cast<int*>(smem + LAST_OFF);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool helper = wave >= 4;
    const int hw = wave & 3;
    const int ht = tid & 255;
    const int64_t wi = blockIdx.x / a.split;
    const int part = blockIdx.x % a.split;
    const uint8_t* bases = a.bases + wi * W;
    float* mp_w[2] = {a.mp + (wi * 2 + 0) * NPAIR, a.mp + (wi * 2 + 1) * NPAIR};
    const int woff = hw * WNBLK_B;

    for (int i = tid; i < CARRY * ROW_U4; i += 512) {            // carry rows of the first step = the causal zero padding
        reinterpret_cast<uint4*>(bufX)[i] = make_uint4(0, 0, 0, 0);
        reinterpret_cast<uint4*>(bufY)[i] = make_uint4(0, 0, 0, 0);
    }
    if (tid >= 256) bias_s[tid - 256] = a.conv_b[(tid - 256) >> 7][tid & 127];
    if (tid == 0) *s_last = -1;
    __syncthreads();
    if (a.yp_c) {
        int last = -1;
        for (int i = tid * 12; i < tid * 12 + 12 && i < W; ++i)
            if (base_code_f(bases[i]) >= 0) last = i;
        if (last >= 0) atomicMax(s_last, last);
    }
    __syncthreads();
    const int nsteps = a.yp_c ? max(1, min(STEPST, (*s_last + 1 + 15 + FTT - 1) / FTT)) : STEPST;
    const int per = (nsteps + a.split - 1) / a.split;
    const int s_lo = min(part * per, nsteps), s_hi = min(s_lo + per, nsteps);
    const int s_begin = s_hi > s_lo ? (s_lo > 0 ? s_lo - 1 : 0) : s_hi;       // one warm-up step per run but the first (gnn_fused_x3.hip)
    if (tid < PROW_N) {
#pragma unroll
        for (int s01 = 0; s01 < 2; ++s01) {
            uint32_t lo, hi;
            const int t = (s_begin + s01) * FTT - CARRY + tid;
            prow_fetch(bases, t, lo, hi);
            prow2((s_begin + s01) & 1)[tid] = prow_make(lo, hi, t);
        }
    }
    __syncthreads();
    unsigned jitter_state = 0x9E3779B9u * (unsigned)(wave + 1) + (unsigned)blockIdx.x * 7919u;     // TC_JITTER builds only
    (void)jitter_state;
    unsigned long long cyc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tick_ = 0;
    // conv1 gather: helper thread ht owns the 16 channels 16 (ht & 7) .. of rows (ht >> 3) + 32 k, k = 0, 1, 2
    const int gpq = ht & 7, grow0 = ht >> 3;

    if (!helper) {
        __builtin_amdgcn_s_setprio(2);
        // The wave's n-block (woff) is folded into the resources' base addresses, so the byte offset of every weight request is a
        // compile-time constant: an `s_mov literal` the compiler rematerialises where it needs it.  As `woff + constant` (an s_add of
        // a register) the 64 offsets of a conv loop were kept in SGPRs across the whole step loop, and when round 6 added the table's
        // resource the overflow came back as v_readlane + wait states INSIDE the conv loops (18 per loop).
        const wrsrc_t cw[2] = {make_wrsrc(a.tcw[0] + woff, 64 * WUNIT_B - woff), make_wrsrc(a.tcw[1] + woff, 64 * WUNIT_B - woff)};
        const wrsrc_t vw = make_wrsrc(a.wv_w[1] + woff, 8 * WUNIT_B - woff);            // head A's w_v is inside the table
        const wrsrc_t tblr = make_wrsrc(a.wva_tbl, (int)(WvaTable::ROWS * WvaTable::ROW_BYTES));
        // one resource for the window's pooled rows of both heads (head h at byte h * POOLED * C * 4): four SGPRs less than one per head -
        // the matrix role lives at the edge of the SGPR file, and spilled SGPRs come back as v_readlane + wait states inside the conv loops
        const wrsrc_t yp_w = make_wrsrc(reinterpret_cast<const unsigned char*>(a.yp + wi * 2 * (size_t)POOLED * C), 2 * POOLED * C * 4);
        WU ring[RINGT];
        prime_tc(ring, cw[0], 0, lane);
        // head A's table rows of step s+1 are requested behind B0 of step s and pooled at the end of that interval: their round trip
        // (~4 k cycles with 384 missing lines per CU in flight) hides behind the w_v B tile.  The row indices are made behind B1 from
        // the next step's pair rows (wva_step_index): no memory round trip in front of the requests
        {                                                                        // the first step's rows: nothing to hide their round trip behind
            WvaRows w0;
            wva_issue(w0, tblr, wva_step_index(prow2(s_begin & 1), bases, s_begin * FTT, hw, lane), lane);
            if (s_begin >= s_lo && s_begin < s_hi) wva_pool_store(w0, yp_w, s_begin * FTT, hw, lane);      // not a warm-up step
        }
        uint32_t wva_next = 0;                                                   // row indices of step s+1: made behind B1 of step s
        __syncthreads();                                                         // x1 of the first step is in bufX
        if constexpr (PROF) tick_ = __builtin_readcyclecounter();
#pragma unroll 1
        for (int step = s_begin; step < s_hi; ++step) {
            const int t0 = step * FTT;
            const bool store = step >= s_lo;
            f32x16 acc[NXI];
#pragma unroll
            for (int xi = 0; xi < NXI; ++xi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[xi][r] = 0.f;
            GNN_TICK(7)
            conv_tc(smem, cw[0], 0, ring, acc, lane, jitter_state);                         // b_0 .. b_7, conv2
            GNN_TICK(0)
            epilogue_f32(bufY, acc, a.inv_s[0], bias_s, hw, lane);
            GNN_TICK(1)
            TC_BARRIER_W();                                                      // ---- B1: x2 is in bufY
            GNN_TICK(2)
            // The interval in which the matrix waves used to compute head A's y @ w_v: conv3's first weights, chunk 1 of V3 (the
            // helpers make chunk 0 meanwhile) and the row indices of the table rows that are requested behind B0
            prime_tc(ring, cw[1], 0, lane);
            {
                const HLane h2m = hlane(smem, BUF_BYTES, hw, lane);
                Raw16 rm;
                load_rows(rm, h2m, 1);
                transform_store(rm, h2m, 1);
            }
            wva_next = wva_step_index(prow2((step + 1) & 1), bases, t0 + FTT, hw, lane);      // the next step's row indices (its pair rows are in LDS since B0 of the last step)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // chunk 1's fragments have landed before b'_0 releases the readers
            GNN_TICK(3)
#pragma unroll
            for (int xi = 0; xi < NXI; ++xi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[xi][r] = 0.f;
            conv_tc(smem, cw[1], 0, ring, acc, lane, jitter_state);                         // b'_0 .. b'_7, conv3
            GNN_TICK(4)
            prime_wv(ring, vw, lane);
            epilogue_x3(bufY, acc, a.inv_s[1], bias_s + C, hw, lane);
            GNN_TICK(5)
            TC_BARRIER_W();                                                      // ---- B0: x3 is in bufY
            GNN_TICK(6)
            {
                // head A's 24 table rows per wave of the NEXT step, all requested here (unconditionally: behind a branch the register
                // allocator spills 8 of the 24 rows and waits for each; the last step of a run reads rows nobody stores): 384 missing
                // lines per CU are more than the vector L1 keeps in flight, so whoever requests memory behind them waits at issue -
                // the w_v B tile has its weights already, and the rows are pooled at the end of the interval
                WvaRows wr;
                wva_issue(wr, tblr, wva_next, lane);
                f32x16 ac[NMB];
#pragma unroll
                for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ac[mb][r] = 0.f;
                wv_tile(smem, BUF_BYTES + CARRY * ROWX, ring, ac, lane);
                prime_tc(ring, cw[0], 0, lane);
                if (store) wv_pool_store(ac, yp_w, POOLED * C * 4, t0, hw, lane);
                {                                                                // V2 chunk 1 of the next step (x1(s+1) is in bufX since B0)
                    const HLane h1m = hlane(smem, 0, hw, lane);
                    Raw16 rm;
                    load_rows(rm, h1m, 1);
                    transform_store(rm, h1m, 1);
                }
                if (step + 1 >= s_lo && step + 1 < s_hi) wva_pool_store(wr, yp_w, t0 + FTT, hw, lane);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // chunk 1's fragments have landed before b_0 releases the readers
            }
        }
    } else {
        {
            GRow g;
#pragma unroll 1
            for (int k = 0; k < 3; ++k) {
                grow_issue(g, prow2(s_begin & 1), a.conv1_k, grow0 + 32 * k, gpq);
                grow_finish(g, bufX, grow0 + 32 * k, gpq);
            }
        }
        // the helpers are this kernel's critical path (the matrix waves wait for their chunks): they outrank the matrix waves, whose
        // one MFMA per 32 cycles needs few issue slots (43.5 k vs 46.1 k cycles per step, profiles/r04/tc_ab_prio_slp.txt)
        __builtin_amdgcn_s_setprio(3);
        uint32_t nlo = 0, nhi = 0;                       // bytes of this thread's pair row of the step AFTER next
        if (ht < PROW_N) prow_fetch(bases, (s_begin + 2) * FTT - CARRY + ht, nlo, nhi);
        const HLane h1 = hlane(smem, 0, hw, lane), h2 = hlane(smem, BUF_BYTES, hw, lane);
        const int cr = ht / ROW_U4, cc = ht - cr * ROW_U4;   // carry rows: 5 rows x 33 chunks of 16 B
        __syncthreads();
        if constexpr (PROF) tick_ = __builtin_readcyclecounter();
        Raw16 ra, rb;
        PairPass p0, p1;
        p0.u = p1.u = 0;
        // V2 chunks 0 and 1 of the first step
        load_rows(ra, h1, 0);
        load_rows(rb, h1, 1);
        transform_store(ra, h1, 0);
        load_rows(ra, h1, 2);
        transform_store(rb, h1, 1);
#pragma unroll 1
        for (int step = s_begin; step < s_hi; ++step) {
            const int t0 = step * FTT;
            const uint16_t* prow = prow2((step + 1) & 1);
            const bool ha = step >= s_lo;                                        // false in the warm-up step of a time-split run
            GNN_TICK(10)
            // Only the chunk transforms run beside the conv loops (the matrix waves wait for every chunk).  The pair products sit where
            // the matrix waves need nothing from the helpers - head A's (x1 in bufX until the gather behind b'_0) beside units 6, 7 and the
            // conv2 epilogue, head B's (x3 in bufY from B0 to the next conv2 epilogue) beside w_v B and, its last pass, beside unit 6 of
            // the next step - 3 passes of 64 entries per head through two register sets (three sets spill).  A warm-up step (time
            // split) stores nothing: ha / hb.
            const PairJob jb = {bufY, a.weff[1], a.pos_sorted[1], mp_w[1], t0, ha ? a.bucket_ptr[1][step] : 0, ha ? a.bucket_ptr[1][step + 1] : 0};
            const bool hb = step - 1 >= s_lo;                                    // head B of the previous step belongs to this run
            const int sb = max(step - 1, 0);
            const PairJob jbp = {bufY, a.weff[1], a.pos_sorted[1], mp_w[1], t0 - FTT, hb ? a.bucket_ptr[1][sb] : 0, hb ? a.bucket_ptr[1][sb + 1] : 0};
            const PairJob ja = {bufX, a.weff[0], a.pos_sorted[0], mp_w[0], t0, ha ? a.bucket_ptr[0][step] : 0, ha ? a.bucket_ptr[0][step + 1] : 0};
            // ---- conv2 phase: chunks 2 .. 7 (ra holds the rows of chunk 2; p0 / p1 the B passes 0 / 1, requested behind B0).  Only
            // the chunk transforms run beside the conv loop; the pair products sit in the intervals in which the matrix waves finish
            // the loop and run their epilogue without needing anything from the helpers.
            TC_HPRIO_HIGH();
            HBAR_W(8, 9);                                                        // b_0
            load_rows(rb, h1, 3);
            transform_store(ra, h1, 2);
            HBAR_W(8, 9);                                                        // b_1
            load_rows(ra, h1, 4);
            transform_store(rb, h1, 0);
            HBAR_W(8, 9);                                                        // b_2
            load_rows(rb, h1, 5);
            transform_store(ra, h1, 1);
            HBAR_W(8, 9);                                                        // b_3
            load_rows(ra, h1, 6);
            transform_store(rb, h1, 2);
            HBAR_W(8, 9);                                                        // b_4
            load_rows(rb, h1, 7);
            transform_store(ra, h1, 0);
            HBAR_W(8, 9);                                                        // b_5
            transform_store(rb, h1, 1);
            uint4 carry = make_uint4(0, 0, 0, 0);
            // head A's pair products of this step (x1 in bufX until the gather behind b'_0) beside units 6, 7 and the conv2 epilogue:
            // two register sets; the third pass is requested when the first is done and used last, where the helpers would wait for
            // the matrix waves anyway
            // The conv1 gather of the next step (3 rows per lane): rounds 0 and 1 are computed here and beside w_v A, where the helpers have
            // slack, and wait in 16 registers each for b'_0, behind which bufX may be written; round 2 runs beside conv3.
            GRow ga;
            GOut o0, o1;
            pass_issue(p1, ja, 0, hw, lane);
            grow_issue(ga, prow, a.conv1_k, grow0, gpq);
            HBAR_W(8, 9);                                                        // b_6
            pass_compute(p0, jbp, 2, hw, lane);                                  // head B's last pass of step s-1 (requested behind B0(s-1)):
            pass_rest(p0, jbp, 3, hw, lane);                                     // bufY holds x3(s-1) until the conv2 epilogue behind b_7
            pass_issue(p0, ja, 1, hw, lane);
            HBAR(8, 9);                                                          // b_7
            pass_compute<PairComputeF32>(p1, ja, 0, hw, lane);
            pass_issue(p1, ja, 2, hw, lane);
            pass_compute<PairComputeF32>(p0, ja, 1, hw, lane);
            if (ht < CARRY * ROW_U4) carry = *reinterpret_cast<const uint4*>(bufX + (FTT + cr) * ROWX + cc * 16);
            grow_compute(o0, ga);
            grow_issue(ga, prow, a.conv1_k, grow0 + 32, gpq);
            HBAR(8, 10);                                                         // ---- B1: x2 is in bufY
            TC_HPRIO_LOW();
            // head A's last pass, V3 chunk 0 (the matrix waves make chunk 1 while their table rows travel) and gather round 1
            load_rows(ra, h2, 0);
            pass_compute<PairComputeF32>(p1, ja, 2, hw, lane);
            pass_rest<PairComputeF32>(p1, ja, 3, hw, lane);
            transform_store(ra, h2, 0);
            load_rows(ra, h2, 2);
            grow_compute(o1, ga);
            GNN_TICK(12)
            // ---- conv3 phase: chunks 2 .. 7; the conv1 gather of the next step (3 rows per lane, table loads two intervals ahead of
            // their use) in its second half, carry rows, pair rows
            {
                TC_HPRIO_HIGH();
                HBAR_W(13, 14);                                                  // b'_0: nobody reads bufX any more
                if (ht < CARRY * ROW_U4) *reinterpret_cast<uint4*>(bufX + cr * ROWX + cc * 16) = carry;
                grow_store(o0, bufX, grow0, gpq);
                grow_store(o1, bufX, grow0 + 32, gpq);
                load_rows(rb, h2, 3);
                transform_store(ra, h2, 2);
                HBAR_W(13, 14);                                                  // b'_1
                load_rows(ra, h2, 4);
                transform_store(rb, h2, 0);
                HBAR_W(13, 14);                                                  // b'_2
                load_rows(rb, h2, 5);
                transform_store(ra, h2, 1);
                HBAR_W(13, 14);                                                  // b'_3
                load_rows(ra, h2, 6);
                transform_store(rb, h2, 2);
                HBAR_W(13, 14);                                                  // b'_4
                load_rows(rb, h2, 7);
                transform_store(ra, h2, 0);
                grow_issue(ga, prow, a.conv1_k, grow0 + 64, gpq);
                HBAR_W(13, 14);                                                  // b'_5
                transform_store(rb, h2, 1);
                HBAR_W(13, 14);                                                  // b'_6: V3 is complete
                grow_finish(ga, bufX, grow0 + 64, gpq);
                // x2 carry rows: nobody reads rows 0..4 of bufY any more, the conv3 epilogue (behind b'_7) overwrites rows 96..100
                if (ht < CARRY * ROW_U4) {
                    const uint4 c2 = *reinterpret_cast<const uint4*>(bufY + (FTT + cr) * ROWX + cc * 16);
                    *reinterpret_cast<uint4*>(bufY + cr * ROWX + cc * 16) = c2;
                }
                HBAR_W(13, 14);                                                  // b'_7
            }
            if (ht < PROW_N) {                                                   // pair rows of step s+2 (parity buffer of step s: read last before b'_7)
                const int t = t0 + 2 * FTT - CARRY + ht;
                prow2(step & 1)[ht] = prow_make(nlo, nhi, t);
                prow_fetch(bases, t + FTT, nlo, nhi);
            }
            // head B's pair products of this step (x3 in bufY from B0 to the next conv2 epilogue): the first two passes requested
            // beside the conv3 epilogue; behind B0, beside the matrix waves' w_v B: pass 0, the third pass's request, pass 1, V2 chunk
            // 0 of the next step (chunk 1: the matrix waves, behind their w_v B tile), pass 2
            pass_issue(p0, jb, 0, hw, lane);
            pass_issue(p1, jb, 1, hw, lane);
            GNN_TICK(13)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            HBAR(13, 11);                                                        // ---- B0: x3 is in bufY, x1(s+1) in bufX
            TC_HPRIO_LOW();
            load_rows(ra, h1, 0);
            pass_compute(p0, jb, 0, hw, lane);
            pass_issue(p0, jb, 2, hw, lane);
            pass_compute(p1, jb, 1, hw, lane);
            transform_store(ra, h1, 0);
            load_rows(ra, h1, 2);
            GNN_TICK(15)
        }
        if (s_hi > s_lo) {                                  // head B's last pass of this run's last step
            const PairJob jl = {bufY, a.weff[1], a.pos_sorted[1], mp_w[1], (s_hi - 1) * FTT, a.bucket_ptr[1][s_hi - 1], a.bucket_ptr[1][s_hi]};
            pass_compute(p0, jl, 2, hw, lane);
            pass_rest(p0, jl, 3, hw, lane);
        }
    }
    if (nsteps < STEPST && part == a.split - 1) {   // the all-N tail: copy instead of compute
        const int q0 = nsteps * (FTT / GNN_POOL);
        const int nrow4 = (POOLED - q0) * (C / 4);
        for (int i = tid; i < 2 * nrow4; i += 512) {
            const int h = i >= nrow4, j = i - h * nrow4;
            const size_t off = (size_t)h * POOLED * C + (size_t)q0 * C + (size_t)j * 4;
            *reinterpret_cast<float4*>(a.yp + wi * 2 * (size_t)POOLED * C + off) = *reinterpret_cast<const float4*>(a.yp_c + off);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
            for (int e = a.bucket_ptr[h][nsteps] + tid; e < NPAIR; e += 512) mp_w[h][e] = a.mp_c[h * NPAIR + e];
    }
    if constexpr (PROF) {
        if (tid == 0)
            for (int i = 0; i < 8; ++i) atomicAdd(a.cycles + i, cyc[i]);
        if (tid == 256)
            for (int i = 8; i < 16; ++i) atomicAdd(a.cycles + i, cyc[i]);
    }
}

// Builds WvaTable: one workgroup = 128 channels x WVA_BUILD_ROWS consecutive table rows.  x1 of a row exactly as grow_compute makes it
// (the same three pair-table rows in the same order, LeakyReLU as max(v, 0.1 v)), then y[c] = sum_k x1[k] w_v[k][c] accumulated in f64
// and rounded to f32 once.
constexpr int WVA_BUILD_ROWS = 64;
__global__ __launch_bounds__(128) void wva_table_kernel(const float* __restrict__ pairs6, const float* __restrict__ w_v, float* __restrict__ tbl) {
    __shared__ float xs[C];
    const int c = threadIdx.x;
    float wcol[C];
#pragma unroll
    for (int k = 0; k < C; ++k) wcol[k] = w_v[k * C + c];
    const int perm = ((c >> 2) & 3) * 32 + (c >> 4) * 4 + (c & 3);          // channel 16 pq + 4 i + e of a pair-table row (pack_fused_c6_weights)
    const uint32_t r_end = min((uint32_t)(blockIdx.x + 1) * WVA_BUILD_ROWS, WvaTable::ROWS);
    for (uint32_t row = blockIdx.x * WVA_BUILD_ROWS; row < r_end; ++row) {
        // the row's nine base slots (positions t-5 .. t+3): 0..3 = ACGT, 4 = any other byte, -1 = before the window start
        int dig[9];
        if (row < WvaTable::D4_ROWS) {
#pragma unroll
            for (int i = 0; i < 9; ++i) dig[i] = (int)((row >> (2 * (8 - i))) & 3u);
        } else {
            int nabs = 0;
            uint32_t idx = row - WvaTable::D5_OFF;
            if (row < WvaTable::D5_OFF) {
                int t = 4;
                while (row < WvaTable::s5_off(t)) --t;
                nabs = 5 - t;
                idx = row - WvaTable::s5_off(t);
            }
#pragma unroll
            for (int i = 8; i >= 0; --i) {
                dig[i] = i >= nabs ? (int)(idx % 5u) : -1;
                if (i >= nabs) idx /= 5u;
            }
        }
        int tk[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int d0 = dig[i], d1 = dig[i + 1], d2 = dig[i + 2], d3 = dig[i + 3];
            tk[i] = d0 < 0 ? -1 : ((d0 | d1 | d2 | d3) & 4) ? 0 : 1 + d0 * 64 + d1 * 16 + d2 * 4 + d3;
        }
        float v[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) v[j] = pairs6[((size_t)j * PAIR_ROWS + pair_row(tk[2 * j], tk[2 * j + 1])) * C + perm];
        const float t = v[0] + v[1] + v[2];
        xs[c] = vmax_raw(t, t * LRELU);
        __syncthreads();
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < C; ++k) acc = __builtin_fma((double)xs[k], (double)wcol[k], acc);
        tbl[(size_t)row * C + c] = (float)acc;
        __syncthreads();
    }
}

static void fill_args(const gnn_ctx* ctx, Args& a, const uint8_t* bases) {
    const DeviceWeights& d = ctx->w;
    a.bases = bases;
    a.conv1_k = d.conv1_pairs6;
    for (int i = 0; i < 2; ++i) {
        a.tcw[i] = reinterpret_cast<const unsigned char*>(d.tc_frag[i]);
        a.inv_s[i] = d.tc_inv_s[i];
        a.conv_b[i] = d.conv_b[i];
        a.wv_w[i] = reinterpret_cast<const unsigned char*>(d.wv_frag_h[i]);
        a.weff[i] = d.weff6[i];
        a.pos_sorted[i] = d.pos_sorted[i];
        a.bucket_ptr[i] = d.bucket_ptr96[i];
    }
    a.wva_tbl = reinterpret_cast<const unsigned char*>(d.tc_wva_tbl);
    a.cycles = nullptr;
    a.split = 1;
}

static void launch(const Args& a, bool prof, unsigned nwin, hipStream_t stream) {
    const unsigned n = nwin * (unsigned)a.split;
    if (prof) hipLaunchKernelGGL((fused_front_tc_kernel<true>), dim3(n), dim3(512), 0, stream, a);
    else hipLaunchKernelGGL((fused_front_tc_kernel<false>), dim3(n), dim3(512), 0, stream, a);
}

// f32 -> f16 bits (round to nearest even) on the host, via the compiler's _Float16
static uint16_t f16_bits_of(double v) {
    const _Float16 h = (_Float16)v;
    uint16_t b;
    std::memcpy(&b, &h, 2);
    return b;
}
static double f16_value_of(uint16_t b) {
    _Float16 h;
    std::memcpy(&h, &b, 2);
    return (double)h;
}

}  // namespace tc

// Transformed conv weights U_xi = s * sum_k G[xi][k] w[k] in f64 (G of F(3,6), oracle/toomcook.py), split into f16 hi | lo limbs in
// MFMA fragment order [k16 unit][xi][n-block][hi | lo][lane 64][8]; s = the power of two that puts max |U| into [512, 1024) (the
// low limbs leave the f16 subnormal range; 1 / s goes into the inverse transform).  And the IGLOO entry ranges per 96-row step.
int pack_fused_tc_weights(gnn_ctx* ctx, const gnn_weights* w) {
    using namespace tc;
    DeviceWeights& d = ctx->w;
    static const double G[NXI][KS] = {{-1, 0, 0, 0, 0, 0},
                                      {-2. / 9, -2. / 9, -2. / 9, -2. / 9, -2. / 9, -2. / 9},
                                      {-2. / 9, 2. / 9, -2. / 9, 2. / 9, -2. / 9, 2. / 9},
                                      {1. / 90, 1. / 45, 2. / 45, 4. / 45, 8. / 45, 16. / 45},
                                      {1. / 90, -1. / 45, 2. / 45, -4. / 45, 8. / 45, -16. / 45},
                                      {32. / 45, 16. / 45, 8. / 45, 4. / 45, 2. / 45, 1. / 45},
                                      {32. / 45, -16. / 45, 8. / 45, -4. / 45, 2. / 45, -1. / 45},
                                      {0, 0, 0, 0, 0, 1}};
    const float* ck[2] = {w->conv2_kernel, w->conv3_kernel};
    for (int cv = 0; cv < 2; ++cv) {
        std::vector<double> U((size_t)NXI * C * C);
        double amax = 0.0;
        for (int xi = 0; xi < NXI; ++xi)
            for (int c = 0; c < C; ++c)
                for (int n = 0; n < C; ++n) {
                    double s = 0.0;
                    for (int k = 0; k < KS; ++k) s += G[xi][k] * (double)ck[cv][((size_t)k * C + c) * C + n];
                    U[((size_t)xi * C + c) * C + n] = s;
                    amax = std::max(amax, std::fabs(s));
                }
        const double scale = amax > 0 ? std::exp2(std::floor(std::log2(1024.0 / amax))) : 1.0;
        std::vector<uint16_t> frag((size_t)8 * NXI * 4 * 2 * 64 * 8);
        for (int u = 0; u < 8; ++u)
            for (int xi = 0; xi < NXI; ++xi)
                for (int nb = 0; nb < 4; ++nb)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 8; ++e) {
                            const int k = u * 16 + (l >> 5) * 8 + e, n = nb * 32 + (l & 31);
                            const double v = U[((size_t)xi * C + k) * C + n] * scale;
                            const uint16_t hi = f16_bits_of(v), lo = f16_bits_of(v - f16_value_of(hi));
                            const size_t base = ((((size_t)u * NXI + xi) * 4 + nb) * 2) * 64 * 8;
                            frag[base + (size_t)l * 8 + e] = hi;
                            frag[base + 64 * 8 + (size_t)l * 8 + e] = lo;
                        }
        // GNN_TC_WLO_MASK=<hex> (energy probe, round 5; wrong results): mantissa bits of the weights' low limbs masked off
        static const bool wlo_probe = debug_switch("GNN_TC_WLO_MASK");
        if (wlo_probe) {
            const uint16_t mask = (uint16_t)std::strtoul(std::getenv("GNN_TC_WLO_MASK"), nullptr, 16);
            for (size_t b0 = 0; b0 < frag.size(); b0 += 2 * 64 * 8)
                for (size_t i = 0; i < 64 * 8; ++i) frag[b0 + 64 * 8 + i] &= mask;
        }
        void* p = nullptr;
        GNN_HIP(hipMalloc(&p, frag.size() * 2));
        ctx->owned.push_back(p);
        GNN_HIP(hipMemcpy(p, frag.data(), frag.size() * 2, hipMemcpyHostToDevice));
        d.tc_frag[cv] = static_cast<uint16_t*>(p);
        d.tc_inv_s[cv] = (float)(1.0 / scale);
    }
    const gnn_igloo_weights* ig[2] = {&w->igloo_a, &w->igloo_b};
    for (int h = 0; h < 2; ++h) {
        std::vector<int32_t> ptr(STEPST + 1, 0);
        for (int i = 0; i < NPAIR; ++i) ptr[ig[h]->patches[i] / FTT + 1] += 1;      // range-checked by gnn_load_weights before
        for (int s = 0; s < STEPST; ++s) ptr[s + 1] += ptr[s];
        void* p = nullptr;
        GNN_HIP(hipMalloc(&p, ptr.size() * 4));
        ctx->owned.push_back(p);
        GNN_HIP(hipMemcpy(p, ptr.data(), ptr.size() * 4, hipMemcpyHostToDevice));
        d.bucket_ptr96[h] = static_cast<int32_t*>(p);
    }
    // head A's y @ w_v table (WvaTable): 1.38 GB, built on the device from the pair tables and w_v A
    {
        void* p = nullptr;
        const size_t bytes = (size_t)WvaTable::ROWS * WvaTable::ROW_BYTES;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            set_error("hipMalloc of the " + std::to_string(bytes >> 20) + " MiB table of head A's y @ w_v rows failed: " + hipGetErrorString(e));
            return GNN_ERR_NOMEM;
        }
        ctx->owned.push_back(p);
        d.tc_wva_tbl = static_cast<float*>(p);
        hipLaunchKernelGGL(wva_table_kernel, dim3((WvaTable::ROWS + WVA_BUILD_ROWS - 1) / WVA_BUILD_ROWS), dim3(128), 0, ctx->stream, d.conv1_pairs6,
                           d.w_v[0], d.tc_wva_tbl);
        GNN_HIP(hipGetLastError());
    }
    // the all-N window's outputs, computed once by the kernel itself (padding skip)
    void* bn = nullptr;
    GNN_HIP(hipMalloc(&bn, W));
    ctx->owned.push_back(bn);
    GNN_HIP(hipMemsetAsync(bn, 'N', W, ctx->stream));
    void *yc = nullptr, *mc = nullptr;
    GNN_HIP(hipMalloc(&yc, (size_t)2 * POOLED * C * sizeof(float)));
    ctx->owned.push_back(yc);
    GNN_HIP(hipMalloc(&mc, (size_t)2 * NPAIR * sizeof(float)));
    ctx->owned.push_back(mc);
    Args a;
    fill_args(ctx, a, static_cast<const uint8_t*>(bn));
    a.mp = static_cast<float*>(mc);
    a.yp = static_cast<float*>(yc);
    a.yp_c = nullptr;
    a.mp_c = nullptr;
    launch(a, false, 1, ctx->stream);
    GNN_HIP(hipGetLastError());
    GNN_HIP(hipStreamSynchronize(ctx->stream));
    d.tc_yp_const = static_cast<float*>(yc);
    d.tc_mp_const = static_cast<float*>(mc);
    return GNN_OK;
}

int launch_front_tc(gnn_ctx* ctx, const uint8_t* bases, int64_t n) {
    using namespace tc;
    if (reinterpret_cast<uintptr_t>(bases) & 3u) {
        set_error("f16x3tc: the window buffer must be 4-byte aligned");
        return GNN_ERR_ARG;
    }
    Args a;
    fill_args(ctx, a, bases);
    a.mp = ctx->ws.mp;
    a.yp = ctx->ws.yp;
    a.yp_c = ctx->c6_pad_skip ? ctx->w.tc_yp_const : nullptr;
    a.mp_c = ctx->c6_pad_skip ? ctx->w.tc_mp_const : nullptr;
    a.cycles = ctx->phase_cycles;
    if (ctx->time_split && n > 0 && ctx->cu_count > 0) a.split = (int)std::max<int64_t>(1, std::min<int64_t>(4, ctx->cu_count / n));
    ctx->last_split = a.split;
    launch(a, ctx->phase_cycles != nullptr, (unsigned)n, ctx->stream);
    GNN_HIP(hipGetLastError());
    return GNN_OK;
}

}  // namespace gnn
