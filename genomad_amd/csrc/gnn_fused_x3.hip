// Fused front end, GNN_PREC_F16X3 (the default arithmetic) and GNN_PREC_BF16X3: every f32 operand of the four contractions
// (conv2, conv3, y @ w_v of both IGLOO heads) is split into two 16-bit limbs, hi = round16(x), lo = round16(x - hi), and
//
//   x * w  ~=  hi(x) * lo(w)  +  lo(x) * hi(w)  +  hi(x) * hi(w)          three v_mfma_f32_32x32x16_{f16,bf16} per k16, f32 accumulate
//
// With f16 limbs (11 + 11 significant bits) the class scores are f32-class: max |dscore| 1-2e-5 against the exact-f32 path
// over 1 048 576 windows (profiles/), which is why this is the arithmetic main() and bench.py default to; bf16 limbs
// (8 + 8 bits, 6e-5 ... 1e-4) keep the f32 range and are the fallback when an activation leaves the f16 range.
//
// Structure: the streaming structure of gnn_fused_c6.hip (one workgroup = one window, steps of 128 positions, both
// activation buffers resident in LDS with 5 carry rows, 4 matrix waves + 4 helper waves, 4 workgroup barriers per step),
// which replaces the round-1 kernel of gnn_fused.hip for these two modes:
//   * a GEMM tile is fully unrolled over its k16 units (static ring slots, static LDS offsets as DS immediates);
//   * weight fragments come through buffer loads (resource + 32-bit lane offset + SGPR unit offset) three units ahead, and
//     the stream runs THROUGH the tile boundaries: the last three units of a tile request the first three of the next, so no
//     tile starts behind an L2 round trip (the round-1 kernel started each of its 4 tiles per step cold);
//   * conv1 gather on lane pairs with the line-friendly pair tables, pair rows made one step ahead, IGLOO pair products 4 lanes
//     per entry with the weights of the next pass in flight (gnn_fused_helpers.h);
//   * padding skip: the steps that lie entirely in a window's all-N tail copy the rows an all-N window produces.
//
// LDS row (528 B = 33 x 16: an odd multiple of 16 B keeps the 16-lane groups of ds_read_b128 on distinct slots):
//   [0,256) 128 ch hi limbs | [256,512) 128 ch lo limbs | pad
// Weights: the fragment-order stream of pack_frags (gnn_fused.hip): [k16 unit][n-block 4][hi 1 KiB | lo 1 KiB].
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "gnn_fused_helpers.h"

namespace gnn {
namespace x3 {

constexpr int NMB = 4;                       // 32-row blocks per step
constexpr int FTX = 32 * NMB;                // rows per step
constexpr int STEPSX = (T + FTX - 1) / FTX;
constexpr int ROWX = 528, LOX = 256;
constexpr int BUFX_ROWS = CARRY + FTX;
constexpr int BUFX_BYTES = BUFX_ROWS * ROWX;
constexpr int PROW_OFF = 2 * BUFX_BYTES;
constexpr int PROW_N = FTX + 4;              // pair rows a step's conv1 gather reads
constexpr int PROW_BYTES = ((PROW_N * 2 + 15) / 16) * 16;
constexpr int BIAS_OFF = PROW_OFF + 2 * PROW_BYTES;   // two pair-row buffers (step parity), then conv2 | conv3 bias, 2 x 128 f32
constexpr int LAST_OFF = BIAS_OFF + 2 * C * 4;        // index of the window's last ACGT base
constexpr int SMEMX = LAST_OFF + 16;
constexpr int ROW_U4 = ROWX / 16;            // 33
constexpr int WNBLK_B = 2048;                // weight bytes per (k16 unit, n-block): hi fragment | lo fragment
constexpr int WUNIT_B = 4 * WNBLK_B;         // per k16 unit
constexpr int RINGW = 4;                     // weight ring slots: RINGW - 1 units are in flight ahead of the MFMAs
static_assert(FTX == FT && STEPSX == FSTEPS, "bucket_ptr of gnn_load_weights is cut for steps of FT rows");
static_assert(SMEMX <= 160 * 1024, "LDS budget");
static_assert(PROW_N <= 256 && CARRY * ROW_U4 <= 256, "one helper thread per pair row / carry chunk");

struct Args {
    const uint8_t* bases;
    const float* conv1_k;             // pair tables in the gather's lane order, bias folded into table 0 (DeviceWeights::conv1_pairs6)
    const unsigned char* conv_w[2];   // [k16 unit 48][nblk 4][hi | lo] x 1 KiB
    const float* conv_b[2];
    const unsigned char* wv_w[2];     // the same, 8 units
    const float* weff[2];             // folded IGLOO weights, PairW layout (DeviceWeights::weff6)
    const int32_t* pos_sorted[2];
    const int32_t* bucket_ptr[2];     // (STEPSX + 1,) entry ranges per step
    float* mp;
    float* yp;
    const float* yp_c;                // outputs of an all-N window (padding skip), nullptr = compute everything
    const float* mp_c;
    unsigned long long* cycles;       // PROF builds: 16 phase counters, matrix wave 0 -> 0..7, helper wave 4 -> 8..15
    int split;                        // workgroups per window (time split for launches smaller than the chip, see the kernel)
};

struct WU {
    uint4 h, l;        // hi / lo fragment of one k16 unit
};
struct XU {
    uint4 h[NMB], l[NMB];
};

template <bool F16>
__device__ __forceinline__ f32x16 mma16(uint4 a, uint4 b, f32x16 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ void load_wu(WU& w, wrsrc_t r, uint32_t l16, int soff) {
    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, l16, soff, 0);
    const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(r, l16 + 1024, soff, 0);
    w.h = make_uint4(a[0], a[1], a[2], a[3]);
    w.l = make_uint4(b[0], b[1], b[2], b[3]);
}
// xh = lane base of the tile's first row (row l & 31, + 16 B for lanes 32-63); OFF = tap * ROWX + (k16 inside the tap) * 32:
// every offset is an immediate of the DS instruction
template <int OFF>
__device__ __forceinline__ void load_xu(XU& f, const unsigned char* __restrict__ xh) {
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
        f.h[mb] = *reinterpret_cast<const uint4*>(xh + OFF + mb * 32 * ROWX);
        f.l[mb] = *reinterpret_cast<const uint4*>(xh + OFF + LOX + mb * 32 * ROWX);
    }
}

// One k16 unit = one scheduling region: its 12 MFMAs with the
// activation fragments of the next unit (8 LDS reads) and the weight fragments of the unit RINGW - 1 ahead (2 L2 loads)
// issued between them.
template <bool SWAP, bool F16, bool LX, int OFFN>
__device__ __forceinline__ void unit(const WU& wcur, WU& wload, const XU& xcur, XU& xload, const unsigned char* __restrict__ xh,
                                     wrsrc_t wr, int wnext, uint32_t l16, f32x16 (&acc)[NMB]) {
#ifndef GNN_ABL_NOX
    if constexpr (LX) load_xu<OFFN>(xload, xh);
#endif
#ifndef GNN_ABL_NOW
    load_wu(wload, wr, l16, wnext);
#endif
#if defined(GNN_ABL_CONSTX) || defined(GNN_ABL_CONSTW)
    // measurement only (wrong results): every load is issued and lands in its registers, but the MFMAs read the SAME operand
    // registers in every unit - separates what moving the operands costs from what feeding the matrix pipe changing data costs
    {
        const uint32_t* px = reinterpret_cast<const uint32_t*>(&xcur);
        const uint32_t* pw = reinterpret_cast<const uint32_t*>(&wcur);
#pragma unroll
        for (int i = 0; i < (int)(sizeof(XU) / 4); ++i) asm volatile("" ::"v"(px[i]));
#pragma unroll
        for (int i = 0; i < (int)(sizeof(WU) / 4); ++i) asm volatile("" ::"v"(pw[i]));
    }
#endif
#ifdef GNN_ABL_CONSTX
    const uint4 kx = make_uint4(0x3C003C00u + l16, 0x3C003800u, 0x38003C00u, 0x3C003C00u);
#define GNN_XH(mb) make_uint4(kx.x, kx.y + (mb), kx.z, kx.w)
#define GNN_XL(mb) make_uint4(kx.x, kx.y, kx.z + (mb), kx.w)
#else
#define GNN_XH(mb) xcur.h[mb]
#define GNN_XL(mb) xcur.l[mb]
#endif
#ifdef GNN_ABL_CONSTW
    const uint4 kw = make_uint4(0x38003C00u + l16, 0x3C003C00u, 0x3C003800u, 0x38003800u);
#define GNN_WH kw
#define GNN_WL kw
#else
#define GNN_WH wcur.h
#define GNN_WL wcur.l
#endif
#ifndef GNN_X3_TERM_MAJOR
    // every MFMA shares one operand register with its predecessor (w_l x_h0, w_h x_h0, w_h x_l0 | w_h x_l1, w_h x_h1, w_l x_h1 |
    // ...): the matrix pipe draws a little less when one operand does not change - 27.80 vs 28.01 ms per 4096 windows against
    // the term-major order (all w_l x_h, then all w_h x_l, then all w_h x_h; GNN_X3_TERM_MAJOR), the loop cycles are the same
    // (three dependent MFMAs per accumulator in a row do not stall): profiles/history/r03_x3_operand_chain_ab.txt
#define GNN_MM(W_, X_, mb) acc[mb] = SWAP ? mma16<F16>(W_, X_, acc[mb]) : mma16<F16>(X_, W_, acc[mb])
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
        if (mb % 2 == 0) {
            GNN_MM(GNN_WL, GNN_XH(mb), mb);
            GNN_MM(GNN_WH, GNN_XH(mb), mb);
            GNN_MM(GNN_WH, GNN_XL(mb), mb);
        } else {
            GNN_MM(GNN_WH, GNN_XL(mb), mb);
            GNN_MM(GNN_WH, GNN_XH(mb), mb);
            GNN_MM(GNN_WL, GNN_XH(mb), mb);
        }
    }
#undef GNN_MM
#else
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) acc[mb] = SWAP ? mma16<F16>(GNN_WL, GNN_XH(mb), acc[mb]) : mma16<F16>(GNN_XH(mb), GNN_WL, acc[mb]);
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) acc[mb] = SWAP ? mma16<F16>(GNN_WH, GNN_XL(mb), acc[mb]) : mma16<F16>(GNN_XL(mb), GNN_WH, acc[mb]);
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) acc[mb] = SWAP ? mma16<F16>(GNN_WH, GNN_XH(mb), acc[mb]) : mma16<F16>(GNN_XH(mb), GNN_WH, acc[mb]);
#endif
#undef GNN_XH
#undef GNN_XL
#undef GNN_WH
#undef GNN_WL
#pragma unroll
    for (int i = 0; i < 3 * NMB; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                           // 1 MFMA
        if (LX && i < 2 * NMB) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // 1 DS read
        if (i == 1 || i == 5) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // 1 VMEM read
    }
    GNN_REGION_END();
}

// FTX rows x 32 columns, K = NTAPS * 128, as NTAPS * 8 fully unrolled k16 units.  SWAP: D = W^T X^T for the convs (a lane
// ends up with 16 channels of one row), D = X W for y @ w_v (a lane ends up with 16 rows of one channel: the max-pool is
// register local).  The last RINGW - 1 units request the first units of the NEXT tile (wr_next).
template <bool SWAP, bool F16, int NTAPS, int NTAPS_NEXT>
__device__ __forceinline__ void gemm_tile(const unsigned char* __restrict__ smem, int xoff, wrsrc_t wr, wrsrc_t wr_next, int woff,
                                          WU (&ring)[RINGW], f32x16 (&acc)[NMB], int lane) {
    constexpr int NK = NTAPS * 8;
    static_assert(NK % RINGW == 0 && NTAPS_NEXT * 8 >= RINGW - 1, "the next tile's first units land in the slots it expects them in");
    // the lane's row offset is made opaque to the compiler (one base register per tile, every other offset an immediate)
    uint32_t rowoff = (uint32_t)xoff + (uint32_t)(lane & 31) * ROWX + (uint32_t)(lane >> 5) * 16u;
    asm volatile("" : "+v"(rowoff));
    const unsigned char* xh = smem + rowoff;
    const uint32_t l16 = (uint32_t)lane * 16u;
    XU xa, xb;
    load_xu<0>(xa, xh);
    GNN_REGION_END();
    static_for(std::make_integer_sequence<int, NK>{}, [&](auto kc) {
        constexpr int k = decltype(kc)::value, kn = k + 1;
        constexpr int OFFN = (kn / 8) * ROWX + (kn % 8) * 32;
        constexpr bool LX = kn < NK, NEXT = k + RINGW - 1 >= NK;       // NEXT: this unit's request belongs to the next tile
        constexpr int kw = NEXT ? k + RINGW - 1 - NK : k + RINGW - 1;
        if constexpr (k % 2 == 0)
            unit<SWAP, F16, LX, OFFN>(ring[k % RINGW], ring[(k + RINGW - 1) % RINGW], xa, xb, xh, NEXT ? wr_next : wr, woff + kw * WUNIT_B, l16, acc);
        else
            unit<SWAP, F16, LX, OFFN>(ring[k % RINGW], ring[(k + RINGW - 1) % RINGW], xb, xa, xh, NEXT ? wr_next : wr, woff + kw * WUNIT_B, l16, acc);
    });
}

__device__ __forceinline__ void prefetch_w(WU (&ring)[RINGW], wrsrc_t wr, int woff, int lane) {
    const uint32_t l16 = (uint32_t)lane * 16u;
#pragma unroll
    for (int u = 0; u < RINGW - 1; ++u) load_wu(ring[u], wr, l16, woff + u * WUNIT_B);
    asm volatile("" ::: "memory");
}

// two values -> packed 16-bit hi / lo words: hi = round16(x) (RNE), lo = round16(x - hi).  The conv epilogues are the matrix
// waves' only VALU-bound phase (12 % of a step), so the f16 form is spelled out: v_cvt_pk_f16_f32 for hi, x - hi straight from
// the packed f16 halves with v_fma_mix_f32 (sub_f16_lo / _hi, gnn_fused_common.h: hi * -1.0 + x, exact, the same value as
// converting back and subtracting), one more
// v_cvt_pk for lo: 4 instructions per pair instead of 6; LeakyReLU as a raw v_max_f32 (vmax_raw, gnn_fused_common.h: hipcc wraps
// fmaxf of values it cannot prove quiet in two canonicalising v_max: 2 more per pair).  NaN stays NaN (max(NaN, 0.1 NaN)); 7 instead of 10 per pair.
template <bool F16>
__device__ __forceinline__ void split2(f32x2 v, uint32_t& hi, uint32_t& lo) {
    if constexpr (F16) {
        hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
        lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{sub_f16_lo(v[0], hi), sub_f16_hi(v[1], hi)}, f16x2));
    } else {
        hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
        const f32x2 back = {__uint_as_float(hi << 16), __uint_as_float(hi & 0xFFFF0000u)};
        lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(v - back, bf16x2));
    }
}
template <bool F16>
__device__ __forceinline__ void lrelu_split2(f32x2 v, uint32_t& hi, uint32_t& lo) {
    const f32x2 s = v * LRELU;
    split2<F16>(f32x2{vmax_raw(v[0], s[0]), vmax_raw(v[1], s[1])}, hi, lo);
}

// bias pre-loaded into the accumulators, D = W^T X^T layout: register r of lane l = channel 8 (r >> 2) + 4 (l >> 5) + (r & 3)
__device__ __forceinline__ void acc_init_bias(f32x16 (&acc)[NMB], const float* __restrict__ bias, int wave, int lane) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias + wave * 32 + rg * 8 + (lane >> 5) * 4);
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[mb][rg * 4 + e] = b[e];
    }
}

// conv epilogue: LeakyReLU (the bias is already in the accumulators), split, rows CARRY .. of the output buffer.  A lane
// holds 4 groups of 4 consecutive channels of one row per m-block: 8-byte stores into both planes (rows r and r + 16 share
// their banks, the two lane halves fill the gaps: 128 dwords over 64 banks in the minimal two passes).
template <bool F16>
__device__ __forceinline__ void conv_epilogue(unsigned char* __restrict__ obuf, const f32x16 (&acc)[NMB], int wave, int lane) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const int f0 = wave * 32 + rg * 8 + (lane >> 5) * 4;
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb) {
            uint2 h, l;
            lrelu_split2<F16>(f32x2{acc[mb][rg * 4], acc[mb][rg * 4 + 1]}, h.x, l.x);
            lrelu_split2<F16>(f32x2{acc[mb][rg * 4 + 2], acc[mb][rg * 4 + 3]}, h.y, l.y);
            unsigned char* o = obuf + (CARRY + mb * 32 + (lane & 31)) * ROWX + f0 * 2;
            *reinterpret_cast<uint2*>(o) = h;
            *reinterpret_cast<uint2*>(o + LOX) = l;
        }
    }
}

// MaxPool1D(8) of the y @ w_v tile -> yp rows (igloo.py:209-210): rows 8rg..8rg+3 of a 32-row block sit in lanes 0-31, rows
// 8rg+4..8rg+7 in lanes 32-63, so the 8-row maximum is 4 registers + one exchange with lane ^ 32
__device__ __forceinline__ void wv_pool_store(const f32x16 (&acc)[NMB], wrsrc_t yp_w, int t0, int wave, int lane) {
    float m[4 * NMB];
#pragma unroll
    for (int i = 0; i < 4 * NMB; ++i) {
        const int mb = i >> 2, rg = i & 3;
        const float v = max_nan(max_nan(acc[mb][rg * 4], acc[mb][rg * 4 + 1]), max_nan(acc[mb][rg * 4 + 2], acc[mb][rg * 4 + 3]));
        const unsigned bits = __float_as_uint(v);
        // v_permlane32_swap(v, v) = {[v.lo | v.lo], [v.hi | v.hi]}: their maximum is the 8-row maximum in BOTH lane halves, no select
        const auto sw = __builtin_amdgcn_permlane32_swap(bits, bits, false, false);
        m[i] = max_nan(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    const int q0 = t0 / GNN_POOL;
    const int nq = min(4 * NMB, POOLED - q0);
    if (lane < 32) {
        const uint32_t voff = (uint32_t)(wave * 32 + lane) * 4u;
#pragma unroll
        for (int i = 0; i < 4 * NMB; ++i)
            if (i < nq) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m[i]), yp_w, voff, (q0 + i) * (C * 4), 0);
    }
}

// conv1 + LeakyReLU of one (row, 32-channel block) unit -> both planes of the row: 4 + 4 stores of 16 B
template <bool F16>
__device__ __forceinline__ void store_block32(unsigned char* __restrict__ buf, int buf_row, int blk, const float (&x)[32]) {
    unsigned char* row = buf + buf_row * ROWX + blk * 64;
    uint32_t hi[16], lo[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) split2<F16>(f32x2{x[2 * i], x[2 * i + 1]}, hi[i], lo[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<uint4*>(row + i * 16) = make_uint4(hi[4 * i], hi[4 * i + 1], hi[4 * i + 2], hi[4 * i + 3]);
        *reinterpret_cast<uint4*>(row + LOX + i * 16) = make_uint4(lo[4 * i], lo[4 * i + 1], lo[4 * i + 2], lo[4 * i + 3]);
    }
}
// the lane pair (q = 0, 1) swaps halves of its two rows (gnn_fused_c6.hip, gather_store)
template <bool F16>
__device__ __forceinline__ void gather_store(const GatherSum& g, unsigned char* __restrict__ xbuf, int ua, int pq) {
    const bool odd = pq & 1;
    float x[32];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 sa = g.sa[i], sb = g.sb[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float got = dpp_xor1(odd ? sa[k] : sb[k]);             // even lane keeps row A and gets the partner's row A half
            const float c0 = odd ? got : sa[k], c1 = odd ? sb[k] : got;
            x[4 * i + k] = vmax_raw(c0, c0 * LRELU);                     // channels 0..15 of the block
            x[16 + 4 * i + k] = vmax_raw(c1, c1 * LRELU);                // channels 16..31
        }
    }
    store_block32<F16>(xbuf, CARRY + ua + (odd ? 1 : 0), pq >> 1, x);
}
template <bool F16>
__device__ __forceinline__ void gather_finish(const GatherUnit& g, unsigned char* __restrict__ xbuf, int ua, int pq) {
    GatherSum t;
    gather_sum(t, g);
    gather_store<F16>(t, xbuf, ua, pq);
}

// dot product of the unit's 32 folded weights with block p of row u (x = hi + lo), summed over the entry's 4 lanes
template <bool F16>
struct PairCompute {
    static __device__ __forceinline__ void run(const PairW& w, const PairJob& jb, int e, int u, int p) {
        const unsigned char* xr = jb.xbuf + (CARRY + u - jb.t0) * ROWX + p * 64;
        uint4 hx[4], lx[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            hx[i] = *reinterpret_cast<const uint4*>(xr + i * 16);
            lx[i] = *reinterpret_cast<const uint4*>(xr + LOX + i * 16);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t hv[4] = {hx[i].x, hx[i].y, hx[i].z, hx[i].w}, lv[4] = {lx[i].x, lx[i].y, lx[i].z, lx[i].w};
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
                const float4 w0 = w.w[2 * i + (k >> 1)];          // channels 8 i + 2 k, + 1 of the block
                s = fmaf(x0, (k & 1) ? w0.z : w0.x, s);
                s = fmaf(x1, (k & 1) ? w0.w : w0.y, s);
            }
        }
        s += dpp_xor1(s);
        s += dpp_xor2(s);
        if (p == 0) jb.mp[e] = s;
    }
};

// Barriers B1..B4 per step as in gnn_fused_c6.hip:
//   matrix : w_v A(s), conv2 loop [bufX] | B1 | epilogue -> bufY (x2) | B2 | conv3 loop [bufY] | B3 |
//            epilogue -> bufY (x3) | B4 | w_v B(s) [bufY]   -> straight into step s+1
//   helpers: pair products B(s-1) [bufY] and A(s) [bufX], gather(s+1) table loads | B1 | x1 carry rows | B2 |
//            x1(s+1) -> bufX (both halves), read x2 carry | B3 | x2 carry rows -> bufY, pair rows of step s+2 | B4
//
// Time split (a.split > 1; launches with fewer windows than the chip has CUs, e.g. the reference's own call shape of 128 windows
// per predict, nn_classification.py:316-317): a window's steps are dealt to a.split workgroups in contiguous runs.  A workgroup
// whose run starts at step s_lo > 0 first executes step s_lo - 1 as a WARM-UP: everything is computed, nothing is stored (no yp
// rows, no pair products).  Its carry rows start as zeros, so the first 5 rows of x2 and the first 10 of x3 of the warm-up step
// are wrong - and unused: what step s_lo reads from it are the LAST 5 rows of x1 (a function of the bases alone) and of x2 (a
// function of x1 rows at least 118 rows into the warm-up step).  Every stored row goes through the same instruction sequence
// with the same operands as in the one-workgroup launch: bit-identical, one extra step per additional workgroup.
template <bool F16, bool PROF>
__global__ __launch_bounds__(512, 2) void fused_front_x3_kernel(Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEMX];
    unsigned char* bufX = smem;
    unsigned char* bufY = smem + BUFX_BYTES;
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
    const int woff = hw * WNBLK_B;                       // this wave's n-block inside every k16 unit

    // carry rows of the first step = the causal zero padding
    for (int i = tid; i < CARRY * ROW_U4; i += 512) {
        reinterpret_cast<uint4*>(bufX)[i] = make_uint4(0, 0, 0, 0);
        reinterpret_cast<uint4*>(bufY)[i] = make_uint4(0, 0, 0, 0);
    }
    if (tid >= 256) bias_s[tid - 256] = a.conv_b[(tid - 256) >> 7][tid & 127];
    // padding skip (see gnn_fused_c6.hip): steps made only of rows t >= p + 15, p = first position behind which every base
    // is non-ACGT, carry the values of an all-N window and are copied instead of computed
    if (tid == 0) *s_last = -1;
    __syncthreads();
    if (a.yp_c) {
        int last = -1;
        for (int i = tid * 12; i < tid * 12 + 12 && i < W; ++i)
            if (base_code_f(bases[i]) >= 0) last = i;
        if (last >= 0) atomicMax(s_last, last);
    }
    __syncthreads();
    const int nsteps = a.yp_c ? max(1, min(STEPSX, (*s_last + 1 + 15 + FTX - 1) / FTX)) : STEPSX;
    // this workgroup's run of steps [s_lo, s_hi) and the step it starts executing at (one warm-up step for every run but the first)
    const int per = (nsteps + a.split - 1) / a.split;
    const int s_lo = min(part * per, nsteps), s_hi = min(s_lo + per, nsteps);
    const int s_begin = s_hi > s_lo ? (s_lo > 0 ? s_lo - 1 : 0) : s_hi;      // an empty run (short window, padding skip) executes nothing
    // pair rows of the first two steps executed: prow2(s & 1)[i] = pair row of positions (t0 - 5 + i, t0 - 4 + i), t0 = s * FTX
    if (tid < PROW_N) {
#pragma unroll
        for (int s01 = 0; s01 < 2; ++s01) {
            uint32_t lo, hi;
            const int t = (s_begin + s01) * FTX - CARRY + tid;
            prow_fetch(bases, t, lo, hi);
            prow2((s_begin + s01) & 1)[tid] = prow_make(lo, hi, t);
        }
    }
    __syncthreads();
    unsigned long long cyc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tick_ = 0;
    const int gpq = ht & 7, gua = (ht >> 3) * 2;         // conv1 gather: 2 * block + channel half, first row of the lane pair (+ 64 for the second round)

    if (!helper) {
        __builtin_amdgcn_s_setprio(2);
        const wrsrc_t cw[2] = {make_wrsrc(a.conv_w[0], KS * 8 * WUNIT_B), make_wrsrc(a.conv_w[1], KS * 8 * WUNIT_B)};
        const wrsrc_t vw[2] = {make_wrsrc(a.wv_w[0], 8 * WUNIT_B), make_wrsrc(a.wv_w[1], 8 * WUNIT_B)};
        const wrsrc_t yp_w[2] = {make_wrsrc(reinterpret_cast<const unsigned char*>(a.yp + (wi * 2 + 0) * (size_t)POOLED * C), POOLED * C * 4),
                                 make_wrsrc(reinterpret_cast<const unsigned char*>(a.yp + (wi * 2 + 1) * (size_t)POOLED * C), POOLED * C * 4)};
        WU ring[RINGW];
        prefetch_w(ring, vw[0], woff, lane);
        __syncthreads();                                                         // x1 of step 0 is in bufX
        if constexpr (PROF) tick_ = __builtin_readcyclecounter();
#pragma unroll 1
        for (int step = s_begin; step < s_hi; ++step) {
            const int t0 = step * FTX;
            const bool store = step >= s_lo;                                     // false in the warm-up step of a time-split run
            GNN_TICK(7)
            f32x16 acc[NMB];
#pragma unroll
            for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
            gemm_tile<false, F16, 1, KS>(smem, CARRY * ROWX, vw[0], cw[0], woff, ring, acc, lane);
            if (store) wv_pool_store(acc, yp_w[0], t0, hw, lane);
            GNN_TICK(0)
            acc_init_bias(acc, bias_s, hw, lane);
#ifndef GNN_ABL_NOCONV2
            gemm_tile<true, F16, KS, KS>(smem, 0, cw[0], cw[1], woff, ring, acc, lane);
#else       // measurement only (wrong results): the launch without conv2's MFMAs - what a table-lookup conv2 would leave on the matrix pipe
            prefetch_w(ring, cw[1], woff, lane);
#endif
            GNN_TICK(1)
            __syncthreads();                                                     // ---- B1
            GNN_TICK(2)
            conv_epilogue<F16>(bufY, acc, hw, lane);
            __syncthreads();                                                     // ---- B2
            GNN_TICK(3)
            acc_init_bias(acc, bias_s + C, hw, lane);
            gemm_tile<true, F16, KS, 1>(smem, BUFX_BYTES, cw[1], vw[1], woff, ring, acc, lane);
            GNN_TICK(4)
            __syncthreads();                                                     // ---- B3
            GNN_TICK(5)
            conv_epilogue<F16>(bufY, acc, hw, lane);
            __syncthreads();                                                     // ---- B4
            GNN_TICK(6)
#pragma unroll
            for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
            gemm_tile<false, F16, 1, 1>(smem, BUFX_BYTES + CARRY * ROWX, vw[1], vw[0], woff, ring, acc, lane);
            if (store) wv_pool_store(acc, yp_w[1], t0, hw, lane);
        }
    } else {
        {
            GatherUnit g;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                gather_issue(g, prow2(s_begin & 1), a.conv1_k, gua + 64 * k, gpq);
                gather_finish<F16>(g, bufX, gua + 64 * k, gpq);
            }
        }
        uint32_t nlo = 0, nhi = 0;                       // bytes of this thread's pair row of the step AFTER next
        if (ht < PROW_N) prow_fetch(bases, (s_begin + 2) * FTX - CARRY + ht, nlo, nhi);
        __syncthreads();
        if constexpr (PROF) tick_ = __builtin_readcyclecounter();
#pragma unroll 1
        for (int step = s_begin; step < s_hi; ++step) {
            const int t0 = step * FTX;
            const uint16_t* prow = prow2((step + 1) & 1);                        // pair rows of the next step: written before B4 of the previous one
            GNN_TICK(10)
            {
                // head B's entries of step s-1 belong to this run if s-1 is one of its steps (not the warm-up step, whose x3 rows
                // are not valid everywhere: the previous run computes them after its last step); head A's if s is
                const bool hb = step - 1 >= s_lo, ha = step >= s_lo;
                const int sb = max(step - 1, 0);
                const PairJob jb = {bufY, a.weff[1], a.pos_sorted[1], mp_w[1], t0 - FTX,
                                    hb ? a.bucket_ptr[1][sb] : 0, hb ? a.bucket_ptr[1][sb + 1] : 0};
                const PairJob ja = {bufX, a.weff[0], a.pos_sorted[0], mp_w[0], t0, ha ? a.bucket_ptr[0][step] : 0, ha ? a.bucket_ptr[0][step + 1] : 0};
#ifndef GNN_ABL_NOHELP
                m_partials2<PairCompute<F16>>(jb, ja, hw, lane);
#endif
            }
            uint4 carry = make_uint4(0, 0, 0, 0);
            const int cr = ht / ROW_U4, cc = ht - cr * ROW_U4;   // 5 rows x 33 chunks of 16 B
            if (ht < CARRY * ROW_U4) carry = *reinterpret_cast<const uint4*>(bufX + (FTX + cr) * ROWX + cc * 16);
            // conv1 gather of the next step: table loads requested before B1, the first half converted in the window in which
            // the matrix waves are in their conv2 epilogue (no MFMA in flight on the CU), the second beside the conv3 loop
#ifndef GNN_ABL_NOHELP
            GatherUnit g0, g1;
            gather_issue(g0, prow, a.conv1_k, gua, gpq);
            gather_issue(g1, prow, a.conv1_k, gua + 64, gpq);
#endif
            GNN_TICK(8)
            __syncthreads();                                                     // ---- B1
            GNN_TICK(11)
            if (ht < CARRY * ROW_U4) *reinterpret_cast<uint4*>(bufX + cr * ROWX + cc * 16) = carry;
#if !defined(GNN_ABL_NOHELP) && defined(GNN_X3_GATHER_EARLY)
            // measurement variant (the f16c6 placement): first half converted between B1 and B2 - the matrix waves then wait
            // at B2 for the table loads' latency (conv2 epilogue + B2: 3.9 k instead of 2 k cycles per step)
            gather_finish<F16>(g0, bufX, gua, gpq);
            GatherSum s1;
            gather_sum(s1, g1);
#endif
            GNN_TICK(12)
            __syncthreads();                                                     // ---- B2
            GNN_TICK(13)
#ifndef GNN_ABL_NOHELP
            // both halves of x1(s+1) beside the conv3 loop: bufX is free from B1 on, the helpers have nothing else to do here
            // (they waited 17 k cycles per step at B3), and this kernel's row conversion is light enough (two packed
            // conversions per pair, no block maxima, no fp6 images) to run beside the MFMA stream
#ifndef GNN_X3_GATHER_EARLY
            gather_finish<F16>(g0, bufX, gua, gpq);
            GatherSum s1;
            gather_sum(s1, g1);
#endif
            gather_store<F16>(s1, bufX, gua + 64, gpq);
#endif
            if (ht < CARRY * ROW_U4) carry = *reinterpret_cast<const uint4*>(bufY + (FTX + cr) * ROWX + cc * 16);
            GNN_TICK(9)
            __syncthreads();                                                     // ---- B3
            GNN_TICK(14)
            if (ht < CARRY * ROW_U4) *reinterpret_cast<uint4*>(bufY + cr * ROWX + cc * 16) = carry;
            // pair rows of step s+2 into the buffer step s's rows were in (last read before B1 of step s-1), from the bytes
            // requested a step ago; then the request for the step after
            if (ht < PROW_N) {
                const int t = t0 + 2 * FTX - CARRY + ht;
                prow2(step & 1)[ht] = prow_make(nlo, nhi, t);
                prow_fetch(bases, t + FTX, nlo, nhi);
            }
            GNN_TICK(15)
            __syncthreads();                                                     // ---- B4
            if constexpr (PROF) tick_ = __builtin_readcyclecounter();
        }
        if (s_hi > s_lo) {                                  // head B's entries of this run's last step
            const PairJob jb = {bufY, a.weff[1], a.pos_sorted[1], mp_w[1], (s_hi - 1) * FTX, a.bucket_ptr[1][s_hi - 1], a.bucket_ptr[1][s_hi]};
            const PairJob none = {bufX, a.weff[0], a.pos_sorted[0], mp_w[0], 0, 0, 0};
            m_partials2<PairCompute<F16>>(jb, none, hw, lane);
        }
    }
    if (nsteps < STEPSX && part == a.split - 1) {   // the all-N tail: copy instead of compute (disjoint from what the steps above wrote)
        const int q0 = nsteps * (FTX / GNN_POOL);
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

static void fill_args(const gnn_ctx* ctx, Args& a, const uint8_t* bases, bool f16) {
    const DeviceWeights& d = ctx->w;
    a.bases = bases;
    a.conv1_k = d.conv1_pairs6;
    for (int i = 0; i < 2; ++i) {
        a.conv_w[i] = reinterpret_cast<const unsigned char*>(f16 ? d.conv_frag_h[i] : d.conv_frag[i]);
        a.conv_b[i] = d.conv_b[i];
        a.wv_w[i] = reinterpret_cast<const unsigned char*>(f16 ? d.wv_frag_h[i] : d.wv_frag[i]);
        a.weff[i] = d.weff6[i];
        a.pos_sorted[i] = d.pos_sorted[i];
        a.bucket_ptr[i] = d.bucket_ptr[i];
    }
    a.cycles = nullptr;
    a.split = 1;
}

static void launch(const Args& a, bool f16, bool prof, unsigned nwin, hipStream_t stream) {
    const unsigned n = nwin * (unsigned)a.split;
    if (f16) {
        if (prof) hipLaunchKernelGGL((fused_front_x3_kernel<true, true>), dim3(n), dim3(512), 0, stream, a);
        else hipLaunchKernelGGL((fused_front_x3_kernel<true, false>), dim3(n), dim3(512), 0, stream, a);
    } else {
        if (prof) hipLaunchKernelGGL((fused_front_x3_kernel<false, true>), dim3(n), dim3(512), 0, stream, a);
        else hipLaunchKernelGGL((fused_front_x3_kernel<false, false>), dim3(n), dim3(512), 0, stream, a);
    }
}

}  // namespace x3

// The all-N window's outputs of both limb formats, computed once by the kernel itself (padding skip).  Called by
// gnn_load_weights after pack_fused_weights and pack_fused_c6_weights (whose pair tables and folded IGLOO weights this kernel
// shares).
int pack_fused_x3_consts(gnn_ctx* ctx) {
    using namespace x3;
    DeviceWeights& d = ctx->w;
    void* bn = nullptr;
    GNN_HIP(hipMalloc(&bn, W));
    ctx->owned.push_back(bn);
    GNN_HIP(hipMemsetAsync(bn, 'N', W, ctx->stream));
    for (int m = 0; m < 2; ++m) {              // 0: bf16 limbs, 1: f16 limbs
        void *yc = nullptr, *mc = nullptr;
        GNN_HIP(hipMalloc(&yc, (size_t)2 * POOLED * C * sizeof(float)));
        ctx->owned.push_back(yc);
        GNN_HIP(hipMalloc(&mc, (size_t)2 * NPAIR * sizeof(float)));
        ctx->owned.push_back(mc);
        Args a;
        fill_args(ctx, a, static_cast<const uint8_t*>(bn), m == 1);
        a.mp = static_cast<float*>(mc);
        a.yp = static_cast<float*>(yc);
        a.yp_c = nullptr;
        a.mp_c = nullptr;
        launch(a, m == 1, false, 1, ctx->stream);
        GNN_HIP(hipGetLastError());
        GNN_HIP(hipStreamSynchronize(ctx->stream));
        d.x3_yp_const[m] = static_cast<float*>(yc);
        d.x3_mp_const[m] = static_cast<float*>(mc);
    }
    return GNN_OK;
}

int launch_front_x3(gnn_ctx* ctx, const uint8_t* bases, int64_t n, int precision) {
    using namespace x3;
    if (reinterpret_cast<uintptr_t>(bases) & 3u) {
        set_error("f16x3 / bf16x3: the window buffer must be 4-byte aligned");
        return GNN_ERR_ARG;
    }
    const bool f16 = precision == GNN_PREC_F16X3;
    Args a;
    fill_args(ctx, a, bases, f16);
    a.mp = ctx->ws.mp;
    a.yp = ctx->ws.yp;
    a.yp_c = ctx->c6_pad_skip ? ctx->w.x3_yp_const[f16] : nullptr;
    a.mp_c = ctx->c6_pad_skip ? ctx->w.x3_mp_const[f16] : nullptr;
    a.cycles = ctx->phase_cycles;
    // fewer windows than CUs (one workgroup owns a CU): deal every window's steps to several workgroups.  Capped at 4: each
    // additional workgroup recomputes one step of 47, and below 12 steps per run the prologue starts to show
    if (ctx->time_split && n > 0 && ctx->cu_count > 0) a.split = (int)std::max<int64_t>(1, std::min<int64_t>(4, ctx->cu_count / n));
    ctx->last_split = a.split;
    launch(a, f16, ctx->phase_cycles != nullptr, (unsigned)n, ctx->stream);
    GNN_HIP(hipGetLastError());
    return GNN_OK;
}

}  // namespace gnn
