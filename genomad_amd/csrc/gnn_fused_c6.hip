// Fused front end, GNN_PREC_F16C6: the streaming structure of gnn_fused_c8.hip (one workgroup = one window,
// steps of FT6 positions, activations resident in LDS, 4 matrix waves + 4 helper waves) with the two
// correction products on the MX-fp6 (e2m3) rate of the matrix pipe: 1.5 instead of 2.0 f16-pass equivalents.
//
//   x * w  ~=  f16(x) * f16(w)                           v_mfma_f32_32x32x16_f16, K = 16 per instruction (32 cycles)
//            + e2m3_mx(x) * e2m3_mx(w - f16(w))          \  one v_mfma_scale_f32_32x32x64_f8f6f4 (cbsz = blgp = 2: fp6,
//            + e2m3_mx(x - f16(x)) * e2m3_mx(w)          /  32 cycles instead of the 64 of fp8) per 32 channels
//
// e2m3 has the 4 significant bits of e4m3 but only 2 exponent bits, so BOTH operands are block scaled (OCP MX: 32
// consecutive K elements share one E8M0 exponent): weights per (32 k, output column) at packing time, activations per
// (row, 32 channels) when a row is produced — the largest |x| (and the largest residual) of the block puts its
// exponent into the scale byte the MFMA reads with the fragment.  Emulation (oracle/precision_study.py, "fp16 + e2m3
// (fp6, MX both sides)"): max |dscore| 3.7e-5 / 5.7e-5 on the two weight seeds, the class of f16c8 (4.1e-5 / 6.6e-5).
//
// Everything that touches an activation row works on (row, 32-channel block) units held by ONE lane, because that is
// the granularity of the hardware's fp6 conversions (v_cvt_scalef32_pk32_fp6_f16: 32 values -> 24 bytes, natural
// order: scripts/probe_mx.hip) and of the MX scale: the conv epilogues bring a row's 32 channels together with 16
// v_permlane32_swap, the conv1 gather works on lane pairs (each lane fetches half a block of two neighbouring rows, the
// halves are swapped with DPP moves) and the IGLOO pair products are mapped 4 lanes per row.
//
// Operand layout facts (scripts/probe_mx.hip -> profiles/history/r02_probe_mx.txt): an fp6 operand of 32x32x64 is 6 dwords
// per lane: lane l = row (or column) l & 31, K block l >> 5, element i of the block in bits [6i, 6i+6); the scale of
// block 0 is read from lanes 0-31, that of block 1 from lanes 32-63, byte OPSEL of the scale VGPR.
//
// LDS row (464 B = 29 x 16: odd multiple of 16 B keeps the 16-lane groups of ds_read_b128 on distinct slots):
//   [0,256) 128 ch f16 | [256,320) x6: first 16 B of the 4 blocks | [320,352) x6: last 8 B of the 4 blocks |
//   [352,416) [416,448) the same for the residual image | [448,452) E8M0 of x6 per block | [452,456) E8M0 of the
//   residual image per block | pad
// The narrower row (464 instead of 528 B) is what lets a step hold FT6 = 160 rows (GNN_C6_NMB = 5) in 160 KB.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <utility>

#include "gnn_fused_common.h"
#include "gnn_fused_helpers.h"

namespace gnn {
namespace c6 {

#ifndef GNN_C6_NMB
#define GNN_C6_NMB 4
#endif
constexpr int NMB = GNN_C6_NMB;              // 32-row blocks per step
constexpr int FT6 = 32 * NMB;                // rows per step
constexpr int STEPS6 = (T + FT6 - 1) / FT6;
constexpr int ROW6 = 464;
constexpr int X6A = 256, X6B = 320, L6A = 352, L6B = 416, SXO = 448, SLO = 452;
constexpr int BUF6_ROWS = CARRY + FT6;
constexpr int BUF6_BYTES = BUF6_ROWS * ROW6;
constexpr int PROW_OFF = 2 * BUF6_BYTES;
constexpr int PROW_N = FT6 + 4;              // pair rows a step's conv1 gather reads
constexpr int PROW_BYTES = ((PROW_N * 2 + 15) / 16) * 16;
constexpr int BIAS_OFF = PROW_OFF + 2 * PROW_BYTES;                   // two pair-row buffers (step parity), then conv2 | conv3 bias, 2 x 128 f32
constexpr int LAST_OFF = BIAS_OFF + 2 * C * 4;                        // index of the window's last ACGT base
constexpr int SMEM6 = LAST_OFF + 16;
constexpr int WNBLK_B = 3584;                // weight bytes per (k32 step, n-block): f16 k16 even | f16 k16 odd | fp6 16-B parts | fp6 8-B parts
constexpr int WSTEP_B = 4 * WNBLK_B;         // per k32 step
constexpr int ROW_U4 = ROW6 / 16;            // 29
#ifndef GNN_C6_RING
#define GNN_C6_RING 4
#endif
constexpr int RING = GNN_C6_RING;            // weight ring slots: RING - 1 k32 steps of weights are in flight ahead of the MFMAs
// 8-B parts: slot (block ^ bit 4 of the buffer row) — with the 464-B stride rows R and R+16 start on the same bank, and a
// ds_read_b64 serves 32 consecutive rows per LDS cycle; swapping neighbouring slots in every other group of 16 rows makes
// those reads conflict free.  GNN_C6_NOSWZ: measurement variant.
#ifdef GNN_C6_NOSWZ
__device__ __forceinline__ int swz(int) { return 0; }
#else
__device__ __forceinline__ int swz(int buf_row) { return (buf_row >> 4) & 1; }
#endif
static_assert(SMEM6 <= 160 * 1024, "LDS budget");
static_assert(FT6 % 32 == 0, "carry-row copies keep bit 4 of the row index");
static_assert(PROW_N <= 256, "one helper thread per pair row");
static_assert(FT6 % 2 == 0 && FT6 >= 128, "the conv1 gather works on row pairs, two rounds of 64 rows per step at least");

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x6 __attribute__((ext_vector_type(6)));

struct Args {
    const uint8_t* bases;
    const float* conv1_k;             // (3, PAIR_ROWS, 128) f32 pair tables in the gather's lane order (GatherUnit), bias folded into table 0
    const unsigned char* conv_w[2];   // [k32 step 24][nblk 4][3584 B], then [tap 6][nblk 4][lane 64] u32: byte j = E8M0 scale of k32 step 4*tap + j
    const float* conv_b[2];
    const unsigned char* wv_w[2];     // same layouts, 4 k32 steps / 1 tap
    const float* weff[2];
    const int32_t* pos_sorted[2];
    const int32_t* bucket_ptr[2];     // (STEPS6 + 1,) entry ranges per step of FT6 rows
    float* mp;
    float* yp;
    // outputs of an all-N window (pack_fused_c6_weights runs the kernel on one at load time): the yp rows and pair products of
    // a window's all-N tail are copied from here instead of being recomputed; nullptr = compute everything
    const float* yp_c;
    const float* mp_c;
    unsigned long long* cycles;       // PROF builds: 16 phase counters (GNN_TICK in gnn_fused_common.h), summed over workgroups:
                                      // matrix wave 0 -> 0..7, helper wave 4 -> 8..15 (names in scripts/c6_check.py)
};

struct WStep {
    uint4 h0, h1;      // f16 fragments of the two k16 halves
    uint4 c0;          // fp6 fragment dwords 0-3 (lanes 0-31: K block 0 = e2m3_mx(w - f16 w), lanes 32-63: K block 1 = e2m3_mx(w))
    uint2 c1;          // fp6 fragment dwords 4-5
};
struct XF {
    uint4 v[2][NMB];   // [k16 half][m-block]
};
struct XC {
    i32x8 v[NMB];      // fp6 fragment dwords 0-5 (lanes 0-31: x image, lanes 32-63: residual image), 6-7 unused
};

__device__ __forceinline__ void load_w_h(WStep& w, wrsrc_t r, uint32_t l16, int soff) {
    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, l16, soff, 0);
    const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(r, l16 + 1024, soff, 0);
    w.h0 = make_uint4(a[0], a[1], a[2], a[3]);
    w.h1 = make_uint4(b[0], b[1], b[2], b[3]);
}
__device__ __forceinline__ void load_w_c(WStep& w, wrsrc_t r, uint32_t l16, int soff) {
    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, l16 + 2048, soff, 0);
#if defined(GNN_ABL_NOC1)      // measurement variants: is the cost of a weight load its bytes or its instruction?
    const u32x2 b = {a[0], a[1]};
#elif defined(GNN_ABL_C1X4)
    const u32x4 b4 = __builtin_amdgcn_raw_buffer_load_b128(r, l16 + 2560, soff, 0);
    const u32x2 b = {b4[0], b4[1]};
#else
    const u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(r, (l16 >> 1) + 3072, soff, 0);
#endif
    w.c0 = make_uint4(a[0], a[1], a[2], a[3]);
    w.c1 = make_uint2(b[0], b[1]);
}
// xh = lane base of the tap row's f16 plane (row l & 31, + 16 B for lanes 32-63); J = k32 step inside the tap
template <int J, int TOFF>   // TOFF = tap * ROW6: every offset is an immediate of the DS instruction
__device__ __forceinline__ void load_xf(XF& f, const unsigned char* __restrict__ xh) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb) f.v[s][mb] = *reinterpret_cast<const uint4*>(xh + TOFF + mb * 32 * ROW6 + (J * 2 + s) * 32);
}
// xq = lane base of the tap row's fp6 planes (x image for lanes 0-31, residual image for lanes 32-63)
// x8[mb] = the 8-B parts of m-block mb's tap row, + 8 * swz(row) for even J / - 8 * swz(row) for odd J (slot J is then the
// row's swizzled slot).  One opaque pointer per m-block: with a common base the compiler fuses the b64 reads of two m-blocks
// into one ds_read2st64_b64, whose four result registers then have to be MOVED behind the two b128 parts to form the 6-register
// MFMA operands - 8 v_mov per k32 step, 14 % of all VALU instructions of the workgroup.
template <int J, int TOFF>
__device__ __forceinline__ void load_xc(XC& f, const unsigned char* __restrict__ xq, const unsigned char* const (&x8)[NMB]) {
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
        const uint4 a = *reinterpret_cast<const uint4*>(xq + TOFF + mb * 32 * ROW6 + J * 16);
        const uint2 b = *reinterpret_cast<const uint2*>(x8[mb] + J * 8);
        f.v[mb] = i32x8{(int)a.x, (int)a.y, (int)a.z, (int)a.w, (int)b.x, (int)b.y, 0, 0};
    }
}
// activation scale words of the tap row (4 bytes = the row's 4 blocks; OPSEL picks the k32 step's)
template <int TOFF>
__device__ __forceinline__ void load_sx(int (&sx)[NMB], const unsigned char* __restrict__ xs) {
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) sx[mb] = *reinterpret_cast<const int*>(xs + TOFF + mb * 32 * ROW6);
}

template <bool SWAP>
__device__ __forceinline__ void mfma_f16_phase(const WStep& w, const XF& x, f32x16 (&acc)[NMB]) {
    const f16x8 w0 = __builtin_bit_cast(f16x8, w.h0), w1 = __builtin_bit_cast(f16x8, w.h1);
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
        const f16x8 xv = __builtin_bit_cast(f16x8, x.v[0][mb]);
        acc[mb] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, xv, acc[mb], 0, 0, 0)
                       : __builtin_amdgcn_mfma_f32_32x32x16_f16(xv, w0, acc[mb], 0, 0, 0);
    }
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
        const f16x8 xv = __builtin_bit_cast(f16x8, x.v[1][mb]);
        acc[mb] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, xv, acc[mb], 0, 0, 0)
                       : __builtin_amdgcn_mfma_f32_32x32x16_f16(xv, w1, acc[mb], 0, 0, 0);
    }
}

// J selects the byte of both scale words (OPSEL): the weight word holds the 4 k32 steps of the tap, the activation
// word the 4 blocks of the row
template <bool SWAP, int J>
__device__ __forceinline__ void mfma_c6_phase(const WStep& w, const XC& x, int ws, const int (&sx)[NMB], f32x16 (&acc)[NMB]) {
    const i32x8 wv = {(int)w.c0.x, (int)w.c0.y, (int)w.c0.z, (int)w.c0.w, (int)w.c1.x, (int)w.c1.y, 0, 0};
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
        const i32x8 xv = x.v[mb];
        acc[mb] = SWAP ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wv, xv, acc[mb], 2, 2, J, ws, J, sx[mb])
                       : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xv, wv, acc[mb], 2, 2, J, sx[mb], J, ws);
    }
}

// One k32 step = two scheduling regions.  Region F: the 2*NMB f16 MFMAs of step j, with the fp6 activation fragments
// of the SAME step (2*NMB LDS reads, + the tap's NMB scale words when J == 0) and the f16 weight fragments of step
// j+3 (2 L2 loads) issued between them.  Region C: the NMB fp6 MFMAs (32 cycles each), with the f16 activation
// fragments of step j+1 and the fp6 weight fragment of step j+3.
template <bool SWAP, int J, int JN, bool LW, bool LX, int TOFF, int TOFFN>
__device__ __forceinline__ void k32_step(const WStep& wcur, WStep& wload, XF& xf, XC& xc, int (&sx)[NMB],
                                         const unsigned char* __restrict__ xh, const unsigned char* __restrict__ xq,
                                         const unsigned char* const (&x8)[NMB], const unsigned char* __restrict__ xs, wrsrc_t wr, int wnext,
                                         uint32_t l16, int ws, f32x16 (&acc)[NMB]) {
    // GNN_ABL_*: measurement-only ablations (scripts/mkvariant.sh) that compile parts of the work out — wrong results
    // by construction, used to see what the launch time is made of (profiles/README.md)
#ifndef GNN_ABL_NOX
    load_xc<J, TOFF>(xc, xq, x8);
    if constexpr (J == 0) load_sx<TOFF>(sx, xs);
#endif
#ifndef GNN_ABL_NOW
    if constexpr (LW) load_w_h(wload, wr, l16, wnext);
#endif
#ifndef GNN_ABL_NOF16
    mfma_f16_phase<SWAP>(wcur, xf, acc);
#endif
#pragma unroll
    for (int i = 0; i < 2 * NMB; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                           // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                           // 1 DS read
        if (J == 0 && i < NMB) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // + the tap's scale words
        if (LW && (i == 1 || i == 5)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
    }
    GNN_REGION_END();
#ifndef GNN_ABL_NOX
    if constexpr (LX) load_xf<JN, TOFFN>(xf, xh);
#endif
#ifndef GNN_ABL_NOW
    if constexpr (LW) load_w_c(wload, wr, l16, wnext);
#endif
#ifndef GNN_ABL_NOC6
    mfma_c6_phase<SWAP, J>(wcur, xc, ws, sx, acc);
#endif
#pragma unroll
    for (int i = 0; i < NMB; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (LX) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        if (LW && (i == 0 || i == 2)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    GNN_REGION_END();
}

// The first RING - 1 k32 steps of a tile's weights + its first scale word are loaded BEFORE the epilogue and barrier that
// precede the tile (prefetch_w), and inside the tile every step requests the weights of the step RING - 1 ahead into the
// slot the previous step just released.  Measured with 2 ... 6 slots (profiles/README.md): the launch time does not depend
// on the depth — the weight stream is not latency bound — so the default keeps the four slots of gnn_fused_c8.hip (with
// the stream chained through the tile boundaries the depth has to divide the 4 steps of a tap: 2 or 4).
struct WRing {
    WStep w[RING];
    int ws;
};
template <int NTAPS>
__device__ __forceinline__ void prefetch_w(WRing& r, wrsrc_t wr, int woff, int hw, int lane) {
    const uint32_t l16 = (uint32_t)lane * 16u;
    constexpr int NP = NTAPS * 4 < RING - 1 ? NTAPS * 4 : RING - 1;
    static_for(std::make_integer_sequence<int, NP>{}, [&](auto sc) {
        constexpr int st = decltype(sc)::value;
        load_w_h(r.w[st], wr, l16, woff + st * WSTEP_B);
        load_w_c(r.w[st], wr, l16, woff + st * WSTEP_B);
    });
    r.ws = (int)__builtin_amdgcn_raw_buffer_load_b32(wr, (uint32_t)lane * 4u, NTAPS * 4 * WSTEP_B + hw * 256, 0);   // scale word of tap 0
    asm volatile("" ::: "memory");
}

// FT6 rows x 32 columns, K = NTAPS * 128, as NTAPS * 4 fully unrolled k32 steps (static ring slots, static LDS offsets).
// SWAP: D = W^T X^T for the convs (a lane ends up with 16 channels of one row), D = X W for y @ w_v (a lane ends up with
// 16 rows of one channel: the max-pool is register local).
// The weight stream runs THROUGH the tile boundaries: the last RING - 1 steps of a tile request the first RING - 1 steps (and
// the first scale word) of the NEXT tile (wr_next, NTAPS_NEXT), so a tile never starts behind the L2 round trip of its own
// first weights (GNN_C6_NOCHAIN: measurement variant that requests them after the tile instead, as gnn_fused_c8.hip does).
template <bool SWAP, int NTAPS, int ROW0, int NTAPS_NEXT>   // ROW0 = buffer row of the tile's first row (the swizzle needs absolute rows)
__device__ __forceinline__ void gemm_tile(const unsigned char* __restrict__ smem, int xoff, wrsrc_t wr, wrsrc_t wr_next, int woff,
                                          int hw, WRing& ring, f32x16 (&acc)[NMB], int lane) {
    constexpr int NK = NTAPS * 4;
    static_assert(NK % RING == 0 && NTAPS_NEXT * 4 >= RING - 1, "the next tile's first steps land in the slots it expects them in");
    // The lane's row offset is made opaque to the compiler: otherwise it keeps ONE base register for both activation
    // buffers and spends a v_add per LDS read whose folded offset (buffer + m-block + plane) passes the 16-bit DS offset
    // field - about 20 VALU issues per k32 step in a loop that is bound by issue slots.
    uint32_t rowoff = (uint32_t)xoff + (uint32_t)(lane & 31) * ROW6;
    asm volatile("" : "+v"(rowoff));
    const unsigned char* xrow = smem + rowoff;
    const unsigned char* xh = xrow + (lane >> 5) * 16;
    const unsigned char* xq = xrow + X6A + (lane >> 5) * (L6A - X6A);
    const unsigned char* xs = xrow + SXO + (lane >> 5) * (SLO - SXO);
    const uint32_t l16 = (uint32_t)lane * 16u, lane_s = (uint32_t)lane * 4u;
    XF xf;
    XC xc;
    int sx[NMB];
    int ws = ring.ws, ws_next = 0;
    const unsigned char *te[NMB], *to[NMB];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) te[mb] = to[mb] = xq;
    load_xf<0, 0>(xf, xh);
    GNN_REGION_END();
    static_for(std::make_integer_sequence<int, NK>{}, [&](auto kc) {
        constexpr int k = decltype(kc)::value, tap = k / 4, J = k % 4, TOFF = tap * ROW6;
        constexpr int kn = k + 1, TOFFN = (kn / 4) * ROW6, JN = kn % 4;
#ifdef GNN_C6_NOCHAIN
        constexpr bool LW = k + RING - 1 < NK, LX = kn < NK, NEXT = false;
#else
        constexpr bool LW = true, LX = kn < NK, NEXT = k + RING - 1 >= NK;   // NEXT: this step's request belongs to the next tile
#endif
        if constexpr (J == 0) {
            // 8-B parts of this tap's rows: slot J for even J (te) / odd J (to), see swz()
            const int sw8 = swz((lane & 31) + tap + ROW0) * 8;
#pragma unroll
            for (int mb = 0; mb < NMB; ++mb) {
                uint32_t oe = rowoff + X6B + (lane >> 5) * (L6A - X6A) + TOFF + mb * 32 * ROW6 + sw8, oo = oe - 2 * sw8;
                asm volatile("" : "+v"(oe));
                asm volatile("" : "+v"(oo));
                te[mb] = smem + oe;
                to[mb] = smem + oo;
            }
            if constexpr (tap > 0) ws = ws_next;
            if constexpr (tap + 1 < NTAPS)
                ws_next = (int)__builtin_amdgcn_raw_buffer_load_b32(wr, lane_s, NK * WSTEP_B + (tap + 1) * 1024 + hw * 256, 0);
        }
        if constexpr (NEXT && k == NK - 1)
            ws_next = (int)__builtin_amdgcn_raw_buffer_load_b32(wr_next, lane_s, NTAPS_NEXT * 4 * WSTEP_B + hw * 256, 0);
        k32_step<SWAP, J, JN, LW, LX, TOFF, TOFFN>(ring.w[k % RING], ring.w[(k + RING - 1) % RING], xf, xc, sx, xh, xq, (J & 1) ? to : te, xs,
                                                  NEXT ? wr_next : wr, woff + (NEXT ? k + RING - 1 - NK : k + RING - 1) * WSTEP_B, l16, ws, acc);
    });
#ifndef GNN_C6_NOCHAIN
    ring.ws = ws_next;      // scale word of the next tile's tap 0
#endif
}

// biased f16 exponent (at least 1: the subnormal range shares the exponent of the smallest normals) of a
// non-negative f32 value rounded to f16
__device__ __forceinline__ uint32_t f16_exp(float amax) {
    const _Float16 h = (_Float16)amax;
    const uint32_t e = ((uint32_t)__builtin_bit_cast(unsigned short, h) >> 10) & 31u;
    return e > 1u ? e : 1u;
}

// One (row, 32-channel block) unit, x[] = the block's activations after LeakyReLU in channel order -> the row's
// operand images: h = f16(x) (RNE), x6 = e2m3(h / 2^(E-2)) with E = exponent of the block's largest |h|, and
// xl6 = e2m3 of the f16 rounding residual x - h, block scaled by the exponent of its own largest element
// (v_cvt_scalef32_2xpk16_fp6_f32 takes the f32 residuals directly: output slot 2i = first source [i], slot 2i+1 = second
// source [i], profiles/history/r02_probe_cvt6.txt).  The scale bytes are what the MFMA multiplies the fragments with
// (2^(byte - 127)).  Instruction count matters here: the helper waves issue beside the matrix waves' MFMA stream, at a
// fraction of the nominal VALU rate (profiles/README.md), so the maxima are three-operand and nothing is converted twice.
__device__ __forceinline__ void store_block32(unsigned char* __restrict__ buf, int buf_row, int blk, const float (&x)[32]) {
    unsigned char* row = buf + buf_row * ROW6;
    const int b8 = (blk ^ swz(buf_row)) * 8;
    f16x32 hv;
    f32x16 re, ro;         // residuals of the even / odd channels
    float ax = 0.f, ar = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const f16x2 h = __builtin_convertvector(f32x2{x[2 * i], x[2 * i + 1]}, f16x2);
#ifdef GNN_C6_RESID_CVT      // measurement variant: convert back and subtract (two instructions per value instead of one)
        const float r0 = x[2 * i] - (float)h[0], r1 = x[2 * i + 1] - (float)h[1];
#else
        const uint32_t hb = __builtin_bit_cast(uint32_t, h);
        const float r0 = sub_f16_lo(x[2 * i], hb), r1 = sub_f16_hi(x[2 * i + 1], hb);
#endif
        hv[2 * i] = h[0];
        hv[2 * i + 1] = h[1];
        re[i] = r0;
        ro[i] = r1;
        ax = __builtin_fmaxf(__builtin_fmaxf(ax, __builtin_fabsf(x[2 * i])), __builtin_fabsf(x[2 * i + 1]));
        ar = __builtin_fmaxf(__builtin_fmaxf(ar, __builtin_fabsf(r0)), __builtin_fabsf(r1));
    }
    // E8M0 byte b <-> scale 2^(b - 127); e2m3 tops out at 7.5 = 1.875 * 2^2, so a block whose largest element has the
    // exponent e is divided by 2^(e - 2).  x6 is converted from h: E = biased f16 exponent of the largest |h|, b = E - 15 - 2
    // + 127; the residual from f32: b = its biased f32 exponent - 2 (at least 1)
    const uint32_t bx = f16_exp(ax) + 110u;
    const uint32_t er = (__float_as_uint(ar) >> 23) & 255u;
    const uint32_t br = (er > 3u ? er : 3u) - 2u;
    const i32x6 x6 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(hv, __uint_as_float(bx << 23));
    const i32x6 l6 = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(re, ro, __uint_as_float(br << 23));
    uint4* hp = reinterpret_cast<uint4*>(row + blk * 64);
    const uint4* hs = reinterpret_cast<const uint4*>(&hv);
#pragma unroll
    for (int i = 0; i < 4; ++i) hp[i] = hs[i];
    *reinterpret_cast<uint4*>(row + X6A + blk * 16) = make_uint4((uint32_t)x6[0], (uint32_t)x6[1], (uint32_t)x6[2], (uint32_t)x6[3]);
    *reinterpret_cast<uint2*>(row + X6B + b8) = make_uint2((uint32_t)x6[4], (uint32_t)x6[5]);
    *reinterpret_cast<uint4*>(row + L6A + blk * 16) = make_uint4((uint32_t)l6[0], (uint32_t)l6[1], (uint32_t)l6[2], (uint32_t)l6[3]);
    *reinterpret_cast<uint2*>(row + L6B + b8) = make_uint2((uint32_t)l6[4], (uint32_t)l6[5]);
    row[SXO + blk] = (unsigned char)bx;
    row[SLO + blk] = (unsigned char)br;
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

// The wave's 32 output channels are one block of the row.  A lane holds 16 of them (the other 16 sit in lane ^ 32),
// so the m-blocks are taken in pairs: after 16 v_permlane32_swap lanes 0-31 hold all 32 channels of m-block p's row
// l, lanes 32-63 all 32 channels of m-block p+1's row l & 31.
__device__ __forceinline__ void conv_epilogue(unsigned char* __restrict__ obuf, const f32x16 (&acc)[NMB], int wave, int lane) {
#pragma unroll
    for (int p = 0; p < NMB; p += 2) {
        const bool pair = p + 1 < NMB;
        float v[32];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned a = __float_as_uint(acc[p][r]);
            const unsigned b = __float_as_uint(acc[pair ? p + 1 : p][r]);
            const auto sw = __builtin_amdgcn_permlane32_swap(a, b, false, false);   // {[a lanes 0-31 | b lanes 0-31], [a lanes 32-63 | b lanes 32-63]}
            v[8 * (r >> 2) + (r & 3)] = lrelu_f(__uint_as_float(sw[0]));
            v[8 * (r >> 2) + 4 + (r & 3)] = lrelu_f(__uint_as_float(sw[1]));
        }
        if (pair || lane < 32) store_block32(obuf, CARRY + (p + (lane >> 5)) * 32 + (lane & 31), wave, v);
    }
}

template <int NTAPS_NEXT>
__device__ __forceinline__ void wv_mfma(const unsigned char* __restrict__ smem, int xoff, wrsrc_t wr, wrsrc_t wr_next, int woff,
                                        int hw, WRing& ring, f32x16 (&acc)[NMB], int lane) {
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
    gemm_tile<false, 1, CARRY, NTAPS_NEXT>(smem, xoff + CARRY * ROW6, wr, wr_next, woff, hw, ring, acc, lane);
}
// MaxPool1D(8) -> yp rows (igloo.py:209-210), as in gnn_fused_c8.hip
__device__ __forceinline__ void wv_pool_store(const f32x16 (&acc)[NMB], wrsrc_t yp_w, int t0, int wave, int lane) {
    float m[4 * NMB];
#pragma unroll
    for (int i = 0; i < 4 * NMB; ++i) {
        const int mb = i >> 2, rg = i & 3;
        const float v = max_nan(max_nan(acc[mb][rg * 4], acc[mb][rg * 4 + 1]), max_nan(acc[mb][rg * 4 + 2], acc[mb][rg * 4 + 3]));
        const unsigned bits = __float_as_uint(v);
        const auto sw = __builtin_amdgcn_permlane32_swap(bits, bits, false, false);
        m[i] = max_nan(v, __uint_as_float(lane < 32 ? sw[1] : sw[0]));
    }
    const int q0 = t0 / GNN_POOL;
    const int nq = min(4 * NMB, POOLED - q0);
    if (lane < 32) {
        const uint32_t voff = (uint32_t)(wave * 32 + lane) * 4u;      // buffer stores: no 64-bit per-lane address lives across the tiles
#pragma unroll
        for (int i = 0; i < 4 * NMB; ++i)
            if (i < nq) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m[i]), yp_w, voff, (q0 + i) * (C * 4), 0);
    }
}

// conv1 + LeakyReLU of one (row, block) unit: conv1 on a one-hot input is a row gather-sum of its kernel (model.py:11 +
// igloo.py:45-48), 3 rows with the pair tables (gnn_fused_common.h).  The tables of this kernel are stored so that a
// lane's 32 channels come as 8 loads of 16 B of which the 4 lanes of a row read 64 contiguous bytes each time.
// The tables of this kernel (pack_fused_c6_weights) keep a row as [i 0..3][block 4][half 2][4 ch]: channel 32 p + 16 q + 4 i + k
// at float i * 32 + (2 p + q) * 4 + k.  A lane PAIR (q = 0, 1) owns block p of two neighbouring rows; each lane fetches ITS
// 16 channels of BOTH rows (4 loads per table row: the 8 lanes of a row cover one whole 128-B line per load, a wave 8 lines —
// with 4 lanes per row it was 16 half lines, and the vector memory path is paid per line touched), sums the three table rows,
// and the two lanes then swap halves (16 DPP moves) so that each holds all 32 channels of ONE row for the conversion.
// (Assigning the loads by role instead — each lane fetching the lower half of the row it keeps and the upper half of its
// partner's — removes the 44 lane-parity selects of gather_store but makes every load touch 16 half lines again: measured
// 18.64 vs 18.30 ms per 4096 windows, profiles/history/r02c6_ab_gather_roles.txt.)
// (GatherUnit, gather_issue, gather_sum: gnn_fused_helpers.h)
__device__ __forceinline__ void gather_store(const GatherSum& g, unsigned char* __restrict__ xbuf, int ua, int pq) {
    const bool odd = pq & 1;
    float x[32];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 sa = g.sa[i], sb = g.sb[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float got = dpp_xor1(odd ? sa[k] : sb[k]);             // even lane keeps row A and gets the partner's row A half
            x[4 * i + k] = lrelu_f(odd ? got : sa[k]);                   // channels 0..15 of the block
            x[16 + 4 * i + k] = lrelu_f(odd ? sb[k] : got);              // channels 16..31
        }
    }
    store_block32(xbuf, CARRY + ua + (odd ? 1 : 0), pq >> 1, x);
}
__device__ __forceinline__ void gather_finish(const GatherUnit& g, unsigned char* __restrict__ xbuf, int ua, int pq) {
    GatherSum t;
    gather_sum(t, g);
    gather_store(t, xbuf, ua, pq);
}

// IGLOO pair dot products (igloo.py:192-204 with w_mult * w_summer folded) of head B for the previous step's x3 rows
// and of head A for this step's x1 rows in ONE loop: 4 lanes per entry, one 32-channel block per lane.  A row is read
// back as f16 + residual image (x = f16 + e2m3 * 2^(scale byte - 127)); the hardware unpacks the 32 fp6 values.
// (PairJob, PairW, pair_load_w, m_partials2: gnn_fused_helpers.h)
// dot product of the unit's 32 weights with row u of the job's buffer (f16 + residual image), summed over the entry's 4 lanes
struct PairComputeC6 {
static __device__ __forceinline__ void run(const PairW& w, const PairJob& jb, int e, int u, int p) {
    const int br = CARRY + u - jb.t0;
    const unsigned char* xr = jb.xbuf + br * ROW6;
    uint4 hx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) hx[i] = *reinterpret_cast<const uint4*>(xr + p * 64 + i * 16);
    const uint4 la = *reinterpret_cast<const uint4*>(xr + L6A + p * 16);
    const uint2 lb = *reinterpret_cast<const uint2*>(xr + L6B + (p ^ swz(br)) * 8);
    const uint32_t sb = xr[SLO + p];
    const i32x6 l6 = {(int)la.x, (int)la.y, (int)la.z, (int)la.w, (int)lb.x, (int)lb.y};
    const f32x32 q = __builtin_amdgcn_cvt_scalef32_pk32_f32_fp6(l6, 1.0f);
    float s = 0.f, r = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f16x8 hh = __builtin_bit_cast(f16x8, hx[i]);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = i * 8 + k;
            const float4 wq = w.w[c >> 2];
            const float wc = (c & 3) == 0 ? wq.x : ((c & 3) == 1 ? wq.y : ((c & 3) == 2 ? wq.z : wq.w));
            s = fmaf((float)hh[k], wc, s);
            r = fmaf(q[c], wc, r);
        }
    }
    s = fmaf(r, __uint_as_float(sb << 23), s);
    s += dpp_xor1(s);
    s += dpp_xor2(s);
    if (p == 0) jb.mp[e] = s;
}
};
// Barriers B1..B4 per step exactly as in gnn_fused_c8.hip:
//   matrix : w_v A(s), conv2 loop [bufX] | B1 | epilogue -> bufY (x2) | B2 | conv3 loop [bufY] | B3 |
//            epilogue -> bufY (x3) | B4 | w_v B(s) [bufY]   -> straight into step s+1
//   helpers: pair rows of step s+1, pair products B(s-1) [bufY] and A(s) [bufX], gather(s+1) table loads | B1 | x1 carry rows,
//            first half of x1(s+1) -> bufX | B2 | read x2 carry | B3 | x2 carry rows -> bufY, second half of x1(s+1) -> bufX | B4
template <bool PROF>
__global__ __launch_bounds__(512, 2) void fused_front_c6_kernel(Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM6];
    unsigned char* bufX = smem;
    unsigned char* bufY = smem + BUF6_BYTES;
    auto prow2 = [&](int parity) { return reinterpret_cast<uint16_t*>(smem + PROW_OFF + parity * PROW_BYTES); };
    float* bias_s = reinterpret_cast<float*>(smem + BIAS_OFF);   // read at the head of every conv tile: LDS, not an L2 round trip
    int* s_last = reinterpret_cast<int*>(smem + LAST_OFF);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool helper = wave >= 4;
    const int hw = wave & 3;
    const int ht = tid & 255;
    const int64_t wi = blockIdx.x;
    const uint8_t* bases = a.bases + wi * W;
    float* mp_w[2] = {a.mp + (wi * 2 + 0) * NPAIR, a.mp + (wi * 2 + 1) * NPAIR};
    const int woff = hw * WNBLK_B;                       // this wave's n-block inside every k32 step

    // carry rows of the first step = the causal zero padding (zero fragments, scale bytes irrelevant but finite)
    for (int i = tid; i < CARRY * ROW_U4; i += 512) {
        reinterpret_cast<uint4*>(bufX)[i] = make_uint4(0, 0, 0, 0);
        reinterpret_cast<uint4*>(bufY)[i] = make_uint4(0, 0, 0, 0);
    }
    if (tid >= 256) bias_s[tid - 256] = a.conv_b[(tid - 256) >> 7][tid & 127];
    // Padding skip.  From the first position p behind which every base is non-ACGT (the N padding of a contig's last window,
    // nn_classification.py:72) every token is 0; x3[t] sees bases t-15 .. t+3, so all rows t >= p + 15 of all three layers
    // carry the values they carry in an all-N window, whatever lies before p: the steps made of such rows only are not
    // computed — their yp rows and pair products are copied from the all-N window's (same arithmetic, same rows: bit
    // identical, tests/test_gpu_parity.py).  A contig's last window is half padding on average.
    if (tid == 0) *s_last = -1;
    __syncthreads();
    if (a.yp_c) {
        int last = -1;
        for (int i = tid * 12; i < tid * 12 + 12 && i < W; ++i)
            if (base_code_f(bases[i]) >= 0) last = i;
        if (last >= 0) atomicMax(s_last, last);
    }
    // pair rows of steps 0 and 1: prow2(s & 1)[i] = pair row of positions (t0 - 5 + i, t0 - 4 + i), t0 = s * FT6
    if (tid < PROW_N) {
#pragma unroll
        for (int s01 = 0; s01 < 2; ++s01) {
            uint32_t lo, hi;
            const int t = s01 * FT6 - CARRY + tid;
            prow_fetch(bases, t, lo, hi);
            prow2(s01)[tid] = prow_make(lo, hi, t);
        }
    }
    __syncthreads();
    // steps [0, nsteps) contain a row t < p + 15 and are computed (at least one: an all-N window computes its first step)
    const int nsteps = a.yp_c ? max(1, min(STEPS6, (*s_last + 1 + 15 + FT6 - 1) / FT6)) : STEPS6;
    unsigned long long cyc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tick_ = 0;
    const int gpq = ht & 7, gua = (ht >> 3) * 2;         // conv1 gather: 2 * block + channel half, first row of the lane pair (+ 64 per round)
    constexpr int GROUNDS = (FT6 + 63) / 64;

    if (!helper) {
#ifndef GNN_C6_NOPRIO
        __builtin_amdgcn_s_setprio(2);
#endif
        const wrsrc_t cw[2] = {make_wrsrc(a.conv_w[0], KS * 4 * WSTEP_B + KS * 1024), make_wrsrc(a.conv_w[1], KS * 4 * WSTEP_B + KS * 1024)};
        const wrsrc_t vw[2] = {make_wrsrc(a.wv_w[0], 4 * WSTEP_B + 1024), make_wrsrc(a.wv_w[1], 4 * WSTEP_B + 1024)};
        const wrsrc_t yp_w[2] = {make_wrsrc(reinterpret_cast<const unsigned char*>(a.yp + (wi * 2 + 0) * (size_t)POOLED * C), POOLED * C * 4),
                                 make_wrsrc(reinterpret_cast<const unsigned char*>(a.yp + (wi * 2 + 1) * (size_t)POOLED * C), POOLED * C * 4)};
        WRing ring;
        prefetch_w<1>(ring, vw[0], woff, hw, lane);
        __syncthreads();                                                         // x1 of step 0 is in bufX
        if constexpr (PROF) tick_ = __builtin_readcyclecounter();
#pragma unroll 1
        for (int step = 0; step < nsteps; ++step) {
            const int t0 = step * FT6;
            GNN_TICK(7)
            f32x16 acc[NMB];
            wv_mfma<KS>(smem, 0, vw[0], cw[0], woff, hw, ring, acc, lane);
#ifdef GNN_C6_NOCHAIN
            prefetch_w<KS>(ring, cw[0], woff, hw, lane);                                // conv2 weights, hidden by the pooling
#endif
            wv_pool_store(acc, yp_w[0], t0, hw, lane);
            GNN_TICK(0)
            acc_init_bias(acc, bias_s, hw, lane);
            gemm_tile<true, KS, 0, KS>(smem, 0, cw[0], cw[1], woff, hw, ring, acc, lane);
#ifdef GNN_C6_NOCHAIN
            prefetch_w<KS>(ring, cw[1], woff, hw, lane);                                // conv3 weights, hidden by epilogue + barriers
#endif
            GNN_TICK(1)
            __syncthreads();                                                     // ---- B1
            GNN_TICK(2)
            conv_epilogue(bufY, acc, hw, lane);
            __syncthreads();                                                     // ---- B2
            GNN_TICK(3)
            acc_init_bias(acc, bias_s + C, hw, lane);
            gemm_tile<true, KS, 0, 1>(smem, BUF6_BYTES, cw[1], vw[1], woff, hw, ring, acc, lane);
#ifdef GNN_C6_NOCHAIN
            prefetch_w<1>(ring, vw[1], woff, hw, lane);                                // w_v of head B
#endif
            GNN_TICK(4)
            __syncthreads();                                                     // ---- B3
            GNN_TICK(5)
            conv_epilogue(bufY, acc, hw, lane);
            __syncthreads();                                                     // ---- B4
            GNN_TICK(6)
            wv_mfma<1>(smem, BUF6_BYTES, vw[1], vw[0], woff, hw, ring, acc, lane);
#ifdef GNN_C6_NOCHAIN
            prefetch_w<1>(ring, vw[0], woff, hw, lane);                                // w_v of head A for the next step
#endif
            wv_pool_store(acc, yp_w[1], t0, hw, lane);
        }
    } else {
        {
            GatherUnit g;
#pragma unroll
            for (int k = 0; k < GROUNDS; ++k)
                if (gua + 64 * k < FT6) {
                    gather_issue(g, prow2(0), a.conv1_k, gua + 64 * k, gpq);
                    gather_finish(g, bufX, gua + 64 * k, gpq);
                }
        }
        uint32_t nlo = 0, nhi = 0;                       // bytes of this thread's pair row of the step AFTER next
        if (ht < PROW_N) prow_fetch(bases, 2 * FT6 - CARRY + ht, nlo, nhi);
        __syncthreads();
        if constexpr (PROF) tick_ = __builtin_readcyclecounter();
#pragma unroll 1
        for (int step = 0; step < nsteps; ++step) {
            const int t0 = step * FT6;
            const uint16_t* prow = prow2((step + 1) & 1);                        // pair rows of the next step: written before B4 of the previous one
#ifdef GNN_RACE_DELAY
            // measurement only (scripts/prow_race_demo.py): helper wave 5 reaches the step late, as a cold TLB / instruction
            // cache makes one wave do.  The round-2 kernel wrote the pair rows here, with no barrier before the other waves'
            // gathers read them; this kernel writes them between B3 and B4 of the previous step and is indifferent to the delay.
            if (wave == 5 && (step % 8) == 3)
                for (int i = 0; i < GNN_RACE_DELAY; ++i) __builtin_amdgcn_s_sleep(127);
#endif
            GNN_TICK(10)
            {
                const int sb = max(step - 1, 0);                                 // step 0: empty head-B range
                const PairJob jb = {bufY, a.weff[1], a.pos_sorted[1], mp_w[1], t0 - FT6,
                                    step > 0 ? a.bucket_ptr[1][sb] : 0, step > 0 ? a.bucket_ptr[1][sb + 1] : 0};
                const PairJob ja = {bufX, a.weff[0], a.pos_sorted[0], mp_w[0], t0, a.bucket_ptr[0][step], a.bucket_ptr[0][step + 1]};
#ifndef GNN_ABL_NOHELP
                m_partials2<PairComputeC6>(jb, ja, hw, lane);
#endif
            }
            uint4 carry = make_uint4(0, 0, 0, 0);
            const int cr = ht / ROW_U4, cc = ht - cr * ROW_U4;   // 5 rows x 29 chunks of 16 B
            if (ht < CARRY * ROW_U4) carry = *reinterpret_cast<const uint4*>(bufX + (FT6 + cr) * ROW6 + cc * 16);
            // conv1 gather of the next step.  Beside the matrix waves' MFMA stream a helper wave gets VALU issue slots at
            // about a quarter of the nominal rate (one row conversion: 2.9 k cycles alone, 10 k beside the conv3 loop), so
            // the table loads are requested here, before B1; the first conversion runs in the window in which the matrix waves
            // are in their conv2 epilogue and no MFMA is in flight on the CU (B1..B2), the second beside the conv3 loop, where
            // the helpers have nothing else to do (both in the B3..B4 window as well made the matrix waves wait at B2 AND B4).
#ifndef GNN_ABL_NOHELP
            GatherUnit g0, g1;
            gather_issue(g0, prow, a.conv1_k, gua, gpq);
            gather_issue(g1, prow, a.conv1_k, gua + 64, gpq);
#endif
            GNN_TICK(8)
            __syncthreads();                                                     // ---- B1
            GNN_TICK(11)
            if (ht < CARRY * ROW_U4) *reinterpret_cast<uint4*>(bufX + cr * ROW6 + cc * 16) = carry;
#ifndef GNN_ABL_NOHELP
            gather_finish(g0, bufX, gua, gpq);
            GatherSum s1;
            gather_sum(s1, g1);
#endif
            GNN_TICK(12)
            __syncthreads();                                                     // ---- B2
            GNN_TICK(13)
#ifndef GNN_ABL_NOHELP
#ifndef GNN_C6_X1B_LATE
            gather_store(s1, bufX, gua + 64, gpq);                              // beside the conv3 loop, where the helpers have nothing else to do
#endif
#pragma unroll
            for (int k = 2; k < GROUNDS; ++k)
                if (gua + 64 * k < FT6) {
                    GatherUnit g;
                    gather_issue(g, prow, a.conv1_k, gua + 64 * k, gpq);
                    gather_finish(g, bufX, gua + 64 * k, gpq);
                }
#endif
            if (ht < CARRY * ROW_U4) carry = *reinterpret_cast<const uint4*>(bufY + (FT6 + cr) * ROW6 + cc * 16);
            GNN_TICK(9)
            __syncthreads();                                                     // ---- B3
            GNN_TICK(14)
            if (ht < CARRY * ROW_U4) *reinterpret_cast<uint4*>(bufY + cr * ROW6 + cc * 16) = carry;
            // pair rows of step s+2 into the buffer step s's rows were in (last read before B1 of step s-1), from the bytes
            // requested a step ago; then the request for the step after
            if (ht < PROW_N) {
                const int t = t0 + 2 * FT6 - CARRY + ht;
                prow2(step & 1)[ht] = prow_make(nlo, nhi, t);
                prow_fetch(bases, t + FT6, nlo, nhi);
            }
#if defined(GNN_C6_X1B_LATE) && !defined(GNN_ABL_NOHELP)
            gather_store(s1, bufX, gua + 64, gpq);                              // measurement variant: in the conv3 epilogue window
#endif
            GNN_TICK(15)
            __syncthreads();                                                     // ---- B4
            if constexpr (PROF) tick_ = __builtin_readcyclecounter();
        }
        const PairJob jb = {bufY, a.weff[1], a.pos_sorted[1], mp_w[1], (nsteps - 1) * FT6, a.bucket_ptr[1][nsteps - 1],
                            a.bucket_ptr[1][nsteps]};
        const PairJob none = {bufX, a.weff[0], a.pos_sorted[0], mp_w[0], 0, 0, 0};
        m_partials2<PairComputeC6>(jb, none, hw, lane);
    }
    if (nsteps < STEPS6) {            // the all-N tail: copy instead of compute (disjoint from what the steps above wrote)
        const int q0 = nsteps * (FT6 / GNN_POOL);
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

// ---------------------------------------------------------------------------------- host side
static double e2m3_value(uint32_t c) {
    const int e = (c >> 3) & 3, m = c & 7;
    return e == 0 ? m / 8.0 : (1 + m / 8.0) * std::ldexp(1.0, e - 1);
}
// e2m3 encode, round to nearest (ties to the even code), saturating at 7.5
static uint32_t e2m3_encode(double x) {
    const uint32_t s = std::signbit(x) ? 32u : 0u;
    const double a = std::fabs(x);
    uint32_t best = 0;
    double bd = 1e300;
    for (uint32_t c = 0; c < 32; ++c) {
        const double d = std::fabs(e2m3_value(c) - a);
        if (d < bd || (d == bd && !(c & 1))) bd = d, best = c;
    }
    return s | best;
}
// OCP MX block exponent for e2m3 elements: floor(log2(amax)) - 2 (oracle/precision_study.py rnd_mx)
static int mx6_exponent(double amax) {
    if (!(amax > 0)) return -127 + 1;
    int e;
    std::frexp(amax, &e);           // amax = f * 2^e, f in [0.5, 1)  ->  floor(log2 amax) = e - 1
    return std::min(std::max(e - 1 - 2, -126), 127);
}
static void put6(uint32_t* w, int i, uint32_t c) {
    const int bit = i * 6, k = bit >> 5, o = bit & 31;
    w[k] |= c << o;
    if (o > 26) w[k + 1] |= c >> (32 - o);
}

// Wmat (K x N row major; K, N multiples of 32) -> the Args weight stream and scale words.
static void pack_c6(const float* wmat, int K, int N, std::vector<uint32_t>& frags, std::vector<uint32_t>& scales) {
    const int nk32 = K / 32, nblks = N / 32, ntaps = (nk32 + 3) / 4;
    frags.assign((size_t)nk32 * nblks * (WNBLK_B / 4), 0);
    scales.assign((size_t)ntaps * nblks * 64, 0x7F7F7F7Fu);
    for (int ks = 0; ks < nk32; ++ks)
        for (int nb = 0; nb < nblks; ++nb) {
            unsigned char* bytes = reinterpret_cast<unsigned char*>(&frags[((size_t)ks * nblks + nb) * (WNBLK_B / 4)]);
            for (int l = 0; l < 64; ++l) {
                const int n = nb * 32 + (l & 31), half = l >> 5;
                for (int s = 0; s < 2; ++s)             // f16 fragments: k = ks*32 + s*16 + half*8 + e
                    for (int e = 0; e < 8; ++e) {
                        const _Float16 h = (_Float16)wmat[(size_t)(ks * 32 + s * 16 + half * 8 + e) * N + n];
                        std::memcpy(bytes + (size_t)s * 1024 + l * 16 + e * 2, &h, 2);
                    }
                // fp6 fragment of this lane: K block `half` = (w - f16 w) for lanes 0-31, w for lanes 32-63, over
                // the 32 k of the step in natural order
                double v[32], amax = 0;
                for (int i = 0; i < 32; ++i) {
                    const float wv = wmat[(size_t)(ks * 32 + i) * N + n];
                    v[i] = half == 0 ? (double)wv - (double)(float)(_Float16)wv : (double)wv;
                    amax = std::max(amax, std::fabs(v[i]));
                }
                const int e = mx6_exponent(amax);
                uint32_t f[6] = {0, 0, 0, 0, 0, 0};
                for (int i = 0; i < 32; ++i) put6(f, i, e2m3_encode(v[i] / std::ldexp(1.0, e)));
                std::memcpy(bytes + 2048 + l * 16, f, 16);
                std::memcpy(bytes + 3072 + l * 8, f + 4, 8);
                unsigned char* sc = reinterpret_cast<unsigned char*>(&scales[((size_t)(ks >> 2) * nblks + nb) * 64 + l]);
                sc[ks & 3] = (unsigned char)(e + 127);
            }
        }
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

}  // namespace c6

namespace c6 {
static void fill_args(const gnn_ctx* ctx, Args& a, const uint8_t* bases) {
    const DeviceWeights& d = ctx->w;
    a.bases = bases;
    a.conv1_k = d.conv1_pairs6;
    for (int i = 0; i < 2; ++i) {
        a.conv_w[i] = reinterpret_cast<const unsigned char*>(d.conv_c6[i]);
        a.conv_b[i] = d.conv_b[i];
        a.wv_w[i] = reinterpret_cast<const unsigned char*>(d.wv_c6[i]);
        a.weff[i] = d.weff6[i];
        a.pos_sorted[i] = d.pos_sorted[i];
        a.bucket_ptr[i] = d.bucket_ptr6[i];
    }
}
}  // namespace c6

int pack_fused_c6_weights(gnn_ctx* ctx, const gnn_weights* w) {
    using namespace c6;
    DeviceWeights& d = ctx->w;
    const float* ck[2] = {w->conv2_kernel, w->conv3_kernel};
    const gnn_igloo_weights* ig[2] = {&w->igloo_a, &w->igloo_b};
    std::vector<uint32_t> f, s;
    int rc;
    for (int i = 0; i < 2; ++i) {
        pack_c6(ck[i], KS * C, C, f, s);          // one allocation: fragments, then the scale words (one buffer resource in the kernel)
        f.insert(f.end(), s.begin(), s.end());
        if ((rc = upload_vec(ctx, f, &d.conv_c6[i]))) return rc;
        pack_c6(ig[i]->w_v, C, C, f, s);
        f.insert(f.end(), s.begin(), s.end());
        if ((rc = upload_vec(ctx, f, &d.wv_c6[i]))) return rc;
        // entry ranges of the position-sorted IGLOO pairs per step of FT6 rows
        std::vector<int32_t> ptr(STEPS6 + 1, 0);
        for (int e = 0; e < NPAIR; ++e) ptr[ig[i]->patches[e] / FT6 + 1] += 1;
        for (int st = 0; st < STEPS6; ++st) ptr[st + 1] += ptr[st];
        if ((rc = upload_vec(ctx, ptr, &d.bucket_ptr6[i]))) return rc;
        // folded IGLOO weights in entry (position-sorted) order, re-laid for 4 lanes per entry: see PairW
        std::vector<int32_t> order(NPAIR);
        for (int e = 0; e < NPAIR; ++e) order[e] = e;
        std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return ig[i]->patches[x] < ig[i]->patches[y]; });
        std::vector<float> w6((size_t)(NPAIR + 1) / 2 * 2 * C, 0.f);
        for (int e = 0; e < NPAIR; ++e) {
            const int pair = order[e], j = pair % PS;
            float* dst = &w6[(size_t)(e >> 1) * 2 * C + (e & 1) * 16];
            for (int c = 0; c < C; ++c)     // channel 32 p + 4 i + k  ->  i * 32 + p * 4 + k  (+ 16 for the odd entry)
                dst[((c >> 2) & 7) * 32 + (c >> 5) * 4 + (c & 3)] = ig[i]->w_mult[(size_t)pair * C + c] * ig[i]->w_summer[j * C + c];
        }
        if ((rc = upload_vec(ctx, w6, &d.weff6[i]))) return rc;
    }
    // conv1 pair tables in the lane order of this kernel's gather, conv1 bias folded into table 0 (every position adds
    // exactly one row of each table)
    std::vector<float> pt;
    build_conv1_pair_tables(w->conv1_kernel, pt);
    std::vector<float> pq(pt.size());
    for (int j = 0; j < 3; ++j)
        for (int r = 0; r < PAIR_ROWS; ++r) {
            const float* src = &pt[((size_t)j * PAIR_ROWS + r) * C];
            float* dst = &pq[((size_t)j * PAIR_ROWS + r) * C];
            for (int c = 0; c < C; ++c) {
                const int pq = c >> 4, i = (c >> 2) & 3, e = c & 3;      // channel 32 p + 16 q + 4 i + e
                dst[i * 32 + pq * 4 + e] = src[c] + (j == 0 ? w->conv1_bias[c] : 0.f);
            }
        }
    if ((rc = upload_vec(ctx, pq, &d.conv1_pairs6))) return rc;
    // the all-N window's outputs, computed once by the kernel itself (padding skip: see the kernel's prologue)
    {
        void *bn = nullptr, *yc = nullptr, *mc = nullptr;
        GNN_HIP(hipMalloc(&bn, W));
        GNN_HIP(hipMalloc(&yc, (size_t)2 * POOLED * C * sizeof(float)));
        GNN_HIP(hipMalloc(&mc, (size_t)2 * NPAIR * sizeof(float)));
        ctx->owned.push_back(bn);
        ctx->owned.push_back(yc);
        ctx->owned.push_back(mc);
        GNN_HIP(hipMemsetAsync(bn, 'N', W, ctx->stream));
        Args a;
        fill_args(ctx, a, static_cast<const uint8_t*>(bn));
        a.mp = static_cast<float*>(mc);
        a.yp = static_cast<float*>(yc);
        a.yp_c = nullptr;
        a.mp_c = nullptr;
        a.cycles = nullptr;
        hipLaunchKernelGGL((fused_front_c6_kernel<false>), dim3(1), dim3(512), 0, ctx->stream, a);
        GNN_HIP(hipGetLastError());
        GNN_HIP(hipStreamSynchronize(ctx->stream));
        d.c6_yp_const = static_cast<float*>(yc);
        d.c6_mp_const = static_cast<float*>(mc);
    }
    return GNN_OK;
}

int c6_rows_per_step() { return c6::FT6; }

// host-only: the f16c6 weight stream of one K x N matrix (fragments, then scale words) for the CPU test-suite
int c6_pack_matrix(const float* wmat, int K, int N, std::vector<uint32_t>& out) {
    if (!wmat || K <= 0 || N <= 0 || K % 128 || N % 32) return GNN_ERR_ARG;
    std::vector<uint32_t> sc;
    c6::pack_c6(wmat, K, N, out, sc);
    out.insert(out.end(), sc.begin(), sc.end());
    return GNN_OK;
}

int launch_front_c6(gnn_ctx* ctx, const uint8_t* bases, int64_t n) {
    using namespace c6;
    if (reinterpret_cast<uintptr_t>(bases) & 3u) {
        set_error("f16c6: the window buffer must be 4-byte aligned");
        return GNN_ERR_ARG;
    }
    const bool pad_skip = ctx->c6_pad_skip;
    Args a;
    fill_args(ctx, a, bases);
    a.mp = ctx->ws.mp;
    a.yp = ctx->ws.yp;
    a.yp_c = pad_skip ? ctx->w.c6_yp_const : nullptr;
    a.mp_c = pad_skip ? ctx->w.c6_mp_const : nullptr;
    a.cycles = ctx->phase_cycles;
    if (ctx->phase_cycles) hipLaunchKernelGGL((fused_front_c6_kernel<true>), dim3((unsigned)n), dim3(512), 0, ctx->stream, a);
    else hipLaunchKernelGGL((fused_front_c6_kernel<false>), dim3((unsigned)n), dim3(512), 0, ctx->stream, a);
    GNN_HIP(hipGetLastError());
    return GNN_OK;
}

}  // namespace gnn
