// Fused front end, GNN_PREC_F16X3TK (round 6): the f16x3tc arithmetic with everything that is a function of a short k-mer read
// from tables in HBM instead of being computed - the design point that 288 GB per GPU open.
//
//   x1[t] = LeakyReLU(conv1)(t)  depends on the bases t-5 .. t+3    (model.py:11, igloo.py:45-48)           a  9-mer
//   x2[t] = LeakyReLU(conv2(x1))(t) depends on x1[t-5 .. t], i.e. on the bases t-10 .. t+3 (igloo.py:65-67)  a 14-mer
//
//   X2Table   4^14 + 1 rows x 128 f32 = 137.4 GB   x2[t] per 14-mer (row 4^14: every token of the 14 bases is the N token)
//   MpaTable  8400 x (4^9 + 1) f32    =   8.8 GB   head A's pair product of entry e (igloo.py:192-204 folded) per 9-mer at pos[e]
//   WvaTable  (gnn_tc_dev.h)          =   1.4 GB   head A's y @ w_v row per 9-mer
//   PT        6 x 2.7 M rows x 128 f32 =  8.3 GB   conv2's tap j applied to x1 of every row of WvaTable's index space (rows no 14-mer indexes)
//   X1T       2.7 M rows x 128 f32   =   1.4 GB   x1 over the same index space (head A's entries MpaTable has no column for)
//
// conv2 - 43 % of a window's FLOPs, half of the default kernel's MFMAs, weight stream, input transforms and barriers - becomes ONE
// 512-byte row gather per position (3.1 MB per window; scripts/probe_gather_big.hip: the memory system delivers such rows from a
// 137 GB table at 11.6 G rows/s = 5.9 TB/s, 1.9 M windows/s worth), head A's pair products one 4-byte read per entry.  What is left
// on the matrix pipe is conv3 (Toom-Cook F(3,6), three f16 products, as gnn_fused_tc.hip) and head B's y @ w_v; x1 is never
// materialised.  All tables are built on the device (f64 accumulation, rounded to f32 once: closer to the reference's f32 than the
// three f16 products they replace) by gnn_build_kmer_tables() when the device has the memory; otherwise the library keeps serving
// GNN_PREC_F16X3TC.
//
// What no dense k-mer index covers - the first ten positions of a window (tokens before the window start are absent, not N), k-mers
// that mix ACGT with other bytes - is summed from conv2's six TAP TABLES over WvaTable's index space (PT[j][row of x1[t-5+j]], 8.3 GB:
// <= 6 row reads per such position, dirty_rows_fill), head A's entries at such 9-mers are a dot product of a row of X1T (x1 over the
// same index space, 1.4 GB) with the entry's weights (pair_a_slow).  A k-mer in which every 4-mer holds a non-ACGT byte (all its
// tokens are the N token: N runs, the padding of a contig's last window) has its own row.  Every index comes from the window's 2-bit
// codes in LDS, built once per window.
//
// Structure (one workgroup = one window, 4 matrix waves + 4 helper waves, steps of 96 rows, the ring of 3 x 16 KB of transformed
// activations - all as gnn_fused_tc.hip): two row buffers P, Q of 101 rows x 528 B alternate; step s finds x2(s) (rows t0-5 ..
// t0+95: the five carry rows are gathered again instead of being carried) in buf[s & 1], the conv3 epilogue overwrites it with x3(s)
// (hi | lo planes), and x2(s+1) lands in the other buffer, whose x3(s-1) is dead by then:
//
//   matrix : [c0 u0 | ... | c7 u7] conv3 -> inverse transform -> x3 -> buf[s&1] | E | head A's 24 table rows per wave of step s+1
//            requested, w_v B(s), pooled -> yp B, V3(s+1) chunk 1, 8-row max of the table rows -> yp A
//   helpers: top of the step: the 26 X2Table rows per wave of step s+1 requested in one burst (registers), behind them head A's
//            MpaTable read | beside units 0..5: V3 chunks 2..7 | c6: rows -> buf[(s+1)&1], rows no 14-mer indexes from the tap tables |
//            c7: V3(s+1) chunk 0, head B's three passes of weights and head A's next entry requested | E | next entry located, head B's
//            pair products, head A's product stored
//   (tests/test_kernel_schedule.py::tk_schedule is an executable model of this schedule)
//
// 9 workgroup barriers per step instead of 18.  Two rules shaped the helpers' side (DESIGN.md section 4.0 has the A/Bs): a wave's
// loads return IN ORDER, and the compiler waits with vmcnt(0) for any load it issued under a condition - so the burst of row
// requests sits where nothing is pending in front of it and nothing requested behind it is needed before the rows are.
// LDS: 2 x 53.3 KB + 48 KB ring + 3 KB of 2-bit codes + small lists = 160.5 KB.
//
// Compile-time switches: -DTC_JITTER only (libgenomad_nn_hip_jitter.so, a test build).  The measurement variants behind
// profiles/r06/tk_first/ (TK_ABL_NOX2LOAD, TK_ABL_SMALLTBL, TK_ABL_NOMPA, TK_X2_TRICKLE, TK_HPRIO_*, ...) are those of commit 75dfcae;
// they were removed from this file afterwards (the device code did not change by a byte).
#include <chrono>
#include <vector>
#include <cstdio>

#include "gnn_tc_dev.h"

#define __noinline__ __noinline__

namespace gnn {
namespace tk {

using namespace tc;

constexpr int YP_NT = 2;                            // cache policy of the pooled rows' stores: nt (bit 1) - this kernel never reads them again (+0.5 %)
constexpr uint32_t X2_ROWS = (1u << 28) + 1u;       // 4^14 fourteen-mers + the all-N-token row
constexpr uint32_t X2_NN = 1u << 28;
constexpr uint32_t K9_ROWS = (1u << 18) + 1u;       // 4^9 nine-mers + the all-N-token 9-mer
constexpr uint32_t K9_NN = 1u << 18;
constexpr int QUADS = W / 4;                        // 1500 aligned groups of four bases
constexpr int QUAD_N = QUADS + 4;                   // + zero tail: a 14-mer at the worst alignment reads one group past the end
constexpr int QUAD_OFF = VRING_OFF + VRING * VSLOT;
constexpr int BIASK_OFF = QUAD_OFF + ((QUAD_N * 2 + 15) / 16) * 16;
constexpr int DIRTY_OFF = BIASK_OFF + C * 4;        // [0] count, [1..] buffer rows no table holds
constexpr int DIRTY_MAX = BUF_ROWS;
constexpr int PIDX_N = 32;                          // per helper wave: WvaTable rows of the positions its 26 buffer rows' taps touch (5 + 26)
constexpr int PIDX_OFF = DIRTY_OFF + ((4 + DIRTY_MAX + 15) / 16) * 16;
constexpr int BKT_OFF = PIDX_OFF + 4 * PIDX_N * 4;  // entry ranges per step of both heads: 2 x (STEPST + 1) ints
constexpr int LASTK_OFF = BKT_OFF + 2 * (STEPST + 1) * 4;
constexpr int SMEMK = LASTK_OFF + 16;
constexpr int X2_PER_WAVE = (BUF_ROWS + 3) / 4;     // 26 rows of the next step per helper wave (the last wave: 23)
static_assert(SMEMK <= 160 * 1024, "LDS budget");

struct ArgsK {
    const uint8_t* bases;
    const unsigned char* tcw3;        // transformed conv3 weights (gnn_fused_tc.hip, pack_fused_tc_weights)
    float inv_s3;
    const float* conv_b3;
    const unsigned char* wv_w;        // head B's w_v fragments
    const float* weff_b;              // head B's folded IGLOO weights (weff6 layout)
    const int32_t* pos_sorted[2];
    const int32_t* bucket_ptr[2];     // (STEPST + 1,) entry ranges per 96-row step
    const unsigned char* wva_tbl;     // WvaTable
    const float* x2_tbl;              // X2Table
    const float* mpa_tbl;             // MpaTable
    const float* x1t_tbl;             // x1 over WvaTable's index space (head A's entries MpaTable has no column for)
    const float* pt_tbl;              // conv2's tap tables over WvaTable's index space (rows the 14-mer table cannot index)
    const float* conv2_b;
    const float* weff_a;              // (8400, 128) f32, entry order (slow path)
    float* mp;
    float* yp;
    const float* yp_c;                // outputs of an all-N window (padding skip), nullptr = compute everything
    const float* mp_c;
    unsigned long long* cycles;
    int split;
};

// ---------------------------------------------------------------- the window as 2-bit codes
// One u16 per aligned group of four bases: bits 7..0 the four digits (A, C, G, T -> 0..3 in the order of sequence.py:170-193, first
// base on top, 0 for any other byte), bits 11..8 one flag per base (same order) set for a non-ACGT byte.  Built once per window.
__device__ __forceinline__ uint32_t quad_of(uint32_t w4) {
    uint32_t code = 0, dirty = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t d = wva_digit((w4 >> (8 * i)) & 255u);
        code = (code << 2) | (d & 3u);
        dirty = (dirty << 1) | (d >> 2);
    }
    return (dirty << 8) | code;
}
// the LEN-mer that starts at base b (0 <= b <= W - LEN): 2 LEN bits, first base on top; `dirty` one bit per base (same order)
template <int LEN>
__device__ __forceinline__ void kmer_q(const uint16_t* __restrict__ quads, int b, uint32_t& code, uint32_t& dirty) {
    constexpr int NQ = (LEN + 6) / 4;                  // groups a LEN-mer spans at the worst alignment
    const int q0 = b >> 2, r = b & 3;
    unsigned long long c = 0;
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const uint32_t v = quads[q0 + i];
        c = (c << 8) | (v & 255u);
        d = (d << 4) | (v >> 8);
    }
    const int drop = 4 * NQ - r - LEN;
    code = (uint32_t)(c >> (2 * drop)) & ((1u << (2 * LEN)) - 1u);
    dirty = (d >> drop) & ((1u << LEN) - 1u);
}
// every token (4-mer) inside the LEN bases holds a flagged base: the activations are those of an all-N stretch
template <int LEN>
__device__ __forceinline__ bool all_n_tokens(uint32_t dirty) {
    const uint32_t clean = ~dirty & ((1u << LEN) - 1u);
    return ((clean & (clean >> 1) & (clean >> 2) & (clean >> 3)) & ((1u << (LEN - 3)) - 1u)) == 0u;
}
constexpr uint32_t ROW_DIRTY = 0x80000000u;            // flag on a row index: no table row, dirty_rows_pass computes it (the index then is X2_NN: a valid row)
// X2Table row of position t (any int: the rows of a step run from t0 - 5 to t0 + 95)
__device__ __forceinline__ uint32_t x2_index(const uint16_t* __restrict__ quads, int t) {
    if (t < 10) return ROW_DIRTY | X2_NN;              // t < 0: a zero row (causal padding of conv3); 0 .. 9: tokens before the window start are absent
    uint32_t code, dirty;
    kmer_q<14>(quads, min(t, T - 1) - 10, code, dirty);     // rows past the last token are never used: any finite row
    if (dirty == 0u) return code;
    return all_n_tokens<14>(dirty) ? X2_NN : (ROW_DIRTY | X2_NN);
}
// WvaTable row of position t (wva_index of gnn_tc_dev.h: D4 | S5 | D5) from the 2-bit codes instead of the bytes: no memory round trip.
// Digit of a base = its code, 4 when flagged; t < 5: the bases 0 .. t+3 in base 5; t >= 5: the 9-mer, dense when no base is flagged
__device__ __forceinline__ uint32_t wva_index_q(const uint16_t* __restrict__ quads, int t) {
    uint32_t code, dirty;
    kmer_q<9>(quads, max(t - 5, 0), code, dirty);
    if (t >= 5 && dirty == 0u) return code;
    const int np = min(t, 5) + 4;
    uint32_t i5 = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const uint32_t d = ((dirty >> (8 - k)) & 1u) ? 4u : (code >> (2 * (8 - k))) & 3u;
        i5 = k < np ? i5 * 5u + d : i5;
    }
    return t < 5 ? WvaTable::s5_off(t) + i5 : WvaTable::D5_OFF + i5;
}
// WvaTable row of the matrix wave's lane (lane l < 24 of wave hw: row 24 hw + l of the step), from the 2-bit codes; a lane whose
// 9-mer is not all ACGT (or t < 5) indexes the side tables S5 / D5
__device__ __forceinline__ uint32_t wva_step_index_q(const uint16_t* __restrict__ quads, int t0, int hw, int lane) {
    return wva_index_q(quads, wva_row(t0, hw, lane));
}

// ---------------------------------------------------------------- what the k-mer tables cannot index
// head A's pair product of entry e at position u (igloo.py:192-204 folded) when MpaTable has no column for the 9-mer (a non-ACGT byte
// among the nine, or u < 5): the x1 row of that position - a row of X1T, x1 over WvaTable's index space - times the entry's folded
// weights, one lane, 128 channels, four accumulators
__device__ __noinline__ float pair_a_slow(const uint16_t* __restrict__ quads, const float* __restrict__ x1t, const float* __restrict__ weff_a, int e, int u) {
    const float4* x = reinterpret_cast<const float4*>(x1t + (size_t)wva_index_q(quads, u) * C);
    const float4* w = reinterpret_cast<const float4*>(weff_a + (size_t)e * C);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 8
    for (int i = 0; i < C / 4; ++i) {
        const float4 xv = x[i], wv = w[i];
        s0 = fmaf(xv.x, wv.x, s0);
        s1 = fmaf(xv.y, wv.y, s1);
        s2 = fmaf(xv.z, wv.z, s2);
        s3 = fmaf(xv.w, wv.w, s3);
    }
    return (s0 + s2) + (s1 + s3);
}
// The rows of buffer `buf` (row r = position tb + r) that the list in LDS names: x2[t] = LeakyReLU(b2 + sum_j W2[j]^T x1[t-5+j])
// (igloo.py:65-67) as the sum of <= 6 rows of the tap tables PT[j][row of x1[t-5+j] in WvaTable's index space] (f32; the taps whose
// x1 row lies before the window start add nothing), a zero row for t < 0.  NT threads (the helper waves: 256, the prologue: all 512),
// 128 per row, no barrier inside; the caller puts a barrier behind it and resets the list.
__device__ __forceinline__ uint32_t dirty_count(const unsigned char* __restrict__ smem) {
    return *reinterpret_cast<const volatile uint32_t*>(smem + DIRTY_OFF);
}
template <int NT, bool PIDX>
__device__ __noinline__ void dirty_rows_fill(unsigned char* __restrict__ smem, unsigned char* __restrict__ buf, int tb, const uint16_t* __restrict__ quads,
                                             const float* __restrict__ pt, const float* __restrict__ conv2_b, int tid) {
    const int n = (int)dirty_count(smem);
    const unsigned char* drows = smem + DIRTY_OFF + 4;
    const int c = tid & (C - 1);
    const float b2 = conv2_b[c];
    // four listed rows per thread and pass: 24 table rows requested before the first is used (a window with scattered non-ACGT bytes
    // lists dozens of rows per step; one round trip per row made such windows 7x slower than the default kernel, this way 2x)
    constexpr int G = NT / 128, U = 4;
    for (int i0 = tid >> 7; i0 < n; i0 += G * U) {
        float v[U][KS];
        int rr[U], tt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = min(i0 + u * G, n - 1);
            rr[u] = drows[i];
            tt[u] = min(tb + rr[u], T - 1);
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                const int p = tt[u] - CARRY + j;
                // PIDX (the step loop): buffer row r of helper wave r / 26 -> its position list, entry r - 26 (r / 26) + j
                const uint32_t row = PIDX ? reinterpret_cast<const uint32_t*>(smem + PIDX_OFF)[(rr[u] / X2_PER_WAVE) * PIDX_N + rr[u] % X2_PER_WAVE + j]
                                          : wva_index_q(quads, max(p, 0));
                v[u][j] = pt[((size_t)j * WvaTable::ROWS + row) * C + c];
                v[u][j] = p >= 0 ? v[u][j] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + u * G < n) {
                float s = b2;
#pragma unroll
                for (int j = 0; j < KS; ++j) s += v[u][j];
                s = vmax_raw(s, s * LRELU);
                *reinterpret_cast<float*>(buf + rr[u] * ROWX + c * 4) = tt[u] >= 0 ? s : 0.f;
            }
        }
    }
}

// The same for the prologue (all 512 threads, barriers allowed, the ring free as scratch): every window starts with ten such rows, and
// the walk above - per row and tap a dependent chain bytes -> index -> table row - cost 31 k cycles per window.  Two phases: the
// table row of every (listed row, tap) by one thread each into LDS, then every thread requests all table rows of its <= 4 listed
// rows at once: two round trips instead of one per row and tap.  More than 16 listed rows (an N-rich first step): the walk above.
constexpr int DIRTY_FAST_MAX = 16;
__device__ __forceinline__ void dirty_rows_fill_prologue(unsigned char* __restrict__ smem, unsigned char* __restrict__ buf, int tb,
                                                         const uint16_t* __restrict__ quads, const float* __restrict__ pt,
                                                         const float* __restrict__ conv2_b, int tid) {
    const int n = (int)dirty_count(smem);
    if (n > DIRTY_FAST_MAX) {
        dirty_rows_fill<512, false>(smem, buf, tb, quads, pt, conv2_b, tid);
        return;
    }
    const unsigned char* drows = smem + DIRTY_OFF + 4;
    uint32_t* idxs = reinterpret_cast<uint32_t*>(smem + VRING_OFF);          // [listed row][tap]: table row, ~0u = the tap adds nothing
    if (tid < n * KS) {
        const int i = tid / KS, j = tid - i * KS;
        const int t = min(tb + drows[i], T - 1), p = t - CARRY + j;
        idxs[tid] = (t >= 0 && p >= 0) ? wva_index_q(quads, p) : ~0u;
    }
    __syncthreads();
    const int c = tid & (C - 1), q = tid >> 7;
    float v[DIRTY_FAST_MAX / 4][KS];
#pragma unroll
    for (int k = 0; k < DIRTY_FAST_MAX / 4; ++k)
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const int i = q + 4 * k;
            const uint32_t idx = i < n ? idxs[i * KS + j] : ~0u;
            v[k][j] = idx != ~0u ? pt[((size_t)j * WvaTable::ROWS + idx) * C + c] : 0.f;
        }
    const float b2 = conv2_b[c];
#pragma unroll
    for (int k = 0; k < DIRTY_FAST_MAX / 4; ++k) {
        const int i = q + 4 * k;
        if (i < n) {
            const int r = drows[i];
            float s = b2;
#pragma unroll
            for (int j = 0; j < KS; ++j) s += v[k][j];           // the order of dirty_rows_fill: b2 + tap 0 + tap 1 + ... (absent taps add 0)
            s = vmax_raw(s, s * LRELU);
            *reinterpret_cast<float*>(buf + r * ROWX + c * 4) = (tb + r) >= 0 ? s : 0.f;
        }
    }
}

// ---------------------------------------------------------------- x2 rows of a step: indices, requests, stores
struct X2Rows {
    u32x4 v[X2_PER_WAVE / 2];      // instruction i: lanes 0..31 hold row 2 i, lanes 32..63 row 2 i + 1 (16 B = 4 channels per lane)
};
// lane l < X2_PER_WAVE of wave part w (0..3): buffer row 26 w + l (position tb + row); rows behind the buffer's 101 are clamped
// (requested twice, stored once).  A row no table holds goes onto the list in LDS (`list`) and keeps its flag: it is requested as the
// all-N row (any valid row) and not stored - dirty_rows_fill writes it.
__device__ __forceinline__ uint32_t x2_rows_index(unsigned char* __restrict__ smem, const uint16_t* __restrict__ quads, int tb, int w, int lane, bool list) {
    const int r = min(X2_PER_WAVE * w + min(lane, X2_PER_WAVE - 1), BUF_ROWS - 1);
    const uint32_t idx = x2_index(quads, tb + r);
    const bool mine = list && (idx & ROW_DIRTY) && lane < X2_PER_WAVE && X2_PER_WAVE * w + lane < BUF_ROWS;
    if (mine) {
        uint32_t* dl = reinterpret_cast<uint32_t*>(smem + DIRTY_OFF);
        const uint32_t slot = atomicAdd(dl, 1u);
        smem[DIRTY_OFF + 4 + slot] = (unsigned char)r;
    }
    // a wave that lists a row leaves the WvaTable rows of the 31 positions its rows' taps touch in LDS (lane l: position tb + 26 w - 5 + l):
    // the fill behind c_6 then costs one LDS read per (row, tap) instead of 60 integer instructions
    if (__builtin_amdgcn_ballot_w64(mine) && lane < PIDX_N - 1)
        reinterpret_cast<uint32_t*>(smem + PIDX_OFF)[w * PIDX_N + lane] = wva_index_q(quads, min(max(tb + X2_PER_WAVE * w - CARRY + lane, 0), T - 1));
    return idx;
}
template <int I0, int I1>
__device__ __forceinline__ void x2_rows_issue(X2Rows& x, const float* __restrict__ tbl, uint32_t my_row, int lane) {
    static_assert(I0 % 2 == 0 && I1 % 2 == 0, "two rows per request");
#pragma unroll
    for (int i = I0 / 2; i < I1 / 2; ++i) {
        // two rows per request (16 B per lane): the vector memory pipe is ISSUE-bound beside the matrix waves' weight and table-row
        // requests, and a 16-byte request costs what an 8-byte one does (MI355X_MICROARCH.md, the store tail of T21)
        const uint32_t r0 = __builtin_amdgcn_readlane(my_row, 2 * i), r1 = __builtin_amdgcn_readlane(my_row, 2 * i + 1);
        const uint32_t row = (lane < 32 ? r0 : r1) & ~ROW_DIRTY;
        const unsigned char* p = reinterpret_cast<const unsigned char*>(tbl) + (size_t)row * (C * 4) + (lane & 31) * 16;
        // non-temporal: a row is read once per window and never again - without the hint the 52 KB a step gathers
        // push weights and head A's table rows out of the L2 / Infinity Cache (+1.8 %, profiles/r06/tk_first/ab_x2_rows_nontemporal.txt)
        x.v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    }
}
__device__ __forceinline__ void x2_rows_store(const X2Rows& x, unsigned char* __restrict__ buf, uint32_t my_row, int w, int lane) {
#pragma unroll
    for (int i = 0; i < X2_PER_WAVE / 2; ++i) {
        const int r = X2_PER_WAVE * w + 2 * i + (lane >> 5);
        const uint32_t r0 = __builtin_amdgcn_readlane(my_row, 2 * i), r1 = __builtin_amdgcn_readlane(my_row, 2 * i + 1);
        const bool dirty = ((lane < 32 ? r0 : r1) & ROW_DIRTY) != 0u;
        if (r < BUF_ROWS && !dirty) *reinterpret_cast<u32x4*>(buf + r * ROWX + (lane & 31) * 16) = x.v[i];
    }
}

template <bool PROF>
__global__ __launch_bounds__(512, 2) void fused_front_tk_kernel(ArgsK a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEMK];
    const unsigned long long t_entry = PROF ? __builtin_readcyclecounter() : 0ull;
    uint16_t* quads = reinterpret_cast<uint16_t*>(smem + QUAD_OFF);
    float* bias_s = reinterpret_cast<float*>(smem + BIASK_OFF);
    int* s_last = reinterpret_cast<int*>(smem + LASTK_OFF);
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

    for (int q = tid; q < QUAD_N; q += 512)
        quads[q] = q < QUADS ? (uint16_t)quad_of(reinterpret_cast<const uint32_t*>(bases)[q]) : (uint16_t)0;
    if (tid < C) bias_s[tid] = a.conv_b3[tid];
    int* bkt = reinterpret_cast<int*>(smem + BKT_OFF);             // bkt[h * (STEPST + 1) + s]
    if (tid >= 128 && tid < 128 + 2 * (STEPST + 1)) bkt[tid - 128] = a.bucket_ptr[(tid - 128) / (STEPST + 1)][(tid - 128) % (STEPST + 1)];
    if (tid == 0) {
        *s_last = -1;
        *reinterpret_cast<uint32_t*>(smem + DIRTY_OFF) = 0u;
    }
    __syncthreads();
    if (a.yp_c) {
        int last = -1;
        for (int q = tid; q < QUADS; q += 512) {
            const uint32_t clean = ~((uint32_t)quads[q] >> 8) & 15u;         // bit 3 = first base of the group
            if (clean) last = max(last, 4 * q + 3 - (int)__builtin_ctz(clean));
        }
        if (last >= 0) atomicMax(s_last, last);
    }
    __syncthreads();
    const int nsteps = a.yp_c ? max(1, min(STEPST, (*s_last + 1 + 15 + FTT - 1) / FTT)) : STEPST;
    const int per = (nsteps + a.split - 1) / a.split;
    const int s_lo = min(part * per, nsteps), s_hi = min(s_lo + per, nsteps);       // no warm-up step: a step depends on nothing before it
    unsigned jitter_state = 0x9E3779B9u * (unsigned)(wave + 1) + (unsigned)blockIdx.x * 7919u;     // TC_JITTER builds only
    (void)jitter_state;
    unsigned long long cyc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tick_ = 0;
    if constexpr (PROF) tick_ = t_entry;

    if (s_hi > s_lo) {
        // x2 of the first step: every wave gathers 14 of its 101 rows - all requested at once, two rows per 16-byte request like the
        // step loop's burst, ONE round trip that nothing hides (once per window / part); behind them, on the matrix waves, head A's table
        // rows of the first step, so that their round trip runs beside this one
        WvaRows w0;
        {
            const int tb = s_lo * FTT - CARRY;
            unsigned char* buf = smem + (s_lo & 1) * BUF_BYTES;
            constexpr int PR = 14;                                           // 8 waves x 14 rows >= 101
            const int r = min(PR * wave + min(lane, PR - 1), BUF_ROWS - 1);
            const uint32_t idx = x2_index(quads, tb + r);
            if ((idx & ROW_DIRTY) && lane < PR && PR * wave + lane < BUF_ROWS) {
                const uint32_t slot = atomicAdd(reinterpret_cast<uint32_t*>(smem + DIRTY_OFF), 1u);
                smem[DIRTY_OFF + 4 + slot] = (unsigned char)r;
            }
            u32x4 v[PR / 2];
#pragma unroll
            for (int i = 0; i < PR / 2; ++i) {
                const uint32_t r0 = __builtin_amdgcn_readlane(idx, 2 * i), r1 = __builtin_amdgcn_readlane(idx, 2 * i + 1);
                const uint32_t row = (lane < 32 ? r0 : r1) & ~ROW_DIRTY;
                const unsigned char* p = reinterpret_cast<const unsigned char*>(a.x2_tbl) + (size_t)row * (C * 4) + (lane & 31) * 16;
                v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
            }
            if (!helper)
                wva_issue(w0, make_wrsrc(a.wva_tbl, (int)(WvaTable::ROWS * WvaTable::ROW_BYTES)), wva_step_index_q(quads, s_lo * FTT, hw, lane), lane);
#pragma unroll
            for (int i = 0; i < PR / 2; ++i) {
                const int rr = PR * wave + 2 * i + (lane >> 5);
                const uint32_t r0 = __builtin_amdgcn_readlane(idx, 2 * i), r1 = __builtin_amdgcn_readlane(idx, 2 * i + 1);
                const bool dirty = ((lane < 32 ? r0 : r1) & ROW_DIRTY) != 0u;
                if (rr < BUF_ROWS && !dirty) *reinterpret_cast<u32x4*>(buf + rr * ROWX + (lane & 31) * 16) = v[i];
            }
            __syncthreads();
            GNN_TICK(6)
            if (dirty_count(smem)) {
                dirty_rows_fill_prologue(smem, buf, tb, quads, a.pt_tbl, a.conv2_b, tid);
                __syncthreads();
                if (tid == 0) *reinterpret_cast<uint32_t*>(smem + DIRTY_OFF) = 0u;
            }
            __syncthreads();
            GNN_TICK(7)
        }

        if (!helper) {
            __builtin_amdgcn_s_setprio(2);
            const wrsrc_t cw = make_wrsrc(a.tcw3 + woff, 64 * WUNIT_B - woff);
            const wrsrc_t vw = make_wrsrc(a.wv_w + woff, 8 * WUNIT_B - woff);
            const wrsrc_t tblr = make_wrsrc(a.wva_tbl, (int)(WvaTable::ROWS * WvaTable::ROW_BYTES));
            const wrsrc_t yp_w = make_wrsrc(reinterpret_cast<const unsigned char*>(a.yp + wi * 2 * (size_t)POOLED * C), 2 * POOLED * C * 4);
            WU ring[RINGT];
            prime_tc(ring, cw, 0, lane);
            wva_pool_store<YP_NT>(w0, yp_w, s_lo * FTT, hw, lane);               // head A's rows of the first step (requested beside the first x2 rows)
            {                                                                    // V3 chunk 1 of the first step (the helpers make chunk 0)
                const HLane h0 = hlane(smem, (s_lo & 1) * BUF_BYTES, hw, lane);
                Raw16 rm;
                load_rows(rm, h0, 1);
                transform_store(rm, h0, 1);
            }
            uint32_t wva_next = wva_step_index_q(quads, (s_lo + 1) * FTT, hw, lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            GNN_TICK(6)
#pragma unroll 1
            for (int step = s_lo; step < s_hi; ++step) {
                const int t0 = step * FTT;
                const int xoff = (step & 1) * BUF_BYTES, yoff = BUF_BYTES - xoff;
                f32x16 acc[NXI];
#pragma unroll
                for (int xi = 0; xi < NXI; ++xi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[xi][r] = 0.f;
                GNN_TICK(5)
                conv_tc(smem, cw, 0, ring, acc, lane, jitter_state);             // c_0 .. c_7, conv3
                GNN_TICK(0)
                prime_wv(ring, vw, lane);
                epilogue_x3(smem + xoff, acc, a.inv_s3, bias_s, hw, lane);
                GNN_TICK(1)
                TC_BARRIER_W();                                                  // ---- E: x3 is in buf[s & 1]
                GNN_TICK(2)
                {
                    WvaRows wr;
                    wva_issue(wr, tblr, wva_next, lane);                         // head A's table rows of the next step (unconditionally, as gnn_fused_tc.hip)
                    f32x16 ac[NMB];
#pragma unroll
                    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) ac[mb][r] = 0.f;
                    wv_tile(smem, xoff + CARRY * ROWX, ring, ac, lane);
                    prime_tc(ring, cw, 0, lane);
                    wv_pool_store<YP_NT>(ac, yp_w, POOLED * C * 4, t0, hw, lane);
                    GNN_TICK(3)
                    if (step + 1 < s_hi) {
                        const HLane hn = hlane(smem, yoff, hw, lane);            // V3 chunk 1 of the next step: x2(s+1) is in buf[(s+1) & 1] since c_7
                        Raw16 rm;
                        load_rows(rm, hn, 1);
                        transform_store(rm, hn, 1);
                        wva_pool_store<YP_NT>(wr, yp_w, t0 + FTT, hw, lane);
                    }
                    wva_next = wva_step_index_q(quads, t0 + 2 * FTT, hw, lane);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // chunk 1's fragments have landed before c_0 releases the readers
                    GNN_TICK(4)
                }
            }
        } else {
            __builtin_amdgcn_s_setprio(3);
            if constexpr (PROF) tick_ = __builtin_readcyclecounter();
            Raw16 ra, rb;
            PairPass p0, p1, p2;          // head B's three passes: all requested in front of E (three sets fit: the helper role has no x1 gather to hold)
            p0.u = p1.u = p2.u = 0;
            // Per-step bookkeeping of the helpers - head A's entry of this thread (MpaTable read), the entry ranges of both heads - is
            // fetched at the END of the step before, in front of the x2 rows' requests: a wave's loads return in order and the compiler
            // waits with vmcnt(0) for loads it issued under a condition, so anything consumed while the x2 rows travel would wait for
            // their whole round trip to HBM (measured: +3 k cycles per step).
            struct StepA {
                int ea, ea_end, ua;
            };
            StepA cur, nxt;               // this step; the next one (position requested behind c_7, located at the top of its own iteration: behind E)
            cur.ea = cur.ea_end = cur.ua = 0;
            float va = 0.f;               // MpaTable value of cur's entry (requested at the top of the iteration, behind the x2 rows; stored behind c_7)
            bool va_slow = false, nxt_slow = false;
            const float* nxt_ptr = a.mpa_tbl;
            auto fetch = [&](int st, bool valid) {               // nxt = entry and position of step st (valid: st < s_hi); entry ranges from LDS
                nxt.ea = valid ? bkt[st] + ht : 0;
                nxt.ea_end = valid ? bkt[st + 1] : 0;
                nxt.ua = nxt.ea < nxt.ea_end ? a.pos_sorted[0][nxt.ea] : 0;
            };
            // nxt -> where head A's pair product of that entry sits in MpaTable.  This CONSUMES the position: the compiler waits for it
            // with vmcnt(0) (a load under a condition), so it happens where nothing is pending - first thing behind E and its explicit wait
            auto locate = [&]() {
                nxt_slow = false;
                uint32_t code = 0, dirty = 0;
                if (nxt.ea < nxt.ea_end) {
                    kmer_q<9>(quads, max(nxt.ua - 5, 0), code, dirty);
                    const bool nn = all_n_tokens<9>(dirty);
                    nxt_slow = nxt.ua < 5 || (dirty != 0u && !nn);
                }
                nxt_ptr = a.mpa_tbl + ((size_t)max(nxt.ea, 0) * K9_ROWS + (dirty == 0u ? code : K9_NN));
            };
            fetch(s_lo, true);
            {                                                                    // V3 chunk 0 of the first step
                const HLane h0 = hlane(smem, (s_lo & 1) * BUF_BYTES, hw, lane);
                load_rows(ra, h0, 0);
                transform_store(ra, h0, 0);
                load_rows(ra, h0, 2);
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);                                  // vmcnt(0): the first entry's position (the loop's top consumes it)

            // The loop is ROTATED: an iteration starts behind E of the step before and ends with E of its own step, so that the x2 rows
            // of the next step are requested and stored inside ONE iteration (carried around the loop they cost a vmcnt(0) at the loop
            // header) and still travel beside head B's pair products of the step before instead of beside conv3's weight stream: the
            // vector memory pipe is what bounds the conv loop (512 KB of weights per step at 52 of its 64 B/clk), and 52 KB of rows
            // arriving in its first units cost it their share.  What IS carried around the loop - the three passes' weights, the next
            // entry's position - has landed at the latch: the explicit wait behind E.
#pragma unroll 1
            for (int step = s_lo; step < s_hi; ++step) {
                const int t0 = step * FTT;
                const int xoff = (step & 1) * BUF_BYTES, yoff = BUF_BYTES - xoff;
                const HLane hx = hlane(smem, xoff, hw, lane), hy = hlane(smem, yoff, hw, lane);
                const bool more = step + 1 < s_hi;
                const bool prev = step > s_lo;
                const PairJob jb = {smem + xoff, a.weff_b, a.pos_sorted[1], mp_w[1], t0, bkt[STEPST + 1 + step], bkt[STEPST + 1 + step + 1]};
                // head B's pair products of the step BEFORE (x3(s-1) is in the other buffer until c_6); none in front of the first step
                const PairJob jp = {smem + yoff, a.weff_b, a.pos_sorted[1], mp_w[1], t0 - FTT, prev ? bkt[STEPST + 1 + step - 1] : 0, prev ? bkt[STEPST + 1 + step] : 0};
                // ---- behind E of the step before: this step's entry of head A located, then the x2 rows of the NEXT step: 26 per wave in
                // 13 requests, all at once, with NO load pending in front of them; behind them the one long read nobody waits for before
                // c_7: head A's table.  (A wave's loads return in order and the compiler waits with vmcnt(0) for loads issued under a
                // condition: whatever is consumed while the rows travel would wait for their whole round trip to HBM - the pair products
                // below consume nothing that is pending.)
                locate();
                cur = nxt;
                va_slow = nxt_slow;
                const float* va_ptr = nxt_ptr;
                X2Rows xr;
                const uint32_t my_row = x2_rows_index(smem, quads, t0 + FTT - CARRY, hw, lane, more);
                GNN_REGION_END();
                x2_rows_issue<0, X2_PER_WAVE>(xr, a.x2_tbl, my_row, lane);
                GNN_REGION_END();
                va = cur.ea < cur.ea_end ? __builtin_nontemporal_load(va_ptr) : 0.f;       // read once: non-temporal like the x2 rows (+0.4 %)
                GNN_TICK(15)
                pass_compute(p0, jp, 0, hw, lane);
                pass_compute(p1, jp, 1, hw, lane);
                pass_compute(p2, jp, 2, hw, lane);
                pass_rest(p0, jp, 3, hw, lane);
                TC_HPRIO_HIGH();
                HBAR_W(14, 8);                                                   // c_0
                load_rows(rb, hx, 3);
                transform_store(ra, hx, 2);
                HBAR_W(9, 8);                                                    // c_1
                load_rows(ra, hx, 4);
                transform_store(rb, hx, 0);
                HBAR_W(10, 8);                                                   // c_2
                load_rows(rb, hx, 5);
                transform_store(ra, hx, 1);
                HBAR_W(10, 8);                                                   // c_3
                load_rows(ra, hx, 6);
                transform_store(rb, hx, 2);
                HBAR_W(10, 8);                                                   // c_4
                load_rows(rb, hx, 7);
                transform_store(ra, hx, 0);
                HBAR_W(10, 8);                                                   // c_5
                transform_store(rb, hx, 1);
                HBAR_W(11, 8);                                                   // c_6: V3 is complete, buf[(s+1) & 1] (x3(s-1)) is dead
                x2_rows_store(xr, smem + yoff, my_row, hw, lane);
                if (more && dirty_count(smem)) {                                 // rows the 14-mer table cannot index (the list is the next step's)
                    dirty_rows_fill<256, true>(smem, smem + yoff, t0 + FTT - CARRY, quads, a.pt_tbl, a.conv2_b, ht);
                }
                HBAR_W(12, 8);                                                   // c_7: x2(s+1) is in buf[(s+1) & 1]
                TC_HPRIO_LOW();                                                  // the matrix waves' last unit and epilogue are the critical path now
                if (ht == 0) *reinterpret_cast<volatile uint32_t*>(smem + DIRTY_OFF) = 0u;
                if (more) {                                                      // V3(s+1) chunk 0 (ring slot 0: chunk 6 was read in front of c_7)
                    load_rows(ra, hy, 0);
                    transform_store(ra, hy, 0);
                    load_rows(ra, hy, 2);
                }
                // head A's pair product of this step's entry (read from its table since the top of the step), in front of the passes'
                // requests: the wait for it covers nothing younger
                if (cur.ea < cur.ea_end) {
                    if (va_slow) va = pair_a_slow(quads, a.x1t_tbl, a.weff_a, cur.ea, cur.ua);
                    mp_w[0][cur.ea] = va;
                    for (int e = cur.ea + 256; e < cur.ea_end; e += 256) {       // a crowded step (more than 256 entries; rare)
                        const int u = a.pos_sorted[0][e];
                        uint32_t code, dirty;
                        kmer_q<9>(quads, max(u - 5, 0), code, dirty);
                        const bool nn = all_n_tokens<9>(dirty);
                        mp_w[0][e] = (u < 5 || (dirty != 0u && !nn)) ? pair_a_slow(quads, a.x1t_tbl, a.weff_a, e, u)
                                                                      : a.mpa_tbl[(size_t)e * K9_ROWS + (dirty == 0u ? code : K9_NN)];
                    }
                }
                pass_issue(p0, jb, 0, hw, lane);
                pass_issue(p1, jb, 1, hw, lane);
                pass_issue(p2, jb, 2, hw, lane);
                fetch(step + 1, more);                                           // head A's entry of the next step: its position, located behind E
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                HBAR(13, 8);                                                     // ---- E: x3 is in buf[s & 1]
                // everything requested in front of E (the three passes' weights, the next entry's position) has to be here now, and the
                // compiler must KNOW it: every pass sits under a condition, so without this it waits with vmcnt(0) at the top of each
                // pass - behind the store of the pass before (stores count in vmcnt on gfx9): four store round trips in a row
                __builtin_amdgcn_s_waitcnt(0x0F70);                              // vmcnt(0)
            }
            {                                                                    // head B's pair products of the last step
                const int ls = s_hi - 1;
                const PairJob jl = {smem + (ls & 1) * BUF_BYTES, a.weff_b, a.pos_sorted[1], mp_w[1], ls * FTT, bkt[STEPST + 1 + ls], bkt[STEPST + 1 + ls + 1]};
                pass_compute(p0, jl, 0, hw, lane);
                pass_compute(p1, jl, 1, hw, lane);
                pass_compute(p2, jl, 2, hw, lane);
                pass_rest(p0, jl, 3, hw, lane);
            }
        }
    }
    __syncthreads();
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

// ---------------------------------------------------------------- table builders (gnn_build_kmer_tables)
// x1 per 9-mer, natural channel order, exactly as the kernels' conv1 gather makes it; row 4^9 = the all-N-token 9-mer
__global__ __launch_bounds__(128) void x1tab_kernel(const float* __restrict__ pairs6, float* __restrict__ x1tab) {
    const uint32_t n = blockIdx.x;
    const int c = threadIdx.x;
    int tk[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) tk[i] = n == K9_NN ? 0 : 1 + (int)((n >> (2 * (5 - i))) & 255u);
    const int perm = ((c >> 2) & 3) * 32 + (c >> 4) * 4 + (c & 3);
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) v += pairs6[((size_t)j * PAIR_ROWS + pair_row(tk[2 * j], tk[2 * j + 1])) * C + perm];
    x1tab[(size_t)n * C + c] = vmax_raw(v, v * LRELU);
}
// P[j][n][c] = sum_k x1tab[n][k] W2[j][k][c] in f64: conv2's contribution of tap j when the 9-mer n sits at that tap
constexpr int PJ_ROWS = 64;
__global__ __launch_bounds__(128) void pj_kernel(const float* __restrict__ x1tab, const float* __restrict__ conv2_k, double* __restrict__ P) {
    __shared__ float xs[C];
    const int c = threadIdx.x, j = blockIdx.y;
    float wcol[C];
#pragma unroll
    for (int k = 0; k < C; ++k) wcol[k] = conv2_k[((size_t)j * C + k) * C + c];
    const uint32_t r_end = min((uint32_t)(blockIdx.x + 1) * PJ_ROWS, K9_ROWS);
    for (uint32_t n = blockIdx.x * PJ_ROWS; n < r_end; ++n) {
        xs[c] = x1tab[(size_t)n * C + c];
        __syncthreads();
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < C; ++k) acc = __builtin_fma((double)xs[k], (double)wcol[k], acc);
        P[((size_t)j * K9_ROWS + n) * C + c] = acc;
        __syncthreads();
    }
}
// X2Table row n14: LeakyReLU(f32(b2 + sum_j P[j][9-mer at tap j])), the sum in f64, rounded once.  32 lanes x 4 channels per row.
constexpr uint32_t X2TAB_ROWS = 128;      // rows per workgroup (a launch must stay below 2^32 threads: 4^14 rows x 32 lanes would not)
__global__ __launch_bounds__(256) void x2tab_kernel(const double* __restrict__ P, const float* __restrict__ conv2_b, float* __restrict__ tbl) {
    const int sub = threadIdx.x & 31;
    double b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = (double)conv2_b[sub * 4 + i];
    for (uint32_t k = 0; k < X2TAB_ROWS / 8; ++k) {
        const uint32_t row = blockIdx.x * X2TAB_ROWS + k * 8u + (threadIdx.x >> 5);
        if (row >= X2_ROWS) return;
        double s[4] = {b[0], b[1], b[2], b[3]};
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const uint32_t n9 = row == X2_NN ? K9_NN : (row >> (2 * (5 - j))) & 0x3FFFFu;
            const double4 p = *reinterpret_cast<const double4*>(P + ((size_t)j * K9_ROWS + n9) * C + sub * 4);
            s[0] += p.x; s[1] += p.y; s[2] += p.z; s[3] += p.w;
        }
        float4 o;
        float v;
        v = (float)s[0]; o.x = vmax_raw(v, v * LRELU);
        v = (float)s[1]; o.y = vmax_raw(v, v * LRELU);
        v = (float)s[2]; o.z = vmax_raw(v, v * LRELU);
        v = (float)s[3]; o.w = vmax_raw(v, v * LRELU);
        *reinterpret_cast<float4*>(tbl + (size_t)row * C + sub * 4) = o;
    }
}
// MpaTable[e][n] = sum_k x1tab[n][k] weff_a[e][k] in f64, rounded once: a 64 (9-mers) x 64 (entries) tile per workgroup
constexpr int MPA_T = 64, MPA_LD = C + 1;
__global__ __launch_bounds__(256) void mpa_kernel(const float* __restrict__ x1tab, const float* __restrict__ weff_a, float* __restrict__ tbl) {
    __shared__ float xa[MPA_T * MPA_LD], wb[MPA_T * MPA_LD];
    const uint32_t n0 = blockIdx.x * MPA_T;
    const int e0 = blockIdx.y * MPA_T;
    for (int i = threadIdx.x; i < MPA_T * C; i += 256) {
        const int r = i >> 7, k = i & (C - 1);
        xa[r * MPA_LD + k] = n0 + r < K9_ROWS ? x1tab[(size_t)(n0 + r) * C + k] : 0.f;
        wb[r * MPA_LD + k] = e0 + r < NPAIR ? weff_a[(size_t)(e0 + r) * C + k] : 0.f;
    }
    __syncthreads();
    const int tn = threadIdx.x & 15, te = threadIdx.x >> 4;
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    for (int k = 0; k < C; ++k) {
        double x[4], w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x[i] = (double)xa[(tn + 16 * i) * MPA_LD + k];
            w[i] = (double)wb[(te * 4 + i) * MPA_LD + k];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fma(x[i], w[j], acc[i][j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = e0 + te * 4 + j;
        if (e >= NPAIR) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t n = n0 + tn + 16 * i;
            if (n < K9_ROWS) tbl[(size_t)e * K9_ROWS + n] = (float)acc[i][j];
        }
    }
}

static void fill_args(const gnn_ctx* ctx, ArgsK& a, const uint8_t* bases) {
    const DeviceWeights& d = ctx->w;
    a.bases = bases;
    a.tcw3 = reinterpret_cast<const unsigned char*>(d.tc_frag[1]);
    a.inv_s3 = d.tc_inv_s[1];
    a.conv_b3 = d.conv_b[1];
    a.wv_w = reinterpret_cast<const unsigned char*>(d.wv_frag_h[1]);
    a.weff_b = d.weff6[1];
    for (int i = 0; i < 2; ++i) {
        a.pos_sorted[i] = d.pos_sorted[i];
        a.bucket_ptr[i] = d.bucket_ptr96[i];
    }
    a.wva_tbl = reinterpret_cast<const unsigned char*>(d.tc_wva_tbl);
    a.x2_tbl = d.tk_x2_tbl;
    a.mpa_tbl = d.tk_mpa_tbl;
    a.x1t_tbl = d.tk_x1t_tbl;
    a.pt_tbl = d.tk_pt_tbl;
    a.conv2_b = d.conv_b[0];
    a.weff_a = d.weff_sorted[0];
    a.cycles = nullptr;
    a.split = 1;
}

static void launch(const ArgsK& a, bool prof, unsigned nwin, hipStream_t stream) {
    const unsigned n = nwin * (unsigned)a.split;
    if (prof) hipLaunchKernelGGL((fused_front_tk_kernel<true>), dim3(n), dim3(512), 0, stream, a);
    else hipLaunchKernelGGL((fused_front_tk_kernel<false>), dim3(n), dim3(512), 0, stream, a);
}

}  // namespace tk

size_t kmer_tables_bytes() {
    return (size_t)tk::X2_ROWS * C * 4 + (size_t)NPAIR * tk::K9_ROWS * 4 + (size_t)(KS + 1) * tc::WvaTable::ROWS * C * 4;
}

void free_kmer_tables(gnn_ctx* ctx) {
    DeviceWeights& d = ctx->w;
    for (float** p : {&d.tk_x2_tbl, &d.tk_mpa_tbl, &d.tk_pt_tbl, &d.tk_x1t_tbl, &d.tk_yp_const, &d.tk_mp_const}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
}

// X2Table and MpaTable on the device.  `reserve` bytes must stay free behind them (workspaces of the launches to come).
int build_kmer_tables(gnn_ctx* ctx, size_t reserve) {
    using namespace tk;
    DeviceWeights& d = ctx->w;
    if (d.tk_x2_tbl && d.tk_mpa_tbl && d.tk_pt_tbl && d.tk_x1t_tbl && d.tk_yp_const) return GNN_OK;
    free_kmer_tables(ctx);
    const size_t x2_b = (size_t)X2_ROWS * C * 4, mpa_b = (size_t)NPAIR * K9_ROWS * 4;
    const size_t x1_b = (size_t)K9_ROWS * C * 4, p_b = (size_t)KS * K9_ROWS * C * 8;
    size_t free_b = 0, total_b = 0;
    GNN_HIP(hipMemGetInfo(&free_b, &total_b));
    const size_t pt_b = (size_t)KS * WvaTable::ROWS * C * 4, x1t_b = (size_t)WvaTable::ROWS * C * 4;
    const size_t need = x2_b + mpa_b + pt_b + x1t_b + x1_b + p_b + reserve;
    if (free_b < need) {
        set_error("k-mer tables: " + std::to_string(need >> 30) + " GiB needed (" + std::to_string((x2_b + mpa_b) >> 30) + " GiB of tables, the rest workspace reserve), " +
                  std::to_string(free_b >> 30) + " GiB free on the device: GNN_PREC_F16X3TC keeps serving");
        return GNN_ERR_NOMEM;
    }
    void *x1 = nullptr, *P = nullptr, *x2 = nullptr, *mpa = nullptr, *pt = nullptr, *x1t = nullptr, *eye = nullptr;
    auto fail = [&](hipError_t e, const char* what) {
        (void)hipGetLastError();
        for (void* p : {x1, P, x2, mpa, pt, x1t, eye})
            if (p) (void)hipFree(p);
        set_error(std::string("k-mer tables: ") + what + " failed: " + hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? GNN_ERR_NOMEM : GNN_ERR_HIP;
    };
    hipError_t e;
    // GNN_TABLE_TIMING=1 (debug aid): seconds of every stage on stderr (a stream synchronisation after each)
    static const bool timing = debug_switch("GNN_TABLE_TIMING");
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "genomad_nn k-mer tables: %-40s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };
    if ((e = hipMalloc(&x2, x2_b)) != hipSuccess) return fail(e, "hipMalloc of the 14-mer table");
    lap("hipMalloc 137 GB");
    if ((e = hipMalloc(&mpa, mpa_b)) != hipSuccess) return fail(e, "hipMalloc of head A's pair-product table");
    if ((e = hipMalloc(&x1, x1_b)) != hipSuccess) return fail(e, "hipMalloc of the 9-mer x1 table");
    if ((e = hipMalloc(&pt, pt_b)) != hipSuccess) return fail(e, "hipMalloc of conv2's tap tables");
    if ((e = hipMalloc(&x1t, x1t_b)) != hipSuccess) return fail(e, "hipMalloc of the x1 table");
    if ((e = hipMalloc(&eye, (size_t)C * C * 4)) != hipSuccess) return fail(e, "hipMalloc of the identity");
    if ((e = hipMalloc(&P, p_b)) != hipSuccess) return fail(e, "hipMalloc of the f64 tap tables");
    {   // x1 over WvaTable's index space = the same builder with the identity as the matrix (products with 0 and 1, summed in f64: exact)
        std::vector<float> id((size_t)C * C, 0.f);
        for (int k = 0; k < C; ++k) id[(size_t)k * C + k] = 1.f;
        if ((e = hipMemcpy(eye, id.data(), id.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "upload of the identity");
        if (build_wva_rows_table(ctx, static_cast<const float*>(eye), static_cast<float*>(x1t))) return fail(hipGetLastError(), "launch of the x1 table build");
    }
    lap("hipMalloc of the other four");
    for (int j = 0; j < KS; ++j)
        if (build_wva_rows_table(ctx, d.conv_k[0] + (size_t)j * C * C, static_cast<float*>(pt) + (size_t)j * WvaTable::ROWS * C)) return fail(hipGetLastError(), "launch of a tap-table build");
    lap("tap tables over WvaTable's rows (6 x 2.7 M)");
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(x1tab_kernel, dim3(K9_ROWS), dim3(128), 0, st, d.conv1_pairs6, static_cast<float*>(x1));
    if ((e = hipGetLastError()) != hipSuccess) return fail(e, "launch of x1tab_kernel");
    lap("x1 per 9-mer");
    hipLaunchKernelGGL(pj_kernel, dim3((K9_ROWS + PJ_ROWS - 1) / PJ_ROWS, KS), dim3(128), 0, st, static_cast<const float*>(x1), d.conv_k[0], static_cast<double*>(P));
    if ((e = hipGetLastError()) != hipSuccess) return fail(e, "launch of pj_kernel");
    lap("f64 tap tables per 9-mer");
    hipLaunchKernelGGL(x2tab_kernel, dim3((X2_ROWS + X2TAB_ROWS - 1) / X2TAB_ROWS), dim3(256), 0, st, static_cast<const double*>(P), d.conv_b[0], static_cast<float*>(x2));
    if ((e = hipGetLastError()) != hipSuccess) return fail(e, "launch of x2tab_kernel");
    lap("x2 per 14-mer (137 GB)");
    hipLaunchKernelGGL(mpa_kernel, dim3((K9_ROWS + MPA_T - 1) / MPA_T, (NPAIR + MPA_T - 1) / MPA_T), dim3(256), 0, st, static_cast<const float*>(x1),
                       d.weff_sorted[0], static_cast<float*>(mpa));
    if ((e = hipGetLastError()) != hipSuccess) return fail(e, "launch of mpa_kernel");
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return fail(e, "building the tables");
    lap("head A's pair products (8.8 GB)");
    (void)hipFree(x1);
    (void)hipFree(P);
    (void)hipFree(eye);
    x1 = P = eye = nullptr;
    d.tk_x2_tbl = static_cast<float*>(x2);
    d.tk_mpa_tbl = static_cast<float*>(mpa);
    d.tk_pt_tbl = static_cast<float*>(pt);
    d.tk_x1t_tbl = static_cast<float*>(x1t);
    // the all-N window's outputs, computed once by the kernel itself (padding skip)
    void *bn = nullptr, *yc = nullptr, *mc = nullptr;
    if ((e = hipMalloc(&bn, W)) != hipSuccess || (e = hipMalloc(&yc, (size_t)2 * POOLED * C * sizeof(float))) != hipSuccess ||
        (e = hipMalloc(&mc, (size_t)2 * NPAIR * sizeof(float))) != hipSuccess) {
        for (void* p : {bn, yc, mc})
            if (p) (void)hipFree(p);
        free_kmer_tables(ctx);
        x2 = mpa = pt = x1t = nullptr;
        return fail(e, "hipMalloc of the all-N window's outputs");
    }
    (void)hipMemsetAsync(bn, 'N', W, st);
    ArgsK a;
    fill_args(ctx, a, static_cast<const uint8_t*>(bn));
    a.mp = static_cast<float*>(mc);
    a.yp = static_cast<float*>(yc);
    a.yp_c = nullptr;
    a.mp_c = nullptr;
    launch(a, false, 1, st);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(bn);
    d.tk_yp_const = static_cast<float*>(yc);
    d.tk_mp_const = static_cast<float*>(mc);
    if (e != hipSuccess) {
        free_kmer_tables(ctx);
        x2 = mpa = pt = x1t = nullptr;
        return fail(e, "the all-N window's launch");
    }
    return GNN_OK;
}

int launch_front_tk(gnn_ctx* ctx, const uint8_t* bases, int64_t n) {
    using namespace tk;
    if (!ctx->w.tk_x2_tbl || !ctx->w.tk_mpa_tbl) {
        set_error("f16x3tk: the k-mer tables have not been built (gnn_build_kmer_tables)");
        return GNN_ERR_STATE;
    }
    if (reinterpret_cast<uintptr_t>(bases) & 3u) {
        set_error("f16x3tk: the window buffer must be 4-byte aligned");
        return GNN_ERR_ARG;
    }
    ArgsK a;
    fill_args(ctx, a, bases);
    a.mp = ctx->ws.mp;
    a.yp = ctx->ws.yp;
    a.yp_c = ctx->c6_pad_skip ? ctx->w.tk_yp_const : nullptr;
    a.mp_c = ctx->c6_pad_skip ? ctx->w.tk_mp_const : nullptr;
    a.cycles = ctx->phase_cycles;
    if (ctx->time_split && n > 0 && ctx->cu_count > 0) a.split = (int)std::max<int64_t>(1, std::min<int64_t>(4, ctx->cu_count / n));
    ctx->last_split = a.split;
    launch(a, ctx->phase_cycles != nullptr, (unsigned)n, ctx->stream);
    GNN_HIP(hipGetLastError());
    return GNN_OK;
}

}  // namespace gnn
