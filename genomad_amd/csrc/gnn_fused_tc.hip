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
#include "gnn_tc_dev.h"

namespace gnn {
namespace tc {

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

}  // namespace tc

// tbl[row][c] = sum_k x1(row)[k] w[k][c] for every row of WvaTable's index space (all 9-mers, window starts, 9-mers with a non-ACGT
// byte): head A's y @ w_v table with w = w_v A, and the tap tables of gnn_fused_tk.hip with w = one tap of conv2
int build_wva_rows_table(gnn_ctx* ctx, const float* w_kc, float* tbl) {
    using namespace tc;
    hipLaunchKernelGGL(wva_table_kernel, dim3((WvaTable::ROWS + WVA_BUILD_ROWS - 1) / WVA_BUILD_ROWS), dim3(128), 0, ctx->stream, ctx->w.conv1_pairs6, w_kc, tbl);
    GNN_HIP(hipGetLastError());
    return GNN_OK;
}

namespace tc {

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
        const int rc = build_wva_rows_table(ctx, d.w_v[0], d.tc_wva_tbl);
        if (rc) return rc;
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
