// Pieces shared by the streaming fused front ends that work on (row, 32-channel block) units held by one lane
// (gnn_fused_c6.hip: f16 + MX-fp6 corrections; gnn_fused_x3.hip: split-f16 / split-bf16, three passes): scheduling-region
// fence, buffer resources, the per-step pair-row pipeline of the conv1 gather, the gather's loads, and the IGLOO
// pair-product loop.  What depends on a kernel's LDS row format (the row producer, the row reader of the pair products)
// stays in that kernel's file and is passed in as a policy.
#pragma once
#include <utility>

#include "gnn_fused_common.h"

namespace gnn {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define GNN_REGION_END()               \
    __builtin_amdgcn_sched_barrier(0); \
    asm volatile("" ::: "memory")

// Weight fragments come through buffer loads: resource descriptor (SGPRs) + 32-bit lane offset (VGPR) + wave-uniform
// step offset (SGPR) + immediate, so no 64-bit per-lane address ever lives in VGPRs (with global loads the compiler
// kept one address pair per (step, fragment) of the ring: 73 dwords of spills).
typedef __amdgpu_buffer_rsrc_t wrsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ wrsrc_t make_wrsrc(const unsigned char* base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base), 0, bytes, 0x00020000);
}

template <int... Ks, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, Ks...>, F&& f) {
    (f(std::integral_constant<int, Ks>{}), ...);
}

// Pair row of the adjacent positions (t, t + 1) from the aligned 8 bytes that hold bases[t .. t+4] (sequence.py:170-193 in
// closed form, gnn_fused_common.h).  The bytes are requested a step before they are turned into pair rows (prow_fetch ->
// prow_make), and the pair rows of step s+2 are written between the barriers B3 and B4 of step s into the buffer of that
// parity: a workgroup barrier always lies between a thread writing a pair row and the other waves' gathers reading it, and
// no memory round trip sits in front of a step.
__device__ __forceinline__ int prow_base(int t) { return min(max(t, 0) & ~3, W - 8); }
__device__ __forceinline__ void prow_fetch(const uint8_t* __restrict__ bases, int t, uint32_t& lo, uint32_t& hi) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(bases + prow_base(t));   // windows start 4-byte aligned (checked at launch)
    lo = src[0];
    hi = src[1];
}
__device__ __forceinline__ int token_from(uint32_t lo, uint32_t hi, int a, int q) {
    if (q < 0) return -1;
    if (q >= T) return 0;
    const uint32_t w4 = (uint32_t)((((unsigned long long)hi << 32) | lo) >> (8 * (q - a)));
    const int c0 = base_code_f(w4 & 255u), c1 = base_code_f((w4 >> 8) & 255u), c2 = base_code_f((w4 >> 16) & 255u),
              c3 = base_code_f(w4 >> 24);
    return (c0 | c1 | c2 | c3) < 0 ? 0 : 1 + c0 * 64 + c1 * 16 + c2 * 4 + c3;
}
__device__ __forceinline__ uint16_t prow_make(uint32_t lo, uint32_t hi, int t) {
    const int a = prow_base(t);
    return (uint16_t)pair_row(token_from(lo, hi, a, t), token_from(lo, hi, a, t + 1));
}

// conv1 gather of the streaming kernels: pair tables stored as [i 0..3][block 4][half 2][4 ch] (pack_fused_c6_weights), a lane
// pair owning block p of two neighbouring rows; see gnn_fused_c6.hip for the layout's rationale.
struct GatherUnit {
    f32x4 v[2][3][4];      // [row A / B][table][i]
};
__device__ __forceinline__ float dpp_xor1(float v) {     // value of lane ^ 1 (quad_perm [1,0,3,2])
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor2(float v) {     // value of lane ^ 2 (quad_perm [2,3,0,1])
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}
// ua = first row of the lane pair (even), pq = 2 p + q
__device__ __forceinline__ void gather_issue(GatherUnit& g, const uint16_t* __restrict__ prow, const float* __restrict__ pt, int ua, int pq) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const uint32_t row = prow[ua + r + 2 * j];
            const float* src = pt + ((size_t)j * PAIR_ROWS + row) * C + pq * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) g.v[r][j][i] = *reinterpret_cast<const f32x4*>(src + i * 32);
        }
}
// the three table rows summed (bias is folded into table 0): 32 registers instead of 96 while a round waits for its slot
struct GatherSum {
    f32x4 sa[4], sb[4];    // row A / row B, channels 16 q + 4 i ..
};
__device__ __forceinline__ void gather_sum(GatherSum& o, const GatherUnit& g) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o.sa[i] = g.v[0][0][i] + g.v[0][1][i] + g.v[0][2][i];
        o.sb[i] = g.v[1][0][i] + g.v[1][1][i] + g.v[1][2][i];
    }
}

// IGLOO pair dot products (igloo.py:192-204 with w_mult * w_summer folded) of head B for the previous step's x3 rows and of
// head A for this step's x1 rows in ONE loop: 4 lanes per entry, one 32-channel block per lane.  Compute::run(w, job, e, u, p)
// reads block p of row u in the kernel's LDS row format and stores the entry's dot product (lane p == 0).
struct PairJob {
    const unsigned char* xbuf;
    const float* weff;
    const int32_t* pos;
    float* mp;
    int t0, e, e_end;
};
// folded weights of one (entry, block) unit: 32 f32.  Layout of this kernel's copy (pack_fused_c6_weights): entries in pairs,
// [e >> 1][i 0..7][e & 1][block 4][4 ch] -> the 8 lanes of two neighbouring entries read one 128-B line per load
struct PairW {
    float4 w[8];
};
__device__ __forceinline__ void pair_load_w(PairW& o, const PairJob& jb, int e, int p) {
    const float* wr = jb.weff + (size_t)(e >> 1) * (2 * C) + (e & 1) * 16 + p * 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) o.w[i] = *reinterpret_cast<const float4*>(wr + i * 32);
}
// Both heads in one loop, 64 entries of each per pass (4 lanes per entry, 16 entries per wave).  The weights of pass
// k+1 are requested before pass k is computed (two register sets, the loop is unrolled by two), and the positions one
// pass further ahead still: the loads of a step, whose round trip is several thousand cycles beside the matrix waves'
// weight stream, overlap instead of queueing one round trip per pass.
template <class Compute>
__device__ __forceinline__ void m_partials2(PairJob jb, PairJob ja, int wave, int lane) {
    const int p = lane & 3;
    PairJob job[2] = {jb, ja};
    int e[2], u[2], un[2] = {0, 0};
    PairW wa[2], wb[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        e[h] = job[h].e + wave * 16 + (lane >> 2);
        u[h] = job[h].t0;
        if (e[h] < job[h].e_end) {
            u[h] = job[h].pos[e[h]];
            pair_load_w(wa[h], job[h], e[h], p);
        }
    }
    // the 4 lanes of an entry share e: a lane group enters / leaves together and the width-4 shuffles only read active lanes
    while (e[0] < job[0].e_end || e[1] < job[1].e_end) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (e[h] + 64 < job[h].e_end) {
                pair_load_w(wb[h], job[h], e[h] + 64, p);
                un[h] = job[h].pos[e[h] + 64];
            }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (e[h] < job[h].e_end) Compute::run(wa[h], job[h], e[h], u[h], p);
            e[h] += 64;
            u[h] = un[h];
        }
        if (!(e[0] < job[0].e_end || e[1] < job[1].e_end)) break;
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (e[h] + 64 < job[h].e_end) {
                pair_load_w(wa[h], job[h], e[h] + 64, p);
                un[h] = job[h].pos[e[h] + 64];
            }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (e[h] < job[h].e_end) Compute::run(wb[h], job[h], e[h], u[h], p);
            e[h] += 64;
            u[h] = un[h];
        }
    }
}


}  // namespace gnn
