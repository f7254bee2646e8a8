// Pricing probe for VERDICT r03 item 1: Toom-Cook F(3,6) minimal filtering over time for conv2 / conv3 on the f16x3 limbs.
//
// What it measures: the matrix waves' conv loops of the fused front end, whole chip (one 512-thread workgroup per window, 4
// matrix waves + 4 helper waves, ~150 KB of LDS so that one workgroup owns a CU), in two shapes on random f16 data:
//
//   direct : today's kernel (gnn_fused_x3.hip): a 128-row step = 2 convs x 48 k16 units x [12 MFMAs, 8 ds_read_b128 of
//            activation fragments, 2 buffer_load_b128 of weight fragments (hi | lo)]; 47 steps per window.
//   tc     : F(3,6): a 96-row step = 32 tiles = ONE 32-column MFMA block per transform point xi; 2 convs x 8 k16 units x 8 xi x
//            [3 MFMAs, 2 ds_read_b128 of transformed activations V_xi (hi | lo), 2 buffer_load_b128 of transformed weights
//            U_xi (hi | lo)]; 63 steps per window.  0.444x the MFMAs of the direct form.
//
// The point: a transformed weight fragment feeds 3 MFMAs (one 32-tile block) where a direct one feeds 12 (four 32-row blocks),
// and the number of tiles in flight per CU is fixed by the LDS (two activation buffers) and the accumulator registers (8 xi x
// 16 per wave), so F(3,6) asks the L2 -> CU path for 4x the weight bytes per MFMA: 85 B/clk/CU inside the conv loops against a
// measured chip-wide L2 ceiling of 56 B/clk/CU (MI355X_MICROARCH.md, L2: 34.5 TB/s).  Flags switch the weight stream, the
// LDS reads and a stand-in for the helpers' input transform (+ one workgroup barrier per k16 unit: the V ring) on and off.
//
// Round 5: -DNXI=9 prices F(4,6) (9 points, 4 rows per tile, 128-row steps, 47 steps, a V ring slot of 18 KB, 144 accumulator
// registers; -DRINGT_SLOTS=6 leaves the weight ring the 16 registers the ninth accumulator takes) with the same loops: the one design
// candidate the round-5 time model (profiles/r05/MODEL.md) ranks above the shipped kernel.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DNXI=9 -DRINGT_SLOTS=6] -o build_variants/probe_tc_loop scripts/probe_tc_loop.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

#define REGION_END()                   \
    __builtin_amdgcn_sched_barrier(0); \
    asm volatile("" ::: "memory")

template <int... Ks, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, Ks...>, F&& f) {
    (f(std::integral_constant<int, Ks>{}), ...);
}

constexpr int SMEM = 150 * 1024;
constexpr int ROWX = 528;
struct WU {
    uint4 h, l;
};

__device__ __forceinline__ f32x16 mma(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void load_wu(WU& w, rsrc_t r, uint32_t l16, int soff) {
    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, l16, soff, 0);
    const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(r, l16 + 1024, soff, 0);
    w.h = make_uint4(a[0], a[1], a[2], a[3]);
    w.l = make_uint4(b[0], b[1], b[2], b[3]);
}

// ---------------------------------------------------------------- direct form (the loop of gnn_fused_x3.hip, 4 row blocks)
constexpr int RINGD = 4;
struct XU {
    uint4 h[4], l[4];
};
template <bool LW, bool LX, int OFFN>
__device__ __forceinline__ void unit_direct(const WU& wc, WU& wl, const XU& xc, XU& xl, const unsigned char* xh, rsrc_t wr, int wnext,
                                            uint32_t l16, f32x16 (&acc)[4]) {
    if constexpr (LX) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            xl.h[mb] = *reinterpret_cast<const uint4*>(xh + OFFN + mb * 32 * ROWX);
            xl.l[mb] = *reinterpret_cast<const uint4*>(xh + OFFN + 256 + mb * 32 * ROWX);
        }
    }
    if constexpr (LW) load_wu(wl, wr, l16, wnext);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        if (mb % 2 == 0) {
            acc[mb] = mma(wc.l, xc.h[mb], acc[mb]);
            acc[mb] = mma(wc.h, xc.h[mb], acc[mb]);
            acc[mb] = mma(wc.h, xc.l[mb], acc[mb]);
        } else {
            acc[mb] = mma(wc.h, xc.l[mb], acc[mb]);
            acc[mb] = mma(wc.h, xc.h[mb], acc[mb]);
            acc[mb] = mma(wc.l, xc.h[mb], acc[mb]);
        }
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (LX && i < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (LW && (i == 1 || i == 5)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    REGION_END();
}

template <bool LW, bool LX>
__device__ __forceinline__ void conv_direct(const unsigned char* smem, rsrc_t wr, int woff, WU (&ring)[RINGD], f32x16 (&acc)[4], int lane) {
    constexpr int NK = 48;
    uint32_t rowoff = (uint32_t)(lane & 31) * ROWX + (uint32_t)(lane >> 5) * 16u;
    asm volatile("" : "+v"(rowoff));
    const unsigned char* xh = smem + rowoff;
    const uint32_t l16 = (uint32_t)lane * 16u;
    XU xa, xb;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        xa.h[mb] = *reinterpret_cast<const uint4*>(xh + mb * 32 * ROWX);
        xa.l[mb] = *reinterpret_cast<const uint4*>(xh + 256 + mb * 32 * ROWX);
        xb.h[mb] = xa.h[mb];
        xb.l[mb] = xa.l[mb];
    }
    REGION_END();
    static_for(std::make_integer_sequence<int, NK>{}, [&](auto kc) {
        constexpr int k = decltype(kc)::value, kn = (k + 1) % NK;
        constexpr int OFFN = (kn / 8) * ROWX + (kn % 8) * 32;
        constexpr int kw = (k + RINGD - 1) % NK;       // the stream wraps onto the same conv's first units: the same bytes per step
        if constexpr (k % 2 == 0)
            unit_direct<LW, LX, OFFN>(ring[k % RINGD], ring[(k + RINGD - 1) % RINGD], xa, xb, xh, wr, woff + kw * 8192, l16, acc);
        else
            unit_direct<LW, LX, OFFN>(ring[k % RINGD], ring[(k + RINGD - 1) % RINGD], xb, xa, xh, wr, woff + kw * 8192, l16, acc);
    });
}

// ---------------------------------------------------------------- Toom-Cook form
// V ring in LDS: slot = one k16 unit: [xi 8][hi | lo][lane 64] x 16 B = 16 KB; 3 slots.  Weights: [unit 8][xi 8][nblk 4][hi | lo] x 1 KiB.
#ifndef NXI
#define NXI 8                            // transform points: 8 = F(3,6) (3 rows per tile, 96-row steps), 9 = F(4,6) (4 rows per tile, 128-row steps)
#endif
constexpr int TROWS = NXI - 5;           // output rows per tile
#ifndef RINGT_SLOTS
#define RINGT_SLOTS 8
#endif
constexpr int RINGT = RINGT_SLOTS;       // weight ring slots of one (unit, xi): RINGT - 1 in flight ahead of the MFMAs
constexpr int VSLOT = NXI * 2048, VRING = 3;
constexpr int VOFF = SMEM - VRING * VSLOT;
struct XV {
    uint4 h, l;
};
template <bool LW, bool LX, int VOFFN>
__device__ __forceinline__ void xi_tc(const WU& wc, WU& wl, const XV& vc, XV& vl, const unsigned char* vb, rsrc_t wr, int wnext, uint32_t l16,
                                      f32x16& acc) {
    if constexpr (LX) {
        vl.h = *reinterpret_cast<const uint4*>(vb + VOFFN);
        vl.l = *reinterpret_cast<const uint4*>(vb + VOFFN + 1024);
    }
    if constexpr (LW) load_wu(wl, wr, l16, wnext);
#ifdef NOMMA      // measurement variant: the helpers' transform and the barriers alone - what the helper code takes with the SIMD to itself
    acc[0] += __uint_as_float(wc.l.x ^ vc.h.x) + __uint_as_float(wc.h.y ^ vc.l.y);
#else
    acc = mma(wc.l, vc.h, acc);
    acc = mma(wc.h, vc.h, acc);
    acc = mma(wc.h, vc.l, acc);
#endif
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if (LX) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    if (LW) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if (LX) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    if (LW) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    REGION_END();
}

// stand-in for the helpers' input transform of ONE k16 unit's quarter (a helper wave = 32 tiles x 2 halves, 2 channels per
// lane): 8 input rows x 2 channels (hi | lo planes) -> f32, the 8-point transform B^T of the points {0, +-1, +-2, +-1/2, inf}
// (26 VALU per channel), split into f16 hi | lo, 16 ds_write_b32 into the ring slot.  Same instruction mix as the real thing.
struct HRaw {
    uint32_t h[NXI], l[NXI];
};
__device__ __forceinline__ void helper_load(HRaw& r, const unsigned char* smem, int unit, int hw, int lane) {
#ifdef HELP_NOLDS        // measurement variant: barriers only on the helper side
    for (int j = 0; j < NXI; ++j) r.h[j] = r.l[j] = (uint32_t)(unit + lane + j);
    return;
#endif
    // lane -> (tile, k half, channel pair) so that both the row reads and the fragment stores are bank-conflict free: the 4 lanes
    // of a (tile, half) cover its 8 channels = 16 consecutive bytes; a wave covers 16 (tile, half) combinations
    const int pr = lane & 3, th = hw * 16 + (lane >> 2), tile = th & 31, half = th >> 5;
    const unsigned char* xr = smem + (TROWS * tile) * ROWX + (unit * 16 + half * 8 + pr * 2) * 2;
#pragma unroll
    for (int j = 0; j < NXI; ++j) {
        r.h[j] = *reinterpret_cast<const uint32_t*>(xr + j * ROWX);
        r.l[j] = *reinterpret_cast<const uint32_t*>(xr + j * ROWX + 256);
    }
}
__device__ __forceinline__ void helper_transform(const HRaw& r, unsigned char* smem, int slot, int hw, int lane) {
#ifdef HELP_NOLDS
    asm volatile("" ::"v"(r.h[0]), "v"(r.l[7]));
    return;
#endif
#ifdef HELP_NOVALU       // measurement variant: the 16 loads and 16 stores without the arithmetic between them
    {
        unsigned char* vo = smem + VOFF + slot * VSLOT + (hw * 64 + lane) * 4;
#pragma unroll
        for (int xi = 0; xi < NXI; ++xi) {
            *reinterpret_cast<uint32_t*>(vo + xi * 2048) = r.h[xi];
            *reinterpret_cast<uint32_t*>(vo + xi * 2048 + 1024) = r.l[xi];
        }
        return;
    }
#endif
    float d[NXI][2];
#pragma unroll
    for (int j = 0; j < NXI; ++j) {
        const f16x2 hh = __builtin_bit_cast(f16x2, r.h[j]), ll = __builtin_bit_cast(f16x2, r.l[j]);
        d[j][0] = (float)hh[0] + (float)ll[0];
        d[j][1] = (float)hh[1] + (float)ll[1];
    }
    unsigned char* vo = smem + VOFF + slot * VSLOT + (hw * 64 + lane) * 4;     // fragment lane th = hw * 16 + (lane >> 2), its dword lane & 3
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        float v[NXI];
        const float d0 = d[0][c], d1 = d[1][c], d2 = d[2][c], d3 = d[3][c], d4 = d[4][c], d5 = d[5][c], d6 = d[6][c], d7 = d[7][c];
        v[0] = fmaf(d4 - d2, 5.25f, d0 - d6);
        const float t1 = fmaf(d4, -4.25f, d2 + d6), t2 = fmaf(d3, -4.25f, d1 + d5);
        v[1] = t1 + t2;
        v[2] = t1 - t2;
        const float t3 = fmaf(d4, -1.25f, fmaf(d2, 0.25f, d6)), t4 = fmaf(d5, 2.f, fmaf(d3, -2.5f, d1 * 0.5f));
        v[3] = t3 + t4;
        v[4] = t3 - t4;
        const float t5 = fmaf(fmaf(d4, -1.25f, d2), 4.f, d6), t6 = fmaf(d5, 0.5f, fmaf(d3, -2.5f, d1 * 2.f));
        v[5] = t5 + t6;
        v[6] = t5 - t6;
        v[7] = fmaf(d3 - d5, 5.25f, d7 - d1);
#if NXI == 9      // stand-in for the ninth point of F(4,6) (1/4: the densest row of B^T) and the ninth input row: same instruction classes
        {
            const float d8 = d[8][c];
            const float t7 = fmaf(d4, -5.3125f, fmaf(d2, 1.0625f, d6)), t8 = fmaf(d5, 4.25f, fmaf(d3, -21.25f, d1 * 4.f));
            v[8] = fmaf(d7, 0.25f, t7 + t8) + d8;
            v[0] = fmaf(d8, 0.0625f, v[0]);
            v[7] = fmaf(d8, -0.25f, v[7]);
        }
#endif
#pragma unroll
        for (int i = 0; i < NXI; ++i) d[i][c] = v[i];
    }
#pragma unroll
    for (int xi = 0; xi < NXI; ++xi) {
        const f32x2 v = {d[xi][0], d[xi][1]};
        const f16x2 hi = __builtin_convertvector(v, f16x2);
        const f32x2 back = {(float)hi[0], (float)hi[1]};
        const f16x2 lo = __builtin_convertvector(v - back, f16x2);
#ifdef HELP_NOSTORE      // measurement variant: the transform's VALU work without its 16 LDS stores
        asm volatile("" ::"v"(__builtin_bit_cast(uint32_t, hi)), "v"(__builtin_bit_cast(uint32_t, lo)));
#else
        *reinterpret_cast<uint32_t*>(vo + xi * 2048) = __builtin_bit_cast(uint32_t, hi);
        *reinterpret_cast<uint32_t*>(vo + xi * 2048 + 1024) = __builtin_bit_cast(uint32_t, lo);
#endif
    }
}

template <bool LW, bool LX, bool HELP>
__device__ __forceinline__ void conv_tc(unsigned char* smem, rsrc_t wr, int woff, WU (&ring)[RINGT], f32x16 (&acc)[NXI], int lane) {
    const uint32_t l16 = (uint32_t)lane * 16u;
    uint32_t voff = (uint32_t)VOFF + l16;
    asm volatile("" : "+v"(voff));
    const unsigned char* vb = smem + voff;
    XV va, vc;
    va.h = *reinterpret_cast<const uint4*>(vb);
    va.l = *reinterpret_cast<const uint4*>(vb + 1024);
    vc = va;
    REGION_END();
    static_for(std::make_integer_sequence<int, 8 * NXI>{}, [&](auto kc) {
        constexpr int k = decltype(kc)::value, kn = (k + 1) % (8 * NXI);          // k = unit * NXI + xi
        constexpr int VOFFN = ((kn / NXI) % VRING) * VSLOT + (kn % NXI) * 2048;
        constexpr int kw = (k + RINGT - 1) % (8 * NXI);
        // the V ring's hand-over: one barrier per k16 unit.  A bare s_barrier: __syncthreads() would also drain this wave's weight
        // loads in flight (s_waitcnt vmcnt(0)) and expose an L2 round trip per unit; the matrix waves wait for nothing of their own
        if constexpr (HELP && k % NXI == 0 && k > 0) __builtin_amdgcn_s_barrier();
        if constexpr (k % 2 == 0)
            xi_tc<LW, LX, VOFFN>(ring[k % RINGT], ring[(k + RINGT - 1) % RINGT], va, vc, vb, wr, woff + kw * 8192, l16, acc[k % NXI]);
        else
            xi_tc<LW, LX, VOFFN>(ring[k % RINGT], ring[(k + RINGT - 1) % RINGT], vc, va, vb, wr, woff + kw * 8192, l16, acc[k % NXI]);
    });
}

template <int MODE, bool LW, bool LX, bool HELP>      // MODE 0 direct, 1 tc
__global__ __launch_bounds__(512, 2) void probe_kernel(const unsigned char* wts, int wbytes, int steps, float* sink, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // random f16 data in [0.5, 2) with random signs in the whole LDS
    uint32_t x = 0x9E3779B9u * (tid + 1) + blockIdx.x;
    for (int i = tid; i < SMEM / 4; i += 512) {
        x = x * 1664525u + 1013904223u;
        const uint32_t a = (x >> 16) & 0x83FFu, b = x & 0x83FFu;
        reinterpret_cast<uint32_t*>(smem)[i] = ((a | 0x3800u | ((x >> 3) & 0x400u)) << 16) | (b | 0x3800u | ((x >> 5) & 0x400u));
    }
    __syncthreads();
    const unsigned long long t_start = __builtin_readcyclecounter();
    float s = 0.f;
    if (wave < 4) {
        __builtin_amdgcn_s_setprio(2);
        const rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(wts), 0, wbytes, 0x00020000);
        const int woff = wave * 2048;
        if constexpr (MODE == 0) {
            WU ring[RINGD];
#pragma unroll
            for (int u = 0; u < RINGD - 1; ++u) load_wu(ring[u], wr, lane * 16u, woff + u * 8192);
            ring[RINGD - 1] = ring[0];
#pragma unroll 1
            for (int step = 0; step < steps; ++step) {
#pragma unroll 1
                for (int cv = 0; cv < 2; ++cv) {
                    f32x16 acc[4];
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
                    conv_direct<LW, LX>(smem + cv * 70224, wr, woff + cv * 48 * 8192, ring, acc, lane);
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) s += acc[mb][0] + acc[mb][7];
                }
            }
        } else {
            WU ring[RINGT];
#pragma unroll
            for (int u = 0; u < RINGT - 1; ++u) load_wu(ring[u], wr, lane * 16u, woff + u * 8192);
            ring[RINGT - 1] = ring[0];
#pragma unroll 1
            for (int step = 0; step < steps; ++step) {
#pragma unroll 1
                for (int cv = 0; cv < 2; ++cv) {
                    f32x16 acc[NXI];
#pragma unroll
                    for (int xi = 0; xi < NXI; ++xi)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[xi][r] = 0.f;
                    if constexpr (HELP) __builtin_amdgcn_s_barrier();
                    conv_tc<LW, LX, HELP>(smem, wr, woff + cv * 8 * NXI * 8192, ring, acc, lane);
#pragma unroll
                    for (int xi = 0; xi < NXI; ++xi) s += acc[xi][0] + acc[xi][7];
                }
            }
        }
    } else if (HELP && MODE == 1) {
        const int hw = wave - 4;
#pragma unroll 1
        for (int step = 0; step < steps; ++step) {
#pragma unroll 1
            for (int cv = 0; cv < 2; ++cv) {
                HRaw ra, rb;
                helper_load(ra, smem + cv * 16 * 1024, 1, hw, lane);
                __builtin_amdgcn_s_barrier();
#pragma unroll 1
                for (int u = 0; u < 8; u += 2) {
                    // while the matrix waves consume unit u (slot u % 3) the helpers fill the slot of unit u + 1; the rows of the
                    // unit after that are requested before the transform (software pipeline: no LDS round trip per barrier interval)
                    helper_load(rb, smem + cv * 16 * 1024, (u + 2) & 7, hw, lane);
                    helper_transform(ra, smem, (u + 1) % VRING, hw, lane);
                    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");      // the stores are out; (most of) the loads of the next unit may still fly
                    __builtin_amdgcn_s_barrier();
                    helper_load(ra, smem + cv * 16 * 1024, (u + 3) & 7, hw, lane);
                    helper_transform(rb, smem, (u + 2) % VRING, hw, lane);
                    if (u < 6) {
                        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                    }
                }
            }
        }
    }
    if (s == 12345.678f) sink[0] = s;
    if (cyc && tid == 0) atomicAdd(cyc, __builtin_readcyclecounter() - t_start);
}

template <int MODE, bool LW, bool LX, bool HELP>
static void run(const char* name, int windows, int steps, int wbytes, double rows_per_step) {
    std::vector<uint16_t> h(wbytes / 2);
    uint32_t x = 0x12345u;
    for (auto& v : h) {               // weights: random f16 around +-2^-4 .. 2^-3
        x = x * 1664525u + 1013904223u;
        v = (uint16_t)(((x >> 31) << 15) | ((11u + ((x >> 29) & 1u)) << 10) | ((x >> 8) & 0x3FFu));
    }
    unsigned char* dw;
    float* sink;
    unsigned long long* cyc;
    hipMalloc(&dw, wbytes);
    hipMalloc(&sink, 16);
    hipMalloc(&cyc, 8);
    hipMemcpy(dw, h.data(), wbytes, hipMemcpyHostToDevice);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e30f;
    unsigned long long hc = 0;
    for (int rep = 0; rep < 4; ++rep) {
        hipMemset(cyc, 0, 8);
        hipEventRecord(a);
        hipLaunchKernelGGL((probe_kernel<MODE, LW, LX, HELP>), dim3(windows), dim3(512), 0, 0, dw, wbytes, steps, sink, cyc);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms < best) {
            best = ms;
            hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        }
    }
    hipError_t e = hipGetLastError();
    const double cyc_step = (double)hc / windows / steps;
    printf("%-44s %8.3f ms per %d windows | %8.0f cycles per step (%3.0f rows) = %8.0f per 128 rows | conv-only ceiling %8.0f windows/s | %s\n", name, best,
           windows, cyc_step, rows_per_step, cyc_step * 128.0 / rows_per_step, windows / (best * 1e-3), hipGetErrorString(e));
    fflush(stdout);
    hipFree(dw);
    hipFree(sink);
    hipFree(cyc);
    hipEventDestroy(a);
    hipEventDestroy(b);
}

int main(int argc, char** argv) {
    const int windows = argc > 1 ? atoi(argv[1]) : 4096;
    const int WD = 2 * 48 * 8192, WT = 2 * 8 * NXI * 8192;
    const int TSTEPS = NXI == 8 ? 63 : 47;
    const double TR = 32.0 * TROWS;
    printf("conv2 + conv3 loops only (no w_v, no epilogues, no conv1 / pair products); weights %d KB direct, %d KB Toom-Cook per step\n", WD / 1024, WT / 1024);
    if (argc <= 2) {
        run<0, true, true, false>("direct: weights + LDS reads + MFMAs", windows, 47, WD, 128);
        run<0, false, true, false>("direct: no weight stream", windows, 47, WD, 128);
        run<0, false, false, false>("direct: MFMAs only", windows, 47, WD, 128);
        run<1, true, true, false>(NXI == 8 ? "tc F(3,6): weights + LDS reads + MFMAs" : "tc F(4,6): weights + LDS reads + MFMAs", windows, TSTEPS, WT, TR);
        run<1, false, true, false>(NXI == 8 ? "tc F(3,6): no weight stream" : "tc F(4,6): no weight stream", windows, TSTEPS, WT, TR);
        run<1, true, false, false>(NXI == 8 ? "tc F(3,6): no LDS reads" : "tc F(4,6): no LDS reads", windows, TSTEPS, WT, TR);
        run<1, false, false, false>(NXI == 8 ? "tc F(3,6): MFMAs only" : "tc F(4,6): MFMAs only", windows, TSTEPS, WT, TR);
    }
    run<1, true, true, true>(NXI == 8 ? "tc F(3,6): all + helper transform + barriers" : "tc F(4,6): all + helper transform + barriers", windows, TSTEPS, WT, TR);
    run<1, false, true, true>(NXI == 8 ? "tc F(3,6): helpers + barriers, no weights" : "tc F(4,6): helpers + barriers, no weights", windows, TSTEPS, WT, TR);
    return 0;
}
