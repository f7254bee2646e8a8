// MFMA ceiling probe (measurement aid, not on the hot path): what does this chip sustain on
// v_mfma_f32_32x32x16_bf16 when nothing but the matrix pipe is busy?  The chip is power-managed, so
// the sustained figure on random operands is well below the 2.5 PFLOP/s datasheet peak; bench.py
// reports it next to the fused kernel's issued-MFMA rate.
//
// Geometry mirrors the fused kernel's matrix waves: one wave per SIMD (4 per CU, 256 workgroups per
// round), 4 independent 32x32 accumulators per wave, operands in registers (loaded once from a
// random buffer so nothing constant-folds).
#include "gnn_common.h"

namespace gnn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef _Float16 f16x8p __attribute__((ext_vector_type(8)));

// KIND 0: bf16 MFMAs (the round-1 probe); KIND 1: f16 MFMAs - the instruction of the default arithmetic (f16x3), same
// geometry: 12 MFMAs per iteration on 4 accumulators, operands resident in registers
template <int KIND>
__global__ __launch_bounds__(256, 1) void mfma_probe_kernel(const uint4* __restrict__ operands, int iters,
                                                            float* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    uint4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = operands[(i * 2 + 0) * 64 + lane];
        b[i] = operands[(i * 2 + 1) * 64 + lane];
    }
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 3; ++rep)          // 12 MFMAs per iteration, like one k-step of the fused kernel
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if constexpr (KIND == 1)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8p, a[(i + rep) & 3]), __builtin_bit_cast(f16x8p, b[i]), acc[i], 0, 0, 0);
                else
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(i + rep) & 3]), __builtin_bit_cast(bf16x8, b[i]), acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[0] = s;               // keeps the accumulators live
}

// KIND 2: the MFMA mix of the f16c6 arithmetic - per iteration the 8 f16 MFMAs and the 4 MX-fp6 (e2m3) scaled MFMAs of one k32
// step over 4 row blocks (gnn_fused_c6.hip), operands resident in registers.  The rate is reported in ALGORITHMIC flops: the 8
// f16 MFMAs compute the step's products once, the fp6 ones are the correction (0.5 pass equivalents of cost).
typedef int i32x8p __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256, 1) void mfma_probe_c6_kernel(const uint4* __restrict__ operands, int iters, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    uint4 a[4], b[2];
    i32x8p xa[4], xb;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = operands[(i * 2 + 0) * 64 + lane];
        const uint4 q = operands[(i * 2 + 1) * 64 + lane];
        xa[i] = i32x8p{(int)q.x, (int)q.y, (int)q.z, (int)q.w, (int)a[i].x, (int)a[i].y, 0, 0};
    }
    b[0] = operands[1 * 64 + lane];
    b[1] = operands[3 * 64 + lane];
    xb = i32x8p{(int)b[0].x, (int)b[1].y, (int)b[0].z, (int)b[1].w, (int)b[1].x, (int)b[0].y, 0, 0};
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int sc = 0x7F7F7F7F;                       // E8M0 scale 2^0 in every byte
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8p, b[h]), __builtin_bit_cast(f16x8p, a[(i + h) & 3]), acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xb, xa[i], acc[i], 2, 2, 0, sc, 0, sc);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[0] = s;
}

}  // namespace gnn

using namespace gnn;

// Runs the probe for roughly `ms_target` milliseconds (after a calibration launch) and returns the
// sustained dense MFMA rate in TFLOP/s: kind 0 = bf16 (gnn_mfma_probe), kind 1 = f16, kind 2 = the f16 + MX-fp6 mix of f16c6
// (algorithmic TFLOP/s: the f16 MFMAs only).
extern "C" int gnn_mfma_probe(gnn_ctx* ctx, int ms_target, double* tflops_out) { return gnn_mfma_probe_kind(ctx, 0, ms_target, tflops_out); }

extern "C" int gnn_mfma_probe_kind(gnn_ctx* ctx, int kind, int ms_target, double* tflops_out) {
    if (!ctx || !tflops_out || ms_target < 1 || kind < 0 || kind > 2) {
        set_error("bad argument to gnn_mfma_probe");
        return GNN_ERR_ARG;
    }
    GNN_HIP(hipSetDevice(ctx->device));
    if (int frc = finish_pending(ctx)) return frc;
    const int blocks = ctx->cu_count > 0 ? ctx->cu_count : 256;
    std::vector<uint16_t> host(8 * 64 * 8);
    uint32_t x = 0x12345u;
    for (auto& v : host) {                          // random values in roughly [-2, 2) in the probe's 16-bit format
        x = x * 1664525u + 1013904223u;
        if (kind >= 1) {                            // f16: 5 exponent bits (bias 15), 10 mantissa bits (kind 2 reads the same words as fp6 data too)
            const uint32_t sign = (x >> 31) << 15, exp = 14u + ((x >> 29) & 1u), man = (x >> 8) & 0x3FFu;
            v = (uint16_t)(sign | (exp << 10) | man);
        } else {
            const uint32_t sign = (x >> 31) << 15, exp = 126u + ((x >> 29) & 1u), man = (x >> 8) & 0x7Fu;
            v = (uint16_t)(sign | (exp << 7) | man);
        }
    }
    void *dop = nullptr, *dsink = nullptr;
    GNN_HIP(hipMalloc(&dop, host.size() * 2));
    GNN_HIP(hipMalloc(&dsink, 16));
    GNN_HIP(hipMemcpy(dop, host.data(), host.size() * 2, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    GNN_HIP(hipEventCreate(&e0));
    GNN_HIP(hipEventCreate(&e1));
    auto run = [&](int iters, float* ms) -> int {
        GNN_HIP(hipEventRecord(e0, ctx->stream));
        if (kind == 2)
            hipLaunchKernelGGL(mfma_probe_c6_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const uint4*)dop, iters, (float*)dsink);
        else if (kind == 1)
            hipLaunchKernelGGL(mfma_probe_kernel<1>, dim3(blocks), dim3(256), 0, ctx->stream, (const uint4*)dop, iters, (float*)dsink);
        else
            hipLaunchKernelGGL(mfma_probe_kernel<0>, dim3(blocks), dim3(256), 0, ctx->stream, (const uint4*)dop, iters, (float*)dsink);
        GNN_HIP(hipEventRecord(e1, ctx->stream));
        GNN_HIP(hipEventSynchronize(e1));
        GNN_HIP(hipEventElapsedTime(ms, e0, e1));
        return GNN_OK;
    };
    float ms = 0.f;
    int rc = run(20000, &ms);                       // calibration (also warms the clocks)
    int iters = 20000;
    if (!rc) {
        iters = (int)std::min<double>(2.0e9, std::max(20000.0, 20000.0 * ms_target / std::max(ms, 1e-3f)));
        rc = run(iters, &ms);
    }
    // kind 2: only the 8 f16 MFMAs of an iteration are algorithmic flops
    if (!rc) *tflops_out = (double)blocks * 4 /*waves*/ * iters * (kind == 2 ? 8.0 : 12.0) * 32768.0 / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(dop);
    (void)hipFree(dsink);
    return rc;
}
