// Pricing probe for "conv2 as a k-mer table in HBM" (round 6): x2[t] = LeakyReLU(conv2(x1))[t] depends on x1[t-5 .. t], i.e. on the
// bases t-10 .. t+3: a 14-mer.  4^14 rows x 128 f32 = 137 GB - inside one MI355X's 288 GB.  conv2 (43 % of the window's FLOPs)
// would become ONE 512-byte row gather per position.  What the probe measures: the rate at which the memory system delivers
// 512-byte rows at the indices of a sliding k-mer of a random base stream (consecutive positions share k-1 bases but their rows
// are far apart), as a function of the table's size - the translation reach (TLB) and HBM's random-access rate are the unknowns.
// Also the two-half variant: two tables of 11-mers (2 GB each), two rows added per position.
// Build: hipcc --offload-arch=gfx950 -O3 -o build_variants/probe_gather_big scripts/probe_gather_big.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// one row = 512 B = 32 lanes x 16 B; a 256-thread block handles 8 rows per pass, ROWS_IN_FLIGHT passes unrolled
template <int TABLES, int INFLIGHT>
__global__ __launch_bounds__(256) void gather_kernel(const uint4* __restrict__ table, const uint32_t* __restrict__ codes, long positions,
                                                    int k, long rows_per_table, uint4* __restrict__ out) {
    const int sub = threadIdx.x & 31, r = threadIdx.x >> 5;
    const unsigned long mask = (unsigned long)rows_per_table - 1;
    const long per_block = (positions + gridDim.x - 1) / gridDim.x;     // a block walks a CONTIGUOUS stretch of the stream, like a workgroup walks its window
    const long p0 = (long)blockIdx.x * per_block, p1 = min(positions, p0 + per_block);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long p = p0 + r * INFLIGHT; p < p1; p += 8 * INFLIGHT) {
        uint4 v[INFLIGHT][TABLES];
#pragma unroll
        for (int i = 0; i < INFLIGHT; ++i) {
            const unsigned long code = codes[p + i] >> (32 - 2 * (k + (TABLES - 1) * 3));   // the 16 bases from p + i on, 2 bits each, first base on top
#pragma unroll
            for (int tb = 0; tb < TABLES; ++tb) {
                const unsigned long row = (code >> (2 * 3 * (TABLES - 1 - tb))) & mask;
                v[i][tb] = table[((size_t)tb * rows_per_table + row) * 32 + sub];
            }
        }
#pragma unroll
        for (int i = 0; i < INFLIGHT; ++i)
#pragma unroll
            for (int tb = 0; tb < TABLES; ++tb) {
                acc.x += __uint_as_float(v[i][tb].x); acc.y += __uint_as_float(v[i][tb].y);
                acc.z += __uint_as_float(v[i][tb].z); acc.w += __uint_as_float(v[i][tb].w);
            }
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = make_uint4(__float_as_uint(acc.x), __float_as_uint(acc.y), __float_as_uint(acc.z), __float_as_uint(acc.w));
}

// the access form of the product kernels: one row = 64 lanes x 8 B (gnn_fused_tk.hip: x2_rows_issue, gnn_tc_dev.h: wva_issue)
template <int INFLIGHT>
__global__ __launch_bounds__(256) void gather_b64_kernel(const uint2* __restrict__ table, const uint32_t* __restrict__ codes, long positions,
                                                        int k, long rows, uint2* __restrict__ out) {
    const int sub = threadIdx.x & 63, r = threadIdx.x >> 6;
    const unsigned long mask = (unsigned long)rows - 1;
    const long per_block = (positions + gridDim.x - 1) / gridDim.x;
    const long p0 = (long)blockIdx.x * per_block, p1 = min(positions, p0 + per_block);
    float2 acc = make_float2(0.f, 0.f);
    for (long p = p0 + r * INFLIGHT; p < p1; p += 4 * INFLIGHT) {
        uint2 v[INFLIGHT];
#pragma unroll
        for (int i = 0; i < INFLIGHT; ++i) {
            const unsigned long row = (codes[p + i] >> (32 - 2 * k)) & mask;
            v[i] = table[(size_t)row * 64 + sub];
        }
#pragma unroll
        for (int i = 0; i < INFLIGHT; ++i) {
            acc.x += __uint_as_float(v[i].x);
            acc.y += __uint_as_float(v[i].y);
        }
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = make_uint2(__float_as_uint(acc.x), __float_as_uint(acc.y));
}

// probe_gather_big calib <k>: ONE launch of each access form over a 4^k-row table, for a rocprofv3 --pmc FETCH_SIZE pass: the counter
// against the known byte count (positions x 512) calibrates it for THIS access pattern (MI355X_MICROARCH.md, HBM section)
static void calib(int k, long positions, const uint32_t* codes) {
    const long rows = 1L << (2 * k);
    uint4* table; uint4* out;
    CK(hipMalloc(&table, (size_t)rows * 512));
    CK(hipMalloc(&out, (size_t)2048 * 256 * 16));
    CK(hipMemset(table, 0, (size_t)rows * 512));
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((gather_kernel<1, 4>), dim3(2048), dim3(256), 0, 0, table, codes, positions, k, rows, out);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((gather_b64_kernel<8>), dim3(2048), dim3(256), 0, 0, reinterpret_cast<const uint2*>(table), codes, positions, k, rows,
                       reinterpret_cast<uint2*>(out));
    CK(hipDeviceSynchronize());
    printf("calib: %d-mers (%.2f GB): each kernel gathered %ld rows x 512 B = %.0f KiB\n", k, rows * 512 / 1e9, positions, positions * 512.0 / 1024);
    CK(hipFree(table)); CK(hipFree(out));
}

template <int TABLES, int INFLIGHT>
static void run(int k, long positions, const uint32_t* bases, int blocks) {
    const long rows = 1L << (2 * k);
    const size_t table_bytes = (size_t)TABLES * rows * 512;
    uint4* table; uint4* out;
    hipError_t e = hipMalloc(&table, table_bytes);
    if (e != hipSuccess) { printf("k=%d: hipMalloc of %.1f GB failed: %s\n", k, table_bytes / 1e9, hipGetErrorString(e)); (void)hipGetLastError(); return; }
    CK(hipMalloc(&out, (size_t)blocks * 256 * 16));
    CK(hipMemset(table, 0, table_bytes));
    CK(hipDeviceSynchronize());
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((gather_kernel<TABLES, INFLIGHT>), dim3(blocks), dim3(256), 0, 0, table, bases, positions, k, rows, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep > 0 && ms < best) best = ms;
    }
    printf("%d table(s) of %d-mers, %.2f GB, %d blocks x %d rows in flight per 32 lanes: %ld positions in %.3f ms = %.2f G rows/s = %.2f TB/s = %.0f windows/s at 5997 positions\n",
           TABLES, k, table_bytes / 1e9, blocks, INFLIGHT, positions, best, positions * (double)TABLES / (best * 1e-3) / 1e9,
           positions * 512.0 * TABLES / (best * 1e-3) / 1e12, positions / 5997.0 / (best * 1e-3));
    fflush(stdout);
    CK(hipFree(table)); CK(hipFree(out));
}

int main(int argc, char** argv) {
    const long positions = 16384L * 5997;                                 // one 16 384-window launch
    std::vector<uint32_t> h(positions + 64);
    uint64_t s = 88172645463325252ull; uint32_t code = 0;
    for (long i = 0; i < (long)h.size() + 15; ++i) {                      // h[p] = bases p .. p+15 packed, base p in the top two bits
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; code = (code << 2) | (uint32_t)((s >> 20) & 3);
        if (i >= 15) h[i - 15] = code;
    }
    uint32_t* bases; CK(hipMalloc(&bases, h.size() * 4)); CK(hipMemcpy(bases, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    size_t fr, tot; CK(hipMemGetInfo(&fr, &tot)); printf("device memory: %.1f GB free of %.1f GB\n", fr / 1e9, tot / 1e9);
    if (argc > 2 && std::string(argv[1]) == "calib") { calib(atoi(argv[2]), positions, bases); return 0; }
    const int kmax = argc > 1 ? atoi(argv[1]) : 14;
    for (int k = 9; k <= kmax; ++k) {
        run<1, 4>(k, positions, bases, 256 * 8);
        if (k >= 13) { run<1, 8>(k, positions, bases, 256 * 8); run<1, 4>(k, positions, bases, 256); run<1, 8>(k, positions, bases, 256 * 16); }
    }
    run<2, 4>(11, positions, bases, 256 * 8);
    run<2, 4>(12, positions, bases, 256 * 8);
    return 0;
}
