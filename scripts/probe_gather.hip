// Pricing probe for "conv2 as a 9-mer table lookup" (DESIGN.md section 8): x1[t] depends on 9 bases, so
// x1(9-mer) @ W2[k] could be tabulated (6 taps x 262 144 nine-mers x 128 channels) and conv2 would become 6 row gathers +
// adds per position instead of a K = 768 contraction.  This measures what the memory system delivers for exactly that access
// pattern: every wave sums 6 rows (one per tap table) at random 9-mer indices, 64 lanes x 8 B (f32 table: 2 lanes-rounds of
// 256 B ... here a row is ROW_BYTES bytes read as 16-B pieces by ROW_BYTES/16 lanes), one output row per position.
// Build: hipcc --offload-arch=gfx950 -O3 -o build_variants/probe_gather scripts/probe_gather.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

template <int ROW_BYTES>
__global__ __launch_bounds__(256) void gather_kernel(const uint4* __restrict__ table, const uint32_t* __restrict__ idx, int positions,
                                                    uint4* __restrict__ out) {
    constexpr int LANES = ROW_BYTES / 16;            // lanes per row
    constexpr int ROWS_PER_BLOCK = 256 / LANES;
    const int sub = threadIdx.x % LANES, r = threadIdx.x / LANES;
    if (r >= ROWS_PER_BLOCK) return;
    for (int p = blockIdx.x * ROWS_PER_BLOCK + r; p < positions; p += gridDim.x * ROWS_PER_BLOCK) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const uint32_t row = idx[p + k];             // 9-mer of position p + k - 5 (sliding: consecutive positions share 8 bases)
            const uint4 v = table[((size_t)k * 262144 + row) * LANES + sub];
            acc.x += __uint_as_float(v.x); acc.y += __uint_as_float(v.y); acc.z += __uint_as_float(v.z); acc.w += __uint_as_float(v.w);
        }
        out[(size_t)(p % 65536) * LANES + sub] = make_uint4(__float_as_uint(acc.x), __float_as_uint(acc.y), __float_as_uint(acc.z), __float_as_uint(acc.w));
    }
}

template <int ROW_BYTES>
static void run(const char* name, int positions) {
    const size_t table_bytes = (size_t)6 * 262144 * ROW_BYTES;
    uint4* table; uint32_t* idx; uint4* out;
    hipMalloc(&table, table_bytes); hipMalloc(&idx, (size_t)(positions + 8) * 4); hipMalloc(&out, (size_t)65536 * ROW_BYTES);
    hipMemset(table, 0, table_bytes);
    std::vector<uint32_t> h(positions + 8);
    uint32_t code = 12345u; uint64_t s = 88172645463325252ull;
    for (int i = 0; i < positions + 8; ++i) {           // a random base stream: the 9-mer index slides by one base per position
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        code = ((code << 2) | (uint32_t)(s & 3)) & 262143u;
        h[i] = code;
    }
    hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(gather_kernel<ROW_BYTES>, dim3(256 * 8), dim3(256), 0, 0, table, idx, positions, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double bytes = (double)positions * 6 * ROW_BYTES;
        if (rep == 2)
            printf("%s: table %.0f MB, %d positions x 6 rows of %d B: %.3f ms = %.2f TB/s gathered = %.0f windows/s (5997 positions each)\n",
                   name, table_bytes / 1e6, positions, ROW_BYTES, ms, bytes / (ms * 1e-3) / 1e12, positions / 5997.0 / (ms * 1e-3));
    }
    hipFree(table); hipFree(idx); hipFree(out);
}

// probe_gather loop <seconds>: the f32-row gather back to back for about that long (to run BESIDE another process's kernels),
// one line per second with the rate sustained
static void loop_f32(double seconds) {
    constexpr int ROW_BYTES = 512;
    const int positions = 4096 * 5997 / 4;
    const size_t table_bytes = (size_t)6 * 262144 * ROW_BYTES;
    uint4* table; uint32_t* idx; uint4* out;
    hipMalloc(&table, table_bytes); hipMalloc(&idx, (size_t)(positions + 8) * 4); hipMalloc(&out, (size_t)65536 * ROW_BYTES);
    hipMemset(table, 0, table_bytes);
    std::vector<uint32_t> h(positions + 8);
    uint32_t code = 12345u; uint64_t s = 88172645463325252ull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; code = ((code << 2) | (uint32_t)(s & 3)) & 262143u; v = code; }
    hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    double total = 0;
    while (total < seconds) {
        hipEventRecord(a);
        int n = 0;
        for (; n < 100; ++n) hipLaunchKernelGGL(gather_kernel<ROW_BYTES>, dim3(256 * 8), dim3(256), 0, 0, table, idx, positions, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        total += ms * 1e-3;
        printf("gather beside: %.2f TB/s (%.3f ms per quarter launch)\n", (double)positions * 6 * ROW_BYTES * n / (ms * 1e-3) / 1e12, ms / n);
        fflush(stdout);
    }
}

int main(int argc, char** argv) {
    if (argc > 2 && std::string(argv[1]) == "loop") { loop_f32(atof(argv[2])); return 0; }
    const int positions = 4096 * 5997 / 4;               // a quarter of a 4096-window launch
    run<512>("f32 rows (805 MB)", positions);
    run<384>("f16 hi + fp8 lo rows (604 MB)", positions);
    run<256>("f16 rows (403 MB, single-pass-f16 accuracy: outside the tolerance)", positions);
    return 0;
}
