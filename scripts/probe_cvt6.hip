// Element order of v_cvt_scalef32_2xpk16_fp6_f32 (two 16 x f32 sources -> 32 fp6 values): which source element lands in
// which output slot, and the direction of its scale operand.  Build: hipcc --offload-arch=gfx950 -O2 scripts/probe_cvt6.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v6i __attribute__((ext_vector_type(6)));
__global__ void k(const float* a, const float* b, float s, int* out) {
    v16f x, y;
    for (int i = 0; i < 16; ++i) x[i] = a[i], y[i] = b[i];
    const v6i r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(x, y, s);
    for (int i = 0; i < 6; ++i) out[i] = r[i];
}
static double val(int c) { const int e = (c >> 3) & 3, m = c & 7; return e == 0 ? m / 8.0 : (1 + m / 8.0) * std::ldexp(1.0, e - 1); }
int main() {
    float ha[16], hb[16];
    for (int i = 0; i < 16; ++i) ha[i] = (float)val(i), hb[i] = (float)val(16 + i);   // a[i] encodes as code i, b[i] as code 16 + i
    float *da, *db; int* dout;
    hipMalloc(&da, 64); hipMalloc(&db, 64); hipMalloc(&dout, 24);
    for (float s : {1.0f, 4.0f}) {
        float sa[16], sb[16];
        for (int i = 0; i < 16; ++i) sa[i] = ha[i] * s, sb[i] = hb[i] * s;             // inputs times s: x / s recovers the codes
        hipMemcpy(da, sa, 64, hipMemcpyHostToDevice); hipMemcpy(db, sb, 64, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, da, db, s, dout);
        unsigned w[7] = {0}; hipMemcpy(w, dout, 24, hipMemcpyDeviceToHost);
        printf("scale %g: slot -> source:", s);
        bool natural_interleave = true;
        for (int i = 0; i < 32; ++i) {
            const int bit = i * 6, kk = bit >> 5, o = bit & 31;
            unsigned long long v = w[kk] | ((unsigned long long)w[kk + 1] << 32);
            const int c = (int)((v >> o) & 63);
            printf(" %d:%s%d", i, c < 16 ? "a" : "b", c & 15);
            if (c != ((i & 1) ? 16 + (i >> 1) : (i >> 1))) natural_interleave = false;
        }
        printf("\n  -> slot 2i = a[i], slot 2i+1 = b[i]: %s\n", natural_interleave ? "YES" : "no");
    }
    return 0;
}
