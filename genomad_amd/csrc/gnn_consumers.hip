// Downstream score consumers as a device epilogue.  They are the immediate
// readers of the nn-classification NPZ in `genomad end-to-end`:
//   branch_attention         genomad/modules/aggregated_classification.py:10-34   (constants inline there)
//   score_batch_correction   genomad/modules/score_calibration.py:15-43           (6->20->20->3 tanh MLP)
// Both are float64 numpy in the reference and O(contigs) work, so they are written as one thread per
// contig in f64 and are bit-comparable (to libm rounding) with the reference functions, which — unlike
// the TensorFlow half — can be executed in this container: tests/golden/consumers_golden.npz holds
// their outputs.
#include "gnn_common.h"

namespace gnn {

__device__ __forceinline__ void softmax3(const double* x, double temperature, double* out) {
    const double a = x[0] / temperature, b = x[1] / temperature, c = x[2] / temperature;   // utils.softmax
    const double m = fmax(a, fmax(b, c));
    const double ea = exp(a - m), eb = exp(b - m), ec = exp(c - m);
    const double s = ea + eb + ec;
    out[0] = ea / s;
    out[1] = eb / s;
    out[2] = ec / s;
}

// aggregated_classification.py:16-34
__global__ void branch_attention_kernel(const double* __restrict__ w, const double* __restrict__ b1,
                                        const double* __restrict__ b2, int64_t n, double temperature,
                                        double* __restrict__ out) {
    const double w_1[6] = {0.3598502, 2.912244, -1.0668367, 1.3729712, -2.1972055, 0.9363847};
    const double w_2[6] = {1.5372132, 2.6216774, -2.8225133, 3.0680428, 2.803005, -1.1982375};
    const double dense[3][3] = {{1.6666023, -1.1003100, -2.1425622},
                                {-2.2625937, 2.7540822, -1.5622343},
                                {1.9745151, 1.0952991, -2.7467837}};
    const double bias[3] = {0.14732242, -0.6838019, 0.5594167};
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double h[3];
    for (int k = 0; k < 3; ++k) {
        const double a1 = w[i] * w_1[k] + w_2[k], a2 = w[i] * w_1[3 + k] + w_2[3 + k];
        h[k] = (b1[i * 3 + k] * a1 + b2[i * 3 + k] * a2) / 2;
    }
    double o[3];
    for (int j = 0; j < 3; ++j) {
        double s = 0.0;                                    // np.matmul order: sum over k ascending
        for (int k = 0; k < 3; ++k) s += h[k] * dense[k][j];
        o[j] = s + bias[j];
    }
    softmax3(o, temperature, out + i * 3);
}

// score_calibration.py:37-43; `comp` is the already smoothed composition (:18-21, computed on the host)
__global__ void calibration_kernel(const double* __restrict__ scores, const double* __restrict__ comp,
                                   const double* __restrict__ k1, const double* __restrict__ c1,
                                   const double* __restrict__ k2, const double* __restrict__ c2,
                                   const double* __restrict__ k3, const double* __restrict__ c3, int64_t n,
                                   double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x[6] = {comp[0], comp[1], comp[2], scores[i * 3], scores[i * 3 + 1], scores[i * 3 + 2]};
    double h1[20], h2[20], o[3];
    for (int j = 0; j < 20; ++j) {
        double s = 0.0;
        for (int k = 0; k < 6; ++k) s += x[k] * k1[k * 20 + j];
        h1[j] = tanh(s + c1[j]);
    }
    for (int j = 0; j < 20; ++j) {
        double s = 0.0;
        for (int k = 0; k < 20; ++k) s += h1[k] * k2[k * 20 + j];
        h2[j] = tanh(s + c2[j]);
    }
    for (int j = 0; j < 3; ++j) {
        double s = 0.0;
        for (int k = 0; k < 20; ++k) s += h2[k] * k3[k * 3 + j];
        o[j] = s + c3[j];
    }
    softmax3(o, 1.0, out + i * 3);
}

// host arrays -> device, run, copy back; `ins` are (pointer, count) pairs of doubles
template <typename Launch>
static int run_f64(gnn_ctx* ctx, std::initializer_list<std::pair<const double*, size_t>> ins, int64_t n,
                   double* out_host, Launch launch) {
    GNN_HIP(hipSetDevice(ctx->device));
    if (int frc = finish_pending(ctx)) return frc;
    std::vector<double*> dev;
    int rc = GNN_OK;
    auto fail = [&](hipError_t e, const char* what) {
        if (rc == GNN_OK && e != hipSuccess) {
            set_error(std::string(what) + " failed: " + hipGetErrorString(e));
            rc = GNN_ERR_HIP;
        }
    };
    for (auto& in : ins) {
        double* p = nullptr;
        fail(hipMalloc((void**)&p, std::max<size_t>(in.second, 1) * sizeof(double)), "hipMalloc");
        dev.push_back(p);
        if (rc == GNN_OK && in.second)
            fail(hipMemcpyAsync(p, in.first, in.second * sizeof(double), hipMemcpyHostToDevice, ctx->stream), "copy in");
    }
    double* dout = nullptr;
    fail(hipMalloc((void**)&dout, (size_t)n * 3 * sizeof(double)), "hipMalloc");
    if (rc == GNN_OK) {
        launch(dev, dout);
        fail(hipGetLastError(), "launch");
        fail(hipMemcpyAsync(out_host, dout, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream), "copy out");
        fail(hipStreamSynchronize(ctx->stream), "sync");
    }
    for (double* p : dev)
        if (p) (void)hipFree(p);
    if (dout) (void)hipFree(dout);
    return rc;
}

}  // namespace gnn

using namespace gnn;

extern "C" int gnn_branch_attention(gnn_ctx* ctx, const double* w_host, const double* b1_host, const double* b2_host,
                                    int64_t n, double temperature, double* out_host) {
    if (!ctx || n < 0 || temperature == 0.0 || (n > 0 && (!w_host || !b1_host || !b2_host || !out_host))) {
        set_error("bad argument to gnn_branch_attention");
        return GNN_ERR_ARG;
    }
    if (n == 0) return GNN_OK;
    return run_f64(ctx, {{w_host, (size_t)n}, {b1_host, (size_t)n * 3}, {b2_host, (size_t)n * 3}}, n, out_host,
                   [&](std::vector<double*>& d, double* out) {
                       hipLaunchKernelGGL(branch_attention_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                                          ctx->stream, d[0], d[1], d[2], n, temperature, out);
                   });
}

extern "C" int gnn_score_calibration(gnn_ctx* ctx, const double* scores_host, const double* composition3,
                                     const double* kernel1, const double* bias1, const double* kernel2,
                                     const double* bias2, const double* kernel3, const double* bias3, int64_t n,
                                     double* out_host) {
    if (!ctx || n < 0 || !composition3 || !kernel1 || !bias1 || !kernel2 || !bias2 || !kernel3 || !bias3 ||
        (n > 0 && (!scores_host || !out_host))) {
        set_error("bad argument to gnn_score_calibration");
        return GNN_ERR_ARG;
    }
    if (n == 0) return GNN_OK;
    return run_f64(ctx,
                   {{scores_host, (size_t)n * 3}, {composition3, 3}, {kernel1, 120}, {bias1, 20}, {kernel2, 400},
                    {bias2, 20}, {kernel3, 60}, {bias3, 3}},
                   n, out_host, [&](std::vector<double*>& d, double* out) {
                       hipLaunchKernelGGL(calibration_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                                          ctx->stream, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], n, out);
                   });
}
