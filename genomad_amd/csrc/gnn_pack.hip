// Host-side packing of weight matrices into MFMA fragment order (bf16 and f16 hi / lo limbs) for the streaming front ends
// (gnn_fused_x3.hip, gnn_fused_tc.hip), the logits GEMM and the dense stack (gnn_backend.hip).  Until round 5 this lived beside the round-1
// fused kernel (gnn_fused.hip: single-pass bf16 and the byte-load server of misaligned buffers), which round 6 removed together with the
// experimental f16c8 kernel (gnn_fused_c8.hip): GNN_PREC_BF16 and GNN_PREC_F16C8 keep their enum values and answer GNN_ERR_STATE, a
// window buffer that is not 4-byte aligned goes through one aligned staging copy in classify_chunks and then through the default kernel.
#include <cstring>

#include "gnn_common.h"

namespace gnn {

static inline uint16_t bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// Wmat (K x N, row major) -> [kstep Kpad/16][nblk Npad/32][plane hi,lo][lane 64][8] bf16 in the operand
// layout of v_mfma_f32_32x32x16_bf16: lane l holds W[kstep*16 + (l>>5)*8 + e][nblk*32 + (l&31)];
// rows >= K and columns >= N are zero.
static uint16_t f16_bits(float f) {
    const _Float16 h = (_Float16)f;          // round to nearest even
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}
static float f16_value(uint16_t u) {
    _Float16 h;
    memcpy(&h, &u, 2);
    return (float)h;
}

std::vector<uint16_t> pack_frags(const float* wmat, int K, int N, bool f16) {
    const int ksteps = (K + 15) / 16, nblks = (N + 31) / 32;
    std::vector<uint16_t> out((size_t)ksteps * nblks * 2 * 64 * 8, 0);
    for (int ks = 0; ks < ksteps; ++ks)
        for (int nb = 0; nb < nblks; ++nb)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    const int k = ks * 16 + (l >> 5) * 8 + e, n = nb * 32 + (l & 31);
                    if (k >= K || n >= N) continue;
                    const float v = wmat[(size_t)k * N + n];
                    const uint16_t hi = f16 ? f16_bits(v) : bf16_rne(v);
                    const uint16_t lo = f16 ? f16_bits(v - f16_value(hi)) : bf16_rne(v - bf16_to_f32(hi));
                    const size_t base = ((size_t)(ks * nblks + nb) * 2) * 64 * 8;
                    out[base + (size_t)l * 8 + e] = hi;
                    out[base + 64 * 8 + (size_t)l * 8 + e] = lo;
                }
    return out;
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

int pack_fused_weights(gnn_ctx* ctx, const gnn_weights* w) {
    DeviceWeights& d = ctx->w;
    int rc;
    const float* ck[2] = {w->conv2_kernel, w->conv3_kernel};
    const gnn_igloo_weights* ig[2] = {&w->igloo_a, &w->igloo_b};
    for (int i = 0; i < 2; ++i) {
        if ((rc = upload_vec(ctx, pack_frags(ck[i], KS * C, C), &d.conv_frag[i]))) return rc;
        if ((rc = upload_vec(ctx, pack_frags(ig[i]->w_v, C, C), &d.wv_frag[i]))) return rc;
        if ((rc = upload_vec(ctx, pack_frags(ig[i]->w_qk, NP, POOLED), &d.wqk_frag[i]))) return rc;
        if ((rc = upload_vec(ctx, pack_frags(ig[i]->w_qk, NP, POOLED, true), &d.wqk_frag_h[i]))) return rc;
        if ((rc = upload_vec(ctx, pack_frags(ck[i], KS * C, C, true), &d.conv_frag_h[i]))) return rc;
        if ((rc = upload_vec(ctx, pack_frags(ig[i]->w_v, C, C, true), &d.wv_frag_h[i]))) return rc;
    }
    return GNN_OK;
}

}  // namespace gnn

// kept for ABI compatibility (include/genomad_nn.h): no build of this library carries an experimental kernel any more
extern "C" int gnn_has_experimental(void) { return 0; }
