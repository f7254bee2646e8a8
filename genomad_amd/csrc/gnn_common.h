// Shared declarations of libgenomad_nn_hip.so (host side + kernel launchers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/genomad_nn.h"

namespace gnn {

constexpr int W = GNN_WINDOW;
constexpr int T = GNN_TOKENS;
constexpr int C = GNN_CH;
constexpr int KS = GNN_KSIZE;
constexpr int NP = GNN_PATCHES;
constexpr int PS = GNN_PATCH_SIZE;
constexpr int NPAIR = NP * PS;            // 8400 (patch, slot) pairs per head
constexpr int POOLED = GNN_POOLED;
constexpr int FEAT = GNN_FEAT;
constexpr int HID = GNN_HIDDEN;
constexpr float LRELU = 0.1f;             // igloo.py:48
constexpr float BN_EPS = 1e-3f;           // Keras BatchNormalization default

// ---- fused front end geometry (gnn_fused_x3.hip, gnn_fused_c6.hip) ----
constexpr int FT = 128;                   // rows (token positions) per step of a workgroup
constexpr int FSTEPS = (T + FT - 1) / FT; // 47 steps per window
// rows of a conv1 pair table: 1024 five-mers | 256 (N, 4-mer) | 256 (4-mer, N) | (N, N) | (absent,
// absent) | 257 (absent, token) — see pair_row() in gnn_fused.hip
constexpr int PAIR_ROWS = 1024 + 256 + 256 + 1 + 1 + 257;   // 1795
// logits GEMM on the matrix pipe (gnn_backend.hip): K = 2100 patches padded to k-steps of 16,
// N = 749 pooled positions padded to n-blocks of 32
constexpr int QK_KSTEPS = (NP + 15) / 16;        // 132
constexpr int QK_NBLK = (POOLED + 31) / 32;      // 24

void set_error(const std::string& msg);
bool debug_switch(const char* name);   // gnn_api.hip: environment switch read once, announced on stderr when set

#define GNN_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            gnn::set_error(std::string(#call) + " failed: " + hipGetErrorString(e_));          \
            return GNN_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

// Device-resident, re-packed weights.
struct DeviceWeights {
    // f32 reference layouts
    float* conv1_k = nullptr;   // (6,257,128)
    float* conv1_b = nullptr;
    float* conv_k[2] = {nullptr, nullptr};   // conv2, conv3 (6,128,128)
    float* conv_b[2] = {nullptr, nullptr};
    // IGLOO (patch, slot) pairs, SORTED BY POSITION ("bucket order"): entry e of head h reads
    // activation row pos_sorted[e] and weights weff_sorted[e,:]; pair p*4+j lives at e = slot[p*4+j].
    // bucket_ptr[s] .. bucket_ptr[s+1] are the entries whose position falls into fused step s.
    float* weff_sorted[2] = {nullptr, nullptr};    // (8400,128)  w_mult * w_summer folded
    int32_t* pos_sorted[2] = {nullptr, nullptr};   // (8400,)
    int32_t* slot[2] = {nullptr, nullptr};         // (8400,) pair -> entry
    int32_t* bucket_ptr[2] = {nullptr, nullptr};   // (FSTEPS+1,)
    float* conv1_pairs = nullptr;  // (3, PAIR_ROWS, 128): W1[2j][a] + W1[2j+1][b] per token pair
    float* w_bias[2] = {nullptr, nullptr};   // (2100,)
    float* w_qk[2] = {nullptr, nullptr};     // (2100,749)
    float* w_v[2] = {nullptr, nullptr};      // (128,128) [in][out]
    float* d1_k = nullptr;  // (256,512) BN folded
    float* d1_b = nullptr;
    float* d2_k = nullptr;  // (512,512) BN folded
    float* d2_b = nullptr;
    float* d3_k = nullptr;  // (512,3)
    float* d3_b = nullptr;
    uint16_t* d1_frag = nullptr;   // BN-folded Dense512 kernels in MFMA fragment order, f16 hi / lo (dense_mfma_kernel):
    uint16_t* d2_frag = nullptr;   // [kstep][nblk 16][plane 2][lane 64][8]
    // fused path packs (gnn_pack.hip): MFMA fragment order, bf16 hi / lo planes
    uint16_t* conv_frag[2] = {nullptr, nullptr};  // conv2, conv3: [kstep 48][nblk 4][plane 2][lane 64][8]
    uint16_t* wv_frag[2] = {nullptr, nullptr};    // head A, B:   [kstep 8][nblk 4][plane 2][lane 64][8]
    uint16_t* wqk_frag[2] = {nullptr, nullptr};   // head A, B:   [kstep 132][nblk 24][plane 2][lane 64][8], zero padded
    uint16_t* wqk_frag_h[2] = {nullptr, nullptr}; // the same with f16 hi / lo limbs (logits GEMM of GNN_PREC_F16X3)
    uint16_t* conv_frag_h[2] = {nullptr, nullptr};   // the same fragment layouts with f16 hi / lo limbs (GNN_PREC_F16X3)
    uint16_t* wv_frag_h[2] = {nullptr, nullptr};
    // f16 + MX-fp6-correction packs (gnn_fused_c6.hip): [k32 step][nblk 4][f16 even 1 KiB | f16 odd 1 KiB | fp6 dwords 0-3 1 KiB |
    // fp6 dwords 4-5 512 B], scale words as above, pair tables in the gather's lane order (bias folded), entry ranges per step
    uint32_t* conv_c6[2] = {nullptr, nullptr};
    uint32_t* wv_c6[2] = {nullptr, nullptr};
    float* conv1_pairs6 = nullptr;
    float* weff6[2] = {nullptr, nullptr};      // folded IGLOO weights, entry pairs x [i 8][entry parity][block 4][4 ch]
    int32_t* bucket_ptr6[2] = {nullptr, nullptr};
    float* c6_yp_const = nullptr;   // (2, 749, 128) / (2, 8400): the f16c6 kernel's yp and mp of an all-N window (padding skip)
    float* c6_mp_const = nullptr;
    // Toom-Cook F(3,6) front end (gnn_fused_tc.hip): transformed conv weights [k16 unit 8][xi 8][nblk 4][hi | lo][lane 64][8] f16,
    // the power of two the inverse transform multiplies by, IGLOO entry ranges per 96-row step, all-N window outputs
    uint16_t* tc_frag[2] = {nullptr, nullptr};
    float tc_inv_s[2] = {1.f, 1.f};
    int32_t* bucket_ptr96[2] = {nullptr, nullptr};
    float* tc_yp_const = nullptr;
    float* tc_mp_const = nullptr;
    float* tc_wva_tbl = nullptr;    // head A's y @ w_v per 9-mer: gnn_fused_tc.hip, WvaTable (1.38 GB)
    // k-mer tables of GNN_PREC_F16X3TK (gnn_fused_tk.hip, gnn_build_kmer_tables): x2 per 14-mer (137.4 GB), head A's pair products per
    // (entry, 9-mer) (8.8 GB), that kernel's outputs of an all-N window.  Not in ctx->owned: gnn_drop_kmer_tables frees them
    float* tk_x2_tbl = nullptr;
    float* tk_mpa_tbl = nullptr;
    float* tk_x1t_tbl = nullptr;    // x1 over WvaTable's index space (1.38 GB): head A's entries at 9-mers MpaTable has no column for
    float* tk_pt_tbl = nullptr;     // conv2's six tap tables over WvaTable's index space (8.3 GB): the rows the 14-mer table cannot index
    float* tk_yp_const = nullptr;
    float* tk_mp_const = nullptr;
    float* x3_yp_const[2] = {nullptr, nullptr};   // the same of gnn_fused_x3.hip: [0] bf16 limbs, [1] f16 limbs
    float* x3_mp_const[2] = {nullptr, nullptr};
};

struct Workspace {
    int64_t chunk = 0;          // windows per launch
    uint16_t* tokens = nullptr; // (chunk, 5997)           f32 path only
    float* x[3] = {nullptr, nullptr, nullptr};  // (chunk,5997,128) each, f32 path only
    int64_t x_chunk = 0;        // windows the x buffers hold
    float* mp = nullptr;        // (chunk, 2, 8400) pair dot products in bucket order
    float* m = nullptr;         // (chunk, 2, 2100)  bias + the four pair products of every patch
    float* yp = nullptr;        // (chunk, 2, 749, 128)
    float* logits = nullptr;    // (chunk, 2, 749)
    float* alpha = nullptr;     // (chunk, 2, 749)   (tap)
    float* feat = nullptr;      // (chunk, 256)
};

struct ContigWorkspace;

struct ProfileSlot {
    double total_ms = 0.0;
    int64_t launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

}  // namespace gnn

struct gnn_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool has_weights = false;
    gnn::DeviceWeights w;
    gnn::Workspace ws;
    // second workspace + stream: the back end of chunk i runs beside the front end of chunk i+1 (classify_chunks)
    gnn::Workspace ws_alt;
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_front[2] = {nullptr, nullptr}, ev_back[2] = {nullptr, nullptr};
    bool back_pending[2] = {false, false};   // ev_back[i] recorded on stream2 and not yet waited for by `stream`
    int buf_cur = 0;                         // which of the two alternating workspaces `ws` currently is
    int64_t chunk_fused = 16384;     // windows per launch of a fused front end (gnn_set_chunk): larger grids amortise the launch's tail (185.9 vs
                                     // 184.6 k windows/s against 8192, profiles/r04/backend_overlap_ab.txt); 13 GB of workspace at full size, one set.
                                     // The DEFAULT is a ceiling, not a demand: classify_chunks clamps it to a quarter of the device memory that is
                                     // free when the workspace first grows, and any size (explicit or not) is halved and retried when the
                                     // allocation fails (shared / partitioned GPUs)
    int64_t alt_failed_chunk = 0;    // a second workspace of this many windows did not fit (classify_chunks does not retry at or above it)
    bool chunk_explicit = false;     // gnn_set_chunk was called: no clamp against free memory, only the halve-and-retry on failure
    int64_t chunk_f32 = 64;
    bool profile = false;
    gnn::ProfileSlot prof[GNN_K_COUNT];
    std::vector<hipEvent_t> event_pool;
    std::vector<void*> owned;   // device allocations to free at destroy
    int cu_count = 0;
    int last_split = 1;                           // workgroups per window of the last streaming-kernel launch (gnn_debug_last_split)
    bool time_split = true;                       // x3 kernel: several workgroups per window when a launch is smaller than the chip (gnn_debug_set_time_split)
    bool c6_pad_skip = true;                      // f16c6: copy the all-N tail of a window instead of computing it (gnn_debug_set_pad_skip)
    unsigned long long* phase_cycles = nullptr;   // non-null: fused kernel runs its instrumented build
    gnn::ContigWorkspace* contig_ws = nullptr;    // gnn_contigs.hip: persistent buffers of gnn_classify_contigs
    // gnn_classify / gnn_debug_forward (host windows in, host scores out): persistent, grow-only staging - a device slab for
    // the windows and their scores, two pinned bounce buffers the windows go through in pieces (the copy of piece i+1 into
    // its bounce buffer overlaps the DMA of piece i) and a pinned landing buffer for the scores.  No allocation per call.
    uint8_t* stage_bases = nullptr;
    float* stage_scores = nullptr;
    float* stage_scores_host = nullptr;
    int64_t stage_windows = 0;
    void* pin[2] = {nullptr, nullptr};
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    bool pin_busy[2] = {false, false};
    // classify_chunks: staging of a window buffer that is not 4-byte aligned (the streaming kernels fetch bases as aligned dwords)
    uint8_t* align_buf = nullptr;
    int64_t align_windows = 0;
    // RCCL communicator of this ctx (gnn_comm.hip); ncclComm_t kept opaque here
    void* comm = nullptr;
    int comm_ranks = 1, comm_rank = 0;
    void* comm_scratch = nullptr;
    size_t comm_scratch_bytes = 0;
};

namespace gnn {

// ---- kernel launchers (each enqueues on ctx->stream, returns gnn_status) ----
int launch_tokenize(gnn_ctx* ctx, const uint8_t* bases, int64_t n, uint16_t* tokens);
int launch_onehot(gnn_ctx* ctx, const uint8_t* bases, int64_t n, int dtype, void* out);
int launch_synth(gnn_ctx* ctx, uint64_t seed, int64_t first, int64_t n, uint8_t* bases);
int launch_span_count(gnn_ctx* ctx, const uint8_t* seq, const int64_t* starts, const int32_t* lens, int64_t n,
                      int byte, int32_t* counts);
int launch_materialize(gnn_ctx* ctx, const uint8_t* seq, const int64_t* starts, const int32_t* lens, int64_t n,
                       uint8_t* bases);
int launch_front_f32(gnn_ctx* ctx, const uint8_t* bases, int64_t n);         // -> ws.mp, ws.yp (+ ws.x)
int launch_backend(gnn_ctx* ctx, int64_t n, int precision, float* scores_dev);   // ws.mp, ws.yp -> scores
// one pass of the hot path over n windows whose padded bases are on the device (gnn_api.hip)
// defer_last: leave the last chunk's back end pending on the second stream (gnn_classify_dev_async).  flush_backend()
// makes ctx->stream wait for whatever is pending (the end of a synchronous classify_chunks); finish_pending() waits for it on
// the host - every other entry point that enqueues on ctx->stream, reads scores or touches the workspaces calls it first
int classify_chunks(gnn_ctx* ctx, const uint8_t* bases_dev, int64_t n, int precision, float* scores_dev, bool defer_last = false);
int flush_backend(gnn_ctx* ctx);
int finish_pending(gnn_ctx* ctx);
void free_contig_ws(gnn_ctx* ctx);     // gnn_contigs.hip
void free_stage(gnn_ctx* ctx);         // gnn_api.hip: staging of the host-buffer entry points

int launch_front_c6(gnn_ctx* ctx, const uint8_t* bases, int64_t n);             // GNN_PREC_F16C6 -> ws.mp, ws.yp
int launch_front_x3(gnn_ctx* ctx, const uint8_t* bases, int64_t n, int precision);   // GNN_PREC_F16X3 / BF16X3 (gnn_fused_x3.hip) -> ws.mp, ws.yp
int launch_front_tc(gnn_ctx* ctx, const uint8_t* bases, int64_t n);             // GNN_PREC_F16X3TC (gnn_fused_tc.hip) -> ws.mp, ws.yp
int launch_front_tk(gnn_ctx* ctx, const uint8_t* bases, int64_t n);             // GNN_PREC_F16X3TK (gnn_fused_tk.hip) -> ws.mp, ws.yp
int build_kmer_tables(gnn_ctx* ctx, size_t reserve);                             // gnn_fused_tk.hip: GNN_ERR_NOMEM when free memory < tables + reserve
void free_kmer_tables(gnn_ctx* ctx);
size_t kmer_tables_bytes();
int build_wva_rows_table(gnn_ctx* ctx, const float* w_kc, float* tbl);          // gnn_fused_tc.hip: x1(row) @ w for every row of WvaTable's index space
int pack_fused_tc_weights(gnn_ctx* ctx, const gnn_weights* w);                   // after pack_fused_c6_weights (shares its pair tables)
int pack_fused_x3_consts(gnn_ctx* ctx);                                          // all-N window outputs of that kernel (after the other packs)

// host-side packing for the fused paths (gnn_pack.hip)
int pack_fused_weights(gnn_ctx* ctx, const gnn_weights* w);
// K x N row-major f32 -> [kstep][nblk][plane hi, lo][lane 64][8 x 16 bit] (bf16 or f16 limbs), zero padded (gnn_pack.hip)
std::vector<uint16_t> pack_frags(const float* wmat, int K, int N, bool f16 = false);
int pack_fused_c6_weights(gnn_ctx* ctx, const gnn_weights* w);
int c6_rows_per_step();
int c6_pack_matrix(const float* wmat, int K, int N, std::vector<uint32_t>& out);   // host only (tests)
// conv1 pair tables (3, PAIR_ROWS, 128): W1[2j][a] + W1[2j+1][b] for every token pair of adjacent positions (gnn_api.hip)
void build_conv1_pair_tables(const float* conv1_kernel, std::vector<float>& pt);

}  // namespace gnn
