/*
 * genomad_nn.h — C ABI of libgenomad_nn_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the nn-classification hot path of geNomad.  The reference
 * has no FFI of its own: the seam is the Python function
 *   genomad/modules/nn_classification.py:21-30  main(input_path, output_path, ...)
 * and, inside it, the TensorFlow/numba calls listed per entry point below.  A
 * maintainer binds these functions with ctypes; all arguments
 * are plain pointers and sizes, no torch/numpy types.
 *
 * Conventions: every function returns 0 on success or a negative gnn_status; the
 * message for the last failure on the calling thread is gnn_last_error().  The
 * caller allocates all outputs.  One gnn_ctx is bound to one HIP device and one
 * HIP stream; a ctx is not thread safe, different ctxs are independent (one process
 * per GPU, or one ctx per device in one process).  "host"/"dev" in a parameter name
 * says where the pointer must live.  No exceptions cross the ABI.
 */
#ifndef GENOMAD_NN_H
#define GENOMAD_NN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNN_WINDOW 6000      /* nn_classification.py:68,72  window length / ljust width   */
#define GNN_TOKENS 5997      /* model.py:15                 6000 - 4 + 1 tokens            */
#define GNN_DEPTH 257        /* model.py:11                 one-hot depth (token 0 = N)    */
#define GNN_CH 128           /* model.py:20                 nb_filters_conv1d              */
#define GNN_KSIZE 6          /* model.py:22                 conv1d_kernel                  */
#define GNN_PATCHES 2100     /* model.py:19                 nb_patches                     */
#define GNN_PATCH_SIZE 4     /* igloo.py:33                 patch_size                     */
#define GNN_POOL 8           /* model.py:23                 pooling_size                   */
#define GNN_POOLED 749       /* igloo.py:176                int(5997 / 8)                  */
#define GNN_FEAT 256         /* igloo.py:83                 concat of the two IGLOO heads  */
#define GNN_HIDDEN 512       /* model.py:28,40                                             */
#define GNN_CLASSES 3        /* model.py:44                 chromosome, plasmid, virus     */

typedef enum gnn_status {
    GNN_OK = 0,
    GNN_ERR_ARG = -1,        /* bad argument (null pointer, negative size, bad enum) */
    GNN_ERR_HIP = -2,        /* a HIP runtime call failed (message has the HIP error)  */
    GNN_ERR_STATE = -3,      /* e.g. classify before gnn_load_weights                  */
    GNN_ERR_WEIGHTS = -4,    /* weight tensor invalid (patch index out of range ...)   */
    GNN_ERR_NOMEM = -5
} gnn_status;

/* Arithmetic of the conv / w_v contractions (everything else is always f32). */
typedef enum gnn_precision {
    GNN_PREC_F32 = 0,        /* f32 reference path: unfused f32 kernels, activations in HBM   */
    GNN_PREC_BF16X3 = 1,     /* fused path: split-bf16 (hi+lo), 3 MFMA passes, f32 accumulate (gnn_fused_x3.hip); f32 range:
                                what main() recomputes a batch with when an f16-operand mode returns non-finite scores */
    GNN_PREC_BF16 = 2,       /* REMOVED in round 6 (single bf16 MFMA pass, 7e-3: a roofline experiment).  The value is kept so that
                                an old caller gets GNN_ERR_STATE with a message instead of another arithmetic                    */
    GNN_PREC_F16C8 = 3,      /* REMOVED in round 6 (f16 + MX-fp8 correction MFMAs, 1.3e-4 on 10^6 windows).  GNN_ERR_STATE          */
    GNN_PREC_F16C6 = 5,      /* FROZEN opt-in fast mode (gnn_fused_c6.hip, not tuned any more): one f16 MFMA pass + MX-scaled fp6
                                (e2m3) correction MFMAs, both operands block scaled = 1.5 pass equivalents; 8.2e-5 on config 2,
                                1.2e-4 on a few of 10^6 windows - NO head-room under the 1e-4 tolerance (bench.py exits non-zero
                                with it), so it cannot be a default; needs |activation| < 65504 and a 4-byte aligned buffer   */
    GNN_PREC_F16X3TC = 6,    /* THE DEFAULT of main(), NNEngine and bench.py since round 4 (gnn_fused_tc.hip): the F16X3 arithmetic (split-f16
                                hi + lo limbs, three MFMA products per operand pair, f32 accumulate) with conv2 / conv3 evaluated by
                                Toom-Cook minimal filtering F(3,6) over the time axis - 0.444x their MFMAs, f32 transforms; y @ w_v direct,
                                logits GEMM split-f16 x 3 on the matrix pipe, dense head exact f32.  Class scores as F16X3's: within
                                2e-5 / 4e-5 of the exact-f32 path on every one of 10^6 windows for two weight sets
                                (profiles/r04_tails.txt).  Needs |activation| < ~2000 (the transformed activations, up to 32x the
                                activations, are f16 operands): beyond that the scores are non-finite, never silently wrong, and
                                main() recomputes the batch along the chain F16X3TC -> F16X3 (f16 range, 65504) -> BF16X3 (f32
                                range), logging every hop with the arithmetic that failed.  Since round 6 head A's y @ w_v rows
                                are not computed but gathered from a 1.38 GB table of all 9-mers that gnn_load_weights builds on
                                the device (x1[t] is a function of the bases t-5 .. t+3; f64 accumulation, rounded once).  A window
                                buffer that is not 4-byte aligned goes through one aligned staging copy and the same kernel: the
                                scores do not depend on the buffer's address                                                   */
    GNN_PREC_F16X3TK = 7,    /* round 6 (gnn_fused_tk.hip): F16X3TC with everything that is a function of a short k-mer READ FROM TABLES IN HBM
                                instead of computed - x2[t] = LeakyReLU(conv2(x1))[t] depends on the bases t-10 .. t+3, so conv2 (43 % of a
                                window's FLOPs) is one 512-byte row gather per position from a table of all 4^14 fourteen-mers (137.4 GB),
                                head A's pair products one 4-byte read per entry from an (entry, 9-mer) table (8.8 GB), rows no 14-mer
                                indexes (window starts, k-mers with a non-ACGT byte) <= 6 row reads of conv2's tap tables (8.3 GB); conv3 and head B's
                                y @ w_v stay on the matrix pipe with the F16X3TC arithmetic.  The tables are built on the device by
                                gnn_build_kmer_tables (f64 accumulation, rounded once: closer to exact f32 than the three f16 products);
                                without them this value answers GNN_ERR_STATE and the caller stays on F16X3TC.  Same range rule as F16X3TC */
    GNN_PREC_F16X3 = 4       /* the direct three-pass form (gnn_fused_x3.hip), the default of round 3: split-f16 (hi+lo, 11+11 significant
                                bits), 3 MFMA passes, logits GEMM split-f16 x 3 on the matrix pipe, dense head exact f32: f32-class
                                accuracy (within 2e-5 of the exact-f32 path on every one of 10^6 windows); needs |activation| < 65504
                                (f16 range)                                   */
} gnn_precision;

/* Kept for ABI compatibility: rounds 4 and 5 could link an experimental f16c8 kernel (GNN_EXPERIMENTAL=1); round 6 deleted it
 * together with the round-1 kernel.  Always 0. */
int gnn_has_experimental(void);

typedef enum gnn_onehot_dtype { GNN_OH_U8 = 0, GNN_OH_BF16 = 1, GNN_OH_F32 = 2 } gnn_onehot_dtype;

/*
 * Weights in the reference's own layouts (what Keras load_weights would put into the
 * graph of model.py:34-45; nn_classification.py:309-310).  All pointers are HOST
 * pointers, C-contiguous, float32 unless noted; the library copies and re-packs them.
 */
typedef struct gnn_igloo_weights {
    const int32_t* patches;  /* (2100,4,1) int32  igloo.py:129-135 "random_patches" */
    const float* w_mult;     /* (1,2100,4,128)    igloo.py:137-143                  */
    const float* w_summer;   /* (1,512,1)         igloo.py:144-150                  */
    const float* w_bias;     /* (1,2100)          igloo.py:166-172                  */
    const float* w_qk;       /* (2100,749)        igloo.py:174-180                  */
    const float* w_v;        /* (1,128,128)       igloo.py:182-188                  */
} gnn_igloo_weights;

typedef struct gnn_dense_bn {
    const float* kernel;     /* (in,out) Keras Dense kernel          model.py:28,40 */
    const float* bias;       /* (out,)                                              */
    const float* gamma;      /* (out,) BatchNormalization, eps=1e-3  model.py:29,41 */
    const float* beta;
    const float* mean;       /* moving_mean     */
    const float* var;        /* moving_variance */
} gnn_dense_bn;

typedef struct gnn_weights {
    const float* conv1_kernel;   /* (6,257,128)  igloo.py:45-47 on the one-hot input */
    const float* conv1_bias;     /* (128,)                                           */
    const float* conv2_kernel;   /* (6,128,128)  igloo.py:66 (loop iteration 1)      */
    const float* conv2_bias;
    const float* conv3_kernel;   /* (6,128,128)  igloo.py:66 (loop iteration 2)      */
    const float* conv3_bias;
    gnn_igloo_weights igloo_a;   /* igloo.py:54-62, applied to conv1 output           */
    gnn_igloo_weights igloo_b;   /* igloo.py:73-81, applied to conv3 output           */
    gnn_dense_bn enc;            /* Dense(512)+BN, in=256   model.py:28-30            */
    gnn_dense_bn head;           /* Dense(512)+BN, in=512   model.py:40-42            */
    const float* out_kernel;     /* (512,3)                 model.py:44               */
    const float* out_bias;       /* (3,)                                              */
} gnn_weights;

/* Host pointers that receive intermediates of gnn_debug_forward (any may be NULL). */
typedef struct gnn_taps {
    float* x1;       /* (n,5997,128) conv1+LeakyReLU            (F32 path only) */
    float* x2;       /* (n,5997,128)                            (F32 path only) */
    float* x3;       /* (n,5997,128)                            (F32 path only) */
    float* m_a;      /* (n,2100)   igloo.py:205-206 "mpi" of head A             */
    float* m_b;
    float* yp_a;     /* (n,749,128) max-pooled y @ w_v   igloo.py:208-210       */
    float* yp_b;
    float* alpha_a;  /* (n,749)    igloo.py:211-212                             */
    float* alpha_b;
    float* feat;     /* (n,256)    igloo.py:83 concat                           */
} gnn_taps;

typedef struct gnn_ctx gnn_ctx;

/* ---- life cycle -------------------------------------------------------------------- */
const char* gnn_last_error(void);
int gnn_version(void);
int gnn_device_count(int* count);
int gnn_create(int device, gnn_ctx** out);
int gnn_destroy(gnn_ctx* ctx);
int gnn_sync(gnn_ctx* ctx);                      /* hipStreamSynchronize on the ctx stream */
int gnn_device_info(gnn_ctx* ctx, char* name, size_t name_len, int* cus, int64_t* hbm_bytes);
int gnn_device_pci_bus_id(gnn_ctx* ctx, char* out, size_t out_len);   /* "0000:05:00.0" of the ctx's device (hipDeviceGetPCIBusId): what bench.py
                                                                         prints per rank, so that two ranks bound to one GPU are visible */
int gnn_device_mem_info(gnn_ctx* ctx, int64_t* free_bytes, int64_t* total_bytes);   /* hipMemGetInfo of the ctx's device (what the default launch size is clamped against) */

/* replaces nn_model.load_weights(GenomadData.nn_model_file), nn_classification.py:310 */
int gnn_load_weights(gnn_ctx* ctx, const gnn_weights* w);

/* The k-mer tables of GNN_PREC_F16X3TK: gnn_kmer_tables_bytes() bytes (155.9 GB) of device memory + 1.7 GB of temporaries while they
 * are built (about a second).  reserve_bytes = device memory that must stay free behind them (the workspaces of the launches to come and
 * the caller's own buffers; < 0 = the library's default: two workspaces of the ctx's launch size + 8 GiB).  GNN_ERR_NOMEM - nothing
 * allocated, message says how much is missing - on a device that cannot hold them: the caller keeps GNN_PREC_F16X3TC.  Idempotent.
 * gnn_drop_kmer_tables frees them (gnn_destroy does too). */
int gnn_build_kmer_tables(gnn_ctx* ctx, int64_t reserve_bytes);
int gnn_has_kmer_tables(gnn_ctx* ctx);
int gnn_drop_kmer_tables(gnn_ctx* ctx);
int64_t gnn_kmer_tables_bytes(void);
/* test aid: which = 0: row `row` (a 14-mer, first base in the top two of 28 bits; 4^14 = the all-N-token row) of the x2 table -> 128
 * floats; which = 1: head A's table, entry row >> 32 at the 9-mer row & 0xffffffff (4^9 = all-N-token) -> 1 float */
int gnn_debug_kmer_table_row(gnn_ctx* ctx, int which, uint64_t row, float* out_host);

/* ---- raw device memory (so a ctypes host needs no torch for buffers) ------------------ */
int gnn_dev_alloc(gnn_ctx* ctx, size_t bytes, void** dev_ptr);
int gnn_dev_free(gnn_ctx* ctx, void* dev_ptr);
int gnn_memcpy_h2d(gnn_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int gnn_memcpy_d2h(gnn_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);

/* ---- hot path ------------------------------------------------------------------------ */
/* replaces sequence.tokenize_dna(window, 4) (sequence.py:170-193) on n padded, upper-cased
 * 6000-byte windows: tokens_out[n][5997] in [0,256].  Host in / host out. */
int gnn_tokenize(gnn_ctx* ctx, const uint8_t* bases_host, int64_t n_windows, uint16_t* tokens_host);
int gnn_tokenize_dev(gnn_ctx* ctx, const uint8_t* bases_dev, int64_t n_windows, uint16_t* tokens_dev);

/* replaces OneHotLayer (model.py:9-11): bases -> tokens -> one-hot depth 257, written to
 * onehot_dev[n][5997][257] of the requested dtype (device pointer, caller-allocated). */
int gnn_onehot_dev(gnn_ctx* ctx, const uint8_t* bases_dev, int64_t n_windows, int onehot_dtype,
                   void* onehot_dev);

/* replaces the predict loop nn_classification.py:316-317 (+ tokenisation :72-73):
 * scores[n][3] = softmax class scores per window.  precision: gnn_precision. */
int gnn_classify(gnn_ctx* ctx, const uint8_t* bases_host, int64_t n_windows, int precision,
                 float* scores_host);
/* gnn_classify_dev_async: the same, but the back end (pair-product reduction, logits, attention sum, dense head) of the call's
 * last chunk may still be running on the library's second stream when the call returns, beside the front end of the NEXT
 * asynchronous call — for loops over batches that fit one launch.  scores_dev is complete once gnn_classify_flush (or any
 * other entry point of this ctx: gnn_sync, gnn_memcpy_d2h, gnn_comm_gather_dev, gnn_classify_dev ...) has been called — they
 * wait on the host for what is pending on the second stream — and the ctx stream has been synchronised; results are
 * bit-identical to gnn_classify_dev (bench.py checks every window of every run).  Worth +8-10 % for F16C6 at 2048 windows per
 * call, nothing for the power-bound default arithmetic.
 * Debug switches of the library (environment, read once): GNN_NO_BACKEND_OVERLAP=1 keeps every back end on the ctx stream;
 * GNN_BACKEND_OVERLAP=1 also overlaps the chunks of a synchronous multi-chunk call (the policy of rounds 2-4, slower beside the
 * power-bound default kernel: profiles/r04/backend_overlap_ab.txt);
 * GNN_DEBUG_POISON=1 fills the workspaces with NaN patterns before every launch (a kernel reading what its launch has not
 * written turns the scores into NaN); GNN_X3_ROUND1=1 serves F16X3 / BF16X3 with the round-1 kernel (A/B measurements);
 * GNN_NO_PAD_SKIP=1 / GNN_NO_TIME_SPLIT=1 start every ctx with the padding skip / the time split off (gnn_debug_set_*);
 * GNN_LOGITS_F32=1 keeps the logits GEMM on the f32 FMA path.  Each switch is read once and announced on stderr.
 * Memory: the asynchronous entry point alternates TWO workspaces (2 x 0.86 MB per window of a launch, see gnn_set_chunk). */
int gnn_classify_dev_async(gnn_ctx* ctx, const uint8_t* bases_dev, int64_t n_windows, int precision, float* scores_dev);
int gnn_classify_flush(gnn_ctx* ctx);

int gnn_classify_dev(gnn_ctx* ctx, const uint8_t* bases_dev, int64_t n_windows, int precision,
                     float* scores_dev);   /* asynchronous on the ctx stream */

/* replaces tf.math.segment_mean(pred, contig_ids) nn_classification.py:320.
 * ids sorted ascending, out has n_segments rows (zero row for an id with no window). */
int gnn_segment_mean(gnn_ctx* ctx, const float* scores_host, const int64_t* ids_host, int64_t n,
                     int64_t n_segments, float* out_host);

/* ---- contig front end: windows are spans of one packed contig buffer ------ */
/* counts[i] = number of bytes equal to `byte` in seq_dev[starts[i] .. starts[i]+lens[i]); used for the
 * window skip rule `window_n > 0 and seq_window.count("N") > 4000` (nn_classification.py:70-71, counted
 * on the raw, not upper-cased, sequence: sequence.py:38-39). */
int gnn_span_byte_count(gnn_ctx* ctx, const uint8_t* seq_dev, const int64_t* starts_host,
                        const int32_t* lens_host, int64_t n_spans, int byte, int32_t* counts_host);
/* replaces seq_window.seq_ascii.ljust(6000, b"N") + tokenize + predict for windows given as spans
 * (start, len <= 6000) of the raw contig buffer on the device (nn_classification.py:72-73, :316-317):
 * each span is upper-cased (sequence.py:35-36) and right-padded with 'N' on the device. */
int gnn_classify_spans(gnn_ctx* ctx, const uint8_t* seq_dev, const int64_t* starts_host,
                       const int32_t* lens_host, int64_t n_spans, int precision, float* scores_host);

/* The whole contig front end in one call: replaces generate_data (window cutting seq_windows(seq, 6000, 2500,
 * max_windows), the skip rule, upper-casing, padding, tokenising: nn_classification.py:54-82), the predict loop
 * (:316-318) and tf.math.segment_mean (:320) for a packed buffer of raw contig bytes: contig c =
 * seq[offsets[c] .. offsets[c+1]) (n_contigs + 1 offsets, non-decreasing).  seq is a HOST pointer when
 * seq_on_host != 0 (uploaded in pieces on a copy stream while earlier pieces are classified) and a device
 * pointer otherwise.  contig_scores_host[n_contigs][3] = mean class scores of the contig's kept windows;
 * window_ids_host (capacity >= number of CANDIDATE windows, sum over contigs of ceil(len / 6000) is enough)
 * receives the contig index of every KEPT window in order, *n_windows_out their number — the `contig_ids` the
 * reference stores in <prefix>_seq_window_id.npz.  Window scores never leave the device; every buffer is
 * persistent in the ctx.  Synchronous (returns when the scores are on the host). */
int gnn_classify_contigs(gnn_ctx* ctx, const uint8_t* seq, int seq_on_host, int64_t seq_bytes,
                         const int64_t* offsets_host, int64_t n_contigs, int single_window, int precision,
                         float* contig_scores_host, int64_t* window_ids_host, int64_t ids_capacity,
                         int64_t* n_windows_out);

/* ---- host-side FASTA record packer (no GPU needed) ------------------------------------------------ */
/* replaces the line loop of sequence.read_fasta(path, strip_n) (genomad/sequence.py:96-121) on an
 * in-memory text buffer (already decompressed, newlines normalised to '\n').
 * gnn_fasta_scan: number of header lines (= upper bound of the record count), total bytes of header
 * text, and whether the buffer contains a '\r' (the caller then normalises newlines first, as the
 * reference's text-mode read does).
 * gnn_fasta_pack: sequence bytes of every surviving record back to back into seq_out (capacity n;
 * MAY BE THE SAME BUFFER AS text: packing only moves bytes towards lower addresses), record i =
 * seq_out[offsets[i] .. offsets[i+1]) (capacity+1 entries), and its header text (without the '>')
 * = headers_out[header_offsets[i] .. header_offsets[i+1]) (headers_out: header_bytes from the scan).
 * seq_out == NULL (with strip_n == 0) is the index mode: nothing is copied, offsets hold cumulative
 * raw lengths — the headers of the non-empty records, which is what check_fasta needs
 * (sequence.py:124-131).
 * Rules: header = line starting with '>', text before the first header dropped, only '\n' removed,
 * strip_n strips leading/trailing n/N, empty records dropped. */
int gnn_fasta_scan(const uint8_t* text, int64_t n, int64_t* n_headers, int64_t* header_bytes, int* has_cr);
int gnn_fasta_pack(const uint8_t* text, int64_t n, int strip_n, uint8_t* seq_out, int64_t* offsets,
                   uint8_t* headers_out, int64_t* header_offsets, int64_t capacity, int64_t* n_records);
/* replaces, for a multi-rank run, the accession bookkeeping of sequence.check_fasta (genomad/sequence.py:124-131: no record /
 * two records with one accession = header.split()[0], sequence.py:24-25): 64-bit digests of the accessions of every record of
 * `text` whose raw sequence is non-empty, in file order, in ONE pass (capacity from gnn_fasta_scan).  *needs_python = 1 when a
 * header has a byte >= 0x80 before the end of its first token (Python's split() knows non-ASCII white space) or an empty
 * accession: the caller then recomputes this text with the Python mirror of the hash (genomad_amd/sequence.py).  Host only. */
int gnn_fasta_accession_digests(const uint8_t* text, int64_t n, uint64_t* digests, int64_t capacity, int64_t* n_records,
                                int* needs_python);

/* ---- downstream score consumers as a device epilogue, float64 like the
 * reference's numpy ------------------------------------------------------------------------------ */
/* replaces branch_attention(w, b1, b2, temperature) (aggregated_classification.py:10-34): w[n] marker
 * frequency, b1[n][3] marker scores, b2[n][3] nn scores -> out[n][3]. */
int gnn_branch_attention(gnn_ctx* ctx, const double* w_host, const double* b1_host, const double* b2_host,
                         int64_t n, double temperature, double* out_host);
/* replaces the inference part of score_batch_correction (score_calibration.py:37-43): scores[n][3] and
 * the (already smoothed, :18-21) composition[3] through the 6->20->20->3 tanh MLP whose weights are the
 * arrays of genomad/data/score_calibration_weights.npz (kernel_1 (6,20), bias_1, kernel_2 (20,20), ...). */
int gnn_score_calibration(gnn_ctx* ctx, const double* scores_host, const double* composition3,
                          const double* kernel1, const double* bias1, const double* kernel2,
                          const double* bias2, const double* kernel3, const double* bias3, int64_t n,
                          double* out_host);

/* CRC-32C (Castagnoli) of a host buffer: the checksum of the TFRecord framing that the reference's
 * write_tfrecord produces (nn_classification.py:43-52); used by genomad_amd/tfrecord.py. Host only. */
uint32_t gnn_crc32c(const void* data_host, size_t n_bytes);

/* same forward as gnn_classify, also copying intermediates out (parity tests). */
int gnn_debug_forward(gnn_ctx* ctx, const uint8_t* bases_host, int64_t n_windows, int precision,
                      float* scores_host, const gnn_taps* taps);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ---------------------------------
 * The path shards with no data-path collective: rank r classifies its own contiguous range of windows
 * (or contigs) with the replicated weights; the only exchange is the END-OF-RUN gather of the 12 B/window
 * class scores (the `predictions` the reference concatenates at nn_classification.py:316-319) to rank 0,
 * plus small control messages.  librccl.so is dlopen()ed on first use.  All collectives are enqueued on
 * the ctx stream; the host-buffer variants stage through device memory and synchronise before returning.
 * Bootstrap: rank 0 calls gnn_comm_unique_id and hands the 128 bytes to the other ranks out of band
 * (genomad_amd/rccl.py uses a file next to the launcher's rendezvous); every rank then calls gnn_comm_init. */
#define GNN_COMM_ID_BYTES 128
int gnn_comm_unique_id(uint8_t* id128);                                   /* ncclGetUniqueId */
int gnn_comm_init(gnn_ctx* ctx, int n_ranks, int rank, const uint8_t* id128);   /* ncclCommInitRank on ctx's device */
int gnn_comm_destroy(gnn_ctx* ctx);
int gnn_comm_info(gnn_ctx* ctx, int* n_ranks, int* rank);                  /* (1, 0) without a communicator */
/* ncclGather (rccl.h:745) of bytes_per_rank bytes from every rank to `root`: recv holds n_ranks blocks in rank
 * order (ignored on other ranks).  _dev: device pointers, asynchronous on the ctx stream. */
int gnn_comm_gather_dev(gnn_ctx* ctx, const void* send_dev, void* recv_dev, size_t bytes_per_rank, int root);
int gnn_comm_gather(gnn_ctx* ctx, const void* send_host, void* recv_host, size_t bytes_per_rank, int root);
int gnn_comm_allgather(gnn_ctx* ctx, const void* send_host, void* recv_host, size_t bytes_per_rank);
int gnn_comm_allreduce_max(gnn_ctx* ctx, double* values_host, int n);      /* in place; used for max-over-ranks timing */
int gnn_comm_barrier(gnn_ctx* ctx);

/* ---- synthetic data + measurement ------------------------------------------------------ */
/* windows first..first+n of the counter-based synthetic set (genomad_amd/synthetic.py) */
int gnn_synth_windows_dev(gnn_ctx* ctx, uint64_t seed, int64_t first, int64_t n_windows,
                          uint8_t* bases_dev);

/* HIP-event timing of the kernels launched on the ctx stream.  kernel ids: */
#define GNN_K_FUSED 0        /* fused tokens->conv1..3->IGLOO front end (dominant kernel) */
#define GNN_K_BACKEND 1      /* logits GEMM + softmax + attention + dense stack           */
#define GNN_K_ENCODER 2      /* stand-alone byte -> one-hot encoder                       */
#define GNN_K_F32_FRONT 3    /* unfused f32 front end (all its kernels)                   */
#define GNN_K_COUNT 4
int gnn_profile_enable(gnn_ctx* ctx, int on);
int gnn_profile_reset(gnn_ctx* ctx);
/* synchronises the stream, then total milliseconds and number of launches of kernel_id */
int gnn_profile_get(gnn_ctx* ctx, int kernel_id, double* total_ms, int64_t* launches);

/* debug aid: per-phase shader-cycle sums of the fused kernel (instrumented build).  on=1 starts
 * (zeroes the counters), on=0 stops; out16 (may be NULL) receives the 16 counters collected so far
 * (their meaning depends on the kernel variant: see GNN_TICK in gnn_fused.hip / gnn_fused2.hip). */
int gnn_phase_cycles(gnn_ctx* ctx, int on, unsigned long long* out16);

/* measurement aid: sustained dense bf16 rate (TFLOP/s) of v_mfma_f32_32x32x16_bf16 with one wave per
 * SIMD and register operands, run for about ms_target milliseconds — the practical MFMA ceiling of
 * this (power-managed) chip, reported by bench.py beside the fused kernel's issued-MFMA rate. */
int gnn_mfma_probe(gnn_ctx* ctx, int ms_target, double* tflops_out);
/* the same with the MFMA of `kind`: 0 = v_mfma_f32_32x32x16_bf16, 1 = v_mfma_f32_32x32x16_f16 (the instruction of the default
 * arithmetic: bench.py prints the issued f16 MFMA rate of the f16x3 kernel as a fraction of it, `frac_of_power_floor`), 2 = the
 * MFMA mix of f16c6 (8 f16 + 4 MX-fp6 scaled MFMAs per k32 step), reported in algorithmic TFLOP/s (the f16 MFMAs only) */
int gnn_mfma_probe_kind(gnn_ctx* ctx, int kind, int ms_target, double* tflops_out);

/* measurement aid: rows (token positions) a workgroup of the fused front end of `precision` streams per step
 * (128 for the f16c8 / x3 kernels; 32 * GNN_C6_NMB for f16c6), 0 for GNN_PREC_F32, negative on a bad enum. */
int gnn_fused_rows_per_step(int precision);

/* test aid: the f16c6 kernel does not compute the steps of a window that lie entirely in its all-N tail (the padding of a
 * contig's last window, nn_classification.py:72) — it copies the rows an all-N window produces, which the library computed
 * with the same kernel at gnn_load_weights time (bit-identical by construction).  on = 0 makes it compute everything; the
 * default is on (environment GNN_NO_PAD_SKIP=1 turns it off at gnn_create). */
int gnn_debug_set_pad_skip(gnn_ctx* ctx, int on);

/* test aid: launches of the default arithmetic with fewer windows than the device has CUs (the reference's own call shape, one
 * predict per 128 windows: nn_classification.py:316-317) deal every window's steps to up to 4 workgroups, each with one
 * warm-up step (gnn_fused_x3.hip) - bit-identical to the one-workgroup launch by construction.  on = 0 launches one workgroup per
 * window whatever the batch size; the default is on (environment GNN_NO_TIME_SPLIT=1 turns it off at gnn_create). */
int gnn_debug_set_time_split(gnn_ctx* ctx, int on);
/* workgroups per window of the ctx's last launch of a streaming kernel (F16X3TC / F16X3 / BF16X3): 1, or 2..4 under the time split */
int gnn_debug_last_split(gnn_ctx* ctx, int* workgroups_per_window);

/* test aid, host only (no GPU, no ctx): the f16c6 weight stream of a row-major K x N matrix (K multiple of 128, N of 32) as
 * gnn_load_weights builds it — per (k32 step, 32-column block) 3584 B: the f16 fragments of the two k16 halves (2 x 1 KiB:
 * lane l holds column l & 31, k = 16 s + 8 (l >> 5) + 0..7), the fp6 (e2m3) fragment dwords 0-3 (1 KiB) and 4-5 (512 B) of
 * MX block l >> 5 (0: w - f16(w), 1: w; element i of the step's 32 k in bits 6i..6i+5) — followed by the E8M0 scale words
 * [k / 128][block][lane], byte (k / 32) % 4.  need_words receives the size in 32-bit words; out may be NULL to query it. */
int gnn_debug_pack_c6(const float* w, int k, int n, uint32_t* out, size_t out_words, size_t* need_words);

/* Windows the ctx processes per launch of the fused front end.  Workspace: 0.86 MB per window of a launch (pair products, pooled
 * y @ w_v rows, logits, attention weights, features) - 14 GB at the default ceiling of 16384, allocated on the first large call and
 * TWICE that once gnn_classify_dev_async is used (two alternating workspaces).  Without a call to this function the library clamps
 * the default to a quarter of the device memory that is free when the workspace first grows (shared / partitioned GPUs) and says so
 * on stderr; with or without it, a failed workspace allocation is retried with half the launch size (down to 256 windows) before
 * GNN_ERR_NOMEM is returned, and a second workspace that does not fit makes the asynchronous path run in order on one.
 * 16384 vs 2048 windows per launch: +0.7 % throughput (profiles/r04/backend_overlap_ab.txt). */
int gnn_set_chunk(gnn_ctx* ctx, int64_t windows_per_chunk);

#ifdef __cplusplus
}
#endif
#endif /* GENOMAD_NN_H */
