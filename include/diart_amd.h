/*
 * diart_amd.h — C ABI of libdiart_amd.so: diart's per-chunk diarization hot path
 * on MI355X (gfx950), hand-written HIP.
 *
 * The reference (juanmc2005/diart v0.9) has no FFI: its plugin boundary is Python
 * duck typing ("Custom models", /root/reference/README.md:186-209).  Each entry point
 * below states which reference call it replaces; `diart_amd/_lib.py` holds the ctypes
 * binding a maintainer would add, and INTEGRATION.md shows how the resulting objects
 * plug into diart.models.SegmentationModel / EmbeddingModel unchanged.
 *
 * Conventions: every function returns 0 on success, non-zero on failure with a
 * message retrievable through dz_last_error() (thread local).  All `d_*` pointers
 * are DEVICE pointers owned by the caller (torch-ROCm tensors); the library only
 * borrows them for the duration of the call and owns nothing but its scratch arena.
 * `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls on one
 * handle must be serialised by the caller; distinct handles are independent.
 */
#ifndef DIART_AMD_H
#define DIART_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define DZ_VERSION 230   /* 2.3: kb-major f16 planes for the LDS-DMA kernels; dz_seg_front / dz_seg_back */

typedef struct dz_ctx dz_ctx;
typedef struct dz_seg dz_seg;
typedef struct dz_emb dz_emb;
typedef struct dz_clu dz_clu;

const char* dz_last_error(void);
int dz_version(void);
/* Run-time options of the library, by name; both return 2 (+ dz_last_error) for an unknown name.
 *   "f32_gemm"  (default 1): the wide exact-f32 layers on k_gemm_f32.hip; 0 = on k_convgemm.hip
 *   "pool_fuse" (default 1): statistics pooling inside the last x-vector layer's epilogue; 0 = two launches
 *   "pack_cache" (default 0): dz_k_sinc_conv0_split / dz_k_conv_pool re-order their register-resident operand into the
 *               kernel's fragment order on every call (the handles do it once, at create); 1 = skip that when the operand
 *               pointer is the previous call's (timing tools with fixed weights only)
 * Process-wide, read at every launch: set them while no forward pass is being enqueued.               */
int dz_set_option(const char* name, int value);
int dz_get_option(const char* name, int* value);
/* 1 when the library was compiled with -DDZ_EXPERIMENTS (diart_amd_experiments.h; the never-default kernels
 * and the timing-only modes whose results are wrong exist in that build only), else 0.                */
int dz_has_experiments(void);
/* Host worker pool (clustering / output tail of the N streams of a step): how long an idle worker polls for
 * the next job before it sleeps, in microseconds (default 40; 0 = sleep at once, for ranks with < 4 cores). */
int dz_host_pool_set_spin(int microseconds);

/* one context per (process, GPU) */
int dz_ctx_create(int hip_device, dz_ctx** out);
int dz_ctx_destroy(dz_ctx* ctx);
/* The default arithmetic ("f16x3") represents every GEMM operand as two f16 numbers, i.e. |x| <=
 * 65504; an f32 reference has no such limit.  Operands beyond it are clamped AND flagged: this
 * returns 0 if no kernel of the context has seen one since the last reset, 6 (with a message)
 * otherwise.  Call it after the stream(s) have been synchronised; reset != 0 clears the flag.   */
int dz_range_check(dz_ctx* ctx, int reset);

/* SincNet(stride 10) frame count for a chunk of `num_samples` (293 for 80000):
 * the F of SegmentationModel's (batch, frames, speakers) output,
 * /root/reference/src/diart/models.py:188-198. */
int dz_seg_frames_for(int num_samples);
/* frames left after the 5 TDNN layers (279 for 80000). */
int dz_emb_frames_for(int num_samples);

/* ---- packed weights (device pointers, fp32; layouts in DESIGN.md §3) ------- */
typedef struct {
    float wav_gamma, wav_beta;   /* InstanceNorm1d(1, affine) on the waveform        */
    const float* filt;           /* [128][96]  folded sinc FIR bank (weights.py fold_sinc_filters) */
    const float* in0_g;          /* [80]  InstanceNorm1d(80) gamma                   */
    const float* in0_b;          /* [80]                     beta                    */
    const float* w1;             /* [64][416]  conv1 [co][tap*80+ci], zero padded    */
    const float* b1;             /* [64]                                             */
    const float* in1_g;          /* [64]                                             */
    const float* in1_b;          /* [64]                                             */
    const float* w2;             /* [64][320]  conv2 [co][tap*64+ci]                 */
    const float* b2;             /* [64]                                             */
    const float* in2_g;          /* [64]                                             */
    const float* in2_b;          /* [64]                                             */
    const void* w1_split;        /* split-f16 planes of w1 / w2 (optional, NULL = exact f32)  */
    const void* w2_split;
    const void* filt_split;      /* [2][96][256] f16 planes of the UNFOLDED sinc bank, rows >= 80 and
                                    taps >= 251 zero (optional; NULL = the exact-f32 folded kernel) */
} dz_sincnet_weights;

typedef struct {
    dz_sincnet_weights sinc;
    const float* wih[4];         /* [1024][Kpad] rows = dir*512 + unit*4 + gate (unit-major: the
                                    recurrence reads the four gates of a unit as one 16-byte word) */
    const float* bih[4];         /* [1024]  b_ih + b_hh, same row order                */
    const float* whh[4];         /* [2][512][128]  PyTorch row order (gate*128 + unit) */
    const float* lin0_w;         /* [128][256] */
    const float* lin0_b;         /* [128]      */
    const float* lin1_w;         /* [128][128] */
    const float* lin1_b;         /* [128]      */
    const float* cls_w;          /* [64][128]  classifier rows, zero padded          */
    const float* cls_b;          /* [64]       */
    int num_classes;             /* K (multilabel) or 7 (powerset)                   */
    int powerset;                /* 1: log-softmax -> hard multilabel (models.py:29-39) */
    int num_speakers;            /* speakers of the multilabel output (3)            */
    /* split-f16 matrix path (optional; NULL = exact-f32 MFMA for that layer): the same matrices
     * as two f16 planes [2][Npad][Kpad], hi = f16(W), lo = f16((W - hi) * 2^11) (weights.py split_f16).
     * wih_split[0] row-major; wih_split[1..3], lin0_split and lin1_split — the layers whose operands go
     * global -> LDS by LDS-DMA — in the "kb-major" order [2][Kpad / 32][Npad][32] (weights.py kb_major):
     * element (n, k) of a plane at ((k / 32) * Npad + n) * 32 + k % 32                                  */
    const void* wih_split[4];
    const void* lin0_split;
    const void* lin1_split;
    /* W_hh as f16 planes [2 dir][2 (hi, lo*2^11)][512][128], PyTorch row order: the recurrence then
     * runs 16 chains per workgroup on the matrix cores (k_lstm_mfma.hip); NULL = one chain per CU on
     * the f32 vector units (k_lstm.hip, exact f32)                                               */
    const void* whh_split[4];
    int lstm_variant;            /* how whh_split was prepared (weights.py lstm_whh_planes): 0 / 3 = split_f16 of
                                    W_hh (3: gx by LDS-DMA); 1 / 2 (experiments build) = activation scales folded
                                    in, H scaled by 2^0 / 2^8; 4 = the software-pipelined kernel: gate rows times
                                    -log2(e) (i, f, o) / -2 log2(e) (g), columns in the kernel's k' order, AND
                                    wih / wih_split / bih carry the same row scales (gx arrives pre-scaled)      */
    const void* wih0_split_kb;   /* W_ih of layer 0 once more as kb-major planes [2][64 / 32][1024][32] (optional): the first
                                    projection then reads the SincNet output as planes a one-off norm_split_kernel wrote
                                    and runs on k_gemm_pre.hip like layers 1..3 (see dz_emb_weights.tw0_split_kb)        */
} dz_seg_weights;

typedef struct {
    dz_sincnet_weights sinc;
    const float* tw[5];          /* TDNN conv weights [Npad][Kpad], [co][tap*Cin+ci] */
    const float* tb[5];          /* conv bias [Npad]                                 */
    const float* ts[5];          /* folded BatchNorm1d(eval) scale  [Npad]           */
    const float* th[5];          /* folded BatchNorm1d(eval) shift  [Npad]           */
    const float* emb_w;          /* [512][3008]  Linear(3000, D), zero padded        */
    const float* emb_b;          /* [512] */
    int dimension;               /* D = 512 */
    const void* tw_split[5];     /* split-f16 planes of tw[i] (optional, NULL = exact f32): tw_split[0] row-major
                                    [2][Npad][Kpad], tw_split[1..4] kb-major (see dz_seg_weights)  */
    const void* tw0_split_kb;    /* tdnn1's planes once more in the kb-major order (optional): with them — and the SincNet's fused
                                    norms — the last SincNet stage's output is normalised and split ONCE (norm_split_kernel)
                                    and tdnn1 runs on the pre-split GEMM (k_gemm_pre.hip) like tdnn2..5                     */
    int pool_nearest;            /* how StatsPool resamples the (N, Fw) pooling weights to the T feature frames: 0 =
                                    F.interpolate(mode="linear", align_corners=False) (pyannote.audio 2.x .. 3.0),
                                    1 = mode="nearest" (pyannote.audio >= 3.1)                                     */
} dz_emb_weights;

/* ---- segmentation: replaces the callable behind SegmentationModel.__call__ --
 * /root/reference/src/diart/models.py:188-198 (-> pyannote PyanNet.forward, :133)
 * waveform (B,1,S) -> (B,F,K).  d_wave rows are `wave_stride` floats apart so a
 * rolling window can be addressed in place (operators.py:44-100).
 * A window holding a NaN / Inf sample gives a NaN row (what PyTorch's InstanceNorm1d makes of
 * it; blocks/clustering.py:137-145 then ignores the chunk) in BOTH arithmetic modes, and never
 * touches the other rows of the batch; the same holds for the embedding entry points below.  */
int dz_seg_create(dz_ctx* ctx, const dz_seg_weights* w, int max_batch, int num_samples, dz_seg** out);
int dz_seg_forward(dz_seg* seg, const float* d_wave, long long wave_stride, int batch,
                   float* d_out, void* stream);
/* The same forward pass that also leaves the OverlappedSpeechPenalty weights of its output
 * (blocks/embedding.py:98-107 -> functional.py:6-13; d_weights (B,K,F) speaker-major, the layout
 * dz_emb_pool consumes) — the N-stream driver's seg -> OSP hand-off without a launch of its own. */
int dz_seg_forward_osp(dz_seg* seg, const float* d_wave, long long wave_stride, int batch, float* d_out,
                       float gamma, float beta, int normalize, float* d_weights, void* stream);
int dz_seg_destroy(dz_seg* seg);

/* ---- embedding: replaces the callable behind EmbeddingModel.__call__ --------
 * /root/reference/src/diart/models.py:248-265 (-> XVectorSincNet.forward, :262)
 * waveform (N,1,S), weights (N,Fw) or NULL -> (N,D)                             */
int dz_emb_create(dz_ctx* ctx, const dz_emb_weights* w, int max_batch, int num_samples, dz_emb** out);
int dz_emb_forward(dz_emb* emb, const float* d_wave, long long wave_stride,
                   const float* d_weights, int n_rows, int weight_frames,
                   float* d_out, void* stream);
/* De-duplicated form of SpeakerEmbedding.__call__ (blocks/embedding.py:51-65): the
 * reference repeats each waveform K times and runs the full network per copy; only
 * the statistics pooling depends on the speaker, so frame features are computed once
 * per chunk and pooled K times.  d_weights is (B,K,Fw) speaker-major ("(batch spk)
 * frame", embedding.py:58); output (B,K,D).  normalize!=0 additionally applies
 * EmbeddingNormalization(norm=1) (functional.py:16-27).                          */
int dz_emb_forward_multi(dz_emb* emb, const float* d_wave, long long wave_stride,
                         const float* d_weights, int batch, int num_speakers,
                         int weight_frames, int normalize, float* d_out, void* stream);
/* dz_seg_forward_osp in two halves, for a caller that keeps the stateless front end of its NEXT step off
 * the dependent chain of the current one: dz_seg_front — SincNet + the first LSTM x-projection — may be
 * enqueued (on any stream) while dz_seg_back of the previous step on the SAME handle is still running its
 * recurrences; the handle orders the one buffer they share on the GPU.  dz_seg_back consumes what the last
 * dz_seg_front left (same batch) and must be ordered behind it by the caller (stream order or an event).   */
int dz_seg_front(dz_seg* seg, const float* d_wave, long long wave_stride, int batch, void* stream);
int dz_seg_back(dz_seg* seg, int batch, float* d_out, float gamma, float beta, int normalize,
                float* d_weights /* may be NULL: no OSP weights */, void* stream);

/* InstanceNorm1d(1) statistics of `batch` windows (the first op of BOTH networks' SincNet: the
 * reference computes them once per model, models.py:133 and :262 each run their own front end) as
 * dz_wave_stats_floats() floats per window (slice means and M2s, merged by the consumer).  A handle
 * told about them with dz_*_use_wave_stats skips its own pass over the waveform in its NEXT forward /
 * dz_emb_frames call (one use, rows in the same order as that call's windows; the caller orders the
 * streams).                                                                                      */
int dz_wave_stats_floats(void);
int dz_wave_stats(dz_ctx* ctx, const float* d_wave, long long wave_stride, int batch, int num_samples,
                  float* d_moments, void* stream);
int dz_seg_use_wave_stats(dz_seg* seg, const float* d_moments);
int dz_emb_use_wave_stats(dz_emb* emb, const float* d_moments);

/* The two halves of dz_emb_forward_multi.  dz_emb_frames (SincNet + TDNN stack, 99.5 % of the
 * embedding FLOPs) does not depend on the segmentation, so it can run on a second stream while
 * dz_seg_forward's latency-bound LSTM occupies a handful of CUs; dz_emb_pool then consumes the
 * OSP weights.  The frame features stay in the handle's scratch between the two calls, and until the
 * next dz_emb_frames: dz_emb_pool may be called again on the same frames with other weights (on the
 * split-f16 path with windows of >= 128 frames the last TDNN layer runs inside dz_emb_pool, with the
 * pooling in its epilogue, so every call pays for that layer again).                               */
int dz_emb_frames(dz_emb* emb, const float* d_wave, long long wave_stride, int batch, void* stream);
int dz_emb_pool(dz_emb* emb, const float* d_weights, int batch, int num_speakers,
                int weight_frames, int normalize, float* d_out, void* stream);
int dz_emb_destroy(dz_emb* emb);

/* ---- ECAPA-TDNN embedding (BASELINE.json config 3): replaces the callable behind
 * EmbeddingModel.__call__ when the embedding is speechbrain/spkrec-ecapa-voxceleb, i.e.
 * pyannote's PretrainedSpeakerEmbedding.__call__(waveforms, masks) reached through the fallback
 * of /root/reference/src/diart/models.py:59 and called at :262.
 * waveform (N,1,S), masks (N,Fw) or NULL -> (N,192); rows whose mask keeps fewer than 640
 * samples come back as NaN (they are dropped by clustering.py:143-145).  The mask selects
 * samples (nearest resampling, > 0.5), rows are zero padded to the longest row of the call and
 * the relative lengths drive the sentence normalisation, the squeeze-excitation means and the
 * attentive statistics pooling exactly as speechbrain's encode_batch does.                  */
typedef struct {
    const float* w;   /* [Npad][Kpad] packed like every convgemm weight                      */
    const float* b;   /* [Npad] bias                                                          */
    const float* s;   /* [Npad] folded BatchNorm scale (NULL if the layer has no norm)        */
    const float* h;   /* [Npad] folded BatchNorm shift                                        */
    const void* wsplit; /* optional split-f16 planes of w: the layer then runs on the f16 matrix cores.  Row-major
                           [2][Npad][Kpad] for block0, the Res2Net convolutions, asp_tdnn and asp_conv (dz_k_gemm_split);
                           kb-major [2][Kpad / 32][Npad][32] for the wide 1 x 1 layers tdnn1, tdnn2 and mfa, which
                           read pre-split activations (dz_k_gemm_pre, see dz_convgemm_desc.Xsplit)       */
} dz_layer;
typedef struct {
    dz_layer tdnn1;    /* 1x1, 1024 -> 1024                                                   */
    dz_layer res[7];   /* Res2Net: 128 -> 128, k = 3, dilation d, reflect "same" padding      */
    dz_layer tdnn2;    /* 1x1                                                                  */
    dz_layer se1;      /* [128][1024]  squeeze                                                 */
    dz_layer se2;      /* [1024][128]  excite                                                  */
} dz_seres2net;
typedef struct {
    const float* dft;      /* [448][416] hamming-windowed DFT: rows 0..200 cos, 201..401 sin  */
    const float* mel;      /* [128][224] triangular mel bank, [mel][bin], zero padded         */
    dz_layer block0;       /* [1024][416], k = tap*80 + mel                                    */
    dz_seres2net ser[3];   /* dilations 2, 3, 4                                                */
    dz_layer mfa;          /* [3072][3072]                                                     */
    dz_layer asp_tdnn;     /* w = columns of the 9216-wide input that multiply x: [128][3072]  */
    const float* asp_wms;  /* [128][6144] columns that multiply the global (mean | std)        */
    dz_layer asp_conv;     /* [3072][128]                                                      */
    dz_layer fc;           /* [192][6144] with asp_bn folded in                                */
    const float* zeros;    /* [6144] zeros                                                     */
    const void* dft_split; /* optional split-f16 planes [2][512][416] of dft (zero padded rows)        */
} dz_ecapa_weights;
typedef struct dz_ecapa dz_ecapa;
int dz_ecapa_frames_for(int num_samples);   /* 1 + S / 160 */
int dz_ecapa_create(dz_ctx* ctx, const dz_ecapa_weights* w, int max_rows, int num_samples,
                    dz_ecapa** out);
int dz_ecapa_forward(dz_ecapa* e, const float* d_wave, long long wave_stride, const float* d_masks,
                     int n_rows, int mask_frames, float* d_out, void* stream);
/* device pointer + element count of an intermediate of the LAST forward (parity tests):
 * 0 features (N,T,80)  1 block0 (N,T,1024)  2 cat (N,T,3072)  3 mfa (N,T,3072)
 * 4 pooled (N,6144)    5 kept-sample counts (N) as int32, -(count + 1) for a row whose kept
 * samples hold a NaN / Inf (its embedding is NaN);  *frames receives T                   */
int dz_ecapa_peek(dz_ecapa* e, int which, const void** d_ptr, long long* count, int* frames);
int dz_ecapa_destroy(dz_ecapa* e);

/* ---- OverlappedSpeechPenalty: functional.py:6-13 + blocks/embedding.py:98-107
 * d_seg (B,F,K) -> weights.  speaker_major=0: (B,F,K) like the reference block;
 * speaker_major=1: (B,K,F), the layout dz_emb_forward_multi consumes.           */
int dz_osp(dz_ctx* ctx, const float* d_seg, int batch, int frames, int speakers,
           float gamma, float beta, int normalize, int speaker_major,
           float* d_out, void* stream);

/* ---- the `.cpu()` of blocks/segmentation.py:47 and blocks/embedding.py:68 for a whole step: two device
 * buffers (n_a, n_b floats; n_b may be 0) -> two PINNED host buffers in one kernel launch on `stream`
 * (device stores over the host link; all four pointers 16-byte aligned).  Not hipMemcpyAsync: that call
 * now and then blocks its caller for a whole step's latency (csrc/ring.hip).  Complete when an event
 * recorded on `stream` behind it has fired.                                                       */
int dz_results_to_host(dz_ctx* ctx, const float* d_a, float* h_a, long long n_a,
                       const float* d_b, float* h_b, long long n_b, void* stream);

/* ---- EmbeddingNormalization(norm): functional.py:16-27; rows (R,D) in place  */
int dz_l2_normalize(dz_ctx* ctx, float* d_emb, int rows, int dim, float norm, void* stream);

/* ---- cosine distances for N streams at once: mapping.py:171-176 (scipy cdist,
 * fp64).  d_emb (N,K,D) f32, d_centers (N,G,D) f64 -> d_out (N,K,G) f64.        */
int dz_cdist_cosine(dz_ctx* ctx, const float* d_emb, const double* d_centers,
                    int n_streams, int k_local, int g_global, int dim,
                    double* d_out, void* stream);

/* ---- kernel-level entry points ---------------------------------------------
 * The building blocks of dz_seg_forward / dz_emb_forward, exported so that each HIP
 * kernel can be parity-tested on its own against a torch fp32 restatement of the same
 * op (tests/test_gpu_kernels.py).  Layouts: DESIGN.md §3.                          */
enum { DZ_EPI_BIAS = 0, DZ_EPI_BIAS_LEAKY = 1, DZ_EPI_BIAS_SIGMOID = 2, DZ_EPI_TDNN = 3,
       DZ_EPI_POOL3 = 4, DZ_EPI_BIAS_RELU = 5, DZ_EPI_RELU_BN = 6, DZ_EPI_RELU_BN_TANH = 7 };
typedef struct {
    const float* X;       /* [B][Tin][ldx] channels-last input                       */
    const float* W;       /* [Npad][Kpad], k = tap*Cin + c, zero padded              */
    const float* bias;    /* [Npad]                                                  */
    const float* e0;      /* TDNN: folded BatchNorm scale [Npad]                     */
    const float* e1;      /* TDNN: folded BatchNorm shift [Npad]                     */
    const float* nscale;  /* norm-on-load [B][nld] (InstanceNorm+LeakyReLU of input) */
    const float* nshift;
    float* Y;             /* [B][Tstore][ldy]                                        */
    float* partials;      /* POOL3: [B][ntile][Npad][2] (sum, sumsq) of pooled rows  */
    int B, Tin, Tout, Cin, taps, dil, K, Kpad, Npad, Nstore, ldx, ldy, nld, Tstore;
    long long xbs, ybs;   /* batch strides in floats                                 */
    int norm_on_load;     /* 0/1                                                     */
    int epi;              /* DZ_EPI_*                                                */
    int ksplit;           /* 0/1: off.  >1 (DZ_EPI_BIAS only): split z of the K loop writes its
                             partial sums to Y + z*ysplit (bias in split 0); the caller reduces */
    long long ysplit;     /* floats between the partial outputs of consecutive splits */
    int agroup;           /* activation tiles swept together per XCD (0 = default 4)  */
    int pad;              /* >0: "same" convolution, reflect padding of `pad` frames  */
    const float* X2;      /* optional second input with X's geometry, added on load   */
    const float* rowbias; /* optional [B][Npad] per-batch-item bias added to `bias`   */
    const void* Wsplit;   /* split-f16 path: W as two f16 planes [2][Npad][Kpad], hi = f16(W),
                             lo = f16((W - hi) * 2^11) (weights.py split_f16); NULL on the f32 path   */
    /* pre-split activations (k_gemm_pre.hip): the input as two f16 planes of R = xplane / ldx >= Tin rows
     * x ldx columns, hi at Xsplit, lo (scaled by 2^11) xplane ELEMENTS further, each plane in kb-major
     * order: [ldx / 32][R][32], element (t, c) at ((c / 32) * R + t) * 32 + c % 32 — the 32-wide k-tile
     * of consecutive rows is contiguous.  dz_k_gemm_pre also expects Wsplit in that order
     * ([Kpad / 32][Npad][32] per plane).  The output, when Ysplit is set (dz_k_gemm_pre, dz_k_gemm_split),
     * is written in the same form (yplane / ldy rows x ldy columns, lo plane yplane elements further; batch
     * item b of dz_k_gemm_split owns rows b * ybs / ldy ..) so that the next layer reads plain bytes.
     * Y (f32, row-major) and Ysplit may both be set.                                            */
    const void* Xsplit;
    long long xplane;
    void* Ysplit;
    long long yplane;
    /* norm-on-load without a finalize launch: instead of nscale / nshift the kernel is handed the
     * producer's tile partials [B][npart_tiles][nld][2] (sum, sumsq over npart_T values per channel)
     * and the InstanceNorm affine, and derives scale / shift itself (same f64 fixed-order
     * arithmetic as dz_k_finalize_norm).  Split-f16 path only (dz_k_gemm_split, dz_k_conv_pool).  */
    const float* npart;
    const float* ngamma;
    const float* nbeta;
    int npart_tiles, npart_T;
    int* oflag;           /* device-visible int set to 1 when an operand of the split-f16 path lies
                             outside +-65504 (it is then clamped); NULL = the context's flag, read
                             with dz_range_check                                                 */
} dz_convgemm_desc;
int dz_k_convgemm(dz_ctx* ctx, const dz_convgemm_desc* desc, void* stream);
/* the exact-f32 kernel of the wide layers alone (k_gemm_f32.hip; dz_k_convgemm routes to it by itself when the
 * layer is in its domain: f32 operands, no prologue / padding / split-K, Npad % 128 == 0, K = taps * Cin unpadded
 * with Cin % 32 == 0; dz_set_option("f32_gemm", 0) keeps every layer on the round-1 kernel).  Same arithmetic per product (one
 * exact f32 FMA), another order of the k sum.  Error if the layer is outside the domain.               */
int dz_k_gemm_f32(dz_ctx* ctx, const dz_convgemm_desc* desc, void* stream);
/* the same layer on the split-f16 matrix-core path (desc->Wsplit must be set)       */
int dz_k_gemm_split(dz_ctx* ctx, const dz_convgemm_desc* desc, void* stream);
/* ... with the activations pre-split as well (desc->Wsplit and desc->Xsplit set; B = 1, K = taps*Cin
 * unpadded, Cin % 32 == 0): operand tiles go global -> LDS by LDS-DMA                            */
int dz_k_gemm_pre(dz_ctx* ctx, const dz_convgemm_desc* desc, void* stream);
/* SincNet stages 1 / 2 (DZ_EPI_POOL3, k = 5, 64 output columns, Cin 80 or 64, norm-on-load, dense
 * rows) on the dedicated kernel: input tile resident in LDS, weights in registers; same
 * descriptor, outputs and partials as the POOL3 call of dz_k_gemm_split                       */
/* Tail of the segmentation network in one launch (csrc/k_mlp_head.hip: linear[0] -> linear[1] ->
 * classifier -> activation -> OSP weights without min-max) and the stand-alone classifier + activation
 * + OSP kernel it is checked against; inputs as the internal layers pass them (kb-major f16 hi/lo planes:
 * xsplit of exactly `rows` rows x 256, w0split [2][8][128][32], w1split [2][4][128][32]).                */
int dz_k_mlp_head(dz_ctx* ctx, const void* xsplit, long long xplane, const void* w0split, const void* w1split,
                  const float* b0, const float* b1, const float* cw, const float* cb, int rows, int frames,
                  int classes, int speakers, int powerset, float gamma, float beta, float* d_seg,
                  float* d_weights, void* stream);
int dz_k_seg_head(dz_ctx* ctx, const float* m1, const float* cw, const float* cb, int batch, int frames,
                  int classes, int speakers, int powerset, float* d_seg, float gamma, float beta,
                  int normalize, float* d_weights, void* stream);
/* (dz_k_conv_pool and dz_k_sinc_conv0_split first re-order their register-resident operand — desc->Wsplit / d_filt_split —
 * into the kernel's fragment order, in a scratch buffer of the CONTEXT: calls that share a context must be stream-ordered.
 * The network handles keep their own re-ordered copies, made once in dz_seg_create / dz_emb_create.)               */
int dz_k_conv_pool(dz_ctx* ctx, const dz_convgemm_desc* desc, void* stream);
int dz_k_convgemm_ntile(int t_out);
/* d_stats (B, 2) = (mean, 1/sqrt(biased var + 1e-5)) of each window: InstanceNorm1d(1).  Inside
 * the forward passes the 8 slice moments stay separate and the consumer merges them; this entry
 * point runs the slice kernel plus the merge and synchronises the stream.                     */
int dz_k_wave_stats(dz_ctx* ctx, const float* d_wave, long long stride, int batch, int samples,
                    float* d_stats, void* stream);
/* y0 (B, P0, 80) with P0 = ((S-251)/10+1)/3; partials (B, ntile0, 80, 2), ntile0 = ceil(F0/192);
 * d_filt (128, 96): the folded symmetric bank, see dz_sincnet_weights.filt */
int dz_k_sinc_conv0(dz_ctx* ctx, const float* d_wave, long long stride, int batch, int samples,
                    const float* d_stats, float gamma, float beta, const float* d_filt,
                    float* d_y0, float* d_partials, void* stream);
/* the same layer on the f16 matrix cores (split operands); d_filt_split as dz_sincnet_weights.filt_split;
 * partials (B, ntile, 80, 2) with ntile = dz_k_conv0_split_ntile(samples) (96-frame tiles) */
int dz_k_sinc_conv0_split(dz_ctx* ctx, const float* d_wave, long long stride, int batch, int samples,
                          const float* d_stats, float gamma, float beta, const void* d_filt_split,
                          float* d_y0, float* d_partials, void* stream);
int dz_k_conv0_split_ntile(int samples);
int dz_k_finalize_norm(dz_ctx* ctx, const float* d_partials, int batch, int ntile, int channels,
                       int frames, const float* d_gamma, const float* d_beta, float* d_scale,
                       float* d_shift, void* stream);
/* gx (B*T, 1024) = x-projection incl. biases (PyTorch column order dir*512 + gate*128 + unit),
 * whh (2,512,128) -> hout (B,T,256)                                                  */
int dz_k_lstm(dz_ctx* ctx, const float* d_gx, const float* d_whh, float* d_hout, int batch,
              int frames, void* stream);
/* the same recurrence on the f16 matrix cores, 16 chains per workgroup; d_whh_split / variant as
 * dz_seg_weights.whh_split / lstm_variant; unit_major != 0: gx columns are dir*512 + unit*4 + gate
 * (variants 3 / 4: unit-major only; variant 4: gx already times the gates' activation scales)      */
int dz_k_lstm_mfma(dz_ctx* ctx, const float* d_gx, const void* d_whh_split, float* d_hout,
                   int batch, int frames, int unit_major, int variant, void* stream);
/* either recurrence kernel (d_whh_split NULL: the f32 vector kernel on d_whh) writing h as the two
 * f16 planes of hplane / 256 >= B*T rows x 256 columns a dz_k_gemm_pre consumer reads, in kb-major order
 * (see dz_convgemm_desc.Xsplit): hi = f16(h) at d_hsplit, lo = f16((h - hi) * 2^11) hplane elements
 * further; gx in PyTorch column order (variants 3 / 4: unit-major, as dz_k_lstm_mfma)             */
int dz_k_lstm_planes(dz_ctx* ctx, const float* d_gx, const float* d_whh, const void* d_whh_split,
                     int variant, void* d_hsplit, long long hplane, int batch, int frames, void* stream);
/* weight_frames < 0: |weight_frames| weights per row, resampled with mode="nearest" (see dz_emb_weights.pool_nearest) */
int dz_k_stats_pool(dz_ctx* ctx, const float* d_x, int frames, int channels, int ldx,
                    const float* d_weights, int weight_frames, int rows, int rows_per_x,
                    float* d_out, int ldo, void* stream);
int dz_k_powerset(dz_ctx* ctx, const float* d_logits, int rows, int classes, int speakers,
                  float* d_out, void* stream);

/* ---- OnlineSpeakerClustering (host, fp64): blocks/clustering.py:10-218 with
 * the SpeakerMap algebra of mapping.py:179-360 and scipy's rectangular LSAP.   */
int dz_clu_create(double tau_active, double rho_update, double delta_new,
                  int max_speakers, dz_clu** out);
int dz_clu_reset(dz_clu* clu);
/* one chunk: seg (F,K) f32, emb (K,D) f32 (NaN allowed) -> scores (F,G) f64
 * (zeros for unassigned global speakers, mapping.py:341-360).
 * assign_out (K) receives the global speaker of each local speaker or -1.       */
int dz_clu_step(dz_clu* clu, const float* seg, int frames, int k_local,
                const float* emb, int dim, double* scores_out, int* assign_out);
/* the same for n independent streams (one clu handle each), run on host threads;
 * seg (n,F,K), emb (n,K,D), scores (n,F,G), assign (n,K).                        */
int dz_clu_step_batch(dz_clu** clus, int n, const float* seg, int frames, int k_local,
                      const float* emb, int dim, double* scores_out, int* assign_out,
                      int num_threads);
/* state: centers (G,D) f64 copied to `out` (returns 1 if not initialised yet),
 * active mask (G) ints.                                                          */
int dz_clu_get_centers(dz_clu* clu, double* out, int dim);
int dz_clu_get_active(dz_clu* clu, int* out_mask);
/* embedding dimension of the centroid matrix, 0 before the first chunk (centers is None) */
int dz_clu_dim(dz_clu* clu);
int dz_clu_set_state(dz_clu* clu, const double* centers, const int* active_mask, int dim);
int dz_clu_destroy(dz_clu* clu);

/* ---- per-kernel timing (HIP events on the launch stream) for bench.py's roofline leg ----
 * dz_prof_enable(1) starts bracketing every kernel the forward passes launch; dz_prof_collect()
 * synchronises the device and accumulates; dz_prof_get(tag) reads name / total ms / launches. */
int dz_prof_enable(int on);
int dz_prof_pause(int paused);   /* suspend / resume bracketing, accumulators untouched */
int dz_prof_collect(void);
/* chunks: total number of 5 s chunks the tag's bracketed launches processed (a launch of a
 * 32-chunk sub-batch counts 32): the unit the per-launch algorithmic work is priced in */
int dz_prof_get(int tag, const char** name, double* total_ms, long long* launches,
                long long* chunks);

/* ---- device-resident rolling window of N streams ------------------------------------------
 * Replaces rearrange_audio_stream (/root/reference/src/diart/operators.py:44-100) plus the
 * per-chunk upload of the full window (blocks/segmentation.py:47, blocks/embedding.py:52) for the
 * N-stream driver: every step only the `hop` new samples of each stream are pushed (32 KB
 * instead of 320 KB per stream at 5 s / 500 ms); dz_ring_window returns the (pointer, row
 * stride) pair dz_seg_forward / dz_emb_frames read in place.  window % hop == 0, hop % 4 == 0.
 * slack_blocks extra blocks of history are kept so that pushing block t+1 never overwrites a
 * sample of windows t-slack_blocks+1 .. t (forward passes of those may still be in flight).     */
typedef struct dz_ring dz_ring;
int dz_ring_create(dz_ctx* ctx, int n_streams, int window, int hop, int slack_blocks, dz_ring** out);
int dz_ring_reset(dz_ring* r);
int dz_ring_destroy(dz_ring* r);
/* block (n_streams, hop) with block_stride floats between rows; host memory (on_device = 0;
 * pinned memory makes the copy asynchronous), device memory (on_device = 1), or pinned host memory
 * to be read in place by the GPU when the runtime can map it (on_device = 2; falls back to 0).   */
int dz_ring_push(dz_ring* r, const float* block, long long block_stride, int on_device, void* stream);
/* *filled = min(window, samples pushed): the window is complete once *filled == window.     */
int dz_ring_window(const dz_ring* r, const float** d_wave, long long* stride, int* filled);
/* contiguous (n_streams, window) copy of the current window, device to device.               */
int dz_ring_read(const dz_ring* r, float* d_out, void* stream);
/* Streams that advance at their own pace (one rearrange_audio_stream per stream,
 * /root/reference/src/diart/inference.py:101-147 + console/serve.py:105-127, served as ONE batch):
 * every row has its own write position.  dz_ring_push_rows: row j of block (k, hop) is the next
 * block of stream rows[j] (distinct rows; on_device as for dz_ring_push).  dz_ring_gather: the
 * current windows of the listed streams as a dense (k, window) device batch for dz_seg_forward /
 * dz_emb_frames — each must be complete (dz_ring_filled_row).  dz_ring_reset_row: the stream left. */
int dz_ring_push_rows(dz_ring* r, const float* block, long long block_stride, int on_device,
                      const int* rows, int k, void* stream);
int dz_ring_filled_row(const dz_ring* r, int row, int* filled);
int dz_ring_gather(const dz_ring* r, const int* rows, int k, float* d_out, long long out_stride, void* stream);
int dz_ring_reset_row(dz_ring* r, int row);

/* ---- output tail of one stream: DelayedAggregation + Binarize, host fp64 -------------------
 * Replaces, for the N-stream driver, the per-chunk Python tail of SpeakerDiarization.__call__
 * (/root/reference/src/diart/blocks/diarization.py:203-232): DelayedAggregation
 * (blocks/aggregation.py:120-218; strategies :60-118; first-chunk prepend :188-211) followed by
 * Binarize (blocks/utils.py:11-59).  State: the last round(latency/step) permuted score
 * buffers of the stream.  dz_tail_step consumes the (frames, speakers) fp64 scores of the newest
 * chunk (what dz_clu_step wrote), whose frame grid starts at chunk_start seconds with
 * `resolution` seconds per frame (diarization.py:190,195-199), and returns
 *   agg_out  (rows, speakers) aggregated scores of the region [t0, t0 + rows*res), rows <=
 *            dz_tail_max_rows() = frames + 2 (the first chunk of a stream outputs everything up
 *            to the end of its region);
 *   turns_out (nturns, 3) = (start, end, speaker index) of `score > threshold` runs, ordered by
 *            speaker then time; turns_out may be NULL.  More than max_turns turns -> error 5.  */
enum { DZ_AGG_HAMMING = 0, DZ_AGG_MEAN = 1, DZ_AGG_FIRST = 2 };
enum { DZ_CROP_STRICT = 0, DZ_CROP_LOOSE = 1, DZ_CROP_CENTER = 2 };
typedef struct dz_tail dz_tail;
int dz_tail_create(int frames, int speakers, double step, double latency, double threshold,
                   int strategy, int cropping_mode, const double* hamming /* [frames] */,
                   dz_tail** out);
int dz_tail_reset(dz_tail* t);
int dz_tail_destroy(dz_tail* t);
int dz_tail_max_rows(const dz_tail* t);
int dz_tail_step(dz_tail* t, const double* scores, double chunk_start, double resolution,
                 double* agg_out, int* rows_out, double* t0_out, double* res_out,
                 double* turns_out, int max_turns, int* nturns_out);
/* n streams on host threads: scores (n, frames, speakers); chunk_start, resolution (n);
 * agg_out (n, frames + 2, speakers); rows_out, t0_out, res_out, nturns_out (n);
 * turns_out (n, max_turns, 3) or NULL.                                                        */
int dz_tail_step_batch(dz_tail** tails, int n, const double* scores, const double* chunk_start,
                       const double* resolution, double* agg_out, int* rows_out, double* t0_out,
                       double* res_out, double* turns_out, int max_turns, int* nturns_out,
                       int num_threads);

/* ---- file-parallel evaluation: the host half of a GPU step over several files --------------
 * Replaces the per-chunk Python loop of SpeakerDiarization.__call__
 * (/root/reference/src/diart/blocks/diarization.py:193-232) as driven by Benchmark, one file at a
 * time in batches of consecutive windows (/root/reference/src/diart/inference.py:392-432).  The rows
 * of the GPU batch are file-major: file i contributes its next count[i] CONSECUTIVE windows, rows
 * row0[i] .. row0[i] + count[i] - 1 of seg (rows, frames, k_local) / emb (rows, k_local, dim) /
 * chunk_start (rows).  Per file, in window order: dz_clu_step -> dz_tail_step on that file's own
 * handles; files run in parallel on host threads.  turns_out (rows, max_turns, 3), nturns_out (rows):
 * the speech turns each window finalises; assign_out (rows, k_local) or NULL.                    */
int dz_file_step_batch(dz_clu** clus, dz_tail** tails, int n_files, const int* row0, const int* count,
                       const float* seg, int frames, int k_local, const float* emb, int dim,
                       int max_speakers, const double* chunk_start, double resolution,
                       double* turns_out, int max_turns, int* nturns_out, int* assign_out,
                       int num_threads);

/* sizeof() of the five structs that cross this boundary, in declaration order
 * (dz_sincnet_weights, dz_seg_weights, dz_emb_weights, dz_ecapa_weights, dz_convgemm_desc): a
 * binding checks its own mirror of the layouts against the library it loaded.            */
int dz_abi_struct_sizes(int out[5]);

/* exposed for tests: scipy.optimize.linear_sum_assignment (minimise), rows<=cols
 * or transposed internally; col4row (nr) gets the column of each row.            */
int dz_lsap(const double* cost, int nr, int nc, int* col4row);

#ifdef __cplusplus
}
#endif
#endif /* DIART_AMD_H */
