/*
 * mcm.h — C ABI of libmcm_hip.so, the MI355X (gfx950) Maximum-Concept-Matching scorer.
 *
 * The reference (deeplearning-wisc/MCM) has no FFI: its boundary is the duck-typed
 * Python contract of utils/detection_util.py:209-249 (`get_ood_scores_clip`) over a
 * `net` exposing `get_image_features` / `get_text_features` (HF transformers
 * modeling_clip.py:683-753).  This header is the C boundary a maintainer binds
 * *underneath* that contract (ctypes stub: INTEGRATION.md).  Each entry point names the
 * reference interface it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / C++ types cross the boundary;
 *   - every function returns 0 on success or a negative MCM_E* code and never throws;
 *     mcm_last_error(h) returns a human-readable message for the last failure;
 *   - `*_dev` pointers are device (HBM) addresses, `*_host` pointers are host addresses;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all compute
 *     entry points are asynchronous on it;
 *   - no device allocation happens after mcm_create (workspace is sized from
 *     cfg.max_batch / cfg.max_prompt_tokens), so calls are hipGraph-capturable;
 *   - one handle is used from one host thread at a time (the reference is
 *     single-threaded: utils/detection_util.py:219).
 */
#ifndef MCM_H_
#define MCM_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCM_ABI_VERSION 4 /* 2: mcm_set_weight takes the host element type; mcm_config.weight_operands; split-weight
                          * arm and mcm_weights_operand_exact; mcm_op_linear_ex / mcm_op_split_weight; the round-2
                          * mcm_debug_* exports live in libmcm_hip_harness.so only
                          * 3: MCM_KC_COUNT 7 -> 11 (per-shape GEMM classes: mcm_profile_read's arrays grew); the
                          * split-activation arm (mcm_score_x2 ...) and the MCM_LINEAR_SPLIT_X / _OUT flags
                          * 4: mcm_config.x2_max_batch (the split-activation workspace is sized by the caller);
                          * mcm_kernel_faults + the sticky fault word (every wait of a persistent kernel is bounded) */

/* error codes */
#define MCM_OK 0
#define MCM_EINVAL (-1)   /* bad argument (shape, size, NULL)                         */
#define MCM_ENOMEM (-2)   /* hipMalloc failed                                         */
#define MCM_EHIP (-3)     /* a HIP runtime call failed                                */
#define MCM_ENOWEIGHT (-4)/* a required parameter was never set                       */
#define MCM_ENAME (-5)    /* unknown parameter name                                   */
#define MCM_ESHAPE (-6)   /* parameter shape does not match the config                */
#define MCM_ERANGE (-7)   /* batch / prompt count / sequence exceeds the config       */

/* arithmetic modes for the GEMM/attention operands of the VISION tower (accumulation is always fp32,
 * the residual stream, LayerNorm statistics, softmax and the whole scoring tail are always fp32).
 * The TEXT tower always runs in MCM_PREC_F32: the prompt bank is encoded once per dataset, off the hot
 * loop, and rounding it would perturb every image's cosines in the same direction. */
#define MCM_PREC_BF16 0   /* bf16 MFMA operands  (v_mfma_f32_16x16x32_bf16)            */
#define MCM_PREC_F32 1    /* exact fp32 MFMA     (v_mfma_f32_16x16x4_f32) — parity arm  */
#define MCM_PREC_F16 2    /* fp16 MFMA operands  (v_mfma_f32_16x16x32_f16): same rate as bf16,
                           * 10-bit mantissa; BASELINE config 4 (ViT-L/14 fp16)        */

/* element type of a host parameter buffer handed to mcm_set_weight */
#define MCM_DT_F32 0
#define MCM_DT_F16 1      /* IEEE binary16 (the dtype OpenAI's CLIP checkpoints carry their Linear / conv weights in) */
#define MCM_DT_BF16 2

/* How the 16-bit vision modes hold a GEMM weight (cfg.weight_operands).  A weight that IS an fp16 (bf16) number —
 * the reference's checkpoints: values trained and released in fp16, widened to fp32 by the HF conversion — is
 * lossless as one 16-bit MFMA operand.  A weight that is not (seeded fp32 parameters, a fine-tuned fp32 checkpoint)
 * would be ROUNDED, a fixed perturbation of the model that moves AUROC / FPR95 by more than the activation rounding
 * does.  The split form keeps such a weight as W_hi + W_lo (two 16-bit operands, 22 significand bits), two MFMAs per
 * fragment pair into the same fp32 accumulator: exact for the weight operand at twice the GEMM work. */
#define MCM_WEIGHTS_AUTO 0   /* split exactly when some GEMM weight of the vision tower does not round-trip through
                              * the operand dtype (counted on the device in mcm_finalize_weights)               */
#define MCM_WEIGHTS_SINGLE 1 /* one rounded operand whatever the values (rounds 1 - 3 behaviour)                 */
#define MCM_WEIGHTS_SPLIT 2  /* always the split form                                                            */

/* score kinds — the epilogues of utils/detection_util.py:233-248 */
#define MCM_SCORE_MCM 0       /* -max_k softmax(cos/T)             (:236,:248) */
#define MCM_SCORE_MAX_LOGIT 1 /* -max_k cos                        (:234,:248) */
#define MCM_SCORE_ENERGY 2    /* -T*logsumexp(cos/T)               (:239)      */
#define MCM_SCORE_ENTROPY 3   /* Shannon entropy of softmax(cos/T) (:243)      */
#define MCM_SCORE_VAR 4       /* -var_k softmax(cos/T), ddof=0     (:246)      */

typedef struct mcm_handle mcm_handle;

/* Model geometry = the fields of transformers CLIPVisionConfig / CLIPTextConfig that
 * the path reads (configuration_clip.py), plus workspace bounds. */
typedef struct mcm_config {
  int32_t abi_version;      /* must be MCM_ABI_VERSION                                */
  int32_t device;           /* HIP device ordinal                                     */
  int32_t precision;        /* MCM_PREC_*                                             */
  /* vision tower (modeling_clip.py:138-218, 594-656) */
  int32_t image_size;       /* 224                                                    */
  int32_t patch_size;       /* 16 (B/16), 32 (B/32), 14 (L/14)                        */
  int32_t v_width;          /* 768 / 1024                                             */
  int32_t v_heads;          /* 12 / 16  (head_dim = width / heads must be 64)         */
  int32_t v_layers;         /* 12 / 24                                                */
  int32_t v_mlp;            /* 3072 / 4096                                            */
  /* text tower (modeling_clip.py:221-256, 494-586) */
  int32_t vocab_size;       /* 49408                                                  */
  int32_t max_positions;    /* 77                                                     */
  int32_t t_width;          /* 512 / 768                                              */
  int32_t t_heads;          /* 8 / 12                                                 */
  int32_t t_layers;         /* 12                                                     */
  int32_t t_mlp;            /* 2048 / 3072                                            */
  int32_t proj_dim;         /* 512 / 768  (visual_projection / text_projection rows)  */
  float ln_eps;             /* 1e-5 (configuration_clip.py layer_norm_eps)            */
  /* workspace bounds */
  int32_t max_batch;        /* images per mcm_encode_image / mcm_score call           */
  int32_t max_prompt_tokens;/* K*S per mcm_encode_text call                           */
  int32_t weight_operands;  /* MCM_WEIGHTS_*; ignored in MCM_PREC_F32                 */
  int32_t x2_max_batch;     /* MCM_PREC_F16 handles: images per mcm_score_x2 / mcm_encode_image_x2 call.  0 = max_batch (the
                             * activation buffers at twice the bytes: +1.5 GB at ViT-B/16 batch 512), n > 0 = at most n (the
                             * buffers hold max(max_batch rows, 2 x the rows of n images): nothing extra for n <= max_batch / 2),
                             * < 0 = no split-activation workspace (those calls return MCM_EINVAL)  */
} mcm_config;

int mcm_abi_version(void);

/* Replaces CLIPModel.__init__ + .cuda() (utils/train_eval_util.py:23-24): allocates the
 * parameter store and the activation workspace on cfg->device. */
int mcm_create(const mcm_config* cfg, mcm_handle** out);
void mcm_destroy(mcm_handle* h);
const char* mcm_last_error(const mcm_handle* h); /* h may be NULL: last create error */

/* Replaces load_state_dict / from_pretrained (utils/train_eval_util.py:23): copies one
 * parameter, keyed by its HF state_dict name, host → device.  `dtype` = MCM_DT_*: the element
 * type of host_ptr (an fp16 / bf16 checkpoint is handed over as it is; the fp32 master the library keeps is the exact
 * widening).  The caller keeps ownership of host_ptr.  `shape`/`ndim` are checked against cfg. */
int mcm_set_weight(mcm_handle* h, const char* hf_name, const void* host_ptr, int32_t dtype,
                   const int64_t* shape, int32_t ndim);
/* Verifies every parameter was set, counts the vision-tower GEMM weights that are not exactly representable in the
 * operand dtype, chooses the weight form (cfg.weight_operands) and builds the packed operand copies the kernels
 * read (concatenated QKV, 16-bit or split copies).  Must be called once before any encode. */
int mcm_finalize_weights(mcm_handle* h);
/* After mcm_finalize_weights: *inexact_host = number of GEMM-weight elements of the vision tower (q/k/v/out_proj, fc1,
 * fc2 of every layer and the patch embedding) whose fp32 value is not an fp16 (bf16 in MCM_PREC_BF16) number — 0 for the
 * reference's checkpoints in fp16, ~99.9 % of the elements for fp32-valued parameters; always 0 in MCM_PREC_F32.
 * *split_host = 1 when the handle runs the split-weight GEMMs (MCM_WEIGHTS_*), else 0.  Either pointer may be NULL. */
int mcm_weights_operand_exact(mcm_handle* h, uint64_t* inexact_host, int32_t* split_host);

/* Replaces `net.get_text_features(input_ids, attention_mask).float()` followed by
 * `/= norm` (utils/detection_util.py:229-231; modeling_clip.py:683-715).
 * ids_host: int32 [K,S] row-major, each row BOS … EOS pad…; the pooled row is the
 * first EOS (= argmax id, modeling_clip.py:561-581).  attention_mask is not needed:
 * the mask is causal and pooling reads the first EOS.
 * out_dev: fp32 [K, proj_dim], L2-normalised rows. */
int mcm_encode_text(mcm_handle* h, const int32_t* ids_host, int32_t K, int32_t S,
                    float* out_dev, void* stream);

/* Replaces `net.get_image_features(pixel_values=images).float()` followed by `/= norm`
 * (utils/detection_util.py:225-226; modeling_clip.py:719-753).
 * pixels_dev: fp32 NCHW [B,3,image_size,image_size] contiguous, already normalised by
 * the preprocess of utils/train_eval_util.py:27-33.
 * out_dev: fp32 [B, proj_dim], L2-normalised rows. */
int mcm_encode_image(mcm_handle* h, const float* pixels_dev, int32_t B, float* out_dev,
                     void* stream);

/* The two calls above with the reference's `/= norm` step optional: normalize == 0 returns exactly what
 * HF `get_image_features` / `get_text_features` return (modeling_clip.py:751,713 — the projection
 * output), which is what unmodified reference code expects from `net` (utils/detection_util.py:158,
 * 187,225,229) before it normalises the rows itself.  pixel_format selects the ingest layout. */
#define MCM_PIXELS_F32_NCHW 0 /* fp32 [B,3,S,S], already normalised (mcm_encode_image)            */
#define MCM_PIXELS_U8_NHWC 1  /* uint8 [B,S,S,3]; ToTensor + Normalize fused (mcm_encode_image_u8) */
int mcm_encode_image_ex(mcm_handle* h, const void* pixels_dev, int32_t pixel_format, int32_t B,
                        int32_t normalize, float* out_dev, void* stream);
int mcm_encode_text_ex(mcm_handle* h, const int32_t* ids_host, int32_t K, int32_t S, int32_t normalize,
                       float* out_dev, void* stream);

/* Replaces the scoring tail utils/detection_util.py:232-248 on already-normalised
 * features: sim = img @ text.T, softmax(sim/T), reduction `kind` → scores_dev fp32 [B].
 * Never materialises [B,K]. */
int mcm_score_features(mcm_handle* h, const float* img_feat_dev, int32_t B,
                       const float* text_feat_dev, int32_t K, float T, int32_t kind,
                       float* scores_dev, void* stream);

/* One batch of the whole hot loop body utils/detection_util.py:223-248:
 * mcm_encode_image + mcm_score_features, text bank pre-encoded (hoisted out of the
 * loop; the reference re-encodes it every batch with identical results). */
int mcm_score(mcm_handle* h, const float* pixels_dev, int32_t B, const float* text_feat_dev,
              int32_t K, float T, int32_t kind, float* scores_dev, void* stream);

/* uint8 ingest: pixels_dev is uint8 NHWC [B,image_size,image_size,3] (what a
 * JPEG decoder + resize/crop produces); ToTensor (/255) and Normalize with the CLIP mean/std of
 * utils/train_eval_util.py:27-33 are fused into the patch gather.  Same results as
 * mcm_encode_image / mcm_score on the equivalent fp32 NCHW tensor. */
int mcm_encode_image_u8(mcm_handle* h, const uint8_t* pixels_dev, int32_t B, float* out_dev,
                        void* stream);
int mcm_score_u8(mcm_handle* h, const uint8_t* pixels_dev, int32_t B, const float* text_feat_dev,
                 int32_t K, float T, int32_t kind, float* scores_dev, void* stream);

/* ---- split-activation arm (ABI 3): the re-scorer of threshold refinement ------------------------------------------
 * The same image tower on the same weights and the same workspace, fp16 handles only, with every activation that feeds
 * an MFMA carried as a split pair x = hi + lo of fp16 numbers (~22 significand bits; hi = round(x), lo = round(x - hi))
 * and every GEMM run as X_hi W^T + X_lo W^T in one fp32 accumulator chain — the mirror image of the split-WEIGHT form
 * (MCM_WEIGHTS_SPLIT; with split weights as well: four passes per K-step); attention as three MFMA passes per product.
 * Scores agree with the exact-fp32 arm (MCM_PREC_F32) to fp32 round-off at about half the fp16 arm's throughput, where
 * the fp32 arm runs at a tenth of it: what mcm_amd/refine.py re-scores the images near the FPR95 threshold with
 * (reference utils/detection_util.py:66-106: FPR95 is a count at one threshold).  No second handle, no extra weights:
 * the split rows (twice the width) live in the same activation buffers, which mcm_create sizes for cfg.x2_max_batch images
 * of them: B <= mcm_x2_max_batch(h) (0 = this handle is not fp16, or was created without the workspace).  pixel_format: MCM_PIXELS_*.
 * Asynchronous on `stream`, no allocation, like mcm_score. */
int mcm_x2_max_batch(const mcm_handle* h);
int mcm_encode_image_x2(mcm_handle* h, const void* pixels_dev, int32_t pixel_format, int32_t B, int32_t normalize,
                        float* out_dev, void* stream);
int mcm_score_x2(mcm_handle* h, const void* pixels_dev, int32_t pixel_format, int32_t B, const float* text_feat_dev,
                 int32_t K, float T, int32_t kind, float* scores_dev, void* stream);

/* Resize(S) + CenterCrop(S), S = cfg.image_size, on the device: the first two
 * steps of the reference's loader transform, utils/train_eval_util.py:27-33 (transforms.Resize(224),
 * transforms.CenterCrop(224) on the PIL image; torchvision's size / crop rules around Pillow's
 * antialiased BILINEAR Image.resize).  Bit-exact against Pillow (tests/golden/preprocess.npz).
 * src_dev_ptrs / heights / widths are HOST arrays of length B: device pointers to [H_i, W_i, 3]
 * uint8 RGB images and their sizes.  dst_dev [B, S, S, 3] uint8 is the layout mcm_score_u8 and
 * mcm_encode_image_u8 take.  MCM_ERANGE: B > max_batch, or a scale factor above 31 (more filter
 * taps than the kernel holds).  Asynchronous on `stream` (the geometry is staged through a small ring of pinned
 * buffers; the call never drains the stream), so a batch can be resized while the previous one is being scored. */
int mcm_resize_crop_u8(mcm_handle* h, const uint8_t* const* src_dev_ptrs, const int32_t* heights,
                       const int32_t* widths, int32_t B, uint8_t* dst_dev, void* stream);

/* Host side of the raw-image ingest: copies n images (host pointers srcs[i], sizes[i] bytes) to dst + offsets[i] — the ONE
 * pinned buffer a batch is uploaded from with a single asynchronous copy (replaces the loader's per-batch `.cuda()` of
 * reference utils/detection_util.py:222-223) — with `threads` native threads, the work cut by bytes.  Pure host code;
 * MCM_ERANGE when an image does not fit dst_bytes. */
int mcm_pack_u8(const uint8_t* const* srcs, const int64_t* sizes, const int64_t* offsets, int32_t n, uint8_t* dst,
                int64_t dst_bytes, int32_t threads);

/* JPEG ingest, host half (csrc/jpeg_entropy.cpp): file -> markers -> Huffman decode -> quantised DCT coefficients, written by
 * `threads` native threads into dst (the pinned buffer a batch is uploaded from) — the part of the reference loader's
 * `Image.open(path).convert("RGB")` (torchvision ImageFolder, utils/train_eval_util.py:96-146) that is bit-serial; the
 * device half (mcm_jpeg_reconstruct) turns the coefficients into the RGB pixels libjpeg would have produced.
 * Baseline / extended-sequential and progressive Huffman JPEGs are taken.
 * meta[i].status: 0 taken; 1 a JPEG this path does not take (arithmetic, 12-bit, CMYK / RGB-coded, sequential multi-scan,
 * sampling other than 4:4:4 / 4:2:2 / 4:2:0): decode it with the fallback decoder; 2 unreadable, corrupt, or merely
 * suspicious — a scan is taken only if every RSTn and the EOI sit exactly where a clean scan has them and every dequantised
 * coefficient is plausible for 8-bit samples; libjpeg's warn-and-recover output for anything else is the fallback's to give.
 * Component c of image i: int16 [hb][wb][64] coefficients in natural order at dst + coef_off[c]; its quantisation table at
 * quant[(i * 3 + c) * 64].  *bytes_used = bytes of dst the batch needs; MCM_ERANGE (nothing decoded) when dst_bytes is less. */
typedef struct mcm_jpeg_image {
  int32_t status;
  int32_t width, height, ncomp;
  int32_t hs[3], vs[3];   /* sampling factors */
  int32_t wb[3], hb[3];   /* blocks per row / per column of the coefficient planes (whole MCUs) */
  int64_t coef_off[3];
} mcm_jpeg_image;
int mcm_jpeg_entropy_decode(const char* const* paths, int32_t n, void* dst, int64_t dst_bytes, mcm_jpeg_image* meta,
                            uint16_t* quant, int32_t threads, int64_t* bytes_used);

/* JPEG ingest, device half (csrc/jpeg.hip): the coefficients of a batch (uploaded as they were written by
 * mcm_jpeg_entropy_decode; coef_dev = device copy of its dst) -> dequantisation, libjpeg's default inverse DCT
 * (jpeg_idct_islow), fancy chroma upsampling and YCbCr -> RGB: uint8 [height_i, width_i, 3] at rgb_dev + rgb_offsets[i] —
 * byte for byte what Pillow's Image.open(path).convert("RGB") returns (the reference's loader decodes with it).  meta / quant /
 * rgb_offsets are HOST arrays of length n; images whose status is not 0 are skipped (the caller decodes those with its
 * fallback and writes their pixels itself).  Asynchronous on `stream`; the output is what mcm_resize_crop_u8 takes. */
int mcm_jpeg_reconstruct(mcm_handle* h, const void* coef_dev, const mcm_jpeg_image* meta, const uint16_t* quant, int32_t n,
                         uint8_t* rgb_dev, const int64_t* rgb_offsets, void* stream);

/* Prompt-ensemble bank (BASELINE config 5): feats_dev = unit-norm text
 * features [K*T, proj_dim], class-major (row k*T + t = template t of class k), as written by
 * mcm_encode_text; bank_dev [K, proj_dim] = normalise(mean over the T templates).  The reference
 * ships the 80 templates (utils/imagenet_templates.py) but never calls them; this is the
 * standard CLIP zero-shot recipe. */
int mcm_reduce_bank(mcm_handle* h, const float* feats_dev, int32_t K, int32_t T, float* bank_dev,
                    void* stream);

/* Detection metrics on the device: replaces get_measures /
 * fpr_and_fdr_at_recall, reference utils/detection_util.py:66-119, incl. the sklearn
 * roc_auc_score / average_precision_score calls at :115-116.  pos_dev [n_pos] = scores of the
 * ID set (the positive class, :112-113), neg_dev [n_neg] = scores of one OOD set, both device
 * fp32; negate != 0 evaluates on -score, which is how the reference calls it
 * (get_and_print_results, :255: the stored scores are negated confidences).
 * out_host[0..2] = AUROC, AUPR, FPR at the operating point whose recall is closest to
 * recall_level (0.95 in the reference), same tie rules as the reference.  Synchronises `stream`.
 * Scratch (12 bytes per score) is borrowed from the activation workspace mcm_create sized (the B/16
 * batch-512 workspace holds 25 million scores); larger inputs get a one-off stream-ordered allocation
 * (hipMallocAsync / hipFreeAsync on `stream`).
 * STREAM REQUIREMENT: because the scratch is the workspace of the encode / score calls, `stream` must be
 * the stream those calls were issued on (or be ordered after them by an event): stream order is the only
 * thing that keeps mcm_measures from overwriting a tower's activations in flight. */
int mcm_measures(mcm_handle* h, const float* pos_dev, int64_t n_pos, const float* neg_dev,
                 int64_t n_neg, int32_t negate, double recall_level, double* out_host, void* stream);

/* Fixed-edge histogram of a device score vector: the constant-size per-rank payload `north_star` names
 * ("RCCL all-gather of per-shard score histograms"), summed over ranks by the caller's all-reduce.
 * edges_dev: fp32 [n_bins + 1] ascending; counts_dev: int64 [n_bins], overwritten.  numpy.histogram
 * semantics (bin i = [e_i, e_{i+1}), last bin closed, out-of-range scores dropped); n_bins <= 8192. */
int mcm_score_histogram(mcm_handle* h, const float* scores_dev, int64_t n, const float* edges_dev,
                        int32_t n_bins, int64_t* counts_dev, void* stream);

/* ---- Mahalanobis baseline (--score maha) ---------------------------------------
 * mcm_encode_image_raw: HF get_image_features WITHOUT the reference's `/= norm` — what
 * get_mean_prec / get_Mahalanobis_score consume when args.normalize is False (their default),
 * reference utils/detection_util.py:158-161,187-190.
 * mcm_maha_prepare + mcm_maha_score_features replace the per-class loop of get_Mahalanobis_score
 * (:191-198): means_dev [C, proj_dim] fp32 and prec_dev [proj_dim, proj_dim] fp32 are
 * get_mean_prec's classwise_mean and precision; w_dev [C, proj_dim] and k_dev [C] (fp64) are
 * scratch the caller provides, filled by prepare and read by score; scores_dev [B] receives
 * min_c 0.5 (f - mu_c) P (f - mu_c)^T, the value the reference returns per sample. */
int mcm_encode_image_raw(mcm_handle* h, const float* pixels_dev, int32_t B, float* out_dev, void* stream);
int mcm_maha_prepare(mcm_handle* h, const float* means_dev, const float* prec_dev, int32_t C,
                     double* w_dev, double* k_dev, void* stream);
int mcm_maha_score_features(mcm_handle* h, const float* feats_dev, int32_t B, const float* prec_dev,
                            const double* w_dev, const double* k_dev, int32_t C, float* scores_dev,
                            void* stream);

/* ---- CLIP byte-level BPE tokenizer, host side --------------------------------
 * Replaces CLIPTokenizer.from_pretrained(args.ckpt) + tokenizer(list[str], padding=True,
 * return_tensors="pt") of reference utils/detection_util.py:216,228.  vocab.json / merges.txt are the
 * checkpoint's tokenizer files.  Pure host code: usable without a GPU. */
typedef struct mcm_tokenizer mcm_tokenizer;
int mcm_tokenizer_create(const char* vocab_json_path, const char* merges_txt_path, mcm_tokenizer** out);
void mcm_tokenizer_destroy(mcm_tokenizer* t);
const char* mcm_tokenizer_last_error(const mcm_tokenizer* t); /* t == NULL: the last create error */
int32_t mcm_tokenizer_vocab_size(const mcm_tokenizer* t);
/* n prompts (UTF-8, NUL-terminated) -> ids_out / mask_out [n, *seq_len_out] int32, row-major,
 * *seq_len_out = longest prompt incl. BOS/EOS; shorter rows are padded with the pad token
 * (<|endoftext|>) and mask 0.  capacity = row capacity of the output buffers (e.g. 77);
 * MCM_ERANGE (with *seq_len_out set) when the longest prompt needs more. mask_out may be NULL. */
int mcm_tokenizer_encode(mcm_tokenizer* t, const char* const* texts, int32_t n, int32_t capacity,
                         int32_t* ids_out, int32_t* mask_out, int32_t* seq_len_out);

/* ---- per-kernel timing (HIP events on the caller's stream) -------------------------
 * When enabled, every kernel launch of the encode path is bracketed by a pair of
 * pre-created hipEvents.  mcm_profile_read synchronises the stream, accumulates the
 * elapsed times per kernel class and returns them; it resets the accumulators.
 * Classes: see MCM_KC_*. */
#define MCM_KC_PATCHIFY 0
#define MCM_KC_GEMM 1
#define MCM_KC_LAYERNORM 2
#define MCM_KC_ATTENTION 3
#define MCM_KC_POOL_PROJECT 4
#define MCM_KC_SCORE 5
#define MCM_KC_EMBED 6
/* sub-classes of MCM_KC_GEMM (ABI 3): the whole-batch GEMM shapes of an encoder layer, counted in MCM_KC_GEMM as well */
#define MCM_KC_GEMM_QKV 7
#define MCM_KC_GEMM_OUTPROJ 8
#define MCM_KC_GEMM_FC1 9
#define MCM_KC_GEMM_FC2 10
#define MCM_KC_COUNT 11
int mcm_profile_enable(mcm_handle* h, int32_t on);
/* ms_out[MCM_KC_COUNT], launches_out[MCM_KC_COUNT], flops_out[MCM_KC_COUNT] (algorithmic
 * FLOP issued by that class since the last read; GEMM counts 2*M*N*K of the *logical*
 * problem, not the padded tile grid). */
int mcm_profile_read(mcm_handle* h, double* ms_out, int64_t* launches_out, double* flops_out);

/* ---- operator-level entry points (same kernels the towers launch; used by the parity
 * tests to pin each kernel against the oracle, and usable as building blocks) --------
 * All tensors row-major.  `prec` = MCM_PREC_*: in BF16 mode activations/weights that
 * feed MFMA are bf16 (uint16 storage), in F32 mode they are fp32. */

/* y[M,N] = x[M,K] · w[N,K]^T + bias[N]  with epilogue `epi`:
 *   0: store (+bias)                out dtype = operand dtype
 *   1: QuickGELU(acc+bias)          out dtype = operand dtype   (activations.py:117-123)
 *   2: resid[M,N] (fp32) += acc+bias   in place; `y` ignored
 * M ≥ 1 arbitrary (ragged tiles are clamped/masked); K % 64 == 0 (bf16) or K % 32 == 0
 * (fp32): one K-step is 128 bytes per row; N % 16 == 0. */
int mcm_op_linear(mcm_handle* h, int32_t prec, const void* x_dev, const void* w_dev,
                  const float* bias_dev, void* y_dev, float* resid_dev, int32_t M, int32_t N,
                  int32_t K, int32_t epi, void* stream);
/* mcm_op_linear with flags: bit 0 = w_dev is the split image [N, 2K] written by mcm_op_split_weight (16-bit modes;
 * K % 64 == 0): y = x (W_hi + W_lo)^T, both products in one fp32 accumulator chain. */
#define MCM_LINEAR_SPLIT_W 1
/* bit 1 (fp16 mode, K % 64 == 0): x_dev is the SPLIT image [M, 2K] of a logical [M, K] activation — per 64 columns
 * hi[64] = round(x) then lo[64] = round(x - hi), the layout mcm_op_split_weight writes for M rows — y = (X_hi + X_lo) W^T;
 * bit 2 (fp16 mode, epilogues 0 / 1, N % 64 == 0): y_dev is written as a split image [M, 2N] (QuickGELU in its exact form).
 * Together they are the GEMMs of the split-activation arm (mcm_score_x2). */
#define MCM_LINEAR_SPLIT_X 2
#define MCM_LINEAR_SPLIT_OUT 4
int mcm_op_linear_ex(mcm_handle* h, int32_t prec, const void* x_dev, const void* w_dev,
                     const float* bias_dev, void* y_dev, float* resid_dev, int32_t M, int32_t N,
                     int32_t K, int32_t epi, int32_t flags, void* stream);
/* fp32 w [N,K] (device) → split operand image [N, 2K] in the 16-bit dtype of `prec` (K % 64 == 0). */
int mcm_op_split_weight(mcm_handle* h, int32_t prec, const float* w_dev, int32_t N, int32_t K,
                        void* out_dev, void* stream);
/* LayerNorm over the last dim (modeling_clip.py:370,379,605,608): x fp32 [M,D] →
 * y (operand dtype of `prec`, or fp32 when out_f32 != 0) [M,D]. */
int mcm_op_layernorm(mcm_handle* h, int32_t prec, const float* x_dev, const float* gamma_dev,
                     const float* beta_dev, void* y_dev, int32_t M, int32_t D, float eps,
                     int32_t out_f32, void* stream);
/* The split-activation arm's LayerNorm and attention (fp16): y [M, 2D] / out [rows, 2 heads 64] are split images (per 64
 * columns hi[64] then lo[64]); mcm_op_attention_split reads qkv as a split image [rows, 6 heads 64]. */
int mcm_op_layernorm_split(mcm_handle* h, const float* x_dev, const float* gamma_dev, const float* beta_dev,
                           void* y_dev, int32_t M, int32_t D, float eps, void* stream);
int mcm_op_attention_split(mcm_handle* h, const void* qkv_dev, void* out_dev, int32_t nseq, int32_t seq_len,
                           int32_t heads, void* stream);
/* Multi-head SDPA (modeling_clip.py:259-277,313-331): qkv [nseq*seq_len, 3*heads*64]
 * packed as [q | k | v], head_dim 64, scale 0.125; causal != 0 for the text tower;
 * seq_len ≤ 288.  out [nseq*seq_len, heads*64]. */
int mcm_op_attention(mcm_handle* h, int32_t prec, const void* qkv_dev, void* out_dev,
                     int32_t nseq, int32_t seq_len, int32_t heads, int32_t causal,
                     void* stream);

/* fp16 saturation watch.  In MCM_PREC_F16 every 16-bit activation write saturates at +-65504 instead of
 * overflowing to inf (MODE.FP16_OVFL) — which keeps a row finite but costs that element's precision.  So that
 * this never happens unnoticed (a real checkpoint has outlier channels the seeded weights do not), the GEMM
 * epilogues and the LayerNorm that write fp16 activations track the largest magnitude they packed and bump a
 * sticky per-handle device counter once per wave that packed a saturating value (|v| >= 65520).
 * mcm_saturation_count: *count_host = events since the last reset (0 = no activation of any call left the fp16
 * range); synchronises `stream`; reset != 0 clears the counter.  mcm_saturation_check(h, 0) stops the kernels
 * from reporting (the tracking itself is a handful of VALU instructions per tile and is always compiled in).
 * Always 0 in MCM_PREC_BF16 / MCM_PREC_F32 (fp32 exponent range). */
int mcm_saturation_check(mcm_handle* h, int32_t on);
int mcm_saturation_count(mcm_handle* h, int32_t reset, uint64_t* count_host, void* stream);

/* Kernel faults (ABI 4).  The persistent attention kernel (one workgroup per CU: loader waves and compute waves that wait for each
 * other through LDS counters) bounds every wait: a wave that has polled 2^22 times (three orders of magnitude above the kernel's
 * whole run time) stops its workgroup and stores 1 to a sticky per-handle word in host-mapped memory, so that a protocol slip ends
 * the launch with wrong rows instead of holding the GPU until a watchdog fires.  From then on every compute call on the handle
 * returns MCM_EHIP (mcm_last_error says why): the handle must be destroyed.  mcm_kernel_faults reads the word without
 * synchronising (non-zero = faulted; for a definitive answer synchronise the stream first); 0 in every correct run. */
int mcm_kernel_faults(const mcm_handle* h);

#ifdef MCM_HARNESS
/* libmcm_hip_harness.so only (built with -DMCM_HARNESS next to the shipped library; loaded by the A/B tests
 * and tools, never by the product path).  Process-wide switches.
 * GEMM variant: -1 the shipped size policy (64x128 / 128x128 tile kernels, persistent 256x256, ping-pong),
 * 0 = the 128x128 tile kernel always, 11 = the 64x128 tile kernel always, 3/4 = persistent 256x256 2-stage (4: counted
 * epilogue stores), 5 = persistent 256x256 ping-pong (problems whose M and N are multiples of 256; others run as 3),
 * 9 = the flagged ping-pong text of gemm_arms.hpp with every flag off (honours mcm_debug_gemm_group_n; whole tiles, others as 5).
 * Removed in round 6 after measuring negative (EXPERIMENTS.md "Removed arms"): 1/2 persistent 256x128 3-stage, 6 the ping-pong
 * loop on 32x32x16 MFMAs, 7 balanced DMA, 8 staggered epilogues.  Returns MCM_OK, or MCM_EINVAL for an unknown variant. */
int mcm_debug_gemm_variant(int32_t variant);
/* 16-bit attention kernel: 1 = the shipped policy (the transpose-read kernel; its persistent form at the B/16 shape from 16 jobs
 * per CU on), 0 = the round-1 kernel, 10 = XCD-aware deal of the (sequence, head) workgroups, 11 = the q-blocks dealt to the waves
 * rotated per workgroup (SIMD balance; bit-identical, no gain), 21 = the persistent form at every size and query count of the
 * 13-tile shape, 36 = the 8-wave kernel at every size.  Anything else: MCM_EINVAL (the arms 2 - 9, 12 - 20, 22 - 35 of rounds
 * 2 - 5 were removed in round 6: EXPERIMENTS.md "Removed arms"). */
int mcm_debug_attention_variant(int32_t variant);
/* Polls a wait of the persistent attention kernel may take before it gives up (the shipped library always passes 2^22);
 * 0 = every wait that has to wait gives up at once.  mcm_debug_clear_faults resets the handle's sticky fault word.  Both exist
 * for the test of the fault path only. */
int mcm_debug_attn_spin_budget(int64_t polls);
int mcm_debug_clear_faults(mcm_handle* h);
/* mcm_op_attention with the two launch parameters only the model sets: query rows (0 = all; the CLS-only last layer passes 1)
 * and the walk direction (reverse != 0: jobs in descending order). */
int mcm_debug_op_attention(mcm_handle* h, int32_t prec, const void* qkv_dev, void* out_dev, int32_t nseq, int32_t seq_len,
                           int32_t heads, int32_t causal, int32_t qrows, int32_t reverse, void* stream);
/* A/B and ablation bits of the GEMM kernels (gemm.hip, GemmArgs::dbg; 0 = shipped behaviour). */
int mcm_debug_gemm_dbg(int32_t bits);
/* A/B: run the QKV projection + attention of every layer per chunk of the batch (n chunks; 1 = shipped). */
int mcm_debug_qkv_chunks(int32_t n);
/* A/B: the wide store GEMMs (QKV projection, fc1) as n launches over column blocks of N / n (1 = shipped: one launch).
 * The W re-fetch experiment of DESIGN.md / EXPERIMENTS.md: only N / n of W is live in an XCD's L2 per launch. */
int mcm_debug_nsplit(int32_t n);
/* A/B: the persistent kernels walk their N tiles in groups of gn (0 = plain n-fastest walk, the shipped behaviour): only gn
 * N-tiles of W are live in an XCD's L2 at a time.  Honoured by gemm_p256_kernel and by the arms text of the ping-pong kernel
 * (variant 9); the shipped ping-pong kernel walks plainly. */
int mcm_debug_gemm_group_n(int32_t gn);
/* A/B: 1 (default, shipped) = the patch-embedding GEMM gathers its A operand from the fp32 NCHW pixels itself; 0 = the
 * rounds 1 - 3 route (patchify writes a patch matrix, the GEMM reads it back).  Bit-identical. */
int mcm_debug_patch_fold(int32_t on);
/* 0 (shipped): mcm_resize_crop_u8 stages the source window in LDS where it fits; 1: the per-pixel fused form of
   rounds 2 - 3 for every workgroup (A/B). */
int mcm_debug_resize_fused_only(int32_t on);
/* 0 (shipped): the persistent GEMM kernels launch one workgroup per CU; n (a multiple of 8): n workgroups, so that two
   handles on two streams can share the chip (tools/dual_stream_probe.py, removed in round 6: git history). */
int mcm_debug_persistent_grid(int32_t n);
/* A/B: 1 = the LayerNorms of the vision tower between a residual GEMM and its consumer folded into the two GEMM
 * epilogues (16-bit modes, widths that are multiples of 256; bit-identical for every batch size); 0 (default, the
 * shipped behaviour) = every LayerNorm as its own launch.  Measured 1 % slower end to end, DESIGN.md 5.5. */
int mcm_debug_ln_fold(int32_t on);
/* A/B: 1 = the 16-bit towers hand q / k / v from the QKV projection to attention head-major ([3 heads][rows][64]: an
 * attention workgroup's rows are consecutive bytes); 0 (default, shipped) = [rows][3 D].  Bit-identical; no net gain. */
int mcm_debug_qkv_head_major(int32_t on);
/* A/B: 1 = the LayerNorm behind a whole-batch out-proj / fc2 of a 16-bit vision tower is computed in the tail of that
 * GEMM's kernel by the waves that have run out of tiles (same bits as the LayerNorm launch); 0 (default, shipped) =
 * every LayerNorm is its own launch.  Measured equal at ViT-B/16 batch 512, slower on smaller problems, DESIGN.md 5.5. */
int mcm_debug_ln_tail(int32_t on);
/* A/B (round 6): 1 = the LayerNorm behind a whole-batch out-proj / fc2 of a 16-bit vision tower is written by that GEMM's own
 * EPILOGUE from the accumulator registers: the N / 256 workgroups that hold the tiles of one 256-row panel act as one
 * full-row tile — row moments exchanged through an XCD's L2, no re-read of the residual stream, no LayerNorm launch
 * (gemm_arms.hpp "LNC"; the full-row epilogue of VERDICT r5 item 2).  Scores agree with the shipped path to fp32 round-off
 * (Chan-combined slot moments instead of whole-row two-pass statistics), not bit for bit.  0 (default) = shipped behaviour.
 * Waits that gave up are counted by mcm_debug_ln_tail_timeouts. */
int mcm_debug_ln_cluster(int32_t on);
/* LNC, second form (R6.4): polls < 0 (default) = a wave WAITS for its row panel's partner workgroups (the first form); polls >= 0 =
 * the DEFER form: after that many polls of the partners' counter a wave leaves its 128 x 64 segment's LayerNorm to a clean-up
 * launch behind the GEMM (one mask word per panel half says which segments are owed; same arithmetic, same bits) and goes on
 * to its next tile.  mcm_debug_ln_cluster_deferred: segments normalised by the clean-up launches since mcm_create. */
int mcm_debug_ln_cluster_spin(int32_t polls);
/* A/B (round 6, R6.7): 1 = out-proj / fc2 of a whole-batch 16-bit vision layer run as 64-row FULL-ROW tiles (one workgroup holds
 * 64 x 768 / 1024 outputs) whose epilogue writes x once and the LayerNorm output from the accumulator registers — the literal
 * design of VERDICT r5 item 2, no cross-workgroup traffic (gemm_arms.hpp ROW64).  Round-off-equal to the shipped path.
 * 2 = three W stages per wave (N = 768); 3 / 4 = as 1 / 2 on a piece-contiguous copy of the weight, built on first use. */
int mcm_debug_ln_row(int32_t on);
int mcm_debug_ln_cluster_deferred(mcm_handle* h, uint64_t* count_host);
/* Tickets of the LayerNorm tail that gave up waiting for their rows (a bounded spin: wrong rows rather than a hung
 * device); 0 in a correct run.  Synchronises the device. */
int mcm_debug_ln_tail_timeouts(mcm_handle* h, uint64_t* count_host);
#endif

#ifdef __cplusplus
}
#endif
#endif /* MCM_H_ */
