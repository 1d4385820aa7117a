/*
 * libmi355_retrieval — C ABI of the MI355X-native embed-then-rank hot path.
 *
 * Drop-in boundary.  The reference (vitasoftAI/ImageRetrievalResearch) has no
 * FFI: its "operator interface" for this path is the timm model object plus a handful of torch
 * calls.  Each entry point below names the reference call it replaces (file:line relative to the
 * reference checkout).  Plain pointers and sizes only; no torch types.  All pointers are DEVICE
 * pointers on the current HIP device unless a parameter says "host".  `stream` is a hipStream_t
 * passed as void* (NULL = the null stream); every launch goes on that stream and nothing here
 * synchronises, so torch's `.item()` / `.cpu()` order correctly behind it.
 *
 * Error model: every function returns 0 on success, nonzero otherwise, and never aborts;
 * mi355_last_error() returns a thread-local message for the last failure.
 *
 * The Python host side (imageretrievalresearch_amd/) binds exactly these symbols via ctypes; the
 * binding a reference maintainer would add is shown in INTEGRATION.md.
 */
#ifndef MI355_RETRIEVAL_H
#define MI355_RETRIEVAL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_ABI_VERSION 3

/* ------------------------------------------------------------------ library / errors */
int mi355_abi_version(void);
const char* mi355_last_error(void);
/* Number of visible HIP devices (0 when there is no GPU); never fails. */
int mi355_device_count(void);

/* ------------------------------------------------------------------ synthetic data
 * Portable counter-based generator, bit-identical to imageretrievalresearch_amd/synth.py.
 * kind 0 = uniform [0,1), 1 = unit normal (Irwin-Hall 4).  out[i] = f(seed, offset + i). */
int mi355_synth_fill(float* out, int64_t n, uint64_t seed, int64_t offset, int kind, void* stream);

/* ------------------------------------------------------------------ rank: cosine + top-k
 * Replaces torch.nn.CosineSimilarity(dim=1, eps=1e-6) + torch.topk at
 *   train/train.py:250-251, :345-356 ; inference/inference.py:226-242 ; notebook raw :231-251.
 * Semantics (pinned): score[q][g] = sum_d (Q[q][d]/max(|Q[q]|,eps)) * (G[g][d]/max(|G[g]|,eps)),
 * fp32 throughout (exact-f32 MFMA); top-k sorted by descending score, ties -> lower index first; a NaN score orders as
 * the largest value (as torch.topk does), two NaNs tie. */

/* out[r][:] = in[r][:] / max(||in[r]||_2, eps); in == out allowed.  rows x dim fp32 row-major. */
int mi355_l2_normalize_rows(const float* in, float* out, int64_t rows, int dim, float eps, void* stream);

/* Bytes of scratch mi355_rank_topk needs for (Q, G, D, k). */
size_t mi355_rank_workspace_bytes(int64_t Q, int64_t G, int dim, int k);

/* All-pairs cosine + top-k.
 *   queries  [Q][dim] fp32 (raw, normalised internally)
 *   gallery  [G][dim] fp32; gallery_is_normalized != 0 promises rows already went through
 *            mi355_l2_normalize_rows (the resident-gallery fast path), else norms are applied here
 *   out_val  [Q][k] fp32, out_idx [Q][k] int64 (index into gallery + idx_offset)
 *   idx_offset: added to every index (global row of this shard's first row)
 * Arithmetic: fp32 in, fp32 accumulation, fp32 scores.  For Q > 4 and 16-byte aligned gallery rows the products run on the
 * bf16 matrix pipe as a three-way split of each fp32 operand (six bf16 x bf16 products per element, each exact in the
 * fp32 accumulator; dropped terms <= 2^-23 of a product - one fp32 rounding); the environment variable
 * MI355_RANK_EXACT_F32=1 selects the exact fp32 MFMA (an fmaf chain) instead.  A score depends on its query row and
 * gallery row only (not on Q, G, tiles or shards), so sharded and unsharded results are bit-identical.
 * Errors: k < 1, k > G, Q < 1, dim < 1, workspace too small. */
int mi355_rank_topk(const float* queries, int64_t Q, const float* gallery, int64_t G, int dim,
                    int gallery_is_normalized, int k, float eps, int64_t idx_offset,
                    float* out_val, int64_t* out_idx, void* workspace, size_t workspace_bytes,
                    void* stream);

/* Scores only: out[Q][G] fp32 cosine matrix (same kernel as above without the selection). */
/* Prepared gallery for RESIDENT galleries (the reference re-reads and re-normalises its gallery for every query,
 * train/train.py:250; here it is normalised once when rows are added, and - optionally - split once into the three bf16 planes
 * the cosine GEMM multiplies, stored in the GEMM's fragment order: 6 B per element, mi355_gallery_planes_bytes(G, dim) bytes).
 * mi355_rank_topk_prepared then does no per-call work on the gallery side at all; k <= 8 and Q > 4 (other shapes: mi355_rank_topk
 * with the fp32 rows).  Values and indices are bit-identical to mi355_rank_topk(gallery_is_normalized = 1) on the same rows. */
size_t mi355_gallery_planes_bytes(int64_t G, int dim);
int mi355_gallery_prepare(const float* gallery_normalized, int64_t G, int dim, void* planes, size_t planes_bytes, void* stream);
int mi355_rank_topk_prepared(const float* queries, int64_t Q, const void* gallery_planes, int64_t G, int dim, int k, float eps,
                             int64_t idx_offset, float* out_val, int64_t* out_idx, void* workspace, size_t workspace_bytes,
                             void* stream);

int mi355_cosine_scores(const float* queries, int64_t Q, const float* gallery, int64_t G, int dim,
                        int gallery_is_normalized, float eps, float* out, void* workspace,
                        size_t workspace_bytes, void* stream);

/* Row-wise top-k of an explicit score matrix scores[Q][G] (torch.topk, train/train.py:251).
 * workspace: mi355_rank_workspace_bytes(Q, G, 0, k). */
int mi355_topk_rows(const float* scores, int64_t Q, int64_t G, int k, int64_t idx_offset,
                    float* out_val, int64_t* out_idx, void* workspace, size_t workspace_bytes,
                    void* stream);

/* Merge per-shard candidates: cand_val/cand_idx [Q][ncand] (ncand = shards*k, any
 * order) -> global top-k with the same ordering rule.  Used after the RCCL all-gather. */
int mi355_merge_topk(const float* cand_val, const int64_t* cand_idx, int64_t Q, int ncand, int k,
                     float* out_val, int64_t* out_idx, void* workspace, size_t workspace_bytes,
                     void* stream);

/* Sharded search (no reference counterpart: inference/inference.py:271 is single-device).  One int32 tensor
 * per rank goes through the candidate all-gather: packed[q][j] = {bits of the f32 score, LOCAL row index}, slots
 * j >= kk (a shard with fewer than k rows) = {-inf, -1}.  val / idx: (Q, kk) results of mi355_rank_topk on the shard. */
int mi355_pack_candidates(const float* val, const int64_t* idx, int64_t Q, int kk, int k, int32_t* packed, void* stream);
/* Merge of the all-gathered lists packed[world][Q][k][2]: adds shard_offsets[r] (device int64[world]) to rank r's local
 * indices and selects the k best of the world * k candidates of every query (higher score, then lower global index):
 * (Q, k) values + int64 global indices, identical to ranking against the unsharded gallery. */
size_t mi355_merge_packed_workspace_bytes(int64_t Q, int world, int k);
int mi355_merge_packed_topk(const int32_t* packed, const int64_t* shard_offsets, int world, int64_t Q, int k,
                            float* out_val, int64_t* out_idx, void* workspace, size_t workspace_bytes, void* stream);

/* Row-wise pair cosine, inference/inference.py:226,229: out[i] = cos(a[i], b[i]). */
int mi355_pair_cosine(const float* a, const float* b, int64_t rows, int dim, float eps, float* out,
                      void* stream);

/* utils/contrastive_loss.py:36-61 ContrastiveLoss.forward(fm1, fm2, label, mean):
 *   dis = sum_d (fm2-fm1)^2 ; 0.5*(label*dis + (1-label)*relu(margin - sqrt(dis+1e-9))^2) ;
 *   out[0] = mean or sum over rows (deterministic order).  per_row (optional, may be NULL) gets
 *   the per-row losses. */
int mi355_contrastive_loss(const float* fm1, const float* fm2, int64_t rows, int dim, float label,
                           float margin, int mean, float* out, float* per_row, void* stream);

/* torch.nn.CosineEmbeddingLoss(margin)(x1, x2, target) with a scalar target of +1 or -1 broadcast over the rows —
 * the validation-step loss of train/train.py:214-216, :308-310.  out[0] = mean (or sum). */
int mi355_cosine_embedding_loss(const float* x1, const float* x2, int64_t rows, int dim, float target, float margin,
                                int mean, float* out, void* stream);

/* Hit counting, train/train.py:252-255: counts[0] += #queries whose class equals the class of
 * their top-1 result, counts[1] += #queries whose class is among their top-min(3,k).
 * idx [Q][k] int64 into gallery_cls [G]; counts int64[2] must be zeroed by the caller.  An index outside [0, G)
 * (the pad entry of a list with fewer than k real candidates) counts as a miss. */
int mi355_hit_counts(const int64_t* idx, int64_t Q, int k, const int64_t* query_cls,
                     const int64_t* gallery_cls, int64_t G, int64_t* counts, void* stream);

/* Notebook variant, inference/training_analysis.ipynb raw :240-251: walk each ranked list and keep
 * the first n (<= 8) DISTINCT classes.  out_cls/out_idx [Q][n] int64 (-1 padded), out_val [Q][n].
 * gallery_cls [G]; indices outside [0, G) are skipped. */
int mi355_distinct_class_topn(const int64_t* idx, const float* val, int64_t Q, int k,
                              const int64_t* gallery_cls, int64_t G, int n, int64_t* out_cls, int64_t* out_idx,
                              float* out_val, void* stream);

/* ------------------------------------------------------------------ backbone models
 * Replaces timm.create_model(name, num_classes=N) and the methods the reference calls on it
 * (inference/inference.py:102,110,133,146,199-201 ; train/train.py:194-195,288,396 ;
 *  train/train_efficientnet.py:226,230 ; train/train_vit_triplet.py:354-357).
 * Names: "efficientnet_b3a" ("efficientnet_b3"), "rexnet_150", "rexnet_200",
 *        "swin_base_patch4_window7_224".  Input 224x224 (effnet/rexnet accept any H,W multiple of 32).
 */
typedef struct mi355_model* mi355_model_t;

/* num_classes: 0 = identity classifier (pooled features out), >0 = Linear head of that width. */
int mi355_model_create(const char* name, int num_classes, mi355_model_t* out);
void mi355_model_destroy(mi355_model_t m);

/* Parameter/buffer table in timm-0.4.12 state-dict order.  kind: 0 = trainable parameter,
 * 1 = float buffer (BN running stats), 2 = int64 buffer (num_batches_tracked, relative_position_index). */
int mi355_model_num_tensors(mi355_model_t m);
int mi355_model_tensor_info(mi355_model_t m, int i, const char** name, int* ndim, int64_t shape[4],
                            int* kind);
int mi355_model_feature_dim(mi355_model_t m);   /* D: 1536 / 1920 / 2560 / 1024 */
int mi355_model_num_classes(mi355_model_t m);

/* Hand one state-dict tensor (HOST pointer, fp32, contiguous, numel elements) to the model by its
 * timm key — the load_state_dict half of inference/inference.py:117-124.  int64 buffers are ignored. */
int mi355_model_set_tensor(mi355_model_t m, const char* name, const float* host_data, int64_t numel);

/* One-time pack (SURVEY §5 "checkpoint"): fold eval-mode BN into the conv weights, round to bf16,
 * lay out for the kernels, upload to the current device.  Must be called after the last set_tensor
 * and again whenever weights change. */
int mi355_model_pack(mi355_model_t m, void* stream);

/* forward_features: x [B][3][H][W] fp32 NCHW (device) ->
 *   effnet/rexnet: out [B][D][H/32][W/32] fp32 NCHW (un-pooled, as timm returns it)
 *   swin:          out [B][D] fp32 (timm's swin forward_features pools)
 * forward: -> out [B][num_classes] fp32 logits, or [B][D] pooled features when num_classes == 0.
 * pooled_out (optional, may be NULL): [B][D] fp32 global-average-pooled features (get_fm,
 * train/train.py:84-103) produced on the way. */
int mi355_model_forward_features(mi355_model_t m, const float* x, int B, int H, int W, float* out,
                                 float* pooled_out, void* stream);
int mi355_model_forward(mi355_model_t m, const float* x, int B, int H, int W, float* out,
                        float* pooled_out, void* stream);

/* The same forward with the pre-processing fused into the stem's input load: images
 * [B][h][w][3] uint8 on the device (one size for the batch) -> SquarePad(fill) (utils/square_pad.py:20-36) -> ToTensor
 * (/255) -> Normalize(mean, std: HOST float[3]) (inference/inference.py:48-52) -> optional conv_input (conv_input_w:
 * DEVICE fp32 [3][3][3][3] as Conv2d(3,3,3,1,1,bias=False).weight, NULL = none) + SiLU (inference/inference.py:101-105)
 * -> stem, in ONE kernel: no fp32 NCHW batch is written.  Output as mi355_model_forward (features_only = 0) or
 * mi355_model_forward_features (1) for an S x S input, S = max(h, w).  Bit-identical to running the separate
 * mi355_square_pad_normalize / mi355_conv_input_silu / forward chain.  Conv backbones only (efficientnet, rexnet). */
int mi355_model_forward_u8(mi355_model_t m, const unsigned char* images, int B, int h, int w, int fill,
                           const float* mean, const float* stdv, const float* conv_input_w, int features_only,
                           float* out, float* pooled_out, void* stream);

/* Debug/parity tap: copy the bf16 NHWC activation the executor produced for layer `tap_name`
 * (e.g. "stem", "blocks.1.0") during the LAST forward into out as fp32 NCHW.  Taps are recorded
 * only after mi355_model_enable_taps(m, 1). */
int mi355_model_enable_taps(mi355_model_t m, int enable);
int mi355_model_read_tap(mi355_model_t m, const char* tap_name, float* out, int64_t out_numel,
                         int64_t shape[4], void* stream);

/* Parity tool: run ONLY the layers behind tap `from_tap` up to and including the one that records tap `to_tap` on an
 * activation supplied by the caller: x [B][C][h][w] fp32 NCHW on the device (rounded to bf16 on the way in; feed it the
 * oracle's bf16-rounded tap of the previous layer).  Taps are recorded as in a normal forward (enable them first), so a
 * layer is compared with the oracle on the ORACLE's input and errors do not compound through the network. */
int mi355_model_run_between_taps(mi355_model_t m, const char* from_tap, const char* to_tap, const float* x, int B,
                                 int C, int h, int w, void* stream);

/* Algorithmic HBM bytes of one forward at batch B (layer-granular model) and the
 * MACs; used by bench.py's roofline. */
int mi355_model_traffic(mi355_model_t m, int B, int H, int W, double* act_bytes, double* weight_bytes,
                        double* macs);

/* Per-kernel-family view of the same model, for the roofline of the dominant kernel.  Arrays of n >= 8
 * doubles indexed by kind: 0 stem, 1 1x1-conv GEMM, 2 depthwise, 3 SE, 4 other, 5 window attention, 6 layernorm,
 * 7 fused expand+depthwise (pairs the executor runs as one kernel; bytes are still the layer-granular model). */
int mi355_model_traffic_kinds(mi355_model_t m, int B, int H, int W, double* bytes_by_kind, double* macs_by_kind, int n);

/* Executor options: "microbatch" (images per pass through the layer plan; 0 = whole batch),
 * "fuse" (1 = run expand+depthwise pairs on whole-image tiles as one LDS-resident kernel; default 1),
 * "fuse_block" (1 = run whole MBConv blocks of the 14x14 / 7x7 stages - expand, depthwise, SE, gated projection, residual -
 *  as ONE kernel per block; default 1), "fuse_block_min_batch" (use it only for batches of at least this many images;
 *  default 192: one workgroup per image needs about a CU per image to win),
 * "fuse_sweep" (1 = run expand+depthwise of the 112x112 .. 28x28 blocks with the row-sweep kernel, MFMA depthwise from an
 *  LDS row window; default 1), "fuse_band" (older band kernel for shapes the row-sweep kernel does not cover),
 * "profile" (1 = bracket every op with hipEvents on the launch stream; resets the accumulators). */
int mi355_model_set_option(mi355_model_t m, const char* key, int64_t value);
/* Accumulated per-kind kernel time (ms) and launch counts since "profile" was enabled; synchronises. */
int mi355_model_profile_read(mi355_model_t m, double* ms_by_kind, int64_t* launches_by_kind, int n);

/* Per-op profile table (plan order): average launch ms since "profile" was enabled, algorithmic bytes at
 * batch B, kind, and a text label (labels: max_ops * label_stride chars).  Returns the number of ops
 * (callers size the arrays with max_ops >= that; 1024 is always enough).  Developer/bench tool. */
int mi355_model_profile_ops(mi355_model_t m, int B, int H, int W, int max_ops, double* avg_ms, double* bytes,
                            int* kinds, char* labels, int label_stride);

/* Diagnosis of the whole-MBConv-block kernel (option "block_stamps" = 1): per-phase shader-cycle counts of the LAST forward,
 * averaged over its images.  out[op][16] in plan order (rows of ops that did not run as a block kernel stay 0); the 16
 * buckets are listed at the end of k_mbconv_block (csrc/mbconv_block.hip).  Synchronises the device.  Returns the number
 * of ops (size out with 16 * max_ops doubles, max_ops >= that; 1024 is always enough) or a negative error.  Developer tool. */
int mi355_model_block_stamps(mi355_model_t m, double* out, int max_ops);

/* timm ClassifierHead / get_fm on an un-pooled map (train/train.py:84-103 get_fm; :194-195 fm = forward_features(x);
 * lbl = model.head(fm)): fm [B][C][HW] fp32 NCHW (device) -> pooled_out (optional) [B][C] fp32 global average ->
 * out [B][N] = Linear(weight [N][C] fp32 device, bias [N] or NULL) on the bf16-rounded pooled features with bf16-rounded
 * weights and fp32 accumulation (the rounding points of the in-model classifier).  weight NULL: pooling only. */
int mi355_pool_linear(const float* fm, int B, int C, int HW, const float* weight, const float* bias, int N, float* out,
                      float* pooled_out, void* stream);

/* Stand-alone 1x1-conv / linear kernel (the model executor's GEMM): out[M][N] bf16 = act(A[M][K] bf16 * W^T + bias).
 * W is bf16 [ceil16(N)][ldw] with ldw = K rounded up to 32, zero padded; bias fp32 [ceil16(N)]; K, N multiples of 8.
 * act: 0 none, 1 SiLU, 2 ReLU, 3 ReLU6, 4 GELU, 5 sigmoid. */
int mi355_gemm_bf16(const void* A, const void* W, const float* bias, void* out, int M, int N, int K, int ldw, int act,
                    void* stream);

/* Inference pre-processing: SquarePad(fill) -> ToTensor -> Normalize, utils/square_pad.py:20-36 +
 * inference/inference.py:48-52.  img: uint8 RGB, HWC (h, w, 3) on the device; mean/std: HOST float[3];
 * out: fp32 (3, S, S) with S = max(h, w), i.e. one image slot of the model's NCHW input batch. */
int mi355_square_pad_normalize(const unsigned char* img, int h, int w, int fill, const float* mean, const float* stdv,
                               float* out, void* stream);

/* Training-time resize: transforms.Resize((out_h, out_w)) of train/train.py:48-50 applied to a PIL
 * image, i.e. PIL.Image.resize((out_w, out_h), BILINEAR) — Pillow's two-pass antialiased resample (Resample.c, 8-bit
 * path), reproduced bit-exactly.  img / out: uint8 RGB HWC on the device; tmp: device scratch of h * out_w * 3 bytes,
 * needed only when both sides change (may be NULL otherwise). */
int mi355_resize_bilinear_u8(const unsigned char* img, int h, int w, unsigned char* out, int out_h, int out_w,
                             unsigned char* tmp, void* stream);

/* Score booster (utils/score_booster.py:1-37) over n fp32 scores on the device.
 * mode 0: cos_sim_score_with_threshold (score >= threshold ? (s+eps)/(eps+alpha) : |(s+alpha/eps)/(2 eps)|),
 * mode 1: cos_sim_score_booster(mode="for_pos"), mode 2: mode="for_neg".  out may alias scores. */
int mi355_score_boost(const float* scores, int64_t n, float eps, float alpha, float threshold, int mode, float* out,
                      void* stream);

/* conv_input pre-stem (inference/inference.py:103-105): out = SiLU(Conv2d(3,3,3,1,1,bias=False)(x)),
 * x/out [B][3][H][W] fp32 NCHW, w [3][3][3][3] fp32 (device). */
int mi355_conv_input_silu(const float* x, const float* w, int B, int H, int W, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355_RETRIEVAL_H */
