// Internal launcher interface between the model executor (model.hip) and the kernel files.
// All tensors are device pointers; activations are NHWC bf16 with the channel count a multiple of 8.
#pragma once
#include "common.h"

namespace mi355 {

// ---- 1x1 conv / linear as a bf16 MFMA GEMM (gemm_bf16.hip) -------------------------------------
//   out[m][n] = act( sum_k A'[m][k] * W[n][k] + bias[n] ) (+ res[m][n])
//   A'[m][k]  = a_relu6 ? relu6(A[m][k] * gate) : A[m][k] * gate,   gate = gate[m / rows_per_img][k] or 1
// W is [Npad][ldw] bf16 with ldw = K rounded up to 32 and zero padding (rows up to Npad = N rounded up to 16),
// bias is fp32 [Npad].
struct GemmArgs {
    const bf16_t* A; int lda;
    const bf16_t* W; int ldw;
    const float* bias;
    const bf16_t* res; int ldr; int res_n;   // residual added to outputs n < res_n (rexnet: first cin channels)
    const float* gate; int gate_ld; int rows_per_img;
    void* out; int ldo; int out_f32;
    int M, N, K;
    int act;
    int a_relu6;
    const bf16_t* zeros;   // >= 64 bytes of device zeros (source of out-of-range DMA chunks); null disables the DMA path
    long M_sel;            // rows of the caller's WHOLE batch (0: use M): decides split-K, so that a chunk of a larger batch rounds like the batch
    float* splitk_ws;      // optional fp32 scratch for the split-K path (gemm_splitk_bytes); null disables it
    size_t splitk_ws_bytes;
    // LayerNorm of the A rows folded into the epilogue (k_gemm_big only): out = rstd[m] * (acc - mean[m] * ln_colsum[n]) + bias[n]
    const float* ln_stats;   // [M][2] = (mean, rstd) of every A row, or null
    const float* ln_colsum;  // [ceil16(N)] column sums of the (gamma-folded) weights
};
int launch_gemm_bf16(const GemmArgs& a, hipStream_t st);
// persistent wide-tile kernel for the compute-bound linears (gemm_wide.hip); launch_gemm_bf16 routes the shapes it supports there
bool gemm_wide_supported(const GemmArgs& a);
int launch_gemm_wide(const GemmArgs& a, hipStream_t st);
// Split-K for the tiny-M late projections (7x7 / 14x14 maps at small batch: a handful of output tiles, each with a serial
// K loop of 20-70 L2 round trips).  Number of 256-deep K chunks the shape is split into, 0 = not a split-K shape.  The
// decision looks at the layer (rows per image, K) and caps M, never at the batch position.
int gemm_splitk_chunks(long M, int rows_per_img, int N, int K);
size_t gemm_splitk_bytes(long M, int N, int K);

// ---- fused expand(1x1, MFMA) -> depthwise -> SE squeeze for whole-image tiles (fused_mbconv.hip) ------------
struct FusedArgs {
    const bf16_t* X;      // [B][H][W][Cin] block input
    const bf16_t* We;     // expand weights, GEMM packing [midPad16][Kp]
    const float* be;      // expand bias [midPad16]
    const bf16_t* Wd;     // depthwise weights [k*k][mid]
    const float* bd;      // depthwise bias [mid]
    bf16_t* D;            // [B][Ho][Wo][mid] depthwise output
    float* pool;          // optional [B][mid]: complete per-channel sums of D (SE squeeze, nblk = 1)
    int H, W, Cin, Kp, mid, Ho, Wo;
    int act_e, act_d;
    int TH;               // band variant: output rows per workgroup (pool is then [B][ceil(Ho/TH)][mid])
    int debug_skip;       // diagnosis only: bit0 skip the expand GEMM phase, bit1 skip the depthwise phase
};
bool fused_late_supported(int H, int W, int Cin, int mid, int k, int stride);
int launch_fused_late(const FusedArgs& a, int B, int k, int stride, hipStream_t st);
int fused_band_rows(int H, int W, int Cin, int mid, int k, int stride);   // 0 = unsupported
int launch_fused_band(const FusedArgs& a, int B, int k, int stride, hipStream_t st);

// ---- row-sweep front half for the early stages (sweep_mbconv.hip): expand -> MFMA depthwise -> D + complete squeeze sums
struct SweepArgs {
    const bf16_t* X;      // [B][H][W][Cin] block input
    const bf16_t* We;     // expand weights, GEMM packing [midPad16][Kp]
    const float* be;      // expand bias [midPad16]
    const bf16_t* Wd;     // depthwise weights [k*k][mid]
    const float* bd;      // depthwise bias [mid]
    bf16_t* D;            // [B][Ho][Wo][mid] depthwise output
    float* pool;          // optional [B][mid]: complete per-channel sums of D (SE squeeze, nblk = 1)
    int B, H, W, Cin, Kp, mid, Ho, Wo;
    int act_e, act_d;
    int csplit;           // workgroups per image (set by the launcher)
    int csplit_override;  // > 0: tuning override
    int variant;          // tuning: alternative band height / occupancy class (0 = default)
    int debug_skip;       // diagnosis only: bit0 skip the expand phase, bit1 skip the depthwise phase, bit2 skip the X loads
    long long* stamps;    // diagnosis only: [B][16] cycle buckets, summed over the image's workgroups (wave 0's view)
};
bool sweep_mbconv_supported(int H, int W, int Cin, int mid, int k, int stride, int act_e, int act_d);
int launch_sweep_mbconv(const SweepArgs& a, int B, int k, int stride, hipStream_t st);

// ---- whole MBConv block for the late stages (mbconv_block.hip): expand -> depthwise -> SE -> gated projection (+ residual)
struct BlockArgs {
    const bf16_t* X;                      // [B][H][W][Cin] block input (also the residual)
    const bf16_t* We; const float* be;    // expand weights [midPad16][Kp] (GEMM packing) + folded-BN bias
    const bf16_t* Wd; const float* bd;    // depthwise weights [k*k][mid] + bias [mid]
    const bf16_t* W1; const float* b1;    // SE reduce [rd][mid] bf16 + bias [rd]
    const bf16_t* W2; const float* b2;    // SE expand, transposed [rd][mid] bf16 + bias [mid]
    const bf16_t* Wp; const float* bp;    // projection weights [CoutPad16][Kp2] + folded-BN bias
    bf16_t* D;                            // scratch [B][Ho*Wo][mid]: depthwise output (round trip through L2)
    bf16_t* Y;                            // [B][Ho][Wo][Cout] block output
    long long* stamps;                    // optional [B][8] phase cycle counts (diagnosis), may be null
    int H, W, Cin, Kp, mid, Ho, Wo, Cout, Kp2, rd;
    int XLD;                              // LDS row stride of the X image (set by the launcher)
    int KCS;                              // k per staged super-chunk of the projection's A operand (set by the launcher)
    int norot;                            // diagnosis: bit0 slabs, bit1 SE FC1, bit2 SE FC2, bit3 projection columns NOT rotated by image
    int variant;                          // tuning (option "block_variant"): bit0 = the round-2 14x14 geometry (two waves share a channel tile, barriers per slab)
    int has_res;
    int res_n;                            // the residual covers output channels [0, res_n) (rexnet: the block's input channels)
    int a_relu6;                          // ReLU6 between the SE gate and the projection (rexnet)
    int act_e, act_d, se_act;
    float inv_hw;
};
bool mbconv_block_supported(int H, int W, int Cin, int mid, int Cout, int k, int stride, int rd, int act_e, int act_d);
int launch_mbconv_block(const BlockArgs& a, int B, int k, int stride, hipStream_t st);

// ---- convolution-side kernels (conv_kernels.hip) -----------------------------------------------
// Stem: x [B][3][H][W] fp32 NCHW -> out [B][Ho][Wo][Cout] bf16, 3x3 stride 2 pad 1, + bias + act.
// w [3*3*3][Cout] fp32 laid out (ky, kx, ci) major, bias fp32 [Cout].
int launch_stem(const float* x, const float* w, const float* bias, bf16_t* out, int B, int H, int W, int Cout,
                int act, hipStream_t st);

// Fused input + stem: img [B][h][w][3] uint8 -> SquarePad(fill) / ToTensor / Normalize (host mean / std) -> optional
// conv_input (conv_w: device fp32 [3][3][3][3], null = none) + SiLU -> stem; out [B][S/2][S/2][Cout] bf16, S = max(h, w).
int launch_stem_u8(const unsigned char* img, int h, int w, int fill, const float* mean, const float* stdv,
                   const float* conv_w, const float* sw, const float* bias, bf16_t* out, int B, int Cout, int act,
                   hipStream_t st);

// Depthwise k x k (k = 3 or 5), stride 1 or 2, pad k/2.  w [k*k][C] bf16, bias fp32 [C].
// pool_partial (optional): [B][dw_pool_blocks(...)][C] fp32 partial sums of the un-rounded output
// (the SE squeeze), reduced in a fixed order.
int dw_pool_blocks(int Ho, int Wo, int C);
// pool_nblk (out): squeeze partials per image this launch produced (pool_partial is [B][nblk][C]).
int launch_dwconv(const bf16_t* in, const bf16_t* w, const float* bias, bf16_t* out, float* pool_partial, int B,
                  int H, int W, int C, int k, int stride, int act, int* pool_nblk, hipStream_t st);

// Squeeze-excite gate: s = (sum over nblk partials) / hw ; r = act1(W1 s + b1) ; gate = sigmoid(W2 r + b2).
// W1 [rd][C] fp32, W2T [rd][C] fp32 (transposed).  gate out [B][C] fp32.
int launch_se(const float* pool_partial, int nblk, float inv_hw, const float* w1, const float* b1, const float* w2,
              const float* b2, float* gate, int B, int C, int rd, int act1, hipStream_t st);

// Global average pool of NHWC bf16 -> pooled fp32 [B][C] (+ optional bf16 copy for the classifier GEMM).
int launch_gap(const bf16_t* in, float* pooled, bf16_t* pooled_bf16, int B, int HW, int C, hipStream_t st);
// head 1x1 conv + bias + act + global average pool in one kernel (gemm_bf16.hip); pooled / pooled_bf16: [B][ldp]
bool head_gap_supported(int HW, int N, int K, int lda, int ldw, int act);
int launch_head_gap(const bf16_t* A, int lda, const bf16_t* W, int ldw, const float* bias, float* pooled, bf16_t* pooled_bf16,
                    int ldp, int B, int HW, int N, int K, int act, hipStream_t st);

// ClassifierHead on an un-pooled NCHW fp32 map: pooled_out (optional) [B][C] fp32 = mean over HW; out (when w != null) [B][N]
// = Linear(bf16(pooled); bf16(w) [N][C], bias [N] or null) with fp32 accumulation.
int launch_pool_linear(const float* fm, const float* w, const float* bias, float* out, float* pooled_out, int B, int C,
                       int HW, int N, hipStream_t st);

// NHWC bf16 [B][HW][C] -> NCHW fp32 [B][Cvalid][HW] (forward_features output, taps).
int launch_nhwc_to_nchw_f32(const bf16_t* in, float* out, int B, int HW, int C, int Cvalid, hipStream_t st);
// NCHW fp32 [B][Cvalid][HW] -> NHWC bf16 [B][HW][C] (channels >= Cvalid are written as zeros); parity tool.
int launch_nchw_f32_to_nhwc_bf16(const float* in, bf16_t* out, int B, int HW, int Cvalid, int C, hipStream_t st);

// SquarePad + ToTensor + Normalize: uint8 HWC (h, w, 3) -> fp32 CHW (3, S, S), S = max(h, w); mean/std are HOST arrays.
int launch_square_pad_normalize(const unsigned char* img, int h, int w, int fill, const float* mean, const float* stdv,
                                float* out, hipStream_t st);

// conv_input pre-stem: SiLU(conv3x3 s1 p1, 3->3, no bias), fp32 NCHW in/out.
int launch_conv_input_silu(const float* x, const float* w, int B, int H, int W, float* out, hipStream_t st);

}  // namespace mi355
