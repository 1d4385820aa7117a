// Model description + executor plan shared by the architecture builders and model.hip.
#pragma once
#include "ops.h"

#include <map>
#include <string>
#include <vector>

namespace mi355 {

struct TensorSpec {
    std::string name;
    std::vector<int64_t> shape;
    int kind = 0;  // 0 parameter, 1 float buffer, 2 int64 buffer
    std::vector<float> data;
    bool set = false;
    int64_t numel() const {
        int64_t n = 1;
        for (auto d : shape) n *= d;
        return n;
    }
};

enum OpKind {
    OP_STEM,      // 3x3/s2 conv from the NCHW fp32 input
    OP_GEMM,      // 1x1 conv / linear
    OP_DW,        // depthwise conv (+ SE squeeze partials)
    OP_SE,        // SE gate from the squeeze partials
    // swin ops are appended by swin.cpp (see model.hip for their execution)
    OP_PATCH_EMBED, OP_LAYERNORM, OP_WINATTN, OP_PATCH_MERGE_LN, OP_TOKEN_MEAN,
};

// Activation slots of the arena.  The conv nets are a chain with one residual, so a handful of
// rotating buffers is enough; each slot is sized to the largest tensor any op writes into it.
enum Slot {
    SLOT_NONE = -1, SLOT_X0 = 0, SLOT_X1, SLOT_E, SLOT_D, SLOT_POOLPART, SLOT_GATE, SLOT_HEAD, SLOT_POOLED,
    SLOT_POOLED_BF16, SLOT_T0, SLOT_T1, SLOT_T2, SLOT_T3, SLOT_SPLITK, SLOT_LNSTATS, SLOT_COUNT
};

struct Op {
    OpKind kind;
    int in = SLOT_NONE, out = SLOT_NONE, res = SLOT_NONE;
    int cin = 0, cout = 0;          // padded (multiple of 8) channel counts as stored in HBM
    int cin_real = 0, cout_real = 0;
    int k = 1, stride = 1;
    int act = ACT_NONE;
    int a_relu6 = 0;                // GEMM: relu6 on the A operand while loading (rexnet act_dw)
    bool use_gate = false;          // GEMM: multiply A by the SE gate while loading
    bool pool = false;              // DW: emit SE squeeze partial sums
    int rd = 0;                     // SE: reduced channels
    int se_act = ACT_SILU;          // SE: activation after the reduce FC
    int res_channels = 0;           // GEMM: residual covers only the first res_channels outputs (rexnet); 0 = all
    // swin
    int heads = 0, window = 0, shift = 0, tokens_h = 0;
    float ln_eps = 1e-5f;
    // LayerNorm folded into the consumer GEMM (swin norm1 -> qkv, norm2 -> fc1).  LN op: fuse_next = the next op is that GEMM.
    // GEMM op: ln_w_name / ln_b_name = the LayerNorm's gamma / beta; pack() then also stores W' = bf16(W * gamma) (w_ln_off),
    // b' = b + W beta (b_ln_off) and the column sums of W' (cs_off), so that  LN(x) W^T + b = rstd (x W'^T - mean cs) + b'.
    bool fuse_next = false;
    std::string ln_w_name, ln_b_name;
    size_t w_ln_off = 0, b_ln_off = 0, cs_off = 0;
    // packed-weight offsets (bytes into the device blob), filled by pack()
    size_t w_off = 0, b_off = 0, w2_off = 0, b2_off = 0, aux_off = 0;
    size_t w3_off = 0, w4_off = 0;  // SE: bf16 copies of the two FC matrices (whole-block kernel, mbconv_block.hip)
    // source tensors (timm keys) for packing
    std::string w_name, bn_name, bias_name, w2_name, bias2_name, bn2_name, aux_name;
    float bn_eps = 1e-5f;
    std::string tap;                // record a tap with this name after the op (when taps are enabled)
};

struct ModelDef {
    std::string arch;
    int num_classes = 0;
    int feat_dim = 0;
    int feat_dim_pad = 0;
    bool pools_in_features = false;   // swin: forward_features already returns (B, D)
    std::vector<TensorSpec> tensors;
    std::map<std::string, int> index;
    std::vector<Op> ops;              // backbone up to and including the head conv / final norm
    Op classifier;                    // GEMM, valid when num_classes > 0
    int final_slot = SLOT_HEAD;       // slot holding the un-pooled features (conv nets) / tokens (swin)

    int add(const std::string& name, std::vector<int64_t> shape, int kind = 0) {
        TensorSpec t;
        t.name = name; t.shape = std::move(shape); t.kind = kind;
        tensors.push_back(std::move(t));
        index[name] = (int)tensors.size() - 1;
        return (int)tensors.size() - 1;
    }
    void add_bn(const std::string& p, int c) {
        add(p + ".weight", {c}); add(p + ".bias", {c});
        add(p + ".running_mean", {c}, 1); add(p + ".running_var", {c}, 1);
        add(p + ".num_batches_tracked", {}, 2);
    }
    void add_ln(const std::string& p, int c) { add(p + ".weight", {c}); add(p + ".bias", {c}); }
};

static inline int make_divisible(double v, int divisor = 8, int min_value = 0, double round_limit = 0.9) {
    if (min_value == 0) min_value = divisor;
    int new_v = (int)(v + divisor / 2.0) / divisor * divisor;
    if (new_v < min_value) new_v = min_value;
    if (new_v < round_limit * v) new_v += divisor;
    return new_v;
}
static inline int pad8(int c) { return (c + 7) & ~7; }

// architecture builders (arch_*.cpp); return 0 or set_error + nonzero
int build_efficientnet_b3(ModelDef& m);
int build_rexnet(ModelDef& m, double width_mult);
int build_swin_base(ModelDef& m);

}  // namespace mi355
