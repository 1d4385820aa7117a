// swin_base_patch4_window7_224 (timm 0.4.12 swin_transformer.py): tensor table in state-dict order + plan.
// Replaces timm.create_model(...) at train/train_vit_triplet.py:354 (with model.head = Identity() at :357).
// Tokens are [B][L][C] bf16; the residual stream is updated in place by the proj / fc2 GEMM epilogues.
#include "model.h"

namespace mi355 {

int build_swin_base(ModelDef& m) {
    const int embed = 128, depths[4] = {2, 2, 18, 2}, heads[4] = {4, 8, 16, 32}, ws = 7, grid0 = 56;
    m.feat_dim = m.feat_dim_pad = embed * 8;
    m.pools_in_features = true;
    m.final_slot = SLOT_X0;

    m.add("patch_embed.proj.weight", {embed, 3, 4, 4});
    m.add("patch_embed.proj.bias", {embed});
    m.add_ln("patch_embed.norm", embed);
    {
        Op op; op.kind = OP_PATCH_EMBED; op.out = SLOT_X0; op.cin = op.cin_real = 3; op.cout = op.cout_real = embed;
        op.tokens_h = grid0; op.w_name = "patch_embed.proj.weight"; op.bias_name = "patch_embed.proj.bias";
        op.w2_name = "patch_embed.norm.weight"; op.bias2_name = "patch_embed.norm.bias"; op.tap = "patch_embed";
        m.ops.push_back(op);
    }
    for (int s = 0; s < 4; ++s) {
        const int dim = embed << s, res = grid0 >> s, nh = heads[s];
        for (int b = 0; b < depths[s]; ++b) {
            const std::string p = "layers." + std::to_string(s) + ".blocks." + std::to_string(b);
            const int shift = (b % 2 == 0 || res <= ws) ? 0 : ws / 2;
            const int nW = (res / ws) * (res / ws);
            if (shift > 0) m.add(p + ".attn_mask", {nW, ws * ws, ws * ws}, 1);
            m.add_ln(p + ".norm1", dim);
            m.add(p + ".attn.relative_position_bias_table", {(2 * ws - 1) * (2 * ws - 1), nh});
            m.add(p + ".attn.relative_position_index", {ws * ws, ws * ws}, 2);
            m.add(p + ".attn.qkv.weight", {3 * dim, dim});
            m.add(p + ".attn.qkv.bias", {3 * dim});
            m.add(p + ".attn.proj.weight", {dim, dim});
            m.add(p + ".attn.proj.bias", {dim});
            m.add_ln(p + ".norm2", dim);
            m.add(p + ".mlp.fc1.weight", {4 * dim, dim});
            m.add(p + ".mlp.fc1.bias", {4 * dim});
            m.add(p + ".mlp.fc2.weight", {dim, 4 * dim});
            m.add(p + ".mlp.fc2.bias", {dim});

            Op ln1; ln1.kind = OP_LAYERNORM; ln1.in = SLOT_X0; ln1.out = SLOT_T0; ln1.cin = ln1.cout = ln1.cin_real = ln1.cout_real = dim;
            ln1.tokens_h = res; ln1.w_name = p + ".norm1.weight"; ln1.bias_name = p + ".norm1.bias"; ln1.fuse_next = true;
            m.ops.push_back(ln1);
            Op qkv; qkv.kind = OP_GEMM; qkv.in = SLOT_T0; qkv.out = SLOT_T1; qkv.cin = qkv.cin_real = dim;
            qkv.cout = qkv.cout_real = 3 * dim; qkv.tokens_h = res; qkv.w_name = p + ".attn.qkv.weight"; qkv.bias_name = p + ".attn.qkv.bias";
            qkv.ln_w_name = p + ".norm1.weight"; qkv.ln_b_name = p + ".norm1.bias";
            m.ops.push_back(qkv);
            Op at; at.kind = OP_WINATTN; at.in = SLOT_T1; at.out = SLOT_T2; at.cin = at.cin_real = 3 * dim; at.cout = at.cout_real = dim;
            at.heads = nh; at.window = ws; at.shift = shift; at.tokens_h = res; at.aux_name = p + ".attn.relative_position_bias_table";
            m.ops.push_back(at);
            Op pr; pr.kind = OP_GEMM; pr.in = SLOT_T2; pr.out = SLOT_X0; pr.res = SLOT_X0; pr.cin = pr.cin_real = dim;
            pr.cout = pr.cout_real = dim; pr.tokens_h = res; pr.w_name = p + ".attn.proj.weight"; pr.bias_name = p + ".attn.proj.bias";
            m.ops.push_back(pr);
            Op ln2 = ln1; ln2.w_name = p + ".norm2.weight"; ln2.bias_name = p + ".norm2.bias";
            m.ops.push_back(ln2);
            Op f1; f1.kind = OP_GEMM; f1.in = SLOT_T0; f1.out = SLOT_T1; f1.cin = f1.cin_real = dim; f1.cout = f1.cout_real = 4 * dim;
            f1.act = ACT_GELU; f1.tokens_h = res; f1.w_name = p + ".mlp.fc1.weight"; f1.bias_name = p + ".mlp.fc1.bias";
            f1.ln_w_name = p + ".norm2.weight"; f1.ln_b_name = p + ".norm2.bias";
            m.ops.push_back(f1);
            Op f2; f2.kind = OP_GEMM; f2.in = SLOT_T1; f2.out = SLOT_X0; f2.res = SLOT_X0; f2.cin = f2.cin_real = 4 * dim;
            f2.cout = f2.cout_real = dim; f2.tokens_h = res; f2.w_name = p + ".mlp.fc2.weight"; f2.bias_name = p + ".mlp.fc2.bias";
            f2.tap = p;
            m.ops.push_back(f2);
        }
        if (s < 3) {
            const std::string p = "layers." + std::to_string(s) + ".downsample";
            m.add(p + ".reduction.weight", {2 * dim, 4 * dim});
            m.add_ln(p + ".norm", 4 * dim);
            Op mg; mg.kind = OP_PATCH_MERGE_LN; mg.in = SLOT_X0; mg.out = SLOT_T0; mg.cin = mg.cin_real = dim;
            mg.cout = mg.cout_real = 4 * dim; mg.tokens_h = res / 2; mg.w_name = p + ".norm.weight"; mg.bias_name = p + ".norm.bias";
            m.ops.push_back(mg);
            Op rd; rd.kind = OP_GEMM; rd.in = SLOT_T0; rd.out = SLOT_X0; rd.cin = rd.cin_real = 4 * dim; rd.cout = rd.cout_real = 2 * dim;
            rd.tokens_h = res / 2; rd.w_name = p + ".reduction.weight"; rd.tap = p;
            m.ops.push_back(rd);
        }
    }
    m.add_ln("norm", embed * 8);
    {
        Op f; f.kind = OP_TOKEN_MEAN; f.in = SLOT_X0; f.out = SLOT_NONE; f.cin = f.cin_real = f.cout = f.cout_real = embed * 8;
        f.tokens_h = 7; f.w_name = "norm.weight"; f.bias_name = "norm.bias";
        m.ops.push_back(f);
    }
    if (m.num_classes > 0) {
        m.add("head.weight", {m.num_classes, embed * 8});
        m.add("head.bias", {m.num_classes});
        Op c; c.kind = OP_GEMM; c.in = SLOT_POOLED_BF16; c.cin = c.cin_real = embed * 8; c.cout = c.cout_real = m.num_classes;
        c.w_name = "head.weight"; c.bias_name = "head.bias";
        m.classifier = c;
    }
    return OK;
}

}  // namespace mi355
