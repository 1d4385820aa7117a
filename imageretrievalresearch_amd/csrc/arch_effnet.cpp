// efficientnet_b3a (timm 0.4.12): tensor table in state-dict order + executor plan.
// Structure: decode of the efficientnet_b0 arch strings with channel_multiplier 1.2 / depth_multiplier 1.4
// (SURVEY §3.4, §8a T1); replaces timm.create_model('efficientnet_b3a') at inference/inference.py:102.
#include "model.h"

#include <math.h>

namespace mi355 {

namespace {
struct StageDef { bool ir; int k, s, e, c, r; };
const StageDef kB0[] = {{false, 3, 1, 1, 16, 1}, {true, 3, 2, 6, 24, 2}, {true, 5, 2, 6, 40, 2}, {true, 3, 2, 6, 80, 3},
                        {true, 5, 1, 6, 112, 3}, {true, 5, 2, 6, 192, 4}, {true, 3, 1, 6, 320, 1}};
}  // namespace

int build_efficientnet_b3(ModelDef& m) {
    const double cm = 1.2, dm = 1.4;
    auto rc = [&](int c) { return make_divisible(c * cm, 8); };
    const int stem = rc(32);
    m.feat_dim = rc(1280);
    m.feat_dim_pad = m.feat_dim;

    m.add("conv_stem.weight", {stem, 3, 3, 3});
    m.add_bn("bn1", stem);
    {
        Op op; op.kind = OP_STEM; op.out = SLOT_X0; op.cin = op.cin_real = 3; op.cout = op.cout_real = stem;
        op.k = 3; op.stride = 2; op.act = ACT_SILU; op.w_name = "conv_stem.weight"; op.bn_name = "bn1"; op.tap = "stem";
        m.ops.push_back(op);
    }
    int cur = SLOT_X0, cin = stem;
    for (int si = 0; si < 7; ++si) {
        const StageDef& sd = kB0[si];
        const int cout = rc(sd.c);
        const int reps = (int)ceil(sd.r * dm);
        for (int bi = 0; bi < reps; ++bi) {
            const std::string p = "blocks." + std::to_string(si) + "." + std::to_string(bi);
            const int stride = bi == 0 ? sd.s : 1;
            const int mid = cin * sd.e;
            const int rd = make_divisible(cin * 0.25, 1);
            const int nxt = cur == SLOT_X0 ? SLOT_X1 : SLOT_X0;
            int dw_in = cur;
            std::string dw_bn, pw_name, pw_bn;
            if (sd.ir) {
                m.add(p + ".conv_pw.weight", {mid, cin, 1, 1});
                m.add_bn(p + ".bn1", mid);
                m.add(p + ".conv_dw.weight", {mid, 1, sd.k, sd.k});
                m.add_bn(p + ".bn2", mid);
                Op e; e.kind = OP_GEMM; e.in = cur; e.out = SLOT_E; e.cin = e.cin_real = cin; e.cout = e.cout_real = mid;
                e.act = ACT_SILU; e.w_name = p + ".conv_pw.weight"; e.bn_name = p + ".bn1";
                m.ops.push_back(e);
                dw_in = SLOT_E; dw_bn = p + ".bn2"; pw_name = p + ".conv_pwl.weight"; pw_bn = p + ".bn3";
            } else {
                m.add(p + ".conv_dw.weight", {mid, 1, sd.k, sd.k});
                m.add_bn(p + ".bn1", mid);
                dw_bn = p + ".bn1"; pw_name = p + ".conv_pw.weight"; pw_bn = p + ".bn2";
            }
            m.add(p + ".se.conv_reduce.weight", {rd, mid, 1, 1});
            m.add(p + ".se.conv_reduce.bias", {rd});
            m.add(p + ".se.conv_expand.weight", {mid, rd, 1, 1});
            m.add(p + ".se.conv_expand.bias", {mid});
            m.add(pw_name, {cout, mid, 1, 1});
            m.add_bn(pw_bn, cout);

            Op d; d.kind = OP_DW; d.in = dw_in; d.out = SLOT_D; d.cin = d.cout = d.cin_real = d.cout_real = mid;
            d.k = sd.k; d.stride = stride; d.act = ACT_SILU; d.pool = true;
            d.w_name = p + ".conv_dw.weight"; d.bn_name = dw_bn;
            m.ops.push_back(d);

            Op s; s.kind = OP_SE; s.in = SLOT_POOLPART; s.out = SLOT_GATE; s.cin = s.cout = s.cin_real = s.cout_real = mid;
            s.rd = rd; s.se_act = ACT_SILU;
            s.w_name = p + ".se.conv_reduce.weight"; s.bias_name = p + ".se.conv_reduce.bias";
            s.w2_name = p + ".se.conv_expand.weight"; s.bias2_name = p + ".se.conv_expand.bias";
            m.ops.push_back(s);

            Op g; g.kind = OP_GEMM; g.in = SLOT_D; g.out = nxt; g.cin = g.cin_real = mid; g.cout = g.cout_real = cout;
            g.act = ACT_NONE; g.use_gate = true; g.w_name = pw_name; g.bn_name = pw_bn;
            if (stride == 1 && cin == cout) g.res = cur;
            g.tap = p;
            m.ops.push_back(g);
            cur = nxt;
            cin = cout;
        }
    }
    m.add("conv_head.weight", {m.feat_dim, cin, 1, 1});
    m.add_bn("bn2", m.feat_dim);
    {
        Op h; h.kind = OP_GEMM; h.in = cur; h.out = SLOT_HEAD; h.cin = h.cin_real = cin; h.cout = h.cout_real = m.feat_dim;
        h.act = ACT_SILU; h.w_name = "conv_head.weight"; h.bn_name = "bn2"; h.tap = "head";
        m.ops.push_back(h);
    }
    m.final_slot = SLOT_HEAD;
    if (m.num_classes > 0) {
        m.add("classifier.weight", {m.num_classes, m.feat_dim});
        m.add("classifier.bias", {m.num_classes});
        Op c; c.kind = OP_GEMM; c.in = SLOT_POOLED_BF16; c.cin = c.cin_real = m.feat_dim;
        c.cout = c.cout_real = m.num_classes; c.act = ACT_NONE;
        c.w_name = "classifier.weight"; c.bias_name = "classifier.bias";
        m.classifier = c;
    }
    return OK;
}

}  // namespace mi355
