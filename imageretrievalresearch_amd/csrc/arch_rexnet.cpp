// rexnet_{100,130,150,200} (timm 0.4.12 rexnet.py): tensor table in state-dict order + executor plan.
// Replaces timm.create_model('rexnet_150') (inference/inference.py:268 default) and the forward_features/head
// pair of train/train.py:194-195.  Channel counts are arbitrary integers (54, 77, 167, ... SURVEY H4): every
// activation is stored with its channel count padded to a multiple of 8; pad channels carry exact zeros
// (zero weights, zero bias, SiLU(0) = ReLU6(0) = 0, 0.5 * 0 under the SE gate).
#include "model.h"

#include <math.h>

namespace mi355 {

static int py_round(double v) { return (int)nearbyint(v); }   // Python round(): half to even

int build_rexnet(ModelDef& m, double wm) {
    const int layers[6] = {1, 2, 2, 3, 3, 5};
    const int strides6[6] = {1, 2, 2, 2, 1, 2};
    std::vector<int> strides, exps;
    std::vector<double> ses;
    int total = 0;
    for (int i = 0; i < 6; ++i) {
        for (int j = 0; j < layers[i]; ++j) {
            strides.push_back(j == 0 ? strides6[i] : 1);
            exps.push_back(i == 0 ? 1 : 6);
            ses.push_back(i < 2 ? 0.0 : 1.0 / 12.0);
        }
        total += layers[i];
    }
    double base = wm < 1.0 ? 16.0 / wm : 16.0;
    std::vector<int> outs;
    for (int i = 0; i < total; ++i) {
        outs.push_back(make_divisible(py_round(base * wm), 1));
        base += 180.0 / total;
    }
    const int stem = make_divisible(py_round(32.0 * wm), 1);
    const int pen = make_divisible(1280.0 * wm, 1);
    m.feat_dim = pen;
    m.feat_dim_pad = pad8(pen);

    m.add("stem.conv.weight", {stem, 3, 3, 3});
    m.add_bn("stem.bn", stem);
    {
        Op op; op.kind = OP_STEM; op.out = SLOT_X0; op.cin = op.cin_real = 3; op.cout_real = stem; op.cout = pad8(stem);
        op.k = 3; op.stride = 2; op.act = ACT_SILU; op.w_name = "stem.conv.weight"; op.bn_name = "stem.bn"; op.tap = "stem";
        m.ops.push_back(op);
    }
    int cur = SLOT_X0, prev = stem;
    for (int i = 0; i < total; ++i) {
        const std::string p = "features." + std::to_string(i);
        const int cout = outs[i], e = exps[i], s = strides[i];
        const int dw = e != 1 ? make_divisible(py_round((double)prev * e), 1) : prev;
        const int rd = ses[i] > 0 ? make_divisible((int)(dw * ses[i]), 1) : 0;
        const int nxt = cur == SLOT_X0 ? SLOT_X1 : SLOT_X0;
        int dw_in = cur;
        if (e != 1) {
            m.add(p + ".conv_exp.conv.weight", {dw, prev, 1, 1});
            m.add_bn(p + ".conv_exp.bn", dw);
            Op g; g.kind = OP_GEMM; g.in = cur; g.out = SLOT_E; g.cin_real = prev; g.cin = pad8(prev);
            g.cout_real = dw; g.cout = pad8(dw); g.act = ACT_SILU;
            g.w_name = p + ".conv_exp.conv.weight"; g.bn_name = p + ".conv_exp.bn";
            m.ops.push_back(g);
            dw_in = SLOT_E;
        }
        m.add(p + ".conv_dw.conv.weight", {dw, 1, 3, 3});
        m.add_bn(p + ".conv_dw.bn", dw);
        Op d; d.kind = OP_DW; d.in = dw_in; d.out = SLOT_D; d.cin_real = d.cout_real = dw; d.cin = d.cout = pad8(dw);
        d.k = 3; d.stride = s; d.act = ACT_NONE; d.pool = rd > 0;
        d.w_name = p + ".conv_dw.conv.weight"; d.bn_name = p + ".conv_dw.bn";
        m.ops.push_back(d);
        if (rd > 0) {
            m.add(p + ".se.fc1.weight", {rd, dw, 1, 1});
            m.add(p + ".se.fc1.bias", {rd});
            m.add_bn(p + ".se.bn", rd);
            m.add(p + ".se.fc2.weight", {dw, rd, 1, 1});
            m.add(p + ".se.fc2.bias", {dw});
            Op se; se.kind = OP_SE; se.in = SLOT_POOLPART; se.out = SLOT_GATE; se.cin_real = se.cout_real = dw;
            se.cin = se.cout = pad8(dw); se.rd = rd; se.se_act = ACT_RELU;
            se.w_name = p + ".se.fc1.weight"; se.bias_name = p + ".se.fc1.bias"; se.bn2_name = p + ".se.bn";
            se.w2_name = p + ".se.fc2.weight"; se.bias2_name = p + ".se.fc2.bias";
            m.ops.push_back(se);
        }
        m.add(p + ".conv_pwl.conv.weight", {cout, dw, 1, 1});
        m.add_bn(p + ".conv_pwl.bn", cout);
        Op g; g.kind = OP_GEMM; g.in = SLOT_D; g.out = nxt; g.cin_real = dw; g.cin = pad8(dw);
        g.cout_real = cout; g.cout = pad8(cout); g.act = ACT_NONE; g.use_gate = rd > 0; g.a_relu6 = 1;
        g.w_name = p + ".conv_pwl.conv.weight"; g.bn_name = p + ".conv_pwl.bn";
        if (s == 1 && prev <= cout) { g.res = cur; g.res_channels = prev; }
        g.tap = p;
        m.ops.push_back(g);
        cur = nxt;
        prev = cout;
    }
    const std::string p = "features." + std::to_string(total);
    m.add(p + ".conv.weight", {pen, prev, 1, 1});
    m.add_bn(p + ".bn", pen);
    {
        Op h; h.kind = OP_GEMM; h.in = cur; h.out = SLOT_HEAD; h.cin_real = prev; h.cin = pad8(prev);
        h.cout_real = pen; h.cout = pad8(pen); h.act = ACT_SILU; h.w_name = p + ".conv.weight"; h.bn_name = p + ".bn";
        h.tap = "head";
        m.ops.push_back(h);
    }
    m.final_slot = SLOT_HEAD;
    if (m.num_classes > 0) {
        m.add("head.fc.weight", {m.num_classes, pen});
        m.add("head.fc.bias", {m.num_classes});
        Op c; c.kind = OP_GEMM; c.in = SLOT_POOLED_BF16; c.cin_real = pen; c.cin = pad8(pen);
        c.cout = c.cout_real = m.num_classes; c.act = ACT_NONE;
        c.w_name = "head.fc.weight"; c.bias_name = "head.fc.bias";
        m.classifier = c;
    }
    return OK;
}

}  // namespace mi355
