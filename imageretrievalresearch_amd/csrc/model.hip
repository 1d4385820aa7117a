// Model object behind the C ABI: tensor table, one-time weight pack (BN fold -> bf16 -> kernel layout),
// arena, and the executor that walks the op plan launching the HIP kernels on the caller's stream.
// Replaces timm.create_model(...) + .forward/.forward_features (see include/mi355_retrieval.h).
#include "model_exec.h"
#include "../../include/mi355_retrieval.h"

#include <string.h>

#include <algorithm>

namespace mi355 {

int pack_gemm(Packer& pk, Op& op) {
    const TensorSpec* w = pk.get(op.w_name);
    if (!w) return ERR_STATE;
    const int N = op.cout_real, K = op.cin_real;
    MI355_REQUIRE(w->numel() == (int64_t)N * K, "pack: %s has %lld elements, expected %d x %d", op.w_name.c_str(),
                  (long long)w->numel(), N, K);
    std::vector<float> scale, shift;
    if (!pk.bn_fold(op.bn_name, op.bn_eps, N, scale, shift)) return ERR_STATE;
    if (!op.bias_name.empty()) {
        const TensorSpec* b = pk.get(op.bias_name);
        if (!b) return ERR_STATE;
        for (int i = 0; i < N; ++i) shift[i] = b->data[i] * scale[i] + shift[i];
    }
    const int Np = (op.cout + 15) & ~15, Kp = (op.cin + 31) & ~31;
    op.w_off = pk.alloc((size_t)Np * Kp * 2);
    op.b_off = pk.alloc((size_t)Np * 4);
    uint16_t* W = (uint16_t*)(pk.blob.data() + op.w_off);
    float* Bv = (float*)(pk.blob.data() + op.b_off);
    for (int n = 0; n < N; ++n) {
        for (int k = 0; k < K; ++k) W[(size_t)n * Kp + k] = f2bf_host(w->data[(size_t)n * K + k] * scale[n]);
        Bv[n] = shift[n];
    }
    if (!op.ln_w_name.empty()) {   // LayerNorm folded into this GEMM (see Op::fuse_next)
        const TensorSpec* g = pk.get(op.ln_w_name);
        const TensorSpec* be = pk.get(op.ln_b_name);
        if (!g || !be) return ERR_STATE;
        MI355_REQUIRE(g->numel() == K && be->numel() == K, "pack: %s / %s must have %d elements", op.ln_w_name.c_str(),
                      op.ln_b_name.c_str(), K);
        op.w_ln_off = pk.alloc((size_t)Np * Kp * 2);
        op.b_ln_off = pk.alloc((size_t)Np * 4);
        op.cs_off = pk.alloc((size_t)Np * 4);
        uint16_t* W2 = (uint16_t*)(pk.blob.data() + op.w_ln_off);
        float* B2 = (float*)(pk.blob.data() + op.b_ln_off);
        float* CS = (float*)(pk.blob.data() + op.cs_off);
        const float* Bv1 = (const float*)(pk.blob.data() + op.b_off);     // (alloc may have moved the blob)
        for (int n = 0; n < N; ++n) {
            double cs = 0.0, bb = 0.0;
            for (int k = 0; k < K; ++k) {
                const float wv = w->data[(size_t)n * K + k] * scale[n];
                const uint16_t h = f2bf_host(wv * g->data[k]);
                W2[(size_t)n * Kp + k] = h;
                uint32_t u = (uint32_t)h << 16; float hf; memcpy(&hf, &u, 4);
                cs += hf;
                bb += (double)wv * be->data[k];
            }
            CS[n] = (float)cs;
            B2[n] = Bv1[n] + (float)bb;
        }
    }
    return OK;
}

static int pack_dw(Packer& pk, Op& op) {
    const TensorSpec* w = pk.get(op.w_name);
    if (!w) return ERR_STATE;
    const int C = op.cin_real, Cp = op.cin, kk = op.k * op.k;
    MI355_REQUIRE(w->numel() == (int64_t)C * kk, "pack: %s has %lld elements, expected %d x %d", op.w_name.c_str(),
                  (long long)w->numel(), C, kk);
    std::vector<float> scale, shift;
    if (!pk.bn_fold(op.bn_name, op.bn_eps, C, scale, shift)) return ERR_STATE;
    op.w_off = pk.alloc((size_t)kk * Cp * 2);
    op.b_off = pk.alloc((size_t)Cp * 4);
    uint16_t* W = (uint16_t*)(pk.blob.data() + op.w_off);
    float* Bv = (float*)(pk.blob.data() + op.b_off);
    for (int c = 0; c < C; ++c) {
        for (int t = 0; t < kk; ++t) W[(size_t)t * Cp + c] = f2bf_host(w->data[(size_t)c * kk + t] * scale[c]);
        Bv[c] = shift[c];
    }
    return OK;
}

static int pack_stem(Packer& pk, Op& op) {
    const TensorSpec* w = pk.get(op.w_name);
    if (!w) return ERR_STATE;
    const int Co = op.cout_real, Cp = op.cout;
    MI355_REQUIRE(w->numel() == (int64_t)Co * 27, "pack: %s has %lld elements, expected %d x 27", op.w_name.c_str(),
                  (long long)w->numel(), Co);
    std::vector<float> scale, shift;
    if (!pk.bn_fold(op.bn_name, op.bn_eps, Co, scale, shift)) return ERR_STATE;
    op.w_off = pk.alloc((size_t)27 * Cp * 4);
    op.b_off = pk.alloc((size_t)Cp * 4);
    float* W = (float*)(pk.blob.data() + op.w_off);
    float* Bv = (float*)(pk.blob.data() + op.b_off);
    for (int co = 0; co < Co; ++co) {
        for (int ci = 0; ci < 3; ++ci)
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx)
                    W[(size_t)((ky * 3 + kx) * 3 + ci) * Cp + co] =
                        bf_round_host(w->data[(((size_t)co * 3 + ci) * 3 + ky) * 3 + kx] * scale[co]);
        Bv[co] = shift[co];
    }
    return OK;
}

static int pack_se(Packer& pk, Op& op) {
    const TensorSpec *w1 = pk.get(op.w_name), *b1 = pk.get(op.bias_name), *w2 = pk.get(op.w2_name),
                     *b2 = pk.get(op.bias2_name);
    if (!w1 || !b1 || !w2 || !b2) return ERR_STATE;
    const int C = op.cin_real, Cp = op.cin, rd = op.rd;
    MI355_REQUIRE(w1->numel() == (int64_t)rd * C && w2->numel() == (int64_t)C * rd, "pack: SE %s shape mismatch",
                  op.w_name.c_str());
    std::vector<float> scale, shift;  // rexnet: BN between the reduce FC and its ReLU
    if (!pk.bn_fold(op.bn2_name, op.bn_eps, rd, scale, shift)) return ERR_STATE;
    op.w_off = pk.alloc((size_t)rd * Cp * 4);
    op.b_off = pk.alloc((size_t)rd * 4);
    op.w2_off = pk.alloc((size_t)Cp * rd * 4);
    op.b2_off = pk.alloc((size_t)Cp * 4);
    float* W1 = (float*)(pk.blob.data() + op.w_off);
    float* B1 = (float*)(pk.blob.data() + op.b_off);
    float* W2 = (float*)(pk.blob.data() + op.w2_off);
    float* B2 = (float*)(pk.blob.data() + op.b2_off);
    for (int j = 0; j < rd; ++j) {
        for (int c = 0; c < C; ++c) W1[(size_t)j * Cp + c] = bf_round_host(w1->data[(size_t)j * C + c] * scale[j]);   // bf16 values (see below)
        B1[j] = b1->data[j] * scale[j] + shift[j];
    }
    for (int c = 0; c < C; ++c) {
        for (int j = 0; j < rd; ++j) W2[(size_t)j * Cp + c] = bf_round_host(w2->data[(size_t)c * rd + j]);   // transposed [rd][Cp]
        B2[c] = b2->data[c];
    }
    // bf16 copies for the whole-block kernel (every workgroup streams both matrices from L2: half the bytes).  The fp32
    // copies above hold the same bf16-rounded values, so k_se and the block kernel compute the same gate.
    op.w3_off = pk.alloc((size_t)rd * Cp * 2);
    op.w4_off = pk.alloc((size_t)rd * Cp * 2);
    {
        // (alloc may have moved the blob: re-derive the fp32 views)
        const float* W1f = (const float*)(pk.blob.data() + op.w_off);
        const float* W2f = (const float*)(pk.blob.data() + op.w2_off);
        uint16_t* W1b = (uint16_t*)(pk.blob.data() + op.w3_off);
        uint16_t* W2b = (uint16_t*)(pk.blob.data() + op.w4_off);
        for (size_t i = 0; i < (size_t)rd * Cp; ++i) { W1b[i] = f2bf_host(W1f[i]); W2b[i] = f2bf_host(W2f[i]); }
    }
    return OK;
}

int swin_pack(Packer& pk, Op& op);  // swin_kernels.hip

static int pack_op(Packer& pk, Op& op) {
    switch (op.kind) {
        case OP_STEM: return pack_stem(pk, op);
        case OP_GEMM: return pack_gemm(pk, op);
        case OP_DW: return pack_dw(pk, op);
        case OP_SE: return pack_se(pk, op);
        default: return swin_pack(pk, op);
    }
}

// ------------------------------------------------------------------------------------ shape pass
static inline int conv_out(int h, int k, int s) { return (h + 2 * (k / 2) - k) / s + 1; }

// Walk the plan for a chunk of nb images: fill slot dims and sizes.  Returns total arena bytes.
static size_t plan_slots(mi355_model* m, int nb, int H, int W, int B_full = 0) {
    if (B_full < nb) B_full = nb;       // the caller's whole batch decides the kernel choices that change rounding (split-K)
    SlotState* S = m->slots;
    for (int i = 0; i < SLOT_COUNT; ++i) S[i] = SlotState();
    auto need = [&](int s, size_t bytes) { if (s != SLOT_NONE && bytes > S[s].bytes) S[s].bytes = bytes; };
    for (const Op& op : m->def.ops) {
        switch (op.kind) {
            case OP_STEM: {
                const int ho = conv_out(H, 3, 2), wo = conv_out(W, 3, 2);
                S[op.out].h = ho; S[op.out].w = wo; S[op.out].c = op.cout;
                need(op.out, (size_t)nb * ho * wo * op.cout * 2);
                break;
            }
            case OP_GEMM: {
                const int h = S[op.in].h, w = S[op.in].w;
                S[op.out].h = h; S[op.out].w = w; S[op.out].c = op.cout;
                if (gemm_splitk_chunks((long)B_full * h * w, h * w, op.cout, op.cin) >= 2)
                    need(SLOT_SPLITK, gemm_splitk_bytes((long)nb * h * w, op.cout, op.cin));
                need(op.out, (size_t)nb * h * w * op.cout * 2);
                break;
            }
            case OP_DW: {
                const int ho = conv_out(S[op.in].h, op.k, op.stride), wo = conv_out(S[op.in].w, op.k, op.stride);
                S[op.out].h = ho; S[op.out].w = wo; S[op.out].c = op.cout;
                need(op.out, (size_t)nb * ho * wo * op.cout * 2);
                // squeeze partials: per 256-item block (k_dwconv) or per row band (fused kernels, <= ho bands)
                if (op.pool) need(SLOT_POOLPART, (size_t)nb * std::max(dw_pool_blocks(ho, wo, op.cout), ho) * op.cout * 4);
                break;
            }
            case OP_SE:
                need(SLOT_GATE, (size_t)nb * op.cin * 4);
                break;
            default: {
                // swin ops: sizes are declared by the builder through cin/cout/tokens_h
                const int th = op.tokens_h;
                if (op.out != SLOT_NONE) {
                    S[op.out].h = th; S[op.out].w = th; S[op.out].c = op.cout;
                    need(op.out, (size_t)nb * th * th * op.cout * 2);
                }
                if (op.kind == OP_LAYERNORM && op.fuse_next) need(SLOT_LNSTATS, (size_t)nb * th * th * 8);
                break;
            }
        }
    }
    need(SLOT_POOLED, (size_t)nb * m->def.feat_dim_pad * 4);
    need(SLOT_POOLED_BF16, (size_t)nb * m->def.feat_dim_pad * 2);
    size_t off = 0;
    for (int i = 0; i < SLOT_COUNT; ++i) {
        S[i].off = off;
        off += align_up(S[i].bytes, 256);
    }
    return off + 256;
}

static int ensure_arena(mi355_model* m, size_t bytes) {
    int dev = 0;
    MI355_CHECK_HIP(hipGetDevice(&dev));
    if (m->arena && m->arena_device != dev) {      // the model moved to another GPU: its scratch must follow
        MI355_CHECK_HIP(hipDeviceSynchronize());
        (void)hipFree(m->arena);                   // (hipFree finds the owning device by itself)
        m->arena = nullptr;
        m->arena_bytes = 0;
        for (auto& kv : m->tapbufs) { if (kv.second.ptr) (void)hipFree(kv.second.ptr); kv.second = TapBuf(); }
        if (m->stamp_buf) { (void)hipFree(m->stamp_buf); m->stamp_buf = nullptr; m->stamp_bytes = 0; }
    }
    m->arena_device = dev;
    if (bytes <= m->arena_bytes) return OK;
    if (m->arena) {
        MI355_CHECK_HIP(hipDeviceSynchronize());
        MI355_CHECK_HIP(hipFree(m->arena));
        m->arena = nullptr;
        m->arena_bytes = 0;
    }
    MI355_CHECK_HIP(hipMalloc(&m->arena, bytes));
    m->arena_bytes = bytes;
    return OK;
}

// ------------------------------------------------------------------------------------ execution
static int prof_kind(const Op& op) {
    switch (op.kind) {
        case OP_STEM: return PK_STEM;
        case OP_GEMM: return PK_GEMM;
        case OP_DW: return PK_DW;
        case OP_SE: return PK_SE;
        case OP_WINATTN: return PK_ATTN;
        case OP_LAYERNORM: case OP_PATCH_MERGE_LN: case OP_TOKEN_MEAN: return PK_LN;
        case OP_PATCH_EMBED: return PK_STEM;
        default: return PK_OTHER;
    }
}

static int record_tap(ExecCtx& cx, const Op& op) {
    mi355_model* m = cx.m;
    const SlotState& s = m->slots[op.out];
    TapBuf& t = m->tapbufs[op.tap];
    const size_t per_img = (size_t)s.h * s.w * s.c * 2;
    const size_t bytes = per_img * cx.B;
    if (t.bytes < bytes) {
        if (t.ptr) MI355_CHECK_HIP(hipFree(t.ptr));
        MI355_CHECK_HIP(hipMalloc(&t.ptr, bytes));
        t.bytes = bytes;
    }
    t.B = cx.B; t.h = s.h; t.w = s.w; t.c = s.c; t.c_real = op.cout_real ? op.cout_real : s.c;
    MI355_CHECK_HIP(hipMemcpyAsync((char*)t.ptr + per_img * cx.b0, cx.slot_ptr(op.out), per_img * cx.nb,
                                   hipMemcpyDeviceToDevice, cx.st));
    return OK;
}

static int exec_op(ExecCtx& cx, const Op& op) {
    mi355_model* m = cx.m;
    SlotState* S = m->slots;
    switch (op.kind) {
        case OP_STEM:
            if (cx.x_u8)
                return launch_stem_u8(cx.x_u8, cx.img_h, cx.img_w, cx.fill, cx.mean, cx.stdv, cx.conv_w,
                                      (const float*)cx.w(op.w_off), (const float*)cx.w(op.b_off),
                                      (bf16_t*)cx.slot_ptr(op.out), cx.nb, op.cout, op.act, cx.st);
            return launch_stem(cx.x, (const float*)cx.w(op.w_off), (const float*)cx.w(op.b_off),
                               (bf16_t*)cx.slot_ptr(op.out), cx.nb, cx.H, cx.W, op.cout, op.act, cx.st);
        case OP_GEMM: {
            const int hw = S[op.in].h * S[op.in].w;
            GemmArgs a{};
            a.A = (const bf16_t*)cx.slot_ptr(op.in); a.lda = op.cin;
            a.W = (const bf16_t*)cx.w(op.w_off); a.ldw = (op.cin + 31) & ~31;
            a.bias = (const float*)cx.w(op.b_off);
            a.res = op.res != SLOT_NONE ? (const bf16_t*)cx.slot_ptr(op.res) : nullptr;
            a.ldr = op.res != SLOT_NONE ? S[op.res].c : 0;
            a.res_n = op.res_channels ? op.res_channels : op.cout;
            a.gate = op.use_gate ? (const float*)cx.slot_ptr(SLOT_GATE) : nullptr;
            a.gate_ld = op.cin; a.rows_per_img = hw;
            a.out = cx.slot_ptr(op.out); a.ldo = op.cout; a.out_f32 = 0;
            a.M = cx.nb * hw; a.N = op.cout; a.K = op.cin;
            a.M_sel = (long)cx.B * hw;
            a.act = op.act; a.a_relu6 = op.a_relu6;
            a.zeros = (const bf16_t*)cx.w(0);
            if (cx.ln_pending_in != SLOT_NONE) {      // the preceding LayerNorm only left (mean, rstd) per row: fold it in here
                MI355_REQUIRE(op.w_ln_off && op.in != SLOT_NONE, "exec: GEMM after a fused LayerNorm has no folded weights");
                a.A = (const bf16_t*)cx.slot_ptr(cx.ln_pending_in);
                a.W = (const bf16_t*)cx.w(op.w_ln_off);
                a.bias = (const float*)cx.w(op.b_ln_off);
                a.ln_stats = (const float*)cx.slot_ptr(SLOT_LNSTATS);
                a.ln_colsum = (const float*)cx.w(op.cs_off);
                cx.ln_pending_in = SLOT_NONE;
            }
            if (m->slots[SLOT_SPLITK].bytes) {
                a.splitk_ws = (float*)cx.slot_ptr(SLOT_SPLITK);
                a.splitk_ws_bytes = m->slots[SLOT_SPLITK].bytes;
            }
            return launch_gemm_bf16(a, cx.st);
        }
        case OP_DW:
            return launch_dwconv((const bf16_t*)cx.slot_ptr(op.in), (const bf16_t*)cx.w(op.w_off),
                                 (const float*)cx.w(op.b_off), (bf16_t*)cx.slot_ptr(op.out),
                                 op.pool ? (float*)cx.slot_ptr(SLOT_POOLPART) : nullptr, cx.nb, S[op.in].h, S[op.in].w,
                                 op.cin, op.k, op.stride, op.act, &m->pool_nblk, cx.st);
        case OP_SE: {
            // the squeeze partials were produced by the preceding depthwise conv into SLOT_D's geometry
            const int ho = S[SLOT_D].h, wo = S[SLOT_D].w;
            return launch_se((const float*)cx.slot_ptr(SLOT_POOLPART), m->pool_nblk,
                             1.0f / (float)(ho * wo), (const float*)cx.w(op.w_off), (const float*)cx.w(op.b_off),
                             (const float*)cx.w(op.w2_off), (const float*)cx.w(op.b2_off),
                             (float*)cx.slot_ptr(SLOT_GATE), cx.nb, op.cin, op.rd, op.se_act, cx.st);
        }
        default:
            return swin_exec(m->def, op, cx);
    }
}

// row-sweep kernel (MFMA depthwise, complete squeeze sums): the early-stage shape classes of sweep_mbconv.hip
static bool use_sweep(const mi355_model* m, const Op& g, const Op& d, int h, int w) {
    return m->fuse_sweep && sweep_mbconv_supported(h, w, g.cin, g.cout, d.k, d.stride, g.act, d.act);
}

// expand GEMM (-> SLOT_E) immediately followed by the depthwise conv that consumes it, on a whole-image tile
static bool can_fuse(const mi355_model* m, size_t i, int h, int w) {
    if (!m->fuse || i + 1 >= m->def.ops.size()) return false;
    const Op& g = m->def.ops[i];
    const Op& d = m->def.ops[i + 1];
    if (g.kind != OP_GEMM || d.kind != OP_DW || g.out != SLOT_E || d.in != SLOT_E) return false;
    if (g.use_gate || g.res != SLOT_NONE || g.a_relu6 || !g.tap.empty()) return false;
    if (fused_late_supported(h, w, g.cin, g.cout, d.k, d.stride)) return true;
    if (use_sweep(m, g, d, h, w)) return true;
    // Band variant, measured per layer on EfficientNet-B3a B=256 (fused vs expand + depthwise): 3x3 s1 C192 @56x56 307 vs
    // 339 us (wins); 3x3 s2 C144 @112x112 780 vs 619, 5x5 s2 C192 @56x56 465 vs 276, 5x5 s1 C288 @28x28 247 vs 172 (lose:
    // short bands recompute too much halo and leave most threads idle in the depthwise phase).
    // RexNet-200: 3x3 s1 C324 @56x56 (7-row bands) 582 vs 661 us and 3x3 s2 C192 @112x112 714 vs 771 us (win: its 32->192
    // expand alone costs 0.5 ms).
    const int rows = m->fuse_band ? fused_band_rows(h, w, g.cin, g.cout, d.k, d.stride) : 0;
    const bool measured_win = d.k == 3 && ((d.stride == 1 && rows >= 7) || (d.stride == 2 && w >= 112 && g.cout >= 192));
    return rows > 0 && (m->fuse_band == 1 || measured_win);
}

// expand GEMM -> depthwise -> SE -> gated projection on a whole-image tile: one kernel (mbconv_block.hip)
static bool can_fuse_block(const mi355_model* m, size_t i, int h, int w, int nb) {
    if (!m->fuse_block || nb < m->fuse_block_min_batch || i + 3 >= m->def.ops.size()) return false;
    const Op& g = m->def.ops[i];
    const Op& d = m->def.ops[i + 1];
    const Op& s = m->def.ops[i + 2];
    const Op& p = m->def.ops[i + 3];
    if (g.kind != OP_GEMM || d.kind != OP_DW || s.kind != OP_SE || p.kind != OP_GEMM) return false;
    if (g.out != SLOT_E || d.in != SLOT_E || d.out != SLOT_D || p.in != SLOT_D || !p.use_gate || !d.pool) return false;
    if (g.use_gate || g.res != SLOT_NONE || g.a_relu6 || !g.tap.empty() || !d.tap.empty()) return false;
    // (rexnet: ReLU6 behind the gate, a shortcut over the first res_channels outputs and channel counts padded to 8 are all
    //  handled by the kernel: pad channels carry exact zeros through every phase, as in the unfused chain)
    if (p.act != ACT_NONE || (p.res != SLOT_NONE && p.res != g.in)) return false;
    if (p.res != SLOT_NONE && p.res_channels && ((p.res_channels + 7) & ~7) != g.cin) return false;
    return mbconv_block_supported(h, w, g.cin, g.cout, p.cout, d.k, d.stride, s.rd, g.act, d.act);
}

// diagnosis buffer [ops][B][16] of cycle buckets (option "block_stamps"); *out stays null when the option is off
static int stamp_ptr(ExecCtx& cx, size_t oi, long long** out) {
    mi355_model* m = cx.m;
    *out = nullptr;
    if (!m->block_stamps) return OK;
    const size_t need = m->def.ops.size() * (size_t)cx.B * 16 * sizeof(long long);
    if (m->stamp_bytes < need) {
        if (m->stamp_buf) MI355_CHECK_HIP(hipFree(m->stamp_buf));
        MI355_CHECK_HIP(hipMalloc((void**)&m->stamp_buf, need));
        m->stamp_bytes = need;
    }
    m->stamp_B = cx.B;
    *out = m->stamp_buf + (oi * (size_t)cx.B + cx.b0) * 16;
    MI355_CHECK_HIP(hipMemsetAsync(*out, 0, (size_t)cx.nb * 16 * sizeof(long long), cx.st));
    return OK;
}

static int exec_block(ExecCtx& cx, size_t oi) {
    mi355_model* m = cx.m;
    SlotState* S = m->slots;
    const Op& g = m->def.ops[oi];
    const Op& d = m->def.ops[oi + 1];
    const Op& s = m->def.ops[oi + 2];
    const Op& p = m->def.ops[oi + 3];
    BlockArgs a{};
    a.X = (const bf16_t*)cx.slot_ptr(g.in);
    a.We = (const bf16_t*)cx.w(g.w_off); a.be = (const float*)cx.w(g.b_off);
    a.Wd = (const bf16_t*)cx.w(d.w_off); a.bd = (const float*)cx.w(d.b_off);
    a.W1 = (const bf16_t*)cx.w(s.w3_off); a.b1 = (const float*)cx.w(s.b_off);
    a.W2 = (const bf16_t*)cx.w(s.w4_off); a.b2 = (const float*)cx.w(s.b2_off);
    a.Wp = (const bf16_t*)cx.w(p.w_off); a.bp = (const float*)cx.w(p.b_off);
    a.D = (bf16_t*)cx.slot_ptr(d.out);
    a.Y = (bf16_t*)cx.slot_ptr(p.out);
    a.H = S[g.in].h; a.W = S[g.in].w; a.Cin = g.cin; a.Kp = (g.cin + 31) & ~31; a.mid = g.cout;
    a.Ho = S[d.out].h; a.Wo = S[d.out].w; a.Cout = p.cout; a.Kp2 = (p.cin + 31) & ~31; a.rd = s.rd;
    a.has_res = p.res != SLOT_NONE;
    a.res_n = p.res != SLOT_NONE ? (p.res_channels ? ((p.res_channels + 7) & ~7) : p.cout) : 0;
    a.a_relu6 = p.a_relu6;
    a.act_e = g.act; a.act_d = d.act; a.se_act = s.se_act;
    a.inv_hw = 1.0f / (float)(a.Ho * a.Wo);
    a.norot = m->block_norot;
    a.variant = m->block_variant;
    a.stamps = nullptr;
    { const int rc = stamp_ptr(cx, oi, &a.stamps); if (rc != OK) return rc; }
    return launch_mbconv_block(a, cx.nb, d.k, d.stride, cx.st);
}

static int exec_fused(ExecCtx& cx, const Op& g, const Op& d) {
    mi355_model* m = cx.m;
    SlotState* S = m->slots;
    FusedArgs a{};
    a.X = (const bf16_t*)cx.slot_ptr(g.in);
    a.We = (const bf16_t*)cx.w(g.w_off); a.be = (const float*)cx.w(g.b_off);
    a.Wd = (const bf16_t*)cx.w(d.w_off); a.bd = (const float*)cx.w(d.b_off);
    a.D = (bf16_t*)cx.slot_ptr(d.out);
    a.pool = d.pool ? (float*)cx.slot_ptr(SLOT_POOLPART) : nullptr;
    a.H = S[g.in].h; a.W = S[g.in].w; a.Cin = g.cin; a.Kp = (g.cin + 31) & ~31; a.mid = g.cout;
    a.Ho = S[d.out].h; a.Wo = S[d.out].w; a.act_e = g.act; a.act_d = d.act;
    a.debug_skip = m->fuse_debug;
    if (fused_late_supported(a.H, a.W, g.cin, g.cout, d.k, d.stride)) {
        m->pool_nblk = 1;
        return launch_fused_late(a, cx.nb, d.k, d.stride, cx.st);
    }
    if (use_sweep(m, g, d, a.H, a.W)) {
        SweepArgs sa{};
        sa.X = a.X; sa.We = a.We; sa.be = a.be; sa.Wd = a.Wd; sa.bd = a.bd; sa.D = a.D; sa.pool = a.pool;
        sa.H = a.H; sa.W = a.W; sa.Cin = a.Cin; sa.Kp = a.Kp; sa.mid = a.mid; sa.Ho = a.Ho; sa.Wo = a.Wo;
        sa.act_e = a.act_e; sa.act_d = a.act_d; sa.csplit_override = m->sweep_csplit; sa.variant = m->sweep_variant; sa.debug_skip = m->sweep_skip;
        { const int rc = stamp_ptr(cx, (size_t)(&g - m->def.ops.data()), &sa.stamps); if (rc != OK) return rc; }
        m->pool_nblk = 1;
        return launch_sweep_mbconv(sa, cx.nb, d.k, d.stride, cx.st);
    }
    a.TH = fused_band_rows(a.H, a.W, g.cin, g.cout, d.k, d.stride);
    m->pool_nblk = cdiv(a.Ho, a.TH);
    return launch_fused_band(a, cx.nb, d.k, d.stride, cx.st);
}

// roctx range label of one executor launch (same wording as mi355_model_profile_ops)
static void op_range_label(const Op& op, int H, int W, int in_h, int in_w, const char* how, char* lab, size_t n) {
    switch (op.kind) {
        case OP_STEM: snprintf(lab, n, "embed/%sstem 3->%d @%dx%d", how, op.cout_real, conv_out(H, 3, 2), conv_out(W, 3, 2)); break;
        case OP_GEMM: snprintf(lab, n, "embed/%spw %d->%d @%dx%d%s%s", how, op.cin_real, op.cout_real, in_h, in_w,
                               op.use_gate ? " gate" : "", op.res != SLOT_NONE ? " res" : ""); break;
        case OP_DW: snprintf(lab, n, "embed/%sdw k%d s%d C%d @%dx%d", how, op.k, op.stride, op.cin_real, in_h, in_w); break;
        case OP_SE: snprintf(lab, n, "embed/%sse C%d rd%d", how, op.cin_real, op.rd); break;
        default: snprintf(lab, n, "embed/%sop%d C%d->%d t%d", how, (int)op.kind, op.cin_real, op.cout_real, op.tokens_h); break;
    }
}

static int run_backbone(ExecCtx& cx, size_t op_begin = 0, size_t op_end = (size_t)-1) {
    mi355_model* m = cx.m;
    // NOTE: slot dims for DW/SE depend on walk order; plan_slots() left the LAST writer's dims in each
    // slot, so re-derive dims incrementally while executing.
    SlotState* S = m->slots;
    if (op_end > m->def.ops.size()) op_end = m->def.ops.size();
    for (size_t oi = op_begin; oi < op_end; ++oi) {
        const Op& op = m->def.ops[oi];
        const int op_index = (int)oi;
        const int in_h = S[op.in == SLOT_NONE ? 0 : op.in].h, in_w = S[op.in == SLOT_NONE ? 0 : op.in].w;
        // (the whole-block kernel is chosen by the caller's WHOLE batch, not by the chunk: it rounds differently from the unfused chain)
        const bool block = op.in != SLOT_NONE && can_fuse_block(m, oi, in_h, in_w, cx.B);
        const bool fused = block || can_fuse(m, oi, in_h, in_w);
        if (fused) {   // dims of the (virtual) expand output and of the depthwise output
            const Op& d = m->def.ops[oi + 1];
            S[op.out].h = S[op.in].h; S[op.out].w = S[op.in].w; S[op.out].c = op.cout;
            S[d.out].h = conv_out(S[op.in].h, d.k, d.stride); S[d.out].w = conv_out(S[op.in].w, d.k, d.stride);
            S[d.out].c = d.cout;
            if (block) {
                const Op& pj = m->def.ops[oi + 3];
                S[pj.out].h = S[d.out].h; S[pj.out].w = S[d.out].w; S[pj.out].c = pj.cout;
            }
        } else
        switch (op.kind) {
            case OP_STEM: S[op.out].h = conv_out(cx.H, 3, 2); S[op.out].w = conv_out(cx.W, 3, 2); S[op.out].c = op.cout; break;
            case OP_GEMM: S[op.out].h = S[op.in].h; S[op.out].w = S[op.in].w; S[op.out].c = op.cout; break;
            case OP_DW: {
                const int ho = conv_out(S[op.in].h, op.k, op.stride), wo = conv_out(S[op.in].w, op.k, op.stride);
                S[op.out].h = ho; S[op.out].w = wo; S[op.out].c = op.cout;
                break;
            }
            case OP_SE: break;
            default: if (op.out != SLOT_NONE) { S[op.out].h = op.tokens_h; S[op.out].w = op.tokens_h; S[op.out].c = op.cout; } break;
        }
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (m->profile) {
            MI355_CHECK_HIP(hipEventCreate(&e0));
            MI355_CHECK_HIP(hipEventCreate(&e1));
            MI355_CHECK_HIP(hipEventRecord(e0, cx.st));
        }
        {
            char lab[192] = "";
            if (roctx_active())
                op_range_label(op, cx.H, cx.W, in_h, in_w, block ? "block: " : (fused ? "fused: " : ""), lab, sizeof lab);
            RoctxRange range(lab);
            if (block) {
                if (int e = exec_block(cx, oi)) return e;
            } else if (fused) {
                if (int e = exec_fused(cx, op, m->def.ops[oi + 1])) return e;
            } else if (int e = exec_op(cx, op)) return e;
        }
        if (m->profile) {
            MI355_CHECK_HIP(hipEventRecord(e1, cx.st));
            m->prof_events.push_back({op_index, {e0, e1}});
            if (m->prof_fused.size() < m->def.ops.size()) m->prof_fused.resize(m->def.ops.size(), 0);
            m->prof_fused[op_index] = fused ? 1 : 0;
        }
        if (block) {
            const Op& pj = m->def.ops[oi + 3];
            if (m->taps && !pj.tap.empty())
                if (int e = record_tap(cx, pj)) return e;
            oi += 3;       // depthwise, SE and projection ran inside the block kernel
            continue;
        }
        if (m->taps && !op.tap.empty())
            if (int e = record_tap(cx, op)) return e;
        if (fused) ++oi;   // the depthwise op was executed together with the expand
    }
    return OK;
}

struct U8Source {                     // uint8 images in front of the stem (mi355_model_forward_u8)
    const unsigned char* img = nullptr;
    int h = 0, w = 0, fill = 255;
    float mean[3] = {0.f, 0.f, 0.f}, stdv[3] = {1.f, 1.f, 1.f};
    const float* conv_w = nullptr;
};

static int forward_impl(mi355_model* m, const float* x, int B, int H, int W, float* out, float* pooled_out,
                        bool features_only, hipStream_t st, const U8Source* u8 = nullptr) {
    MI355_REQUIRE(m, "forward: null model");
    MI355_REQUIRE(m->packed, "forward: weights not packed (call mi355_model_pack after set_tensor)");
    MI355_REQUIRE((x || u8) && out, "forward: null input/output pointer");
    MI355_REQUIRE(B >= 1 && H >= 32 && W >= 32, "forward: bad shape B=%d H=%d W=%d", B, H, W);
    const ModelDef& d = m->def;
    if (d.pools_in_features) MI355_REQUIRE(H == 224 && W == 224, "forward: %s needs 224x224 input", d.arch.c_str());
    // Chunking.  "lanes" > 1: the batch is cut into that many chunks which run CONCURRENTLY on internal streams (forked from
    // and joined back into the caller's stream with events), each with its own copy of the arena: the early layers are
    // HBM-bound and the late ones VALU-bound, so two half-batches a few kernels apart keep both busy where one
    // batch alternates between them.  Results do not depend on the chunking: every kernel is batch-position invariant, and the two
    // kernel choices that round differently from their alternatives (whole-block kernel vs the unfused chain, split-K vs the serial
    // K loop) are decided by the caller's whole batch B, never by the chunk.  (They DO depend on B itself: an image embedded alone
    // and the same image inside a batch of 256 agree to bf16 rounding, not bit for bit - tests/test_effnet_gpu.py.)
    int nl = m->lanes;
    if (nl > 4) nl = 4;
    if (nl < 1 || B < 32 * nl || m->profile || m->taps) nl = 1;
    int mb = (m->microbatch > 0 && m->microbatch < B) ? m->microbatch : B;
    if (nl > 1) mb = (B + nl - 1) / nl;
    const size_t bytes = plan_slots(m, mb, H, W, B);
    m->lane_bytes = nl > 1 ? align_up(bytes, 4096) : 0;
    if (int e = ensure_arena(m, nl > 1 ? m->lane_bytes * nl : bytes)) return e;
    if (nl > 1) {
        if (!m->lane_fork) MI355_CHECK_HIP(hipEventCreateWithFlags(&m->lane_fork, hipEventDisableTiming));
        for (int l = 0; l < nl; ++l) {
            if (!m->lane_stream[l]) MI355_CHECK_HIP(hipStreamCreateWithFlags(&m->lane_stream[l], hipStreamNonBlocking));
            if (!m->lane_join[l]) MI355_CHECK_HIP(hipEventCreateWithFlags(&m->lane_join[l], hipEventDisableTiming));
        }
        MI355_CHECK_HIP(hipEventRecord(m->lane_fork, st));
    }
    hipStream_t caller_st = st;
    int lane = 0;
    const int D = d.feat_dim, Dp = d.feat_dim_pad;
    for (int b0 = 0; b0 < B; b0 += mb, ++lane) {
        const int nb = std::min(mb, B - b0);
        if (nl > 1) {
            st = m->lane_stream[lane];
            MI355_CHECK_HIP(hipStreamWaitEvent(st, m->lane_fork, 0));
        }
        ExecCtx cx{m, st, nb, H, W, x ? x + (size_t)b0 * 3 * H * W : nullptr, b0, B};
        cx.lane = nl > 1 ? lane : 0;
        if (u8) {
            cx.x_u8 = u8->img + (size_t)b0 * u8->h * u8->w * 3;
            cx.img_h = u8->h; cx.img_w = u8->w; cx.fill = u8->fill; cx.conv_w = u8->conv_w;
            for (int c = 0; c < 3; ++c) { cx.mean[c] = u8->mean[c]; cx.stdv[c] = u8->stdv[c]; }
        }
        const bool want_logits = !features_only && d.num_classes > 0;
        // Pooled embedding wanted and the last op is the head 1x1 conv: conv + bias + act + GAP as ONE kernel (k_head_gap: the
        // head tensor never reaches HBM).  forward_features, taps and per-op profiling need the un-pooled map: two-kernel path.
        bool head_fused = false, backbone_done = false;
        if (!features_only && !d.pools_in_features && m->fuse_head_gap && !m->taps && !m->profile && !d.ops.empty()) {
            const Op& h = d.ops.back();
            if (h.kind == OP_GEMM && h.out == d.final_slot && h.in != SLOT_NONE && !h.use_gate && h.res == SLOT_NONE && !h.a_relu6 &&
                h.ln_w_name.empty()) {
                if (int e = run_backbone(cx, 0, d.ops.size() - 1)) return e;
                const SlotState& I = m->slots[h.in];
                const int hwi = I.h * I.w;
                if (head_gap_supported(hwi, h.cout, h.cin, h.cin, (h.cin + 31) & ~31, h.act)) {
                    RoctxRange range(roctx_active() ? "embed/head 1x1 + global average pool" : "");
                    if (int e = launch_head_gap((const bf16_t*)cx.slot_ptr(h.in), h.cin, (const bf16_t*)cx.w(h.w_off), (h.cin + 31) & ~31,
                                                (const float*)cx.w(h.b_off), (float*)cx.slot_ptr(SLOT_POOLED),
                                                want_logits ? (bf16_t*)cx.slot_ptr(SLOT_POOLED_BF16) : nullptr, Dp, nb, hwi, h.cout, h.cin,
                                                h.act, st))
                        return e;
                    head_fused = true;
                } else if (int e = run_backbone(cx, d.ops.size() - 1)) return e;
                backbone_done = true;
            }
        }
        if (!backbone_done)
            if (int e = run_backbone(cx)) return e;
        const SlotState& F = m->slots[d.final_slot];
        const int hw = F.h * F.w;
        if (d.pools_in_features) {
            // swin: final op wrote pooled fp32 (+bf16) into SLOT_POOLED / SLOT_POOLED_BF16
            float* pooled = (float*)cx.slot_ptr(SLOT_POOLED);
            if (!want_logits)
                MI355_CHECK_HIP(hipMemcpy2DAsync(out + (size_t)b0 * D, (size_t)D * 4, pooled, (size_t)Dp * 4, (size_t)D * 4,
                                                 nb, hipMemcpyDeviceToDevice, st));
            if (pooled_out)
                MI355_CHECK_HIP(hipMemcpy2DAsync(pooled_out + (size_t)b0 * D, (size_t)D * 4, pooled, (size_t)Dp * 4,
                                                 (size_t)D * 4, nb, hipMemcpyDeviceToDevice, st));
        } else {
            if (features_only) {
                if (int e = launch_nhwc_to_nchw_f32((const bf16_t*)cx.slot_ptr(d.final_slot),
                                                    out + (size_t)b0 * D * hw, nb, hw, F.c, D, st))
                    return e;
            }
            const bool need_pool = !features_only || pooled_out;
            if (need_pool) {
                float* pooled = (float*)cx.slot_ptr(SLOT_POOLED);   // [nb][Dp]
                if (!head_fused)
                    if (int e = launch_gap((const bf16_t*)cx.slot_ptr(d.final_slot), pooled,
                                           want_logits ? (bf16_t*)cx.slot_ptr(SLOT_POOLED_BF16) : nullptr, nb, hw, F.c, st))
                        return e;
                if (!features_only && !want_logits)
                    MI355_CHECK_HIP(hipMemcpy2DAsync(out + (size_t)b0 * D, (size_t)D * 4, pooled, (size_t)Dp * 4,
                                                     (size_t)D * 4, nb, hipMemcpyDeviceToDevice, st));
                if (pooled_out)
                    MI355_CHECK_HIP(hipMemcpy2DAsync(pooled_out + (size_t)b0 * D, (size_t)D * 4, pooled, (size_t)Dp * 4,
                                                     (size_t)D * 4, nb, hipMemcpyDeviceToDevice, st));
            }
        }
        if (want_logits) {
            const Op& c = d.classifier;
            GemmArgs a{};
            a.A = (const bf16_t*)cx.slot_ptr(SLOT_POOLED_BF16); a.lda = Dp;
            a.W = (const bf16_t*)cx.w(c.w_off); a.ldw = (c.cin + 31) & ~31;
            a.bias = (const float*)cx.w(c.b_off);
            a.out = out + (size_t)b0 * d.num_classes; a.ldo = d.num_classes; a.out_f32 = 1;
            a.M = nb; a.N = d.num_classes; a.K = c.cin; a.act = ACT_NONE; a.rows_per_img = 1; a.res_n = a.N;
            a.zeros = (const bf16_t*)cx.w(0);
            if (int e = launch_gemm_bf16(a, st)) return e;
        }
        if (nl > 1) {
            MI355_CHECK_HIP(hipEventRecord(m->lane_join[lane], st));
            MI355_CHECK_HIP(hipStreamWaitEvent(caller_st, m->lane_join[lane], 0));
        }
    }
    return OK;
}

}  // namespace mi355

// ====================================================================================== C ABI
extern "C" {

int mi355_model_create(const char* name, int num_classes, mi355_model_t* out) {
    MI355_REQUIRE(name && out, "model_create: null argument");
    MI355_REQUIRE(num_classes >= 0, "model_create: num_classes=%d must be >= 0", num_classes);
    mi355_model* m = new mi355_model();
    m->def.arch = name;
    m->def.num_classes = num_classes;
    const std::string n = name;
    int e;
    if (n == "efficientnet_b3a" || n == "efficientnet_b3") e = build_efficientnet_b3(m->def);
    else if (n == "rexnet_100") e = build_rexnet(m->def, 1.0);
    else if (n == "rexnet_130") e = build_rexnet(m->def, 1.3);
    else if (n == "rexnet_150") e = build_rexnet(m->def, 1.5);
    else if (n == "rexnet_200") e = build_rexnet(m->def, 2.0);
    else if (n == "swin_base_patch4_window7_224") e = build_swin_base(m->def);
    else {
        // same wording as the reference's guard (train/train.py:400)
        set_error("Unknown model name '%s'. Known: efficientnet_b3a, rexnet_100/130/150/200, swin_base_patch4_window7_224", name);
        e = ERR_ARG;
    }
    if (e) { delete m; return e; }
    *out = m;
    return OK;
}

void mi355_model_destroy(mi355_model_t m) {
    if (!m) return;
    if (m->dev_blob) (void)hipFree(m->dev_blob);
    if (m->arena) (void)hipFree(m->arena);
    if (m->stamp_buf) (void)hipFree(m->stamp_buf);
    for (int l = 0; l < 4; ++l) {
        if (m->lane_stream[l]) (void)hipStreamDestroy(m->lane_stream[l]);
        if (m->lane_join[l]) (void)hipEventDestroy(m->lane_join[l]);
    }
    if (m->lane_fork) (void)hipEventDestroy(m->lane_fork);
    for (auto& kv : m->tapbufs)
        if (kv.second.ptr) (void)hipFree(kv.second.ptr);
    delete m;
}

int mi355_model_num_tensors(mi355_model_t m) { return m ? (int)m->def.tensors.size() : 0; }

int mi355_model_tensor_info(mi355_model_t m, int i, const char** name, int* ndim, int64_t shape[4], int* kind) {
    MI355_REQUIRE(m && i >= 0 && i < (int)m->def.tensors.size(), "tensor_info: index %d out of range", i);
    const TensorSpec& t = m->def.tensors[i];
    if (name) *name = t.name.c_str();
    if (ndim) *ndim = (int)t.shape.size();
    if (shape)
        for (int d = 0; d < 4; ++d) shape[d] = d < (int)t.shape.size() ? t.shape[d] : 1;
    if (kind) *kind = t.kind;
    return OK;
}

int mi355_model_feature_dim(mi355_model_t m) { return m ? m->def.feat_dim : 0; }
int mi355_model_num_classes(mi355_model_t m) { return m ? m->def.num_classes : 0; }

int mi355_model_set_tensor(mi355_model_t m, const char* name, const float* host_data, int64_t numel) {
    MI355_REQUIRE(m && name, "set_tensor: null argument");
    auto it = m->def.index.find(name);
    MI355_REQUIRE(it != m->def.index.end(), "set_tensor: unexpected key '%s' for %s", name, m->def.arch.c_str());
    TensorSpec& t = m->def.tensors[it->second];
    if (t.kind == 2) return OK;  // int64 buffers carry no arithmetic
    MI355_REQUIRE(host_data, "set_tensor: null data for '%s'", name);
    MI355_REQUIRE(numel == t.numel(), "set_tensor: size mismatch for '%s': got %lld elements, expected %lld", name,
                  (long long)numel, (long long)t.numel());
    t.data.assign(host_data, host_data + numel);
    t.set = true;
    m->packed = false;
    return OK;
}

int mi355_model_pack(mi355_model_t m, void* stream) {
    MI355_REQUIRE(m, "pack: null model");
    m->blob.clear();
    m->blob.resize(256, 0);   // zero page at offset 0 (source of out-of-range DMA chunks in k_gemm_big)
    Packer pk{m, m->blob};
    for (Op& op : m->def.ops)
        if (int e = pack_op(pk, op)) return e;
    if (m->def.num_classes > 0)
        if (int e = pack_gemm(pk, m->def.classifier)) return e;
    const size_t bytes = align_up(m->blob.size(), 256);
    m->blob.resize(bytes, 0);
    int dev = 0;
    MI355_CHECK_HIP(hipGetDevice(&dev));
    if (bytes > m->dev_blob_bytes || dev != m->blob_device) {   // (re-)allocate on the CURRENT device
        if (m->dev_blob) MI355_CHECK_HIP(hipFree(m->dev_blob));
        m->dev_blob = nullptr;
        MI355_CHECK_HIP(hipMalloc(&m->dev_blob, bytes));
        m->dev_blob_bytes = bytes;
        m->blob_device = dev;
    }
    hipStream_t st = (hipStream_t)stream;
    MI355_CHECK_HIP(hipMemcpyAsync(m->dev_blob, m->blob.data(), bytes, hipMemcpyHostToDevice, st));
    MI355_CHECK_HIP(hipStreamSynchronize(st));  // blob is pageable host memory
    m->packed = true;
    return OK;
}

int mi355_model_forward_features(mi355_model_t m, const float* x, int B, int H, int W, float* out, float* pooled_out,
                                 void* stream) {
    return forward_impl(m, x, B, H, W, out, pooled_out, true, (hipStream_t)stream);
}

int mi355_model_forward(mi355_model_t m, const float* x, int B, int H, int W, float* out, float* pooled_out,
                        void* stream) {
    return forward_impl(m, x, B, H, W, out, pooled_out, false, (hipStream_t)stream);
}

// Parity tool: run ONLY the ops behind tap `from_tap` up to and including the op that records tap `to_tap`, on an activation
// supplied by the caller (x: [B][C][h][w] fp32 NCHW, rounded to bf16 on the way in - feed it the oracle's bf16-rounded tap
// of the previous layer).  Taps are recorded as in a normal forward (enable them first), so each layer / block can be
// compared with the oracle on the ORACLE's input: errors do not compound through the network.
int mi355_model_run_between_taps(mi355_model_t m, const char* from_tap, const char* to_tap, const float* x, int B, int C,
                                 int h, int w, void* stream) {
    MI355_REQUIRE(m && from_tap && to_tap && x, "run_between_taps: null argument");
    MI355_REQUIRE(m->packed, "run_between_taps: weights not packed");
    MI355_REQUIRE(B >= 1 && C >= 1 && h >= 1 && w >= 1, "run_between_taps: bad shape");
    const auto& ops = m->def.ops;
    size_t i0 = ops.size(), i1 = ops.size();
    for (size_t i = 0; i < ops.size(); ++i) {
        if (ops[i].tap == from_tap) i0 = i;
        if (ops[i].tap == to_tap) i1 = i;
    }
    MI355_REQUIRE(i0 < ops.size() && i1 < ops.size() && i0 < i1, "run_between_taps: taps '%s' -> '%s' not found in that order",
                  from_tap, to_tap);
    const Op& src = ops[i0];
    MI355_REQUIRE(src.out != SLOT_NONE && (src.cout_real ? src.cout_real : src.cout) == C,
                  "run_between_taps: tap '%s' has %d channels, got %d", from_tap, src.cout_real ? src.cout_real : src.cout, C);
    hipStream_t st = (hipStream_t)stream;
    // size the arena as for a full forward whose maps are at least as large as the ones reached from here
    int H = 32, W = 32;
    for (const Op& op : ops) { (void)op; }
    {   // input size that gives the tap this resolution: walk the strides in front of it
        int sh = 1;
        for (size_t i = 0; i <= i0; ++i)
            if (ops[i].kind == OP_STEM || (ops[i].kind == OP_DW && ops[i].stride == 2)) sh *= 2;
        H = h * sh; W = w * sh;
    }
    const size_t bytes = plan_slots(m, B, H, W);
    m->lane_bytes = 0;
    if (int e = ensure_arena(m, bytes)) return e;
    ExecCtx cx{m, st, B, H, W, nullptr, 0, B};
    SlotState* S = m->slots;
    S[src.out].h = h; S[src.out].w = w; S[src.out].c = src.cout;
    if (int e = launch_nchw_f32_to_nhwc_bf16(x, (bf16_t*)cx.slot_ptr(src.out), B, h * w, C, src.cout, st)) return e;
    return run_backbone(cx, i0 + 1, i1 + 1);
}

int mi355_model_forward_u8(mi355_model_t m, const unsigned char* images, int B, int h, int w, int fill, const float* mean,
                           const float* stdv, const float* conv_input_w, int features_only, float* out, float* pooled_out,
                           void* stream) {
    MI355_REQUIRE(m && images && mean && stdv && out, "forward_u8: null pointer");
    MI355_REQUIRE(B >= 1 && h >= 1 && w >= 1 && h <= 16384 && w <= 16384, "forward_u8: bad image size %dx%d", h, w);
    MI355_REQUIRE(fill >= 0 && fill <= 255, "forward_u8: fill must be a byte value");
    for (int c = 0; c < 3; ++c) MI355_REQUIRE(stdv[c] != 0.f, "forward_u8: std[%d] is zero", c);
    MI355_REQUIRE(!m->def.ops.empty() && (m->def.ops[0].kind == OP_STEM || m->def.ops[0].kind == OP_PATCH_EMBED),
                  "forward_u8: %s has no stem / patch embedding to fuse the pre-processing into", m->def.arch.c_str());
    MI355_REQUIRE(m->def.ops[0].kind == OP_STEM || !conv_input_w, "forward_u8: conv_input belongs to the convolutional backbones");
    U8Source u{};
    u.img = images; u.h = h; u.w = w; u.fill = fill; u.conv_w = conv_input_w;
    for (int c = 0; c < 3; ++c) { u.mean[c] = mean[c]; u.stdv[c] = stdv[c]; }
    const int S = h > w ? h : w;
    return forward_impl(m, nullptr, B, S, S, out, pooled_out, features_only != 0, (hipStream_t)stream, &u);
}

int mi355_model_enable_taps(mi355_model_t m, int enable) {
    MI355_REQUIRE(m, "enable_taps: null model");
    m->taps = enable != 0;
    return OK;
}

int mi355_model_read_tap(mi355_model_t m, const char* tap_name, float* out, int64_t out_numel, int64_t shape[4],
                         void* stream) {
    MI355_REQUIRE(m && tap_name, "read_tap: null argument");
    auto it = m->tapbufs.find(tap_name);
    MI355_REQUIRE(it != m->tapbufs.end(), "read_tap: no tap named '%s' was recorded", tap_name);
    const TapBuf& t = it->second;
    if (shape) { shape[0] = t.B; shape[1] = t.c_real; shape[2] = t.h; shape[3] = t.w; }
    if (!out) return OK;
    MI355_REQUIRE(out_numel >= (int64_t)t.B * t.c_real * t.h * t.w, "read_tap: output too small");
    return launch_nhwc_to_nchw_f32((const bf16_t*)t.ptr, out, t.B, t.h * t.w, t.c, t.c_real, (hipStream_t)stream);
}

int mi355_model_set_option(mi355_model_t m, const char* key, int64_t value) {
    MI355_REQUIRE(m && key, "set_option: null argument");
    const std::string k = key;
    if (k == "microbatch") m->microbatch = (int)value;
    else if (k == "lanes") m->lanes = (int)value;
    else if (k == "fuse") m->fuse = value != 0;
    else if (k == "fuse_band") m->fuse_band = (int)value;
    else if (k == "fuse_sweep") m->fuse_sweep = (int)value;
    else if (k == "sweep_csplit") m->sweep_csplit = (int)value;
    else if (k == "sweep_variant") m->sweep_variant = (int)value;
    else if (k == "sweep_skip") m->sweep_skip = (int)value;
    else if (k == "fuse_debug") m->fuse_debug = (int)value;
    else if (k == "fuse_ln") m->fuse_ln = (int)value;
    else if (k == "fuse_block") m->fuse_block = (int)value;
    else if (k == "fuse_block_min_batch") m->fuse_block_min_batch = (int)value;
    else if (k == "block_stamps") m->block_stamps = value != 0;
    else if (k == "block_norot") m->block_norot = (int)value;
    else if (k == "block_variant") m->block_variant = (int)value;
    else if (k == "fuse_head_gap") m->fuse_head_gap = value != 0;
    else if (k == "roctx") roctx_enable(value != 0);   // process-wide: ranges around every executor op and rank phase
    else if (k == "profile") {
        m->profile = value != 0;
        for (int i = 0; i < PK_COUNT; ++i) { m->prof_ms[i] = 0; m->prof_launches[i] = 0; }
        m->prof_op_ms.assign(m->def.ops.size(), 0.0);
        m->prof_op_n.assign(m->def.ops.size(), 0);
    } else {
        set_error("set_option: unknown option '%s'", key);
        return ERR_ARG;
    }
    return OK;
}

int mi355_model_profile_read(mi355_model_t m, double* ms_by_kind, int64_t* launches_by_kind, int n) {
    MI355_REQUIRE(m && ms_by_kind && launches_by_kind && n >= PK_COUNT, "profile_read: need arrays of >= %d", PK_COUNT);
    for (auto& pe : m->prof_events) {
        MI355_CHECK_HIP(hipEventSynchronize(pe.second.second));
        float ms = 0.f;
        MI355_CHECK_HIP(hipEventElapsedTime(&ms, pe.second.first, pe.second.second));
        int kd = prof_kind(m->def.ops[pe.first]);
        if (m->fuse && m->def.ops[pe.first].kind == OP_GEMM && m->def.ops[pe.first].out == SLOT_E &&
            (size_t)pe.first < m->prof_fused.size() && m->prof_fused[pe.first]) kd = PK_FUSED;
        m->prof_ms[kd] += ms;
        m->prof_launches[kd] += 1;
        if (m->prof_op_ms.size() < m->def.ops.size()) { m->prof_op_ms.resize(m->def.ops.size(), 0.0); m->prof_op_n.resize(m->def.ops.size(), 0); }
        m->prof_op_ms[pe.first] += ms;
        m->prof_op_n[pe.first] += 1;
        (void)hipEventDestroy(pe.second.first);
        (void)hipEventDestroy(pe.second.second);
    }
    m->prof_events.clear();
    for (int i = 0; i < PK_COUNT; ++i) { ms_by_kind[i] = m->prof_ms[i]; launches_by_kind[i] = m->prof_launches[i]; }
    return OK;
}

// Per-op view of the profile + traffic model: for op i (plan order) the average launch time (ms), its
// algorithmic bytes at batch B, its kind and a short label.  Call after mi355_model_profile_read.
int mi355_model_profile_ops(mi355_model_t m, int B, int H, int W, int max_ops, double* avg_ms, double* bytes,
                            int* kinds, char* labels, int label_stride) {
    MI355_REQUIRE(m && avg_ms && bytes && kinds, "profile_ops: null argument");
    const int n = (int)m->def.ops.size();
    MI355_REQUIRE(max_ops >= n, "profile_ops: need room for %d ops", n);
    SlotState S[SLOT_COUNT];
    for (int i = 0; i < n; ++i) {
        const Op& op = m->def.ops[i];
        avg_ms[i] = (i < (int)m->prof_op_n.size() && m->prof_op_n[i]) ? m->prof_op_ms[i] / m->prof_op_n[i] : 0.0;
        kinds[i] = prof_kind(op);
        double by = 0;
        char lab[128] = "";
        switch (op.kind) {
            case OP_STEM: {
                const int ho = conv_out(H, 3, 2), wo = conv_out(W, 3, 2);
                S[op.out].h = ho; S[op.out].w = wo;
                by = (double)B * (3.0 * H * W * 4 + (double)ho * wo * op.cout_real * 2);
                snprintf(lab, sizeof lab, "stem 3->%d @%dx%d", op.cout_real, ho, wo);
                break;
            }
            case OP_GEMM: {
                const double hw = (double)S[op.in].h * S[op.in].w;
                S[op.out].h = S[op.in].h; S[op.out].w = S[op.in].w;
                double el = hw * (op.cin_real + op.cout_real);
                if (op.res != SLOT_NONE) el += hw * (op.res_channels ? op.res_channels : op.cout_real);
                by = B * el * 2;
                snprintf(lab, sizeof lab, "pw %d->%d @%dx%d%s%s", op.cin_real, op.cout_real, S[op.in].h, S[op.in].w,
                         op.use_gate ? " gate" : "", op.res != SLOT_NONE ? " res" : "");
                break;
            }
            case OP_DW: {
                const int ho = conv_out(S[op.in].h, op.k, op.stride), wo = conv_out(S[op.in].w, op.k, op.stride);
                by = (double)B * ((double)S[op.in].h * S[op.in].w + (double)ho * wo) * op.cin_real * 2;
                snprintf(lab, sizeof lab, "dw k%d s%d C%d @%dx%d", op.k, op.stride, op.cin_real, S[op.in].h, S[op.in].w);
                S[op.out].h = ho; S[op.out].w = wo;
                break;
            }
            case OP_SE: snprintf(lab, sizeof lab, "se C%d rd%d", op.cin_real, op.rd); break;
            default: {
                if (op.out != SLOT_NONE) { S[op.out].h = op.tokens_h; S[op.out].w = op.tokens_h; }
                by = (double)B * op.tokens_h * op.tokens_h * (op.cin_real + op.cout_real) * 2;
                snprintf(lab, sizeof lab, "op%d C%d->%d t%d", (int)op.kind, op.cin_real, op.cout_real, op.tokens_h);
                break;
            }
        }
        bytes[i] = by;
        if (labels && label_stride > 0) { strncpy(labels + (size_t)i * label_stride, lab, label_stride - 1); labels[(size_t)i * label_stride + label_stride - 1] = 0; }
    }
    return n;
}

// Layer-granular algorithmic traffic: every conv/dw/1x1 layer reads its input once and
// writes its output once; BN/act/SE-gate/GAP are epilogues; each residual adds one read of the block input.
int mi355_model_traffic_kinds(mi355_model_t m, int B, int H, int W, double* bytes_by_kind, double* macs_by_kind, int n) {
    MI355_REQUIRE(m && bytes_by_kind && macs_by_kind && n >= PK_COUNT, "traffic_kinds: need arrays of >= %d", PK_COUNT);
    for (int i = 0; i < PK_COUNT; ++i) { bytes_by_kind[i] = 0; macs_by_kind[i] = 0; }
    SlotState S[SLOT_COUNT];
    int fused_left = 0;   // > 0 while walking the two ops of a pair the executor runs as one fused kernel
    for (size_t oi = 0; oi < m->def.ops.size(); ++oi) {
        const Op& op = m->def.ops[oi];
        int kd = prof_kind(op);
        if (fused_left == 0 && op.in != SLOT_NONE && can_fuse_block(m, oi, S[op.in].h, S[op.in].w, B)) fused_left = 4;
        if (fused_left == 0 && op.in != SLOT_NONE && can_fuse(m, oi, S[op.in].h, S[op.in].w)) fused_left = 2;
        if (fused_left > 0) { kd = PK_FUSED; --fused_left; }
        switch (op.kind) {
            case OP_STEM: {
                const int ho = conv_out(H, 3, 2), wo = conv_out(W, 3, 2);
                S[op.out].h = ho; S[op.out].w = wo;
                bytes_by_kind[kd] += (double)B * (3.0 * H * W * 4 + (double)ho * wo * op.cout_real * 2);
                macs_by_kind[kd] += (double)B * ho * wo * op.cout_real * 27;
                break;
            }
            case OP_GEMM: {
                const double hw = (double)S[op.in].h * S[op.in].w;
                S[op.out].h = S[op.in].h; S[op.out].w = S[op.in].w;
                double el = hw * (op.cin_real + op.cout_real);
                if (op.res != SLOT_NONE) el += hw * (op.res_channels ? op.res_channels : op.cout_real);
                bytes_by_kind[kd] += B * el * 2;
                macs_by_kind[kd] += B * hw * op.cin_real * op.cout_real;
                break;
            }
            case OP_DW: {
                const int ho = conv_out(S[op.in].h, op.k, op.stride), wo = conv_out(S[op.in].w, op.k, op.stride);
                bytes_by_kind[kd] += (double)B * ((double)S[op.in].h * S[op.in].w + (double)ho * wo) * op.cin_real * 2;
                macs_by_kind[kd] += (double)B * ho * wo * op.cin_real * op.k * op.k;
                S[op.out].h = ho; S[op.out].w = wo;
                break;
            }
            case OP_SE:
                macs_by_kind[kd] += (double)B * 2.0 * op.cin_real * op.rd;
                break;
            default: {
                if (op.out != SLOT_NONE) { S[op.out].h = op.tokens_h; S[op.out].w = op.tokens_h; }
                const double t = (double)op.tokens_h * op.tokens_h;
                if (op.kind == OP_PATCH_EMBED) bytes_by_kind[kd] += (double)B * (3.0 * H * W * 4 + t * op.cout_real * 2);
                else if (op.kind == OP_TOKEN_MEAN) bytes_by_kind[kd] += B * t * op.cin_real * 2;
                else bytes_by_kind[kd] += B * t * (op.cin_real + op.cout_real) * 2;
                if (op.kind == OP_WINATTN) macs_by_kind[kd] += B * t * 49.0 * op.cout_real * 2;   // QK^T + PV
                if (op.kind == OP_PATCH_EMBED) macs_by_kind[kd] += B * t * 48.0 * op.cout_real;
                break;
            }
        }
    }
    return OK;
}

int mi355_model_traffic(mi355_model_t m, int B, int H, int W, double* act_bytes, double* weight_bytes, double* macs) {
    MI355_REQUIRE(m, "traffic: null model");
    double by[PK_COUNT], mc[PK_COUNT];
    if (int e = mi355_model_traffic_kinds(m, B, H, W, by, mc, PK_COUNT)) return e;
    double tb = 0, tm = 0;
    for (int i = 0; i < PK_COUNT; ++i) { tb += by[i]; tm += mc[i]; }
    if (act_bytes) *act_bytes = tb;
    if (macs) *macs = tm;
    if (weight_bytes) *weight_bytes = (double)m->blob.size();
    return OK;
}

// Diagnosis: per-phase cycle counts of the whole-block kernel (option "block_stamps"), averaged over the images of the
// last forward.  out[op][16]: cycle buckets of wave 0 (the list is at the end of k_mbconv_block).
// Synchronises the device.  Returns the number of ops (rows) or a negative error.
int mi355_model_block_stamps(mi355_model_t m, double* out, int max_ops) {
    MI355_REQUIRE(m && out, "block_stamps: null argument");
    const int n = (int)m->def.ops.size();
    MI355_REQUIRE(max_ops >= n, "block_stamps: need room for %d ops", n);
    for (int i = 0; i < n * 16; ++i) out[i] = 0.0;
    if (!m->stamp_buf || m->stamp_B <= 0) return n;
    MI355_CHECK_HIP(hipDeviceSynchronize());
    std::vector<long long> h((size_t)n * m->stamp_B * 16);
    MI355_CHECK_HIP(hipMemcpy(h.data(), m->stamp_buf, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i)
        for (int b = 0; b < m->stamp_B; ++b)
            for (int j = 0; j < 16; ++j) out[i * 16 + j] += (double)h[((size_t)i * m->stamp_B + b) * 16 + j] / m->stamp_B;
    return n;
}

int mi355_gemm_bf16(const void* A, const void* W, const float* bias, void* out, int M, int N, int K, int ldw, int act,
                    void* stream) {
    MI355_REQUIRE(A && W && bias && out, "gemm_bf16: null pointer");
    GemmArgs a{};
    a.A = (const bf16_t*)A; a.lda = K; a.W = (const bf16_t*)W; a.ldw = ldw; a.bias = bias;
    a.out = out; a.ldo = N; a.out_f32 = 0; a.M = M; a.N = N; a.K = K; a.act = act; a.rows_per_img = 1; a.res_n = N;
    static void* zero_page[MI355_MAX_DEVICES] = {nullptr};   // a 256-byte zero page per device for the DMA kernel's out-of-range chunks
    int dev = 0;
    MI355_CHECK_HIP(hipGetDevice(&dev));
    MI355_REQUIRE(dev >= 0 && dev < MI355_MAX_DEVICES, "gemm_bf16: device ordinal %d out of range", dev);
    if (!zero_page[dev]) {
        MI355_CHECK_HIP(hipMalloc(&zero_page[dev], 256));
        MI355_CHECK_HIP(hipMemset(zero_page[dev], 0, 256));
    }
    a.zeros = (const bf16_t*)zero_page[dev];
    return launch_gemm_bf16(a, (hipStream_t)stream);
}

int mi355_pool_linear(const float* fm, int B, int C, int HW, const float* weight, const float* bias, int N, float* out,
                      float* pooled_out, void* stream) {
    MI355_REQUIRE(fm && (out || pooled_out), "pool_linear: null pointer");
    MI355_REQUIRE(B >= 1 && C >= 1 && HW >= 1, "pool_linear: bad shape B=%d C=%d HW=%d", B, C, HW);
    MI355_REQUIRE(!weight || (N >= 1 && out), "pool_linear: a weight needs N >= 1 and an output");
    return launch_pool_linear(fm, weight, bias, out, pooled_out, B, C, HW, weight ? N : 0, (hipStream_t)stream);
}

int mi355_square_pad_normalize(const unsigned char* img, int h, int w, int fill, const float* mean, const float* stdv,
                               float* out, void* stream) {
    MI355_REQUIRE(img && mean && stdv && out, "square_pad_normalize: null pointer");
    MI355_REQUIRE(h >= 1 && w >= 1 && h <= 16384 && w <= 16384, "square_pad_normalize: bad image size %dx%d", h, w);
    MI355_REQUIRE(fill >= 0 && fill <= 255, "square_pad_normalize: fill must be a byte value");
    for (int c = 0; c < 3; ++c) MI355_REQUIRE(stdv[c] != 0.f, "square_pad_normalize: std[%d] is zero", c);
    return launch_square_pad_normalize(img, h, w, fill, mean, stdv, out, (hipStream_t)stream);
}

int mi355_conv_input_silu(const float* x, const float* w, int B, int H, int W, float* out, void* stream) {
    MI355_REQUIRE(x && w && out, "conv_input_silu: null pointer");
    MI355_REQUIRE(B >= 1 && H >= 1 && W >= 1, "conv_input_silu: bad shape");
    return launch_conv_input_silu(x, w, B, H, W, out, (hipStream_t)stream);
}

}  // extern "C"
