// Executor-side structures shared by model.hip and swin_kernels.hip: the model object behind the C ABI,
// the per-forward execution context and the weight packer.
#pragma once
#include "model.h"

#include <string.h>

namespace mi355 {

// implemented in swin_kernels.hip
int swin_exec(const ModelDef& def, const Op& op, struct ExecCtx& cx);

static inline uint16_t f2bf_host(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf_round_host(float f) {
    uint32_t u = ((uint32_t)f2bf_host(f)) << 16;
    float r;
    memcpy(&r, &u, 4);
    return r;
}

struct SlotState {
    size_t off = 0, bytes = 0;
    int h = 0, w = 0, c = 0;
};

struct TapBuf {
    void* ptr = nullptr;
    size_t bytes = 0;
    int B = 0, h = 0, w = 0, c = 0, c_real = 0;
};

enum { PK_STEM = 0, PK_GEMM, PK_DW, PK_SE, PK_OTHER, PK_ATTN, PK_LN, PK_FUSED, PK_COUNT };

}  // namespace mi355

using namespace mi355;

struct mi355_model {
    ModelDef def;
    std::vector<char> blob;
    void* dev_blob = nullptr;
    size_t dev_blob_bytes = 0;
    bool packed = false;
    void* arena = nullptr;
    size_t arena_bytes = 0;
    int arena_device = -1, blob_device = -1;   // HIP device ordinals the arena / packed weights were allocated on
    SlotState slots[SLOT_COUNT];
    int microbatch = 0;
    int lanes = 1;              // option "lanes": chunks of a forward run concurrently on this many internal HIP streams (1 = caller's stream only)
    hipStream_t lane_stream[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t lane_fork = nullptr, lane_join[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t lane_bytes = 0;      // arena bytes per lane
    int fuse_sweep = 1;         // row-sweep kernel (sweep_mbconv.hip, MFMA depthwise) for the early-stage shapes it supports:
                                // 0 never, 1 (default) the shape classes measured faster than the alternatives
    int sweep_variant = 0, sweep_skip = 0;   // tuning / diagnosis knobs of the row-sweep kernel
    int sweep_csplit = 0;       // tuning: workgroups per image in the row-sweep kernel (0 = launcher's choice)
    int fuse_band = 2;          // band variant for the early stages: 0 never, 1 wherever it fits, 2 (default) only the shape
                                // classes where it was measured faster than the unfused pair (see can_fuse in model.hip)
    bool fuse = true;           // fused expand+depthwise for whole-image tiles (option "fuse")
    int fuse_block = 1;         // whole MBConv block in one kernel for the 14x14 / 7x7 stages (option "fuse_block"; 0 = off)
    int fuse_block_min_batch = 96;  // ... only when the caller's whole batch has at least this many images (one workgroup per image: measured on
                                    // EfficientNet-B3a in round 3, tools/bench_thresholds.py: B = 64 2.20 vs 1.98 ms unfused, B = 96 2.50 vs 2.66, B = 128 2.65 vs 2.86)
                                    // (measured B=128: 3.16 ms with, 3.01 without; B=256: 4.38 with, 4.7 without; B<=32: +0.4 ms)
    bool fuse_head_gap = true;  // option "fuse_head_gap": head 1x1 conv + GAP as one kernel when the pooled embedding is all the caller wants
    int block_variant = 0;      // tuning (option "block_variant"): see BlockArgs::variant
    int block_norot = 0;        // diagnosis (option "block_norot"): see BlockArgs::norot
    bool block_stamps = false;  // option "block_stamps": record per-phase cycle counts of the block kernel
    long long* stamp_buf = nullptr; size_t stamp_bytes = 0; int stamp_B = 0;
    int fuse_debug = 0;
    int fuse_ln = 1;            // swin: LayerNorm as a statistics pass + the consumer GEMM's epilogue (option "fuse_ln"; 0 = separate LayerNorm kernel)
    int pool_nblk = 0;          // squeeze partials per image produced by the last depthwise stage
    bool taps = false;
    std::map<std::string, TapBuf> tapbufs;
    // per-kind profiling with hipEvents (option "profile")
    bool profile = false;
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> prof_events;  // (op index, events)
    std::vector<char> prof_fused;   // op index -> executed as a fused expand+depthwise pair
    std::vector<double> prof_op_ms;
    std::vector<long> prof_op_n;
    double prof_ms[PK_COUNT] = {0};
    long prof_launches[PK_COUNT] = {0};
};

namespace mi355 {

struct ExecCtx {
    mi355_model* m;
    hipStream_t st;
    int nb, H, W;          // chunk batch, input size
    const float* x;        // chunk input (NCHW fp32), or null when the chunk comes as uint8 images:
    int b0, B;             // chunk offset / full batch (for taps)
    int lane = 0;          // which arena copy / internal stream this chunk uses
    const unsigned char* x_u8 = nullptr;   // [nb][img_h][img_w][3] (fused pre-processing + stem)
    int img_h = 0, img_w = 0, fill = 255;
    float mean[3] = {0.f, 0.f, 0.f}, stdv[3] = {1.f, 1.f, 1.f};
    const float* conv_w = nullptr;          // optional conv_input weights (device)
    int ln_pending_in = SLOT_NONE;          // a fused LayerNorm left its row statistics in SLOT_LNSTATS: the next GEMM reads this slot instead
    char* base() const { return (char*)m->arena + (size_t)lane * m->lane_bytes; }
    void* slot_ptr(int s) const { return s == SLOT_NONE ? nullptr : base() + m->slots[s].off; }
    const char* w(size_t off) const { return (const char*)m->dev_blob + off; }
};

// ------------------------------------------------------------------------------------ packing
struct Packer {
    mi355_model* m;
    std::vector<char>& blob;
    size_t alloc(size_t bytes) {
        const size_t off = align_up(blob.size(), 256);
        blob.resize(off + bytes, 0);
        return off;
    }
    const TensorSpec* get(const std::string& name) {
        auto it = m->def.index.find(name);
        if (it == m->def.index.end()) { set_error("pack: unknown tensor '%s'", name.c_str()); return nullptr; }
        const TensorSpec& t = m->def.tensors[it->second];
        if (!t.set) { set_error("pack: tensor '%s' was never set (mi355_model_set_tensor)", name.c_str()); return nullptr; }
        return &t;
    }
    // per-output-channel (scale, shift) of an eval-mode BN, in the same fp32 op order as the oracle
    bool bn_fold(const std::string& bn, float eps, int n, std::vector<float>& scale, std::vector<float>& shift) {
        scale.assign(n, 1.f);
        shift.assign(n, 0.f);
        if (bn.empty()) return true;
        const TensorSpec *g = get(bn + ".weight"), *b = get(bn + ".bias"), *mu = get(bn + ".running_mean"),
                         *var = get(bn + ".running_var");
        if (!g || !b || !mu || !var) return false;
        for (int i = 0; i < n; ++i) {
            const float s = g->data[i] / sqrtf(var->data[i] + eps);
            scale[i] = s;
            shift[i] = b->data[i] - mu->data[i] * s;
        }
        return true;
    }
};

int pack_gemm(Packer& pk, Op& op);

}  // namespace mi355
