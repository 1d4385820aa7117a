// One whole MBConv block per launch for the late stages (14x14 and 7x7 maps): per image ONE 8-wave workgroup runs
//   1x1 expand (MFMA) + bias + act  ->  depthwise kxk + bias + act  ->  SE squeeze / FC / FC / sigmoid
//   ->  [ReLU6] gated 1x1 projection (MFMA) + bias (+ residual).
// gfx950 only.  Replaces the timm InvertedResidual.forward (efficientnet_b3a) / LinearBottleneck.forward (rexnet) reached
// from inference/inference.py:199-201 for the 14x14 / 7x7 blocks.
//
// Why one kernel: at these sizes every separate kernel starts from a cold L2 and is latency-bound (the gated
// projections ran at ~1 TB/s, the SE kernel at 10-30 us of pure latency).  Here the expanded tensor E lives only in
// LDS (a 128-channel slab at a time), the block input X stays in LDS for the whole block (expand operand), the
// depthwise output D makes one round trip through the L2 it was just written to (it cannot stay in LDS: the SE gate
// needs every channel's spatial mean before the projection can start), and the SE squeeze is complete inside the
// workgroup (fixed summation order, no partials in HBM).
//
// Round 3 structure of the front half (expand + depthwise), from the round-2 measurements (2 waves per SIMD, 38 % of the
// wave time parked at barriers / waits, MFMA pipe 19 % busy):
//   * a slab is 8 waves x ONE 16-channel tile; a wave expands all pixels of ITS channels and its depthwise phase consumes
//     exactly those channels of the E image, so the slab loop has NO workgroup barrier (LDS operations of one wave execute
//     in order) and the eight waves drift apart: one wave's MFMA chain runs beside its SIMD partner's SiLU / stores;
//   * 16-channel slabs per wave also balance the tail (1392 channels = 87 tiles = 11 slabs with ONE idle wave slot, against
//     6 slabs of 2 tiles with four idle waves in the last one);
//   * depthwise: the B fragments of the NEXT pixel tile are requested as each MFMA of the current tile retires its
//     fragment (a 13-deep rotating register set: every LDS read has a whole tile of MFMAs to land - the round-2 loop read
//     four fragments ahead and stalled on LDS latency in front of every group);
//   * expand: X fragments are read two k-steps ahead of the MFMAs that consume them;
//   * the activations are template parameters (no switch inside the pixel-tile loops).
//
// What bounds it (tools/microbench_valu.hip): the vector issue port.  v_exp_f32 / v_rcp_f32 hold it for 8 cycles, plain
// VALU for 4, a 16x16x32 MFMA for 8 of its 16; the two SiLUs per expanded element and the MFMA issue slots add up to
// ~55 k cycles per 7x7 C1392 block per SIMD, the matrix pipe alone to ~40 k.
#include "ops.h"

namespace mi355 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int MB_THREADS = 512;    // 8 waves: 256 VGPRs per lane, enough to keep every global load several steps ahead
constexpr int MB_WAVES = MB_THREADS / 64;
constexpr int MB_MAX_RD = 192;         // SE hidden units (efficientnet_b3a <= 96, rexnet_200 <= 173)
constexpr int MB_MC = MB_WAVES * 16;      // expanded channels per slab: one 16-channel tile per wave

__device__ __forceinline__ float mb_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float mb_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup release/acquire fence: hipcc drains
// vmcnt(0) in front of it, i.e. every wave would wait for its depthwise-output stores and for every prefetched weight
// fragment.  Used in the SE / projection phases, where nothing but LDS is exchanged between the waves.
__device__ __forceinline__ void mb_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Projection decomposition: 16-column output tiles per wave (1..NTW), chosen to keep the most waves busy (ties: more tiles per
// wave, so that every A fragment read from LDS feeds more MFMAs).
__host__ __device__ static inline int mb_proj_ntw(int NTp, int MTp, int NTW) {
    int best = 1, best_busy = 0;
    for (int ntw = 1; ntw <= NTW; ++ntw) {
        const int nwn = (NTp + ntw - 1) / ntw;
        if (nwn > MB_WAVES) continue;
        int msplit = MB_WAVES / nwn;
        if (msplit > MTp) msplit = MTp;
        const int busy = nwn * msplit;
        if (busy >= best_busy) { best_busy = busy; best = ntw; }
    }
    return best;
}

// Column tiles per wave the launcher and the support check use: the best split with at most three (every shape of round 2);
// 14x14 maps whose 13 ... 16 column tiles would then leave a wave more than `mwp` row tiles take four (4 x 2 waves of 7 row tiles).
__host__ __device__ static inline int mb_pick_ntw(int NTp, int MTp, int wi, int mwp) {
    const int ntw = mb_proj_ntw(NTp, MTp, 3);
    if (wi != 14) return ntw;
    const int nwn = (NTp + ntw - 1) / ntw;
    int msplit = nwn <= MB_WAVES ? MB_WAVES / nwn : 1;
    if (msplit > MTp) msplit = MTp;
    if (nwn <= MB_WAVES && (MTp + msplit - 1) / msplit <= mwp) return ntw;
    return 4;
}

// Tap pairs of the MFMA depthwise (phase 2): one 16x16x32 MFMA covers TWO taps x 16 channels of K.  Pairs are vertical
// (ky, kx) + (ky+1, kx) for ky = 0, 2, .. and horizontal along the last row, so that the second tap of a pair is always
// "+1 E row" or "+1 E pixel" from the first: the per-lane part of the LDS address is then one of two bases and the
// per-pair part a compile-time immediate.
template <int KS> struct MbTaps {
    static constexpr int KK = KS * KS;
    static constexpr int NV = (KS / 2) * KS;                 // vertical pairs
    static constexpr int NH = (KS + 1) / 2;                  // horizontal pairs in the last row
    static constexpr int NP = NV + NH;
    __host__ __device__ static constexpr bool vertical(int tp) { return tp < NV; }
    __host__ __device__ static constexpr int tap_a(int tp) {
        return tp < NV ? (tp / KS) * 2 * KS + tp % KS : (KS - 1) * KS + (tp - NV) * 2;
    }
    __host__ __device__ static constexpr int tap_b(int tp) {         // -1: no second tap
        return tp < NV ? tap_a(tp) + KS : ((tp - NV) * 2 + 1 < KS ? tap_a(tp) + 1 : -1);
    }
};

// Geometry classes (template parameters):
//   WI   image width of the block input (14 or 7): makes every LDS offset of phase 2 a compile-time immediate
//   MW   16-pixel tiles per pass of the expand GEMM, MH passes (pass h covers tiles h * MW .. h * MW + MW - 1 with the same
//        weight fragments): 4 x 1 for 7x7, 7 x 2 for 14x14
//   NTW  max 16-column output tiles per wave in the projection, MWP max 16-row tiles per wave there
//   KST    k-steps of the expand GEMM = Kp / 32, exact (a compile-time trip count: no branches inside the k-loop); a wave
//          holds a whole slab's weight fragments in registers, requested one slab ahead, and the k-loop issues no loads
//   A_IT   16-byte pieces of a projection A chunk (128 deep, 256 for the 7x7 class) staged per thread
//   ACT_E / ACT_D  activation after the expand / the depthwise conv (SiLU / SiLU: efficientnet; SiLU / none: rexnet)
template <int KS, int S, int WI, int MW, int MH, int NTW, int MWP, int KST, int A_IT, int ACT_E, int ACT_D>
__global__ __launch_bounds__(MB_THREADS) void k_mbconv_block(const BlockArgs a) {
    using TP = MbTaps<KS>;
    constexpr int MC = MB_MC;                         // expanded channels per slab
    constexpr int PAD = KS / 2;
    constexpr int NP = TP::NP;
    constexpr int EW = WI + 2 * PAD;                  // E image: zero columns left/right AND zero rows above/below
    constexpr int WO = (WI + 2 * PAD - KS) / S + 1;   // output width
    // E image of ONE wave: [E row][RP pixels][16 channels] bf16 (32 B per pixel), the eight waves' images side by side.
    // Depthwise pixel tiles are ROW ALIGNED - one output row of up to 16 pixels (14x14 maps), or two rows of up to 8 (7x7
    // maps: lanes 0-7 / 8-15) - and the row pitch RP is a multiple of 8 pixels for the two-row tiles: the 16 lanes of a
    // ds_read_b128 group (lanes {0-3, 12-15} with one 8-channel half, lanes {4-11} with the other, MI355X_MICROARCH LDS table)
    // then cover 16 different 16-byte slots of the 256-byte bank row.  (Round 2 tiled 16 CONSECUTIVE output pixels over a
    // [pixel][slab channels] image: at the row wraps of a 7- or 14-wide map two or three lanes of a group met in one slot -
    // PMC: 42 % of the LDS cycles of these kernels were bank-conflict cycles.)
    constexpr int TR = WO <= 8 ? 2 : 1;               // output rows per depthwise pixel tile
    constexpr int RP = (TR == 2 && S == 1) ? ((EW + 7) & ~7) : EW;   // E row pitch in pixels (stride-2 tiles read every other pixel: 2-way either way)
    static_assert(WO <= 16, "depthwise tile geometry");
    // Projection flavour: the 7x7 blocks whose waves own two column tiles stage a whole super-chunk of A and run a barrier-free K
    // loop with an 8-deep weight ring (6 x 232->1392 blocks: 99 -> 96 us); every other shape keeps the double-buffered K chunks
    // (the single staging pass exposed its L2 round trips where two to four super-chunks were needed: 14x14 +8..10 us)
    constexpr bool PROJ_WHOLE = WI == 7 && NTW == 2;
    constexpr int KC = WI == 7 ? 256 : 128;           // chunked flavour: K chunk (7x7 class: 64 rows, so twice as deep fits)
    constexpr int ALD = KC + 16;                      // ... and its A row stride: rows 32 B (mod 64 B) apart - see mb_pick_xld
    extern __shared__ __attribute__((aligned(16))) unsigned char mb_smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: everything derived from it stays scalar
    const int fr = lane & 15, fq = lane >> 4, fk = fq * 8;
    const int b = blockIdx.x;
    const int P = a.H * a.W, MT = (P + 15) >> 4, XLD = a.XLD;
    const int EH = a.H + 2 * PAD;
    const int EP = (EH + (TR == 2 ? S : 0)) * RP + 8;   // pixels of one wave's E image (+ slack: the odd row of the last two-row tile, lanes past the row end and
                                                    // the unused second tap of the last pair read on)
    const int Pout = a.Ho * a.Wo;
    const int MTo = TR == 2 ? (a.Ho + 1) >> 1 : a.Ho;   // depthwise pixel tiles (row aligned)
    const int midp = (a.mid + 15) & ~15;

    // ---- LDS carve.  X image | squeeze sums (later: gate at the start of the LDS) | SE hidden vector | E slab.
    // The SE partials and the projection's A chunks re-use the space afterwards.
    bf16_t* Xs = reinterpret_cast<bf16_t*>(mb_smem);                       // [MT*16][XLD] (+ a zero tail)
    size_t off = ((size_t)MT * 16 * XLD * 2 + 64 + 15) & ~(size_t)15;
    float* pool = reinterpret_cast<float*>(mb_smem + off);                 // [mid] squeeze sums
    off += (size_t)((a.mid + 255) & ~255) * 4;
    float* rvec = reinterpret_cast<float*>(mb_smem + off);                 // [rd]
    off += MB_MAX_RD * 4;
    unsigned char* R = mb_smem + off;
    bf16_t* Es = reinterpret_cast<bf16_t*>(R) + (size_t)wave * EP * 16;    // this wave's E image [EH*RP + slack][16]

    // optional phase timing (diagnosis): wave-uniform, so the counters live in scalar registers
    // tick(i): add the cycles since the previous tick to bucket i (wave 0's view; buckets are listed at the end)
    // (eight 32-bit buckets: sixteen 64-bit ones cost 34 scalar registers and pushed ~60 SGPR spills into the loops)
    const bool stamping = a.stamps != nullptr;
    unsigned t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned t_last = 0;
    if (stamping) t_last = (unsigned)__builtin_readcyclecounter();
    auto tick = [&](int i) {
        if (stamping) { const unsigned t = (unsigned)__builtin_readcyclecounter(); t_acc[i] += t - t_last; t_last = t; }
    };

    const int nslabs = (a.mid + MC - 1) / MC;
    // every workgroup walks the slabs from a different start so that 256 of them do not stream the same weight lines
    // through the same L2 channels in lock-step (the slabs are independent: order does not change any sum)
    const int rb0 = (a.norot & 1) ? 0 : b, rb1 = (a.norot & 2) ? 0 : b, rb2 = (a.norot & 4) ? 0 : b, rb3 = (a.norot & 8) ? 0 : b;
    auto slab_of = [&](int ci) { return (ci + rb0) % nslabs; };

    // W fragments come straight from L2 in MFMA layout (every element is read once per workgroup), a whole slab (WRING
    // k-steps) per wave; the next slab's are requested before this slab's activation epilogue, so they travel while the
    // VALU works.  Rows past the padded weight matrix are clamped to its last row: such a wave skips both phases.
    // (sched_barrier: under register pressure the scheduler sinks prefetch loads down to their first use, which turns
    //  every prefetch into an exposed round trip; the barrier pins them where they are written)
    u32x4 wq[KST];
    f32x4 be_reg;                           // the slab's expand bias (accumulator init), requested with the weights
    auto w_prefetch_slab = [&](int cbase) {
        const int n = min(cbase + wave * 16 + fr, midp - 1);
#pragma unroll
        for (int h = 0; h < KST; ++h) wq[h] = *reinterpret_cast<const u32x4*>(a.We + (n * a.Kp + fk + h * 32));
        be_reg = *reinterpret_cast<const f32x4*>(a.be + min(cbase + wave * 16 + fq * 4, midp - 4));
        __builtin_amdgcn_sched_barrier(0);
    };
    // depthwise weights of this wave's channel tile: lane (n = lane & 15, tap half = lane >> 5) needs w[tap][c] of ITS
    // channel for both taps of every pair - 2-byte loads requested at the top of phase 1, used in phase 2
    // (two sets: the NEXT slab's are requested in front of this slab's depthwise phase, i.e. before its output stores -
    //  vmcnt retires in order, so everything a slab needs is then older than the stores and a counted wait can leave them in flight)
    unsigned short wd_raw[NP], wd_nxt[NP];
    f32x4 bd_reg, bd_nxt;
    auto wd_fetch = [&](int cbase) {
        const int ch = min(cbase + wave * 16 + fr, a.mid - 1);
#pragma unroll
        for (int tp = 0; tp < NP; ++tp) {
            const int ta = TP::tap_a(tp), tb = TP::tap_b(tp) < 0 ? TP::tap_a(tp) : TP::tap_b(tp);
            const int t = (lane & 32) ? tb : ta;
            wd_nxt[tp] = a.Wd[t * a.mid + ch];
        }
        bd_nxt = *reinterpret_cast<const f32x4*>(a.bd + min(cbase + wave * 16 + fq * 4, a.mid - 4));
        __builtin_amdgcn_sched_barrier(0);
    };

    // ------------------------------------------------------------------ phase 0: X[b] -> LDS, zero the E image
    w_prefetch_slab(slab_of(0) * MC);
    wd_fetch(slab_of(0) * MC);
    {
        const int kc = (a.Cin + 7) >> 3;             // 16-byte pieces per row that hold data
        const int kcl = XLD >> 3;                    // pieces per LDS row (>= kc + 1: at least one zero piece)
        const bf16_t* xb = a.X + (size_t)b * P * a.Cin;
        const int total = MT * 16 * kcl;
        constexpr int U = 9;             // the whole image in flight at once (<= 9 x 16 B per thread, checked on the host)
        for (int id0 = tid; id0 < total; id0 += MB_THREADS * U) {
            u32x4 v[U];
            int dst[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int id = id0 + u * MB_THREADS;
                const int row = id / kcl, c = id - row * kcl;
                // unconditional (clamped) loads: a branch around a load makes hipcc wait vmcnt(0) at every later use
                const u32x4 t = *reinterpret_cast<const u32x4*>(xb + (min(row, P - 1) * a.Cin + min(c, kc - 1) * 8));
                const bool ok = id < total && row < P && c < kc;
                v[u] = ok ? t : (u32x4){0u, 0u, 0u, 0u};
                dst[u] = id < total ? row * XLD + c * 8 : -1;
            }
            if (id0 == tid) {            // zero the E image while the loads travel (pads must be zero; the interior is rewritten)
                bf16_t* Eall = reinterpret_cast<bf16_t*>(R);
                for (int id = tid; id < MB_WAVES * EP * 2; id += MB_THREADS)
                    *reinterpret_cast<u32x4*>(&Eall[id * 8]) = (u32x4){0u, 0u, 0u, 0u};
                if (tid < 4) *reinterpret_cast<u32x4*>(&Xs[MT * 16 * XLD + tid * 8]) = (u32x4){0u, 0u, 0u, 0u};   // zero tail
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (dst[u] >= 0) *reinterpret_cast<u32x4*>(&Xs[dst[u]]) = v[u];
        }
    }
    mb_lds_barrier();
    tick(0);

    bf16_t* Db = a.D + (size_t)b * Pout * a.mid;

    // per-lane constants of phase 2 (element offsets into this wave's E image): this lane's 8-channel half of the pixel, plus the
    // second tap of a pair - one E row down (vertical pairs) or one pixel right (horizontal pairs) - for the upper lane half
    const int e_cv = (fq & 1) * 8 + ((fq >> 1) ? RP * 16 : 0);
    const int e_ch = (fq & 1) * 8 + ((fq >> 1) ? 16 : 0);
    // this lane's output pixel inside a tile: (row within the tile, column); lanes past the row end compute on whatever the
    // E image holds there (always inside the image + slack) and are masked at the store / the squeeze sum
    const int t_row = TR == 2 ? (fr >> 3) : 0, t_col = TR == 2 ? (fr & 7) : fr;
    const int e_lane = (t_row * S * RP + t_col * S) * 16;            // E offset of tap (0,0) in tile 0
    constexpr int E_TILE = TR * S * RP * 16;                          // ... and its step from tile to tile
    const bool col_ok = t_col < WO;

    // diagnosis only (option "block_variant", results are garbage): bit0 skip the expand phase, bit1 the depthwise phase, bit2 the SE
    // FCs, bit3 the projection's K loop, bit4 the weight requests of the slab loop
    const int abl = a.variant;
    bool prev_active = false;
    for (int ci = 0; ci < nslabs; ++ci) {
        const int cbase = slab_of(ci) * MC;
        const int ch0 = cbase + wave * 16;                       // this wave's first channel
        const bool active = ch0 < a.mid;                         // (wave-uniform; false only in the last slab's tail)
        // ---- this slab's weights (expand fragments + bias, depthwise taps + bias) were requested one slab ago, BEFORE the
        // previous slab's depthwise stores: wait for everything but those stores (vmcnt retires in order; the count must not
        // exceed the stores actually issued, so any shape other than the class's square map drains everything), then "read"
        // the registers with an empty asm - they become asm results for hipcc, which otherwise waits vmcnt(0) at the first
        // use of a load issued in an earlier loop iteration.
        {
            constexpr int MTO_SQ = TR == 2 ? (WO + 1) / 2 : WO;        // stores per active wave and slab on the class's square map
            if (ci > 0 && prev_active && MTo == MTO_SQ) {
                if constexpr (MTO_SQ >= 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
                else if constexpr (MTO_SQ >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int h = 0; h < KST; ++h) asm volatile("" : "+v"(wq[h]));
            asm volatile("" : "+v"(be_reg));
            asm volatile("" : "+v"(bd_nxt));
#pragma unroll
            for (int tp = 0; tp < NP; ++tp) { asm volatile("" : "+v"(wd_nxt[tp])); wd_raw[tp] = wd_nxt[tp]; }
            bd_reg = bd_nxt;
        }
        tick(1);

        // ---- phase 1: E slab = act(X W^T + b) -> Es.  D = W x X^T: a lane holds 4 consecutive channels of one pixel.
#pragma unroll
        for (int h = 0; h < MH; ++h) {
            f32x4 acc[MW];
#pragma unroll
            for (int i = 0; i < MW; ++i) acc[i] = be_reg;
            // X fragments of a k-step are read as one batch (tiles past the image are clamped to the last one: their
            // results are never stored), PD k-steps ahead of the MFMAs: four MFMAs (64 matrix cycles) do not cover an LDS
            // round trip, two k-steps of them do
            constexpr int PD = MW <= 4 ? 2 : 1;       // (the 14x14 class has no registers for a third fragment set)
            int arow[MW];
#pragma unroll
            for (int i = 0; i < MW; ++i) arow[i] = (min(h * MW + i, MT - 1) * 16 + fr) * XLD + fk;
            bf16x8 af[PD + 1][MW];
#pragma unroll
            for (int d = 0; d < PD; ++d)
#pragma unroll
                for (int i = 0; i < MW; ++i) af[d][i] = *reinterpret_cast<const bf16x8*>(&Xs[arow[i] + (d < KST ? d : KST - 1) * 32]);
            if (active && !(abl & 1)) {
#pragma unroll
                for (int ks = 0; ks < KST; ++ks) {
                    if (ks + PD < KST) {
#pragma unroll
                        for (int i = 0; i < MW; ++i)
                            af[(ks + PD) % (PD + 1)][i] = *reinterpret_cast<const bf16x8*>(&Xs[arow[i] + (ks + PD) * 32]);
                    }
#pragma unroll
                    for (int i = 0; i < MW; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wq[ks]), af[ks % (PD + 1)][i], acc[i], 0, 0, 0);
                }
            }
            if (h == MH - 1 && ci + 1 < nslabs && !(abl & 16)) {
                // next slab's weights: requested behind this slab's last MFMAs, in front of its depthwise stores
                w_prefetch_slab(slab_of(ci + 1) * MC);
                wd_fetch(slab_of(ci + 1) * MC);
            }
            if (active && !(abl & 1)) {
#pragma unroll
                for (int i = 0; i < MW; ++i) {
                    acc[i].x = act_c<ACT_E>(acc[i].x); acc[i].y = act_c<ACT_E>(acc[i].y);
                    acc[i].z = act_c<ACT_E>(acc[i].z); acc[i].w = act_c<ACT_E>(acc[i].w);
                }
#pragma unroll
                for (int i = 0; i < MW; ++i) {
                    const int mt = h * MW + i;
                    const int p = mt * 16 + fr;                  // this lane's pixel in tile i
                    if (mt < MT && p < P) {
                        const int y = p / WI;
                        const int eoff = (y + PAD) * RP + (p - y * WI) + PAD;
                        u32x2 o;
                        o.x = pack2bf(acc[i].x, acc[i].y);
                        o.y = pack2bf(acc[i].z, acc[i].w);
                        *reinterpret_cast<u32x2*>(&Es[eoff * 16 + fq * 4]) = o;
                    }
                }
            }
        }
        tick(1);
        asm volatile("" ::: "memory");
        tick(1);

        // ---- phase 2: depthwise on the (otherwise idle) matrix pipe.  A depthwise conv is a contraction with a DIAGONAL
        // weight matrix per tap: one 16x16x32 MFMA takes K = 2 taps x 16 channels, A = diag(w[tap][c]) (this wave's 16
        // channels, 1 nonzero per lane), B = 16 pixels x (2 taps x 16 channels) read straight from the E image (one
        // ds_read_b128 per lane, immediate offsets).  1/16 of the MFMA is useful work, which still beats the VALU: 13 MFMAs
        // (208 cycles) replace 1600 VALU cycles per 16x16 outputs.
        if (active && !(abl & 2)) {
            // diagonal weight fragments: lane (n = lane & 15, kg = lane >> 4) holds k = kg*8 + j -> tap (kg >> 1),
            // channel (kg & 1)*8 + j of the tile: nonzero only where that channel is the lane's own n
            u32x4 dwf[NP];
            {
                const bool mine = ((fq & 1) == (fr >> 3)) && (ch0 + fr < a.mid);
#pragma unroll
                for (int tp = 0; tp < NP; ++tp) {
                    const bool has = mine && !((lane & 32) && TP::tap_b(tp) < 0);
                    const unsigned v = has ? (unsigned)wd_raw[tp] : 0u;
                    const unsigned word = (fr & 1) ? (v << 16) : v;
                    const int q = (fr & 7) >> 1;
                    dwf[tp] = (u32x4){q == 0 ? word : 0u, q == 1 ? word : 0u, q == 2 ? word : 0u, q == 3 ? word : 0u};
                }
            }
            auto e_read = [&](int tbase, int tp) -> bf16x8 {
                const int ta = TP::tap_a(tp);
                const int offs = ((ta / KS) * RP + (ta % KS)) * 16;         // compile-time immediate after unrolling
                return *reinterpret_cast<const bf16x8*>(Es + tbase + (TP::vertical(tp) ? e_cv : e_ch) + offs);
            };
            // B fragments: ONE rotating set of NP register quads.  Fragment tp of pixel tile mt + 1 is requested right behind
            // the MFMA that consumed fragment tp of tile mt, so every LDS read has NP MFMAs (80 - 208 matrix cycles) to land
            bf16x8 ef[NP];
#pragma unroll
            for (int tp = 0; tp < NP; ++tp) ef[tp] = e_read(e_lane, tp);
            float psum[4] = {0.f, 0.f, 0.f, 0.f};
            const bool chok = ch0 + fq * 4 < a.mid;
            bf16_t* dlane = Db + ((t_row * a.Wo + t_col) * a.mid + ch0 + fq * 4);
            const size_t d_tile = (size_t)TR * a.Wo * a.mid;
            for (int mt = 0; mt < MTo; ++mt) {
                const int tbn = e_lane + min(mt + 1, MTo - 1) * E_TILE;   // (the last tile re-reads itself: never used)
                f32x4 acc = bd_reg;
#pragma unroll
                for (int tp = 0; tp < NP; ++tp) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&dwf[tp]), ef[tp], acc, 0, 0, 0);
                    ef[tp] = e_read(tbn, tp);
                }
                acc.x = act_c<ACT_D>(acc.x); acc.y = act_c<ACT_D>(acc.y); acc.z = act_c<ACT_D>(acc.z); acc.w = act_c<ACT_D>(acc.w);
                if (col_ok && mt * TR + t_row < a.Ho) {
                    psum[0] += acc.x; psum[1] += acc.y; psum[2] += acc.z; psum[3] += acc.w;
                    if (chok) {
                        u32x2 ov;
                        ov.x = pack2bf(acc.x, acc.y);
                        ov.y = pack2bf(acc.z, acc.w);
                        *reinterpret_cast<u32x2*>(dlane + mt * d_tile) = ov;
                    }
                }
            }
            // squeeze: this wave saw every pixel of its 16 channels - fold the 16 pixel lanes (fixed order), done
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) psum[j] += __shfl_xor(psum[j], o, 64);
            }
            if (fr == 0 && chok)
                *reinterpret_cast<f32x4*>(&pool[ch0 + fq * 4]) = (f32x4){psum[0], psum[1], psum[2], psum[3]};
        }
        prev_active = active;
        tick(2);
        asm volatile("" ::: "memory");   // (the next slab rewrites only this wave's own E channels: no barrier)
    }
    // ---- SE FC1 weights of this wave's first group of four hidden units: requested BEFORE the wait for the slowest wave (they
    // need nothing from the other waves), so the L2 round trip runs under that wait instead of behind it.  Groups of four units
    // are rotated by image (L2 spreading); a unit keeps its slot q = unit % 4 in every image, so that the code that produces it -
    // and with it the last bit of its sum - does not depend on the batch position.
    constexpr int C8MAX = 5;                              // ceil(mid / 8 / 64) <= 5  (mid <= 2560, checked on the host)
    const int ngroups = (abl & 4) ? 0 : (a.rd + 3) >> 2;
    auto fc1_load = [&](int gi, u32x4 (&w)[C8MAX][4], float (&b1v)[4]) {
        const int j0 = ((gi + rb1) % max(ngroups, 1)) * 4;
#pragma unroll
        for (int it = 0; it < C8MAX; ++it) {
            const int c8 = min(lane + it * 64, (a.mid >> 3) - 1);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                w[it][q] = *reinterpret_cast<const u32x4*>(a.W1 + (min(j0 + q, a.rd - 1) * a.mid + c8 * 8));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) b1v[q] = a.b1[min(j0 + q, a.rd - 1)];
        __builtin_amdgcn_sched_barrier(0);
    };
    auto fc1_compute = [&](int gi, const u32x4 (&w)[C8MAX][4], const float (&b1v)[4]) {
        const int j0 = ((gi + rb1) % max(ngroups, 1)) * 4;
        float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int it = 0; it < C8MAX; ++it) {
            const int c8 = lane + it * 64;
            if (c8 * 8 < a.mid) {
                const f32x4 p0 = *reinterpret_cast<const f32x4*>(&pool[c8 * 8]);
                const f32x4 p1 = *reinterpret_cast<const f32x4*>(&pool[c8 * 8 + 4]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    s[q] += mb_lo(w[it][q].x) * p0.x; s[q] += mb_hi(w[it][q].x) * p0.y; s[q] += mb_lo(w[it][q].y) * p0.z; s[q] += mb_hi(w[it][q].y) * p0.w;
                    s[q] += mb_lo(w[it][q].z) * p1.x; s[q] += mb_hi(w[it][q].z) * p1.y; s[q] += mb_lo(w[it][q].w) * p1.z; s[q] += mb_hi(w[it][q].w) * p1.w;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float t = wave_sum(s[q]);
            if (lane == 0 && j0 + q < a.rd) rvec[j0 + q] = apply_act(t * a.inv_hw + b1v[q], a.se_act);
        }
    };
    u32x4 f1a[C8MAX][4], f1b[C8MAX][4];
    float f1ba[4], f1bb[4];
    if (wave < ngroups) fc1_load(wave, f1a, f1ba);
    __syncthreads();      // full fence: the depthwise output (global) is re-read by other waves in the projection
    tick(3);

        const float* gate = reinterpret_cast<const float*>(mb_smem);                                   // [ceil256(mid)]
        bf16_t* As = reinterpret_cast<bf16_t*>(mb_smem + (size_t)((a.mid + 255) & ~255) * 4);          // [Pout][KCS + 16]: one super-chunk of gate * D
        const int MTp = (Pout + 15) >> 4, NTp = (a.Cout + 15) >> 4;
        constexpr int ntw = NTW;                              // column tiles per wave (= mb_proj_ntw(NTp, MTp, 3): the launcher's choice)
        const int nwn = (NTp + ntw - 1) / ntw;                // waves along N
        int msplit = MB_WAVES / nwn;                          // row groups
        if (msplit > MTp) msplit = MTp;
        const int mper = (MTp + msplit - 1) / msplit;         // row tiles per wave (<= MWP)
        const int wn = (wave % nwn + rb3) % nwn, wmh = wave / nwn;   // column group rotated by image: spreads the Wp lines over time
        const bool wactive = wmh < msplit;
        const int mt0 = wmh * mper;
        const int KST2 = a.Kp2 >> 5;
        const int Coutp = (a.Cout + 15) & ~15;
        // Projection operands.  A = gate * D (bf16) is staged into LDS a SUPER-CHUNK of KCS k at a time - the whole K of the
        // 7x7 blocks up to mid = 1536, two or three pieces for the others (the launcher picks the largest KCS that fits the
        // LDS; a multiple of 256 = PR k-steps) - with ONE barrier per super-chunk instead of two per 128 / 256 k: the MFMA K
        // loop of a wave then runs without any barrier, and its weight fragments stream through a ring of PR k-steps in
        // registers (the fragment of k-step g + PR is requested right behind the MFMAs of k-step g, into the registers they
        // just freed).  Round 2 staged double-buffered chunks with a vmcnt(0) + two barriers per chunk: 36 us of the 100 us
        // of a 7x7 block were the projection, against ~5 us of MFMA time.
        const int KCS = a.KCS, ALD2 = KCS + 16, CPR = KCS >> 3;
        const int nsc = (a.Kp2 + KCS - 1) / KCS;
        constexpr int PR = MWP > 4 ? 4 : 8;                         // (the 14x14 class keeps up to 21 accumulator tiles: a shorter ring)
        u32x4 pq[PR][NTW];
        const bf16_t* wrow[NTW];                                     // this lane's row of Wp for each of the wave's column tiles
#pragma unroll
        for (int j = 0; j < NTW; ++j) wrow[j] = a.Wp + ((size_t)min((wn * NTW + j) * 16 + fr, Coutp - 1) * a.Kp2 + fk);
        // (NTW is exact, so the K loop below is branch-free straight-line code per ring round: hipcc then counts its vmcnt waits
        //  per fragment - a conditional load per column tile made it wait for all but the 7 youngest of 24 loads in flight,
        //  and inline-asm loads hidden from its bookkeeping were copied out of their registers before they had landed)
        auto p_fetch = [&](int r, int g) {
            const int ks = min(g, KST2 - 1);                         // clamped: k-steps past the end are skipped by the MFMA loop
#pragma unroll
            for (int j = 0; j < NTW; ++j) pq[r][j] = *reinterpret_cast<const u32x4*>(wrow[j] + ks * 32);
        };

        // ---- chunked flavour (see PROJ_WHOLE)
        const int nchunks = (a.Kp2 + KC - 1) / KC;
        constexpr int CCPR = KC / 8;                           // 16-byte pieces per A row per K chunk
        constexpr int KPC = KC / 32;                           // k-steps per chunk
        const int a_items = Pout * CCPR;
        const int abuf = MTp * 16 * ALD;


        // unconditional, clamped loads (a branch around a load makes hipcc wait vmcnt(0) at its use): rows / columns past
        // the end re-read valid data that is then multiplied by a zero gate or never stored
        u32x4 sreg[A_IT];
        auto a_load = [&](int chunk) {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int id = min(tid + i * MB_THREADS, a_items - 1);
                const int row = id / CCPR, c = id - row * CCPR;
                const int k = min(chunk * KC + c * 8, a.mid - 8);
                sreg[i] = *reinterpret_cast<const u32x4*>(Db + (row * a.mid + k));
            }
        };
        auto a_store = [&](int chunk) {
            bf16_t* dst = As + (chunk & 1) * abuf;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int id = tid + i * MB_THREADS;
                const int row = id / CCPR, c = id - row * CCPR;
                const int k = chunk * KC + c * 8;                    // < ceil256(mid) = pool's padded size
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(&gate[k]);
                const f32x4 g1 = *reinterpret_cast<const f32x4*>(&gate[k + 4]);
                const u32x4 v = sreg[i];
                u32x4 o;
                float e[8] = {mb_lo(v.x) * g0.x, mb_hi(v.x) * g0.y, mb_lo(v.y) * g0.z, mb_hi(v.y) * g0.w,
                              mb_lo(v.z) * g1.x, mb_hi(v.z) * g1.y, mb_lo(v.w) * g1.z, mb_hi(v.w) * g1.w};
                if (a.a_relu6) {                                     // (wave-uniform) RexNet: ReLU6 between the SE gate and the projection
#pragma unroll
                    for (int q = 0; q < 8; ++q) e[q] = fminf(fmaxf(e[q], 0.f), 6.f);
                }
                o.x = pack2bf(e[0], e[1]); o.y = pack2bf(e[2], e[3]);
                o.z = pack2bf(e[4], e[5]); o.w = pack2bf(e[6], e[7]);
                if (id < a_items) *reinterpret_cast<u32x4*>(&dst[row * ALD + c * 8]) = o;
            }
        };
        // weight fragments of one chunk (KPC k-steps x NTW tiles), ONE set: a k-step pair is re-requested for the next
        // chunk as soon as this chunk's MFMAs have consumed it
        u32x4 cq[KPC][NTW];
        auto c_fetch = [&](int chunk, int kk) {
            const int ks = min(chunk * KPC + kk, KST2 - 1);          // clamped: the tail k-steps are skipped by the MFMA loop
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const int n = min((wn * ntw + (j < ntw ? j : 0)) * 16 + fr, Coutp - 1);
                cq[kk][j] = *reinterpret_cast<const u32x4*>(a.Wp + (n * a.Kp2 + ks * 32 + fk));
            }
        };

    // ------------------------------------------------------------------ SE gate (bf16 weights, fp32 math)
    // FC1: a wave per group of four hidden units; all weight loads of a group (<= 5 x 4 x 16 B per lane) are in flight one group
    // ahead of the multiplies (two register sets; the first group was requested in front of the barrier above)
    for (int gi = wave; gi < ngroups; gi += 2 * MB_WAVES) {
        if (gi + MB_WAVES < ngroups) fc1_load(gi + MB_WAVES, f1b, f1bb);
        fc1_compute(gi, f1a, f1ba);
        if (gi + MB_WAVES < ngroups) {
            if (gi + 2 * MB_WAVES < ngroups) fc1_load(gi + 2 * MB_WAVES, f1a, f1ba);
            fc1_compute(gi + MB_WAVES, f1b, f1bb);
        }
    }
    // FC2: thread = (8-channel chunk, slice of the hidden units).  Its first batch of weights needs only addresses: requested
    // in front of the barrier that publishes the hidden vector.
    const int nch = a.mid >> 3;
    int JS = MB_THREADS / nch;
    if (JS > 8) JS = 8;
    if (JS < 1) JS = 1;
    constexpr int NBJ = 16;
    const int f2_ch0 = tid % nch + (tid / nch >= JS ? nch : 0);
    const bool f2_has = f2_ch0 < nch;                          // (nch <= 320 < 512 threads: at most one chunk per thread)
    const int f2_ch = (min(f2_ch0, nch - 1) + rb2 * 7) % nch;  // channel chunks rotated by image (L2 spreading)
    const int f2_js = JS == 1 ? 0 : min(tid / nch, JS - 1);
    const int f2_rd = (abl & 4) ? 0 : a.rd;
    auto fc2_load = [&](int jb, u32x4 (&w)[NBJ]) {
#pragma unroll
        for (int t = 0; t < NBJ; ++t) w[t] = *reinterpret_cast<const u32x4*>(a.W2 + (min(jb + t * JS, a.rd - 1) * a.mid + f2_ch * 8));
        __builtin_amdgcn_sched_barrier(0);
    };
    u32x4 f2a[NBJ], f2b[NBJ];
    if (f2_js < f2_rd) fc2_load(f2_js, f2a);
    mb_lds_barrier();
    tick(4);
    // FC2 (continued): partials through LDS, summed in slice order
    {
        float* part = reinterpret_cast<float*>(R);           // [JS][mid]
        // the gate's bias, requested before anything else of this phase: as a load inside the reduction loop below it waited (vmcnt
        // is in order) for the whole first A chunk of the projection that is requested in between
        float b2v[5];                                        // mid <= 2560 (host check): <= 5 channels per thread
#pragma unroll
        for (int i = 0; i < 5; ++i) b2v[i] = a.b2[min(tid + i * MB_THREADS, a.mid - 1)];
        if (f2_has) {
            float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            // batches of NBJ loads, all in flight a batch ahead of their multiplies (two register sets); slots past rd re-read the
            // last row and multiply by zero - the order of the real terms is unchanged
            auto fc2_mac = [&](int jb, const u32x4 (&w)[NBJ]) {
#pragma unroll
                for (int t = 0; t < NBJ; ++t) {
                    const int j = jb + t * JS;
                    const float r = j < a.rd ? rvec[j] : 0.f;
                    s[0] += mb_lo(w[t].x) * r; s[1] += mb_hi(w[t].x) * r; s[2] += mb_lo(w[t].y) * r; s[3] += mb_hi(w[t].y) * r;
                    s[4] += mb_lo(w[t].z) * r; s[5] += mb_hi(w[t].z) * r; s[6] += mb_lo(w[t].w) * r; s[7] += mb_hi(w[t].w) * r;
                }
            };
            const int step = JS * NBJ;
            for (int jb = f2_js; jb < f2_rd; jb += 2 * step) {
                if (jb + step < f2_rd) fc2_load(jb + step, f2b);
                fc2_mac(jb, f2a);
                if (jb + step < f2_rd) {
                    if (jb + 2 * step < f2_rd) fc2_load(jb + 2 * step, f2a);
                    fc2_mac(jb + step, f2b);
                }
            }
            *reinterpret_cast<f32x4*>(&part[f2_js * a.mid + f2_ch * 8]) = (f32x4){s[0], s[1], s[2], s[3]};
            *reinterpret_cast<f32x4*>(&part[f2_js * a.mid + f2_ch * 8 + 4]) = (f32x4){s[4], s[5], s[6], s[7]};
        }
        mb_lds_barrier();
        // the first projection operands are requested here: they travel while the gate is reduced (and A is staged / zeroed)
        if constexpr (PROJ_WHOLE) {
#pragma unroll
            for (int r = 0; r < PR; ++r) p_fetch(r, r);
        } else {
            a_load(0);
#pragma unroll
            for (int kk = 0; kk < KPC; ++kk) c_fetch(0, kk);
        }
        __builtin_amdgcn_sched_barrier(0);
        // the gate goes to the START of the LDS (the X image is dead: the residual is re-read from L2), so that the
        // projection's A double buffer can take everything behind it
        float* gate_w = reinterpret_cast<float*>(mb_smem);
#pragma unroll
        for (int ci = 0; ci < 5; ++ci) {
            const int c = tid + ci * MB_THREADS;
            if (c >= ((a.mid + 255) & ~255)) break;
            float g = 0.f;                                   // K tail of the last projection chunk: gate 0
            if (c < a.mid) {
                g = b2v[ci];
                for (int js = 0; js < JS; ++js) g += part[js * a.mid + c];
                g = sigmoid_f(g);
            }
            gate_w[c] = g;
        }
        mb_lds_barrier();
    }
    tick(5);

    {
        f32x4 acc[NTW][MWP];
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int n4 = min((wn * ntw + (j < ntw ? j : 0)) * 16 + fq * 4, Coutp - 4);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(a.bp + n4);
#pragma unroll
            for (int i = 0; i < MWP; ++i) acc[j][i] = bb;
        }
        if constexpr (PROJ_WHOLE) {
            // A fragments: row tiles past the end are clamped to the last real row (their results are never stored)
            int prow[MWP];
    #pragma unroll
            for (int i = 0; i < MWP; ++i) prow[i] = min((mt0 + min(i, mper - 1)) * 16 + fr, Pout - 1) * ALD2 + fk;
            tick(6);
            for (int sc = 0; sc < ((abl & 8) ? 0 : nsc); ++sc) {
                // ---- stage k in [sc * KCS, min(Kp2, (sc + 1) * KCS)) of every row: D -> registers (SB pieces in flight per thread) ->
                // gate [-> ReLU6] -> bf16 -> LDS.  k >= mid (the K padding of the weight matrix) is written as zeros.
                {
                    const int k0 = sc * KCS;
                    const int cpr = (min(a.Kp2, k0 + KCS) - k0) >> 3;          // pieces per row that the MFMA loop reads
                    constexpr int SB = 8;                                        // rows in flight per wave
                    // a wave-instruction moves 64 consecutive pieces (1 KB) of ONE row: lane = piece within the column block, the
                    // waves take rows wave, wave + 8, ...; the gate of a lane's 8 channels is read once per column block
                    for (int cb = 0; cb * 64 < cpr; ++cb) {
                        const int c = cb * 64 + lane;
                        const bool cok = c < cpr;
                        const int k = k0 + min(c, cpr - 1) * 8;
                        const bool real = k < a.mid;
                        const int kk = min(k, a.mid - 8);
                        const f32x4 g0 = *reinterpret_cast<const f32x4*>(&gate[kk]);
                        const f32x4 g1 = *reinterpret_cast<const f32x4*>(&gate[kk + 4]);
                        bf16_t* dcol = As + min(c, cpr - 1) * 8;
                        for (int r0 = wave; r0 < Pout; r0 += MB_WAVES * SB) {
                            u32x4 v[SB];
    #pragma unroll
                            for (int u = 0; u < SB; ++u)
                                v[u] = *reinterpret_cast<const u32x4*>(Db + (min(r0 + u * MB_WAVES, Pout - 1) * a.mid + kk));
    #pragma unroll
                            for (int u = 0; u < SB; ++u) {
                                const int row = r0 + u * MB_WAVES;
                                float e[8] = {mb_lo(v[u].x) * g0.x, mb_hi(v[u].x) * g0.y, mb_lo(v[u].y) * g0.z, mb_hi(v[u].y) * g0.w,
                                              mb_lo(v[u].z) * g1.x, mb_hi(v[u].z) * g1.y, mb_lo(v[u].w) * g1.z, mb_hi(v[u].w) * g1.w};
                                if (a.a_relu6) {                                 // (wave-uniform) RexNet: ReLU6 between the SE gate and the projection
    #pragma unroll
                                    for (int q = 0; q < 8; ++q) e[q] = fminf(fmaxf(e[q], 0.f), 6.f);
                                }
                                u32x4 o;
                                o.x = pack2bf(e[0], e[1]); o.y = pack2bf(e[2], e[3]);
                                o.z = pack2bf(e[4], e[5]); o.w = pack2bf(e[6], e[7]);
                                if (!real) o = (u32x4){0u, 0u, 0u, 0u};
                                if (cok && row < Pout) *reinterpret_cast<u32x4*>(dcol + row * ALD2) = o;
                            }
                        }
                    }
                }
                mb_lds_barrier();          // A staged
                tick(6);
                // ---- MFMA K loop over this super-chunk: no barrier, weights from the register ring
                const int g_begin = sc * (KCS >> 5);
                const int g_end = min(KST2, (sc + 1) * (KCS >> 5));
                for (int g0 = g_begin; g0 < g_end; g0 += PR) {
                    const bf16_t* as = As + (g0 - g_begin) * 32;
                    bf16x8 paf[2][MWP];
    #pragma unroll
                    for (int i = 0; i < MWP; ++i) paf[0][i] = *reinterpret_cast<const bf16x8*>(&as[prow[i]]);
    #pragma unroll
                    for (int r = 0; r < PR; ++r) {
                        if (r + 1 < PR) {
    #pragma unroll
                            for (int i = 0; i < MWP; ++i) paf[(r + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(&as[prow[i] + (r + 1) * 32]);
                        }
                        if (wactive && g0 + r < g_end) {
    #pragma unroll
                            for (int i = 0; i < MWP; ++i)
    #pragma unroll
                                for (int j = 0; j < NTW; ++j)
                                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&pq[r][j]), paf[r & 1][i], acc[j][i], 0, 0, 0);
                        }
                        p_fetch(r, g0 + r + PR);     // the same registers, PR k-steps on
                    }
                }
                tick(6);
                if (sc + 1 < nsc) mb_lds_barrier();      // everybody has read this super-chunk: the buffer may be restaged
            }
        } else {
            for (int id = tid; id < 2 * abuf / 8; id += MB_THREADS)
                *reinterpret_cast<u32x4*>(&As[id * 8]) = (u32x4){0u, 0u, 0u, 0u};
            mb_lds_barrier();          // As zeroed
            a_store(0);
            if (nchunks > 1) a_load(1);
            __builtin_amdgcn_sched_barrier(0);
            mb_lds_barrier();
            // One wait point per iteration (the a_store at the top, vmcnt(0)): everything it waits for was requested at least
            // an MFMA phase earlier.
            tick(6);
            for (int ch = 0; ch < ((abl & 8) ? 0 : nchunks); ++ch) {
                if (ch + 1 < nchunks) a_store(ch + 1);     // chunk ch+1: registers -> gated bf16 -> the other LDS buffer (vmcnt(0))
                // every loop-carried register is "read" HERE, right behind the wait (an empty asm that takes and returns it):
                // from now on it is an asm result for hipcc, no longer a pending load, so its later uses do not wait for
                // the requests issued below
    #pragma unroll
                for (int kk = 0; kk < KPC; ++kk)
    #pragma unroll
                    for (int j = 0; j < NTW; ++j) asm volatile("" : "+v"(cq[kk][j]));
                __builtin_amdgcn_sched_barrier(0);
                if (ch + 2 < nchunks) a_load(ch + 2);      // chunk ch+2's rows travel during this chunk's MFMAs
                __builtin_amdgcn_sched_barrier(0);
                tick(6);
                const bf16_t* as = As + (ch & 1) * abuf;
                // A fragments of a k-step as one batch, one k-step ahead of the MFMAs (row tiles past the end are clamped: their
                // results are never stored)
                int prow[MWP];
    #pragma unroll
                for (int i = 0; i < MWP; ++i) prow[i] = (min(mt0 + min(i, mper - 1), MTp - 1) * 16 + fr) * ALD + fk;
                constexpr bool AHEAD = MWP <= 4;                // a second fragment set only where the registers allow it
                bf16x8 paf[AHEAD ? 2 : 1][MWP];
    #pragma unroll
                for (int i = 0; i < MWP; ++i) paf[0][i] = *reinterpret_cast<const bf16x8*>(&as[prow[i]]);
    #pragma unroll
                for (int kk = 0; kk < KPC; ++kk) {
                    if (AHEAD && kk + 1 < KPC) {
    #pragma unroll
                        for (int i = 0; i < MWP; ++i) paf[(kk + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(&as[prow[i] + (kk + 1) * 32]);
                    }
                    if (!AHEAD && kk > 0) {
    #pragma unroll
                        for (int i = 0; i < MWP; ++i) paf[0][i] = *reinterpret_cast<const bf16x8*>(&as[prow[i] + kk * 32]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (wactive && ch * KPC + kk < KST2) {
    #pragma unroll
                        for (int i = 0; i < MWP; ++i)
    #pragma unroll
                            for (int j = 0; j < NTW; ++j)
                                acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&cq[kk][j]), paf[AHEAD ? (kk & 1) : 0][i], acc[j][i], 0, 0, 0);
                    }
                    if ((kk & 1) == 1 && ch + 1 < nchunks) {     // this k-step pair is consumed: request it for the next chunk
                        __builtin_amdgcn_sched_barrier(0);
                        c_fetch(ch + 1, kk - 1);
                        c_fetch(ch + 1, kk);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                tick(6);
                mb_lds_barrier();
                tick(6);
            }
        }
        // epilogue: lane holds 4 consecutive output channels of one pixel; the residual comes back from L2 (all loads
        // requested before the first use)
        if (wactive) {
            bf16_t* Yb = a.Y + (size_t)b * Pout * a.Cout;
            const bf16_t* Xb = a.X + (size_t)b * P * a.Cin;
            u32x2 rr[NTW][MWP];
            if (a.has_res) {
#pragma unroll
                for (int i = 0; i < MWP; ++i)
#pragma unroll
                    for (int j = 0; j < NTW; ++j) {
                        const int m = min((mt0 + i) * 16 + fr, Pout - 1);
                        const int n = min((wn * ntw + j) * 16 + fq * 4, a.res_n - 4);
                        rr[j][i] = *reinterpret_cast<const u32x2*>(Xb + (m * a.Cin + n));
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < MWP; ++i) {
                const int m = (mt0 + i) * 16 + fr;
                if (i < mper && m < Pout) {
#pragma unroll
                    for (int j = 0; j < NTW; ++j) {
                        const int n = (wn * ntw + j) * 16 + fq * 4;
                        if (j < ntw && n < a.Cout) {
                            float v[4] = {acc[j][i].x, acc[j][i].y, acc[j][i].z, acc[j][i].w};
                            if (a.has_res && n < a.res_n) {      // (RexNet: the shortcut covers the first Cin output channels only)
                                v[0] += mb_lo(rr[j][i].x); v[1] += mb_hi(rr[j][i].x); v[2] += mb_lo(rr[j][i].y); v[3] += mb_hi(rr[j][i].y);
                            }
                            u32x2 o;
                            o.x = pack2bf(v[0], v[1]);
                            o.y = pack2bf(v[2], v[3]);
                            *reinterpret_cast<u32x2*>(Yb + (m * a.Cout + n)) = o;
                        }
                    }
                }
            }
        }
    }
    tick(7);
    if (stamping && tid == 0) {
        // 0 X load | 1 expand (weight requests, MFMA loop, activation, E write) | 2 depthwise (MFMA) + squeeze |
        // 3 wait for the slowest wave + fence | 4 SE FC1 | 5 SE FC2 | 6 projection (A staging, MFMAs, barriers) | 7 output epilogue
        long long* o = a.stamps + (size_t)b * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (long long)t_acc[i];
    }
}

// ------------------------------------------------------------------------------------------ host side
struct MbGeom { int wi, mw, mh, ntw, mwp, a_it; };
static bool mb_geom(int H, int W, MbGeom* g) {
    if (W == 7 && H * W <= 64) { *g = {7, 4, 1, 3, 4, 4}; return true; }
    if (W == 14 && H * W <= 208) { *g = {14, 7, 2, 3, 7, 7}; return true; }
    return false;
}
// The instantiated kernels: (depthwise k, stride, map width, expand k-steps Kp / 32, projection column tiles per wave,
// depthwise activation).  Every whole-block shape of the three model families at 224 x 224 is listed; anything else keeps the
// unfused chain (a kernel of this size costs ~2 s of build time and ~40 KB of code object).
//   efficientnet_b3a (SiLU after the depthwise conv): blocks 3.1-3.4 | 4.0 | 4.1-4.4 | 5.0 (stride 2) | 5.1-5.5 | 6.0 | 6.1
//   rexnet_150 / rexnet_200 (linear depthwise, ReLU6 behind the SE gate, 3x3 only): 14x14 blocks 6-10 / 6-8, 7x7 blocks 12-15
#define MB_INSTANCES(X)                                                                                              \
    X(3, 1, 14, 3, 3, ACT_SILU) X(5, 1, 14, 3, 3, ACT_SILU) X(5, 1, 14, 5, 3, ACT_SILU) X(5, 2, 14, 5, 2, ACT_SILU)     \
    X(5, 1, 7, 8, 2, ACT_SILU) X(3, 1, 7, 8, 3, ACT_SILU) X(3, 1, 7, 12, 3, ACT_SILU)                                 \
    X(3, 1, 14, 4, 2, ACT_NONE) X(3, 1, 14, 4, 3, ACT_NONE) X(3, 1, 14, 5, 3, ACT_NONE) X(3, 1, 14, 6, 3, ACT_NONE)   \
    X(3, 1, 7, 7, 2, ACT_NONE) X(3, 1, 7, 8, 2, ACT_NONE) X(3, 1, 7, 8, 3, ACT_NONE) X(3, 1, 7, 9, 3, ACT_NONE)       \
    X(3, 1, 7, 10, 3, ACT_NONE) X(3, 1, 7, 11, 3, ACT_NONE) X(3, 1, 14, 6, 4, ACT_NONE)
static bool mb_instance_ok(int k, int stride, int wi, int kst, int ntw, int act_d) {
#define X(KS, S, WI, KST, NTW, AD) if (k == KS && stride == S && wi == WI && kst == KST && ntw == NTW && act_d == AD) return true;
    MB_INSTANCES(X)
#undef X
    return false;
}

// Projection super-chunk: the largest multiple of the weight ring's depth (8 k-steps = 256 k, 14x14 class: 4 = 128 k) whose A image [Pout][KCS + 16] bf16
// fits behind the gate, then balanced over the pieces.  0 = not even 256 fit.
static int mb_pick_kcs(const BlockArgs& a) {
    const int Pout = a.Ho * a.Wo;
    const int q = a.W == 7 ? 256 : 128;                  // 32 k x the ring depth of the class (PR = 8 / 4 k-steps)
    const size_t pl = (size_t)((a.mid + 255) & ~255) * 4;
    const long avail = 160l * 1024 - (long)pl;
    const int kmax = (int)((avail / Pout / 2 - 16) / q) * q;
    if (kmax < q) return 0;
    const int nsc = (a.Kp2 + kmax - 1) / kmax;
    const int per = (a.Kp2 + nsc - 1) / nsc;
    return (per + q - 1) / q * q;
}

// LDS bytes for a given X row stride (elements); 0 = does not fit
static size_t mb_lds_bytes_xld(const BlockArgs& a, int k, int xld, const MbGeom& g) {
    const int P = a.H * a.W, MT = (P + 15) / 16, pad = k / 2;
    const int wo = a.Wo, ew = a.W + 2 * pad;
    const int stride = (a.H + 2 * pad - k) / (a.Ho > 1 ? a.Ho - 1 : 1) >= 2 ? 2 : 1;
    const int rp = (wo <= 8 && stride == 1) ? ((ew + 7) & ~7) : ew;         // E row pitch (pixels), as in the kernel
    const int EP = (a.H + 2 * pad + (wo <= 8 ? stride : 0)) * rp + 8;
    const size_t xs = ((size_t)MT * 16 * xld * 2 + 64 + 15) & ~(size_t)15, pl = (size_t)((a.mid + 255) & ~255) * 4;
    const size_t fixed = xs + pl + MB_MAX_RD * 4;
    const size_t slab = (size_t)MB_WAVES * EP * 16 * 2;
    const int Pout = a.Ho * a.Wo;
    const int nch = a.mid / 8;
    int JS = MB_THREADS / nch;
    if (JS > 8) JS = 8;
    if (JS < 1) JS = 1;
    const size_t se = (size_t)JS * a.mid * 4;
    const int kcs = mb_pick_kcs(a);
    if (!kcs) return 0;
    const int kc = g.wi == 7 ? 256 : 128, MTp = (Pout + 15) / 16;
    const bool whole = g.wi == 7 && mb_proj_ntw((a.Cout + 15) / 16, MTp, 3) == 2;       // the kernel's PROJ_WHOLE
    const size_t proj = pl + (whole ? (size_t)Pout * (kcs + 16) * 2                      // gate, then one super-chunk of A
                                    : (size_t)2 * MTp * 16 * (kc + 16) * 2);             // ... or the double-buffered K chunks
    size_t r = slab;
    if (se > r) r = se;
    const size_t total = fixed + r > proj ? fixed + r : proj;
    return total <= 160 * 1024 ? total : 0;
}

// X row stride: Kp + 16 elements, i.e. rows 32 bytes (mod 64) apart: the 16 lanes of a ds_read_b128 group are rows
// {0-3, 12-15} at one 16-byte k-quarter and rows {4-11} at the next, and with a slot step of 2 (mod 4) per row the first set
// lands on the even and the second on the odd slots of the bank row - conflict-free (Kp + 8, a step of 1, put rows 11 and 12
// of every group into one slot: each fragment read took two passes).  When that does not fit: the compact ceil8(Cin) + 8
// (the k-steps past Cin then read the zero pad and the next row's first elements against zero weight columns).
static int mb_pick_xld(const BlockArgs& a, int k, const MbGeom& g) {
    const int wide = a.Kp + 16, compact = ((a.Cin + 7) & ~7) + 8;
    if (mb_lds_bytes_xld(a, k, wide, g)) return wide;
    if (mb_lds_bytes_xld(a, k, compact, g)) return compact;
    return 0;
}

bool mbconv_block_supported(int H, int W, int Cin, int mid, int Cout, int k, int stride, int rd, int act_e, int act_d) {
    MbGeom g;
    if (!mb_geom(H, W, &g)) return false;
    if (act_e != ACT_SILU) return false;
    if (Cin % 8 || mid % 8 || mid > 2560 || Cout % 8 || rd < 1 || rd > MB_MAX_RD) return false;
    if (!((k == 3 || k == 5) && (stride == 1 || stride == 2))) return false;
    if (W == 7 && stride == 2) return false;                                 // (no 4x4 depthwise tile geometry)
    if ((long)mid * ((Cin + 31) & ~31) >= (1l << 30) || (long)Cout * mid >= (1l << 30)) return false;   // 32-bit offsets
    BlockArgs a{};
    a.H = H; a.W = W; a.Cin = Cin; a.Kp = (Cin + 31) & ~31; a.mid = mid; a.Cout = Cout;
    const int pad = k / 2;
    a.Ho = (H + 2 * pad - k) / stride + 1; a.Wo = (W + 2 * pad - k) / stride + 1;
    const int Pout = a.Ho * a.Wo, MTp = (Pout + 15) / 16, NTp = (Cout + 15) / 16;
    const int ntw = mb_pick_ntw(NTp, MTp, g.wi, g.mwp);
    const int nwn = (NTp + ntw - 1) / ntw;
    if (nwn > MB_WAVES) return false;
    int msplit = MB_WAVES / nwn;
    if (msplit > MTp) msplit = MTp;
    if ((MTp + msplit - 1) / msplit > g.mwp) return false;
    a.Kp2 = (mid + 31) & ~31;
    if (Pout * (g.wi == 7 ? 32 : 16) > g.a_it * MB_THREADS) return false;   // chunked projection: staged pieces per thread per K chunk
    if ((H * W + 15) / 16 > g.mw * g.mh) return false;
    if (!mb_instance_ok(k, stride, g.wi, a.Kp / 32, ntw, act_d)) return false;
    return mb_pick_xld(a, k, g) != 0;
}

template <int KS, int S, int WI, int MW, int MH, int NTW, int MWP, int KST, int A_IT, int ACT_E, int ACT_D>
static int launch_mb(BlockArgs a, int B, hipStream_t st) {
    MbGeom g;
    mb_geom(a.H, a.W, &g);
    a.XLD = mb_pick_xld(a, KS, g);
    a.KCS = mb_pick_kcs(a);
    const size_t lds = mb_lds_bytes_xld(a, KS, a.XLD, g);
    static bool attr_done[64] = {false};     // per device: the attribute is a property of the loaded code object
    int dev = 0;
    MI355_CHECK_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !attr_done[dev]) {
        MI355_CHECK_HIP(hipFuncSetAttribute((const void*)k_mbconv_block<KS, S, WI, MW, MH, NTW, MWP, KST, A_IT, ACT_E, ACT_D>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done[dev] = true;
    }
    hipLaunchKernelGGL((k_mbconv_block<KS, S, WI, MW, MH, NTW, MWP, KST, A_IT, ACT_E, ACT_D>), dim3(B), dim3(MB_THREADS), lds, st, a);
    MI355_LAUNCH_CHECK();
    return OK;
}

int launch_mbconv_block(const BlockArgs& a, int B, int k, int stride, hipStream_t st) {
    MI355_REQUIRE(mbconv_block_supported(a.H, a.W, a.Cin, a.mid, a.Cout, k, stride, a.rd, a.act_e, a.act_d), "mbconv_block: unsupported shape");
    const int Pout = a.Ho * a.Wo;
    MbGeom gg;
    mb_geom(a.H, a.W, &gg);
    const int ntw = mb_pick_ntw((a.Cout + 15) / 16, (Pout + 15) / 16, gg.wi, gg.mwp);   // column tiles per wave of the projection (the kernel's NTW is exact)
    const int kst = a.Kp / 32;
#define X(KS, S, WI, KST, NTW, AD)                                                                  \
    if (k == KS && stride == S && a.W == WI && kst == KST && ntw == NTW && a.act_d == AD)           \
        return launch_mb<KS, S, WI, (WI == 7 ? 4 : 7), (WI == 7 ? 1 : 2), NTW, (WI == 7 ? 4 : 7), KST, (WI == 7 ? 4 : 7), ACT_SILU, AD>(a, B, st);
    MB_INSTANCES(X)
#undef X
    set_error("mbconv_block: no instance for k=%d s=%d W=%d Kp=%d ntw=%d", k, stride, a.W, a.Kp, ntw);
    return ERR_UNSUPPORTED;
}

}  // namespace mi355
