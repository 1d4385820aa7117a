// One whole MBConv block per launch for the late stages (14x14 and 7x7 maps): per image ONE 16-wave workgroup runs
//   1x1 expand (MFMA) + bias + act  ->  depthwise kxk + bias + act  ->  SE squeeze / FC / FC / sigmoid
//   ->  gated 1x1 projection (MFMA) + bias (+ residual).
// gfx950 only.  Replaces the timm InvertedResidual.forward reached from inference/inference.py:199-201 for the
// eighteen 14x14 / 7x7 blocks of efficientnet_b3a.
//
// Why one kernel: at these sizes every separate kernel starts from a cold L2 and is latency-bound (the gated
// projections ran at ~1 TB/s, the SE kernel at 10-30 us of pure latency).  Here the expanded tensor E lives only in
// LDS (a 128/256-channel slab at a time), the block input X stays in LDS for the whole block (expand operand and
// residual), the depthwise output D makes one round trip through the L2 it was just written to (it cannot stay in LDS:
// the SE gate needs every channel's spatial mean before the projection can start), and the SE squeeze is complete inside
// the workgroup (fixed summation order, no partials in HBM).
//
// What bounds it (tools/microbench_valu.hip): the VALU.  v_exp_f32 / v_rcp_f32 cost 3.5 FMA issue slots each, so one SiLU
// is ~10 slots and the two SiLUs per expanded element cost as much as the 9-25 depthwise FMAs; v_dot2_f32_bf16,
// v_pk_fma_f32 and v_perm_b32 are half rate (no gain over plain FMAs); VALU issue runs at 5.7 / 3.4 / 2.9 cycles per
// instruction with 1 / 2 / 4 waves per SIMD.  The first version ran 16 waves (4 per SIMD, 128 VGPRs): it spilled, and every
// scratch reload waits (vmcnt is in order) for all prefetched weights - slower than the separate kernels.  This version:
// 8 waves with 256 VGPRs, every global load issued a phase (or several k-steps) ahead into registers, LDS-only barriers
// inside the loops, bias folded into the MFMA accumulator init.
#include "ops.h"

namespace mi355 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int MB_THREADS = 512;    // 8 waves: 256 VGPRs per lane, enough to keep every global load several steps ahead
constexpr int MB_WAVES = MB_THREADS / 64;
constexpr int MB_MAX_RD = 128;

__device__ __forceinline__ float mb_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float mb_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup release/acquire fence: hipcc drains
// vmcnt(0) in front of it, i.e. every wave would wait for its depthwise-output stores and for every prefetched weight
// fragment at each of the ~20 barriers of a block.  Nothing but LDS is exchanged inside the slab loop.
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
//   NWM  waves along the pixel dimension in the expand GEMM (the other 8/NWM waves split the slab's channel tiles)
//   CW   16-channel tiles per wave           -> slab width MC = (8/NWM) * CW * 16
//   MW   16-pixel tiles per wave (interleaved by NWM)
//   NTW  max 16-column output tiles per wave in the projection, MWP max 16-row tiles per wave there
//   WRING  k-steps of expand weights a wave holds in registers = the largest Kp/32 the class supports: a whole slab's
//          fragments are requested one phase ahead and the k-loop issues no loads (hipcc cannot count waits for
//          loop-carried loads: every in-loop refill became a vmcnt(0), i.e. an exposed L2 round trip per k-step)
//   A_IT   16-byte pieces of a projection A chunk (128 deep, 256 for the 7x7 class) staged per thread
//   MH   passes over the pixel tiles in the expand GEMM (NWM == 1 only): pass h covers tiles h * MW .. h * MW + MW - 1 with the
//        same weight fragments, which halves the accumulator / fragment registers of the 14x14 class
// NWM == 1 makes the slab loop BARRIER-FREE: a wave then expands all pixels of its own CW channel tiles and the depthwise
// phase of the same wave consumes exactly those channels of the E image (LDS operations of one wave execute in order), so
// the eight waves drift apart and one wave's MFMA phase overlaps its SIMD partner's SiLU / LDS phase.  With NWM == 2 two
// waves shared a channel tile and every slab needed two workgroup barriers (12 - 20 % of the block at 2 waves per SIMD).
template <int KS, int S, int WI, int NWM, int CW, int MW, int NTW, int MWP, int WRING, int A_IT, int MH = 1>
__global__ __launch_bounds__(MB_THREADS) void k_mbconv_block(const BlockArgs a) {
    using TP = MbTaps<KS>;
    constexpr int NWN = MB_WAVES / NWM;
    constexpr bool INDEP = NWM == 1;                  // a wave consumes only the E channels it produced
    static_assert(MH == 1 || NWM == 1, "several passes over the pixel tiles need a wave that owns whole channel tiles");
    constexpr int MC = NWN * CW * 16;                 // expanded channels per slab
    constexpr int CTW = MC / 16 / MB_WAVES;           // 16-channel tiles per wave in phase 2 (1 or 2)
    constexpr int PAD = KS / 2;
    constexpr int ELD = MC + 8;                       // E row stride (elements): +8 keeps 16 pixels x 16 B on distinct banks
    constexpr int NP = TP::NP;
    constexpr int EW = WI + 2 * PAD;                  // E image: zero columns left/right AND zero rows above/below
    constexpr int WO = (WI + 2 * PAD - KS) / S + 1;   // output width
    constexpr int KC = WI == 7 ? 256 : 128;           // projection K chunk (7x7 class: 64 rows, so twice as deep fits)
    constexpr int ALD = KC + 8;                       // projection A-chunk row stride
    static_assert(CTW >= 1 && CTW * 16 * MB_WAVES == MC, "phase 2: a wave owns whole 16-channel tiles");
    extern __shared__ __attribute__((aligned(16))) unsigned char mb_smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: everything derived from it stays scalar
    const int fr = lane & 15, fq = lane >> 4, fk = fq * 8;
    const int b = blockIdx.x;
    const int P = a.H * a.W, MT = (P + 15) >> 4, XLD = a.XLD;
    const int EH = a.H + 2 * PAD;
    const int EP = (EH * EW + 8 + 7) & ~7;          // + zero slack: the unused second tap of the last pair reads one pixel past the end
    const int Pout = a.Ho * a.Wo;
    const int MTo = (Pout + 15) >> 4;
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
    bf16_t* Es = reinterpret_cast<bf16_t*>(R);                             // [EH*EW][ELD]

    // optional phase timing (diagnosis): wave-uniform, so the counters live in scalar registers
    // tick(i): add the cycles since the previous tick to bucket i (wave 0's view; buckets are listed at the end)
    const bool stamping = a.stamps != nullptr;
    long long t_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long t_last = 0;
    if (stamping) t_last = (long long)__builtin_readcyclecounter();
    auto tick = [&](int i) {
        if (stamping) { const long long t = (long long)__builtin_readcyclecounter(); t_acc[i] += t - t_last; t_last = t; }
    };

    const int nslabs = (a.mid + MC - 1) / MC;
    const int KST = a.Kp >> 5;
    // every workgroup walks the slabs from a different start so that 256 of them do not stream the same weight lines
    // through the same L2 channels in lock-step (the slabs are independent: order does not change any sum)
    const int rb0 = (a.norot & 1) ? 0 : b, rb1 = (a.norot & 2) ? 0 : b, rb2 = (a.norot & 4) ? 0 : b, rb3 = (a.norot & 8) ? 0 : b;
    auto slab_of = [&](int ci) { return (ci + rb0) % nslabs; };

    // expand GEMM coordinates of this wave
    const int cwi = wave % NWN;          // which group of CW channel tiles
    const int mq = wave / NWN;           // pixel-tile phase (tiles mq, mq + NWM, ...)
    // W fragments come straight from L2 in MFMA layout (every element is read once per workgroup), a whole slab (WRING
    // k-steps) per wave; the next slab's are requested before this slab's activation epilogue, so they travel while the
    // VALU works.  Rows past the padded weight matrix are clamped to its last row: their (finite) results land in E
    // columns >= mid, which phase 2 never reads, and the loop keeps no per-lane branches.
    // (sched_barrier: under register pressure the scheduler sinks prefetch loads down to their first use, which turns
    //  every prefetch into an exposed round trip; the barrier pins them where they are written)
    u32x4 wq[WRING][CW];
    auto w_prefetch_slab = [&](int cbase) {
#pragma unroll
        for (int h = 0; h < WRING; ++h) {
            const int ks = min(h, KST - 1);
#pragma unroll
            for (int j = 0; j < CW; ++j) {
                const int n = min(cbase + (cwi * CW + j) * 16 + fr, midp - 1);
                wq[h][j] = *reinterpret_cast<const u32x4*>(a.We + (n * a.Kp + fk + ks * 32));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // depthwise weights of this wave's channel tiles: lane (n = lane & 15, tap half = lane >> 5) needs w[tap][c] of ITS
    // channel for both taps of every pair - 2-byte loads requested at the top of phase 1, used in phase 2
    unsigned short wd_raw[CTW][NP];
    f32x4 bd_reg[CTW];
    auto wd_fetch = [&](int cbase) {
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
            const int ch = min(cbase + (wave * CTW + c) * 16 + fr, a.mid - 1);
#pragma unroll
            for (int tp = 0; tp < NP; ++tp) {
                constexpr int dummy = 0; (void)dummy;
                const int ta = TP::tap_a(tp), tb = TP::tap_b(tp) < 0 ? TP::tap_a(tp) : TP::tap_b(tp);
                const int t = (lane & 32) ? tb : ta;
                wd_raw[c][tp] = a.Wd[t * a.mid + ch];
            }
            bd_reg[c] = *reinterpret_cast<const f32x4*>(a.bd + min(cbase + (wave * CTW + c) * 16 + fq * 4, a.mid - 4));
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // ------------------------------------------------------------------ phase 0: X[b] -> LDS, zero the E image
    w_prefetch_slab(slab_of(0) * MC);
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
                for (int id = tid; id < EP * (ELD / 8); id += MB_THREADS)
                    *reinterpret_cast<u32x4*>(&Es[id * 8]) = (u32x4){0u, 0u, 0u, 0u};
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

    for (int ci = 0; ci < nslabs; ++ci) {
        const int cbase = slab_of(ci) * MC;
        // ---- this slab's depthwise weights: requested now, used in phase 2.  (vmcnt completes in order and counts
        // stores: requested here, the only older traffic is the previous slab's depthwise output.)
        wd_fetch(cbase);
        tick(1);

        // ---- phase 1: E slab = act(X W^T + b) -> Es.  D = W x X^T: a lane holds 4 consecutive channels of one pixel.
#pragma unroll
        for (int h = 0; h < MH; ++h) {
            f32x4 acc[CW][MW];
#pragma unroll
            for (int j = 0; j < CW; ++j) {
                const int n4 = min(cbase + (cwi * CW + j) * 16 + fq * 4, midp - 4);
                const f32x4 bb = *reinterpret_cast<const f32x4*>(a.be + n4);
#pragma unroll
                for (int i = 0; i < MW; ++i) acc[j][i] = bb;
            }
            // A fragments of a k-step are read as one batch (tiles past the image are clamped to the last one: their
            // results are never stored), one k-step ahead of the MFMAs: with two waves per SIMD nothing else hides the
            // ~130-cycle LDS latency, and read-wait-MFMA per tile cost 3x the MFMA time
            int arow[MW];
#pragma unroll
            for (int i = 0; i < MW; ++i) arow[i] = (min(mq + NWM * (h * MW + i), MT - 1) * 16 + fr) * XLD + fk;
            bf16x8 af[2][MW];
#pragma unroll
            for (int i = 0; i < MW; ++i) af[0][i] = *reinterpret_cast<const bf16x8*>(&Xs[arow[i]]);
#pragma unroll
            for (int ks = 0; ks < WRING; ++ks) {
                if (ks < KST) {                                                     // (wave-uniform)
                    if (ks + 1 < WRING && ks + 1 < KST) {
#pragma unroll
                        for (int i = 0; i < MW; ++i) af[(ks + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(&Xs[arow[i] + (ks + 1) * 32]);
                    }
#pragma unroll
                    for (int i = 0; i < MW; ++i)
#pragma unroll
                        for (int j = 0; j < CW; ++j)
                            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wq[ks][j]), af[ks & 1][i], acc[j][i], 0, 0, 0);
                }
            }
            if (h == MH - 1) {
                tick(2);
                if (ci + 1 < nslabs) w_prefetch_slab(slab_of(ci + 1) * MC);
                tick(3);
            }
            MI355_ACT_DISPATCH(a.act_e, {
_Pragma("unroll")
                for (int j = 0; j < CW; ++j)
_Pragma("unroll")
                    for (int i = 0; i < MW; ++i) {
                        acc[j][i].x = act_c<ACT>(acc[j][i].x); acc[j][i].y = act_c<ACT>(acc[j][i].y);
                        acc[j][i].z = act_c<ACT>(acc[j][i].z); acc[j][i].w = act_c<ACT>(acc[j][i].w);
                    }
            })
#pragma unroll
            for (int i = 0; i < MW; ++i) {
                const int mt = mq + NWM * (h * MW + i);
                const int p = mt * 16 + fr;                  // this lane's pixel in tile i
                if (mt < MT && p < P) {
                    const int y = p / WI;
                    const int eoff = (y + PAD) * EW + (p - y * WI) + PAD;
#pragma unroll
                    for (int j = 0; j < CW; ++j) {
                        const int nl = (cwi * CW + j) * 16 + fq * 4;
                        u32x2 o;
                        o.x = pack2bf(acc[j][i].x, acc[j][i].y);
                        o.y = pack2bf(acc[j][i].z, acc[j][i].w);
                        *reinterpret_cast<u32x2*>(&Es[eoff * ELD + nl]) = o;
                    }
                }
            }
        }
        tick(4);
        if (!INDEP || (a.variant & 2)) mb_lds_barrier();      // (bit 1 of the tuning option: barriers kept, for A/B timing)
        else asm volatile("" ::: "memory");
        tick(5);

        // ---- phase 2: depthwise on the (otherwise idle) matrix pipe.  The VALU is the bottleneck of this kernel (two
        // SiLUs per expanded element); 9-25 FMAs per output on top made phase 2 twice as long as everything else.  A
        // depthwise conv is a contraction with a DIAGONAL weight matrix per tap: one 16x16x32 MFMA takes K = 2 taps x 16
        // channels, A = diag(w[tap][c]) (this wave's 16 channels, 1 nonzero per lane), B = 16 pixels x (2 taps x 16
        // channels) read straight from the E image (one ds_read_b128 per lane, immediate offsets).  1/16 of the MFMA
        // is useful work, which still beats the VALU: 13 MFMAs (208 cycles) replace 1600 VALU cycles per 16x16 outputs.
        // The next slab's expand weights were requested before the activation epilogue; retire them NOW, before this phase
        // issues its output stores: vmcnt completes in order and counts stores, so waiting for those weights at the top of
        // the next phase 1 would wait for every depthwise store as well.  Behind the wait the registers are "read" by an
        // empty asm, which makes them asm results (no pending load) for hipcc.
        {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int h = 0; h < WRING; ++h)
#pragma unroll
                for (int j = 0; j < CW; ++j) asm volatile("" : "+v"(wq[h][j]));
        }
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
            const int ct = wave * CTW + c;                       // channel tile of the slab
            const int ch0 = cbase + ct * 16;                      // its first channel
            if (ch0 < a.mid) {                                    // (wave-uniform)
                // diagonal weight fragments: lane (n = lane & 15, kg = lane >> 4) holds k = kg*8 + j -> tap (kg >> 1),
                // channel (kg & 1)*8 + j of the tile: nonzero only where that channel is the lane's own n
                u32x4 dwf[NP];
                {
                    const bool mine = ((fq & 1) == (fr >> 3)) && (ch0 + fr < a.mid);
#pragma unroll
                    for (int tp = 0; tp < NP; ++tp) {
                        const bool has = mine && !((lane & 32) && TP::tap_b(tp) < 0);
                        const unsigned v = has ? (unsigned)wd_raw[c][tp] : 0u;
                        const unsigned word = (fr & 1) ? (v << 16) : v;
                        const int q = (fr & 7) >> 1;
                        dwf[tp] = (u32x4){q == 0 ? word : 0u, q == 1 ? word : 0u, q == 2 ? word : 0u, q == 3 ? word : 0u};
                    }
                }
                float psum[4] = {0.f, 0.f, 0.f, 0.f};
                for (int mt = 0; mt < MTo; ++mt) {
                    const int p = mt * 16 + fr;
                    const int pc = min(p, Pout - 1);
                    const int oy = pc / WO, ox = pc - oy * WO;
                    // E pixel of tap (0,0) for this output pixel (E has PAD zero rows / columns on every side)
                    const int e0 = (oy * S) * EW + ox * S;
                    const bf16_t* ebase = Es + e0 * ELD + ct * 16 + (fq & 1) * 8;
                    const bf16_t* ev = ebase + ((fq >> 1) ? EW * ELD : 0);      // vertical pairs: second tap one row down
                    const bf16_t* eh = ebase + ((fq >> 1) ? ELD : 0);           // horizontal pairs: one pixel right
                    auto e_read = [&](int tp) -> bf16x8 {
                        const int ta = TP::tap_a(tp);
                        const int offs = ((ta / KS) * EW + (ta % KS)) * ELD;    // compile-time immediate after unrolling
                        return *reinterpret_cast<const bf16x8*>((TP::vertical(tp) ? ev : eh) + offs);
                    };
                    // reads in groups of NB, one group ahead of the MFMAs that consume them (left alone, hipcc emits
                    // read - wait - MFMA per tap pair: a full LDS round trip per MFMA, 6x slower)
                    constexpr int NB = 4, NG = (NP + NB - 1) / NB;
                    bf16x8 ef[2][NB];
#pragma unroll
                    for (int i = 0; i < NB; ++i)
                        if (i < NP) ef[0][i] = e_read(i);
                    f32x4 acc = bd_reg[c];
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        if (g + 1 < NG) {
#pragma unroll
                            for (int i = 0; i < NB; ++i)
                                if ((g + 1) * NB + i < NP) ef[(g + 1) & 1][i] = e_read((g + 1) * NB + i);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < NB; ++i)
                            if (g * NB + i < NP)
                                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&dwf[g * NB + i]), ef[g & 1][i], acc, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    MI355_ACT_DISPATCH(a.act_d, {
                        acc.x = act_c<ACT>(acc.x); acc.y = act_c<ACT>(acc.y); acc.z = act_c<ACT>(acc.z); acc.w = act_c<ACT>(acc.w);
                    })
                    if (p < Pout) {
                        psum[0] += acc.x; psum[1] += acc.y; psum[2] += acc.z; psum[3] += acc.w;
                        if (ch0 + fq * 4 < a.mid) {
                            u32x2 ov;
                            ov.x = pack2bf(acc.x, acc.y);
                            ov.y = pack2bf(acc.z, acc.w);
                            *reinterpret_cast<u32x2*>(Db + (p * a.mid + ch0 + fq * 4)) = ov;
                        }
                    }
                }
                // squeeze: this wave saw every pixel of its 16 channels - fold the 16 pixel lanes (fixed order), done
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) psum[j] += __shfl_xor(psum[j], o, 64);
                }
                if (fr == 0 && ch0 + fq * 4 < a.mid)
                    *reinterpret_cast<f32x4*>(&pool[ch0 + fq * 4]) = (f32x4){psum[0], psum[1], psum[2], psum[3]};
            }
        }
        tick(6);
        if (!INDEP || (a.variant & 2)) mb_lds_barrier();      // Es is rewritten by the next slab
        else asm volatile("" ::: "memory");
        tick(7);
    }
    __syncthreads();      // full fence: the depthwise output (global) is re-read by other waves in the projection
    tick(8);

        const float* gate = reinterpret_cast<const float*>(mb_smem);                                   // [ceil256(mid)]
        bf16_t* As = reinterpret_cast<bf16_t*>(mb_smem + (size_t)((a.mid + 255) & ~255) * 4);          // [2][MTp*16][ALD]
        const int MTp = (Pout + 15) >> 4, NTp = (a.Cout + 15) >> 4;
        const int ntw = mb_proj_ntw(NTp, MTp, NTW);           // column tiles per wave (<= NTW)
        const int nwn = (NTp + ntw - 1) / ntw;                // waves along N
        int msplit = MB_WAVES / nwn;                          // row groups
        if (msplit > MTp) msplit = MTp;
        const int mper = (MTp + msplit - 1) / msplit;         // row tiles per wave (<= MWP)
        const int wn = (wave % nwn + rb3) % nwn, wmh = wave / nwn;   // column group rotated by image: spreads the Wp lines over time
        const bool wactive = wmh < msplit;
        const int mt0 = wmh * mper;
        const int KST2 = a.Kp2 >> 5;
        const int nchunks = (a.Kp2 + KC - 1) / KC;
        constexpr int CPR = KC / 8;                            // 16-byte pieces per A row per K chunk
        constexpr int KPC = KC / 32;                           // k-steps per chunk
        const int a_items = Pout * CPR;
        const int abuf = MTp * 16 * ALD;
        const int Coutp = (a.Cout + 15) & ~15;


        // unconditional, clamped loads (a branch around a load makes hipcc wait vmcnt(0) at its use): rows / columns past
        // the end re-read valid data that is then multiplied by a zero gate or never stored
        u32x4 sreg[A_IT];
        auto a_load = [&](int chunk) {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int id = min(tid + i * MB_THREADS, a_items - 1);
                const int row = id / CPR, c = id - row * CPR;
                const int k = min(chunk * KC + c * 8, a.mid - 8);
                sreg[i] = *reinterpret_cast<const u32x4*>(Db + (row * a.mid + k));
            }
        };
        auto a_store = [&](int chunk) {
            bf16_t* dst = As + (chunk & 1) * abuf;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int id = tid + i * MB_THREADS;
                const int row = id / CPR, c = id - row * CPR;
                const int k = chunk * KC + c * 8;                    // < ceil256(mid) = pool's padded size
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(&gate[k]);
                const f32x4 g1 = *reinterpret_cast<const f32x4*>(&gate[k + 4]);
                const u32x4 v = sreg[i];
                u32x4 o;
                o.x = pack2bf(mb_lo(v.x) * g0.x, mb_hi(v.x) * g0.y); o.y = pack2bf(mb_lo(v.y) * g0.z, mb_hi(v.y) * g0.w);
                o.z = pack2bf(mb_lo(v.z) * g1.x, mb_hi(v.z) * g1.y); o.w = pack2bf(mb_lo(v.w) * g1.z, mb_hi(v.w) * g1.w);
                if (id < a_items) *reinterpret_cast<u32x4*>(&dst[row * ALD + c * 8]) = o;
            }
        };
        // weight fragments of one chunk (KPC k-steps x NTW tiles), ONE set: a k-step pair is re-requested for the next
        // chunk as soon as this chunk's MFMAs have consumed it
        u32x4 pq[KPC][NTW];
        auto p_fetch = [&](int chunk, int kk) {
            const int ks = min(chunk * KPC + kk, KST2 - 1);          // clamped: the tail k-steps are skipped by the MFMA loop
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const int n = min((wn * ntw + (j < ntw ? j : 0)) * 16 + fr, Coutp - 1);
                pq[kk][j] = *reinterpret_cast<const u32x4*>(a.Wp + (n * a.Kp2 + ks * 32 + fk));
            }
        };

    // ------------------------------------------------------------------ SE gate (bf16 weights, fp32 math)
    // FC1: a wave per hidden unit, four units per pass; ALL of a pass's weight loads (<= 5 x 4 x 16 B per lane) are requested
    // before the first multiply, so a pass costs one L2 round trip
    {
        constexpr int C8MAX = 5;                              // ceil(mid / 8 / 64) <= 5  (mid <= 2560, checked on the host)
        // groups of four units are rotated by image (L2 spreading); a unit keeps its slot q = unit % 4 in every image, so
        // that the code that produces it - and with it the last bit of its sum - does not depend on the batch position
        const int ngroups = (a.rd + 3) >> 2;
        for (int gi = wave; gi < ngroups; gi += MB_WAVES) {
            const int j0 = ((gi + rb1) % ngroups) * 4;
            u32x4 w[C8MAX][4];
#pragma unroll
            for (int it = 0; it < C8MAX; ++it) {
                const int c8 = min(lane + it * 64, (a.mid >> 3) - 1);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    w[it][q] = *reinterpret_cast<const u32x4*>(a.W1 + (min(j0 + q, a.rd - 1) * a.mid + c8 * 8));
            }
            float b1v[4];                                      // requested with the weights: as a load behind the reduction it was
#pragma unroll                                             // one more exposed L2 round trip per pass
            for (int q = 0; q < 4; ++q) b1v[q] = a.b1[min(j0 + q, a.rd - 1)];
            __builtin_amdgcn_sched_barrier(0);
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
        }
    }
    mb_lds_barrier();
    tick(9);
    // FC2: thread = (8-channel chunk, slice of the hidden units); partials through LDS, summed in slice order
    {
        const int nch = a.mid >> 3;
        int JS = MB_THREADS / nch;
        if (JS > 8) JS = 8;
        if (JS < 1) JS = 1;
        float* part = reinterpret_cast<float*>(R);           // [JS][mid]
        // the gate's bias, requested before anything else of this phase: as a load inside the reduction loop below it waited (vmcnt
        // is in order) for the whole first A chunk of the projection that is requested in between
        float b2v[5];                                        // mid <= 2560 (host check): <= 5 channels per thread
#pragma unroll
        for (int i = 0; i < 5; ++i) b2v[i] = a.b2[min(tid + i * MB_THREADS, a.mid - 1)];
        for (int ch0 = tid % nch + (tid / nch >= JS ? nch : 0); ch0 < nch; ch0 += (JS == 1 ? MB_THREADS : nch)) {
            const int ch = (ch0 + rb2 * 7) % nch;              // channel chunks rotated by image (L2 spreading)
            const int js = JS == 1 ? 0 : tid / nch;
            float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            // batches of NBJ loads, all requested before the first multiply (a `#pragma unroll 12` over the runtime trip count put
            // every iteration of short loops into the one-at-a-time remainder loop: a dependent L2 round trip per hidden unit);
            // slots past rd re-read the last row and multiply by zero - the order of the real terms is unchanged
            constexpr int NBJ = 16;
            for (int jb = js; jb < a.rd; jb += JS * NBJ) {
                u32x4 w[NBJ];
                float r[NBJ];
#pragma unroll
                for (int t = 0; t < NBJ; ++t) {
                    const int j = jb + t * JS;
                    w[t] = *reinterpret_cast<const u32x4*>(a.W2 + (min(j, a.rd - 1) * a.mid + ch * 8));
                    r[t] = j < a.rd ? rvec[j] : 0.f;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NBJ; ++t) {
                    s[0] += mb_lo(w[t].x) * r[t]; s[1] += mb_hi(w[t].x) * r[t]; s[2] += mb_lo(w[t].y) * r[t]; s[3] += mb_hi(w[t].y) * r[t];
                    s[4] += mb_lo(w[t].z) * r[t]; s[5] += mb_hi(w[t].z) * r[t]; s[6] += mb_lo(w[t].w) * r[t]; s[7] += mb_hi(w[t].w) * r[t];
                }
            }
            *reinterpret_cast<f32x4*>(&part[js * a.mid + ch * 8]) = (f32x4){s[0], s[1], s[2], s[3]};
            *reinterpret_cast<f32x4*>(&part[js * a.mid + ch * 8 + 4]) = (f32x4){s[4], s[5], s[6], s[7]};
            if (JS > 1) break;
        }
        mb_lds_barrier();
        // chunk 0 of the projection operands is requested here: it travels while the gate is reduced and the A buffers zeroed
        a_load(0);
#pragma unroll
        for (int kk = 0; kk < KPC; ++kk) p_fetch(0, kk);
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
    tick(10);

    {
        f32x4 acc[NTW][MWP];
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int n4 = min((wn * ntw + (j < ntw ? j : 0)) * 16 + fq * 4, Coutp - 4);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(a.bp + n4);
#pragma unroll
            for (int i = 0; i < MWP; ++i) acc[j][i] = bb;
        }
        for (int id = tid; id < 2 * abuf / 8; id += MB_THREADS)
            *reinterpret_cast<u32x4*>(&As[id * 8]) = (u32x4){0u, 0u, 0u, 0u};
        mb_lds_barrier();          // As zeroed
        a_store(0);
        if (nchunks > 1) a_load(1);
        __builtin_amdgcn_sched_barrier(0);
        mb_lds_barrier();
        // One wait point per iteration (the a_store at the top, vmcnt(0)): everything it waits for was requested at least
        // an MFMA phase earlier.
        tick(11);
        for (int ch = 0; ch < nchunks; ++ch) {
            if (ch + 1 < nchunks) a_store(ch + 1);     // chunk ch+1: registers -> gated bf16 -> the other LDS buffer (vmcnt(0))
            // every loop-carried register is "read" HERE, right behind the wait (an empty asm that takes and returns it):
            // from now on it is an asm result for hipcc, no longer a pending load, so its later uses do not wait for
            // the requests issued below
#pragma unroll
            for (int kk = 0; kk < KPC; ++kk)
#pragma unroll
                for (int j = 0; j < NTW; ++j) asm volatile("" : "+v"(pq[kk][j]));
            __builtin_amdgcn_sched_barrier(0);
            if (ch + 2 < nchunks) a_load(ch + 2);      // chunk ch+2's rows travel during this chunk's MFMAs
            __builtin_amdgcn_sched_barrier(0);
            tick(12);
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
                            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&pq[kk][j]), paf[AHEAD ? (kk & 1) : 0][i], acc[j][i], 0, 0, 0);
                }
                if ((kk & 1) == 1 && ch + 1 < nchunks) {     // this k-step pair is consumed: request it for the next chunk
                    __builtin_amdgcn_sched_barrier(0);
                    p_fetch(ch + 1, kk - 1);
                    p_fetch(ch + 1, kk);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            tick(13);
            mb_lds_barrier();
            tick(14);
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
                        const int n = min((wn * ntw + j) * 16 + fq * 4, a.Cout - 4);
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
                            if (a.has_res) {
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
    tick(15);
    if (stamping && tid == 0) {
        // 0 X load | 1 dw-weight request | 2 expand MFMA loop | 3 next W request | 4 act epilogue + E write | 5 barrier |
        // 6 depthwise (MFMA) + squeeze | 7 barrier | 8 fence | 9 SE FC1 | 10 SE FC2 | 11 projection prologue |
        // 12 A gate+store / next loads | 13 projection MFMAs | 14 barrier | 15 output epilogue
        long long* o = a.stamps + (size_t)b * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] = t_acc[i];
    }
}

// ------------------------------------------------------------------------------------------ host side
struct MbGeom { int wi, nwm, cw, mw, ntw, mwp, mc, wring, a_it, mh; };
static bool mb_geom(int H, int W, MbGeom* g) {
    if (W == 7 && H * W <= 64) { *g = {7, 1, 2, 4, 3, 4, 256, 12, 4, 1}; return true; }
    // (both 14x14 geometries - one wave per channel tile in two pixel passes, or two waves per pair of channel tiles - have the
    //  same slab width, LDS footprint and limits)
    if (W == 14 && H * W <= 208) { *g = {14, 1, 1, 7, 3, 7, 128, 5, 7, 2}; return true; }
    return false;
}

// LDS bytes for a given X row stride (elements); 0 = does not fit
static size_t mb_lds_bytes_xld(const BlockArgs& a, int k, int xld, const MbGeom& g) {
    const int P = a.H * a.W, MT = (P + 15) / 16, pad = k / 2;
    const int EP = ((a.H + 2 * pad) * (a.W + 2 * pad) + 8 + 7) & ~7;
    const size_t xs = ((size_t)MT * 16 * xld * 2 + 64 + 15) & ~(size_t)15, pl = (size_t)((a.mid + 255) & ~255) * 4;
    const size_t fixed = xs + pl + MB_MAX_RD * 4;
    const size_t slab = (size_t)EP * (g.mc + 8) * 2;
    const int Pout = a.Ho * a.Wo, MTp = (Pout + 15) / 16;
    const int nch = a.mid / 8;
    int JS = MB_THREADS / nch;
    if (JS > 8) JS = 8;
    if (JS < 1) JS = 1;
    const size_t se = (size_t)JS * a.mid * 4;
    const int kc = g.wi == 7 ? 256 : 128;
    const size_t proj = pl + (size_t)2 * MTp * 16 * (kc + 8) * 2;            // gate, then the A double buffer
    size_t r = slab;
    if (se > r) r = se;
    const size_t total = fixed + r > proj ? fixed + r : proj;
    return total <= 160 * 1024 ? total : 0;
}

// X row stride: Kp + 8 (conflict-free A-fragment reads) when it fits, else the compact ceil8(Cin) + 8 (the k-steps past
// Cin then read the zero pad and the next row's first elements against zero weight columns; 2-way bank conflicts)
static int mb_pick_xld(const BlockArgs& a, int k, const MbGeom& g) {
    const int wide = a.Kp + 8, compact = ((a.Cin + 7) & ~7) + 8;
    if (mb_lds_bytes_xld(a, k, wide, g)) return wide;
    if (mb_lds_bytes_xld(a, k, compact, g)) return compact;
    return 0;
}

bool mbconv_block_supported(int H, int W, int Cin, int mid, int Cout, int k, int stride, int rd) {
    MbGeom g;
    if (!mb_geom(H, W, &g)) return false;
    if (Cin % 8 || mid % 8 || mid > 2560 || Cout % 8 || rd < 1 || rd > MB_MAX_RD) return false;
    if (!((k == 3 || k == 5) && (stride == 1 || stride == 2))) return false;
    if ((long)mid * ((Cin + 31) & ~31) >= (1l << 30) || (long)Cout * mid >= (1l << 30)) return false;   // 32-bit offsets
    BlockArgs a{};
    a.H = H; a.W = W; a.Cin = Cin; a.Kp = (Cin + 31) & ~31; a.mid = mid; a.Cout = Cout;
    const int pad = k / 2;
    a.Ho = (H + 2 * pad - k) / stride + 1; a.Wo = (W + 2 * pad - k) / stride + 1;
    const int Pout = a.Ho * a.Wo, MTp = (Pout + 15) / 16, NTp = (Cout + 15) / 16;
    const int ntw = mb_proj_ntw(NTp, MTp, g.ntw);
    const int nwn = (NTp + ntw - 1) / ntw;
    if (nwn > MB_WAVES) return false;
    int msplit = MB_WAVES / nwn;
    if (msplit > MTp) msplit = MTp;
    if ((MTp + msplit - 1) / msplit > g.mwp) return false;
    if (Pout * (g.wi == 7 ? 32 : 16) > g.a_it * MB_THREADS) return false;   // staged pieces per thread per K chunk
    if ((H * W + 15) / 16 > g.nwm * g.mw * g.mh) return false;
    if (a.Kp / 32 > g.wring) return false;
    // register budget (hipcc spills, and a spill reload waits for every prefetch in flight): the 7x7 class with a 5x5
    // depthwise holds at most 8 k-steps of weights
    if (W == 7 && k == 5 && Cin > 256) return false;
    return mb_pick_xld(a, k, g) != 0;
}

template <int KS, int S, int WI, int NWM, int CW, int MW, int NTW, int MWP, int WRING, int A_IT, int MH = 1>
static int launch_mb(BlockArgs a, int B, hipStream_t st) {
    MbGeom g;
    mb_geom(a.H, a.W, &g);
    a.XLD = mb_pick_xld(a, KS, g);
    const size_t lds = mb_lds_bytes_xld(a, KS, a.XLD, g);
    static bool attr_done[64] = {false};     // per device: the attribute is a property of the loaded code object
    int dev = 0;
    MI355_CHECK_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !attr_done[dev]) {
        MI355_CHECK_HIP(hipFuncSetAttribute((const void*)k_mbconv_block<KS, S, WI, NWM, CW, MW, NTW, MWP, WRING, A_IT, MH>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done[dev] = true;
    }
    hipLaunchKernelGGL((k_mbconv_block<KS, S, WI, NWM, CW, MW, NTW, MWP, WRING, A_IT, MH>), dim3(B), dim3(MB_THREADS), lds, st, a);
    MI355_LAUNCH_CHECK();
    return OK;
}

template <int KS, int S>
static int launch_mb_ks(const BlockArgs& a, int B, hipStream_t st) {
    if (a.W == 7) {
        if (a.Kp <= 256) return launch_mb<KS, S, 7, 1, 2, 4, 3, 4, 8, 4>(a, B, st);
        return launch_mb<KS, S, 7, 1, 2, 4, 3, 4, 12, 4>(a, B, st);
    }
    if (a.variant & 1) {
        if (a.Kp <= 96) return launch_mb<KS, S, 14, 2, 2, 7, 3, 7, 3, 7>(a, B, st);
        return launch_mb<KS, S, 14, 2, 2, 7, 3, 7, 5, 7>(a, B, st);
    }
    if (a.Kp <= 96) return launch_mb<KS, S, 14, 1, 1, 7, 3, 7, 3, 7, 2>(a, B, st);
    return launch_mb<KS, S, 14, 1, 1, 7, 3, 7, 5, 7, 2>(a, B, st);
}

int launch_mbconv_block(const BlockArgs& a, int B, int k, int stride, hipStream_t st) {
    MI355_REQUIRE(mbconv_block_supported(a.H, a.W, a.Cin, a.mid, a.Cout, k, stride, a.rd), "mbconv_block: unsupported shape");
    if (k == 3 && stride == 1) return launch_mb_ks<3, 1>(a, B, st);
    if (k == 3 && stride == 2) return launch_mb_ks<3, 2>(a, B, st);
    if (k == 5 && stride == 1) return launch_mb_ks<5, 1>(a, B, st);
    return launch_mb_ks<5, 2>(a, B, st);
}

}  // namespace mi355
