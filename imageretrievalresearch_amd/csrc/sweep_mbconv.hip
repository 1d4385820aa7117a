// Front half of an MBConv block for the early stages (112x112 .. 28x28 maps): 1x1 expand (MFMA) + bias + act ->
// depthwise kxk (MFMA, diagonal weights) + bias + act -> D, plus the complete SE squeeze sums.  gfx950 only.
// Replaces the expand / depthwise part of timm's InvertedResidual.forward reached from inference/inference.py:199-201
//  for the blocks whose maps are too large for the whole-block kernel (mbconv_block.hip).
//
// Decomposition ("row sweep"): a workgroup owns ONE image and a few 16-channel slabs of the expanded tensor.  For each
// slab it walks down the image in bands of TH output rows:
//   phase 1  the band's NEW input rows: E[row][x][16] = act(X W^T + b) with X fragments loaded straight from global
//            (L2) into MFMA layout, written to an LDS row buffer of IH = (TH-1)*S + KS rows (zero columns left/right)
//   phase 2  depthwise on the matrix pipe: per 16 output pixels x 16 channels, one 16x16x32 MFMA per PAIR of taps with
//            A = diag(w[tap][c]) and B = 16 pixels x (2 taps x 16 channels) read from the row buffer (ds_read_b128 at
//            compile-time offsets) - the VALU only sees the two activations per element
//   halo     the last KS-S rows of the buffer are the next band's first rows: one LDS->LDS copy instead of recomputing
//            them (the band kernel of fused_mbconv.hip recomputes the halo: 25-40 % more expand work for short bands)
// Because a workgroup sees every pixel of its channels, the squeeze sums are complete (no partials in HBM), the slab's
// expand and depthwise weights are loaded once into registers and stay there for the whole sweep, and the LDS need is one
// row buffer of <= ~55 KB: two or three workgroups share a CU and hide each other's barriers and L2 latency.
// Stride 2: the row buffer keeps even and odd columns in separate planes, so 16 consecutive output pixels read 16
// consecutive LDS pixels for every tap (a stride-2 walk over 48-byte pixels would be a 2-way bank conflict).
#include "ops.h"

namespace mi355 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void sw_lds_barrier() {          // orders LDS traffic only (see mbconv_block.hip)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

typedef __bf16 sw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float sw_f32x2 __attribute__((ext_vector_type(2)));
// four fp32 -> four bf16 (two dwords) with two v_cvt_pk_bf16_f32; written with vector converts because hipcc otherwise pairs
// the activation's last multiply as (x,z),(y,w) and then spends six ALU ops re-ordering the halves
__device__ __forceinline__ u32x2 sw_pack4(const f32x4 v) {
    const sw_bf16x2 lo = __builtin_convertvector((sw_f32x2){v.x, v.y}, sw_bf16x2);
    const sw_bf16x2 hi = __builtin_convertvector((sw_f32x2){v.z, v.w}, sw_bf16x2);
    u32x2 o;
    o.x = *reinterpret_cast<const unsigned*>(&lo);
    o.y = *reinterpret_cast<const unsigned*>(&hi);
    return o;
}

// activation: compile-time for the common case (both SiLU: EfficientNet), runtime switch otherwise (AE / AD = -1)
template <int ACT>
__device__ __forceinline__ float sw_act(float x, int act_runtime) {
    if constexpr (ACT >= 0) return act_c<ACT>(x);
    else return apply_act(x, act_runtime);
}

// tap pairs as in mbconv_block.hip: vertical pairs (ky, kx)+(ky+1, kx) for even ky, horizontal pairs along the last row
template <int KS> struct SwTaps {
    static constexpr int NV = (KS / 2) * KS;
    static constexpr int NH = (KS + 1) / 2;
    static constexpr int NP = NV + NH;
    __host__ __device__ static constexpr bool vertical(int tp) { return tp < NV; }
    __host__ __device__ static constexpr int tap_a(int tp) {
        return tp < NV ? (tp / KS) * 2 * KS + tp % KS : (KS - 1) * KS + (tp - NV) * 2;
    }
    __host__ __device__ static constexpr int tap_b(int tp) {
        return tp < NV ? tap_a(tp) + KS : ((tp - NV) * 2 + 1 < KS ? tap_a(tp) + 1 : -1);
    }
};

template <int KS, int S, int WI, int TH>
struct SwGeom {
    static constexpr int PAD = KS / 2;
    static constexpr int HALO = KS - S;                       // rows shared by consecutive bands
    static constexpr int NEWR = TH * S;                       // new input rows per band
    static constexpr int IH = NEWR + HALO;                    // rows of an E buffer
    static constexpr int EW = WI + 2 * PAD;
    static constexpr int EWH = (EW + 1) / 2;                  // stride 2: columns per parity plane
    static constexpr int RP = S == 1 ? EW : 2 * EWH;          // pixels per E row
    static constexpr int HOFF = S == 1 ? 1 : EWH;             // pixel offset of "one column to the right" for even kx
    static constexpr int WO = (WI + 2 * PAD - KS) / S + 1;
    static constexpr int ELD = 24;                            // 16 channels + 8: 16 pixels x 16 B land on distinct banks
    static constexpr int EBUF = (IH + 1) * RP * ELD;          // elements per buffer (+ one zero slack row: unused second taps)
    static constexpr size_t lds_bytes(int nw, int ns = 1) { return (size_t)ns * 2 * EBUF * 2 + (size_t)nw * 16 * ns * 4; }
    __host__ __device__ static constexpr int epx(int r, int c) {   // E pixel index of (buffer row, padded column)
        return r * RP + (S == 1 ? c : (c & 1) * EWH + (c >> 1));
    }
};

// One 16-channel slab at a time; NW waves split the 16-pixel tiles of both phases; KST = Kp / 32 (1 ... 4); OCC = workgroups
// the host expects per CU (register budget); PREF: request the next band's X fragments before the depthwise phase.
//
// Per band ONE barrier: the E rows live in two buffers used alternately, so a wave that finishes the depthwise phase of
// band j goes straight on to the expand phase of band j+1 (other buffer) while slower waves still read band j - VALU-heavy
// and MFMA/LDS-heavy work of the same workgroup overlap.  The halo rows are copied buffer -> buffer at the start of the
// depthwise phase.  "Band -1" is expand-only: it produces band 0's halo rows (zeros above the image), and rows below the
// image are written as zeros by the same rule, so there is no separate zero-fill.
// NS = 16-channel tiles per pass over the image (1, or 2: RexNet's layers with two or more k-steps of input channels were bound by
// re-reading X from L2 once per tile - 22 passes over a 400 KB image at 6.5 TB/s; a pair of tiles per pass halves that, with the
// E row windows, the slab constants and the squeeze sums held twice).
template <int KS, int S, int WI, int TH, int NW, int KST, int OCC, bool PREF, int AE, int AD, int NS = 1>
__global__ __launch_bounds__(NW * 64, (NW * OCC + 3) / 4) void k_sweep_mbconv(const SweepArgs a) {
    using G = SwGeom<KS, S, WI, TH>;
    using TP = SwTaps<KS>;
    constexpr int NTHR = NW * 64;
    constexpr int PAD = G::PAD, HALO = G::HALO, NEWR = G::NEWR, RP = G::RP, ELD = G::ELD, WO = G::WO, NP = TP::NP;
    constexpr int NPX1 = NEWR * WI, NT1 = (NPX1 + 15) / 16, MW1 = (NT1 + NW - 1) / NW;   // expand: pixels / tiles / tiles per wave
    constexpr int NPX2 = TH * WO, NT2 = (NPX2 + 15) / 16, MW2 = (NT2 + NW - 1) / NW;     // depthwise
    constexpr int EBUF = G::EBUF;
    extern __shared__ __attribute__((aligned(16))) unsigned char sw_smem[];
    bf16_t* Es = reinterpret_cast<bf16_t*>(sw_smem);                                  // [NS][2][EBUF]
    float* red = reinterpret_cast<float*>(sw_smem + (size_t)NS * 2 * EBUF * 2);       // [NW][16 NS]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4, fk = fq * 8;
    // XCD-aware placement (workgroups are dealt round-robin over the 8 XCDs): the workgroups of one image are consecutive on
    // ONE XCD, so the image's X rows are fetched into one L2 once and re-read from there by the other slabs
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int b = xcd + 8 * (seq / a.csplit), g = seq % a.csplit;
    if (b >= a.B) return;
    const int midp = (a.mid + 15) & ~15;
    const int nslab = (midp / 16 + NS - 1) / NS;          // passes: NS channel tiles each
    const int nbands = (a.Ho + TH - 1) / TH;

    // optional phase timing (diagnosis): wave-uniform, the counters live in scalar registers
    const bool stamping = a.stamps != nullptr;
    long long t_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long t_last = 0;
    if (stamping) t_last = (long long)__builtin_readcyclecounter();
    auto tick = [&](int i) {
        if (stamping) { const long long t = (long long)__builtin_readcyclecounter(); t_acc[i] += t - t_last; t_last = t; }
    };

    // zero both buffers once: pad columns and the slack rows stay zero, the phases rewrite interior pixels only
    for (int id = tid; id < NS * 2 * EBUF / 8; id += NTHR) *reinterpret_cast<u32x4*>(&Es[id * 8]) = (u32x4){0u, 0u, 0u, 0u};

    // per-lane constants of the two tile loops (the same for every band and slab: no index arithmetic in the loops)
    int eoff[MW1];            // expand: E offset of this lane's pixel of tile i (row HALO + p / WI, 4 channels from fq * 4)
#pragma unroll
    for (int i = 0; i < MW1; ++i) {
        const int p = min((wave + NW * i) * 16 + fr, NPX1 - 1);
        const int r = p / WI, x = p - r * WI;
        eoff[i] = G::epx(HALO + r, x + PAD) * ELD + fq * 4;
    }
    int rbase[MW2];           // depthwise: E offset of tap (0,0) of this lane's output pixel of tile i (+ its 8-channel half)
#pragma unroll
    for (int i = 0; i < MW2; ++i) {
        const int p = min((wave + NW * i) * 16 + fr, NPX2 - 1);
        const int oyl = p / WO, ox = p - oyl * WO;
        rbase[i] = (oyl * S * RP + ox) * ELD + (fq & 1) * 8;
    }
    const int vsel = (fq >> 1) ? RP * ELD : 0;             // vertical pairs: second tap one row down
    const int hsel = (fq >> 1) ? G::HOFF * ELD : 0;        // horizontal pairs: one column right
    sw_lds_barrier();
    tick(0);

    const bf16_t* xb = a.X + (size_t)b * a.H * WI * a.Cin;
    bf16_t* Db = a.D + (size_t)b * a.Ho * WO * a.mid;
    const int hw1 = a.H * WI - 1;
    const int kclamp = a.Cin - 8;

    // X fragments of one band straight from global (L2) in MFMA layout; the band's new rows are contiguous in memory, so the
    // pixel index is just first-row * WI + p.  Rows outside the image are clamped (their results are replaced by zeros);
    // k >= Cin re-reads valid data that meets zero weight columns.
    u32x4 xa[MW1][KST];
    auto load_x = [&](int band) {
        const int pix0 = (band * NEWR - PAD + HALO) * WI;
#pragma unroll
        for (int i = 0; i < MW1; ++i) {
            const int pix = min(max(pix0 + (wave + NW * i) * 16 + fr, 0), hw1);
#pragma unroll
            for (int ks = 0; ks < KST; ++ks) {
                u32x4 t = {0u, 0u, 0u, 0u};
                if (!(a.debug_skip & 4)) t = *reinterpret_cast<const u32x4*>(xb + pix * a.Cin + min(ks * 32 + fk, kclamp));
                xa[i][ks] = t;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // slabs are walked from an image-dependent start so co-resident workgroups do not stream the same weight lines in
    // lock-step (slabs are independent: no sum changes order)
    for (int si = g; si < nslab; si += a.csplit) {
        const int ch0 = ((si + b) % nslab) * (16 * NS);       // first channel of this pass; tile s covers ch0 + 16 s ...
        load_x(-1);
        // ---- slab constants: expand weights (MFMA A fragments) + bias, depthwise taps of this lane's channel + bias
        u32x4 wf[NS][KST];
        f32x4 bb[NS], bdr[NS];
        unsigned wd_raw[NS][(NP + 1) / 2];                    // two tap pairs per register
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int c0 = ch0 + 16 * s;
            const int n = min(c0 + fr, midp - 1);
#pragma unroll
            for (int ks = 0; ks < KST; ++ks) wf[s][ks] = *reinterpret_cast<const u32x4*>(a.We + (n * a.Kp + ks * 32 + fk));
            bb[s] = *reinterpret_cast<const f32x4*>(a.be + min(c0 + fq * 4, midp - 4));
            const int ch = min(c0 + fr, a.mid - 1);
#pragma unroll
            for (int tp = 0; tp < NP; ++tp) {
                const int ta = TP::tap_a(tp), tb = TP::tap_b(tp) < 0 ? TP::tap_a(tp) : TP::tap_b(tp);
                const unsigned v = a.Wd[((lane & 32) ? tb : ta) * a.mid + ch];
                if (tp & 1) wd_raw[s][tp >> 1] |= v << 16;
                else wd_raw[s][tp >> 1] = v;
            }
            bdr[s] = *reinterpret_cast<const f32x4*>(a.bd + min(c0 + fq * 4, a.mid - 4));
        }
        // diagonal weight fragments of the depthwise MFMAs: lane (n = lane & 15, kg = lane >> 4) holds k = kg*8 + j -> tap
        // (kg >> 1), channel (kg & 1)*8 + j of the tile: nonzero only where that channel is the lane's own n
        auto build_dwf = [&](int s, u32x4* dwf) {
            const bool mine = ((fq & 1) == (fr >> 3)) && (ch0 + 16 * s + fr < a.mid);
            const int q = (fr & 7) >> 1;
#pragma unroll
            for (int tp = 0; tp < NP; ++tp) {
                const bool has = mine && !((lane & 32) && TP::tap_b(tp) < 0);
                const unsigned v = has ? ((tp & 1) ? wd_raw[s][tp >> 1] >> 16 : wd_raw[s][tp >> 1] & 0xffffu) : 0u;
                const unsigned word = (fr & 1) ? (v << 16) : v;
                dwf[tp] = (u32x4){q == 0 ? word : 0u, q == 1 ? word : 0u, q == 2 ? word : 0u, q == 3 ? word : 0u};
            }
        };
        constexpr bool DWF_PER_SLAB = KS == 3;               // 20 registers for 3x3; the 52 of 5x5 are rebuilt per band
        u32x4 dwf_s[NS][DWF_PER_SLAB ? NP : 1];
        float psum[NS][4];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) psum[s][j] = 0.f;
        // retire the slab constants (and band -1's X fragments) HERE and pass them through an empty asm: hipcc then treats them
        // as plain register values.  Left alone it put a vmcnt(0) in front of the first depthwise MFMA of every band (first use
        // of the depthwise bias inside the loop), which also waited for the X fragments prefetched for the next band.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int ks = 0; ks < KST; ++ks) asm volatile("" : "+v"(wf[s][ks]));
            asm volatile("" : "+v"(bb[s]));
            asm volatile("" : "+v"(bdr[s]));
#pragma unroll
            for (int i = 0; i < (NP + 1) / 2; ++i) asm volatile("" : "+v"(wd_raw[s][i]));
        }
#pragma unroll
        for (int i = 0; i < MW1; ++i)
#pragma unroll
            for (int ks = 0; ks < KST; ++ks) asm volatile("" : "+v"(xa[i][ks]));
        if constexpr (DWF_PER_SLAB) {
#pragma unroll
            for (int s = 0; s < NS; ++s) build_dwf(s, dwf_s[s]);
        }
        tick(1);

        for (int band = -1; band < nbands; ++band) {
            bf16_t* Ec = Es + ((band + 1) & 1) * EBUF;         // this band's buffer of tile 0 (tile s: + s * 2 * EBUF)
            // ---- expand: rows [HALO, IH) of Ec = act(X W^T + b) for the band's new input rows, zeros outside the image
            {
                const int iyn0 = band * NEWR - PAD + HALO;                       // first new input row
                const int plo = max(0, -iyn0) * WI, phi = max(0, min(NEWR, a.H - iyn0)) * WI;   // valid pixels: [plo, phi)
                if (!PREF && band >= 0) load_x(band);
                tick(3);
                if (stamping) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tick(4); }
#pragma unroll
                for (int i = 0; i < MW1; ++i) {
                    const int t0 = (wave + NW * i) * 16;
                    // (t0 < NPX1 is compile-time for all but the last i; band -1 only matters for the rows that become band 0's halo)
                    if (t0 < NPX1 && (band >= 0 || t0 + 16 > (NEWR - HALO) * WI)) {
#pragma unroll
                        for (int s = 0; s < NS; ++s) {
                            u32x2 o = {0u, 0u};
                            if (t0 < phi && t0 + 16 > plo && !(a.debug_skip & 1)) {  // (wave-uniform) some pixel inside the image
                                f32x4 acc = bb[s];
#pragma unroll
                                for (int ks = 0; ks < KST; ++ks)
                                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wf[s][ks]),
                                                                                  *reinterpret_cast<bf16x8*>(&xa[i][ks]), acc, 0, 0, 0);
                                acc.x = sw_act<AE>(acc.x, a.act_e); acc.y = sw_act<AE>(acc.y, a.act_e);
                                acc.z = sw_act<AE>(acc.z, a.act_e); acc.w = sw_act<AE>(acc.w, a.act_e);
                                o = sw_pack4(acc);
                                if (t0 < plo || t0 + 16 > phi) {                     // (wave-uniform) tile straddles the image edge
                                    const bool in = t0 + fr >= plo && t0 + fr < phi;
                                    o.x = in ? o.x : 0u; o.y = in ? o.y : 0u;
                                }
                            }
                            if (t0 + fr < NPX1) *reinterpret_cast<u32x2*>(&Ec[s * 2 * EBUF + eoff[i]]) = o;
                        }
                    }
                }
                if (PREF && band + 1 < nbands) load_x(band + 1);
            }
            tick(5);
            sw_lds_barrier();
            tick(6);
            // ---- halo: this band's last HALO rows are the next band's first rows (other buffer; nobody reads it now)
            if (band + 1 < nbands) {
                bf16_t* En = Es + (band & 1) * EBUF;
                constexpr int NV16 = HALO * RP * ELD / 8;
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    for (int id = tid; id < NV16; id += NTHR)
                        *reinterpret_cast<u32x4*>(&En[s * 2 * EBUF + id * 8]) = *reinterpret_cast<const u32x4*>(&Ec[s * 2 * EBUF + NEWR * RP * ELD + id * 8]);
            }
            tick(2);
            // ---- depthwise on the matrix pipe
#pragma unroll
            for (int s = 0; s < NS; ++s) {
            const int c0 = ch0 + 16 * s;
            if (band >= 0 && c0 < a.mid && !(a.debug_skip & 2)) {      // (wave-uniform)
                u32x4 dwf_b[DWF_PER_SLAB ? 1 : NP];
                if constexpr (!DWF_PER_SLAB) build_dwf(s, dwf_b);
                const u32x4* dwf = DWF_PER_SLAB ? dwf_s[s] : dwf_b;
                const int np2 = min(TH, a.Ho - band * TH) * WO;
                bf16_t* Dband = Db + band * NPX2 * a.mid + c0 + fq * 4;
                const bf16_t* Ecs = Ec + s * 2 * EBUF;
#pragma unroll
                for (int i = 0; i < MW2; ++i) {
                    const int t0 = (wave + NW * i) * 16;
                    if (t0 < np2) {                                     // (wave-uniform)
                        const bf16_t* ev = Ecs + rbase[i] + vsel;
                        const bf16_t* eh = Ecs + rbase[i] + hsel;
                        auto e_read = [&](int tp) -> bf16x8 {
                            const int ta = TP::tap_a(tp);
                            const int offs = G::epx(ta / KS, ta % KS) * ELD;            // compile-time immediate after unrolling
                            return *reinterpret_cast<const bf16x8*>((TP::vertical(tp) ? ev : eh) + offs);
                        };
                        constexpr int NB = KS == 5 ? 2 : NP, NG = (NP + NB - 1) / NB;
                        bf16x8 ef[2][NB];
#pragma unroll
                        for (int k = 0; k < NB; ++k)
                            if (k < NP) ef[0][k] = e_read(k);
                        f32x4 acc = bdr[s];
#pragma unroll
                        for (int gq = 0; gq < NG; ++gq) {
                            if (gq + 1 < NG) {
#pragma unroll
                                for (int k = 0; k < NB; ++k)
                                    if ((gq + 1) * NB + k < NP) ef[(gq + 1) & 1][k] = e_read((gq + 1) * NB + k);
                            }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int k = 0; k < NB; ++k)
                                if (gq * NB + k < NP)
                                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&dwf[gq * NB + k]), ef[gq & 1][k], acc, 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        acc.x = sw_act<AD>(acc.x, a.act_d); acc.y = sw_act<AD>(acc.y, a.act_d);
                        acc.z = sw_act<AD>(acc.z, a.act_d); acc.w = sw_act<AD>(acc.w, a.act_d);
                        const int p = t0 + fr;
                        if (p < np2) {
                            psum[s][0] += acc.x; psum[s][1] += acc.y; psum[s][2] += acc.z; psum[s][3] += acc.w;
                            if (c0 + fq * 4 < a.mid) {
                                *reinterpret_cast<u32x2*>(Dband + p * a.mid) = sw_pack4(acc);
                            }
                        }
                    }
                }
            }
            }
            tick(7);
        }

        // ---- squeeze: fold the 16 pixel lanes, then the waves, both in a fixed order
        if (a.pool) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) psum[s][j] += __shfl_xor(psum[s][j], o, 64);
                }
                if (fr == 0) *reinterpret_cast<f32x4*>(&red[(wave * NS + s) * 16 + fq * 4]) = (f32x4){psum[s][0], psum[s][1], psum[s][2], psum[s][3]};
            }
            sw_lds_barrier();
            if (tid < 16 * NS) {
                const int s = tid >> 4, c = tid & 15;
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) sum += red[(w * NS + s) * 16 + c];
                if (ch0 + tid < a.mid) a.pool[(size_t)b * a.mid + ch0 + tid] = sum;
            }
        }
        sw_lds_barrier();       // the next slab's band -1 rewrites the buffers (and red)
        tick(9);
    }
    // buckets: 0 init | 1 slab constants (issue) | 2 halo copy | 3 X loads (issue) | 4 wait for the loads | 5 expand + act +
    //          E writes | 6 barrier | 7 depthwise + act + D stores | 9 squeeze + barrier
    if (stamping && tid == 0) {
#pragma unroll
        for (int i = 0; i < 10; ++i) atomicAdd(reinterpret_cast<unsigned long long*>(a.stamps + (size_t)b * 16 + i), (unsigned long long)t_acc[i]);
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------
struct SwPlan { int cls; int wgs_per_cu; int nct; };

// shape classes (template instances).  NCT = 1 everywhere: one 16-channel tile per slab keeps the row buffer <= 55 KB
// (2-3 workgroups per CU); 7 waves because 49 pixel tiles (14 or 28 rows of 56 / 28 pixels) split evenly over them.
enum { SW_NONE = 0, SW_3_2_112, SW_3_1_56, SW_5_2_56, SW_5_1_28, SW_3_2_28, SW_3_2_56, SW_3_1_28 };

static int sw_class(int H, int W, int k, int stride) {
    if (k == 3 && stride == 2 && W == 112 && H == 112) return SW_3_2_112;
    if (k == 3 && stride == 1 && W == 56 && H == 56) return SW_3_1_56;
    if (k == 5 && stride == 2 && W == 56 && H == 56) return SW_5_2_56;
    if (k == 5 && stride == 1 && W == 28 && H == 28) return SW_5_1_28;
    if (k == 3 && stride == 2 && W == 28 && H == 28) return SW_3_2_28;
    if (k == 3 && stride == 2 && W == 56 && H == 56) return SW_3_2_56;
    if (k == 3 && stride == 1 && W == 28 && H == 28) return SW_3_1_28;
    return SW_NONE;
}

// RexNet's squeeze-excite blocks at 56x56 / 28x28 (SiLU after the expand, linear depthwise; 58 ... 122 input channels = two to four
// k-steps): (class, k-steps) pairs that exist for exactly that activation pair
static bool sw_rex_instance(int cls, int kst) {
    // (measured, B = 256: 77->462 s2 @56 0.57 -> 0.42 ms, 58->348 0.40 -> 0.35, 75->450 @28 0.20 -> 0.16, 100->600 0.28 -> 0.26, 92->552 s2 0.18 -> 0.16;
    //  122->732 s2 @28 with four k-steps LOST, 0.23 -> 0.39 ms (11 spilled registers in the expand loop): not instantiated)
    return (cls == SW_3_2_56 && (kst == 2 || kst == 3)) || (cls == SW_3_1_28 && (kst == 3 || kst == 4)) || (cls == SW_3_2_28 && kst == 3);
}

bool sweep_mbconv_supported(int H, int W, int Cin, int mid, int k, int stride, int act_e, int act_d) {
    if (Cin % 8 || mid % 8 || Cin < 8) return false;
    const int cls = sw_class(H, W, k, stride), kst = (Cin + 31) / 32;
    if (cls == SW_NONE) return false;
    if (act_e == ACT_SILU && act_d == ACT_NONE && sw_rex_instance(cls, kst)) return true;
    return Cin <= 64 && cls <= SW_3_2_28;
}

template <int KS, int S, int WI, int TH, int NW, int KST, int OCC, bool PREF, int AE, int AD, int NS = 1>
static int launch_sw_act(SweepArgs a, int B, hipStream_t st) {
    using G = SwGeom<KS, S, WI, TH>;
    const size_t lds = G::lds_bytes(NW, NS);
    static_assert(G::lds_bytes(NW, NS) <= 160 * 1024, "sweep: the row windows do not fit the LDS");
    static bool attr_done[MI355_MAX_DEVICES] = {false};   // per device
    if (first_time_on_this_device(attr_done)) {
        MI355_CHECK_HIP(hipFuncSetAttribute((const void*)k_sweep_mbconv<KS, S, WI, TH, NW, KST, OCC, PREF, AE, AD, NS>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    // workgroups per image (each takes nslab / csplit slabs).  Measured at B = 256 (tools/tune_sweep.py): what matters is that the
    // slabs divide evenly (5 + 4 slabs per pair of workgroups cost 15 % against 3 + 3 + 3) and that a workgroup lives long
    // enough to amortise its start; the per-class preference below is the measured optimum, moved to the nearest divisor of
    // nslab, and raised for small batches until every CU slot has a workgroup.
    const int nslab = cdiv(cdiv(a.mid, 16), NS);          // passes over the image: NS channel tiles each
    const int slots = 256 * OCC;
    int c = WI >= 56 ? (WI == 112 ? 3 : 4) : 2;
    while (c < nslab && (nslab % c != 0 || (long)B * c < slots)) ++c;
    if (c > nslab) c = nslab;
    a.csplit = a.csplit_override > 0 ? std::min(a.csplit_override, nslab) : c;
    a.B = B;
    const unsigned grid = (unsigned)(8 * cdiv(B, 8) * a.csplit);
    hipLaunchKernelGGL((k_sweep_mbconv<KS, S, WI, TH, NW, KST, OCC, PREF, AE, AD, NS>), dim3(grid), dim3(NW * 64), lds, st, a);
    MI355_LAUNCH_CHECK();
    return OK;
}

template <int KS, int S, int WI, int TH, int NW, int KST, int OCC, bool PREF>
static int launch_sw(const SweepArgs& a, int B, hipStream_t st) {
    if (a.act_e == ACT_SILU && a.act_d == ACT_SILU) return launch_sw_act<KS, S, WI, TH, NW, KST, OCC, PREF, ACT_SILU, ACT_SILU>(a, B, st);
    // RexNet: SiLU after the expand, nothing after the depthwise (SE + ReLU6 follow in the projection's A load)
    if (a.act_e == ACT_SILU && a.act_d == ACT_NONE) return launch_sw_act<KS, S, WI, TH, NW, KST, OCC, PREF, ACT_SILU, ACT_NONE>(a, B, st);
    return launch_sw_act<KS, S, WI, TH, NW, KST, OCC, PREF, -1, -1>(a, B, st);
}

// variant table {TH, NW, OCC, PREF} per class; variant 0 is the default (both K depths, any activation), the others exist only
// for the EfficientNet-B3a instances and are kept for tuning runs (tools/tune_sweep.py)
// (two k-steps: no cross-phase X prefetch - its 2 x MW1 x 4 extra live registers spill at the 128-VGPR budget)
#define SW_V0(KS, S, WI, TH, NW, OCC, PREF) \
    return k2 ? launch_sw<KS, S, WI, TH, NW, 2, OCC, false>(a, B, st) : launch_sw<KS, S, WI, TH, NW, 1, OCC, PREF>(a, B, st)
#define SW_VT(KS, S, WI, TH, NW, KST, OCC, PREF) \
    return launch_sw_act<KS, S, WI, TH, NW, KST, OCC, PREF, ACT_SILU, ACT_SILU>(a, B, st)

int launch_sweep_mbconv(const SweepArgs& a, int B, int k, int stride, hipStream_t st) {
    MI355_REQUIRE(sweep_mbconv_supported(a.H, a.W, a.Cin, a.mid, k, stride, a.act_e, a.act_d) && a.Kp % 32 == 0 && a.Kp >= 32 && a.Kp <= 128,
                  "sweep_mbconv: unsupported shape %dx%d k%d s%d Cin %d", a.H, a.W, k, stride, a.Cin);
    const bool k2 = a.Kp == 64;
    const bool silu = a.act_e == ACT_SILU && a.act_d == ACT_SILU;
    const int cls = sw_class(a.H, a.W, k, stride);
    if (a.act_e == ACT_SILU && a.act_d == ACT_NONE && (sw_rex_instance(cls, a.Kp / 32) || (cls == SW_3_1_56 && a.Kp == 64))) {
        // RexNet (SiLU after the expand, linear depthwise).  REX(class geometry..., k-steps, workgroups per CU, X prefetch, tiles per pass)
#define REX(KS, S, WI, TH, NW, KST, OCC, PREF, NS) return launch_sw_act<KS, S, WI, TH, NW, KST, OCC, PREF, ACT_SILU, ACT_NONE, NS>(a, B, st)
        const int kst = a.Kp / 32;
        // measured at B = 256 (rexnet_150 / rexnet_200 forward): a pair of channel tiles per pass at 56 x 56 with one workgroup per CU:
        // 41->246 0.35 -> 0.22 ms, 58->348 s2 0.35 -> 0.29, 54->324 0.46 -> 0.31, 77->462 s2 0.42 -> 0.41; the same at 28 x 28, and
        // four-row bands for the stride-2 class, are level (and four-row bands change the order of the squeeze sums): not used
        if (a.variant == 3) {      // tuning: one tile per pass at 56 x 56 as well
            if (cls == SW_3_1_56) REX(3, 1, 56, 8, 8, 2, 1, true, 1);
            if (cls == SW_3_2_56) { if (kst == 2) REX(3, 2, 56, 2, 7, 2, 1, true, 1); REX(3, 2, 56, 2, 7, 3, 2, false, 1); }
        }
        if (cls == SW_3_1_56) REX(3, 1, 56, 8, 8, 2, 1, true, 2);
        if (cls == SW_3_2_56) { if (kst == 2) REX(3, 2, 56, 2, 7, 2, 1, true, 2); REX(3, 2, 56, 2, 7, 3, 1, false, 2); }
        if (cls == SW_3_1_28) { if (kst == 3) REX(3, 1, 28, 4, 7, 3, 2, false, 1); REX(3, 1, 28, 4, 7, 4, 2, false, 1); }
        REX(3, 2, 28, 4, 7, 3, 2, false, 1);
#undef REX
    }
    MI355_REQUIRE(a.Kp <= 64 && cls <= SW_3_2_28, "sweep_mbconv: no instance for %dx%d k%d s%d Kp %d", a.H, a.W, k, stride, a.Kp);
    const int v = (silu && k2 == (cls == SW_5_1_28 || cls == SW_3_2_28)) ? a.variant : 0;
    switch (cls) {
        case SW_3_2_112:
            if (v == 1) SW_VT(3, 2, 112, 2, 7, 1, 2, true);
            if (v == 2) SW_VT(3, 2, 112, 2, 8, 1, 3, false);
            if (v == 3) SW_VT(3, 2, 112, 1, 7, 1, 2, true);
            if (v == 4) SW_VT(3, 2, 112, 1, 4, 1, 4, true);
            SW_V0(3, 2, 112, 2, 8, 2, true);
        case SW_3_1_56:
            if (v == 1) SW_VT(3, 1, 56, 8, 7, 1, 2, true);
            if (v == 2) SW_VT(3, 1, 56, 4, 8, 1, 2, true);
            if (v == 3) SW_VT(3, 1, 56, 2, 7, 1, 2, true);
            if (v == 4) SW_VT(3, 1, 56, 2, 7, 1, 3, true);
            SW_V0(3, 1, 56, 8, 8, 2, true);
        case SW_5_2_56:
            if (v == 1) SW_VT(5, 2, 56, 4, 7, 1, 2, true);
            if (v == 2) SW_VT(5, 2, 56, 2, 8, 1, 2, true);
            if (v == 3) SW_VT(5, 2, 56, 2, 7, 1, 2, false);
            if (v == 4) SW_VT(5, 2, 56, 1, 7, 1, 2, true);
            SW_V0(5, 2, 56, 2, 7, 2, true);
        case SW_5_1_28:
            if (v == 1) SW_VT(5, 1, 28, 4, 7, 2, 2, true);
            if (v == 2) SW_VT(5, 1, 28, 4, 8, 2, 2, false);
            if (v == 3) SW_VT(5, 1, 28, 2, 7, 2, 2, false);
            if (v == 4) SW_VT(5, 1, 28, 4, 4, 2, 4, false);
            SW_V0(5, 1, 28, 4, 7, 2, false);
        default:
            if (v == 1) SW_VT(3, 2, 28, 7, 7, 2, 2, false);
            if (v == 2) SW_VT(3, 2, 28, 4, 8, 2, 2, true);
            if (v == 3) SW_VT(3, 2, 28, 2, 7, 2, 2, true);
            if (v == 4) SW_VT(3, 2, 28, 4, 7, 2, 2, true);
            SW_V0(3, 2, 28, 4, 7, 2, false);
    }
}
#undef SW_V0
#undef SW_VT

}  // namespace mi355
