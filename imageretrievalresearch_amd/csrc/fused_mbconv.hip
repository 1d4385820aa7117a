// Fused MBConv front half for the late stages: 1x1 expand (MFMA) -> bias+act -> depthwise kxk -> bias+act
// -> SE squeeze, with the expanded tensor living ONLY in LDS.  gfx950 only.
//
// Why: in the layer-granular traffic model the expanded activation (6x the block width) is written by the
// expand conv and read back by the depthwise conv — 46 % of all HBM bytes of EfficientNet-B3.  MI355X has
// 160 KB of LDS per CU, enough to hold a whole 14x14 or 7x7 image of X (all input channels) plus a 128..512
// channel slab of the expanded tensor, so one workgroup per image can run expand -> depthwise back to back:
// HBM sees X once and the depthwise output once.  Whole-image tiles need no halo recompute, and the SE
// squeeze of a channel is complete inside one workgroup (no partial sums, fixed summation order).
//
// Workgroup = 8 waves.  Per channel slab (MC = 128*NIW channels):
//   phase 1  each wave owns 16*NIW output channels and sweeps all pixels: A fragments (pixels) come from the
//            LDS image of X, W fragments straight from L2 in MFMA layout (each W element is read once per
//            workgroup), D = W x X^T so a lane holds 4 consecutive channels of one pixel -> 8-byte LDS writes
//   phase 2  thread = 8 channels x PX output pixels of one row, taps read from the LDS slab (zero padding by
//            bounds checks), result streamed to HBM as 16-byte NHWC vectors; per-channel sums reduced through
//            LDS in thread order.
#include "ops.h"

namespace mi355 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void fl_unpack8(u32x4 v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}

constexpr int FL_THREADS = 512;

template <int KS, int S, int NIW, int PX>
__global__ __launch_bounds__(FL_THREADS) void k_fused_late(const FusedArgs a) {
    constexpr int MC = 128 * NIW;                               // channels per slab
    constexpr int MTP = NIW == 1 ? 13 : (NIW == 2 ? 7 : 4);     // 16-pixel sub-tiles per GEMM pass
    constexpr int PAD = KS / 2;
    constexpr int IW = (PX - 1) * S + KS;
    constexpr int CGC = MC / 8;                                 // channel groups per slab
    extern __shared__ __attribute__((aligned(16))) bf16_t fsm[];
    const int P = a.H * a.W;
    const int MT = (P + 15) >> 4;
    const int XLD = a.Kp + 8;
    constexpr int ELD = MC + 8;
    // E slab image: rows of (W + 2*PAD) pixels with zero columns left and right, so the depthwise taps need no
    // x bounds checks (the branchy checked form costs ~8x the FMAs it feeds); EP pixels + IW slack for the
    // discarded lanes of a partial strip.
    const int EW = a.W + 2 * PAD;
    const int EP = a.H * EW + IW;
    bf16_t* Xs = fsm;                        // [MT*16][XLD]
    bf16_t* Es = fsm + (size_t)MT * 16 * XLD;  // [EP][ELD]
    float* red = reinterpret_cast<float*>(Es + (size_t)((EP + 7) & ~7) * ELD);   // [FL_THREADS][8]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int nchunks = (a.mid + MC - 1) / MC;
    const int cpb = (nchunks + gridDim.x - 1) / gridDim.x;
    const int chunk0 = blockIdx.x * cpb;
    const int chunk1 = min(nchunks, chunk0 + cpb);

    // ---- phase 0: X[b] -> LDS (zero fill for k >= Cin and rows >= P).  Four 16-byte loads are issued per thread
    // before the first LDS store (one load in flight per thread left the whole CU waiting on HBM latency 8 times over).
    {
        const int kc = a.Kp >> 3;
        const bf16_t* xb = a.X + (size_t)b * P * a.Cin;
        const int total = MT * 16 * kc;
        constexpr int U = 4;
        // zero the whole E image once: the pad columns are never written again
        for (int id = tid; id < EP * (ELD / 8); id += FL_THREADS)
            *reinterpret_cast<u32x4*>(&Es[(size_t)id * 8]) = (u32x4){0u, 0u, 0u, 0u};
        for (int id0 = tid; id0 < total; id0 += FL_THREADS * U) {
            u32x4 v[U];
            int dst[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int id = id0 + u * FL_THREADS;
                const int row = id / kc, c = id - row * kc;
                v[u] = (u32x4){0u, 0u, 0u, 0u};
                dst[u] = id < total ? row * XLD + c * 8 : -1;
                if (id < total && row < P && c * 8 < a.Cin) v[u] = *reinterpret_cast<const u32x4*>(xb + (size_t)row * a.Cin + c * 8);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (dst[u] >= 0) *reinterpret_cast<u32x4*>(&Xs[dst[u]]) = v[u];
        }
    }
    __syncthreads();

    const int fr = lane & 15, fk = (lane >> 4) * 8;
    const int midp = (a.mid + 15) & ~15;
    const int strips = (a.Wo + PX - 1) / PX;
    const int nitems = CGC * strips * a.Ho;

    // Every workgroup needs every slab's weights; start each one at a different slab (rotated by image index) so the
    // 256 co-resident workgroups do not stream the same W lines through the same L2 channels in lock-step.
    const int nch = chunk1 - chunk0;
    for (int ci = 0; ci < nch; ++ci) {
        const int chunk = chunk0 + (ci + b) % nch;
        const int cbase = chunk * MC;
        // ---- phase 1: E slab = act(X W^T + b) -> Es (bf16)
        // Wave layout: CW channel tiles (16 channels each) per wave x all pixels (NIW >= 2), or - for the one-image
        // case - 2 channel tiles x half of the pixel tiles, so that every A fragment read from LDS feeds two MFMAs
        // (with one channel tile per wave the eight waves re-read the whole X image 8x and phase 1 was LDS-bound).
        constexpr int CW = NIW == 1 ? 2 : NIW;              // channel tiles per wave
        constexpr int MSPLIT = NIW == 1 ? 2 : 1;            // pixel-tile halves
        constexpr int MTW = (MTP + MSPLIT - 1) / MSPLIT;    // pixel tiles per wave and pass
        const int cw = NIW == 1 ? (wave & 3) : wave;        // which group of CW channel tiles
        const int mh = NIW == 1 ? (wave >> 2) : 0;          // which half of the pixel tiles
        if (!(a.debug_skip & 1))
        for (int mb0 = 0; mb0 < MT; mb0 += MTP) {
            const int mb = mb0 + mh * MTW;
            f32x4 acc[CW][MTW];
#pragma unroll
            for (int j = 0; j < CW; ++j)
#pragma unroll
                for (int m = 0; m < MTW; ++m) acc[j][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const bf16_t* wrow[CW];
            bool wok[CW];
#pragma unroll
            for (int j = 0; j < CW; ++j) {
                const int n = cbase + (cw * CW + j) * 16 + fr;
                wok[j] = n < midp;
                wrow[j] = a.We + (size_t)(wok[j] ? n : 0) * a.Kp + fk;
            }
            u32x4 wnext[CW];
#pragma unroll
            for (int j = 0; j < CW; ++j) wnext[j] = wok[j] ? *reinterpret_cast<const u32x4*>(wrow[j]) : (u32x4){0u, 0u, 0u, 0u};
            for (int ks = 0; ks < a.Kp; ks += 32) {
                bf16x8 wf[CW];
#pragma unroll
                for (int j = 0; j < CW; ++j) {
                    wf[j] = *reinterpret_cast<bf16x8*>(&wnext[j]);
                    // prefetch the next k-step's W fragment while this step's MFMAs run
                    if (ks + 32 < a.Kp && wok[j]) wnext[j] = *reinterpret_cast<const u32x4*>(wrow[j] + ks + 32);
                }
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    if (mb + m < MT && mh * MTW + m < MTP) {
                        const bf16x8 af = *reinterpret_cast<const bf16x8*>(&Xs[((mb + m) * 16 + fr) * XLD + ks + fk]);
#pragma unroll
                        for (int j = 0; j < CW; ++j)
                            acc[j][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af, acc[j][m], 0, 0, 0);
                    }
                }
            }
            MI355_ACT_DISPATCH(a.act_e, {
_Pragma("unroll")
                for (int j = 0; j < CW; ++j) {
                    const int n = cbase + (cw * CW + j) * 16 + (lane >> 4) * 4;
                    f32x4 bb = {0.f, 0.f, 0.f, 0.f};
                    if (n < midp) bb = *reinterpret_cast<const f32x4*>(a.be + n);
_Pragma("unroll")
                    for (int m = 0; m < MTW; ++m) {
                        acc[j][m].x = act_c<ACT>(acc[j][m].x + bb.x); acc[j][m].y = act_c<ACT>(acc[j][m].y + bb.y);
                        acc[j][m].z = act_c<ACT>(acc[j][m].z + bb.z); acc[j][m].w = act_c<ACT>(acc[j][m].w + bb.w);
                    }
                }
            })
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const int p = (mb + m) * 16 + fr;          // this lane's pixel
                if (mb + m < MT && mh * MTW + m < MTP && p < P) {
                    const int y = p / a.W;
                    const int erow = y * EW + (p - y * a.W) + PAD;
#pragma unroll
                    for (int j = 0; j < CW; ++j) {
                        const int nl = (cw * CW + j) * 16 + (lane >> 4) * 4;   // channel within the slab
                        u32x2 o;
                        o.x = pack2bf(acc[j][m].x, acc[j][m].y);
                        o.y = pack2bf(acc[j][m].z, acc[j][m].w);
                        *reinterpret_cast<u32x2*>(&Es[(size_t)erow * ELD + nl]) = o;
                    }
                }
            }
        }
        __syncthreads();

        // ---- phase 2: depthwise from the LDS slab
        float psum[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) psum[j] = 0.f;
        const int my_cg = tid % CGC;                 // FL_THREADS % CGC == 0: a thread keeps one channel group
        const int c0 = cbase + my_cg * 8;
        const bool cok = c0 < a.mid;
        if (!(a.debug_skip & 2))
        for (int item = tid; item < nitems; item += FL_THREADS) {
            const int rest = item / CGC;
            const int sx = rest % strips, oy = rest / strips;
            const int ox0 = sx * PX;
            if (!cok) continue;
            float acc[PX][8];
            {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bd + c0);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(a.bd + c0 + 4);
#pragma unroll
                for (int p = 0; p < PX; ++p) {
                    acc[p][0] = b0.x; acc[p][1] = b0.y; acc[p][2] = b0.z; acc[p][3] = b0.w;
                    acc[p][4] = b1.x; acc[p][5] = b1.y; acc[p][6] = b1.z; acc[p][7] = b1.w;
                }
            }
#pragma unroll 1
            for (int ky = 0; ky < KS; ++ky) {
                const int iy = oy * S - PAD + ky;
                if (iy < 0 || iy >= a.H) continue;
                float wk[KS][8];
#pragma unroll
                for (int kx = 0; kx < KS; ++kx)
                    fl_unpack8(*reinterpret_cast<const u32x4*>(a.Wd + (size_t)(ky * KS + kx) * a.mid + c0), wk[kx]);
                const bf16_t* erow = Es + (size_t)(iy * EW + ox0 * S) * ELD + my_cg * 8;   // padded x: no checks
#pragma unroll
                for (int i = 0; i < IW; ++i) {
                    float v[8];
                    fl_unpack8(*reinterpret_cast<const u32x4*>(erow + (size_t)i * ELD), v);
#pragma unroll
                    for (int p = 0; p < PX; ++p) {
                        const int kx = i - p * S;
                        if (kx >= 0 && kx < KS) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[p][j] += wk[kx][j] * v[j];
                        }
                    }
                }
            }
            bf16_t* o = a.D + (((size_t)b * a.Ho + oy) * a.Wo + ox0) * a.mid + c0;
            MI355_ACT_DISPATCH(a.act_d, {
_Pragma("unroll")
                for (int p = 0; p < PX; ++p)
_Pragma("unroll")
                    for (int j = 0; j < 8; ++j) acc[p][j] = act_c<ACT>(acc[p][j]);
            })
#pragma unroll
            for (int p = 0; p < PX; ++p) {
                if (ox0 + p < a.Wo) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) psum[j] += acc[p][j];
                    u32x4 ov;
                    ov.x = pack2bf(acc[p][0], acc[p][1]); ov.y = pack2bf(acc[p][2], acc[p][3]);
                    ov.z = pack2bf(acc[p][4], acc[p][5]); ov.w = pack2bf(acc[p][6], acc[p][7]);
                    *reinterpret_cast<u32x4*>(o + (size_t)p * a.mid) = ov;
                }
            }
        }
        if (a.pool != nullptr) {
            // lanes l, l+CGC, l+2*CGC.. of a wave hold the same channel group: fold them with shuffles (fixed order),
            // then one LDS pass over the 8 per-wave partials.
#pragma unroll
            for (int o = 32; o >= CGC; o >>= 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) psum[j] += __shfl_xor(psum[j], o, 64);
            }
            constexpr int LPW = CGC < 64 ? CGC : 64;          // distinct channel groups per wave
            if (lane < LPW) {
#pragma unroll
                for (int j = 0; j < 8; ++j) red[(wave * LPW + lane) * 8 + j] = psum[j];
            }
            __syncthreads();
            if (tid < CGC && cbase + tid * 8 < a.mid) {
                float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if constexpr (CGC <= 64) {
#pragma unroll
                    for (int w = 0; w < FL_THREADS / 64; ++w) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) s[j] += red[(w * LPW + tid) * 8 + j];
                    }
                }
                float* pp = a.pool + (size_t)b * a.mid + cbase + tid * 8;
                *reinterpret_cast<f32x4*>(pp) = (f32x4){s[0], s[1], s[2], s[3]};
                *reinterpret_cast<f32x4*>(pp + 4) = (f32x4){s[4], s[5], s[6], s[7]};
            }
        }
        __syncthreads();   // Es / red are rewritten by the next slab
    }
}

// =====================================================================================
// Band variant for the early stages (28x28 .. 112x112, Cin <= 64): a workgroup owns TH output rows of one image
// (full width) plus the KS-S halo rows above/below, recomputes the expand GEMM for the halo (cheap: K <= 64), and
// walks the expanded channels in slabs of 64.  Here the waves split the PIXELS (each wave sweeps 16-pixel
// sub-tiles with the slab's W fragments held in registers); phase 2 is the same depthwise as above.
// The SE squeeze becomes one partial sum per (image, band): pool[b][band][mid].
// =====================================================================================
template <int KS, int S, int KST, int PX, int MC>
__global__ __launch_bounds__(FL_THREADS) void k_fused_band(const FusedArgs a) {
    // MC = channels per slab (48 / 64 / 96, picked to divide the expanded width: a 64-wide slab wasted a third of both
    // phases on mid = 144); NACT = the largest multiple of the slab's channel groups <= 512, so that a thread keeps ONE
    // channel group over all its items (its squeeze partial sums stay per channel)
    constexpr int NT = MC / 16;
    constexpr int PAD = KS / 2;
    constexpr int IW = (PX - 1) * S + KS;
    constexpr int CGC = MC / 8;
    constexpr int NACT = (FL_THREADS / CGC) * CGC;
    constexpr int Kp = 32 * KST, XLD = Kp + 8, ELD = MC + 8;
    extern __shared__ __attribute__((aligned(16))) bf16_t fsm[];
    const int band = blockIdx.x, b = blockIdx.y;
    const int oy0 = band * a.TH;
    const int th = min(a.TH, a.Ho - oy0);
    const int IH = (a.TH - 1) * S + KS;
    const int iy0 = oy0 * S - PAD;
    const int P = IH * a.W;
    const int MT = (P + 15) >> 4;
    const int EW = a.W + 2 * PAD;
    const int EP = IH * EW + IW;
    bf16_t* Xs = fsm;
    bf16_t* Es = fsm + (size_t)MT * 16 * XLD;
    float* red = reinterpret_cast<float*>(Es + (size_t)((EP + 7) & ~7) * ELD);   // [FL_THREADS][8]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        // X band -> LDS with four 16-byte loads in flight per thread; of the E image only the pad columns (and the slack
        // behind the last row) must be zero - phase 1 rewrites every interior pixel for every slab
        constexpr int kc = Kp >> 3;
        constexpr int U = 4;
        const bf16_t* xb = a.X + (size_t)b * a.H * a.W * a.Cin;
        const int total = MT * 16 * kc;
        for (int id0 = tid; id0 < total; id0 += FL_THREADS * U) {
            u32x4 v[U];
            int dst[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int id = id0 + u * FL_THREADS;
                const int row = id / kc, c = id - row * kc;
                const int r = row / a.W, x = row - r * a.W;
                const int iy = iy0 + r;
                v[u] = (u32x4){0u, 0u, 0u, 0u};
                dst[u] = id < total ? row * XLD + c * 8 : -1;
                if (id < total && row < P && iy >= 0 && iy < a.H && c * 8 < a.Cin)
                    v[u] = *reinterpret_cast<const u32x4*>(xb + ((size_t)iy * a.W + x) * a.Cin + c * 8);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (dst[u] >= 0) *reinterpret_cast<u32x4*>(&Xs[dst[u]]) = v[u];
        }
        constexpr int EV = ELD / 8;                      // 16-byte vectors per E pixel
        const int npad = IH * 2 * PAD + IW;              // pad pixels per row (left + right) and the tail slack
        for (int id = tid; id < npad * EV; id += FL_THREADS) {
            const int q = id / EV, vv = id - q * EV;
            int px;
            if (q < IH * 2 * PAD) {
                const int r = q / (2 * PAD), j = q - r * 2 * PAD;
                px = r * EW + (j < PAD ? j : a.W + j);    // j >= PAD: right pad column PAD + W + (j - PAD)
            } else {
                px = IH * EW + (q - IH * 2 * PAD);
            }
            *reinterpret_cast<u32x4*>(&Es[(size_t)px * ELD + vv * 8]) = (u32x4){0u, 0u, 0u, 0u};
        }
    }
    __syncthreads();

    const int fr = lane & 15, fq = lane >> 4, fk = fq * 8;
    const int midp = (a.mid + 15) & ~15;
    const int strips = (a.Wo + PX - 1) / PX;
    const int nitems = CGC * strips * th;
    const int nchunks = (a.mid + MC - 1) / MC;

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int cbase = chunk * MC;
        // ---- phase 1: waves split the pixel sub-tiles; the slab's W fragments + bias live in registers
        if (!(a.debug_skip & 1)) {
            bf16x8 wf[NT][KST];
            f32x4 bb[NT];
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
                const int n = cbase + ni * 16 + fr;
#pragma unroll
                for (int ks = 0; ks < KST; ++ks) {
                    u32x4 v = {0u, 0u, 0u, 0u};
                    if (n < midp) v = *reinterpret_cast<const u32x4*>(a.We + (size_t)n * a.Kp + ks * 32 + fk);
                    wf[ni][ks] = *reinterpret_cast<bf16x8*>(&v);
                }
                const int n4 = cbase + ni * 16 + fq * 4;
                bb[ni] = n4 < midp ? *reinterpret_cast<const f32x4*>(a.be + n4) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            for (int mt = wave; mt < MT; mt += FL_THREADS / 64) {
                bf16x8 af[KST];
#pragma unroll
                for (int ks = 0; ks < KST; ++ks)
                    af[ks] = *reinterpret_cast<const bf16x8*>(&Xs[(mt * 16 + fr) * XLD + ks * 32 + fk]);
                f32x4 acc[NT];
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) {
                    acc[ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KST; ++ks)
                        acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni][ks], af[ks], acc[ni], 0, 0, 0);
                }
                MI355_ACT_DISPATCH(a.act_e, {
_Pragma("unroll")
                    for (int ni = 0; ni < NT; ++ni) {
                        acc[ni].x = act_c<ACT>(acc[ni].x + bb[ni].x); acc[ni].y = act_c<ACT>(acc[ni].y + bb[ni].y);
                        acc[ni].z = act_c<ACT>(acc[ni].z + bb[ni].z); acc[ni].w = act_c<ACT>(acc[ni].w + bb[ni].w);
                    }
                })
                const int p = mt * 16 + fr;
                if (p < P) {
                    const int r = p / a.W;
                    const int erow = r * EW + (p - r * a.W) + PAD;
#pragma unroll
                    for (int ni = 0; ni < NT; ++ni) {
                        u32x2 o;
                        o.x = pack2bf(acc[ni].x, acc[ni].y);
                        o.y = pack2bf(acc[ni].z, acc[ni].w);
                        *reinterpret_cast<u32x2*>(&Es[(size_t)erow * ELD + ni * 16 + fq * 4]) = o;
                    }
                }
            }
        }
        __syncthreads();

        // ---- phase 2: depthwise from the LDS slab (rows outside the image are skipped, x is zero padded)
        float psum[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) psum[j] = 0.f;
        const int my_cg = tid % CGC;
        const int c0 = cbase + my_cg * 8;
        const bool cok = c0 < a.mid && tid < NACT;
        if (!(a.debug_skip & 2))
        for (int item = tid; item < nitems && tid < NACT; item += NACT) {
            const int rest = item / CGC;
            const int sx = rest % strips, oy = oy0 + rest / strips;
            const int ox0 = sx * PX;
            if (!cok) continue;
            float acc[PX][8];
            {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bd + c0);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(a.bd + c0 + 4);
#pragma unroll
                for (int p = 0; p < PX; ++p) {
                    acc[p][0] = b0.x; acc[p][1] = b0.y; acc[p][2] = b0.z; acc[p][3] = b0.w;
                    acc[p][4] = b1.x; acc[p][5] = b1.y; acc[p][6] = b1.z; acc[p][7] = b1.w;
                }
            }
#pragma unroll 1
            for (int ky = 0; ky < KS; ++ky) {
                const int iy = oy * S - PAD + ky;
                if (iy < 0 || iy >= a.H) continue;
                float wk[KS][8];
#pragma unroll
                for (int kx = 0; kx < KS; ++kx)
                    fl_unpack8(*reinterpret_cast<const u32x4*>(a.Wd + (size_t)(ky * KS + kx) * a.mid + c0), wk[kx]);
                const bf16_t* erow = Es + (size_t)((iy - iy0) * EW + ox0 * S) * ELD + my_cg * 8;
#pragma unroll
                for (int i = 0; i < IW; ++i) {
                    float v[8];
                    fl_unpack8(*reinterpret_cast<const u32x4*>(erow + (size_t)i * ELD), v);
#pragma unroll
                    for (int p = 0; p < PX; ++p) {
                        const int kx = i - p * S;
                        if (kx >= 0 && kx < KS) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[p][j] += wk[kx][j] * v[j];
                        }
                    }
                }
            }
            bf16_t* o = a.D + (((size_t)b * a.Ho + oy) * a.Wo + ox0) * a.mid + c0;
            MI355_ACT_DISPATCH(a.act_d, {
_Pragma("unroll")
                for (int p = 0; p < PX; ++p)
_Pragma("unroll")
                    for (int j = 0; j < 8; ++j) acc[p][j] = act_c<ACT>(acc[p][j]);
            })
#pragma unroll
            for (int p = 0; p < PX; ++p) {
                if (ox0 + p < a.Wo) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) psum[j] += acc[p][j];
                    u32x4 ov;
                    ov.x = pack2bf(acc[p][0], acc[p][1]); ov.y = pack2bf(acc[p][2], acc[p][3]);
                    ov.z = pack2bf(acc[p][4], acc[p][5]); ov.w = pack2bf(acc[p][6], acc[p][7]);
                    *reinterpret_cast<u32x4*>(o + (size_t)p * a.mid) = ov;
                }
            }
        }
        if (a.pool != nullptr) {
#pragma unroll
            for (int j = 0; j < 8; ++j) red[tid * 8 + j] = psum[j];
            __syncthreads();
            // two fixed-order stages: 8 threads per channel group add a contiguous share of that group's NACT / CGC
            // entries, then one thread per group adds the 8 parts
            constexpr int PER = NACT / CGC, SHARE = (PER + 7) / 8;
            float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (tid < CGC * 8) {
                const int cg = tid % CGC, part = tid / CGC;
                for (int i = part * SHARE; i < (part + 1) * SHARE && i < PER; ++i) {
                    const int u = cg + CGC * i;
#pragma unroll
                    for (int j = 0; j < 8; ++j) s[j] += red[u * 8 + j];
                }
            }
            __syncthreads();
            if (tid < CGC * 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) red[tid * 8 + j] = s[j];
            }
            __syncthreads();
            if (tid < CGC && cbase + tid * 8 < a.mid) {
                float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int part = 0; part < 8; ++part)
#pragma unroll
                    for (int j = 0; j < 8; ++j) t[j] += red[(part * CGC + tid) * 8 + j];
                float* pp = a.pool + ((size_t)b * gridDim.x + band) * a.mid + cbase + tid * 8;
                *reinterpret_cast<f32x4*>(pp) = (f32x4){t[0], t[1], t[2], t[3]};
                *reinterpret_cast<f32x4*>(pp + 4) = (f32x4){t[4], t[5], t[6], t[7]};
            }
        }
        __syncthreads();
    }
}

static int band_slab(int mid) { return mid % 64 == 0 ? 64 : (mid % 48 == 0 ? 48 : (mid % 96 == 0 ? 96 : 64)); }

static size_t band_lds_bytes(int W, int Kp, int k, int stride, int TH, int px, int mc) {
    const int pad = k / 2, IH = (TH - 1) * stride + k;
    const int P = IH * W, MT = (P + 15) / 16, iw = (px - 1) * stride + k;
    const int EP = IH * (W + 2 * pad) + iw;
    return (size_t)MT * 16 * (Kp + 8) * 2 + (size_t)((EP + 7) & ~7) * (mc + 8) * 2 + FL_THREADS * 8 * 4;
}

// Largest band height (output rows) whose LDS image fits; 0 = unsupported.
int fused_band_rows(int H, int W, int Cin, int mid, int k, int stride) {
    if (Cin % 8 || mid % 8 || Cin > 64 || W > 128) return 0;
    if (!((k == 3 || k == 5) && (stride == 1 || stride == 2))) return 0;
    const int Kp = (Cin + 31) & ~31, pad = k / 2;
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    const int px = Wo % 7 == 0 ? 7 : 4;
    int best = 0;
    for (int th = 1; th <= Ho && th <= 16; ++th)
        if (band_lds_bytes(W, Kp, k, stride, th, px, band_slab(mid)) <= 158 * 1024) best = th;
    return best;
}

template <int KS, int S, int KST, int PX, int MC>
static int launch_fb(const FusedArgs& a, int B, hipStream_t st) {
    const size_t lds = band_lds_bytes(a.W, a.Kp, KS, S, a.TH, PX, MC);
    static bool attr_done[MI355_MAX_DEVICES] = {false};   // per device
    if (first_time_on_this_device(attr_done)) {
        MI355_CHECK_HIP(hipFuncSetAttribute((const void*)k_fused_band<KS, S, KST, PX, MC>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    hipLaunchKernelGGL((k_fused_band<KS, S, KST, PX, MC>), dim3(cdiv(a.Ho, a.TH), B), dim3(FL_THREADS), lds, st, a);
    MI355_LAUNCH_CHECK();
    return OK;
}

template <int KS, int S, int KST, int PX>
static int launch_fb_mc(const FusedArgs& a, int B, hipStream_t st) {
    const int mc = band_slab(a.mid);
    if (mc == 48) return launch_fb<KS, S, KST, PX, 48>(a, B, st);
    if (mc == 96) return launch_fb<KS, S, KST, PX, 96>(a, B, st);
    return launch_fb<KS, S, KST, PX, 64>(a, B, st);
}

template <int KS, int S>
static int launch_fb_ks(const FusedArgs& a, int B, hipStream_t st) {
    const bool px7 = (a.Wo % 7 == 0);
    if (a.Kp == 32) return px7 ? launch_fb_mc<KS, S, 1, 7>(a, B, st) : launch_fb_mc<KS, S, 1, 4>(a, B, st);
    return px7 ? launch_fb_mc<KS, S, 2, 7>(a, B, st) : launch_fb_mc<KS, S, 2, 4>(a, B, st);
}

int launch_fused_band(const FusedArgs& a, int B, int k, int stride, hipStream_t st) {
    MI355_REQUIRE(a.TH >= 1 && (a.Kp == 32 || a.Kp == 64), "fused_band: unsupported shape");
    if (k == 3 && stride == 1) return launch_fb_ks<3, 1>(a, B, st);
    if (k == 3 && stride == 2) return launch_fb_ks<3, 2>(a, B, st);
    if (k == 5 && stride == 1) return launch_fb_ks<5, 1>(a, B, st);
    return launch_fb_ks<5, 2>(a, B, st);
}

static size_t fused_lds_bytes(int H, int W, int Kp, int MC, int k, int stride) {
    const int P = H * W, MT = (P + 15) / 16, pad = k / 2;
    const int px = ((W + 2 * pad - k) / stride + 1) % 7 == 0 ? 7 : 4;
    const int iw = (px - 1) * stride + k;
    const int EP = H * (W + 2 * pad) + iw;
    return (size_t)MT * 16 * (Kp + 8) * 2 + (size_t)((EP + 7) & ~7) * (MC + 8) * 2 + FL_THREADS * 8 * 4;
}

bool fused_late_supported(int H, int W, int Cin, int mid, int k, int stride) {
    if (H * W > 208 || Cin % 8 || mid % 8) return false;
    if (!((k == 3 || k == 5) && (stride == 1 || stride == 2))) return false;
    const int P = H * W;
    const int niw = P <= 64 ? 4 : (P <= 112 ? 2 : 1);
    const int Kp = (Cin + 31) & ~31;
    return fused_lds_bytes(H, W, Kp, 128 * niw, k, stride) <= 160 * 1024;
}

template <int KS, int S, int NIW, int PX>
static int launch_fl(const FusedArgs& a, int B, hipStream_t st) {
    constexpr int MC = 128 * NIW;
    const size_t lds = fused_lds_bytes(a.H, a.W, a.Kp, MC, KS, S);
    static bool attr_done[MI355_MAX_DEVICES] = {false};   // per device
    if (first_time_on_this_device(attr_done)) {
        MI355_CHECK_HIP(hipFuncSetAttribute((const void*)k_fused_late<KS, S, NIW, PX>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const int nchunks = cdiv(a.mid, MC);
    int G = 256 / (B > 0 ? B : 1);
    if (G < 1) G = 1;
    if (G > nchunks) G = nchunks;
    hipLaunchKernelGGL((k_fused_late<KS, S, NIW, PX>), dim3(G, B), dim3(FL_THREADS), lds, st, a);
    MI355_LAUNCH_CHECK();
    return OK;
}

template <int KS, int S>
static int launch_fl_ks(const FusedArgs& a, int B, hipStream_t st) {
    const int P = a.H * a.W;
    const bool px7 = (a.Wo % 7 == 0);
    if (P <= 64) return px7 ? launch_fl<KS, S, 4, 7>(a, B, st) : launch_fl<KS, S, 4, 4>(a, B, st);
    if (P <= 112) return px7 ? launch_fl<KS, S, 2, 7>(a, B, st) : launch_fl<KS, S, 2, 4>(a, B, st);
    return px7 ? launch_fl<KS, S, 1, 7>(a, B, st) : launch_fl<KS, S, 1, 4>(a, B, st);
}

int launch_fused_late(const FusedArgs& a, int B, int k, int stride, hipStream_t st) {
    MI355_REQUIRE(fused_late_supported(a.H, a.W, a.Cin, a.mid, k, stride), "fused_late: unsupported shape");
    if (k == 3 && stride == 1) return launch_fl_ks<3, 1>(a, B, st);
    if (k == 3 && stride == 2) return launch_fl_ks<3, 2>(a, B, st);
    if (k == 5 && stride == 1) return launch_fl_ks<5, 1>(a, B, st);
    return launch_fl_ks<5, 2>(a, B, st);
}

}  // namespace mi355
