// Non-GEMM kernels of the conv backbones: stem 3x3/s2, depthwise kxk (+ SE squeeze partial sums),
// SE gate, global average pool, layout conversion, conv_input pre-stem.  gfx950 only.
// Activations are NHWC bf16 (channels multiple of 8 -> every access is a 16-byte vector of 8 channels);
// all arithmetic is fp32 on the VALU: these layers are byte-bound, not FLOP-bound.
#include "ops.h"

#include <stdlib.h>

namespace mi355 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void unpack8(u32x4 v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
    u32x4 o;
    o.x = pack2bf(f[0], f[1]); o.y = pack2bf(f[2], f[3]); o.z = pack2bf(f[4], f[5]); o.w = pack2bf(f[6], f[7]);
    return o;
}

// =====================================================================================
// stem: 3x3 stride 2 pad 1, 3 -> Cout, NCHW fp32 in, NHWC bf16 out.  One thread = one output pixel,
// the 27-tap patch lives in registers; the weights [27][Cout] are read at wave-uniform addresses, i.e. as scalar
// loads feeding SGPR operands of packed FMAs (a broadcast-read LDS copy was LDS-issue-bound: 0.222 -> 0.196 ms).
// =====================================================================================
__global__ __launch_bounds__(256) void k_stem(const float* __restrict__ x, const float* __restrict__ w,
                                              const float* __restrict__ bias, bf16_t* __restrict__ out, int H, int W,
                                              int Ho, int Wo, int Cout, int act) {
    const int b = blockIdx.z;
    const int ox = blockIdx.x * 32 + (threadIdx.x & 31);
    const int oy = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (ox >= Wo || oy >= Ho) return;
    float p[27];
    const float* xb = x + (size_t)b * 3 * H * W;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * 2 - 1 + ky;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * 2 - 1 + kx;
            const bool ok = (iy >= 0 && iy < H && ix >= 0 && ix < W);
#pragma unroll
            for (int ci = 0; ci < 3; ++ci)
                p[(ky * 3 + kx) * 3 + ci] = ok ? xb[((size_t)ci * H + iy) * W + ix] : 0.f;
        }
    }
    bf16_t* o = out + (((size_t)b * Ho + oy) * Wo + ox) * Cout;
    for (int c0 = 0; c0 < Cout; c0 += 8) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = bias[c0 + j];
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            // wave-uniform addresses: the compiler turns these into scalar loads (SGPR operands of the FMAs)
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(&w[t * Cout + c0]);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(&w[t * Cout + c0 + 4]);
            acc[0] += p[t] * w0.x; acc[1] += p[t] * w0.y; acc[2] += p[t] * w0.z; acc[3] += p[t] * w0.w;
            acc[4] += p[t] * w1.x; acc[5] += p[t] * w1.y; acc[6] += p[t] * w1.z; acc[7] += p[t] * w1.w;
        }
        MI355_ACT_DISPATCH(act, {
_Pragma("unroll")
            for (int j = 0; j < 8; ++j) acc[j] = act_c<ACT>(acc[j]);
        })
        *reinterpret_cast<u32x4*>(o + c0) = pack8(acc);
    }
}

int launch_stem(const float* x, const float* w, const float* bias, bf16_t* out, int B, int H, int W, int Cout, int act,
                hipStream_t st) {
    MI355_REQUIRE(Cout % 8 == 0 && Cout <= 256, "stem: Cout=%d must be a multiple of 8 and <= 256", Cout);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    dim3 grid(cdiv(Wo, 32), cdiv(Ho, 8), B);
    hipLaunchKernelGGL(k_stem, grid, dim3(256), 0, st, x, w, bias, out, H, W, Ho, Wo, Cout, act);
    MI355_LAUNCH_CHECK();
    return OK;
}

// =====================================================================================
// Fused input: uint8 HWC image -> SquarePad(fill) -> /255 -> Normalize -> [conv_input 3x3 + SiLU] -> stem 3x3/s2 ->
// NHWC bf16, one kernel, no fp32 NCHW batch in HBM (SURVEY §8f f-1 + §8a a5; utils/square_pad.py:20-36,
// inference/inference.py:48-52,101-105).  Same thread = output pixel structure and the SAME fp32 operation order as
// k_square_pad_normalize -> k_conv_input_silu -> k_stem, so its output is bit-identical to that chain.
// =====================================================================================
struct StemU8Args {
    const unsigned char* img;   // [B][h][w][3]
    int h, w, S, hp, vp, fill;  // S = max(h, w); hp / vp = left / top padding
    float mean[3], stdv[3];
    const float* cw;            // conv_input weights [3][3][3][3] (co, ci, ky, kx) on the device, or null
};

template <bool CONV_INPUT>
__global__ __launch_bounds__(256) void k_stem_u8(const StemU8Args a, const float* __restrict__ w,
                                                 const float* __restrict__ bias, bf16_t* __restrict__ out, int Ho, int Wo,
                                                 int Cout, int act) {
    // CONV_INPUT: the workgroup's 32 x 8 output pixels need conv_input + SiLU on a 65 x 17 window, which needs the
    // pre-processed image on a 67 x 19 window.  Both windows are computed ONCE per workgroup into LDS (the first version
    // recomputed a private 5 x 5 window and nine conv_input taps per thread: 256 VGPRs + 298 spilled, 2.65 ms per batch).
    constexpr int PW = 67, PH = 19, CWD = 65, CHT = 17;
    __shared__ float scw[81];
    __shared__ float Pt[CONV_INPUT ? PH * PW * 3 : 1];
    __shared__ float Ct[CONV_INPUT ? CHT * CWD * 3 : 1];
    const int b = blockIdx.z;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int ox = blockIdx.x * 32 + lx;
    const int oy = blockIdx.y * 8 + ly;
    const unsigned char* ib = a.img + (size_t)b * a.h * a.w * 3;
    // the model input at (y, x, c): 0 outside the S x S square (the convolutions' zero padding), the normalised fill
    // colour in the SquarePad border, the normalised pixel inside the image - one rounding per fp32 op, as torch does
    auto pre = [&](int y, int x, int c) -> float {
        if (y < 0 || y >= a.S || x < 0 || x >= a.S) return 0.f;
        const int iy = y - a.vp, ix = x - a.hp;
        int v = a.fill;
        if (iy >= 0 && iy < a.h && ix >= 0 && ix < a.w) v = ib[((size_t)iy * a.w + ix) * 3 + c];
        return __fdiv_rn(__fsub_rn(__fdiv_rn((float)v, 255.0f), a.mean[c]), a.stdv[c]);
    };
    float p[27];
    if (!CONV_INPUT) {
        if (ox >= Wo || oy >= Ho) return;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) p[(ky * 3 + kx) * 3 + ci] = pre(oy * 2 - 1 + ky, ox * 2 - 1 + kx, ci);
    } else {
        if (threadIdx.x < 81) scw[threadIdx.x] = a.cw[threadIdx.x];
        const int y0 = blockIdx.y * 16 - 2, x0 = blockIdx.x * 64 - 2;      // pre-processed window origin
        for (int i = threadIdx.x; i < PH * PW; i += 256) {
            const int r = i / PW, c = i - r * PW;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) Pt[i * 3 + ci] = pre(y0 + r, x0 + c, ci);
        }
        __syncthreads();
        // conv_input + SiLU on the window the stem taps touch (origin y0 + 1, x0 + 1); same (ci, ky, kx) order as
        // k_conv_input_silu; positions outside the S x S square are the stem's zero padding
        for (int i = threadIdx.x; i < CHT * CWD; i += 256) {
            const int r = i / CWD, c = i - r * CWD;
            const int y = y0 + 1 + r, x = x0 + 1 + c;
            float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int ci = 0; ci < 3; ++ci)
#pragma unroll
                for (int ky2 = 0; ky2 < 3; ++ky2)
#pragma unroll
                    for (int kx2 = 0; kx2 < 3; ++kx2) {
                        const float v = Pt[((r + ky2) * PW + c + kx2) * 3 + ci];
#pragma unroll
                        for (int co = 0; co < 3; ++co) acc[co] += v * scw[((co * 3 + ci) * 3 + ky2) * 3 + kx2];
                    }
            const bool in = y >= 0 && y < a.S && x >= 0 && x < a.S;
#pragma unroll
            for (int co = 0; co < 3; ++co) Ct[i * 3 + co] = in ? silu_f(acc[co]) : 0.f;
        }
        __syncthreads();
        if (ox >= Wo || oy >= Ho) return;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int co = 0; co < 3; ++co) p[(ky * 3 + kx) * 3 + co] = Ct[((ly * 2 + ky) * CWD + lx * 2 + kx) * 3 + co];
    }
    bf16_t* o = out + (((size_t)b * Ho + oy) * Wo + ox) * Cout;
    for (int c0 = 0; c0 < Cout; c0 += 8) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = bias[c0 + j];
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(&w[t * Cout + c0]);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(&w[t * Cout + c0 + 4]);
            acc[0] += p[t] * w0.x; acc[1] += p[t] * w0.y; acc[2] += p[t] * w0.z; acc[3] += p[t] * w0.w;
            acc[4] += p[t] * w1.x; acc[5] += p[t] * w1.y; acc[6] += p[t] * w1.z; acc[7] += p[t] * w1.w;
        }
        MI355_ACT_DISPATCH(act, {
_Pragma("unroll")
            for (int j = 0; j < 8; ++j) acc[j] = act_c<ACT>(acc[j]);
        })
        *reinterpret_cast<u32x4*>(o + c0) = pack8(acc);
    }
}

int launch_stem_u8(const unsigned char* img, int h, int w, int fill, const float* mean, const float* stdv,
                   const float* conv_w, const float* sw, const float* bias, bf16_t* out, int B, int Cout, int act,
                   hipStream_t st) {
    MI355_REQUIRE(Cout % 8 == 0 && Cout <= 256, "stem: Cout=%d must be a multiple of 8 and <= 256", Cout);
    StemU8Args a{};
    a.img = img; a.h = h; a.w = w; a.S = h > w ? h : w; a.hp = (a.S - w) / 2; a.vp = (a.S - h) / 2; a.fill = fill;
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean[c]; a.stdv[c] = stdv[c]; }
    a.cw = conv_w;
    const int Ho = (a.S + 2 - 3) / 2 + 1, Wo = Ho;
    dim3 grid(cdiv(Wo, 32), cdiv(Ho, 8), B);
    if (conv_w) hipLaunchKernelGGL((k_stem_u8<true>), grid, dim3(256), 0, st, a, sw, bias, out, Ho, Wo, Cout, act);
    else hipLaunchKernelGGL((k_stem_u8<false>), grid, dim3(256), 0, st, a, sw, bias, out, Ho, Wo, Cout, act);
    MI355_LAUNCH_CHECK();
    return OK;
}

// =====================================================================================
// depthwise k x k.  One thread = 8 channels x PX consecutive output pixels of one row.
// Threads are laid out channel-group fastest, so a wave reads/writes contiguous NHWC bytes.
// SE squeeze: each thread sums its (un-rounded, activated) outputs; threads of a block that share a
// channel group are combined through LDS in thread order -> pool_partial[b][blk][C], deterministic.
// =====================================================================================
#ifndef MI355_DW_PX
#define MI355_DW_PX 4
#endif
constexpr int DW_PX = MI355_DW_PX;

int dw_pool_blocks(int Ho, int Wo, int C) {
    const long items = (long)(C / 8) * cdiv(Wo, DW_PX) * Ho;
    return cdiv(items, 256);
}

template <int KS, int S>
__global__ __launch_bounds__(256) void k_dwconv(const bf16_t* __restrict__ in, const bf16_t* __restrict__ w,
                                                const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                float* __restrict__ pool_partial, int B, int nblk, int H, int W, int C,
                                                int Ho, int Wo, int act) {
    constexpr int PAD = KS / 2;
    constexpr int IW = (DW_PX - 1) * S + KS;  // input columns touched by one thread
    __shared__ float red[256][8];
    const int CG = C >> 3;
    const int strips = (Wo + DW_PX - 1) / DW_PX;
    // XCD-aware placement: workgroups are dealt round-robin over the 8 XCDs (private 4 MB L2 each), so with a
    // plain (chunk, image) grid the output rows that share input rows land on different L2s and every input row
    // is fetched from HBM ~2.5x (PMC: 8.2 GB fetched for 4.2 GB algorithmic).  Here XCD x works through images
    // x, x+8, ... one whole image at a time, so an image's input is fetched into ONE L2 once.
    const int xcd = blockIdx.x & 7;
    const int seq = blockIdx.x >> 3;
    const int b = xcd + 8 * (seq / nblk);
    const int blk = seq - (seq / nblk) * nblk;
    if (b >= B) return;
    const long item = (long)blk * 256 + threadIdx.x;
    const long nitems = (long)CG * strips * Ho;
    const bool live = item < nitems;
    const int cg = (int)(item % CG);
    const long rest = item / CG;
    const int sx = (int)(rest % strips);
    const int oy = (int)(rest / strips);
    const int ox0 = sx * DW_PX;
    const int c0 = cg * 8;

    float acc[DW_PX][8];
    float psum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) psum[j] = 0.f;

    if (live) {
        float bs[8];
        {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + c0);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias + c0 + 4);
            bs[0] = b0.x; bs[1] = b0.y; bs[2] = b0.z; bs[3] = b0.w; bs[4] = b1.x; bs[5] = b1.y; bs[6] = b1.z; bs[7] = b1.w;
        }
#pragma unroll
        for (int p = 0; p < DW_PX; ++p)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[p][j] = bs[j];

        const bf16_t* inb = in + (size_t)b * H * W * C + c0;
        // NOTE (measured, profiles/r01_effnet_per_op_*.txt): a branch-free "interior" fast path and a rolled ky
        // loop were both SLOWER here (0.25 -> 0.31 ms on the 56x56 C192 layer): this kernel is bound by load
        // latency, and the fully unrolled form lets the compiler issue every tap's load up front.
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            const int iy = oy * S - PAD + ky;
            if (iy < 0 || iy >= H) continue;
            float wk[KS][8];
#pragma unroll
            for (int kx = 0; kx < KS; ++kx)
                unpack8(*reinterpret_cast<const u32x4*>(w + (size_t)(ky * KS + kx) * C + c0), wk[kx]);
            const bf16_t* row = inb + (size_t)iy * W * C;
#pragma unroll
            for (int i = 0; i < IW; ++i) {
                const int ix = ox0 * S - PAD + i;
                if (ix < 0 || ix >= W) continue;
                float v[8];
                unpack8(*reinterpret_cast<const u32x4*>(row + (size_t)ix * C), v);
#pragma unroll
                for (int p = 0; p < DW_PX; ++p) {
                    const int kx = i - p * S;  // compile-time after unrolling
                    if (kx >= 0 && kx < KS) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[p][j] += wk[kx][j] * v[j];
                    }
                }
            }
        }
        bf16_t* o = out + (((size_t)b * Ho + oy) * Wo + ox0) * C + c0;
        MI355_ACT_DISPATCH(act, {
_Pragma("unroll")
            for (int p = 0; p < DW_PX; ++p)
_Pragma("unroll")
                for (int j = 0; j < 8; ++j) acc[p][j] = act_c<ACT>(acc[p][j]);
        })
#pragma unroll
        for (int p = 0; p < DW_PX; ++p) {
            if (ox0 + p < Wo) {
#pragma unroll
                for (int j = 0; j < 8; ++j) psum[j] += acc[p][j];
                // streaming store: the output is not re-read by this kernel, keep the L2 for the input rows that the
                // neighbouring output rows are about to re-read
                // (narrow layers: a pixel's channels are < 128 bytes, so a store instruction writes partial lines; streamed
                //  past the L2 they cost 1.6-2.1x their bytes at HBM (PMC WRITE_SIZE) - let the L2 merge them instead)
                if (C >= 64) __builtin_nontemporal_store(pack8(acc[p]), reinterpret_cast<u32x4*>(o + (size_t)p * C));
                else *reinterpret_cast<u32x4*>(o + (size_t)p * C) = pack8(acc[p]);
            }
        }
    }

    if (pool_partial != nullptr) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = psum[j];
        __syncthreads();
        // thread t < min(CG,256) owns the channel group (first_cg + t) % CG of this block
        const int ncg = CG < 256 ? CG : 256;
        if ((int)threadIdx.x < ncg) {
            float s[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] = 0.f;
            for (int u = threadIdx.x; u < 256; u += CG) {   // same cg: every CG-th thread, in thread order
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] += red[u][j];
            }
            const int my_cg = (int)(((long)blk * 256 + threadIdx.x) % CG);
            float* pp = pool_partial + ((size_t)b * nblk + blk) * C + my_cg * 8;
            *reinterpret_cast<f32x4*>(pp) = (f32x4){s[0], s[1], s[2], s[3]};
            *reinterpret_cast<f32x4*>(pp + 4) = (f32x4){s[4], s[5], s[6], s[7]};
        }
        // channel groups this block did not touch (CG > 256) must read as zero
        if (CG > 256) {
            const int first = (int)(((long)blk * 256) % CG);
            for (int t = 256 + threadIdx.x; t < CG; t += 256) {
                const int z = (first + t) % CG;
                float* pp = pool_partial + ((size_t)b * nblk + blk) * C + z * 8;
                *reinterpret_cast<f32x4*>(pp) = (f32x4){0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(pp + 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    }
}

// -------------------------------------------------------------------------------------
// Stride-1 depthwise through an LDS tile (the early 28x28 .. 112x112 layers).  A workgroup owns TH output rows x the
// full width x a 64-channel slab: the (TH + K - 1) x (W + K - 1) input tile is fetched ONCE with coalesced 16-byte
// loads that are all in flight together (the direct kernel re-reads every input 4.5x (k3) / 10x (k5) through L1
// and spends 74 % of its wave-cycles waiting), borders are zero-filled in LDS so the tap loops have no bounds
// checks, and the per-workgroup footprint stays under 55 KB (3 workgroups per CU).
// SE squeeze: one partial per (image, row band): pool_partial[b][band][C].
// -------------------------------------------------------------------------------------
template <int KS, int PX>
__global__ __launch_bounds__(256) void k_dw_tiled(const bf16_t* __restrict__ in, const bf16_t* __restrict__ w,
                                                  const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                  float* __restrict__ pool_partial, int H, int W, int C, int TH, int CGC,
                                                  int nbands, int act) {
    constexpr int PAD = KS / 2;
    constexpr int IW = PX - 1 + KS;
    extern __shared__ __attribute__((aligned(16))) bf16_t tile[];   // [IH][EW][CGC*8]
    const int b = blockIdx.y;
    const int band = blockIdx.x % nbands, chunk = blockIdx.x / nbands;
    const int oy0 = band * TH;
    const int th = min(TH, H - oy0);
    const int IH = TH + KS - 1, EW = W + 2 * PAD + (PX - 1);   // slack columns for a partial last strip
    const int cg0 = chunk * CGC;
    const int CG = C >> 3;
    const int ncg = min(CGC, CG - cg0);
    const int PS = CGC * 8;                                     // pixel stride in elements
    const bf16_t* inb = in + (size_t)b * H * W * C;
    for (int id = threadIdx.x; id < IH * EW * CGC; id += 256) {
        const int cg = id % CGC;
        const int px = id / CGC;
        const int r = px / EW, x = px - r * EW;
        const int iy = oy0 - PAD + r, ix = x - PAD;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (cg < ncg && iy >= 0 && iy < H && ix >= 0 && ix < W)
            v = *reinterpret_cast<const u32x4*>(inb + ((size_t)iy * W + ix) * C + (cg0 + cg) * 8);
        *reinterpret_cast<u32x4*>(&tile[(size_t)px * PS + cg * 8]) = v;
    }
    __syncthreads();

    const int strips = (W + PX - 1) / PX;
    const int nitems = CGC * strips * th;
    float psum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) psum[j] = 0.f;
    for (int item = threadIdx.x; item < nitems; item += 256) {
        const int cg = item % CGC;
        const int rest = item / CGC;
        const int sx = rest % strips, oyl = rest / strips;
        if (cg >= ncg) continue;
        const int ox0 = sx * PX;
        const int c0 = (cg0 + cg) * 8;
        float acc[PX][8];
        {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + c0);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias + c0 + 4);
#pragma unroll
            for (int p = 0; p < PX; ++p) {
                acc[p][0] = b0.x; acc[p][1] = b0.y; acc[p][2] = b0.z; acc[p][3] = b0.w;
                acc[p][4] = b1.x; acc[p][5] = b1.y; acc[p][6] = b1.z; acc[p][7] = b1.w;
            }
        }
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            float wk[KS][8];
#pragma unroll
            for (int kx = 0; kx < KS; ++kx)
                unpack8(*reinterpret_cast<const u32x4*>(w + (size_t)(ky * KS + kx) * C + c0), wk[kx]);
            const bf16_t* row = tile + ((size_t)(oyl + ky) * EW + ox0) * PS + cg * 8;
#pragma unroll
            for (int i = 0; i < IW; ++i) {
                float v[8];
                unpack8(*reinterpret_cast<const u32x4*>(row + (size_t)i * PS), v);
#pragma unroll
                for (int p = 0; p < PX; ++p) {
                    const int kx = i - p;
                    if (kx >= 0 && kx < KS) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[p][j] += wk[kx][j] * v[j];
                    }
                }
            }
        }
        MI355_ACT_DISPATCH(act, {
_Pragma("unroll")
            for (int p = 0; p < PX; ++p)
_Pragma("unroll")
                for (int j = 0; j < 8; ++j) acc[p][j] = act_c<ACT>(acc[p][j]);
        })
        bf16_t* o = out + (((size_t)b * H + oy0 + oyl) * W + ox0) * C + c0;
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            if (ox0 + p < W) {
#pragma unroll
                for (int j = 0; j < 8; ++j) psum[j] += acc[p][j];
                *reinterpret_cast<u32x4*>(o + (size_t)p * C) = pack8(acc[p]);
            }
        }
    }
    if (pool_partial != nullptr) {
        __syncthreads();                       // the tile is dead: reuse it for the fixed-order reduction
        float* red = reinterpret_cast<float*>(tile);
#pragma unroll
        for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = psum[j];
        __syncthreads();
        // a thread's items all share cg = tid % CGC only when 256 % CGC == 0; otherwise cg = (tid + 256*i) % CGC
        // varies per item, so psum was accumulated per item-cg only if nitems <= 256 (one item per thread).
        if ((int)threadIdx.x < ncg) {
            float sacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int u = threadIdx.x; u < 256; u += CGC) {
#pragma unroll
                for (int j = 0; j < 8; ++j) sacc[j] += red[u * 8 + j];
            }
            float* pp = pool_partial + ((size_t)b * nbands + band) * C + (cg0 + threadIdx.x) * 8;
            *reinterpret_cast<f32x4*>(pp) = (f32x4){sacc[0], sacc[1], sacc[2], sacc[3]};
            *reinterpret_cast<f32x4*>(pp + 4) = (f32x4){sacc[4], sacc[5], sacc[6], sacc[7]};
        }
    }
}

// Geometry of the tiled kernel for one layer; returns false when the direct kernel should be used.
static bool dw_tiled_plan(int H, int W, int C, int k, int stride, int* TH, int* CGC, int* PX, size_t* lds) {
    if (stride != 1 || W < 14 || C % 8) return false;
    const int CG = C / 8;
    const int cgc = CG < 8 ? CG : 8;
    const int px = W % 7 == 0 ? 7 : 4;
    const int strips = (W + px - 1) / px;
    // one item per thread (the squeeze reduction relies on it): CGC * strips * TH <= 256
    int th = 256 / (cgc * strips);
    if (th < 1) return false;
    if (th > H) th = H;
    const int pad = k / 2;
    const size_t bytes = (size_t)(th + k - 1) * (W + 2 * pad + px - 1) * cgc * 16;
    if (bytes > 64 * 1024 || bytes < 256 * 8 * 4) return false;
    *TH = th; *CGC = cgc; *PX = px; *lds = bytes;
    return true;
}

template <int KS, int PX>
static int launch_dw_tiled(const bf16_t* in, const bf16_t* w, const float* bias, bf16_t* out, float* pool_partial, int B,
                           int H, int W, int C, int TH, int CGC, size_t lds, int act, int* pool_nblk, hipStream_t st) {
    static bool attr_done[MI355_MAX_DEVICES] = {false};   // per device
    if (first_time_on_this_device(attr_done)) {
        MI355_CHECK_HIP(hipFuncSetAttribute((const void*)k_dw_tiled<KS, PX>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            64 * 1024));
    }
    const int nbands = cdiv(H, TH), nchunks = cdiv(C / 8, CGC);
    hipLaunchKernelGGL((k_dw_tiled<KS, PX>), dim3(nbands * nchunks, B), dim3(256), lds, st, in, w, bias, out, pool_partial,
                       H, W, C, TH, CGC, nbands, act);
    MI355_LAUNCH_CHECK();
    if (pool_nblk) *pool_nblk = nbands;
    return OK;
}

// =====================================================================================
// 3x3 stride-1 depthwise for NARROW layers (C <= 64: the 112x112 C40 / C24 maps of EfficientNet's first stage, RexNet's
// C32) on the matrix pipe.  The direct kernel above spends 9 FMAs + unpacking per output on the VALU next to the SiLU
// and ran at 2.3 TB/s (VALU 56 % busy, waves waiting 73 %).  Here one 16x16x32 MFMA computes one ROW of taps for 8
// channels x 32 consecutive pixels: M = (pixel parity p, channel c), N = 16 pixel pairs, K = (t' = p + tap, channel) with
// A[(p,c)][(t',c')] = [c == c'] * w[ky][t' - p][c] and B[(t',c)][n] = in[c][2n + t' - 1] - each lane loads the 8 channels
// (16 B) of ONE input pixel straight from global as its B fragment, no LDS, three MFMAs per 8 x 32 outputs.
// A wave owns one 8-channel unit (its A fragments are built once), the C/8 waves of a workgroup walk the same pixel
// tiles together (they hit the same lines in L1), ~50 VGPRs so 8 waves per SIMD hide the load latency.  Rows above /
// below the image fall outside the image's buffer resource and read as zero (hardware range check); the horizontal
// wrap of the linear pixel index is masked per lane.  Squeeze sums: one partial per (workgroup, channel), fixed order.
// =====================================================================================
typedef __bf16 dw_bf16x8 __attribute__((ext_vector_type(8)));
// Wave-private LDS staging: loading the B fragments straight from global moves 16 B per lane at an 80-byte stride (every
// instruction touches ~22 cache lines for 1 KB; that version ran at 1.3-2.5 TB/s and was removed).  Here a wave copies the three 34-pixel row windows of its tile with
// contiguous 16-byte loads (8 per lane instead of 15 scattered ones), reads the B fragments from LDS, and writes its 32
// output pixels through LDS as one contiguous 32*C*2-byte run.  No barriers: a wave only touches its own LDS region, and
// the LDS executes one wave's instructions in order.  The next tile's loads are requested before this tile's MFMAs.
template <int NU>
__global__ __launch_bounds__(256, NU >= 6 ? 2 : (NU >= 4 ? 3 : 4)) void k_dw3_lds(
    const bf16_t* __restrict__ in, const bf16_t* __restrict__ w, const float* __restrict__ bias, bf16_t* __restrict__ out,
    float* __restrict__ pool_partial, int B, int nblk, int H, int W, int act, unsigned magicW, int tiles_per_wg) {
    constexpr int C = NU * 8, PB = C * 2;                 // channels (the whole layer), bytes per pixel
    constexpr int ROWB = 34 * PB;                         // one row window: pixels q0-1 .. q0+32
    constexpr int NCH = 3 * ROWB / 16;                    // 16-byte chunks of the three windows
    constexpr int NLD = (NCH + 63) / 64;                  // staging loads per lane
    constexpr int OUTB = 32 * PB, NOC = OUTB / 16, NST = (NOC + 63) / 64;
    constexpr int WS = 3 * ROWB + OUTB + NU * 16 + NU * 32;   // LDS bytes per wave: windows | output tile | zeros | bias
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * WS];
    __shared__ float red[4][NU * 8];
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int b = xcd + 8 * (seq / nblk), blk = seq - (seq / nblk) * nblk;
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int HW = H * W;
    unsigned char* Lw = lds + wave * WS;
    unsigned char* Lo = Lw + 3 * ROWB;
    unsigned char* Lz = Lo + OUTB;
    float* Lb = reinterpret_cast<float*>(Lz + NU * 16);      // this wave's copy of the bias (an LDS read per unit is cheaper than 4*NU registers)
    if (lane < NU) *reinterpret_cast<u32x4*>(Lz + lane * 16) = (u32x4){0u, 0u, 0u, 0u};
    if (lane < NU * 8) Lb[lane] = bias[lane];
    u32x4 dwf[NU][3];
    {
        const int pm = fr >> 3, c = fr & 7, t = fq - pm;
        const int q = c >> 1;
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                unsigned v = 0u;
                if (t >= 0 && t < 3) v = w[(size_t)(ky * 3 + t) * C + u * 8 + c];
                const unsigned word = (c & 1) ? (v << 16) : v;
                dwf[u][ky] = (u32x4){q == 0 ? word : 0u, q == 1 ? word : 0u, q == 2 ? word : 0u, q == 3 ? word : 0u};
            }
    }
    const int ch4 = (fq & 1) * 4, pp = fq >> 1;
    const bf16_t* img = in + (size_t)b * HW * C;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(img), 0, HW * PB, 0x00020000);
    unsigned char* ob = reinterpret_cast<unsigned char*>(out + (size_t)b * HW * C);
    const int rowb = W * PB;
    const int ntiles = (HW + 31) >> 5;
    const int t_end = min(ntiles, (blk + 1) * tiles_per_wg);
    float psum[NU][4];
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) psum[u][j] = 0.f;
    // staging chunk i of this lane: chunk id = lane + 64 i -> row window id / ROWB-relative byte
    int s_goff[NLD], s_loff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int ch = lane + 64 * i;
        const int ky = ch / (ROWB / 16), rem = ch - ky * (ROWB / 16);
        s_loff[i] = ch < NCH ? ky * ROWB + rem * 16 : -1;
        s_goff[i] = ch < NCH ? (ky - 1) * rowb + rem * 16 - PB : 0x40000000;     // relative to the tile's first pixel
    }
    u32x4 sv[NLD];
    auto stage_load = [&](int tile) {
        const int base = tile * 32 * PB;
#pragma unroll
        for (int i = 0; i < NLD; ++i) sv[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + s_goff[i], 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    const int t_begin = blk * tiles_per_wg + wave;
    if (t_begin < t_end) stage_load(t_begin);
    for (int tile = t_begin; tile < t_end; tile += 4) {
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            if (s_loff[i] >= 0) *reinterpret_cast<u32x4*>(Lw + s_loff[i]) = sv[i];
        if (tile + 4 < t_end) stage_load(tile + 4);
        const int q = tile * 32 + 2 * fr;
        const int y = (int)__umulhi((unsigned)q, magicW);
        const int x = q - y * W;
        const bool colok = !(fq == 0 && x == 0) && !(fq == 3 && x == W - 2);
        const unsigned char* rb = Lw + (2 * fr + fq) * PB;
        const unsigned char* r0 = colok ? rb : Lz;
        const unsigned char* r1 = colok ? rb + ROWB : Lz;
        const unsigned char* r2 = colok ? rb + 2 * ROWB : Lz;
        const bool live = q + pp < HW;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const u32x4 x0 = *reinterpret_cast<const u32x4*>(r0 + u * 16);
            const u32x4 x1 = *reinterpret_cast<const u32x4*>(r1 + u * 16);
            const u32x4 x2 = *reinterpret_cast<const u32x4*>(r2 + u * 16);
            f32x4 acc = *reinterpret_cast<const f32x4*>(Lb + u * 8 + ch4);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const dw_bf16x8*>(&dwf[u][0]), *reinterpret_cast<const dw_bf16x8*>(&x0), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const dw_bf16x8*>(&dwf[u][1]), *reinterpret_cast<const dw_bf16x8*>(&x1), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const dw_bf16x8*>(&dwf[u][2]), *reinterpret_cast<const dw_bf16x8*>(&x2), acc, 0, 0, 0);
            MI355_ACT_DISPATCH(act, {
                acc.x = act_c<ACT>(acc.x); acc.y = act_c<ACT>(acc.y); acc.z = act_c<ACT>(acc.z); acc.w = act_c<ACT>(acc.w);
            })
            if (live) { psum[u][0] += acc.x; psum[u][1] += acc.y; psum[u][2] += acc.z; psum[u][3] += acc.w; }
            u32x2 o;
            o.x = pack2bf(acc.x, acc.y);
            o.y = pack2bf(acc.z, acc.w);
            *reinterpret_cast<u32x2*>(Lo + (2 * fr + pp) * PB + u * 16 + ch4 * 2) = o;
        }
        // the tile's 32 output pixels are one contiguous run in memory
        const int nvalid = min(32, HW - tile * 32) * PB;
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int ob_off = (lane + 64 * i) * 16;
            if (ob_off < nvalid && ob_off < OUTB)
                *reinterpret_cast<u32x4*>(ob + (size_t)tile * 32 * PB + ob_off) = *reinterpret_cast<const u32x4*>(Lo + ob_off);
        }
    }
    if (pool_partial) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) psum[u][j] += __shfl_xor(psum[u][j], o, 64);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) psum[u][j] += __shfl_xor(psum[u][j], 32, 64);
            if (fr == 0 && fq < 2)
                *reinterpret_cast<f32x4*>(&red[wave][u * 8 + ch4]) = (f32x4){psum[u][0], psum[u][1], psum[u][2], psum[u][3]};
        }
        __syncthreads();
        if (threadIdx.x < NU * 8)
            pool_partial[((size_t)b * nblk + blk) * C + threadIdx.x] =
                ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    }
}

template <int NU>
static void launch_dw3_lds(const bf16_t* in, const bf16_t* w, const float* bias, bf16_t* out, float* pool_partial, int B, int nb,
                           int H, int W, int act, unsigned magic, int tpw, hipStream_t st) {
    hipLaunchKernelGGL((k_dw3_lds<NU>), dim3((unsigned)(8 * cdiv(B, 8) * nb)), dim3(256), 0, st, in, w, bias, out, pool_partial, B,
                       nb, H, W, act, magic, tpw);
}

int launch_dwconv(const bf16_t* in, const bf16_t* w, const float* bias, bf16_t* out, float* pool_partial, int B, int H,
                  int W, int C, int k, int stride, int act, int* pool_nblk, hipStream_t st) {
    MI355_REQUIRE(C % 8 == 0, "dwconv: C=%d must be a multiple of 8", C);
    MI355_REQUIRE((k == 3 || k == 5) && (stride == 1 || stride == 2), "dwconv: unsupported k=%d stride=%d", k, stride);
    const int pad = k / 2;
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    // measured slower than the direct kernel on every EfficientNet layer (0.27 -> 0.32 ms at 56x56 C192): opt-in only
    static const int use_tiled = getenv("MI355_DW_TILED") ? atoi(getenv("MI355_DW_TILED")) : 0;
    int TH, CGC, PX;
    size_t lds;
    if (use_tiled && dw_tiled_plan(H, W, C, k, stride, &TH, &CGC, &PX, &lds)) {
        if (k == 3) return PX == 7 ? launch_dw_tiled<3, 7>(in, w, bias, out, pool_partial, B, H, W, C, TH, CGC, lds, act, pool_nblk, st)
                                   : launch_dw_tiled<3, 4>(in, w, bias, out, pool_partial, B, H, W, C, TH, CGC, lds, act, pool_nblk, st);
        return PX == 7 ? launch_dw_tiled<5, 7>(in, w, bias, out, pool_partial, B, H, W, C, TH, CGC, lds, act, pool_nblk, st)
                       : launch_dw_tiled<5, 4>(in, w, bias, out, pool_partial, B, H, W, C, TH, CGC, lds, act, pool_nblk, st);
    }
    // narrow 3x3 stride-1 layers (C <= 48): matrix-pipe kernel with wave-private LDS staging (C40 @112x112: 0.218 -> 0.126 ms);
    // MI355_DW_MFMA=0 keeps the direct kernel
    static const int use_mfma = getenv("MI355_DW_MFMA") ? atoi(getenv("MI355_DW_MFMA")) : 1;
    if (use_mfma && k == 3 && stride == 1 && C <= 48 && W % 2 == 0 && W >= 4 && (long)H * W * C * 2 < (1L << 30)) {
        const int ntiles = cdiv((long)H * W, 32);
        int nb = std::min(std::min(dw_pool_blocks(Ho, Wo, C), 14), ntiles);      // <= the squeeze-partial slot the planner sized
        const int tpw = cdiv(ntiles, nb);
        nb = cdiv(ntiles, tpw);
        if (pool_nblk) *pool_nblk = nb;
        const unsigned magic = (unsigned)((0x100000000ULL + (unsigned)W - 1) / (unsigned)W);
        switch (C / 8) {
            case 1: launch_dw3_lds<1>(in, w, bias, out, pool_partial, B, nb, H, W, act, magic, tpw, st); break;
            case 2: launch_dw3_lds<2>(in, w, bias, out, pool_partial, B, nb, H, W, act, magic, tpw, st); break;
            case 3: launch_dw3_lds<3>(in, w, bias, out, pool_partial, B, nb, H, W, act, magic, tpw, st); break;
            case 4: launch_dw3_lds<4>(in, w, bias, out, pool_partial, B, nb, H, W, act, magic, tpw, st); break;
            case 5: launch_dw3_lds<5>(in, w, bias, out, pool_partial, B, nb, H, W, act, magic, tpw, st); break;
            default: launch_dw3_lds<6>(in, w, bias, out, pool_partial, B, nb, H, W, act, magic, tpw, st); break;   // (rexnet_150's C48 @112x112)
        }
        MI355_LAUNCH_CHECK();
        return OK;
    }
    if (pool_nblk) *pool_nblk = dw_pool_blocks(Ho, Wo, C);
    const int nblk = dw_pool_blocks(Ho, Wo, C);
    dim3 grid((unsigned)(8 * cdiv(B, 8) * nblk));
#define DW_LAUNCH(KS, S)                                                                                            \
    hipLaunchKernelGGL((k_dwconv<KS, S>), grid, dim3(256), 0, st, in, w, bias, out, pool_partial, B, nblk, H, W, C, Ho, Wo, act)
    if (k == 3 && stride == 1) DW_LAUNCH(3, 1);
    else if (k == 3 && stride == 2) DW_LAUNCH(3, 2);
    else if (k == 5 && stride == 1) DW_LAUNCH(5, 1);
    else DW_LAUNCH(5, 2);
#undef DW_LAUNCH
    MI355_LAUNCH_CHECK();
    return OK;
}

// =====================================================================================
// SE gate.  One 1024-thread block (16 waves) per image: the kernel is latency-bound (tiny FLOPs, weights
// hot in L2), so what matters is enough waves in flight and coalesced weight reads.
// W1 [rd][C], W2T [rd][C] (transposed at pack time so both FCs read weights coalesced along C).
// All fp32; summation orders are fixed.
// =====================================================================================
constexpr int SE_THREADS = 1024;
constexpr int SE_MAX_C = 4096;
constexpr int SE_MAX_RD = 512;
__global__ __launch_bounds__(SE_THREADS) void k_se(const float* __restrict__ pool_partial, int nblk, float inv_hw,
                                                   const float* __restrict__ w1, const float* __restrict__ b1,
                                                   const float* __restrict__ w2t, const float* __restrict__ b2,
                                                   float* __restrict__ gate, int C, int rd, int act1) {
    __shared__ float s[SE_MAX_C];
    __shared__ float r[SE_MAX_RD];
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* pp = pool_partial + (size_t)b * nblk * C;
    for (int c = tid; c < C; c += SE_THREADS) {
        float a = 0.f;
#pragma unroll 8
        for (int k = 0; k < nblk; ++k) a += pp[(size_t)k * C + c];   // fixed order
        s[c] = a * inv_hw;
    }
    __syncthreads();
    // every image's workgroup reads the same FC weights: rotate the starting row / column by the image index so the
    // co-resident workgroups do not hit the same L2 lines in lock-step (outputs are independent: order is free)
    for (int jj = wave; jj < rd; jj += SE_THREADS / 64) {
        const int j = (jj + b) % rd;
        const float* wr = w1 + (size_t)j * C;
        float a = 0.f;
#pragma unroll 4
        for (int c = lane; c < C; c += 64) a += wr[c] * s[c];
        a = wave_sum(a);
        if (lane == 0) r[j] = apply_act(a + b1[j], act1);
    }
    __syncthreads();
    const int rot = (b * 192) % C;
    for (int cc = tid; cc < C; cc += SE_THREADS) {
        int c = cc + rot;
        if (c >= C) c -= C;
        float a = b2[c];
#pragma unroll 8
        for (int j = 0; j < rd; ++j) a += w2t[(size_t)j * C + c] * r[j];
        gate[(size_t)b * C + c] = sigmoid_f(a);
    }
}

// Small layers (rd <= 16, C <= 1024: the early EfficientNet blocks): every load of a phase in one batch.  For the wide RexNet
// layers (rd up to 173) the batched form was 30-50 % slower than the plain loops above, so it is selected by shape.
__global__ __launch_bounds__(SE_THREADS) void k_se_small(const float* __restrict__ pool_partial, int nblk, float inv_hw,
                                                   const float* __restrict__ w1, const float* __restrict__ b1,
                                                   const float* __restrict__ w2t, const float* __restrict__ b2,
                                                   float* __restrict__ gate, int C, int rd, int act1) {
    __shared__ float s[SE_MAX_C];
    __shared__ float r[SE_MAX_RD];
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* pp = pool_partial + (size_t)b * nblk * C;
    for (int c = tid; c < C; c += SE_THREADS) {
        float a = 0.f;
#pragma unroll 8
        for (int k = 0; k < nblk; ++k) a += pp[(size_t)k * C + c];   // fixed order
        s[c] = a * inv_hw;
    }
    __syncthreads();
    // FC2's weights do not depend on FC1's result: the first batch (16 hidden units of this thread's channel) is requested
    // now and travels during FC1.  (Both FC loops used `#pragma unroll N` over runtime trip counts: for rd = 6 .. 12 and C <=
    // 288 every iteration ran in the one-at-a-time remainder loop, a dependent L2 round trip each - 8 us per launch.)
    constexpr int NBJ = 16;
    const int rot = (b * 192) % C;
    int c2 = (int)tid + rot;
    if (c2 >= C) c2 -= C;
    const bool has_c = (int)tid < C;
    float w2v[NBJ];
    float b2v = 0.f;
    if (has_c) {
        b2v = b2[c2];
#pragma unroll
        for (int t = 0; t < NBJ; ++t) w2v[t] = w2t[(size_t)min(t, rd - 1) * C + c2];
    }
    __builtin_amdgcn_sched_barrier(0);
    // every image's workgroup reads the same FC weights: rotate the starting row / column by the image index so the
    // co-resident workgroups do not hit the same L2 lines in lock-step (outputs are independent: order is free)
    for (int jj = wave; jj < rd; jj += SE_THREADS / 64) {
        const int j = (jj + b) % rd;
        const float* wr = w1 + (size_t)j * C;
        const float bj = b1[j];
        float a = 0.f;
        constexpr int NBC = 8;                                  // 8 x 64 channels per batch, all loads first
        for (int cb = 0; cb < C; cb += 64 * NBC) {
            float wv[NBC];
#pragma unroll
            for (int t = 0; t < NBC; ++t) wv[t] = wr[min(cb + t * 64 + lane, C - 1)];
#pragma unroll
            for (int t = 0; t < NBC; ++t) {
                const int c = cb + t * 64 + lane;
                a += c < C ? wv[t] * s[c] : 0.f;               // ascending channel order per lane, as before
            }
        }
        a = wave_sum(a);
        if (lane == 0) r[j] = apply_act(a + bj, act1);
    }
    __syncthreads();
    // thread tid owns channel c2 (and tid + 1024, ... for wide layers: those take the plain loop)
    if (has_c) {
        float a = b2v;
#pragma unroll
        for (int t = 0; t < NBJ; ++t) a += t < rd ? w2v[t] * r[t] : 0.f;
        for (int j0 = NBJ; j0 < rd; j0 += NBJ) {
            float wv[NBJ];
#pragma unroll
            for (int t = 0; t < NBJ; ++t) wv[t] = w2t[(size_t)min(j0 + t, rd - 1) * C + c2];
#pragma unroll
            for (int t = 0; t < NBJ; ++t) a += j0 + t < rd ? wv[t] * r[j0 + t] : 0.f;
        }
        gate[(size_t)b * C + c2] = sigmoid_f(a);
    }
    for (int cc = tid + SE_THREADS; cc < C; cc += SE_THREADS) {
        int c = cc + rot;
        if (c >= C) c -= C;
        float a = b2[c];
        for (int j0 = 0; j0 < rd; j0 += NBJ) {
            float wv[NBJ];
#pragma unroll
            for (int t = 0; t < NBJ; ++t) wv[t] = w2t[(size_t)min(j0 + t, rd - 1) * C + c];
#pragma unroll
            for (int t = 0; t < NBJ; ++t) a += j0 + t < rd ? wv[t] * r[j0 + t] : 0.f;
        }
        gate[(size_t)b * C + c] = sigmoid_f(a);
    }
}

int launch_se(const float* pool_partial, int nblk, float inv_hw, const float* w1, const float* b1, const float* w2t,
              const float* b2, float* gate, int B, int C, int rd, int act1, hipStream_t st) {
    MI355_REQUIRE(C <= SE_MAX_C && rd <= SE_MAX_RD, "se: C=%d rd=%d exceed limits", C, rd);
    if (rd <= 16 && C <= SE_THREADS)
        hipLaunchKernelGGL(k_se_small, dim3(B), dim3(SE_THREADS), 0, st, pool_partial, nblk, inv_hw, w1, b1, w2t, b2, gate, C, rd, act1);
    else
        hipLaunchKernelGGL(k_se, dim3(B), dim3(SE_THREADS), 0, st, pool_partial, nblk, inv_hw, w1, b1, w2t, b2, gate, C, rd, act1);
    MI355_LAUNCH_CHECK();
    return OK;
}

// =====================================================================================
// global average pool, layout conversion, conv_input
// =====================================================================================
__global__ __launch_bounds__(256) void k_gap(const bf16_t* __restrict__ in, float* __restrict__ pooled,
                                             bf16_t* __restrict__ pooled_bf16, int HW, int C, long total) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int CG = C >> 3;
    const int cg = (int)(t % CG);
    const long b = t / CG;
    const bf16_t* p = in + (size_t)b * HW * C + cg * 8;
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < HW; ++i) {
        float v[8];
        unpack8(*reinterpret_cast<const u32x4*>(p + (size_t)i * C), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += v[j];
    }
    const float inv = 1.0f / (float)HW;
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] *= inv;
    float* o = pooled + (size_t)b * C + cg * 8;
    *reinterpret_cast<f32x4*>(o) = (f32x4){a[0], a[1], a[2], a[3]};
    *reinterpret_cast<f32x4*>(o + 4) = (f32x4){a[4], a[5], a[6], a[7]};
    if (pooled_bf16) *reinterpret_cast<u32x4*>(pooled_bf16 + (size_t)b * C + cg * 8) = pack8(a);
}

int launch_gap(const bf16_t* in, float* pooled, bf16_t* pooled_bf16, int B, int HW, int C, hipStream_t st) {
    MI355_REQUIRE(C % 8 == 0, "gap: C=%d must be a multiple of 8", C);
    const long total = (long)B * (C / 8);
    hipLaunchKernelGGL(k_gap, dim3(cdiv(total, 256)), dim3(256), 0, st, in, pooled, pooled_bf16, HW, C, total);
    MI355_LAUNCH_CHECK();
    return OK;
}

// timm ClassifierHead on the un-pooled map (train/train.py:194-195: fm = forward_features(x); lbl = model.head(fm)):
// fm [B][C][HW] fp32 NCHW -> global average pool (fp32, same summation order as k_gap) -> bf16 -> Linear with bf16
// weights and fp32 accumulation (the rounding points of the in-model classifier GEMM).  One workgroup per image.
__global__ __launch_bounds__(256) void k_pool_linear(const float* __restrict__ fm, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ out,
                                                     float* __restrict__ pooled_out, int C, int HW, int N) {
    extern __shared__ float pl[];   // [C] pooled, bf16-rounded values
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* f = fm + (size_t)b * C * HW;
    const float inv = 1.0f / (float)HW;
    for (int c = tid; c < C; c += 256) {
        float a = 0.f;
        for (int i = 0; i < HW; ++i) a += f[(size_t)c * HW + i];
        a *= inv;
        if (pooled_out) pooled_out[(size_t)b * C + c] = a;
        pl[c] = bf2f(f2bf(a));
    }
    __syncthreads();
    if (w == nullptr) return;
    for (int n = wave; n < N; n += 4) {
        const float* wr = w + (size_t)n * C;
        float a = 0.f;
        for (int c = lane; c < C; c += 64) a += bf2f(f2bf(wr[c])) * pl[c];
        a = wave_sum(a);
        if (lane == 0) out[(size_t)b * N + n] = a + (bias ? bias[n] : 0.f);
    }
}

int launch_pool_linear(const float* fm, const float* w, const float* bias, float* out, float* pooled_out, int B, int C,
                       int HW, int N, hipStream_t st) {
    MI355_REQUIRE((size_t)C * 4 <= 64 * 1024, "pool_linear: C=%d too large", C);
    hipLaunchKernelGGL(k_pool_linear, dim3(B), dim3(256), (size_t)C * 4, st, fm, w, bias, out, pooled_out, C, HW, N);
    MI355_LAUNCH_CHECK();
    return OK;
}

// out[b][c][p] = in[b][p][c]; 32x32 tile transpose through LDS so both sides are coalesced.
__global__ __launch_bounds__(256) void k_nhwc_to_nchw(const bf16_t* __restrict__ in, float* __restrict__ out, int HW,
                                                      int C, int Cvalid) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int p = p0 + i, c = c0 + tx;
        tile[i][tx] = (p < HW && c < C) ? bf2f(in[((size_t)b * HW + p) * C + c]) : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, p = p0 + tx;
        if (c < Cvalid && p < HW) out[((size_t)b * Cvalid + c) * HW + p] = tile[tx][i];
    }
}

int launch_nhwc_to_nchw_f32(const bf16_t* in, float* out, int B, int HW, int C, int Cvalid, hipStream_t st) {
    dim3 grid(cdiv(HW, 32), cdiv(C, 32), B);
    hipLaunchKernelGGL(k_nhwc_to_nchw, grid, dim3(256), 0, st, in, out, HW, C, Cvalid);
    MI355_LAUNCH_CHECK();
    return OK;
}

// out[b][p][c] = bf16(in[b][c][p]) (zeros for c >= Cvalid); same 32x32 LDS transpose.  Parity tool (run_between_taps).
__global__ __launch_bounds__(256) void k_nchw_to_nhwc(const float* __restrict__ in, bf16_t* __restrict__ out, int HW,
                                                      int Cvalid, int C) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, p = p0 + tx;
        tile[i][tx] = (c < Cvalid && p < HW) ? in[((size_t)b * Cvalid + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int p = p0 + i, c = c0 + tx;
        if (p < HW && c < C) out[((size_t)b * HW + p) * C + c] = f2bf(tile[tx][i]);
    }
}

int launch_nchw_f32_to_nhwc_bf16(const float* in, bf16_t* out, int B, int HW, int Cvalid, int C, hipStream_t st) {
    dim3 grid(cdiv(HW, 32), cdiv(C, 32), B);
    hipLaunchKernelGGL(k_nchw_to_nhwc, grid, dim3(256), 0, st, in, out, HW, Cvalid, C);
    MI355_LAUNCH_CHECK();
    return OK;
}

__global__ __launch_bounds__(256) void k_conv_input_silu(const float* __restrict__ x, const float* __restrict__ w,
                                                         float* __restrict__ out, int H, int W) {
    __shared__ float sw[81];
    if (threadIdx.x < 81) sw[threadIdx.x] = w[threadIdx.x];
    __syncthreads();
    const int b = blockIdx.z;
    const int ox = blockIdx.x * 32 + (threadIdx.x & 31);
    const int oy = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (ox >= W || oy >= H) return;
    const float* xb = x + (size_t)b * 3 * H * W;
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy - 1 + ky;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox - 1 + kx;
                const float v = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? xb[((size_t)ci * H + iy) * W + ix] : 0.f;
#pragma unroll
                for (int co = 0; co < 3; ++co) acc[co] += v * sw[((co * 3 + ci) * 3 + ky) * 3 + kx];
            }
        }
    float* ob = out + (size_t)b * 3 * H * W;
#pragma unroll
    for (int co = 0; co < 3; ++co) ob[((size_t)co * H + oy) * W + ox] = silu_f(acc[co]);
}

// SquarePad(fill) -> ToTensor (/255) -> Normalize((x - mean) / std): uint8 HWC in, fp32 CHW out (S = max(h, w)).
// utils/square_pad.py:20-36 + inference/inference.py:48-52.  One rounding per fp32 op, like torch on the CPU.
__global__ __launch_bounds__(256) void k_square_pad_normalize(const unsigned char* __restrict__ img, int h, int w, int S,
                                                             int hp, int vp, int fill, float m0, float m1, float m2,
                                                             float s0, float s1, float s2, float* __restrict__ out) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= S || y >= S) return;
    const int iy = y - vp, ix = x - hp;
    int p0 = fill, p1 = fill, p2 = fill;
    if (iy >= 0 && iy < h && ix >= 0 && ix < w) {
        const unsigned char* p = img + ((size_t)iy * w + ix) * 3;
        p0 = p[0]; p1 = p[1]; p2 = p[2];
    }
    const size_t plane = (size_t)S * S, o = (size_t)y * S + x;
    out[o] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)p0, 255.0f), m0), s0);
    out[plane + o] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)p1, 255.0f), m1), s1);
    out[2 * plane + o] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)p2, 255.0f), m2), s2);
}

int launch_square_pad_normalize(const unsigned char* img, int h, int w, int fill, const float* mean, const float* stdv,
                                float* out, hipStream_t st) {
    const int S = h > w ? h : w;
    const int hp = (S - w) / 2, vp = (S - h) / 2;
    hipLaunchKernelGGL(k_square_pad_normalize, dim3(cdiv(S, 64), cdiv(S, 4)), dim3(256), 0, st, img, h, w, S, hp, vp, fill,
                       mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2], out);
    MI355_LAUNCH_CHECK();
    return OK;
}

int launch_conv_input_silu(const float* x, const float* w, int B, int H, int W, float* out, hipStream_t st) {
    dim3 grid(cdiv(W, 32), cdiv(H, 8), B);
    hipLaunchKernelGGL(k_conv_input_silu, grid, dim3(256), 0, st, x, w, out, H, W);
    MI355_LAUNCH_CHECK();
    return OK;
}

}  // namespace mi355
