// 1x1 convolution / linear layer as a bf16 MFMA GEMM with fused prologue (SE gate, ReLU6 on load)
// and epilogue (folded-BN bias, activation, residual add).  gfx950 only.
//
// These GEMMs are skinny (K, N <= 2304, M = B*H*W up to 3.2 M rows) and bandwidth-bound: the job of
// the kernel is to stream A once, keep W hot in L2, and write the output once.  MFMA (16x16x32 bf16)
// is used because a 1x1 conv IS a dense contraction; operands are swapped
// (D = W_tile x A_tile^T) so each lane ends up with 4 consecutive output channels of one pixel.
//
// Block = 4 waves, tile 128 (pixels) x BN = 16*NT (channels), BK = 32, double-buffered LDS with
// register-staged prefetch (issue tile t+1 loads, compute tile t, write tile t+1, one barrier).
#include "ops.h"

#include <stdlib.h>

namespace mi355 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float lo_bf(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi_bf(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

__device__ __forceinline__ u32x4 gate_chunk(u32x4 v, const float* __restrict__ gp, int relu6) {
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp);
    const f32x4 g1 = *reinterpret_cast<const f32x4*>(gp + 4);
    float f[8] = {lo_bf(v.x) * g0.x, hi_bf(v.x) * g0.y, lo_bf(v.y) * g0.z, hi_bf(v.y) * g0.w,
                  lo_bf(v.z) * g1.x, hi_bf(v.z) * g1.y, lo_bf(v.w) * g1.z, hi_bf(v.w) * g1.w};
    if (relu6) {
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = fminf(fmaxf(f[i], 0.f), 6.f);
    }
    u32x4 o;
    o.x = pack2bf(f[0], f[1]); o.y = pack2bf(f[2], f[3]); o.z = pack2bf(f[4], f[5]); o.w = pack2bf(f[6], f[7]);
    return o;
}

__device__ __forceinline__ u32x4 gate_chunk_regs(u32x4 v, f32x4 g0, f32x4 g1, int relu6) {
    float f[8] = {lo_bf(v.x) * g0.x, hi_bf(v.x) * g0.y, lo_bf(v.y) * g0.z, hi_bf(v.y) * g0.w,
                  lo_bf(v.z) * g1.x, hi_bf(v.z) * g1.y, lo_bf(v.w) * g1.z, hi_bf(v.w) * g1.w};
    if (relu6) {
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = fminf(fmaxf(f[i], 0.f), 6.f);
    }
    u32x4 o;
    o.x = pack2bf(f[0], f[1]); o.y = pack2bf(f[2], f[3]); o.z = pack2bf(f[4], f[5]); o.w = pack2bf(f[6], f[7]);
    return o;
}

__device__ __forceinline__ u32x4 relu6_chunk(u32x4 v) {
    float f[8] = {lo_bf(v.x), hi_bf(v.x), lo_bf(v.y), hi_bf(v.y), lo_bf(v.z), hi_bf(v.z), lo_bf(v.w), hi_bf(v.w)};
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = fminf(fmaxf(f[i], 0.f), 6.f);
    u32x4 o;
    o.x = pack2bf(f[0], f[1]); o.y = pack2bf(f[2], f[3]); o.z = pack2bf(f[4], f[5]); o.w = pack2bf(f[6], f[7]);
    return o;
}

// WM = 16-row sub-tiles per wave (block rows BM = 64*WM), NT = 16-column sub-tiles (BN = 16*NT), BK = 32 or 64.
// LDS row stride BK+8 elements keeps ds_read_b128 of 16 consecutive rows conflict-free for both BK.
template <int WM, int NT, int BK>
__global__ __launch_bounds__(256) void k_gemm_bf16(const GemmArgs g, const int n_tiles, const int nwg) {
    constexpr int BM = 64 * WM;
    constexpr int BN = NT * 16;
    constexpr int LD = BK + 8;
    constexpr int KC = BK / 8;                       // 16-byte chunks per row per K-tile
    constexpr int A_ITERS = BM * KC / 256;
    constexpr int W_ITERS = (BN * KC + 255) / 256;
    constexpr int CLD = BN + 8;                      // epilogue staging row stride
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    bf16_t* As = smem;                               // [2][BM*LD]
    bf16_t* Ws = smem + 2 * BM * LD;                 // [2][BN*LD]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // XCD-aware tile order: blocks b, b+8, ... share an XCD (and its L2); give each XCD a contiguous
    // run of logical tiles so the n-tiles that re-read one A panel hit the same L2.  Bijective for any nwg.
    int logical;
    {
        const int bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int mb = logical / n_tiles, nb = logical - mb * n_tiles;
    const int m0 = mb * BM;
    const int n0 = nb * BN;
    const int Npad = (g.N + 15) & ~15;

    // per-thread staging coordinates (rows fixed across the K loop)
    const bf16_t* a_ptr[A_ITERS];
    const float* a_gate[A_ITERS];
    bool a_ok[A_ITERS];
    int a_lds[A_ITERS], a_k[A_ITERS];
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
        const int id = tid + 256 * i;
        const int row = id / KC, c = id % KC;
        const int m = m0 + row;
        a_ok[i] = m < g.M;
        a_k[i] = c * 8;
        a_lds[i] = row * LD + c * 8;
        a_ptr[i] = g.A + (size_t)(a_ok[i] ? m : 0) * g.lda + c * 8;
        a_gate[i] = g.gate ? g.gate + (size_t)((a_ok[i] ? m : 0) / g.rows_per_img) * g.gate_ld + c * 8 : nullptr;
    }

    u32x4 ra[A_ITERS], rw[W_ITERS];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (a_ok[i] && (k0 + a_k[i]) < g.K) {
                v = *reinterpret_cast<const u32x4*>(a_ptr[i] + k0);
                if (g.gate) v = gate_chunk(v, a_gate[i] + k0, g.a_relu6);
                else if (g.a_relu6) v = relu6_chunk(v);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < W_ITERS; ++i) {
            const int id = tid + 256 * i;
            const int row = id / KC, c = id % KC;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (row < BN && (n0 + row) < Npad && (k0 + c * 8) < g.ldw)
                v = *reinterpret_cast<const u32x4*>(g.W + (size_t)(n0 + row) * g.ldw + k0 + c * 8);
            rw[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) *reinterpret_cast<u32x4*>(&As[buf * BM * LD + a_lds[i]]) = ra[i];
#pragma unroll
        for (int i = 0; i < W_ITERS; ++i) {
            const int id = tid + 256 * i;
            const int row = id / KC, c = id % KC;
            if (row < BN) *reinterpret_cast<u32x4*>(&Ws[buf * BN * LD + row * LD + c * 8]) = rw[i];
        }
    };

    f32x4 acc[NT][WM];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < WM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nt = (g.K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int fr = lane & 15;          // fragment row (n for W, m for A)
    const int fk = (lane >> 4) * 8;    // fragment k offset
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) load_tile((t + 1) * BK);
        const bf16_t* as = As + buf * BM * LD;
        const bf16_t* ws = Ws + buf * BN * LD;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            bf16x8 af[WM];
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
                af[mi] = *reinterpret_cast<const bf16x8*>(&as[(wave * 16 * WM + mi * 16 + fr) * LD + ks * 32 + fk]);
            // W fragments are fetched WG at a time so several ds_read_b128 are in flight behind the MFMAs
            // (hipcc otherwise pairs them and exposes the ~100-cycle LDS latency six times per k-step)
            constexpr int WG = NT % 4 == 0 ? 4 : (NT % 3 == 0 ? 3 : 2);
#pragma unroll
            for (int n0i = 0; n0i < NT; n0i += WG) {
                bf16x8 wf[WG];
#pragma unroll
                for (int q = 0; q < WG; ++q)
                    wf[q] = *reinterpret_cast<const bf16x8*>(&ws[((n0i + q) * 16 + fr) * LD + ks * 32 + fk]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < WG; ++q)
#pragma unroll
                    for (int mi = 0; mi < WM; ++mi)
                        acc[n0i + q][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[q], af[mi], acc[n0i + q][mi], 0, 0, 0);
            }
        }
        if (t + 1 < nt) store_tile(buf ^ 1);
        __syncthreads();
    }

    // bias + activation once, straight-line per activation (see MI355_ACT_DISPATCH); the store paths below only
    // pack / add the residual / write.
    MI355_ACT_DISPATCH(g.act, {
_Pragma("unroll")
        for (int ni = 0; ni < NT; ++ni) {
            const int n = n0 + ni * 16 + (lane >> 4) * 4;
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            if (n < Npad) b = *reinterpret_cast<const f32x4*>(g.bias + n);
_Pragma("unroll")
            for (int mi = 0; mi < WM; ++mi) {
                acc[ni][mi].x = act_c<ACT>(acc[ni][mi].x + b.x); acc[ni][mi].y = act_c<ACT>(acc[ni][mi].y + b.y);
                acc[ni][mi].z = act_c<ACT>(acc[ni][mi].z + b.z); acc[ni][mi].w = act_c<ACT>(acc[ni][mi].w + b.w);
            }
        }
    })
    // epilogue: D[row = n][col = m]; lane holds n = nbase + (lane>>4)*4 + r (r = 0..3), m = lane & 15.
    const bool staged = (!g.out_f32) && (g.res == nullptr);
    if (staged) {
        // bf16 tile -> LDS -> 16-byte row-contiguous stores (the stage buffers are free after the last barrier)
        bf16_t* Cs = smem;   // [BM][CLD]
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) {
            const int ml = wave * 16 * WM + mi * 16 + fr;
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
                const int nl = ni * 16 + (lane >> 4) * 4;
                u32x2 o;
                o.x = pack2bf(acc[ni][mi].x, acc[ni][mi].y);
                o.y = pack2bf(acc[ni][mi].z, acc[ni][mi].w);
                *reinterpret_cast<u32x2*>(&Cs[ml * CLD + nl]) = o;
            }
        }
        __syncthreads();
        constexpr int CPR = BN / 8;   // 16-byte chunks per tile row
        for (int id = tid; id < BM * CPR; id += 256) {
            const int row = id / CPR, c = id - row * CPR;
            const int m = m0 + row, n = n0 + c * 8;
            if (m < g.M && n < g.N)   // N is a multiple of 8 for bf16 outputs
                *reinterpret_cast<u32x4*>((bf16_t*)g.out + (size_t)m * g.ldo + n) =
                    *reinterpret_cast<const u32x4*>(&Cs[row * CLD + c * 8]);
        }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < WM; ++mi) {
        const int m = m0 + wave * 16 * WM + mi * 16 + fr;
        if (m >= g.M) continue;
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            const int n = n0 + ni * 16 + (lane >> 4) * 4;
            if (n >= g.N) continue;
            float v[4] = {acc[ni][mi].x, acc[ni][mi].y, acc[ni][mi].z, acc[ni][mi].w};
            const bool res_vec = g.res && (n + 3 < g.res_n);
            const bool res_part = g.res && !res_vec && (n < g.res_n);
            if (n + 3 < g.N && !res_part) {
                if (res_vec) {
                    const u32x2 rr = *reinterpret_cast<const u32x2*>(g.res + (size_t)m * g.ldr + n);
                    v[0] += lo_bf(rr.x); v[1] += hi_bf(rr.x); v[2] += lo_bf(rr.y); v[3] += hi_bf(rr.y);
                }
                if (g.out_f32) {
                    float* o = (float*)g.out + (size_t)m * g.ldo + n;
                    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
                } else {
                    u32x2 o;
                    o.x = pack2bf(v[0], v[1]);
                    o.y = pack2bf(v[2], v[3]);
                    *reinterpret_cast<u32x2*>((bf16_t*)g.out + (size_t)m * g.ldo + n) = o;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (n + r < g.N) {
                        float x = v[r];
                        if (g.res && n + r < g.res_n) x += bf2f(g.res[(size_t)m * g.ldr + n + r]);
                        if (g.out_f32) ((float*)g.out)[(size_t)m * g.ldo + n + r] = x;
                        else ((bf16_t*)g.out)[(size_t)m * g.ldo + n + r] = f2bf(x);
                    }
                }
            }
        }
    }
}

// =====================================================================================
// Large-K path (K % 64 == 0, Swin's linears and the wide EfficientNet layers): 128 x 128 x 64 tiles, operands
// streamed HBM -> LDS with global_load_lds (16 B per lane, no VGPR round trip), double-buffered so tile t+1
// lands while tile t is multiplied.  The DMA writes LDS lane-linearly (8 rows x 128 B per wave instruction), so
// the bank-conflict swizzle lives on the per-lane SOURCE address: physical 16-byte chunk = logical ^ ((row>>1)&7),
// and the fragment reads apply the same XOR (cdna guide rule 21).  4 waves as 2(M) x 2(N), 64 x 64 per wave.
// Rows past M / N are clamped to the last valid row (finite garbage, never stored), so no guards in the loop.
// =====================================================================================
constexpr int BG_BM = 128, BG_BN = 128, BG_BK = 64;

__device__ __forceinline__ void glds16(const bf16_t* gsrc, bf16_t* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// GATED = the SE-gated (and/or ReLU6'd) projection layers: the DMA cannot transform A on the way to LDS, so the gate is
// applied to the A FRAGMENTS after the LDS read (same value and rounding as gating on load: bf16(A * g)); the four
// waves then split M only (32 rows x all 128 columns each) so that no A fragment is gated twice.
// K does not have to be a multiple of 64: lanes whose 16-byte chunk lies past K (or past row M / N) read from a
// zero page instead (g.zeros), so no garbage ever enters the accumulation.
// BKT = K depth of a stage.  64: the form above (64 KB ring, 67.6 KB with the fp32 residual tile: two workgroups per CU).
// 32 (ungated only): a 32 KB ring and an epilogue that passes the tile through LDS in two 64-row halves - 34.8 KB, FOUR
// workgroups per CU.  For the short-K, wide-N layers (Swin's K = 128 / 256 linears: 2 - 8 k-steps, outputs 3 - 4x the
// inputs) a workgroup's life is dominated by its epilogue: its slot stays occupied until the L2 has acknowledged the
// stores (measured: 128->512 @56x56 takes 202 us, 86 us with the stores removed, and the same 411 MB written by a
// store-only kernel take 67 us), so what helps is more workgroups per CU to wait beside each other.
template <bool GATED, bool KTAIL, int BKT = 64>
__global__ __launch_bounds__(256, BKT == 32 ? 3 : 1) void k_gemm_big(const GemmArgs g, const int n_tiles, const int nwg) {
    constexpr int MI = GATED ? 2 : 4;      // 16-row sub-tiles per wave
    constexpr int NI = GATED ? 8 : 4;      // 16-column sub-tiles per wave
    static_assert(BKT == 64 || (BKT == 32 && !GATED), "stage depths");
    constexpr int PPW = BKT / 16;                  // 1 KB pieces per operand per wave per stage (4 or 2)
    constexpr int RPP = 512 / BKT;                 // tile rows per piece (8 or 16)
    constexpr int CPRW = BKT / 8;                  // 16-byte chunks per tile row (8 or 4)
    extern __shared__ __attribute__((aligned(16))) bf16_t bsm[];
    bf16_t* As = bsm;                              // [2][128*BKT]
    bf16_t* Ws = bsm + 2 * BG_BM * BKT;            // [2][128*BKT]
    // chunk swizzle (applied to the DMA's source address and to the fragment reads alike).  ds_read_b128 is serviced in the lane
    // groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: for 64-byte rows the 16-byte slot is 4*(row%4) + (chunk ^ swz), and
    // swz = -(row/4) mod 4 is what makes the four lanes of a group that share row%4 land on four different slots (the
    // obvious (row/4)%4 is a 2-way conflict: PMC SQ_LDS_BANK_CONFLICT was 47 % of the LDS cycles of this kernel)
    auto swz = [](int row) { return BKT == 64 ? ((row >> 1) & 7) : ((0 - (row >> 2)) & 3); };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int logical;
    {
        const int bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int mb = logical / n_tiles, nb = logical - mb * n_tiles;
    const int m0 = mb * BG_BM, n0 = nb * BG_BN;
    const int Npad = (g.N + 15) & ~15;

    // staging: wave w issues PPW A pieces and PPW W pieces per stage; piece p covers tile rows p*RPP .. p*RPP+RPP-1
    const bf16_t* a_src[PPW];
    const bf16_t* w_src[PPW];
    int k_off[PPW];
    bool a_ok[PPW], w_ok[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int row = (wave * PPW + i) * RPP + lane / CPRW;
        const int lchunk = (lane % CPRW) ^ swz(row);
        k_off[i] = lchunk * 8;
        a_ok[i] = (m0 + row) < g.M;
        w_ok[i] = (n0 + row) < Npad;
        a_src[i] = g.A + (size_t)(a_ok[i] ? m0 + row : 0) * g.lda + lchunk * 8;
        w_src[i] = g.W + (size_t)(w_ok[i] ? n0 + row : 0) * g.ldw + lchunk * 8;
    }
    auto stage = [&](int buf, int k0) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int piece = wave * PPW + i;          // wave-uniform
            // KTAIL = false: K and ldw are multiples of the stage depth and out-of-range rows were clamped to row 0 (finite data,
            // never stored), so the loop has no selects; KTAIL = true routes every out-of-range chunk to the zero page
            const bf16_t* pa = KTAIL ? ((a_ok[i] && k0 + k_off[i] < g.K) ? a_src[i] + k0 : g.zeros) : a_src[i] + k0;
            const bf16_t* pw = KTAIL ? ((w_ok[i] && k0 + k_off[i] < g.ldw) ? w_src[i] + k0 : g.zeros) : w_src[i] + k0;
            glds16(pa, As + buf * BG_BM * BKT + piece * 512);
            glds16(pw, Ws + buf * BG_BN * BKT + piece * 512);
        }
    };

    const int wm = GATED ? wave : (wave >> 1), wn = GATED ? 0 : (wave & 1);
    const int fr = lane & 15, fq = lane >> 4;
    const int row_base = wm * (MI * 16), col_base = wn * (NI * 16);
    f32x4 acc[NI][MI];   // [ni][mi]
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* gate_row[MI];
    if (GATED) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = min(m0 + row_base + mi * 16 + fr, g.M - 1);
            gate_row[mi] = g.gate ? g.gate + (size_t)(m / g.rows_per_img) * g.gate_ld + fq * 8 : nullptr;
        }
    }

    // gate values for one K-tile (this lane's 8 k-values per k-step), fetched one tile ahead so the L2 latency
    // hides behind the previous tile's MFMAs
    f32x4 gnext[MI][2][2];
    auto load_gate = [&](int t) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int k = t * BG_BK + ks * 32;
                const bool ok = GATED && g.gate && (k + fq * 8 < g.K);
                gnext[mi][ks][0] = ok ? *reinterpret_cast<const f32x4*>(gate_row[mi] + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
                gnext[mi][ks][1] = ok ? *reinterpret_cast<const f32x4*>(gate_row[mi] + k + 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
    };

    const int nt = (g.K + BKT - 1) / BKT;
    stage(0, 0);
    if (GATED && g.gate) load_gate(0);
    __syncthreads();   // the fence drains vmcnt for the LDS-DMA as well
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) stage(buf ^ 1, (t + 1) * BKT);
        f32x4 gcur[MI][2][2];
        if (GATED && g.gate) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) { gcur[mi][ks][0] = gnext[mi][ks][0]; gcur[mi][ks][1] = gnext[mi][ks][1]; }
            if (t + 1 < nt) load_gate(t + 1);
        }
        const bf16_t* as = As + buf * BG_BM * BKT;
        const bf16_t* ws = Ws + buf * BG_BN * BKT;
#pragma unroll
        for (int ks = 0; ks < BKT / 32; ++ks) {
            bf16x8 af[MI], wf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int ra = row_base + i * 16 + fr;
                u32x4 v = *reinterpret_cast<const u32x4*>(&as[ra * BKT + (((ks * 4 + fq) ^ swz(ra)) << 3)]);
                if (GATED) {
                    if (g.gate) v = gate_chunk_regs(v, gcur[i][ks][0], gcur[i][ks][1], g.a_relu6);   // zeros past K
                    else if (g.a_relu6) v = relu6_chunk(v);
                }
                af[i] = *reinterpret_cast<bf16x8*>(&v);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int rw = col_base + i * 16 + fr;
                wf[i] = *reinterpret_cast<const bf16x8*>(&ws[rw * BKT + (((ks * 4 + fq) ^ swz(rw)) << 3)]);
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
        }
        __syncthreads();   // next tile has landed (vmcnt(0)) and everyone is done reading this one
    }

    // LayerNorm of the A rows, folded in (swin norm1 -> qkv, norm2 -> fc1): the weights carry gamma, the bias carries W beta, and
    // LN(x) W^T = rstd (x W'^T - mean colsum(W')) - two FMAs per output here instead of a pass that writes and re-reads LN(x)
    if (!GATED && g.ln_stats) {
        float mean[MI], rstd[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = min(m0 + row_base + mi * 16 + fr, g.M - 1);
            const mi355_f32x2 st = *reinterpret_cast<const mi355_f32x2*>(g.ln_stats + (size_t)m * 2);
            mean[mi] = st.x; rstd[mi] = st.y;
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + col_base + ni * 16 + fq * 4;
            f32x4 cs = {0.f, 0.f, 0.f, 0.f};
            if (n < Npad) cs = *reinterpret_cast<const f32x4*>(g.ln_colsum + n);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                acc[ni][mi].x = rstd[mi] * (acc[ni][mi].x - mean[mi] * cs.x); acc[ni][mi].y = rstd[mi] * (acc[ni][mi].y - mean[mi] * cs.y);
                acc[ni][mi].z = rstd[mi] * (acc[ni][mi].z - mean[mi] * cs.z); acc[ni][mi].w = rstd[mi] * (acc[ni][mi].w - mean[mi] * cs.w);
            }
        }
    }
    // epilogue (same contract as k_gemm_bf16): lane holds n = .. + fq*4 + r, m = .. + fr
    MI355_ACT_DISPATCH(g.act, {
_Pragma("unroll")
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + col_base + ni * 16 + fq * 4;
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            if (n < Npad) b = *reinterpret_cast<const f32x4*>(g.bias + n);
_Pragma("unroll")
            for (int mi = 0; mi < MI; ++mi) {
                acc[ni][mi].x = act_c<ACT>(acc[ni][mi].x + b.x); acc[ni][mi].y = act_c<ACT>(acc[ni][mi].y + b.y);
                acc[ni][mi].z = act_c<ACT>(acc[ni][mi].z + b.z); acc[ni][mi].w = act_c<ACT>(acc[ni][mi].w + b.w);
            }
        }
    })
    const bool staged = (!g.out_f32) && (g.res == nullptr);
    if (staged) {
        constexpr int CLD = BG_BN + 8;
        bf16_t* Cs = bsm;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int ml = row_base + mi * 16 + fr;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int nl = col_base + ni * 16 + fq * 4;
                u32x2 o;
                o.x = pack2bf(acc[ni][mi].x, acc[ni][mi].y);
                o.y = pack2bf(acc[ni][mi].z, acc[ni][mi].w);
                *reinterpret_cast<u32x2*>(&Cs[ml * CLD + nl]) = o;
            }
        }
        __syncthreads();
        constexpr int CPR = BG_BN / 8;
        for (int id = tid; id < BG_BM * CPR; id += 256) {
            const int row = id / CPR, c = id - row * CPR;
            const int m = m0 + row, n = n0 + c * 8;
            if (m < g.M && n < g.N)
                *reinterpret_cast<u32x4*>((bf16_t*)g.out + (size_t)m * g.ldo + n) =
                    *reinterpret_cast<const u32x4*>(&Cs[row * CLD + c * 8]);
        }
        return;
    }
    if (!g.out_f32 && g.res_n >= g.N) {
        // residual layers (Swin proj / fc2 update the stream in place): stage the fp32 tile so the add happens
        // before the single bf16 rounding, then stream rows with 16-byte residual loads and stores.
        constexpr int FLD = BG_BN + 4;
        float* Cf = reinterpret_cast<float*>(bsm);   // [128][132] fp32 = 67.6 KB; BKT = 32: [64][132] = 33.8 KB, two passes
        constexpr int HALVES = BKT == 32 ? 2 : 1, HROWS = BG_BM / HALVES;
#pragma unroll
        for (int h = 0; h < HALVES; ++h) {
            if (HALVES == 1 || row_base / HROWS == h) {      // (ungated: a wave's 64 rows lie in one half)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int ml = row_base + mi * 16 + fr - h * HROWS;
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const int nl = col_base + ni * 16 + fq * 4;
                        *reinterpret_cast<f32x4*>(&Cf[ml * FLD + nl]) = acc[ni][mi];
                    }
                }
            }
            __syncthreads();
            constexpr int CPR = BG_BN / 8;
            for (int id = tid; id < HROWS * CPR; id += 256) {
                const int row = id / CPR, c = id - row * CPR;
                const int m = m0 + h * HROWS + row, n = n0 + c * 8;
                if (m < g.M && n < g.N) {
                    const f32x4 c0 = *reinterpret_cast<const f32x4*>(&Cf[row * FLD + c * 8]);
                    const f32x4 c1 = *reinterpret_cast<const f32x4*>(&Cf[row * FLD + c * 8 + 4]);
                    const u32x4 rr = *reinterpret_cast<const u32x4*>(g.res + (size_t)m * g.ldr + n);
                    u32x4 o;
                    o.x = pack2bf(c0.x + lo_bf(rr.x), c0.y + hi_bf(rr.x));
                    o.y = pack2bf(c0.z + lo_bf(rr.y), c0.w + hi_bf(rr.y));
                    o.z = pack2bf(c1.x + lo_bf(rr.z), c1.y + hi_bf(rr.z));
                    o.w = pack2bf(c1.z + lo_bf(rr.w), c1.w + hi_bf(rr.w));
                    *reinterpret_cast<u32x4*>((bf16_t*)g.out + (size_t)m * g.ldo + n) = o;
                }
            }
            if (h + 1 < HALVES) __syncthreads();
        }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + row_base + mi * 16 + fr;
        if (m >= g.M) continue;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + col_base + ni * 16 + fq * 4;
            if (n >= g.N) continue;
            float v[4] = {acc[ni][mi].x, acc[ni][mi].y, acc[ni][mi].z, acc[ni][mi].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (n + r < g.N) {
                    float x = v[r];
                    if (g.res && n + r < g.res_n) x += bf2f(g.res[(size_t)m * g.ldr + n + r]);
                    v[r] = x;
                }
            }
            if (n + 3 < g.N && !g.out_f32) {
                u32x2 o;
                o.x = pack2bf(v[0], v[1]);
                o.y = pack2bf(v[2], v[3]);
                *reinterpret_cast<u32x2*>((bf16_t*)g.out + (size_t)m * g.ldo + n) = o;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (n + r < g.N) {
                        if (g.out_f32) ((float*)g.out)[(size_t)m * g.ldo + n + r] = v[r];
                        else ((bf16_t*)g.out)[(size_t)m * g.ldo + n + r] = f2bf(v[r]);
                    }
                }
            }
        }
    }
}

template <bool GATED, bool KTAIL, int BKT = 64>
static int launch_big(const GemmArgs& a, hipStream_t st) {
    // 67.6 KB: fp32 epilogue tile (>= the 64 KB of stage buffers); BKT = 32: 34.8 KB bf16 tile (>= 33.8 KB half fp32 tile, 32 KB ring)
    const size_t lds = BKT == 64 ? (size_t)BG_BM * (BG_BN + 4) * 4 : (size_t)BG_BM * (BG_BN + 8) * 2;
    static bool attr_done[MI355_MAX_DEVICES] = {false};   // per device
    if (first_time_on_this_device(attr_done)) {
        MI355_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm_big<GATED, KTAIL, BKT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)lds));
    }
    const int n_tiles = cdiv(a.N, BG_BN), m_tiles = cdiv(a.M, BG_BM);
    const long nwg = (long)n_tiles * m_tiles;
    MI355_REQUIRE(nwg < (1l << 31), "gemm: grid too large");
    hipLaunchKernelGGL((k_gemm_big<GATED, KTAIL, BKT>), dim3((unsigned)nwg), dim3(256), lds, st, a, n_tiles, (int)nwg);
    MI355_LAUNCH_CHECK();
    return OK;
}

// =====================================================================================
// Streaming path for the early layers (W small enough to sit in LDS: N*K <= 32k elements, K <= 320):
// the weights are loaded into LDS ONCE per workgroup and stay there; every wave then streams 16-row slabs of A
// straight from HBM into MFMA fragments (a lane needs 16 contiguous bytes of one row: no LDS, no barrier in the
// loop), multiplies against the resident W and writes its rows out.  Slabs are prefetched PF deep in registers
// (8 VGPRs per slab per k-step), so each wave keeps several HBM requests in flight: these layers are pure
// bandwidth (A or the output is 6x the other operand), the tiled kernel spent its time in per-tile
// load -> barrier -> store chains.  Same prologue/epilogue contract as k_gemm_bf16 (gate, ReLU6, bias, act, residual).
// =====================================================================================
template <int NT, int KST>
__global__ __launch_bounds__(256) void k_gemm_stream(const GemmArgs g) {
    constexpr int PF = KST == 1 ? 4 : (KST <= 2 ? 3 : 1);     // slabs in flight per wave
    constexpr int WLD = KST * 32 + 8;
    constexpr int CLD = NT * 16 + 8;
    extern __shared__ __attribute__((aligned(16))) bf16_t wsm[];   // [NT*16][WLD], then 4 x [16][CLD] output strips
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Npad = (g.N + 15) & ~15;
    for (int id = tid; id < NT * 16 * KST * 4; id += 256) {
        const int row = id / (KST * 4), c = id - row * (KST * 4);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row < Npad && c * 8 < g.ldw) v = *reinterpret_cast<const u32x4*>(g.W + (size_t)row * g.ldw + c * 8);
        *reinterpret_cast<u32x4*>(&wsm[row * WLD + c * 8]) = v;
    }
    {
        float* sb = reinterpret_cast<float*>(wsm + NT * 16 * WLD + 4 * 16 * CLD);
        for (int n = tid; n < NT * 16; n += 256) sb[n] = n < Npad ? g.bias[n] : 0.f;
    }
    __syncthreads();

    bf16_t* cst = wsm + NT * 16 * WLD + wave * 16 * CLD;
    // bias lives in LDS (after the four output strips), not in NT*4 registers per lane: the registers buy occupancy
    const float* sbias = reinterpret_cast<const float*>(wsm + NT * 16 * WLD + 4 * 16 * CLD);
    const int fr = lane & 15, fq = lane >> 4;
    const int nslabs = (g.M + 15) >> 4;
    const int stride = gridDim.x * 4;
    int slab = blockIdx.x * 4 + wave;

    auto load_slab = [&](int sl, u32x4 (&dst)[KST]) {
        const int m = sl * 16 + fr;
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) {
            u32x4 v = {0u, 0u, 0u, 0u};
            const int k = ks * 32 + fq * 8;
            if (sl < nslabs && m < g.M && k < g.K) {
                v = *reinterpret_cast<const u32x4*>(g.A + (size_t)m * g.lda + k);
                if (g.gate) v = gate_chunk(v, g.gate + (size_t)(m / g.rows_per_img) * g.gate_ld + k, g.a_relu6);
                else if (g.a_relu6) v = relu6_chunk(v);
            }
            dst[ks] = v;
        }
    };

    u32x4 ring[PF][KST];
#pragma unroll
    for (int p = 0; p < PF; ++p) load_slab(slab + p * stride, ring[p]);


    while (slab < nslabs) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {          // static ring index: slot p holds slab + p*stride
            const int sl = slab + p * stride;
            if (sl < nslabs) {
                f32x4 acc[NT];
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) acc[ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KST; ++ks) {
                    const bf16x8 af = *reinterpret_cast<bf16x8*>(&ring[p][ks]);
#pragma unroll
                    for (int ni = 0; ni < NT; ++ni) {
                        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(&wsm[(ni * 16 + fr) * WLD + ks * 32 + fq * 8]);
                        acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af, acc[ni], 0, 0, 0);
                    }
                }
                load_slab(sl + PF * stride, ring[p]);   // refill this slot PF slabs ahead
                MI355_ACT_DISPATCH(g.act, {
_Pragma("unroll")
                    for (int ni = 0; ni < NT; ++ni) {
                        const f32x4 bb = *reinterpret_cast<const f32x4*>(&sbias[ni * 16 + fq * 4]);
                        acc[ni].x = act_c<ACT>(acc[ni].x + bb.x); acc[ni].y = act_c<ACT>(acc[ni].y + bb.y);
                        acc[ni].z = act_c<ACT>(acc[ni].z + bb.z); acc[ni].w = act_c<ACT>(acc[ni].w + bb.w);
                    }
                })
                // residual in registers (8-byte loads), then the slab goes through this wave's private LDS strip so
                // the HBM writes are whole 16-byte-per-lane rows (32-byte fragments straight from the accumulator
                // layout ran the pure-write layers at 2.2 TB/s)
                const int m = sl * 16 + fr;
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) {
                    const int n = ni * 16 + fq * 4;
                    float v[4] = {acc[ni].x, acc[ni].y, acc[ni].z, acc[ni].w};
                    if (g.res && m < g.M && n < g.N) {
                        if (n + 3 < g.res_n) {
                            const u32x2 rr = *reinterpret_cast<const u32x2*>(g.res + (size_t)m * g.ldr + n);
                            v[0] += lo_bf(rr.x); v[1] += hi_bf(rr.x); v[2] += lo_bf(rr.y); v[3] += hi_bf(rr.y);
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (n + r < g.res_n) v[r] += bf2f(g.res[(size_t)m * g.ldr + n + r]);
                        }
                    }
                    u32x2 o;
                    o.x = pack2bf(v[0], v[1]);
                    o.y = pack2bf(v[2], v[3]);
                    *reinterpret_cast<u32x2*>(&cst[fr * CLD + n]) = o;
                }
                // (same wave wrote and reads: LDS ops complete in order, no barrier needed)
                constexpr int CPR = NT * 2;               // 16-byte chunks per row (padded width)
#pragma unroll
                for (int i = 0; i < (16 * CPR + 63) / 64; ++i) {
                    const int id = lane + 64 * i;
                    const int row = id / CPR, c = id - row * CPR;
                    const int mm = sl * 16 + row;
                    if (id < 16 * CPR && mm < g.M && c * 8 < g.N)
                        *reinterpret_cast<u32x4*>((bf16_t*)g.out + (size_t)mm * g.ldo + c * 8) =
                            *reinterpret_cast<const u32x4*>(&cst[row * CLD + c * 8]);
                }
            }
        }
        slab += PF * stride;
    }
}

template <int NT, int KST>
static int launch_stream_cfg(const GemmArgs& a, hipStream_t st) {
    const size_t lds = (size_t)NT * 16 * (KST * 32 + 8) * 2 + (size_t)4 * 16 * (NT * 16 + 8) * 2 + (size_t)NT * 16 * 4;
    static bool attr_done[MI355_MAX_DEVICES] = {false};   // per device
    if (first_time_on_this_device(attr_done)) {
        MI355_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm_stream<NT, KST>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    }
    const int nslabs = cdiv(a.M, 16);
    int blocks = cdiv(nslabs, 4);
    if (blocks > 256 * 8) blocks = 256 * 8;   // a few workgroups per CU, each streaming many slabs
    hipLaunchKernelGGL((k_gemm_stream<NT, KST>), dim3(blocks), dim3(256), lds, st, a);
    MI355_LAUNCH_CHECK();
    return OK;
}

// =====================================================================================
// Narrow-output projections (N <= 48, K <= 288: the SE-gated 1x1 projections of the 112x112 .. 28x28 blocks).  The tiled
// kernel ran them at 3.4-4.1 TB/s with its waves waiting 70-80 % of the time (load -> barrier -> store per 32-deep
// k-step).  Here a wave owns 16 consecutive rows (one contiguous 16*K*2-byte run of A): it copies them with contiguous
// 16-byte loads into its PRIVATE LDS strip - applying the SE gate / ReLU6 on the way, same rounding point bf16(A * g) as
// gate_chunk - multiplies against W resident in LDS, and writes its 16 x N outputs back as one contiguous run.  No
// workgroup barrier after the prologue; the next tile's loads are requested before this tile's MFMAs.  (Same structure
// as k_dw3_lds, which took the narrow depthwise layers from 2.4 to 4.0 TB/s.)
// =====================================================================================
template <int NT, int KST>
__global__ __launch_bounds__(512) void k_proj_lds(const GemmArgs g) {
    constexpr int NW = 8;
    constexpr int KP = KST * 32;                           // padded K
    constexpr int WLD = KP + 8, ALD = KP + 8, CLD = NT * 16 + 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
    bf16_t* Wsh = reinterpret_cast<bf16_t*>(psm);                                   // [NT*16][WLD]
    float* sbias = reinterpret_cast<float*>(psm + (size_t)NT * 16 * WLD * 2);       // [NT*16]
    unsigned char* wbase = psm + (size_t)NT * 16 * WLD * 2 + NT * 16 * 4;
    constexpr int WS = 16 * ALD * 2 + 16 * CLD * 2 + KP * 4;                        // per wave: A strip | C strip | gate
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    bf16_t* As = reinterpret_cast<bf16_t*>(wbase + (size_t)wave * WS);
    bf16_t* Cs = As + 16 * ALD;
    float* Gs = reinterpret_cast<float*>(Cs + 16 * CLD);
    const int Npad = (g.N + 15) & ~15;
    for (int id = tid; id < NT * 16 * KST * 4; id += 512) {
        const int row = id / (KST * 4), c = id - row * (KST * 4);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row < Npad && c * 8 < g.ldw) v = *reinterpret_cast<const u32x4*>(g.W + (size_t)row * g.ldw + c * 8);
        *reinterpret_cast<u32x4*>(&Wsh[row * WLD + c * 8]) = v;
    }
    for (int n = tid; n < NT * 16; n += 512) sbias[n] = n < Npad ? g.bias[n] : 0.f;
    // the strip's columns K .. KP stay zero (A is only written for k < K)
    for (int id = lane; id < 16 * (ALD / 8); id += 64) *reinterpret_cast<u32x4*>(&As[id * 8]) = (u32x4){0u, 0u, 0u, 0u};
    __syncthreads();

    const int fr = lane & 15, fq = lane >> 4;
    const int kc = g.K >> 3;                               // 16-byte chunks per row
    const int nchunk = 16 * kc;                            // chunks of a 16-row tile (contiguous in memory when lda == K)
    const int ntiles = (g.M + 15) >> 4;
    const int stride = gridDim.x * NW;
    int img = -1;
    u32x4 sv[KST];
    auto stage_load = [&](int tile) {
#pragma unroll
        for (int i = 0; i < KST; ++i) {
            const int id = min(lane + 64 * i, nchunk - 1);
            const int row = id / kc, c = id - row * kc;
            const int m = min(tile * 16 + row, g.M - 1);   // clamped: rows past M are never stored
            sv[i] = *reinterpret_cast<const u32x4*>(g.A + (size_t)m * g.lda + c * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    int tile = blockIdx.x * NW + wave;
    if (tile < ntiles) stage_load(tile);
    for (; tile < ntiles; tile += stride) {
        if (g.gate) {
            const int im = (tile * 16) / g.rows_per_img;   // (host: rows_per_img % 16 == 0, a tile lies in one image)
            if (im != img) {
                img = im;
                for (int k4 = lane; k4 * 4 < g.K; k4 += 64)
                    *reinterpret_cast<f32x4*>(&Gs[k4 * 4]) = *reinterpret_cast<const f32x4*>(g.gate + (size_t)im * g.gate_ld + k4 * 4);
            }
        }
#pragma unroll
        for (int i = 0; i < KST; ++i) {
            const int id = lane + 64 * i;
            if (id < nchunk) {
                const int row = id / kc, c = id - row * kc;
                u32x4 v = sv[i];
                if (g.gate) v = gate_chunk(v, Gs + c * 8, g.a_relu6);
                else if (g.a_relu6) v = relu6_chunk(v);
                *reinterpret_cast<u32x4*>(&As[row * ALD + c * 8]) = v;
            }
        }
        if (tile + stride < ntiles) stage_load(tile + stride);
        // accumulate from zero and add the bias afterwards, k-steps in ascending order: bit-identical to k_gemm_bf16, so which
        // of the two kernels a batch size selects never shows in the result (tests: batch-neighbour invariance)
        f32x4 acc[NT];
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(&As[fr * ALD + ks * 32 + fq * 8]);
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(&Wsh[(ni * 16 + fr) * WLD + ks * 32 + fq * 8]);
                acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af, acc[ni], 0, 0, 0);
            }
        }
        MI355_ACT_DISPATCH(g.act, {
_Pragma("unroll")
            for (int ni = 0; ni < NT; ++ni) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(&sbias[ni * 16 + fq * 4]);
                acc[ni].x = act_c<ACT>(acc[ni].x + bb.x); acc[ni].y = act_c<ACT>(acc[ni].y + bb.y);
                acc[ni].z = act_c<ACT>(acc[ni].z + bb.z); acc[ni].w = act_c<ACT>(acc[ni].w + bb.w);
            }
        })
        const int m = tile * 16 + fr;
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            const int n = ni * 16 + fq * 4;
            float v[4] = {acc[ni].x, acc[ni].y, acc[ni].z, acc[ni].w};
            if (g.res && m < g.M && n < g.N) {
                if (n + 3 < g.res_n) {
                    const u32x2 rr = *reinterpret_cast<const u32x2*>(g.res + (size_t)m * g.ldr + n);
                    v[0] += lo_bf(rr.x); v[1] += hi_bf(rr.x); v[2] += lo_bf(rr.y); v[3] += hi_bf(rr.y);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n + r < g.res_n) v[r] += bf2f(g.res[(size_t)m * g.ldr + n + r]);
                }
            }
            u32x2 o;
            o.x = pack2bf(v[0], v[1]);
            o.y = pack2bf(v[2], v[3]);
            *reinterpret_cast<u32x2*>(&Cs[fr * CLD + n]) = o;
        }
        // (same wave wrote and reads: LDS ops complete in order, no barrier needed)
        constexpr int CPR = NT * 2;               // 16-byte chunks per row (padded width)
#pragma unroll
        for (int i = 0; i < (16 * CPR + 63) / 64; ++i) {
            const int id = lane + 64 * i;
            const int row = id / CPR, c = id - row * CPR;
            const int mm = tile * 16 + row;
            if (id < 16 * CPR && mm < g.M && c * 8 < g.N)
                *reinterpret_cast<u32x4*>((bf16_t*)g.out + (size_t)mm * g.ldo + c * 8) =
                    *reinterpret_cast<const u32x4*>(&Cs[row * CLD + c * 8]);
        }
    }
}

template <int NT, int KST>
static int launch_proj_cfg(const GemmArgs& a, hipStream_t st) {
    constexpr int KP = KST * 32;
    const size_t lds = (size_t)NT * 16 * (KP + 8) * 2 + (size_t)NT * 16 * 4 +
                       (size_t)8 * (16 * (KP + 8) * 2 + 16 * (NT * 16 + 8) * 2 + KP * 4);
    static bool attr_done[MI355_MAX_DEVICES] = {false};   // per device
    if (first_time_on_this_device(attr_done)) {
        MI355_CHECK_HIP(hipFuncSetAttribute((const void*)k_proj_lds<NT, KST>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            160 * 1024));
    }
    const int ntiles = cdiv(a.M, 16);
    int blocks = cdiv(ntiles, 8);
    if (blocks > 256 * 6) blocks = 256 * 6;     // a few workgroups per CU, each streaming many tiles
    hipLaunchKernelGGL((k_proj_lds<NT, KST>), dim3(blocks), dim3(512), lds, st, a);
    MI355_LAUNCH_CHECK();
    return OK;
}

// -1 = shape not covered
static int try_launch_proj(const GemmArgs& a, hipStream_t st) {
    // measured (EfficientNet-B3a B=256): 112x112 / 56x56 projections 0.111 -> 0.087, 0.135 -> 0.108, 0.082 -> 0.058, 0.105 -> 0.096 ms;
    // the 28x28 ones (M = 200k rows, K = 192 / 288: one workgroup per CU) lose 0.030 -> 0.039, 0.044 -> 0.057: large M only
    if (a.out_f32 || a.M < (1 << 19) || a.N % 8 || a.ldo % 8 || a.N > 64 || a.K > 288 || a.lda != a.K) return -1;
    if (a.gate && (a.rows_per_img % 16 != 0 || a.gate_ld % 4 != 0)) return -1;
    const int kst = (a.K + 31) / 32, nt = (a.N + 15) / 16;
#define PROJ_CASE(NTV, KSTV) if (nt == NTV && kst == KSTV) return launch_proj_cfg<NTV, KSTV>(a, st)
    PROJ_CASE(2, 1); PROJ_CASE(2, 2); PROJ_CASE(2, 5); PROJ_CASE(2, 6); PROJ_CASE(3, 6); PROJ_CASE(3, 9); PROJ_CASE(2, 9);
    PROJ_CASE(4, 8); PROJ_CASE(3, 5);        // rexnet_150: 246->58 @56x56 (+ residual), 144->41 @56x56
#undef PROJ_CASE
    return -1;
}

// returns -1 when the shape is not covered (caller falls back to the tiled kernel).
// Measured per layer (profiles/r01_effnet_per_op.txt, tools/gemm_sweep.py): streaming wins where the output row is wide and
// K is ONE k-step: 24->144 @112x112 0.375 -> 0.284 ms, 32->192 @56x56 0.129 -> 0.107 ms (once the bias moved from 48
// registers per lane to LDS; RexNet's 32->192 @112x112: 0.52 -> 0.40 ms).  The tiled kernel wins on 48->288 (two k-steps:
// 0.052 vs 0.078 ms) and on every gated projection (re-measured after the epilogue fixes: 0.10 vs 0.12 ms on 192->32).
static int try_launch_stream(const GemmArgs& a, hipStream_t st) {
    if (a.out_f32 || a.M < 4096 || a.N % 8 || a.ldo % 8 || a.gate || a.a_relu6) return -1;
    const int kst = (a.K + 31) / 32;
    const int nt = (a.N + 15) / 16;
    if (kst == 1 && nt == 9) return launch_stream_cfg<9, 1>(a, st);
    if (kst == 1 && nt == 12) return launch_stream_cfg<12, 1>(a, st);
    return -1;
}

// =====================================================================================
// Split-K for tiny-M, deep-K projections (the SE-gated 1x1 projections of the 14x14 / 7x7 blocks at small batch: M = 49
// .. a few thousand rows, K = 576 .. 2304).  The tiled kernels give such a layer 1-30 workgroups, each walking K serially
// (one L2 round trip per 32- or 64-deep step: 30-68 us at B = 1).  Here the grid is (column tile, row tile, K chunk of
// 256): a workgroup requests its whole 64 x 256 A chunk (gate / ReLU6 applied on the way, same rounding as gate_chunk)
// and its 64 x 256 W chunk at once - ONE round trip - multiplies them from LDS and writes an fp32 partial tile; a second
// kernel adds the partials in ascending chunk order, then bias, activation and residual.  Deterministic, and the chunking
// depends on the layer only, so an image's result does not depend on its batch position or (below the M cap) batch size.
// =====================================================================================
constexpr int SK_KC = 256, SK_BM = 64, SK_BN = 64, SK_LD = SK_KC + 8;
constexpr size_t SK_MAX_BYTES = (size_t)64 << 20;       // partials of one layer: beyond this the round trip of the partials costs more than the serial K loop

int gemm_splitk_chunks(long M, int rows_per_img, int N, int K) {
    // measured (EfficientNet-B3a forward, tools/bench_small_batch.py): B=1 1.49 -> 1.04 ms, B=16 1.84 -> 1.21, B=32 1.66 -> 1.37,
    // B=64 2.09 -> 1.99; at B=128 (M = 25 088 rows at 14x14) the partials' round trip loses (2.88 -> 3.05): M <= 16 384 rows
    if (M > 16384 || rows_per_img > 196 || K < 512 || N < 64 || N % 8 || gemm_splitk_bytes(M, N, K) > SK_MAX_BYTES) return 0;
    return (K + SK_KC - 1) / SK_KC;
}
size_t gemm_splitk_bytes(long M, int N, int K) {
    return (size_t)((K + SK_KC - 1) / SK_KC) * (size_t)M * (size_t)((N + 15) & ~15) * sizeof(float);
}

__global__ __launch_bounds__(256) void k_gemm_splitk(const GemmArgs g, float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) bf16_t sksm[];
    bf16_t* As = sksm;                         // [64][SK_LD]
    bf16_t* Ws = sksm + SK_BM * SK_LD;         // [64][SK_LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * SK_BN, m0 = blockIdx.y * SK_BM, k0 = blockIdx.z * SK_KC;
    const int Npad = (g.N + 15) & ~15;
    u32x4 ra[8], rw[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {                           // all 16 loads of the thread first: one round trip
        const int id = tid + 256 * i;
        const int row = id >> 5, c = id & 31;
        const int k = k0 + c * 8;
        const int m = min(m0 + row, g.M - 1), kk = min(k, g.K - 8);
        ra[i] = *reinterpret_cast<const u32x4*>(g.A + (size_t)m * g.lda + kk);
        const int n = min(n0 + row, Npad - 1), kw = min(k, g.ldw - 8);
        rw[i] = *reinterpret_cast<const u32x4*>(g.W + (size_t)n * g.ldw + kw);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int id = tid + 256 * i;
        const int row = id >> 5, c = id & 31;
        const int k = k0 + c * 8;
        u32x4 v = ra[i];
        if (m0 + row < g.M && k < g.K) {
            if (g.gate) v = gate_chunk(v, g.gate + (size_t)((m0 + row) / g.rows_per_img) * g.gate_ld + k, g.a_relu6);
            else if (g.a_relu6) v = relu6_chunk(v);
        } else {
            v = (u32x4){0u, 0u, 0u, 0u};
        }
        *reinterpret_cast<u32x4*>(&As[row * SK_LD + c * 8]) = v;
        u32x4 w = rw[i];
        if (n0 + row >= Npad || k >= g.ldw) w = (u32x4){0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(&Ws[row * SK_LD + c * 8]) = w;
    }
    __syncthreads();
    const int fr = lane & 15, fq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    f32x4 acc[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < SK_KC / 32; ++ks) {
        bf16x8 af[2], wf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8*>(&As[(wm * 32 + i * 16 + fr) * SK_LD + ks * 32 + fq * 8]);
#pragma unroll
        for (int j = 0; j < 2; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(&Ws[(wn * 32 + j * 16 + fr) * SK_LD + ks * 32 + fq * 8]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[j][i], 0, 0, 0);
    }
    float* wsk = ws + (size_t)blockIdx.z * g.M * Npad;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 32 + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 32 + j * 16 + fq * 4;
            if (m < g.M && n < Npad) *reinterpret_cast<f32x4*>(wsk + (size_t)m * Npad + n) = acc[j][i];
        }
    }
}

__global__ __launch_bounds__(256) void k_splitk_reduce(const GemmArgs g, const float* __restrict__ ws, int nchunks) {
    const int Npad = (g.N + 15) & ~15, n4s = Npad >> 2;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)g.M * n4s) return;
    const int m = (int)(t / n4s), n = (int)(t - (long)m * n4s) * 4;
    if (n >= g.N) return;
    f32x4 s = *reinterpret_cast<const f32x4*>(ws + (size_t)m * Npad + n);
    for (int c = 1; c < nchunks; ++c) {                     // ascending chunk order
        const f32x4 p = *reinterpret_cast<const f32x4*>(ws + ((size_t)c * g.M + m) * Npad + n);
        s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    const f32x4 bb = *reinterpret_cast<const f32x4*>(g.bias + n);
    float v[4] = {apply_act(s.x + bb.x, g.act), apply_act(s.y + bb.y, g.act), apply_act(s.z + bb.z, g.act), apply_act(s.w + bb.w, g.act)};
    if (g.res) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n + r < g.res_n) v[r] += bf2f(g.res[(size_t)m * g.ldr + n + r]);
    }
    u32x2 o;
    o.x = pack2bf(v[0], v[1]);
    o.y = pack2bf(v[2], v[3]);
    *reinterpret_cast<u32x2*>((bf16_t*)g.out + (size_t)m * g.ldo + n) = o;
}

static int launch_splitk(const GemmArgs& a, int nchunks, hipStream_t st) {
    const size_t lds = (size_t)(SK_BM + SK_BN) * SK_LD * 2;
    static bool attr_done[MI355_MAX_DEVICES] = {false};   // per device
    if (first_time_on_this_device(attr_done)) {
        MI355_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm_splitk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const int Npad = (a.N + 15) & ~15;
    hipLaunchKernelGGL(k_gemm_splitk, dim3((unsigned)cdiv(Npad, SK_BN), (unsigned)cdiv(a.M, SK_BM), (unsigned)nchunks), dim3(256), lds, st,
                       a, a.splitk_ws);
    MI355_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)cdiv((long)a.M * (Npad / 4), 256)), dim3(256), 0, st, a,
                       (const float*)a.splitk_ws, nchunks);
    MI355_LAUNCH_CHECK();
    return OK;
}

// Tile selection.  BN = 16*NT minimising padded columns (prefer fewer, larger tiles: fewer A-panel re-reads);
// BM = 64 when a 128-row tiling would leave the 256 CUs with < 4 blocks each (late 14x14 / 7x7 layers are
// latency-bound: more, smaller blocks overlap their load latency); BK = 64 once K >= 64.
static int pick_nt(int N, int max_nt = 12) {
    static const int opts[] = {2, 3, 4, 6, 8, 9, 12};
    int best = 2;
    double best_cost = 1e30;
    for (int nt : opts) {
        if (nt > max_nt) continue;
        const int bn = nt * 16;
        const int tiles = (N + bn - 1) / bn;
        const double cost = (double)tiles * bn * (1.0 + 24.0 / bn) + 8.0 * tiles;
        if (cost < best_cost) { best_cost = cost; best = nt; }
    }
    return best;
}

template <int WM, int NT, int BK>
static int launch_cfg(const GemmArgs& a, hipStream_t st) {
    constexpr int BM = 64 * WM, BN = NT * 16, LD = BK + 8;
    constexpr size_t stage = (size_t)2 * (BM + BN) * LD * 2, ctile = (size_t)BM * (BN + 8) * 2;
    constexpr size_t lds = stage > ctile ? stage : ctile;
    static bool attr_done[MI355_MAX_DEVICES] = {false};   // per device
    if (first_time_on_this_device(attr_done)) {
        MI355_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm_bf16<WM, NT, BK>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const int n_tiles = cdiv(a.N, BN);
    const int m_tiles = cdiv(a.M, BM);
    const long nwg = (long)n_tiles * m_tiles;
    MI355_REQUIRE(nwg < (1l << 31), "gemm: grid too large");
    hipLaunchKernelGGL((k_gemm_bf16<WM, NT, BK>), dim3((unsigned)nwg), dim3(256), lds, st, a, n_tiles, (int)nwg);
    MI355_LAUNCH_CHECK();
    return OK;
}

template <int WM, int BK>
static int launch_nt(const GemmArgs& a, int nt, hipStream_t st) {
    switch (nt) {
        case 2: return launch_cfg<WM, 2, BK>(a, st);
        case 3: return launch_cfg<WM, 3, BK>(a, st);
        case 4: return launch_cfg<WM, 4, BK>(a, st);
        case 6: return launch_cfg<WM, 6, BK>(a, st);
        case 8: return launch_cfg<WM, 8, BK>(a, st);
        case 9: return launch_cfg<WM, 9, BK>(a, st);
        default: return launch_cfg<WM, 12, BK>(a, st);
    }
}

// =====================================================================================
// Head 1x1 conv + bias + activation + GLOBAL AVERAGE POOL in one kernel (get_fm, train/train.py:84-103, is the
// epilogue of conv_head / features.16 whenever the caller wants the pooled embedding): one workgroup = one image x 128 output
// channels, so a tile never spans two images and the pooled value of a channel is complete inside the workgroup.  The head
// tensor (B x 49 x 1536 bf16: 38.5 MB written and re-read at B = 256) and the k_gap launch (25 us) disappear.
// Same arithmetic as the two-kernel path, bit for bit: fp32 MFMA accumulation over ascending k-steps from zero, bias added
// last, activation, ONE bf16 rounding per element, then the sequential fp32 sum over the pixels in ascending order and one
// multiply by 1 / HW (k_gap's order).  Operands come straight from L2 into MFMA fragments (K <= 512, HW <= 64: 12 k-steps).
// =====================================================================================
template <int ACT>
__global__ __launch_bounds__(256) void k_head_gap(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W, int ldw,
                                                  const float* __restrict__ bias, float* __restrict__ pooled,
                                                  bf16_t* __restrict__ pooled_bf16, int ldp, int HW, int N, int K) {
    constexpr int TLD = 132;
    __shared__ __attribute__((aligned(16))) float T[64 * TLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int b = blockIdx.y, n0 = blockIdx.x * 128;
    const int Npad = (N + 15) & ~15;
    const bf16_t* Ab = A + (size_t)b * HW * lda;
    const bf16_t* arow[2];
    const bf16_t* wrow[4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) arow[mi] = Ab + (size_t)min(wm * 32 + mi * 16 + fr, HW - 1) * lda;   // rows past the image: clamped, never summed
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) wrow[ni] = W + (size_t)min(n0 + wn * 64 + ni * 16 + fr, Npad - 1) * ldw;
    f32x4 acc[4][2];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nks = (K + 31) >> 5;
    u32x4 af[2][2], wf[2][4];
    // k >= lda (the weight matrix is zero there): re-read the row's last 16 bytes instead of running past it
    auto load = [&](int s, int ks) {
        const int ka = min(ks * 32 + fq * 8, lda - 8), kw = ks * 32 + fq * 8;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) af[s][mi] = *reinterpret_cast<const u32x4*>(arow[mi] + ka);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) wf[s][ni] = *reinterpret_cast<const u32x4*>(wrow[ni] + kw);
    };
    load(0, 0);
    for (int ks = 0; ks < nks; ks += 2) {
        if (ks + 1 < nks) load(1, ks + 1);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wf[0][ni]), *reinterpret_cast<bf16x8*>(&af[0][mi]), acc[ni][mi], 0, 0, 0);
        if (ks + 1 < nks) {
            if (ks + 2 < nks) load(0, ks + 2);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wf[1][ni]), *reinterpret_cast<bf16x8*>(&af[1][mi]), acc[ni][mi], 0, 0, 0);
        }
    }
    // bias + activation + the bf16 rounding the head tensor would have had, into the LDS tile [pixel][channel]
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int nl = wn * 64 + ni * 16 + fq * 4;
        f32x4 bb = {0.f, 0.f, 0.f, 0.f};
        if (n0 + nl < Npad) bb = *reinterpret_cast<const f32x4*>(bias + n0 + nl);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int ml = wm * 32 + mi * 16 + fr;
            f32x4 v;
            v.x = bf2f(f2bf(act_c<ACT>(acc[ni][mi].x + bb.x))); v.y = bf2f(f2bf(act_c<ACT>(acc[ni][mi].y + bb.y)));
            v.z = bf2f(f2bf(act_c<ACT>(acc[ni][mi].z + bb.z))); v.w = bf2f(f2bf(act_c<ACT>(acc[ni][mi].w + bb.w)));
            *reinterpret_cast<f32x4*>(&T[ml * TLD + nl]) = v;
        }
    }
    __syncthreads();
    if (tid < 128 && n0 + tid < N) {
        float s = 0.f;
        for (int i = 0; i < HW; ++i) s += T[i * TLD + tid];
        s *= 1.0f / (float)HW;
        pooled[(size_t)b * ldp + n0 + tid] = s;
        if (pooled_bf16) pooled_bf16[(size_t)b * ldp + n0 + tid] = f2bf(s);
    }
}

bool head_gap_supported(int HW, int N, int K, int lda, int ldw, int act) {
    return HW >= 1 && HW <= 64 && K >= 32 && K <= 512 && lda % 8 == 0 && lda >= 8 && ldw % 32 == 0 && ldw >= ((K + 31) & ~31) && N % 8 == 0 &&
           (act == ACT_SILU || act == ACT_NONE);
}

int launch_head_gap(const bf16_t* A, int lda, const bf16_t* W, int ldw, const float* bias, float* pooled, bf16_t* pooled_bf16,
                    int ldp, int B, int HW, int N, int K, int act, hipStream_t st) {
    MI355_REQUIRE(head_gap_supported(HW, N, K, lda, ldw, act), "head_gap: unsupported shape HW=%d N=%d K=%d", HW, N, K);
    const dim3 grid((unsigned)cdiv(N, 128), (unsigned)B);
    if (act == ACT_SILU) hipLaunchKernelGGL(k_head_gap<ACT_SILU>, grid, dim3(256), 0, st, A, lda, W, ldw, bias, pooled, pooled_bf16, ldp, HW, N, K);
    else hipLaunchKernelGGL(k_head_gap<ACT_NONE>, grid, dim3(256), 0, st, A, lda, W, ldw, bias, pooled, pooled_bf16, ldp, HW, N, K);
    MI355_LAUNCH_CHECK();
    return OK;
}

int launch_gemm_bf16(const GemmArgs& a, hipStream_t st) {
    MI355_REQUIRE(a.M >= 1 && a.N >= 1 && a.K >= 1, "gemm: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
    MI355_REQUIRE(a.K % 8 == 0 && a.lda % 8 == 0 && a.ldw % 32 == 0, "gemm: K=%d lda=%d ldw=%d alignment", a.K, a.lda,
                  a.ldw);
    MI355_REQUIRE(a.out_f32 || (a.ldo % 8 == 0 && a.N % 8 == 0), "gemm: bf16 output needs N, ldo multiples of 8 (N=%d ldo=%d)",
                  a.N, a.ldo);
    MI355_REQUIRE(!a.res || a.ldr % 4 == 0, "gemm: residual stride %d must be a multiple of 4", a.ldr);
    static const int use_splitk = getenv("MI355_GEMM_SPLITK") ? atoi(getenv("MI355_GEMM_SPLITK")) : 1;
    if (use_splitk && a.splitk_ws && !a.out_f32 && a.ldo % 4 == 0 && !a.ln_stats) {   // (the split-K reduction has no LayerNorm epilogue)
        // (decided by the caller's whole batch: split-K sums K in 256-deep chunks, i.e. rounds differently from the serial loop, and a
        //  microbatch / lane chunk of a big batch must give the same bits as the unchunked forward)
        const int nch = gemm_splitk_chunks(a.M_sel > 0 ? a.M_sel : a.M, a.rows_per_img > 0 ? a.rows_per_img : a.M, a.N, a.K);
        if (nch >= 2 && gemm_splitk_bytes(a.M, a.N, a.K) <= a.splitk_ws_bytes && (!a.gate || a.gate_ld % 4 == 0))
            return launch_splitk(a, nch, st);
    }
    // Opt-in (MI355_GEMM_WIDE=1): the persistent 256-wide tile kernel (gemm_wide.hip), bit-identical to k_gemm_big on the same shape.
    // Off by default: on Swin's linears it reaches 0.7 - 1.0x of k_gemm_big (profiles/r03_gemm_wide_ab.txt) - both are bound by the
    // ~27 B/clk a CU takes in through LDS-DMA / stores (tools/dma_probe.hip), and the lock-step 8-wave tile exposes its epilogue.
    const int use_wide = getenv("MI355_GEMM_WIDE") ? atoi(getenv("MI355_GEMM_WIDE")) : 0;   // (read per call: tools / tests A/B it in one process)
    if (use_wide && gemm_wide_supported(a)) return launch_gemm_wide(a, st);
    // Measured on MI355X (profiles/r01_effnet_per_op_*.txt): the 64-row / BK=64 variants lose to 128 x BN x 32
    // on every EfficientNet layer (each wave re-reads the whole W tile from LDS, so halving the rows per wave
    // makes the block LDS-bound); they stay instantiated for tiny-M problems (classifier, M = batch).
    // K-deep shapes: DMA-staged 128x128x64 kernel (2.6x the register-staged kernel on the 7x7 projections:
    // 21 us vs 55 us at M=12544, N=232, K=1392); gated / ReLU6'd A operands use the fragment-gating variant
    if (a.ln_stats) {   // only k_gemm_big<false, ..> has the LayerNorm epilogue: the executor must not ask for it elsewhere
        const bool ktail_ = (a.K % 64 != 0) || (a.ldw % 64 != 0);
        MI355_REQUIRE(a.zeros && a.K >= 128 && a.N >= 96 && a.M >= 1024 && ((uintptr_t)a.A % 16 == 0) && a.lda % 8 == 0 && !a.gate &&
                      !a.a_relu6 && !ktail_, "gemm: LayerNorm folding needs the DMA-tiled kernel (M=%d N=%d K=%d)", a.M, a.N, a.K);
    }
    // Short K (64 .. 128, any tail), outputs at least twice as wide as the inputs, many rows (RexNet's 77->462 @56x56, 100->600 and
    // 122->732 @28x28; Swin's K = 128 linears take the same instantiation below): the three-workgroup form of the DMA kernel
    static const int short_min_k = getenv("MI355_GEMM_SHORT_MIN_K") ? atoi(getenv("MI355_GEMM_SHORT_MIN_K")) : 64;
    if (a.zeros && a.K >= short_min_k && a.K < 128 && a.N >= 2 * a.K && a.N >= 96 && a.M >= 32768 && ((uintptr_t)a.A % 16 == 0) &&
        a.lda % 8 == 0 && !a.gate && !a.a_relu6 && !a.ln_stats) {
        const bool ktail32 = (a.K % 32 != 0) || (a.ldw % 32 != 0);
        return ktail32 ? launch_big<false, true, 32>(a, st) : launch_big<false, false, 32>(a, st);
    }
    if (a.zeros && a.K >= 128 && a.N >= 64 && a.M >= 1024 && ((uintptr_t)a.A % 16 == 0) && a.lda % 8 == 0 &&
        (!a.gate || a.gate_ld >= a.K)) {
        // 128-wide column tiles for the gated variant: one tile needs N >= 72, several need < ~37 % padding
        // (measured: N = 136 -> 256 loses to the 144-wide register-staged tile, 0.065 vs 0.055 ms on 816->136 @14x14;
        //  N = 77..96 and 167..280 win: rexnet_200 GEMM time 3.96 -> 3.79 ms, efficientnet_b3a 2.35 -> 2.31 ms)
        const bool fits = a.N <= BG_BN ? a.N >= 72 : (long)a.N * 16 >= (long)cdiv(a.N, BG_BN) * BG_BN * 10;
        const bool ktail = (a.K % 64 != 0) || (a.ldw % 64 != 0);
        if ((a.gate || a.a_relu6) && fits) return ktail ? launch_big<true, true>(a, st) : launch_big<true, false>(a, st);
        if (!a.gate && !a.a_relu6 && a.N >= 96) {
            // short K, outputs at least as wide as the inputs: four small-ring workgroups per CU (see k_gemm_big)
            static const int short_k = getenv("MI355_GEMM_SHORT_K") ? atoi(getenv("MI355_GEMM_SHORT_K")) : 128;
            if (!ktail && a.K <= short_k && a.N >= a.K && a.M >= 32768) return launch_big<false, false, 32>(a, st);
            return ktail ? launch_big<false, true>(a, st) : launch_big<false, false>(a, st);
        }
    }
    // small-K layers are pure streaming (one or two K tiles): narrower tiles keep 4+ waves per SIMD resident
    static const int use_stream = getenv("MI355_GEMM_STREAM") ? atoi(getenv("MI355_GEMM_STREAM")) : 1;
    if (use_stream) {
        const int e = try_launch_stream(a, st);
        if (e >= 0) return e;
    }
    static const int use_proj = getenv("MI355_GEMM_PROJ") ? atoi(getenv("MI355_GEMM_PROJ")) : 1;
    if (use_proj) {
        const int e = try_launch_proj(a, st);
        if (e >= 0) return e;
    }
    // (measured, tools/gemm_sweep.py: N=192,K=32 runs 172 us as one 192-wide tile, 133 us as three 64-wide tiles;
    //  N=144,K=24 is best as one 144-wide tile)
    const int small_cap = a.N % 64 == 0 ? 4 : 9;
    const int nt = pick_nt(a.N, a.K <= 64 ? small_cap : 12);
    if (a.M <= 64) return a.K >= 64 ? launch_nt<1, 64>(a, nt, st) : launch_nt<1, 32>(a, nt, st);
    // (64-row tiles for the late layers with few 128-row tiles were measured: no change, so they are not used)
    return launch_nt<2, 32>(a, nt, st);
}

}  // namespace mi355
