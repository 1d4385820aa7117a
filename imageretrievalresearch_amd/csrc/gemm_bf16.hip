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

namespace mi355 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int GM_BM = 128;
constexpr int GM_BK = 32;
constexpr int GM_LD = 40;  // LDS row stride in bf16 elements (80 B): ds_read_b128 of 16 rows is conflict-free

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

__device__ __forceinline__ u32x4 relu6_chunk(u32x4 v) {
    float f[8] = {lo_bf(v.x), hi_bf(v.x), lo_bf(v.y), hi_bf(v.y), lo_bf(v.z), hi_bf(v.z), lo_bf(v.w), hi_bf(v.w)};
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = fminf(fmaxf(f[i], 0.f), 6.f);
    u32x4 o;
    o.x = pack2bf(f[0], f[1]); o.y = pack2bf(f[2], f[3]); o.z = pack2bf(f[4], f[5]); o.w = pack2bf(f[6], f[7]);
    return o;
}

template <int NT>
__global__ __launch_bounds__(256) void k_gemm_bf16(const GemmArgs g, const int n_tiles, const int nwg) {
    constexpr int BN = NT * 16;
    constexpr int W_ITERS = (NT * 64 + 255) / 256;
    __shared__ __attribute__((aligned(16))) bf16_t As[2][GM_BM * GM_LD];
    __shared__ __attribute__((aligned(16))) bf16_t Ws[2][BN * GM_LD];

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
    const int m0 = mb * GM_BM;
    const int n0 = nb * BN;
    const int Npad = (g.N + 15) & ~15;

    // per-thread staging coordinates
    const int a_c = tid & 3;
    int a_row[2];
    const float* a_gate[2];
    const bf16_t* a_ptr[2];
    bool a_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a_row[i] = (tid >> 2) + 64 * i;
        const int m = m0 + a_row[i];
        a_ok[i] = m < g.M;
        a_ptr[i] = g.A + (size_t)(a_ok[i] ? m : 0) * g.lda + a_c * 8;
        a_gate[i] = g.gate ? g.gate + (size_t)((a_ok[i] ? m : 0) / g.rows_per_img) * g.gate_ld + a_c * 8 : nullptr;
    }

    u32x4 ra[2], rw[W_ITERS];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (a_ok[i] && (k0 + a_c * 8) < g.K) {
                v = *reinterpret_cast<const u32x4*>(a_ptr[i] + k0);
                if (g.gate) v = gate_chunk(v, a_gate[i] + k0, g.a_relu6);
                else if (g.a_relu6) v = relu6_chunk(v);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < W_ITERS; ++i) {
            const int id = tid + 256 * i;
            const int row = id >> 2, c = id & 3;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (row < BN && (n0 + row) < Npad)
                v = *reinterpret_cast<const u32x4*>(g.W + (size_t)(n0 + row) * g.ldw + k0 + c * 8);
            rw[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            *reinterpret_cast<u32x4*>(&As[buf][a_row[i] * GM_LD + a_c * 8]) = ra[i];
#pragma unroll
        for (int i = 0; i < W_ITERS; ++i) {
            const int id = tid + 256 * i;
            const int row = id >> 2, c = id & 3;
            if (row < BN) *reinterpret_cast<u32x4*>(&Ws[buf][row * GM_LD + c * 8]) = rw[i];
        }
    };

    f32x4 acc[NT][2];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nt = (g.K + GM_BK - 1) / GM_BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int fr = lane & 15;          // fragment row (n for W, m for A)
    const int fk = (lane >> 4) * 8;    // fragment k offset
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) load_tile((t + 1) * GM_BK);
        bf16x8 af[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
            af[mi] = *reinterpret_cast<const bf16x8*>(&As[buf][(wave * 32 + mi * 16 + fr) * GM_LD + fk]);
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(&Ws[buf][(ni * 16 + fr) * GM_LD + fk]);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[mi], acc[ni][mi], 0, 0, 0);
        }
        if (t + 1 < nt) store_tile(buf ^ 1);
        __syncthreads();
    }

    // epilogue: D[row = n][col = m]; lane holds n = nbase + (lane>>4)*4 + r (r = 0..3), m = lane & 15.
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int m = m0 + wave * 32 + mi * 16 + fr;
        if (m >= g.M) continue;
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            const int n = n0 + ni * 16 + (lane >> 4) * 4;
            if (n >= g.N) continue;
            const f32x4 b = *reinterpret_cast<const f32x4*>(g.bias + n);  // bias is padded to Npad
            float v[4] = {acc[ni][mi].x + b.x, acc[ni][mi].y + b.y, acc[ni][mi].z + b.z, acc[ni][mi].w + b.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], g.act);
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

// Pick BN = 16*NT: minimise padded columns, prefer fewer/larger tiles (fewer A-panel re-reads).
static int pick_nt(int N) {
    static const int opts[] = {2, 3, 4, 5, 6, 8, 9, 12};
    int best = 2;
    double best_cost = 1e30;
    for (int nt : opts) {
        const int bn = nt * 16;
        const int tiles = (N + bn - 1) / bn;
        const double cost = (double)tiles * bn * (1.0 + 24.0 / bn) + 8.0 * tiles;
        if (cost < best_cost) { best_cost = cost; best = nt; }
    }
    return best;
}

template <int NT>
static int launch_nt(const GemmArgs& a, hipStream_t st) {
    const int bn = NT * 16;
    const int n_tiles = cdiv(a.N, bn);
    const int m_tiles = cdiv(a.M, GM_BM);
    const long nwg = (long)n_tiles * m_tiles;
    MI355_REQUIRE(nwg < (1l << 31), "gemm: grid too large");
    hipLaunchKernelGGL((k_gemm_bf16<NT>), dim3((unsigned)nwg), dim3(256), 0, st, a, n_tiles, (int)nwg);
    MI355_LAUNCH_CHECK();
    return OK;
}

int launch_gemm_bf16(const GemmArgs& a, hipStream_t st) {
    MI355_REQUIRE(a.M >= 1 && a.N >= 1 && a.K >= 1, "gemm: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
    MI355_REQUIRE(a.K % 8 == 0 && a.lda % 8 == 0 && a.ldw % 32 == 0, "gemm: K=%d lda=%d ldw=%d alignment", a.K, a.lda,
                  a.ldw);
    MI355_REQUIRE(a.out_f32 || a.ldo % 4 == 0, "gemm: bf16 output stride %d must be a multiple of 4", a.ldo);
    MI355_REQUIRE(!a.res || a.ldr % 4 == 0, "gemm: residual stride %d must be a multiple of 4", a.ldr);
    switch (pick_nt(a.N)) {
        case 2: return launch_nt<2>(a, st);
        case 3: return launch_nt<3>(a, st);
        case 4: return launch_nt<4>(a, st);
        case 5: return launch_nt<5>(a, st);
        case 6: return launch_nt<6>(a, st);
        case 8: return launch_nt<8>(a, st);
        case 9: return launch_nt<9>(a, st);
        default: return launch_nt<12>(a, st);
    }
}

}  // namespace mi355
