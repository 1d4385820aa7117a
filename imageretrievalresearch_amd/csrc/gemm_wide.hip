// Wide-tile bf16 GEMM for the compute-bound linears (Swin's K = 256 ... 4096 layers at M = 6 k ... 100 k rows): out = act(A W^T + b)
// [+ LayerNorm folding, + residual], same contract and the same bits as k_gemm_big (gemm_bf16.hip) - every output element sums
// its k-steps in the same order through the same MFMA (16x16x32 bf16, fp32 accumulate).  gfx950 only.
//
// Why another tile: k_gemm_big's 128 x 128 tile pulls 64 B per clock and CU through the L2 -> LDS path at full MFMA rate, twice
// what that path sustains, and its two-deep ring + full-drain barrier exposed an L2 round trip on every 64-deep step (PMC round 2:
// matrix pipe 25 % busy, waves waiting 45 %).  Here:
//   * one PERSISTENT workgroup per CU (8 waves as 2 x 4, wave tile 16 MI x 64, MI = 5 ... 8) walks over (32 MI) x 256 tiles:
//     32 B/clk per CU at full rate for MI = 8;
//   * operands arrive by LDS-DMA in 32-deep stages through a four-slot ring, always four stages ahead, ACROSS tile boundaries:
//     the next tile's first stages are requested while this tile's last steps and its epilogue run;
//   * one counted `s_waitcnt vmcnt(n)` + one LDS-only barrier per step; n is computed from what this wave has issued since
//     (DMA groups, the previous tile's stores), so neither the ring nor the epilogue's stores are ever drained inside a tile;
//   * A fragments rotate in place (fragment mi of step t+1 is requested as soon as the MFMAs of step t have consumed it), W
//     fragments are double-buffered: no LDS read latency in front of an MFMA, 12 ds_read_b128 per 32 MFMAs;
//   * the epilogue's operands (bias, LayerNorm column sums, per-row (mean, rstd)) also come by LDS-DMA with the tile's first
//     stage; outputs are stored straight from the accumulator layout (4 channels x 16 rows per instruction).
#include "ops.h"

#include <stdlib.h>
#include <type_traits>

namespace mi355 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int GW_BN = 256;        // tile columns
constexpr int GW_BK = 32;         // k depth of a stage (64 bytes per tile row)
constexpr int GW_RING = 4;        // stages in the ring = stages requested ahead

__device__ __forceinline__ float lo_bf16(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi_bf16(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// One LDS-DMA piece: 64 lanes x 16 bytes from base + (32-bit lane offset) to LDS bytes [lds_off, lds_off + 1024), lane-linear.
// Written as asm on purpose: through the builtin hipcc's wait-count pass kept inserting `s_waitcnt vmcnt(0)` between the requests
// of one stage (pending-load state merged conservatively over the persistent loop), i.e. drained the ring on every step.  There is
// no destination register, so the compiler has nothing to track; all waits on these requests are the counted ones in this file.
__device__ __forceinline__ void gw_dma(const void* base, unsigned lane_off, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(lane_off), "s"(base), "s"(lds_off) : "memory");   // (m0 is reserved: hipcc never keeps a value in it across statements)
}
__device__ __forceinline__ unsigned gw_lds_off(const void* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the immediate has 6 bits)
__device__ __forceinline__ void gw_vm_wait(int n) {
    switch (n) {
#define GW_C(i) case i: asm volatile("s_waitcnt vmcnt(" #i ")" ::: "memory"); break;
        GW_C(1) GW_C(2) GW_C(3) GW_C(4) GW_C(5) GW_C(6) GW_C(7) GW_C(8) GW_C(9) GW_C(10) GW_C(11) GW_C(12) GW_C(13) GW_C(14) GW_C(15)
        GW_C(16) GW_C(17) GW_C(18) GW_C(19) GW_C(20) GW_C(21) GW_C(22) GW_C(23) GW_C(24) GW_C(25) GW_C(26) GW_C(27) GW_C(28) GW_C(29)
        GW_C(30) GW_C(31) GW_C(32) GW_C(33) GW_C(34) GW_C(35) GW_C(36) GW_C(37) GW_C(38) GW_C(39) GW_C(40) GW_C(41) GW_C(42) GW_C(43)
        GW_C(44) GW_C(45) GW_C(46) GW_C(47) GW_C(48) GW_C(49) GW_C(50) GW_C(51) GW_C(52) GW_C(53) GW_C(54) GW_C(55) GW_C(56) GW_C(57)
        GW_C(58) GW_C(59) GW_C(60) GW_C(61) GW_C(62) GW_C(63)
#undef GW_C
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

template <int MI>
struct GwCfg {
    static constexpr int BM = 32 * MI;                 // two wave rows of MI x 16
    static constexpr int NPA = BM / 16;                // A pieces (16 rows x 64 B = 1 KB, one wave instruction) per stage
    static constexpr int NPW = GW_BN / 16;             // 16 W pieces per stage
    static constexpr int A_STAGE = BM * GW_BK;         // bf16 elements
    static constexpr int W_STAGE = GW_BN * GW_BK;
    static constexpr int STAGE = A_STAGE + W_STAGE;
    static constexpr int RING_BYTES = GW_RING * STAGE * 2;
    static constexpr int AUX_FLOATS = 256 + 256 + 512;         // bias | LayerNorm column sums | (mean, rstd) of up to 256 rows
    static constexpr int LDS_BYTES = RING_BYTES + 2 * AUX_FLOATS * 4;
    static constexpr int STORES = 4 * MI;              // store instructions of one wave per tile
};

// Developer build only (tools/gemm_wide_probe.hip defines GW_STAMPS): shader-clock cycles of wave 0 per phase, summed over its steps,
// written to g.splitk_ws as 8 x uint64 per workgroup: wait + barrier | DMA issue | MFMA + fragment reads | epilogue | tile top reads
#ifdef GW_STAMPS
#define GW_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define GW_ACC(slot, a, b) st_acc[slot] += (b) - (a)
#else
#define GW_T(var)
#define GW_ACC(slot, a, b)
#endif

template <int MI>
__global__ __launch_bounds__(512) void k_gemm_wide(const GemmArgs g, const int n_tiles, const int total_tiles) {
    using C = GwCfg<MI>;
    extern __shared__ __attribute__((aligned(16))) unsigned char gw_smem[];
    bf16_t* const ring = reinterpret_cast<bf16_t*>(gw_smem);
    float* const aux = reinterpret_cast<float*>(gw_smem + C::RING_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const int row_base = wm * (MI * 16), col_base = wn * 64;
    const int nt = g.K / GW_BK;                                  // k-steps per tile (>= 4: checked by the launcher)
    const int Npad = (g.N + 15) & ~15;

    // ---- this workgroup's tiles: rounds of gridDim.x tiles; inside a round the eight XCDs take contiguous runs of the logical
    // order (column tile fastest), so the workgroups that share an XCD's L2 share A row panels
    const int G = (int)gridDim.x;
    int my_logical;
    {
        const int bid = (int)blockIdx.x;
        const int q = G >> 3, r = G & 7, xcd = bid & 7;
        my_logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int my_tiles = my_logical < total_tiles ? (total_tiles - 1 - my_logical) / G + 1 : 0;
    if (my_tiles == 0) return;
    const int total_steps = my_tiles * nt;

    // ---- DMA side.  Wave w moves A pieces w and w + 8 (where they exist) and W pieces w and w + 8 of every stage; lane l of a piece
    // reads 16 bytes of tile row piece * 16 + l / 4.  The bank swizzle sits on the SOURCE chunk (the DMA writes LDS lane-linearly):
    // physical chunk c of a 64-byte row holds logical chunk c ^ swz(row), swz = -(row / 4) mod 4 - which depends on the lane only.
    const int pa_cnt = 1 + (wave + 8 < C::NPA ? 1 : 0);          // A pieces of this wave per stage
    const int pw_cnt = pa_cnt + 2;                               // pieces per stage
    const int ex_cnt = wave < 4 ? 1 : 0;                         // + one epilogue-operand piece with a tile's first stage
    const int d_chunk = ((lane & 3) ^ ((0 - (lane >> 4)) & 3)) * 8;
    const int d_row = lane >> 2;
    unsigned a_off[2], w_off[2];                                 // byte offsets of this lane's rows from g.A / g.W (the launcher checks < 4 GB)
    int i_tile = 0, i_k = 0, i_slot = 0;                         // what the next issue will request: tile ordinal, k offset, ring slot
    auto set_tile_sources = [&](int ordinal) {
        const int logical = my_logical + ordinal * G;
        const int mb = logical / n_tiles, nb = logical - mb * n_tiles;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ar = min(mb * C::BM + (wave + 8 * i) * 16 + d_row, g.M - 1);
            const int wr = nb * GW_BN + (wave + 8 * i) * 16 + d_row;
            a_off[i] = ((unsigned)ar * (unsigned)g.lda + d_chunk) * 2u;
            w_off[i] = ((unsigned)(wr < Npad ? wr : 0) * (unsigned)g.ldw + d_chunk) * 2u;
        }
    };
    const unsigned ring_off = gw_lds_off(ring), aux_off = gw_lds_off(aux);
    auto issue_stage = [&]() {
        const unsigned dst = ring_off + (unsigned)(i_slot * C::STAGE * 2 + wave * 1024);
        const bf16_t* const ak = g.A + i_k;
        const bf16_t* const wk = g.W + i_k;
        gw_dma(ak, a_off[0], dst);
        if (C::NPA == 16 || pa_cnt == 2) gw_dma(ak, a_off[1], dst + 8 * 1024);   // (MI = 8: no branch between the requests)
        gw_dma(wk, w_off[0], dst + C::A_STAGE * 2);
        gw_dma(wk, w_off[1], dst + C::A_STAGE * 2 + 8 * 1024);
        if (i_k == 0 && ex_cnt) {
            // epilogue operands of this tile (consumed a whole tile later; two buffers by tile parity)
            const int logical = my_logical + i_tile * G;
            const int mb = logical / n_tiles, nb = logical - mb * n_tiles;
            const unsigned ax_off = aux_off + (unsigned)(((i_tile & 1) * C::AUX_FLOATS + wave * 256) * 4);
            const void* src;
            unsigned loff;
            if (wave == 0) {
                const int n = nb * GW_BN + lane * 4;
                src = g.bias;
                loff = (unsigned)(n + 4 <= Npad ? n : 0) * 4u;
            } else if (wave == 1) {
                const int n = nb * GW_BN + lane * 4;
                src = g.ln_colsum ? (const void*)g.ln_colsum : (const void*)g.zeros;
                loff = g.ln_colsum ? (unsigned)(n + 4 <= Npad ? n : 0) * 4u : 0u;
            } else {
                const int m = mb * C::BM + (wave - 2) * 128 + lane * 2;           // two rows of (mean, rstd) per lane
                src = g.ln_stats ? (const void*)g.ln_stats : (const void*)g.zeros;
                loff = g.ln_stats ? (unsigned)min(m, g.M - 2) * 8u : 0u;
            }
            gw_dma(src, loff, ax_off);
        }
        i_k += GW_BK;
        i_slot = (i_slot + 1) & (GW_RING - 1);
        if (i_k == g.K) {
            i_k = 0;
            ++i_tile;
            if (i_tile < my_tiles) set_tile_sources(i_tile);
        }
    };
    // pieces this wave issues for the stage with local k-step index lk
    auto group_size = [&](int lk) { return pw_cnt + (lk == 0 ? ex_cnt : 0); };

    // ---- MFMA side
    const int f_chunk = (fq ^ ((0 - (fr >> 2)) & 3)) * 8;
    const int a_lane = (row_base + fr) * GW_BK + f_chunk;                 // + mi * 512 elements
    const int w_lane = C::A_STAGE + (col_base + fr) * GW_BK + f_chunk;    // + ni * 512
    bf16x8 af[MI], wf[4];
    f32x4 acc[4][MI];

#ifdef GW_STAMPS
    // (probe builds: g.gate_ld = start stagger in units of 64 cycles per (workgroup % 4))
    for (int d = 0; d < ((int)blockIdx.x >> 3 & 3) * g.gate_ld; d += 100) __builtin_amdgcn_s_sleep(100);
#endif
    set_tile_sources(0);
#pragma unroll 1
    for (int s = 0; s < GW_RING; ++s)
        if (s < total_steps) issue_stage();
    {
        // stage 0 has landed once at most the groups behind it are still in flight
        int young = 0;
        for (int s = 1; s < GW_RING; ++s)
            if (s < total_steps) young += group_size(s % nt);
        gw_vm_wait(__builtin_amdgcn_readfirstlane(young));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

#ifdef GW_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    GW_T(t_begin);
#endif
    int u = 0;                 // virtual step = tile ordinal * nt + local step
    int r_slot = 0;            // ring slot of step u
    int e_young = 0;           // store instructions of the previous tile's epilogue that may still be in flight
    for (int it = 0; it < my_tiles; ++it) {
        // fragments of this tile's first step (the stage landed before the previous tile's last step / the prologue's barrier; they
        // are not read during that last step so that the epilogue has the registers)
        GW_T(t_top0);
        asm volatile("" ::: "memory");   // (keeps these reads below the previous tile's epilogue: hoisted above it they cost 48 registers there)
        __builtin_amdgcn_sched_barrier(0);
        {
            const bf16_t* const a0 = ring + r_slot * C::STAGE + a_lane;
            const bf16_t* const w0 = ring + r_slot * C::STAGE + w_lane;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const bf16x8*>(a0 + mi * 512);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) wf[ni] = *reinterpret_cast<const bf16x8*>(w0 + ni * 512);
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
        GW_T(t_top1);
        GW_ACC(4, t_top0, t_top1);

        auto step = [&](auto read_c, int t) {
            constexpr bool READ = decltype(read_c)::value != 0;
            // top of step u: stage u + 1 must have landed (its fragments are read below); everything this wave issued after it may stay
            // in flight: the groups of stages u + 2, u + 3 and, during a tile's first three steps, the previous tile's stores
            GW_T(t_s0);
            int young = 0;
            if (u + 2 < total_steps) young += group_size(t + 2 >= nt ? t + 2 - nt : t + 2);
            if (u + 3 < total_steps) young += group_size(t + 3 >= nt ? t + 3 - nt : t + 3);
            if (t <= 2) young += e_young;
            gw_vm_wait(__builtin_amdgcn_readfirstlane(young));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            GW_T(t_s1);
            // everybody has the fragments of step u in registers: its slot takes stage u + 4
            if (u + GW_RING < total_steps) issue_stage();
            GW_T(t_s2);
            const int n_slot = (r_slot + 1) & (GW_RING - 1);
            const bf16_t* const an = ring + n_slot * C::STAGE + a_lane;
            const bf16_t* const wn_ = ring + n_slot * C::STAGE + w_lane;
            // A fragment mi of the next step is requested as soon as this step's four MFMAs have consumed it; the W fragments follow
            // behind the last row block (the kernel waits on the L2 -> LDS stream, not on these reads)
#ifdef GW_STAMPS
            if (!(g.a_relu6 & 1))      // (probe builds: bit 0 of a_relu6 switches the MFMAs and fragment reads off, bit 1 the epilogue)
#endif
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
#ifdef GW_STAMPS
                    if (!(g.a_relu6 & 8))
#endif
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
#ifdef GW_STAMPS
                    if (!(g.a_relu6 & 4))      // (probe builds: bit 2 = MFMAs on stale fragments, no LDS reads; bit 3 = reads only)
#endif
                    if (READ && mi == MI - 1) wf[ni] = *reinterpret_cast<const bf16x8*>(wn_ + ni * 512);
                }
#ifdef GW_STAMPS
                if (!(g.a_relu6 & 4))
#endif
                if (READ) af[mi] = *reinterpret_cast<const bf16x8*>(an + mi * 512);
            }
            r_slot = n_slot;
            ++u;
            GW_T(t_s3);
            GW_ACC(0, t_s0, t_s1); GW_ACC(1, t_s1, t_s2); GW_ACC(2, t_s2, t_s3);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
#pragma unroll 1
        for (int t = 0; t < nt - 1; ++t) step(I1{}, t);
        step(I0{}, nt - 1);

        // ---- epilogue of tile `it` (the DMA of the next tile's first four stages is in flight behind it)
        GW_T(t_e0);
        // (nothing of the epilogue may be scheduled into the last step: its operand reads there spilled the step's fragments)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const int logical = my_logical + it * G;
        const int mb = logical / n_tiles, nb = logical - mb * n_tiles;
        const int m0 = mb * C::BM, n0 = nb * GW_BN;
        const bool interior = (m0 + C::BM <= g.M) && (n0 + GW_BN <= g.N);
        const float* const ax = aux + (it & 1) * C::AUX_FLOATS;
        const bool has_ln = g.ln_stats != nullptr;
        const bool has_res = g.res != nullptr;
        // addresses as (uniform base) + (one 32-bit lane offset): the mi / ni terms are uniform and stay on the scalar side
        // (the row strides pass through an empty asm per tile: as loop invariants hipcc hoists every multiple of them out of the
        //  tile loop - 70 scalar registers spilled into vector lanes and, behind them, vector spills inside the k loop)
        int ldo_b = g.ldo * 2, ldr_b = g.ldr * 2;
        asm volatile("" : "+s"(ldo_b), "+s"(ldr_b));
        const unsigned res_lane = (unsigned)((row_base + fr) * ldr_b + (col_base + fq * 4) * 2);
        const unsigned out_lane = (unsigned)((row_base + fr) * ldo_b + (col_base + fq * 4) * 2);
        const char* const res_tile = reinterpret_cast<const char*>(g.res) + (size_t)m0 * ldr_b + n0 * 2;
        char* const out_tile = reinterpret_cast<char*>(g.out) + (size_t)m0 * ldo_b + n0 * 2;
        // bias / column sums of this lane's sixteen columns (LDS, landed with the tile's first stage)
        f32x4 bq[4], cq[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            bq[ni] = *reinterpret_cast<const f32x4*>(ax + col_base + ni * 16 + fq * 4);
            cq[ni] = has_ln ? *reinterpret_cast<const f32x4*>(ax + 256 + col_base + ni * 16 + fq * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        u32x2 rr[2][4];
        auto load_res = [&](int mi, int buf) {
            const int m = m0 + row_base + mi * 16 + fr;
            const char* const rbase = res_tile + (size_t)(mi * 16) * ldr_b;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = n0 + col_base + ni * 16 + fq * 4;
                rr[buf][ni] = (interior || (m < g.M && n < g.N)) ? *reinterpret_cast<const u32x2*>(rbase + ni * 32 + res_lane)
                                                                 : (u32x2){0u, 0u};
            }
        };
        if (has_res) load_res(0, 0);
        // every store instruction writes 16 rows x 32 bytes; the four of one mi complete 128-byte lines within a few cycles of each
        // other (the L2 merges them; an LDS transpose to whole-line stores cost two LDS round trips per mi and was slower)
#ifdef GW_STAMPS
        if (!(g.a_relu6 & 2))
#endif
        MI355_ACT_DISPATCH(g.act, {
_Pragma("unroll")
            for (int mi = 0; mi < MI; ++mi) {
                if (has_res && mi + 1 < MI) load_res(mi + 1, (mi + 1) & 1);
                // folded LayerNorm: LN(x) W^T = rstd (x W'^T - mean colsum(W'))
                mi355_f32x2 st = {0.f, 1.f};
                if (has_ln) st = *reinterpret_cast<const mi355_f32x2*>(ax + 512 + (row_base + mi * 16 + fr) * 2);
                const int m = m0 + row_base + mi * 16 + fr;
                char* const obase = out_tile + (size_t)(mi * 16) * ldo_b;
_Pragma("unroll")
                for (int ni = 0; ni < 4; ++ni) {
                    f32x4 v = acc[ni][mi];
                    if (has_ln) {
                        v.x = st.y * (v.x - st.x * cq[ni].x); v.y = st.y * (v.y - st.x * cq[ni].y);
                        v.z = st.y * (v.z - st.x * cq[ni].z); v.w = st.y * (v.w - st.x * cq[ni].w);
                    }
                    v.x = act_c<ACT>(v.x + bq[ni].x); v.y = act_c<ACT>(v.y + bq[ni].y);
                    v.z = act_c<ACT>(v.z + bq[ni].z); v.w = act_c<ACT>(v.w + bq[ni].w);
                    if (has_res) {
                        const u32x2 r2 = rr[mi & 1][ni];
                        v.x += lo_bf16(r2.x); v.y += hi_bf16(r2.x); v.z += lo_bf16(r2.y); v.w += hi_bf16(r2.y);
                    }
                    u32x2 o;
                    o.x = pack2bf(v.x, v.y);
                    o.y = pack2bf(v.z, v.w);
                    const int n = n0 + col_base + ni * 16 + fq * 4;
                    if (interior || (m < g.M && n < g.N)) *reinterpret_cast<u32x2*>(obase + ni * 32 + out_lane) = o;
                }
            }
        })
        if (interior) {
            e_young = C::STORES;
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (a border tile's store count is not known: drain, count nothing)
            e_young = 0;
        }
        GW_T(t_e1);
        GW_ACC(3, t_e0, t_e1);
    }
#ifdef GW_STAMPS
    GW_T(t_end);
    st_acc[5] = t_end - t_begin;
    st_acc[6] = (unsigned long long)my_tiles;
    if (g.splitk_ws && tid == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(g.splitk_ws) + (size_t)blockIdx.x * 8;
        for (int i = 0; i < 8; ++i) o[i] = st_acc[i];
    }
#endif
}

template <int MI>
int launch_wide_mi(const GemmArgs& a, int grid, hipStream_t st) {
    using C = GwCfg<MI>;
    static bool attr_done[MI355_MAX_DEVICES] = {false};
    if (first_time_on_this_device(attr_done))
        MI355_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemm_wide<MI>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    const int n_tiles = cdiv(a.N, GW_BN), m_tiles = cdiv(a.M, C::BM);
    const int total = n_tiles * m_tiles;
    hipLaunchKernelGGL((k_gemm_wide<MI>), dim3((unsigned)min(grid, total)), dim3(512), C::LDS_BYTES, st, a, n_tiles, total);
    MI355_LAUNCH_CHECK();
    return OK;
}

}  // namespace

// The shapes this kernel takes (everything else stays with launch_gemm_bf16's other kernels)
bool gemm_wide_supported(const GemmArgs& a) {
    return a.zeros && a.K >= 128 && a.K % 32 == 0 && a.ldw % 32 == 0 && a.ldw >= a.K && a.lda % 8 == 0 && ((uintptr_t)a.A % 16 == 0) &&
           a.N >= 256 && a.N % 8 == 0 && a.M >= 4096 && a.M % 2 == 0 && (size_t)a.M * a.lda * 2 < ((size_t)1 << 32) &&
           (size_t)((a.N + 15) & ~15) * a.ldw * 2 < ((size_t)1 << 32) && (size_t)a.M * 8 < ((size_t)1 << 32) && !a.gate &&
#ifndef GW_STAMPS
           !a.a_relu6 &&
#endif
           !a.out_f32 && a.ldo % 4 == 0 &&
           (!a.res || (a.res_n >= a.N && a.ldr % 4 == 0)) && (!a.ln_stats || a.ln_colsum);
}

// rows per tile: the MI whose ceil(tiles / CUs) rounds x bytes-per-tile-step is smallest (all tiles of a launch cost the same, so a
// partly filled last round costs a whole one); ties go to the taller tile
int gemm_wide_pick_mi(int M, int N, int cus) {
    static const int forced = getenv("MI355_GEMM_WIDE_MI") ? atoi(getenv("MI355_GEMM_WIDE_MI")) : 0;
    if (forced >= 5 && forced <= 8) return forced;
    int best = 8;
    long best_cost = -1;
    for (int mi = 8; mi >= 5; --mi) {
        const long tiles = (long)cdiv(M, 32 * mi) * cdiv(N, GW_BN);
        const long rounds = (tiles + cus - 1) / cus;
        // a tile's time is its operand bytes (the L2 -> LDS path is the bound: tools/dma_probe.hip, ~53 GB/s per CU): rows + columns
        const long cost = rounds * (32 * mi + GW_BN);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = mi; }
    }
    return best;
}

int launch_gemm_wide(const GemmArgs& a, hipStream_t st) {
    MI355_REQUIRE(gemm_wide_supported(a), "gemm_wide: unsupported shape M=%d N=%d K=%d", a.M, a.N, a.K);
    static int cus[MI355_MAX_DEVICES] = {0};
    int dev = 0;
    MI355_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= MI355_MAX_DEVICES) dev = 0;
    if (!cus[dev]) {
        hipDeviceProp_t p;
        MI355_CHECK_HIP(hipGetDeviceProperties(&p, dev));
        cus[dev] = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }
    switch (gemm_wide_pick_mi(a.M, a.N, cus[dev])) {
        case 5: return launch_wide_mi<5>(a, cus[dev], st);
        case 6: return launch_wide_mi<6>(a, cus[dev], st);
        case 7: return launch_wide_mi<7>(a, cus[dev], st);
        default: return launch_wide_mi<8>(a, cus[dev], st);
    }
}

}  // namespace mi355
