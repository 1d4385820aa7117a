// Rank half of the hot path: row normalisation, cosine GEMM with fused top-k selection, shard merge, pair cosine,
// ContrastiveLoss, hit counting.  gfx950 only.
//
// Reference semantics: torch.nn.CosineSimilarity(dim=1, eps=1e-6) + torch.topk as called at
// train/train.py:250-251 (see include/mi355_retrieval.h).  Inputs, norms, accumulation and scores are fp32.  The
// GEMM has two loops with the same tiling and epilogue:
//   * default: fp32 operands split into three bf16 planes, six of the nine bf16 products per element on
//     v_mfma_f32_32x32x16_bf16 with fp32 accumulation (k_cos_gemm_split; error per product <= 2^-23, the size of the
//     one rounding an fmaf spends - measured as close to the float64 cosine as the exact loop, ~1e-7);
//   * MI355_RANK_EXACT_F32=1, gallery rows not 16-byte aligned, or Q <= 4 (GEMV): v_mfma_f32_32x32x2_f32 / fmaf, bit-for-bit
//     an fp32 fmaf chain (2.7x the matrix-pipe time).
// Either way an index can only differ from the CPU oracle where two scores are closer than fp32 summation noise.
#include "common.h"
#include "../../include/mi355_retrieval.h"

#include <limits.h>
#include <stdlib.h>
#include <math.h>

namespace mi355 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef long long i64;

// =====================================================================================
// row norms
// =====================================================================================
// One wave per row; float4 loads when dim % 4 == 0 and rows are 16-B aligned.
template <bool WRITE_ROWS>
__global__ __launch_bounds__(256) void k_row_norm(const float* __restrict__ in, float* __restrict__ out,
                                                  float* __restrict__ inv, i64 rows, int dim, float eps,
                                                  int vec) {
    const int lane = threadIdx.x & 63;
    const i64 row = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* x = in + row * dim;
    float ss = 0.f;
    if (vec) {
        const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
        for (int i = lane; i < dim / 4; i += 64) {
            f32x4 v = x4[i];
            ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
    } else {
        for (int i = lane; i < dim; i += 64) ss += x[i] * x[i];
    }
    ss = wave_sum(ss);
    const float r = 1.0f / fmaxf(sqrtf(ss), eps);
    if (inv != nullptr && lane == 0) inv[row] = r;
    if (WRITE_ROWS) {
        float* y = out + row * dim;
        if (vec) {
            const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
            f32x4* y4 = reinterpret_cast<f32x4*>(y);
            for (int i = lane; i < dim / 4; i += 64) {
                f32x4 v = x4[i];
                v.x *= r; v.y *= r; v.z *= r; v.w *= r;
                y4[i] = v;
            }
        } else {
            for (int i = lane; i < dim; i += 64) y[i] = x[i] * r;
        }
    }
}

static inline int vec_ok(const void* p, int dim) { return (dim % 4 == 0) && (((uintptr_t)p & 15) == 0); }

// =====================================================================================
// top-k selection
// =====================================================================================
// Ordering of (score, index) candidates: higher score first, ties -> lower index.  NaN orders as the LARGEST value, as in
// torch.topk (a NaN score - e.g. from an Inf/NaN embedding - is returned with its in-range index instead of starving
// the list and leaving pad entries behind); two NaNs tie.
__device__ __forceinline__ bool better(float a, i64 ia, float b, i64 ib) {
    const bool an = a != a, bn = b != b;
    if (an || bn) return (an && !bn) || (an && bn && ia < ib);
    return (a > b) || (a == b && ia < ib);
}

constexpr float NEG_INF = -INFINITY;
constexpr i64 IDX_PAD = LLONG_MAX;

constexpr int IDX32_PAD = INT_MAX;   // missing candidate in the fused per-tile lists (local int32 indices)

// Order-preserving map float -> uint32 for the fused selection: larger key = better score.  NaN maps to the largest key
// (torch.topk's order), -0 to the key of +0 (they compare equal as floats), every real score to a key > 0.
__device__ __forceinline__ unsigned score_key(float x) {
    const unsigned u = __float_as_uint(x + 0.0f);                  // -0 -> +0
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;       // NaN
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_score(unsigned key) {
    if (key == 0xffffffffu) return __uint_as_float(0x7fc00000u);   // canonical NaN
    return __uint_as_float((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key);
}

// =====================================================================================
// fp32 operands on the bf16 matrix pipe: three-way split, six products
// =====================================================================================
// x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (round-to-nearest-even; both subtractions are exact
// in fp32, and |x - h - m - l| <= 2^-25 |x|: the three 8-bit significands cover fp32's 24).  A product x*y is then the sum
// of nine bf16 x bf16 products, each EXACT in the fp32 accumulator; the kernel keeps the six largest
//     h*h' + (h*m' + m*h') + (h*l' + l*h' + m*m')
// and drops m*l', l*m', l*l' (<= 2^-24 |x*y| each, either sign).  Per product that is the size of ONE fp32 rounding -
// what the exact-fp32 MFMA (an fmaf chain) spends on every product anyway - so scores agree with the fp32 chain to
// ~1e-7 (measured against the fp64 oracle in tests/test_rank_gpu.py); the accumulation itself stays fp32.  Six
// v_mfma_f32_32x32x16_bf16 do the work of eight v_mfma_f32_32x32x2_f32 in 3/8 of the matrix-pipe time (192 vs 512 cycles).
// NaN / Inf inputs give NaN scores (Inf - Inf in the split), which the selection orders as torch.topk does.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = pack2bf(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = pack2bf(r0, r1);
    l = pack2bf(r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xffff0000u));
}
__device__ __forceinline__ void split3(const f32x4 a, const f32x4 b, u32x4& h, u32x4& m, u32x4& l) {
    unsigned hh[4], mm[4], ll[4];
    split3_pair(a.x, a.y, hh[0], mm[0], ll[0]);
    split3_pair(a.z, a.w, hh[1], mm[1], ll[1]);
    split3_pair(b.x, b.y, hh[2], mm[2], ll[2]);
    split3_pair(b.z, b.w, hh[3], mm[3], ll[3]);
    h = (u32x4){hh[0], hh[1], hh[2], hh[3]};
    m = (u32x4){mm[0], mm[1], mm[2], mm[3]};
    l = (u32x4){ll[0], ll[1], ll[2], ll[3]};
}

// Normalised queries -> split planes in MFMA-fragment order: Qs[row block of 32][k step of 16][plane h,m,l][lane][8 bf16],
// lane = row + 32 * (k half): every (row block, k step, plane) is 1 KB that one global_load_lds moves into LDS exactly as
// the A operand of v_mfma_f32_32x32x16_bf16 wants it (lane-linear ds_read_b128, no padding, no swizzle).  Rows >= Q and
// k >= D are zero.  A 256-byte zero page follows the planes (source of the GEMM's out-of-range gallery loads).
__global__ __launch_bounds__(256) void k_split_queries(const float* __restrict__ Qn, bf16_t* __restrict__ Qs, int Q, int D,
                                                       int n_steps, int n_frag) {
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6);   // fragment = (row block, k step)
    if (blockIdx.x == 0 && threadIdx.x < 64) reinterpret_cast<unsigned*>(Qs + (size_t)n_frag * 3 * 512)[threadIdx.x] = 0u;   // zero page
    if (f >= n_frag) return;
    const int lane = threadIdx.x & 63;
    const int rb = f / n_steps, s = f % n_steps;
    const int row = rb * 32 + (lane & 31), k0 = s * 16 + (lane >> 5) * 8;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (row < Q && k0 + e < D) ? Qn[(i64)row * D + k0 + e] : 0.f;
    u32x4 h, m, l;
    split3((f32x4){x[0], x[1], x[2], x[3]}, (f32x4){x[4], x[5], x[6], x[7]}, h, m, l);
    u32x4* o = reinterpret_cast<u32x4*>(Qs + (size_t)f * 3 * 512) + lane;
    o[0] = h; o[64] = m; o[128] = l;
}

__device__ __forceinline__ void glds16(const bf16_t* gsrc, bf16_t* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// =====================================================================================
// cosine GEMM:  S[q][g] = sum_d Qn[q][d] * Gal[g][d] * (ginv ? ginv[g] : 1)
// =====================================================================================
// Block = 4 waves as 2(M) x 2(N); wave tile = (MT*32) queries x 64 gallery rows; block tile =
// (64*MT) x 128 x BK.  Both operands are "row x k-contiguous", so they share one LDS image:
// [rows][BK + 4] floats (the 4-float pad makes ds_read_b128 of 16 distinct rows bank-conflict free).
// Per lane a float4 at k = 8t + 4*(lane>>5) feeds four 32x32x2 k-steps; A and B use the same k
// permutation, which only reorders the (exact) fma chain.
constexpr int RK_BN = 128;

// Launch order of the GEMM tiles (speed only, every result is the same for any order): the grid is one-dimensional and the
// query blocks of ONE gallery tile get the linear ids L, L + 8, L + 16, ...  The dispatcher deals workgroups round-robin
// over the 8 XCDs, so those workgroups share an L2 and start in the same round: the gallery tile comes from HBM once and
// the other query blocks hit it in L2.  (With a (tile, query block) grid, x fastest, all tiles of query block 0 filled the
// machine before query block 1 started: PMC showed every gallery row fetched from HBM once per query block.)
__device__ __forceinline__ void rank_tile_of(int L, int ntiles, int ny, int& tx, int& ty) {
    const int full = (ntiles >> 3) << 3;
    if (L < full * ny) {
        const int g = L / (8 * ny), r = L - g * 8 * ny;
        tx = g * 8 + (r & 7);
        ty = r >> 3;
    } else {
        const int r = L - full * ny, rem = ntiles - full;
        ty = r / rem;
        tx = full + r - ty * rem;
    }
}

// Epilogue shared by the exact-fp32 and the split-bf16 loops (same accumulator layout: the C/D map of the 32x32 MFMAs does
// not depend on the input type): FK = 0 writes the score slab, FK > 0 selects per-tile candidates.  Called after a
// __syncthreads() that retired every read of the staging buffers (smem is reused).
template <int MT, int FK>
__device__ __forceinline__ void cos_gemm_epilogue(f32x16 (&acc)[MT][2], float* smem, const float* __restrict__ ginv,
                                                  float* __restrict__ S, int Q, i64 G, int k, float* __restrict__ cand_val,
                                                  int* __restrict__ cand_idx, int x0, int ntx, i64 n0, int m0) {
    constexpr int BM = 64 * MT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 31;
    if constexpr (FK > 0) {
        // (the loop's last __syncthreads() retired every read of the staging buffers)
        // 64 query rows at a time, so that the transposed tile (64 x 132 floats = 33.8 KB) fits inside the staging
        // buffers: a bigger LDS request would cost the third resident workgroup per CU and with it a round of tiles
        constexpr int CLD = RK_BN + 4;                 // 132 floats: a thread per row reads float4s conflict-free
        float* Ct = smem;                              // [64][CLD]
        const i64 ncol = (G - n0 < RK_BN) ? G - n0 : RK_BN;     // valid columns of this tile
#pragma unroll 1
        for (int h = 0; h < BM / 64; ++h) {
            if ((wm * MT * 32) / 64 == h) {            // this wave's rows belong to pass h
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const i64 col = n0 + wn * 64 + j * 32 + lr;
                    const float gs = (ginv && col < G) ? ginv[col] : 1.0f;
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = (wm * MT * 32) % 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                            Ct[row * CLD + wn * 64 + j * 32 + lr] = acc[i][j][r] * gs;
                        }
                }
            }
            __syncthreads();
            // Selection: FOUR threads per query row, each scans 32 columns in ascending order into a sorted FK-list held as
            // order-preserving integer keys (NaN = largest, -0 = +0); the insertion is branch-free (a divergent insertion
            // sort cost 10 % of the tile: some lane of the wave inserts at almost every column) and skipped by a wave vote
            // when no lane beats its FK-th entry.  The row's four lists are merged through shuffles in column order, so
            // ties keep resolving to the lower index.
            {
                const int lrow = tid >> 2, part = tid & 3;
                unsigned kv[FK];
                int ki[FK];
#pragma unroll
                for (int i = 0; i < FK; ++i) { kv[i] = 0u; ki[i] = IDX32_PAD; }      // key 0 = below every real score (-inf is 0x007fffff)
                auto insert = [&](unsigned key, int id) {
                    bool g[FK];
#pragma unroll
                    for (int i = 0; i < FK; ++i) g[i] = key > kv[i];          // strict: an equal score keeps the earlier (lower) index
#pragma unroll
                    for (int i = FK - 1; i > 0; --i) {
                        kv[i] = g[i] ? (g[i - 1] ? kv[i - 1] : key) : kv[i];
                        ki[i] = g[i] ? (g[i - 1] ? ki[i - 1] : id) : ki[i];
                    }
                    kv[0] = g[0] ? key : kv[0];
                    ki[0] = g[0] ? id : ki[0];
                };
                const float* rowp = Ct + lrow * CLD + part * 32;
#pragma unroll 2
                for (int c4i = 0; c4i < 8; ++c4i) {
                    const f32x4 v4 = *reinterpret_cast<const f32x4*>(rowp + c4i * 4);
                    const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = part * 32 + c4i * 4 + e;
                        unsigned key = score_key(vv[e]);
                        if (c >= ncol) key = 0u;
                        if (__any(key > kv[FK - 1])) insert(key, (int)n0 + c);      // n0 + c < 2^31 (checked on the host)
                    }
                }
                // merge parts 1..3 into part 0 (lanes 4r .. 4r+3 of one wave)
#pragma unroll
                for (int src = 1; src < 4; ++src) {
#pragma unroll
                    for (int i = 0; i < FK; ++i) {
                        const unsigned ok = (unsigned)__shfl(kv[i], (lane & ~3) + src, 64);
                        const int oi = __shfl(ki[i], (lane & ~3) + src, 64);
                        if (part == 0) insert(ok, oi);
                    }
                }
                const int qrow = m0 + h * 64 + lrow;
                if (part == 0 && qrow < Q) {
                    const size_t o = ((size_t)qrow * ntx + (size_t)(n0 / RK_BN)) * k;
#pragma unroll
                    for (int i = 0; i < FK; ++i)
                        if (i < k) { cand_val[o + i] = ki[i] == IDX32_PAD ? NEG_INF : key_score(kv[i]); cand_idx[o + i] = ki[i]; }
                }
            }
            __syncthreads();
        }
        return;
    }
    // epilogue: C[row = query][col = gallery]; lane: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const i64 col = n0 + wn * 64 + j * 32 + lr;
        if (col >= G) continue;
        const float gs = ginv ? ginv[col] : 1.0f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * MT * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < Q) S[(i64)row * G + col] = acc[i][j][r] * gs;
            }
        }
    }
}

// FK = 0: write the score slab S.  FK = 1/2/4/8 (fused selection, k <= FK): the score tile never leaves the CU - it is
// transposed through LDS (the staging buffers are free after the K loop), each of the tile's query rows is scanned by one
// thread in ascending column order into a sorted FK-list, and k (score, local int32 index) candidates per (query, column
// tile) go to cand_val / cand_idx [Q][n_tiles][k]; the existing multi-level selection then merges Q x n_tiles x k
// candidates instead of reading Q x G scores (train/train.py:250-251 semantics, same tie rule).
template <int MT, int RK_BK, bool VEC, int FK>
__global__ __launch_bounds__(256) void k_cos_gemm(const float* __restrict__ Qn, const float* __restrict__ Gal,
                                                  const float* __restrict__ ginv, float* __restrict__ S,
                                                  int Q, i64 G, int D, int k, float* __restrict__ cand_val,
                                                  int* __restrict__ cand_idx, int x0, int ntx, int xtiles, int ny) {
    // x0 / ntx: this launch covers the column tiles [x0, x0 + xtiles) of ntx for ny query blocks (the host splits a call into a main launch
    // of whole rounds and a tail launch of smaller tiles)
    constexpr int BM = 64 * MT;
    constexpr int RK_LD = RK_BK + 4;      // +4 floats: ds_read_b128 of 16 distinct rows is bank-conflict free (36 and 20)
    constexpr int CPR = RK_BK / 4;        // float4 columns per row of a K-tile
    constexpr int RPP = 256 / CPR;        // rows covered by one pass of the 256 threads
    constexpr int A_LOADS = BM / RPP;     // float4 loads per thread per K-tile for A
    constexpr int B_LOADS = RK_BN / RPP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                          // [2][BM][RK_LD]
    float* Bs = smem + 2 * BM * RK_LD;         // [2][RK_BN][RK_LD]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bx, by;
    rank_tile_of((int)blockIdx.x, xtiles, ny, bx, by);
    const i64 n0 = (i64)(bx + x0) * RK_BN;
    const int m0 = by * BM;

    const int c4 = tid % CPR;  // float4 column within the K-tile
    const int r0 = tid / CPR;

    f32x16 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    f32x4 ra[A_LOADS], rb[B_LOADS];

    auto load_tile = [&](int k0) {
        const int k = k0 + c4 * 4;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int row = m0 + r0 + RPP * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < Q) {
                const float* p = Qn + (i64)row * D + k;
                if (VEC) {
                    if (k + 3 < D) v = *reinterpret_cast<const f32x4*>(p);
                } else {
                    if (k + 0 < D) v.x = p[0];
                    if (k + 1 < D) v.y = p[1];
                    if (k + 2 < D) v.z = p[2];
                    if (k + 3 < D) v.w = p[3];
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const i64 row = n0 + r0 + RPP * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < G) {
                const float* p = Gal + row * D + k;
                if (VEC) {
                    if (k + 3 < D) v = *reinterpret_cast<const f32x4*>(p);
                } else {
                    if (k + 0 < D) v.x = p[0];
                    if (k + 1 < D) v.y = p[1];
                    if (k + 2 < D) v.z = p[2];
                    if (k + 3 < D) v.w = p[3];
                }
            }
            rb[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
        float* a = As + buf * BM * RK_LD;
        float* b = Bs + buf * RK_BN * RK_LD;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i)
            *reinterpret_cast<f32x4*>(a + (r0 + RPP * i) * RK_LD + c4 * 4) = ra[i];
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i)
            *reinterpret_cast<f32x4*>(b + (r0 + RPP * i) * RK_LD + c4 * 4) = rb[i];
    };

    const int nt = (D + RK_BK - 1) / RK_BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int lr = lane & 31;
    const int lk = (lane >> 5) * 4;
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) load_tile((t + 1) * RK_BK);
        const float* a = As + buf * BM * RK_LD + (wm * MT * 32 + lr) * RK_LD + lk;
        const float* b = Bs + buf * RK_BN * RK_LD + (wn * 64 + lr) * RK_LD + lk;
#pragma unroll
        for (int t8 = 0; t8 < RK_BK / 8; ++t8) {
            f32x4 af[MT], bfr[2];
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const f32x4*>(a + i * 32 * RK_LD + t8 * 8);
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[j] = *reinterpret_cast<const f32x4*>(b + j * 32 * RK_LD + t8 * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bfr[j][e], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < nt) store_tile(buf ^ 1);
        __syncthreads();
    }

    cos_gemm_epilogue<MT, FK>(acc, smem, ginv, S, Q, G, k, cand_val, cand_idx, x0, ntx, n0, m0);
}

// =====================================================================================
// The same GEMM on the bf16 matrix pipe (three-way split, six products; see split3 above).  Same block / wave tiling and
// the same epilogue as k_cos_gemm; BK = 16 (one 32x32x16 k-step per K-tile).  Needs 16-byte aligned gallery rows
// (D % 4 == 0); other shapes stay on the fp32 loop.
//   A (queries): pre-split planes in fragment order (k_split_queries), 1 KB per (row block, plane) by LDS-DMA, one
//     k-step ahead (they come from L2);
//   B (gallery): fp32 rows by LDS-DMA as well, TWO k-steps ahead (they come from HBM): a piece is 16 rows x 64 B, the
//     16-byte chunk c of row r lands at position c ^ ((r >> 2) & 3) (the swizzle is applied to the per-lane SOURCE address,
//     the DMA writes lane-linear), so that the ds_read_b128 of 8 rows hit 8 different bank groups without padding.  Each
//     wave reads its two 32-row fragments (8 consecutive k per lane) and splits them in registers - 44 VALU instructions
//     per fragment next to the 12 MFMAs (384 matrix-pipe cycles) that consume it.
// No load in the loop has a register destination, so the only waits are the ones written here: ONE vmcnt(2) per k-step
// (the counter retires in order: everything but this iteration's two B pieces - issued last - has landed) and one
// LDS-only barrier.  (With register-staged B loads hipcc drained vmcnt(0) before every store to LDS.)
// LDS: 2 x 12 KB (A) + 3 x 8 KB (B) = 48 KB at MT = 2 -> three workgroups per CU; 3 x 6 + 24 = 42 KB at MT = 1.
// =====================================================================================
template <int MT, int FK>
__global__ __launch_bounds__(256, 3) void k_cos_gemm_split(const bf16_t* __restrict__ Qs, const float* __restrict__ Gal,
                                                           const float* __restrict__ ginv, float* __restrict__ S, int Q,
                                                           i64 G, int D, int k, float* __restrict__ cand_val,
                                                           int* __restrict__ cand_idx, int x0, int ntx, int n_steps,
                                                           const float* __restrict__ zeros, int xtiles, int ny) {
    constexpr int BM = 64 * MT;
    constexpr int BK = 16;
    constexpr int A_STAGE = (BM / 32) * 3 * 512;      // bf16 elements per stage
    constexpr int A_PIECES = (BM / 32) * 3;           // 1 KB pieces per stage
    constexpr int B_STAGE = RK_BN * BK;               // floats per stage (8 KB)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    bf16_t* As = reinterpret_cast<bf16_t*>(smem);                       // [A_RING][BM/32][3][512]
    // A ring: 2 stages at MT = 2 (the pieces come from L2 one k-step ahead; a third stage would cost the third workgroup per
    // CU), 3 stages at MT = 1 (two k-steps ahead: the 64-row tiles are the tail launch and the small-Q shapes, few
    // workgroups per CU with nothing else to hide a piece's latency behind)
    constexpr int A_RING = MT == 1 ? 3 : 2;
    float* Bs = smem + (A_RING * A_STAGE * 2) / 4;                      // [3][128][16], chunks swizzled

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bx, by;
    rank_tile_of((int)blockIdx.x, xtiles, ny, bx, by);
    const i64 n0 = (i64)(bx + x0) * RK_BN;
    const int m0 = by * BM;
    // the wave index as a scalar: piece selection becomes scalar branches (a per-lane branch around a load makes hipcc
    // drain vmcnt)
    const int swave = __builtin_amdgcn_readfirstlane(wave);

    // B: wave w moves pieces 2w and 2w + 1 (rows 32w .. 32w + 31); lane -> row 16 * piece + lane / 4, position lane % 4
    const float* b_row[2];
    int b_k[2];
    bool b_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (swave * 2 + i) * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((r >> 2) & 3);
        b_ok[i] = n0 + r < G;
        b_row[i] = Gal + (b_ok[i] ? (n0 + r) * D : 0);
        b_k[i] = c * 4;
    }
    auto dma_b = [&](int stage, int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool ok = b_ok[i] && k0 + b_k[i] < D;                // D % 4 == 0: a chunk is inside or outside
            glds16(reinterpret_cast<const bf16_t*>(ok ? b_row[i] + k0 + b_k[i] : zeros),
                   reinterpret_cast<bf16_t*>(Bs + stage * B_STAGE + (swave * 2 + i) * 256));
        }
    };
    // A: piece (row block rbl, plane p) of k-step t sits at Qs + (((m0/32 + rbl) * n_steps + t) * 3 + p) * 512.
    // 12 (MT = 2) or 6 (MT = 1) pieces per stage: wave w moves pieces w, w + 4, w + 8 / pieces w and (w < 2) w + 4.
    const bf16_t* a_src = Qs + (size_t)(m0 / 32) * n_steps * 3 * 512 + lane * 8;
    const bf16_t* a_piece[(A_PIECES + 3) / 4];
#pragma unroll
    for (int i = 0; i < (A_PIECES + 3) / 4; ++i) {
        const int piece = (swave + 4 * i) % A_PIECES;
        a_piece[i] = a_src + ((size_t)((piece / 3) * n_steps) * 3 + piece % 3) * 512;
    }
    auto dma_a = [&](int buf, int t) {
#pragma unroll
        for (int i = 0; i < (A_PIECES + 3) / 4; ++i) {
            const int piece = swave + 4 * i;
            if (A_PIECES % 4 == 0 || i < A_PIECES / 4 || swave < A_PIECES % 4)
                glds16(a_piece[i] + (size_t)t * 3 * 512, As + buf * A_STAGE + piece * 512);
        }
    };

    f32x16 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int lr = lane & 31;
    // B fragment reads: row r = wn * 64 + j * 32 + lr, chunks 2 * (lane >> 5) and + 1 at their swizzled positions
    int b_off[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = wn * 64 + j * 32 + lr, sw = (r >> 2) & 3, c0 = (lane >> 5) * 2;
        b_off[j][0] = r * BK + ((c0 ^ sw) << 2);
        b_off[j][1] = r * BK + (((c0 + 1) ^ sw) << 2);
    }
    auto compute = [&](int abuf, int bstage) {
        const bf16_t* a = As + abuf * A_STAGE + (wm * MT * 3) * 512 + lane * 8;
        const float* b = Bs + bstage * B_STAGE;
        bf16x8 af[MT][3];
        u32x4 bh[2], bm[2], bl[2];
        f32x4 v0[2], v1[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            v0[j] = *reinterpret_cast<const f32x4*>(b + b_off[j][0]);
            v1[j] = *reinterpret_cast<const f32x4*>(b + b_off[j][1]);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) af[i][p] = *reinterpret_cast<const bf16x8*>(a + (i * 3 + p) * 512);
        split3(v0[0], v1[0], bh[0], bm[0], bl[0]);
        // Per fragment j: six products for each of the MT row blocks, smallest terms first (the order is the same for every
        // (query, gallery row) pair wherever its tile lies).  The split of fragment 1 is issued in the gaps of fragment 0's
        // MFMAs (an MFMA holds the vector issue for 8 of its 32 cycles): sched_group_barrier pins "1 MFMA, 4 VALU" groups.
        auto products = [&](int j) {
            const bf16x8 gh = *reinterpret_cast<const bf16x8*>(&bh[j]);
            const bf16x8 gm = *reinterpret_cast<const bf16x8*>(&bm[j]);
            const bf16x8 gl = *reinterpret_cast<const bf16x8*>(&bl[j]);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], gh, acc[i][j], 0, 0, 0);   // l * h'
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], gl, acc[i][j], 0, 0, 0);   // h * l'
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], gm, acc[i][j], 0, 0, 0);   // m * m'
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], gh, acc[i][j], 0, 0, 0);   // m * h'
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], gm, acc[i][j], 0, 0, 0);   // h * m'
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], gh, acc[i][j], 0, 0, 0);   // h * h'
            }
        };
        split3(v0[1], v1[1], bh[1], bm[1], bl[1]);
        products(0);
#pragma unroll
        for (int g = 0; g < 6 * MT; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // four VALU (of fragment 1's split)
        }
        products(1);
    };

    dma_a(0, 0);
    if (A_RING == 3 && n_steps > 1) dma_a(1, 1);
    dma_b(0, 0);
    dma_b(1, BK);                      // (zeros past D)
    __syncthreads();                   // drains vmcnt: everything has landed

    int bs_cur = 0, bs_far = 2;        // B stage of k-step t / of k-step t + 2 (and, at A_RING == 3, the A stages)
    for (int t = 0; t < n_steps; ++t) {
        if constexpr (A_RING == 2) {
            if (t + 1 < n_steps) dma_a((t & 1) ^ 1, t + 1);  // everybody left these buffers at the previous barrier
        } else {
            if (t + 2 < n_steps) dma_a(bs_far, t + 2);
        }
        __builtin_amdgcn_sched_barrier(0);                   // (the counts below need the A pieces issued BEFORE the B pieces)
        dma_b(bs_far, (t + 2) * BK);
        __builtin_amdgcn_sched_barrier(0);
        compute(A_RING == 2 ? (t & 1) : bs_cur, bs_cur);
        if constexpr (A_RING == 2) {
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); // A(t+1) and B(t+1) have landed; B(t+2) stays in flight
        } else {
            // A(t+2) (two pieces from waves 0 and 1, one from waves 2 and 3; none at the end) and B(t+2) stay in flight
            if (t + 2 >= n_steps) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (swave < A_PIECES % 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        bs_cur = bs_cur == 2 ? 0 : bs_cur + 1;
        bs_far = bs_far == 2 ? 0 : bs_far + 1;
    }
    __syncthreads();                   // the last look-ahead pieces (zeros) have landed before the epilogue reuses the LDS
    cos_gemm_epilogue<MT, FK>(acc, smem, ginv, S, Q, G, k, cand_val, cand_idx, x0, ntx, n0, m0);
}

// =====================================================================================
// The same GEMM against a PREPARED gallery (mi355_gallery_prepare): the resident gallery holds, beside nothing else, the three
// bf16 planes of its normalised rows in the fragment order of k_split_queries ([row block of 32][k step][h, m, l][64 lanes][8],
// 6 B per element instead of 4).  Both operands then arrive by LDS-DMA as ready MFMA fragments: no split in the loop (round 2's
// PMC: matrix pipe 57 % busy + VALU 45 % busy, the 88 VALU instructions per fragment split did not co-issue with the MFMAs).
// Per k-step a workgroup moves 12 A pieces (one k-step ahead, from L2) and 12 B pieces (two ahead, from HBM), three + three per
// wave; ONE counted vmcnt (the three youngest = this iteration's B pieces stay in flight) + one LDS-only barrier per k-step.
// Scores are bit-identical to k_cos_gemm_split: the planes are the same values (split3 of the same fp32 rows) and the six
// products are accumulated in the same order.  LDS: 2 x 12 KB (A) + 3 x 12 KB (B) = 60 KB at MT = 2 (two workgroups per CU).
// =====================================================================================
template <int MT, int FK>
__global__ __launch_bounds__(256, 2) void k_cos_gemm_pre(const bf16_t* __restrict__ Qs, const bf16_t* __restrict__ Gs,
                                                         float* __restrict__ S, int Q, i64 G, int k,
                                                         float* __restrict__ cand_val, int* __restrict__ cand_idx, int x0,
                                                         int ntx, int n_steps, int xtiles, int ny) {
    constexpr int BM = 64 * MT;
    constexpr int A_PIECES = (BM / 32) * 3;           // 1 KB pieces per stage
    constexpr int A_STAGE = A_PIECES * 512;           // bf16 elements per stage
    constexpr int B_PIECES = (RK_BN / 32) * 3;        // 12
    constexpr int B_STAGE = B_PIECES * 512;
    constexpr int A_RING = MT == 1 ? 3 : 2;           // (MT = 1: the tail launch / small-Q shapes, nothing else hides a piece's latency)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    bf16_t* As = reinterpret_cast<bf16_t*>(smem);                       // [A_RING][BM/32][3][512]
    bf16_t* Bs = As + A_RING * A_STAGE;                                 // [3][4][3][512]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bx, by;
    rank_tile_of((int)blockIdx.x, xtiles, ny, bx, by);
    const i64 n0 = (i64)(bx + x0) * RK_BN;
    const int m0 = by * BM;
    const int swave = __builtin_amdgcn_readfirstlane(wave);

    // piece (row block rbl, plane p) of k-step t sits at base + (((row0/32 + rbl) * n_steps + t) * 3 + p) * 512; wave w moves
    // pieces w, w + 4, w + 8 of a 12-piece stage (MT = 1, A: pieces w and (w < 2) w + 4)
    const bf16_t* a_src = Qs + (size_t)(m0 / 32) * n_steps * 3 * 512 + lane * 8;
    const bf16_t* b_src = Gs + (size_t)(n0 / 32) * n_steps * 3 * 512 + lane * 8;
    const bf16_t* a_piece[(A_PIECES + 3) / 4];
    const bf16_t* b_piece[3];
#pragma unroll
    for (int i = 0; i < (A_PIECES + 3) / 4; ++i) {
        const int piece = (swave + 4 * i) % A_PIECES;
        a_piece[i] = a_src + ((size_t)((piece / 3) * n_steps) * 3 + piece % 3) * 512;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int piece = swave + 4 * i;
        b_piece[i] = b_src + ((size_t)((piece / 3) * n_steps) * 3 + piece % 3) * 512;
    }
    auto dma_a = [&](int buf, int t) {
#pragma unroll
        for (int i = 0; i < (A_PIECES + 3) / 4; ++i) {
            const int piece = swave + 4 * i;
            if (A_PIECES % 4 == 0 || i < A_PIECES / 4 || swave < A_PIECES % 4)
                glds16(a_piece[i] + (size_t)t * 3 * 512, As + buf * A_STAGE + piece * 512);
        }
    };
    // (k-steps past the end re-read the last one: the data is never used, the count of pieces in flight stays uniform)
    auto dma_b = [&](int stage, int t) {
        const int tt = t < n_steps ? t : n_steps - 1;
#pragma unroll
        for (int i = 0; i < 3; ++i) glds16(b_piece[i] + (size_t)tt * 3 * 512, Bs + stage * B_STAGE + (swave + 4 * i) * 512);
    };

    f32x16 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto compute = [&](int abuf, int bstage) {
        const bf16_t* a = As + abuf * A_STAGE + (wm * MT * 3) * 512 + lane * 8;
        const bf16_t* b = Bs + bstage * B_STAGE + (wn * 2 * 3) * 512 + lane * 8;
        bf16x8 af[MT][3], bfr[2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) bfr[j][p] = *reinterpret_cast<const bf16x8*>(b + (j * 3 + p) * 512);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) af[i][p] = *reinterpret_cast<const bf16x8*>(a + (i * 3 + p) * 512);
        // six products per (row block, gallery fragment), smallest terms first - the order of k_cos_gemm_split
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bfr[j][0], acc[i][j], 0, 0, 0);   // l * h'
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[j][2], acc[i][j], 0, 0, 0);   // h * l'
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bfr[j][1], acc[i][j], 0, 0, 0);   // m * m'
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bfr[j][0], acc[i][j], 0, 0, 0);   // m * h'
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[j][1], acc[i][j], 0, 0, 0);   // h * m'
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[j][0], acc[i][j], 0, 0, 0);   // h * h'
            }
    };

    dma_a(0, 0);
    if (A_RING == 3 && n_steps > 1) dma_a(1, 1);
    dma_b(0, 0);
    dma_b(1, 1);
    __syncthreads();                   // drains vmcnt: everything has landed

    int bs_cur = 0, bs_far = 2;        // B stage of k-step t / of k-step t + 2 (and, at A_RING == 3, the A stages)
    for (int t = 0; t < n_steps; ++t) {
        if constexpr (A_RING == 2) {
            if (t + 1 < n_steps) dma_a((t & 1) ^ 1, t + 1);  // everybody left these buffers at the previous barrier
        } else {
            if (t + 2 < n_steps) dma_a(bs_far, t + 2);
        }
        __builtin_amdgcn_sched_barrier(0);                   // (the counts below need the A pieces issued BEFORE the B pieces)
        dma_b(bs_far, t + 2);
        __builtin_amdgcn_sched_barrier(0);
        compute(A_RING == 2 ? (t & 1) : bs_cur, bs_cur);
        if constexpr (A_RING == 2) {
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); // A(t+1) and B(t+1) have landed; the three B(t+2) pieces stay in flight
        } else {
            // A(t+2) (two pieces from waves 0 and 1, one from waves 2 and 3; none at the end) and B(t+2) stay in flight
            if (t + 2 >= n_steps) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (swave < A_PIECES % 4) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        bs_cur = bs_cur == 2 ? 0 : bs_cur + 1;
        bs_far = bs_far == 2 ? 0 : bs_far + 1;
    }
    __syncthreads();                   // the last look-ahead pieces have landed before the epilogue reuses the LDS
    cos_gemm_epilogue<MT, FK>(acc, smem, nullptr, S, Q, G, k, cand_val, cand_idx, x0, ntx, n0, m0);
}

// =====================================================================================
// few queries (Q <= 4, the reference's own per-query call shape cos(q[None], G), train/train.py:250): a GEMM tile
// would be 98 % padding; this is a GEMV, bound by streaming the gallery once (4*D bytes per row).  One wave per
// gallery row (6 KB contiguous for D = 1536), the normalised queries sit in LDS, two rows in flight per wave.
// =====================================================================================
template <int NQ>
__global__ __launch_bounds__(256) void k_cos_gemv(const float* __restrict__ Qn, const float* __restrict__ Gal,
                                                  const float* __restrict__ ginv, float* __restrict__ S, i64 G, int D,
                                                  int vec) {
    extern __shared__ __attribute__((aligned(16))) float qs[];   // [NQ][D]
    for (int i = threadIdx.x; i < NQ * D; i += 256) qs[i] = Qn[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const i64 wave_id = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    const i64 nwaves = (i64)gridDim.x * 4;
    for (i64 g = wave_id; g < G; g += nwaves) {
        const float* row = Gal + g * D;
        float acc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = 0.f;
        if (vec) {
            const f32x4* r4 = reinterpret_cast<const f32x4*>(row);
#pragma unroll 2
            for (int i = lane; i < D / 4; i += 64) {
                const f32x4 v = r4[i];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const f32x4 u = *reinterpret_cast<const f32x4*>(&qs[q * D + i * 4]);
                    acc[q] += v.x * u.x + v.y * u.y + v.z * u.z + v.w * u.w;
                }
            }
        } else {
            for (int i = lane; i < D; i += 64) {
                const float v = row[i];
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[q] += v * qs[q * D + i];
            }
        }
        const float gs = ginv ? ginv[g] : 1.0f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float t = wave_sum(acc[q]);
            if (lane == 0) S[(i64)q * G + g] = t * gs;
        }
    }
}

// ---- small k (<= 8): per-thread sorted list in registers, then k rounds of block arg-max.
// grid = (nchunk, Q).  Input row q: vals[q*in_stride + j], j in [0,rowlen); implicit index j (+offset)
// when idxs == nullptr.  Output: out[(q*nchunk + chunk)*k + r].
// idxs32 (optional): int32 LOCAL indices from the fused GEMM epilogue (IDX32_PAD = missing); idx_offset is added to them.
template <int K>
__global__ __launch_bounds__(256) void k_topk_small(const float* __restrict__ vals, const i64* __restrict__ idxs,
                                                    const int* __restrict__ idxs32,
                                                    i64 rowlen, i64 in_stride, i64 chunk_len, int k,
                                                    i64 idx_offset, float* __restrict__ ov, i64* __restrict__ oi) {
    const int tid = threadIdx.x;
    const i64 q = blockIdx.y;
    const i64 c0 = (i64)blockIdx.x * chunk_len;
    const i64 c1 = min(rowlen, c0 + chunk_len);
    const float* v = vals + q * in_stride;
    const i64* ix = idxs ? idxs + q * in_stride : nullptr;
    const int* ix32 = idxs32 ? idxs32 + q * in_stride : nullptr;

    float lv[K];
    i64 li[K];
#pragma unroll
    for (int i = 0; i < K; ++i) { lv[i] = NEG_INF; li[i] = IDX_PAD; }

    for (i64 j = c0 + tid; j < c1; j += 256) {
        const float x = v[j];
        i64 id;
        if (ix32) { const int t = ix32[j]; id = t == IDX32_PAD ? IDX_PAD : (i64)t + idx_offset; }
        else id = ix ? ix[j] : j + idx_offset;
        if (id != IDX_PAD && better(x, id, lv[K - 1], li[K - 1])) {
            lv[K - 1] = x; li[K - 1] = id;
#pragma unroll
            for (int i = K - 1; i > 0; --i) {
                if (better(lv[i], li[i], lv[i - 1], li[i - 1])) {
                    float tv = lv[i]; lv[i] = lv[i - 1]; lv[i - 1] = tv;
                    i64 ti = li[i]; li[i] = li[i - 1]; li[i - 1] = ti;
                }
            }
        }
    }

    __shared__ float sv[4];
    __shared__ i64 si[4];
    const int lane = tid & 63, wave = tid >> 6;
    float* o_v = ov + (q * gridDim.x + blockIdx.x) * k;
    i64* o_i = oi + (q * gridDim.x + blockIdx.x) * k;
    for (int r = 0; r < k; ++r) {
        float bv = lv[0];
        i64 bi = li[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov2 = __shfl_xor(bv, o, 64);
            const i64 oi2 = __shfl_xor(bi, o, 64);
            if (better(ov2, oi2, bv, bi)) { bv = ov2; bi = oi2; }
        }
        if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
        __syncthreads();
        bv = sv[0]; bi = si[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (better(sv[w], si[w], bv, bi)) { bv = sv[w]; bi = si[w]; }
        if (tid == 0) { o_v[r] = bv; o_i[r] = bi; }
        // the owner pops its head (indices are unique among real entries; pads never win a real slot)
        if (li[0] == bi && bi != IDX_PAD) {      // (by index only: a NaN score does not compare equal to itself)
#pragma unroll
            for (int i = 0; i < K - 1; ++i) { lv[i] = lv[i + 1]; li[i] = li[i + 1]; }
            lv[K - 1] = NEG_INF; li[K - 1] = IDX_PAD;
        }
        __syncthreads();
    }
}

// ---- any k <= 1024: bitonic sort of a 2048-element chunk in LDS, keep the first k.
constexpr int BT_N = 2048;
__global__ __launch_bounds__(256) void k_topk_bitonic(const float* __restrict__ vals, const i64* __restrict__ idxs,
                                                      i64 rowlen, i64 in_stride, int k, i64 idx_offset,
                                                      float* __restrict__ ov, i64* __restrict__ oi) {
    __shared__ float sv[BT_N];
    __shared__ i64 si[BT_N];
    const int tid = threadIdx.x;
    const i64 q = blockIdx.y;
    const i64 c0 = (i64)blockIdx.x * BT_N;
    const float* v = vals + q * in_stride;
    const i64* ix = idxs ? idxs + q * in_stride : nullptr;
    for (int j = tid; j < BT_N; j += 256) {
        const i64 g = c0 + j;
        if (g < rowlen) {
            sv[j] = v[g];
            si[j] = ix ? ix[g] : g + idx_offset;
        } else {
            sv[j] = NEG_INF;
            si[j] = IDX_PAD;
        }
    }
    __syncthreads();
    for (int size = 2; size <= BT_N; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < BT_N / 2; t += 256) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);  // first half of each size-block sorted descending
                const float a = sv[lo], b = sv[hi];
                const i64 ia = si[lo], ib = si[hi];
                const bool swap = desc ? better(b, ib, a, ia) : better(a, ia, b, ib);
                if (swap) { sv[lo] = b; sv[hi] = a; si[lo] = ib; si[hi] = ia; }
            }
            __syncthreads();
        }
    }
    float* o_v = ov + (q * gridDim.x + blockIdx.x) * k;
    i64* o_i = oi + (q * gridDim.x + blockIdx.x) * k;
    for (int j = tid; j < k; j += 256) { o_v[j] = sv[j]; o_i[j] = si[j]; }
}

// =====================================================================================
// small ops
// =====================================================================================
__global__ __launch_bounds__(256) void k_pair_cosine(const float* __restrict__ a, const float* __restrict__ b,
                                                     i64 rows, int dim, float eps, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const i64 row = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* x = a + row * dim;
    const float* y = b + row * dim;
    float xx = 0.f, yy = 0.f, xy = 0.f;
    for (int i = lane; i < dim; i += 64) {
        const float u = x[i], w = y[i];
        xx += u * u; yy += w * w; xy += u * w;
    }
    xx = wave_sum(xx); yy = wave_sum(yy); xy = wave_sum(xy);
    if (lane == 0) out[row] = xy / (fmaxf(sqrtf(xx), eps) * fmaxf(sqrtf(yy), eps));
}

// One block of 16 waves; wave w owns rows w, w+16, ... and adds them in that order, then thread 0
// adds the 16 wave partials in wave order: the result is a fixed function of the inputs.
__global__ __launch_bounds__(1024) void k_contrastive(const float* __restrict__ f1, const float* __restrict__ f2,
                                                      i64 rows, int dim, float label, float margin, int mean,
                                                      float* __restrict__ out, float* __restrict__ per_row) {
    __shared__ float part[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    for (i64 r = wave; r < rows; r += 16) {
        const float* x = f1 + r * dim;
        const float* y = f2 + r * dim;
        float d = 0.f;
        for (int i = lane; i < dim; i += 64) {
            const float t = y[i] - x[i];
            d += t * t;
        }
        d = wave_sum(d);
        const float h = fmaxf(margin - sqrtf(d + 1e-9f), 0.f);
        const float l = 0.5f * (label * d + (1.0f - label) * h * h);
        if (per_row && lane == 0) per_row[r] = l;
        acc += l;
    }
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int w = 0; w < 16; ++w) s += part[w];
        out[0] = mean ? s / (float)rows : s;
    }
}

// torch.nn.CosineEmbeddingLoss (ATen cosine_embedding_loss): cos = xy / sqrt((xx + 1e-12)(yy + 1e-12));
// target +1 -> 1 - cos, target -1 -> max(0, cos - margin); mean (or sum) over rows in a fixed order.
__global__ __launch_bounds__(1024) void k_cos_embedding_loss(const float* __restrict__ f1, const float* __restrict__ f2,
                                                             i64 rows, int dim, float target, float margin, int mean,
                                                             float* __restrict__ out) {
    __shared__ float part[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    for (i64 r = wave; r < rows; r += 16) {
        const float* x = f1 + r * dim;
        const float* y = f2 + r * dim;
        float xx = 0.f, yy = 0.f, xy = 0.f;
        for (int i = lane; i < dim; i += 64) {
            const float u = x[i], v = y[i];
            xx += u * u; yy += v * v; xy += u * v;
        }
        xx = wave_sum(xx); yy = wave_sum(yy); xy = wave_sum(xy);
        const float c = xy / sqrtf((xx + 1e-12f) * (yy + 1e-12f));
        acc += target > 0.f ? 1.0f - c : fmaxf(c - margin, 0.f);
    }
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += part[w];
        out[0] = mean ? t / (float)rows : t;
    }
}

// An index outside [0, G) (a pad entry of a list with fewer than k real candidates) counts as a miss.
// packed candidate lists of the sharded search (see mi355_pack_candidates / mi355_merge_packed_topk)
__global__ __launch_bounds__(256) void k_pack_candidates(const float* __restrict__ val, const i64* __restrict__ idx, i64 Q,
                                                         int kk, int k, int* __restrict__ packed) {
    const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    if (t >= Q * k) return;
    const i64 q = t / k;
    const int j = (int)(t - q * k);
    int2 o;
    if (j < kk) { o.x = __float_as_int(val[q * kk + j]); o.y = (int)idx[q * kk + j]; }
    else { o.x = __float_as_int(NEG_INF); o.y = -1; }
    reinterpret_cast<int2*>(packed)[t] = o;
}

constexpr i64 PACKED_PAD_IDX = (i64)1 << 62;    // "no candidate": sorts behind every real index at equal (-inf) score
__global__ __launch_bounds__(256) void k_unpack_candidates(const int* __restrict__ packed, const i64* __restrict__ offsets,
                                                           int world, i64 Q, int k, float* __restrict__ cv, i64* __restrict__ ci) {
    const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;      // output slot: query q, candidate (r, j) = r * k + j
    const i64 per = (i64)world * k;
    if (t >= Q * per) return;
    const i64 q = t / per;
    const int c = (int)(t - q * per), r = c / k, j = c - r * k;
    const int2 p = reinterpret_cast<const int2*>(packed)[((i64)r * Q + q) * k + j];
    cv[t] = __int_as_float(p.x);
    ci[t] = p.y >= 0 ? (i64)p.y + offsets[r] : PACKED_PAD_IDX;
}

__global__ void k_hit_counts(const i64* __restrict__ idx, i64 Q, int k, const i64* __restrict__ qcls,
                             const i64* __restrict__ gcls, i64 G, i64* __restrict__ counts) {
    const i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    int h1 = 0, h3 = 0;
    if (q < Q) {
        const i64 c = qcls[q];
        for (int j = 0; j < min(k, 3); ++j) {
            const i64 g = idx[q * k + j];
            const int hit = (g >= 0 && g < G) ? (gcls[g] == c) : 0;
            if (j == 0) h1 = hit;
            h3 |= hit;
        }
    }
    // integer atomics: order-independent
    const unsigned long long m1 = __ballot(h1), m3 = __ballot(h3);
    if ((threadIdx.x & 63) == 0) {
        if (m1) atomicAdd((unsigned long long*)&counts[0], (unsigned long long)__popcll(m1));
        if (m3) atomicAdd((unsigned long long*)&counts[1], (unsigned long long)__popcll(m3));
    }
}

__global__ void k_distinct_topn(const i64* __restrict__ idx, const float* __restrict__ val, i64 Q, int k,
                                const i64* __restrict__ gcls, i64 G, int n, i64* __restrict__ ocls,
                                i64* __restrict__ oidx, float* __restrict__ oval) {
    const i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    i64 seen[8];
    int ns = 0;
    for (int j = 0; j < n; ++j) { ocls[q * n + j] = -1; oidx[q * n + j] = -1; oval[q * n + j] = NAN; }
    for (int j = 0; j < k && ns < n; ++j) {
        const i64 g = idx[q * k + j];
        if (g < 0 || g >= G) continue;          // pad entry: not a candidate
        const i64 c = gcls[g];
        bool dup = false;
#pragma unroll
        for (int s = 0; s < 8; ++s) dup |= (s < ns && seen[s] == c);
        if (!dup) {
#pragma unroll
            for (int s = 0; s < 8; ++s)
                if (s == ns) seen[s] = c;
            ocls[q * n + ns] = c; oidx[q * n + ns] = g; oval[q * n + ns] = val[q * k + j];
            ++ns;
        }
    }
}

// =====================================================================================
// host drivers
// =====================================================================================
constexpr int SMALL_K = 8;
constexpr int LARGE_K = 1024;
constexpr i64 SMALL_CHUNK = 8192;

struct TopkPlan {
    bool small;
    i64 nchunk1;        // level-1 chunks per row
    size_t cand_elems;  // elements per ping/pong candidate buffer (per query row) * Q
};

static i64 level1_chunks(i64 G, int k) { return k <= SMALL_K ? cdiv(G, SMALL_CHUNK) : cdiv(G, BT_N); }

static size_t topk_ws_bytes(i64 Q, i64 G, int k) {
    const i64 per_row = level1_chunks(G, k) * k;
    // two ping-pong buffers of (val f32 + idx i64)
    return 2 * align_up((size_t)Q * per_row * (sizeof(float) + sizeof(i64)), 256) + 256;
}

template <int K>
static void launch_small(const float* v, const i64* ix, const int* ix32, i64 rowlen, i64 in_stride, i64 chunk_len, int k,
                         i64 off, float* ov, i64* oi, i64 nchunk, i64 Q, hipStream_t st) {
    hipLaunchKernelGGL((k_topk_small<K>), dim3((unsigned)nchunk, (unsigned)Q), dim3(256), 0, st, v, ix, ix32, rowlen,
                       in_stride, chunk_len, k, off, ov, oi);
}
static void dispatch_small(const float* v, const i64* ix, const int* ix32, i64 rowlen, i64 in_stride, i64 chunk_len, int k,
                           i64 off, float* ov, i64* oi, i64 nchunk, i64 Q, hipStream_t st) {
    if (k <= 1) launch_small<1>(v, ix, ix32, rowlen, in_stride, chunk_len, k, off, ov, oi, nchunk, Q, st);
    else if (k <= 2) launch_small<2>(v, ix, ix32, rowlen, in_stride, chunk_len, k, off, ov, oi, nchunk, Q, st);
    else if (k <= 4) launch_small<4>(v, ix, ix32, rowlen, in_stride, chunk_len, k, off, ov, oi, nchunk, Q, st);
    else launch_small<8>(v, ix, ix32, rowlen, in_stride, chunk_len, k, off, ov, oi, nchunk, Q, st);
}

// Select top-k of each row of vals[Q][rowlen] (implicit or explicit indices) into out_val/out_idx [Q][k].
// idxs32 (with idxs == nullptr): int32 local candidate indices of the fused GEMM epilogue, k <= SMALL_K only.
static int topk_select(const float* vals, const i64* idxs, i64 Q, i64 rowlen, i64 in_stride, int k, i64 idx_offset,
                       float* out_val, i64* out_idx, void* ws, size_t ws_bytes, hipStream_t st,
                       const int* idxs32 = nullptr) {
    MI355_REQUIRE(k >= 1 && k <= LARGE_K, "top-k: k=%d outside [1,%d]", k, LARGE_K);
    MI355_REQUIRE(k <= rowlen, "top-k: k=%d exceeds row length %lld", k, (long long)rowlen);
    MI355_REQUIRE(!idxs32 || k <= SMALL_K, "top-k: int32 candidate lists need k <= %d", SMALL_K);
    MI355_REQUIRE(Q >= 1 && Q <= 65535 * 16, "top-k: Q=%lld out of range", (long long)Q);
    MI355_REQUIRE(ws_bytes >= topk_ws_bytes(Q, rowlen, k), "top-k: workspace %zu < %zu bytes", ws_bytes,
                  topk_ws_bytes(Q, rowlen, k));
    const i64 per_row_max = level1_chunks(rowlen, k) * k;
    const size_t half = align_up((size_t)Q * per_row_max * (sizeof(float) + sizeof(i64)), 256);
    char* base = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    float* cv[2];
    i64* ci[2];
    for (int b = 0; b < 2; ++b) {
        ci[b] = (i64*)(base + b * half);
        cv[b] = (float*)(base + b * half + (size_t)Q * per_row_max * sizeof(i64));
    }
    // grid.y limit: split Q into slabs of 65535 rows
    for (i64 qs = 0; qs < Q; qs += 65535) {
        const i64 qn = (Q - qs < 65535) ? Q - qs : 65535;
        const float* v = vals + qs * in_stride;
        const i64* ix = idxs ? idxs + qs * in_stride : nullptr;
        const int* ix32 = idxs32 ? idxs32 + qs * in_stride : nullptr;
        i64 len = rowlen, stride = in_stride;
        i64 off = idx_offset;
        int cur = 0;
        while (true) {
            const bool small = (k <= SMALL_K);
            const i64 chunk = small ? SMALL_CHUNK : BT_N;
            const i64 nchunk = cdiv(len, chunk);
            const bool last = (nchunk == 1);
            float* ovp = last ? out_val + qs * k : cv[cur];
            i64* oip = last ? out_idx + qs * k : ci[cur];
            if (small) dispatch_small(v, ix, ix32, len, stride, chunk, k, off, ovp, oip, nchunk, qn, st);
            else hipLaunchKernelGGL(k_topk_bitonic, dim3((unsigned)nchunk, (unsigned)qn), dim3(256), 0, st, v, ix,
                                    len, stride, k, off, ovp, oip);
            MI355_LAUNCH_CHECK();
            if (last) break;
            v = cv[cur]; ix = ci[cur]; ix32 = nullptr;
            len = nchunk * k; stride = len; off = 0;
            cur ^= 1;
        }
    }
    return OK;
}

static bool fused_select(i64 Q, i64 G, int k) { return k >= 1 && k <= SMALL_K && Q > 4 && G < ((i64)1 << 31) - RK_BN; }

static i64 query_block(i64 Q, i64 G, int k) {
    if (fused_select(Q, G, k)) {
        // no score slab: candidates are cdiv(G,128) * k * 8 B per query; keep them <= 32 MiB per block of queries (a second
        // block costs one more pass over the gallery, which a GEMM of >= 1000 queries hides)
        i64 qb = ((i64)1 << 25) / (cdiv(G, RK_BN) * (i64)k * 8);
        qb = qb / 128 * 128;
        if (qb < 128) qb = 128;
        return qb < Q ? qb : Q;
    }
    // keep the score slab S[qb][G] around <= 1 GiB, in multiples of 256 queries
    i64 qb = ((i64)1 << 28) / (G > 0 ? G : 1);
    qb = qb / 256 * 256;
    if (qb < 256) qb = 256;
    return qb < Q ? qb : Q;
}

struct RankWs {
    float* qn; bf16_t* qs; float* ginv; float* S; float* cand_val; int* cand_idx; void* topk; size_t topk_bytes; size_t total;
};
// split query planes (k_split_queries): whole 128-row tiles, 16-deep k steps, three planes
static size_t split_queries_bytes(i64 Q, int D) { return (size_t)cdiv(Q, 128) * 4 * cdiv(D, 16) * 3 * 1024 + 256; }
static RankWs carve(void* ws, i64 Q, i64 G, int D, int k, bool need_ginv, bool need_S = true) {
    RankWs r{};
    const bool fused = need_S && fused_select(Q, G, k);
    const i64 qb = query_block(Q, G, need_S ? k : 0);
    size_t off = 0;
    char* base = ws ? (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255) : nullptr;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes, 256); return p; };
    r.qn = (float*)take((size_t)Q * D * sizeof(float));
    const i64 q_split = need_S ? qb : (Q < 256 * 64 ? Q : 256 * 64);      // queries of one cos_gemm call
    r.qs = (bf16_t*)take(split_queries_bytes(q_split, D));   // (also for Q <= 4: rows too long for the GEMV's LDS copy take the GEMM)
    r.ginv = (float*)take(need_ginv ? (size_t)G * sizeof(float) : 0);
    if (fused) {
        const size_t ncand = (size_t)qb * cdiv(G, RK_BN) * k;
        r.cand_val = (float*)take(ncand * sizeof(float));
        r.cand_idx = (int*)take(ncand * sizeof(int));
        r.topk_bytes = topk_ws_bytes(qb, cdiv(G, RK_BN) * (i64)k, k);
    } else {
        r.S = (float*)take(need_S ? (size_t)qb * G * sizeof(float) : 0);
        r.topk_bytes = k > 0 ? topk_ws_bytes(qb, G, k) : 0;
    }
    r.topk = take(r.topk_bytes);
    r.total = off + 256;
    return r;
}

// resident workgroups per CU x CUs of the current device for one kernel instantiation (cached per device by the caller)
static int kernel_slots(const void* fn, size_t lds, int* cache, int* slots_out) {
    int dev = 0;
    MI355_CHECK_HIP(hipGetDevice(&dev));
    MI355_REQUIRE(dev >= 0 && dev < MI355_MAX_DEVICES, "rank: device ordinal %d out of range", dev);
    if (!cache[dev]) {
        MI355_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int per_cu = 0, cus = 0;
        MI355_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds));
        MI355_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        cache[dev] = (per_cu > 0 ? per_cu : 1) * (cus > 0 ? cus : 1);
    }
    *slots_out = cache[dev];
    return OK;
}
template <int MT, int BK, bool VEC, int FK>
static int gemm_slots(size_t lds, int* slots_out) {
    static int slots[MI355_MAX_DEVICES] = {0};
    return kernel_slots((const void*)k_cos_gemm<MT, BK, VEC, FK>, lds, slots, slots_out);
}
template <int MT, int FK>
static int split_slots(size_t lds, int* slots_out) {
    static int slots[MI355_MAX_DEVICES] = {0};
    return kernel_slots((const void*)k_cos_gemm_split<MT, FK>, lds, slots, slots_out);
}

template <int MT, int BK>
static size_t gemm_lds(bool fk) {
    const size_t stage = (size_t)2 * (64 * MT + RK_BN) * (BK + 4) * sizeof(float);
    const size_t tile = fk ? (size_t)64 * (RK_BN + 4) * sizeof(float) : 0;   // fused selection: 64 rows of the score tile at a time
    return stage > tile ? stage : tile;
}
template <int MT>
static size_t split_lds(bool fk) {
    const size_t stage = (size_t)(MT == 1 ? 3 : 2) * (64 * MT / 32) * 3 * 1024 + (size_t)3 * RK_BN * 16 * sizeof(float);   // A ring of 2 (3 at MT = 1), B ring of 3
    const size_t tile = fk ? (size_t)64 * (RK_BN + 4) * sizeof(float) : 0;
    return stage > tile ? stage : tile;
}

// Wave quantisation: 1564 tiles on 768 slots run as 2.04 rounds and the 28 tiles of the third round cost a whole round
// (0.15 ms of 0.83 at Q=256 x 100k on the fp32 loop).  Whole rounds go out as 128-row tiles, the remainder as a second
// launch of 64-row tiles (same column tiles, same k order: every score is bit-identical), which halves the tiles' length
// and doubles their number.  Returns the number of column tiles of the main launch.
static int whole_round_tiles(int ntx, int ny, int slots) {
    if ((long)ntx * ny > slots && ((long)ntx * ny) % slots != 0) return (int)(((long)ntx * ny / slots) * slots / ny);
    return ntx;
}

template <int MT, int BK, bool VEC, int FK>
static int launch_gemm(const float* qn, const float* gal, const float* ginv, float* S, int Q, i64 G, int D, int k,
                       float* cand_val, int* cand_idx, hipStream_t st) {
    constexpr int BM = 64 * MT;
    const size_t lds = gemm_lds<MT, BK>(FK > 0);
    int slots = 0;
    if (int e = gemm_slots<MT, BK, VEC, FK>(lds, &slots)) return e;
    const int ntx = cdiv(G, RK_BN), ny = cdiv(Q, BM);
    const int xm = MT == 2 ? whole_round_tiles(ntx, ny, slots) : ntx;
    if (xm > 0) {
        hipLaunchKernelGGL((k_cos_gemm<MT, BK, VEC, FK>), dim3((unsigned)xm * (unsigned)ny), dim3(256), lds, st, qn, gal, ginv, S,
                           Q, G, D, k, cand_val, cand_idx, 0, ntx, xm, ny);
        MI355_LAUNCH_CHECK();
    }
    if (xm < ntx) {
        const size_t lds1 = gemm_lds<1, 32>(FK > 0);
        int slots1 = 0;
        if (int e = gemm_slots<1, 32, VEC, FK>(lds1, &slots1)) return e;
        hipLaunchKernelGGL((k_cos_gemm<1, 32, VEC, FK>), dim3((unsigned)(ntx - xm) * (unsigned)cdiv(Q, 64)), dim3(256), lds1, st,
                           qn, gal, ginv, S, Q, G, D, k, cand_val, cand_idx, xm, ntx, ntx - xm, (int)cdiv(Q, 64));
        MI355_LAUNCH_CHECK();
    }
    return OK;
}

// split-bf16 loop: qs = the split planes of these Q queries (k_split_queries, whole 128-row tiles)
template <int MT, int FK>
static int launch_split(const bf16_t* qs, const float* gal, const float* ginv, float* S, int Q, i64 G, int D, int k,
                        float* cand_val, int* cand_idx, hipStream_t st) {
    constexpr int BM = 64 * MT;
    const size_t lds = split_lds<MT>(FK > 0);
    int slots = 0;
    if (int e = split_slots<MT, FK>(lds, &slots)) return e;
    const int ntx = cdiv(G, RK_BN), ny = cdiv(Q, BM), n_steps = cdiv(D, 16);
    const float* zeros = reinterpret_cast<const float*>(qs + (size_t)cdiv(Q, 128) * 4 * n_steps * 3 * 512);
    const int xm = MT == 2 ? whole_round_tiles(ntx, ny, slots) : ntx;
    if (xm > 0) {
        hipLaunchKernelGGL((k_cos_gemm_split<MT, FK>), dim3((unsigned)xm * (unsigned)ny), dim3(256), lds, st, qs, gal, ginv,
                           S, Q, G, D, k, cand_val, cand_idx, 0, ntx, n_steps, zeros, xm, ny);
        MI355_LAUNCH_CHECK();
    }
    if (xm < ntx) {
        const size_t lds1 = split_lds<1>(FK > 0);
        int slots1 = 0;
        if (int e = split_slots<1, FK>(lds1, &slots1)) return e;
        hipLaunchKernelGGL((k_cos_gemm_split<1, FK>), dim3((unsigned)(ntx - xm) * (unsigned)cdiv(Q, 64)), dim3(256), lds1,
                           st, qs, gal, ginv, S, Q, G, D, k, cand_val, cand_idx, xm, ntx, n_steps, zeros, ntx - xm, (int)cdiv(Q, 64));
        MI355_LAUNCH_CHECK();
    }
    return OK;
}

template <int MT, int BK, bool VEC>
static int launch_gemm_fk(const float* qn, const float* gal, const float* ginv, float* S, int Q, i64 G, int D, int k,
                          float* cand_val, int* cand_idx, hipStream_t st) {
    if (!cand_val) return launch_gemm<MT, BK, VEC, 0>(qn, gal, ginv, S, Q, G, D, 0, nullptr, nullptr, st);
    if (k <= 1) return launch_gemm<MT, BK, VEC, 1>(qn, gal, ginv, S, Q, G, D, k, cand_val, cand_idx, st);
    if (k <= 2) return launch_gemm<MT, BK, VEC, 2>(qn, gal, ginv, S, Q, G, D, k, cand_val, cand_idx, st);
    if (k <= 4) return launch_gemm<MT, BK, VEC, 4>(qn, gal, ginv, S, Q, G, D, k, cand_val, cand_idx, st);
    return launch_gemm<MT, BK, VEC, 8>(qn, gal, ginv, S, Q, G, D, k, cand_val, cand_idx, st);
}
template <int MT>
static int launch_split_fk(const bf16_t* qs, const float* gal, const float* ginv, float* S, int Q, i64 G, int D, int k,
                           float* cand_val, int* cand_idx, hipStream_t st) {
    if (!cand_val) return launch_split<MT, 0>(qs, gal, ginv, S, Q, G, D, 0, nullptr, nullptr, st);
    if (k <= 1) return launch_split<MT, 1>(qs, gal, ginv, S, Q, G, D, k, cand_val, cand_idx, st);
    if (k <= 2) return launch_split<MT, 2>(qs, gal, ginv, S, Q, G, D, k, cand_val, cand_idx, st);
    if (k <= 4) return launch_split<MT, 4>(qs, gal, ginv, S, Q, G, D, k, cand_val, cand_idx, st);
    return launch_split<MT, 8>(qs, gal, ginv, S, Q, G, D, k, cand_val, cand_idx, st);
}

template <int MT>
static size_t pre_lds(bool fk) {
    const size_t stage = (size_t)(MT == 1 ? 3 : 2) * (64 * MT / 32) * 3 * 1024 + (size_t)3 * (RK_BN / 32) * 3 * 1024;   // A ring of 2 (3 at MT = 1), B ring of 3
    const size_t tile = fk ? (size_t)64 * (RK_BN + 4) * sizeof(float) : 0;
    return stage > tile ? stage : tile;
}
template <int MT, int FK>
static int pre_slots(size_t lds, int* slots_out) {
    static int slots[MI355_MAX_DEVICES] = {0};
    return kernel_slots((const void*)k_cos_gemm_pre<MT, FK>, lds, slots, slots_out);
}
// prepared gallery: qs = the split planes of these Q queries, gs = the gallery's planes (whole 128-row tiles)
template <int MT, int FK>
static int launch_pre(const bf16_t* qs, const bf16_t* gs, int Q, i64 G, int D, int k, float* cand_val, int* cand_idx, hipStream_t st) {
    constexpr int BM = 64 * MT;
    const size_t lds = pre_lds<MT>(FK > 0);
    int slots = 0;
    if (int e = pre_slots<MT, FK>(lds, &slots)) return e;
    const int ntx = cdiv(G, RK_BN), ny = cdiv(Q, BM), n_steps = cdiv(D, 16);
    const int xm = MT == 2 ? whole_round_tiles(ntx, ny, slots) : ntx;
    if (xm > 0) {
        hipLaunchKernelGGL((k_cos_gemm_pre<MT, FK>), dim3((unsigned)xm * (unsigned)ny), dim3(256), lds, st, qs, gs, (float*)nullptr, Q, G, k,
                           cand_val, cand_idx, 0, ntx, n_steps, xm, ny);
        MI355_LAUNCH_CHECK();
    }
    if (xm < ntx) {
        const size_t lds1 = pre_lds<1>(FK > 0);
        int slots1 = 0;
        if (int e = pre_slots<1, FK>(lds1, &slots1)) return e;
        hipLaunchKernelGGL((k_cos_gemm_pre<1, FK>), dim3((unsigned)(ntx - xm) * (unsigned)cdiv(Q, 64)), dim3(256), lds1, st, qs, gs,
                           (float*)nullptr, Q, G, k, cand_val, cand_idx, xm, ntx, n_steps, ntx - xm, (int)cdiv(Q, 64));
        MI355_LAUNCH_CHECK();
    }
    return OK;
}
template <int MT>
static int launch_pre_fk(const bf16_t* qs, const bf16_t* gs, int Q, i64 G, int D, int k, float* cand_val, int* cand_idx, hipStream_t st) {
    if (k <= 1) return launch_pre<MT, 1>(qs, gs, Q, G, D, k, cand_val, cand_idx, st);
    if (k <= 2) return launch_pre<MT, 2>(qs, gs, Q, G, D, k, cand_val, cand_idx, st);
    if (k <= 4) return launch_pre<MT, 4>(qs, gs, Q, G, D, k, cand_val, cand_idx, st);
    return launch_pre<MT, 8>(qs, gs, Q, G, D, k, cand_val, cand_idx, st);
}

// MI355_RANK_EXACT_F32=1 keeps the GEMM on v_mfma_f32_32x32x2_f32 (a bit-for-bit fmaf chain, 2.7x the matrix-pipe time);
// the default is the three-way bf16 split with six products (fp32-equivalent, see split3).
static bool rank_exact_f32() {
    const char* e = getenv("MI355_RANK_EXACT_F32");
    return e && e[0] && e[0] != '0';
}

// S != nullptr: score slab.  cand_val / cand_idx != nullptr: fused per-tile top-k lists [Q][cdiv(G,128)][k] (Q > 4 only).
// qs: scratch for the split planes of these Q queries (split_queries_bytes(Q, D)); may be null for Q <= 4.
static int cos_gemm(const float* qn, bf16_t* qs, const float* gal, const float* ginv, float* S, i64 Q, i64 G, int D,
                    hipStream_t st, int k = 0, float* cand_val = nullptr, int* cand_idx = nullptr) {
    const bool vec = vec_ok(qn, D) && vec_ok(gal, D);
    const int q = (int)Q;
    if (!cand_val && Q <= 4 && (size_t)Q * D * sizeof(float) <= 60 * 1024) {
        const size_t lds = (size_t)Q * D * sizeof(float);
        const unsigned blocks = (unsigned)(cdiv(G, 4) < 4096 ? cdiv(G, 4) : 4096);
#define GEMV_LAUNCH(NQ) hipLaunchKernelGGL((k_cos_gemv<NQ>), dim3(blocks), dim3(256), lds, st, qn, gal, ginv, S, G, D, (int)vec)
        if (Q == 1) GEMV_LAUNCH(1); else if (Q == 2) GEMV_LAUNCH(2); else if (Q == 3) GEMV_LAUNCH(3); else GEMV_LAUNCH(4);
#undef GEMV_LAUNCH
        MI355_LAUNCH_CHECK();
        return OK;
    }
    if (qs && vec_ok(gal, D) && !rank_exact_f32()) {
        const int n_steps = cdiv(D, 16), n_frag = cdiv(q, 128) * 4 * n_steps;
        hipLaunchKernelGGL(k_split_queries, dim3((unsigned)cdiv(n_frag, 4)), dim3(256), 0, st, qn, qs, q, D, n_steps, n_frag);
        MI355_LAUNCH_CHECK();
        if (Q > 64) return launch_split_fk<2>(qs, gal, ginv, S, q, G, D, k, cand_val, cand_idx, st);
        return launch_split_fk<1>(qs, gal, ginv, S, q, G, D, k, cand_val, cand_idx, st);
    }
    // Exact-fp32 loop.  Tile choice, measured on MI355X (tools/bench_rank.py, D = 1536): it plateaus at 95-110 TFLOP/s
    // for every tile shape, so what differs is the partial last round of tiles.  128-query tiles with BK = 16 keep two
    // workgroups on a CU (41 KB of staging, 68 KB with the fused selection's score tile): a lone workgroup in the last
    // round runs at full speed, which halves the wave-quantisation loss (Q = 256, G = 100k: 0.83 ms, against 0.95 ms
    // with 256 x 128 tiles, one per CU).
    if (Q > 64) return vec ? launch_gemm_fk<2, 16, true>(qn, gal, ginv, S, q, G, D, k, cand_val, cand_idx, st)
                           : launch_gemm_fk<2, 16, false>(qn, gal, ginv, S, q, G, D, k, cand_val, cand_idx, st);
    return vec ? launch_gemm_fk<1, 32, true>(qn, gal, ginv, S, q, G, D, k, cand_val, cand_idx, st)
               : launch_gemm_fk<1, 32, false>(qn, gal, ginv, S, q, G, D, k, cand_val, cand_idx, st);
}

static int check_rank_args(const float* queries, i64 Q, const float* gallery, i64 G, int dim) {
    MI355_REQUIRE(queries && gallery, "rank: null queries/gallery pointer");
    MI355_REQUIRE(Q >= 1, "rank: Q=%lld must be >= 1", (long long)Q);
    MI355_REQUIRE(G >= 1, "rank: G=%lld must be >= 1", (long long)G);
    MI355_REQUIRE(dim >= 1, "rank: dim=%d must be >= 1", dim);
    return OK;
}

}  // namespace mi355

using namespace mi355;

extern "C" {

int mi355_l2_normalize_rows(const float* in, float* out, int64_t rows, int dim, float eps, void* stream) {
    MI355_REQUIRE(in && out, "l2_normalize_rows: null pointer");
    MI355_REQUIRE(rows >= 0 && dim >= 1, "l2_normalize_rows: bad shape rows=%lld dim=%d", (long long)rows, dim);
    if (rows == 0) return OK;
    const int vec = vec_ok(in, dim) && vec_ok(out, dim);
    hipLaunchKernelGGL((k_row_norm<true>), dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, in, out,
                       (float*)nullptr, (i64)rows, dim, eps, vec);
    MI355_LAUNCH_CHECK();
    return OK;
}

size_t mi355_rank_workspace_bytes(int64_t Q, int64_t G, int dim, int k) {
    if (Q < 1 || G < 1) return 0;
    if (dim <= 0) return topk_ws_bytes(Q, G, k < 1 ? 1 : k) + 256;
    return carve(nullptr, Q, G, dim, k, true).total;
}

int mi355_cosine_scores(const float* queries, int64_t Q, const float* gallery, int64_t G, int dim,
                        int gallery_is_normalized, float eps, float* out, void* workspace, size_t workspace_bytes,
                        void* stream) {
    if (int e = check_rank_args(queries, Q, gallery, G, dim)) return e;
    MI355_REQUIRE(out, "cosine_scores: null output");
    hipStream_t st = (hipStream_t)stream;
    RankWs w = carve(workspace, Q, G, dim, 0, !gallery_is_normalized, false);
    MI355_REQUIRE(workspace && workspace_bytes >= w.total, "cosine_scores: workspace %zu < %zu bytes",
                  workspace_bytes, w.total);
    const int vq = vec_ok(queries, dim) && vec_ok(w.qn, dim);
    hipLaunchKernelGGL((k_row_norm<true>), dim3((unsigned)cdiv(Q, 4)), dim3(256), 0, st, queries, w.qn,
                       (float*)nullptr, (i64)Q, dim, eps, vq);
    MI355_LAUNCH_CHECK();
    if (!gallery_is_normalized) {
        hipLaunchKernelGGL((k_row_norm<false>), dim3((unsigned)cdiv(G, 4)), dim3(256), 0, st, gallery,
                           (float*)nullptr, w.ginv, (i64)G, dim, eps, vec_ok(gallery, dim));
        MI355_LAUNCH_CHECK();
    }
    const float* ginv = gallery_is_normalized ? nullptr : w.ginv;
    for (i64 qs = 0; qs < Q; qs += 256 * 64) {  // grid.y stays small
        const i64 qn = (Q - qs < 256 * 64) ? Q - qs : 256 * 64;
        if (int e = cos_gemm(w.qn + qs * dim, w.qs, gallery, ginv, out + qs * G, qn, G, dim, st)) return e;
    }
    return OK;
}

int mi355_rank_topk(const float* queries, int64_t Q, const float* gallery, int64_t G, int dim,
                    int gallery_is_normalized, int k, float eps, int64_t idx_offset, float* out_val,
                    int64_t* out_idx, void* workspace, size_t workspace_bytes, void* stream) {
    if (int e = check_rank_args(queries, Q, gallery, G, dim)) return e;
    MI355_REQUIRE(out_val && out_idx, "rank_topk: null output");
    MI355_REQUIRE(k >= 1 && k <= LARGE_K, "rank_topk: k=%d outside [1,%d]", k, LARGE_K);
    MI355_REQUIRE(k <= G, "rank_topk: k=%d exceeds gallery rows %lld", k, (long long)G);
    hipStream_t st = (hipStream_t)stream;
    RankWs w = carve(workspace, Q, G, dim, k, !gallery_is_normalized);
    MI355_REQUIRE(workspace && workspace_bytes >= w.total, "rank_topk: workspace %zu < %zu bytes", workspace_bytes,
                  w.total);
    const int vq = vec_ok(queries, dim) && vec_ok(w.qn, dim);
    {
        RoctxRange range("rank/normalize");
        hipLaunchKernelGGL((k_row_norm<true>), dim3((unsigned)cdiv(Q, 4)), dim3(256), 0, st, queries, w.qn,
                           (float*)nullptr, (i64)Q, dim, eps, vq);
        MI355_LAUNCH_CHECK();
        if (!gallery_is_normalized) {
            hipLaunchKernelGGL((k_row_norm<false>), dim3((unsigned)cdiv(G, 4)), dim3(256), 0, st, gallery,
                               (float*)nullptr, w.ginv, (i64)G, dim, eps, vec_ok(gallery, dim));
            MI355_LAUNCH_CHECK();
        }
    }
    const float* ginv = gallery_is_normalized ? nullptr : w.ginv;
    const i64 qb = query_block(Q, G, k);
    const bool fused = fused_select(Q, G, k);
    const i64 ntiles = cdiv(G, RK_BN);
    for (i64 qs = 0; qs < Q; qs += qb) {
        const i64 qn = (Q - qs < qb) ? Q - qs : qb;
        if (fused) {
            // per-tile top-k straight from the GEMM's accumulators, then a merge of qn x ntiles x k candidates
            {
                RoctxRange range("rank/cosine gemm + per-tile top-k");
                if (int e = cos_gemm(w.qn + qs * dim, w.qs, gallery, ginv, nullptr, qn, G, dim, st, k, w.cand_val, w.cand_idx)) return e;
            }
            RoctxRange range("rank/merge candidates");
            if (int e = topk_select(w.cand_val, nullptr, qn, ntiles * k, ntiles * k, k, idx_offset, out_val + qs * k,
                                    (i64*)out_idx + qs * k, w.topk, w.topk_bytes, st, w.cand_idx))
                return e;
            continue;
        }
        if (int e = cos_gemm(w.qn + qs * dim, w.qs, gallery, ginv, w.S, qn, G, dim, st)) return e;
        if (int e = topk_select(w.S, nullptr, qn, G, G, k, idx_offset, out_val + qs * k, (i64*)out_idx + qs * k,
                                w.topk, w.topk_bytes, st))
            return e;
    }
    return OK;
}

// ---- prepared gallery (resident galleries: Gallery / ShardedGallery hold it next to nothing else for k <= 8 searches)
// planes of the NORMALISED rows, in the fragment order of the GEMM's B operand: whole 128-row tiles, 16-deep k steps, 6 B per element
size_t mi355_gallery_planes_bytes(int64_t G, int dim) {
    if (G < 1 || dim < 1) return 0;
    return split_queries_bytes(G, dim);
}

int mi355_gallery_prepare(const float* gallery_normalized, int64_t G, int dim, void* planes, size_t planes_bytes, void* stream) {
    MI355_REQUIRE(gallery_normalized && planes, "gallery_prepare: null pointer");
    MI355_REQUIRE(G >= 1 && dim >= 1 && G < ((int64_t)1 << 31) - RK_BN, "gallery_prepare: bad shape G=%lld dim=%d", (long long)G, dim);
    MI355_REQUIRE(planes_bytes >= mi355_gallery_planes_bytes(G, dim), "gallery_prepare: planes buffer %zu < %zu bytes", planes_bytes,
                  mi355_gallery_planes_bytes(G, dim));
    const int n_steps = cdiv(dim, 16);
    const i64 n_frag = (i64)cdiv(G, 128) * 4 * n_steps;
    MI355_REQUIRE(n_frag < ((i64)1 << 31), "gallery_prepare: gallery too large for one call");
    hipLaunchKernelGGL(k_split_queries, dim3((unsigned)cdiv(n_frag, 4)), dim3(256), 0, (hipStream_t)stream, gallery_normalized,
                       (bf16_t*)planes, (int)G, dim, n_steps, (int)n_frag);
    MI355_LAUNCH_CHECK();
    return OK;
}

// mi355_rank_topk against a prepared gallery (rows normalised when the planes were made): k <= 8, Q > 4 (the fused selection's
// range; other shapes go through mi355_rank_topk with the fp32 rows).  Scores and indices are bit-identical to
// mi355_rank_topk(gallery_is_normalized = 1) on the same rows.  workspace: mi355_rank_workspace_bytes(Q, G, dim, k).
int mi355_rank_topk_prepared(const float* queries, int64_t Q, const void* gallery_planes, int64_t G, int dim, int k, float eps,
                             int64_t idx_offset, float* out_val, int64_t* out_idx, void* workspace, size_t workspace_bytes,
                             void* stream) {
    MI355_REQUIRE(queries && gallery_planes && out_val && out_idx, "rank_topk_prepared: null pointer");
    MI355_REQUIRE(Q >= 1 && G >= 1 && dim >= 1, "rank_topk_prepared: bad shape Q=%lld G=%lld dim=%d", (long long)Q, (long long)G, dim);
    MI355_REQUIRE(k >= 1 && k <= G, "rank_topk_prepared: k=%d outside [1, %lld]", k, (long long)G);
    MI355_REQUIRE(fused_select(Q, G, k), "rank_topk_prepared: needs k <= %d and more than 4 queries (got k=%d, Q=%lld)", SMALL_K, k, (long long)Q);
    hipStream_t st = (hipStream_t)stream;
    RankWs w = carve(workspace, Q, G, dim, k, false);
    MI355_REQUIRE(workspace && workspace_bytes >= w.total, "rank_topk_prepared: workspace %zu < %zu bytes", workspace_bytes, w.total);
    const int vq = vec_ok(queries, dim) && vec_ok(w.qn, dim);
    {
        RoctxRange range("rank/normalize");
        hipLaunchKernelGGL((k_row_norm<true>), dim3((unsigned)cdiv(Q, 4)), dim3(256), 0, st, queries, w.qn, (float*)nullptr, (i64)Q, dim, eps, vq);
        MI355_LAUNCH_CHECK();
    }
    const i64 qb = query_block(Q, G, k);
    const i64 ntiles = cdiv(G, RK_BN);
    const int n_steps = cdiv(dim, 16);
    for (i64 qs = 0; qs < Q; qs += qb) {
        const i64 qn = (Q - qs < qb) ? Q - qs : qb;
        {
            RoctxRange range("rank/cosine gemm (prepared gallery) + per-tile top-k");
            const int n_frag = cdiv(qn, 128) * 4 * n_steps;
            hipLaunchKernelGGL(k_split_queries, dim3((unsigned)cdiv(n_frag, 4)), dim3(256), 0, st, w.qn + qs * dim, w.qs, (int)qn, dim, n_steps, n_frag);
            MI355_LAUNCH_CHECK();
            const int e = qn > 64 ? launch_pre_fk<2>(w.qs, (const bf16_t*)gallery_planes, (int)qn, G, dim, k, w.cand_val, w.cand_idx, st)
                                  : launch_pre_fk<1>(w.qs, (const bf16_t*)gallery_planes, (int)qn, G, dim, k, w.cand_val, w.cand_idx, st);
            if (e) return e;
        }
        RoctxRange range("rank/merge candidates");
        if (int e = topk_select(w.cand_val, nullptr, qn, ntiles * k, ntiles * k, k, idx_offset, out_val + qs * k,
                                (i64*)out_idx + qs * k, w.topk, w.topk_bytes, st, w.cand_idx))
            return e;
    }
    return OK;
}

int mi355_topk_rows(const float* scores, int64_t Q, int64_t G, int k, int64_t idx_offset, float* out_val,
                    int64_t* out_idx, void* workspace, size_t workspace_bytes, void* stream) {
    MI355_REQUIRE(scores && out_val && out_idx, "topk_rows: null pointer");
    MI355_REQUIRE(Q >= 1 && G >= 1, "topk_rows: bad shape Q=%lld G=%lld", (long long)Q, (long long)G);
    MI355_REQUIRE(workspace, "topk_rows: null workspace");
    return topk_select(scores, nullptr, Q, G, G, k, idx_offset, out_val, (i64*)out_idx, workspace, workspace_bytes,
                       (hipStream_t)stream);
}

int mi355_merge_topk(const float* cand_val, const int64_t* cand_idx, int64_t Q, int ncand, int k, float* out_val,
                     int64_t* out_idx, void* workspace, size_t workspace_bytes, void* stream) {
    MI355_REQUIRE(cand_val && cand_idx && out_val && out_idx, "merge_topk: null pointer");
    MI355_REQUIRE(Q >= 1 && ncand >= 1, "merge_topk: bad shape Q=%lld ncand=%d", (long long)Q, ncand);
    MI355_REQUIRE(workspace, "merge_topk: null workspace");
    return topk_select(cand_val, (const i64*)cand_idx, Q, ncand, ncand, k, 0, out_val, (i64*)out_idx, workspace,
                       workspace_bytes, (hipStream_t)stream);
}

// ---- packed candidates of the sharded search (sharded.py): ONE int32 tensor per rank travels through the all-gather.
// packed[q][j] = {bits of the f32 score, LOCAL row index}; slots j >= kk (a shard with fewer than k rows) = {-inf, -1}.
int mi355_pack_candidates(const float* val, const int64_t* idx, int64_t Q, int kk, int k, int32_t* packed, void* stream) {
    MI355_REQUIRE(packed && Q >= 1 && k >= 1 && kk >= 0 && kk <= k, "pack_candidates: bad arguments Q=%lld kk=%d k=%d",
                  (long long)Q, kk, k);
    MI355_REQUIRE(kk == 0 || (val && idx), "pack_candidates: null candidates");
    const i64 n = (i64)Q * k;
    hipLaunchKernelGGL(k_pack_candidates, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, val,
                       (const i64*)idx, (i64)Q, kk, k, packed);
    MI355_LAUNCH_CHECK();
    return OK;
}

// Merge of the all-gathered packed candidates [world][Q][k][2]: the shard offsets (device, int64[world]) are added here and
// the world * k candidates of a query are merged with the rule of every other selection (higher score, then LOWER global
// index) -> (Q, k) identical to the unsharded result.  workspace: mi355_merge_packed_workspace_bytes(Q, world, k).
size_t mi355_merge_packed_workspace_bytes(int64_t Q, int world, int k) {
    if (Q < 1 || world < 1 || k < 1) return 0;
    const size_t n = (size_t)Q * world * k;
    return align_up(n * sizeof(i64), 256) + align_up(n * sizeof(float), 256) + topk_ws_bytes(Q, (i64)world * k, k) + 512;
}

int mi355_merge_packed_topk(const int32_t* packed, const int64_t* shard_offsets, int world, int64_t Q, int k,
                            float* out_val, int64_t* out_idx, void* workspace, size_t workspace_bytes, void* stream) {
    MI355_REQUIRE(packed && shard_offsets && out_val && out_idx, "merge_packed_topk: null pointer");
    MI355_REQUIRE(Q >= 1 && world >= 1 && k >= 1, "merge_packed_topk: bad shape Q=%lld world=%d k=%d", (long long)Q, world, k);
    MI355_REQUIRE(workspace && workspace_bytes >= mi355_merge_packed_workspace_bytes(Q, world, k),
                  "merge_packed_topk: workspace %zu < %zu bytes", workspace_bytes, mi355_merge_packed_workspace_bytes(Q, world, k));
    const size_t n = (size_t)Q * world * k;
    char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    i64* ci = (i64*)base;
    float* cv = (float*)(base + align_up(n * sizeof(i64), 256));
    char* rest = base + align_up(n * sizeof(i64), 256) + align_up(n * sizeof(float), 256);
    hipLaunchKernelGGL(k_unpack_candidates, dim3((unsigned)cdiv((i64)n, 256)), dim3(256), 0, (hipStream_t)stream, packed,
                       (const i64*)shard_offsets, world, (i64)Q, k, cv, ci);
    MI355_LAUNCH_CHECK();
    return topk_select(cv, ci, Q, (i64)world * k, (i64)world * k, k, 0, out_val, (i64*)out_idx, rest,
                       workspace_bytes - (size_t)(rest - (char*)workspace), (hipStream_t)stream);
}

int mi355_pair_cosine(const float* a, const float* b, int64_t rows, int dim, float eps, float* out, void* stream) {
    MI355_REQUIRE(a && b && out, "pair_cosine: null pointer");
    MI355_REQUIRE(rows >= 0 && dim >= 1, "pair_cosine: bad shape");
    if (rows == 0) return OK;
    hipLaunchKernelGGL(k_pair_cosine, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, a, b,
                       (i64)rows, dim, eps, out);
    MI355_LAUNCH_CHECK();
    return OK;
}

int mi355_contrastive_loss(const float* fm1, const float* fm2, int64_t rows, int dim, float label, float margin,
                           int mean, float* out, float* per_row, void* stream) {
    MI355_REQUIRE(fm1 && fm2 && out, "contrastive_loss: null pointer");
    MI355_REQUIRE(rows >= 1 && dim >= 1, "contrastive_loss: bad shape rows=%lld dim=%d", (long long)rows, dim);
    hipLaunchKernelGGL(k_contrastive, dim3(1), dim3(1024), 0, (hipStream_t)stream, fm1, fm2, (i64)rows, dim, label,
                       margin, mean, out, per_row);
    MI355_LAUNCH_CHECK();
    return OK;
}

int mi355_cosine_embedding_loss(const float* x1, const float* x2, int64_t rows, int dim, float target, float margin,
                                int mean, float* out, void* stream) {
    MI355_REQUIRE(x1 && x2 && out, "cosine_embedding_loss: null pointer");
    MI355_REQUIRE(rows >= 1 && dim >= 1, "cosine_embedding_loss: bad shape rows=%lld dim=%d", (long long)rows, dim);
    MI355_REQUIRE(target == 1.0f || target == -1.0f, "cosine_embedding_loss: target must be +1 or -1");
    hipLaunchKernelGGL(k_cos_embedding_loss, dim3(1), dim3(1024), 0, (hipStream_t)stream, x1, x2, (i64)rows, dim, target,
                       margin, mean, out);
    MI355_LAUNCH_CHECK();
    return OK;
}

int mi355_hit_counts(const int64_t* idx, int64_t Q, int k, const int64_t* query_cls, const int64_t* gallery_cls,
                     int64_t G, int64_t* counts, void* stream) {
    MI355_REQUIRE(idx && query_cls && gallery_cls && counts, "hit_counts: null pointer");
    MI355_REQUIRE(Q >= 1 && k >= 1 && G >= 1, "hit_counts: bad shape");
    hipLaunchKernelGGL(k_hit_counts, dim3((unsigned)cdiv(Q, 256)), dim3(256), 0, (hipStream_t)stream, (const i64*)idx,
                       (i64)Q, k, (const i64*)query_cls, (const i64*)gallery_cls, (i64)G, (i64*)counts);
    MI355_LAUNCH_CHECK();
    return OK;
}

int mi355_distinct_class_topn(const int64_t* idx, const float* val, int64_t Q, int k, const int64_t* gallery_cls,
                              int64_t G, int n, int64_t* out_cls, int64_t* out_idx, float* out_val, void* stream) {
    MI355_REQUIRE(idx && val && gallery_cls && out_cls && out_idx && out_val, "distinct_class_topn: null pointer");
    MI355_REQUIRE(Q >= 1 && k >= 1 && G >= 1 && n >= 1 && n <= 8, "distinct_class_topn: bad shape (n must be 1..8)");
    hipLaunchKernelGGL(k_distinct_topn, dim3((unsigned)cdiv(Q, 128)), dim3(128), 0, (hipStream_t)stream,
                       (const i64*)idx, val, (i64)Q, k, (const i64*)gallery_cls, (i64)G, n, (i64*)out_cls, (i64*)out_idx,
                       out_val);
    MI355_LAUNCH_CHECK();
    return OK;
}

}  // extern "C"
