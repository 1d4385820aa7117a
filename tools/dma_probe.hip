// Developer probe (standalone, not part of the library): how fast does one CU take bytes in by LDS-DMA (global_load_lds, 16 B per
// lane) as a function of the source pattern (bytes per row segment), the number of pieces kept in flight and where the data lives?
// Build: hipcc -O3 --offload-arch=gfx950 tools/dma_probe.hip -o tools/dma_probe ; run on the GPU box: tools/dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// SEG = bytes of one row segment (64, 128, 256 or 1024 = fully contiguous piece); INFL = pieces each wave keeps in flight (1..32);
// WAVES = loader waves per workgroup.  Every wave streams `steps` x INFL pieces of 1 KB; the region a workgroup walks over is
// `span` bytes long (wraps), rows are `pitch` bytes apart.  One workgroup per CU (LDS 128 KB).
template <int SEG, int INFL>
__global__ __launch_bounds__(512) void k_probe(const char* __restrict__ src, size_t wg_stride, size_t span, int pitch, int steps, int barrier) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    constexpr int LPR = SEG / 16;              // lanes per row segment
    constexpr int RPP = 64 / LPR;              // rows per piece
    const char* base = src + (size_t)blockIdx.x * wg_stride;
    // piece p of this wave: rows (p * nw + wave) * RPP ..; along a row the segments of consecutive k-steps follow each other
    const int rows_total = (int)(span / pitch);                  // rows in the region
    const int segs_per_row = pitch / SEG;
    size_t lane_off = (size_t)(lane / LPR) * pitch + (size_t)(lane % LPR) * 16;
    int piece = 0;
    char* dst = lds + wave * 8192;
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int i = 0; i < INFL; ++i) {
            // walk: consecutive pieces of a wave go down the rows (GEMM-like: a stage = many rows x one segment), then to the next segment
            const int idx = piece * nw + wave;
            const int rb = (idx * RPP) % rows_total;
            const int sg = ((idx * RPP) / rows_total) % segs_per_row;
            glds16(base + (size_t)rb * pitch + (size_t)sg * SEG + lane_off, dst + (i & 7) * 1024);
            ++piece;
        }
        if (barrier) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else if (INFL >= 2) {
            // keep half in flight
            if constexpr (INFL == 32) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if constexpr (INFL == 16) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (INFL == 8) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr (INFL == 4) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int SEG, int INFL>
static double run(const char* src, int waves, size_t wg_stride, size_t span, int pitch, int steps, int barrier, int grid) {
    CHECK(hipFuncSetAttribute((const void*)k_probe<SEG, INFL>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_probe<SEG, INFL>), dim3(grid), dim3(waves * 64), 131072, 0, src, wg_stride, span, pitch, steps, barrier);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_probe<SEG, INFL>), dim3(grid), dim3(waves * 64), 131072, 0, src, wg_stride, span, pitch, steps, barrier);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes_per_wg = (double)steps * INFL * 1024.0 * waves;
    return bytes_per_wg / (ms / reps * 1e-3) / 1e9;   // GB/s per workgroup (= per CU at one workgroup per CU)
}

int main() {
    const size_t total = (size_t)1 << 30;
    char* buf;
    CHECK(hipMalloc(&buf, total));
    CHECK(hipMemset(buf, 1, total));
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int grid = p.multiProcessorCount;
    printf("CUs %d\n", grid);
    struct Foot { const char* name; size_t wg_stride, span; } foots[] = {
        {"shared 256 KB (every CU the same rows: L2 hits)", 0, 256 << 10},
        {"64 KB per CU (16 MB total: L2 / MALL)", 64 << 10, 64 << 10},
        {"512 KB per CU, 8 neighbours share (GEMM-like A panels)", 0, 0},   // filled below
        {"2 MB per CU (512 MB total: HBM)", 2 << 20, 2 << 20},
    };
    const int pitch = 1024;   // bytes between rows (K = 512 bf16)
    for (int f = 0; f < 4; ++f) {
        size_t wg_stride = foots[f].wg_stride, span = foots[f].span;
        if (f == 2) { wg_stride = (512 << 10) / 8; span = 512 << 10; }
        printf("== %s\n", foots[f].name);
        for (int waves : {4, 8}) {
            const int steps = 400;
#define ROW(SEG) printf("  waves %d seg %4d B: in flight/wave 2: %6.1f  4: %6.1f  8: %6.1f  16: %6.1f  32: %6.1f  | barrier per 8: %6.1f  per 16: %6.1f GB/s per CU\n", waves, SEG, \
            run<SEG, 2>(buf, waves, wg_stride, span, pitch, steps * 8, 0, grid), run<SEG, 4>(buf, waves, wg_stride, span, pitch, steps * 4, 0, grid), \
            run<SEG, 8>(buf, waves, wg_stride, span, pitch, steps * 2, 0, grid), run<SEG, 16>(buf, waves, wg_stride, span, pitch, steps, 0, grid), \
            run<SEG, 32>(buf, waves, wg_stride, span, pitch, steps / 2, 0, grid), \
            run<SEG, 8>(buf, waves, wg_stride, span, pitch, steps * 2, 1, grid), run<SEG, 16>(buf, waves, wg_stride, span, pitch, steps, 1, grid));
            ROW(64) ROW(128) ROW(256) ROW(1024)
            fflush(stdout);
        }
    }
    return 0;
}
