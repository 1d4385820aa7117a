// Developer microbenchmark: sustained rate of v_mfma_f32_32x32x2_f32 with nothing else in the loop (no LDS, no HBM),
// to separate "the MFMA pipe at the clock the chip sustains" from kernel overheads.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 512, 1024}) {
        const int iters = 20000;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 100, 1.f, 1.f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 1.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * 4 * iters * 8 * 32 * 32 * 2 * 2;
        printf("blocks=%d (x4 waves): %.3f ms  %.1f TFLOP/s\n", blocks, ms, flops / ms / 1e9);
    }
    return 0;
}
