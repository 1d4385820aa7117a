// Developer microbenchmark: issue cost of the SiLU building blocks (v_exp_f32 / v_rcp_f32 vs v_fma_f32 / v_pk_fma_f32)
// with 8 waves per CU (2 per SIMD) of independent chains.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, float s) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) v[i] = v[i] * s + 0.5f;                                   // fma
            if (MODE == 1) v[i] = __builtin_amdgcn_exp2f(v[i] * s);                  // mul + exp
            if (MODE == 2) v[i] = __builtin_amdgcn_rcpf(v[i] + s);                   // add + rcp
            if (MODE == 3) v[i] = v[i] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v[i])) + s;  // SiLU + add
        }
        if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 8; i += 2) { f2 t = {v[i], v[i + 1]}; t = t * s + 0.5f; v[i] = t.x; v[i + 1] = t.y; }   // pk_fma
        }
    }
    float acc = 0.f;
    for (int i = 0; i < 8; ++i) acc += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}
template <int MODE>
static void run(const char* name, float* out, int per_iter_ops) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096, blocks = 1024;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, 16, 0.999f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, iters, 0.999f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double elems = (double)blocks * 512 * iters * 8;
    // cycles per wave-level "element op" per SIMD: 1024 SIMDs at 2.4 GHz
    const double cyc = ms * 1e-3 * 2.4e9 * 1024 / (elems / 64);
    printf("%-28s %.3f ms  %.2f cycles per wave-element (%d source ops)\n", name, ms, cyc, per_iter_ops);
}
int main() {
    float* out; (void)hipMalloc(&out, 1024 * 512 * 4);
    run<0>("fma", out, 1);
    run<4>("pk_fma (per element)", out, 1);
    run<1>("mul + exp2", out, 2);
    run<2>("add + rcp", out, 2);
    run<3>("silu + add", out, 5);
    return 0;
}
