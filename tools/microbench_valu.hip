// Developer microbenchmark: VALU issue cost of the building blocks of the fused MBConv kernels on gfx950, at 1 / 2 / 4
// waves per SIMD (one workgroup of 256 / 512 / 1024 threads per CU, pinned by a 128 KB LDS request).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_valu.hip -o tools/valu.bin
// Prints cycles per wave-instruction per SIMD (2.4 GHz nominal) for: fma, pk_fma, exp2, rcp, SiLU, v_dot2_f32_bf16,
// v_dot2c_f32_bf16, bf16 unpack, v_perm_b32, v_cvt_pk_bf16_f32, and FMA/transcendental mixes.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));

constexpr int NCH = 16;

template <int MODE>
__global__ void k(float* out, int iters, float s, unsigned us) {
    extern __shared__ float lds[];
    float v[NCH];
    unsigned u[NCH];
    for (int i = 0; i < NCH; ++i) { v[i] = threadIdx.x * 1e-3f + i; u[i] = threadIdx.x * 2654435761u + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (MODE == 0) v[i] = v[i] * s + 0.5f;
            if (MODE == 2) v[i] = __builtin_amdgcn_exp2f(v[i]);
            if (MODE == 3) v[i] = __builtin_amdgcn_rcpf(v[i]);
            if (MODE == 4) v[i] = v[i] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v[i]));
            if (MODE == 5) v[i] = __builtin_amdgcn_fdot2_f32_bf16(*(bf2*)&u[i], *(bf2*)&us, v[i], false);
            if (MODE == 6) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(v[i]) : "v"(u[i]), "v"(us));
            if (MODE == 7) { v[i] += __uint_as_float(u[i] << 16); }                 // shift + add
            if (MODE == 8) u[i] = __builtin_amdgcn_perm(u[i], us, 0x07060302u);
            if (MODE == 9) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(v[i]), "v"(s));
            if (MODE == 10) { v[i] = v[i] * s + 0.5f; if ((i & 3) == 0) v[i] = __builtin_amdgcn_exp2f(v[i]); }   // 4 fma : 1 exp
            if (MODE == 11) { v[i] = v[i] * s + 0.5f; if ((i & 1) == 0) v[i] = __builtin_amdgcn_exp2f(v[i]); }   // 2 fma : 1 exp
        }
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < NCH; i += 2) { f2 t = {v[i], v[i + 1]}; t = t * s + 0.5f; v[i] = t.x; v[i + 1] = t.y; }
        }
    }
    float acc = 0.f;
    for (int i = 0; i < NCH; ++i) acc += v[i] + __uint_as_float(u[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + lds[threadIdx.x];
}

template <int MODE>
static void run(const char* name, float* out, double insts_per_chain) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    printf("%-34s", name);
    for (int threads : {256, 512, 1024}) {
        const int iters = 2048, blocks = 256;
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 128 * 1024, 0, out, 16, 0.999f, 0x3f803f80u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 128 * 1024, 0, out, iters, 0.999f, 0x3f803f80u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double winst = (double)blocks * (threads / 64) * iters * NCH * insts_per_chain;   // wave-instructions
        const double cyc = ms * 1e-3 * 2.4e9 * 1024 / winst;   // SIMD-cycles per wave-instruction
        printf("  %dw/SIMD: %6.2f cyc", threads / 256, cyc);
    }
    printf("   (per instruction, %g inst/chain-step)\n", insts_per_chain);
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 1024 * 4);
    run<0>("v_fma_f32", out, 1);
    run<1>("v_pk_fma_f32 (per pk inst)", out, 0.5);
    run<2>("v_exp_f32", out, 1);
    run<3>("v_rcp_f32", out, 1);
    run<4>("silu: mul exp add rcp mul", out, 5);
    run<5>("v_dot2_f32_bf16 (builtin)", out, 1);
    run<6>("v_dot2c_f32_bf16", out, 1);
    run<7>("lshl + add", out, 2);
    run<8>("v_perm_b32", out, 1);
    run<9>("v_cvt_pk_bf16_f32", out, 1);
    run<10>("4 fma : 1 exp (per inst)", out, 1.25);
    run<11>("2 fma : 1 exp (per inst)", out, 1.5);
    return 0;
}
