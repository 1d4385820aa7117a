#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// each block writes PER_BLOCK bytes contiguous: thread t writes chunk t + 256*i
template <int ITERS>
__global__ __launch_bounds__(256) void wr(u32x4* out, int lds_dummy) {
    extern __shared__ char sm[];
    if (lds_dummy == 12345) sm[threadIdx.x] = 1;
    u32x4 v = {threadIdx.x, blockIdx.x, 0u, 1u};
    u32x4* o = out + (size_t)blockIdx.x * 256 * ITERS + threadIdx.x;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) o[256 * i] = v;
}
template <int ITERS> void run(size_t bytes, int lds, const char* tag) {
    u32x4* buf; hipMalloc(&buf, bytes);
    int blocks = bytes / (256 * ITERS * 16);
    hipFuncSetAttribute((const void*)wr<ITERS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(wr<ITERS>, dim3(blocks), dim3(256), lds, 0, buf, 0);
    hipEventRecord(e0);
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(wr<ITERS>, dim3(blocks), dim3(256), lds, 0, buf, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("%s iters=%d lds=%6d blocks=%d  %.1f us  %.0f GB/s\n", tag, ITERS, lds, blocks, ms * 1e3, bytes / ms / 1e6);
    hipFree(buf);
}
int main() {
    size_t bytes = 308ull << 20;
    run<12>(bytes, 0, "occ-free ");
    run<12>(bytes, 52 * 1024, "3blk/CU  ");
    run<12>(bytes, 70 * 1024, "2blk/CU  ");
    run<12>(bytes, 100 * 1024, "1blk/CU  ");
    run<3>(bytes, 0, "occ-free ");
    run<3>(bytes, 22 * 1024, "7blk/CU  ");
    run<48>(bytes, 70 * 1024, "2blk/CU  ");
    return 0;
}
