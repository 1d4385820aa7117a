// Store bandwidth of GEMM-epilogue-shaped writes (developer tool): a block writes a tile of ROWS x SEG bytes into a matrix
// whose rows are `pitch` bytes apart; blocks walk column tiles first (as k_gemm_big's tile order does).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_store_tiles.hip -o tools/store_tiles.bin && tools/store_tiles.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void wr(char* out, int rows, int seg, long pitch, int n_tiles) {
    const int mb = blockIdx.x / n_tiles, nb = blockIdx.x - mb * n_tiles;
    const int cpr = seg / 16;                       // 16-byte chunks per row segment
    u32x4 v = {threadIdx.x, blockIdx.x, 0u, 1u};
    for (int id = threadIdx.x; id < rows * cpr; id += 256) {
        const int r = id / cpr, c = id - r * cpr;
        *reinterpret_cast<u32x4*>(out + ((long)mb * rows + r) * pitch + (long)nb * seg + c * 16) = v;
    }
}
static void run(long M, long pitch, int rows, int seg) {
    char* buf; hipMalloc(&buf, M * pitch);
    const int n_tiles = pitch / seg, m_tiles = M / rows;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(wr, dim3(n_tiles * m_tiles), dim3(256), 0, 0, buf, rows, seg, pitch, n_tiles);
    hipEventRecord(e0);
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(wr, dim3(n_tiles * m_tiles), dim3(256), 0, 0, buf, rows, seg, pitch, n_tiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("M=%ld pitch=%5ld B  tile %3d rows x %4d B  (%d x %d blocks)  %.1f us  %.0f GB/s\n", M, pitch, rows, seg, m_tiles, n_tiles,
           ms * 1e3, M * pitch / ms / 1e6);
    hipFree(buf);
}
int main() {
    const long M = 401408;
    run(M, 256, 128, 256);
    run(M, 768, 128, 256);
    run(M, 1024, 128, 256);
    run(M, 1024, 64, 512);
    run(M, 1024, 32, 1024);
    run(M, 1024, 128, 512);
    run(M, 1024, 128, 1024);
    run(M, 2048, 128, 256);
    run(M, 2048, 64, 512);
    run(M, 2048, 32, 1024);
    run(100352, 4096, 128, 256);
    run(100352, 4096, 32, 1024);
    return 0;
}
