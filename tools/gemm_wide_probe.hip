// Developer probe (standalone): the wide-tile GEMM kernel compiled WITH its phase stamps, run on a few Swin shapes; prints the mean
// cycles wave 0 of a workgroup spends per phase.  Build (from the repo root):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DGW_STAMPS -DMI355_DW_PX=4 -Iimageretrievalresearch_amd/csrc tools/gemm_wide_probe.hip -o tools/gemm_wide_probe
#include <stdarg.h>
#include "gemm_wide.hip"
#include <vector>
namespace mi355 { void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); printf("\n"); }
bool roctx_active() { return false; } void roctx_push(const char*) {} void roctx_pop() {} void roctx_enable(bool) {} }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main(int argc, char** argv) {
    using namespace mi355;
    struct Sh { int M, N, K, act, abl; } shapes[] = {{25088, 2048, 512, 0, 0}, {25088, 2048, 512, 0, 100}, {25088, 2048, 512, 0, 200}, {25088, 2048, 512, 0, 300}, {25088, 1536, 512, 0, 0}, {25088, 1536, 512, 0, 150}, {25088, 2048, 512, 4, 150}};
    for (auto sh : shapes) {
        bf16_t *A, *W, *out, *zeros; float* bias; unsigned long long* st;
        CK(hipMalloc(&A, (size_t)sh.M * sh.K * 2)); CK(hipMalloc(&W, (size_t)sh.N * sh.K * 2)); CK(hipMalloc(&out, (size_t)sh.M * sh.N * 2));
        CK(hipMalloc(&bias, sh.N * 4)); CK(hipMalloc(&zeros, 256)); CK(hipMalloc(&st, 256 * 64));
        CK(hipMemset(A, 0x3c, (size_t)sh.M * sh.K * 2)); CK(hipMemset(W, 0x3c, (size_t)sh.N * sh.K * 2)); CK(hipMemset(bias, 0, sh.N * 4)); CK(hipMemset(zeros, 0, 256));
        CK(hipMemset(st, 0, 256 * 64));
        GemmArgs a{};
        a.A = A; a.lda = sh.K; a.W = W; a.ldw = sh.K; a.bias = bias; a.out = out; a.ldo = sh.N; a.M = sh.M; a.N = sh.N; a.K = sh.K; a.act = sh.act;
        a.rows_per_img = 1; a.res_n = sh.N; a.gate_ld = sh.abl; a.zeros = zeros; a.splitk_ws = (float*)st;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) if (launch_gemm_wide(a, 0)) return 1;
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) launch_gemm_wide(a, 0);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        std::vector<unsigned long long> h(256 * 8);
        CK(hipMemcpy(h.data(), st, 256 * 64, hipMemcpyDeviceToHost));
        double s[8] = {0}; int n = 0;
        for (int b = 0; b < 256; ++b) if (h[b * 8 + 5]) { for (int i = 0; i < 8; ++i) s[i] += (double)h[b * 8 + i]; ++n; }
        const int mi = gemm_wide_pick_mi(sh.M, sh.N, 256);
        const double steps = s[6] / n * (sh.K / 32);
        printf("M=%d N=%d K=%d act=%d abl=%d MI=%d: %.1f us %.0f TF | per workgroup (mean of %d): tiles %.2f, total %.0f cycles; per step: wait+barrier %.0f, issue %.0f, mfma+reads %.0f; per tile: epilogue %.0f, top reads %.0f\n",
               sh.M, sh.N, sh.K, sh.act, sh.abl, mi, ms * 1e3, 2.0 * sh.M * sh.N * sh.K / ms / 1e9, n, s[6] / n, s[5] / n, s[0] / n / steps, s[1] / n / steps, s[2] / n / steps,
               s[3] / s[6], s[4] / s[6]);
        hipFree(A); hipFree(W); hipFree(out); hipFree(bias); hipFree(zeros); hipFree(st);
    }
    return 0;
}
