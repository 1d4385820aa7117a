// Portable counter-based generator on device; bit-identical to imageretrievalresearch_amd/synth.py.
#include "common.h"
#include "../../include/mi355_retrieval.h"

namespace mi355 {
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void k_synth_fill(float* __restrict__ out, int64_t n, uint64_t key, int64_t offset,
                                                    int kind, float scale) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t h = splitmix64(key + (uint64_t)(offset + i));
        float v;
        if (kind == 0) {
            v = (float)(h >> 40) * scale;
        } else {
            const int t = (int)((h & 0xFFFF) + ((h >> 16) & 0xFFFF) + ((h >> 32) & 0xFFFF) + (h >> 48)) - 131070;
            v = (float)t * scale;
        }
        out[i] = v;
    }
}
}  // namespace mi355

extern "C" int mi355_synth_fill(float* out, int64_t n, uint64_t seed, int64_t offset, int kind, void* stream) {
    using namespace mi355;
    MI355_REQUIRE(out || n == 0, "synth_fill: null output");
    MI355_REQUIRE(n >= 0 && (kind == 0 || kind == 1), "synth_fill: bad n/kind");
    if (n == 0) return OK;
    // scales are the float32 constants of synth.py (computed in double there, rounded once)
    const float scale = kind == 0 ? 5.9604644775390625e-08f
                                  : (float)(1.0 / __builtin_sqrt(4.0 * (65536.0 * 65536.0 - 1.0) / 12.0));
    const uint64_t key = splitmix64(seed);
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_synth_fill, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, (int64_t)n, key,
                       (int64_t)offset, kind, scale);
    MI355_LAUNCH_CHECK();
    return OK;
}
