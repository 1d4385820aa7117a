// Shared host/device helpers for libmi355_retrieval (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

namespace mi355 {

// ---- error plumbing: every C-ABI entry returns 0 on success, nonzero + mi355_last_error() text.
void set_error(const char* fmt, ...);
// roctx ranges around executor ops / rank phases (api.cpp; no-ops unless enabled and the marker library is present)
void roctx_enable(bool on);
bool roctx_active();
void roctx_push(const char* label);
void roctx_pop();
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char* label) : on(roctx_active()) { if (on) roctx_push(label); }
    ~RoctxRange() { if (on) roctx_pop(); }
};
enum { OK = 0, ERR_ARG = 1, ERR_HIP = 2, ERR_STATE = 3, ERR_UNSUPPORTED = 4 };

#define MI355_CHECK_HIP(expr)                                                              \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            mi355::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return mi355::ERR_HIP;                                                         \
        }                                                                                  \
    } while (0)

#define MI355_REQUIRE(cond, ...)                                                           \
    do {                                                                                   \
        if (!(cond)) {                                                                     \
            mi355::set_error(__VA_ARGS__);                                                 \
            return mi355::ERR_ARG;                                                         \
        }                                                                                  \
    } while (0)

#define MI355_LAUNCH_CHECK() MI355_CHECK_HIP(hipGetLastError())

// Per-device one-shot (e.g. hipFuncSetAttribute, which applies to the code object loaded on ONE device): true the first
// time it is asked for the current device.  `flags` is a static bool[MI355_MAX_DEVICES] owned by the call site.
constexpr int MI355_MAX_DEVICES = 64;
static inline bool first_time_on_this_device(bool* flags) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MI355_MAX_DEVICES) return true;   // unknown: just redo it
    if (flags[dev]) return false;
    flags[dev] = true;
    return true;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// ---- bf16 <-> f32 bit helpers (device).  Plain casts so NaN stays NaN (v_cvt_pk_bf16_f32).
typedef unsigned short bf16_t;  // storage type for bf16 tensors across the library

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    __hip_bfloat16 h = __float2bfloat16(f);  // round-to-nearest-even
    return *reinterpret_cast<bf16_t*>(&h);
}
// two fp32 -> one dword of two bf16: ONE v_cvt_pk_bf16_f32.  (Written as a vector convert: the scalar form
// f2bf(lo) | f2bf(hi) << 16 let hipcc pair the producers of lo / hi with those of another pack2bf call and then spend
// and/shift/or-sdwa instructions re-ordering the halves - 6 VALU ops per 4 values in the VALU-bound epilogues.)
typedef __bf16 mi355_bf16x2 __attribute__((ext_vector_type(2)));
typedef float mi355_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    const mi355_bf16x2 v = __builtin_convertvector((mi355_f32x2){lo, hi}, mi355_bf16x2);
    return *reinterpret_cast<const unsigned*>(&v);
}

// sigmoid/SiLU on the transcendental pipe: exp2 + rcp (1 ulp each), no IEEE division sequence.  Every result is
// rounded to bf16 (8 bits) right after, so the ~1e-7 relative difference to expf()/div is invisible.
__device__ __forceinline__ float sigmoid_f(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }

// exact-erf GELU with erf from Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7, far below the bf16 rounding that
// follows): rcp + exp2 + 5 FMAs instead of libm erff (~3x the instructions; 50 us per Swin fc1 epilogue).
__device__ __forceinline__ float gelu_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const float e = 1.0f - poly * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);   // erf(|x|/sqrt2)
    return 0.5f * x * (1.0f + copysignf(e, x));
}

// activation codes shared by the conv / GEMM epilogues
enum Act { ACT_NONE = 0, ACT_SILU = 1, ACT_RELU = 2, ACT_RELU6 = 3, ACT_GELU = 4, ACT_SIGMOID = 5 };

__device__ __forceinline__ float apply_act(float x, int act) {
    switch (act) {
        case ACT_SILU: return silu_f(x);
        case ACT_RELU: return fmaxf(x, 0.f);
        case ACT_RELU6: return fminf(fmaxf(x, 0.f), 6.f);
        case ACT_GELU: return gelu_f(x);
        case ACT_SIGMOID: return sigmoid_f(x);
        default: return x;
    }
}

// Compile-time activation.  A runtime `switch (act)` inside an unrolled epilogue is expanded per value: the
// NT=12 GEMM had 2813 scalar branches and 24k instructions (196 KB of code, far beyond the instruction cache) and
// its epilogue ran at 2.2 TB/s where plain stores do 6 TB/s.  MI355_ACT_DISPATCH hoists the switch out once and
// compiles BODY straight-line for the common activations (ACT is a constexpr int inside BODY).
template <int ACT>
__device__ __forceinline__ float act_c(float x) {
    if constexpr (ACT == ACT_SILU) return silu_f(x);
    else if constexpr (ACT == ACT_RELU) return fmaxf(x, 0.f);
    else if constexpr (ACT == ACT_RELU6) return fminf(fmaxf(x, 0.f), 6.f);
    else if constexpr (ACT == ACT_GELU) return gelu_f(x);
    else if constexpr (ACT == ACT_SIGMOID) return sigmoid_f(x);
    else return x;
}
#define MI355_ACT_DISPATCH(act_runtime, ...)                                      \
    switch (act_runtime) {                                                        \
        case ::mi355::ACT_SILU: { constexpr int ACT = ::mi355::ACT_SILU; __VA_ARGS__ } break;   \
        case ::mi355::ACT_GELU: { constexpr int ACT = ::mi355::ACT_GELU; __VA_ARGS__ } break;   \
        case ::mi355::ACT_RELU6: { constexpr int ACT = ::mi355::ACT_RELU6; __VA_ARGS__ } break; \
        case ::mi355::ACT_RELU: { constexpr int ACT = ::mi355::ACT_RELU; __VA_ARGS__ } break;   \
        case ::mi355::ACT_SIGMOID: { constexpr int ACT = ::mi355::ACT_SIGMOID; __VA_ARGS__ } break; \
        default: { constexpr int ACT = ::mi355::ACT_NONE; __VA_ARGS__ } break;    \
    }

// wave64 reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace mi355
