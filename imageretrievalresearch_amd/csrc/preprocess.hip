// Training-side pre-processing and score post-processing that sit either side of the hot path.
//
//   mi355_resize_bilinear_u8   transforms.Resize((h, w)) of train/train.py:48-50 on a PIL image ==
//                              Pillow's two-pass antialiased BILINEAR resample (Resample.c, 8-bit path): per-axis
//                              triangle-filter coefficients in double, fixed point 2^22, uint8 between the passes.
//                              The coefficient tables are built on the host exactly as Pillow builds them and cached on
//                              the device per (device, in, out); the passes are integer MACs => bit-exact with Pillow.
//   mi355_score_boost          utils/score_booster.py:1-37 over a whole score tensor.
#include "common.h"
#include "../../include/mi355_retrieval.h"
#include <map>
#include <math.h>
#include <mutex>
#include <tuple>
#include <vector>

namespace mi355 {

constexpr int RS_PRECISION_BITS = 32 - 8 - 2;

struct ResizeCoeffs {
    int ksize = 0;
    std::vector<int> host;   // [out][2 + ksize]: first source index, tap count, taps (kept alive for the async upload)
    int* dev = nullptr;
};

// Pillow precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle, support 1.0) filter, box = (0, in).
static void build_coeffs(int in_size, int out_size, ResizeCoeffs& rc) {
    const double scale = (double)((float)in_size - 0.0f) / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    rc.ksize = ksize;
    rc.host.assign((size_t)out_size * (2 + ksize), 0);
    std::vector<double> k(ksize);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = 0.0 + (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        int x = 0;
        for (; x < xmax; ++x) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            const double w = a < 1.0 ? 1.0 - a : 0.0;
            k[x] = w;
            ww += w;
        }
        for (x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        for (; x < ksize; ++x) k[x] = 0.0;
        int* row = rc.host.data() + (size_t)xx * (2 + ksize);
        row[0] = xmin;
        row[1] = xmax;
        for (x = 0; x < ksize; ++x)
            row[2 + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << RS_PRECISION_BITS)) : (int)(0.5 + k[x] * (1 << RS_PRECISION_BITS));
    }
}

static std::mutex g_rs_mu;
static std::map<std::tuple<int, int, int>, ResizeCoeffs> g_rs_cache;

static int get_coeffs(int in_size, int out_size, hipStream_t st, const ResizeCoeffs** out) {
    int dev = 0;
    MI355_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_rs_mu);
    auto key = std::make_tuple(dev, in_size, out_size);
    auto it = g_rs_cache.find(key);
    if (it == g_rs_cache.end()) {
        ResizeCoeffs rc;
        build_coeffs(in_size, out_size, rc);
        MI355_CHECK_HIP(hipMalloc((void**)&rc.dev, rc.host.size() * sizeof(int)));
        it = g_rs_cache.emplace(key, std::move(rc)).first;
        MI355_CHECK_HIP(hipMemcpyAsync(it->second.dev, it->second.host.data(), it->second.host.size() * sizeof(int),
                                       hipMemcpyHostToDevice, st));
        // one-time upload: finish it before the table is published, so that a later caller on ANOTHER stream never
        // reads a table whose copy is still queued behind the first caller's stream
        MI355_CHECK_HIP(hipStreamSynchronize(st));
    }
    *out = &it->second;
    return OK;
}

// One pass along x: out[r][ox][c] = clip8((2^21 + sum_i in[r0 + r][xmin + i][c] * k_i) >> 22).  thread = (row, ox).
__global__ __launch_bounds__(256) void k_resize_h(const unsigned char* __restrict__ in, int w, int row0, int rows,
                                                  unsigned char* __restrict__ out, int out_w,
                                                  const int* __restrict__ coef, int ksize) {
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int r = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox >= out_w || r >= rows) return;
    const int* row = coef + (size_t)ox * (2 + ksize);
    const int xmin = row[0], n = row[1];
    int a0 = 1 << (RS_PRECISION_BITS - 1), a1 = a0, a2 = a0;
    const unsigned char* p = in + ((size_t)(row0 + r) * w + xmin) * 3;
    for (int i = 0; i < n; ++i) {
        const int k = row[2 + i];
        a0 += (int)p[i * 3 + 0] * k;
        a1 += (int)p[i * 3 + 1] * k;
        a2 += (int)p[i * 3 + 2] * k;
    }
    unsigned char* o = out + ((size_t)r * out_w + ox) * 3;
    o[0] = (unsigned char)min(max(a0 >> RS_PRECISION_BITS, 0), 255);
    o[1] = (unsigned char)min(max(a1 >> RS_PRECISION_BITS, 0), 255);
    o[2] = (unsigned char)min(max(a2 >> RS_PRECISION_BITS, 0), 255);
}

// One pass along y over an image of `w` pixels per row: thread = (oy, byte of the row), coalesced along the row.
__global__ __launch_bounds__(256) void k_resize_v(const unsigned char* __restrict__ in, int w, int row_shift,
                                                  unsigned char* __restrict__ out, int out_h,
                                                  const int* __restrict__ coef, int ksize) {
    const int xb = blockIdx.x * 256 + threadIdx.x;   // byte within the row (3 * w of them)
    const int oy = blockIdx.y;
    if (xb >= 3 * w || oy >= out_h) return;
    const int* row = coef + (size_t)oy * (2 + ksize);
    const int ymin = row[0] - row_shift, n = row[1];
    int acc = 1 << (RS_PRECISION_BITS - 1);
    for (int i = 0; i < n; ++i) acc += (int)in[(size_t)(ymin + i) * 3 * w + xb] * row[2 + i];
    out[(size_t)oy * 3 * w + xb] = (unsigned char)min(max(acc >> RS_PRECISION_BITS, 0), 255);
}

__global__ __launch_bounds__(256) void k_score_boost(const float* __restrict__ s, float* __restrict__ out, long n,
                                                     float eps, float alpha, float threshold, int mode) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = s[i];
    // mode 0: by threshold; 1: "for_pos"; 2: "for_neg" (utils/score_booster.py:17-20, 33-36); same fp32 op order
    const bool pos = mode == 1 || (mode == 0 && v >= threshold);
    const bool neg = mode == 2 || (mode == 0 && v < threshold);
    float r = v;                                     // NaN under mode 0 matches neither branch (python returns None)
    if (pos) r = (v + eps) / (eps + alpha);
    else if (neg) r = fabsf((v + (alpha / eps)) / (2.0f * eps));
    out[i] = r;
}

}  // namespace mi355

using namespace mi355;

extern "C" {

int mi355_resize_bilinear_u8(const unsigned char* img, int h, int w, unsigned char* out, int out_h, int out_w,
                             unsigned char* tmp, void* stream) {
    MI355_REQUIRE(img && out, "resize: null pointer");
    MI355_REQUIRE(h >= 1 && w >= 1 && h <= 16384 && w <= 16384, "resize: bad input size %dx%d", h, w);
    MI355_REQUIRE(out_h >= 1 && out_w >= 1 && out_h <= 16384 && out_w <= 16384, "resize: bad output size %dx%d", out_h, out_w);
    hipStream_t st = (hipStream_t)stream;
    const bool need_h = w != out_w, need_v = h != out_h;
    if (!need_h && !need_v) {   // PIL returns a copy
        MI355_CHECK_HIP(hipMemcpyAsync(out, img, (size_t)h * w * 3, hipMemcpyDeviceToDevice, st));
        return OK;
    }
    const ResizeCoeffs *ch = nullptr, *cv = nullptr;
    if (need_h) { const int e = get_coeffs(w, out_w, st, &ch); if (e) return e; }
    if (need_v) { const int e = get_coeffs(h, out_h, st, &cv); if (e) return e; }
    int first = 0, last = h;
    if (need_v) {   // the horizontal pass only produces the rows the vertical pass reads (ybox_first .. ybox_last)
        const int stride = 2 + cv->ksize;
        first = cv->host[0];
        last = cv->host[(size_t)(out_h - 1) * stride] + cv->host[(size_t)(out_h - 1) * stride + 1];
    }
    const unsigned char* vsrc = img;
    int shift = 0;
    if (need_h) {
        unsigned char* hdst = need_v ? tmp : out;
        MI355_REQUIRE(hdst != nullptr, "resize: tmp (h * out_w * 3 bytes) is required when both sides change");
        const int rows = last - first;
        hipLaunchKernelGGL(k_resize_h, dim3(cdiv(out_w, 64), cdiv(rows, 4)), dim3(256), 0, st, img, w, first, rows, hdst,
                           out_w, ch->dev, ch->ksize);
        MI355_LAUNCH_CHECK();
        vsrc = hdst;
        shift = first;
    }
    if (need_v) {
        hipLaunchKernelGGL(k_resize_v, dim3(cdiv(3 * out_w, 256), out_h), dim3(256), 0, st, vsrc, out_w, shift, out, out_h,
                           cv->dev, cv->ksize);
        MI355_LAUNCH_CHECK();
    }
    return OK;
}

int mi355_score_boost(const float* scores, int64_t n, float eps, float alpha, float threshold, int mode, float* out,
                      void* stream) {
    MI355_REQUIRE(n >= 0, "score_boost: n=%lld", (long long)n);
    MI355_REQUIRE(mode >= 0 && mode <= 2, "score_boost: mode %d (0 = threshold, 1 = for_pos, 2 = for_neg)", mode);
    if (n == 0) return OK;
    MI355_REQUIRE(scores && out, "score_boost: null pointer");
    hipLaunchKernelGGL(k_score_boost, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, scores, out, (long)n,
                       eps, alpha, threshold, mode);
    MI355_LAUNCH_CHECK();
    return OK;
}

}  // extern "C"
