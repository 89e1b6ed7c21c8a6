// GPU image preprocessing: the step before the hot path.
// Byte/integer work, HBM-bound: Pillow's 8-bit separable resample (libImaging/Resample.c) restated bit for bit
// -- per output index a window [xmin, xmin+n) of 22-bit fixed-point coefficients (prepared on the host),
// accumulate in int32 from 1 << 21, shift, clamp to uint8 -- and ToTensor + Normalize (+ crop) in fp32.
#include "common.h"
#include "../../include/cvlm.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

// src uint8 [N][H][W][C] -> dst uint8 with `axis` (0 = rows, 1 = columns) resampled to n_out entries
__global__ __launch_bounds__(256) void resample_u8_kernel(const uint8_t* __restrict__ src, int H, int W, int C,
                                                          const int32_t* __restrict__ bounds,
                                                          const int32_t* __restrict__ kk, int ksize, int n_out,
                                                          int axis, uint8_t* __restrict__ dst, int64_t total) {
    const int oh = axis == 0 ? n_out : H, ow = axis == 1 ? n_out : W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int x = (int)((i / C) % ow);
        const int y = (int)((i / ((int64_t)C * ow)) % oh);
        const int64_t n = i / ((int64_t)C * ow * oh);
        const int o = axis == 0 ? y : x;
        const int lo = bounds[2 * o], cnt = bounds[2 * o + 1];
        const int32_t* k = kk + (int64_t)o * ksize;
        const uint8_t* p = src + n * (int64_t)H * W * C;
        int acc = 1 << (PRECISION_BITS - 1);
        if (axis == 0) {
            for (int t = 0; t < cnt; ++t) acc += (int)p[((int64_t)(lo + t) * W + x) * C + c] * k[t];
        } else {
            for (int t = 0; t < cnt; ++t) acc += (int)p[((int64_t)y * W + lo + t) * C + c] * k[t];
        }
        int v = acc >> PRECISION_BITS;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        dst[i] = (uint8_t)v;
    }
}

// uint8 [N][H][W][C] (crop window top/left/ch/cw) -> f32 [N][C][ch][cw]: (x / 255 - mean[c]) / std[c]
__global__ __launch_bounds__(256) void u8_to_tensor_kernel(const uint8_t* __restrict__ src, int H, int W, int C, int top,
                                                           int left, int ch, int cw, const float* __restrict__ mean,
                                                           const float* __restrict__ stdv, float* __restrict__ dst,
                                                           int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % cw);
        const int y = (int)((i / cw) % ch);
        const int c = (int)((i / ((int64_t)cw * ch)) % C);
        const int64_t n = i / ((int64_t)cw * ch * C);
        const float t = (float)src[((n * H + top + y) * (int64_t)W + left + x) * C + c] / 255.0f;
        dst[i] = (t - mean[c]) / stdv[c];
    }
}

inline int grid_for(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" {

int cvlm_resample_u8(const uint8_t* src, int32_t N, int32_t H, int32_t W, int32_t C, const int32_t* bounds,
                     const int32_t* kk, int32_t ksize, int32_t n_out, int32_t axis, uint8_t* dst, void* stream) {
    if (!src || !dst || !bounds || !kk || N <= 0 || H <= 0 || W <= 0 || C <= 0 || ksize <= 0 || n_out <= 0 ||
        (axis != 0 && axis != 1))
        return CVLM_E_BADARG;
    const int64_t total = (int64_t)N * (axis == 0 ? n_out : H) * (axis == 1 ? n_out : W) * C;
    hipLaunchKernelGGL(resample_u8_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, H, W, C, bounds,
                       kk, ksize, n_out, axis, dst, total);
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_u8_to_tensor(const uint8_t* src, int32_t N, int32_t H, int32_t W, int32_t C, int32_t top, int32_t left,
                      int32_t ch, int32_t cw, const float* mean, const float* stdv, float* dst, void* stream) {
    if (!src || !dst || !mean || !stdv || top < 0 || left < 0 || top + ch > H || left + cw > W) return CVLM_E_BADARG;
    const int64_t total = (int64_t)N * C * ch * cw;
    hipLaunchKernelGGL(u8_to_tensor_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, H, W, C, top,
                       left, ch, cw, mean, stdv, dst, total);
    CVLM_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
