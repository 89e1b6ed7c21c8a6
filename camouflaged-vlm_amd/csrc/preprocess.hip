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

// ---- fast paths (same arithmetic, wider memory accesses) ------------------------------------------------------------
// horizontal pass: one workgroup per source row; the row is staged in LDS with 16-byte loads, every thread forms whole
// output pixels (all C channels) from LDS bytes, and the output row goes back through LDS as 16-byte stores
template <int C>
__global__ __launch_bounds__(256) void resample_row_kernel(const uint8_t* __restrict__ src, int H, int W,
                                                           const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk,
                                                           int ksize, int n_out, uint8_t* __restrict__ dst) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];     // source row, then output row (each 16-byte padded)
    const int64_t r = blockIdx.x;                                    // n * H + y
    const int in_b = W * C, out_b = n_out * C;
    const uint8_t* s = src + r * in_b;
    uint8_t* d = dst + r * out_b;
    // both rows sit in LDS at the same offset modulo 16 as in memory, so every 16-byte global access pairs with a
    // 16-byte LDS access
    const int sh_in = (int)((uintptr_t)s & 15), sh_out = (int)((uintptr_t)d & 15);
    uint8_t* row = lds + sh_in;
    uint8_t* orow = lds + ((sh_in + in_b + 15) & ~15) + 16 + sh_out;
    const int head = (16 - sh_in) & 15;                              // bytes up to the first 16-byte boundary
    for (int i = threadIdx.x; i < head && i < in_b; i += 256) row[i] = s[i];
    const int nvec = in_b > head ? (in_b - head) >> 4 : 0;
    for (int i = threadIdx.x; i < nvec; i += 256) *(uint4*)(row + head + 16 * i) = *(const uint4*)(s + head + 16 * i);
    for (int i = head + 16 * nvec + threadIdx.x; i < in_b; i += 256) row[i] = s[i];
    __syncthreads();
    for (int o = threadIdx.x; o < n_out; o += 256) {
        const int lo = bounds[2 * o], cnt = bounds[2 * o + 1];
        const int32_t* k = kk + (int64_t)o * ksize;
        int acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 1 << (PRECISION_BITS - 1);
        for (int t = 0; t < cnt; ++t) {
            const int kt = k[t];
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] += (int)row[(lo + t) * C + c] * kt;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            int v = acc[c] >> PRECISION_BITS;
            orow[o * C + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
    __syncthreads();
    const int ohead = (16 - sh_out) & 15;
    for (int i = threadIdx.x; i < ohead && i < out_b; i += 256) d[i] = orow[i];
    const int ovec = out_b > ohead ? (out_b - ohead) >> 4 : 0;
    for (int i = threadIdx.x; i < ovec; i += 256) *(uint4*)(d + ohead + 16 * i) = *(const uint4*)(orow + ohead + 16 * i);
    for (int i = ohead + 16 * ovec + threadIdx.x; i < out_b; i += 256) d[i] = orow[i];
}

// vertical pass, rows of a multiple of 4 bytes: four consecutive bytes per thread, one 4-byte load per tap
__global__ __launch_bounds__(256) void resample_col4_kernel(const uint8_t* __restrict__ src, int H, int rowb,
                                                            const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk,
                                                            int ksize, int n_out, uint8_t* __restrict__ dst, int64_t total4) {
    const int rw = rowb >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        const int xw = (int)(i % rw);
        const int y = (int)((i / rw) % n_out);
        const int64_t n = i / ((int64_t)rw * n_out);
        const int lo = bounds[2 * y], cnt = bounds[2 * y + 1];
        const int32_t* k = kk + (int64_t)y * ksize;
        const uint8_t* p = src + (n * H + lo) * (int64_t)rowb + 4 * xw;
        int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0, a3 = a0;
        for (int t = 0; t < cnt; ++t) {
            const uint32_t v = *(const uint32_t*)(p + (int64_t)t * rowb);
            const int kt = k[t];
            a0 += (int)(v & 255) * kt; a1 += (int)((v >> 8) & 255) * kt; a2 += (int)((v >> 16) & 255) * kt; a3 += (int)(v >> 24) * kt;
        }
        auto clamp8 = [](int v) { v >>= PRECISION_BITS; return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
        *(uint32_t*)(dst + (n * n_out + y) * (int64_t)rowb + 4 * xw) = clamp8(a0) | (clamp8(a1) << 8) | (clamp8(a2) << 16) | (clamp8(a3) << 24);
    }
}

// ToTensor + Normalize for C = 3, four pixels per thread: twelve source bytes -> one float4 per channel plane
__global__ __launch_bounds__(256) void u8_to_tensor3x4_kernel(const uint8_t* __restrict__ src, int H, int W, int top, int left,
                                                              int ch, int cw, const float* __restrict__ mean,
                                                              const float* __restrict__ stdv, float* __restrict__ dst,
                                                              int64_t total4) {
    const int cw4 = cw >> 2;
    const float m0 = mean[0], m1 = mean[1], m2 = mean[2], s0 = stdv[0], s1 = stdv[1], s2 = stdv[2];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        const int x4 = (int)(i % cw4);
        const int y = (int)((i / cw4) % ch);
        const int64_t n = i / ((int64_t)cw4 * ch);
        const uint8_t* p = src + ((n * H + top + y) * (int64_t)W + left + 4 * x4) * 3;
        uint8_t b[12];
        if ((((uintptr_t)p) & 3) == 0) {
            *(uint32_t*)(b) = *(const uint32_t*)p; *(uint32_t*)(b + 4) = *(const uint32_t*)(p + 4); *(uint32_t*)(b + 8) = *(const uint32_t*)(p + 8);
        } else {
#pragma unroll
            for (int j = 0; j < 12; ++j) b[j] = p[j];
        }
        float4 o0, o1, o2;
        o0.x = ((float)b[0] / 255.0f - m0) / s0; o1.x = ((float)b[1] / 255.0f - m1) / s1; o2.x = ((float)b[2] / 255.0f - m2) / s2;
        o0.y = ((float)b[3] / 255.0f - m0) / s0; o1.y = ((float)b[4] / 255.0f - m1) / s1; o2.y = ((float)b[5] / 255.0f - m2) / s2;
        o0.z = ((float)b[6] / 255.0f - m0) / s0; o1.z = ((float)b[7] / 255.0f - m1) / s1; o2.z = ((float)b[8] / 255.0f - m2) / s2;
        o0.w = ((float)b[9] / 255.0f - m0) / s0; o1.w = ((float)b[10] / 255.0f - m1) / s1; o2.w = ((float)b[11] / 255.0f - m2) / s2;
        const int64_t plane = (int64_t)ch * cw;
        float* q = dst + n * 3 * plane + (int64_t)y * cw + 4 * x4;
        *(float4*)q = o0; *(float4*)(q + plane) = o1; *(float4*)(q + 2 * plane) = o2;
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
    hipStream_t st = (hipStream_t)stream;
    if (axis == 1 && (C == 3 || C == 1) && (int64_t)N * H < (1 << 30)) {     // horizontal: one workgroup per row through LDS
        const int smem = (((W * C) + 31) & ~15) + 16 + (((n_out * C) + 31) & ~15) + 16;
        if (smem <= 60 * 1024) {
            if (C == 3) hipLaunchKernelGGL(resample_row_kernel<3>, dim3((unsigned)(N * H)), dim3(256), smem, st, src, H, W, bounds, kk, ksize, n_out, dst);
            else hipLaunchKernelGGL(resample_row_kernel<1>, dim3((unsigned)(N * H)), dim3(256), smem, st, src, H, W, bounds, kk, ksize, n_out, dst);
            CVLM_CHECK_LAUNCH();
            return 0;
        }
    }
    if (axis == 0 && ((W * C) & 3) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 3) == 0) {   // vertical: 4 bytes per thread
        const int64_t total4 = total >> 2;
        hipLaunchKernelGGL(resample_col4_kernel, dim3(grid_for(total4)), dim3(256), 0, st, src, H, W * C, bounds, kk, ksize, n_out, dst, total4);
        CVLM_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(resample_u8_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, H, W, C, bounds,
                       kk, ksize, n_out, axis, dst, total);
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_u8_to_tensor(const uint8_t* src, int32_t N, int32_t H, int32_t W, int32_t C, int32_t top, int32_t left,
                      int32_t ch, int32_t cw, const float* mean, const float* stdv, float* dst, void* stream) {
    if (!src || !dst || !mean || !stdv || top < 0 || left < 0 || top + ch > H || left + cw > W) return CVLM_E_BADARG;
    const int64_t total = (int64_t)N * C * ch * cw;
    if (C == 3 && (cw & 3) == 0 && (((uintptr_t)dst) & 15) == 0) {
        const int64_t total4 = (int64_t)N * ch * (cw >> 2);
        hipLaunchKernelGGL(u8_to_tensor3x4_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, src, H, W, top, left,
                           ch, cw, mean, stdv, dst, total4);
        CVLM_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(u8_to_tensor_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, H, W, C, top,
                       left, ch, cw, mean, stdv, dst, total);
    CVLM_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
