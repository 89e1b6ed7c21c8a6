// Evaluation tail on the device: the step after the hot path.
//   mask logits -> sigmoid -> bilinear resize to the ground-truth size -> uint8       (test_ovcos_maskdecoder_edge.py:103,116-130)
//   uint8 mask x ground truth -> per-quadrant joint histograms                         (recorder/ovcos_metricer.py:158-180 and
//                                                                                       pysodmetrics 1.4.2, which only ever looks
//                                                                                       at the 256 levels of the uint8 mask)
//   class scores -> top-1 / top-5 counters                                             (recorder/new_evaluator.py:47-59)
// Byte/integer work, HBM-bound: one read of the 4 B/px logits, one write + one read of the 1 B/px mask and ground truth;
// 8 KB of counters leave the device instead of a 4 MB float mask per image.
#include "common.h"
#include "../../include/cvlm.h"

namespace {

// cv::resize(INTER_LINEAR) on float32 restated: per axis f = (d + 0.5) * (src / dst) - 0.5 in double -> float,
// i = floor(f), f -= i; columns: i < 0 -> (0, weight 0), i >= W-1 -> (W-1, weight 0); rows: indices clamped,
// weight kept.  Horizontal blend first, then vertical, each as two fp32 products and one fp32 sum (no fma).
__global__ __launch_bounds__(256) void mask_to_u8_kernel(const float* __restrict__ logits, int Hs, int Ws, int h, int w,
                                                         uint8_t* __restrict__ dst, int64_t total) {
    const double sx = (double)Ws / (double)w, sy = (double)Hs / (double)h;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % w);
        const int y = (int)((i / w) % h);
        const int64_t n = i / ((int64_t)w * h);
        float fx = (float)(((double)x + 0.5) * sx - 0.5);
        int x0 = (int)floorf(fx);
        fx -= (float)x0;
        if (x0 < 0) { x0 = 0; fx = 0.f; }
        if (x0 >= Ws - 1) { x0 = Ws - 1; fx = 0.f; }
        const int x1 = x0 + 1 < Ws ? x0 + 1 : Ws - 1;
        float fy = (float)(((double)y + 0.5) * sy - 0.5);
        const int yf = (int)floorf(fy);
        fy -= (float)yf;
        const int y0 = yf < 0 ? 0 : (yf > Hs - 1 ? Hs - 1 : yf);
        const int y1 = yf + 1 < 0 ? 0 : (yf + 1 > Hs - 1 ? Hs - 1 : yf + 1);
        const float* p = logits + n * (int64_t)Hs * Ws;
        auto sig = [](float v) { return 1.0f / (1.0f + expf(-v)); };
        const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
        const float t0 = __fadd_rn(__fmul_rn(sig(p[(int64_t)y0 * Ws + x0]), a0), __fmul_rn(sig(p[(int64_t)y0 * Ws + x1]), a1));
        const float t1 = __fadd_rn(__fmul_rn(sig(p[(int64_t)y1 * Ws + x0]), a0), __fmul_rn(sig(p[(int64_t)y1 * Ws + x1]), a1));
        const float v = __fadd_rn(__fmul_rn(t0, b0), __fmul_rn(t1, b1));
        dst[i] = (uint8_t)(int)__fmul_rn(v, 255.0f);     // (pred * 255).astype(np.uint8): truncation
    }
}

// gt > 128 -> (count, sum x, sum y) per image; stats is u64 [N][3], zeroed by the launcher
__global__ __launch_bounds__(256) void gt_centroid_kernel(const uint8_t* __restrict__ gt, int h, int w,
                                                          unsigned long long* __restrict__ stats) {
    const int n = blockIdx.y;
    const uint8_t* g = gt + (int64_t)n * h * w;
    const int64_t total = (int64_t)h * w;
    unsigned long long cnt = 0, sx = 0, sy = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (g[i] > 128) { cnt += 1; sx += (unsigned long long)(i % w); sy += (unsigned long long)(i / w); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        cnt += __shfl_xor(cnt, o, 64); sx += __shfl_xor(sx, o, 64); sy += __shfl_xor(sy, o, 64);
    }
    if ((threadIdx.x & 63) == 0 && cnt) {
        atomicAdd(&stats[3 * n + 0], cnt); atomicAdd(&stats[3 * n + 1], sx); atomicAdd(&stats[3 * n + 2], sy);
    }
}

// one wave adds its 64 (bin, valid) pairs to an LDS histogram with one atomic per distinct bin
__device__ __forceinline__ void wave_hist_add(unsigned* hist, int bin, bool valid) {
    unsigned long long todo = __ballot(valid);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int b = __shfl(bin, leader, 64);
        const unsigned long long same = __ballot(valid && bin == b) & todo;
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[b], (unsigned)__popcll(same));
        todo &= ~same;
    }
}

// hist u32 [N][4 quadrants][2 gt][256 levels]; quadrant split at the S-measure centroid (x, y) = round(mean) + 1,
// (round(w/2), round(h/2)) + 1 for an empty ground truth: LT = [0:y, 0:x], RT = [0:y, x:], LB = [y:, 0:x], RB.
__global__ __launch_bounds__(256) void joint_hist_kernel(const uint8_t* __restrict__ pre, const uint8_t* __restrict__ gt,
                                                         int h, int w, const unsigned long long* __restrict__ stats,
                                                         unsigned* __restrict__ hist) {
    __shared__ unsigned lh[4 * 2 * 256];
    const int n = blockIdx.y;
    for (int i = threadIdx.x; i < 2048; i += 256) lh[i] = 0;
    const unsigned long long cnt = stats[3 * n];
    int cx, cy;
    if (cnt == 0) {
        cx = (int)rint((double)w / 2.0) + 1; cy = (int)rint((double)h / 2.0) + 1;
    } else {
        cx = (int)rint((double)stats[3 * n + 1] / (double)cnt) + 1; cy = (int)rint((double)stats[3 * n + 2] / (double)cnt) + 1;
    }
    __syncthreads();
    const uint8_t* p = pre + (int64_t)n * h * w;
    const uint8_t* g = gt + (int64_t)n * h * w;
    const int64_t total = (int64_t)h * w;
    const int64_t span = (int64_t)gridDim.x * blockDim.x;
    const int64_t rounds = (total + span - 1) / span;          // every lane runs every round: ballots stay wave-uniform
    for (int64_t r = 0; r < rounds; ++r) {
        const int64_t i = r * span + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        const bool valid = i < total;
        int bin = 0;
        if (valid) {
            const int x = (int)(i % w), y = (int)(i / w);
            const int q = (y >= cy ? 2 : 0) + (x >= cx ? 1 : 0);
            bin = (q * 2 + (g[i] > 128 ? 1 : 0)) * 256 + (int)p[i];
        }
        wave_hist_add(lh, bin, valid);
    }
    __syncthreads();
    unsigned* out = hist + (int64_t)n * 2048;
    for (int i = threadIdx.x; i < 2048; i += 256)
        if (lh[i]) atomicAdd(&out[i], lh[i]);
}

// scores f32 [B][C], labels i32 [B] -> pred i32 [B] (first maximum), counters u32 {top-1 hits, top-5 hits, rows}
__global__ __launch_bounds__(64) void topk_kernel(const float* __restrict__ scores, const int32_t* __restrict__ labels, int B,
                                                  int C, int32_t* __restrict__ pred, unsigned* __restrict__ counters) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* s = scores + (int64_t)b * C;
    const int lab = labels[b];
    int best = 0;
    for (int c = 1; c < C; ++c) if (s[c] > s[best]) best = c;
    int ahead = 0;                                              // entries ranked before the label by a stable descending sort
    if (lab >= 0 && lab < C) {
        for (int c = 0; c < C; ++c) ahead += (s[c] > s[lab] || (s[c] == s[lab] && c < lab)) ? 1 : 0;
    } else {
        ahead = C;
    }
    if (pred) pred[b] = best;
    if (best == lab) atomicAdd(&counters[0], 1u);
    if (ahead < 5) atomicAdd(&counters[1], 1u);
    atomicAdd(&counters[2], 1u);
}

inline int grid_for(int64_t n, int cap) {
    int64_t g = (n + 255) / 256;
    return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" {

int cvlm_mask_to_u8(const float* logits, int32_t N, int32_t Hs, int32_t Ws, int32_t h, int32_t w, uint8_t* dst, void* stream) {
    if (!logits || !dst || N <= 0 || Hs <= 0 || Ws <= 0 || h <= 0 || w <= 0) return CVLM_E_BADARG;
    const int64_t total = (int64_t)N * h * w;
    hipLaunchKernelGGL(mask_to_u8_kernel, dim3(grid_for(total, 16384)), dim3(256), 0, (hipStream_t)stream, logits, Hs, Ws, h, w,
                       dst, total);
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_mask_joint_hist(const uint8_t* pre, const uint8_t* gt, int32_t N, int32_t h, int32_t w, uint64_t* stats, uint32_t* hist,
                         void* stream) {
    if (!pre || !gt || !stats || !hist || N <= 0 || h <= 0 || w <= 0) return CVLM_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(stats, 0, (size_t)N * 3 * sizeof(uint64_t), st);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(hist, 0, (size_t)N * 2048 * sizeof(uint32_t), st);
    if (e != hipSuccess) return (int)e;
    const int gx = grid_for((int64_t)h * w, 256);
    hipLaunchKernelGGL(gt_centroid_kernel, dim3(gx, N), dim3(256), 0, st, gt, h, w, (unsigned long long*)stats);
    CVLM_CHECK_LAUNCH();
    hipLaunchKernelGGL(joint_hist_kernel, dim3(gx, N), dim3(256), 0, st, pre, gt, h, w, (const unsigned long long*)stats, hist);
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_topk_accumulate(const float* scores, const int32_t* labels, int32_t B, int32_t C, int32_t* pred, uint32_t* counters,
                         void* stream) {
    if (!scores || !labels || !counters || B <= 0 || C <= 0) return CVLM_E_BADARG;
    hipLaunchKernelGGL(topk_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, scores, labels, B, C, pred, counters);
    CVLM_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
