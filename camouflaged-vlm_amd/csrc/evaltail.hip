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

// gt > 128 -> per-workgroup (count, sum x, sum y) partials, image n's at partial[n * 1024 + 3 * block .. ].
// 16 pixels per thread and iteration (one 16-byte load); x / y of a 16-pixel run are derived once per run.
__global__ __launch_bounds__(256) void gt_centroid_kernel(const uint8_t* __restrict__ gt, int h, int w,
                                                          unsigned long long* __restrict__ partial) {
    const int n = blockIdx.y;
    const uint8_t* g = gt + (int64_t)n * h * w;
    const int64_t total = (int64_t)h * w;
    const bool vec = ((total & 15) == 0) && ((((uintptr_t)g) & 15) == 0);
    unsigned long long cnt = 0, sx = 0, sy = 0;
    const int64_t nrun = (total + 15) >> 4;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrun; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i0 = r << 4;
        uint8_t px[16];
        if (vec) {
            *(uint4*)px = *(const uint4*)(g + i0);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) px[j] = i0 + j < total ? g[i0 + j] : 0;
        }
        int y = (int)(i0 / w), x = (int)(i0 - (int64_t)y * w);
        unsigned c = 0, ax = 0, ay = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (px[j] > 128) { c += 1; ax += (unsigned)x; ay += (unsigned)y; }
            if (++x == w) { x = 0; ++y; }
        }
        cnt += c; sx += ax; sy += ay;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        cnt += __shfl_xor(cnt, o, 64); sx += __shfl_xor(sx, o, 64); sy += __shfl_xor(sy, o, 64);
    }
    // one partial per workgroup, no atomics: 768 same-address device-scope atomics per image took 150 us
    __shared__ unsigned long long part[4][3];
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6][0] = cnt; part[threadIdx.x >> 6][1] = sx; part[threadIdx.x >> 6][2] = sy; }
    __syncthreads();
    if (threadIdx.x < 3)
        partial[(int64_t)n * 1024 + blockIdx.x * 3 + threadIdx.x] =          // image n's triples live in image n's 8 KB
            part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}

// sums the per-workgroup partials of one image into stats[n] and clears that image's histogram (whose memory the
// partials were parked in)
__global__ __launch_bounds__(256) void centroid_finalize_kernel(unsigned long long* __restrict__ partial, int nparts,
                                                                unsigned long long* __restrict__ stats,
                                                                unsigned* __restrict__ hist) {
    const int n = blockIdx.x;
    __shared__ unsigned long long red[256][3];
    unsigned long long a[3] = {0, 0, 0};
    for (int i = threadIdx.x; i < nparts; i += 256)
#pragma unroll
        for (int k = 0; k < 3; ++k) a[k] += partial[(int64_t)n * 1024 + i * 3 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) red[threadIdx.x][k] = a[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o)
#pragma unroll
            for (int k = 0; k < 3; ++k) red[threadIdx.x][k] += red[threadIdx.x + o][k];
        __syncthreads();
    }
    if (threadIdx.x < 3) stats[3 * n + threadIdx.x] = red[0][threadIdx.x];
    __syncthreads();                                                  // every partial has been read
    for (int i = threadIdx.x; i < 2048; i += 256) hist[(int64_t)n * 2048 + i] = 0;
}

// hist u32 [N][4 quadrants][2 gt][256 levels]; quadrant split at the S-measure centroid (x, y) = round(mean) + 1,
// (round(w/2), round(h/2)) + 1 for an empty ground truth: LT = [0:y, 0:x], RT = [0:y, x:], LB = [y:, 0:x], RB.
// One LDS histogram per wave (4 x 8 KB), 16 pixels per lane and iteration.  Camouflage masks are mostly flat: when
// all 64 lanes of a wave hold the same bin for their j-th pixel, one lane adds 64; otherwise plain LDS atomics.
__global__ __launch_bounds__(256) void joint_hist_kernel(const uint8_t* __restrict__ pre, const uint8_t* __restrict__ gt,
                                                         int h, int w, const unsigned long long* __restrict__ stats,
                                                         unsigned* __restrict__ hist) {
    __shared__ unsigned lh[4][2048];
    const int n = blockIdx.y, lane = threadIdx.x & 63;
    unsigned* my = lh[threadIdx.x >> 6];
    for (int i = threadIdx.x; i < 4 * 2048; i += 256) (&lh[0][0])[i] = 0;
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
    const bool vec = ((total & 15) == 0) && ((((uintptr_t)p) & 15) == 0) && ((((uintptr_t)g) & 15) == 0);
    const int64_t nrun = (total + 15) >> 4;
    const int64_t span = (int64_t)gridDim.x * blockDim.x;
    const int64_t rounds = (nrun + span - 1) / span;                 // every lane runs every round: ballots stay wave-uniform
    for (int64_t rd = 0; rd < rounds; ++rd) {
        const int64_t r = rd * span + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        const int64_t i0 = r << 4;
        uint8_t pp[16], gg[16];
        if (vec && r < nrun) {
            *(uint4*)pp = *(const uint4*)(p + i0);
            *(uint4*)gg = *(const uint4*)(g + i0);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const bool ok = r < nrun && i0 + j < total;
                pp[j] = ok ? p[i0 + j] : 0; gg[j] = ok ? g[i0 + j] : 0;
            }
        }
        int y = (int)(i0 / w), x = (int)(i0 - (int64_t)y * w);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const bool valid = r < nrun && i0 + j < total;
            const int q = (y >= cy ? 2 : 0) + (x >= cx ? 1 : 0);
            const int bin = (q * 2 + (gg[j] > 128 ? 1 : 0)) * 256 + (int)pp[j];
            const int first = __shfl(bin, 0, 64);
            if (__all(valid && bin == first)) {
                if (lane == 0) atomicAdd(&my[first], 64u);
            } else if (valid) {
                atomicAdd(&my[bin], 1u);
            }
            if (++x == w) { x = 0; ++y; }
        }
    }
    __syncthreads();
    unsigned* out = hist + (int64_t)n * 2048;
    for (int i = threadIdx.x; i < 2048; i += 256) {
        const unsigned v = lh[0][i] + lh[1][i] + lh[2][i] + lh[3][i];
        if (v) atomicAdd(&out[i], v);
    }
}

// ---- weighted F-measure (pysodmetrics 1.4.2 WeightedFmeasure.cal_wfm as called by ovcos_metricer.py:49-66) --------
// Exact Euclidean distance transform with the index of the nearest foreground pixel, resolved like scipy's
// distance_transform_edt (checked on the CPU against scipy on random masks): nearest foreground row inside each column,
// ties to the smaller row; then along the row the column minimising dx^2 + dy^2, ties to the smaller column.
// Workspace per image: near_y i32 [h][w] | d2 i32 [h][w] | Et f64 [h][w].
constexpr int WFM_INF = 1 << 29;

__device__ __forceinline__ double wfm_norm(int v, int lo, int hi) {      // prepare_data per level, same IEEE operations
    const double p = (double)v / 255.0, pl = (double)lo / 255.0, ph = (double)hi / 255.0;
    return hi != lo ? (p - pl) / (ph - pl) : p;
}

// lowest / highest level present in an image's joint histogram -> lohi i32 [N][2]
__global__ __launch_bounds__(256) void wfm_levels_kernel(const unsigned* __restrict__ hist, int* __restrict__ lohi) {
    const int n = blockIdx.x, v = threadIdx.x;
    unsigned c = 0;
    for (int k = 0; k < 8; ++k) c += hist[(int64_t)n * 2048 + k * 256 + v];
    __shared__ int lo, hi;
    if (v == 0) { lo = 255; hi = 0; }
    __syncthreads();
    if (c) { atomicMin(&lo, v); atomicMax(&hi, v); }
    __syncthreads();
    if (v == 0) { lohi[2 * n] = lo; lohi[2 * n + 1] = hi; }
}

// Where the weighted F-measure takes the prepared prediction of a pixel from (`prepare_data`, sod_metric.py:12-26: v / 255, min-max
// normalised), as a double:
//   WfmU8:   a uint8 mask and the lowest / highest level present (OVCOSMetricer.step: the 256 levels, normalised per level);
//   WfmProb: a FLOAT probability map and its min / max (utils.calc_cod feeds `y_pred * 255` as float32: no uint8 step) -- the same
//            float32 operations numpy performs: p * 255, / 255, (v - min) / (max - min).
__device__ __forceinline__ float cod_pred(float p) { return __fdiv_rn(__fmul_rn(p, 255.0f), 255.0f); }
__device__ __forceinline__ float cod_norm(float p, float mn, float mx) {
    const float v = cod_pred(p);
    return mx != mn ? __fdiv_rn(__fsub_rn(v, mn), __fsub_rn(mx, mn)) : v;
}
struct WfmU8 {
    const uint8_t* pre; const int* lohi;
    struct Img {
        const uint8_t* pre; int lo, hi;
        __device__ __forceinline__ double operator()(int64_t i) const { return wfm_norm(pre[i], lo, hi); }
    };
    __device__ __forceinline__ Img image(int n) const { return Img{pre, lohi[2 * n], lohi[2 * n + 1]}; }
};
struct WfmProb {
    const float* prob; const float* minmax;
    struct Img {
        const float* prob; float mn, mx;
        __device__ __forceinline__ double operator()(int64_t i) const { return (double)cod_norm(prob[i], mn, mx); }
    };
    __device__ __forceinline__ Img image(int n) const { return Img{prob, minmax[2 * n], minmax[2 * n + 1]}; }
};

// ---- utils.calc_cod on float probability maps (utils.py:143-165) -----------------------------------------------------------------
constexpr int COD_PARTS = 64;
// per-workgroup (min, max) of fl(fl(p * 255) / 255) over a share of image n -> partial f32 [N][COD_PARTS][2]
__global__ __launch_bounds__(256) void cod_minmax_kernel(const float* __restrict__ prob, int64_t hw, float* __restrict__ partial) {
    const int n = blockIdx.y;
    const float* p = prob + (int64_t)n * hw;
    float mn = INFINITY, mx = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)COD_PARTS * 256) {
        const float v = cod_pred(p[i]);
        mn = fminf(mn, v); mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
    __shared__ float red[4][2];
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = mn; red[threadIdx.x >> 6][1] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[((int64_t)n * COD_PARTS + blockIdx.x) * 2] = fminf(fminf(red[0][0], red[1][0]), fminf(red[2][0], red[3][0]));
        partial[((int64_t)n * COD_PARTS + blockIdx.x) * 2 + 1] = fmaxf(fmaxf(red[0][1], red[1][1]), fmaxf(red[2][1], red[3][1]));
    }
}

// every workgroup folds the COD_PARTS partials of its image (block 0 publishes minmax[n]), then q = uint8(pn * 255): the levels the
// E-measure's cumulative histograms count (sod_metric.py:420, `(pred * 255).astype(np.uint8)`)
__global__ __launch_bounds__(256) void cod_quant_kernel(const float* __restrict__ prob, int64_t hw, const float* __restrict__ partial,
                                                        float* __restrict__ minmax, uint8_t* __restrict__ q) {
    const int n = blockIdx.y;
    __shared__ float mm[2];
    if (threadIdx.x < 64) {
        float mn = partial[((int64_t)n * COD_PARTS + threadIdx.x) * 2], mx = partial[((int64_t)n * COD_PARTS + threadIdx.x) * 2 + 1];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
        if (threadIdx.x == 0) { mm[0] = mn; mm[1] = mx; if (blockIdx.x == 0) { minmax[2 * n] = mn; minmax[2 * n + 1] = mx; } }
    }
    __syncthreads();
    const float mn = mm[0], mx = mm[1];
    const float* p = prob + (int64_t)n * hw;
    uint8_t* o = q + (int64_t)n * hw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256)
        o[i] = (uint8_t)(int)__fmul_rn(cod_norm(p[i], mn, mx), 255.0f);
}

// sums of pn and pn^2 per S-measure quadrant (split at the ground truth's centroid, as joint_hist_kernel) and ground-truth class:
// partial f64 [N][COD_PARTS][4][2][2]
__global__ __launch_bounds__(256) void cod_moments_kernel(const float* __restrict__ prob, const uint8_t* __restrict__ gt, int h, int w,
                                                          const float* __restrict__ minmax, const unsigned long long* __restrict__ stats,
                                                          double* __restrict__ partial) {
    const int n = blockIdx.y;
    const unsigned long long cnt = stats[3 * n];
    int cx, cy;
    if (cnt == 0) {
        cx = (int)rint((double)w / 2.0) + 1; cy = (int)rint((double)h / 2.0) + 1;
    } else {
        cx = (int)rint((double)stats[3 * n + 1] / (double)cnt) + 1; cy = (int)rint((double)stats[3 * n + 2] / (double)cnt) + 1;
    }
    const float mn = minmax[2 * n], mx = minmax[2 * n + 1];
    const int64_t hw = (int64_t)h * w;
    const float* p = prob + (int64_t)n * hw;
    const uint8_t* g = gt + (int64_t)n * hw;
    double acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)COD_PARTS * 256) {
        const int y = (int)(i / w), x = (int)(i - (int64_t)y * w);
        const int cell = ((y >= cy ? 2 : 0) + (x >= cx ? 1 : 0)) * 2 + (g[i] > 128 ? 1 : 0);
        const double v = (double)cod_norm(p[i], mn, mx);
#pragma unroll
        for (int k = 0; k < 8; ++k) {                                 // branch-free: registers cannot be indexed by `cell`
            const double m = cell == k ? 1.0 : 0.0;
            acc[2 * k] += m * v;
            acc[2 * k + 1] += m * v * v;
        }
    }
    __shared__ double red[4][16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        double a = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = a;
    }
    __syncthreads();
    if (threadIdx.x < 16)
        partial[((int64_t)n * COD_PARTS + blockIdx.x) * 16 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ __launch_bounds__(64) void cod_moments_finalize_kernel(const double* __restrict__ partial, double* __restrict__ out) {
    const int n = blockIdx.x;
    if (threadIdx.x < 16) {
        double a = 0.0;
        for (int b = 0; b < COD_PARTS; ++b) a += partial[((int64_t)n * COD_PARTS + b) * 16 + threadIdx.x];   // fixed order
        out[(int64_t)n * 16 + threadIdx.x] = a;
    }
}

// Column pass: nearest foreground row inside each column (above-or-at vs below, ties to the smaller row).
// A workgroup owns 64 columns; the column is cut into segments of 64 rows and a thread keeps the foreground flags of a
// segment as one 64-bit mask (segments s = ty, ty + 16, ...: h <= 4096).  First / last foreground row of every segment go
// through LDS; a row's answer is then two bit scans of its own mask plus the carries of the neighbouring segments -- the
// column is read once, coalesced, by 16 segments at a time (round 3: one thread walked its whole column twice, 540 us at
// 1365 x 2048: 32 waves on the whole chip waiting for one global load after the other).
constexpr int WFM_SEG = 64, WFM_MAXSEG = 64;
__global__ __launch_bounds__(1024) void wfm_colpass_kernel(const uint8_t* __restrict__ gt, int h, int w, int* __restrict__ near_y) {
    __shared__ int first[WFM_MAXSEG][64], last[WFM_MAXSEG][64];
    const int n = blockIdx.y, tx = threadIdx.x, ty = threadIdx.y;
    const int x = blockIdx.x * 64 + tx;
    const int S = (h + WFM_SEG - 1) / WFM_SEG;
    const uint8_t* g = gt + (int64_t)n * h * w;
    int* ny = near_y + (int64_t)n * h * w;
    unsigned long long mask[WFM_MAXSEG / 16];
#pragma unroll
    for (int k = 0; k < WFM_MAXSEG / 16; ++k) {
        const int sgm = ty + 16 * k;
        unsigned long long m = 0;
        if (sgm < S && x < w) {
            const int y0 = sgm * WFM_SEG, cnt = min(WFM_SEG, h - y0);
            for (int r = 0; r < cnt; ++r) m |= (unsigned long long)(g[(int64_t)(y0 + r) * w + x] > 128) << r;
        }
        mask[k] = m;
        if (sgm < S) {
            first[sgm][tx] = m ? sgm * WFM_SEG + __builtin_ctzll(m) : -1;
            last[sgm][tx] = m ? sgm * WFM_SEG + 63 - __builtin_clzll(m) : -1;
        }
    }
    __syncthreads();
    if (x >= w) return;
#pragma unroll
    for (int k = 0; k < WFM_MAXSEG / 16; ++k) {
        const int sgm = ty + 16 * k;
        if (sgm >= S) continue;
        int ca = -1, cb = -1;                                         // nearest foreground row in the segments above / below
        for (int t = sgm - 1; t >= 0 && ca < 0; --t) ca = last[t][tx];
        for (int t = sgm + 1; t < S && cb < 0; ++t) cb = first[t][tx];
        const unsigned long long m = mask[k];
        const int y0 = sgm * WFM_SEG, cnt = min(WFM_SEG, h - y0);
        for (int r = 0; r < cnt; ++r) {
            const int y = y0 + r;
            const unsigned long long lo = m & (~0ull >> (63 - r)), hi = m >> r;
            const int above = lo ? y0 + 63 - __builtin_clzll(lo) : ca;
            const int below = hi ? y + __builtin_ctzll(hi) : cb;
            int best = above;
            if (below >= 0 && (above < 0 || below - y < y - above)) best = below;
            ny[(int64_t)y * w + x] = best;
        }
    }
}

// the same for columns taller than WFM_SEG * WFM_MAXSEG rows: one thread per column
__global__ __launch_bounds__(256) void wfm_colpass_tall_kernel(const uint8_t* __restrict__ gt, int h, int w, int* __restrict__ near_y) {
    const int n = blockIdx.y, x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= w) return;
    const uint8_t* g = gt + (int64_t)n * h * w;
    int* ny = near_y + (int64_t)n * h * w;
    int last = -1;
    for (int y = 0; y < h; ++y) {                                     // nearest at or above
        if (g[(int64_t)y * w + x] > 128) last = y;
        ny[(int64_t)y * w + x] = last;
    }
    int below = -1;
    for (int y = h - 1; y >= 0; --y) {                                // combine with nearest at or below
        if (g[(int64_t)y * w + x] > 128) below = y;
        const int above = ny[(int64_t)y * w + x];
        int best = above;
        if (below >= 0 && (above < 0 || below - y < y - above)) best = below;
        ny[(int64_t)y * w + x] = best;
    }
}

// Row pass, one workgroup per row: Et[y][x] = E at the nearest foreground pixel (E itself on the foreground), d2 = squared
// distance.  Only columns that hold foreground at all can be nearest (near_y >= 0, the same set for every row): the row's
// candidates are compacted into LDS (column, dy^2), `rank[x]` = candidates left of x.  A background pixel scans the
// candidates outwards from its own position, left side first, and stops a side once dx^2 alone exceeds the best distance --
// the same minimum as the full scan in column order, ties to the smaller column: on the left a later (smaller) column
// replaces an equal distance, on the right only a strictly smaller one does (round 3 scanned all w columns for every pixel).
template <typename SRC>
__global__ __launch_bounds__(256) void wfm_rowpass_kernel(const SRC src, const uint8_t* __restrict__ gt, int h,
                                                          int w, const int* __restrict__ near_y,
                                                          int* __restrict__ d2o, double* __restrict__ Et) {
    extern __shared__ int rowbuf[];                                   // candidate column [w] | its dy^2 [w] | its near row [w] | rank [w]
    __shared__ int wave_tot[4], total_c;
    int* colx = rowbuf;
    int* dy2 = rowbuf + w;
    int* nry = rowbuf + 2 * w;
    int* rank = rowbuf + 3 * w;
    const int n = blockIdx.y, y = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t base = (int64_t)n * h * w;
    const auto val = src.image(n);                                    // val(i): the prepared prediction of pixel i of this image, as a double
    // compaction: thread t owns the columns [t * per, t * per + per)
    const int per = (w + 255) / 256;
    const int c0 = min(tid * per, w), c1 = min(c0 + per, w);
    int mine = 0;
    for (int x = c0; x < c1; ++x) mine += near_y[base + (int64_t)y * w + x] >= 0;
    int incl = mine;                                                  // inclusive scan over the workgroup
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    int off = incl - mine;
    for (int k = 0; k < wv; ++k) off += wave_tot[k];
    if (tid == 255) total_c = off + mine;
    for (int x = c0; x < c1; ++x) {
        const int r = near_y[base + (int64_t)y * w + x];
        rank[x] = off;
        if (r >= 0) { colx[off] = x; dy2[off] = (r - y) * (r - y); nry[off] = r; ++off; }
    }
    __syncthreads();
    const int cnt = total_c;
    for (int x = tid; x < w; x += 256) {
        const int64_t i = base + (int64_t)y * w + x;
        if (gt[i] > 128) {
            d2o[i] = 0;
            Et[i] = fabs(val(i) - 1.0);
            continue;
        }
        int best = WFM_INF + WFM_INF, bj = -1;
        const int p = rank[x];
        for (int j = p - 1; j >= 0; --j) {                            // left of x, nearest column first
            const int dx = x - colx[j], dd = dx * dx;
            if (dd > best) break;
            const int d = dd + dy2[j];
            if (d <= best) { best = d; bj = j; }
        }
        for (int j = p; j < cnt; ++j) {                               // x itself and right of it
            const int dx = colx[j] - x, dd = dx * dx;
            if (dd >= best) break;
            const int d = dd + dy2[j];
            if (d < best) { best = d; bj = j; }
        }
        d2o[i] = best;
        Et[i] = bj < 0 ? 0.0 : fabs(val(base + (int64_t)nry[bj] * w + colx[bj]) - 1.0);
    }
}

// 7x7 Gaussian of Et (zero outside), pixel importance, block partial sums {sum Ew over fg, sum Ew over bg, fg count}
template <typename SRC>
__global__ __launch_bounds__(256) void wfm_weight_kernel(const SRC src, const uint8_t* __restrict__ gt, int h,
                                                         int w, const int* __restrict__ d2,
                                                         const double* __restrict__ Et, const double* __restrict__ gauss,
                                                         double* __restrict__ partial) {
    __shared__ double K[49];
    __shared__ double red[4][3];
    if (threadIdx.x < 49) K[threadIdx.x] = gauss[threadIdx.x];
    __syncthreads();
    const int n = blockIdx.y;
    const int64_t base = (int64_t)n * h * w, total = (int64_t)h * w;
    const auto val = src.image(n);
    double sfg = 0.0, sbg = 0.0, cfg = 0.0;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        const int y = (int)(i / w), x = (int)(i - (int64_t)y * w);
        double ea = 0.0;
        for (int a = 0; a < 7; ++a) {
            const int yy = y + a - 3;
            if (yy < 0 || yy >= h) continue;
            for (int b = 0; b < 7; ++b) {
                const int xx = x + b - 3;
                if (xx < 0 || xx >= w) continue;
                ea += K[a * 7 + b] * Et[base + (int64_t)yy * w + xx];
            }
        }
        const bool fg = gt[base + i] > 128;
        const double e = fabs(val(base + i) - (fg ? 1.0 : 0.0));
        const double m = (fg && ea < e) ? ea : e;
        const double bw = fg ? 1.0 : 2.0 - exp(log(0.5) / 5.0 * sqrt((double)d2[base + i]));
        const double ew = m * bw;
        if (fg) { sfg = ew; cfg = 1.0; } else sbg = ew;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sfg += __shfl_xor(sfg, o, 64); sbg += __shfl_xor(sbg, o, 64); cfg += __shfl_xor(cfg, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = sfg; red[threadIdx.x >> 6][1] = sbg; red[threadIdx.x >> 6][2] = cfg; }
    __syncthreads();
    if (threadIdx.x < 3)
        partial[((int64_t)n * gridDim.x + blockIdx.x) * 3 + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void wfm_finalize_kernel(const double* __restrict__ partial, int nparts, double* __restrict__ out) {
    const int n = blockIdx.x;
    __shared__ double red[256][3];
    double a[3] = {0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < nparts; i += 256)
#pragma unroll
        for (int k = 0; k < 3; ++k) a[k] += partial[((int64_t)n * nparts + i) * 3 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) red[threadIdx.x][k] = a[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o)
#pragma unroll
            for (int k = 0; k < 3; ++k) red[threadIdx.x][k] += red[threadIdx.x + o][k];
        __syncthreads();
    }
    if (threadIdx.x < 3) out[3 * n + threadIdx.x] = red[0][threadIdx.x];
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

// shared tail of the two weighted-F entries: column pass, row pass, weights, partial sums -> out3
template <typename SRC>
static int wfm_launch(const SRC& src, const uint8_t* gt, int N, int h, int w, const double* gauss49, int* near_y, int* d2, double* Et,
                      double* partial, int nparts, double* out3, hipStream_t st) {
    if (h <= WFM_SEG * WFM_MAXSEG)
        hipLaunchKernelGGL(wfm_colpass_kernel, dim3((w + 63) / 64, N), dim3(64, 16), 0, st, gt, h, w, near_y);
    else
        hipLaunchKernelGGL(wfm_colpass_tall_kernel, dim3((w + 255) / 256, N), dim3(256), 0, st, gt, h, w, near_y);
    CVLM_CHECK_LAUNCH();
    const int smem_row = 4 * w * (int)sizeof(int);                   // 128 KB at the widest supported row (w = 8192)
    if (smem_row > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)wfm_rowpass_kernel<SRC>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_row);
    hipLaunchKernelGGL(wfm_rowpass_kernel<SRC>, dim3(h, N), dim3(256), smem_row, st, src, gt, h, w, (const int*)near_y, d2, Et);
    CVLM_CHECK_LAUNCH();
    hipLaunchKernelGGL(wfm_weight_kernel<SRC>, dim3(nparts, N), dim3(256), 0, st, src, gt, h, w, (const int*)d2, (const double*)Et, gauss49,
                       partial);
    CVLM_CHECK_LAUNCH();
    hipLaunchKernelGGL(wfm_finalize_kernel, dim3(N), dim3(256), 0, st, (const double*)partial, nparts, out3);
    CVLM_CHECK_LAUNCH();
    return 0;
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
    // partials are parked in the histogram buffer itself: image n's 8 KB (1024 u64) hold its <= 256 triples
    const int gx = grid_for(((int64_t)h * w + 15) / 16, 256);         // 16 pixels per thread and round; <= 256 partials
    unsigned long long* partial = (unsigned long long*)hist;
    hipLaunchKernelGGL(gt_centroid_kernel, dim3(gx, N), dim3(256), 0, st, gt, h, w, partial);
    CVLM_CHECK_LAUNCH();
    hipLaunchKernelGGL(centroid_finalize_kernel, dim3(N), dim3(256), 0, st, partial, gx, (unsigned long long*)stats, hist);
    CVLM_CHECK_LAUNCH();
    hipLaunchKernelGGL(joint_hist_kernel, dim3(gx, N), dim3(256), 0, st, pre, gt, h, w, (const unsigned long long*)stats, hist);
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_mask_wfm(const uint8_t* pre, const uint8_t* gt, int32_t N, int32_t h, int32_t w, const uint32_t* hist,
                  const double* gauss49, void* workspace, double* out3, void* stream) {
    if (!pre || !gt || !hist || !gauss49 || !workspace || !out3 || N <= 0 || h <= 0 || w <= 0 || w > 8192 || h > 16384)
        return CVLM_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const size_t px = (size_t)N * h * w;
    int* near_y = (int*)workspace;
    int* d2 = near_y + px;
    double* Et = (double*)(d2 + px);                                 // 8-byte aligned: px * 8 bytes precede it
    const int nparts = (int)(((int64_t)h * w + 255) / 256);
    double* partial = Et + px;                                       // N * nparts * 3 doubles
    int* lohi = (int*)(partial + (size_t)N * nparts * 3);
    hipLaunchKernelGGL(wfm_levels_kernel, dim3(N), dim3(256), 0, st, (const unsigned*)hist, lohi);
    CVLM_CHECK_LAUNCH();
    return wfm_launch(WfmU8{pre, (const int*)lohi}, gt, N, h, w, gauss49, near_y, d2, Et, partial, nparts, out3, st);
}

// ---- utils.calc_cod on the device (ABI 8): float probability maps, no uint8 step (utils.py:143-165, sod_metric.py:12-26) ---------
int cvlm_prob_quantise(const float* prob, int32_t N, int32_t h, int32_t w, float* minmax, uint8_t* q, void* workspace, void* stream) {
    if (!prob || !minmax || !q || !workspace || N <= 0 || h <= 0 || w <= 0) return CVLM_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t hw = (int64_t)h * w;
    float* partial = (float*)workspace;                              // [N][COD_PARTS][2] at the head of the workspace
    hipLaunchKernelGGL(cod_minmax_kernel, dim3(COD_PARTS, N), dim3(256), 0, st, prob, hw, partial);
    CVLM_CHECK_LAUNCH();
    hipLaunchKernelGGL(cod_quant_kernel, dim3(grid_for(hw, 1024), N), dim3(256), 0, st, prob, hw, (const float*)partial, minmax, q);
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_prob_moments(const float* prob, const uint8_t* gt, int32_t N, int32_t h, int32_t w, const float* minmax, const uint64_t* stats,
                      void* workspace, double* out, void* stream) {
    if (!prob || !gt || !minmax || !stats || !workspace || !out || N <= 0 || h <= 0 || w <= 0) return CVLM_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    double* partial = (double*)workspace;                            // [N][COD_PARTS][16]
    hipLaunchKernelGGL(cod_moments_kernel, dim3(COD_PARTS, N), dim3(256), 0, st, prob, gt, h, w, minmax,
                       (const unsigned long long*)stats, partial);
    CVLM_CHECK_LAUNCH();
    hipLaunchKernelGGL(cod_moments_finalize_kernel, dim3(N), dim3(64), 0, st, (const double*)partial, out);
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_prob_wfm(const float* prob, const uint8_t* gt, int32_t N, int32_t h, int32_t w, const float* minmax, const double* gauss49,
                  void* workspace, double* out3, void* stream) {
    if (!prob || !gt || !minmax || !gauss49 || !workspace || !out3 || N <= 0 || h <= 0 || w <= 0 || w > 8192 || h > 16384)
        return CVLM_E_BADARG;
    const size_t px = (size_t)N * h * w;
    int* near_y = (int*)workspace;
    int* d2 = near_y + px;
    double* Et = (double*)(d2 + px);
    const int nparts = (int)(((int64_t)h * w + 255) / 256);
    double* partial = Et + px;
    return wfm_launch(WfmProb{prob, minmax}, gt, N, h, w, gauss49, near_y, d2, Et, partial, nparts, out3, (hipStream_t)stream);
}

int cvlm_topk_accumulate(const float* scores, const int32_t* labels, int32_t B, int32_t C, int32_t* pred, uint32_t* counters,
                         void* stream) {
    if (!scores || !labels || !counters || B <= 0 || C <= 0) return CVLM_E_BADARG;
    hipLaunchKernelGGL(topk_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, scores, labels, B, C, pred, counters);
    CVLM_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
