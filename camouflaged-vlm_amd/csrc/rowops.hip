// HBM-bound row / gather kernels of the path (LayerNorm, broadcasts, patch gathers, im2col, ...).
// All arithmetic fp32; outputs optionally emitted as split-half (h2) planes for the MFMA GEMMs.
// Loads/stores are 16-byte (float4) / 8-byte (half4) vectors, one wave per row for reductions.
#include "common.h"
#include "../../include/cvlm.h"

namespace {

__device__ __forceinline__ void store_h2x4(half_t* hi, half_t* lo, int64_t off, const float v[4]) {
    half_t h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split_h2(v[j], h[j], l[j]);
    *(half4*)(hi + off) = half4{h[0], h[1], h[2], h[3]};
    if (lo) *(half4*)(lo + off) = half4{l[0], l[1], l[2], l[3]};
}

// ------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row kept in registers (D <= 2048, D % 4 == 0)
// ------------------------------------------------------------------------------------------------
constexpr int LN_MAXV = 8;

__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ add, int add_rows,
                                                        float* __restrict__ sum_out,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, int act,
                                                        float* __restrict__ out_f32, half_t* __restrict__ out_hi,
                                                        half_t* __restrict__ out_lo, int M, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = D >> 2;
    float4 v[LN_MAXV];
    const float4* xr = (const float4*)(x + (int64_t)row * ldx);
    const float4* ar = add ? (const float4*)(add + (int64_t)(row % add_rows) * D) : nullptr;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            float4 t = xr[c];
            if (ar) { const float4 a = ar[c]; t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w; }
            v[i] = t;
            s += (t.x + t.y) + (t.z + t.w);
        } else {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (sum_out) {
        float4* so = (float4*)(sum_out + (int64_t)row * D);
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) { const int c = lane + i * 64; if (c < nv) so[c] = v[i]; }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
    const float4* gr = (const float4*)gamma;
    const float4* br = (const float4*)beta;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const float4 g = gr[c], b = br[c];
            float o[4];
            o[0] = (v[i].x - mean) * rstd * g.x + b.x;
            o[1] = (v[i].y - mean) * rstd * g.y + b.y;
            o[2] = (v[i].z - mean) * rstd * g.z + b.z;
            o[3] = (v[i].w - mean) * rstd * g.w + b.w;
            if (act != ACT_NONE) {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = apply_act(o[j], act);
            }
            const int64_t off = (int64_t)row * D + c * 4;
            if (out_f32) *(float4*)(out_f32 + off) = make_float4(o[0], o[1], o[2], o[3]);
            if (out_hi) store_h2x4(out_hi, out_lo, off, o);
        }
    }
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add_rows_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       int b_rows, float scale, float* __restrict__ out_f32,
                                                       half_t* __restrict__ out_hi, half_t* __restrict__ out_lo,
                                                       int64_t nvec, int dv) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / dv;
        const int c = (int)(i - row * dv);
        float4 t = ((const float4*)a)[i];
        if (b) {
            const float4 u = ((const float4*)b)[(row % b_rows) * dv + c];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        float o[4] = {t.x * scale, t.y * scale, t.z * scale, t.w * scale};
        if (out_f32) ((float4*)out_f32)[i] = make_float4(o[0], o[1], o[2], o[3]);
        if (out_hi) store_h2x4(out_hi, out_lo, i * 4, o);
    }
}

__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ s0, int C0,
                                                       const float* __restrict__ s1, int C1, int B, int H, int W,
                                                       int p, half_t* __restrict__ out_hi,
                                                       half_t* __restrict__ out_lo, int ldk) {
    const int gh = H / p, gw = W / p;
    const int kg = ldk >> 3;                                       // groups of 8 columns
    const int64_t total = (int64_t)B * gh * gw * kg;
    const int K = (C0 + C1) * p * p;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / kg;
        const int k0 = (int)(i - m * kg) * 8;
        const int px = (int)(m % gw);
        const int py = (int)((m / gw) % gh);
        const int b = (int)(m / ((int64_t)gw * gh));
        half_t hi[8], lo[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + j;
            float v = 0.f;
            if (k < K) {
                const int c = k / (p * p), r = k - c * p * p;
                const int iy = r / p, ix = r - iy * p;
                const float* src = c < C0 ? s0 + ((int64_t)(b * C0 + c) * H) * W
                                          : s1 + ((int64_t)(b * C1 + (c - C0)) * H) * W;
                v = src[(int64_t)(py * p + iy) * W + px * p + ix];
            }
            split_h2(v, hi[j], lo[j]);
        }
        *(half8*)(out_hi + m * ldk + k0) = half8{hi[0], hi[1], hi[2], hi[3], hi[4], hi[5], hi[6], hi[7]};
        if (out_lo) *(half8*)(out_lo + m * ldk + k0) = half8{lo[0], lo[1], lo[2], lo[3], lo[4], lo[5], lo[6], lo[7]};
    }
}

__global__ __launch_bounds__(256) void im2col3x3_kernel(const float* __restrict__ x, int B, int H, int W, int C,
                                                        half_t* __restrict__ out_hi, half_t* __restrict__ out_lo) {
    const int cg = C >> 3;
    const int64_t total = (int64_t)B * H * W * 9 * cg;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % cg) * 8;
        const int tap = (int)((i / cg) % 9);
        const int64_t m = i / ((int64_t)cg * 9);
        const int xx = (int)(m % W), yy = (int)((m / W) % H);
        const int b = (int)(m / ((int64_t)W * H));
        const int sy = yy + tap / 3 - 1, sx = xx + tap % 3 - 1;
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (sy >= 0 && sy < H && sx >= 0 && sx < W) {
            const float4* src = (const float4*)(x + (((int64_t)b * H + sy) * W + sx) * C + c0);
            const float4 a = src[0], d = src[1];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = d.x; v[5] = d.y; v[6] = d.z; v[7] = d.w;
        }
        half_t hi[8], lo[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) split_h2(v[j], hi[j], lo[j]);
        const int64_t off = m * (9 * C) + tap * C + c0;
        *(half8*)(out_hi + off) = half8{hi[0], hi[1], hi[2], hi[3], hi[4], hi[5], hi[6], hi[7]};
        if (out_lo) *(half8*)(out_lo + off) = half8{lo[0], lo[1], lo[2], lo[3], lo[4], lo[5], lo[6], lo[7]};
    }
}

// out[b][t][c] = xflat[b][c*T + t]: transpose of the (D x T) re-read of each image's token matrix
__global__ __launch_bounds__(256) void reinterpret_transpose_kernel(const float* __restrict__ x, int T, int D, float scale,
                                                                    half_t* __restrict__ out_hi,
                                                                    half_t* __restrict__ out_lo) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
    const float* xb = x + (int64_t)b * T * D;
    for (int r = ty; r < 32; r += 8) {                                 // read Y[c0+r][t0+tx]
        const int c = c0 + r, t = t0 + tx;
        tile[r][tx] = (c < D && t < T) ? xb[(int64_t)c * T + t] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {                                 // write out[t0+r][c0+tx]
        const int t = t0 + r, c = c0 + tx;
        if (t < T && c < D) {
            half_t h, l;
            split_h2(tile[tx][r] * scale, h, l);
            const int64_t off = ((int64_t)b * T + t) * D + c;
            out_hi[off] = h;
            if (out_lo) out_lo[off] = l;
        }
    }
}

__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ x, half_t* __restrict__ hi,
                                                    half_t* __restrict__ lo, int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 t = ((const float4*)x)[i];
        const float o[4] = {t.x, t.y, t.z, t.w};
        store_h2x4(hi, lo, i * 4, o);
    }
}

// one wave per row: h2 = x * scale; stats[p][row] = (sum, centred sum of squares) of columns [64p, 64p + 64) of the unscaled
// row -- the piece layout the LayerNorm-folded GEMM merges (include/cvlm.h).  A lane takes 8 consecutive columns, the 8 lanes
// of a piece exchange their partial sums.
__global__ __launch_bounds__(256) void row_stats_split_kernel(const float* __restrict__ x, float scale, half_t* __restrict__ hi,
                                                              half_t* __restrict__ lo, float* __restrict__ stats, int M, int D,
                                                              int64_t dst_row_stride, int64_t stats_rows) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (int64_t)row * D;
    // blockIdx.y: copy number -- the same M source rows land dst_row_stride rows further down for each copy
    hi += (int64_t)blockIdx.y * dst_row_stride * D;
    lo += (int64_t)blockIdx.y * dst_row_stride * D;
    stats += 2 * (int64_t)blockIdx.y * dst_row_stride;
    const int nchunk = D >> 3;
    for (int c0 = 0; c0 < nchunk; c0 += 64) {                         // wave-uniform trip count: the exchanges need every lane
        const int ch = c0 + lane;
        const bool live = ch < nchunk;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (live) {
            const float4 t0 = *(const float4*)(xr + ch * 8), t1 = *(const float4*)(xr + ch * 8 + 4);
            v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w; v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
            const float o0[4] = {t0.x * scale, t0.y * scale, t0.z * scale, t0.w * scale};
            const float o1[4] = {t1.x * scale, t1.y * scale, t1.z * scale, t1.w * scale};
            store_h2x4(hi, lo, (int64_t)row * D + ch * 8, o0);
            store_h2x4(hi, lo, (int64_t)row * D + ch * 8 + 4, o1);
        }
        const int piece = ch >> 3;
        const int np = D - piece * 64 < 64 ? D - piece * 64 : 64;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s1 += v[j];
        s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64); s1 += __shfl_xor(s1, 4, 64);
        const float pm = s1 * (np == 64 ? 0.015625f : 1.0f / (float)(np > 0 ? np : 1));
        if (live) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[j] - pm; s2 = fmaf(d, d, s2); }
        }
        s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64); s2 += __shfl_xor(s2, 4, 64);
        if (live && (lane & 7) == 0) *(float2*)(stats + 2 * ((int64_t)piece * stats_rows + row)) = make_float2(s1, s2);
    }
}

// cvlm_row_stats_split with the rows written as an mx operand (include/cvlm.h ABI 10: image + block exponents + fp16 lo plane) -- the
// CLIP tower's residual stream in the `mx` precision.  One wave per row, a lane owns 8 columns per pass: the four lanes of a 32-column
// block are a DPP quad, as in the GEMM epilogue that writes the same format (gemm_kernel.h, out_mx), and the bytes are the same
// (tests/test_gemm_mx_gpu.py holds both to hip.mx_pack).
__global__ __launch_bounds__(256) void row_stats_split_mx_kernel(const float* __restrict__ x, float scale, unsigned char* __restrict__ img,
                                                                 int64_t ld_img, unsigned char* __restrict__ sc, int64_t ld_s,
                                                                 half_t* __restrict__ lo, int64_t ld_lo, float* __restrict__ stats, int M,
                                                                 int D, int64_t dst_row_stride, int64_t stats_rows) {
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    typedef short s2v __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (int64_t)row * D;
    const int64_t drow = (int64_t)blockIdx.y * dst_row_stride + row;       // destination row of this copy
    stats += 2 * (int64_t)blockIdx.y * dst_row_stride;
    const int nchunk = D >> 3;
    for (int c0 = 0; c0 < nchunk; c0 += 64) {                         // wave-uniform trip count: the exchanges need every lane
        const int ch = c0 + lane;
        const bool live = ch < nchunk;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        unsigned hi4[4] = {0u, 0u, 0u, 0u}, lo4[4] = {0u, 0u, 0u, 0u};
        if (live) {
            const float4 t0 = *(const float4*)(xr + ch * 8), t1 = *(const float4*)(xr + ch * 8 + 4);
            v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w; v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
#pragma unroll
            for (int j = 0; j < 4; ++j) split_h2_pk(v[2 * j] * scale, v[2 * j + 1] * scale, hi4[j], lo4[j]);
        }
        // block exponent: largest |hi| of the 32 columns (D % 64 == 0: a quad is live or dead as a whole)
        h2v hv[4], lv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { hv[j] = __builtin_bit_cast(h2v, hi4[j]); lv[j] = __builtin_bit_cast(h2v, lo4[j]); }
        h2v mx2 = __builtin_elementwise_abs(hv[0]);
#pragma unroll
        for (int j = 1; j < 4; ++j) mx2 = __builtin_elementwise_max(mx2, __builtin_elementwise_abs(hv[j]));
        float bmax = fmaxf((float)mx2[0], (float)mx2[1]);
        bmax = fmaxf(bmax, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, bmax), 0xB1, 0xF, 0xF, true)));
        bmax = fmaxf(bmax, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, bmax), 0x4E, 0xF, 0xF, true)));
        int ex = (__builtin_bit_cast(int, bmax) >> 23) & 0xff;
        ex = (ex < 103 ? 103 : ex) - 7;
        const float s_hi = __builtin_bit_cast(float, ex << 23), s_lo = __builtin_bit_cast(float, (ex - 11) << 23);
        if (live) {
            s2v h8a = {0, 0}, h8b = {0, 0}, l8a = {0, 0}, l8b = {0, 0};
            h8a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(h8a, hv[0], s_hi, false);
            h8a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(h8a, hv[1], s_hi, true);
            h8b = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(h8b, hv[2], s_hi, false);
            h8b = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(h8b, hv[3], s_hi, true);
            l8a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(l8a, lv[0], s_lo, false);
            l8a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(l8a, lv[1], s_lo, true);
            l8b = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(l8b, lv[2], s_lo, false);
            l8b = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(l8b, lv[3], s_lo, true);
            const int n = ch * 8, j8 = (n & 63) >> 3;
            unsigned char* grp = img + (drow * ld_img + ((n >> 6) << 7)) * 2;                 // the 256-byte group of these columns
            *(uint4*)(grp + j8 * 16) = make_uint4(hi4[0], hi4[1], hi4[2], hi4[3]);
            *(uint2*)(grp + 128 + j8 * 8) = make_uint2(__builtin_bit_cast(unsigned, h8a), __builtin_bit_cast(unsigned, h8b));
            *(uint2*)(grp + 192 + j8 * 8) = make_uint2(__builtin_bit_cast(unsigned, l8a), __builtin_bit_cast(unsigned, l8b));
            if ((lane & 3) == 0) {
                unsigned char* e = sc + (drow * 4 + (j8 >> 2)) * ld_s + (n >> 6);
                e[0] = (unsigned char)ex;
                e[2 * ld_s] = (unsigned char)(ex - 11);
            }
            *(uint4*)(lo + drow * ld_lo + n) = make_uint4(lo4[0], lo4[1], lo4[2], lo4[3]);
        }
        const int piece = ch >> 3;
        const int np = D - piece * 64 < 64 ? D - piece * 64 : 64;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s1 += v[j];
        s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64); s1 += __shfl_xor(s1, 4, 64);
        const float pm = s1 * (np == 64 ? 0.015625f : 1.0f / (float)(np > 0 ? np : 1));
        if (live) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[j] - pm; s2 = fmaf(d, d, s2); }
        }
        s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64); s2 += __shfl_xor(s2, 4, 64);
        if (live && (lane & 7) == 0) *(float2*)(stats + 2 * ((int64_t)piece * stats_rows + row)) = make_float2(s1, s2);
    }
}


// Piece statistics -> the (rstd, mu * rstd) pair per row the LayerNorm-folded GEMM epilogue reads (include/cvlm.h).  One thread per
// row; the pieces of a row are added in index order (bit-reproducible), as centred moments:
//   mu = sum_p s1_p / D,   M2 = sum_p [ m2_p + n_p (s1_p / n_p - mu)^2 ],   rstd = 1 / sqrt(M2 / D + eps).
// A row with |mu| * rstd > max_ratio is refused: its pair becomes NaN (every output of that row of the folded GEMM is NaN then)
// and *refused (optional) counts it.
__global__ __launch_bounds__(64) void ln_stats_merge_kernel(const float* __restrict__ pieces, int64_t piece_rows, int M, int D, float eps,
                                                           float max_ratio, float* __restrict__ merged, unsigned* __restrict__ refused) {
    const int m = blockIdx.x * 64 + threadIdx.x;
    if (m >= M) return;
    const int P = (D + 63) >> 6;
    const float2* sp = (const float2*)pieces + m;
    const float inv_d = 1.0f / (float)D;
    auto centred = [&](float2 v, int pc, float mu) -> float {
        const int np = D - (pc << 6) < 64 ? D - (pc << 6) : 64;
        const float dm = v.x * (np == 64 ? 0.015625f : 1.0f / (float)np) - mu;
        return fmaf((float)np * dm, dm, v.y);
    };
    float mu, m2 = 0.f;
    constexpr int PMAX = 32;                                              // D <= 2048: every piece of the row in registers, one batch of loads
    if (P <= PMAX) {
        float2 v[PMAX];
#pragma unroll
        for (int pc = 0; pc < PMAX; ++pc) v[pc] = pc < P ? sp[(int64_t)pc * piece_rows] : make_float2(0.f, 0.f);
        float tot = 0.f;
#pragma unroll
        for (int pc = 0; pc < PMAX; ++pc) tot += pc < P ? v[pc].x : 0.f;
        mu = tot * inv_d;
#pragma unroll
        for (int pc = 0; pc < PMAX; ++pc) m2 += pc < P ? centred(v[pc], pc, mu) : 0.f;
    } else {
        float tot = 0.f;
        for (int pc = 0; pc < P; ++pc) tot += sp[(int64_t)pc * piece_rows].x;
        mu = tot * inv_d;
        for (int pc = 0; pc < P; ++pc) m2 += centred(sp[(int64_t)pc * piece_rows], pc, mu);
    }
    float rs = __builtin_amdgcn_rsqf(fmaxf(m2 * inv_d, 0.f) + eps);
    if (fabsf(mu) * rs > max_ratio) {
        rs = __builtin_nanf("");
        if (refused) atomicAdd(refused, 1u);
    }
    *(float2*)(merged + 2 * (int64_t)m) = make_float2(rs, mu * rs);
}

__global__ __launch_bounds__(256) void dense_pe_kernel(const float* __restrict__ gauss, int size, int C,
                                                       float* __restrict__ out) {
    const int half = C >> 1;
    const int64_t total = (int64_t)size * size * half;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % half);
        const int t = (int)(i / half);
        const int xx = t % size, yy = t / size;
        const float cx = 2.0f * (((float)xx + 0.5f) / (float)size) - 1.0f;
        const float cy = 2.0f * (((float)yy + 0.5f) / (float)size) - 1.0f;
        float v = cx * gauss[c] + cy * gauss[half + c];
        v = 6.283185307179586f * v;
        out[(int64_t)t * C + c] = sinf(v);
        out[(int64_t)t * C + half + c] = cosf(v);
    }
}

__global__ __launch_bounds__(256) void mask_head_kernel(const float* __restrict__ up, const float* __restrict__ edge,
                                                        const float* __restrict__ hyper, int HW, int C,
                                                        float* __restrict__ low) {
    const int b = blockIdx.y;
    const float* h0 = hyper + (int64_t)b * 5 * C;
    const float* h4 = h0 + 4 * C;
    for (int pix = blockIdx.x * blockDim.x + threadIdx.x; pix < HW; pix += gridDim.x * blockDim.x) {
        const float4* u = (const float4*)(up + ((int64_t)b * HW + pix) * C);
        float m = 0.f, g = 0.f;
        for (int c = 0; c < (C >> 2); ++c) {
            const float4 a = u[c];
            const float4 w0 = ((const float4*)h0)[c];
            m += a.x * w0.x + a.y * w0.y + a.z * w0.z + a.w * w0.w;
        }
        if (!edge) {                                             // vanilla decoder: plain hypernetwork product
            low[(int64_t)b * HW + pix] = m;
            continue;
        }
        const float4* e = (const float4*)(edge + ((int64_t)b * HW + pix) * C);
        for (int c = 0; c < (C >> 2); ++c) {
            const float4 d = e[c];
            const float4 w4 = ((const float4*)h4)[c];
            g += d.x * w4.x + d.y * w4.y + d.z * w4.z + d.w * w4.w;
        }
        const float s = 1.0f / (1.0f + expf(-g));
        low[(int64_t)b * HW + pix] = m * s + m;
    }
}

__global__ __launch_bounds__(256) void bilinear_kernel(const float* __restrict__ in, int hin, int win,
                                                       float* __restrict__ out, int hout, int wout, int sigmoid_in,
                                                       int64_t total) {
    const float sy = (float)hin / (float)hout, sx = (float)win / (float)wout;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % wout), oy = (int)((i / wout) % hout);
        const int64_t n = i / ((int64_t)wout * hout);
        float fy = sy * ((float)oy + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
        float fx = sx * ((float)ox + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < hin - 1 ? 1 : 0), x1 = x0 + (x0 < win - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float* p = in + n * (int64_t)hin * win;
        float v00 = p[(int64_t)y0 * win + x0], v01 = p[(int64_t)y0 * win + x1];
        float v10 = p[(int64_t)y1 * win + x0], v11 = p[(int64_t)y1 * win + x1];
        if (sigmoid_in) {
            v00 = 1.0f / (1.0f + expf(-v00)); v01 = 1.0f / (1.0f + expf(-v01));
            v10 = 1.0f / (1.0f + expf(-v10)); v11 = 1.0f / (1.0f + expf(-v11));
        }
        out[i] = (1.0f - ly) * ((1.0f - lx) * v00 + lx * v01) + ly * ((1.0f - lx) * v10 + lx * v11);
    }
}

__global__ __launch_bounds__(256) void clip_assemble_kernel(const float* __restrict__ patches,
                                                            const float* __restrict__ cls,
                                                            const float* __restrict__ pos,
                                                            const float* __restrict__ ctx, int P, int Wd, int nctx,
                                                            float* __restrict__ out, int64_t total) {
    const int L = 1 + P + nctx, wv = Wd >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % wv);
        const int t = (int)((i / wv) % L);
        const int64_t b = i / ((int64_t)wv * L);
        float4 v;
        if (t == 0) {
            const float4 a = ((const float4*)cls)[c], q = ((const float4*)pos)[c];
            v = make_float4(a.x + q.x, a.y + q.y, a.z + q.z, a.w + q.w);
        } else if (t <= P) {
            const float4 a = ((const float4*)patches)[(b * P + (t - 1)) * wv + c];
            const float4 q = ((const float4*)pos)[(int64_t)t * wv + c];
            v = make_float4(a.x + q.x, a.y + q.y, a.z + q.z, a.w + q.w);
        } else {
            v = ((const float4*)ctx)[(int64_t)(t - 1 - P) * wv + c];
        }
        ((float4*)out)[i] = v;
    }
}

__global__ __launch_bounds__(256) void overwrite_rows_kernel(float* __restrict__ x, int L, int Wd, int first, int n,
                                                             const float* __restrict__ src, int64_t total) {
    const int wv = Wd >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % wv);
        const int r = (int)((i / wv) % n);
        const int64_t b = i / ((int64_t)wv * n);
        ((float4*)x)[(b * L + first + r) * wv + c] = ((const float4*)src)[(int64_t)r * wv + c];
    }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x, int L, int Wd,
                                                          const int32_t* __restrict__ idx, int fixed,
                                                          float* __restrict__ out, int64_t total) {
    const int wv = Wd >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % wv);
        const int64_t b = i / wv;
        const int r = idx ? idx[b] : fixed;
        ((float4*)out)[i] = ((const float4*)x)[(b * L + r) * wv + c];
    }
}

// the same pick from an h2 stream: out[b] = (hi + lo)[b][r] * scale
__global__ __launch_bounds__(256) void gather_rows_h2_kernel(const half_t* __restrict__ hi, const half_t* __restrict__ lo, float scale,
                                                             int L, int Wd, const int32_t* __restrict__ idx, int fixed,
                                                             float* __restrict__ out, int64_t total) {
    const int wv = Wd >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % wv);
        const int64_t b = i / wv;
        const int r = idx ? idx[b] : fixed;
        const half4 h = ((const half4*)hi)[(b * L + r) * wv + c], l = ((const half4*)lo)[(b * L + r) * wv + c];
        ((float4*)out)[i] = make_float4(((float)h[0] + (float)l[0]) * scale, ((float)h[1] + (float)l[1]) * scale,
                                        ((float)h[2] + (float)l[2]) * scale, ((float)h[3] + (float)l[3]) * scale);
    }
}

// one 1024-thread block per image: the norm by the first four waves (the summation order of round 1), then sixteen waves share the
// classes.  A lane keeps its D / 64 scaled features in registers (round 1-3 divided by the norm again for every class: 192 divisions
// per lane and one dependent load after the other, 90 us for 61 classes -- 0.4 % of a one-image step) and asks for a class row's
// values before it multiplies: same products, same order of additions, same bits.
__global__ __launch_bounds__(1024) void clip_head_kernel(const float* __restrict__ img, const float* __restrict__ txt,
                                                         float lscale, int C, int D, float* __restrict__ img_n,
                                                         float* __restrict__ logits, int64_t* __restrict__ pred,
                                                         float* __restrict__ txt_sel) {
    constexpr int KMAX = 16;                                          // D <= 1024 on the register path
    __shared__ float red[4];
    __shared__ float slog[1024];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* x = img + (int64_t)b * D;
    if (tid < 256) {
        float s = 0.f;
        for (int d = tid; d < D; d += 256) s += x[d] * x[d];
        s = wave_sum(s);
        if (lane == 0) red[wave] = s;
    }
    __syncthreads();
    const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]);
    for (int d = tid; d < D; d += 1024) img_n[(int64_t)b * D + d] = x[d] / nrm;
    if (D <= 64 * KMAX) {
        float pr[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) { const int d = lane + 64 * k; pr[k] = d < D ? lscale * (x[d] / nrm) : 0.f; }
        for (int c = wave; c < C; c += 16) {
            const float* t = txt + (int64_t)c * D;
            float tv[KMAX];
#pragma unroll
            for (int k = 0; k < KMAX; ++k) { const int d = lane + 64 * k; tv[k] = d < D ? t[d] : 0.f; }
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) if (lane + 64 * k < D) a += pr[k] * tv[k];
            a = wave_sum(a);
            if (lane == 0) { slog[c] = a; logits[(int64_t)b * C + c] = a; }
        }
    } else {
        for (int c = wave; c < C; c += 16) {
            const float* t = txt + (int64_t)c * D;
            float a = 0.f;
            for (int d = lane; d < D; d += 64) a += (lscale * (x[d] / nrm)) * t[d];
            a = wave_sum(a);
            if (lane == 0) { slog[c] = a; logits[(int64_t)b * C + c] = a; }
        }
    }
    __syncthreads();
    if (tid == 0) {
        int best = 0; float bv = slog[0];
        for (int c = 1; c < C; ++c) if (slog[c] > bv) { bv = slog[c]; best = c; }
        pred[b] = best;
        red[0] = __int_as_float(best);
    }
    __syncthreads();
    const int best = __float_as_int(red[0]);
    for (int d = tid; d < D; d += 1024) txt_sel[(int64_t)b * D + d] = txt[(int64_t)best * D + d];
}

__global__ __launch_bounds__(64) void normalize_add_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                                           int D, float* __restrict__ out) {
    const int r = blockIdx.x, lane = threadIdx.x;
    const float* xr = x + (int64_t)r * D;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += xr[d] * xr[d];
    const float nrm = sqrtf(wave_sum(s));
    for (int d = lane; d < D; d += 64) out[(int64_t)r * D + d] = xr[d] / nrm + (add ? add[(int64_t)r * D + d] : 0.f);
}

// ------------------------------------------------------------------------------------------------
// small fp32 attention (two-way decoder).  hd = 16 or 32.
// ------------------------------------------------------------------------------------------------

// Output of a small-attention row piece: f32 and / or h2 planes (the out_proj GEMM's operand: no cvlm_split_f32 launch in between)
template <int HD>
__device__ __forceinline__ void sa_store(const float (&o)[HD], float* __restrict__ of, half_t* __restrict__ oh, half_t* __restrict__ ol) {
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
        if (of) *(float4*)(of + d) = make_float4(o[d], o[d + 1], o[d + 2], o[d + 3]);
        if (oh) {
            half_t h[4], l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split_h2(o[d + j], h[j], l[j]);
            *(half4*)(oh + d) = half4{h[0], h[1], h[2], h[3]};
            *(half4*)(ol + d) = half4{l[0], l[1], l[2], l[3]};
        }
    }
}

// few keys: one thread per (b, q, h); HD = 16 / 32 (the decoder's head dims), 16-byte loads and stores
template <int HD>
__global__ __launch_bounds__(256) void small_attn_thread_kernel(const float* __restrict__ q, int64_t ldq,
                                                                const float* __restrict__ k, int64_t ldk,
                                                                const float* __restrict__ v, int64_t ldv,
                                                                float* __restrict__ out, int64_t ldo, half_t* __restrict__ out_hi,
                                                                half_t* __restrict__ out_lo, int64_t ldoh, int nq, int nk,
                                                                int heads, int64_t total) {
    const float sc = 1.0f / sqrtf((float)HD);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int h = (int)(i % heads);
        const int qi = (int)((i / heads) % nq);
        const int64_t b = i / ((int64_t)heads * nq);
        const float* qp = q + (b * nq + qi) * ldq + h * HD;
        float qr[HD], acc[HD];
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
            const float4 a = *(const float4*)(qp + d);
            qr[d] = a.x; qr[d + 1] = a.y; qr[d + 2] = a.z; qr[d + 3] = a.w;
            acc[d] = acc[d + 1] = acc[d + 2] = acc[d + 3] = 0.f;
        }
        float mx = -INFINITY, l = 0.f;
        for (int j = 0; j < nk; ++j) {
            const float* kp = k + (b * nk + j) * ldk + h * HD;
            const float* vp = v + (b * nk + j) * ldv + h * HD;
            float s = 0.f, vr[HD];
#pragma unroll
            for (int d = 0; d < HD; d += 4) {
                const float4 a = *(const float4*)(kp + d), c = *(const float4*)(vp + d);
                s += qr[d] * a.x; s += qr[d + 1] * a.y; s += qr[d + 2] * a.z; s += qr[d + 3] * a.w;
                vr[d] = c.x; vr[d + 1] = c.y; vr[d + 2] = c.z; vr[d + 3] = c.w;
            }
            s *= sc;
            const float mn = fmaxf(mx, s);
            const float f = expf(mx - mn), pj = expf(s - mn);
            l = l * f + pj;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] = acc[d] * f + pj * vr[d];
            mx = mn;
        }
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] = acc[d] / l;
        const int64_t row = b * nq + qi;
        sa_store<HD>(acc, out ? out + row * ldo + h * HD : nullptr, out_hi ? out_hi + row * ldoh + h * HD : nullptr,
                     out_lo ? out_lo + row * ldoh + h * HD : nullptr);
    }
}

// many keys: one workgroup (4 waves) per (b, q, h); threads stride over keys, FOUR keys' rows requested before the first is used
// (one key per trip made every trip wait a full load latency: 28 us for 16 trips), partial softmax states are merged through LDS.
template <int HD>
__global__ __launch_bounds__(256) void small_attn_wave_kernel(const float* __restrict__ q, int64_t ldq,
                                                              const float* __restrict__ k, int64_t ldk,
                                                              const float* __restrict__ v, int64_t ldv,
                                                              float* __restrict__ out, int64_t ldo, half_t* __restrict__ out_hi,
                                                              half_t* __restrict__ out_lo, int64_t ldoh, int nq, int nk, int heads) {
    constexpr int U = 4;
    const float sc = 1.0f / sqrtf((float)HD);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = blockIdx.x;
    const int h = (int)(i % heads);
    const int qi = (int)((i / heads) % nq);
    const int64_t b = i / ((int64_t)heads * nq);
    const float* qp = q + (b * nq + qi) * ldq + h * HD;
    float qr[HD], acc[HD];
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
        const float4 a = *(const float4*)(qp + d);
        qr[d] = a.x; qr[d + 1] = a.y; qr[d + 2] = a.z; qr[d + 3] = a.w;
        acc[d] = acc[d + 1] = acc[d + 2] = acc[d + 3] = 0.f;
    }
    float mx = -INFINITY, l = 0.f;
    for (int j0 = threadIdx.x; j0 < nk; j0 += 256 * U) {
        float4 kr[U][HD / 4], vr[U][HD / 4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + 256 * u;
            const int jc = j < nk ? j : nk - 1;
            const float* kp = k + (b * nk + jc) * ldk + h * HD;
            const float* vp = v + (b * nk + jc) * ldv + h * HD;
#pragma unroll
            for (int d = 0; d < HD / 4; ++d) { kr[u][d] = *(const float4*)(kp + 4 * d); vr[u][d] = *(const float4*)(vp + 4 * d); }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (j0 + 256 * u >= nk) break;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < HD / 4; ++d)
                s += qr[4 * d] * kr[u][d].x + qr[4 * d + 1] * kr[u][d].y + qr[4 * d + 2] * kr[u][d].z + qr[4 * d + 3] * kr[u][d].w;
            s *= sc;
            const float mn = fmaxf(mx, s);
            const float f = expf(mx - mn), pj = expf(s - mn);
            l = l * f + pj;
#pragma unroll
            for (int d = 0; d < HD / 4; ++d) {
                acc[4 * d] = acc[4 * d] * f + pj * vr[u][d].x;         acc[4 * d + 1] = acc[4 * d + 1] * f + pj * vr[u][d].y;
                acc[4 * d + 2] = acc[4 * d + 2] * f + pj * vr[u][d].z; acc[4 * d + 3] = acc[4 * d + 3] * f + pj * vr[u][d].w;
            }
            mx = mn;
        }
    }
    // wave-level merge, then the four waves through LDS
    __shared__ float red[4][HD + 2];
    const float M = wave_max(mx);
    const float f = (mx == -INFINITY) ? 0.f : expf(mx - M);
    l = wave_sum(l * f);
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        const float a = wave_sum(acc[d] * f);
        if (lane == 0) red[wave][d] = a;
    }
    if (lane == 0) { red[wave][HD] = M; red[wave][HD + 1] = l; }
    __syncthreads();
    if (threadIdx.x < HD) {
        const float Mb = fmaxf(fmaxf(red[0][HD], red[1][HD]), fmaxf(red[2][HD], red[3][HD]));
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float fw = (red[w][HD] == -INFINITY) ? 0.f : expf(red[w][HD] - Mb);
            num += red[w][threadIdx.x] * fw;
            den += red[w][HD + 1] * fw;
        }
        const float o = num / den;
        const int64_t row = b * nq + qi;
        if (out) out[row * ldo + h * HD + threadIdx.x] = o;
        if (out_hi) {
            half_t hh, ll;
            split_h2(o, hh, ll);
            out_hi[row * ldoh + h * HD + threadIdx.x] = hh;
            out_lo[row * ldoh + h * HD + threadIdx.x] = ll;
        }
    }
}

inline int grid_for(int64_t n, int block = 256, int cap = 8192) {
    int64_t g = (n + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" {

int cvlm_layernorm(const float* x, int64_t ldx, const float* add, int32_t add_rows, float* sum_out,
                   const float* gamma, const float* beta, float eps, int32_t act, float* out_f32, void* out_hi,
                   void* out_lo, int32_t M, int32_t D, void* stream) {
    if (!x || !gamma || !beta || M <= 0 || D <= 0 || (D & 3) || D > LN_MAXV * 256 || (ldx & 3)) return CVLM_E_BADARG;
    if (add && add_rows <= 0) return CVLM_E_BADARG;
    hipLaunchKernelGGL(layernorm_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, add, add_rows,
                       sum_out, gamma, beta, eps, act, out_f32, (half_t*)out_hi, (half_t*)out_lo, M, D);
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_add_rows(const float* a, const float* b, int32_t b_rows, float scale, float* out_f32, void* out_hi,
                  void* out_lo, int32_t M, int32_t D, void* stream) {
    if (!a || M <= 0 || D <= 0 || (D & 3) || (b && b_rows <= 0)) return CVLM_E_BADARG;
    const int64_t nvec = (int64_t)M * (D >> 2);
    hipLaunchKernelGGL(add_rows_kernel, dim3(grid_for(nvec)), dim3(256), 0, (hipStream_t)stream, a, b, b_rows, scale,
                       out_f32, (half_t*)out_hi, (half_t*)out_lo, nvec, D >> 2);
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_row_stats_split(const float* x, float scale, void* out_hi, void* out_lo, float* stats, int64_t stats_rows, int32_t M,
                         int32_t D, int32_t copies, int64_t dst_row_stride, void* stream) {
    if (!x || !out_hi || !out_lo || !stats || M <= 0 || D <= 0 || (D & 7) || copies < 1 || copies > 65535) return CVLM_E_BADARG;
    if (copies > 1 && dst_row_stride < M) return CVLM_E_BADARG;
    if (stats_rows < M + (int64_t)(copies - 1) * dst_row_stride) return CVLM_E_BADARG;
    hipLaunchKernelGGL(row_stats_split_kernel, dim3((M + 3) / 4, copies), dim3(256), 0, (hipStream_t)stream, x, scale, (half_t*)out_hi,
                       (half_t*)out_lo, stats, M, D, dst_row_stride, stats_rows);
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_row_stats_split_mx(const float* x, float scale, void* out_img, int64_t ld_img, void* out_scales, int64_t ld_s, void* out_lo,
                            int64_t ld_lo, float* stats, int64_t stats_rows, int32_t M, int32_t D, int32_t copies, int64_t dst_row_stride,
                            void* stream) {
    if (!x || !out_img || !out_scales || !out_lo || !stats || M <= 0 || D <= 0 || (D & 63) || copies < 1 || copies > 65535) return CVLM_E_BADARG;
    if (ld_img < 2 * (int64_t)D || (ld_img & 7) || ld_s * 64 < D || (ld_lo & 7) || ld_lo < D) return CVLM_E_BADARG;
    if (copies > 1 && dst_row_stride < M) return CVLM_E_BADARG;
    if (stats_rows < M + (int64_t)(copies - 1) * dst_row_stride) return CVLM_E_BADARG;
    hipLaunchKernelGGL(row_stats_split_mx_kernel, dim3((M + 3) / 4, copies), dim3(256), 0, (hipStream_t)stream, x, scale,
                       (unsigned char*)out_img, ld_img, (unsigned char*)out_scales, ld_s, (half_t*)out_lo, ld_lo, stats, M, D,
                       dst_row_stride, stats_rows);
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_ln_stats_merge(const float* pieces, int64_t piece_rows, int32_t M, int32_t D, float eps, float* merged, void* gemm_workspace,
                        void* stream) {
    if (!pieces || !merged || M <= 0 || D <= 0 || piece_rows < M) return CVLM_E_BADARG;
    unsigned* refused = gemm_workspace ? (unsigned*)gemm_workspace + CVLM_WS_WORD_LN_REFUSED : nullptr;
    hipLaunchKernelGGL(ln_stats_merge_kernel, dim3((M + 63) / 64), dim3(64), 0, (hipStream_t)stream, pieces, piece_rows, M, D, eps,
                       (float)CVLM_LN_FOLD_MAX_RATIO, merged, refused);
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_split_f32(const float* x, void* out_hi, void* out_lo, int64_t n, void* stream) {
    if (!x || !out_hi || n <= 0 || (n & 3)) return CVLM_E_BADARG;
    hipLaunchKernelGGL(split_kernel, dim3(grid_for(n >> 2)), dim3(256), 0, (hipStream_t)stream, x, (half_t*)out_hi,
                       (half_t*)out_lo, n >> 2);
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_patchify(const float* src0, int32_t C0, const float* src1, int32_t C1, int32_t B, int32_t H, int32_t W,
                  int32_t p, void* out_hi, void* out_lo, int32_t ldk, void* stream) {
    if (!src0 || !out_hi || p <= 0 || (H % p) || (W % p) || (ldk & 7) || ldk < (C0 + C1) * p * p) return CVLM_E_BADARG;
    if (C1 > 0 && !src1) return CVLM_E_BADARG;
    const int64_t total = (int64_t)B * (H / p) * (W / p) * (ldk >> 3);
    hipLaunchKernelGGL(patchify_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src0, C0, src1, C1,
                       B, H, W, p, (half_t*)out_hi, (half_t*)out_lo, ldk);
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_im2col3x3(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, void* out_hi, void* out_lo,
                   void* stream) {
    if (!x || !out_hi || (C & 7)) return CVLM_E_BADARG;
    const int64_t total = (int64_t)B * H * W * 9 * (C >> 3);
    hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0, (hipStream_t)stream, x, B, H,
                       W, C, (half_t*)out_hi, (half_t*)out_lo);
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_reinterpret_transpose(const float* x, int32_t B, int32_t T, int32_t D, float scale, void* out_hi, void* out_lo,
                               void* stream) {
    if (!x || !out_hi || B <= 0 || T <= 0 || D <= 0) return CVLM_E_BADARG;
    hipLaunchKernelGGL(reinterpret_transpose_kernel, dim3((T + 31) / 32, (D + 31) / 32, B), dim3(256), 0,
                       (hipStream_t)stream, x, T, D, scale, (half_t*)out_hi, (half_t*)out_lo);
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_dense_pe(const float* gauss, int32_t size, int32_t C, float* out, void* stream) {
    if (!gauss || !out || size <= 0 || (C & 1)) return CVLM_E_BADARG;
    hipLaunchKernelGGL(dense_pe_kernel, dim3(grid_for((int64_t)size * size * (C >> 1))), dim3(256), 0,
                       (hipStream_t)stream, gauss, size, C, out);
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_mask_head(const float* up, const float* edge_emb, const float* hyper, int32_t B, int32_t HW, int32_t C,
                   float* low, void* stream) {
    if (!up || !hyper || !low || (C & 3)) return CVLM_E_BADARG;
    hipLaunchKernelGGL(mask_head_kernel, dim3(grid_for(HW, 256, 1024), B), dim3(256), 0, (hipStream_t)stream, up,
                       edge_emb, hyper, HW, C, low);
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_bilinear(const float* in, int32_t N, int32_t hin, int32_t win, float* out, int32_t hout, int32_t wout,
                  int32_t sigmoid_in, void* stream) {
    if (!in || !out || N <= 0) return CVLM_E_BADARG;
    const int64_t total = (int64_t)N * hout * wout;
    hipLaunchKernelGGL(bilinear_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, hin, win, out,
                       hout, wout, sigmoid_in, total);
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_clip_assemble(const float* patches, const float* cls, const float* pos, const float* ctx, int32_t B,
                       int32_t P, int32_t W, int32_t nctx, float* out, void* stream) {
    if (!patches || !cls || !pos || !ctx || !out || (W & 3)) return CVLM_E_BADARG;
    const int64_t total = (int64_t)B * (1 + P + nctx) * (W >> 2);
    hipLaunchKernelGGL(clip_assemble_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, patches, cls,
                       pos, ctx, P, W, nctx, out, total);
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_overwrite_rows(float* x, int32_t B, int32_t L, int32_t W, int32_t first, int32_t n, const float* src,
                        void* stream) {
    if (!x || !src || (W & 3) || first < 0 || first + n > L) return CVLM_E_BADARG;
    const int64_t total = (int64_t)B * n * (W >> 2);
    hipLaunchKernelGGL(overwrite_rows_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, L, W, first,
                       n, src, total);
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_gather_rows(const float* x, int32_t B, int32_t L, int32_t W, const int32_t* idx, int32_t fixed, float* out,
                     void* stream) {
    if (!x || !out || (W & 3)) return CVLM_E_BADARG;
    const int64_t total = (int64_t)B * (W >> 2);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, L, W, idx,
                       fixed, out, total);
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_gather_rows_h2(const void* x_hi, const void* x_lo, float scale, int32_t B, int32_t L, int32_t W, const int32_t* idx,
                        int32_t fixed, float* out, void* stream) {
    if (!x_hi || !x_lo || !out || (W & 3) || B <= 0 || L <= 0) return CVLM_E_BADARG;
    const int64_t total = (int64_t)B * (W >> 2);
    hipLaunchKernelGGL(gather_rows_h2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const half_t*)x_hi,
                       (const half_t*)x_lo, scale, L, W, idx, fixed, out, total);
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_clip_head(const float* img, const float* txt, float logit_scale_exp, int32_t B, int32_t C, int32_t D,
                   float* img_n, float* logits, int64_t* pred, float* txt_sel, void* stream) {
    if (!img || !txt || !img_n || !logits || !pred || !txt_sel || C <= 0 || C > 1024) return CVLM_E_BADARG;
    hipLaunchKernelGGL(clip_head_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, img, txt, logit_scale_exp, C, D,
                       img_n, logits, pred, txt_sel);
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_normalize_add(const float* x, const float* add, int32_t R, int32_t D, float* out, void* stream) {
    if (!x || !out || R <= 0) return CVLM_E_BADARG;
    hipLaunchKernelGGL(normalize_add_kernel, dim3(R), dim3(64), 0, (hipStream_t)stream, x, add, D, out);
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_small_attention_h2(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, float* out, int64_t ldo,
                            void* out_hi, void* out_lo, int64_t ldoh, int32_t B, int32_t nq, int32_t nk, int32_t heads, int32_t hd,
                            void* stream) {
    if (!q || !k || !v || (!out && !out_hi) || (out_hi && !out_lo) || nk <= 0 || nq <= 0 || B <= 0 || heads <= 0) return CVLM_E_BADARG;
    if (hd != 16 && hd != 32) return CVLM_E_UNSUPPORTED;             // the decoder's head dims (internal 128 / 256 over 8 heads)
    // 16-byte accesses: every row pitch and base a multiple of four floats (four halves for the planes)
    if (((ldq | ldk | ldv | ldo | ldoh) & 3) || (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) ||
        (((uintptr_t)out_hi | (uintptr_t)out_lo) & 7))
        return CVLM_E_BADARG;
    const int64_t total = (int64_t)B * nq * heads;
    hipStream_t s = (hipStream_t)stream;
    half_t* oh = (half_t*)out_hi;
    half_t* ol = (half_t*)out_lo;
    if (nk <= 64) {
        if (hd == 16) hipLaunchKernelGGL(small_attn_thread_kernel<16>, dim3(grid_for(total)), dim3(256), 0, s, q, ldq, k, ldk, v, ldv, out, ldo, oh, ol, ldoh, nq, nk, heads, total);
        else hipLaunchKernelGGL(small_attn_thread_kernel<32>, dim3(grid_for(total)), dim3(256), 0, s, q, ldq, k, ldk, v, ldv, out, ldo, oh, ol, ldoh, nq, nk, heads, total);
    } else {
        if (hd == 16) hipLaunchKernelGGL(small_attn_wave_kernel<16>, dim3((unsigned)total), dim3(256), 0, s, q, ldq, k, ldk, v, ldv, out, ldo, oh, ol, ldoh, nq, nk, heads);
        else hipLaunchKernelGGL(small_attn_wave_kernel<32>, dim3((unsigned)total), dim3(256), 0, s, q, ldq, k, ldk, v, ldv, out, ldo, oh, ol, ldoh, nq, nk, heads);
    }
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_small_attention(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                         float* out, int64_t ldo, int32_t B, int32_t nq, int32_t nk, int32_t heads, int32_t hd,
                         void* stream) {
    return cvlm_small_attention_h2(q, ldq, k, ldk, v, ldv, out, ldo, nullptr, nullptr, 0, B, nq, nk, heads, hd, stream);
}

int cvlm_abi_version(void) { return CVLM_ABI_VERSION; }
const char* cvlm_target_arch(void) { return "gfx950"; }

}  // extern "C"
