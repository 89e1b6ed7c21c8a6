// Split-half MFMA GEMM for gfx950:  out = post(act(alpha * A . W^T + bias) + residual)
//
// Both operands arrive as fp16 planes with K contiguous ("NT" form: activations [M][K], torch
// Linear weights [N][K]).  A 128x128 output tile per 256-thread workgroup (2x2 waves, 64x64 per
// wave, 4x4 MFMA 16x16x32 f16 tiles), BK = 32, LDS double-buffered and filled by 16-byte
// global->LDS DMA (global_load_lds_dwordx4) so the next K-tile streams in under the MFMAs of the
// current one.  The LDS image is lane-linear (DMA constraint); bank conflicts of the ds_read_b128
// fragment reads are removed by permuting the 16-byte chunks of each 64-byte row on the *source*
// address and applying the same involution on the read (guide §5.4 rule 21).
//
// The MFMA is issued "swapped" (A-operand = weight rows, B-operand = activation rows) so that each
// lane ends up with 4 consecutive output columns of one output row: the epilogue then stores
// 16-byte float4 / 8-byte half4 vectors instead of scalars.
//
// split == 3: acc += Whi.Ahi + Wlo.Ahi + Whi.Alo  (fp32 accumulate; ~2^-22 relative products)
// split == 1: acc += Whi.Ahi
#include "common.h"
#include "../../include/cvlm.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int PLANE_BYTES = 128 * BK * 2;          // one 128-row fp16 plane of a K-tile: 8 KiB

// chunk permutation g(q), q = (row >> 2) & 3 (derived for the ds_read_b128 lane groups)
__device__ __forceinline__ int swz4(int q) { return (0x78 >> (2 * q)) & 3; }

struct GemmParams {
    cvlm_gemm_args a;
    int nbx, nby;
};

template <int SPLIT>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const GemmParams p) {
    constexpr int NPL = (SPLIT == 3) ? 4 : 2;          // planes per stage: Ahi [Alo] Whi [Wlo]
    constexpr int STAGE = NPL * PLANE_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const cvlm_gemm_args& g = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- tile coordinates: XCD-aware bijective remap of the 1-D tile id (8 XCDs, round-robin dispatch)
    const int ntiles = p.nbx * p.nby;
    int pid = blockIdx.x;
    {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = pid & 7, idx = pid >> 3;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int by = pid / p.nbx, bx = pid - by * p.nbx;
    const int bm = by * BM, bn = bx * BN;
    const int z = blockIdx.y;

    const half_t* Ahi = (const half_t*)g.a_hi + (int64_t)z * g.stride_a;
    const half_t* Alo = (const half_t*)g.a_lo + (int64_t)z * g.stride_a;
    const half_t* Whi = (const half_t*)g.w_hi + (int64_t)z * g.stride_w;
    const half_t* Wlo = (const half_t*)g.w_lo + (int64_t)z * g.stride_w;

    // ---- staging assignment: NPL*8 wave-instructions (16 rows x 64 B each) per stage, spread over 4 waves
    constexpr int PER_WAVE = NPL * 8 / 4;
    const half_t* src[PER_WAVE];
    int dst_off[PER_WAVE];
    {
        const int rsub = lane >> 2;                                   // row within the 16-row group
        const int chunk = (lane & 3) ^ swz4((lane >> 4) & 3);         // source chunk for LDS position lane&3
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j) {
            const int i = wave * PER_WAVE + j;
            const int plane = i >> 3, sub = i & 7;
            const int row = sub * 16 + rsub;
            const bool isW = (SPLIT == 3) ? (plane >= 2) : (plane >= 1);
            const bool isLo = (SPLIT == 3) ? (plane & 1) : false;
            const half_t* base = isW ? (isLo ? Wlo : Whi) : (isLo ? Alo : Ahi);
            const int64_t ld = isW ? g.ldw : g.lda;
            int grow = (isW ? bn : bm) + row;
            const int lim = (isW ? g.N : g.M) - 1;
            grow = grow < lim ? grow : lim;
            src[j] = base + (int64_t)grow * ld + chunk * 8;
            dst_off[j] = plane * PLANE_BYTES + sub * 1024;
        }
    }

    // ---- fragment read offsets (bytes within a plane)
    const int fr = lane & 15, fq = lane >> 4;
    const int rswz = (fq ^ swz4((lane >> 2) & 3)) * 16;
    const int a_off = (wm * 64 + fr) * 64 + rswz;       // + mt*16*64
    const int w_off = (wn * 64 + fr) * 64 + rswz;       // + nt*16*64

    floatx4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nk = g.K / BK;
    // prologue: stage K-tile 0 into buffer 0
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) glds16(src[j], smem + dst_off[j]);
    __syncthreads();                                     // emits vmcnt(0) for the in-flight DMA

    for (int t = 0; t < nk; ++t) {
        unsigned char* cur = smem + (t & 1) * STAGE;
        if (t + 1 < nk) {
            unsigned char* nxt = smem + ((t + 1) & 1) * STAGE;
#pragma unroll
            for (int j = 0; j < PER_WAVE; ++j) glds16(src[j] + (int64_t)(t + 1) * BK, nxt + dst_off[j]);
        }
        const unsigned char* pAhi = cur;
        const unsigned char* pAlo = cur + PLANE_BYTES;
        const unsigned char* pWhi = cur + (SPLIT == 3 ? 2 : 1) * PLANE_BYTES;
        const unsigned char* pWlo = cur + 3 * PLANE_BYTES;

        half8 ah[4], wh[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ah[i] = *(const half8*)(pAhi + a_off + i * 1024);
            wh[i] = *(const half8*)(pWhi + w_off + i * 1024);
        }
        if (SPLIT == 3) {
            half8 al[4], wl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                al[i] = *(const half8*)(pAlo + a_off + i * 1024);
                wl[i] = *(const half8*)(pWlo + w_off + i * 1024);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[nt], ah[mt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt], al[mt], acc[mt][nt], 0, 0, 0);
                }
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt], ah[mt], acc[mt][nt], 0, 0, 0);
        __syncthreads();                                 // next tile landed (vmcnt(0)) and cur fully read
    }

    // ---- epilogue: lane holds out[m][n..n+3], m = .. + (lane&15), n = .. + (lane>>4)*4
    const float alpha = g.alpha;
    const bool vec_f32 = ((g.ldo & 3) == 0) && ((g.stride_o & 3) == 0);
    const bool vec_res = ((g.ldr & 3) == 0) && ((g.stride_r & 3) == 0);
    const bool vec_h = ((g.ldoh & 3) == 0) && ((g.stride_oh & 3) == 0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = bm + wm * 64 + mt * 16 + fr;
        if (m >= g.M) continue;
        int64_t ps_base = 0;
        if (g.ps_c2 > 0) {
            const int x = m % g.ps_w, t = m / g.ps_w;
            const int y = t % g.ps_h, b = t / g.ps_h;
            ps_base = ((int64_t)(b * 2 * g.ps_h + 2 * y) * (2 * g.ps_w) + 2 * x) * (g.ps_c2 >> 1);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = bn + wn * 64 + nt * 16 + fq * 4;
            if (n >= g.N) continue;
            const bool full = (n + 3 < g.N);
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[mt][nt][j] * alpha;
            if (g.bias) {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (n + j < g.N) v[j] += g.bias[n + j];
            }
            if (g.act != ACT_NONE && g.act != ACT_ABS_POST) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], g.act);
            }
            if (g.residual) {
                const float* r = g.residual + (int64_t)z * g.stride_r + (int64_t)m * g.ldr + n;
                if (full && vec_res) {
                    const float4 rv = *(const float4*)r;
                    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (n + j < g.N) v[j] += r[j];
                }
            }
            if (g.act == ACT_ABS_POST) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fabsf(v[j]);
            }
            int64_t off_f, off_h;
            if (g.ps_c2 > 0) {
                const int dy = n / g.ps_c2, r = n - dy * g.ps_c2;
                off_f = off_h = ps_base + (int64_t)dy * (2 * g.ps_w) * (g.ps_c2 >> 1) + r;
            } else {
                off_f = (int64_t)m * g.ldo + n;
                off_h = (int64_t)m * g.ldoh + n;
            }
            if (g.out_f32) {
                float* o = g.out_f32 + (int64_t)z * g.stride_o + off_f;
                if (full && vec_f32 && ((off_f & 3) == 0)) {
                    *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (n + j < g.N) o[j] = v[j];
                }
            }
            if (g.out_hi) {
                half_t hi[4], lo[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) split_h2(v[j], hi[j], lo[j]);
                half_t* oh = (half_t*)g.out_hi + (int64_t)z * g.stride_oh + off_h;
                half_t* ol = g.out_lo ? (half_t*)g.out_lo + (int64_t)z * g.stride_oh + off_h : nullptr;
                if (full && vec_h && ((off_h & 3) == 0)) {
                    *(half4*)oh = half4{hi[0], hi[1], hi[2], hi[3]};
                    if (ol) *(half4*)ol = half4{lo[0], lo[1], lo[2], lo[3]};
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (n + j < g.N) { oh[j] = hi[j]; if (ol) ol[j] = lo[j]; }
                }
            }
        }
    }
}

}  // namespace

extern "C" int cvlm_gemm(const cvlm_gemm_args* args, void* stream) {
    if (!args || !args->a_hi || !args->w_hi) return CVLM_E_BADARG;
    const cvlm_gemm_args& g = *args;
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || (g.K % BK) != 0) return CVLM_E_BADARG;
    if ((g.lda & 7) || (g.ldw & 7) || (g.stride_a & 7) || (g.stride_w & 7)) return CVLM_E_BADARG;
    if (g.split != 1 && g.split != 3) return CVLM_E_BADARG;
    if (g.split == 3 && (!g.a_lo || !g.w_lo)) return CVLM_E_BADARG;
    if (!g.out_f32 && !g.out_hi) return CVLM_E_BADARG;
    if (g.ps_c2 > 0 && ((g.ps_c2 & 3) || g.ps_h <= 0 || g.ps_w <= 0)) return CVLM_E_BADARG;
    GemmParams p;
    p.a = g;
    if (p.a.batch <= 0) p.a.batch = 1;
    p.nbx = (g.N + BN - 1) / BN;
    p.nby = (g.M + BM - 1) / BM;
    dim3 grid(p.nbx * p.nby, p.a.batch), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (g.split == 3) {
        constexpr int smem = 2 * 4 * PLANE_BYTES;
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr = true; }
        hipLaunchKernelGGL(gemm_nt_kernel<3>, grid, block, smem, s, p);
    } else {
        constexpr int smem = 2 * 2 * PLANE_BYTES;
        hipLaunchKernelGGL(gemm_nt_kernel<1>, grid, block, smem, s, p);
    }
    CVLM_CHECK_LAUNCH();
    return 0;
}
