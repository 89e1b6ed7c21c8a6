// Power roofline of the matrix pipe: an MFMA-only loop (no LDS, no memory) on every CU, sustained for seconds, so that
// DVFS settles.  Prints wall-clock stamps per configuration; tools/power_roofline.py samples rocm-smi meanwhile.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_power.hip -o tools/micro/bin/mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/time.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

static double now() { timeval tv; gettimeofday(&tv, nullptr); return tv.tv_sec + 1e-6 * tv.tv_usec; }

// SHAPE 0: 32x32x16, 1: 16x16x32.  Operands rotate through 4 + 4 fragments loaded from memory (random fp16 in [-2, 2), or
// zeros): what the GEMM's inner loop feeds the pipe, minus everything else.
template <int SHAPE>
__global__ __launch_bounds__(512) void k(const half8* __restrict__ frag, const half8* __restrict__ fragb, float* out, int iters) {
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = frag[(i * 64 + (threadIdx.x & 63))]; b[i] = fragb[((4 + i) * 64 + (threadIdx.x & 63))]; }
    floatx16 acc[4];
    floatx4 acc4[8];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            if (SHAPE == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + rep) & 3], b[i], acc[i], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + rep) & 3], b[i & 3], acc4[i], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc4[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Operand reuse between consecutive MFMAs (16x16x32): does the pipe draw less when one (or both) source operands of an instruction
// are the registers the previous instruction read?  ORDER 0: both change every instruction; 1: A held for four instructions;
// 2: A held for all eight of a group (B rotates); 3: both held for four instructions (accumulators differ).
template <int ORDER>
__global__ __launch_bounds__(512) void k_order(const half8* __restrict__ frag, float* out, int iters) {
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = frag[(i * 64 + (threadIdx.x & 63))]; b[i] = frag[((4 + i) * 64 + (threadIdx.x & 63))]; }
    floatx4 acc4[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ia = ORDER == 0 ? (i + rep) & 3 : ORDER == 1 ? ((i >> 2) + rep) & 3 : ORDER == 2 ? rep : ((i >> 2) + rep) & 3;
                const int ib = ORDER == 3 ? ((i >> 2) + 2 * rep) & 3 : i & 3;
                acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ia], b[ib], acc4[i], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc4[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ORDER>
void run_order(const char* name, const half8* frag, float* out, double seconds) {
    const int iters = 20000, threads = 512;
    const double flop_per_launch = 256.0 * (threads / 64) * iters * 32.0 * 16384.0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_order<ORDER><<<256, threads>>>(frag, out, iters);
    hipDeviceSynchronize();
    const double t0 = now();
    int n = 0;
    hipEventRecord(e0, 0);
    while (now() - t0 < seconds) { for (int i = 0; i < 20; ++i) k_order<ORDER><<<256, threads>>>(frag, out, iters); n += 20; hipDeviceSynchronize(); }
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    const double t1 = now();
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    printf("CONFIG %s | waves/SIMD 2 | t0 %.3f t1 %.3f | %.1f TFLOP/s issued (f16 dense)\n", name, t0, t1, n * flop_per_launch / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

// The split-3 GEMM's own inner block (4 x 4 output tiles of 16 x 16, three products each: lo.hi + hi.lo + hi.hi) in several
// instruction orders; operands as in the GEMM (hi planes random, lo planes what the split leaves).
//   PAT 0: the GEMM's order: for mt, for nt: (wl,ah) (wh,al) (wh,ah) on acc[mt][nt]           -- 2 of 3 transitions share an operand
//   PAT 1: snake: every transition shares one operand (the triple is walked forwards / backwards alternately)
//   PAT 2: product-major per m-tile: for nt (wh,ah); for nt (wl,ah); for nt (wh,al)            -- B held for 8, then 4 instructions
//   PAT 3: product-major over the block: all 16 (wh,ah), all 16 (wl,ah), all 16 (wh,al)
template <int PAT>
__global__ __launch_bounds__(512) void k_gemm(const half8* __restrict__ hi, const half8* __restrict__ lo, float* out, int iters) {
    half8 wh[4], wl[4], ah[4], al[4];
    for (int i = 0; i < 4; ++i) {
        wh[i] = hi[i * 64 + (threadIdx.x & 63)]; ah[i] = hi[(4 + i) * 64 + (threadIdx.x & 63)];
        wl[i] = lo[i * 64 + (threadIdx.x & 63)]; al[i] = lo[(4 + i) * 64 + (threadIdx.x & 63)];
    }
    floatx4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
#define MF(A, B, C) C = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, C, 0, 0, 0)
    for (int it = 0; it < iters; ++it) {
        if (PAT == 0) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) { MF(wl[nt], ah[mt], acc[mt][nt]); MF(wh[nt], al[mt], acc[mt][nt]); MF(wh[nt], ah[mt], acc[mt][nt]); }
        } else if (PAT == 1) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    if ((nt & 1) == 0) { MF(wl[nt], ah[mt], acc[mt][nt]); MF(wh[nt], ah[mt], acc[mt][nt]); MF(wh[nt], al[mt], acc[mt][nt]); }
                    else { MF(wh[nt], al[mt], acc[mt][nt]); MF(wh[nt], ah[mt], acc[mt][nt]); MF(wl[nt], ah[mt], acc[mt][nt]); }
                }
        } else if (PAT == 2) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) MF(wh[nt], ah[mt], acc[mt][nt]);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) MF(wl[nt], ah[mt], acc[mt][nt]);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) MF(wh[nt], al[mt], acc[mt][nt]);
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) MF(wh[nt], ah[mt], acc[mt][nt]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) MF(wl[nt], ah[mt], acc[mt][nt]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) MF(wh[nt], al[mt], acc[mt][nt]);
        }
    }
#undef MF
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int PAT>
void run_gemm(const char* name, const half8* hi, const half8* lo, float* out, double seconds) {
    const int iters = 4000, threads = 512;
    const double flop_per_launch = 256.0 * (threads / 64) * iters * 48.0 * 16384.0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_gemm<PAT><<<256, threads>>>(hi, lo, out, iters);
    hipDeviceSynchronize();
    const double t0 = now();
    int n = 0;
    hipEventRecord(e0, 0);
    while (now() - t0 < seconds) { for (int i = 0; i < 20; ++i) k_gemm<PAT><<<256, threads>>>(hi, lo, out, iters); n += 20; hipDeviceSynchronize(); }
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    const double t1 = now();
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    printf("CONFIG %s | waves/SIMD 2 | t0 %.3f t1 %.3f | %.1f TFLOP/s issued (f16 dense)\n", name, t0, t1, n * flop_per_launch / (ms * 1e-3) / 1e12);
    fflush(stdout);
}


// ---- narrow formats (round 5): what does a CORRECTION product of the split GEMM cost on the block-scaled pipe? ----
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef int intx4 __attribute__((ext_vector_type(4)));
// FMT 0: e4m3, 2: e2m3 (fp6), 4: e2m1 (fp4) through v_mfma_scale_f32_16x16x128_f8f6f4 (scales 2^0); 8: int8 16x16x64.
// Random operand bytes (e4m3: NaN codes masked out).  Flops are counted as 2.M.N.K per instruction.
template <int FMT>
__global__ __launch_bounds__(512) void k_fmt(const intx8* __restrict__ frag, float* out, int iters) {
    intx8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = frag[i * 64 + (threadIdx.x & 63)]; b[i] = frag[(4 + i) * 64 + (threadIdx.x & 63)]; }
    floatx4 acc[8]; intx4 acci[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) { acc[i][r] = 0.f; acci[i][r] = 0; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (FMT == 8) {
                    const intx8 x = a[(i + rep) & 3], y = b[i & 3];
                    acci[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8((intx4){x[0], x[1], x[2], x[3]}, (intx4){y[0], y[1], y[2], y[3]}, acci[i], 0, 0, 0);
                } else {
                    acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[(i + rep) & 3], b[i & 3], acc[i], FMT, FMT, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                }
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r] + (float)acci[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int FMT>
void run_fmt(const char* name, int threads, const intx8* frag, float* out, double seconds) {
    const int iters = FMT == 8 ? 20000 : 10000;
    const double flop_per_launch = 256.0 * (threads / 64) * iters * 32.0 * (FMT == 8 ? 2.0 * 16 * 16 * 64 : 2.0 * 16 * 16 * 128);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_fmt<FMT><<<256, threads>>>(frag, out, iters);
    hipDeviceSynchronize();
    const double t0 = now();
    int n = 0;
    hipEventRecord(e0, 0);
    while (now() - t0 < seconds) { for (int i = 0; i < 20; ++i) k_fmt<FMT><<<256, threads>>>(frag, out, iters); n += 20; hipDeviceSynchronize(); }
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    const double t1 = now();
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    printf("CONFIG %s | waves/SIMD %d | t0 %.3f t1 %.3f | %.1f TFLOP/s issued (2MNK of the format)\n", name, threads / 256, t0, t1,
           n * flop_per_launch / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

// The split GEMM's inner block over K = 128 with the two correction products on the block-scaled pipe: 4 x 4 output tiles, per tile
// 4 x (wh, ah) f16 16x16x32 + (wh8, al8) + (wl8, ah8) as two 16x16x128 instructions of format FMT.  FMT -1: the exact mode's own
// block (12 f16 instructions per tile and K = 128).  TFLOP/s printed = ALGORITHMIC (2.M.N.K of the block), comparable across modes.
template <int FMT>
__global__ __launch_bounds__(256) void k_mix(const half8* __restrict__ hi, const half8* __restrict__ lo, const intx8* __restrict__ q,
                                             float* out, int iters) {
    half8 wh[2][4], ah[2][4], wl[2][4], al[2][4];
    intx8 wh8[4], wl8[4], ah8[4], al8[4];
    const int l = threadIdx.x & 63;
    for (int i = 0; i < 4; ++i) {
        for (int s = 0; s < 2; ++s) {
            wh[s][i] = hi[((s * 8 + i) & 7) * 64 + l]; ah[s][i] = hi[((s * 8 + 4 + i + s) & 7) * 64 + l];
            wl[s][i] = lo[((s * 8 + i) & 7) * 64 + l]; al[s][i] = lo[((s * 8 + 4 + i + s) & 7) * 64 + l];
        }
        wh8[i] = q[i * 64 + l]; wl8[i] = q[(4 + i) * 64 + l]; ah8[i] = q[(8 + i) * 64 + l]; al8[i] = q[(12 + i) * 64 + l];
    }
    floatx4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
#define MF(A, B, C) C = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, C, 0, 0, 0)
#define MS(A, B, C) C = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, C, FMT < 0 ? 0 : FMT, FMT < 0 ? 0 : FMT, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    MF(wh[ks & 1][nt], ah[ks & 1][mt], acc[mt][nt]);
                    if (FMT < 0) { MF(wl[ks & 1][nt], ah[ks & 1][mt], acc[mt][nt]); MF(wh[ks & 1][nt], al[ks & 1][mt], acc[mt][nt]); }
                }
                if (FMT >= 0) { MS(wh8[nt], al8[mt], acc[mt][nt]); MS(wl8[nt], ah8[mt], acc[mt][nt]); }
            }
    }
#undef MF
#undef MS
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int FMT>
void run_mix(const char* name, const half8* hi, const half8* lo, const intx8* q, float* out, double seconds) {
    const int iters = 2000, threads = 256;
    const double flop_per_launch = 256.0 * (threads / 64) * iters * 16.0 * (2.0 * 16 * 16 * 128);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_mix<FMT><<<256, threads>>>(hi, lo, q, out, iters);
    hipDeviceSynchronize();
    const double t0 = now();
    int n = 0;
    hipEventRecord(e0, 0);
    while (now() - t0 < seconds) { for (int i = 0; i < 20; ++i) k_mix<FMT><<<256, threads>>>(hi, lo, q, out, iters); n += 20; hipDeviceSynchronize(); }
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    const double t1 = now();
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    printf("CONFIG %s | waves/SIMD 1 | t0 %.3f t1 %.3f | %.1f TFLOP/s ALGORITHMIC (2MNK of the block)\n", name, t0, t1,
           n * flop_per_launch / (ms * 1e-3) / 1e12);
    fflush(stdout);
}


// Option A: 32 x 32 output tiles.  Per tile and K = 32: two 32x32x16 f16 (wh, ah) + ONE 32x32x64 e4m3 whose K = 64 is [al8 | ah8] . [wh8 | wl8]
// (lane group q = lane / 32 holds 32 k-elements: q 0 the lo-plane bytes, q 1 the hi-plane bytes) -- both corrections in one instruction.
// Wave tile 128 x 64 = 4 x 2 tiles, as the 256^2 kernel's waves.  FMT -1: exact on 32x32x16 (6 f16 per tile); -2: exact on 16x16x32 at K = 32.
// Option G (FMT 100): 16 x 16 tiles, per tile and K = 32 one 16x16x32 f16 + one 16x16x128 e4m3 whose lanes 32..63 hold zeros.
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int FMT>
__global__ __launch_bounds__(256) void k_mix32(const half8* __restrict__ hi, const half8* __restrict__ lo, const intx8* __restrict__ q,
                                               float* out, int iters) {
    const int l = threadIdx.x & 63;
    half8 wh[2][2], wl[2][2], ah[4][2], al[4][2];
    intx8 w8[2], a8[4];
    for (int i = 0; i < 4; ++i) for (int s = 0; s < 2; ++s) { ah[i][s] = hi[((i * 2 + s) & 7) * 64 + l]; al[i][s] = lo[((i * 2 + s) & 7) * 64 + l]; }
    for (int i = 0; i < 2; ++i) for (int s = 0; s < 2; ++s) { wh[i][s] = hi[((i * 2 + s + 3) & 7) * 64 + l]; wl[i][s] = lo[((i * 2 + s + 3) & 7) * 64 + l]; }
    for (int i = 0; i < 4; ++i) a8[i] = q[i * 64 + l];
    for (int i = 0; i < 2; ++i) w8[i] = q[(4 + i) * 64 + l];
    floatx16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[nt][s], ah[mt][s], acc[mt][nt], 0, 0, 0);
                    if (FMT < 0) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[nt][s], ah[mt][s], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[nt][s], al[mt][s], acc[mt][nt], 0, 0, 0);
                    }
                }
                if (FMT >= 0) acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w8[nt], a8[mt], acc[mt][nt], FMT, FMT, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int HALF>
__global__ __launch_bounds__(256) void k_mixg(const half8* __restrict__ hi, const half8* __restrict__ lo, const intx8* __restrict__ q,
                                              float* out, int iters) {
    const int l = threadIdx.x & 63;
    half8 wh[4], ah[8], wl[4], al[8];
    intx8 w8[4], a8[8];
    for (int i = 0; i < 8; ++i) { ah[i] = hi[(i & 7) * 64 + l]; al[i] = lo[(i & 7) * 64 + l]; a8[i] = q[i * 64 + l]; if (HALF && l >= 32) a8[i] = intx8{0, 0, 0, 0, 0, 0, 0, 0}; }
    for (int i = 0; i < 4; ++i) { wh[i] = hi[((i + 3) & 7) * 64 + l]; wl[i] = lo[((i + 3) & 7) * 64 + l]; w8[i] = q[(8 + i) * 64 + l]; if (HALF && l >= 32) w8[i] = intx8{0, 0, 0, 0, 0, 0, 0, 0}; }
    floatx4 acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt], ah[mt], acc[mt][nt], 0, 0, 0);
                if (HALF == 2) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[nt], ah[mt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt], al[mt], acc[mt][nt], 0, 0, 0);
                } else {
                    acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w8[nt], a8[mt], acc[mt][nt], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                }
            }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int WHICH, int P>
void run_mix2(const char* name, const half8* hi, const half8* lo, const intx8* q, float* out, double seconds) {
    const int iters = 4000, threads = 256;
    const double flop_per_launch = 256.0 * (threads / 64) * iters * (2.0 * 128 * 64 * 32);      // wave tile 128 x 64, K = 32 per trip
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto go = [&]() { if (WHICH == 0) k_mix32<P><<<256, threads>>>(hi, lo, q, out, iters); else k_mixg<P><<<256, threads>>>(hi, lo, q, out, iters); };
    go();
    hipDeviceSynchronize();
    const double t0 = now();
    int n = 0;
    hipEventRecord(e0, 0);
    while (now() - t0 < seconds) { for (int i = 0; i < 20; ++i) go(); n += 20; hipDeviceSynchronize(); }
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    const double t1 = now();
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    printf("CONFIG %s | waves/SIMD 1 | t0 %.3f t1 %.3f | %.1f TFLOP/s ALGORITHMIC (2MNK of the block)\n", name, t0, t1,
           n * flop_per_launch / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

template <int SHAPE>
void run(const char* name, int threads, const half8* frag, float* out, double seconds, const half8* fragb = nullptr) {
    if (!fragb) fragb = frag;
    const int iters = 20000;                                        // ~ 5-10 ms per launch
    const double flop_per_launch = 256.0 * (threads / 64) * iters * (SHAPE == 0 ? 16.0 * 32768.0 : 32.0 * 16384.0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<SHAPE><<<256, threads>>>(frag, fragb, out, iters);
    hipDeviceSynchronize();
    const double t0 = now();
    int n = 0;
    hipEventRecord(e0, 0);
    while (now() - t0 < seconds) { for (int i = 0; i < 20; ++i) k<SHAPE><<<256, threads>>>(frag, fragb, out, iters); n += 20; hipDeviceSynchronize(); }
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    const double t1 = now();
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    printf("CONFIG %s | waves/SIMD %d | t0 %.3f t1 %.3f | %.1f TFLOP/s issued (f16 dense)\n", name, threads / 256, t0, t1,
           n * flop_per_launch / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    half8* frag; float* out;
    hipMalloc(&out, 256 * 512 * 4);
    static _Float16 h[5][8 * 64 * 8];
    srand(7);
    for (int i = 0; i < 8 * 64 * 8; ++i) {
        const float v = (rand() / (float)RAND_MAX) * 4.f - 2.f;
        h[0][i] = (_Float16)v; h[1][i] = (_Float16)0.f;
        // "lo plane" operands: what the split leaves, v - fp16(v) for v with 22 significant bits (tiny exponent, random mantissa)
        const float w = v * (1.0f + (rand() / (float)RAND_MAX) * 9.7e-4f);
        h[2][i] = (_Float16)(w - (float)(_Float16)w);
        // the same, mantissa cut to its top 5 / 2 explicit bits
        unsigned short b; __builtin_memcpy(&b, &h[2][i], 2);
        unsigned short b5 = b & 0xFFE0, b2 = b & 0xFF00;
        __builtin_memcpy(&h[3][i], &b5, 2); __builtin_memcpy(&h[4][i], &b2, 2);
    }
    hipMalloc(&frag, sizeof(h));
    hipMemcpy(frag, h, sizeof(h), hipMemcpyHostToDevice);
    if (argc > 2 && atoi(argv[2]) == 1) {                            // operand-reuse study only
        run_order<0>("16x16x32 order 0: both operands change every instruction", frag, out, seconds);
        run_order<1>("16x16x32 order 1: A held for 4 instructions", frag, out, seconds);
        run_order<2>("16x16x32 order 2: A held for 8 instructions", frag, out, seconds);
        run_order<3>("16x16x32 order 3: A and B held for 4 instructions", frag, out, seconds);
        run_order<0>("16x16x32 order 0 again", frag, out, seconds);
        return 0;
    }


    if (argc > 2 && atoi(argv[2]) == 4) {                            // tile shapes for the mixed block (round 5)
        static unsigned char qb[16 * 64 * 32];
        for (size_t i = 0; i < sizeof(qb); ++i) { unsigned char v = (unsigned char)(rand() >> 7); if ((v & 0x7f) == 0x7f) v ^= 1; qb[i] = v; }
        intx8* q; hipMalloc(&q, sizeof(qb)); hipMemcpy(q, qb, sizeof(qb), hipMemcpyHostToDevice);
        const half8* hi = frag; const half8* lo = frag + 2 * 8 * 64;
        run_mix2<1, 2>("wave tile 128x64, K=32: exact, 16x16x32 (3 f16 per tile)", hi, lo, q, out, seconds);
        run_mix2<0, -1>("wave tile 128x64, K=32: exact, 32x32x16 (6 f16 per tile)", hi, lo, q, out, seconds);
        run_mix2<0, 0>("wave tile 128x64, K=32: 2 x 32x32x16 f16 + ONE 32x32x64 e4m3 [lo8|hi8]", hi, lo, q, out, seconds);
        run_mix2<0, 2>("wave tile 128x64, K=32: 2 x 32x32x16 f16 + ONE 32x32x64 e2m3 [lo6|hi6]", hi, lo, q, out, seconds);
        run_mix2<1, 1>("wave tile 128x64, K=32: 16x16x32 f16 + 16x16x128 e4m3 with lanes 32-63 zero", hi, lo, q, out, seconds);
        run_mix2<1, 0>("wave tile 128x64, K=32: 16x16x32 f16 + 16x16x128 e4m3 full (= K 64 of corrections; 2x the needed work)", hi, lo, q, out, seconds);
        run_mix2<1, 2>("wave tile 128x64, K=32: exact, 16x16x32 again", hi, lo, q, out, seconds);
        return 0;
    }
    if (argc > 2 && atoi(argv[2]) == 3) {                            // narrow formats for the correction products (round 5)
        static unsigned char qb[16 * 64 * 32];
        for (size_t i = 0; i < sizeof(qb); ++i) { unsigned char v = (unsigned char)(rand() >> 7); if ((v & 0x7f) == 0x7f) v ^= 1; qb[i] = v; }
        intx8* q; hipMalloc(&q, sizeof(qb)); hipMemcpy(q, qb, sizeof(qb), hipMemcpyHostToDevice);
        const half8* hi = frag; const half8* lo = frag + 2 * 8 * 64;
        run<1>("f16 16x16x32 random operands", 256, frag, out, seconds);
        run_fmt<0>("e4m3 16x16x128 scaled, random bytes", 256, q, out, seconds);
        run_fmt<0>("e4m3 16x16x128 scaled, random bytes", 512, q, out, seconds);
        run_fmt<2>("e2m3 (fp6) 16x16x128 scaled, random bits", 256, q, out, seconds);
        run_fmt<4>("e2m1 (fp4) 16x16x128 scaled, random bits", 256, q, out, seconds);
        run_fmt<8>("int8 16x16x64, random bytes", 256, q, out, seconds);
        run_mix<-1>("split block K=128: exact (3 f16 products)", hi, lo, q, out, seconds);
        run_mix<0>("split block K=128: f16 hi.hi + 2 e4m3 corrections", hi, lo, q, out, seconds);
        run_mix<2>("split block K=128: f16 hi.hi + 2 e2m3 corrections", hi, lo, q, out, seconds);
        run_mix<-1>("split block K=128: exact again", hi, lo, q, out, seconds);
        return 0;
    }
    if (argc > 2 && atoi(argv[2]) == 2) {                            // instruction order of the split-3 GEMM block
        const half8* hi = frag; const half8* lo = frag + 2 * 8 * 64;
        run_gemm<0>("gemm block, order of the kernel (lo.hi, hi.lo, hi.hi per tile)", hi, lo, out, seconds);
        run_gemm<1>("gemm block, snake (every transition shares an operand)", hi, lo, out, seconds);
        run_gemm<2>("gemm block, product-major per m-tile", hi, lo, out, seconds);
        run_gemm<3>("gemm block, product-major over the block", hi, lo, out, seconds);
        run_gemm<0>("gemm block, order of the kernel again", hi, lo, out, seconds);
        return 0;
    }
    run<1>("16x16x32 random operands", 256, frag, out, seconds);
    run<1>("16x16x32 random operands", 512, frag, out, seconds);
    run<0>("32x32x16 random operands", 256, frag, out, seconds);
    run<0>("32x32x16 random operands", 512, frag, out, seconds);
    run<1>("16x16x32 zero operands", 512, frag + 8 * 64, out, seconds);
    run<0>("32x32x16 zero operands", 512, frag + 8 * 64, out, seconds);
    // one operand a lo plane (the hi.lo / lo.hi products of the exact mode), the other a hi plane
    run<1>("16x16x32 lo-plane x hi-plane", 512, frag + 2 * 8 * 64, out, seconds, frag);
    run<1>("16x16x32 lo-plane (6 significant bits) x hi-plane", 512, frag + 3 * 8 * 64, out, seconds, frag);
    run<1>("16x16x32 lo-plane (3 significant bits) x hi-plane", 512, frag + 4 * 8 * 64, out, seconds, frag);
    return 0;
}
