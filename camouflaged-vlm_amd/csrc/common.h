// Shared device helpers for the gfx950 (CDNA4 / MI355X) kernels of the camouflaged-vlm hot path.
//
// Number formats used between kernels
//   f32  : plain float tensors (residual streams, final outputs)
//   h2   : "split half" -- a value v is carried as two fp16 planes  hi = fp16(v), lo = fp16(v - hi)
//          (~22 significand bits).  MFMA products are formed as hi*hi + lo*hi + hi*lo with fp32
//          accumulation, which keeps the 1e-3 parity budget of the fp32 reference while running on
//          the fp16 matrix pipe (gfx950 has no xf32/TF32, and fp32 MFMA is 1/16 of the fp16 rate).
//          split == 1 ("fast") uses the hi plane only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef int intx4 __attribute__((ext_vector_type(4)));
typedef int intx8 __attribute__((ext_vector_type(8)));

#define CVLM_WAVE 64

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// 16-byte asynchronous global -> LDS copy (LDS destination = wave-uniform base + lane*16).
__device__ __forceinline__ void glds16(const void* gptr, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gptr, (LDS_AS void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void split_h2(float v, half_t& hi, half_t& lo) {
    // Pin v in a register first: with fp contraction hipcc otherwise re-derives `hi` for the `lo`
    // computation with v_fma_mixlo_f16 (product rounded once) while the stored hi comes from the
    // rounded product -- on near-ties the two differ by one fp16 ulp and hi + lo is off by that ulp.
    asm volatile("" : "+v"(v));
    hi = (half_t)v;
    lo = (half_t)(v - (float)hi);
}

// lo plane of two values whose hi plane is the fp16 pair packed in `h` (whatever rounding produced it): fp16(v - hi), ONE instruction
// per value -- v_fma_mix reads the fp16 half of `h` and the f32 value in the same fused multiply-add, the f16 result lands in the
// half of the destination it belongs to.  The softmax phases of the attention kernels split 32 probabilities per lane per 64 keys
// this way; written as cvt + sub + pack it took 5-6 instructions per pair, 30 % of the phase's VALU work (global attention: 271 -> 223
// VALU instructions per 64-key phase, 2.40 -> 2.37 ms per launch; lo is now rounded to nearest instead of truncated: cascade mask
// error 5.0e-5 -> 3.9e-5).  The window kernels keep the cvt form: both this and the packed-convert form measured 1-3 % slower
// there (256 VGPRs, one basic block scheduled around the MFMAs: profiles/r03_attn_split_lo_ab.log).
__device__ __forceinline__ unsigned split_lo_pk(unsigned h, float vx, float vy) {
    unsigned lo;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(h), "v"(vx));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(h), "v"(vy));
    return lo;
}

// split_h2 of two values at once, three instructions instead of eight: hi = v_cvt_pk_f16_f32 (round to nearest, both values), lo =
// fp16(v - hi) by split_lo_pk -- the bits of split_h2 (the difference is exact in f32 either way and is rounded once).  Vector
// instructions are not hidden behind MFMAs on this part (profiles/r04_mfma_valu_overlap.log): every GEMM tile's epilogue splits 128
// values per lane.  hi / lo: the two fp16 values packed as they are stored ([31:16] = second value).
__device__ __forceinline__ void split_h2_pk(float vx, float vy, unsigned& hi, unsigned& lo) {
    asm volatile("" : "+v"(vx), "+v"(vy));                               // pin the rounded f32 values first: see split_h2
    typedef float f32x2_cvt __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_cvt{vx, vy}, half2v));
    lo = split_lo_pk(hi, vx, vy);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Exchange between the two 32-lane halves of a wave without the LDS crossbar (gfx950 v_permlane32_swap): after the
// swap one register holds the lower half's value in both halves and the other the upper half's.
__device__ __forceinline__ float half_swap_max(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
__device__ __forceinline__ float half_swap_sum(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// activation codes shared by the GEMM epilogue and the row kernels
enum { ACT_NONE = 0, ACT_GELU = 1, ACT_QUICKGELU = 2, ACT_RELU = 3, ACT_ABS_POST = 4 };

// erf, branch-free (Abramowitz & Stegun 7.1.26, |abs err| <= 1.5e-7 -- fp32-grade for the exact-erf GELU
// of common.py:13-26; the library erff is ~4x the instructions and divergent, and cost 25 % of the
// lin1 GEMM when it ran in the epilogue).
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    float p = 1.061405429f;
    p = p * t - 1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t - 0.284496736f;
    p = p * t + 0.254829592f;
    const float r = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(r, x);
}

// The same GELU on two values at once: every step but the reciprocal and the exponential is a packed-f32 instruction
// (v_pk_mul / v_pk_fma: two values per issue slot), which is what the lin1 epilogue of the ViT blocks spends its time on.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_erf2(f32x2_t v) {
    const f32x2_t x = v * 0.70710678118654752440f;
    const f32x2_t ax = __builtin_elementwise_abs(x);
    const f32x2_t d = ax * 0.3275911f + 1.0f;
    const f32x2_t t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    f32x2_t p = t * 1.061405429f - 1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t - 0.284496736f;
    p = p * t + 0.254829592f;
    const f32x2_t a = ax * ax * -1.4426950408889634f;                 // exp(-x^2) = exp2(-x^2 log2 e)
    const f32x2_t e = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
    const f32x2_t r = 1.0f - p * t * e;
    const f32x2_t erf = {copysignf(r.x, x.x), copysignf(r.y, x.y)};
    return 0.5f * v * (1.0f + erf);
}

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case ACT_GELU: return 0.5f * v * (1.0f + erf_as(v * 0.70710678118654752440f)); // exact-erf GELU
        case ACT_QUICKGELU: return v * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * v)); // x*sigmoid(1.702x)
        case ACT_RELU: return fmaxf(v, 0.0f);
        default: return v;
    }
}

// Element offset of the head slice of operand op (0 = q, 1 = k, 2 = v) of token t of image b:
//   off = b*sb + t*st + op*sop + head*sh   (branch-free; the four strides are wave-uniform scalars)
// layout 0: token-major  [B*S][3][H][hd]   (what a plain qkv GEMM writes)
// layout 1: head-major   [3][B][H][S][hd]  (each (b, head) K / V matrix contiguous: full-line streaming)
struct QkvStrides { int64_t sb, st, sop, sh; };
__device__ __forceinline__ QkvStrides qkv_strides(int layout, int S, int B, int H, int HD) {
    QkvStrides q;
    const int64_t D = (int64_t)H * HD;
    if (layout == 0) { q.sb = (int64_t)S * 3 * D; q.st = 3 * D; q.sop = D; q.sh = HD; }
    else { q.sb = (int64_t)H * S * HD; q.st = HD; q.sop = (int64_t)B * H * S * HD; q.sh = (int64_t)S * HD; }
    return q;
}
__device__ __forceinline__ int64_t qkv_offset(const QkvStrides& q, int b, int t, int op, int head) {
    return b * q.sb + t * q.st + op * q.sop + head * q.sh;
}

// hipFuncSetAttribute (dynamic LDS above 48 KiB) is a per-device setting: one flag per (kernel instantiation, device).
// Usage:  static bool once[16] = {};  if (cvlm_first_on_device(once)) hipFuncSetAttribute(...);
static inline bool cvlm_first_on_device(bool (&done)[16]) {
    int d = 0;
    (void)hipGetDevice(&d);
    d &= 15;
    const bool first = !done[d];
    done[d] = true;
    return first;
}

#define CVLM_CHECK_LAUNCH() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)
