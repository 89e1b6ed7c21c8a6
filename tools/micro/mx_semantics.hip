// What exactly do gfx950's fp8 conversion and block-scaled MFMA instructions compute?  Checked once on the device against a host model,
// before the GEMM's `mx` mode (include/cvlm.h, cvlm_gemm_args.mx_*) relies on it:
//   (A) v_cvt_pk_fp8_f32: OCP e4m3fn, round to nearest even, what happens above 448;
//   (B) v_cvt_scalef32_pk_fp8_f32: does the scale multiply or divide;
//   (C) v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands: k-order within a lane (lane (r, q) = row r; bytes 0-15 are k 16q .. 16q + 15,
//       bytes 16-31 are k 64 + 16q ..: two K = 64 halves side by side), the E8M0 scale of k-block b = the byte lane (r, b) supplies, picked
//       by op_sel, value = 2^(E - 127), and the output layout (that of v_mfma_f32_16x16x32_f16);
//   (D) the probe that found (C)'s layout: one lane's half at a time.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mx_semantics.hip -o tools/micro/bin/mx_semantics && tools/micro/bin/mx_semantics
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef short short2_ __attribute__((ext_vector_type(2)));

static float e4m3_decode(unsigned char c) {
    const int s = c >> 7, e = (c >> 3) & 15, m = c & 7;
    float v;
    if (e == 15 && m == 7) return NAN;
    if (e == 0) v = ldexpf((float)m, -9);
    else v = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -v : v;
}
static unsigned char e4m3_encode_rne(float x) {               // nearest representable, ties to even mantissa, saturating at 448
    if (x != x) return 0x7f;
    const unsigned char sgn = x < 0 ? 0x80 : 0;
    float a = fabsf(x);
    if (a >= 464.0f) return sgn | 0x7e;                        // (448 + 480) / 2: beyond it the nearest value would be the missing 480
    unsigned char best = 0; float bd = 1e30f;
    for (int c = 0; c < 0x7f; ++c) {
        const float d = fabsf(e4m3_decode((unsigned char)c) - a);
        if (d < bd || (d == bd && (c & 1) == 0)) { bd = d; best = (unsigned char)c; }
    }
    return sgn | best;
}

__global__ void k_cvt(const float* x, unsigned* o_plain, unsigned* o_scaled, float scale) {
    const float a = x[2 * threadIdx.x], b = x[2 * threadIdx.x + 1];
    o_plain[threadIdx.x] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xffffu;
    short2_ s = {0, 0};
    s = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(s, a, b, scale, false);
    o_scaled[threadIdx.x] = (unsigned)(unsigned short)s[0];
}

template <int SEL>
__global__ void k_mfma(const unsigned char* A, const unsigned char* B, const unsigned char* SA, const unsigned char* SB, float* D) {
    const int l = threadIdx.x, r = l & 15, q = l >> 4;
    intx8 a, b;
    // measured (run (D) below): a lane's first 16 bytes are k = 16q .. 16q + 15, its second 16 bytes k = 64 + 16q .. 64 + 16q + 15;
    // the scale of k-block b (k in [32b, 32b + 32)) is the byte lane (r, b) supplies
    memcpy(&a, A + r * 128 + q * 16, 16); memcpy((char*)&a + 16, A + r * 128 + 64 + q * 16, 16);
    memcpy(&b, B + r * 128 + q * 16, 16); memcpy((char*)&b + 16, B + r * 128 + 64 + q * 16, 16);
    const int sa = (int)SA[r * 4 + q] << (8 * SEL), sb = (int)SB[r * 4 + q] << (8 * SEL);
    floatx4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, SEL, sa, SEL, sb);
    for (int i = 0; i < 4; ++i) D[l * 4 + i] = c[i];
}


// Diagnostic: which (row, k-block) does each operand byte belong to, and whose scale covers it?  Block b: only lane b / 2 of the probed
// operand holds data (sixteen 1.0 codes in half b % 2 of its 32 bytes), the other operand is all 1.0; the probed operand's scale in lane
// (r, q) is 2^q, the other's 1.  D[i][j] = 16 * 2^(q of the scale that covers those bytes) in the row (column) the lane feeds.
template <int SIDE>
__global__ void k_diag(float* D) {
    const int l = threadIdx.x, q = l >> 4, La = blockIdx.x >> 1, h = blockIdx.x & 1;
    intx8 ones, probe = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) ones[i] = 0x38383838;
    if (l == La) for (int i = 0; i < 4; ++i) probe[4 * h + i] = 0x38383838;
    floatx4 c = {0.f, 0.f, 0.f, 0.f};
    if (SIDE == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(probe, ones, c, 0, 0, 0, 127 + q, 0, 127);
    else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones, probe, c, 0, 0, 0, 127, 0, 127 + q);
    for (int i = 0; i < 4; ++i) D[(blockIdx.x * 64 + l) * 4 + i] = c[i];
}

int main() {
    srand(11);
    int bad = 0;
    // ---- (A), (B)
    const int N = 64;
    float hx[2 * N];
    const float special[] = {0.f, 1.f, 1.0625f, 1.1875f, 17.f, 18.f, 19.f, 447.f, 448.f, 456.f, 463.9f, 464.f, 500.f, 1e-3f, 0.001953125f, 0.0009765625f, 0.0029296875f, -3.3f, 240.f, 255.9f};
    for (int i = 0; i < 2 * N; ++i) hx[i] = i < (int)(sizeof(special) / 4) ? special[i] : ((rand() / (float)RAND_MAX) * 2.f - 1.f) * ldexpf(1.f, rand() % 18 - 9);
    float* dx; unsigned *dp, *ds;
    hipMalloc(&dx, sizeof(hx)); hipMalloc(&dp, N * 4); hipMalloc(&ds, N * 4);
    hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
    k_cvt<<<1, N>>>(dx, dp, ds, 4.0f);
    unsigned hp[N], hs[N];
    hipMemcpy(hp, dp, N * 4, hipMemcpyDeviceToHost); hipMemcpy(hs, ds, N * 4, hipMemcpyDeviceToHost);
    int mult = 0, divi = 0;
    for (int i = 0; i < 2 * N; ++i) {
        const unsigned char got = (hp[i / 2] >> (8 * (i & 1))) & 0xff, want = e4m3_encode_rne(hx[i]);
        if (got != want) { if (fabsf(hx[i]) < 464.f) ++bad; printf("(A) x = %g: device 0x%02x (%g), host RNE 0x%02x (%g)%s\n", hx[i], got, e4m3_decode(got), want, e4m3_decode(want), fabsf(hx[i]) >= 464.f ? "   [out of range: informational]" : "   MISMATCH"); }
        const unsigned char gs = (hs[i / 2] >> (8 * (i & 1))) & 0xff;
        if (gs == e4m3_encode_rne(hx[i] * 4.0f)) ++mult;
        if (gs == e4m3_encode_rne(hx[i] / 4.0f)) ++divi;
    }
    printf("(A) v_cvt_pk_fp8_f32 against host e4m3fn round-to-nearest-even: %s\n", bad ? "MISMATCHES above" : "all equal");
    printf("(B) v_cvt_scalef32_pk_fp8_f32 with scale 4: equals fp8(x * 4) on %d of %d, fp8(x / 4) on %d of %d\n", mult, 2 * N, divi, 2 * N);
    // ---- (C)
    unsigned char hA[16 * 128], hB[16 * 128], hSA[64], hSB[64];
    for (int i = 0; i < 16 * 128; ++i) {
        hA[i] = (unsigned char)(rand() >> 5); if ((hA[i] & 0x7f) == 0x7f) hA[i] ^= 1;
        hB[i] = (unsigned char)(rand() >> 5); if ((hB[i] & 0x7f) == 0x7f) hB[i] ^= 1;
    }
    for (int i = 0; i < 64; ++i) { hSA[i] = (unsigned char)(120 + rand() % 15); hSB[i] = (unsigned char)(110 + rand() % 20); }
    unsigned char *dA, *dB, *dSA, *dSB; float* dD;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dSA, 64); hipMalloc(&dSB, 64); hipMalloc(&dD, 256 * 4);
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipMemcpy(dSA, hSA, 64, hipMemcpyHostToDevice); hipMemcpy(dSB, hSB, 64, hipMemcpyHostToDevice);
    for (int sel = 0; sel < 4; ++sel) {
        if (sel == 0) k_mfma<0><<<1, 64>>>(dA, dB, dSA, dSB, dD);
        else if (sel == 1) k_mfma<1><<<1, 64>>>(dA, dB, dSA, dSB, dD);
        else if (sel == 2) k_mfma<2><<<1, 64>>>(dA, dB, dSA, dSB, dD);
        else k_mfma<3><<<1, 64>>>(dA, dB, dSA, dSB, dD);
        float hD[256];
        hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
        double worst = 0.0, ref_max = 0.0;
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 4; ++i) {
                const int ia = 4 * (l >> 4) + i, ib = l & 15;          // layout of v_mfma_f32_16x16x32_f16: A row 4 * (lane / 16) + reg, B row lane % 16
                double ref = 0.0;
                for (int k = 0; k < 128; ++k)
                    ref += (double)e4m3_decode(hA[ia * 128 + k]) * ldexp(1.0, hSA[ia * 4 + k / 32] - 127) *
                           (double)e4m3_decode(hB[ib * 128 + k]) * ldexp(1.0, hSB[ib * 4 + k / 32] - 127);
                worst = fmax(worst, fabs(ref - hD[l * 4 + i])); ref_max = fmax(ref_max, fabs(ref));
            }
        const bool ok = worst <= 1e-5 * ref_max;
        if (!ok) ++bad;
        printf("(C) 16x16x128 e4m3, op_sel byte %d: max |device - host| = %.3e of max |ref| %.3e  %s\n", sel, worst, ref_max, ok ? "ok" : "MISMATCH");
    }

    {   // ---- diagnostic dump
        float* dG; hipMalloc(&dG, 128 * 256 * 4);
        static float hG[128 * 256];
        for (int side = 0; side < 2; ++side) {
            if (side == 0) k_diag<0><<<128, 64>>>(dG); else k_diag<1><<<128, 64>>>(dG);
            hipMemcpy(hG, dG, sizeof(hG), hipMemcpyDeviceToHost);
            printf("(D) probe operand %s: lane/half -> [out lanes with non-zero: reg mask, value]\n", side == 0 ? "A (first)" : "B (second)");
            for (int b = 0; b < 128; ++b) {
                int nz = 0, first = -1, mask = 0; float val = 0.f;
                for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) if (hG[(b * 64 + l) * 4 + i] != 0.f) { if (first < 0) { first = l; val = hG[(b * 64 + l) * 4 + i]; } if (l == first) mask |= 1 << i; ++nz; }
                if (b < 16 || (b % 32) < 2) printf("    lane %2d half %d: %3d non-zero outputs, first in out-lane %2d reg mask %x value %g\n", b >> 1, b & 1, nz, first, mask, val);
            }
        }
    }
    printf(bad ? "FAILED\n" : "PASSED\n");
    return bad ? 1 : 0;
}
