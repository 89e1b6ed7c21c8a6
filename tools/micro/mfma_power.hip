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
