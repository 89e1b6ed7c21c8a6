// Micro-benchmark: issue rate of dependent vs independent MFMA accumulation chains on gfx950 (one or two waves per SIMD).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_chain.hip -o /tmp/mfma_chain && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int NACC, int SHAPE>   // SHAPE 0: 32x32x16, 1: 16x16x32
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters) {
    half8 a = half8{1, 2, 3, 4, 5, 6, 7, 8} * (_Float16)(0.001f * (threadIdx.x & 63));
    half8 b = half8{8, 7, 6, 5, 4, 3, 2, 1} * (_Float16)(0.002f * (threadIdx.x & 63));
    floatx16 acc[NACC];
    floatx4 acc4[NACC];
    for (int i = 0; i < NACC; ++i) { for (int r = 0; r < 16; ++r) acc[i][r] = 0.f; for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f; }
    __syncthreads();
    const unsigned long long w0 = wall_clock64();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 24 / NACC; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (SHAPE == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
                else acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc4[i], 0, 0, 0);
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) { for (int r = 0; r < 16; ++r) s += acc[i][r]; for (int r = 0; r < 4; ++r) s += acc4[i][r]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}

template <int NACC, int SHAPE>
void run(const char* name, int threads) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 16);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, SHAPE><<<256, threads>>>(out, cyc, iters);
    hipEventRecord(e0, 0);
    k<NACC, SHAPE><<<256, threads>>>(out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2]; hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    const double us = c[1] / 100.0, n_mfma = 256.0 * (threads / 64) * iters * 24.0, flop = SHAPE == 0 ? 32768.0 : 16384.0;
    printf("%-28s waves/SIMD %d: %6.1f s_memtime ticks per MFMA per wave; %7.1f us wall; tick rate %.2f GHz; grid %7.1f us by events = %6.0f TFLOP/s chip\n", name,
           threads / 256, (double)c[0] / (iters * 24.0), us, c[0] / us / 1e3, ms * 1e3, n_mfma * flop / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int threads : {256, 512}) {
        if (threads == 256) {
            run<1, 0>("32x32x16 1 accumulator", 256); run<2, 0>("32x32x16 2 accumulators", 256); run<3, 0>("32x32x16 3 accumulators", 256);
            run<1, 1>("16x16x32 1 accumulator", 256); run<2, 1>("16x16x32 2 accumulators", 256); run<4, 1>("16x16x32 4 accumulators", 256);
        } else {
            run<1, 0>("32x32x16 1 accumulator", 512); run<3, 0>("32x32x16 3 accumulators", 512);
            run<1, 1>("16x16x32 1 accumulator", 512); run<4, 1>("16x16x32 4 accumulators", 512);
            run<3, 0>("32x32x16 3 accumulators", 1024); run<4, 1>("16x16x32 4 accumulators", 1024);
        }
    }
    return 0;
}
