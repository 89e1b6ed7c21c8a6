// Do the matrix pipe and the vector ALU of a SIMD run at the same time -- and what does it buy at the socket power cap?
// One workgroup of 8 waves per CU (waves w and w + 4 share a SIMD), every wave runs the same number of loop trips of
//   MODE 0: 8 independent 16x16x32 f16 MFMAs per trip (128 matrix-pipe cycles), every wave
//   MODE 1: 32 independent v_fma_f32 per trip (128 VALU cycles), every wave
//   MODE 2: waves 0..3 the MFMA trips, waves 4..7 the VALU trips (one of each kind per SIMD)
//   MODE 3: every wave both, interleaved in one loop body (8 MFMAs + 32 FMAs per trip)
//   MODE 4: waves 0..3 only, both interleaved (one wave per SIMD)
//   MODE 5 / 6 / 7 (16 waves per CU, four per SIMD): every wave both / waves 0..7 MFMA and 8..15 VALU (two of each kind per SIMD) / MFMA only
// If the two pipes overlap, MODE 2 takes max(MODE 0, MODE 1) / 2 per unit of work and MODE 3 takes max, not the sum.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_valu_overlap.hip -o tools/micro/bin/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/time.h>
static double now() { timeval tv; gettimeofday(&tv, nullptr); return tv.tv_sec + 1e-6 * tv.tv_usec; }
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(MODE >= 5 ? 1024 : 512) void k(const half8* __restrict__ frag, float* out, int iters, float fa, float fb) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = frag[i * 64 + lane]; b[i] = frag[(4 + i) * 64 + lane]; }
    floatx4 acc[8];
    float x[32];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int i = 0; i < 32; ++i) x[i] = (float)(lane + i);
    const bool do_m = MODE == 0 || MODE == 3 || (MODE == 2 && wave < 4) || (MODE == 4 && wave < 4) || MODE == 5 || (MODE == 6 && wave < 8) || MODE == 7;
    const bool do_v = MODE == 1 || MODE == 3 || (MODE == 2 && wave >= 4) || (MODE == 4 && wave < 4) || MODE == 5 || (MODE == 6 && wave >= 8);
    if (do_m && do_v) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i + 1) & 3], acc[i], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) x[4 * i + j] = __builtin_fmaf(x[4 * i + j], fa, fb);
            }
        }
    } else if (do_m) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i + 1) & 3], acc[i], 0, 0, 0);
        }
    } else if (do_v) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) x[i] = __builtin_fmaf(x[i], fa, fb);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    for (int i = 0; i < 32; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 1.0;
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    half8* frag; float* out;
    hipMalloc(&frag, 8 * 64 * sizeof(half8));
    hipMalloc(&out, (size_t)cus * 1024 * sizeof(float));
    { _Float16 h[8 * 64 * 8]; srand(1); for (auto& v : h) v = (_Float16)((rand() % 4096) / 1024.0f - 2.0f); hipMemcpy(frag, h, sizeof(h), hipMemcpyHostToDevice); }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[8] = {"MFMA only, 2 waves / SIMD", "VALU only, 2 waves / SIMD", "one MFMA wave + one VALU wave per SIMD", "both in every wave, 2 waves / SIMD",
                            "both in one wave per SIMD", "both in every wave, 4 waves / SIMD", "two MFMA waves + two VALU waves per SIMD", "MFMA only, 4 waves / SIMD"};
    for (int mode = 0; mode < 8; ++mode) {
        const int iters = 200000;
        auto launch = [&]() {
            switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(cus), dim3(512), 0, 0, frag, out, iters, 0.999f, 0.001f); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(cus), dim3(512), 0, 0, frag, out, iters, 0.999f, 0.001f); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(cus), dim3(512), 0, 0, frag, out, iters, 0.999f, 0.001f); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(cus), dim3(512), 0, 0, frag, out, iters, 0.999f, 0.001f); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(cus), dim3(512), 0, 0, frag, out, iters, 0.999f, 0.001f); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(cus), dim3(1024), 0, 0, frag, out, iters, 0.999f, 0.001f); break;
                case 6: hipLaunchKernelGGL(k<6>, dim3(cus), dim3(1024), 0, 0, frag, out, iters, 0.999f, 0.001f); break;
                default: hipLaunchKernelGGL(k<7>, dim3(cus), dim3(1024), 0, 0, frag, out, iters, 0.999f, 0.001f); break;
            }
        };
        launch(); hipDeviceSynchronize();
        float ms1 = 0.f;
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms1, e0, e1);
        int n = (int)(secs * 1e3 / ms1) + 1;
        const double w0 = now();
        hipEventRecord(e0);
        for (int i = 0; i < n; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
        ms /= n;
        // cycles per trip at the sustained rate: 128 pipe cycles of each kind per trip
        printf("%-46s %8.3f ms per launch of %d trips = %6.1f ns per trip | t0 %.3f t1 %.3f\n", names[mode], ms, iters, 1e6 * ms / iters, w0, now());
        fflush(stdout);
    }
    return 0;
}
