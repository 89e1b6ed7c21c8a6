// The gemm_nt_kernel variants whose operands are mx images (include/cvlm.h, ABI 10: fp16 hi.hi product + the two correction products
// on the block-scaled e4m3 matrix instruction) and their launcher; called from gemm.hip (cvlm_gemm), which has validated the
// arguments and chosen the K-parts of a partial last round.
#include "gemm_kernel.h"
using namespace cvlm_gemm_k;

template <int MT_, int EPI_, int DBG_>
static int launch_one(GemmParams& p, int extra_blocks, hipStream_t s) {
    constexpr int smem_ = 5 * 32768;                                      // the ring: five half-slots (gemm_kernel.h, MX branch)
    p.nbx = (p.a.N + 255) / 256;
    p.nby = (p.a.M + MT_ * 32 - 1) / (MT_ * 32);
    auto kern_ = gemm_nt_kernel<3, 2, 4, 5, 32, DBG_, MT_, false, EPI_, false, false, true, true, true>;
    static bool attr_[16] = {};
    if (cvlm_first_on_device(attr_))
        (void)hipFuncSetAttribute((const void*)kern_, hipFuncAttributeMaxDynamicSharedMemorySize, smem_);
    hipLaunchKernelGGL(kern_, dim3(p.nbx * p.nby + extra_blocks, 1), dim3(512), smem_, s, p);
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_gemm_k::launch_mx(GemmParams& p, int mt, int epi, int extra_blocks, int probe, hipStream_t s) {
#ifdef CVLM_PROBES   /* CVLM_GEMM_VARIANT = 100 + DBG: the probe forms of the fold-epilogue kernel (gemm_kernel.h, MX branch; tools/probe_gemm_mx.py) */
    if (mt == 8 && epi == 1) {
        switch (probe) {
            case 1: return launch_one<8, 1, 1>(p, extra_blocks, s);
            case 2: return launch_one<8, 1, 2>(p, extra_blocks, s);
            case 3: return launch_one<8, 1, 3>(p, extra_blocks, s);
            case 5: return launch_one<8, 1, 5>(p, extra_blocks, s);
            case 6: return launch_one<8, 1, 6>(p, extra_blocks, s);
            case 7: return launch_one<8, 1, 7>(p, extra_blocks, s);
            case 9: return launch_one<8, 1, 9>(p, extra_blocks, s);
            case 10: return launch_one<8, 1, 10>(p, extra_blocks, s);
            case 11: return launch_one<8, 1, 11>(p, extra_blocks, s);
            case 12: return launch_one<8, 1, 12>(p, extra_blocks, s);
            case 13: return launch_one<8, 1, 13>(p, extra_blocks, s);
            case 14: return launch_one<8, 1, 14>(p, extra_blocks, s);
            case 15: return launch_one<8, 1, 15>(p, extra_blocks, s);
            case 16: return launch_one<8, 1, 16>(p, extra_blocks, s);
            case 17: return launch_one<8, 1, 17>(p, extra_blocks, s);
            case 18: return launch_one<8, 1, 18>(p, extra_blocks, s);
            case 19: return launch_one<8, 1, 19>(p, extra_blocks, s);
            case 20: return launch_one<8, 1, 20>(p, extra_blocks, s);
            case 21: return launch_one<8, 1, 21>(p, extra_blocks, s);
            case 22: return launch_one<8, 1, 22>(p, extra_blocks, s);
            default: break;
        }
    }
    if (mt == 8 && epi == 2) {                                            /* the h2-residual form (lin2): a few of the same probes */
        switch (probe) {
            case 1: return launch_one<8, 2, 1>(p, extra_blocks, s);
            case 2: return launch_one<8, 2, 2>(p, extra_blocks, s);
            case 3: return launch_one<8, 2, 3>(p, extra_blocks, s);
            case 6: return launch_one<8, 2, 6>(p, extra_blocks, s);
            case 16: return launch_one<8, 2, 16>(p, extra_blocks, s);
            case 17: return launch_one<8, 2, 17>(p, extra_blocks, s);
            case 18: return launch_one<8, 2, 18>(p, extra_blocks, s);
            case 19: return launch_one<8, 2, 19>(p, extra_blocks, s);
            case 20: return launch_one<8, 2, 20>(p, extra_blocks, s);
            case 21: return launch_one<8, 2, 21>(p, extra_blocks, s);
            case 22: return launch_one<8, 2, 22>(p, extra_blocks, s);
            default: break;
        }
    }
#endif
    (void)probe;
    if (mt == 6 && epi == 2) return launch_one<6, 2, 0>(p, extra_blocks, s);
    if (mt == 8 && epi == 1) return launch_one<8, 1, 0>(p, extra_blocks, s);
    if (mt == 8 && epi == 2) return launch_one<8, 2, 0>(p, extra_blocks, s);
    return CVLM_E_UNSUPPORTED;                                            /* the plain epilogue has no mx instantiation (no caller) */
}
