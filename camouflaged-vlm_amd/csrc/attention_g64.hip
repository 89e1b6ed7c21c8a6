// ViT-H *global* attention specialised for the 64x64 token map of the 1024^2 SAM encoder
// (image_encoder.py:488-504 + 589-625, blocks 7/15/23/31): S = 4096, head_dim = 80, decomposed
// rel-pos bias.  Same orientation as attention.hip (query on the lane for S^T and O^T), plus:
//   * key tiles of 32 slots = half an image row, so for a whole tile kh is constant and kw is a
//     fixed function of the accumulator register: bias = Th[q][kh] (one LDS word per row pair)
//     + Tw[q][kw] held in 32 VGPRs for the whole kernel -- no per-element index math or LDS gathers;
//   * separate LDS pitches for K (176 B, conflict-free ds_read_b128) and V (192 B, conflict-free
//     ds_read_b64_tr_b16: four consecutive rows land on disjoint bank quarters);
//   * ~58 KB of LDS and <= 256 VGPRs so two workgroups (8 waves) share a CU.
#include "common.h"
#include "../../include/cvlm.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ half4 lds_read_tr16(const half_t* p) {
    typedef __fp16 fp16x4 __attribute__((ext_vector_type(4)));
    fp16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LDS_AS fp16x4*)p);
    return __builtin_bit_cast(half4, r);
}

template <int SQK, int SPV>
__global__ __launch_bounds__(256, 2) void attn_g64_kernel(const cvlm_attn_args g) {
    constexpr int HD = 80, KS = 5, ND = 3, CPR = 10, KP = 88, VP = 96, KT = 32, NT = 256, L = 64, LTP = 65;
    constexpr int NPL = (SQK == 3 || SPV == 3) ? 2 : 1;
    constexpr int KPLANE = KT * KP, VPLANE = KT * VP;
    constexpr int UNITS = 2 * NPL * KT * CPR;
    constexpr int UPT = (UNITS + NT - 1) / NT;
    constexpr int S = L * L;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half_t* Ks = (half_t*)smem;                              // [NPL][KT][KP]
    half_t* Vs = Ks + NPL * KPLANE;                          // [NPL][KT][VP]
    float* T = (float*)(Vs + NPL * VPLANE + 64);             // [128][LTP]  (Tw first, then Th)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qc = lane & 31, half = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int D = g.heads * HD;
    const half_t* qkv_hi = (const half_t*)g.qkv_hi;
    const half_t* qkv_lo = (const half_t*)g.qkv_lo;
    const QkvStrides QS = qkv_strides(g.qkv_layout, S, g.B, g.heads, HD);

    const int qslot = blockIdx.x * 128 + wave * 32 + qc;     // 4096 % 128 == 0: every query is valid
    half8 qh[KS], ql[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int64_t qo = qkv_offset(QS, b, qslot, 0, head);
        qh[ks] = *(const half8*)(qkv_hi + qo + 16 * ks + 8 * half);
        if (SQK == 3) ql[ks] = *(const half8*)(qkv_lo + qo + 16 * ks + 8 * half);
    }
    const int qhh = qslot >> 6, qww = qslot & 63;
    float* Tq = T + (wave * 32 + qc) * LTP;

    // U = Q . R^T over the 127 table rows (MFMA), scattered to T[q][k] = U[q][c - k + 63]
    auto build_table = [&](const half_t* Rhi, const half_t* Rlo, int cq) {
#pragma unroll 1
        for (int st = 0; st < 4; ++st) {
            floatx16 u;
#pragma unroll
            for (int r = 0; r < 16; ++r) u[r] = 0.f;
            int rr = st * 32 + qc;
            rr = rr < 127 ? rr : 126;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const half8 ah = *(const half8*)(Rhi + rr * HD + 16 * ks + 8 * half);
                u = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[ks], u, 0, 0, 0);
                if (SQK == 3) {
                    const half8 al = *(const half8*)(Rlo + rr * HD + 16 * ks + 8 * half);
                    u = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[ks], u, 0, 0, 0);
                    u = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[ks], u, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = st * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int kidx = cq + 63 - j;
                if (j < 127 && kidx >= 0 && kidx < L) Tq[kidx] = u[r];
            }
        }
    };

    // Tw -> 32 registers: twr[pz][r] = Tw[q][32*pz + (r&3) + 8*(r>>2) + 4*half]
    float twr[2][16];
    build_table((const half_t*)g.relw_hi, (const half_t*)g.relw_lo, qww);
    __syncthreads();
#pragma unroll
    for (int pz = 0; pz < 2; ++pz)
#pragma unroll
        for (int r = 0; r < 16; ++r) twr[pz][r] = Tq[32 * pz + (r & 3) + 8 * (r >> 2) + 4 * half];
    __syncthreads();
    build_table((const half_t*)g.relh_hi, (const half_t*)g.relh_lo, qhh);

    // ---------------- K/V staging (global -> registers -> LDS), one 32-slot tile at a time
    half8 stage[UPT];
    auto prefetch = [&](int t) {
#pragma unroll
        for (int i = 0; i < UPT; ++i) {
            const int u = tid + i * NT;
            if (UNITS % NT == 0 || u < UNITS) {
                const int chunk = u % CPR;
                const int row = (u / CPR) % KT;
                const int po = u / (CPR * KT);
                const int op = po / NPL, pl = po - op * NPL;
                const half_t* base = (pl ? qkv_lo : qkv_hi) + qkv_offset(QS, b, t * KT + row, op + 1, head);
                stage[i] = *(const half8*)(base + chunk * 8);
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < UPT; ++i) {
            const int u = tid + i * NT;
            if (UNITS % NT == 0 || u < UNITS) {
                const int chunk = u % CPR;
                const int row = (u / CPR) % KT;
                const int po = u / (CPR * KT);
                const int op = po / NPL, pl = po - op * NPL;
                half_t* dst = op ? Vs + pl * VPLANE + row * VP : Ks + pl * KPLANE + row * KP;
                *(half8*)(dst + chunk * 8) = stage[i];
            }
        }
    };

    float m_run = -INFINITY, l_run = 0.f;
    floatx16 o[ND];
#pragma unroll
    for (int n = 0; n < ND; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[n][r] = 0.f;
    const float scale = g.scale;
    const int tg = lane >> 4, ti = lane & 15;
    const int v_lane_off = (4 * (tg >> 1) + (ti >> 2)) * VP + 16 * (tg & 1) + 4 * (ti & 3);
    const half_t* kr = Ks + qc * KP + 8 * half;

    auto tile = [&](const float (&tw)[16], float th) {
        floatx16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const half8 kh = *(const half8*)(kr + 16 * ks);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], s, 0, 0, 0);
            if (SQK == 3) {
                const half8 kl = *(const half8*)(kr + KPLANE + 16 * ks);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], s, 0, 0, 0);
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = s[r] * scale + (th + tw[r]);
            mx = fmaxf(mx, s[r]);
        }
        mx = half_swap_max(mx);
        const float m_new = fmaxf(m_run, mx);
        const float alpha = exp2f((m_run - m_new) * LOG2E);
        const float mneg = m_new * LOG2E;
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = exp2f(s[r] * LOG2E - mneg);
            s[r] = e;
            ps += e;
        }
        l_run = l_run * alpha + ps;
        if (!__all(m_new == m_run)) {
#pragma unroll
            for (int n = 0; n < ND; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[n][r] *= alpha;
        }
        m_run = m_new;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            half8 ph, pl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                half_t eh, el;
                split_h2(s[8 * k2 + j], eh, el);
                ph[j] = eh;
                if (SPV == 3) pl[j] = el;
            }
            const half_t* vb = Vs + (16 * k2) * VP + v_lane_off;
#pragma unroll
            for (int n = 0; n < ND; ++n) {
                const half4 v0 = lds_read_tr16(vb + 32 * n);
                const half4 v1 = lds_read_tr16(vb + 32 * n + 8 * VP);
                const half8 vh = half8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                o[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o[n], 0, 0, 0);
                if (SPV == 3) {
                    const half4 w0 = lds_read_tr16(vb + VPLANE + 32 * n);
                    const half4 w1 = lds_read_tr16(vb + VPLANE + 32 * n + 8 * VP);
                    const half8 vl = half8{w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
                    o[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o[n], 0, 0, 0);
                    o[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o[n], 0, 0, 0);
                }
            }
        }
    };

    prefetch(0);
#pragma unroll 1
    for (int t2 = 0; t2 < L; ++t2) {                          // key row kh = t2: two 32-slot tiles
        commit();
        __syncthreads();
        prefetch(2 * t2 + 1);
        const float th = Tq[t2];
        tile(twr[0], th);
        __syncthreads();
        commit();
        __syncthreads();
        if (t2 + 1 < L) prefetch(2 * t2 + 2);
        tile(twr[1], th);
        __syncthreads();
    }

    const float l_tot = half_swap_sum(l_run);
    const float inv = 1.0f / l_tot;
    const int64_t orow = ((int64_t)b * S + qslot) * D + head * HD;
    half_t* oh = (half_t*)g.out_hi + orow;
    half_t* ol = g.out_lo ? (half_t*)g.out_lo + orow : nullptr;
#pragma unroll
    for (int n = 0; n < ND; ++n)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int d = 32 * n + 8 * rg + 4 * half;
            if (d < HD) {
                half_t h[4], l4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) split_h2(o[n][4 * rg + j] * inv, h[j], l4[j]);
                *(half4*)(oh + d) = half4{h[0], h[1], h[2], h[3]};
                if (ol) *(half4*)(ol + d) = half4{l4[0], l4[1], l4[2], l4[3]};
            }
        }
}

template <int SQK, int SPV>
int launch_g64(const cvlm_attn_args& g, hipStream_t s) {
    constexpr int NPL = (SQK == 3 || SPV == 3) ? 2 : 1;
    constexpr int smem = NPL * 32 * (88 + 96) * 2 + 128 + 128 * 65 * 4;
    auto kern = attn_g64_kernel<SQK, SPV>;
    static bool attr[16] = {};
    if (smem > 48 * 1024 && cvlm_first_on_device(attr))
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(kern, dim3(4096 / 128, g.heads, g.B), dim3(256), smem, s, g);
    CVLM_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// called from cvlm_attention() for mode 1, grid 64, head_dim 80
int cvlm_attention_global64_pp(const cvlm_attn_args& g, hipStream_t s);   // attention_g64pp.hip

int cvlm_attention_global64(const cvlm_attn_args& g, hipStream_t s) {
    if (g.split_qk == 3 && g.split_pv == 3) {
        const int rc = cvlm_attention_global64_pp(g, s);
        if (rc != CVLM_E_UNSUPPORTED) return rc;                     // incl. CVLM_E_WORKSPACE: a missing workspace is an error, not a silent fallback
    }
    if (g.grid != 64) return CVLM_E_UNSUPPORTED;                     // the single-group kernel below is 64 x 64 only
    if (g.split_qk == 3 && g.split_pv == 3) return launch_g64<3, 3>(g, s);
    if (g.split_qk == 3 && g.split_pv == 1) return launch_g64<3, 1>(g, s);
    if (g.split_qk == 1 && g.split_pv == 1) return launch_g64<1, 1>(g, s);
    return CVLM_E_UNSUPPORTED;
}
