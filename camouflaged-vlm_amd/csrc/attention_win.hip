// ViT-H *window* attention (14x14 windows, head_dim 80) with decomposed rel-pos bias
// (image_encoder.py:488-504, 507-553, 589-625; 28 of the 32 SAM blocks).
//
// Two workgroups of 4 waves x 32 queries (128 + 68 of the 196 queries; 2 workgroups/CU at <= 256 VGPRs)
// per (image, window, head); K/V of the whole window
// stream through LDS in seven 32-slot tiles (query on the lane for S^T and O^T, pad tokens = qkv bias
// rows, not stored).  The tiles arrive by LDS-DMA into a two-slot ring, one tile ahead of the MFMAs and
// with one barrier per tile: with register staging and two barriers the loop ran at the latency of a
// global load per tile (31 us per workgroup for 9 us of MFMA work).  What differs is the bias: instead of gathering
// Th[q][kh] + Tw[q][kw] per score element (index math + two LDS reads per element made the generic
// kernel VALU-bound), the bias is folded into the QK^T contraction:
//     Q_aug = [ q (80) | Th[q][0..13]/scale | Tw[q][0..13]/scale | 0 0 0 0 ]      (112 = 7 k-steps of 16)
//     K_aug = [ k (80) | onehot14(kh)       | onehot14(kw)       | 0 0 0 0 ]
// so S^T = K_aug . Q_aug^T already contains (q.k + bias/scale).  The one-hot block is the same for every
// window and head (a 224 x 32 constant in LDS); Th/Tw are formed once per workgroup with MFMA
// (U = Q . R^T, 27 table rows) and kept as two extra split-half B fragments per lane.
#include "common.h"
#include "../../include/cvlm.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ half4 lds_read_tr16(const half_t* p) {
    typedef __fp16 fp16x4 __attribute__((ext_vector_type(4)));
    fp16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LDS_AS fp16x4*)p);
    return __builtin_bit_cast(half4, r);
}

// timeline probe (tools/trace_attn_win.py): 4 x u64 per workgroup when set
__device__ unsigned long long* g_win_trace = nullptr;

template <int SQK, int SPV>
__global__ __launch_bounds__(256, 2) void attn_win14_kernel(const cvlm_attn_args g, const int nwx) {
    constexpr int HD = 80, KS = 5, ND = 3, CPR = 10, KP = 88, VP = 96, L = 14, S_SEQ = 196;
    constexpr int OP = 40;                                    // one-hot row pitch (halves): 5 chunks, odd
    constexpr int NPL = (SQK == 3 || SPV == 3) ? 2 : 1;
    constexpr int KT = 32, NKT = 7;                           // 7 x 32 = 224 >= 196 key slots
    constexpr int KPLANE = KT * KP, VPLANE = KT * VP;
    constexpr int DMA_PER_WAVE = 3 * NPL;                      // 12 * NPL one-KiB instructions per tile over 4 waves

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // slot image (bytes): K planes at pl * 6144 (32 rows x 176 B, padded to six 1-KiB DMA instructions),
    // V planes at NPL * 6144 + pl * 6144 (32 rows x 192 B)
    constexpr int PLANE_B = 6144, SLOT_B = 2 * NPL * PLANE_B;
    static_assert(KPLANE * 2 <= PLANE_B && VPLANE * 2 == PLANE_B, "plane images");
    constexpr int TAUG_B = 128 * 33 * 4, TAIL_OFF = 2 * SLOT_B + 224 * OP * 2 + 224 * 8;
    constexpr int TAUG_OFF = TAUG_B <= SLOT_B ? SLOT_B : TAIL_OFF;   // aliases slot 1 when it fits, else its own region
    half_t* OH = (half_t*)(smem + 2 * SLOT_B);                // [224][OP] one-hot(kh) | one-hot(kw)
    long long* tokoff = (long long*)(OH + 224 * OP);          // [224] K-row element offset (>= 0: qkv, < 0: -(1 + pad offset))
    float* Taug = (float*)(smem + TAUG_OFF);                  // prologue only: [128][33]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long* const trace = g_win_trace;
    const unsigned long long tr0 = trace ? wall_clock64() : 0;
    unsigned long long tr1 = 0, tr2 = 0, ta = 0, tb_ = 0, tc = 0, td = 0;
    const int qc = lane & 31, half = lane >> 5;
    const int head = blockIdx.y, seq = blockIdx.z;
    const int D = g.heads * HD;
    const half_t* qkv_hi = (const half_t*)g.qkv_hi;
    const half_t* qkv_lo = (const half_t*)g.qkv_lo;
    const half_t* pad_hi = (const half_t*)g.pad_hi;
    const half_t* pad_lo = (const half_t*)g.pad_lo;
    const int64_t qkv_plane = qkv_lo - qkv_hi, pad_plane = pad_lo - pad_hi;   // lo-plane displacement (elements)
    const int nwin = nwx * nwx;
    const int b = seq / nwin, w = seq - b * nwin;
    const int wy = w / nwx, wx = w - wy * nwx;
    auto token_of = [&](int slot) -> int {
        const int iy = slot / L, ix = slot - iy * L;
        const int y = wy * L + iy, x = wx * L + ix;
        if (y >= g.grid || x >= g.grid) return -1;
        return y * g.grid + x;
    };
    const int SI = g.grid * g.grid;
    const QkvStrides QS = qkv_strides(g.qkv_layout, SI, g.B, g.heads, HD);

    // ---- rel-pos table fragments (A operand of U = R . Q^T): independent of everything else, so their global
    // loads go out first and land under the setup below
    half8 rfh[2][KS], rfl[2][KS];
    {
        const int rr = qc < 27 ? qc : 26;
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const half_t* Rhi = (const half_t*)(tb ? g.relw_hi : g.relh_hi);
            const half_t* Rlo = (const half_t*)(tb ? g.relw_lo : g.relh_lo);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                rfh[tb][ks] = *(const half8*)(Rhi + rr * HD + 16 * ks + 8 * half);
                if (SQK == 3) rfl[tb][ks] = *(const half8*)(Rlo + rr * HD + 16 * ks + 8 * half);
            }
        }
    }
    // ---- K-row offsets of the 224 key slots (slots >= 196 repeat the last key; they are masked in the softmax)
    if (tid < 224) {
        const int tok = token_of(tid < S_SEQ ? tid : S_SEQ - 1);
        tokoff[tid] = tok < 0 ? -(1 + (long long)D + head * HD) : (long long)qkv_offset(QS, b, tok, 1, head);
    }
    // ---- constant one-hot block of K_aug: one slot row per thread (224 rows, four 16-byte stores each)
    if (tid < 224) {
        const int kh = tid / L, kw = tid - kh * L;
        half_t row[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) row[c] = (half_t)((tid < S_SEQ && (c == kh || c == 14 + kw)) ? 1.0f : 0.0f);
#pragma unroll
        for (int c8 = 0; c8 < 4; ++c8)
            *(half8*)(OH + tid * OP + 8 * c8) = half8{row[8 * c8], row[8 * c8 + 1], row[8 * c8 + 2], row[8 * c8 + 3],
                                                      row[8 * c8 + 4], row[8 * c8 + 5], row[8 * c8 + 6], row[8 * c8 + 7]};
    }

    // ---- K/V tiles by LDS-DMA.  Instruction i of a tile (i < 12 * NPL): operand i / (6 * NPL), plane
    // (i / 6) % NPL, sixth i % 6 of the plane image; its 64 lanes write consecutive 16-byte chunks.  A lane
    // whose chunk is row padding (K: 11th chunk of a row, V: 11th / 12th, K image rows >= 32) re-reads chunk 0.
    auto issue_tile = [&](int t, int slot) {
        unsigned char* sbase = smem + slot * SLOT_B;
        const half_t* srcs[DMA_PER_WAVE];
#pragma unroll
        for (int j = 0; j < DMA_PER_WAVE; ++j) {                // all offset reads first, then the DMA instructions
            const int i = wave * DMA_PER_WAVE + j;
            const int op = i / (6 * NPL), pl = (i / 6) % NPL, sub = i % 6;
            const int c = sub * 64 + lane;
            const int cpr = op ? 12 : 11;
            int row = c / cpr, ch = c - row * cpr;
            if (row >= KT) row = KT - 1;
            if (ch >= CPR) ch = 0;
            const long long ko = tokoff[t * KT + row];
            const bool pad = ko < 0;
            const long long o = pad ? (-ko - 1) + (op ? D : 0) : ko + (op ? QS.sop : 0);
            srcs[j] = (pad ? pad_hi + pl * pad_plane : qkv_hi + pl * qkv_plane) + o + ch * 8;
        }
#pragma unroll
        for (int j = 0; j < DMA_PER_WAVE; ++j) {
            const int i = wave * DMA_PER_WAVE + j;
            const int op = i / (6 * NPL), pl = (i / 6) % NPL, sub = i % 6;
            glds16(srcs[j], sbase + (op * NPL + pl) * PLANE_B + sub * 1024);
        }
    };
    __syncthreads();                                          // tokoff visible
    issue_tile(0, 0);                                         // lands under the Th / Tw prologue below
    if (trace) ta = wall_clock64();

    // ---- queries
    const int q0 = blockIdx.x * 128 + wave * 32;
    const bool wave_active = q0 < S_SEQ;                      // wave-uniform
    const int qslot = q0 + qc;
    const bool qvalid = qslot < S_SEQ;
    const int qs = qvalid ? qslot : S_SEQ - 1;
    const int qtok = token_of(qs);
    half8 qh[KS + 2], ql[KS + 2];
    {
        const int64_t qo = qtok < 0 ? (int64_t)head * HD : qkv_offset(QS, b, qtok, 0, head);
        const half_t* bh = (qtok < 0 ? pad_hi : qkv_hi) + qo;
        const half_t* bl = bh + (qtok < 0 ? pad_plane : qkv_plane);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qh[ks] = *(const half8*)(bh + 16 * ks + 8 * half);
            if (SQK == 3) ql[ks] = *(const half8*)(bl + 16 * ks + 8 * half);
        }
    }
    if (trace) { asm volatile("" ::"v"(qh[0]), "v"(qh[4])); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tb_ = wall_clock64(); }
    // ---- Th / Tw for this query: U = Q . R^T (27 rows -> one 32-row MFMA tile per table), scattered to Taug[q][..]
    {
        const int qhh = qs / L, qww = qs - qhh * L;
        float* Tq = Taug + (wave * 32 + qc) * 33;
        if (half == 0) { Tq[28] = 0.f; Tq[29] = 0.f; Tq[30] = 0.f; Tq[31] = 0.f; }
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const int cq = tb ? qww : qhh;
            floatx16 u;
#pragma unroll
            for (int r = 0; r < 16; ++r) u[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                u = __builtin_amdgcn_mfma_f32_32x32x16_f16(rfh[tb][ks], qh[ks], u, 0, 0, 0);
                if (SQK == 3) {
                    u = __builtin_amdgcn_mfma_f32_32x32x16_f16(rfl[tb][ks], qh[ks], u, 0, 0, 0);
                    u = __builtin_amdgcn_mfma_f32_32x32x16_f16(rfh[tb][ks], ql[ks], u, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int kidx = cq + L - 1 - j;
                if (j < 27 && kidx >= 0 && kidx < L) Tq[tb * 14 + kidx] = u[r];
            }
        }
    }
    if (trace) tc = wall_clock64();
    __syncthreads();
    {
        // augmented B fragments: k-step 5 covers aug dims 0..15, k-step 6 dims 16..31; lane holds dims 8*half .. +8
        // the scores use q * scale (image_encoder.py:496), the rel-pos tables q itself (:497-500): fold the scale
        // into the query fragments now that U is done; the augmented columns then carry T unscaled
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                half_t hh, ll;
                split_h2(((float)qh[ks][j] + (SQK == 3 ? (float)ql[ks][j] : 0.f)) * g.scale, hh, ll);
                qh[ks][j] = hh;
                if (SQK == 3) ql[ks][j] = ll;
            }
        const float* Tq = Taug + (wave * 32 + qc) * 33;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                half_t h, l;
                split_h2(Tq[16 * a + 8 * half + j], h, l);
                qh[KS + a][j] = h;
                ql[KS + a][j] = (SQK == 3) ? l : (half_t)0.f;
            }
    }
    __syncthreads();                                          // Taug (aliasing K/V) is dead from here on
    if (trace) td = wall_clock64();

    float m_run = -INFINITY, l_run = 0.f;
    floatx16 o[ND];
#pragma unroll
    for (int n = 0; n < ND; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[n][r] = 0.f;
    if (trace) tr1 = wall_clock64();
    const int tg = lane >> 4, ti = lane & 15;
    const int v_lane_off = (4 * (tg >> 1) + (ti >> 2)) * VP + 16 * (tg & 1) + 4 * (ti & 3);

#pragma unroll 1
    for (int t = 0; t < NKT; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of tile t has landed
        __syncthreads();                                      // ... everyone's has, and slot (t + 1) & 1 is free
        if (t + 1 < NKT) issue_tile(t + 1, (t + 1) & 1);
        const half_t* Ks = (const half_t*)(smem + (t & 1) * SLOT_B);
        const half_t* Vs = Ks + NPL * (PLANE_B / 2);
        if (wave_active) {
            const int sub = 0;
            floatx16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const half_t* kr = Ks + (sub * 32 + qc) * KP + 8 * half;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const half8 kh = *(const half8*)(kr + 16 * ks);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], s, 0, 0, 0);
                if (SQK == 3) {
                    const half8 kl = *(const half8*)(kr + PLANE_B / 2 + 16 * ks);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], s, 0, 0, 0);
                }
            }
            const half_t* ohr = OH + (t * KT + sub * 32 + qc) * OP + 8 * half;
#pragma unroll
            for (int a = 0; a < 2; ++a) {                     // bias: exact one-hot rows x (T/scale) hi (+ lo)
                const half8 oh8 = *(const half8*)(ohr + 16 * a);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(oh8, qh[KS + a], s, 0, 0, 0);
                if (SQK == 3) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(oh8, ql[KS + a], s, 0, 0, 0);
            }
            // online softmax on packed fp32 pairs; only the last tile holds slots beyond the 196 keys
            if (t == NKT - 1) {
                const int b0 = t * KT + sub * 32 + 4 * half;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (b0 + (r & 3) + 8 * (r >> 2) >= S_SEQ) s[r] = -INFINITY;
            }
            float mx = fmaxf(s[0], s[1]);
#pragma unroll
            for (int r = 2; r < 16; r += 2) mx = fmaxf(fmaxf(mx, s[r]), s[r + 1]);
            mx = half_swap_max(mx);
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
            const f32x2 c2 = f32x2{-m_new * LOG2E, -m_new * LOG2E}, l2 = f32x2{LOG2E, LOG2E};
            f32x2 z[8], acc = f32x2{0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x2 a = f32x2{s[2 * i], s[2 * i + 1]} * l2 + c2;
                z[i] = f32x2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
                acc += z[i];
            }
            l_run = l_run * alpha + (acc.x + acc.y);
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
                for (int p2 = 0; p2 < 4; ++p2) {                      // hi truncated (cvt_pkrtz), lo = e - hi: exact remainder
                    const f32x2 e = z[4 * k2 + p2];
                    const half2v h = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(e.x, e.y));
                    ph[2 * p2] = h[0]; ph[2 * p2 + 1] = h[1];
                    if (SPV == 3) {
                        const half2v l = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(e.x - (float)h[0], e.y - (float)h[1]));
                        pl[2 * p2] = l[0]; pl[2 * p2 + 1] = l[1];
                    }
                }
                const half_t* vb = Vs + (sub * 32 + 16 * k2) * VP + v_lane_off;
#pragma unroll
                for (int n = 0; n < ND; ++n) {
                    const half4 v0 = lds_read_tr16(vb + 32 * n);
                    const half4 v1 = lds_read_tr16(vb + 32 * n + 8 * VP);
                    const half8 vh = half8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    o[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o[n], 0, 0, 0);
                    if (SPV == 3) {
                        const half4 w0 = lds_read_tr16(vb + PLANE_B / 2 + 32 * n);
                        const half4 w1 = lds_read_tr16(vb + PLANE_B / 2 + 32 * n + 8 * VP);
                        const half8 vl = half8{w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
                        o[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o[n], 0, 0, 0);
                        o[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o[n], 0, 0, 0);
                    }
                }
            }
        }
    }

    if (trace) tr2 = wall_clock64();
    const float l_tot = half_swap_sum(l_run);
    if (wave_active && qvalid && qtok >= 0) {
        const float inv = 1.0f / l_tot;
        const int64_t orow = ((int64_t)b * SI + qtok) * D + head * HD;
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
    if (trace && tid == 0) {
        unsigned long long* o = trace + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8;
        o[0] = tr0; o[1] = tr1; o[2] = tr2; o[3] = wall_clock64(); o[4] = ta; o[5] = tb_; o[6] = tc; o[7] = td;
    }
}

template <int SQK, int SPV>
int launch_win(const cvlm_attn_args& g, hipStream_t s) {
    constexpr int NPL = (SQK == 3 || SPV == 3) ? 2 : 1;
    constexpr int slot = 2 * NPL * 6144, taug = 128 * 33 * 4;
    constexpr int smem = 2 * slot + 224 * 40 * 2 + 224 * 8 + (taug <= slot ? 0 : taug);
    const int nwx = (g.grid + 13) / 14;
    auto kern = attn_win14_kernel<SQK, SPV>;
    static bool attr[16] = {};
    if (smem > 48 * 1024 && cvlm_first_on_device(attr))
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(kern, dim3(2, g.heads, g.B * nwx * nwx), dim3(256), smem, s, g, nwx);
    CVLM_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// called from cvlm_attention() for mode 2, window 14, head_dim 80
int cvlm_attention_window14(const cvlm_attn_args& g, hipStream_t s) {
    if (g.split_qk == 3 && g.split_pv == 3) return launch_win<3, 3>(g, s);
    if (g.split_qk == 3 && g.split_pv == 1) return launch_win<3, 1>(g, s);
    if (g.split_qk == 1 && g.split_pv == 1) return launch_win<1, 1>(g, s);
    return CVLM_E_UNSUPPORTED;
}

// Probe hook (not part of include/cvlm.h).
extern "C" int cvlm_debug_set_attn_win_trace(void* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_win_trace), &buf, sizeof(buf));
}
