// Fused multi-head attention for gfx950 (flash style: the S x S score matrix never exists).
//
// One workgroup = NW waves x 32 query rows of one (sequence, head); key/value tiles of 64 slots are
// staged through LDS by the whole workgroup (register staging, next tile prefetched into VGPRs under
// the MFMAs of the current one).
//
// Orientation (guide §3 "accumulator tile as the next MFMA's operand"): both products are issued
// transposed so that the QUERY index sits on the lane for the scores *and* for the output:
//     S^T[slot][q] = K . Q^T      mfma_32x32x16(A = K rows (LDS, ds_read_b128), B = Q (registers))
//     O^T[d][q]    = V^T . P^T    mfma_32x32x16(A = V^T (LDS, ds_read_b64_tr_b16), B = P (own accumulators))
// Each lane owns one query: the online-softmax running max / sum and the O rescale factor are
// lane-local scalars, the row reductions are 16 in-register ops + one exchange with lane^32, and P
// never leaves the register file (the k-order of the PV product is chosen to match the S^T
// accumulator register order, so no permutes are needed).
//
// Decomposed relative position bias (image_encoder.py:589-625): bias[q][key] = q.Rh[qh-kh+L-1] +
// q.Rw[qw-kw+L-1] with the *unscaled* q.  The workgroup first forms U = Q.R^T for all 2L-1 table rows
// with MFMA and scatters it into two per-query LDS tables Th[q][kh], Tw[q][kw]; the main loop adds
// Th + Tw to the scaled scores.
//
// Window mode reproduces window_partition's zero padding *after* norm1 (image_encoder.py:432-436,
// 520-523): pad tokens carry q = k = v = qkv bias, take part in the softmax, and are not stored.
//
// Precision: operands are split-half planes; SQK / SPV = 3 issue hi*hi + lo*hi + hi*lo.
#include "common.h"
#include <stdlib.h>
#include "../../include/cvlm.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct AttnParams {
    cvlm_attn_args a;
    int S_seq;      // slots per sequence (S, or window*window)
    int L;          // rel-pos side length (grid or window); 0 in mode 0
    int LTP;        // pitch of the bias tables (floats)
    int nwx;        // windows per row (mode 2)
    int D;          // heads*hd
};

__device__ __forceinline__ half4 lds_read_tr16(const half_t* p) {
    typedef __fp16 fp16x4 __attribute__((ext_vector_type(4)));
    fp16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LDS_AS fp16x4*)p);
    return __builtin_bit_cast(half4, r);
}

template <int HD, int NW, int MODE, int SQK, int SPV, bool CAUSAL>
__global__ __launch_bounds__(NW * 64) void attn_kernel(const AttnParams p) {
    constexpr int CPR = HD / 8;                   // 16-byte chunks per K/V row
    constexpr int KP = HD + 8;                    // LDS row pitch in halves (odd number of 16-B chunks)
    constexpr int KS = HD / 16;                   // k-steps of the QK^T product
    constexpr int ND = (HD + 31) / 32;            // 32-row d tiles of O^T
    constexpr int NPL = (SQK == 3 || SPV == 3) ? 2 : 1;
    constexpr int NT = NW * 64;
    constexpr int PLANE = 64 * KP;                // halves per staged plane
    constexpr int UNITS = 2 * NPL * 64 * CPR;     // 16-byte units per K+V tile
    constexpr int UPT = (UNITS + NT - 1) / NT;    // units per thread

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half_t* Ks = (half_t*)smem;                                    // [NPL][64][KP]
    half_t* Vs = Ks + NPL * PLANE;                                 // [NPL][64][KP]
    int* rowoff = (int*)(Vs + NPL * PLANE + 64);                   // [64] (mode 2), after 128 B of slack
    float* Th = (float*)(rowoff + 64);                             // [NW*32][LTP]
    float* Tw = Th + (MODE != 0 ? NW * 32 * p.LTP : 0);

    const cvlm_attn_args& g = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qc = lane & 31, half = lane >> 5;
    const int head = blockIdx.y, seq = blockIdx.z;
    const int S_seq = p.S_seq, L = p.L, D = p.D;
    const half_t* qkv_hi = (const half_t*)g.qkv_hi;
    const half_t* qkv_lo = (const half_t*)g.qkv_lo;
    const half_t* pad_hi = (const half_t*)g.pad_hi;
    const half_t* pad_lo = (const half_t*)g.pad_lo;
    const int64_t qkv_plane = qkv_lo - qkv_hi, pad_plane = pad_lo - pad_hi;   // lo-plane displacement (elements)

    int b = seq, wy = 0, wx = 0;
    if (MODE == 2) {
        const int nwin = p.nwx * p.nwx;
        b = seq / nwin;
        const int w = seq - b * nwin;
        wy = w / p.nwx; wx = w - wy * p.nwx;
    }
    // token index (within image b) of a slot of this sequence, or -1 for a window pad token
    auto token_of = [&](int slot) -> int {
        if (MODE == 2) {
            const int iy = slot / g.window, ix = slot - iy * g.window;
            const int y = wy * g.window + iy, x = wx * g.window + ix;
            if (y >= g.grid || x >= g.grid) return -1;
            return y * g.grid + x;
        }
        return slot;
    };
    const QkvStrides QS = qkv_strides(g.qkv_layout, g.S, g.B, g.heads, HD);

    // ---------------- queries: one per lane (lane & 31), fragments straight from global memory
    const int q0 = blockIdx.x * (NW * 32) + wave * 32;
    const bool wave_active = q0 < S_seq;
    const int qslot = q0 + qc;
    const bool qvalid = qslot < S_seq;
    const int qs = qvalid ? qslot : S_seq - 1;
    const int qtok = token_of(qs);
    half8 qh[KS], ql[KS];
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

    // ---------------- rel-pos tables Th[q][kh], Tw[q][kw] (fp32, LDS)
    if (MODE != 0) {
        const int qhh = qs / L, qww = qs - qhh * L;
        const int nrel = 2 * L - 1;
#pragma unroll 1
        for (int tb = 0; tb < 2; ++tb) {
            const half_t* Rhi = (const half_t*)(tb ? g.relw_hi : g.relh_hi);
            const half_t* Rlo = (const half_t*)(tb ? g.relw_lo : g.relh_lo);
            const int cq = tb ? qww : qhh;
            float* T = (tb ? Tw : Th) + (wave * 32 + qc) * p.LTP;
#pragma unroll 1
            for (int st = 0; st * 32 < nrel; ++st) {
                floatx16 u;
#pragma unroll
                for (int r = 0; r < 16; ++r) u[r] = 0.f;
                int rr = st * 32 + qc;
                rr = rr < nrel ? rr : nrel - 1;
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
                    const int kidx = cq + L - 1 - j;
                    if (j < nrel && kidx >= 0 && kidx < L) T[kidx] = u[r];
                }
            }
        }
    }

    // ---------------- K/V tile staging (global -> registers -> LDS)
    const int nkt = (S_seq + 63) / 64;
    half8 stage[UPT];
    auto row_token = [&](int t, int row) -> int {
        int slot = t * 64 + row;
        slot = slot < S_seq ? slot : S_seq - 1;
        return token_of(slot);
    };
    auto prefetch = [&](int t) {
#pragma unroll
        for (int i = 0; i < UPT; ++i) {
            const int u = tid + i * NT;
            if (UNITS % NT == 0 || u < UNITS) {
                const int chunk = u % CPR;
                const int row = (u / CPR) & 63;
                const int po = u / (CPR * 64);             // (operand, plane)
                const int op = po / NPL, pl = po - op * NPL;
                const int tok = (MODE == 2) ? rowoff[row] : row_token(t, row);
                const int64_t ro = tok < 0 ? (int64_t)(op + 1) * D + head * HD
                                           : qkv_offset(QS, b, tok, op + 1, head);
                // plane / pad selection by integer arithmetic (a 4-way pointer select was turned into a
                // stack lookup table by the compiler, i.e. scratch traffic in the staging loop)
                const int64_t po_ = tok < 0 ? pad_plane * pl : qkv_plane * pl;
                const half_t* base = (tok < 0 ? pad_hi : qkv_hi) + po_ + ro;
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
                const int row = (u / CPR) & 63;
                const int po = u / (CPR * 64);
                *(half8*)(Ks + po * PLANE + row * KP + chunk * 8) = stage[i];   // Vs follows Ks contiguously
            }
        }
    };

    if (MODE == 2) {
        if (tid < 64) rowoff[tid] = row_token(0, tid);
        __syncthreads();
    }
    prefetch(0);

    // ---------------- running state (one query per lane)
    float m_run = -INFINITY, l_run = 0.f;
    floatx16 o[ND];
#pragma unroll
    for (int n = 0; n < ND; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[n][r] = 0.f;

    const int qhh = (MODE != 0) ? qs / L : 0;
    const int qww = (MODE != 0) ? qs - qhh * L : 0;
    (void)qhh; (void)qww;
    const float* Thq = Th + (wave * 32 + qc) * p.LTP;
    const float* Twq = Tw + (wave * 32 + qc) * p.LTP;
    const float scale = g.scale;
    // transposed V read: group = lane>>4, lane i = lane&15 supplies row (i>>2), 4 halves at column 4*(i&3)
    const int tg = lane >> 4, ti = lane & 15;
    const int v_lane_off = (4 * (tg >> 1) + (ti >> 2)) * KP + 16 * (tg & 1) + 4 * (ti & 3);

    for (int t = 0; t < nkt; ++t) {
        commit();
        if (MODE == 2 && t + 1 < nkt && tid < 64) rowoff[tid] = row_token(t + 1, tid);
        __syncthreads();
        if (t + 1 < nkt) prefetch(t + 1);

        if (wave_active) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                if (t * 64 + sub * 32 >= S_seq) break;                 // wave-uniform
                // ---- S^T = K . Q^T
                floatx16 s;
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = 0.f;
                const half_t* kr = Ks + (sub * 32 + qc) * KP + 8 * half;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const half8 kh = *(const half8*)(kr + 16 * ks);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], s, 0, 0, 0);
                    if (SQK == 3) {
                        const half8 kl = *(const half8*)(kr + PLANE + 16 * ks);
                        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], s, 0, 0, 0);
                        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], s, 0, 0, 0);
                    }
                }
                // ---- scale, bias, mask
                const int b0 = t * 64 + sub * 32 + 4 * half;
                int kh0 = 0, kw0 = 0;
                if (MODE != 0) { kh0 = b0 / L; kw0 = b0 - kh0 * L; }
                float mx = -INFINITY;
                // MODE 0 (the CLIP towers): nothing is added to the scores, so they stay UNSCALED in the registers -- the factor rides
                // on the exponent's multiplier below and on the one maximum -- and slots are masked only where a mask can bite (the last
                // key tile; the causal text tower): 16 multiplies and 32 compare / select instructions less per 32 keys in a kernel
                // whose vector instructions are not hidden behind its MFMAs (profiles/r04_mfma_valu_overlap.log)
                if (MODE == 0) {
                    if (CAUSAL || t * 64 + sub * 32 + 32 > S_seq) {       // wave-uniform
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int slot = b0 + (r & 3) + 8 * (r >> 2);
                            const bool ok = (slot < S_seq) && (!CAUSAL || slot <= qslot);
                            s[r] = ok ? s[r] : -INFINITY;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
                    mx *= scale;                                          // scale > 0
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (MODE == 0) break;
                    const int c = (r & 3) + 8 * (r >> 2);
                    const int slot = b0 + c;
                    float v = s[r] * scale;
                    if (MODE != 0) {
                        int kw = kw0 + c, kh = kh0;
                        if (kw >= L) { kw -= L; ++kh; }
                        if (kw >= L) { kw -= L; ++kh; }
                        if (kw >= L) { kw -= L; ++kh; }
                        kh = kh < L ? kh : L - 1;
                        v += Thq[kh] + Twq[kw];
                    }
                    const bool ok = (slot < S_seq) && (!CAUSAL || slot <= qslot);
                    v = ok ? v : -INFINITY;
                    s[r] = v;
                    mx = fmaxf(mx, v);
                }
                mx = half_swap_max(mx);
                // the reference point of the exponentials moves only when some query of the wave meets a score more than TAU above its
                // own (attention_win2.hip): otherwise accumulators, sum and factor are left alone
                constexpr float TAU = 5.0f;
                if (__builtin_amdgcn_ballot_w64(mx > m_run + TAU) != 0) {   // wave-uniform; always on the first tile (m_run = -inf)
                    const float m_new = fmaxf(m_run, mx);
                    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
                    l_run *= alpha;
#pragma unroll
                    for (int n = 0; n < ND; ++n)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[n][r] *= alpha;
                    m_run = m_new;
                }
                const float lscale = MODE == 0 ? scale * LOG2E : LOG2E;
                const f32x2 c2 = f32x2{-m_run * LOG2E, -m_run * LOG2E}, l2 = f32x2{lscale, lscale};
                f32x2 z[8], acc = f32x2{0.f, 0.f};                   // packed fp32: half the VALU instructions
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x2 a = f32x2{s[2 * i], s[2 * i + 1]} * l2 + c2;
                    z[i] = f32x2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
                    acc += z[i];
                }
                l_run += acc.x + acc.y;
                // ---- O^T += V^T . P^T   (k-order of each 16-slot step = accumulator register order)
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    half8 ph, pl;
#pragma unroll
                    for (int p2 = 0; p2 < 4; ++p2) {                  // hi truncated (cvt_pkrtz), lo = e - hi: exact remainder
                        const f32x2 e = z[4 * k2 + p2];
                        const half2v h = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(e.x, e.y));
                        ph[2 * p2] = h[0]; ph[2 * p2 + 1] = h[1];
                        if (SPV == 3) {
                            const half2v l = __builtin_bit_cast(half2v, split_lo_pk(__builtin_bit_cast(unsigned, h), e.x, e.y));
                            pl[2 * p2] = l[0]; pl[2 * p2 + 1] = l[1];
                        }
                    }
                    const half_t* vb = Vs + (sub * 32 + 16 * k2) * KP + v_lane_off;
#pragma unroll
                    for (int n = 0; n < ND; ++n) {
                        const half4 v0 = lds_read_tr16(vb + 32 * n);
                        const half4 v1 = lds_read_tr16(vb + 32 * n + 8 * KP);
                        const half8 vh = half8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                        o[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o[n], 0, 0, 0);
                        if (SPV == 3) {
                            const half4 w0 = lds_read_tr16(vb + PLANE + 32 * n);
                            const half4 w1 = lds_read_tr16(vb + PLANE + 32 * n + 8 * KP);
                            const half8 vl = half8{w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
                            o[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o[n], 0, 0, 0);
                            o[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o[n], 0, 0, 0);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---------------- epilogue: O^T[d][q] / l -> out[token(q)][head*HD + d]
    const float l_tot = half_swap_sum(l_run);
    if (wave_active && qvalid && qtok >= 0) {
        const float inv = 1.0f / l_tot;
        const int64_t orow = ((int64_t)b * g.S + qtok) * D + head * HD;
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
}

template <int HD, int NW, int MODE, int SQK, int SPV, bool CAUSAL>
int launch(const AttnParams& p, hipStream_t s) {
    constexpr int KP = HD + 8;
    constexpr int NPL = (SQK == 3 || SPV == 3) ? 2 : 1;
    size_t smem = (size_t)2 * NPL * 64 * KP * 2 + 128 + 64 * 4;
    if (MODE != 0) smem += (size_t)2 * NW * 32 * p.LTP * 4;
    auto kern = attn_kernel<HD, NW, MODE, SQK, SPV, CAUSAL>;
    if (smem > 160 * 1024) return CVLM_E_UNSUPPORTED;
    if (smem > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    const int nseq = (MODE == 2) ? p.a.B * p.nwx * p.nwx : p.a.B;
    const int q_rows = (MODE == 0 && p.a.q_rows > 0 && p.a.q_rows < p.S_seq) ? p.a.q_rows : p.S_seq;   // ABI 9: leading query blocks only
    dim3 grid((q_rows + NW * 32 - 1) / (NW * 32), p.a.heads, nseq), block(NW * 64);
    hipLaunchKernelGGL(kern, grid, block, smem, s, p);
    CVLM_CHECK_LAUNCH();
    return 0;
}

template <int HD, int NW, int MODE, bool CAUSAL>
int launch_split(const AttnParams& p, hipStream_t s) {
    const int sq = p.a.split_qk, sp = p.a.split_pv;
    // split 2 (include/cvlm.h) has kernels for the ViT-H geometries only: everywhere else it runs as split 3, which holds its products and more
    // (1, 2) -- K's and Q's hi planes in the scores, round 6 -- likewise
    if ((sq == 3 && sp == 3) || (sq == 2 && sp == 2) || (sq == 1 && sp == 2)) return launch<HD, NW, MODE, 3, 3, CAUSAL>(p, s);
    if (sq == 3 && sp == 1) return launch<HD, NW, MODE, 3, 1, CAUSAL>(p, s);
    if (sq == 1 && sp == 1) return launch<HD, NW, MODE, 1, 1, CAUSAL>(p, s);
    return CVLM_E_UNSUPPORTED;
}

}  // namespace

// The two ViT-H kernels of the parity modes (split 3 or split 2 on both products): every other precision, window size and map size runs
// the generic kernel of this file.  (Rounds 1-4 also built a single-group global kernel and a two-workgroups-per-pair window kernel for
// the non-parity precisions: they carried no headline and left the build in round 5.)
int cvlm_attention_global64_pp(const cvlm_attn_args& g, hipStream_t s);    // attention_g64pp.hip: 64 x 64 and 96 x 96 maps
int cvlm_attention_window14_pc(const cvlm_attn_args& g, hipStream_t s);    // attention_win2.hip: 14 x 14 windows, h2 output

int64_t cvlm_attention_global64_pp_workspace_bytes(const cvlm_attn_args& g);   // attention_g64pp.hip

extern "C" int64_t cvlm_attention_workspace_bytes(const cvlm_attn_args* args) {
    if (!args) return 0;
    return cvlm_attention_global64_pp_workspace_bytes(*args);
}

extern "C" int cvlm_attention(const cvlm_attn_args* args, void* stream) {
    if (!args || !args->qkv_hi || !args->out_hi) return CVLM_E_BADARG;
    const cvlm_attn_args& g = *args;
    if (g.B <= 0 || g.S <= 0 || g.heads <= 0 || g.q_rows < 0 || (g.q_rows > 0 && g.mode != 0)) return CVLM_E_BADARG;
    if ((g.split_qk >= 2 || g.split_pv >= 2) && !g.qkv_lo) return CVLM_E_BADARG;
    AttnParams p;
    p.a = g;
    p.D = g.heads * g.hd;
    p.S_seq = g.S; p.L = 0; p.LTP = 0; p.nwx = 1;
    hipStream_t s = (hipStream_t)stream;
    if (g.mode == 0) {
        if (g.hd != 64) return CVLM_E_UNSUPPORTED;
        return g.causal ? launch_split<64, 4, 0, true>(p, s) : launch_split<64, 4, 0, false>(p, s);
    }
    if (g.hd != 80 || g.causal) return CVLM_E_UNSUPPORTED;
    if (!g.relh_hi || !g.relw_hi || (g.split_qk >= 2 && (!g.relh_lo || !g.relw_lo))) return CVLM_E_BADARG;
    if (g.grid <= 0 || g.S != g.grid * g.grid) return CVLM_E_BADARG;
    if (g.mode == 1) {
        const bool parity_split = (g.split_qk == g.split_pv && g.split_qk >= 2) || (g.split_qk == 1 && g.split_pv == 2);
        if ((g.grid == 64 || g.grid == 96) && parity_split) {   // 1024^2 / 1536^2 SAM geometries, parity modes
            const int rc = cvlm_attention_global64_pp(g, s);
            if (rc != CVLM_E_UNSUPPORTED) return rc;                           // incl. CVLM_E_WORKSPACE: a missing workspace is an error, not a silent fallback
        }
        p.L = g.grid; p.LTP = g.grid | 1;
        return launch_split<80, 4, 1, false>(p, s);
    }
    if (g.mode == 2) {
        if (g.window <= 0 || !g.pad_hi || ((g.split_qk >= 2 || g.split_pv >= 2) && !g.pad_lo)) return CVLM_E_BADARG;
        const bool parity_split = (g.split_qk == g.split_pv && g.split_qk >= 2) || (g.split_qk == 1 && g.split_pv == 2);
        if (g.window == 14 && parity_split && g.out_lo) return cvlm_attention_window14_pc(g, s);   // SAM window geometry, parity modes
        p.L = g.window; p.LTP = g.window | 1;
        p.nwx = (g.grid + g.window - 1) / g.window;
        p.S_seq = g.window * g.window;
        if (p.S_seq <= 7 * 32 && p.S_seq > 4 * 32) return launch_split<80, 7, 2, false>(p, s);
        return launch_split<80, 4, 2, false>(p, s);
    }
    return CVLM_E_BADARG;
}
