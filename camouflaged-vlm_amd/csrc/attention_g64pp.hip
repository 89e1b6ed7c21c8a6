// ViT-H *global* attention on the 64x64 token map, ping-pong schedule (exact mode: hi/lo operands on both GEMMs).
// Query on the lane for S^T = K.Q^T and O^T = V^T.P^T (the layouts of the single-group kernel of rounds 1-4, which left the build in round 5;
// rel-pos bias = Th[q][kh] from LDS + Tw[q][kw] from 32 registers); what this form changed is WHO runs WHEN:
//
//   * one workgroup of 8 waves per 256 queries; waves w and w + 4 share a SIMD.  A key tile costs a wave two phases,
//         X(t): softmax of S(t)                      -- VALU / transcendental only
//         Y(t): O += V(t)^T.P(t), then S(t+1) = K(t+1).Q^T   -- MFMA + LDS reads only
//     and waves 4..7 run one phase behind waves 0..3, a raw s_barrier closing every phase.  In every phase one
//     wave of each SIMD pair is in X and its partner in Y, so the matrix pipe and the VALU are both busy all the
//     time (with two independent 4-wave workgroups per CU the same phases collided at random: MFMA busy 44 %,
//     VALU 46 %, sum ~ 90 %).
//   * V is consumed TRANSPOSED (rows = head dims, 32 keys per row): a small kernel writes V^T once per launch, so the
//     A fragments of O^T = V^T.P^T are plain ds_read_b128 (256 B/clk).  Read from the [key][dim] image with
//     ds_read_b64_tr_b16 (64 B/clk, 24 per tile and wave) the four waves of a phase saturated the LDS array: Y took
//     1.13 us against 0.79 us for its 33 MFMAs alone (probe builds, tools/trace_attn_g64.py).
//   * K and V tiles (32 keys) arrive by LDS-DMA into two three-slot rings, K running one tile ahead of V; they are
//     issued three (K) / two (V) tiles before their first use and retired with counted vmcnt, never drained.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "../../include/cvlm.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef CVLM_G64_PD
#define CVLM_G64_PD 3
#endif
// phase-time probe (tools/trace_attn_g64.py): per wave 8 x u64 {prologue, sum X, sum X-side wait+barrier, sum Y, sum Y-side wait+barrier, total}
__device__ unsigned long long* g_g64_trace = nullptr;

__device__ __forceinline__ half4 lds_read_tr16(const half_t* p) {
    typedef __fp16 fp16x4 __attribute__((ext_vector_type(4)));
    fp16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LDS_AS fp16x4*)p);
    return __builtin_bit_cast(half4, r);
}

// A operand of sixteen rows of ones: the 16x16x32 product ones . P^T puts the sum over a tile's 32 keys of the fp16 probabilities the P.V
// product really multiplies with in every row -- the softmax denominator of the form that keeps one fp16 per probability (PLO == false)
__device__ __forceinline__ half8 ones8() {
    const half_t o = (half_t)1.0f;
    return half8{o, o, o, o, o, o, o, o};
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void phase_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// PLO / QLO: the lo planes of P and of Q take part in their products (cvlm_attn_args.split_pv / split_qk == 3); false: two MFMAs per product,
// P as ONE fp16 per probability (rounded to nearest, the softmax denominator summed from the rounded values by a ones-row product), Q
// as its hi plane (split == 2: see include/cvlm.h).  KLO == false (split_qk == 1, round 6): K's lo plane stays out too -- ONE MFMA per
// k-step of the scores, and the plane is neither fetched nor read (profiles/r06_probe_kv_lo.log: fp16(k) costs the masks 4e-5; the
// rel-pos tables keep their three terms).  V always enters with both planes (fp16(v) alone costs 4e-4).
template <int L, bool PLO, bool QLO, bool KLO>
__global__ __launch_bounds__(512, 2) void attn_g64pp_kernel(const cvlm_attn_args g, const half_t* __restrict__ vt_hi,
                                                            const half_t* __restrict__ vt_lo) {
    // L = side of the token map (64: 1024^2 images, 96: 1536^2).  A key row is TPR = L / 32 tiles; the 96 map needs a
    // 99 KB Th table, so its V ring has two slots instead of three (fetched one phase later, drained, see tile_phases)
    constexpr int HD = 80, KS = 5, NDB = 5, KP = 88, KT = 32, LTP = L + 1, S = L * L, NTILE = S / KT, TPR = L / KT;
    constexpr int NSV = (L == 64) ? 3 : 2;
    static_assert(L % KT == 0 && S % 256 == 0 && NTILE % TPR == 0, "map side");
    constexpr int VROWS = 96, VROW_B = KT * 2;                       // V^T image: 96 rows (80 dims + 16 filler) x 32 keys
    constexpr int KPL_B = KT * KP * 2, VPL_B = VROWS * VROW_B;       // plane bytes: 5632 / 6144
    constexpr int KSLOT_B = 2 * KPL_B, VSLOT_B = 2 * VPL_B, NSLOT = 3;
    constexpr int KPL = KPL_B / 2;                                   // K plane pitch in halves
    constexpr int DPW = 3;                                           // DMA instructions per wave per tile

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Kring = smem;
    unsigned char* Vring = smem + NSLOT * KSLOT_B;
    float* T = (float*)(Vring + NSV * VSLOT_B);                      // [256][LTP]  (Tw first, then Th)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned long long* const trace = g_g64_trace;
    const unsigned long long tr_start = trace ? wall_clock64() : 0;
    unsigned long long tr_x = 0, tr_xb = 0, tr_y = 0, tr_yb = 0, tr_pro = 0;
    const bool grpB = wave >= 4;
    const int qc = lane & 31, half = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int D = g.heads * HD;
    const half_t* qkv_hi = (const half_t*)g.qkv_hi;
    const half_t* qkv_lo = (const half_t*)g.qkv_lo;
    const QkvStrides QS = qkv_strides(g.qkv_layout, S, g.B, g.heads, HD);

    // ---- DMA plan: waves 0..3 carry K (12 one-KiB instructions per tile: 2 planes x 6), waves 4..7 carry V.
    // Lane chunk c of a plane image: K rows are 11 chunks (10 data + 1 pad), V rows 12 (10 + 2); pad chunks re-read
    // chunk 0; the K image ends at chunk 352, so the upper half of its sixth instruction is masked off.
    const half_t* dsrc[DPW];
    int ddst[DPW];
    bool dok[DPW];
    int64_t tile_stride;                                             // elements between consecutive key tiles
    if (!grpB) {                                                     // K: [key][dim] rows of 11 chunks (10 data + 1 pad)
        tile_stride = (int64_t)KT * QS.st;
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            const int i = wave * DPW + j;                            // 0..11
            const int pl = i / 6, sub = i - pl * 6;
            const int c = sub * 64 + lane;
            int row = c / 11, ch = c - row * 11;
            dok[j] = row < KT && (KLO || pl == 0);                   // the K image ends at chunk 352; without K's lo plane waves 2, 3 carry nothing
            if (row >= KT) row = KT - 1;
            if (ch >= 10) ch = 0;
            dsrc[j] = (pl ? qkv_lo : qkv_hi) + qkv_offset(QS, b, row, 1, head) + ch * 8;
            ddst[j] = pl * KPL_B + sub * 1024;
        }
    } else {                                                         // V^T: [dim][key] rows of 4 chunks, rows 80..95 repeat row 79
        tile_stride = KT;
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            const int i = (wave - 4) * DPW + j;
            const int pl = i / 6, sub = i - pl * 6;
            const int row = sub * 16 + (lane >> 2), pos = lane & 3;
            const int chunk = pos ^ ((row >> 2) & 3);                // LDS position pos holds key chunk pos ^ f(row): conflict-free b128 reads
            const int drow = row < HD ? row : HD - 1;
            dok[j] = true;
            dsrc[j] = (pl ? vt_lo : vt_hi) + (((int64_t)b * g.heads + head) * HD + drow) * S + chunk * 8;
            ddst[j] = pl * VPL_B + sub * 1024;
        }
    }
    // dsrc[] always points at the next tile this wave has to fetch (running pointers: a `t * stride` product
    // ended up spilled and reloaded behind a vmcnt(0) in front of every DMA instruction)
    auto issue_next = [&](int slot) {                                // this wave's share of the next K or V tile
        unsigned char* base = (grpB ? Vring + slot * VSLOT_B : Kring + slot * KSLOT_B);
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            if (dok[j]) glds16(dsrc[j], base + ddst[j]);
            dsrc[j] += tile_stride;
        }
    };
    // prologue DMA: K(0..2) by waves 0..3, V(0..1) by waves 4..7 (they land under the table build below)
    issue_next(0);
    issue_next(1);
    if (!grpB) issue_next(2);

    // ---- queries, rel-pos tables
    const int qslot = blockIdx.x * 256 + wave * 32 + qc;
    half8 qh[KS], ql[KS];
    {
        const int64_t qo = qkv_offset(QS, b, qslot, 0, head);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qh[ks] = *(const half8*)(qkv_hi + qo + 16 * ks + 8 * half);
            ql[ks] = *(const half8*)(qkv_lo + qo + 16 * ks + 8 * half);
        }
    }
    const int qhh = qslot / L, qww = qslot - qhh * L;
    float* Tq = T + (wave * 32 + qc) * LTP;
    auto build_table = [&](const half_t* Rhi, const half_t* Rlo, int cq) {   // T[q][k] = (Q . R^T)[q][cq - k + L - 1]
#pragma unroll 1
        for (int st = 0; st < (2 * L - 1 + 31) / 32; ++st) {
            floatx16 u;
#pragma unroll
            for (int r = 0; r < 16; ++r) u[r] = 0.f;
            int rr = st * 32 + qc;
            rr = rr < 2 * L - 1 ? rr : 2 * L - 2;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const half8 ah = *(const half8*)(Rhi + rr * HD + 16 * ks + 8 * half);
                const half8 al = *(const half8*)(Rlo + rr * HD + 16 * ks + 8 * half);
                u = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[ks], u, 0, 0, 0);
                u = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[ks], u, 0, 0, 0);
                u = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[ks], u, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = st * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int kidx = cq + (L - 1) - j;
                if (j < 2 * L - 1 && kidx >= 0 && kidx < L) Tq[kidx] = u[r];
            }
        }
    };
    f32x2 twr[TPR][8];                                               // Tw[q][32*pz + (r&3) + 8*(r>>2) + 4*half], r = 2i, 2i+1
    build_table((const half_t*)g.relw_hi, (const half_t*)g.relw_lo, qww);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int pz = 0; pz < TPR; ++pz)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r0 = 2 * i, r1 = 2 * i + 1;
            twr[pz][i] = f32x2{Tq[32 * pz + (r0 & 3) + 8 * (r0 >> 2) + 4 * half], Tq[32 * pz + (r1 & 3) + 8 * (r1 >> 2) + 4 * half]};
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // each lane re-reads only the row it wrote
    build_table((const half_t*)g.relh_hi, (const half_t*)g.relh_lo, qhh);
    // the rel-pos tables use q as is (image_encoder.py:497-500); the scores use q * scale (:496): fold it in now
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            half_t hh, ll;
            split_h2(((float)qh[ks][j] + (float)ql[ks][j]) * g.scale, hh, ll);
            qh[ks][j] = hh; ql[ks][j] = ll;
        }

    // ---- state
    float m_run = -INFINITY, l_run = 0.f;
    floatx16 s;
    floatx4 o[NDB][2];                                               // O^T tiles (16x16x32 P.V, see attn_g64pair_kernel): [16-dim block][query block]
#pragma unroll
    for (int n = 0; n < NDB; ++n)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) o[n][qb] = floatx4{0.f, 0.f, 0.f, 0.f};
    floatx4 osum[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};   // !PLO: row sums of the ROUNDED probabilities (see ones8)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 xh[2] = {}, xl[2] = {};                                    // P^T as B operands of the 16x16x32 shape, per query block
    // V^T fragment of this lane: row (lane & 15) of a 16-row block, key group lane >> 4; position swizzled like the DMA image
    const int v_lane_off = (lane & 15) * VROW_B + (((lane >> 4) ^ ((lane >> 2) & 3)) * 16);
    const int k_lane_off = qc * KP + 8 * half;

    auto QK = [&](int slot) {                                        // S^T = K_tile . Q^T  (3 MFMAs per k-step)
        const half_t* kr = (const half_t*)(Kring + slot * KSLOT_B) + k_lane_off;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const half8 kh = *(const half8*)(kr + 16 * ks);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], s, 0, 0, 0);
            if constexpr (KLO) {
                const half8 kl = *(const half8*)(kr + KPL + 16 * ks);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], s, 0, 0, 0);
            }
            if constexpr (QLO) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], s, 0, 0, 0);
        }
    };
    // Y(t): O^T += V(t)^T . P^T (five 16-dim blocks, six 16x16x32 MFMAs each), then S^T = K(t+1) . Q^T (five k-steps, three
    // 32x32x16 MFMAs each) as ONE stream of ten stages; the LDS fragments of stage i + PD are requested before the MFMAs of
    // stage i.  The partner wave on this SIMD is in its VALU phase, so nothing else would cover this wave's LDS latency.
    auto Y = [&](int vslot, int kslot) {
        const unsigned char* vbase = Vring + vslot * VSLOT_B + v_lane_off;
        const half_t* kr = (const half_t*)(Kring + kslot * KSLOT_B) + k_lane_off;
        constexpr int PD = CVLM_G64_PD, RS = PD + 1, NST = NDB + 5;    // fragments requested PD stages ahead, ring of PD + 1
        half8 fa[RS], fb[RS];                                        // (hi, lo) fragment pair per stage
        auto load = [&](auto ic) {
            constexpr int I = decltype(ic)::value;
            if constexpr (I < NDB) {
                const unsigned char* vr = vbase + (16 * I) * VROW_B;
                fa[I % RS] = *(const half8*)vr;
                fb[I % RS] = *(const half8*)(vr + VPL_B);
            } else {
                constexpr int ks = I - NDB;
                fa[I % RS] = *(const half8*)(kr + 16 * ks);
                if constexpr (KLO) fb[I % RS] = *(const half8*)(kr + KPL + 16 * ks);
            }
        };
        auto compute = [&](auto ic) {
            constexpr int I = decltype(ic)::value;
            const half8 a = fa[I % RS], bq = fb[I % RS];
            if constexpr (I < NDB) {
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    const half8 bh = __builtin_bit_cast(half8, xh[qb]), bl = __builtin_bit_cast(half8, xl[qb]);
                    if constexpr (!PLO && I == 0) osum[qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones8(), bh, osum[qb], 0, 0, 0);
                    o[I][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bh, o[I][qb], 0, 0, 0);
                    o[I][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bq, bh, o[I][qb], 0, 0, 0);
                    if constexpr (PLO) o[I][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bl, o[I][qb], 0, 0, 0);
                }
            } else {
                constexpr int ks = I - NDB;
                if constexpr (ks == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = 0.f;
                }
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qh[ks], s, 0, 0, 0);
                if constexpr (KLO) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(bq, qh[ks], s, 0, 0, 0);
                if constexpr (QLO) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, ql[ks], s, 0, 0, 0);
            }
        };
        load(std::integral_constant<int, 0>{});
        if constexpr (PD >= 2) load(std::integral_constant<int, 1>{});
        if constexpr (PD >= 3) load(std::integral_constant<int, 2>{});
        auto stage = [&](auto ic) {
            constexpr int I = decltype(ic)::value;
            if constexpr (I + PD < NST) load(std::integral_constant<int, I + PD>{});
            compute(ic);
            __builtin_amdgcn_sched_barrier(0);
        };
        stage(std::integral_constant<int, 0>{}); stage(std::integral_constant<int, 1>{});
        stage(std::integral_constant<int, 2>{}); stage(std::integral_constant<int, 3>{});
        stage(std::integral_constant<int, 4>{}); stage(std::integral_constant<int, 5>{});
        stage(std::integral_constant<int, 6>{}); stage(std::integral_constant<int, 7>{});
        stage(std::integral_constant<int, 8>{}); stage(std::integral_constant<int, 9>{});
    };
    // X(t): online softmax of S -> P (hi, lo fragments).  Q carries `scale`, so S is already in score units; the VALU
    // issue port is what bounds this phase (the partner's MFMAs take 8 of every 32 issue cycles), hence: packed fp32
    // adds / fmas, the key-row bias th folded into the exponent's constant, single-instruction exp2, and the hi/lo
    // split as cvt_pkrtz + (e - hi) + cvt_pkrtz (hi truncated instead of rounded: lo absorbs the difference exactly).
    auto X = [&](const f32x2 (&tw)[8], float th) {
#ifdef CVLM_G64_NOX
        asm volatile("" : "+v"(xh[0]), "+v"(xh[1]), "+v"(xl[0]), "+v"(xl[1]));   // probe: no VALU phase
        return;
#endif
        f32x2 z[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) z[i] = f32x2{s[2 * i], s[2 * i + 1]} + tw[i];
        float mx = fmaxf(z[0].x, z[0].y);
#pragma unroll
        for (int i = 1; i < 8; ++i) mx = fmaxf(fmaxf(mx, z[i].x), z[i].y);
        mx = half_swap_max(mx) + th;
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
        const f32x2 c2 = f32x2{(th - m_new) * LOG2E, (th - m_new) * LOG2E}, l2 = f32x2{LOG2E, LOG2E};
        f32x2 acc = f32x2{0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x2 a = z[i] * l2 + c2;
            z[i] = f32x2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
            acc += z[i];
        }
        l_run = l_run * alpha + (acc.x + acc.y);
        if (!__all(m_new == m_run)) {
            const auto ax = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, alpha), __builtin_bit_cast(unsigned, alpha), false, false);
            const float a0 = __builtin_bit_cast(float, (unsigned)ax[0]), a1 = __builtin_bit_cast(float, (unsigned)ax[1]);
#pragma unroll
            for (int n = 0; n < NDB; ++n) { o[n][0] *= a0; o[n][1] *= a1; }
            osum[0] *= a0; osum[1] *= a1;
        }
        m_run = m_new;
#pragma unroll
        for (int p = 0; p < 4; ++p) {                                // first / last eight values -> the two B operands (see attn_g64pair_kernel)
            const f32x2 v0 = z[p], v1 = z[4 + p];
            unsigned h0, h1;
            if constexpr (PLO) {
                h0 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v0.x, v0.y));
                h1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v1.x, v1.y));
                const unsigned l0 = split_lo_pk(h0, v0.x, v0.y);
                const unsigned l1 = split_lo_pk(h1, v1.x, v1.y);
                const auto rl = __builtin_amdgcn_permlane16_swap(l0, l1, false, false);
                xl[0][p] = (unsigned)rl[0]; xl[1][p] = (unsigned)rl[1];
            } else {                                                 // one fp16 per probability, rounded to nearest
                h0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v0, half2v));
                h1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v1, half2v));
            }
            const auto rh = __builtin_amdgcn_permlane16_swap(h0, h1, false, false);
            xh[0][p] = (unsigned)rh[0]; xh[1][p] = (unsigned)rh[1];
        }
    };

    // ---- every wave runs the same sequence  QK(0) | X(0) Y(0) | X(1) Y(1) | ...  with Y(t) = PV(t), QK(t+1);
    // group B enters it one barrier later and leaves it one barrier earlier, so that in every phase one wave of a
    // SIMD pair is in X and the other in Y.  Global phase p: A runs X(t) at p = 2t, Y(t) at 2t + 1; B one later.
    //   reads:  K(t+1), V(t) by A in phase 2t + 1, by B in phase 2t + 2
    //   DMA:    at the start of its Y(t) a K-carrier (group A) fetches K(t+3) into K(t)'s slot, a V-carrier (group B)
    //           V(t+2) into V(t-1)'s slot -- both last read one phase earlier -- and the batch is retired two of the
    //           carrier's own phases later with vmcnt(DPW) (one younger batch may stay in flight), in front of the
    //           barrier that closes an even global phase; first use is A's Y(t+2) right behind that barrier.
    wait_vm<0>();
    phase_barrier();
    if (grpB) phase_barrier();                                       // B starts one phase late
    QK(0);
    phase_barrier();
    if (trace) tr_pro = wall_clock64() - tr_start;
    int s0 = 0, s1 = 1, s2 = 2;                                      // t % 3, (t + 1) % 3, (t + 2) % 3
    auto tile_phases = [&](auto pz_c, int t) {
        constexpr int PZ = decltype(pz_c)::value;
        if (NSV == 2 && grpB && t >= 1 && t + 1 < NTILE) issue_next((t + 1) & 1);   // V(t+1) into V(t-1)'s slot (read last phase)
        unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        if (trace) c0 = wall_clock64();
        X(twr[PZ], Tq[t / TPR]);
        if (trace) { asm volatile("" ::"v"(xh[0]), "v"(xl[1])); c1 = wall_clock64(); }
        if (!grpB) { if (t + 5 >= NTILE) wait_vm<0>(); else wait_vm<DPW>(); }
        phase_barrier();
        if (trace) c2 = wall_clock64();
        if (!grpB) { if (t + 3 < NTILE) issue_next(s0); }
        else if (NSV == 3) { if (t + 2 < NTILE) issue_next(s2); }
        Y(NSV == 3 ? s0 : (t & 1), s1);                             // the last S (tile NTILE) is computed and dropped
        if (trace) { asm volatile("" ::"v"(s[0]), "v"(o[4][1][0])); c3 = wall_clock64(); }
        if (grpB) { if (NSV == 2 || t + 5 >= NTILE) wait_vm<0>(); else wait_vm<DPW>(); }
        phase_barrier();
        if (trace) { const unsigned long long c4 = wall_clock64(); tr_x += c1 - c0; tr_xb += c2 - c1; tr_y += c3 - c2; tr_yb += c4 - c3; }
        const int n0 = s1; s1 = s2; s2 = s0; s0 = n0;
    };
#pragma unroll 1
    for (int t = 0; t < NTILE; t += TPR) {
        tile_phases(std::integral_constant<int, 0>{}, t);
        tile_phases(std::integral_constant<int, 1>{}, t + 1);
        if constexpr (TPR == 3) tile_phases(std::integral_constant<int, 2>{}, t + 2);
    }
    if (!grpB) phase_barrier();                                      // match B's extra leading barrier

    float invq[2];
    if constexpr (PLO) {
        const float l_tot = half_swap_sum(l_run);
        const float inv = 1.0f / l_tot;
        const auto ix = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, inv), __builtin_bit_cast(unsigned, inv), false, false);
        invq[0] = __builtin_bit_cast(float, (unsigned)ix[0]); invq[1] = __builtin_bit_cast(float, (unsigned)ix[1]);
    } else {                                                         // every row of the ones product holds the sum of query 16 qb + (lane & 15)
        invq[0] = 1.0f / osum[0][0]; invq[1] = 1.0f / osum[1][0];
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qs2 = blockIdx.x * 256 + wave * 32 + 16 * qb + (lane & 15);
        const int64_t orow = ((int64_t)b * S + qs2) * D + head * HD + 4 * (lane >> 4);
        half_t* oh = (half_t*)g.out_hi + orow;
        half_t* ol = g.out_lo ? (half_t*)g.out_lo + orow : nullptr;
#pragma unroll
        for (int n = 0; n < NDB; ++n) {
            half_t h[4], l4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split_h2(o[n][qb][j] * invq[qb], h[j], l4[j]);
            *(half4*)(oh + 16 * n) = half4{h[0], h[1], h[2], h[3]};
            if (ol) *(half4*)(ol + 16 * n) = half4{l4[0], l4[1], l4[2], l4[3]};
        }
    }
    if (trace && lane == 0) {
        unsigned long long* o8 = trace + ((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + wave) * 8;
        o8[0] = tr_pro; o8[1] = tr_x; o8[2] = tr_xb; o8[3] = tr_y; o8[4] = tr_yb; o8[5] = wall_clock64() - tr_start;
    }
}

// =====================================================================================================================
// 64 x 64 map, TWO key tiles (= one key row of the map, 64 keys) per phase.
// Why (tools/trace_attn_g64.py, probe build): in the kernel above a wave's softmax phase X takes 0.96 us per 32-key tile
// for ~90 VALU instructions -- it is latency-bound (one wave per SIMD is in X at a time, nothing hides its dependency
// chains): a probe that ran a second, independent softmax of the same size inside X made the phase only 10-23 % longer.
// So the phase is given twice the work: X normalises 64 keys at once (one running-max update, one rescale), Y runs
// 66 MFMAs back to back (P.V of both tiles, then the scores of the next two).  Half the phases and barriers per key.
// Rings: four K and four V slots (two pairs each); a carrier refills the pair that was read in the previous round, the
// batch has two phases to land and is retired with vmcnt(0) (no younger batch is in flight at that point).
//
// Round 3: P.V on 16x16x32 MFMAs.  The scores keep the 32x32x16 shape (its K = 16 steps cover head_dim 80 exactly; the
// 16-wide shape would pad the contraction to 96), but O^T = V^T.P^T now runs as 16 x 16 output tiles: head_dim 80 = 5 tiles,
// no padding to 96 (60 MFMAs of 16 cycles per 64 keys instead of 36 of 32: -17 % of the P.V matrix cycles), on the shape
// whose bare loop delivers 1.12-1.15x the flops per joule (MI355X_MICROARCH.md) -- the kernel runs at the power cap.
// The P values leave the softmax in the 32x32 accumulator layout (lane = (query, half), 16 keys); one v_permlane16_swap per
// register pair turns them into the two B operands of the 16-wide shape (queries 0-15 / 16-31 of the wave, four key groups of
// eight), V^T is stored in the matching key order by transpose_v_kernel.
template <bool PLO, bool QLO, bool KLO>
__global__ __launch_bounds__(512, 2) void attn_g64pair_kernel(const cvlm_attn_args g, const half_t* __restrict__ vt_hi,
                                                              const half_t* __restrict__ vt_lo) {
    constexpr int L = 64, HD = 80, KS = 5, NDB = 5, KP = 88, KT = 32, LTP = L + 1, S = L * L, NTILE = S / KT, NPAIR = NTILE / 2;
    constexpr int VROWS = 96, VROW_B = KT * 2;
    constexpr int KPL_B = KT * KP * 2, VPL_B = VROWS * VROW_B;
    constexpr int KSLOT_B = 2 * KPL_B, VSLOT_B = 2 * VPL_B, NSLOT = 4;
    constexpr int KPL = KPL_B / 2;
    constexpr int DPW = 3;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Kring = smem;
    unsigned char* Vring = smem + NSLOT * KSLOT_B;
    float* T = (float*)(Vring + NSLOT * VSLOT_B);                    // [256][LTP]  (Tw first, then Th)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned long long* const trace = g_g64_trace;
    const unsigned long long tr_start = trace ? wall_clock64() : 0;
    const unsigned long long tk_start = trace ? __builtin_amdgcn_s_memtime() : 0;
    unsigned long long tr_x = 0, tr_xb = 0, tr_y = 0, tr_yb = 0, tr_pro = 0;
    const bool grpB = wave >= 4;
    const int qc = lane & 31, half = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int D = g.heads * HD;
    const half_t* qkv_hi = (const half_t*)g.qkv_hi;
    const half_t* qkv_lo = (const half_t*)g.qkv_lo;
    const QkvStrides QS = qkv_strides(g.qkv_layout, S, g.B, g.heads, HD);

    // ---- DMA plan (as above): waves 0..3 carry K, waves 4..7 carry V^T; running source pointers
    const half_t* dsrc[DPW];
    int ddst[DPW];
    bool dok[DPW];
    int64_t tile_stride;
    if (!grpB) {
        tile_stride = (int64_t)KT * QS.st;
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            const int i = wave * DPW + j;
            const int pl = i / 6, sub = i - pl * 6;
            const int c = sub * 64 + lane;
            int row = c / 11, ch = c - row * 11;
            dok[j] = row < KT && (KLO || pl == 0);
            if (row >= KT) row = KT - 1;
            if (ch >= 10) ch = 0;
            dsrc[j] = (pl ? qkv_lo : qkv_hi) + qkv_offset(QS, b, row, 1, head) + ch * 8;
            ddst[j] = pl * KPL_B + sub * 1024;
        }
    } else {
        tile_stride = KT;
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            const int i = (wave - 4) * DPW + j;
            const int pl = i / 6, sub = i - pl * 6;
            const int row = sub * 16 + (lane >> 2), pos = lane & 3;
            const int chunk = pos ^ ((row >> 2) & 3);
            const int drow = row < HD ? row : HD - 1;
            dok[j] = true;
            dsrc[j] = (pl ? vt_lo : vt_hi) + (((int64_t)b * g.heads + head) * HD + drow) * S + chunk * 8;
            ddst[j] = pl * VPL_B + sub * 1024;
        }
    }
    auto issue_next = [&](int slot) {                                // this wave's share of the next K or V tile
        unsigned char* base = (grpB ? Vring + slot * VSLOT_B : Kring + slot * KSLOT_B);
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            if (dok[j]) glds16(dsrc[j], base + ddst[j]);
            dsrc[j] += tile_stride;
        }
    };
    // prologue DMA: K pairs 0 and 1 by waves 0..3, V pair 0 by waves 4..7 (they land under the table build below)
    issue_next(0);
    issue_next(1);
    if (!grpB) { issue_next(2); issue_next(3); }

    // ---- queries, rel-pos tables (as above)
    const int qslot = blockIdx.x * 256 + wave * 32 + qc;
    half8 qh[KS], ql[KS];
    {
        const int64_t qo = qkv_offset(QS, b, qslot, 0, head);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qh[ks] = *(const half8*)(qkv_hi + qo + 16 * ks + 8 * half);
            ql[ks] = *(const half8*)(qkv_lo + qo + 16 * ks + 8 * half);
        }
    }
    const int qhh = qslot / L, qww = qslot - qhh * L;
    float* Tq = T + (wave * 32 + qc) * LTP;
    auto build_table = [&](const half_t* Rhi, const half_t* Rlo, int cq) {   // T[q][k] = (Q . R^T)[q][cq - k + L - 1]
#pragma unroll 1
        for (int st = 0; st < (2 * L - 1 + 31) / 32; ++st) {
            floatx16 u;
#pragma unroll
            for (int r = 0; r < 16; ++r) u[r] = 0.f;
            int rr = st * 32 + qc;
            rr = rr < 2 * L - 1 ? rr : 2 * L - 2;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const half8 ah = *(const half8*)(Rhi + rr * HD + 16 * ks + 8 * half);
                const half8 al = *(const half8*)(Rlo + rr * HD + 16 * ks + 8 * half);
                u = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[ks], u, 0, 0, 0);
                u = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[ks], u, 0, 0, 0);
                u = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[ks], u, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = st * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int kidx = cq + (L - 1) - j;
                if (j < 2 * L - 1 && kidx >= 0 && kidx < L) Tq[kidx] = u[r];
            }
        }
    };
    f32x2 twr[2][8];                                                 // Tw[q][32*pz + (r&3) + 8*(r>>2) + 4*half], r = 2i, 2i+1
    build_table((const half_t*)g.relw_hi, (const half_t*)g.relw_lo, qww);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int pz = 0; pz < 2; ++pz)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r0 = 2 * i, r1 = 2 * i + 1;
            twr[pz][i] = f32x2{Tq[32 * pz + (r0 & 3) + 8 * (r0 >> 2) + 4 * half], Tq[32 * pz + (r1 & 3) + 8 * (r1 >> 2) + 4 * half]};
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // each lane re-reads only the row it wrote
    build_table((const half_t*)g.relh_hi, (const half_t*)g.relh_lo, qhh);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            half_t hh, ll;
            split_h2(((float)qh[ks][j] + (float)ql[ks][j]) * g.scale, hh, ll);
            qh[ks][j] = hh; ql[ks][j] = ll;
        }

    // ---- state
    float m_run = -INFINITY, l_run = 0.f;
    floatx16 s[2];
    floatx4 o[NDB][2];                                               // O^T tiles: [16-dim block][query block]: dims 16 db + 4 (lane >> 4) + j, query 16 qb + (lane & 15)
#pragma unroll
    for (int n = 0; n < NDB; ++n)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) o[n][qb] = floatx4{0.f, 0.f, 0.f, 0.f};
    floatx4 osum[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};   // !PLO: row sums of the ROUNDED probabilities (see ones8)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 xh[2][2] = {}, xl[2][2] = {};                              // P^T as B operands of the 16x16x32 shape: [tile of the pair][query block]
    // V^T fragment of this lane: row (lane & 15) of a 16-row block, key group lane >> 4; position swizzled like the DMA image
    const int v_lane_off = (lane & 15) * VROW_B + (((lane >> 4) ^ ((lane >> 2) & 3)) * 16);
    const int k_lane_off = qc * KP + 8 * half;

    auto QK2 = [&](int kslot0) {                                     // scores of the pair in slots kslot0, kslot0 + 1
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const half_t* kr = (const half_t*)(Kring + (kslot0 + e) * KSLOT_B) + k_lane_off;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[e][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const half8 kh = *(const half8*)(kr + 16 * ks);
                s[e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], s[e], 0, 0, 0);
                if constexpr (KLO) {
                    const half8 kl = *(const half8*)(kr + KPL + 16 * ks);
                    s[e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], s[e], 0, 0, 0);
                }
                if constexpr (QLO) s[e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], s[e], 0, 0, 0);
            }
        }
    };
    // Y(q): O^T += V^T.P^T for both tiles of pair q (10 fragment groups: tile x 16-dim block, six 16x16x32 MFMAs each), then the
    // scores of pair q + 1 (10 k-steps, three 32x32x16 MFMAs each), as ONE stream of 20 stages of 96 matrix cycles; the LDS
    // fragments of stage i + PD are requested before the MFMAs of stage i.
    auto Y = [&](int vslot0, int kslot0) {
        // Fragment reads are inline asm with immediate offsets and COUNTED waits: left to hipcc, the stream got an
        // `s_waitcnt lgkmcnt(0)` every third stage, i.e. the full LDS latency (reads of three stages ahead included) was
        // exposed seven times per phase.  LDS returns in order, so "all but the 2 * PD youngest reads" is exactly
        // "stage I's two fragments have arrived".
        const unsigned va = (unsigned)(size_t)(LDS_AS const unsigned char*)(Vring + vslot0 * VSLOT_B) + (unsigned)v_lane_off;
        unsigned ka = (unsigned)(size_t)(LDS_AS const unsigned char*)(Kring + kslot0 * KSLOT_B) + 2u * (unsigned)k_lane_off;
        // KLO == false: the score stages carry TWO k-steps of K's hi plane each (k-steps 2n and 2n + 1 of the pair's ten) instead of one
        // k-step of both planes: every stage still requests exactly two fragments -- the counted waits below stay `2 * ahead`, and no
        // fragment is requested that no MFMA consumes (an asm read whose result is dead leaves its register to the allocator while the
        // read is still in flight: that is what broke the `k` / `v` probe builds of round 5, profiles/r06_probe_kv_lo.log)
        constexpr int PD = CVLM_G64_PD, RS = PD + 1, NPV = 2 * NDB, NKS = KLO ? 10 : 5, NST = NPV + NKS;
        half8 fa[RS], fb[RS];
        auto load = [&](auto ic) {
            constexpr int I = decltype(ic)::value;
            const unsigned a_v = va, a_k = ka;                        // named outside the `if constexpr` so that the lambda captures them
            if constexpr (I < NPV) {
                constexpr int e = I / NDB, db = I % NDB;
                constexpr int off = e * VSLOT_B + (16 * db) * VROW_B;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[I % RS]) : "v"(a_v), "n"(off));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[I % RS]) : "v"(a_v), "n"(off + VPL_B));
            } else if constexpr (KLO) {
                constexpr int e = (I - NPV) / 5, ks = (I - NPV) % 5;
                constexpr int off = e * KSLOT_B + 32 * ks;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[I % RS]) : "v"(a_k), "n"(off));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[I % RS]) : "v"(a_k), "n"(off + KPL_B));
            } else {
                constexpr int i0 = 2 * (I - NPV), i1 = i0 + 1;
                constexpr int off0 = (i0 / 5) * KSLOT_B + 32 * (i0 % 5), off1 = (i1 / 5) * KSLOT_B + 32 * (i1 % 5);
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[I % RS]) : "v"(a_k), "n"(off0));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[I % RS]) : "v"(a_k), "n"(off1));
            }
        };
        auto compute = [&](auto ic) {
            constexpr int I = decltype(ic)::value;
            const half8 a = fa[I % RS], bq = fb[I % RS];
            if constexpr (I < NPV) {
                constexpr int e = I / NDB, db = I % NDB;
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    const half8 bh = __builtin_bit_cast(half8, xh[e][qb]), bl = __builtin_bit_cast(half8, xl[e][qb]);
                    if constexpr (!PLO && db == 0) osum[qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones8(), bh, osum[qb], 0, 0, 0);
                    o[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bh, o[db][qb], 0, 0, 0);
                    o[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bq, bh, o[db][qb], 0, 0, 0);
                    if constexpr (PLO) o[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bl, o[db][qb], 0, 0, 0);
                }
            } else if constexpr (KLO) {
                constexpr int e = (I - NPV) / 5, ks = (I - NPV) % 5;
                if constexpr (ks == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[e][r] = 0.f;
                }
                s[e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qh[ks], s[e], 0, 0, 0);
                s[e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bq, qh[ks], s[e], 0, 0, 0);
                if constexpr (QLO) s[e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, ql[ks], s[e], 0, 0, 0);
            } else {
                static_assert(KLO || !QLO, "one-term scores: K's and Q's hi planes");
                constexpr int i0 = 2 * (I - NPV), i1 = i0 + 1;
                constexpr int e0 = i0 / 5, k0 = i0 % 5, e1 = i1 / 5, k1 = i1 % 5;
                if constexpr (k0 == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[e0][r] = 0.f;
                }
                s[e0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qh[k0], s[e0], 0, 0, 0);
                if constexpr (k1 == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[e1][r] = 0.f;
                }
                s[e1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bq, qh[k1], s[e1], 0, 0, 0);
            }
        };
        load(std::integral_constant<int, 0>{});
        if constexpr (PD >= 2) load(std::integral_constant<int, 1>{});
        if constexpr (PD >= 3) load(std::integral_constant<int, 2>{});
        if constexpr (PD >= 4) load(std::integral_constant<int, 3>{});
        auto stage = [&](auto ic) {
            constexpr int I = decltype(ic)::value;
            constexpr int ahead = (NST - 1 - I) < PD ? (NST - 1 - I) : PD;     // stages whose reads are younger than stage I's
            if constexpr (I + PD < NST) load(std::integral_constant<int, I + PD>{});
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * ahead) : "memory");
            __builtin_amdgcn_sched_barrier(0);
#ifdef CVLM_G64_NOMFMA
            asm volatile("" ::"v"(fa[I % RS]), "v"(fb[I % RS]));                  // probe: fragment reads only
            if (I == NST - 1) { asm volatile("" : "+v"(s[0]), "+v"(s[1])); }
            return;
#endif
            compute(ic);
            __builtin_amdgcn_sched_barrier(0);
        };
        stage(std::integral_constant<int, 0>{}); stage(std::integral_constant<int, 1>{});
        stage(std::integral_constant<int, 2>{}); stage(std::integral_constant<int, 3>{});
        stage(std::integral_constant<int, 4>{}); stage(std::integral_constant<int, 5>{});
        stage(std::integral_constant<int, 6>{}); stage(std::integral_constant<int, 7>{});
        stage(std::integral_constant<int, 8>{}); stage(std::integral_constant<int, 9>{});
        stage(std::integral_constant<int, 10>{}); stage(std::integral_constant<int, 11>{});
        stage(std::integral_constant<int, 12>{}); stage(std::integral_constant<int, 13>{});
        stage(std::integral_constant<int, 14>{});
        if constexpr (NST == 20) {
            stage(std::integral_constant<int, 15>{});
            stage(std::integral_constant<int, 16>{}); stage(std::integral_constant<int, 17>{});
            stage(std::integral_constant<int, 18>{}); stage(std::integral_constant<int, 19>{});
        }
    };
    // X(q): online softmax over the 64 keys of key row q (both tiles share the row bias th; the column bias is per tile)
    auto X = [&](float th) {
#ifdef CVLM_G64_NOX
        asm volatile("" : "+v"(xh[0][0]), "+v"(xh[1][1]), "+v"(xl[0][0]), "+v"(xl[1][1]));   // probe: no VALU phase
        return;
#endif
        f32x2 z[2][8];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int i = 0; i < 8; ++i) z[e][i] = f32x2{s[e][2 * i], s[e][2 * i + 1]} + twr[e][i];
        float mx0 = fmaxf(z[0][0].x, z[0][0].y), mx1 = fmaxf(z[1][0].x, z[1][0].y);
#pragma unroll
        for (int i = 1; i < 8; ++i) {
            mx0 = fmaxf(fmaxf(mx0, z[0][i].x), z[0][i].y);
            mx1 = fmaxf(fmaxf(mx1, z[1][i].x), z[1][i].y);
        }
        const float mx = half_swap_max(fmaxf(mx0, mx1)) + th;
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
        const f32x2 c2 = f32x2{(th - m_new) * LOG2E, (th - m_new) * LOG2E}, l2 = f32x2{LOG2E, LOG2E};
        f32x2 acc0 = f32x2{0.f, 0.f}, acc1 = f32x2{0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x2 a0 = z[0][i] * l2 + c2, a1 = z[1][i] * l2 + c2;
            z[0][i] = f32x2{__builtin_amdgcn_exp2f(a0.x), __builtin_amdgcn_exp2f(a0.y)};
            z[1][i] = f32x2{__builtin_amdgcn_exp2f(a1.x), __builtin_amdgcn_exp2f(a1.y)};
            acc0 += z[0][i];
            acc1 += z[1][i];
        }
        l_run = l_run * alpha + ((acc0.x + acc0.y) + (acc1.x + acc1.y));
        if (!__all(m_new == m_run)) {
            // the O^T tiles hold queries (lane & 15) + 16 qb: this lane's own factor serves one block, the lane 16 away holds the other
            const auto ax = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, alpha), __builtin_bit_cast(unsigned, alpha), false, false);
            const float a0 = __builtin_bit_cast(float, (unsigned)ax[0]), a1 = __builtin_bit_cast(float, (unsigned)ax[1]);
#pragma unroll
            for (int n = 0; n < NDB; ++n) { o[n][0] *= a0; o[n][1] *= a1; }
            osum[0] *= a0; osum[1] *= a1;
        }
        m_run = m_new;
        // P (hi, lo): the first eight values of a tile (keys {0-3, 8-11} + 4 half) and the last eight ({16-19, 24-27} + 4 half)
        // of every lane, register by register through v_permlane16_swap: the first result is the B operand of query block 0
        // (lanes 0-15 / 32-47 keep their first eight, lanes 16-31 / 48-63 receive the last eight of the lane 16 below), the
        // second that of query block 1.  Key order of the 32 k-slots: transpose_v_kernel stores V^T to match.
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const f32x2 v0 = z[e][p], v1 = z[e][4 + p];
                unsigned h0, h1;
                if constexpr (PLO) {
                    h0 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v0.x, v0.y));
                    h1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v1.x, v1.y));
                    const unsigned l0 = split_lo_pk(h0, v0.x, v0.y);
                    const unsigned l1 = split_lo_pk(h1, v1.x, v1.y);
                    const auto rl = __builtin_amdgcn_permlane16_swap(l0, l1, false, false);
                    xl[e][0][p] = (unsigned)rl[0]; xl[e][1][p] = (unsigned)rl[1];
                } else {                                             // one fp16 per probability, rounded to nearest
                    h0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v0, half2v));
                    h1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v1, half2v));
                }
                const auto rh = __builtin_amdgcn_permlane16_swap(h0, h1, false, false);
                xh[e][0][p] = (unsigned)rh[0]; xh[e][1][p] = (unsigned)rh[1];
            }
    };

    // ---- every wave runs  QK(pair 0) | X(0) Y(0) | X(1) Y(1) | ...  with Y(q) = PV(pair q), QK(pair q+1); group B one
    // phase behind group A.  Global phase: A runs X(q) at 2q, Y(q) at 2q + 1; B one later.
    //   reads:  K pair q+1 and V pair q by A in phase 2q + 1, by B in phase 2q + 2; pair q lives in slots 2 (q & 1), +1
    //   DMA:    K carriers (A) fetch K pair q+2 at the start of their Y(q) into the slots of K pair q (last read in phase
    //           2q), wait for it at the end of their X(q+1) (phase 2q + 2): first use is their own Y(q+1) in phase 2q + 3;
    //           V carriers (B) fetch V pair q+1 at the start of their X(q) (phase 2q + 1) into the slots of V pair q-1 (last
    //           read in phase 2q), wait for it at the end of their Y(q) (phase 2q + 2): first use is A's Y(q+1) in 2q + 3.
    wait_vm<0>();
    phase_barrier();
    if (grpB) phase_barrier();                                       // B starts one phase late
    QK2(0);
    phase_barrier();
    if (trace) tr_pro = wall_clock64() - tr_start;
#pragma unroll 1
    for (int q = 0; q < NPAIR; ++q) {
        const int cur = 2 * (q & 1), oth = 2 - cur;                  // slots of pair q / of pairs q - 1 and q + 1
        unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        if (trace) c0 = wall_clock64();
        if (grpB && q + 1 < NPAIR) { issue_next(oth); issue_next(oth + 1); }
        X(Tq[q]);
        if (trace) { asm volatile("" ::"v"(xh[0][0]), "v"(xl[1][1])); c1 = wall_clock64(); }
        if (!grpB) wait_vm<0>();
        phase_barrier();
        if (trace) c2 = wall_clock64();
        if (!grpB && q + 2 < NPAIR) { issue_next(cur); issue_next(cur + 1); }
        Y(cur, oth);                                                 // the last scores (pair NPAIR) are computed and dropped
        if (trace) { asm volatile("" ::"v"(s[0][0]), "v"(s[1][0]), "v"(o[4][1][0])); c3 = wall_clock64(); }
        if (grpB) wait_vm<0>();
        phase_barrier();
        if (trace) { const unsigned long long c4 = wall_clock64(); tr_x += c1 - c0; tr_xb += c2 - c1; tr_y += c3 - c2; tr_yb += c4 - c3; }
    }
    if (!grpB) phase_barrier();                                      // match B's extra leading barrier

    float invq[2];
    if constexpr (PLO) {
        const float l_tot = half_swap_sum(l_run);
        const float inv = 1.0f / l_tot;
        const auto ix = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, inv), __builtin_bit_cast(unsigned, inv), false, false);
        invq[0] = __builtin_bit_cast(float, (unsigned)ix[0]); invq[1] = __builtin_bit_cast(float, (unsigned)ix[1]);
    } else {                                                         // every row of the ones product holds the sum of query 16 qb + (lane & 15)
        invq[0] = 1.0f / osum[0][0]; invq[1] = 1.0f / osum[1][0];
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qs2 = blockIdx.x * 256 + wave * 32 + 16 * qb + (lane & 15);
        const int64_t orow = ((int64_t)b * S + qs2) * D + head * HD + 4 * (lane >> 4);
        half_t* oh = (half_t*)g.out_hi + orow;
        half_t* ol = g.out_lo ? (half_t*)g.out_lo + orow : nullptr;
#pragma unroll
        for (int n = 0; n < NDB; ++n) {
            half_t h[4], l4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split_h2(o[n][qb][j] * invq[qb], h[j], l4[j]);
            *(half4*)(oh + 16 * n) = half4{h[0], h[1], h[2], h[3]};
            if (ol) *(half4*)(ol + 16 * n) = half4{l4[0], l4[1], l4[2], l4[3]};
        }
    }
    if (trace && lane == 0) {
        unsigned long long* o8 = trace + ((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + wave) * 8;
        o8[0] = tr_pro; o8[1] = tr_x; o8[2] = tr_xb; o8[3] = tr_y; o8[4] = tr_yb; o8[5] = wall_clock64() - tr_start;
        o8[6] = __builtin_amdgcn_s_memtime() - tk_start;
    }
}

// V [key][dim] (as the qkv GEMM leaves it) -> V^T [b][head][dim][key'], both planes; 64 keys x 80 dims per workgroup.
// ORDER16: key order of the 16x16x32 P.V operands (both kernels above), else that of 32x32x16 operands (round 2; kept for A/B builds).
template <bool ORDER16>
__global__ __launch_bounds__(256) void transpose_v_kernel(const cvlm_attn_args g, half_t* __restrict__ vt_hi,
                                                          half_t* __restrict__ vt_lo) {
    constexpr int HD = 80, TP = 88;                                  // LDS row pitch in halves
    const int S = g.grid * g.grid;
    __shared__ __attribute__((aligned(16))) half_t tile[2][64 * TP];
    const int tid = threadIdx.x, head = blockIdx.y, b = blockIdx.z, s0 = blockIdx.x * 64;
    const QkvStrides QS = qkv_strides(g.qkv_layout, S, g.B, g.heads, HD);
    const half_t* src[2] = {(const half_t*)g.qkv_hi, (const half_t*)g.qkv_lo};
    for (int u = tid; u < 2 * 64 * 10; u += 256) {
        const int pl = u / 640, r = (u % 640) / 10, ch = u % 10;
        *(half8*)(&tile[pl][r * TP + ch * 8]) = *(const half8*)(src[pl] + qkv_offset(QS, b, s0 + r, 2, head) + ch * 8);
    }
    __syncthreads();
    half_t* dst[2] = {vt_hi, vt_lo};
    for (int u = tid; u < 2 * HD * 8; u += 256) {
        const int pl = u / 640, d = (u % 640) / 8, c8 = u % 8;
        half8 v;
        if (ORDER16) {
            // 32-key tile e = c8 >> 2, key group gq = c8 & 3 of the 16x16x32 B operand: k-slot j of the group is key
            // (j & 3) + 8 (j >> 2) + 16 (gq & 1) + 4 (gq >> 1) -- what the v_permlane16_swap of the 32x32 score layout leaves there
            const int e = c8 >> 2, gq = c8 & 3;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = tile[pl][(32 * e + (j & 3) + 8 * (j >> 2) + 16 * (gq & 1) + 4 * (gq >> 1)) * TP + d];
        } else {
            // keys are stored in the order the P fragments hold them: element j of k-half h of a 16-key step is key
            // (j & 3) + 8 (j >> 2) + 4 h (the row map of the 32x32 accumulator, which becomes the B operand unchanged)
            const int g16 = c8 >> 1, h = c8 & 1;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = tile[pl][(16 * g16 + (j & 3) + 8 * (j >> 2) + 4 * h) * TP + d];
        }
        *(half8*)(dst[pl] + (((int64_t)b * g.heads + head) * HD + d) * S + s0 + c8 * 8) = v;
    }
}

}  // namespace

// ---- V^T workspace: caller-owned (cvlm_attn_args.workspace, size from cvlm_attention_workspace_bytes())
int64_t cvlm_attention_global64_pp_workspace_bytes(const cvlm_attn_args& g) {
    const bool splits = (g.split_qk == 3 && g.split_pv == 3) || (g.split_qk == 2 && g.split_pv == 2) || (g.split_qk == 1 && g.split_pv == 2);
    if (g.mode != 1 || g.hd != 80 || !splits || (g.grid != 64 && g.grid != 96)) return 0;
    return (int64_t)2 * g.B * g.heads * 80 * g.grid * g.grid * (int64_t)sizeof(half_t);
}

// Probe hook (not part of include/cvlm.h).
extern "C" int cvlm_debug_set_attn_g64_trace(void* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_g64_trace), &buf, sizeof(buf));
}

// split 3/3 (PLO = QLO = KLO = true), 2/2 (K's lo plane only) or 1/2 (no lo plane in the scores) form of cvlm_attention_global64()
template <int L, bool PLO, bool QLO, bool KLO>
static int launch_pp(const cvlm_attn_args& g, hipStream_t s) {
    constexpr int S = L * L;
    constexpr int smem = 3 * (2 * 5632) + (L == 64 ? 3 : 2) * (2 * 6144) + 256 * (L + 1) * 4;
    const size_t plane = (size_t)g.B * g.heads * 80 * S;
    if (!g.workspace || g.workspace_bytes < cvlm_attention_global64_pp_workspace_bytes(g)) return CVLM_E_WORKSPACE;
    half_t* vt = (half_t*)g.workspace;
    hipLaunchKernelGGL(transpose_v_kernel<true>, dim3(S / 64, g.heads, g.B), dim3(256), 0, s, g, vt, vt + plane);
    CVLM_CHECK_LAUNCH();
    if constexpr (L == 64) {                                          // 64 x 64 map: a key ROW of the map per phase (round 2)
        constexpr int smem2 = 4 * (2 * 5632) + 4 * (2 * 6144) + 256 * (L + 1) * 4;
        static bool attr2[16] = {};
        if (cvlm_first_on_device(attr2))
            (void)hipFuncSetAttribute((const void*)attn_g64pair_kernel<PLO, QLO, KLO>, hipFuncAttributeMaxDynamicSharedMemorySize, smem2);
        hipLaunchKernelGGL((attn_g64pair_kernel<PLO, QLO, KLO>), dim3(S / 256, g.heads, g.B), dim3(512), smem2, s, g, (const half_t*)vt,
                           (const half_t*)(vt + plane));
    } else {                                                          // 96 x 96 map: one 32-key tile per phase
        static bool attr[16] = {};
        if (cvlm_first_on_device(attr))
            (void)hipFuncSetAttribute((const void*)attn_g64pp_kernel<L, PLO, QLO, KLO>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        hipLaunchKernelGGL((attn_g64pp_kernel<L, PLO, QLO, KLO>), dim3(S / 256, g.heads, g.B), dim3(512), smem, s, g, (const half_t*)vt,
                           (const half_t*)(vt + plane));
    }
    CVLM_CHECK_LAUNCH();
    return 0;
}

// global attention on a 64x64 or 96x96 token map: split 3/3 (hi/lo operands on both sides of both products), 2/2, or 1/2
int cvlm_attention_global64_pp(const cvlm_attn_args& g, hipStream_t s) {
    const bool full = g.split_qk == 3 && g.split_pv == 3, two = g.split_qk == 2 && g.split_pv == 2, one = g.split_qk == 1 && g.split_pv == 2;
    if (!full && !two && !one) return CVLM_E_UNSUPPORTED;
    if (g.grid == 64) return full ? launch_pp<64, true, true, true>(g, s) : two ? launch_pp<64, false, false, true>(g, s) : launch_pp<64, false, false, false>(g, s);
    if (g.grid == 96) return full ? launch_pp<96, true, true, true>(g, s) : two ? launch_pp<96, false, false, true>(g, s) : launch_pp<96, false, false, false>(g, s);
    return CVLM_E_UNSUPPORTED;
}
