// ViT-H *window* attention (14x14 windows, head_dim 80, exact mode), producer / consumer form (round 3).
// image_encoder.py:488-504, 507-553, 589-625.  Query on the lane for S^T and O^T; pad tokens = qkv bias rows read by source
// address, never stored.  The rel-pos bias is folded into the QK^T contraction instead of being gathered per score element:
//     Q_aug = [ q (80) | Th[q][0..13]/scale | Tw[q][0..13]/scale | 0 0 0 0 ]      (112 = 7 k-steps of 16)
//     K_aug = [ k (80) | onehot14(kh)       | onehot14(kw)       | 0 0 0 0 ]
// so S^T = K_aug . Q_aug^T already contains (q.k + bias/scale).  The one-hot block is the same for every window and head: a
// 224 x 32 constant image in the code object, copied to LDS by 14 DMA instructions; Th / Tw are formed once per pair with MFMA
// (U = Q . R^T, 27 table rows).  (Rounds 1-4 also carried the kernel this one replaced -- two 4-wave workgroups per (window,
// head) pair, each streaming all keys and issuing its own DMA: "the older form" below; it served the non-parity precisions and
// left the build in round 5, which now take the generic kernel of attention.hip.)  Division of labour:
//
//   * ONE workgroup per CU, persistent over a list of (window, head) pairs; 8 waves:
//       waves 0..6  CONSUMERS, 32 query slots each = 224 >= 196 queries of the pair: K / V tiles are read ONCE per pair
//                   (the older form ran two 4-wave workgroups per pair, each streaming all keys);
//       wave  7     PRODUCER: issues every LDS-DMA instruction of the workgroup -- rel-pos tables and one-hot block once, then
//                   the K / V tiles of pair after pair as one endless stream of 32-key tiles, two tiles ahead of the consumers.
//     In the older form a third of a tile's time went into ISSUING the five DMA instructions a wave owes per tile (row-offset
//     reads, 64-bit address arithmetic, m0 set-up, ~125 ns each), and the first tiles of a workgroup land while it waits: here
//     the consumers issue no vector-memory instruction inside the key loop at all, and the first tiles of pair n + 1 arrive
//     under the last tiles of pair n (the producer does not know item boundaries, only tile numbers).
//   * Rings: three K slots, three V slots (10 KB each, both planes).  Step g (global tile number): consumers form the scores of
//     tile g + 1 and P.V of tile g; the producer, behind the same barrier, requests K(g + 3) and V(g + 2) -- the slots of K(g) and
//     V(g - 1), last read in step g - 1 -- and waits for everything older (vmcnt = size of the newest batch) before it arrives at
//     the next barrier, which thereby guarantees K(g + 2), V(g + 1).
//   * Row offsets of the 224 key slots of a pair are the producer's private LDS table (double-buffered by pair parity).
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "../../include/cvlm.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int OFF>
__device__ __forceinline__ half4 lds_read_tr16(unsigned lds_addr) {
    half4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "n"(OFF));
    return r;
}
__device__ __forceinline__ void lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void step_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// one-hot block of K_aug (file header): row = key slot, 32 columns = onehot14(kh) | 0 0 | onehot14(kw) | 0 0,
// the four 16-byte chunks of a row stored at chunk ^ ((row >> 2) & 3)
struct OneHotImage2 { half_t v[224 * 32]; };
constexpr OneHotImage2 make_onehot_image2() {
    OneHotImage2 im{};
    for (int s = 0; s < 224; ++s)
        for (int c = 0; c < 32; ++c) {
            const int kh = s / 14, kw = s % 14;
            const bool one = s < 196 && (c == kh || c == 16 + kw);
            const int chunk = (c >> 3) ^ ((s >> 2) & 3);
            im.v[s * 32 + chunk * 8 + (c & 7)] = one ? (half_t)1.0f : (half_t)0.0f;
        }
    return im;
}
__device__ const OneHotImage2 g_onehot2 = make_onehot_image2();

#ifdef CVLM_PROBES
// probe builds (tools/trace_attn_win2.py): per workgroup and consumer wave, the time in the three stages of a pair, summed over its pairs:
// [U = Q.R^T + scatter | seven key tiles | output | pairs | first stamp | last stamp | of the tiles: waiting at the step barriers | HW_ID],
// 100-MHz wall-clock ticks
__device__ unsigned long long* g_win2_trace = nullptr;
#define WIN2_STAMP(x) do { if (trace) { __builtin_amdgcn_sched_barrier(0); x = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define WIN2_STAMP(x) do { } while (0)
#endif

// PLO / QLO / KLO: as in attention_g64pp.hip (split 3: all true; split 2: one fp16 per probability, Q's hi plane; split_qk 1: K's hi plane
// too -- its lo plane is then not fetched: the producer's K batches are one plane)
template <bool PLO, bool QLO, bool KLO>
__global__ __launch_bounds__(512, 1) void attn_win14p_kernel(const cvlm_attn_args g, const int nwx, const int npairs) {
    constexpr int HD = 80, KS = 5, NDB = 5, CPR = 10, KP = 80, VP = 80, L = 14, S_SEQ = 196;
    constexpr int KT = 32, NKT = 7, NCW = 7;                        // 7 key tiles, 7 consumer waves
    constexpr int PLANE_B = 5120, SLOT_B = 2 * PLANE_B, IPP = 5;    // 32 rows x 160 B per plane = five 1-KiB DMA pieces
    constexpr int OFF_K = 0, OFF_V = 3 * SLOT_B, OFF_T = 6 * SLOT_B;
    constexpr int RT_B = 27 * HD * 2, RS_PIECES = (4 * RT_B + 1023) / 1024, RS_B = RS_PIECES * 1024;   // rel-pos tables: 17 pieces
    constexpr int OFF_OH = OFF_T + RS_B, OH_B = 224 * 32 * 2;
    constexpr int OFF_TOK = OFF_OH + OH_B, TOK_B = 2 * 224 * 8;     // per pair parity: [K | V][224] row offsets
    constexpr int TP = 17, OFF_TAUG = OFF_TOK + 2 * TOK_B;          // [224 queries][14 values, 2 zeros, 1 dump slot]

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qc = lane & 31, half = lane >> 5;
    const int D = g.heads * HD;
    const half_t* qkv_hi = (const half_t*)g.qkv_hi;
    const half_t* qkv_lo = (const half_t*)g.qkv_lo;
    const half_t* pad_hi = (const half_t*)g.pad_hi;
    const half_t* pad_lo = (const half_t*)g.pad_lo;
    const int64_t qkv_plane = qkv_lo - qkv_hi, pad_plane = pad_lo - pad_hi;   // lo-plane displacement (elements)
    const int nwin = nwx * nwx;
    const int SI = g.grid * g.grid;
    const QkvStrides QS = qkv_strides(g.qkv_layout, SI, g.B, g.heads, HD);
    // this workgroup's pairs: blockIdx.x, + gridDim.x, ...; T = its number of key tiles
    const int my_items = blockIdx.x < npairs ? (npairs - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int T = my_items * NKT;
    auto pair_of = [&](int k) -> int { return (int)blockIdx.x + k * (int)gridDim.x; };
    auto token_of = [&](int pair, int slot) -> int {
        const int seq = pair / g.heads;
        const int w = seq % nwin;
        const int wy = w / nwx, wx = w - wy * nwx;
        const int iy = slot / L, ix = slot - iy * L;
        const int y = wy * L + iy, x = wx * L + ix;
        if (y >= g.grid || x >= g.grid) return -1;
        return y * g.grid + x;
    };

    if (wave == NCW) {
        // ================================================= PRODUCER =================================================
        long long* tokoff = (long long*)(smem + OFF_TOK);
        auto fill_tokoff = [&](int k) {                              // row offsets of pair k's 224 key slots -> buffer k & 1
            const int pair = pair_of(k);
            const int head = pair % g.heads, b = (pair / g.heads) / nwin;
            long long* tk = tokoff + (k & 1) * (TOK_B / 8);
            const long long pad_delta = pad_hi - qkv_hi;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int sl = lane + 64 * r;
                if (sl < 224) {
                    const int tok = token_of(pair, sl < S_SEQ ? sl : S_SEQ - 1);
                    const long long ko = tok < 0 ? pad_delta + D + head * HD : (long long)qkv_offset(QS, b, tok, 1, head);
                    tk[sl] = ko | (tok < 0 ? 1 : 0);
                    tk[224 + sl] = (ko + (tok < 0 ? (long long)D : (long long)QS.sop)) | (tok < 0 ? 1 : 0);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // wave-private table: LDS keeps this wave's accesses in order
        };
        int row_s[IPP], ch_s[IPP];                                    // this lane's row / 16-byte chunk in piece `sub` of a plane image
#pragma unroll
        for (int sub = 0; sub < IPP; ++sub) {
            const int c = sub * 64 + lane;
            row_s[sub] = c / CPR;
            ch_s[sub] = c - row_s[sub] * CPR;
        }
        auto issue_tile = [&](int op, int gt) {                       // key tile gt of the stream (pair gt / 7, tile gt % 7) of K (0) or V (1)
            const int k = gt / NKT, t = gt - k * NKT;
            const long long* tk = tokoff + (k & 1) * (TOK_B / 8) + op * 224 + t * KT;
            unsigned char* dst = smem + (op ? OFF_V : OFF_K) + (gt % 3) * SLOT_B;
            const half_t* src[IPP][2];
#pragma unroll
            for (int sub = 0; sub < IPP; ++sub) {
                const long long ko = tk[row_s[sub]];
                const long long disp = (ko & 1) ? pad_plane : qkv_plane;
                src[sub][0] = qkv_hi + ((ko & ~7ll) + ch_s[sub] * 8);
                src[sub][1] = src[sub][0] + disp;
            }
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                if (!KLO && pl == 1 && op == 0) continue;             // K's lo plane is not multiplied: not fetched (op is a literal at every call)
#pragma unroll
                for (int sub = 0; sub < IPP; ++sub) glds16(src[sub][pl], dst + pl * PLANE_B + sub * 1024);
            }
        };
        // once: rel-pos tables (4 planes back to back, 17 pieces) and the one-hot block (14 pieces)
        {
            const unsigned char* t0 = (const unsigned char*)g.relh_hi;
            const unsigned char* t1 = (const unsigned char*)g.relh_lo;
            const unsigned char* t2 = (const unsigned char*)g.relw_hi;
            const unsigned char* t3 = (const unsigned char*)g.relw_lo;
#pragma unroll 1
            for (int i = 0; i < RS_PIECES; ++i) {
                int byte = i * 1024 + lane * 16;
                if (byte >= 4 * RT_B) byte = 0;                       // tail of the last piece: lands in the slack behind the image
                const int tbl = byte / RT_B, off = byte - tbl * RT_B;
                const unsigned char* src = (tbl == 0 ? t0 : (tbl == 1 ? t1 : (tbl == 2 ? t2 : t3))) + off;
                glds16(src, smem + OFF_T + i * 1024);
            }
            const unsigned char* oh = (const unsigned char*)g_onehot2.v;
#pragma unroll 1
            for (int i = 0; i < OH_B / 1024; ++i) glds16(oh + i * 1024 + lane * 16, smem + OFF_OH + i * 1024);
        }
        if (T > 0) {
            fill_tokoff(0);
            issue_tile(0, 0);
            if (T > 1) issue_tile(0, 1);
            if (T > 2) issue_tile(0, 2);
            issue_tile(1, 0);
            if (T > 1) issue_tile(1, 1);
        }
        wait_vm<0>();
        step_barrier();                                               // B_start: tables, one-hot block, K(0..2), V(0..1) are in LDS
#pragma unroll 1
        for (int gt = 0; gt < T; ++gt) {
            step_barrier();                                           // B_gt: every consumer has left step gt - 1
            const int kn = gt + 3, vn = gt + 2;
            if (kn < T && kn % NKT == 0) fill_tokoff(kn / NKT);       // the stream enters a new pair with its first K tile
#if defined(CVLM_WIN2_PROBE) && CVLM_WIN2_PROBE == 2
            const bool ik = false, iv = false;                        // probe: no DMA in the loop (consumers read stale tiles)
#else
            const bool ik = kn < T, iv = vn < T;
#endif
            if (ik) issue_tile(0, kn);
            if (iv) issue_tile(1, vn);
            // everything but the batch just issued has landed -> K(gt + 2), V(gt + 1) are there when we arrive at B_(gt+1)
            constexpr int KB = (KLO ? 2 : 1) * IPP, VB = 2 * IPP;      // DMA instructions of a K / V batch
            if (ik && iv) wait_vm<KB + VB>();
            else if (ik) wait_vm<KB>();
            else if (iv) wait_vm<VB>();
            else wait_vm<0>();
        }
        return;
    }

    // ===================================================== CONSUMERS =====================================================
    float* Taug = (float*)(smem + OFF_TAUG);
    const half_t* OH = (const half_t*)(smem + OFF_OH);
    auto k_slot = [&](int gt) -> const unsigned char* { return smem + OFF_K + (gt % 3) * SLOT_B; };
    auto v_slot = [&](int gt) -> const unsigned char* { return smem + OFF_V + (gt % 3) * SLOT_B; };
    const int tg = lane >> 4, ti = lane & 15;
    // V^T fragment of the 16x16x32 shape by transposing reads: key group tg holds k-slots {a..a+3, a+8..a+11}, a = 16 (tg & 1) +
    // 4 (tg >> 1) -- the key order v_permlane16_swap leaves in the P operands (attention_g64pp.hip); a 16-lane group reads 4 keys
    // x 16 dims and each lane receives its dim (lane & 15) for those 4 keys
    const int v_lane_off = (16 * (tg & 1) + 4 * (tg >> 1) + (ti >> 2)) * VP + 4 * (ti & 3);
    typedef std::integral_constant<bool, false> no_c;
    typedef std::integral_constant<bool, true> yes_c;
    bool first = true;
    const int qslot = wave * 32 + qc;
    const bool qvalid = qslot < S_SEQ;
    const int qs = qvalid ? qslot : S_SEQ - 1;
    // The query rows of the NEXT pair are requested at the top of a pair's last key tile: from there on the augmented query
    // fragments are dead (the last scores were formed one tile earlier), so the 40 registers are free, and the loads' latency
    // (~1.5 us of a ~20-us pair when asked for at the top of the pair) passes under that tile and the output stage.
    half8 qn_h[KS], qn_l[KS];
    auto load_q = [&](int pair) {
        const int head = pair % g.heads, b = (pair / g.heads) / nwin;
        const int qtok = token_of(pair, qs);
        const int64_t qo = qtok < 0 ? (int64_t)head * HD : qkv_offset(QS, b, qtok, 0, head);
        const half_t* bh = (qtok < 0 ? pad_hi : qkv_hi) + qo;
        const half_t* bl = bh + (qtok < 0 ? pad_plane : qkv_plane);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qn_h[ks] = *(const half8*)(bh + 16 * ks + 8 * half);
            qn_l[ks] = *(const half8*)(bl + 16 * ks + 8 * half);
        }
    };
    if (my_items > 0) load_q(pair_of(0));
#ifdef CVLM_PROBES
    unsigned long long* const trace = g_win2_trace;
    unsigned long long ta = 0, tb = 0, tc = 0, td = 0, acc_u = 0, acc_t = 0, acc_o = 0, acc_bar = 0, t_first = 0;
#endif

#pragma unroll 1
    for (int k = 0; k < my_items; ++k) {
        const int pair = pair_of(k);
        const int head = pair % g.heads, b = (pair / g.heads) / nwin;
        const int gt0 = k * NKT;
        half8 qh[KS + 2], ql[KS + 2];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { qh[ks] = qn_h[ks]; ql[ks] = qn_l[ks]; }
        if (first) { step_barrier(); first = false; }                // B_start (once): tables and one-hot block are in LDS
        WIN2_STAMP(ta);
        // ---- Th / Tw for this query: U = Q . R^T (27 rows -> one 32-row MFMA tile per table), scattered through Taug
        {
            const int qhh = qs / L, qww = qs - qhh * L;
            float* Tq = Taug + (wave * 32 + qc) * TP;
            if (half == 0) { Tq[14] = 0.f; Tq[15] = 0.f; }
            const int rr = qc < 27 ? qc : 26;
            floatx16 u[2];
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                const unsigned char* rt = smem + OFF_T + (2 * tb) * RT_B + rr * (HD * 2) + 16 * half;
#pragma unroll
                for (int r = 0; r < 16; ++r) u[tb][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const half8 rh = *(const half8*)(rt + 32 * ks);
                    const half8 rl = *(const half8*)(rt + RT_B + 32 * ks);
                    u[tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rh, qh[ks], u[tb], 0, 0, 0);
                    u[tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rl, qh[ks], u[tb], 0, 0, 0);
                    u[tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rh, ql[ks], u[tb], 0, 0, 0);
                }
            }
            // the scores use q * scale (image_encoder.py:496), the rel-pos tables q itself (:497-500).  scale == 1: the caller folded
            // the factor into the q rows of the qkv projection (and its inverse into the tables): the planes are used as they are
            if (g.scale != 1.0f) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        half_t hh, ll;
                        split_h2(((float)qh[ks][j] + (float)ql[ks][j]) * g.scale, hh, ll);
                        qh[ks][j] = hh;
                        ql[ks][j] = ll;
                    }
            }
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                int cq = tb ? qww : qhh;
                asm volatile("" : "+v"(cq), "+v"(u[tb][0]));          // keep the scatter's index arithmetic behind the MFMAs (register pressure)
#pragma unroll
                for (int r = 0; r < 16; ++r) {                        // branch-free scatter: rows that do not exist land in the dump slot
                    const int j = (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int kidx = cq + L - 1 - j;
                    const bool ok = j < 27 && kidx >= 0 && kidx < L;
                    Tq[ok ? kidx : 16] = u[tb][r];
                }
                __builtin_amdgcn_wave_barrier();                      // rows are wave-private: LDS keeps a wave's accesses in order
                {
                    typedef unsigned u32x4_s __attribute__((ext_vector_type(4)));
                    u32x4_s h4, l4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        unsigned a, b;
                        split_h2_pk(Tq[8 * half + 2 * j], Tq[8 * half + 2 * j + 1], a, b);
                        h4[j] = a; l4[j] = b;
                    }
                    qh[KS + tb] = __builtin_bit_cast(half8, h4);
                    ql[KS + tb] = __builtin_bit_cast(half8, l4);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }

        WIN2_STAMP(tb);
        float m_run = -INFINITY, l_run = 0.f;
        // O^T as 16 x 16 tiles of the 16x16x32 MFMA (as in attention_g64pp.hip): [16-dim block][query block], this lane holds dims
        // 16 db + 4 (lane >> 4) + j of query 16 qb + (lane & 15): head_dim 80 = 5 blocks, no padding to 96 (15 instead of 18
        // matrix units per key tile), on the cheaper MFMA shape
        floatx4 o[NDB][2];
#pragma unroll
        for (int n = 0; n < NDB; ++n)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) o[n][qb] = floatx4{0.f, 0.f, 0.f, 0.f};
        floatx4 osum[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};   // !PLO: sums of the rounded probabilities (ones . P^T, as in attention_g64pp.hip)

        // S^T tile of key tile t of this pair (stream tile gt0 + t): 15 + 4 (bias) MFMAs
        auto scores = [&](int t, auto last_c) -> floatx16 {
            floatx16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const half_t* kr = (const half_t*)k_slot(gt0 + t) + qc * KP + 8 * half;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const half8 kh = *(const half8*)(kr + 16 * ks);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], s, 0, 0, 0);
                if constexpr (KLO) {
                    const half8 kl = *(const half8*)(kr + PLANE_B / 2 + 16 * ks);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], s, 0, 0, 0);
                }
                if constexpr (QLO) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], s, 0, 0, 0);
            }
            const int row = t * KT + qc;
            const half_t* ohr = OH + row * 32;
#pragma unroll
            for (int a = 0; a < 2; ++a) {                             // bias: exact one-hot rows x (T/scale) hi + lo
                const half8 oh8 = *(const half8*)(ohr + 8 * ((half + 2 * a) ^ ((row >> 2) & 3)));
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(oh8, qh[KS + a], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(oh8, ql[KS + a], s, 0, 0, 0);
            }
            if (decltype(last_c)::value) {                            // only the last tile holds slots beyond the 196 keys
                const int b0 = t * KT + 4 * half;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (b0 + (r & 3) + 8 * (r >> 2) >= S_SEQ) s[r] = -INFINITY;
            }
            return s;
        };
        // K(gt0) is in LDS: B_(gt0 - 1) of the previous pair (or B_start) guaranteed it, and its slot is not written again before
        // the batch behind B_(gt0)
        floatx16 s_cur = scores(0, no_c{});

        auto tile = [&](int t, auto has_next_c, auto next_last_c) {
            constexpr bool HAS_NEXT = decltype(has_next_c)::value;
#ifdef CVLM_PROBES
            unsigned long long te = 0, tf = 0;
            WIN2_STAMP(te);
#endif
            step_barrier();                                           // B_(gt0 + t): K(gt0 + t + 1) and V(gt0 + t) have landed
#ifdef CVLM_PROBES
            WIN2_STAMP(tf);
            if (trace) acc_bar += tf - te;
#endif
#if defined(CVLM_WIN2_PROBE) && CVLM_WIN2_PROBE == 1
            return;                                                   // probe: consumers only keep step with the producer
#endif
            floatx16 s_next;
            if (HAS_NEXT) s_next = scores(t + 1, next_last_c);
            const unsigned vaddr = (unsigned)(size_t)(LDS_AS const unsigned char*)v_slot(gt0 + t) + 2u * (unsigned)v_lane_off;
            // V^T fragments of dim block DB: [plane] = two transposed 8-byte reads (keys a.. and a + 8..), immediate offsets
            auto read_v = [&](auto db_c, half4 (&v0)[2], half4 (&v1)[2]) {
                constexpr int DB = decltype(db_c)::value;
                v0[0] = lds_read_tr16<2 * (16 * DB)>(vaddr);                 v1[0] = lds_read_tr16<2 * (16 * DB + 8 * VP)>(vaddr);
                v0[1] = lds_read_tr16<PLANE_B + 2 * (16 * DB)>(vaddr);       v1[1] = lds_read_tr16<PLANE_B + 2 * (16 * DB + 8 * VP)>(vaddr);
            };
            half4 va0[3][2], va1[3][2], vb0[2][2], vb1[2][2];
            read_v(std::integral_constant<int, 0>{}, va0[0], va1[0]);   // dim blocks 0..2: in flight under the softmax
            read_v(std::integral_constant<int, 1>{}, va0[1], va1[1]);
            read_v(std::integral_constant<int, 2>{}, va0[2], va1[2]);
            const floatx16 s = s_cur;
            float mx = fmaxf(s[0], s[1]);
#pragma unroll
            for (int r = 2; r < 16; r += 2) mx = fmaxf(fmaxf(mx, s[r]), s[r + 1]);
            mx = half_swap_max(mx);
            // The reference point of the exponentials moves only when some query of the wave finds a score more than TAU above its
            // own: until then P = exp(s - m_run) may exceed 1 (by at most e^TAU = 148, nothing to an fp16 hi / lo pair or to the f32
            // sums) and the 40 accumulator values, the running sum and their factor need no touching -- 24 VALU instructions of a
            // tile's ~150, and on this part VALU instructions are not hidden behind the MFMAs: a SIMD runs the one or the other
            // (profiles/r04_mfma_valu_overlap.log).  Same quotient O / l; the rounding points move by the common factor.
            constexpr float TAU = 5.0f;
            if (__builtin_amdgcn_ballot_w64(mx > m_run + TAU) != 0) {     // wave-uniform; the first tile of a pair always (m_run = -inf)
                const float m_new = fmaxf(m_run, mx);
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
                l_run *= alpha;
                // the O^T tiles hold queries (lane & 15) + 16 qb: this lane's own factor serves one block, the lane 16 away holds the other
                const auto ax = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, alpha), __builtin_bit_cast(unsigned, alpha), false, false);
                const float a0 = __builtin_bit_cast(float, (unsigned)ax[0]), a1 = __builtin_bit_cast(float, (unsigned)ax[1]);
#pragma unroll
                for (int n = 0; n < NDB; ++n) { o[n][0] *= a0; o[n][1] *= a1; }
                osum[0] *= a0; osum[1] *= a1;
                m_run = m_new;
            }
            const f32x2 c2 = f32x2{-m_run * LOG2E, -m_run * LOG2E}, l2 = f32x2{LOG2E, LOG2E};
            f32x2 z[8], acc = f32x2{0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x2 a = f32x2{s[2 * i], s[2 * i + 1]} * l2 + c2;
                z[i] = f32x2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
                acc += z[i];
            }
            l_run += acc.x + acc.y;
            // P (hi truncated by cvt_pkrtz, lo = e - hi: exact remainder): first / last eight values of the lane, register by register
            // through v_permlane16_swap -> the B operands of query blocks 0 and 1
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 xh[2], xl[2];
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2) {
                const f32x2 e0 = z[p2], e1 = z[4 + p2];
                unsigned h0, h1;
                if constexpr (PLO) {
                    h0 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(e0.x, e0.y));
                    h1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(e1.x, e1.y));
                    const half2v f0 = __builtin_bit_cast(half2v, h0), f1 = __builtin_bit_cast(half2v, h1);
                    const unsigned l0 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(e0.x - (float)f0[0], e0.y - (float)f0[1]));
                    const unsigned l1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(e1.x - (float)f1[0], e1.y - (float)f1[1]));
                    const auto rl = __builtin_amdgcn_permlane16_swap(l0, l1, false, false);
                    xl[0][p2] = (unsigned)rl[0]; xl[1][p2] = (unsigned)rl[1];
                } else {                                             // one fp16 per probability, rounded to nearest; denominator: osum
                    h0 = __builtin_bit_cast(unsigned, __builtin_convertvector(e0, half2v));
                    h1 = __builtin_bit_cast(unsigned, __builtin_convertvector(e1, half2v));
                }
                const auto rh = __builtin_amdgcn_permlane16_swap(h0, h1, false, false);
                xh[0][p2] = (unsigned)rh[0]; xh[1][p2] = (unsigned)rh[1];
            }
            auto pv = [&](int db, half4 (&v0)[2], half4 (&v1)[2]) {
                const half8 vh = half8{v0[0][0], v0[0][1], v0[0][2], v0[0][3], v1[0][0], v1[0][1], v1[0][2], v1[0][3]};
                const half8 vl = half8{v0[1][0], v0[1][1], v0[1][2], v0[1][3], v1[1][0], v1[1][1], v1[1][2], v1[1][3]};
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    const half8 bh = __builtin_bit_cast(half8, xh[qb]), bl = __builtin_bit_cast(half8, xl[qb]);
                    if (!PLO && db == 0) {
                        const half_t one = (half_t)1.0f;
                        osum[qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(half8{one, one, one, one, one, one, one, one}, bh, osum[qb], 0, 0, 0);
                    }
                    o[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, bh, o[db][qb], 0, 0, 0);
                    o[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, bh, o[db][qb], 0, 0, 0);
                    if constexpr (PLO) o[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, bl, o[db][qb], 0, 0, 0);
                }
            };
            lds_wait();                                               // blocks 0..2 (and everything older) are in registers
            read_v(std::integral_constant<int, 3>{}, vb0[0], vb1[0]);   // blocks 3, 4: in flight under the first 18 MFMAs
            read_v(std::integral_constant<int, 4>{}, vb0[1], vb1[1]);
            pv(0, va0[0], va1[0]);
            pv(1, va0[1], va1[1]);
            pv(2, va0[2], va1[2]);
            lds_wait();
            pv(3, vb0[0], vb1[0]);
            pv(4, vb0[1], vb1[1]);
            if (HAS_NEXT) s_cur = s_next;
        };
#pragma unroll 1
        for (int t = 0; t < NKT - 2; ++t) tile(t, yes_c{}, no_c{});
        tile(NKT - 2, yes_c{}, yes_c{});
        if (k + 1 < my_items) load_q(pair_of(k + 1));
        tile(NKT - 1, no_c{}, no_c{});
        WIN2_STAMP(tc);

        // ---- output.  A lane holds dims 16 db + 4 g .. + 3 of queries (lane & 15) + 16 qb; one v_permlane16_swap per register
        // pair hands a neighbouring 4-dim piece across (even g: the next four dims of block db from lane + 16; odd g: the four
        // dims below of block db + 1 from lane - 16), so blocks (0, 1) and (2, 3) leave in 16-byte stores, block 4 in 8-byte ones
        float invq[2];
        if constexpr (PLO) {
            const float l_tot = half_swap_sum(l_run);
            const float inv = 1.0f / l_tot;
            const auto ix = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, inv), __builtin_bit_cast(unsigned, inv), false, false);
            invq[0] = __builtin_bit_cast(float, (unsigned)ix[0]); invq[1] = __builtin_bit_cast(float, (unsigned)ix[1]);
        } else {                                                      // every row of the ones product holds the sum of query 16 qb + (lane & 15)
            invq[0] = 1.0f / osum[0][0]; invq[1] = 1.0f / osum[1][0];
        }
        const int g4 = lane >> 4;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int qs2 = wave * 32 + 16 * qb + (lane & 15);
            const int tok2 = token_of(pair, qs2 < S_SEQ ? qs2 : S_SEQ - 1);
            const bool ok2 = qs2 < S_SEQ && tok2 >= 0;
            const int64_t orow = ((int64_t)b * SI + (ok2 ? tok2 : 0)) * D + head * HD;
            half_t* oh = (half_t*)g.out_hi + orow;
            half_t* ol = (half_t*)g.out_lo + orow;
            unsigned ph2[NDB][2][2];                                  // [block][plane][dword]: this lane's 4 dims as packed halves
#pragma unroll
            for (int n = 0; n < NDB; ++n) {
                split_h2_pk(o[n][qb][0] * invq[qb], o[n][qb][1] * invq[qb], ph2[n][0][0], ph2[n][1][0]);
                split_h2_pk(o[n][qb][2] * invq[qb], o[n][qb][3] * invq[qb], ph2[n][0][1], ph2[n][1][1]);
            }
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                half_t* dst = pl ? ol : oh;
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {                      // block pairs (0, 1) and (2, 3)
                    const auto r0 = __builtin_amdgcn_permlane16_swap(ph2[2 * pr][pl][0], ph2[2 * pr + 1][pl][0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane16_swap(ph2[2 * pr][pl][1], ph2[2 * pr + 1][pl][1], false, false);
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 v = u32x4{(unsigned)r0[0], (unsigned)r1[0], (unsigned)r0[1], (unsigned)r1[1]};
                    // even g: own piece of block 2 pr, then lane + 16's piece -> dims 4 g .. 4 g + 7 of that block;
                    // odd g: lane - 16's piece of block 2 pr + 1, then the own one -> dims 4 (g - 1) .. 4 g + 3 of that block
                    const int d = (g4 & 1) ? 16 * (2 * pr + 1) + 4 * (g4 - 1) : 16 * (2 * pr) + 4 * g4;
                    if (ok2) *(u32x4*)(dst + d) = v;
                }
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                if (ok2) *(u32x2*)(dst + 64 + 4 * g4) = u32x2{ph2[4][pl][0], ph2[4][pl][1]};
            }
        }
#ifdef CVLM_PROBES
        WIN2_STAMP(td);
        if (trace) { acc_u += tb - ta; acc_t += tc - tb; acc_o += td - tc; if (k == 0) t_first = ta; }
#endif
    }
#ifdef CVLM_PROBES
    if (trace && lane == 0) {
        unsigned long long* o = trace + ((size_t)blockIdx.x * 8 + wave) * 8;
        o[0] = acc_u; o[1] = acc_t; o[2] = acc_o; o[3] = (unsigned long long)my_items; o[4] = t_first; o[5] = td; o[6] = acc_bar;
        o[7] = __builtin_amdgcn_s_getreg((31 << 11) | 4);                 // HW_ID: SIMD in bits 5:4
    }
#endif
    if (first) step_barrier();                                        // a workgroup without pairs still meets the producer at B_start
}

}  // namespace

#ifdef CVLM_PROBES
extern "C" int cvlm_debug_set_attn_win2_trace(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_win2_trace), &buf, sizeof(buf)); }
#endif

// split 3/3, 2/2 or 1/2, producer / consumer form; called from cvlm_attention() for the SAM window geometry
template <bool PLO, bool QLO, bool KLO>
static int launch_win14(const cvlm_attn_args& g, hipStream_t s) {
    constexpr int smem = 6 * 10240 + 17 * 1024 + 224 * 32 * 2 + 2 * (2 * 224 * 8) + 224 * 17 * 4;
    const int nwx = (g.grid + 13) / 14;
    const int npairs = g.heads * g.B * nwx * nwx;
    static bool attr[16] = {};
    if (cvlm_first_on_device(attr))
        (void)hipFuncSetAttribute((const void*)attn_win14p_kernel<PLO, QLO, KLO>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    static int cus_[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    int& cus = cus_[dev & 15];
    if (cus == 0 && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const int wgs = npairs < cus ? npairs : cus;
    hipLaunchKernelGGL((attn_win14p_kernel<PLO, QLO, KLO>), dim3(wgs), dim3(512), smem, s, g, nwx, npairs);
    CVLM_CHECK_LAUNCH();
    return 0;
}

int cvlm_attention_window14_pc(const cvlm_attn_args& g, hipStream_t s) {
    if (g.split_qk == 3 && g.split_pv == 3) return launch_win14<true, true, true>(g, s);
    if (g.split_qk == 2 && g.split_pv == 2) return launch_win14<false, false, true>(g, s);
    if (g.split_qk == 1 && g.split_pv == 2) return launch_win14<false, false, false>(g, s);
    return CVLM_E_UNSUPPORTED;
}
