// ViT-H *window* attention (14x14 windows, head_dim 80) with decomposed rel-pos bias
// (image_encoder.py:488-504, 507-553, 589-625; 28 of the 32 SAM blocks).
//
// Two workgroups of 4 waves x 32 queries (128 + 68 of the 196 queries; 2 workgroups/CU at <= 256 VGPRs) per
// (image, window, head); K/V of the whole window stream through LDS in seven 32-slot tiles (query on the lane for
// S^T and O^T, pad tokens = qkv bias rows, not stored).
//
// Software pipeline (round 2).  In the round-1 loop every wave ran  QK^T -> softmax -> P.V  of one tile back to back,
// and the two waves of a SIMD did so in lockstep behind the per-tile barrier: the matrix pipe idled through both
// softmax phases (VALU) and through the DMA address math (tile time 4050 cycles for 2 x 1184 cycles of MFMA).  Now the
// scores of tile t+1 are formed while the softmax of tile t runs: the 19 QK^T MFMAs of the next tile and the ~150 VALU
// instructions of the current softmax are independent, so one wave keeps the matrix pipe busy under its own VALU work.
// K tiles therefore arrive two tiles ahead (three-slot K ring), V tiles one ahead (two-slot V ring): 50 KB of rings,
// 14 KB one-hot block, 3.5 KB of row offsets, 8.5 KB for the prologue's Th / Tw transposition = 76 KB, two workgroups per CU.
//
// What bounds the kernel (tools/trace_attn_win.py on a probe build, B = 8: 6400 workgroups, 318 us before the changes
// below): vector-memory INSTRUCTION issue, not MFMA and not bytes -- without the rel-pos table loads -23 us, without
// the Q loads -24 us, without the output stores -43 us, without the in-loop DMA -52 us.  Hence: the tables arrive by
// 17 DMA pieces per workgroup and are read as fragments from LDS (was 20 fragment-shaped global loads per wave); K / V
// rows are staged unpadded (five DMA pieces per plane, was six); the output leaves in 16-byte stores (v_permlane32_swap
// pairs, 10 per wave, was 20).  345 -> 279 us.  Tried and dropped: ONE 8-wave workgroup per (window, head) with deep
// rings (a third of the memory instructions, but one workgroup per CU: nothing runs under its 8-us prologue, 318 us).
//
// The bias is folded into the QK^T contraction instead of being gathered per score element:
//     Q_aug = [ q (80) | Th[q][0..13]/scale | Tw[q][0..13]/scale | 0 0 0 0 ]      (112 = 7 k-steps of 16)
//     K_aug = [ k (80) | onehot14(kh)       | onehot14(kw)       | 0 0 0 0 ]
// so S^T = K_aug . Q_aug^T already contains (q.k + bias/scale).  The one-hot block is the same for every window and
// head: a 224 x 32 constant image in the code object, copied to LDS by 14 DMA instructions; Th/Tw are formed once per
// workgroup with MFMA (U = Q . R^T, 27 table rows) and kept as two extra split-half B fragments per lane.
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "../../include/cvlm.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ds_read_b64_tr_b16 as inline asm with an immediate byte offset.  The builtin form made hipcc put an
// `s_waitcnt vmcnt(0)` in front of the first transposed read of every tile (it treats the read as aliasing the LDS-DMA
// writes still in flight), i.e. the K / V tiles issued at the top of an iteration had to land inside it.  The caller
// waits with lds_wait() before the first use (guide rule 18: the wait needs a sched_barrier behind it).
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

// ---- constant one-hot block of K_aug: row = key slot (kh = slot / 14, kw = slot % 14; slots >= 196 are all zero),
// 32 columns = onehot14(kh) | 0 0 | onehot14(kw) | 0 0 (each table fills one 16-wide k-step of the contraction, so
// the Th and Tw halves of Q_aug can be built one after the other).  64-byte rows; the four 16-byte chunks of a row are stored at
// chunk ^ ((row >> 2) & 3) so that the ds_read_b128 lane groups (rows 0-3,12-15,20-27 / 4-11,16-19,28-31) hit distinct banks.
struct OneHotImage { half_t v[224 * 32]; };
constexpr OneHotImage make_onehot_image() {
    OneHotImage im{};
    for (int s = 0; s < 224; ++s)
        for (int c = 0; c < 32; ++c) {
            const int kh = s / 14, kw = s % 14;
            const bool one = s < 196 && (c == kh || c == 16 + kw);
            const int chunk = (c >> 3) ^ ((s >> 2) & 3);
            im.v[s * 32 + chunk * 8 + (c & 7)] = one ? (half_t)1.0f : (half_t)0.0f;
        }
    return im;
}
__device__ const OneHotImage g_onehot = make_onehot_image();

#ifdef CVLM_PROBES
// timeline probe (tools/trace_attn_win.py): 8 x u64 per workgroup when set
__device__ unsigned long long* g_win_trace = nullptr;
#endif
#ifndef CVLM_EXP
#define CVLM_EXP 0     // timing experiments of the probe build (wrong results): 1 no R loads, 2 no Q/R loads, 3 no stores, 4 no loop DMA
#endif

template <int SQK, int SPV>
__global__ __launch_bounds__(256, 2) void attn_win14_kernel(const cvlm_attn_args g, const int nwx, const int npairs) {
    constexpr int HD = 80, KS = 5, ND = 3, CPR = 10, KP = 80, VP = 80, L = 14, S_SEQ = 196;
    constexpr int NPL = (SQK == 3 || SPV == 3) ? 2 : 1;
    constexpr int KT = 32, NKT = 7;                           // 7 x 32 = 224 >= 196 key slots
    constexpr int KPLANE = KT * KP, VPLANE = KT * VP;
    // plane images (bytes): 32 rows x 160 B, unpadded = five 1-KiB DMA instructions exactly.  (Round 1 padded the rows to
    // 176 / 192 B for conflict-free fragment reads: six instructions per plane.  The kernel is bound by vector-memory
    // instruction issue, not by LDS cycles: two-way conflicts on the reads are the cheaper side of that trade.)
    constexpr int PLANE_B = 5120, OPSLOT_B = NPL * PLANE_B, IPP = PLANE_B / 1024;
    static_assert(KPLANE * 2 == PLANE_B && VPLANE * 2 == PLANE_B, "plane images");
    // ring layout: K0 K1 V0 K2 V1 -- K2 + V1 are contiguous and idle until iteration 0, so the prologue's Taug lives there
    constexpr int OFF_K0 = 0, OFF_K1 = OPSLOT_B, OFF_V0 = 2 * OPSLOT_B, OFF_K2 = 3 * OPSLOT_B, OFF_V1 = 4 * OPSLOT_B;
    constexpr int RING_B = 5 * OPSLOT_B;
    constexpr int OH_B = 224 * 32 * 2, TOK_B = 2 * 224 * 8;
    constexpr int TP = 17, TAUG_B = 128 * TP * 4;             // one table at a time: [query][14 values, 2 zeros, 1 dump slot]
    constexpr int TAUG_OFF = RING_B + OH_B + TOK_B;
    // rel-pos tables: the four 4320-byte planes (Rh hi, Rh lo, Rw hi, Rw lo) back to back, copied by 17 DMA pieces; the
    // image lives in the K2 + V1 slots, which stay idle until iteration 0
    constexpr int RT_B = 27 * HD * 2, RS_PIECES = (4 * RT_B + 1023) / 1024, RS_B = RS_PIECES * 1024;
    constexpr int RS_OFF = (RS_B <= 2 * OPSLOT_B) ? OFF_K2 : TAUG_OFF + TAUG_B;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half_t* OH = (half_t*)(smem + RING_B);
    // [2][224] element offsets of the K / V row of every key slot, relative to qkv_hi (a pad token's row lives in the
    // bias vector: the distance of the two allocations is folded in); bit 0 marks pad rows (offsets are multiples of 8).
    // One base pointer + selects between wave-uniform values keeps the DMA address math branch-free: with two base
    // pointers hipcc emitted an exec-masked branch pair per DMA instruction.
    long long* tokoff = (long long*)(smem + RING_B + OH_B);
    float* Taug = (float*)(smem + TAUG_OFF);                  // prologue only: [128][TP]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __builtin_assume(wave >= 0 && wave < 4);
#ifdef CVLM_PROBES
    unsigned long long* const trace = g_win_trace;
    const unsigned long long tr0 = trace ? wall_clock64() : 0;
    unsigned long long tr1 = 0, tr2 = 0, ta = 0, tb = 0, tx[6] = {0, 0, 0, 0, 0, 0};
#define STAMP(i) do { if (trace) tx[i] = wall_clock64(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
    const int qc = lane & 31, half = lane >> 5;
    // 1-D grid; the two query halves of a (window, head) pair are the workgroups id and id + 8: round-robin dispatch puts them
    // on the same XCD a moment apart, so the second reader of the pair's K / V tiles finds them in that XCD's L2
    const int wgid = blockIdx.x, qhalf = (wgid >> 3) & 1;
    const int pair = (wgid >> 4) * 8 + (wgid & 7);
    const int head = pair % g.heads, seq = pair / g.heads;
    if (pair >= npairs) return;
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

    // ---- queries first: their global loads have the longest way to go
    const int q0 = qhalf * 128 + wave * 32;
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
            if (CVLM_EXP == 2) { qh[ks] = half8{1, 2, 3, 4, 5, 6, 7, 8} * (half_t)(float)(lane + ks); ql[ks] = qh[ks] * (half_t)0.001f; continue; }
            qh[ks] = *(const half8*)(bh + 16 * ks + 8 * half);
            if (SQK == 3) ql[ks] = *(const half8*)(bl + 16 * ks + 8 * half);
        }
    }
    STAMP(0);                                                 // global loads issued
    // ---- K / V row offsets of the 224 key slots (slots >= 196 repeat the last key; they are masked in the softmax)
    if (tid < 224) {
        const long long pad_delta = pad_hi - qkv_hi;          // element distance between the two allocations
        const int tok = token_of(tid < S_SEQ ? tid : S_SEQ - 1);
        const long long ko = tok < 0 ? pad_delta + D + head * HD : (long long)qkv_offset(QS, b, tok, 1, head);
        tokoff[tid] = ko | (tok < 0 ? 1 : 0);
        tokoff[224 + tid] = (ko + (tok < 0 ? (long long)D : (long long)QS.sop)) | (tok < 0 ? 1 : 0);
    }

    // ---- tiles by LDS-DMA.  Instruction i of an operand tile (i < 6 * NPL): plane i / 6, sixth i % 6 of the plane
    // image; its 64 lanes write consecutive 16-byte chunks.  A lane whose chunk is row padding (K: 11th chunk of a
    // row, V: 11th / 12th, K image rows >= 32) re-reads chunk 0 of a valid row.
    constexpr int IPW = (IPP * NPL + 3) / 4;                  // DMA instructions per wave and operand tile
    auto op_sources = [&](int op, int t, const half_t* (&src)[IPW]) {      // all row-offset reads of a tile first ...
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            const int i = wave + 4 * j;                       // wave-uniform
            int pl = i / IPP;
            const int sub = i - IPP * pl;
            if (pl >= NPL) pl = NPL - 1;                      // instruction beyond the tile (never issued): keep its reads in range
            const int c = sub * 64 + lane;
            const int row = c / CPR, ch = c - row * CPR;
            const long long ko = tokoff[op * 224 + t * KT + row];
            const long long disp = (ko & 1) ? pad_plane : qkv_plane;          // lo-plane displacement of this row
            src[j] = qkv_hi + ((ko & ~7ll) + (pl ? disp : 0ll) + ch * 8);
        }
    };
    auto op_issue = [&](const half_t* const (&src)[IPW], unsigned char* dst) {   // ... then the DMA instructions back to back
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            const int i = wave + 4 * j;
            const int pl = i / IPP, sub = i - IPP * pl;
            if (i < IPP * NPL) glds16(src[j], dst + pl * PLANE_B + sub * 1024);
        }
    };
    auto issue_op = [&](int op, int t, unsigned char* dst) {
        const half_t* src[IPW];
        op_sources(op, t, src);
        op_issue(src, dst);
    };
    auto k_slot = [&](int t) -> unsigned char* {              // t % 3 -> K0 K1 K2
        const int r = t % 3;
        return smem + (r == 0 ? OFF_K0 : (r == 1 ? OFF_K1 : OFF_K2));
    };
    auto v_slot = [&](int t) -> unsigned char* { return smem + ((t & 1) ? OFF_V1 : OFF_V0); };

    STAMP(1);                                                 // offsets computed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // tokoff stored (the loads above stay in flight)
    __builtin_amdgcn_s_barrier();
    STAMP(2);                                                 // barrier passed
    {                                                         // rel-pos tables: 17 one-KiB pieces, lane -> (plane, byte) of the image
        const unsigned char* t0 = (const unsigned char*)g.relh_hi;
        const unsigned char* t1 = (const unsigned char*)(SQK == 3 ? g.relh_lo : g.relh_hi);
        const unsigned char* t2 = (const unsigned char*)g.relw_hi;
        const unsigned char* t3 = (const unsigned char*)(SQK == 3 ? g.relw_lo : g.relw_hi);
#pragma unroll
        for (int j = 0; j < (RS_PIECES + 3) / 4; ++j) {
            const int i = wave + 4 * j;
            int byte = i * 1024 + lane * 16;
            if (byte >= 4 * RT_B) byte = 0;                   // tail of the last piece: lands in the slack behind the image
            const int tbl = byte / RT_B, off = byte - tbl * RT_B;
            const unsigned char* src = (tbl == 0 ? t0 : (tbl == 1 ? t1 : (tbl == 2 ? t2 : t3))) + off;
            if (i < RS_PIECES) glds16(src, smem + RS_OFF + i * 1024);
        }
    }
    issue_op(0, 0, k_slot(0));
    issue_op(0, 1, k_slot(1));
    issue_op(1, 0, v_slot(0));
    {                                                         // one-hot block: 14 one-KiB copies of the constant image
        const unsigned char* src = (const unsigned char*)g_onehot.v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = wave + 4 * j;
            if (i < OH_B / 1024) glds16(src + i * 1024 + lane * 16, (unsigned char*)OH + i * 1024);
        }
    }

#ifdef CVLM_PROBES
    if (trace) ta = wall_clock64();
#endif
    // ---- Th / Tw for this query: U = Q . R^T (27 rows -> one 32-row MFMA tile per table); the table fragments come
    // from the LDS image, the result is scattered to Taug[q][..] one table at a time
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // Q loads back; this wave's DMA pieces landed
    __builtin_amdgcn_s_barrier();                             // ... everyone's: tables, one-hot block, K0, K1, V0
    STAMP(3);
    {
        const int qhh = qs / L, qww = qs - qhh * L;
        float* Tq = Taug + (wave * 32 + qc) * TP;
        if (half == 0) { Tq[14] = 0.f; Tq[15] = 0.f; }
        const int rr = qc < 27 ? qc : 26;
        floatx16 u[2];
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const unsigned char* rt = smem + RS_OFF + (2 * tb) * RT_B + rr * (HD * 2) + 16 * half;
#pragma unroll
            for (int r = 0; r < 16; ++r) u[tb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const half8 rh = *(const half8*)(rt + 32 * ks);
                u[tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rh, qh[ks], u[tb], 0, 0, 0);
                if (SQK == 3) {
                    const half8 rl = *(const half8*)(rt + RT_B + 32 * ks);
                    u[tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rl, qh[ks], u[tb], 0, 0, 0);
                    u[tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rh, ql[ks], u[tb], 0, 0, 0);
                }
            }
        }
        // The scores use q * scale (image_encoder.py:496), the rel-pos tables q itself (:497-500): fold the scale into the
        // query fragments now that U is done; the augmented columns then carry T unscaled
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                half_t hh, ll;
                split_h2(((float)qh[ks][j] + (SQK == 3 ? (float)ql[ks][j] : 0.f)) * g.scale, hh, ll);
                qh[ks][j] = hh;
                if (SQK == 3) ql[ks][j] = ll;
            }
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const int cq = tb ? qww : qhh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {                    // branch-free scatter: rows that do not exist land in the dump slot
                const int j = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int kidx = cq + L - 1 - j;
                const bool ok = j < 27 && kidx >= 0 && kidx < L;
                Tq[ok ? kidx : 16] = u[tb][r];
            }
            __builtin_amdgcn_wave_barrier();                  // rows are wave-private: LDS keeps a wave's accesses in order
            // augmented B fragment of this table: k-step 5 + tb covers its 16 columns; lane holds columns 8*half .. +8
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                half_t h, l;
                split_h2(Tq[8 * half + j], h, l);
                qh[KS + tb][j] = h;
                ql[KS + tb][j] = (SQK == 3) ? l : (half_t)0.f;
            }
            __builtin_amdgcn_wave_barrier();
        }
        STAMP(4);
    }

    float m_run = -INFINITY, l_run = 0.f;
    floatx16 o[ND];
#pragma unroll
    for (int n = 0; n < ND; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[n][r] = 0.f;
    const int tg = lane >> 4, ti = lane & 15;
    const int v_lane_off = (4 * (tg >> 1) + (ti >> 2)) * VP + 16 * (tg & 1) + 4 * (ti & 3);

    // S^T tile of key tile t: 15 (5 k-steps x 3 products) + 4 (bias) MFMAs
    auto scores = [&](int t, auto last_c) -> floatx16 {
        floatx16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const half_t* kr = (const half_t*)k_slot(t) + qc * KP + 8 * half;
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
        const int row = t * KT + qc;
        const half_t* ohr = OH + row * 32;
#pragma unroll
        for (int a = 0; a < 2; ++a) {                         // bias: exact one-hot rows x (T/scale) hi (+ lo)
            const half8 oh8 = *(const half8*)(ohr + 8 * ((half + 2 * a) ^ ((row >> 2) & 3)));
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(oh8, qh[KS + a], s, 0, 0, 0);
            if (SQK == 3) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(oh8, ql[KS + a], s, 0, 0, 0);
        }
        if (decltype(last_c)::value) {                        // only the last tile holds slots beyond the 196 keys
            const int b0 = t * KT + 4 * half;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (b0 + (r & 3) + 8 * (r >> 2) >= S_SEQ) s[r] = -INFINITY;
        }
        return s;
    };

#ifdef CVLM_PROBES
    if (trace) { asm volatile("" ::"v"(qh[0]), "v"(qh[KS + 1])); tb = wall_clock64(); }
#endif
#ifdef CVLM_PROBES
    if (trace) tr1 = wall_clock64();
#endif
    typedef std::integral_constant<bool, false> no_c;
    typedef std::integral_constant<bool, true> yes_c;
    floatx16 s_cur;
    if (wave_active) s_cur = scores(0, no_c{});

    // One key tile.  HAS_NEXT: also form the scores of tile t + 1 (NEXT_LAST: that tile is the masked last one).  The body is
    // one basic block (rescale unconditional, flags compile-time) so that the scheduler can run the next tile's QK^T MFMAs
    // underneath the softmax VALU work.
    auto tile = [&](int t, auto has_next_c, auto next_last_c) {
        constexpr bool HAS_NEXT = decltype(has_next_c)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of K(t+1) / V(t) has landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // ... everyone's; K slot (t+2)%3 and V slot (t+1)&1 are free
        {
            const half_t* ksrc[IPW];
            const half_t* vsrc[IPW];
            const int tk = t + 2 < NKT ? t + 2 : NKT - 1, tv = t + 1 < NKT ? t + 1 : NKT - 1;   // clamped: reads stay in range
            op_sources(0, tk, ksrc);
            op_sources(1, tv, vsrc);
            if (CVLM_EXP != 4) {
                if (t + 2 < NKT) op_issue(ksrc, k_slot(t + 2));
                if (HAS_NEXT) op_issue(vsrc, v_slot(t + 1));
            } else {
                asm volatile("" ::"v"(ksrc[0]), "v"(vsrc[0]), "v"(ksrc[IPW - 1]), "v"(vsrc[IPW - 1]));
            }
        }
        if (!wave_active) return;
        floatx16 s_next;
        if (HAS_NEXT) s_next = scores(t + 1, next_last_c);
        const unsigned vaddr = (unsigned)(size_t)(LDS_AS const unsigned char*)v_slot(t) + 2u * (unsigned)v_lane_off;
        // V fragments of one 16-key step: [n][plane] = two transposed 8-byte reads each (rows +0 / +8 of the step);
        // written out by hand because the byte offset must be an immediate
        auto read_v = [&](auto k2_c, half4 (&v0)[ND][2], half4 (&v1)[ND][2]) {
            constexpr int K2 = decltype(k2_c)::value;
            v0[0][0] = lds_read_tr16<2 * (16 * K2 * VP + 0)>(vaddr);       v1[0][0] = lds_read_tr16<2 * (16 * K2 * VP + 0 + 8 * VP)>(vaddr);
            v0[1][0] = lds_read_tr16<2 * (16 * K2 * VP + 32)>(vaddr);      v1[1][0] = lds_read_tr16<2 * (16 * K2 * VP + 32 + 8 * VP)>(vaddr);
            v0[2][0] = lds_read_tr16<2 * (16 * K2 * VP + 64)>(vaddr);      v1[2][0] = lds_read_tr16<2 * (16 * K2 * VP + 64 + 8 * VP)>(vaddr);
            if (SPV == 3) {
                v0[0][1] = lds_read_tr16<PLANE_B + 2 * (16 * K2 * VP + 0)>(vaddr);  v1[0][1] = lds_read_tr16<PLANE_B + 2 * (16 * K2 * VP + 0 + 8 * VP)>(vaddr);
                v0[1][1] = lds_read_tr16<PLANE_B + 2 * (16 * K2 * VP + 32)>(vaddr); v1[1][1] = lds_read_tr16<PLANE_B + 2 * (16 * K2 * VP + 32 + 8 * VP)>(vaddr);
                v0[2][1] = lds_read_tr16<PLANE_B + 2 * (16 * K2 * VP + 64)>(vaddr); v1[2][1] = lds_read_tr16<PLANE_B + 2 * (16 * K2 * VP + 64 + 8 * VP)>(vaddr);
            }
        };
        half4 va0[ND][2], va1[ND][2], vb0[ND][2], vb1[ND][2];
        read_v(std::integral_constant<int, 0>{}, va0, va1);     // in flight under the softmax
        const floatx16 s = s_cur;
        // online softmax on packed fp32 pairs
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
#pragma unroll
        for (int n = 0; n < ND; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[n][r] *= alpha;
        m_run = m_new;
        half8 ph[2], pl[2];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2) {                  // hi truncated (cvt_pkrtz), lo = e - hi: exact remainder
                const f32x2 e = z[4 * k2 + p2];
                const half2v h = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(e.x, e.y));
                ph[k2][2 * p2] = h[0]; ph[k2][2 * p2 + 1] = h[1];
                if (SPV == 3) {
                    const half2v l = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(e.x - (float)h[0], e.y - (float)h[1]));
                    pl[k2][2 * p2] = l[0]; pl[k2][2 * p2 + 1] = l[1];
                }
            }
        auto pv = [&](int k2, half4 (&v0)[ND][2], half4 (&v1)[ND][2]) {
#pragma unroll
            for (int n = 0; n < ND; ++n) {
                const half8 vh = half8{v0[n][0][0], v0[n][0][1], v0[n][0][2], v0[n][0][3], v1[n][0][0], v1[n][0][1], v1[n][0][2], v1[n][0][3]};
                o[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[k2], o[n], 0, 0, 0);
                if (SPV == 3) {
                    const half8 vl = half8{v0[n][1][0], v0[n][1][1], v0[n][1][2], v0[n][1][3], v1[n][1][0], v1[n][1][1], v1[n][1][2], v1[n][1][3]};
                    o[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[k2], o[n], 0, 0, 0);
                    o[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[k2], o[n], 0, 0, 0);
                }
            }
        };
        lds_wait();                                           // step-0 fragments (and everything older) are in registers
        read_v(std::integral_constant<int, 1>{}, vb0, vb1);     // in flight under the first nine MFMAs
        pv(0, va0, va1);
        lds_wait();
        pv(1, vb0, vb1);
        if (HAS_NEXT) s_cur = s_next;
    };
#pragma unroll 1
    for (int t = 0; t < NKT - 2; ++t) tile(t, yes_c{}, no_c{});
    tile(NKT - 2, yes_c{}, yes_c{});
    tile(NKT - 1, no_c{}, no_c{});

#ifdef CVLM_PROBES
    if (trace) tr2 = wall_clock64();
#endif
    const float l_tot = half_swap_sum(l_run);
    if (wave_active) {
        // 16-byte stores (guide T21): a lane pair (q, half 0 / 1) holds dims d..d+3 / d+4..d+7 of every 8-dim group; one
        // v_permlane32_swap per register pair hands the even group's upper four dims to the lower lane and the odd group's
        // lower four to the upper lane, so each lane stores 8 consecutive dims: 10 store instructions per wave instead of
        // 20 (the stores were 43 us of a 318-us launch, tools/trace_attn_win.py experiment 3).
        const float inv = 1.0f / l_tot;
        const bool st_ok = qvalid && qtok >= 0;
        const int64_t orow = ((int64_t)b * SI + (st_ok ? qtok : 0)) * D + head * HD;
        half_t* oh = (half_t*)g.out_hi + orow;
        half_t* ol = g.out_lo ? (half_t*)g.out_lo + orow : nullptr;
#pragma unroll
        for (int n = 0; n < ND; ++n)
#pragma unroll
            for (int rp = 0; rp < 2; ++rp) {
                if (32 * n + 16 * rp < HD) {                  // compile-time
                    unsigned xe[2][2], xo[2][2];              // [plane][dword] of the even / odd 8-dim group
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        half_t h[4], l4[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) split_h2(o[n][4 * (2 * rp + e) + j] * inv, h[j], l4[j]);
                        unsigned (&x)[2][2] = e ? xo : xe;
                        x[0][0] = __builtin_bit_cast(unsigned, half2v{h[0], h[1]});
                        x[0][1] = __builtin_bit_cast(unsigned, half2v{h[2], h[3]});
                        x[1][0] = __builtin_bit_cast(unsigned, half2v{l4[0], l4[1]});
                        x[1][1] = __builtin_bit_cast(unsigned, half2v{l4[2], l4[3]});
                    }
                    const int d = 32 * n + 16 * rp + 8 * half;   // lower lanes: the even group, upper lanes: the odd group
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        const auto r0 = __builtin_amdgcn_permlane32_swap(xe[pl][0], xo[pl][0], false, false);
                        const auto r1 = __builtin_amdgcn_permlane32_swap(xe[pl][1], xo[pl][1], false, false);
                        // after the swap: first result = own even (lower lanes) / lower lanes' odd (upper lanes);
                        // second = upper lanes' even (lower lanes) / own odd (upper lanes)
                        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                        const u32x4 v = u32x4{(unsigned)r0[0], (unsigned)r1[0], (unsigned)r0[1], (unsigned)r1[1]};
                        half_t* dst = pl ? ol : oh;
                        if (st_ok && dst) *(u32x4*)(dst + d) = v;
                    }
                }
            }
    }
#ifdef CVLM_PROBES
    if (trace && tid == 0) {
        unsigned long long* o8 = trace + (size_t)blockIdx.x * 8;
        o8[0] = tr0; o8[1] = tr1; o8[2] = tr2; o8[3] = wall_clock64(); o8[4] = ta; o8[5] = tb;
        o8[6] = __builtin_amdgcn_s_getreg((31 << 11) | 4); o8[7] = 0;
        unsigned long long* x8 = trace + ((size_t)gridDim.x + blockIdx.x) * 8;
        for (int i = 0; i < 6; ++i) x8[i] = tx[i];
    }
#endif
}

template <int SQK, int SPV>
int launch_win(const cvlm_attn_args& g, hipStream_t s) {
    constexpr int NPL = (SQK == 3 || SPV == 3) ? 2 : 1;
    constexpr int opslot = NPL * 5120, taug = 128 * 17 * 4, rs = 17 * 1024;
    constexpr int smem = 5 * opslot + 224 * 32 * 2 + 2 * 224 * 8 + taug + (rs <= 2 * opslot ? 0 : rs);
    const int nwx = (g.grid + 13) / 14;
    auto kern = attn_win14_kernel<SQK, SPV>;
    static bool attr[16] = {};
    if (smem > 48 * 1024 && cvlm_first_on_device(attr))
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    const int npairs = g.heads * g.B * nwx * nwx;
    hipLaunchKernelGGL(kern, dim3(2 * ((npairs + 7) / 8) * 8), dim3(256), smem, s, g, nwx, npairs);
    CVLM_CHECK_LAUNCH();
    return 0;
}

}  // namespace

int cvlm_attention_window14_pc(const cvlm_attn_args& g, hipStream_t s);   // attention_win2.hip: producer / consumer form (exact mode)

// called from cvlm_attention() for mode 2, window 14, head_dim 80
int cvlm_attention_window14(const cvlm_attn_args& g, hipStream_t s) {
    // exact mode with an h2 output: producer / consumer form, persistent over (window, head) pairs (attention_win2.hip); the kernel
    // of this file (two 4-wave workgroups per (window, head)) serves the other precisions and a hi-plane-only output
    if (g.split_qk == 3 && g.split_pv == 3 && g.out_lo) return cvlm_attention_window14_pc(g, s);
    if (g.split_qk == 3 && g.split_pv == 3) return launch_win<3, 3>(g, s);
    if (g.split_qk == 3 && g.split_pv == 1) return launch_win<3, 1>(g, s);
    if (g.split_qk == 1 && g.split_pv == 1) return launch_win<1, 1>(g, s);
    return CVLM_E_UNSUPPORTED;
}

#ifdef CVLM_PROBES
// Probe hook (not part of include/cvlm.h).
extern "C" int cvlm_debug_set_attn_win_trace(void* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_win_trace), &buf, sizeof(buf));
}
#endif
