// The split-half GEMM kernel template of gemm.hip / gemm_il.hip (see gemm.hip for the description).
#pragma once
#include <type_traits>
#include "common.h"
#include <stdlib.h>
#include "../../include/cvlm.h"

namespace cvlm_gemm_k {

constexpr int BK_MIN = 32;
constexpr int SK_COUNTER0 = 4 * 128 + 2;        // split-K arrival counters live behind the tail words and the two error words
constexpr int SK_MAX_TILES = 1024 - SK_COUNTER0;  // ... in the first 4-KiB page of the workspace

// chunk permutation g(q), q = (row >> 2) & 3 (derived for the ds_read_b128 lane groups)
__device__ __forceinline__ int swz4(int q) { return (0x78 >> (2 * q)) & 3; }

// 16 zero bytes: what the DMA of an implicit 3x3 convolution reads for a tap that falls outside the image
static __device__ uint4 g_zero16 = {0u, 0u, 0u, 0u};

struct GemmParams {
    cvlm_gemm_args a;
    int nbx, nby;
    int group_m;       // tile rows per L2 super-tile (consecutive ids walk group_m x nbx tiles column-major)
    // tail split (256^2 staggered kernel only): the last `tail_rem` tiles of a grid that does not fill its final
    // round are cut into `tail_split` K-parts, one workgroup each; parts 0..S-2 leave f32 partial slabs in `ws`
    // and raise `flags` (chain: part k adds part k-1's running sum), part S-1 (highest block index) runs the epilogue.
    int tail_rem, tail_split;
    int total_blocks;  // PERSIST: workgroup b walks ids b, b + gridDim.x, ... < total_blocks
    int sk_parts;      // SK kernels: K-parts per tile (grid = tiles x sk_parts)
    float* ws;
    unsigned* flags;   // [tail_rem][4] arrival words (1 when ready; the consumer puts 0 back), then error words at [4 * 128] (abandoned hand-offs), [4 * 128 + 1] (LayerNorm-fold rows out of range)
#ifdef CVLM_PROBES
    unsigned long long* trace;   // DBG == 4 only: 8 x u64 per workgroup (timeline probe, tools/trace_gemm.py)
#endif
};

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// PERSIST (256^2 staggered kernel, no tail parts): one workgroup per CU walks the tile list; the first K-tile of the
// next tile is requested before the epilogue of the current one, so the tile prologue (address set-up + first DMA latency,
// ~4 us of a ~110 us K = 1280 tile) runs under the epilogue's stores.  The epilogue stages through LDS behind slot 0.
// EPI >= 0 (256^2 kernel): only epilogue form EPI of the LDS-staged path is compiled in (0 plain, 1 LayerNorm fold, 2 h2
// residual + row statistics) -- one function with all three let the register needs of one form decide the allocation of
// the others (batched statistics in form 2 cost the fold-form launches 4 %).  -1: run-time dispatch, every form.
// CONV (2-stage loop): A is an NHWC image (conv_h x conv_w x conv_c per batch item, lda = conv_c) and the K axis runs
// over the 9 taps of a 3x3 / pad 1 / stride 1 convolution, k = (ky*3 + kx)*C + c: the gather of im2col happens in the
// DMA source addresses (a tap outside the image reads 16 zero bytes), nothing is materialised.
// SK (small grids: one image, the CLIP towers at M = 581): EVERY tile is cut into p.sk_parts K-parts, one workgroup each.  A part
// stores its fp32 accumulators as a slab in the workspace and counts itself in; the part that arrives LAST (an atomic counter,
// no polling: safe whatever else shares the chip) adds the slabs of all parts in index order -- its own included, read back
// like the others, so the order of the fp32 additions does not depend on who was last -- and runs the epilogue.
// WIL (ABI 6, cvlm_gemm_args.w_il): the weight operand is staged from the image whose planes are interleaved per 32 k-elements --
// a weight row's K-tile is ONE 128-byte line (hi 64 B | lo 64 B), a DMA instruction covers 8 rows x 128 bytes instead of 16 rows x 64
// bytes of one plane, and every L2 line is requested once instead of once per half.  In LDS the weight region becomes [BN rows][128
// bytes], 16-byte chunks of row r permuted by ^ ((r >> 1) & 7) (the conflict-free image of the 128-byte-row loop); chunks 0-3 of a
// row are the hi plane's k 0-31, chunks 4-7 the lo plane's.  Same fragments, same MFMAs, same bits.
template <int SPLIT, int WM, int WN, int NSTAGE, int BK, int DBG = 0, int MT = 4, bool PERSIST = false, int EPI = -1, bool CONV = false,
          bool SK = false, bool WIL = false, bool AIL = false, bool MX = false>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_nt_kernel(const GemmParams p) {
    // MX (ABI 10, cvlm_gemm_args.a_mx / w_mx): both operands are mx images -- per row and 64 k-elements one 128-byte line of fp16 hi
    // values and one 128-byte line of e4m3 bytes (hi8 | lo8).  The ring's K-tiles become UNITS of one line per row, alternately an f16
    // unit (two 16x16x32 f16 MFMAs per output tile: k 0-31 and 32-63 of hi.hi) and an fp8 unit (ONE v_mfma_scale_f32_16x16x128_f8f6f4
    // per output tile whose K = 128 is [Whi8 . Alo8 | Wlo8 . Ahi8] over the same 64 k): 2048 matrix-pipe cycles per wave and 64 k
    // instead of 3072, at 0.65 x the joules (profiles/r05_power_formats.log).  Staging, LDS image, fragment addresses and the
    // accumulator layout are those of the WIL + AIL form: same bytes per row and unit, same ds_read_b128 pattern.
    static_assert(!MX || (WIL && AIL && NSTAGE == 5 && SPLIT == 3 && !PERSIST && !SK && !CONV && (MT % 2) == 0), "mx operands: the staggered 256-column kernel");
    static_assert(!WIL || (SPLIT == 3 && BK == 32 && !CONV && NSTAGE != 4 && NSTAGE != 6), "interleaved weights: split-3 kernels with 32-wide K-tiles");
    // AIL (cvlm_gemm_args.a_il): the same for the ACTIVATION operand -- a_hi is an image [M][K / 32][plane][32] (what an out_il launch
    // or cvlm_row_stats_split with il wrote), lda its row stride in halves; LDS region [BM rows][128 bytes], same chunk permutation.
    static_assert(!AIL || WIL, "interleaved activations come with interleaved weights (one set of instantiations)");
    static_assert(!SK || (!PERSIST && !CONV && NSTAGE != 5), "split-K form: plain tile loops only");
    static_assert(!PERSIST || NSTAGE == 5, "persistent form exists for the staggered 256^2 loop only");
    static_assert(!CONV || (NSTAGE == 2 && BK == 32), "implicit 3x3 convolution: 2-stage loop, 32-wide K-tiles");
    constexpr int WROWS = MT * 16;                                  // activation rows per wave
    constexpr int BM = WM * WROWS, BN = WN * 64, NWAVE = WM * WN;
    constexpr int NPA = (SPLIT == 3) ? 2 : 1;                       // planes per operand
    constexpr int A_PLANE = BM * BK * 2, W_PLANE = BN * BK * 2;     // bytes
    constexpr int STAGE = NPA * (A_PLANE + W_PLANE);
    constexpr int ROWB = BK * 2;                                    // bytes per tile row (64 or 128)
    constexpr int RPI = 1024 / ROWB;                                // rows per 1-KiB DMA instruction (16 or 8)
    constexpr int CPR = ROWB / 16;                                  // 16-byte chunks per row (4 or 8)
    constexpr int A_INSTR = BM / RPI, W_INSTR = BN / RPI;           // DMA instructions per plane
    constexpr int TOTAL = NPA * (A_INSTR + W_INSTR);
    static_assert(TOTAL % NWAVE == 0, "staging must divide evenly over the waves");
    constexpr int PER_WAVE = TOTAL / NWAVE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const cvlm_gemm_args& g = p.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // provably wave-uniform: tile offsets stay in SGPRs
    const int wm = wave / WN, wn = wave - wm * WN;
    [[maybe_unused]] unsigned long long tr0 = 0, tr1 = 0, tr2 = 0, tr3 = 0, tr4 = 0;   // DBG == 4 timeline stamps
    if (DBG == 4) tr0 = wall_clock64();

    const int ntiles = p.nbx * p.nby;
    const int z = blockIdx.y;
    const half_t* Ahi = (const half_t*)g.a_hi + (int64_t)z * g.stride_a;
    const half_t* Alo = (const half_t*)g.a_lo + (int64_t)z * g.stride_a;
    const half_t* Whi = MX ? (const half_t*)g.w_mx : WIL ? (const half_t*)g.w_il : (const half_t*)g.w_hi + (int64_t)z * g.stride_w;
    const half_t* Wlo = (const half_t*)g.w_lo + (int64_t)z * g.stride_w;

    // ---- per-tile state (set_tile): coordinates, K range, DMA source of every staging instruction of this wave
    int bm = 0, bn = 0, kpart = 0, kparts = 1, tail_j = 0, nk = 0;
    [[maybe_unused]] int k0_units = 0;                                // MX: first unit of this workgroup's K range (a multiple of 8)
    const half_t* src[PER_WAVE];
    int dst_off[PER_WAVE];                                            // stage layout: [Ahi][Alo][Whi][Wlo]; instruction i covers 16 rows x 64 B
    // WIL: K advance of staging instruction j in halves per K-tile element -- weight instructions walk the interleaved image, where a
    // K-tile is 64 halves (hi | lo) of a row; the instruction index is wave-uniform, so this is scalar arithmetic
    auto kmul = [&](int j) -> int { return (wave * PER_WAVE + j >= NPA * A_INSTR ? WIL : AIL) ? 2 : 1; };
    [[maybe_unused]] int tapmask[PER_WAVE];                           // CONV: bit t = tap t of this lane's row lies inside the image; bit 9 = weight row
    // Everything a run-time ?: selects between comes in as a parameter or is a local of the body: a conditional between two
    // by-reference captures becomes a run-time index into the closure, which pins it -- and every capture -- in scratch.
    auto set_tile_ = [&](int pid, const half_t* ahi, const half_t* alo, const half_t* whi, const half_t* wlo,
                         int64_t lda_, int64_t ldw_, int M_, int N_) {
        // tile coordinates: XCD-aware bijective remap of the 1-D tile id (8 XCDs, round-robin dispatch)
        kpart = 0; kparts = 1; tail_j = 0;
        if (SK) {                                                        // parts-major: all part-0 workgroups first
            kparts = p.sk_parts;
            kpart = pid / ntiles;
            pid -= kpart * ntiles;
            tail_j = pid;                                                // slab / counter index of the tile
        }
        if (NSTAGE == 5 && !PERSIST && p.tail_rem > 0 && pid >= ntiles - p.tail_rem) {
            const int j = pid - (ntiles - p.tail_rem);
            kparts = p.tail_split;
            kpart = j / p.tail_rem;                                      // producers first, the owner (S-1) last
            tail_j = j - kpart * p.tail_rem;
            pid = ntiles - p.tail_rem + tail_j;
        }
        {
            const int q = ntiles >> 3, r = ntiles & 7, xcd = pid & 7, idx = pid >> 3;
            pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        }
        // grouped order: the workgroups co-resident on one XCD cover a (group_m x n) patch of tiles, so each
        // A / W k-slice they stream is fetched from HBM/L3 once and hit in the XCD's L2 afterwards.
        int by, bx;
        {
            const int per_group = p.group_m * p.nbx;
            const int grp = pid / per_group;
            const int first = grp * p.group_m;
            const int gm = (p.nby - first) < p.group_m ? (p.nby - first) : p.group_m;
            const int rem = pid - grp * per_group;
            by = first + rem % gm;
            bx = rem / gm;
        }
        const int bm_ = by * BM, bn_ = bx * BN;
        bm = bm_; bn = bn_;
        const int rsub = lane / CPR;                                  // row within the instruction's row group
        const int pos = lane % CPR;                                   // LDS chunk position within the row
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j) {
            const int i = wave * PER_WAVE + j;
            const bool isW = i >= NPA * A_INSTR;
            if (AIL && !isW) {                                            // activations: 8 rows x 128 bytes of the interleaved image
                const int row8 = i * 8 + (lane >> 3), pos8 = lane & 7;
                int grow8 = bm_ + row8;
                grow8 = grow8 < M_ - 1 ? grow8 : M_ - 1;
                src[j] = ahi + (int64_t)grow8 * lda_ + ((pos8 ^ ((row8 >> 1) & 7)) * 8);
                dst_off[j] = i * 1024;
                continue;
            }
            if (WIL && isW) {                                             // 8 rows x 128 bytes of the interleaved image
                const int ii2 = i - NPA * A_INSTR;
                const int row8 = ii2 * 8 + (lane >> 3), pos8 = lane & 7;
                int grow8 = bn_ + row8;
                grow8 = grow8 < N_ - 1 ? grow8 : N_ - 1;
                src[j] = whi + (int64_t)grow8 * ldw_ + ((pos8 ^ ((row8 >> 1) & 7)) * 8);
                dst_off[j] = NPA * A_PLANE + ii2 * 1024;
                continue;
            }
            const int ii = isW ? i - NPA * A_INSTR : i;
            const int per = isW ? W_INSTR : A_INSTR;
            const int plane = ii / per, sub = ii - plane * per;
            const int row = sub * RPI + rsub;
            // source chunk for this LDS position (involution; BK=32: 4-chunk rows, BK=64: 8-chunk rows)
            const int chunk = (BK == 32) ? (pos ^ swz4((row >> 2) & 3)) : (pos ^ ((row >> 1) & 7));
            const half_t* base = isW ? (plane ? wlo : whi) : (plane ? alo : ahi);
            const int64_t ld = isW ? ldw_ : lda_;
            int grow = (isW ? bn_ : bm_) + row;
            const int lim = (isW ? N_ : M_) - 1;
            grow = grow < lim ? grow : lim;
            src[j] = base + (int64_t)grow * ld + chunk * 8;
            dst_off[j] = (isW ? NPA * A_PLANE + plane * W_PLANE : plane * A_PLANE) + sub * 1024;
            if (CONV) {
                const int px = grow % g.conv_w, py = (grow / g.conv_w) % g.conv_h;
                int mk = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
                    mk |= (yy >= 0 && yy < g.conv_h && xx >= 0 && xx < g.conv_w) ? (1 << t) : 0;
                }
                tapmask[j] = isW ? 512 : mk;
            }
        }
        nk = g.K / BK;
        k0_units = 0;
        if ((NSTAGE == 5 || SK) && kparts > 1) {                         // this workgroup's share of the K-tiles
            // MX: parts are whole groups of 8 units (4 unit pairs = one dword of scale bytes per fragment row)
            const int nku = MX ? (nk + 7) / 8 : nk;
            const int base = nku / kparts, extra = nku - base * kparts;
            int k0 = kpart * base + (kpart < extra ? kpart : extra);
            int mine = base + (kpart < extra ? 1 : 0);
            if (MX) { k0 *= 8; mine = mine * 8 < nk - k0 ? mine * 8 : nk - k0; k0_units = k0; }
            nk = mine;
#pragma unroll
            for (int j = 0; j < PER_WAVE; ++j) src[j] += (int64_t)k0 * BK * kmul(j);
        }
    };
    auto set_tile = [&](int pid) { set_tile_(pid, Ahi, Alo, Whi, Wlo, g.lda, MX ? g.ldw_mx : WIL ? g.ldw_il : g.ldw, g.M, g.N); };
    int vblk = blockIdx.x;
    set_tile(vblk);

    // ---- fragment read offsets (bytes within a plane)
    const int fr = lane & 15, fq = lane >> 4;
    const int a_row = (wm * WROWS + fr) * ROWB;         // + mt*16*ROWB
    const int w_row = (wn * 64 + fr) * ROWB;            // + nt*16*ROWB
    auto chunk_off = [&](int ks) -> int {               // byte offset of this lane's 16-B chunk of k-step ks
        return (BK == 32) ? ((fq ^ swz4((fr >> 2) & 3)) * 16) : (((4 * ks + fq) ^ ((fr >> 1) & 7)) * 16);
    };
    // activation fragment of m-tile mt, plane pl
    auto a_frag = [&](const unsigned char* cur, int mt, int pl, int co) -> half8 {
        if (AIL) return *(const half8*)(cur + (wm * WROWS + fr + mt * 16) * 128 + (((4 * pl + fq) ^ ((fr >> 1) & 7)) * 16));
        return *(const half8*)(cur + pl * A_PLANE + a_row + mt * 16 * ROWB + co);
    };
    // weight fragment i (16 rows) of plane pl from the stage at `cur` (byte address); `co` = chunk_off of the k-step
    auto w_frag = [&](const unsigned char* cur, int i, int pl, int co) -> half8 {
        if (WIL) return *(const half8*)(cur + NPA * A_PLANE + (wn * 64 + fr + i * 16) * 128 + (((4 * pl + fq) ^ ((fr >> 1) & 7)) * 16));
        return *(const half8*)(cur + NPA * A_PLANE + pl * W_PLANE + w_row + i * 16 * ROWB + co);
    };

    floatx4 acc[MT][4];
    // DMA source of staging instruction j for K-tile t.  CONV: K-tile t lies in tap t / (C / 32) (C a power of two), channels
    // from (t % (C / 32)) * 32; the tap moves the pixel by (dy, dx), i.e. the address by a wave-uniform offset.
    const int conv_lc = CONV ? 31 - __builtin_clz((unsigned)(g.conv_c / BK)) : 0;
    auto src_at = [&](int j, int t) -> const void* {
        if (!CONV) return src[j] + (int64_t)t * BK * kmul(j);
        const int tap = t >> conv_lc, c0 = (t - (tap << conv_lc)) * BK;
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;                            // tap / 3, tap % 3 for tap in 0..8
        const int64_t off = (int64_t)((ky - 1) * g.conv_w + (kx - 1)) * g.conv_c + c0;
        const half_t* inside = src[j] + off;
        const half_t* plain = src[j] + (int64_t)t * BK;
        const void* a = ((tapmask[j] >> tap) & 1) ? (const void*)inside : (const void*)&g_zero16;
        return (tapmask[j] & 512) ? (const void*)plain : a;
    };
    auto issue = [&](int t, int slot) {
        if (DBG == 1 && t > 1) return;                       // timing probe: no DMA in the steady state
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j) glds16(src_at(j, t), smem + slot * STAGE + dst_off[j]);
    };
    // One K-tile of MFMAs from `slot`; the DMA of K-tile `tn` into `sn` (tn < 0: none) is issued in pieces
    // between the MFMA groups so the load issue (readfirstlane + m0 + TA acceptance, ~100 cycles each)
    // hides in the matrix pipe's shadow instead of stalling all waves right after the barrier.
    constexpr int NG = (BK / 32) * MT;                               // MFMA groups (one per (k-step, m-tile))
    constexpr int MG = (MT % 4 == 0) ? 4 : (MT % 3 == 0 ? 3 : (MT % 2 == 0 ? 2 : 1));   // m-tiles whose fragments are held at once
    auto compute = [&](int slot, int tn, int sn) {
        if (DBG == 2) {
            if (tn >= 0) issue(tn, sn);
            return;
        }
        const unsigned char* cur = smem + slot * STAGE;
        const bool dma = tn >= 0 && !(DBG == 1 && tn > 1);
        unsigned char* nxt = smem + sn * STAGE;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            const int co = chunk_off(ks);
            half8 wh[4], wl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wh[i] = w_frag(cur, i, 0, co);
                if (SPLIT == 3) wl[i] = w_frag(cur, i, 1, co);
            }
#pragma unroll
            for (int mh = 0; mh < MT / MG; ++mh) {                  // 4 m-tiles at a time keeps fragments at 64 VGPRs
                half8 ah[MG], al[MG];
#pragma unroll
                for (int i = 0; i < MG; ++i) {
                    ah[i] = a_frag(cur, mh * MG + i, 0, co);
                    if (SPLIT == 3) al[i] = a_frag(cur, mh * MG + i, 1, co);
                }
#pragma unroll
                for (int mt = 0; mt < MG; ++mt) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        floatx4 c = acc[mh * MG + mt][nt];
                        if (SPLIT == 3) {
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[nt], ah[mt], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt], al[mt], c, 0, 0, 0);
                        }
                        acc[mh * MG + mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt], ah[mt], c, 0, 0, 0);
                    }
                    const int gidx = (ks * (MT / MG) + mh) * MG + mt;
                    // DMA pieces go out during the first half of the MFMA groups: the last piece then has half a
                    // K-tile of MFMA time to land before the end-of-tile wait
                    constexpr int NGI = (NSTAGE == 2 && NG >= 2) ? NG / 2 : NG;
                    const int g0 = gidx < NGI ? gidx : NGI, g1 = gidx + 1 < NGI ? gidx + 1 : NGI;
                    const int j0 = (g0 * PER_WAVE) / NGI, j1 = (g1 * PER_WAVE) / NGI;
                    if (dma) {
#pragma unroll
                        for (int j = j0; j < j1; ++j) glds16(src_at(j, tn), nxt + dst_off[j]);
                    }
                }
            }
        }
    };

    bool first_tile = true;
  for (;;) {                                                 // tile loop: one pass unless PERSIST
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    if (NSTAGE == 2) {
        issue(0, 0);
        __syncthreads();                                     // emits vmcnt(0) for the in-flight DMA
        for (int t = 0; t < nk; ++t) {
            compute(t & 1, t + 1 < nk ? t + 1 : -1, (t + 1) & 1);
            __syncthreads();                                 // next tile landed (vmcnt(0)) and cur fully read
        }
    } else if (NSTAGE == 4) {
        // Two slots with mid-tile recycling (MT = 8, BK = 32): all fragments of tile t are pulled into
        // registers at the top of the iteration, a barrier then frees slot t&1 and the DMA of tile t+2
        // streams into it underneath the 96 MFMAs of tile t.  Each DMA has ~1.5 iterations to land and
        // the memory pipe never drains: s_waitcnt vmcnt is counted (never 0 in steady state) and the
        // barriers are raw s_barrier (guide §5 "Pipelining across barriers").
        issue(0, 0);
        if (nk > 1) { issue(1, 1); wait_vmcnt<PER_WAVE>(); } else { wait_vmcnt<0>(); }
        __builtin_amdgcn_s_barrier();
        const int co = chunk_off(0);
        for (int t = 0; t < nk; ++t) {
            const unsigned char* cur = smem + (t & 1) * STAGE;
            const unsigned char* pAhi = cur;
            const unsigned char* pAlo = cur + A_PLANE;
            const unsigned char* pWhi = cur + NPA * A_PLANE;
            const unsigned char* pWlo = pWhi + W_PLANE;
            half8 wh[4], wl[4], ah[MT], al[MT];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wh[i] = *(const half8*)(pWhi + w_row + i * 16 * ROWB + co);
                if (SPLIT == 3) wl[i] = *(const half8*)(pWlo + w_row + i * 16 * ROWB + co);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                ah[i] = *(const half8*)(pAhi + a_row + i * 16 * ROWB + co);
                if (SPLIT == 3) al[i] = *(const half8*)(pAlo + a_row + i * 16 * ROWB + co);
            }
            const bool dma = (t + 2 < nk) && DBG != 1;
            const int64_t koff = (int64_t)(t + 2) * BK;
            unsigned char* nxt = smem + (t & 1) * STAGE;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (DBG != 2) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        floatx4 c = acc[mt][nt];
                        if (SPLIT == 3) {
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[nt], ah[mt], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt], al[mt], c, 0, 0, 0);
                        }
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt], ah[mt], c, 0, 0, 0);
                    }
                }
                if (mt == 0) {
                    // every fragment of tile t is in registers: slot t&1 may be overwritten once ALL waves are here
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
                if (dma && mt >= 1 && mt <= 4) {
                    constexpr int PP = PER_WAVE / 4;
#pragma unroll
                    for (int j = (mt - 1) * PP; j < (mt == 4 ? PER_WAVE : mt * PP); ++j)
                        glds16(src[j] + koff * kmul(j), nxt + dst_off[j]);
                }
            }
            // tile t+1 (issued one iteration ago) must have landed for every wave; tile t+2 may stay in flight
            if (dma) wait_vmcnt<PER_WAVE>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
        }
    } else if (NSTAGE == 6) {
        // ONE slot, recycled mid-tile (MT = 8, BK = 32): 48 KB of LDS for a 256 x 128 tile with 4 waves, so TWO
        // workgroups share a CU -- one workgroup's LDS latency, barriers and, above all, its store-bound epilogue
        // (25 % of a K = 1280 GEMM) run underneath the other's MFMAs.
        issue(0, 0);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        const int co = chunk_off(0);
        for (int t = 0; t < nk; ++t) {
            const unsigned char* pAhi = smem;
            const unsigned char* pAlo = smem + A_PLANE;
            const unsigned char* pWhi = smem + NPA * A_PLANE;
            const unsigned char* pWlo = pWhi + W_PLANE;
            half8 wh[4], wl[4], ah[MT], al[MT];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wh[i] = *(const half8*)(pWhi + w_row + i * 16 * ROWB + co);
                if (SPLIT == 3) wl[i] = *(const half8*)(pWlo + w_row + i * 16 * ROWB + co);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                ah[i] = *(const half8*)(pAhi + a_row + i * 16 * ROWB + co);
                if (SPLIT == 3) al[i] = *(const half8*)(pAlo + a_row + i * 16 * ROWB + co);
            }
            const bool dma = (t + 1 < nk) && DBG != 1;
            const int64_t koff = (int64_t)(t + 1) * BK;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (DBG != 2) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        floatx4 c = acc[mt][nt];
                        if (SPLIT == 3) {
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[nt], ah[mt], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt], al[mt], c, 0, 0, 0);
                        }
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt], ah[mt], c, 0, 0, 0);
                    }
                }
                if (mt == 0) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // tile t is in registers everywhere ...
                    __builtin_amdgcn_s_barrier();                        // ... so the slot may be overwritten
                }
                if (dma && mt >= 1 && mt <= 4) {
                    constexpr int PP = (PER_WAVE + 3) / 4;
#pragma unroll
                    for (int j = (mt - 1) * PP; j < (mt * PP < PER_WAVE ? mt * PP : PER_WAVE); ++j)
                        glds16(src[j] + koff * kmul(j), smem + dst_off[j]);
                }
            }
            wait_vmcnt<0>();                                             // tile t+1 landed (this wave's share)
            __builtin_amdgcn_s_barrier();
        }
    } else if constexpr (NSTAGE == 5 && MX) {
        // UNITS: one 128-byte line per row -- even units fp16 lines (k 0-31 | k 32-63 of hi), odd units e4m3 lines (hi8 k 0-63 | lo8 k 0-63)
        // of the same 64 k-elements; eight units (four unit pairs = the four scale bytes of one dword) are one pass of the loop body.
        //   fragment registers: wf[i] = (chunk fq | chunk 4 + fq) of weight rows, af[i] = (chunk 4 + fq | chunk fq) of activation rows
        //   f16 unit : chunk c of a line = k 8c .. 8c + 7:  acc += W(chunk fq) . A(chunk fq) + W(chunk 4 + fq) . A(chunk 4 + fq)
        //   fp8 unit : the instruction's lane (r, q) supplies k' = 16q .. 16q + 15 with its first 16 bytes and k' = 64 + 16q .. with its
        //              second 16 (tools/micro/mx_semantics.hip, measured), so with the registers above its K = 128 is
        //              [Whi8 (k 0-63) | Wlo8 (k 0-63)] . [Alo8 (k 0-63) | Ahi8 (k 0-63)]: both corrections in one instruction.
        //   scales   : the E8M0 byte of k'-block b (32 elements) comes from lane (r, b): weight lanes read scale plane fq (hi8 first),
        //              activation lanes plane fq ^ 2 (lo8 first); one dword = the bytes of four unit pairs, op_sel picks the pair.
        // RING.  With a third fewer matrix cycles per k the two-slot loop below stops hiding the operand stream: a unit's DMA (64 KiB per
        // CU) needs ~1.3 us to land when every CU streams (profiles/r05_gemm_mx_probes.log: the DMA skeleton alone 534 us of a 700-us
        // launch, the multiplies without DMA 545), and the staggered two-slot schedule gives a piece half a unit to a unit.  Here the
        // 160 KiB hold FIVE half-slots of 32 KiB -- the 256 activation rows or the 256 weight rows of one unit; unit u lives in
        // half-slots (2u) % 5 and (2u + 1) % 5 -- and ONE barrier per unit: behind barrier t (unit t + 1 has landed, unit t's
        // half-slots are free) every wave requests its pieces of W(t + 2), which lands in A(t)'s half-slot and has one unit to
        // arrive, and of A(t + 3), which lands in W(t)'s and has two.  The wave groups stay staggered by half a unit, by program order:
        // waves 0-3 run a unit between two barriers, waves 4-7 the second half of one unit and the first half of the next.
        const bool grpB = __builtin_amdgcn_readfirstlane(tid) >= (NWAVE / 2) * 64;
        constexpr int MH = MT / 2;
        constexpr int HALF_SLOT = 32768, NHS = 5;
        static_assert(BN == 256 && PER_WAVE * NWAVE == (BM + BN) / 8, "one staging instruction per 8 rows");
        constexpr int PW_A = BM / 8 / NWAVE, PW_W = BN / 8 / NWAVE;        // pieces per wave and unit: activation rows, weight rows
        constexpr bool DMA_ONLY = DBG == 2 || (DBG >= 12 && DBG <= 18);
        // probes 20-22 (round 6, tools/probe_gemm_mx.py): what a 3-byte operand (fp16 hi + lo8; hi8 formed in registers) would cost and buy, BEFORE
        // building it -- timing only, wrong numbers.  HALF_E4: the e4m3 units fetch half their lines (the stream of a 3-byte operand).
        // CONVERT: the e4m3 units build their operands the way that kernel would: the fp16 lines of the unit before (two ds_read_b128 per
        // fragment), eight v_cvt_scalef32_pk_fp8_f16 + the scale's extraction per fragment, and ONE ds_read_b128 of lo8 bytes.
        constexpr bool HALF_E4 = DBG == 20 || DBG == 21, CONVERT = DBG == 21 || DBG == 22;
        // probes 16-19 (tools/probe_gemm_mx.py): where do the operand stream's lines come from?  16 / 19: the activation rows of every tile wrapped
        // into rows 0..255 (always L2-resident; 19: in the full kernel -- wrong numbers, right timing), 17: into rows 0..8191 (infinity-cache
        // resident), 18: activation and weight rows both wrapped into 0..255
        constexpr int WRAP_A = (DBG == 16 || DBG == 18 || DBG == 19) ? 255 : DBG == 17 ? 8191 : 0, WRAP_W = DBG == 18 ? 255 : 0;
        intx8 wf[4], af[MH];
        if (DBG == 9 || DMA_ONLY) {
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[i] = intx8{lane, i, lane, 1, 2, lane, 3, 4};
#pragma unroll
            for (int i = 0; i < MH; ++i) af[i] = intx8{i, lane, 5, lane, 6, 7, lane, 8};
        }
        int sw[4], sa[MT];
        // DMA sources without per-instruction vector registers: a staging instruction covers 8 rows x 128 bytes; a lane's source is
        // base + (first row of the block) * pitch + unit * 128  -- all scalar -- plus  (lane / 8) * pitch + (chunk its LDS position
        // holds) * 16, and the chunk permutation ^ ((row >> 1) & 7) depends on the block only through its parity: two vector registers
        // per operand instead of the general kernel's 64-bit pointers.  Rows come in whole blocks (the launcher checks M % 8 == 0,
        // N % 8 == 0 and that both images are below 4 GiB), so the clamp of the last tile is scalar too.
        unsigned va[2], vw[2];
        {
            const int r = lane >> 3, pos8 = lane & 7;
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                va[par] = (unsigned)r * (unsigned)g.lda * 2u + (unsigned)((pos8 ^ (4 * par + (r >> 1))) * 16);
                vw[par] = (unsigned)r * (unsigned)g.ldw_mx * 2u + (unsigned)((pos8 ^ (4 * par + (r >> 1))) * 16);
            }
        }
        const int k0_bytes = k0_units * 128;
        // piece j of this wave for unit u: j < PW_W weight rows, else activation rows
        auto dma_piece = [&](int j, int u) {
            const bool isW = j < PW_W;
            const int blk = isW ? wave * PW_W + j : wave * PW_A + (j - PW_W);
            if (HALF_E4 && (u & 1) && (j & 1)) return;               // probe: an e4m3 unit's lines are 64 bytes per row
            int row = (isW ? bn : bm) + blk * 8;
            const int lim = (isW ? g.N : g.M) - 8;
            row = row < lim ? row : lim;
            if (WRAP_A && !isW) row &= WRAP_A;
            if (WRAP_W && isW) row &= WRAP_W;
            const char* base = (const char*)(isW ? Whi : Ahi) + (int64_t)row * (isW ? g.ldw_mx : g.lda) * 2 + k0_bytes + (int64_t)u * 128;   // a unit is 128 bytes of a row
            unsigned o = isW ? vw[blk & 1] : va[blk & 1];
            asm volatile("" : "+s"(base), "+v"(o));  // keep the address a scalar base + a 32-bit vector offset (left alone, the zero-extended offsets are hoisted as 64-bit pairs)
            const int hs = (2 * u + (isW ? 1 : 0)) % NHS;
            constexpr int AUX = DBG == 12 ? 1 : DBG == 13 ? 2 : DBG == 14 ? 16 : DBG == 15 ? 17 : 0;
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(base + o), (LDS_AS void*)(smem + hs * HALF_SLOT + blk * 1024), 16, 0, AUX);
        };
        // the batch every wave requests behind barrier k: its pieces of W(k + 2), then of A(k + 3); pieces [j0, j1) of PER_WAVE
        auto dma_batch = [&](int k, int j0, int j1) {
            if (DBG == 1 && k > 0) return;
#pragma unroll
            for (int j = j0; j < j1; ++j) {
                if (j < PW_W) { if (k + 2 < nk) dma_piece(j, k + 2); }
                else if (k + 3 < nk) dma_piece(j, k + 3);
            }
        };
        const unsigned char* a_s = (const unsigned char*)g.a_mxs;
        const unsigned char* w_s = (const unsigned char*)g.w_mxs;
        auto load_scales_into = [&](int grp, int* sw, int* sa) {    // grp: index of the 8-unit group within the whole K range
            // addresses are rebuilt from the lane's row index at every call (12 loads per 8 units): kept across the loop they are 24
            // registers this kernel does not have
            int frx = fr;
            asm volatile("" : "+v"(frx));
            const unsigned wstep = 4u * (unsigned)g.ldw_s, astep = 4u * (unsigned)g.lda_s;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r = bn + wn * 64 + i * 16 + frx;
                r = r < g.N ? r : g.N - 1;
                sw[i] = *(const int*)(w_s + ((unsigned)r * wstep + (unsigned)fq * (unsigned)g.ldw_s + 4u * (unsigned)grp));
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                int r = bm + wm * WROWS + i * 16 + frx;
                r = r < g.M ? r : g.M - 1;
                sa[i] = *(const int*)(a_s + ((unsigned)r * astep + (unsigned)(fq ^ 2) * (unsigned)g.lda_s + 4u * (unsigned)grp));
            }
        };
        auto load_scales = [&](int grp) { load_scales_into(grp, sw, sa); };
        // the compiler's own wait for the scale loads lands HERE, next to a counted wait this wave executes anyway (it cannot see the
        // counted waits of this loop: left to itself it would wait for vmcnt(0) at the first fp8 instruction)
        auto touch_scales = [&]() {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(sw[i]));
#pragma unroll
            for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(sa[i]));
        };
        auto half_of = [&](const intx8& v, int h) -> half8 {
            return __builtin_bit_cast(half8, h ? intx4{v[4], v[5], v[6], v[7]} : intx4{v[0], v[1], v[2], v[3]});
        };
        auto cat = [&](half8 lo, half8 hi) -> intx8 {
            const intx4 a = __builtin_bit_cast(intx4, lo), b = __builtin_bit_cast(intx4, hi);
            return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
        };
        // fragment addresses: (row of the lane) * 128 + chunk * 16 inside a half-slot; chunk fq and chunk 4 + fq, permuted by ^ ((row >> 1) & 7)
        const int sx = (fr >> 1) & 7;
        const int fa0 = (wm * WROWS + fr) * 128 + ((fq ^ sx) * 16), fa1 = (wm * WROWS + fr) * 128 + (((4 + fq) ^ sx) * 16);
        const int fw0 = (wn * 64 + fr) * 128 + ((fq ^ sx) * 16), fw1 = (wn * 64 + fr) * 128 + (((4 + fq) ^ sx) * 16);
        auto read_w = [&](int u) {
            const unsigned char* b = smem + ((2 * u + 1) % NHS) * HALF_SLOT;
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[i] = cat(*(const half8*)(b + fw0 + i * 2048), *(const half8*)(b + fw1 + i * 2048));
        };
        auto read_a = [&](int u, int mh) {
            const unsigned char* b = smem + ((2 * u) % NHS) * HALF_SLOT;
#pragma unroll
            for (int i = 0; i < MH; ++i) af[i] = cat(*(const half8*)(b + fa1 + (mh * MH + i) * 2048), *(const half8*)(b + fa0 + (mh * MH + i) * 2048));
        };
        // MH m-tiles of MFMAs of unit kind KIND (0: f16 unit; 1 .. 4: fp8 unit, scale byte KIND - 1); the pieces of batch k (k < -1: none)
        // go out between the m-tiles
        auto mfma_half = [&](auto kind_c, int mh, int k, bool dma) {
            constexpr int KIND = decltype(kind_c)::value;
#pragma unroll
            for (int mt = 0; mt < MH; ++mt) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    floatx4 c = acc[mh * MH + mt][nt];
                    if (DMA_ONLY || (DBG == 10 && KIND != 0) || (DBG == 11 && KIND == 0)) continue;
                    if (KIND == 0) {
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(half_of(wf[nt], 0), half_of(af[mt], 1), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(half_of(wf[nt], 1), half_of(af[mt], 0), c, 0, 0, 0);
                    } else {
                        constexpr int SEL = KIND > 0 ? KIND - 1 : 0;
                        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[nt], af[mt], c, 0, 0, SEL, sw[nt], SEL, sa[mh * MH + mt]);
                    }
                    acc[mh * MH + mt][nt] = c;
                }
                // the weight pieces (one unit to land) all go out behind the first m-tile, the activation pieces (two units) behind the others
                if (dma) {
                    if (MH >= 2) dma_batch(k, mt == 0 ? 0 : PW_W + ((mt - 1) * PW_A) / (MH - 1), mt == 0 ? PW_W : PW_W + (mt * PW_A) / (MH - 1));
                    else dma_batch(k, 0, PER_WAVE);
                }
            }
        };
        // probe CONVERT: fragment i of an e4m3 unit u from the fp16 lines of unit u - 1 (16 values of this lane -> 16 e4m3 bytes) + 16 lo8 bytes of unit u
        auto conv_frag = [&](const unsigned char* b16, const unsigned char* b8, int off0, int off1, int word, int sel, bool wside) -> intx8 {
            typedef _Float16 h2v __attribute__((ext_vector_type(2)));
            typedef short s2v __attribute__((ext_vector_type(2)));
            const half8 h0 = *(const half8*)(b16 + off0), h1 = *(const half8*)(b16 + off1);
            const intx4 lo = *(const intx4*)(b8 + off0);
            const float sc = __builtin_bit_cast(float, (int)(__builtin_amdgcn_ubfe((unsigned)word, 8u * (unsigned)sel, 8u) << 23));
            s2v c[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const half8 h = q ? h1 : h0;
                c[2 * q] = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(c[2 * q], h2v{h[0], h[1]}, sc, false);
                c[2 * q] = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(c[2 * q], h2v{h[2], h[3]}, sc, true);
                c[2 * q + 1] = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(c[2 * q + 1], h2v{h[4], h[5]}, sc, false);
                c[2 * q + 1] = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(c[2 * q + 1], h2v{h[6], h[7]}, sc, true);
            }
            const intx4 hi8 = intx4{__builtin_bit_cast(int, c[0]), __builtin_bit_cast(int, c[1]), __builtin_bit_cast(int, c[2]), __builtin_bit_cast(int, c[3])};
            return wside ? __builtin_shufflevector(hi8, lo, 0, 1, 2, 3, 4, 5, 6, 7) : __builtin_shufflevector(lo, hi8, 0, 1, 2, 3, 4, 5, 6, 7);
        };
        auto conv_w = [&](int u, int sel) {
            const unsigned char* b8 = smem + ((2 * u + 1) % NHS) * HALF_SLOT;
            const unsigned char* b16 = smem + ((2 * u + 4) % NHS) * HALF_SLOT;       // W(u - 1): (2 (u - 1) + 1) % 5
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[i] = conv_frag(b16 + i * 2048, b8 + i * 2048, fw0, fw1, sw[i], sel, true);
        };
        auto conv_a = [&](int u, int mh, int sel) {
            const unsigned char* b8 = smem + ((2 * u) % NHS) * HALF_SLOT;
            const unsigned char* b16 = smem + ((2 * u + 3) % NHS) * HALF_SLOT;       // A(u - 1): (2 (u - 1)) % 5
#pragma unroll
            for (int i = 0; i < MH; ++i) af[i] = conv_frag(b16 + (mh * MH + i) * 2048, b8 + (mh * MH + i) * 2048, fa0, fa1, sa[mh * MH + i], sel, false);
        };
        auto kind_of = [](auto r_c) { constexpr int R = decltype(r_c)::value; return std::integral_constant<int, (R & 1) ? 1 + (R >> 1) : 0>{}; };
        auto frag_p0 = [&](int t) { if (!DMA_ONLY && DBG != 9) { read_w(t); read_a(t, 0); } };
        auto frag_p1 = [&](int t) { if (!DMA_ONLY && DBG != 9) read_a(t, 1); };
        // everything but the youngest batch's activation pieces (and, where said, the scale loads behind them) has landed
        auto wait_landed = [&](int t, bool scales_behind) {
            if (t + 2 < nk && !(DBG == 1 && t > 1)) {
                if (HALF_E4 && (t & 1)) { if (scales_behind) wait_vmcnt<PW_A / 2 + 4 + MT>(); else wait_vmcnt<PW_A / 2>(); }   // the youngest batch carries A(t + 2): half its pieces
                else if (scales_behind) wait_vmcnt<PW_A + 4 + MT>(); else wait_vmcnt<PW_A>();
            }
            else wait_vmcnt<0>();
        };
        // ---- prologue: A(0), W(0), A(1) (the batches "-3" and "-2" of the steady state); unit 0 landed behind the barrier
#pragma unroll
        for (int j = PW_W; j < PER_WAVE; ++j) dma_piece(j, 0);
#pragma unroll
        for (int j = 0; j < PW_W; ++j) dma_piece(j, 0);
        if (nk > 1) {
#pragma unroll
            for (int j = PW_W; j < PER_WAVE; ++j) dma_piece(j, 1);
            wait_vmcnt<PW_A>();
        } else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                                  // barrier "-1"
        // One loop body for both groups (two loops, one per group, met at the epilogue with the accumulators in different registers:
        // 70 spilled).  Waves 0-3: between barrier t - 1 and barrier t the whole unit t, batch t - 1 under its first half.  Waves 4-7,
        // half a unit behind: their barrier t sits in the MIDDLE of unit t -- between two barriers they run the second half of unit
        // t - 1, with batch t - 1 under it, and the first half of unit t (whose first unit carries batch -1).  Scale words of a group:
        // waves 0-3 request them at the top of its first unit, waves 4-7 at the top of the previous group's last second half (into
        // registers of their own: that half still multiplies with the old ones) -- both IN FRONT of a batch, so that the counted
        // wait before the next barrier covers them.
        auto unit = [&](auto r_c, int t) {
            constexpr int R = decltype(r_c)::value;
            // scale words of this group: requested at the top of its first unit.  Waves 0-3: in front of batch t - 1 -- the counted wait
            // before barrier t covers them.  Waves 4-7: BEHIND batch t - 1 (it went out under the previous second half; the first group
            // aside) -- they stay in flight across barrier t and are waited for in front of the group's first fp8 instructions, behind
            // batch t (no register of their own for a prefetch: 16 accumulators were spilled for it)
            if (R == 0) load_scales((k0_units + t) >> 3);
            if (R == 1 && grpB) {
                if (t + 2 < nk) wait_vmcnt<PER_WAVE>(); else wait_vmcnt<0>();
                touch_scales();
            }
            if constexpr (CONVERT && (R & 1)) { conv_w(t, R >> 1); conv_a(t, 0, R >> 1); } else frag_p0(t);
            mfma_half(kind_of(r_c), 0, t - 1, !grpB || t == 0);
            if constexpr (CONVERT && (R & 1)) conv_a(t, 1, R >> 1); else frag_p1(t);
            if (grpB) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // every fragment of unit t is in registers: its half-slots may go
                if (R == 0 && t != 0) wait_landed(t, true); else wait_landed(t, false);
                __builtin_amdgcn_s_barrier();                          // barrier t
            }
            mfma_half(kind_of(r_c), 1, t, grpB);
            if (!grpB) {
                wait_landed(t, false);
                if (R == 0) touch_scales();
                __builtin_amdgcn_s_barrier();                          // barrier t
            }
        };
        for (int tg = 0; tg < nk; tg += 8) {                         // nk is even (K % 64 == 0; parts are whole groups)
            unit(std::integral_constant<int, 0>{}, tg);
            unit(std::integral_constant<int, 1>{}, tg + 1);
            if (tg + 2 >= nk) break;
            unit(std::integral_constant<int, 2>{}, tg + 2);
            unit(std::integral_constant<int, 3>{}, tg + 3);
            if (tg + 4 >= nk) break;
            unit(std::integral_constant<int, 4>{}, tg + 4);
            unit(std::integral_constant<int, 5>{}, tg + 5);
            if (tg + 6 >= nk) break;
            unit(std::integral_constant<int, 6>{}, tg + 6);
            unit(std::integral_constant<int, 7>{}, tg + 7);
        }
    } else if (NSTAGE == 5) {
        // Two slots, wave groups staggered by half a K-tile (MT = 8, BK = 32).  Waves 0..3 (group A) and 4..7
        // (group B) share SIMDs pairwise (wave w and w+4).  Each K-tile has two phases per wave,
        //   P0: read W + first-half activation fragments, 48 MFMAs, then request the second-half fragments
        //   P1: 48 MFMAs on the second half
        // and group B runs one phase behind group A: in every half-tile slot one wave of a SIMD pair is in
        // its pure-MFMA phase while its partner sits in LDS latency, so the matrix pipe always has work
        // (microarch guide, "Two waves per SIMD", item 9).  Raw s_barrier at every phase boundary; the group
        // predicate goes through readfirstlane so the extra barriers are provably wave-uniform.
        const bool grpB = __builtin_amdgcn_readfirstlane(tid) >= (NWAVE / 2) * 64;
        constexpr bool PRIO = (DBG == 7);                              // probe: s_setprio around the MFMA groups
        if (DBG == 8 && grpB) __builtin_amdgcn_s_setprio(1);          // probe: static priority for the younger half
        const int co = chunk_off(0);
        static_assert(NSTAGE != 5 || (MT % 2) == 0, "staggered loop: the m-tiles of a wave split into two halves");
        constexpr int MH = MT >= 2 ? MT / 2 : 1;                                     // m-tiles per phase (4 for the 256-row tile, 3 for the 192-row one)
        half8 wh[4], wl[4], ah[MH], al[MH];
        auto mfma_half = [&](int mh, int tn, int sn) {           // MH m-tiles; optional DMA of tile tn into slot sn
            const bool dma = tn >= 0 && tn < nk && DBG != 1;
            const int64_t koff = (int64_t)tn * BK;
            unsigned char* nxt = smem + sn * STAGE;
#pragma unroll
            for (int mt = 0; mt < MH; ++mt) {
                if (DBG != 2) {
                    if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        floatx4 c = acc[mh * MH + mt][nt];
                        if (SPLIT == 3) {
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[nt], ah[mt], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt], al[mt], c, 0, 0, 0);
                        }
                        acc[mh * MH + mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt], ah[mt], c, 0, 0, 0);
                    }
                    if (PRIO) __builtin_amdgcn_s_setprio(0);
                }
                if (dma) {                                        // this wave's DMA pieces, spread over the MFMA groups of the phase
#pragma unroll
                    for (int j = (mt * PER_WAVE) / MH; j < ((mt + 1) * PER_WAVE) / MH; ++j)
                        glds16(src[j] + koff * kmul(j), nxt + dst_off[j]);
                }
            }
        };
        auto read_a = [&](const unsigned char* cur, int mh) {
#pragma unroll
            for (int i = 0; i < MH; ++i) {
                ah[i] = a_frag(cur, mh * MH + i, 0, co);
                if (SPLIT == 3) al[i] = a_frag(cur, mh * MH + i, 1, co);
            }
        };
        auto read_w = [&](const unsigned char* cur) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wh[i] = w_frag(cur, i, 0, co);
                if (SPLIT == 3) wl[i] = w_frag(cur, i, 1, co);
            }
        };
        // prologue: tile 0 resident for everyone; group B also launches its share of tile 1 (its "P1(-1)").
        // PERSIST, later tiles: both were requested before the previous epilogue and waited for inside it (every wave
        // its own share): only the barrier is left.
        if (!PERSIST || first_tile) {
            issue(0, 0);
            if (grpB && nk > 1) { issue(1, 1); wait_vmcnt<PER_WAVE>(); }   // B's share of tile 1 rides behind tile 0
            else wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        if (DBG == 4) tr1 = wall_clock64();
        if (grpB) __builtin_amdgcn_s_barrier();                  // B starts one phase late
        for (int t = 0; t < nk; ++t) {
            const unsigned char* cur = smem + (t & 1) * STAGE;
            // ---- P0
            read_w(cur);
            read_a(cur, 0);
            mfma_half(0, grpB ? -1 : t + 1, (t + 1) & 1);        // group A streams tile t+1 under its P0
            read_a(cur, 1);                                       // second-half fragments: in flight across the barrier
            // B's share of tile t+1 (issued one phase ago).  PERSIST: at t = 0 of a later tile that share landed during the
            // previous epilogue, and a vmcnt(0) here would wait for that epilogue's stores (vmcnt retires in issue order)
            if (grpB && !(PERSIST && !first_tile && t == 0)) wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // every fragment of tile t is in registers
            __builtin_amdgcn_s_barrier();
            // ---- P1
            mfma_half(1, grpB ? t + 2 : -1, t & 1);               // group B streams tile t+2 into the slot it just left
            if (!grpB) wait_vmcnt<0>();                           // A's share of tile t+1
            __builtin_amdgcn_s_barrier();
        }
        if (!grpB) __builtin_amdgcn_s_barrier();                  // match B's extra leading barrier
    } else {
        // R-slot ring (R = 3, or NSTAGE - 10 for the deep rings of the small-grid launches): tile t computes from slot t % R while
        // tiles t+1 .. t+R-1 are in flight / landing.  The deep form is for grids that leave most of the chip idle (one image: the
        // CLIP towers at M = 581 are 40 tiles of 128^2): such a workgroup is alone on its CU, its weights come cold from HBM
        // (~2 us per round trip against 0.37 us of MFMAs per K-tile), and the time of the launch is K-tiles x latency / depth.
        constexpr int R = NSTAGE >= 13 ? NSTAGE - 10 : 3;
        static_assert((R - 1) * PER_WAVE < 64, "vmcnt is a 6-bit counter");
#pragma unroll
        for (int i = 0; i < R - 1; ++i)
            if (i < nk) issue(i, i);
        // tile 0 landed; the (up to R - 2) younger tiles stay in flight
        if (nk >= R - 1) wait_vmcnt<(R - 2) * PER_WAVE>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        int slot = 0;
        for (int t = 0; t < nk; ++t) {
            const int sp = slot == 0 ? R - 1 : slot - 1;
            // slot sp held tile t-1: all waves left it at the last barrier
            compute(slot, t + R - 1 < nk ? t + R - 1 : -1, sp);
            // tile t+1 must have landed for every wave before anyone reads it; the tiles behind it may stay in flight
            const int younger = nk - 2 - t;                      // tiles issued after tile t+1 (at most R - 2)
            if (younger >= R - 2) wait_vmcnt<(R - 2) * PER_WAVE>();
            else if (R > 3 && younger == R - 3) wait_vmcnt<(R > 3 ? R - 3 : 0) * PER_WAVE>();
            else if (R > 4 && younger == R - 4) wait_vmcnt<(R > 4 ? R - 4 : 0) * PER_WAVE>();
            else wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            slot = slot + 1 == R ? 0 : slot + 1;
        }
    }

    if (DBG == 4) tr2 = wall_clock64();
    int m_lim = g.M;
    if (NSTAGE == 5 && !PERSIST && kparts > 1) {
        // K-parts of one tile reduce along a chain: part k waits for part k-1's slab, adds it to its accumulators
        // and (unless it is the last part, which runs the epilogue) publishes the running sum as its own slab;
        // the order of the fp32 additions is fixed.  (Written as "owner reads all slabs | others store", two
        // exclusive branches over the 128 accumulator registers, hipcc spilled accumulators inside the main loop.)
        // Slabs are in accumulator layout [(wave * MT + mt) * 4 + nt][lane] float4: every store / load instruction
        // moves one contiguous KiB.  Hand-off per MI355X_MICROARCH.md "Valid forms": storing waves drain,
        // barrier, lane-0 agent release, drain, relaxed agent flag store; the reader polls relaxed, takes one
        // agent acquire, drains, barrier, then plain loads.  The poll is bounded: a lost partner must not hang
        // the device (cvlm_debug_gemm_tail_errors counts give-ups).
        constexpr int SLAB = BM * BN;                                // floats
        // the lane offset is made opaque here so that no slab address is computed (and kept alive) above the
        // main loop: hoisted, those 64-bit addresses pushed the loop over its register budget
        int lane_x = lane;
        asm volatile("" : "+v"(lane_x));
        if (kpart > 0) {
            // (the mx kernel's ring is the whole dynamic LDS of a workgroup: its flag is the last word of the ring, which the loop has left)
            int* gave_up_p;
            if constexpr (MX) gave_up_p = (int*)(smem + 5 * 32768 - 16);
            else { __shared__ int tail_gave_up_word; gave_up_p = &tail_gave_up_word; }
            int& tail_gave_up = *gave_up_p;
            if (tid == 0) {
                int spins = 0, bad = 0;
                while (__hip_atomic_load(&p.flags[tail_j * 4 + kpart - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > (1 << 22)) {
                        // give up: mark the word ABANDONED (2) so that the late producer cleans it instead of leaving a stale
                        // "ready" behind for the next launch; if the slab arrived in this very moment, take it after all
                        unsigned expect = 0u;
                        if (__hip_atomic_compare_exchange_strong(&p.flags[tail_j * 4 + kpart - 1], &expect, 2u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                                 __HIP_MEMORY_SCOPE_AGENT) || expect != 1u) {
                            atomicAdd(&p.flags[4 * 128], 1u);
                            bad = 1;
                        }
                        break;
                    }
                }
                tail_gave_up = bad;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // this workgroup is the word's only reader: put 0 back so the next launch on this workspace (stream
                // ordered behind this one) starts from a clean word -- no host-side epoch, safe under graph replay
                if (!bad) __hip_atomic_store(&p.flags[tail_j * 4 + kpart - 1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            // a partner that never arrived must not pass for a result: the whole tile becomes NaN (fails loudly downstream)
            const float poison = tail_gave_up ? __builtin_nanf("") : 0.f;
            const float4* slab = (const float4*)(p.ws + ((size_t)tail_j * 3 + kpart - 1) * SLAB) + (size_t)wave * MT * 4 * 64 + lane_x;
            float4 buf[2][4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) buf[0][nt] = slab[nt * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (mt + 1 < MT) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) buf[(mt + 1) & 1][nt] = slab[((mt + 1) * 4 + nt) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const float4 v = buf[mt & 1][nt];
                    acc[mt][nt][0] += v.x + poison; acc[mt][nt][1] += v.y + poison; acc[mt][nt][2] += v.z + poison; acc[mt][nt][3] += v.w + poison;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (kpart < kparts - 1) {
            float4* slab = (float4*)(p.ws + ((size_t)tail_j * 3 + kpart) * SLAB) + (size_t)wave * MT * 4 * 64;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const floatx4 c = acc[mt][nt];
                    slab[(mt * 4 + nt) * 64 + lane_x] = make_float4(c[0], c[1], c[2], c[3]);
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // 0 -> 1 (ready); a consumer that gave up left 2 (abandoned): put the word back to 0, nobody will read this slab
                unsigned expect = 0u;
                if (!__hip_atomic_compare_exchange_strong(&p.flags[tail_j * 4 + kpart], &expect, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT))
                    __hip_atomic_store(&p.flags[tail_j * 4 + kpart], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            m_lim = 0;
        }
    }
    if (SK && kparts > 1) {
        constexpr int SLAB4 = BM * BN / 4;                               // float4 per slab, accumulator layout as above
        int lane_x = lane;
        asm volatile("" : "+v"(lane_x));
        float4* mine = (float4*)p.ws + ((size_t)tail_j * kparts + kpart) * SLAB4 + (size_t)wave * MT * 4 * 64;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const floatx4 c = acc[mt][nt];
                mine[(mt * 4 + nt) * 64 + lane_x] = make_float4(c[0], c[1], c[2], c[3]);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __shared__ int sk_last;
        __syncthreads();                                                 // every wave's slab stores have been issued and retired
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const unsigned old = __hip_atomic_fetch_add(&p.flags[SK_COUNTER0 + tail_j], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == (unsigned)(kparts - 1);
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                // every part has counted itself: the word goes back to 0 for the next launch on this workspace
                __hip_atomic_store(&p.flags[SK_COUNTER0 + tail_j], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            sk_last = last;
        }
        __syncthreads();
        if (!sk_last) return;
        const float4* slab = (const float4*)p.ws + (size_t)tail_j * kparts * SLAB4 + (size_t)wave * MT * 4 * 64 + lane_x;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                float4 t = slab[(mt * 4 + nt) * 64];
                for (int q = 1; q < kparts; ++q) {                       // index order: the same bits whoever arrived last
                    const float4 v = slab[(size_t)q * SLAB4 + (mt * 4 + nt) * 64];
                    t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
                }
                acc[mt][nt] = floatx4{t.x, t.y, t.z, t.w};
            }
    }
    auto trace_end = [&]() {
#ifdef CVLM_PROBES
        if (DBG == 4 && p.trace && tid == 0) {
            const unsigned long long t3 = wall_clock64();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t4 = wall_clock64();
            unsigned long long* o = p.trace + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
            o[0] = tr0; o[1] = tr1; o[2] = tr2; o[3] = t3; o[4] = t4;
            o[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
            o[6] = tr3; o[7] = tr4;
        }
#endif
    };
    if (DBG == 6) {                                                  // probe: main loop only
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    // PERSIST: request the next tile's first K-tiles now (both slots are free after the last barrier of the main loop);
    // the epilogue below works on copies of this tile's coordinates and stages through LDS behind the ring
    const int e_bm = bm, e_bn = bn;
    bool more = false;
    if (PERSIST) {
        const int vn = vblk + (int)gridDim.x;
        more = vn < p.total_blocks;
        if (more) {
            vblk = vn; set_tile(vn); issue(0, 0);
            if (__builtin_amdgcn_readfirstlane(tid) >= (NWAVE / 2) * 64 && nk > 1) issue(1, 1);    // group B: its share of K-tile 1
        }
    }
    // ---- epilogue: lane holds out[m][n..n+3], m = .. + (lane&15), n = .. + (lane>>4)*4
    const float alpha = g.alpha;
    const float oscale = g.out_scale;                                // h2 planes carry value * oscale (launcher maps 0 -> 1)
    const bool vec_f32 = ((g.ldo & 3) == 0) && ((g.stride_o & 3) == 0);
    const bool vec_res = ((g.ldr & 3) == 0) && ((g.stride_r & 3) == 0);
    const bool vec_h = ((g.ldoh & 3) == 0) && ((g.stride_oh & 3) == 0);
    // bias of this lane's 4 x 4 output columns: loaded once per tile (it was 128 scalar loads per lane inside
    // the m-tile loop: +240 us on the 32768 x 5120 GEMM)
    float bv[4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = e_bn + wn * 64 + nt * 16 + fq * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[nt][j] = (g.bias && !g.ln_stats && n + j < g.N) ? g.bias[n + j] : 0.f;
    }
    // ---- fast epilogue: each 16 x 64 accumulator slab goes through a per-wave LDS buffer so that global
    // stores (and the residual read) are whole 128- / 256-byte row segments, 16 bytes per lane.  The direct
    // store of the MFMA layout (8 bytes per lane, 32-byte row pieces) ran at 2.2 TB/s and cost 300 us on the
    // 32768 x 5120 GEMM (tools/bench_epi.py).
    if (g.ps_c2 == 0 && (g.N & 7) == 0 && (g.hm_S == 0 || ((g.hm_hd & 7) == 0 && g.hm_S >= WROWS)) && vec_f32 && vec_res &&
        ((g.ldoh & 7) == 0) && ((g.stride_oh & 7) == 0)) {
        // The activation is a compile-time constant of the body: one wave-uniform switch per tile.  (A per-value
        // runtime switch compiled to ~8 scalar branches per output and a 240 KB epilogue that missed the
        // instruction cache on every slab: 19 us of a 112 us tile, profiles/r01_gemm_probes.md.)
        // MODE 0: plain.  MODE 1: LayerNorm folded into this GEMM (ln_stats): the staged value is alpha * acc; the row
        // statistics, the column sums of the weight, bias and activation are applied on the way out of LDS, where a lane
        // owns 8 consecutive columns of one row.  MODE 2: residual in h2 planes + row statistics of the result
        // (res_hi / row_stats): the producer side of MODE 1.
        auto fast_epi = [&](auto act_c, auto mode_c) {
            constexpr int ACT = decltype(act_c)::value;
            constexpr int MODE = decltype(mode_c)::value;
            // floats per staged row: 64 + 4 pad; PERSIST: 64, the 16-byte chunks of row r permuted by ^ r instead (the ring
            // keeps its 128 KB, so the slabs of the 8 waves have to fit the last 32 KB of the CU's 160)
            constexpr int EP = PERSIST ? 64 : 68;
            constexpr int LDS_BYTES = (NSTAGE >= 13 ? NSTAGE - 10 : NSTAGE == 6 ? 1 : (NSTAGE >= 4 ? 2 : NSTAGE)) * STAGE;
            constexpr int NBUF = (!PERSIST && LDS_BYTES >= NWAVE * 2 * 16 * EP * 4) ? 2 : 1;   // two slabs in flight when LDS allows
            float* ebuf = (float*)(smem + (PERSIST ? 2 * STAGE : 0)) + wave * (NBUF * 16 * EP);
            auto sw = [&](int row, int chunk) -> int { return PERSIST ? ((chunk ^ row) << 2) : (chunk << 2); };   // float offset of a 16-byte chunk
            const int n0 = e_bn + wn * 64;
            const int64_t zo = (int64_t)z * g.stride_o, zr = (int64_t)z * g.stride_r, zh = (int64_t)z * g.stride_oh;
            // per-lane constants of the two store shapes
            const int rowf = lane >> 4, nf = n0 + (lane & 15) * 4;    // f32: 4 rows x 256 B per instruction
            const int rowh = lane >> 3, nh = n0 + (lane & 7) * 8;     // h2:  8 rows x 128 B per instruction and plane
            const int mw = e_bm + wm * WROWS;                           // first row of this wave
            int64_t hm_col = 0, hm_bstride = 0;
            int hm_b = 0, hm_t = 0;                                   // image / token of row mw + rowh
            bool hm_lo = true;                                        // ABI 12, hm_nolo: this lane's columns are in a third (q / k / v) whose lo plane nobody reads
            if (g.hm_S > 0) {                                         // head-major qkv store (8 | hd): column part once
                const int Dh = g.hm_H * g.hm_hd;
                const int which = nh / Dh, r2 = nh - which * Dh, h = r2 / g.hm_hd, d = r2 - h * g.hm_hd;
                hm_lo = ((g.hm_nolo >> which) & 1) == 0;
                hm_bstride = (int64_t)g.hm_H * g.hm_S * g.hm_hd;
                hm_col = (int64_t)which * (g.M / g.hm_S) * hm_bstride + (int64_t)h * g.hm_S * g.hm_hd + d;
                hm_b = (mw + rowh) / g.hm_S;
                hm_t = (mw + rowh) - hm_b * g.hm_S;
            }
            float cs8[8], b8[8];                                      // MODE 1: column sums / bias of this lane's 8 columns
            float ln_rs[MT][2], ln_c[MT][2];                          // MODE 1: rstd and mu * rstd of this lane's 2 * MT rows
            if (MODE == 1) {
                // all row statistics of the tile up front: one batch of 2 * MT loads of (rstd, mu * rstd) pairs per lane.  The pairs
                // are merged from the producer's per-piece statistics by cvlm_ln_stats_merge between the two launches: merged
                // inside this kernel (ten piece loads per thread, LDS exchange) the fold cost qkv / lin1 7-8 % -- the loads'
                // latency sits between main loop and epilogue with nothing to hide it; read ready-made it costs 0.4 %
                // (profiles/r03_ln_merge_probe.log).
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        int m = mw + mt * 16 + rowh + 8 * i;
                        m = m < g.M ? m : g.M - 1;
                        const float2 v = *(const float2*)(g.ln_stats + 2 * (int64_t)m);
                        ln_rs[mt][i] = v.x;
                        ln_c[mt][i] = v.y;
                    }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool okc = nh + j < g.N;
                    cs8[j] = okc ? g.ln_colsum[nh + j] : 0.f;
                    b8[j] = (okc && g.bias) ? g.bias[nh + j] : 0.f;
                }
            }
            // MODE 2: the residual planes of slab mt + 1 are requested before slab mt is staged -- asked for where they are
            // used, each of the 16 row pieces of a wave waited out a full HBM/L2 latency (proj 286 -> 358 us)
            float ps1[8], ps2[8];                                     // MODE 2: piece statistics of the last four slabs (8 row pieces per lane group)
            [[maybe_unused]] const float piece_inv_n = g.N - n0 >= 64 ? 0.015625f : 1.0f / (float)(g.N - n0 > 0 ? g.N - n0 : 1);
            half8 res_h[2][2], res_l[2][2];
            auto load_res = [&](int mt_, int buf) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    int m = mw + mt_ * 16 + rowh + 8 * i;
                    m = m < g.M ? m : g.M - 1;                        // clamped, always a valid address: masked at the use
                    const int nc = nh < g.N ? nh : 0;
                    if (g.res_mx) {                                   // hi halves of the mx image, lo from the fp16 plane beside it
                        res_h[buf][i] = *(const half8*)((const half_t*)g.res_hi + (int64_t)m * g.ldrh + ((nc >> 6) << 7) + (nc & 63));
                        res_l[buf][i] = *(const half8*)((const half_t*)g.res_lo + (int64_t)m * g.ldrl + nc);
                        continue;
                    }
                    const int64_t ro = (int64_t)m * g.ldrh + (g.res_il ? ((nc >> 5) << 6) + (nc & 31) : nc);
                    res_h[buf][i] = *(const half8*)((const half_t*)g.res_hi + ro);
                    res_l[buf][i] = g.res_il ? *(const half8*)((const half_t*)g.res_hi + ro + 32) : *(const half8*)((const half_t*)g.res_lo + ro);
                }
            };
            const bool have_res = MODE == 2 && g.res_hi != nullptr;
            if (have_res) load_res(0, 0);
            if (DBG == 4) { asm volatile("" ::"v"(bv[0][0]), "v"(bv[3][3])); tr3 = wall_clock64(); }
#pragma clang loop unroll(full)
            for (int mt = 0; mt < MT; ++mt) {
                if (DBG == 4 && mt == 1) tr4 = wall_clock64();
                if (have_res && mt + 1 < MT) load_res(mt + 1, (mt + 1) & 1);
                float* eb = ebuf + (NBUF == 2 ? (mt & 1) * (16 * EP) : 0);
                const int m0 = mw + mt * 16;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (MODE == 1) { v[j] = acc[mt][nt][j] * alpha; continue; }
                        v[j] = acc[mt][nt][j] * alpha + bv[nt][j];
                        if (ACT != ACT_NONE && ACT != ACT_ABS_POST) v[j] = apply_act(v[j], ACT);
                    }
                    *(float4*)(eb + fr * EP + sw(fr, nt * 4 + fq)) = make_float4(v[0], v[1], v[2], v[3]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (MODE == 0 && g.out_f32) {                        // forms 1 and 2 write h2 planes only (launcher contract)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = rowf + 4 * i;
                        const int m = m0 + row;
                        float4 t = *(const float4*)(eb + row * EP + sw(row, lane & 15));
                        if (m < m_lim && nf < g.N) {
                            if (g.residual) {
                                const float4 r = *(const float4*)(g.residual + zr + (int64_t)m * g.ldr + nf);
                                t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w;
                            }
                            if (ACT == ACT_ABS_POST) { t.x = fabsf(t.x); t.y = fabsf(t.y); t.z = fabsf(t.z); t.w = fabsf(t.w); }
                            if (DBG == 3) asm volatile("" ::"v"(t.x), "v"(t.y), "v"(t.z), "v"(t.w));
                            else *(float4*)(g.out_f32 + zo + (int64_t)m * g.ldo + nf) = t;
                        }
                    }
                }
                if (MODE != 0 || g.out_hi) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int row = rowh + 8 * i;
                        const int m = m0 + row;
                        const float4 t0 = *(const float4*)(eb + row * EP + sw(row, (lane & 7) * 2));
                        const float4 t1 = *(const float4*)(eb + row * EP + sw(row, (lane & 7) * 2 + 1));
                        float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                        if (MODE == 2) {
                            const bool live = m < m_lim && nh < g.N;
                            if (live && have_res) {
                                const half8 rh = res_h[mt & 1][i], rl = res_l[mt & 1][i];
#pragma unroll
                                for (int j = 0; j < 8; ++j) v[j] += ((float)rh[j] + (float)rl[j]) * g.res_scale;
                            }
                            if (g.row_stats) {
                                // the 8 lanes of a row piece: three exchange steps for the sum, the piece mean, three more for
                                // the centred squares; every lane ends with both
                                float s1 = 0.f, s2 = 0.f;
                                if (live) {
#pragma unroll
                                    for (int j = 0; j < 8; ++j) s1 += v[j];
                                }
                                s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64); s1 += __shfl_xor(s1, 4, 64);
                                const float pm = s1 * piece_inv_n;
                                if (live) {
#pragma unroll
                                    for (int j = 0; j < 8; ++j) { const float d = v[j] - pm; s2 = fmaf(d, d, s2); }
                                }
                                s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64); s2 += __shfl_xor(s2, 4, 64);
                                ps1[(mt & 3) * 2 + i] = s1; ps2[(mt & 3) * 2 + i] = s2;
                            }
                        }
                        if (MODE == 1) {
                            const float rstd = ln_rs[mt][i], c = ln_c[mt][i];      // c = mu * rstd
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], rstd, fmaf(-c, cs8[j], b8[j]));
                            if (ACT == ACT_GELU) {                         // two values per instruction (common.h)
#pragma unroll
                                for (int j = 0; j < 8; j += 2) {
                                    const f32x2_t y = gelu_erf2(f32x2_t{v[j], v[j + 1]});
                                    v[j] = y.x; v[j + 1] = y.y;
                                }
                            } else if (ACT != ACT_NONE && ACT != ACT_ABS_POST) {
#pragma unroll
                                for (int j = 0; j < 8; ++j) v[j] = apply_act(v[j], ACT);
                            }
                        }
                        if (m < m_lim && nh < g.N) {
                            if (MODE == 0 && g.residual) {
                                const float* r = g.residual + zr + (int64_t)m * g.ldr + nh;
                                const float4 r0 = *(const float4*)r, r1 = *(const float4*)(r + 4);
                                v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
                                v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
                            }
                            if (DBG == 5) {                                // probe: staging only, no split / stores
                                asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
                                continue;
                            }
                            typedef unsigned u32x4_s __attribute__((ext_vector_type(4)));
                            u32x4_s hi4, lo4;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                float u0 = v[2 * j] * oscale, u1 = v[2 * j + 1] * oscale;
                                if (ACT == ACT_ABS_POST) { u0 = fabsf(u0); u1 = fabsf(u1); }
                                unsigned a, b2;
                                split_h2_pk(u0, u1, a, b2);
                                hi4[j] = a; lo4[j] = b2;
                            }
                            const half8 hi = __builtin_bit_cast(half8, hi4), lo = __builtin_bit_cast(half8, lo4);
                            if (g.out_mx) {
                                // mx image (include/cvlm.h ABI 10): this lane's 8 columns are a quarter of a 32-column scale block, the
                                // four lanes of the block are a DPP quad.  E = exponent field of the block's largest |hi| (as f32, at
                                // least 103) - 7: every hi / 2^(E - 127) is below 256, every lo / 2^(E - 138) below 128 -- inside e4m3.
                                typedef _Float16 h2v __attribute__((ext_vector_type(2)));
                                typedef short s2v __attribute__((ext_vector_type(2)));
                                // (element reads go through scalars: __builtin_bit_cast of a vector ELEMENT expression reads the vector's first bytes)
                                h2v hv[4], lv[4];
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    const unsigned hj = hi4[j], lj = lo4[j];
                                    hv[j] = __builtin_bit_cast(h2v, hj);
                                    lv[j] = __builtin_bit_cast(h2v, lj);
                                }
                                h2v mx2 = __builtin_elementwise_abs(hv[0]);
#pragma unroll
                                for (int j = 1; j < 4; ++j) mx2 = __builtin_elementwise_max(mx2, __builtin_elementwise_abs(hv[j]));
                                float bmax = fmaxf((float)mx2[0], (float)mx2[1]);
                                bmax = fmaxf(bmax, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, bmax), 0xB1, 0xF, 0xF, true)));   // quad_perm [1,0,3,2]
                                bmax = fmaxf(bmax, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, bmax), 0x4E, 0xF, 0xF, true)));   // quad_perm [2,3,0,1]
                                int ex = (__builtin_bit_cast(int, bmax) >> 23) & 0xff;
                                ex = (ex < 103 ? 103 : ex) - 7;
                                const float s_hi = __builtin_bit_cast(float, ex << 23), s_lo = __builtin_bit_cast(float, (ex - 11) << 23);
                                s2v h8a = {0, 0}, h8b = {0, 0}, l8a = {0, 0}, l8b = {0, 0};
                                h8a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(h8a, hv[0], s_hi, false);
                                h8a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(h8a, hv[1], s_hi, true);
                                h8b = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(h8b, hv[2], s_hi, false);
                                h8b = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(h8b, hv[3], s_hi, true);
                                l8a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(l8a, lv[0], s_lo, false);
                                l8a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(l8a, lv[1], s_lo, true);
                                l8b = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(l8b, lv[2], s_lo, false);
                                l8b = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(l8b, lv[3], s_lo, true);
                                if (DBG == 3) { asm volatile("" ::"v"(hi), "v"(h8a), "v"(h8b), "v"(l8a), "v"(l8b)); continue; }
                                unsigned char* row = (unsigned char*)g.out_hi + ((int64_t)m * g.ldoh + ((nh >> 6) << 7)) * 2;      // the 256-byte group
                                if (DBG == 7) row = (unsigned char*)g.out_hi + tid * 256;     // probe: the same store instructions into 128 KB that stay in L2
                                const int j8 = (nh & 63) >> 3;
                                *(half8*)(row + j8 * 16) = hi;
                                *(uint2*)(row + 128 + j8 * 8) = make_uint2(__builtin_bit_cast(unsigned, h8a), __builtin_bit_cast(unsigned, h8b));
                                *(uint2*)(row + 192 + j8 * 8) = make_uint2(__builtin_bit_cast(unsigned, l8a), __builtin_bit_cast(unsigned, l8b));
                                if ((lane & 3) == 0) {
                                    unsigned char* sc = (unsigned char*)g.out_mxs + ((int64_t)m * 4 + (j8 >> 2)) * g.ldo_s + (nh >> 6);
                                    sc[0] = (unsigned char)ex;
                                    sc[2 * g.ldo_s] = (unsigned char)(ex - 11);
                                }
                                if (g.out_lo) *(half8*)((half_t*)g.out_lo + (int64_t)m * g.ldol + nh) = lo;
                                continue;
                            }
                            int64_t off = (int64_t)m * g.ldoh + (g.out_il ? ((nh >> 5) << 6) + (nh & 31) : nh);   // out_il: [m][n / 32][plane][32]
                            if (g.hm_S > 0) {                              // row part: rows advance by mt * 16 + 8 * i < hm_S
                                int tk = hm_t + mt * 16 + 8 * i, bi = hm_b;
                                while (tk >= g.hm_S) { tk -= g.hm_S; ++bi; }
                                off = hm_col + (int64_t)bi * hm_bstride + (int64_t)tk * g.hm_hd;
                            }
                            if (DBG == 7) off = tid * 8;                   // probe: the same store instructions into a few KB that stay in L2
                            if (DBG == 3) {                                // probe: all the work, no global stores
                                asm volatile("" ::"v"(hi), "v"(lo), "v"(off));
                            } else {
                                *(half8*)((half_t*)g.out_hi + zh + off) = hi;
                                if (g.out_il) *(half8*)((half_t*)g.out_hi + zh + off + 32) = lo;
                                else if (g.out_lo && hm_lo) *(half8*)((half_t*)g.out_lo + zh + off) = lo;
                            }
                        }
                    }
                }
                if (MODE == 2 && ((mt & 3) == 3 || mt == MT - 1) && g.row_stats && n0 < g.N) {
                    // The 64 row pieces of four slabs go out as ONE 512-byte store with all lanes active: lane (r, q) writes row
                    // piece q of row group r into this wave's piece plane, row_stats[piece][m] (plain stores: the consumer adds
                    // the pieces of a row in a fixed order, so the statistics are bit-reproducible and need no zeroing).
                    const int q = lane & 7;
                    float a1 = ps1[0], a2 = ps2[0];
#pragma unroll
                    for (int k = 1; k < 8; ++k) { a1 = (q == k) ? ps1[k] : a1; a2 = (q == k) ? ps2[k] : a2; }
                    // a group is four slabs, or what is left of the wave's m-tiles (192-row tile: 4 + 2)
                    const int m = mw + (mt - (mt & 3) + (q >> 1)) * 16 + rowh + 8 * (q & 1);
                    if ((q >> 1) <= (mt & 3) && m < m_lim)
                        *(float2*)(g.row_stats + 2 * ((int64_t)(n0 >> 6) * g.M + m)) = make_float2(a1, a2);
                }
                if (NBUF == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // slab fully read before it is overwritten
                // PERSIST: the next tile's K-tiles 0 / 1 (requested before this epilogue) have had two slabs' time to land; the
                // wait also covers the few stores issued so far, which drain from an empty queue -- unlike a wait at the end
                if (PERSIST && mt == 1 && more) wait_vmcnt<0>();
            }
        };
        typedef std::integral_constant<int, 0> mode0;
        typedef std::integral_constant<int, 1> mode1;
        typedef std::integral_constant<int, 2> mode2;
        if (EPI == -1 ? g.ln_stats != nullptr : EPI == 1) {              // launcher: out_hi only, act in {none, GELU, QuickGELU}
            switch (g.act) {
                case ACT_GELU: fast_epi(std::integral_constant<int, ACT_GELU>{}, mode1{}); break;
                case ACT_QUICKGELU: fast_epi(std::integral_constant<int, ACT_QUICKGELU>{}, mode1{}); break;
                default: fast_epi(std::integral_constant<int, ACT_NONE>{}, mode1{}); break;
            }
        } else if (EPI == -1 ? (g.res_hi || g.row_stats) : EPI == 2) {   // launcher: out_hi only, act none
            fast_epi(std::integral_constant<int, ACT_NONE>{}, mode2{});
        } else {
            switch (g.act) {
                case ACT_GELU: fast_epi(std::integral_constant<int, ACT_GELU>{}, mode0{}); break;
                case ACT_QUICKGELU: fast_epi(std::integral_constant<int, ACT_QUICKGELU>{}, mode0{}); break;
                case ACT_RELU: fast_epi(std::integral_constant<int, ACT_RELU>{}, mode0{}); break;
                case ACT_ABS_POST: fast_epi(std::integral_constant<int, ACT_ABS_POST>{}, mode0{}); break;
                default: fast_epi(std::integral_constant<int, ACT_NONE>{}, mode0{}); break;
            }
        }
        trace_end();
        if (!PERSIST || !more) return;
        first_tile = false;
        continue;
    }
    if (PERSIST || EPI >= 0) return;                                 // the launcher sends only LDS-staged shapes here
#pragma clang loop unroll(full)
    for (int mt = 0; mt < MT; ++mt) {
        const int m = e_bm + wm * WROWS + mt * 16 + fr;
        if (m >= m_lim) continue;
        int64_t ps_base = 0;
        if (g.ps_c2 > 0) {
            const int x = m % g.ps_w, t = m / g.ps_w;
            const int y = t % g.ps_h, b = t / g.ps_h;
            ps_base = ((int64_t)(b * 2 * g.ps_h + 2 * y) * (2 * g.ps_w) + 2 * x) * (g.ps_c2 >> 1);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = e_bn + wn * 64 + nt * 16 + fq * 4;
            if (n >= g.N) continue;
            const bool full = (n + 3 < g.N);
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[mt][nt][j] * alpha;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += bv[nt][j];
            if (g.act != ACT_NONE && g.act != ACT_ABS_POST) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], g.act);
            }
            if (g.residual) {
                const float* r = g.residual + (int64_t)z * g.stride_r + (int64_t)m * g.ldr + n;
                if (full && vec_res) {
                    const float4 rv = *(const float4*)r;
                    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (n + j < g.N) v[j] += r[j];
                }
            }
            if (g.act == ACT_ABS_POST) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fabsf(v[j]);
            }
            int64_t off_f, off_h;
            bool slow_lo = true;                                        // hm_nolo (see the LDS-staged path)
            if (g.ps_c2 > 0) {
                const int dy = n / g.ps_c2, r = n - dy * g.ps_c2;
                off_f = off_h = ps_base + (int64_t)dy * (2 * g.ps_w) * (g.ps_c2 >> 1) + r;
            } else {
                off_f = (int64_t)m * g.ldo + n;
                off_h = (int64_t)m * g.ldoh + n;
                if (g.hm_S > 0) {                                      // head-major qkv store (4 | hd: never straddles a head)
                    const int Dh = g.hm_H * g.hm_hd;
                    const int which = n / Dh, r = n - which * Dh, h = r / g.hm_hd, d = r - h * g.hm_hd;
                    const int bi = m / g.hm_S, tk = m - bi * g.hm_S;
                    off_h = ((((int64_t)which * (g.M / g.hm_S) + bi) * g.hm_H + h) * g.hm_S + tk) * g.hm_hd + d;
                    slow_lo = ((g.hm_nolo >> which) & 1) == 0;
                }
            }
            if (g.out_f32 && DBG != 3) {
                float* o = g.out_f32 + (int64_t)z * g.stride_o + off_f;
                if (full && vec_f32 && ((off_f & 3) == 0)) {
                    *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (n + j < g.N) o[j] = v[j];
                }
            }
            if (g.out_hi && DBG != 3) {
                half_t hi[4], lo[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) split_h2(v[j] * oscale, hi[j], lo[j]);
                half_t* oh = (half_t*)g.out_hi + (int64_t)z * g.stride_oh + off_h;
                half_t* ol = (g.out_lo && slow_lo) ? (half_t*)g.out_lo + (int64_t)z * g.stride_oh + off_h : nullptr;
                if (full && vec_h && ((off_h & 3) == 0)) {
                    *(half4*)oh = half4{hi[0], hi[1], hi[2], hi[3]};
                    if (ol) *(half4*)ol = half4{lo[0], lo[1], lo[2], lo[3]};
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (n + j < g.N) { oh[j] = hi[j]; if (ol) ol[j] = lo[j]; }
                }
            }
        }
    }
    return;
  }  // tile loop
}

}  // namespace cvlm_gemm_k

// The kernels that stage operands from the 128-byte-row images (ABI 6) are instantiated in gemm_il.hip, a translation unit of their
// own: the two files compile side by side (this template is 45 instantiations; one file took 3.5 minutes).  X(prefix) expands to one
// explicit-instantiation statement per kernel: `extern template` in gemm.hip, `template` in gemm_il.hip.
#define CVLM_GEMM_IL_KERNEL(PFX, ...) PFX __global__ void cvlm_gemm_k::gemm_nt_kernel<__VA_ARGS__>(const cvlm_gemm_k::GemmParams);
#define CVLM_GEMM_IL_KERNELS(PFX)                                                                    \
    CVLM_GEMM_IL_KERNEL(PFX, 3, 2, 4, 5, 32, 0, 8, false, 0, false, false, true, false)              \
    CVLM_GEMM_IL_KERNEL(PFX, 3, 2, 4, 5, 32, 0, 8, false, 1, false, false, true, false)              \
    CVLM_GEMM_IL_KERNEL(PFX, 3, 2, 4, 5, 32, 0, 8, false, 2, false, false, true, false)              \
    CVLM_GEMM_IL_KERNEL(PFX, 3, 2, 4, 5, 32, 0, 8, false, 0, false, false, true, true)               \
    CVLM_GEMM_IL_KERNEL(PFX, 3, 2, 4, 5, 32, 0, 8, false, 1, false, false, true, true)               \
    CVLM_GEMM_IL_KERNEL(PFX, 3, 2, 4, 5, 32, 0, 8, false, 2, false, false, true, true)               \
    CVLM_GEMM_IL_KERNEL(PFX, 3, 2, 4, 5, 32, 0, 8, true, 0, false, false, true, false)               \
    CVLM_GEMM_IL_KERNEL(PFX, 3, 2, 4, 5, 32, 0, 8, true, 1, false, false, true, false)               \
    CVLM_GEMM_IL_KERNEL(PFX, 3, 2, 4, 5, 32, 0, 8, true, 2, false, false, true, false)               \
    CVLM_GEMM_IL_KERNEL(PFX, 3, 2, 4, 5, 32, 0, 8, true, 0, false, false, true, true)                \
    CVLM_GEMM_IL_KERNEL(PFX, 3, 2, 4, 5, 32, 0, 8, true, 1, false, false, true, true)                \
    CVLM_GEMM_IL_KERNEL(PFX, 3, 2, 4, 5, 32, 0, 8, true, 2, false, false, true, true)                \
    CVLM_GEMM_IL_KERNEL(PFX, 3, 2, 4, 5, 32, 0, 6, false, 2, false, false, true, false)              \
    CVLM_GEMM_IL_KERNEL(PFX, 3, 2, 4, 5, 32, 0, 6, false, 2, false, false, true, true)               \
    CVLM_GEMM_IL_KERNEL(PFX, 3, 4, 2, 3, 32, 0, 4, false, -1, false, false, true, false)             \
    CVLM_GEMM_IL_KERNEL(PFX, 3, 4, 2, 3, 32, 0, 4, false, -1, false, false, true, true)              \
    CVLM_GEMM_IL_KERNEL(PFX, 3, 4, 2, 14, 32, 0, 1, false, -1, false, false, true, false)            \
    CVLM_GEMM_IL_KERNEL(PFX, 3, 4, 2, 14, 32, 0, 2, false, -1, false, false, true, false)            \
    CVLM_GEMM_IL_KERNEL(PFX, 3, 4, 2, 14, 32, 0, 2, false, -1, false, true, true, false)             \
    CVLM_GEMM_IL_KERNEL(PFX, 3, 4, 2, 14, 32, 0, 1, false, -1, false, false, true, true)             \
    CVLM_GEMM_IL_KERNEL(PFX, 3, 4, 2, 14, 32, 0, 2, false, -1, false, false, true, true)             \
    CVLM_GEMM_IL_KERNEL(PFX, 3, 4, 2, 14, 32, 0, 2, false, -1, false, true, true, true)

// The mx kernels (ABI 10: both operands mx images, staggered 256-column loop) are instantiated and launched in gemm_mx.hip, a third
// translation unit: mt = 8 (256-row tiles) or 6 (192-row tiles, h2-residual form only), epi = 1 (LayerNorm fold) or 2 (h2 residual +
// row statistics); probe (probe builds only) = DBG form of the fold kernel.  p.tail_rem / tail_split / ws / flags are set by the caller.
namespace cvlm_gemm_k { int launch_mx(GemmParams& p, int mt, int epi, int extra_blocks, int probe, hipStream_t s); }
