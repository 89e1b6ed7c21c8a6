// Split-half MFMA GEMM for gfx950:  out = post(act(alpha * A . W^T + bias) + residual)
//
// Both operands arrive as fp16 planes with K contiguous ("NT" form: activations [M][K], torch
// Linear weights [N][K]).  Workgroup tile (WM*64) x (WN*64), one wave per 64x64 sub-tile (4x4 MFMA
// 16x16x32 f16 tiles), BK = 32.  K-tiles stream through an NSTAGE-deep LDS ring filled by 16-byte
// global->LDS DMA (global_load_lds_dwordx4):
//   NSTAGE = 2: one tile ahead, __syncthreads() per K-tile (drains the DMA);
//   NSTAGE = 3: two tiles ahead, counted s_waitcnt vmcnt(N) + raw s_barrier so the newest tile's DMA
//               stays in flight across the barrier (guide §5 "Pipelining across barriers");
//   NSTAGE = 13, 14, 15: the same ring with 3, 4, 5 slots (small grids: latency of cold weights, not issue, bounds them).
// The LDS image is lane-linear (DMA constraint); bank conflicts of the ds_read_b128 fragment reads
// are removed by permuting the 16-byte chunks of each 64-byte row on the *source* address and
// applying the same involution on the read (guide §5.4 rule 21).
//
// The MFMA is issued "swapped" (A-operand = weight rows, B-operand = activation rows) so that each
// lane ends up with 4 consecutive output columns of one output row: the epilogue then stores
// 16-byte float4 / 8-byte half4 vectors instead of scalars.
//
// split == 3: acc += Whi.Ahi + Wlo.Ahi + Whi.Alo  (fp32 accumulate; ~2^-22 relative products)
// split == 1: acc += Whi.Ahi
#include "gemm_kernel.h"
using namespace cvlm_gemm_k;
CVLM_GEMM_IL_KERNELS(extern template)

// Modelled time of a tail round cut into S chained K-parts: one part's main loop + epilogue, first slab
// publish (~16 us), last read-back (~8 us), ~20 us per middle hop (read + publish), and the slabs of all `rem` tiles
// moving at once (qkv at K = 1280 with 128 tail tiles measured no gain).
static double tail_us(int S, int K, int rem) {
    return 0.0685 * K / S + 14.0 + 16.0 + 8.0 + 20.0 * (S - 2) + 0.15 * rem;   // + concurrent hand-offs (rem slabs at once)
}
// Number of K-parts for a last round of `rem` tiles (1 = leave it whole).
static int tail_parts(int rem, int K) {
    if (rem <= 0 || rem > 128) return 1;
    const int smax = 256 / rem < 4 ? 256 / rem : 4;
    int best = 1;
    double t = 0.0685 * K + 14.0;
    // a split has to win by a margin: at K = 1280 / 128 tail tiles the model calls it even and the measurement does not
    // (proj 32768 x 1280 x 1280: 312-355 us split, 303-338 us whole, tools/ab_tail.py)
    double need = 0.92 * t;
    for (int S = 2; S <= smax; ++S)
        if (K / 32 >= 4 * S && tail_us(S, K, rem) < need) { t = need = tail_us(S, K, rem); best = S; }
    return best;
}

// ---- tail-split workspace (caller-owned, include/cvlm.h): [4 KiB hand-off words][128 tiles x 3 parts of 256 x 256 f32]
constexpr size_t TAIL_FLAG_BYTES = 4096;
constexpr size_t TAIL_WS_BYTES = (size_t)128 * 3 * 256 * 256 * sizeof(float);
static_assert((4 * 128 + 2) * sizeof(unsigned) <= TAIL_FLAG_BYTES, "hand-off words fit the flag page");

extern "C" int64_t cvlm_gemm_workspace_bytes(void) { return (int64_t)(TAIL_FLAG_BYTES + TAIL_WS_BYTES); }

#ifdef CVLM_PROBES
static unsigned long long* g_trace = nullptr;
// Probe hook (not part of include/cvlm.h): device buffer of 8 x u64 per workgroup for CVLM_GEMM_VARIANT=47.
extern "C" void cvlm_debug_set_gemm_trace(void* buf) { g_trace = (unsigned long long*)buf; }
#endif

static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

extern "C" int cvlm_gemm(const cvlm_gemm_args* args, void* stream) {
    if (!args || !args->a_hi || !args->w_hi) return CVLM_E_BADARG;
    const cvlm_gemm_args& g = *args;
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || (g.K % BK_MIN) != 0) return CVLM_E_BADARG;
    if ((g.lda & 7) || (g.ldw & 7) || (g.stride_a & 7) || (g.stride_w & 7)) return CVLM_E_BADARG;
    if (g.split != 1 && g.split != 3) return CVLM_E_BADARG;
    if (g.split == 3 && ((!g.a_lo && !g.a_il && !g.a_mx) || !g.w_lo)) return CVLM_E_BADARG;
    if (!g.out_f32 && !g.out_hi) return CVLM_E_BADARG;
    if (g.ps_c2 > 0 && ((g.ps_c2 & 3) || g.ps_h <= 0 || g.ps_w <= 0)) return CVLM_E_BADARG;
    if (g.hm_S > 0 && ((g.hm_hd & 3) || g.hm_H <= 0 || (g.M % g.hm_S) || g.N != 3 * g.hm_H * g.hm_hd || !g.out_hi)) return CVLM_E_BADARG;
    const bool conv = g.conv_c > 0;
    if (conv) {
        // implicit 3x3 convolution: channels a power of two >= 32 (a K-tile never straddles a tap), rows = whole images
        if (g.conv_h <= 0 || g.conv_w <= 0 || g.conv_c < 32 || (g.conv_c & (g.conv_c - 1)) || g.K != 9 * g.conv_c ||
            g.lda != g.conv_c || (g.M % (g.conv_h * g.conv_w)) || g.batch > 1)
            return CVLM_E_UNSUPPORTED;
    }
    const bool fold = g.ln_stats != nullptr, h2res = g.res_hi != nullptr || g.row_stats != nullptr;
    if (fold || h2res) {
        // these two epilogue forms exist on the LDS-staged path only: h2 output, 8-column row pieces, one problem per launch
        if (fold && h2res) return CVLM_E_BADARG;
        if (!g.out_hi || g.out_f32 || g.residual || g.ps_c2 > 0 || g.batch > 1) return CVLM_E_BADARG;
        if ((g.N & 7) || (g.ldoh & 7) || (g.hm_S > 0 && ((g.hm_hd & 7) || g.hm_S < 128))) return CVLM_E_UNSUPPORTED;
        if (fold && (!g.ln_colsum || (g.act != ACT_NONE && g.act != ACT_GELU && g.act != ACT_QUICKGELU))) return CVLM_E_BADARG;
        if (h2res && (g.act != ACT_NONE || (g.res_hi && ((!g.res_lo && !g.res_il) || (g.ldrh & 7))))) return CVLM_E_BADARG;
    }
    const bool il_any = g.a_il || g.out_il || g.res_il;
    if (il_any) {
        // 128-byte-row images of activations (ABI 6): split-3, one problem, the LDS-staged epilogues, no head-major / pixel-shuffle store
        if (g.split != 3 || conv || g.batch > 1 || (g.out_il && g.hm_S > 0) || (g.hm_S > 0 && ((g.hm_hd & 7) || g.hm_S < 256)) || g.ps_c2 > 0 ||
            (g.N & 7) || (g.ldoh & 7) || (g.stride_oh & 7) ||
            (g.out_f32 && ((g.ldo & 3) || (g.stride_o & 3))) || (g.residual && ((g.ldr & 3) || (g.stride_r & 3))))
            return CVLM_E_UNSUPPORTED;
        if (g.a_il && (!g.w_il || (g.lda & 7) || g.lda < 2 * (int64_t)g.K)) return CVLM_E_UNSUPPORTED;
        if (g.res_il && !g.res_hi) return CVLM_E_BADARG;
    }
    if (g.a_mx || g.out_mx || g.res_mx) {
        // mx images (ABI 10): split-3, one problem, the LDS-staged epilogues; whole 64-column groups
        if (g.split != 3 || conv || g.batch > 1 || g.ps_c2 > 0 || (g.a_mx && g.a_il) || (g.out_mx && g.out_il) || (g.res_mx && g.res_il) ||
            (g.N & 7) || (g.ldoh & 7) || (g.stride_oh & 7) || (g.hm_S > 0 && ((g.hm_hd & 7) || g.hm_S < 256)) ||
            (g.out_f32 && ((g.ldo & 3) || (g.stride_o & 3))) || (g.residual && ((g.ldr & 3) || (g.stride_r & 3))))
            return CVLM_E_UNSUPPORTED;
        if (g.a_mx && (!g.a_mxs || !g.w_mx || !g.w_mxs || (g.K & 63) || (g.lda & 7) || g.lda < 2 * (int64_t)g.K || (g.ldw_mx & 7) ||
                       g.ldw_mx < 2 * (int64_t)g.K || (g.lda_s & 3) || (g.ldw_s & 3) || g.lda_s * 64 < g.K || g.ldw_s * 64 < g.K))
            return CVLM_E_BADARG;
        if (g.out_mx && (!g.out_hi || !g.out_mxs || g.hm_S > 0 || (g.N & 63) || (g.ldo_s & 3) || (g.out_lo && (g.ldol & 7))))
            return CVLM_E_BADARG;
        if (g.res_mx && (!g.res_hi || !g.res_lo || (g.ldrl & 7) || (g.ldrh & 7))) return CVLM_E_BADARG;
    }
    GemmParams p;
    p.a = g;
    if (fold || h2res) {
        // these launches have no f32 output, no f32 residual and one problem: whatever the caller left in the fields that describe
        // them must not send the kernel to the scalar epilogue, which knows neither form (it would return rc 0 and wrong numbers)
        p.a.ldo = p.a.stride_o = p.a.ldr = p.a.stride_r = p.a.stride_oh = 0;
        p.a.stride_a = p.a.stride_w = 0;
    }
    if (p.a.batch <= 0) p.a.batch = 1;
    if (p.a.out_scale == 0.f) p.a.out_scale = 1.f;
    // tuning knobs, read once per process.  A process started with CVLM_GEMM_VARIANT_LIVE=1 (tests/conftest.py,
    // tools/ab_gemm.py) re-reads them on every call so that variants can be A/B-ed and raced inside one process.
    static int tail_env = env_int("CVLM_GEMM_TAIL", 1), variant_env = env_int("CVLM_GEMM_VARIANT", 0), persist_env = env_int("CVLM_GEMM_PERSIST", 1);
    static const bool live_env = env_int("CVLM_GEMM_VARIANT_LIVE", 0) != 0;
    if (live_env) {
        tail_env = env_int("CVLM_GEMM_TAIL", 1); variant_env = env_int("CVLM_GEMM_VARIANT", 0);
        persist_env = env_int("CVLM_GEMM_PERSIST", 1);
    }
    // ---- column split (one image): a grid of 256^2 tiles a little over one round -- lin1 of a ViT-H block at M = 4096 is 16 x 20 =
    // 320 tiles on 256 CUs -- spends a second round (or a chain of K-parts with its slab traffic and hand-offs, 58 us) on a
    // quarter round of work.  Two launches instead: the columns that make exactly one round of 256^2 tiles, then the rest as
    // 128^2 tiles on the deep-ring kernel, one round of those (30-40 us).  Whole tiles both times: no slabs, no flags, and the bits
    // of every output are those of an unsplit launch.  Plain and LayerNorm-folded epilogues only (their per-column operands just
    // move with the column offset).
    static thread_local int in_colsplit = 0;
    static int colsplit_env = env_int("CVLM_GEMM_COLSPLIT", 1);
    if (live_env) colsplit_env = env_int("CVLM_GEMM_COLSPLIT", 1);
    // (For grids of several rounds whose last round is partial -- proj / lin2 of a batch of 8 -- the same split measured slower,
    // profiles/r03_colsplit_ab.log; that form is gone.)
    if (!in_colsplit && colsplit_env && g.split == 3 && !conv && g.batch <= 1 && g.hm_S == 0 && g.ps_c2 == 0 && variant_env == 0 &&
        g.M <= 4096 && !h2res && !g.a_mx && !g.out_mx) {   // (an mx image's groups and scale bytes do not move with a plain column offset: ADVICE r5)
        const int nby = (g.M + 255) / 256, nbx = (g.N + 255) / 256;
        const int c0 = nby > 0 ? 256 / nby : 0;                          // column tiles of the first launch: one round of tiles
        const int rest_ = g.N - c0 * 256;
        const long t1r = rest_ > 0 ? (long)((g.M + 127) / 128) * ((rest_ + 127) / 128) : 0;
        const bool ok = nbx > c0 && c0 * nby >= 232 && nbx * nby < 2 * 256 && rest_ >= 128 && t1r <= 256;
        const int n0 = c0 * 256, rest = g.N - n0;
        if (ok && (rest & 7) == 0 && (n0 & 63) == 0) {
            cvlm_gemm_args a1 = g, a2 = g;
            a1.N = n0;
            a2.N = rest;
            a2.w_hi = (const char*)g.w_hi + (int64_t)n0 * g.ldw * 2;
            if (g.w_lo) a2.w_lo = (const char*)g.w_lo + (int64_t)n0 * g.ldw * 2;
            if (g.w_il) a2.w_il = (const char*)g.w_il + (int64_t)n0 * g.ldw_il * 2;
            if (g.bias) a2.bias = g.bias + n0;
            if (g.ln_colsum) a2.ln_colsum = g.ln_colsum + n0;
            if (g.residual) a2.residual = g.residual + n0;
            if (g.out_f32) a2.out_f32 = g.out_f32 + n0;
            if (g.out_hi) a2.out_hi = (char*)g.out_hi + (int64_t)n0 * 2 * (g.out_il ? 2 : 1);   // image: column c0 starts 2 * c0 halves into a row
            if (g.out_lo) a2.out_lo = (char*)g.out_lo + (int64_t)n0 * 2;
            in_colsplit = 1;
            int rc = cvlm_gemm(&a1, stream);
            if (rc == 0) rc = cvlm_gemm(&a2, stream);
            in_colsplit = 0;
            return rc;
        }
    }
#ifdef CVLM_PROBES
    p.trace = g_trace;
#endif
    p.group_m = 8;
    p.tail_rem = 0; p.tail_split = 1; p.ws = nullptr; p.flags = nullptr;
    p.total_blocks = 0;
    const bool have_ws = g.workspace && g.workspace_bytes >= cvlm_gemm_workspace_bytes();
    hipStream_t s = (hipStream_t)stream;
    // interleaved weight image (ABI 6): used by the big-tile kernels when the caller provides it
    const bool wil = g.w_il && g.split == 3 && !conv && p.a.batch == 1 && g.ldw_il >= 2 * (int64_t)g.K && (g.ldw_il & 7) == 0;
    const bool ail = g.a_il != 0;
    if (ail && (!wil || (variant_env != 0 && variant_env != 2 && variant_env != 7))) return CVLM_E_UNSUPPORTED;
    // variant 0: auto (big tile for big problems); 1: 128x128 2-stage; 2: 256x128 3-stage; 3: 128x128 3-stage(4 waves)
    int variant = variant_env;
    if (variant == 0) {
        // 256x256 tiles (1 workgroup/CU, half the L2->LDS bytes per FLOP) when they fill the 256 CUs well,
        // otherwise 128x128 tiles (2 workgroups/CU -> 512 slots) for small / skinny problems.
        const long t5 = (long)((g.M + 255) / 256) * ((g.N + 255) / 256) * p.a.batch;
        const long t1 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128) * p.a.batch;
        const double e5 = (double)t5 / (double)(((t5 + 255) / 256) * 256);
        const double e1 = (double)t1 / (double)(((t1 + 511) / 512) * 512);
        // padded work / relative throughput (measured, profiles/r01_gemm_probes.md)
        const long t2 = (long)((g.M + 255) / 256) * ((g.N + 127) / 128) * p.a.batch;
        const double c1 = (double)(((t1 + 511) / 512) * 512) * 1.0;
        (void)e5; (void)e1;
        if (g.split == 3) {
            // microsecond model fitted to tools/ab_gemm.py (B = 8 cascade shapes): rounds x time per tile,
            //   128^2 (2 workgroups/CU): 0.0544 us per K;  256x128: 0.0469 us per K;
            //   256^2 staggered: 0.0685 us per K + 14 us per tile, last partial round cut into S K-parts (tail_us).
            const double K = (double)g.K;
            const double m1 = (double)((t1 + 511) / 512) * 0.0544 * K;
            const double m2 = (double)((t2 + 255) / 256) * 0.0469 * K;
            const double tile = 0.0685 * K + 14.0;
            const int rem = (int)(t5 % 256);
            double tail = rem > 0 ? tile : 0.0;
            if (tail_env && have_ws && p.a.batch == 1) {
                const int S = tail_parts(rem, g.K);
                if (S >= 2) tail = tail_us(S, g.K, rem);
            }
            const double m7 = (double)(t5 / 256) * tile + tail;
            // a near tie goes to the 256^2 tile: it moves a third fewer L2->LDS bytes per flop, and at the power cap the
            // joules count (proj 32768 x 1280 x 1280 is modelled 305 vs 300 us; with 256^2 tiles the cascade gains 0.85 %)
            variant = (m7 <= 1.04 * m1 && m7 <= 1.04 * m2) ? 5 : (m2 <= m1 ? 2 : 1);
            if (g.a_il) variant = m7 <= 1.04 * m2 ? 5 : 2;                    // kernels that stage the activation image: 256^2 family, 256 x 128
        } else if (t5 >= 200 && t5 <= 256) variant = 5;                          // one full wave of 256^2 tiles
        else if (t5 >= 1536) {
            const double c2 = (double)(((t2 + 255) / 256) * 256) * 2.0 / 1.08;
            const double c5 = (double)(((t5 + 255) / 256) * 256) * 4.0 / 1.20;
            variant = (c5 <= c2 && c5 <= c1) ? 5 : (c2 <= c1 ? 2 : 1);
        } else {
            const double c2 = (double)(((t2 + 255) / 256) * 256) * 2.0 / (g.M >= 16384 ? 1.08 : 1.05);
            variant = (c2 < c1) ? 2 : 1;
        }
    }
    // ---- small grids: one image (M = 4096 in the ViT-H blocks, 581 in the CLIP towers -- the reference's own call pattern,
    // demo.py / DataLoader batch_size = 1) and everything below.  The constants of the model above were fitted on grids that fill
    // the chip several times; here a workgroup often has a CU (or the L2) to itself and runs up to twice as fast, hand-offs are
    // cheap (every K-part is resident at once) and a fourth form exists: split-K over ALL tiles of a 128^2 grid (SK kernel).
    // Per-workgroup cost = t0 + K * c(fill), c rising linearly with the fill of the workgroup slots; fitted on the eight shapes of
    // tools/ab_gemm.py SHAPES=b1 (profiles/r03_ab_gemm_b1.log: the model picks the measured winner for each).
    p.sk_parts = 1;
    // 128^2 launches of a small grid (whole or in K-parts) run the deep ring: see the R-slot loop of the kernel
    static int ring_env = env_int("CVLM_GEMM_RING", 4);
    if (live_env) ring_env = env_int("CVLM_GEMM_RING", 4);
    // ... with eight waves of 32 x 64 instead of four of 64 x 64 (CVLM_GEMM_W8): a wave issues one KiB of DMA per ~100 cycles, and
    // with the chip mostly idle it is the issue of a K-tile's 32 DMA instructions by four waves (800 cycles against 768 of MFMAs),
    // not the matrix pipe, that a workgroup waits for
    // CVLM_GEMM_W8: 0 four waves of 64 x 64; 1 (default) eight waves, 64 x 128 tiles when those fit one round of workgroups, else
    // 128^2; 2 / 3 force the 64-row / 128-row form (A/B tools)
    static int w8_env = env_int("CVLM_GEMM_W8", 1);
    if (live_env) w8_env = env_int("CVLM_GEMM_W8", 1);
    auto pick_ring = [&](long wgs) { return (wgs <= 256 && ring_env >= 3 && ring_env <= 5) ? ring_env : 0; };
    int small_tail_S = 0;                                             // > 0: K-parts of the 256^2 tail chain chosen here
    int small_ring = 0;                                               // > 0: slots of the deep LDS ring for a 128^2 launch of a small grid
    static int sk_env = env_int("CVLM_GEMM_SK", 1);
    if (live_env) sk_env = env_int("CVLM_GEMM_SK", 1);
    if (g.split == 3 && !conv && p.a.batch == 1 && g.M <= 4096 && !g.a_mx && (variant_env == 0 || (variant_env == 1 && sk_env > 1))) {
        const double K = (double)g.K;
        const long t1 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
        const long t2 = (long)((g.M + 255) / 256) * ((g.N + 127) / 128);
        const long t5 = (long)((g.M + 255) / 256) * ((g.N + 255) / 256);
        auto fillc = [](double lo, double hi, double wgs, double slots) { const double f = wgs / slots; return lo + (hi - lo) * (f < 1.0 ? f : 1.0); };
        auto over = [](double wgs, double slots) { const double r = wgs / slots; return r > 1.0 ? r : 1.0; };
        // 128^2 family: under one round of workgroups the deep-ring / eight-wave kernels run (0.78 of the fitted two-slot cost per K;
        // 0.62 with 64 x 128 tiles when those still fit one round), tools/ab_gemm.py SHAPES=b1 COLD=1
        const long t64 = (long)((g.M + 63) / 64) * ((g.N + 127) / 128);
        const bool ring_on = ring_env == 4 && w8_env != 0;
        const double k1 = !ring_on || t1 > 256 ? 1.0 : (t64 <= 256 && w8_env != 3 ? 0.62 : 0.78);
        const double m1 = 4.0 + k1 * K * fillc(0.021, 0.054, (double)t1, 512.0) * over((double)t1, 512.0);
        const double m2 = 4.0 + K * fillc(0.028, 0.047, (double)t2, 256.0) * over((double)t2, 256.0);
        // 256^2 tiles: full rounds at the fitted tile time, the last round whole or cut into S chained K-parts
        const long full5 = t5 / 256, rem5 = t5 % 256;
        double m7 = (double)full5 * (14.0 + 0.0685 * K);
        int tailS = 0;
        if (rem5 > 0) {
            double last = (14.0 + 0.0685 * K) * (0.75 + 0.25 * (double)rem5 / 256.0);
            if (tail_env && have_ws && rem5 <= 128) {
                const int smax = 256 / rem5 < 4 ? (int)(256 / rem5) : 4;
                for (int S = 2; S <= smax; ++S) {
                    if (g.K / 32 < 4 * S) break;
                    const double f = (double)(rem5 * S) / 256.0;
                    const double t = (14.0 + 0.0685 * K / S) * (0.75 + 0.25 * (f < 1.0 ? f : 1.0)) + 6.0 + 4.0 * S;
                    if (t < 0.95 * last) { last = t; tailS = S; }
                }
            }
            m7 += last;
        }
        // split-K over every 128^2 tile: S x (slab write + read) of the whole output is what it costs
        int skS = 1;
        double msk = 1e30;
        if (have_ws && sk_env != 0 && t1 <= SK_MAX_TILES) {
            for (int S = 2; S <= 8; ++S) {
                if (g.K / 32 < 4 * S || (size_t)t1 * S * 128 * 128 * sizeof(float) > TAIL_WS_BYTES) break;
                const double wg = (double)(t1 * S);
                const double t = 4.0 + (ring_on && wg <= 256.0 ? 0.78 : 1.0) * (K / S) * fillc(0.021, 0.054, wg, 512.0) * over(wg, 512.0) + 6.0 + 2.0 * S + 0.052 * wg;
                if (t < msk) { msk = t; skS = S; }
            }
        }
        if (sk_env > 1) {                                                // forced (A/B tools, tests): the same limits
            const bool okS = have_ws && sk_env <= 8 && t1 <= SK_MAX_TILES && g.K / 32 >= 4 * sk_env &&
                             (size_t)t1 * sk_env * 128 * 128 * sizeof(float) <= TAIL_WS_BYTES;
            skS = okS ? sk_env : 1;
            msk = okS ? 0.0 : 1e30;
        }
        const bool fast_ok = true;
        (void)fast_ok;
        // activation image: of the 128^2 family only the eight-wave ring kernels stage it (one round of workgroups)
        const bool ail1_ok = !ail || (ring_on && w8_env != 0 && t1 <= 256);
        if (ail && skS > 1 && !(ring_on && (long)t1 * skS <= 256)) { skS = 1; msk = 1e30; }
        if (variant_env == 0) {
            double best = ail1_ok ? m1 : 1e30;
            variant = 1;
            if (m2 < best) { best = m2; variant = 2; }
            if (m7 < best) { best = m7; variant = 5; small_tail_S = tailS > 0 ? tailS : -1; }
            if (skS > 1 && msk < 0.9 * best) { best = msk; variant = 1; p.sk_parts = skS; }
        } else if (skS > 1) {
            p.sk_parts = skS;
        }
        if (p.sk_parts > 1) {
            small_ring = pick_ring(t1 * p.sk_parts);
            if (small_ring > 4) small_ring = 4;                          // five slots are the whole LDS; this kernel has a static word beside them
            p.flags = (unsigned*)g.workspace;
            p.ws = (float*)((unsigned char*)g.workspace + TAIL_FLAG_BYTES);
            p.nbx = (g.N + 127) / 128; p.nby = (g.M + 127) / 128;
#define CVLM_LAUNCH_SK(WM_, MT_, NS_, SLOTS_)                                                                       \
    do {                                                                                                            \
        constexpr int smem_sk = SLOTS_ * 2 * (128 + 128) * 32 * 2;                                                  \
        auto ksk = gemm_nt_kernel<3, WM_, 2, NS_, 32, 0, MT_, false, -1, false, true>;                              \
        static bool attr_sk[16] = {};                                                                               \
        if (cvlm_first_on_device(attr_sk))                                                                          \
            (void)hipFuncSetAttribute((const void*)ksk, hipFuncAttributeMaxDynamicSharedMemorySize, smem_sk);      \
        hipLaunchKernelGGL(ksk, dim3(p.nbx* p.nby * p.sk_parts, 1), dim3(WM_ * 2 * 64), smem_sk, s, p);             \
    } while (0)
            if (ail && !(small_ring == 4 && w8_env && wil)) return CVLM_E_UNSUPPORTED;
            if (small_ring == 4 && w8_env && wil) {                       /* the same, operands from the 128-byte-row images */
                constexpr int smem_sk = 4 * 2 * (128 + 128) * 32 * 2;
                auto ksk = gemm_nt_kernel<3, 4, 2, 14, 32, 0, 2, false, -1, false, true, true>;
                auto kska = gemm_nt_kernel<3, 4, 2, 14, 32, 0, 2, false, -1, false, true, true, true>;
                static bool attr_skw[16] = {};
                if (cvlm_first_on_device(attr_skw)) {
                    (void)hipFuncSetAttribute((const void*)ksk, hipFuncAttributeMaxDynamicSharedMemorySize, smem_sk);
                    (void)hipFuncSetAttribute((const void*)kska, hipFuncAttributeMaxDynamicSharedMemorySize, smem_sk);
                }
                hipLaunchKernelGGL(ail ? kska : ksk, dim3(p.nbx * p.nby * p.sk_parts, 1), dim3(512), smem_sk, s, p);
            }
            else if (small_ring == 4 && w8_env) CVLM_LAUNCH_SK(4, 2, 14, 4);
            else if (small_ring == 3) CVLM_LAUNCH_SK(2, 4, 13, 3);
            else if (small_ring == 4) CVLM_LAUNCH_SK(2, 4, 14, 4);
            else CVLM_LAUNCH_SK(2, 4, 2, 2);
#undef CVLM_LAUNCH_SK
            CVLM_CHECK_LAUNCH();
            return 0;
        }
    }
#define CVLM_LAUNCH(SPLIT, WM, WN, NS) CVLM_LAUNCH_D(SPLIT, WM, WN, NS, 32, 0, 4)
#define CVLM_LAUNCH_D(SPLIT, WM, WN, NS, BKT, DBG, MT)                                                        \
    do {                                                                                                      \
        constexpr int NPA_ = (SPLIT == 3) ? 2 : 1;                                                            \
        constexpr int smem_ = (NS >= 13 ? NS - 10 : NS == 6 ? 1 : (NS >= 4 ? 2 : NS)) * NPA_ * (WM * MT * 16 + WN * 64) * BKT * 2;                                      \
        p.nbx = (g.N + WN * 64 - 1) / (WN * 64);                                                              \
        p.nby = (g.M + WM * MT * 16 - 1) / (WM * MT * 16);                                                            \
        auto kern_ = gemm_nt_kernel<SPLIT, WM, WN, NS, BKT, DBG, MT>;                                                     \
        static bool attr_[16] = {};                                                                           \
        if (smem_ > 48 * 1024 && cvlm_first_on_device(attr_))                                                             \
            (void)hipFuncSetAttribute((const void*)kern_, hipFuncAttributeMaxDynamicSharedMemorySize, smem_); \
        hipLaunchKernelGGL(kern_, dim3(p.nbx* p.nby + extra_blocks, p.a.batch), dim3(WM* WN * 64), smem_, s, p); \
    } while (0)
#define CVLM_LAUNCH_W(WM, WN, NS, MT)    /* split-3 ring kernels: weight from the interleaved image when there is one */       \
    do {                                                                                                           \
        constexpr int smem_ = (NS - 10) * 2 * (WM * MT * 16 + WN * 64) * 32 * 2;                                   \
        p.nbx = (g.N + WN * 64 - 1) / (WN * 64);                                                                   \
        p.nby = (g.M + WM * MT * 16 - 1) / (WM * MT * 16);                                                         \
        auto kern_ = gemm_nt_kernel<3, WM, WN, NS, 32, 0, MT>;                                                     \
        auto kernw_ = gemm_nt_kernel<3, WM, WN, NS, 32, 0, MT, false, -1, false, false, true>;                     \
        auto kernwa_ = gemm_nt_kernel<3, WM, WN, NS, 32, 0, MT, false, -1, false, false, true, true>;              \
        static bool attr_[16] = {};                                                                                \
        if (cvlm_first_on_device(attr_)) {                                                                         \
            (void)hipFuncSetAttribute((const void*)kern_, hipFuncAttributeMaxDynamicSharedMemorySize, smem_);     \
            (void)hipFuncSetAttribute((const void*)kernw_, hipFuncAttributeMaxDynamicSharedMemorySize, smem_);    \
            (void)hipFuncSetAttribute((const void*)kernwa_, hipFuncAttributeMaxDynamicSharedMemorySize, smem_);   \
        }                                                                                                          \
        hipLaunchKernelGGL(ail ? kernwa_ : wil ? kernw_ : kern_, dim3(p.nbx* p.nby + extra_blocks, p.a.batch), dim3(WM* WN * 64), smem_, s, p); \
    } while (0)
#define CVLM_LAUNCH_E(EPI_)                                                                                        \
    do {                                                                                                           \
        constexpr int smem_ = 2 * 2 * (256 + 256) * 32 * 2;                                                        \
        p.nbx = (g.N + 255) / 256; p.nby = (g.M + 255) / 256;                                                      \
        auto kern_ = gemm_nt_kernel<3, 2, 4, 5, 32, 0, 8, false, EPI_>;                                            \
        auto kernw_ = gemm_nt_kernel<3, 2, 4, 5, 32, 0, 8, false, EPI_, false, false, true>;                       \
        auto kernwa_ = gemm_nt_kernel<3, 2, 4, 5, 32, 0, 8, false, EPI_, false, false, true, true>;                \
        static bool attr_[16] = {};                                                                                \
        if (cvlm_first_on_device(attr_)) {                                                                         \
            (void)hipFuncSetAttribute((const void*)kern_, hipFuncAttributeMaxDynamicSharedMemorySize, smem_);     \
            (void)hipFuncSetAttribute((const void*)kernw_, hipFuncAttributeMaxDynamicSharedMemorySize, smem_);    \
            (void)hipFuncSetAttribute((const void*)kernwa_, hipFuncAttributeMaxDynamicSharedMemorySize, smem_);   \
        }                                                                                                          \
        hipLaunchKernelGGL(ail ? kernwa_ : wil ? kernw_ : kern_, dim3(p.nbx* p.nby + extra_blocks, 1), dim3(512), smem_, s, p);    \
    } while (0)
    int extra_blocks = 0;
    // slots of the deep ring for a plain (whole-tile) 128^2-family launch of a small grid, 0: the two-slot loop
    const int small_ring_pick = (g.split == 3 && !conv && p.a.batch == 1 && g.M <= 4096) ? pick_ring((long)((g.M + 127) / 128) * ((g.N + 127) / 128)) : 0;
    if (conv) {
        // 256 x 64 tiles (4 waves, 2 workgroups per CU): the edge head's N is 32 / 64, the neck's 256
        constexpr int smem_c = 2 * 2 * (256 + 64) * 32 * 2;
        p.nbx = (g.N + 63) / 64; p.nby = (g.M + 255) / 256;
        static bool attr_c[16] = {};
        if (cvlm_first_on_device(attr_c)) {
            (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<3, 4, 1, 2, 32, 0, 4, false, -1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_c);
            (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<1, 4, 1, 2, 32, 0, 4, false, -1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_c);
        }
        if (g.split == 3) hipLaunchKernelGGL((gemm_nt_kernel<3, 4, 1, 2, 32, 0, 4, false, -1, true>), dim3(p.nbx * p.nby, 1), dim3(256), smem_c, s, p);
        else hipLaunchKernelGGL((gemm_nt_kernel<1, 4, 1, 2, 32, 0, 4, false, -1, true>), dim3(p.nbx * p.nby, 1), dim3(256), smem_c, s, p);
        CVLM_CHECK_LAUNCH();
        return 0;
    }
    if (g.split == 3) {
        if (variant == 5 && variant_env == 0) variant = 7;      // auto: staggered wave groups (3-5 % over the plain 256^2 loop)
        // tile rows per L2 super-tile: 8 for the small tiles; the 256^2 kernel is 2-3 % faster with 4 (2 at long K),
        // i.e. ~20 (10) of an XCD's 32 co-resident tiles sharing their activation panels (tools/ab_gemm.py sweep)
        if (variant == 7) p.group_m = g.K >= 4096 ? 2 : 4;
        if (variant == 7 && tail_env && have_ws && p.a.batch == 1) {
            const long T = (long)((g.M + 255) / 256) * ((g.N + 255) / 256);
            const int rem = (int)(T % 256);
            const int S = small_tail_S > 0 ? small_tail_S : (small_tail_S < 0 ? 1 : tail_parts(rem, g.K));   // small grids: chosen above
            if (S >= 2) {
                p.tail_rem = rem; p.tail_split = S;
                p.flags = (unsigned*)g.workspace;
                p.ws = (float*)((unsigned char*)g.workspace + TAIL_FLAG_BYTES);
                extra_blocks = rem * (S - 1);
            }
        }
        // Persistent form of the 256^2 kernel: one workgroup per CU walks the tile list.  Not for launches with tail parts:
        // their hand-off chain relies on in-order dispatch (every producer is resident or done before its consumer
        // starts), which a persistent grid sharing the chip with another stream cannot promise.
        const cvlm_gemm_args& ga = p.a;                                  // the sanitised copy (fold / h2-residual launches)
        const bool lds_staged = ga.ps_c2 == 0 && (ga.N & 7) == 0 && (ga.hm_S == 0 || ((ga.hm_hd & 7) == 0 && ga.hm_S >= 128)) &&
                                (ga.ldo & 3) == 0 && (ga.stride_o & 3) == 0 && (ga.ldr & 3) == 0 && (ga.stride_r & 3) == 0 &&
                                (ga.ldoh & 7) == 0 && (ga.stride_oh & 7) == 0;
        static int t192_env = env_int("CVLM_GEMM_T192", 1);
        if (live_env) t192_env = env_int("CVLM_GEMM_T192", 1);
        if (g.a_mx) {
            // both operands mx: the staggered 256-column kernel in its unit form (gemm_kernel.h, MX), one instantiation per epilogue form;
            // 192-row tiles under one round of 256-row ones (the h2-residual form, as above); K-parts of a partial last round are whole
            // groups of 8 units
            // 32-bit quantities of the kernel: the offset of a row inside its 8-row DMA block (the block's base is a 64-bit scalar) and the
            // byte offset of a row's scale words -- not the size of the images (batches above 4 GiB of operand are fine)
            if (!lds_staged || p.a.batch != 1 || (g.M & 7) || (g.N & 7) || (int64_t)8 * g.lda * 2 >= ((int64_t)1 << 32) ||
                (int64_t)8 * g.ldw_mx * 2 >= ((int64_t)1 << 32) || (int64_t)g.M * 4 * g.lda_s >= ((int64_t)1 << 32) ||
                (int64_t)g.N * 4 * g.ldw_s >= ((int64_t)1 << 32))
                return CVLM_E_UNSUPPORTED;
            const long t5 = (long)((g.M + 255) / 256) * ((g.N + 255) / 256), t6 = (long)((g.M + 191) / 192) * ((g.N + 255) / 256);
            const bool use192 = t192_env && h2res && t5 <= 256 && t6 <= 256 && t6 > t5 && g.M > 4096;
            p.tail_rem = 0; p.tail_split = 1; extra_blocks = 0;
            if (!use192 && tail_env && have_ws) {
                const int rem = (int)(t5 % 256);
                int S = tail_parts(rem, g.K);
                while (S > 1 && (g.K / 32 + 7) / 8 < 2 * S) --S;
                if (S >= 2) {
                    p.tail_rem = rem; p.tail_split = S;
                    p.flags = (unsigned*)g.workspace;
                    p.ws = (float*)((unsigned char*)g.workspace + TAIL_FLAG_BYTES);
                    extra_blocks = rem * (S - 1);
                }
            }
            p.group_m = g.K >= 4096 ? 2 : 4;
            if (live_env && env_int("CVLM_GEMM_GROUP_M", 0) > 0) p.group_m = env_int("CVLM_GEMM_GROUP_M", 0);   /* tools only */
            const int probe = variant_env >= 100 ? variant_env - 100 : 0;          /* probe builds: tools/probe_gemm_mx.py */
            return launch_mx(p, use192 ? 6 : 8, fold ? 1 : (h2res ? 2 : 0), extra_blocks, probe, s);
        }
        // 192 x 256 tiles (MT = 6, same staggered loop) for grids UNDER one round of 256^2 tiles: the CLIP out_proj / c_proj of the
        // fused 16-image forward are 37 x 4 = 148 tiles on 256 CUs; 49 x 4 = 196 tiles of 192 rows put 48 more CUs to work and
        // every workgroup finishes a quarter earlier.  Only the h2-residual form is instantiated (what those launches use).
        if (variant == 7 && variant_env == 0 && t192_env && h2res && lds_staged && p.a.batch == 1 && p.tail_rem == 0 && g.M > 4096) {
            const long t5 = (long)((g.M + 255) / 256) * ((g.N + 255) / 256), t6 = (long)((g.M + 191) / 192) * ((g.N + 255) / 256);
            if (t5 <= 256 && t6 <= 256 && t6 > t5) {
                constexpr int smem6 = 2 * 2 * (192 + 256) * 32 * 2;
                p.nbx = (g.N + 255) / 256; p.nby = (g.M + 191) / 192;
                auto k6 = gemm_nt_kernel<3, 2, 4, 5, 32, 0, 6, false, 2>;
                auto k6w = gemm_nt_kernel<3, 2, 4, 5, 32, 0, 6, false, 2, false, false, true>;
                auto k6wa = gemm_nt_kernel<3, 2, 4, 5, 32, 0, 6, false, 2, false, false, true, true>;
                static bool attr6[16] = {};
                if (cvlm_first_on_device(attr6)) {
                    (void)hipFuncSetAttribute((const void*)k6, hipFuncAttributeMaxDynamicSharedMemorySize, smem6);
                    (void)hipFuncSetAttribute((const void*)k6w, hipFuncAttributeMaxDynamicSharedMemorySize, smem6);
                    (void)hipFuncSetAttribute((const void*)k6wa, hipFuncAttributeMaxDynamicSharedMemorySize, smem6);
                }
                hipLaunchKernelGGL(ail ? k6wa : wil ? k6w : k6, dim3(p.nbx * p.nby, 1), dim3(512), smem6, s, p);
                CVLM_CHECK_LAUNCH();
                return 0;
            }
        }
        if (variant == 7 && persist_env && (variant_env == 0 || variant_env == 7) && lds_staged && p.a.batch == 1) {
            static int cus_[16] = {};
            int dev = 0;
            (void)hipGetDevice(&dev);
            int& cus = cus_[dev & 15];
            if (cus == 0 && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
            p.nbx = (g.N + 255) / 256; p.nby = (g.M + 255) / 256;
            const int T = p.nbx * p.nby;
            if (T > cus && p.tail_rem == 0) {
                constexpr int STAGE_ = 2 * (256 + 256) * 32 * 2;
                constexpr int smem_p = 2 * STAGE_ + 8 * 16 * 64 * 4;            // ring + one 16 x 64 f32 slab per wave = 160 KB
                p.total_blocks = T;
#define CVLM_LAUNCH_P(EPI_)                                                                                        \
    do {                                                                                                           \
        auto kp = gemm_nt_kernel<3, 2, 4, 5, 32, 0, 8, true, EPI_>;                                                \
        auto kpw = gemm_nt_kernel<3, 2, 4, 5, 32, 0, 8, true, EPI_, false, false, true>;                           \
        auto kpwa = gemm_nt_kernel<3, 2, 4, 5, 32, 0, 8, true, EPI_, false, false, true, true>;                    \
        static bool attr_p[16] = {};                                                                               \
        if (cvlm_first_on_device(attr_p)) {                                                                        \
            (void)hipFuncSetAttribute((const void*)kp, hipFuncAttributeMaxDynamicSharedMemorySize, smem_p);       \
            (void)hipFuncSetAttribute((const void*)kpw, hipFuncAttributeMaxDynamicSharedMemorySize, smem_p);      \
            (void)hipFuncSetAttribute((const void*)kpwa, hipFuncAttributeMaxDynamicSharedMemorySize, smem_p);     \
        }                                                                                                          \
        hipLaunchKernelGGL(ail ? kpwa : wil ? kpw : kp, dim3(cus, 1), dim3(512), smem_p, s, p);                    \
    } while (0)
                if (fold) CVLM_LAUNCH_P(1); else if (h2res) CVLM_LAUNCH_P(2); else CVLM_LAUNCH_P(0);
#undef CVLM_LAUNCH_P
                CVLM_CHECK_LAUNCH();
                return 0;
            }
        }
        if (variant == 2 && wil) {                                           /* 256 x 128 tiles, weight from the interleaved image */
            constexpr int smem2 = 3 * 2 * (256 + 128) * 32 * 2;
            p.nbx = (g.N + 127) / 128; p.nby = (g.M + 255) / 256;
            auto k2w = gemm_nt_kernel<3, 4, 2, 3, 32, 0, 4, false, -1, false, false, true>;
            auto k2wa = gemm_nt_kernel<3, 4, 2, 3, 32, 0, 4, false, -1, false, false, true, true>;
            static bool attr2[16] = {};
            if (cvlm_first_on_device(attr2)) {
                (void)hipFuncSetAttribute((const void*)k2w, hipFuncAttributeMaxDynamicSharedMemorySize, smem2);
                (void)hipFuncSetAttribute((const void*)k2wa, hipFuncAttributeMaxDynamicSharedMemorySize, smem2);
            }
            hipLaunchKernelGGL(ail ? k2wa : k2w, dim3(p.nbx * p.nby, 1), dim3(512), smem2, s, p);
        }
        else if (variant == 2) CVLM_LAUNCH(3, 4, 2, 3);
        else if (variant == 7 && lds_staged && p.a.batch == 1) {             /* 256x256, 8 waves, wave groups staggered; one epilogue form */
            if (fold) CVLM_LAUNCH_E(1); else if (h2res) CVLM_LAUNCH_E(2); else CVLM_LAUNCH_E(0);
        }
        else if (variant == 1 && small_ring_pick == 4 && (w8_env == 2 || (w8_env == 1 && (long)((g.M + 63) / 64) * ((g.N + 127) / 128) <= 256)))
            CVLM_LAUNCH_W(4, 2, 14, 1);                                      /* 64 x 128 tiles, eight waves of 16 x 64 */
        else if (variant == 1 && small_ring_pick == 4 && w8_env) CVLM_LAUNCH_W(4, 2, 14, 2);     /* 128^2 tiles, eight waves of 32 x 64 */
        else if (ail) return CVLM_E_UNSUPPORTED;                             /* no other kernel stages the activation image */
        else if (variant == 7) CVLM_LAUNCH_D(3, 2, 4, 5, 32, 0, 8);          /* same, every epilogue form (pixel shuffle, odd N, batched) */
#ifdef CVLM_PROBES
        else if (variant == 77) CVLM_LAUNCH_D(3, 2, 4, 5, 32, 7, 8);        /* probe: s_setprio around MFMA groups */
        else if (variant == 87) CVLM_LAUNCH_D(3, 2, 4, 5, 32, 8, 8);        /* probe: static priority for waves 4..7 */
#endif
#ifdef CVLM_PROBES   /* make EXTRA=-DCVLM_PROBES: the variants behind profiles/r01_gemm_probes.md and tools/{ab,trace}_gemm.py */
        else if (variant == 4 && (g.K % 64) == 0) CVLM_LAUNCH_D(3, 2, 2, 2, 64, 0, 4);
        else if (variant == 24 && (g.K % 64) == 0) CVLM_LAUNCH_D(3, 2, 2, 2, 64, 2, 4);   /* probe: DMA only, 128-byte rows (full L2 lines) */
        else if (variant == 21) CVLM_LAUNCH_D(3, 2, 2, 2, 32, 2, 4);                      /* probe: DMA only, 64-byte rows, same tile */
        else if (variant == 5) CVLM_LAUNCH_D(3, 2, 4, 2, 32, 0, 8);          /* 256x256, 8 waves of 128x64 */
        else if (variant == 6) CVLM_LAUNCH_D(3, 2, 4, 4, 32, 0, 8);          /* same tile, mid-tile slot recycling */
        else if (variant == 8) CVLM_LAUNCH_D(3, 2, 2, 6, 32, 0, 8);          /* 256x128, 4 waves, one recycled slot, 2 WG/CU */
        else if (variant == 18) CVLM_LAUNCH_D(3, 2, 2, 6, 32, 1, 8);
        else if (variant == 28) CVLM_LAUNCH_D(3, 2, 2, 6, 32, 2, 8);
        else if (variant == 17) CVLM_LAUNCH_D(3, 2, 4, 5, 32, 1, 8);
        else if (variant == 27) CVLM_LAUNCH_D(3, 2, 4, 5, 32, 2, 8);
        else if (variant == 37) CVLM_LAUNCH_D(3, 2, 4, 5, 32, 3, 8);        /* probe: no epilogue stores */
        else if (variant == 57) CVLM_LAUNCH_D(3, 2, 4, 5, 32, 5, 8);        /* probe: epilogue staging only */
        else if (variant == 67) CVLM_LAUNCH_D(3, 2, 4, 5, 32, 6, 8);        /* probe: main loop only */
        else if (variant == 47 && g_trace) CVLM_LAUNCH_D(3, 2, 4, 5, 32, 4, 8);  /* probe: per-workgroup timeline */
        else if (variant == 16) CVLM_LAUNCH_D(3, 2, 4, 4, 32, 1, 8);
        else if (variant == 26) CVLM_LAUNCH_D(3, 2, 4, 4, 32, 2, 8);
        else if (variant == 15) CVLM_LAUNCH_D(3, 2, 4, 2, 32, 1, 8);
        else if (variant == 25) CVLM_LAUNCH_D(3, 2, 4, 2, 32, 2, 8);
        else if (variant == 3) CVLM_LAUNCH(3, 2, 2, 3);
#endif
        else if (variant == 1 && small_ring_pick == 3) CVLM_LAUNCH(3, 2, 2, 13);
        else if (variant == 1 && small_ring_pick == 4) CVLM_LAUNCH(3, 2, 2, 14);
        else if (variant == 1 && small_ring_pick == 5) CVLM_LAUNCH(3, 2, 2, 15);
        else CVLM_LAUNCH(3, 2, 2, 2);
    } else {
        if (variant == 2) CVLM_LAUNCH(1, 4, 2, 3);
        else if (variant == 5) CVLM_LAUNCH_D(1, 2, 4, 3, 32, 0, 8);
#ifdef CVLM_PROBES
        else if (variant == 4 && (g.K % 64) == 0) CVLM_LAUNCH_D(1, 2, 2, 3, 64, 0, 4);
        else if (variant == 3) CVLM_LAUNCH(1, 2, 2, 3);
#endif
        else CVLM_LAUNCH(1, 2, 2, 2);
    }
#undef CVLM_LAUNCH
#undef CVLM_LAUNCH_D
#undef CVLM_LAUNCH_E
#undef CVLM_LAUNCH_W
    CVLM_CHECK_LAUNCH();
    return 0;
}
