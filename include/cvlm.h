/* cvlm.h -- C ABI of libcvlm_hip.so: the MI355X (gfx950) kernels behind the camouflaged-vlm
 * cascaded forward pass.
 *
 * The reference (intcomp/camouflaged-vlm) is pure PyTorch and has NO FFI / plugin boundary
 * (SURVEY.md §8b); its hot path calls torch eager ops.  This header is therefore the boundary the
 * build defines: every entry point names the reference site (file:line, relative to the reference
 * root) whose arithmetic it replaces.  Conventions:
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless stated otherwise;
 *   - launchers never allocate, never synchronise, never retain pointers; `stream` is a hipStream_t
 *     passed as void* (NULL = default stream) and must belong to the CURRENT device (one device per process is the
 *     intended use; a process driving several devices calls hipSetDevice before each launch);
 *   - workspace is caller-owned: the two entries that need scratch memory (cvlm_gemm, cvlm_attention) take a
 *     `workspace` pointer + size in their argument structs and publish the size they want through
 *     cvlm_gemm_workspace_bytes() / cvlm_attention_workspace_bytes().  A workspace must not be shared by launches
 *     that can run concurrently (different streams);
 *   - return value: 0 on success, otherwise a hipError_t value or a negative CVLM_E_* code;
 *   - tensors are row-major.  "f32" = float.  "h2" = split-half pair: two fp16 planes (hi, lo) of
 *     identical shape, value = hi + lo; `lo` may be NULL where noted (fast mode, split = 1).
 */
#ifndef CVLM_H
#define CVLM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CVLM_ABI_VERSION 12
#define CVLM_E_BADARG (-1)
#define CVLM_E_UNSUPPORTED (-2)
#define CVLM_E_WORKSPACE (-3)     /* workspace missing or smaller than cvlm_*_workspace_bytes() */

enum { CVLM_ACT_NONE = 0, CVLM_ACT_GELU = 1, CVLM_ACT_QUICKGELU = 2, CVLM_ACT_RELU = 3, CVLM_ACT_ABS_POST = 4 };

int cvlm_abi_version(void);
/* Device the library was built for ("gfx950") */
const char* cvlm_target_arch(void);

/* ---------------------------------------------------------------------------------------------
 * GEMM  out[z][m][n] = post( act( alpha * sum_k A[z][m][k] * W[z][n][k] + bias[n] ) + residual[z][m][n] )
 * Replaces every nn.Linear / 1x1 / patch conv / ConvTranspose2d(k2,s2) on the path, e.g.
 *   image_encoder.py:491 (qkv), :502 (proj), common.py:25 (MLP lin1/lin2), :651-659 (patch embed),
 *   alpha_clip_rw/model.py:225,254 (in_proj/out_proj), :296-300 (c_fc/c_proj),
 *   transformer_maskdecoder_edge.py:252-254,270 (decoder projections), mask_decoder_edge.py:53-59.
 * A and W are h2 (K contiguous; lda/ldw in elements, multiples of 8; K multiple of 32).
 * split = 3: hi*hi + lo*hi + hi*lo (needs both lo planes); split = 1: hi planes only.
 * Outputs (any subset): out_f32 (ldo) and/or out h2 (ldoh).  ps_c2 > 0 selects the pixel-shuffle
 * store used for ConvTranspose2d(k=2,s=2): row m = (b,y,x) on a ps_h x ps_w grid, column
 * n = dy*ps_c2 + r  ->  out[((b*2*ps_h + 2y+dy) * 2*ps_w + 2x) * (ps_c2/2) + r].
 * batch > 1 runs `batch` independent problems with the given element strides (0 = shared).
 * out_scale (0 = 1): the h2 output is stored as value * out_scale (a power of two keeps it exact); the consumer folds
 * 1 / out_scale into its alpha.  It moves an unbounded activation (the GELU output of common.py:25) into fp16 range.
 * workspace: optional scratch of cvlm_gemm_workspace_bytes() bytes.  With it, a grid of 256 x 256 tiles whose last
 * round fills at most half the chip has the tiles of that round cut along K and reduced through fp32 slabs in the
 * workspace (fixed summation order); without it every tile is computed whole (same result up to fp32 summation
 * order, slower for 2.5-round shapes).  The first 4 KiB of the workspace are hand-off words: zero them ONCE after
 * allocation (hipMemset); kernels leave them zero.  Word 512 counts abandoned hand-offs (a partner workgroup that
 * never arrived; the affected tile is written as NaN), word 513 rows cvlm_ln_stats_merge refused (below): 0 in a
 * healthy run.
 */
typedef struct cvlm_gemm_args {
    const void* a_hi; const void* a_lo; int64_t lda; int64_t stride_a;
    const void* w_hi; const void* w_lo; int64_t ldw; int64_t stride_w;
    const float* bias;
    const float* residual; int64_t ldr; int64_t stride_r;
    float* out_f32; int64_t ldo; int64_t stride_o;
    void* out_hi; void* out_lo; int64_t ldoh; int64_t stride_oh;
    int32_t M, N, K, batch;
    float alpha;
    int32_t act;
    int32_t split;
    int32_t ps_h, ps_w, ps_c2;
    int32_t hm_S, hm_H, hm_hd;   /* hm_S > 0: h2 output stored head-major [3][M/hm_S][hm_H][hm_S][hm_hd] (qkv for cvlm_attention layout 1) */
    float out_scale;             /* ABI 2 */
    void* workspace;             /* ABI 2 */
    int64_t workspace_bytes;     /* ABI 2 */
    /* ABI 3 -- LayerNorm folded into the GEMM that consumes it (image_encoder.py:432,444 + :491 / common.py:25): with
     * W' = W.diag(gamma) packed as the weight, bias' = bias + W.beta and ln_colsum[n] = sum_k W'[n][k],
     *     out = act( rstd_m * (alpha * acc - mu_m * ln_colsum[n]) + bias'[n] ),   mu = mean of row m,  rstd = 1 / sqrt(var + eps)
     * of the UN-normalised input (A holds x, possibly scaled: alpha carries the inverse).  ABI 5: ln_stats[m] = (rstd_m,
     * mu_m * rstd_m), float [M][2], as cvlm_ln_stats_merge writes it from the piece statistics of the producing launch
     * (`row_stats` below / cvlm_row_stats_split); ln_eps / ln_D are not read by the GEMM any more (they are arguments of the
     * merge).  Rows the merge refused (|mu| / sigma beyond CVLM_LN_FOLD_MAX_RATIO) carry NaN: every output of such a row is NaN.
     * h2 output only, act in {NONE, GELU, QUICKGELU}, N % 8 == 0. */
    const float* ln_stats; const float* ln_colsum; float ln_eps; int32_t ln_D;
    /* ABI 3 -- the producer side: residual given as h2 planes (value = (hi + lo) * res_scale, leading dimension ldrh) and
     * row_stats[p][m] = (sum, centred sum of squares) of the final values v of columns [64p, 64p + 64) of row m (ABI 5:
     * float [ceil(N / 64)][M][2], plain stores -- nothing to zero, no atomics; ABI 3/4 accumulated (sum, sum of squares) with
     * atomic adds).  h2 output only, act NONE, N % 8 == 0.  Together the two forms keep a pre-norm residual stream in h2 between the
     * GEMMs of a transformer block with no separate LayerNorm pass. */
    const void* res_hi; const void* res_lo; int64_t ldrh; float res_scale;
    float* row_stats;
    /* ABI 4 -- implicit 3x3 / pad 1 / stride 1 convolution (conv_c > 0): A is an NHWC image in h2 planes, rows m = (b, y, x)
     * on a conv_h x conv_w grid with conv_c channels (lda = conv_c), and K = 9 * conv_c runs over the taps,
     * k = (ky*3 + kx)*conv_c + c -- the layout cvlm_im2col3x3 materialises; here the gather happens in the DMA source
     * addresses and a tap outside the image reads zeros.  conv_c a power of two >= 32, one problem per launch.
     * Replaces the 3x3 convolutions of image_encoder.py:150 (neck) and mask_decoder_edge.py:88-93 (edge feature head). */
    int32_t conv_h, conv_w, conv_c;
    /* ABI 6 -- optional second image of the SAME weight with its planes interleaved per 32 k-elements:
     *     w_il[n][k / 32][plane][k % 32]   (fp16; plane 0 = hi, 1 = lo; row n starts at w_il + n * ldw_il, ldw_il >= 2 * K halves)
     * so that the 64 bytes of hi and the 64 bytes of lo a K-tile needs from a weight row are ONE 128-byte line.  The big-tile kernels
     * stage the weight operand from this image when it is given (8 rows x 128 bytes per DMA instruction instead of 16 rows x 64
     * bytes: every L2 line is requested once, not once per plane-half -- 1.4 % of the GEMM's energy at the power cap,
     * profiles/r03_wil_ab.log); the other kernels, and every kernel when it is NULL, read w_hi / w_lo.  Same bits either way.
     * NULL for batched launches and implicit convolutions. */
    const void* w_il; int64_t ldw_il;
    /* ABI 6 -- the same 128-byte-row image for ACTIVATIONS that only GEMMs touch (the h2 residual stream and the MLP hidden
     * activations of a transformer block): t_il[m][c / 32][plane][c % 32], one pointer, row stride in halves.
     *   a_il:   a_hi is such an image (a_lo unused), lda its row stride; needs w_il
     *   out_il: out_hi is such an image (out_lo unused), ldoh its row stride; h2 output of the LDS-staged epilogues, not head-major
     *   res_il: res_hi is such an image (res_lo unused), ldrh its row stride
     * A column offset c0 (c0 % 32 == 0) into an image is the pointer offset 2 * c0 halves. */
    int32_t a_il, out_il, res_il;
    /* ABI 10 -- `mx` operands: the two CORRECTION products of the split (lo.hi and hi.lo, each 2^-11 of hi.hi) on gfx950's block-scaled
     * fp8 matrix instruction (v_mfma_scale_f32_16x16x128_f8f6f4, e4m3 operands: twice the fp16 rate, ~2.1 x the flops per joule), the
     * main product hi.hi on fp16 as before:  acc += Whi.Ahi + Whi8.Alo8 + Wlo8.Ahi8, fp32 accumulation.  The corrections need ~5
     * significant bits for a 2^-16 product; measured on the reference's outputs at the demo geometry (tools/precision_emulate.py,
     * profiles/r05_precision_emulation.log): mask 1.3e-4 / IoU 0.99998 with qkv, lin1 and lin2 of all 32 ViT-H blocks in this form
     * (3 x fp16: 4e-5; fp16 alone: 1.5e-2; gate 1e-3).
     * An mx operand is an IMAGE plus a SCALE array, both derived from the h2 planes (hi, lo) of the same values:
     *   image   uint8 t[r][c / 64][256]  bytes 0-127: fp16 hi of the group's 64 columns; 128-191: hi8 (e4m3) of the same; 192-255: lo8
     *           (row stride given in HALVES like the *_il images, >= 2 * C; a column offset c0, c0 % 64 == 0, is 2 * c0 halves)
     *   scales  uint8 s[r][4][ld_s]      plane 0 / 1: exponent E (E8M0, value 2^(E - 127)) of hi8 for columns 0-31 / 32-63 of group u
     *           (byte u), plane 2 / 3: the same for lo8 (E - 11).  E = exponent field of the largest |hi| of the 32 columns as an f32,
     *           at least 103, minus 7;  hi8 = e4m3_rne(hi / 2^(E - 127)),  lo8 = e4m3_rne(lo / 2^(E - 138)).  ld_s % 4 == 0, >= C / 64.
     *   a_mx:   a_hi = image (lda its row stride), a_mxs / lda_s = scales; needs w_mx; K % 64 == 0.  Served by the 256 x 256 kernel.
     *   w_mx / ldw_mx, w_mxs / ldw_s: the weight in the same form.
     *   out_mx: out_hi = image (ldoh), out_mxs / ldo_s = scales; out_lo, when given, receives the fp16 lo PLANE [M][ldol] (what a later
     *           launch reads as the lo half of its h2 residual: the residual stream keeps its 22 bits).  LDS-staged epilogues, N % 64 == 0
     *           at 64-aligned columns, not head-major.
     *   res_mx: res_hi = image (ldrh; its hi halves are read), res_lo = fp16 lo plane [M][ldrl]. */
    int32_t a_mx, out_mx, res_mx;
    const void* a_mxs; int64_t lda_s;
    const void* w_mx; int64_t ldw_mx; const void* w_mxs; int64_t ldw_s;
    void* out_mxs; int64_t ldo_s; int64_t ldol; int64_t ldrl;
    /* ABI 12 -- head-major stores (hm_S > 0): bit w of hm_nolo (0 = q, 1 = k, 2 = v) set: the lo plane of that third of the columns is
     * NOT written.  cvlm_attention with split_qk == 1 never reads K's lo plane: the qkv projection of a ViT-H block then leaves a sixth
     * of its 503 MB of stores unwritten (hm_nolo = 2).  0: every plane is written, as before. */
    int32_t hm_nolo;
} cvlm_gemm_args;
int cvlm_gemm(const cvlm_gemm_args* args, void* stream);
int64_t cvlm_gemm_workspace_bytes(void);

/* Row LayerNorm over the last axis: y = LN(x + add) * gamma + beta, biased variance, then act.
 * Replaces nn.LayerNorm (image_encoder.py:432,444; alpha_clip_rw/model.py:162-168;
 * transformer_maskdecoder_edge.py:180-212) and LayerNorm2d on NHWC rows (common.py:31-43).
 * x f32 [M][D] (ldx); add optional f32 [M % add_rows][D]; sum_out optional f32 (x + add);
 * outputs: out_f32 and/or h2 (ld = D). */
int cvlm_layernorm(const float* x, int64_t ldx, const float* add, int32_t add_rows, float* sum_out,
                   const float* gamma, const float* beta, float eps, int32_t act,
                   float* out_f32, void* out_hi, void* out_lo, int32_t M, int32_t D, void* stream);

/* out = a + b[m % b_rows] (row-broadcast add), f32 and/or h2 outputs; scale applied to the sum.
 * Replaces the PE adds / `x + pos_embed` / `src + dense` (image_encoder.py:140,
 * transformer_maskdecoder_edge.py:177-209, mask_decoder_edge.py:157). */
int cvlm_add_rows(const float* a, const float* b, int32_t b_rows, float scale, float* out_f32,
                  void* out_hi, void* out_lo, int32_t M, int32_t D, void* stream);

/* Piece statistics -> the pair per row the LayerNorm-folded cvlm_gemm reads (ABI 5): merged[m] = (rstd_m, mu_m * rstd_m) from
 * pieces[p][m] = (sum, centred sum of squares) of columns [64p, 64p + 64) of row m, p < ceil(D / 64), piece planes `piece_rows`
 * rows apart.  The pieces of a row are added in index order, as centred moments (no s2 / D - mu^2 cancellation): the result is
 * bit-reproducible.  Guaranteed range of the fold: `alpha * acc - mu * colsum` costs |mu| / sigma of the h2 format's 22 bits
 * (measured 4e-6 * |mu| / sigma abs per output); a row with |mu| / sigma > CVLM_LN_FOLD_MAX_RATIO is refused -- its pair is
 * NaN and, when `gemm_workspace` (a cvlm_gemm workspace) is given, word CVLM_WS_WORD_LN_REFUSED of its first page counts it:
 * never a finite wrong value.  A separate launch because the merge inside the consuming GEMM (piece loads between its main loop
 * and its epilogue) cost that GEMM 7-8 %; this kernel takes ~3 us.  Replaces the statistics half of nn.LayerNorm
 * (image_encoder.py:432,444; alpha_clip_rw/model.py:315-362). */
#define CVLM_LN_FOLD_MAX_RATIO 128
#define CVLM_WS_WORD_LN_REFUSED 513
int cvlm_ln_stats_merge(const float* pieces, int64_t piece_rows, int32_t M, int32_t D, float eps, float* merged, void* gemm_workspace,
                        void* stream);

/* Row statistics + split: out h2 = x * scale (both planes), stats[p][m] = (sum, centred sum of squares) of columns
 * [64p, 64p + 64) of the unscaled row m, piece planes `stats_rows` rows apart (the piece layout cvlm_ln_stats_merge reads; D % 8 == 0).
 * Seeds the h2 residual stream of the LayerNorm-folded GEMMs from an f32 tensor x [M][D].
 * copies > 1 (ABI 4): the same M rows are written `copies` times, copy c at rows c * dst_row_stride of out / stats -- the
 * MaPLe deep visual prompts that replace the last n_ctx tokens of every image before a block
 * (alpha_clip_rw/model.py:392-434) on an h2 stream: out = planes + first_row * D, stats + 2 * first_row, stride = L. */
int cvlm_row_stats_split(const float* x, float scale, void* out_hi, void* out_lo, float* stats, int64_t stats_rows, int32_t M,
                         int32_t D, int32_t copies, int64_t dst_row_stride, void* stream);
/* ABI 10: the same with the rows written as an mx operand (cvlm_gemm_args above: image `out_img` with row stride ld_img halves, block
 * exponents `out_scales` [rows][4][ld_s], fp16 lo plane `out_lo` [rows][ld_lo]); D % 64 == 0.  The pointers address the first
 * destination row.  Seeds (and, for the deep prompts, overwrites rows of) the CLIP tower's residual stream when that stream is an mx
 * operand of the LayerNorm-folded in_proj / c_fc GEMMs (alpha_clip_rw/model.py:392-434, 296-300). */
int cvlm_row_stats_split_mx(const float* x, float scale, void* out_img, int64_t ld_img, void* out_scales, int64_t ld_s, void* out_lo,
                            int64_t ld_lo, float* stats, int64_t stats_rows, int32_t M, int32_t D, int32_t copies, int64_t dst_row_stride,
                            void* stream);

/* f32 -> h2 planes (elementwise split), n elements.  No reference counterpart: it produces the operand format of
 * cvlm_gemm / cvlm_attention from tensors the reference keeps in fp32 (e.g. the sparse prompts, models/sam_maskdecoder_edge.py:342-344). */
int cvlm_split_f32(const float* x, void* out_hi, void* out_lo, int64_t n, void* stream);

/* Patch gather for stride==kernel convolutions (image_encoder.py:651-659, :369-380;
 * alpha_clip_rw/model.py:529-531).  src0 (B,C0,H,W) f32 and optional src1 (B,C1,H,W) f32 are cut
 * into p x p patches; row m = (b, py, px); column k = c*p*p + iy*p + ix with the channels of src1
 * appended after src0; columns K..ldk-1 are zero.  Output h2 [B*(H/p)*(W/p)][ldk]. */
int cvlm_patchify(const float* src0, int32_t C0, const float* src1, int32_t C1, int32_t B, int32_t H,
                  int32_t W, int32_t p, void* out_hi, void* out_lo, int32_t ldk, void* stream);

/* 3x3 / pad 1 / stride 1 im2col on NHWC f32 (B,H,W,C): row m = (b,y,x), column k = (ky*3+kx)*C + c,
 * zero padded borders.  Output h2 [B*H*W][9*C].  Serves the neck conv (image_encoder.py:106-112)
 * and, with flipped weights, ConvTranspose2d(3,1,1) (mask_decoder_edge.py:88-93). */
int cvlm_im2col3x3(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, void* out_hi, void* out_lo,
                   void* stream);

/* Matrix transpose-reinterpretation used by PromptGenerator.init_embeddings (image_encoder.py:278-281):
 * per image, the (T x D) f32 token matrix is re-read as (D x T) and transposed:
 * out[b][t][c] = scale * x_flat[b][c*T + t].  Output h2 [B*T][D].  (scale: a power of two that moves the raw
 * residual stream into fp16 range; the consuming GEMM folds 1 / scale into its alpha.) */
int cvlm_reinterpret_transpose(const float* x, int32_t B, int32_t T, int32_t D, float scale, void* out_hi, void* out_lo,
                               void* stream);

/* Fused multi-head attention (flash style, scores never materialised), replaces
 *   image_encoder.py:488-504 + 589-625 (ViT-H window / global attention with decomposed rel-pos),
 *   image_encoder.py:507-553 (window partition / unpartition incl. zero padding after norm1),
 *   alpha_clip_rw/model.py:223-256 (CLIP ViT-L attention) and nn.MultiheadAttention with the causal
 *   mask (alpha_clip_rw/model.py:388-390, 751-757).
 * qkv: h2 [B*S_img][3*heads*hd] rows = tokens, columns [q | k | v], each heads*hd wide (qkv_layout 0),
 *      or head-major [3][B][heads][S_img][hd] as written by cvlm_gemm with hm_S > 0 (qkv_layout 1).
 * mode 0: plain (S = tokens per image); mode 1: global with rel-pos on a grid x grid token map;
 * mode 2: windows of `window` x `window` tokens on a grid x grid map, zero padded: pad tokens carry
 *         q = k = v = qkv bias (pad_hi/pad_lo = h2 of the 3*heads*hd bias vector).
 * rel_h/rel_w: h2 [(2*L-1)][hd] tables (L = grid or window).  scale = hd^-0.5 applied to q.k only.
 * out: h2 [B*S_img][heads*hd].
 * split_qk / split_pv: fp16 MFMA products per multiply of q.k^T / P.v.  3: hi/lo operands on both sides (hi.hi + lo.hi + hi.lo, fp32-grade);
 *   1: hi planes only (lo pointers may be NULL); (3, 1) mixes them.  ABI 11 -- (2, 2): K and V keep their lo planes, Q and the
 *   probabilities P do not: q.k^T = q_hi.(k_hi + k_lo), P.v = fp16(P).(v_hi + v_lo) with P rounded to nearest and the softmax denominator
 *   summed from the rounded values (a ones-row product), so the quotient is an exact weighted mean of V rows with weights off by
 *   <= 2^-12 relative.  ABI 12 -- (1, 2): K enters the scores as its hi plane too, q.k^T = q_hi.k_hi with ONE MFMA per k-step; K's lo
 *   plane is neither fetched nor read (the rel-pos tables keep q_hi + q_lo against both table planes, P.v is that of (2, 2)).
 *   Measured on the 16 reference images (profiles/r06_probe_kv_lo.log, profiles/r06_precision_emulate_attn.log): fp16(k) costs the masks
 *   4e-5 -- less than fp16(q); fp16(v) costs 4e-4 and stays out.  (Round 5's probe builds said otherwise for K and V: their kernels
 *   requested LDS fragments that no instruction consumed, and the allocator reused the registers while the reads were in flight.)
 *   The two ViT-H kernels (mode 1 on 64x64 / 96x96 maps, mode 2 with 14x14 windows) have these forms; every other shape runs (2, 2) and
 *   (1, 2) as (3, 3).  All lo pointers are required, as for 3.
 * workspace: the split 3/3, 2/2 and 1/2 global kernels for the 64x64 and 96x96 maps keep V transposed
 * (cvlm_attention_workspace_bytes(args) bytes, 0 for every other mode); CVLM_E_WORKSPACE if it is missing there. */
typedef struct cvlm_attn_args {
    const void* qkv_hi; const void* qkv_lo;
    const void* pad_hi; const void* pad_lo;
    const void* relh_hi; const void* relh_lo;
    const void* relw_hi; const void* relw_lo;
    void* out_hi; void* out_lo;
    int32_t B, S, heads, hd;
    int32_t mode, grid, window, causal;
    int32_t split_qk, split_pv;
    float scale;
    int32_t qkv_layout;          /* 0: token-major [B*S][3][H][hd]; 1: head-major [3][B][H][S][hd] */
    void* workspace;             /* ABI 2 */
    int64_t workspace_bytes;     /* ABI 2 */
    /* ABI 9 -- mode 0 only: q_rows > 0 computes the attention output of the FIRST q_rows queries of every sequence only (whole
     * 128-query blocks: rows up to the end of the last block touched are written, the rest of `out` is left alone); keys and values
     * are all S rows as ever.  The last block of the CLIP vision tower feeds nothing but its class token (row 0) into ln_post
     * (alpha_clip_rw/model.py:558-561): its attention is called with q_rows = 1.  0 = every query. */
    int32_t q_rows;
} cvlm_attn_args;
int cvlm_attention(const cvlm_attn_args* args, void* stream);
int64_t cvlm_attention_workspace_bytes(const cvlm_attn_args* args);

/* Small fp32 attention for the two-way decoder (transformer_maskdecoder_edge.py:250-272):
 * q f32 [B][nq][heads*hd] (ldq), k,v f32 [B][nk][heads*hd]; out f32 [B][nq][heads*hd].
 * softmax(q.k / sqrt(hd)) v per head; any nq/nk.  Since ABI 7 this entry is cvlm_small_attention_h2 without the h2 output and shares
 * its limits: hd = 16 or 32 (anything else: CVLM_E_UNSUPPORTED), row pitches multiples of 4 floats and bases 16-byte aligned
 * (otherwise CVLM_E_BADARG) -- the scalar-load form that served other head dims and alignments is gone. */
int cvlm_small_attention(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                         float* out, int64_t ldo, int32_t B, int32_t nq, int32_t nk, int32_t heads, int32_t hd,
                         void* stream);
/* ABI 7: the same with the result also / only as h2 planes [B][nq][heads*hd] (row pitch ldoh halves) -- the operand of the
 * out_proj GEMM that follows every attention of the two-way transformer (transformer_maskdecoder_edge.py:268-271), so no
 * cvlm_split_f32 launch sits between them.  `out` and (out_hi, out_lo) are each optional, one of them is required.  q / k / v may
 * be column blocks of one wider matrix (merged q|k|v projection): pass the block's first column and the matrix's row pitch.
 * hd = 16 or 32; row pitches multiples of 4, bases 16-byte aligned. */
int cvlm_small_attention_h2(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, float* out, int64_t ldo,
                            void* out_hi, void* out_lo, int64_t ldoh, int32_t B, int32_t nq, int32_t nk, int32_t heads, int32_t hd,
                            void* stream);

/* Dense positional encoding (models/sam_maskdecoder_edge.py:90-110): gauss f32 [2][C/2] ->
 * out f32 [size*size][C] (token-major: row = y*size+x). */
int cvlm_dense_pe(const float* gauss, int32_t size, int32_t C, float* out, void* stream);

/* Mask head (mask_decoder_edge.py:181-186 + models/sam_maskdecoder_edge.py:380-387):
 * up, edge_emb f32 NHWC [B][h][w][C]; hyper f32 [B][5][C] (rows 0..3 mask MLPs, row 4 edge MLP).
 * low[b][y][x] = m*sigmoid(e) + m with m = hyper[b][0].up, e = hyper[b][4].edge_emb  (mask 0 only).
 * edge_emb == NULL: low = m, the vanilla decoder's product (models/mmseg/models/sam/mask_decoder.py:139). */
int cvlm_mask_head(const float* up, const float* edge_emb, const float* hyper, int32_t B, int32_t HW, int32_t C,
                   float* low, void* stream);

/* Bilinear resize, align_corners=False (F.interpolate): in f32 [N][hin][win] -> out [N][hout][wout];
 * sigmoid_in != 0 applies sigmoid to the input first (demo.py:117-120). */
int cvlm_bilinear(const float* in, int32_t N, int32_t hin, int32_t win, float* out, int32_t hout, int32_t wout,
                  int32_t sigmoid_in, void* stream);

/* CLIP token assembly (alpha_clip_rw/model.py:532-544): patches f32 [B][P][W] -> tokens f32
 * [B][1+P+nctx][W] = [cls+pos0 | patches+pos | ctx]. */
int cvlm_clip_assemble(const float* patches, const float* cls, const float* pos, const float* ctx,
                       int32_t B, int32_t P, int32_t W, int32_t nctx, float* out, void* stream);

/* Overwrite `n` consecutive token rows starting at row `first` of every sequence with `src` [n][W]
 * (MaPLe deep prompts, alpha_clip_rw/model.py:319-355).  x f32 [B][L][W]. */
int cvlm_overwrite_rows(float* x, int32_t B, int32_t L, int32_t W, int32_t first, int32_t n, const float* src,
                        void* stream);

/* Gather one row per sequence: out[b] = x[b][idx[b]] (idx NULL -> row `fixed`), x f32 [B][L][W].
 * Replaces the CLS pick `x[:, 0, :]` (alpha_clip_rw/model.py:556) and the EOT pick
 * `x[arange, tokenized_prompts.argmax(-1)]` (cocotrainers/mapleAlphaCLIP.py:76). */
int cvlm_gather_rows(const float* x, int32_t B, int32_t L, int32_t W, const int32_t* idx, int32_t fixed,
                     float* out, void* stream);
/* ABI 5: the same pick from a residual stream kept in h2 planes: out[b] = (hi + lo)[b][idx[b]] * scale (f32).  The class-token
 * rows at the end of the LayerNorm-folded CLIP vision tower (alpha_clip_rw/model.py:556). */
int cvlm_gather_rows_h2(const void* x_hi, const void* x_lo, float scale, int32_t B, int32_t L, int32_t W, const int32_t* idx,
                        int32_t fixed, float* out, void* stream);

/* CLIP head (cocotrainers/mapleAlphaCLIP.py:289-294): img f32 [B][D] (un-normalised), txt f32 [C][D]
 * (= normalise(text features) + bank, precomputed), logit_scale_exp.  Outputs: img_n [B][D],
 * logits [B][C], pred int64 [B], txt_sel [B][D] = txt[pred]. */
int cvlm_clip_head(const float* img, const float* txt, float logit_scale_exp, int32_t B, int32_t C, int32_t D,
                   float* img_n, float* logits, int64_t* pred, float* txt_sel, void* stream);

/* Row L2 normalise + add: out[r] = x[r]/||x[r]|| + add[r] (text bank, mapleAlphaCLIP.py:290-291). */
int cvlm_normalize_add(const float* x, const float* add, int32_t R, int32_t D, float* out, void* stream);

/* ---- N1: preprocessing before the path (demo.py:93-107, datasets/wrappers.py:22-27,
 * alpha_clip_rw/alpha_clip.py:79-94).  torchvision Resize on a PIL image == PIL.Image.resize (Pillow
 * libImaging/Resample.c, 8-bit path).  One separable pass of that resample on uint8 NHWC images:
 * output index o along `axis` (0 = rows, 1 = columns) = clamp8((2^21 + sum_t src[lo_o + t] * kk[o][t]) >> 22),
 * (lo_o, count_o) = bounds[o]; bounds/kk are the host-side Pillow coefficient tables (int32, device memory). */
int cvlm_resample_u8(const uint8_t* src, int32_t N, int32_t H, int32_t W, int32_t C, const int32_t* bounds,
                     const int32_t* kk, int32_t ksize, int32_t n_out, int32_t axis, uint8_t* dst, void* stream);

/* ToTensor + Normalize (+ CenterCrop window): uint8 [N][H][W][C] -> f32 [N][C][ch][cw] = (x/255 - mean[c]) / std[c].
 * Replaces transforms.ToTensor / Normalize of demo.py:95-97 and CenterCrop / ToTensor / Normalize of
 * alpha_clip_rw/alpha_clip.py:83-85. */
int cvlm_u8_to_tensor(const uint8_t* src, int32_t N, int32_t H, int32_t W, int32_t C, int32_t top, int32_t left,
                      int32_t ch, int32_t cw, const float* mean, const float* stdv, float* dst, void* stream);

/* ---- evaluation tail ----------------------------------------------------------------------------
 * Replaces `torch.sigmoid` + `.cpu().numpy()` + cv2 `resize` + `(pred * 255).astype(np.uint8)`
 * (test_ovcos_maskdecoder_edge.py:103,116-130): logits f32 [N][Hs][Ws] -> uint8 [N][h][w]. */
int cvlm_mask_to_u8(const float* logits, int32_t N, int32_t Hs, int32_t Ws, int32_t h, int32_t w, uint8_t* dst, void* stream);

/* Replaces the per-pixel numpy passes of OVCOSMetricer.step (recorder/ovcos_metricer.py:8-180, pysodmetrics 1.4.2):
 * pre/gt uint8 [N][h][w] -> stats u64 [N][3] = (count, sum x, sum y) of gt > 128 and
 * hist u32 [N][4][2][256] = per S-measure quadrant (LT, RT, LB, RB around the gt centroid), per gt class, per level.
 * Both outputs are zeroed by the call.  MAE / F / E / S measures and IoU are functions of these counters alone. */
int cvlm_mask_joint_hist(const uint8_t* pre, const uint8_t* gt, int32_t N, int32_t h, int32_t w, uint64_t* stats, uint32_t* hist,
                         void* stream);

/* Weighted F-measure ingredients (pysodmetrics 1.4.2 WeightedFmeasure.cal_wfm behind recorder/ovcos_metricer.py:49-66):
 * exact Euclidean distance transform with nearest-foreground index (scipy's tie order), E carried over from the nearest
 * foreground pixel, 7x7 Gaussian (gauss49: the 49 f64 weights, device memory), pixel importance, all in f64.
 * hist = the counters of cvlm_mask_joint_hist for the same images (gives the min / max level of each mask).
 * workspace: N*h*w*16 + N*ceil(h*w/256)*24 + N*8 bytes of device memory.  out3 f64 [N][3] = (sum Ew over gt, sum Ew over
 * ~gt, |gt|): wfm = 2 R P / (R + P) with TPw = |gt| - out[0], P = TPw / (TPw + out[1]), R = 1 - out[0] / |gt|. */
int cvlm_mask_wfm(const uint8_t* pre, const uint8_t* gt, int32_t N, int32_t h, int32_t w, const uint32_t* hist,
                  const double* gauss49, void* workspace, double* out3, void* stream);

/* ABI 8 -- `utils.calc_cod` (utils.py:143-165): the loop's second metric set, Sm / Em / wFm / MAE of the FLOAT probability map (the
 * in-tree classes recorder/sod_metric.py:39-581 fed `y_pred * 255` as float32: no uint8 step).  prob f32 [N][h][w] (sigmoid output),
 * gt u8 [N][h][w] (> 128 = foreground).  Three calls around cvlm_mask_joint_hist, each with the same caller-owned workspace of
 * max(N*h*w*16 + N*ceil(h*w/256)*24 + N*8, N*8192) bytes:
 *   cvlm_prob_quantise: minmax f32 [N][2] = min / max of fl(fl(p * 255) / 255) (`_prepare_data`, sod_metric.py:12-26) and
 *                       q u8 [N][h][w] = uint8(pn * 255), pn = (v - min) / (max - min) in float32 -- the levels the E-measure's
 *                       cumulative histograms count (:420); then cvlm_mask_joint_hist(q, gt) gives centroid + counters;
 *   cvlm_prob_moments:  out f64 [N][4 quadrants][2 classes][2] = (sum pn, sum pn^2): S-measure object / region terms and MAE;
 *   cvlm_prob_wfm:      the weighted F-measure's three sums as cvlm_mask_wfm, E = |pn - gt| from the float map. */
int cvlm_prob_quantise(const float* prob, int32_t N, int32_t h, int32_t w, float* minmax, uint8_t* q, void* workspace, void* stream);
int cvlm_prob_moments(const float* prob, const uint8_t* gt, int32_t N, int32_t h, int32_t w, const float* minmax, const uint64_t* stats,
                      void* workspace, double* out, void* stream);
int cvlm_prob_wfm(const float* prob, const uint8_t* gt, int32_t N, int32_t h, int32_t w, const float* minmax, const double* gauss49,
                  void* workspace, double* out3, void* stream);

/* Replaces Classification.process (recorder/new_evaluator.py:47-59): scores f32 [B][C], labels i32 [B] ->
 * pred i32 [B] (may be NULL) and counters u32 [3] += (top-1 hits, top-5 hits, rows).  Counters are NOT zeroed. */
int cvlm_topk_accumulate(const float* scores, const int32_t* labels, int32_t B, int32_t C, int32_t* pred, uint32_t* counters,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CVLM_H */
