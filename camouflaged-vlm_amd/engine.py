"""Weight packing and kernel launch schedule of the cascaded forward pass on one MI355X.

Everything numeric is a launch of libcvlm_hip.so through ``hip.py``; torch only owns device
buffers (allocation, slicing, copies of constants).  All activations are token-major (NHWC): a
(B, C, H, W) tensor of the reference is a [B*H*W][C] matrix here, so every Linear / 1x1 conv /
ConvTranspose(2,2) is one NT GEMM and LayerNorm2d is a row LayerNorm.

Schedules follow the reference line by line (cited per method); numeric format per tensor:
f32 for residual streams and module outputs, h2 (split-half planes) for every GEMM/attention operand.
"""
from __future__ import annotations

import dataclasses
import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import hip
from .hip import ACT_ABS_POST, ACT_GELU, ACT_NONE, ACT_QUICKGELU, ACT_RELU, H2
from .spec import ClipGeometry, SamGeometry


@dataclass(frozen=True)
class Precision:
    """split = 3: hi*hi + lo*hi + hi*lo products (~22 significant bits per operand, the reference-grade mode `exact`); 1: fp16 operands.
    mx: the GEMMs whose operands only GEMMs touch (qkv / lin1 / lin2 of the ViT-H blocks, the CLIP MLPs) form the two correction products
    lo*hi and hi*lo on the block-scaled e4m3 matrix instruction (include/cvlm.h ABI 10); the other GEMMs are split = 3.
    qk / pv (the ViT-H attention kernels): (2, 2), ABI 11 -- the products keep the lo planes of K and V and drop those of Q and of the
    probabilities (one fp16 per probability, the softmax denominator summed from the rounded values); (1, 2), ABI 12 -- K enters the
    scores as its hi plane too: ONE MFMA per k-step of q.k^T, two per step of P.v.
    `mx` = the mx GEMMs + (1, 2) attention, on batches of two or more images; one image per call runs the `exact` arithmetic
    (SamEncoder.attn_split), and a batch's first use of the mode on a set of weights is preceded by Cascade._mx_self_check.
    `mx22` / `mx33`: the mx GEMMs with (2, 2) / (3, 3) attention (A/B modes of bench.py)."""
    gemm: int = 3
    qk: int = 3
    pv: int = 3
    mx: bool = False

    @staticmethod
    def named(name: str) -> "Precision":
        return {"exact": Precision(3, 3, 3), "mx": Precision(3, 1, 2, True), "mx33": Precision(3, 3, 3, True), "mx22": Precision(3, 2, 2, True),
                "mx12": Precision(3, 1, 2, True), "fast": Precision(1, 1, 1), "mixed": Precision(3, 3, 1)}[name]


# Static power-of-two scales that keep UNBOUNDED activations inside fp16 range when they become h2 GEMM operands
# (DESIGN.md §3): LayerNorm outputs are bounded by sqrt(D) * max|gamma| and everything downstream of them by the
# weights; what is not bounded is (a) the raw residual stream (kept in h2 between the GEMMs of a block; fed to a Linear
# directly by PromptGenerator.init_embeddings, image_encoder.py:278-281, and by the neck, :150) and (b) MLP hidden
# activations (common.py:25, alpha_clip_rw/model.py:296-300).  Stored value = true value * scale; the consuming GEMM's
# alpha carries 1 / scale.  Guaranteed range: |x| < 65504 / scale = 2.6e5.  The shift is kept small on purpose: the lo
# plane of a value v is ~ v * 2^-11, and below 6.1e-5 fp16 goes subnormal (absolute step 6e-8) -- with 2^-8 / 2^-6
# (first choice of this round) the lo planes of O(1) activations lost most of their bits and the tiny cascade's error
# grew from 3e-5 to 1.4e-4; with 2^-2 values down to ~0.1 keep all 22 bits.
X_SCALE = 2.0 ** -2
HID_SCALE = 2.0 ** -2


def _ceil(a: int, b: int) -> int:
    return (a + b - 1) // b * b


class Workspace:
    """Named device buffers, reused across calls.  One flat buffer per name, sized for the largest request seen so
    far; a request hands out a view of its head, so a changing batch size (a ragged last batch) re-uses the same
    memory instead of keeping one full buffer set per batch size.  Peak at ViT-H 1024^2: ~0.35 GB per image.
    Also owns the scratch memory of the two C-ABI entries that need some (include/cvlm.h: cvlm_gemm's split-K slabs,
    cvlm_attention's transposed V).  One Workspace belongs to one engine object and is used on one stream at a time."""

    def __init__(self, device):
        self.device = device
        self._flat: Dict[tuple, torch.Tensor] = {}
        self._gemm_ws: Optional[torch.Tensor] = None

    def _get(self, kind: str, name: str, numel: int, dtype, zero: bool) -> torch.Tensor:
        t = self._flat.get((kind, name))
        if t is None or t.numel() < numel:
            t = (torch.zeros if zero else torch.empty)(numel, dtype=dtype, device=self.device)
            self._flat[(kind, name)] = t
        return t[:numel]

    def f32(self, name: str, *shape: int) -> torch.Tensor:
        return self._get("f32", name, int(np.prod(shape)), torch.float32, False).view(shape)

    def h2(self, name: str, *shape: int, zero: bool = False) -> H2:
        n = int(np.prod(shape))
        flat = self._get("h2", name, 2 * n, torch.float16, zero)
        return H2(flat.view((2,) + tuple(shape)))

    def h2il(self, name: str, M: int, C: int) -> hip.H2IL:
        """Activation in the 128-byte-row image (include/cvlm.h ABI 6): fp16 [M][2 * C], C % 32 == 0."""
        flat = self._get("h2", name, 2 * M * C, torch.float16, False)
        return hip.H2IL(flat.view(M, 2 * C))

    def h2mx(self, name: str, M: int, C: int, lo_plane: bool = False) -> hip.H2MX:
        """Activation as an mx operand (include/cvlm.h ABI 10): image uint8 [M][4 * C], block exponents uint8 [M][4][pitch] and,
        on request, the fp16 lo plane [M][C] beside it (the residual stream)."""
        img = self._get("u8", name, 4 * M * C, torch.uint8, False).view(M, 4 * C)
        su = hip.mx_scale_pitch(C)
        sc = self._get("u8", name + ".scales", 4 * M * su, torch.uint8, True).view(M, 4, su)
        lo = self._get("h2", name + ".lo", M * C, torch.float16, False).view(M, C) if lo_plane else None
        return hip.H2MX(img, sc, lo, C)

    def scratch(self, name: str, nbytes: int) -> Optional[torch.Tensor]:
        return self._get("u8", name, nbytes, torch.uint8, False) if nbytes > 0 else None

    def gemm_ws(self) -> torch.Tensor:
        if self._gemm_ws is None:
            self._gemm_ws = hip.new_gemm_workspace(self.device)      # zero-filled once: the hand-off page starts clean
        return self._gemm_ws

    def gemm_errors(self) -> int:
        return 0 if self._gemm_ws is None else hip.gemm_workspace_errors(self._gemm_ws)


class Linear:
    """Packed weight of one NT GEMM: rows scaled by a power of two so the lo plane stays in fp16's
    normal range, K zero-padded to a multiple of 32, optional N padding (zero rows)."""

    def __init__(self, w: torch.Tensor, b: Optional[torch.Tensor], device, n_pad: int = 0, k_pad: int = 0, il: bool = True, mx: bool = False):
        # Packed ON THE DEVICE (round 5): the planes, the 128-byte-row image and the mx image of 1.04 G parameters were 25-60 s of host
        # time per process -- times the ranks of a node that share its CPU quota; the host only reshapes.  Same arithmetic (fp16 / e4m3
        # round-to-nearest conversions of torch on either device; tests/test_gemm_mx_gpu.py holds the device's mx_pack to the host's).
        w = w.detach().float().reshape(w.shape[0], -1)
        N, K = w.shape
        self.N = max(N, n_pad)
        self.K = max(_ceil(K, 32), k_pad)
        wp = torch.zeros(self.N, self.K, device=device)
        wp[:N, :K] = w.to(device)
        mx_abs = float(wp.abs().max())
        e = 0 if mx_abs == 0.0 else int(math.floor(math.log2(2048.0 / mx_abs)))
        e = max(min(e, 24), -24)
        self.alpha = float(2.0 ** (-e))
        planes = H2.pack(wp * (2.0 ** e))
        del wp
        self.w = planes
        # third image: the mx operand (cvlm_gemm_args.w_mx, ABI 10) for the launches whose activation is an mx image
        self.w_mx = None
        if mx and self.K % 64 == 0 and self.N * self.K >= (1 << 18):
            self.w_mx = hip.H2MX.from_planes(planes)
        # second image with the planes interleaved per 32 k-elements (cvlm_gemm_args.w_il, ABI 6): what the big-tile kernels stage
        # the weight from; small matrices never reach those kernels, and `il=False` marks weights that only the tap / fallback
        # schedules launch (the unfused twins of LayerNorm-folded GEMMs): they stay planar and cost no second copy
        self.w_il = hip.interleave_planes(self.w) if il and self.N * self.K >= (1 << 18) else None
        self.bias = None
        if b is not None:
            bp = torch.zeros(self.N)
            bp[:N] = b.detach().float().cpu()
            self.bias = bp.to(device)


class LnLinear(Linear):
    """Linear that consumes a LayerNorm, with the norm folded in (include/cvlm.h, cvlm_gemm_args.ln_stats):
        Linear(LN(x)) = rstd * (x . W'^T - mu * colsum) + bias',   W' = W . diag(gamma),  bias' = bias + W . beta,
    colsum[n] = sum_k W'[n][k] -- taken from the PACKED planes so that it is the sum of exactly the numbers the MFMAs see."""

    def __init__(self, w: torch.Tensor, b: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor, device, mx: bool = False):
        w = w.detach().double().reshape(w.shape[0], -1).cpu()
        g, be = gamma.detach().double().cpu(), beta.detach().double().cpu()
        b2 = (b.detach().double().cpu() if b is not None else torch.zeros(w.shape[0], dtype=torch.float64)) + w @ be
        super().__init__((w * g[None, :]).float(), b2.float(), device, mx=mx)
        self.colsum = (self.w.t.double().sum(0).sum(1) * self.alpha).float().contiguous()


def implicit_conv_ok(C: int) -> bool:
    """cvlm_gemm's implicit 3x3 convolution wants a power-of-two channel count >= 32 (include/cvlm.h, ABI 4); the tiny test
    geometries fall back to cvlm_im2col3x3 + GEMM."""
    return C >= 32 and (C & (C - 1)) == 0


class _Base:
    def __init__(self, device, precision: Precision):
        self.device = device
        self.prec = precision
        self.ws = Workspace(device)
        self.ksplit = True          # False: GEMMs get no workspace, i.e. no K-split of any kind: the summation order is independent of M
        self.class_token_tail = True    # CLIP vision tower: the last block behind its qkv projection runs for the class tokens only

    def gemm(self, a: H2, lin: Linear, M: int, **kw) -> None:
        kw.setdefault("split", self.prec.gemm)
        alpha = kw.pop("alpha", 1.0)
        if "bias" not in kw:
            kw["bias"] = lin.bias
        w_il = lin.w_il if kw.get("batch", 1) == 1 and kw.get("conv3x3") is None and "ldw" not in kw else None
        hip.gemm(a, lin.w, M, kw.pop("N", lin.N), lin.K, alpha=alpha * lin.alpha,
                 workspace=self.ws.gemm_ws() if self.ksplit else None, w_il=w_il, w_mx=lin.w_mx if getattr(a, "mx", False) else None, **kw)

    def attention(self, qkv: H2, out: H2, B: int, S: int, heads: int, hd: int, **kw) -> None:
        hip.attention(qkv, out, B, S, heads, hd,
                      workspace=self.ws.scratch("attn_ws", hip.attention_workspace_bytes(B, S, heads, hd, **kw)), **kw)

    def dev(self, t) -> torch.Tensor:
        return torch.as_tensor(t).detach().float().contiguous().to(self.device)


# ================================================================================================
# SAM-Adapter ViT encoder  (models/mmseg/models/sam/image_encoder.py:132-155)
# ================================================================================================
def lowpass_matrices(N: int, line: int):
    """Real/imag parts of L = F^-1 diag(box) F with box = 1 on frequencies [-line, line): the
    centred square zeroed by PromptGenerator.fft (image_encoder.py:335-343) is x -> L x L^T, so the
    high-pass image is |x - Re(L x L^T)| = |x - Lr x Lr^T + Li x Li^T| (two dense projections)."""
    k = np.arange(-line, line, dtype=np.float64)
    d = np.arange(N, dtype=np.float64)
    ang = 2.0 * np.pi * np.outer(d, k) / N            # angle for (n - m) = d, frequency k
    c, s = np.cos(ang).sum(1) / N, np.sin(ang).sum(1) / N
    idx = (np.arange(N)[:, None] - np.arange(N)[None, :]) % N
    return torch.from_numpy(c[idx]).float(), torch.from_numpy(s[idx]).float()


class SamEncoder(_Base):
    def __init__(self, sd: Dict[str, torch.Tensor], g: SamGeometry, device, precision: Precision,
                 prefix: str = "image_encoder."):
        super().__init__(device, precision)
        self.g = g
        # residual stream / hidden rows as 128-byte-row images: CVLM_GEMM_AIL = 1 (default) always, b batches only (M > 4096), 0 never
        self.act_il = os.environ.get("CVLM_GEMM_AIL", "1") != "0"
        self.act_il_small = os.environ.get("CVLM_GEMM_AIL", "1") not in ("0", "b")
        P = prefix
        D, Pd = g.embed_dim, g.prompt_dim
        L = lambda name, **kw: Linear(sd[P + name + ".weight"], sd.get(P + name + ".bias"), device, **kw)
        self.patch = L("patch_embed.proj")
        self.pos = self.dev(sd[P + "pos_embed"].reshape(g.grid * g.grid, D))
        pg = "prompt_generator."
        self.PK = _ceil(Pd, 32)                                   # padded prompt width (40 -> 64)
        self.emb_gen = L(pg + "embedding_generator", n_pad=self.PK)
        self.hc_proj = L(pg + "prompt_generator.proj", n_pad=self.PK)
        self.shared = L(pg + "shared_mlp", k_pad=self.PK)
        self.light = [L(f"{pg}lightweight_mlp_{i}.0", n_pad=self.PK, k_pad=self.PK) for i in range(g.depth)]
        lr, li = lowpass_matrices(g.inp_size, g.fft_halfwidth)
        # [2N][KN]: the contraction length of cvlm_gemm is a multiple of 32, an input size need only be a multiple of the patch (16):
        # L gets zero columns up to KN = ceil32(N) (336 px -> 352), see `highpass`
        self.hp_k = _ceil(g.inp_size, 32)
        lpad = torch.zeros(2 * g.inp_size, self.hp_k)
        lpad[:, :g.inp_size] = torch.cat([lr, li], 0)
        self.lstack = H2(H2.pack(lpad).t.to(device))
        # Prompt fold: x_{i+1} = x_i + lin2_i(hid) + shared_mlp(prm_{i+1}) is ONE contraction over K = mlp_dim + PK,
        #   [hid | prm_{i+1}] . [W2_i | W_shared]^T + (b2_i + b_shared) + x_i,
        # so the adapter's per-block `x = prompt_i + x` (image_encoder.py:145) costs 1.25 % more MLP FLOPs instead of
        # a separate pass that reads and rewrites the whole residual stream (168 MB at B = 8, 32 times).
        shw = torch.zeros(D, self.PK)
        shw[:, :Pd] = sd[P + pg + "shared_mlp.weight"].detach().float().cpu()
        shb = sd[P + pg + "shared_mlp.bias"].detach().float().cpu()
        mx = precision.mx
        self.lin2cat = [Linear(torch.cat([sd[P + f"blocks.{i}.mlp.lin2.weight"].detach().float().cpu(), shw], 1),
                               sd[P + f"blocks.{i}.mlp.lin2.bias"].detach().float().cpu() + shb, device, mx=mx)
                        for i in range(g.depth - 1)]
        # LayerNorm folded into the GEMMs that consume it + the residual stream kept in h2 between them: no LayerNorm
        # kernels inside the blocks (64 launches, 4.5 ms per step at B = 8).  CVLM_LN_FOLD=0 (both towers) builds no folded weights
        # and runs the separate passes; `fold_disabled` switches to them at run time (Cascade's refusal guard).
        self.ln_fold = os.environ.get("CVLM_LN_FOLD", "1") == "1"
        self.fold_disabled = False          # set by Cascade's refusal guard: rows with |mu|/sigma > 128 met -> separate LayerNorm passes
        self.blocks = []
        # The attention's `(q * scale) k^T` (image_encoder.py:496) with the factor folded into the q rows of the qkv projection
        # (weight and bias; the pad tokens' q = bias row follows), and its inverse into the rel-pos tables, which the reference applies
        # to the UNSCALED q (:497-500): q' . (R / scale) = q . R.  The kernels are then called with scale = 1 and use the projected
        # planes as they are -- re-splitting 80 values per query per (window, head) pair was 5 % of the window kernel (262 -> 251 us,
        # tools/bench_attn.py SCALE=1).  Same mathematics, rounding points 2^-22 apart.
        qs = float(g.head_dim) ** -0.5

        def scaled_qkv(i):
            w = sd[P + f"blocks.{i}.attn.qkv.weight"].detach().float().cpu().clone()
            bq = sd[P + f"blocks.{i}.attn.qkv.bias"].detach().float().cpu().clone()
            w[:D] *= qs
            bq[:D] *= qs
            return w, bq
        for i in range(g.depth):
            b = f"blocks.{i}."
            qkv_w, qkv_b = scaled_qkv(i)
            blk = dict(
                n1w=self.dev(sd[P + b + "norm1.weight"]), n1b=self.dev(sd[P + b + "norm1.bias"]),
                n2w=self.dev(sd[P + b + "norm2.weight"]), n2b=self.dev(sd[P + b + "norm2.bias"]),
                # with the fold on, qkv / lin1 (and lin2 of every block but the last: lin2cat carries it) are launched by the tap
                # schedule and the refusal fallback only: planar, no 128-byte-row image (~1 GB at ViT-H)
                qkv=Linear(qkv_w, qkv_b, device, il=not self.ln_fold), proj=L(b + "attn.proj"), lin1=L(b + "mlp.lin1", il=not self.ln_fold),
                lin2=L(b + "mlp.lin2", il=not self.ln_fold or i == g.depth - 1, mx=mx and i == g.depth - 1),
                pad=H2(H2.pack(qkv_b).t.to(device)),
                rel_h=H2(H2.pack(sd[P + b + "attn.rel_pos_h"].detach().float().cpu() / qs).t.to(device)),
                rel_w=H2(H2.pack(sd[P + b + "attn.rel_pos_w"].detach().float().cpu() / qs).t.to(device)),
                window=0 if i in g.global_attn_indexes else g.window_size)
            if self.ln_fold:
                blk["qkv_f"] = LnLinear(qkv_w, qkv_b, sd[P + b + "norm1.weight"], sd[P + b + "norm1.bias"], device, mx=mx)
                blk["lin1_f"] = LnLinear(sd[P + b + "mlp.lin1.weight"], sd[P + b + "mlp.lin1.bias"],
                                         sd[P + b + "norm2.weight"], sd[P + b + "norm2.bias"], device, mx=mx)
            self.blocks.append(blk)
        self.neck0 = L("neck.0")
        w2 = sd[P + "neck.2.weight"].detach().float().cpu()                   # (O, I, 3, 3) -> (O, ky, kx, I)
        self.neck2 = Linear(w2.permute(0, 2, 3, 1).reshape(w2.shape[0], -1), None, device)
        self.nk1 = (self.dev(sd[P + "neck.1.weight"]), self.dev(sd[P + "neck.1.bias"]))
        self.nk3 = (self.dev(sd[P + "neck.3.weight"]), self.dev(sd[P + "neck.3.bias"]))

    # image_encoder.py:332-353
    def highpass(self, inp: torch.Tensor) -> torch.Tensor:
        B, C, N, _ = inp.shape
        ws, sp, KN = self.ws, self.prec.gemm, self.hp_k
        # KN > N (N not a multiple of 32): the image / P^T / Q^T rows keep their pitch N and the K-tiles of a row run up to 31 elements
        # into the next row -- finite numbers that meet the zero columns of L.  The last row of a plane runs into the next plane or
        # into `slack` elements behind it.  Those are zeroed on EVERY call: the workspace hands out the head of the largest
        # buffer seen so far, so after a larger batch the slack of a smaller one lies inside stale data, and 0 * inf is not 0.
        slack = KN - N
        xs = ws.h2("hp_x", B * C * N * N + slack)                   # flat planes: [B*C*N rows][N] + slack
        hip.split_f32(inp, xs)                                      # writes numel(inp) elements at the head of each plane
        pq = ws.h2("hp_pq", B * C * 2 * N * N + slack)              # per problem: rows [0,N) = P^T, [N,2N) = Q^T
        if slack > 0:
            xs.t[:, B * C * N * N:].zero_()
            pq.t[:, B * C * 2 * N * N:].zero_()
        hip.gemm(self.lstack, xs, 2 * N, N, KN, lda=KN, ldw=N, out_h2=pq, ldoh=N, batch=B * C, stride_a=0, stride_w=N * N,
                 stride_oh=2 * N * N, split=sp)
        t = ws.f32("hp_t", B, C, N, N)
        lr = H2(self.lstack.t[:, :N])
        li = H2(self.lstack.t[:, N:])
        hip.gemm(lr, pq, N, N, KN, lda=KN, ldw=N, residual=inp, out_f32=t, alpha=-1.0, batch=B * C, stride_a=0,
                 stride_w=2 * N * N, stride_r=N * N, stride_o=N * N, split=sp)
        qt = H2(pq.t[:, N * N:])                                    # Q^T of problem 0; same per-problem stride
        out = ws.f32("hp_out", B, C, N, N)
        hip.gemm(li, qt, N, N, KN, lda=KN, ldw=N, residual=t, out_f32=out, alpha=1.0, act=ACT_ABS_POST, batch=B * C, stride_a=0,
                 stride_w=2 * N * N, stride_r=N * N, stride_o=N * N, split=sp)
        return out

    # blocks whose launches the host issues before it runs `issue_hook` (Cascade: the side stream's CLIP launches)
    HOOK_AFTER_BLOCKS = 5

    def _hook(self, i: int) -> None:
        """Called after the launches of block i: runs the caller's issue hook once enough of the encoder is queued."""
        if self._issue_hook is not None and i + 1 >= min(self.HOOK_AFTER_BLOCKS, self.g.depth):
            hook, self._issue_hook = self._issue_hook, None
            hook()

    def forward(self, inp: torch.Tensor, taps: Optional[dict] = None, out_name: str = "features", issue_hook=None) -> torch.Tensor:
        """inp (B,3,S,S) f32 on device -> features f32 [B*G*G][out_chans] (token-major NHWC), in workspace buffer
        `out_name` (a caller that keeps two batches in flight alternates two names)."""
        self._out_name = out_name
        self._issue_hook = issue_hook
        try:
            feats = self._forward(inp, taps)
            if self._issue_hook is not None:                         # never fired (cannot happen for depth >= 1): run it now
                self._hook(self.g.depth)
            return feats
        finally:
            self._issue_hook = None                                  # an exception before block 5 must not leave a stale hook behind

    def _forward(self, inp: torch.Tensor, taps: Optional[dict]) -> torch.Tensor:
        g, ws, pr = self.g, self.ws, self.prec
        B = inp.shape[0]
        assert inp.shape[1:] == (3, g.inp_size, g.inp_size), \
            f"Input image size {tuple(inp.shape[2:])} doesn't match model ({g.inp_size}*{g.inp_size})."
        G, D, T = g.grid, g.embed_dim, g.grid * g.grid
        M, PK = B * T, self.PK
        KP = self.patch.K
        # :134 patch embed
        a0 = ws.h2("patches", M, KP)
        hip.patchify(inp, None, g.patch_size, a0, KP)
        x = ws.f32("x", M, D)
        self.gemm(a0, self.patch, M, out_f32=x)
        if taps is not None:
            taps["patch_embed"] = x.clone()
        # :136 init_embeddings -- (T x D) matrix re-read as (D x T) and transposed (reference quirk)
        xt = ws.h2("xn", M, D)
        hip.reinterpret_transpose(x, B, T, D, xt, scale=X_SCALE)
        emb = ws.f32("emb", M, PK)
        self.gemm(xt, self.emb_gen, M, out_f32=emb, alpha=1.0 / X_SCALE)
        # :137 init_handcrafted: FFT high-pass -> patch conv
        hp = self.highpass(inp)
        if taps is not None:
            taps["highpass"] = hp.clone()
        hip.patchify(hp, None, g.patch_size, a0, KP)
        hc = ws.f32("hc", M, PK)
        self.gemm(a0, self.hc_proj, M, out_f32=hc)
        feat = ws.h2("feat", M, PK)
        hip.add_rows(hc, emb, M, M, PK, out_h2=feat)
        # :140 pos embed
        hip.add_rows(x, self.pos, T, M, D, out_f32=x)
        xn = ws.h2("xn", M, D)
        qkv = ws.h2("qkv", M, 3 * D)
        att = ws.h2("att", M, D)
        prm = ws.h2("prm", M, PK)
        HK = g.mlp_dim + PK                                        # hidden row: [GELU(lin1) | prm of the NEXT block]
        hid = ws.h2("hid", M, HK)
        hid_prm = H2(hid.t[:, :, g.mlp_dim:])                      # the PK trailing columns (same row pitch)
        fold = taps is None                                        # block taps need x before the next prompt is added
        sq, sp = self.attn_split(M)
        if fold and self.ln_fold and not self.fold_disabled:
            return self._blocks_folded(x, feat, prm, qkv, att, hid, hid_prm, B)
        for i, blk in enumerate(self.blocks):
            # :138/:145 prompt_i = shared_mlp(GELU(lightweight_mlp_i(feat))) ; x = prompt_i + x
            if i == 0 or not fold:
                self.gemm(feat, self.light[i], M, out_h2=prm, act=ACT_GELU)
                self.gemm(prm, self.shared, M, residual=x, out_f32=x)
            # :430-446 block
            hip.layernorm(x, blk["n1w"], blk["n1b"], 1e-6, M, D, out_h2=xn)
            # qkv is stored head-major [3][B][H][T][hd]: every (image, head) K / V matrix is contiguous, so
            # the attention kernels stream whole cache lines instead of 160-byte slices of 7.7 KB token rows
            self.gemm(xn, blk["qkv"], M, out_h2=qkv, head_major=(T, g.num_heads, g.head_dim))
            if blk["window"] > 0:
                self.attention(qkv, att, B, T, g.num_heads, g.head_dim, mode=2, grid=G, window=blk["window"],
                               pad=blk["pad"], rel_h=blk["rel_h"], rel_w=blk["rel_w"], split_qk=sq, split_pv=sp,
                               head_major=True, scale=1.0)
            else:
                self.attention(qkv, att, B, T, g.num_heads, g.head_dim, mode=1, grid=G, rel_h=blk["rel_h"],
                               rel_w=blk["rel_w"], split_qk=sq, split_pv=sp, head_major=True, scale=1.0)
            self.gemm(att, blk["proj"], M, residual=x, out_f32=x)
            hip.layernorm(x, blk["n2w"], blk["n2b"], 1e-6, M, D, out_h2=xn)
            self.gemm(xn, blk["lin1"], M, out_h2=hid, ldoh=HK, act=ACT_GELU, out_scale=HID_SCALE)
            if fold and i + 1 < g.depth:
                self.gemm(feat, self.light[i + 1], M, out_h2=hid_prm, ldoh=HK, act=ACT_GELU, out_scale=HID_SCALE)
                self.gemm(hid, self.lin2cat[i], M, residual=x, out_f32=x, alpha=1.0 / HID_SCALE)
            else:
                self.gemm(hid, blk["lin2"], M, lda=HK, residual=x, out_f32=x, alpha=1.0 / HID_SCALE)
            if taps is not None:
                taps[f"block{i}"] = x.clone()
            self._hook(i)
        hip.add_rows(x, None, 1, M, D, scale=X_SCALE, out_h2=xn)
        return self._neck(xn, B)

    def attn_split(self, M: int):
        """Terms of the attention products for a forward over M token rows: precision `mx` spends its error budget where it buys time --
        batches; one image per call (M <= 4096) keeps three terms, like its GEMMs keep split-3 operands (_blocks_folded)."""
        pr = self.prec
        return (3, 3) if (pr.mx and M <= 4096) else (pr.qk, pr.pv)

    def _blocks_folded(self, x, feat, prm, qkv, att, hid, hid_prm, B: int) -> torch.Tensor:
        """The 32 blocks without LayerNorm passes.  The residual stream lives in h2 (xh = x * X_SCALE); proj and lin2
        read it as their residual, write it back and leave the row sums (sum x, sum x^2) behind; qkv and lin1 consume
        the un-normalised rows with the norm folded into their epilogue (LnLinear).  image_encoder.py:430-446."""
        g, ws, pr = self.g, self.ws, self.prec
        G, D, T = g.grid, g.embed_dim, g.grid * g.grid
        M, HK = B * T, g.mlp_dim + self.PK
        xh = ws.h2("xh", M, D)
        # Row statistics travel as per-64-column pieces written with plain stores (include/cvlm.h, ABI 5): nothing to zero
        # between launches, no atomics -- the folded LayerNorm is bit-reproducible from run to run.  A 3-us kernel merges the
        # pieces of a row into the (rstd, mu * rstd) pair the consuming GEMM's epilogue reads.
        pcs, mrg = ws.f32("ln_pieces", hip.stats_pieces(D), M, 2), ws.f32("ln_merged", M, 2)
        gws = self.ws.gemm_ws()
        self.gemm(feat, self.light[0], M, out_h2=prm, act=ACT_GELU)               # prompt of block 0 (:145)
        self.gemm(prm, self.shared, M, residual=x, out_f32=x)
        hip.row_stats_split(x, X_SCALE, xh, pcs, M, D)
        inv = 1.0 / X_SCALE
        # The two activations that only GEMMs touch -- the residual stream and the MLP hidden rows -- live in the 128-byte-row image
        # (cvlm_gemm a_il / out_il / res_il): the GEMMs that read them as their operand fetch whole L2 lines, as they do for the
        # weights.  Block 0 reads the planar seed that cvlm_row_stats_split wrote and its proj writes the image; the attention
        # output stays in planes (the attention kernels write it).
        use_il = ((M > 4096 or self.act_il_small) and self.act_il and pr.gemm == 3 and D % 32 == 0 and HK % 32 == 0 and self.neck0.w_il is not None and
                  all(b["qkv_f"].w_il is not None and b["lin1_f"].w_il is not None for b in self.blocks) and
                  self.blocks[-1]["lin2"].w_il is not None and all(l.w_il is not None for l in self.lin2cat))
        xo = xh                                                      # where proj / lin2 write the stream
        x_last = None
        if use_il:
            xo = ws.h2il("xh_il", M, D)
            hid = ws.h2il("hid_il", M, HK)
            hid_prm = hid.cols(g.mlp_dim)
        # Precision `mx` (include/cvlm.h ABI 10): the same two activations as mx operands -- fp16 hi values + e4m3 hi8 / lo8 bytes with
        # block exponents --, so that qkv, lin1 and lin2 (92 % of the blocks' GEMM flops) run hi.hi on fp16 and both correction
        # products on the block-scaled e4m3 instruction.  The stream keeps its fp16 lo plane beside the image: proj / lin2 read and
        # write it as their h2 residual at full 22 bits.  Block 0's qkv still reads the planar seed; the last lin2 writes the
        # 128-byte-row image the neck reads.
        # One image per call (M <= 4096, the reference's own call pattern) stays on split-3 operands: the mx kernel has only its 256-row
        # tile form, the small-grid forms of the split-3 launcher (column split, K-parts, deep rings) are as fast there (24.2 ms per image
        # either way, profiles/r05_per_shape_times_b1.log) and keep the 3e-5 of the `exact` arithmetic.
        use_mx = (use_il and pr.mx and M > 4096 and D % 64 == 0 and HK % 64 == 0 and self.blocks[-1]["lin2"].w_mx is not None and
                  all(b["qkv_f"].w_mx is not None and b["lin1_f"].w_mx is not None for b in self.blocks) and
                  all(l.w_mx is not None for l in self.lin2cat))
        if use_mx:
            x_last = xo
            xo = ws.h2mx("xh_mx", M, D, lo_plane=True)
            # (lin2 on split-3 operands -- hidden rows as the 128-byte-row image -- measured 0.7 % slower per step, mask 1.7e-4 instead of 2.1e-4)
            hid = ws.h2mx("hid_mx", M, HK)
            hid_prm = hid.cols(g.mlp_dim)
        hk = {} if use_il else {"ldoh": HK}                          # the image carries its own row stride
        sq, sp = self.attn_split(M)
        for i, blk in enumerate(self.blocks):
            hip.ln_stats_merge(pcs, M, D, 1e-6, mrg, gws)
            # split_qk == 1: the attention kernels read K's hi plane only -- the projection does not write the other (a sixth of its stores)
            self.gemm(xh, blk["qkv_f"], M, out_h2=qkv, head_major=(T, g.num_heads, g.head_dim), head_major_nolo=2 if sq == 1 else 0, alpha=inv,
                      ln_fold=(mrg, blk["qkv_f"].colsum))
            if blk["window"] > 0:
                self.attention(qkv, att, B, T, g.num_heads, g.head_dim, mode=2, grid=G, window=blk["window"],
                               pad=blk["pad"], rel_h=blk["rel_h"], rel_w=blk["rel_w"], split_qk=sq, split_pv=sp,
                               head_major=True, scale=1.0)
            else:
                self.attention(qkv, att, B, T, g.num_heads, g.head_dim, mode=1, grid=G, rel_h=blk["rel_h"],
                               rel_w=blk["rel_w"], split_qk=sq, split_pv=sp, head_major=True, scale=1.0)
            self.gemm(att, blk["proj"], M, out_h2=xo, residual_h2=(xh, inv), out_scale=X_SCALE, row_stats=pcs)
            xh = xo                                                  # from here on the stream is read where it was written
            hip.ln_stats_merge(pcs, M, D, 1e-6, mrg, gws)
            self.gemm(xh, blk["lin1_f"], M, out_h2=hid, act=ACT_GELU, out_scale=HID_SCALE, alpha=inv,
                      ln_fold=(mrg, blk["lin1_f"].colsum), **hk)
            if i + 1 < g.depth:
                self.gemm(feat, self.light[i + 1], M, out_h2=hid_prm, act=ACT_GELU, out_scale=HID_SCALE, **hk)
                self.gemm(hid, self.lin2cat[i], M, out_h2=xh, residual_h2=(xh, inv), out_scale=X_SCALE,
                          alpha=1.0 / HID_SCALE, row_stats=pcs)
            else:
                self.gemm(hid, blk["lin2"], M, lda=HK, out_h2=xh if x_last is None else x_last, residual_h2=(xh, inv), out_scale=X_SCALE,
                          alpha=1.0 / HID_SCALE)
                if x_last is not None:
                    xh = x_last
            self._hook(i)
        return self._neck(xh, B)

    def _neck(self, xn: H2, B: int) -> torch.Tensor:
        """:150 neck (LayerNorm2d == row LN on NHWC); xn = x * X_SCALE in h2."""
        g, ws = self.g, self.ws
        G, D, T = g.grid, g.embed_dim, g.grid * g.grid
        M = B * T
        C = g.out_chans
        c1 = ws.f32("neck_c1", M, C)
        self.gemm(xn, self.neck0, M, out_f32=c1, alpha=1.0 / X_SCALE)
        feats = ws.f32(getattr(self, "_out_name", "features"), M, C)
        if implicit_conv_ok(C):                                      # 3x3 as an implicit GEMM: nothing materialised
            c1h = ws.h2("neck_c1h", M, C)
            hip.layernorm(c1, self.nk1[0], self.nk1[1], 1e-6, M, C, out_h2=c1h)
            self.gemm(c1h, self.neck2, M, out_f32=feats, conv3x3=(G, G, C))
        else:
            hip.layernorm(c1, self.nk1[0], self.nk1[1], 1e-6, M, C, out_f32=c1)
            col = ws.h2("neck_col", M, 9 * C)
            hip.im2col3x3(c1, B, G, G, C, col)
            self.gemm(col, self.neck2, M, out_f32=feats)
        hip.layernorm(feats, self.nk3[0], self.nk3[1], 1e-6, M, C, out_f32=feats)
        return feats


# ================================================================================================
# Edge mask decoder  (models/mmseg/models/sam/mask_decoder_edge.py, transformer_maskdecoder_edge.py)
# ================================================================================================
class MaskDecoder(_Base):
    def __init__(self, sd: Dict[str, torch.Tensor], g: SamGeometry, device, precision: Precision,
                 prefix: str = "mask_decoder."):
        super().__init__(device, precision)
        self.g = g
        P = prefix
        C = g.prompt_embed_dim
        self.lin: Dict[str, Linear] = {}
        self.ln: Dict[str, tuple] = {}
        names = {k[len(P):] for k in sd if k.startswith(P)}
        for name in names:
            if not name.endswith(".weight") or name in ("iou_token.weight", "mask_tokens.weight", "edge_token.weight"):
                continue
            stem, t = name[:-7], sd[P + name]
            if t.dim() == 2 and stem + ".bias" in names:            # nn.Linear
                self.lin[stem] = Linear(t, sd[P + stem + ".bias"], device)
            elif t.dim() == 1:                                       # LayerNorm / LayerNorm2d
                self.ln[stem] = (self.dev(t), self.dev(sd[P + stem + ".bias"]))
        self.tokens = self.dev(torch.cat([sd[P + "iou_token.weight"], sd[P + "mask_tokens.weight"],
                                          sd[P + "edge_token.weight"]], 0))                      # (6, C)

        def convT2(name):                       # ConvTranspose2d(k2,s2): weight (in, out, 2, 2) -> rows (dy, dx, co)
            w = sd[P + name + ".weight"].detach().float().cpu()
            return Linear(w.permute(2, 3, 1, 0).reshape(-1, w.shape[0]),
                          sd[P + name + ".bias"].detach().float().cpu().repeat(4), device)

        def convT3(name):                       # ConvTranspose2d(3,1,1) == conv3x3 with flipped taps
            w = sd[P + name + ".weight"].detach().float().cpu()          # (in, out, 3, 3)
            wf = torch.flip(w, dims=(2, 3)).permute(1, 2, 3, 0)           # (out, ky, kx, in)
            return Linear(wf.reshape(wf.shape[0], -1), sd[P + name + ".bias"], device)

        self.up = {n: (convT2(n + ".0"), convT2(n + ".3")) for n in ("output_upscaling", "embedding_encoder")}
        self.mf = (convT3("embedding_maskfeature.0"), convT3("embedding_maskfeature.3"))
        self.pe: Optional[torch.Tensor] = None
        self._merged(sd, P)

    def _upscale(self, x_h2: H2, B: int, G: int, name: str, final_gelu: bool, out: torch.Tensor,
                 out_h2: Optional[H2] = None) -> torch.Tensor:
        """mask_decoder_edge.py:53-59 / 82-87 on token-major input [B*G*G][C]; out_h2: the same values as h2 planes too."""
        C, ws = self.g.prompt_embed_dim, self.ws
        l0, l3 = self.up[name]
        u1 = ws.f32(name + "_u1", B * 4 * G * G, C // 4)
        self.gemm(x_h2, l0, B * G * G, out_f32=u1, pixel_shuffle=(G, G, 2 * (C // 4)))
        u1h = ws.h2(name + "_u1h", B * 4 * G * G, C // 4)
        w, b = self.ln[name + ".1"]
        hip.layernorm(u1, w, b, 1e-6, B * 4 * G * G, C // 4, act=ACT_GELU, out_h2=u1h)
        kw = dict(out_h2=out_h2) if out_h2 is not None else {}
        self.gemm(u1h, l3, B * 4 * G * G, out_f32=out, pixel_shuffle=(2 * G, 2 * G, 2 * (C // 8)),
                  act=ACT_GELU if final_gelu else ACT_NONE, **kw)
        return out

    # ---- merged projections and image-independent terms (round 4: 133 -> ~75 launches per forward) ---------------------------------
    # Every attention of the two-way transformer projects `x + pe` for q / k and `x` for v (transformer_maskdecoder_edge.py:174-212):
    #   (x + pe) W^T + b = x W^T + b + (pe W^T),
    # and pe (the dense positional encoding for image rows, the token embeddings for token rows) does not depend on the image: the
    # products pe W^T are constants of the model, added to the GEMM as its f32 residual.  So the projections read the stream itself
    # (the h2 planes its LayerNorm writes beside the f32 rows: no add / split launches), q | k | v that share an input run as ONE GEMM
    # over the concatenated weights, the four condition attentions' k / v projections (inputs 2 cond and cond, :98-99) as two GEMMs
    # for both layers, the attention writes h2 planes for out_proj, out_proj adds the stream as its residual, and layer 0's self
    # attention (tokens only, :174-176) is computed once.
    def _merged(self, sd, P: str) -> None:
        g = self.g
        W = lambda n: sd[P + n + ".weight"].detach().float().cpu()
        Bv = lambda n: sd[P + n + ".bias"].detach().float().cpu()
        cat = lambda names: Linear(torch.cat([W(n) for n in names], 0), torch.cat([Bv(n) for n in names], 0), self.device)
        self.m: Dict[str, Linear] = {}
        cond_names = []
        for i in range(g.dec_depth):
            L = f"transformer.layers.{i}."
            self.m[L + "self_qkv"] = cat([L + "self_attn." + x for x in ("q_proj", "k_proj", "v_proj")])
            self.m[L + "t2i_kv"] = cat([L + "cross_attn_token_to_image." + x for x in ("k_proj", "v_proj")])
            self.m[L + "i2t_kv"] = cat([L + "cross_attn_image_to_token." + x for x in ("k_proj", "v_proj")])
            cond_names += [L + "cross_attn_token_to_cond", L + "cross_attn_image_to_cond"]
        self.m["fin_kv"] = cat(["transformer.final_attn_token_to_image." + x for x in ("k_proj", "v_proj")])
        self.m["cond_k"] = cat([n + ".k_proj" for n in cond_names])
        self.m["cond_v"] = cat([n + ".v_proj" for n in cond_names])
        self.cond_slot = {n: j for j, n in enumerate(cond_names)}
        self.consts: Optional[Dict[str, object]] = None

    def _build_consts(self, gauss: torch.Tensor) -> None:
        """pe W^T for every projection that the reference feeds `x + pe`, and layer 0's queries (image independent)."""
        g, dev = self.g, self.device
        G, C, T, NT = g.grid, g.prompt_embed_dim, g.grid * g.grid, self.tokens.shape[0]
        self.pe = torch.empty(T, C, device=dev)
        hip.dense_pe(gauss, G, C, self.pe)
        peh, tokh = H2.empty(T, C, device=dev), H2.empty(NT, C, device=dev)
        hip.split_f32(self.pe, peh)
        hip.split_f32(self.tokens, tokh)
        c: Dict[str, object] = {}

        def proj(a_h2, rows, name, width, col0=0, out=None):          # out[:, col0:col0 + N] = a . W^T (no bias); other columns stay 0
            lin = self.lin[name]
            out = torch.zeros(rows, width, device=dev) if out is None else out
            self.gemm(a_h2, lin, rows, out_f32=out[:, col0:], ldo=width, bias=None)
            return out
        for i in range(g.dec_depth):
            L = f"transformer.layers.{i}."
            I = self.lin[L + "cross_attn_token_to_image.q_proj"].N
            c[L + "t2i_q"] = proj(tokh, NT, L + "cross_attn_token_to_image.q_proj", I)
            c[L + "t2i_kv"] = proj(peh, T, L + "cross_attn_token_to_image.k_proj", 2 * I)
            c[L + "t2c_q"] = proj(tokh, NT, L + "cross_attn_token_to_cond.q_proj", I)
            c[L + "i2c_q"] = proj(peh, T, L + "cross_attn_image_to_cond.q_proj", I)
            c[L + "i2t_q"] = proj(peh, T, L + "cross_attn_image_to_token.q_proj", I)
            c[L + "i2t_kv"] = proj(tokh, NT, L + "cross_attn_image_to_token.k_proj", 2 * I)
            if i > 0:
                sq = proj(tokh, NT, L + "self_attn.q_proj", 3 * C)
                c[L + "self_qkv"] = proj(tokh, NT, L + "self_attn.k_proj", 3 * C, col0=C, out=sq)
        I = self.lin["transformer.final_attn_token_to_image.q_proj"].N
        c["fin_q"] = proj(tokh, NT, "transformer.final_attn_token_to_image.q_proj", I)
        c["fin_kv"] = proj(peh, T, "transformer.final_attn_token_to_image.k_proj", 2 * I)
        # layer 0: queries = LN(self_attn(tokens, tokens, tokens)) -- "replaces instead of adds" (:174-176)
        L = "transformer.layers.0."
        qkv = torch.empty(NT, 3 * C, device=dev)
        self.gemm(tokh, self.m[L + "self_qkv"], NT, out_f32=qkv)
        oh = H2.empty(NT, C, device=dev)
        hip.small_attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], None, 1, NT, NT, g.dec_heads, C // g.dec_heads, out_h2=oh)
        ao = torch.empty(NT, C, device=dev)
        self.gemm(oh, self.lin[L + "self_attn.out_proj"], NT, out_f32=ao)
        q0 = torch.empty(NT, C, device=dev)
        hip.layernorm(ao, *self.ln[L + "norm1"], 1e-5, NT, C, out_f32=q0)
        c["q0"] = q0
        self.consts = c

    def _bgemm(self, a: H2, lin: Linear, B: int, rows: int, out: torch.Tensor, const: torch.Tensor, **kw) -> None:
        """out[b] = a[b] . W^T + bias + const for B images of `rows` rows each; const f32 [rows][N] is shared by the images
        (one problem per image with residual stride 0 when B > 1)."""
        if B == 1:
            self.gemm(a, lin, rows, out_f32=out, residual=const, **kw)
        else:
            self.gemm(a, lin, rows, out_f32=out, residual=const, batch=B, stride_a=rows * lin.K, stride_r=0, stride_o=rows * lin.N, **kw)

    def forward(self, feats: torch.Tensor, sparse: torch.Tensor, no_mask: torch.Tensor, gauss: torch.Tensor,
                B: int, taps: Optional[dict] = None) -> torch.Tensor:
        """feats f32 [B*T][C]; sparse f32 [B][2][C] -> low-res mask logits f32 [B][4G][4G] (mask 0, :133-135).
        mask_decoder_edge.py:96-190, transformer_maskdecoder_edge.py:62-214."""
        g, ws = self.g, self.ws
        G, C, T, H = g.grid, g.prompt_embed_dim, g.grid * g.grid, g.dec_heads
        NT = self.tokens.shape[0]
        if self.consts is None:
            self._build_consts(gauss)
        cst = self.consts
        fh = ws.h2("feats_h", B * T, C)
        hip.split_f32(feats, fh)
        edge_feat = self._upscale(fh, B, G, "embedding_encoder", False, ws.f32("edge_feat", B * 16 * T, C // 8))
        # :150-158 tokens / src; every stream lives as f32 rows + the h2 planes of the same values
        keys, keys_h = ws.f32("keys", B * T, C), ws.h2("d_keys_h", B * T, C)
        hip.add_rows(feats, no_mask, 1, B * T, C, out_f32=keys, out_h2=keys_h)
        queries, queries_h = ws.f32("queries", B * NT, C), ws.h2("d_queries_h", B * NT, C)
        zeros = ws._get("f32", "d_zeros", B * NT * C, torch.float32, True).view(B * NT, C)    # zero-filled once, never written
        hip.add_rows(zeros, cst["q0"], NT, B * NT, C, out_f32=queries, out_h2=queries_h)        # layer 0's queries for every image
        # condition tokens: k = (cond + cond_pe) W_k = 2 cond W_k (:98-99), v = cond W_v, for the four condition attentions at once
        cond_h = ws.h2("cond_h", B * 2, C)
        hip.split_f32(sparse, cond_h)
        NC = self.m["cond_k"].N
        ck, cv = ws.f32("cond_kp", B * 2, NC), ws.f32("cond_vp", B * 2, NC)
        self.gemm(cond_h, self.m["cond_k"], B * 2, out_f32=ck, alpha=2.0)
        self.gemm(cond_h, self.m["cond_v"], B * 2, out_f32=cv)
        ao_q, ao_k = ws.f32("ao_q", B * NT, C), ws.f32("ao_k", B * T, C)
        hidh = ws.h2("d_hid", B * NT, g.dec_mlp)

        def token_attn(name: str, qp, k, v, nk: int, norm) -> None:
            """softmax(q k^T) v -> out_proj + queries -> LayerNorm -> queries (f32 + h2); q / k / v already projected"""
            I = self.lin[name + ".out_proj"].K
            oh = ws.h2("t_o%d" % I, B * NT, I)
            hip.small_attention(qp, k, v, None, B, NT, nk, H, I // H, out_h2=oh)
            self.gemm(oh, self.lin[name + ".out_proj"], B * NT, residual=queries, out_f32=ao_q)
            hip.layernorm(ao_q, *norm, 1e-5, B * NT, C, out_f32=queries, out_h2=queries_h)

        def image_attn(name: str, qconst, k, v, nk: int, norm) -> None:
            I = self.lin[name + ".out_proj"].K
            iq, ioh = ws.f32("i_q", B * T, I), ws.h2("i_o", B * T, I)
            self._bgemm(keys_h, self.lin[name + ".q_proj"], B, T, iq, qconst)
            hip.small_attention(iq, k, v, None, B, T, nk, H, I // H, out_h2=ioh)
            self.gemm(ioh, self.lin[name + ".out_proj"], B * T, residual=keys, out_f32=ao_k)
            hip.layernorm(ao_k, *norm, 1e-5, B * T, C, out_f32=keys, out_h2=keys_h)

        def tokens_to_image(name: str, qconst, kvlin: Linear, kvconst, norm) -> None:
            I = self.lin[name + ".q_proj"].N
            qp, kv = ws.f32("t_q", B * NT, I), ws.f32("i_kv", B * T, 2 * I)
            self._bgemm(queries_h, self.lin[name + ".q_proj"], B, NT, qp, qconst)
            self._bgemm(keys_h, kvlin, B, T, kv, kvconst)
            token_attn(name, qp, kv[:, :I], kv[:, I:], T, norm)

        for i in range(g.dec_depth):
            L = f"transformer.layers.{i}."
            ln = lambda n: self.ln[L + n]
            if i > 0:                                                    # self attention (:177-180); layer 0's is in cst["q0"]
                qkv = ws.f32("sa_qkv", B * NT, 3 * C)
                self._bgemm(queries_h, self.m[L + "self_qkv"], B, NT, qkv, cst[L + "self_qkv"])
                token_attn(L + "self_attn", qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], NT, ln("norm1"))
            # tokens -> image (:183-187)
            tokens_to_image(L + "cross_attn_token_to_image", cst[L + "t2i_q"], self.m[L + "t2i_kv"], cst[L + "t2i_kv"], ln("norm2"))
            # tokens -> cond (:189-193)
            name = L + "cross_attn_token_to_cond"
            I = self.lin[name + ".q_proj"].N
            qp = ws.f32("t_q", B * NT, I)
            self._bgemm(queries_h, self.lin[name + ".q_proj"], B, NT, qp, cst[L + "t2c_q"])
            j = self.cond_slot[name]
            token_attn(name, qp, ck[:, j * I:(j + 1) * I], cv[:, j * I:(j + 1) * I], 2, ln("norm2_cond"))
            # MLP (:196-198)
            self.gemm(queries_h, self.lin[L + "mlp.lin1"], B * NT, out_h2=hidh, act=ACT_RELU)
            self.gemm(hidh, self.lin[L + "mlp.lin2"], B * NT, residual=queries, out_f32=ao_q)
            hip.layernorm(ao_q, *ln("norm3"), 1e-5, B * NT, C, out_f32=queries, out_h2=queries_h)
            # image -> cond (:201-205)
            name = L + "cross_attn_image_to_cond"
            j = self.cond_slot[name]
            image_attn(name, cst[L + "i2c_q"], ck[:, j * I:(j + 1) * I], cv[:, j * I:(j + 1) * I], 2, ln("norm4_cond"))
            # image -> tokens (:208-212): k = queries + token pe, v = queries
            name = L + "cross_attn_image_to_token"
            tkv = ws.f32("t_kv", B * NT, 2 * I)
            self._bgemm(queries_h, self.m[L + "i2t_kv"], B, NT, tkv, cst[L + "i2t_kv"])
            image_attn(name, cst[L + "i2t_q"], tkv[:, :I], tkv[:, I:], NT, ln("norm4"))
        # final token -> image attention (:103-107); hs = the LayerNorm's output, f32 + h2
        tokens_to_image("transformer.final_attn_token_to_image", cst["fin_q"], self.m["fin_kv"], cst["fin_kv"],
                        self.ln["transformer.norm_final_attn"])
        hs, hs_h = queries, queries_h
        # :167-170 upscaling + edge feature head
        HW = 16 * T
        m1 = ws.f32("mf_1", B * HW, C // 4)
        edge_emb = ws.f32("edge_emb", B * HW, C // 8)
        if implicit_conv_ok(C // 8):                                 # both 3x3 convolutions as implicit GEMMs
            up_h = ws.h2("upscaled_h", B * HW, C // 8)
            up = self._upscale(keys_h, B, G, "output_upscaling", True, ws.f32("upscaled", B * HW, C // 8), out_h2=up_h)
            self.gemm(up_h, self.mf[0], B * HW, out_f32=m1, conv3x3=(4 * G, 4 * G, C // 8))
            m1h = ws.h2("mf_1h", B * HW, C // 4)
            hip.layernorm(m1, *self.ln["embedding_maskfeature.1"], 1e-6, B * HW, C // 4, act=ACT_GELU, out_h2=m1h)
            self.gemm(m1h, self.mf[1], B * HW, residual=edge_feat, out_f32=edge_emb, conv3x3=(4 * G, 4 * G, C // 4))
        else:
            up = self._upscale(keys_h, B, G, "output_upscaling", True, ws.f32("upscaled", B * HW, C // 8))
            col1 = ws.h2("mf_col1", B * HW, 9 * (C // 8))
            hip.im2col3x3(up, B, 4 * G, 4 * G, C // 8, col1)
            self.gemm(col1, self.mf[0], B * HW, out_f32=m1)
            hip.layernorm(m1, *self.ln["embedding_maskfeature.1"], 1e-6, B * HW, C // 4, act=ACT_GELU, out_f32=m1)
            col2 = ws.h2("mf_col2", B * HW, 9 * (C // 4))
            hip.im2col3x3(m1, B, 4 * G, 4 * G, C // 4, col2)
            self.gemm(col2, self.mf[1], B * HW, residual=edge_feat, out_f32=edge_emb)
        # :172-186 hyper-network rows actually used: mask token 0 (hs row 1) and edge token (hs row 5), read in place from the h2
        # planes of hs (row b * NT + token: pointer offset + row pitch NT * C)
        hyper = ws.f32("hyper", B, 5, C // 8)
        t1, t2 = ws.h2("h_t1", B, C), ws.h2("h_t2", B, C)
        for tok_row, mlp, slot in ((1, "output_hypernetworks_mlps.0", 0), (NT - 1, "edge_mlp", 4)):
            rowh = H2(hs_h.t[:, tok_row:])
            self.gemm(rowh, self.lin[mlp + ".layers.0"], B, lda=NT * C, out_h2=t1, act=ACT_RELU)
            self.gemm(t1, self.lin[mlp + ".layers.1"], B, out_h2=t2, act=ACT_RELU)
            self.gemm(t2, self.lin[mlp + ".layers.2"], B, out_f32=hyper[:, slot], ldo=5 * (C // 8))
        low = ws.f32("low", B, HW)
        hip.mask_head(up, edge_emb, hyper, B, HW, C // 8, low)
        if taps is not None:
            taps.update(hs=hs.clone(), src=keys.clone(), upscaled=up.clone(), edge_emb=edge_emb.clone(),
                        hyper=hyper.clone(), low_res_masks=low.clone())
        return low


class VanillaMaskDecoder(MaskDecoder):
    """Vanilla SAM MaskDecoder of the registry entry ``sam`` (models/sam.py:322-333; mask_decoder.py:73-150,
    transformer.py:62-107,151-183): 1 IoU + 4 mask tokens, no prompts, no condition attentions, no edge branch;
    `multimask_output=False` keeps mask 0.  Same kernels as the edge decoder."""

    NT = 5

    def __init__(self, sd: Dict[str, torch.Tensor], g: SamGeometry, device, precision: Precision,
                 prefix: str = "mask_decoder."):
        _Base.__init__(self, device, precision)
        self.g = g
        P = prefix
        self.lin, self.ln = {}, {}
        names = {k[len(P):] for k in sd if k.startswith(P)}
        for name in names:
            if not name.endswith(".weight") or name in ("iou_token.weight", "mask_tokens.weight"):
                continue
            stem, t = name[:-7], sd[P + name]
            if t.dim() == 2 and stem + ".bias" in names:
                self.lin[stem] = Linear(t, sd[P + stem + ".bias"], device)
            elif t.dim() == 1:
                self.ln[stem] = (self.dev(t), self.dev(sd[P + stem + ".bias"]))
        self.tokens = self.dev(torch.cat([sd[P + "iou_token.weight"], sd[P + "mask_tokens.weight"]], 0))   # (5, C)

        def convT2(name):
            w = sd[P + name + ".weight"].detach().float().cpu()
            return Linear(w.permute(2, 3, 1, 0).reshape(-1, w.shape[0]),
                          sd[P + name + ".bias"].detach().float().cpu().repeat(4), device)

        self.up = {"output_upscaling": (convT2("output_upscaling.0"), convT2("output_upscaling.3"))}
        self.pe = None
        self.heads = 8                                               # models/sam.py:328

    def _attn(self, name, q, k, v, B, nq, nk, out):
        ws, heads = self.ws, self.heads
        I = self.lin[name + ".q_proj"].N
        qp, kp, vp = ws.f32("a_q", B * nq, I), ws.f32("a_k", B * nk, I), ws.f32("a_v", B * nk, I)
        self.gemm(q, self.lin[name + ".q_proj"], B * nq, out_f32=qp)
        self.gemm(k, self.lin[name + ".k_proj"], B * nk, out_f32=kp)
        self.gemm(v, self.lin[name + ".v_proj"], B * nk, out_f32=vp)
        o = ws.f32("a_o", B * nq, I)
        hip.small_attention(qp, kp, vp, o, B, nq, nk, heads, I // heads)
        oh = ws.h2("a_oh", B * nq, I)
        hip.split_f32(o, oh)
        self.gemm(oh, self.lin[name + ".out_proj"], B * nq, out_f32=out)

    def forward(self, feats: torch.Tensor, no_mask: torch.Tensor, gauss: torch.Tensor, B: int,
                taps: Optional[dict] = None) -> torch.Tensor:
        """feats f32 [B*T][C] -> low-res mask logits f32 [B][4G][4G] (mask 0)."""
        g, ws, NT = self.g, self.ws, self.NT
        G, C, T = g.grid, g.prompt_embed_dim, g.grid * g.grid
        if self.pe is None:
            self.pe = torch.empty(T, C, device=self.device)
            hip.dense_pe(gauss, G, C, self.pe)
        queries = ws.f32("vqueries", B * NT, C)
        queries.view(B, NT, C).copy_(self.tokens)
        keys = ws.f32("vkeys", B * T, C)
        hip.add_rows(feats, no_mask, 1, B * T, C, out_f32=keys)          # src = image_embeddings + dense (mask_decoder.py:126)
        qh, kh, vh = ws.h2("vd_q", B * NT, C), ws.h2("vd_k", B * T, C), ws.h2("vd_v", B * T, C)
        tq, tk = ws.h2("vd_tq", B * NT, C), ws.h2("vd_tk", B * NT, C)
        ao_q, ao_k = ws.f32("vao_q", B * NT, C), ws.f32("vao_k", B * T, C)
        hidh = ws.h2("vd_hid", B * NT, 2048)
        mo = ws.f32("vd_mlp", B * NT, C)
        for i in range(2):
            L = f"transformer.layers.{i}."
            ln = lambda n: self.ln[L + n]
            if i == 0:                                                    # transformer.py:157-158
                hip.split_f32(queries, tq)
                self._attn(L + "self_attn", tq, tq, tq, B, NT, NT, ao_q)
                hip.layernorm(ao_q, *ln("norm1"), 1e-5, B * NT, C, out_f32=queries)
            else:
                hip.add_rows(queries, self.tokens, NT, B * NT, C, out_h2=tq)
                hip.split_f32(queries, tk)
                self._attn(L + "self_attn", tq, tq, tk, B, NT, NT, ao_q)
                hip.layernorm(queries, *ln("norm1"), 1e-5, B * NT, C, add=ao_q, add_rows=B * NT, out_f32=queries)
            hip.add_rows(queries, self.tokens, NT, B * NT, C, out_h2=qh)   # :166-170
            hip.add_rows(keys, self.pe, T, B * T, C, out_h2=kh)
            hip.split_f32(keys, vh)
            self._attn(L + "cross_attn_token_to_image", qh, kh, vh, B, NT, T, ao_q)
            hip.layernorm(queries, *ln("norm2"), 1e-5, B * NT, C, add=ao_q, add_rows=B * NT, out_f32=queries, out_h2=qh)
            self.gemm(qh, self.lin[L + "mlp.lin1"], B * NT, out_h2=hidh, act=ACT_RELU)   # :173-175
            self.gemm(hidh, self.lin[L + "mlp.lin2"], B * NT, out_f32=mo)
            hip.layernorm(queries, *ln("norm3"), 1e-5, B * NT, C, add=mo, add_rows=B * NT, out_f32=queries)
            hip.add_rows(queries, self.tokens, NT, B * NT, C, out_h2=qh)   # :178-182
            hip.split_f32(queries, tq)
            self._attn(L + "cross_attn_image_to_token", kh, qh, tq, B, T, NT, ao_k)
            hip.layernorm(keys, *ln("norm4"), 1e-5, B * T, C, add=ao_k, add_rows=B * T, out_f32=keys)
        hip.add_rows(queries, self.tokens, NT, B * NT, C, out_h2=qh)       # :99-105
        hip.add_rows(keys, self.pe, T, B * T, C, out_h2=kh)
        hip.split_f32(keys, vh)
        self._attn("transformer.final_attn_token_to_image", qh, kh, vh, B, NT, T, ao_q)
        hs = ws.f32("vhs", B * NT, C)
        hip.layernorm(queries, *self.ln["transformer.norm_final_attn"], 1e-5, B * NT, C, add=ao_q, add_rows=B * NT,
                      out_f32=hs)
        up = self._upscale(vh, B, G, "output_upscaling", True, ws.f32("vupscaled", B * 16 * T, C // 8))
        HW = 16 * T
        hyper = ws.f32("vhyper", B, 5, C // 8)                              # row 0 = mask token 0's hypernetwork output
        row, rowh = ws.f32("vh_row", B, C), ws.h2("vh_rowh", B, C)
        t1, t2 = ws.h2("vh_t1", B, C), ws.h2("vh_t2", B, C)
        hip.gather_rows(hs, B, NT, C, None, 1, row)
        hip.split_f32(row, rowh)
        mlp = "output_hypernetworks_mlps.0"
        self.gemm(rowh, self.lin[mlp + ".layers.0"], B, out_h2=t1, act=ACT_RELU)
        self.gemm(t1, self.lin[mlp + ".layers.1"], B, out_h2=t2, act=ACT_RELU)
        self.gemm(t2, self.lin[mlp + ".layers.2"], B, out_f32=hyper[:, 0], ldo=5 * (C // 8))
        low = ws.f32("vlow", B, HW)
        hip.mask_head(up, None, hyper, B, HW, C // 8, low)                 # plain hyper . upscaled (mask_decoder.py:139)
        if taps is not None:
            taps.update(hs=hs.clone(), low_res_masks=low.clone())
        return low


class SamPlain(_Base):
    """Registry entry ``sam`` (models/sam.py:417-440 `infer`): encoder -> vanilla decoder -> bilinear to inp_size."""

    def __init__(self, sd: Dict[str, torch.Tensor], g: SamGeometry, device, precision: Precision = Precision()):
        super().__init__(device, precision)
        self.g = g
        self.encoder = SamEncoder(sd, g, device, precision)
        self.decoder = VanillaMaskDecoder(sd, g, device, precision)
        self.no_mask = self.dev(sd["no_mask_embed.weight"].reshape(1, -1))
        self.gauss = self.dev(sd["pe_layer.positional_encoding_gaussian_matrix"])

    def infer(self, inp: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
        g, B = self.g, inp.shape[0]
        feats = self.encoder.forward(inp, taps)
        low = self.decoder.forward(feats, self.no_mask, self.gauss, B, taps)
        masks = torch.empty(B, 1, g.inp_size, g.inp_size, device=self.device)
        hip.bilinear(low, B, 4 * g.grid, 4 * g.grid, masks, g.inp_size, g.inp_size)
        return masks


# ================================================================================================
# MaPLe / Alpha-CLIP  (alpha_clip_rw/model.py:507-563, cocotrainers/mapleAlphaCLIP.py:55-78,210-294)
# ================================================================================================
class ClipModel(_Base):
    def __init__(self, sd: Dict[str, torch.Tensor], c: ClipGeometry, device, precision: Precision,
                 prefix: str = "clip_model."):
        super().__init__(device, precision)
        self.c = c
        self.sd_prefix = prefix
        P = prefix
        ie, te, pl = P + "image_encoder.", P + "text_encoder.", P + "prompt_learner."
        W = c.vision_width
        wcat = torch.cat([sd[ie + "conv1.weight"].detach().float().cpu().reshape(W, -1),
                          sd[ie + "conv1_alpha.weight"].detach().float().cpu().reshape(W, -1)], 1)
        self.conv = Linear(wcat, None, device)
        self.cls = self.dev(sd[ie + "class_embedding"])
        self.pos = self.dev(sd[ie + "positional_embedding"])
        self.ln_pre = (self.dev(sd[ie + "ln_pre.weight"]), self.dev(sd[ie + "ln_pre.bias"]))
        self.ln_post = (self.dev(sd[ie + "ln_post.weight"]), self.dev(sd[ie + "ln_post.bias"]))
        self.vproj = Linear(sd[ie + "proj"].detach().float().cpu().t().contiguous(), None, device)

        def block(p, text):
            ipw = sd[p + ("attn.in_proj_weight" if text else "attn.in_proj.weight")]
            ipb = sd[p + ("attn.in_proj_bias" if text else "attn.in_proj.bias")]
            return dict(inp=Linear(ipw, ipb, device),
                        out=Linear(sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"], device),
                        fc=Linear(sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"], device),
                        pj=Linear(sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"], device, mx=precision.mx and not text),
                        ln1=(self.dev(sd[p + "ln_1.weight"]), self.dev(sd[p + "ln_1.bias"])),
                        ln2=(self.dev(sd[p + "ln_2.weight"]), self.dev(sd[p + "ln_2.bias"])))

        self.vblocks = [block(f"{ie}transformer.resblocks.{i}.", False) for i in range(c.vision_layers)]
        # vision tower with ln_1 / ln_2 folded into in_proj / c_fc and the residual stream in h2 (as the SAM blocks, §4):
        # CVLM_LN_FOLD=0 keeps the separate LayerNorm passes (the text tower, run once, always does)
        self.ln_fold = os.environ.get("CVLM_LN_FOLD", "1") == "1" and c.vision_width % 8 == 0
        self.fold_disabled = False          # as SamEncoder.fold_disabled
        if self.ln_fold:
            for i, blk in enumerate(self.vblocks):
                p = f"{ie}transformer.resblocks.{i}."
                blk["inp_f"] = LnLinear(sd[p + "attn.in_proj.weight"], sd[p + "attn.in_proj.bias"],
                                        sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], device, mx=precision.mx)
                blk["fc_f"] = LnLinear(sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"],
                                       sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], device, mx=precision.mx)
        self.tblocks = [block(f"{te}transformer.resblocks.{i}.", True) for i in range(c.text_layers)]
        self.tpos = self.dev(sd[te + "positional_embedding"])
        self.ln_final = (self.dev(sd[te + "ln_final.weight"]), self.dev(sd[te + "ln_final.bias"]))
        self.tproj = Linear(sd[te + "text_projection"].detach().float().cpu().t().contiguous(), None, device)
        self.logit_scale_exp = float(torch.as_tensor(sd[P + "logit_scale"]).detach().float().exp())
        # MaPLe prompts (mapleAlphaCLIP.py:210-227): shared ctx and deep visual prompts are weight-only
        self.ctx = self.dev(sd[pl + "ctx"])
        self.deep_text = [self.dev(sd[f"{pl}compound_prompts_text.{i}"]) for i in range(c.prompt_depth - 1)]
        proj = Linear(sd[pl + "proj.weight"], sd[pl + "proj.bias"], device)
        self.shared_ctx = torch.empty(c.n_ctx, W, device=device)
        ch = H2.empty(c.n_ctx, c.text_width, device=device)
        hip.split_f32(self.ctx, ch)
        self.gemm(ch, proj, c.n_ctx, out_f32=self.shared_ctx)
        self.deep_vis = []
        for i in range(c.prompt_depth - 1):
            lin = Linear(sd[f"{pl}compound_prompt_projections.{i}.weight"],
                         sd[f"{pl}compound_prompt_projections.{i}.bias"], device)
            hip.split_f32(self.deep_text[i], ch)
            o = torch.empty(c.n_ctx, W, device=device)
            self.gemm(ch, lin, c.n_ctx, out_f32=o)
            self.deep_vis.append(o)
        self.prefix = {s: self.dev(sd[pl + "token_prefix" + ("_test" if s == "test" else "")]) for s in ("train", "test")}
        self.suffix = {s: self.dev(sd[pl + "token_suffix" + ("_test" if s == "test" else "")]) for s in ("train", "test")}
        self.txt: Dict[str, torch.Tensor] = {}

    def _blocks(self, x: torch.Tensor, blocks, Bn: int, L: int, Wd: int, heads: int, deep, first_row: int,
                causal: bool) -> None:
        """alpha_clip_rw/model.py:315-362 / 392-434: residual attention blocks on x f32 [Bn][L][Wd] (in place)."""
        ws, pr, c = self.ws, self.prec, self.c
        M = Bn * L
        tag = "t" if causal else "v"
        xn, qkv, att = ws.h2(tag + "xn", M, Wd), ws.h2(tag + "qkv", M, 3 * Wd), ws.h2(tag + "att", M, Wd)
        hid = ws.h2(tag + "hid", M, 4 * Wd)
        for i, blk in enumerate(blocks):
            if 1 <= i <= len(deep):
                hip.overwrite_rows(x, Bn, L, Wd, first_row, c.n_ctx, deep[i - 1])
            hip.layernorm(x, *blk["ln1"], 1e-5, M, Wd, out_h2=xn)
            self.gemm(xn, blk["inp"], M, out_h2=qkv)
            self.attention(qkv, att, Bn, L, heads, Wd // heads, mode=0, causal=causal, split_qk=pr.qk, split_pv=pr.pv)
            self.gemm(att, blk["out"], M, residual=x, out_f32=x)
            hip.layernorm(x, *blk["ln2"], 1e-5, M, Wd, out_h2=xn)
            self.gemm(xn, blk["fc"], M, out_h2=hid, act=ACT_QUICKGELU, out_scale=HID_SCALE)
            self.gemm(hid, blk["pj"], M, residual=x, out_f32=x, alpha=1.0 / HID_SCALE)

    def _vision_blocks_folded(self, x: torch.Tensor, Bn: int, L: int, Wd: int, heads: int, first_row: int) -> torch.Tensor:
        """The vision tower's blocks (alpha_clip_rw/model.py:392-434) on an h2 residual stream: no LayerNorm pass, no f32
        stream -- `out_proj` / `c_proj` add into the planes and leave the row sums the next folded GEMM normalises with;
        the deep prompts overwrite planes AND sums (cvlm_row_stats_split, copies = images).  -> class-token rows f32 [Bn][Wd]."""
        ws, pr, c = self.ws, self.prec, self.c
        M, inv = Bn * L, 1.0 / X_SCALE
        xh, qkv, att = ws.h2("vxh", M, Wd), ws.h2("vqkv", M, 3 * Wd), ws.h2("vatt", M, Wd)
        # the MLP hidden rows are written by c_fc and read by c_proj only: they travel as a 128-byte-row image (SamEncoder._blocks_folded)
        hid_il = (os.environ.get("CVLM_GEMM_AIL", "1") not in ("0", "b") or M > 4096) and os.environ.get("CVLM_GEMM_AIL", "1") != "0" \
            and pr.gemm == 3 and (4 * Wd) % 32 == 0 and all(b["pj"].w_il is not None and b["fc_f"].w_il is not None for b in self.vblocks)
        hid = ws.h2il("vhid_il", M, 4 * Wd) if hid_il else ws.h2("vhid", M, 4 * Wd)
        # precision `mx`: the hidden rows as an mx operand (SamEncoder._blocks_folded): c_proj, a third of the tower's GEMM flops, runs
        # its two correction products on the block-scaled e4m3 instruction
        if hid_il and pr.mx and (4 * Wd) % 64 == 0 and M % 8 == 0 and all(b["pj"].w_mx is not None for b in self.vblocks):
            hid = ws.h2mx("vhid_mx", M, 4 * Wd)
            # ... and the residual stream itself (image + block exponents + lo plane: cvlm_row_stats_split_mx seeds it and writes the
            # deep prompts' rows into it; out_proj / c_proj write it back): in_proj and c_fc, the other half of the tower's GEMM
            # flops, read mx operands too.  Needs the class-token tail (the last block reads the stream by its class rows).
            if Wd % 64 == 0 and self.class_token_tail and all(b["inp_f"].w_mx is not None and b["fc_f"].w_mx is not None for b in self.vblocks):
                xh = ws.h2mx("vxh_mx", M, Wd, lo_plane=True)
        pcs, mrg = ws.f32("vln_pieces", hip.stats_pieces(Wd), M, 2), ws.f32("vln_merged", M, 2)
        gws = self.ws.gemm_ws()
        hip.row_stats_split(x.view(M, Wd), X_SCALE, xh, pcs, M, Wd)
        last = len(self.vblocks) - 1
        for i, blk in enumerate(self.vblocks):
            if 1 <= i <= len(self.deep_vis):
                hip.row_stats_split(self.deep_vis[i - 1], X_SCALE, xh, pcs, c.n_ctx, Wd, row0=first_row, copies=Bn,
                                    dst_row_stride=L)
            hip.ln_stats_merge(pcs, M, Wd, 1e-5, mrg, gws)
            self.gemm(xh, blk["inp_f"], M, out_h2=qkv, alpha=inv, ln_fold=(mrg, blk["inp_f"].colsum))
            if i == last and self.class_token_tail:
                return self._class_token_tail(blk, xh, qkv, att, Bn, L, Wd, heads)
            self.attention(qkv, att, Bn, L, heads, Wd // heads, mode=0, causal=False, split_qk=pr.qk, split_pv=pr.pv)
            self.gemm(att, blk["out"], M, out_h2=xh, residual_h2=(xh, inv), out_scale=X_SCALE, row_stats=pcs)
            hip.ln_stats_merge(pcs, M, Wd, 1e-5, mrg, gws)
            self.gemm(xh, blk["fc_f"], M, out_h2=hid, act=ACT_QUICKGELU, out_scale=HID_SCALE, alpha=inv,
                      ln_fold=(mrg, blk["fc_f"].colsum))
            self.gemm(hid, blk["pj"], M, out_h2=xh, residual_h2=(xh, inv), out_scale=X_SCALE, alpha=1.0 / HID_SCALE,
                      row_stats=pcs)
        cls = ws.f32("ccls", Bn, Wd)                                             # class token = row 0 of every image
        hip.gather_rows_h2(xh, inv, Bn, L, Wd, None, 0, cls)
        return cls

    def _class_token_tail(self, blk, xh: H2, qkv: H2, att: H2, Bn: int, L: int, Wd: int, heads: int) -> torch.Tensor:
        """The LAST block of the vision tower behind its qkv projection, for the class tokens only.  Nothing but row 0 of every image
        leaves the tower (`ln_post(x[:, 0, :])`, alpha_clip_rw/model.py:558-561), so of this block only the class token's attention
        output (its query against all L keys), out_proj, LayerNorm, c_fc and c_proj are ever read: the attention is asked for the first
        query block alone (cvlm_attn_args.q_rows), the three GEMMs run on Bn rows picked out of the [Bn * L]-row planes by their row
        stride.  Same arithmetic per row; of the block's 0.63 ms at sixteen images 0.06 remain.  -> class-token rows f32 [Bn][Wd]."""
        ws, pr, inv = self.ws, self.prec, 1.0 / X_SCALE
        gws = self.ws.gemm_ws()
        self.attention(qkv, att, Bn, L, heads, Wd // heads, mode=0, causal=False, split_qk=pr.qk, split_pv=pr.pv, q_rows=1)
        xc, hidc = ws.h2("vx_cls", Bn, Wd), ws.h2("vhid_cls", Bn, 4 * Wd)
        pcs, mrg = ws.f32("vln_pieces_cls", hip.stats_pieces(Wd), Bn, 2), ws.f32("vln_merged_cls", Bn, 2)
        res = (xh.every(L), inv) if getattr(xh, "mx", False) else (xh, inv)       # the class rows of the stream (an mx stream: a strided view)
        self.gemm(att, blk["out"], Bn, lda=L * Wd, out_h2=xc, residual_h2=res, ldrh=L * Wd, out_scale=X_SCALE, row_stats=pcs)
        hip.ln_stats_merge(pcs, Bn, Wd, 1e-5, mrg, gws)
        self.gemm(xc, blk["fc_f"], Bn, out_h2=hidc, act=ACT_QUICKGELU, out_scale=HID_SCALE, alpha=inv, ln_fold=(mrg, blk["fc_f"].colsum))
        self.gemm(hidc, blk["pj"], Bn, out_h2=xc, residual_h2=(xc, inv), out_scale=X_SCALE, alpha=1.0 / HID_SCALE)
        cls = ws.f32("ccls", Bn, Wd)
        hip.gather_rows_h2(xc, inv, Bn, 1, Wd, None, 0, cls)
        return cls

    def image_features(self, image, alpha) -> torch.Tensor:
        """alpha_clip_rw/model.py:528-563 -> f32 [B][embed_dim] (un-normalised).  `image` / `alpha` may be LISTS of tensors:
        the groups are stacked along the batch axis inside the patch matrix (no copy of the inputs) and run as one forward
        -- the fused stage-2 + next-pass-1 step of Cascade.cascade(pipelined=True)."""
        c, ws = self.c, self.ws
        images = list(image) if isinstance(image, (list, tuple)) else [image]
        alphas = list(alpha) if isinstance(alpha, (list, tuple)) else [alpha]
        B = sum(int(t.shape[0]) for t in images)
        P, Wd, L = c.grid * c.grid, c.vision_width, c.n_tokens
        pt = ws.h2("cpatch", B * P, self.conv.K)
        r0 = 0
        for im, al in zip(images, alphas):
            b = int(im.shape[0])
            hip.patchify(im, al, c.patch_size, H2(pt.t[:, r0 * P:(r0 + b) * P]), self.conv.K)
            r0 += b
        pe = ws.f32("cpe", B * P, Wd)
        self.gemm(pt, self.conv, B * P, out_f32=pe)
        x = ws.f32("cx", B, L, Wd)
        hip.clip_assemble(pe, self.cls, self.pos, self.shared_ctx, B, P, Wd, c.n_ctx, x)
        hip.layernorm(x, *self.ln_pre, 1e-5, B * L, Wd, out_f32=x)
        if self.ln_fold and not self.fold_disabled:
            cls = self._vision_blocks_folded(x, B, L, Wd, c.vision_heads, L - c.n_ctx)
        else:
            self._blocks(x, self.vblocks, B, L, Wd, c.vision_heads, self.deep_vis, L - c.n_ctx, causal=False)
            cls = ws.f32("ccls", B, Wd)
            hip.gather_rows(x, B, L, Wd, None, 0, cls)
        ch = ws.h2("cclsh", B, Wd)
        hip.layernorm(cls, *self.ln_post, 1e-5, B, Wd, out_h2=ch)
        out = ws.f32("cfeat", B, c.embed_dim)
        self.gemm(ch, self.vproj, B, out_f32=out)
        return out

    def text_features(self, eot: Sequence[int], split: str = "test", rows: Optional[slice] = None) -> torch.Tensor:
        """mapleAlphaCLIP.py:64-78 on the MaPLe prompts; image independent.  Sequences are truncated to
        max(eot)+1 positions, which is exact under the causal mask.  -> f32 [n][embed_dim]."""
        c = self.c
        pre, suf = self.prefix[split], self.suffix[split]
        eot = list(int(e) for e in eot)
        L, Wd = max(eot) + 1, c.text_width                        # of ALL prompts: a shard runs the same launch shapes
        if rows is not None:                                      # per row as the full bank (bit-identical shards)
            pre, suf, eot = pre[rows], suf[rows], eot[rows]
        n = pre.shape[0]
        full = torch.cat([pre, self.ctx.unsqueeze(0).expand(n, -1, -1), suf], dim=1)[:, :L].contiguous()
        x = torch.empty(n, L, Wd, device=self.device)
        hip.add_rows(full, self.tpos[:L].contiguous(), L, n * L, Wd, out_f32=x)
        # A rank's shard must give the bits of the full bank: no K-split here -- tail chain and split-K pick their
        # number of parts from M = prompts x positions, which differs between a shard and the whole (runs once per weight load)
        self.ksplit = False
        try:
            self._blocks(x, self.tblocks, n, L, Wd, c.text_heads, self.deep_text, 1, causal=True)
            rows_f = torch.empty(n, Wd, device=self.device)
            hip.gather_rows(x, n, L, Wd, torch.tensor(eot, dtype=torch.int32, device=self.device), 0, rows_f)
            rh = H2.empty(n, Wd, device=self.device)
            hip.layernorm(rows_f, *self.ln_final, 1e-5, n, Wd, out_h2=rh)
            out = torch.empty(n, c.embed_dim, device=self.device)
            self.gemm(rh, self.tproj, n, out_f32=out)
        finally:
            self.ksplit = True
        return out

    def set_text_bank(self, text_feat: torch.Tensor, bank: torch.Tensor, split: str = "test") -> None:
        """txt = normalise(text_feat) + bank, no re-normalisation (mapleAlphaCLIP.py:290-291)."""
        n, D = text_feat.shape
        out = torch.empty(n, D, device=self.device)
        hip.normalize_add(text_feat, self.dev(bank), n, D, out)
        self.txt[split] = out

    def forward(self, image: torch.Tensor, alpha: torch.Tensor, split: str = "test"):
        """CustomCLIP.forward test branch (mapleAlphaCLIP.py:281-294)."""
        txt = self.txt[split]
        feat = self.image_features(image, alpha)              # lists of tensors: one forward over the stacked groups
        B = feat.shape[0]
        n, D = txt.shape
        img_n = torch.empty(B, D, device=self.device)
        logits = torch.empty(B, n, device=self.device)
        pred = torch.empty(B, dtype=torch.int64, device=self.device)
        sel = torch.empty(B, D, device=self.device)
        hip.clip_head(feat, txt, self.logit_scale_exp, B, n, D, img_n, logits, pred, sel)
        return img_n.unsqueeze(1), sel.unsqueeze(1), pred, logits


# ================================================================================================
# The cascade  (models/sam_maskdecoder_edge.py:331-357 + demo.py:116-122)
# ================================================================================================
class Cascade(_Base):
    def __init__(self, sd: Dict[str, torch.Tensor], g: SamGeometry, c: ClipGeometry, device,
                 precision: Precision = Precision(), clip: Optional[ClipModel] = None):
        super().__init__(device, precision)
        self.g, self.c = g, c
        self.encoder = SamEncoder(sd, g, device, precision)
        self.decoder = MaskDecoder(sd, g, device, precision)
        self.clip = clip if clip is not None else ClipModel(sd, c, device, precision)
        self.no_mask = self.dev(sd["no_mask_embed.weight"].reshape(1, -1))
        self.gauss = self.dev(sd["pe_layer.positional_encoding_gaussian_matrix"])
        self.vproj = (self.dev(sd["sam_visual_proj.0.weight"]), self.dev(sd["sam_visual_proj.0.bias"]),
                      Linear(sd["sam_visual_proj.1.weight"], sd["sam_visual_proj.1.bias"], device),
                      self.dev(sd["sam_visual_proj.2.weight"]), self.dev(sd["sam_visual_proj.2.bias"]))
        self.tproj = (self.dev(sd["sam_text_proj.0.weight"]), self.dev(sd["sam_text_proj.0.bias"]),
                      Linear(sd["sam_text_proj.1.weight"], sd["sam_text_proj.1.bias"], device))
        # CLIP pass 1 runs on a side stream under the SAM encoder (+1.3 % at B = 8, same-box A/B).  `overlap_clip = False`
        # (bench.py --no-overlap, the profiling tools) serialises it: co-running kernels stretch each other's durations,
        # which blurs per-kernel evidence.
        self.overlap_clip = True
        # pipelined loop: stage 2 of batch i and CLIP pass 1 of batch i+1 -- same weights, back to back on the side stream -- run
        # as ONE vision-tower forward over both batches (M = 9296 at B = 8: 36 row tiles instead of twice 18.2; out_proj 148
        # tiles on 256 CUs instead of twice 76).  `fuse_clip = False` keeps the two forwards apart (tests).
        self.fuse_clip = True
        # Host issue order inside a step: the encoder's first blocks BEFORE the side stream's CLIP launches (rounds 1-2 issued the
        # CLIP pass first).  It decides nothing at B = 8 (the host is far ahead of the GPU); with one image the host IS the pace
        # of a CLIP pass and the encoder behind it started 4 ms late (profiles/r03_encoder_first_ab.log).
        self._pending = None                                         # (masks, clip_image, pred, logits) of the batch whose stage 2 is still owed
        self._pending_stream = None
        self._clip_done = None                                       # end of the last fused CLIP forward (reader of the owned input copies)
        self._side = None
        self._done = [None, None]                                    # side-stream completion events of the last two batches
        self._parity = 0
        self._guard, self._guard_host, self._guard_free, self._refused_seen, self.fold_refusals = [], None, [], 0, 0
        # precision `mx` is an economy measured on synthetic weights: the first batch that would take it is preceded by a self-check on THESE weights
        self.mx_self_check = os.environ.get("CVLM_MX_SELF_CHECK", "1") != "0"
        self.mx_self_check_result: Optional[dict] = None

    # ---- precision `mx` self-check -----------------------------------------------------------------------
    # The mx arithmetic (e4m3 correction products, one-term q.k^T, P as one f16) was adopted on the evidence of synthetic weights.  A
    # checkpoint nobody has measured gets a measurement of its own: before the first batch of two or more images (the forwards that take
    # the mx path; one image per call runs the `exact` arithmetic anyway, SamEncoder.attn_split) the engine runs the first two images as a
    # batch (mx) and image 0 alone (exact) and compares every mask logit and the prediction of image 0.  Beyond MX_SELF_CHECK_TOL -- 3/4 of
    # the 1e-3 gate, the exact arithmetic itself sitting 5e-5 from the reference -- or with another prediction, it says so and serves
    # every later forward in `exact`.  Cost: one B = 2 and one B = 1 forward (~60 ms) and one host synchronisation per weight load.
    MX_SELF_CHECK_TOL = 7.5e-4

    def _mx_self_check(self, inp, clip_image, clip_mask) -> None:
        if (not self.prec.mx or not self.mx_self_check or self.mx_self_check_result is not None or inp.shape[0] < 2 or
                inp.shape[0] * self.g.grid * self.g.grid <= 4096 or torch.cuda.is_current_stream_capturing()):
            return
        self.mx_self_check_result = {"running": True}                  # (the forwards below come back through here)
        m2, p2, l2 = (t.clone() for t in self.cascade(inp[:2], clip_image[:2], clip_mask[:2]))
        m1, p1, l1 = self.cascade(inp[:1], clip_image[:1], clip_mask[:1])
        dm, dl = float((m2[:1] - m1).abs().max()), float((l2[:1] - l1).abs().max())
        same = bool(torch.equal(p2[:1], p1))
        bad = not (dm <= self.MX_SELF_CHECK_TOL and dl <= self.MX_SELF_CHECK_TOL and same) or not math.isfinite(dm)
        self.mx_self_check_result = {"max_abs_mask_diff": dm, "max_abs_class_logit_diff": dl, "pred_equal": same, "tolerance": self.MX_SELF_CHECK_TOL,
                                     "demoted_to_exact": bad}
        if bad:
            import warnings
            warnings.warn(f"camouflaged_vlm_amd: precision `mx` differs from `exact` on these weights by {dm:.2e} (mask logits) / {dl:.2e} (class "
                          f"logits), prediction {'equal' if same else 'DIFFERENT'} on the first image (tolerance {self.MX_SELF_CHECK_TOL:.1e}): "
                          "serving in precision `exact` from here on", RuntimeWarning)
            exact = dataclasses.replace(self.prec, mx=False, qk=3, pv=3)
            for eng in (self, self.encoder, self.decoder, self.clip):
                eng.prec = exact

    # ---- LayerNorm-fold refusal guard ---------------------------------------------------------------------------
    # A row with |mu| / sigma > 128 cannot be served by the folded LayerNorm within the error budget:
    # cvlm_ln_stats_merge writes NaN for it and counts it in word 513 of the GEMM workspace -- a NaN mask, never a finite wrong
    # one.  The reference's nn.LayerNorm serves such rows, so the product must not keep returning NaN: after every forward the
    # two counters (encoder, CLIP tower) are copied to pinned host memory asynchronously (no synchronisation on the path); the
    # next call that finds the copy complete and a counter raised switches BOTH towers to the separate two-pass LayerNorm
    # schedule for the rest of the model's life and says so.  The batch that met the rows has NaN outputs (loud); every later
    # batch is served.
    # The refusal guard: after every forward the two refusal counters (word 513 of the encoder's and the CLIP
    # tower's GEMM workspaces; cumulative) are copied to pinned host memory asynchronously; a later call that finds a copy complete and a
    # counter raised switches both towers to the separate LayerNorm passes.  The copies in flight form a FIFO over a small pool of pinned
    # slots and the check polls the OLDEST (one slot re-armed on every call is never complete when the next call looks at it
    # in a loop that runs ahead of the GPU -- the pipelined loops -- and the switch never fired there).  With every slot in flight a
    # call arms nothing: the counters are cumulative, a later copy carries the same news.
    _GUARD_SLOTS = 4

    def _fold_guard_check(self) -> None:
        if torch.cuda.is_current_stream_capturing():                 # (an event query would invalidate a hipGraph capture)
            return
        q = self._guard
        while q and q[0][1].query():
            slot, _ = q.pop(0)
            refused = int(self._guard_host[slot, 0]) + int(self._guard_host[slot, 1])
            self._guard_free.append(slot)
            if refused > self._refused_seen:
                self._refused_seen = refused
                self.fold_refusals = refused
                self.encoder.fold_disabled = True
                self.clip.fold_disabled = True
                import warnings
                warnings.warn(f"camouflaged_vlm_amd: {refused} token row(s) with |mean| / std > 128 were refused by the folded LayerNorm "
                              "(their images came out NaN); switching to the separate LayerNorm passes from this batch on "
                              "(set CVLM_LN_FOLD=0 to start that way)", RuntimeWarning, stacklevel=3)

    def _fold_guard_arm(self, stream) -> None:
        if not ((self.encoder.ln_fold and not self.encoder.fold_disabled) or (self.clip.ln_fold and not self.clip.fold_disabled)):
            return
        if torch.cuda.is_current_stream_capturing():                 # a captured step carries no host-side check: the caller of a
            return                                                   # graph reads ws.gemm_errors() itself (tools/graph_step.py)
        if self._guard_host is None:
            self._guard_host = torch.zeros(self._GUARD_SLOTS, 2, dtype=torch.int32).pin_memory()
            self._guard_free = list(range(self._GUARD_SLOTS))
        if not self._guard_free:
            return
        slot = self._guard_free.pop(0)
        with torch.cuda.stream(stream):
            for k, eng in enumerate((self.encoder, self.clip)):
                w = eng.ws._gemm_ws
                if w is not None:
                    self._guard_host[slot, k:k + 1].copy_(w[2052:2056].view(torch.int32), non_blocking=True)   # word 513
                else:
                    self._guard_host[slot, k] = 0
            ev = torch.cuda.Event()
            ev.record(stream)
        self._guard.append((slot, ev))

    def sparse_prompts(self, img_f: torch.Tensor, txt_f: torch.Tensor, B: int) -> torch.Tensor:
        """models/sam_maskdecoder_edge.py:342-344."""
        ws, C = self.ws, self.g.prompt_embed_dim
        h = ws.h2("sp_h", B, 768)
        sparse = ws.f32("sparse", B, 2, C)
        tmp = ws.f32("sp_tmp", B, C)
        hip.layernorm(img_f.reshape(B, 768), self.vproj[0], self.vproj[1], 1e-5, B, 768, out_h2=h)
        self.gemm(h, self.vproj[2], B, out_f32=tmp)
        hip.layernorm(tmp, self.vproj[3], self.vproj[4], 1e-5, B, C, out_f32=tmp)
        sparse[:, 0].copy_(tmp)
        hip.layernorm(txt_f.reshape(B, 768), self.tproj[0], self.tproj[1], 1e-5, B, 768, out_h2=h)
        self.gemm(h, self.tproj[2], B, out_f32=tmp)
        sparse[:, 1].copy_(tmp)
        return sparse

    def infer_test(self, inp, clip_image, clip_mask, taps: Optional[dict] = None) -> torch.Tensor:
        self._mx_self_check(inp, clip_image, clip_mask)
        self.flush()
        self._fold_guard_check()
        g = self.g
        B = inp.shape[0]
        # CLIP pass 1 only needs (clip_image, clip_mask): it runs on a side stream underneath the SAM encoder so
        # its small grids (228..1184 workgroups) fill CUs the big encoder GEMMs leave idle at tile boundaries
        if self.overlap_clip:
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.device)
            main = torch.cuda.current_stream()
            # the host issues the ENCODER's first blocks before the CLIP pass: one image's pass is ~190 launches of a few
            # workgroups each, the GPU runs them as fast as the host can issue them, and an encoder issued behind them started
            # 4 ms late with the chip nearly idle until then (tools/step_timeline.py).  Five blocks (3 ms of GPU work for one
            # image) are queued first, then the pass, then the rest: a slow host (a profiler, a loaded node) does not push the
            # pass behind ALL of the encoder's launches either.  The side stream waits for the inputs, not for the encoder.
            ready = torch.cuda.Event()
            ready.record(main)
            res = []

            def issue_clip():
                self._side.wait_event(ready)
                with torch.cuda.stream(self._side):
                    res.append(self.clip.forward(clip_image, clip_mask))

            feats = self.encoder.forward(inp, taps, issue_hook=issue_clip)
            img_f, txt_f, pred, score = res[0]
            main.wait_stream(self._side)
        else:
            feats = self.encoder.forward(inp, taps)
            img_f, txt_f, pred, score = self.clip.forward(clip_image, clip_mask)
        sparse = self.sparse_prompts(img_f, txt_f, B)
        low = self.decoder.forward(feats, sparse, self.no_mask, self.gauss, B, taps)
        masks = torch.empty(B, 1, g.inp_size, g.inp_size, device=self.device)
        hip.bilinear(low, B, 4 * g.grid, 4 * g.grid, masks, g.inp_size, g.inp_size)   # :380-387 (2nd resize = identity)
        if taps is not None:
            taps.update(features=feats.clone(), sparse=sparse.clone(), pass1_logits=score.clone())
        self._fold_guard_arm(torch.cuda.current_stream())
        return masks

    def stage2(self, mask_logits: torch.Tensor, clip_image: torch.Tensor):
        """demo.py:117-122: alpha = resize(sigmoid(mask)) -> clip_model(image, alpha)."""
        B, R = mask_logits.shape[0], self.c.image_resolution
        alpha = self.ws.f32("alpha2", B, 1, R, R)
        hip.bilinear(mask_logits, B, self.g.inp_size, self.g.inp_size, alpha, R, R, sigmoid_in=True)
        return self.clip.forward(clip_image, alpha)

    def cascade(self, inp, clip_image, clip_mask, pipelined: bool = False):
        """Stage 1 + stage 2.  With the side stream enabled only the SAM encoder runs on the caller's stream; CLIP pass
        1, the sparse prompts, the mask decoder (hundreds of launches of a few workgroups each), the resize and stage 2
        run on the side stream.
        pipelined=False: the current stream waits for the side stream before returning (plain stream semantics).
        pipelined=True: it does not -- the caller's next batch starts its encoder underneath this batch's decoder and
        stage 2 (a serving loop; the next `cascade()` orders itself behind the previous one, at most one batch is in flight
        there).  CONTRACT of the pipelined loop (default form: `fuse_clip`, `_cascade_fused`): `masks` of a call are
        complete after `torch.cuda.synchronize()`, but its `pred` / `logits` are OWED -- they are filled by the NEXT
        `cascade(pipelined=True)` call or by `flush()`, then complete after the following synchronize.  A caller that stops
        feeding batches MUST call `flush()`; until then `pred` holds -1 and `logits` NaN (sentinels, never stale data).
        The inputs may be refilled in place as soon as the call has returned: `clip_image` / `clip_mask` are copied (on the
        caller's stream) into buffers the engine owns before anything reads them later."""
        self._mx_self_check(inp, clip_image, clip_mask)
        if pipelined and self.fuse_clip:
            return self._cascade_fused(inp, clip_image, clip_mask)
        self.flush()
        self._fold_guard_check()
        if not self.overlap_clip:
            masks = self.infer_test(inp, clip_image, clip_mask)
            _, _, pred, logits = self.stage2(masks, clip_image)
            return masks, pred, logits
        g, B = self.g, inp.shape[0]
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        main, side = torch.cuda.current_stream(), self._side
        # the batch before the previous one has left the side stream: its decoder read the features buffer this encoder
        # run is about to overwrite (two buffers alternate)
        if self._done[self._parity] is not None:
            main.wait_event(self._done[self._parity])
        ready = torch.cuda.Event()                                   # encoder's first blocks, then the CLIP pass: see infer_test
        ready.record(main)
        res = []

        def issue_clip():
            side.wait_event(ready)
            with torch.cuda.stream(side):
                res.append(self.clip.forward(clip_image, clip_mask))

        feats = self.encoder.forward(inp, None, out_name="features%d" % self._parity, issue_hook=issue_clip)
        enc_done = torch.cuda.Event()
        enc_done.record(main)
        img_f, txt_f, _, _ = res[0]
        with torch.cuda.stream(side):
            side.wait_event(enc_done)
            sparse = self.sparse_prompts(img_f, txt_f, B)
            low = self.decoder.forward(feats, sparse, self.no_mask, self.gauss, B, None)
            masks = torch.empty(B, 1, g.inp_size, g.inp_size, device=self.device)
            hip.bilinear(low, B, 4 * g.grid, 4 * g.grid, masks, g.inp_size, g.inp_size)
            _, _, pred, logits = self.stage2(masks, clip_image)
            done = torch.cuda.Event()
            done.record(side)
        self._fold_guard_arm(side)
        self._done[self._parity] = done
        self._parity ^= 1
        for t in (inp, clip_image, clip_mask):
            t.record_stream(side)
        for t in (masks, pred, logits):
            t.record_stream(main)
        if not pipelined:
            main.wait_stream(side)
        return masks, pred, logits

    def _cascade_fused(self, inp, clip_image, clip_mask):
        """pipelined=True with the two CLIP forwards of a step fused: this call launches, on the side stream, ONE vision-tower
        forward over [stage 2 of the PREVIOUS batch | pass 1 of this batch], then this batch's decoder behind its encoder; the
        stage 2 of this batch is owed until the next call -- or `flush()`, which a caller that stops feeding batches must
        issue (bench.py does, inside the timed region).  The returned `pred` / `logits` tensors hold sentinels (-1 / NaN) until
        that later launch fills them; `masks` is filled by this one.  cocotrainers/mapleAlphaCLIP.py:281-294 twice,
        demo.py:117-122."""
        g, B = self.g, inp.shape[0]
        self._fold_guard_check()
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream()
        side = self._side if self.overlap_clip else main             # overlap_clip = False: same launches, one stream (profiling)
        self._pending_stream = side
        if self._done[self._parity] is not None:                     # the batch before the previous one has left the side stream
            main.wait_event(self._done[self._parity])
        prev = self._pending
        # The side stream reads clip_image / clip_mask now (pass 1) and clip_image again one call later (the owed stage 2): keep
        # copies the engine owns, made on the caller's stream, so that a serving loop may refill its input buffers in place
        # (ADVICE r3).  3.6 MB at B = 8; two buffers alternate with the batches in flight.
        R = self.c.image_resolution
        if self._clip_done is not None:                              # the forward that read these buffers two batches ago
            main.wait_event(self._clip_done)                         # (it ran under the previous encoder: long finished)
        own_image = self.ws.f32("own_clip_image%d" % self._parity, B, 3, R, R)
        own_mask = self.ws.f32("own_clip_mask%d" % self._parity, B, 1, R, R)
        own_image.copy_(clip_image)
        own_mask.copy_(clip_mask)
        clip_image, clip_mask = own_image, own_mask

        def clip_forwards():
            try:
                if prev is None:
                    i_f, t_f, _, _ = self.clip.forward(clip_image, clip_mask)
                    return i_f, t_f
                p_masks, p_image, p_pred, p_logits = prev
                Bp = p_masks.shape[0]
                alpha = self.ws.f32("alpha2", Bp, 1, R, R)
                hip.bilinear(p_masks, Bp, g.inp_size, g.inp_size, alpha, R, R, sigmoid_in=True)
                img_n, sel, pred_all, logits_all = self.clip.forward([p_image, clip_image], [alpha, clip_mask])
                p_pred.copy_(pred_all[:Bp])                          # results of the previous batch land in the tensors it returned
                p_logits.copy_(logits_all[:Bp])
                return img_n[Bp:], sel[Bp:]
            finally:
                self._clip_done = torch.cuda.Event()
                self._clip_done.record(torch.cuda.current_stream())

        if side is not main:                                         # encoder's first blocks, then the CLIP pass: see infer_test
            ready = torch.cuda.Event()
            ready.record(main)
            res = []

            def issue_clip():
                side.wait_event(ready)
                with torch.cuda.stream(side):
                    res.append(clip_forwards())

            feats = self.encoder.forward(inp, None, out_name="features%d" % self._parity, issue_hook=issue_clip)
            enc_done = torch.cuda.Event()
            enc_done.record(main)
            img_f, txt_f = res[0]
        else:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                img_f, txt_f = clip_forwards()
            feats = self.encoder.forward(inp, None, out_name="features%d" % self._parity)
            enc_done = torch.cuda.Event()
            enc_done.record(main)
        with torch.cuda.stream(side):
            side.wait_event(enc_done)
            sparse = self.sparse_prompts(img_f, txt_f, B)
            low = self.decoder.forward(feats, sparse, self.no_mask, self.gauss, B, None)
            masks = torch.empty(B, 1, g.inp_size, g.inp_size, device=self.device)
            hip.bilinear(low, B, 4 * g.grid, 4 * g.grid, masks, g.inp_size, g.inp_size)
            n_cls = self.clip.txt["test"].shape[0]
            # owed until the next call / flush(): sentinels, so that a missing flush() reads as "not computed", never as data
            pred = torch.full((B,), -1, dtype=torch.int64, device=self.device)
            logits = torch.full((B, n_cls), float("nan"), device=self.device)
            done = torch.cuda.Event()
            done.record(side)
        self._fold_guard_arm(side)
        self._done[self._parity] = done
        self._parity ^= 1
        self._pending = (masks, clip_image, pred, logits)
        for t in (inp,):
            t.record_stream(side)
        for t in (masks, pred, logits):
            t.record_stream(main)
        return masks, pred, logits

    def batch_done_event(self) -> "torch.cuda.Event":
        """Event behind everything the LAST `cascade(pipelined=True)` call put on the side stream: that batch's decoder and, in front of
        it, the forward that filled the PREVIOUS batch's pred / logits.  A consumer on another stream (DeviceEvalLoop's tail stream)
        waits for it instead of reading the engine's bookkeeping.  Valid for every configuration: where a call left no
        side-stream event (no side stream in use), one is recorded on the current stream."""
        ev = self._done[self._parity ^ 1]
        if ev is None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
        return ev

    def results_ready_event(self) -> "torch.cuda.Event":
        """After `flush()`: event behind the forward that filled the last owed pred / logits."""
        ev = self._clip_done if self._clip_done is not None else self._done[self._parity ^ 1]
        if ev is None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
        return ev

    def flush(self) -> None:
        """Launch the stage 2 that `cascade(pipelined=True)` still owes for its last batch (no-op otherwise); results are
        complete once the side stream has been waited for / after `torch.cuda.synchronize()`."""
        prev, self._pending = self._pending, None
        if prev is None:
            return
        p_masks, p_image, p_pred, p_logits = prev
        side = self._pending_stream
        with torch.cuda.stream(side):
            _, _, pred, logits = self.stage2(p_masks, p_image)
            p_pred.copy_(pred)
            p_logits.copy_(logits)
            done = torch.cuda.Event()
            done.record(side)
        self._clip_done = done                                       # this forward read the owned copy of the batch's clip_image
        self._done[self._parity ^ 1] = done                          # the slot of the batch just completed
