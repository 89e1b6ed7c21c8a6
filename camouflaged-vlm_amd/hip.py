"""ctypes binding of libcvlm_hip.so (include/cvlm.h) -- the only way compute happens in this package.

There is no CPU fallback: if the shared library is missing or a launch fails, a RuntimeError is
raised.  torch is used for device memory and streams only (tensors' ``data_ptr()`` are handed to the
C ABI together with the current HIP stream).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libcvlm_hip.so")

ACT_NONE, ACT_GELU, ACT_QUICKGELU, ACT_RELU, ACT_ABS_POST = 0, 1, 2, 3, 4

EXPORTS = [
    "cvlm_abi_version", "cvlm_target_arch", "cvlm_gemm", "cvlm_layernorm", "cvlm_add_rows", "cvlm_split_f32",
    "cvlm_patchify", "cvlm_im2col3x3", "cvlm_reinterpret_transpose", "cvlm_attention", "cvlm_small_attention",
    "cvlm_dense_pe", "cvlm_mask_head", "cvlm_bilinear", "cvlm_clip_assemble", "cvlm_overwrite_rows",
    "cvlm_gather_rows", "cvlm_clip_head", "cvlm_normalize_add", "cvlm_resample_u8", "cvlm_u8_to_tensor",
    "cvlm_mask_to_u8", "cvlm_mask_joint_hist", "cvlm_mask_wfm", "cvlm_topk_accumulate",
    "cvlm_gemm_workspace_bytes", "cvlm_attention_workspace_bytes", "cvlm_row_stats_split", "cvlm_row_stats_split_mx", "cvlm_gather_rows_h2",
    "cvlm_ln_stats_merge", "cvlm_small_attention_h2", "cvlm_prob_quantise", "cvlm_prob_moments", "cvlm_prob_wfm",
]
ABI_VERSION = 12


class GemmArgs(C.Structure):
    _fields_ = [
        ("a_hi", C.c_void_p), ("a_lo", C.c_void_p), ("lda", C.c_int64), ("stride_a", C.c_int64),
        ("w_hi", C.c_void_p), ("w_lo", C.c_void_p), ("ldw", C.c_int64), ("stride_w", C.c_int64),
        ("bias", C.c_void_p),
        ("residual", C.c_void_p), ("ldr", C.c_int64), ("stride_r", C.c_int64),
        ("out_f32", C.c_void_p), ("ldo", C.c_int64), ("stride_o", C.c_int64),
        ("out_hi", C.c_void_p), ("out_lo", C.c_void_p), ("ldoh", C.c_int64), ("stride_oh", C.c_int64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("batch", C.c_int32),
        ("alpha", C.c_float), ("act", C.c_int32), ("split", C.c_int32),
        ("ps_h", C.c_int32), ("ps_w", C.c_int32), ("ps_c2", C.c_int32),
        ("hm_S", C.c_int32), ("hm_H", C.c_int32), ("hm_hd", C.c_int32),
        ("out_scale", C.c_float), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("ln_stats", C.c_void_p), ("ln_colsum", C.c_void_p), ("ln_eps", C.c_float), ("ln_D", C.c_int32),
        ("res_hi", C.c_void_p), ("res_lo", C.c_void_p), ("ldrh", C.c_int64), ("res_scale", C.c_float),
        ("row_stats", C.c_void_p),
        ("conv_h", C.c_int32), ("conv_w", C.c_int32), ("conv_c", C.c_int32),
        ("w_il", C.c_void_p), ("ldw_il", C.c_int64),
        ("a_il", C.c_int32), ("out_il", C.c_int32), ("res_il", C.c_int32),
        ("a_mx", C.c_int32), ("out_mx", C.c_int32), ("res_mx", C.c_int32),
        ("a_mxs", C.c_void_p), ("lda_s", C.c_int64),
        ("w_mx", C.c_void_p), ("ldw_mx", C.c_int64), ("w_mxs", C.c_void_p), ("ldw_s", C.c_int64),
        ("out_mxs", C.c_void_p), ("ldo_s", C.c_int64), ("ldol", C.c_int64), ("ldrl", C.c_int64),
        ("hm_nolo", C.c_int32),
    ]


class AttnArgs(C.Structure):
    _fields_ = [
        ("qkv_hi", C.c_void_p), ("qkv_lo", C.c_void_p), ("pad_hi", C.c_void_p), ("pad_lo", C.c_void_p),
        ("relh_hi", C.c_void_p), ("relh_lo", C.c_void_p), ("relw_hi", C.c_void_p), ("relw_lo", C.c_void_p),
        ("out_hi", C.c_void_p), ("out_lo", C.c_void_p),
        ("B", C.c_int32), ("S", C.c_int32), ("heads", C.c_int32), ("hd", C.c_int32),
        ("mode", C.c_int32), ("grid", C.c_int32), ("window", C.c_int32), ("causal", C.c_int32),
        ("split_qk", C.c_int32), ("split_pv", C.c_int32), ("scale", C.c_float), ("qkv_layout", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64), ("q_rows", C.c_int32),
    ]


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the shared library (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("CVLM_PROBE_LIB") or LIB_PATH              # CVLM_PROBE_LIB: probe builds of the same ABI (tools/)
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: the HIP kernels are the only compute path of this package. "
            "Build them with `python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950).")
    lib = C.CDLL(path)
    lib.cvlm_abi_version.restype = C.c_int
    lib.cvlm_target_arch.restype = C.c_char_p
    for name in EXPORTS[2:]:
        getattr(lib, name).restype = C.c_int64 if name.endswith("_workspace_bytes") else C.c_int
    if lib.cvlm_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{path}: ABI {lib.cvlm_abi_version()}, this binding needs {ABI_VERSION}: rebuild the library")
    _lib = lib
    return lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}")


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream() -> Optional[int]:
    return torch.cuda.current_stream().cuda_stream or None


def _on_current_device(t: torch.Tensor) -> None:
    """Kernels launch on the CURRENT device's stream: a tensor living elsewhere would fault or silently go over xGMI
    (one device per process is the contract, include/cvlm.h; checked on the two heavy entries)."""
    if t.device.type != "cuda" or t.device.index != torch.cuda.current_device():
        raise RuntimeError(f"tensor on {t.device}, current device is cuda:{torch.cuda.current_device()}: call "
                           "torch.cuda.set_device() first (launches use the current device's stream)")


def gemm_workspace_bytes() -> int:
    return int(load().cvlm_gemm_workspace_bytes())


def new_gemm_workspace(device) -> torch.Tensor:
    """Zero-filled (the 4-KiB hand-off page must start at zero, include/cvlm.h) split-K workspace for cvlm_gemm."""
    return torch.zeros(gemm_workspace_bytes(), dtype=torch.uint8, device=device)


def gemm_workspace_errors(ws: torch.Tensor) -> int:
    """Abandoned split-K hand-offs (word 512) + rows a LayerNorm-folded launch refused (word 513) recorded in a workspace
    (synchronises).  Either kind has already turned the affected outputs into NaN.  0 in a healthy run."""
    words = ws[2048:2056].view(torch.int32).tolist()
    return int(words[0]) + int(words[1])


class H2IL:
    """Split-half activation in the 128-byte-row image of include/cvlm.h (ABI 6): one fp16 tensor [M][2 * C], row m =
    (hi c0..31 | lo c0..31 | hi c32..63 | lo c32..63 | ...).  Only cvlm_gemm (a_il / out_il / res_il) and cvlm_row_stats_split read or
    write it; `cols(c0)` is the view that starts at column c0 (c0 % 32 == 0) with the same row stride."""

    __slots__ = ("t",)
    il = True

    def __init__(self, t: torch.Tensor):
        assert t.dtype == torch.float16 and t.dim() == 2 and t.stride(1) == 1
        self.t = t

    @staticmethod
    def empty(M: int, C_: int, device="cuda") -> "H2IL":
        assert C_ % 32 == 0
        return H2IL(torch.empty(M, 2 * C_, dtype=torch.float16, device=device))

    @staticmethod
    def from_planes(x: "H2") -> "H2IL":
        return H2IL(interleave_planes(x))

    def cols(self, c0: int) -> "H2IL":
        assert c0 % 32 == 0
        return H2IL(self.t[:, 2 * c0:])

    def planes(self) -> "H2":
        M, C2 = self.t.shape
        return H2(self.t.reshape(M, C2 // 64, 2, 32).permute(2, 0, 1, 3).reshape(2, M, C2 // 2).contiguous())

    def float(self) -> torch.Tensor:
        return self.planes().float()


def mx_scale_pitch(C_: int) -> int:
    """Bytes per scale plane and row of an mx operand with C_ columns (include/cvlm.h ABI 10): one byte per 64-column group, padded to
    whole dwords (the kernel reads the bytes of four groups at once)."""
    return (C_ // 64 + 3) // 4 * 4


def mx_pack(x: "H2"):
    """h2 planes [2][R][C] (C % 64 == 0) -> (image uint8 [R][C / 64][256], scales uint8 [R][4][mx_scale_pitch(C)]) of include/cvlm.h
    ABI 10, in torch (weights at load time; the reference the tests hold the GEMM epilogue's own mx output against, bit for bit):
    per 32 columns E = max(exponent field of the largest |hi| as f32, 103) - 7, hi8 = e4m3(hi / 2^(E - 127)), lo8 = e4m3(lo / 2^(E - 138)).
    `lo` must be what the split leaves (|lo| <= half an ulp of its hi: H2.pack, every producer kernel): the lo8 scale is sized for that, and a
    plane of unrelated values overflows e4m3 (NaN bytes, as the hardware conversion gives)."""
    two, R, C_ = x.t.shape
    assert two == 2 and C_ % 64 == 0
    hi, lo = x.t[0], x.t[1]
    hf = hi.float()
    bmax = hf.abs().view(R, C_ // 32, 32).amax(-1)
    ex = ((bmax.view(torch.int32) >> 23) & 0xff).clamp_min(103) - 7                      # [R][C / 32]
    s_hi = (ex << 23).view(torch.float32)
    s_lo = ((ex - 11) << 23).view(torch.float32)
    hi8 = (hf.view(R, C_ // 32, 32) / s_hi[..., None]).to(torch.float8_e4m3fn).view(torch.uint8).view(R, C_ // 64, 64)
    lo8 = (lo.float().view(R, C_ // 32, 32) / s_lo[..., None]).to(torch.float8_e4m3fn).view(torch.uint8).view(R, C_ // 64, 64)
    img = torch.cat([hi.contiguous().view(torch.uint8).view(R, C_ // 64, 128), hi8, lo8], dim=2).contiguous()
    su = mx_scale_pitch(C_)
    sc = torch.zeros(R, 4, su, dtype=torch.uint8, device=x.t.device)
    e2 = ex.view(R, C_ // 64, 2).to(torch.uint8)
    sc[:, 0, :C_ // 64], sc[:, 1, :C_ // 64] = e2[:, :, 0], e2[:, :, 1]
    sc[:, 2, :C_ // 64], sc[:, 3, :C_ // 64] = e2[:, :, 0] - 11, e2[:, :, 1] - 11
    return img, sc


class H2MX:
    """Split-half activation as an mx operand (include/cvlm.h ABI 10): `t` uint8 [M][4 * C] -- per 64 columns 128 bytes of fp16 hi, 64
    of e4m3 hi8, 64 of e4m3 lo8 --, `s` uint8 [M][4][pitch] the block exponents, `lo` (optional) the fp16 lo PLANE [M][C] kept beside
    the image where a later launch reads the tensor as its h2 residual.  Written by cvlm_gemm (out_mx), read by it (a_mx / res_mx)."""

    __slots__ = ("t", "s", "lo", "C", "c0")
    mx = True

    def __init__(self, t: torch.Tensor, s: torch.Tensor, lo: Optional[torch.Tensor], C_: int, c0: int = 0):
        assert t.dtype == torch.uint8 and t.dim() == 2 and s.dtype == torch.uint8 and s.dim() == 3 and s.shape[1] == 4
        self.t, self.s, self.lo, self.C, self.c0 = t, s, lo, C_, c0

    @staticmethod
    def empty(M: int, C_: int, device="cuda", lo_plane: bool = False) -> "H2MX":
        assert C_ % 64 == 0
        return H2MX(torch.empty(M, 4 * C_, dtype=torch.uint8, device=device),
                    torch.zeros(M, 4, mx_scale_pitch(C_), dtype=torch.uint8, device=device),
                    torch.empty(M, C_, dtype=torch.float16, device=device) if lo_plane else None, C_)

    @staticmethod
    def from_planes(x: "H2", lo_plane: bool = False) -> "H2MX":
        img, sc = mx_pack(x)
        return H2MX(img.view(img.shape[0], -1), sc, x.t[1].contiguous() if lo_plane else None, x.t.shape[2])

    def cols(self, c0: int) -> "H2MX":
        """The operand that starts at column c0 (c0 % 64 == 0): same rows, same pitches."""
        assert c0 % 64 == 0 and self.lo is None
        return H2MX(self.t, self.s, None, self.C - c0, self.c0 + c0)

    def every(self, step: int) -> "H2MX":
        """Rows 0, step, 2 * step, ... as an operand of their own (views: same memory, `step` times the row pitches) -- the class-token
        rows of a [B * L]-row stream."""
        return H2MX(self.t[::step], self.s[::step], None if self.lo is None else self.lo[::step], self.C, self.c0)

    def data_ptr(self) -> int:
        return self.t.data_ptr() + 4 * self.c0

    def scale_ptr(self) -> int:
        return self.s.data_ptr() + self.c0 // 64

    def hi(self) -> torch.Tensor:
        M = self.t.shape[0]
        return self.t.view(M, -1, 256)[:, :, :128].contiguous().view(torch.float16).view(M, -1)[:, self.c0:self.c0 + self.C]

    def planes8(self):
        """(hi8, lo8) decoded to f32 [M][C]: what the fp8 products multiply."""
        M = self.t.shape[0]
        g = self.t.view(M, -1, 256)
        n = g.shape[1]
        e = self.s[:, :, :n].permute(0, 2, 1).float()                                     # [M][groups][4]
        dec = lambda b, pl: (b.contiguous().view(torch.float8_e4m3fn).float().view(M, n, 2, 32) *
                             torch.exp2(e[:, :, pl:pl + 2] - 127.0)[..., None]).view(M, -1)[:, self.c0:self.c0 + self.C]
        return dec(g[:, :, 128:192], 0), dec(g[:, :, 192:256], 2)


def interleave_planes(w: "H2") -> torch.Tensor:
    """[2][N][K] planes -> fp16 [N][2K] with row n = (hi k0..31 | lo k0..31 | hi k32..63 | lo k32..63 | ...): the `w_il` image of
    cvlm_gemm (include/cvlm.h, ABI 6).  K % 32 == 0."""
    two, N, K = w.t.shape
    assert two == 2 and K % 32 == 0
    return w.t.reshape(2, N, K // 32, 32).permute(1, 2, 0, 3).reshape(N, 2 * K).contiguous()


class H2:
    """Split-half tensor: two fp16 planes stored as one (2, *shape) fp16 tensor."""

    __slots__ = ("t",)

    def __init__(self, t: torch.Tensor):
        assert t.dtype == torch.float16 and t.shape[0] == 2
        self.t = t

    @staticmethod
    def empty(*shape, device="cuda") -> "H2":
        return H2(torch.empty((2,) + tuple(shape), dtype=torch.float16, device=device))

    @staticmethod
    def zeros(*shape, device="cuda") -> "H2":
        return H2(torch.zeros((2,) + tuple(shape), dtype=torch.float16, device=device))

    @staticmethod
    def pack(x: torch.Tensor) -> "H2":
        """Host-side (torch) packing of weights / constants: hi = fp16(x), lo = fp16(x - hi)."""
        x = x.float()
        hi = x.half()
        lo = (x - hi.float()).half()
        return H2(torch.stack([hi, lo]).contiguous())

    @property
    def hi(self) -> torch.Tensor:
        return self.t[0]

    @property
    def lo(self) -> torch.Tensor:
        return self.t[1]

    @property
    def shape(self):
        return self.t.shape[1:]

    def float(self) -> torch.Tensor:
        return self.t[0].float() + self.t[1].float()

    def view(self, *shape) -> "H2":
        return H2(self.t.view((2,) + tuple(shape)))


def gemm(a: H2, w: H2, M: int, N: int, K: int, *, lda: Optional[int] = None, ldw: Optional[int] = None,
         bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, ldr: Optional[int] = None,
         out_f32: Optional[torch.Tensor] = None, ldo: Optional[int] = None, out_h2: Optional[H2] = None,
         ldoh: Optional[int] = None, alpha: float = 1.0, act: int = ACT_NONE, split: int = 3, batch: int = 1,
         stride_a: int = 0, stride_w: int = 0, stride_r: int = 0, stride_o: int = 0, stride_oh: int = 0,
         pixel_shuffle: Optional[Tuple[int, int, int]] = None,
         head_major: Optional[Tuple[int, int, int]] = None, head_major_nolo: int = 0, out_scale: float = 1.0,
         workspace: Optional[torch.Tensor] = None,
         ln_fold: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
         residual_h2: Optional[Tuple["H2", float]] = None, ldrh: Optional[int] = None,
         row_stats: Optional[torch.Tensor] = None, conv3x3: Optional[Tuple[int, int, int]] = None,
         w_il: Optional[torch.Tensor] = None, w_mx: Optional["H2MX"] = None) -> None:
    """w_mx: the weight as an mx operand (`H2MX.from_planes(w)`, ABI 10) -- used when `a` is an H2MX: hi.hi on fp16, the two correction
    products on the block-scaled e4m3 matrix instruction.  An H2MX `out_h2` / `residual_h2` selects out_mx / res_mx.
    w_il: the same weight with its planes interleaved per 32 k-elements (`interleave_planes(w)`, ABI 6): the big-tile kernels stage
    the weight from it (whole 128-byte lines per row and K-tile), same bits.
    conv3x3 = (H, W, C): `a` is an NHWC image [B*H*W][C] and K = 9*C runs over the taps of a 3x3 / pad 1 convolution
    (implicit GEMM: the im2col gather happens in the DMA addresses).
    ln_fold = (merged [M][2] f32 = (rstd, mu * rstd) from ln_stats_merge, colsum [N] f32): LayerNorm of the input folded into this
    GEMM (include/cvlm.h);
    residual_h2 = (x h2, scale): residual given as h2 planes; row_stats [ceil(N/64)][M][2] f32: piece statistics of the result rows
    (plain stores, bit-reproducible: nothing to zero)."""
    _on_current_device(a.t)
    g = GemmArgs()
    if getattr(a, "mx", False):
        assert w_mx is not None and w_mx.C >= K and a.C >= K, "an mx activation needs the mx image of the weight"
        g.a_hi, g.a_lo, g.lda, g.stride_a, g.a_mx = a.data_ptr(), 0, a.t.stride(0) // 2, 0, 1
        assert a.s.stride(0) == 4 * a.s.stride(1), "an H2MX.every() view is a residual only: the kernels take the scale row stride as 4 * ld_s"
        g.a_mxs, g.lda_s = a.scale_ptr(), a.s.stride(1)
        g.w_mx, g.ldw_mx, g.w_mxs, g.ldw_s = w_mx.data_ptr(), w_mx.t.stride(0) // 2, w_mx.scale_ptr(), w_mx.s.stride(1)
    elif getattr(a, "il", False):
        g.a_hi, g.a_lo, g.lda, g.stride_a, g.a_il = a.t.data_ptr(), 0, a.t.stride(0), 0, 1
    else:
        g.a_hi, g.a_lo, g.lda, g.stride_a = a.hi.data_ptr(), a.lo.data_ptr(), lda if lda is not None else K, stride_a
    g.w_hi, g.w_lo, g.ldw, g.stride_w = w.hi.data_ptr(), w.lo.data_ptr(), ldw if ldw is not None else K, stride_w
    if w_il is not None:
        assert w_il.dtype == torch.float16 and w_il.dim() == 2 and w_il.shape[1] >= 2 * K and w_il.is_contiguous() and batch == 1
        g.w_il, g.ldw_il = w_il.data_ptr(), w_il.shape[1]
    g.bias = _p(bias)
    g.residual, g.ldr, g.stride_r = _p(residual), (ldr if ldr is not None else N), stride_r
    g.out_f32, g.ldo, g.stride_o = _p(out_f32), (ldo if ldo is not None else N), stride_o
    g.ldoh, g.stride_oh = (ldoh if ldoh is not None else N), stride_oh
    if out_h2 is not None and getattr(out_h2, "mx", False):
        g.out_hi, g.ldoh, g.out_mx = out_h2.data_ptr(), out_h2.t.stride(0) // 2, 1
        assert out_h2.s.stride(0) == 4 * out_h2.s.stride(1), "an H2MX.every() view is a residual only: the kernels take the scale row stride as 4 * ld_s"
        g.out_mxs, g.ldo_s = out_h2.scale_ptr(), out_h2.s.stride(1)
        if out_h2.lo is not None:
            g.out_lo, g.ldol = out_h2.lo.data_ptr(), out_h2.lo.stride(0)
    elif out_h2 is not None and getattr(out_h2, "il", False):
        g.out_hi, g.out_lo, g.ldoh, g.out_il = out_h2.t.data_ptr(), 0, out_h2.t.stride(0), 1
    elif out_h2 is not None:
        g.out_hi, g.out_lo = out_h2.hi.data_ptr(), out_h2.lo.data_ptr()
    g.M, g.N, g.K, g.batch = M, N, K, batch
    g.alpha, g.act, g.split = alpha, act, split
    if pixel_shuffle is not None:
        g.ps_h, g.ps_w, g.ps_c2 = pixel_shuffle
    if head_major is not None:
        g.hm_S, g.hm_H, g.hm_hd = head_major
        g.hm_nolo = head_major_nolo                                # ABI 12: bit w set = the lo plane of q (0) / k (1) / v (2) is not written
    g.out_scale = out_scale
    if ln_fold is not None:
        assert tuple(ln_fold[0].shape) == (M, 2) and ln_fold[0].is_contiguous(), "ln_fold: merged statistics [M][2]"
        g.ln_stats, g.ln_colsum = ln_fold[0].data_ptr(), ln_fold[1].data_ptr()
    if residual_h2 is not None:
        r, rs = residual_h2
        if getattr(r, "mx", False):
            assert r.lo is not None, "an mx residual needs its fp16 lo plane"
            g.res_hi, g.res_lo, g.ldrh, g.res_scale, g.res_mx = r.data_ptr(), r.lo.data_ptr(), r.t.stride(0) // 2, rs, 1
            g.ldrl = r.lo.stride(0)
        elif getattr(r, "il", False):
            g.res_hi, g.res_lo, g.ldrh, g.res_scale, g.res_il = r.t.data_ptr(), 0, r.t.stride(0), rs, 1
        else:
            g.res_hi, g.res_lo, g.ldrh, g.res_scale = r.hi.data_ptr(), r.lo.data_ptr(), (ldrh if ldrh is not None else N), rs
    if row_stats is not None:
        assert tuple(row_stats.shape) == (stats_pieces(N), M, 2) and row_stats.is_contiguous(), "row_stats: [pieces][M][2]"
        g.row_stats = row_stats.data_ptr()
    if conv3x3 is not None:
        g.conv_h, g.conv_w, g.conv_c = conv3x3
        g.lda = conv3x3[2]
    if workspace is not None:
        g.workspace, g.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    _check(load().cvlm_gemm(C.byref(g), C.c_void_p(_stream())), "cvlm_gemm")


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, M: int, D: int, *,
              ldx: Optional[int] = None, add: Optional[torch.Tensor] = None, add_rows: int = 0,
              sum_out: Optional[torch.Tensor] = None, act: int = ACT_NONE, out_f32: Optional[torch.Tensor] = None,
              out_h2: Optional[H2] = None) -> None:
    _check(load().cvlm_layernorm(
        C.c_void_p(x.data_ptr()), C.c_int64(ldx if ldx is not None else D), C.c_void_p(_p(add)), C.c_int32(add_rows),
        C.c_void_p(_p(sum_out)), C.c_void_p(gamma.data_ptr()), C.c_void_p(beta.data_ptr()), C.c_float(eps),
        C.c_int32(act), C.c_void_p(_p(out_f32)), C.c_void_p(out_h2.hi.data_ptr() if out_h2 else None),
        C.c_void_p(out_h2.lo.data_ptr() if out_h2 else None), C.c_int32(M), C.c_int32(D), C.c_void_p(_stream())),
        "cvlm_layernorm")


def add_rows(a: torch.Tensor, b: Optional[torch.Tensor], b_rows: int, M: int, D: int, *, scale: float = 1.0,
             out_f32: Optional[torch.Tensor] = None, out_h2: Optional[H2] = None) -> None:
    _check(load().cvlm_add_rows(
        C.c_void_p(a.data_ptr()), C.c_void_p(_p(b)), C.c_int32(b_rows), C.c_float(scale), C.c_void_p(_p(out_f32)),
        C.c_void_p(out_h2.hi.data_ptr() if out_h2 else None), C.c_void_p(out_h2.lo.data_ptr() if out_h2 else None),
        C.c_int32(M), C.c_int32(D), C.c_void_p(_stream())), "cvlm_add_rows")


def stats_pieces(D: int) -> int:
    """Piece planes of a row-statistics buffer for rows of D columns (include/cvlm.h: one (sum, centred squares) pair per 64 columns)."""
    return (D + 63) // 64


def ln_stats_merge(pieces: torch.Tensor, M: int, D: int, eps: float, merged: torch.Tensor,
                   workspace: Optional[torch.Tensor] = None) -> None:
    """pieces f32 [ceil(D/64)][rows >= M][2] (sum, centred sum of squares per 64 columns) -> merged f32 [M][2] = (rstd, mu * rstd):
    what `gemm(ln_fold=...)` reads.  workspace: a cvlm_gemm workspace whose error word counts refused rows."""
    assert pieces.dim() == 3 and pieces.shape[0] == stats_pieces(D) and pieces.shape[1] >= M and pieces.is_contiguous()
    assert tuple(merged.shape) == (M, 2) and merged.is_contiguous()
    _check(load().cvlm_ln_stats_merge(C.c_void_p(pieces.data_ptr()), C.c_int64(pieces.shape[1]), C.c_int32(M), C.c_int32(D),
                                      C.c_float(eps), C.c_void_p(merged.data_ptr()), C.c_void_p(_p(workspace)),
                                      C.c_void_p(_stream())), "cvlm_ln_stats_merge")


def row_stats_split(x: torch.Tensor, scale: float, out: H2, stats: torch.Tensor, M: int, D: int, *, row0: int = 0,
                    copies: int = 1, dst_row_stride: int = 0) -> None:
    """out rows [row0 + c * dst_row_stride + m] = x[m] * scale as h2; stats f32 [pieces][rows][2] gets the piece statistics of the
    same rows (sum, centred sum of squares per 64 columns of the unscaled row)."""
    assert stats.dim() == 3 and stats.shape[0] == stats_pieces(D) and stats.shape[2] == 2 and stats.is_contiguous()
    if getattr(out, "mx", False):                                     # the rows as an mx operand (image + block exponents + lo plane), ABI 10
        assert out.lo is not None and out.c0 == 0 and out.C == D
        _check(load().cvlm_row_stats_split_mx(
            C.c_void_p(x.data_ptr()), C.c_float(scale), C.c_void_p(out.t.data_ptr() + row0 * out.t.stride(0)), C.c_int64(out.t.stride(0) // 2),
            C.c_void_p(out.s.data_ptr() + row0 * out.s.stride(0)), C.c_int64(out.s.stride(1)),
            C.c_void_p(out.lo.data_ptr() + 2 * row0 * out.lo.stride(0)), C.c_int64(out.lo.stride(0)), C.c_void_p(stats.data_ptr() + 8 * row0),
            C.c_int64(stats.shape[1]), C.c_int32(M), C.c_int32(D), C.c_int32(copies), C.c_int64(dst_row_stride), C.c_void_p(_stream())),
            "cvlm_row_stats_split_mx")
        return
    _check(load().cvlm_row_stats_split(C.c_void_p(x.data_ptr()), C.c_float(scale), C.c_void_p(out.hi.data_ptr() + 2 * row0 * D),
                                       C.c_void_p(out.lo.data_ptr() + 2 * row0 * D), C.c_void_p(stats.data_ptr() + 8 * row0),
                                       C.c_int64(stats.shape[1]), C.c_int32(M), C.c_int32(D), C.c_int32(copies),
                                       C.c_int64(dst_row_stride), C.c_void_p(_stream())), "cvlm_row_stats_split")


def split_f32(x: torch.Tensor, out: H2) -> None:
    _check(load().cvlm_split_f32(C.c_void_p(x.data_ptr()), C.c_void_p(out.hi.data_ptr()),
                                 C.c_void_p(out.lo.data_ptr()), C.c_int64(x.numel()), C.c_void_p(_stream())),
           "cvlm_split_f32")


def patchify(src0: torch.Tensor, src1: Optional[torch.Tensor], p: int, out: H2, ldk: int) -> None:
    B, C0, H, W = src0.shape
    C1 = 0 if src1 is None else src1.shape[1]
    _check(load().cvlm_patchify(
        C.c_void_p(src0.data_ptr()), C.c_int32(C0), C.c_void_p(_p(src1)), C.c_int32(C1), C.c_int32(B), C.c_int32(H),
        C.c_int32(W), C.c_int32(p), C.c_void_p(out.hi.data_ptr()), C.c_void_p(out.lo.data_ptr()), C.c_int32(ldk),
        C.c_void_p(_stream())), "cvlm_patchify")


def im2col3x3(x: torch.Tensor, B: int, H: int, W: int, Cc: int, out: H2) -> None:
    _check(load().cvlm_im2col3x3(C.c_void_p(x.data_ptr()), C.c_int32(B), C.c_int32(H), C.c_int32(W), C.c_int32(Cc),
                                 C.c_void_p(out.hi.data_ptr()), C.c_void_p(out.lo.data_ptr()), C.c_void_p(_stream())),
           "cvlm_im2col3x3")


def reinterpret_transpose(x: torch.Tensor, B: int, T: int, D: int, out: H2, scale: float = 1.0) -> None:
    _check(load().cvlm_reinterpret_transpose(C.c_void_p(x.data_ptr()), C.c_int32(B), C.c_int32(T), C.c_int32(D),
                                             C.c_float(scale), C.c_void_p(out.hi.data_ptr()), C.c_void_p(out.lo.data_ptr()),
                                             C.c_void_p(_stream())), "cvlm_reinterpret_transpose")


def attention(qkv: H2, out: H2, B: int, S: int, heads: int, hd: int, *, mode: int = 0, grid: int = 0, window: int = 0,
              causal: bool = False, pad: Optional[H2] = None, rel_h: Optional[H2] = None, rel_w: Optional[H2] = None,
              split_qk: int = 3, split_pv: int = 3, scale: Optional[float] = None, head_major: bool = False,
              workspace: Optional[torch.Tensor] = None, q_rows: int = 0) -> None:
    """q_rows > 0 (mode 0, ABI 9): only the first q_rows queries of every sequence (whole 128-query blocks are written).
    workspace: uint8 tensor of >= attention_workspace_bytes(...) bytes for the modes that need one; when omitted a
    temporary is taken from torch's allocator (stream-ordered, so it may be released right after the launch)."""
    _on_current_device(qkv.t)
    a = AttnArgs()
    a.qkv_hi, a.qkv_lo = qkv.hi.data_ptr(), qkv.lo.data_ptr()
    if pad is not None:
        a.pad_hi, a.pad_lo = pad.hi.data_ptr(), pad.lo.data_ptr()
    if rel_h is not None:
        a.relh_hi, a.relh_lo = rel_h.hi.data_ptr(), rel_h.lo.data_ptr()
        a.relw_hi, a.relw_lo = rel_w.hi.data_ptr(), rel_w.lo.data_ptr()
    a.out_hi, a.out_lo = out.hi.data_ptr(), out.lo.data_ptr()
    a.B, a.S, a.heads, a.hd = B, S, heads, hd
    a.mode, a.grid, a.window, a.causal = mode, grid, window, int(causal)
    a.split_qk, a.split_pv = split_qk, split_pv
    a.scale = float(hd) ** -0.5 if scale is None else scale
    a.qkv_layout = int(head_major)
    a.q_rows = q_rows
    need = int(load().cvlm_attention_workspace_bytes(C.byref(a)))
    if need > 0:
        if workspace is None or workspace.numel() * workspace.element_size() < need:
            workspace = torch.empty(need, dtype=torch.uint8, device=qkv.t.device)
        a.workspace, a.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    _check(load().cvlm_attention(C.byref(a), C.c_void_p(_stream())), "cvlm_attention")


def attention_workspace_bytes(B: int, S: int, heads: int, hd: int, *, mode: int = 0, grid: int = 0, split_qk: int = 3,
                              split_pv: int = 3, **_unused) -> int:
    """Bytes of caller-owned scratch cvlm_attention wants for these arguments (0 for most modes)."""
    a = AttnArgs()
    a.B, a.S, a.heads, a.hd, a.mode, a.grid, a.split_qk, a.split_pv = B, S, heads, hd, mode, grid, split_qk, split_pv
    return int(load().cvlm_attention_workspace_bytes(C.byref(a)))


def small_attention(q, k, v, out, B: int, nq: int, nk: int, heads: int, hd: int, out_h2: Optional[H2] = None) -> None:
    """q f32 [B*nq][>= heads*hd], k / v f32 [B*nk][>= heads*hd]: 2-D tensors (or column blocks of one: the row pitch is taken from
    `.stride(0)`); out f32 [B*nq][heads*hd] and / or out_h2 planes of that shape."""
    ld = heads * hd
    pitch = lambda t: t.stride(-2) if t.dim() >= 2 else ld
    for t in (q, k, v):
        assert t.dtype == torch.float32 and t.stride(-1) == 1
    _check(load().cvlm_small_attention_h2(
        C.c_void_p(q.data_ptr()), C.c_int64(pitch(q)), C.c_void_p(k.data_ptr()), C.c_int64(pitch(k)), C.c_void_p(v.data_ptr()),
        C.c_int64(pitch(v)), C.c_void_p(_p(out)), C.c_int64(ld), C.c_void_p(out_h2.hi.data_ptr() if out_h2 is not None else None),
        C.c_void_p(out_h2.lo.data_ptr() if out_h2 is not None else None), C.c_int64(ld), C.c_int32(B), C.c_int32(nq), C.c_int32(nk),
        C.c_int32(heads), C.c_int32(hd), C.c_void_p(_stream())), "cvlm_small_attention_h2")


def dense_pe(gauss: torch.Tensor, size: int, Cc: int, out: torch.Tensor) -> None:
    _check(load().cvlm_dense_pe(C.c_void_p(gauss.data_ptr()), C.c_int32(size), C.c_int32(Cc),
                                C.c_void_p(out.data_ptr()), C.c_void_p(_stream())), "cvlm_dense_pe")


def mask_head(up, edge_emb, hyper, B: int, HW: int, Cc: int, low) -> None:
    _check(load().cvlm_mask_head(C.c_void_p(up.data_ptr()), C.c_void_p(_p(edge_emb)),
                                 C.c_void_p(hyper.data_ptr()), C.c_int32(B), C.c_int32(HW), C.c_int32(Cc),
                                 C.c_void_p(low.data_ptr()), C.c_void_p(_stream())), "cvlm_mask_head")


def bilinear(x, N: int, hin: int, win: int, out, hout: int, wout: int, sigmoid_in: bool = False) -> None:
    _check(load().cvlm_bilinear(C.c_void_p(x.data_ptr()), C.c_int32(N), C.c_int32(hin), C.c_int32(win),
                                C.c_void_p(out.data_ptr()), C.c_int32(hout), C.c_int32(wout),
                                C.c_int32(int(sigmoid_in)), C.c_void_p(_stream())), "cvlm_bilinear")


def clip_assemble(patches, cls, pos, ctx, B: int, P: int, W: int, nctx: int, out) -> None:
    _check(load().cvlm_clip_assemble(C.c_void_p(patches.data_ptr()), C.c_void_p(cls.data_ptr()),
                                     C.c_void_p(pos.data_ptr()), C.c_void_p(ctx.data_ptr()), C.c_int32(B),
                                     C.c_int32(P), C.c_int32(W), C.c_int32(nctx), C.c_void_p(out.data_ptr()),
                                     C.c_void_p(_stream())), "cvlm_clip_assemble")


def overwrite_rows(x, B: int, L: int, W: int, first: int, n: int, src) -> None:
    _check(load().cvlm_overwrite_rows(C.c_void_p(x.data_ptr()), C.c_int32(B), C.c_int32(L), C.c_int32(W),
                                      C.c_int32(first), C.c_int32(n), C.c_void_p(src.data_ptr()),
                                      C.c_void_p(_stream())), "cvlm_overwrite_rows")


def gather_rows_h2(x: H2, scale: float, B: int, L: int, W: int, idx, fixed: int, out) -> None:
    """out[b] = (hi + lo)[b][idx[b] or fixed] * scale: one row per sequence of an h2 stream [B][L][W] as f32."""
    _check(load().cvlm_gather_rows_h2(C.c_void_p(x.hi.data_ptr()), C.c_void_p(x.lo.data_ptr()), C.c_float(scale), C.c_int32(B),
                                      C.c_int32(L), C.c_int32(W), C.c_void_p(_p(idx)), C.c_int32(fixed),
                                      C.c_void_p(out.data_ptr()), C.c_void_p(_stream())), "cvlm_gather_rows_h2")


def gather_rows(x, B: int, L: int, W: int, idx, fixed: int, out) -> None:
    _check(load().cvlm_gather_rows(C.c_void_p(x.data_ptr()), C.c_int32(B), C.c_int32(L), C.c_int32(W),
                                   C.c_void_p(_p(idx)), C.c_int32(fixed), C.c_void_p(out.data_ptr()),
                                   C.c_void_p(_stream())), "cvlm_gather_rows")


def clip_head(img, txt, logit_scale_exp: float, B: int, Cc: int, D: int, img_n, logits, pred, txt_sel) -> None:
    _check(load().cvlm_clip_head(C.c_void_p(img.data_ptr()), C.c_void_p(txt.data_ptr()), C.c_float(logit_scale_exp),
                                 C.c_int32(B), C.c_int32(Cc), C.c_int32(D), C.c_void_p(img_n.data_ptr()),
                                 C.c_void_p(logits.data_ptr()), C.c_void_p(pred.data_ptr()),
                                 C.c_void_p(txt_sel.data_ptr()), C.c_void_p(_stream())), "cvlm_clip_head")


def normalize_add(x, add, R: int, D: int, out) -> None:
    _check(load().cvlm_normalize_add(C.c_void_p(x.data_ptr()), C.c_void_p(_p(add)), C.c_int32(R), C.c_int32(D),
                                     C.c_void_p(out.data_ptr()), C.c_void_p(_stream())), "cvlm_normalize_add")


def resample_u8(src: torch.Tensor, bounds: torch.Tensor, kk: torch.Tensor, n_out: int, axis: int, dst: torch.Tensor) -> None:
    """src/dst uint8 [N][H][W][C]; bounds int32 [n_out][2]; kk int32 [n_out][ksize] (device tensors)."""
    N, H, W, Cc = src.shape
    _check(load().cvlm_resample_u8(C.c_void_p(src.data_ptr()), C.c_int32(N), C.c_int32(H), C.c_int32(W), C.c_int32(Cc),
                                   C.c_void_p(bounds.data_ptr()), C.c_void_p(kk.data_ptr()), C.c_int32(kk.shape[1]),
                                   C.c_int32(n_out), C.c_int32(axis), C.c_void_p(dst.data_ptr()), C.c_void_p(_stream())),
           "cvlm_resample_u8")


def u8_to_tensor(src: torch.Tensor, top: int, left: int, ch: int, cw: int, mean: torch.Tensor, std: torch.Tensor,
                 dst: torch.Tensor) -> None:
    N, H, W, Cc = src.shape
    _check(load().cvlm_u8_to_tensor(C.c_void_p(src.data_ptr()), C.c_int32(N), C.c_int32(H), C.c_int32(W), C.c_int32(Cc),
                                    C.c_int32(top), C.c_int32(left), C.c_int32(ch), C.c_int32(cw),
                                    C.c_void_p(mean.data_ptr()), C.c_void_p(std.data_ptr()), C.c_void_p(dst.data_ptr()),
                                    C.c_void_p(_stream())), "cvlm_u8_to_tensor")


def mask_to_u8(logits: torch.Tensor, h: int, w: int, dst: torch.Tensor) -> None:
    """logits f32 [N][Hs][Ws] -> dst uint8 [N][h][w]: sigmoid, cv2-style bilinear resize, * 255, truncate."""
    N, Hs, Ws = logits.shape
    assert logits.dtype == torch.float32 and dst.dtype == torch.uint8 and tuple(dst.shape) == (N, h, w)
    assert logits.is_contiguous() and dst.is_contiguous()
    _check(load().cvlm_mask_to_u8(C.c_void_p(logits.data_ptr()), C.c_int32(N), C.c_int32(Hs), C.c_int32(Ws), C.c_int32(h),
                                  C.c_int32(w), C.c_void_p(dst.data_ptr()), C.c_void_p(_stream())), "cvlm_mask_to_u8")


def mask_joint_hist(pre: torch.Tensor, gt: torch.Tensor, stats: torch.Tensor, hist: torch.Tensor) -> None:
    """pre/gt uint8 [N][h][w] -> stats int64 [N][3], hist int32 [N][4][2][256] (both overwritten)."""
    N, h, w = pre.shape
    assert pre.dtype == torch.uint8 and gt.dtype == torch.uint8 and tuple(gt.shape) == (N, h, w)
    assert stats.dtype == torch.int64 and tuple(stats.shape) == (N, 3) and hist.dtype == torch.int32
    assert tuple(hist.shape) == (N, 4, 2, 256) and pre.is_contiguous() and gt.is_contiguous()
    _check(load().cvlm_mask_joint_hist(C.c_void_p(pre.data_ptr()), C.c_void_p(gt.data_ptr()), C.c_int32(N), C.c_int32(h),
                                       C.c_int32(w), C.c_void_p(stats.data_ptr()), C.c_void_p(hist.data_ptr()),
                                       C.c_void_p(_stream())), "cvlm_mask_joint_hist")


def topk_accumulate(scores: torch.Tensor, labels: torch.Tensor, pred: Optional[torch.Tensor], counters: torch.Tensor) -> None:
    """scores f32 [B][C], labels int32 [B] -> pred int32 [B]; counters int32 [3] += (top-1, top-5, rows)."""
    B, Cc = scores.shape
    assert scores.dtype == torch.float32 and labels.dtype == torch.int32 and counters.dtype == torch.int32
    assert scores.is_contiguous() and labels.numel() == B and counters.numel() == 3
    _check(load().cvlm_topk_accumulate(C.c_void_p(scores.data_ptr()), C.c_void_p(labels.data_ptr()), C.c_int32(B),
                                       C.c_int32(Cc), C.c_void_p(_p(pred)), C.c_void_p(counters.data_ptr()),
                                       C.c_void_p(_stream())), "cvlm_topk_accumulate")


def mask_wfm_workspace_bytes(N: int, h: int, w: int) -> int:
    return N * h * w * 16 + N * ((h * w + 255) // 256) * 24 + N * 8


def mask_wfm(pre: torch.Tensor, gt: torch.Tensor, hist: torch.Tensor, gauss49: torch.Tensor, workspace: torch.Tensor,
             out3: torch.Tensor) -> None:
    """pre/gt uint8 [N][h][w], hist int32 [N][4][2][256] (from mask_joint_hist), gauss49 float64 [49],
    workspace uint8 [>= mask_wfm_workspace_bytes], out3 float64 [N][3]."""
    N, h, w = pre.shape
    assert pre.dtype == torch.uint8 and gt.dtype == torch.uint8 and tuple(gt.shape) == (N, h, w)
    assert hist.dtype == torch.int32 and gauss49.dtype == torch.float64 and gauss49.numel() == 49
    assert out3.dtype == torch.float64 and tuple(out3.shape) == (N, 3) and workspace.numel() * workspace.element_size() >= mask_wfm_workspace_bytes(N, h, w)
    assert pre.is_contiguous() and gt.is_contiguous() and hist.is_contiguous()
    _check(load().cvlm_mask_wfm(C.c_void_p(pre.data_ptr()), C.c_void_p(gt.data_ptr()), C.c_int32(N), C.c_int32(h), C.c_int32(w),
                                C.c_void_p(hist.data_ptr()), C.c_void_p(gauss49.data_ptr()), C.c_void_p(workspace.data_ptr()),
                                C.c_void_p(out3.data_ptr()), C.c_void_p(_stream())), "cvlm_mask_wfm")


def prob_workspace_bytes(N: int, h: int, w: int) -> int:
    """workspace of the three cvlm_prob_* calls (include/cvlm.h, ABI 8)"""
    return max(mask_wfm_workspace_bytes(N, h, w), N * 8192)


def prob_quantise(prob: torch.Tensor, minmax: torch.Tensor, q: torch.Tensor, workspace: torch.Tensor) -> None:
    """prob f32 [N][h][w] -> minmax f32 [N][2], q uint8 [N][h][w] (utils.calc_cod's `_prepare_data` + the E-measure's levels)."""
    N, h, w = prob.shape
    assert prob.dtype == torch.float32 and prob.is_contiguous() and minmax.dtype == torch.float32 and tuple(minmax.shape) == (N, 2)
    assert q.dtype == torch.uint8 and tuple(q.shape) == (N, h, w) and q.is_contiguous() and workspace.numel() >= prob_workspace_bytes(N, h, w)
    _check(load().cvlm_prob_quantise(C.c_void_p(prob.data_ptr()), C.c_int32(N), C.c_int32(h), C.c_int32(w), C.c_void_p(minmax.data_ptr()),
                                     C.c_void_p(q.data_ptr()), C.c_void_p(workspace.data_ptr()), C.c_void_p(_stream())), "cvlm_prob_quantise")


def prob_moments(prob: torch.Tensor, gt: torch.Tensor, minmax: torch.Tensor, stats: torch.Tensor, workspace: torch.Tensor,
                 out: torch.Tensor) -> None:
    """-> out f64 [N][4][2][2]: (sum pn, sum pn^2) per S-measure quadrant and ground-truth class; stats from mask_joint_hist."""
    N, h, w = prob.shape
    assert gt.dtype == torch.uint8 and tuple(gt.shape) == (N, h, w) and gt.is_contiguous() and stats.dtype == torch.int64
    assert out.dtype == torch.float64 and tuple(out.shape) == (N, 4, 2, 2) and workspace.numel() >= prob_workspace_bytes(N, h, w)
    _check(load().cvlm_prob_moments(C.c_void_p(prob.data_ptr()), C.c_void_p(gt.data_ptr()), C.c_int32(N), C.c_int32(h), C.c_int32(w),
                                    C.c_void_p(minmax.data_ptr()), C.c_void_p(stats.data_ptr()), C.c_void_p(workspace.data_ptr()),
                                    C.c_void_p(out.data_ptr()), C.c_void_p(_stream())), "cvlm_prob_moments")


def prob_wfm(prob: torch.Tensor, gt: torch.Tensor, minmax: torch.Tensor, gauss49: torch.Tensor, workspace: torch.Tensor,
             out3: torch.Tensor) -> None:
    N, h, w = prob.shape
    assert out3.dtype == torch.float64 and tuple(out3.shape) == (N, 3) and gauss49.dtype == torch.float64 and gauss49.numel() == 49
    assert workspace.numel() >= prob_workspace_bytes(N, h, w)
    _check(load().cvlm_prob_wfm(C.c_void_p(prob.data_ptr()), C.c_void_p(gt.data_ptr()), C.c_int32(N), C.c_int32(h), C.c_int32(w),
                                C.c_void_p(minmax.data_ptr()), C.c_void_p(gauss49.data_ptr()), C.c_void_p(workspace.data_ptr()),
                                C.c_void_p(out3.data_ptr()), C.c_void_p(_stream())), "cvlm_prob_wfm")
