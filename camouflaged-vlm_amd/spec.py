"""Geometry + state_dict contract of the cascaded forward path.

The reference has no FFI; its weight contract is the ``state_dict()`` key layout of the
assembled model (measured through the oracle).  This module restates that
layout as a table ``[(dotted_name, shape, kind)]`` derived from the geometry alone, so the same
table drives

* the host mirror modules (``dropin/models``), which register parameters under these names,
* the synthetic weight generator (``synth.py``),
* the golden-vector script (``tools/make_golden.py``), which loads the synthetic weights into the
  reference with ``load_state_dict(strict=True)`` -- that call is the check that this table is
  key-for-key the reference's layout.

Reference sites the shapes follow:
  image encoder      models/mmseg/models/sam/image_encoder.py:25-155,218-296,383-504,628-659
  mask decoder       models/mmseg/models/sam/mask_decoder_edge.py:43-94,195-217
  two-way transformer models/mmseg/models/sam/transformer_maskdecoder_edge.py:38-60,135-162,223-238
  wrapper            models/sam_maskdecoder_edge.py:114-182
  CLIP / MaPLe       alpha_clip_rw/model.py:507-527,629-705 ; cocotrainers/mapleAlphaCLIP.py:81-168,229-238
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple


@dataclass(frozen=True)
class SamGeometry:
    inp_size: int = 1024
    patch_size: int = 16
    embed_dim: int = 1280
    depth: int = 32
    num_heads: int = 16
    mlp_ratio: float = 4.0
    out_chans: int = 256
    window_size: int = 14
    global_attn_indexes: Tuple[int, ...] = (7, 15, 23, 31)
    prompt_embed_dim: int = 256
    scale_factor: int = 32          # hard-coded image_encoder.py:116
    freq_nums: float = 0.25         # hard-coded image_encoder.py:120
    # decoder constants hard-coded in models/sam_maskdecoder_edge.py:137-148
    dec_depth: int = 2
    dec_heads: int = 8
    dec_mlp: int = 2048

    @property
    def grid(self) -> int:
        return self.inp_size // self.patch_size

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads

    @property
    def prompt_dim(self) -> int:
        return self.embed_dim // self.scale_factor

    @property
    def mlp_dim(self) -> int:
        return int(self.embed_dim * self.mlp_ratio)

    @property
    def fft_halfwidth(self) -> int:
        # image_encoder.py:337  line = int((w*h*rate) ** .5 // 2)
        return int((self.inp_size * self.inp_size * self.freq_nums) ** 0.5 // 2)

    @classmethod
    def from_encoder_mode(cls, inp_size: int, encoder_mode: dict) -> "SamGeometry":
        """Build from the YAML ``model.args`` dict (configs/demo.yaml:1-31); unknown keys ignored."""
        return cls(
            inp_size=int(inp_size),
            patch_size=int(encoder_mode["patch_size"]),
            embed_dim=int(encoder_mode["embed_dim"]),
            depth=int(encoder_mode["depth"]),
            num_heads=int(encoder_mode["num_heads"]),
            mlp_ratio=float(encoder_mode["mlp_ratio"]),
            out_chans=int(encoder_mode["out_chans"]),
            window_size=int(encoder_mode["window_size"]),
            global_attn_indexes=tuple(int(i) for i in encoder_mode["global_attn_indexes"]),
            prompt_embed_dim=int(encoder_mode["prompt_embed_dim"]),
        )


@dataclass(frozen=True)
class ClipGeometry:
    image_resolution: int = 336
    patch_size: int = 14
    vision_width: int = 1024        # MaPLe proj is hard-wired to 1024 (mapleAlphaCLIP.py:116,129)
    vision_layers: int = 24
    embed_dim: int = 768
    context_length: int = 77
    text_width: int = 768           # compound prompts hard-wired to 768 (mapleAlphaCLIP.py:124)
    text_layers: int = 12
    n_ctx: int = 4
    prompt_depth: int = 9
    n_cls_train: int = 14
    n_cls_test: int = 61

    @property
    def vision_heads(self) -> int:
        return self.vision_width // 64      # alpha_clip_rw/model.py:662

    @property
    def text_heads(self) -> int:
        return self.text_width // 64        # alpha_clip_rw/model.py:849

    @property
    def grid(self) -> int:
        return self.image_resolution // self.patch_size

    @property
    def n_tokens(self) -> int:
        return self.grid * self.grid + 1 + self.n_ctx


DEMO_SAM = SamGeometry()
DEMO_CLIP = ClipGeometry()

# Small geometry used by golden vectors / GPU parity tests: 320 px -> 20x20 grid -> padded to 28
# (2x2 windows *with* padding), head_dim stays 80, two global blocks.
TINY_SAM = SamGeometry(inp_size=320, embed_dim=160, depth=4, num_heads=2, global_attn_indexes=(1, 3))
TINY_CLIP = ClipGeometry(image_resolution=56, vision_layers=3, text_layers=3, prompt_depth=3,
                         n_cls_train=3, n_cls_test=5)

Entry = Tuple[str, Tuple[int, ...], str]


def _lin(out: List[Entry], p: str, n_out: int, n_in: int, kind: str = "linear") -> None:
    out.append((p + ".weight", (n_out, n_in), kind))
    out.append((p + ".bias", (n_out,), "bias"))


def _ln(out: List[Entry], p: str, n: int) -> None:
    out.append((p + ".weight", (n,), "ln_w"))
    out.append((p + ".bias", (n,), "ln_b"))


def sam_encoder_entries(g: SamGeometry, prefix: str = "image_encoder.") -> List[Entry]:
    D, G, P = g.embed_dim, g.grid, g.prompt_dim
    e: List[Entry] = []
    e.append((prefix + "pos_embed", (1, G, G, D), "pos"))
    e.append((prefix + "patch_embed.proj.weight", (D, 3, g.patch_size, g.patch_size), "conv"))
    e.append((prefix + "patch_embed.proj.bias", (D,), "bias"))
    for i in range(g.depth):
        b = f"{prefix}blocks.{i}."
        _ln(e, b + "norm1", D)
        rel = 2 * (G if i in g.global_attn_indexes else g.window_size) - 1
        e.append((b + "attn.rel_pos_h", (rel, g.head_dim), "relpos"))
        e.append((b + "attn.rel_pos_w", (rel, g.head_dim), "relpos"))
        _lin(e, b + "attn.qkv", 3 * D, D)
        _lin(e, b + "attn.proj", D, D)
        _ln(e, b + "norm2", D)
        _lin(e, b + "mlp.lin1", g.mlp_dim, D)
        _lin(e, b + "mlp.lin2", D, g.mlp_dim)
    e.append((prefix + "neck.0.weight", (g.out_chans, D, 1, 1), "conv"))
    _ln(e, prefix + "neck.1", g.out_chans)
    e.append((prefix + "neck.2.weight", (g.out_chans, g.out_chans, 3, 3), "conv"))
    _ln(e, prefix + "neck.3", g.out_chans)
    pg = prefix + "prompt_generator."
    _lin(e, pg + "shared_mlp", D, P)
    _lin(e, pg + "embedding_generator", P, D)
    for i in range(g.depth):
        _lin(e, f"{pg}lightweight_mlp_{i}.0", P, P)
    e.append((pg + "prompt_generator.proj.weight", (P, 3, g.patch_size, g.patch_size), "conv"))
    e.append((pg + "prompt_generator.proj.bias", (P,), "bias"))
    return e


def _dec_attn(e: List[Entry], p: str, dim: int, internal: int) -> None:
    _lin(e, p + ".q_proj", internal, dim)
    _lin(e, p + ".k_proj", internal, dim)
    _lin(e, p + ".v_proj", internal, dim)
    _lin(e, p + ".out_proj", dim, internal)


def mask_decoder_entries(g: SamGeometry, prefix: str = "mask_decoder.") -> List[Entry]:
    C = g.prompt_embed_dim
    e: List[Entry] = []
    t = prefix + "transformer."
    for i in range(g.dec_depth):
        l = f"{t}layers.{i}."
        _dec_attn(e, l + "self_attn", C, C)
        _ln(e, l + "norm1", C)
        _dec_attn(e, l + "cross_attn_token_to_image", C, C // 2)
        _ln(e, l + "norm2", C)
        _dec_attn(e, l + "cross_attn_token_to_cond", C, C // 2)
        _ln(e, l + "norm2_cond", C)
        _lin(e, l + "mlp.lin1", g.dec_mlp, C)
        _lin(e, l + "mlp.lin2", C, g.dec_mlp)
        _ln(e, l + "norm3", C)
        _ln(e, l + "norm4", C)
        _dec_attn(e, l + "cross_attn_image_to_token", C, C // 2)
        _ln(e, l + "norm4_cond", C)
        _dec_attn(e, l + "cross_attn_image_to_cond", C, C // 2)
    _dec_attn(e, t + "final_attn_token_to_image", C, C // 2)
    _ln(e, t + "norm_final_attn", C)
    e.append((prefix + "iou_token.weight", (1, C), "embed"))
    e.append((prefix + "mask_tokens.weight", (4, C), "embed"))

    def upscaler(p: str) -> None:
        e.append((p + ".0.weight", (C, C // 4, 2, 2), "convT"))
        e.append((p + ".0.bias", (C // 4,), "bias"))
        _ln(e, p + ".1", C // 4)
        e.append((p + ".3.weight", (C // 4, C // 8, 2, 2), "convT"))
        e.append((p + ".3.bias", (C // 8,), "bias"))

    upscaler(prefix + "output_upscaling")
    for i in range(4):
        m = f"{prefix}output_hypernetworks_mlps.{i}.layers."
        _lin(e, m + "0", C, C)
        _lin(e, m + "1", C, C)
        _lin(e, m + "2", C // 8, C, "hyper_tail")
    m = prefix + "iou_prediction_head.layers."
    _lin(e, m + "0", 256, C)
    _lin(e, m + "1", 256, 256)
    _lin(e, m + "2", 4, 256)
    e.append((prefix + "edge_token.weight", (1, C), "embed"))
    m = prefix + "edge_mlp.layers."
    _lin(e, m + "0", C, C)
    _lin(e, m + "1", C, C)
    _lin(e, m + "2", C // 8, C, "hyper_tail")
    upscaler(prefix + "embedding_encoder")
    mf = prefix + "embedding_maskfeature"
    e.append((mf + ".0.weight", (C // 8, C // 4, 3, 3), "convT"))
    e.append((mf + ".0.bias", (C // 4,), "bias"))
    _ln(e, mf + ".1", C // 4)
    e.append((mf + ".3.weight", (C // 4, C // 8, 3, 3), "convT"))
    e.append((mf + ".3.bias", (C // 8,), "bias"))
    return e


def vanilla_decoder_entries(g: SamGeometry, prefix: str = "mask_decoder.") -> List[Entry]:
    """Vanilla SAM MaskDecoder of the registry entry ``sam`` (models/sam.py:322-333;
    models/mmseg/models/sam/mask_decoder.py:17-71, transformer.py:17-60,109-150,185-204): depth 2, 8 heads,
    mlp 2048 hard-coded, 1 IoU + 4 mask tokens, no edge branch, no condition attentions."""
    C = g.prompt_embed_dim
    e: List[Entry] = []
    t = prefix + "transformer."
    for i in range(2):
        l = f"{t}layers.{i}."
        _dec_attn(e, l + "self_attn", C, C)
        _ln(e, l + "norm1", C)
        _dec_attn(e, l + "cross_attn_token_to_image", C, C // 2)
        _ln(e, l + "norm2", C)
        _lin(e, l + "mlp.lin1", 2048, C)
        _lin(e, l + "mlp.lin2", C, 2048)
        _ln(e, l + "norm3", C)
        _ln(e, l + "norm4", C)
        _dec_attn(e, l + "cross_attn_image_to_token", C, C // 2)
    _dec_attn(e, t + "final_attn_token_to_image", C, C // 2)
    _ln(e, t + "norm_final_attn", C)
    e.append((prefix + "iou_token.weight", (1, C), "embed"))
    e.append((prefix + "mask_tokens.weight", (4, C), "embed"))
    p = prefix + "output_upscaling"
    e.append((p + ".0.weight", (C, C // 4, 2, 2), "convT"))
    e.append((p + ".0.bias", (C // 4,), "bias"))
    _ln(e, p + ".1", C // 4)
    e.append((p + ".3.weight", (C // 4, C // 8, 2, 2), "convT"))
    e.append((p + ".3.bias", (C // 8,), "bias"))
    for i in range(4):
        m = f"{prefix}output_hypernetworks_mlps.{i}.layers."
        _lin(e, m + "0", C, C)
        _lin(e, m + "1", C, C)
        _lin(e, m + "2", C // 8, C, "hyper_tail")
    m = prefix + "iou_prediction_head.layers."
    _lin(e, m + "0", 256, C)
    _lin(e, m + "1", 256, 256)
    _lin(e, m + "2", 4, 256)
    return e


def sam_plain_entries(g: SamGeometry) -> List[Entry]:
    """state_dict of the registry entry ``sam`` (models/sam.py:298-354): encoder, vanilla decoder, PE matrix,
    no-mask embedding."""
    C = g.prompt_embed_dim
    return (sam_encoder_entries(g) + vanilla_decoder_entries(g) +
            [("pe_layer.positional_encoding_gaussian_matrix", (2, C // 2), "gauss"), ("no_mask_embed.weight", (1, C), "embed")])


def wrapper_entries(g: SamGeometry) -> List[Entry]:
    C = g.prompt_embed_dim
    e: List[Entry] = []
    e.append(("pe_layer.positional_encoding_gaussian_matrix", (2, C // 2), "gauss"))
    e.append(("no_mask_embed.weight", (1, C), "embed"))
    _ln(e, "sam_visual_proj.0", 768)
    _lin(e, "sam_visual_proj.1", C, 768)
    _ln(e, "sam_visual_proj.2", C)
    _ln(e, "sam_text_proj.0", 768)
    _lin(e, "sam_text_proj.1", C, 768)
    return e


def clip_entries(c: ClipGeometry, prefix: str = "clip_model.") -> List[Entry]:
    e: List[Entry] = []
    W, T = c.vision_width, c.text_width
    pl = prefix + "prompt_learner."
    e.append((pl + "ctx", (c.n_ctx, T), "embed"))
    e.append((pl + "token_prefix", (c.n_cls_train, 1, T), "embed"))
    e.append((pl + "token_suffix", (c.n_cls_train, c.context_length - 1 - c.n_ctx, T), "embed"))
    e.append((pl + "token_prefix_test", (c.n_cls_test, 1, T), "embed"))
    e.append((pl + "token_suffix_test", (c.n_cls_test, c.context_length - 1 - c.n_ctx, T), "embed"))
    _lin(e, pl + "proj", W, T)
    for i in range(c.prompt_depth - 1):
        e.append((f"{pl}compound_prompts_text.{i}", (c.n_ctx, T), "embed"))
    for i in range(c.prompt_depth - 1):
        _lin(e, f"{pl}compound_prompt_projections.{i}", W, T)
    ie = prefix + "image_encoder."
    e.append((ie + "class_embedding", (W,), "embed_w"))
    e.append((ie + "positional_embedding", (c.grid * c.grid + 1, W), "embed_w"))
    e.append((ie + "proj", (W, c.embed_dim), "proj_w"))
    e.append((ie + "conv1.weight", (W, 3, c.patch_size, c.patch_size), "conv"))
    e.append((ie + "conv1_alpha.weight", (W, 1, c.patch_size, c.patch_size), "conv"))
    _ln(e, ie + "ln_pre", W)
    for i in range(c.vision_layers):
        b = f"{ie}transformer.resblocks.{i}."
        _lin(e, b + "attn.in_proj", 3 * W, W)
        _lin(e, b + "attn.out_proj", W, W)
        _ln(e, b + "ln_1", W)
        _lin(e, b + "mlp.c_fc", 4 * W, W)
        _lin(e, b + "mlp.c_proj", W, 4 * W)
        _ln(e, b + "ln_2", W)
    _ln(e, ie + "ln_post", W)
    te = prefix + "text_encoder."
    for i in range(c.text_layers):
        b = f"{te}transformer.resblocks.{i}."
        e.append((b + "attn.in_proj_weight", (3 * T, T), "linear"))
        e.append((b + "attn.in_proj_bias", (3 * T,), "bias"))
        _lin(e, b + "attn.out_proj", T, T)
        _ln(e, b + "ln_1", T)
        _lin(e, b + "mlp.c_fc", 4 * T, T)
        _lin(e, b + "mlp.c_proj", T, 4 * T)
        _ln(e, b + "ln_2", T)
    e.append((te + "positional_embedding", (c.context_length, T), "embed"))
    _ln(e, te + "ln_final", T)
    e.append((te + "text_projection", (T, c.embed_dim), "proj_w"))
    e.append((prefix + "logit_scale", (), "logit_scale"))
    return e


def sam_entries(g: SamGeometry) -> List[Entry]:
    """state_dict of the wrapper *without* ``clip_model.*`` (models/sam_maskdecoder_edge.py:114-175)."""
    return sam_encoder_entries(g) + mask_decoder_entries(g) + wrapper_entries(g)


def full_entries(g: SamGeometry, c: ClipGeometry) -> List[Entry]:
    """state_dict after ``load_mapleAlphaCLIP`` (1195 tensors at demo geometry)."""
    return sam_entries(g) + clip_entries(c)


# EOT column of each tokenised prompt "a photo of a <class>." (argmax of token ids,
# cocotrainers/mapleAlphaCLIP.py:76).  Position = 1 (SOS) + 4 ("a photo of a") + n_bpe(class) + 1 (".")
# Measured by the golden script from the reference tokenizer; committed as data.
def default_eot(c: ClipGeometry, split: str) -> List[int]:
    n = c.n_cls_test if split == "test" else c.n_cls_train
    return [7 + (i % 4) for i in range(n)]
