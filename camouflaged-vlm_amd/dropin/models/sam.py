"""Drop-in mirror of the reference's registry entry ``sam`` (models/sam.py:298-471, class ``SAM``): the same
ViT-H adapter encoder with the *vanilla* SAM MaskDecoder (no prompts, no CLIP conditioning, no edge branch).
Same constructor arguments and state_dict keys; ``infer(input)`` / ``infer_feat(input)`` / ``get_dense_pe`` /
``postprocess_masks`` run on the MI355X HIP path (camouflaged_vlm_amd.engine.SamPlain) and there is no
PyTorch compute fallback.  Unlike the reference module this one does not import ``open_clip``: the
``ConvNeXtCLIP`` helper (models/sam.py:78-214) is never instantiated by the registry entry.

Deliberate differences: batches are B independent B=1 forwards (the reference hard-codes ``bs = 1``,
models/sam.py:418); training methods raise.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from camouflaged_vlm_amd import hip, host, spec
from camouflaged_vlm_amd.engine import Precision, SamPlain

from .models import register


@register('sam')
class SAM(nn.Module):
    def __init__(self, inp_size=None, encoder_mode=None, loss=None, *, seed: int = 0):
        super().__init__()
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.geometry = spec.SamGeometry.from_encoder_mode(inp_size, encoder_mode)
        self.embed_dim = encoder_mode['embed_dim']
        self.prompt_embed_dim = encoder_mode['prompt_embed_dim']
        self.inp_size = inp_size
        self.image_embedding_size = inp_size // encoder_mode['patch_size']
        self.loss_mode = loss
        host.populate(self, spec.sam_plain_entries(self.geometry), seed=seed)
        self.image_encoder.img_size = inp_size
        self.precision: Optional[Precision] = None
        self._engine: Optional[SamPlain] = None

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._engine = None
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._engine = None
        try:
            self.device = self.no_mask_embed.weight.device
        except AttributeError:
            pass
        return r

    def engine(self) -> SamPlain:
        dev = self.no_mask_embed.weight.device
        if dev.type != "cuda":
            raise RuntimeError("camouflaged_vlm_amd runs on MI355X only: call .cuda() first; there is no CPU fallback")
        if self._engine is None:
            self._engine = SamPlain(dict(self.state_dict()), self.geometry, dev, self.precision or host.precision_from_env())
        return self._engine

    def get_dense_pe(self) -> torch.Tensor:
        """models/sam.py:360-369 -> (1, C, h, w)."""
        G, C = self.image_embedding_size, self.prompt_embed_dim
        out = torch.empty(G * G, C, device=self.no_mask_embed.weight.device)
        hip.dense_pe(self.pe_layer.positional_encoding_gaussian_matrix, G, C, out)
        return out.reshape(G, G, C).permute(2, 0, 1).unsqueeze(0)

    def infer(self, input):
        """models/sam.py:417-440 -> (B,1,inp_size,inp_size) fp32 mask logits."""
        H, W = input.shape[-2:]
        assert H == self.inp_size and W == self.inp_size, \
            f"Input image size ({H}*{W}) doesn't match model ({self.inp_size}*{self.inp_size})."
        return self.engine().infer(input.float().contiguous())

    infer_feat = infer                                         # models/sam.py:442-465 is the same computation

    def postprocess_masks(self, masks, input_size, original_size):
        """models/sam.py:467-493 (bilinear, align_corners=False, crop, bilinear)."""
        B, C, h, w = masks.shape
        S = self.inp_size
        t = torch.empty(B * C, S, S, device=masks.device)
        hip.bilinear(masks.float().contiguous(), B * C, h, w, t, S, S)
        t = t[..., :input_size, :input_size].contiguous()
        out = torch.empty(B * C, original_size, original_size, device=masks.device)
        hip.bilinear(t, B * C, t.shape[-2], t.shape[-1], out, original_size, original_size)
        return out.reshape(B, C, original_size, original_size)

    # ---- training surface: out of scope --------------------------------
    def forward(self, *a, **k):
        raise NotImplementedError("training forward/backward is outside the MI355X inference path")

    set_input = optimize_parameters = backward_G = forward
