"""Drop-in mirror of the reference wrapper ``models/sam_maskdecoder_edge.py`` (class ``SAM``,
registry name ``sam_maskdecoder_edge``): same constructor arguments, same state_dict keys, same
inference methods (``load_mapleAlphaCLIP``, ``infer_test``, ``infer``, ``postprocess_masks``,
``get_dense_pe``) and the ``clip_model`` attribute -- every operator runs on the MI355X HIP path
(camouflaged_vlm_amd.engine); there is no PyTorch compute fallback.

Differences that are deliberate and documented:
  * batched inputs are supported and are defined as B independent B=1 forwards (the reference's
    decoder only works for B=1, mask_decoder_edge.py:156-158);
  * the image-independent MaPLe text encoder is evaluated once per weight load, not per call;
  * training methods (forward/backward_G/optimize_parameters) are out of scope and raise.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from camouflaged_vlm_amd import hip, host, spec
from camouflaged_vlm_amd.engine import Cascade, Precision

from .models import register


@register('sam_maskdecoder_edge')
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
        host.populate(self, spec.sam_entries(self.geometry), seed=seed)
        self.image_encoder.img_size = inp_size
        # models/sam_maskdecoder_edge.py:177-182 (relative to CWD; packaged copy as fallback)
        self.train_text_features = host.load_text_bank("train").to(self.device)
        self.test_text_features = host.load_text_bank("test").to(self.device)
        self.precision: Optional[Precision] = None
        self._cascade: Optional[Cascade] = None

    # ---- reference call surface ------------------------------------------------------------------
    def load_mapleAlphaCLIP(self, maple_clip_model, MaPLeAlphaCLIP_checkpoint=None):
        """models/sam_maskdecoder_edge.py:184-201."""
        self.clip_model = maple_clip_model.float()
        for _, p in self.clip_model.named_parameters():
            p.requires_grad = False
        self.clip_model.to(self.device)
        self.clip_model.load_text_features(self.train_text_features, self.test_text_features)
        if MaPLeAlphaCLIP_checkpoint is not None:
            state_dict = dict(host.load_checkpoint_state_dict(MaPLeAlphaCLIP_checkpoint))   # Dassl: {"state_dict": ...}
            for k in ("prompt_learner.token_prefix", "prompt_learner.token_suffix"):
                state_dict.pop(k, None)                     # fixed token vectors are ignored (:196-199)
            self.clip_model.load_state_dict(state_dict, strict=False)
        self._cascade = None

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._cascade = None
        if hasattr(self, "clip_model"):                      # the recursion bypasses the child's override
            self.clip_model._engine = None
            self.clip_model._engine_text_dirty = True
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._cascade = None
        try:
            self.device = self.no_mask_embed.weight.device
            self.train_text_features = self.train_text_features.to(self.device)
            self.test_text_features = self.test_text_features.to(self.device)
            if hasattr(self, "clip_model"):
                self.clip_model.load_text_features(self.train_text_features, self.test_text_features)
        except AttributeError:
            pass
        return r

    def cascade(self) -> Cascade:
        if not hasattr(self, "clip_model"):
            raise RuntimeError("call load_mapleAlphaCLIP(...) first (demo.py:87)")
        dev = self.no_mask_embed.weight.device
        if dev.type != "cuda":
            raise RuntimeError("camouflaged_vlm_amd runs on MI355X only: call .cuda() first; there is no CPU fallback")
        clip_engine = self.clip_model.engine()
        if self._cascade is None or self._cascade.clip is not clip_engine:
            prec = self.precision or host.precision_from_env()
            sd = {k: v for k, v in self.state_dict().items() if not k.startswith("clip_model.")}
            self._cascade = Cascade(sd, self.geometry, self.clip_model.geometry, dev, prec, clip=clip_engine)
        return self._cascade

    def get_dense_pe(self) -> torch.Tensor:
        """:210-219 -> (1, C, h, w)."""
        G, C = self.image_embedding_size, self.prompt_embed_dim
        out = torch.empty(G * G, C, device=self.no_mask_embed.weight.device)
        hip.dense_pe(self.pe_layer.positional_encoding_gaussian_matrix, G, C, out)
        return out.reshape(G, G, C).permute(2, 0, 1).unsqueeze(0)

    def maple_alpha_clip_process(self, image, alpha):
        """:268-270 (``self.training`` lands in ``label``: always the test branch, Appendix B.2)."""
        return self.clip_model(image, alpha, self.training)

    def infer_test(self, input, clip_image, clip_zero_mask):
        """:331-357 -> (B,1,inp_size,inp_size) fp32 mask logits."""
        H, W = input.shape[-2:]
        assert H == self.inp_size and W == self.inp_size, \
            f"Input image size ({H}*{W}) doesn't match model ({self.inp_size}*{self.inp_size})."
        return self.cascade().infer_test(input.float().contiguous(), clip_image.float().contiguous(),
                                         clip_zero_mask.float().contiguous())

    def infer(self, input, clip_image, clip_zero_mask):
        """:305-329 (bs = 1 variant of infer_test)."""
        return self.infer_test(input, clip_image, clip_zero_mask)

    def postprocess_masks(self, masks, input_size, original_size):
        """:359-388 (bilinear, align_corners=False, twice)."""
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
