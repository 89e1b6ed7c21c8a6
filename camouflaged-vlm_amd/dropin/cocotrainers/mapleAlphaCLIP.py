"""Drop-in mirror of the reference's ``cocotrainers.mapleAlphaCLIP`` inference surface:
``TestMaPLeAlphaCLIP(cfg, train_names, test_names).model`` -> ``CustomCLIP`` whose
``forward(image, mask, label=None, train=False)`` returns
``(image_features (B,1,768), text_features[pred] (B,1,768), pred (B,), logits (B,n_cls))``
(reference: cocotrainers/mapleAlphaCLIP.py:229-294, 478-494).

The module holds weights under the reference's state_dict keys and runs the
MI355X HIP path (camouflaged_vlm_amd.engine.ClipModel).  The MaPLe text encoder is image
independent, so it is evaluated once per weight load and cached (the reference recomputes it on
every call); with torch.distributed initialised the class prompts are sharded over the ranks and the
text features all-gathered (RCCL over xGMI) -- the only collective on the path.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from camouflaged_vlm_amd import host, spec
from camouflaged_vlm_amd.engine import ClipModel, Precision

_BACKBONES = {"ViT-L/14@336px": dict(image_resolution=336, patch_size=14, vision_width=1024, vision_layers=24,
                                     embed_dim=768, text_width=768, text_layers=12)}


def geometry_from_cfg(cfg, n_train: int, n_test: int) -> spec.ClipGeometry:
    name = cfg.MODEL.BACKBONE.NAME
    if name not in _BACKBONES:
        raise KeyError(f"unsupported CLIP backbone {name!r} (supported: {sorted(_BACKBONES)})")
    size = cfg.INPUT.SIZE[0]
    bb = _BACKBONES[name]
    assert size == bb["image_resolution"], f"cfg_imsize ({size}) must equal to clip_imsize ({bb['image_resolution']})"
    assert cfg.TRAINER.MAPLE.PROMPT_DEPTH >= 1, "For MaPLe, PROMPT_DEPTH should be >= 1"
    return spec.ClipGeometry(n_ctx=cfg.TRAINER.MAPLE.N_CTX, prompt_depth=cfg.TRAINER.MAPLE.PROMPT_DEPTH,
                             n_cls_train=n_train, n_cls_test=n_test, **bb)


def gather_text_features(engine: ClipModel, eot: Sequence[int], split: str) -> torch.Tensor:
    """Text features of all class prompts.  With N ranks each rank encodes a contiguous shard of
    ceil(n/N) prompts and the (padded) shards are all-gathered, so every rank ends up with the
    same (n, 768) tensor as a single-GPU run."""
    import torch.distributed as dist
    n = len(eot)
    if not (dist.is_available() and dist.is_initialized()):
        return engine.text_features(eot, split)
    # a process group of ONE rank still takes the collective: the launch shapes and the RCCL call are then the same code at every N
    world, rank = dist.get_world_size(), dist.get_rank()
    per = -(-n // world)
    lo, hi = min(rank * per, n), min((rank + 1) * per, n)
    D = engine.c.embed_dim
    shard = torch.zeros(per, D, device=engine.device)
    if hi > lo:
        shard[:hi - lo] = engine.text_features(eot, split, rows=slice(lo, hi))
    out = [torch.empty_like(shard) for _ in range(world)]
    dist.all_gather(out, shard)
    return torch.cat(out, 0)[:n].contiguous()


class CustomCLIP(nn.Module):
    def __init__(self, cfg=None, classnames=None, classnames_test=None, clip_model=None, *,
                 geometry: Optional[spec.ClipGeometry] = None, eot_train: Optional[Sequence[int]] = None,
                 eot_test: Optional[Sequence[int]] = None, seed: int = 0):
        super().__init__()
        if geometry is None:
            if clip_model is not None and not hasattr(clip_model, "state_dict"):
                # an OpenAI state_dict decides the architecture, as build_model does (alpha_clip_rw/model.py:825-853)
                a = host.clip_geometry_from_openai_state_dict(dict(clip_model))
                size = cfg.INPUT.SIZE[0]
                assert size == a["image_resolution"], f"cfg_imsize ({size}) must equal to clip_imsize ({a['image_resolution']})"
                assert cfg.TRAINER.MAPLE.PROMPT_DEPTH >= 1, "For MaPLe, PROMPT_DEPTH should be >= 1"
                geometry = spec.ClipGeometry(n_ctx=cfg.TRAINER.MAPLE.N_CTX, prompt_depth=cfg.TRAINER.MAPLE.PROMPT_DEPTH,
                                             n_cls_train=len(classnames), n_cls_test=len(classnames_test),
                                             **{k: a[k] for k in ("image_resolution", "patch_size", "vision_width",
                                                                  "vision_layers", "embed_dim", "context_length",
                                                                  "text_width", "text_layers")})
            else:
                geometry = geometry_from_cfg(cfg, len(classnames), len(classnames_test))
        self.geometry = geometry
        self.classnames, self.classnames_test = classnames, classnames_test
        self.eot = {"train": list(eot_train) if eot_train is not None else None,
                    "test": list(eot_test) if eot_test is not None else None}
        host.populate(self, spec.clip_entries(geometry, prefix=""), seed=seed)
        if clip_model is not None:
            self._load_from_clip(clip_model)
        self.dtype = torch.float32
        self.train_text_features = None
        self.test_text_features = None
        self.precision: Optional[Precision] = None
        self._engine: Optional[ClipModel] = None
        # weights may arrive through a PARENT's load_state_dict (demo.py:89 loads the whole SAM): nn.Module recurses
        # with _load_from_state_dict and never calls this module's load_state_dict override, but it does run the
        # post hooks of every module it visits -- the packed engine (GEMM weights, cached text features) is dropped there
        self.register_load_state_dict_post_hook(CustomCLIP._drop_engine_hook)

    @staticmethod
    def _drop_engine_hook(module, incompatible_keys) -> None:
        module._engine = None
        module._engine_text_dirty = True
        module._engine_train_dirty = True

    # ---- weights -------------------------------------------------------------------------------
    def _load_from_clip(self, clip_model, tokens_train=None, tokens_test=None) -> None:
        """Weights from a CLIP module or state_dict: OpenAI names (a JIT archive's / `clip.load`'s state_dict) or the
        reference module's names.  Restates what the reference does at construction time:
          * alpha_clip_rw/model.py:860-881: `in_proj_weight -> in_proj.weight` on the vision tower, zero `conv1_alpha`
            when the archive has none;
          * cocotrainers/mapleAlphaCLIP.py:229-238: the towers land under `image_encoder.` / `text_encoder.`,
            `logit_scale` at the top;
          * :132-168: the fixed prompt vectors `token_prefix/suffix(_test)` = rows of the token-embedding table at the
            tokenised "a photo of a <class>." prompts (the table itself is not part of the module's state_dict)."""
        from_archive = not hasattr(clip_model, "state_dict")
        sd = dict(clip_model) if from_archive else clip_model.state_dict()
        sd = host.convert_openai_clip_state_dict(sd)
        if from_archive:
            sd = host.openai_fp16_roundtrip(sd)               # what build_model + .float() leave in the reference's module
        own = dict(self.named_parameters())
        own.update(dict(self.named_buffers()))
        loaded = set()
        for k, v in sd.items():
            if k.startswith("visual."):
                nk = "image_encoder." + k[len("visual."):]
            elif k.startswith("transformer.") or k in ("positional_embedding", "text_projection") or k.startswith("ln_final."):
                nk = "text_encoder." + k
            elif k == "logit_scale":
                nk = k
            else:
                continue
            if nk not in own:
                continue
            if tuple(own[nk].shape) != tuple(v.shape) and not (own[nk].numel() == 1 and v.numel() == 1):
                raise RuntimeError(f"size mismatch for {nk}: archive {tuple(v.shape)}, model {tuple(own[nk].shape)}")
            own[nk].data.copy_(v.detach().float().reshape(own[nk].shape))
            loaded.add(nk)
        if "token_embedding.weight" in sd:
            table = sd["token_embedding.weight"].detach().float()
            n_ctx = self.geometry.n_ctx
            for split, names, toks, sfx in (("train", self.classnames, tokens_train, ""),
                                            ("test", self.classnames_test, tokens_test, "_test")):
                if toks is None:
                    if names is None:
                        continue
                    toks = host.tokens_for_classes(names)
                toks = torch.as_tensor(np.asarray(toks)).long()
                emb = table[toks]                                            # (n_cls, 77, width)
                own["prompt_learner.token_prefix" + sfx].data.copy_(emb[:, :1])
                own["prompt_learner.token_suffix" + sfx].data.copy_(emb[:, 1 + n_ctx:])
                self.eot[split] = toks.argmax(-1).tolist()                   # EOT column (mapleAlphaCLIP.py:76)
                loaded.update({"prompt_learner.token_prefix" + sfx, "prompt_learner.token_suffix" + sfx})
        self._engine = None
        self._engine_text_dirty = True
        self._engine_train_dirty = True
        self.loaded_from_clip = sorted(loaded)

    def load_text_features(self, train_text_features, test_text_features):
        self.train_text_features = train_text_features
        self.test_text_features = test_text_features
        self._engine_text_dirty = True
        self._engine_train_dirty = True

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._engine = None
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._engine = None
        return r

    # ---- engine --------------------------------------------------------------------------------
    def _eot(self, split: str):
        if self.eot[split] is None:
            names = self.classnames_test if split == "test" else self.classnames
            self.eot[split] = host.eot_for_classes(names) if names is not None else spec.default_eot(self.geometry, split)
        return self.eot[split]

    def engine(self) -> ClipModel:
        dev = self.logit_scale.device
        if dev.type != "cuda":
            raise RuntimeError("camouflaged_vlm_amd runs on MI355X only: move the model to a cuda (HIP) device; "
                               "there is no CPU fallback")
        if self._engine is None:
            prec = self.precision or host.precision_from_env()
            sd = {k: v for k, v in self.state_dict().items()}
            self._engine = ClipModel(sd, self.geometry, dev, prec, prefix="")
            self._engine_text_dirty = True
            self._engine_train_dirty = True
        if getattr(self, "_engine_text_dirty", True):
            bank = self.test_text_features
            if bank is None:
                raise RuntimeError("text features not loaded: call load_text_features(train, test) first "
                                   "(models/sam_maskdecoder_edge.py:190)")
            feat = gather_text_features(self._engine, self._eot("test"), "test")
            self._engine.set_text_bank(feat, bank, "test")
            self._engine_text_dirty = False
        return self._engine

    def forward(self, image, mask, label=None, train=False):
        """cocotrainers/mapleAlphaCLIP.py:264-294.  train=True (:267-280) is the same forward-only arithmetic on the TRAIN prompts
        (`prompt_learner()` instead of `forward_test()`) and the train bank -> logits [B][n_cls_train]; nothing on this path takes
        gradients (the training loop itself, losses and optimizer, is out of scope)."""
        eng = self.engine()
        if not train:
            return eng.forward(image.float().contiguous(), mask.float().contiguous(), "test")
        if getattr(self, "_engine_train_dirty", True) or "train" not in eng.txt:
            if self.train_text_features is None:
                raise RuntimeError("text features not loaded: call load_text_features(train, test) first "
                                   "(models/sam_maskdecoder_edge.py:190)")
            eng.set_text_bank(gather_text_features(eng, self._eot("train"), "train"), self.train_text_features, "train")
            self._engine_train_dirty = False
        return eng.forward(image.float().contiguous(), mask.float().contiguous(), "train")


class TestMaPLeAlphaCLIP(nn.Module):
    """cocotrainers/mapleAlphaCLIP.py:478-494.  The reference downloads the OpenAI archive here (:28-32); this build has
    no network: pass the archive's state_dict (or the path of a file holding it: a `torch.save`d state_dict or a
    TorchScript archive) as `clip_state_dict`, or let the weights arrive through `load_state_dict` (demo.py:88-89)."""

    def __init__(self, cfg, classnames_train, classnames_test, clip_state_dict=None):
        super().__init__()
        self.classnames_train = classnames_train
        self.classnames_test = classnames_test
        if isinstance(clip_state_dict, (str, bytes)):
            try:                                            # JIT archive first, as load_clip_to_cpu does (:34-42)
                clip_state_dict = torch.jit.load(clip_state_dict, map_location="cpu").eval().state_dict()
            except RuntimeError:
                clip_state_dict = host.load_checkpoint_state_dict(clip_state_dict)
        self.model = CustomCLIP(cfg, classnames_train, classnames_test, clip_state_dict)
