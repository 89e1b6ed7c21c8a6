"""CPU: host logic -- C-ABI library exports, registry / state_dict contract, geometry, lowpass operator,
failure without a GPU, sharded text-bank gather over gloo (world_size 2)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import camouflaged_vlm_amd as cv
from camouflaged_vlm_amd import spec, synth

REPO = cv.REPO_DIR


def test_library_exports_every_declared_symbol():
    from camouflaged_vlm_amd import hip
    lib = hip.load()
    hdr = open(os.path.join(REPO, "include", "cvlm.h")).read()
    declared = set(re.findall(r"\b(cvlm_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"cvlm_gemm_args", "cvlm_attn_args"}
    assert len(declared) >= 19
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/cvlm.h but not exported"
    assert set(hip.EXPORTS) == declared
    assert lib.cvlm_abi_version() == hip.ABI_VERSION == 12 and lib.cvlm_target_arch() == b"gfx950"


def test_integration_doc_struct_matches_binding():
    """INTEGRATION.md prints the ctypes struct a reference maintainer would copy: it must have the layout of the real
    binding (a short struct makes the library read past its end)."""
    import ctypes
    from camouflaged_vlm_amd import hip
    doc = open(os.path.join(REPO, "INTEGRATION.md")).read()
    m = re.search(r"class cvlm_attn_args\(ctypes\.Structure\):.*?\n(?=\ndef )", doc, re.S)
    assert m, "struct block not found in INTEGRATION.md"
    ns = {"ctypes": ctypes}
    exec(m.group(0), ns)
    doc_struct = ns["cvlm_attn_args"]
    assert ctypes.sizeof(doc_struct) == ctypes.sizeof(hip.AttnArgs)
    assert [(n, getattr(doc_struct, n).offset) for n, _ in doc_struct._fields_] == \
           [(n, getattr(hip.AttnArgs, n).offset) for n, _ in hip.AttnArgs._fields_]
    # and the binding mirrors the header: every field name of the C struct, in order
    hdr = open(os.path.join(REPO, "include", "cvlm.h")).read()
    for cname, cls in (("cvlm_attn_args", hip.AttnArgs), ("cvlm_gemm_args", hip.GemmArgs)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            first, *rest = decl.split(",")
            names.append(re.findall(r"[A-Za-z_0-9]+", first)[-1])
            names += [re.findall(r"[A-Za-z_0-9]+", r)[-1] for r in rest]
        assert names == [n for n, _ in cls._fields_], cname


def test_clip_engine_dropped_when_parent_loads_weights():
    """nn.Module.load_state_dict on a parent recurses without calling the child's override; the packed
    CLIP engine must still be dropped (post hook)."""
    sys.path.insert(0, cv.DROPIN_DIR)
    from cocotrainers.mapleAlphaCLIP import CustomCLIP
    c = spec.TINY_CLIP
    clip = CustomCLIP(geometry=c, eot_train=[3] * c.n_cls_train, eot_test=[3] * c.n_cls_test)

    class Parent(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.clip_model = clip

    par = Parent()
    clip._engine, clip._engine_text_dirty = "stale", False
    par.load_state_dict(par.state_dict(), strict=True)
    assert clip._engine is None and clip._engine_text_dirty is True


def test_bad_arguments_return_error_codes_without_gpu():
    from camouflaged_vlm_amd import hip
    import ctypes as C
    lib = hip.load()
    assert lib.cvlm_gemm(None, None) == -1
    g = hip.GemmArgs()
    g.M = g.N = 16
    g.K = 20                                            # not a multiple of 32
    assert lib.cvlm_gemm(C.byref(g), None) == -1
    assert lib.cvlm_attention(None, None) == -1


def test_demo_state_dict_contract():
    e = spec.full_entries(spec.DEMO_SAM, spec.DEMO_CLIP)
    assert len(e) == 1195                                # SURVEY.md §8b [measured on the reference]
    is_buffer = lambda nme: "token_prefix" in nme or "token_suffix" in nme or nme.endswith("gaussian_matrix")
    n = sum(int(np.prod(s)) if len(s) else 1 for nme, s, _ in e if not is_buffer(nme))
    assert abs(n / 1e6 - 1039.8) < 0.1                   # parameters (buffers excluded), SURVEY.md §8b
    names = [x[0] for x in e]
    assert len(set(names)) == len(names)
    enc = sum(int(np.prod(s)) for nme, s, _ in e if nme.startswith("image_encoder."))
    assert abs(enc / 1e6 - 637.21) < 0.05
    d = dict((nme, s) for nme, s, _ in e)
    assert d["image_encoder.blocks.7.attn.rel_pos_h"] == (127, 80) and d["image_encoder.blocks.0.attn.rel_pos_h"] == (27, 80)
    assert d["clip_model.text_encoder.transformer.resblocks.0.attn.in_proj_weight"] == (2304, 768)
    assert d["clip_model.image_encoder.transformer.resblocks.0.attn.in_proj.weight"] == (3072, 1024)


def test_synthetic_weights_are_deterministic_and_order_independent():
    a = synth.make_tensor("image_encoder.blocks.3.attn.qkv.weight", (480, 160), "linear", 0)
    b = synth.make_tensor("image_encoder.blocks.3.attn.qkv.weight", (480, 160), "linear", 0)
    c = synth.make_tensor("image_encoder.blocks.3.attn.qkv.weight", (480, 160), "linear", 1)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    i1 = synth.make_inputs(spec.TINY_SAM, spec.TINY_CLIP, 2)[0]
    i2 = synth.make_inputs(spec.TINY_SAM, spec.TINY_CLIP, 1, index0=1)[0]
    assert np.array_equal(i1[1], i2[0])                  # image i of a batch == image i generated alone


def test_lowpass_operator_equals_fft_highpass():
    """|x - Re(L x L^T)| == PromptGenerator.fft (image_encoder.py:332-353) for even and non power-of-two N."""
    from camouflaged_vlm_amd.engine import lowpass_matrices
    from oracle import cvlm_oracle as O
    for N in (64, 320):
        line = int((N * N * 0.25) ** 0.5 // 2)
        lr, li = lowpass_matrices(N, line)
        x = torch.randn(2, 3, N, N, dtype=torch.float64)
        lr, li = lr.double(), li.double()
        y = (x - (lr @ x @ lr.t() - li @ x @ li.t())).abs()
        assert float((y - O.fft_highpass(x, 0.25)).abs().max()) < 1e-5


def test_registry_and_dropin_state_dict(tmp_path):
    if cv.DROPIN_DIR not in sys.path:
        sys.path.insert(0, cv.DROPIN_DIR)
    import models
    from cocotrainers.mapleAlphaCLIP import CustomCLIP
    assert set(models.models.models) >= {"sam", "sam_maskdecoder_edge"}
    g, c = spec.TINY_SAM, spec.TINY_CLIP
    enc = dict(patch_size=16, embed_dim=g.embed_dim, depth=g.depth, num_heads=g.num_heads, mlp_ratio=4, out_chans=256,
               qkv_bias=True, use_rel_pos=True, window_size=14, global_attn_indexes=[1, 3], prompt_embed_dim=256,
               scale_factor=32, freq_nums=0.25, adaptor="adaptor")           # unused YAML keys are tolerated
    m = models.make({"name": "sam_maskdecoder_edge", "args": {"inp_size": 320, "loss": "iou", "encoder_mode": enc}})
    m.load_mapleAlphaCLIP(CustomCLIP(geometry=c))
    assert set(m.state_dict()) == {n for n, _, _ in spec.full_entries(g, c)}
    sd = {k: torch.from_numpy(v) for k, v in synth.make_full_state_dict(g, c, seed=3).items()}
    m.load_state_dict(sd, strict=True)
    assert torch.equal(m.state_dict()["mask_decoder.iou_token.weight"], sd["mask_decoder.iou_token.weight"])
    bad = dict(sd)
    bad.pop("no_mask_embed.weight")
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad, strict=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):          # the product path never falls back
        m.infer_test(torch.zeros(1, 3, 320, 320), torch.zeros(1, 3, 56, 56), torch.zeros(1, 1, 56, 56))
    with pytest.raises(AssertionError):
        m.cuda if False else m.infer_test(torch.zeros(1, 3, 64, 64), None, None)   # wrong image size (image_encoder.py:375)
    # N4: the second registry name builds too (vanilla decoder), with the reference's key layout
    plain = models.make({"name": "sam", "args": {"inp_size": 320, "loss": "iou", "encoder_mode": enc}})
    assert set(plain.state_dict()) == {n for n, _, _ in spec.sam_plain_entries(g)}
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        plain.infer(torch.zeros(1, 3, 320, 320))


def test_demo_yaml_loads(tmp_path):
    """configs/demo.yaml of the reference must drive the drop-in unchanged (geometry only, no weights)."""
    import yaml
    cfg = yaml.safe_load("""
model:
  name: sam_maskdecoder_edge
  args:
    inp_size: 1024
    loss: iou
    encoder_mode: {name: sam, img_size: 1024, mlp_ratio: 4, patch_size: 16, qkv_bias: true, use_rel_pos: true,
      window_size: 14, out_chans: 256, scale_factor: 32, input_type: fft, freq_nums: 0.25, prompt_type: highpass,
      prompt_embed_dim: 256, tuning_stage: 1234, handcrafted_tune: true, embedding_tune: true, adaptor: adaptor,
      embed_dim: 1280, depth: 32, num_heads: 16, global_attn_indexes: [7, 15, 23, 31]}
""")
    g = spec.SamGeometry.from_encoder_mode(cfg["model"]["args"]["inp_size"], cfg["model"]["args"]["encoder_mode"])
    assert g == spec.DEMO_SAM and g.fft_halfwidth == 256 and g.grid == 64 and g.head_dim == 80


def test_eot_lookup():
    from camouflaged_vlm_amd import host
    c = host.ovcamo_constants()
    eot = host.eot_for_classes(c["names_test"].tolist())
    assert eot[:8] == [10, 7, 8, 7, 7, 8, 8, 7] and len(eot) == 61
    with pytest.raises(KeyError):
        host.eot_for_classes(["definitely not a class"])


_GLOO_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["CVLM_REPO"])
import camouflaged_vlm_amd as cv
sys.path.insert(0, cv.DROPIN_DIR)
from cocotrainers.mapleAlphaCLIP import gather_text_features
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
class C: embed_dim = 8
class FakeEngine:
    c, device = C, torch.device("cpu")
    def __init__(self):
        self.calls = []
    def text_features(self, eot, split, rows=None):
        idx = list(range(len(eot)))[rows] if rows is not None else list(range(len(eot)))
        self.calls.append(len(idx))
        return torch.tensor([[100.0 * i + j + eot[i] for j in range(8)] for i in idx])
eng = FakeEngine()
eot = [7 + (i % 4) for i in range(61)]
out = gather_text_features(eng, eot, "test")
ref = FakeEngine().text_features(eot, "test")
assert out.shape == (61, 8) and torch.equal(out, ref), (rank, out.shape)
assert eng.calls == [31 if rank == 0 else 30], eng.calls          # each rank encoded only its shard
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_text_bank_sharding_over_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    import socket
    with socket.socket() as sk:                                        # a port nobody holds right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, CVLM_REPO=REPO, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)


def test_openai_clip_key_mapping():
    """N3 (shape-level): OpenAI CLIP state_dict keys map onto the drop-in's keys as alpha_clip_rw/model.py:864-881 does
    (visual.* -> image_encoder.* with in_proj_weight -> in_proj.weight; text transformer under text_encoder.*)."""
    if cv.DROPIN_DIR not in sys.path:
        sys.path.insert(0, cv.DROPIN_DIR)
    from cocotrainers.mapleAlphaCLIP import CustomCLIP
    c = spec.TINY_CLIP
    W, T = c.vision_width, c.text_width
    fake = {
        "visual.conv1.weight": torch.full((W, 3, 14, 14), 0.5),
        "visual.class_embedding": torch.full((W,), 0.25),
        "visual.transformer.resblocks.0.attn.in_proj_weight": torch.full((3 * W, W), 0.125),
        "visual.transformer.resblocks.0.attn.in_proj_bias": torch.full((3 * W,), -1.0),
        "visual.proj": torch.full((W, c.embed_dim), 2.0),
        "transformer.resblocks.1.attn.in_proj_weight": torch.full((3 * T, T), 3.0),
        "transformer.resblocks.1.mlp.c_fc.bias": torch.full((4 * T,), 4.0),
        "positional_embedding": torch.full((c.context_length, T), 5.0),
        "text_projection": torch.full((T, c.embed_dim), 6.0),
        "ln_final.weight": torch.full((T,), 7.0),
        "logit_scale": torch.tensor(4.6052),
    }
    with pytest.raises(RuntimeError, match="size mismatch"):          # load_state_dict(strict=False) still rejects shapes
        CustomCLIP(geometry=c, clip_model=dict(fake, **{"visual.positional_embedding": torch.zeros(3, W)}))
    m = CustomCLIP(geometry=c, clip_model=fake)
    sd = m.state_dict()
    assert float(sd["image_encoder.conv1.weight"].mean()) == 0.5
    assert float(sd["image_encoder.transformer.resblocks.0.attn.in_proj.weight"].mean()) == 0.125
    assert float(sd["image_encoder.transformer.resblocks.0.attn.in_proj.bias"].mean()) == -1.0
    assert float(sd["image_encoder.proj"].mean()) == 2.0
    assert float(sd["text_encoder.transformer.resblocks.1.attn.in_proj_weight"].mean()) == 3.0
    assert float(sd["text_encoder.transformer.resblocks.1.mlp.c_fc.bias"].mean()) == 4.0
    assert float(sd["text_encoder.positional_embedding"].mean()) == 5.0
    assert float(sd["text_encoder.text_projection"].mean()) == 6.0 and float(sd["text_encoder.ln_final.weight"].mean()) == 7.0
    assert abs(float(sd["logit_scale"]) - 4.6052) < 1e-6
    assert float(sd["image_encoder.conv1_alpha.weight"].abs().max()) == 0     # archive without it: zero-initialised (:877-881)


def test_partial_checkpoint_loads_non_strict():
    """SAM base checkpoints are loaded with strict=False in the reference (train_...py:296-299)."""
    if cv.DROPIN_DIR not in sys.path:
        sys.path.insert(0, cv.DROPIN_DIR)
    import models
    g = spec.TINY_SAM
    enc = dict(patch_size=16, embed_dim=g.embed_dim, depth=g.depth, num_heads=g.num_heads, mlp_ratio=4, out_chans=256,
               qkv_bias=True, use_rel_pos=True, window_size=14, global_attn_indexes=[1, 3], prompt_embed_dim=256)
    m = models.make({"name": "sam_maskdecoder_edge", "args": {"inp_size": 320, "loss": "iou", "encoder_mode": enc}})
    part = {"image_encoder.patch_embed.proj.bias": torch.full((g.embed_dim,), 9.0), "unknown.key": torch.zeros(1)}
    res = m.load_state_dict(part, strict=False)
    assert "unknown.key" in res.unexpected_keys and len(res.missing_keys) > 100
    assert float(m.state_dict()["image_encoder.patch_embed.proj.bias"].mean()) == 9.0


# ---- N3: real-checkpoint formats -------------------------------------------------------------------------------------
class _NS:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _maple_cfg(c):
    return _NS(MODEL=_NS(BACKBONE=_NS(NAME="ViT-L/14@336px")),
               TRAINER=_NS(MAPLE=_NS(N_CTX=c.n_ctx, CTX_INIT="a photo of a", PREC="fp32", PROMPT_DEPTH=c.prompt_depth)),
               INPUT=_NS(SIZE=[c.image_resolution, c.image_resolution]))


def build_from_openai_archive(device=None):
    """The dropin driven the way the reference is: archive state_dict -> TestMaPLeAlphaCLIP -> models.make ->
    load_mapleAlphaCLIP -> strict=False load of everything the archive does not hold."""
    from camouflaged_vlm_amd import host
    if cv.DROPIN_DIR not in sys.path:
        sys.path.insert(0, cv.DROPIN_DIR)
    import models
    from cocotrainers.mapleAlphaCLIP import TestMaPLeAlphaCLIP
    g, c = spec.TINY_SAM, spec.TINY_CLIP
    names = host.ovcamo_constants()
    tr, te = names["names_train"].tolist()[:c.n_cls_train], names["names_test"].tolist()[:c.n_cls_test]
    osd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_openai_clip_state_dict(c).items()}
    clip = TestMaPLeAlphaCLIP(_maple_cfg(c), tr, te, clip_state_dict=osd).model
    enc = dict(name="sam", img_size=g.inp_size, mlp_ratio=4, patch_size=16, qkv_bias=True, use_rel_pos=True,
               window_size=14, out_chans=256, prompt_embed_dim=256, embed_dim=g.embed_dim, depth=g.depth,
               num_heads=g.num_heads, global_attn_indexes=list(g.global_attn_indexes))
    model = models.make({"name": "sam_maskdecoder_edge", "args": {"inp_size": g.inp_size, "loss": "iou", "encoder_mode": enc}})
    if device is not None:
        model = model.to(device)
    model.train_text_features = model.train_text_features[:c.n_cls_train]
    model.test_text_features = model.test_text_features[:c.n_cls_test]
    model.load_mapleAlphaCLIP(clip)
    res = model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.n3_rest_state_dict(g, c).items()}, strict=False)
    assert not res.unexpected_keys and all(k.startswith("clip_model.") for k in res.missing_keys)
    return model.eval(), clip


def test_openai_archive_loads_like_the_reference(golden_dir=os.path.join(REPO, "tests", "golden")):
    """N3: an OpenAI-named CLIP state_dict (packed in_proj_weight, no conv1_alpha, token_embedding table, metadata keys)
    through the dropin == the same archive through the REFERENCE's build_model + CustomCLIP (tools/make_golden.py
    --only-n3): every one of the 99 CLIP tensors by checksum, the four token buffers element-wise, the EOT columns."""
    with np.load(os.path.join(golden_dir, "n3_openai_load.npz")) as z:
        gd = {k: z[k] for k in z.files}
    model, clip = build_from_openai_archive()
    sd = clip.state_dict()
    assert sorted(sd.keys()) == gd["keys"].tolist()
    for k, (s1, s2) in zip(gd["keys"].tolist(), gd["sums"]):
        t = sd[k].double()
        assert abs(float(t.sum()) - s1) <= 1e-9 * max(1.0, abs(s1)) and abs(float(t.pow(2).sum()) - s2) <= 1e-9 * max(1.0, s2), k
    pl = "prompt_learner."
    for name in ("token_prefix", "token_suffix", "token_prefix_test", "token_suffix_test"):
        assert np.array_equal(sd[pl + name].numpy(), gd[name]), name
    assert float(sd["image_encoder.conv1_alpha.weight"].abs().max()) == 0.0        # zero-initialised alpha branch
    assert clip._eot("test") == gd["eot_test"].tolist() and clip._eot("train") == gd["eot_train"].tolist()
    # the matrix tensors carry fp16-rounded values (convert_weights), the others do not
    w = sd["image_encoder.transformer.resblocks.0.attn.in_proj.weight"]
    assert torch.equal(w, w.half().float())
    ln = sd["image_encoder.ln_pre.weight"]
    assert not torch.equal(ln, ln.half().float())


def test_openai_geometry_inference_and_renames():
    from camouflaged_vlm_amd import host
    c = spec.TINY_CLIP
    osd = synth.make_openai_clip_state_dict(c)
    a = host.clip_geometry_from_openai_state_dict(osd)
    assert (a["image_resolution"], a["patch_size"], a["vision_width"], a["vision_layers"]) == \
           (c.image_resolution, c.patch_size, c.vision_width, c.vision_layers)
    assert (a["embed_dim"], a["context_length"], a["text_width"], a["text_heads"], a["text_layers"], a["vocab_size"]) == \
           (c.embed_dim, c.context_length, c.text_width, c.text_heads, c.text_layers, 49408)
    conv = host.convert_openai_clip_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in osd.items()})
    assert "input_resolution" not in conv and "visual.transformer.resblocks.0.attn.in_proj.weight" in conv
    assert "transformer.resblocks.0.attn.in_proj_weight" in conv                 # the text tower keeps nn.MultiheadAttention names
    assert torch.equal(conv["visual.transformer.resblocks.0.attn.in_proj.weight"],
                       torch.from_numpy(osd["visual.transformer.resblocks.0.attn.in_proj_weight"]))   # element-wise, untouched
    assert tuple(conv["visual.conv1_alpha.weight"].shape) == (c.vision_width, 1, c.patch_size, c.patch_size)
    with pytest.raises(NotImplementedError):
        host.clip_geometry_from_openai_state_dict({"visual.layer1.0.conv1.weight": torch.zeros(1)})


def test_dassl_checkpoint_drops_the_fixed_token_vectors(tmp_path):
    """models/sam_maskdecoder_edge.py:192-201: a Dassl `model-best.pth.tar` ({"state_dict": ..., "epoch": ...}) updates the
    prompt learner but never `token_prefix` / `token_suffix`; the `_test` vectors and everything else in it do load."""
    model, clip = build_from_openai_archive()
    before = {k: v.clone() for k, v in clip.state_dict().items()}
    ck = {"prompt_learner.ctx": torch.full_like(before["prompt_learner.ctx"], 0.25),
          "prompt_learner.token_prefix": torch.full_like(before["prompt_learner.token_prefix"], 7.0),
          "prompt_learner.token_suffix": torch.full_like(before["prompt_learner.token_suffix"], 7.0),
          "prompt_learner.token_prefix_test": torch.full_like(before["prompt_learner.token_prefix_test"], 3.0)}
    path = str(tmp_path / "model-best.pth.tar")
    torch.save({"state_dict": ck, "epoch": 5, "val_result": 0.0}, path)
    model.load_mapleAlphaCLIP(clip, path)
    after = clip.state_dict()
    assert torch.equal(after["prompt_learner.token_prefix"], before["prompt_learner.token_prefix"])
    assert torch.equal(after["prompt_learner.token_suffix"], before["prompt_learner.token_suffix"])
    assert float(after["prompt_learner.ctx"].mean()) == 0.25 and float(after["prompt_learner.token_prefix_test"].mean()) == 3.0
    assert torch.equal(after["image_encoder.proj"], before["image_encoder.proj"])
    assert clip._engine is None and clip._engine_text_dirty


# ---- launcher: the reference's own scripts resolve `models` / `cocotrainers` / `recorder` to the drop-in ----------------
_DECOY = 'raise ImportError("the CHECKOUT\'s own %s package was imported (reference classes: mmcv / open_clip needed)")\n'
_DEMO_SHAPED = '''
import argparse
import os
os.environ["CUDA_VISIBLE_DEVICES"] = '3'                      # demo.py:3
import json
import sys
import models                                                 # demo.py:7
import recorder                                               # test_ovcos_maskdecoder_edge.py:12
from cocotrainers.mapleAlphaCLIP import TestMaPLeAlphaCLIP    # demo.py:11
from recorder.new_evaluator import Classification             # test_ovcos_maskdecoder_edge.py:18
from datasets.ovcamo_info.class_names import TRAIN_CLASS_NAMES, TEST_CLASS_NAMES   # demo.py:13: the checkout's package
import utils                                                  # test_ovcos_maskdecoder_edge.py:9: the checkout's module

if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--config', default="./configs/demo.yaml")
    args = parser.parse_args()
    print("RESULT " + json.dumps({
        "models": models.__file__, "cocotrainers": sys.modules["cocotrainers"].__file__, "recorder": recorder.__file__,
        "datasets": sys.modules["datasets"].__file__, "utils": utils.__file__, "registry": sorted(models.models.models),
        "make": callable(models.make), "register": callable(models.register), "config": args.config,
        "argv0": sys.argv[0], "n_test": len(TEST_CLASS_NAMES), "name": __name__,
        "TestMaPLeAlphaCLIP": TestMaPLeAlphaCLIP.__module__, "Classification": Classification.__module__}))
'''


def _fake_checkout(root):
    """A directory shaped like the reference checkout: its own `models`, `cocotrainers`, `recorder` packages (decoys that
    raise, standing for classes that need mmcv / open_clip), a `datasets` package and a `utils` module that must keep
    resolving to the checkout (pip's HuggingFace `datasets` is installed in this image), and a demo.py-shaped script."""
    for pkg in ("models", "cocotrainers", "recorder"):
        os.makedirs(os.path.join(root, pkg))
        with open(os.path.join(root, pkg, "__init__.py"), "w") as f:
            f.write(_DECOY % pkg)
    with open(os.path.join(root, "cocotrainers", "mapleAlphaCLIP.py"), "w") as f:
        f.write(_DECOY % "cocotrainers.mapleAlphaCLIP")
    os.makedirs(os.path.join(root, "datasets", "ovcamo_info"))
    for p in ("datasets/__init__.py", "datasets/ovcamo_info/__init__.py"):
        open(os.path.join(root, p), "w").close()
    with open(os.path.join(root, "datasets", "ovcamo_info", "class_names.py"), "w") as f:
        f.write("TRAIN_CLASS_NAMES = ['a', 'b']\nTEST_CLASS_NAMES = ['c', 'd', 'e']\n")
    with open(os.path.join(root, "utils.py"), "w") as f:
        f.write("def log(*a, **k):\n    pass\n")
    with open(os.path.join(root, "demo.py"), "w") as f:
        f.write(_DEMO_SHAPED)
    return os.path.join(root, "demo.py")


def _run_in_checkout(checkout, cmd, pythonpath):
    env = dict(os.environ, PYTHONPATH=pythonpath)
    return subprocess.run(cmd, cwd=checkout, env=env, capture_output=True, text=True, timeout=300)


def test_launcher_resolves_dropin_packages_from_the_reference_checkout(tmp_path):
    """INTEGRATION.md section 1, as documented: from the checkout, `PYTHONPATH=<repo> python -m camouflaged_vlm_amd.run
    demo.py ...` -- `models`, `cocotrainers`, `recorder` come from the drop-in although the checkout has packages of the
    same names next to the script; `datasets` / `utils` still come from the checkout; the script sees its own argv."""
    import json
    checkout = str(tmp_path / "checkout")
    os.makedirs(checkout)
    _fake_checkout(checkout)
    # the failure mode the launcher exists for: plain `python demo.py` with PYTHONPATH picks the checkout's package
    r = _run_in_checkout(checkout, [sys.executable, "demo.py"], os.pathsep.join([cv.DROPIN_DIR, REPO]))
    assert r.returncode != 0 and "CHECKOUT's own models package" in r.stderr
    r = _run_in_checkout(checkout, [sys.executable, "-m", "camouflaged_vlm_amd.run", "demo.py", "--config", "x.yaml"], REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    drop = os.path.realpath(cv.DROPIN_DIR) + os.sep
    for pkg in ("models", "cocotrainers", "recorder"):
        assert os.path.realpath(res[pkg]).startswith(drop), (pkg, res[pkg])
    for mod in ("datasets", "utils"):
        assert os.path.realpath(res[mod]).startswith(os.path.realpath(checkout) + os.sep), (mod, res[mod])
    assert res["registry"] == ["sam", "sam_maskdecoder_edge"] and res["make"] and res["register"]
    assert res["config"] == "x.yaml" and res["argv0"] == "demo.py" and res["n_test"] == 3 and res["name"] == "__main__"
    assert res["TestMaPLeAlphaCLIP"] == "cocotrainers.mapleAlphaCLIP" and res["Classification"] == "recorder.new_evaluator"


def test_launcher_forgets_reference_modules_imported_earlier(tmp_path):
    """Called as a function from a process that has already imported a foreign `models` (e.g. a notebook kernel started in
    the checkout): the stale module is dropped and the next import resolves to the drop-in."""
    import importlib
    import types
    from camouflaged_vlm_amd import run
    script = str(tmp_path / "s.py")
    open(script, "w").close()
    saved_path, saved_mods = list(sys.path), {k: v for k, v in sys.modules.items() if k.split(".")[0] in run.SHADOWED}
    try:
        for k in saved_mods:
            del sys.modules[k]
        stale = types.ModuleType("models")
        stale.__file__ = str(tmp_path / "models" / "__init__.py")
        sys.modules["models"] = stale
        merged = run.arrange_sys_path(script)
        assert merged[:3] == [cv.DROPIN_DIR, REPO, str(tmp_path)]
        assert "models" not in sys.modules
        m = importlib.import_module("models")
        assert os.path.realpath(m.__file__).startswith(os.path.realpath(cv.DROPIN_DIR))
    finally:
        sys.path[:] = saved_path
        for k in [k for k in sys.modules if k.split(".")[0] in run.SHADOWED]:
            del sys.modules[k]
        sys.modules.update(saved_mods)


def test_every_rank_draws_its_batches_from_the_reference_digest():
    """bench.py --gpus N (BASELINE configs[3]): whatever the rank and the batch size, every image of the timed batches is one the
    reference digest holds -- no rank passes its parity check on finiteness alone"""
    from camouflaged_vlm_amd import digest
    for world in (1, 2, 4, 8):
        seen = set()
        for rank in range(world):
            for B, n in ((8, 16), (4, 4), (1, 16), (3, 16)):
                ids = digest.rank_batches(rank, B, n)
                assert len(ids) == 2 and all(len(b) == B and all(0 <= i < n for i in b) for b in ids)
                if (B, n) == (8, 16):
                    assert sorted(ids[0] + ids[1]) == list(range(16))          # the two batches of a rank cover the whole digest
                    seen.add(tuple(ids[0]))
        assert len(seen) == world                                                # and the ranks' batches are different rotations
    class FakeT:                                                                 # check_* skip ids the digest lacks and say so: ok is None
        def __init__(self, a): self.a = a
        def detach(self): return self
        def float(self): return self
        def cpu(self): return self
        def numpy(self): return self.a
    import numpy as np
    dg = {"pred": np.zeros(2, np.int64), "mask_bits": np.zeros((2, 1), np.uint8), "sample_idx": np.zeros(1, np.int64),
          "mask_samples": np.zeros((2, 1), np.float32), "class_logits": np.zeros((2, 3), np.float32)}
    r = digest.check_cascade(FakeT(np.zeros((1, 1, 2, 2), np.float32)), FakeT(np.zeros(1, np.int64)), FakeT(np.zeros((1, 3), np.float32)), dg, [5])
    assert r == {"checked_images": [], "ok": None}


def test_precision_modes_and_what_one_image_per_call_runs():
    """engine.Precision: `mx` = mx GEMM operands + two-term attention products (include/cvlm.h ABI 10 / 11), on batches; one image per call
    (M <= 4096 token rows of the 64 x 64 map) keeps the three-term products of `exact`, like its GEMMs keep split-3 operands."""
    import types
    from camouflaged_vlm_amd.engine import Precision, SamEncoder
    from camouflaged_vlm_amd import host
    assert Precision.named("exact") == Precision(3, 3, 3, False)
    assert Precision.named("mx") == Precision(3, 1, 2, True) and Precision.named("mx22") == Precision(3, 2, 2, True) and Precision.named("mx33") == Precision(3, 3, 3, True)
    old = os.environ.pop("CVLM_PRECISION", None)
    try:
        assert host.precision_from_env() == Precision.named("mx")
        os.environ["CVLM_PRECISION"] = "exact"
        assert host.precision_from_env() == Precision.named("exact")
    finally:
        os.environ.pop("CVLM_PRECISION", None)
        if old is not None:
            os.environ["CVLM_PRECISION"] = old
    split = lambda name, M: SamEncoder.attn_split(types.SimpleNamespace(prec=Precision.named(name)), M)
    assert split("mx", 4096) == (3, 3) and split("mx", 8192) == (1, 2) and split("mx", 32768) == (1, 2)
    assert split("exact", 32768) == (3, 3) and split("mx33", 32768) == (3, 3) and split("fast", 4096) == (1, 1)
