import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from camouflaged_vlm_amd import spec, synth, hip
from camouflaged_vlm_amd.engine import Cascade, Precision
g, c = spec.TINY_SAM, spec.TINY_CLIP
dev = torch.device("cuda:0")
gold = np.load("tests/golden/tiny_cascade.npz")
inp, ci, cm = (torch.from_numpy(t).to(dev) for t in synth.make_inputs(g, c, batch=2))
for off in (0.0, 400.0, 4000.0):
    sd_np = synth.make_full_state_dict(g, c)
    sd_np["image_encoder.pos_embed"] = sd_np["image_encoder.pos_embed"] + off
    sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    cas = Cascade(sd, g, c, dev, Precision.named("exact"))
    cas.clip.set_text_bank(cas.clip.text_features(gold["eot_test"].tolist(), "test"), torch.from_numpy(gold["bank_test"]), "test")
    m = cas.infer_test(inp, ci, cm)
    torch.cuda.synchronize()
    mrg = cas.encoder.ws.f32("ln_merged", 2 * g.grid * g.grid, 2)
    print(off, "nan frac", float(torch.isnan(m).float().mean()), "errors", cas.encoder.ws.gemm_errors(), "merged", mrg[:2].tolist(),
          "folded" , cas.encoder.ln_fold)
