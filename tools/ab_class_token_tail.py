"""A/B on one box: the CLIP vision tower's last block behind its qkv projection for the class tokens only (engine attribute
class_token_tail, cvlm_attn_args.q_rows) against the whole block.  Whole cascade, B = 1 and B = 8, call by call and pipelined.
python tools/ab_class_token_tail.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from camouflaged_vlm_amd import spec, synth, host
from camouflaged_vlm_amd.engine import Cascade, Precision

g, c = spec.DEMO_SAM, spec.DEMO_CLIP
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(v) for k, v in synth.make_full_state_dict(g, c).items()}
cas = Cascade(sd, g, c, dev, Precision.named("exact"))
del sd
eot = host.eot_for_classes(host.ovcamo_constants()["names_test"].tolist())[:c.n_cls_test]
cas.clip.set_text_bank(cas.clip.text_features(eot, "test"), torch.from_numpy(host.ovcamo_constants()["bank_test"]).float(), "test")

for B, pipelined in ((1, False), (8, False), (8, True), (1, True)):
    inp, ci, cm = (torch.from_numpy(t).to(dev) for t in synth.make_inputs(g, c, batch=B))
    res = {}
    for rep in range(3):
        for flag in (False, True):
            cas.clip.class_token_tail = flag
            for _ in range(3):
                m = cas.cascade(inp, ci, cm, pipelined=pipelined)
            if pipelined:
                cas.flush()
            torch.cuda.synchronize()
            n = 16 if B == 1 else 5
            t0 = time.perf_counter()
            for _ in range(n):
                m = cas.cascade(inp, ci, cm, pipelined=pipelined)
                if not pipelined:
                    torch.cuda.synchronize()
            if pipelined:
                cas.flush()
            torch.cuda.synchronize()
            res.setdefault(flag, []).append(1e3 * (time.perf_counter() - t0) / n)
            res[("out", flag)] = (m[0].clone(), m[2].clone())
    dm = float((res[("out", False)][0] - res[("out", True)][0]).abs().max())
    dl = float((res[("out", False)][1] - res[("out", True)][1]).abs().max())
    print(f"B={B} pipelined={pipelined}: whole last block {min(res[False]):.3f} ms/step, class tokens only {min(res[True]):.3f} ms/step "
          f"(all: {[round(x, 2) for x in res[False]]} vs {[round(x, 2) for x in res[True]]}); max |mask diff| {dm:.1e}, |class logit diff| {dl:.1e}", flush=True)
cas.clip.class_token_tail = True
