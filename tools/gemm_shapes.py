"""Per-shape time of every C-ABI launch kind inside one cascade step (B = 8, demo geometry): HIP events around each
hip.* call, aggregated by (op, shape).  Usage: python tools/gemm_shapes.py [--batch 8] [--reps 2]"""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from camouflaged_vlm_amd import hip, host, spec, synth
from camouflaged_vlm_amd.engine import Cascade, Precision

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--precision", default=os.environ.get("CVLM_PRECISION", "mx"))
ap.add_argument("--pipelined", action="store_true", help="the loop bench.py times: stage 2 of a batch fused with the next batch's CLIP pass 1 (one stream here)")
a = ap.parse_args()
g, c = spec.DEMO_SAM, spec.DEMO_CLIP
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(v) for k, v in synth.make_full_state_dict(g, c).items()}
cas = Cascade(sd, g, c, dev, Precision.named(a.precision))
cas.overlap_clip = False                        # one stream: per-launch event times need the kernels one at a time
eot = host.eot_for_classes(host.ovcamo_constants()["names_test"].tolist())[:c.n_cls_test]
cas.clip.set_text_bank(cas.clip.text_features(eot, "test"), torch.from_numpy(host.ovcamo_constants()["bank_test"]).float(), "test")
inp, ci, cm = (torch.from_numpy(t).to(dev) for t in synth.make_inputs(g, c, batch=a.batch))
for _ in range(2):
    cas.cascade(inp, ci, cm, pipelined=a.pipelined)
torch.cuda.synchronize()
recs = []
names = ["gemm", "layernorm", "add_rows", "split_f32", "patchify", "im2col3x3", "reinterpret_transpose", "attention",
         "small_attention", "mask_head", "bilinear", "clip_assemble", "overwrite_rows", "gather_rows", "clip_head"]
orig = {n: getattr(hip, n) for n in names}


def wrap(n):
    f = orig[n]

    def w(*args, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        f(*args, **kw)
        e1.record()
        if n == "gemm":
            M, N, K = args[2], args[3], args[4]
            key = (n, M, N, K, kw.get("batch", 1), "res" if kw.get("residual") is not None else "", "f32" if kw.get("out_f32") is not None else "h2")
            fl = 2.0 * M * N * K * kw.get("batch", 1)
        elif n == "attention":
            key = (n, "mode%d" % kw.get("mode", 0), args[2], args[3], args[4], args[5])
            fl = 0.0
        elif n == "layernorm":
            key = (n, args[4], args[5])
            fl = 0.0
        else:
            key = (n,)
            fl = 0.0
        recs.append((key, fl, e0, e1))
    return w


for n in names:
    setattr(hip, n, wrap(n))
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(a.reps):
    cas.cascade(inp, ci, cm, pipelined=a.pipelined)
t1.record()
for n in names:                                                      # the flush (stage 2 of the last batch alone) is not part of a steady-state step
    setattr(hip, n, orig[n])
cas.flush()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, fl, e0, e1 in recs:
    d = agg.setdefault(key, [0, 0.0, 0.0])
    d[0] += 1
    d[1] += e0.elapsed_time(e1)
    d[2] += fl
tot = sum(d[1] for d in agg.values())
print(f"step (instrumented) {t0.elapsed_time(t1) / a.reps:.2f} ms; sum of launches {tot / a.reps:.2f} ms")
for key, d in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tf = d[2] / (d[1] * 1e-3) / 1e12 if d[2] else 0.0
    print(f"{d[1] / a.reps:9.3f} ms/step  {d[0] // a.reps:5d} launches  {1e3 * d[1] / d[0]:9.1f} us each  {tf:7.1f} TF/s  {key}")
