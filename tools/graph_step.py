"""does a hipGraph of the step move the number?  Captures one whole cascade step (both streams) with
torch.cuda.CUDAGraph and replays it; compares with the eager loop (plain stream semantics) and the pipelined serving loop."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camouflaged_vlm_amd import hip, host, spec, synth
from camouflaged_vlm_amd.engine import Cascade, Precision
import camouflaged_vlm_amd as cv
sys.path.insert(0, cv.DROPIN_DIR)
from cocotrainers.mapleAlphaCLIP import gather_text_features
g, c, B, dev = spec.DEMO_SAM, spec.DEMO_CLIP, int(os.environ.get("BATCH", "8")), torch.device("cuda", 0)
sd = {k: torch.from_numpy(v) for k, v in synth.make_full_state_dict(g, c).items()}
cas = Cascade(sd, g, c, dev, Precision.named(os.environ.get("PRECISION", "mx")))
eot = host.eot_for_classes(host.ovcamo_constants()["names_test"].tolist())[:c.n_cls_test]
bank = torch.from_numpy(host.ovcamo_constants()["bank_test"][:c.n_cls_test]).float()
cas.clip.set_text_bank(gather_text_features(cas.clip, eot, "test"), bank, "test")
inp, ci, cm = (torch.from_numpy(t).to(dev) for t in synth.make_inputs(g, c, batch=B))
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out
ms_eager, ref = timed(lambda: cas.cascade(inp, ci, cm, pipelined=False))
ms_pipe, _ = timed(lambda: cas.cascade(inp, ci, cm, pipelined=True))
cas.flush()
torch.cuda.synchronize()
s = torch.cuda.Stream(device=dev)
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): cas.cascade(inp, ci, cm, pipelined=False)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    out = cas.cascade(inp, ci, cm, pipelined=False)
torch.cuda.synchronize()
ms_graph, _ = timed(lambda: graph.replay())
ok = torch.equal(out[1], ref[1]) and float((out[0] - ref[0]).abs().max()) < 2e-4 and bool(torch.isfinite(out[0]).all())
print(f"batch {B}")
print(f"eager, plain stream semantics : {ms_eager:8.2f} ms/step")
print(f"hipGraph replay of that step  : {ms_graph:8.2f} ms/step   (outputs match eager: {ok}; hand-off errors {cas.encoder.ws.gemm_errors()})")
print(f"eager, pipelined serving loop : {ms_pipe:8.2f} ms/step   (bench.py's loop)")
