"""A/B of GEMM variants inside one process (same box, same clocks): CVLM_GEMM_VARIANT is re-read per call when
CVLM_GEMM_VARIANT_LIVE=1.  Variants other than 0 / 1 / 2 / 7 exist only in a probe build (`make -C camouflaged-vlm_amd/csrc EXTRA=-DCVLM_PROBES`).
Usage: python tools/ab_gemm.py 7 37 ..."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CVLM_GEMM_VARIANT_LIVE"] = "1"
from camouflaged_vlm_amd import hip
hip.load()
variants = [v for v in sys.argv[1:]] or ["7"]       # "7", "7:0" (variant:CVLM_GEMM_TAIL), "7:1:0" (third field: unused since round 4, was CVLM_GEMM_GROUP_M) or "1:1:0:4" (..:CVLM_GEMM_SK, split-K parts) or "1:1:0:4:3:0" (..:CVLM_GEMM_RING, LDS ring slots of the small-grid 128^2 launches:CVLM_GEMM_W8, eight waves per tile:CVLM_GEMM_COLSPLIT, one round of 256^2 tiles + the remaining columns as 128^2 tiles:1 / 0 = hand the interleaved weight image to the launch or not)
shapes = [("sam qkv", 32768, 3840, 1280), ("sam proj", 32768, 1280, 1280), ("sam lin1", 32768, 5120, 1280), ("sam lin2", 32768, 1280, 5120)]
if os.environ.get("SHAPES") == "win":          # window blocks: 8 images x 25 windows x 196 tokens
    shapes = [("win qkv", 39200, 3840, 1280), ("win proj", 39200, 1280, 1280), ("win lin1", 39200, 5120, 1280), ("win lin2", 39200, 1280, 5120)]
if os.environ.get("SHAPES") == "clip":
    shapes = [("clip in", 4648, 3072, 1024), ("clip out", 4648, 1024, 1024), ("clip fc", 4648, 4096, 1024), ("clip pj", 4648, 1024, 4096)]
if os.environ.get("SHAPES") == "clip2":        # stage 2 of batch i fused with pass 1 of batch i+1: 16 images
    shapes = [("clip in", 9296, 3072, 1024), ("clip out", 9296, 1024, 1024), ("clip fc", 9296, 4096, 1024), ("clip pj", 9296, 1024, 4096)]
if os.environ.get("SHAPES") == "b1":           # one image (the reference's call pattern): SAM blocks at M = 4096, CLIP at M = 581
    shapes = [("sam qkv", 4096, 3840, 1280), ("sam proj", 4096, 1280, 1280), ("sam lin1", 4096, 5120, 1280), ("sam lin2", 4096, 1280, 5184),
              ("clip in", 581, 3072, 1024), ("clip out", 581, 1024, 1024), ("clip fc", 581, 4096, 1024), ("clip pj", 581, 1024, 4096)]
ws = hip.new_gemm_workspace("cuda")            # variants "1:1:0:S" force S split-K parts (CVLM_GEMM_SK); "0" is the launcher's own choice
# COLD=1: every call reads a different copy of the weight (640 MB of copies, more than L2 + the 256-MB infinity cache hold) -- what a
# layer sees inside a forward pass, where the weights of 2 GB of other layers went by since its last use
cold = os.environ.get("COLD") == "1"
for name, M, N, K in shapes:
    a = hip.H2(torch.randn(2, M, K, device="cuda").half())
    ncopy = max(2, (640 << 20) // (4 * N * K)) if cold else 1
    ws_ = [hip.H2(torch.randn(2, N, K, device="cuda").half()) for _ in range(ncopy)]
    wil_ = [hip.interleave_planes(w_) for w_ in ws_]                 # the interleaved image of each copy (cvlm_gemm_args.w_il)
    calls = [0]
    def nextw():
        calls[0] += 1
        return ws_[calls[0] % ncopy]
    def curwil():
        return wil_[calls[0] % ncopy]
    out = hip.H2.empty(M, N)
    res = {v: [] for v in variants}
    for rep in range(3):
        for v in variants:
            os.environ["CVLM_GEMM_VARIANT"] = v.split(":")[0]
            f = v.split(":")
            os.environ["CVLM_GEMM_TAIL"] = f[1] if len(f) > 1 else "1"
            os.environ["CVLM_GEMM_SK"] = f[3] if len(f) > 3 else "1"
            os.environ["CVLM_GEMM_RING"] = f[4] if len(f) > 4 else "4"
            os.environ["CVLM_GEMM_W8"] = f[5] if len(f) > 5 else "1"
            os.environ["CVLM_GEMM_COLSPLIT"] = f[6] if len(f) > 6 else "1"
            use_il = (f[7] if len(f) > 7 else "1") != "0"
            hip.gemm(a, nextw(), M, N, K, out_h2=out, split=3, workspace=ws, w_il=curwil() if use_il else None)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                hip.gemm(a, nextw(), M, N, K, out_h2=out, split=3, workspace=ws, w_il=curwil() if use_il else None)
            e1.record(); torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) * 100)
    print(f"sam {name:9s} " + "  ".join(f"v{v}: {min(r):7.1f} us" for v, r in res.items()), flush=True)
