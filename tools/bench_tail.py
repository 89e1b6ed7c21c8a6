"""Micro-benchmark of the steps before (N1) and after (N2) the path: achieved HBM bandwidth against the ~6.3 TB/s the
chip sustains (MI355X_MICROARCH.md).  Usage: python tools/bench_tail.py"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camouflaged_vlm_amd import hip, evaltail
from camouflaged_vlm_amd.preprocess import GpuPreprocess
hip.load()

def run(name, fn, nbytes, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{name:46s} {us:8.1f} us  {nbytes / us / 1e3:7.1f} GB/s algorithmic ({nbytes / 1e6:.1f} MB)", flush=True)

rng = np.random.default_rng(0)
B = 8
# ---- N1: B camera frames 1080x1920x3 uint8 -> SAM input (1024^2 bilinear + normalise) and CLIP input (336 bicubic + crop)
pp = GpuPreprocess(1024, 336)
frames = torch.from_numpy(rng.integers(0, 256, (B, 1080, 1920, 3), dtype=np.uint8)).cuda()
run("N1 resize 1080x1920 -> 1024^2 bilinear (u8)", lambda: pp.resize(frames, 1024, 1024, "bilinear"),
    B * 3 * (1080 * 1920 + 2 * 1080 * 1024 + 1024 * 1024))
small = pp.resize(frames, 1024, 1024, "bilinear")
mean, std = pp.im_mean, pp.im_std
out = torch.empty(B, 3, 1024, 1024, device="cuda")
run("N1 ToTensor + Normalize 1024^2", lambda: hip.u8_to_tensor(small, 0, 0, 1024, 1024, mean, std, out), B * 3 * 1024 * 1024 * 5)
run("N1 resize 1080x1920 -> 336x597 bicubic (u8)", lambda: pp.resize(frames, 336, 597, "bicubic"),
    B * 3 * (1080 * 1920 + 2 * 1080 * 597 + 336 * 597))
# ---- N2: B mask logits 1024^2 f32 -> uint8 at 768x1024 -> counters
logits = torch.randn(B, 1024, 1024, device="cuda") * 4
gt = (torch.rand(B, 768, 1024, device="cuda") < 0.2).to(torch.uint8) * 255
u8 = torch.empty(B, 768, 1024, dtype=torch.uint8, device="cuda")
run("N2 sigmoid + resize 1024^2 -> 768x1024 + u8", lambda: hip.mask_to_u8(logits, 768, 1024, u8), B * (1024 * 1024 * 4 + 768 * 1024))
stats = torch.empty(B, 3, dtype=torch.int64, device="cuda"); hist = torch.empty(B, 4, 2, 256, dtype=torch.int32, device="cuda")
run("N2 centroid + joint histograms 768x1024 (noise)", lambda: hip.mask_joint_hist(u8, gt, stats, hist), B * 768 * 1024 * 3)
spike = torch.where(torch.rand(B, 768, 1024, device="cuda") < 0.9, 0, 255).to(torch.uint8)
run("N2 centroid + joint histograms (two-spike mask)", lambda: hip.mask_joint_hist(spike, gt, stats, hist), B * 768 * 1024 * 3)
# ---- N2, round 4: the weighted F-measure (blob ground truth, as camouflage masks are) and utils.calc_cod on float maps
yy, xx = torch.meshgrid(torch.arange(768, device="cuda"), torch.arange(1024, device="cuda"), indexing="ij")
blob = (((yy - 380) ** 2 + (xx - 520) ** 2) < 200 ** 2).to(torch.uint8) * 255
gtb = blob[None].repeat(B, 1, 1).contiguous()
pre = torch.clamp(gtb.float() * 0.8 + torch.randn(B, 768, 1024, device="cuda") * 25 + 30, 0, 255).to(torch.uint8)
hip.mask_joint_hist(pre, gtb, stats, hist)
# bytes the pass has to move at least: gt + pre read, near_y / d2 (4 B) + Et (8 B) written and read once
run("N2 weighted F-measure sums 768x1024 (uint8 mask)", lambda: evaltail.mask_wfm_sums(pre, gtb, hist), B * 768 * 1024 * (2 + 2 * 16))
prob = torch.sigmoid(torch.randn(B, 1024, 1024, device="cuda") * 3)
gts = (torch.rand(B, 1024, 1024, device="cuda") < 0.0).to(torch.uint8)
gts[:, 300:700, 250:800] = 255
run("N2 calc_cod counters 1024^2 (float map)", lambda: evaltail.cod_counts(prob, gts), B * 1024 * 1024 * (4 * 4 + 2 + 2 * 16))
