import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from camouflaged_vlm_amd.preprocess import GpuPreprocess
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
for mb in (1, 6, 32):
    a = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    b = torch.empty(mb << 20, dtype=torch.uint8)
    for name, src in (("pinned", a), ("pageable", b)):
        for _ in range(2):
            d = src.to(dev, non_blocking=True)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            d = src.to(dev, non_blocking=True)
        t_issue = time.perf_counter() - t
        torch.cuda.synchronize()
        t = time.perf_counter() - t
        print(f"{name} {mb} MB: issue {1e3*t_issue/10:.3f} ms, done {1e3*t/10:.3f} ms -> {mb/1024/(t/10):.1f} GB/s")
pre = GpuPreprocess(1024, 336, dev)
img = torch.randint(0, 255, (1080, 1920, 3), dtype=torch.uint8, device=dev)
for _ in range(3):
    pre.sam_input(img); pre.clip_input(img)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t = time.perf_counter()
e0.record()
for _ in range(20):
    a = pre.sam_input(img); b = pre.clip_input(img)
e1.record()
t_host = time.perf_counter() - t
torch.cuda.synchronize()
print(f"N1 one image 1080x1920: GPU {e0.elapsed_time(e1)/20:.3f} ms, host issue {1e3*t_host/20:.3f} ms")
