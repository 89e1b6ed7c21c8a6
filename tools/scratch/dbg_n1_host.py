import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from camouflaged_vlm_amd.preprocess import GpuPreprocess
dev = torch.device("cuda:0")
pre = GpuPreprocess(1024, 336, dev)
sizes = [(768, 1024), (1080, 1920), (683, 1024), (1024, 1024), (1365, 2048), (600, 800), (1024, 683), (960, 1280)]
host = [torch.randint(0, 255, (h, w, 3), dtype=torch.uint8).pin_memory() for h, w in sizes]
def step():
    imgs = [t.to(dev, non_blocking=True) for t in host]
    t1 = time.perf_counter()
    inp = torch.cat([pre.sam_input(t) for t in imgs])
    t2 = time.perf_counter()
    ci = torch.cat([pre.clip_input(t) for t in imgs])
    t3 = time.perf_counter()
    return t1, t2, t3
for _ in range(3):
    step()
torch.cuda.synchronize()
st0 = torch.cuda.memory_stats()
for it in range(4):
    t0 = time.perf_counter()
    t1, t2, t3 = step()
    print(f"iter {it}: h2d issue {1e3*(t1-t0):.2f} ms, sam_input x8 + cat {1e3*(t2-t1):.2f} ms, clip_input x8 + cat {1e3*(t3-t2):.2f} ms")
torch.cuda.synchronize()
st1 = torch.cuda.memory_stats()
for k in ("num_device_alloc", "num_device_free", "num_alloc_retries", "allocation.all.allocated"):
    print(k, st0.get(k), "->", st1.get(k))
# per call timing
imgs = [t.to(dev) for t in host]
for t in imgs:
    a = time.perf_counter(); pre.sam_input(t); b = time.perf_counter(); pre.clip_input(t); c = time.perf_counter()
    print(tuple(t.shape), f"sam_input {1e3*(b-a):.3f} ms clip_input {1e3*(c-b):.3f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for t in imgs: pre.sam_input(t)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
