"""Per-workgroup timeline of the 256x256 staggered GEMM (CVLM_GEMM_VARIANT=47): where a tile's time goes.
Variants other than 0 / 1 / 2 / 7 exist only in a probe build (`make -C camouflaged-vlm_amd/csrc EXTRA=-DCVLM_PROBES`).
Usage: CVLM_GEMM_VARIANT=47 python tools/trace_gemm.py"""
import ctypes as C, os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camouflaged_vlm_amd import hip
lib = hip.load()
shapes = [("sam lin1", 32768, 5120, 1280), ("sam lin2", 32768, 1280, 5120), ("sam qkv", 32768, 3840, 1280)]
for name, M, N, K in shapes:
    a = hip.H2(torch.randn(2, M, K, device="cuda").half())
    w = hip.H2(torch.randn(2, N, K, device="cuda").half())
    out = hip.H2.empty(M, N)
    nwg = (M // 256) * (N // 256)
    buf = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
    lib.cvlm_debug_set_gemm_trace(C.c_void_p(buf.data_ptr()))
    for _ in range(3):
        hip.gemm(a, w, M, N, K, out_h2=out, split=3)
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(nwg, 8)
    us = lambda x: x / 100.0                       # wall_clock64: 100 MHz
    start = t[:, 0].min()
    pro, loop, epi, drain = us(t[:, 1] - t[:, 0]), us(t[:, 2] - t[:, 1]), us(t[:, 3] - t[:, 2]), us(t[:, 4] - t[:, 3])
    bias, slab0, slabs = us(t[:, 6] - t[:, 2]), us(t[:, 7] - t[:, 6]), us(t[:, 3] - t[:, 7])
    print(f"{name}: {nwg} workgroups, kernel span {us(t[:, 4].max() - start):.1f} us")
    for lab, v in (("prologue", pro), ("main loop", loop), ("epilogue issue", epi), ("  bias loads", bias), ("  slab 0", slab0),
                   ("  slabs 1..7", slabs), ("store drain", drain)):
        print(f"   {lab:15s} mean {v.mean():7.2f}  p10 {np.percentile(v, 10):7.2f}  p90 {np.percentile(v, 90):7.2f}  max {v.max():7.2f} us")
    # per-CU timelines: gap between one workgroup's end and the next one's start on the same CU
    cu = ((t[:, 5] >> 32) & 0xF) * 4096 + ((t[:, 5] >> 8) & 0xFF)            # xcc, (se, sh, cu) bits of HW_ID
    by = collections.defaultdict(list)
    for i in range(nwg):
        by[int(cu[i])].append((int(t[i, 0]), int(t[i, 4])))
    gaps, per = [], []
    for k, lst in by.items():
        lst.sort()
        per.append(len(lst))
        gaps += [us(lst[i + 1][0] - lst[i][1]) for i in range(len(lst) - 1)]
    gaps = np.asarray(gaps)
    print(f"   CUs seen {len(by)}, workgroups per CU min {min(per)} max {max(per)}; gap between workgroups mean {gaps.mean():.2f} "
          f"p90 {np.percentile(gaps, 90):.2f} max {gaps.max():.2f} us; first start spread {us(np.sort(t[:, 0])[255] - start):.2f} us")
lib.cvlm_debug_set_gemm_trace(None)
