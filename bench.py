#!/usr/bin/env python3
"""Headline benchmark: images/sec at 1024x1024 through the full cascade (SAM-Adapter ViT-H encoder +
edge mask decoder + MaPLe/Alpha-CLIP ViT-L/14@336, stage 1 + stage 2 classification), batch 8 per GPU,
synthetic images + deterministic synthetic weights (BASELINE.json configs[2]; configs[3] for N > 1).

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched under torch.distributed.run)

A step = one cascade pass over one resident batch of 8 images per GPU.  Prints ONE JSON line.
`roofline`   : the dominant kernel (the split-half MFMA GEMM): algorithmic FLOPs (2*M*N*K per launch)
               divided by its HIP-event time over an instrumented repeat of the timed steps.
`cpu_baseline`: the CPU oracle (oracle/cvlm_oracle.py, a port of the reference forward) timed on the
               host cores on a bounded sample (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0        # MI355X_MICROARCH.md: BF16/F16 dense ~2.5 PF
WORK_TFLOP_PER_IMAGE = 6.461               # algorithmic work per image, text bank cached


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step")
    ap.add_argument("--precision", default=os.environ.get("CVLM_PRECISION", "exact"), choices=["exact", "mixed", "fast"])
    ap.add_argument("--geometry", default="demo", choices=["demo", "tiny", "hires1536"])
    ap.add_argument("--workload", default="cascade", choices=["cascade", "encoder"],
                    help="encoder = SAM ViT-H image encoder only (BASELINE configs[1] / [4] with --geometry hires1536)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from camouflaged_vlm_amd import hip, host, spec, synth
    from camouflaged_vlm_amd.engine import Cascade, Precision
    import camouflaged_vlm_amd as cv
    sys.path.insert(0, cv.DROPIN_DIR)
    from cocotrainers.mapleAlphaCLIP import gather_text_features

    g, c = (spec.DEMO_SAM, spec.DEMO_CLIP) if args.geometry in ("demo", "hires1536") else (spec.TINY_SAM, spec.TINY_CLIP)
    if args.geometry == "hires1536":
        import dataclasses
        g = dataclasses.replace(g, inp_size=1536)            # model *built* at 1536 (pos_embed 96^2, rel_pos 191x80)
        args.workload = "encoder"
    B = args.batch
    t0 = time.time()
    sd_np = synth.make_full_state_dict(g, c)
    sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    if args.workload == "encoder":
        from camouflaged_vlm_amd.engine import SamEncoder
        enc = SamEncoder(sd, g, dev, Precision.named(args.precision))
        inp = torch.from_numpy(synth.make_inputs(g, c, batch=B, index0=rank * B)[0]).to(dev)
        for _ in range(args.warmup):
            enc.forward(inp)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            enc.forward(inp)
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        tf_img = {1024: 5.681, 1536: 13.712}.get(g.inp_size)
        print(json.dumps({"metric": f"images/sec, SAM ViT-H image encoder only at {g.inp_size}x{g.inp_size}",
                          "value": round(B * args.steps / el, 3), "unit": "images/s", "n_gpus": 1, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(1e3 * el / args.steps, 3), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
                          "config": {"workload": f"SAM ViT-H encoder only, batch {B}, {g.inp_size}^2", "precision": args.precision},
                          "achieved_tflops_algorithmic": round(B * args.steps / el * tf_img, 1) if tf_img else None}))
        return
    cas = Cascade(sd, g, c, dev, Precision.named(args.precision))
    eot = host.eot_for_classes(host.ovcamo_constants()["names_test"].tolist())[:c.n_cls_test] \
        if args.geometry == "demo" else spec.default_eot(c, "test")
    bank = torch.from_numpy(host.ovcamo_constants()["bank_test"][:c.n_cls_test]).float()
    # shared text-embedding bank: sharded over ranks + all-gathered (the only collective on the path)
    tt = time.time()
    tf = gather_text_features(cas.clip, eot, "test")
    cas.clip.set_text_bank(tf, bank, "test")
    torch.cuda.synchronize()
    text_bank_s = time.time() - tt
    inp, ci, cm = synth.make_inputs(g, c, batch=B, index0=rank * B)       # each rank: its own 8 images
    inp, ci, cm = (torch.from_numpy(t).to(dev) for t in (inp, ci, cm))
    setup_s = time.time() - t0

    def step():
        return cas.cascade(inp, ci, cm)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t1
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    value = world * B * args.steps / elapsed

    # ---- roofline of the dominant kernel (instrumented repeat; not part of `value`)
    roofline = None
    if not args.no_roofline and rank == 0:
        records = []
        orig = hip.gemm

        def timed_gemm(a, w, M, N, K, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            orig(a, w, M, N, K, **kw)
            e1.record()
            records.append((2.0 * M * N * K * kw.get("batch", 1), e0, e1))

        hip.gemm = timed_gemm
        try:
            for _ in range(max(1, min(args.steps, 2))):
                step()
            torch.cuda.synchronize()
        finally:
            hip.gemm = orig
        traffic = None
        tfile = os.path.join(REPO, "profiles", "r01_gemm_traffic.json")     # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
        if args.geometry == "demo" and args.precision == "exact" and B == 8 and os.path.exists(tfile):
            with open(tfile) as f:
                traffic = round(json.load(f)["traffic_bytes_per_launch"])
        flops = sum(r[0] for r in records)
        ms = sum(r[1].elapsed_time(r[2]) for r in records)
        achieved = flops / (ms * 1e-3) / 1e12
        roofline = {"kernel": "gemm_nt_kernel<split=%d>" % cas.prec.gemm, "bound": "mfma",
                    "achieved": round(achieved, 2), "peak": MFMA_F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / MFMA_F16_DENSE_PEAK_TFLOPS, 4), "traffic": traffic,
                    "traffic_note": "HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) and "
                                    "WRITE_SIZE passes of this command (profiles/r01_gemm_traffic.json)" if traffic else None,
                    "launches": len(records), "avg_launch_us": round(1e3 * ms / len(records), 2),
                    "algorithmic_gflop_per_launch": round(flops / len(records) / 1e9, 3),
                    "gemm_share_of_step": round(ms * 1e-3 / (max(1, min(args.steps, 2))) / (elapsed / args.steps), 3)}

    # ---- CPU baseline: the oracle on the host cores, bounded sample (rank 0, N = 1)
    cpu = None
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        from oracle import cvlm_oracle as O
        cores = min(16, os.cpu_count() or 1)          # the GPU box's CPU share for one GPU
        torch.set_num_threads(cores)
        osd = O.to_torch_sd(sd_np)
        n_img = 1
        with torch.no_grad():
            tfc = tf.cpu()
            tc = time.perf_counter()
            O.cascade(inp[:n_img].cpu(), ci[:n_img].cpu(), cm[:n_img].cpu(), osd, g, c, tfc, bank)
            cpu_s = time.perf_counter() - tc
        cpu = {"value": round(n_img / cpu_s, 5), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"{n_img} image(s), full cascade B=1, fp32 torch-CPU oracle, text bank cached "
                         f"({cpu_s:.1f} s)"}

    if rank == 0:
        line = {
            "metric": "images/sec at 1024x1024 (SAM-ViT-H + CLIP ViT-L/14), full cascade",
            "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"exact": "f32-grade (3x f16 split MFMA, f32 accumulate)", "mixed": "f32-grade GEMM/QK, f16 PV",
                      "fast": "f16 operands, f32 accumulate"}[args.precision],
            "data": "synthetic",
            "config": {"workload": "full cascade: SAM-Adapter ViT-H 1024^2 encoder + edge mask decoder + "
                                   "Alpha-CLIP ViT-L/14@336 x2 passes, 61 OVCamo prompts (BASELINE configs[2])"
                       if args.geometry == "demo" else "tiny geometry (debug)",
                       "images_per_gpu_per_step": B, "global_batch": B * world, "precision": args.precision,
                       "parallelism": f"dp{world} (images sharded, text bank all-gathered)",
                       "text_bank_seconds_once": round(text_bank_s, 3), "setup_seconds": round(setup_s, 1)},
            "achieved_tflops_algorithmic": round(value * WORK_TFLOP_PER_IMAGE, 1) if args.geometry == "demo" else None,
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()                                   # rank 0 ran the instrumented roofline repeat meanwhile
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
