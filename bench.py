#!/usr/bin/env python3
"""Headline benchmark: images/sec at 1024x1024 through the full cascade (SAM-Adapter ViT-H encoder +
edge mask decoder + MaPLe/Alpha-CLIP ViT-L/14@336, stage 1 + stage 2 classification), batch 8 per GPU,
synthetic images + deterministic synthetic weights (BASELINE.json configs[2]; configs[3] for N > 1).

  python bench.py --gpus N --steps K --warmup W

N > 1 without RANK in the environment: this process starts N ranks itself (`python -m torch.distributed.run`,
as a child, before anything here has touched the GPU), relays rank 0's JSON line and exits with the child's
code.  Under a launcher (RANK / WORLD_SIZE set) it is one of the ranks.

A step = one cascade pass over one resident batch of 8 images per GPU.  Rank 0 prints ONE JSON line.
`value`       : whole-job images/s over the K timed steps (wall clock, barrier + synchronize on both sides, max over ranks);
                `step_ms` holds the median / p10 / p90 of the per-step HIP-event durations of the same K steps.
`parity`      : the outputs of the LAST TIMED step are checked: all finite; image 0 of rank 0 against the digest of the
                reference's own output (tests/golden/demo_digest.npz): IoU >= 0.999, |mask| and |class logits| <= 1e-3.
`bank_check`  : N > 1: the all-gathered text bank equals, bit for bit, the bank every rank computes alone.
`roofline`    : the dominant kernel (the split-half MFMA GEMM): algorithmic FLOPs (2*M*N*K per launch) divided by its
                HIP-event time over an instrumented repeat of the timed steps; `secondary` holds the two ViT-H attention
                kernels measured the same way (algorithmic 4*S^2*hd per head).
`cpu_baseline`: the CPU oracle (oracle/cvlm_oracle.py, a port of the reference forward) timed on the host cores on a
                bounded sample (rank 0, N = 1 only): 1 warm-up + 3 images on the box's 16-core share.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0        # MI355X_MICROARCH.md: BF16/F16 dense ~2.5 PF
WORK_TFLOP_PER_IMAGE = 6.461               # algorithmic work per image, text bank cached
TOL = 1e-3                                 # BASELINE.json north_star: 1e-3 abs vs the fp32 CPU forward, IoU >= 0.999
# Measured on this part (tools/power_roofline.py, profiles/r02_power_roofline.log): an MFMA-only loop (16x16x32 f16, random
# operands, every CU) settles at 1.88 PF at the 1400-W socket cap (sclk 1.95 GHz of 2.4): what the matrix pipe can sustain.
MFMA_F16_POWER_ROOFLINE_TFLOPS = 1880.0


class PowerSampler:
    """Socket power and shader clock during the timed region: a CHILD process polls rocm-smi (this process only reads its
    output afterwards).  Everything is optional: no rocm-smi, no samples, no field."""

    def __init__(self):
        self.proc, self.path = None, None

    def start(self):
        import shutil, tempfile
        if shutil.which("rocm-smi") is None:
            return
        fd, self.path = tempfile.mkstemp(prefix="cvlm_power_", suffix=".log")
        os.close(fd)
        try:
            self.proc = subprocess.Popen(
                ["bash", "-c", "while true; do echo STAMP $(date +%s.%N); rocm-smi --showpower --showclocks; sleep 0.25; done"],
                stdout=open(self.path, "w"), stderr=subprocess.DEVNULL, start_new_session=True)
        except OSError:
            self.proc = None

    def stop(self, t0: float, t1: float, device_index: int = 0):
        import re, signal
        if self.proc is None:
            return None
        try:
            os.killpg(self.proc.pid, signal.SIGTERM)            # exactly the group started above
            self.proc.wait(timeout=5)
        except Exception:
            pass
        try:
            with open(self.path) as f:
                text = f.read()
            os.unlink(self.path)
        except OSError:
            return None
        pw, sc = [], []
        for chunk in text.split("STAMP ")[1:]:
            try:
                ts = float(chunk.split(None, 1)[0])
            except (ValueError, IndexError):
                continue
            if not (t0 <= ts <= t1):
                continue
            p = re.search(r"GPU\[%d\]\s*: .*Power \(W\): ([0-9.]+)" % device_index, chunk)
            c = re.search(r"GPU\[%d\]\s*: sclk clock level: \d+: \((\d+)Mhz\)" % device_index, chunk)
            if p and c:
                pw.append(float(p.group(1)))
                sc.append(int(c.group(1)))
        if not pw:
            return None
        return {"socket_w_mean": round(sum(pw) / len(pw), 1), "socket_w_max": max(pw), "sclk_mhz_mean": round(sum(sc) / len(sc)),
                "sclk_mhz_max_of_part": 2400, "samples": len(pw),
                "note": "rocm-smi polled by a child process during the timed steps (rank 0's GPU)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step")
    ap.add_argument("--precision", default=os.environ.get("CVLM_PRECISION", "exact"), choices=["exact", "mixed", "fast"])
    ap.add_argument("--geometry", default="demo", choices=["demo", "tiny", "hires1536"])
    ap.add_argument("--workload", default="cascade", choices=["cascade", "encoder"],
                    help="encoder = SAM ViT-H image encoder only (BASELINE configs[1] / [4] with --geometry hires1536)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run CLIP pass 1 after the SAM encoder instead of on a side stream beneath it (profiling runs: "
                         "co-running kernels stretch each other's durations)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --single-device: rehearse the N-rank path on a box with ONE GPU (all ranks on cuda:0)")
    ap.add_argument("--single-device", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def spawn_ranks(args) -> int:
    """Start `--gpus` ranks as a CHILD process (this process has not touched the GPU) and relay its output."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def percentile(xs, q):
    xs = sorted(xs)
    if not xs:
        return None
    pos = (len(xs) - 1) * q
    lo = int(pos)
    hi = min(lo + 1, len(xs) - 1)
    return xs[lo] + (xs[hi] - xs[lo]) * (pos - lo)


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.single_device:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
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
            out = enc.forward(inp)
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        finite = bool(torch.isfinite(out).all())
        tf_img = {1024: 5.681, 1536: 13.712}.get(g.inp_size)
        print(json.dumps({"metric": f"images/sec, SAM ViT-H image encoder only at {g.inp_size}x{g.inp_size}",
                          "value": round(B * args.steps / el, 3), "unit": "images/s", "n_gpus": 1, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(1e3 * el / args.steps, 3), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
                          "config": {"workload": f"SAM ViT-H encoder only, batch {B}, {g.inp_size}^2", "precision": args.precision},
                          "outputs_finite": finite,
                          "achieved_tflops_algorithmic": round(B * args.steps / el * tf_img, 1) if tf_img else None}))
        if not finite:
            sys.exit(3)
        return
    if args.no_overlap:
        os.environ["CVLM_OVERLAP_CLIP"] = "0"
    cas = Cascade(sd, g, c, dev, Precision.named(args.precision))
    eot = host.eot_for_classes(host.ovcamo_constants()["names_test"].tolist())[:c.n_cls_test] \
        if args.geometry == "demo" else spec.default_eot(c, "test")
    bank = torch.from_numpy(host.ovcamo_constants()["bank_test"][:c.n_cls_test]).float()
    # shared text-embedding bank: sharded over ranks + all-gathered (the only collective on the path)
    tt = time.time()
    tf = gather_text_features(cas.clip, eot, "test")
    torch.cuda.synchronize()
    text_bank_s = time.time() - tt
    bank_check = None
    if world > 1:
        # the gathered bank is bit-identical on every rank and equal to the single-GPU bank
        alone = cas.clip.text_features(eot, "test")
        same = torch.tensor([int(torch.equal(alone, tf))], device=dev)
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        bank_check = {"bit_identical_to_single_rank_bank_on_every_rank": bool(int(same.item())),
                      "max_abs_diff_rank0": float((alone - tf).abs().max())}
    cas.clip.set_text_bank(tf, bank, "test")
    inp, ci, cm = synth.make_inputs(g, c, batch=B, index0=rank * B)       # each rank: its own 8 images
    inp, ci, cm = (torch.from_numpy(t).to(dev) for t in (inp, ci, cm))
    setup_s = time.time() - t0

    def step():
        # serving loop: batch i's stage 2 (on the side stream) runs under batch i+1's SAM encoder; everything is
        # complete at the synchronize() that closes the timed region
        return cas.cascade(inp, ci, cm, pipelined=True)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    sampler = PowerSampler()
    if rank == 0:
        sampler.start()
    wall0 = time.time()
    t1 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        out = step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t1
    power = sampler.stop(wall0 + 0.3, time.time(), local_rank) if rank == 0 else None
    my_elapsed = elapsed
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    value = world * B * args.steps / elapsed
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]

    # ---- parity of the timed output (last timed step): finite everywhere, image 0 of rank 0 vs the reference digest
    masks, pred, logits = out
    finite = bool(torch.isfinite(masks).all()) and bool(torch.isfinite(logits).all())
    parity = {"outputs_finite": finite}
    dpath = os.path.join(REPO, "tests", "golden", "demo_digest.npz")
    if args.geometry == "demo" and rank == 0 and os.path.exists(dpath):
        with np.load(dpath) as z:
            dg = {k: z[k] for k in z.files}
        m = masks[0:1].cpu().numpy()
        ref_bits = np.unpackbits(dg["mask_bits"])[:m.size].reshape(m.shape).astype(bool)
        inter, union = float(((m > 0) & ref_bits).sum()), float(((m > 0) | ref_bits).sum())
        e_mask = float(np.abs(m.reshape(1, -1)[:, dg["sample_idx"]] - dg["mask_samples"]).max())
        e_log = float(np.abs(logits[0:1].cpu().numpy() - dg["class_logits"]).max())
        ok = finite and inter / max(union, 1.0) >= 0.999 and e_mask <= TOL and e_log <= TOL and \
            pred[0:1].cpu().tolist() == dg["pred"].tolist()
        parity.update({"parity_checked": True, "reference": "tests/golden/demo_digest.npz (reference's own output, image 0)",
                       "mask_iou": round(inter / max(union, 1.0), 6), "max_abs_mask_err": e_mask,
                       "max_abs_class_logit_err": e_log, "pred_equal": pred[0:1].cpu().tolist() == dg["pred"].tolist(),
                       "tolerance": TOL, "ok": ok})
    else:
        parity.update({"parity_checked": False, "ok": finite})
    # split-K hand-offs a tail workgroup gave up on (the tile is NaN then, caught above too): 0 in a healthy run
    parity["gemm_handoff_errors"] = sum(e.ws.gemm_errors() for e in (cas, cas.encoder, cas.decoder, cas.clip))
    parity["ok"] = bool(parity["ok"] and parity["gemm_handoff_errors"] == 0)
    if world > 1:
        okt = torch.tensor([int(parity["ok"])], device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        parity["all_ranks_ok"] = bool(int(okt.item()))

    # ---- roofline of the dominant kernel (instrumented repeat; not part of `value`)
    roofline = None
    if not args.no_roofline and rank == 0:
        records, arecs = [], {"global": [], "window": []}
        orig, orig_attn = hip.gemm, hip.attention
        nrep = max(1, min(args.steps, 2))

        def timed_gemm(a, w, M, N, K, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            orig(a, w, M, N, K, **kw)
            e1.record()
            records.append((2.0 * M * N * K * kw.get("batch", 1), e0, e1))

        def timed_attn(qkv, o, Bn, S, heads, hd, **kw):
            mode = kw.get("mode", 0)
            if mode not in (1, 2):
                return orig_attn(qkv, o, Bn, S, heads, hd, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            orig_attn(qkv, o, Bn, S, heads, hd, **kw)
            e1.record()
            if mode == 1:                                   # global: every query x every key
                fl = 4.0 * S * S * hd * heads * Bn
            else:                                           # 14x14 windows of the padded map (image_encoder.py:507-530)
                w = kw["window"]
                G = kw["grid"]
                nw = -(-G // w)
                fl = 4.0 * (w * w) ** 2 * hd * heads * Bn * nw * nw
            arecs["global" if mode == 1 else "window"].append((fl, e0, e1))

        hip.gemm, hip.attention = timed_gemm, timed_attn
        was_overlap = cas.overlap_clip
        cas.overlap_clip = False                            # per-kernel event times need the kernels one at a time
        try:
            for _ in range(nrep):
                step()
            torch.cuda.synchronize()
        finally:
            hip.gemm, hip.attention = orig, orig_attn
            cas.overlap_clip = was_overlap
        traffic, tnote = None, None
        for tname in ("r02_gemm_traffic.json",):   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
            tfile = os.path.join(REPO, "profiles", tname)
            if args.geometry == "demo" and args.precision == "exact" and B == 8 and os.path.exists(tfile):
                with open(tfile) as f:
                    traffic = round(json.load(f)["traffic_bytes_per_launch"])
                tnote = ("REPLAYED, not measured in this run: HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE "
                         f"(x2 gfx950 correction) and WRITE_SIZE passes of this command (profiles/{tname})")
                break
        flops = sum(r[0] for r in records)
        ms = sum(r[1].elapsed_time(r[2]) for r in records)
        achieved = flops / (ms * 1e-3) / 1e12
        secondary = []
        for name, kern in (("global", "attn_g64pair_kernel (ViT-H global attention, 64x64 map)"),
                           ("window", "attn_win14_kernel (ViT-H 14x14 window attention)")):
            rs = arecs[name]
            if rs:
                f_, m_ = sum(r[0] for r in rs), sum(r[1].elapsed_time(r[2]) for r in rs)
                secondary.append({"kernel": kern, "bound": "mfma", "achieved": round(f_ / (m_ * 1e-3) / 1e12, 2),
                                  "peak": MFMA_F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": round(f_ / (m_ * 1e-3) / 1e12 / MFMA_F16_DENSE_PEAK_TFLOPS, 4),
                                  "launches": len(rs), "avg_launch_us": round(1e3 * m_ / len(rs), 2),
                                  "algorithmic_gflop_per_launch": round(f_ / len(rs) / 1e9, 3),
                                  "note": "algorithmic FLOPs (4*S^2*hd per head, SURVEY.md §8d); exact mode issues 3 MFMAs per "
                                          "product and pads hd 80 -> 96 in P.V, so the issued rate is 3-3.3x this"})
        roofline = {"kernel": "gemm_nt_kernel<split=%d>" % cas.prec.gemm, "bound": "mfma",
                    "achieved": round(achieved, 2), "peak": MFMA_F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / MFMA_F16_DENSE_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_note": tnote,
                    "launches": len(records), "avg_launch_us": round(1e3 * ms / len(records), 2),
                    "algorithmic_gflop_per_launch": round(flops / len(records) / 1e9, 3),
                    "gemm_share_of_step": round(ms * 1e-3 / nrep / (my_elapsed / args.steps), 3),
                    "issued": round(achieved * cas.prec.gemm, 2),
                    "issued_note": "MFMA flops issued: the exact mode forms every product from 3 f16 MFMAs (hi.hi + lo.hi + hi.lo)",
                    "power_roofline": {"peak": MFMA_F16_POWER_ROOFLINE_TFLOPS, "unit": "TFLOP/s issued",
                                       "frac_issued": round(achieved * cas.prec.gemm / MFMA_F16_POWER_ROOFLINE_TFLOPS, 4),
                                       "note": "what an MFMA-only loop sustains at the 1400-W socket cap with random operands "
                                               "(profiles/r02_power_roofline.log); the GEMM itself runs AT the cap: its energy per "
                                               "launch = matrix pipe 51 % + L2->LDS DMA 26 % + idle 21 % (DESIGN.md section 6)"},
                    "secondary": secondary}

    # ---- CPU baseline: the oracle on the host cores, bounded sample (rank 0, N = 1)
    cpu = None
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        from oracle import cvlm_oracle as O
        host_cores = os.cpu_count() or 1
        cores = min(16, host_cores)                   # the GPU box's CPU share for one GPU
        torch.set_num_threads(cores)
        osd = O.to_torch_sd(sd_np)
        n_warm, n_img = (1, 3) if args.geometry == "demo" else (1, 1)
        cpu_in = [t[:n_warm + n_img].cpu() if t.shape[0] >= n_warm + n_img else None for t in (inp, ci, cm)]
        if cpu_in[0] is None:                           # batch smaller than the sample: make the images
            cpu_in = [torch.from_numpy(t) for t in synth.make_inputs(g, c, batch=n_warm + n_img)]
        with torch.no_grad():
            tfc = tf.cpu()
            times = []
            for i in range(n_warm + n_img):            # sequential B = 1 forwards: the reference's semantics
                tc = time.perf_counter()
                O.cascade(cpu_in[0][i:i + 1], cpu_in[1][i:i + 1], cpu_in[2][i:i + 1], osd, g, c, tfc, bank)
                times.append(time.perf_counter() - tc)
            tc = time.perf_counter()
            O.clip_text_features(osd, c, eot)           # untruncated 77-token encoder, as the reference runs it per call
            text_s = time.perf_counter() - tc
        s_img = sum(times[n_warm:]) / n_img
        s_ref = s_img + 2.0 * text_s                    # cocotrainers/mapleAlphaCLIP.py:285-286: text encoder in both passes
        cpu = {"value": round(1.0 / s_img, 5), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"{n_warm} warm-up + {n_img} image(s), sequential B=1 full cascade, fp32 torch-CPU oracle, text bank "
                         f"cached: {s_img:.2f} s/image (per image: {', '.join('%.2f' % t for t in times[n_warm:])}; warm-up "
                         f"{times[0]:.2f})",
               "value_text_encoder_per_call": round(1.0 / s_ref, 5),
               "text_encoder_seconds": round(text_s, 3),
               "note": "value = text bank cached (what the HIP path does); value_text_encoder_per_call = the reference's "
                       "semantics (61-prompt text encoder re-run in both CLIP passes of every image): s/image + 2 x "
                       "text_encoder_seconds, the encoder timed once on the same cores",
               "cpu_model": cpu_model(), "host_cores": host_cores}

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
                       "clip_pass1_overlap": bool(cas.overlap_clip), "stage2_pipelined_under_next_batch": bool(cas.overlap_clip),
                       "text_bank_seconds_once": round(text_bank_s, 3), "setup_seconds": round(setup_s, 1)},
            "step_ms": {"median": round(percentile(step_ms, 0.5), 3), "p10": round(percentile(step_ms, 0.1), 3),
                        "p90": round(percentile(step_ms, 0.9), 3), "n": len(step_ms),
                        "note": "per-step HIP-event durations of the timed steps on rank 0"},
            "images_per_s_per_gpu": round(value / world, 3),
            "achieved_tflops_algorithmic": round(value * WORK_TFLOP_PER_IMAGE, 1) if args.geometry == "demo" else None,
            "parity": parity, "bank_check": bank_check, "power": power,
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    failed = not parity.get("all_ranks_ok", parity["ok"]) or (bank_check is not None and
                                                             not bank_check["bit_identical_to_single_rank_bank_on_every_rank"])
    if world > 1:
        dist.barrier()                                   # rank 0 ran the instrumented roofline repeat meanwhile
        dist.destroy_process_group()
    if failed:
        sys.exit(3)


if __name__ == "__main__":
    main()
