#!/usr/bin/env python3
"""Headline benchmark: images/sec at 1024x1024 through the full cascade (SAM-Adapter ViT-H encoder +
edge mask decoder + MaPLe/Alpha-CLIP ViT-L/14@336, stage 1 + stage 2 classification), batch 8 per GPU,
synthetic images + deterministic synthetic weights (BASELINE.json configs[2]; configs[3] for N > 1).

  python bench.py --gpus N --steps K --warmup W

N > 1 without RANK in the environment: this process starts N ranks itself (`python -m torch.distributed.run`,
as a child, before anything here has touched the GPU), relays rank 0's JSON line and exits with the child's
code.  Under a launcher (RANK / WORLD_SIZE set) it is one of the ranks.

A step = one pass of the hot path over one resident batch (8 images per GPU by default).  Two DIFFERENT batches alternate
from step to step (images (r + i) mod 16 and (r + B + i) mod 16 on rank r), so that a cross-batch hazard of the pipelined loop could
not hide behind identical inputs.  Rank 0 prints ONE JSON line.
`value`       : whole-job images/s over the K timed steps (wall clock, barrier + synchronize on both sides, max over ranks);
                `step_ms` holds the median / p10 / p90 of the per-step HIP-event durations of the same K steps.
`parity`      : the outputs of the LAST TWO TIMED steps (one per batch) are checked: all finite; EVERY image against the reference
                digest (tests/golden/demo_digest.npz: 16 images, each a B = 1 forward of the reference itself): IoU >= 0.999,
                |mask| and |class logits| <= 1e-3, same prediction -- on EVERY rank (rank r's batches are the digest's images
                rotated by r); `all_ranks_ok` is the AND over ranks, `min_images_checked_against_reference_per_rank` the proof.
`bank_check`  : N > 1: the all-gathered text bank equals, bit for bit, the bank every rank computes alone.
`roofline`    : the dominant kernel (the split-half MFMA GEMM): algorithmic FLOPs (2*M*N*K per launch) divided by its
                HIP-event time over an instrumented repeat of the timed steps; `secondary` holds the ViT-H attention
                kernels measured the same way (algorithmic 4*S^2*hd per head).
`cpu_baseline`: the CPU oracle (oracle/cvlm_oracle.py, a port of the reference forward) timed on the host cores on a
                bounded sample (rank 0, N = 1 only).

Other lines (same JSON shape, `config.workload` says which):
  --workload encoder                                  BASELINE configs[1]: SAM ViT-H image encoder only, batch 8
  --workload encoder --geometry hires1536 --batch 4   BASELINE configs[4]: encoder built at 1536^2, batch 4
  --surface dropin [--batch 1|8]                      the reference's own call pattern (demo.py:110-122,
        test_ovcos_maskdecoder_edge.py:89-113, DataLoader batch_size=1): models.make(cfg).cuda() -> load_mapleAlphaCLIP ->
        load_state_dict(strict) -> per step: infer_test -> torch.sigmoid -> F.interpolate -> clip_model, one host
        synchronisation per step (the scripts read the prediction back every image).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0        # MI355X_MICROARCH.md: BF16/F16 dense ~2.5 PF
WORK_TFLOP_PER_IMAGE = 6.461               # algorithmic work per image, text bank cached
ENCODER_TFLOP_PER_IMAGE = {1024: 5.681, 1536: 13.712}
# Measured on this part (tools/power_roofline.py, profiles/r02_power_roofline.log): an MFMA-only loop (16x16x32 f16, random
# operands, every CU) settles at 1.88 PF at the 1400-W socket cap (sclk 1.95 GHz of 2.4): what the matrix pipe can sustain.
MFMA_F16_POWER_ROOFLINE_TFLOPS = 1880.0
TRAFFIC_FILES = {"mx": ("r06_gemm_traffic.json", "r05_gemm_traffic.json"), "exact": ("r05_gemm_traffic_exact.json", "r04_gemm_traffic.json", "r03_gemm_traffic.json", "r02_gemm_traffic.json")}   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, newest first


def under_profiler() -> bool:
    """rocprofv3 preloads its tool library (it initialises the GPU): no child process may be started from here then."""
    pre = os.environ.get("LD_PRELOAD", "")
    return "rocprof" in pre or "roctx" in pre or any(k.startswith(("ROCP", "ROCPROF", "ROCPROFILER")) for k in os.environ)


class PowerSampler:
    """Socket power and shader clock during the timed region.  Preferred source: the amdgpu hwmon / sysfs files, read by a
    thread of THIS process (no child, nothing to leak, safe under a profiler).  Fallback: a child process polling rocm-smi
    with a scrubbed environment, never under a profiler, which ends by itself when this process goes away.
    Everything is optional: no source, no samples, no field."""

    def __init__(self, device_index: int = 0, enabled: bool = True, pci: str = ""):
        self.enabled = enabled
        self.dev = device_index
        self.pci = pci.lower()
        self.proc, self.path = None, None
        self.thread, self.stop_flag, self.samples = None, threading.Event(), []
        self.power_file, self.sclk_file = None, None

    def _find_sysfs(self):
        import glob
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*"))
        # keep amdgpu cards only, in card order; the rank's GPU is the device_index-th of them
        cards = [c for c in cards if os.path.exists(os.path.join(c, "power1_average")) or os.path.exists(os.path.join(c, "power1_input"))]
        # the host's sysfs lists every GPU of the node, visible to this process or not: pick ours by PCI address
        mine = [c for c in cards if self.pci and os.path.basename(os.path.realpath(os.path.join(os.path.dirname(os.path.dirname(c))))).lower() == self.pci]
        if mine:
            hw = mine[0]
        elif not self.pci and self.dev < len(cards):
            hw = cards[self.dev]
        else:
            return False
        for name in ("power1_average", "power1_input"):
            f = os.path.join(hw, name)
            if os.path.exists(f):
                self.power_file = f
                break
        f = os.path.join(hw, "freq1_input")
        self.sclk_file = f if os.path.exists(f) else None
        try:
            with open(self.power_file) as fh:
                float(fh.read().strip())
        except (OSError, ValueError, TypeError):
            self.power_file = None
        return self.power_file is not None

    def _poll(self):
        while not self.stop_flag.is_set():
            try:
                with open(self.power_file) as fh:
                    w = float(fh.read().strip()) * 1e-6
                mhz = None
                if self.sclk_file:
                    with open(self.sclk_file) as fh:
                        mhz = float(fh.read().strip()) * 1e-6
                self.samples.append((time.time(), w, mhz))
            except (OSError, ValueError):
                pass
            self.stop_flag.wait(0.25)

    def start(self):
        if not self.enabled:
            return
        if self._find_sysfs():
            self.thread = threading.Thread(target=self._poll, daemon=True)
            self.thread.start()
            return
        import shutil, tempfile
        if under_profiler() or shutil.which("rocm-smi") is None:
            return
        fd, self.path = tempfile.mkstemp(prefix="cvlm_power_", suffix=".log")
        os.close(fd)
        env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS"))}
        try:
            # ends by itself when this process is gone (kill -0 $PPID) or after 10 minutes, whatever happens here
            self.proc = subprocess.Popen(
                ["bash", "-c", "n=0; while kill -0 $PPID 2>/dev/null && [ $n -lt 2400 ]; do echo STAMP $(date +%s.%N); "
                               "rocm-smi --showpower --showclocks; sleep 0.25; n=$((n+1)); done"],
                stdout=open(self.path, "w"), stderr=subprocess.DEVNULL, start_new_session=True, env=env)
        except OSError:
            self.proc = None

    def stop(self, t0: float = 0.0, t1: float = float("inf")):
        import re, signal
        if self.thread is not None:
            self.stop_flag.set()
            self.thread.join(timeout=2)
            self.thread = None
            sel = [s for s in self.samples if t0 <= s[0] <= t1]
            if not sel:
                return None
            pw = [s[1] for s in sel]
            sc = [s[2] for s in sel if s[2] is not None]
            return {"socket_w_mean": round(sum(pw) / len(pw), 1), "socket_w_max": round(max(pw), 1),
                    "sclk_mhz_mean": round(sum(sc) / len(sc)) if sc else None, "sclk_mhz_max_of_part": 2400, "samples": len(pw),
                    "source": self.power_file,
                    "note": "amdgpu hwmon (power1_input / freq1_input of this GPU's PCI device) read by a thread of this process "
                            "during the timed steps"}
        if self.proc is None:
            return None
        try:
            os.killpg(self.proc.pid, signal.SIGTERM)            # exactly the group started above
            self.proc.wait(timeout=5)
        except Exception:
            pass
        self.proc = None
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
            p = re.search(r"GPU\[%d\]\s*: .*Power \(W\): ([0-9.]+)" % self.dev, chunk)
            c = re.search(r"GPU\[%d\]\s*: sclk clock level: \d+: \((\d+)Mhz\)" % self.dev, chunk)
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
    ap.add_argument("--precision", default=os.environ.get("CVLM_PRECISION", "mx"), choices=["mx", "mx12", "mx22", "mx33", "exact", "mixed", "fast"])
    ap.add_argument("--geometry", default="demo", choices=["demo", "tiny", "hires1536"])
    ap.add_argument("--workload", default="cascade", choices=["cascade", "encoder"],
                    help="encoder = SAM ViT-H image encoder only (BASELINE configs[1] / [4] with --geometry hires1536)")
    ap.add_argument("--surface", default="engine", choices=["engine", "dropin", "evalloop"],
                    help="dropin = the reference's call surface and call pattern (models.make / infer_test / torch sigmoid + "
                         "interpolate / clip_model, one synchronisation per step), cascade workload only; evalloop = the reference's "
                         "evaluation loop end to end (test_ovcos_maskdecoder_edge.py:89-141): uint8 images + uint8 ground truth on the "
                         "host -> H2D -> GPU preprocessing (N1) -> infer_test -> stage 2 -> Classification + mask_to_u8 + "
                         "OVCOSMetricer on the device (N2), one read-back of the metric dict at the end")
    ap.add_argument("--pipelined", action="store_true",
                    help="--surface evalloop: the engine's pipelined serving loop underneath (decoder / stage 2 of batch i under the encoder "
                         "of batch i + 1, evaluation tail on a third stream) instead of the reference's call-by-call order")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run CLIP pass 1 after the SAM encoder instead of on a side stream beneath it (profiling runs: "
                         "co-running kernels stretch each other's durations)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --single-device: rehearse the N-rank path on a box with ONE GPU (all ranks on cuda:0)")
    ap.add_argument("--single-device", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact-leg", action="store_true", help="cascade workload: skip the 5-step run of the same loop in precision `exact` behind the timed loop")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-power", action="store_true", help="do not sample socket power / clock during the timed steps")
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


def physical_cores() -> int:
    """Physical cores this process may run on: distinct (package, core) pairs of /proc/cpuinfo among the CPUs of its affinity set."""
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    seen, cpu, phys = set(), None, None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                k, _, v = line.partition(":")
                k = k.strip()
                if k == "processor":
                    cpu, phys = int(v), None
                elif k == "physical id":
                    phys = int(v)
                elif k == "core id" and cpu in allowed:
                    seen.add((phys, int(v)))
    except (OSError, ValueError):
        pass
    return len(seen) or len(allowed)


def cgroup_cpu_limit():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unknown."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        return None if q == "max" else round(float(q) / float(p), 2)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = float(f.read())
        return None if q <= 0 else round(q / p, 2)
    except (OSError, ValueError):
        return None


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# the ARITHMETIC of each precision (not a precision claim: BASELINE's gate is 1e-3 abs / IoU 0.999 against the fp32 reference, checked in-run)
_MX = ("f16 hi.hi MFMA + block-scaled e4m3 MFMA for the two correction products (3 mantissa bits) in the ViT-H qkv / lin1 / lin2 and CLIP MLP GEMMs; "
       "3x f16 split MFMA in the other GEMMs; f32 accumulate")
DTYPE_NAMES = {"exact": "3x f16 split MFMA (hi.hi + lo.hi + hi.lo, ~22 significant bits per operand), f32 accumulate",
               "mx": _MX + "; ViT-H attention: q.k^T = ONE f16 product (Q and K as single f16 values), P.v = 2x f16 (P a single f16, V hi + lo)",
               "mx12": _MX + "; ViT-H attention: q.k^T = ONE f16 product (Q and K as single f16 values), P.v = 2x f16 (P a single f16, V hi + lo)",
               "mx22": _MX + "; ViT-H attention: 2x f16 per product (Q and P single f16 values, K and V hi + lo)",
               "mx33": _MX + "; ViT-H attention: 3x f16 split products",
               "mixed": "3x f16 split MFMA in GEMMs and q.k^T, f16 P.v", "fast": "f16 operands, f32 accumulate"}


class Roofline:
    """Instrumented repeat of the step: every cvlm_gemm / ViT-H cvlm_attention launch between two HIP events on the stream it
    is launched on (torch's current stream is the stream hip.py hands to the launchers)."""

    def __init__(self, torch, hip, split: int):
        self.torch, self.hip, self.split = torch, hip, split
        self.records, self.arecs = [], {"global": [], "window": []}

    def __enter__(self):
        torch, hip = self.torch, self.hip
        self.orig, self.orig_attn = hip.gemm, hip.attention
        orig, orig_attn, records, arecs = self.orig, self.orig_attn, self.records, self.arecs

        def timed_gemm(a, w, M, N, K, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            orig(a, w, M, N, K, **kw)
            e1.record()
            # algorithmic bytes of the launch: both operands once (4 B per element: two fp16 planes), the residual if any, every output
            nb = kw.get("batch", 1)
            by = 4.0 * (M * K + N * K) * nb + (4.0 * M * N * nb if (kw.get("residual") is not None or kw.get("residual_h2") is not None) else 0.0) \
                + 4.0 * M * N * nb * ((kw.get("out_f32") is not None) + (kw.get("out_h2") is not None))
            if kw.get("head_major_nolo"):                      # ABI 12: a third's lo plane is not written (2 B per element of that third)
                by -= 2.0 * M * (N // 3) * bin(int(kw["head_major_nolo"]) & 7).count("1")
            records.append((2.0 * M * N * K * nb, e0, e1, by, bool(getattr(a, "mx", False))))

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
            # f16 MFMAs per multiply, averaged over the two products (q.k^T and P.v have the same algorithmic flops): (3, 3) -> 3, (2, 2) -> 2, (1, 2) -> 1.5
            arecs["global" if mode == 1 else "window"].append((fl, e0, e1, S, 0.5 * (kw.get("split_qk", 3) + kw.get("split_pv", 3))))

        hip.gemm, hip.attention = timed_gemm, timed_attn
        return self

    def __exit__(self, *exc):
        self.hip.gemm, self.hip.attention = self.orig, self.orig_attn
        return False

    def result(self, nrep: int, step_seconds: float, traffic_ok: bool, precision: str = "exact") -> dict:
        records, arecs = self.records, self.arecs
        traffic, tnote = None, None
        for tname in TRAFFIC_FILES.get(precision, ()):
            tfile = os.path.join(REPO, "profiles", tname)
            if traffic_ok and os.path.exists(tfile):
                with open(tfile) as f:
                    traffic = round(json.load(f)["traffic_bytes_per_launch"])
                tnote = ("REPLAYED, not measured in this run: HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE "
                         f"(x2 gfx950 correction) and WRITE_SIZE passes of the default command (profiles/{tname})")
                break
        flops = sum(r[0] for r in records)
        abytes = sum(r[3] for r in records)
        ms = sum(r[1].elapsed_time(r[2]) for r in records)
        achieved = flops / (ms * 1e-3) / 1e12
        secondary = []
        for name in ("global", "window"):
            rs = arecs[name]
            if not rs:
                continue
            S = rs[0][3]
            kern = {"global": "attn_g64pair_kernel (ViT-H global attention, 64x64 map)" if S == 4096 else
                              f"global-attention kernel, S = {S} ({int(S ** 0.5)}x{int(S ** 0.5)} map)",
                    "window": "attn_win14p_kernel (ViT-H 14x14 window attention, producer / consumer form)"}[name]
            f_, m_ = sum(r[0] for r in rs), sum(r[1].elapsed_time(r[2]) for r in rs)
            tf = f_ / (m_ * 1e-3) / 1e12
            issued_factor = float(rs[0][4])                 # MFMAs per product (see timed_attn); no padding of head dim 80
            secondary.append({"kernel": kern, "bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_F16_DENSE_PEAK_TFLOPS,
                              "unit": "TFLOP/s", "frac": round(tf / MFMA_F16_DENSE_PEAK_TFLOPS, 4),
                              "issued": round(tf * issued_factor, 1), "frac_issued": round(tf * issued_factor / MFMA_F16_DENSE_PEAK_TFLOPS, 4),
                              "launches": len(rs), "avg_launch_us": round(1e3 * m_ / len(rs), 2),
                              "algorithmic_gflop_per_launch": round(f_ / len(rs) / 1e9, 3),
                              "products_per_multiply": rs[0][4],
                              "note": "achieved / frac: algorithmic FLOPs (4*S^2*hd per head, SURVEY.md §8d); issued: the f16 MFMA flops executed "
                                      "for them (3 per multiply with hi/lo operands on both sides; precision mx: 1 for q.k^T + 2 for P.v = 1.5 on average)"})
        mx_flops = sum(r[0] for r in records if r[4])
        mx_ms = sum(r[1].elapsed_time(r[2]) for r in records if r[4])
        # matrix-pipe work in f16-MFMA equivalents: a split-3 launch issues 3 f16 products per multiply, an mx launch 1 f16 product + 2 e4m3
        # products at twice the f16 rate (= 2 equivalents)
        issued_factor = (3.0 * (flops - mx_flops) + 2.0 * mx_flops) / flops if self.split == 3 else float(self.split)
        kname = "gemm_nt_kernel<split=%d>" % self.split
        if mx_flops > 0:
            kname = ("gemm_nt_kernel: mx unit form (f16 hi.hi + block-scaled e4m3 corrections; %.0f %% of the GEMM flops, %.0f TFLOP/s algorithmic "
                     "on its own) + split-3 forms elsewhere" % (100.0 * mx_flops / flops, mx_flops / (mx_ms * 1e-3) / 1e12))
        return {"kernel": kname, "bound": "mfma",
                "achieved": round(achieved, 2), "peak": MFMA_F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_F16_DENSE_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_note": tnote,
                "algorithmic_bytes_per_launch": round(abytes / len(records)),
                "traffic_over_algorithmic": round(traffic / (abytes / len(records)), 3) if traffic else None,
                "traffic_explained": "each of the 8 XCD L2s pulls every weight panel it multiplies (weights x 8: the price of partitioning "
                                     "the ROWS over the XCDs, which keeps the 5-30x larger activation to ~2 fetches); the re-fetches are "
                                     "served by the 256-MB infinity cache, and HBM + fabric are < 2 % of a launch's energy "
                                     "(profiles/r02_power_gemm_parts.log; DESIGN.md section 6)",
                "launches": len(records), "avg_launch_us": round(1e3 * ms / len(records), 2),
                "algorithmic_gflop_per_launch": round(flops / len(records) / 1e9, 3),
                "gemm_share_of_step": round(ms * 1e-3 / nrep / step_seconds, 3),
                "issued": round(achieved * issued_factor, 2),
                "issued_note": "matrix-pipe work in f16-MFMA equivalents: split-3 launches form every product from 3 f16 MFMAs (hi.hi + lo.hi + "
                               "hi.lo), mx launches from 1 f16 MFMA + 2 e4m3 MFMAs at twice the f16 rate (= 2)",
                "power_roofline": {"peak": MFMA_F16_POWER_ROOFLINE_TFLOPS, "unit": "TFLOP/s issued",
                                   "frac_issued": round(achieved * issued_factor / MFMA_F16_POWER_ROOFLINE_TFLOPS, 4),
                                   "note": "what an MFMA-only loop sustains at the 1400-W socket cap with random operands "
                                           "(profiles/r02_power_roofline.log); the GEMM itself runs AT the cap: its energy per "
                                           "launch = matrix pipe 51 % + L2->LDS DMA 26 % + idle 21 % (DESIGN.md section 6)"},
                "secondary": secondary}


def synthetic_dataset(np, n_items: int, n_cls: int, seed: int = 5):
    """uint8 HWC photographs-in-shape (smooth colour fields + texture + noise) with uint8 {0, 255} blob masks of the same size,
    mixed sizes as a real test split has them, and a label per item."""
    sizes = [(768, 1024), (1080, 1920), (683, 1024), (1024, 1024), (1365, 2048), (600, 800), (1024, 683), (960, 1280)]
    rng = np.random.default_rng(seed)
    items = []
    for i in range(n_items):
        h, w = sizes[i % len(sizes)]
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        img = np.empty((h, w, 3), np.float32)
        for ch in range(3):
            fx, fy, ph = rng.uniform(2, 9) / w, rng.uniform(2, 9) / h, rng.uniform(0, 6.28)
            img[..., ch] = 128 + 70 * np.sin(6.28 * (fx * xx + fy * yy) + ph) + 25 * np.sin(6.28 * 37 * (xx / w + yy / h) + ch)
        img += rng.normal(0, 12, img.shape).astype(np.float32)
        cy, cx, r = rng.uniform(0.3, 0.7) * h, rng.uniform(0.3, 0.7) * w, rng.uniform(0.12, 0.3) * min(h, w)
        gt = np.where((yy - cy) ** 2 + ((xx - cx) * rng.uniform(0.6, 1.4)) ** 2 < r * r, 255, 0).astype(np.uint8)
        items.append((np.clip(img, 0, 255).astype(np.uint8), gt, int(rng.integers(0, n_cls))))
    return items


def evalloop_line(args, torch, np, g, c, dev, B, make_model, sampler, t0, percentile, cpu_model):
    """`--surface evalloop`: test_ovcos_maskdecoder_edge.py:89-141 end to end on the device (camouflaged_vlm_amd.evalloop)."""
    from camouflaged_vlm_amd.evalloop import DeviceEvalLoop, SECTIONS
    model, names = make_model()
    n_items = max(16, 2 * B)
    data = synthetic_dataset(np, n_items, len(names))
    pin = lambda a: torch.from_numpy(a).pin_memory()
    nb = n_items // B
    data = data[:nb * B]                                  # whole batches only (a batch size that does not divide the split drops the rest)
    items = [(pin(im), pin(gt), lab) for im, gt, lab in data]

    def batch(k):
        sel = items[(k % nb) * B:(k % nb + 1) * B]
        return [s[0] for s in sel], [s[1] for s in sel], torch.tensor([s[2] for s in sel], dtype=torch.int64)

    # ---- pass 0, untimed: builds every cache (weights packed, resize tables, text bank) and tells what a random-init model
    # predicts; the synthetic labels are then chosen so that two images of three carry the predicted class (a class mismatch
    # zeroes every metric of its image, ovcos_metricer.py:18-19: with random labels the whole dict would be trivially 0 / 1)
    loop = DeviceEvalLoop(model, names)
    preds = []
    for k in range(nb):
        loop.step(*batch(k))
        preds += loop.last[1].cpu().tolist()
    items = [(im, gt, int(preds[i]) if i % 3 else (int(preds[i]) + 1) % len(names)) for i, (im, gt, _) in enumerate(items)]
    # ---- first pass over the whole synthetic split, untimed: the run the parity check below reads (logits kept)
    loop = DeviceEvalLoop(model, names)
    kept = []
    for k in range(nb):
        loop.step(*batch(k))
        kept.append((loop.last[0].clone(), loop.last[1].clone(), loop.last[2].clone(), [m.clone() for m in loop.last[3]]))
    torch.cuda.synchronize()
    dev_metrics, dev_cls = loop.results()
    dev_cod = loop.cod_results()
    setup_s = time.time() - t0

    # ---- the timed loop: W warm-up steps, K timed steps, the final read-back of the metric dict INSIDE the timed region
    loop = DeviceEvalLoop(model, names, pipelined=args.pipelined)
    for i in range(args.warmup):
        loop.step(*batch(i))
    loop.results()
    torch.cuda.synchronize()
    loop = DeviceEvalLoop(model, names, timed=True, pipelined=args.pipelined)
    sampler.start()
    wall0 = time.time()
    try:
        t1 = time.perf_counter()
        for i in range(args.steps):
            loop.step(*batch(i))
        torch.cuda.synchronize()
        t_sync = time.perf_counter()
        timed_metrics, timed_cls = loop.results()             # the one D2H + the host's float64 arithmetic on the counters
        elapsed = time.perf_counter() - t1
        readback_ms = 1e3 * (time.perf_counter() - t_sync)
    finally:
        power = sampler.stop(wall0 + 0.3, time.time())
    value = B * args.steps / elapsed
    sec = loop.section_ms()
    tot = sum(sec.values())
    shares = {k: round(v / tot, 4) for k, v in sec.items()}
    per_img = {k: round(v / (B * args.steps), 4) for k, v in sec.items()}
    host_per_img = {k: round(v / (B * args.steps), 4) for k, v in loop.section_host_ms().items()}

    # ---- what needs no CPU restatement: finite outputs, no abandoned hand-offs, the timed loop's dict equals the first pass's
    n_all = nb * B
    finite = all(bool(torch.isfinite(kp[0]).all()) and bool(torch.isfinite(kp[2]).all()) for kp in kept)
    cas = model.cascade()
    handoff = sum(e.ws.gemm_errors() for e in (cas, cas.encoder, cas.decoder, cas.clip))
    same_run = max(abs(timed_metrics[k] - dev_metrics[k]) for k in dev_metrics) if args.steps % nb == 0 and args.steps >= nb else None
    parity = {"outputs_finite": finite, "parity_checked": False, "images": n_all, "gemm_handoff_errors": handoff,
              "timed_loop_dict_equals_first_pass": same_run,
              "device_metrics": {k: round(float(v), 6) for k, v in dev_metrics.items()}, "device_classification": dev_cls,
              "device_calc_cod": {k: round(float(v), 6) for k, v in dev_cod.items()}}
    # pipelined: stage 2 shares one CLIP forward with the next batch's pass 1 -- other K-splits, other fp32 summation order: a mask level
    # may move on a handful of pixels
    parity["ok"] = bool(finite and handoff == 0 and (same_run is None or same_run < (1e-4 if args.pipelined else 1e-12)))
    # with --no-cpu-baseline nothing below compares against the oracle; `ok` is then self-consistency only and says so
    parity["ok_means"] = "self-consistency only (finite outputs, no hand-off errors, the timed loop's dict equals the first pass): nothing was " \
                         "compared with the oracle / reference in this run"

    # ---- cpu_baseline leg: the same loop's two tails as the reference runs them, on the host -- Pillow + numpy preprocessing (what
    # torchvision's transforms call, datasets/wrappers.py:22-62) and D2H + cv2-style resize + the six numpy metric classes
    # (oracle/metrics_oracle.py = the reference's OVCOSMetricer / sod_metric classes, pinned bit for bit by tests/golden/evaltail.npz)
    # -- over every image of the synthetic split.  Its results double as the CHECKER of the device loop: the metric dict against the
    # oracle fed (a) the device's own uint8 masks, (b) masks the oracle makes from the device's logits; N1 against the
    # Pillow-pinned oracle bit for bit.
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import metrics_oracle as MO
        from oracle import preprocess_oracle as PO
        from camouflaged_vlm_amd.preprocess import nearest_indices
        host_cores = os.cpu_count() or 1
        torch.set_num_threads(min(16, host_cores))
        cods = []
        try:
            from PIL import Image
            im_mean, im_std = loop.pre.im_mean.cpu().numpy(), loop.pre.im_std.cpu().numpy()
            cl_mean, cl_std = loop.pre.cl_mean.cpu().numpy(), loop.pre.cl_std.cpu().numpy()

            def n1_cpu(im):
                a = np.asarray(Image.fromarray(im).resize((g.inp_size, g.inp_size), Image.BILINEAR), np.float32) / 255.0
                a = ((a - im_mean) / im_std).transpose(2, 0, 1)
                rh, rw = PO.clip_resize_shape(im.shape[0], im.shape[1], c.image_resolution)
                b = np.asarray(Image.fromarray(im).resize((rw, rh), Image.BICUBIC), np.float32) / 255.0
                top, left = PO.center_crop_box(rh, rw, c.image_resolution)
                b = b[top:top + c.image_resolution, left:left + c.image_resolution]
                return a, ((b - cl_mean) / cl_std).transpose(2, 0, 1)
            n1_kind = "Pillow (PIL.Image.resize, what torchvision Resize calls) + numpy ToTensor / Normalize"
        except ImportError:
            def n1_cpu(im):
                return PO.sam_input(im, g.inp_size), PO.clip_input(im, c.image_resolution)
            n1_kind = "oracle/preprocess_oracle.py (numpy restatement of Pillow's resample; Pillow itself not importable)"
        t_n1, t_n2 = [], []
        steps_a, steps_b, cls_ok, worst_levels, frac_moved = [], [], True, 0, 0.0
        scores_all, labels_all = [], []
        for k in range(nb):
            logits, pred_1, score, masks_u8 = kept[k]
            lab = batch(k)[2].numpy()
            sc = score.cpu().numpy()
            scores_all.append(sc); labels_all.append(lab)
            cls_ok = cls_ok and np.array_equal(sc.argmax(axis=1), pred_1.cpu().numpy())
            for j in range(B):
                im, gt, _ = data[k * B + j]
                tc = time.perf_counter(); n1_cpu(im); t_n1.append(time.perf_counter() - tc)
                same = int(pred_1[j]) == int(lab[j])
                tc = time.perf_counter()
                lgj = logits[j, 0].cpu().numpy()                         # the 4-MB D2H of the float mask (:116)
                u8_orc = MO.mask_to_u8(lgj, *gt.shape)
                steps_b.append(MO.ovcos_metrics(u8_orc, gt, same))
                # :105 calc_cod(pred_mask, batch['gt']): the float map against the NEAREST-resized ground truth (wrappers.py:29-32)
                gt_s = (gt[nearest_indices(gt.shape[0], g.inp_size)][:, nearest_indices(gt.shape[1], g.inp_size)] / 255.0).astype(np.float32)
                cods.append(MO.calc_cod(MO.sigmoid_f32(lgj)[None, None], gt_s[None, None]))
                t_n2.append(time.perf_counter() - tc)
                u8_dev = masks_u8[j].cpu().numpy()
                d = np.abs(u8_dev.astype(np.int16) - u8_orc.astype(np.int16))
                worst_levels, frac_moved = max(worst_levels, int(d.max())), max(frac_moved, float((d != 0).mean()))
                steps_a.append(MO.ovcos_metrics(u8_dev, gt, same))
        agg_a, agg_b = MO.aggregate(steps_a), MO.aggregate(steps_b)
        err_a = max(abs(dev_metrics[k] - agg_a[k]) for k in agg_a)
        err_b = max(abs(dev_metrics[k] - agg_b[k]) for k in agg_b)
        _, c1, c5 = MO.classification(np.concatenate(scores_all), np.concatenate(labels_all))
        cls_err = max(abs(dev_cls["accuracy"] - 100.0 * c1 / n_all), abs(dev_cls["top5"] - 100.0 * c5 / n_all))
        im0 = data[1][0]
        n1_equal = bool(np.array_equal(loop.pre.sam_input(torch.from_numpy(im0).to(dev)).cpu().numpy()[0], PO.sam_input(im0, g.inp_size)) and
                        np.array_equal(loop.pre.clip_input(torch.from_numpy(im0).to(dev)).cpu().numpy()[0], PO.clip_input(im0, c.image_resolution)))
        cod_err = max(abs(dev_cod[key] - float(np.mean([cdv[k] for cdv in cods]))) for k, key in enumerate(("sm", "em", "wfm", "mae")))
        parity.update({"parity_checked": True, "metric_dict_vs_oracle_on_device_masks": err_a, "metric_dict_vs_oracle_from_logits": err_b,
                       "calc_cod_vs_oracle": cod_err, "tolerance_calc_cod": 1e-5,
                       "tolerance_on_device_masks": 1e-9, "tolerance_from_logits": 1e-4, "mask_u8_max_level_diff": worst_levels,
                       "mask_u8_max_fraction_of_pixels_moved": frac_moved, "classification_err": cls_err, "pred_is_argmax": bool(cls_ok),
                       "n1_bit_exact_vs_pillow_pinned_oracle": n1_equal,
                       "reference": "oracle/metrics_oracle.py = the reference's OVCOSMetricer / sod_metric classes (bit-exact pin: "
                                    "tests/golden/evaltail.npz); cv2.resize restated, unpinned"})
        parity["ok"] = bool(parity["ok"] and err_a <= 1e-9 and err_b <= 1e-4 and worst_levels <= 1 and frac_moved < 1e-3 and cls_err < 1e-9 and
                            cls_ok and n1_equal and cod_err <= 1e-5)
        parity["ok_means"] = "checked against the oracle (the reference's metric classes, pinned) as listed"
        n1_ms, n2_ms = 1e3 * sum(t_n1) / n_all, 1e3 * sum(t_n2) / n_all
        gpu_path_ms = per_img["path_infer_test_stage2"]
        cpu = {"value": round(1e3 / (gpu_path_ms + n1_ms + n2_ms), 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"all {n_all} images of the synthetic split: CPU preprocessing {n1_ms:.1f} ms/image ({n1_kind}), CPU evaluation tail "
                         f"{n2_ms:.1f} ms/image (D2H of the float mask + cv2-style resize + the six numpy metric classes + calc_cod's four on "
                         f"the {g.inp_size}^2 float map, oracle/metrics_oracle.py), one process, serial with the GPU path ({gpu_path_ms:.2f} ms/image) as in the reference's "
                         "loop body (its DataLoader workers would hide the preprocessing share, not the tail)",
               "n1_cpu_ms_per_image": round(n1_ms, 2), "n2_cpu_ms_per_image": round(n2_ms, 2),
               "cpu_model": cpu_model(), "host_cores": host_cores}
    tail_ms = per_img["h2d"] + per_img["n1_preprocess"] + per_img["n2_eval_tail"]
    line = {"metric": "images/sec through the reference's evaluation loop (uint8 image + uint8 ground truth on the host -> metric dict)",
            "value": round(value, 3), "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_NAMES[args.precision], "data": "synthetic",
            "config": {"workload": f"evaluation loop, batch {B} (test_ovcos_maskdecoder_edge.py:89-141): H2D of uint8 images / masks of mixed "
                                   "sizes -> GpuPreprocess (N1) -> infer_test -> torch.sigmoid -> F.interpolate(336) -> clip_model -> "
                                   "calc_cod + Classification.process + mask_to_u8 + OVCOSMetricer.step on the device (N2); metric dicts read back once, "
                                   "inside the timed region" if args.geometry == "demo" else "tiny geometry (debug)",
                       "images_per_step": B, "distinct_images": n_items, "pipelined": bool(args.pipelined), "image_sizes": sorted(set(d[0].shape[:2] for d in data)),
                       "precision": args.precision, "setup_seconds": round(setup_s, 1)},
            "section_ms_per_image": per_img, "section_share_of_gpu_time": shares, "section_host_issue_ms_per_image": host_per_img,
            "n1_plus_n2_plus_h2d_share": round(tail_ms / (tail_ms + per_img["path_infer_test_stage2"]), 4),
            "final_readback_ms": round(readback_ms, 3),
            "note": "sections are HIP-event times on the current stream (the path section contains the side-stream CLIP pass it joins); "
                    "`value` is wall clock over the K steps including the final read-back",
            "parity": parity, "power": power, "roofline": None, "cpu_baseline": cpu}
    return line, parity["ok"]


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

    from camouflaged_vlm_amd import digest, hip, host, spec, synth
    from camouflaged_vlm_amd.engine import Cascade, Precision
    import camouflaged_vlm_amd as cv
    if cv.DROPIN_DIR not in sys.path:
        sys.path.insert(0, cv.DROPIN_DIR)
    from cocotrainers.mapleAlphaCLIP import gather_text_features

    g, c = (spec.DEMO_SAM, spec.DEMO_CLIP) if args.geometry in ("demo", "hires1536") else (spec.TINY_SAM, spec.TINY_CLIP)
    if args.geometry == "hires1536":
        import dataclasses
        g = dataclasses.replace(g, inp_size=1536)            # model *built* at 1536 (pos_embed 96^2, rel_pos 191x80)
        args.workload = "encoder"
    if args.surface in ("dropin", "evalloop") and (args.workload != "cascade" or world > 1):
        sys.exit(f"--surface {args.surface}: cascade workload on one GPU")
    B = args.batch
    prec = Precision.named(args.precision)
    t0 = time.time()
    sd_np = synth.make_full_state_dict(g, c)
    sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    # the rank's share: two batches of B images.  The reference digests hold images 0..15 (demo) / 0..3 (1536^2); EVERY rank
    # draws its batches from those images, rotated by its rank (rank r, batch k: images (r + k*B + i) mod n), so that the
    # parity check below is a check against the reference on every rank (results identical to 1 GPU),
    # never a finiteness check alone.  Geometries without a digest (tiny) keep disjoint image ranges per rank.
    n_dig = {"demo": 16, "hires1536": 4}.get(args.geometry)
    if n_dig:
        ids = digest.rank_batches(rank, B, n_dig)
    else:
        ids = [list(range((2 * rank + k) * B, (2 * rank + k + 1) * B)) for k in range(2)]

    def make_batch(id_list):
        per = [synth.make_inputs(g, c, batch=1, index0=i) for i in id_list]       # image i never depends on its batch
        return tuple(torch.from_numpy(np.concatenate([p[j] for p in per])).to(dev) for j in range(3))
    batches = [make_batch(ids[k]) for k in range(2)]
    try:
        pr_ = torch.cuda.get_device_properties(dev)
        pci = "%04x:%02x:%02x.0" % (pr_.pci_domain_id, pr_.pci_bus_id, pr_.pci_device_id)
    except Exception:
        pci = ""
    sampler = PowerSampler(local_rank, enabled=(rank == 0 and not args.no_power), pci=pci)

    def timed_loop(step, sync_each: bool = False, flush=lambda: None):
        """W warm-up steps, then K timed steps between barrier + synchronize; returns (elapsed, max-over-ranks elapsed,
        per-step ms, outputs of the last two steps as [(batch index, output)], power)."""
        for i in range(args.warmup):
            step(i % 2)
            if sync_each:
                torch.cuda.synchronize()
        flush()                                              # nothing owed from the warm-up when the clock starts
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        outs = []
        sampler.start()
        wall0 = time.time()
        try:
            t1 = time.perf_counter()
            marks[0].record()
            for i in range(args.steps):
                o = step(i % 2)
                marks[i + 1].record()
                if sync_each:
                    torch.cuda.synchronize()
                outs.append((i % 2, o))
                outs = outs[-2:]
            flush()                                          # the pipelined loop's last stage 2: K steps = K full cascades
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            mine = time.perf_counter() - t1
        finally:
            power = sampler.stop(wall0 + 0.3, time.time())
        total = mine
        if world > 1:
            tmax = torch.tensor([mine], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            total = float(tmax.item())
        step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
        return mine, total, step_ms, outs, power

    def rank_rates(mine: float, images_per_step: int):
        """images/s of every rank over its own timed loop (stragglers show as a low minimum)."""
        r = images_per_step * args.steps / mine
        if world == 1:
            return {"min": round(r, 3), "max": round(r, 3)}
        t = torch.tensor([r], dtype=torch.float64, device=dev)
        lo, hi = t.clone(), t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        return {"min": round(float(lo.item()), 3), "max": round(float(hi.item()), 3)}

    def step_block(step_ms):
        return {"median": round(percentile(step_ms, 0.5), 3), "p10": round(percentile(step_ms, 0.1), 3),
                "p90": round(percentile(step_ms, 0.9), 3), "n": len(step_ms),
                "note": "per-step HIP-event durations of the timed steps on rank 0"}

    def finish(line, ok: bool):
        if rank == 0:
            print(json.dumps(line), flush=True)
        if world > 1:
            dist.barrier()                                   # rank 0 ran the instrumented roofline repeat meanwhile
            dist.destroy_process_group()
        if not ok:
            sys.exit(3)

    def min_over_ranks(n: int) -> int:
        if world == 1:
            return n
        t = torch.tensor([n], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item())

    def all_ranks(ok: bool) -> bool:
        if world == 1:
            return ok
        okt = torch.tensor([int(ok)], device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        return bool(int(okt.item()))

    # ==================================================================================================================
    # SAM ViT-H image encoder only (BASELINE configs[1]; configs[4] at 1536^2)
    # ==================================================================================================================
    if args.workload == "encoder":
        from camouflaged_vlm_amd.engine import SamEncoder
        enc = SamEncoder(sd, g, dev, prec)
        setup_s = time.time() - t0
        names = ["features0", "features1"]

        def step(k):
            return enc.forward(batches[k][0], out_name=names[k])
        mine, elapsed, step_ms, outs, power = timed_loop(step)
        value = world * B * args.steps / elapsed
        finite = all(bool(torch.isfinite(o).all()) for _, o in outs)
        parity = {"outputs_finite": finite, "parity_checked": False, "ok": finite}
        dname = {"demo": "demo_digest.npz", "hires1536": "hires1536_digest.npz"}.get(args.geometry)
        if dname and os.path.exists(digest.golden_path(dname)):
            dg = digest.load(digest.golden_path(dname))
            chk = digest.check_demo_features if args.geometry == "demo" else digest.check_hires_features
            res = [chk(o, g.grid, dg, ids[k]) for k, o in outs]
            res = [r for r in res if r["checked_images"]]
            checked = sorted(set(i for r in res for i in r["checked_images"]))
            if checked:
                parity.update({"parity_checked": True, "checked_images": checked,
                               "reference": f"tests/golden/{dname}: the reference's own encoder output per image (samples + channel means)",
                               "max_abs_feature_err": max(r.get("max_abs_feature_err", 0.0) for r in res),
                               "max_abs_channel_mean_err": max(r.get("max_abs_channel_mean_err", 0.0) for r in res),
                               "tolerance": digest.TOL, "ok": bool(finite and all(r["ok"] for r in res))})
        if n_dig and dname and os.path.exists(digest.golden_path(dname)) and not parity["parity_checked"]:
            parity["ok"] = False                             # a digest exists and this rank verified nothing: not a pass
        parity["gemm_handoff_errors"] = enc.ws.gemm_errors()
        parity["ok"] = bool(parity["ok"] and parity["gemm_handoff_errors"] == 0)
        parity["all_ranks_ok"] = all_ranks(parity["ok"])
        parity["min_images_checked_against_reference_per_rank"] = min_over_ranks(len(parity.get("checked_images", [])))
        rates = rank_rates(mine, B)
        roofline = None
        if not args.no_roofline and rank == 0:
            nrep = max(1, min(args.steps, 2))
            with Roofline(torch, hip, prec.gemm) as rf:
                for i in range(nrep):
                    step(i % 2)
                torch.cuda.synchronize()
            roofline = rf.result(nrep, mine / args.steps, traffic_ok=False)
        cpu = None
        if not args.no_cpu_baseline and rank == 0 and world == 1:
            from oracle import cvlm_oracle as O
            host_cores = os.cpu_count() or 1
            torch.set_num_threads(min(16, host_cores))
            osd = O.to_torch_sd(sd_np)
            n_warm, n_img = (1, 1) if g.inp_size <= 1024 else (0, 1)     # 1536^2: one image is 40-90 s of CPU work
            times = []
            with torch.no_grad():
                for i in range(n_warm + n_img):
                    tc = time.perf_counter()
                    O.sam_encoder(batches[0][0][i:i + 1].cpu(), osd, g)
                    times.append(time.perf_counter() - tc)
            s_img = sum(times[n_warm:]) / n_img
            cpu = {"value": round(1.0 / s_img, 5), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                   "sample": f"{n_warm} warm-up + {n_img} image(s), B=1 SAM encoder forward of the fp32 torch-CPU oracle: "
                             f"{s_img:.2f} s/image", "cpu_model": cpu_model(), "host_cores": host_cores}
        tf_img = ENCODER_TFLOP_PER_IMAGE.get(g.inp_size)
        line = {"metric": f"images/sec, SAM ViT-H image encoder only at {g.inp_size}x{g.inp_size}",
                "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": DTYPE_NAMES[args.precision], "data": "synthetic",
                "config": {"workload": f"SAM-Adapter ViT-H image encoder only, batch {B}, {g.inp_size}^2 "
                                       f"(BASELINE configs[{1 if g.inp_size == 1024 else 4}])" if args.geometry != "tiny" else "tiny geometry (debug)",
                           "images_per_gpu_per_step": B, "global_batch": B * world, "precision": args.precision,
                           "parallelism": f"dp{world} (images sharded)", "setup_seconds": round(setup_s, 1)},
                "step_ms": step_block(step_ms), "images_per_s_per_gpu": round(value / world, 3), "images_per_s_per_rank": rates,
                "achieved_tflops_algorithmic": round(value * tf_img, 1) if tf_img and args.geometry != "tiny" else None,
                "parity": parity, "power": power, "roofline": roofline, "cpu_baseline": cpu}
        return finish(line, parity["all_ranks_ok"])

    # ==================================================================================================================
    # the reference's call surface and call pattern (demo.py:78-122; test_ovcos_maskdecoder_edge.py:68-113, batch_size=1)
    # ==================================================================================================================
    def make_dropin_model():
        """demo.py:78-89 / test_ovcos_maskdecoder_edge.py:165-176: the model through the reference's own construction calls"""
        import models
        from cocotrainers.mapleAlphaCLIP import CustomCLIP
        os.environ["CVLM_PRECISION"] = args.precision
        consts = host.ovcamo_constants()
        eot_te = host.eot_for_classes(consts["names_test"].tolist())[:c.n_cls_test] if args.geometry == "demo" else spec.default_eot(c, "test")
        eot_tr = host.eot_for_classes(consts["names_train"].tolist())[:c.n_cls_train] if args.geometry == "demo" else spec.default_eot(c, "train")
        enc_mode = dict(name="sam", img_size=g.inp_size, mlp_ratio=g.mlp_ratio, patch_size=g.patch_size, qkv_bias=True,
                        use_rel_pos=True, window_size=g.window_size, out_chans=g.out_chans, scale_factor=32, input_type="fft",
                        freq_nums=0.25, prompt_type="highpass", prompt_embed_dim=g.prompt_embed_dim, tuning_stage=1234,
                        handcrafted_tune=True, embedding_tune=True, adaptor="adaptor", embed_dim=g.embed_dim, depth=g.depth,
                        num_heads=g.num_heads, global_attn_indexes=list(g.global_attn_indexes))   # configs/demo.yaml `model.args`
        maple = CustomCLIP(geometry=c, eot_train=eot_tr, eot_test=eot_te)
        model = models.make({"name": "sam_maskdecoder_edge",
                             "args": {"inp_size": g.inp_size, "loss": "iou", "encoder_mode": enc_mode}}).cuda()   # demo.py:84
        model.train_text_features = model.train_text_features[:c.n_cls_train]
        model.test_text_features = model.test_text_features[:c.n_cls_test]
        model.load_mapleAlphaCLIP(maple)                                                                         # demo.py:85
        model.load_state_dict(sd, strict=True)                                                                   # demo.py:88
        model.eval()
        if args.no_overlap:
            model.cascade().overlap_clip = False
        names = consts["names_test"].tolist()[:c.n_cls_test] if args.geometry == "demo" else [f"class{i}" for i in range(c.n_cls_test)]
        return model, names

    if args.surface == "evalloop":
        return finish(*evalloop_line(args, torch, np, g, c, dev, B, make_dropin_model, sampler, t0, percentile, cpu_model))

    if args.surface == "dropin":
        import torch.nn.functional as F
        model, _ = make_dropin_model()
        # every image of the reference digest takes its turn: 16 // B different batches
        nb = max(1, 16 // B) if args.geometry == "demo" else 2
        allimg = synth.make_inputs(g, c, batch=nb * B)
        dbatches = [tuple(torch.from_numpy(np.ascontiguousarray(t[k * B:(k + 1) * B])).to(dev) for t in allimg) for k in range(nb)]
        R = c.image_resolution
        seen = {}

        def one(k):
            inp, clip_image, clip_mask = dbatches[k]
            pred_mask = model.infer_test(inp, clip_image, clip_mask)                                             # demo.py:116
            prob = torch.sigmoid(pred_mask)                                                                      # demo.py:117
            alpha = F.interpolate(prob, (R, R), mode="bilinear", align_corners=False)                          # demo.py:120
            _, _, pred_1, score = model.clip_model(clip_image, alpha, train=False)                               # demo.py:122
            return pred_mask, pred_1, score

        with torch.no_grad():
            t_first = time.perf_counter()
            one(0)
            torch.cuda.synchronize()
            first_call_s = time.perf_counter() - t_first          # packs the weights, encodes the text bank once
            setup_s = time.time() - t0
            for i in range(args.warmup):
                one(i % nb)
                torch.cuda.synchronize()
            marks = []
            sampler.start()
            wall0 = time.time()
            try:
                t1 = time.perf_counter()
                for i in range(args.steps):
                    ts = time.perf_counter()
                    o = one(i % nb)
                    torch.cuda.synchronize()                      # the scripts read pred_1 / the mask back for every image
                    marks.append(1e3 * (time.perf_counter() - ts))
                    seen[i % nb] = o
                elapsed = time.perf_counter() - t1
            finally:
                power = sampler.stop(wall0 + 0.3, time.time())
        value = B * args.steps / elapsed
        finite = all(bool(torch.isfinite(o[0]).all()) and bool(torch.isfinite(o[2]).all()) for o in seen.values())
        parity = {"outputs_finite": finite, "parity_checked": False, "ok": finite}
        if args.geometry == "demo" and os.path.exists(digest.golden_path("demo_digest.npz")):
            dg = digest.load(digest.golden_path("demo_digest.npz"))
            res = [digest.check_cascade(o[0], o[1], o[2], dg, list(range(k * B, (k + 1) * B))) for k, o in sorted(seen.items())]
            checked = sorted(set(i for r in res for i in r["checked_images"]))
            parity.update({"parity_checked": True, "checked_images": checked,
                           "reference": "tests/golden/demo_digest.npz: the reference's own output per image (B = 1 forwards)",
                           "min_mask_iou": round(min(r["min_iou"] for r in res), 6),
                           "max_abs_mask_err": max(r["max_abs_mask_err"] for r in res),
                           "max_abs_class_logit_err": max(r["max_abs_class_logit_err"] for r in res),
                           "pred_equal": all(r["pred_equal"] for r in res), "tolerance": digest.TOL,
                           "ok": bool(finite and all(r["ok"] for r in res))})
        cas = model.cascade()
        parity["mx_self_check"] = cas.mx_self_check_result
        parity["gemm_handoff_errors"] = sum(e.ws.gemm_errors() for e in (cas, cas.encoder, cas.decoder, cas.clip))
        parity["ok"] = bool(parity["ok"] and parity["gemm_handoff_errors"] == 0)
        roofline = None
        if not args.no_roofline:
            nrep = max(1, min(args.steps, 2))
            was = cas.overlap_clip
            cas.overlap_clip = False
            try:
                with Roofline(torch, hip, cas.prec.gemm) as rf, torch.no_grad():
                    for i in range(nrep):
                        one(i % nb)
                    torch.cuda.synchronize()
            finally:
                cas.overlap_clip = was
            roofline = rf.result(nrep, elapsed / args.steps, traffic_ok=False)
        line = {"metric": "images/sec at 1024x1024 (SAM-ViT-H + CLIP ViT-L/14), full cascade through the reference's call surface",
                "value": round(value, 3), "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": DTYPE_NAMES[args.precision], "data": "synthetic",
                "config": {"workload": f"drop-in surface, batch {B}: models.make(cfg).cuda() -> load_mapleAlphaCLIP -> load_state_dict(strict) "
                                       "-> per step infer_test -> torch.sigmoid -> F.interpolate(336) -> clip_model, one "
                                       "torch.cuda.synchronize() per step (demo.py:110-122; BASELINE configs[0] call pattern on the GPU)"
                           if args.geometry == "demo" else "tiny geometry (debug)",
                           "images_per_step": B, "precision": args.precision, "clip_pass1_overlap": bool(cas.overlap_clip),
                           "first_call_seconds": round(first_call_s, 2), "setup_seconds": round(setup_s, 1),
                           "distinct_batches": nb},
                "latency_ms_per_step": {"median": round(percentile(marks, 0.5), 3), "p10": round(percentile(marks, 0.1), 3),
                                        "p90": round(percentile(marks, 0.9), 3), "n": len(marks),
                                        "note": "host wall clock from the call of infer_test to the end of the synchronize after clip_model"},
                "latency_ms_per_image": round(percentile(marks, 0.5) / B, 3),
                "achieved_tflops_algorithmic": round(value * WORK_TFLOP_PER_IMAGE, 1) if args.geometry == "demo" else None,
                "parity": parity, "power": power, "roofline": roofline, "cpu_baseline": None}
        return finish(line, parity["ok"])

    # ==================================================================================================================
    # full cascade, engine level, pipelined serving loop (the headline)
    # ==================================================================================================================
    cas = Cascade(sd, g, c, dev, prec)
    if args.no_overlap:
        cas.overlap_clip = False
    eot = host.eot_for_classes(host.ovcamo_constants()["names_test"].tolist())[:c.n_cls_test] \
        if args.geometry == "demo" else spec.default_eot(c, "test")
    bank = torch.from_numpy(host.ovcamo_constants()["bank_test"][:c.n_cls_test]).float()
    # shared text-embedding bank: sharded over ranks + all-gathered (the only collective on the path)
    tt = time.time()
    tf = gather_text_features(cas.clip, eot, "test")
    # SURVEY.md §8(e) / BASELINE configs[2] "75 OVCamo class prompts": the 14 TRAIN prompts travel the same way (their bank is what
    # CustomCLIP.forward(train=True) scores against, cocotrainers/mapleAlphaCLIP.py:267-280); the timed inference path reads the 61 test rows
    eot_tr = host.eot_for_classes(host.ovcamo_constants()["names_train"].tolist())[:c.n_cls_train] \
        if args.geometry == "demo" else spec.default_eot(c, "train")
    tf_tr = gather_text_features(cas.clip, eot_tr, "train")
    torch.cuda.synchronize()
    text_bank_s = time.time() - tt
    bank_check = None
    if world > 1:
        # the gathered banks are bit-identical on every rank and equal to the single-GPU banks
        alone, alone_tr = cas.clip.text_features(eot, "test"), cas.clip.text_features(eot_tr, "train")
        same = torch.tensor([int(torch.equal(alone, tf) and torch.equal(alone_tr, tf_tr))], device=dev)
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        bank_check = {"bit_identical_to_single_rank_bank_on_every_rank": bool(int(same.item())),
                      "rows_gathered": {"test": int(tf.shape[0]), "train": int(tf_tr.shape[0])},
                      "max_abs_diff_rank0": float(max((alone - tf).abs().max(), (alone_tr - tf_tr).abs().max()))}
    cas.clip.set_text_bank(tf, bank, "test")
    cas.clip.set_text_bank(tf_tr, torch.from_numpy(host.ovcamo_constants()["bank_train"][:c.n_cls_train]).float(), "train")
    # precision mx: the engine's self-check on these weights (two images as a batch against image 0 alone in `exact`) runs here, in the setup,
    # not inside the warm-up or the timed region
    cas._mx_self_check(*batches[0])
    torch.cuda.synchronize()
    setup_s = time.time() - t0

    def step(k):
        # serving loop: batch i's stage 2 (on the side stream) runs under batch i+1's SAM encoder; everything is
        # complete at the synchronize() that closes the timed region
        return cas.cascade(*batches[k], pipelined=True)

    mine, elapsed, step_ms, outs, power = timed_loop(step, flush=cas.flush)
    value = world * B * args.steps / elapsed

    # ---- parity of the timed outputs (last two timed steps = both batches): finite everywhere, and every image against the
    # reference digest -- on EVERY rank (the ranks' batches are rotations of the digest's images)
    finite = all(bool(torch.isfinite(o[0]).all()) and bool(torch.isfinite(o[2]).all()) for _, o in outs)
    parity = {"outputs_finite": finite, "parity_checked": False, "ok": finite}
    dpath = digest.golden_path("demo_digest.npz")
    if args.geometry == "demo" and os.path.exists(dpath):
        dg = digest.load(dpath)
        res = [digest.check_cascade(o[0], o[1], o[2], dg, ids[k]) for k, o in outs]
        res = [r for r in res if r["checked_images"]]
        if res:
            parity.update({"parity_checked": True, "checked_images": sorted(set(i for r in res for i in r["checked_images"])),
                           "reference": "tests/golden/demo_digest.npz: the reference's own output per image (B = 1 forwards)",
                           "mask_iou": round(min(r["min_iou"] for r in res), 6),
                           "max_abs_mask_err": max(r["max_abs_mask_err"] for r in res),
                           "mask_positions_per_image": res[0]["mask_positions_per_image"],
                           "max_abs_mask_err_by_set": {k: max(r["max_abs_mask_err_by_set"][k] for r in res) for k in res[0]["max_abs_mask_err_by_set"]},
                           "max_abs_class_logit_err": max(r["max_abs_class_logit_err"] for r in res),
                           "pred_equal": all(r["pred_equal"] for r in res), "tolerance": digest.TOL,
                           "ok": bool(finite and all(r["ok"] for r in res))})
    # split-K hand-offs a tail workgroup gave up on (the tile is NaN then, caught above too): 0 in a healthy run
    parity["mx_self_check"] = cas.mx_self_check_result
    parity["gemm_handoff_errors"] = sum(e.ws.gemm_errors() for e in (cas, cas.encoder, cas.decoder, cas.clip))
    parity["ok"] = bool(parity["ok"] and parity["gemm_handoff_errors"] == 0)
    if n_dig and os.path.exists(dpath) and not parity["parity_checked"]:
        parity["ok"] = False                                 # a digest exists and this rank verified nothing: not a pass
    parity["all_ranks_ok"] = all_ranks(parity["ok"])
    parity["min_images_checked_against_reference_per_rank"] = min_over_ranks(len(parity.get("checked_images", [])))
    rates = rank_rates(mine, B)

    # ---- roofline of the dominant kernel (instrumented repeat; not part of `value`)
    roofline = None
    if not args.no_roofline and rank == 0:
        nrep = max(1, min(args.steps, 2))
        was_overlap = cas.overlap_clip
        cas.overlap_clip = False                            # per-kernel event times need the kernels one at a time
        try:
            with Roofline(torch, hip, cas.prec.gemm) as rf:
                step(1)                                     # leaves a stage 2 owed, so that both repeats below run the fused CLIP forward
                rf.records.clear(); rf.arecs["global"].clear(); rf.arecs["window"].clear()
                for i in range(nrep):
                    step(i % 2)
                torch.cuda.synchronize()
            cas.flush()
            torch.cuda.synchronize()
        finally:
            cas.overlap_clip = was_overlap
        roofline = rf.result(nrep, mine / args.steps,
                             traffic_ok=(args.geometry == "demo" and args.precision in ("mx", "exact") and B == 8), precision=args.precision)

    # ---- the reference-grade arithmetic beside it: the SAME loop in precision `exact` (3x f16 split everywhere),
    # in this process on this box, behind the timed loop -- not part of `value`.  A second engine (its own packed weights) for 2 warm-up +
    # 5 timed steps, every image of its last two steps against the reference.
    exact_mode = None
    if args.precision != "exact" and not args.no_exact_leg and rank == 0 and world == 1 and args.geometry == "demo":
        cas_x = Cascade(sd, g, c, dev, Precision.named("exact"))
        cas_x.clip.set_text_bank(cas_x.clip.text_features(eot, "test"), bank, "test")
        if args.no_overlap:
            cas_x.overlap_clip = False
        for i in range(2):
            cas_x.cascade(*batches[i % 2], pipelined=True)
        cas_x.flush()
        torch.cuda.synchronize()
        nx = 5
        tx = time.perf_counter()
        xo = []
        for i in range(nx):
            xo.append((i % 2, cas_x.cascade(*batches[i % 2], pipelined=True)))
            xo = xo[-2:]
        cas_x.flush()
        torch.cuda.synchronize()
        ex_s = time.perf_counter() - tx
        exact_mode = {"value": round(B * nx / ex_s, 3), "unit": "images/s", "ms_per_step": round(1e3 * ex_s / nx, 3), "steps": nx, "warmup": 2,
                      "dtype": DTYPE_NAMES["exact"], "note": "same process, same box, same batches, run behind the timed loop; not part of `value`"}
        if os.path.exists(dpath):
            rx = [digest.check_cascade(o[0], o[1], o[2], dg, ids[k]) for k, o in xo]
            rx = [r for r in rx if r["checked_images"]]
            if rx:
                exact_mode["parity"] = {"checked_images": sorted(set(i for r in rx for i in r["checked_images"])),
                                        "mask_iou": round(min(r["min_iou"] for r in rx), 6), "max_abs_mask_err": max(r["max_abs_mask_err"] for r in rx),
                                        "max_abs_class_logit_err": max(r["max_abs_class_logit_err"] for r in rx),
                                        "pred_equal": all(r["pred_equal"] for r in rx), "ok": all(r["ok"] for r in rx)}
        del cas_x
        torch.cuda.empty_cache()

    # ---- CPU baseline: the oracle on the host cores, bounded sample (rank 0, N = 1)
    cpu = None
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        from oracle import cvlm_oracle as O
        host_cores = os.cpu_count() or 1
        cores = min(16, host_cores)                   # the GPU box's CPU share for one GPU
        torch.set_num_threads(cores)
        osd = O.to_torch_sd(sd_np)
        n_warm, n_img = (1, 3) if args.geometry == "demo" else (1, 1)
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
        # `torch.set_num_threads(<all physical host cores>)` -- the best the host can do, one image (+ one warm-up
        # at that thread count), reported beside the 16-thread figure (the CPU share of one GPU on this box)
        phys = physical_cores()
        all_cores = None
        if phys > cores:
            torch.set_num_threads(phys)
            with torch.no_grad():
                ta = []
                for i in range(2):
                    tc = time.perf_counter()
                    O.cascade(cpu_in[0][i:i + 1], cpu_in[1][i:i + 1], cpu_in[2][i:i + 1], osd, g, c, tfc, bank)
                    ta.append(time.perf_counter() - tc)
            all_cores = {"value": round(1.0 / ta[1], 5), "unit": "images/s", "cores": torch.get_num_threads(),
                         "sample": f"1 warm-up ({ta[0]:.2f} s) + 1 image ({ta[1]:.2f} s) of the same oracle cascade with "
                                   f"torch.set_num_threads({phys}) = every physical core this process may use",
                         "cgroup_cpu_limit": cgroup_cpu_limit(),
                         "note": "SURVEY.md section 8(d) protocol (all physical host cores).  Where the container's CPU quota "
                                 "(cgroup_cpu_limit, in CPUs) is far below the core count, the extra threads only contend for the "
                                 "same quota and this figure comes out BELOW the 16-thread one: `value` is then the best the host allows"}
            torch.set_num_threads(cores)
        cpu = {"value": round(1.0 / s_img, 5), "unit": "images/s", "cores": cores, "kind": "port",
               "all_physical_cores": all_cores,
               "sample": f"{n_warm} warm-up + {n_img} image(s), sequential B=1 full cascade, fp32 torch-CPU oracle, text bank "
                         f"cached: {s_img:.2f} s/image (per image: {', '.join('%.2f' % t for t in times[n_warm:])}; warm-up "
                         f"{times[0]:.2f})",
               "value_text_encoder_per_call": round(1.0 / s_ref, 5),
               "text_encoder_seconds": round(text_s, 3),
               "note": "value = text bank cached (what the HIP path does); value_text_encoder_per_call = the reference's "
                       "semantics (61-prompt text encoder re-run in both CLIP passes of every image): s/image + 2 x "
                       "text_encoder_seconds, the encoder timed once on the same cores",
               "cpu_model": cpu_model(), "host_cores": host_cores}

    line = {
        "metric": "images/sec at 1024x1024 (SAM-ViT-H + CLIP ViT-L/14), full cascade",
        "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPE_NAMES[args.precision], "data": "synthetic",
        "config": {"workload": "full cascade: SAM-Adapter ViT-H 1024^2 encoder + edge mask decoder + "
                               "Alpha-CLIP ViT-L/14@336 x2 passes, 61 OVCamo prompts (BASELINE configs[2])"
                   if args.geometry == "demo" else "tiny geometry (debug)",
                   "images_per_gpu_per_step": B, "global_batch": B * world, "precision": args.precision,
                   "parallelism": f"dp{world} (images sharded, text bank all-gathered)",
                   "distinct_batches_alternating": 2,
                   "clip_pass1_overlap": bool(cas.overlap_clip), "stage2_pipelined_under_next_batch": bool(cas.overlap_clip),
                   "stage2_fused_with_next_pass1": bool(cas.fuse_clip),
                   "text_bank_seconds_once": round(text_bank_s, 3), "setup_seconds": round(setup_s, 1)},
        "step_ms": step_block(step_ms),
        "images_per_s_per_gpu": round(value / world, 3), "images_per_s_per_rank": rates,
        "achieved_tflops_algorithmic": round(value * WORK_TFLOP_PER_IMAGE, 1) if args.geometry == "demo" else None,
        "parity": parity, "bank_check": bank_check, "power": power,
        "roofline": roofline, "cpu_baseline": cpu, "exact_mode": exact_mode,
    }
    ok = parity["all_ranks_ok"] and (bank_check is None or bank_check["bit_identical_to_single_rank_bank_on_every_rank"])
    finish(line, ok)


if __name__ == "__main__":
    main()
