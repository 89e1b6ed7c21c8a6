"""Launcher: run one of the reference's own scripts (demo.py, test_ovcos_maskdecoder_edge.py) on the MI355X drop-in.

    cd /path/to/camouflaged-vlm            # the reference checkout: configs/, datasets/ovcamo_info/*.pth, checkpoints
    PYTHONPATH=/path/to/this/repo python -m camouflaged_vlm_amd.run [--device N] [--precision exact] demo.py --config ...

Why a launcher and not PYTHONPATH alone: `python demo.py` puts the script's directory at sys.path[0], in FRONT of
PYTHONPATH, so `import models` (demo.py:7) finds the checkout's own package and `models/__init__.py:1-2` pulls the
reference classes.  Here the drop-in directory (packages `models`, `cocotrainers`, `recorder`) and this repo go to
sys.path[0:2], the script's directory follows (its `datasets/`, `alpha_clip_rw/`, `utils` keep resolving to the
checkout, ahead of any pip package of the same name), and the script is executed IN THIS PROCESS with
`runpy.run_path(..., run_name="__main__")` -- no re-exec, nothing has touched the GPU before the decision is made.

`--device N`: the scripts hard-code `os.environ["CUDA_VISIBLE_DEVICES"] = '3'` / `'2'` in their first lines
(demo.py:3, test_ovcos_maskdecoder_edge.py:3), which hides every GPU of a smaller allocation.  With `--device` the
launcher selects that GPU (`HIP_VISIBLE_DEVICES`) and initialises the HIP runtime before the script's first line runs,
so the later assignment cannot change what the process sees.
"""
from __future__ import annotations

import argparse
import os
import runpy
import sys
from typing import List, Optional

SHADOWED = ("models", "cocotrainers", "recorder")


def dropin_paths() -> List[str]:
    import camouflaged_vlm_amd as cv
    return [cv.DROPIN_DIR, cv.REPO_DIR]


def arrange_sys_path(script: str) -> List[str]:
    """sys.path = [drop-in, repo, script dir, <what was there>]; modules of the shadowed packages that something has
    already imported from elsewhere are forgotten so the next import resolves afresh."""
    front = dropin_paths() + [os.path.dirname(os.path.abspath(script))]
    seen = set()
    merged = []
    for p in front + [p for p in sys.path if p not in ("", ".")]:
        key = os.path.abspath(p) if p else p
        if key not in seen:
            seen.add(key)
            merged.append(p)
    sys.path[:] = merged
    for name in list(sys.modules):
        root = name.split(".", 1)[0]
        if root in SHADOWED:
            f = getattr(sys.modules[name], "__file__", None) or ""
            if not os.path.abspath(f).startswith(os.path.abspath(front[0]) + os.sep):
                del sys.modules[name]
    return merged


def pin_device(device: Optional[int]) -> None:
    if device is None:
        return
    os.environ["HIP_VISIBLE_DEVICES"] = str(device)
    os.environ.pop("CUDA_VISIBLE_DEVICES", None)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit(f"camouflaged_vlm_amd.run: --device {device}: no HIP device visible")
    torch.cuda.init()                                    # the runtime has read the environment: later edits are inert


def main(argv: Optional[List[str]] = None) -> None:
    ap = argparse.ArgumentParser(prog="python -m camouflaged_vlm_amd.run", description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--device", type=int, default=None, help="GPU index to run on (see above)")
    ap.add_argument("--precision", choices=["exact", "mx", "mx22", "mx33", "mixed", "fast"], default=None,
                    help="sets CVLM_PRECISION for the drop-in modules.  Default mx: e4m3 correction products in the large GEMMs, one-term q.k^T / "
                         "two-term P.v in the ViT-H attention on batches of two or more images (mask logits within 5.0e-4 of the reference on 16 "
                         "images x 73728 positions; one image per call runs `exact`, and the first batch on a set of weights is checked against "
                         "`exact` by the engine).  exact: three f16 products per multiply everywhere (5e-5 of the reference, ~10 %% slower on batches)")
    ap.add_argument("script", help="the reference script to run, e.g. demo.py")
    ap.add_argument("args", nargs=argparse.REMAINDER, help="arguments of the script")
    ns = ap.parse_args(argv)
    if not os.path.isfile(ns.script):
        raise SystemExit(f"camouflaged_vlm_amd.run: no such script: {ns.script}")
    if ns.precision:
        os.environ["CVLM_PRECISION"] = ns.precision
    arrange_sys_path(ns.script)
    pin_device(ns.device)
    sys.argv = [ns.script] + list(ns.args)
    runpy.run_path(ns.script, run_name="__main__")


if __name__ == "__main__":
    main()
