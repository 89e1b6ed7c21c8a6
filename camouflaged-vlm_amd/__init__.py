"""MI355X-native inference path for the camouflaged-vlm cascade (SAM-Adapter ViT-H encoder ->
edge mask decoder -> MaPLe/Alpha-CLIP ViT-L/14).  Hot operators are hand-written HIP kernels for
gfx950 behind a C-ABI shared library (include/cvlm.h); this package holds the Python host side:

  spec     geometry + state_dict contract
  synth    deterministic synthetic weights / inputs
  hip      ctypes binding of libcvlm_hip.so (fails loudly when the library is missing)
  engine   weight packing + kernel launch schedule of the forward path
  dropin/  mirrors of the reference's ``models`` / ``cocotrainers`` call surface
"""
import os as _os

PKG_DIR = _os.path.dirname(_os.path.abspath(__file__))
REPO_DIR = _os.path.dirname(PKG_DIR)
DROPIN_DIR = _os.path.join(PKG_DIR, "dropin")

__version__ = "0.1.0"
