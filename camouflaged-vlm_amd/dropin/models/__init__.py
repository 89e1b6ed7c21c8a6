"""Drop-in ``models`` package: same import surface as the reference (models/__init__.py:1-2) --
``models.register``, ``models.make`` and the registry names ``sam`` and ``sam_maskdecoder_edge`` --
backed by the MI355X HIP path of camouflaged_vlm_amd."""
import os as _os
import sys as _sys

_repo = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
if _repo not in _sys.path:
    _sys.path.insert(0, _repo)

from .models import register, make  # noqa: E402,F401
from . import sam, sam_maskdecoder_edge  # noqa: E402,F401
