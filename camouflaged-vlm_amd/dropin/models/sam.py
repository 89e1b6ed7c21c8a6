"""Registry entry ``sam`` (reference models/sam.py:298): kept so that ``import models`` registers
both names without requiring ``open_clip``.  The vanilla-decoder variant is outside this round's
hot path; constructing it fails loudly instead of silently running on CPU."""
from .models import register


@register('sam')
class SAM:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            "registry entry 'sam' (vanilla MaskDecoder) is not built yet; use 'sam_maskdecoder_edge' "
            "(the path named by BASELINE.json).")
