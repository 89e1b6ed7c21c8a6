"""`Classification` with the reference's call shape (recorder/new_evaluator.py:23-122): `process(mo, gt)` accumulates
top-1 / top-5 hits on the device, `evaluate()` returns accuracy / error_rate / top5 / macro_f1."""
import camouflaged_vlm_amd  # noqa: F401
from camouflaged_vlm_amd.evaltail import DeviceClassification


class Classification(DeviceClassification):
    pass
