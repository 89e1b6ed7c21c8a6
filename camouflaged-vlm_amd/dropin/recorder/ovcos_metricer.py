"""`OVCOSMetricer` with the reference's call shape (recorder/ovcos_metricer.py:257-307) over the device counters.

`step(pre, gt, pre_cls, gt_cls, gt_path)` takes uint8 GPU tensors where the reference takes numpy arrays; the
metric set is the reference's (see camouflaged_vlm_amd.evaltail)."""
import camouflaged_vlm_amd  # noqa: F401
from camouflaged_vlm_amd.evaltail import DeviceMetricer


class OVCOSMetricer(DeviceMetricer):
    def __init__(self, class_names, metric_names=("sm", "wfm", "mae", "fm", "em", "iou")):
        super().__init__(class_names, metric_names)

    def step(self, pre, gt, pre_cls, gt_cls, gt_path=None):
        return super().step(pre, gt, pre_cls == gt_cls, gt_path)
