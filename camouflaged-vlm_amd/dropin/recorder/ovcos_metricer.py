"""`OVCOSMetricer` with the reference's call shape (recorder/ovcos_metricer.py:257-307) over the device counters.

`step(pre, gt, pre_cls, gt_cls, gt_path)` takes what the reference's loop hands it -- uint8 numpy arrays
(test_ovcos_maskdecoder_edge.py:130-136: `pre=(pred * 255).astype(np.uint8), gt=mask`), which are uploaded (1 byte per pixel) -- or
uint8 GPU tensors (camouflaged_vlm_amd.evaltail.mask_to_u8 keeps the mask on the device); the metric set is the reference's."""
import numpy as np
import torch

import camouflaged_vlm_amd  # noqa: F401
from camouflaged_vlm_amd.evaltail import DeviceMetricer


class OVCOSMetricer(DeviceMetricer):
    def __init__(self, class_names, metric_names=("sm", "wfm", "mae", "fm", "em", "iou")):
        super().__init__(class_names, metric_names)

    @staticmethod
    def _device_u8(a, gt_path):
        if isinstance(a, np.ndarray):
            assert a.dtype == np.uint8, (a.dtype, gt_path)            # ovcos_metricer.py:271
            return torch.from_numpy(np.ascontiguousarray(a)).cuda()
        return a if a.is_cuda else a.cuda()

    def step(self, pre, gt, pre_cls, gt_cls, gt_path=None):
        assert tuple(pre.shape) == tuple(gt.shape), (pre.shape, gt.shape, gt_path)   # :270
        return super().step(self._device_u8(pre, gt_path), self._device_u8(gt, gt_path), pre_cls == gt_cls, gt_path)
