"""The reference's evaluation loop, device resident (SURVEY.md §8f N1 + path + N2 end to end).

`eval_psnr_ovcamo_both` (test_ovcos_maskdecoder_edge.py:68-149) does, per image: PIL / torchvision preprocessing in DataLoader
workers (datasets/wrappers.py:22-62), H2D of three float tensors, `infer_test`, sigmoid, `F.interpolate` to 336, `clip_model`,
`Classification.process`, then a 4-MB D2H of the float mask, `cv2.resize` to the ground truth's size, `(pred * 255).astype(uint8)`
and six numpy metric classes (`OVCOSMetricer.step`).  `DeviceEvalLoop.step` takes the *uint8 HWC image* and the *uint8
ground-truth mask* (host, any size), copies them to the device and keeps everything there: `GpuPreprocess` (N1), the drop-in
model's `infer_test` and `clip_model` with the reference's own glue calls, `DeviceClassification.process`, `DeviceMetricer.step_batch`
(N2).  Nothing is read back until `results()`: one D2H of the counters (8 KB per image).

There is no CPU path: the model must live on the GPU and the HIP library must load (camouflaged_vlm_amd.hip raises otherwise).
"""
from __future__ import annotations

import time
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from .evaltail import DeviceClassification, DeviceCod, DeviceMetricer
from .preprocess import GpuPreprocess

SECTIONS = ("h2d", "n1_preprocess", "path_infer_test_stage2", "n2_eval_tail")


class DeviceEvalLoop:
    def __init__(self, model, class_names: Sequence[str], metric_names=("sm", "wfm", "mae", "fm", "em", "iou"),
                 clip_mask_convention: str = "wrapper", timed: bool = False, pipelined: bool = False):
        """model: the drop-in `models.make(...)` object (on the GPU, CLIP loaded); class_names: the test split's classes
        (test_ovcos_maskdecoder_edge.py:77-87).  timed: record HIP events around the four sections of every step.
        pipelined: the serving loop of `engine.Cascade.cascade(pipelined=True)` underneath -- decoder and stage 2 of batch i run under
        the encoder of batch i + 1, and the evaluation tail of batch i runs on a third stream once its class scores exist (one call
        later); `results()` flushes.  Same numbers up to the fp32 summation order of the fused CLIP forward."""
        self.model = model
        self.device = model.no_mask_embed.weight.device
        if self.device.type != "cuda":
            raise RuntimeError("DeviceEvalLoop needs the model on the GPU; there is no CPU path")
        self.class_names = list(class_names)
        self.R = model.clip_model.geometry.image_resolution
        self.pre = GpuPreprocess(model.inp_size, self.R, self.device)
        self.convention = clip_mask_convention
        self.evaluator = DeviceClassification({i: n for i, n in enumerate(self.class_names)}, device=str(self.device))
        self.metricer = DeviceMetricer(self.class_names, metric_names)
        self.cod = DeviceCod()                                        # utils.calc_cod + its four Averagers (:70-109)
        self.timed = timed
        self.pipelined = pipelined
        self._tail_stream: Optional[torch.cuda.Stream] = None
        self._owed = None                                             # pipelined: (masks, pred, logits, gts, labels) of the batch whose tail is still to run
        self.events: List[list] = []
        self.images = 0
        self.last: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor, List[torch.Tensor]]] = None

    def _mark(self, marks: Optional[list]) -> None:
        if marks is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append((e, time.perf_counter()))

    @torch.no_grad()
    def step(self, images_u8: Sequence[torch.Tensor], gts_u8: Sequence[torch.Tensor], label_ids: torch.Tensor) -> None:
        """images_u8: B uint8 (h_i, w_i, 3) host tensors; gts_u8: B uint8 (h_i, w_i) host tensors -- PINNED, or the copy waits for
        the stream and the host loses its lead over the GPU; label_ids: (B,) int64 host tensor.  Queues one batch; returns
        without synchronising."""
        dev, marks = self.device, ([] if self.timed else None)
        self._mark(marks)
        imgs = [t.to(dev, non_blocking=True) for t in images_u8]                       # 1 byte per sample crosses PCIe, not 4
        gts = [t.to(dev, non_blocking=True) for t in gts_u8]
        # pageable memory would make this copy -- and with it the host -- wait for the stream (the whole previous step)
        lab_pin = torch.empty(label_ids.shape, dtype=torch.int64, pin_memory=True)
        lab_pin.copy_(label_ids)
        labels = lab_pin.to(dev, non_blocking=True)
        self._mark(marks)
        # N1 (datasets/wrappers.py:22-62): Resize 1024 bilinear + ImageNet normalise; Resize 336 bicubic + crop + OpenAI normalise
        inp = torch.cat([self.pre.sam_input(t) for t in imgs])
        clip_image = torch.cat([self.pre.clip_input(t) for t in imgs])
        clip_mask = self.pre.clip_mask(len(imgs), self.convention)
        self._mark(marks)
        if self.pipelined:
            return self._step_pipelined(inp, clip_image, clip_mask, gts, labels, marks)
        # the path, with the script's own glue (test_ovcos_maskdecoder_edge.py:102-113)
        pred_mask = self.model.infer_test(inp, clip_image, clip_mask)
        prob = torch.sigmoid(pred_mask)
        alpha = F.interpolate(prob, (self.R, self.R), mode="bilinear", align_corners=False)
        _, _, pred_1, score = self.model.clip_model(clip_image, alpha, train=False)
        self._mark(marks)
        # N2 (:113-136): top-1 / top-5 counters, then per image sigmoid -> resize to the mask's size -> uint8 -> six metrics
        self.cod.step(prob, torch.cat([self.pre.mask_input(t) for t in gts]))         # :105 calc_cod(pred_mask, batch['gt'])
        self.evaluator.process(score, labels)
        same = pred_1.to(torch.int64) == labels.to(torch.int64)                        # pre_cls == gt_cls, decided on the device
        masks_u8 = self.metricer.step_batch(pred_mask, gts, same)
        self._mark(marks)
        if marks is not None:
            self.events.append(marks)
        self.images += len(imgs)
        self.last = (pred_mask, pred_1, score, masks_u8)

    # ---- pipelined form ------------------------------------------------------------------------------------------------------------
    def _run_tail(self, owed, after: torch.cuda.Event) -> None:
        """N2 of one batch on the tail stream, behind `after` (the side-stream event that covers its masks AND its class scores)"""
        masks, pred, logits, gts, labels = owed
        tail = self._tail_stream
        tail.wait_event(after)
        with torch.cuda.stream(tail):
            self.cod.step(torch.sigmoid(masks), torch.cat([self.pre.mask_input(t) for t in gts]))
            self.evaluator.process(logits, labels)
            same = pred.to(torch.int64) == labels.to(torch.int64)
            self.metricer.step_batch(masks, gts, same)
        for t in (masks, pred, logits, labels, *gts):
            t.record_stream(tail)

    def _step_pipelined(self, inp, clip_image, clip_mask, gts, labels, marks) -> None:
        cas = self.model.cascade()
        if self._tail_stream is None:
            self._tail_stream = torch.cuda.Stream(device=self.device)
        masks, pred, logits = cas.cascade(inp, clip_image, clip_mask, pipelined=True)
        done = cas.batch_done_event()                                 # side stream: this batch's decoder is through -- and, in front of it,
        self._mark(marks)                                             # the fused CLIP forward that filled the PREVIOUS batch's pred / logits
        if self._owed is not None:
            self._run_tail(self._owed, done)
        self._owed = (masks, pred, logits, gts, labels)
        self._mark(marks)
        if marks is not None:
            self.events.append(marks)
        self.images += int(inp.shape[0])
        self.last = None

    def cod_results(self) -> Dict[str, float]:
        """the loop's second metric set: the means of `calc_cod`'s (sm, em, wfm, mae) over all images (val_metric1..4, :72-109);
        call after `results()`"""
        return self.cod.result()

    def results(self) -> Tuple[Dict[str, float], Dict[str, float]]:
        """-> (`metricer.show(num_bits=None)` dict, `evaluator.evaluate()` dict): the one read-back of the loop."""
        if self.pipelined and self._owed is not None:
            cas = self.model.cascade()
            cas.flush()                                               # the last batch's stage 2 (no-op when nothing is owed there)
            self._run_tail(self._owed, cas.results_ready_event())
            self._owed = None
        torch.cuda.synchronize(self.device)
        return self.metricer.show(num_bits=None), dict(self.evaluator.evaluate())

    def section_ms(self) -> Dict[str, float]:
        """Sum of HIP-event time per section over the recorded steps (after a synchronize)."""
        out = {k: 0.0 for k in SECTIONS}
        for marks in self.events:
            for k, (a, b) in zip(SECTIONS, zip(marks[:-1], marks[1:])):
                out[k] += a[0].elapsed_time(b[0])
        return out

    def section_host_ms(self) -> Dict[str, float]:
        """Host wall clock the issuing thread spent inside each section (a section whose host time approaches the step's GPU
        time is where the host waits for the device)."""
        out = {k: 0.0 for k in SECTIONS}
        for marks in self.events:
            for k, (a, b) in zip(SECTIONS, zip(marks[:-1], marks[1:])):
                out[k] += 1e3 * (b[1] - a[1])
        return out
