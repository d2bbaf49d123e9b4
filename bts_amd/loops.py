"""The pieces of the reference's driver loops that sit either side of the model, on the device kernels (SURVEY.md section 8f
rows 2-4).  Host-side mirrors of what the reference does in numpy / PIL per image; the arithmetic is in csrc/evalops.hip.

  online_eval        <- pytorch/bts_main.py:250-319   no-grad forward over the eval set, `compute_errors` per image, the 10-float
                        `eval_measures` sum, its all-reduce over the ranks, mean over the evaluated samples
  predict_payloads   <- pytorch/bts_test.py:107-128 + 179-185   no-grad forward over the test samples and the uint16 image that is
                        written as PNG (depth * 256 for kitti, * 1000 for nyu)
  TrainBatchPipeline <- pytorch/bts_dataloader.py:94-141, 190-235 + bts_main.py:443-445   decoded samples -> pinned staging ->
                        H2D on a copy stream one batch ahead -> crop / flip / augmentation / ToTensor / Normalize in one kernel

What the reference moves per evaluated image: five full-resolution f32 maps to the host (bts_test.py:119-124) or one plus the
ground truth through ~20 numpy passes (bts_main.py:264-299).  Here: nothing but the 10-float measure vector (online_eval) or
the uint16 payload, one copy per BATCH (predict_payloads).
"""
import random

import torch
import torch.distributed as dist

from . import dataops, evalops
from ._lib import BtsAmdError


def online_eval(model, eval_batches, dataset, min_depth_eval, max_depth_eval, do_kb_crop=False, garg_crop=False, eigen_crop=False,
                device=None, group=None, rank=0, log=None):
    """bts_main.py:250-319 with the per-image host work replaced by `bts_eval_errors`.

    eval_batches: iterable of dicts with 'image' [B,3,H,W] f32, 'focal' [B], 'depth' [B,(1,)Hg,Wg] f32 (metres) and
    'has_valid_depth' (bool, or a [B] tensor / list: the reference's eval loader is batch 1 and skips a sample without valid
    depth, :259-261; here invalid entries of a batch are skipped inside the kernel and not counted).
    Returns the reference's `eval_measures_cpu` (f32[10]: the nine mean metrics + the sample count) on rank 0 / without a process
    group, None elsewhere (:305-319).  The model must be in eval() mode (the caller's job, as in the reference, :507)."""
    dev = torch.device(device) if device is not None else next(model.parameters()).device
    em = torch.zeros(10, dtype=torch.float32, device=dev)                      # :251
    with torch.no_grad():
        for sample in eval_batches:
            hv = sample.get("has_valid_depth", True)
            if isinstance(hv, (bool, int)):
                if not hv:
                    continue                                                   # :259-261
                hv_t = None
            else:
                hv_t = torch.as_tensor(hv).reshape(-1).to(torch.uint8)
                if not bool(hv_t.any()):
                    continue
            image = sample["image"].to(dev, non_blocking=True)
            focal = sample["focal"].to(dev, non_blocking=True)
            gt = sample["depth"].to(dev, non_blocking=True).to(torch.float32)
            pred = model(image, focal)[4]                                      # :263
            # :268-299 for the whole batch: paste-back, clamping, validity + crop masks, nine metrics, eval_measures update
            evalops.compute_errors(pred.to(torch.float32), gt, min_depth_eval, max_depth_eval, dataset, do_kb_crop, garg_crop,
                                   eigen_crop, has_valid_depth=hv_t, eval_measures=em)
    if dist.is_initialized() and dist.get_world_size(group) > 1:               # :301-303
        dist.all_reduce(em, op=dist.ReduceOp.SUM, group=group)
    if dist.is_initialized() and dist.get_world_size(group) > 1 and rank != 0:
        return None
    out = em.cpu()                                                             # the ONLY device -> host copy of the evaluation
    cnt = out[9].item()
    if cnt > 0:
        out = out / cnt                                                        # :307-308 (divides the count slot too, as the reference does)
    if log is not None:
        log("Computing errors for {} eval samples".format(int(cnt)))
        log(", ".join("{:>7}".format(n) for n in evalops.EVAL_METRICS))
        log(", ".join("{:7.3f}".format(out[i]) for i in range(9)))
    return out


def predict_payloads(model, samples, dataset, device=None, keep_lpg=False):
    """bts_test.py's loop with the saved payload formed on the device: per batch one no-grad forward, `depth * 256` (kitti) or
    `* 1000` (nyu) truncated to uint16 by `bts_depth_to_u16`, ONE device -> host copy of 2 bytes per pixel.
    Returns a list of uint16 numpy arrays [H, W], one per image (what cv2.imwrite receives, bts_test.py:185); with keep_lpg
    also the four f32 LPG maps per image (only `--save_lpg` needs them, bts_test.py:187-226)."""
    dev = torch.device(device) if device is not None else next(model.parameters()).device
    payloads, lpgs = [], []
    with torch.no_grad():
        for sample in samples:
            image = sample["image"].to(dev, non_blocking=True)
            focal = sample["focal"].to(dev, non_blocking=True)
            outs = model(image, focal)
            u16 = evalops.depth_to_uint16(outs[4].to(torch.float32), dataset)
            host = u16.cpu().numpy()
            payloads.extend(host[i, 0] for i in range(host.shape[0]))
            if keep_lpg:
                maps = [o.to(torch.float32).cpu().numpy() for o in outs[:4]]
                lpgs.extend(tuple(m[i, 0] for m in maps) for i in range(host.shape[0]))
    return (payloads, lpgs) if keep_lpg else payloads


class TrainBatchPipeline:
    """Decoded training samples -> the reference's collated batch ('image' normalised f32 [B,3,H,W], 'depth' f32 [B,1,H,W],
    'focal' f64 [B]) resident on the device, one batch AHEAD of the consumer.

    The reference runs `DataLoadPreprocess.__getitem__` per sample on `num_threads` CPU workers (PIL crop / flip / numpy gamma,
    brightness, colour, ToTensor, Normalize), collates, and copies image / depth / focal to the GPU inside the step
    (bts_main.py:443-445).  Here the host only decodes: `source` yields per-sample (image_u8 [Hs,Ws,3] uint8 RGB, depth_raw
    [Hs,Ws] int32 payload of the 16-bit PNG, focal) AFTER the steps that stay on the host in the reference's order -- the kb
    crop (:107-113), the NYU boundary crop (:116-118) and the optional random rotation (:120-123: PIL bilinear / nearest resampling,
    not restated on the device): `host_geometry` below.
    Per batch: the draws of random_crop / train_preprocess / augment_image on the host in the reference's order
    (`dataops.draw_train_params`), raw bytes into PINNED staging buffers, an asynchronous H2D copy on a side stream (7 bytes
    per source pixel instead of 16 per cropped pixel), and one `bts_preprocess_train` launch on that stream.  `__next__` hands
    over batch i after making the consumer's stream wait for its event, and has batch i+1 already in flight."""

    def __init__(self, source, batch, height, width, dataset, device="cuda", depth=2):
        self.source, self.batch, self.height, self.width, self.dataset = iter(source), int(batch), int(height), int(width), dataset
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise BtsAmdError("TrainBatchPipeline runs on an MI355X (HIP) device only")
        self.stream = torch.cuda.Stream(device=self.dev)
        self.depth = max(1, int(depth))
        self.slots, self.queue, self._slot = [], [], 0
        self.done = False

    def _staging(self, Hs, Ws):
        s = self.slots[self._slot] if self._slot < len(self.slots) else None
        if s is None or tuple(s[0].shape[1:3]) != (Hs, Ws):
            s = (torch.empty((self.batch, Hs, Ws, 3), dtype=torch.uint8).pin_memory(),
                 torch.empty((self.batch, Hs, Ws), dtype=torch.int32).pin_memory(),
                 torch.empty((self.batch, Hs, Ws, 3), dtype=torch.uint8, device=self.dev),
                 torch.empty((self.batch, Hs, Ws), dtype=torch.int32, device=self.dev), torch.cuda.Event())
            if self._slot < len(self.slots):
                self.slots[self._slot] = s
            else:
                self.slots.append(s)
        self._slot = (self._slot + 1) % (self.depth + 1)
        return s

    def _fill(self):
        items = []
        for _ in range(self.batch):
            try:
                items.append(next(self.source))
            except StopIteration:
                self.done = True
                break
        if len(items) < self.batch:              # the reference's loader drops nothing, but a short last batch has another shape
            if not items:
                return
            raise BtsAmdError("TrainBatchPipeline: the source ended inside a batch (%d of %d samples)" % (len(items), self.batch))
        Hs, Ws = items[0][0].shape[:2]
        pin_i, pin_d, dev_i, dev_d, free_evt = self._staging(Hs, Ws)
        free_evt.synchronize()                   # the launch that last read this slot's device buffers has finished
        params, focals = [], []
        for b, (img, dep, focal) in enumerate(items):
            if tuple(img.shape) != (Hs, Ws, 3) or tuple(dep.shape) != (Hs, Ws):
                raise BtsAmdError("TrainBatchPipeline: samples of one batch must share a source size")
            pin_i[b].copy_(torch.as_tensor(img))
            pin_d[b].copy_(torch.as_tensor(dep))
            params.append(dataops.draw_train_params(Hs, Ws, self.height, self.width, self.dataset))
            focals.append(float(focal))
        with torch.cuda.stream(self.stream):
            dev_i.copy_(pin_i, non_blocking=True)
            dev_d.copy_(pin_d, non_blocking=True)
            image, depth = dataops.preprocess_train(dev_i, dev_d, params, self.height, self.width, self.dataset)
            free_evt.record(self.stream)
            ready = torch.cuda.Event()
            ready.record(self.stream)
        self.queue.append(({"image": image, "depth": depth, "focal": torch.tensor(focals, dtype=torch.float64).to(self.dev, non_blocking=True)},
                           ready, params))

    def __iter__(self):
        return self

    def __next__(self):
        while len(self.queue) < self.depth and not self.done:
            self._fill()
        if not self.queue:
            raise StopIteration
        batch, ready, params = self.queue.pop(0)
        torch.cuda.current_stream(self.dev).wait_event(ready)
        for t in batch.values():
            t.record_stream(torch.cuda.current_stream(self.dev))
        batch["aug_params"] = params
        if not self.done:
            self._fill()                          # batch i+1 goes in flight while the consumer works on batch i
        return batch


def host_geometry(image_pil, depth_pil, dataset, do_kb_crop, do_random_rotate, degree):
    """The host-side steps in front of the device pipeline, in the reference's order (bts_dataloader.py:107-125): the KITTI
    benchmark crop to 352 x 1216, the fixed NYU boundary crop (43, 45, 608, 472), and the random rotation (PIL resampling:
    bilinear for the image -- `rotate_image`'s default, :187-189 -- nearest for the depth).  Returns the PIL pair; consumes
    `random.random()` exactly when the reference does."""
    if do_kb_crop:
        height, width = image_pil.height, image_pil.width
        top, left = int(height - 352), int((width - 1216) / 2)
        depth_pil = depth_pil.crop((left, top, left + 1216, top + 352))
        image_pil = image_pil.crop((left, top, left + 1216, top + 352))
    if dataset == "nyu":
        depth_pil = depth_pil.crop((43, 45, 608, 472))
        image_pil = image_pil.crop((43, 45, 608, 472))
    if do_random_rotate:
        from PIL import Image
        angle = (random.random() - 0.5) * 2 * degree
        image_pil = image_pil.rotate(angle, resample=Image.BILINEAR)
        depth_pil = depth_pil.rotate(angle, resample=Image.NEAREST)
    return image_pil, depth_pil
