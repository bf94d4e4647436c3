"""Device-side input pipeline: the reference's torchvision transform stacks
(utils/data_utils.py:21-81, ``get_transforms``) executed on the GPU on raw uint8 batches, plus an asynchronous
host->device loader.  The reference decodes and resizes every image with PIL inside DataLoader worker processes
and then blocks on ``images.to(device)`` (experiments/mhla_pretrained.py:357-358); here the host only hands over
uint8 HWC bytes (pinned memory, a copy stream, double buffering) and ONE kernel pair does crop / flip /
Pillow-exact bilinear resize / ToTensor / Normalize (csrc/image.hip) while the previous step computes.

Random parameters (crop origin, flip, RandomResizedCrop box) follow torchvision's distributions but are drawn
from a numpy generator owned by the transform: torch's global RNG stream order inside torchvision is not
reproduced (random augmentation has no parity target).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, Iterable, Iterator, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _abi
from . import kernels as K
from . import streams

# CUs the loader's segmentation stream may occupy (of 256; read once).  Measured at the cfg3 batch (bench.py --config
# cfg3 --slic --slic-cus N): see DESIGN.md, round 4.
SEGMENTER_CUS = int(os.environ.get("FAVIT_SEGMENTER_CUS", "128"))

CIFAR10_MEAN, CIFAR10_STD = (0.4914, 0.4822, 0.4465), (0.2470, 0.2435, 0.2616)
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def _resized_size(h: int, w: int, size: int) -> Tuple[int, int]:
    """torchvision Resize(int): the smaller edge becomes `size`, the other int(size * long / short)."""
    if h <= w:
        return size, int(size * w / h)
    return int(size * h / w), size


class DeviceTransform:
    """One of the reference's transform stacks, run by favit_image_transform.

    kind: 'cifar10_train'  RandomCrop(32, padding=4) -> RandomHorizontalFlip -> Resize(S) -> ToTensor -> Normalize
          'imagenet_train' RandomResizedCrop(S) -> RandomHorizontalFlip -> ToTensor -> Normalize
          'resize'         Resize(S) -> ToTensor -> Normalize                       (cifar10 / default test)
          'resize_flip'    Resize(S) -> RandomHorizontalFlip -> ToTensor -> Normalize   (default train)
          'imagenet_test'  Resize(int(1.14 S)) -> CenterCrop(S) -> ToTensor -> Normalize
    """

    def __init__(self, kind: str, img_size: int, mean: Sequence[float], std: Sequence[float], seed: int = 0):
        if kind not in ("cifar10_train", "imagenet_train", "resize", "resize_flip", "imagenet_test"):
            raise ValueError(f"unknown transform kind {kind!r}")
        self.kind, self.S = kind, int(img_size)
        self.mean = (C.c_float * 3)(*[float(m) for m in mean])
        self.std = (C.c_float * 3)(*[float(s) for s in std])
        self.rng = np.random.RandomState(seed)

    # ---- per-image parameter rows (see include/favit.h: favit_image_transform) ----
    def params(self, B: int, H: int, W: int) -> np.ndarray:
        S, r = self.S, self.rng
        p = np.zeros((B, 12), dtype=np.int32)
        if self.kind == "cifar10_train":
            pad, cs = 4, 32
            if H + 2 * pad < cs or W + 2 * pad < cs:
                raise ValueError("cifar10_train expects images of at least 24x24")
            p[:, 0] = r.randint(0, H + 2 * pad - cs + 1, B)
            p[:, 1] = r.randint(0, W + 2 * pad - cs + 1, B)
            p[:, 2], p[:, 3], p[:, 4] = cs, cs, pad
            p[:, 5], p[:, 6] = _resized_size(cs, cs, S)
            p[:, 9] = r.rand(B) < 0.5                      # flip BEFORE the resize (torchvision order)
        elif self.kind == "imagenet_train":
            for b in range(B):                             # RandomResizedCrop.get_params: scale (0.08, 1), ratio (3/4, 4/3)
                area = H * W
                box = None
                for _ in range(10):
                    ta = area * r.uniform(0.08, 1.0)
                    ar = math.exp(r.uniform(math.log(3.0 / 4.0), math.log(4.0 / 3.0)))
                    w, h = int(round(math.sqrt(ta * ar))), int(round(math.sqrt(ta / ar)))
                    if 0 < w <= W and 0 < h <= H:
                        box = (r.randint(0, H - h + 1), r.randint(0, W - w + 1), h, w)
                        break
                if box is None:                            # fallback: central crop with a clamped aspect ratio
                    ir = W / H
                    if ir < 3.0 / 4.0:
                        w, h = W, int(round(W / (3.0 / 4.0)))
                    elif ir > 4.0 / 3.0:
                        h, w = H, int(round(H * (4.0 / 3.0)))
                    else:
                        w, h = W, H
                    box = ((H - h) // 2, (W - w) // 2, h, w)
                p[b, 0:4] = box
            p[:, 5], p[:, 6] = S, S
            p[:, 10] = r.rand(B) < 0.5                     # flip AFTER the resize
        elif self.kind in ("resize", "resize_flip"):
            rh, rw = _resized_size(H, W, S)
            if (rh, rw) != (S, S):
                raise ValueError("Resize(S) of a non-square image gives a non-square tensor; use imagenet_test")
            p[:, 2], p[:, 3], p[:, 5], p[:, 6] = H, W, rh, rw
            if self.kind == "resize_flip":
                p[:, 10] = r.rand(B) < 0.5
        else:                                              # imagenet_test
            rh, rw = _resized_size(H, W, int(S * 1.14))
            p[:, 2], p[:, 3], p[:, 5], p[:, 6] = H, W, rh, rw
            p[:, 7], p[:, 8] = int(round((rh - S) / 2.0)), int(round((rw - S) / 2.0))
        self.check_params(p)
        return p

    MAX_TAPS = 64          # csrc/image.hip: filter taps per output position the resampling kernels hold

    @classmethod
    def check_params(cls, p: np.ndarray) -> None:
        """Pillow's triangle filter spans 2 * max(1, crop / resized) source pixels: beyond MAX_TAPS the kernels would
        truncate the window and silently stop being bit-identical to Pillow, so such a row is refused here (a crop box
        downscaled more than ~31x: a very large source image or crop)."""
        for ax, (c, r) in enumerate(((p[:, 2], p[:, 5]), (p[:, 3], p[:, 6]))):
            scale = np.maximum(1.0, c.astype(np.float64) / np.maximum(r, 1))
            taps = np.ceil(2.0 * scale).astype(np.int64) + 2
            if (taps > cls.MAX_TAPS).any():
                b = int(np.argmax(taps))
                raise ValueError(f"DeviceTransform: image {b} is downscaled {scale[b]:.1f}x along axis {ax} "
                                 f"({int(c[b])} -> {int(r[b])} pixels): {int(taps[b])} filter taps exceed the kernels' "
                                 f"{cls.MAX_TAPS}; resize such sources on the host first")

    def __call__(self, batch_u8: torch.Tensor, params: Optional[np.ndarray] = None, want_bytes: bool = False):
        """batch_u8: uint8 [B, H, W, C] on the GPU -> fp32 [B, C, S, S] (and the resized bytes if want_bytes)."""
        K.require_gpu(batch_u8)
        if batch_u8.dtype != torch.uint8 or batch_u8.dim() != 4:
            raise TypeError("DeviceTransform expects a uint8 [B, H, W, C] batch")
        batch_u8 = batch_u8.contiguous()
        B, H, W, Cc = batch_u8.shape
        if params is None:
            params = self.params(B, H, W)
        else:
            self.check_params(np.asarray(params))
        prm = torch.from_numpy(np.ascontiguousarray(params, dtype=np.int32)).to(batch_u8.device, non_blocking=True)
        ch_max = int(params[:, 2].max())
        S = self.S
        tmp = torch.empty((B, ch_max, S, Cc), dtype=torch.uint8, device=batch_u8.device)
        out = torch.empty((B, Cc, S, S), dtype=torch.float32, device=batch_u8.device)
        u8 = torch.empty((B, S, S, Cc), dtype=torch.uint8, device=batch_u8.device) if want_bytes else None
        _abi.check(_abi.lib().favit_image_transform(K._p(batch_u8), K._p(tmp), K._p(out), K._p(u8), K._p(prm), B, H, W, Cc,
                                                    ch_max, S, self.mean, self.std, K._st()), "favit_image_transform")
        return (out, u8) if want_bytes else out


def get_transforms(dataset_name: str, img_size: int = 224, seed: int = 0) -> Dict[str, DeviceTransform]:
    """Mirror of the reference's get_transforms (utils/data_utils.py:21-81): {'train', 'test'} device transforms."""
    name = dataset_name.lower()
    if name == "cifar10":
        return {"train": DeviceTransform("cifar10_train", img_size, CIFAR10_MEAN, CIFAR10_STD, seed),
                "test": DeviceTransform("resize", img_size, CIFAR10_MEAN, CIFAR10_STD, seed)}
    if name == "imagenet":
        return {"train": DeviceTransform("imagenet_train", img_size, IMAGENET_MEAN, IMAGENET_STD, seed),
                "test": DeviceTransform("imagenet_test", img_size, IMAGENET_MEAN, IMAGENET_STD, seed)}
    half = (0.5, 0.5, 0.5)
    return {"train": DeviceTransform("resize_flip", img_size, half, half, seed),
            "test": DeviceTransform("resize", img_size, half, half, seed)}


class DeviceLoader:
    """Iterates (images fp32 [B,C,S,S], labels int64 [B]) on the GPU from an iterable of HOST batches
    (uint8 [B,H,W,C] array / tensor, integer labels).  Batch k+1 is copied (pinned staging buffers, a dedicated
    copy stream) and transformed while the consumer computes on batch k: no host-side blocking .to(device)."""

    def __init__(self, host_batches: Iterable, transform: DeviceTransform, device: Optional[torch.device] = None,
                 segmenter=None):
        """segmenter (optional): a models.sppp.SuperpixelSegmentation (``model.segmentation`` of the SPPP models).  The
        label maps of batch k+1 are then computed by the device SLIC on the loader's preparation stream while the
        consumer trains on batch k, and installed (``set_label_maps``) when the batch is yielded -- the reference
        segments inside forward (models/sppp_mhla.py:278), on the critical path of every step.

        The preparation stream is confined to ``SEGMENTER_CUS`` compute units; such a stream is a BLOCKING stream (the HIP
        call has no flag), i.e. it synchronises with the default (null) stream in both directions.  A consumer that
        wants the overlap therefore runs its step under ``with torch.cuda.stream(loader.compute_stream):`` (a stream
        of torch's non-blocking pool; cfg3 batch: 3.38 ms per step + segmentation against 4.07 ms back to back); on
        the default stream the results are the same and the two simply run one after the other."""
        self.src, self.tf = host_batches, transform
        self.dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        self.segmenter = segmenter
        # the segmentation stream is confined to SEGMENTER_CUS compute units (streams.py): its grids would otherwise
        # fill every CU and the consumer's short launches would queue behind them
        self.prep_stream = streams.cu_masked_stream(SEGMENTER_CUS, self.dev) if segmenter is not None else None
        self.compute_stream = torch.cuda.Stream(device=self.dev) if segmenter is not None else None
        self._pin = [None, None]

    def __len__(self):
        return len(self.src)

    def _stage(self, slot: int, imgs, labels):
        imgs = torch.as_tensor(np.asarray(imgs)) if not torch.is_tensor(imgs) else imgs
        labels = torch.as_tensor(np.asarray(labels), dtype=torch.int64) if not torch.is_tensor(labels) else labels.to(torch.int64)
        if imgs.is_pinned() and labels.is_pinned():
            # The producer already wrote into page-locked memory: no staging copy.  CONTRACT: the asynchronous
            # host-to-device copy reads that memory until the batch has been yielded, so the producer must not
            # rewrite a pinned buffer before the loader has yielded the batch made from it (hand out a fresh or a
            # rotated buffer per batch; bench.py --host-input rotates four).
            src = (imgs, labels)
        else:
            buf = self._pin[slot]
            if buf is None or buf[0].shape != imgs.shape or buf[1].shape != labels.shape:
                buf = [torch.empty(imgs.shape, dtype=torch.uint8).pin_memory(), torch.empty(labels.shape, dtype=torch.int64).pin_memory(), None]
                self._pin[slot] = buf
            if buf[2] is not None:
                buf[2].synchronize()                      # the previous copy out of this staging buffer has finished
            buf[0].copy_(imgs)
            buf[1].copy_(labels)
            src = (buf[0], buf[1])
        with torch.cuda.stream(self.copy_stream):
            d_img = src[0].to(self.dev, non_blocking=True)
            d_lab = src[1].to(self.dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        if src[0] is not imgs:
            self._pin[slot][2] = ev
        if self.segmenter is not None:
            # transform + SLIC of this (next) batch now, on a stream of their own, under the consumer's current step
            with torch.cuda.stream(self.prep_stream):
                self.prep_stream.wait_event(ev)
                x = self.tf(d_img)
                maps = self.segmenter.segment_device(x)
                derived = self.segmenter.derive_for(maps)        # patch mapping + centroids: functions of the maps alone
                d_img.record_stream(self.prep_stream)
                ev2 = torch.cuda.Event()
                ev2.record(self.prep_stream)
            return (x, maps, derived), d_lab, ev2
        return d_img, d_lab, ev

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        it = iter(self.src)
        nxt = None
        slot = 0
        try:
            first = next(it)
        except StopIteration:
            return
        nxt = self._stage(slot, *first)
        for batch in it:
            cur, slot = nxt, slot ^ 1
            nxt = self._stage(slot, *batch)               # H2D of batch k+1 overlaps the consumer's work on batch k
            yield self._finish(cur)
        yield self._finish(nxt)

    def _finish(self, staged):
        d_img, d_lab, ev = staged
        cs = torch.cuda.current_stream(self.dev)
        cs.wait_event(ev)                                 # device-side wait, the host does not block
        d_lab.record_stream(cs)
        if self.segmenter is not None:
            x, maps, derived = d_img
            x.record_stream(cs)
            maps.record_stream(cs)
            for tens in derived.values():
                for t_ in tens:
                    t_.record_stream(cs)
            # the model's next segment() call returns them; a captured step reads the installed tensors in place
            if getattr(self.segmenter, "_captured", False) and self.segmenter._maps is not None:
                self.segmenter.update_label_maps(maps, derived)
            else:
                self.segmenter.set_label_maps(maps, derived)
            return x, d_lab
        d_img.record_stream(cs)
        return self.tf(d_img), d_lab
