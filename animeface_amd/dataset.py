"""Input transform of the reference's datasets as a GPU-side stage.

Mirrors ``dataset/_base.py:18-37`` (``make_default_transform(image_size, resize_scale=1., hflip=True, normalize=True)``:
torchvision ``Resize`` -> ``CenterCrop`` -> ``RandomHorizontalFlip`` -> ``ToTensor`` -> ``Normalize(0.5, 0.5)``), which the reference runs
per image on PIL objects inside DataLoader worker processes.  At MI355X training speed (thousands of images per second per GPU) that host
pipeline is the bottleneck, so here a whole batch of decoded uint8 images of one size, already in HBM, goes through two HIP kernels
(``agf_image_resample_rows`` / ``agf_image_finish``).  The arithmetic is Pillow's: its BILINEAR resize is separable, anti-aliased,
8-bit fixed point, and is reproduced bit-exactly (tests/test_hip_image_pipeline.py against Pillow's own outputs); decoding JPEG / PNG
files and the file datasets themselves stay outside this package.
"""
import functools

import numpy as np
import torch

from . import _lib, rng

_PRECISION_BITS = 22        # Pillow's 8-bit resampling keeps 22 fractional bits


@functools.lru_cache(maxsize=256)
def _tables(in_size, out_size, lo, hi):
    """Fixed-point triangle-filter taps of output samples [lo, hi) when ``in_size`` samples are resampled to ``out_size``:
    (first input sample [n], tap count [n], taps [n, ksize] int32), all as numpy arrays; vectorised over the samples."""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = fscale                                              # the bilinear filter has support 1, widened when shrinking
    ksize = int(np.ceil(support)) * 2 + 1
    centre = (np.arange(lo, hi, dtype=np.float64) + 0.5) * scale
    first = np.maximum((centre - support + 0.5).astype(np.int64), 0)          # C's (int) truncation of a non-negative value
    last = np.minimum((centre + support + 0.5).astype(np.int64), in_size)
    count = last - first
    pos = first[:, None] + np.arange(ksize)[None, :]
    w = np.abs((pos - centre[:, None] + 0.5) * (1.0 / fscale))
    w = np.where(w < 1.0, 1.0 - w, 0.0)
    w = np.where(np.arange(ksize)[None, :] < count[:, None], w, 0.0)
    # Pillow accumulates the normalisation sum tap by tap in double precision: the same left-to-right order
    total = np.zeros(len(centre))
    for j in range(ksize):
        total = total + w[:, j]
    w = np.where(total[:, None] != 0.0, w / np.where(total[:, None] != 0.0, total[:, None], 1.0), w)
    fixed = np.where(w < 0, -0.5 + w * (1 << _PRECISION_BITS), 0.5 + w * (1 << _PRECISION_BITS))
    taps = np.trunc(fixed).astype(np.int32)
    return first.astype(np.int32), count.astype(np.int32), taps, ksize


def _resized_shape(h, w, size):
    """torchvision ``Resize(int)``: the shorter edge becomes ``size``, the longer int(size * long / short)."""
    short, long_ = (w, h) if w <= h else (h, w)
    if short == size:
        return h, w
    other = int(size * long_ / short)
    return (other, size) if w <= h else (size, other)


class GpuTransform:
    """``transform(batch)``: uint8 [N, H, W, C] on the GPU -> float [N, C, S, S].  The flip decisions of a batch are ONE device-side draw
    (``rand(N) < 0.5`` from the package's generator, no host synchronisation); torchvision's RandomHorizontalFlip draws one CPU number per
    image instead -- same distribution, a different stream (pass ``flips`` to replay given decisions).  The resampling tables of an input
    size are built once and stay on the device."""

    def __init__(self, image_size, resize_scale=1., hflip=True, normalize=True, dtype=torch.float32):
        self.image_size, self.resize_to = int(image_size), int(image_size * resize_scale)
        self.hflip, self.normalize, self.dtype = hflip, normalize, dtype
        self._tables = {}          # (H, W, device) -> uploaded tap tables and the geometry derived from them

    def __call__(self, batch, flips=None):
        _lib.require_gpu(batch, 'image transform')
        if batch.dtype != torch.uint8 or batch.dim() != 4:
            raise RuntimeError('image transform: expected a uint8 tensor [N, H, W, C]')
        batch = batch.contiguous()
        N, H, W, C = batch.shape
        S = self.image_size
        oh, ow = _resized_shape(H, W, self.resize_to)
        top, left = int(round((oh - S) / 2.0)), int(round((ow - S) / 2.0))
        if top < 0 or left < 0:
            raise RuntimeError(f'image transform: {H}x{W} images resize to {oh}x{ow}, smaller than the {S}x{S} crop')
        dev = batch.device
        key = (H, W, str(dev))
        if key not in self._tables:
            hf, hc, ht, hk = _tables(W, ow, left, left + S)          # horizontal pass: only the columns the crop keeps
            vf, vc, vt, vk = _tables(H, oh, top, top + S)            # vertical pass: only the rows the crop keeps
            row0 = int(vf.min())
            rows = int((vf + vc).max()) - row0
            up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            self._tables[key] = (up(hf), up(hc), up(ht), hk, up(vf), up(vc), up(vt), vk, row0, rows)
        hf, hc, ht, hk, vf, vc, vt, vk, row0, rows = self._tables[key]
        if flips is None:
            flip = (rng.rand((N,), device=dev) < 0.5).to(torch.uint8) if self.hflip else torch.zeros(N, dtype=torch.uint8, device=dev)
        else:
            flip = torch.as_tensor([int(bool(f)) for f in flips], dtype=torch.uint8).to(dev)
        tmp = torch.empty((N, rows, S, C), dtype=torch.uint8, device=dev)
        out = torch.empty((N, C, S, S), dtype=self.dtype, device=dev)
        L = _lib.lib()
        rc = L.agf_image_resample_rows(_lib.ptr(batch), _lib.ptr(tmp), _lib.ptr(hf), _lib.ptr(hc), _lib.ptr(ht), hk,
                                       N, H, W, C, row0, rows, S, _lib.stream_ptr(batch))
        _lib.check(rc, 'image_resample_rows')
        rc = L.agf_image_finish(_lib.ptr(tmp), _lib.ptr(out), _lib.ptr(vf), _lib.ptr(vc), _lib.ptr(vt), vk, _lib.ptr(flip),
                                _lib.dtype_code(out), N, rows, row0, S, C, S, int(self.normalize), _lib.stream_ptr(batch))
        _lib.check(rc, 'image_finish')
        return out


def make_default_transform(image_size, resize_scale=1., hflip=True, normalize=True):
    """Same name and arguments as the reference's ``dataset/_base.py::make_default_transform``; the result takes uint8 batches on the GPU."""
    return GpuTransform(image_size, resize_scale, hflip, normalize)
