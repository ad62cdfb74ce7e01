"""Oracle: the input transform of the reference's datasets.  TEST INFRASTRUCTURE ONLY.

``dataset/_base.py:18-37`` (``make_default_transform``) composes torchvision transforms on a PIL image:

    T.Resize(int(image_size * resize_scale)) -> T.CenterCrop(image_size) -> T.RandomHorizontalFlip() -> T.ToTensor() -> T.Normalize(0.5, 0.5)

torchvision is a third-party dependency that is absent from /root/reference and from this image (the reference pins torch 1.11 /
torchvision 0.12, docker/torch/Dockerfile:1); its transforms on PIL images delegate the arithmetic to Pillow:

  * ``Resize(int)``: the SHORTER edge becomes ``size``, the longer ``int(size * long / short)``; then ``PIL.Image.resize(..., BILINEAR)``,
    i.e. Pillow's separable, anti-aliased resampling in 8-bit fixed point (``src/libImaging/Resample.c``: triangle filter whose support
    grows with the down-scaling factor, coefficients normalised in double precision and quantised to 22 fractional bits, a horizontal
    pass then a vertical pass, each rounding to uint8) -- restated below as ``resample_bilinear_u8``;
  * ``CenterCrop(S)``: top = int(round((H - S) / 2)), left = int(round((W - S) / 2));
  * ``RandomHorizontalFlip``: flip when ``torch.rand(1) < 0.5``;
  * ``ToTensor`` + ``Normalize(0.5, 0.5)``: (v / 255 - 0.5) / 0.5 as float32, channels first.

Pinning: ``resample_bilinear_u8`` is checked BIT-EXACTLY against Pillow itself (present in this image; the version is recorded in the
fixture tests/golden/image_pipeline.npz written by tools/make_golden.py, which also stores Pillow's outputs so that the GPU box
needs no Pillow).  The crop / flip / normalise steps follow torchvision's documented formulas; torchvision cannot be run here, so for
those three steps parity is unpinned by the reference and pinned to these formulas.
"""
import numpy as np

PRECISION_BITS = 32 - 8 - 2           # Resample.c: 8-bit images use 22 fractional bits


def resample_tables(in_size, out_size):
    """Per output sample: first input sample, number of taps, fixed-point taps (Resample.c ``precompute_coeffs`` + ``normalize_coeffs_8bpc``)
    for the bilinear (triangle, support 1) filter.  Returns (xmin [out], count [out], taps [out, ksize] int32)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int32)
    count = np.zeros(out_size, np.int32)
    taps = np.zeros((out_size, ksize), np.int32)
    inv = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = int(center - support + 0.5)
        lo = max(lo, 0)
        hi = int(center + support + 0.5)
        hi = min(hi, in_size)
        n = hi - lo
        k = np.zeros(ksize)
        total = 0.0
        for x in range(n):
            w = abs((x + lo - center + 0.5) * inv)
            w = 1.0 - w if w < 1.0 else 0.0
            k[x] = w
            total += w
        if total != 0.0:
            k[:n] /= total
        fixed = np.where(k < 0, -0.5 + k * (1 << PRECISION_BITS), 0.5 + k * (1 << PRECISION_BITS))
        xmin[xx], count[xx] = lo, n
        taps[xx] = np.trunc(fixed).astype(np.int64)
    return xmin, count, taps


def _pass(img, out_size, axis):
    """One separable pass along ``axis`` (0 = rows / vertical, 1 = columns / horizontal) of an [H, W, C] uint8 image."""
    xmin, count, taps = resample_tables(img.shape[axis], out_size)
    src = np.moveaxis(img.astype(np.int64), axis, 0)
    out = np.zeros((out_size,) + src.shape[1:], np.int64)
    for i in range(out_size):
        acc = (src[xmin[i]:xmin[i] + count[i]] * taps[i, :count[i]].reshape((-1,) + (1,) * (src.ndim - 1))).sum(0)
        out[i] = np.clip((acc + (1 << (PRECISION_BITS - 1))) >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis).astype(np.uint8)


def resample_bilinear_u8(img, out_h, out_w):
    """``PIL.Image.resize((out_w, out_h), BILINEAR)`` on an [H, W, C] uint8 array: horizontal pass first, then vertical; a pass whose
    size does not change is skipped (Resample.c ``ImagingResample``)."""
    if out_w != img.shape[1]:
        img = _pass(img, out_w, 1)
    if out_h != img.shape[0]:
        img = _pass(img, out_h, 0)
    return img


def resized_shape(h, w, size):
    """torchvision ``Resize(int)``: shorter edge -> size, longer edge -> int(size * long / short); unchanged if already there."""
    short, long = (w, h) if w <= h else (h, w)
    if short == size:
        return h, w
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def center_crop_origin(h, w, size):
    return int(round((h - size) / 2.0)), int(round((w - size) / 2.0))


def default_transform(img, image_size, resize_scale=1.0, flip=False, normalize=True):
    """The whole ``make_default_transform`` on one [H, W, 3] uint8 image with the flip decision given; float32 [3, S, S]."""
    oh, ow = resized_shape(img.shape[0], img.shape[1], int(image_size * resize_scale))
    img = resample_bilinear_u8(img, oh, ow)
    top, left = center_crop_origin(oh, ow, image_size)
    assert top >= 0 and left >= 0, 'images smaller than the crop are padded by torchvision: not a case the datasets produce'
    img = img[top:top + image_size, left:left + image_size]
    if flip:
        img = img[:, ::-1]
    x = img.astype(np.float32).transpose(2, 0, 1) / np.float32(255)
    if normalize:
        x = (x - np.float32(0.5)) / np.float32(0.5)
    return np.ascontiguousarray(x)
