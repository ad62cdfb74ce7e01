"""Oracle: filtered leaky ReLU (bias -> up-FIR -> gain/lrelu/clamp -> down-FIR).  TEST INFRASTRUCTURE ONLY.

Restates ``thirdparty/stylegan3_ops/ops/filtered_lrelu.py:115-147``
(``_filtered_lrelu_ref``), which composes the two other ops, and the sign-tensor
encoding of the native kernels (``filtered_lrelu.cu:1121-1160``: 2 bits per
upsampled element, 0 = pass, 1 = negative (slope applied), 2 = clamped (zero
gradient), packed 4 elements per byte, row width rounded up to 16 elements,
``filtered_lrelu.cpp:81-88``).
"""
import math
import numpy as np
import torch

from . import upfirdn2d as _up
from . import bias_act as _ba


def out_shape(x_shape, fu, fd, up, down, padding):
    """filtered_lrelu.py:129-133 / filtered_lrelu.cpp:57-73."""
    px0, px1, py0, py1 = _up.parse_padding(padding)
    fu_w, fu_h = _up.filter_size(fu)
    fd_w, fd_h = _up.filter_size(fd)
    n, c, h, w = x_shape
    ow = (w * up + (px0 + px1) - (fu_w - 1) - (fd_w - 1) + (down - 1)) // down
    oh = (h * up + (py0 + py1) - (fu_h - 1) - (fd_h - 1) + (down - 1)) // down
    return n, c, oh, ow


def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0,
                   gain=math.sqrt(2), slope=0.2, clamp=None, flip_filter=False):
    assert x.ndim == 4
    px0, px1, py0, py1 = _up.parse_padding(padding)
    y = _ba.bias_act(x, b)                                                                   # :136
    y = _up.upfirdn2d(y, fu, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)   # :137
    y = _ba.bias_act(y, act='lrelu', alpha=slope, gain=gain, clamp=clamp)                    # :138
    y = _up.upfirdn2d(y, fd, down=down, flip_filter=flip_filter)                             # :139
    assert tuple(y.shape) == out_shape(x.shape, fu, fd, up, down, padding)
    return y


def upsampled_pre_activation(x, fu, b, up, padding, gain, flip_filter=False):
    """The tensor whose signs the native op records: bias -> up-FIR -> *gain (fp32)."""
    px0, px1, py0, py1 = _up.parse_padding(padding)
    y = _ba.bias_act(x, b)
    y = _up.upfirdn2d(y, fu, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
    return y * gain


def sign_codes(v, slope, clamp):
    """Per-element code of the native op (filtered_lrelu.cu:1129-1140): 1 if v<0; 2 if |lrelu(v)|>clamp."""
    v = v.to(torch.float32)
    code = (v < 0).to(torch.uint8)
    a = torch.where(v < 0, v * slope, v)
    if clamp is not None:
        code = torch.where(a.abs() > clamp, torch.full_like(code, 2), code)
    return code


def pack_signs(code):
    """[N,C,H,W] codes -> uint8 [N,C,H,ceil16(W)/4], 2 bits per element, element x in
    bits 2*(x&3) of byte x>>2 (filtered_lrelu.cpp:81-88, filtered_lrelu.cu:1143-1156)."""
    n, c, h, w = code.shape
    w16 = (w + 15) & ~15
    buf = np.zeros((n, c, h, w16), dtype=np.uint8)
    buf[..., :w] = code.cpu().numpy()
    buf = buf.reshape(n, c, h, w16 // 4, 4)
    packed = buf[..., 0] | (buf[..., 1] << 2) | (buf[..., 2] << 4) | (buf[..., 3] << 6)
    return torch.from_numpy(packed.astype(np.uint8))
