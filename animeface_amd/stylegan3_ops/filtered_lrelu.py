"""Filtered leaky ReLU on MI355X: same public surface as the reference's
``thirdparty/stylegan3_ops/ops/filtered_lrelu.py`` (``filtered_lrelu``).

Forward: ``agf_filtered_lrelu`` (fused, one pass through LDS) when a specialised kernel exists for
the parameter set, else -- exactly like the reference when its plugin reports return code -1
(filtered_lrelu.py:217-223) -- the generic composition  bias add -> ``upfirdn2d`` (up) ->
``agf_filtered_lrelu_act`` (in place, writes the 2-bit sign tensor) -> ``upfirdn2d`` (down).
Backward: the same op with up<->down, fu<->fd, flipped filters, gain*up^2/down^2, no clamp, reading the
sign tensor at an offset (filtered_lrelu.py:233-262); so gradients of any order compose.
bf16 is accepted in addition to the reference's fp16/fp32.
"""
import warnings

import numpy as np
import torch

from .. import _lib
from . import upfirdn2d


def _get_filter_size(f):
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor)
    assert 1 <= f.ndim <= 2
    return f.shape[-1], f.shape[0]      # width, height


def _parse_padding(padding):
    if isinstance(padding, int):
        padding = [padding, padding]
    assert isinstance(padding, (list, tuple))
    assert all(isinstance(x, (int, np.integer)) for x in padding)
    padding = [int(x) for x in padding]
    if len(padding) == 2:
        px, py = padding
        padding = [px, px, py, py]
    px0, px1, py0, py1 = padding
    return px0, px1, py0, py1


def _dense(t):
    """contiguous in NCHW or channels-last order (what the kernels read at full width)"""
    return t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last)


def _f_desc(f):
    """(size[2], stride[2]) in the C ABI's convention: rank-1 = {taps, 0}; rank-2 = {fh, fw}."""
    if f.ndim == 1:
        return _lib._i32x2(int(f.shape[0]), 0), _lib._i64x2(int(f.stride(0)), 0)
    return _lib._i32x2(int(f.shape[0]), int(f.shape[1])), _lib._i64x2(int(f.stride(0)), int(f.stride(1)))


def _native_fused(x, fu, fd, b, si, up, down, px0, px1, py0, py1, sx, sy, gain, slope, clamp, flip, write_signs, ysum=None):
    """Counterpart of ``_plugin.filtered_lrelu`` (reference filtered_lrelu.cpp:10-203): returns (y, so, rc)."""
    N, C, xh, xw = x.shape
    b = b.contiguous() if b is not None else None       # None: no bias (the gradient pass; the bf16 kernel then runs its 2-D interpolation on the matrix pipe)
    fut_w, fut_h = int(fu.shape[-1]) - 1, int(fu.shape[0]) - 1
    fdt_w, fdt_h = int(fd.shape[-1]) - 1, int(fd.shape[0]) - 1
    cw = xw * up + (px0 + px1) - fut_w
    ch = xh * up + (py0 + py1) - fut_h
    if not (cw > fdt_w and ch > fdt_h):
        raise RuntimeError('upsampled buffer must be at least the size of downsampling filter')
    yw = (cw - fdt_w + (down - 1)) // down
    yh = (ch - fdt_h + (down - 1)) // down
    if yw <= 0 or yh <= 0:
        raise RuntimeError('output must be at least 1x1')
    y = torch.empty((N, C, yh, yw), dtype=x.dtype, device=x.device)
    so = None
    s = si
    mode = 0
    if write_signs:
        sw_active = yw * down - (down - 1) + fdt_w
        sh = yh * down - (down - 1) + fdt_h
        sw = (sw_active + 15) & ~15
        s = so = torch.empty((N, C, sh, sw >> 2), dtype=torch.uint8, device=x.device)
        mode = 1
    elif si is not None and si.numel():
        mode = 2
    fus, fust = _f_desc(fu)
    fds, fdst = _f_desc(fd)
    ssz = _lib._i32x2(int(s.shape[2]), int(s.shape[3])) if mode else _lib._i32x2(0, 0)
    rc = _lib.lib().agf_filtered_lrelu(
        _lib.ptr(x), _lib.ptr(fu), _lib.ptr(fd), _lib.ptr(b), _lib.ptr(s if mode else None), _lib.ptr(y), _lib.dtype_code(x),
        _lib.sizes4(x), _lib.strides4(x), _lib.sizes4(y), _lib.strides4(y), fus, fust, fds, fdst,
        ssz, _lib._i32x2(sx, sy), mode, up, down, px0, py0, gain, slope, clamp, int(bool(flip)), _lib.ptr(ysum), _lib.stream_ptr(x))
    if rc == _lib.AGF_ENOKERNEL:
        return None, None, -1
    _lib.check(rc, 'filtered_lrelu')
    return y, so, 0


def _native_act_(y, si, sx, sy, gain, slope, clamp, write_signs):
    """Counterpart of ``_plugin.filtered_lrelu_act_`` (reference filtered_lrelu.cpp:207-284): in place on y."""
    so = None
    s = si
    mode = 0
    if write_signs:
        sw = (y.shape[3] + 15) & ~15
        s = so = torch.empty((y.shape[0], y.shape[1], y.shape[2], sw >> 2), dtype=torch.uint8, device=y.device)
        mode = 1
    elif si is not None and si.numel():
        mode = 2
        if not si.is_contiguous() or si.dtype != torch.uint8 or si.dim() != 4:
            raise RuntimeError('signs must be a contiguous rank-4 uint8 tensor')
        if si.shape[0] != y.shape[0] or si.shape[1] != y.shape[1]:
            raise RuntimeError('signs must have same batch & channels as x')
    ssz = _lib._i32x2(int(s.shape[2]), int(s.shape[3])) if mode else _lib._i32x2(0, 0)
    rc = _lib.lib().agf_filtered_lrelu_act(_lib.ptr(y), _lib.ptr(s if mode else None), _lib.dtype_code(y),
                                           _lib.sizes4(y), _lib.strides4(y), ssz, _lib._i32x2(sx, sy), mode,
                                           gain, slope, clamp, _lib.stream_ptr(y))
    _lib.check(rc, 'filtered_lrelu_act_')
    return so


_filtered_lrelu_hip_cache = dict()


def _filtered_lrelu_hip(up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, clamp=None, flip_filter=False):
    assert isinstance(up, int) and up >= 1
    assert isinstance(down, int) and down >= 1
    px0, px1, py0, py1 = _parse_padding(padding)
    assert gain == float(gain) and gain > 0
    gain = float(gain)
    assert slope == float(slope) and slope >= 0
    slope = float(slope)
    assert clamp is None or (clamp == float(clamp) and clamp >= 0)
    clamp = float(clamp if clamp is not None else 'inf')
    key = (up, down, px0, px1, py0, py1, gain, slope, clamp, flip_filter)
    if key in _filtered_lrelu_hip_cache:
        return _filtered_lrelu_hip_cache[key]

    class FilteredLReluHip(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, fu, fd, b, si, sx, sy):
            assert isinstance(x, torch.Tensor) and x.ndim == 4
            _lib.require_gpu(x, 'filtered_lrelu')
            if x.dtype not in (torch.float16, torch.bfloat16, torch.float32):
                raise RuntimeError('x and b must be float16, bfloat16 or float32')
            if fu is None:
                fu = torch.ones([1, 1], dtype=torch.float32, device=x.device)
            if fd is None:
                fd = torch.ones([1, 1], dtype=torch.float32, device=x.device)
            assert 1 <= fu.ndim <= 2 and 1 <= fd.ndim <= 2
            if fu.dtype != torch.float32 or fd.dtype != torch.float32:
                raise RuntimeError('fu and fd must be float32')
            if up == 1 and fu.ndim == 1 and fu.shape[0] == 1:
                fu = fu.square()[None]
            if down == 1 and fd.ndim == 1 and fd.shape[0] == 1:
                fd = fd.square()[None]
            if b is None:
                b = torch.zeros([x.shape[1]], dtype=x.dtype, device=x.device)
            if b.dtype != x.dtype:
                raise RuntimeError('x and b must have the same dtype')
            if b.dim() != 1 or b.shape[0] != x.shape[1]:
                raise RuntimeError('b must be a vector with the same number of channels as x')
            have_si = si is not None and si.numel() > 0
            write_signs = (not have_si) and (x.requires_grad or b.requires_grad)
            strides = [x.stride(i) for i in range(x.ndim) if x.size(i) > 1]
            if any(a < c for a, c in zip(strides[:-1], strides[1:])):
                warnings.warn('low-performance memory layout detected in filtered_lrelu input', RuntimeWarning)

            y, so, rc = _native_fused(x, fu, fd, b, si if have_si else None, up, down, px0, px1, py0, py1, sx, sy,
                                      gain, slope, clamp, flip_filter, write_signs)
            if rc < 0:
                # generic composition; only the bit-packed sign tensor is kept for the gradient
                y = x.add(b.unsqueeze(-1).unsqueeze(-1))
                y = upfirdn2d.upfirdn2d(x=y, f=fu, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
                so = _native_act_(y, si if have_si else None, sx, sy, gain, slope, clamp, write_signs)
                y = upfirdn2d.upfirdn2d(x=y, f=fd, down=down, flip_filter=flip_filter)

            ctx.save_for_backward(fu, fd, (si if have_si else so))
            ctx.x_shape = x.shape
            ctx.y_shape = y.shape
            ctx.s_ofs = sx, sy
            return y

        @staticmethod
        def backward(ctx, dy):
            fu, fd, si = ctx.saved_tensors
            _, _, xh, xw = ctx.x_shape
            _, _, yh, yw = ctx.y_shape
            sx, sy = ctx.s_ofs
            dx = db = None
            if ctx.needs_input_grad[0] or ctx.needs_input_grad[3]:
                pp = [(fu.shape[-1] - 1) + (fd.shape[-1] - 1) - px0,
                      xw * up - yw * down + px0 - (up - 1),
                      (fu.shape[0] - 1) + (fd.shape[0] - 1) - py0,
                      xh * up - yh * down + py0 - (up - 1)]
                gg = gain * (up ** 2) / (down ** 2)
                ff = (not flip_filter)
                sx = sx - (fu.shape[-1] - 1) + px0
                sy = sy - (fu.shape[0] - 1) + py0
                if not torch.is_grad_enabled() and dy.dtype in (torch.float16, torch.bfloat16, torch.float32) and si is not None and si.numel():
                    # no graph is being recorded: run the gradient pass directly and let the kernel accumulate the bias gradient
                    # (sum of dx over n, h, w) while it stores dx -- one pass less over dx
                    dyc = dy if _dense(dy) else dy.contiguous()
                    ysum = torch.zeros(dy.shape[1], dtype=torch.float32, device=dy.device) if ctx.needs_input_grad[3] else None
                    fu2 = fd if fd.ndim == 2 or down > 1 or fd.shape[0] > 1 else fd.square()[None]
                    fd2 = fu if fu.ndim == 2 or up > 1 or fu.shape[0] > 1 else fu.square()[None]
                    dx, _, rc = _native_fused(dyc, fu2, fd2, None, si, down, up, pp[0], pp[1], pp[2], pp[3], sx, sy, gg, slope,
                                              float('inf'), ff, False, ysum=ysum)
                    if rc < 0:
                        dx = None
                    elif ysum is not None:
                        db = ysum.to(dy.dtype)
                if dx is None:
                    dx = _filtered_lrelu_hip(up=down, down=up, padding=pp, gain=gg, slope=slope, clamp=None,
                                             flip_filter=ff).apply(dy, fd, fu, None, si, sx, sy)
            if ctx.needs_input_grad[3] and db is None:
                db = dx.sum([0, 2, 3])
            return dx, None, None, db, None, None, None

    _filtered_lrelu_hip_cache[key] = FilteredLReluHip
    return FilteredLReluHip


def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, clamp=None,
                   flip_filter=False, impl='hip'):
    """bias -> upsample FIR (fu) -> *gain -> leaky ReLU -> clamp -> downsample FIR (fd)
    (reference filtered_lrelu.py:50-110; semantics filtered_lrelu.py:53-76)."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['hip', 'cuda']
    return _filtered_lrelu_hip(up=up, down=down, padding=padding, gain=gain, slope=slope, clamp=clamp,
                               flip_filter=flip_filter).apply(x, fu, fd, b, None, 0, 0)
