"""upfirdn2d on MI355X: same public surface as the reference's
``thirdparty/stylegan3_ops/ops/upfirdn2d.py`` (``setup_filter``, ``upfirdn2d``, ``filter2d``,
``upsample2d``, ``downsample2d``; arguments, defaults and error behaviour), every pass one
``agf_upfirdn2d`` launch.  The gradient is the same op with up<->down and the filter flipped
(reference upfirdn2d.py:245-263), so gradients of any order compose from the one kernel.

Extension: ``edge='clamp'`` (clamp-to-edge instead of zero fill) expresses the StyleGAN2 model's
``nn.Upsample(bilinear)`` exactly (implementations/StyleGAN2/model.py:56-58).
"""
import numpy as np
import torch

from .. import _lib


def _parse_scaling(scaling):
    if isinstance(scaling, int):
        scaling = [scaling, scaling]
    assert isinstance(scaling, (list, tuple))
    assert all(isinstance(x, int) for x in scaling)
    sx, sy = scaling
    assert sx >= 1 and sy >= 1
    return sx, sy


def _parse_padding(padding):
    if isinstance(padding, int):
        padding = [padding, padding]
    assert isinstance(padding, (list, tuple))
    assert all(isinstance(x, (int, np.integer)) for x in padding)
    padding = [int(x) for x in padding]
    if len(padding) == 2:
        padx, pady = padding
        padding = [padx, padx, pady, pady]
    padx0, padx1, pady0, pady1 = padding
    return padx0, padx1, pady0, pady1


def _get_filter_size(f):
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and f.ndim in [1, 2]
    fw, fh = int(f.shape[-1]), int(f.shape[0])
    assert fw >= 1 and fh >= 1
    return fw, fh


def setup_filter(f, device=torch.device('cpu'), normalize=True, flip_filter=False, gain=1, separable=None):
    """Reference upfirdn2d.py:64-108: returns fp32 [fh, fw] (or [taps] when separable: 1-D input with >= 8 taps)."""
    if f is None:
        f = 1
    f = torch.as_tensor(f, dtype=torch.float32)
    assert f.ndim in [0, 1, 2]
    assert f.numel() > 0
    if f.ndim == 0:
        f = f[np.newaxis]
    if separable is None:
        separable = (f.ndim == 1 and f.numel() >= 8)
    if f.ndim == 1 and not separable:
        f = f.ger(f)
    assert f.ndim == (1 if separable else 2)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    f = f * (gain ** (f.ndim / 2))
    return f.to(device=device)


def _launch(x, f2d, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain, edge):
    """One native pass (the counterpart of ``_plugin.upfirdn2d``, reference upfirdn2d.cpp:10-91).
    The caller-side checks reproduce the TORCH_CHECKs at upfirdn2d.cpp:13-34."""
    _lib.require_gpu(x, 'upfirdn2d')
    if f2d.device != x.device:
        raise RuntimeError('f must reside on the same device as x')
    if f2d.dtype != torch.float32:
        raise RuntimeError('f must be float32')
    if x.numel() == 0:
        raise RuntimeError('x has zero size')
    if x.dim() != 4:
        raise RuntimeError('x must be rank 4')
    if f2d.dim() != 2:
        raise RuntimeError('f must be rank 2')
    N, C, H, W = x.shape
    fh, fw = f2d.shape
    ow = (W * upx + padx0 + padx1 - fw + downx) // downx
    oh = (H * upy + pady0 + pady1 - fh + downy) // downy
    if ow < 1 or oh < 1:
        raise RuntimeError('output must be at least 1x1')
    cl = x.stride(1) == 1 and C > 1          # x.suggest_memory_format() (upfirdn2d.cpp:32)
    y = torch.empty((N, C, oh, ow), dtype=x.dtype, device=x.device,
                    memory_format=torch.channels_last if cl else torch.contiguous_format)
    rc = _lib.lib().agf_upfirdn2d(
        _lib.ptr(x), _lib.ptr(f2d), _lib.ptr(y), _lib.dtype_code(x),
        _lib.sizes4(x), _lib.strides4(x), _lib._i32x2(fh, fw), _lib._i64x2(*f2d.stride()),
        _lib.sizes4(y), _lib.strides4(y),
        upx, upy, downx, downy, padx0, pady0, int(bool(flip)), float(gain),
        _lib.EDGE_CLAMP if edge == 'clamp' else _lib.EDGE_ZERO, _lib.stream_ptr(x))
    _lib.check(rc, 'upfirdn2d')
    return y


def _fold_edges(g, rx, ry):
    """Adjoint of replicate padding: fold the rx / ry border columns / rows into the edge pixels."""
    if ry and g.shape[2] - 2 * ry == 1:
        g = g.sum(2, keepdim=True)
    elif ry:
        top = g[:, :, :ry + 1].sum(2, keepdim=True)
        bot = g[:, :, -ry - 1:].sum(2, keepdim=True)
        g = torch.cat([top, g[:, :, ry + 1:-ry - 1], bot], dim=2)
    if rx and g.shape[3] - 2 * rx == 1:
        g = g.sum(3, keepdim=True)
    elif rx:
        left = g[:, :, :, :rx + 1].sum(3, keepdim=True)
        right = g[:, :, :, -rx - 1:].sum(3, keepdim=True)
        g = torch.cat([left, g[:, :, :, rx + 1:-rx - 1], right], dim=3)
    return g


_upfirdn2d_hip_cache = dict()


def _upfirdn2d_hip(up=1, down=1, padding=0, flip_filter=False, gain=1, edge='zero'):
    """Autograd op factory, cached by parameters like the reference's ``_upfirdn2d_cuda`` (upfirdn2d.py:211-267)."""
    upx, upy = _parse_scaling(up)
    downx, downy = _parse_scaling(down)
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    key = (upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip_filter, gain, edge)
    if key in _upfirdn2d_hip_cache:
        return _upfirdn2d_hip_cache[key]

    class Upfirdn2dHip(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, f):
            assert isinstance(x, torch.Tensor) and x.ndim == 4
            if f is None:
                f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
            if f.ndim == 1 and f.shape[0] == 1:
                f = f.square().unsqueeze(0)                     # separable-1 -> full 1x1 (upfirdn2d.py:231-232)
            assert isinstance(f, torch.Tensor) and f.ndim in [1, 2]
            if f.ndim == 2:
                y = _launch(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip_filter, gain, edge)
            else:
                assert edge == 'zero'
                # x-pass with gain 1, then y-pass with the full gain (upfirdn2d.py:238-239)
                y = _launch(x, f.unsqueeze(0), upx, 1, downx, 1, padx0, padx1, 0, 0, flip_filter, 1.0, edge)
                y = _launch(y, f.unsqueeze(1), 1, upy, 1, downy, 0, 0, pady0, pady1, flip_filter, gain, edge)
            ctx.save_for_backward(f)
            ctx.x_shape = x.shape
            return y

        @staticmethod
        def backward(ctx, dy):
            f, = ctx.saved_tensors
            _, _, ih, iw = ctx.x_shape
            _, _, oh, ow = dy.shape
            fw, fh = _get_filter_size(f)
            p = [fw - padx0 - 1, iw * upx - ow * downx + padx0 - upx + 1,
                 fh - pady0 - 1, ih * upy - oh * downy + pady0 - upy + 1]
            dx = None
            if ctx.needs_input_grad[0]:
                if edge == 'zero':
                    dx = _upfirdn2d_hip(up=down, down=up, padding=p, flip_filter=(not flip_filter), gain=gain).apply(dy, f)
                else:
                    rx = (max(padx0, padx1, 0) + upx - 1) // upx + 1
                    ry = (max(pady0, pady1, 0) + upy - 1) // upy + 1
                    if torch.is_grad_enabled() or f.ndim != 2:
                        # differentiable form: adjoint on the replicate-extended domain, then fold the extension onto the edges
                        pe = [p[0] + rx * upx, p[1] + rx * upx, p[2] + ry * upy, p[3] + ry * upy]
                        g = _upfirdn2d_hip(up=down, down=up, padding=pe, flip_filter=(not flip_filter), gain=gain).apply(dy, f)
                        dx = _fold_edges(g, rx, ry)
                    else:
                        # fast form: ordinary zero-mode adjoint + a border-only kernel that adds the folded extension terms
                        dx = _launch(dy, f, downx, downy, upx, upy, p[0], p[1], p[2], p[3], not flip_filter, gain, 'zero')
                        rc = _lib.lib().agf_upfirdn2d_fold_border(
                            _lib.ptr(dy), _lib.ptr(f), _lib.ptr(dx), _lib.dtype_code(dy),
                            _lib.sizes4(dy), _lib.strides4(dy), _lib._i32x2(*f.shape), _lib._i64x2(*f.stride()),
                            _lib.sizes4(dx), _lib.strides4(dx), downx, downy, upx, upy, p[0], p[2],
                            int(not flip_filter), float(gain), rx, ry, _lib.stream_ptr(dy))
                        _lib.check(rc, 'upfirdn2d_fold_border')
            assert not ctx.needs_input_grad[1]
            return dx, None

    _upfirdn2d_hip_cache[key] = Upfirdn2dHip
    return Upfirdn2dHip


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl='hip', edge='zero'):
    """Pad, upsample, filter and downsample a batch of 2D images (reference upfirdn2d.py:112-156).

    ``impl`` is accepted for call-site compatibility ('cuda' and 'hip' both mean the native kernel);
    there is no 'ref' implementation in the product -- the CPU restatement lives in ``oracle/`` for tests."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['hip', 'cuda']
    return _upfirdn2d_hip(up=up, down=down, padding=padding, flip_filter=flip_filter, gain=gain, edge=edge).apply(x, f)


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl='hip'):
    """Reference upfirdn2d.py:271-303."""
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [padx0 + fw // 2, padx1 + (fw - 1) // 2, pady0 + fh // 2, pady1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl='hip', edge='zero'):
    """Reference upfirdn2d.py:307-342."""
    upx, upy = _parse_scaling(up)
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [padx0 + (fw + upx - 1) // 2, padx1 + (fw - upx) // 2, pady0 + (fh + upy - 1) // 2, pady1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy, impl=impl, edge=edge)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl='hip'):
    """Reference upfirdn2d.py:346-381."""
    downx, downy = _parse_scaling(down)
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [padx0 + (fw - downx + 1) // 2, padx1 + (fw - downx) // 2, pady0 + (fh - downy + 1) // 2, pady1 + (fh - downy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)
