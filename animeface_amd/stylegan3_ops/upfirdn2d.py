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


def _xy_factors(value):
    """An up / down factor as (x, y): one int for both axes or a pair of ints, each at least 1 (anything else: AssertionError, as the
    reference's argument checks raise, upfirdn2d.py:27-35)."""
    pair = (value, value) if isinstance(value, int) else value
    assert isinstance(pair, (list, tuple)) and all(isinstance(v, int) for v in pair)
    fx, fy = pair
    assert min(fx, fy) >= 1
    return fx, fy


def _lrtb_padding(value):
    """Padding as (left, right, top, bottom): one int for all four sides, (x, y) for both sides of an axis, or the four values
    (reference upfirdn2d.py:38-48); numpy integers are accepted."""
    sides = (value, value) if isinstance(value, int) else value
    assert isinstance(sides, (list, tuple)) and all(isinstance(v, (int, np.integer)) for v in sides)
    sides = tuple(int(v) for v in sides)
    if len(sides) == 2:
        sides = (sides[0], sides[0], sides[1], sides[1])
    left, right, top, bottom = sides
    return left, right, top, bottom


def _taps_wh(f):
    """(width, height) of a filter tensor: [taps] is separable (the same taps on both axes), [fh, fw] is full, None is the identity."""
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and 1 <= f.ndim <= 2
    w, h = int(f.shape[-1]), int(f.shape[0])
    assert min(w, h) >= 1
    return w, h


def setup_filter(f, device=torch.device('cpu'), normalize=True, flip_filter=False, gain=1, separable=None):
    """Reference upfirdn2d.py:64-108: returns fp32 [fh, fw] (or [taps] when separable: 1-D input with >= 8 taps unless told otherwise)."""
    taps = torch.as_tensor(1 if f is None else f, dtype=torch.float32)
    assert taps.ndim <= 2 and taps.numel() > 0
    if taps.ndim == 0:
        taps = taps.reshape(1)
    keep_1d = (taps.ndim == 1 and taps.numel() >= 8) if separable is None else separable
    if taps.ndim == 1 and not keep_1d:
        taps = torch.outer(taps, taps)
    assert taps.ndim == (1 if keep_1d else 2)
    if normalize:
        taps = taps / taps.sum()
    if flip_filter:
        taps = taps.flip(tuple(range(taps.ndim)))
    return (taps * gain ** (taps.ndim / 2)).to(device=device)


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


def _launch_add(x, f2d, addend, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain):
    """``_launch`` (zero edges) + ``addend`` in the same pass (``agf_upfirdn2d_add``); None where the adding kernel does not take the call."""
    N, C, H, W = x.shape
    fh, fw = f2d.shape
    ow = (W * upx + padx0 + padx1 - fw + downx) // downx
    oh = (H * upy + pady0 + pady1 - fh + downy) // downy
    if not (x.is_cuda and x.dtype in (torch.bfloat16, torch.float32) and (upx, upy, downx, downy, fh, fw) == (2, 2, 1, 1, 4, 4)
            and C % (8 if x.dtype == torch.bfloat16 else 4) == 0 and x.is_contiguous(memory_format=torch.channels_last)
            and tuple(addend.shape) == (N, C, oh, ow) and addend.dtype == x.dtype and addend.is_contiguous(memory_format=torch.channels_last)):
        return None
    y = torch.empty((N, C, oh, ow), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    rc = _lib.lib().agf_upfirdn2d_add(
        _lib.ptr(x), _lib.ptr(f2d), _lib.ptr(y), _lib.ptr(addend), _lib.dtype_code(x),
        _lib.sizes4(x), _lib.strides4(x), _lib._i32x2(fh, fw), _lib._i64x2(*f2d.stride()),
        _lib.sizes4(y), _lib.strides4(y), upx, upy, downx, downy, padx0, pady0, int(bool(flip)), float(gain), _lib.EDGE_ZERO, _lib.stream_ptr(x))
    if rc == _lib.AGF_ENOKERNEL:
        return None
    _lib.check(rc, 'upfirdn2d_add')
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
    """Autograd op factory, one class per parameter set (cached, as the reference's ``_upfirdn2d_cuda`` is: upfirdn2d.py:211-267)."""
    ux, uy = _xy_factors(up)
    dnx, dny = _xy_factors(down)
    pl, pr, pt, pb = _lrtb_padding(padding)
    key = (ux, uy, dnx, dny, pl, pr, pt, pb, flip_filter, gain, edge)
    op = _upfirdn2d_hip_cache.get(key)
    if op is not None:
        return op

    class Upfirdn2dHip(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, f):
            assert isinstance(x, torch.Tensor) and x.ndim == 4
            taps = torch.ones([1, 1], dtype=torch.float32, device=x.device) if f is None else f
            if taps.ndim == 1 and taps.shape[0] == 1:
                taps = taps.square().unsqueeze(0)                       # one separable tap = a full 1x1 filter (upfirdn2d.py:231-232)
            assert isinstance(taps, torch.Tensor) and 1 <= taps.ndim <= 2
            if taps.ndim == 2:
                y = _launch(x, taps, ux, uy, dnx, dny, pl, pr, pt, pb, flip_filter, gain, edge)
            else:
                assert edge == 'zero'
                # separable: a row pass with unit gain, then a column pass that carries the gain (upfirdn2d.py:238-239)
                rows = _launch(x, taps.unsqueeze(0), ux, 1, dnx, 1, pl, pr, 0, 0, flip_filter, 1.0, edge)
                y = _launch(rows, taps.unsqueeze(1), 1, uy, 1, dny, 0, 0, pt, pb, flip_filter, gain, edge)
            ctx.save_for_backward(taps)
            ctx.in_hw = (x.shape[2], x.shape[3])
            return y

        @staticmethod
        def backward(ctx, dy):
            (taps,) = ctx.saved_tensors
            in_h, in_w = ctx.in_hw
            out_h, out_w = dy.shape[2], dy.shape[3]
            tw, th = _taps_wh(taps)
            # the adjoint is the same op with up <-> down, the filter flipped and this padding (upfirdn2d.py:245-263)
            adj = [tw - pl - 1, in_w * ux - out_w * dnx + pl - ux + 1,
                   th - pt - 1, in_h * uy - out_h * dny + pt - uy + 1]
            dx = None
            if ctx.needs_input_grad[0]:
                if edge == 'zero':
                    dx = _upfirdn2d_hip(up=down, down=up, padding=adj, flip_filter=(not flip_filter), gain=gain).apply(dy, taps)
                else:
                    rx = (max(pl, pr, 0) + ux - 1) // ux + 1
                    ry = (max(pt, pb, 0) + uy - 1) // uy + 1
                    if torch.is_grad_enabled() or taps.ndim != 2:
                        # differentiable form: adjoint on the replicate-extended domain, then fold the extension onto the edges
                        ext = [adj[0] + rx * ux, adj[1] + rx * ux, adj[2] + ry * uy, adj[3] + ry * uy]
                        g = _upfirdn2d_hip(up=down, down=up, padding=ext, flip_filter=(not flip_filter), gain=gain).apply(dy, taps)
                        dx = _fold_edges(g, rx, ry)
                    else:
                        # fast form: ordinary zero-mode adjoint + a border-only kernel that adds the folded extension terms
                        dx = _launch(dy, taps, dnx, dny, ux, uy, adj[0], adj[1], adj[2], adj[3], not flip_filter, gain, 'zero')
                        rc = _lib.lib().agf_upfirdn2d_fold_border(
                            _lib.ptr(dy), _lib.ptr(taps), _lib.ptr(dx), _lib.dtype_code(dy),
                            _lib.sizes4(dy), _lib.strides4(dy), _lib._i32x2(*taps.shape), _lib._i64x2(*taps.stride()),
                            _lib.sizes4(dx), _lib.strides4(dx), dnx, dny, ux, uy, adj[0], adj[2],
                            int(not flip_filter), float(gain), rx, ry, _lib.stream_ptr(dy))
                        _lib.check(rc, 'upfirdn2d_fold_border')
            assert not ctx.needs_input_grad[1]                         # the filter is a constant of the op
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
    """Filter at unchanged size (reference upfirdn2d.py:271-303): the taps' extent is split over the two sides, the larger half first."""
    pl, pr, pt, pb = _lrtb_padding(padding)
    tw, th = _taps_wh(f)
    same = [pl + tw // 2, pr + (tw - 1) // 2, pt + th // 2, pb + (th - 1) // 2]
    return upfirdn2d(x, f, padding=same, flip_filter=flip_filter, gain=gain, impl=impl)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl='hip', edge='zero'):
    """Zero-insertion upsampling by ``up`` followed by the filter, output = input * up; the gain makes up for the inserted zeros
    (reference upfirdn2d.py:307-342)."""
    ux, uy = _xy_factors(up)
    pl, pr, pt, pb = _lrtb_padding(padding)
    tw, th = _taps_wh(f)
    grown = [pl + (tw + ux - 1) // 2, pr + (tw - ux) // 2, pt + (th + uy - 1) // 2, pb + (th - uy) // 2]
    return upfirdn2d(x, f, up=up, padding=grown, flip_filter=flip_filter, gain=gain * ux * uy, impl=impl, edge=edge)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl='hip'):
    """The filter followed by keeping every ``down``-th sample, output = input / down (reference upfirdn2d.py:346-381)."""
    dnx, dny = _xy_factors(down)
    pl, pr, pt, pb = _lrtb_padding(padding)
    tw, th = _taps_wh(f)
    shrunk = [pl + (tw - dnx + 1) // 2, pr + (tw - dnx) // 2, pt + (th - dny + 1) // 2, pb + (th - dny) // 2]
    return upfirdn2d(x, f, down=down, padding=shrunk, flip_filter=flip_filter, gain=gain, impl=impl)
