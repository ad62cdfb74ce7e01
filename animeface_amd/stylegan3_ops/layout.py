"""Planar (NCHW) <-> channels-last layout changes with zero padding / cropping and channel padding, on the HIP kernels
``agf_planar_to_cl_pad`` / ``agf_cl_to_planar_crop`` (include/agf_ops.h).  The two ops are adjoint to each other, so each
one's backward is the other: differentiable to any order."""
import torch

from .. import _lib


def _vec(dtype):
    return 4 if dtype == torch.float32 else 8


def padded_channels(c, dtype):
    v = _vec(dtype)
    return (c + v - 1) // v * v


def _to_cl_raw(x, pad, cp, scale=None):
    """``scale`` [N, cp] fp32: the output is multiplied by scale[n, c] on the way (``agf_planar_to_cl_pad_scaled``)."""
    N, C, H, W = x.shape
    x = x.contiguous()
    y = torch.empty((N, cp, H + 2 * pad, W + 2 * pad), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    if scale is not None:
        scale = scale.float().contiguous()
        assert tuple(scale.shape) == (N, cp)
        _lib.check(_lib.lib().agf_planar_to_cl_pad_scaled(_lib.ptr(x), _lib.ptr(y), _lib.ptr(scale), _lib.dtype_code(x), N, C, H, W, pad, cp,
                                                          _lib.stream_ptr(x)), 'planar_to_cl_pad_scaled')
        return y
    _lib.check(_lib.lib().agf_planar_to_cl_pad(_lib.ptr(x), _lib.ptr(y), _lib.dtype_code(x), N, C, H, W, pad, cp,
                                               _lib.stream_ptr(x)), 'planar_to_cl_pad')
    return y


def _to_planar_raw(x, pad, c, scale=None):
    """``scale`` [N, cp] fp32: the output is multiplied by scale[n, c] on the way (``agf_cl_to_planar_crop_scaled``)."""
    N, cp, Hp, Wp = x.shape
    x = x.contiguous(memory_format=torch.channels_last)
    H, W = Hp - 2 * pad, Wp - 2 * pad
    y = torch.empty((N, c, H, W), dtype=x.dtype, device=x.device)
    if scale is not None:
        scale = scale.float().contiguous()
        assert tuple(scale.shape) == (N, cp)
        _lib.check(_lib.lib().agf_cl_to_planar_crop_scaled(_lib.ptr(x), _lib.ptr(y), _lib.ptr(scale), _lib.dtype_code(x), N, c, H, W, pad, cp,
                                                           _lib.stream_ptr(x)), 'cl_to_planar_crop_scaled')
        return y
    _lib.check(_lib.lib().agf_cl_to_planar_crop(_lib.ptr(x), _lib.ptr(y), _lib.dtype_code(x), N, c, H, W, pad, cp,
                                                _lib.stream_ptr(x)), 'cl_to_planar_crop')
    return y


class PlanarToChannelsLast(torch.autograd.Function):
    """x [N,C,H,W] (any strides) -> dense channels-last [N,Cp,H+2p,W+2p]: zero border, zero channels C..Cp-1."""

    @staticmethod
    def forward(ctx, x, pad, cp):
        _lib.require_gpu(x, 'planar_to_cl_pad')
        ctx.pad, ctx.c = pad, x.shape[1]
        return _to_cl_raw(x, pad, cp)

    @staticmethod
    def backward(ctx, g):
        return ChannelsLastToPlanar.apply(g, ctx.pad, ctx.c), None, None


class ChannelsLastToPlanar(torch.autograd.Function):
    """x [N,Cp,H+2p,W+2p] (channels-last) -> dense planar [N,C,H,W]: crop the border and the channels >= C."""

    @staticmethod
    def forward(ctx, x, pad, c):
        _lib.require_gpu(x, 'cl_to_planar_crop')
        ctx.pad, ctx.cp = pad, x.shape[1]
        return _to_planar_raw(x, pad, c)

    @staticmethod
    def backward(ctx, g):
        return PlanarToChannelsLast.apply(g, ctx.pad, ctx.cp), None, None


def planar_to_channels_last(x, pad=0, channels=None):
    cp = padded_channels(x.shape[1], x.dtype) if channels is None else channels
    return PlanarToChannelsLast.apply(x, pad, cp)


def channels_last_to_planar(x, pad=0, channels=None):
    return ChannelsLastToPlanar.apply(x, pad, x.shape[1] if channels is None else channels)
