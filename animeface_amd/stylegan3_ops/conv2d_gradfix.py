"""``conv2d_gradfix`` surface (reference thirdparty/stylegan3_ops/ops/conv2d_gradfix.py:29-47).

The reference keeps ``enabled = False`` and therefore always calls plain ``F.conv2d`` /
``F.conv_transpose2d`` (conv2d_gradfix.py:15,29-47); the same holds here, with tensors and gradients pinned to channels-last.  The MFMA contraction
of the StyleGAN2 path lives in ``animeface_amd.implementations.StyleGAN2.conv``."""
import contextlib

import torch

enabled = False
weight_gradients_disabled = False


@contextlib.contextmanager
def no_weight_gradients(disable=True):
    global weight_gradients_disabled
    old = weight_gradients_disabled
    if disable:
        weight_gradients_disabled = True
    yield
    weight_gradients_disabled = old


class _ChannelsLast(torch.autograd.Function):
    """Identity that pins a tensor AND its gradient (to any order) to dense channels-last.  MIOpen only picks its
    implicit-GEMM solvers when activations, weights and incoming gradients are all packed in one layout; a single
    NCHW gradient arriving at a channels-last convolution sends it to the "naive nonpacked" kernels (100x slower)."""

    @staticmethod
    def forward(ctx, t):
        return t.contiguous(memory_format=torch.channels_last)

    @staticmethod
    def backward(ctx, g):
        return _ChannelsLast.apply(g)


def _pin(t):
    return _ChannelsLast.apply(t) if (t.is_cuda and t.ndim == 4) else t


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    y = torch.nn.functional.conv2d(input=_pin(input), weight=_pin(weight), bias=bias, stride=stride, padding=padding,
                                   dilation=dilation, groups=groups)
    return _pin(y)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    y = torch.nn.functional.conv_transpose2d(input=_pin(input), weight=_pin(weight), bias=bias, stride=stride, padding=padding,
                                             output_padding=output_padding, groups=groups, dilation=dilation)
    return _pin(y)
