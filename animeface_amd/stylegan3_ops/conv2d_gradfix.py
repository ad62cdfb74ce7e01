"""``conv2d_gradfix`` surface (reference thirdparty/stylegan3_ops/ops/conv2d_gradfix.py:29-47).

The reference keeps ``enabled = False`` and therefore always calls plain ``F.conv2d`` /
``F.conv_transpose2d`` (conv2d_gradfix.py:15,29-47); the same holds here.  The MFMA contraction
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


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    return torch.nn.functional.conv2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding,
                                      dilation=dilation, groups=groups)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    return torch.nn.functional.conv_transpose2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding,
                                                output_padding=output_padding, groups=groups, dilation=dilation)
