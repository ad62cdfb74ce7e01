"""``conv2d_gradfix`` surface (reference thirdparty/stylegan3_ops/ops/conv2d_gradfix.py:29-47): ``conv2d`` / ``conv_transpose2d`` with
arbitrary-order gradients.

The reference keeps ``enabled = False`` and therefore always calls plain ``F.conv2d`` / ``F.conv_transpose2d`` (conv2d_gradfix.py:15,29-47).
Here both run on this package's own convolution kernels whenever the shape is one they take (one group, no dilation, a square kernel
of odd size -- 1 or 3 in bf16 on the MFMA kernels, up to 7 in fp32 on the reference-precision kernel), by reduction to the stride-1 "same"
convolution those kernels implement (``implementations.StyleGAN2.conv.conv2d``, differentiable to any order with its own kernels):

    conv2d, padding p, stride 1     same-conv of the input zero-padded by p - k//2 per side (cropped by k//2 - p when that is negative)
    conv2d, stride s                the stride-1 result, every s-th sample (exact: a strided conv IS the decimated stride-1 conv); the
                                    StyleGAN3 discriminator's own stride-2 layers do not come through here -- they run on the strided
                                    MFMA kernel (``conv2d_s2``)
    conv_transpose2d, stride s      zero-insertion by s (``upfirdn2d`` with the unit impulse: one launch), then the stride-1 conv with the
                                    spatially flipped, channel-swapped weights and padding k - 1 - p

Everything else (groups, dilation, other kernel sizes, CPU tensors) goes to ATen as in the reference, pinned to channels-last.  MIOpen's
transposed convolution aborted sporadically on fresh boxes (first-use kernel build) -- another reason not to depend on it for the shapes
the package's callers produce."""
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


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else tuple(int(a) for a in v)


def _own_kernel_takes(x, w, groups, dilation):
    if not (x.is_cuda and x.ndim == 4 and w.ndim == 4 and groups == 1 and _pair(dilation) == (1, 1) and x.dtype == w.dtype):
        return False
    kh, kw = int(w.shape[2]), int(w.shape[3])
    if kh != kw or kh % 2 == 0:
        return False
    return (x.dtype == torch.float32 and kh <= 7) or (x.dtype == torch.bfloat16 and kh in (1, 3))


def _same_conv(x, w, pad_h, pad_w):
    """Stride-1 conv of x with zero padding (pad_h, pad_w) per side through the "same"-padding kernel."""
    from ..implementations.StyleGAN2.conv import conv2d as own_conv2d
    from . import upfirdn2d as _fir
    k = int(w.shape[2])
    eh, ew = pad_h - k // 2, pad_w - k // 2
    if eh > 0 or ew > 0:
        x = _fir.upfirdn2d(x, None, padding=[max(ew, 0), max(ew, 0), max(eh, 0), max(eh, 0)])     # zero border (one launch, differentiable)
    y = own_conv2d(x.contiguous(memory_format=torch.channels_last), w)
    if eh < 0 or ew < 0:
        ch, cw = max(-eh, 0), max(-ew, 0)
        y = y[:, :, ch:y.shape[2] - ch, cw:y.shape[3] - cw]
    return y


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    if _own_kernel_takes(input, weight, groups, dilation) and min(ph, pw) >= 0:
        y = _same_conv(input, weight, ph, pw)
        if sh > 1 or sw > 1:
            y = y[:, :, ::sh, ::sw]
        if bias is not None:
            y = y + bias.to(y.dtype).reshape(1, -1, 1, 1)
        return y
    y = torch.nn.functional.conv2d(input=_pin(input), weight=_pin(weight), bias=bias, stride=stride, padding=padding,
                                   dilation=dilation, groups=groups)
    return _pin(y)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    oh, ow = _pair(output_padding)
    k = int(weight.shape[2]) if weight.ndim == 4 else 0
    if _own_kernel_takes(input, weight, groups, dilation) and 0 <= ph <= k - 1 and 0 <= pw <= k - 1:
        from . import upfirdn2d as _fir
        x = input
        if sh > 1 or sw > 1 or oh or ow:
            # zero insertion: sample (i, j) lands on (i * sh, j * sw); the trailing stride - 1 zeros of the polyphase grid are cut, the
            # output padding appended
            x = _fir.upfirdn2d(x, None, up=(sw, sh), padding=[0, ow - (sw - 1), 0, oh - (sh - 1)])
        wt = weight.flip([2, 3]).transpose(0, 1)                           # [Cout, Cin, k, k] of the equivalent correlation
        y = _same_conv(x, wt, k - 1 - ph, k - 1 - pw)
        if bias is not None:
            y = y + bias.to(y.dtype).reshape(1, -1, 1, 1)
        return y
    y = torch.nn.functional.conv_transpose2d(input=_pin(input), weight=_pin(weight), bias=bias, stride=stride, padding=padding,
                                             output_padding=output_padding, groups=groups, dilation=dilation)
    return _pin(y)
