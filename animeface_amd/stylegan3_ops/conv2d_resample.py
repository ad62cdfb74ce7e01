"""``conv2d_resample``: the FIR-resampling sandwich  D_down . Conv_w . U_up  on the MI355X kernels.

Public surface and results of the reference's ``thirdparty/stylegan3_ops/ops/conv2d_resample.py:40-135``
(``conv2d_resample(x, w, f, up, down, padding, groups, flip_weight, flip_filter)``), written from the operator's
definition rather than from that file:

    U_up   : zero-insert by ``up``, pad, correlate with ``f`` scaled by up**2        (``upfirdn2d``, one launch)
    Conv_w : stride-1 correlation with ``w`` (true convolution when ``flip_weight`` is False)
    D_down : pad, correlate with ``f``, keep every ``down``-th sample              (``upfirdn2d``, one launch)

All three stages are linear and shift-invariant, so they may be merged or reordered as long as the composite impulse
response and the sampling lattice are unchanged.  ``_schedule`` picks the cheapest legal order:

    * 1x1 weights commute with both FIR stages: run the channel mix on the SMALLER of the two maps;
    * a decimating FIR followed by a k x k conv = FIR at full rate, then the conv evaluated only on the kept lattice (stride);
    * zero-insertion followed by a k x k conv = transposed conv with stride ``up``; the FIR then runs on its output.

The stride-1 channel mix runs on the MFMA conv (``implementations.StyleGAN2.conv.conv2d``) whenever its shape allows it
(one group, 1x1 / 3x3, "same" padding, GPU tensor); strided and transposed convs go to ATen through ``conv2d_gradfix``
exactly as the reference's do.
"""
import torch

from . import conv2d_gradfix
from . import upfirdn2d as _fir


class _Margins:
    """Padding of the composite op as [left, right, top, bottom]; FIR stages widen it so that the output size is x * up / down."""

    def __init__(self, padding):
        self.l, self.r, self.t, self.b = _fir._lrtb_padding(padding)

    def widen(self, fw, fh, factor, upsampling):
        # a centred FIR of fw taps on a lattice resampled by `factor`: the same split upsample2d / downsample2d use
        if factor <= 1:
            return
        lead = (fw + factor - 1) // 2 if upsampling else (fw - factor + 1) // 2
        lead_v = (fh + factor - 1) // 2 if upsampling else (fh - factor + 1) // 2
        self.l += lead
        self.r += (fw - factor) // 2
        self.t += lead_v
        self.b += (fh - factor) // 2

    def shift(self, dl, dr, dt, db):
        self.l += dl; self.r += dr; self.t += dt; self.b += db

    @property
    def lrtb(self):
        return [self.l, self.r, self.t, self.b]

    def symmetric(self):
        return self.l == self.r and self.t == self.b and self.l >= 0 and self.t >= 0


def _mix(x, w, flip_weight, stride=1, pad_hw=(0, 0), groups=1, transposed=False):
    """Channel-mixing stage.  ``flip_weight=True`` means ``w`` is used as a correlation kernel (the framework default)."""
    kh, kw = int(w.shape[2]), int(w.shape[3])
    if (kh > 1 or kw > 1) and not flip_weight:
        w = w.flip([2, 3])
    if transposed:
        return conv2d_gradfix.conv_transpose2d(x, w, stride=stride, padding=list(pad_hw), groups=groups)
    same = kh == kw and kh in (1, 3) and tuple(pad_hw) == (kh // 2, kh // 2)
    if stride == 1 and groups == 1 and same and x.is_cuda and x.dtype in (torch.bfloat16, torch.float32):
        from ..implementations.StyleGAN2.conv import conv2d as mfma_conv2d
        return mfma_conv2d(x, w)
    return conv2d_gradfix.conv2d(x, w, stride=stride, padding=list(pad_hw), groups=groups)


def _schedule(kh, kw, up, down, m):
    pointwise = kh == 1 and kw == 1
    if pointwise and up == 1 and down > 1:
        return 'decimate_then_mix'
    if pointwise and down == 1 and up > 1:
        return 'mix_then_interpolate'
    if up == 1 and down > 1:
        return 'filter_then_strided'
    if up > 1:
        return 'transposed_then_filter'
    if m.symmetric():
        return 'mix_only'
    return 'pad_then_mix'


def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False):
    """2-D convolution with optional up / downsampling by the FIR filter ``f`` (see the module docstring).

    x [N, Cin, H, W]; w [Cout, Cin // groups, kh, kw] in x's dtype; f from ``upfirdn2d.setup_filter`` (fp32, 1-D or 2-D) or None.
    ``padding`` is relative to the output lattice: 0 keeps the size at H * up / down."""
    if not (isinstance(x, torch.Tensor) and x.ndim == 4):
        raise AssertionError('x must be a rank-4 tensor')
    if not (isinstance(w, torch.Tensor) and w.ndim == 4 and w.dtype == x.dtype):
        raise AssertionError('w must be a rank-4 tensor of the dtype of x')
    if not (f is None or (isinstance(f, torch.Tensor) and f.ndim in (1, 2) and f.dtype == torch.float32)):
        raise AssertionError('f must be None or a float32 tensor of rank 1 or 2')
    for name, v in (('up', up), ('down', down), ('groups', groups)):
        if not (isinstance(v, int) and v >= 1):
            raise AssertionError(f'{name} must be a positive int')
    cout, cin_g, kh, kw = (int(s) for s in w.shape)
    fw, fh = _fir._taps_wh(f)
    m = _Margins(padding)
    m.widen(fw, fh, up, upsampling=True)
    m.widen(fw, fh, down, upsampling=False)
    plan = _schedule(kh, kw, up, down, m)

    if plan == 'decimate_then_mix':
        x = _fir.upfirdn2d(x, f, down=down, padding=m.lrtb, flip_filter=flip_filter)
        return _mix(x, w, flip_weight, groups=groups)

    if plan == 'mix_then_interpolate':
        x = _mix(x, w, flip_weight, groups=groups)
        return _fir.upfirdn2d(x, f, up=up, padding=m.lrtb, gain=up ** 2, flip_filter=flip_filter)

    if plan == 'filter_then_strided':
        x = _fir.upfirdn2d(x, f, padding=m.lrtb, flip_filter=flip_filter)
        return _mix(x, w, flip_weight, stride=down, groups=groups)

    if plan == 'transposed_then_filter':
        # the transposed conv wants [Cin, Cout // groups, kh, kw]
        if groups == 1:
            wt = w.transpose(0, 1)
        else:
            wt = w.reshape(groups, cout // groups, cin_g, kh, kw).transpose(1, 2).reshape(groups * cin_g, cout // groups, kh, kw)
        # a transposed conv of stride `up` emits (k - 1) extra leading and (k - up) extra trailing samples per axis: take them out of
        # the FIR's margins, and let the transposed conv itself crop whatever part of a negative margin both sides share
        m.shift(-(kw - 1), -(kw - up), -(kh - 1), -(kh - up))
        crop_w = max(min(-m.l, -m.r), 0)
        crop_h = max(min(-m.t, -m.b), 0)
        x = _mix(x, wt, not flip_weight, stride=up, pad_hw=(crop_h, crop_w), groups=groups, transposed=True)
        m.shift(crop_w, crop_w, crop_h, crop_h)
        x = _fir.upfirdn2d(x, f, padding=m.lrtb, gain=up ** 2, flip_filter=flip_filter)
        if down > 1:
            x = _fir.upfirdn2d(x, f, down=down, flip_filter=flip_filter)
        return x

    if plan == 'mix_only':
        return _mix(x, w, flip_weight, pad_hw=(m.t, m.l), groups=groups)

    # asymmetric or negative margins without resampling: one padding / cropping FIR pass (f = None is the unit impulse), then the mix
    x = _fir.upfirdn2d(x, None, padding=m.lrtb)
    return _mix(x, w, flip_weight, groups=groups)
