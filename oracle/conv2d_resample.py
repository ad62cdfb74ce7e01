"""Oracle: conv2d_resample in its definition form.  TEST INFRASTRUCTURE ONLY.

The reference (``thirdparty/stylegan3_ops/ops/conv2d_resample.py:40-135``) evaluates  D_down . Conv_w . U_up  through several fused
fast paths (strided / transposed ATen convolutions, reordered 1x1 cases); its last, generic path (``:130-135``) is the definition
itself: ALL of the padding goes to the first FIR stage (which zero-inserts by ``up``, or only pads when ``up == 1``), the
convolution is unpadded, the decimating FIR follows with zero padding.  Every fast path is that composite evaluated in another
order, so this restatement is compared with the reference's outputs (``tests/golden/conv2d_resample.npz``: 14 parameter sets
that reach each fast path) at fp32 rounding tolerance, not bit-exactly.  ``groups == 1`` only (the networks use nothing else).
"""
import torch
import torch.nn.functional as F

from . import upfirdn2d as U


def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, flip_weight=True, flip_filter=False):
    fw, fh = U.filter_size(f)
    px0, px1, py0, py1 = U.parse_padding(padding)
    if up > 1:      # margins of a centred interpolation filter (conv2d_resample.py:75-79, the split upsample2d uses)
        px0 += (fw + up - 1) // 2; px1 += (fw - up) // 2
        py0 += (fh + up - 1) // 2; py1 += (fh - up) // 2
    if down > 1:    # margins of a centred decimation filter (conv2d_resample.py:80-85)
        px0 += (fw - down + 1) // 2; px1 += (fw - down) // 2
        py0 += (fh - down + 1) // 2; py1 += (fh - down) // 2
    x = U.upfirdn2d(x, f if up > 1 else None, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
    if not flip_weight:                              # conv2d is a correlation; flip_weight=False asks for the true convolution
        w = w.flip([2, 3])
    x = F.conv2d(x, w)
    if down > 1:
        x = U.upfirdn2d(x, f, down=down, flip_filter=flip_filter)
    return x
