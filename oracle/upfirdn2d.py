"""Oracle: upfirdn2d (pad / zero-insert upsample / FIR / decimate).  TEST INFRASTRUCTURE ONLY.

Restates the semantics of the reference's slow path
``thirdparty/stylegan3_ops/ops/upfirdn2d.py:161-205`` (``_upfirdn2d_ref``) and of
the native kernels ``upfirdn2d.cu:23-86,91-194`` in the closed form of
SURVEY.md Appendix A:

    y[n,c,oy,ox] = gain * sum_{ky,kx} U[oy*downy + ky, ox*downx + kx] * F[ky,kx]

where U is the zero-inserted (``up``) and zero-padded / cropped (``padding``)
image and F is the filter, flipped unless ``flip_filter`` (true convolution is
the default, ``upfirdn2d.py:192-193``).  Unlike the reference, which hands the
padded image to a depthwise ``conv2d``, the taps are accumulated one at a time
in row-major (ky, kx) order -- the same order the native small kernel uses
(``upfirdn2d.cu:184-187``) -- which keeps the code independent of the conv
backend and differentiable to any order through plain autograd.

Extension used by the StyleGAN2 mapping ("Bilinear-upsample
borders"): ``edge='clamp'`` replicates the border instead of zero-filling,
which makes ``upsample2d([1,3,3,1])`` equal ``nn.Upsample(bilinear)``
(``implementations/StyleGAN2/model.py:56-58``).  ``edge='zero'`` is the reference op.
"""
import numpy as np
import torch
import torch.nn.functional as F


def parse_scaling(scaling):
    # thirdparty/stylegan3_ops/ops/upfirdn2d.py:29-36
    if isinstance(scaling, int):
        scaling = [scaling, scaling]
    sx, sy = scaling
    assert sx >= 1 and sy >= 1
    return int(sx), int(sy)


def parse_padding(padding):
    # thirdparty/stylegan3_ops/ops/upfirdn2d.py:38-47
    if isinstance(padding, int):
        padding = [padding, padding]
    padding = [int(p) for p in padding]
    if len(padding) == 2:
        px, py = padding
        padding = [px, px, py, py]
    px0, px1, py0, py1 = padding
    return px0, px1, py0, py1


def filter_size(f):
    # thirdparty/stylegan3_ops/ops/upfirdn2d.py:49-60  (returns width, height)
    if f is None:
        return 1, 1
    assert f.ndim in (1, 2)
    return int(f.shape[-1]), int(f.shape[0])


def setup_filter(f, device=torch.device('cpu'), normalize=True, flip_filter=False, gain=1, separable=None):
    """thirdparty/stylegan3_ops/ops/upfirdn2d.py:64-108."""
    if f is None:
        f = 1
    f = torch.as_tensor(f, dtype=torch.float32).clone()
    assert f.ndim in (0, 1, 2) and f.numel() > 0
    if f.ndim == 0:
        f = f[None]
    if separable is None:
        separable = (f.ndim == 1 and f.numel() >= 8)      # :95-96
    if f.ndim == 1 and not separable:
        f = torch.outer(f, f)                              # :97-98
    assert f.ndim == (1 if separable else 2)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    f = f * (gain ** (f.ndim / 2))                         # :106
    return f.to(device)


def out_size(in_size, f_size, up, down, pad0, pad1):
    """upfirdn2d.cpp:29-30."""
    return (in_size * up + pad0 + pad1 - f_size + down) // down


def _fir_pass(x, f2d, upx, upy, downx, downy, px0, px1, py0, py1, flip, gain, edge):
    """One pass with a rank-2 fp32 filter, equal to one native plugin call
    (``upfirdn2d.cpp:10-91``).  Accumulates in fp32 for fp16/bf16/fp32 input and
    fp64 for fp64 (``upfirdn2d.cu:9-12``) and rounds once to the input dtype."""
    N, C, H, W = x.shape
    fh, fw = f2d.shape
    ow = out_size(W, fw, upx, downx, px0, px1)
    oh = out_size(H, fh, upy, downy, py0, py1)
    assert ow >= 1 and oh >= 1, 'output must be at least 1x1'
    acc_t = torch.float64 if x.dtype == torch.float64 else torch.float32
    xa = x.to(acc_t)
    if edge == 'clamp':
        # replicate the border far enough to cover the footprint, then crop the
        # same amount (in upsampled pixels) through the padding.
        rx = (max(px0, px1, 0) + upx - 1) // upx + 1
        ry = (max(py0, py1, 0) + upy - 1) // upy + 1
        xa = F.pad(xa, [rx, rx, ry, ry], mode='replicate')
        H, W = H + 2 * ry, W + 2 * rx
        px0, px1, py0, py1 = px0 - rx * upx, px1 - rx * upx, py0 - ry * upy, py1 - ry * upy
    # zero insertion (upfirdn2d.py:181-183)
    u = xa.reshape(N, C, H, 1, W, 1)
    u = F.pad(u, [0, upx - 1, 0, 0, 0, upy - 1])
    u = u.reshape(N, C, H * upy, W * upx)
    # pad or crop (upfirdn2d.py:186-187)
    u = F.pad(u, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    u = u[:, :, max(-py0, 0): u.shape[2] - max(-py1, 0), max(-px0, 0): u.shape[3] - max(-px1, 0)]
    assert u.shape[2] >= fh and u.shape[3] >= fw
    ff = f2d.to(acc_t)
    if not flip:
        ff = ff.flip([0, 1])
    y = None
    for ky in range(fh):
        for kx in range(fw):
            tap = u[:, :, ky: ky + (oh - 1) * downy + 1: downy, kx: kx + (ow - 1) * downx + 1: downx] * ff[ky, kx]
            y = tap if y is None else y + tap
    y = y * gain
    return y.to(x.dtype)


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, edge='zero'):
    """Same call surface as ``upfirdn2d.upfirdn2d`` (``upfirdn2d.py:112-156``).

    Pass structure follows the *native* path (``upfirdn2d.py:229-239``): rank-2
    filter = one pass; rank-1 filter = x-pass with gain 1 then y-pass with
    ``gain``; 1-tap rank-1 filter is squared into a 1x1.
    """
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    upx, upy = parse_scaling(up)
    downx, downy = parse_scaling(down)
    px0, px1, py0, py1 = parse_padding(padding)
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32)
    f = f.to(torch.float32)
    if f.ndim == 1 and f.shape[0] == 1:
        f = f.square().unsqueeze(0)
    if f.ndim == 2:
        return _fir_pass(x, f, upx, upy, downx, downy, px0, px1, py0, py1, flip_filter, gain, edge)
    assert edge == 'zero'
    y = _fir_pass(x, f.unsqueeze(0), upx, 1, downx, 1, px0, px1, 0, 0, flip_filter, 1.0, edge)
    y = _fir_pass(y, f.unsqueeze(1), 1, upy, 1, downy, 0, 0, py0, py1, flip_filter, gain, edge)
    return y


def filter2d(x, f, padding=0, flip_filter=False, gain=1):
    """upfirdn2d.py:271-303."""
    px0, px1, py0, py1 = parse_padding(padding)
    fw, fh = filter_size(f)
    p = [px0 + fw // 2, px1 + (fw - 1) // 2, py0 + fh // 2, py1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, edge='zero'):
    """upfirdn2d.py:307-342."""
    upx, upy = parse_scaling(up)
    px0, px1, py0, py1 = parse_padding(padding)
    fw, fh = filter_size(f)
    p = [px0 + (fw + upx - 1) // 2, px1 + (fw - upx) // 2, py0 + (fh + upy - 1) // 2, py1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy, edge=edge)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1):
    """upfirdn2d.py:346-381."""
    downx, downy = parse_scaling(down)
    px0, px1, py0, py1 = parse_padding(padding)
    fw, fh = filter_size(f)
    p = [px0 + (fw - downx + 1) // 2, px1 + (fw - downx) // 2, py0 + (fh - downy + 1) // 2, py1 + (fh - downy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain)


def upfirdn2d_grad_padding(x_shape, y_shape, f, up, down, padding):
    """Padding of the adjoint op (``upfirdn2d.py:250-255``)."""
    upx, upy = parse_scaling(up)
    downx, downy = parse_scaling(down)
    px0, _px1, py0, _py1 = parse_padding(padding)
    _, _, ih, iw = x_shape
    _, _, oh, ow = y_shape
    fw, fh = filter_size(f)
    return [fw - px0 - 1, iw * upx - ow * downx + px0 - upx + 1,
            fh - py0 - 1, ih * upy - oh * downy + py0 - upy + 1]


def upfirdn2d_numpy_gather(x, f2d, upx, upy, downx, downy, px0, px1, py0, py1, flip, gain):
    """Independent second statement used to cross-check `_fir_pass` on small
    integer cases: the gather form of the native kernels
    (``upfirdn2d.cu:14-18,46-60``: mid = out*down + up-1 - pad0; in0 = floor(mid/up);
    tap0 = (in0+1)*up - mid - 1, step taps by up).  Pure python loops."""
    x = np.asarray(x, dtype=np.float64)
    f2d = np.asarray(f2d, dtype=np.float64)
    N, C, H, W = x.shape
    fh, fw = f2d.shape
    ow = out_size(W, fw, upx, downx, px0, px1)
    oh = out_size(H, fh, upy, downy, py0, py1)
    y = np.zeros((N, C, oh, ow), dtype=np.float64)
    for oy in range(oh):
        midy = oy * downy + upy - 1 - py0
        iny0 = midy // upy                       # python // floors toward -inf like floor_div
        for ox in range(ow):
            midx = ox * downx + upx - 1 - px0
            inx0 = midx // upx
            acc = np.zeros((N, C))
            ty = (iny0 + 1) * upy - midy - 1     # index into the FLIPPED filter
            iy = iny0
            while ty < fh:
                if 0 <= iy < H:
                    tx = (inx0 + 1) * upx - midx - 1
                    ix = inx0
                    while tx < fw:
                        if 0 <= ix < W:
                            # staged filter sf[ty][tx] = f[fh-1-ty][fw-1-tx] unless p.flip (upfirdn2d.cu:113-122)
                            tap = f2d[fh - 1 - ty, fw - 1 - tx] if not flip else f2d[ty, tx]
                            acc += x[:, :, iy, ix] * tap
                        tx += upx
                        ix += 1
                ty += upy
                iy += 1
            y[:, :, oy, ox] = acc * gain
    return y
