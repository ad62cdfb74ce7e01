"""MFMA convolution of the StyleGAN2 path (3x3 / 1x1, stride 1, "same" padding) as autograd ops.

Replaces the ATen/cuDNN calls under the reference's ``ModulatedConv2d.forward``
(implementations/StyleGAN2/model.py:106-132) and ``ELR(nn.Conv2d)`` (model.py:29-37) with
``agf_conv2d_fwd`` / ``agf_conv2d_wgrad``.  Activations are bf16 channels-last; weights are passed in
the logical [Cout, Cin, k, k] shape and laid out OHWI (== channels_last memory format) for the kernel.

Every backward is expressed with these same differentiable ops, so gradients of any order (R1 differentiates
the discriminator twice, the path-length penalty the generator) compose from three kernels:
    fwd(x, w)            -> dx = fwd(dy, flipT(w)),  dw = wgrad(x, dy)
    wgrad(x, dy)         -> dx = fwd(dy, flipT(ddw)),  d(dy) = fwd(x, ddw)
"""
import torch

from ... import _lib

ACT_LINEAR, ACT_LRELU = 1, 3


class KernelTimer:
    """Optional per-launch HIP-event timing of the MFMA kernels on the stream they are launched on
    (bench.py's roofline numbers).  Disabled unless ``KernelTimer.active`` is set to an instance."""
    active = None

    def __init__(self):
        self.records = {}          # kernel name -> list of (start_event, end_event, flops)

    def start(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def stop(self, name, ev0, flops):
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record()
        self.records.setdefault(name, []).append((ev0, ev1, flops))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            ms = sum(a.elapsed_time(b) for a, b, _ in recs)
            fl = sum(f for _, _, f in recs)
            out[name] = dict(launches=len(recs), total_ms=ms, avg_ms=ms / max(len(recs), 1), flops=fl,
                             tflops=fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0)
        return out


def _f32(t):
    return None if t is None else t.contiguous().float()


def _ohwi(w):
    """[Cout,Cin,k,k] -> dense tensor whose memory order is [Cout][kh][kw][Cin]."""
    if w.shape[2] == 1 and w.shape[3] == 1:
        return w.contiguous()
    return w.contiguous(memory_format=torch.channels_last)


def conv2d_fwd_raw(x, w, in_scale=None, out_scale=None, bias=None, noise=None, residual=None,
                   act=ACT_LINEAR, alpha=0.2, gain=1.0):
    """One ``agf_conv2d_fwd`` launch.  x: [N,Cin,H,W] bf16 channels_last; w: [Cout,Cin,k,k] (any float dtype).
    in_scale [N,Cin], out_scale [N,Cout], bias [Cout], noise [N,1,H,W] are fp32; residual like y.  Returns y bf16 channels_last."""
    _lib.require_gpu(x, 'conv2d')
    N, Cin, H, W = x.shape
    Cout, Cin_w, k, k2 = w.shape
    if Cin_w != Cin or k != k2:
        raise RuntimeError(f'conv2d: weight {tuple(w.shape)} does not match input {tuple(x.shape)}')
    if x.dtype not in (torch.bfloat16, torch.float32):
        raise RuntimeError('conv2d: activations must be bfloat16 (MFMA path) or float32 (reference-precision path)')
    x = x.contiguous(memory_format=torch.channels_last)
    wq = _ohwi(w.to(x.dtype))
    y = torch.empty((N, Cout, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    in_scale, out_scale, bias, noise = _f32(in_scale), _f32(out_scale), _f32(bias), _f32(noise)
    if residual is not None:
        residual = residual.to(x.dtype).contiguous(memory_format=torch.channels_last)
    timer = KernelTimer.active
    ev0 = timer.start() if timer is not None else None
    rc = _lib.lib().agf_conv2d_fwd(_lib.ptr(x), _lib.ptr(wq), _lib.ptr(y), _lib.ptr(in_scale), _lib.ptr(out_scale),
                                   _lib.ptr(bias), _lib.ptr(noise), _lib.ptr(residual), _lib.dtype_code(x),
                                   N, H, W, Cin, Cout, k, act, float(alpha), float(gain), _lib.stream_ptr(x))
    if timer is not None:
        timer.stop('conv2d_fwd_kernel', ev0, 2.0 * N * H * W * Cin * Cout * k * k)
    _lib.check(rc, 'conv2d_fwd')
    return y


def conv2d_wgrad_raw(x, dy, ksize, in_scale=None, out_scale=None):
    """One ``agf_conv2d_wgrad`` launch.  x: [N,Cin,H,W], dy: [N,Cout,H,W], both bf16 channels_last.
    Returns dw fp32 in the logical [Cout,Cin,k,k] shape (memory OHWI)."""
    _lib.require_gpu(x, 'conv2d_wgrad')
    N, Cin, H, W = x.shape
    Cout = dy.shape[1]
    if dy.shape[0] != N or dy.shape[2] != H or dy.shape[3] != W:
        raise RuntimeError(f'conv2d_wgrad: dy {tuple(dy.shape)} does not match x {tuple(x.shape)}')
    if x.dtype not in (torch.bfloat16, torch.float32) or dy.dtype != x.dtype:
        raise RuntimeError('conv2d_wgrad: x and dy must both be bfloat16 or both float32')
    x = x.contiguous(memory_format=torch.channels_last)
    dy = dy.contiguous(memory_format=torch.channels_last)
    dw = torch.zeros((Cout, Cin, ksize, ksize), dtype=torch.float32, device=x.device).contiguous(memory_format=torch.channels_last)
    if ksize == 1:   # channels_last strides of a [Cout,Cin,1,1] tensor are ambiguous; memory is [Cout][Cin] either way
        dw = torch.zeros((Cout, Cin, 1, 1), dtype=torch.float32, device=x.device)
    in_scale, out_scale = _f32(in_scale), _f32(out_scale)
    timer = KernelTimer.active
    ev0 = timer.start() if timer is not None else None
    rc = _lib.lib().agf_conv2d_wgrad(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(in_scale), _lib.ptr(out_scale),
                                     _lib.dtype_code(x), N, H, W, Cin, Cout, ksize, _lib.stream_ptr(x))
    if timer is not None:
        timer.stop('conv2d_wgrad_kernel', ev0, 2.0 * N * H * W * Cin * Cout * ksize * ksize)
    _lib.check(rc, 'conv2d_wgrad')
    return dw


def flip_transpose(w):
    """Weights of the adjoint (dgrad) convolution: spatial flip + swap of the channel axes."""
    return w.flip([2, 3]).transpose(0, 1)


class _ConvFwd(torch.autograd.Function):
    """y[n] = s_out[n,:,None,None] * conv(x[n] * s_in[n,:,None,None], w)   (scales optional)."""

    @staticmethod
    def forward(ctx, x, w, s_in, s_out):
        y = conv2d_fwd_raw(x, w, in_scale=s_in, out_scale=s_out)
        ctx.save_for_backward(x, w, s_in, s_out, y if (s_out is not None) else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, s_in, s_out, y = ctx.saved_tensors
        dx = dw = ds_in = ds_out = None
        dy = dy.to(x.dtype)
        if ctx.needs_input_grad[0] or (s_in is not None and ctx.needs_input_grad[2]):
            if s_in is None:
                dx = _ConvFwd.apply(dy, flip_transpose(w), s_out, None)
            else:
                t = _ConvFwd.apply(dy, flip_transpose(w), s_out, None)           # gradient w.r.t. (x * s_in)
                if ctx.needs_input_grad[0]:
                    dx = t * s_in[:, :, None, None].to(t.dtype)
                if ctx.needs_input_grad[2]:
                    ds_in = (x.float() * t.float()).sum((2, 3))
        if ctx.needs_input_grad[1]:
            dw = _ConvWgrad.apply(x, dy, s_in, s_out, w.shape[2]).to(w.dtype)
        if s_out is not None and ctx.needs_input_grad[3]:
            ds_out = (dy.float() * y.float()).sum((2, 3)) / s_out
        return dx, dw, ds_in, ds_out


class _ConvWgrad(torch.autograd.Function):
    """dw[co,ci,t] = sum_{n,p} (dy*s_out)[n,co,p] * (x*s_in)[n,ci,p+t]  -> fp32 [Cout,Cin,k,k]."""

    @staticmethod
    def forward(ctx, x, dy, s_in, s_out, ksize):
        ctx.save_for_backward(x, dy, s_in, s_out)
        ctx.ksize = ksize
        return conv2d_wgrad_raw(x, dy, ksize, in_scale=s_in, out_scale=s_out)

    @staticmethod
    def backward(ctx, ddw):
        x, dy, s_in, s_out = ctx.saved_tensors
        gx = gdy = gsi = gso = None
        # the bilinear form  <ddw, wgrad(x, dy)>  ==  <dy*s_out, conv(x*s_in, ddw)>
        if ctx.needs_input_grad[0] or (s_in is not None and ctx.needs_input_grad[2]):
            t = _ConvFwd.apply(dy, flip_transpose(ddw), s_out, None)
            if s_in is None:
                gx = t
            else:
                if ctx.needs_input_grad[0]:
                    gx = t * s_in[:, :, None, None].to(t.dtype)
                if ctx.needs_input_grad[2]:
                    gsi = (x.float() * t.float()).sum((2, 3))
        if ctx.needs_input_grad[1] or (s_out is not None and ctx.needs_input_grad[3]):
            u = _ConvFwd.apply(x, ddw, s_in, None)
            if s_out is None:
                gdy = u
            else:
                if ctx.needs_input_grad[1]:
                    gdy = u * s_out[:, :, None, None].to(u.dtype)
                if ctx.needs_input_grad[3]:
                    gso = (dy.float() * u.float()).sum((2, 3))
        return gx, gdy, gsi, gso, None


def _pad_channels(t, mult, dim):
    c = t.shape[dim]
    extra = (-c) % mult
    if extra == 0:
        return t
    pad = [0, 0] * (t.dim() - dim - 1) + [0, extra]
    return torch.nn.functional.pad(t, pad)


def conv2d(x, w, s_in=None, s_out=None):
    """Differentiable (to any order) 3x3 / 1x1 "same" convolution on the MFMA kernels.

    x: [N,Cin,H,W] bf16 (channels_last preferred); w: [Cout,Cin,k,k] fp32 or bf16 master weights;
    s_in [N,Cin] / s_out [N,Cout]: optional fp32 per-sample channel scales (style modulation / demodulation).
    Channel counts that are not multiples of 8 are zero-padded here (the 513-channel minibatch-stddev conv)."""
    Cout, Cin = w.shape[0], w.shape[1]
    if x.dtype == torch.bfloat16 and (Cin % 8 or Cout % 8):
        xp = _pad_channels(x, 8, 1)
        wp = _pad_channels(_pad_channels(w, 8, 1), 8, 0)
        si = _pad_channels(s_in, 8, 1) if s_in is not None else None
        so = _pad_channels(s_out, 8, 1) if s_out is not None else None
        return _ConvFwd.apply(xp.contiguous(memory_format=torch.channels_last), wp, si, so)[:, :Cout]
    return _ConvFwd.apply(x, w, s_in, s_out)
