"""StyleGAN3 generator / discriminator on the MI355X operators.

Same classes, constructor arguments, parameter / buffer names (hence ``state_dict`` keys) and forward semantics as the
reference's ``implementations/StyleGAN3/model.py``; the arithmetic runs on this package's kernels:

  reference                                                   here
  ----------------------------------------------------------  -------------------------------------------------------
  ModulatedConv: grouped conv2d on a [B,Cout,Cin,k,k] weight   MFMA conv with shared weights, per-sample input scale
    (model.py:46-74), padding k-1                               s * ema^-1/2 and output scale d = rsqrt(sum (W*scale*s)^2 + 1e-8);
                                                                "full" padding = zero-pad the input by (k-1) - k//2, then "same" conv
  filtered_lrelu (model.py:186-189)                            fused HIP kernel (agf_filtered_lrelu)
  ConvAct: conv2d_resample + bias_act (model.py:410-417)       stride-1 convs on the MFMA conv; 1x1+down = HIP upfirdn2d(down) then
                                                                MFMA conv; 3x3+down = MFMA conv at stride 1 then HIP upfirdn2d(down)
                                                                (FIR and conv commute, see ConvAct.forward); HIP bias_act
Filter design (``design_filter``, ``get_layer_params``) is scipy / numpy arithmetic exactly as in model.py:76-115.
"""
import math

import numpy as np
import scipy.signal
import scipy.special
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...stylegan3_ops import bias_act, filtered_lrelu, conv2d_resample, upfirdn2d, layout
from ..StyleGAN2.conv import conv2d, conv2d_act


def _native(x):
    return x.is_cuda


class _ZeroPadCL(torch.autograd.Function):
    """Zero-pad H and W of a (channels-last) tensor into a dense channels-last tensor with one copy; differentiable to any
    order (its adjoint is the crop below)."""

    @staticmethod
    def forward(ctx, x, pad):
        ctx.pad = pad
        N, C, H, W = x.shape
        y = torch.empty((N, C, H + 2 * pad, W + 2 * pad), dtype=x.dtype, device=x.device, memory_format=torch.channels_last).zero_()
        y[:, :, pad:-pad, pad:-pad].copy_(x)
        return y

    @staticmethod
    def backward(ctx, g):
        return _CropCL.apply(g, ctx.pad), None


class _CropCL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, pad):
        ctx.pad = pad
        return g[:, :, pad:-pad, pad:-pad].contiguous(memory_format=torch.channels_last)

    @staticmethod
    def backward(ctx, gg):
        return _ZeroPadCL.apply(gg, ctx.pad), None


class Linear(nn.Module):
    """y = act(x @ (W * scale)^T + b), scale = gain / sqrt(fan_in)   (reference model.py:16-30)."""

    def __init__(self, in_features, out_features, bias, act_name='linear', gain=1.) -> None:
        super().__init__()
        self.act_name = act_name
        self.weight = nn.Parameter(torch.randn(out_features, in_features))
        self.bias = nn.Parameter(torch.zeros(out_features)) if bias else None
        self.scale = gain / (self.weight[0].numel() ** 0.5)

    def forward(self, x):
        x = F.linear(x, (self.weight * self.scale).to(x.dtype))
        return bias_act.bias_act(x, self.bias.to(x.dtype) if self.bias is not None else None, act=self.act_name)


class ModulatedConv(nn.Module):
    """reference model.py:32-74.  eps 1e-8; ``input_gain`` multiplies the weights AFTER demodulation, i.e. it scales the
    input channels but does not enter d."""

    def __init__(self, in_channels, out_channels, kernel_size=3, padding=1, demod=True) -> None:
        super().__init__()
        self.in_channels = in_channels
        self.padding = padding
        self.demod = demod
        self.weight = nn.Parameter(torch.randn(out_channels, in_channels, kernel_size, kernel_size))
        self.scale = 1 / (self.weight[0].numel() ** 0.5)

    def forward(self, x, s, input_gain=None):
        k = self.weight.shape[2]
        s = s.float()
        d = None
        if self.demod:
            wsq = self.weight.square().sum((2, 3))
            d = torch.rsqrt((s.square() @ wsq.t()) * (self.scale * self.scale) + 1e-8)
        s_in = s * input_gain if input_gain is not None else s
        extra = self.padding - k // 2                      # reference pads k-1; the kernel pads k//2
        assert extra >= 0
        w = self.weight * self.scale
        cout, cin = w.shape[0], w.shape[1]
        # planar -> channels-last with the zero border and the channel count rounded up to a 16-byte vector in ONE pass
        # (agf_planar_to_cl_pad); the conv then runs on padded channel counts and the way back crops them again
        cin_p, cout_p = layout.padded_channels(cin, x.dtype), layout.padded_channels(cout, x.dtype)
        x = layout.planar_to_channels_last(x, extra, cin_p)
        if cin_p != cin or cout_p != cout:
            w = F.pad(w, [0, 0, 0, 0, 0, cin_p - cin, 0, cout_p - cout])
            s_in = F.pad(s_in, [0, cin_p - cin])
            if d is not None:
                d = F.pad(d, [0, cout_p - cout])
        y = conv2d(x, w, s_in, d)
        return layout.channels_last_to_planar(y, 0, cout)  # planar for filtered_lrelu; its gradient returns channels-last


def design_filter(numtaps, cutoff, width, fs, radial=False):
    """Kaiser low-pass design of the reference (model.py:76-93): separable firwin, or the radially symmetric jinc."""
    assert numtaps >= 1
    if numtaps == 1:
        return None
    if not radial:
        return torch.as_tensor(scipy.signal.firwin(numtaps=numtaps, cutoff=cutoff, width=width, fs=fs), dtype=torch.float32)
    x = (np.arange(numtaps) - (numtaps - 1) / 2) / fs
    r = np.hypot(*np.meshgrid(x, x))
    f = scipy.special.j1(2 * cutoff * (np.pi * r)) / (np.pi * r)
    beta = scipy.signal.kaiser_beta(scipy.signal.kaiser_atten(numtaps, width / (fs / 2)))
    w = np.kaiser(numtaps, beta)
    f *= np.outer(w, w)
    f /= np.sum(f)
    return torch.as_tensor(f, dtype=torch.float32)


def get_layer_params(image_size, num_layers, channels, max_channels=512, image_channels=3, margin_size=10,
                     first_cutoff=2, first_stopband=2 ** 2.1, last_stopband_rel=2 ** 0.3, num_critical=2):
    """Per-layer channels / sizes / sampling rates / cutoffs / half widths (reference model.py:95-115)."""
    last_cutoff = image_size / 2
    last_stopband = last_cutoff * last_stopband_rel
    exponents = np.minimum(np.arange(num_layers + 1) / (num_layers - num_critical), 1)
    cutoffs = first_cutoff * (last_cutoff / first_cutoff) ** exponents
    stopbands = first_stopband * (last_stopband / first_stopband) ** exponents
    sampling_rates = np.exp2(np.ceil(np.log2(np.minimum(stopbands * 2, image_size))))
    half_widths = np.maximum(stopbands, sampling_rates / 2) - cutoffs
    sizes = sampling_rates + margin_size * 2
    sizes[-2:] = image_size
    channels = np.rint(np.minimum((channels / 2) / cutoffs, max_channels))
    channels[-1] = image_channels
    return channels, sizes, sampling_rates, cutoffs, half_widths


class StyleLayer(nn.Module):
    """modulated conv -> filtered leaky ReLU at a temporarily raised sampling rate (reference model.py:117-191)."""

    def __init__(self, in_channels, style_dim, out_channels, kernel_size, in_size, out_size,
                 in_sampling_rate, out_sampling_rate, in_cutoff, out_cutoff, in_half_width, out_half_width,
                 is_rgb, is_critical_sampled, lrelu_sampling=2, filter_size=6, conv_clamp=256, ema_decay=0.999) -> None:
        super().__init__()
        self.conv_clamp = conv_clamp
        self.ema_decay = ema_decay
        self.is_rgb = is_rgb
        self.gain = 1. if is_rgb else 2 ** 0.5
        self.negative_slope = 1. if is_rgb else 0.2
        self.affine = Linear(style_dim, in_channels, True)
        self.affine.bias.data.fill_(1.)
        self.register_buffer('ema', torch.ones([]))

        tmp_srate = max(in_sampling_rate, out_sampling_rate) * (1 if is_rgb else lrelu_sampling)
        self.up_factor = int(np.rint(tmp_srate / in_sampling_rate))
        assert in_sampling_rate * self.up_factor == tmp_srate
        up_taps = filter_size * self.up_factor if self.up_factor > 1 and not is_rgb else 1
        self.register_buffer('up_filter', design_filter(up_taps, in_cutoff, in_half_width * 2, tmp_srate))
        self.down_factor = int(np.rint(tmp_srate / out_sampling_rate))
        assert out_sampling_rate * self.down_factor == tmp_srate
        down_taps = filter_size * self.down_factor if self.down_factor > 1 and not is_rgb else 1
        self.register_buffer('down_filter', design_filter(down_taps, out_cutoff, out_half_width * 2, tmp_srate,
                                                          not is_critical_sampled))
        in_size = np.broadcast_to(np.asarray(in_size), [2])
        out_size = np.broadcast_to(np.asarray(out_size), [2])
        pad_total = (out_size - 1) * self.down_factor + 1
        pad_total -= (in_size + kernel_size - 1) * self.up_factor
        pad_total += up_taps + down_taps - 2
        pad_lo = (pad_total + self.up_factor) // 2
        pad_hi = pad_total - pad_lo
        self.padding = [int(pad_lo[0]), int(pad_hi[0]), int(pad_lo[1]), int(pad_hi[1])]
        self.conv = ModulatedConv(in_channels, out_channels, kernel_size, kernel_size - 1, not is_rgb)
        self.bias = nn.Parameter(torch.zeros(out_channels))

    def forward(self, x, w):
        if self.training:
            # mean(x^2) in fp32 (reference model.py:174-176).  vector_norm casts inside the reduction: one read of the bf16
            # activations instead of an fp32 copy + square + mean (three passes over a tensor of up to 630 MB)
            with torch.no_grad():
                stats = torch.linalg.vector_norm(x.detach(), 2, dtype=torch.float32).square() / x.numel()
                self.ema.copy_(stats.lerp_(self.ema, self.ema_decay))
        input_gain = self.ema.rsqrt()
        s = self.affine(w)
        x = self.conv(x, s, input_gain)
        return filtered_lrelu.filtered_lrelu(x, self.up_filter, self.down_filter, self.bias.to(x.dtype), self.up_factor,
                                             self.down_factor, self.padding, self.gain, self.negative_slope, self.conv_clamp)


class SynthesisInput(nn.Module):
    """Fourier-feature input with a learned rotation / translation (reference model.py:193-267); small fp32 torch math."""

    def __init__(self, style_dim, channels, size, sampling_rate, bandwidth) -> None:
        super().__init__()
        self.channels = channels
        self.bandwidth = bandwidth
        self.sampling_rate = sampling_rate
        self.size = list(map(int, (np.broadcast_to(np.asarray(size), [2]))))
        freqs = torch.randn(channels, 2)
        radii = freqs.square().sum(1, keepdim=True).sqrt()
        freqs /= radii * radii.square().exp().pow(0.25)
        freqs *= bandwidth
        phases = torch.rand(channels) - 0.5
        self.weight = nn.Parameter(torch.randn(channels, channels))
        self.scale = 1 / (channels ** 0.5)
        self.affine = Linear(style_dim, 4, True)
        self.affine.weight.data.fill_(0.)
        self.affine.bias.data.copy_(torch.tensor([1, 0, 0, 0], dtype=torch.float32))
        self.register_buffer('transform', torch.eye(3, 3))
        self.register_buffer('freqs', freqs)
        self.register_buffer('phases', phases)

    def forward(self, w):
        B, device = w.size(0), w.device
        t = self.affine(w.float())
        t = t / t[:, :2].norm(dim=1, keepdim=True)
        m_r = torch.eye(3, device=device).unsqueeze(0).repeat(B, 1, 1)
        m_r[:, 0, 0] = t[:, 0]
        m_r[:, 0, 1] = -t[:, 1]
        m_r[:, 1, 0] = t[:, 1]
        m_r[:, 1, 1] = t[:, 0]
        m_t = torch.eye(3, device=device).unsqueeze(0).repeat(B, 1, 1)
        m_t[:, 0, 2] = -t[:, 2]
        m_t[:, 1, 2] = -t[:, 3]
        transforms = m_r @ m_t @ self.transform.unsqueeze(0)
        phases = self.phases.unsqueeze(0) + (self.freqs.unsqueeze(0) @ transforms[:, :2, 2:]).squeeze(2)
        freqs = self.freqs.unsqueeze(0) @ transforms[:, :2, :2]
        amp = (1 - (freqs.norm(dim=2) - self.bandwidth) / (self.sampling_rate / 2 - self.bandwidth)).clamp(0, 1)
        theta = torch.eye(2, 3, device=device)
        theta[0, 0] = 0.5 * self.size[0] / self.sampling_rate
        theta[1, 1] = 0.5 * self.size[1] / self.sampling_rate
        grids = F.affine_grid(theta.unsqueeze(0), [1, 1, self.size[1], self.size[0]], align_corners=False)
        x = (grids.unsqueeze(3) @ freqs.permute(0, 2, 1).unsqueeze(1).unsqueeze(2)).squeeze(3)
        x = x + phases.unsqueeze(1).unsqueeze(2)
        x = torch.sin(x * (np.pi * 2))
        x = x * amp.unsqueeze(1).unsqueeze(2)
        x = F.linear(x, self.weight * self.scale)
        return x.permute(0, 3, 1, 2)


class PixelNorm(nn.Module):
    def forward(self, x):
        return x / x.pow(2).mean(dim=1, keepdim=True).sqrt().add(1e-8)


class Mapping(nn.Module):
    """reference model.py:275-306 (tracks ``w_avg`` in training mode)."""

    def __init__(self, latent_dim, style_dim, num_layers=2, pixel_norm=True, ema_decay=0.998) -> None:
        super().__init__()
        self.ema_decay = ema_decay
        if pixel_norm:
            self.norm = PixelNorm()
        layers = [Linear(latent_dim, style_dim, True, 'lrelu')]
        for _ in range(num_layers - 1):
            layers.append(Linear(style_dim, style_dim, True, 'lrelu'))
        self.net = nn.Sequential(*layers)
        self.register_buffer('w_avg', torch.zeros(style_dim))

    def forward(self, z, truncation_psi=1.):
        z = z.float()
        if hasattr(self, 'norm'):
            z = self.norm(z)
        w = self.net(z)
        if self.training:
            stats = w.detach().to(torch.float32).mean(dim=0)
            self.w_avg.copy_(stats.lerp(self.w_avg, self.ema_decay))
        if truncation_psi != 1:
            w = self.w_avg.lerp(w, truncation_psi)
        return w


class Synthesis(nn.Module):
    """reference model.py:308-359."""

    def __init__(self, image_size, num_layers=14, channels=32, max_channels=512, style_dim=512, image_channels=3,
                 output_scale=0.25, margin_size=10, first_cutoff=2, first_stopband=2 ** 2.1, last_stopband_rel=2 ** 0.3,
                 kernel_size=3, compute_dtype=torch.bfloat16) -> None:
        super().__init__()
        self.num_ws = num_layers + 2
        self.compute_dtype = compute_dtype
        log_resl_diff = int(math.log2(512) - math.log2(image_size))
        min_c_scale = channels / 64
        channels = int(2 ** (15 - log_resl_diff) * min_c_scale)
        channels, sizes, sampling_rates, cutoffs, half_widths = get_layer_params(
            image_size, num_layers, channels, max_channels, image_channels, margin_size, first_cutoff, first_stopband,
            last_stopband_rel, num_critical=2)
        self.input = SynthesisInput(style_dim, int(channels[0]), sizes[0], sampling_rates[0], cutoffs[0])
        layers = []
        for i in range(num_layers + 1):
            prev = max(i - 1, 0)
            is_rgb = i == num_layers
            layers.append(StyleLayer(
                int(channels[prev]), style_dim, int(channels[i]), 1 if is_rgb else kernel_size,
                int(sizes[prev]), int(sizes[i]), sampling_rates[prev], sampling_rates[i], cutoffs[prev], cutoffs[i],
                half_widths[prev], half_widths[i], is_rgb, i >= num_layers - 2))
        self.net = nn.ModuleList(layers)
        self.register_buffer('output_scale', torch.tensor([output_scale]))

    def forward(self, w):
        if w.ndim == 2:
            w = w.unsqueeze(1).repeat(1, self.num_ws, 1)
        ws = w.unbind(dim=1)
        x = self.input(ws[0])
        if _native(x):
            x = x.to(self.compute_dtype)
        for module, w_i in zip(self.net, ws[1:]):
            x = module(x, w_i)
        return x.float() * self.output_scale


class Generator(nn.Module):
    def __init__(self, image_size, latent_dim, num_layers=14, map_num_layers=2, channels=32, max_channels=512,
                 style_dim=512, pixel_norm=True, image_channels=3, output_scale=0.25, margin_size=10, first_cutoff=2,
                 first_stopband=2 ** 2.1, last_stopband_rel=2 ** 0.3, kernel_size=3, compute_dtype=torch.bfloat16) -> None:
        super().__init__()
        self.map = Mapping(latent_dim, style_dim, map_num_layers, pixel_norm)
        self.synthesis = Synthesis(image_size, num_layers, channels, max_channels, style_dim, image_channels, output_scale,
                                   margin_size, first_cutoff, first_stopband, last_stopband_rel, kernel_size, compute_dtype)

    def forward(self, z, truncation_psi=1.):
        return self.synthesis(self.map(z, truncation_psi))


def binomial_filter(filter_size):
    """Row ``filter_size - 1`` of Pascal's triangle (reference model.py:382-387)."""
    return [math.comb(filter_size - 1, j) for j in range(filter_size)]


class ConvAct(nn.Module):
    """conv (optionally FIR-downsampled by 2) + bias + activation (reference model.py:389-417)."""

    def __init__(self, in_channels, out_channels, kernel_size, bias=True, down=1, filter_size=4, act_name='linear',
                 gain=1., act_gain=None) -> None:
        super().__init__()
        self.down = down
        self.act_name = act_name
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.scale = gain / (self.weight[0].numel() ** 0.5)
        self.act_gain = bias_act.activation_funcs[act_name].def_gain if act_gain is None else act_gain
        if down > 1:
            taps = torch.tensor(binomial_filter(filter_size), dtype=torch.float32)
            kernel = torch.outer(taps, taps)
            kernel /= kernel.sum()
            self.register_buffer('down_filter', kernel)
        else:
            self.down_filter = None

    def forward(self, x):
        weight = self.weight * self.scale
        k = self.weight.shape[2]
        if not _native(x):
            x = conv2d_resample.conv2d_resample(x, weight.to(x.dtype), self.down_filter, 1, self.down, self.padding)
        elif self.down == 1:
            # MFMA conv ("same" padding) with bias + activation + gain in its epilogue (and the fused backward of that epilogue)
            return conv2d_act(x.contiguous(memory_format=torch.channels_last), self.weight, self.bias, coef=self.scale,
                              act=self.act_name, gain=self.act_gain) if self.act_name in ('lrelu', 'linear') else \
                bias_act.bias_act(conv2d(x.contiguous(memory_format=torch.channels_last), weight),
                                  self.bias.to(x.dtype) if self.bias is not None else None, act=self.act_name, gain=self.act_gain)
        elif k == 1:
            # conv2d_resample's "1x1 + down" order (conv2d_resample.py:88-91): FIR-downsample first, then the 1x1 conv on MFMA
            f = self.down_filter
            p0, p1 = (f.shape[-1] - self.down + 1) // 2, (f.shape[-1] - self.down) // 2
            x = upfirdn2d.upfirdn2d(x, f, down=self.down, padding=[p0, p1, p0, p1])
            if self.act_name in ('lrelu', 'linear'):
                return conv2d_act(x.contiguous(memory_format=torch.channels_last), self.weight, self.bias, coef=self.scale,
                                  act=self.act_name, gain=self.act_gain)
            x = conv2d(x.contiguous(memory_format=torch.channels_last), weight)
        else:
            # FIR + stride-2 conv of conv2d_resample.py:100-103.  The FIR and the (channel-mixing) conv commute, so the conv
            # runs first, at stride 1 on the MFMA kernel over the input zero-padded by one more pixel, and ONE upfirdn2d then
            # filters and decimates: same linear map, every piece double-differentiable on this package's kernels (MIOpen's
            # double backward of a strided conv lands on its "naive" kernels: 12 s per R1 iteration at 256x256).
            assert self.down == 2 and k == 3 and self.padding == 1
            x = conv2d(_ZeroPadCL.apply(x, 1), weight)
            x = upfirdn2d.upfirdn2d(x, self.down_filter, down=self.down, padding=0)
        b = self.bias.to(x.dtype) if self.bias is not None else None
        return bias_act.bias_act(x, b, act=self.act_name, gain=self.act_gain)


class ResBlock(nn.Module):
    def __init__(self, in_channels, out_channels, filter_size=4, act_name='lrelu', gain=1.) -> None:
        super().__init__()
        self.conv1 = ConvAct(in_channels, out_channels, 3, True, 1, filter_size, act_name, gain)
        self.conv2 = ConvAct(out_channels, out_channels, 3, True, 2, filter_size, act_name, gain, 0.5 ** 0.5)
        self.skip = ConvAct(in_channels, out_channels, 1, False, 2, filter_size, 'linear', gain, 0.5 ** 0.5)

    def forward(self, x):
        return self.conv2(self.conv1(x)) + self.skip(x)


class MinibatchStdDev(torch.nn.Module):
    """reference model.py:442-462."""

    def __init__(self, group_size, num_channels=1):
        super().__init__()
        self.group_size = group_size
        self.num_channels = num_channels

    def forward(self, x):
        N, C, H, W = x.shape
        G = self.group_size if N % self.group_size == 0 else N
        Fc = self.num_channels
        c = C // Fc
        y = x.float().reshape(G, -1, Fc, c, H, W)
        y = y - y.mean(dim=0)
        y = y.square().mean(dim=0)
        y = (y + 1e-8).sqrt()
        y = y.mean(dim=[2, 3, 4])
        y = y.reshape(-1, Fc, 1, 1)
        y = y.repeat(G, 1, H, W)
        return torch.cat([x, y.to(x.dtype)], dim=1)


class DiscEpilogue(nn.Module):
    def __init__(self, mbsd_group_size, mbsd_channels, channels, bottom, act_name='lrelu', gain=1.) -> None:
        super().__init__()
        self.epilogue = nn.Sequential(
            MinibatchStdDev(mbsd_group_size, mbsd_channels),
            ConvAct(channels + mbsd_channels, channels, 3, True, 1, None, act_name, gain),
            nn.Flatten(),
            Linear(channels * bottom ** 2, channels, True, act_name, gain),
            Linear(channels, 1, True, 'linear', gain))

    def forward(self, x):
        return self.epilogue(x)


class Discriminator(nn.Module):
    """reference model.py:464-510."""

    def __init__(self, image_size, in_channels=3, channels=64, max_channels=512, kernel_size=3, mbsd_group_size=4,
                 mbsd_channels=1, bottom=4, filter_size=4, act_name='lrelu', gain=1., compute_dtype=torch.bfloat16) -> None:
        super().__init__()
        self.compute_dtype = compute_dtype
        num_downs = int(math.log2(image_size) - math.log2(bottom))
        ochannels = channels
        self.from_rgb = ConvAct(in_channels, ochannels, 1, True, 1, None, act_name, gain)
        resblocks = []
        for _ in range(num_downs):
            channels *= 2
            ichannels, ochannels = ochannels, min(max_channels, channels)
            resblocks.append(ResBlock(ichannels, ochannels, filter_size, act_name, gain))
        self.resblocks = nn.Sequential(*resblocks)
        self.epilogue = DiscEpilogue(mbsd_group_size, mbsd_channels, ochannels, bottom, act_name, gain)

    def forward(self, x):
        if _native(x):
            x = x.to(self.compute_dtype)
        x = self.from_rgb(x)
        x = self.resblocks(x)
        return self.epilogue(x).float()
