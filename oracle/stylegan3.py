"""Oracle: the reference's StyleGAN3 generator / discriminator, stated functionally over a flat ``state_dict``.
TEST INFRASTRUCTURE ONLY (imported by tests/ and __graft_entry__.smoke(); never by the product).

The reference builds ``nn.Module`` trees (``implementations/StyleGAN3/model.py``); this restatement takes the state_dict of
those modules (same key names; the design filters ``up_filter`` / ``down_filter`` and the running statistics ``ema`` / ``w_avg``
are buffers of that dict) and evaluates the same arithmetic with the oracle operators + stock torch ops on CPU.  It evaluates
the modules as the reference's modules run in ``train()`` mode as far as outputs go, but is side-effect free: the updated
``ema`` / ``w_avg`` values are returned instead of being written.  Pinned by tests/golden/sg3_model.npz (outputs of the
reference's own modules, tools/make_golden.py)."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import upfirdn2d as U
from . import bias_act as B
from . import filtered_lrelu as FL
from . import stylegan2 as _S2
from .stylegan2 import bf16_storage, q          # noqa: F401  (``with bf16_storage():`` rounds where the product stores bf16, see there)


def linear(sd, prefix, x, act='linear', gain=1.0):
    """model.py:16-30: F.linear(x, W * gain/sqrt(fan_in)) then bias_act."""
    w = sd[prefix + '.weight']
    x = F.linear(x, w * (gain / math.sqrt(w.shape[1])))
    b = sd.get(prefix + '.bias')
    return B.bias_act(x, b, act=act)


def modulated_conv(sd, prefix, x, s, demod=True, input_gain=None):
    """model.py:46-74: per-sample weights W*scale*s, demodulated with eps 1e-8, then * input_gain; grouped conv, padding k-1."""
    w = sd[prefix + '.weight']
    Bn = x.shape[0]
    Cout, Cin, k, _ = w.shape
    if _S2._BF16_STORAGE:
        # the product's factorisation (animeface_amd/implementations/StyleGAN3/model.py ModulatedConv): d * conv(bf16(x * s * gain), bf16(W * scale)),
        # d in fp32 from the unrounded weights; the conv output is stored in bf16
        scale = 1 / math.sqrt(Cin * k * k)
        d = torch.rsqrt((s.square() @ w.square().sum((2, 3)).t()) * (scale * scale) + 1e-8) if demod else None
        s_in = s * input_gain if input_gain is not None else s
        y = F.conv2d(q(x * s_in[:, :, None, None]), q(w * scale), padding=k - 1)
        if d is not None:
            y = y * d[:, :, None, None]
        return q(y)
    wm = w[None] * (1 / math.sqrt(Cin * k * k)) * s[:, None, :, None, None]
    if demod:
        wm = wm * torch.rsqrt(wm.square().sum([2, 3, 4], keepdim=True) + 1e-8)
    if input_gain is not None:
        wm = wm * input_gain.expand(Bn, Cin)[:, None, :, None, None] if input_gain.ndim else wm * input_gain
    y = F.conv2d(x.reshape(1, Bn * Cin, *x.shape[2:]), wm.reshape(Bn * Cout, Cin, k, k), padding=k - 1, groups=Bn)
    return y.reshape(Bn, Cout, *y.shape[2:])


def layer_config(image_size, num_layers, channels, max_channels=512, image_channels=3, margin_size=10,
                 first_cutoff=2, first_stopband=2 ** 2.1, last_stopband_rel=2 ** 0.3, num_critical=2):
    """model.py:95-115 (channels here is the already scaled base passed by Synthesis, model.py:321-324)."""
    last_cutoff = image_size / 2
    last_stopband = last_cutoff * last_stopband_rel
    expo = np.minimum(np.arange(num_layers + 1) / (num_layers - num_critical), 1)
    cutoffs = first_cutoff * (last_cutoff / first_cutoff) ** expo
    stopbands = first_stopband * (last_stopband / first_stopband) ** expo
    rates = np.exp2(np.ceil(np.log2(np.minimum(stopbands * 2, image_size))))
    half_widths = np.maximum(stopbands, rates / 2) - cutoffs
    sizes = rates + margin_size * 2
    sizes[-2:] = image_size
    ch = np.rint(np.minimum((channels / 2) / cutoffs, max_channels))
    ch[-1] = image_channels
    return ch, sizes, rates, cutoffs, half_widths


class Config:
    """Constructor arguments of Generator / Discriminator (model.py:361-376, 464-470)."""

    def __init__(self, image_size, latent_dim, num_layers=14, map_num_layers=2, channels=32, max_channels=512, style_dim=512,
                 pixel_norm=True, image_channels=3, output_scale=0.25, margin_size=10, kernel_size=3,
                 d_channels=64, d_max_channels=512, mbsd_group_size=4, mbsd_channels=1, bottom=4):
        self.__dict__.update(locals())
        base = int(2 ** (15 - int(math.log2(512) - math.log2(image_size))) * (channels / 64))
        self.ch, self.sizes, self.rates, self.cutoffs, self.half_widths = layer_config(
            image_size, num_layers, base, max_channels, image_channels, margin_size)

    def layer(self, i, lrelu_sampling=2, filter_size=6):
        """Static per-layer numbers of StyleLayer.__init__ (model.py:140-167)."""
        prev = max(i - 1, 0)
        is_rgb = i == self.num_layers
        k = 1 if is_rgb else self.kernel_size
        in_rate, out_rate = self.rates[prev], self.rates[i]
        tmp = max(in_rate, out_rate) * (1 if is_rgb else lrelu_sampling)
        up, down = int(np.rint(tmp / in_rate)), int(np.rint(tmp / out_rate))
        up_taps = filter_size * up if up > 1 and not is_rgb else 1
        down_taps = filter_size * down if down > 1 and not is_rgb else 1
        in_size, out_size = int(self.sizes[prev]), int(self.sizes[i])
        pad_total = (out_size - 1) * down + 1 - (in_size + k - 1) * up + up_taps + down_taps - 2
        lo = (pad_total + up) // 2
        return dict(k=k, up=up, down=down, padding=[lo, pad_total - lo, lo, pad_total - lo], is_rgb=is_rgb,
                    gain=1. if is_rgb else math.sqrt(2), slope=1. if is_rgb else 0.2)


def synthesis_input(sd, prefix, w, cfg):
    """model.py:221-267: learned-transform Fourier features."""
    Bn = w.shape[0]
    size = int(cfg.sizes[0])
    rate, bandwidth = cfg.rates[0], cfg.cutoffs[0]
    t = linear(sd, prefix + '.affine', w.float())
    t = t / t[:, :2].norm(dim=1, keepdim=True)
    rot = torch.eye(3).repeat(Bn, 1, 1)
    rot[:, 0, 0], rot[:, 0, 1], rot[:, 1, 0], rot[:, 1, 1] = t[:, 0], -t[:, 1], t[:, 1], t[:, 0]
    trn = torch.eye(3).repeat(Bn, 1, 1)
    trn[:, 0, 2], trn[:, 1, 2] = -t[:, 2], -t[:, 3]
    tf = rot @ trn @ sd[prefix + '.transform'][None]
    freqs0, phases0 = sd[prefix + '.freqs'], sd[prefix + '.phases']
    phases = phases0[None] + (freqs0[None] @ tf[:, :2, 2:]).squeeze(2)
    freqs = freqs0[None] @ tf[:, :2, :2]
    amp = (1 - (freqs.norm(dim=2) - bandwidth) / (rate / 2 - bandwidth)).clamp(0, 1)
    theta = torch.eye(2, 3)
    theta[0, 0] = theta[1, 1] = 0.5 * size / rate
    grid = F.affine_grid(theta[None], [1, 1, size, size], align_corners=False)
    x = (grid.unsqueeze(3) @ freqs.permute(0, 2, 1).unsqueeze(1).unsqueeze(2)).squeeze(3)
    x = torch.sin((x + phases[:, None, None]) * (2 * np.pi)) * amp[:, None, None]
    wgt = sd[prefix + '.weight']
    return F.linear(x, wgt / math.sqrt(wgt.shape[1])).permute(0, 3, 1, 2)


def mapping(sd, cfg, z, truncation_psi=1., training=True):
    """model.py:275-306.  Returns (w, new w_avg)."""
    z = z.float()
    if cfg.pixel_norm:
        z = z / z.pow(2).mean(dim=1, keepdim=True).sqrt().add(1e-8)
    w = z
    for i in range(cfg.map_num_layers):
        w = linear(sd, f'map.net.{i}', w, act='lrelu')
    w_avg = sd['map.w_avg']
    if training:
        w_avg = w.detach().mean(0).lerp(w_avg, 0.998)
    if truncation_psi != 1:
        w = w_avg.lerp(w, truncation_psi)
    return w, w_avg


def synthesis(sd, cfg, w, training=True, ema_decay=0.999, conv_clamp=256, collect=None):
    """model.py:169-191, 346-359.  Returns (image, {layer index: new ema})."""
    x = q(synthesis_input(sd, 'synthesis.input', w, cfg))
    emas = {}
    for i in range(cfg.num_layers + 1):
        pre = f'synthesis.net.{i}'
        L = cfg.layer(i)
        ema = sd[pre + '.ema']
        if training:
            ema = x.detach().float().square().mean().lerp(ema, ema_decay)
            emas[i] = ema
        s = linear(sd, pre + '.affine', w)
        x = modulated_conv(sd, pre + '.conv', x, s, demod=not L['is_rgb'], input_gain=ema.rsqrt())
        x = q(FL.filtered_lrelu(x, sd.get(pre + '.up_filter'), sd.get(pre + '.down_filter'), q(sd[pre + '.bias']), L['up'], L['down'],
                                L['padding'], L['gain'], L['slope'], conv_clamp))
        if collect is not None:
            collect.append(x.detach())
    return x.float() * sd['synthesis.output_scale'], emas


def generator(sd, cfg, z, truncation_psi=1., training=True, collect=None):
    """``collect``: list that receives every layer's output (tests compare layer by layer)."""
    w, w_avg = mapping(sd, cfg, z, truncation_psi, training)
    image, emas = synthesis(sd, cfg, w, training, collect=collect)
    return image, dict(w_avg=w_avg, ema=emas)


RESBLOCK_SUM_IN_SKIP_CONV = True     # which tensors the product stores in bf16 (``bf16_storage`` only; the fp32 arithmetic is the reference's either way)


def conv_act(sd, prefix, x, k, down=1, act='linear', gain=1., act_gain=None, store=True):
    """model.py:389-417: conv2d_resample(down) + bias_act; act_gain defaults to the activation's own gain."""
    w = sd[prefix + '.weight']
    if act_gain is None:
        act_gain = math.sqrt(2) if act == 'lrelu' else 1.
    if not store and act == 'linear' and _S2._BF16_STORAGE:
        # (bf16 storage, the ResBlock's skip branch: the product folds the branch gain into the prepared weights -- same real arithmetic)
        w = q(w * (gain / math.sqrt(w[0].numel()) * act_gain))
        act_gain = 1.
    else:
        w = q(w * (gain / math.sqrt(w[0].numel())))            # (bf16 storage: the prepared weights; the FIR output and the conv output are stored too)
    f = sd.get(prefix + '.down_filter')
    pad = k // 2
    if down == 1:
        x = F.conv2d(x, w, padding=pad)
    else:
        fw = f.shape[-1]
        p0, p1 = pad + (fw - down + 1) // 2, pad + (fw - down) // 2
        if k == 1:                                              # conv2d_resample.py:88-91
            x = F.conv2d(q(U.upfirdn2d(x, f, down=down, padding=[p0, p1, p0, p1])), w)
        else:                                                   # conv2d_resample.py:100-103
            x = F.conv2d(q(U.upfirdn2d(x, f, padding=[p0, p1, p0, p1])), w, stride=down)
    b = sd.get(prefix + '.bias')
    if _S2._BF16_STORAGE and not (down == 1 and act in ('lrelu', 'linear')) and not (down > 1 and k == 1 and act in ('lrelu', 'linear')):
        x = q(x)                                                # a separate bias_act launch reads the stored conv output (bias in bf16)
        b = q(b) if b is not None else None
    y = B.bias_act(x, b, act=act, gain=act_gain)
    return q(y) if store else y


def minibatch_stddev(x, group_size, num_channels=1):
    """model.py:442-462."""
    N, C, H, W = x.shape
    G = group_size if N % group_size == 0 else N
    y = x.float().reshape(G, -1, num_channels, C // num_channels, H, W)
    y = ((y - y.mean(0)).square().mean(0) + 1e-8).sqrt().mean([2, 3, 4])
    y = y.reshape(-1, num_channels, 1, 1).repeat(G, 1, H, W)
    return torch.cat([x, y.to(x.dtype)], dim=1)


def discriminator(sd, cfg, x, collect=None):
    """model.py:464-510.  ``collect``: list that receives every residual block's output."""
    x = conv_act(sd, 'from_rgb', q(x), 1, act='lrelu')
    n = int(math.log2(cfg.image_size) - math.log2(cfg.bottom))
    for i in range(n):
        pre = f'resblocks.{i}'
        t = conv_act(sd, pre + '.conv1', x, 3, act='lrelu')
        t = conv_act(sd, pre + '.conv2', t, 3, down=2, act='lrelu', act_gain=math.sqrt(0.5))
        # (bf16 storage: the product's skip conv takes t as its residual operand and stores the SUM -- the skip branch itself is never stored)
        x = q(t + conv_act(sd, pre + '.skip', x, 1, down=2, act='linear', act_gain=math.sqrt(0.5), store=not RESBLOCK_SUM_IN_SKIP_CONV))
        if collect is not None:
            collect.append(x.detach())
    x = minibatch_stddev(x, cfg.mbsd_group_size, cfg.mbsd_channels)
    x = conv_act(sd, 'epilogue.epilogue.1', x, 3, act='lrelu')
    x = linear(sd, 'epilogue.epilogue.3', x.flatten(1), act='lrelu')
    return linear(sd, 'epilogue.epilogue.4', x)
