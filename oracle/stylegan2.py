"""Oracle: the reference's StyleGAN2 generator / discriminator, stated functionally
over a flat ``state_dict``.  TEST INFRASTRUCTURE ONLY.

The reference builds ``nn.Module`` trees (``implementations/StyleGAN2/model.py``);
this restatement takes the *state_dict of those modules* (same key names, so a
``G_*.pt`` written by the reference loads directly) and evaluates the same
arithmetic with stock torch ops on CPU.  Gradients of any order come from
autograd on the dict's tensors.  Each block cites the lines it follows.
"""
import math
import numpy as np
import torch
import torch.nn.functional as F


class Config:
    """Constructor arguments of Generator / Discriminator (model.py:336-340,371)."""

    def __init__(self, image_size=128, image_channels=3, style_dim=512, channels=32, max_channels=512,
                 block_num_conv=2, map_num_layers=8, normalize_latent=True, map_lr=0.01, mbsd_groups=4):
        self.image_size = image_size
        self.image_channels = image_channels
        self.style_dim = style_dim
        self.channels = channels
        self.max_channels = max_channels
        self.block_num_conv = block_num_conv
        self.map_num_layers = map_num_layers
        self.normalize_latent = normalize_latent
        self.map_lr = map_lr
        self.mbsd_groups = mbsd_groups

    def synthesis_channels(self):
        """[(resolution, in_ch, out_ch)] for every StyleBlock (model.py:288-309)."""
        ch = self.channels * (2 ** int(np.log2(self.image_size) - 2))
        oc = min(self.max_channels, ch)
        out, resl = [], 4
        first = oc
        while resl < self.image_size:
            resl *= 2
            ch = ch // 2
            ic, oc = oc, min(self.max_channels, ch)
            out.append((resl, ic, oc))
        return first, out

    def discriminator_channels(self):
        """[(in_ch, out_ch)] for every DBlock (model.py:373-387)."""
        ch, oc, resl, out = self.channels, self.channels, self.image_size, []
        while resl > 4:
            resl //= 2
            ch *= 2
            ic, oc = oc, min(self.max_channels, ch)
            out.append((ic, oc))
        return out, oc, resl


# ---------------------------------------------------------------------------------------------
# bf16-storage emulation.  The product keeps activations and the prepared weights in bf16 and accumulates in fp32; inside
# ``with bf16_storage():`` this restatement rounds to bf16 at the SAME points (conv operands: x * style and W * coef; every tensor
# a kernel writes: conv + epilogue output, FIR output, RGB skip sums), so the MFMA network path can be compared with it at a few bf16
# ulps instead of the ~6e-2 that separates bf16 from the fp32 reference.  Gradients flow through the rounding unchanged (straight
# through), i.e. they are the fp32 gradients of the rounded forward pass.  Off by default: the fp32 restatement of the reference.

_BF16_STORAGE = False


class bf16_storage:
    def __enter__(self):
        global _BF16_STORAGE
        self.prev, _BF16_STORAGE = _BF16_STORAGE, True

    def __exit__(self, *a):
        global _BF16_STORAGE
        _BF16_STORAGE = self.prev


def q(x):
    if not _BF16_STORAGE:
        return x
    return x + (x.to(torch.bfloat16).to(x.dtype) - x).detach()


# ---------------------------------------------------------------------------------------------
# layers

def elr_linear(sd, prefix, x, gain=1.0):
    """ELR(nn.Linear): x*coef then layer (model.py:29-37,44-47); coef = gain/sqrt(fan_in)."""
    w, b = sd[prefix + '.layer.weight'], sd[prefix + '.layer.bias']
    coef = gain / math.sqrt(w[0].numel())
    return F.linear(x * coef, w, b)


def elr_conv(sd, prefix, x, padding):
    """ELR(nn.Conv2d) (model.py:29-37,50-53)."""
    w, b = sd[prefix + '.layer.weight'], sd[prefix + '.layer.bias']
    coef = 1.0 / math.sqrt(w[0].numel())
    if _BF16_STORAGE:                                   # the product folds coef into the prepared (bf16) weights
        return F.conv2d(q(x), q(w * coef), b, padding=padding)
    return F.conv2d(x * coef, w, b, padding=padding)


def modulated_conv2d(sd, prefix, x, y, demod=True, gain=1.0):
    """ModulatedConv2d.forward (model.py:106-132)."""
    B, _, H, W = x.shape
    weight, bias = sd[prefix + '.weight'], sd[prefix + '.bias']
    cout, cin, k, _ = weight.shape
    s = elr_linear(sd, prefix + '.affine', y) + 1                                   # :110
    coef = gain / math.sqrt(weight[0].numel())                                      # :105
    w = weight[None] * s[:, None, :, None, None] * coef                             # :115
    pad = (k - 1) // 2                                                              # :134-135 with stride 1
    if _BF16_STORAGE:
        # the product's factorisation: d * conv(bf16(x * s), bf16(W * coef)) with d from the unrounded fp32 weights
        d = torch.rsqrt(w.pow(2).sum([2, 3, 4]) + 1e-4) if demod else None
        out = F.conv2d(q(x * s[:, :, None, None]), q(weight * coef), padding=pad)
        if d is not None:
            out = out * d[:, :, None, None]
        return out + bias
    if demod:
        w = w * torch.rsqrt(w.pow(2).sum([2, 3, 4], keepdim=True) + 1e-4)           # :118-120
    out = F.conv2d(x.reshape(1, B * cin, H, W), w.reshape(B * cout, cin, k, k), padding=pad, groups=B)   # :123-129
    return out.reshape(B, cout, H, W) + bias                                        # :132


def blur2d(x):
    """Blur2d.forward (model.py:138-149): depthwise [1,2,1]x[1,2,1]/16, zero pad 1."""
    k = torch.tensor([[1., 2., 1.], [2., 4., 2.], [1., 2., 1.]], dtype=x.dtype, device=x.device) / 16.0
    C = x.shape[1]
    return F.conv2d(x, k[None, None].expand(C, 1, 3, 3), padding=1, groups=C)


def upsample2x(x):
    """Upsample2x('bilinear') (model.py:56-58)."""
    return F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)


def pixel_norm(x):
    """PixelNorm.forward (model.py:253-256): eps added AFTER the sqrt."""
    return x / (x.pow(2).mean(dim=1, keepdim=True).sqrt() + 1e-4)


def minibatch_stddev(x, group_size, eps=1e-4):
    """MiniBatchStdDev.forward (model.py:221-236)."""
    B, C, H, W = x.shape
    g = group_size if B % group_size == 0 else B
    y = x.reshape(g, -1, C, H, W)
    y = y - y.mean(0, keepdim=True)
    y = (y.square().mean(0) + eps).sqrt()
    y = y.mean([1, 2, 3], keepdim=True)
    y = y.repeat(g, 1, H, W)
    return torch.cat([x, y], dim=1)


# ---------------------------------------------------------------------------------------------
# networks

def mapping(sd, cfg, z, prefix='map'):
    """Mapping.forward (model.py:279-282); MapLinear (model.py:71-78)."""
    x = pixel_norm(z) if cfg.normalize_latent else z
    for i in range(cfg.map_num_layers):
        x = elr_linear(sd, f'{prefix}.map.{2 * i}.linear', x) * cfg.map_lr
        x = F.leaky_relu(x, 0.2)
    return x


class NoiseSource:
    """InjectNoise draws ``torch.randn(B,1,H,W)`` per call (model.py:85-88).  ``draws``
    replays a recorded list instead (for golden tests); otherwise draws are recorded."""

    def __init__(self, draws=None, generator=None):
        self.replay = list(draws) if draws is not None else None
        self.generator = generator
        self.record = []

    def __call__(self, B, H, W, device, dtype):
        if self.replay is not None:
            n = self.replay.pop(0).to(device=device, dtype=dtype)
            assert tuple(n.shape) == (B, 1, H, W)
        else:
            n = torch.randn(B, 1, H, W, device=device, generator=self.generator).to(dtype)
        self.record.append(n)
        return n


def synthesis(sd, cfg, x, styles, noise, prefix='synthesis', collect=None):
    """Synthesis.forward (model.py:312-332) with ``styles`` already a per-layer list.  ``collect``: optional list that receives the
    feature map after the input layer and after every StyleBlock."""
    _, blocks = cfg.synthesis_channels()
    keep = (lambda v: collect.append(v.detach())) if collect is not None else (lambda v: None)
    x = q(modulated_conv2d(sd, f'{prefix}.input', q(x), styles[0]))                             # :324
    keep(x)
    pre = q(upsample2x(q(modulated_conv2d(sd, f'{prefix}.input_to_image.conv', x, styles[0], demod=False))))   # :325, ToImage :244-250
    image = pre
    for i, (resl, _ic, _oc) in enumerate(blocks):
        y = styles[i + 1]
        # StyleBlock (model.py:154-180): up, blur, [modconv, noise, lrelu] * num_conv
        x = q(blur2d(upsample2x(x)))                      # (the product runs the pair as one FIR pass: one rounding)
        for j in range(cfg.block_num_conv):
            x = modulated_conv2d(sd, f'{prefix}.blocks.{i}.block.{2 + 3 * j}', x, y)
            x = x + noise(x.shape[0], x.shape[2], x.shape[3], x.device, x.dtype)              # InjectNoise: unscaled (F10)
            x = q(F.leaky_relu(x, 0.2))
        keep(x)
        image = q(q(modulated_conv2d(sd, f'{prefix}.to_images.{i}.conv', x, y, demod=False)) + pre)  # ToImage :245-247
        if resl < cfg.image_size:
            image = q(upsample2x(image))                                                       # :248-249
        pre = image
    return torch.tanh(image)                                                                    # :332


def generator(sd, cfg, z, noise=None, injection=None, collect=None):
    """Generator.forward (model.py:351-363).  Returns (image, style)."""
    noise = noise if noise is not None else NoiseSource()
    n_layers = len(cfg.synthesis_channels()[1]) + 1
    if isinstance(z, (list, tuple)):                                                            # style mixing :354-356, :315-320
        style = [mapping(sd, cfg, z[0]), mapping(sd, cfg, z[1])]
        B = z[0].shape[0]
        assert injection is not None and injection <= n_layers
        styles = [style[0]] * injection + [style[1]] * (n_layers - injection)
    else:
        style = mapping(sd, cfg, z)
        B = z.shape[0]
        styles = [style] * n_layers
    x = sd['const'].expand(B, -1, -1, -1)
    return synthesis(sd, cfg, x, styles, noise, collect=collect), style


def d_block(sd, prefix, x, num_conv):
    """DBlock.forward (model.py:204-212)."""
    t = x
    for j in range(num_conv):
        x = q(F.leaky_relu(elr_conv(sd, f'{prefix}.block.{2 * j}', x, 1), 0.2))
    if _BF16_STORAGE:
        # the product's order: pool the block input first (it commutes with the 1x1 conv), 1/sqrt(2) folded into the skip weights and
        # the pooling gain, the sum formed in the skip conv's epilogue
        c = 1 / np.sqrt(2)
        w, b = sd[f'{prefix}.skip.layer.weight'], sd[f'{prefix}.skip.layer.bias']
        coef = c / math.sqrt(w[0].numel())
        return q(F.conv2d(q(F.avg_pool2d(t, 2)), q(w * coef), b * c) + q(F.avg_pool2d(x, 2) * c))
    t = elr_conv(sd, f'{prefix}.skip', t, 0)
    x = F.avg_pool2d(x, 2)
    t = F.avg_pool2d(t, 2)
    return (x + t) / np.sqrt(2)


def discriminator(sd, cfg, x, collect=None):
    """Discriminator.forward (model.py:398-401) over the nn.Sequential of :375-397.  ``collect``: optional list that receives the
    activation after from_rgb, after every DBlock and after the last conv (layer-wise parity tests)."""
    dblocks, oc, resl = cfg.discriminator_channels()
    keep = (lambda v: collect.append(v.detach())) if collect is not None else (lambda v: None)
    x = q(F.leaky_relu(elr_conv(sd, 'from_rgb.0', q(x), 0), 0.2))
    keep(x)
    for i in range(len(dblocks)):
        x = d_block(sd, f'blocks.{i}', x, cfg.block_num_conv)
        keep(x)
    n = len(dblocks)
    x = q(minibatch_stddev(x, cfg.mbsd_groups))                                 # blocks.{n}
    x = q(F.leaky_relu(elr_conv(sd, f'blocks.{n + 1}', x, 1), 0.2))             # blocks.{n+1}, {n+2}
    keep(x)
    x = x.reshape(x.shape[0], -1)                                               # Flatten blocks.{n+3}
    x = F.leaky_relu(elr_linear(sd, f'blocks.{n + 4}', x), 0.2)                 # blocks.{n+4}, {n+5}
    return elr_linear(sd, f'blocks.{n + 6}', x)                                 # blocks.{n+6}


# ---------------------------------------------------------------------------------------------
# parameter construction (shapes only; values by init_weight_N01, model.py:404-408)

def generator_state_shapes(cfg):
    first, blocks = cfg.synthesis_channels()
    sh = {'const': (1, cfg.style_dim, 4, 4)}
    for i in range(cfg.map_num_layers):
        sh[f'map.map.{2 * i}.linear.layer.weight'] = (cfg.style_dim, cfg.style_dim)
        sh[f'map.map.{2 * i}.linear.layer.bias'] = (cfg.style_dim,)

    def modconv(p, ic, oc, k):
        sh[p + '.affine.layer.weight'] = (ic, cfg.style_dim)
        sh[p + '.affine.layer.bias'] = (ic,)
        sh[p + '.weight'] = (oc, ic, k, k)
        sh[p + '.bias'] = (1, oc, 1, 1)

    modconv('synthesis.input', cfg.style_dim, first, 3)
    modconv('synthesis.input_to_image.conv', first, cfg.image_channels, 1)
    for i, (_r, ic, oc) in enumerate(blocks):
        sh[f'synthesis.blocks.{i}.block.1.kernel'] = (1, 3, 3)
        for j in range(cfg.block_num_conv):
            modconv(f'synthesis.blocks.{i}.block.{2 + 3 * j}', ic if j == 0 else oc, oc, 3)
            sh[f'synthesis.blocks.{i}.block.{3 + 3 * j}.scale'] = (1,)
        modconv(f'synthesis.to_images.{i}.conv', oc, cfg.image_channels, 1)
    return sh


def discriminator_state_shapes(cfg):
    dblocks, oc, resl = cfg.discriminator_channels()
    sh = {}

    def conv(p, ic, o, k):
        sh[p + '.layer.weight'] = (o, ic, k, k)
        sh[p + '.layer.bias'] = (o,)

    conv('from_rgb.0', cfg.image_channels, cfg.channels, 1)
    for i, (ic, o) in enumerate(dblocks):
        for j in range(cfg.block_num_conv):
            conv(f'blocks.{i}.block.{2 * j}', ic if j == 0 else o, o, 3)
        conv(f'blocks.{i}.skip', ic, o, 1)
    n = len(dblocks)
    conv(f'blocks.{n + 1}', oc + 1, oc, 3)
    sh[f'blocks.{n + 4}.layer.weight'] = (oc, oc * resl * resl)
    sh[f'blocks.{n + 4}.layer.bias'] = (oc,)
    sh[f'blocks.{n + 6}.layer.weight'] = (1, oc)
    sh[f'blocks.{n + 6}.layer.bias'] = (1,)
    return sh


def init_state(shapes, cfg, generator=None, which='G'):
    """init_weight_N01 as applied by ``main`` (utils.py:196-201): mapping weights ~ N(0, 1/map_lr),
    every other Linear/Conv/ModulatedConv weight ~ N(0,1), their biases 0; const ~ N(0,1);
    affine Linear inside ModulatedConv2d is visited by ``.apply`` as an nn.Linear too;
    ``InjectNoise.scale`` zeros (model.py:84); blur kernel fixed (model.py:141-145)."""
    sd = {}
    for k, s in shapes.items():
        if k.endswith('.kernel'):
            sd[k] = torch.tensor([[[1., 2., 1.], [2., 4., 2.], [1., 2., 1.]]]) / 16.0
        elif k.endswith('.scale') or k.endswith('bias'):
            sd[k] = torch.zeros(s)
        else:
            std = 1.0 / cfg.map_lr if (which == 'G' and k.startswith('map.')) else 1.0
            sd[k] = torch.randn(s, generator=generator) * std
    return sd
