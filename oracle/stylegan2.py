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
    return F.conv2d(x * coef, w, b, padding=padding)


def modulated_conv2d(sd, prefix, x, y, demod=True, gain=1.0):
    """ModulatedConv2d.forward (model.py:106-132)."""
    B, _, H, W = x.shape
    weight, bias = sd[prefix + '.weight'], sd[prefix + '.bias']
    cout, cin, k, _ = weight.shape
    s = elr_linear(sd, prefix + '.affine', y) + 1                                   # :110
    coef = gain / math.sqrt(weight[0].numel())                                      # :105
    w = weight[None] * s[:, None, :, None, None] * coef                             # :115
    if demod:
        w = w * torch.rsqrt(w.pow(2).sum([2, 3, 4], keepdim=True) + 1e-4)           # :118-120
    pad = (k - 1) // 2                                                              # :134-135 with stride 1
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


def synthesis(sd, cfg, x, styles, noise, prefix='synthesis'):
    """Synthesis.forward (model.py:312-332) with ``styles`` already a per-layer list."""
    _, blocks = cfg.synthesis_channels()
    x = modulated_conv2d(sd, f'{prefix}.input', x, styles[0])                                   # :324
    pre = upsample2x(modulated_conv2d(sd, f'{prefix}.input_to_image.conv', x, styles[0], demod=False))   # :325, ToImage :244-250
    image = pre
    for i, (resl, _ic, _oc) in enumerate(blocks):
        y = styles[i + 1]
        # StyleBlock (model.py:154-180): up, blur, [modconv, noise, lrelu] * num_conv
        x = blur2d(upsample2x(x))
        for j in range(cfg.block_num_conv):
            x = modulated_conv2d(sd, f'{prefix}.blocks.{i}.block.{2 + 3 * j}', x, y)
            x = x + noise(x.shape[0], x.shape[2], x.shape[3], x.device, x.dtype)              # InjectNoise: unscaled (F10)
            x = F.leaky_relu(x, 0.2)
        image = modulated_conv2d(sd, f'{prefix}.to_images.{i}.conv', x, y, demod=False) + pre  # ToImage :245-247
        if resl < cfg.image_size:
            image = upsample2x(image)                                                          # :248-249
        pre = image
    return torch.tanh(image)                                                                    # :332


def generator(sd, cfg, z, noise=None, injection=None):
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
    return synthesis(sd, cfg, x, styles, noise), style


def d_block(sd, prefix, x, num_conv):
    """DBlock.forward (model.py:204-212)."""
    t = x
    for j in range(num_conv):
        x = F.leaky_relu(elr_conv(sd, f'{prefix}.block.{2 * j}', x, 1), 0.2)
    t = elr_conv(sd, f'{prefix}.skip', t, 0)
    x = F.avg_pool2d(x, 2)
    t = F.avg_pool2d(t, 2)
    return (x + t) / np.sqrt(2)


def discriminator(sd, cfg, x):
    """Discriminator.forward (model.py:398-401) over the nn.Sequential of :375-397."""
    dblocks, oc, resl = cfg.discriminator_channels()
    x = F.leaky_relu(elr_conv(sd, 'from_rgb.0', x, 0), 0.2)
    for i in range(len(dblocks)):
        x = d_block(sd, f'blocks.{i}', x, cfg.block_num_conv)
    n = len(dblocks)
    x = minibatch_stddev(x, cfg.mbsd_groups)                                    # blocks.{n}
    x = F.leaky_relu(elr_conv(sd, f'blocks.{n + 1}', x, 1), 0.2)                # blocks.{n+1}, {n+2}
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
