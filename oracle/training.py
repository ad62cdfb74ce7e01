"""Oracle: losses, regularisers, DiffAugment, EMA and one training iteration of the
reference's StyleGAN2 loop, stated functionally.  TEST INFRASTRUCTURE ONLY.

Follows ``implementations/StyleGAN2/utils.py:18-33,53-116,208-221``,
``nnutils/loss/gan.py:98-114``, ``nnutils/loss/penalty.py:11-26,85-101``,
``nnutils/training.py:23-40`` and ``thirdparty/diffaugment/DiffAugment.py:10-53``
in the ``--disable-amp`` configuration (no GradScaler: ``scaler is None`` branches).
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

from . import stylegan2 as sg2


# ---------------------------------------------------------------------------------------------
# losses  (nnutils/loss/gan.py:98-114)

def ns_d_loss(real_prob, fake_prob):
    return F.softplus(-real_prob).mean() + F.softplus(fake_prob).mean()


def ns_g_loss(fake_prob):
    return F.softplus(-fake_prob).mean()


def calc_grad(outputs, inputs):
    """penalty.py:11-26 with scaler=None."""
    ones = torch.ones(outputs.size(), device=outputs.device)
    return torch.autograd.grad(outputs=outputs, inputs=inputs, grad_outputs=ones,
                               create_graph=True, retain_graph=True, only_inputs=True)[0]


def r1_penalty(real, d_fn):
    """r1_regularizer.__call__ (penalty.py:85-101)."""
    real_loc = real.detach().clone().requires_grad_(True)
    d_real = d_fn(real_loc)
    g = calc_grad(d_real, real_loc)
    g = g.reshape(g.size(0), -1)
    return g.norm(2, dim=1).pow(2).mean() / 2.


def pl_penalty(styles, images, pl_mean, noise=None):
    """utils.py:18-29."""
    num_pixels = images.shape[2] * images.shape[3]
    if noise is None:
        noise = torch.randn(images.size(), device=images.device)
    noise = noise / np.sqrt(num_pixels)
    outputs = (images * noise).sum()
    g = calc_grad(outputs, styles)
    g = g.pow(2).sum(dim=1).sqrt()
    return (g - pl_mean).pow(2).mean()


def update_pl_mean(old, new, decay=0.99):
    """utils.py:31-33."""
    return decay * old + (1 - decay) * new


def lazy_adam_hparams(lr, betas, k, lam):
    """utils.py:208-218: rescale only when the regulariser is active."""
    if lam > 0:
        ratio = k / (k + 1)
        return lr * ratio, (betas[0] ** ratio, betas[1] ** ratio)
    return lr, betas


def update_ema(sd, sd_ema, decay=0.999, param_keys=None, copy_buffers=False):
    """nnutils/training.py:23-40 over flat dicts.  ``param_keys`` = names of nn.Parameters
    (buffers such as Blur2d.kernel are only copied when ``copy_buffers``)."""
    with torch.no_grad():
        for k in sd_ema:
            is_param = param_keys is None or k in param_keys
            if is_param:
                sd_ema[k].mul_(decay).add_(sd[k].detach(), alpha=1 - decay)
            elif copy_buffers:
                sd_ema[k].copy_(sd[k])


# ---------------------------------------------------------------------------------------------
# DiffAugment with explicit random draws (DiffAugment.py:23-53)

def diffaug_draws(x, policy, generator=None):
    """Draw the random numbers in the order the reference consumes them."""
    B, dev = x.size(0), x.device
    d = {}
    for p in policy.split(','):
        if p == 'color':
            d['brightness'] = torch.rand(B, 1, 1, 1, dtype=x.dtype, device=dev, generator=generator)
            d['saturation'] = torch.rand(B, 1, 1, 1, dtype=x.dtype, device=dev, generator=generator)
            d['contrast'] = torch.rand(B, 1, 1, 1, dtype=x.dtype, device=dev, generator=generator)
        elif p == 'translation':
            sx, sy = int(x.size(2) * 0.125 + 0.5), int(x.size(3) * 0.125 + 0.5)
            d['tx'] = torch.randint(-sx, sx + 1, size=[B, 1, 1], device=dev, generator=generator)
            d['ty'] = torch.randint(-sy, sy + 1, size=[B, 1, 1], device=dev, generator=generator)
    return d


def diffaugment(x, policy, draws):
    """DiffAugment(x, policy) for 'color' and/or 'translation' given the draws."""
    if not policy:
        return x
    for p in policy.split(','):
        if p == 'color':
            x = x + (draws['brightness'] - 0.5)                                             # :23-25
            m = x.mean(dim=1, keepdim=True)
            x = (x - m) * (draws['saturation'] * 2) + m                                     # :28-31
            m = x.mean(dim=[1, 2, 3], keepdim=True)
            x = (x - m) * (draws['contrast'] + 0.5) + m                                     # :34-37
        elif p == 'translation':                                                            # :40-53
            B, C, H, W = x.shape
            tx, ty = draws['tx'], draws['ty']                                               # "x" = dim 2 (rows)
            gb, gx, gy = torch.meshgrid(torch.arange(B, device=x.device), torch.arange(H, device=x.device),
                                        torch.arange(W, device=x.device), indexing='ij')
            gx = torch.clamp(gx + tx + 1, 0, H + 1)
            gy = torch.clamp(gy + ty + 1, 0, W + 1)
            xp = F.pad(x, [1, 1, 1, 1, 0, 0, 0, 0])
            x = xp.permute(0, 2, 3, 1).contiguous()[gb, gx, gy].permute(0, 3, 1, 2).contiguous()
        else:
            raise NotImplementedError(p)
    return x.contiguous()


# ---------------------------------------------------------------------------------------------
# one iteration of train() (utils.py:53-116), functional

class StepState:
    """Everything train() carries across iterations."""

    def __init__(self, g_cfg, G, G_ema, D, lr=0.001, betas=(0., 0.99), r1_lambda=10., pl_lambda=0.,
                 d_k=16, g_k=8, policy='color,translation'):
        self.cfg = g_cfg
        self.G, self.G_ema, self.D = G, G_ema, D
        for t in list(G.values()) + list(D.values()):
            t.requires_grad_(True)
        self.g_param_keys = [k for k in G if not k.endswith('.kernel')]
        g_lr, g_betas = lazy_adam_hparams(lr, betas, g_k, pl_lambda)
        d_lr, d_betas = lazy_adam_hparams(lr, betas, d_k, r1_lambda)
        self.opt_G = torch.optim.Adam([G[k] for k in self.g_param_keys], lr=g_lr, betas=g_betas)
        self.opt_D = torch.optim.Adam(list(D.values()), lr=d_lr, betas=d_betas)
        self.r1_lambda, self.pl_lambda, self.d_k, self.g_k, self.policy = r1_lambda, pl_lambda, d_k, g_k, policy
        self.pl_mean = 0.
        self.batches_done = 0


def train_iteration(st, real, sampler, rng=None):
    """One pass of the loop body (utils.py:55-116).  ``rng(kind, ...)`` supplies every random draw
    so a test can replay the same numbers through the product path:
      rng('aug', x)          -> DiffAugment draws dict
      rng('noise')           -> a stylegan2.NoiseSource for one G forward
      rng('pl', shape)       -> path-length noise
    ``sampler(size)`` is the latent sampler (utils.py:61,89), called once per step half so the
    RNG stream is consumed in the reference's order.  Returns dict(D_loss, G_loss)."""
    cfg = st.cfg
    if rng is None:
        def rng(kind, *a):
            if kind == 'aug':
                return diffaug_draws(a[0], st.policy)
            if kind == 'noise':
                return sg2.NoiseSource()
            if kind == 'pl':
                return torch.randn(a[0])
    st.opt_G.zero_grad()
    st.opt_D.zero_grad()
    D = lambda x: sg2.discriminator(st.D, cfg, x)
    G = lambda z: sg2.generator(st.G, cfg, z, noise=rng('noise'))

    # discriminator (utils.py:60-86)
    z_d = sampler((real.size(0), cfg.style_dim))
    real_aug = diffaugment(real, st.policy, rng('aug', real))
    real_prob = D(real_aug)
    fake, _ = G(z_d)
    fake_aug = diffaugment(fake, st.policy, rng('aug', fake))
    fake_prob = D(fake_aug.detach())
    if st.batches_done % st.d_k == 0 and st.r1_lambda > 0 and st.batches_done != 0:
        D_loss = r1_penalty(real, D) * st.r1_lambda * st.d_k                # replaces the GAN loss (F10)
    else:
        D_loss = ns_d_loss(real_prob, fake_prob)
    D_loss.backward()
    st.opt_D.step()

    # generator (utils.py:88-113)
    z_g = sampler((real.size(0), cfg.style_dim))
    fake, style = G(z_g)
    fake_aug = diffaugment(fake, st.policy, rng('aug', fake))
    fake_prob = D(fake_aug)
    if st.batches_done % st.g_k == 0 and st.pl_lambda > 0 and st.batches_done != 0:
        pl = pl_penalty(style, fake, st.pl_mean, rng('pl', fake.shape))
        G_loss = pl * st.pl_lambda * st.g_k
        st.pl_mean = update_pl_mean(st.pl_mean, float(pl.detach()))
    else:
        G_loss = ns_g_loss(fake_prob)
    G_loss.backward()
    st.opt_G.step()

    update_ema(st.G, st.G_ema, param_keys=set(st.g_param_keys))            # utils.py:115-116
    st.batches_done += 1
    return dict(D_loss=float(D_loss.detach()), G_loss=float(G_loss.detach()))
