"""DiffAugment (Zhao et al., arXiv:2006.10738) with the reference's call surface
(reference thirdparty/diffaugment/DiffAugment.py:10-53): ``DiffAugment(x, policy, channels_first)``,
policies ``color``, ``translation``, ``cutout``; random draws are made in the reference's order.

The translation is a zero-padded integer shift per sample; it is evaluated as one gather along each
axis on the NCHW tensor instead of the reference's NHWC permute + advanced-index + permute round trip."""
import torch

from .. import rng


def DiffAugment(x, policy='', channels_first=True):
    if policy:
        if not channels_first:
            x = x.permute(0, 3, 1, 2)
        for p in policy.split(','):
            for f in AUGMENT_FNS[p]:
                x = f(x)
        if not channels_first:
            x = x.permute(0, 2, 3, 1)
        x = x.contiguous()
    return x


def rand_brightness(x):
    return x + (rng.rand((x.size(0), 1, 1, 1), dtype=x.dtype, device=x.device) - 0.5)


def rand_saturation(x):
    x_mean = x.mean(dim=1, keepdim=True)
    return (x - x_mean) * (rng.rand((x.size(0), 1, 1, 1), dtype=x.dtype, device=x.device) * 2) + x_mean


def rand_contrast(x):
    x_mean = x.mean(dim=[1, 2, 3], keepdim=True)
    return (x - x_mean) * (rng.rand((x.size(0), 1, 1, 1), dtype=x.dtype, device=x.device) + 0.5) + x_mean


def rand_translation(x, ratio=0.125):
    B, C, H, W = x.shape
    shift_x, shift_y = int(H * ratio + 0.5), int(W * ratio + 0.5)
    tx = rng.randint(-shift_x, shift_x + 1, size=[B, 1, 1], device=x.device)
    ty = rng.randint(-shift_y, shift_y + 1, size=[B, 1, 1], device=x.device)
    # out[b,:,i,j] = x[b,:,i+tx,j+ty] inside the image, else 0   (reference: pad 1, clamp index to the pad ring)
    rows = torch.arange(H, device=x.device).view(1, H, 1) + tx            # [B,H,1]
    cols = torch.arange(W, device=x.device).view(1, 1, W) + ty            # [B,1,W]
    valid = ((rows >= 0) & (rows < H)) & ((cols >= 0) & (cols < W))       # [B,H,W]
    rows = rows.clamp(0, H - 1).view(B, 1, H, 1).expand(B, C, H, W)
    cols = cols.clamp(0, W - 1).view(B, 1, 1, W).expand(B, C, H, W)
    out = x.gather(2, rows).gather(3, cols)
    return out * valid.unsqueeze(1).to(x.dtype)


def rand_cutout(x, ratio=0.5):
    B, _, H, W = x.shape
    cut_h, cut_w = int(H * ratio + 0.5), int(W * ratio + 0.5)
    ox = rng.randint(0, H + (1 - cut_h % 2), size=[B, 1, 1], device=x.device)
    oy = rng.randint(0, W + (1 - cut_w % 2), size=[B, 1, 1], device=x.device)
    rows = torch.arange(H, device=x.device).view(1, H, 1)
    cols = torch.arange(W, device=x.device).view(1, 1, W)
    r0 = (ox - cut_h // 2).clamp(0, H - 1)
    r1 = (ox - cut_h // 2 + cut_h - 1).clamp(0, H - 1)
    c0 = (oy - cut_w // 2).clamp(0, W - 1)
    c1 = (oy - cut_w // 2 + cut_w - 1).clamp(0, W - 1)
    hole = (rows >= r0) & (rows <= r1) & (cols >= c0) & (cols <= c1)
    return x * (~hole).unsqueeze(1).to(x.dtype)


AUGMENT_FNS = {
    'color': [rand_brightness, rand_saturation, rand_contrast],
    'translation': [rand_translation],
    'cutout': [rand_cutout],
}
