"""Smoke test of the hot path (called by __graft_entry__.smoke()): one tiny G forward / D forward checked against
the CPU oracle, then one full training iteration (D-step, G-step, EMA) in bf16 on the HIP kernels."""
import functools

import torch


def run(dev):
    from oracle import stylegan2 as S          # checker only (test infrastructure)
    from . import model as M, utils as U
    from ...nnutils import sample_nnoise, update_ema
    cfgd = dict(image_size=32, image_channels=3, style_dim=64, channels=8, max_channels=64, block_num_conv=2,
                map_num_layers=2, map_lr=0.01, mbsd_groups=4)
    torch.manual_seed(0)
    mk = lambda: M.Generator(cfgd['image_size'], 3, cfgd['style_dim'], cfgd['channels'], cfgd['max_channels'], 2, 2, True, 0.01)
    G, G_ema = mk().to(dev), mk().to(dev)
    D = M.Discriminator(cfgd['image_size'], 3, cfgd['channels'], cfgd['max_channels'], 2, 4).to(dev)
    G.init_weight(functools.partial(M.init_weight_N01, lr=0.01), M.init_weight_N01)
    D.apply(M.init_weight_N01)
    update_ema(G, G_ema, decay=0)
    z = torch.randn(4, cfgd['style_dim'])
    draws = []
    orig = M.InjectNoise.draw

    def rec(x):
        n = orig(x)
        draws.append(n.detach().float().cpu())
        return n
    M.InjectNoise.draw = staticmethod(rec)
    try:
        with torch.no_grad():
            image, style = G(z.to(dev))
            logits = D(image)
    finally:
        M.InjectNoise.draw = orig
    cfg = S.Config(**cfgd)
    sdG = {k: v.detach().float().cpu() for k, v in G.state_dict().items()}
    sdD = {k: v.detach().float().cpu() for k, v in D.state_dict().items()}
    with torch.no_grad():
        ref_img, ref_style = S.generator(sdG, cfg, z, noise=S.NoiseSource(draws))
        ref_logits = S.discriminator(sdD, cfg, ref_img)
    err_i = (image.cpu() - ref_img).abs().max().item()
    err_l = ((logits.cpu() - ref_logits).abs().max() / ref_logits.abs().max().clamp_min(1e-6)).item()
    assert err_i < 0.06, f'generator image differs from the oracle by {err_i}'
    assert err_l < 0.08, f'discriminator logits differ from the oracle by {err_l} (relative)'
    opt_G, opt_D = U.build_optimizers(G, D, 0.001, (0., 0.99), 10., 0., 16, 8)
    step = U.TrainStep(G, G_ema, D, opt_G, opt_D, 10., 0., 1, 8, 'color,translation', cfgd['style_dim'],
                       functools.partial(sample_nnoise, device=dev))
    real = torch.rand(4, 3, 32, 32, device=dev) * 2 - 1
    for _ in range(2):                                   # second iteration is an R1 (double-backward) iteration
        dl, gl, fake = step(real)
    torch.cuda.synchronize()
    assert torch.isfinite(dl) and torch.isfinite(gl) and torch.isfinite(fake).all()
    print(f'StyleGAN2 smoke: image err {err_i:.4f}, logits rel err {err_l:.4f}, D_loss {dl.item():.4f}, G_loss {gl.item():.4f}')
