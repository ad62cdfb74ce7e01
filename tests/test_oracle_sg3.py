"""CPU: the StyleGAN3 oracle (oracle/stylegan3.py) against outputs of the reference's own modules (fixture sg3_model)."""
import numpy as np
import torch

from conftest import t
from oracle import stylegan3 as S3

CFG = dict(image_size=32, latent_dim=16, num_layers=6, map_num_layers=2, channels=32, max_channels=16, style_dim=16, margin_size=4,
           d_channels=8, d_max_channels=16)


def sub(g, prefix):
    return {k[len(prefix):]: t(v).clone() for k, v in g.items() if k.startswith(prefix)}


def test_layer_tables_match_the_reference(golden):
    g = golden('sg3_model')
    for tag, args in [('lp', (32, 6, 2 ** 11 * 0.5, 16, 3, 4)), ('lp256', (256, 14, 2 ** 14 * 0.5, 512, 3, 10))]:
        for name, val in zip(['channels', 'sizes', 'rates', 'cutoffs', 'half_widths'], S3.layer_config(*args)):
            np.testing.assert_allclose(val, g[f'{tag}_{name}'], rtol=1e-12)


def test_generator_and_discriminator_forward_and_gradients(golden):
    g = golden('sg3_model')
    cfg = S3.Config(**CFG)
    sdG = {k: v.requires_grad_(v.is_floating_point() and not k.endswith(('filter', 'ema', 'w_avg', 'transform', 'freqs', 'phases', 'output_scale')))
           for k, v in sub(g, 'G/').items()}
    sdD = {k: v.requires_grad_(not k.endswith('filter')) for k, v in sub(g, 'D/').items()}
    z = t(g['z'])
    image, stats = S3.generator(sdG, cfg, z, training=True)
    torch.testing.assert_close(image, t(g['image']), rtol=1e-4, atol=2e-5)
    for k, v in sub(g, 'G1/').items():                           # running statistics after one training-mode forward
        if k.endswith('w_avg'):
            torch.testing.assert_close(stats['w_avg'], v, rtol=1e-5, atol=1e-7)
        else:
            torch.testing.assert_close(stats['ema'][int(k.split('.')[2])], v, rtol=1e-5, atol=1e-7)
    logits = S3.discriminator(sdD, cfg, image)
    torch.testing.assert_close(logits, t(g['logits']), rtol=1e-4, atol=2e-5)
    loss = torch.nn.functional.softplus(-logits).mean()
    assert abs(loss.item() - float(g['g_loss'])) < 1e-5
    gn = [k[len('gradG/'):] for k in g if k.startswith('gradG/')]
    dn = [k[len('gradD/'):] for k in g if k.startswith('gradD/')]
    grads = torch.autograd.grad(loss, [sdG[k] for k in gn] + [sdD[k] for k in dn])
    for k, gr in zip(gn + dn, grads):
        ref = t(g[('gradG/' if k in gn else 'gradD/') + k])
        assert ((gr - ref).abs().max() / ref.abs().max().clamp_min(1e-8)).item() < 2e-4, k
    with torch.no_grad():
        image_eval, _ = S3.generator(sdG, cfg, z, truncation_psi=0.7, training=False)
    # the fixture's eval image was produced AFTER the training-mode forward had moved ema / w_avg
    sd2 = dict(sdG)
    for k, v in sub(g, 'G1/').items():
        sd2[k] = v
    with torch.no_grad():
        image_eval, _ = S3.generator(sd2, cfg, z, truncation_psi=0.7, training=False)
    torch.testing.assert_close(image_eval, t(g['image_eval_psi07']), rtol=1e-4, atol=2e-5)


def test_r1_penalty_on_the_discriminator(golden):
    g = golden('sg3_model')
    cfg = S3.Config(**CFG)
    sdD = {k: v.requires_grad_(not k.endswith('filter')) for k, v in sub(g, 'D/').items()}
    real = t(g['real']).requires_grad_(True)
    out = S3.discriminator(sdD, cfg, real)
    (gr,) = torch.autograd.grad(out.sum(), real, create_graph=True)
    r1 = gr.reshape(gr.shape[0], -1).norm(2, dim=1).pow(2).mean() / 2
    assert abs(r1.item() - float(g['r1'])) < 1e-4 * abs(float(g['r1']))
    names = [k[len('r1grad/'):] for k in g if k.startswith('r1grad/')]
    grads = torch.autograd.grad(r1, [sdD[k] for k in names], allow_unused=True)
    for k, gg in zip(names, grads):
        ref = t(g['r1grad/' + k])
        if gg is None:
            assert ref.abs().max() == 0, k
            continue
        assert ((gg - ref).abs().max() / ref.abs().max().clamp_min(1e-8)).item() < 5e-4, k
