"""GPU parity of the StyleGAN3 model and training loop on the HIP operators, against golden vectors produced by the
reference's own modules / train() on CPU (tools/make_golden.py, fixtures sg3_model / sg3_train).

fp32 compute mode: <= 1e-3 relative (north_star tolerance for fp32 activations).
bf16 compute mode (the training path): compared with the fp32 reference at bf16-level tolerance."""
import functools

import numpy as np
import pytest
import torch

from conftest import t

pytestmark = pytest.mark.gpu
DEV = 'cuda'
CFG = dict(image_size=32, latent_dim=16, num_layers=6, map_num_layers=2, channels=32, max_channels=16, style_dim=16,
           margin_size=4, d_channels=8, d_max_channels=16)


def sub(g, prefix):
    return {k[len(prefix):]: t(v).clone() for k, v in g.items() if k.startswith(prefix)}


def build(dtype):
    from animeface_amd.implementations.StyleGAN3 import model as M
    c = CFG
    G = M.Generator(c['image_size'], c['latent_dim'], c['num_layers'], c['map_num_layers'], c['channels'], c['max_channels'],
                    c['style_dim'], margin_size=c['margin_size'], compute_dtype=dtype)
    D = M.Discriminator(c['image_size'], 3, c['d_channels'], c['d_max_channels'], compute_dtype=dtype)
    return M, G.to(DEV), D.to(DEV)


def relerr(a, b):
    return ((a.detach().float().cpu() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-8)).item()


def test_layer_params_and_filters_match_the_reference(golden):
    from animeface_amd.implementations.StyleGAN3 import model as M
    g = golden('sg3_model')
    for tag, args in [('lp', (32, 6, 2 ** 11 * 0.5, 16, 3, 4)), ('lp256', (256, 14, 2 ** 14 * 0.5, 512, 3, 10))]:
        for name, val in zip(['channels', 'sizes', 'rates', 'cutoffs', 'half_widths'], M.get_layer_params(*args)):
            np.testing.assert_allclose(val, g[f'{tag}_{name}'], rtol=1e-12)
    # the design filters are buffers: a freshly built model must carry the reference's values
    _, G, D = build(torch.float32)
    ref = sub(g, 'G/')
    sd = G.state_dict()
    assert set(sd) == set(ref)
    for k in sd:
        if k.endswith('up_filter') or k.endswith('down_filter'):
            torch.testing.assert_close(sd[k].cpu(), ref[k], rtol=1e-6, atol=1e-8)
    refd = sub(g, 'D/')
    assert set(D.state_dict()) == set(refd)
    for k, v in D.state_dict().items():
        assert tuple(v.shape) == tuple(refd[k].shape), k
        if k.endswith('down_filter'):
            torch.testing.assert_close(v.cpu(), refd[k])


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 1e-3), (torch.bfloat16, 8e-2)])
def test_generator_discriminator_forward_and_grads_vs_reference(golden, dtype, tol):
    g = golden('sg3_model')
    M, G, D = build(dtype)
    G.load_state_dict(sub(g, 'G/'), strict=True)
    D.load_state_dict(sub(g, 'D/'), strict=True)
    z = t(g['z']).to(DEV)
    G.train()
    image = G(z)
    assert image.dtype == torch.float32 and tuple(image.shape) == (4, 3, 32, 32)
    assert relerr(image, t(g['image'])) < tol
    sd = G.state_dict()
    for k, v in sub(g, 'G1/').items():                      # ema / w_avg buffers moved exactly like the reference's
        torch.testing.assert_close(sd[k].cpu(), v, rtol=tol, atol=tol * 1e-2, msg=lambda m, k=k: f'{k}: {m}')
    if dtype == torch.float32:
        logits = D(image)
    else:
        logits = D(t(g['image']).to(DEV))                    # isolate D from G's bf16 rounding
    assert relerr(logits, t(g['logits'])) < tol
    if dtype != torch.float32:
        return
    loss = torch.nn.functional.softplus(-logits).mean()
    assert abs(loss.item() - float(g['g_loss'])) < tol * max(1.0, abs(float(g['g_loss'])))
    pg, pd = dict(G.named_parameters()), dict(D.named_parameters())
    gn = [k[len('gradG/'):] for k in g if k.startswith('gradG/')]
    dn = [k[len('gradD/'):] for k in g if k.startswith('gradD/')]
    grads = torch.autograd.grad(loss, [pg[k] for k in gn] + [pd[k] for k in dn])
    for k, gr in zip(gn, grads[:len(gn)]):
        assert relerr(gr, t(g['gradG/' + k])) < tol, k
    for k, gr in zip(dn, grads[len(gn):]):
        assert relerr(gr, t(g['gradD/' + k])) < tol, k
    G.eval()
    with torch.no_grad():
        assert relerr(G(z, truncation_psi=0.7), t(g['image_eval_psi07'])) < tol


def test_bf16_backward_tracks_the_fp32_reference(golden):
    g = golden('sg3_model')
    M, G, D = build(torch.bfloat16)
    G.load_state_dict(sub(g, 'G/'))
    D.load_state_dict(sub(g, 'D/'))
    G.train()
    logits = D(G(t(g['z']).to(DEV)))
    loss = torch.nn.functional.softplus(-logits).mean()
    assert abs(loss.item() - float(g['g_loss'])) < 5e-2
    pg, pd = dict(G.named_parameters()), dict(D.named_parameters())
    gn = [k[len('gradG/'):] for k in g if k.startswith('gradG/')]
    dn = [k[len('gradD/'):] for k in g if k.startswith('gradD/')]
    grads = torch.autograd.grad(loss, [pg[k] for k in gn] + [pd[k] for k in dn])
    for k, gr in zip(gn + dn, grads):
        ref = t(g[('gradG/' if k in gn else 'gradD/') + k]).float()
        cos = torch.nn.functional.cosine_similarity(gr.float().cpu().flatten(), ref.flatten(), dim=0).item()
        assert cos > 0.97, (k, cos)


@pytest.mark.parametrize('dtype,tol,gtol', [(torch.float32, 1e-3, 1e-3), (torch.bfloat16, 0.1, 0.3)])
def test_r1_double_backward_vs_reference(golden, dtype, tol, gtol):
    from animeface_amd.nnutils.loss import r1_regularizer
    g = golden('sg3_model')
    M, G, D = build(dtype)
    D.load_state_dict(sub(g, 'D/'))
    r1 = r1_regularizer()(t(g['real']).to(DEV), D, None)
    assert abs(r1.item() - float(g['r1'])) < tol * abs(float(g['r1']))
    r1.backward()
    pd = dict(D.named_parameters())
    # (bf16: 2-13 % on the weights; gradients 100x below the others -- one bias at 3e-7 -- are rounding noise and only bounded absolutely)
    top = max(float(t(g[k]).abs().max()) for k in g if k.startswith('r1grad/'))
    for k in g:
        if k.startswith('r1grad/'):
            got, ref = pd[k[len('r1grad/'):]].grad, t(g[k])
            if dtype == torch.float32 or float(ref.abs().max()) > 0.02 * top:
                assert relerr(got, ref) < gtol, k
            else:
                assert float((got.float().cpu() - ref).abs().max()) < 0.02 * top, k


def test_train_loop_replays_the_references_train(golden):
    """Three iterations of the reference's own StyleGAN3 train() (gp_every = 2: iterations 0 and 2 add R1) replayed
    through the HIP path in fp32 with the same random stream."""
    from animeface_amd.implementations.StyleGAN3 import utils as U
    from animeface_amd.nnutils import sample_nnoise, update_ema, freeze
    from animeface_amd.thirdparty.diffaugment import DiffAugment
    from animeface_amd import rng
    g = golden('sg3_train')
    M, G, D = build(torch.float32)
    _, G_ema, _ = build(torch.float32)
    G.load_state_dict(sub(g, 'G0/'))
    D.load_state_dict(sub(g, 'D0/'))
    freeze(G_ema)
    update_ema(G, G_ema, 0., copy_buffers=True)
    lr, map_lr_scale, b0, b1, gp_lambda, gp_every = [float(v) for v in g['train_hparams']]
    opt_G, opt_D = U.build_optimizers(G, D, lr, map_lr_scale, (b0, b1))
    assert opt_G.param_groups[1]['lr'] == lr * map_lr_scale and opt_G.param_groups[0]['lr'] == lr
    losses = []
    with rng.cpu_stream():
        torch.manual_seed(26)
        const_input = sample_nnoise((2, CFG['latent_dim']), DEV)
        step = U.TrainStep(G, G_ema, D, opt_G, opt_D, gp_lambda, int(gp_every),
                           functools.partial(DiffAugment, policy='color,translation'), CFG['latent_dim'])
        for it in range(3):
            dl, gl, _ = step(t(g['train_real'][it]).to(DEV))
            losses.append([dl.item(), gl.item()])
    np.testing.assert_allclose(np.array(losses), g['train_losses'], rtol=5e-3, atol=1e-5)
    for name, net, prefix in [('G', G, 'G3/'), ('D', D, 'D3/'), ('G_ema', G_ema, 'Gema3/')]:
        sd = net.state_dict()
        ref = sub(g, prefix)
        assert set(sd) == set(ref)
        bad = 0
        total = 0
        for k, v in ref.items():
            d = (sd[k].cpu().float() - v.float()).abs()
            # Adam with beta1 = 0 moves every weight by ~lr * sign(grad): an element whose gradient is at rounding level
            # may legitimately take the other sign, so count outliers instead of demanding every element
            bad += int((d > 5e-5 + 5e-3 * v.float().abs()).sum())
            total += d.numel()
        assert bad <= 0.002 * total, (name, bad, total)


def test_bf16_training_steps_run_and_stay_finite():
    from animeface_amd.implementations.StyleGAN3 import utils as U
    from animeface_amd.nnutils import update_ema, freeze
    from animeface_amd.thirdparty.diffaugment import DiffAugment
    torch.manual_seed(0)
    M, G, D = build(torch.bfloat16)
    _, G_ema, _ = build(torch.bfloat16)
    freeze(G_ema)
    update_ema(G, G_ema, 0., copy_buffers=True)
    opt_G, opt_D = U.build_optimizers(G, D, 0.0025, 0.01, (0., 0.99))
    real = torch.rand(8, 3, 32, 32, device=DEV) * 2 - 1
    hist = U.train(3, [real], CFG['latent_dim'], torch.randn(2, CFG['latent_dim'], device=DEV), G, G_ema, D, opt_G, opt_D,
                   3., 2, functools.partial(DiffAugment, policy='color,translation'), torch.device(DEV), True, log_every=1)
    assert len(hist) == 3 and all(np.isfinite(h[1]) and np.isfinite(h[2]) for h in hist)
    for p in list(G.parameters()) + list(D.parameters()) + list(G_ema.parameters()):
        assert torch.isfinite(p).all()


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
def test_graph_replayed_step_equals_the_eager_step(dtype):
    """StyleGAN3's GraphedTrainStep (one HIP graph per iteration kind: adversarial loss only / with the R1 penalty of every gp_every-th
    iteration) against the eager TrainStep from the same seeds: same kernels in the same order with torch's graph-safe random offsets, so
    the losses agree to summation noise (fp32) / bf16 resolution, the buffers the step advances in place (w_avg, the layers' magnitude
    EMAs) move in both, and nothing non-finite appears.  Also: train(graphs=True) consumes no iteration for the recording."""
    from animeface_amd.implementations.StyleGAN3 import utils as U
    from animeface_amd.nnutils import update_ema, freeze
    from animeface_amd.thirdparty.diffaugment import DiffAugment

    def run(graphed, iters=7):
        torch.manual_seed(5)
        M, G, D = build(dtype)
        _, G_ema, _ = build(dtype)
        freeze(G_ema)
        update_ema(G, G_ema, 0., copy_buffers=True)
        oG, oD = U.build_optimizers(G, D, 0.0025, 0.01, (0., 0.99), capturable=True)
        step = U.TrainStep(G, G_ema, D, oG, oD, 3., 3, functools.partial(DiffAugment, policy='color,translation'), CFG['latent_dim'])
        gen = torch.Generator().manual_seed(9)
        real = (torch.rand(8, 3, 32, 32, generator=gen) * 2 - 1).to(DEV)
        torch.manual_seed(123)
        for _ in range(2):                                   # eager in both arms (iteration 0 carries the R1 penalty)
            step(real)
        runner = U.GraphedTrainStep(step, real, warmup=0) if graphed else step
        if graphed:
            runner.capture_all()
            assert runner.kinds() == {'gan', 'r1'}
        losses = []
        for _ in range(iters):                               # gp_every = 3: iterations 3 and 6 replay the R1 graph
            dl, gl, fake = runner(real)
            losses.append((float(dl), float(gl)))
            assert torch.isfinite(fake).all()
        assert step.batches_done == 2 + iters
        return losses, {k: v.detach().float().clone() for k, v in list(G.state_dict().items()) + [('D.' + k, v) for k, v in D.state_dict().items()]}

    le, we = run(False)
    lg, wg = run(True)
    tol = 2e-3 if dtype == torch.float32 else 5e-2
    for (d0, g0), (d1, g1) in zip(le, lg):
        assert d0 == pytest.approx(d1, rel=tol, abs=tol) and g0 == pytest.approx(g1, rel=tol, abs=tol), (le, lg)
    far = total = 0
    for k in we:
        assert torch.isfinite(wg[k]).all(), k
        d = (we[k] - wg[k]).abs()
        far += int((d > (1e-3 if dtype == torch.float32 else 2e-2)).sum())
        total += d.numel()
    # (Adam with beta1 = 0 moves a weight whose gradient changes sign near zero by a few lr: the bound is on how many differ)
    assert far <= (0.002 if dtype == torch.float32 else 0.05) * total, (far, total)
    assert not torch.equal(we['map.w_avg'], torch.zeros_like(we['map.w_avg'])) and torch.allclose(we['map.w_avg'], wg['map.w_avg'], atol=1e-3)

    # train(graphs=True): 2 eager iterations, then replays; every iteration is logged, none is consumed by the recording
    torch.manual_seed(5)
    M, G, D = build(dtype)
    _, G_ema, _ = build(dtype)
    freeze(G_ema)
    update_ema(G, G_ema, 0., copy_buffers=True)
    oG, oD = U.build_optimizers(G, D, 0.0025, 0.01, (0., 0.99), capturable=True)
    real = torch.rand(8, 3, 32, 32, device=DEV) * 2 - 1
    hist = U.train(6, [real], CFG['latent_dim'], torch.randn(2, CFG['latent_dim'], device=DEV), G, G_ema, D, oG, oD, 3., 3,
                   functools.partial(DiffAugment, policy='color,translation'), torch.device(DEV), dtype == torch.bfloat16, log_every=1, log=None, graphs=True)
    assert [h[0] for h in hist] == list(range(6)) and all(np.isfinite(h[1]) and np.isfinite(h[2]) for h in hist)


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-4), (torch.bfloat16, 3e-2)], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('demod', [True, False])
def test_modulated_conv_with_folded_scales_matches_the_operand_scaled_path(dtype, tol, demod, monkeypatch):
    """model.FOLD_SCALES: the modulated conv's style scale rides in the planar -> channels-last conversion of its input, the demodulation
    scale of the output gradient in the same conversion of dy (agf_planar_to_cl_pad_scaled), and the MFMA launches run unscaled --
    against the composite with the scales inside the launches: output and all four gradients (x, weight, style, and through the
    demodulation the weight again), ragged channel counts (padded to a 16-byte vector) included."""
    from animeface_amd.implementations.StyleGAN3 import model as M
    torch.manual_seed(2)
    conv = M.ModulatedConv(20, 44, 3, 2, demod=demod).to(DEV)
    x0 = torch.randn(3, 20, 18, 22, device=DEV).to(dtype)
    s0 = torch.randn(3, 20, device=DEV) * 0.5 + 1.0
    gy = torch.randn(3, 44, 20, 24, device=DEV).to(dtype)
    gain = torch.tensor(0.7, device=DEV)
    outs = []
    for on in (True, False):
        monkeypatch.setattr(M, 'FOLD_SCALES', on)
        x, s = x0.clone().requires_grad_(True), s0.clone().requires_grad_(True)
        y = conv(x, s, gain)
        assert y.shape == gy.shape
        outs.append((y,) + torch.autograd.grad(y, [x, s, conv.weight], gy))
    for a, b in zip(outs[0], outs[1]):
        assert a.shape == b.shape and relerr(a, b.detach().float().cpu()) < tol, (a.shape, relerr(a, b.detach().float().cpu()))


@pytest.mark.gpu
@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-4), (torch.bfloat16, 3e-2)], ids=['fp32', 'bf16'])
def test_generator_with_fused_layer_scalars_matches_the_torch_ops(dtype, tol, monkeypatch):
    """model.FUSED_SCALARS: one GEMM for the style affines of every layer (parameters packed into one buffer), ``agf_style_demod_fwd_ex`` /
    ``_bwd_ex`` for s, s * gain, d and their gradients, ``agf_ema_gain`` for the input-magnitude EMA, conv weights from the prepared-weight
    cache (``agf_prep_weights_pad``: ragged channel counts) -- against the same generator on the per-layer torch ops (FUSED_SCALARS = False):
    image, every layer's EMA buffer after the pass, and the gradient of every parameter; in eval mode too (the EMA stays put)."""
    from animeface_amd.implementations.StyleGAN3 import model as M
    from animeface_amd.implementations.StyleGAN2.conv import cached_weights
    torch.manual_seed(7)
    kw = dict(image_size=64, latent_dim=24, num_layers=8, map_num_layers=2, channels=20, max_channels=28, style_dim=24, margin_size=6)
    G = M.Generator(kw['image_size'], kw['latent_dim'], kw['num_layers'], kw['map_num_layers'], kw['channels'], kw['max_channels'],
                    kw['style_dim'], margin_size=kw['margin_size'], compute_dtype=dtype).to(DEV)
    with torch.no_grad():
        for n, p in G.named_parameters():
            if n.endswith('bias') and 'affine' not in n:
                p.normal_(0, 0.2)
    assert any(m.conv.weight.shape[0] % 8 or m.conv.weight.shape[1] % 8 for m in G.synthesis.net), 'the configuration has ragged channel counts'
    sd0 = {k: v.detach().clone() for k, v in G.state_dict().items()}
    z = torch.randn(3, 24, device=DEV)
    gy = torch.randn(3, 3, 64, 64, device=DEV)
    res = {}
    for on in (True, False):
        monkeypatch.setattr(M, 'FUSED_SCALARS', on)
        G.load_state_dict(sd0)
        G.train()
        params = [p for p in G.parameters()]
        with cached_weights():
            img = G(z)
            grads = torch.autograd.grad(img, params, gy, allow_unused=True)
        emas = [m.ema.detach().clone() for m in G.synthesis.net]
        G.eval()
        with torch.no_grad():
            img_eval = G(z)
        assert all(torch.equal(m.ema, e) for m, e in zip(G.synthesis.net, emas)), 'eval mode leaves the EMA alone'
        res[on] = (img.detach(), grads, emas, img_eval)
    if True:
        assert getattr(G.synthesis, '_pack', None) is not None, 'the affine parameters were packed'
        assert set(G.state_dict().keys()) == set(sd0.keys())
    assert relerr(res[True][0], res[False][0].float().cpu()) < tol
    assert relerr(res[True][3], res[False][3].float().cpu()) < tol
    for a, b in zip(res[True][2], res[False][2]):
        torch.testing.assert_close(a, b, rtol=2e-5 if dtype == torch.float32 else 2e-3, atol=1e-7)
    names = [n for n, _ in G.named_parameters()]
    worst = 0.0
    for n, a, b in zip(names, res[True][1], res[False][1]):
        assert (a is None) == (b is None), n
        if a is None:
            continue
        r = relerr(a, b.detach().float().cpu())
        worst = max(worst, r)
        assert r < (2e-3 if dtype == torch.float32 else 6e-2), (n, r)
    print(f'fused layer scalars ({dtype}): worst parameter-gradient deviation {worst:.2e} of the largest value')


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('shape,pad', [((20, 12, 3), (24, 16)), ((33, 65, 3), (40, 72)), ((3, 44, 1), (8, 48)), ((16, 16, 3), (16, 16))])
def test_prep_weights_pad_matches_pad_then_prep(dtype, shape, pad):
    """agf_prep_weights_pad (zero-padded operand layouts in one launch, singly and through PrepPlan's descriptor table) against F.pad + agf_prep_weights."""
    import torch.nn.functional as F
    from animeface_amd.implementations.StyleGAN2 import conv as C
    torch.manual_seed(3)
    cout, cin, k = shape
    w = torch.nn.Parameter(torch.randn(cout, cin, k, k, device=DEV))
    wp = F.pad(w.detach(), [0, 0, 0, 0, 0, pad[1] - cin, 0, pad[0] - cout])
    ref_q, ref_ft = C.prep_weights_raw(wp, 0.37, dtype, True, True)
    got_q, got_ft = C.prep_weights_raw(w, 0.37, dtype, True, True, pad=pad)
    assert got_q.shape == ref_q.shape and got_ft.shape == ref_ft.shape
    assert torch.equal(got_q, ref_q) and torch.equal(got_ft, ref_ft)
    plan = C.PrepPlan([w])
    with C.cached_weights(), C.recording_plans(plan):
        C.prepared_weights(w, 0.37, dtype, need_ft=True, pad=pad)
    plan.build()
    with C.cached_weights():
        plan.run()
        ent = C.prepared_weights(w, 0.37, dtype, need_ft=True, pad=pad)
        assert plan.entries and ent.wq is plan.entries[0][2], 'served from the plan'
        assert torch.equal(ent.wq, ref_q) and torch.equal(ent.wq_ft, ref_ft)


@pytest.mark.gpu
def test_resblock_sum_inside_the_skip_conv_is_as_close_to_fp32_as_the_separate_add(monkeypatch):
    """model.RES_FUSED: conv2(conv1(x)) + skip(x) with the sum formed on the skip conv's fp32 accumulators (residual operand; the branch gain in
    the weight coefficient) against the separate bf16 add -- both compared with the SAME discriminator run in fp32: block outputs, logits and
    the input gradient.  The fused form rounds the sum once instead of twice but rounds its weights at another place (W * scale * gain);
    measured (MI355X, this seed): block outputs 4.8e-3 / 4.4e-3, logits 4.0e-3 / 1.0e-2, input gradient 4.6e-2 / 4.0e-2 of the mean magnitude
    (fused / separate) -- the same accuracy class, neither systematically closer.  Gate: no metric more than 1.25x the separate add's."""
    from animeface_amd.implementations.StyleGAN3 import model as M
    torch.manual_seed(11)
    D16 = M.Discriminator(64, 3, 16, 64, compute_dtype=torch.bfloat16).to(DEV)
    D32 = M.Discriminator(64, 3, 16, 64, compute_dtype=torch.float32).to(DEV)
    D32.load_state_dict(D16.state_dict())
    x = (torch.rand(4, 3, 64, 64, device=DEV) * 2 - 1)

    def run(D, xin):
        taps = []
        hooks = [m.register_forward_hook(lambda mod, i, o: taps.append(o.detach().float())) for m in D.resblocks]
        xin = xin.clone().requires_grad_(True)
        out = D(xin)
        g, = torch.autograd.grad(out.float().sum(), xin)
        for h in hooks:
            h.remove()
        return taps, out.detach().float(), g.float()
    ref = run(D32, x)
    errs = {}
    for on in (True, False):
        monkeypatch.setattr(M, 'RES_FUSED', on)
        got = run(D16, x)
        e_blocks = [float((a - b).abs().mean() / b.abs().mean()) for a, b in zip(got[0], ref[0])]
        errs[on] = (sum(e_blocks) / len(e_blocks), float((got[1] - ref[1]).abs().mean() / ref[1].abs().mean()),
                    float((got[2] - ref[2]).abs().mean() / ref[2].abs().mean()), got)
    print(f'ResBlock sum in the skip conv: mean block error vs fp32 {errs[True][0]:.5f} (fused) / {errs[False][0]:.5f} (separate add); '
          f'logits {errs[True][1]:.5f} / {errs[False][1]:.5f}; input gradient {errs[True][2]:.5f} / {errs[False][2]:.5f}')
    for i, what in enumerate(('block outputs', 'logits', 'input gradient')):
        assert errs[True][i] <= errs[False][i] * 1.25 + 1e-3, what
    for a, b in zip(errs[True][3][0], errs[False][3][0]):
        assert relerr(a, b.cpu()) < 2e-2


@pytest.mark.gpu
def test_resblock_input_gradient_summed_inside_the_fir_pass(monkeypatch):
    """model.RES_GRAD_LINK: the gradient of a residual block's input = conv1's data gradient + the adjoint of the skip branch's decimation, with
    the sum formed by the FIR pass (agf_upfirdn2d_add) instead of autograd's bf16 add -- against the unlinked run of the same bf16
    discriminator (one rounding less: within bf16 noise), for the input and every parameter; and a double-backward (R1-style) pass, which must
    not use the link at all: the same launches with and without."""
    from animeface_amd.implementations.StyleGAN3 import model as M
    torch.manual_seed(3)
    D = M.Discriminator(64, 3, 16, 64, compute_dtype=torch.bfloat16).to(DEV)
    x = (torch.rand(6, 3, 64, 64, device=DEV) * 2 - 1)
    params = [p for p in D.parameters()]
    first, second = {}, {}
    for on in (True, False):
        monkeypatch.setattr(M, 'RES_GRAD_LINK', on)
        xin = x.clone().requires_grad_(True)
        out = D(xin)
        first[on] = [g.float() for g in torch.autograd.grad(out.float().square().sum(), [xin] + params)]
        xin = x.clone().requires_grad_(True)
        g, = torch.autograd.grad(D(xin).float().sum(), xin, create_graph=True)
        second[on] = [t.float() for t in torch.autograd.grad(g.square().sum(), params, allow_unused=True) if t is not None]
    for a, b in zip(first[True], first[False]):
        assert float((a - b).abs().mean()) <= 1.5e-2 * float(b.abs().mean()) + 1e-6
    assert len(second[True]) == len(second[False]) > 0
    for a, b in zip(second[True], second[False]):        # (the same launches; the weight gradients' fp32 atomics are not bit-reproducible run to run)
        assert float((a - b).abs().mean()) <= 1e-3 * float(b.abs().mean()) + 1e-9


def test_hip_model_vs_cpu_oracle_on_a_second_configuration():
    """A configuration / seed the fixtures do not contain: HIP fp32 networks against the CPU oracle (oracle/stylegan3.py, itself
    pinned to the reference by tests/test_oracle_sg3.py) on the same state_dict."""
    from animeface_amd.implementations.StyleGAN3 import model as M
    from oracle import stylegan3 as S3
    torch.manual_seed(5)
    kw = dict(image_size=64, latent_dim=24, num_layers=8, map_num_layers=2, channels=16, max_channels=24, style_dim=24, margin_size=6)
    G = M.Generator(kw['image_size'], kw['latent_dim'], kw['num_layers'], kw['map_num_layers'], kw['channels'], kw['max_channels'],
                    kw['style_dim'], margin_size=kw['margin_size'], compute_dtype=torch.float32)
    D = M.Discriminator(64, 3, 8, 24, compute_dtype=torch.float32)
    with torch.no_grad():
        for n, p in list(G.named_parameters()) + list(D.named_parameters()):
            if n.endswith('bias') and 'affine' not in n:
                p.normal_(0, 0.2)
    sdG = {k: v.clone() for k, v in G.state_dict().items()}
    sdD = {k: v.clone() for k, v in D.state_dict().items()}
    cfg = S3.Config(d_channels=8, d_max_channels=24, **kw)
    z = torch.randn(3, 24)
    ref_img, stats = S3.generator(sdG, cfg, z, training=True)
    ref_logits = S3.discriminator(sdD, cfg, ref_img)
    G, D = G.to(DEV).train(), D.to(DEV)
    img = G(z.to(DEV))
    assert relerr(img, ref_img) < 1e-3
    assert relerr(D(img), ref_logits) < 1e-3
    torch.testing.assert_close(G.map.w_avg.cpu(), stats['w_avg'], rtol=1e-4, atol=1e-6)
    for i, e in stats['ema'].items():
        torch.testing.assert_close(G.synthesis.net[i].ema.cpu(), e, rtol=1e-3, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('shape,pad', [((3, 16, 9, 13), 1), ((2, 64, 32, 32), 1), ((1, 8, 5, 7), 2)])
def test_channels_last_zero_pad_and_crop_one_pass(dtype, shape, pad):
    """``agf_cl_pad`` behind ``_ZeroPadCL`` / ``_CropCL`` (the discriminator's FIR + stride-2 conv pads its input by one more pixel):
    equal to F.pad / slicing, and each other's adjoint."""
    import torch.nn.functional as F
    from animeface_amd.implementations.StyleGAN3.model import _ZeroPadCL, _CropCL
    g = torch.Generator().manual_seed(3)
    x = torch.randn(shape, generator=g).to(dtype).to('cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = _ZeroPadCL.apply(x, pad)
    assert y.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(y, F.pad(x.detach(), [pad] * 4))
    gy = torch.randn(y.shape, generator=g).to(dtype).to('cuda').contiguous(memory_format=torch.channels_last)
    (gx,) = torch.autograd.grad(y, x, gy)
    assert torch.equal(gx, gy[:, :, pad:-pad, pad:-pad])
    assert torch.equal(_CropCL.apply(y, pad), x.detach())


@pytest.mark.parametrize('config', ['fixture', 'second'])
def test_bf16_networks_layer_by_layer_vs_the_bf16_emulating_oracle(golden, config):
    """The bf16 (training) path held tightly: the oracle evaluated with bf16 storage emulation (``oracle.stylegan3.bf16_storage``: conv
    operands x * s * gain and W * scale, every tensor a kernel stores) instead of the fp32 reference -- every synthesis layer and every
    residual block within 2e-2 of the largest value (the fp32 comparison above needs 8e-2), the image and the logits likewise, and the
    gradients of the generator loss by relative rms error instead of a cosine."""
    from animeface_amd.implementations.StyleGAN3 import model as M
    from oracle import stylegan3 as S3
    if config == 'fixture':
        g = golden('sg3_model')
        _, G, D = build(torch.bfloat16)
        G.load_state_dict(sub(g, 'G/'), strict=True)
        D.load_state_dict(sub(g, 'D/'), strict=True)
        cfg = S3.Config(**CFG)
        z = t(g['z'])
    else:
        torch.manual_seed(5)
        kw = dict(image_size=64, latent_dim=24, num_layers=8, map_num_layers=2, channels=16, max_channels=24, style_dim=24, margin_size=6)
        G = M.Generator(kw['image_size'], kw['latent_dim'], kw['num_layers'], kw['map_num_layers'], kw['channels'], kw['max_channels'],
                        kw['style_dim'], margin_size=kw['margin_size'], compute_dtype=torch.bfloat16).to(DEV)
        D = M.Discriminator(64, 3, 8, 24, compute_dtype=torch.bfloat16).to(DEV)
        with torch.no_grad():
            for n, p in list(G.named_parameters()) + list(D.named_parameters()):
                if n.endswith('bias') and 'affine' not in n:
                    p.normal_(0, 0.2)
        cfg = S3.Config(d_channels=8, d_max_channels=24, **kw)
        z = torch.randn(3, 24)
    sdG = {k: v.detach().float().cpu().clone() for k, v in G.state_dict().items()}
    sdD = {k: v.detach().float().cpu().clone() for k, v in D.state_dict().items()}
    for sd in (sdG, sdD):
        for k, v in sd.items():
            if v.is_floating_point() and not k.endswith(('filter', 'ema', 'w_avg', 'freqs', 'phases', 'transform', 'output_scale')):
                v.requires_grad_(True)
    G.train()
    taps_g, taps_d = [], []
    hooks = [m.register_forward_hook(lambda mod, i, o: taps_g.append(o.detach())) for m in G.synthesis.net]
    hooks += [m.register_forward_hook(lambda mod, i, o: taps_d.append(o.detach())) for m in D.resblocks]
    image = G(z.to(DEV))
    logits = D(image)
    loss = torch.nn.functional.softplus(-logits).mean()
    names_g = [k for k in ('synthesis.net.0.conv.weight', 'synthesis.net.2.conv.weight', 'synthesis.net.3.affine.weight', 'synthesis.net.4.bias',
                           'synthesis.input.weight', 'map.net.1.weight') if k in dict(G.named_parameters())]
    names_d = [k for k in ('from_rgb.weight', 'resblocks.0.conv1.weight', 'resblocks.0.conv2.weight', 'resblocks.1.skip.weight',
                           'resblocks.1.conv2.bias', 'epilogue.epilogue.3.weight') if k in dict(D.named_parameters())]
    pg, pd = dict(G.named_parameters()), dict(D.named_parameters())
    grads = torch.autograd.grad(loss, [pg[k] for k in names_g] + [pd[k] for k in names_d])
    for h in hooks:
        h.remove()
    ref_g, ref_d = [], []
    with S3.bf16_storage():
        ref_img, _ = S3.generator(sdG, cfg, z, training=True, collect=ref_g)
        ref_logits = S3.discriminator(sdD, cfg, ref_img, collect=ref_d)
        ref_loss = torch.nn.functional.softplus(-ref_logits).mean()
    ref_grads = torch.autograd.grad(ref_loss, [sdG[k] for k in names_g] + [sdD[k] for k in names_d])
    assert len(taps_g) == len(ref_g) and len(taps_d) == len(ref_d)
    for i, (a, b) in enumerate(zip(taps_g, ref_g)):
        r = relerr(a, b)
        print(f'   G layer {i} {tuple(a.shape)}: max rel {r:.4f}')
        assert r < 2e-2, ('G layer', i, r)
    for i, (a, b) in enumerate(zip(taps_d, ref_d)):
        r = relerr(a, b)
        print(f'   D block {i} {tuple(a.shape)}: max rel {r:.4f}')
        assert r < 2e-2, ('D block', i, r)
    assert relerr(image, ref_img.detach()) < 2e-2
    assert relerr(logits, ref_logits.detach()) < 2e-2
    for k, a, b in zip(names_g + names_d, grads, ref_grads):
        af, bf = a.detach().float().cpu(), b.detach().float()
        rms = float((af - bf).square().mean().sqrt() / bf.square().mean().sqrt().clamp_min(1e-12))
        print(f'   grad {k}: rms rel {rms:.4f}')
        assert rms < 0.08, (k, rms)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize('shape', [(2, 3, 5, 7), (16, 512, 36, 36), (4, 72, 148, 148), (1, 1, 1, 1), (3, 1, 1, 2731)])
def test_mean_square_kernel_vs_vector_norm(dtype, shape, monkeypatch):
    """``agf_sum_squares`` (the StyleGAN3 layer's input statistic, reference model.py:174-176) against x.float().square().mean() in fp64, incl. sizes
    that are not a multiple of the vector width, and the channels-last order."""
    from animeface_amd.implementations.StyleGAN3 import model as M
    torch.manual_seed(0)
    x = (torch.randn(shape, device=DEV) * 3).to(dtype)
    want = x.double().square().mean()
    got = M.mean_square(x)
    assert got.dtype == torch.float32 and got.dim() == 0
    assert abs(got.double() - want) / want < 2e-5
    if x.dim() == 4 and shape[1] > 1:
        got_cl = M.mean_square(x.contiguous(memory_format=torch.channels_last))
        assert abs(got_cl.double() - want) / want < 2e-5
    monkeypatch.setattr(M, 'SUM_SQUARES_KERNEL', False)
    ref = M.mean_square(x)
    assert abs(ref.double() - want) / want < 2e-5
