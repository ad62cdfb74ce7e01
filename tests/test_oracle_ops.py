"""Pin the CPU oracle against golden vectors produced by the reference itself
(tools/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import t
from oracle import upfirdn2d as OU
from oracle import bias_act as OB
from oracle import filtered_lrelu as OF


def test_upfirdn2d_integer_matrix_bit_exact(golden):
    g = golden('upfirdn2d_int')
    x = t(g['x'])
    names = [str(n) for n in g['filter_names']]
    assert len(g['specs']) > 300
    for i, spec in enumerate(g['specs']):
        fi, upx, upy, dnx, dny, px0, px1, py0, py1, flip = [int(v) for v in spec]
        f = None if names[fi] == 'none' else t(g['f_' + names[fi]])
        y = OU.upfirdn2d(x, f, up=[upx, upy], down=[dnx, dny], padding=[px0, px1, py0, py1], flip_filter=bool(flip), gain=4)
        ref = t(g[f'y{i}'])
        assert y.shape == ref.shape, (i, spec)
        # integer inputs / filters: exact in fp32 regardless of summation order
        assert torch.equal(y, ref), (i, spec, (y - ref).abs().max())


def test_upfirdn2d_gather_statement_agrees(golden):
    g = golden('upfirdn2d_int')
    x = g['x']
    names = [str(n) for n in g['filter_names']]
    idx = list(range(0, len(g['specs']), 23))
    for i in idx:
        fi, upx, upy, dnx, dny, px0, px1, py0, py1, flip = [int(v) for v in g['specs'][i]]
        if names[fi] in ('none', 'sep12'):
            continue
        y = OU.upfirdn2d_numpy_gather(x[:1, :2], g['f_' + names[fi]], upx, upy, dnx, dny, px0, px1, py0, py1, bool(flip), 4.0)
        assert np.array_equal(y, g[f'y{i}'][:1, :2].astype(np.float64)), (i,)


@pytest.mark.parametrize('name', ['up2_f4', 'down2_f4', 'down2_f2', 'blur_f3', 'up2_f12', 'down2_f12', 'up4_f12_pad'])
def test_upfirdn2d_float_and_grads(golden, name):
    g = golden('upfirdn2d_float')
    f4, f3, f2, f12 = (t(g[k]) for k in ('f4', 'f3', 'f2', 'f12'))
    ops = {
        'up2_f4': lambda a: OU.upsample2d(a, f4, up=2),
        'down2_f4': lambda a: OU.downsample2d(a, f4, down=2),
        'down2_f2': lambda a: OU.downsample2d(a, f2, down=2),
        'blur_f3': lambda a: OU.filter2d(a, f3),
        'up2_f12': lambda a: OU.upsample2d(a, f12, up=2),
        'down2_f12': lambda a: OU.downsample2d(a, f12, down=2),
        'up4_f12_pad': lambda a: OU.upfirdn2d(a, f12, up=4, padding=[3, 2, 1, 4], gain=16),
    }
    x = t(g['x']).requires_grad_(True)
    y = ops[name](x)
    torch.testing.assert_close(y, t(g[name + '_y']), rtol=1e-5, atol=1e-5)
    dy = t(g[name + '_dy']).requires_grad_(True)
    dx, = torch.autograd.grad(y, x, dy, create_graph=True)
    torch.testing.assert_close(dx, t(g[name + '_dx']), rtol=1e-5, atol=1e-5)
    ddy, = torch.autograd.grad(dx, dy, t(g[name + '_ddx']))
    torch.testing.assert_close(ddy, t(g[name + '_ddy']), rtol=1e-5, atol=1e-5)


def test_setup_filter(golden):
    g = golden('setup_filter')
    cases = [([1, 3, 3, 1], {}), ([1, 2, 1], dict(gain=4)), ([1, 1], dict(normalize=False)),
             (list(range(1, 13)), {}), (list(range(1, 13)), dict(flip_filter=True, gain=2)),
             ([[1, 2], [3, 4]], dict(flip_filter=True)), (None, {}), ([1, 2, 3, 4, 5, 6, 7, 8], dict(separable=False))]
    for i, (taps, kw) in enumerate(cases):
        f = OU.setup_filter(taps, **kw)
        ref = t(g[f'sf{i}'])
        assert f.shape == ref.shape and f.dtype == torch.float32
        torch.testing.assert_close(f, ref, rtol=1e-6, atol=1e-7)


def test_sg2_resampling_layers_map_onto_upfirdn2d(golden):
    """Upsample(bilinear) == clamp-edge upsample2d([1,3,3,1]); Blur2d == filter2d([1,2,1]); AvgPool2d(2) == downsample2d([1,1])."""
    g = golden('sg2_resample')
    x = t(g['x'])
    up = OU.upsample2d(x, OU.setup_filter([1, 3, 3, 1]), up=2, edge='clamp')
    torch.testing.assert_close(up, t(g['up']), rtol=1e-6, atol=1e-6)
    blur = OU.filter2d(t(g['up']), OU.setup_filter([1, 2, 1]))
    torch.testing.assert_close(blur, t(g['blur']), rtol=1e-6, atol=1e-6)
    pool = OU.downsample2d(t(g['blur']), OU.setup_filter([1, 1]), down=2)
    torch.testing.assert_close(pool, t(g['pool']), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('act', list(OB.ACTIVATIONS))
def test_bias_act(golden, act):
    g = golden('bias_act')
    for ci, clamp in enumerate([None, 0.5]):
        for bi, use_b in enumerate([False, True]):
            tag = f'{act}_c{ci}_b{bi}'
            x = t(g['x']).requires_grad_(True)
            b = t(g['b']).requires_grad_(True)
            y = OB.bias_act(x, b if use_b else None, dim=1, act=act, clamp=clamp)
            torch.testing.assert_close(y, t(g[tag + '_y']), rtol=1e-6, atol=1e-6)
            dy = t(g[tag + '_dy']).requires_grad_(True)
            grads = torch.autograd.grad(y, [x, b] if use_b else [x], dy, create_graph=True)
            torch.testing.assert_close(grads[0], t(g[tag + '_dx']), rtol=1e-5, atol=1e-6)
            if use_b:
                torch.testing.assert_close(grads[1], t(g[tag + '_db']), rtol=1e-5, atol=1e-5)
            s = (grads[0] * t(g[tag + '_ddx'])).sum()
            g2 = torch.autograd.grad(s, [dy, x], allow_unused=True)
            torch.testing.assert_close(g2[0], t(g[tag + '_ddy']), rtol=1e-5, atol=1e-6)
            d2x = g2[1] if g2[1] is not None else torch.zeros_like(x)
            torch.testing.assert_close(d2x, t(g[tag + '_d2x']), rtol=1e-5, atol=1e-6)
    y2 = OB.bias_act(t(g['x2']), t(g['b']), dim=1, act=act, alpha=0.3, gain=1.7)
    torch.testing.assert_close(y2, t(g[f'{act}_rank2_y']), rtol=1e-6, atol=1e-6)


def test_filtered_lrelu(golden):
    g = golden('filtered_lrelu')
    for name in [str(n) for n in g['names']]:
        up, down, px0, px1, py0, py1 = [int(v) for v in g[name + '_cfg']]
        gain, slope, clamp = [float(v) for v in g[name + '_gsc']]
        clamp = None if clamp < 0 else clamp
        fu = t(g[str(g[name + '_fu'])]) if str(g[name + '_fu']) else None
        fd = t(g[str(g[name + '_fd'])]) if str(g[name + '_fd']) else None
        x = t(g[name + '_x']).requires_grad_(True)
        b = t(g[name + '_b']).requires_grad_(True)
        y = OF.filtered_lrelu(x, fu=fu, fd=fd, b=b, up=up, down=down, padding=[px0, px1, py0, py1], gain=gain, slope=slope, clamp=clamp)
        torch.testing.assert_close(y, t(g[name + '_y']), rtol=1e-5, atol=1e-5)
        dy = t(g[name + '_dy']).requires_grad_(True)
        dx, db = torch.autograd.grad(y, [x, b], dy, create_graph=True)
        torch.testing.assert_close(dx, t(g[name + '_dx']), rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(db, t(g[name + '_db']), rtol=1e-4, atol=1e-4)
        ddy, = torch.autograd.grad(dx, dy, t(g[name + '_ddx']))
        torch.testing.assert_close(ddy, t(g[name + '_ddy']), rtol=1e-4, atol=1e-5)


def test_filtered_lrelu_bf16_input_golden(golden):
    """The oracle on the bf16-representable operands of the second reference fixture (radial down filter, up 2 / up 4, multi-tile maps)."""
    g = golden('filtered_lrelu_bf16')
    for name in [str(n) for n in g['names']]:
        up, down, px0, px1, py0, py1 = [int(v) for v in g[name + '_cfg']]
        x = t(g[name + '_x']).requires_grad_(True)
        b = t(g[name + '_b']).requires_grad_(True)
        y = OF.filtered_lrelu(x, fu=t(g[str(g[name + '_fu'])]), fd=t(g['fd12r']), b=b, up=up, down=down, padding=[px0, px1, py0, py1],
                              gain=2 ** 0.5, slope=0.2, clamp=256.0)
        torch.testing.assert_close(y, t(g[name + '_y']), rtol=1e-5, atol=1e-5)
        dx, db = torch.autograd.grad(y, [x, b], t(g[name + '_dy']))
        torch.testing.assert_close(dx, t(g[name + '_dx']), rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(db, t(g[name + '_db']), rtol=1e-4, atol=1e-4)


def test_sign_packing_layout():
    code = torch.tensor([[[[0, 1, 2, 0, 1, 1, 2, 2, 0, 0, 0, 0, 1, 0, 0, 2, 1]]]], dtype=torch.uint8)
    p = OF.pack_signs(code)
    assert p.shape == (1, 1, 1, 8)          # ceil16(17)=32 elements -> 8 bytes
    assert p[0, 0, 0, 0].item() == (0 | 1 << 2 | 2 << 4 | 0 << 6)
    assert p[0, 0, 0, 1].item() == (1 | 1 << 2 | 2 << 4 | 2 << 6)
    assert p[0, 0, 0, 4].item() == 1


def test_c_oracle_agrees_with_golden_and_python_oracle(golden):
    """oracle/upfirdn2d_oracle.c (gather form, plain C) against the reference-generated integer matrix."""
    import ctypes
    import os
    from conftest import ROOT
    so = os.path.join(ROOT, 'oracle', 'libupfirdn2d_oracle.so')
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle')])
    L = ctypes.CDLL(so)
    g = golden('upfirdn2d_int')
    x = np.ascontiguousarray(g['x'], dtype=np.float32)
    names = [str(n) for n in g['filter_names']]
    checked = 0
    for i, spec in enumerate(g['specs']):
        fi, upx, upy, dnx, dny, px0, px1, py0, py1, flip = [int(v) for v in spec]
        if names[fi] in ('none', 'sep12'):
            continue
        f = np.ascontiguousarray(g['f_' + names[fi]], dtype=np.float32)
        ref = g[f'y{i}']
        y = np.zeros(ref.shape, dtype=np.float32)
        L.upfirdn2d_oracle_f32(x.ctypes.data_as(ctypes.c_void_p), f.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p),
                               x.shape[0], x.shape[1], x.shape[2], x.shape[3], f.shape[0], f.shape[1], ref.shape[2], ref.shape[3],
                               upx, upy, dnx, dny, px0, py0, flip, ctypes.c_float(4.0))
        assert np.array_equal(y, ref), (i, spec.tolist())
        checked += 1
    assert checked > 200


def test_conv2d_resample_oracle_matches_the_reference(golden):
    """oracle/conv2d_resample.py (definition form) against the reference's conv2d_resample, whose 14 fixture cases take each fast path."""
    from oracle import conv2d_resample as OC
    g = golden('conv2d_resample')
    filt = {0: None, 1: t(g['f4']), 2: t(g['f12'])}
    for idx, (k, up, down, px0, px1, py0, py1, fi, flip_w, flip_f) in enumerate(g['cases'].tolist()):
        x = t(g[f'c{idx}_x']).requires_grad_(True)
        w = t(g[f'c{idx}_w']).requires_grad_(True)
        y = OC.conv2d_resample(x, w, f=filt[fi], up=up, down=down, padding=[px0, px1, py0, py1], flip_weight=bool(flip_w), flip_filter=bool(flip_f))
        ref = t(g[f'c{idx}_y'])
        assert y.shape == ref.shape, (idx, y.shape, ref.shape)
        torch.testing.assert_close(y, ref, rtol=1e-5, atol=2e-5)
        dx, dw = torch.autograd.grad(y, [x, w], t(g[f'c{idx}_gy']))
        torch.testing.assert_close(dx, t(g[f'c{idx}_dx']), rtol=1e-5, atol=2e-5)
        torch.testing.assert_close(dw, t(g[f'c{idx}_dw']), rtol=1e-4, atol=5e-5)
