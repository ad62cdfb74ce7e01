"""filtered_lrelu roofline table: every layer of the StyleGAN3-T 512x512 generator (BASELINE.json configs[3]) at batch 16, bf16 --
forward and gradient kernel time, ALGORITHMIC HBM bytes per SURVEY.md section 8(d) ((numel_x + numel_y) * sizeof(T) + the 2-bit sign
tensor when gradients are needed) and the fraction of the 8 TB/s HBM peak; one JSON line per distinct layer configuration.
The op is VALU-bound, so every row also carries its fused-multiply-add count (polyphase taps actually evaluated: taps / up per sample of a
separable pass, (taps / up)^2 for the radial 2-D filter) and the time those FMAs take at the fp32 vector peak (157.3 TFLOP/s = 78.6 T FMA/s,
MI355X_MICROARCH.md): ``*_valu_bound_ms`` and the fraction of it the kernel reaches."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animeface_amd.stylegan3_ops import filtered_lrelu as FL
from animeface_amd.implementations.StyleGAN3 import model as M

dev = 'cuda'
B = int(os.environ.get('B', '16'))
G = M.Generator(512, 512).to(dev)


def timeit(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


seen = {}
size = G.synthesis.input.size[0]
for i, layer in enumerate(G.synthesis.net):
    cin = layer.conv.weight.shape[1]; cout = layer.conv.weight.shape[0]; k = layer.conv.weight.shape[2]
    s_in = size + k - 1                                       # the conv's "full" padding: its output is the filtered_lrelu input
    x = torch.randn(B, cout, s_in, s_in, device=dev).to(torch.bfloat16).requires_grad_(True)
    b = layer.bias.detach().to(torch.bfloat16)
    fu, fd = layer.up_filter, layer.down_filter
    args = dict(up=layer.up_factor, down=layer.down_factor, padding=layer.padding, gain=layer.gain, slope=layer.negative_slope, clamp=layer.conv_clamp)
    y = FL.filtered_lrelu(x, fu, fd, b, **args)
    size = y.shape[-1]
    key = (cout, s_in, layer.up_factor, layer.down_factor, None if fu is None else tuple(fu.shape), None if fd is None else tuple(fd.shape), tuple(layer.padding))
    if key in seen:
        continue
    seen[key] = i
    gy = torch.randn_like(y)
    with torch.no_grad():
        ms_f = timeit(lambda: FL.filtered_lrelu(x.detach(), fu, fd, b, **args))
    ms_fb = timeit(lambda: torch.autograd.grad(FL.filtered_lrelu(x, fu, fd, b, **args), x, gy))
    sw = s_in * layer.up_factor + layer.padding[0] + layer.padding[1] - ((fu.shape[-1] if fu is not None else 1) - 1)
    sign_bytes = B * cout * sw * ((sw + 15) // 16 * 16) // 4
    alg_f = (x.numel() + y.numel()) * 2
    alg_b = alg_f + sign_bytes                                 # the gradient pass reads dy and the signs, writes dx
    radial = fd is not None and fd.ndim == 2

    def fir_fmas(n_in, up, taps_up, up_radial, down, taps_down, down_radial, n_out):
        """FMAs per plane of: zero-insert by `up` + FIR (polyphase), then FIR + keep every `down`-th sample."""
        u = n_in * up + layer.padding[0] + layer.padding[1] - (taps_up - 1) if taps_up > 1 else n_in * up
        tu = max(taps_up // up, 1)
        f_up = u * u * tu * tu if up_radial else (u * n_in * tu + u * u * tu) if taps_up > 1 else 0
        f_dn = n_out * n_out * taps_down * taps_down if down_radial else (n_out * u * taps_down + n_out * n_out * taps_down) if taps_down > 1 else 0
        return f_up + f_dn
    tu_, td_ = (1 if fu is None else int(fu.shape[-1])), (1 if fd is None else int(fd.shape[-1]))
    planes = B * cout
    fma_f = planes * fir_fmas(s_in, layer.up_factor, tu_, False, layer.down_factor, td_, radial, int(y.shape[-1]))
    # the gradient kernel: dy upsampled by `down` with the (flipped) down filter, masked, then decimated by `up` with the up filter
    fma_b = planes * fir_fmas(int(y.shape[-1]), layer.down_factor, td_, radial, layer.up_factor, tu_, False, s_in)
    VALU_FMA_PER_S = 157.3e12 / 2
    print(json.dumps(dict(layer=i, channels=cout, in_size=s_in, out_size=int(y.shape[-1]), up=layer.up_factor, down=layer.down_factor,
                          up_taps=None if fu is None else int(fu.shape[-1]), down_filter=None if fd is None else ('radial %dx%d' % tuple(fd.shape) if radial else 'separable %d' % fd.shape[0]),
                          batch=B, fwd_ms=round(ms_f, 4), fwd_bwd_ms=round(ms_fb, 4), bwd_ms=round(ms_fb - ms_f, 4),
                          algorithmic_MB_fwd=round(alg_f / 1e6, 1), fwd_TBps=round(alg_f / ms_f / 1e9, 3), fwd_frac_of_8TBps=round(alg_f / ms_f / 1e9 / 8, 4),
                          bwd_TBps=round(alg_b / max(ms_fb - ms_f, 1e-6) / 1e9, 3), bwd_frac_of_8TBps=round(alg_b / max(ms_fb - ms_f, 1e-6) / 1e9 / 8, 4),
                          fwd_GFMA=round(fma_f / 1e9, 2), fwd_valu_bound_ms=round(fma_f / VALU_FMA_PER_S * 1e3, 4), fwd_frac_of_valu_bound=round(fma_f / VALU_FMA_PER_S * 1e3 / ms_f, 4),
                          bwd_GFMA=round(fma_b / 1e9, 2), bwd_valu_bound_ms=round(fma_b / VALU_FMA_PER_S * 1e3, 4),
                          bwd_frac_of_valu_bound=round(fma_b / VALU_FMA_PER_S * 1e3 / max(ms_fb - ms_f, 1e-6), 4))), flush=True)
