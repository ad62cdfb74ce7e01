"""Run-to-run repeatability of the raw ops at the tiny-network shapes, with the allocator pool dirtied between runs (exposes reads beyond a
tensor's written extent and races; fp32 atomics alone move results by ~1e-7 relative)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animeface_amd.implementations.StyleGAN2 import conv as C
dev = 'cuda'

def dirty():
    t = [torch.randn(1 << 20, device=dev) * 1e3 for _ in range(8)]
    u = [torch.randn(n, device=dev) * 1e3 for n in (64, 256, 1000, 4096, 5000, 16384, 70000)]
    del t, u

def rel(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-12))

def check(name, fn, n=6):
    ref = None; worst = 0.0
    for i in range(n):
        dirty()
        out = fn()
        outs = [o for o in (out if isinstance(out, (tuple, list)) else (out,)) if isinstance(o, torch.Tensor)]
        outs = [o.clone() for o in outs]
        if ref is None: ref = outs
        else: worst = max([worst] + [rel(a, b) for a, b in zip(outs, ref)])
    print(f'{"!!" if worst > 1e-4 else "  "} {name}: worst run-to-run rel diff {worst:.3g}', flush=True)

g = torch.Generator().manual_seed(0)
def mk(N, C, H, W):
    return torch.randn(N, C, H, W, generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)

for N in (4, 8):
    for (Cin, Cout, H) in [(64, 64, 4), (64, 64, 8), (64, 32, 16), (32, 16, 32), (16, 8, 32), (8, 16, 32), (16, 32, 16), (32, 64, 8), (64, 64, 4), (72, 64, 4)]:
        x, dy = mk(N, Cin, H, H), mk(N, Cout, H, H)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).to(dev)
        s_in = (torch.rand(N, Cin, generator=g) + 0.5).to(dev); s_out = (torch.rand(N, Cout, generator=g) + 0.5).to(dev)
        b = torch.randn(Cout, generator=g).to(dev); nz = torch.randn(N, 1, H, H, generator=g).to(dev)
        tag = f'N{N} {Cin}->{Cout} @{H}'
        check(f'conv fwd plain {tag}', lambda: C.conv2d_fwd_raw(x, w, bias=b, act=3))
        check(f'conv fwd modulated {tag}', lambda: C.conv2d_fwd_raw(x, w, in_scale=s_in, out_scale=s_out, bias=b, noise=nz, act=3))
        check(f'wgrad plain {tag}', lambda: C.conv2d_wgrad_raw(x, dy, 3))
        check(f'wgrad scaled {tag}', lambda: C.conv2d_wgrad_raw(x, dy, 3, in_scale=s_in, out_scale=s_out))
        msum = lambda: torch.zeros(256, Cin, device=dev)
        wt = (torch.randn(Cin, Cout, 3, 3, generator=g) / (Cout * 9) ** 0.5).to(dev)
        def masked():
            ms = msum(); y = C.conv2d_fwd_raw(dy, wt, mask_y=x, mask_alpha=0.2, mask_sum=ms); return y, ms.sum(0)
        check(f'conv fwd mask {tag}', masked)
        if H % 2 == 0:
            r = mk(N, Cin, H // 2, H // 2)
            def both():
                ms = msum(); y = C.conv2d_fwd_raw(dy, wt, mask_y=x, mask_alpha=0.2, mask_sum=ms, res_pooled=r, res_scale=0.3); return y, ms.sum(0)
            check(f'conv fwd mask+pooled {tag}', both)
        check(f'act_bwd_reduce {tag}', lambda: (lambda o: (o[0],) + tuple(o[1]))(C.act_bwd_reduce_raw(dy, mk(N, Cout, H, H) if False else dy, nz, 0.2, (True, True, True))))
        check(f'scale_dot {tag}', lambda: C.scale_dot_raw(x, mk(N, Cin, H, H) if False else x, s_in))
        if H % 2 == 0:
            dh = mk(N, Cout, H // 2, H // 2)
            check(f'act_bwd_reduce_pooled {tag}', lambda: C.act_bwd_reduce_pooled_raw(dh, dy, 0.2, 0.3, True))
        w1 = (torch.randn(8, Cin, 1, 1, generator=g) / Cin ** 0.5).to(dev)
        check(f'conv 1x1 to 8 {tag}', lambda: C.conv2d_fwd_raw(x, w1, in_scale=s_in, bias=torch.zeros(8, device=dev)))
