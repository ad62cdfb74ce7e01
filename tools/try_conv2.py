import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from animeface_amd.implementations.StyleGAN2 import conv as C

dev = 'cuda'
torch.manual_seed(0)

def rel(a, b):
    return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-6)).item()

def check(N, Cin, Cout, H, W, k, scales):
    x = torch.randn(N, Cin, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5).requires_grad_(True)
    s = (torch.rand(N, Cin, device=dev) + 0.5).requires_grad_(True) if scales else None
    d = (torch.rand(N, Cout, device=dev) + 0.5).requires_grad_(True) if scales else None
    y = C.conv2d(x, w, s, d)
    dy = torch.randn_like(y)
    ins = [x, w] + ([s, d] if scales else [])
    g = torch.autograd.grad(y, ins, dy, create_graph=True)
    # second order: sum of squares of dx (R1-like)
    pen = g[0].float().square().sum()
    g2 = torch.autograd.grad(pen, [w] + ([s, d] if scales else []), allow_unused=True)
    # fp32 reference with the same bf16-rounded operands
    xr = x.detach().float().requires_grad_(True)
    wr = w.detach().to(torch.bfloat16).float().requires_grad_(True)
    sr = s.detach().clone().requires_grad_(True) if scales else None
    dr = d.detach().clone().requires_grad_(True) if scales else None
    xin = xr * sr[:, :, None, None] if scales else xr
    yr = F.conv2d(xin, wr, padding=k // 2)
    if scales:
        yr = yr * dr[:, :, None, None]
    insr = [xr, wr] + ([sr, dr] if scales else [])
    gr = torch.autograd.grad(yr, insr, dy.float(), create_graph=True)
    penr = gr[0].square().sum()
    g2r = torch.autograd.grad(penr, [wr] + ([sr, dr] if scales else []), allow_unused=True)
    names = ['y', 'dx', 'dw'] + (['ds', 'dd'] if scales else []) + ['d2w'] + (['d2s', 'd2d'] if scales else [])
    vals = [(y, yr)] + list(zip(g, gr)) + list(zip(g2, g2r))
    errs = {n: round(rel(a, b), 4) for n, (a, b) in zip(names, vals)}
    ok = all(v < 0.03 for v in errs.values())
    print(f'N{N} Cin{Cin} Cout{Cout} {H}x{W} k{k} scales={scales}', errs, 'OK' if ok else 'FAIL')
    return ok

ok = True
for args in [(2, 32, 64, 16, 16, 3), (3, 64, 32, 9, 13, 3), (16, 128, 64, 4, 4, 3), (2, 64, 64, 40, 40, 3), (2, 64, 64, 16, 16, 1), (2, 40, 72, 7, 5, 3), (4, 65, 32, 4, 4, 3)]:
    for sc in (False, True):
        ok &= check(*args, sc)
print('ALL OK' if ok else 'SOME FAILED')

def bench(N, Cin, Cout, H, W, k=3, reps=10):
    x = torch.randn(N, Cin, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(N, Cout, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    for _ in range(3):
        C.conv2d_wgrad_raw(x, dy, k)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        C.conv2d_wgrad_raw(x, dy, k)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    fl = 2.0 * N * H * W * Cin * Cout * k * k
    print(json.dumps(dict(op='wgrad', N=N, Cin=Cin, Cout=Cout, H=H, W=W, k=k, ms=round(ms, 4), TFLOPs=round(fl / ms / 1e9, 1))), flush=True)

for args in [(64, 32, 64, 256, 256), (64, 64, 64, 256, 256), (64, 64, 128, 128, 128), (64, 128, 128, 128, 128), (64, 128, 256, 64, 64),
             (64, 256, 256, 64, 64), (64, 256, 512, 32, 32), (64, 512, 512, 32, 32), (64, 512, 512, 16, 16), (64, 512, 512, 8, 8), (64, 512, 512, 4, 4)]:
    bench(*args)
