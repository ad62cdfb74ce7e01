import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from animeface_amd.implementations.StyleGAN2.conv import conv2d_fwd_raw

dev = 'cuda'
torch.manual_seed(0)

def check(N, Cin, Cout, H, W, k, scales=False):
    x = torch.randn(N, Cin, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5).to(torch.bfloat16)
    kw = {}
    ref_x = x.float()
    if scales:
        s = torch.rand(N, Cin, device=dev) + 0.5
        d = torch.rand(N, Cout, device=dev) + 0.5
        b = torch.randn(Cout, device=dev)
        nz = torch.randn(N, 1, H, W, device=dev)
        kw = dict(in_scale=s, out_scale=d, bias=b, noise=nz, act=3, alpha=0.2, gain=1.0)
        ref_x = (x.float() * s[:, :, None, None]).to(torch.bfloat16).float()
    y = conv2d_fwd_raw(x, w, **kw)
    ref = F.conv2d(ref_x, w.float(), padding=k // 2)
    if scales:
        ref = F.leaky_relu(ref * d[:, :, None, None] + b[None, :, None, None] + nz, 0.2)
    err = (y.float() - ref).abs().max().item()
    tol = ref.abs().max().item() * 2 ** -7
    print(f'N{N} Cin{Cin} Cout{Cout} {H}x{W} k{k} scales={scales}: max err {err:.4g} (tol {tol:.4g})', 'OK' if err <= tol else 'FAIL')
    return err <= tol

ok = True
for args in [(2, 32, 64, 16, 16, 3), (3, 64, 32, 9, 13, 3), (2, 32, 32, 32, 64, 3), (16, 512, 512, 4, 4, 3), (5, 128, 256, 8, 8, 3),
             (2, 64, 128, 40, 40, 3), (2, 64, 64, 16, 16, 1), (2, 40, 36, 7, 5, 3), (2, 520, 512, 4, 4, 3)]:
    ok &= check(*args)
    ok &= check(*args, scales=True)
print('ALL OK' if ok else 'SOME FAILED')

def bench(N, Cin, Cout, H, W, k=3, reps=10):
    x = torch.randn(N, Cin, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5).to(torch.bfloat16)
    wq = w.contiguous(memory_format=torch.channels_last)
    for _ in range(3):
        conv2d_fwd_raw(x, w)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        conv2d_fwd_raw(x, w)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    fl = 2.0 * N * H * W * Cin * Cout * k * k
    for _ in range(3):
        F.conv2d(x, wq, padding=k // 2)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        F.conv2d(x, wq, padding=k // 2)
    e.record(); torch.cuda.synchronize()
    ms2 = s.elapsed_time(e) / reps
    print(json.dumps(dict(N=N, Cin=Cin, Cout=Cout, H=H, W=W, k=k, ms=round(ms, 4), TFLOPs=round(fl / ms / 1e9, 1),
                          miopen_ms=round(ms2, 4), miopen_TFLOPs=round(fl / ms2 / 1e9, 1))), flush=True)

for args in [(64, 32, 64, 256, 256), (64, 64, 64, 256, 256), (64, 64, 128, 128, 128), (64, 128, 128, 128, 128), (64, 128, 256, 64, 64),
             (64, 256, 256, 64, 64), (64, 256, 512, 32, 32), (64, 512, 512, 32, 32), (64, 512, 512, 16, 16), (64, 512, 512, 8, 8), (64, 512, 512, 4, 4)]:
    bench(*args)
