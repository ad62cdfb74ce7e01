"""Time the stride-2 conv launches (forward, data gradient) of the StyleGAN3 discriminator shapes: python tools/time_s2.py [N]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animeface_amd.implementations.StyleGAN2.conv import _ConvS2Fwd, _ConvS2Dgrad
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
def bench(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for C, H in [(64, 512), (128, 256), (256, 128), (512, 64), (512, 32)]:
    z = torch.randn(N, C, H + 1, H + 1, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(C, C, 3, 3, device='cuda') / (3 * C ** 0.5)
    with torch.no_grad():
        y = _ConvS2Fwd.apply(z, w)
        fl = 2.0 * N * (H // 2) ** 2 * C * C * 9
        tf = bench(lambda: _ConvS2Fwd.apply(z, w))
        td = bench(lambda: _ConvS2Dgrad.apply(y, w, H + 1, H + 1))
    gb = (z.numel() + y.numel()) * 2 / 1e9
    print(json.dumps(dict(N=N, C=C, H=H, fwd_ms=round(tf, 3), fwd_TF=round(fl / tf / 1e9, 1), fwd_TBps=round(gb / tf, 2), dgrad_ms=round(td, 3), dgrad_TF=round(fl / td / 1e9, 1), dgrad_TBps=round(gb / td, 2))))
