import sys, os, torch, time
sys.path.insert(0, '/root/repo')
from animeface_amd.implementations.StyleGAN2 import conv as C
dev = torch.device('cuda')
def bench(N, Cin, Cout, H, W, res):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, Cin, H, W, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    r = torch.randn(N, Cout, H, W, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last) if res else None
    wq = C.prep_weights_raw(w, 1.0, torch.bfloat16)[0]
    y = C.conv2d_fwd_raw(x, wq, bias=b, residual=r, act=C.ACT_LINEAR, prepared=True)
    ref = torch.nn.functional.conv2d(x.float(), w.bfloat16().float()) + b.view(1, -1, 1, 1) + (r.float() if res else 0)
    err = (y.float() - ref).abs().max().item() / ref.abs().max().item()
    for _ in range(3): C.conv2d_fwd_raw(x, wq, bias=b, residual=r, act=C.ACT_LINEAR, prepared=True)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): C.conv2d_fwd_raw(x, wq, bias=b, residual=r, act=C.ACT_LINEAR, prepared=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    by = N * H * W * (Cin + Cout * (2 if res else 1)) * 2
    print(f'N{N} {Cin}->{Cout} {H}x{W} res={res}: {ms*1e3:7.1f} us  {by/ms/1e9:5.2f} TB/s  rel err {err:.2e}')
for a in [(128, 32, 64, 128, 128, True), (128, 64, 128, 64, 64, True), (128, 128, 256, 32, 32, True), (64, 32, 64, 128, 128, True), (128, 64, 32, 128, 128, False), (128, 256, 512, 16, 16, True)]:
    bench(*a)
