"""act_bwd_reduce / scale_dot at the shapes of the SG2 step: achieved HBM GB/s (3 tensor passes each)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animeface_amd.implementations.StyleGAN2.conv import act_bwd_reduce_raw, scale_dot_raw
dev = 'cuda'
def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for N, C, H in [(128, 64, 256), (128, 128, 128), (128, 256, 64), (128, 512, 32), (128, 512, 16), (128, 512, 8), (64, 32, 256), (64, 64, 128), (64, 512, 4)]:
    y = torch.randn(N, C, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn_like(y)
    nz = torch.randn(N, 1, H, H, device=dev)
    s = torch.rand(N, C, device=dev) + 0.5
    b = 3 * y.numel() * 2
    for name, fn in [('act_bwd (B only)', lambda: act_bwd_reduce_raw(dy, y, None, 0.2, (False, True, False))),
                     ('act_bwd (A,B,C)', lambda: act_bwd_reduce_raw(dy, y, nz, 0.2, (True, True, True))),
                     ('scale_dot', lambda: scale_dot_raw(y, dy, s))]:
        ms = timeit(fn)
        print(f'{name:18s} N{N} C{C} {H}x{H}: {ms:.4f} ms  {b / ms / 1e6:8.0f} GB/s ({b / ms / 1e6 / 8000:.0%} of 8 TB/s)')
