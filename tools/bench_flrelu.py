"""filtered_lrelu: fused kernel vs the generic 4-pass composition on StyleGAN3-512 layer shapes."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from animeface_amd.stylegan3_ops import filtered_lrelu as FL, upfirdn2d as U
from animeface_amd import _lib
import scipy.signal

def lowpass(numtaps, cutoff, width, fs):
    return torch.as_tensor(scipy.signal.firwin(numtaps=numtaps, cutoff=cutoff, width=width, fs=fs), dtype=torch.float32)

dev = 'cuda'
fu = lowpass(12, 2.0, 2.2, 8.0).to(dev)
fd = lowpass(12, 2.0, 2.2, 8.0).to(dev)
fdr = torch.outer(fd, fd).contiguous()

def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps

for dtype in (torch.float32, torch.bfloat16):
    for name, B, C, S, fdd, pad in [('L12 up2/down2 sep', 8, 32, 534, fd, [9, 8, 9, 8]), ('L1 up2/down2 radial', 8, 512, 38, fdr, [9, 8, 9, 8])]:
        x = torch.randn(B, C, S, S, device=dev).to(dtype)
        b = torch.randn(C, device=dev).to(dtype)
        with torch.no_grad():
            y = FL.filtered_lrelu(x, fu, fdd, b, up=2, down=2, padding=pad, clamp=256)
            ms = timeit(lambda: FL.filtered_lrelu(x, fu, fdd, b, up=2, down=2, padding=pad, clamp=256))
            # generic composition for comparison
            def generic():
                t = x.add(b[None, :, None, None])
                t = U.upfirdn2d(t, fu, up=2, padding=pad, gain=4)
                FL._native_act_(t, None, 0, 0, float(np.sqrt(2)), 0.2, 256.0, False)
                return U.upfirdn2d(t, fdd, down=2)
            yg = generic()
            msg = timeit(generic)
        es = 2 if dtype == torch.bfloat16 else 4
        nbytes = (x.numel() + y.numel()) * es
        err = (y.float() - yg.float()).abs().max().item()
        print(json.dumps(dict(case=name, dtype=str(dtype), fused_ms=round(ms, 4), generic_ms=round(msg, 4), fused_GBps=round(nbytes / ms / 1e6, 1),
                              frac_8TBps=round(nbytes / ms / 1e6 / 8000, 4), max_diff_vs_generic=err, out=list(y.shape))), flush=True)
