"""Layout passes of the StyleGAN3-T 512 generator (planar <-> channels-last with border / channel padding, style scale folded in) per layer shape:
ms and TB/s by algorithmic bytes (one read + one write of the tensor).   python tools/bench_layout.py [batch]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animeface_amd import _lib
if os.environ.get('AGF_PROBE_LIB'):          # a probe build (tools/probe/build_variant.sh)
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libagf_ops_%s.so' % os.environ['AGF_PROBE_LIB'])
from animeface_amd.stylegan3_ops import layout as L
dev = 'cuda'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
shapes = [(512, 36), (512, 52), (512, 84), (362, 148), (242, 148), (161, 276), (108, 276), (72, 532), (48, 532), (32, 532)]


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    best = 1e9
    for _ in range(4):                       # best of four timed bursts (a shared box: single bursts scatter by +-15 %)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / reps)
    return best


tot = {}
for C, S in shapes:
    x = torch.randn(B, C, S, S, device=dev).to(torch.bfloat16)
    cp = L.padded_channels(C, x.dtype)
    sc = torch.rand(B, cp, device=dev) + 0.5
    y = L._to_cl_raw(x, 1, cp, scale=sc)
    nbytes = x.numel() * 2 + y.numel() * 2
    t1 = timeit(lambda: L._to_cl_raw(x, 1, cp, scale=sc))
    t2 = timeit(lambda: L._to_planar_raw(y, 1, C, scale=sc))
    yc = L._to_cl_raw(x, 0, cp)
    t3 = timeit(lambda: L._to_planar_raw(yc, 0, C))
    t4 = timeit(lambda: L._to_cl_raw(x, 0, cp, scale=sc))
    print(json.dumps(dict(C=C, size=S, batch=B, MB=round(nbytes / 1e6, 1), to_cl_pad1_scaled_ms=round(t1, 4), TBps=round(nbytes / t1 / 1e9, 2),
                          to_planar_crop1_scaled_ms=round(t2, 4), TBps2=round(nbytes / t2 / 1e9, 2), to_planar_ms=round(t3, 4), TBps3=round(nbytes / t3 / 1e9, 2),
                          to_cl_pad0_scaled_ms=round(t4, 4), TBps4=round(nbytes / t4 / 1e9, 2))), flush=True)
    for k, v in (('to_cl_pad1', t1), ('to_planar_crop1', t2), ('to_planar', t3), ('to_cl_pad0', t4)):
        tot[k] = tot.get(k, 0) + v
print(json.dumps({k: round(v, 3) for k, v in tot.items()}))
