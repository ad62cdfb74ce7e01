#!/usr/bin/env python3
"""bias_act through the C ABI in a tight loop (no autograd wrapper) vs through the Python op: is the microbenchmark's 0.6 the kernel or the host?"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from animeface_amd import _lib
from animeface_amd.stylegan3_ops import bias_act as B
dev = 'cuda'
for dt in (torch.bfloat16, torch.float32):
    for rnd in (True, False):
        x = (torch.randn(64, 64, 256, 256, device=dev) if rnd else torch.full((64, 64, 256, 256), 0.0115, device=dev)).to(dt).contiguous(memory_format=torch.channels_last)
        b = torch.randn(64, device=dev).to(dt)
        y = torch.empty_like(x)
        L = _lib.lib()
        st = _lib.stream_ptr(x)
        def raw():
            L.agf_bias_act(_lib.ptr(x), _lib.ptr(b), None, None, None, _lib.ptr(y), _lib.dtype_code(x), x.numel(), 64, 1, 0, 3, 0.2, 2 ** 0.5, -1.0, st)
        def op():
            B.bias_act(x, b, act='lrelu')
        for name, fn in (('raw C ABI', raw), ('python op', op)):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): fn()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20)
            nb = 2 * x.numel() * x.element_size()
            print(f'{dt} {"random" if rnd else "constant"} data, {name}: {best * 1e3:.1f} us  {nb / best / 1e9 / 8:.3f} of 8 TB/s', flush=True)
