#!/usr/bin/env python3
"""A/B of strip height x register-pipeline depth of upfirdn2d_nhwc_rows (bf16, channels-last, batch 64, 64 channels): one subprocess per
variant id (AGF_X_VAR, read once per process by the experiment hook in launch_nhwc)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import torch
    from animeface_amd.stylegan3_ops import upfirdn2d as U
    from bench_kernels import timeit
    dev = 'cuda'
    f4, f3 = U.setup_filter([1, 3, 3, 1], device=dev), U.setup_filter([1, 2, 1], device=dev)
    f6 = U.setup_filter([1, 5, 10, 10, 5, 1], device=dev)
    mk = lambda s: torch.randn(64, 64, s, s, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x128, x256 = mk(128), mk(256)
    cases = [('up2_f4', lambda: U.upsample2d(x128, f4, up=2), x128.numel() * 5 * 2), ('blur_f3', lambda: U.filter2d(x256, f3), x256.numel() * 4),
             ('down2_f4', lambda: U.downsample2d(x256, f4, down=2), x256.numel() * 2.5), ('up2_f6', lambda: U.upsample2d(x128, f6, up=2), x128.numel() * 10),
             ('down2_f6', lambda: U.downsample2d(x256, f6, down=2), x256.numel() * 2.5)]
    for name, fn, nb in cases:
        with torch.no_grad():
            sec = min(timeit(fn, 30) for _ in range(3))
            chk = fn().float().abs().sum().item()
        print(json.dumps(dict(var=os.environ.get('AGF_X_VAR', '0'), kernel=name, us=round(sec * 1e6, 1), frac=round(nb / sec / 8e12, 4), chk=chk)), flush=True)
else:
    for v in range(5):
        subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=dict(os.environ, AGF_X_VAR=str(v)))
