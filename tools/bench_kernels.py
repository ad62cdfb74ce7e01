#!/usr/bin/env python3
"""Kernel-level micro-benchmarks: algorithmic GB/s of upfirdn2d / bias_act at the
256^2 shapes, HIP-event timed on the launch stream.  Prints one JSON line per row."""
import argparse
import json
import sys
import os

import torch
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from animeface_amd import _lib as _agf_lib
if _os.environ.get('AGF_PROBE_LIB'):
    _agf_lib.LIB_PATH = _os.path.join(_os.path.dirname(_agf_lib.LIB_PATH), 'libagf_ops_%s.so' % _os.environ['AGF_PROBE_LIB'])

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animeface_amd.stylegan3_ops import upfirdn2d as U, bias_act as B  # noqa: E402

HBM_PEAK = 8.0e12


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


def cpu_rows():
    """BASELINE.md section 3 kernel rows on the CPU: oracle/ (pinned to the reference's `_ref` paths), fp32, NCHW."""
    import time
    from oracle import upfirdn2d as OU, bias_act as OB, filtered_lrelu as OF
    import scipy.signal
    g = torch.Generator().manual_seed(0)
    x128 = torch.randn(16, 64, 128, 128, generator=g)
    x256 = torch.randn(16, 64, 256, 256, generator=g)
    f4, f3 = OU.setup_filter([1, 3, 3, 1]), OU.setup_filter([1, 2, 1])
    b = torch.randn(64, generator=g)
    fl = torch.as_tensor(scipy.signal.firwin(numtaps=12, cutoff=2.0, width=2.2, fs=8.0), dtype=torch.float32)
    xf = torch.randn(4, 32, 534, 534, generator=g)
    bf = torch.randn(32, generator=g)
    cases = [('upsample2d f4x4 [16,64,128,128]->256', lambda: OU.upsample2d(x128, f4, up=2), x128.numel() * 5 * 4),
             ('downsample2d f4x4 [16,64,256,256]->128', lambda: OU.downsample2d(x256, f4, down=2), x256.numel() * 1.25 * 4),
             ('filter2d f3x3 [16,64,256,256]', lambda: OU.filter2d(x256, f3), x256.numel() * 2 * 4),
             ('bias_act lrelu [16,64,256,256]', lambda: OB.bias_act(x256, b, act='lrelu'), x256.numel() * 2 * 4),
             ('filtered_lrelu up2/down2 12-tap sep [4,32,534,534] (SG3-512 layer 12)',
              lambda: OF.filtered_lrelu(xf, fl, fl, bf, up=2, down=2, padding=[9, 8, 9, 8], clamp=256), None)]
    with torch.no_grad():
        for name, fn, nbytes in cases:
            y = fn()
            if nbytes is None:
                nbytes = (xf.numel() + y.numel()) * 4
            t0 = time.time(); n = 0
            while time.time() - t0 < 2.0 or n < 2:
                fn(); n += 1
            sec = (time.time() - t0) / n
            print(json.dumps(dict(kernel=name, device='cpu', kind='port (oracle/)', cores=torch.get_num_threads(), dtype='float32',
                                  ms=round(sec * 1e3, 2), algorithmic_GBps=round(nbytes / sec / 1e9, 2))), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--cpu', action='store_true', help='also time the CPU oracle (the port of the reference\'s pure-PyTorch op fallback) on the '
                                                       'BASELINE.md section 3 shapes (batch 16, fp32), on this box\'s host cores')
    a = ap.parse_args()
    if a.cpu:
        cpu_rows()
    dev = 'cuda'
    N = a.batch
    f4 = U.setup_filter([1, 3, 3, 1], device=dev)
    f3 = U.setup_filter([1, 2, 1], device=dev)
    f2 = U.setup_filter([1, 1], device=dev)
    rows = []
    for dtype in [torch.bfloat16, torch.float32]:
        es = 2 if dtype == torch.bfloat16 else 4
        for layout in ['nhwc', 'nchw']:
            mf = torch.channels_last if layout == 'nhwc' else torch.contiguous_format
            x128 = torch.randn(N, 64, 128, 128, device=dev).to(dtype).contiguous(memory_format=mf)
            x256 = torch.randn(N, 64, 256, 256, device=dev).to(dtype).contiguous(memory_format=mf)
            b = torch.randn(64, device=dev).to(dtype)
            cases = [
                ('up2_f4', lambda: U.upsample2d(x128, f4, up=2), x128.numel() * 5 * es),
                ('up2_f4_clamp', lambda: U.upsample2d(x128, f4, up=2, edge='clamp'), x128.numel() * 5 * es),
                ('blur_f3', lambda: U.filter2d(x256, f3), x256.numel() * 2 * es),
                ('down2_f2', lambda: U.downsample2d(x256, f2, down=2), x256.numel() * 1.25 * es),
                ('down2_f4', lambda: U.downsample2d(x256, f4, down=2), x256.numel() * 1.25 * es),
                ('bias_lrelu', lambda: B.bias_act(x256, b, act='lrelu'), x256.numel() * 2 * es),
            ]
            for name, fn, nbytes in cases:
                with torch.no_grad():
                    sec = timeit(fn, a.reps)
                row = dict(kernel=name, dtype=str(dtype).split('.')[-1], layout=layout, batch=N, ms=round(sec * 1e3, 4),
                           algorithmic_GBps=round(nbytes / sec / 1e9, 1), frac_of_8TBps=round(nbytes / sec / HBM_PEAK, 4))
                print(json.dumps(row), flush=True)
                rows.append(row)
            del x128, x256
            torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
