"""Planar (NCHW) upfirdn2d cases of tools/bench_kernels.py only: fraction of the 8 TB/s HBM peak per case (A/B of the planar_vec kernel:
per-case rows per lane and rows in flight are compile-time choices in launch_planar_vec_cases)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animeface_amd.stylegan3_ops import upfirdn2d as U

dev = torch.device('cuda')
HBM_PEAK = 8e12


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


def main():
    N = 64
    f4, f3, f2 = (U.setup_filter(t, device=dev) for t in ([1, 3, 3, 1], [1, 2, 1], [1, 1]))
    out = {}
    for dtype in [torch.bfloat16, torch.float32]:
        es = 2 if dtype == torch.bfloat16 else 4
        x128 = torch.randn(N, 64, 128, 128, device=dev).to(dtype)
        x256 = torch.randn(N, 64, 256, 256, device=dev).to(dtype)
        cases = [('up2_f4', lambda: U.upsample2d(x128, f4, up=2), x128.numel() * 5 * es),
                 ('blur_f3', lambda: U.filter2d(x256, f3), x256.numel() * 2 * es),
                 ('down2_f2', lambda: U.downsample2d(x256, f2, down=2), x256.numel() * 1.25 * es),
                 ('down2_f4', lambda: U.downsample2d(x256, f4, down=2), x256.numel() * 1.25 * es)]
        with torch.no_grad():
            for name, fn, nbytes in cases:
                out[f'{name}_{"bf16" if es == 2 else "f32"}'] = round(nbytes / timeit(fn) / HBM_PEAK, 3)
        del x128, x256
    print(json.dumps(out))


if __name__ == '__main__':
    main()
