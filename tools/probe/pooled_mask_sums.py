"""agf_act_bwd_reduce_pooled_mask in isolation: the same inputs many times, alone and beside a second process that runs training half-steps on
the same GPU; the per-(n, c) sums must not depend on what else the GPU is doing.   python tools/probe/pooled_mask_sums.py"""
import os, sys, time, functools
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('AGF_DP_TEST_FULL', '1'); os.environ['SAME'] = '1'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools', 'probe'))
import torch
import torch.multiprocessing as mp

if __name__ == '__main__':
    import contended_determinism as CD
    from animeface_amd import _lib
    from animeface_amd.implementations.StyleGAN2 import conv as C
    dev = torch.device('cuda', 0)
    det = len(sys.argv) > 1 and sys.argv[1] == 'det'
    _lib.set_deterministic(det)
    g = torch.Generator().manual_seed(3)
    for (N, Cc, H, W) in [(8, 64, 128, 128), (8, 512, 16, 16), (8, 32, 256, 256)]:
        dy = torch.randn(N, Cc, H // 2, W // 2, generator=g).to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last)
        mask = torch.randint(-2 ** 31, 2 ** 31 - 1, (N, H // 2, W // 2, Cc // 8), generator=g, dtype=torch.int64).to(torch.int32).to(dev)
        def once():
            gg, B, R = C.act_bwd_reduce_pooled_mask_raw(dy, mask, (N, Cc, H, W), 0.2, 0.25, True, True)
            torch.cuda.synchronize()
            return gg.clone(), B.clone(), R.clone()
        ref = once()
        if os.environ.get('SIDE_STREAM'):
            # contention from a second stream of THIS process (what an overlapped all-reduce or a side-stream ADA plan is to the backward pass)
            side = torch.cuda.Stream()
            a_ = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
            x_ = torch.randn(32, 64, 256, 256, device=dev, dtype=torch.bfloat16)
            bad = [0, 0, 0]; worst = 0.0
            for it in range(200):
                with torch.cuda.stream(side):
                    for _ in range(3):
                        (a_ @ a_).sum(); x_.mul_(1.0001)
                cur = once()
                for i in range(3):
                    if not torch.equal(cur[i], ref[i]):
                        bad[i] += 1
                        if i: worst = max(worst, float((cur[i] - ref[i]).abs().max() / ref[i].abs().max()))
            torch.cuda.synchronize()
            print(f'[{N},{Cc},{H},{W}] deterministic={det} side stream busy: of 200 runs g differs {bad[0]}, sum_g {bad[1]}, sum_dy {bad[2]}; worst relative deviation {worst:.2e}', flush=True)
            continue
        for contended in (False, True):
            stop = proc = None
            if contended:
                ctx = mp.get_context('spawn'); stop = ctx.Event()
                proc = ctx.Process(target=CD.hammer_same, args=(stop,)); proc.start(); time.sleep(25)
            bad = [0, 0, 0]; worst = 0.0
            for _ in range(200):
                cur = once()
                for i in range(3):
                    if not torch.equal(cur[i], ref[i]):
                        bad[i] += 1
                        if i: worst = max(worst, float((cur[i] - ref[i]).abs().max() / ref[i].abs().max()))
            print(f'[{N},{Cc},{H},{W}] deterministic={det} {"contended" if contended else "alone"}: of 200 runs g differs {bad[0]}, sum_g {bad[1]}, sum_dy {bad[2]}; worst relative deviation {worst:.2e}', flush=True)
            if proc is not None:
                stop.set(); proc.join()
