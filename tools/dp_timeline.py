#!/usr/bin/env python3
"""Where the gradient exchange sits in the replayed data-parallel iteration (dp_mode 'segmented'): events on the compute stream at every
segment boundary of N iterations on a ONE-rank RCCL group (AGF_FORCE_DP=1), printed as medians.  Under
``rocprofv3 --kernel-trace`` the same run gives the start/end timestamps of the RCCL kernels against the generator-forward kernels.

    AGF_FORCE_DP=1 python tools/dp_timeline.py [--iters 12] [--image-size 256] [--batch 64]
"""
import argparse
import functools
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=12)
    ap.add_argument('--image-size', type=int, default=256)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--no-overlap', action='store_true', help="wait for D's exchange BEFORE the generator-forward graph (the r04 order)")
    args = ap.parse_args()
    os.environ.setdefault('AGF_FORCE_DP', '1')
    from animeface_amd import distributed as dp
    from animeface_amd.implementations.StyleGAN2 import utils as U, model as M
    from animeface_amd.nnutils import sample_nnoise, update_ema
    rank, world, local = dp.init_distributed()
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    S = args.image_size
    G, G_ema, D = M.Generator(S).to(dev), M.Generator(S).to(dev), M.Discriminator(S).to(dev)
    G.init_weight(functools.partial(M.init_weight_N01, lr=0.01), M.init_weight_N01)
    D.apply(M.init_weight_N01)
    G_ema.eval()
    update_ema(G, G_ema, decay=0)
    opt_G, opt_D = U.build_optimizers(G, D, 0.001, (0., 0.99), 10., 0., 16, 8, capturable=True)
    red_G = dp.GradReducer(G.parameters(), never_used=dp.never_used_parameters(G))
    red_D = dp.GradReducer(D.parameters())
    step = U.TrainStep(G, G_ema, D, opt_G, opt_D, 10., 0., 16, 8, 'color,translation', 512, functools.partial(sample_nnoise, device=dev), red_G, red_D)
    real = (torch.rand(args.batch, 3, S, S) * 2 - 1).to(dev)
    for _ in range(2):
        step(real)
    runner = U.GraphedTrainStep(step, real, warmup=0, dp_mode='segmented', pace=0)
    step.batches_done = 1
    runner._capture(1)
    (g1, g2a, g2b, g3), _, _ = runner.graphs[('gan', 0)]
    for _ in range(3):
        runner._replay((g1, g2a, g2b, g3))
    torch.cuda.synchronize()
    names = ['seg1 (D half-step)', "launch D exchange -> seg2a (G forward)", "wait for D's exchange", 'seg2b (Adam D, rest of the G half-step)', "G's exchange (launch + wait)", 'seg3 (Adam G, EMA)']
    rows = []
    for _ in range(args.iters):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
        ev[0].record()
        g1.replay()
        ev[1].record()
        red_D.launch_all()
        if args.no_overlap:
            red_D.wait_all()
        g2a.replay()
        ev[2].record()
        if not args.no_overlap:
            red_D.wait_all()
        ev[3].record()
        g2b.replay()
        ev[4].record()
        red_G.launch_all()
        red_G.wait_all()
        ev[5].record()
        g3.replay()
        ev[6].record()
        rows.append(ev)
    torch.cuda.synchronize()
    import statistics
    print(f'# dp_mode segmented, one-rank RCCL group, {S}x{S} batch {args.batch}, {"exchange NOT overlapped (r04 order)" if args.no_overlap else "D exchange beside the generator forward"}; medians over {args.iters} replayed iterations, ms')
    tot = []
    for k, n in enumerate(names):
        v = [r[k].elapsed_time(r[k + 1]) for r in rows]
        print(f'{statistics.median(v):9.3f}  {n}')
    tot = [r[0].elapsed_time(r[6]) for r in rows]
    print(f'{statistics.median(tot):9.3f}  whole iteration')
    dp.dist.barrier()
    dp.dist.destroy_process_group()


if __name__ == '__main__':
    main()
