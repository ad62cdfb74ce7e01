"""Kernel time of lazy-R1 iterations only (d_k = 1: the penalty replaces the GAN loss on every step), for rocprofv3 --kernel-trace --stats:
   rocprofv3 --kernel-trace --stats --output-format csv -d OUT -o r1 -- python tools/r1_iteration_stats.py"""
import sys, os, functools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animeface_amd.implementations.StyleGAN2 import utils as U, model as M
from animeface_amd.nnutils import sample_nnoise, update_ema
dev = torch.device('cuda')
torch.manual_seed(0)
G, G_ema, D = M.Generator(256).to(dev), M.Generator(256).to(dev), M.Discriminator(256).to(dev)
G.init_weight(functools.partial(M.init_weight_N01, lr=0.01), M.init_weight_N01); D.apply(M.init_weight_N01); update_ema(G, G_ema, decay=0)
oG, oD = U.build_optimizers(G, D, 0.001, (0., 0.99), 10., 0., 1, 8, capturable=True)
step = U.TrainStep(G, G_ema, D, oG, oD, 10., 0., 1, 8, 'color,translation', 512, functools.partial(sample_nnoise, device=dev))
real = torch.rand(64, 3, 256, 256, device=dev) * 2 - 1
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for _ in range(N):
    step(real)
torch.cuda.synchronize()
print('ran', N, 'R1 iterations (iteration 0 is a GAN-loss one)')
