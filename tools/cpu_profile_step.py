"""cProfile of the host side of the SG2 training step in the launch-bound regime (128x128, batch 32)."""
import sys, os, functools, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animeface_amd.implementations.StyleGAN2 import utils as U, model as M
from animeface_amd.nnutils import sample_nnoise, update_ema
dev = torch.device('cuda')
torch.manual_seed(0)
S, B = 128, 32
G, G_ema, D = M.Generator(S).to(dev), M.Generator(S).to(dev), M.Discriminator(S).to(dev)
G.init_weight(functools.partial(M.init_weight_N01, lr=0.01), M.init_weight_N01); D.apply(M.init_weight_N01); update_ema(G, G_ema, decay=0)
oG, oD = U.build_optimizers(G, D, 0.001, (0., 0.99), 10., 0., 16, 8)
step = U.TrainStep(G, G_ema, D, oG, oD, 10., 0., 16, 8, 'color,translation', 512, functools.partial(sample_nnoise, device=dev))
real = torch.rand(B, 3, S, S, device=dev) * 2 - 1
for _ in range(3): step(real)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10): step(real)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print(s.getvalue()[:6000])
