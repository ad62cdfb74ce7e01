"""Per-step wall times of the SG2 training step on a fresh process (is the first run on a fresh box slow, and for how long?)."""
import sys, os, time, functools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animeface_amd.implementations.StyleGAN2 import utils as U, model as M
from animeface_amd.nnutils import sample_nnoise, update_ema
t00 = time.time()
dev = torch.device('cuda')
torch.manual_seed(0)
G, G_ema, D = M.Generator(256).to(dev), M.Generator(256).to(dev), M.Discriminator(256).to(dev)
G.init_weight(functools.partial(M.init_weight_N01, lr=0.01), M.init_weight_N01); D.apply(M.init_weight_N01); update_ema(G, G_ema, decay=0)
oG, oD = U.build_optimizers(G, D, 0.001, (0., 0.99), 10., 0., 16, 8)
step = U.TrainStep(G, G_ema, D, oG, oD, 10., 0., 16, 8, 'color,translation', 512, functools.partial(sample_nnoise, device=dev))
real = torch.rand(64, 3, 256, 256, device=dev) * 2 - 1
torch.cuda.synchronize()
print('setup %.1f s' % (time.time() - t00))
ts = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    t0 = time.time(); step(real); torch.cuda.synchronize(); ts.append((time.time() - t0) * 1e3)
print(' '.join('%.0f' % t for t in ts))
print('reserved GB %.1f' % (torch.cuda.memory_reserved() / 2**30))
