"""Which activation tensors still go through the separate lrelu-gradient pass (agf_act_bwd_reduce) in one training step."""
import sys, os, functools, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animeface_amd.implementations.StyleGAN2 import utils as U, conv as C, model as M
from animeface_amd.nnutils import sample_nnoise, update_ema
dev = torch.device('cuda')
torch.manual_seed(0)
G, G_ema, D = M.Generator(256).to(dev), M.Generator(256).to(dev), M.Discriminator(256).to(dev)
G.init_weight(functools.partial(M.init_weight_N01, lr=0.01), M.init_weight_N01); D.apply(M.init_weight_N01); update_ema(G, G_ema, decay=0)
oG, oD = U.build_optimizers(G, D, 0.001, (0., 0.99), 10., 0., 16, 8)
step = U.TrainStep(G, G_ema, D, oG, oD, 10., 0., 16, 8, 'color,translation', 512, functools.partial(sample_nnoise, device=dev))
real = torch.rand(64, 3, 256, 256, device=dev) * 2 - 1
for _ in range(2): step(real)
cnt = collections.Counter()
orig = C.act_bwd_reduce_raw
def wrapped(dy, y, noise, alpha, want):
    cnt[(tuple(y.shape), tuple(bool(w) for w in want))] += 1
    return orig(dy, y, noise, alpha, want)
C.act_bwd_reduce_raw = wrapped
step(real)
for k, v in sorted(cnt.items(), key=lambda kv: -kv[0][0][0] * kv[0][0][1] * kv[0][0][2] * kv[0][0][3]):
    n, c, h, w = k[0]
    print(v, k, f'{n * c * h * w * 6 / 1e6:.0f} MB moved')
