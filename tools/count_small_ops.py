"""Count python-level call sites of small allocation / elementwise helpers during one SG2 training step."""
import sys, os, functools, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from animeface_amd.implementations.StyleGAN2 import utils as U, model as M
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
def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if 'animeface_amd' in fr.filename:
            return f'{fr.filename.split("animeface_amd/")[-1]}:{fr.lineno}'
    return '?'
def wrap(obj, name):
    orig = getattr(obj, name)
    def f(*a, **k):
        cnt[(name, site())] += 1
        return orig(*a, **k)
    setattr(obj, name, f)
for obj, names in [(torch, ['zeros', 'zeros_like', 'full', 'ones', 'ones_like', 'full_like', 'empty_like']), (F, ['pad', 'linear']),
                   (torch.Tensor, ['zero_', 'fill_', 'float', 'to', 'contiguous', 'sum', 'square', 'clone', '__mul__', '__rmul__', '__add__', '__radd__', '__truediv__', '__sub__'])]:
    for n in names:
        wrap(obj, n)
step(real)
torch.cuda.synchronize()
for (name, s), n in cnt.most_common(70):
    print(f'{n:5d}  {name:12s} {s}')
