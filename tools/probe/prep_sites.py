"""Which call sites still prepare a single weight (conv.prep_weights_raw outside the per-iteration PrepPlan launch) in one eager StyleGAN2 iteration."""
import sys, os, functools, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from animeface_amd.implementations.StyleGAN2 import utils as U, model as M, conv as C
from animeface_amd.nnutils import sample_nnoise, update_ema
dev = torch.device('cuda')
torch.manual_seed(0)
G, G_ema, D = M.Generator(256).to(dev), M.Generator(256).to(dev), M.Discriminator(256).to(dev)
G.init_weight(functools.partial(M.init_weight_N01, lr=0.01), M.init_weight_N01); D.apply(M.init_weight_N01); update_ema(G, G_ema, decay=0)
oG, oD = U.build_optimizers(G, D, 0.001, (0., 0.99), 10., 0., 16, 8, capturable=True)
step = U.TrainStep(G, G_ema, D, oG, oD, 10., 0., 16, 8, 'color,translation', 512, functools.partial(sample_nnoise, device=dev))
real = torch.rand(64, 3, 256, 256, device=dev) * 2 - 1
for _ in range(3): step(real)
cnt = collections.Counter()
orig = C.prep_weights_raw
def rec(w, *a, **k):
    fr = [f for f in traceback.extract_stack() if 'animeface_amd' in f.filename and 'prep_weights_raw' not in f.name]
    cnt[(tuple(w.shape), ' <- '.join(f'{f.filename.split("/")[-1]}:{f.lineno} {f.name}' for f in fr[-3:]))] += 1
    return orig(w, *a, **k)
C.prep_weights_raw = rec
with torch.autograd.set_multithreading_enabled(False):
    step(real)
torch.cuda.synchronize()
for (shape, site), n in cnt.most_common(40):
    print(n, shape, site)
