"""Where do the small ATen ops of one SG2 step come from?  (python call sites of aten::copy_/mul/sum/add_/fill_/mm)"""
import sys, os, functools, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from animeface_amd.implementations.StyleGAN2 import utils as U, model as M
from animeface_amd.nnutils import sample_nnoise, update_ema
dev = torch.device('cuda')
torch.manual_seed(0)
G, G_ema, D = M.Generator(256).to(dev), M.Generator(256).to(dev), M.Discriminator(256).to(dev)
G.init_weight(functools.partial(M.init_weight_N01, lr=0.01), M.init_weight_N01); D.apply(M.init_weight_N01); update_ema(G, G_ema, decay=0)
oG, oD = U.build_optimizers(G, D, 0.001, (0., 0.99), 10., 0., 16, 8)
step = U.TrainStep(G, G_ema, D, oG, oD, 10., 0., 16, 8, 'color,translation', 512, functools.partial(sample_nnoise, device=dev))
real = torch.rand(64, 3, 256, 256, device=dev) * 2 - 1
for _ in range(3): step(real)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step(real)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::copy_', 'aten::mul', 'aten::sum', 'aten::add_', 'aten::fill_', 'aten::mm', 'aten::addmm', 'aten::add', 'aten::zero_', 'aten::empty', 'aten::to', 'aten::_to_copy', 'aten::contiguous', 'aten::clone'):
        st = [s for s in ev.stack if 'animeface_amd' in s or 'bench' in s or 'tools/' in s]
        site = st[0] if st else (ev.stack[0] if ev.stack else '?')
        cnt[(ev.name, site.split('/root/repo/')[-1][:110])] += 1
for (name, site), n in cnt.most_common(60):
    print(f'{n:5d}  {name:16s} {site}')
