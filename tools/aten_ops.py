"""ATen operators (with input shapes and the innermost animeface_amd source line) launched by one eager StyleGAN2 training iteration."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(real)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith('aten::'):
        continue
    kt = sum(k.duration for k in ev.kernels) if ev.kernels else 0
    if not ev.kernels:
        continue
    site = '?'
    for fr in (ev.stack or []):
        if 'animeface_amd' in fr:
            site = fr.split('animeface_amd/')[-1][:60]
            break
    shapes = str([s for s in (ev.input_shapes or []) if s])[:70]
    a = agg[(ev.name, shapes, site)]
    a[0] += len(ev.kernels); a[1] += kt
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print('aten kernels: %d launches, %.2f ms' % (sum(v[0] for _, v in rows), tot / 1e3))
for (name, shapes, site), (n, t) in rows[:int(os.environ.get('TOP', '70'))]:
    print('%7.1f us %4d  %-28s %-72s %s' % (t, n, name, shapes, site))
