"""ATen operators (with input shapes) launched by one eager StyleGAN3-T 512x512 training iteration."""
import sys, os, functools, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from animeface_amd.implementations.StyleGAN3 import utils as U, model as M
from animeface_amd.nnutils import update_ema, freeze
from animeface_amd.thirdparty.diffaugment import DiffAugment
dev = torch.device('cuda')
torch.manual_seed(0)
S, B = int(os.environ.get('SIZE', '512')), int(os.environ.get('BATCH', '16'))
G = M.Generator(S, 512, compute_dtype=torch.bfloat16).to(dev); G_ema = M.Generator(S, 512, compute_dtype=torch.bfloat16).to(dev)
freeze(G_ema); update_ema(G, G_ema, 0., copy_buffers=True)
D = M.Discriminator(S, 3, 32, 512, compute_dtype=torch.bfloat16).to(dev)
oG, oD = U.build_optimizers(G, D, 0.0025, 0.01, (0., 0.99))
step = U.TrainStep(G, G_ema, D, oG, oD, 3., 16, functools.partial(DiffAugment, policy='color,translation'), 512)
real = torch.rand(B, 3, S, S, device=dev) * 2 - 1
for _ in range(2): step(real)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(real); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith('aten::') or not ev.kernels:
        continue
    a = agg[(ev.name, str([s for s in (ev.input_shapes or []) if s])[:80])]
    a[0] += len(ev.kernels); a[1] += sum(k.duration for k in ev.kernels)
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print('aten kernels: %d launches, %.2f ms' % (sum(v[0] for _, v in rows), sum(v[1] for _, v in rows) / 1e3))
for (name, shapes), (n, t) in rows[:40]:
    print('%8.1f us %4d  %-28s %s' % (t, n, name, shapes))
