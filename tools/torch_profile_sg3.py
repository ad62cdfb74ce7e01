"""torch.profiler summary of one SG3 training iteration: which ATen ops (the glue around the HIP kernels) cost what."""
import sys, os, functools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from animeface_amd.implementations.StyleGAN3 import utils as U, model as M
from animeface_amd.nnutils import update_ema, freeze
from animeface_amd.thirdparty.diffaugment import DiffAugment
dev = torch.device('cuda')
torch.manual_seed(0)
G, G_ema, D = M.Generator(256, 512).to(dev), M.Generator(256, 512).to(dev), M.Discriminator(256, 3, 32, 512).to(dev)
freeze(G_ema); update_ema(G, G_ema, 0., copy_buffers=True)
oG, oD = U.build_optimizers(G, D, 0.0025, 0.01, (0., 0.99))
step = U.TrainStep(G, G_ema, D, oG, oD, 3., 16, functools.partial(DiffAugment, policy='color,translation'), 512)
real = torch.rand(32, 3, 256, 256, device=dev) * 2 - 1
for _ in range(3): step(real)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(2): step(real)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.key.startswith('aten::') or 'Backward' in e.key or e.key[0].isupper() or e.key.startswith('_')]
rows.sort(key=lambda e: -e.self_device_time_total)
for e in rows[:40]:
    print(f'{e.key[:50]:50s} calls {e.count:5d}  self_gpu_ms {e.self_device_time_total / 1e3 / 2:8.2f} per iter')
