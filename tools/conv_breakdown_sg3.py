"""Per-layer-shape breakdown of the MFMA conv launches of one StyleGAN3 training iteration (HIP-event timed)."""
import sys, os, functools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animeface_amd.implementations.StyleGAN3 import utils as U, model as M
from animeface_amd.implementations.StyleGAN2 import conv as C
from animeface_amd.nnutils import update_ema, freeze
from animeface_amd.thirdparty.diffaugment import DiffAugment
dev = torch.device('cuda')
torch.manual_seed(0)
SIZE, B = int(os.environ.get('SIZE', '256')), int(os.environ.get('B', '32'))      # SIZE=512 B=16: BASELINE configs[3]
G, G_ema, D = M.Generator(SIZE, 512).to(dev), M.Generator(SIZE, 512).to(dev), M.Discriminator(SIZE, 3, 32, 512).to(dev)
freeze(G_ema); update_ema(G, G_ema, 0., copy_buffers=True)
oG, oD = U.build_optimizers(G, D, 0.0025, 0.01, (0., 0.99))
step = U.TrainStep(G, G_ema, D, oG, oD, 3., 16, functools.partial(DiffAugment, policy='color,translation'), 512)
real = torch.rand(B, 3, SIZE, SIZE, device=dev) * 2 - 1
for _ in range(2): step(real)
t = C.KernelTimer(); C.KernelTimer.active = t
for _ in range(4): step(real)
C.KernelTimer.active = None
torch.cuda.synchronize()
rows = []
for key, recs in t.by_shape.items():
    ms = sum(a.elapsed_time(b) for a, b, _ in recs) / 4
    fl = sum(f for _, _, f in recs) / 4
    rows.append((ms, key, len(recs) / 4, fl / (ms * 1e-3) / 1e12))
rows.sort(reverse=True)
print('total conv ms/iter', round(sum(r[0] for r in rows), 2))
for ms, key, n, tf in rows[:36]:
    print(f'{ms:6.2f} ms  x{n:4.1f}  {tf:7.1f} TF/s  {key}')
