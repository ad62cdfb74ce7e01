"""Per-layer-shape breakdown of the MFMA conv launches of one training step (HIP-event timed)."""
import sys, os, functools
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
tot = sum(r[0] for r in rows)
print('total conv ms/step', round(tot, 2))
for ms, key, n, tf in rows[:int(os.environ.get("ROWS", "40"))]:
    print(f'{ms:6.2f} ms  x{n:4.1f}  {tf:7.1f} TF/s  {key}')
print('--- by time lost against 1000 TFLOP/s')
for ms, key, n, tf in sorted(rows, key=lambda r: -r[0] * max(0, 1 - r[3] / 1000))[:30]:
    print(f'{ms * max(0, 1 - tf / 1000):6.2f} lost of {ms:5.2f} ms  x{n:4.1f}  {tf:7.1f} TF/s  {key}')
