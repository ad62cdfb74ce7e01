"""What the non-kernel parts of the training step cost: time the step with one component switched off at a time."""
import sys, os, functools, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animeface_amd.implementations.StyleGAN2 import utils as U, conv as C, model as M
from animeface_amd import nnutils
from animeface_amd.nnutils import sample_nnoise, update_ema
dev = torch.device('cuda')


def build():
    torch.manual_seed(0)
    G, G_ema, D = M.Generator(256).to(dev), M.Generator(256).to(dev), M.Discriminator(256).to(dev)
    G.init_weight(functools.partial(M.init_weight_N01, lr=0.01), M.init_weight_N01); D.apply(M.init_weight_N01); update_ema(G, G_ema, decay=0)
    oG, oD = U.build_optimizers(G, D, 0.001, (0., 0.99), 10., 0., 16, 8)
    return U.TrainStep(G, G_ema, D, oG, oD, 10., 0., 16, 8, 'color,translation', 512, functools.partial(sample_nnoise, device=dev))


def timeit(step, n=12):
    real = torch.rand(64, 3, 256, 256, device=dev) * 2 - 1
    for _ in range(3): step(real)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step(real)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


step = build()
base = timeit(step)
print(f'baseline (no R1 in window)      {base:7.2f} ms')
orig = U.update_ema
U.update_ema = lambda *a, **k: None
print(f'without the EMA update          {timeit(step):7.2f} ms')
U.update_ema = orig
so_g, so_d = step.optimizer_G.step, step.optimizer_D.step
step.optimizer_G.step = lambda *a, **k: None
step.optimizer_D.step = lambda *a, **k: None
print(f'without the optimizer steps     {timeit(step):7.2f} ms')
step.optimizer_G.step, step.optimizer_D.step = so_g, so_d
smp = step.sampler
z_fixed = smp((64, 512))
step.sampler = lambda size: z_fixed
print(f'with a fixed latent batch       {timeit(step):7.2f} ms')
step.sampler = smp
# cost of the ~180 zero-fill launches per step (dw, reduction buffers): replace torch.zeros by torch.empty (results become garbage; timing only)
_z = torch.zeros
torch.zeros = lambda *a, **k: torch.empty(*a, **k)
print(f'without zero fills (timing only)  {timeit(step):7.2f} ms')
torch.zeros = _z
