"""End-to-end health check of the StyleGAN3 / ADA trainers: a few hundred bf16 iterations on synthetic blobs stay finite."""
import sys, os, functools, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animeface_amd.implementations.StyleGAN3 import utils as U, model as M
from animeface_amd.implementations.ADA.model import ADA
from animeface_amd.nnutils import update_ema, freeze, sample_nnoise
from animeface_amd.thirdparty.diffaugment import DiffAugment
dev = torch.device('cuda')
torch.manual_seed(0)
S, B, iters = 64, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 200
use_ada = len(sys.argv) > 2 and sys.argv[2] == 'ada'
G, G_ema = M.Generator(S, 512).to(dev), M.Generator(S, 512).to(dev)
freeze(G_ema); update_ema(G, G_ema, 0., copy_buffers=True)
D = M.Discriminator(S, 3, 32, 512).to(dev)
D(G(sample_nnoise((4, 512), dev)))
oG, oD = U.build_optimizers(G, D, 0.0025, 0.01, (0., 0.99))
aug = ADA(4, 0.5, 0.6, B, xflip=1, rotate90=1, xint=1, scale=1, rotate=1, aniso=1, xfrac=1, brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1).to(dev) \
    if use_ada else functools.partial(DiffAugment, policy='color,translation')
def batch():
    z = torch.randn(B, 3, 4, 4, device=dev)
    return torch.tanh(torch.nn.functional.interpolate(z, size=(S, S), mode='bicubic', align_corners=False))
data = [batch() for _ in range(8)]
t0 = time.time()
hist = U.train(iters, data, 512, sample_nnoise((4, 512), dev), G, G_ema, D, oG, oD, 3., 16, aug, dev, True, save=10 ** 9, log_every=25)
torch.cuda.synchronize()
print('%.1f s for %d iterations%s' % (time.time() - t0, iters, ' (ADA p = %.3f)' % float(aug.p) if use_ada else ''))
for it, d, g in hist:
    print(f'it {it:4d}  D {d:8.4f}  G {g:8.4f}')
print('all parameters finite:', all(torch.isfinite(p).all().item() for p in list(G.parameters()) + list(D.parameters()) + list(G_ema.parameters())))
