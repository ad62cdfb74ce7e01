"""End-to-end health check: a few hundred bf16 iterations on synthetic 'images' (smooth random blobs), losses stay finite and move."""
import sys, os, functools, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animeface_amd.implementations.StyleGAN2 import utils as U, model as M
from animeface_amd.nnutils import sample_nnoise, update_ema
dev = torch.device('cuda')
torch.manual_seed(0)
S, B, iters = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 32, int(sys.argv[2]) if len(sys.argv) > 2 else 300
G, G_ema, D = M.Generator(S).to(dev), M.Generator(S).to(dev), M.Discriminator(S).to(dev)
G.init_weight(functools.partial(M.init_weight_N01, lr=0.01), M.init_weight_N01); D.apply(M.init_weight_N01); G_ema.eval(); update_ema(G, G_ema, decay=0)
oG, oD = U.build_optimizers(G, D, 0.001, (0., 0.99), 10., 0., 16, 8)
# a toy data distribution: low-frequency colour blobs in [-1, 1]
def batch():
    z = torch.randn(B, 3, 4, 4, device=dev)
    return torch.tanh(torch.nn.functional.interpolate(z, size=(S, S), mode='bicubic', align_corners=False))
data = [batch() for _ in range(8)]
t0 = time.time()
hist = U.train(iters, data, functools.partial(sample_nnoise, device=dev), sample_nnoise((4, 512), dev), 512, G, G_ema, D, oG, oD,
               10., 0., 16, 8, 'color,translation', dev, True, save=10 ** 9, log_every=25)
torch.cuda.synchronize()
print('%.1f s for %d iterations' % (time.time() - t0, iters))
for it, d, g in hist:
    print(f'it {it:4d}  D {d:8.4f}  G {g:8.4f}')
ok = all(torch.isfinite(p).all().item() for p in list(G.parameters()) + list(D.parameters()) + list(G_ema.parameters()))
print('all parameters finite:', ok)
with torch.no_grad():
    img, _ = G_ema(sample_nnoise((8, 512), dev))
print('G_ema image range %.3f .. %.3f, mean abs %.3f' % (img.min().item(), img.max().item(), img.abs().mean().item()))
