"""Find reads of uninitialised memory: every torch.empty / empty_like / new_empty of a floating tensor is NaN-filled, then the tiny StyleGAN2
training step runs; a NaN that reaches a loss, gradient or weight marks an output some kernel does not fully write (or an input it reads
beyond what was written).   python tools/poison_empty.py [iters]"""
import os, sys, functools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

_empty, _empty_like, _new_empty = torch.empty, torch.empty_like, torch.Tensor.new_empty
def _poison(t):
    if t.is_floating_point() and t.is_cuda and t.numel():
        t.fill_(float('nan'))
    return t
torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))
torch.Tensor.new_empty = lambda self, *a, **k: _poison(_new_empty(self, *a, **k))

from animeface_amd.implementations.StyleGAN2 import model as M, utils as U, conv as C
from animeface_amd.nnutils import sample_nnoise, update_ema
dev = torch.device('cuda')
cfg = dict(image_size=32, style_dim=64, channels=8, max_channels=64)
torch.manual_seed(0)
mk = lambda: M.Generator(cfg['image_size'], 3, cfg['style_dim'], cfg['channels'], cfg['max_channels'], 2, 2, True, 0.01)
G, G_ema = mk().to(dev), mk().to(dev)
D = M.Discriminator(cfg['image_size'], 3, cfg['channels'], cfg['max_channels'], 2, 4).to(dev)
G.init_weight(functools.partial(M.init_weight_N01, lr=0.01), M.init_weight_N01); D.apply(M.init_weight_N01); update_ema(G, G_ema, decay=0)
oG, oD = U.build_optimizers(G, D, 0.001, (0., 0.99), 10., 0., 2, 8)
step = U.TrainStep(G, G_ema, D, oG, oD, 10., 0., 2, 8, 'color,translation', cfg['style_dim'], functools.partial(sample_nnoise, device=dev))
real = torch.rand(int(os.environ.get('B', '4')), 3, 32, 32, device=dev) * 2 - 1

# report the first custom op whose output contains NaN
def wrap(mod, name):
    fn = getattr(mod, name)
    def w(*a, **k):
        out = fn(*a, **k)
        outs = out if isinstance(out, (tuple, list)) else (out,)
        flat = []
        for o in outs:
            if isinstance(o, (tuple, list)): flat += list(o)
            else: flat.append(o)
        for i, o in enumerate(flat):
            if isinstance(o, torch.Tensor) and o.is_floating_point() and torch.isnan(o).any():
                shapes = [tuple(x.shape) for x in a if isinstance(x, torch.Tensor)]
                print(f'NaN in output {i} of {name}: out shape {tuple(o.shape)}, nan frac {float(torch.isnan(o).float().mean()):.4f}, inputs {shapes}, kwargs {[k2 for k2, v in k.items() if v is not None]}', flush=True)
        return out
    setattr(mod, name, w)
for n in ['conv2d_fwd_raw', 'conv2d_wgrad_raw', 'act_bwd_reduce_raw', 'act_bwd_reduce_pooled_raw', 'scale_dot_raw', 'prep_weights_raw']:
    wrap(C, n)
from animeface_amd.stylegan3_ops import upfirdn2d as UF
wrap(UF, '_launch')

for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    dl, gl, fake = step(real)
    torch.cuda.synchronize()
    bad = [n for n, p in list(G.named_parameters()) + list(D.named_parameters()) if torch.isnan(p).any()]
    print('iteration', it, 'losses', float(dl), float(gl), 'params with NaN:', bad[:8], flush=True)
