"""Which python call sites make real tensor copies (to / contiguous / float / clone / add / mul on big tensors) during one SG2 step,
with bytes moved.  Backward functions run in the autograd thread: the monkey patches are global, so they are seen too."""
import sys, os, functools, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from animeface_amd.implementations.StyleGAN2 import utils as U, model as M
from animeface_amd.nnutils import sample_nnoise, update_ema
dev = torch.device('cuda')
torch.manual_seed(0)
G, G_ema, D = M.Generator(256).to(dev), M.Generator(256).to(dev), M.Discriminator(256).to(dev)
G.init_weight(functools.partial(M.init_weight_N01, lr=0.01), M.init_weight_N01); D.apply(M.init_weight_N01); update_ema(G, G_ema, decay=0)
oG, oD = U.build_optimizers(G, D, 0.001, (0., 0.99), 10., 0., 16, 8)
step = U.TrainStep(G, G_ema, D, oG, oD, 10., 0., 16, 8, 'color,translation', 512, functools.partial(sample_nnoise, device=dev))
real = torch.rand(64, 3, 256, 256, device=dev) * 2 - 1
for _ in range(2): step(real)
cnt = collections.Counter(); byt = collections.Counter()
def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if 'animeface_amd' in fr.filename:
            return f'{fr.filename.split("animeface_amd/")[-1]}:{fr.lineno}'
    return '?'
def wrap(name):
    orig = getattr(torch.Tensor, name)
    def f(self, *a, **k):
        out = orig(self, *a, **k)
        if isinstance(out, torch.Tensor) and out.is_cuda and out.numel() > 1 << 16 and (out.data_ptr() != self.data_ptr()):
            s = site(); cnt[(name, s)] += 1; byt[(name, s)] += out.numel() * out.element_size() + self.numel() * self.element_size()
        return out
    setattr(torch.Tensor, name, f)
for n in ['to', 'contiguous', 'float', 'clone', '__mul__', '__rmul__', '__add__', '__radd__', '__truediv__', '__sub__', 'sum', 'square', 'bfloat16', 'repeat', 'expand', 'reshape']:
    wrap(n)
orig_cat = torch.cat
def cat(ts, *a, **k):
    out = orig_cat(ts, *a, **k)
    if out.is_cuda and out.numel() > 1 << 16:
        s = site(); cnt[('cat', s)] += 1; byt[('cat', s)] += 2 * out.numel() * out.element_size()
    return out
torch.cat = cat
step(real)
torch.cuda.synchronize()
tot = sum(byt.values())
print('total GB moved by these sites: %.2f' % (tot / 1e9))
for k, b in byt.most_common(40):
    print(f'{b / 1e6:9.1f} MB  x{cnt[k]:3d}  {k[0]:12s} {k[1]}')
