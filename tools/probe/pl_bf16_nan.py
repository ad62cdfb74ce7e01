"""Which gradients of the bf16 path-length penalty are not finite, and from which op the first NaN comes (anomaly mode)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np, torch
from test_hip_sg2 import build, sub, ReplayNoise, relerr, t
from animeface_amd.implementations.StyleGAN2 import utils as U
from animeface_amd import rng
g = dict(np.load(os.path.join(os.path.dirname(__file__), '..', '..', 'tests', 'golden', 'sg2_train.npz')))
M, G, D = build(torch.bfloat16)
G.load_state_dict(sub(g, 'G0/'))
G.set_fused_epilogue(False)
with ReplayNoise(M, [t(g[f'pl_noise{i}']) for i in range(4)]):
    fake, style = G(t(g['pl_z']).to('cuda'))
with rng.cpu_stream():
    torch.manual_seed(11)
    pl = U.pl_penalty(style, fake, 0.3, None)
print('pl', pl.item(), float(g['pl']))
names = [k[len('plgrad/'):] for k in g if k.startswith('plgrad/')]
pg = dict(G.named_parameters())
if os.environ.get('ANOMALY') == '1':
    torch.autograd.set_detect_anomaly(True)
grads = torch.autograd.grad(pl, [pg[k] for k in names], allow_unused=True)
for k, gr in zip(names, grads):
    if gr is None:
        print(k, 'None'); continue
    bad = (~torch.isfinite(gr)).sum().item()
    print('%-50s nonfinite %6d / %6d  relerr %s' % (k, bad, gr.numel(), relerr(gr, t(g['plgrad/' + k])) if bad == 0 else '-'))
