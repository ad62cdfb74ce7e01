"""Which Python lines of the package issue the ATen ops of one eager StyleGAN2 iteration (a TorchDispatchMode records every dispatched op
with the innermost package frame; the backward pass runs on the calling thread so that its ops are seen too -- they are attributed to
the line that called backward()).    python tools/aten_sites.py [--augment ada] [--top 90]"""
import sys, os, functools, collections, argparse, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from animeface_amd.implementations.StyleGAN2 import utils as U, model as M
from animeface_amd.nnutils import sample_nnoise, update_ema
ap = argparse.ArgumentParser()
ap.add_argument('--augment', default='color,translation')
ap.add_argument('--top', type=int, default=90)
args = ap.parse_args()
dev = torch.device('cuda')
torch.manual_seed(0)
G, G_ema, D = M.Generator(256).to(dev), M.Generator(256).to(dev), M.Discriminator(256).to(dev)
G.init_weight(functools.partial(M.init_weight_N01, lr=0.01), M.init_weight_N01); D.apply(M.init_weight_N01); update_ema(G, G_ema, decay=0)
oG, oD = U.build_optimizers(G, D, 0.001, (0., 0.99), 10., 0., 16, 8, capturable=True)
step = U.TrainStep(G, G_ema, D, oG, oD, 10., 0., 16, 8, args.augment, 512, functools.partial(sample_nnoise, device=dev))
real = torch.rand(64, 3, 256, 256, device=dev) * 2 - 1
for _ in range(3): step(real)
if step.ada is not None:
    step.ada.p.fill_(0.3)
torch.cuda.synchronize()
VIEWS = {'view', 'reshape', 'as_strided', 'detach', 'alias', 'expand', 'permute', 'transpose', 't', 'slice', 'select', 'unsqueeze', 'squeeze', '_unsafe_view', 'unbind',
         'split', 'split_with_sizes', 'narrow', 'empty', 'empty_like', 'empty_strided', 'new_empty', 'new_empty_strided', 'lift_fresh', '_reshape_alias', 'unfold',
         'sym_size', 'sym_stride', 'sym_numel', 'is_same_size', 'view_as', 'chunk', 'diagonal', 'resize_', 'set_', 'unsafe_split', 'unsafe_chunk', 'movedim', 'result_type',
         'is_nonzero', '_local_scalar_dense', 'item', 'record_stream', 'is_pinned', 'prim.device', 'device', 'dim', 'stride', 'size', 'numel', 'is_contiguous', 'storage_offset',
         'sym_storage_offset', 'conj', '_has_compatible_shallow_copy_type', 'unflatten', 'flatten'}
cnt, els, names = collections.Counter(), collections.Counter(), collections.defaultdict(collections.Counter)


class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, a=(), kw=None):
        out = func(*a, **(kw or {}))
        name = func.__name__.split('.')[0]
        if name in VIEWS:
            return out
        fr = [f for f in traceback.extract_stack() if 'animeface_amd' in f.filename]
        site = f'{fr[-1].filename.split("animeface_amd/")[-1]}:{fr[-1].lineno} {fr[-1].name}' if fr else 'outside the package'
        if fr and len(fr) > 1 and ('backward' in fr[-1].line or ''):
            pass
        o = out[0] if isinstance(out, (tuple, list)) and out else out
        n = o.numel() if isinstance(o, torch.Tensor) else 0
        cnt[site] += 1; els[site] += n; names[site][name] += 1
        return out


with torch.autograd.set_multithreading_enabled(False), Rec():
    step(real)
torch.cuda.synchronize()
print(f'dispatched non-view ATen ops: {sum(cnt.values())}, output elements {sum(els.values()) / 1e6:.1f} M')
for site, n_ in cnt.most_common(args.top):
    print(f'{n_:4d} ops {els[site] / 1e6:9.2f} Mel  {site}   [{", ".join(f"{k}x{v}" for k, v in names[site].most_common(6))}]')
