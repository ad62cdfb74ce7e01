"""Which Python lines of the package issue the ATen ops of one eager StyleGAN3-T iteration (see tools/aten_sites.py): a TorchDispatchMode
records every dispatched non-view op with the innermost package frame; backward ops are attributed to the line that called backward(), split
by the autograd node that is running (torch.autograd.graph hooks are not needed: the node's name is in the Python stack for custom Functions,
native nodes show up as the op itself).    SIZE=512 BATCH=16 python tools/aten_sites_sg3.py [--top 120]"""
import sys, os, functools, collections, argparse, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from animeface_amd.implementations.StyleGAN3 import utils as U, model as M
from animeface_amd.nnutils import update_ema, freeze
from animeface_amd.thirdparty.diffaugment import DiffAugment
ap = argparse.ArgumentParser()
ap.add_argument('--top', type=int, default=120)
args = ap.parse_args()
dev = torch.device('cuda')
torch.manual_seed(0)
S, B = int(os.environ.get('SIZE', '512')), int(os.environ.get('BATCH', '16'))
G = M.Generator(S, 512, compute_dtype=torch.bfloat16).to(dev); G_ema = M.Generator(S, 512, compute_dtype=torch.bfloat16).to(dev)
freeze(G_ema); update_ema(G, G_ema, 0., copy_buffers=True)
D = M.Discriminator(S, 3, 32, 512, compute_dtype=torch.bfloat16).to(dev)
oG, oD = U.build_optimizers(G, D, 0.0025, 0.01, (0., 0.99))
step = U.TrainStep(G, G_ema, D, oG, oD, 3., 16, functools.partial(DiffAugment, policy='color,translation'), 512)
real = torch.rand(B, 3, S, S, device=dev) * 2 - 1
for _ in range(2): step(real)
torch.cuda.synchronize()
VIEWS = {'view', 'reshape', 'as_strided', 'detach', 'alias', 'expand', 'permute', 'transpose', 't', 'slice', 'select', 'unsqueeze', 'squeeze', '_unsafe_view', 'unbind',
         'split', 'split_with_sizes', 'narrow', 'empty', 'empty_like', 'empty_strided', 'new_empty', 'new_empty_strided', 'lift_fresh', '_reshape_alias', 'unfold',
         'sym_size', 'sym_stride', 'sym_numel', 'is_same_size', 'view_as', 'chunk', 'diagonal', 'resize_', 'set_', 'unsafe_split', 'unsafe_chunk', 'movedim', 'result_type',
         'is_nonzero', '_local_scalar_dense', 'item', 'record_stream', 'is_pinned', 'prim.device', 'device', 'dim', 'stride', 'size', 'numel', 'is_contiguous', 'storage_offset',
         'sym_storage_offset', 'conj', '_has_compatible_shallow_copy_type', 'unflatten', 'flatten', '_record_function_enter_new', '_record_function_exit'}
cnt, els, names = collections.Counter(), collections.Counter(), collections.defaultdict(collections.Counter)


class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, a=(), kw=None):
        out = func(*a, **(kw or {}))
        name = func.__name__.split('.')[0]
        if name in VIEWS:
            return out
        fr = [f for f in traceback.extract_stack() if 'animeface_amd' in f.filename]
        site = f'{fr[-1].filename.split("animeface_amd/")[-1]}:{fr[-1].lineno} {fr[-1].name}' if fr else 'outside the package'
        node = torch._C._current_autograd_node() if hasattr(torch._C, '_current_autograd_node') else None
        if node is not None and fr and 'backward(' in (fr[-1].line or ''):
            site += f'  <- {node.name()}'
        o = out[0] if isinstance(out, (tuple, list)) and out else out
        n = o.numel() if isinstance(o, torch.Tensor) else 0
        cnt[site] += 1; els[site] += n; names[site][name] += 1
        return out


with torch.autograd.set_multithreading_enabled(False), Rec():
    step(real)
torch.cuda.synchronize()
print(f'dispatched non-view ATen ops: {sum(cnt.values())}, output elements {sum(els.values()) / 1e6:.1f} M')
for site, n_ in cnt.most_common(args.top):
    print(f'{n_:4d} ops {els[site] / 1e6:9.2f} Mel  {site}   [{", ".join(f"{k}x{v}" for k, v in names[site].most_common(8))}]')
