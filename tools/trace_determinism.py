"""Trace every raw op call (name, optional-argument set, checksums of tensor inputs and outputs) of the two-shard emulation's first
iteration in two runs and print the first call where the traces part: a different op order, or equal inputs with a different output."""
import os, sys, functools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import test_hip_dp as T
from animeface_amd import rng
from animeface_amd.implementations.StyleGAN2 import utils as U, conv as C
from animeface_amd.implementations.StyleGAN2.conv import cached_weights, invalidate_cached, ZeroArena, zero_arena
from animeface_amd.stylegan3_ops import upfirdn2d as UF
from animeface_amd.thirdparty import diffaugment as DA
from animeface_amd.nnutils import sample_nnoise, update_ema
dev = torch.device('cuda', 0)
TRACE = None

def cs(t):
    if not isinstance(t, torch.Tensor) or t.numel() == 0: return None
    f = t.detach().float()
    return (tuple(t.shape), float(f.sum()), float(f.abs().sum()), float((f * torch.arange(f.numel(), device=f.device).view_as(f).remainder(7)).sum()))

def flat(o):
    if isinstance(o, (tuple, list)):
        r = []
        for x in o: r += flat(x)
        return r
    return [o]

def wrap(mod, name):
    fn = getattr(mod, name)
    def w(*a, **k):
        ins = [cs(x) for x in flat(list(a) + [k[key] for key in sorted(k)])]
        out = fn(*a, **k)
        if TRACE is not None:
            TRACE.append((name, tuple(key for key in sorted(k) if k[key] is not None), ins, [cs(x) for x in flat(out)]))
        return out
    setattr(mod, name, w)
for n in ['conv2d_fwd_raw', 'conv2d_wgrad_raw', 'act_bwd_reduce_raw', 'act_bwd_reduce_pooled_raw', 'scale_dot_raw', 'prep_weights_raw']:
    wrap(C, n)
wrap(UF, '_launch')
for n in [x for x in dir(DA) if x.startswith('_fused') or x in ('DiffAugment',)]:
    if callable(getattr(DA, n)): wrap(DA, n)

def run():
    global TRACE
    G, G_ema, D, opt_G, opt_D = T._build(dev)
    step = U.TrainStep(G, G_ema, D, opt_G, opt_D, 10., 0., T.D_K, 8, 'color,translation', T.CFG['style_dim'], functools.partial(sample_nnoise, device=dev))
    reals = [T._shard(r, dev) for r in range(2)]
    streams = []
    for r in range(2):
        torch.manual_seed(1000 + r); streams.append(torch.get_rng_state())
    TRACE = []
    with rng.cpu_stream():
        it = 0
        opt_G.zero_grad(set_to_none=True); opt_D.zero_grad(set_to_none=True)
        with cached_weights():
            for r in range(2):
                torch.set_rng_state(streams[r])
                TRACE.append((f'--- D-half shard {r}', (), [], []))
                with zero_arena(ZeroArena(), dev): step._d_half(reals[r], it)
                streams[r] = torch.get_rng_state()
            for p in D.parameters():
                if p.grad is not None: p.grad.div_(2)
            opt_D.step(); invalidate_cached(D.parameters())
            for p in D.parameters(): p.requires_grad_(False)
            for r in range(2):
                torch.set_rng_state(streams[r])
                TRACE.append((f'--- G-half shard {r}', (), [], []))
                with zero_arena(ZeroArena(), dev): step._g_half(reals[r], it)
                streams[r] = torch.get_rng_state()
    tr, TRACE = TRACE, None
    gr = {n: p.grad.clone() for n, p in G.named_parameters() if p.grad is not None}
    return tr, gr

def close(a, b):
    # checksums equal up to fp32 summation noise (atomics): relative to the tensor's abs-sum
    if a is None or b is None: return a == b
    if a[0] != b[0]: return False
    scale = max(abs(a[2]), 1e-30)
    return all(abs(x - y) <= 2e-5 * scale * (7 if i == 3 else 1) for i, (x, y) in enumerate(zip(a[1:], b[1:]), 1))
def same(A, B):
    return len(A) == len(B) and all(close(x, y) for x, y in zip(A, B))

ref, gref = run()
for trial in range(8):
    cur, gcur = run()
    gd = max(float((gcur[k] - gref[k]).abs().max() / gref[k].abs().max().clamp_min(1e-20)) for k in gref)
    print('trial', trial, 'G grad max rel diff', round(gd, 5), flush=True)
    if gd < 1e-4: continue
    sect, shown = '', 0
    for i, (a, b) in enumerate(zip(ref, cur)):
        if a[0].startswith('---'): sect = a[0]
        if a[0] != b[0] or a[1] != b[1]:
            print(f'  {sect} call {i}: different op {a[0]}{a[1]} vs {b[0]}{b[1]}'); break
        din = [j for j, (x, y) in enumerate(zip(a[2], b[2])) if x != y]
        dout = [j for j, (x, y) in enumerate(zip(a[3], b[3])) if x != y]
        if din or dout:
            shown += 1
            print(f'  {sect} call {i} {a[0]}{a[1]}: inputs differing {din} outputs differing {dout}' +
                  ''.join(f'\n      out{j} {a[3][j]} | {b[3][j]}' for j in dout[:2]) + ''.join(f'\n      in{j} {a[2][j]} | {b[2][j]}' for j in din[:2]))
            if shown >= 12: break
    break
