"""Deterministic mode under GPU contention: the D and G half-steps of the 256 x 256 networks (batch 4, bf16, agf_set_deterministic) run several
times in one process -- alone, then while a second process keeps the GPU busy -- and the parameter gradients are compared bit for bit.  A kernel
whose result depends on timing (a latent race, an unordered scratch buffer) shows up as the first parameter whose gradient differs.
    python tools/probe/contended_determinism.py [repeats]"""
import os, sys, time, functools
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('AGF_DP_TEST_FULL', '1')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import torch.multiprocessing as mp


def hammer_same(stop):
    """the same half-steps in a second process (what a second rank on the same GPU does)"""
    import test_hip_dp as T
    from animeface_amd import _lib
    from animeface_amd.implementations.StyleGAN2 import utils as U
    from animeface_amd.nnutils import sample_nnoise
    dev = torch.device('cuda', 0)
    _lib.set_deterministic(True)
    G, G_ema, D, opt_G, opt_D = T._build(dev, torch.bfloat16)
    step = U.TrainStep(G, G_ema, D, opt_G, opt_D, 10., 0., T.D_K, 8, 'color,translation', T.CFG['style_dim'], functools.partial(sample_nnoise, device=dev))
    real = T._shard(1, dev)
    while not stop.is_set():
        grads_of(step, real, (G, D), 'G', 1)
        grads_of(step, real, (G, D), 'D', 1)


def hammer(stop):
    dev = torch.device('cuda', 0)
    a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
    x = torch.randn(64, 64, 256, 256, device=dev, dtype=torch.bfloat16)
    while not stop.is_set():
        for _ in range(20):
            (a @ a).sum()
            x.mul_(1.0001)
        torch.cuda.synchronize()


def grads_of(step, real, nets, half, it):
    from animeface_amd import rng
    from animeface_amd.implementations.StyleGAN2.conv import cached_weights, ZeroArena, zero_arena
    G, D = nets
    for p in list(G.parameters()) + list(D.parameters()):
        p.grad = None
    with rng.cpu_stream():
        torch.manual_seed(1234)
        with cached_weights(), zero_arena(ZeroArena(), real.device):
            if half == 'D':
                loss = step._d_half(real, it)
            else:
                for p in D.parameters():
                    p.requires_grad_(False)
                loss = step._g_half(real, it)[0]
                for p in D.parameters():
                    p.requires_grad_(True)
    torch.cuda.synchronize()
    net = D if half == 'D' else G
    return float(loss), {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}


CALLS = []


def spy_on(conv_mod):
    """record checksums of the inputs and outputs of the reduction passes of one half-step"""
    def wrap(name):
        orig = getattr(conv_mod, name)

        def f(*a, **k):
            out = orig(*a, **k)
            def cs(t):
                return None if not isinstance(t, torch.Tensor) or t.device.type == 'meta' else (float(t.double().sum()), float(t.double().abs().sum()))
            CALLS.append((name, [cs(t) for t in a], [cs(t) for t in (out if isinstance(out, tuple) else (out,))]))
            return out
        setattr(conv_mod, name, f)
    for n in ('act_bwd_reduce_pooled_mask_raw', 'act_bwd_reduce_pooled_raw', 'act_bwd_reduce_raw', 'act_bwd_reduce_scaled_raw', 'demod_grad_finish_raw',
              'scale_dot_raw', 'channel_sum_raw'):
        if hasattr(conv_mod, n):
            wrap(n)


if __name__ == '__main__':
    import test_hip_dp as T
    from animeface_amd import _lib
    from animeface_amd.implementations.StyleGAN2 import utils as U
    from animeface_amd.nnutils import sample_nnoise
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device('cuda', 0)
    _lib.set_deterministic(True)
    G, G_ema, D, opt_G, opt_D = T._build(dev, torch.bfloat16)
    step = U.TrainStep(G, G_ema, D, opt_G, opt_D, 10., 0., T.D_K, 8, 'color,translation', T.CFG['style_dim'], functools.partial(sample_nnoise, device=dev))
    real = T._shard(0, dev)
    from animeface_amd.implementations.StyleGAN2 import conv as C
    if os.environ.get('SPY'):
        spy_on(C)
    for half, it in (('D', 1),) if os.environ.get('SPY') else (('G', 1), ('D', 1)):
        CALLS.clear()
        ref = grads_of(step, real, (G, D), half, it)
        ref_calls = list(CALLS)
        for contended in (False, True):
            stop = proc = None
            if contended:
                ctx = mp.get_context('spawn')
                stop = ctx.Event()
                proc = ctx.Process(target=hammer_same if os.environ.get('SAME') else hammer, args=(stop,))
                proc.start()
                time.sleep(25 if os.environ.get('SAME') else 8)
            for r in range(reps):
                CALLS.clear()
                cur = grads_of(step, real, (G, D), half, it)
                if os.environ.get('SPY'):
                    for i, (ca, cb) in enumerate(zip(CALLS, ref_calls)):
                        if ca != cb:
                            print(f'   first reduction call that differs: #{i} {ca[0]}: inputs equal {ca[1] == cb[1]}, outputs equal {[x == y for x, y in zip(ca[2], cb[2])]}', flush=True)
                            print(f'      {ca[2]}\n      {cb[2]}', flush=True)
                            break
                bad = [n for n in ref[1] if not torch.equal(ref[1][n], cur[1][n])]
                print(f'{half}-half (iteration kind {it}) {"contended" if contended else "alone    "} run {r}: loss {cur[0]:.9f} (ref {ref[0]:.9f}); '
                      f'{len(bad)} of {len(ref[1])} gradients differ' + (f', first: {bad[:6]}, last: {bad[-3:]}' if bad else ''), flush=True)
            if proc is not None:
                stop.set(); proc.join()
