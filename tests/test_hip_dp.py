"""The data-parallel path with the REAL StyleGAN2 ``TrainStep`` on the GPU: two ranks share GPU 0 (``AGF_SINGLE_DEVICE=1``, gloo), each runs
the actual half-steps -- custom autograd Functions on the HIP kernels, ``ZeroArena`` gradient scratch, D frozen / unfrozen per half-step,
fused Adam on bucket views, ``GradReducer`` hooks, a lazy-R1 (double backward) iteration -- and the result is compared with ONE process
that runs the two shards one after the other, accumulates their gradients and averages them before each optimizer step (the same
arithmetic a 2-GPU run performs; ``MiniBatchStdDev`` groups stay per shard as they stay per GPU).

Two precisions.  fp32: the two runs must agree to fp32 summation noise (weights to 2e-5 after three Adam steps) -- the strict check of
the exchange arithmetic.  bf16 (the benchmarked path, with the fused lrelu-mask / pooled-gradient launches): the per-iteration losses
must agree and the replicas must be bit-identical, but weights are only compared statistically -- fp32 atomics (DiffAugment's means,
split-K partial sums) are summed in a timing-dependent order, a last-bit difference flips bf16 roundings downstream, and Adam with
beta1 = 0 turns a gradient that changes sign near zero into a full +-lr step (tools/trace_determinism.py follows one such chain: a
1e-7 difference in a bias-gradient sum -> D's bias after Adam differs in the 9th digit -> bf16 roundings flip in the next forward -> the
next gradients differ by 0.3 %; the single-process reference itself lands on one of two or three outcomes from run to run)."""
import os
import socket

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

CFG = dict(image_size=32, image_channels=3, style_dim=64, channels=8, max_channels=64, block_num_conv=2, map_num_layers=2)
ITERS, BATCH, D_K = int(os.environ.get('AGF_DP_TEST_ITERS', '3')), 4, 2          # iteration 2 is a lazy-R1 iteration (it % d_k == 0, it != 0)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(dev, dtype=torch.bfloat16):
    import functools
    from animeface_amd.implementations.StyleGAN2 import model as M, utils as U
    from animeface_amd.nnutils import update_ema
    torch.manual_seed(0)
    mk = lambda: M.Generator(CFG['image_size'], 3, CFG['style_dim'], CFG['channels'], CFG['max_channels'], 2, CFG['map_num_layers'], True, 0.01,
                             compute_dtype=dtype)
    G, G_ema = mk().to(dev), mk().to(dev)
    D = M.Discriminator(CFG['image_size'], 3, CFG['channels'], CFG['max_channels'], 2, 4, compute_dtype=dtype).to(dev)
    G.init_weight(functools.partial(M.init_weight_N01, lr=0.01), M.init_weight_N01)
    D.apply(M.init_weight_N01)
    G_ema.eval()
    update_ema(G, G_ema, decay=0)
    opt_G, opt_D = U.build_optimizers(G, D, 0.001, (0., 0.99), 10., 0., D_K, 8)
    return G, G_ema, D, opt_G, opt_D


def _shard(rank, dev):
    g = torch.Generator().manual_seed(50 + rank)
    return (torch.rand(BATCH, 3, CFG['image_size'], CFG['image_size'], generator=g) * 2 - 1).to(dev)


def _worker(rank, world, port, out, dtype=torch.bfloat16):
    import sys
    import functools
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      AGF_SINGLE_DEVICE='1', AGF_DIST_BACKEND='gloo')
    from animeface_amd import distributed as dp, rng
    from animeface_amd.implementations.StyleGAN2 import utils as U
    from animeface_amd.nnutils import sample_nnoise
    r, w, local = dp.init_distributed()
    assert (r, w, local) == (rank, world, 0)
    dev = torch.device('cuda', 0)
    G, G_ema, D, opt_G, opt_D = _build(dev, dtype)
    dp.broadcast_module(G), dp.broadcast_module(G_ema), dp.broadcast_module(D)
    red_G = dp.GradReducer(G.parameters(), bucket_bytes=1 << 18, never_used=dp.never_used_parameters(G))
    red_D = dp.GradReducer(D.parameters(), bucket_bytes=1 << 18)
    assert len(red_G.buckets) >= 3 and len(red_D.buckets) >= 2
    step = U.TrainStep(G, G_ema, D, opt_G, opt_D, 10., 0., D_K, 8, 'color,translation', CFG['style_dim'],
                       functools.partial(sample_nnoise, device=dev), red_G, red_D)
    real = _shard(rank, dev)
    losses = []
    with rng.cpu_stream():                       # every draw from torch's CPU generator: replayable by the single-process run
        torch.manual_seed(1000 + rank)
        for _ in range(ITERS):
            dl, gl, _ = step(real)
            losses.append((float(dl), float(gl)))
    torch.cuda.synchronize()
    for m in (G, G_ema, D):
        dp.check_replica_consistency(m)
    rep_G, rep_D = red_G.overlap_report(), red_D.overlap_report()
    # most buckets must have been launched from the backward hooks (the ones finish() launches hold never-used parameters, or D's last
    # bias on the R1 iteration)
    if red_G.early:
        assert rep_G['buckets_from_hooks'] >= (len(red_G.buckets) - 1) * ITERS, rep_G
        assert rep_D['buckets_from_hooks'] >= (len(red_D.buckets) - 1) * ITERS - 1, rep_D
    # parameters that received no gradient must have been skipped by Adam, as in the single-process loop
    scale = dp.never_used_parameters(G)[0]
    assert not opt_G.state.get(scale), 'Adam stepped a parameter that never received a gradient'
    torch.save(dict(G={k: v.cpu() for k, v in G.state_dict().items()}, D={k: v.cpu() for k, v in D.state_dict().items()},
                    G_ema={k: v.cpu() for k, v in G_ema.state_dict().items()}, losses=losses), f'{out}.{rank}')
    dp.dist.barrier()
    dp.dist.destroy_process_group()


def _single_process(dev, dtype=torch.bfloat16):
    """Both shards in one process: shard r's half-step runs with shard r's random stream, gradients accumulate, are halved, then the
    optimizer steps -- TrainStep.__call__ with the gradient exchange written out."""
    import functools
    from animeface_amd import rng
    from animeface_amd.implementations.StyleGAN2 import utils as U
    from animeface_amd.implementations.StyleGAN2.conv import cached_weights, invalidate_cached, ZeroArena, zero_arena
    from animeface_amd.nnutils import sample_nnoise, update_ema
    G, G_ema, D, opt_G, opt_D = _build(dev, dtype)
    step = U.TrainStep(G, G_ema, D, opt_G, opt_D, 10., 0., D_K, 8, 'color,translation', CFG['style_dim'],
                       functools.partial(sample_nnoise, device=dev))
    NS = int(os.environ.get('AGF_DP_TEST_SHARDS', '2'))
    reals = [_shard(r, dev) for r in range(2)]
    streams = []
    for r in range(2):
        torch.manual_seed(1000 + r)
        streams.append(torch.get_rng_state())
    losses = [[], []]
    # the same kind of gradient-scratch arenas the trainer uses (one per half-step and shard), persistent across iterations
    arenas = {(h, r): ZeroArena() for h in 'DG' for r in range(2)}
    with rng.cpu_stream():
        for it in range(ITERS):
            opt_G.zero_grad(set_to_none=True)
            opt_D.zero_grad(set_to_none=True)
            with cached_weights():
                dls = []
                for r in range(NS):
                    torch.set_rng_state(streams[r])
                    with zero_arena(arenas['D', r], dev):
                        dls.append(float(step._d_half(reals[r], it)))
                    streams[r] = torch.get_rng_state()
                if os.environ.get('AGF_DP_TEST_MANUAL'):
                    pass
                for p in D.parameters():
                    if p.grad is not None:
                        p.grad.div_(2)
                opt_D.step()
                invalidate_cached(D.parameters())
                for p in D.parameters():
                    p.requires_grad_(False)
                gls = []
                for r in range(NS):
                    torch.set_rng_state(streams[r])
                    with zero_arena(arenas['G', r], dev):
                        gls.append(float(step._g_half(reals[r], it)[0]))
                    streams[r] = torch.get_rng_state()
                for p in D.parameters():
                    p.requires_grad_(True)
            for p in G.parameters():
                if p.grad is not None:
                    p.grad.div_(2)
            opt_G.step()
            update_ema(G, G_ema)
            step.batches_done += 1
            for r in range(NS):
                losses[r].append((dls[r], gls[r]))
    return G, G_ema, D, losses


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
def test_two_ranks_on_one_gpu_match_the_accumulated_single_process_run(tmp_path, dtype):
    import torch.multiprocessing as mp
    out = str(tmp_path / 'dp')
    mp.start_processes(_worker, args=(2, _free_port(), out, dtype), nprocs=2, join=True, start_method='spawn')
    st = [torch.load(f'{out}.{r}') for r in range(2)]
    # replicas are identical ...
    for name in ('G', 'D', 'G_ema'):
        for k in st[0][name]:
            assert torch.equal(st[0][name][k], st[1][name][k]), (name, k)
    # ... and equal to the single-process run that accumulates the two shards' gradients
    G, G_ema, D, losses = _single_process(torch.device('cuda', 0), dtype)
    print('losses 2-rank run :', st[0]['losses'], st[1]['losses'])
    print('losses one process:', losses[0], losses[1])
    ltol = 1e-4 if dtype == torch.float32 else 2e-2
    for r in range(2):
        for (dl, gl), (dl2, gl2) in zip(st[r]['losses'], losses[r]):
            assert abs(dl - dl2) <= ltol * max(1.0, abs(dl2)) and abs(gl - gl2) <= ltol * max(1.0, abs(gl2)), (r, st[r]['losses'], losses[r])
    lr, worst, n_far, n_all = 1e-3, 0.0, 0, 0
    n_far32 = n_all32 = 0
    for name, mod in (('G', G), ('D', D), ('G_ema', G_ema)):
        for k, v in mod.state_dict().items():
            a, b = st[0][name][k].float(), v.detach().float().cpu()
            diff = (a - b).abs()
            worst = max(worst, float(diff.max()))
            if dtype == torch.float32:
                # typically ~1e-5 (summation order only); a near-zero gradient that changes sign costs a single weight a few lr
                assert diff.max() <= 8e-3, (name, k, float(diff.max()))
                n_far32 += int((diff > 1e-4).sum())
                n_all32 += diff.numel()
            else:
                # three Adam steps with beta1 = 0 move a weight by at most (1 + 1.41 + 1.72) lr in either direction
                assert diff.max() <= 2 * 4.2 * lr, (name, k, float(diff.max()))
                n_far += int((diff > 0.1 * lr).sum())
                n_all += diff.numel()
    print(f'{dtype}: largest weight difference between the 2-rank run and the accumulated single-process run: {worst:.3g}'
          + (f'; {n_far} of {n_all} weights differ by more than 0.1 lr' if n_all else ''))
    if n_all:
        assert n_far <= 0.2 * n_all, (n_far, n_all)
    if n_all32:
        assert n_far32 <= 1e-3 * n_all32 + 2, (n_far32, n_all32)


def _worker_graphs(rank, world, port, out, graphed):
    """Two ranks on GPU 0; fp32; the device generator drives every draw (graph-safe).  ``graphed``: iterations 2.. replayed from three
    HIP graphs per iteration kind with the bucket all-reduce between the launches; else the eager loop with ``GradReducer.finish()``."""
    import sys
    import functools
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      AGF_SINGLE_DEVICE='1', AGF_DIST_BACKEND='gloo')
    from animeface_amd import distributed as dp
    from animeface_amd.implementations.StyleGAN2 import model as M, utils as U
    from animeface_amd.nnutils import sample_nnoise, update_ema
    dp.init_distributed()
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    mk = lambda: M.Generator(CFG['image_size'], 3, CFG['style_dim'], CFG['channels'], CFG['max_channels'], 2, CFG['map_num_layers'], True, 0.01,
                             compute_dtype=torch.float32)
    G, G_ema = mk().to(dev), mk().to(dev)
    D = M.Discriminator(CFG['image_size'], 3, CFG['channels'], CFG['max_channels'], 2, 4, compute_dtype=torch.float32).to(dev)
    G.init_weight(functools.partial(M.init_weight_N01, lr=0.01), M.init_weight_N01)
    D.apply(M.init_weight_N01)
    G_ema.eval()
    update_ema(G, G_ema, decay=0)
    dp.broadcast_module(G), dp.broadcast_module(G_ema), dp.broadcast_module(D)
    opt_G, opt_D = U.build_optimizers(G, D, 0.001, (0., 0.99), 10., 0., D_K, 8, capturable=True)
    red_G = dp.GradReducer(G.parameters(), bucket_bytes=1 << 18, never_used=dp.never_used_parameters(G))
    red_D = dp.GradReducer(D.parameters(), bucket_bytes=1 << 18)
    step = U.TrainStep(G, G_ema, D, opt_G, opt_D, 10., 0., D_K, 8, 'color,translation', CFG['style_dim'],
                       functools.partial(sample_nnoise, device=dev), red_G, red_D)
    real = _shard(rank, dev)
    torch.manual_seed(1000 + rank)
    for _ in range(2):
        step(real)
    runner = U.GraphedTrainStep(step, real, warmup=0) if graphed else step
    losses = []
    for _ in range(4):                                     # d_k = 2: both iteration kinds, each captured once and replayed once
        dl, gl, _ = runner(real)
        losses.append((float(dl), float(gl)))
    torch.cuda.synchronize()
    if graphed:
        assert runner.segmented and runner.kinds() == {'gan', 'r1'}
    for m in (G, G_ema, D):
        dp.check_replica_consistency(m)
    scale = dp.never_used_parameters(G)[0]
    assert not opt_G.state.get(scale), 'Adam stepped a parameter that never received a gradient'
    torch.save(dict(G={k: v.cpu() for k, v in G.state_dict().items()}, D={k: v.cpu() for k, v in D.state_dict().items()},
                    G_ema={k: v.cpu() for k, v in G_ema.state_dict().items()}, losses=losses), f'{out}.{int(graphed)}.{rank}')
    dp.dist.barrier()
    dp.dist.destroy_process_group()


def test_graph_replay_under_data_parallelism_equals_the_eager_exchange(tmp_path):
    """GraphedTrainStep with reducers: four graphs per iteration kind, the bucket all-reduces between the launches (D's beside the generator forward of the G half-step) -- against
    the eager two-rank loop from the same seeds (fp32: weights to summation noise)."""
    import torch.multiprocessing as mp
    out = str(tmp_path / 'dpg')
    for graphed in (False, True):
        mp.start_processes(_worker_graphs, args=(2, _free_port(), out, graphed), nprocs=2, join=True, start_method='spawn')
    eager, graph = torch.load(f'{out}.0.0'), torch.load(f'{out}.1.0')
    print('losses eager :', eager['losses'])
    print('losses graphs:', graph['losses'])
    for (d0, g0), (d1, g1) in zip(eager['losses'], graph['losses']):
        assert abs(d0 - d1) <= 1e-4 * max(1.0, abs(d0)) and abs(g0 - g1) <= 1e-4 * max(1.0, abs(g0))
    worst, far, total = 0.0, 0, 0
    for name in ('G', 'D', 'G_ema'):
        for k in eager[name]:
            diff = (eager[name][k].float() - graph[name][k].float()).abs()
            worst = max(worst, float(diff.max()))
            far += int((diff > 1e-4).sum())
            total += diff.numel()
    print(f'largest weight difference graph-replayed vs eager two-rank run: {worst:.3g}; {far} of {total} weights differ by more than 1e-4')
    # fp32: the two runs differ by summation order only (rocBLAS split-K / fp32 atomics); six Adam steps of lr 1e-3 with beta1 = 0 turn a
    # gradient that changes sign near zero into a +-lr step, so single weights may sit up to a few lr apart -- but only a handful
    assert worst <= 6e-3 and far <= 1e-3 * total, (worst, far, total)          # (one run in four takes another of the 2-3 run-to-run outcomes: ~50 of 390 000 weights)


def _worker_rccl_ingraph(rank, world, port, out, mode):
    """ONE rank, fp32, device generator.  mode: 'single' = no process group, the single-process single-graph replay; 'ingraph' = a one-rank
    RCCL group (AGF_FORCE_DP=1) with the bucket all-reduces recorded into the graph from the backward hooks; 'segmented' = the same group
    with the iteration cut into four graphs at the two exchanges."""
    import sys
    import functools
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    if mode != 'single':
        os.environ['AGF_FORCE_DP'] = '1'
    from animeface_amd import distributed as dp
    from animeface_amd.implementations.StyleGAN2 import model as M, utils as U
    from animeface_amd.nnutils import sample_nnoise, update_ema
    dp.init_distributed()
    assert dp.dist.is_initialized() == (mode != 'single')
    if mode != 'single':
        assert dp.dist.get_backend() == 'nccl'
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    mk = lambda: M.Generator(CFG['image_size'], 3, CFG['style_dim'], CFG['channels'], CFG['max_channels'], 2, CFG['map_num_layers'], True, 0.01,
                             compute_dtype=torch.float32)
    G, G_ema = mk().to(dev), mk().to(dev)
    D = M.Discriminator(CFG['image_size'], 3, CFG['channels'], CFG['max_channels'], 2, 4, compute_dtype=torch.float32).to(dev)
    G.init_weight(functools.partial(M.init_weight_N01, lr=0.01), M.init_weight_N01)
    D.apply(M.init_weight_N01)
    G_ema.eval()
    update_ema(G, G_ema, decay=0)
    opt_G, opt_D = U.build_optimizers(G, D, 0.001, (0., 0.99), 10., 0., D_K, 8, capturable=True)
    red_G = red_D = None
    if mode != 'single':
        red_G = dp.GradReducer(G.parameters(), bucket_bytes=1 << 18, never_used=dp.never_used_parameters(G))
        red_D = dp.GradReducer(D.parameters(), bucket_bytes=1 << 18)
        assert red_G.collectives and red_G.capturable and len(red_G.buckets) >= 3
    step = U.TrainStep(G, G_ema, D, opt_G, opt_D, 10., 0., D_K, 8, 'color,translation', CFG['style_dim'],
                       functools.partial(sample_nnoise, device=dev), red_G, red_D)
    real = _shard(0, dev)
    torch.manual_seed(1000)
    for _ in range(2):
        step(real)
    import datetime
    mark = lambda what: print(datetime.datetime.now().strftime('%H:%M:%S.%f'), mode, what, flush=True)
    mark('eager done')
    runner = U.GraphedTrainStep(step, real, warmup=0, dp_mode=None if mode == 'single' else mode)
    if mode == 'ingraph':
        assert runner.dp_mode == 'ingraph' and not runner.segmented
    mark('capture begins')
    runner.capture_all()
    mark('captured')
    losses = []
    for _ in range(6):
        dl, gl, _ = runner(real)
        losses.append((float(dl), float(gl)))
        mark('replayed')
    torch.cuda.synchronize()
    if mode == 'ingraph':
        # every bucket except the never-used one (and, on lazy-R1 iterations, the one holding D's last bias) was launched from a hook
        # while the backward pass was being recorded, i.e. its all-reduce sits beside the rest of backward inside the graph
        rep = red_G.overlap_report()
        assert rep['steps'] == 4 and rep['buckets_from_hooks'] >= (len(red_G.buckets) - 1) * rep['steps'], rep      # 2 eager + 2 recorded
        scale = dp.never_used_parameters(G)[0]
        assert not opt_G.state.get(scale), 'Adam stepped a parameter that never received a gradient'
    torch.save(dict(G={k: v.cpu() for k, v in G.state_dict().items()}, D={k: v.cpu() for k, v in D.state_dict().items()},
                    G_ema={k: v.cpu() for k, v in G_ema.state_dict().items()}, losses=losses), f'{out}.{mode}')
    if mode != 'single':
        dp.dist.barrier()
        dp.dist.destroy_process_group()


def test_rccl_all_reduce_recorded_inside_the_graph_equals_the_single_process_replay(tmp_path):
    """The one-graph data-parallel mode: ONE HIP graph per iteration kind, the bucket all-reduces recorded on the RCCL stream from the backward
    hooks.  A one-rank RCCL group (AGF_FORCE_DP=1) exercises process group, hooks, capture of the collectives and replay on a single GPU;
    averaged over one rank the exchange is the identity, so losses and weights must equal the single-process replay from the same seeds
    (fp32: summation noise), and the segmented mode on the same group."""
    import torch.multiprocessing as mp
    out = str(tmp_path / 'rccl')
    for mode in ('single', 'ingraph', 'segmented'):
        mp.start_processes(_worker_rccl_ingraph, args=(1, _free_port(), out, mode), nprocs=1, join=True, start_method='spawn')
    ref = torch.load(f'{out}.single')
    for mode in ('ingraph', 'segmented'):
        got = torch.load(f'{out}.{mode}')
        print(f'losses single : {ref["losses"]}\nlosses {mode}: {got["losses"]}')
        for (d0, g0), (d1, g1) in zip(ref['losses'], got['losses']):
            assert abs(d0 - d1) <= 1e-4 * max(1.0, abs(d0)) and abs(g0 - g1) <= 1e-4 * max(1.0, abs(g0))
        worst, far, total = 0.0, 0, 0
        for name in ('G', 'D', 'G_ema'):
            for k in ref[name]:
                diff = (ref[name][k].float() - got[name][k].float()).abs()
                worst = max(worst, float(diff.max()))
                far += int((diff > 1e-4).sum())
                total += diff.numel()
        print(f'{mode}: largest weight difference vs the single-process replay {worst:.3g}; {far} of {total} beyond 1e-4')
        assert worst <= 8e-3 and far <= 1e-3 * total, (mode, worst, far, total)
